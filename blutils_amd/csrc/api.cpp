// C-ABI entry point of the hot path (include/blu_consensus.h): validates the
// hit table, stages host buffers when asked to, launches the HIP kernel.
// There is no CPU implementation behind this call: without a HIP device it
// fails with BLU_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "blu_internal.h"

using namespace blu;

#define HIP_TRY(expr)                                                              \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) {                                                    \
            set_error("%s failed: %s", #expr, hipGetErrorString(_e));              \
            rc = BLU_ERR_HIP;                                                      \
            goto done;                                                             \
        }                                                                          \
    } while (0)

extern "C" {

int blu_consensus_run(const blu_taxonomy* tax, const blu_hits* hits, const blu_run_params* params,
                      blu_result* out) {
    if (!tax || !hits || !params) { set_error("null argument"); return BLU_ERR_INVALID_ARG; }
    if (tax->device < 0) { set_error("host-only taxonomy handle: blu_consensus_run needs a HIP device (no CPU fallback)"); return BLU_ERR_NO_DEVICE; }
    if (params->strategy != BLU_CAUTIOUS && params->strategy != BLU_RELAXED) { set_error("unknown strategy %d", params->strategy); return BLU_ERR_INVALID_ARG; }
    if (hits->n_hits >= 0xFFFFFFFFull) { set_error("n_hits must be < 2^32 - 1 per call"); return BLU_ERR_INVALID_ARG; }
    if (hits->n_queries == 0) return BLU_OK;
    if (!out || !hits->seg_off) { set_error("null output or seg_off"); return BLU_ERR_INVALID_ARG; }
    if (hits->n_hits && (!hits->bitscore || !hits->tax_row || !hits->align_len || !hits->acc_rank)) {
        set_error("null hit column"); return BLU_ERR_INVALID_ARG;
    }
    if (hits->n_hits && ((hits->pident != nullptr) == (hits->pident_milli != nullptr))) {
        set_error("exactly one of pident / pident_milli must be given"); return BLU_ERR_INVALID_ARG;
    }
    if (hipSetDevice(tax->device) != hipSuccess) { set_error("hipSetDevice(%d) failed", tax->device); return BLU_ERR_NO_DEVICE; }

    TaxDev td{tax->d_lin, tax->d_codes, tax->sc, tax->d_lcp8, tax->d_rmq, tax->rmq_nb, tax->d_cutvals, tax->n_cutvals, tax->n_tax, tax->dev_stride, tax->node_base, tax->max_depth};
    if (tax->ws_capacity < hits->n_queries || !tax->ws_count) {
        // grows only when a larger table than any before arrives (first call): not graph-capturable
        if (tax->ws_worklist) (void)hipFree(tax->ws_worklist);
        if (!tax->ws_count) {
            if (hipMalloc((void**)&tax->ws_count, 256) != hipSuccess) { set_error("hipMalloc(workspace) failed"); return BLU_ERR_ALLOC; }
            if (hipMemset(tax->ws_count, 0, 256) != hipSuccess) { set_error("hipMemset(workspace) failed"); return BLU_ERR_HIP; }
        }
        tax->ws_worklist = nullptr;
        tax->ws_capacity = 0;
        if (hipMalloc((void**)&tax->ws_worklist, (hits->n_queries + 64) * sizeof(uint32_t)) != hipSuccess) { set_error("hipMalloc(workspace) failed"); return BLU_ERR_ALLOC; }
        tax->ws_capacity = hits->n_queries;
    }
    if (hits->on_device) {
        HitsDev hd{hits->bitscore, hits->tax_row, hits->pident, hits->pident_milli, hits->align_len, hits->acc_rank, hits->seg_off,
                   hits->n_hits, hits->n_queries};
        return launch_consensus(td, hd, params->strategy, out, params->stream, tax->device, tax->num_cus, tax->ws_worklist, tax->ws_count);
    }

    // host pointers: stage over PCIe, run, copy the records back (synchronous)
    int rc = BLU_OK;
    hipStream_t s = (hipStream_t)params->stream;
    const size_t nh = hits->n_hits, nq = hits->n_queries;
    void *d_bs = nullptr, *d_tax = nullptr, *d_pid = nullptr, *d_aln = nullptr, *d_acc = nullptr, *d_seg = nullptr, *d_out = nullptr;
    {
        const size_t pad = 64;  // keeps zero-length columns allocatable
        HIP_TRY(hipMalloc(&d_bs, nh * 4 + pad));
        HIP_TRY(hipMalloc(&d_tax, nh * 4 + pad));
        const bool milli = hits->pident_milli != nullptr;
        HIP_TRY(hipMalloc(&d_pid, nh * (milli ? 4 : 8) + pad));
        HIP_TRY(hipMalloc(&d_aln, nh * 4 + pad));
        HIP_TRY(hipMalloc(&d_acc, nh * 4 + pad));
        HIP_TRY(hipMalloc(&d_seg, (nq + 1) * 8));
        HIP_TRY(hipMalloc(&d_out, nq * sizeof(blu_result)));
        if (nh) {
            HIP_TRY(hipMemcpyAsync(d_bs, hits->bitscore, nh * 4, hipMemcpyHostToDevice, s));
            HIP_TRY(hipMemcpyAsync(d_tax, hits->tax_row, nh * 4, hipMemcpyHostToDevice, s));
            if (milli) HIP_TRY(hipMemcpyAsync(d_pid, hits->pident_milli, nh * 4, hipMemcpyHostToDevice, s));
            else HIP_TRY(hipMemcpyAsync(d_pid, hits->pident, nh * 8, hipMemcpyHostToDevice, s));
            HIP_TRY(hipMemcpyAsync(d_aln, hits->align_len, nh * 4, hipMemcpyHostToDevice, s));
            HIP_TRY(hipMemcpyAsync(d_acc, hits->acc_rank, nh * 4, hipMemcpyHostToDevice, s));
        }
        HIP_TRY(hipMemcpyAsync(d_seg, hits->seg_off, (nq + 1) * 8, hipMemcpyHostToDevice, s));
        HitsDev hd{(const int32_t*)d_bs, (const uint32_t*)d_tax, milli ? nullptr : (const double*)d_pid,
                   milli ? (const uint32_t*)d_pid : nullptr, (const int32_t*)d_aln,
                   (const uint32_t*)d_acc, (const uint64_t*)d_seg, hits->n_hits, hits->n_queries};
        rc = launch_consensus(td, hd, params->strategy, (blu_result*)d_out, s, tax->device, tax->num_cus, tax->ws_worklist, tax->ws_count);
        if (rc != BLU_OK) goto done;
        HIP_TRY(hipMemcpyAsync(out, d_out, nq * sizeof(blu_result), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
    }
done:
    if (d_bs) (void)hipFree(d_bs);
    if (d_tax) (void)hipFree(d_tax);
    if (d_pid) (void)hipFree(d_pid);
    if (d_aln) (void)hipFree(d_aln);
    if (d_acc) (void)hipFree(d_acc);
    if (d_seg) (void)hipFree(d_seg);
    if (d_out) (void)hipFree(d_out);
    return rc;
}

int blu_shard_ranges(const uint64_t* seg_off, uint64_t n_queries, uint32_t n_shards, uint64_t* bounds) {
    if (!seg_off || !bounds || n_shards == 0) { set_error("null argument"); return BLU_ERR_INVALID_ARG; }
    const uint64_t base = seg_off[0];
    const uint64_t total = seg_off[n_queries] >= base ? seg_off[n_queries] - base : 0;
    bounds[0] = 0;
    for (uint32_t p = 1; p < n_shards; ++p) {
        const uint64_t target = base + (uint64_t)((unsigned __int128)total * p / n_shards);
        const uint64_t* it = std::lower_bound(seg_off, seg_off + n_queries + 1, target);
        uint64_t q = (uint64_t)(it - seg_off);
        if (q > n_queries) q = n_queries;
        // the boundary nearest to the target, never moving backwards
        if (q > 0 && target - seg_off[q - 1] <= seg_off[q] - target) --q;
        bounds[p] = std::max(q, bounds[p - 1]);
    }
    bounds[n_shards] = n_queries;
    return BLU_OK;
}

int blu_consensus_run_multi(const blu_taxonomy* const* taxes, uint32_t n, const blu_hits* hits, const blu_run_params* params,
                            blu_result* out) {
    if (!taxes || n == 0 || !hits || !params) { set_error("null argument"); return BLU_ERR_INVALID_ARG; }
    if (hits->on_device) { set_error("blu_consensus_run_multi takes host pointers"); return BLU_ERR_INVALID_ARG; }
    for (uint32_t i = 0; i < n; ++i) {
        if (!taxes[i]) { set_error("null taxonomy handle"); return BLU_ERR_INVALID_ARG; }
        for (uint32_t j = 0; j < i; ++j)
            if (taxes[j] == taxes[i]) { set_error("a handle may appear once (its scratch is per handle)"); return BLU_ERR_INVALID_ARG; }
        if (taxes[i]->n_tax != taxes[0]->n_tax) { set_error("handles of different taxonomies"); return BLU_ERR_INVALID_ARG; }
    }
    if (hits->n_queries == 0) return BLU_OK;
    if (!out || !hits->seg_off) { set_error("null output or seg_off"); return BLU_ERR_INVALID_ARG; }
    for (uint64_t q = 0; q < hits->n_queries; ++q)
        if (hits->seg_off[q] > hits->seg_off[q + 1]) { set_error("seg_off must be ascending for a sharded run"); return BLU_ERR_INVALID_ARG; }
    std::vector<uint64_t> bounds(n + 1);
    int rc = blu_shard_ranges(hits->seg_off, hits->n_queries, n, bounds.data());
    if (rc != BLU_OK) return rc;
    std::vector<int> rcs(n, BLU_OK);
    std::vector<std::string> errs(n);
    std::vector<std::thread> pool;
    for (uint32_t i = 0; i < n; ++i)
        pool.emplace_back([&, i]() {
            const uint64_t q0 = bounds[i], q1 = bounds[i + 1];
            if (q1 == q0) return;
            const uint64_t r0 = hits->seg_off[q0], r1 = hits->seg_off[q1];
            std::vector<uint64_t> seg(q1 - q0 + 1);
            for (uint64_t q = q0; q <= q1; ++q) seg[q - q0] = hits->seg_off[q] - r0;      // offsets rebased to the slice
            blu_hits h = *hits;
            h.bitscore = hits->bitscore + r0; h.tax_row = hits->tax_row + r0; h.align_len = hits->align_len + r0;
            h.acc_rank = hits->acc_rank + r0;
            h.pident = hits->pident ? hits->pident + r0 : nullptr;
            h.pident_milli = hits->pident_milli ? hits->pident_milli + r0 : nullptr;
            h.seg_off = seg.data(); h.n_hits = r1 - r0; h.n_queries = q1 - q0;
            blu_run_params p = *params;
            p.stream = nullptr;                                                          // each shard on its device's null stream
            rcs[i] = blu_consensus_run(taxes[i], &h, &p, out + q0);
            if (rcs[i] != BLU_OK) { char b[512]; blu_last_error(b, sizeof b); errs[i] = b; return; }
            for (uint64_t q = q0; q < q1; ++q)                                           // reference rows -> rows of the whole table
                if (out[q].ref_row != 0xFFFFFFFFu) out[q].ref_row += (uint32_t)r0;
        });
    for (auto& th : pool) th.join();
    for (uint32_t i = 0; i < n; ++i)
        if (rcs[i] != BLU_OK) { set_error("shard %u: %s", i, errs[i].c_str()); return rcs[i]; }
    return BLU_OK;
}

int blu_consensus_last_launch(char* kernel_name, size_t len, uint32_t* grid, uint32_t* block) {
    if (kernel_name && len) snprintf(kernel_name, len, "%s", consensus_kernel_name());
    consensus_last_geometry(grid, block);
    return BLU_OK;
}

}  // extern "C"
