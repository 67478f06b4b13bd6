// C-ABI entry point of the hot path (include/blu_consensus.h): validates the
// hit table, stages host buffers when asked to, launches the HIP kernel.
// There is no CPU implementation behind this call: without a HIP device it
// fails with BLU_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include <algorithm>
#include <string>
#include <thread>
#include <vector>

#include "blu_internal.h"

using namespace blu;

#ifndef BLU_STAGE_KEEP_BYTES
#define BLU_STAGE_KEEP_BYTES (4ull << 30)   // staging buffers the handle keeps between host-pointer calls (a 200 M-row packed table)
#endif

#define HIP_TRY(expr)                                                              \
    do {                                                                           \
        hipError_t _e = (expr);                                                    \
        if (_e != hipSuccess) {                                                    \
            set_error("%s failed: %s", #expr, hipGetErrorString(_e));              \
            rc = BLU_ERR_HIP;                                                      \
            goto done;                                                             \
        }                                                                          \
    } while (0)

// The stream-kernel kind the handle's last table wanted, as far as the device has reported it (pinned word, read without
// synchronisation); 0 = classify on the device again: the first calls, and every 64th.
// The remembered kind — and the "its last run queued next to nothing" bit that lets a call skip the worklist kernel — belong
// to ONE table, recognised by the device addresses of its offsets, bit-scores and side values and by its query and row counts.
// staged = the table sits in the handle's own staging buffers (host-pointer path): those addresses are the same for every
// host table, so the kind is still remembered (a stale kind costs time only — and never much: both kinds take every table)
// but the worklist kernel is always launched.
static uint32_t known_kind(const blu_taxonomy* tax, const HitsDev& h, bool staged) {
    // BLU_STREAM_KIND=ring | noring: that build for every table (tests: both builds must give the same records on any table)
    uint32_t forced_kind = 0;
    if (const char* env = getenv("BLU_STREAM_KIND")) {
        if (strcmp(env, "ring") == 0) forced_kind = 1u;
        if (strcmp(env, "noring") == 0) forced_kind = 2u;
    }
    const uint64_t call = __atomic_fetch_add(&tax->ws_calls, 1, __ATOMIC_RELAXED);
    const void* side = h.packed ? (const void*)h.packed : h.packed64 ? (const void*)h.packed64 : (const void*)h.tax_row;
    const bool same_table = tax->ws_kind_key_ptr[0] == h.seg_off && tax->ws_kind_key_ptr[1] == h.bitscore && tax->ws_kind_key_ptr[2] == side &&
                            tax->ws_kind_key_n[0] == h.n_queries && tax->ws_kind_key_n[1] == h.n_hits;
    if (!tax->ws_kind_host || (call & 63u) == 0 || !same_table) {
        if (tax->ws_kind_host) {   // (until the device reports this table's kind and queue length)
            __atomic_store_n(tax->ws_kind_host, 0u, __ATOMIC_RELAXED);
            __atomic_store_n(tax->ws_kind_host + 1, 0xFFFFFFFFu, __ATOMIC_RELAXED);
        }
        tax->ws_kind_key_ptr[0] = h.seg_off; tax->ws_kind_key_ptr[1] = h.bitscore; tax->ws_kind_key_ptr[2] = side;
        tax->ws_kind_key_n[0] = h.n_queries; tax->ws_kind_key_n[1] = h.n_hits;
        return forced_kind;
    }
    const uint32_t kind = forced_kind ? forced_kind : __atomic_load_n(tax->ws_kind_host, __ATOMIC_RELAXED);
    // the queue length the table's last run reported: next to nothing -> no launch of the worklist kernel (BLU_NO_TAIL=1: always launch it)
    const uint32_t last_len = __atomic_load_n(tax->ws_kind_host + 1, __ATOMIC_RELAXED);
    static const bool never = getenv("BLU_NO_TAIL") != nullptr;
    return kind | ((last_len <= 32u && !never && !staged) ? 4u : 0u);
}

extern "C" {

int blu_consensus_run(const blu_taxonomy* tax, const blu_hits* hits, const blu_run_params* params,
                      blu_result* out) {
    if (!tax || !hits || !params) { set_error("null argument"); return BLU_ERR_INVALID_ARG; }
    if (tax->device < 0) { set_error("host-only taxonomy handle: blu_consensus_run needs a HIP device (no CPU fallback)"); return BLU_ERR_NO_DEVICE; }
    if (params->strategy != BLU_CAUTIOUS && params->strategy != BLU_RELAXED) { set_error("unknown strategy %d", params->strategy); return BLU_ERR_INVALID_ARG; }
    if (hits->n_hits >= 0xFFFFFFFFull) { set_error("n_hits must be < 2^32 - 1 per call"); return BLU_ERR_INVALID_ARG; }
    if (hits->n_queries == 0) return BLU_OK;
    if (hits->n_queries >= 0xFFFFFFFFull) { set_error("n_queries must be < 2^32 - 1 per call"); return BLU_ERR_INVALID_ARG; }   // (worklist entries and task arithmetic are 32-bit)
    if (!out || !hits->seg_off) { set_error("null output or seg_off"); return BLU_ERR_INVALID_ARG; }
    if (hits->n_hits && (!hits->bitscore || (!hits->packed && !hits->packed64 && (!hits->tax_row || !hits->align_len || !hits->acc_rank)))) {
        set_error("null hit column"); return BLU_ERR_INVALID_ARG;
    }
    if (hits->n_hits && ((hits->pident != nullptr) + (hits->pident_milli != nullptr) + (hits->packed != nullptr) + (hits->packed64 != nullptr) != 1)) {
        set_error("exactly one of pident / pident_milli / packed / packed64 must be given"); return BLU_ERR_INVALID_ARG;
    }
    if (hits->packed && ((uintptr_t)hits->packed & 15u)) { set_error("packed records must be 16-byte aligned"); return BLU_ERR_INVALID_ARG; }
    if (hits->packed64 && ((uintptr_t)hits->packed64 & 7u)) { set_error("packed64 records must be 8-byte aligned"); return BLU_ERR_INVALID_ARG; }
    if (hipSetDevice(tax->device) != hipSuccess) { set_error("hipSetDevice(%d) failed", tax->device); return BLU_ERR_NO_DEVICE; }

    TaxDev td{tax->d_lin, tax->d_codes, tax->d_kthr, tax->sc, tax->d_lcp8, tax->d_rmq, tax->rmq_nb, tax->d_cutvals, tax->n_cutvals, tax->n_tax, tax->dev_stride, tax->node_base, tax->max_depth, std::max<uint32_t>(tax->n_shapes, 1u),
               tax->d_wblk, tax->d_wchain, tax->d_wchain_hi, tax->wide_levels};
    if (tax->ws_capacity < hits->n_queries || !tax->ws_count) {
        // grows only when a larger table than any before arrives (first call): not graph-capturable
        if (tax->ws_worklist) (void)hipFree(tax->ws_worklist);
        if (!tax->ws_count) {
            // (counters of the run, then the 64 worklist queues' counters, one memory line each: consensus_kernel.hip WL_BASE / WL_STRIDE)
            if (hipMalloc((void**)&tax->ws_count, 16384) != hipSuccess) { set_error("hipMalloc(workspace) failed"); return BLU_ERR_ALLOC; }
            if (hipMemset(tax->ws_count, 0, 16384) != hipSuccess) { set_error("hipMemset(workspace) failed"); return BLU_ERR_HIP; }
            // (without the pinned word every call classifies its table on the device: slower by two kernel boundaries, not wrong)
            if (hipHostMalloc((void**)&tax->ws_kind_host, 64, hipHostMallocDefault) == hipSuccess) {
                tax->ws_kind_host[0] = 0; tax->ws_kind_host[1] = 0xFFFFFFFFu;
                if (hipHostGetDevicePointer((void**)&tax->ws_kind_dev, tax->ws_kind_host, 0) != hipSuccess) tax->ws_kind_dev = nullptr;
            } else { (void)hipGetLastError(); tax->ws_kind_host = nullptr; }
        }
        tax->ws_worklist = nullptr;
        tax->ws_capacity = 0;
        if (hipMalloc((void**)&tax->ws_worklist, (hits->n_queries + 8192) * sizeof(uint32_t)) != hipSuccess) { set_error("hipMalloc(workspace) failed"); return BLU_ERR_ALLOC; }
        tax->ws_capacity = hits->n_queries;
    }
    if (hits->on_device) {
        HitsDev hd{hits->bitscore, hits->tax_row, hits->pident, hits->pident_milli, hits->packed, hits->align_len, hits->acc_rank, hits->seg_off,
                   hits->n_hits, hits->n_queries, hits->packed64};
        return launch_consensus(td, hd, params->strategy, out, params->stream, tax->device, tax->num_cus, tax->ws_worklist, tax->ws_count,
                                tax->ws_kind_dev, known_kind(tax, hd, false));
    }

    // host pointers: stage over PCIe, run, copy the records back (synchronous).  The table goes over in chunks of whole
    // queries, double-buffered: while the kernels of chunk k run on one stream, chunk k+1 is copied on the other (the
    // handle's worklist is shared, so the kernels themselves are chained by an event).  A table that fits is one chunk.
    int rc = BLU_OK;
    const size_t nh = hits->n_hits, nq = hits->n_queries;
    const bool milli = hits->pident_milli != nullptr, packed = hits->packed != nullptr, wide = hits->packed64 != nullptr;
    const size_t row_bytes = wide ? 28 : ((milli || packed) ? 20 : 24);
    size_t chunk_rows = nh;
    {
        size_t free_b = 0, total_b = 0;
        // slots this layout does not use are given back first (a columns call leaves slots 2..4 behind; counting them as room
        // for a packed table's larger slot 1 made its hipMalloc fail)
        for (int k = 0; k < 2; ++k)
            for (int slot = 2; slot <= 4; ++slot)
                if ((packed || wide) && tax->ws_stage[k][slot]) { (void)hipFree(tax->ws_stage[k][slot]); tax->ws_stage[k][slot] = nullptr; tax->ws_stage_bytes[k][slot] = 0; }
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            for (auto& set : tax->ws_stage_bytes) for (size_t b : set) free_b += b;   // (what the handle still holds is there to be reused)
            const size_t budget = free_b / 5 * 2 / row_bytes;          // two buffer sets within 80 % of what is free
            if (budget < chunk_rows) chunk_rows = budget;
        }
        if (const char* env = getenv("BLU_STAGE_ROWS")) { const size_t v = (size_t)strtoull(env, nullptr, 10); if (v && v < chunk_rows) chunk_rows = v; }
    }
    // chunk boundaries (whole queries, greedy); a table cut into chunks needs an ascending offset table
    std::vector<uint64_t> cuts{0};
    size_t max_rows = 0, max_q = 0;
    if (chunk_rows >= nh) { cuts.push_back(nq); max_rows = nh; max_q = nq; }
    else {
        for (uint64_t q = 0; q < nq; ++q)
            if (hits->seg_off[q] > hits->seg_off[q + 1] || hits->seg_off[q + 1] > nh) { set_error("seg_off must be ascending and within n_hits for a table staged in chunks"); return BLU_ERR_INVALID_ARG; }
        uint64_t q0 = 0;
        while (q0 < nq) {
            uint64_t q1 = q0 + 1;
            const uint64_t r0 = hits->seg_off[q0];
            // as many whole queries as fit; a single query longer than the chunk gets a chunk of its own
            const uint64_t* it = std::upper_bound(hits->seg_off + q0 + 1, hits->seg_off + nq + 1, r0 + chunk_rows);
            if ((uint64_t)(it - hits->seg_off) - 1 > q1) q1 = (uint64_t)(it - hits->seg_off) - 1;
            cuts.push_back(q1);
            max_rows = std::max<size_t>(max_rows, hits->seg_off[q1] - r0);
            max_q = std::max<size_t>(max_q, q1 - q0);
            q0 = q1;
        }
    }
    const size_t n_chunks = cuts.size() - 1;
    const int n_sets = n_chunks > 1 ? 2 : 1;
    struct Set { void *bs = nullptr, *tax = nullptr, *pid = nullptr, *aln = nullptr, *acc = nullptr, *seg = nullptr, *out = nullptr; hipStream_t s = nullptr; std::vector<uint64_t> seg_host; uint64_t q0 = 0, q1 = 0, r0 = 0; bool busy = false; };
    Set sets[2];
    hipEvent_t kernels_done = nullptr;
    auto finish = [&](Set& st) -> int {      // wait for the chunk in this set, point its records at rows of the whole table
        if (!st.busy) return BLU_OK;
        st.busy = false;
        if (hipStreamSynchronize(st.s) != hipSuccess) { set_error("HIP error while waiting for a staged chunk"); return BLU_ERR_HIP; }
        if (st.r0)
            for (uint64_t q = st.q0; q < st.q1; ++q)
                if (out[q].ref_row != 0xFFFFFFFFu) out[q].ref_row += (uint32_t)st.r0;
        return BLU_OK;
    };
    {
        const size_t pad = 64;  // keeps zero-length columns allocatable
        // the staging buffers live with the handle (grow-only): a caller that runs table after table pays for them once
        auto stage = [&](int k, int slot, size_t bytes, void** p) -> hipError_t {
            if (tax->ws_stage_bytes[k][slot] < bytes) {
                if (tax->ws_stage[k][slot]) (void)hipFree(tax->ws_stage[k][slot]);
                tax->ws_stage[k][slot] = nullptr; tax->ws_stage_bytes[k][slot] = 0;
                const hipError_t e = hipMalloc(&tax->ws_stage[k][slot], bytes);
                if (e != hipSuccess) return e;
                tax->ws_stage_bytes[k][slot] = bytes;
            }
            *p = tax->ws_stage[k][slot];
            return hipSuccess;
        };
        for (int k = 0; k < n_sets; ++k) {
            Set& st = sets[k];
            HIP_TRY(stage(k, 0, max_rows * 4 + pad, &st.bs));
            if (packed) HIP_TRY(stage(k, 1, max_rows * 16 + pad, &st.pid));   // the 16-byte records
            else if (wide) HIP_TRY(stage(k, 1, max_rows * 24 + pad, &st.pid));   // the 24-byte records
            else {
                HIP_TRY(stage(k, 2, max_rows * 4 + pad, &st.tax));
                HIP_TRY(stage(k, 1, max_rows * (milli ? 4 : 8) + pad, &st.pid));
                HIP_TRY(stage(k, 3, max_rows * 4 + pad, &st.aln));
                HIP_TRY(stage(k, 4, max_rows * 4 + pad, &st.acc));
            }
            HIP_TRY(stage(k, 5, (max_q + 1) * 8, &st.seg));
            HIP_TRY(stage(k, 6, std::max<size_t>(max_q, 1) * sizeof(blu_result), &st.out));
            if (n_chunks > 1) HIP_TRY(hipStreamCreateWithFlags(&st.s, hipStreamNonBlocking));
            else st.s = (hipStream_t)params->stream;
        }
        if (n_chunks > 1) HIP_TRY(hipEventCreateWithFlags(&kernels_done, hipEventDisableTiming));
        for (size_t c = 0; c < n_chunks; ++c) {
            Set& st = sets[c % n_sets];
            rc = finish(st);
            if (rc != BLU_OK) goto done;
            st.q0 = cuts[c]; st.q1 = cuts[c + 1];
            const uint64_t cq = st.q1 - st.q0;
            st.r0 = n_chunks > 1 ? hits->seg_off[st.q0] : 0;
            const uint64_t cr = n_chunks > 1 ? hits->seg_off[st.q1] - st.r0 : nh;
            const uint64_t* seg_src = hits->seg_off + st.q0;
            if (st.r0) {      // offsets rebased to the chunk
                st.seg_host.resize(cq + 1);
                for (uint64_t q = 0; q <= cq; ++q) st.seg_host[q] = hits->seg_off[st.q0 + q] - st.r0;
                seg_src = st.seg_host.data();
            }
            if (cr) {
                HIP_TRY(hipMemcpyAsync(st.bs, hits->bitscore + st.r0, cr * 4, hipMemcpyHostToDevice, st.s));
                if (packed) HIP_TRY(hipMemcpyAsync(st.pid, hits->packed + 4 * st.r0, cr * 16, hipMemcpyHostToDevice, st.s));
                else if (wide) HIP_TRY(hipMemcpyAsync(st.pid, hits->packed64 + 6 * st.r0, cr * 24, hipMemcpyHostToDevice, st.s));
                else {
                    HIP_TRY(hipMemcpyAsync(st.tax, hits->tax_row + st.r0, cr * 4, hipMemcpyHostToDevice, st.s));
                    if (milli) HIP_TRY(hipMemcpyAsync(st.pid, hits->pident_milli + st.r0, cr * 4, hipMemcpyHostToDevice, st.s));
                    else HIP_TRY(hipMemcpyAsync(st.pid, hits->pident + st.r0, cr * 8, hipMemcpyHostToDevice, st.s));
                    HIP_TRY(hipMemcpyAsync(st.aln, hits->align_len + st.r0, cr * 4, hipMemcpyHostToDevice, st.s));
                    HIP_TRY(hipMemcpyAsync(st.acc, hits->acc_rank + st.r0, cr * 4, hipMemcpyHostToDevice, st.s));
                }
            }
            HIP_TRY(hipMemcpyAsync(st.seg, seg_src, (cq + 1) * 8, hipMemcpyHostToDevice, st.s));
            if (n_chunks > 1 && c > 0) HIP_TRY(hipStreamWaitEvent(st.s, kernels_done, 0));   // the previous chunk's kernels own the worklist
            HitsDev hd{(const int32_t*)st.bs, (const uint32_t*)st.tax, (milli || packed || wide) ? nullptr : (const double*)st.pid,
                       milli ? (const uint32_t*)st.pid : nullptr, packed ? (const uint32_t*)st.pid : nullptr, (const int32_t*)st.aln,
                       (const uint32_t*)st.acc, (const uint64_t*)st.seg, cr, cq, wide ? (const uint32_t*)st.pid : nullptr};
            rc = launch_consensus(td, hd, params->strategy, (blu_result*)st.out, st.s, tax->device, tax->num_cus, tax->ws_worklist, tax->ws_count,
                                  tax->ws_kind_dev, known_kind(tax, hd, true));
            if (rc != BLU_OK) goto done;
            if (n_chunks > 1) HIP_TRY(hipEventRecord(kernels_done, st.s));
            HIP_TRY(hipMemcpyAsync(out + st.q0, st.out, cq * sizeof(blu_result), hipMemcpyDeviceToHost, st.s));
            st.busy = true;
        }
        for (int k = 0; k < n_sets && rc == BLU_OK; ++k) rc = finish(sets[k]);
    }
done:
    for (Set& st : sets) {
        if (st.busy && st.s) (void)hipStreamSynchronize(st.s);
        if (n_chunks > 1 && st.s) (void)hipStreamDestroy(st.s);
    }
    if (kernels_done) (void)hipEventDestroy(kernels_done);
    {
        // What the handle keeps for the next call: the buffers of a table that went over in ONE chunk, up to BLU_STAGE_KEEP_BYTES
        // in all.  A chunked run sized its buffers from (most of) the free memory of the card: keeping those would leave GPU
        // ingest, torch or a second handle in the same process with a nearly full device (blu_taxonomy_trim frees the rest).
        size_t held = 0;
        for (auto& set : tax->ws_stage_bytes) for (size_t b : set) held += b;
        if (n_chunks > 1 || held > BLU_STAGE_KEEP_BYTES || rc != BLU_OK) (void)blu_taxonomy_trim(tax);
    }
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
    if (hits->seg_off[hits->n_queries] > hits->n_hits) { set_error("seg_off runs past n_hits"); return BLU_ERR_INVALID_ARG; }
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
            h.bitscore = hits->bitscore + r0;
            h.tax_row = hits->tax_row ? hits->tax_row + r0 : nullptr; h.align_len = hits->align_len ? hits->align_len + r0 : nullptr;
            h.acc_rank = hits->acc_rank ? hits->acc_rank + r0 : nullptr;
            h.pident = hits->pident ? hits->pident + r0 : nullptr;
            h.pident_milli = hits->pident_milli ? hits->pident_milli + r0 : nullptr;
            h.packed = hits->packed ? hits->packed + 4 * r0 : nullptr;
            h.packed64 = hits->packed64 ? hits->packed64 + 6 * r0 : nullptr;
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
