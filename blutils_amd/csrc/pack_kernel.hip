// blu_hits_pack / blu_hits_pack64 (include/blu_consensus.h): the side records of the packed hit-table layouts from the
// four non-bit-score columns.  An ingest-time pass (like the taxid join that produced tax_row, mod.rs:72-76): it moves
// values next to each other and adds the shape hint — a function of the joined taxonomy row — and computes nothing of
// the consensus.  HBM-bound copy work: one 16- or 24-byte record written per hit.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>

#include "blu_internal.h"

namespace blu {
namespace {

// k = round(p * 1000) when fl(k / 1000.0) == p bit for bit (the milli-percent encoding is lossless exactly then)
__host__ __device__ inline bool exact_milli(double p, uint32_t* k_out) {
    bool ok = p >= 0.0 && p < 4.0e6;
    uint32_t k = 0;
    if (ok) {
        k = (uint32_t)(p * 1000.0 + 0.5);
        const double back = (double)k / 1000.0;
        uint64_t a, b;
        memcpy(&a, &back, 8); memcpy(&b, &p, 8);
        ok = a == b;
    }
    *k_out = k;
    return ok;
}

__host__ __device__ inline uint32_t hint_of(uint32_t engine_row, const uint16_t* hint_of_pos, uint64_t n_tax) {
    const uint32_t pos = engine_row & ((1u << BLU_ROW_BITS) - 1u);
    return (engine_row != BLU_UNMATCHED_TAXID && pos < n_tax) ? (uint32_t)hint_of_pos[pos] : 0u;
}

// one record; returns false when the 16-byte layout cannot hold the row's perc_identity
__host__ __device__ inline bool make16(uint32_t row, const double* pid, const uint32_t* pm, uint64_t i, const uint16_t* hint_of_pos,
                                       uint64_t n_tax, uint32_t* w1) {
    uint32_t k;
    bool ok = true;
    if (pm) k = pm[i];
    else ok = exact_milli(pid[i], &k);
    ok = ok && k < BLU_PACKED_PIDENT_LIMIT;
    *w1 = (k & BLU_KTHR_NEVER) | (hint_of(row, hint_of_pos, n_tax) << BLU_KTHR_BITS);
    return ok;
}

// Four rows per thread, a block stride apart: the sixteen column loads of a thread are in flight together (the one-row-per-thread
// form had 16 bytes in flight per lane and ran at 5 TB/s of its 32 B/row; this is a copy, it should run at the copy rate) and every
// load / store instruction of a wave still covers contiguous memory (4-byte columns: 256 B, records: 1 KiB).  Columns and records
// are touched once: non-temporal both ways.
#define PACK_ROWS 4
__global__ __launch_bounds__(256) void pack16_kernel(const uint32_t* __restrict__ tax_row, const double* __restrict__ pid,
                                                      const uint32_t* __restrict__ pm, const int32_t* __restrict__ aln,
                                                      const uint32_t* __restrict__ acc, uint64_t n, const uint16_t* __restrict__ hint_of_pos,
                                                      uint64_t n_tax, uint4* __restrict__ out, uint32_t* __restrict__ bad) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint64_t tile = (uint64_t)blockDim.x * PACK_ROWS;
    bool any_bad = false;
    for (uint64_t base = (uint64_t)blockIdx.x * tile; base < n; base += (uint64_t)gridDim.x * tile) {
        uint32_t row[PACK_ROWS], k[PACK_ROWS], al[PACK_ROWS], ac[PACK_ROWS];
        double pd[PACK_ROWS];
#pragma unroll
        for (int r = 0; r < PACK_ROWS; ++r) {
            const uint64_t i = base + (uint64_t)r * blockDim.x + threadIdx.x;
            const bool on = i < n;
            const uint64_t j = on ? i : 0;
            row[r] = __builtin_nontemporal_load(tax_row + j);
            if (pm) k[r] = __builtin_nontemporal_load(pm + j); else pd[r] = __builtin_nontemporal_load(pid + j);
            al[r] = (uint32_t)__builtin_nontemporal_load(aln + j);
            ac[r] = __builtin_nontemporal_load(acc + j);
        }
#pragma unroll
        for (int r = 0; r < PACK_ROWS; ++r) {
            const uint64_t i = base + (uint64_t)r * blockDim.x + threadIdx.x;
            if (i >= n) continue;
            bool ok = true;
            if (!pm) ok = exact_milli(pd[r], &k[r]);
            ok = ok && k[r] < BLU_PACKED_PIDENT_LIMIT;
            any_bad |= !ok;
            const u32x4 rec = {row[r], (k[r] & BLU_KTHR_NEVER) | (hint_of(row[r], hint_of_pos, n_tax) << BLU_KTHR_BITS), al[r], ac[r]};
            __builtin_nontemporal_store(rec, reinterpret_cast<u32x4*>(out) + i);
        }
    }
    if (any_bad) *bad = 1u;
}

__global__ __launch_bounds__(256) void pack24_kernel(const uint32_t* __restrict__ tax_row, const double* __restrict__ pid,
                                                      const uint32_t* __restrict__ pm, const int32_t* __restrict__ aln,
                                                      const uint32_t* __restrict__ acc, uint64_t n, const uint16_t* __restrict__ hint_of_pos,
                                                      uint64_t n_tax, uint2* __restrict__ out) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t row = tax_row[i];
    const double p = pid ? pid[i] : (double)pm[i] / 1000.0;   // (the engine's own conversion of a milli-percent value)
    out[3 * i] = make_uint2(row, hint_of(row, hint_of_pos, n_tax) << BLU_KTHR_BITS);
    out[3 * i + 1] = make_uint2((uint32_t)aln[i], acc[i]);
    out[3 * i + 2] = make_uint2((uint32_t)__double2loint(p), (uint32_t)__double2hiint(p));
}

int check_args(const blu_taxonomy* tax, const blu_hits* c, const void* out) {
    if (!tax || !c) { set_error("null argument"); return BLU_ERR_INVALID_ARG; }
    if (c->n_hits == 0) return BLU_OK;
    if (c->n_hits >= 0xFFFFFFFFull) { set_error("n_hits must be < 2^32 - 1 per call"); return BLU_ERR_INVALID_ARG; }   // (as blu_consensus_run; one thread per row: the grid stays below 2^31 blocks)
    if (!out || !c->tax_row || !c->align_len || !c->acc_rank || ((c->pident != nullptr) + (c->pident_milli != nullptr) != 1)) {
        set_error("blu_hits_pack needs tax_row, align_len, acc_rank and one of pident / pident_milli"); return BLU_ERR_INVALID_ARG;
    }
    if (c->on_device && tax->device < 0) { set_error("host-only taxonomy handle: device columns need a HIP device"); return BLU_ERR_NO_DEVICE; }
    return BLU_OK;
}

}  // namespace
}  // namespace blu

using namespace blu;

extern "C" {

int blu_hits_pack(const blu_taxonomy* tax, const blu_hits* c, uint32_t* out, void* stream) {
    int rc = check_args(tax, c, out);
    if (rc != BLU_OK || c->n_hits == 0) return rc;
    const uint64_t n = c->n_hits;
    if (!c->on_device) {
        bool ok = true;
        for (uint64_t i = 0; i < n; ++i) {
            uint32_t w1;
            ok &= make16(c->tax_row[i], c->pident, c->pident_milli, i, tax->hint_of_pos.data(), tax->n_tax, &w1);
            out[4 * i] = c->tax_row[i]; out[4 * i + 1] = w1; out[4 * i + 2] = (uint32_t)c->align_len[i]; out[4 * i + 3] = c->acc_rank[i];
        }
        if (!ok) { set_error("blu_hits_pack: a perc_identity is not an exact milli-percent value below 131.071 (use the column layouts or blu_hits_pack64)"); return BLU_ERR_INVALID_ARG; }
        return BLU_OK;
    }
    if ((uintptr_t)out & 15u) { set_error("packed records must be 16-byte aligned"); return BLU_ERR_INVALID_ARG; }
    if (hipSetDevice(tax->device) != hipSuccess) { set_error("hipSetDevice(%d) failed", tax->device); return BLU_ERR_NO_DEVICE; }
    hipStream_t s = (hipStream_t)stream;
    // the "a value does not fit" word lives with the handle (allocated by the first call): no hipMalloc / hipFree per call
    if (!tax->ws_pack_flag && hipMalloc((void**)&tax->ws_pack_flag, 64) != hipSuccess) { tax->ws_pack_flag = nullptr; set_error("hipMalloc failed"); return BLU_ERR_ALLOC; }
    uint32_t* const d_bad = tax->ws_pack_flag;
    uint32_t bad = 0;
    hipError_t e = hipMemsetAsync(d_bad, 0, 4, s);
    if (e == hipSuccess) {
        const uint64_t tiles = (n + 256 * PACK_ROWS - 1) / (256 * PACK_ROWS), cap = (uint64_t)(tax->num_cus > 0 ? tax->num_cus : 256) * 16;
        hipLaunchKernelGGL(pack16_kernel, dim3((unsigned)(tiles < cap ? tiles : cap)), dim3(256), 0, s, c->tax_row, c->pident, c->pident_milli, c->align_len,
                           c->acc_rank, n, tax->d_hint_of_pos, tax->n_tax, reinterpret_cast<uint4*>(out), d_bad);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(&bad, d_bad, 4, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { set_error("blu_hits_pack: %s", hipGetErrorString(e)); return BLU_ERR_HIP; }
    if (bad) { set_error("blu_hits_pack: a perc_identity is not an exact milli-percent value below 131.071 (use the column layouts or blu_hits_pack64)"); return BLU_ERR_INVALID_ARG; }
    return BLU_OK;
}

int blu_hits_pack64(const blu_taxonomy* tax, const blu_hits* c, uint32_t* out, void* stream) {
    int rc = check_args(tax, c, out);
    if (rc != BLU_OK || c->n_hits == 0) return rc;
    const uint64_t n = c->n_hits;
    if (!c->on_device) {
        for (uint64_t i = 0; i < n; ++i) {
            const double p = c->pident ? c->pident[i] : (double)c->pident_milli[i] / 1000.0;
            uint64_t bits;
            memcpy(&bits, &p, 8);
            uint32_t* r = out + 6 * i;
            r[0] = c->tax_row[i]; r[1] = hint_of(c->tax_row[i], tax->hint_of_pos.data(), tax->n_tax) << BLU_KTHR_BITS;
            r[2] = (uint32_t)c->align_len[i]; r[3] = c->acc_rank[i]; r[4] = (uint32_t)bits; r[5] = (uint32_t)(bits >> 32);
        }
        return BLU_OK;
    }
    if ((uintptr_t)out & 7u) { set_error("packed64 records must be 8-byte aligned"); return BLU_ERR_INVALID_ARG; }
    if (hipSetDevice(tax->device) != hipSuccess) { set_error("hipSetDevice(%d) failed", tax->device); return BLU_ERR_NO_DEVICE; }
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(pack24_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, c->tax_row, c->pident, c->pident_milli, c->align_len,
                       c->acc_rank, n, tax->d_hint_of_pos, tax->n_tax, reinterpret_cast<uint2*>(out));
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) { set_error("blu_hits_pack64: %s", hipGetErrorString(e)); return BLU_ERR_HIP; }
    return BLU_OK;
}

}  // extern "C"
