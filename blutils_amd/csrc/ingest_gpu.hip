// GPU ingest of BLAST outfmt-6 text ("GPU-side parser"): the whole file goes to HBM once and comes back as
// the grouped SoA columns the engine reads.  It produces exactly what the CPU ingest in pipeline.cpp produces
// (reference: build_consensus_identities/mod.rs:134-244,329-373 — 13 tab-separated columns, rows regrouped by query
// in first-appearance order with file order kept inside a query, bit_score truncated toward zero, left join on the
// taxid), for files in the plain form BLAST writes; any other file (quotes, empty lines, odd numbers) is handed back
// to the CPU path untouched (BLU_INGEST_FALLBACK).
//
// Stages (all on the handle's device; the kernels on the default stream):
//   0. upload          pread() into pinned staging slots -> HBM, three reader threads, all copies on the null stream;
//                      the file is never mapped into the process
//   1. line index      newline count per 4 KiB tile -> exclusive scan -> line starts
//   2. parse           one thread per line: tab scan over aligned 16-byte loads, decimal fast path for the four numeric
//                      columns (mantissa <= 15 digits and |exp10| <= 22: one IEEE operation, so the value is strtod's),
//                      taxid -> taxonomy row through the uploaded open-addressing map, two 64-bit hashes of the query
//                      and accession fields
//   3. dictionaries    open-addressing tables keyed by hash (insert = CAS on the hash + atomicMin of the row), finalised
//                      with length + first 12 key bytes; every row then verifies its own text against its slot (a
//                      mismatch = two strings with one 64-bit hash -> fallback), so ids are exact, not probabilistic
//   4. ids             queries numbered by first row (the first rows marked in a row-indexed array, a prefix sum over it);
//                      accessions ranked in byte order on the host (only the distinct strings travel; the sort runs on a
//                      thread of its own while the device builds the query dictionary) and the ranks uploaded
//   5. grouping        stable radix sort by query id unless the file is grouped already; gathers; segment offsets
//   6. hand-over       the grouped columns stay on the device for the engine (host copies only on request); the distinct
//                      query / accession strings come back packed and a background thread turns them into the host
//                      tables; after the engine, top_rows_kernel compacts the top-score rows of the rendered queries —
//                      all the writer reads of the table
// Every kernel of the ingest is written here — the device-wide prefix sums, block-level sums and scans, the parsing and dictionary
// kernels, and the stable radix sort that regroups the rows of a file whose queries are not contiguous; no device library is
// called.  HBM-bound byte work: no MFMA.
#include <hip/hip_runtime.h>
#include <cstring>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <string_view>
#include <thread>
#include <vector>

#include "blu_consensus.h"
#include "blu_internal.h"
#include "ingest.h"

namespace blu {
namespace {

enum : uint32_t {
    FB_EMPTY_LINE = 1, FB_COLUMNS = 2, FB_QUOTE = 4, FB_NUMBER = 8, FB_RANGE = 16, FB_HASH_COLLISION = 32, FB_TABLE_FULL = 64,
};

struct Slot {            // 32 bytes
    unsigned long long hash;   // 0 = empty
    uint32_t first_row;        // smallest row index holding this key
    uint32_t id;               // query id / accession rank
    uint32_t len;
    unsigned char head[12];
};

struct DevTaxidMap { const TaxidMap::E* tab; uint64_t mask; };

// out of device memory is not an error of the call: the CPU parser takes the file (oom_fallback is set by the callers
// that have one)
#define HIPCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
        if (e_ == hipErrorOutOfMemory && oom_fallback) { (void)hipGetLastError(); if (oom_why) *oom_why = "not enough free device memory"; rc = BLU_INGEST_FALLBACK; } \
        else { set_error("GPU ingest: %s failed: %s", #x, hipGetErrorString(e_)); rc = BLU_ERR_HIP; } \
        goto done; } } while (0)

// ---- block-level sum and exclusive scan: 64-wide wavefronts reduce / scan in registers (DPP), the wave totals meet in LDS ----
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_u32(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true); }
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v) {   // inclusive prefix sum over the 64 lanes
    v += dpp_u32<0x111>(v);                                        // row_shr:1, :2, :4, :8 (zero shifted in): prefix inside each 16-lane row
    v += dpp_u32<0x112>(v);
    v += dpp_u32<0x114>(v);
    v += dpp_u32<0x118>(v);
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const uint32_t t0 = (uint32_t)__builtin_amdgcn_readlane((int)v, 15), t1 = (uint32_t)__builtin_amdgcn_readlane((int)v, 31),
                   t2 = (uint32_t)__builtin_amdgcn_readlane((int)v, 47);
    const uint32_t r = lane >> 4;
    return v + (r == 0 ? 0u : (r == 1 ? t0 : (r == 2 ? t0 + t1 : t0 + t1 + t2)));
}
// sum over the block in every thread's return value; `wave_sums` = LDS, one word per wave
template <int THREADS>
__device__ __forceinline__ uint32_t block_sum(uint32_t v, uint32_t* wave_sums) {
    const uint32_t incl = wave_incl_scan(v);
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) wave_sums[wave] = incl;
    __syncthreads();
    uint32_t total = 0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; ++w) total += wave_sums[w];
    __syncthreads();
    return total;
}
// exclusive prefix of v over the block
template <int THREADS>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* wave_sums) {
    const uint32_t incl = wave_incl_scan(v);
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) wave_sums[wave] = incl;
    __syncthreads();
    uint32_t before = 0;
#pragma unroll
    for (int w = 0; w < THREADS / 64; ++w) before += (uint32_t)w < wave ? wave_sums[w] : 0u;
    __syncthreads();
    return before + incl - v;
}

// ---- device-wide exclusive prefix sum (u32 / u64), three launches: sums of 4096-element blocks, a one-block scan of those sums,
// the blocks again with their offsets.  The arrays scanned here are bookkeeping (newline counts per 4 KiB tile, string lengths,
// top-row counts: 2 M elements at most per 100 M rows), so reading them twice costs microseconds; written here rather than
// taken from a library so that the ingest's device code is the kernels in this file plus two radix sorts.
constexpr int SCAN_THREADS = 1024, SCAN_ITEMS = 4, SCAN_BLOCK = SCAN_THREADS * SCAN_ITEMS;

template <class T>
__device__ __forceinline__ T wave_incl_scan_t(T v) {
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) { const T u = __shfl_up(v, d); if (lane >= d) v += u; }
    return v;
}
// exclusive prefix of v over the 1024 threads of the block, and the block's total; `wave_tot` = LDS, 16 entries
template <class T>
__device__ __forceinline__ T block_excl_scan_t(T v, T* wave_tot, T* total) {
    const T incl = wave_incl_scan_t(v);
    const uint32_t wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 63) wave_tot[wave] = incl;
    __syncthreads();
    T before = 0, all = 0;
#pragma unroll
    for (uint32_t w = 0; w < SCAN_THREADS / 64; ++w) { const T x = wave_tot[w]; all += x; if (w < wave) before += x; }
    __syncthreads();
    *total = all;
    return before + incl - v;
}
template <class T>
__global__ __launch_bounds__(SCAN_THREADS) void scan_block_sums(const T* __restrict__ in, size_t n, T* __restrict__ block_sum) {
    __shared__ T wave_tot[SCAN_THREADS / 64];
    const size_t base = ((size_t)blockIdx.x * SCAN_THREADS + threadIdx.x) * SCAN_ITEMS;
    T v = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) if (base + k < n) v += in[base + k];
    T total;
    (void)block_excl_scan_t(v, wave_tot, &total);
    if (threadIdx.x == 0) block_sum[blockIdx.x] = total;
}
template <class T>
__global__ __launch_bounds__(SCAN_THREADS) void scan_block_offsets(const T* __restrict__ block_sum, size_t n_blocks, T* __restrict__ block_off) {
    __shared__ T wave_tot[SCAN_THREADS / 64];
    T carry = 0;
    for (size_t b0 = 0; b0 < n_blocks; b0 += SCAN_THREADS) {        // (one block: the sums of 4096-element blocks are few)
        const size_t b = b0 + threadIdx.x;
        const T v = b < n_blocks ? block_sum[b] : (T)0;
        T total;
        const T ex = block_excl_scan_t(v, wave_tot, &total);
        if (b < n_blocks) block_off[b] = carry + ex;
        carry += total;
    }
}
template <class T>
__global__ __launch_bounds__(SCAN_THREADS) void scan_apply(const T* __restrict__ in, size_t n, const T* __restrict__ block_off, T* __restrict__ out) {
    __shared__ T wave_tot[SCAN_THREADS / 64];
    const size_t base = ((size_t)blockIdx.x * SCAN_THREADS + threadIdx.x) * SCAN_ITEMS;
    T x[SCAN_ITEMS], v = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) { x[k] = base + k < n ? in[base + k] : (T)0; v += x[k]; }
    T total;
    T run = block_off[blockIdx.x] + block_excl_scan_t(v, wave_tot, &total);
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; ++k) { if (base + k < n) out[base + k] = run; run += x[k]; }
}
template <class T> size_t scan_tmp_bytes(size_t n) { return 2 * ((n + SCAN_BLOCK - 1) / SCAN_BLOCK) * sizeof(T) + 16; }
// out[i] = in[0] + ... + in[i - 1] for i < n (in and out distinct); tmp: scan_tmp_bytes<T>(n) bytes of device memory
template <class T>
hipError_t exclusive_scan_dev(const T* in, T* out, size_t n, void* tmp) {
    if (n == 0) return hipSuccess;
    const size_t n_blocks = (n + SCAN_BLOCK - 1) / SCAN_BLOCK;
    T* const block_sum = (T*)tmp;
    T* const block_off = block_sum + n_blocks;
    hipLaunchKernelGGL((scan_block_sums<T>), dim3((unsigned)n_blocks), dim3(SCAN_THREADS), 0, 0, in, n, block_sum);
    hipLaunchKernelGGL((scan_block_offsets<T>), dim3(1), dim3(SCAN_THREADS), 0, 0, (const T*)block_sum, n_blocks, block_off);
    hipLaunchKernelGGL((scan_apply<T>), dim3((unsigned)n_blocks), dim3(SCAN_THREADS), 0, 0, in, n, (const T*)block_off, out);
    return hipGetLastError();
}

// ---- stable radix sort of (key, value) pairs of 32-bit words, least significant digit first, 8 bits per pass: regroups the rows
// of a file whose queries are not contiguous (key = query id, value = row: file order inside a query survives because every
// pass is stable).  Per pass: a 256-bin histogram per 4096-element block, one prefix sum over the (digit, block) table, and a
// scatter in which an element's place = the table's entry for (its digit, its block) + the elements of that digit before it
// in the block — counted per round of 1024 by wave (ballot match masks: the rank inside the wave) and across waves in LDS.
constexpr int RS_THREADS = 1024, RS_ITEMS = 4, RS_BLOCK = RS_THREADS * RS_ITEMS;

__global__ __launch_bounds__(RS_THREADS) void radix_hist(const uint32_t* __restrict__ keys, uint32_t n, uint32_t shift, uint32_t n_blocks,
                                                         uint32_t* __restrict__ hist) {
    __shared__ uint32_t h[256];
    if (threadIdx.x < 256) h[threadIdx.x] = 0;
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * RS_BLOCK;
#pragma unroll
    for (int r = 0; r < RS_ITEMS; ++r) {
        const size_t i = base + (size_t)r * RS_THREADS + threadIdx.x;
        if (i < n) atomicAdd(&h[(keys[i] >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 256) hist[(size_t)threadIdx.x * n_blocks + blockIdx.x] = h[threadIdx.x];   // digit-major: the prefix sum runs over digits, then blocks
}

__global__ __launch_bounds__(RS_THREADS) void radix_scatter(const uint32_t* __restrict__ keys_in, const uint32_t* __restrict__ vals_in, uint32_t n,
                                                            uint32_t shift, uint32_t n_blocks, const uint32_t* __restrict__ offs,
                                                            uint32_t* __restrict__ keys_out, uint32_t* __restrict__ vals_out) {
    __shared__ uint32_t next[256];                           // where the block's next element of each digit goes
    __shared__ uint32_t wcount[RS_THREADS / 64][256];        // this round: elements per (wave, digit), then their first place
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    if (tid < 256) next[tid] = offs[(size_t)tid * n_blocks + blockIdx.x];
    const size_t base = (size_t)blockIdx.x * RS_BLOCK;
    for (int r = 0; r < RS_ITEMS; ++r) {
        for (uint32_t k = tid; k < (RS_THREADS / 64) * 256; k += RS_THREADS) (&wcount[0][0])[k] = 0;
        __syncthreads();
        const size_t i = base + (size_t)r * RS_THREADS + tid;
        const bool valid = i < n;
        const uint32_t key = valid ? keys_in[i] : 0u, d = (key >> shift) & 255u;
        unsigned long long same = __ballot(valid);           // lanes of this wave with a valid element of the same digit
#pragma unroll
        for (int b = 0; b < 8; ++b) { const unsigned long long bb = __ballot((d >> b) & 1u); same &= ((d >> b) & 1u) ? bb : ~bb; }
        const uint32_t rank_in_wave = (uint32_t)__popcll(same & ((1ull << lane) - 1ull));
        if (valid && rank_in_wave == 0) wcount[wave][d] = (uint32_t)__popcll(same);
        __syncthreads();
        if (tid < 256) {                                      // digit `tid`: the waves' first places, in wave order
            uint32_t run = next[tid];
#pragma unroll
            for (uint32_t w = 0; w < RS_THREADS / 64; ++w) { const uint32_t c = wcount[w][tid]; wcount[w][tid] = run; run += c; }
            next[tid] = run;
        }
        __syncthreads();
        if (valid) { const uint32_t at = wcount[wave][d] + rank_in_wave; keys_out[at] = key; vals_out[at] = vals_in[i]; }
        __syncthreads();
    }
}

// sorts n pairs by the low `bits` bits of the key; the sorted arrays are (*keys, *vals) on return (the two buffers of each swap roles
// per pass).  table: 2 x 256 x ceil(n / 4096) words of device memory; scan_tmp: scan_tmp_bytes<uint32_t>(256 x ceil(n / 4096))
static hipError_t radix_sort_pairs_dev(uint32_t** keys, uint32_t** keys_alt, uint32_t** vals, uint32_t** vals_alt, uint32_t n, int bits,
                                       uint32_t* table, void* scan_tmp) {
    if (n == 0) return hipSuccess;
    const uint32_t n_blocks = (uint32_t)(((size_t)n + RS_BLOCK - 1) / RS_BLOCK);
    const size_t cells = (size_t)256 * n_blocks;
    uint32_t* const hist = table;
    uint32_t* const offs = table + cells;
    for (int shift = 0; shift < bits; shift += 8) {
        hipLaunchKernelGGL(radix_hist, dim3(n_blocks), dim3(RS_THREADS), 0, 0, (const uint32_t*)*keys, n, (uint32_t)shift, n_blocks, hist);
        const hipError_t e = exclusive_scan_dev<uint32_t>(hist, offs, cells, scan_tmp);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(radix_scatter, dim3(n_blocks), dim3(RS_THREADS), 0, 0, (const uint32_t*)*keys, (const uint32_t*)*vals, n, (uint32_t)shift, n_blocks,
                           (const uint32_t*)offs, *keys_alt, *vals_alt);
        std::swap(*keys, *keys_alt);
        std::swap(*vals, *vals_alt);
    }
    return hipGetLastError();
}

// ---- 1. line index ---------------------------------------------------------------------------------------
constexpr int TILE_THREADS = 256;
constexpr uint64_t TILE_BYTES = TILE_THREADS * 16;

__device__ __forceinline__ uint32_t nl_mask16(const uint4 v, uint64_t base, uint64_t size) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t m = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const uint32_t c = (w[k >> 2] >> (8 * (k & 3))) & 0xFF;
        if (c == '\n' && base + k < size) m |= 1u << k;
    }
    return m;
}

__global__ __launch_bounds__(TILE_THREADS) void count_newlines(const uint4* __restrict__ text, uint64_t size, uint32_t* __restrict__ tile_count) {
    const uint64_t base = ((uint64_t)blockIdx.x * TILE_THREADS + threadIdx.x) * 16;
    uint32_t c = 0;
    if (base < size) c = __popc(nl_mask16(text[base / 16], base, size));
    __shared__ uint32_t wave_sums[TILE_THREADS / 64];
    const uint32_t total = block_sum<TILE_THREADS>(c, wave_sums);
    if (threadIdx.x == 0) tile_count[blockIdx.x] = total;
}

__global__ __launch_bounds__(TILE_THREADS) void write_line_starts(const uint4* __restrict__ text, uint64_t size, const uint32_t* __restrict__ tile_base,
                                                                  uint64_t* __restrict__ line_start) {
    const uint64_t base = ((uint64_t)blockIdx.x * TILE_THREADS + threadIdx.x) * 16;
    uint32_t m = 0;
    if (base < size) m = nl_mask16(text[base / 16], base, size);
    __shared__ uint32_t wave_sums[TILE_THREADS / 64];
    const uint32_t before = block_excl_scan<TILE_THREADS>((uint32_t)__popc(m), wave_sums);
    uint64_t k = (uint64_t)tile_base[blockIdx.x] + before;
    while (m) {
        const int b = __ffs(m) - 1;
        m &= m - 1;
        line_start[++k] = base + b + 1;     // line k+1 starts after newline k
    }
}

// ---- 2. parse ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t taxid_lookup(const DevTaxidMap& t, long long k) {
    unsigned long long x = (unsigned long long)k * 0x9E3779B97F4A7C15ull;
    x ^= x >> 32;
    uint64_t i = x & t.mask;
    for (;;) {
        const TaxidMap::E e = t.tab[i];
        if (!e.used) return BLU_UNMATCHED_TAXID;
        if (e.key == k) return e.val;
        i = (i + 1) & t.mask;
    }
}

__device__ const double P10[23] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

struct NumState {
    unsigned long long mant;
    int digits, frac, exp10, exp_digits;
    bool neg, exp_neg, exp_sign, seen_dot, seen_exp, bad, any;
    __device__ void reset() { mant = 0; digits = frac = exp10 = exp_digits = 0; neg = exp_neg = exp_sign = seen_dot = seen_exp = bad = any = false; }
    __device__ void feed(uint32_t c, bool first) {
        if (c - '0' < 10u) {
            if (seen_exp) { exp10 = exp10 * 10 + (int)(c - '0'); if (++exp_digits > 3) bad = true; }
            else { mant = mant * 10 + (c - '0'); if (mant != 0 || seen_dot) ++digits; if (seen_dot) ++frac; any = true; }
        } else if (c == '-' && first) neg = true;
        else if (c == '.' && !seen_dot && !seen_exp) seen_dot = true;
        else if ((c == 'e' || c == 'E') && any && !seen_exp) seen_exp = true;
        else if ((c == '-' || c == '+') && seen_exp && exp_digits == 0 && !exp_sign) { exp_sign = true; exp_neg = c == '-'; }
        else bad = true;
    }
    // the value strtod gives, or bad: mantissa and 10^k exact, one correctly rounded operation (Clinger's fast path)
    __device__ bool value(double* out) const {
        if (bad || !any || digits > 15 || frac > 300 || (seen_exp && exp_digits == 0)) return false;
        const int e = (exp_neg ? -exp10 : exp10) - frac;
        if (e < -22 || e > 22) return false;
        const double m = (double)mant;
        const double x = e < 0 ? m / P10[-e] : m * P10[e];
        *out = neg ? -x : x;
        return true;
    }
};

__device__ __forceinline__ unsigned long long hash_step(unsigned long long h, uint32_t c) { return (h ^ c) * 1099511628211ull; }
__device__ __forceinline__ unsigned long long hash_finish(unsigned long long h, uint32_t len) {
    h ^= (unsigned long long)len * 0x9E3779B97F4A7C15ull;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    return h ? h : 1ull;   // 0 marks an empty slot
}

struct RowOut {
    unsigned long long* qh; unsigned long long* ah;     // hashes of the query / accession fields
    unsigned long long* qpos; unsigned long long* apos;  // field offset | length << 44
    uint32_t* tax; double* pid; int32_t* aln; int32_t* bs;
};

// One line, read from global memory a byte at a time through one state machine for all thirteen columns: the form that
// takes any line length.  parse_rows uses it for the blocks whose 256 lines do not fit its LDS stage.
__device__ __noinline__ void parse_row_general(const unsigned char* __restrict__ text, const uint64_t* __restrict__ line_start, uint32_t i,
                                               const DevTaxidMap& taxmap, const RowOut& o, uint32_t* __restrict__ flags,
                                               unsigned long long* __restrict__ n_unmatched) {
    const uint64_t p = line_start[i];
    uint64_t e = line_start[i + 1] - 1;                 // the newline (or one past the end of a last line without one)
    if (e > p && text[e - 1] == '\r') --e;
    if (e <= p) { atomicOr(flags, FB_EMPTY_LINE); return; }
    int col = 0;
    uint64_t fstart = p;                                // start of the current field
    unsigned long long h = 1469598103934665603ull;
    NumState num;
    num.reset();
    double v_tax = 0, v_pid = 0, v_aln = 0, v_bs = 0;
    uint32_t fb = 0;
    auto end_field = [&](uint64_t fend) {
        const uint64_t len = fend - fstart;
        if (col == 0) { o.qh[i] = hash_finish(h, (uint32_t)len); o.qpos[i] = fstart | (len << 44); if (len >= (1u << 20)) fb |= FB_COLUMNS; }
        else if (col == 1) { o.ah[i] = hash_finish(h, (uint32_t)len); o.apos[i] = fstart | (len << 44); if (len >= (1u << 20)) fb |= FB_COLUMNS; }
        // (subject_taxid and align_length are Int64 columns, mod.rs:226-244: a fraction or an exponent is left to the CPU
        // parser, which refuses the file as the reference would)
        else if (col == 2) { if (!num.value(&v_tax) || num.seen_dot || num.seen_exp) fb |= FB_NUMBER; }
        else if (col == 3) { if (!num.value(&v_pid)) fb |= FB_NUMBER; }
        else if (col == 4) { if (!num.value(&v_aln) || num.seen_dot || num.seen_exp) fb |= FB_NUMBER; }
        else if (col == 12) { if (!num.value(&v_bs)) fb |= FB_NUMBER; }
        ++col;
        fstart = fend + 1;
        h = 1469598103934665603ull;
        num.reset();
    };
    // aligned 16-byte loads; bytes outside [p, e) are skipped
    for (uint64_t a = p & ~15ull; a < e && col < 13; a += 16) {
        const uint4 v = *reinterpret_cast<const uint4*>(text + a);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const uint64_t q = a + k;
            if (q < p || q >= e || col >= 13) continue;
            const uint32_t c = (w[k >> 2] >> (8 * (k & 3))) & 0xFF;
            if (c == '\t') { end_field(q); continue; }
            if (col <= 1) { h = hash_step(h, c); if (c == '"') fb |= FB_QUOTE; }
            else if (col <= 4 || col == 12) num.feed(c, q == fstart);
        }
    }
    if (col < 13) end_field(e);                         // the last field ends with the line
    if (col < 13) fb |= FB_COLUMNS;
    if (fb) { atomicOr(flags, fb); return; }
    // mod.rs:184 AnyValue::Float64 -> try_extract::<i64> (truncation); the engine columns are 32-bit
    const double bs_t = trunc(v_bs);
    if (!(bs_t >= -2147483648.0 && bs_t <= 2147483647.0) || !(v_aln >= -2147483648.0 && v_aln <= 2147483647.0) ||
        !(v_tax >= -9.2e18 && v_tax <= 9.2e18)) { atomicOr(flags, FB_RANGE); return; }
    const uint32_t row = taxid_lookup(taxmap, (long long)v_tax);   // left join (mod.rs:72-76)
    if (row == BLU_UNMATCHED_TAXID) atomicAdd(n_unmatched, 1ull);
    o.tax[i] = row; o.pid[i] = v_pid; o.aln[i] = (int32_t)v_aln; o.bs[i] = (int32_t)bs_t;
}

// The parse kernel proper: a block takes 256 consecutive lines, whose text is one contiguous span of the file (17 KB for
// BLAST's usual 67-byte lines).  The span is staged in LDS with coalesced 16-byte loads; every lane then (1) finds the
// tabs of its line with word-wide compares and leaves their positions in LDS, (2) walks the six columns the engine
// reads ONE COLUMN AT A TIME, so that the 64 lanes of a wave are in the same column and the same branch of the same
// small loop: the one-state-machine-per-line form above spends its time in divergence (a wave has lanes in every
// column at once and runs the end-of-field code of all of them at nearly every byte) and in instruction fetch (the
// 16-byte unrolled body does not fit the instruction cache).  Same grammar, same hashes, same flags as the general form;
// a block whose span exceeds the stage (lines of 128 bytes and more on average) is parsed by the general form.
constexpr uint32_t STAGE_BYTES = 32768;
constexpr int PARSE_THREADS = 256;

__global__ __launch_bounds__(PARSE_THREADS) void parse_rows(const unsigned char* __restrict__ text, const uint64_t* __restrict__ line_start, uint32_t n_rows,
                                                            DevTaxidMap taxmap, RowOut o, uint32_t* __restrict__ flags,
                                                            unsigned long long* __restrict__ n_unmatched) {
    __shared__ uint4 stage16[STAGE_BYTES / 16];
    __shared__ uint16_t tab_at[13 * PARSE_THREADS];                  // [column][thread]: position of the tab that ends the column
    const uint32_t r0 = blockIdx.x * PARSE_THREADS, r1 = min(r0 + (uint32_t)PARSE_THREADS, n_rows), i = r0 + threadIdx.x;
    // (the four line starts travel together: two round trips to memory per block — these, then the text — and no more)
    const uint64_t ls0 = line_start[r0], s1 = line_start[r1];       // (line_start[n_rows] = one past the last newline, or size + 1)
    const uint64_t my_p = line_start[min(i, r1 - 1)], my_e = line_start[min(i, r1 - 1) + 1];
    const uint64_t s0 = ls0 & ~15ull;
    if (s1 - s0 > STAGE_BYTES) {                                     // uniform over the block
        if (i < r1) parse_row_general(text, line_start, i, taxmap, o, flags, n_unmatched);
        return;
    }
    {
        const uint32_t n16 = (uint32_t)((s1 - s0 + 15) >> 4);        // (reads at most 15 bytes past s1 <= size + 1: inside the 64 bytes of padding)
        const uint4* src = reinterpret_cast<const uint4*>(text + s0);
        constexpr int PER_THREAD = STAGE_BYTES / 16 / PARSE_THREADS; // all of a thread's loads are in flight before the first is stored
        uint4 v[PER_THREAD];
#pragma unroll
        for (int j = 0; j < PER_THREAD; ++j) { const uint32_t k = threadIdx.x + (uint32_t)j * PARSE_THREADS; v[j] = src[min(k, n16 - 1)]; }
#pragma unroll
        for (int j = 0; j < PER_THREAD; ++j) { const uint32_t k = threadIdx.x + (uint32_t)j * PARSE_THREADS; if (k < n16) stage16[k] = v[j]; }
    }
    __syncthreads();
    if (i >= r1) return;
    const unsigned char* sb = reinterpret_cast<const unsigned char*>(stage16);
    const uint32_t* sw = reinterpret_cast<const uint32_t*>(stage16);
    const uint32_t p = (uint32_t)(my_p - s0);
    uint32_t e = (uint32_t)(my_e - 1 - s0);                          // the newline (or one past the end of a last line without one)
    if (e > p && sb[e - 1] == '\r') --e;
    if (e <= p) { atomicOr(flags, FB_EMPTY_LINE); return; }
    // ---- (1) the tabs: four bytes per compare (exact zero-byte test on word ^ 0x09090909), bytes outside [p, e) masked off
    uint32_t n_tabs = 0;
    for (uint32_t a = p & ~3u; a < e && n_tabs < 13; a += 4) {
        const uint32_t x = sw[a >> 2] ^ 0x09090909u;
        uint32_t m = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);   // bit 7 of every byte that is a tab
        if (a < p) m &= 0xFFFFFFFFu << (8 * (p - a));
        if (e - a < 4) m &= (1u << (8 * (e - a))) - 1u;
        while (m && n_tabs < 13) {
            tab_at[n_tabs * PARSE_THREADS + threadIdx.x] = (uint16_t)(a + ((uint32_t)__builtin_ctz(m) >> 3));
            ++n_tabs;
            m &= m - 1;
        }
    }
    if (n_tabs < 12) { atomicOr(flags, FB_COLUMNS); return; }       // fewer than 13 columns
    auto tab = [&](int c) { return (uint32_t)tab_at[c * PARSE_THREADS + threadIdx.x]; };
    uint32_t fb = 0;
    // ---- (2) query and accession: FNV-1a over the field, as in the general form
    auto hash_field = [&](uint32_t s, uint32_t t) {
        unsigned long long h = 1469598103934665603ull;
        for (uint32_t q = s; q < t; ++q) { const uint32_t c = sb[q]; h = hash_step(h, c); if (c == '"') fb |= FB_QUOTE; }
        return hash_finish(h, t - s);
    };
    const uint32_t t0 = tab(0), t1 = tab(1);
    const unsigned long long qh = hash_field(p, t0), ah = hash_field(t0 + 1, t1);
    // ---- the four numbers
    auto number = [&](uint32_t s, uint32_t t, bool integer, double* v) {
        NumState num;
        num.reset();
        for (uint32_t q = s; q < t; ++q) num.feed(sb[q], q == s);
        if (!num.value(v) || (integer && (num.seen_dot || num.seen_exp))) fb |= FB_NUMBER;
    };
    double v_tax = 0, v_pid = 0, v_aln = 0, v_bs = 0;
    const uint32_t t2 = tab(2), t3 = tab(3);
    number(t1 + 1, t2, true, &v_tax);                               // (subject_taxid and align_length are Int64 columns, mod.rs:226-244)
    number(t2 + 1, t3, false, &v_pid);
    number(t3 + 1, tab(4), true, &v_aln);
    number(tab(11) + 1, n_tabs == 13 ? tab(12) : e, false, &v_bs);
    if (fb) { atomicOr(flags, fb); return; }
    const double bs_t = trunc(v_bs);                                // mod.rs:184 AnyValue::Float64 -> try_extract::<i64> (truncation)
    if (!(bs_t >= -2147483648.0 && bs_t <= 2147483647.0) || !(v_aln >= -2147483648.0 && v_aln <= 2147483647.0) ||
        !(v_tax >= -9.2e18 && v_tax <= 9.2e18)) { atomicOr(flags, FB_RANGE); return; }
    const uint32_t row = taxid_lookup(taxmap, (long long)v_tax);   // left join (mod.rs:72-76)
    if (row == BLU_UNMATCHED_TAXID) atomicAdd(n_unmatched, 1ull);
    o.qh[i] = qh; o.qpos[i] = (s0 + p) | ((unsigned long long)(t0 - p) << 44);
    o.ah[i] = ah; o.apos[i] = (s0 + t0 + 1) | ((unsigned long long)(t1 - t0 - 1) << 44);
    o.tax[i] = row; o.pid[i] = v_pid; o.aln[i] = (int32_t)v_aln; o.bs[i] = (int32_t)bs_t;
}

// ---- 3. dictionaries ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void count_run_heads(const unsigned long long* __restrict__ h, uint32_t n, unsigned long long* __restrict__ count) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t head = i < n && (i == 0 || h[i] != h[i - 1]);
    __shared__ uint32_t wave_sums[1024 / 64];
    const uint32_t total = block_sum<1024>(head, wave_sums);
    if (threadIdx.x == 0 && total) atomicAdd(count, (unsigned long long)total);   // one atomic per 1024 rows
}

__global__ void dict_init(Slot* __restrict__ tab, uint64_t n_slots) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s < n_slots) { Slot e; memset(&e, 0, sizeof e); e.first_row = 0xFFFFFFFFu; tab[s] = e; }
}

// insert = CAS on the hash, atomicMin on the row; only_heads: rows whose hash equals the previous row's are skipped
__global__ void dict_insert(const unsigned long long* __restrict__ h, uint32_t n, Slot* __restrict__ tab, uint64_t mask, bool only_heads,
                            uint32_t* __restrict__ n_keys, uint32_t* __restrict__ flags) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    bool active = i < n;
    unsigned long long key = 0;
    if (active) {
        key = h[i];
        if (only_heads && i > 0 && h[i - 1] == key) active = false;
    }
    bool claimed = false, full = false;
    if (active) {
        uint64_t s = key & mask;
        full = true;
        for (uint32_t probes = 0; probes < 8192; ++probes) {
            unsigned long long cur = __hip_atomic_load(&tab[s].hash, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (cur == 0) {
                cur = atomicCAS(&tab[s].hash, 0ull, key);
                if (cur == 0) { claimed = true; cur = key; }
            }
            if (cur == key) {
                // (first_row only ever decreases: a value read here is an upper bound of the current one, so a row that is not
                // below it has nothing to store — 100 M rows, 300 k accessions: all but a few of the atomics go)
                if (__hip_atomic_load(&tab[s].first_row, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > i) atomicMin(&tab[s].first_row, i);
                full = false;
                break;
            }
            s = (s + 1) & mask;
        }
    }
    (void)claimed; (void)n_keys;   // (the keys are counted by dict_count afterwards: one atomic per new key on one word — 2 M
                                   // of them for a query dictionary — took 18 of this kernel's 19.6 ms)
    if (full) atomicOr(flags, FB_TABLE_FULL);
}

// number of occupied slots -> *n_keys (zeroed by the caller): a block reduction, one atomic per block
__global__ __launch_bounds__(1024) void dict_count(const Slot* __restrict__ tab, uint64_t n_slots, uint32_t* __restrict__ n_keys) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t used = s < n_slots && tab[s].hash != 0ull;
    __shared__ uint32_t wave_sums[1024 / 64];
    const uint32_t c = block_sum<1024>(used, wave_sums);
    if (threadIdx.x == 0 && c) atomicAdd(n_keys, c);
}

// occupied slots: key length and first bytes from the text of their first row; list of (first_row, slot)
__global__ void dict_finalize(Slot* __restrict__ tab, uint64_t n_slots, const unsigned char* __restrict__ text,
                              const unsigned long long* __restrict__ pos, uint32_t* __restrict__ list_row, uint32_t* __restrict__ list_slot,
                              uint32_t* __restrict__ cursor) {
    const uint64_t s = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n_slots || tab[s].hash == 0) return;
    const unsigned long long fp = pos[tab[s].first_row];
    const uint64_t off = fp & ((1ull << 44) - 1);
    const uint32_t len = (uint32_t)(fp >> 44);
    tab[s].len = len;
    for (uint32_t k = 0; k < 12; ++k) tab[s].head[k] = k < len ? text[off + k] : 0;
    const uint32_t at = atomicAdd(cursor, 1u);
    list_row[at] = tab[s].first_row;
    list_slot[at] = (uint32_t)s;
}

// queries are numbered in the order of their first rows (mod.rs:192-208) without sorting anything: the first rows are marked in a
// row-indexed array, an exclusive prefix sum over it gives every marked row the number of marked rows before it — the id
__global__ void mark_first_rows(const uint32_t* __restrict__ list_row, uint32_t n, uint32_t* __restrict__ mark) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k < n) mark[list_row[k]] = 1u;
}
__global__ void number_queries(Slot* __restrict__ tab, const uint32_t* __restrict__ list_row, const uint32_t* __restrict__ list_slot, uint32_t n,
                               const uint32_t* __restrict__ rank_of_row, const unsigned long long* __restrict__ pos,
                               unsigned long long* __restrict__ pos_by_id) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const uint32_t row = list_row[k], id = rank_of_row[row];
    tab[list_slot[k]].id = id;
    pos_by_id[id] = pos[row];          // the query's name: the text of its first row
}

__global__ void dict_assign_ids(Slot* __restrict__ tab, const uint32_t* __restrict__ slot_of_rank, const uint32_t* __restrict__ id_of_rank, uint32_t n) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) tab[slot_of_rank[r]].id = id_of_rank ? id_of_rank[r] : r;
}

// every row: find its slot, compare its own text with the slot's key (exact ids), write the id
__global__ void dict_lookup(const unsigned long long* __restrict__ h, const unsigned long long* __restrict__ pos, uint32_t n,
                            const Slot* __restrict__ tab, uint64_t mask, const unsigned char* __restrict__ text,
                            const unsigned long long* __restrict__ pos_all, uint32_t* __restrict__ id_out, uint32_t* __restrict__ flags) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long key = h[i];
    uint64_t s = key & mask;
    for (;;) {                                           // present: it was inserted (or merged into a run) above
        const unsigned long long cur = tab[s].hash;
        if (cur == key) break;
        if (cur == 0) { atomicOr(flags, FB_TABLE_FULL); id_out[i] = 0; return; }
        s = (s + 1) & mask;
    }
    const Slot sl = tab[s];
    const uint64_t off = pos[i] & ((1ull << 44) - 1);
    const uint32_t len = (uint32_t)(pos[i] >> 44);
    bool same = len == sl.len;
    {   // the first min(len, 12) bytes against the slot's: ONE unaligned 12-byte load (the text has 64 bytes of padding behind
        // it; the slot's bytes beyond its length are zero) instead of twelve byte loads with 64 lines each
        uint32_t w[3], hw[3];
        __builtin_memcpy(w, text + off, 12);
        __builtin_memcpy(hw, sl.head, 12);
        const uint32_t n = len < 12u ? len : 12u;
#pragma unroll
        for (uint32_t j = 0; j < 3; ++j) {
            const uint32_t nb = n > 4 * j ? n - 4 * j : 0u;
            const uint32_t m = nb >= 4 ? 0xFFFFFFFFu : (1u << (8 * nb)) - 1u;
            same = same && ((w[j] ^ hw[j]) & m) == 0;
        }
    }
    if (same && len > 12 && sl.first_row != i) {
        const uint64_t roff = pos_all[sl.first_row] & ((1ull << 44) - 1);
        for (uint32_t k = 12; same && k < len; ++k) same = text[off + k] == text[roff + k];
    }
    if (!same) atomicOr(flags, FB_HASH_COLLISION);
    id_out[i] = sl.id;
}

__global__ void gather_pos(const unsigned long long* __restrict__ pos, const uint32_t* __restrict__ rows, uint32_t n, unsigned long long* __restrict__ out) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) out[r] = pos[rows[r]];
}

// first 16 bytes of each listed field as two big-endian words: integer compares order them like memcmp
__global__ void gather_key16(const unsigned long long* __restrict__ pos, uint32_t n, const unsigned char* __restrict__ text,
                             unsigned long long* __restrict__ k0, unsigned long long* __restrict__ k1) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const uint64_t off = pos[r] & ((1ull << 44) - 1);
    const uint32_t len = (uint32_t)(pos[r] >> 44);
    unsigned long long a = 0, b = 0;
    for (uint32_t k = 0; k < 8; ++k) a = (a << 8) | (k < len ? text[off + k] : 0);
    for (uint32_t k = 8; k < 16; ++k) b = (b << 8) | (k < len ? text[off + k] : 0);
    k0[r] = a; k1[r] = b;
}

// ---- 5. grouping ---------------------------------------------------------------------------------------------------
// the distinct strings themselves, packed back to back: lengths -> (exclusive scan) -> bytes
__global__ void pos_lengths(const unsigned long long* __restrict__ pos, uint32_t n, unsigned long long* __restrict__ len) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= n) len[i] = i < n ? pos[i] >> 44 : 0ull;
}
__global__ void gather_bytes(const unsigned long long* __restrict__ pos, uint32_t n, const unsigned char* __restrict__ text,
                             const unsigned long long* __restrict__ off, unsigned char* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t src = pos[i] & ((1ull << 44) - 1);
    const uint32_t len = (uint32_t)(pos[i] >> 44);
    unsigned char* d = out + off[i];
    for (uint32_t k = 0; k < len; ++k) d[k] = text[src + k];
}
__global__ void check_grouped(const uint32_t* __restrict__ qid, uint32_t n, uint32_t* __restrict__ unsorted) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > 0 && i < n && qid[i] < qid[i - 1]) *unsorted = 1u;
}
__global__ void iota_u32(uint32_t* __restrict__ v, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = i;
}
// segment offsets from the query ids in grouped (non-decreasing) order: the first row of every id, and the row count
// behind the last one.  Every id 0 .. n_queries - 1 occurs (ids are handed out to keys that are in the table).  (A
// histogram with one atomic per row and a scan did the same in 5.7 ms per 100 M rows; this reads the ids once.)
__global__ void segment_starts(const uint32_t* __restrict__ qid_sorted, uint32_t n, uint32_t n_queries, unsigned long long* __restrict__ seg_off) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t q = qid_sorted[i];
    if (i == 0 || qid_sorted[i - 1] != q) seg_off[q] = i;
    if (i == n - 1) seg_off[n_queries] = n;
}
struct Cols { const int32_t* bs; const int32_t* aln; const uint32_t* tax; const uint32_t* arank; const double* pid; };
struct ColsOut { int32_t* bs; int32_t* aln; uint32_t* tax; uint32_t* arank; double* pid; };
__global__ void gather_cols(Cols in, ColsOut out, const uint32_t* __restrict__ perm, uint32_t n) {
    const uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const uint32_t i = perm ? perm[j] : j;
    out.bs[j] = in.bs[i]; out.aln[j] = in.aln[i]; out.tax[j] = in.tax[i]; out.arank[j] = in.arank[i]; out.pid[j] = in.pid[i];
}

const char* fallback_text(uint32_t f) {
    if (f & FB_EMPTY_LINE) return "an empty line";
    if (f & FB_COLUMNS) return "a line with fewer than 13 columns (or an oversized field)";
    if (f & FB_QUOTE) return "a quoted query / accession field";
    if (f & FB_NUMBER) return "a numeric field outside the decimal fast path";
    if (f & FB_RANGE) return "a number outside the engine's 32-bit columns";
    if (f & FB_HASH_COLLISION) return "two strings with one 64-bit hash";
    if (f & FB_TABLE_FULL) return "a dictionary table that filled up";
    return "unknown";
}

uint64_t pow2_at_least(uint64_t x) { uint64_t p = 1024; while (p < x) p <<= 1; return p; }

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

// Device allocations of one ingest: everything still registered is freed when the arena goes out of scope, whichever
// way the function is left.
struct DeviceArena {
    std::vector<void*> ptrs;
    // keep: work buffers that are done with are handed back only when the arena goes (on some boxes an allocation that
    // follows a hipFree of GBs takes 0.1 - 0.4 s per GB: 0.3 s of a 0.7 s ingest); set when the device has room for it
    bool keep = false;
    hipError_t alloc(void** p, size_t bytes) {
        const hipError_t e = hipMalloc(p, bytes);
        if (e == hipSuccess) ptrs.push_back(*p);
        return e;
    }
    void free(void* p) {
        if (keep) return;
        auto it = std::find(ptrs.begin(), ptrs.end(), p);
        if (it == ptrs.end()) return;
        (void)hipFree(p);
        ptrs.erase(it);
    }
    void release(void* p) {   // ownership moves elsewhere
        auto it = std::find(ptrs.begin(), ptrs.end(), p);
        if (it != ptrs.end()) ptrs.erase(it);
    }
    void free_all() { for (void* p : ptrs) (void)hipFree(p); ptrs.clear(); }
    ~DeviceArena() { free_all(); }
};

// The text goes page cache -> pinned staging -> HBM without ever being mapped into the process: three reader threads,
// each with two pinned slots, pread() a piece while the previous one is on the wire.  What the parts cost on a box
// (scripts/probe/upload_probe.hip, 6.7 GB): the copies alone 0.12 s (56 GB/s, the PCIe rate; one stream carries 50 of it),
// pread alone 30 GB/s per thread, so two or three readers keep the wire busy — and EVERY hipStreamCreate 15 ms (the first
// one of the process 40-170 ms: an HSA queue each).  Six readers with a stream and a hipHostMalloc each, as this was
// written first, spent more time setting up than copying.  Hence: one pinned block for all slots, and all copies on the
// null stream, which the kernels that follow need anyway (in order on one queue: a slot's event still says when it is free).
// Mapping the file and handing it to hipMemcpy moves the bytes as fast (the runtime pins the page-cache pages) but
// costs 0.06 s to populate and 0.07-0.10 s to unmap 6.7 GB of page table.
static int upload_file(int fd, size_t size, unsigned char* d_text, int device, std::string* err) {
    unsigned nt = 3;
    if (const char* env = getenv("BLU_UPLOAD_THREADS")) nt = (unsigned)atoi(env);
    const size_t piece = 8u << 20, n_pieces = (size + piece - 1) / piece;
    nt = std::max(1u, std::min<unsigned>(std::min(nt, 16u), (unsigned)std::max<size_t>(n_pieces, 1)));
    std::atomic<size_t> next{0};
    std::atomic<int> failed{0};      // 1: the file could not be read; 2: the staging path could not be set up or used (HIP)
    std::vector<std::string> errs(nt + 1);
    auto fail = [&](unsigned t, const char* what, hipError_t e) {
        errs[t] = std::string(what) + ": " + (e == hipSuccess ? strerror(errno) : hipGetErrorString(e));
        int none = 0;
        failed.compare_exchange_strong(none, e == hipSuccess ? 1 : 2);
    };
    char* block = nullptr;
    std::vector<hipEvent_t> events((size_t)nt * 2, nullptr);
    const bool trace = getenv("BLU_INGEST_TRACE") != nullptr;
    const double t_in = now_s();
    hipError_t e0 = hipHostMalloc((void**)&block, (size_t)nt * 2 * piece, hipHostMallocDefault);
    for (size_t k = 0; k < events.size() && e0 == hipSuccess; ++k) e0 = hipEventCreateWithFlags(&events[k], hipEventDisableTiming);
    if (e0 != hipSuccess) fail(nt, "staging set-up", e0);
    const double t_setup = now_s();
    auto work = [&](unsigned t) {
        char* const slots = block + (size_t)t * 2 * piece;
        hipEvent_t* const ev = &events[(size_t)t * 2];
        bool busy[2] = {false, false};
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) fail(t, "hipSetDevice", e);
        unsigned turn = 0;
        while (!failed.load(std::memory_order_relaxed)) {
            const size_t k = next.fetch_add(1);
            if (k >= n_pieces) break;
            const unsigned sl = turn++ & 1;
            if (busy[sl] && (e = hipEventSynchronize(ev[sl])) != hipSuccess) { fail(t, "hipEventSynchronize", e); break; }
            const size_t off = k * piece, len = std::min(piece, size - off);
            size_t got = 0;
            while (got < len) {
                const ssize_t r = pread(fd, slots + sl * piece + got, len - got, (off_t)(off + got));
                if (r < 0 && errno == EINTR) continue;
                if (r <= 0) { fail(t, "pread", hipSuccess); break; }
                got += (size_t)r;
            }
            if (got < len) break;
            if ((e = hipMemcpyAsync(d_text + off, slots + sl * piece, len, hipMemcpyHostToDevice, nullptr)) != hipSuccess) { fail(t, "hipMemcpyAsync", e); break; }
            if ((e = hipEventRecord(ev[sl], nullptr)) != hipSuccess) { fail(t, "hipEventRecord", e); break; }
            busy[sl] = true;
        }
    };
    if (!failed) {
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < nt; ++t) pool.emplace_back(work, t);
        work(0);
        for (auto& th : pool) th.join();
    }
    {   // the slots are free once everything queued has been sent (also on the way out of a failure: copies may be in flight)
        const double t_queued = now_s();
        const hipError_t e = hipStreamSynchronize(nullptr);
        if (e != hipSuccess && !failed) fail(nt, "hipStreamSynchronize", e);
        if (trace) fprintf(stderr, "[ingest-gpu]   upload: pinned block + events %.3f s, pieces read and queued %.3f s, last ones on the wire %.3f s\n",
                           t_setup - t_in, t_queued - t_setup, now_s() - t_queued);
    }
    for (hipEvent_t ev : events) if (ev) (void)hipEventDestroy(ev);
    if (block) (void)hipHostFree(block);
    if (failed) {
        for (auto& m : errs) if (!m.empty()) { *err = m; break; }
        (void)hipGetLastError();
        return failed == 1 ? BLU_ERR_IO : BLU_INGEST_FALLBACK;   // (no pinned memory to be had, say: the CPU parser takes the file)
    }
    return BLU_OK;
}

// (defined in consensus_kernel.hip; named here only so that the warm-up can ask for its attributes, which makes the runtime load
// the code object of that translation unit — the two dozen builds of the consensus kernels — ahead of the engine stage)
__global__ void blu_classify_tasks(const uint64_t* __restrict__ seg_off, uint64_t n_queries, uint64_t n_hits, uint64_t, uint64_t, uint32_t, uint32_t*, uint32_t*);

void warm_up_device(int device) {
    void* p = nullptr;
    if (hipSetDevice(device) == hipSuccess && hipMalloc(&p, 256) == hipSuccess) {
        (void)hipMemsetAsync(p, 0, 256, nullptr);
        // one kernel of this library: its code object (5 MB, two dozen builds of the consensus kernels) is loaded onto the
        // device by the first launch — 8-10 ms that otherwise sit in front of the line index
        hipLaunchKernelGGL(iota_u32, dim3(1), dim3(64), 0, nullptr, (uint32_t*)p, 64u);
        hipFuncAttributes attr;
        (void)hipFuncGetAttributes(&attr, (const void*)blu_classify_tasks);
        (void)hipStreamSynchronize(nullptr);
        (void)hipFree(p);
    }
    (void)hipGetLastError();
}

int load_hits_gpu(int fd, size_t size, const TaxidMap& row_of, int device, bool host_columns, HitTable& ht, std::string* why) {
    int rc = BLU_OK;
    DeviceArena mem;
    const bool oom_fallback = true;
    std::string* const oom_why = why;
    const bool trace = getenv("BLU_INGEST_TRACE") != nullptr;
    double tp = now_s();
    auto lap = [&](const char* what) {
        if (!trace) return;
        (void)hipDeviceSynchronize();
        const double t = now_s();
        fprintf(stderr, "[ingest-gpu] %-26s %.3f s\n", what, t - tp);
        tp = t;
    };
    auto fallback = [&](const char* reason) { if (why) *why = reason; return BLU_INGEST_FALLBACK; };
    if (size == 0) return fallback("an empty file");
    if (size >= (1ull << 44)) return fallback("a file of 16 TiB or more");
    if (hipSetDevice(device) != hipSuccess) { set_error("hipSetDevice(%d) failed", device); return BLU_ERR_NO_DEVICE; }
    {
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && (double)size * 2.8 + (1ull << 30) > (double)free_b)
            return fallback("a file too large for this device's free memory");
        mem.keep = (double)size * 4.0 + (4ull << 30) < (double)free_b;
    }

    unsigned char* d_text = nullptr;
    uint32_t *d_tile = nullptr, *d_tile_base = nullptr, *d_flags = nullptr, *d_counter = nullptr;
    uint64_t* d_line = nullptr;
    void* d_tmp = nullptr;
    size_t tmp_bytes = 0;
    TaxidMap::E* d_taxmap = nullptr;
    unsigned long long *d_qh = nullptr, *d_ah = nullptr, *d_qpos = nullptr, *d_apos = nullptr, *d_big = nullptr, *d_poslist = nullptr, *d_aposlist = nullptr;
    uint32_t *d_tax = nullptr, *d_qid = nullptr, *d_arank = nullptr, *d_list_row = nullptr, *d_list_slot = nullptr, *d_alist_row = nullptr, *d_alist_slot = nullptr, *d_mark = nullptr,
             *d_rank_of_row = nullptr, *d_perm = nullptr, *d_perm2 = nullptr, *d_qid2 = nullptr, *d_ranks = nullptr;
    double *d_pid = nullptr, *d_pid2 = nullptr;
    int32_t *d_aln = nullptr, *d_bs = nullptr, *d_aln2 = nullptr, *d_bs2 = nullptr;
    uint32_t *d_tax2 = nullptr, *d_arank2 = nullptr;
    Slot *d_qtab = nullptr, *d_atab = nullptr;
    unsigned long long* d_seg = nullptr;
    uint32_t h_flags = 0, n_rows = 0, n_queries = 0, n_acc = 0;
    // the distinct query / accession strings packed back to back (bytes + n + 1 offsets) and the accessions' byte order:
    // what the strings thread started at the end needs
    std::vector<char> q_bytes, a_bytes;
    std::vector<unsigned long long> q_off, a_off;
    std::vector<uint32_t> order;
    std::vector<unsigned long long> k0, k1;      // first 16 bytes of every distinct accession as two big-endian words
    uint64_t acap = 0;                           // slots of the accession table
    std::thread acc_sort;                        // the host's sort of the distinct accessions, beside the query dictionary
    std::atomic<bool> acc_sort_failed{false};
    struct JoinSort { std::thread& t; ~JoinSort() { if (t.joinable()) t.join(); } } join_acc_sort{acc_sort};
    uint64_t n_tiles = (size + TILE_BYTES - 1) / TILE_BYTES;
    auto need_tmp = [&](size_t bytes) -> hipError_t {
        if (bytes <= tmp_bytes) return hipSuccess;
        if (d_tmp) mem.free(d_tmp);
        d_tmp = nullptr; tmp_bytes = 0;
        hipError_t e = mem.alloc(&d_tmp, bytes);
        if (e == hipSuccess) tmp_bytes = bytes;
        return e;
    };
    auto grid = [](uint64_t n, uint32_t b = 256) { return dim3((unsigned)((n + b - 1) / b)); };
    // distinct strings at d_pos[0 .. n) of the device text -> host
    auto download_strings = [&](const unsigned long long* d_pos, uint32_t n, std::vector<char>& bytes, std::vector<unsigned long long>& off) -> int {
        int rc = BLU_OK;
        unsigned long long *d_len = nullptr, *d_off = nullptr;
        unsigned char* d_out = nullptr;
        off.assign((size_t)n + 1, 0);
        HIPCHK(mem.alloc((void**)&d_len, ((size_t)n + 1) * 8));
        HIPCHK(mem.alloc((void**)&d_off, ((size_t)n + 1) * 8));
        hipLaunchKernelGGL(pos_lengths, dim3((n + 256) / 256), dim3(256), 0, 0, d_pos, n, d_len);
        {
            HIPCHK(need_tmp(scan_tmp_bytes<unsigned long long>((size_t)n + 1)));
            HIPCHK(exclusive_scan_dev<unsigned long long>(d_len, d_off, (size_t)n + 1, d_tmp));
        }
        HIPCHK(hipMemcpy(off.data(), d_off, ((size_t)n + 1) * 8, hipMemcpyDeviceToHost));
        bytes.resize(off[n]);
        HIPCHK(mem.alloc((void**)&d_out, std::max<size_t>(off[n], 16)));
        if (n) hipLaunchKernelGGL(gather_bytes, dim3((n + 255) / 256), dim3(256), 0, 0, d_pos, n, d_text, d_off, d_out);
        HIPCHK(hipMemcpy(bytes.data(), d_out, off[n], hipMemcpyDeviceToHost));
    done:
        mem.free(d_len); mem.free(d_off); mem.free(d_out);
        return rc;
    };

    // ---- upload + line index
    lap("device start-up");
    HIPCHK(mem.alloc((void**)&d_text, size + 64));
    HIPCHK(hipMemset(d_text + size, 0, 64));
    lap("  upload: device buffer");   // the padding only (the readers' pieces end at `size`)
    {
        std::string io;
        rc = upload_file(fd, size, d_text, device, &io);
        if (rc == BLU_INGEST_FALLBACK) { if (why) *why = "the pinned staging path failed (" + io + ")"; goto done; }
        if (rc != BLU_OK) { set_error("GPU ingest: reading the table failed: %s", io.c_str()); goto done; }
    }
    lap("upload text");
    HIPCHK(mem.alloc((void**)&d_tile, (n_tiles + 1) * 4));
    HIPCHK(mem.alloc((void**)&d_tile_base, (n_tiles + 1) * 4));
    HIPCHK(mem.alloc((void**)&d_flags, 64));
    HIPCHK(hipMemset(d_flags, 0, 64));
    d_counter = d_flags + 4;                                         // {flags, -, -, -, counter, -, big counters at +8}
    d_big = reinterpret_cast<unsigned long long*>(d_flags + 8);      // [0] unmatched rows, [1] run heads
    hipLaunchKernelGGL(count_newlines, grid(n_tiles, 1), dim3(TILE_THREADS), 0, 0, (const uint4*)d_text, (uint64_t)size, d_tile);
    {
        HIPCHK(need_tmp(scan_tmp_bytes<uint32_t>((size_t)n_tiles + 1)));
        HIPCHK(hipMemset(d_tile + n_tiles, 0, 4));
        HIPCHK(exclusive_scan_dev<uint32_t>(d_tile, d_tile_base, (size_t)n_tiles + 1, d_tmp));
    }
    {
        uint32_t n_newlines = 0;
        HIPCHK(hipMemcpy(&n_newlines, d_tile_base + n_tiles, 4, hipMemcpyDeviceToHost));
        char last = 0;
        if (pread(fd, &last, 1, (off_t)(size - 1)) != 1) { set_error("GPU ingest: reading the table failed"); rc = BLU_ERR_IO; goto done; }
        const bool open_tail = last != '\n';
        const uint64_t rows64 = (uint64_t)n_newlines + (open_tail ? 1 : 0);
        if (rows64 >= 0x7FFFFFF0ull) { rc = fallback("2^31 rows or more"); goto done; }
        n_rows = (uint32_t)rows64;
        if (n_rows == 0) { rc = fallback("no rows"); goto done; }
        HIPCHK(mem.alloc((void**)&d_line, ((size_t)n_rows + 2) * 8));
        HIPCHK(hipMemset(d_line, 0, 8));
        hipLaunchKernelGGL(write_line_starts, grid(n_tiles, 1), dim3(TILE_THREADS), 0, 0, (const uint4*)d_text, (uint64_t)size, d_tile_base, d_line);
        if (open_tail) { const uint64_t end = size + 1; HIPCHK(hipMemcpy(d_line + n_rows, &end, 8, hipMemcpyHostToDevice)); }
    }
    lap("line index");

    // ---- parse
    HIPCHK(mem.alloc((void**)&d_taxmap, row_of.tab.size() * sizeof(TaxidMap::E)));
    HIPCHK(hipMemcpy(d_taxmap, row_of.tab.data(), row_of.tab.size() * sizeof(TaxidMap::E), hipMemcpyHostToDevice));
    HIPCHK(mem.alloc((void**)&d_qh, (size_t)n_rows * 8)); HIPCHK(mem.alloc((void**)&d_ah, (size_t)n_rows * 8));
    HIPCHK(mem.alloc((void**)&d_qpos, (size_t)n_rows * 8)); HIPCHK(mem.alloc((void**)&d_apos, (size_t)n_rows * 8));
    HIPCHK(mem.alloc((void**)&d_tax, (size_t)n_rows * 4)); HIPCHK(mem.alloc((void**)&d_pid, (size_t)n_rows * 8));
    HIPCHK(mem.alloc((void**)&d_aln, (size_t)n_rows * 4)); HIPCHK(mem.alloc((void**)&d_bs, (size_t)n_rows * 4));
    {
        RowOut o{d_qh, d_ah, d_qpos, d_apos, d_tax, d_pid, d_aln, d_bs};
        DevTaxidMap tm{d_taxmap, row_of.tab.size() - 1};
        hipLaunchKernelGGL(parse_rows, grid(n_rows, PARSE_THREADS), dim3(PARSE_THREADS), 0, 0, d_text, d_line, n_rows, tm, o, d_flags, d_big);
        HIPCHK(hipMemcpy(&h_flags, d_flags, 4, hipMemcpyDeviceToHost));
        if (h_flags) { rc = fallback(fallback_text(h_flags)); goto done; }
    }
    lap("parse");

    // ---- accession dictionary, first half: distinct count unknown; the table grows until the load stays under one half.
    // (It comes before the query dictionary so that the host's sort of the distinct accessions — 12 ms for 300 k — runs on a
    // thread of its own while the device builds the query dictionary.)
    {
        uint64_t cap = pow2_at_least(std::min<uint64_t>((uint64_t)n_rows * 2 + 16, 1ull << 20));   // (32 MB: stays in the caches while 100 M rows probe it; x4 when more than half full)
        for (;;) {
            HIPCHK(mem.alloc((void**)&d_atab, cap * sizeof(Slot)));
            lap("  acc: table allocation");
            hipLaunchKernelGGL(dict_init, grid(cap), dim3(256), 0, 0, d_atab, cap);
            HIPCHK(hipMemset(d_counter, 0, 4));
            HIPCHK(hipMemset(d_flags, 0, 4));
            hipLaunchKernelGGL(dict_insert, grid(n_rows), dim3(256), 0, 0, d_ah, n_rows, d_atab, cap - 1, false, d_counter, d_flags);
            hipLaunchKernelGGL(dict_count, grid(cap, 1024), dim3(1024), 0, 0, d_atab, cap, d_counter);
            HIPCHK(hipMemcpy(&n_acc, d_counter, 4, hipMemcpyDeviceToHost));
            HIPCHK(hipMemcpy(&h_flags, d_flags, 4, hipMemcpyDeviceToHost));
            if (!(h_flags & FB_TABLE_FULL) && (uint64_t)n_acc * 2 <= cap) break;
            mem.free(d_atab); d_atab = nullptr;
            if (cap >= pow2_at_least((uint64_t)n_rows * 2 + 16)) { rc = fallback(fallback_text(FB_TABLE_FULL)); goto done; }
            cap *= 4;
        }
        lap("  acc: insert");
        HIPCHK(hipMemset(d_flags, 0, 4));
        HIPCHK(mem.alloc((void**)&d_alist_row, (size_t)n_acc * 4)); HIPCHK(mem.alloc((void**)&d_alist_slot, (size_t)n_acc * 4));
        HIPCHK(hipMemset(d_counter, 0, 4));
        hipLaunchKernelGGL(dict_finalize, grid(cap), dim3(256), 0, 0, d_atab, cap, d_text, d_apos, d_alist_row, d_alist_slot, d_counter);
        HIPCHK(mem.alloc((void**)&d_aposlist, (size_t)n_acc * 8));
        hipLaunchKernelGGL(gather_pos, grid(n_acc), dim3(256), 0, 0, d_apos, d_alist_row, n_acc, d_aposlist);
        rc = download_strings(d_aposlist, n_acc, a_bytes, a_off);
        if (rc != BLU_OK) goto done;
        lap("  acc: distinct to host");
        // byte order of the distinct accessions (String::cmp), on the host: only the distinct strings are touched, and
        // mostly not even those — the GPU hands over their first 16 bytes as two big-endian integers; the text is read
        // only to order keys that agree on all 16
        k0.resize(n_acc); k1.resize(n_acc);
        {
            unsigned long long *d_k0 = nullptr, *d_k1 = nullptr;
            HIPCHK(mem.alloc((void**)&d_k0, (size_t)n_acc * 8 + 8));
            hipError_t e2 = mem.alloc((void**)&d_k1, (size_t)n_acc * 8 + 8);
            if (e2 == hipSuccess) {
                hipLaunchKernelGGL(gather_key16, grid(n_acc), dim3(256), 0, 0, d_aposlist, n_acc, d_text, d_k0, d_k1);
                e2 = hipMemcpy(k0.data(), d_k0, (size_t)n_acc * 8, hipMemcpyDeviceToHost);
                if (e2 == hipSuccess) e2 = hipMemcpy(k1.data(), d_k1, (size_t)n_acc * 8, hipMemcpyDeviceToHost);
            }
            mem.free(d_k0);
            mem.free(d_k1);
            HIPCHK(e2);
        }
        acap = cap;
        order.resize(n_acc);
        for (uint32_t k = 0; k < n_acc; ++k) order[k] = k;
        acc_sort = std::thread([&]() {
            try {
                auto view = [&](uint32_t k) { return std::string_view(a_bytes.data() + a_off[k], (size_t)(a_off[k + 1] - a_off[k])); };
                auto less = [&](uint32_t a, uint32_t b) {
                    if (k0[a] != k0[b]) return k0[a] < k0[b];
                    if (k1[a] != k1[b]) return k1[a] < k1[b];
                    const size_t la = (size_t)(a_off[a + 1] - a_off[a]), lb = (size_t)(a_off[b + 1] - a_off[b]);
                    if (la <= 16 && lb <= 16) return la < lb;          // equal padded prefixes: the shorter string sorts first
                    return view(a) < view(b);
                };
                {
                    unsigned nt = std::thread::hardware_concurrency();
                    if (const char* env = getenv("BLU_INGEST_THREADS")) nt = (unsigned)atoi(env);
                    nt = std::max(1u, std::min(nt, 32u));
                    if (n_acc < 65536) nt = 1;
                    std::vector<std::thread> pool;
                    for (unsigned t = 0; t < nt; ++t)
                        pool.emplace_back([&, t]() { std::sort(order.begin() + (size_t)n_acc * t / nt, order.begin() + (size_t)n_acc * (t + 1) / nt, less); });
                    for (auto& th : pool) th.join();
                    for (unsigned w = 1; w < nt; w *= 2)
                        for (unsigned t = 0; t + w < nt; t += 2 * w)
                            std::inplace_merge(order.begin() + (size_t)n_acc * t / nt, order.begin() + (size_t)n_acc * (t + w) / nt,
                                               order.begin() + (size_t)n_acc * std::min(t + 2 * w, nt) / nt, less);
                }
            } catch (...) { acc_sort_failed = true; }   // (no exception leaves a std::thread)
        });
    }

    // ---- query dictionary: sized by the number of runs of equal hashes (an upper bound of the distinct count)
    {
        hipLaunchKernelGGL(count_run_heads, grid(n_rows, 1024), dim3(1024), 0, 0, d_qh, n_rows, d_big + 1);
        unsigned long long runs = 0;
        HIPCHK(hipMemcpy(&runs, d_big + 1, 8, hipMemcpyDeviceToHost));
        const uint64_t cap = pow2_at_least(runs * 2 + 16);
        HIPCHK(mem.alloc((void**)&d_qtab, cap * sizeof(Slot)));
        hipLaunchKernelGGL(dict_init, grid(cap), dim3(256), 0, 0, d_qtab, cap);   // empty slots, first_row = all ones for atomicMin
        HIPCHK(hipMemset(d_counter, 0, 4));
        hipLaunchKernelGGL(dict_insert, grid(n_rows), dim3(256), 0, 0, d_qh, n_rows, d_qtab, cap - 1, true, d_counter, d_flags);
        hipLaunchKernelGGL(dict_count, grid(cap, 1024), dim3(1024), 0, 0, d_qtab, cap, d_counter);
        HIPCHK(hipMemcpy(&n_queries, d_counter, 4, hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(&h_flags, d_flags, 4, hipMemcpyDeviceToHost));
        if (h_flags) { rc = fallback(fallback_text(h_flags)); goto done; }
        lap("  query: insert");
        HIPCHK(mem.alloc((void**)&d_list_row, (size_t)n_queries * 4)); HIPCHK(mem.alloc((void**)&d_list_slot, (size_t)n_queries * 4));
        HIPCHK(hipMemset(d_counter, 0, 4));
        hipLaunchKernelGGL(dict_finalize, grid(cap), dim3(256), 0, 0, d_qtab, cap, d_text, d_qpos, d_list_row, d_list_slot, d_counter);
        // ids in first-appearance order, and the names' positions in id order
        HIPCHK(mem.alloc((void**)&d_mark, ((size_t)n_rows + 1) * 4)); HIPCHK(mem.alloc((void**)&d_rank_of_row, ((size_t)n_rows + 1) * 4));
        HIPCHK(hipMemset(d_mark, 0, ((size_t)n_rows + 1) * 4));
        hipLaunchKernelGGL(mark_first_rows, grid(n_queries), dim3(256), 0, 0, (const uint32_t*)d_list_row, n_queries, d_mark);
        HIPCHK(need_tmp(scan_tmp_bytes<uint32_t>((size_t)n_rows + 1)));
        HIPCHK(exclusive_scan_dev<uint32_t>(d_mark, d_rank_of_row, (size_t)n_rows + 1, d_tmp));
        HIPCHK(mem.alloc((void**)&d_poslist, (size_t)n_queries * 8));
        hipLaunchKernelGGL(number_queries, grid(n_queries), dim3(256), 0, 0, d_qtab, (const uint32_t*)d_list_row, (const uint32_t*)d_list_slot, n_queries,
                           (const uint32_t*)d_rank_of_row, (const unsigned long long*)d_qpos, d_poslist);
        HIPCHK(mem.alloc((void**)&d_qid, (size_t)n_rows * 4));
        hipLaunchKernelGGL(dict_lookup, grid(n_rows), dim3(256), 0, 0, d_qh, d_qpos, n_rows, d_qtab, cap - 1, d_text, d_qpos, d_qid, d_flags);
        HIPCHK(hipMemcpy(&h_flags, d_flags, 4, hipMemcpyDeviceToHost));
        if (h_flags) { rc = fallback(fallback_text(h_flags)); goto done; }
        lap("  query: ids of the rows");
        // query names: the text of each query's first row, in id order
        rc = download_strings(d_poslist, n_queries, q_bytes, q_off);
        if (rc != BLU_OK) goto done;
        lap("  query: names");
        mem.free(d_poslist); d_poslist = nullptr;
        mem.free(d_list_row); mem.free(d_list_slot); mem.free(d_mark); mem.free(d_rank_of_row);
        d_list_row = d_list_slot = d_mark = d_rank_of_row = nullptr;
        mem.free(d_qtab); d_qtab = nullptr;
        mem.free(d_qh); d_qh = nullptr;
        mem.free(d_qpos); d_qpos = nullptr;
    }
    lap("query dictionary");

    // ---- accession dictionary, second half: the ranks go up, every row gets its accession's
    {
        acc_sort.join();
        if (acc_sort_failed) { rc = fallback("the host could not sort the accessions"); goto done; }
        lap("  acc: wait for the host sort");
        const uint64_t cap = acap;
        std::vector<uint32_t> rank_of(n_acc);
        for (uint32_t r = 0; r < n_acc; ++r) rank_of[order[r]] = r;
        HIPCHK(mem.alloc((void**)&d_ranks, (size_t)n_acc * 4));
        HIPCHK(hipMemcpy(d_ranks, rank_of.data(), (size_t)n_acc * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(dict_assign_ids, grid(n_acc), dim3(256), 0, 0, d_atab, d_alist_slot, (const uint32_t*)d_ranks, n_acc);
        HIPCHK(mem.alloc((void**)&d_arank, (size_t)n_rows * 4));
        hipLaunchKernelGGL(dict_lookup, grid(n_rows), dim3(256), 0, 0, d_ah, d_apos, n_rows, d_atab, cap - 1, d_text, d_apos, d_arank, d_flags);
        HIPCHK(hipMemcpy(&h_flags, d_flags, 4, hipMemcpyDeviceToHost));
        if (h_flags) { rc = fallback(fallback_text(h_flags)); goto done; }
        lap("  acc: ranks of the rows");
        mem.free(d_atab); d_atab = nullptr;
        mem.free(d_ah); d_ah = nullptr;
        mem.free(d_apos); d_apos = nullptr;
        mem.free(d_text); d_text = nullptr;
    }
    lap("accession dictionary");

    // ---- grouping: queries in first-appearance order, file order inside a query (mod.rs:192-208)
    {
        HIPCHK(hipMemset(d_counter, 0, 4));
        hipLaunchKernelGGL(check_grouped, grid(n_rows), dim3(256), 0, 0, d_qid, n_rows, d_counter);
        uint32_t unsorted = 0;
        HIPCHK(hipMemcpy(&unsorted, d_counter, 4, hipMemcpyDeviceToHost));
        lap(unsorted ? "  grouping: check (unsorted)" : "  grouping: check (sorted)");
        if (unsorted) {
            HIPCHK(mem.alloc((void**)&d_perm, (size_t)n_rows * 4)); HIPCHK(mem.alloc((void**)&d_perm2, (size_t)n_rows * 4));
            HIPCHK(mem.alloc((void**)&d_qid2, (size_t)n_rows * 4));
            hipLaunchKernelGGL(iota_u32, grid(n_rows), dim3(256), 0, 0, d_perm, n_rows);
            int bits = 1;
            while ((1ull << bits) < n_queries) ++bits;
            const size_t cells = (size_t)256 * (((size_t)n_rows + RS_BLOCK - 1) / RS_BLOCK);
            uint32_t* d_table = nullptr;
            HIPCHK(mem.alloc((void**)&d_table, 2 * cells * 4));
            HIPCHK(need_tmp(scan_tmp_bytes<uint32_t>(cells)));
            // (stable: file order survives inside a query; afterwards d_qid2 / d_perm2 name the sorted arrays whichever buffer they are)
            uint32_t *k0 = d_qid, *k1 = d_qid2, *v0 = d_perm, *v1 = d_perm2;
            HIPCHK(radix_sort_pairs_dev(&k0, &k1, &v0, &v1, n_rows, bits, d_table, d_tmp));
            d_qid2 = k0; d_qid = k1; d_perm2 = v0; d_perm = v1;
            mem.free(d_table);
        }
        HIPCHK(mem.alloc((void**)&d_seg, ((size_t)n_queries + 1) * 8 * 2));
        HIPCHK(hipMemset(d_seg, 0, ((size_t)n_queries + 1) * 8 * 2));
        hipLaunchKernelGGL(segment_starts, grid(n_rows), dim3(256), 0, 0, (const uint32_t*)(unsorted ? d_qid2 : d_qid), n_rows, n_queries,
                           d_seg + n_queries + 1);
        HIPCHK(mem.alloc((void**)&d_bs2, (size_t)n_rows * 4)); HIPCHK(mem.alloc((void**)&d_aln2, (size_t)n_rows * 4));
        HIPCHK(mem.alloc((void**)&d_tax2, (size_t)n_rows * 4)); HIPCHK(mem.alloc((void**)&d_arank2, (size_t)n_rows * 4));
        HIPCHK(mem.alloc((void**)&d_pid2, (size_t)n_rows * 8));
        lap("  grouping: offsets + allocations");
        Cols in{d_bs, d_aln, d_tax, d_arank, d_pid};
        ColsOut out{d_bs2, d_aln2, d_tax2, d_arank2, d_pid2};
        hipLaunchKernelGGL(gather_cols, grid(n_rows), dim3(256), 0, 0, in, out, (const uint32_t*)(unsorted ? d_perm2 : nullptr), n_rows);
    }
    lap("grouping");

    // ---- the grouped columns stay on the device for the engine; the host copies are made only on request
    {
        unsigned long long unmatched = 0;
        HIPCHK(hipMemcpy(&unmatched, d_big, 8, hipMemcpyDeviceToHost));
        ht.unmatched = unmatched;
        ht.n_hits = n_rows;
        ht.host_columns = false;
        ht.dev.reset(new DeviceHits());
        ht.dev->device = device; ht.dev->n_hits = n_rows; ht.dev->n_queries = n_queries;
        ht.dev->bitscore = d_bs2; ht.dev->align_len = d_aln2; ht.dev->tax_desc_row = d_tax2; ht.dev->acc_rank = d_arank2; ht.dev->pident = d_pid2;
        ht.dev->seg_off = d_seg + n_queries + 1; ht.dev->seg_block = d_seg;
        for (void* q : {(void*)d_bs2, (void*)d_aln2, (void*)d_tax2, (void*)d_arank2, (void*)d_pid2, (void*)d_seg}) mem.release(q);
        if (host_columns) {
            rc = download_columns(ht);
            if (rc != BLU_OK) goto done;
            lap("download columns");
        }
        // the host strings (query names in id order, accessions in byte order) are built from the packed bytes by a
        // background thread while the caller goes on to the engine: nothing on the device waits for them
        ht.n_queries = n_queries;
        unsigned nt = std::thread::hardware_concurrency();
        if (const char* env = getenv("BLU_INGEST_THREADS")) nt = (unsigned)atoi(env);
        nt = std::max(1u, std::min(nt, 16u));
        if ((uint64_t)n_queries + n_acc < 65536) nt = 1;
        HitTable* const hp = &ht;
        ht.strings_thread = std::thread([hp, nt, q_bytes = std::move(q_bytes), q_off = std::move(q_off), a_bytes = std::move(a_bytes),
                                         a_off = std::move(a_off), order = std::move(order)]() {
            std::atomic<bool> oom{false};            // (an exception must not leave a std::thread: the caller checks strings_ok)
            // (sizing 2 M strings is 64 MB of fresh pages: 10 ms that need not stand between the ingest and the engine)
            try { hp->query_names.resize(q_off.size() - 1); hp->accessions.resize(a_off.size() - 1); }
            catch (const std::bad_alloc&) { hp->strings_ok = false; return; }
            auto work = [&](unsigned t) {
                try {
                    const size_t nq = q_off.size() - 1, na = a_off.size() - 1;
                    for (size_t q = nq * t / nt; q < nq * (t + 1) / nt; ++q)
                        hp->query_names[q].assign(q_bytes.data() + q_off[q], (size_t)(q_off[q + 1] - q_off[q]));
                    for (size_t r = na * t / nt; r < na * (t + 1) / nt; ++r) {
                        const uint32_t k = order[r];
                        hp->accessions[r].assign(a_bytes.data() + a_off[k], (size_t)(a_off[k + 1] - a_off[k]));
                    }
                } catch (const std::bad_alloc&) { oom = true; }
            };
            try {
                std::vector<std::thread> pool;
                struct JoinAll { std::vector<std::thread>& p; ~JoinAll() { for (auto& th : p) if (th.joinable()) th.join(); } } join_all{pool};
                for (unsigned t = 1; t < nt; ++t) pool.emplace_back(work, t);
                work(0);
            } catch (...) { oom = true; }
            if (oom) hp->strings_ok = false;
        });
    }

done:
    if (rc != BLU_OK) {
        ht.clear();
    }
    lap("hand-over");
    // with room on the card (mem.keep) the work buffers — the text, the hashes, the dictionaries: 10 GB for a 2 M-query table —
    // are freed with the columns, off the caller's path (5-7 ms of hipFree); otherwise here
    if (rc == BLU_OK && mem.keep && ht.dev) { ht.dev->trash.insert(ht.dev->trash.end(), mem.ptrs.begin(), mem.ptrs.end()); mem.ptrs.clear(); }
    mem.free_all();
    lap("free the work buffers");
    return rc;
}


// Device -> pageable host memory in 8 MiB pieces by a pool of host threads: the first touch of the fresh destination pages
// (and the runtime's pinning of them) costs more than the transfer, and it parallelises.
struct D2HPiece { char* dst; const char* src; size_t bytes; };
static void d2h_add(std::vector<D2HPiece>& v, void* dst, const void* src, size_t bytes, size_t piece = 8u << 20) {
    for (size_t o = 0; o < bytes; o += piece) v.push_back({(char*)dst + o, (const char*)src + o, std::min(piece, bytes - o)});
}
static hipError_t d2h_parallel(const std::vector<D2HPiece>& pieces, int device, unsigned max_threads = 16) {
    if (pieces.empty()) return hipSuccess;
    unsigned nt = std::thread::hardware_concurrency();
    if (const char* env = getenv("BLU_INGEST_THREADS")) nt = (unsigned)atoi(env);
    nt = std::max(1u, std::min<unsigned>(std::min(nt, max_threads), (unsigned)pieces.size()));
    std::vector<hipError_t> errs(nt, hipSuccess);
    std::atomic<size_t> next{0};
    auto work = [&](unsigned t) {
        (void)hipSetDevice(device);
        for (size_t k = next.fetch_add(1); k < pieces.size() && errs[t] == hipSuccess; k = next.fetch_add(1))
            errs[t] = hipMemcpy(pieces[k].dst, pieces[k].src, pieces[k].bytes, hipMemcpyDeviceToHost);
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < nt; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto& th : pool) th.join();
    for (hipError_t e : errs) if (e != hipSuccess) return e;
    return hipSuccess;
}

// The columns come back in 64 MiB pieces copied by a pool of host threads: the first touch of the fresh host pages
// costs more than the transfer, and it parallelises (the vectors are resized without initialisation).
int download_columns(HitTable& ht) {
    if (ht.host_columns) return BLU_OK;
    if (!ht.dev) { set_error("download_columns: no device columns"); return BLU_ERR_INVALID_ARG; }
    const DeviceHits& d = *ht.dev;
    const size_t n_rows = d.n_hits, n_queries = d.n_queries;
    ht.seg_off.resize(n_queries + 1);
    ht.bitscore.resize(n_rows); ht.align_len.resize(n_rows); ht.tax_desc_row.resize(n_rows); ht.acc_rank.resize(n_rows); ht.pident.resize(n_rows);
    std::vector<D2HPiece> pieces;
    auto add = [&](void* dst, const void* src, size_t bytes) { d2h_add(pieces, dst, src, bytes, 64u << 20); };
    add(ht.seg_off.data(), d.seg_off, (n_queries + 1) * 8);
    add(ht.bitscore.data(), d.bitscore, n_rows * 4); add(ht.align_len.data(), d.align_len, n_rows * 4);
    add(ht.tax_desc_row.data(), d.tax_desc_row, n_rows * 4); add(ht.acc_rank.data(), d.acc_rank, n_rows * 4);
    add(ht.pident.data(), d.pident, n_rows * 8);
    // (pinning the destination pieces first — hipHostRegister — made it slower: 0.33 s instead of 0.21 s for 2 GB)
    const hipError_t e = d2h_parallel(pieces, d.device);
    if (e != hipSuccess) { set_error("GPU ingest: column download failed: %s", hipGetErrorString(e)); return BLU_ERR_HIP; }
    ht.host_columns = true;
    return BLU_OK;
}

// ---- engine on the resident columns ---------------------------------------------------------------------------
namespace {
constexpr uint32_t ROW_NO_TAXID = BLU_UNMATCHED_TAXID;
__device__ __forceinline__ uint32_t engine_row_of(uint32_t r, const uint32_t* __restrict__ fwd, uint64_t n_tax) {
    return (r == ROW_NO_TAXID || r >= n_tax) ? ROW_NO_TAXID : fwd[r];
}
// k = round(p * 1000) is used only if fl(k / 1000.0) == p bit for bit for every row (the 20 B/hit layout is lossless then)
__device__ __forceinline__ bool milli_of(double p, uint32_t* k_out) {
    bool ok = p >= 0.0 && p < 4.0e6;
    uint32_t k = 0;
    if (ok) {
        k = (uint32_t)(p * 1000.0 + 0.5);
        const double back = (double)k / 1000.0;
        ok = __double_as_longlong(back) == __double_as_longlong(p);
    }
    *k_out = k;
    return ok;
}
// 16-byte side records of the packed layout (include/blu_consensus.h: blu_hits.packed) straight from the ingest's columns
// (what blu_hits_pack builds from engine row ids, pack_kernel.hip, here straight from the taxonomy rows of the join: word 1 =
// milli-percent perc_identity | shape hint of the row << 17)
__global__ void pack_side_records(const uint32_t* __restrict__ desc_row, const double* __restrict__ pid, const int32_t* __restrict__ aln,
                                  const uint32_t* __restrict__ acc, uint64_t n, const uint32_t* __restrict__ fwd, uint64_t n_tax,
                                  const uint16_t* __restrict__ hint_of_pos, uint4* __restrict__ rec, uint32_t* __restrict__ inexact) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t k;
    if (!milli_of(pid[i], &k) || k >= BLU_PACKED_PIDENT_LIMIT) *inexact = 1u;
    const uint32_t row = engine_row_of(desc_row[i], fwd, n_tax);
    const uint32_t pos = row & ((1u << BLU_ROW_BITS) - 1u);
    const uint32_t hint = (row != ROW_NO_TAXID && pos < n_tax) ? (uint32_t)hint_of_pos[pos] : 0u;
    rec[i] = make_uint4(row, (k & BLU_KTHR_NEVER) | (hint << BLU_KTHR_BITS), (uint32_t)aln[i], acc[i]);
}
__global__ void to_engine_rows(const uint32_t* __restrict__ desc_row, uint64_t n, const uint32_t* __restrict__ fwd, uint64_t n_tax,
                               uint32_t* __restrict__ rows) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) rows[i] = engine_row_of(desc_row[i], fwd, n_tax);
}
__global__ void to_milli(const double* __restrict__ pid, uint64_t n, uint32_t* __restrict__ milli, uint32_t* __restrict__ inexact) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t k;
    if (!milli_of(pid[i], &k)) *inexact = 1u;
    milli[i] = k;
}

// Top-score rows of the rendered queries (status 0 / 1), one wave per TOP_QPW consecutive queries: the lanes sweep a
// segment 64 rows at a time.  Pass 1 counts (out_rows == nullptr), pass 2 writes the rows at the scanned offsets.
constexpr uint32_t TOP_QPW = 8;
__global__ __launch_bounds__(256) void top_rows_kernel(const blu_result* __restrict__ recs, const unsigned long long* __restrict__ seg_off,
                                                       const int32_t* __restrict__ bitscore, uint64_t n_queries,
                                                       unsigned long long* __restrict__ count, const unsigned long long* __restrict__ off,
                                                       const uint32_t* __restrict__ desc_row, const uint32_t* __restrict__ acc,
                                                       const int32_t* __restrict__ aln, const double* __restrict__ pid,
                                                       TopRow* __restrict__ out_rows, int32_t* __restrict__ out_score) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    for (uint32_t k = 0; k < TOP_QPW; ++k) {
        const uint64_t q = wave * TOP_QPW + k;
        if (q >= n_queries) return;
        const blu_result r = recs[q];
        unsigned long long n_top = 0;
        if (r.status < 2 && r.ref_row != 0xFFFFFFFFu) {
            const int32_t top = bitscore[r.ref_row];
            const unsigned long long b = seg_off[q], e = seg_off[q + 1];
            const unsigned long long base = out_rows ? off[q] : 0;
            for (unsigned long long i0 = b; i0 < e; i0 += 64) {
                const unsigned long long i = i0 + lane;
                const bool hit = i < e && bitscore[i] == top;
                const unsigned long long m = __ballot(hit);
                if (out_rows && hit) {
                    const unsigned long long at = base + n_top + __popcll(m & ((1ull << lane) - 1));
                    TopRow t;
                    t.row = (uint32_t)i; t.desc_row = desc_row[i]; t.acc_rank = acc[i]; t.align_len = aln[i]; t.pident = pid[i];
                    out_rows[at] = t;
                }
                n_top += __popcll(m);
            }
            if (out_score && lane == 0) out_score[q] = top;
        } else if (out_score && lane == 0) out_score[q] = 0;
        if (!out_rows && lane == 0) count[q] = n_top;
    }
}
}  // namespace

DeviceHits::~DeviceHits() {
    if (device < 0) return;
    (void)hipSetDevice(device);
    for (void* p : {(void*)bitscore, (void*)align_len, (void*)tax_desc_row, (void*)acc_rank, (void*)pident, seg_block})
        if (p) (void)hipFree(p);
    for (void* p : trash) (void)hipFree(p);
}

int device_run_consensus(const blu_taxonomy* tax, DeviceHits& dev, const uint32_t* fwd, uint64_t n_tax, int strategy, blu_result* out,
                         TopTable* top) {
    int rc = BLU_OK;
    const bool oom_fallback = false;
    std::string* const oom_why = nullptr;
    uint32_t *d_fwd = nullptr, *d_milli = nullptr, *d_rows = nullptr, *d_flag = nullptr;
    uint4* d_rec = nullptr;
    blu_result* d_out = nullptr;
    unsigned long long* d_cnt = nullptr;   // [2 (Q + 1)]: counts, then their exclusive scan
    void* d_tmp = nullptr;
    TopRow* d_top = nullptr;
    int32_t* d_score = nullptr;
    uint32_t inexact = 0;
    const uint64_t n = dev.n_hits, nq = dev.n_queries;
    auto grid = [](uint64_t m) { return dim3((unsigned)((m + 255) / 256)); };
    if (hipSetDevice(dev.device) != hipSuccess) { set_error("hipSetDevice(%d) failed", dev.device); return BLU_ERR_NO_DEVICE; }
    HIPCHK(hipMalloc((void**)&d_fwd, std::max<uint64_t>(n_tax, 1) * 4));
    HIPCHK(hipMemcpy(d_fwd, fwd, n_tax * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void**)&d_flag, 4));
    HIPCHK(hipMemset(d_flag, 0, 4));
    HIPCHK(hipMalloc((void**)&d_out, std::max<uint64_t>(nq, 1) * sizeof(blu_result)));
    {
        blu_hits h{};
        h.bitscore = dev.bitscore;
        h.seg_off = (const uint64_t*)dev.seg_off;
        // packed layout: a top row's four values in one memory line (if the records do not fit, or a perc_identity is not
        // exactly k/1000, the columns go in)
        bool packed = n && hipMalloc((void**)&d_rec, n * 16) == hipSuccess;
        if (!packed) (void)hipGetLastError();
        if (packed) {
            hipLaunchKernelGGL(pack_side_records, grid(n), dim3(256), 0, 0, dev.tax_desc_row, dev.pident, dev.align_len, dev.acc_rank, n,
                               d_fwd, n_tax, tax->d_hint_of_pos, d_rec, d_flag);
            HIPCHK(hipMemcpy(&inexact, d_flag, 4, hipMemcpyDeviceToHost));
            if (inexact) { (void)hipFree(d_rec); d_rec = nullptr; packed = false; }
        }
        if (packed) h.packed = (const uint32_t*)d_rec;
        else {
            HIPCHK(hipMalloc((void**)&d_rows, std::max<uint64_t>(n, 1) * 4));
            if (n) hipLaunchKernelGGL(to_engine_rows, grid(n), dim3(256), 0, 0, dev.tax_desc_row, n, d_fwd, n_tax, d_rows);
            h.tax_row = d_rows; h.align_len = dev.align_len; h.acc_rank = dev.acc_rank;
            if (!inexact && n) {   // the records did not fit: milli-percent column
                HIPCHK(hipMalloc((void**)&d_milli, n * 4));
                hipLaunchKernelGGL(to_milli, grid(n), dim3(256), 0, 0, dev.pident, n, d_milli, d_flag);
                HIPCHK(hipMemcpy(&inexact, d_flag, 4, hipMemcpyDeviceToHost));
            }
            if (inexact || !n) h.pident = dev.pident; else h.pident_milli = d_milli;
        }
        h.n_hits = n; h.n_queries = nq; h.on_device = 1;
        blu_run_params rp{strategy, 0, nullptr};
        rc = blu_consensus_run(tax, &h, &rp, d_out);
        if (rc != BLU_OK) goto done;
    }
    HIPCHK(hipStreamSynchronize(nullptr));
    {
        std::vector<D2HPiece> pieces;
        d2h_add(pieces, out, d_out, nq * sizeof(blu_result));
        HIPCHK(d2h_parallel(pieces, dev.device));
    }
    if (top) {
        const uint64_t waves = (nq + TOP_QPW - 1) / TOP_QPW;
        const dim3 g((unsigned)((waves * 64 + 255) / 256));
        HIPCHK(hipMalloc((void**)&d_cnt, (nq + 1) * 8 * 2));
        HIPCHK(hipMemset(d_cnt, 0, (nq + 1) * 8 * 2));
        HIPCHK(hipMalloc((void**)&d_score, std::max<uint64_t>(nq, 1) * 4));
        if (nq) hipLaunchKernelGGL(top_rows_kernel, g, dim3(256), 0, 0, d_out, dev.seg_off, dev.bitscore, nq, d_cnt, nullptr, nullptr, nullptr,
                                   nullptr, nullptr, nullptr, nullptr);
        HIPCHK(hipMalloc(&d_tmp, scan_tmp_bytes<unsigned long long>((size_t)nq + 1)));
        HIPCHK(exclusive_scan_dev<unsigned long long>(d_cnt, d_cnt + nq + 1, (size_t)nq + 1, d_tmp));
        top->off.resize(nq + 1);
        HIPCHK(hipMemcpy(top->off.data(), d_cnt + nq + 1, (nq + 1) * 8, hipMemcpyDeviceToHost));
        const uint64_t n_top = top->off[nq];
        HIPCHK(hipMalloc((void**)&d_top, std::max<uint64_t>(n_top, 1) * sizeof(TopRow)));
        if (nq) hipLaunchKernelGGL(top_rows_kernel, g, dim3(256), 0, 0, d_out, dev.seg_off, dev.bitscore, nq, nullptr, d_cnt + nq + 1,
                                   dev.tax_desc_row, dev.acc_rank, dev.align_len, dev.pident, d_top, d_score);
        top->rows.resize(n_top);
        top->score.resize(nq);
        HIPCHK(hipStreamSynchronize(nullptr));
        std::vector<D2HPiece> pieces;
        d2h_add(pieces, top->rows.data(), d_top, n_top * sizeof(TopRow));
        d2h_add(pieces, top->score.data(), d_score, nq * 4);
        HIPCHK(d2h_parallel(pieces, dev.device));
    }
done:
    // the work buffers are freed with the columns (DeviceHits::trash): ten hipFree calls were 10-15 ms of this function
    for (void* p : {(void*)d_fwd, (void*)d_milli, (void*)d_rows, (void*)d_flag, (void*)d_out, (void*)d_rec, (void*)d_cnt, d_tmp, (void*)d_top,
                    (void*)d_score})
        if (p) dev.trash.push_back(p);
    return rc;
}

}  // namespace blu
