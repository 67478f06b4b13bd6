// Per-query taxonomic consensus on gfx950 (CDNA4).
//
// Reference semantics: core/src/use_cases/build_consensus_identities/
// find_single_query_consensus.rs:17-173, find_multi_taxa_consensus.rs:22-217,
// build_blast_consensus_identity.rs:9-105 (restated in SURVEY §3.3).
//
// Kernel A  blu_consensus_stream_kernel — segments of up to 512 hits.
//   A wave task is 64 consecutive queries; the waves stride over the tasks on their own (no block-level barrier in the
//   loop); the offsets of a task are requested one task ahead.
//   phase 1, ring tasks (queries back to back, none over 128 rows): the bit-score column streams through a per-wave
//     LDS ring filled by LDS-DMA (256-row chunks, requested as far ahead as the ring has room — across task boundaries —
//     and waited for with counted s_waitcnt vmcnt); a lane scans 16 or 32 consecutive rows of its query out of the ring,
//     the maximum and the top-row counts cross lanes by DPP, and a lane with top rows leaves one descriptor in the list;
//     the list entries are worked out of the descriptors one entry per lane and the 16 other bytes of those rows
//     gathered, every load in flight at once.  A step whose top rows would not fit an empty list (whole groups tied)
//     is reduced by the lanes that scanned it (dense step), without the list.
//   phase 1, other tasks (lane = 4 consecutive hit rows): segments of up to 128 rows in steps of 64 lanes with 4 .. 32
//     lanes per query (chosen per task); segments of 129 .. 512 rows in a long pass of two 256-row slots per step.  A
//     step loads the bit-scores (16-byte buffer loads), finds the top score by a DPP reduction over the query's lanes
//     and ranks the top rows by a DPP scan; the other four values of a hit (taxonomy row, perc_identity, align_length,
//     accession rank — 16-byte side records in the packed layout, four columns otherwise) are then loaded for the top
//     rows only, and compacted in file order into the per-wave LDS list.
//   The row id carries the lineage length: no taxonomy lookup in phase 1.
//   phase 2a (lane = query, LDS only): parse errors, reference row by the stable-sort rule, shortest lineage, group-max
//     pident, span [lo, hi] in sorted lineage order — one uniform loop over the list.
//   phase 2c (lane = query): one 128-byte line of the reference row gives, per level, how far the neighbouring rows
//     share it (-> the first disagreeing level of the span, four levels per subtraction; a range minimum over the
//     adjacent-row LCP array for wide spans, requested together with the row) and the node ids; rank codes and cutoffs
//     come from the table rows of the lineage's shape — in the milli-percent layouts the cutoff tests are integer
//     compares against per-level thresholds; record.
//   Records are staged in LDS and leave as one write-through 2 KiB run per wave task.
//   Queries with more than 512 hits, or whose top group does not fit the LDS list, are appended to a worklist.
// Kernel B  blu_consensus_long_kernel — worklist queries, one wave per query: chunked passes over the bit-scores (any
//   length), top rows collected in LDS slots and their side records gathered one row per lane, wave-parallel finalisation.
//
// Integer/compare work only: no MFMA.  HBM-bound: every bit-score is read (4 B/hit), the other 16 B/hit for top rows
// only, one reference-row line and one 32-byte record per query.
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdint>
#include <type_traits>

#include "blu_internal.h"

namespace blu {

// Timing-only switches for attributing time / instruction counts to the phases of the stream kernel (make exp
// FLAGS="-DBLU_EXPERIMENTS -DBLU_X_...=true"): they produce WRONG records.  The product build leaves them all false and
// the compiler drops the tests.
#ifndef BLU_EXPERIMENTS
#define BLU_X_SKIP_PUSH false
#define BLU_X_SKIP_STORES false
#define BLU_X_SKIP_P1 false
#define BLU_X_SKIP_2A false
#define BLU_X_SKIP_2C false
#define BLU_X_SKIP_RUNLEN false
#define BLU_X_SKIP_LEVELS false
#define BLU_X_SKIP_GATHER false
#define BLU_X_SCAN_MIN false
#else
#ifndef BLU_X_SCAN_MIN
#define BLU_X_SCAN_MIN false
#endif
#ifndef BLU_X_SKIP_P1
#define BLU_X_SKIP_P1 false
#endif
#ifndef BLU_X_SKIP_PUSH
#define BLU_X_SKIP_PUSH false
#endif
#ifndef BLU_X_SKIP_STORES
#define BLU_X_SKIP_STORES false
#endif
#ifndef BLU_X_SKIP_2A
#define BLU_X_SKIP_2A false
#endif
#ifndef BLU_X_SKIP_2C
#define BLU_X_SKIP_2C false
#endif
#ifndef BLU_X_SKIP_RUNLEN
#define BLU_X_SKIP_RUNLEN false
#endif
#ifndef BLU_X_SKIP_LEVELS
#define BLU_X_SKIP_LEVELS false
#endif
#ifndef BLU_X_SKIP_GATHER
#define BLU_X_SKIP_GATHER false
#endif
#endif


// In-kernel stamps (experiment builds only: -DBLU_EXPERIMENTS -DBLU_X_STAMPS): s_memtime at the phase boundaries of the
// stream kernel, summed per wave and written over the first records of `out` when the wave is done (scripts/stamps.py).
#if defined(BLU_EXPERIMENTS) && defined(BLU_X_STAMPS)
__device__ uint32_t g_stamps[8192 * 16];   // [wave][16]: cycles per phase, summed over the wave's tasks (read by blu_debug_stamps)
#define STAMP_DECL uint64_t st_sum[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; uint64_t st_prev = __builtin_amdgcn_s_memtime();
#define STAMP(i) { const uint64_t st_now = __builtin_amdgcn_s_memtime(); st_sum[i] += st_now - st_prev; st_prev = st_now; }
#define STAMP_DRAIN asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#else
#define STAMP_DECL
#define STAMP(i)
#define STAMP_DRAIN
#endif

#define WAVE 64
#ifndef BLOCK_A
#define BLOCK_A 768   // 12 waves per CU (three per SIMD): what the per-wave LDS (ring 8 KiB + list) leaves room for
#endif
#define WAVES_A (BLOCK_A / WAVE)
#ifndef BLOCK_N
#define BLOCK_N 1024  // the kernel without the ring: 16 waves per CU (four per SIMD, 128 VGPRs)
#endif
#ifndef BLU_N_WAVES_PER_SIMD
#define BLU_N_WAVES_PER_SIMD 4
#endif
#ifndef LIST_CAP
#define LIST_CAP 208       // top-group entries per wave task (64 queries; mean ~183, sigma ~18 at geometric(0.35) groups)
#endif
#ifndef LIST_CAP_F64
#define LIST_CAP_F64 232   // the same in the f64 layouts (4 more bytes per entry) with the ring: 11 waves per CU (BLOCK_F) so that a C3 task's top
                           // rows (mean 183) fit one round — at 12 waves the list held 160 entries and most tasks took two
#endif
#ifndef BLOCK_F
#define BLOCK_F 704        // f64 layouts, kernel with the ring
#endif

// ---- cross-lane helpers -----------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

template <int CTRL>
__device__ __forceinline__ int dpp(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }

// DPP controls: quad_perm [1,0,3,2] = 0xB1, quad_perm [2,3,0,1] = 0x4E, row_ror:4 = 0x124, row_ror:8 = 0x128
#define ROW_REDUCE(x, OP)            \
    x = OP(x, dpp<0xB1>(x));         \
    x = OP(x, dpp<0x4E>(x));         \
    x = OP(x, dpp<0x124>(x));        \
    x = OP(x, dpp<0x128>(x));

__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t umin(uint32_t a, uint32_t b) { return a < b ? a : b; }
__device__ __forceinline__ int rl(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

__device__ __forceinline__ int wave_max_i32(int x) {
    ROW_REDUCE(x, imax)
    return imax(imax(rl(x, 0), rl(x, 16)), imax(rl(x, 32), rl(x, 48)));
}
__device__ __forceinline__ int wave_min_i32(int x) {
    ROW_REDUCE(x, imin)
    return imin(imin(rl(x, 0), rl(x, 16)), imin(rl(x, 32), rl(x, 48)));
}
// Layouts (bit = LAYOUT) whose phase 2c reads the node id of the reported level back from the reference row's line — an L2
// hit behind the eight loads that brought it — instead of keeping the row's 20 node ids in registers through the level
// tests.  Measured, one box per pair: f64 side records 1.129 -> 1.031 ms (their build had 20 B/lane of scratch, none with
// this), milli-percent columns 1.608 -> 1.437 (24 -> 8 B), f64 columns 1.410 -> 1.425 and the packed layout 0.953 -> 0.968
// (no scratch either way: the extra round trip shows) — hence layout by layout: 1 and 3.
// ... and every build of the kernel without the ring (128 VGPRs: its packed build goes from 60 to 40 B/lane of scratch;
// uniform 1..200-row segments 0.947 -> 0.888 ms, C5 0.450 -> 0.441)
// Where the row is not read back: levels whose node ids are kept in registers through phase 2c (the pick of the reported
// level's id is a compare/select pair per kept level); a deeper reported level is read back.  12 instead of all 20 a
// 128-byte row holds: C3 (8 levels) -1.0 %, zymo-like -2.1 %, C4 slice -1.3 % on one box (9: -0.4 / -2.1 / -1.3).
#ifndef BLU_NID_REGS
#define BLU_NID_REGS 12
#endif
#ifndef BLU_NODE_RELOAD_CAUTIOUS
#define BLU_NODE_RELOAD_CAUTIOUS 1   // ... and every cautious build (12 B/lane of scratch in its packed ring build: C3 cautious 0.988 -> 0.921 ms, what relaxed takes)
#endif
#ifndef BLU_NODE_RELOAD_NORING
#define BLU_NODE_RELOAD_NORING 1
#endif
#ifndef BLU_NODE_RELOAD_LAYOUTS
#define BLU_NODE_RELOAD_LAYOUTS 0xAu
#endif
#ifndef BLU_FLAT_ALWAYS
#define BLU_FLAT_ALWAYS 0   // experiment: every round of the kernel without the ring takes the flat pass
#endif
#ifndef BLU_FLAT_PASS
#define BLU_FLAT_PASS 1
#endif
#ifndef FLAT_ROWS
#define FLAT_ROWS 8u             // rows per lane of a flat step (two 16-byte loads)
#endif
#define FLAT_SEG (64u * FLAT_ROWS)   // longest segment the flat pass takes (64 lane units)
#ifndef FLAT_STEPS
#define FLAT_STEPS 8u            // steps (of 64 units) a flat round takes at most: two registers per step and lane
#endif
#ifndef FLAT_DEPTH
#define FLAT_DEPTH 4u            // steps whose bit-scores are in flight together
#endif
__device__ __forceinline__ int iadd(int a, int b) { return a + b; }
__device__ __forceinline__ int wave_sum_u32(uint32_t v) {
    int x = (int)v;
    ROW_REDUCE(x, iadd)
    return rl(x, 0) + rl(x, 16) + rl(x, 32) + rl(x, 48);
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) { return (uint32_t)wave_max_i32((int)(v ^ 0x80000000u)) ^ 0x80000000u; }
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) { return (uint32_t)wave_min_i32((int)(v ^ 0x80000000u)) ^ 0x80000000u; }

// inclusive prefix sum over the 64 lanes
__device__ __forceinline__ uint32_t wave_incl_scan_u32(uint32_t v) {
    uint32_t incl = v;
    incl += (uint32_t)dpp<0x111>((int)incl);
    incl += (uint32_t)dpp<0x112>((int)incl);
    incl += (uint32_t)dpp<0x114>((int)incl);
    incl += (uint32_t)dpp<0x118>((int)incl);
    const uint32_t t0 = (uint32_t)rl((int)incl, 15), t1 = (uint32_t)rl((int)incl, 31), t2 = (uint32_t)rl((int)incl, 47);
    const uint32_t r16 = (uint32_t)lane_id() >> 4;
    return incl + (r16 == 0 ? 0u : (r16 == 1 ? t0 : (r16 == 2 ? t0 + t1 : t0 + t1 + t2)));
}
// The worklist is WL_QUEUES queues, each with its own counter on a memory line of its own: a task appends to queue
// (task % WL_QUEUES).  (One counter for all: the 15 600 tasks of C5 each append once per kernel, and 4096 waves adding to ONE
// address get about one atomic per 11 ns out of the memory side — 175 us of a 260-us kernel went into waiting for that
// counter, whatever the rest of the kernel did.)  Queue s: entries worklist[s * cap ..], cap = the queries of the tasks that
// append to it; live counter work_count[WL_BASE + s * WL_STRIDE], the length published for the worklist kernel one word on.
#define WL_QUEUES 64u
#define WL_BASE 64u
#define WL_STRIDE 32u
__device__ __forceinline__ uint32_t wl_capacity(uint64_t n_queries) {
    const uint64_t n_tasks = (n_queries + WAVE - 1) / WAVE;
    return (uint32_t)((n_tasks + WL_QUEUES - 1) / WL_QUEUES) * WAVE;
}
// A query goes to its task's queue: the slot from the queue's counter (agent-scope atomic), the entry as a write-through
// (sc1) store — the entries may be read by the last block of the SAME kernel (its in-kernel drain, no kernel boundary in between),
// which loads them with sc1 loads (wl_entry) after every wave has waited for its own stores (end of the stream kernel).
__device__ __forceinline__ void wl_push(uint32_t* const wl_q, uint32_t* const wl_cnt, const uint32_t q) {
    __hip_atomic_store(wl_q + atomicAdd(wl_cnt, 1u), q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// entry wi of the queues taken one after the other (incl: inclusive prefix sums of their lengths, lane = queue)
__device__ __forceinline__ uint32_t wl_entry(const uint32_t* __restrict__ worklist, const uint32_t cap, const uint32_t incl, const uint32_t wi) {
    const uint32_t sub = (uint32_t)__builtin_popcountll(__ballot(incl <= wi));
    const uint32_t before = sub ? (uint32_t)rl((int)incl, (int)sub - 1) : 0u;
    return __hip_atomic_load(worklist + (uint64_t)sub * cap + (wi - before), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    return __hiloint2double(dpp<CTRL>(__double2hiint(v)), dpp<CTRL>(__double2loint(v)));
}
__device__ __forceinline__ double rl_f64(double v, int lane) {
    return __hiloint2double(rl(__double2hiint(v), lane), rl(__double2loint(v), lane));
}
// plain compare-select (callers exclude NaN)
__device__ __forceinline__ double dmax(double a, double b) { return b > a ? b : a; }
__device__ __forceinline__ double dmin(double a, double b) { return b < a ? b : a; }
__device__ __forceinline__ double wave_max_f64(double x) {
    x = dmax(x, dpp_f64<0xB1>(x)); x = dmax(x, dpp_f64<0x4E>(x));
    x = dmax(x, dpp_f64<0x124>(x)); x = dmax(x, dpp_f64<0x128>(x));
    return dmax(dmax(rl_f64(x, 0), rl_f64(x, 16)), dmax(rl_f64(x, 32), rl_f64(x, 48)));
}
__device__ __forceinline__ double wave_min_f64(double x) {
    x = dmin(x, dpp_f64<0xB1>(x)); x = dmin(x, dpp_f64<0x4E>(x));
    x = dmin(x, dpp_f64<0x124>(x)); x = dmin(x, dpp_f64<0x128>(x));
    return dmin(dmin(rl_f64(x, 0), rl_f64(x, 16)), dmin(rl_f64(x, 32), rl_f64(x, 48)));
}
__device__ __forceinline__ int first_lane(uint64_t m) { return __builtin_ctzll(m); }
__device__ __forceinline__ int last_lane(uint64_t m) { return 63 - __builtin_clzll(m); }
__device__ __forceinline__ uint64_t rl_u64(uint64_t v, int lane) {
    return (uint64_t)(uint32_t)rl((int)(uint32_t)v, lane) | ((uint64_t)(uint32_t)rl((int)(uint32_t)(v >> 32), lane) << 32);
}
__device__ __forceinline__ uint64_t uniform64(uint64_t v) {
    // (the builtin returns int: without the casts a low word with bit 31 set sign-extends into the high word)
    return (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)v) |
           ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(v >> 32)) << 32);
}

// ---- record store -------------------------------------------------------------
__device__ __forceinline__ void store_result(blu_result* out, uint64_t q, uint32_t status, uint32_t flags,
                                             uint32_t bean_index, uint32_t mar_level, uint32_t reached_rank,
                                             uint32_t mar_code, uint32_t identifier, uint32_t ref_row,
                                             uint64_t level_mask, double ident) {
    uint4 a;
    a.x = (status & 0xFF) | ((flags & 0xFF) << 8) | ((bean_index & 0xFF) << 16) | ((mar_level & 0xFF) << 24);
    a.y = (reached_rank & 0xFFFF) | ((mar_code & 0xFFFF) << 16);
    a.z = identifier;
    a.w = ref_row;
    uint4 b;
    b.x = (uint32_t)level_mask;
    b.y = (uint32_t)(level_mask >> 32);
    b.z = (uint32_t)__double2loint(ident);
    b.w = (uint32_t)__double2hiint(ident);
    uint4* p = reinterpret_cast<uint4*>(out + q);
    p[0] = a;
    p[1] = b;
}
__device__ __forceinline__ void pack_result(uint4& a, uint4& b, uint32_t status, uint32_t flags, uint32_t bean_index,
                                            uint32_t mar_level, uint32_t reached_rank, uint32_t mar_code, uint32_t identifier,
                                            uint32_t ref_row, uint64_t level_mask, double ident) {
    a.x = (status & 0xFF) | ((flags & 0xFF) << 8) | ((bean_index & 0xFF) << 16) | ((mar_level & 0xFF) << 24);
    a.y = (reached_rank & 0xFFFF) | ((mar_code & 0xFFFF) << 16);
    a.z = identifier;
    a.w = ref_row;
    b.x = (uint32_t)level_mask;
    b.y = (uint32_t)(level_mask >> 32);
    b.z = (uint32_t)__double2loint(ident);
    b.w = (uint32_t)__double2hiint(ident);
}
__device__ __forceinline__ void pack_status(uint4& a, uint4& b, uint32_t status, uint32_t ref_row) {
    pack_result(a, b, status, 0, BLU_NONE_U8, BLU_NONE_U8, BLU_NONE_U16, BLU_NONE_U16, 0xFFFFFFFFu, ref_row, 0ull, 0.0);
}
__device__ __forceinline__ void store_status(blu_result* out, uint64_t q, uint32_t status, uint32_t ref_row) {
    store_result(out, q, status, 0, BLU_NONE_U8, BLU_NONE_U8, BLU_NONE_U16, BLU_NONE_U16, 0xFFFFFFFFu, ref_row, 0ull, 0.0);
}

// perc_identity from its milli-percent encoding: the correctly rounded k / 1000, i.e. the double the reference's text
// parser yields for a value printed with at most three decimals (IEEE f64 division; this file is built without
// fast-math)
__device__ __forceinline__ double milli_to_f64(uint32_t k) { return (double)k / 1000.0; }
// The same for k < 2^17 (every value the packed layouts and the keyed f64 rounds hold) without the division: with y = fl(1/1000),
// q = fl(k y), r = fma(-q, 1000, k) (exact) and fma(r, y, q) is the correctly rounded k / 1000 for EVERY k in [0, 131072) —
// checked exhaustively against the exact quotient on the host (rational arithmetic) and on the device against the division
// (tests/test_gpu_milli_exact.py); three f64 operations instead of the dozen of an IEEE division.
__device__ __forceinline__ double milli17_to_f64(uint32_t k) {
    const double x = (double)k, y = 0.001;
    const double q = x * y;
    const double r = __builtin_fma(-q, 1000.0, x);
    return __builtin_fma(r, y, q);
}

// packed level word of a lineage row -> the ABI's 16-bit rank codes
__device__ __forceinline__ uint32_t packed_rank(uint32_t p) { return (p >> BLU_PACK_CUT_BITS) & BLU_PACK_CODE_MASK; }
__device__ __forceinline__ uint32_t packed_mar(uint32_t p) {
    const uint32_t m = p >> (BLU_PACK_CUT_BITS + BLU_PACK_CODE_BITS);
    return m == BLU_PACK_NEVER ? BLU_MAR_NEVER_EQUAL : m;
}

// Stable sort by (len, pident, align_len, accession), then .first() (Cautious) or
// .last() (Relaxed): find_multi_taxa_consensus.rs:39-68.  Candidates arrive in file
// order, so Relaxed replaces the incumbent on ties (>=), Cautious keeps it (<).
// PK = double, or the milli-percent integer itself: k -> fl(k / 1000.0) is strictly increasing, so the integer
// compares order the hits exactly as the reference's f64 compares do (and there is no NaN in that layout).
template <int STRAT, typename PK>
__device__ __forceinline__ bool key_better(uint32_t len, PK pid, int aln, uint32_t acc, uint32_t blen, PK bpid,
                                           int baln, uint32_t bacc) {
    const bool gt = (len > blen) | ((len == blen) & ((pid > bpid) | ((pid == bpid) & ((aln > baln) | ((aln == baln) & (acc > bacc))))));
    const bool eq = (len == blen) & (pid == bpid) & (aln == baln) & (acc == bacc);
    if (STRAT == BLU_RELAXED) return gt | eq;
    return !(gt | eq);
}

// ---- shared levels of a group of rows: range minimum over the adjacent-row LCP array -------------------
typedef unsigned short us2 __attribute__((ext_vector_type(2)));

// OR-mask that forces the bytes of one 32-bit word outside [s, e) (byte indices relative to the word) to 0xFF
__device__ __forceinline__ uint32_t exclude_mask(int s, int e) {
    const int a = s < 0 ? 0 : (s > 4 ? 4 : s);
    const int b = e < 0 ? 0 : (e > 4 ? 4 : e);
    const uint64_t low = (1ull << (8 * a)) - 1ull;
    const uint64_t high = ~((1ull << (8 * b)) - 1ull);
    return (uint32_t)(low | high);
}
// min(lcp8[lo .. hi-1]) for lo < hi: number of leading levels shared by every row with lo <= pos <= hi.
// Branch-free: the (at most) two partial 16-entry blocks and the two sparse-table bytes are requested together and
// combined afterwards, so a lookup is ONE memory round trip, not up to three one after the other.
__device__ __forceinline__ uint32_t shared_levels(const TaxDev& t, uint32_t lo, uint32_t hi) {
    const uint32_t b0 = (lo + 15) >> 4, b1 = hi >> 4;          // whole blocks b0 .. b1 - 1 lie inside [lo, hi)
    const bool one = b0 > b1;                                  // lo and hi inside one block
    const uint32_t blk_a = lo >> 4;
    const int s_a = (int)(lo & 15), e_a = one ? (int)(hi - (blk_a << 4)) : 16;      // head: entries [s_a, e_a) of block blk_a
    const int e_b = one ? 0 : (int)(hi & 15);                                       // tail: entries [0, e_b) of block b1
    const bool mid = b0 < b1;
    const uint32_t k = mid ? 31 - __builtin_clz(b1 - b0) : 0;
    const uint8_t* lvl = t.rmq + (uint64_t)k * t.rmq_nb;
    const uint4 wa = *reinterpret_cast<const uint4*>(t.lcp8 + (uint64_t)blk_a * 16);
    const uint4 wb = *reinterpret_cast<const uint4*>(t.lcp8 + (uint64_t)b1 * 16);
    const uint32_t r0 = lvl[mid ? b0 : 0u], r1 = lvl[mid ? b1 - (1u << k) : 0u];
    auto bmin = [](const uint4 w, const int s, const int e) {  // min of bytes [s, e) of a 16-byte block (0xFF if empty)
        const uint32_t x[4] = {w.x | exclude_mask(s, e), w.y | exclude_mask(s - 4, e - 4), w.z | exclude_mask(s - 8, e - 8),
                               w.w | exclude_mask(s - 12, e - 12)};
        us2 acc = {0xFF, 0xFF};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t a = x[i] & 0x00FF00FFu, b = (x[i] >> 8) & 0x00FF00FFu;
            acc = __builtin_elementwise_min(acc, __builtin_elementwise_min(__builtin_bit_cast(us2, a), __builtin_bit_cast(us2, b)));
        }
        return (uint32_t)(acc.x < acc.y ? acc.x : acc.y);
    };
    uint32_t m = bmin(wa, s_a, (s_a != 0 || one) ? e_a : s_a);   // (a head starting at entry 0 of a whole block belongs to the middle part)
    m = umin(m, bmin(wb, 0, e_b));
    m = umin(m, mid ? umin(r0, r1) : 0xFFu);
    return m;
}

// Shared levels of a WIDE group from the wide-node tables (TaxDev::wblk / wchain): the deepest wide node around `lo` out of
// its 64-row block's entry, then the number of its ancestors-or-self — one per level from 0 — whose run still holds `hi`.
// wide_node(): 0 = none, 0xFFFFFFFF = the block has more changes than its entry holds (ask the range-minimum tables).
__device__ __forceinline__ uint32_t wide_node(const uint2 e, const uint32_t lo) {
    const uint32_t o = lo & ((1u << BLU_WBLK_SHIFT) - 1u), s1 = (e.y >> 16) & 0xFFu, s2 = e.y >> 24;
    const uint32_t w = o >= s2 ? (e.y & 0xFFFFu) : (o >= s1 ? e.x >> 16 : e.x & 0xFFFFu);
    return s1 == BLU_WBLK_OVERFLOW ? 0xFFFFFFFFu : w;
}
__device__ __forceinline__ uint32_t chain_count4(const uint4 c, const uint32_t hi) {
    return (uint32_t)(c.x > hi) + (uint32_t)(c.y > hi) + (uint32_t)(c.z > hi) + (uint32_t)(c.w > hi);
}
// entries of a chain requested together with the reference row (levels 0 .. 11: three 16-byte loads, twelve registers in
// flight — the fourth group and wchain_hi are asked for only by a wave in which some chain is that deep and still holds hi)
#define BLU_WCHAIN_FIRST 12u

// Which of a lane's N rows tie on the query's top score, row 0 in the top bit of the N-bit mask: mask = 2 mask + (b == M),
// four rows per group — four compares into scalar pairs, then four add-with-carry (the compare's lane mask is the carry in).
// Two vector instructions per row; as C++ ((mask << 1) | (b == M)) hipcc emits compare, select, or, shift — 3.5 per row — and the
// scan is a third of the stream kernel's vector instructions.  (gfx950: a VALU read of a scalar pair needs two wait states
// after the VALU write: the three instructions between a compare and its add cover them.)
#ifndef BLU_ADDC_MASK
#define BLU_ADDC_MASK 1
#endif
template <uint32_t N>
__device__ __forceinline__ uint32_t tie_mask(const int (&b)[N], const int M) {
    static_assert(N % 8u == 0u, "two half masks, four rows per group each");
#if BLU_ADDC_MASK
    // two accumulators (rows 0 .. N/2 - 1 and N/2 .. N - 1) so that consecutive adds do not depend on each other
    constexpr uint32_t H = N / 2u;
    uint32_t ma = 0, mb = 0;
#pragma unroll
    for (uint32_t i = 0; i < H; i += 4) {
        uint64_t s0, s1, s2, s3, s4, s5, s6, s7, j;
        asm("v_cmp_eq_u32_e64 %[s0], %[a0], %[M]\n\t"
            "v_cmp_eq_u32_e64 %[s4], %[b0], %[M]\n\t"
            "v_cmp_eq_u32_e64 %[s1], %[a1], %[M]\n\t"
            "v_cmp_eq_u32_e64 %[s5], %[b1], %[M]\n\t"
            "v_cmp_eq_u32_e64 %[s2], %[a2], %[M]\n\t"
            "v_cmp_eq_u32_e64 %[s6], %[b2], %[M]\n\t"
            "v_cmp_eq_u32_e64 %[s3], %[a3], %[M]\n\t"
            "v_cmp_eq_u32_e64 %[s7], %[b3], %[M]\n\t"
            "v_addc_co_u32_e64 %[ma], %[j], %[ma], %[ma], %[s0]\n\t"
            "v_addc_co_u32_e64 %[mb], %[j], %[mb], %[mb], %[s4]\n\t"
            "v_addc_co_u32_e64 %[ma], %[j], %[ma], %[ma], %[s1]\n\t"
            "v_addc_co_u32_e64 %[mb], %[j], %[mb], %[mb], %[s5]\n\t"
            "v_addc_co_u32_e64 %[ma], %[j], %[ma], %[ma], %[s2]\n\t"
            "v_addc_co_u32_e64 %[mb], %[j], %[mb], %[mb], %[s6]\n\t"
            "v_addc_co_u32_e64 %[ma], %[j], %[ma], %[ma], %[s3]\n\t"
            "v_addc_co_u32_e64 %[mb], %[j], %[mb], %[mb], %[s7]"
            : [ma] "+v"(ma), [mb] "+v"(mb), [s0] "=&s"(s0), [s1] "=&s"(s1), [s2] "=&s"(s2), [s3] "=&s"(s3), [s4] "=&s"(s4), [s5] "=&s"(s5),
              [s6] "=&s"(s6), [s7] "=&s"(s7), [j] "=&s"(j)
            : [a0] "v"(b[i]), [a1] "v"(b[i + 1]), [a2] "v"(b[i + 2]), [a3] "v"(b[i + 3]), [b0] "v"(b[H + i]), [b1] "v"(b[H + i + 1]),
              [b2] "v"(b[H + i + 2]), [b3] "v"(b[H + i + 3]), [M] "v"(M));
    }
    return (ma << H) | mb;
#else
    uint32_t mask = 0;
#pragma unroll
    for (uint32_t i = 0; i < N; ++i) mask = (mask << 1) | (uint32_t)(b[i] == M);
    return mask;
#endif
}

// ===============================================================================
// Kernel A
// ===============================================================================
struct HitsDev;
template <int STRAT, int LAYOUT>
__device__ __forceinline__ void consensus_of_long_query(const HitsDev& h, const TaxDev& t, blu_result* __restrict__ out, const uint64_t q,
                                                        uint32_t* const slot, const int lane);   // (defined with kernel B)
template <bool PID32> struct PidKey { typedef double type; };
template <> struct PidKey<true> { typedef uint32_t type; };
template <bool PID32>
__device__ __forceinline__ double pid_f64(typename PidKey<PID32>::type k) {
    if constexpr (PID32) return milli_to_f64(k);
    else return k;
}

#define META_SLOW 0x80000000u
#define META_DENSE 0x40000000u   // the query was reduced by a dense step of phase 1: nothing of it in the list
#define KEY_PID_BITS 17u         // milli-percent perc_identity below 2^17 packs with the lineage length into one sort word
// Comparison-ready list entries (gather_list -> phase 2a): .x = sorted position | high 7 bits of the shape hint << 25,
// .y = lineage length << 25 | milli-percent << 8 | low 8 bits of the shape hint (masked out of the compares),
// .z = align_length biased to compare unsigned, .w = accession rank
#define KEYED_LEN_SHIFT 25u
#define KEYED_PID_SHIFT 8u
#define PM_MASK ((1u << KEY_PID_BITS) - 1u)
#ifndef SHORT_SEG
#define SHORT_SEG 128u           // segments up to here are streamed (4 .. 32 lanes per query); longer ones take the sparse long pass
#endif
#ifndef BLU_LONG_COST
#define BLU_LONG_COST 96u   // lane-steps one query costs in the long pass (half a 64-lane step, not pipelined; 64 / 96 / 128 / 192 measured)
#endif
#ifndef BLU_DENSE_Q
#define BLU_DENSE_Q 3u      // a dense step reads every record of its rows (four lanes per scanning lane) when at least 1 / BLU_DENSE_Q of them are top rows
#endif
#ifndef MAX_TASK_SEG
#define MAX_TASK_SEG 512u        // longest segment the stream kernel takes (64 lanes x 4 rows, twice); longer ones go to the worklist
#endif
#ifndef BLOCK_B
#define BLOCK_B 256
#endif
#define LONG_SPAN (1u << 28)    // rows one bit-score descriptor of the worklist kernel covers
#ifndef BLU_B_WAVES_PER_SIMD
#define BLU_B_WAVES_PER_SIMD 7   // worklist kernel: 72 VGPRs, 28 waves per CU (it hides memory round trips with waves, not with registers).
                                 // At 8 (64 VGPRs) every column-layout and cautious build parked 8-20 B per lane in scratch: the same time on
                                 // C5 relaxed, 12 % slower on C5 cautious (0.471 -> 0.413 ms, scripts/calls/r4_call30.sh)
#endif
#define KEEP_ROWS 1024u         // worklist kernel: a segment of up to this many rows is held in registers (4 x 16 bytes per lane)
#define SLOT_CAP 256u           // worklist kernel: rows of a top group collected before their side records are gathered
#define TASK_SPAN (1ull << 27)   // rows one task's buffer descriptors cover
#define CUT_LDS 512        // distinct cutoff values kept in LDS (4 KiB); larger tables are read from global memory
#define ROW_MASK ((1u << BLU_ROW_BITS) - 1u)
// the five hit columns are read exactly once per run: non-temporal loads keep them from displacing the
// lineage rows and cutoff tables (re-read by every query) in L2 / Infinity Cache
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// Plain (not nt) loads for the five columns: a 128-byte line is shared by consecutive queries (50 hits = 200 B per
// 4-byte column), and with nt the line is dropped before the wave's next step needs its other half — measured
// +12 % HBM read requests (TCC_EA0_RDREQ) and +10 % time.
#define STREAM_AUX 0
#ifndef RING_DMA_MOD
#define RING_DMA_MOD " sc1 nt"   // cache-policy bits of the ring's DMA requests: every bit-score is read exactly once, so the stream
                                 // should not displace the lineage rows (re-read ~4 times per run) from L2 / Infinity Cache: -6 % on C3
#endif
#ifndef GATHER_AUX
#define GATHER_AUX 2         // cache-policy bits of the gathered side records (2 = nt: read once as well)
#endif
#ifndef RECORD_AUX
#define RECORD_AUX 18  // sc1 | nt
#endif
#ifndef BLU_REF_NT
#define BLU_REF_NT 0   // reference-row loads non-temporal (experiment)
#endif
#ifndef BLU_LANE_LAUNDER_LONG
#define BLU_LANE_LAUNDER_LONG 0   // (the worklist kernel at 64 registers: laundering its lane id moved scratch from 8 to 28 B/lane in the packed build — off)
#endif
#ifndef BLU_PRIO_DENSE
#define BLU_PRIO_DENSE 1   // dense steps of phase 1 (side records fetched and reduced by the scanning lanes) at raised wave priority: zymo-like
                           // 1.217 -> 1.199 ms, all 50 hits tied 0.548 -> 0.528, C3 0.9554 -> 0.9527 (one box)
#endif
#ifndef BLU_PRIO_LONG
#define BLU_PRIO_LONG 1   // the worklist kernel's finalisation (side-record gather, reference row, record) at raised wave priority: C5 0.4149 -> 0.4110 ms
#endif
#ifndef BLU_PRIO
#define BLU_PRIO 4   // wave priority by phase (s_setprio): 0 = none; 4 = level 3 from the request of a task's side records to the request of
                     // its reference rows — the stretch in which the wave is about to start its next memory round trip, and must not queue
                     // behind other waves' scan arithmetic, whose data is prefetched anyway — then level 1 for the finalisation, 0 for the
                     // scan.  One box, medians of seven: C3 0.9074 -> 0.8836 ms, zymo-like 1.179 -> 1.146, 10 hits per query 1.390 -> 1.335,
                     // f64 side records 1.043 -> 1.021; modes 1 (level 2, back to 0), 2 (level 3), 3 (level 2 to the end of the task) and
                     // 5 (3 / 2 / 0) within 0.5 % of it
#endif
#ifndef BLU_WIDE_RMQ
#define BLU_WIDE_RMQ 0   // 1: wide groups ask the range-minimum tables as before round 4 (A/B and a test of the fallback)
#endif

// The bit-score stream of a task whose segments are all streamed goes through a per-wave LDS ring, filled by LDS-DMA
// (buffer_load_dwordx4 ... lds: 1 KiB = 256 rows per wave instruction, no VGPR destination) well ahead of the steps that
// read it, across task boundaries: a wave no longer pays a memory round trip per step.
#ifndef BLU_MIXED_RING
#define BLU_MIXED_RING 0
#endif
#ifndef BLU_MIXED_MAX_LPQ
#define BLU_MIXED_MAX_LPQ 16u    // most lanes per query a task of mixed lengths may take to make its steps fit the ring
#endif
#ifndef RING_ROWS
#define RING_ROWS 2048u          // power of two, multiple of 256: the chunks of one step (up to 1792 rows + alignment slack) / what is requested ahead
#endif
#define RING_PAD 32u              // >= rows one lane scans in a step
#define RING_CHUNKS (RING_ROWS / 256u)
#define RING_MASK (RING_ROWS - 1u)
static_assert((RING_ROWS & RING_MASK) == 0 && RING_ROWS >= 1024u, "ring size");
// A lane of a ring step scans RPL = 16 or 32 consecutive rows (16 for tasks of short segments, 32 otherwise: half as many
// steps per task).  Lane descriptor in the list: top-row mask | first row (13 bits) | position / RPL (3 bits) — one word with
// the mask (top-aligned) is the first word of the list slot, the rest the second (the first row can be far into a task that
// holds long segments).
#define DESC_SUB_BITS 8u
#define DESC_WORD1(RPL, row0, sub) (((row0) << DESC_SUB_BITS) | (sub))   // first row relative to the task (13 bits) | its position in the segment (< 128)
static_assert(LIST_CAP >= 128 && LIST_CAP_F64 >= 128, "the list area also stages the 64 records of a task");

// the flat pass of the kernel without the ring (tasks of mixed segment lengths): per query its first 4-row quad in the round's
// quad numbering, its top bit-score and its top-row count
struct FlatLds { uint32_t qs[WAVE + 4]; int32_t qm[WAVE]; uint32_t qk[WAVE]; };
struct NoFlatLds {};
template <bool F64, bool RING>
struct WaveLds : std::conditional_t<RING, NoFlatLds, FlatLds> {
    static constexpr uint32_t CAP = F64 ? LIST_CAP_F64 : LIST_CAP;
    alignas(16) uint32_t ring[RING ? RING_ROWS + RING_PAD : 4u];   // bit-scores: row v sits at ring[v & RING_MASK]; the pad mirrors ring[0 .. RING_PAD) so that a lane's run of rows never wraps
    // top-group rows of the task's queries in file order: {engine row id (sorted position | length << BLU_ROW_BITS),
    // perc_identity (milli-percent, or the low f64 word), align_length, accession rank} — a side record of the packed
    // layout as it is.  Between phase 1 and the gather of a ring task .x holds the row (relative to the task's first row).
    uint4 rec[CAP];
    uint32_t p1[F64 ? CAP : 1];             // high f64 word of perc_identity (f64 layout only)
    uint16_t pq[CAP];                       // position of the row in its segment
    uint32_t meta[WAVE + 4];    // first entry | k << 16, or META_SLOW
    uint2 seg[WAVE + 4];        // {first row relative to the task's first row, row count (0 if > MAX_TASK_SEG or outside the span)}
    uint32_t vx[WAVE + 4];      // first row in the task's ring numbering (whole chunks inside longer segments are left out), [nq] = its end
};

// waits until at most k of the wave's vector-memory operations are outstanding (k is wave-uniform; s_waitcnt takes an immediate)
__device__ __forceinline__ void wait_vmcnt(uint32_t k) {
    switch (k) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
        case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
        case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    }
}
static_assert(RING_CHUNKS <= 16, "wait_vmcnt covers 0..15 younger chunks");

#ifndef BLU_STEP_SETS
#define BLU_STEP_SETS 2
#endif
#ifndef BLU_WAVES_PER_SIMD
#define BLU_WAVES_PER_SIMD 3   // one 768-thread block per CU: 168 VGPRs
#endif
// LAYOUT: 0 = perc_identity f64 column, 1 = milli-percent u32 column, 2 = packed 16-byte side records
// {tax_row, pident_milli | shape hint << 17, align_len, acc_rank} next to the bit-score column, 3 = 24-byte side records
// {tax_row, shape hint << 17, align_len, acc_rank, perc_identity f64}
// RING = false: the same kernel without the bit-score ring — every task takes the direct-load phase 1 — at 128 VGPRs and
// 16 waves per CU instead of 168 and 12.  A table whose tasks are mostly of mixed segment lengths (none of them could use
// the ring) runs on that one: its phase 1 is bound by instruction issue, which a fourth wave per SIMD helps and the
// ring's registers and LDS do not (C5: 0.626 -> 0.573 ms).  blu_classify_tasks decides per run (work_count[9]); the
// kernel of the other kind returns at once.
template <int STRAT, int LAYOUT, bool RING>
__global__ __launch_bounds__((LAYOUT == 0 || LAYOUT == 3) ? (RING ? BLOCK_F : BLOCK_A) : (RING ? BLOCK_A : BLOCK_N), (RING || LAYOUT == 0 || LAYOUT == 3) ? BLU_WAVES_PER_SIMD : BLU_N_WAVES_PER_SIMD)
void blu_consensus_stream_kernel(HitsDev h, TaxDev t, blu_result* __restrict__ out, uint32_t* __restrict__ worklist, uint32_t* __restrict__ work_count,
                                 uint32_t mode, uint32_t* __restrict__ host_len) {
    // mode bit 0 ("forced"): the host launched this kind alone; bit 1 ("no worklist kernel"): no launch of kernel B follows this
    // one — the last run on this table left (next to) nothing in the queue — so whatever is queued after all is drained by the
    // last block of this kernel to finish (slow if it is much, never wrong); host_len: pinned host word the queue length goes to
    const uint32_t forced = mode & 1u;
    const bool no_long = (mode & 2u) != 0u;
    // PACKED: 16-byte side records (milli-percent), WIDE: 24-byte side records (f64 perc_identity in words 4, 5)
    constexpr bool PID32 = LAYOUT == 1 || LAYOUT == 2, PACKED = LAYOUT == 2, WIDE = LAYOUT == 3;
    constexpr uint32_t BLOCK_T = !PID32 ? (RING ? BLOCK_F : BLOCK_A) : (RING ? BLOCK_A : BLOCK_N), WAVES_T = BLOCK_T / WAVE;   // (the f64 layouts do not fit 128 VGPRs)
    constexpr uint32_t CAP = WaveLds<!PID32, RING>::CAP;   // list entries per wave task
    // forced: the host launched this kind alone (it remembered the kind of the handle's last table); else both kinds are
    // in the stream and the one blu_classify_tasks did not pick returns
    if (!forced && __hip_atomic_load(work_count + 9, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (RING ? 1u : 2u)) return;
    // work_count = {queue length, blocks done, published length}: the first two are zero on entry and on exit
    __shared__ WaveLds<!PID32, RING> s_lds[WAVES_T];
    // the distinct cutoff values of this (taxonomy, backbone): a few hundred doubles, read per level in phase 2c
    __shared__ double s_cut[CUT_LDS];
    const bool cut_in_lds = t.n_cutvals <= CUT_LDS;
    if (cut_in_lds) {
        for (uint32_t i = threadIdx.x; i < t.n_cutvals; i += BLOCK_T) s_cut[i] = t.cutvals[i];
        __syncthreads();
    }
    int lane = lane_id();   // (not const: see BLU_LANE_LAUNDER at the head of the task loop)
    const int wib = __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
    WaveLds<!PID32, RING>& L = s_lds[wib];
    const uint32_t wl_cap = wl_capacity(h.n_queries);
    // (task arithmetic in 32 bits: the ABI keeps n_queries below 2^32 — worklist entries are 32-bit query ids — and every
    // kernel-lifetime scalar of this kernel costs a register it does not have)
    const uint32_t n_q32 = (uint32_t)h.n_queries;
    const uint32_t wave = blockIdx.x * WAVES_T + (uint32_t)wib;
    const uint32_t n_waves = gridDim.x * WAVES_T;
    // Tasks: 64 consecutive queries each while there is a whole round of them (one task per wave of the grid); the queries left
    // after the last whole round — up to a round's worth — are dealt to ALL waves in equal pieces of `tail_q` queries (a
    // multiple of 8, at least 16) instead of 64-query tasks for some waves and nothing for the others: the kernel ends when
    // the slowest wave does, and a piece costs less than a task.  (C4 slice, 1.25 M queries: 6 whole rounds + 1099 tasks
    // became 6 rounds + 3072 pieces of 24 queries; C2, 100 k queries, less than one round: 1563 tasks on 131 CUs became 2500
    // pieces of 40 on all CUs.)
#ifndef BLU_TAIL_SPLIT
#define BLU_TAIL_SPLIT 1
#endif
    // (kept to three kernel-lifetime scalars — the whole rounds' task count and this wave's own piece: every further one costs
    // a register the ring build does not have)
    uint32_t t_full, tail_q0, tail_nq;
    {
        const uint32_t n_tasks64 = n_q32 / WAVE + ((n_q32 % WAVE) != 0u ? 1u : 0u);
        t_full = n_tasks64 / n_waves * n_waves;                         // tasks of the whole rounds
        const uint32_t q_full = t_full == n_tasks64 ? n_q32 : t_full * WAVE;
        uint32_t tail_q = WAVE;                                         // (BLU_TAIL_SPLIT = 0: the tail as 64-query tasks, as before)
        if (BLU_TAIL_SPLIT) {
            tail_q = (((n_q32 - q_full) + n_waves - 1u) / n_waves + 7u) / 8u * 8u;   // (at most 64: the tail is less than a round)
            tail_q = tail_q < 16u ? 16u : (tail_q > WAVE ? WAVE : tail_q);
        }
        const uint64_t p0 = (uint64_t)q_full + (uint64_t)wave * tail_q;   // this wave's piece of the tail
        tail_q0 = p0 < n_q32 ? (uint32_t)p0 : n_q32;
        tail_nq = n_q32 - tail_q0 < tail_q ? n_q32 - tail_q0 : tail_q;
    }

    // ---- the bit-score ring of this wave (see WaveLds).  The table's rows are numbered v = row + mis, where mis (0..3)
    // makes v = 0 fall on a 16-byte boundary of the column; chunk c = rows 256 c .. 256 c + 255 lands in ring slot
    // c % RING_CHUNKS whichever task asks for it, so a task finds what the task before it prefetched.
    const uint32_t mis = (uint32_t)(((uintptr_t)h.bitscore >> 2) & 3u);
    const char* const bs_al = reinterpret_cast<const char*>(h.bitscore) - 4u * mis;
    const uint64_t v_total = h.n_hits + mis;
    uint32_t ring_head = 0, ring_landed = 0, ring_tail = 0;   // chunk ids: next to request / all before it have landed / first still needed
    uint32_t ring_c0 = 0, ring_end = 0;                        // chunk the DMA descriptor is based at / end of the task's chunks
    uint32_t pref_q0 = 0xFFFFFFFFu;                            // first query of the task whose first chunks were requested ahead
    // The DMA is issued as inline assembly (m0 = LDS address of the slot; buffer_load_dwordx4 ... lds) rather than through
    // __builtin_amdgcn_raw_ptr_buffer_load_lds: with the builtin hipcc puts an s_waitcnt vmcnt(0) in front of every LDS
    // read that follows (any LDS access may alias a pending LDS-DMA in its book-keeping), which drains the requests this
    // ring exists to keep in flight.  The waits for ring data are the explicit wait_vmcnt() calls; the compiler's own
    // vmcnt counts for ordinary loads do not know about these requests and are therefore only ever too strict.
    u32x4 rs_ring = {0u, 0u, 0u, 0x00020000u};
    auto ring_desc = [&](const uint32_t c0) {                  // buffer descriptor over the chunks from c0 on (at most 2^31 bytes)
        const uint64_t left = (v_total - ((uint64_t)c0 << 8)) * 4ull;
        const uint64_t base = (uint64_t)(uintptr_t)bs_al + ((uint64_t)c0 << 10);
        return u32x4{(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base),
                     (uint32_t)__builtin_amdgcn_readfirstlane((int)((uint32_t)(base >> 32) & 0xFFFFu)),
                     (uint32_t)__builtin_amdgcn_readfirstlane((int)(left < 0x80000000ull ? (uint32_t)left : 0x80000000u)), 0x00020000u};
    };
    const uint32_t ring_lds = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)L.ring);
    auto ring_dma = [&](const u32x4 rs, const uint32_t c0, const uint32_t c, const uint32_t vc) {   // chunk c of the column into the slot of ring chunk vc: 64 lanes x 16 bytes, no VGPR destination
        const uint32_t dst = ring_lds + (vc & (RING_CHUNKS - 1u)) * 1024u;
        const uint32_t voff = (c - c0) * 1024u + (uint32_t)lane * 16u;
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, 0 offen" RING_DMA_MOD " lds\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(dst), "s"(rs) : "memory");
    };
    // chunks head .. lim - 1 (lim > head; ring numbering = column numbering) in one go: the loop is inside the asm statement —
    // seven scalar instructions and one vector add per chunk where the C++ loop around ring_dma cost nineteen (m0 saved and
    // restored per chunk, the slot address from scratch, the loop's own compares): the requests are a quarter of the stream
    // kernel's scalar instructions.  (The slot-address add into m0 is followed by the counter increment: the wait state an
    // LDS-DMA needs after a write of m0.)
#ifndef BLU_DMA_RUN
#define BLU_DMA_RUN 1
#endif
    auto ring_dma_run = [&](const u32x4 rs, const uint32_t c0, uint32_t head, const uint32_t lim) {
#if BLU_DMA_RUN
        uint32_t voff = (head - c0) * 1024u + (uint32_t)lane * 16u, keep, t;
        asm volatile("s_mov_b32 %[keep], m0\n"
                     "1:\n\t"
                     "s_and_b32 %[t], %[head], %[cm]\n\t"
                     "s_lshl_b32 %[t], %[t], 10\n\t"
                     "s_add_u32 m0, %[t], %[base]\n\t"
                     "s_add_u32 %[head], %[head], 1\n\t"
                     "buffer_load_dwordx4 %[voff], %[rs], 0 offen" RING_DMA_MOD " lds\n\t"
                     "v_add_u32 %[voff], 0x400, %[voff]\n\t"
                     "s_cmp_lt_u32 %[head], %[lim]\n\t"
                     "s_cbranch_scc1 1b\n\t"
                     "s_mov_b32 m0, %[keep]"
                     : [keep] "=&s"(keep), [t] "=&s"(t), [head] "+s"(head), [voff] "+v"(voff)
                     : [cm] "n"(RING_CHUNKS - 1u), [base] "s"(ring_lds), [rs] "s"(rs), [lim] "s"(lim) : "memory", "scc");
#else
        for (; head < lim; ++head) ring_dma(rs, c0, head, head);
#endif
    };

    // The offsets of a task are requested one task ahead (lane i: the row range of query q0 + i), so that their round
    // trip is never on the critical path and the next task's first chunks can be requested while this one finishes.
    // (The offsets are 64-bit in the ABI and below 2^32 by its n_hits limit: the kernel reads their low words only and does
    // all row arithmetic in 32 bits.  A corrupt table with high words set reads as some other in-range table: clamped like
    // any other, nothing faults.)
    const uint32_t n_hits32 = (uint32_t)h.n_hits;
    auto load_seg = [&](const uint32_t tq0, const uint32_t tnq, uint32_t& o, uint32_t& e) {   // (tnq = 0: no such task)
        o = 0; e = 0;
        if ((uint32_t)lane < tnq) {   // (lanes past the task's queries hold an empty segment)
            const uint32_t* lo32 = reinterpret_cast<const uint32_t*>(h.seg_off + ((uint64_t)tq0 + (uint32_t)lane));
            o = lo32[0]; e = lo32[2];
        }
    };
    uint32_t nx_off, nx_end;
    // the wave's tasks: wave, wave + n_waves, .. below t_full (64 queries each), then its piece of the tail
    uint32_t task = wave;
    bool in_tail = task >= t_full, tail_done = false;
    load_seg(in_tail ? tail_q0 : task * WAVE, in_tail ? tail_nq : (uint32_t)WAVE, nx_off, nx_end);
    STAMP_DECL

    for (; !in_tail || (tail_nq != 0u && !tail_done); tail_done = in_tail, task += n_waves, in_tail = task >= t_full) {
#ifndef BLU_LANE_LAUNDER
#define BLU_LANE_LAUNDER 1
#endif
        if (BLU_PRIO) __builtin_amdgcn_s_setprio(0);
#if BLU_LANE_LAUNDER
        // The lane id goes through an opaque statement once per task, so that nothing derived from it is loop-invariant to the
        // compiler: it had hoisted fifty-odd lane-derived addresses and masks (one instruction each to recompute) out of this loop
        // into registers of their own for the whole kernel — a third of the register file of a kernel that spills for lack of them.
        asm volatile("" : "+v"(lane));
#endif
        const uint32_t q0_32 = in_tail ? tail_q0 : task * WAVE;
        const uint64_t q0 = q0_32;
        const uint32_t nq = in_tail ? tail_nq : (uint32_t)WAVE;
        // the worklist queue this task appends to: by the 64-query block its first query lies in, so that a queue receives what
        // wl_capacity() sized it for whether the tail was cut into pieces or not
        const uint32_t wl_sel = (uint32_t)(q0 / WAVE) & (WL_QUEUES - 1u);
        uint32_t* const wl_cnt = work_count + WL_BASE + wl_sel * WL_STRIDE;
        uint32_t* const wl_q = worklist + (uint64_t)wl_sel * wl_cap;
        // lane i holds the row range of query q0 + i
        uint32_t my_off = nx_off, my_end = nx_end;
        if (my_end > n_hits32) my_end = n_hits32;   // defend the column reads against a corrupt offset table
        if (my_off > my_end) my_off = my_end;
        // the task after this one: the next whole-round task, else this wave's tail piece, else none (nx_nq = 0)
        const bool nx_tail = task + n_waves >= t_full;
        const uint32_t nx_q0 = nx_tail ? tail_q0 : (task + n_waves) * WAVE;
        const uint32_t nx_nq = in_tail ? 0u : (nx_tail ? tail_nq : (uint32_t)WAVE);
        load_seg(nx_q0, nx_nq, nx_off, nx_end);     // consumed after this task's phase 1 (prefetch decision) and by the next iteration
        asm volatile("" ::: "memory");              // (keeps these loads in front of the ring requests below: the counted waits rely on it)
        uint32_t fill = 0;   // wave-uniform: entries used in the LDS list
        bool keyed = false;  // wave-uniform: the list of this round holds comparison-ready entries (see gather_list)
        bool keyed_bad = false;   // wave-uniform: some comparison-ready entry of the round is a row without a (well-formed) lineage
        uint32_t scan_rpl = 16;   // wave-uniform: rows per lane of this round's ring steps (16 or 32: the descriptor format)
        uint32_t fill_ring = 0;   // wave-uniform: list entries of this round that came from ring steps (lane descriptors until the gather)
        // per-lane (= per-query) results of phase 2a
        // mode: 0 multi, 2 single, 3 nothing more to compute; rec_kind: 0 no record (worklist), 1 record in (ra, rb), 2 a status
        // record — (st_code, st_ref), put together when the records are staged: two registers through the task instead of
        // eight, and no packed status constant for the compiler to hoist into a register of its own for the whole kernel
        // (it did, five of them, and spilled one)
        const uint64_t q = q0 + (uint32_t)lane;
        const uint32_t row0 = my_off;
        // (r_len: lineage length of the reference row in bits 0..7, the shape hint of its side record above — packed layout,
        // 0 = none: the shape then comes from the row)
        uint32_t mode = 3, r_len = 0, r_row = 0, r_pos = 0, r_hdr = 0, minlen = 0, d = 0, rec_kind = 0, g_lo = 0, g_hi = 0;
        typedef typename PidKey<PID32>::type PK;
        PK r_pid = 0, max_pid = 0;   // fold(0.0, max): find_multi_taxa_consensus.rs:182-185
        uint4 ra = {0, 0, 0, 0}, rb = {0, 0, 0, 0};
        uint32_t st_code = 0, st_ref = 0;
        // a query reduced by a dense step (below): 0 no, 1 result in the variables above, 2 parse error in (dn_err, dn_pos),
        // 3 hand to the worklist kernel (a perc_identity that does not fit the packed key)
        uint32_t dn_flag = 0, dn_err = 0, dn_pos = 0, dn_k = 0;
        bool in_span = true;
        // One buffer descriptor per column, based at the task's first row and TASK_SPAN rows long: 32-bit lane byte
        // offsets (< 2^31 also for the 8-byte column), no 64-bit VALU address math, and the hardware range check
        // returns 0 for lanes past the end of the table or of the span.  A query that does not lie inside the span
        // (a giant segment earlier in the task, or an offset table that is not ascending) goes to the worklist.
        const uint32_t task_start = (uint32_t)rl((int)my_off, 0);
        const uint64_t rem = (uint64_t)(n_hits32 - task_start) < TASK_SPAN ? (uint64_t)(n_hits32 - task_start) : TASK_SPAN;
        const uint32_t rem4 = (uint32_t)(rem * 4), rem8 = (uint32_t)(rem * 8);
        const auto rs_bs = __builtin_amdgcn_make_buffer_rsrc((void*)(h.bitscore + task_start), 0, rem4, 0x00020000);
        // (packed layout: rs_tax is the descriptor of the 16-byte records; the other three column descriptors are unused)
        const auto rs_tax = PACKED ? __builtin_amdgcn_make_buffer_rsrc((void*)(h.packed + 4ull * task_start), 0, (uint32_t)(rem * 16), 0x00020000)
                            : WIDE ? __builtin_amdgcn_make_buffer_rsrc((void*)(h.packed64 + 6ull * task_start), 0, (uint32_t)(rem * 24), 0x00020000)
                                   : __builtin_amdgcn_make_buffer_rsrc((void*)(h.tax_row + task_start), 0, rem4, 0x00020000);
        const auto rs_aln = __builtin_amdgcn_make_buffer_rsrc((void*)(h.align_len + task_start), 0, rem4, 0x00020000);
        const auto rs_acc = __builtin_amdgcn_make_buffer_rsrc((void*)(h.acc_rank + task_start), 0, rem4, 0x00020000);
        const auto rs_pid = (PACKED || WIDE) ? rs_tax : PID32 ? __builtin_amdgcn_make_buffer_rsrc((void*)(h.pident_milli + task_start), 0, rem4, 0x00020000)
                                  : __builtin_amdgcn_make_buffer_rsrc((void*)(h.pident + task_start), 0, rem8, 0x00020000);
        // per-query {first row, row count} of the task, relative to task_start; count 0 also for segments > 64 rows
        // (those go to the worklist in phase 2a) so that phase 1 simply finds no top row in them
        {
            const uint32_t nrows = my_end - my_off;
            in_span = my_off >= task_start && (my_end - task_start) <= (uint32_t)TASK_SPAN;
            L.seg[lane] = make_uint2(my_off - task_start, (nrows <= MAX_TASK_SEG && in_span) ? nrows : 0u);
        }
        // A task whose queries lie back to back in ascending order, none longer than SHORT_SEG rows, can take its
        // bit-scores through the ring (lane i: does query i start where query i - 1 ends?)
        const uint32_t task_rows = ((uint32_t)lane < nq ? my_end : 0u) - ((uint32_t)lane < nq ? task_start : 0u);   // (meaningful in lane nq - 1)
        bool contiguous, all_short;
        {
            const uint32_t rel_end = my_end - task_start, rel_off = my_off - task_start;
            const uint32_t prev_end = (uint32_t)__shfl_up((int)rel_end, 1);
            const bool ok = in_span && (lane == 0 || rel_off == prev_end);
            contiguous = __ballot((uint32_t)lane < nq && !ok) == 0ull;
            all_short = __ballot((uint32_t)lane < nq && (my_end - my_off) > SHORT_SEG) == 0ull;
        }
        const uint32_t task_nrows = (uint32_t)rl((int)task_rows, (int)nq - 1);   // rows of the whole task (contiguous tasks)
        const uint64_t vbase = (uint64_t)task_start + mis;                       // v of the task's first row
        // Segments over SHORT_SEG rows are not read through the ring (long pass / worklist kernel): the whole chunks inside
        // them are left out of the ring's numbering, so that a task of mixed lengths streams only what its steps read.
        // Lane i (= query i): chunks left out before it (sk_before), where its own left-out chunks begin in the ring's
        // numbering and how many are gone after it (ev_*), and its first row in that numbering (seg_x).
        uint32_t sk_before = 0, ev_v = 0, ev_cum = 0, sk_total = 0;
        bool ev_on = false;
        if (BLU_MIXED_RING && contiguous && !all_short) {
            const uint64_t v0 = vbase + (my_off - task_start), v1 = vbase + (my_end - task_start);
            const uint32_t c_first = (uint32_t)((v0 + 255u) >> 8), c_past = (uint32_t)(v1 >> 8);
            const uint32_t kq = ((uint32_t)lane < nq && (my_end - my_off) > SHORT_SEG && c_past > c_first) ? c_past - c_first : 0u;
            uint32_t incl = kq;
            incl += (uint32_t)dpp<0x111>((int)incl);
            incl += (uint32_t)dpp<0x112>((int)incl);
            incl += (uint32_t)dpp<0x114>((int)incl);
            incl += (uint32_t)dpp<0x118>((int)incl);
            const uint32_t t0 = (uint32_t)rl((int)incl, 15), t1 = (uint32_t)rl((int)incl, 31), t2 = (uint32_t)rl((int)incl, 47), t3 = (uint32_t)rl((int)incl, 63);
            const uint32_t r16 = (uint32_t)lane >> 4;
            incl += r16 == 0 ? 0u : (r16 == 1 ? t0 : (r16 == 2 ? t0 + t1 : t0 + t1 + t2));
            sk_before = incl - kq;
            sk_total = t0 + t1 + t2 + t3;
            ev_on = kq != 0u;
            ev_v = c_first - sk_before;
            ev_cum = incl;
        }
        const uint32_t seg_x = (my_off - task_start) - 256u * sk_before;   // lane i: first row of query i in the ring's numbering, relative to the task
        L.vx[lane] = seg_x;
        auto ring_phys = [&](const uint32_t vc) {                               // chunk vc of the ring's numbering -> chunk of the column
            const uint64_t m = __ballot(ev_on && ev_v <= vc);
            return m ? vc + (uint32_t)rl((int)ev_cum, 63 - __builtin_clzll(m)) : vc;
        };
        // ---------------- phase 1: LPQ lanes per query, 4 consecutive rows per lane, 64 / LPQ queries per step ----------------
        // LPQ is chosen per task from its longest segment: 4 lanes (<= 16 rows: blutils' own default is
        // max_target_seqs = 10), 8 (<= 32), 16 (<= 64), 32 (<= 128) or all 64 lanes (<= 256 rows: BLAST's own default
        // of 500 target sequences rarely leaves more than that after blutils' identity / coverage filters), so that
        // short segments do not leave most lanes without a row to load and long ones still share the lane-per-query
        // finalisation of phase 2.
        struct StepRegs { u32x4 vbs, vtax, vp01, vp23, valn, vacc, vq; int left; uint32_t qi; };   // (vq: wide records only)
        // The non-bit-score values of a lane's four rows.  Column layouts: one 16-byte load per column (vtax, vp01[/vp23],
        // valn, vacc = the column's four rows).  Packed layout: one 16-byte load per ROW (vtax, vp01, valn, vacc = rows
        // 0..3, each {tax_row, pident_milli, align_len, acc_rank}), and only the rows in `rows` are requested.
        // voff = byte offset of the lane's first row in a 4-byte column, or the out-of-range sentinel.
        auto fetch_rest = [&](StepRegs& R, const uint32_t voff, const uint32_t rows) {
            const bool any = voff != 0xFFFFFFF0u && rows != 0u;
            if (PACKED) {
                const uint32_t base = voff * 4u;
                R.vtax = __builtin_amdgcn_raw_buffer_load_b128(rs_tax, (any && (rows & 1u)) ? base : 0xFFFFFFC0u, 0, STREAM_AUX);
                R.vp01 = __builtin_amdgcn_raw_buffer_load_b128(rs_tax, (any && (rows & 2u)) ? base + 16u : 0xFFFFFFC0u, 0, STREAM_AUX);
                R.valn = __builtin_amdgcn_raw_buffer_load_b128(rs_tax, (any && (rows & 4u)) ? base + 32u : 0xFFFFFFC0u, 0, STREAM_AUX);
                R.vacc = __builtin_amdgcn_raw_buffer_load_b128(rs_tax, (any && (rows & 8u)) ? base + 48u : 0xFFFFFFC0u, 0, STREAM_AUX);
                R.vp23 = R.vp01;
                return;
            }
            if (WIDE) {   // 24-byte records: the first 16 bytes of row r in vtax / vp01 / valn / vacc, its f64 in (vp23, vq)[r]
                const uint32_t base = voff * 6u;
                R.vtax = __builtin_amdgcn_raw_buffer_load_b128(rs_tax, (any && (rows & 1u)) ? base : 0xFFFFFFC0u, 0, STREAM_AUX);
                R.vp01 = __builtin_amdgcn_raw_buffer_load_b128(rs_tax, (any && (rows & 2u)) ? base + 24u : 0xFFFFFFC0u, 0, STREAM_AUX);
                R.valn = __builtin_amdgcn_raw_buffer_load_b128(rs_tax, (any && (rows & 4u)) ? base + 48u : 0xFFFFFFC0u, 0, STREAM_AUX);
                R.vacc = __builtin_amdgcn_raw_buffer_load_b128(rs_tax, (any && (rows & 8u)) ? base + 72u : 0xFFFFFFC0u, 0, STREAM_AUX);
                const u32x2 p0 = __builtin_amdgcn_raw_buffer_load_b64(rs_tax, (any && (rows & 1u)) ? base + 16u : 0xFFFFFFE0u, 0, STREAM_AUX);
                const u32x2 p1 = __builtin_amdgcn_raw_buffer_load_b64(rs_tax, (any && (rows & 2u)) ? base + 40u : 0xFFFFFFE0u, 0, STREAM_AUX);
                const u32x2 p2 = __builtin_amdgcn_raw_buffer_load_b64(rs_tax, (any && (rows & 4u)) ? base + 64u : 0xFFFFFFE0u, 0, STREAM_AUX);
                const u32x2 p3 = __builtin_amdgcn_raw_buffer_load_b64(rs_tax, (any && (rows & 8u)) ? base + 88u : 0xFFFFFFE0u, 0, STREAM_AUX);
                R.vp23 = u32x4{p0.x, p0.y, p1.x, p1.y};
                R.vq = u32x4{p2.x, p2.y, p3.x, p3.y};
                return;
            }
            const uint32_t vo = any ? voff : 0xFFFFFFF0u;
            R.vtax = __builtin_amdgcn_raw_buffer_load_b128(rs_tax, vo, 0, STREAM_AUX);
            if (PID32) {
                R.vp01 = __builtin_amdgcn_raw_buffer_load_b128(rs_pid, vo, 0, STREAM_AUX);   // four milli-percent values
                R.vp23 = R.vp01;
            } else {
                const uint32_t vo2 = any ? voff * 2u : 0xFFFFFFE0u;
                R.vp01 = __builtin_amdgcn_raw_buffer_load_b128(rs_pid, vo2, 0, STREAM_AUX);
                R.vp23 = __builtin_amdgcn_raw_buffer_load_b128(rs_pid, vo2 + 16u, 0, STREAM_AUX);
            }
            R.valn = __builtin_amdgcn_raw_buffer_load_b128(rs_aln, vo, 0, STREAM_AUX);
            R.vacc = __builtin_amdgcn_raw_buffer_load_b128(rs_acc, vo, 0, STREAM_AUX);
        };
        // row r of the lane as a list entry
        auto put_entry = [&](const uint32_t idx, const StepRegs& R, const int r, const uint32_t pos) {
            L.pq[idx] = (uint16_t)pos;
            if (PACKED) {   // the loaded record goes to the list as one 16-byte write
                const u32x4 rec = r == 0 ? R.vtax : (r == 1 ? R.vp01 : (r == 2 ? R.valn : R.vacc));
                L.rec[idx] = make_uint4(rec.x, rec.y, rec.z, rec.w);
                return;
            }
            if (WIDE) {     // {tax_row, pident low word, align_len, acc_rank} + the high word: the list entry of the f64 layout
                const u32x4 rec = r == 0 ? R.vtax : (r == 1 ? R.vp01 : (r == 2 ? R.valn : R.vacc));
                const uint32_t plo = r == 0 ? R.vp23.x : (r == 1 ? R.vp23.z : (r == 2 ? R.vq.x : R.vq.z));
                const uint32_t phi = r == 0 ? R.vp23.y : (r == 1 ? R.vp23.w : (r == 2 ? R.vq.y : R.vq.w));
                L.rec[idx] = make_uint4(rec.x, plo, rec.z, rec.w);
                L.p1[idx] = phi;
                return;
            }
            const uint32_t xt[4] = {R.vtax.x, R.vtax.y, R.vtax.z, R.vtax.w}, xa[4] = {R.valn.x, R.valn.y, R.valn.z, R.valn.w};
            const uint32_t xc[4] = {R.vacc.x, R.vacc.y, R.vacc.z, R.vacc.w};
            const uint32_t xm[4] = {R.vp01.x, R.vp01.y, R.vp01.z, R.vp01.w};   // milli-percent column
            const uint32_t xlo[4] = {R.vp01.x, R.vp01.z, R.vp23.x, R.vp23.z}, xhi[4] = {R.vp01.y, R.vp01.w, R.vp23.y, R.vp23.w};   // f64 column
            L.rec[idx] = make_uint4(xt[r], PID32 ? xm[r] : xlo[r], xa[r], xc[r]);
            if (!PID32) L.p1[idx] = xhi[r];
        };
        uint32_t stop_q = WAVE;      // first query index phase 1 did not get to in this round (the list was full), or 64
        uint32_t short_seg = SHORT_SEG;   // longest segment the streamed pass takes in this round (longer ones: the long pass)
        uint32_t first_q = 0;        // first pending query of the round
        auto phase1 = [&](const auto lpq, const bool sparse8) {   // wave-uniform width: a constant for 16 lanes, a variable otherwise
        const uint32_t LPQ = lpq;
        const uint32_t QPS = WAVE / LPQ;                          // queries per step
        // From 25-row segments on, the step is two-staged: bit-scores first, then the other four columns only in the lanes
        // that hold a top row (the others get an offset the descriptor rejects: no memory access).  Lines without a top
        // row are never fetched; measured on C3 (50 hits): 2.24 -> 1.82 ms; at 30 hits -6 %, at 20 hits +3 % (hence 25).
        const bool sparse = LPQ >= 16u || sparse8;
        const uint32_t grp = (uint32_t)lane / LPQ, sub4 = ((uint32_t)lane & (LPQ - 1u)) * 4u, row16 = (uint32_t)lane >> 4;
        // the loads of one step; lanes past the end of the segment get an offset the descriptor's range check
        // rejects: no memory access, no branch around the loads (so the wait counts below are exact).  No VALU write
        // touches a register with a load in flight, so nothing waits in front of the issue.
        auto issue = [&](uint32_t qb, StepRegs& R) {
            R.qi = qb + grp;                                     // this lane's query (>= nq: empty slot of the table)
            const uint2 sg = L.seg[R.qi];
            // rows of the segment from this lane's first row on (segments over SHORT_SEG rows belong to the long pass)
            R.left = (sg.y > short_seg ? 0 : (int)sg.y) - (int)sub4;
            const uint32_t voff = R.left > 0 ? (sg.x + sub4) * 4u : 0xFFFFFFF0u;
            R.vbs = __builtin_amdgcn_raw_buffer_load_b128(rs_bs, voff, 0, STREAM_AUX);
            if (sparse) return;                                  // the other columns: for top rows only, in tops()
            fetch_rest(R, voff, 0xFu);
        };
        // per step: what the second half (writing the list entries) needs from the first (scores -> top rows)
        struct StepTops { uint32_t idx0, tmask; bool fits; };
        // first half of a step: top score, top rows, list slots; in sparse mode the other four columns are requested here,
        // for the lanes that hold a top row only, so that the requests of all the steps of an iteration are in flight
        // together
        auto tops = [&](StepRegs& R, StepTops& T) {
            const int left = R.left;
            const uint32_t qi = R.qi;
            const u32x4 vbs = R.vbs;
            const int b0 = left > 0 ? (int)vbs.x : INT_MIN, b1 = left > 1 ? (int)vbs.y : INT_MIN;
            const int b2 = left > 2 ? (int)vbs.z : INT_MIN, b3 = left > 3 ? (int)vbs.w : INT_MIN;
            int M = imax(imax(b0, b1), imax(b2, b3));
            // M = the query's top bit-score: reduction over its LPQ lanes (quad_perm swaps, then half-row / row mirrors:
            // after the quad steps the lanes of a quad agree, so a mirror pairs every quad with its partner)
            M = imax(M, dpp<0xB1>(M));
            M = imax(M, dpp<0x4E>(M));
            if (LPQ >= 8) M = imax(M, dpp<0x141>(M));             // row_half_mirror
            if (LPQ >= 16) M = imax(M, dpp<0x140>(M));            // row_mirror
            if (LPQ >= 32) {                                      // the 16-lane rows agree inside; combine two or all four
                const int m0 = rl(M, 0), m1 = rl(M, 16), m2 = rl(M, 32), m3 = rl(M, 48);
                M = LPQ == 64 ? imax(imax(m0, m1), imax(m2, m3)) : (row16 < 2 ? imax(m0, m1) : imax(m2, m3));
            }
            const bool t0 = left > 0 && b0 == M, t1 = left > 1 && b1 == M, t2 = left > 2 && b2 == M, t3 = left > 3 && b3 == M;
            const uint32_t c = (uint32_t)t0 + (uint32_t)t1 + (uint32_t)t2 + (uint32_t)t3;
            uint32_t incl = c;                                    // inclusive prefix of the top-row counts inside the 16-lane row
            incl += (uint32_t)dpp<0x111>((int)incl);
            incl += (uint32_t)dpp<0x112>((int)incl);
            incl += (uint32_t)dpp<0x114>((int)incl);
            incl += (uint32_t)dpp<0x118>((int)incl);
            const uint32_t k0 = (uint32_t)rl((int)incl, 15), k1 = (uint32_t)rl((int)incl, 31);
            const uint32_t k2 = (uint32_t)rl((int)incl, 47), k3 = (uint32_t)rl((int)incl, 63);
            const bool fits = fill + k0 + k1 + k2 + k3 <= CAP;   // else: the queries of this step go to the worklist
            const uint32_t p1 = fill + k0, p2 = p1 + k1, p3 = p2 + k2;
            const uint32_t rbase = row16 == 0 ? fill : (row16 == 1 ? p1 : (row16 == 2 ? p2 : p3));   // list slot where this 16-lane row starts
            uint32_t gk;                                          // top rows of this lane's query
            if (LPQ == 64) gk = k0 + k1 + k2 + k3;
            else if (LPQ == 32) gk = row16 < 2 ? k0 + k1 : k2 + k3;
            else if (LPQ == 16) gk = row16 == 0 ? k0 : (row16 == 1 ? k1 : (row16 == 2 ? k2 : k3));
            else {
                gk = c;
                gk += (uint32_t)dpp<0xB1>((int)gk);
                gk += (uint32_t)dpp<0x4E>((int)gk);
                if (LPQ >= 8) gk += (uint32_t)dpp<0x141>((int)gk);
            }
            const uint32_t tmask = (uint32_t)t0 | ((uint32_t)t1 << 1) | ((uint32_t)t2 << 2) | ((uint32_t)t3 << 3);
            if (sparse && fits) fetch_rest(R, (L.seg[qi].x + sub4) * 4u, tmask);   // (offset recomputed rather than kept across the wait)
            T.idx0 = rbase + incl - c;                            // list slot of this lane's first top row (file order)
            T.tmask = tmask;
            T.fits = fits;
            if (sub4 == 0) L.meta[qi] = fits ? (T.idx0 | (gk << 16)) : META_SLOW;   // first lane of the query: its exclusive prefix
            if (fits) fill = p3 + k3;
        };
        // second half: the top rows' entries, in file order
        auto emit = [&](const StepRegs& R, const StepTops& T) {
            uint32_t idx = T.idx0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (T.fits && ((T.tmask >> r) & 1u)) {
                    put_entry(idx, R, r, sub4 + r);   // the row id carries the lineage length: phase 1 touches no taxonomy table
                    ++idx;
                }
            }
        };
        if (sparse) {
            // two-stage steps, software-pipelined: while the second-stage requests of iteration i are in flight, the
            // bit-scores of iteration i+1 are requested, so a wave pays one exposed round trip per iteration, not two
            StepRegs R[BLU_STEP_SETS];
            StepTops T[BLU_STEP_SETS];
            const uint32_t qb0 = first_q / (QPS * BLU_STEP_SETS) * (QPS * BLU_STEP_SETS);   // (queries before it are done)
#pragma unroll
            for (int u = 0; u < BLU_STEP_SETS; ++u) issue(qb0 + QPS * u, R[u]);
            for (uint32_t qb = qb0; qb < nq; qb += QPS * BLU_STEP_SETS) {
                bool all_fit = true;
#pragma unroll
                for (int u = 0; u < BLU_STEP_SETS; ++u) { asm volatile("" ::"v"(R[u].vbs)); tops(R[u], T[u]); all_fit &= T[u].fits; }
                StepRegs N[BLU_STEP_SETS];
                const bool more = all_fit && qb + QPS * BLU_STEP_SETS < nq;
                if (more) {
#pragma unroll
                    for (int u = 0; u < BLU_STEP_SETS; ++u) issue(qb + QPS * BLU_STEP_SETS + QPS * u, N[u]);
                }
#pragma unroll
                for (int u = 0; u < BLU_STEP_SETS; ++u) {
                    asm volatile("" ::"v"(R[u].vtax), "v"(R[u].vp01), "v"(R[u].vp23), "v"(R[u].valn), "v"(R[u].vacc)); if (WIDE) asm volatile("" ::"v"(R[u].vq));
                    emit(R[u], T[u]);
                }
                if (!all_fit) { stop_q = qb + QPS * BLU_STEP_SETS; break; }   // the list is full: the rest of the task in the next round
#pragma unroll
                for (int u = 0; u < BLU_STEP_SETS; ++u) { asm volatile("" ::"v"(N[u].vbs)); R[u].vbs = N[u].vbs; R[u].left = N[u].left; R[u].qi = N[u].qi; }
            }
            return;
        }
        for (uint32_t qb = first_q / (QPS * BLU_STEP_SETS) * (QPS * BLU_STEP_SETS); qb < nq; qb += QPS * BLU_STEP_SETS) {
            StepRegs R[BLU_STEP_SETS];
            StepTops T[BLU_STEP_SETS];
            bool all_fit = true;
#pragma unroll
            for (int u = 0; u < BLU_STEP_SETS; ++u) issue(qb + QPS * u, R[u]);
            if (sparse) {
#pragma unroll
                for (int u = 0; u < BLU_STEP_SETS; ++u) { asm volatile("" ::"v"(R[u].vbs)); tops(R[u], T[u]); all_fit &= T[u].fits; }
#pragma unroll
                for (int u = 0; u < BLU_STEP_SETS; ++u) {
                    asm volatile("" ::"v"(R[u].vtax), "v"(R[u].vp01), "v"(R[u].vp23), "v"(R[u].valn), "v"(R[u].vacc)); if (WIDE) asm volatile("" ::"v"(R[u].vq));
                    emit(R[u], T[u]);
                }
            } else {
#pragma unroll
                for (int u = 0; u < BLU_STEP_SETS; ++u) {
                    // every loaded register is read here on every path (otherwise hipcc parks a vmcnt(0) at the loop head)
                    asm volatile("" ::"v"(R[u].vbs), "v"(R[u].vtax), "v"(R[u].vp01), "v"(R[u].vp23), "v"(R[u].valn), "v"(R[u].vacc)); if (WIDE) asm volatile("" ::"v"(R[u].vq));
                    tops(R[u], T[u]);
                    all_fit &= T[u].fits;
                    emit(R[u], T[u]);
                }
            }
            if (!all_fit) { stop_q = qb + QPS * BLU_STEP_SETS; break; }   // the list is full: the rest of the task in the next round
        }
        };
        // Segments of 129..512 rows: a step holds two 256-row slots — two queries, or the two halves of one query longer
        // than 256 rows.  The bit-scores of both slots are loaded together; the other four columns are then read only by
        // the lanes that hold a top row (the rest get an offset the descriptor rejects: no memory access): top rows are
        // sparse in segments this long, so most of their lines are never fetched.
        auto phase1_long = [&](uint64_t long_mask) {
            const uint32_t sub4 = (uint32_t)lane * 4u, row16 = (uint32_t)lane >> 4;
            while (long_mask) {
                // next long query; one longer than 256 rows takes both slots (its two halves), otherwise the following
                // long query — if it fits a slot too — shares the step
                const uint32_t qi = (uint32_t)__builtin_ctzll(long_mask);
                long_mask &= long_mask - 1;
                const bool halves = L.seg[qi].y > 256u;
                uint32_t q_other = 64u;                           // 64 = none
                if (!halves && long_mask) {
                    const uint32_t cand = (uint32_t)__builtin_ctzll(long_mask);
                    if (L.seg[cand].y <= 256u) { q_other = cand; long_mask &= long_mask - 1; }
                }
                u32x4 vb[2];
                int left[2];
                uint32_t voff[2], qs[2], ro[2];
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    qs[hf] = (halves || hf == 0) ? qi : q_other;
                    ro[hf] = halves ? 256u * (uint32_t)hf : 0u;
                    uint2 sg = L.seg[qs[hf] < 64u ? qs[hf] : 0u];
                    if (qs[hf] >= 64u) sg.y = 0;
                    left[hf] = (int)sg.y - (int)ro[hf] - (int)sub4;
                    voff[hf] = left[hf] > 0 ? (sg.x + ro[hf] + sub4) * 4u : 0xFFFFFFF0u;
                    vb[hf] = __builtin_amdgcn_raw_buffer_load_b128(rs_bs, voff[hf], 0, STREAM_AUX);
                }
                asm volatile("" ::"v"(vb[0]), "v"(vb[1]));
                int bs[2][4], m[2];
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    const uint32_t v[4] = {vb[hf].x, vb[hf].y, vb[hf].z, vb[hf].w};
                    m[hf] = INT_MIN;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { bs[hf][r] = left[hf] > r ? (int)v[r] : INT_MIN; m[hf] = imax(m[hf], bs[hf][r]); }
                }
                if (halves) { m[0] = imax(m[0], m[1]); m[0] = wave_max_i32(m[0]); m[1] = m[0]; }   // one query: one top bit-score
                else { m[0] = wave_max_i32(m[0]); m[1] = wave_max_i32(m[1]); }
                uint32_t c[2], slot[2], tot[2], tmask[2];
#pragma unroll
                for (int hf = 0; hf < 2; ++hf) {
                    c[hf] = 0; tmask[hf] = 0;
#pragma unroll
                    for (int r = 0; r < 4; ++r) { const bool t = left[hf] > r && bs[hf][r] == m[hf]; tmask[hf] |= (uint32_t)t << r; c[hf] += (uint32_t)t; }
                    uint32_t in = c[hf];
                    in += (uint32_t)dpp<0x111>((int)in);
                    in += (uint32_t)dpp<0x112>((int)in);
                    in += (uint32_t)dpp<0x114>((int)in);
                    in += (uint32_t)dpp<0x118>((int)in);
                    const uint32_t k0 = (uint32_t)rl((int)in, 15), k1 = (uint32_t)rl((int)in, 31), k2 = (uint32_t)rl((int)in, 47), k3 = (uint32_t)rl((int)in, 63);
                    tot[hf] = k0 + k1 + k2 + k3;
                    slot[hf] = (row16 == 0 ? 0u : (row16 == 1 ? k0 : (row16 == 2 ? k0 + k1 : k0 + k1 + k2))) + in - c[hf];
                }
                const uint32_t gk = tot[0] + tot[1];
                const bool fits = fill + gk <= CAP;
                if (lane == 0) {
                    if (halves) L.meta[qi] = fits ? (fill | (gk << 16)) : META_SLOW;
                    else {
                        L.meta[qs[0]] = fits ? (fill | (tot[0] << 16)) : META_SLOW;
                        if (qs[1] < 64u) L.meta[qs[1]] = fits ? ((fill + tot[0]) | (tot[1] << 16)) : META_SLOW;
                    }
                }
                if (!fits) {                                      // the list is full: the remaining long queries in the next round
                    uint64_t rest = long_mask;
                    while (rest) { const uint32_t qn = (uint32_t)__builtin_ctzll(rest); rest &= rest - 1; if (lane == 0) L.meta[qn] = META_SLOW; }
                    break;
                }
                {
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        if (tot[hf] == 0) continue;                       // wave-uniform: no top row in this slot
                        StepRegs S;
                        fetch_rest(S, voff[hf], tmask[hf]);
                        uint32_t idx = fill + (hf ? tot[0] : 0u) + slot[hf];   // file order: first slot, then second
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            if ((tmask[hf] >> r) & 1u) {
                                put_entry(idx, S, r, ro[hf] + sub4 + r);
                                ++idx;
                            }
                        }
                    }
                    fill += gk;
                }
            }
        };
        // ---------------- phase 1, FLAT pass (kernel without the ring): lanes dealt to the queries by need ----------------
        // A task of mixed segment lengths (Zipf-like hit counts) wastes most lanes of the passes above: every streamed query
        // pays the lanes of the widest one, and each longer one takes half a step of its own.  Here the rows of the round's
        // queries (up to FLAT_SEG = MAX_TASK_SEG rows: all of them) are cut into units of FLAT_ROWS consecutive rows, numbered
        // through the task; step s = units 64 s .. 64 s + 63, lane l its l-th unit, of whichever query owns it (binary search
        // over the queries' first units, in LDS).  Two sub-passes over the steps, neither with a memory round trip per step:
        //   A  the bit-scores of FLAT_DEPTH steps are in flight at a time; a lane keeps, per step, its own maximum and which of
        //      its rows reach it (two registers), and the query's top bit-score meets in LDS (atomic max over its lanes);
        //   B  a lane whose maximum is the query's marks its rows; the top rows get their list slots by a scan over the units —
        //      unit order is file order — and what goes to the list is a lane descriptor, as in a ring round (top-row mask,
        //      first row, position): the side records are gathered afterwards, one list entry per lane and all of a round's
        //      requests in flight together (gather_list).
        // A round takes the queries whose units end within FLAT_STEPS steps; the others come in the next round.
        // (Round 3, first version: whole queries per step, quads, 256 rows, side records fetched inside the step — one round
        // trip per step, and the segments of 257..512 rows in the unpipelined long pass: phase 1 was 77 % of a C5 task.)
        auto phase1_flat = [&]() {
            if constexpr (!RING) {
            constexpr uint32_t UR = FLAT_ROWS, MAXS = FLAT_STEPS, DEPTH = FLAT_DEPTH;
            static_assert(UR == 8u, "two 16-byte loads per lane");
            static_assert(FLAT_SEG >= MAX_TASK_SEG, "the flat pass takes every segment of the task");
            static_assert(DEPTH <= MAXS && MAXS * WAVE >= FLAT_SEG / UR, "a query's units fit one round");
            const uint32_t rows = L.seg[lane].y;                 // this lane's query: 0 = nothing to do in this round
            const uint32_t nunit = (rows + UR - 1u) / UR;
            uint32_t incl = nunit;
            incl += (uint32_t)dpp<0x111>((int)incl);
            incl += (uint32_t)dpp<0x112>((int)incl);
            incl += (uint32_t)dpp<0x114>((int)incl);
            incl += (uint32_t)dpp<0x118>((int)incl);
            const uint32_t t0 = (uint32_t)rl((int)incl, 15), t1 = (uint32_t)rl((int)incl, 31), t2 = (uint32_t)rl((int)incl, 47);
            const uint32_t r16 = (uint32_t)lane >> 4;
            incl += r16 == 0 ? 0u : (r16 == 1 ? t0 : (r16 == 2 ? t0 + t1 : t0 + t1 + t2));
            const uint32_t q_end = incl, q_begin = incl - nunit;   // this query's units in the round's numbering
            // the round: the queries whose units end within MAXS steps (a prefix of the pending ones; never empty: one query has
            // at most 64 units)
            const uint64_t out_m = __ballot(rows != 0u && q_end > MAXS * WAVE);
            const uint32_t q_lim = out_m ? (uint32_t)__builtin_ctzll(out_m) : WAVE;
            const uint32_t total = q_lim < WAVE ? (uint32_t)rl((int)q_begin, (int)q_lim) : (uint32_t)rl((int)q_end, WAVE - 1);
            const uint32_t ns = (total + WAVE - 1u) / WAVE;      // steps of the round (<= MAXS)
            L.qs[lane] = q_begin;
            L.qm[lane] = INT_MIN;
            L.qk[lane] = 0u;
            if (rows != 0u) L.meta[lane] = META_SLOW;            // until the query is complete in the list
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            // lane's unit of step st: owner = the last query whose first unit is <= it; packed as query | unit in the query << 6 |
            // rows it holds (0 .. 8) << 12; the two loads are issued whether the unit exists or not (an offset the descriptor
            // rejects: no memory access), so the wait counts of sub-pass A are exact
            auto issue = [&](const uint32_t st, u32x4& lo4, u32x4& hi4, uint32_t& um) {
                um = 0u;
                uint32_t voff = 0xFFFFFFF0u;
                int left = 0;
                if (st < ns) {                                   // (wave-uniform)
                    const uint32_t g = st * WAVE + (uint32_t)lane;
                    uint32_t q = 0;
#pragma unroll
                    for (uint32_t sp = 32; sp; sp >>= 1) { const uint32_t c = q + sp; if (L.qs[c] <= g) q = c; }
                    const uint2 sg = L.seg[q];
                    const uint32_t oq = g - L.qs[q];
                    left = g < total ? (int)sg.y - (int)(UR * oq) : 0;
                    if (left > 0) { voff = (sg.x + UR * oq) * 4u; um = q | (oq << 6) | (umin((uint32_t)left, UR) << 12); }
                }
                lo4 = __builtin_amdgcn_raw_buffer_load_b128(rs_bs, voff, 0, STREAM_AUX);
                hi4 = __builtin_amdgcn_raw_buffer_load_b128(rs_bs, left > 4 ? voff + 16u : 0xFFFFFFF0u, 0, STREAM_AUX);
            };
            u32x4 ba[DEPTH], bb[DEPTH];
            uint32_t um[MAXS];
            int lmx[MAXS];
#pragma unroll
            for (uint32_t st = 0; st < MAXS; ++st) { um[st] = 0u; lmx[st] = INT_MIN; }
#pragma unroll
            for (uint32_t st = 0; st < DEPTH; ++st) issue(st, ba[st], bb[st], um[st]);
            // ---- A: per step the lane's maximum and the rows that reach it; the query's maximum in LDS
#pragma unroll
            for (uint32_t st = 0; st < MAXS; ++st) {
                if (st >= ns) break;
                const u32x4 va = ba[st % DEPTH], vb = bb[st % DEPTH];
                const uint32_t left = (um[st] >> 12) & 15u;
                const uint32_t v[8] = {va.x, va.y, va.z, va.w, vb.x, vb.y, vb.z, vb.w};
                int bsv[UR], mx = INT_MIN;
#pragma unroll
                for (uint32_t r = 0; r < UR; ++r) { bsv[r] = left > r ? (int)v[r] : INT_MIN; mx = imax(mx, bsv[r]); }
                uint32_t lmask = 0;
#pragma unroll
                for (uint32_t r = 0; r < UR; ++r) lmask |= (uint32_t)(left > r && bsv[r] == mx) << r;
                um[st] |= lmask << 16;
                lmx[st] = mx;
                if (left) atomicMax(&L.qm[um[st] & 63u], mx);
                if (st + DEPTH < MAXS) issue(st + DEPTH, ba[st % DEPTH], bb[st % DEPTH], um[st + DEPTH]);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            // ---- B: top rows, list slots, descriptors
            uint32_t running = fill, q_cut = q_lim;
#pragma unroll
            for (uint32_t st = 0; st < MAXS; ++st) {
                if (st >= ns) break;
                const uint32_t q = um[st] & 63u, oq = (um[st] >> 6) & 63u;
                const bool have = ((um[st] >> 12) & 15u) != 0u;
                const int M = L.qm[q];
                const uint32_t tmask = (have && lmx[st] == M) ? (um[st] >> 16) & 0xFFu : 0u;   // bit r = row r of the unit ties on the query's top score
                const uint32_t c = (uint32_t)__builtin_popcount(tmask);
                uint32_t in2 = c;
                in2 += (uint32_t)dpp<0x111>((int)in2);
                in2 += (uint32_t)dpp<0x112>((int)in2);
                in2 += (uint32_t)dpp<0x114>((int)in2);
                in2 += (uint32_t)dpp<0x118>((int)in2);
                const uint32_t k0 = (uint32_t)rl((int)in2, 15), k1 = (uint32_t)rl((int)in2, 31), k2 = (uint32_t)rl((int)in2, 47), k3 = (uint32_t)rl((int)in2, 63);
                in2 += r16 == 0 ? 0u : (r16 == 1 ? k0 : (r16 == 2 ? k0 + k1 : k0 + k1 + k2));
                const uint32_t idx0 = running + in2 - c;          // list slot of this lane's first top row
                if (have && oq == 0u) L.meta[q] = META_SLOW | idx0;   // the query's first slot (under the flag until the query is complete)
                if (c) atomicAdd(&L.qk[q], c);
                const bool over = have && idx0 + c > CAP;
                const uint64_t over_m = __ballot(over);
                const bool room = !over && c != 0u && (over_m == 0ull || (uint32_t)lane < (uint32_t)__builtin_ctzll(over_m));
                // the lane's descriptor, at the slot of its first top row: mask with row 0 in bit 31 (what gather_list counts
                // through), first row relative to the task, position in the segment
                if (room) L.rec[idx0] = make_uint4(__builtin_bitreverse32(tmask), L.seg[q].x + UR * oq, UR * oq, 0u);
                if (over_m) { q_cut = (uint32_t)rl((int)q, __builtin_ctzll(over_m)); break; }   // the list is full: from this query on, the next round
                running += k0 + k1 + k2 + k3;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            // lane = query again: complete queries get {first slot, top-row count}; the one the list filled up in and those
            // behind it stay marked for the next round (a query larger than the whole list makes no progress: worklist)
            const uint32_t m = L.meta[lane];
            if (rows != 0u && (uint32_t)lane < q_cut) L.meta[lane] = (m & 0xFFFFu) | (L.qk[lane] << 16);
            fill = q_cut < q_lim ? (L.meta[q_cut] & 0xFFFFu) : running;
            if (q_cut < WAVE) stop_q = q_cut;
            }
        };
        // ---------------- phase 1 of a ring task: lane = RPL consecutive rows, read from the LDS ring ----------------
        // A step takes 64 / LPQ queries, LPQ = 1, 2, 4 or 8 lanes per query by the task's longest segment (<= 16 RPL / 8 = 128
        // rows); a lane scans its rows one after the other — the bit-scores are in LDS, so a per-lane address costs
        // nothing — and only the maximum and the counts cross lanes: a quarter of the vector instructions of the
        // four-rows-per-lane steps above.  Top rows go to the list as ROW INDICES; their 16 other bytes are gathered
        // afterwards, one list entry per lane (gather_list).
        auto ring_mark_landed = [&](const uint32_t upto) {      // chunks below `upto` have landed; keeps the pad a mirror of ring[0 .. RING_PAD)
            const uint32_t m0 = (ring_landed + RING_CHUNKS - 1u) & ~(RING_CHUNKS - 1u);   // first slot-0 chunk not yet marked
            if (m0 < upto) {
                if ((uint32_t)lane < RING_PAD) L.ring[RING_ROWS + lane] = L.ring[lane];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
            ring_landed = upto;
        };
        auto ring_refill = [&]() {                              // request what the ring has room for, in chunk order
            if (ring_head < ring_tail) { ring_head = ring_tail; if (ring_landed < ring_tail) ring_landed = ring_tail; }
            if (BLU_MIXED_RING && sk_total) {
                while (ring_head < ring_end && ring_head - ring_tail < RING_CHUNKS) { ring_dma(rs_ring, ring_c0, ring_phys(ring_head), ring_head); ++ring_head; }
            } else {
                const uint32_t lim = ring_end < ring_tail + RING_CHUNKS ? ring_end : ring_tail + RING_CHUNKS;
                if (ring_head < lim) { ring_dma_run(rs_ring, ring_c0, ring_head, lim); ring_head = lim; }   // (raised priority around these requests: no effect)
            }
        };
        // FULL (every streamed segment of the round has at least RPL rows): the lane that would run past the end of its segment
        // takes the segment's LAST RPL rows instead — rows it shares with the lane before it are cleared from its top-row mask —
        // so every lane that has rows has RPL of them and no row needs masking: the two instructions per row that cost went
        // into nothing else (C3: 50-row segments, 32 rows per lane, the second lane of a query held 18).
        auto phase1_scan = [&](const auto rpl_c, const auto full_c, const uint32_t LPQ) {
            constexpr uint32_t RPL = decltype(rpl_c)::value;
            constexpr bool FULL = decltype(full_c)::value;
            static_assert(RPL <= RING_PAD && RPL <= 32u, "rows per lane");
            const uint32_t QPS = WAVE / LPQ;
            const uint32_t grp = (uint32_t)lane / LPQ, sub = ((uint32_t)lane & (LPQ - 1u)) * RPL, row16 = (uint32_t)lane >> 4;
            {
                // A round that starts behind what the ring still holds (a query that an earlier round left to the long pass
                // and this round's width takes into the steps) starts the ring over at its first row.
                const uint32_t cs = (uint32_t)((vbase + (uint32_t)rl((int)seg_x, (int)first_q)) >> 8);
                if (cs < ring_tail) { wait_vmcnt(0u); ring_head = ring_landed = ring_tail = cs; }
            }
            for (uint32_t qb = first_q / QPS * QPS; qb < nq; qb += QPS) {
                // the rows of this step lie back to back: [first row of query qb, first row of query qb + QPS)
                const uint32_t r_lo = (uint32_t)rl((int)seg_x, (int)(qb < first_q ? first_q : qb));   // (queries before first_q are done)
                const uint32_t qn = qb + QPS;
                const uint32_t r_hi = qn < nq ? (uint32_t)rl((int)seg_x, (int)qn) : task_nrows - 256u * sk_total;
                STAMP(1)
                ring_tail = (uint32_t)((vbase + r_lo) >> 8);
                ring_refill();
                if (r_hi > r_lo) {
                    const uint32_t need = (uint32_t)((vbase + r_hi - 1u) >> 8);
                    if (need >= ring_landed) { wait_vmcnt(ring_head - 1u - need); ring_mark_landed(need + 1u); }   // all but the chunks requested after `need`
                }
                STAMP(10)   // (waiting for ring data)
                const uint32_t qi = qb + grp;
                const uint2 sg = L.seg[qi];
                const int left0 = (sg.y > short_seg ? 0 : (int)sg.y) - (int)sub;     // rows of the segment from this lane's nominal first row on
                // FULL: a lane with fewer than RPL rows left moves back by `over` rows, to the last RPL rows of its segment
                const uint32_t over = (FULL && left0 > 0 && left0 < (int)RPL) ? RPL - (uint32_t)left0 : 0u;
                const uint32_t sub_e = sub - over;                                   // position of the lane's first row in its segment
                const int left = FULL ? (left0 > 0 ? (int)RPL : 0) : left0;          // rows the lane holds
                const uint32_t row0 = sg.x + sub_e;                                  // (in the column, relative to the task: what the gather reads)
                const uint32_t a = ((uint32_t)vbase + L.vx[qi] + sub_e) & RING_MASK;
                int b[RPL];
#pragma unroll
                for (uint32_t i = 0; i < RPL; ++i) b[i] = (int)L.ring[a + i];
                // the step's rows are in registers: its chunks can be overwritten, so what follows is requested now — two
                // steps ahead of the step that reads it (the DMA writes LDS: not before the reads above have returned)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                for (uint32_t i = 0; i < RPL; ++i) asm volatile("" : "+v"(b[i]));
                if (r_hi > r_lo) { ring_tail = (uint32_t)((vbase + r_hi) >> 8); ring_refill(); }
                int M = INT_MIN;
                if constexpr (FULL) {
#pragma unroll
                    for (uint32_t i = 0; i < RPL; ++i) M = imax(M, b[i]);
                    M = left > 0 ? M : INT_MIN;                       // (a lane past the end of its segment read another query's rows)
                } else {
#pragma unroll
                    for (uint32_t i = 0; i < RPL; ++i) { b[i] = (int)i < left ? b[i] : INT_MIN; M = imax(M, b[i]); }
                }
                if (LPQ >= 2) M = imax(M, dpp<0xB1>(M));
                if (LPQ >= 4) M = imax(M, dpp<0x4E>(M));
                if (LPQ >= 8) M = imax(M, dpp<0x141>(M));             // row_half_mirror
                if (LPQ >= 16) M = imax(M, dpp<0x140>(M));            // row_mirror
                if (BLU_X_SCAN_MIN) { if (sub == 0) L.meta[qi] = (uint32_t)M & 0xFFu; continue; }
                uint32_t mask = tie_mask<RPL>(b, M);                  // bit RPL - 1 - i = row i ties on the query's top score
                mask = left > 0 ? mask : 0u;
                if constexpr (FULL) mask &= 0xFFFFFFFFu >> (32u - RPL + over);   // rows 0 .. over - 1 are the previous lane's
                const uint32_t c = (uint32_t)__builtin_popcount(mask);
                uint32_t incl = c;                                    // inclusive prefix of the top-row counts inside the 16-lane row
                incl += (uint32_t)dpp<0x111>((int)incl);
                incl += (uint32_t)dpp<0x112>((int)incl);
                incl += (uint32_t)dpp<0x114>((int)incl);
                incl += (uint32_t)dpp<0x118>((int)incl);
                const uint32_t k0 = (uint32_t)rl((int)incl, 15), k1 = (uint32_t)rl((int)incl, 31);
                const uint32_t k2 = (uint32_t)rl((int)incl, 47), k3 = (uint32_t)rl((int)incl, 63);
                const bool fits = fill + k0 + k1 + k2 + k3 <= CAP;
                const uint32_t p1 = fill + k0, p2 = p1 + k1, p3 = p2 + k2;
                const uint32_t rbase = row16 == 0 ? fill : (row16 == 1 ? p1 : (row16 == 2 ? p2 : p3));
                uint32_t gk = c;                                      // top rows of this lane's query
                if (LPQ >= 2) gk += (uint32_t)dpp<0xB1>((int)gk);
                if (LPQ >= 4) gk += (uint32_t)dpp<0x4E>((int)gk);
                if (LPQ >= 8) gk += (uint32_t)dpp<0x141>((int)gk);
                if (LPQ >= 16) gk += (uint32_t)dpp<0x140>((int)gk);
#ifndef BLU_DENSE_WHEN_FULL
#define BLU_DENSE_WHEN_FULL 1
#endif
                if (PID32 && (BLU_DENSE_WHEN_FULL ? !fits : k0 + k1 + k2 + k3 > CAP)) {
                    if (BLU_PRIO_DENSE) __builtin_amdgcn_s_setprio(2);   // (its record loads are round trips of the task's own chain)
                    // ---- a DENSE step: the top rows of this step do not fit what is left of the list (many hits tie on the top
                    // score — identical database sequences; round 3: also when the steps before it have filled the list — that used
                    // to end the round and send the rest of the task through phase 1 again: tables with the reference's real
                    // top-group sizes 1.356 -> 1.267 ms).  No list: every lane fetches the side records of ITS OWN top rows,
                    // four requests in flight, and reduces them as they come; the lanes of a query then merge (DPP) and the
                    // result goes to the query's lane of phase 2 (ds_bpermute).  Same rule as phase 2a: the reference row is
                    // the maximum (Relaxed) / minimum (Cautious) of (length, perc_identity, align_length, accession), full ties
                    // settled by the position in the file (the later / the earlier row).
                    // running result of a lane: best row (BK, bacc, bpos -> brow), smallest (length, pident) word, largest pident
                    // (bit 31: a pident that does not fit the key), span [dlo, dhi], first parse error as position << 8 | status.
                    // kmin == all ones: no record yet.
                    uint64_t BK = 0;
                    uint32_t bacc = 0, bpos = 0, brow = 0, kmin = 0xFFFFFFFFu, pmax = 0, dlo = 0xFFFFFFFFu, dhi = 0, err = 0xFFFFFFFFu;
                    auto reset = [&]() { BK = 0; bacc = 0; bpos = 0; brow = 0; kmin = 0xFFFFFFFFu; pmax = 0; dlo = 0xFFFFFFFFu; dhi = 0; err = 0xFFFFFFFFu; };
                    auto better = [&](const uint64_t K, const uint32_t acc, const uint32_t pos, const uint64_t K2, const uint32_t acc2, const uint32_t pos2) {
                        const bool gt = (K > K2) | ((K == K2) & (acc > acc2)), eq = (K == K2) & (acc == acc2);
                        return STRAT == BLU_RELAXED ? (gt | (eq & (pos > pos2))) : ((!gt & !eq) | (eq & (pos < pos2)));
                    };
                    // one side record (position gp in its segment; on = the lane has one) into the lane's running result
                    auto take_record = [&](const u32x4 rec, const uint32_t gp, const bool on) {
                        const uint32_t len = umin(rec.x >> BLU_ROW_BITS, t.max_depth), pos = rec.x & ROW_MASK;
                        const bool unmatched = pos >= t.n_tax, bad = !unmatched && (rec.x >> BLU_ROW_BITS) == 0;
                        const uint32_t e = (gp << 8) | (bad ? (uint32_t)BLU_ST_ERR_BAD_LINEAGE : (uint32_t)BLU_ST_ERR_UNMATCHED_TAXID);
                        err = (on && (unmatched | bad)) ? umin(err, e) : err;
                        const uint32_t pm = rec.y & ((1u << KEY_PID_BITS) - 1u);
                        const uint32_t k1 = (len << KEY_PID_BITS) | pm;
                        const uint64_t K = ((uint64_t)k1 << 32) | (rec.z ^ 0x80000000u);
                        const bool take = on & ((kmin == 0xFFFFFFFFu) | better(K, rec.w, gp, BK, bacc, bpos));
                        BK = take ? K : BK; bacc = take ? rec.w : bacc; bpos = take ? gp : bpos; brow = take ? pos : brow;
                        kmin = on ? umin(kmin, k1) : kmin;
                        const uint32_t pmo = pm | ((!PACKED && rec.y >= (1u << KEY_PID_BITS)) ? 0x80000000u : 0u);   // (packed: bits 17.. are the shape hint)
                        pmax = (on && pmo > pmax) ? pmo : pmax;
                        dlo = on ? umin(dlo, pos) : dlo;
                        dhi = (on && pos > dhi) ? pos : dhi;
                    };
                    auto load_record = [&](const uint32_t row, const bool on) {
                        u32x4 r;
                        if (PACKED) r = __builtin_amdgcn_raw_buffer_load_b128(rs_tax, on ? row * 16u : 0xFFFFFFC0u, 0, GATHER_AUX);
                        else {
                            const uint32_t o4 = on ? row * 4u : 0xFFFFFFF0u;
                            r.x = __builtin_amdgcn_raw_buffer_load_b32(rs_tax, o4, 0, GATHER_AUX);
                            r.y = __builtin_amdgcn_raw_buffer_load_b32(rs_pid, o4, 0, GATHER_AUX);
                            r.z = __builtin_amdgcn_raw_buffer_load_b32(rs_aln, o4, 0, GATHER_AUX);
                            r.w = __builtin_amdgcn_raw_buffer_load_b32(rs_acc, o4, 0, GATHER_AUX);
                        }
                        return r;
                    };
                    // merge with the lane the DPP control pairs this one with
                    auto merge = [&](auto ctrl) {
                        constexpr int C = decltype(ctrl)::value;
                        const uint64_t K2 = ((uint64_t)(uint32_t)dpp<C>((int)(uint32_t)(BK >> 32)) << 32) | (uint32_t)dpp<C>((int)(uint32_t)BK);
                        const uint32_t acc2 = (uint32_t)dpp<C>((int)bacc), pos2 = (uint32_t)dpp<C>((int)bpos), row2 = (uint32_t)dpp<C>((int)brow);
                        const uint32_t kmin2 = (uint32_t)dpp<C>((int)kmin);
                        const bool take = (kmin2 != 0xFFFFFFFFu) & ((kmin == 0xFFFFFFFFu) | better(K2, acc2, pos2, BK, bacc, bpos));
                        BK = take ? K2 : BK; bacc = take ? acc2 : bacc; bpos = take ? pos2 : bpos; brow = take ? row2 : brow;
                        kmin = umin(kmin, kmin2);
                        const uint32_t pm2 = (uint32_t)dpp<C>((int)pmax); pmax = pm2 > pmax ? pm2 : pmax;
                        dlo = umin(dlo, (uint32_t)dpp<C>((int)dlo));
                        const uint32_t hi2 = (uint32_t)dpp<C>((int)dhi); dhi = hi2 > dhi ? hi2 : dhi;
                        err = umin(err, (uint32_t)dpp<C>((int)err));
                    };
                    // the query's lane of phase 2 (lane = query) fetches the merged result from lane src / 4 (ds_bpermute)
                    auto deliver = [&](const int src, const bool mine) {
                        const uint32_t f_khi = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)(uint32_t)(BK >> 32));
                        const uint32_t f_pos = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)bpos), f_row = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)brow);
                        const uint32_t f_kmin = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)kmin), f_pmax = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)pmax);
                        const uint32_t f_lo = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)dlo), f_hi = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)dhi);
                        const uint32_t f_err = (uint32_t)__builtin_amdgcn_ds_bpermute(src, (int)err);
                        // (selects, not branches: conditional stores to these by-reference captures end up in scratch memory)
                        const bool got = mine && f_kmin != 0xFFFFFFFFu;
                        const bool g_ovf = got && (f_pmax & 0x80000000u) != 0u, g_err = got && !g_ovf && f_err != 0xFFFFFFFFu, g_ok = got && !g_ovf && !g_err;
                        dn_flag = g_ovf ? 3u : (g_err ? 2u : (g_ok ? 1u : dn_flag));
                        dn_err = g_err ? (f_err & 0xFFu) : dn_err;
                        dn_pos = g_err ? (f_err >> 8) : dn_pos;
                        if constexpr (PID32) { r_pid = g_ok ? (f_khi & ((1u << KEY_PID_BITS) - 1u)) : r_pid; max_pid = g_ok ? f_pmax : max_pid; }
                        r_len = g_ok ? (f_khi >> KEY_PID_BITS) : r_len; r_row = g_ok ? f_row : r_row; r_pos = g_ok ? f_pos : r_pos;
                        minlen = g_ok ? (f_kmin >> KEY_PID_BITS) : minlen; g_lo = g_ok ? f_lo : g_lo; g_hi = g_ok ? f_hi : g_hi;
                    };
                    const uint32_t tq = (uint32_t)lane;                               // this lane as a query of the task
                    const bool mine = tq >= qb && tq < qn && tq < nq;
                    const uint32_t own0 = (mine ? tq - qb : 0u) * LPQ;                 // first scanning lane of this lane's query
                    if (LPQ <= 4u && BLU_DENSE_Q * (k0 + k1 + k2 + k3) >= r_hi - r_lo) {
                        // At least a third of the step's rows are top rows (whole groups tied): reading every record of the step is
                        // cheaper than picking them out (2 M x 50 hits, first g rows tied: g = 15 / 20 at 4.25 / 4.16 Gq/s, 3.59 / 3.09
                        // with one half as the bar; g = 50 at 3.4).
                        // The scanning lanes ("owners": RPL consecutive rows each) are served 16 at a time, FOUR LANES PER OWNER:
                        // lane 4 a + c takes rows c, 4 + c, 8 + c .. of owner 16 g + a, so a request of the wave is 16 runs of 64
                        // contiguous bytes instead of 64 scattered 16-byte pieces (all-tied 50-hit table: 1.44 -> 0.63 ms — the
                        // step was bound by the number of requests, not by their bytes).  The quad merges into the owner's
                        // result, 4 LPQ lanes into the query's.
                        const uint32_t c4 = (uint32_t)lane & 3u;
#pragma nounroll
                        for (uint32_t g = 0; g < 4u; ++g) {
                            const int own = (int)((16u * g + ((uint32_t)lane >> 2)) * 4u);
                            const uint32_t o_row0 = (uint32_t)__builtin_amdgcn_ds_bpermute(own, (int)row0);
                            const uint32_t o_mask = (uint32_t)__builtin_amdgcn_ds_bpermute(own, (int)mask);
                            const uint32_t o_sub = (uint32_t)__builtin_amdgcn_ds_bpermute(own, (int)sub_e);
                            if (__ballot(o_mask != 0u) == 0ull) continue;
                            reset();
                            constexpr uint32_t NF = PACKED ? 4u : 2u;   // requests per lane in flight (a record of the column layout is four loads)
#pragma unroll
                            for (uint32_t b0 = 0; b0 < RPL; b0 += 4u * NF) {
                                if (__ballot(((o_mask >> (RPL - 4u * NF - b0)) & ((1u << (4u * NF)) - 1u)) != 0u) == 0ull) continue;
                                u32x4 rq[NF];
#pragma unroll
                                for (uint32_t j = 0; j < NF; ++j) {
                                    const uint32_t i = b0 + 4u * j + c4;
                                    rq[j] = load_record(o_row0 + i, ((o_mask >> (RPL - 1u - i)) & 1u) != 0u);   // (mask bit RPL - 1 - i = row i: file order)
                                }
#pragma unroll
                                for (uint32_t j = 0; j < NF; ++j) {
                                    const uint32_t i = b0 + 4u * j + c4;
                                    take_record(rq[j], o_sub + i, ((o_mask >> (RPL - 1u - i)) & 1u) != 0u);
                                }
                            }
                            merge(std::integral_constant<int, 0xB1>());
                            merge(std::integral_constant<int, 0x4E>());
                            if (LPQ >= 2) merge(std::integral_constant<int, 0x141>());
                            if (LPQ >= 4) merge(std::integral_constant<int, 0x140>());
                            deliver((int)(((own0 & 15u) * 4u) * 4u), mine && (own0 >> 4) == g);
                        }
                    } else {
                        // fewer top rows than that (or queries of 8 / 16 scanning lanes): every lane fetches the side records of
                        // ITS OWN top rows only, four requests in flight, and reduces them as they come; the lanes of a query then merge
                        uint32_t m = mask;
                        while (__ballot(m != 0u)) {
                            u32x4 rq[4];
                            uint32_t gp[4];
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const bool on = m != 0u;
                                const uint32_t hb = on ? 31u - (uint32_t)__builtin_clz(m) : 0u, i = RPL - 1u - hb;   // (mask bit RPL - 1 - i = row i: file order)
                                m = on ? (m & ~(1u << hb)) : 0u;
                                gp[j] = on ? sub_e + i : 0xFFFFFFFFu;
                                rq[j] = load_record(row0 + i, on);
                            }
#pragma unroll
                            for (int j = 0; j < 4; ++j) take_record(rq[j], gp[j], gp[j] != 0xFFFFFFFFu);
                        }
                        if (LPQ >= 2) merge(std::integral_constant<int, 0xB1>());
                        if (LPQ >= 4) merge(std::integral_constant<int, 0x4E>());
                        if (LPQ >= 8) merge(std::integral_constant<int, 0x141>());
                        if (LPQ >= 16) merge(std::integral_constant<int, 0x140>());
                        deliver((int)(own0 * 4u), mine);
                    }
                    {
                        const uint32_t f_gk = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(own0 * 4u), (int)gk);
                        dn_k = (mine && dn_flag == 1u) ? f_gk : dn_k;   // (a query is reduced in one step only: dn_flag is this step's)
                    }
                    if (sub == 0 && gk != 0u) L.meta[qi] = META_DENSE;
                    if (BLU_PRIO_DENSE) __builtin_amdgcn_s_setprio(0);
                    continue;
                }
                uint32_t idx = rbase + incl - c;                      // list slot of this lane's first top row (file order)
                if (!fits) {
                    // The list is full.  The queries of this step whose top rows still fit are taken (slots go in query
                    // order, so they are the ones before the first lane that runs past the end); the rest of the task
                    // comes in the next round, which reads its rows again — their chunks were given back above, so the ring
                    // starts over at the first row of the first query left.
                    const uint32_t lane_o = (uint32_t)__builtin_ctzll(__ballot(idx + c > CAP));
                    const uint32_t qo = qb + lane_o / LPQ;            // first query that does not fit
                    const bool taken = qi < qo;
                    if (sub == 0) L.meta[qi] = taken ? (idx | (gk << 16)) : META_SLOW;
                    if (taken && c) { L.rec[idx].x = mask << (32u - RPL); L.rec[idx].y = DESC_WORD1(RPL, row0, sub_e); }
                    fill = (uint32_t)rl((int)idx, (int)((qo - qb) * LPQ));   // where the first query left would have started
                    stop_q = qo;
                    wait_vmcnt(0u);
                    ring_head = ring_landed = ring_tail = (uint32_t)((vbase + (uint32_t)rl((int)seg_x, (int)qo)) >> 8);
                    break;
                }
                if (sub == 0) L.meta[qi] = idx | (gk << 16);
                fill = p3 + k3;
                // a lane with top rows leaves ONE word in the list, at the slot of its first top row: which of its rows are top
                // rows (RPL bits), its first row relative to the task (13 bits: a ring task has at most 64 x 128 rows) and its
                // position in the segment / RPL; the slots in between stay 0 and gather_list works the entries out
                static_assert((RPL == 16u || RPL == 32u) && SHORT_SEG <= 128u, "descriptor word of a lane");
                if (c) { L.rec[idx].x = mask << (32u - RPL); L.rec[idx].y = DESC_WORD1(RPL, row0, sub_e); }
            }
        };
        // The list entries of a ring round: lane e of a 64-entry chunk finds the lane descriptor its entry belongs to (the last
        // non-zero word at or before slot e: a max-scan over the chunk, carried from chunk to chunk), picks the row out of
        // the descriptor's bit mask, and requests the 16 other bytes of that row — every load of the round in flight before
        // the first is used.  `between` runs while they travel.
        auto gather_list = [&](auto&& between) {
            constexpr int NG = (CAP + WAVE - 1) / WAVE;
            u32x4 g[NG];
            uint32_t ghi[NG], gpos[NG], ghint[NG];
            uint32_t carry = 0;                                        // slot of the last descriptor seen so far (slot 0 always holds one)
#pragma unroll
            for (int u = 0; u < NG; ++u) {
                g[u] = u32x4{0u, 0u, 0u, 0u}; ghi[u] = 0u; gpos[u] = 0u; ghint[u] = 0u;
                if ((uint32_t)u * WAVE >= fill_ring) continue;             // (wave-uniform)
                const uint32_t idx = (uint32_t)u * WAVE + (uint32_t)lane;
                const bool valid = idx < fill_ring;
                const uint32_t d0 = L.rec[idx < CAP ? idx : 0u].x;
                int s = (valid && d0 != 0u) ? (int)idx : 0;           // inclusive max-scan: slot of the descriptor that owns entry idx
                s = imax(s, dpp<0x111>(s));
                s = imax(s, dpp<0x112>(s));
                s = imax(s, dpp<0x114>(s));
                s = imax(s, dpp<0x118>(s));
                s = imax(s, __builtin_amdgcn_update_dpp(0, s, 0x142, 0xA, 0xF, false));   // row_bcast:15 into rows 1 and 3
                s = imax(s, __builtin_amdgcn_update_dpp(0, s, 0x143, 0xC, 0xF, false));   // row_bcast:31 into rows 2 and 3
                s = imax(s, (int)carry);
                carry = (uint32_t)rl(s, 63);
                const uint4 dd = L.rec[s];                          // mask | (first row, position); kernel without the ring: mask | first row | position
                const uint32_t d0w = dd.x, d = dd.y;
                // the (idx - s + 1)-th set bit of the mask, counted from its top bit (= the lane's row 0)
                uint32_t r = idx - (uint32_t)s, x = scan_rpl == 32u ? d0w : d0w >> 16, i = 0;
                if (scan_rpl == 32u) {               // (wave-uniform)
                    const uint32_t h = x >> 16, ch = (uint32_t)__builtin_popcount(h);
                    const bool low = r >= ch;
                    r -= low ? ch : 0u; x = low ? (x & 0xFFFFu) : h; i += low ? 16u : 0u;
                }
                {
                    const uint32_t h = x >> 8, ch = (uint32_t)__builtin_popcount(h);
                    const bool low = r >= ch;
                    r -= low ? ch : 0u; x = low ? (x & 0xFFu) : h; i += low ? 8u : 0u;
                }
                {
                    const uint32_t h = x >> 4, ch = (uint32_t)__builtin_popcount(h);
                    const bool low = r >= ch;
                    r -= low ? ch : 0u; x = low ? (x & 0xFu) : h; i += low ? 4u : 0u;
                }
                {
                    const uint32_t h = x >> 2, ch = (uint32_t)__builtin_popcount(h);
                    const bool low = r >= ch;
                    r -= low ? ch : 0u; x = low ? (x & 0x3u) : h; i += low ? 2u : 0u;
                }
                i += (r >= (x >> 1)) ? 1u : 0u;
                const uint32_t row = (RING ? (d >> DESC_SUB_BITS) : d) + i;
                gpos[u] = (RING ? (d & ((1u << DESC_SUB_BITS) - 1u)) : dd.z) + i;
                if (PACKED) g[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_tax, valid ? row * 16u : 0xFFFFFFC0u, 0, GATHER_AUX);
                else if (WIDE) {
                    g[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_tax, valid ? row * 24u : 0xFFFFFFC0u, 0, GATHER_AUX);
                    const u32x2 p = __builtin_amdgcn_raw_buffer_load_b64(rs_tax, valid ? row * 24u + 16u : 0xFFFFFFE0u, 0, GATHER_AUX);
                    ghint[u] = g[u].y;             // (word 1: the shape hint << 17)
                    g[u].y = p.x; ghi[u] = p.y;
                } else {
                    const uint32_t o4 = valid ? row * 4u : 0xFFFFFFF0u;
                    g[u].x = __builtin_amdgcn_raw_buffer_load_b32(rs_tax, o4, 0, GATHER_AUX);
                    if (PID32) g[u].y = __builtin_amdgcn_raw_buffer_load_b32(rs_pid, o4, 0, GATHER_AUX);
                    else {
                        const u32x2 p = __builtin_amdgcn_raw_buffer_load_b64(rs_pid, valid ? row * 8u : 0xFFFFFFE0u, 0, GATHER_AUX);
                        g[u].y = p.x; ghi[u] = p.y;
                    }
                    g[u].z = __builtin_amdgcn_raw_buffer_load_b32(rs_aln, o4, 0, GATHER_AUX);
                    g[u].w = __builtin_amdgcn_raw_buffer_load_b32(rs_acc, o4, 0, GATHER_AUX);
                }
            }
            between();
            // Milli-percent layouts: the entries go to the list ready to be compared — lineage length clamped, (length,
            // perc_identity) as ONE word (length << 17 | milli-percent: the first two sort keys of
            // find_multi_taxa_consensus.rs:39-68), align_length biased to compare unsigned.  A perc_identity that does not
            // fit 17 bits (> 131 %: not BLAST output) keeps the round on the plain records.
            // f64 layouts: the same, when every perc_identity of the round is an exact milli-percent value k / 1000 below
            // 131.071 (what BLAST prints): k = round(p * 1000) is taken only if the engine's own k / 1000.0 gives p back bit for
            // bit, so the integer compares order the hits as the f64 compares would and the identity the record carries is p.
            bool ovf = false;
            uint32_t gk[NG];
#pragma unroll
            for (int u = 0; u < NG; ++u) {
                const bool valid = (uint32_t)u * WAVE + (uint32_t)lane < fill_ring;
                if constexpr (PID32) { gk[u] = g[u].y & PM_MASK; ovf |= !PACKED && valid && g[u].y >= (1u << KEY_PID_BITS); }
                else {
                    const double p = __hiloint2double((int)ghi[u], (int)g[u].y);
                    const bool in_range = p >= 0.0 && p < 131.0705;
                    const uint32_t k = in_range ? (uint32_t)(p * 1000.0 + 0.5) : 0u;
                    const double back = milli17_to_f64(k);   // (k < 2^17: in_range)
                    gk[u] = k;
                    ovf |= valid && !(in_range && __double2hiint(back) == (int)ghi[u] && __double2loint(back) == (int)g[u].y && k < BLU_KTHR_NEVER);
                }
            }
            keyed = __ballot(ovf) == 0ull && fill == fill_ring;   // (entries appended by the long pass are plain records)
            bool any_bad = false;
#pragma unroll
            for (int u = 0; u < NG; ++u) {
                const uint32_t idx = (uint32_t)u * WAVE + (uint32_t)lane;
                if (idx < fill_ring) {
                    if (keyed) {
                        const uint32_t len = umin(g[u].x >> BLU_ROW_BITS, t.max_depth);
                        any_bad |= len == 0u || (g[u].x & ROW_MASK) >= t.n_tax;
                        const uint32_t hint = (PID32 ? g[u].y : ghint[u]) >> KEY_PID_BITS;   // (the layouts with side records; else 0)
                        L.rec[idx] = make_uint4((g[u].x & ROW_MASK) | ((hint >> 8) << BLU_ROW_BITS),
                                                (len << KEYED_LEN_SHIFT) | (gk[u] << KEYED_PID_SHIFT) | (hint & 0xFFu), g[u].z ^ 0x80000000u, g[u].w);
                    } else L.rec[idx] = make_uint4(g[u].x, g[u].y, g[u].z, g[u].w);
                    L.pq[idx] = (uint16_t)gpos[u];
                    if (!PID32) L.p1[idx] = ghi[u];
                }
            }
            keyed_bad = __ballot(any_bad) != 0ull;   // (a round without one — nine in ten on C3 — skips the error tests of phase 2a)
        };
        // Phase 1 and phase 2a run in ROUNDS: a round compacts the top rows of as many pending queries as the LDS list holds
        // (the steps that do not fit are marked and come again), phase 2a reduces them, the list is reused.  With small
        // top groups (the usual case) everything fits and there is one round; with many ties per query — identical
        // database sequences — a task takes a few rounds instead of sending its queries to the worklist kernel.
        bool pend = (uint32_t)lane < nq;
        uint32_t pend_before = WAVE + 1;
        if (RING && contiguous) {                                    // this task's chunks; what the task before requested ahead stays
            ring_c0 = (uint32_t)(vbase >> 8);
            rs_ring = ring_desc(ring_c0);
            ring_end = task_nrows ? (uint32_t)((vbase + task_nrows - 1u) >> 8) + 1u - sk_total : ring_c0;
            if (pref_q0 != q0_32) ring_head = ring_landed = ring_c0;
            ring_tail = ring_c0;
        }
        STAMP(0)   // task set-up: offsets, descriptors, contiguity
        for (;;) {
        fill = 0;
        stop_q = WAVE;
        bool ring_round = false;
        bool list_round = false;   // the round's list holds lane descriptors (ring scan or flat pass): gather_list fills the entries in
        keyed = false;
        {
            const uint32_t rows = L.seg[lane].y;                    // this lane's query (0: empty, done, too long, or outside the span)
            const uint64_t live = __ballot(rows != 0u);
            first_q = live ? (uint32_t)__builtin_ctzll(live) : 0u;
#ifndef BLU_FIXED_WIDTH
            // Width of the streamed pass.  Every streamed query pays the lanes of the widest one, so a few long segments
            // among short ones (Zipf-like hit counts) are cheaper in the long pass: estimated cost in lane-steps =
            // streamed queries x lanes per query + BLU_LONG_COST per query left to the long pass; smallest wins, ties to
            // the wider pass.
            {
                const uint32_t n16 = (uint32_t)__builtin_popcountll(__ballot(rows != 0u && rows <= 16u));
                const uint32_t n32 = (uint32_t)__builtin_popcountll(__ballot(rows != 0u && rows <= 32u));
                const uint32_t n64 = (uint32_t)__builtin_popcountll(__ballot(rows != 0u && rows <= 64u));
                const uint32_t n128 = (uint32_t)__builtin_popcountll(__ballot(rows != 0u && rows <= SHORT_SEG));
                const uint32_t c16 = 4u * n16 + BLU_LONG_COST * (n128 - n16), c32 = 8u * n32 + BLU_LONG_COST * (n128 - n32);
                const uint32_t c64 = 16u * n64 + BLU_LONG_COST * (n128 - n64), c128 = 32u * n128;
                uint32_t best = c128;
                short_seg = SHORT_SEG;
                if (c64 < best) { best = c64; short_seg = 64u; }
                if (c32 < best) { best = c32; short_seg = 32u; }
                if (c16 < best) { best = c16; short_seg = 16u; }
            }
#endif
            // Kernel without the ring: the round takes the flat pass when its units (at ~80 % of the lanes, + the gather) are fewer
            // lane-steps than the cheapest width of the passes above plus their long pass
            bool flat_round = false;
            if constexpr (!RING) {
                const uint32_t n128 = (uint32_t)__builtin_popcountll(__ballot(rows != 0u && rows <= SHORT_SEG));
                const uint32_t n_over = (uint32_t)__builtin_popcountll(__ballot(rows > SHORT_SEG));
                uint32_t best;
                {
                    const uint32_t n16 = (uint32_t)__builtin_popcountll(__ballot(rows != 0u && rows <= 16u));
                    const uint32_t n32 = (uint32_t)__builtin_popcountll(__ballot(rows != 0u && rows <= 32u));
                    const uint32_t n64 = (uint32_t)__builtin_popcountll(__ballot(rows != 0u && rows <= 64u));
                    best = 32u * n128;
                    best = umin(best, 16u * n64 + BLU_LONG_COST * (n128 - n64));
                    best = umin(best, 8u * n32 + BLU_LONG_COST * (n128 - n32));
                    best = umin(best, 4u * n16 + BLU_LONG_COST * (n128 - n16));
                }
                const uint32_t units = (uint32_t)wave_sum_u32((rows + FLAT_ROWS - 1u) / FLAT_ROWS);
                flat_round = BLU_FLAT_PASS && units != 0u && (BLU_FLAT_ALWAYS || units + units / 4u + 64u < best + BLU_LONG_COST * n_over);
                if (flat_round) { short_seg = FLAT_SEG; scan_rpl = 16u; }   // (scan_rpl: the descriptor format gather_list reads)
            }
            const uint32_t longest = flat_round ? 0u : wave_max_u32(rows > short_seg ? 0u : rows);   // longest streamed segment of the task
            // (BLU_MIXED_RING: tasks that also hold longer segments through the ring, their whole chunks left out — correct
            // (full GPU suite green with it) but no gain on C5, whose steps shrink to a few queries each: off)
            ring_round = RING && contiguous && longest != 0u && (all_short ? __ballot(rows > short_seg) == 0ull : (bool)BLU_MIXED_RING);
            uint32_t lpq = 1;
            if (ring_round) {
                scan_rpl = longest > 32u ? 32u : 16u;                     // rows per lane (measured on C3: 32 -3 %; on 10-hit tables: 16 -2.5 %)
                while (lpq * scan_rpl < longest) lpq *= 2;                // lanes per query
                while ((WAVE / lpq) * longest + 256u > RING_ROWS) lpq *= 2;   // and a step's rows (+ alignment slack) inside the ring
                if (!all_short) {
                    // a task of mixed lengths: the partial chunks of the longer segments between a step's queries count too —
                    // fewer queries per step (more lanes per query) until every step's rows fit the ring, or no ring
                    if ((uint32_t)lane == 0) L.vx[nq] = task_nrows - 256u * sk_total;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    for (;;) {
                        const uint32_t qps = WAVE / lpq, qe = (uint32_t)lane + qps < nq ? (uint32_t)lane + qps : nq;
                        const uint32_t span = L.vx[qe] - seg_x;
                        if (__ballot((uint32_t)lane < nq && ((uint32_t)lane & (qps - 1u)) == 0u && span + 256u > RING_ROWS) == 0ull) break;
                        if (lpq >= BLU_MIXED_MAX_LPQ) { ring_round = false; break; }
                        lpq *= 2;
                    }
                }
            }
            list_round = ring_round || flat_round;
            if (list_round) {
#pragma unroll
                for (uint32_t u = 0; u * WAVE < CAP; ++u) { const uint32_t i = u * WAVE + (uint32_t)lane; if (i < CAP) L.rec[i].x = 0u; }   // (no entry starts here)
            }
            if (ring_round) {
                // (rows of the lanes' windows may overlap only inside one segment: every streamed segment at least a window long)
                const bool all_full = __ballot(rows != 0u && rows <= short_seg && rows < scan_rpl) == 0ull;
                if (scan_rpl == 32u) {
                    if (all_full) phase1_scan(std::integral_constant<uint32_t, 32>(), std::true_type(), lpq);
                    else phase1_scan(std::integral_constant<uint32_t, 32>(), std::false_type(), lpq);
                } else phase1_scan(std::integral_constant<uint32_t, 16>(), std::false_type(), lpq);
            }
            else if (BLU_X_SKIP_P1) {}
            else if (flat_round) phase1_flat();
            else if (longest > 32u && longest <= 64u) phase1(std::integral_constant<uint32_t, 16>(), false);   // the C3 shape, specialised
            else if (longest) phase1(longest <= 16u ? 4u : (longest <= 32u ? 8u : 32u), longest >= (PACKED ? 9u : 25u));   // two-stage steps: measured break-even (packed records: 10 hits -2.5 %, 20 hits -17 %; columns: 20 hits +3 %, 30 hits -6 %)
            // queries phase 1 did not get to (the list filled up): marked for the next round, long ones included
            if (stop_q < WAVE && rows != 0u && ((uint32_t)lane >= stop_q || rows > short_seg)) L.meta[lane] = META_SLOW;
            fill_ring = list_round ? fill : 0u;                     // (what follows appends whole records, not lane descriptors)
            const uint64_t long_mask = __ballot(rows > short_seg);
            if (long_mask && stop_q == WAVE) phase1_long(long_mask);   // after the streamed pass: it overwrites their (empty) list heads
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        STAMP(1)   // phase 1
        // Last round of the task (nothing left pending): the ring is free, so the first chunks of the NEXT task are
        // requested now and travel while this task finishes (its offsets were requested when this task began).
        const bool last_round = __ballot(pend && (my_end - my_off) != 0u && (my_end - my_off) <= MAX_TASK_SEG && in_span && (L.meta[lane] & META_SLOW)) == 0ull;
        uint32_t nxt_c0 = 0, nxt_lim = 0;
        if (RING && last_round && nx_nq != 0u) {
            // (opaque to the optimizer: otherwise what follows is hoisted out of the rounds loop to right behind the load and
            // the task would start by waiting for the next task's offsets)
            uint32_t n_off = nx_off, n_end = nx_end;
            asm volatile("" : "+v"(n_off), "+v"(n_end));
            if (n_end > n_hits32) n_end = n_hits32;
            if (n_off > n_end) n_off = n_end;
            const uint32_t nqn = nx_nq;
            const uint32_t n_start = (uint32_t)rl((int)n_off, 0);
            const bool n_span = n_off >= n_start && (n_end - n_start) <= (uint32_t)TASK_SPAN;
            const uint32_t rel_end = n_end - n_start, rel_off = n_off - n_start;
            const uint32_t prev_end = (uint32_t)__shfl_up((int)rel_end, 1);
            const bool ok = n_span && (n_end - n_off) <= SHORT_SEG && (lane == 0 || rel_off == prev_end);
            const uint32_t n_rows = (uint32_t)rl((int)rel_end, (int)nqn - 1);
            if (__ballot((uint32_t)lane < nqn && !ok) == 0ull && n_rows != 0u) {
                const uint64_t vb = (uint64_t)n_start + mis;
                nxt_c0 = (uint32_t)(vb >> 8);
                const uint32_t cend = (uint32_t)((vb + n_rows - 1u) >> 8) + 1u;
                nxt_lim = cend < nxt_c0 + RING_CHUNKS ? cend : nxt_c0 + RING_CHUNKS;
            }
        }
        auto prefetch_next = [&]() {
            if (nxt_lim > nxt_c0) {
                const auto rsn = ring_desc(nxt_c0);
                ring_dma_run(rsn, nxt_c0, nxt_c0, nxt_lim);
                ring_head = nxt_lim; ring_landed = nxt_c0; ring_tail = nxt_c0;
                pref_q0 = nx_q0;
            }
        };
        if (BLU_PRIO) __builtin_amdgcn_s_setprio(BLU_PRIO == 2 || BLU_PRIO >= 4 ? 3 : 2);
        if (list_round && !BLU_X_SKIP_GATHER) gather_list(prefetch_next);
        else prefetch_next();
        STAMP(2)   // next-task decision, gather issue + wait + list write
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

#if defined(BLU_EXPERIMENTS) && defined(BLU_X_PAD_VALU)
        {   // (timing only: BLU_X_PAD_VALU independent vector instructions per round — is the kernel bound by instruction issue?)
            uint32_t pad0 = (uint32_t)lane, pad1 = fill;
#pragma unroll
            for (int i = 0; i < BLU_X_PAD_VALU / 2; ++i) { asm volatile("v_add_u32 %0, %0, %0" : "+v"(pad0)); asm volatile("v_xor_b32 %0, %0, %0" : "+v"(pad1)); }
            asm volatile("" ::"v"(pad0), "v"(pad1));
        }
#endif
        // ---------------- phase 2a: lane = query, LDS only ----------------
        // One pass over the list entries of every query at once: the trip count is the task's largest top group, a lane
        // whose group is shorter re-reads its last entry (every update below is idempotent), so there is no divergent
        // control flow and the LDS reads of consecutive entries overlap.
        {
            const uint32_t nrows = my_end - my_off;
            const uint32_t m = L.meta[lane];
            const bool listed = pend && nrows != 0 && nrows <= MAX_TASK_SEG && in_span && !(m & (META_SLOW | META_DENSE)) && !BLU_X_SKIP_2A;
            const uint32_t first = listed ? (m & 0xFFFFu) : 0u, k = listed ? ((m >> 16) & 0x3FFu) : 0u;
            const uint32_t kmax = wave_max_u32(k);
            // parse errors in file order (find_single_query_consensus.rs:51-64), then NaN perc_identity; reference row,
            // shortest lineage, group-max pident (find_multi_taxa_consensus.rs:39-68,142-145,182-185) and the span
            // [lo, hi] of the group in the sorted lineage order
            uint32_t err = 0, err_pos = 0, nan_pos = 0xFFFFFFFFu;
            uint32_t b_len = 0, b_acc = 0, lo = 0xFFFFFFFFu, hi = 0, l_minlen = 0xFFFFFFFFu, l_row = 0, l_pos = 0, l_hint = 0;
            int b_aln = 0;
            PK b_pid = 0, l_maxpid = 0;   // fold(0.0, max): find_multi_taxa_consensus.rs:182-185
            const uint32_t last_e = k ? k - 1u : 0u;
            uint4 nx_rec = make_uint4(0, 0, 0, 0);
            uint32_t nx_pq = 0;
            uint32_t l_keyed = 0;                        // 1: the lane's result comes from comparison-ready entries (identities = exact milli-percent)
            if (!keyed) { nx_rec = L.rec[first]; nx_pq = L.pq[first]; }   // (generic loop: entry e + 1 is read while entry e is worked on)
            if (keyed) {
                l_keyed = 1u;
                uint64_t BK = 0;                         // best (length, perc_identity, align_length) so far
                uint32_t kmin = 0xFFFFFFFFu, pmax = 0;
                uint32_t l_idx = first;                  // list slot of the entry taken last: its position, row and shape hint are read after the loop
                auto step = [&](const auto check_c, const uint4 x, const uint32_t xidx, const bool is_first) {
                    const uint32_t pos = x.x & ROW_MASK;
                    if constexpr (decltype(check_c)::value) {   // (only in a round that holds a row without a lineage: parse errors in file order)
                        const bool unmatched = pos >= t.n_tax, bad = x.y < (1u << KEYED_LEN_SHIFT);   // (lineage length 0)
                        const bool first_err = (err == 0) & (unmatched | bad);
                        err = first_err ? (unmatched ? (uint32_t)BLU_ST_ERR_UNMATCHED_TAXID : (uint32_t)BLU_ST_ERR_BAD_LINEAGE) : err;
                        err_pos = first_err ? (uint32_t)L.pq[xidx] : err_pos;
                    }
                    kmin = umin(kmin, x.y);
                    lo = umin(lo, pos);
                    hi = pos > hi ? pos : hi;
                    const uint32_t ky = x.y & ~0xFFu;      // (length, perc_identity): the hint bits do not order anything
                    const uint32_t pm = (x.y >> KEYED_PID_SHIFT) & PM_MASK;
                    pmax = pm > pmax ? pm : pmax;
                    const uint64_t K = ((uint64_t)ky << 32) | x.z;
                    const bool gt = (K > BK) | ((K == BK) & (x.w > b_acc)), eq = (K == BK) & (x.w == b_acc);
                    const bool take = is_first | (STRAT == BLU_RELAXED ? (gt | eq) : !(gt | eq));
                    BK = take ? K : BK;
                    b_acc = take ? x.w : b_acc;
                    l_idx = take ? xidx : l_idx;
                };
                // four entries per trip, their LDS reads issued together: one read latency per four entries
                auto reduce = [&](const auto check_c) {
                    for (uint32_t e = 0; e < kmax; e += 4) {
                        uint4 x[4];
                        uint32_t xi[4];
#pragma unroll
                        for (uint32_t j = 0; j < 4; ++j) {
                            xi[j] = first + (e + j < k ? e + j : last_e);
                            x[j] = L.rec[xi[j]];
                        }
#pragma unroll
                        for (uint32_t j = 0; j < 4; ++j) step(check_c, x[j], xi[j], e == 0 && j == 0);
                    }
                };
                if (keyed_bad) reduce(std::true_type()); else reduce(std::false_type());
                const uint32_t l_x = L.rec[l_idx].x, l_y = L.rec[l_idx].y;
                l_pos = L.pq[l_idx];
                const uint32_t k1 = (uint32_t)(BK >> 32);
                b_len = k1 >> KEYED_LEN_SHIFT;
                l_minlen = kmin >> KEYED_LEN_SHIFT;
                l_row = l_x & ROW_MASK;
                l_hint = ((l_x >> BLU_ROW_BITS) << 8) | (l_y & 0xFFu);
                if constexpr (PID32) { b_pid = (k1 >> KEYED_PID_SHIFT) & PM_MASK; l_maxpid = pmax; }
                else { b_pid = milli17_to_f64((k1 >> KEYED_PID_SHIFT) & PM_MASK); l_maxpid = milli17_to_f64(pmax); }   // (the doubles they came from, bit for bit)
            } else {
                for (uint32_t e = 0; e < kmax; ++e) {
                    const uint4 x = nx_rec;              // {row id, pident, align_len, accession rank}
                    const uint32_t xpos = nx_pq;
                    const uint32_t idx = first + (e < k ? e : last_e), idn = first + (e + 1u < k ? e + 1u : last_e);
                    nx_rec = L.rec[idn];
                    nx_pq = L.pq[idn];
                    const uint32_t len = umin(x.x >> BLU_ROW_BITS, t.max_depth), pos = x.x & ROW_MASK;
                    const bool unmatched = pos >= t.n_tax, bad = (x.x >> BLU_ROW_BITS) == 0;
                    const bool first_err = (err == 0) & (unmatched | bad);
                    err = first_err ? (unmatched ? (uint32_t)BLU_ST_ERR_UNMATCHED_TAXID : (uint32_t)BLU_ST_ERR_BAD_LINEAGE) : err;
                    err_pos = first_err ? xpos : err_pos;
                    PK xpid;
                    if constexpr (PID32) xpid = PACKED ? (x.y & PM_MASK) : x.y;   // (packed: the bits above are the shape hint)
                    else {
                        xpid = __hiloint2double((int)L.p1[idx], (int)x.y);
                        nan_pos = (nan_pos == 0xFFFFFFFFu && xpid != xpid) ? xpos : nan_pos;
                    }
                    l_minlen = umin(l_minlen, len);
                    lo = umin(lo, pos);
                    hi = pos > hi ? pos : hi;
                    l_maxpid = xpid > l_maxpid ? xpid : l_maxpid;
                    const bool take = (e == 0) | key_better<STRAT, PK>(len, xpid, (int)x.z, x.w, b_len, b_pid, b_aln, b_acc);
                    b_len = take ? len : b_len;
                    b_pid = take ? xpid : b_pid;
                    b_aln = take ? (int)x.z : b_aln;
                    b_acc = take ? x.w : b_acc;
                    l_row = take ? pos : l_row;
                    l_pos = take ? xpos : l_pos;
                    if (PACKED) l_hint = take ? (x.y >> KEY_PID_BITS) : l_hint;
                }
            }
            if (pend) {
                bool done = true;
                if (nrows == 0) { st_code = BLU_ST_NO_HITS; st_ref = 0xFFFFFFFFu; rec_kind = 2; }   // mod.rs:107-113
                else if (nrows > MAX_TASK_SEG || !in_span) { if (!BLU_X_SKIP_PUSH) wl_push(wl_q, wl_cnt, (uint32_t)q); }
                else if (m & META_SLOW) done = false;   // its step did not fit the list this round: again in the next one
                else if (dn_flag == 1) mode = dn_k == 1 ? 2u : 0u;                                    // reduced by a dense step
                else if (dn_flag == 2) { st_code = dn_err; st_ref = row0 + dn_pos; rec_kind = 2; }
                else if (dn_flag == 3) wl_push(wl_q, wl_cnt, (uint32_t)q);
                else if (BLU_X_SKIP_2A) { mode = 2; r_row = L.rec[m & 0xFF].x & ROW_MASK; r_len = 5; minlen = 5; }
                else if (err) { st_code = err; st_ref = row0 + err_pos; rec_kind = 2; }
                else if (!PID32 && nan_pos != 0xFFFFFFFFu) { st_code = BLU_ST_ERR_BAD_PIDENT; st_ref = row0 + nan_pos; rec_kind = 2; }
                else {
                    r_len = b_len | (l_hint << 8) | (l_keyed << 30); r_pid = b_pid; r_row = l_row; r_pos = l_pos; minlen = l_minlen; max_pid = l_maxpid;
                    mode = k == 1 ? 2u : 0u;
                    g_lo = lo; g_hi = hi;    // span of the group in sorted order: phase 2c turns it into the shared levels
                }
                if (done) { pend = false; L.seg[lane].y = 0u; }   // later rounds skip it
            }
        }
        STAMP(3)   // phase 2a
        const uint32_t pend_now = (uint32_t)__builtin_popcountll(__ballot(pend));
        if (pend_now == 0 || pend_now >= pend_before) break;      // all reduced, or a round without progress
        pend_before = pend_now;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        }
        if (pend) wl_push(wl_q, wl_cnt, (uint32_t)q);   // a single step larger than the whole list

        // ---------------- phase 2c: lane = query, cutoff tests and the record ----------------
        if (BLU_X_SKIP_2C) { if (mode != 3) { st_code = mode; st_ref = r_row + minlen + r_pos + r_len + g_lo + g_hi; rec_kind = 2; } }
        else if (mode != 3) {
            const bool single = mode == 2;
            rec_kind = 1;
            const uint32_t r_hint = (r_len >> 8) & ((1u << BLU_HINT_BITS) - 1u);
            const bool r_keyed = (r_len >> 30) != 0u;   // f64 layouts: the identities of this query's top rows are exact milli-percent values
            r_len &= 0xFFu;
            // The reference row: header, neighbour run lengths of 20 levels and the node ids in one 128-byte line (up to 20
            // levels), read with eight 16-byte loads issued back to back: one memory request.  (Reading the node id
            // later, after the codes lookup, fetched the line a second time for half of the queries: the stream had
            // pushed it out of L2 in between.)
            // Levels shared by the whole group (find_multi_taxa_consensus.rs:137-180): every row agrees with the reference row on
            // exactly the levels all rows of the span [lo, hi] share, and the scan never looks past the shortest lineage.
            // Seen from the reference row r that is the number of levels whose run reaches dl = r - lo rows to the left and
            // dh = hi - r rows to the right: 20 byte compares on the row that is read anyway, exact below 127 rows either
            // side.  A wider group takes the wide-node tables (two L2 lookups, the first requested here, IN FRONT of the
            // reference row: vmcnt retires in issue order, so a lookup requested behind the row could not be used before the row
            // has arrived) — or, where those do not reach, the range-minimum tables, requested together with the row.
            const bool spread = !single && g_lo < g_hi && !BLU_X_SKIP_RUNLEN;
            const uint32_t dl = r_row - g_lo, dh = g_hi - r_row;     // lo <= reference row <= hi
#ifdef BLU_X_NO_WIDE
            const bool wide = false;   // (timing only: wrong records for wide groups)
#else
            const bool wide = spread && (dl > BLU_ROW_RUN_MAX || dh > BLU_ROW_RUN_MAX);   // saturated run lengths: not decidable from the row
#endif
            const bool wide_tab = wide && t.wblk != nullptr && !BLU_WIDE_RMQ;
            // (requesting the entry at the end of phase 2a instead, a hundred instructions earlier, changed nothing: 0.9866 vs 0.9863 ms)
            uint2 wentry = make_uint2(0u, 0u);
            if (wide_tab) wentry = t.wblk[g_lo >> BLU_WBLK_SHIFT];
            const uint32_t* ref = t.lin + (uint64_t)r_row * t.stride;   // sorted order: row index = pos
            const uint4* ref4 = reinterpret_cast<const uint4*>(ref);
            constexpr bool NODE_RELOAD = ((BLU_NODE_RELOAD_LAYOUTS >> LAYOUT) & 1u) != 0u || (!RING && BLU_NODE_RELOAD_NORING) ||
                                         (STRAT == BLU_CAUTIOUS && BLU_NODE_RELOAD_CAUTIOUS);
            // node ids of the first NID_REGS levels stay in registers (words of the row that are loaded anyway); when a deeper level
            // is the reported one its node id is read back from the row's line
            constexpr uint32_t NID_REGS = NODE_RELOAD ? 0u : (uint32_t)BLU_NID_REGS;
            static_assert(NID_REGS <= 20u, "a 128-byte row holds 20 node ids");
            uint4 w[8];
#if BLU_REF_NT
            {
                const u32x4* refv = reinterpret_cast<const u32x4*>(ref);
#pragma unroll
                for (int k = 0; k < 8; ++k) { const u32x4 x = __builtin_nontemporal_load(refv + k); w[k] = make_uint4(x.x, x.y, x.z, x.w); }
            }
#elif defined(BLU_REF_AUX)
            {   // (experiment: cache-policy bits on the reference-row loads; tables under 4 GB only)
                const auto rs_lin = __builtin_amdgcn_make_buffer_rsrc((void*)t.lin, 0, (uint32_t)(t.n_tax * t.stride * 4u), 0x00020000);
#pragma unroll
                for (int k = 0; k < 8; ++k) { const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(rs_lin, r_row * (t.stride * 4u) + 16u * k, 0, BLU_REF_AUX); w[k] = make_uint4(x.x, x.y, x.z, x.w); }
            }
#else
#pragma unroll
            for (int k = 0; k < (NODE_RELOAD ? 3 : (int)((BLU_ROW_NODE_BASE + NID_REGS + 3u) / 4u)); ++k) w[k] = ref4[k];
#endif
            // What the finalisation needs per LEVEL depends on the row's shape only (TaxDev::kthr: threshold, rank code and
            // max-allowed-rank bit in one word per level).  In the packed layout the side record of the reference hit carries
            // the shape as a hint, so those words are requested NOW, together with the row, instead of after it: one memory
            // round trip on the task's critical path instead of two.  The hint is checked against the row's header below; a
            // lane whose hint is missing or wrong asks again with the shape the row gives.
            uint4 ck[4] = {};
            uint32_t shape_req = 0xFFFFFFFFu;
            if constexpr (PACKED || WIDE) {
                if (r_hint) {   // (no hint — more shapes than the hint has room for, a record put together by hand, a dense step: below)
                    shape_req = umin(r_hint - 1u, t.n_shapes - 1u);
                    const uint4* kg = reinterpret_cast<const uint4*>(t.kthr + (uint64_t)shape_req * t.cstride);
#pragma unroll
                    for (int k = 0; k < 4; ++k) ck[k] = kg[k];
                }
            }
            uint32_t d_tab = 0;
            if (wide_tab) {
                // the chain of the deepest wide node around lo (row 0 of the table: no such node -> nothing shared)
                const uint32_t wn = wide_node(wentry, g_lo);
                const uint4* ch = reinterpret_cast<const uint4*>(t.wchain + (uint64_t)(wn == 0xFFFFFFFFu ? 0u : wn) * BLU_WCHAIN);
                const uint4 c0 = ch[0], c1 = ch[1], c2 = ch[2];
                d_tab = chain_count4(c0, g_hi) + chain_count4(c1, g_hi) + chain_count4(c2, g_hi);
                if (t.wide_levels > BLU_WCHAIN_FIRST && d_tab == BLU_WCHAIN_FIRST) {   // (a chain deeper than 12 levels that holds hi that far)
                    d_tab += chain_count4(ch[3], g_hi);
                    if (t.wide_levels > BLU_WCHAIN && d_tab == BLU_WCHAIN) {           // (deep taxonomies: levels 16 .. 31)
                        const uint4* cg = reinterpret_cast<const uint4*>(t.wchain_hi + (uint64_t)wn * BLU_WCHAIN);
                        d_tab += chain_count4(cg[0], g_hi) + chain_count4(cg[1], g_hi) + chain_count4(cg[2], g_hi) + chain_count4(cg[3], g_hi);
                    }
                }
                if (wn == 0xFFFFFFFFu) d_tab = shared_levels(t, g_lo, g_hi);
            } else if (wide) d_tab = shared_levels(t, g_lo, g_hi);
            if (BLU_PRIO && BLU_PRIO != 3) __builtin_amdgcn_s_setprio(BLU_PRIO == 4 || BLU_PRIO == 6 ? 1 : (BLU_PRIO == 5 ? 2 : 0));
            STAMP_DRAIN
            STAMP(4)   // reference rows arrive
            r_hdr = w[0].x;   // (requesting it back in phase 2a costs a second fetch: the line leaves L2 in between)
            // its length field equals the id's for a well-formed id and bounds the loops for a corrupt one
            const uint32_t len_ref = umin(r_len, r_hdr & 0xFF);
            d = minlen;
            if (wide) d = umin(minlen, d_tab);
            else if (spread) {
                // bytes 0x80 | min(run, 127), levels 0 .. 19: left runs in words 1 .. 5, right runs in words 6 .. 10 (levels the row
                // does not have hold 0x80).  byte - dl keeps bit 7 exactly when run >= dl, and no byte borrows from its neighbour.
                const uint32_t ca = dl * 0x01010101u, cb = dh * 0x01010101u, H = 0x80808080u;
                uint32_t cnt = 0;
                cnt += (uint32_t)__builtin_popcount((w[0].y - ca) & (w[1].z - cb) & H);
                cnt += (uint32_t)__builtin_popcount((w[0].z - ca) & (w[1].w - cb) & H);
                cnt += (uint32_t)__builtin_popcount((w[0].w - ca) & (w[2].x - cb) & H);
                cnt += (uint32_t)__builtin_popcount((w[1].x - ca) & (w[2].y - cb) & H);
                cnt += (uint32_t)__builtin_popcount((w[1].y - ca) & (w[2].z - cb) & H);
                if (cnt >= BLU_ROW_IV_LEVELS && minlen > BLU_ROW_IV_LEVELS) d = umin(minlen, shared_levels(t, g_lo, g_hi));   // agreement deeper than the row's run lengths
                else d = umin(minlen, cnt);
            }
            STAMP(5)   // shared levels from the run lengths (or the RMQ tables)
            const bool agree = single | (d >= minlen);
            if (!agree && d == 0) { st_code = BLU_ST_ERR_ROOT_DISAGREE; st_ref = row0 + r_pos; rec_kind = 2; }   // `index - 1` underflow (:181)
            else {
                const uint32_t shape = umin(r_hdr >> 8, t.n_shapes - 1u);
                const uint32_t* codes = t.codes + (uint64_t)shape * t.cstride;
                const uint4* codes4 = reinterpret_cast<const uint4*>(codes);
                const uint32_t* lvl = t.kthr + (uint64_t)shape * t.cstride;
                const uint4* lvl4 = reinterpret_cast<const uint4*>(lvl);
                const uint32_t b = single ? len_ref : (agree ? minlen - 1 : d - 1);
                double ident;   // the one f64 the cutoff tests need / the record carries
                if constexpr (PACKED) ident = milli17_to_f64(((single | agree) ? r_pid : max_pid) & PM_MASK);   // (17-bit identities: no division)
                else ident = pid_f64<PID32>((single | agree) ? r_pid : max_pid);
                // Milli-percent layouts: `fl(k / 1000) >= c` is monotone in k, so every cutoff c has a smallest k that passes
                // it (taxonomy.cpp: kthr, 17 bits, + 1 bit "fl(k / 1000) == c", i.e. `>` needs one more) and the level tests
                // are integer compares of the query's milli-percent identity — no cutoff value is read.  Identities of
                // 131.071 % and more (not BLAST output; the packed layout cannot hold them) take the f64 tests.
                uint32_t ident_k = 0;
                if constexpr (PID32) ident_k = (single | agree) ? r_pid : max_pid;
                else ident_k = r_keyed ? (uint32_t)(ident * 1000.0 + 0.5) : BLU_KTHR_NEVER;   // (exact: ident is k / 1000.0 for that k)
                if constexpr (PACKED) ident_k = umin(ident_k, BLU_KTHR_NEVER - 1u);   // (records put together by hand with the one value blu_hits_pack refuses)
                const bool by_k = PACKED || __ballot(ident_k >= BLU_KTHR_NEVER) == 0ull;
                // linnaean_ranks.rs:174-212 + build_blast_consensus_identity.rs:67-82
                uint64_t F = 0, A = 0;
                uint32_t mar_level = BLU_NONE_U8, nF = 0;
                uint4 c[4] = {};
                // the per-lane table of the level words / codes words of the first 16 levels: the list area of the wave (dead
                // since phase 2a; the records are staged there afterwards), 16 words per lane, so that "the word of level j"
                // with j different in every lane is one LDS read instead of a 16-way select chain over the registers
                uint32_t* const stage = reinterpret_cast<uint32_t*>(&L.rec[0]) + (uint32_t)lane * 16u;
                {
                  if (by_k) {
                    if (!(PACKED || WIDE) || shape != shape_req) {       // no hint, or not the row's shape: the words of the shape the row gives
#pragma unroll
                        for (int k = 0; k < 4; ++k) ck[k] = lvl4[k];
                    }
                    STAMP_DRAIN
                    STAMP(6)   // (5: run lengths / RMQ) level words arrive
                    // bit j of NGE / NGT: identity < / <= the cutoff of level j.  ident_k - threshold is negative exactly then, and
                    // v_alignbit shifts that sign bit into the mask: two instructions per test, levels taken from the deepest down
                    // so that level 0 ends in bit 0.  No dependence between levels.
                    uint64_t NGE = 0, NGT = 0;
                    auto test4 = [&](const uint4 x, uint32_t& nge, uint32_t& ngt) {
                        const uint32_t xs[4] = {x.w, x.z, x.y, x.x};
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const uint32_t dge = ident_k - (xs[i] & BLU_KTHR_NEVER);
                            nge = __builtin_amdgcn_alignbit(nge, dge, 31);
                            ngt = __builtin_amdgcn_alignbit(ngt, dge - ((xs[i] >> BLU_KTHR_BITS) & 1u), 31);
                        }
                    };
                    {
                        uint32_t nge = 0, ngt = 0;
#pragma unroll
                        for (int k = (BLU_X_SKIP_LEVELS ? -1 : 3); k >= 0; --k) test4(ck[k], nge, ngt);
                        NGE = nge & 0xFFFFu; NGT = ngt & 0xFFFFu;
                    }
                    for (uint32_t k = 4; 4 * k < len_ref; ++k) {   // lineages deeper than 16 levels
                        uint32_t nge = 0, ngt = 0;
                        test4(lvl4[k], nge, ngt);
                        NGE |= (uint64_t)(nge & 0xFu) << (4 * k); NGT |= (uint64_t)(ngt & 0xFu) << (4 * k);
                    }
                    uint4* const st4 = reinterpret_cast<uint4*>(stage);
#pragma unroll
                    for (int k = 0; k < 4; ++k) st4[k] = ck[k];
                    const uint64_t lenmask = len_ref >= 64u ? ~0ull : ((1ull << len_ref) - 1ull);
                    F = ~NGE & lenmask;                                              // filter(identity >= cutoff)
                    const uint64_t NG = NGT & lenmask;                               // skip_while(identity > cutoff): first level that stops it
                    mar_level = NG ? (uint32_t)__builtin_ctzll(NG) : (uint32_t)BLU_NONE_U8;
                    // the first (b + 1) elements of the filtered list: everything up to the set bit of rank b, if F has that many
                    A = F;
                    if ((uint32_t)__builtin_popcountll(F) > b + 1u) {
                        uint32_t r = b, pos = 0, x = (uint32_t)F;
                        { const uint32_t cl = (uint32_t)__builtin_popcount((uint32_t)F); const bool up = r >= cl; r -= up ? cl : 0u; x = up ? (uint32_t)(F >> 32) : (uint32_t)F; pos = up ? 32u : 0u; }
                        { const uint32_t cl = (uint32_t)__builtin_popcount(x & 0xFFFFu); const bool up = r >= cl; r -= up ? cl : 0u; x = up ? x >> 16 : x & 0xFFFFu; pos += up ? 16u : 0u; }
                        { const uint32_t cl = (uint32_t)__builtin_popcount(x & 0xFFu); const bool up = r >= cl; r -= up ? cl : 0u; x = up ? x >> 8 : x & 0xFFu; pos += up ? 8u : 0u; }
                        { const uint32_t cl = (uint32_t)__builtin_popcount(x & 0xFu); const bool up = r >= cl; r -= up ? cl : 0u; x = up ? x >> 4 : x & 0xFu; pos += up ? 4u : 0u; }
                        { const uint32_t cl = (uint32_t)__builtin_popcount(x & 0x3u); const bool up = r >= cl; r -= up ? cl : 0u; x = up ? x >> 2 : x & 0x3u; pos += up ? 2u : 0u; }
                        pos += (r >= (x & 1u)) ? 1u : 0u;
                        A = F & ((2ull << pos) - 1ull);
                    }
                  }
                }
                if (!by_k) {
                    auto level = [&](uint32_t j, uint32_t packed) {
                        if (j < len_ref) {
                            const uint32_t cid = packed & ((1u << BLU_PACK_CUT_BITS) - 1u);
                            double cj = s_cut[cut_in_lds ? cid : 0u];
                            asm volatile("" : "+v"(cj));     // (an LDS read of its own: see ranks_of)
                            if (!cut_in_lds) cj = t.cutvals[cid];
                            if (mar_level == BLU_NONE_U8 && !(ident > cj)) mar_level = j;   // skip_while(identity > cutoff)
                            if (ident >= cj) {                                             // filter(identity >= cutoff)
                                F |= 1ull << j;
                                if (nF <= b) A |= 1ull << j;                               // first (b + 1) elements of the filtered list
                                ++nF;
                            }
                        }
                    };
#pragma unroll
                    for (int k = 0; k < 4; ++k) c[k] = codes4[k];
                    STAMP_DRAIN
                    STAMP(6)   // (5: run lengths / RMQ) codes arrive
#pragma unroll
                    for (int k = 0; k < (BLU_X_SKIP_LEVELS ? 0 : 4); ++k) { level(4 * k, c[k].x); level(4 * k + 1, c[k].y); level(4 * k + 2, c[k].z); level(4 * k + 3, c[k].w); }
                    for (uint32_t k = 4; 4 * k < len_ref; ++k) {   // lineages deeper than 16 levels
                        const uint4 x = codes4[k]; level(4 * k, x.x); level(4 * k + 1, x.y); level(4 * k + 2, x.z); level(4 * k + 3, x.w);
                    }
                    uint4* const st4 = reinterpret_cast<uint4*>(stage);
#pragma unroll
                    for (int k = 0; k < 4; ++k) st4[k] = c[k];
                }
                // {canonical rank code, max-allowed-rank code} of level j (the lane's own words: written and read by the same lane)
                auto ranks_of = [&](const uint32_t j, uint32_t& rank, uint32_t& mar) {
                    // (two statements, not one conditional expression: that made hipcc merge the LDS and the global address into one
                    // generic pointer — flat loads, 64-bit address arithmetic per lookup and a spilled pointer)
                    uint32_t v = stage[j < 16u ? j : 15u];
                    asm volatile("" : "+v"(v));          // (the LDS read is a read of its own, whatever follows)
                    if (j >= 16u) v = by_k ? lvl[j] : codes[j];
                    if (by_k) { rank = (v >> BLU_LVL_RANK_SHIFT) & BLU_PACK_CODE_MASK; mar = ((v >> BLU_LVL_NEVER_SHIFT) & 1u) ? (uint32_t)BLU_MAR_NEVER_EQUAL : rank; }
                    else { rank = packed_rank(v); mar = packed_mar(v); }
                };
                // node id of level j: words 11..30 of the line are in registers, deeper levels are read from the row
                auto node_of = [&](uint32_t j) {
                    if (NODE_RELOAD) return ref[BLU_ROW_NODE_BASE + j];   // (read back from the row's line: see BLU_NODE_RELOAD_LAYOUTS)
                    uint32_t v = 0;
                    if (j >= NID_REGS) v = ref[BLU_ROW_NODE_BASE + j];
#pragma unroll
                    for (uint32_t i = 0; i < NID_REGS; ++i) {
                        const uint4 q4 = w[(BLU_ROW_NODE_BASE + i) / 4u];
                        const uint32_t m4 = (BLU_ROW_NODE_BASE + i) % 4u;
                        const uint32_t x = m4 == 0u ? q4.x : (m4 == 1u ? q4.y : (m4 == 2u ? q4.z : q4.w));
                        v = (j == i) ? x : v;
                    }
                    return v;
                };
                if (agree) A = F;                                                  // single hit / single-flag branch (:74-75)
                if (single) {
                    if (!A) { st_code = BLU_ST_ERR_SINGLE_BELOW_CUTOFFS; st_ref = row0 + r_pos; rec_kind = 2; }   // find_single_query_consensus.rs:113-119
                    else {
                        const uint32_t last = (uint32_t)last_lane(A);
                        uint32_t rank_last, mar_unused;
                        ranks_of(last, rank_last, mar_unused);
                        pack_result(ra, rb, BLU_ST_CONSENSUS_SINGLE, 0, last, BLU_NONE_U8, rank_last, BLU_NONE_U16,
                                     node_of(last), row0 + r_pos, A, ident);
                    }
                } else {
                    const uint32_t last = A ? (uint32_t)last_lane(A) : b;          // .last().unwrap_or(taxonomy[bean_index])
                    uint32_t flags = agree ? BLU_FLAG_AGREE : 0u, mar_code = BLU_NONE_U16;
                    uint32_t rank_last, rank_b, mar_unused;
                    ranks_of(last, rank_last, mar_unused);
                    if (mar_level != BLU_NONE_U8) {
                        uint32_t rank_unused;
                        ranks_of(mar_level, rank_unused, mar_code);
                        ranks_of(b, rank_b, mar_unused);
                        if (mar_code != rank_b) flags |= BLU_FLAG_MUTATED;         // bean.reached_rank != allowed_rank (:35-37)
                    }
                    pack_result(ra, rb, BLU_ST_CONSENSUS_MULTI, flags, b, mar_level, rank_last, mar_code,
                                 node_of(last), row0 + r_pos, A, ident);
                }
            }
        }
        STAMP(7)   // levels, record
        // what was requested ahead for the next task has to be in the ring before that task counts its own requests
        if (ring_landed < ring_head) { wait_vmcnt(0u); ring_mark_landed(ring_head); }
        STAMP(8)   // drain of the requests made ahead
        // ---------------- records: staged through LDS, stored as two fully coalesced 1 KiB rows ----------------
        // (a 32-byte record per lane straight to memory is 64 scattered 16-byte pieces per store instruction;
        // measured: 0.7 ms of a 2.7 ms launch.)  The list area is dead after phase 2a and is reused.
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        uint4* rec = L.rec;
        if (rec_kind == 2) pack_status(ra, rb, st_code, st_ref);
        rec[2 * lane] = ra;
        rec[2 * lane + 1] = rb;
        L.meta[lane] = rec_kind;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        // write-through + non-temporal (sc1 nt): a record is written once and never read by the GPU; letting the
        // lines sit dirty in L2 until the read stream evicts them one by one costs ~2x more HBM time (probe:
        // scripts/probe/pattern_probe.hip store modes 1 vs 18)
        const auto rs_out = __builtin_amdgcn_make_buffer_rsrc((void*)(out + q0), 0, nq * 32u, 0x00020000);
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const uint32_t c = (uint32_t)lane + 64u * half, qc = c >> 1;
            if (qc < nq && L.meta[qc] && !BLU_X_SKIP_STORES) {
                const uint4 v = rec[c];
                const u32x4 w = {v.x, v.y, v.z, v.w};
                __builtin_amdgcn_raw_buffer_store_b128(w, rs_out, c * 16u, 0, RECORD_AUX);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        STAMP(9)   // record staging and stores
    }
#if defined(BLU_EXPERIMENTS) && defined(BLU_X_STAMPS)
    if (lane < 12 && wave < 8192) {
        uint32_t v = 0;
#pragma unroll
        for (int i = 0; i < 12; ++i) v = lane == i ? (uint32_t)st_sum[i] : v;
        g_stamps[wave * 16 + lane] = v;
    }
#endif
    // The last block to finish publishes the queue length for the worklist kernel and zeroes the two counters: a run
    // leaves them as it found them — no memset between runs, and a captured graph of the kernels can be replayed.  The length
    // also goes to a pinned host word: the next call on this table launches no worklist kernel when it was (next to) nothing
    // — a kernel boundary costs more than a C4 slice can afford — and if the queue is not empty after all, this block, the
    // last one running, works it off itself (every other block has finished and published its entries).
    __shared__ uint32_t s_drain, s_last;
    uint32_t* const s_cnt = s_lds[0].meta;   // (the queues' lengths, for the drain below: the first wave's task table is dead by now)
    // The entries this wave queued (write-through stores, wl_push) may be read by the last block of THIS kernel: every wave
    // waits for its own stores before the block's ticket — the barrier below is a workgroup-scope release and does not wait
    // for other waves' stores to have left the CU.  (An agent-scope release fence here — an L2 write-back per wave — cost
    // 0.08 ms per C3 run and 0.11 ms per C4 slice: measured and replaced by the sc1 stores.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        const uint32_t ticket = atomicAdd(work_count + 1, 1u);
        s_last = ticket == gridDim.x - 1 ? 1u : 0u;
        s_drain = 0u;
    }
    __syncthreads();
    if (s_last && wib == 0) {   // (one wave: lane = queue)
        static_assert(WL_QUEUES == WAVE, "one lane per queue");
        uint32_t* const c = work_count + WL_BASE + (uint32_t)lane * WL_STRIDE;
        const uint32_t n_s = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t n = (uint32_t)wave_sum_u32(n_s);
        __hip_atomic_store(c + 1, no_long ? 0u : n_s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_cnt[lane] = n_s;
        if (lane == 0) {
            if (host_len) __hip_atomic_store(host_len, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(work_count + 2, no_long ? 0u : n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(work_count + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_drain = no_long ? n : 0u;
        }
    }
    if (no_long) {
        __syncthreads();
        const uint32_t n = s_drain;
        if (n) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // the entries other blocks queued
            uint32_t* const slot = reinterpret_cast<uint32_t*>(&L.rec[0]);
            static_assert(sizeof(L.rec) >= SLOT_CAP * sizeof(uint32_t), "the list area holds the slots of a worklist query");
            const uint32_t incl = wave_incl_scan_u32(s_cnt[lane]);
            for (uint32_t wi = (uint32_t)wib; wi < n; wi += WAVES_T)
                consensus_of_long_query<STRAT, LAYOUT>(h, t, out, (uint64_t)wl_entry(worklist, wl_cap, incl, wi), slot, lane);
        }
    }
}

// ===============================================================================
// Kernel B: worklist queries (segments of any length), one wave per query.
// Lane-local running state is updated with selects only (no data-dependent
// branches around the updates).
// ===============================================================================
template <int STRAT>
__device__ __forceinline__ int select_reference(bool valid, uint32_t len, uint32_t len_ext, double pid, int aln,
                                                uint32_t acc, uint32_t pos) {
    const int lane = lane_id();
    uint64_t cand = __ballot(valid && len == len_ext);
    if (__builtin_popcountll(cand) > 1) {
        bool in = (cand >> lane) & 1;
        const double e = STRAT == BLU_RELAXED ? wave_max_f64(in ? pid : -__builtin_huge_val())
                                              : wave_min_f64(in ? pid : __builtin_huge_val());
        cand &= __ballot(in && pid == e);
    }
    if (__builtin_popcountll(cand) > 1) {
        bool in = (cand >> lane) & 1;
        const int e = STRAT == BLU_RELAXED ? wave_max_i32(in ? aln : INT_MIN) : wave_min_i32(in ? aln : INT_MAX);
        cand &= __ballot(in && aln == e);
    }
    if (__builtin_popcountll(cand) > 1) {
        bool in = (cand >> lane) & 1;
        const uint32_t e = STRAT == BLU_RELAXED ? wave_max_u32(in ? acc : 0u) : wave_min_u32(in ? acc : 0xFFFFFFFFu);
        cand &= __ballot(in && acc == e);
    }
    if (__builtin_popcountll(cand) > 1) {
        bool in = (cand >> lane) & 1;
        const uint32_t e = STRAT == BLU_RELAXED ? wave_max_u32(in ? pos : 0u) : wave_min_u32(in ? pos : 0xFFFFFFFFu);
        cand &= __ballot(in && pos == e);
    }
    return first_lane(cand);
}

// One worklist query, handled by one whole wave (any segment length).  `slot`: SLOT_CAP words of LDS of this wave.
// Called by the worklist kernel (one wave per query, 32 waves per CU) and by the tail of the stream kernel, whose waves drain
// a short queue themselves instead of leaving it to a second launch.
template <int STRAT, int LAYOUT>
__device__ __forceinline__ void consensus_of_long_query(const HitsDev& h, const TaxDev& t, blu_result* __restrict__ out, const uint64_t q,
                                                        uint32_t* const slot, const int lane) {
    constexpr bool PID32 = LAYOUT == 1 || LAYOUT == 2, PACKED = LAYOUT == 2, WIDE = LAYOUT == 3;
    if (BLU_PRIO_LONG) __builtin_amdgcn_s_setprio(0);
    uint64_t start = h.seg_off[q], end = h.seg_off[q + 1];
    if (end > h.n_hits) end = h.n_hits;
    if (start > end) start = end;
    start = uniform64(start);
    const uint32_t n = __builtin_amdgcn_readfirstlane((uint32_t)(end - start));   // n_hits < 2^32
    const int32_t* c_bs = h.bitscore + start;
    const uint32_t* c_tax = (PACKED || WIDE) ? nullptr : h.tax_row + start;
    const double* c_pid = (PID32 || WIDE) ? nullptr : h.pident + start;
    const uint32_t* c_pm = (PID32 && !PACKED) ? h.pident_milli + start : nullptr;
    const int32_t* c_aln = (PACKED || WIDE) ? nullptr : h.align_len + start;
    const uint32_t* c_acc = (PACKED || WIDE) ? nullptr : h.acc_rank + start;
    const u32x4* c_pk = PACKED ? reinterpret_cast<const u32x4*>(h.packed) + start : nullptr;   // 16-byte records
    const u32x2* c_pw = WIDE ? reinterpret_cast<const u32x2*>(h.packed64) + 3 * start : nullptr;   // 24-byte records
    // group size, errors in file order, lane-local best key / shortest lineage / max pident (filled by either path below)
    uint32_t k = 0, err_status = 0, err_row = 0;
    uint32_t have = 0, b_len = 0, b_acc = 0, b_pos = 0, b_row = 0, l_minlen = 0xFFFFFFFFu;
    uint32_t l_lo = 0xFFFFFFFFu, l_hi = 0;   // span of this lane's top rows in the sorted lineage order
    uint32_t l_nan = 0xFFFFFFFFu;             // this lane's first top row with a NaN perc_identity (f64 layout only)
    int b_aln = 0;
    double b_pid = 0.0, l_maxpid = 0.0;
    // The rows of the top group are collected in the slots (in no particular order) and their side records gathered one
    // row per lane: one memory round trip per 64 top rows wherever they sit in the segment.  A lane's running best
    // settles ties on all four keys by the row index (Relaxed: the later row, Cautious: the earlier one) — the rule
    // the stable sort + .last() / .first() of find_multi_taxa_consensus.rs:39-68 amounts to.
    uint32_t l_err_row = 0xFFFFFFFFu, l_err_kind = 0;   // this lane's first failing top row (parse_taxonomy Err, find_single_query_consensus.rs:58-60)
    uint32_t kk = 0;                                     // slots in use (wave-uniform)
    auto flush = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        for (uint32_t b = 0; b < kk; b += WAVE) {
            const bool top = b + (uint32_t)lane < kk;
            const uint32_t i = top ? slot[b + (uint32_t)lane] : 0u;
            uint32_t tax, acc;
            int aln;
            double pid;
            if (PACKED) { const u32x4 rec = c_pk[i]; tax = rec.x; pid = milli_to_f64(rec.y & PM_MASK); aln = (int)rec.z; acc = rec.w; }   // (bits 17.. of word 1: the shape hint)
            else if (WIDE) {
                const u32x2 a = c_pw[3ull * i], b = c_pw[3ull * i + 1], c = c_pw[3ull * i + 2];
                tax = a.x; aln = (int)b.x; acc = b.y; pid = __hiloint2double((int)c.y, (int)c.x);
            }
            else { tax = c_tax[i]; pid = PID32 ? milli_to_f64(c_pm[i]) : c_pid[i]; aln = c_aln[i]; acc = c_acc[i]; }
            const uint32_t pos = tax & ROW_MASK;
            const bool unmatched = top && pos >= t.n_tax;
            const uint32_t len = umin(tax >> BLU_ROW_BITS, t.max_depth);
            const bool bad = top && !unmatched && len == 0;
            const bool first_err = (unmatched | bad) && i < l_err_row;
            l_err_kind = first_err ? (bad ? (uint32_t)BLU_ST_ERR_BAD_LINEAGE : (uint32_t)BLU_ST_ERR_UNMATCHED_TAXID) : l_err_kind;
            l_err_row = first_err ? i : l_err_row;
            // Relaxed keeps the greatest key (ties on all four: the later row), Cautious the smallest (the earlier row): one strict
            // comparison chain in the strategy's own direction (a NaN compares false either way; NaN groups are refused below)
            const bool ahead = STRAT == BLU_RELAXED
                ? (len > b_len) | ((len == b_len) & ((pid > b_pid) | ((pid == b_pid) & ((aln > b_aln) | ((aln == b_aln) & (acc > b_acc))))))
                : (len < b_len) | ((len == b_len) & ((pid < b_pid) | ((pid == b_pid) & ((aln < b_aln) | ((aln == b_aln) & (acc < b_acc))))));
            const bool eq = (len == b_len) & (pid == b_pid) & (aln == b_aln) & (acc == b_acc);
            const bool better = ahead | (eq & (STRAT == BLU_RELAXED ? i > b_pos : i < b_pos));
            const bool take = top & ((have == 0) | better);
            have = top ? 1u : have;
            b_len = take ? len : b_len;
            b_pid = take ? pid : b_pid;
            b_aln = take ? aln : b_aln;
            b_acc = take ? acc : b_acc;
            b_pos = take ? i : b_pos;
            b_row = take ? pos : b_row;
            l_minlen = top ? umin(l_minlen, len) : l_minlen;
            l_lo = top ? umin(l_lo, pos) : l_lo;
            l_hi = (top && pos > l_hi) ? pos : l_hi;
            l_maxpid = (top && pid > l_maxpid) ? pid : l_maxpid;
            if (!PID32) l_nan = (top && pid != pid) ? umin(l_nan, i) : l_nan;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        kk = 0;
    };
    auto collect = [&](const bool top, const uint32_t i) {   // one 64-row group of the segment: its top rows into the slots
        const uint64_t mask = __ballot(top);
        if (!mask) return;
        if (top) slot[kk + __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u))] = i;
        const uint32_t add = (uint32_t)__builtin_popcountll(mask);
        kk += add;
        k += add;
        if (kk > SLOT_CAP - WAVE) flush();
    };
    if (n <= KEEP_ROWS) {
        // ---- a segment of up to 1024 rows: one round trip for its bit-scores (four 16-byte loads per lane), the top score
        // and the top rows come out of the registers
        const auto rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)c_bs, 0, n * 4u, 0x00020000);
        u32x4 w[KEEP_ROWS / 256];
#pragma unroll
        for (uint32_t u = 0; u < KEEP_ROWS / 256; ++u) {
            w[u] = u32x4{0u, 0u, 0u, 0u};
            if (u * 256u < n) w[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, (u * 256u + (uint32_t)lane * 4u) * 4u, 0, 0);
        }
        int m = INT_MIN;
#pragma unroll
        for (uint32_t u = 0; u < KEEP_ROWS / 256; ++u) {
            const uint32_t i0 = u * 256u + (uint32_t)lane * 4u;
            m = imax(m, i0 < n ? (int)w[u].x : INT_MIN);
            m = imax(m, i0 + 1 < n ? (int)w[u].y : INT_MIN);
            m = imax(m, i0 + 2 < n ? (int)w[u].z : INT_MIN);
            m = imax(m, i0 + 3 < n ? (int)w[u].w : INT_MIN);
        }
        const int M = wave_max_i32(m);
#pragma unroll
        for (uint32_t u = 0; u < KEEP_ROWS / 256; ++u) {
            if (u * 256u >= n) continue;      // (wave-uniform)
            const uint32_t vv[4] = {w[u].x, w[u].y, w[u].z, w[u].w};
#pragma unroll
            for (uint32_t c = 0; c < 4; ++c) {
                const uint32_t i = u * 256u + (uint32_t)lane * 4u + c;
                collect(i < n && (int)vv[c] == M, i);
            }
        }
    } else {
    // pass 1: top score
    // 16-byte loads, four per lane in flight (1024 rows per iteration); rows past the segment read as 0 from the
    // range-checked descriptor and are masked by index
    int m = INT_MIN;
    uint32_t c_first = 0, c_last = 0;   // first / last 1024-row chunk (its first row) in which this lane saw its running maximum
    u32x4 v[4] = {};   // (a segment of up to 1024 rows stays in these registers for pass 2)
    for (uint64_t sb = 0; sb < n; sb += LONG_SPAN) {   // descriptors cover LONG_SPAN rows: byte offsets stay below 2^32
        const uint32_t ns = (n - sb) < LONG_SPAN ? (uint32_t)(n - sb) : LONG_SPAN;
        const auto rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)(c_bs + sb), 0, ns * 4u, 0x00020000);
        for (uint32_t base = 0; base < ns; base += 1024) {
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, (base + u * 256 + (uint32_t)lane * 4u) * 4u, 0, 0);
            int cl = INT_MIN;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const uint32_t i0 = base + u * 256 + (uint32_t)lane * 4u;
                cl = imax(cl, i0 < ns ? (int)v[u].x : INT_MIN);
                cl = imax(cl, i0 + 1 < ns ? (int)v[u].y : INT_MIN);
                cl = imax(cl, i0 + 2 < ns ? (int)v[u].z : INT_MIN);
                cl = imax(cl, i0 + 3 < ns ? (int)v[u].w : INT_MIN);
            }
            const uint32_t cb = (uint32_t)sb + base;
            c_first = cl > m ? cb : c_first;
            c_last = cl >= m ? cb : c_last;
            m = imax(m, cl);
        }
    }
    const int M = wave_max_i32(m);
    // pass 2 only walks the chunks that can hold a top row: BLAST writes a query's hits best first, so this is
    // usually the first chunk alone (lanes that never reached M do not count; chunk starts are multiples of 1024)
    const uint32_t w_first = wave_min_u32(m == M ? c_first : 0xFFFFFFFFu);
    const uint32_t w_last = wave_max_u32(m == M ? c_last : 0u);
    // pass 2: group size, errors in file order, lane-local best key / shortest lineage / max pident
    for (uint64_t cb = w_first; cb <= w_last && cb < n; cb += 1024) {
    {
    const uint64_t sb = cb / LONG_SPAN * LONG_SPAN;
    const uint32_t base = (uint32_t)(cb - sb);
    const uint32_t ns = (n - sb) < LONG_SPAN ? (uint32_t)(n - sb) : LONG_SPAN;
    const auto rs_b = __builtin_amdgcn_make_buffer_rsrc((void*)(c_bs + sb), 0, ns * 4u, 0x00020000);
      // 1024 rows per round trip, as in pass 1 (four 16-byte loads per lane in flight); top rows are sparse, so most of
      // the sixteen 64-row groups end at the ballot.  Lane l holds rows base + 256 u + 4 l + c: a lane sees its rows
      // in file order, which is all the running selects below need (ties across lanes are settled by row index).
      if (n > 1024u) {   // (wave-uniform) a shorter segment is still in the registers pass 1 filled: one round trip less
#pragma unroll
          for (int u = 0; u < 4; ++u) v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, (base + u * 256 + (uint32_t)lane * 4u) * 4u, 0, 0);
      }
      for (int u = 0; u < 4; ++u) {
        if (base + (uint32_t)u * 256 >= ns) break;
        const u32x4 cur = u == 0 ? v[0] : (u == 1 ? v[1] : (u == 2 ? v[2] : v[3]));
        const uint32_t vv[4] = {cur.x, cur.y, cur.z, cur.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const uint32_t iloc = base + (uint32_t)u * 256 + (uint32_t)lane * 4u + (uint32_t)c;
            collect(iloc < ns && (int)vv[c] == M, (uint32_t)sb + iloc);   // (row index inside the segment)
        }
      }
    }
    }
    }   // (segments over KEEP_ROWS rows)
    if (BLU_PRIO_LONG) __builtin_amdgcn_s_setprio(2);   // (the round trips of the finalisation in front of other waves' streaming passes)
    if (kk) flush();
    {
        const uint32_t first_err = wave_min_u32(l_err_row);
        if (first_err != 0xFFFFFFFFu) {
            err_row = first_err;
            err_status = (uint32_t)rl((int)l_err_kind, first_lane(__ballot(l_err_row == first_err)));
        }
    }
    if (err_status) {
        if (lane == 0) store_status(out, q, err_status, (uint32_t)start + err_row);
        return;
    }
    // NaN pident anywhere in the top group (after the parse errors, which win): its first row in file order, gathered
    // by pass 2 (the milli-percent layouts have no NaN)
    if (!PID32) {
        const uint32_t nan_row = wave_min_u32(l_nan);
        if (nan_row != 0xFFFFFFFFu) {
            if (lane == 0) store_status(out, q, BLU_ST_ERR_BAD_PIDENT, (uint32_t)start + nan_row);
            return;
        }
    }
    if (n == 0) {
        if (lane == 0) store_status(out, q, BLU_ST_NO_HITS, 0xFFFFFFFFu);
        return;
    }
    int rlane;
    if (k == 1) rlane = first_lane(__ballot(have != 0));
    else {
        const uint32_t len_ext = STRAT == BLU_RELAXED ? wave_max_u32(have ? b_len : 0u) : wave_min_u32(have ? b_len : 0xFFFFFFFFu);
        rlane = select_reference<STRAT>(have != 0, b_len, len_ext, b_pid, b_aln, b_acc, b_pos);
    }
    const uint32_t row_ref = (uint32_t)rl((int)b_row, rlane);
    const uint32_t len_ref = (uint32_t)rl((int)b_len, rlane);

    const uint32_t pos_ref = (uint32_t)rl((int)b_pos, rlane);
    const double pid_ref = rl_f64(b_pid, rlane);
    const bool in_l = (uint32_t)lane < len_ref;
    const uint32_t lvl = in_l ? (uint32_t)lane : 0u;
    // the span of the group is known here: its range-minimum lookup travels together with the reference row
    const uint32_t minlen = wave_min_u32(l_minlen);
    const uint32_t lo = wave_min_u32(l_lo), hi = wave_max_u32(l_hi);
    const uint32_t hdr_ref = t.lin[(uint64_t)row_ref * t.stride];
    const uint32_t ref_node = t.lin[(uint64_t)row_ref * t.stride + BLU_ROW_NODE_BASE + lvl];
    uint32_t d = minlen;
    if (k != 1 && lo < hi) d = umin(minlen, shared_levels(t, lo, hi));   // levels shared by the whole top group (:137-180)
    const uint32_t shape_ref = hdr_ref >> 8;
    const uint32_t packed = t.codes[(uint64_t)shape_ref * t.cstride + lvl];
    const double cut = t.cutvals[packed & ((1u << BLU_PACK_CUT_BITS) - 1u)];
    const uint32_t codes = packed_rank(packed) | (packed_mar(packed) << 16);
    const uint32_t ref_row = (uint32_t)start + pos_ref;
    if (k == 1) {   // find_single_query_consensus.rs:74-150
        const uint64_t A = __ballot(in_l && pid_ref >= cut);
        if (A == 0) { if (lane == 0) store_status(out, q, BLU_ST_ERR_SINGLE_BELOW_CUTOFFS, ref_row); return; }
        const uint32_t last = (uint32_t)last_lane(A);
        const uint32_t ident_node = (uint32_t)rl((int)ref_node, (int)last);
        const uint32_t reached = (uint32_t)rl((int)codes, (int)last) & 0xFFFF;
        if (lane == 0) store_result(out, q, BLU_ST_CONSENSUS_SINGLE, 0, last, BLU_NONE_U8, reached, BLU_NONE_U16, ident_node, ref_row, A, pid_ref);
        return;
    }
    const bool agree = d >= minlen;
    if (!agree && d == 0) { if (lane == 0) store_status(out, q, BLU_ST_ERR_ROOT_DISAGREE, ref_row); return; }
    const uint32_t b = agree ? minlen - 1 : d - 1;
    double max_pid = 0.0;
    if (!agree) max_pid = wave_max_f64(l_maxpid);   // fold(0.0, |acc, i| if i > acc {i} else {acc})
    const double ident = agree ? pid_ref : max_pid;
    const uint64_t F = __ballot(in_l && ident >= cut);
    const uint64_t NG = __ballot(in_l && !(ident > cut));
    uint64_t A = F;
    if (!agree) {
        const uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(F >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)F, 0u));
        A = __ballot(((F >> lane) & 1) && before <= b);
    }
    const uint32_t last = A ? (uint32_t)last_lane(A) : b;
    const uint32_t ident_node = (uint32_t)rl((int)ref_node, (int)last);
    const uint32_t reached = (uint32_t)rl((int)codes, (int)last) & 0xFFFF;
    uint32_t mar_level = BLU_NONE_U8, mar_code = BLU_NONE_U16, flags = agree ? BLU_FLAG_AGREE : 0u;
    if (NG) {
        mar_level = (uint32_t)first_lane(NG);
        mar_code = ((uint32_t)rl((int)codes, (int)mar_level)) >> 16;
        if (mar_code != ((uint32_t)rl((int)codes, (int)b) & 0xFFFF)) flags |= BLU_FLAG_MUTATED;
    }
    if (lane == 0) store_result(out, q, BLU_ST_CONSENSUS_MULTI, flags, b, mar_level, reached, mar_code, ident_node, ref_row, A, ident);
}

template <int STRAT, int LAYOUT>
__global__ __launch_bounds__(BLOCK_B, BLU_B_WAVES_PER_SIMD) void blu_consensus_long_kernel(HitsDev h, TaxDev t, blu_result* __restrict__ out,
                                                                 const uint32_t* __restrict__ worklist,
                                                                 const uint32_t* __restrict__ work_count) {
    __shared__ uint32_t s_slot[BLOCK_B / WAVE][SLOT_CAP];   // rows of the top group found so far (segments kept in registers)
    uint32_t* const slot = s_slot[__builtin_amdgcn_readfirstlane(threadIdx.x / WAVE)];
    const uint32_t n_work = work_count[2];   // entries of all queues, published by the stream kernel's last block (0 when its waves drained them themselves)
    const uint32_t wave = blockIdx.x * (blockDim.x / WAVE) + __builtin_amdgcn_readfirstlane(threadIdx.x / WAVE);
    const uint32_t n_waves = gridDim.x * (blockDim.x / WAVE);
    const int lane = lane_id();
    if (wave >= n_work) return;
    const uint32_t cap = wl_capacity(h.n_queries);
    const uint32_t incl = wave_incl_scan_u32(work_count[WL_BASE + (uint32_t)lane * WL_STRIDE + 1u]);   // lane = queue: its published length
    for (uint32_t wi = wave; wi < n_work; wi += n_waves) {
        int lane_q = lane;
#if BLU_LANE_LAUNDER_LONG
        asm volatile("" : "+v"(lane_q));   // (nothing derived from the lane id is loop-invariant: see the stream kernel's task loop)
#endif
        consensus_of_long_query<STRAT, LAYOUT>(h, t, out, (uint64_t)wl_entry(worklist, cap, incl, wi), slot, lane_q);
    }
}

// ===============================================================================
// Which stream kernel runs this table: a sample of its 64-query tasks (at most 16384 of them, evenly spread) is checked
// for what the ring needs — offsets ascending, no segment over SHORT_SEG rows.  work_count[9] = 1: the kernel with the
// ring (most sampled tasks can use it), 2: the one without.  work_count[8] / [10] count the votes and are back at zero
// when the last thread has decided.  The decision also goes to a word of pinned host memory that belongs to the handle:
// the next calls on the handle read it (no synchronisation: whatever has arrived) and launch that kind alone — a kernel
// boundary costs ~20 us, 2 % of a C3 run — and every 64th call asks again.  A stale or wrong kind costs time, not
// correctness: both kinds compute the same records.
// ===============================================================================
__global__ __launch_bounds__(256) void blu_classify_tasks(const uint64_t* __restrict__ seg_off, uint64_t n_queries, uint64_t n_hits,
                                                          uint64_t n_tasks, uint64_t stride, uint32_t n_sampled, uint32_t* __restrict__ work_count,
                                                          uint32_t* __restrict__ kind_out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_sampled) return;
    const uint64_t task = (uint64_t)i * stride;
    bool ok = task < n_tasks;
    if (ok) {
        const uint64_t q0 = task * WAVE, q1 = (q0 + WAVE) < n_queries ? (q0 + WAVE) : n_queries;
        uint64_t prev = seg_off[q0];
        for (uint64_t q = q0; q < q1 && ok; ++q) {
            const uint64_t e = seg_off[q + 1];
            ok = e >= prev && e - prev <= SHORT_SEG && e <= n_hits;
            prev = e;
        }
    }
    if (ok) atomicAdd(work_count + 8, 1u);
    __threadfence();
    if (atomicAdd(work_count + 10, 1u) == n_sampled - 1u) {
        const uint32_t capable = __hip_atomic_load(work_count + 8, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t kind = 2u * capable >= n_sampled ? 1u : 2u;
        __hip_atomic_store(work_count + 9, kind, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (kind_out) __hip_atomic_store(kind_out, kind, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (pinned host memory: the next call reads it)
        __hip_atomic_store(work_count + 8, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(work_count + 10, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ===============================================================================
// launch
// ===============================================================================
static thread_local uint32_t g_grid = 0, g_block = 0;

const char* consensus_kernel_name() { return "blu_consensus_stream_kernel"; }
void consensus_last_geometry(uint32_t* grid, uint32_t* block) {
    if (grid) *grid = g_grid;
    if (block) *block = g_block;
}

template <int STRAT, int LAYOUT>
static int launch_t(const TaxDev& tax, const HitsDev& hits, blu_result* out, hipStream_t s, int num_cus,
                    uint32_t* worklist, uint32_t* work_count, uint32_t* kind_dev, uint32_t known) {
    // known: bits 0..1 = the stream kernel kind the table wants (1 ring, 2 no ring, 0 = classify on the device), bit 2 = the
    // table's last run left next to nothing for the worklist kernel: it is not launched (kernel A's last block drains the queue)
    const uint32_t known_kind = known & 3u;
    const bool no_long = (known & 4u) != 0u && (known_kind == 1u || known_kind == 2u);
    uint32_t* const host_len = kind_dev ? kind_dev + 1 : nullptr;
    const uint64_t n_tasks = (hits.n_queries + WAVE - 1) / WAVE;
    const uint32_t cus = (uint32_t)(num_cus > 0 ? num_cus : 256);
    if (known_kind != 1u && known_kind != 2u) {
        const uint64_t stride = (n_tasks + 16383) / 16384;
        const uint32_t n_sampled = (uint32_t)((n_tasks + stride - 1) / stride);
        hipLaunchKernelGGL(blu_classify_tasks, dim3((n_sampled + 255) / 256), dim3(256), 0, s, hits.seg_off, hits.n_queries, hits.n_hits, n_tasks, stride,
                           n_sampled, work_count, kind_dev);
    }
    // one block per CU of either kind (LDS: 12 waves with the ring, 16 without); with both in the stream, the kind that
    // was not picked for this table returns at once
    const uint32_t forced = ((known_kind == 1u || known_kind == 2u) ? 1u : 0u) | (no_long ? 2u : 0u);
    if (known_kind != 2u) {
        constexpr uint32_t block_r = (LAYOUT == 0 || LAYOUT == 3) ? BLOCK_F : BLOCK_A;
        const uint64_t want = BLU_TAIL_SPLIT ? (hits.n_queries + 16ull * (block_r / WAVE) - 1) / (16ull * (block_r / WAVE)) : (n_tasks + (block_r / WAVE) - 1) / (block_r / WAVE);
        const uint32_t grid = (uint32_t)(want < cus ? (want ? want : 1) : cus);
        g_grid = grid;
        g_block = block_r;
        hipLaunchKernelGGL((blu_consensus_stream_kernel<STRAT, LAYOUT, true>), dim3(grid), dim3(block_r), 0, s, hits, tax, out, worklist, work_count, forced, host_len);
    }
    if (known_kind != 1u) {
        constexpr uint32_t block_n = (LAYOUT == 0 || LAYOUT == 3) ? BLOCK_A : BLOCK_N;
        const uint64_t want = BLU_TAIL_SPLIT ? (hits.n_queries + 16ull * (block_n / WAVE) - 1) / (16ull * (block_n / WAVE)) : (n_tasks + (block_n / WAVE) - 1) / (block_n / WAVE);
        const uint32_t grid = (uint32_t)(want < cus ? (want ? want : 1) : cus);
        if (known_kind == 2u) { g_grid = grid; g_block = block_n; }
        hipLaunchKernelGGL((blu_consensus_stream_kernel<STRAT, LAYOUT, false>), dim3(grid), dim3(block_n), 0, s, hits, tax, out, worklist, work_count, forced, host_len);
    }
    // 32 waves per CU: the kernel is latency-bound per query (block size 256 / 512 / 1024: 1.11 / 1.135 / 1.14 ms on C5)
    const uint32_t grid_b = (uint32_t)(num_cus > 0 ? num_cus : 256) * (BLU_B_WAVES_PER_SIMD * 4u * WAVE / BLOCK_B);
    if (!no_long) hipLaunchKernelGGL((blu_consensus_long_kernel<STRAT, LAYOUT>), dim3(grid_b), dim3(BLOCK_B), 0, s, hits, tax, out, worklist, work_count);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("kernel launch failed: %s", hipGetErrorString(e)); return BLU_ERR_HIP; }
    return BLU_OK;
}

int launch_consensus(const TaxDev& tax, const HitsDev& hits, int strategy, blu_result* out, void* stream, int device,
                     int num_cus, uint32_t* worklist, uint32_t* work_count, uint32_t* kind_dev, uint32_t known_kind) {
    (void)device;
    if (hits.n_queries == 0) return BLU_OK;
    const int layout = hits.packed64 ? 3 : (hits.packed ? 2 : (hits.pident_milli ? 1 : 0));
    hipStream_t s = (hipStream_t)stream;
    if (strategy == BLU_RELAXED) {
        if (layout == 3) return launch_t<BLU_RELAXED, 3>(tax, hits, out, s, num_cus, worklist, work_count, kind_dev, known_kind);
        if (layout == 2) return launch_t<BLU_RELAXED, 2>(tax, hits, out, s, num_cus, worklist, work_count, kind_dev, known_kind);
        if (layout == 1) return launch_t<BLU_RELAXED, 1>(tax, hits, out, s, num_cus, worklist, work_count, kind_dev, known_kind);
        return launch_t<BLU_RELAXED, 0>(tax, hits, out, s, num_cus, worklist, work_count, kind_dev, known_kind);
    }
    if (layout == 3) return launch_t<BLU_CAUTIOUS, 3>(tax, hits, out, s, num_cus, worklist, work_count, kind_dev, known_kind);
    if (layout == 2) return launch_t<BLU_CAUTIOUS, 2>(tax, hits, out, s, num_cus, worklist, work_count, kind_dev, known_kind);
    if (layout == 1) return launch_t<BLU_CAUTIOUS, 1>(tax, hits, out, s, num_cus, worklist, work_count, kind_dev, known_kind);
    return launch_t<BLU_CAUTIOUS, 0>(tax, hits, out, s, num_cus, worklist, work_count, kind_dev, known_kind);
}

}  // namespace blu

#if defined(BLU_EXPERIMENTS) && defined(BLU_X_STAMPS)
extern "C" int blu_debug_stamps(uint32_t* dst, size_t n_words) {   // experiment builds only (scripts/stamps.py)
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(blu::g_stamps), n_words * sizeof(uint32_t), 0, hipMemcpyDeviceToHost);
}
#endif
