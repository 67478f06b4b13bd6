// Per-query taxonomic consensus on gfx950 (CDNA4): one 64-lane wavefront per
// query, persistent waves striding over the query list.
//
// What one wave computes (reference: core/src/use_cases/build_consensus_identities/
// find_single_query_consensus.rs:17-173, find_multi_taxa_consensus.rs:22-217,
// build_blast_consensus_identity.rs:9-105; restated in SURVEY §3.3):
//   1. coalesced loads of the query's segment of the five SoA columns (lane = hit)
//   2. M = max bit_score (DPP row reduction + 4 readlanes), top group = ballot(bs == M)
//   3. top lanes gather the first 16 bytes of their lineage row (len | shape, levels 0..2)
//   4. |G| == 1: lane j tests pident >= cutoff[shape][j]            -> level mask by ballot
//      |G| >  1: reference row = staged lexicographic arg-min/max over
//                (lineage_len, pident, align_len, accession, file order);
//                first disagreeing level d = first level whose mismatch ballot is non-zero;
//                lane j tests ident against cutoff[shape(R)][j]     -> F, max_allowed_rank by ballot/ctz
//   5. lane 0 stores one 32-byte record.
// Integer/compare work only: no MFMA, no LDS staging (the taxonomy rows are
// gathered through L2/MALL; cutoffs are a per-shape table, coalesced per level).
// Segments longer than 64 hits take the chunked path (three passes over the
// segment, lane-local running state, same finalisation).
#include <hip/hip_runtime.h>

#include <climits>
#include <cstdint>

#include "blu_internal.h"

namespace blu {

#define WAVE 64

// ---- cross-lane helpers -----------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

template <int CTRL>
__device__ __forceinline__ int dpp(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }

// DPP controls: quad_perm [1,0,3,2] = 0xB1, quad_perm [2,3,0,1] = 0x4E, row_ror:4 = 0x124, row_ror:8 = 0x128
#define ROW_REDUCE(x, OP)            \
    x = OP(x, dpp<0xB1>(x));         \
    x = OP(x, dpp<0x4E>(x));         \
    x = OP(x, dpp<0x124>(x));        \
    x = OP(x, dpp<0x128>(x));

__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }
__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ uint32_t umax(uint32_t a, uint32_t b) { return a > b ? a : b; }
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
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    int x = (int)(v ^ 0x80000000u);
    return (uint32_t)wave_max_i32(x) ^ 0x80000000u;
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    int x = (int)(v ^ 0x80000000u);
    return (uint32_t)wave_min_i32(x) ^ 0x80000000u;
}

template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    return __hiloint2double(dpp<CTRL>(hi), dpp<CTRL>(lo));
}
__device__ __forceinline__ double rl_f64(double v, int lane) {
    return __hiloint2double(rl(__double2hiint(v), lane), rl(__double2loint(v), lane));
}
// plain compare-select (no NaN/-0 canonicalisation): callers exclude NaN
__device__ __forceinline__ double dmax(double a, double b) { return b > a ? b : a; }
__device__ __forceinline__ double dmin(double a, double b) { return b < a ? b : a; }

__device__ __forceinline__ double wave_max_f64(double x) {
    x = dmax(x, dpp_f64<0xB1>(x));
    x = dmax(x, dpp_f64<0x4E>(x));
    x = dmax(x, dpp_f64<0x124>(x));
    x = dmax(x, dpp_f64<0x128>(x));
    return dmax(dmax(rl_f64(x, 0), rl_f64(x, 16)), dmax(rl_f64(x, 32), rl_f64(x, 48)));
}
__device__ __forceinline__ double wave_min_f64(double x) {
    x = dmin(x, dpp_f64<0xB1>(x));
    x = dmin(x, dpp_f64<0x4E>(x));
    x = dmin(x, dpp_f64<0x124>(x));
    x = dmin(x, dpp_f64<0x128>(x));
    return dmin(dmin(rl_f64(x, 0), rl_f64(x, 16)), dmin(rl_f64(x, 32), rl_f64(x, 48)));
}

__device__ __forceinline__ int first_lane(uint64_t m) { return __builtin_ctzll(m); }
__device__ __forceinline__ int last_lane(uint64_t m) { return 63 - __builtin_clzll(m); }

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

__device__ __forceinline__ void store_error(blu_result* out, uint64_t q, int lane, uint32_t status, uint32_t ref_row) {
    if (lane == 0)
        store_result(out, q, status, 0, BLU_NONE_U8, BLU_NONE_U8, BLU_NONE_U16, BLU_NONE_U16, 0xFFFFFFFFu, ref_row, 0ull, 0.0);
}

// ---- reference-row selection ---------------------------------------------------
// Stable sort by (len, pident, align_len, accession) then .first() (Cautious) /
// .last() (Relaxed)  — find_multi_taxa_consensus.rs:39-68.  Equal keys keep file
// order, so the winner among equals is the smallest (Cautious) or largest
// (Relaxed) file position.  Each stage narrows the candidate ballot; stages are
// skipped (wave-uniform branch) once a single candidate is left.
template <int STRAT>
__device__ __forceinline__ int select_reference(bool valid, uint32_t len, uint32_t len_ext, double pid, int aln,
                                                uint32_t acc, uint32_t pos, bool use_pos) {
    uint64_t cand = __ballot(valid && len == len_ext);
    if (__builtin_popcountll(cand) > 1) {
        bool in = (cand >> lane_id()) & 1;
        double e;
        if (STRAT == BLU_RELAXED) e = wave_max_f64(in ? pid : -__builtin_huge_val());
        else e = wave_min_f64(in ? pid : __builtin_huge_val());
        cand &= __ballot(in && pid == e);
        if (__builtin_popcountll(cand) > 1) {
            in = (cand >> lane_id()) & 1;
            int ea;
            if (STRAT == BLU_RELAXED) ea = wave_max_i32(in ? aln : INT_MIN);
            else ea = wave_min_i32(in ? aln : INT_MAX);
            cand &= __ballot(in && aln == ea);
            if (__builtin_popcountll(cand) > 1) {
                in = (cand >> lane_id()) & 1;
                uint32_t ec;
                if (STRAT == BLU_RELAXED) ec = wave_max_u32(in ? acc : 0u);
                else ec = wave_min_u32(in ? acc : 0xFFFFFFFFu);
                cand &= __ballot(in && acc == ec);
                if (use_pos && __builtin_popcountll(cand) > 1) {
                    in = (cand >> lane_id()) & 1;
                    uint32_t ep;
                    if (STRAT == BLU_RELAXED) ep = wave_max_u32(in ? pos : 0u);
                    else ep = wave_min_u32(in ? pos : 0xFFFFFFFFu);
                    cand &= __ballot(in && pos == ep);
                }
            }
        }
    }
    // lanes hold rows in file order on the single-chunk path (use_pos == false)
    return STRAT == BLU_RELAXED ? last_lane(cand) : first_lane(cand);
}

// ---- finalisation (lane j = level j of the reference lineage) -----------------
// build_blast_consensus_identity.rs:9-105 with InterpolatedIdentity::
// get_rank_adjusted_by_identity / get_adjusted_taxonomy_by_identity
// (linnaean_ranks.rs:174-212) evaluated as ballots.
__device__ __forceinline__ void finalize_multi(const TaxDev& t, blu_result* out, uint64_t q, int lane,
                                               uint32_t tax_ref, uint32_t len_ref, uint32_t shape_ref,
                                               uint32_t ref_node /*lane j: node of level j*/, uint32_t ref_row,
                                               double pid_ref, double max_pid, uint32_t minlen, uint32_t d) {
    (void)tax_ref;
    const bool in_l = (uint32_t)lane < len_ref;
    double cut = 0.0;
    uint32_t codes = 0;
    if (in_l) {
        cut = t.cut[(uint64_t)shape_ref * t.sc + lane];
        codes = t.codes[(uint64_t)shape_ref * t.sc + lane];
    }
    const bool agree = d >= minlen;
    if (!agree && d == 0) {  // `index - 1` underflow, find_multi_taxa_consensus.rs:181
        store_error(out, q, lane, BLU_ST_ERR_ROOT_DISAGREE, ref_row);
        return;
    }
    const uint32_t b = agree ? minlen - 1 : d - 1;
    const double ident = agree ? pid_ref : max_pid;
    const uint64_t F = __ballot(in_l && ident >= cut);          // get_adjusted_taxonomy_by_identity
    const uint64_t NG = __ballot(in_l && !(ident > cut));       // skip_while(identity > cutoff)
    uint64_t A = F;
    if (!agree) {
        // first (b + 1) ELEMENTS of the filtered list (build_blast_consensus_identity.rs:76-82)
        uint32_t before = __builtin_amdgcn_mbcnt_hi((uint32_t)(F >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)F, 0u));
        A = __ballot(((F >> lane) & 1) && before <= b);
    }
    const uint32_t last = A ? (uint32_t)last_lane(A) : b;       // adjusted.last().unwrap_or(taxonomy[bean_index])
    const uint32_t identifier = (uint32_t)rl((int)ref_node, (int)last);
    const uint32_t reached = (uint32_t)rl((int)codes, (int)last) & 0xFFFF;
    uint32_t mar_level = BLU_NONE_U8, mar_code = BLU_NONE_U16, flags = agree ? BLU_FLAG_AGREE : 0u;
    if (NG) {
        mar_level = (uint32_t)first_lane(NG);
        mar_code = ((uint32_t)rl((int)codes, (int)mar_level)) >> 16;
        const uint32_t bean_rank = (uint32_t)rl((int)codes, (int)b) & 0xFFFF;
        if (mar_code != bean_rank) flags |= BLU_FLAG_MUTATED;   // bean.reached_rank != allowed_rank (:35-37)
    }
    if (lane == 0)
        store_result(out, q, BLU_ST_CONSENSUS_MULTI, flags, b, mar_level, reached, mar_code, identifier, ref_row, A, ident);
}

// find_single_query_consensus.rs:74-150
__device__ __forceinline__ void finalize_single(const TaxDev& t, blu_result* out, uint64_t q, int lane, uint32_t tax_h,
                                                uint32_t len_h, uint32_t shape_h, uint32_t row_h, double pid_h) {
    const bool in_l = (uint32_t)lane < len_h;
    double cut = 0.0;
    uint32_t codes = 0, node = 0;
    if (in_l) {
        node = t.lin[(uint64_t)tax_h * t.stride + 1 + lane];
        cut = t.cut[(uint64_t)shape_h * t.sc + lane];
        codes = t.codes[(uint64_t)shape_h * t.sc + lane];
    }
    const uint64_t A = __ballot(in_l && pid_h >= cut);
    if (A == 0) {  // panic!("No taxonomy found for result") :113-119
        store_error(out, q, lane, BLU_ST_ERR_SINGLE_BELOW_CUTOFFS, row_h);
        return;
    }
    const uint32_t last = (uint32_t)last_lane(A);
    const uint32_t identifier = (uint32_t)rl((int)node, (int)last);
    const uint32_t reached = (uint32_t)rl((int)codes, (int)last) & 0xFFFF;
    if (lane == 0)
        store_result(out, q, BLU_ST_CONSENSUS_SINGLE, 0, last, BLU_NONE_U8, reached, BLU_NONE_U16, identifier, row_h, A, pid_h);
}

// First level (< bound) at which some valid lane's lineage differs from the
// reference lineage; `bound` when none does.  w0 = words 0..3 of the lane's own
// lineage row (header + levels 0..2), already in registers.
__device__ __forceinline__ uint32_t first_disagreement(const TaxDev& t, bool valid, uint32_t tax, uint4 w0,
                                                       uint32_t ref_node, uint32_t bound) {
    const uint32_t* row = t.lin + (uint64_t)tax * t.stride;
    uint4 w = w0;
    for (uint32_t c = 0;; ++c) {
        if (c) {
            if (4 * c - 1 >= bound) return bound;
            w = valid ? *reinterpret_cast<const uint4*>(row + 4 * c) : uint4{0, 0, 0, 0};
        }
        const uint32_t wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (c == 0 && k == 0) continue;  // header word
            const uint32_t lvl = 4 * c + k - 1;
            if (lvl >= bound) return bound;
            const uint32_t rn = (uint32_t)rl((int)ref_node, (int)lvl);
            if (__ballot(valid && wv[k] != rn)) return lvl;
        }
    }
}

// ---- single-chunk path: the whole segment sits in one wave (n <= 64) ----------
template <int STRAT>
__device__ __forceinline__ void query_small(const HitsDev& h, const TaxDev& t, blu_result* out, uint64_t q,
                                            uint64_t start, uint32_t n) {
    const int lane = lane_id();
    const bool act = (uint32_t)lane < n;
    const uint64_t row = start + (uint32_t)lane;
    int bs = INT_MIN;
    uint32_t tax = 0, acc = 0;
    int aln = 0;
    double pid = 0.0;
    if (act) {
        bs = h.bitscore[row];
        tax = h.tax_row[row];
        pid = h.pident[row];
        aln = h.align_len[row];
        acc = h.acc_rank[row];
    }
    const int M = wave_max_i32(bs);
    const bool top = act && bs == M;
    const uint64_t mask = __ballot(top);

    const bool unmatched = top && tax >= t.n_tax;  // BLU_UNMATCHED_TAXID or out of range: left-join miss
    uint4 w0 = {0, 0, 0, 0};
    if (top && !unmatched) w0 = *reinterpret_cast<const uint4*>(t.lin + (uint64_t)tax * t.stride);
    const uint32_t len = w0.x & 0xFF, shape = w0.x >> 8;
    const bool bad = top && !unmatched && len == 0;
    const uint64_t errm = __ballot(unmatched || bad);
    if (errm) {  // parse_taxonomy Err -> panic at the first failing row (find_single_query_consensus.rs:58-60)
        const int fl = first_lane(errm);
        const bool um = (__ballot(unmatched) >> fl) & 1;
        store_error(out, q, lane, um ? BLU_ST_ERR_UNMATCHED_TAXID : BLU_ST_ERR_BAD_LINEAGE, (uint32_t)(start + fl));
        return;
    }
    const uint64_t nanm = __ballot(top && pid != pid);
    if (nanm) {
        store_error(out, q, lane, BLU_ST_ERR_BAD_PIDENT, (uint32_t)(start + first_lane(nanm)));
        return;
    }
    if (__builtin_popcountll(mask) == 1) {
        const int hl = first_lane(mask);
        finalize_single(t, out, q, lane, (uint32_t)rl((int)tax, hl), (uint32_t)rl((int)len, hl),
                        (uint32_t)rl((int)shape, hl), (uint32_t)(start + hl), rl_f64(pid, hl));
        return;
    }
    // shortest lineage bounds the level scan (take_while over the length-ascending sort, :142-145);
    // Relaxed additionally needs the longest one as the first sort key
    const uint32_t minlen = wave_min_u32(top ? len : 0xFFFFFFFFu);
    uint32_t len_ext = minlen;
    if (STRAT == BLU_RELAXED) len_ext = wave_max_u32(top ? len : 0u);
    const int rlane = select_reference<STRAT>(top, len, len_ext, pid, aln, acc, 0u, false);
    const uint32_t tax_ref = (uint32_t)rl((int)tax, rlane);
    const uint32_t len_ref = (uint32_t)rl((int)len, rlane);
    const uint32_t shape_ref = (uint32_t)rl((int)shape, rlane);
    const double pid_ref = rl_f64(pid, rlane);
    uint32_t ref_node = 0;
    if ((uint32_t)lane < len_ref) ref_node = t.lin[(uint64_t)tax_ref * t.stride + 1 + lane];
    const uint32_t d = first_disagreement(t, top, tax, w0, ref_node, minlen);
    double max_pid = 0.0;
    if (d < minlen) max_pid = wave_max_f64(top ? pid : 0.0);  // fold(0.0, |acc, i| if i > acc {i} else {acc}) :182-185
    finalize_multi(t, out, q, lane, tax_ref, len_ref, shape_ref, ref_node, (uint32_t)(start + rlane), pid_ref, max_pid,
                   minlen, d);
}

// ---- chunked path: segments longer than one wave --------------------------------
template <int STRAT>
__device__ __forceinline__ bool key_better(uint32_t len, double pid, int aln, uint32_t acc, uint32_t blen, double bpid,
                                           int baln, uint32_t bacc) {
    // Relaxed keeps the LAST maximum (>=), Cautious the FIRST minimum (<); rows arrive in file order.
    if (len != blen) return STRAT == BLU_RELAXED ? len > blen : len < blen;
    if (pid != bpid) return STRAT == BLU_RELAXED ? pid > bpid : pid < bpid;
    if (aln != baln) return STRAT == BLU_RELAXED ? aln > baln : aln < baln;
    if (acc != bacc) return STRAT == BLU_RELAXED ? acc > bacc : acc < bacc;
    return STRAT == BLU_RELAXED;
}

template <int STRAT>
__device__ __noinline__ void query_large(const HitsDev& h, const TaxDev& t, blu_result* out, uint64_t q, uint64_t start,
                                         uint64_t n) {
    const int lane = lane_id();
    // pass 1: top score
    int m = INT_MIN;
    for (uint64_t i = lane; i < n; i += WAVE) m = imax(m, h.bitscore[start + i]);
    const int M = wave_max_i32(m);
    // pass 2: group size, length range, lane-local best key, max pident
    uint64_t k = 0;
    bool have = false;
    uint32_t b_len = 0, b_acc = 0, b_pos = 0, b_tax = 0, b_shape = 0;
    int b_aln = 0;
    double b_pid = 0.0, l_maxpid = 0.0;
    uint32_t l_minlen = 0xFFFFFFFFu;
    for (uint64_t base = 0; base < n; base += WAVE) {
        const uint64_t i = base + (uint32_t)lane;
        const bool act = i < n;
        const bool top = act && h.bitscore[start + i] == M;
        const uint64_t mask = __ballot(top);
        if (!mask) continue;
        k += (uint64_t)__builtin_popcountll(mask);
        uint32_t tax = 0, acc = 0, hdr = 0;
        int aln = 0;
        double pid = 0.0;
        bool unmatched = false;
        if (top) {
            tax = h.tax_row[start + i];
            pid = h.pident[start + i];
            aln = h.align_len[start + i];
            acc = h.acc_rank[start + i];
            unmatched = tax >= t.n_tax;
            if (!unmatched) hdr = t.lin[(uint64_t)tax * t.stride];
        }
        const uint32_t len = hdr & 0xFF;
        const bool bad = top && !unmatched && len == 0;
        const uint64_t errm = __ballot(unmatched || bad);
        if (errm) {
            const int fl = first_lane(errm);
            const bool um = (__ballot(unmatched) >> fl) & 1;
            store_error(out, q, lane, um ? BLU_ST_ERR_UNMATCHED_TAXID : BLU_ST_ERR_BAD_LINEAGE, (uint32_t)(start + base + fl));
            return;
        }
        const uint64_t nanm = __ballot(top && pid != pid);
        if (nanm) {
            store_error(out, q, lane, BLU_ST_ERR_BAD_PIDENT, (uint32_t)(start + base + first_lane(nanm)));
            return;
        }
        if (top) {
            l_minlen = umin(l_minlen, len);
            if (pid > l_maxpid) l_maxpid = pid;
            if (!have || key_better<STRAT>(len, pid, aln, acc, b_len, b_pid, b_aln, b_acc)) {
                have = true;
                b_len = len; b_pid = pid; b_aln = aln; b_acc = acc; b_pos = (uint32_t)i; b_tax = tax; b_shape = hdr >> 8;
            }
        }
    }
    if (k == 1) {
        const int hl = first_lane(__ballot(have));
        finalize_single(t, out, q, lane, (uint32_t)rl((int)b_tax, hl), (uint32_t)rl((int)b_len, hl),
                        (uint32_t)rl((int)b_shape, hl), (uint32_t)(start + (uint32_t)rl((int)b_pos, hl)), rl_f64(b_pid, hl));
        return;
    }
    const uint32_t minlen = wave_min_u32(l_minlen);
    uint32_t len_ext = minlen;
    if (STRAT == BLU_RELAXED) len_ext = wave_max_u32(have ? b_len : 0u);
    const int rlane = select_reference<STRAT>(have, b_len, len_ext, b_pid, b_aln, b_acc, b_pos, true);
    const uint32_t tax_ref = (uint32_t)rl((int)b_tax, rlane);
    const uint32_t len_ref = (uint32_t)rl((int)b_len, rlane);
    const uint32_t shape_ref = (uint32_t)rl((int)b_shape, rlane);
    const uint32_t pos_ref = (uint32_t)rl((int)b_pos, rlane);
    const double pid_ref = rl_f64(b_pid, rlane);
    uint32_t ref_node = 0;
    if ((uint32_t)lane < len_ref) ref_node = t.lin[(uint64_t)tax_ref * t.stride + 1 + lane];
    // pass 3: first disagreeing level over every top row; the bound shrinks as mismatches are found
    uint32_t d = minlen;
    for (uint64_t base = 0; base < n && d > 0; base += WAVE) {
        const uint64_t i = base + (uint32_t)lane;
        const bool top = i < n && h.bitscore[start + i] == M;
        if (!__ballot(top)) continue;
        uint32_t tax = 0;
        uint4 w0 = {0, 0, 0, 0};
        if (top) {
            tax = h.tax_row[start + i];
            w0 = *reinterpret_cast<const uint4*>(t.lin + (uint64_t)tax * t.stride);
        }
        d = first_disagreement(t, top, tax, w0, ref_node, d);
    }
    double max_pid = 0.0;
    if (d < minlen) max_pid = wave_max_f64(l_maxpid);
    finalize_multi(t, out, q, lane, tax_ref, len_ref, shape_ref, ref_node, (uint32_t)(start + pos_ref), pid_ref, max_pid,
                   minlen, d);
}

template <int STRAT>
__global__ __launch_bounds__(256) void blu_consensus_wave_kernel(HitsDev h, TaxDev t, blu_result* __restrict__ out) {
    const uint64_t waves_per_block = blockDim.x / WAVE;
    const uint64_t wave = (uint64_t)blockIdx.x * waves_per_block + (threadIdx.x / WAVE);
    const uint64_t n_waves = (uint64_t)gridDim.x * waves_per_block;
    for (uint64_t q = wave; q < h.n_queries; q += n_waves) {
        uint64_t start = h.seg_off[q], end = h.seg_off[q + 1];
        // defend the column reads against a corrupt offset table
        if (end > h.n_hits) end = h.n_hits;
        if (start > end) start = end;
        start = __builtin_amdgcn_readfirstlane((uint32_t)start) | ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)(start >> 32)) << 32);
        const uint64_t n = __builtin_amdgcn_readfirstlane((uint32_t)(end - start)) |
                           ((uint64_t)__builtin_amdgcn_readfirstlane((uint32_t)((end - start) >> 32)) << 32);
        if (n == 0) {
            store_error(out, q, lane_id(), BLU_ST_NO_HITS, 0xFFFFFFFFu);
        } else if (n <= WAVE) {
            query_small<STRAT>(h, t, out, q, start, (uint32_t)n);
        } else {
            query_large<STRAT>(h, t, out, q, start, n);
        }
    }
}

static thread_local uint32_t g_grid = 0, g_block = 0;

const char* consensus_kernel_name() { return "blu_consensus_wave_kernel"; }
void consensus_last_geometry(uint32_t* grid, uint32_t* block) {
    if (grid) *grid = g_grid;
    if (block) *block = g_block;
}

int launch_consensus(const TaxDev& tax, const HitsDev& hits, int strategy, blu_result* out, void* stream, int device,
                     int num_cus) {
    (void)device;
    if (hits.n_queries == 0) return BLU_OK;
    const int block = 256;
    int per_cu = 0;
    hipError_t e;
    if (strategy == BLU_RELAXED)
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, blu_consensus_wave_kernel<BLU_RELAXED>, block, 0);
    else
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, blu_consensus_wave_kernel<BLU_CAUTIOUS>, block, 0);
    if (e != hipSuccess || per_cu <= 0) per_cu = 4;
    if (per_cu > 8) per_cu = 8;
    uint64_t want = (hits.n_queries + 3) / 4;
    uint64_t cap = (uint64_t)(num_cus > 0 ? num_cus : 256) * (uint64_t)per_cu;
    uint32_t grid = (uint32_t)(want < cap ? want : cap);
    if (grid == 0) grid = 1;
    g_grid = grid;
    g_block = block;
    hipStream_t s = (hipStream_t)stream;
    if (strategy == BLU_RELAXED)
        hipLaunchKernelGGL(blu_consensus_wave_kernel<BLU_RELAXED>, dim3(grid), dim3(block), 0, s, hits, tax, out);
    else
        hipLaunchKernelGGL(blu_consensus_wave_kernel<BLU_CAUTIOUS>, dim3(grid), dim3(block), 0, s, hits, tax, out);
    e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("kernel launch failed: %s", hipGetErrorString(e));
        return BLU_ERR_HIP;
    }
    return BLU_OK;
}

}  // namespace blu
