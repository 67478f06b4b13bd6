// ============================================================================
// oracle/blu_oracle_columnar.cpp — TEST INFRASTRUCTURE ONLY.
//
// The same reference semantics as blu_oracle.cpp, transliterated onto the
// interned SoA inputs the GPU engine consumes (canonical layout)
// so that large runs can be compared record for record.  It keeps the
// reference's control flow (stable sort of the top group, per-level loop with
// take_while, filtered-index truncation) and gets its cutoffs from the
// string-faithful restatement (blu_oracle_interpolate) — it shares no code with
// blutils_amd/csrc.  tests/test_oracle_columnar.py checks it against the
// string-faithful oracle on random tables; that oracle is the one pinned on the
// reference's golden vectors.
// ============================================================================
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <string>
#include <thread>
#include <vector>

extern "C" int32_t blu_oracle_interpolate(int32_t taxon, int32_t has_custom, const int16_t* custom,
                                          const uint8_t* custom_has, int32_t n, const char* const* ranks,
                                          double* out_cutoff, uint8_t* out_is_default);
extern "C" int32_t blu_oracle_rank_display(const char* rank, char* buf, int32_t buflen);

namespace {

// record layout = include/blu_consensus.h blu_result (32 bytes), restated so
// the oracle does not depend on the product tree
struct Rec {
    uint8_t status, flags, bean_index, max_allowed_level;
    uint16_t reached_rank, max_allowed_rank;
    uint32_t identifier_node, ref_row;
    uint64_t level_mask;
    double ident_used;
};
static_assert(sizeof(Rec) == 32, "record layout");

enum : uint8_t { ST_MULTI = 0, ST_SINGLE = 1, ST_NO_HITS = 2, ST_UNMATCHED = 16, ST_BAD_LINEAGE = 17,
                 ST_ROOT = 18, ST_SINGLE_EMPTY = 19, ST_BAD_PIDENT = 20 };
constexpr uint8_t NONE8 = 0xFF;
constexpr uint16_t NONE16 = 0xFFFF, NEVER = 0xFFFE;
constexpr uint32_t UNMATCHED = 0xFFFFFFFFu;

struct Shape {
    std::vector<double> cut;
    std::vector<uint8_t> isdef;
    std::vector<uint16_t> canon;  // canonical code per level (same Display string <=> same code)
    std::vector<uint8_t> is_enum; // level's rank parses to one of the nine enum kinds
};

struct Tax {
    uint64_t n_tax;
    const uint64_t* off;
    const uint32_t* node;
    const uint16_t* rank;
    const uint8_t* bad;
    std::vector<uint32_t> shape_of;
    std::vector<Shape> shapes;
};

Rec err(uint8_t st, uint32_t row) {
    Rec r{}; r.status = st; r.bean_index = NONE8; r.max_allowed_level = NONE8; r.reached_rank = NONE16;
    r.max_allowed_rank = NONE16; r.identifier_node = 0xFFFFFFFFu; r.ref_row = row; return r;
}

Rec one_query(const Tax& T, uint64_t start, uint64_t n, const int32_t* bs, const uint32_t* tax, const double* pid,
              const int32_t* aln, const uint32_t* acc, int strategy) {
    if (n == 0) return err(ST_NO_HITS, 0xFFFFFFFFu);
    int32_t M = bs[start];                                       // find_single_query_consensus.rs:28-50
    for (uint64_t i = 1; i < n; ++i) M = std::max(M, bs[start + i]);
    std::vector<uint32_t> G;                                     // :51-64, file order
    for (uint64_t i = 0; i < n; ++i) {
        if (bs[start + i] != M) continue;
        uint32_t row = (uint32_t)(start + i), t = tax[row];
        if (t == UNMATCHED || t >= T.n_tax) return err(ST_UNMATCHED, row);     // lineage "null" fails parse
        if ((T.bad && T.bad[t]) || T.off[t + 1] == T.off[t]) return err(ST_BAD_LINEAGE, row);
        G.push_back(row);
    }
    for (uint32_t row : G) if (std::isnan(pid[row])) return err(ST_BAD_PIDENT, row);  // outside the restated domain
    auto len_of = [&](uint32_t row) { uint32_t t = tax[row]; return (uint32_t)(T.off[t + 1] - T.off[t]); };
    auto node_of = [&](uint32_t row, uint32_t lvl) { return T.node[T.off[tax[row]] + lvl]; };
    if (G.size() == 1) {                                         // :74-150
        uint32_t h = G[0];
        const Shape& S = T.shapes[T.shape_of[tax[h]]];
        uint32_t L = len_of(h);
        uint64_t A = 0;
        for (uint32_t j = 0; j < L; ++j) if (pid[h] >= S.cut[j]) A |= 1ull << j;   // linnaean_ranks.rs:194-212
        if (!A) return err(ST_SINGLE_EMPTY, h);                  // :113-119
        uint32_t last = 63 - (uint32_t)__builtin_clzll(A);
        Rec r{}; r.status = ST_SINGLE; r.flags = 0; r.bean_index = (uint8_t)last; r.max_allowed_level = NONE8;
        r.reached_rank = S.canon[last]; r.max_allowed_rank = NONE16; r.identifier_node = node_of(h, last);
        r.ref_row = h; r.level_mask = A; r.ident_used = pid[h];
        return r;
    }
    // find_multi_taxa_consensus.rs:39-54 stable sort
    std::vector<uint32_t> S = G;
    std::stable_sort(S.begin(), S.end(), [&](uint32_t a, uint32_t b) {
        uint32_t la = len_of(a), lb = len_of(b);
        if (la != lb) return la < lb;
        if (pid[a] < pid[b]) return true;
        if (pid[a] > pid[b]) return false;
        if (aln[a] != aln[b]) return aln[a] < aln[b];
        return acc[a] < acc[b];
    });
    uint32_t R = strategy == 0 ? S.front() : S.back();           // :60-68
    const Shape& SH = T.shapes[T.shape_of[tax[R]]];
    uint32_t LR = len_of(R);
    Rec fin{}; bool have = false;
    auto build = [&](uint32_t b, double ident, bool single_flag, size_t n_beans) {   // build_blast_consensus_identity.rs:9-105
        Rec r{}; r.status = ST_MULTI; r.ref_row = R; r.bean_index = (uint8_t)b; r.ident_used = ident;
        r.max_allowed_level = NONE8; r.max_allowed_rank = NONE16; r.flags = 0;
        for (uint32_t j = 0; j < LR; ++j)                        // linnaean_ranks.rs:174-192
            if (!(ident > SH.cut[j])) {
                r.max_allowed_level = (uint8_t)j;
                // DefaultRank(rank) -> rank ; NonDefaultRank(name) -> Other(name)
                r.max_allowed_rank = SH.isdef[j] ? SH.canon[j] : (SH.is_enum[j] ? NEVER : SH.canon[j]);
                if (r.max_allowed_rank != SH.canon[b]) r.flags |= 1;  // mutated (:35-37)
                break;
            }
        std::vector<uint32_t> F;                                 // :67-72
        for (uint32_t j = 0; j < LR; ++j) if (ident >= SH.cut[j]) F.push_back(j);
        std::vector<uint32_t> A;
        if (single_flag && n_beans == 1) A = F;                  // :74-75
        else for (size_t i = 0; i < F.size() && i <= b; ++i) A.push_back(F[i]);   // :76-82
        uint32_t last = A.empty() ? b : A.back();                // :85
        r.identifier_node = node_of(R, last);
        r.reached_rank = SH.canon[last];
        for (uint32_t j : A) r.level_mask |= 1ull << j;
        return r;
    };
    for (uint32_t index = 0; index < LR; ++index) {              // :137
        size_t take = 0;                                         // :142-145 take_while over the length-ascending sort
        while (take < S.size() && index < len_of(S[take])) ++take;
        std::vector<uint32_t> level_set;                         // :150-159
        for (size_t r = 0; r < take; ++r) {
            uint32_t e = node_of(S[r], index);
            if (std::find(level_set.begin(), level_set.end(), e) == level_set.end()) level_set.push_back(e);
        }
        if (level_set.empty()) continue;                         // :161-163
        if (level_set.size() > 1) {                              // :180
            if (index == 0) return err(ST_ROOT, R);              // :181
            double mx = 0.0;                                     // :182-185
            for (size_t r = 0; r < take; ++r) if (pid[S[r]] > mx) mx = pid[S[r]];
            fin = build(index - 1, mx, false, level_set.size()); // :190-199
            have = true;
            break;
        }
        fin = build(index, pid[R], true, level_set.size());      // :204-213
        fin.flags |= 2;  // agree so far
        have = true;
    }
    if (!have) return err(ST_BAD_LINEAGE, R);
    return fin;
}

}  // namespace

extern "C" {

// Columnar oracle over the canonical SoA layout.  rank_names[] as given to the
// engine; cutoffs come from the string-faithful interpolate.  Returns 0 or the
// oracle panic code of the cutoff configuration.
int32_t blu_oracle_columnar_run(uint64_t n_tax, const uint64_t* lin_off, const uint32_t* lin_node,
                                const uint16_t* lin_rank, uint32_t n_ranks, const char* const* rank_names,
                                const uint8_t* bad, int32_t taxon, int32_t has_custom, const int16_t* custom,
                                const uint8_t* custom_has, uint64_t n_queries, const uint64_t* seg_off,
                                const int32_t* bs, const uint32_t* tax, const double* pid, const int32_t* aln,
                                const uint32_t* acc, int32_t strategy, int32_t threads, void* out_records) {
    Tax T{n_tax, lin_off, lin_node, lin_rank, bad, {}, {}};
    // canonical codes by Display string (linnaean_ranks.rs:74-89): enum kinds 0..8, Other(slug) from 9
    std::vector<uint16_t> canon(n_ranks);
    std::vector<uint8_t> is_enum(n_ranks);
    std::map<std::string, uint16_t> others;
    for (uint32_t r = 0; r < n_ranks; ++r) {
        char buf[256];
        int kind = blu_oracle_rank_display(rank_names[r], buf, sizeof buf);
        if (kind < 9) { canon[r] = (uint16_t)kind; is_enum[r] = 1; }
        else {
            auto it = others.find(buf);
            if (it == others.end()) it = others.emplace(buf, (uint16_t)(9 + others.size())).first;
            canon[r] = it->second; is_enum[r] = 0;
        }
    }
    std::map<std::vector<uint16_t>, uint32_t> ids;
    T.shape_of.assign(n_tax, 0);
    for (uint64_t t = 0; t < n_tax; ++t) {
        std::vector<uint16_t> seq(lin_rank + lin_off[t], lin_rank + lin_off[t + 1]);
        auto it = ids.find(seq);
        if (it == ids.end()) {
            Shape s;
            std::vector<const char*> names;
            for (uint16_t r : seq) { names.push_back(rank_names[r]); s.canon.push_back(canon[r]); s.is_enum.push_back(is_enum[r]); }
            s.cut.resize(seq.size()); s.isdef.resize(seq.size());
            if (!seq.empty()) {
                int32_t rc = blu_oracle_interpolate(taxon, has_custom, custom, custom_has, (int32_t)seq.size(),
                                                    names.data(), s.cut.data(), s.isdef.data());
                if (rc) return rc;
            }
            it = ids.emplace(seq, (uint32_t)T.shapes.size()).first;
            T.shapes.push_back(std::move(s));
        }
        T.shape_of[t] = it->second;
    }
    Rec* out = (Rec*)out_records;
    std::atomic<uint64_t> next{0};
    auto worker = [&]() {
        for (;;) {
            uint64_t q0 = next.fetch_add(256);
            if (q0 >= n_queries) break;
            uint64_t q1 = std::min<uint64_t>(n_queries, q0 + 256);
            for (uint64_t q = q0; q < q1; ++q)
                out[q] = one_query(T, seg_off[q], seg_off[q + 1] - seg_off[q], bs, tax, pid, aln, acc, strategy);
        }
    };
    int nt = threads > 0 ? threads : 1;
    if (nt == 1) worker();
    else { std::vector<std::thread> pool; for (int i = 0; i < nt; ++i) pool.emplace_back(worker); for (auto& th : pool) th.join(); }
    return 0;
}

}  // extern "C"
