// Host side of the taxonomy handle: canonical rank codes, rank-sequence
// "shapes", per-shape cutoff tables, fixed-stride lineage rows, upload.
//
// Reference semantics restated here (product code — independent of oracle/):
//   LinnaeanRank::from_str / Display        core/src/domain/dtos/linnaean_ranks.rs:52-89
//   Taxon::get_taxon_cutoff + tables        core/src/domain/dtos/taxon.rs:104-185
//   InterpolatedIdentity::interpolate_identities   linnaean_ranks.rs:220-383
//   round(value, 3)                         core/src/domain/utils/mod.rs:1-4
// The cutoffs are a pure function of (Taxon, CustomTaxon, rank sequence), so
// they are computed once per distinct rank sequence instead of once per query.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>

#include "blu_internal.h"

namespace blu {

static thread_local std::string g_last_error;

void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
}

// slugify 0.1 (third-party crate; ASCII behaviour): lowercase, keep [a-z0-9],
// collapse every other run into one '-', trim '-' at both ends.
static std::string slugify_ascii(const std::string& s) {
    std::string out;
    bool pending = false;
    for (unsigned char c : s) {
        if (c >= 'A' && c <= 'Z') c = (unsigned char)(c + 32);
        bool keep = (c >= 'a' && c <= 'z') || (c >= '0' && c <= '9');
        if (keep) {
            if (pending && !out.empty()) out.push_back('-');
            pending = false;
            out.push_back((char)c);
        } else {
            pending = true;
        }
    }
    return out;
}

// linnaean_ranks.rs:52-72: lowercase + trim, letters and full names map to the
// enum, anything else to Other(slugify(x)).  Returns the enum kind (0..8) or
// K_FIRST_OTHER with `other` set.
uint16_t parse_rank(const char* name, std::string* other) {
    std::string low;
    for (const unsigned char* p = (const unsigned char*)name; *p; ++p)
        low.push_back((*p >= 'A' && *p <= 'Z') ? (char)(*p + 32) : (char)*p);
    size_t a = 0, b = low.size();
    while (a < b && isspace((unsigned char)low[a])) ++a;
    while (b > a && isspace((unsigned char)low[b - 1])) --b;
    low = low.substr(a, b - a);
    static const struct { const char* letter; const char* full; uint16_t kind; } tab[] = {
        {"u", "undefined", K_UNDEFINED}, {"d", "domain", K_DOMAIN}, {"k", "kingdom", K_KINGDOM},
        {"p", "phylum", K_PHYLUM},       {"c", "class", K_CLASS},   {"o", "order", K_ORDER},
        {"f", "family", K_FAMILY},       {"g", "genus", K_GENUS},   {"s", "species", K_SPECIES}};
    for (auto& e : tab)
        if (low == e.letter || low == e.full) return e.kind;
    *other = slugify_ascii(low);
    return K_FIRST_OTHER;
}

struct Backbone {
    bool has[9] = {false};   // enum kind present as DefaultRank in the backbone
    double cut[9] = {0};
    double first = 0.0;      // backbone[0]'s identity (linnaean_ranks.rs:343-346)
};

// taxon.rs:104-185
static int make_backbone(const blu_cutoff_config& cfg, Backbone* bb) {
    auto set = [&](uint16_t k, double v) { bb->has[k] = true; bb->cut[k] = v; };
    switch (cfg.taxon) {
        case BLU_TAXON_FUNGI:
        case BLU_TAXON_EUKARYOTES:  // taxon.rs:144-154, 174-184
            set(K_SPECIES, 97.0); set(K_GENUS, 95.0); set(K_FAMILY, 90.0); set(K_ORDER, 85.0);
            set(K_CLASS, 80.0); set(K_PHYLUM, 75.0); set(K_DOMAIN, 60.0);
            bb->first = 97.0;
            return BLU_OK;
        case BLU_TAXON_BACTERIA:    // taxon.rs:159-169
            set(K_SPECIES, 99.0); set(K_GENUS, 97.0); set(K_FAMILY, 92.0); set(K_ORDER, 85.0);
            set(K_CLASS, 80.0); set(K_PHYLUM, 75.0); set(K_DOMAIN, 60.0);
            bb->first = 99.0;
            return BLU_OK;
        case BLU_TAXON_CUSTOM: {    // taxon.rs:113-139
            if (!cfg.has_custom) {
                set_error("Custom taxon values are required (taxon.rs:117)");
                return BLU_ERR_CUSTOM_MISSING;
            }
            static const uint16_t order[8] = {K_DOMAIN, K_KINGDOM, K_PHYLUM, K_CLASS, K_ORDER, K_FAMILY, K_GENUS, K_SPECIES};
            for (int i = 0; i < 8; ++i) {
                bool mandatory = (i == 0 || i == 7);
                int16_t v = (mandatory || cfg.custom_has[i]) ? cfg.custom[i] : (int16_t)0;
                set(order[i], (double)v);
            }
            bb->first = bb->cut[K_DOMAIN];
            return BLU_OK;
        }
    }
    set_error("unknown taxon %d", cfg.taxon);
    return BLU_ERR_INVALID_ARG;
}

// linnaean_ranks.rs:220-383 in index form.  Two mapped entries compare equal
// (the `position(|level| level == x)` searches) exactly when their canonical
// rank codes are equal, so "first equal element" = first level with that code.
static void interpolate_shape(const Backbone& bb, const uint16_t* code, int n, double* cut, uint8_t* isdef) {
    bool all_default = true;
    for (int i = 0; i < n; ++i) {
        bool d = code[i] < K_FIRST_OTHER && bb.has[code[i]];
        isdef[i] = d;
        cut[i] = d ? bb.cut[code[i]] : 0.0;
        all_default &= d;
    }
    if (all_default) return;  // :265-270
    std::vector<double> base(cut, cut + n);
    auto first_with_code = [&](uint16_t c) { for (int j = 0; j < n; ++j) if (code[j] == c) return j; return 0; };
    for (int i = 0; i < n; ++i) {
        if (isdef[i]) continue;
        int prev = 0;                                     // :292-300
        for (int j = i - 1; j >= 0; --j) if (isdef[j]) { prev = j; break; }
        int prev_idx = first_with_code(code[prev]);       // :302-305
        int next = n - 1;                                 // :307-317
        for (int j = i; j < n; ++j) if (isdef[j]) { next = j; break; }
        int next_idx = first_with_code(code[next]);       // :319-322
        int wlen = std::min(next_idx + 1, n - prev_idx);  // :324-329 skip_while(!= previous).take(next_index + 1)
        int wlast = prev_idx + wlen - 1;
        double first = isdef[prev_idx] ? base[prev_idx] : bb.first;   // :341-347
        double last = isdef[wlast] ? base[wlast] : 100.0;             // :349-353
        double weight = last - first;                                 // :355
        double size = (double)(wlen - 1);                             // :356
        double step = weight / size;
        double scaled = (double)(i - prev_idx) * step;                // :339, :360
        double v = first + scaled;
        cut[i] = std::round(v * 1000.0) / 1000.0;                     // utils/mod.rs:1-4
    }
}

}  // namespace blu

using namespace blu;

extern "C" {

uint32_t blu_abi_version(void) { return BLU_ABI_VERSION; }

size_t blu_last_error(char* buf, size_t len) {
    const std::string& e = blu::g_last_error;
    if (buf && len) {
        size_t n = std::min(len - 1, e.size());
        memcpy(buf, e.data(), n);
        buf[n] = 0;
    }
    return e.size();
}

int blu_taxonomy_create(const blu_taxonomy_desc* desc, const blu_cutoff_config* cfg, int device,
                        blu_taxonomy** out) {
    if (!desc || !cfg || !out) { set_error("null argument"); return BLU_ERR_INVALID_ARG; }
    *out = nullptr;
    if (desc->n_tax && (!desc->lin_off || !desc->lin_node || !desc->lin_rank)) {
        set_error("lineage arrays missing"); return BLU_ERR_INVALID_ARG;
    }
    if (desc->n_tax >= (1ull << BLU_ROW_BITS)) { set_error("n_tax must be < 2^%u", BLU_ROW_BITS); return BLU_ERR_INVALID_ARG; }
    Backbone bb;
    int rc = make_backbone(*cfg, &bb);
    if (rc != BLU_OK) return rc;

    auto* tax = new blu_taxonomy();
    tax->device = device;
    tax->cfg = *cfg;
    tax->n_tax = desc->n_tax;

    // canonical rank codes
    static const char* letters[9] = {"u", "d", "k", "p", "c", "o", "f", "g", "s"};
    static const char* fulls[9] = {"undefined", "domain", "kingdom", "phylum", "class", "order", "family", "genus", "species"};
    for (int k = 0; k < 9; ++k) tax->ranks.push_back({letters[k], fulls[k]});
    std::vector<uint16_t> code_of(desc->n_ranks);
    std::map<std::string, uint16_t> other_codes;
    for (uint32_t r = 0; r < desc->n_ranks; ++r) {
        std::string other;
        uint16_t k = parse_rank(desc->rank_names[r], &other);
        if (k < K_FIRST_OTHER) { code_of[r] = k; continue; }
        auto it = other_codes.find(other);
        if (it == other_codes.end()) {
            if (tax->ranks.size() >= BLU_MAR_NEVER_EQUAL) { delete tax; set_error("too many rank names"); return BLU_ERR_INVALID_ARG; }
            it = other_codes.emplace(other, (uint16_t)tax->ranks.size()).first;
            tax->ranks.push_back({other, other});
        }
        code_of[r] = it->second;
    }

    // depth scan
    uint32_t max_depth = 0;
    for (uint64_t t = 0; t < desc->n_tax; ++t) {
        if (desc->lin_off[t + 1] < desc->lin_off[t]) { delete tax; set_error("lin_off not monotone at %llu", (unsigned long long)t); return BLU_ERR_INVALID_ARG; }
        uint64_t len = desc->lin_off[t + 1] - desc->lin_off[t];
        if (len > BLU_MAX_DEPTH) { delete tax; set_error("lineage of row %llu has %llu levels (max %u)", (unsigned long long)t, (unsigned long long)len, BLU_MAX_DEPTH); return BLU_ERR_DEPTH; }
        max_depth = std::max<uint32_t>(max_depth, (uint32_t)len);
    }
    tax->max_depth = max_depth;
    tax->stride = ((max_depth + 1 + 15) / 16) * 16;
    if (tax->stride < 16) tax->stride = 16;
    tax->sc = ((std::max<uint32_t>(max_depth, 1) + 15) / 16) * 16;

    // shapes + lineage rows
    std::map<std::vector<uint16_t>, uint32_t> shape_ids;
    std::vector<std::vector<uint16_t>> shapes;
    tax->h_lin.assign((size_t)desc->n_tax * tax->stride, 0u);
    std::vector<uint16_t> seq;
    for (uint64_t t = 0; t < desc->n_tax; ++t) {
        uint64_t o = desc->lin_off[t];
        uint32_t len = (uint32_t)(desc->lin_off[t + 1] - o);
        uint32_t* row = &tax->h_lin[(size_t)t * tax->stride];
        bool bad = (desc->bad && desc->bad[t]) || len == 0;  // "" fails parse_taxonomy too (blast_result.rs:65-67)
        if (bad) { row[0] = 0; continue; }
        seq.resize(len);
        for (uint32_t j = 0; j < len; ++j) {
            uint16_t rk = desc->lin_rank[o + j];
            if (rk >= desc->n_ranks) { delete tax; set_error("lin_rank out of range at row %llu", (unsigned long long)t); return BLU_ERR_INVALID_ARG; }
            seq[j] = code_of[rk];
            row[1 + j] = desc->lin_node[o + j];
        }
        auto it = shape_ids.find(seq);
        if (it == shape_ids.end()) {
            if (shapes.size() >= (1u << 24)) { delete tax; set_error("too many lineage shapes"); return BLU_ERR_INVALID_ARG; }
            it = shape_ids.emplace(seq, (uint32_t)shapes.size()).first;
            shapes.push_back(seq);
        }
        row[0] = len | (it->second << 8);
    }
    tax->n_shapes = (uint32_t)shapes.size();
    size_t nsh = std::max<size_t>(shapes.size(), 1);
    tax->h_cut.assign(nsh * tax->sc, 0.0);
    tax->h_codes.assign(nsh * tax->sc, 0u);
    tax->h_isdef.assign(nsh * tax->sc, 0);
    for (size_t s = 0; s < shapes.size(); ++s) {
        const auto& sq = shapes[s];
        double* cut = &tax->h_cut[s * tax->sc];
        uint8_t* isdef = &tax->h_isdef[s * tax->sc];
        interpolate_shape(bb, sq.data(), (int)sq.size(), cut, isdef);
        for (size_t j = 0; j < sq.size(); ++j) {
            // max_allowed_rank of level j (build_blast_consensus_identity.rs:22-30): DefaultRank(rank) -> rank;
            // NonDefaultRank(name) -> Other(name), which equals a parsed rank only when that rank is itself Other.
            uint32_t mar = isdef[j] ? sq[j] : (sq[j] >= K_FIRST_OTHER ? sq[j] : BLU_MAR_NEVER_EQUAL);
            tax->h_codes[s * tax->sc + j] = (uint32_t)sq[j] | (mar << 16);
        }
    }
    if (desc->taxid) {
        tax->taxid_row.reserve((size_t)desc->n_tax * 2);
        // polars left join on a table with duplicate keys would duplicate rows; blutils DBs have unique taxids.
        for (uint64_t t = 0; t < desc->n_tax; ++t) tax->taxid_row.emplace(desc->taxid[t], (uint32_t)t);
    }

    // ---- lexicographic order of the lineage rows, adjacent-row LCP, block sparse table -------------------
    const uint64_t n = desc->n_tax;
    std::vector<uint32_t> order(n);
    for (uint64_t i = 0; i < n; ++i) order[i] = (uint32_t)i;
    const uint32_t stride = tax->stride;
    const uint32_t* L = tax->h_lin.data();
    std::sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) {
        const uint32_t* ra = L + (size_t)a * stride;
        const uint32_t* rb = L + (size_t)b * stride;
        const uint32_t la = ra[0] & 0xFF, lb = rb[0] & 0xFF, m = la < lb ? la : lb;
        for (uint32_t j = 1; j <= m; ++j)
            if (ra[j] != rb[j]) return ra[j] < rb[j];
        if (la != lb) return la < lb;
        return a < b;
    });
    tax->pos_of.assign(n, 0);
    tax->hint_of_pos.assign(std::max<uint64_t>(n, 1), 0);
    for (uint64_t i = 0; i < n; ++i) {  // engine row id: sorted position | lineage length << BLU_ROW_BITS
        const uint32_t hdr = L[(size_t)order[i] * stride];
        tax->pos_of[order[i]] = (uint32_t)i | ((hdr & 0xFF) << BLU_ROW_BITS);
        // shape hint of the packed layout's side records: shape id + 1 in 15 bits, 0 = none (a bad lineage, or a shape id that
        // does not fit: the engine then reads the shape from the row, one memory round trip later)
        const uint32_t shape = hdr >> 8;
        tax->hint_of_pos[i] = ((hdr & 0xFF) != 0 && shape + 1 < (1u << BLU_HINT_BITS)) ? (uint16_t)(shape + 1) : (uint16_t)0;
    }
    std::vector<uint32_t> lin_sorted((size_t)n * stride);   // host layout (hdr, nodes), sorted: for the LCP pass below
    for (uint64_t i = 0; i < n; ++i)
        memcpy(&lin_sorted[(size_t)i * stride], L + (size_t)order[i] * stride, stride * sizeof(uint32_t));
    // distinct cutoff values (by bit pattern, NaN included) and the device rows (interleaved node / packed level words)
    if (tax->ranks.size() > BLU_PACK_NEVER) { delete tax; set_error("more than %u distinct rank names", BLU_PACK_NEVER); return BLU_ERR_INVALID_ARG; }
    std::map<uint64_t, uint32_t> cut_ids;
    std::vector<double> cutvals;
    auto cut_id = [&](double v) {
        uint64_t bits;
        memcpy(&bits, &v, 8);
        auto it = cut_ids.find(bits);
        if (it == cut_ids.end()) { it = cut_ids.emplace(bits, (uint32_t)cutvals.size()).first; cutvals.push_back(v); }
        return it->second;
    };
    // device rows = the sorted host rows (header, node ids): 64 bytes for lineages of up to 15 levels; the per-level
    // cutoff ids and rank codes depend on the shape only and live in a small table of their own (dev_codes)
    // (the rows themselves are laid out further down, once the adjacent-row LCPs are known)
    std::vector<uint32_t> dev_codes(std::max<size_t>((size_t)tax->n_shapes * tax->sc, 16), 0u);
    for (size_t k = 0; k < (size_t)tax->n_shapes * tax->sc; ++k) {
        const uint32_t code = tax->h_codes[k];
        const uint32_t rank = code & 0xFFFF, mar = code >> 16;
        dev_codes[k] = cut_id(tax->h_cut[k]) | (rank << BLU_PACK_CUT_BITS) |
                       ((mar == BLU_MAR_NEVER_EQUAL ? BLU_PACK_NEVER : mar) << (BLU_PACK_CUT_BITS + BLU_PACK_CODE_BITS));
    }
    // per level: the smallest milli-percent identity that passes the cutoff (see TaxDev::kthr).  fl(k / 1000.0) is
    // increasing in k, so a binary search over k with the very f64 comparison the reference makes is exact.
    std::vector<uint32_t> dev_kthr(dev_codes.size(), BLU_KTHR_NEVER);
    {
        std::map<uint64_t, uint32_t> memo;
        for (size_t k = 0; k < (size_t)tax->n_shapes * tax->sc; ++k) {
            const double c = tax->h_cut[k];
            uint64_t bits;
            memcpy(&bits, &c, 8);
            auto it = memo.find(bits);
            if (it == memo.end()) {
                uint32_t w = BLU_KTHR_NEVER;
                if (c == c) {                                   // (a NaN cutoff passes nothing)
                    uint32_t lo = 0, hi = BLU_KTHR_NEVER;         // smallest k in [0, NEVER) with fl(k / 1000) >= c, else NEVER
                    while (lo < hi) {
                        const uint32_t mid = lo + (hi - lo) / 2;
                        if ((double)mid / 1000.0 >= c) hi = mid; else lo = mid + 1;
                    }
                    w = lo;
                    if (w != BLU_KTHR_NEVER && (double)w / 1000.0 == c) w |= 1u << BLU_KTHR_BITS;
                }
                it = memo.emplace(bits, w).first;
            }
            dev_kthr[k] = it->second;
        }
    }
    // One word per level with everything the integer level tests and the record need: that threshold (18 bits) | canonical
    // rank code (10 bits) << 18 | "max_allowed_rank can never equal a parsed rank" (BLU_MAR_NEVER_EQUAL) << 28 — the
    // max-allowed-rank code of a level is its own rank code or that constant (see h_codes above), so one bit says which.
    for (size_t k = 0; k < (size_t)tax->n_shapes * tax->sc; ++k) {
        const uint32_t code = tax->h_codes[k];
        const uint32_t rank = code & 0xFFFF, mar = code >> 16;
        dev_kthr[k] |= (rank << BLU_LVL_RANK_SHIFT) | ((mar == BLU_MAR_NEVER_EQUAL ? 1u : 0u) << BLU_LVL_NEVER_SHIFT);
    }
    if (cutvals.size() >= (1u << BLU_PACK_CUT_BITS)) { delete tax; set_error("more than %u distinct cutoff values", (1u << BLU_PACK_CUT_BITS) - 1); return BLU_ERR_INVALID_ARG; }
    if (cutvals.empty()) cutvals.push_back(0.0);
    tax->n_cutvals = (uint32_t)cutvals.size();
    tax->order = order;
    const uint64_t n_lcp = n > 0 ? n - 1 : 0;
    const uint32_t nb = (uint32_t)((n_lcp + 15) / 16) + 1;            // 16-entry blocks (+1 block of padding)
    std::vector<uint8_t> lcp8((size_t)nb * 16 + 16, 0xFF);
    for (uint64_t i = 0; i < n_lcp; ++i) {
        const uint32_t* ra = &lin_sorted[(size_t)i * stride];
        const uint32_t* rb = ra + stride;
        const uint32_t la = ra[0] & 0xFF, lb = rb[0] & 0xFF, m = la < lb ? la : lb;
        uint32_t c = 0;
        while (c < m && ra[1 + c] == rb[1 + c]) ++c;
        lcp8[i] = (uint8_t)c;
    }
    // Device rows: word 0 = len | shape << 8; words 1..5 / 6..10 = for each of the first 20 levels a byte 0x80 | a_j / 0x80 | b_j,
    // a_j / b_j = how many sorted rows to the left / right of this row still share its levels 0..j, saturated at 127;
    // words 11.. = the node ids.  With them the levels shared by a group spanning [lo, hi] around row r are the levels
    // with a_j >= r - lo and b_j >= hi - r — exact whenever both distances are at most 127, which is the common case (the
    // group sits inside one genus or family); wider groups, and agreement deeper than 20 levels, use the range-minimum
    // tables.  One 128-byte line holds it all for lineages of up to 20 levels.
    const uint32_t D = std::max<uint32_t>(tax->max_depth, 1);
    const uint32_t node_base = BLU_ROW_NODE_BASE;
    const uint32_t dstride = ((node_base + D + 31) / 32) * 32;
    tax->dev_stride = dstride;
    tax->node_base = node_base;
    std::vector<uint32_t> dev_rows(std::max<size_t>((size_t)n * dstride, 32), 0u);
    {
        std::vector<uint32_t> run(D, 0);        // run[j] = start (left pass) / end (right pass) of the current run at level j
        for (uint64_t r = 0; r < n; ++r) {
            const uint32_t* src = &lin_sorted[(size_t)r * stride];
            uint32_t* dst = &dev_rows[(size_t)r * dstride];
            const uint32_t len = src[0] & 0xFF;
            dst[0] = src[0];
            memset(dst + 1, 0x80, 2 * BLU_ROW_IV_LEVELS);               // run length 0 at every level until the passes below say otherwise
            for (uint32_t j = 0; j < len; ++j) dst[node_base + j] = src[1 + j];
            const uint32_t shared = r > 0 ? lcp8[r - 1] : 0;          // levels shared with the previous row
            uint8_t* ab = reinterpret_cast<uint8_t*>(dst + 1);
            for (uint32_t j = 0; j < len; ++j) {
                if (j >= shared) run[j] = (uint32_t)r;                   // a new run starts here at level j
                if (j < BLU_ROW_IV_LEVELS) ab[j] = (uint8_t)(0x80u | std::min<uint64_t>(r - run[j], BLU_ROW_RUN_MAX));
            }
            for (uint32_t j = len; j < D; ++j) run[j] = (uint32_t)r + 1; // levels this row does not have break the runs
        }
        for (uint64_t r = n; r-- > 0;) {
            uint32_t* dst = &dev_rows[(size_t)r * dstride];
            const uint32_t len = dst[0] & 0xFF;
            const uint32_t shared = r + 1 < n ? lcp8[r] : 0;           // levels shared with the next row
            uint8_t* ab = reinterpret_cast<uint8_t*>(dst + 1);
            for (uint32_t j = 0; j < len; ++j) {
                if (j >= shared) run[j] = (uint32_t)r;
                if (j < BLU_ROW_IV_LEVELS) ab[BLU_ROW_IV_LEVELS + j] = (uint8_t)(0x80u | std::min<uint64_t>(run[j] - r, BLU_ROW_RUN_MAX));
            }
            for (uint32_t j = len; j < D; ++j) run[j] = (uint32_t)r;   // (never read before being reset: j >= shared for the next row)
        }
    }
    uint32_t levels = 1;
    while ((1u << levels) <= nb) ++levels;
    std::vector<uint8_t> rmq((size_t)levels * nb, 0xFF);
    for (uint32_t j = 0; j < nb; ++j) {
        uint8_t m = 0xFF;
        for (int b = 0; b < 16; ++b) m = std::min(m, lcp8[(size_t)j * 16 + b]);
        rmq[j] = m;
    }
    for (uint32_t k = 1; k < levels; ++k)
        for (uint32_t j = 0; j + (1u << k) <= nb; ++j)
            rmq[(size_t)k * nb + j] = std::min(rmq[(size_t)(k - 1) * nb + j], rmq[(size_t)(k - 1) * nb + j + (1u << (k - 1))]);
    tax->rmq_nb = nb;

    // ---- wide-group tables (TaxDev::wblk / wchain): the nodes with at least BLU_WIDE_MIN rows, in preorder --------------
    // A node at level j is a maximal run of sorted rows that share levels 0..j (and have that many).  Runs are closed when a
    // row shares fewer levels with its predecessor; the wide ones are kept as {level, start, end}.
    struct WideNode { uint32_t level, s, e, parent; };
    std::vector<WideNode> wide;
    std::vector<uint2> wblk;
    std::vector<uint32_t> wchain, wchain_hi;
    uint32_t wide_levels = 0;
    bool wide_ok = true;
    {
        std::vector<uint32_t> open_s(D, 0);
        std::vector<uint8_t> open_on(D, 0);
        auto close_from = [&](uint32_t j0, uint32_t upto, uint32_t end) {
            for (uint32_t j = j0; j < upto; ++j) {
                if (open_on[j] && end - open_s[j] >= BLU_WIDE_MIN) wide.push_back({j, open_s[j], end, 0u});
                open_on[j] = 0;
            }
        };
        uint32_t prev_len = 0;
        for (uint64_t r = 0; r < n; ++r) {
            const uint32_t len = lin_sorted[(size_t)r * stride] & 0xFF;
            const uint32_t shared = r > 0 ? std::min<uint32_t>(lcp8[r - 1], std::min(len, prev_len)) : 0;
            close_from(shared, prev_len, (uint32_t)r);
            for (uint32_t j = shared; j < len; ++j) { open_s[j] = (uint32_t)r; open_on[j] = 1; }
            prev_len = len;
        }
        close_from(0, prev_len, (uint32_t)n);
        // preorder: by start, then by level (an ancestor starts no later than its descendants and is shallower)
        std::sort(wide.begin(), wide.end(), [](const WideNode& a, const WideNode& b) { return a.s != b.s ? a.s < b.s : a.level < b.level; });
        for (const auto& w : wide) wide_levels = std::max(wide_levels, w.level + 1);
        if (wide.size() >= 65535u || wide_levels > 2 * BLU_WCHAIN) wide_ok = false;
    }
    if (wide_ok) {
        // deepest wide node per position (id + 1), parents from a stack of the open ancestors
        std::vector<uint16_t> deepest(std::max<uint64_t>(n, 1), 0);
        std::vector<uint32_t> stack;
        wchain.assign((wide.size() + 1) * BLU_WCHAIN, 0u);
        if (wide_levels > BLU_WCHAIN) wchain_hi.assign((wide.size() + 1) * BLU_WCHAIN, 0u);
        for (size_t w = 0; w < wide.size(); ++w) {
            while (!stack.empty() && wide[stack.back()].e <= wide[w].s) stack.pop_back();
            wide[w].parent = stack.empty() ? 0xFFFFFFFFu : stack.back();
            stack.push_back((uint32_t)w);
            // the chain of w: its parent's, plus its own end at its own level (levels in between cannot be missing: every
            // ancestor of a wide node is wide)
            uint32_t* ch = &wchain[(w + 1) * BLU_WCHAIN];
            uint32_t* ch_hi = wchain_hi.empty() ? nullptr : &wchain_hi[(w + 1) * BLU_WCHAIN];
            if (wide[w].parent != 0xFFFFFFFFu) {
                memcpy(ch, &wchain[((size_t)wide[w].parent + 1) * BLU_WCHAIN], BLU_WCHAIN * sizeof(uint32_t));
                if (ch_hi) memcpy(ch_hi, &wchain_hi[((size_t)wide[w].parent + 1) * BLU_WCHAIN], BLU_WCHAIN * sizeof(uint32_t));
            }
            const uint32_t lv = wide[w].level;
            if (lv < BLU_WCHAIN) ch[lv] = wide[w].e; else ch_hi[lv - BLU_WCHAIN] = wide[w].e;
            for (uint32_t p = wide[w].s; p < wide[w].e; ++p) deepest[p] = (uint16_t)(w + 1);   // (preorder: deeper nodes overwrite)
        }
        const uint64_t nblk = (n >> BLU_WBLK_SHIFT) + 1;
        wblk.assign(nblk, make_uint2(0u, 0u));
        for (uint64_t b = 0; b < nblk; ++b) {
            const uint64_t p0 = b << BLU_WBLK_SHIFT, p1 = std::min<uint64_t>(n, p0 + (1u << BLU_WBLK_SHIFT));
            uint32_t w[3] = {p0 < n ? deepest[p0] : 0u, 0u, 0u}, sp[2] = {1u << BLU_WBLK_SHIFT, 1u << BLU_WBLK_SHIFT};
            uint32_t changes = 0;
            for (uint64_t p = p0 + 1; p < p1; ++p)
                if (deepest[p] != deepest[p - 1]) {
                    if (changes < 2) { sp[changes] = (uint32_t)(p - p0); w[changes + 1] = deepest[p]; }
                    ++changes;
                }
            if (changes > 2) wblk[b] = make_uint2(0u, (uint32_t)BLU_WBLK_OVERFLOW << 16);
            else wblk[b] = make_uint2(w[0] | (w[1] << 16), w[2] | (sp[0] << 16) | (sp[1] << 24));
        }
    }
    tax->n_wide = (uint32_t)wide.size();
    tax->wide_levels = wide_ok ? wide_levels : 0;
    tax->h_lcp8 = lcp8;
    tax->h_rmq = rmq;
    if (wide_ok) { tax->h_wblk = wblk; tax->h_wchain = wchain; tax->h_wchain_hi = wchain_hi; }

    if (device >= 0) {
        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
            delete tax; set_error("no HIP device available (this engine has no CPU fallback)"); return BLU_ERR_NO_DEVICE;
        }
        if (device >= ndev) { delete tax; set_error("device %d out of range (%d devices)", device, ndev); return BLU_ERR_INVALID_ARG; }
        hipError_t e = hipSetDevice(device);
        hipDeviceProp_t prop;
        if (e == hipSuccess) e = hipGetDeviceProperties(&prop, device);
        if (e == hipSuccess) tax->num_cus = prop.multiProcessorCount;
        // every table in ONE allocation (256-byte aligned pieces): one hipMalloc at creation, one hipFree at the end
        struct Part { void** view; const void* src; size_t bytes; size_t at; };
        const bool wide_tables = wide_ok && !wblk.empty();
        std::vector<Part> parts = {
            {(void**)&tax->d_lin, dev_rows.data(), dev_rows.size() * sizeof(uint32_t), 0},
            {(void**)&tax->d_lcp8, lcp8.data(), lcp8.size(), 0},
            {(void**)&tax->d_rmq, rmq.data(), rmq.size(), 0},
            {(void**)&tax->d_cutvals, cutvals.data(), cutvals.size() * sizeof(double), 0},
            {(void**)&tax->d_codes, dev_codes.data(), dev_codes.size() * sizeof(uint32_t), 0},
            {(void**)&tax->d_kthr, dev_kthr.data(), dev_codes.size() * sizeof(uint32_t), 0},
            {(void**)&tax->d_hint_of_pos, tax->hint_of_pos.data(), tax->hint_of_pos.size() * sizeof(uint16_t), 0},
        };
        if (wide_tables) {
            parts.push_back({(void**)&tax->d_wblk, wblk.data(), wblk.size() * sizeof(uint2), 0});
            parts.push_back({(void**)&tax->d_wchain, wchain.data(), wchain.size() * sizeof(uint32_t), 0});
            if (!wchain_hi.empty()) parts.push_back({(void**)&tax->d_wchain_hi, wchain_hi.data(), wchain_hi.size() * sizeof(uint32_t), 0});
        }
        size_t total = 0, payload = 0;
        for (Part& p : parts) { p.at = total; total += (std::max<size_t>(p.bytes, 1) + 255) & ~(size_t)255; payload += p.bytes; }
        if (e == hipSuccess) e = hipMalloc((void**)&tax->d_block, total);
        for (Part& p : parts) {
            if (e != hipSuccess) break;
            *p.view = tax->d_block + p.at;
            if (p.bytes) e = hipMemcpy(*p.view, p.src, p.bytes, hipMemcpyHostToDevice);
        }
        if (e != hipSuccess) {
            set_error("HIP error while uploading the taxonomy: %s", hipGetErrorString(e));
            blu_taxonomy_destroy(tax);
            return BLU_ERR_HIP;
        }
        tax->device_bytes = payload;
    }
    *out = tax;
    return BLU_OK;
}

void blu_taxonomy_destroy(blu_taxonomy* tax) {
    if (!tax) return;
    if (tax->device >= 0) {
        (void)hipSetDevice(tax->device);
        if (tax->d_block) (void)hipFree(tax->d_block);      // (d_lin ... d_wchain_hi are views into it)
        if (tax->ws_worklist) (void)hipFree(tax->ws_worklist);
        if (tax->ws_count) (void)hipFree(tax->ws_count);
        if (tax->ws_pack_flag) (void)hipFree(tax->ws_pack_flag);
        if (tax->ws_kind_host) (void)hipHostFree(tax->ws_kind_host);
        for (auto& set : tax->ws_stage) for (void* p : set) if (p) (void)hipFree(p);
    }
    delete tax;
}

uint64_t blu_taxonomy_n_tax(const blu_taxonomy* tax) { return tax ? tax->n_tax : 0; }
uint32_t blu_taxonomy_n_shapes(const blu_taxonomy* tax) { return tax ? tax->n_shapes : 0; }
uint32_t blu_taxonomy_n_rank_codes(const blu_taxonomy* tax) { return tax ? (uint32_t)tax->ranks.size() : 0; }
uint32_t blu_taxonomy_max_depth(const blu_taxonomy* tax) { return tax ? tax->max_depth : 0; }
uint64_t blu_taxonomy_device_bytes(const blu_taxonomy* tax) { return tax ? tax->device_bytes : 0; }

const char* blu_taxonomy_rank_name(const blu_taxonomy* tax, uint32_t rank_code, int serde) {
    if (!tax || rank_code >= tax->ranks.size()) return nullptr;
    return serde ? tax->ranks[rank_code].serde.c_str() : tax->ranks[rank_code].display.c_str();
}

int32_t blu_taxonomy_row_cutoffs(const blu_taxonomy* tax, uint64_t tax_row, uint32_t cap, double* cutoff,
                                 uint8_t* is_default, uint16_t* rank_code) {
    if (!tax || tax_row >= tax->n_tax) return -1;
    uint32_t hdr = tax->h_lin[(size_t)tax_row * tax->stride];
    uint32_t len = hdr & 0xFF, shape = hdr >> 8;
    for (uint32_t j = 0; j < len && j < cap; ++j) {
        size_t k = (size_t)shape * tax->sc + j;
        if (cutoff) cutoff[j] = tax->h_cut[k];
        if (is_default) is_default[j] = tax->h_isdef[k];
        if (rank_code) rank_code[j] = (uint16_t)(tax->h_codes[k] & 0xFFFF);
    }
    return (int32_t)len;
}

int blu_taxonomy_lookup(const blu_taxonomy* tax, const int64_t* taxid, uint64_t n, uint32_t* out_row) {
    if (!tax || (n && (!taxid || !out_row))) { set_error("null argument"); return BLU_ERR_INVALID_ARG; }
    if (tax->taxid_row.empty() && tax->n_tax) { set_error("taxonomy was created without taxids"); return BLU_ERR_INVALID_ARG; }
    for (uint64_t i = 0; i < n; ++i) {
        auto it = tax->taxid_row.find(taxid[i]);
        out_row[i] = it == tax->taxid_row.end() ? BLU_UNMATCHED_TAXID : tax->pos_of[it->second];
    }
    return BLU_OK;
}

int blu_taxonomy_shared_levels(const blu_taxonomy* tax, uint32_t lo, uint32_t hi, uint32_t* by_scan, uint32_t* by_tables, int32_t* via) {
    if (!tax || lo > hi || hi >= tax->n_tax) { set_error("blu_taxonomy_shared_levels: need lo <= hi < n_tax"); return BLU_ERR_INVALID_ARG; }
    const std::vector<uint8_t>& lcp = tax->h_lcp8;
    if (by_scan) {
        uint32_t m = 0xFFu;
        for (uint32_t i = lo; i < hi; ++i) m = std::min<uint32_t>(m, lcp[i]);
        *by_scan = lo == hi ? (tax->h_lin[(size_t)tax->order[lo] * tax->stride] & 0xFFu) : m;
    }
    if (by_tables || via) {
        uint32_t r = 0;
        int32_t v = 0;
        if (lo == hi) r = tax->h_lin[(size_t)tax->order[lo] * tax->stride] & 0xFFu;
        else if (hi - lo >= BLU_WIDE_MIN && !tax->h_wblk.empty()) {
            // what phase 2c of the stream kernel does (consensus_kernel.hip: wide_node, chain_count)
            const uint2 e = tax->h_wblk[lo >> BLU_WBLK_SHIFT];
            const uint32_t o = lo & ((1u << BLU_WBLK_SHIFT) - 1u), s1 = (e.y >> 16) & 0xFFu, s2 = e.y >> 24;
            if (s1 != BLU_WBLK_OVERFLOW) {
                const uint32_t w = o >= s2 ? (e.y & 0xFFFFu) : (o >= s1 ? e.x >> 16 : e.x & 0xFFFFu);
                for (uint32_t i = 0; i < BLU_WCHAIN; ++i) r += tax->h_wchain[(size_t)w * BLU_WCHAIN + i] > hi;
                if (r == BLU_WCHAIN && !tax->h_wchain_hi.empty())
                    for (uint32_t i = 0; i < BLU_WCHAIN; ++i) r += tax->h_wchain_hi[(size_t)w * BLU_WCHAIN + i] > hi;
                v = 1;
            }
        }
        if (lo != hi && v == 0) {
            // the sparse table over 16-entry blocks of lcp8 (consensus_kernel.hip: shared_levels)
            const uint32_t b0 = (lo + 15) >> 4, b1 = hi >> 4;
            uint32_t m = 0xFFu;
            if (b0 > b1) { for (uint32_t i = lo; i < hi; ++i) m = std::min<uint32_t>(m, lcp[i]); }
            else {
                for (uint32_t i = lo; i < (b0 << 4); ++i) m = std::min<uint32_t>(m, lcp[i]);
                for (uint32_t i = b1 << 4; i < hi; ++i) m = std::min<uint32_t>(m, lcp[i]);
                if (b0 < b1) {
                    uint32_t k = 0;
                    while ((2u << k) <= b1 - b0) ++k;
                    m = std::min<uint32_t>(m, tax->h_rmq[(size_t)k * tax->rmq_nb + b0]);
                    m = std::min<uint32_t>(m, tax->h_rmq[(size_t)k * tax->rmq_nb + b1 - (1u << k)]);
                }
            }
            r = m;
        }
        if (by_tables) *by_tables = r;
        if (via) *via = v;
    }
    return BLU_OK;
}

int blu_taxonomy_trim(const blu_taxonomy* tax) {
    if (!tax) { set_error("null argument"); return BLU_ERR_INVALID_ARG; }
    if (tax->device < 0) return BLU_OK;
    if (hipSetDevice(tax->device) != hipSuccess) { set_error("hipSetDevice(%d) failed", tax->device); return BLU_ERR_NO_DEVICE; }
    for (auto& set : tax->ws_stage) for (void*& p : set) if (p) { (void)hipFree(p); p = nullptr; }
    for (auto& set : tax->ws_stage_bytes) for (size_t& b : set) b = 0;
    return BLU_OK;
}

int blu_taxonomy_row_map(const blu_taxonomy* tax, uint32_t* out_map, uint32_t* out_inverse) {
    if (!tax) { set_error("null argument"); return BLU_ERR_INVALID_ARG; }
    if (out_map) memcpy(out_map, tax->pos_of.data(), tax->pos_of.size() * sizeof(uint32_t));
    if (out_inverse) memcpy(out_inverse, tax->order.data(), tax->order.size() * sizeof(uint32_t));
    return BLU_OK;
}

}  // extern "C"
