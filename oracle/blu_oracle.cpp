// ============================================================================
// oracle/blu_oracle.cpp — TEST INFRASTRUCTURE ONLY.
//
// String-faithful CPU restatement of blutils' per-query taxonomic consensus
// (reference @ 8.3.1, pure Rust, cannot be compiled here: no cargo/rustc).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load this; the product library (blutils_amd/csrc) never links or calls it.
//
// Parity pin: golden-derived vectors from the reference's own
// test/mock/output/zymo-mock/blutils.consensus.json (reconstruction recipe,
// SURVEY §8c) and the worked example in docs/book/02_*.md:192-249 — see
// tests/test_oracle_golden.py.  Third-party edges (polars CSV parsing,
// slugify of non-ASCII rank names) are "parity unpinned": no reference test
// pins them.
//
// Written in C++17 (g++) rather than plain C because the reference is built
// from Vec<String>/HashMap<String,_> values; std::string/std::vector restate
// them 1:1.  Every function cites the reference file:line it follows
// (paths relative to /root/reference/core/src).
// ============================================================================
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

namespace {

// ---- panic sites become exceptions carrying a status code ------------------
enum Status : int {
    ST_CONSENSUS = 0,        // ConsensusResult::ConsensusFound
    ST_NO_CONSENSUS = 1,     // ConsensusResult::NoConsensusFound
    ST_PANIC_PARSE = 2,      // find_single_query_consensus.rs:58-60 (parse_taxonomy Err)
    ST_PANIC_SINGLE_EMPTY = 3,   // find_single_query_consensus.rs:113-119
    ST_PANIC_ROOT_DISAGREE = 4,  // find_multi_taxa_consensus.rs:181 (index - 1 underflow)
    ST_PANIC_INTERP_LEN = 5,     // find_multi_taxa_consensus.rs:122-127
    ST_PANIC_CUSTOM_MISSING = 6, // domain/dtos/taxon.rs:117
    ST_PANIC_NAN_SORT = 7,       // find_multi_taxa_consensus.rs:86-88 partial_cmp().unwrap()
    ST_PANIC_OTHER = 8,
};

struct Panic : std::runtime_error {
    int code;
    Panic(int c, const std::string& m) : std::runtime_error(m), code(c) {}
};

// ---- domain/dtos/linnaean_ranks.rs:16-29  LinnaeanRank ---------------------
enum RankKind { Undefined, Domain, Kingdom, Phylum, Class, Order, Family, Genus, Species, Other };

struct LinnaeanRank {
    RankKind kind = Undefined;
    std::string other;  // payload of Other(String)
    bool operator==(const LinnaeanRank& o) const {  // #[derive(PartialEq)]
        return kind == o.kind && (kind != Other || other == o.other);
    }
    bool operator!=(const LinnaeanRank& o) const { return !(*this == o); }
};

// slugify 0.1 (third-party, not under /root/reference): ASCII restatement of
// its published behaviour — lowercase, [a-z0-9] kept, every other run of
// characters becomes one '-', leading/trailing '-' trimmed.  Non-ASCII
// transliteration (unidecode) is NOT restated: parity unpinned there.
std::string slugify(const std::string& s) {
    std::string out;
    bool dash = false;
    for (unsigned char c : s) {
        if (c >= 'A' && c <= 'Z') c = (unsigned char)(c - 'A' + 'a');
        if ((c >= 'a' && c <= 'z') || (c >= '0' && c <= '9')) {
            if (dash && !out.empty()) out.push_back('-');
            dash = false;
            out.push_back((char)c);
        } else {
            dash = true;
        }
    }
    return out;
}

// linnaean_ranks.rs:52-72  impl FromStr for LinnaeanRank
LinnaeanRank rank_from_str(const std::string& input) {
    std::string low;
    for (unsigned char c : input) low.push_back((c >= 'A' && c <= 'Z') ? (char)(c - 'A' + 'a') : (char)c);
    size_t a = 0, b = low.size();
    while (a < b && std::isspace((unsigned char)low[a])) ++a;
    while (b > a && std::isspace((unsigned char)low[b - 1])) --b;
    std::string t = low.substr(a, b - a);
    LinnaeanRank r;
    if (t == "u" || t == "undefined") r.kind = Undefined;
    else if (t == "d" || t == "domain") r.kind = Domain;
    else if (t == "k" || t == "kingdom") r.kind = Kingdom;
    else if (t == "p" || t == "phylum") r.kind = Phylum;
    else if (t == "c" || t == "class") r.kind = Class;
    else if (t == "o" || t == "order") r.kind = Order;
    else if (t == "f" || t == "family") r.kind = Family;
    else if (t == "g" || t == "genus") r.kind = Genus;
    else if (t == "s" || t == "species") r.kind = Species;
    else { r.kind = Other; r.other = slugify(t); }
    return r;
}

// linnaean_ranks.rs:74-89  impl Display
std::string rank_display(const LinnaeanRank& r) {
    switch (r.kind) {
        case Domain: return "d"; case Kingdom: return "k"; case Phylum: return "p";
        case Class: return "c"; case Order: return "o"; case Family: return "f";
        case Genus: return "g"; case Species: return "s"; case Undefined: return "u";
        default: return r.other;
    }
}

// serde: #[serde(rename_all = "camelCase")] + #[serde(untagged)] Other (linnaean_ranks.rs:14-29)
std::string rank_serde(const LinnaeanRank& r) {
    switch (r.kind) {
        case Domain: return "domain"; case Kingdom: return "kingdom"; case Phylum: return "phylum";
        case Class: return "class"; case Order: return "order"; case Family: return "family";
        case Genus: return "genus"; case Species: return "species"; case Undefined: return "undefined";
        default: return r.other;
    }
}

// linnaean_ranks.rs:109-114  RankedLinnaeanIdentity
struct RankedIdentity {
    bool is_default = false;   // DefaultRank(LinnaeanRank, f64) | NonDefaultRank(String, f64)
    LinnaeanRank rank;         // DefaultRank payload
    std::string name;          // NonDefaultRank payload
    double identity = 0.0;
    bool operator==(const RankedIdentity& o) const {  // #[derive(PartialEq)]
        if (is_default != o.is_default) return false;
        if (is_default) return rank == o.rank && identity == o.identity;
        return name == o.name && identity == o.identity;
    }
    bool operator!=(const RankedIdentity& o) const { return !(*this == o); }
};

// domain/dtos/taxon.rs:16-25 CustomTaxon ; taxon.rs:68-87 Taxon
struct CustomTaxon {
    int16_t v[8];       // domain kingdom phylum class order family genus species
    uint8_t has[8];     // Option<i16> for the six middle ranks (domain/species mandatory)
};
enum Taxon { Fungi = 0, Bacteria = 1, Eukaryotes = 2, Custom = 3 };

RankedIdentity def(RankKind k, double v) {
    RankedIdentity r; r.is_default = true; r.rank.kind = k; r.identity = v; return r;
}

// taxon.rs:104-185  Taxon::get_taxon_cutoff + the four tables
std::vector<RankedIdentity> get_taxon_cutoff(int taxon, const CustomTaxon* custom) {
    switch (taxon) {
        case Fungi:       // taxon.rs:144-154
        case Eukaryotes:  // taxon.rs:174-184 (same numbers)
            return {def(Species, 97.0), def(Genus, 95.0), def(Family, 90.0), def(Order, 85.0),
                    def(Class, 80.0), def(Phylum, 75.0), def(Domain, 60.0)};
        case Bacteria:    // taxon.rs:159-169
            return {def(Species, 99.0), def(Genus, 97.0), def(Family, 92.0), def(Order, 85.0),
                    def(Class, 80.0), def(Phylum, 75.0), def(Domain, 60.0)};
        case Custom: {    // taxon.rs:113-118, 123-139
            if (!custom) throw Panic(ST_PANIC_CUSTOM_MISSING, "Custom taxon values are required");
            static const RankKind order[8] = {Domain, Kingdom, Phylum, Class, Order, Family, Genus, Species};
            std::vector<RankedIdentity> out;
            for (int i = 0; i < 8; ++i) {
                bool mandatory = (i == 0 || i == 7);
                int16_t v = (mandatory || custom->has[i]) ? custom->v[i] : (int16_t)0;  // unwrap_or(0)
                out.push_back(def(order[i], (double)v));
            }
            return out;
        }
    }
    throw Panic(ST_PANIC_OTHER, "unknown taxon");
}

// domain/utils/mod.rs:1-4
double round_dec(double value, unsigned decimals) {
    double y = (double)(int32_t)std::pow(10.0, (double)decimals);
    return std::round(value * y) / y;  // f64::round = half away from zero = C round()
}

// linnaean_ranks.rs:220-383  InterpolatedIdentity::interpolate_identities
std::vector<RankedIdentity> interpolate_identities(int taxon, const std::vector<LinnaeanRank>& taxonomy,
                                                   const CustomTaxon* custom) {
    std::vector<RankedIdentity> backbone = get_taxon_cutoff(taxon, custom);           // :231
    std::vector<RankedIdentity> mapped;                                               // :239-261
    for (const LinnaeanRank& rank : taxonomy) {
        RankedIdentity binding; binding.is_default = false; binding.name = rank_display(rank); binding.identity = 0.0;
        const RankedIdentity* found = &binding;
        for (const RankedIdentity& level : backbone)
            if (level.is_default && level.rank == rank) { found = &level; break; }
        mapped.push_back(*found);
    }
    bool all_default = true;                                                          // :265-270
    for (auto& r : mapped) if (!r.is_default) { all_default = false; break; }
    if (all_default) return mapped;

    const size_t n = mapped.size();
    std::vector<std::pair<size_t, double>> updated;                                   // :335 HashMap<i32,f64>
    for (size_t nd = 0; nd < n; ++nd) {                                               // :275-333
        if (mapped[nd].is_default) continue;
        // previous: nearest preceding default rank, else element 0        (:292-300)
        const RankedIdentity* previous = &mapped[0];
        for (size_t j = nd; j-- > 0;) if (mapped[j].is_default) { previous = &mapped[j]; break; }
        // previous_index: FIRST position equal to previous               (:302-305)
        size_t previous_index = 0;
        for (size_t j = 0; j < n; ++j) if (mapped[j] == *previous) { previous_index = j; break; }
        // next: first default rank at or after nd, else last element      (:307-317)
        const RankedIdentity* next = &mapped[n - 1];
        for (size_t j = nd; j < n; ++j) if (mapped[j].is_default) { next = &mapped[j]; break; }
        // next_index: FIRST position equal to next                        (:319-322)
        size_t next_index = n - 1;
        for (size_t j = 0; j < n; ++j) if (mapped[j] == *next) { next_index = j; break; }
        // window = skip_while(level != previous).take(next_index + 1)     (:324-329)
        std::vector<RankedIdentity> window;
        {
            size_t j = 0;
            while (j < n && mapped[j] != *previous) ++j;
            for (size_t t = 0; t < next_index + 1 && j < n; ++t, ++j) window.push_back(mapped[j]);
        }
        size_t target_index = nd - previous_index;                                    // :339
        double first_window_identity;                                                 // :341-347
        if (window[0].is_default) first_window_identity = window[0].identity;
        else first_window_identity = backbone[0].identity;  // backbone[0] is always DefaultRank
        double last_window_identity =                                                 // :349-353
            window[window.size() - 1].is_default ? window[window.size() - 1].identity : 100.0;
        double window_weight = last_window_identity - first_window_identity;          // :355
        double window_size = (double)(window.size() - 1);                             // :356
        double target_identity =                                                      // :358-362
            round_dec(first_window_identity + ((double)target_index * (window_weight / window_size)), 3);
        updated.emplace_back(nd, target_identity);
    }
    for (size_t i = 0; i < n; ++i) {                                                  // :368-382
        if (mapped[i].is_default) continue;
        double identity = 100.0;
        for (auto& kv : updated) if (kv.first == i) identity = kv.second;
        mapped[i].identity = identity;
    }
    return mapped;
}

// domain/dtos/consensus_result.rs:38-46  ConsensusBean
struct ConsensusBean {
    LinnaeanRank rank;
    std::string identifier;
    int32_t occurrences = 0;
    std::string taxonomy;
    std::vector<std::string> accessions;
};

// domain/dtos/taxonomy_bean.rs:7-19  TaxonomyBean
struct TaxonomyBean {
    LinnaeanRank reached_rank;
    bool has_max_allowed_rank = false;
    LinnaeanRank max_allowed_rank;
    std::string identifier;
    double perc_identity = 0.0;
    double bit_score = 0.0;
    bool has_taxonomy = false;
    std::string taxonomy;
    bool mutated = false;
    bool single_match = false;
    bool has_beans = false;
    std::vector<ConsensusBean> consensus_beans;
    std::string taxonomy_to_string() const {  // taxonomy_bean.rs:22-27
        return rank_display(reached_rank) + "__" + identifier;
    }
};

// taxonomy_bean.rs:38-45
std::string taxonomy_beans_to_string(const std::vector<TaxonomyBean>& t) {
    std::string s;
    for (size_t i = 0; i < t.size(); ++i) { if (i) s += ";"; s += t[i].taxonomy_to_string(); }
    return s;
}

// domain/dtos/blast_result.rs:12-26  BlastResultRow (the fields the path reads)
struct BlastResultRow {
    std::string subject_accession;
    int64_t subject_taxid = 0;
    double perc_identity = 0.0;
    int64_t align_length = 0;
    int64_t bit_score = 0;
    std::string taxonomy_literal;          // Taxonomy::Literal
    bool parsed = false;                   // Taxonomy::Parsed
    std::vector<TaxonomyBean> taxonomy;
};

std::vector<std::string> split(const std::string& s, const std::string& sep) {  // str::split
    std::vector<std::string> out;
    size_t pos = 0;
    for (;;) {
        size_t f = s.find(sep, pos);
        if (f == std::string::npos) { out.push_back(s.substr(pos)); break; }
        out.push_back(s.substr(pos, f - pos));
        pos = f + sep.size();
    }
    return out;
}

// blast_result.rs:38-120  BlastResultRow::parse_taxonomy
void parse_taxonomy(BlastResultRow& row) {
    if (row.parsed) return;
    std::vector<std::string> splitted = split(row.taxonomy_literal, ";");             // :40-49
    std::vector<TaxonomyBean> parsed;
    for (const std::string& tax : splitted) {                                         // :51-107
        std::vector<std::string> parts = split(tax, "__");
        if (parts.size() != 2) continue;                                              // :65-67 → None
        TaxonomyBean b;
        b.reached_rank = rank_from_str(parts[0]);
        b.identifier = parts[1];
        b.perc_identity = row.perc_identity;                                          // :99
        b.bit_score = (double)row.bit_score;                                          // :100
        parsed.push_back(std::move(b));
    }
    if (parsed.size() != splitted.size())                                             // :109-114
        throw Panic(ST_PANIC_PARSE, "Unexpected error on parse taxonomy");
    row.taxonomy = std::move(parsed);
    row.parsed = true;
}

// consensus_result.rs:48-63  ConsensusBean::from_taxonomy_bean
ConsensusBean bean_from_taxonomy_bean(const TaxonomyBean& bean, const std::string& accession,
                                      const std::string& taxonomy) {
    ConsensusBean c;
    c.rank = bean.reached_rank; c.identifier = bean.identifier; c.occurrences = 0;
    c.taxonomy = taxonomy; c.accessions = {accession};
    return c;
}

// consensus_result.rs:65-88  ConsensusBean::fold_consensus_list
// HashMap iteration order is unspecified in the reference; first-insertion
// order is used here (the caller sorts; ties on the sort key stay unpinned).
std::vector<ConsensusBean> fold_consensus_list(const std::vector<ConsensusBean>& consensus) {
    std::vector<std::pair<std::string, ConsensusBean>> acc;
    for (const ConsensusBean& bean : consensus) {
        std::string key = rank_display(bean.rank) + "__" + bean.identifier;           // :73
        ConsensusBean* slot = nullptr;
        for (auto& kv : acc) if (kv.first == key) { slot = &kv.second; break; }
        if (!slot) {                                                                  // :74-77 or_insert(bean, occurrences 0)
            ConsensusBean fresh = bean; fresh.occurrences = 0;
            acc.emplace_back(key, fresh);
            slot = &acc.back().second;
            // or_insert clones `bean` INCLUDING its accessions; the extend below then
            // appends the same accession again and dedup() collapses the pair.
        }
        slot->accessions.insert(slot->accessions.end(), bean.accessions.begin(), bean.accessions.end());  // :79
        slot->accessions.erase(std::unique(slot->accessions.begin(), slot->accessions.end()),
                               slot->accessions.end());                               // :80 Vec::dedup (consecutive)
        slot->occurrences += 1;                                                       // :81
    }
    std::vector<ConsensusBean> out;
    for (auto& kv : acc) out.push_back(kv.second);
    return out;
}

struct InterpolatedIdentity {
    std::vector<RankedIdentity> interpolation;
    // linnaean_ranks.rs:174-192
    const RankedIdentity* get_rank_adjusted_by_identity(double identity) const {
        for (const RankedIdentity& r : interpolation)
            if (!(identity > r.identity)) return &r;   // skip_while(identity > rank_identity), first
        return nullptr;
    }
    // linnaean_ranks.rs:194-212
    std::vector<TaxonomyBean> get_adjusted_taxonomy_by_identity(double identity,
                                                                const std::vector<TaxonomyBean>& taxonomy) const {
        std::vector<TaxonomyBean> out;
        size_t n = std::min(interpolation.size(), taxonomy.size());                  // zip
        for (size_t i = 0; i < n; ++i)
            if (identity >= interpolation[i].identity) out.push_back(taxonomy[i]);
        return out;
    }
};

struct QueryWithConsensus {
    bool has_taxon = false;
    TaxonomyBean taxon;
};

// build_consensus_identities/build_blast_consensus_identity.rs:9-105
QueryWithConsensus build_blast_consensus_identity(TaxonomyBean bean, double max_allowed_identity,
                                                  bool target_as_single_match, size_t bean_index,
                                                  const std::vector<TaxonomyBean>& taxonomy,
                                                  const InterpolatedIdentity& interpolated,
                                                  const std::vector<ConsensusBean>& beans_in) {
    const RankedIdentity* adj = interpolated.get_rank_adjusted_by_identity(max_allowed_identity);  // :22-30
    if (adj) {
        bean.has_max_allowed_rank = true;
        if (adj->is_default) bean.max_allowed_rank = adj->rank;
        else { bean.max_allowed_rank.kind = Other; bean.max_allowed_rank.other = adj->name; }
    } else {
        bean.has_max_allowed_rank = false;
    }
    if (bean.has_max_allowed_rank) bean.mutated = bean.reached_rank != bean.max_allowed_rank;      // :35-37

    std::vector<ConsensusBean> consensus_beans = fold_consensus_list(beans_in);                    // :43-44
    if (!consensus_beans.empty()) {                                                                // :49-63
        std::stable_sort(consensus_beans.begin(), consensus_beans.end(),
                         [](const ConsensusBean& a, const ConsensusBean& b) {
                             if (a.occurrences != b.occurrences) return a.occurrences > b.occurrences;
                             return a.identifier < b.identifier;
                         });
        bean.has_beans = true;
        bean.consensus_beans = consensus_beans;
    }
    if (bean_index >= taxonomy.size())                                                             // :65,96-98
        throw Panic(ST_PANIC_OTHER, "No taxonomy found for bean at index");
    const TaxonomyBean& fallback = taxonomy[bean_index];
    std::vector<TaxonomyBean> filtered =
        interpolated.get_adjusted_taxonomy_by_identity(max_allowed_identity, taxonomy);            // :67-72
    std::vector<TaxonomyBean> adjusted;
    if (target_as_single_match && consensus_beans.size() == 1) {                                   // :74-75
        adjusted = filtered;
    } else {                                                                                       // :76-82
        for (size_t i = 0; i < filtered.size() && i <= bean_index; ++i) adjusted.push_back(filtered[i]);
    }
    const TaxonomyBean& last = adjusted.empty() ? fallback : adjusted.back();                      // :85
    bean.identifier = last.identifier;                                                             // :87-88
    bean.reached_rank = last.reached_rank;
    bean.has_taxonomy = true;                                                                      // :89-95
    bean.taxonomy = taxonomy_beans_to_string(adjusted);
    QueryWithConsensus q; q.has_taxon = true; q.taxon = std::move(bean);
    return q;
}

std::vector<LinnaeanRank> ranks_of(const std::vector<TaxonomyBean>& t) {
    std::vector<LinnaeanRank> r; for (auto& b : t) r.push_back(b.reached_rank); return r;
}

// build_consensus_identities/find_multi_taxa_consensus.rs:22-217
QueryWithConsensus find_multi_taxa_consensus(const std::vector<BlastResultRow>& records, int taxon,
                                             int strategy /*0 cautious, 1 relaxed*/, const CustomTaxon* custom) {
    std::vector<BlastResultRow> sorted = records;                                     // :39-54 (stable sort_by)
    std::stable_sort(sorted.begin(), sorted.end(), [](const BlastResultRow& a, const BlastResultRow& b) {
        if (a.taxonomy.size() != b.taxonomy.size()) return a.taxonomy.size() < b.taxonomy.size();
        // partial_cmp(...).unwrap_or(Equal): NaN compares Equal
        if (a.perc_identity < b.perc_identity) return true;
        if (a.perc_identity > b.perc_identity) return false;
        if (a.align_length != b.align_length) return a.align_length < b.align_length;
        return a.subject_accession < b.subject_accession;                             // String::cmp = bytewise
    });
    const BlastResultRow& reference = (strategy == 0) ? sorted.front() : sorted.back();   // :60-68
    const std::vector<TaxonomyBean>& reference_taxonomy = reference.taxonomy;

    // :83-91 lowest_taxonomy_of_higher_rank — sort of the first lineage by
    // perc_identity with partial_cmp().unwrap(): panics on NaN when len >= 2.
    if (sorted.front().taxonomy.size() >= 2 && std::isnan(sorted.front().perc_identity))
        throw Panic(ST_PANIC_NAN_SORT, "partial_cmp unwrap on NaN");
    QueryWithConsensus final_taxon;                                                   // :97-101
    final_taxon.has_taxon = true;
    final_taxon.taxon = sorted.front().taxonomy.front();

    InterpolatedIdentity interp;                                                      // :112-120
    interp.interpolation = interpolate_identities(taxon, ranks_of(reference_taxonomy), custom);
    if (interp.interpolation.size() != reference_taxonomy.size())                     // :122-127
        throw Panic(ST_PANIC_INTERP_LEN, "Interpolated identities length mismatch");

    for (size_t index = 0; index < reference_taxonomy.size(); ++index) {              // :137
        // take_while(index < taxonomy.len()) over the length-ascending sort  (:142-145)
        size_t take = 0;
        while (take < sorted.size() && index < sorted[take].taxonomy.size()) ++take;
        std::vector<std::string> level_set;                                           // :150-159 HashSet<String>
        for (size_t r = 0; r < take; ++r) {
            const TaxonomyBean& e = sorted[r].taxonomy[index];
            std::string key = rank_display(e.reached_rank) + e.identifier;            // "{rank}{identifier}"
            if (std::find(level_set.begin(), level_set.end(), key) == level_set.end()) level_set.push_back(key);
        }
        if (level_set.empty()) continue;                                              // :161-163
        std::vector<ConsensusBean> consensus_beans;                                   // :169-178
        for (size_t r = 0; r < take; ++r)
            consensus_beans.push_back(bean_from_taxonomy_bean(
                sorted[r].taxonomy[index], sorted[r].subject_accession,
                taxonomy_beans_to_string(sorted[r].taxonomy)));
        if (level_set.size() > 1) {                                                   // :180
            if (index == 0)                                                           // :181 `index - 1` on usize
                throw Panic(ST_PANIC_ROOT_DISAGREE, "attempt to subtract with overflow");
            size_t target_index = index - 1;
            double max_pident = 0.0;                                                  // :182-185
            for (size_t r = 0; r < take; ++r) if (sorted[r].perc_identity > max_pident) max_pident = sorted[r].perc_identity;
            final_taxon = build_blast_consensus_identity(                             // :190-199
                reference_taxonomy[target_index], max_pident, false, target_index,
                reference_taxonomy, interp, consensus_beans);
            break;                                                                    // :201
        }
        final_taxon = build_blast_consensus_identity(                                 // :204-213
            reference_taxonomy[index], reference_taxonomy[index].perc_identity, true, index,
            reference_taxonomy, interp, consensus_beans);
    }
    return final_taxon;
}

struct QueryResult {
    int status = ST_NO_CONSENSUS;
    std::string message;
    QueryWithConsensus found;
};

// build_consensus_identities/find_single_query_consensus.rs:17-173
QueryResult find_single_query_consensus(const std::vector<BlastResultRow>& result, int taxon, int strategy,
                                        const CustomTaxon* custom) {
    QueryResult out;
    // :28-44 group by bit_score, keys sorted descending; the loop below always
    // returns in its first iteration, so only the top key matters.
    std::vector<int64_t> keys;
    for (auto& r : result) if (std::find(keys.begin(), keys.end(), r.bit_score) == keys.end()) keys.push_back(r.bit_score);
    std::sort(keys.begin(), keys.end(), [](int64_t a, int64_t b) { return a > b; });
    for (int64_t score : keys) {                                                      // :50
        std::vector<BlastResultRow> matches;                                          // :51-64
        for (const BlastResultRow& r : result) {
            if (r.bit_score != score) continue;
            BlastResultRow copy = r;
            parse_taxonomy(copy);                                                     // Err → panic!
            matches.push_back(std::move(copy));
        }
        if (matches.empty()) { out.status = ST_NO_CONSENSUS; return out; }            // :68-70
        if (matches.size() == 1) {                                                    // :74-150
            const BlastResultRow& target = matches.front();
            const std::vector<TaxonomyBean>& taxonomies = target.taxonomy;            // :88-89
            InterpolatedIdentity interp;                                              // :93-101
            interp.interpolation = interpolate_identities(taxon, ranks_of(taxonomies), custom);
            std::vector<TaxonomyBean> adjusted =                                      // :105-109
                interp.get_adjusted_taxonomy_by_identity(target.perc_identity, taxonomies);
            if (adjusted.empty())                                                     // :113-119
                throw Panic(ST_PANIC_SINGLE_EMPTY, "No taxonomy found for result");
            TaxonomyBean target_bean = adjusted.back();
            ConsensusBean cb = bean_from_taxonomy_bean(target_bean, target.subject_accession,   // :123-127
                                                       taxonomy_beans_to_string(taxonomies));
            TaxonomyBean t = target_bean;                                             // :131-147 `..target_bean`
            t.single_match = true;
            t.has_taxonomy = true;
            t.taxonomy = taxonomy_beans_to_string(adjusted);
            t.has_beans = true;
            t.consensus_beans = fold_consensus_list({cb});
            out.status = ST_CONSENSUS; out.found.has_taxon = true; out.found.taxon = std::move(t);
            return out;
        }
        out.found = find_multi_taxa_consensus(matches, taxon, strategy, custom);      // :154-165
        out.status = ST_CONSENSUS;
        return out;
    }
    out.status = ST_NO_CONSENSUS;                                                     // :172
    return out;
}

// ---------------------------------------------------------------------------
// JSON rendering of one result (serde camelCase names: taxonomy_bean.rs:5-19,
// consensus_result.rs:36-46).  Used by tests to compare field by field.
// ---------------------------------------------------------------------------
void json_escape(std::string& o, const std::string& s) {
    o.push_back('"');
    for (unsigned char c : s) {
        if (c == '"' || c == '\\') { o.push_back('\\'); o.push_back((char)c); }
        else if (c < 0x20) { char b[8]; std::snprintf(b, sizeof b, "\\u%04x", c); o += b; }
        else o.push_back((char)c);
    }
    o.push_back('"');
}
void json_double(std::string& o, double v) {
    if (std::isnan(v) || std::isinf(v)) { o += "null"; return; }
    char b[40]; std::snprintf(b, sizeof b, "%.17g", v); o += b;
}

std::string result_to_json(const QueryResult& r) {
    std::string o = "{\"status\":" + std::to_string(r.status);
    if (r.status >= ST_PANIC_PARSE) { o += ",\"panic\":"; json_escape(o, r.message); }
    o += ",\"taxon\":";
    if (r.status != ST_CONSENSUS || !r.found.has_taxon) { o += "null}"; return o; }
    const TaxonomyBean& t = r.found.taxon;
    o += "{\"reachedRank\":"; json_escape(o, rank_serde(t.reached_rank));
    o += ",\"maxAllowedRank\":";
    if (t.has_max_allowed_rank) json_escape(o, rank_serde(t.max_allowed_rank)); else o += "null";
    o += ",\"identifier\":"; json_escape(o, t.identifier);
    o += ",\"percIdentity\":"; json_double(o, t.perc_identity);
    o += ",\"bitScore\":"; json_double(o, t.bit_score);
    o += ",\"taxonomy\":"; if (t.has_taxonomy) json_escape(o, t.taxonomy); else o += "null";
    o += ",\"mutated\":"; o += t.mutated ? "true" : "false";
    o += ",\"singleMatch\":"; o += t.single_match ? "true" : "false";
    o += ",\"consensusBeans\":";
    if (!t.has_beans) o += "null";
    else {
        o += "[";
        for (size_t i = 0; i < t.consensus_beans.size(); ++i) {
            const ConsensusBean& b = t.consensus_beans[i];
            if (i) o += ",";
            o += "{\"rank\":"; json_escape(o, rank_serde(b.rank));
            o += ",\"identifier\":"; json_escape(o, b.identifier);
            o += ",\"occurrences\":" + std::to_string(b.occurrences);
            o += ",\"taxonomy\":"; json_escape(o, b.taxonomy);
            o += ",\"accessions\":[";
            for (size_t k = 0; k < b.accessions.size(); ++k) { if (k) o += ","; json_escape(o, b.accessions[k]); }
            o += "]}";
        }
        o += "]";
    }
    o += "}}";
    return o;
}

}  // namespace

// ===========================================================================
// C interface (ctypes).  Hit rows arrive grouped by query (seg_off), in file
// order within a query, each row carrying what the polars left join gives it
// (mod.rs:72-76, 134-221): an accession string, a lineage string (or the
// literal "null" for an unmatched taxid, mod.rs:185), perc_identity,
// align_length and the already-truncated i64 bit_score (mod.rs:184).
// ===========================================================================
extern "C" {

struct blu_oracle_cfg {
    int32_t taxon;          // 0 fungi, 1 bacteria, 2 eukaryotes, 3 custom
    int32_t strategy;       // 0 cautious, 1 relaxed
    int32_t has_custom;
    int16_t custom[8];      // domain kingdom phylum class order family genus species
    uint8_t custom_has[8];
    int32_t threads;        // workers over queries (mod.rs:104-128 rayon global pool)
};

struct blu_oracle_results {
    std::vector<QueryResult> r;
};

blu_oracle_results* blu_oracle_run(uint64_t n_queries, const uint64_t* seg_off,
                                   const uint32_t* acc_idx, const char* const* acc_table,
                                   const int64_t* tax_row, const char* const* lineage_table,
                                   const int64_t* subject_taxid,
                                   const double* pident, const int64_t* align_len, const int64_t* bit_score,
                                   const blu_oracle_cfg* cfg) {
    auto* res = new blu_oracle_results();
    res->r.resize(n_queries);
    CustomTaxon custom{};
    const CustomTaxon* cp = nullptr;
    if (cfg->has_custom) {
        for (int i = 0; i < 8; ++i) { custom.v[i] = cfg->custom[i]; custom.has[i] = cfg->custom_has[i]; }
        cp = &custom;
    }
    int nthreads = cfg->threads > 0 ? cfg->threads : 1;
    std::atomic<uint64_t> next{0};
    auto worker = [&]() {
        const uint64_t chunk = 64;
        for (;;) {
            uint64_t q0 = next.fetch_add(chunk);
            if (q0 >= n_queries) break;
            uint64_t q1 = std::min<uint64_t>(n_queries, q0 + chunk);
            for (uint64_t q = q0; q < q1; ++q) {
                QueryResult& out = res->r[q];
                try {
                    // mod.rs:192-208: per-query Vec<BlastResultRow>, strings owned per row
                    std::vector<BlastResultRow> rows;
                    rows.reserve(seg_off[q + 1] - seg_off[q]);
                    for (uint64_t i = seg_off[q]; i < seg_off[q + 1]; ++i) {
                        BlastResultRow r;
                        r.subject_accession = acc_table[acc_idx[i]];
                        r.subject_taxid = subject_taxid ? subject_taxid[i] : 0;
                        r.perc_identity = pident[i];
                        r.align_length = align_len[i];
                        r.bit_score = bit_score[i];
                        r.taxonomy_literal = tax_row[i] < 0 ? std::string("null") : std::string(lineage_table[tax_row[i]]);
                        rows.push_back(std::move(r));
                    }
                    if (rows.empty()) { out.status = ST_NO_CONSENSUS; continue; }   // mod.rs:107-113
                    out = find_single_query_consensus(rows, cfg->taxon, cfg->strategy, cp);
                } catch (const Panic& p) {
                    out = QueryResult(); out.status = p.code; out.message = p.what();
                } catch (const std::exception& e) {
                    out = QueryResult(); out.status = ST_PANIC_OTHER; out.message = e.what();
                }
            }
        }
    };
    if (nthreads == 1) worker();
    else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nthreads; ++t) pool.emplace_back(worker);
        for (auto& t : pool) t.join();
    }
    return res;
}

int32_t blu_oracle_status(const blu_oracle_results* res, uint64_t q) { return res->r[q].status; }

// malloc'd JSON text of result q (caller frees with blu_oracle_free_str)
char* blu_oracle_result_json(const blu_oracle_results* res, uint64_t q) {
    std::string s = result_to_json(res->r[q]);
    char* p = (char*)std::malloc(s.size() + 1);
    std::memcpy(p, s.c_str(), s.size() + 1);
    return p;
}

// All results as one JSON array (faster for large Q)
char* blu_oracle_results_json(const blu_oracle_results* res) {
    std::string s = "[";
    for (size_t q = 0; q < res->r.size(); ++q) { if (q) s += ",\n"; s += result_to_json(res->r[q]); }
    s += "]";
    char* p = (char*)std::malloc(s.size() + 1);
    std::memcpy(p, s.c_str(), s.size() + 1);
    return p;
}

void blu_oracle_free_str(char* p) { std::free(p); }
void blu_oracle_free(blu_oracle_results* res) { delete res; }

// Cutoffs for one rank sequence (linnaean_ranks.rs:154-162,220-383): ranks are
// the rank strings as they appear in a lineage.  Writes n doubles and, per
// level, 1 if the level mapped to a DefaultRank of the backbone.
int32_t blu_oracle_interpolate(int32_t taxon, int32_t has_custom, const int16_t* custom, const uint8_t* custom_has,
                               int32_t n, const char* const* ranks, double* out_cutoff, uint8_t* out_is_default) {
    try {
        CustomTaxon c{}; const CustomTaxon* cp = nullptr;
        if (has_custom) { for (int i = 0; i < 8; ++i) { c.v[i] = custom[i]; c.has[i] = custom_has[i]; } cp = &c; }
        std::vector<LinnaeanRank> rk;
        for (int i = 0; i < n; ++i) rk.push_back(rank_from_str(ranks[i]));
        std::vector<RankedIdentity> v = interpolate_identities(taxon, rk, cp);
        for (int i = 0; i < n; ++i) { out_cutoff[i] = v[i].identity; if (out_is_default) out_is_default[i] = v[i].is_default; }
        return 0;
    } catch (const Panic& p) { return p.code; }
}

// Rank-string helpers exposed for tests of a6 (linnaean_ranks.rs:52-89).
int32_t blu_oracle_rank_display(const char* rank, char* buf, int32_t buflen) {
    std::string s = rank_display(rank_from_str(rank));
    std::snprintf(buf, buflen, "%s", s.c_str());
    return (int32_t)rank_from_str(rank).kind;
}
int32_t blu_oracle_rank_serde(const char* rank, char* buf, int32_t buflen) {
    std::string s = rank_serde(rank_from_str(rank));
    std::snprintf(buf, buflen, "%s", s.c_str());
    return (int32_t)rank_from_str(rank).kind;
}

}  // extern "C"

// ---------------------------------------------------------------------------
// Helpers for the cpu_baseline leg of bench.py: materialise the string side of
// a synthetic table (blutils DB lineage grammar `rank__identifier;...`,
// build_taxonomy_database.rs:406-465; accession `NR_%010u.1`) without a Python
// loop over millions of rows.  Data preparation only — not timed.
// ---------------------------------------------------------------------------
extern "C" {

struct blu_oracle_strtab {
    std::vector<std::string> s;
    std::vector<const char*> p;
};

blu_oracle_strtab* blu_oracle_lineage_strings(uint64_t n_tax, const uint64_t* lin_off, const uint32_t* lin_node,
                                              const uint16_t* lin_rank, const char* const* rank_names,
                                              const char* id_prefix) {
    auto* t = new blu_oracle_strtab();
    t->s.resize(n_tax);
    t->p.resize(n_tax ? n_tax : 1);
    char buf[32];
    for (uint64_t i = 0; i < n_tax; ++i) {
        std::string& o = t->s[i];
        for (uint64_t k = lin_off[i]; k < lin_off[i + 1]; ++k) {
            if (k > lin_off[i]) o.push_back(';');
            o += rank_names[lin_rank[k]];
            o += "__";
            o += id_prefix;
            std::snprintf(buf, sizeof buf, "%u", lin_node[k]);
            o += buf;
        }
    }
    for (uint64_t i = 0; i < n_tax; ++i) t->p[i] = t->s[i].c_str();
    return t;
}

blu_oracle_strtab* blu_oracle_accession_strings(uint64_t n, const uint32_t* acc_rank) {
    auto* t = new blu_oracle_strtab();
    t->s.resize(n);
    t->p.resize(n ? n : 1);
    char buf[32];
    for (uint64_t i = 0; i < n; ++i) {
        std::snprintf(buf, sizeof buf, "NR_%010u.1", acc_rank[i]);
        t->s[i] = buf;
    }
    for (uint64_t i = 0; i < n; ++i) t->p[i] = t->s[i].c_str();
    return t;
}

const char* const* blu_oracle_strtab_ptr(const blu_oracle_strtab* t) { return t->p.data(); }
void blu_oracle_strtab_free(blu_oracle_strtab* t) { delete t; }

}  // extern "C"
