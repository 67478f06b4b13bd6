/*
 * blu_consensus.h — C ABI of the MI355X-native consensus engine.
 *
 * Drop-in boundary for blutils' per-query taxonomic consensus.  The reference
 * (pure Rust, no FFI of its own) exposes this path as
 *
 *   core/src/use_cases/build_consensus_identities/mod.rs:40-47
 *     pub fn build_consensus_identities(blast_output, taxonomies_file, taxon,
 *                                       strategy, use_taxid, custom_taxon_values)
 *   core/src/use_cases/build_consensus_identities/find_single_query_consensus.rs:17-23
 *     fn find_single_query_consensus(query, result: Vec<BlastResultRow>, taxon,
 *                                    strategy, custom_taxon_values)
 *
 * A Rust shim replacing the rayon map at mod.rs:104-128 binds exactly the
 * entry points below (see INTEGRATION.md for the `extern "C"` block).  Plain
 * pointers and sizes only; no exceptions or aborts cross this boundary: every
 * reference panic site becomes a per-query status (blu_status) or a call-level
 * error code (blu_error).
 */
#ifndef BLU_CONSENSUS_H
#define BLU_CONSENSUS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BLU_ABI_VERSION 5u
#define BLU_UNMATCHED_TAXID 0xFFFFFFFFu /* hit whose subject_taxid is not in the taxonomy (left join miss, mod.rs:72-76) */
#define BLU_MAX_DEPTH 64u               /* level_mask is 64 bits wide */
#define BLU_ROW_BITS 25u                /* engine row id = sorted position | lineage length << 25: at most 2^25 taxids */
#define BLU_NONE_U8 0xFFu
#define BLU_NONE_U16 0xFFFFu
#define BLU_MAR_NEVER_EQUAL 0xFFFEu     /* Other("k")/Other("u"): a default-letter rank outside the backbone */

/* call-level error codes (return values) */
enum blu_error {
    BLU_OK = 0,
    BLU_ERR_INVALID_ARG = 1,
    BLU_ERR_NO_DEVICE = 2,      /* HIP runtime/device missing: the engine has no CPU fallback */
    BLU_ERR_HIP = 3,
    BLU_ERR_DEPTH = 4,          /* a lineage deeper than BLU_MAX_DEPTH */
    BLU_ERR_CUSTOM_MISSING = 5, /* Taxon::Custom without values (domain/dtos/taxon.rs:117) */
    BLU_ERR_ALLOC = 6,
    BLU_ERR_IO = 7,
    BLU_ERR_PARSE = 8
};

/* domain/dtos/taxon.rs:68-87 */
enum blu_taxon { BLU_TAXON_FUNGI = 0, BLU_TAXON_BACTERIA = 1, BLU_TAXON_EUKARYOTES = 2, BLU_TAXON_CUSTOM = 3 };
/* domain/dtos/consensus_strategy.rs:4-10 */
enum blu_strategy { BLU_CAUTIOUS = 0, BLU_RELAXED = 1 };

/* Taxon + Option<CustomTaxon> (domain/dtos/taxon.rs:16-25): value order is
 * domain, kingdom, phylum, class, order, family, genus, species. */
typedef struct blu_cutoff_config {
    int32_t taxon;         /* enum blu_taxon */
    int32_t has_custom;    /* Option<CustomTaxon> is Some */
    int16_t custom[8];
    uint8_t custom_has[8]; /* the six middle ranks are Option<i16>; unwrap_or(0) when 0 */
} blu_cutoff_config;

/* Taxonomy table: one row per taxid of the blutils DB (a3), lineage as CSR of
 * interned (rank, identifier) node ids, root -> leaf (a6).  rank_names[] are
 * the rank strings as they appear in lineages ("d", "clade", "species-group"
 * ...); the library applies LinnaeanRank::from_str (linnaean_ranks.rs:52-72).
 * lin_node must be interned on the CANONICAL pair (Display(rank), identifier),
 * the key the reference compares levels on (find_multi_taxa_consensus.rs:150-159). */
typedef struct blu_taxonomy_desc {
    uint64_t n_tax;
    const int64_t* taxid;          /* [n_tax] or NULL */
    const uint64_t* lin_off;       /* [n_tax + 1] */
    const uint32_t* lin_node;      /* [lin_off[n_tax]] */
    const uint16_t* lin_rank;      /* [lin_off[n_tax]] index into rank_names */
    uint32_t n_ranks;
    const char* const* rank_names; /* [n_ranks] */
    const uint8_t* bad;            /* [n_tax] or NULL; 1 = lineage string fails parse_taxonomy (blast_result.rs:109-114) */
} blu_taxonomy_desc;

typedef struct blu_taxonomy blu_taxonomy; /* opaque; owns the device copy */

/* Hit table, SoA, rows grouped by query with in-query FILE ORDER preserved
 * (stable-sort ties depend on it, find_multi_taxa_consensus.rs:39-68).  Only
 * the columns the reference semantics read; e_value and the six
 * coordinate columns are dead inputs and are not part of the layout. */
typedef struct blu_hits {
    const int32_t* bitscore;   /* [n_hits] bit_score truncated toward zero to integer (mod.rs:184) */
    const uint32_t* tax_row;   /* [n_hits] ENGINE row id of subject_taxid (blu_taxonomy_lookup / blu_taxonomy_row_map:
                                  the left join of mod.rs:72-76) or BLU_UNMATCHED_TAXID.  Engine row ids are opaque:
                                  the row's rank in lexicographic lineage order (low BLU_ROW_BITS bits) and its lineage
                                  length (bits above); they are NOT the desc row indices. */
    const double* pident;      /* [n_hits] perc_identity as f64, or NULL when pident_milli is given */
    const int32_t* align_len;  /* [n_hits] */
    const uint32_t* acc_rank;  /* [n_hits] order-preserving rank of subject_accession (bytewise String::cmp) */
    const uint64_t* seg_off;   /* [n_queries + 1] row offsets, seg_off[0] = 0, seg_off[n_queries] = n_hits */
    uint64_t n_hits;           /* < 2^32 - 1 per call (0xFFFFFFFF is the "no row" value of blu_result.ref_row) */
    uint64_t n_queries;
    int32_t on_device;         /* 1: every pointer (and `out`) is a device pointer on the handle's GPU;
                                  0: host pointers, the library stages them over PCIe */
    int32_t reserved;
    const uint32_t* pident_milli; /* [n_hits] or NULL.  Narrow lossless encoding of perc_identity for tables whose
                                  text has at most 3 decimals (BLAST outfmt 6 prints %.3f): k = perc_identity * 1000
                                  as an exact integer.  The engine rebuilds the f64 the reference's parser produces,
                                  the correctly rounded k / 1000, for the few rows that need it.  20 B/hit instead of
                                  24. */
    const uint32_t* packed;    /* [n_hits][4] or NULL (ABI v3).  The four non-bit-score values of a hit side by side,
                                  16 bytes per hit: {tax_row, pident_milli | shape hint << 17, align_len, acc_rank};
                                  tax_row, pident, pident_milli, align_len and acc_rank are then ignored (may be NULL).
                                  Same 20 B/hit as the milli-percent columns, but the engine — which reads those four
                                  values for the top-scoring rows only — finds a row's values in ONE memory line instead
                                  of four.  16-byte aligned.  Built by blu_hits_pack (ABI v4): pident_milli must be below
                                  BLU_PACKED_PIDENT_LIMIT (131.071 %; larger identities go in the column layouts), and
                                  the bits above it carry a hint for the engine — the lineage shape of the hit's taxonomy
                                  row + 1, a function of the joined taxid like the row id itself — which lets it ask for
                                  the per-level tables together with the reference row instead of after it.  A hint of 0
                                  (records put together by hand) or a wrong one costs a memory round trip, never a
                                  result: the engine checks it against the row. */
    const uint32_t* packed64;  /* [n_hits][6] or NULL (ABI v4).  The same for perc_identity values that are not exact
                                  milli-percent: 24 bytes per hit {tax_row, shape hint << 17, align_len, acc_rank,
                                  pident f64 (low word, high word)}, 8-byte aligned; built by blu_hits_pack64.  28 B/hit
                                  with the bit-score column.  Exactly one of pident / pident_milli / packed / packed64
                                  is non-NULL. */
} blu_hits;

#define BLU_PACKED_PIDENT_LIMIT 131071u /* packed layout: pident_milli < this (17 bits, the top value is the engine's "never") */

typedef struct blu_run_params {
    int32_t strategy; /* enum blu_strategy */
    int32_t flags;    /* reserved, 0 */
    void* stream;     /* hipStream_t to launch on (NULL = default stream) */
} blu_run_params;

/* per-query status: 0/1 are the two reference outcomes with a taxon, 2 is
 * NoConsensusFound, >= 16 are the reference's panic sites */
enum blu_status {
    BLU_ST_CONSENSUS_MULTI = 0,   /* find_multi_taxa_consensus outcome */
    BLU_ST_CONSENSUS_SINGLE = 1,  /* single top-score hit (find_single_query_consensus.rs:74-150) */
    BLU_ST_NO_HITS = 2,           /* empty segment: NoConsensusFound (mod.rs:107-113) */
    BLU_ST_ERR_UNMATCHED_TAXID = 16, /* top-group row whose taxid is not in the DB (find_single_query_consensus.rs:58-60) */
    BLU_ST_ERR_BAD_LINEAGE = 17,     /* top-group row whose lineage fails parse_taxonomy (blast_result.rs:109-114) */
    BLU_ST_ERR_ROOT_DISAGREE = 18,   /* disagreement at level 0: `index - 1` underflow (find_multi_taxa_consensus.rs:181) */
    BLU_ST_ERR_SINGLE_BELOW_CUTOFFS = 19, /* single hit below every cutoff (find_single_query_consensus.rs:113-119) */
    BLU_ST_ERR_BAD_PIDENT = 20       /* NaN perc_identity in the top group: comparator/unwrap behaviour not restated */
};

#define BLU_FLAG_MUTATED 0x01u /* TaxonomyBean.mutated (build_blast_consensus_identity.rs:35-37) */
#define BLU_FLAG_AGREE 0x02u   /* every examined level agreed: taxonomy = whole cutoff-filtered reference lineage */

/* One 32-byte record per query.  Strings (taxonomy, consensus beans) are
 * rebuilt on the host from (ref_row, level_mask, bean_index). */
typedef struct blu_result {
    uint8_t status;            /* enum blu_status */
    uint8_t flags;             /* BLU_FLAG_* */
    uint8_t bean_index;        /* index into the reference lineage passed to build_blast_consensus_identity */
    uint8_t max_allowed_level; /* level of the reference lineage whose rank is max_allowed_rank; BLU_NONE_U8 = None */
    uint16_t reached_rank;     /* canonical rank code of the final element (blu_taxonomy_rank_name) */
    uint16_t max_allowed_rank; /* canonical rank code, BLU_MAR_NEVER_EQUAL, or BLU_NONE_U16 */
    uint32_t identifier_node;  /* interned node id of the final element: TaxonomyBean.identifier */
    uint32_t ref_row;          /* absolute hit row of the reference row R: perc_identity, bit_score, lineage */
    uint64_t level_mask;       /* bit j set = level j of R's lineage is in `taxonomy` */
    double ident_used;         /* identity tested against the cutoffs (R's pident, or the group max on disagreement) */
} blu_result;

/* -------------------------------------------------------------------------- */

uint32_t blu_abi_version(void);

/* Copies the last error message of the calling thread into buf (NUL-terminated,
 * truncated to len); returns the message length. */
size_t blu_last_error(char* buf, size_t len);

/* Builds the device-resident taxonomy: lineage rows, rank-sequence shapes and
 * the per-shape f64 cutoff tables (InterpolatedIdentity::interpolate_identities,
 * linnaean_ranks.rs:220-383; Taxon::get_taxon_cutoff, taxon.rs:104-185).
 * device >= 0: HIP device ordinal.  device == -1: host-only handle (cutoff and
 * shape queries work, blu_consensus_run refuses) — used by CPU-side tests.
 * Caller keeps ownership of desc arrays; they may be freed after the call. */
int blu_taxonomy_create(const blu_taxonomy_desc* desc, const blu_cutoff_config* cfg, int device,
                        blu_taxonomy** out);
void blu_taxonomy_destroy(blu_taxonomy* tax);

/* Introspection used by the host-side renderer and by tests. */
uint64_t blu_taxonomy_n_tax(const blu_taxonomy* tax);
uint32_t blu_taxonomy_n_shapes(const blu_taxonomy* tax);
uint32_t blu_taxonomy_n_rank_codes(const blu_taxonomy* tax);
uint32_t blu_taxonomy_max_depth(const blu_taxonomy* tax);
uint64_t blu_taxonomy_device_bytes(const blu_taxonomy* tax);
/* canonical rank code -> Display string (linnaean_ranks.rs:74-89); serde!=0 gives
 * the serde name ("species", raw string for Other; linnaean_ranks.rs:14-29). */
const char* blu_taxonomy_rank_name(const blu_taxonomy* tax, uint32_t rank_code, int serde);
/* cutoffs of one taxonomy row: writes up to cap entries, returns the lineage length
 * (0 for a bad lineage, -1 for an invalid row). is_default[j]=1: level j mapped to a
 * DefaultRank of the backbone.  rank_code[j]: canonical code of level j. */
int32_t blu_taxonomy_row_cutoffs(const blu_taxonomy* tax, uint64_t desc_row, uint32_t cap, double* cutoff,
                                 uint8_t* is_default, uint16_t* rank_code);
/* taxid -> ENGINE row id (BLU_UNMATCHED_TAXID when absent); needs desc.taxid at create.  This is the join of
 * the hit table with the taxonomy (mod.rs:72-76); its output is what blu_hits.tax_row holds.  A taxid the descriptor
 * lists more than once maps to its FIRST row (the reference's left join would multiply the hit rows: the whole-use-case
 * path of blu_pipeline.h does; blutils' own databases have unique taxids). */
int blu_taxonomy_lookup(const blu_taxonomy* tax, const int64_t* taxid, uint64_t n, uint32_t* out_row);
/* desc row index -> engine row id for every row of the table (out_map[n_tax]); inverse in out_inverse[n_tax]
 * (either may be NULL).  For callers that already hold desc row indices. */
int blu_taxonomy_row_map(const blu_taxonomy* tax, uint32_t* out_map, uint32_t* out_inverse);
/* (ABI v5, introspection) Number of leading lineage levels shared by ALL rows at sorted positions lo..hi (the low BLU_ROW_BITS
 * bits of engine row ids; lo <= hi < n_tax) — what the per-level scan of find_multi_taxa_consensus.rs:137-180 finds for a
 * top group spanning those positions, before the clamp to the shortest lineage.  *by_scan: from the adjacent-row prefix
 * lengths, one by one; *by_tables: what the engine's tables give (the wide-node chains for spans of 128 rows and more,
 * the range-minimum tables otherwise and where the chains do not reach; *via = 1 chains, 0 range minimum).  The two
 * must agree; any pointer may be NULL.  Host only, no device needed. */
int blu_taxonomy_shared_levels(const blu_taxonomy* tax, uint32_t lo, uint32_t hi, uint32_t* by_scan, uint32_t* by_tables, int32_t* via);
/* (ABI v5) Frees the device buffers the handle keeps from call to call for the host-pointer path of blu_consensus_run
 * (the staged columns and records of the largest table so far).  The handle stays valid; the next host-pointer call
 * allocates what it needs again.  No run on the handle may be in flight. */
int blu_taxonomy_trim(const blu_taxonomy* tax);

/* The hot path: one blu_result per query.  `out` has n_queries records, on the
 * device when hits->on_device, else on the host.  Asynchronous on
 * params->stream when on_device (no host sync inside); synchronous otherwise.
 * Consecutive runs on one handle must be ordered (same stream, or synchronised): the handle's scratch
 * (worklist and its counters) is reused from run to run. */
int blu_consensus_run(const blu_taxonomy* tax, const blu_hits* hits, const blu_run_params* params,
                      blu_result* out);

/* The side records of the packed layouts from the four columns (an ingest-time pass, like the join that produced
 * tax_row): `columns` holds tax_row (engine row ids), align_len, acc_rank and pident_milli or pident, n_hits and
 * on_device (1: device pointers on the handle's GPU, out too; the kernel runs on `stream` and the call returns after
 * it has finished — it reports values the layout cannot hold).  blu_hits_pack writes 4 words per hit and fails with
 * BLU_ERR_INVALID_ARG if a perc_identity is not an exact milli-percent value below BLU_PACKED_PIDENT_LIMIT;
 * blu_hits_pack64 writes 6 words per hit and takes any f64 (or milli-percent column, converted as the engine would). */
int blu_hits_pack(const blu_taxonomy* tax, const blu_hits* columns, uint32_t* packed_out, void* stream);
int blu_hits_pack64(const blu_taxonomy* tax, const blu_hits* columns, uint32_t* packed64_out, void* stream);

/* One host table over several GPUs: `taxes[0..n_tax_handles)` are handles of the SAME taxonomy and cutoff
 * configuration on different devices (the same device twice is allowed).  Queries are cut into contiguous ranges
 * balanced by hit count (blu_shard_ranges), every range runs on its handle from a host thread of its own (staging,
 * kernels and record copy-back overlap across devices), and the records land in `out` in query order with `ref_row`
 * pointing into the whole table.  Host pointers only (hits->on_device must be 0); no collective, nothing shared but
 * the read-only inputs.  Returns the first error of any shard. */
int blu_consensus_run_multi(const blu_taxonomy* const* taxes, uint32_t n_tax_handles, const blu_hits* hits,
                            const blu_run_params* params, blu_result* out);
/* bounds[0..n_shards]: query indices cutting seg_off[0..n_queries] into n_shards contiguous ranges whose hit counts
 * are as equal as the segment boundaries allow (host pointers). */
int blu_shard_ranges(const uint64_t* seg_off, uint64_t n_queries, uint32_t n_shards, uint64_t* bounds);

/* Name of the dominant kernel and its launch geometry for the last run on this
 * thread (for profiles/ bookkeeping). */
int blu_consensus_last_launch(char* kernel_name, size_t len, uint32_t* grid, uint32_t* block);

#ifdef __cplusplus
}
#endif
#endif /* BLU_CONSENSUS_H */
