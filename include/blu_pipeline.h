/*
 * blu_pipeline.h — host-side drop-in for blutils' `build_consensus_identities`
 * use-case and the result writer behind `blu blastn build-consensus`.
 *
 * Reference entry points mirrored (same arguments, same meaning):
 *   core/src/use_cases/build_consensus_identities/mod.rs:40-47
 *     build_consensus_identities(blast_output: ParallelBlastOutput{output_file, headers},
 *                                taxonomies_file, taxon, strategy, use_taxid, custom_taxon_values)
 *   core/src/use_cases/write_blutils_output.rs:33-38
 *     write_blutils_output(results, config, blutils_out_file, out_format)
 *
 * What runs where: text ingest (outfmt-6 TSV, blutils DB JSON), lineage
 * interning, the taxid join and the per-query grouping are host C++; the
 * per-query consensus itself is the HIP engine (blu_consensus_run); strings and
 * consensus beans are rebuilt on the host from the 32-byte records.
 */
#ifndef BLU_PIPELINE_H
#define BLU_PIPELINE_H

#include <stddef.h>
#include <stdint.h>

#include "blu_consensus.h"

#ifdef __cplusplus
extern "C" {
#endif

/* write_blutils_output.rs:20-31 OutputFormat */
enum blu_out_format { BLU_OUT_JSON = 0, BLU_OUT_JSONL = 1, BLU_OUT_YAML = 2,
                      BLU_OUT_JSON_COMPACT = 3 /* serde_json::to_writer: what the CLI prints to stdout (:152-163) */ };

typedef struct blu_pipeline_params {
    blu_cutoff_config cutoffs;   /* taxon + Option<CustomTaxon> */
    int32_t strategy;            /* enum blu_strategy */
    int32_t use_taxid;           /* Option<bool>: != 0 -> numericLineage, else textLineage (mod.rs:287-291) */
    int32_t device;              /* HIP device ordinal */
    int32_t out_format;          /* enum blu_out_format */
    int32_t lenient;             /* 0: a query that makes the reference panic fails the call (BLU_ERR_REFERENCE_PANIC),
                                    like the reference aborts; 1: such queries are written with "taxon": null */
    int32_t reserved;
} blu_pipeline_params;

#define BLU_ERR_REFERENCE_PANIC 9 /* a per-query condition on which the reference panics (see blu_status >= 16) */

typedef struct blu_pipeline_stats {
    uint64_t n_hits, n_queries, n_taxids, n_unmatched_rows;
    double t_load_db_s, t_load_hits_s, t_engine_s, t_render_s;
} blu_pipeline_stats;

/* Runs the whole use-case.  headers/n_headers: Option<Vec<String>> of FASTA ids (NULL/0 = None): ids without a
 * hit row become NoConsensusFound entries (mod.rs:86-102).  On success *out_text is a malloc'd buffer with the
 * serialized results, sorted by query (write_blutils_output.rs:111), in `out_format`:
 *   JSON : {"results":[QueryWithConsensus...],"config":null} pretty-printed like serde_json::to_string_pretty
 *          (runId is a fresh UUID v4 per call; config is None on this path, cmds/blast/mod.rs:137-142)
 *   JSON_COMPACT: the same document on one line
 *   JSONL: the config line (`null`) then one QueryWithConsensus per line
 *   YAML : block style as serde_yaml 0.9 emits BlutilsOutput (scalar quoting rules of the third-party emitter are
 *          approximated: parity unpinned there)
 * Free with blu_free_text. */
int blu_build_consensus_identities(const char* blast_output_file, const char* const* headers, uint64_t n_headers,
                                   const char* taxonomies_file, const blu_pipeline_params* params, char** out_text,
                                   size_t* out_len, blu_pipeline_stats* stats);
/* The same with Some(BlastBuilder) as `config` (run_blast_and_build_consensus/mod.rs:53-67 -> write_blutils_output):
 * run_id_text = the config's run id (36 characters; NULL/"" = a fresh UUID), which every result carries
 * (write_blutils_output.rs:82-104); config_text = the config already serialized for `out_format` at its place in the
 * document (JSON: the value after "config": — for the pretty form with its inner lines indented by two spaces;
 * JSONL: the first line; YAML: the block under `config:`), NULL/"" = null.  blutils_amd/blast.py produces it. */
int blu_build_consensus_identities_cfg(const char* blast_output_file, const char* const* headers, uint64_t n_headers,
                                       const char* taxonomies_file, const blu_pipeline_params* params,
                                       const char* run_id_text, const char* config_text, char** out_text, size_t* out_len,
                                       blu_pipeline_stats* stats);
/* The same, with the document written straight to `out_path` (no copy through the caller): what the CLI does with
 * --blutils-out-file. */
int blu_build_consensus_identities_to_file(const char* blast_output_file, const char* const* headers, uint64_t n_headers,
                                           const char* taxonomies_file, const blu_pipeline_params* params,
                                           const char* run_id_text, const char* config_text, const char* out_path,
                                           blu_pipeline_stats* stats);
void blu_free_text(char* text);

/* The text-ingest half alone (no GPU): DB JSON + outfmt-6 TSV -> SoA columns, as blu_build_consensus_identities does it.
 * Fills stats (rows, queries, taxids, unmatched rows, load times) and *checksum with an FNV-1a hash over every SoA
 * column, the segment offsets and the query names — identical for any BLU_INGEST_THREADS value.  For tests and for
 * timing the ingest (rows/s) apart from the engine. */
int blu_ingest_only(const char* blast_output_file, const char* taxonomies_file, int use_taxid, blu_pipeline_stats* stats,
                    uint64_t* checksum);
/* The same on HIP device `device` (>= 0): the table is parsed by the GPU ingest (csrc/ingest_gpu.hip) when it is in the
 * plain form BLAST writes, by the CPU otherwise; the columns — and the checksum — are identical either way.
 * BLU_INGEST=cpu|gpu in the environment forces one of the two (gpu also for files under 1 MiB). */
int blu_ingest_only_on(const char* blast_output_file, const char* taxonomies_file, int use_taxid, int device,
                       blu_pipeline_stats* stats, uint64_t* checksum);

/* The columns themselves (what blu_ingest_only[_on] hashes), for a caller that wants the SoA table and for tests that
 * compare the parsers column by column with an independent reading of the file.  Every array is malloc'd by the library
 * and released by blu_ingest_columns_free; query_names / accessions are the strings back to back, each NUL-terminated
 * (queries in first-appearance order, accessions in byte order = acc_rank order). */
typedef struct blu_ingest_columns {
    uint64_t n_hits, n_queries, n_accessions;
    uint64_t* seg_off;        /* [n_queries + 1] */
    int32_t* bitscore;        /* [n_hits] truncated toward zero (mod.rs:184) */
    int32_t* align_len;       /* [n_hits] */
    uint32_t* tax_desc_row;   /* [n_hits] row of the taxonomies file (left join, mod.rs:72-76) or BLU_UNMATCHED_TAXID */
    uint32_t* acc_rank;       /* [n_hits] */
    double* pident;           /* [n_hits] */
    char* query_names; uint64_t query_names_bytes;
    char* accessions; uint64_t accessions_bytes;
} blu_ingest_columns;
int blu_ingest_columns_on(const char* blast_output_file, const char* taxonomies_file, int use_taxid, int device,
                          blu_ingest_columns* out);
void blu_ingest_columns_free(blu_ingest_columns* cols);

/* Which parser the calling thread's last ingest used: 0 = CPU, 1 = GPU. */
int blu_last_ingest_path(void);

/* Binary cache of the taxonomies file.  The reference re-parses the `*.blutils.json` on every run
 * (mod.rs:246-327, taxonomies_map.rs:6-32) and keeps only {taxid, numericLineage | textLineage}; this writes exactly
 * that — interned lineages of the chosen flavour — as a flat file.  Wherever a `taxonomies_file` is taken
 * (blu_build_consensus_identities, blu_ingest_only) a cache file is recognised by its magic and mapped instead of
 * parsed; results are identical.  A cache built for the other lineage flavour is refused (BLU_ERR_INVALID_ARG), a
 * truncated or altered one fails its checksum (BLU_ERR_PARSE). */
int blu_db_cache_build(const char* taxonomies_file, int use_taxid, const char* cache_file);

/* CustomTaxon::from_file (domain/dtos/taxon.rs:28-66): .yaml or .json with the eight cutoff fields. */
int blu_custom_taxon_from_file(const char* path, blu_cutoff_config* cfg);

#ifdef __cplusplus
}
#endif
#endif /* BLU_PIPELINE_H */
