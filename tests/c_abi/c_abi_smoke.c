/* A caller of the C ABI that is not Python: plain C99, linked against libblu_consensus.so.
 *
 *   gcc -std=c99 -Wall -Iinclude tests/c_abi/c_abi_smoke.c -Lblutils_amd/lib -lblu_consensus -o c_abi_smoke
 *   LD_LIBRARY_PATH=blutils_amd/lib ./c_abi_smoke            (GPU: runs the C1-shaped table, prints the checksum)
 *   LD_LIBRARY_PATH=blutils_amd/lib ./c_abi_smoke --no-gpu   (host-only handle: ABI version, cutoffs, refusal to run)
 *
 * It stands where a Rust shim replacing the rayon map of
 * core/src/use_cases/build_consensus_identities/mod.rs:104-128 would stand (seam B): build the
 * taxonomy handle, join, hand the grouped SoA columns over as host pointers, read 32-byte records back.
 * The table is BASELINE config 1's shape (1 000 queries x 10 hits, 2 048-taxid taxonomy, assets 16S cutoffs) from a
 * 64-bit LCG that tests/test_c_abi.py restates in Python; that test feeds the same table to the oracle and compares
 * the FNV-1a checksum of the records printed here. */
#include <inttypes.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "blu_consensus.h"

#define N_TAX 2048u
#define DEPTH 8u
#define N_Q 1000u
#define HITS_PER_Q 10u

static uint64_t lcg_state = 0xB10751ull;
static uint64_t rnd(void) {
    lcg_state = lcg_state * 6364136223846793005ull + 1442695040888963407ull;
    return lcg_state >> 33;
}

static void die(const char* what, int rc) {
    char msg[512];
    blu_last_error(msg, sizeof msg);
    fprintf(stderr, "%s failed: rc %d: %s\n", what, rc, msg);
    exit(1);
}

int main(int argc, char** argv) {
    const int no_gpu = argc > 1 && strcmp(argv[1], "--no-gpu") == 0;
    static const char* rank_names[DEPTH] = {"d", "k", "p", "c", "o", "f", "g", "s"};
    static const uint32_t div_of[DEPTH] = {2048, 1024, 256, 64, 32, 8, 2, 1};
    if (blu_abi_version() != BLU_ABI_VERSION) { fprintf(stderr, "ABI version mismatch\n"); return 1; }

    /* taxonomy: taxid t has the lineage d;k;p;c;o;f;g;s with node j = 100000 j + t / div[j] */
    uint64_t* lin_off = malloc((N_TAX + 1) * sizeof *lin_off);
    uint32_t* lin_node = malloc(N_TAX * DEPTH * sizeof *lin_node);
    uint16_t* lin_rank = malloc(N_TAX * DEPTH * sizeof *lin_rank);
    for (uint32_t t = 0; t <= N_TAX; ++t) lin_off[t] = (uint64_t)t * DEPTH;
    for (uint32_t t = 0; t < N_TAX; ++t)
        for (uint32_t j = 0; j < DEPTH; ++j) {
            lin_node[t * DEPTH + j] = 100000u * j + t / div_of[j];
            lin_rank[t * DEPTH + j] = (uint16_t)j;
        }
    blu_taxonomy_desc desc;
    memset(&desc, 0, sizeof desc);
    desc.n_tax = N_TAX; desc.lin_off = lin_off; desc.lin_node = lin_node; desc.lin_rank = lin_rank;
    desc.n_ranks = DEPTH; desc.rank_names = rank_names;
    blu_cutoff_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.taxon = BLU_TAXON_CUSTOM; cfg.has_custom = 1;
    static const int16_t cuts[8] = {50, 60, 75, 80, 85, 92, 97, 99};   /* assets/custom-taxon-cutoffs-bacteria-16S.yaml */
    for (int i = 0; i < 8; ++i) { cfg.custom[i] = cuts[i]; cfg.custom_has[i] = 1; }

    blu_taxonomy* tax = NULL;
    int rc = blu_taxonomy_create(&desc, &cfg, no_gpu ? -1 : 0, &tax);
    if (rc != BLU_OK) die("blu_taxonomy_create", rc);
    if (blu_taxonomy_n_tax(tax) != N_TAX || blu_taxonomy_max_depth(tax) != DEPTH) { fprintf(stderr, "introspection mismatch\n"); return 1; }
    double cut[DEPTH];
    if (blu_taxonomy_row_cutoffs(tax, 5, DEPTH, cut, NULL, NULL) != (int32_t)DEPTH || cut[0] != 50.0 || cut[7] != 99.0) {
        fprintf(stderr, "cutoffs mismatch\n"); return 1;
    }
    uint32_t* row_map = malloc(N_TAX * sizeof *row_map);
    rc = blu_taxonomy_row_map(tax, row_map, NULL);
    if (rc != BLU_OK) die("blu_taxonomy_row_map", rc);

    /* hit table */
    const uint64_t n_hits = (uint64_t)N_Q * HITS_PER_Q;
    int32_t* bitscore = malloc(n_hits * sizeof *bitscore);
    uint32_t* tax_row = malloc(n_hits * sizeof *tax_row);
    double* pident = malloc(n_hits * sizeof *pident);
    int32_t* align_len = malloc(n_hits * sizeof *align_len);
    uint32_t* acc_rank = malloc(n_hits * sizeof *acc_rank);
    uint64_t* seg_off = malloc((N_Q + 1) * sizeof *seg_off);
    static const uint32_t masks[5] = {0, 1, 7, 31, 63};
    for (uint32_t q = 0; q < N_Q; ++q) {
        seg_off[q] = (uint64_t)q * HITS_PER_Q;
        const uint32_t anchor = (uint32_t)(rnd() % N_TAX), mask = masks[rnd() % 5], g = 1 + (uint32_t)(rnd() % 4);
        for (uint32_t j = 0; j < HITS_PER_Q; ++j) {
            const uint64_t r = seg_off[q] + j;
            const uint32_t subject = (anchor & ~mask) + (uint32_t)(rnd() % (mask + 1));
            bitscore[r] = j < g ? 500 : 500 - 1 - (int32_t)(rnd() % 16);
            pident[r] = (double)(80000 + rnd() % 20001) / 1000.0;
            align_len[r] = 380 + (int32_t)(rnd() % 101);
            acc_rank[r] = subject * 7u + 1u;
            tax_row[r] = row_map[subject];                      /* the join (mod.rs:72-76) */
        }
    }
    seg_off[N_Q] = n_hits;
    blu_hits hits;
    memset(&hits, 0, sizeof hits);
    hits.bitscore = bitscore; hits.tax_row = tax_row; hits.pident = pident; hits.align_len = align_len;
    hits.acc_rank = acc_rank; hits.seg_off = seg_off; hits.n_hits = n_hits; hits.n_queries = N_Q; hits.on_device = 0;
    blu_run_params params;
    memset(&params, 0, sizeof params);
    params.strategy = BLU_RELAXED;
    blu_result* out = calloc(N_Q, sizeof *out);

    rc = blu_consensus_run(tax, &hits, &params, out);
    if (no_gpu) {
        if (rc != BLU_ERR_NO_DEVICE) { fprintf(stderr, "a host-only handle must refuse to run (got rc %d)\n", rc); return 1; }
        printf("c_abi_smoke ok (no gpu): abi %u, host-only handle refused with BLU_ERR_NO_DEVICE\n", blu_abi_version());
        blu_taxonomy_destroy(tax);
        return 0;
    }
    if (rc != BLU_OK) die("blu_consensus_run", rc);
    uint64_t fnv = 0xcbf29ce484222325ull;
    const unsigned char* p = (const unsigned char*)out;
    for (size_t i = 0; i < (size_t)N_Q * sizeof *out; ++i) { fnv ^= p[i]; fnv *= 0x100000001b3ull; }
    unsigned n_multi = 0, n_single = 0, n_err = 0;
    for (uint32_t q = 0; q < N_Q; ++q) {
        if (out[q].status == BLU_ST_CONSENSUS_MULTI) ++n_multi;
        else if (out[q].status == BLU_ST_CONSENSUS_SINGLE) ++n_single;
        else ++n_err;
    }
    char kernel[128];
    uint32_t grid = 0, block = 0;
    blu_consensus_last_launch(kernel, sizeof kernel, &grid, &block);
    printf("c_abi_smoke ok: %u queries, multi %u single %u other %u, kernel %s, checksum %016" PRIx64 "\n", N_Q, n_multi, n_single,
           n_err, kernel, fnv);
    blu_taxonomy_destroy(tax);
    return 0;
}
