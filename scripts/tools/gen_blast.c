/* Synthetic BlastN outfmt-6 table (13 columns, the form `blutils build-consensus` reads) and the matching blutils
 * taxonomy JSON, written fast enough for the 2 M-query end-to-end run (100 M rows, 6.7 GB) to be set up in seconds.
 *   gen_blast table <out.tsv> <queries> <hits> <taxa> <seed> [clustered|uniform]
 *   gen_blast db    <out.json> <taxa>
 * Same row format as scripts/ingest_bench.py's generator. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint64_t s_state;
static inline uint64_t rnd(void) {
    uint64_t z = (s_state += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static inline char* put_u(char* p, uint64_t v, int width) {   /* zero-padded to `width` (0 = no padding) */
    char tmp[24];
    int n = 0;
    do { tmp[n++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (n < width) tmp[n++] = '0';
    while (n) *p++ = tmp[--n];
    return p;
}
static inline char* put_s(char* p, const char* s) { size_t n = strlen(s); memcpy(p, s, n); return p + n; }

int main(int argc, char** argv) {
    if (argc >= 4 && strcmp(argv[1], "db") == 0) {
        FILE* f = fopen(argv[2], "wb");
        if (!f) return 1;
        long taxa = atol(argv[3]);
        fputs("{\"blutilsVersion\":\"8.3.1\",\"sourceDatabase\":\"synthetic\",\"taxonomies\":[", f);
        for (long t = 0; t < taxa; ++t) {
            long g = t / 12, fam = t / 96;
            fprintf(f, "%s{\"taxid\": %ld, \"rank\": \"species\", \"numericLineage\": \"d__2;f__%ld;g__%ld;s__%ld\", "
                       "\"textLineage\": \"d__bacteria;f__fam%ld;g__gen%ld;s__sp%ld\", \"accessions\": []}",
                    t ? "," : "", 1000 + t, fam, g, 1000 + t, fam, g, t);
        }
        fputs("]}", f);
        return fclose(f) != 0;
    }
    if (argc < 7 || strcmp(argv[1], "table") != 0) { fprintf(stderr, "usage: see the header of gen_blast.c\n"); return 2; }
    FILE* f = fopen(argv[2], "wb");
    if (!f) return 1;
    const long queries = atol(argv[3]), hits = atol(argv[4]), taxa = atol(argv[5]);
    s_state = (uint64_t)atol(argv[6]) * 0x2545F4914F6CDD1Dull + 1;
    const int uniform = argc > 7 && strcmp(argv[7], "uniform") == 0;
    const size_t cap = 1u << 22;
    char* buf = (char*)malloc(cap + 256);
    char* p = buf;
    for (long q = 0; q < queries; ++q) {
        const uint64_t centre = rnd() % (uint64_t)taxa;
        for (long h = 0; h < hits; ++h) {
            const uint64_t sub = uniform ? rnd() % (uint64_t)taxa : (centre + rnd() % 96) % (uint64_t)taxa;
            const uint64_t pid = 80000 + rnd() % 20001, aln = 380 + rnd() % 100, bs = 200 + rnd() % 1800;
            *p++ = 'q'; p = put_u(p, (uint64_t)q, 8);
            p = put_s(p, "\tNR_"); p = put_u(p, sub, 6); p = put_s(p, ".1\t");
            p = put_u(p, 1000 + sub, 0); *p++ = '\t';
            p = put_u(p, pid / 1000, 0); *p++ = '.'; p = put_u(p, pid % 1000, 3); *p++ = '\t';
            p = put_u(p, aln, 0);
            p = put_s(p, "\t3\t1\t1\t400\t5\t404\t1e-120\t");
            p = put_u(p, bs, 0); *p++ = '\n';
            if ((size_t)(p - buf) >= cap) { fwrite(buf, 1, (size_t)(p - buf), f); p = buf; }
        }
    }
    fwrite(buf, 1, (size_t)(p - buf), f);
    free(buf);
    return fclose(f) != 0;
}
