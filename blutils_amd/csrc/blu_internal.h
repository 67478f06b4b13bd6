// Internal declarations shared by the host side and the HIP kernels of
// libblu_consensus.so.  Not part of the ABI (include/blu_consensus.h is).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

#include "blu_consensus.h"

namespace blu {

// canonical rank codes: 0..8 = LinnaeanRank enum kinds (linnaean_ranks.rs:16-29),
// >= 9 = Other(slug), interned in order of first appearance
enum RankKind : uint16_t { K_UNDEFINED = 0, K_DOMAIN, K_KINGDOM, K_PHYLUM, K_CLASS, K_ORDER, K_FAMILY, K_GENUS,
                           K_SPECIES, K_FIRST_OTHER };

struct RankInfo {
    std::string display;  // impl Display (linnaean_ranks.rs:74-89)
    std::string serde;    // serde camelCase name / raw Other string
};

// Device view of the taxonomy (passed to kernels by value).
//
// Lineage rows sit in LEXICOGRAPHIC order of their node sequences (row index = "pos"; engine row ids are
// pos | lineage length << BLU_ROW_BITS, so the streaming phase needs no taxonomy lookup).  One row:
//   word 0             len | shape << 8   (len 0 = lineage that fails parse_taxonomy)
//   bytes 4+j, 24+j    0x80 | a_j and 0x80 | b_j, j < 20: how many sorted rows to the left (a) / right (b) of this one still
//                      share its levels 0..j, saturated at 127 (levels the row does not have: 0x80).  The levels shared by a
//                      group spanning [lo, hi] around this row are the levels with a_j >= pos - lo and b_j >= hi - pos, four
//                      levels per 32-bit subtraction (bit 7 of a byte survives `byte - distance` exactly when the run is long
//                      enough); exact when both distances are <= 127 — wider groups, and agreement deeper than 20 levels, use
//                      the lcp8 / rmq tables below
//   word 11+j          node id of level j   interned (Display(rank), identifier)
// i.e. one 128-byte line for lineages of up to 20 levels (stride 32 words; deeper taxonomies get longer rows).  What the finalisation needs per LEVEL — cutoff, rank code, max-allowed-rank code — depends on the
// row's shape only: codes[shape][cstride], one word per level =
//   cutoff id (12 bits) | canonical rank code (10 bits) << 12 | max-allowed-rank code (10 bits) << 22
// (a few hundred KB, re-read by every query, so it stays in L2).  Cutoffs are stored by id into `cutvals`, the
// table of the few hundred DISTINCT f64 cutoff values of this (taxonomy, backbone) pair, held in LDS.
struct TaxDev {
    const uint32_t* lin;     // [n_tax][stride]
    const uint32_t* codes;   // [n_shapes][cstride]
    const uint32_t* kthr;    // [n_shapes][cstride] per level: smallest milli-percent identity k with fl(k / 1000) >= cutoff (17 bits,
                             // BLU_KTHR_NEVER if none below it) | (fl(that k / 1000) == cutoff) << 17: `>=` and `>` as integer compares
                             // | canonical rank code << BLU_LVL_RANK_SHIFT | (max-allowed-rank code is BLU_MAR_NEVER_EQUAL, else
                             // the rank code itself) << BLU_LVL_NEVER_SHIFT: the integer level tests and the record read nothing else
    uint32_t cstride;        // words per shape row, multiple of 16
    // lcp8[i] = number of leading levels shared by sorted rows i and i+1.  In that order the levels shared by a
    // whole group of rows = min(lcp8[lo .. hi-1]) for the group's smallest/largest pos: a range-minimum query
    // replaces the per-row level scan of find_multi_taxa_consensus.rs:137-180.
    const uint8_t* lcp8;     // [n_tax - 1], padded with 0xFF to a multiple of 16 (+16)
    const uint8_t* rmq;      // sparse table over 16-entry blocks of lcp8: level k at rmq + k * rmq_nb, entry j = min of blocks j .. j+2^k-1
    uint32_t rmq_nb;         // blocks per level
    const double* cutvals;   // [n_cutvals] distinct cutoff values (NaN included, compared by bit pattern)
    uint32_t n_cutvals;
    uint64_t n_tax;
    uint32_t stride;         // words per lineage row, multiple of 32 (128 bytes)
    uint32_t node_base;      // word of the row where the node ids start (BLU_ROW_NODE_BASE)
    uint32_t max_depth;      // longest lineage: bounds the length bits of a (possibly corrupt) row id
    uint32_t n_shapes;       // rows of codes / kthr (>= 1): bounds a (possibly corrupt) shape id
    // Shared levels of a WIDE group (more than BLU_ROW_RUN_MAX sorted rows either side of the reference row) without the
    // range-minimum tables.  A node of the taxonomy with at least BLU_WIDE_MIN rows is "wide"; every node that holds a wide
    // group is wide, and so is each of its ancestors, so the wide nodes form a small tree (15 k nodes at 2.4 M taxids).
    //   wblk[pos >> 6]    deepest wide node around each sorted position of a 64-row block: {w0 | w1 << 16, w2 | s1 << 16 | s2 << 24} —
    //                     w0 up to offset s1, w1 up to s2, w2 after (node id + 1, 0 = none); s1 == BLU_WBLK_OVERFLOW: more than two
    //                     changes inside the block, ask the range-minimum tables
    //   wchain[w][16]     end (exclusive sorted position) of the ancestor at level i of wide node w - 1, i = 0 .. its own level;
    //                     0 beyond.  Row 0 is all zeros (no wide node).
    // The group [lo, hi] shares exactly the levels i with hi < wchain[w(lo)][i]: the nodes that hold lo AND hi are wide, hold lo,
    // and are therefore that chain's prefix.  0.3 MB + 1 MB, hot in L2, instead of four lines of the 5 MB lcp8 / rmq tables.
    const uint2* wblk = nullptr;       // null: no wide tables (more than 65534 wide nodes, or a wide node deeper than 32 levels)
    const uint32_t* wchain = nullptr;  // [n_wide + 1][BLU_WCHAIN]
    const uint32_t* wchain_hi = nullptr;   // levels 16 .. 31 of the chains, when wide_levels > 16
    uint32_t wide_levels = 0;          // deepest wide node's level + 1
};
#define BLU_WIDE_MIN 128u          // rows of a node from which it is "wide" (a wide group spans at least BLU_ROW_RUN_MAX + 2 rows)
#define BLU_WCHAIN 16u             // chain entries per table row
#define BLU_WBLK_SHIFT 6u          // 64 sorted positions per wblk entry
#define BLU_WBLK_OVERFLOW 0xFFu

#define BLU_ROW_IV_LEVELS 20u    // levels whose neighbour run lengths sit in the row (words 1..10)
#define BLU_ROW_RUN_MAX 127u     // a run length byte saturates here
#define BLU_ROW_NODE_BASE 11u    // first node-id word of a row
#define BLU_KTHR_BITS 17u
#define BLU_KTHR_NEVER ((1u << BLU_KTHR_BITS) - 1u)
#define BLU_LVL_RANK_SHIFT 18u
#define BLU_LVL_NEVER_SHIFT 28u
#define BLU_HINT_BITS 15u        // shape hint of a packed side record: word 1 = pident_milli | (shape id + 1) << BLU_KTHR_BITS, 0 = none
#define BLU_PACK_CUT_BITS 12u
#define BLU_PACK_CODE_BITS 10u
#define BLU_PACK_CODE_MASK ((1u << BLU_PACK_CODE_BITS) - 1u)
#define BLU_PACK_NEVER (BLU_PACK_CODE_MASK - 1u)   // BLU_MAR_NEVER_EQUAL in 10 bits

struct HitsDev {
    const int32_t* bitscore;
    const uint32_t* tax_row;
    const double* pident;          // f64 layout, or nullptr
    const uint32_t* pident_milli;  // milli-percent layout, or nullptr
    const uint32_t* packed;        // packed layout: 4 words per hit {tax_row, pident_milli | shape hint << 17, align_len, acc_rank}, or nullptr
    const int32_t* align_len;
    const uint32_t* acc_rank;
    const uint64_t* seg_off;
    uint64_t n_hits;
    uint64_t n_queries;
    const uint32_t* packed64 = nullptr;   // 6 words per hit {tax_row, shape hint << 17, align_len, acc_rank, pident f64 lo, hi}, or nullptr
};

// launch wrapper implemented in consensus_kernel.hip
// worklist: n_queries uint32 slots; work_count: {queue length, blocks done}, zero between runs (the worklist kernel
// leaves them so)
// kind_dev: device address of a pinned host word the run's classification goes to (may be null); known_kind: 1 / 2 = launch
// the stream kernel with / without the bit-score ring alone, anything else = classify on the device and launch both
int launch_consensus(const TaxDev& tax, const HitsDev& hits, int strategy, blu_result* out, void* stream,
                     int device, int num_cus, uint32_t* worklist, uint32_t* work_count, uint32_t* kind_dev, uint32_t known_kind);
const char* consensus_kernel_name();
void consensus_last_geometry(uint32_t* grid, uint32_t* block);

void set_error(const char* fmt, ...);
// LinnaeanRank::from_str (linnaean_ranks.rs:52-72): enum kind 0..8, or K_FIRST_OTHER with the slug in *other
uint16_t parse_rank(const char* name, std::string* other);

}  // namespace blu

struct blu_taxonomy {
    int device = -1;
    int num_cus = 0;
    uint64_t n_tax = 0;
    uint32_t stride = 16;
    uint32_t sc = 16;
    uint32_t max_depth = 0;
    blu_cutoff_config cfg{};
    std::vector<blu::RankInfo> ranks;        // by canonical code
    std::vector<uint32_t> h_lin;             // host copy of the lineage rows
    std::vector<double> h_cut;               // [n_shapes * sc]
    std::vector<uint32_t> h_codes;           // [n_shapes * sc]
    std::vector<uint8_t> h_isdef;            // [n_shapes * sc]
    uint32_t n_shapes = 0;
    std::unordered_map<int64_t, uint32_t> taxid_row;
    unsigned char* d_block = nullptr;   // ONE device allocation; the table pointers below are views into it (a hipFree each cost 1.5-2 ms)
    uint32_t* d_lin = nullptr;
    std::vector<uint32_t> order;             // engine row id -> desc row
    uint8_t* d_lcp8 = nullptr;
    uint8_t* d_rmq = nullptr;
    uint32_t rmq_nb = 0;
    uint2* d_wblk = nullptr;                 // wide-group tables (TaxDev::wblk, wchain, wchain_hi); null = not built
    uint32_t* d_wchain = nullptr;
    uint32_t* d_wchain_hi = nullptr;
    uint32_t wide_levels = 0;
    uint32_t n_wide = 0;
    std::vector<uint8_t> h_lcp8;             // host copies of the span tables (blu_taxonomy_shared_levels: the tests' view of them)
    std::vector<uint8_t> h_rmq;
    std::vector<uint2> h_wblk;
    std::vector<uint32_t> h_wchain, h_wchain_hi;
    std::vector<uint32_t> pos_of;            // caller's tax_row -> sorted position
    std::vector<uint16_t> hint_of_pos;       // sorted position -> shape hint of the packed layout (shape id + 1, 0 = none)
    uint16_t* d_hint_of_pos = nullptr;
    double* d_cutvals = nullptr;
    uint32_t* d_codes = nullptr;
    uint32_t* d_kthr = nullptr;
    uint32_t n_cutvals = 0;
    uint32_t dev_stride = 32;                // words per DEVICE row
    uint32_t node_base = 1;                  // see TaxDev
    uint64_t device_bytes = 0;
    // per-handle scratch of the run call (worklist of long / overflowing queries); grown on demand,
    // so one handle must not be used by two concurrent blu_consensus_run calls
    mutable uint32_t* ws_worklist = nullptr;
    mutable uint32_t* ws_count = nullptr;
    mutable uint64_t ws_capacity = 0;
    mutable uint32_t* ws_pack_flag = nullptr;        // blu_hits_pack's "a value does not fit the record" word
    // which stream kernel the handle's last table wanted (consensus_kernel.hip: blu_classify_tasks): a pinned host word the
    // device writes, its device address, and the number of run calls so far
    mutable uint32_t* ws_kind_host = nullptr;
    mutable uint32_t* ws_kind_dev = nullptr;
    mutable uint64_t ws_calls = 0;
    // the table the remembered kind and queue length belong to: the pointers of its offsets, bit-scores and side values, its query
    // and row counts (a caching allocator hands the same offsets address to the next table of the same size: all five must agree)
    mutable const void* ws_kind_key_ptr[3] = {nullptr, nullptr, nullptr};
    mutable uint64_t ws_kind_key_n[2] = {0, 0};
    // Staging buffers of the host-pointer path of blu_consensus_run (two sets: a table staged in chunks is double-buffered),
    // kept with the handle and grown on demand instead of seven hipMalloc / hipFree pairs per call; [k][slot]: slot 0 bit-scores,
    // 1 side values (records, or the perc_identity column), 2 tax rows, 3 align_length, 4 accession ranks, 5 offsets, 6 records out
    mutable void* ws_stage[2][7] = {{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}};
    mutable size_t ws_stage_bytes[2][7] = {{0, 0, 0, 0, 0, 0, 0}, {0, 0, 0, 0, 0, 0, 0}};
};
