// Types shared by the CPU ingest (pipeline.cpp) and the GPU ingest (ingest_gpu.hip).
#ifndef BLU_INGEST_H
#define BLU_INGEST_H

#include <cstddef>
#include <cstdint>
#include <memory>
#include <utility>
#include <string>
#include <thread>
#include <vector>

#include "blu_consensus.h"

namespace blu {

// std::vector whose resize() leaves trivially-constructible elements uninitialised: the columns are filled right
// after by parallel writers (scatter threads, device-to-host copies), and a serial zero-fill of GBs of fresh pages
// would cost more than the fill itself.
template <class T>
struct NoInitAlloc : std::allocator<T> {
    template <class U> struct rebind { using other = NoInitAlloc<U>; };
    NoInitAlloc() = default;
    template <class U> NoInitAlloc(const NoInitAlloc<U>&) {}
    template <class U, class... A>
    void construct(U* p, A&&... a) {
        if constexpr (sizeof...(A) == 0) ::new ((void*)p) U;
        else ::new ((void*)p) U(std::forward<A>(a)...);
    }
};
template <class T> using Column = std::vector<T, NoInitAlloc<T>>;

// taxid -> row of the taxonomy table; open addressing, 16-byte entries (one cache line touched per lookup: the join
// runs once per hit row).  A duplicated taxid is refused by the loaders (pipeline.cpp).
struct TaxidMap {
    struct E { int64_t key; uint32_t val, used; };
    std::vector<E> tab = std::vector<E>(1024, E{0, 0, 0});
    size_t n = 0;
    static uint64_t mixk(int64_t k) { uint64_t x = (uint64_t)k * 0x9E3779B97F4A7C15ull; return x ^ (x >> 32); }
    void rehash(size_t cap) {
        std::vector<E> old(cap, E{0, 0, 0});
        old.swap(tab);
        for (const E& e : old) if (e.used) put(e.key, e.val);
    }
    void put(int64_t k, uint32_t v) {
        const size_t m = tab.size() - 1;
        size_t i = mixk(k) & m;
        while (tab[i].used) { if (tab[i].key == k) return; i = (i + 1) & m; }
        tab[i] = E{k, v, 1};
    }
    void reserve(size_t want) { size_t cap = tab.size(); while (cap < want * 2) cap *= 2; if (cap != tab.size()) rehash(cap); }
    // false: the key was there already (the table is left as it was)
    bool emplace(int64_t k, uint32_t v) {
        if ((n + 1) * 2 > tab.size()) rehash(tab.size() * 2);
        const size_t m = tab.size() - 1;
        size_t i = mixk(k) & m;
        while (tab[i].used) { if (tab[i].key == k) return false; i = (i + 1) & m; }
        tab[i] = E{k, v, 1};
        ++n;
        return true;
    }
    uint32_t find_or(int64_t k, uint32_t missing) const {
        const size_t m = tab.size() - 1;
        size_t i = mixk(k) & m;
        while (tab[i].used) { if (tab[i].key == k) return tab[i].val; i = (i + 1) & m; }
        return missing;
    }
};

// Device copies of the grouped columns, left behind by the GPU ingest so that the engine reads them in place (no second
// trip over PCIe).  Owned by the HitTable; freed with it.
struct DeviceHits {
    int device = -1;
    uint64_t n_hits = 0, n_queries = 0;
    int32_t* bitscore = nullptr;
    int32_t* align_len = nullptr;
    uint32_t* tax_desc_row = nullptr;   // desc row indices (the engine's row ids are derived in device_run_consensus)
    uint32_t* acc_rank = nullptr;
    double* pident = nullptr;
    unsigned long long* seg_off = nullptr;   // [n_queries + 1]
    void* seg_block = nullptr;               // allocation seg_off lives in
    std::vector<void*> trash;                // device work buffers of device_run_consensus: freed with the columns (a hipFree is 1-2 ms)
    ~DeviceHits();
};

// What the ingest hands to the engine and the renderer (a2 + a4 + a5 of SURVEY 8a).
struct HitTable {
    std::vector<std::string> query_names;        // first-appearance order
    Column<uint64_t> seg_off;
    Column<int32_t> bitscore, align_len;
    Column<uint32_t> tax_desc_row, acc_rank;   // acc_rank: rank of the accession in byte order = index into `accessions`
    Column<double> pident;
    std::vector<std::string> accessions;         // sorted (String::cmp)
    uint64_t unmatched = 0;
    uint64_t n_hits = 0;                         // rows of the table (the host columns may be absent, see below)
    uint64_t n_queries = 0;                      // = query_names.size() once the strings are there (wait_strings())
    bool host_columns = true;                    // false: the GPU ingest was asked to leave the columns on the device only
    std::unique_ptr<DeviceHits> dev;             // set by the GPU ingest
    // The GPU ingest returns as soon as the device columns are complete; query_names and accessions are still being
    // sized and filled by this thread from the distinct strings the device sent back (n_queries is set).  wait_strings()
    // before either is touched.
    std::thread strings_thread;
    bool strings_ok = true;                      // false: the strings thread ran out of memory (read after wait_strings())
    void wait_strings() { if (strings_thread.joinable()) strings_thread.join(); }
    void clear() {
        wait_strings();
        query_names.clear(); accessions.clear();
        seg_off = {}; bitscore = {}; align_len = {}; tax_desc_row = {}; acc_rank = {}; pident = {};
        unmatched = n_hits = n_queries = 0; host_columns = true; strings_ok = true;
        dev.reset();
    }
    HitTable() = default;
    HitTable(const HitTable&) = delete;
    HitTable& operator=(const HitTable&) = delete;
    ~HitTable() { wait_strings(); }
};

// What the writer reads of the hit table: the rows of every rendered query's top-score group, in file order
// (find_single_query_consensus.rs:51-64 — the top group is all the reference ever parses).  On the GPU path it is
// compacted on the device and is the only part of the table that crosses PCIe on the way back.
struct TopRow { uint32_t row, desc_row, acc_rank; int32_t align_len; double pident; };   // 24 bytes
struct TopTable {
    Column<uint64_t> off;        // [n_queries + 1]; empty group: query not rendered (status >= 2)
    Column<TopRow> rows;
    Column<int32_t> score;       // [n_queries] the group's bit-score
};


#define BLU_INGEST_FALLBACK (-100)   // load_hits_gpu: the file is not in the plain form the GPU parser handles; use the CPU path

// outfmt-6 text -> grouped SoA columns on the GPU (ingest_gpu.hip).  Same result as the CPU ingest, bit for bit, for
// files in the plain BLAST form (no quotes, no empty lines, numbers of <= 15 significant digits); anything else returns
// BLU_INGEST_FALLBACK with the reason in *why and the caller parses on the CPU.
// host_columns = false: the grouped columns stay on the device only (ht.dev) — download_columns() fetches them later
// if they turn out to be needed.
// fd: the table, open for reading (the text is read with pread straight into pinned staging buffers).
int load_hits_gpu(int fd, size_t size, const TaxidMap& row_of, int device, bool host_columns, HitTable& ht, std::string* why);
int download_columns(HitTable& ht);
// Brings the HIP runtime and the device's null stream up (the first call into the runtime costs 0.06-0.26 s, the first
// queue another 0.04-0.17 s: whoever comes first pays); the use-case starts it on a thread of its own before it reads
// the taxonomy file, so that part of that time passes beside host work.  Errors are left to the calls that follow.
void warm_up_device(int device);

// The engine on the columns the GPU ingest left on the device: taxonomy rows -> engine row ids (fwd = blu_taxonomy_row_map's
// forward table), perc_identity as milli-percent when every value is exactly k/1000 (checked on the device), one
// blu_consensus_run with device pointers, records copied to `out` (host) and the top-score rows of the rendered
// queries compacted into `top`.  The device columns are left as they were.
int device_run_consensus(const blu_taxonomy* tax, DeviceHits& dev, const uint32_t* fwd, uint64_t n_tax, int strategy, blu_result* out,
                         TopTable* top);

}  // namespace blu
#endif
