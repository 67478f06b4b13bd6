// Host-side drop-in for `build_consensus_identities` + the result writer
// (include/blu_pipeline.h).  Reference files followed:
//   core/src/use_cases/build_consensus_identities/mod.rs:40-129   orchestration, headers without hits
//   mod.rs:134-221   fold_results_by_query: per-query rows in file order, quotes stripped, bit_score f64 -> i64
//   mod.rs:226-244   outfmt-6 schema (13 tab-separated columns, no header)
//   mod.rs:246-327   blutils DB JSON -> {taxid, numericLineage | textLineage}
//   domain/dtos/blast_result.rs:38-120   lineage grammar `rank__identifier(;rank__identifier)*`
//   domain/dtos/consensus_result.rs:47-88, build_blast_consensus_identity.rs:43-63   consensus beans
//   use_cases/write_blutils_output.rs:87-111,126-232   flattening, sort by query, JSON / JSONL
// Third-party edges not pinned by any reference test: polars' CSV number parsing (strtod/strtoll
// here), serde_json's float printing (shortest round-trip digits here, ".0" appended to integral values).
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <cerrno>
#include <charconv>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <iterator>
#include <memory>
#include <string>
#include <string_view>
#include <thread>
#include <unordered_map>
#include <vector>

#include "blu_internal.h"
#include "blu_pipeline.h"
#include "ingest.h"

using namespace blu;

namespace {

double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// The memory of a finished use-case call (the hit table's strings, the records, the top rows, the taxonomy DB: 30 ms of
// free() and munmap() for a 2 M-query table) is given back by a thread of its own instead of on the caller's path.  At most
// one such thread exists: the next call that has something to release joins it first, and so does the unloading of the
// library.
struct Graveyard {
    std::mutex mu;
    std::thread t;
    void bury(std::thread&& next) { std::lock_guard<std::mutex> lk(mu); if (t.joinable()) t.join(); t = std::move(next); }
    void drain() { std::lock_guard<std::mutex> lk(mu); if (t.joinable()) t.join(); }
    ~Graveyard() { if (t.joinable()) t.join(); }
};
Graveyard g_graveyard;

// stage trace of the use-case (BLU_INGEST_TRACE=1): one stderr line per lap
struct Trace {
    const bool on = getenv("BLU_INGEST_TRACE") != nullptr;
    double tp = now_s();
    void lap(const char* what) { if (on) { const double t = now_s(); fprintf(stderr, "[pipeline] %-26s %.3f s\n", what, t - tp); tp = t; } }
};

struct MappedFile {
    const char* data = nullptr;
    size_t size = 0;
    int fd = -1;
    bool open(const char* path) {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) return false;
        struct stat st;
        if (fstat(fd, &st) != 0) return false;
        size = (size_t)st.st_size;
        if (size == 0) { data = ""; return true; }
        void* p = mmap(nullptr, size, PROT_READ, MAP_PRIVATE, fd, 0);
        if (p == MAP_FAILED) return false;
        data = (const char*)p;
        return true;
    }
    void close() {
        if (data && size) munmap((void*)data, size);
        if (fd >= 0) ::close(fd);
        data = nullptr; size = 0; fd = -1;
    }
    ~MappedFile() { close(); }
};

// ---------------------------------------------------------------------------------------------------------
// minimal JSON reader (enough for the blutils DB and the custom-cutoff file)
// ---------------------------------------------------------------------------------------------------------
struct Json {
    const char* p;
    const char* end;
    bool ok = true;
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
    bool eat(char c) { ws(); if (p < end && *p == c) { ++p; return true; } return false; }
    bool peek(char c) { ws(); return p < end && *p == c; }
    bool string(std::string* out) {
        ws();
        if (p >= end || *p != '"') { ok = false; return false; }
        ++p;
        if (out) out->clear();
        while (p < end && *p != '"') {
            if (*p == '\\' && p + 1 < end) {
                ++p;
                char c = *p++;
                char r = c;
                switch (c) {
                    case 'n': r = '\n'; break; case 't': r = '\t'; break; case 'r': r = '\r'; break;
                    case 'b': r = '\b'; break; case 'f': r = '\f'; break;
                    case 'u': {
                        unsigned v = 0;
                        for (int i = 0; i < 4 && p < end; ++i, ++p) v = v * 16 + (unsigned)(isdigit((unsigned char)*p) ? *p - '0' : (tolower(*p) - 'a' + 10));
                        if (out) {  // UTF-8 encode the BMP code point (surrogate pairs are not expected in lineages)
                            if (v < 0x80) out->push_back((char)v);
                            else if (v < 0x800) { out->push_back((char)(0xC0 | (v >> 6))); out->push_back((char)(0x80 | (v & 0x3F))); }
                            else { out->push_back((char)(0xE0 | (v >> 12))); out->push_back((char)(0x80 | ((v >> 6) & 0x3F))); out->push_back((char)(0x80 | (v & 0x3F))); }
                        }
                        continue;
                    }
                    default: break;
                }
                if (out) out->push_back(r);
            } else {
                if (out) out->push_back(*p);
                ++p;
            }
        }
        if (p >= end) { ok = false; return false; }
        ++p;
        return true;
    }
    bool number(double* out) {
        ws();
        // the token is copied first: the mapped file is not NUL-terminated, and a number may be its last bytes
        char tok[64];
        size_t n = 0;
        while (p + n < end && n < sizeof tok - 1 && (((unsigned)(p[n] - '0') < 10u) || p[n] == '-' || p[n] == '+' || p[n] == '.' || p[n] == 'e' || p[n] == 'E')) {
            tok[n] = p[n];
            ++n;
        }
        tok[n] = 0;
        char* e = nullptr;
        double v = strtod(tok, &e);
        if (e == tok) { ok = false; return false; }
        p += e - tok;
        if (out) *out = v;
        return true;
    }
    void skip() {
        ws();
        if (p >= end) { ok = false; return; }
        if (*p == '"') { string(nullptr); return; }
        if (*p == '{') {
            ++p;
            if (eat('}')) return;
            do { string(nullptr); if (!eat(':')) { ok = false; return; } skip(); } while (ok && eat(','));
            if (!eat('}')) ok = false;
            return;
        }
        if (*p == '[') {
            ++p;
            if (eat(']')) return;
            do { skip(); } while (ok && eat(','));
            if (!eat(']')) ok = false;
            return;
        }
        while (p < end && *p != ',' && *p != '}' && *p != ']' && *p != ' ' && *p != '\n' && *p != '\r' && *p != '\t') ++p;  // number / literal
    }
};

// ---------------------------------------------------------------------------------------------------------
// taxonomy side
// ---------------------------------------------------------------------------------------------------------
struct Db {
    std::vector<int64_t> taxid;
    std::vector<uint64_t> lin_off{0};
    std::vector<uint32_t> lin_node;
    std::vector<uint16_t> lin_rank;          // index into rank_raw
    std::vector<uint8_t> bad;
    std::vector<std::string> rank_raw;       // rank strings as they appear in lineages
    std::vector<std::string> rank_display;   // canonical Display of rank_raw[i]
    std::vector<std::string> node_ident;     // node id -> identifier string
    std::unordered_map<std::string, uint16_t> rank_ids;
    std::unordered_map<std::string, uint32_t> node_ids;   // key: Display(rank) + '\x1f' + identifier
    TaxidMap row_of;                         // taxid -> its (first) row
    // taxids listed more than once -> all their rows in file order.  The reference's polars left join (mod.rs:72-76) emits one
    // joined row per matching taxonomy row, so every hit of such a subject appears once per listing (load_hits does the same).
    std::unordered_map<int64_t, std::vector<uint32_t>> dup_rows;
    void note_row(int64_t taxid, uint32_t row) {
        if (row_of.emplace(taxid, row)) return;
        auto& v = dup_rows[taxid];
        if (v.empty()) v.push_back(row_of.find_or(taxid, 0));
        v.push_back(row);
    }
};

std::string canonical_display(const std::string& raw) {
    static const char* letters[9] = {"u", "d", "k", "p", "c", "o", "f", "g", "s"};
    std::string other;
    uint16_t k = parse_rank(raw.c_str(), &other);
    return k < K_FIRST_OTHER ? std::string(letters[k]) : other;
}

// blast_result.rs:38-120: split on ';' then on "__", every element must give exactly two parts
void add_lineage(Db& db, int64_t taxid, const std::string& lineage) {
    const size_t first = db.lin_node.size();
    bool bad = false;
    size_t pos = 0;
    for (;;) {
        size_t semi = lineage.find(';', pos);
        std::string_view el(lineage.data() + pos, (semi == std::string::npos ? lineage.size() : semi) - pos);
        size_t sep = el.find("__");
        if (sep == std::string_view::npos || el.find("__", sep + 2) != std::string_view::npos) bad = true;   // != 2 parts
        if (!bad) {
            std::string rank(el.substr(0, sep)), ident(el.substr(sep + 2));
            auto rit = db.rank_ids.find(rank);
            if (rit == db.rank_ids.end()) {
                rit = db.rank_ids.emplace(rank, (uint16_t)db.rank_raw.size()).first;
                db.rank_raw.push_back(rank);
                db.rank_display.push_back(canonical_display(rank));
            }
            std::string key = db.rank_display[rit->second];
            key.push_back('\x1f');
            key += ident;
            auto nit = db.node_ids.find(key);
            if (nit == db.node_ids.end()) {
                nit = db.node_ids.emplace(std::move(key), (uint32_t)db.node_ident.size()).first;
                db.node_ident.push_back(ident);
            }
            db.lin_node.push_back(nit->second);
            db.lin_rank.push_back(rit->second);
        }
        if (semi == std::string::npos) break;
        pos = semi + 1;
    }
    if (bad) { db.lin_node.resize(first); db.lin_rank.resize(first); }
    db.bad.push_back(bad ? 1 : 0);
    db.note_row(taxid, (uint32_t)db.taxid.size());
    db.taxid.push_back(taxid);
    db.lin_off.push_back(db.lin_node.size());
}

// ---- binary cache of the parsed taxonomies file ---------------------------------------------
// What load_db produces from the `*.blutils.json` (mod.rs:246-327: only {taxid, numericLineage | textLineage} of each
// entry is used), stored flat so a later run maps it instead of parsing ~100 MB of JSON per 300 k taxids:
//   header | taxid i64[n] | lin_off u64[n+1] | lin_node u32[m] | lin_rank u16[m] | bad u8[n] |
//   rank_off u32[r+1] | rank bytes | node_off u64[k+1] | node bytes      (sections padded to 8 bytes)
// The header records which lineage flavour was interned (use_taxid) and an FNV-1a hash of everything after it.
struct CacheHeader {
    char magic[8];            // "BLUDBC01"
    uint32_t version, use_taxid;
    uint64_t n_tax, n_lin, n_ranks, n_nodes, rank_bytes, node_bytes, payload_bytes, payload_hash;
};
const char kCacheMagic[8] = {'B', 'L', 'U', 'D', 'B', 'C', '0', '1'};

uint64_t fnv1a(const void* data, size_t n, uint64_t h = 1469598103934665603ull) {
    // 8 bytes per step (a hash of 64-bit words, tail bytewise): this guards against truncation / bit rot, not malice
    const unsigned char* p = (const unsigned char*)data;
    size_t i = 0;
    for (; i + 8 <= n; i += 8) { uint64_t w; memcpy(&w, p + i, 8); h = (h ^ w) * 1099511628211ull; }
    for (; i < n; ++i) h = (h ^ p[i]) * 1099511628211ull;
    return h;
}

size_t pad8(size_t n) { return (n + 7) & ~(size_t)7; }

bool is_cache(const MappedFile& f) { return f.size >= sizeof(CacheHeader) && memcmp(f.data, kCacheMagic, 8) == 0; }

int load_db_cache(const MappedFile& f, bool use_taxid, Db& db) {
    CacheHeader h;
    memcpy(&h, f.data, sizeof(h));
    if (h.version != 1) { set_error("taxonomy cache: unsupported version %u", h.version); return BLU_ERR_PARSE; }
    if ((h.use_taxid != 0) != use_taxid) {
        set_error("taxonomy cache was built for %s lineages but the run asks for %s", h.use_taxid ? "numeric (-u)" : "text",
                  use_taxid ? "numeric (-u)" : "text");
        return BLU_ERR_INVALID_ARG;
    }
    const size_t sizes[9] = {h.n_tax * 8, (h.n_tax + 1) * 8, h.n_lin * 4, h.n_lin * 2, h.n_tax, (h.n_ranks + 1) * 4, h.rank_bytes,
                             (h.n_nodes + 1) * 8, h.node_bytes};
    size_t total = 0;
    for (size_t x : sizes) total += pad8(x);
    if (h.n_tax >= (1ull << BLU_ROW_BITS) || total != h.payload_bytes || sizeof(h) + total != f.size) {
        set_error("taxonomy cache: size mismatch (truncated or not a cache file)");
        return BLU_ERR_PARSE;
    }
    const char* p = f.data + sizeof(h);
    if (fnv1a(p, total) != h.payload_hash) { set_error("taxonomy cache: checksum mismatch"); return BLU_ERR_PARSE; }
    auto take = [&](void* dst, size_t bytes) { memcpy(dst, p, bytes); p += pad8(bytes); };
    db.taxid.resize(h.n_tax); take(db.taxid.data(), sizes[0]);
    db.lin_off.resize(h.n_tax + 1); take(db.lin_off.data(), sizes[1]);
    db.lin_node.resize(h.n_lin); take(db.lin_node.data(), sizes[2]);
    db.lin_rank.resize(h.n_lin); take(db.lin_rank.data(), sizes[3]);
    db.bad.resize(h.n_tax); take(db.bad.data(), sizes[4]);
    std::vector<uint32_t> roff(h.n_ranks + 1); take(roff.data(), sizes[5]);
    const char* rbytes = p; p += pad8(sizes[6]);
    std::vector<uint64_t> noff(h.n_nodes + 1); take(noff.data(), sizes[7]);
    const char* nbytes = p;
    // offsets are data: check them before they index anything
    bool ok = db.lin_off[0] == 0 && db.lin_off[h.n_tax] == h.n_lin && roff[0] == 0 && roff[h.n_ranks] == h.rank_bytes &&
              noff[0] == 0 && noff[h.n_nodes] == h.node_bytes;
    for (uint64_t i = 0; ok && i < h.n_tax; ++i) ok = db.lin_off[i] <= db.lin_off[i + 1];
    for (uint64_t i = 0; ok && i < h.n_ranks; ++i) ok = roff[i] <= roff[i + 1];
    for (uint64_t i = 0; ok && i < h.n_nodes; ++i) ok = noff[i] <= noff[i + 1];
    for (uint64_t i = 0; ok && i < h.n_lin; ++i) ok = db.lin_node[i] < h.n_nodes && db.lin_rank[i] < h.n_ranks;
    if (!ok) { set_error("taxonomy cache: inconsistent offsets"); return BLU_ERR_PARSE; }
    db.rank_raw.resize(h.n_ranks); db.rank_display.resize(h.n_ranks);
    for (uint64_t i = 0; i < h.n_ranks; ++i) {
        db.rank_raw[i].assign(rbytes + roff[i], roff[i + 1] - roff[i]);
        db.rank_display[i] = canonical_display(db.rank_raw[i]);
    }
    db.node_ident.resize(h.n_nodes);
    for (uint64_t i = 0; i < h.n_nodes; ++i) db.node_ident[i].assign(nbytes + noff[i], noff[i + 1] - noff[i]);
    db.row_of.reserve(h.n_tax);
    for (uint64_t i = 0; i < h.n_tax; ++i) db.note_row(db.taxid[i], (uint32_t)i);
    return BLU_OK;
}

int write_db_cache(const Db& db, bool use_taxid, const char* path) {
    CacheHeader h{};
    memcpy(h.magic, kCacheMagic, 8);
    h.version = 1; h.use_taxid = use_taxid ? 1 : 0;
    h.n_tax = db.taxid.size(); h.n_lin = db.lin_node.size(); h.n_ranks = db.rank_raw.size(); h.n_nodes = db.node_ident.size();
    std::vector<uint32_t> roff{0};
    std::string rbytes, nbytes;
    for (auto& r : db.rank_raw) { rbytes += r; roff.push_back((uint32_t)rbytes.size()); }
    std::vector<uint64_t> noff{0};
    for (auto& n : db.node_ident) { nbytes += n; noff.push_back(nbytes.size()); }
    h.rank_bytes = rbytes.size(); h.node_bytes = nbytes.size();
    std::string out;
    auto put = [&](const void* src, size_t bytes) { out.append((const char*)src, bytes); out.append(pad8(bytes) - bytes, '\0'); };
    put(db.taxid.data(), db.taxid.size() * 8);
    put(db.lin_off.data(), db.lin_off.size() * 8);
    put(db.lin_node.data(), db.lin_node.size() * 4);
    put(db.lin_rank.data(), db.lin_rank.size() * 2);
    put(db.bad.data(), db.bad.size());
    put(roff.data(), roff.size() * 4);
    put(rbytes.data(), rbytes.size());
    put(noff.data(), noff.size() * 8);
    put(nbytes.data(), nbytes.size());
    h.payload_bytes = out.size();
    h.payload_hash = fnv1a(out.data(), out.size());
    std::string tmp = std::string(path) + ".tmp";
    FILE* fp = fopen(tmp.c_str(), "wb");
    if (!fp) { set_error("cannot write %s", tmp.c_str()); return BLU_ERR_IO; }
    bool ok = fwrite(&h, sizeof(h), 1, fp) == 1 && (out.empty() || fwrite(out.data(), out.size(), 1, fp) == 1);
    ok = (fclose(fp) == 0) && ok;
    if (!ok || rename(tmp.c_str(), path) != 0) { remove(tmp.c_str()); set_error("cannot write %s", path); return BLU_ERR_IO; }
    return BLU_OK;
}

// mod.rs:246-327 + domain/dtos/taxonomies_map.rs:6-32 (or the binary cache of the same content, see above)
int load_db(const char* path, bool use_taxid, Db& db) {
    MappedFile f;
    if (!f.open(path)) { set_error("Taxonomies file not found: %s", path); return BLU_ERR_IO; }
    if (is_cache(f)) return load_db_cache(f, use_taxid, db);
    Json j{f.data, f.data + f.size};
    if (!j.eat('{')) { set_error("taxonomies file is not a JSON object"); return BLU_ERR_PARSE; }
    bool found = false;
    std::string key, numeric, text;
    if (!j.peek('}')) {
        do {
            if (!j.string(&key) || !j.eat(':')) { j.ok = false; break; }
            if (key != "taxonomies") { j.skip(); continue; }
            found = true;
            if (!j.eat('[')) { j.ok = false; break; }
            if (j.eat(']')) continue;
            do {
                if (!j.eat('{')) { j.ok = false; break; }
                double taxid = 0;
                bool has_taxid = false, has_n = false, has_t = false;
                if (!j.peek('}')) {
                    do {
                        if (!j.string(&key) || !j.eat(':')) { j.ok = false; break; }
                        if (key == "taxid") has_taxid = j.number(&taxid);
                        else if (key == "numericLineage") has_n = j.string(&numeric);
                        else if (key == "textLineage") has_t = j.string(&text);
                        else j.skip();
                    } while (j.ok && j.eat(','));
                }
                if (!j.eat('}')) j.ok = false;
                if (!j.ok) break;
                if (!has_taxid || !has_n || !has_t) { set_error("taxonomy entry %zu lacks taxid/numericLineage/textLineage", db.taxid.size()); return BLU_ERR_PARSE; }
                // mod.rs:278 (u64 -> f64 -> i64), :287-291; a value no i64 holds cannot equal any subject_taxid of the Int64 column
                const int64_t taxid_i = taxid >= 9223372036854775808.0 ? INT64_MAX : (taxid <= -9223372036854775808.0 ? INT64_MIN : (int64_t)taxid);
                add_lineage(db, taxid_i, use_taxid ? numeric : text);
            } while (j.ok && j.eat(','));
            if (!j.eat(']')) j.ok = false;
        } while (j.ok && j.eat(','));
    }
    if (!j.ok || !found) { set_error("Unexpected error detected on parse `taxonomies` as json (offset %zu)", (size_t)(j.p - f.data)); return BLU_ERR_PARSE; }
    return BLU_OK;
}

// ---------------------------------------------------------------------------------------------------------
// hit table side
// ---------------------------------------------------------------------------------------------------------
std::string strip_quotes(std::string_view v) {   // mod.rs:169-172 `.replace("\"", "")`
    std::string s;
    s.reserve(v.size());
    for (char c : v) if (c != '"') s.push_back(c);
    return s;
}

// Parses one number the way the columns are typed (mod.rs:226-244).  Fast path for what BLAST prints — [-]digits[.digits]
// with at most 15 significant digits: mantissa and 10^k are exact doubles and IEEE division rounds correctly (Clinger),
// so the value is the one strtod / Rust's parser gives.  Everything else (exponents, '+', inf/nan, longer digit
// strings) takes std::from_chars, then strtod for the forms from_chars refuses.  (libstdc++ 11's from_chars for
// doubles goes through strtod with a locale switch: ~4 of them per row were most of the ingest time.)
bool parse_f64(std::string_view v, double* out) {
    const char* b = v.data();
    const char* e = b + v.size();
    if (b == e) return false;
    {
        static const double p10[16] = {1e0, 1e1, 1e2, 1e3, 1e4, 1e5, 1e6, 1e7, 1e8, 1e9, 1e10, 1e11, 1e12, 1e13, 1e14, 1e15};
        const char* p = b;
        const bool neg = p < e && *p == '-';
        if (neg) ++p;
        uint64_t mant = 0;
        int digits = 0, frac = 0;
        const char* d0 = p;
        while (p < e && (unsigned)(*p - '0') < 10u) { mant = mant * 10 + (unsigned)(*p - '0'); ++p; ++digits; }
        const bool had_int = p > d0;
        if (p < e && *p == '.') {
            ++p;
            const char* f0 = p;
            while (p < e && (unsigned)(*p - '0') < 10u) { mant = mant * 10 + (unsigned)(*p - '0'); ++p; ++digits; ++frac; }
            if (!had_int && p == f0) digits = 99;      // "." alone: not a number
        } else if (!had_int) digits = 99;
        if (p == e && digits <= 15) {
            const double x = (double)mant / p10[frac];
            *out = neg ? -x : x;
            return true;
        }
    }
    auto r = std::from_chars(b, e, *out);
    if (r.ec == std::errc() && r.ptr == e) return true;
    if ((unsigned char)*b <= ' ') return false;         // (strtod would skip leading blanks; a typed CSV column does not)
    for (const char* q = b; q < e; ++q) if (*q == 'x' || *q == 'X') return false;   // (nor does it read C's hexadecimal floats)
    std::string tmp(v);
    char* endp = nullptr;
    *out = strtod(tmp.c_str(), &endp);
    return endp == tmp.c_str() + tmp.size();            // the whole field, or it is not a number ("12abc")
}

// subject_taxid and align_length are Int64 columns of the reference's schema (mod.rs:226-244): [+-]digits and nothing
// else — no fraction, exponent or blanks ("12.7" fails there, so it fails here)
bool parse_i64(std::string_view v, int64_t* out) {
    const char* p = v.data();
    const char* e = p + v.size();
    if (p == e) return false;
    const bool neg = *p == '-';
    if (*p == '-' || *p == '+') ++p;
    if (p == e) return false;
    uint64_t x = 0;                                        // the whole i64 range, leading zeros included; beyond it: not an Int64
    const uint64_t lim = neg ? (1ull << 63) : (1ull << 63) - 1;
    for (; p < e; ++p) {
        const unsigned d = (unsigned)(*p - '0');
        if (d >= 10u) return false;
        if (x > (lim - d) / 10) return false;
        x = x * 10 + d;
    }
    *out = neg ? (int64_t)(0 - x) : (int64_t)x;
    return true;
}

struct RawRow { uint32_t lq, la, tax; int32_t bs, aln; double pid; };   // lq / la: the chunk's own query / accession ids

inline uint64_t hash_sv(std::string_view s) {
    uint64_t h = 0x9E3779B97F4A7C15ull ^ (s.size() * 0xFF51AFD7ED558CCDull);
    const char* p = s.data();
    size_t n = s.size();
    while (n >= 8) { uint64_t w; memcpy(&w, p, 8); h = (h ^ w) * 0x9FB21C651E98DF25ull; h ^= h >> 29; p += 8; n -= 8; }
    if (n) { uint64_t w = 0; memcpy(&w, p, n); h = (h ^ w) * 0x9FB21C651E98DF25ull; h ^= h >> 29; }
    return h ^ (h >> 32);
}

// string_view -> dense id in first-appearance order; open addressing.  A slot holds the hash, the id and the first 16
// bytes of the key, so a lookup of a short key (accessions, query ids) touches one cache line and never the text it
// was first seen in (random reads into a mapped file of many GB: that, not the parsing, was the ingest's cost).  Keys
// are views into the mapped file or into an arena that outlives the dictionary.
struct SvDict {
    struct Slot { uint64_t hash; uint32_t id1, len; char head[16]; };   // id1 = id + 1, 0 = empty
    std::vector<Slot> slot;
    std::vector<std::string_view> keys;      // by id
    uint64_t mask;
    explicit SvDict(size_t cap = 1024) : slot(cap, Slot{0, 0, 0, {0}}), mask(cap - 1) {}
    void grow() {
        std::vector<Slot> ns(slot.size() * 2, Slot{0, 0, 0, {0}});
        const uint64_t nm = ns.size() - 1;
        for (const Slot& e : slot) {
            if (!e.id1) continue;
            uint64_t i = e.hash & nm;
            while (ns[i].id1) i = (i + 1) & nm;
            ns[i] = e;
        }
        slot.swap(ns);
        mask = nm;
    }
    uint32_t intern(std::string_view s, bool* is_new = nullptr) {
        const uint64_t h = hash_sv(s);
        const size_t hl = s.size() < 16 ? s.size() : 16;
        uint64_t i = h & mask;
        while (slot[i].id1) {
            const Slot& e = slot[i];
            if (e.hash == h && e.len == s.size() && memcmp(e.head, s.data(), hl) == 0 &&
                (s.size() <= 16 || memcmp(keys[e.id1 - 1].data() + 16, s.data() + 16, s.size() - 16) == 0)) {
                if (is_new) *is_new = false;
                return e.id1 - 1;
            }
            i = (i + 1) & mask;
        }
        const uint32_t id = (uint32_t)keys.size();
        Slot& e = slot[i];
        e.hash = h; e.id1 = id + 1; e.len = (uint32_t)s.size();
        memcpy(e.head, s.data(), hl);
        keys.push_back(s);
        if (is_new) *is_new = true;
        if (keys.size() * 3 > slot.size() * 2) grow();
        return id;
    }
};

// One worker's share of the file: [begin, end) starts and ends on line boundaries.
struct Chunk {
    const char* begin = nullptr;
    const char* end = nullptr;
    std::vector<RawRow> rows;
    SvDict queries, accs;                    // this chunk's dictionaries (quotes already stripped)
    std::deque<std::string> arena;           // owned copies of the fields that had quotes in them
    std::vector<uint32_t> q_count;           // rows per local query
    std::vector<uint32_t> qmap, amap;        // local id -> global query id / global accession rank
    std::vector<uint64_t> at;                // local query -> next row slot in the grouped table
    std::vector<std::string_view> acc_sorted;
    uint64_t unmatched = 0, first_line = 0, n_lines = 0;
    int rc = BLU_OK;
    std::string err;
    std::string_view clean(std::string_view v) {   // mod.rs:169-172 `.replace("\"", "")`
        if (!memchr(v.data(), '"', v.size())) return v;
        arena.push_back(strip_quotes(v));
        return arena.back();
    }
};

void parse_chunk(Chunk& c, const Db& db, const char* path) {
    const char* p = c.begin;
    uint64_t line_no = 0;
    c.rows.reserve((size_t)(c.end - c.begin) / 96 + 16);
    std::string_view last_q_raw, last_a_raw;
    uint32_t last_q = 0, last_a = 0;
    bool have_last = false;
    while (p < c.end) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(c.end - p));
        const char* le = nl ? nl : c.end;
        const char* lend = le;
        if (lend > p && lend[-1] == '\r') --lend;
        ++line_no;
        if (lend > p) {
            std::string_view col[13];
            int nc = 0;
            const char* s = p;
            while (nc < 13) {
                const char* tab = (const char*)memchr(s, '\t', (size_t)(lend - s));
                const char* ce = tab ? tab : lend;
                col[nc++] = std::string_view(s, (size_t)(ce - s));
                if (!tab) break;
                s = tab + 1;
            }
            char msg[512];
            if (nc < 13) {
                snprintf(msg, sizeof msg, "line %llu(+%llu) of %s has %d columns, outfmt-6 with 13 expected (mod.rs:226-244)",
                         (unsigned long long)line_no, (unsigned long long)c.first_line, path, nc);
                c.rc = BLU_ERR_PARSE; c.err = msg; return;
            }
            double pid, bs;
            int64_t taxid_i, aln;
            if (!parse_i64(col[2], &taxid_i) || !parse_f64(col[3], &pid) || !parse_i64(col[4], &aln) || !parse_f64(col[12], &bs)) {
                snprintf(msg, sizeof msg, "line %llu(+%llu) of %s: numeric column does not parse", (unsigned long long)line_no,
                         (unsigned long long)c.first_line, path);
                c.rc = BLU_ERR_PARSE; c.err = msg; return;
            }
            const double bs_t = std::trunc(bs);               // mod.rs:184 AnyValue::Float64 -> try_extract::<i64>
            if (!(bs_t >= -2147483648.0 && bs_t <= 2147483647.0) || aln < INT32_MIN || aln > INT32_MAX) {
                snprintf(msg, sizeof msg, "line %llu(+%llu) of %s: bit_score / align_length outside the 32-bit range of the engine columns",
                         (unsigned long long)line_no, (unsigned long long)c.first_line, path);
                c.rc = BLU_ERR_PARSE; c.err = msg; return;
            }
            RawRow r;
            // rows of one query usually sit next to each other: the previous row's ids are tried before the dictionaries
            if (!have_last || col[0] != last_q_raw) {
                bool is_new = false;
                last_q = c.queries.intern(c.clean(col[0]), &is_new);
                if (is_new) c.q_count.push_back(0);
                last_q_raw = col[0];
            }
            if (!have_last || col[1] != last_a_raw) { last_a = c.accs.intern(c.clean(col[1])); last_a_raw = col[1]; }
            have_last = true;
            ++c.q_count[last_q];
            r.lq = last_q; r.la = last_a;
            r.tax = db.row_of.find_or(taxid_i, BLU_UNMATCHED_TAXID);    // left join (mod.rs:72-76)
            if (r.tax == BLU_UNMATCHED_TAXID) ++c.unmatched;
            r.bs = (int32_t)bs_t; r.aln = (int32_t)aln; r.pid = pid;
            c.rows.push_back(r);
            if (!db.dup_rows.empty()) {                                  // a subject listed m times: m joined rows, in the DB's order
                auto dit = db.dup_rows.find(taxid_i);
                if (dit != db.dup_rows.end())
                    for (size_t k = 1; k < dit->second.size(); ++k) { r.tax = dit->second[k]; c.rows.push_back(r); ++c.q_count[last_q]; }
            }
        }
        p = nl ? nl + 1 : c.end;
    }
    c.n_lines = line_no;
    // this chunk's accessions in byte order (String::cmp), for the merge below
    c.acc_sorted = c.accs.keys;
    std::sort(c.acc_sorted.begin(), c.acc_sorted.end());
}

template <class F>
void parallel_for(unsigned n, unsigned nthreads, F&& f) {
    if (nthreads <= 1 || n <= 1) { for (unsigned i = 0; i < n; ++i) f(i); return; }
    std::vector<std::thread> pool;
    for (unsigned i = 0; i < n; ++i) pool.emplace_back([&f, i]() { f(i); });
    for (auto& th : pool) th.join();
}

// a2 + a4 + a5: outfmt-6 text -> SoA columns.  The file is cut into line-aligned chunks; each worker parses its
// chunk (numbers, tab scanning, the taxid join) and interns query / accession strings in dictionaries of its own.
// Afterwards only the DISTINCT strings are merged — queries in file order (first appearance decides a query's
// position, mod.rs:192-208), accessions by a tree of sorted-list merges (their id is their rank in byte order) —
// and the workers scatter their rows into the grouped table.  The result does not depend on the thread count.
thread_local double g_t_body_end = 0;   // stage trace: when build_document's last statement ran (what follows is its tear-down)
thread_local int g_last_ingest_path = 0;   // 0 = CPU parser, 1 = GPU parser (blu_last_ingest_path)

int load_hits(const char* path, const Db& db, HitTable& ht, int device = -1, bool host_columns = true) {
    g_last_ingest_path = 0;
    // GPU parser first (ingest_gpu.hip) when a device is given: same columns bit for bit; files it does not handle
    // (quotes, empty lines, unusual numbers), small files and BLU_INGEST=cpu take the CPU path below.  The GPU path
    // reads the file through its descriptor and never maps it.
    {
        const char* mode = getenv("BLU_INGEST");
        struct stat sb;
        const int fd = ::open(path, O_RDONLY);
        if (fd < 0 || fstat(fd, &sb) != 0) { if (fd >= 0) ::close(fd); set_error("Unexpected error occurred on load table: %s", path); return BLU_ERR_IO; }
        const size_t fsize = (size_t)sb.st_size;
        // (a DB that lists a taxid more than once multiplies hit rows in the join: the CPU parser does that)
        const bool want_gpu = device >= 0 && db.dup_rows.empty() && !(mode && strcmp(mode, "cpu") == 0) && (fsize >= (1u << 20) || (mode && strcmp(mode, "gpu") == 0));
        int rc = BLU_INGEST_FALLBACK;
        std::string why;
        if (want_gpu) rc = load_hits_gpu(fd, fsize, db.row_of, device, host_columns, ht, &why);
        ::close(fd);
        if (want_gpu) {
            if (rc == BLU_OK) { g_last_ingest_path = 1; return BLU_OK; }
            if (rc != BLU_INGEST_FALLBACK) return rc;
            if (getenv("BLU_INGEST_TRACE")) fprintf(stderr, "[ingest] GPU parser declined (%s): CPU path\n", why.c_str());
        }
    }
    MappedFile f;
    if (!f.open(path)) { set_error("Unexpected error occurred on load table: %s", path); return BLU_ERR_IO; }
    unsigned nthreads = std::thread::hardware_concurrency();
    if (const char* env = getenv("BLU_INGEST_THREADS")) nthreads = (unsigned)atoi(env);
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 32) nthreads = 32;
    if (f.size < (1u << 20)) nthreads = 1;
    std::vector<Chunk> chunks(nthreads);
    const char* base = f.data;
    const char* fend = f.data + f.size;
    const char* cur = base;
    for (unsigned t = 0; t < nthreads; ++t) {
        const char* target = t + 1 == nthreads ? fend : base + (f.size / nthreads) * (t + 1);
        if (target < cur) target = cur;
        if (target < fend) {
            const char* nl = (const char*)memchr(target, '\n', (size_t)(fend - target));
            target = nl ? nl + 1 : fend;
        }
        chunks[t].begin = cur;
        chunks[t].end = target;
        cur = target;
    }
    const bool trace = getenv("BLU_INGEST_TRACE") != nullptr;
    double tp = now_s();
    auto lap = [&](const char* what) { if (trace) { const double t = now_s(); fprintf(stderr, "[ingest] %-22s %.3f s\n", what, t - tp); tp = t; } };
    parallel_for(nthreads, nthreads, [&](unsigned t) { parse_chunk(chunks[t], db, path); });
    lap("parse (parallel)");
    uint64_t lines_before = 0;
    size_t nh = 0;
    for (auto& c : chunks) {
        if (c.rc != BLU_OK) { set_error("%s (chunk starting at line %llu)", c.err.c_str(), (unsigned long long)(lines_before + 1)); return c.rc; }
        lines_before += c.n_lines;
        nh += c.rows.size();
        ht.unmatched += c.unmatched;
    }
    if (nh >= 0xFFFFFFFFull) { set_error("more than 2^32 - 2 hit rows in %s", path); return BLU_ERR_INVALID_ARG; }
    // queries: the chunks' distinct names in file order -> global ids, and each chunk's first slot inside every segment
    {
        SvDict global(1u << 16);
        std::vector<uint64_t> per_query;
        std::vector<std::vector<uint64_t>> prior(nthreads);
        for (unsigned t = 0; t < nthreads; ++t) {
            Chunk& c = chunks[t];
            c.qmap.resize(c.queries.keys.size());
            prior[t].resize(c.queries.keys.size());
            for (uint32_t l = 0; l < c.queries.keys.size(); ++l) {
                bool is_new = false;
                const uint32_t g = global.intern(c.queries.keys[l], &is_new);
                if (is_new) { ht.query_names.emplace_back(c.queries.keys[l]); per_query.push_back(0); }
                c.qmap[l] = g;
                prior[t][l] = per_query[g];
                per_query[g] += c.q_count[l];
            }
        }
        const size_t nq = ht.query_names.size();
        ht.seg_off.assign(nq + 1, 0);
        for (size_t q = 0; q < nq; ++q) ht.seg_off[q + 1] = ht.seg_off[q] + per_query[q];
        for (unsigned t = 0; t < nthreads; ++t) {
            Chunk& c = chunks[t];
            c.at.resize(c.qmap.size());
            for (uint32_t l = 0; l < c.qmap.size(); ++l) c.at[l] = ht.seg_off[c.qmap[l]] + prior[t][l];
        }
    }
    lap("queries (merge)");
    // accessions: tree of merges of the chunks' sorted lists; rank in the merged list = acc_rank = index into ht.accessions
    {
        std::vector<std::vector<std::string_view>> lists(nthreads);
        for (unsigned t = 0; t < nthreads; ++t) lists[t] = chunks[t].acc_sorted;
        for (unsigned step = 1; step < nthreads; step *= 2) {
            std::vector<unsigned> heads;
            for (unsigned i = 0; i + step < nthreads; i += 2 * step) heads.push_back(i);
            parallel_for((unsigned)heads.size(), nthreads, [&](unsigned k) {
                const unsigned i = heads[k];
                std::vector<std::string_view> out;
                out.reserve(lists[i].size() + lists[i + step].size());
                std::set_union(lists[i].begin(), lists[i].end(), lists[i + step].begin(), lists[i + step].end(), std::back_inserter(out));
                lists[i].swap(out);
                std::vector<std::string_view>().swap(lists[i + step]);
            });
        }
        const std::vector<std::string_view>& all = lists[0];
        ht.accessions.resize(all.size());
        parallel_for(nthreads, nthreads, [&](unsigned t) {
            for (size_t k = all.size() * t / nthreads; k < all.size() * (t + 1) / nthreads; ++k) ht.accessions[k].assign(all[k]);
            Chunk& c = chunks[t];
            c.amap.resize(c.accs.keys.size());
            for (uint32_t l = 0; l < c.accs.keys.size(); ++l)
                c.amap[l] = (uint32_t)(std::lower_bound(all.begin(), all.end(), c.accs.keys[l]) - all.begin());
        });
    }
    lap("accessions (merge)");
    // stable grouping: queries in first-appearance order, rows of a query in file order (mod.rs:192-208); the chunks
    // write disjoint slots
    ht.bitscore.resize(nh); ht.align_len.resize(nh); ht.tax_desc_row.resize(nh); ht.acc_rank.resize(nh); ht.pident.resize(nh);
    parallel_for(nthreads, nthreads, [&](unsigned t) {
        Chunk& c = chunks[t];
        for (const RawRow& r : c.rows) {
            const uint64_t d = c.at[r.lq]++;
            ht.bitscore[d] = r.bs; ht.align_len[d] = r.aln; ht.tax_desc_row[d] = r.tax; ht.acc_rank[d] = c.amap[r.la];
            ht.pident[d] = r.pid;
        }
    });
    lap("scatter (parallel)");
    ht.n_hits = nh;
    ht.n_queries = ht.query_names.size();
    return BLU_OK;
}

// ---------------------------------------------------------------------------------------------------------
// rendering
// ---------------------------------------------------------------------------------------------------------
// Output buffer of the JSON writer: the std::string calls the writers use (append / push_back / +=), inlined — a
// record is ~70 short appends, and the out-of-line call per append was a third of the rendering time.
struct Out {
    char* base = nullptr;
    size_t len = 0, cap = 0;
    Out() = default;
    Out(const Out&) = delete;
    Out& operator=(const Out&) = delete;
    Out(Out&& o) noexcept : base(o.base), len(o.len), cap(o.cap) { o.base = nullptr; o.len = o.cap = 0; }
    Out& operator=(Out&& o) noexcept { if (this != &o) { free(base); base = o.base; len = o.len; cap = o.cap; o.base = nullptr; o.len = o.cap = 0; } return *this; }
    ~Out() { free(base); }
    void reserve(size_t n) {
        if (n <= cap) return;
        char* nb = (char*)realloc(base, n);
        if (!nb) throw std::bad_alloc();
        base = nb; cap = n;
    }
    __attribute__((noinline)) void grow(size_t n) { reserve(std::max(cap * 2, len + n + 4096)); }
    __attribute__((always_inline)) inline void need(size_t n) { if (__builtin_expect(len + n > cap, 0)) grow(n); }
    __attribute__((always_inline)) inline void append(const char* p, size_t n) { need(n); memcpy(base + len, p, n); len += n; }
    __attribute__((always_inline)) inline void append(size_t n, char c) { need(n); memset(base + len, c, n); len += n; }
    __attribute__((always_inline)) inline void push_back(char c) { need(1); base[len++] = c; }
    __attribute__((always_inline)) inline Out& operator+=(const char* z) { append(z, __builtin_strlen(z)); return *this; }
    Out& operator+=(const std::string& z) { append(z.data(), z.size()); return *this; }
    void assign(const std::string& z) { len = 0; append(z.data(), z.size()); }
    const char* data() const { return base; }
    size_t size() const { return len; }
    void clear() { len = 0; }
};

// body of a JSON string (no quotes): runs of plain bytes are appended whole
template <class O>
void json_esc(O& o, const char* p, size_t n) {
    size_t i0 = 0;
    for (size_t i = 0; i < n; ++i) {
        const unsigned char c = (unsigned char)p[i];
        if (c >= 0x20 && c != '"' && c != '\\') continue;
        o.append(p + i0, i - i0);
        i0 = i + 1;
        switch (c) {
            case '"': o += "\\\""; break; case '\\': o += "\\\\"; break; case '\n': o += "\\n"; break;
            case '\r': o += "\\r"; break; case '\t': o += "\\t"; break; case '\b': o += "\\b"; break; case '\f': o += "\\f"; break;
            default: { char b[8]; snprintf(b, sizeof b, "\\u%04x", c); o += b; }
        }
    }
    o.append(p + i0, n - i0);
}
template <class O>
void json_str(O& o, std::string_view s) {
    o.push_back('"');
    json_esc(o, s.data(), s.size());
    o.push_back('"');
}
template <class O>
void json_f64(O& o, double v) {   // serde_json: shortest digits that round-trip, integral values keep ".0"
    if (!std::isfinite(v)) { o += "null"; return; }
    char b[64];
    auto r = std::to_chars(b, b + sizeof b, v);
    const size_t n = (size_t)(r.ptr - b);
    o.append(b, n);
    bool integral = true;
    for (size_t i = 0; i < n; ++i) if (b[i] == '.' || b[i] == 'e' || b[i] == 'E') { integral = false; break; }
    if (integral) o += ".0";
}

// serde_yaml 0.9 scalar style (third-party emitter, approximated): plain when the text cannot be read back as
// anything but that string, single quotes otherwise, double quotes when it holds control characters.
void yaml_str(std::string& o, const std::string& s) {
    static const char* reserved[] = {"null", "Null", "NULL", "~", "true", "True", "TRUE", "false", "False", "FALSE",
                                     "y", "Y", "yes", "Yes", "YES", "n", "N", "no", "No", "NO", "on", "On", "ON", "off", "Off", "OFF"};
    bool plain = !s.empty();
    bool control = false;
    for (unsigned char c : s) if (c < 0x20 || c == 0x7F) control = true;
    if (plain) {
        const unsigned char f = (unsigned char)s[0];
        if (!(isalpha(f) || f == '_' || f == '/')) plain = false;               // digits, '-', '.', indicators, spaces
        for (const char* r : reserved) if (s == r) plain = false;
        if (s.back() == ' ' || s.back() == ':') plain = false;
        if (s.find(": ") != std::string::npos || s.find(" #") != std::string::npos) plain = false;
        for (unsigned char c : s) if (c >= 0x80 || c == '"' || c == '\'' || c == '\\' || c == '`') { /* allowed in plain */ }
    }
    if (control) {
        o.push_back('"');
        for (unsigned char c : s) {
            if (c == '"' || c == '\\') { o.push_back('\\'); o.push_back((char)c); }
            else if (c == '\n') o += "\\n";
            else if (c == '\t') o += "\\t";
            else if (c < 0x20 || c == 0x7F) { char b[8]; snprintf(b, sizeof b, "\\x%02x", c); o += b; }
            else o.push_back((char)c);
        }
        o.push_back('"');
    } else if (plain) {
        o += s;
    } else {
        o.push_back('\'');
        for (char c : s) { if (c == '\'') o.push_back('\''); o.push_back(c); }
        o.push_back('\'');
    }
}

// The writer's view of one query's result: everything the reference's TaxonomyBean carries, as indices into the
// taxonomy / accession tables (no strings are built until the text is emitted).  The scratch vectors are reused from
// record to record by the worker that owns them.
struct Renderer {
    const Db& db;
    const std::vector<std::string>& accessions;
    const TopTable& top;
    const blu_taxonomy* tax;
    struct Bean { uint32_t node, desc_row; int32_t occurrences; const char* rank_serde; };
    struct Scratch {
        std::vector<uint32_t> S, bean_of, order;   // S: the top rows (indices into top.rows) in the reference's sorted order
        std::vector<Bean> beans;                   // first-seen order; `order` = the order they are written in
        std::string tmp;
    };
    struct View {
        const TopRow* ref = nullptr;
        uint32_t drow = 0;
        bool single = false, has_mar = false;
        const char* reached = nullptr;
        const char* mar = nullptr;
    };
    uint32_t len_of(uint32_t d) const { return (uint32_t)(db.lin_off[d + 1] - db.lin_off[d]); }
    const char* rank_serde_of(uint32_t desc_row, uint32_t level) const {
        uint16_t codes[BLU_MAX_DEPTH];
        blu_taxonomy_row_cutoffs(tax, desc_row, BLU_MAX_DEPTH, nullptr, nullptr, codes);
        return blu_taxonomy_rank_name(tax, codes[level], 1);
    }
    // taxonomy_beans_to_string (taxonomy_bean.rs:38-45) of the levels in `mask`; json: escaped for a JSON string body
    template <class O>
    void lineage(O& o, uint32_t desc_row, uint64_t mask, bool json) const {
        bool first = true;
        uint32_t jl = 0;
        for (uint64_t i = db.lin_off[desc_row]; i < db.lin_off[desc_row + 1]; ++i, ++jl) {
            if (!((mask >> jl) & 1)) continue;
            if (!first) o.push_back(';');
            first = false;
            const std::string& rk = db.rank_display[db.lin_rank[i]];
            const std::string& id = db.node_ident[db.lin_node[i]];
            if (json) { json_esc(o, rk.data(), rk.size()); o += "__"; json_esc(o, id.data(), id.size()); }
            else { o += rk; o += "__"; o += id; }
        }
    }
    void compute(uint64_t q, const blu_result& r, Scratch& sc, View& v) const {
        const uint64_t b = top.off[q], e = top.off[q + 1];
        const TopRow* rows = top.rows.data();
        v.ref = rows + b;
        for (uint64_t i = b; i < e; ++i) if (rows[i].row == r.ref_row) { v.ref = rows + i; break; }
        v.drow = v.ref->desc_row;
        v.single = r.status == BLU_ST_CONSENSUS_SINGLE;
        v.reached = blu_taxonomy_rank_name(tax, r.reached_rank, 1);
        v.has_mar = r.max_allowed_level != BLU_NONE_U8;
        if (v.has_mar) {
            uint8_t isdef[BLU_MAX_DEPTH]; uint16_t codes[BLU_MAX_DEPTH];
            blu_taxonomy_row_cutoffs(tax, v.drow, BLU_MAX_DEPTH, nullptr, isdef, codes);
            // DefaultRank(rank) -> rank (serde name); NonDefaultRank(name) -> Other(name) (build_blast_consensus_identity.rs:22-30)
            v.mar = blu_taxonomy_rank_name(tax, codes[r.max_allowed_level], isdef[r.max_allowed_level] ? 1 : 0);
        }
        sc.S.clear(); sc.bean_of.clear(); sc.beans.clear(); sc.order.clear();
        if (v.single) {   // find_single_query_consensus.rs:123-145
            sc.S.push_back((uint32_t)(v.ref - rows));
            sc.bean_of.push_back(0);
            sc.beans.push_back(Bean{r.identifier_node, v.drow, 1, v.reached});
            sc.order.push_back(0);
            return;
        }
        for (uint64_t i = b; i < e; ++i) sc.S.push_back((uint32_t)i);
        std::stable_sort(sc.S.begin(), sc.S.end(), [&](uint32_t x, uint32_t y) {   // find_multi_taxa_consensus.rs:39-54
            const TopRow &a = rows[x], &c = rows[y];
            const uint32_t la = len_of(a.desc_row), lc = len_of(c.desc_row);
            if (la != lc) return la < lc;
            if (a.pident < c.pident) return true;
            if (a.pident > c.pident) return false;
            if (a.align_len != c.align_len) return a.align_len < c.align_len;
            return a.acc_rank < c.acc_rank;
        });
        const uint32_t lvl = (r.flags & BLU_FLAG_AGREE) ? r.bean_index : (uint32_t)r.bean_index + 1;   // level the scan stopped at
        for (uint32_t i : sc.S) {   // consensus_result.rs:65-88 fold, first-seen order
            const uint32_t d = rows[i].desc_row;
            const uint32_t node = db.lin_node[db.lin_off[d] + lvl];
            uint32_t bi = 0;
            while (bi < sc.beans.size() && sc.beans[bi].node != node) ++bi;
            if (bi == sc.beans.size()) sc.beans.push_back(Bean{node, d, 0, rank_serde_of(d, lvl)});
            sc.bean_of.push_back(bi);
            sc.beans[bi].occurrences += 1;
        }
        for (uint32_t bi = 0; bi < sc.beans.size(); ++bi) sc.order.push_back(bi);
        std::stable_sort(sc.order.begin(), sc.order.end(), [&](uint32_t x, uint32_t y) {   // build_blast_consensus_identity.rs:49-60
            const Bean &a = sc.beans[x], &c = sc.beans[y];
            if (a.occurrences != c.occurrences) return a.occurrences > c.occurrences;
            return db.node_ident[a.node] < db.node_ident[c.node];
        });
    }
    // the accessions of bean `bi` in the order they were pushed, consecutive repeats dropped (extend + dedup())
    template <class F>
    void bean_accessions(const Scratch& sc, uint32_t bi, F&& f) const {
        uint32_t last = 0;
        bool any = false;
        for (size_t k = 0; k < sc.S.size(); ++k) {
            if (sc.bean_of[k] != bi) continue;
            const uint32_t a = top.rows[sc.S[k]].acc_rank;
            if (any && a == last) continue;
            any = true; last = a;
            f(accessions[a]);
        }
    }
    void taxon_yaml(std::string& o, uint64_t q, const blu_result& r, Scratch& sc) const {
        View v;
        compute(q, r, sc, v);
        auto kv = [&](const char* k) { o += "    "; o += k; o += ": "; };
        auto lin = [&](uint32_t d, uint64_t mask) { sc.tmp.clear(); lineage(sc.tmp, d, mask, false); yaml_str(o, sc.tmp); };
        kv("reachedRank"); yaml_str(o, v.reached); o.push_back('\n');
        kv("maxAllowedRank"); if (v.has_mar) yaml_str(o, v.mar); else o += "null"; o.push_back('\n');
        kv("identifier"); yaml_str(o, db.node_ident[r.identifier_node]); o.push_back('\n');
        kv("percIdentity"); json_f64(o, v.ref->pident); o.push_back('\n');
        kv("bitScore"); json_f64(o, (double)top.score[q]); o.push_back('\n');
        kv("taxonomy"); lin(v.drow, r.level_mask); o.push_back('\n');
        kv("mutated"); o += (r.flags & BLU_FLAG_MUTATED) ? "true" : "false"; o.push_back('\n');
        kv("singleMatch"); o += v.single ? "true" : "false"; o.push_back('\n');
        if (sc.beans.empty()) { kv("consensusBeans"); o += "[]\n"; return; }
        o += "    consensusBeans:\n";
        for (uint32_t bi : sc.order) {
            const Bean& b = sc.beans[bi];
            o += "    - rank: "; yaml_str(o, b.rank_serde); o.push_back('\n');
            o += "      identifier: "; yaml_str(o, db.node_ident[b.node]); o.push_back('\n');
            o += "      occurrences: " + std::to_string(b.occurrences) + "\n";
            o += "      taxonomy: "; lin(b.desc_row, ~0ull); o.push_back('\n');
            bool any = false;
            bean_accessions(sc, bi, [&](const std::string& a) {
                if (!any) o += "      accessions:\n";
                any = true;
                o += "      - "; yaml_str(o, a); o.push_back('\n');
            });
            if (!any) o += "      accessions: []\n";
        }
    }
    // pretty = serde_json::to_string_pretty layout (2-space indent); ind = indentation of the object's own line
    template <class O>
    void taxon(O& o, uint64_t q, const blu_result& r, bool pretty, int ind, Scratch& sc) const {
        View v;
        compute(q, r, sc, v);
        auto nl = [&](int extra) { if (pretty) { o.push_back('\n'); o.append((size_t)(ind + extra) * 2, ' '); } };
        const char* colon = pretty ? ": " : ":";
        auto key = [&](int extra, const char* k, bool comma = true) { if (comma) o.push_back(','); nl(extra); o += k; o += colon; };
        o.push_back('{');
        key(1, "\"reachedRank\"", false); json_str(o, v.reached);
        key(1, "\"maxAllowedRank\"");
        if (v.has_mar) json_str(o, v.mar); else o += "null";
        key(1, "\"identifier\""); json_str(o, db.node_ident[r.identifier_node]);
        key(1, "\"percIdentity\""); json_f64(o, v.ref->pident);
        key(1, "\"bitScore\""); json_f64(o, (double)top.score[q]);
        key(1, "\"taxonomy\""); o.push_back('"'); lineage(o, v.drow, r.level_mask, true); o.push_back('"');
        key(1, "\"mutated\""); o += (r.flags & BLU_FLAG_MUTATED) ? "true" : "false";
        key(1, "\"singleMatch\""); o += v.single ? "true" : "false";
        key(1, "\"consensusBeans\""); o.push_back('[');
        bool first_bean = true;
        for (uint32_t bi : sc.order) {
            const Bean& b = sc.beans[bi];
            if (!first_bean) o.push_back(',');
            first_bean = false;
            nl(2); o.push_back('{');
            key(3, "\"rank\"", false); json_str(o, b.rank_serde);
            key(3, "\"identifier\""); json_str(o, db.node_ident[b.node]);
            key(3, "\"occurrences\""); { char nb[16]; auto rr = std::to_chars(nb, nb + sizeof nb, b.occurrences); o.append(nb, (size_t)(rr.ptr - nb)); }
            key(3, "\"taxonomy\""); o.push_back('"'); lineage(o, b.desc_row, ~0ull, true); o.push_back('"');
            key(3, "\"accessions\""); o.push_back('[');
            bool any = false;
            bean_accessions(sc, bi, [&](const std::string& a) { if (any) o.push_back(','); any = true; nl(4); json_str(o, a); });
            if (any) nl(3);
            o.push_back(']');
            nl(2); o.push_back('}');
        }
        if (!first_bean) nl(1);
        o.push_back(']');
        nl(0); o.push_back('}');
    }
};

// TopTable from host columns (the CPU ingest, or the staging path of the engine): same content as the device-side
// compaction in ingest_gpu.hip
void top_rows_from_columns(const HitTable& ht, const Column<blu_result>& recs, unsigned nthreads, TopTable& top) {
    const size_t nq = recs.size();
    top.off.resize(nq + 1);
    top.score.resize(nq);
    std::vector<uint64_t> cnt(nq, 0);
    auto slice = [&](unsigned t, unsigned nt, auto&& f) { for (size_t q = nq * t / nt; q < nq * (t + 1) / nt; ++q) f(q); };
    const unsigned nt = nq < 65536 ? 1 : nthreads;
    parallel_for(nt, nt, [&](unsigned t) {
        slice(t, nt, [&](size_t q) {
            const blu_result& r = recs[q];
            top.score[q] = 0;
            if (r.status >= 2 || r.ref_row == 0xFFFFFFFFu) return;
            const int32_t M = ht.bitscore[r.ref_row];
            top.score[q] = M;
            uint64_t c = 0;
            for (uint64_t i = ht.seg_off[q]; i < ht.seg_off[q + 1]; ++i) c += ht.bitscore[i] == M;
            cnt[q] = c;
        });
    });
    top.off[0] = 0;
    for (size_t q = 0; q < nq; ++q) top.off[q + 1] = top.off[q] + cnt[q];
    top.rows.resize(top.off[nq]);
    parallel_for(nt, nt, [&](unsigned t) {
        slice(t, nt, [&](size_t q) {
            if (!cnt[q]) return;
            uint64_t at = top.off[q];
            const int32_t M = top.score[q];
            for (uint64_t i = ht.seg_off[q]; i < ht.seg_off[q + 1]; ++i)
                if (ht.bitscore[i] == M) top.rows[at++] = TopRow{(uint32_t)i, ht.tax_desc_row[i], ht.acc_rank[i], ht.align_len[i], ht.pident[i]};
        });
    });
}

std::string uuid_v4() {
    unsigned char b[16];
    FILE* f = fopen("/dev/urandom", "rb");
    if (!f || fread(b, 1, 16, f) != 16) for (auto& x : b) x = (unsigned char)rand();
    if (f) fclose(f);
    b[6] = (unsigned char)((b[6] & 0x0F) | 0x40);
    b[8] = (unsigned char)((b[8] & 0x3F) | 0x80);
    char s[40];
    snprintf(s, sizeof s, "%02x%02x%02x%02x-%02x%02x-%02x%02x-%02x%02x-%02x%02x%02x%02x%02x%02x", b[0], b[1], b[2], b[3], b[4], b[5],
             b[6], b[7], b[8], b[9], b[10], b[11], b[12], b[13], b[14], b[15]);
    return s;
}

const char* status_site(uint8_t st) {
    switch (st) {
        case BLU_ST_ERR_UNMATCHED_TAXID: return "taxid of a top-score hit is not in the taxonomies file: lineage `null` fails parse_taxonomy (find_single_query_consensus.rs:58-60)";
        case BLU_ST_ERR_BAD_LINEAGE: return "lineage of a top-score hit fails parse_taxonomy (blast_result.rs:109-114)";
        case BLU_ST_ERR_ROOT_DISAGREE: return "top-score hits disagree at the first lineage level: `index - 1` underflow (find_multi_taxa_consensus.rs:181)";
        case BLU_ST_ERR_SINGLE_BELOW_CUTOFFS: return "single top-score hit below every identity cutoff (find_single_query_consensus.rs:113-119)";
        case BLU_ST_ERR_BAD_PIDENT: return "NaN perc_identity among the top-score hits";
        default: return "unknown";
    }
}

// The serialized document as the pieces the workers rendered, in order: they are copied (or written) in parallel
// straight from the worker buffers, never concatenated.
struct Document {
    std::vector<Out> pieces;
    bool written = false;      // the pieces went to the output file while they were rendered (and were freed)
    size_t size() const { size_t n = 0; for (auto& p : pieces) n += p.size(); return n; }
};

bool write_all(int fd, const char* p, size_t left) {
    while (left) {
        const ssize_t w = write(fd, p, left);
        if (w < 0) { if (errno == EINTR) continue; return false; }
        p += w; left -= (size_t)w;
    }
    return true;
}
double thread_cpu_s() { timespec ts; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts); return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec; }

unsigned worker_threads() {
    unsigned nthreads = std::thread::hardware_concurrency();
    if (const char* env = getenv("BLU_INGEST_THREADS")) nthreads = (unsigned)atoi(env);
    if (nthreads < 1) nthreads = 1;
    if (nthreads > 32) nthreads = 32;
    return nthreads;
}

// f(k) for k in [0, n), handed out one at a time to `nthreads` workers
// Returns false if a worker ran out of memory (std::bad_alloc from an output buffer): an exception must not leave a
// std::thread — that is std::terminate for the whole host process — so it is caught here and the remaining work dropped.
template <class F>
bool parallel_dynamic(size_t n, unsigned nthreads, F&& f) {
    std::atomic<bool> oom{false};
    auto guarded = [&](size_t k) { try { f(k); } catch (const std::bad_alloc&) { oom = true; } };
    if (nthreads <= 1 || n <= 1) { for (size_t k = 0; k < n && !oom; ++k) guarded(k); return !oom; }
    std::atomic<size_t> next{0};
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nthreads && t < n; ++t)
        pool.emplace_back([&]() { for (size_t k = next.fetch_add(1); k < n && !oom.load(std::memory_order_relaxed); k = next.fetch_add(1)) guarded(k); });
    for (auto& th : pool) th.join();
    return !oom;
}

// out_path != nullptr: the document is written there (an existing file is replaced, write_blutils_output.rs:57-63) by a
// writer thread that follows the renderers piece by piece; otherwise the pieces are left in `document`.
int build_document(const char* blast_output_file, const char* const* headers, uint64_t n_headers,
                   const char* taxonomies_file, const blu_pipeline_params* params, const char* run_id_text,
                   const char* config_text, const char* out_path, Document* document, blu_pipeline_stats* stats) {
    if (!blast_output_file || !taxonomies_file || !params || !document) { set_error("null argument"); return BLU_ERR_INVALID_ARG; }
    blu_pipeline_stats st{};
    Trace tr;
    const unsigned nthreads = worker_threads();
    g_graveyard.drain();   // (a previous call's memory — device memory too — is back before this one sizes itself against what is free)
    double t0 = now_s();
    // (HIP start-up beside the reading of the taxonomy file; BLU_INGEST=cpu callers with no device never get here with one)
    std::thread warm_up;
    struct JoinWarmUp { std::thread& t; ~JoinWarmUp() { if (t.joinable()) t.join(); } } join_warm_up{warm_up};
    if (params->device >= 0) warm_up = std::thread([dev = params->device]() { warm_up_device(dev); });
    auto db_owner = std::make_unique<Db>();       // (on the heap, as the hit table below: handed to g_graveyard at the end)
    Db& db = *db_owner;
    int rc = load_db(taxonomies_file, params->use_taxid != 0, db);     // mod.rs:64
    if (rc != BLU_OK) return rc;
    st.t_load_db_s = now_s() - t0;
    tr.lap("load db");
    // the taxonomy table (sorted lineages, cutoff tables, upload) is built by a second thread while the hits are ingested
    std::vector<const char*> names;
    for (auto& s : db.rank_raw) names.push_back(s.c_str());
    blu_taxonomy_desc desc{};
    desc.n_tax = db.taxid.size();
    desc.taxid = db.taxid.data();
    desc.lin_off = db.lin_off.data();
    desc.lin_node = db.lin_node.data();
    desc.lin_rank = db.lin_rank.data();
    desc.n_ranks = (uint32_t)names.size();
    desc.rank_names = names.data();
    desc.bad = db.bad.data();
    blu_taxonomy* tax = nullptr;
    std::vector<uint32_t> fwd(std::max<size_t>(db.taxid.size(), 1));
    int tax_rc = BLU_OK;
    std::string tax_err;
    double t_tax = 0;
    std::thread tax_thread([&]() {
        const double ts = now_s();
        tax_rc = blu_taxonomy_create(&desc, &params->cutoffs, params->device, &tax);
        if (tax_rc == BLU_OK) blu_taxonomy_row_map(tax, fwd.data(), nullptr);
        else { char b[1024]; blu_last_error(b, sizeof b); tax_err = b; }
        t_tax = now_s() - ts;
    });
    struct TaxGuard { blu_taxonomy*& t; std::thread& th; ~TaxGuard() { if (th.joinable()) th.join(); if (t) blu_taxonomy_destroy(t); } } tax_guard{tax, tax_thread};
    t0 = now_s();
    auto ht_owner = std::make_unique<HitTable>();
    HitTable& ht = *ht_owner;
    rc = load_hits(blast_output_file, db, ht, params->device, /*host_columns=*/false);   // mod.rs:54, 72-82
    if (rc != BLU_OK) return rc;
    st.t_load_hits_s = now_s() - t0;
    tr.lap("load hits");
    st.n_hits = ht.n_hits; st.n_queries = ht.n_queries; st.n_taxids = db.taxid.size(); st.n_unmatched_rows = ht.unmatched;

    t0 = now_s();
    tax_thread.join();
    if (tax_rc == BLU_ERR_HIP || tax_rc == BLU_ERR_ALLOC) {
        // the table was built while the ingest held its work buffers (sized from a snapshot of the free memory that did not
        // know about it): on a nearly full card that can fail where the sequential order would not have — once more, now
        // that the ingest has handed its buffers back
        if (tr.on) fprintf(stderr, "[pipeline] taxonomy create failed beside the ingest (%s): once more\n", tax_err.c_str());
        tax_rc = blu_taxonomy_create(&desc, &params->cutoffs, params->device, &tax);
        if (tax_rc == BLU_OK) blu_taxonomy_row_map(tax, fwd.data(), nullptr);
        else { char b[1024]; blu_last_error(b, sizeof b); tax_err = b; }
    }
    if (tax_rc != BLU_OK) { set_error("%s", tax_err.c_str()); return tax_rc; }
    if (tr.on) fprintf(stderr, "[pipeline] (taxonomy create, 2nd thread %.3f s)\n", t_tax);
    tr.lap("wait for the taxonomy");
    Column<blu_result> recs;                      // (not zero-filled: the engine writes every record; the fresh pages are touched by
    recs.resize(ht.n_queries);                    //  the threads that copy the records back)
    TopTable top;
    bool done_on_device = false;
    // (BLU_PIPELINE_HOST_COLUMNS=1, tests: take the fallback below although the device path would work)
    const bool force_host = getenv("BLU_PIPELINE_HOST_COLUMNS") != nullptr;
    if (force_host && ht.dev) { rc = download_columns(ht); if (rc != BLU_OK) return rc; }
    if (ht.dev && !recs.empty() && !force_host) {
        // the GPU ingest left the grouped columns on the device: the engine reads them in place and only the records and
        // the top-score rows come back (if that fails — e.g. no room for the work buffers — the columns are downloaded and
        // go through the staging path below)
        done_on_device = device_run_consensus(tax, *ht.dev, fwd.data(), db.taxid.size(), params->strategy, recs.data(), &top) == BLU_OK;   // mod.rs:104-128
        if (!done_on_device) { rc = download_columns(ht); if (rc != BLU_OK) return rc; }
    }
    // (the device columns and the engine's work buffers stay with the hit table: they are freed off the caller's path at the end)
    if (!done_on_device) ht.dev.reset();
    if (!done_on_device) {
        std::vector<uint32_t> eng_rows(ht.tax_desc_row.size());
        for (size_t i = 0; i < eng_rows.size(); ++i)
            eng_rows[i] = ht.tax_desc_row[i] == BLU_UNMATCHED_TAXID ? BLU_UNMATCHED_TAXID : fwd[ht.tax_desc_row[i]];
        if (!recs.empty()) {
            blu_hits h{};
            h.bitscore = ht.bitscore.data(); h.tax_row = eng_rows.data();
            // 20 B/hit layout when every perc_identity is exactly k/1000 (BLAST prints <= 3 decimals): verified per value,
            // so the engine's fl(k / 1000.0) is bit-identical to the parsed f64; otherwise the f64 column goes over.
            std::vector<uint32_t> milli(ht.pident.size());
            bool exact = true;
            for (size_t i = 0; i < milli.size() && exact; ++i) {
                const double p = ht.pident[i];
                if (!(p >= 0.0 && p < 4.0e6)) { exact = false; break; }
                const uint32_t k = (uint32_t)(p * 1000.0 + 0.5);
                const double back = (double)k / 1000.0;
                exact = memcmp(&back, &p, 8) == 0;
                milli[i] = k;
            }
            if (exact) h.pident_milli = milli.data(); else h.pident = ht.pident.data();
            h.align_len = ht.align_len.data(); h.acc_rank = ht.acc_rank.data(); h.seg_off = ht.seg_off.data();
            h.n_hits = ht.bitscore.size(); h.n_queries = ht.n_queries; h.on_device = 0;
            blu_run_params rp{params->strategy, 0, nullptr};
            rc = blu_consensus_run(tax, &h, &rp, recs.data());             // mod.rs:104-128
            if (rc != BLU_OK) return rc;
        }
        top_rows_from_columns(ht, recs, nthreads, top);
    }
    st.t_engine_s = now_s() - t0;
    tr.lap("engine + top rows");

    t0 = now_s();
    ht.wait_strings();
    if (!ht.strings_ok) { set_error("out of memory while building the query / accession strings"); return BLU_ERR_ALLOC; }
    tr.lap("wait for the strings");
    // results + headers without hits (mod.rs:86-102), sorted by query (write_blutils_output.rs:111)
    struct Item { const std::string* name; int64_t q; };
    std::vector<Item> items;
    items.reserve(ht.query_names.size());
    for (size_t q = 0; q < ht.query_names.size(); ++q) items.push_back({&ht.query_names[q], (int64_t)q});
    std::vector<std::string> extra;
    if (headers) {
        std::unordered_map<std::string, uint32_t> have;
        for (size_t q = 0; q < ht.query_names.size(); ++q) have.emplace(ht.query_names[q], (uint32_t)q);
        extra.reserve(n_headers);
        for (uint64_t i = 0; i < n_headers; ++i)
            if (!have.count(headers[i])) extra.emplace_back(headers[i]);
        for (auto& s : extra) items.push_back({&s, -1});
    }
    {
        // stable sort by name: sorted runs per worker, then a tree of stable merges
        auto less = [](const Item& a, const Item& b) { return *a.name < *b.name; };
        const unsigned nt = items.size() < 65536 ? 1 : nthreads;
        const size_t n = items.size();
        // BLAST writes its queries in the order of the FASTA file, which is often sorted already: checked first (in slices)
        std::atomic<bool> sorted{true};
        parallel_for(nt, nt, [&](unsigned t) {
            const size_t i0 = n * t / nt, i1 = std::min(n, n * (t + 1) / nt + 1);
            if (i1 > i0 && !std::is_sorted(items.begin() + i0, items.begin() + i1, less)) sorted = false;
        });
        if (sorted) {}
        else if (nt == 1) std::stable_sort(items.begin(), items.end(), less);
        else {
            parallel_for(nt, nt, [&](unsigned t) { std::stable_sort(items.begin() + n * t / nt, items.begin() + n * (t + 1) / nt, less); });
            for (unsigned w = 1; w < nt; w *= 2) {
                std::vector<unsigned> heads;
                for (unsigned t = 0; t + w < nt; t += 2 * w) heads.push_back(t);
                parallel_for((unsigned)heads.size(), nt, [&](unsigned k) {
                    const unsigned t = heads[k];
                    std::inplace_merge(items.begin() + n * t / nt, items.begin() + n * (t + w) / nt, items.begin() + n * std::min(t + 2 * w, nt) / nt, less);
                });
            }
        }
    }
    tr.lap("sort by query");
    for (const Item& it : items) {
        if (it.q < 0) continue;
        const uint8_t s = recs[(size_t)it.q].status;
        if (s >= 16 && !params->lenient) {
            set_error("query `%s`: %s", it.name->c_str(), status_site(s));
            return BLU_ERR_REFERENCE_PANIC;
        }
    }
    const bool pretty = params->out_format == BLU_OUT_JSON;
    const bool doc = pretty || params->out_format == BLU_OUT_JSON_COMPACT;   // one {results, config} document
    // write_blutils_output.rs:82-85: the config's run id, or a fresh one
    const std::string run_id = (run_id_text && *run_id_text) ? std::string(run_id_text) : uuid_v4();
    const std::string cfg = (config_text && *config_text) ? std::string(config_text) : std::string();
    Renderer R{db, ht.accessions, top, tax};
    std::vector<Out>& pieces = document->pieces;
    pieces.clear();
    if (params->out_format == BLU_OUT_YAML) {
        std::string o;
        Renderer::Scratch sc;
        o += items.empty() ? "results: []\n" : "results:\n";
        for (const Item& it : items) {
            o += "- runId: "; yaml_str(o, run_id); o.push_back('\n');
            o += "  query: "; yaml_str(o, *it.name); o.push_back('\n');
            if (it.q < 0 || recs[(size_t)it.q].status >= 2) { o += "  taxon: null\n"; continue; }
            o += "  taxon:\n";
            R.taxon_yaml(o, (uint64_t)it.q, recs[(size_t)it.q], sc);
        }
        if (cfg.empty()) o += "config: null\n";
        else { o += "config:\n"; o += cfg; if (o.back() != '\n') o.push_back('\n'); }
        pieces.emplace_back();
        pieces.back().assign(o);
    } else {
        // records are independent: blocks of the sorted list are rendered by worker threads, one piece per block; with an
        // output file a writer thread sends the finished pieces out in order while the later ones are being rendered
        const size_t block = 4096, n_blocks = (items.size() + block - 1) / block;
        pieces.resize(n_blocks + 2);
        Out& head = pieces.front();
        if (pretty) head += "{\n  \"results\": [";
        else if (doc) head += "{\"results\":[";
        else { if (cfg.empty()) head += "null"; else head += cfg; head.push_back('\n'); }     // JSONL: the config line comes first
        std::string run_id_json;
        json_str(run_id_json, run_id);
        const char* const rid = run_id_json.data();
        const size_t rid_n = run_id_json.size();
        int fd = -1;
        std::thread old_file;   // releases the replaced file's pages off the caller's path
        struct Join { std::thread& t; ~Join() { if (t.joinable()) t.join(); } } join_old{old_file};
        if (out_path) {
            // write_blutils_output.rs:58-63: an existing file is removed, then a new one created.  (Truncating it in place
            // instead makes ext4 allocate and flush the new blocks inside close().)  The old inode is held open across the
            // unlink, so that dropping its pages happens in close(old) on a thread of its own.
            // Only a regular file is replaced that way: a FIFO would block in open(O_RDONLY), and a device node
            // (/dev/stdout, /dev/null) must be written to, not deleted.
            struct stat sb;
            if (stat(out_path, &sb) == 0 && S_ISREG(sb.st_mode)) {
                const int old = open(out_path, O_RDONLY | O_NONBLOCK);
                if (old >= 0) {
                    if (unlink(out_path) != 0) { close(old); set_error("cannot replace %s", out_path); return BLU_ERR_IO; }
                    old_file = std::thread([old]() { close(old); });
                }
            }
            fd = open(out_path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
            if (fd < 0) { set_error("cannot write %s", out_path); return BLU_ERR_IO; }
        }
        std::vector<uint8_t> ready(pieces.size(), 0);
        std::vector<Out> pool;      // buffers the writer is done with, taken again by the renderers (guarded by mu)
        std::mutex mu;
        std::condition_variable cv;
        bool write_ok = true;
        double t_write = 0;
        auto publish = [&](size_t k) { { std::lock_guard<std::mutex> lk(mu); ready[k] = 1; } cv.notify_all(); };
        std::thread writer;
        if (fd >= 0)
            writer = std::thread([&]() {
                for (size_t k = 0; k < pieces.size(); ++k) {
                    { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&]() { return ready[k] != 0; }); }
                    const double ts = now_s();
                    if (write_ok && !write_all(fd, pieces[k].data(), pieces[k].size())) write_ok = false;
                    t_write += now_s() - ts;
                    pieces[k].clear();
                    std::lock_guard<std::mutex> lk(mu);
                    pool.push_back(std::move(pieces[k]));
                }
            });
        // (should this thread leave by an exception — an allocation of its own failing — the writer is released and joined
        // first: a joinable std::thread must not be destroyed, and the writer must not wait for pieces that never come)
        struct WriterGuard {
            std::thread& w; std::vector<uint8_t>& ready; std::mutex& mu; std::condition_variable& cv; bool& ok;
            ~WriterGuard() {
                if (!w.joinable()) return;
                { std::lock_guard<std::mutex> lk(mu); ok = false; std::fill(ready.begin(), ready.end(), (uint8_t)1); }
                cv.notify_all();
                w.join();
            }
        } writer_guard{writer, ready, mu, cv, write_ok};
        publish(0);
        std::atomic<uint64_t> cpu_us{0};
        std::atomic<bool> render_oom{false};   // a worker ran out of memory: its piece and all later ones are published empty (the writer must not wait for them)
        parallel_dynamic(n_blocks, items.size() < 4096 ? 1 : nthreads, [&](size_t bk) {
            if (render_oom.load(std::memory_order_relaxed)) { publish(bk + 1); return; }
            try {
            thread_local Renderer::Scratch sc;
            const double tc = thread_cpu_s();
            Out po;
            if (fd >= 0) { std::lock_guard<std::mutex> lk(mu); if (!pool.empty()) { po = std::move(pool.back()); pool.pop_back(); } }
            const size_t i0 = bk * block, i1 = std::min(items.size(), i0 + block);
            po.reserve((i1 - i0) * (pretty ? 1100 : 520));
            const int ind = pretty ? 2 : 0;
            auto nl = [&](int extra) { if (pretty) { po.push_back('\n'); po.append((size_t)(ind + extra) * 2, ' '); } };
            const char* colon = pretty ? ": " : ":";
            for (size_t ii = i0; ii < i1; ++ii) {
                const Item& it = items[ii];
                if (pretty) { if (ii) po.push_back(','); nl(0); }
                else if (doc && ii) po.push_back(',');
                po.push_back('{');
                nl(1); po += "\"runId\""; po += colon; po.append(rid, rid_n);
                po.push_back(','); nl(1); po += "\"query\""; po += colon; json_str(po, *it.name);
                po.push_back(','); nl(1); po += "\"taxon\""; po += colon;
                if (it.q < 0 || recs[(size_t)it.q].status >= 2) po += "null";
                else R.taxon(po, (uint64_t)it.q, recs[(size_t)it.q], pretty, ind + 1, sc);
                nl(0); po.push_back('}');
                if (!doc) po.push_back('\n');
            }
            pieces[bk + 1] = std::move(po);
            cpu_us += (uint64_t)((thread_cpu_s() - tc) * 1e6);
            } catch (const std::bad_alloc&) { render_oom = true; }
            publish(bk + 1);
        });
        Out& tail = pieces.back();
        const std::string cfg_or_null = cfg.empty() ? std::string("null") : cfg;
        if (pretty) { if (!items.empty()) tail += "\n  "; tail += "],\n  \"config\": "; tail += cfg_or_null; tail += "\n}"; }
        else if (doc) { tail += "],\"config\":"; tail += cfg_or_null; tail.push_back('}'); }
        publish(pieces.size() - 1);
        if (tr.on) fprintf(stderr, "[pipeline] (render workers: %.3f s of CPU time in all)\n", (double)cpu_us.load() * 1e-6);
        tr.lap("render");
        if (fd >= 0) {
            writer.join();
            tr.lap("wait for the writer");
            const bool closed = close(fd) == 0;
            if (render_oom) { set_error("out of memory while rendering the results"); return BLU_ERR_ALLOC; }
            if (!closed || !write_ok) { set_error("cannot write %s", out_path); return BLU_ERR_IO; }
            document->written = true;
            if (tr.on) fprintf(stderr, "[pipeline] (writer thread: %.3f s inside write())\n", t_write);
            tr.lap("close file");
        }
        if (render_oom) { set_error("out of memory while rendering the results"); return BLU_ERR_ALLOC; }
    }
    st.t_render_s = now_s() - t0;
    if (stats) *stats = st;
    // the big objects and the taxonomy handle (whose 2 GB of packed staging take 25 ms to hipFree) go to the graveyard thread;
    // if it cannot be started they die here, as they would have anyway
    {
        blu_taxonomy* const tax_to_free = tax;
        tax = nullptr;                                // (tax_guard below finds nothing left to destroy)
        try {
            g_graveyard.bury(std::thread([tax_to_free, db_owner = std::move(db_owner), ht_owner = std::move(ht_owner), recs = std::move(recs),
                                          top = std::move(top), items = std::move(items)]() mutable { blu_taxonomy_destroy(tax_to_free); }));
        } catch (...) { blu_taxonomy_destroy(tax_to_free); }
    }
    tr.lap("hand the memory to its thread");
    g_t_body_end = now_s();
    return BLU_OK;
}

}  // namespace

extern "C" {

int blu_build_consensus_identities(const char* blast_output_file, const char* const* headers, uint64_t n_headers,
                                   const char* taxonomies_file, const blu_pipeline_params* params, char** out_text,
                                   size_t* out_len, blu_pipeline_stats* stats) {
    return blu_build_consensus_identities_cfg(blast_output_file, headers, n_headers, taxonomies_file, params, nullptr, nullptr,
                                              out_text, out_len, stats);
}

int blu_build_consensus_identities_cfg(const char* blast_output_file, const char* const* headers, uint64_t n_headers,
                                       const char* taxonomies_file, const blu_pipeline_params* params,
                                       const char* run_id_text, const char* config_text, char** out_text, size_t* out_len,
                                       blu_pipeline_stats* stats) {
    if (!out_text) { set_error("null argument"); return BLU_ERR_INVALID_ARG; }
    *out_text = nullptr;
    if (out_len) *out_len = 0;
    Document d;
    int rc;
    try { rc = build_document(blast_output_file, headers, n_headers, taxonomies_file, params, run_id_text, config_text, nullptr, &d, stats); }
    catch (const std::bad_alloc&) { set_error("out of memory"); return BLU_ERR_ALLOC; }   // (no exception crosses the C ABI)
    if (rc != BLU_OK) return rc;
    const size_t total = d.size();
    char* buf = (char*)malloc(total + 1);
    if (!buf) { set_error("out of memory"); return BLU_ERR_ALLOC; }
    std::vector<size_t> at(d.pieces.size() + 1, 0);
    for (size_t k = 0; k < d.pieces.size(); ++k) at[k + 1] = at[k] + d.pieces[k].size();
    parallel_dynamic(d.pieces.size(), total < (1u << 24) ? 1 : worker_threads(),
                     [&](size_t k) { memcpy(buf + at[k], d.pieces[k].data(), d.pieces[k].size()); });
    buf[total] = 0;
    *out_text = buf;
    if (out_len) *out_len = total;
    return BLU_OK;
}

int blu_build_consensus_identities_to_file(const char* blast_output_file, const char* const* headers, uint64_t n_headers,
                                           const char* taxonomies_file, const blu_pipeline_params* params,
                                           const char* run_id_text, const char* config_text, const char* out_path,
                                           blu_pipeline_stats* stats) {
    if (!out_path) { set_error("null argument"); return BLU_ERR_INVALID_ARG; }
    Document d;
    int rc;
    try { rc = build_document(blast_output_file, headers, n_headers, taxonomies_file, params, run_id_text, config_text, out_path, &d, stats); }
    catch (const std::bad_alloc&) { set_error("out of memory"); return BLU_ERR_ALLOC; }
    if (rc != BLU_OK) return rc;
    if (getenv("BLU_INGEST_TRACE")) fprintf(stderr, "[pipeline] %-26s %.3f s\n", "tear-down (tables, strings)", now_s() - g_t_body_end);
    if (d.written) return BLU_OK;
    // (YAML: one piece, written here) write_blutils_output.rs:57-63: an existing file is replaced
    Trace tr;
    const int fd = open(out_path, O_WRONLY | O_CREAT | O_TRUNC, 0644);
    if (fd < 0) { set_error("cannot write %s", out_path); return BLU_ERR_IO; }
    std::vector<size_t> at(d.pieces.size() + 1, 0);
    for (size_t k = 0; k < d.pieces.size(); ++k) at[k + 1] = at[k] + d.pieces[k].size();
    std::atomic<bool> ok{true};
    parallel_dynamic(d.pieces.size(), at.back() < (1u << 24) ? 1 : std::min(worker_threads(), 16u), [&](size_t k) {
        const char* p = d.pieces[k].data();
        size_t left = d.pieces[k].size(), off = at[k];
        while (left && ok.load(std::memory_order_relaxed)) {
            const ssize_t w = pwrite(fd, p, left, (off_t)off);
            if (w < 0) { if (errno == EINTR) continue; ok = false; break; }
            p += w; left -= (size_t)w; off += (size_t)w;
        }
    });
    if (close(fd) != 0 || !ok) { set_error("cannot write %s", out_path); return BLU_ERR_IO; }
    tr.lap("write file");
    return BLU_OK;
}

void blu_free_text(char* text) { free(text); }

int blu_ingest_only(const char* blast_output_file, const char* taxonomies_file, int use_taxid, blu_pipeline_stats* stats,
                    uint64_t* checksum) {
    return blu_ingest_only_on(blast_output_file, taxonomies_file, use_taxid, -1, stats, checksum);
}

int blu_ingest_only_on(const char* blast_output_file, const char* taxonomies_file, int use_taxid, int device,
                       blu_pipeline_stats* stats, uint64_t* checksum) {
    if (!blast_output_file || !taxonomies_file) { set_error("null argument"); return BLU_ERR_INVALID_ARG; }
    blu_pipeline_stats st{};
    double t0 = now_s();
    Db db;
    int rc = load_db(taxonomies_file, use_taxid != 0, db);
    if (rc != BLU_OK) return rc;
    st.t_load_db_s = now_s() - t0;
    t0 = now_s();
    HitTable ht;
    rc = load_hits(blast_output_file, db, ht, device);
    if (rc != BLU_OK) return rc;
    st.t_load_hits_s = now_s() - t0;
    st.n_hits = ht.bitscore.size(); st.n_queries = ht.n_queries; st.n_taxids = db.taxid.size(); st.n_unmatched_rows = ht.unmatched;
    if (stats) *stats = st;
    ht.wait_strings();
    if (!ht.strings_ok) { set_error("out of memory while building the query / accession strings"); return BLU_ERR_ALLOC; }
    if (checksum) {
        uint64_t h = 1469598103934665603ull;
        auto mix = [&](const void* p, size_t n) { const unsigned char* b = (const unsigned char*)p; for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } };
        mix(ht.seg_off.data(), ht.seg_off.size() * 8); mix(ht.bitscore.data(), ht.bitscore.size() * 4);
        mix(ht.align_len.data(), ht.align_len.size() * 4); mix(ht.tax_desc_row.data(), ht.tax_desc_row.size() * 4);
        mix(ht.acc_rank.data(), ht.acc_rank.size() * 4); mix(ht.pident.data(), ht.pident.size() * 8);
        for (auto& q : ht.query_names) mix(q.data(), q.size() + 1);
        for (auto& a : ht.accessions) mix(a.data(), a.size() + 1);
        *checksum = h;
    }
    return BLU_OK;
}

int blu_ingest_columns_on(const char* blast_output_file, const char* taxonomies_file, int use_taxid, int device,
                          blu_ingest_columns* out) {
    if (!blast_output_file || !taxonomies_file || !out) { set_error("null argument"); return BLU_ERR_INVALID_ARG; }
    memset(out, 0, sizeof *out);
    Db db;
    int rc = load_db(taxonomies_file, use_taxid != 0, db);
    if (rc != BLU_OK) return rc;
    HitTable ht;
    rc = load_hits(blast_output_file, db, ht, device);
    if (rc != BLU_OK) return rc;
    ht.wait_strings();
    if (!ht.strings_ok) { set_error("out of memory while building the query / accession strings"); return BLU_ERR_ALLOC; }
    const size_t nh = ht.bitscore.size(), nq = ht.query_names.size();
    auto dup = [](const void* src, size_t bytes) -> void* { void* p = malloc(bytes ? bytes : 1); if (p && bytes) memcpy(p, src, bytes); return p; };
    auto pack = [](const std::vector<std::string>& v, uint64_t* bytes) -> char* {
        size_t n = 0;
        for (auto& s : v) n += s.size() + 1;
        char* p = (char*)malloc(n ? n : 1);
        if (!p) return nullptr;
        size_t at = 0;
        for (auto& s : v) { memcpy(p + at, s.data(), s.size()); at += s.size(); p[at++] = 0; }
        *bytes = n;
        return p;
    };
    out->n_hits = nh; out->n_queries = nq; out->n_accessions = ht.accessions.size();
    out->seg_off = (uint64_t*)dup(ht.seg_off.data(), (nq + 1) * 8);
    out->bitscore = (int32_t*)dup(ht.bitscore.data(), nh * 4);
    out->align_len = (int32_t*)dup(ht.align_len.data(), nh * 4);
    out->tax_desc_row = (uint32_t*)dup(ht.tax_desc_row.data(), nh * 4);
    out->acc_rank = (uint32_t*)dup(ht.acc_rank.data(), nh * 4);
    out->pident = (double*)dup(ht.pident.data(), nh * 8);
    out->query_names = pack(ht.query_names, &out->query_names_bytes);
    out->accessions = pack(ht.accessions, &out->accessions_bytes);
    if (!out->seg_off || !out->bitscore || !out->align_len || !out->tax_desc_row || !out->acc_rank || !out->pident || !out->query_names || !out->accessions) {
        blu_ingest_columns_free(out);
        set_error("out of memory");
        return BLU_ERR_ALLOC;
    }
    return BLU_OK;
}

void blu_ingest_columns_free(blu_ingest_columns* c) {
    if (!c) return;
    free(c->seg_off); free(c->bitscore); free(c->align_len); free(c->tax_desc_row); free(c->acc_rank); free(c->pident);
    free(c->query_names); free(c->accessions);
    memset(c, 0, sizeof *c);
}

// domain/dtos/taxon.rs:28-66: YAML (flat `key: value` lines) or JSON object with the eight fields
int blu_last_ingest_path(void) { return g_last_ingest_path; }

int blu_db_cache_build(const char* taxonomies_file, int use_taxid, const char* cache_file) {
    if (!taxonomies_file || !cache_file) { set_error("null argument"); return BLU_ERR_INVALID_ARG; }
    Db db;
    int rc = load_db(taxonomies_file, use_taxid != 0, db);
    if (rc != BLU_OK) return rc;
    return write_db_cache(db, use_taxid != 0, cache_file);
}

int blu_custom_taxon_from_file(const char* path, blu_cutoff_config* cfg) {
    if (!path || !cfg) { set_error("null argument"); return BLU_ERR_INVALID_ARG; }
    const char* dot = strrchr(path, '.');
    if (!dot || (strcmp(dot, ".yaml") != 0 && strcmp(dot, ".json") != 0)) { set_error("Custom taxon file must be a YAML or JSON file"); return BLU_ERR_INVALID_ARG; }
    MappedFile f;
    if (!f.open(path)) { set_error("Could not open custom taxon file: %s", path); return BLU_ERR_IO; }
    static const char* fields[8] = {"domain", "kingdom", "phylum", "class", "order", "family", "genus", "species"};
    memset(cfg, 0, sizeof *cfg);
    cfg->taxon = BLU_TAXON_CUSTOM;
    cfg->has_custom = 1;
    auto set = [&](const std::string& k, double v) { for (int i = 0; i < 8; ++i) if (k == fields[i]) { cfg->custom[i] = (int16_t)v; cfg->custom_has[i] = 1; } };
    if (strcmp(dot, ".json") == 0) {
        Json j{f.data, f.data + f.size};
        std::string key;
        if (!j.eat('{')) { set_error("Could not parse custom taxon file from JSON"); return BLU_ERR_PARSE; }
        if (!j.peek('}')) do {
            double v;
            if (!j.string(&key) || !j.eat(':')) { j.ok = false; break; }
            if (j.peek('n')) { j.skip(); continue; }
            if (!j.number(&v)) break;
            set(key, v);
        } while (j.ok && j.eat(','));
        if (!j.ok) { set_error("Could not parse custom taxon file from JSON"); return BLU_ERR_PARSE; }
    } else {
        std::string text(f.data, f.size);
        size_t pos = 0;
        while (pos < text.size()) {
            size_t nl = text.find('\n', pos);
            std::string line = text.substr(pos, nl == std::string::npos ? std::string::npos : nl - pos);
            pos = nl == std::string::npos ? text.size() : nl + 1;
            size_t hash = line.find('#');
            if (hash != std::string::npos) line.resize(hash);
            size_t colon = line.find(':');
            if (colon == std::string::npos) continue;
            std::string k = line.substr(0, colon), v = line.substr(colon + 1);
            auto trim = [](std::string& s) { size_t a = s.find_first_not_of(" \t\r"), b = s.find_last_not_of(" \t\r"); s = a == std::string::npos ? "" : s.substr(a, b - a + 1); };
            trim(k); trim(v);
            if (v.empty() || v == "null" || v == "~") continue;
            set(k, atof(v.c_str()));
        }
    }
    if (!cfg->custom_has[0] || !cfg->custom_has[7]) { set_error("custom taxon file lacks the mandatory `domain`/`species` fields (taxon.rs:16-25)"); return BLU_ERR_PARSE; }
    return BLU_OK;
}

}  // extern "C"
