// Measurement aid (not product code): where the time of "text file in the page cache -> HBM" goes on a box.
//   upload_probe FILE [threads=6] [piece_MiB=8]
// Prints seconds and GB/s for
//   alloc     hipMalloc of the device buffer, hipHostMalloc of the staging slots
//   dma       the slots sent over and over with no file read at all          (what PCIe gives this process)
//   pread     the file read into the slots with nothing sent                 (what the page-cache copy gives)
//   both      the product's loop: pread a piece while the previous one is on the wire (ingest_gpu.hip upload_file)
//   mapped    the file mapped (MAP_POPULATE) and handed to hipMemcpyAsync piece by piece by the same threads: the runtime
//             pins the page-cache pages and the copy engine reads them, no CPU copy
//   munmap    what unmapping the populated mapping costs afterwards
// DESIGN.md §7 quotes the lines next to the end-to-end figure.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

static double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

enum Mode { DMA_ONLY, PREAD_ONLY, BOTH, MAPPED };

struct Worker { char* slots; hipStream_t st; hipEvent_t ev[2]; };

static double run(Mode mode, int fd, size_t size, unsigned char* d_text, const char* map, std::vector<Worker>& w, size_t piece) {
    const size_t n_pieces = (size + piece - 1) / piece;
    std::atomic<size_t> next{0};
    const double t0 = now_s();
    auto work = [&](unsigned t) {
        CHECK(hipSetDevice(0));
        Worker& me = w[t];
        bool busy[2] = {false, false};
        unsigned turn = 0;
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= n_pieces) break;
            const unsigned sl = turn++ & 1;
            const size_t off = k * piece, len = std::min(piece, size - off);
            if (mode == MAPPED) { CHECK(hipMemcpyAsync(d_text + off, map + off, len, hipMemcpyHostToDevice, me.st)); continue; }
            if (busy[sl]) CHECK(hipEventSynchronize(me.ev[sl]));
            if (mode != DMA_ONLY) {
                size_t got = 0;
                while (got < len) {
                    const ssize_t r = pread(fd, me.slots + sl * piece + got, len - got, (off_t)(off + got));
                    if (r <= 0) { perror("pread"); exit(2); }
                    got += (size_t)r;
                }
            }
            if (mode != PREAD_ONLY) {
                CHECK(hipMemcpyAsync(d_text + off, me.slots + sl * piece, len, hipMemcpyHostToDevice, me.st));
                CHECK(hipEventRecord(me.ev[sl], me.st));
                busy[sl] = true;
            }
        }
        CHECK(hipStreamSynchronize(me.st));
    };
    std::vector<std::thread> pool;
    for (unsigned t = 1; t < w.size(); ++t) pool.emplace_back(work, t);
    work(0);
    for (auto& th : pool) th.join();
    return now_s() - t0;
}

int main(int argc, char** argv) {
    if (argc < 2) { fprintf(stderr, "usage: upload_probe FILE [threads] [piece_MiB]\n"); return 1; }
    const unsigned nt = argc > 2 ? (unsigned)atoi(argv[2]) : 6;
    const size_t piece = (size_t)(argc > 3 ? atoi(argv[3]) : 8) << 20;
    const int fd = open(argv[1], O_RDONLY);
    if (fd < 0) { perror("open"); return 1; }
    struct stat sb;
    fstat(fd, &sb);
    const size_t size = (size_t)sb.st_size;
    const double gb = size / 1e9;
    double t = now_s();
    CHECK(hipSetDevice(0));
    CHECK(hipFree(nullptr));
    printf("%-10s %.3f s\n", "start-up", now_s() - t);
    t = now_s();
    unsigned char* d_text = nullptr;
    CHECK(hipMalloc((void**)&d_text, size + 64));
    printf("%-10s %.3f s  (hipMalloc of %.2f GB)\n", "alloc", now_s() - t, gb);
    std::vector<Worker> w(nt);
    // set-up, call by call: one pinned block for all the slots, then a stream and two events per reader
    t = now_s();
    char* block = nullptr;
    CHECK(hipHostMalloc((void**)&block, (size_t)nt * 2 * piece, hipHostMallocDefault));
    printf("%-10s %.3f s  (hipHostMalloc of %u x 2 slots of %zu MiB in one block)\n", "alloc", now_s() - t, nt, piece >> 20);
    for (unsigned k = 0; k < nt; ++k) {
        w[k].slots = block + (size_t)k * 2 * piece;
        t = now_s();
        CHECK(hipStreamCreateWithFlags(&w[k].st, hipStreamNonBlocking));
        const double ts = now_s() - t;
        t = now_s();
        for (auto& e : w[k].ev) CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        printf("%-10s %.3f s stream %u, %.3f s its two events\n", "alloc", ts, k, now_s() - t);
    }
    t = now_s();
    { char* extra = nullptr; CHECK(hipHostMalloc((void**)&extra, 2 * piece, hipHostMallocDefault)); printf("%-10s %.3f s  (a second hipHostMalloc, 2 slots)\n", "alloc", now_s() - t);
      t = now_s(); CHECK(hipHostFree(extra)); printf("%-10s %.3f s  (hipHostFree of it)\n", "alloc", now_s() - t); }
    auto line = [&](const char* what, double s) { printf("%-10s %.3f s  %.1f GB/s\n", what, s, gb / s); fflush(stdout); };
    for (int rep = 0; rep < 2; ++rep) {
        line("dma", run(DMA_ONLY, fd, size, d_text, nullptr, w, piece));
        line("pread", run(PREAD_ONLY, fd, size, d_text, nullptr, w, piece));
        line("both", run(BOTH, fd, size, d_text, nullptr, w, piece));
    }
    for (int rep = 0; rep < 2; ++rep) {
        t = now_s();
        void* m = mmap(nullptr, size, PROT_READ, MAP_SHARED | MAP_POPULATE, fd, 0);
        if (m == MAP_FAILED) { perror("mmap"); return 1; }
        const double t_map = now_s() - t;
        const double t_copy = run(MAPPED, fd, size, d_text, (const char*)m, w, std::max<size_t>(piece, 64u << 20));
        t = now_s();
        munmap(m, size);
        const double t_unmap = now_s() - t;
        printf("%-10s map+populate %.3f s, copies %.3f s (%.1f GB/s), munmap %.3f s: %.3f s in all = %.1f GB/s\n", "mapped", t_map, t_copy, gb / t_copy,
               t_unmap, t_map + t_copy + t_unmap, gb / (t_map + t_copy + t_unmap));
    }
    return 0;
}
