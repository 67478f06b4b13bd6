// Measurement aid (not product code): does ANY cache policy on a load make the L2 fetch less than a whole 128-byte line
// from HBM?  The stream kernel needs 16-48 bytes of side records and 4-40 bytes of a lineage row per query and pays a
// 128-byte line for each (2.8 of its 5.6 GB per launch); a 32- or 64-byte fetch would cut that in half or more.
//   sector_probe [rows_log2=23]      (2^23 rows x 128 B = 1 GB: four times the Infinity Cache; 24 at most)
// Every lane gathers ONE 16-byte (or 4-byte) piece out of a random 128-byte row, 8 loads in flight per lane, 16 waves per
// CU, with the policy bits of the load varied: none, nt, sc0, sc1, sc0 sc1, sc0 sc1 nt.  Printed: gathers per second and
// what that is in 128-byte lines per second against the box's streaming rate.  If a policy fetched sectors, its gather
// rate would exceed (streaming bytes/s) / 128.  Outcome: none does — every variant runs at 54-56 G gathers/s,
// the streaming rate of the same table counted in 128-byte lines (7.05 TB/s on that box).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// (compiler builtins, not inline asm: an asm load's output registers are "defined" for the compiler as soon as the statement
// has issued, so it reuses them — for the next load's address, say — before the data has arrived: the first version of this
// probe faulted that way.)  AUX bits of the gfx94x/gfx950 buffer loads: 1 = sc0, 2 = nt, 16 = sc1.
template <int AUX, int WIDTH>
__global__ __launch_bounds__(1024) void gather_kernel(const unsigned char* __restrict__ table, uint32_t row_mask, uint32_t iters, uint32_t* sink) {
    const uint32_t tid = blockIdx.x * blockDim.x + threadIdx.x;
    const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)table, 0, (row_mask + 1u) << 7, 0x00020000);   // (<= 2 GB: byte offsets fit 32 bits)
    uint32_t acc = 0, seed = tid * 0x9E3779B9u + 12345u;
    for (uint32_t it = 0; it < iters; ++it) {
        u32x4 v[8];
        uint32_t w[8], off[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            seed = mix32(seed + k + 1);
            off[k] = ((seed & row_mask) << 7) + ((seed >> 27) & 7u) * 16u;   // a random 16-byte piece of a random row
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (WIDTH == 16) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, off[k], 0, AUX);
            else w[k] = __builtin_amdgcn_raw_buffer_load_b32(rs, off[k], 0, AUX);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc ^= WIDTH == 16 ? (v[k].x ^ v[k].w) : w[k];
    }
    if (acc == 0x12345678u) *sink = acc;
}

__global__ void stream_kernel(const u32x4* __restrict__ p, uint64_t n16, uint32_t* sink) {
    uint32_t acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (uint64_t)gridDim.x * blockDim.x) { const u32x4 v = __builtin_nontemporal_load(p + i); acc ^= v.x ^ v.w; }
    if (acc == 0x12345678u) *sink = acc;
}

template <int AUX, int WIDTH>
static double run(const unsigned char* table, uint32_t row_mask, uint32_t iters, uint32_t* sink, int cus) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((gather_kernel<AUX, WIDTH>), dim3(cus), dim3(1024), 0, 0, table, row_mask, 4u, sink);     // warm-up
    CHECK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL((gather_kernel<AUX, WIDTH>), dim3(cus), dim3(1024), 0, 0, table, row_mask, iters, sink);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e-3;
}

int main(int argc, char** argv) {
    int lg = argc > 1 ? atoi(argv[1]) : 23;
    if (lg < 10 || lg > 24) lg = 24;             // (the buffer descriptor takes 32-bit byte offsets: 2 GB at most here)
    setvbuf(stdout, nullptr, _IOLBF, 0);
    const uint64_t rows = 1ull << lg, bytes = rows * 128;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    unsigned char* table = nullptr;
    uint32_t* sink = nullptr;
    CHECK(hipMalloc((void**)&table, bytes));
    CHECK(hipMalloc((void**)&sink, 64));
    CHECK(hipMemset(table, 1, bytes));
    // streaming rate of the box over the same table
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(stream_kernel, dim3(cus * 8), dim3(512), 0, 0, (const u32x4*)table, bytes / 16, sink);
    CHECK(hipEventRecord(e0, 0));
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(stream_kernel, dim3(cus * 8), dim3(512), 0, 0, (const u32x4*)table, bytes / 16, sink);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double stream = 4.0 * bytes / (ms * 1e-3);
    printf("table %.2f GB, %d CUs; streaming %.2f TB/s = %.1f G lines/s\n", bytes / 1e9, cus, stream / 1e12, stream / 128 / 1e9);
    const uint32_t iters = 256;
    const double n = (double)cus * 1024 * iters * 8;
    const char* names[6] = {"(none)", "nt", "sc0", "sc1", "sc0 sc1", "sc0 sc1 nt"};
    const uint32_t m = (uint32_t)rows - 1;
    double t16[6] = {run<0, 16>(table, m, iters, sink, cus), run<2, 16>(table, m, iters, sink, cus), run<1, 16>(table, m, iters, sink, cus),
                     run<16, 16>(table, m, iters, sink, cus), run<17, 16>(table, m, iters, sink, cus), run<19, 16>(table, m, iters, sink, cus)};
    double t4[6] = {run<0, 4>(table, m, iters, sink, cus), run<2, 4>(table, m, iters, sink, cus), run<1, 4>(table, m, iters, sink, cus),
                    run<16, 4>(table, m, iters, sink, cus), run<17, 4>(table, m, iters, sink, cus), run<19, 4>(table, m, iters, sink, cus)};
    for (int p = 0; p < 6; ++p)
        printf("%-12s 16-byte gathers %.2f G/s (as whole lines: %.2f TB/s = %.2f of streaming) | 4-byte gathers %.2f G/s (%.2f of streaming)\n", names[p],
               n / t16[p] / 1e9, n / t16[p] * 128 / 1e12, n / t16[p] * 128 / stream, n / t4[p] / 1e9, n / t4[p] * 128 / stream);
    return 0;
}
