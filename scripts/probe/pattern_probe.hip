// Measurement aid (not product code): HBM rate of the streaming kernel's ACCESS PATTERN alone —
// five SoA columns, one wave per 50-row segment (lane = row, 50 of 64 lanes), U segments in flight
// per wave, waves striding over 64-segment tasks — with no reduction work behind the loads.
#include <hip/hip_runtime.h>
#include <cstdint>

template <int U, int W>   // U segments in flight; W = 1: dword loads lane=row, W = 4: 16 lanes x 4 rows (dwordx4)
__global__ __launch_bounds__(256) void pattern_kernel(const int* __restrict__ c0, const int* __restrict__ c1,
                                                      const double* __restrict__ c2, const int* __restrict__ c3,
                                                      const int* __restrict__ c4, uint64_t n_seg, int seg, uint32_t* sink, uint4* __restrict__ outp, int store_mode) {
    const int lane = threadIdx.x & 63;
    const uint64_t wave = (uint64_t)blockIdx.x * 4 + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * 4;
    const uint64_t n_tasks = n_seg / 64;
    extern __shared__ uint32_t dyn_lds[];
    uint32_t acc = 0;
    if (n_seg == 1) dyn_lds[threadIdx.x] = 1;   // keeps the allocation; never true in the probe
    for (uint64_t task = wave; task < n_tasks; task += n_waves) {
        const uint64_t base = task * 64 * seg;
        if (W == 1) {
            for (int q = 0; q < 64; q += U) {
                int a[U], b[U], d[U], e[U]; double c[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint64_t row = base + (uint64_t)(q + u) * seg + lane;
                    a[u] = 0; b[u] = 0; c[u] = 0; d[u] = 0; e[u] = 0;
                    if (lane < seg) {
                        a[u] = __builtin_nontemporal_load(c0 + row); b[u] = __builtin_nontemporal_load(c1 + row);
                        c[u] = __builtin_nontemporal_load(c2 + row); d[u] = __builtin_nontemporal_load(c3 + row);
                        e[u] = __builtin_nontemporal_load(c4 + row);
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) acc ^= a[u] ^ b[u] ^ d[u] ^ e[u] ^ (uint32_t)__double2loint(c[u]);
            }
        } else if (W == 5) {
            typedef int i4 __attribute__((ext_vector_type(4), aligned(4)));
            const int g = lane >> 4, sub = lane & 15;
            for (int q = 0; q < 64; q += 4 * U) {
                i4 a[U];
                uint64_t row[U];
                int top0[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint64_t segi = task * 64 + q + 4 * u + g;
                    row[u] = base + (uint64_t)(q + 4 * u + g) * seg + 4 * sub;
                    a[u] = 0;
                    if (4 * sub < seg) a[u] = *(const i4*)(c0 + row[u]);
                    top0[u] = (int)((segi * 2654435761ull) % (uint64_t)(seg - 3));   // rows top0..top0+2 are the top group
                }
                int x[U]; double y[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    x[u] = a[u].x ^ a[u].y ^ a[u].z ^ a[u].w; y[u] = 0;   // consumes the bitscores first (dependent phase)
                    const int bias = (x[u] == 0x7fffffff) ? 1 : 0;
                    for (int r = 0; r < 4; ++r) {
                        const int rr = 4 * sub + r;
                        if (rr >= top0[u] + bias && rr < top0[u] + 3 && rr < seg) { x[u] ^= c1[row[u] + r] ^ c3[row[u] + r] ^ c4[row[u] + r]; y[u] += c2[row[u] + r]; }
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) acc ^= x[u] ^ (uint32_t)__double2loint(y[u]);
            }
        } else {
            // 4 segments per wave instruction: 16 lanes x 4 consecutive rows
            typedef int i4 __attribute__((ext_vector_type(4), aligned(4)));
            typedef double d2 __attribute__((ext_vector_type(2), aligned(8)));
            const int g = lane >> 4, sub = lane & 15;
            for (int q = 0; q < 64; q += 4 * U) {
                i4 a[U], b[U], d[U], e[U]; d2 c[U], c2b[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const uint64_t row = base + (uint64_t)(q + 4 * u + g) * seg + 4 * sub;
                    a[u] = 0; b[u] = 0; d[u] = 0; e[u] = 0; c[u] = 0; c2b[u] = 0;
                    if (4 * sub < seg) {
                        a[u] = __builtin_nontemporal_load((const i4*)(c0 + row)); b[u] = __builtin_nontemporal_load((const i4*)(c1 + row));
                        c[u] = __builtin_nontemporal_load((const d2*)(c2 + row)); c2b[u] = __builtin_nontemporal_load((const d2*)(c2 + row + 2));
                        d[u] = __builtin_nontemporal_load((const i4*)(c3 + row)); e[u] = __builtin_nontemporal_load((const i4*)(c4 + row));
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    acc ^= a[u].x ^ a[u].w ^ b[u].y ^ d[u].z ^ e[u].x ^ (uint32_t)__double2loint(c[u].x) ^ (uint32_t)__double2loint(c2b[u].y);
            }
        }
        if (store_mode == 1) {          // 2 coalesced 1 KiB rows per 64 segments
            outp[task * 128 + lane] = make_uint4(acc, 1, 2, 3);
            outp[task * 128 + 64 + lane] = make_uint4(acc, 4, 5, 6);
        } else if (store_mode == 2) {   // 32 B per lane, strided
            outp[task * 128 + 2 * lane] = make_uint4(acc, 1, 2, 3);
            outp[task * 128 + 2 * lane + 1] = make_uint4(acc, 4, 5, 6);
        } else if (store_mode >= 16) {  // coalesced buffer stores with cache-policy bits = store_mode (16 sc1, 17 sc0 sc1, 18 sc1 nt, 19 all)
            typedef uint32_t u4 __attribute__((ext_vector_type(4)));
            const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(outp + task * 128), 0, 2048, 0x00020000);
            u4 v = {acc, 1, 2, 3};
            if (store_mode == 16) { __builtin_amdgcn_raw_buffer_store_b128(v, rs, lane * 16, 0, 16); __builtin_amdgcn_raw_buffer_store_b128(v, rs, 1024 + lane * 16, 0, 16); }
            else if (store_mode == 17) { __builtin_amdgcn_raw_buffer_store_b128(v, rs, lane * 16, 0, 17); __builtin_amdgcn_raw_buffer_store_b128(v, rs, 1024 + lane * 16, 0, 17); }
            else if (store_mode == 18) { __builtin_amdgcn_raw_buffer_store_b128(v, rs, lane * 16, 0, 18); __builtin_amdgcn_raw_buffer_store_b128(v, rs, 1024 + lane * 16, 0, 18); }
            else { __builtin_amdgcn_raw_buffer_store_b128(v, rs, lane * 16, 0, 1); __builtin_amdgcn_raw_buffer_store_b128(v, rs, 1024 + lane * 16, 0, 1); }
        } else if (store_mode == 7 || store_mode == 8) {   // sc1 nt bursts: 8 KiB every 4th task (7) / 32 KiB every 16th task (8)
            typedef uint32_t u4 __attribute__((ext_vector_type(4)));
            const uint64_t it = task / n_waves;
            const int per = store_mode == 7 ? 4 : 16;
            if ((it % per) == (uint64_t)(per - 1)) {
                const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)(outp + (task - (per - 1) * n_waves) * 128), 0, 2048 * per, 0x00020000);
                u4 v = {acc, 1, 2, 3};
                for (int j = 0; j < 2 * per; ++j) __builtin_amdgcn_raw_buffer_store_b128(v, rs, j * 1024 + lane * 16, 0, 18);
            }
        } else if (store_mode == 4) {   // half the bytes: 16 B per segment
            outp[task * 64 + lane] = make_uint4(acc, 1, 2, 3);
        } else if (store_mode == 5) {   // same bytes, 4x fewer/larger bursts: 8 KiB every 4th task of the wave
            const uint64_t it = task / n_waves;
            if ((it & 3) == 3) {
                for (int j = 0; j < 8; ++j) outp[(task - 3 * n_waves) * 128 + 0 + 64 * j + lane] = make_uint4(acc, j, 2, 3);
            }
        } else if (store_mode == 6) {   // same bytes, 16x larger bursts: 32 KiB every 16th task
            const uint64_t it = task / n_waves;
            if ((it & 15) == 15) {
                for (int j = 0; j < 32; ++j) outp[(task - 15 * n_waves) * 128 + 64 * j + lane] = make_uint4(acc, j, 2, 3);
            }
        } else if (store_mode == 3) {   // nontemporal coalesced
            typedef uint32_t u4 __attribute__((ext_vector_type(4)));
            u4 v = {acc, 1, 2, 3};
            __builtin_nontemporal_store(v, (u4*)(outp + task * 128 + lane));
            __builtin_nontemporal_store(v, (u4*)(outp + task * 128 + 64 + lane));
        }
    }
    if (acc == 0x12345678u) *sink = acc;
}

#define LAUNCH(U, W) hipLaunchKernelGGL((pattern_kernel<U, W>), dim3(grid), dim3(256), lds_bytes, (hipStream_t)stream, (const int*)c0, (const int*)c1, (const double*)c2, (const int*)c3, (const int*)c4, n_seg, seg, (uint32_t*)sink, (uint4*)outp, store_mode)
extern "C" int pattern_run(const void* c0, const void* c1, const void* c2, const void* c3, const void* c4, uint64_t n_seg,
                           int seg, void* sink, int grid, int U, int W, void* stream, int lds_bytes, void* outp, int store_mode) {
    if (W == 1) { if (U == 2) LAUNCH(2, 1); else if (U == 4) LAUNCH(4, 1); else if (U == 8) LAUNCH(8, 1); else LAUNCH(16, 1); }
    else if (W == 5) { if (U == 1) LAUNCH(1, 5); else if (U == 2) LAUNCH(2, 5); else LAUNCH(4, 5); }
    else { if (U == 1) LAUNCH(1, 4); else if (U == 2) LAUNCH(2, 4); else LAUNCH(4, 4); }
    return (int)hipGetLastError();
}
