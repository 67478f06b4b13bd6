// Measurement aid (not product code): what the memory system does with the stream kernel's LINE MIX on C3 when there is
// no consensus work at all — per 64-query task 12.5 KiB of bit-scores streamed, 64 side-record pieces gathered out of the
// task's own 51 KB of the 16-byte side-record array (one or two lines each), 64 random 128-byte lineage rows out of a
// 307 MB table, 2 KiB of records written — under three dependency shapes:
//   mode 0  every load of a task independent of every other (the ceiling of the traffic itself at this occupancy)
//   mode 1  the real chain: side-record addresses come out of the bit-scores, row addresses out of the side records —
//           three round trips one after the other per task, nothing overlapping inside a wave
//   mode 2  the same chain with TWO tasks in flight per wave (task i+1's bit-scores and side records travel while task i
//           waits for its rows): what a software-pipelined wave would see
// DESIGN.md §8 quotes the three next to the product kernel's time on the same box.
#include <hip/hip_runtime.h>
#include <cstdint>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

struct MixArgs {
    const u32x4* bits;      // n_q * 50 * 4 bytes
    const u32x4* side;      // n_q * 50 * 16 bytes
    const u32x4* rows;      // n_rows * 128 bytes
    u32x4* out;             // n_q * 32 bytes
    uint64_t n_q;
    uint32_t n_rows;
    uint32_t mode;
    uint32_t* sink;
};

__device__ __forceinline__ uint32_t mix32(uint32_t x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

struct TaskRegs { u32x4 a[13]; u32x4 s0, s1; u32x4 r[8]; };

template <int BLOCK>
__global__ __launch_bounds__(BLOCK) void mix_kernel(MixArgs p) {
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t wave = (uint64_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6), n_waves = (uint64_t)gridDim.x * (BLOCK / 64);
    const uint64_t n_tasks = p.n_q / 64;
    uint32_t acc = 0;
    auto issue_bits = [&](TaskRegs& T, uint64_t task) {
        const u32x4* b = p.bits + task * 800;                 // 64 queries x 50 rows x 4 B = 12800 B = 800 x 16 B
#pragma unroll
        for (int k = 0; k < 13; ++k) { const uint32_t i = (uint32_t)k * 64u + lane; T.a[k] = i < 800u ? __builtin_nontemporal_load(b + i) : u32x4{0, 0, 0, 0}; }
    };
    auto bits_word = [&](const TaskRegs& T) { uint32_t x = 0;
#pragma unroll
        for (int k = 0; k < 13; ++k) x ^= T.a[k].x ^ T.a[k].y ^ T.a[k].z ^ T.a[k].w;
        return x; };
    auto issue_side = [&](TaskRegs& T, uint64_t task, uint32_t dep) {
        const uint32_t w = mix32((uint32_t)task * 64u + lane + (dep & 1u)) % 47u;       // the top group: rows w .. w + 2 of the lane's query
        const u32x4* s = p.side + (task * 64 + lane) * 50 + w;
        T.s0 = __builtin_nontemporal_load(s); T.s1 = __builtin_nontemporal_load(s + 2);
    };
    auto issue_rows = [&](TaskRegs& T, uint64_t task, uint32_t dep) {
        const uint32_t r = mix32((uint32_t)task * 64u + lane + 0x9E3779B9u + (dep & 1u)) % p.n_rows;
        const u32x4* q = p.rows + (uint64_t)r * 8;
#pragma unroll
        for (int k = 0; k < 8; ++k) T.r[k] = q[k];
    };
    auto finish = [&](const TaskRegs& T, uint64_t task) {
        uint32_t x = T.s0.x ^ T.s1.y;
#pragma unroll
        for (int k = 0; k < 8; ++k) x ^= T.r[k].x ^ T.r[k].w;
        acc ^= x;
        u32x4* o = p.out + task * 128;                        // 64 records x 32 B = 128 x 16 B, two coalesced 1 KiB stores
        const u32x4 v = {x, lane, (uint32_t)task, 7u};
        __builtin_nontemporal_store(v, o + lane); __builtin_nontemporal_store(v, o + 64 + lane);
    };
    if (p.mode == 0) {
        for (uint64_t task = wave; task < n_tasks; task += n_waves) {
            TaskRegs T;
            issue_bits(T, task); issue_side(T, task, 0); issue_rows(T, task, 0);
            acc ^= bits_word(T);
            finish(T, task);
        }
    } else if (p.mode == 1) {
        for (uint64_t task = wave; task < n_tasks; task += n_waves) {
            TaskRegs T;
            issue_bits(T, task);
            const uint32_t d1 = bits_word(T);                  // (waits for the bit-scores)
            issue_side(T, task, d1 == 0x12345678u);
            const uint32_t d2 = T.s0.x ^ T.s1.x;               // (waits for the side records)
            issue_rows(T, task, d2 == 0x12345678u);
            acc ^= d1;
            finish(T, task);
        }
    } else {
        // two tasks in flight: [bits(i+1)] issued before the wait for side(i); [side(i+1)] before the wait for rows(i)
        TaskRegs A, B;
        uint64_t task = wave;
        if (task < n_tasks) issue_bits(A, task);
        while (task < n_tasks) {
            const uint64_t nxt = task + n_waves;
            const uint32_t d1 = bits_word(A);
            issue_side(A, task, d1 == 0x12345678u);
            if (nxt < n_tasks) issue_bits(B, nxt);             // travels while this task's side records and rows do
            const uint32_t d2 = A.s0.x ^ A.s1.x;
            issue_rows(A, task, d2 == 0x12345678u);
            acc ^= d1;
            finish(A, task);
            // swap roles (explicit copies of the bit-score registers: the compiler renames)
#pragma unroll
            for (int k = 0; k < 13; ++k) A.a[k] = B.a[k];
            task = nxt;
        }
    }
    if (acc == 0x12345678u) *p.sink = acc;
}

extern "C" int mix_run(const void* bits, const void* side, const void* rows, void* out, uint64_t n_q, uint32_t n_rows, uint32_t mode, void* sink,
                       int waves_per_cu, int cus, void* stream) {
    MixArgs a{(const u32x4*)bits, (const u32x4*)side, (const u32x4*)rows, (u32x4*)out, n_q, n_rows, mode, (uint32_t*)sink};
    hipStream_t s = (hipStream_t)stream;
    if (waves_per_cu == 8) hipLaunchKernelGGL(mix_kernel<512>, dim3(cus), dim3(512), 0, s, a);
    else if (waves_per_cu == 12) hipLaunchKernelGGL(mix_kernel<768>, dim3(cus), dim3(768), 0, s, a);
    else if (waves_per_cu == 16) hipLaunchKernelGGL(mix_kernel<1024>, dim3(cus), dim3(1024), 0, s, a);
    else return -1;
    return (int)hipGetLastError();
}
