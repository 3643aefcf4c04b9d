// Experiment: LDS atomic-add rate on gfx950 by operand width and address pattern, the ceiling of
// the gradInput scatter (synth_gin_tiled accumulates 300 contributions per pixel in LDS).
// Standalone: hipcc --offload-arch=gfx950 -O3 tools/exp/exp_ldsatomic.hip -o /tmp/ldsat && /tmp/ldsat
// One workgroup per CU (WAVES waves), each wave issues ITERS x 12 atomics; reports cycles per
// wave-instruction per CU (wall time x clock / instructions per CU).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// MODE 0: u32 atomics, 1: u64 atomics, 2: u64 carrying two 32-bit fields at adjacent columns
// (even/odd column pairing: half as many instructions per tap), 3: f32 atomics
// JIT: 0 = lane l hits column l (conflict-free), S > 0 = column l + (hash % (2S+1)) - S and row
// jitter likewise (the sigma ~ S regime)
template <int MODE, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void ldsat(int iters, int jit, uint32_t *sink)
{
    constexpr int LW = 101, LH = 53;          // the R = 16 window of a 64 x 16 tile
    __shared__ unsigned long long acc[3 * LW * LH];   // 128.5 KB
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 3 * LW * LH; i += 64 * WAVES) acc[i] = 0;
    __syncthreads();
    uint32_t h = hash32(blockIdx.x * 1024 + threadIdx.x);
    for (int it = 0; it < iters; ++it) {
        h = hash32(h + it);
        const int dx = jit ? (int)(h % (2 * jit + 1)) - jit : 0;
        const int dy = jit ? (int)((h >> 8) % (2 * jit + 1)) - jit : 0;
        const int row = 16 + (wv * 2 + (it & 1)) + dy, col = 16 + lane + dx;
        const int base = row * LW + col;
        const uint32_t v = h >> 4;
        if (MODE == 0) {
            uint32_t *a32 = (uint32_t *)acc;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                atomicAdd(a32 + c * LW * LH + base, v);
                atomicAdd(a32 + c * LW * LH + base + 1, v);
                atomicAdd(a32 + c * LW * LH + base + LW, v);
                atomicAdd(a32 + c * LW * LH + base + LW + 1, v);
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                atomicAdd(acc + c * LW * LH + base, (unsigned long long)v);
                atomicAdd(acc + c * LW * LH + base + 1, (unsigned long long)v);
                atomicAdd(acc + c * LW * LH + base + LW, (unsigned long long)v);
                atomicAdd(acc + c * LW * LH + base + LW + 1, (unsigned long long)v);
            }
        } else if (MODE == 2) {
            // two images (even / odd start column), each 64-bit slot = columns (2k, 2k+1) or (2k+1, 2k+2)
            const int pair = (col >> 1) + (col & 1) * (LW * LH / 2);
            const int b2 = row * (LW / 2 + 1) + pair;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                atomicAdd(acc + c * LW * LH + b2, ((unsigned long long)v << 32) + v);
                atomicAdd(acc + c * LW * LH + b2 + (LW / 2 + 1), ((unsigned long long)v << 32) + v);
            }
        } else {
            float *af = (float *)acc;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                atomicAdd(af + c * LW * LH + base, (float)v);
                atomicAdd(af + c * LW * LH + base + 1, (float)v);
                atomicAdd(af + c * LW * LH + base + LW, (float)v);
                atomicAdd(af + c * LW * LH + base + LW + 1, (float)v);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = (uint32_t)acc[17 * LW + 20];
}

template <int MODE, int WAVES>
void run(const char *name, int jit, uint32_t *sink)
{
    const int iters = 4000, blocks = 256;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL((ldsat<MODE, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, 10, jit, sink);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0));
    hipLaunchKernelGGL((ldsat<MODE, WAVES>), dim3(blocks), dim3(64 * WAVES), 0, 0, iters, jit, sink);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double per_tap = MODE == 2 ? 6.0 : 12.0;
    const double instr_per_cu = (double)iters * per_tap * WAVES;
    printf("%-34s waves %d jitter %2d: %8.3f ms  %6.1f ns per tap-row per CU  %5.1f cycles/instr @2.4GHz\n", name,
           WAVES, jit, ms, ms * 1e6 / (iters * WAVES), ms * 1e-3 * 2.4e9 / instr_per_cu);
}

int main()
{
    uint32_t *sink;
    CHECK(hipMalloc(&sink, 4096));
    for (int jit : {0, 1, 2, 8}) {
        run<0, 4>("ds_add_u32 x12", jit, sink);
        run<0, 8>("ds_add_u32 x12", jit, sink);
        run<1, 4>("ds_add_u64 x12", jit, sink);
        run<1, 8>("ds_add_u64 x12", jit, sink);
        run<2, 8>("ds_add_u64 x6 (column pairs)", jit, sink);
        run<3, 8>("ds_add_f32 x12", jit, sink);
    }
    return 0;
}
