// Experiment: HBM read bandwidth for randomly placed contiguous chunks of G bytes per wave visit
// (how much contiguity per "plane visit" does DRAM need?).  Also write and read+write forms.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t hash32(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// Each wave: ITERS visits; visit = G bytes contiguous at a random G-aligned chunk; dword/lane loads
// (256 B per wave instruction), G/256 instructions per visit.
template <int GDW /* G in units of 256 B */, int MODE /*0 read,1 write,2 copy*/>
__global__ __launch_bounds__(256) void gran(const float *__restrict__ src, float *__restrict__ dst,
                                             uint32_t nchunks, int iters)
{
    const uint32_t wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const uint32_t lane = threadIdx.x & 63;
    float acc = 0;
    for (int it = 0; it < iters; ++it) {
        const uint32_t c = hash32(wave * 7919u + it * 104729u + 12345u) % nchunks;
        const size_t base = (size_t)c * GDW * 64 + lane;
        float v[GDW < 16 ? GDW : 16];
#pragma unroll 1
        for (int g0 = 0; g0 < GDW; g0 += 16) {
#pragma unroll
            for (int g = 0; g < (GDW < 16 ? GDW : 16); ++g) {
                if (MODE != 1) v[g] = __builtin_nontemporal_load(src + base + (size_t)(g0 + g) * 64);
                else v[g] = (float)it;
            }
#pragma unroll
            for (int g = 0; g < (GDW < 16 ? GDW : 16); ++g) {
                if (MODE == 0) acc += v[g];
                else __builtin_nontemporal_store(v[g] + 1.f, dst + base + (size_t)(g0 + g) * 64);
            }
        }
    }
    if (MODE == 0 && acc == 1.2345f) dst[0] = acc;
}

extern "C" int gran_run(int gdw, int mode, const float *src, float *dst, size_t nbytes, int blocks, int iters, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    const uint32_t nchunks = (uint32_t)(nbytes / ((size_t)gdw * 256));
#define CASE(G) case G: \
    if (mode == 0) hipLaunchKernelGGL((gran<G, 0>), dim3(blocks), dim3(256), 0, st, src, dst, nchunks, iters); \
    else if (mode == 1) hipLaunchKernelGGL((gran<G, 1>), dim3(blocks), dim3(256), 0, st, src, dst, nchunks, iters); \
    else hipLaunchKernelGGL((gran<G, 2>), dim3(blocks), dim3(256), 0, st, src, dst, nchunks, iters); break;
    switch (gdw) {
    CASE(1) CASE(2) CASE(4) CASE(8) CASE(16) CASE(32) CASE(64) CASE(128) CASE(256)
    default: return -1;
    }
    return (int)hipGetLastError();
}
