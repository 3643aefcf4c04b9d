// Experiment: what read / copy bandwidth does this box sustain for simple linear streams, as a
// function of size and load shape?  (Ceiling calibration for the synth kernels.)
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// each block reads a contiguous chunk; each thread UNROLL independent 16-B loads per iteration
template <int UNROLL>
__global__ __launch_bounds__(256) void read_lin4(const f32x4 *__restrict__ src, float *__restrict__ dst, size_t n)
{
    f32x4 a[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) a[u] = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        f32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) a[u] += v[u];
    }
    for (; i < n; i += stride) a[0] += src[i];
    f32x4 s = 0;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) s += a[u];
    if (s.x + s.y + s.z + s.w == 1.2345f) dst[0] = s.x;
}
template <int UNROLL>
__global__ __launch_bounds__(256) void read_lin1(const float *__restrict__ src, float *__restrict__ dst, size_t n)
{
    float a[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) a[u] = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        float v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) a[u] += v[u];
    }
    for (; i < n; i += stride) a[0] += src[i];
    float s = 0;
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) s += a[u];
    if (s == 1.2345f) dst[0] = s;
}
template <int UNROLL>
__global__ __launch_bounds__(256) void copy_lin4(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        f32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) dst[i + u * stride] = v[u];
    }
    for (; i < n; i += stride) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void write_lin4(f32x4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = f32x4{1, 2, 3, 4};
}


// nt variants and block-contiguous chunking
template <int UNROLL, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy_nt(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        f32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NTL ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { if (NTS) __builtin_nontemporal_store(v[u], dst + i + u * stride); else dst[i + u * stride] = v[u]; }
    }
}
template <int UNROLL, bool NTL, bool NTS>
__global__ __launch_bounds__(256) void copy1_nt(const float *__restrict__ src, float *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        float v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = NTL ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) { if (NTS) __builtin_nontemporal_store(v[u], dst + i + u * stride); else dst[i + u * stride] = v[u]; }
    }
}
template <bool NT>
__global__ __launch_bounds__(256) void write1(float *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) { if (NT) __builtin_nontemporal_store(1.5f, dst + i); else dst[i] = 1.5f; }
}

// each block copies ONE contiguous chunk (n / gridDim.x elements), 8 independent dword loads per thread
// in flight: the shape of a tile-owning workgroup, as opposed to the grid-stride kernels above
template <bool NT>
__global__ __launch_bounds__(256) void copy1_chunk(const float *__restrict__ src, float *__restrict__ dst, size_t n)
{
    const size_t chunk = (n / gridDim.x) / 2048 * 2048;  // multiple of 8 x 256 elements
    const size_t base = (size_t)blockIdx.x * chunk;
    for (size_t i = base + threadIdx.x; i < base + chunk; i += 2048) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = NT ? __builtin_nontemporal_load(src + i + u * 256) : src[i + u * 256];
#pragma unroll
        for (int u = 0; u < 8; ++u) { if (NT) __builtin_nontemporal_store(v[u], dst + i + u * 256); else dst[i + u * 256] = v[u]; }
    }
}

// variant encodes kernel; nbytes = size of the stream
extern "C" int bw_run(int variant, const void *src, void *dst, size_t nbytes, int blocks, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    const size_t n4 = nbytes / 16, n1 = nbytes / 4;
    switch (variant) {
    case 0: hipLaunchKernelGGL(read_lin4<4>, dim3(blocks), dim3(256), 0, st, (const f32x4 *)src, (float *)dst, n4); break;
    case 1: hipLaunchKernelGGL(read_lin4<8>, dim3(blocks), dim3(256), 0, st, (const f32x4 *)src, (float *)dst, n4); break;
    case 2: hipLaunchKernelGGL(read_lin1<8>, dim3(blocks), dim3(256), 0, st, (const float *)src, (float *)dst, n1); break;
    case 3: hipLaunchKernelGGL(read_lin1<16>, dim3(blocks), dim3(256), 0, st, (const float *)src, (float *)dst, n1); break;
    case 4: hipLaunchKernelGGL(copy_lin4<4>, dim3(blocks), dim3(256), 0, st, (const f32x4 *)src, (f32x4 *)dst, n4); break;
    case 5: hipLaunchKernelGGL(copy_lin4<8>, dim3(blocks), dim3(256), 0, st, (const f32x4 *)src, (f32x4 *)dst, n4); break;
    case 6: hipLaunchKernelGGL(write_lin4, dim3(blocks), dim3(256), 0, st, (f32x4 *)dst, n4); break;
    case 7: hipLaunchKernelGGL((copy_nt<4, false, true>), dim3(blocks), dim3(256), 0, st, (const f32x4 *)src, (f32x4 *)dst, n4); break;
    case 8: hipLaunchKernelGGL((copy_nt<4, true, true>), dim3(blocks), dim3(256), 0, st, (const f32x4 *)src, (f32x4 *)dst, n4); break;
    case 9: hipLaunchKernelGGL((copy1_nt<8, false, false>), dim3(blocks), dim3(256), 0, st, (const float *)src, (float *)dst, n1); break;
    case 10: hipLaunchKernelGGL((copy1_nt<8, false, true>), dim3(blocks), dim3(256), 0, st, (const float *)src, (float *)dst, n1); break;
    case 11: hipLaunchKernelGGL((copy1_nt<8, true, true>), dim3(blocks), dim3(256), 0, st, (const float *)src, (float *)dst, n1); break;
    case 12: hipLaunchKernelGGL((write1<false>), dim3(blocks), dim3(256), 0, st, (float *)dst, n1); break;
    case 13: hipLaunchKernelGGL((write1<true>), dim3(blocks), dim3(256), 0, st, (float *)dst, n1); break;
    case 14: hipLaunchKernelGGL((copy1_chunk<false>), dim3(blocks), dim3(256), 0, st, (const float *)src, (float *)dst, n1); break;
    case 15: hipLaunchKernelGGL((copy1_chunk<true>), dim3(blocks), dim3(256), 0, st, (const float *)src, (float *)dst, n1); break;
    default: return -1;
    }
    return (int)hipGetLastError();
}
