// Experiment (round 3): do the 75 plane streams of the forward read faster through
// LDS-DMA (global_load_lds_dwordx4: 16 B per lane, no VGPR destination) than through the product's
// dword-per-lane loads?  Stream-only stand-ins in the product's shape: 128 x 16 tile, 8 waves as
// 2 across x 4 down, a wave owns 64 columns of rows wy, wy+4, wy+8, wy+12, taps in the outer loop,
// one workgroup per CU (dynamic LDS pad), trivial arithmetic (sums), 3 output planes written.
//   variant 0      dword loads (nt), one tap ahead in registers: what synth_fwd_tapmajor does
//   variant 1..3   LDS-DMA ring per wave, 1 / 2 / 3 taps ahead: per tap and plane ONE wave
//                  instruction fetches the wave's 4 rows x 256 B (lane l: row l/16, 16-byte chunk
//                  l%16) into a 1 KB slot; the lanes read their pixels back with ds_read_b32
//   variant 11..13 the same with the nt cache policy (aux = 2)
// A wave reads only what it fetched itself, so a counted vmcnt orders the reads (no barrier).
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((address_space(3))) void lds_void;
typedef __attribute__((address_space(1))) const void gbl_void;

template <int N>
__device__ __forceinline__ void wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int V>
struct Tag {
    static constexpr int value = V;
};
template <int B, int E, class F>
__device__ __forceinline__ void sfor(F &&f)
{
    if constexpr (B < E) {
        f(Tag<B>{});
        sfor<B + 1, E>(f);
    }
}

__global__ __launch_bounds__(512, 1) void stream_dword(const float *__restrict__ w, const float *__restrict__ oy,
                                                        const float *__restrict__ ox, float *__restrict__ out,
                                                        int H, int W)
{
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int x = blockIdx.x * 128 + (wv & 1) * 64 + threadIdx.x;
    const int ty0 = blockIdx.y * 16 + (wv >> 1);
    const size_t plane = (size_t)H * W;
    size_t ro[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ro[j] = (size_t)min(ty0 + 4 * j, H - 1) * W;
    float cur[12], nxt[12], a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
    auto issue = [&](int t, float *d) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const size_t o = (size_t)t * plane + ro[j] + x;
            d[3 * j] = __builtin_nontemporal_load(w + o);
            d[3 * j + 1] = __builtin_nontemporal_load(oy + o);
            d[3 * j + 2] = __builtin_nontemporal_load(ox + o);
        }
    };
    issue(0, cur);
    sfor<0, 25>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        if (t < 24) issue(t + 1, nxt);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a0[j] += cur[3 * j], a1[j] += cur[3 * j + 1], a2[j] += cur[3 * j + 2];
            asm volatile("" : "+v"(a0[j]), "+v"(a1[j]), "+v"(a2[j]));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 12; ++i) cur[i] = nxt[i];
    });
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (ty0 + 4 * j >= H) continue;
        float *o = out + ro[j] + x;
        __builtin_nontemporal_store(a0[j], o);
        __builtin_nontemporal_store(a1[j], o + plane);
        __builtin_nontemporal_store(a2[j], o + 2 * plane);
    }
}

template <int D, int AUX>
__global__ __launch_bounds__(512, 1) void stream_glds(const float *__restrict__ w, const float *__restrict__ oy,
                                                       const float *__restrict__ ox, float *__restrict__ out,
                                                       int H, int W)
{
    // per wave: (D + 1) slots x 3 planes x 4 rows x 64 pixels
    __shared__ __attribute__((aligned(16))) float ring[8 * (D + 1) * 3 * 256];
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int l = threadIdx.x;
    const int x0 = blockIdx.x * 128 + (wv & 1) * 64;
    const int ty0 = blockIdx.y * 16 + (wv >> 1);
    const size_t plane = (size_t)H * W;
    size_t ro[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) ro[j] = (size_t)min(ty0 + 4 * j, H - 1) * W;
    // this lane's source for a wave instruction: row l / 16, 16-byte chunk l % 16 of the wave's 64 columns
    const size_t src = (size_t)min(ty0 + 4 * (l >> 4), H - 1) * W + x0 + 4 * (l & 15);
    float *mine = ring + wv * ((D + 1) * 3 * 256);
    auto issue = [&](int t, int slot) {
        const size_t o = (size_t)t * plane + src;
        lds_void *d = (lds_void *)(mine + slot * 768);
        __builtin_amdgcn_global_load_lds((gbl_void *)(w + o), d, 16, 0, AUX);
        __builtin_amdgcn_global_load_lds((gbl_void *)(oy + o), (lds_void *)((__attribute__((address_space(3))) float *)d + 256), 16, 0, AUX);
        __builtin_amdgcn_global_load_lds((gbl_void *)(ox + o), (lds_void *)((__attribute__((address_space(3))) float *)d + 512), 16, 0, AUX);
    };
    float a0[4] = {0, 0, 0, 0}, a1[4] = {0, 0, 0, 0}, a2[4] = {0, 0, 0, 0};
    sfor<0, D>([&](auto tc) { issue(decltype(tc)::value, decltype(tc)::value); });
    sfor<0, 25>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        if constexpr (t + D < 25) issue(t + D, (t + D) % (D + 1));
        constexpr int ahead = (24 - t) < D ? (24 - t) : D;  // taps still in flight behind tap t
        wait_vm<3 * ahead>();
        const float *s = mine + (t % (D + 1)) * 768 + l;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a0[j] += s[j * 64], a1[j] += s[256 + j * 64], a2[j] += s[512 + j * 64];
            asm volatile("" : "+v"(a0[j]), "+v"(a1[j]), "+v"(a2[j]));  // (the reads are consumed before
        }                                                              // their slot is fetched into again)
        __builtin_amdgcn_sched_barrier(0);
    });
    const int x = x0 + l;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (ty0 + 4 * j >= H) continue;
        float *o = out + ro[j] + x;
        __builtin_nontemporal_store(a0[j], o);
        __builtin_nontemporal_store(a1[j], o + plane);
        __builtin_nontemporal_store(a2[j], o + 2 * plane);
    }
}

template <int D, int AUX>
static void launch_glds(dim3 grid, unsigned pad, hipStream_t st, const float *w, const float *oy, const float *ox,
                        float *out, int H, int W)
{
    // static ring: 8 waves x (D + 1) slots x 3 KB; pad only up to "a second workgroup does not fit"
    const unsigned ring = 8u * (D + 1) * 3072u;
    pad = ring >= pad ? 0u : pad - ring;
    hipLaunchKernelGGL((stream_glds<D, AUX>), grid, dim3(64, 8), pad, st, w, oy, ox, out, H, W);
}

extern "C" int exp_run(int variant, const float *inp, const float *w, const float *oy, const float *ox, float *out,
                       int N, int H, int W, void *stream)
{
    (void)inp;
    (void)N;
    if (W % 128) return 2;  // (the stand-ins have no right-edge handling)
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(W / 128, (H + 15) / 16);
    const unsigned pad = 82 * 1024;  // a second workgroup does not fit the CU's 160 KB
    switch (variant) {
    case 0: hipLaunchKernelGGL(stream_dword, grid, dim3(64, 8), pad, st, w, oy, ox, out, H, W); break;
    case 1: launch_glds<1, 0>(grid, pad, st, w, oy, ox, out, H, W); break;
    case 2: launch_glds<2, 0>(grid, pad, st, w, oy, ox, out, H, W); break;
    case 3: launch_glds<3, 0>(grid, pad, st, w, oy, ox, out, H, W); break;
    case 11: launch_glds<1, 2>(grid, pad, st, w, oy, ox, out, H, W); break;
    case 12: launch_glds<2, 2>(grid, pad, st, w, oy, ox, out, H, W); break;
    case 13: launch_glds<3, 2>(grid, pad, st, w, oy, ox, out, H, W); break;
    default: return 1;
    }
    return (int)hipGetLastError();
}
