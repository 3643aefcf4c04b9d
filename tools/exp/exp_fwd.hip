// Experiment kernels (developer tool, not product): ablations of the LDS-staged forward to find
// what bounds it.  Built by tools/exp/run_exp.py into artemis_amd/lib/exp/ (git-ignored).
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;

struct Geom { int N, C, H, W, k, d, Hin, Win, oy0, ox0; };

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }
__device__ __forceinline__ int wadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }

enum { F_NO_LDS = 1, F_STREAM_ONLY = 2, F_NO_FILL = 4, F_NO_STORE = 8 };

struct TapRow { float w[5], a[5], b[5]; };

__device__ __forceinline__ void load_tap_row(TapRow &t, const float *__restrict__ wrow,
                                             const float *__restrict__ arow,
                                             const float *__restrict__ brow, size_t plane, int r,
                                             unsigned x)
{
#pragma unroll
    for (int s = 0; s < 5; ++s) {
        const size_t o = (size_t)(r * 5 + s) * plane;
        t.w[s] = (wrow + o)[x];
        t.a[s] = (arow + o)[x];
        t.b[s] = (brow + o)[x];
    }
}

template <int R, int TH, int WAVES, int FLAGS>
__global__ __launch_bounds__(WAVES * 64) void exp_fwd(const float *__restrict__ in,
                                                       const float *__restrict__ w,
                                                       const float *__restrict__ oy,
                                                       const float *__restrict__ ox,
                                                       float *__restrict__ out, Geom g)
{
    constexpr int LW = 64 + 5 + 2 * R, LH = TH + 5 + 2 * R;
    __shared__ f32x4 tile[LH * LW];
    const lds_f32x4 *ltile = (const lds_f32x4 *)tile;
    const int tx0 = blockIdx.x * 64, ty0 = blockIdx.y * TH, n = blockIdx.z;
    const int wy = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int tid = wy * 64 + threadIdx.x;
    const size_t plane = (size_t)g.H * g.W, iplane = (size_t)g.Hin * g.Win;
    const float *ip = in + (size_t)n * 3 * iplane;
    const int px = threadIdx.x, x = tx0 + px;
    const bool active = x < g.W && ty0 + wy < g.H;
    const size_t nbase = (size_t)n * 25 * plane;

    TapRow cur;
    size_t rowoff = nbase + (size_t)(ty0 + wy) * g.W;
    if (active) load_tap_row(cur, w + rowoff, oy + rowoff, ox + rowoff, plane, 0, (unsigned)x);

    if (!(FLAGS & (F_NO_FILL | F_STREAM_ONLY | F_NO_LDS))) {
        const int gy0 = ty0 - R, gx0 = tx0 - R;
        for (int i = tid; i < LH * LW; i += WAVES * 64) {
            const int ly = i / LW, lx = i - ly * LW;
            const int gy = clampi(gy0 + ly, 0, g.Hin - 1), gx = clampi(gx0 + lx, 0, g.Win - 1);
            const int o = gy * g.Win + gx;
            tile[i] = f32x4{ip[o], ip[iplane + o], ip[2 * iplane + o], 0.f};
        }
        __syncthreads();
    }
    if (!active) return;

#pragma unroll 1
    for (int py = wy; py < TH; py += WAVES) {
        const int y = ty0 + py;
        const bool more = py + WAVES < TH && y + WAVES < g.H;
        const float *wrow = w + rowoff, *arow = oy + rowoff, *brow = ox + rowoff;
        float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f;
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            TapRow nxt;
            if (r < 4) {
                load_tap_row(nxt, wrow, arow, brow, plane, r + 1, (unsigned)x);
            } else if (more) {
                const size_t o = (size_t)WAVES * g.W;
                load_tap_row(nxt, wrow + o, arow + o, brow + o, plane, 0, (unsigned)x);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                const float wv = cur.w[s], a = cur.a[s], b = cur.b[s];
                if (FLAGS & F_STREAM_ONLY) {
                    acc0 += wv; acc1 += a; acc2 += b;
                    continue;
                }
                const int ia = (int)a, ib = (int)b;
                const float fa = a - (float)ia, fb = b - (float)ib;
                const int ly = wadd(py + r + R, ia), lx = wadd(px + s + R, ib);
                const unsigned idx = min(__umul24((unsigned)ly, (unsigned)LW) + (unsigned)lx,
                                         (unsigned)((LH - 1) * LW - 2));
                f32x4 bl, br, tl, tr;
                if (FLAGS & F_NO_LDS) {
                    const float v = __uint_as_float(idx);
                    bl = f32x4{v, fa, fb, 0}; br = f32x4{fa, v, fb, 0};
                    tl = f32x4{fb, fa, v, 0}; tr = f32x4{v, v, fa, 0};
                } else {
                    const lds_f32x4 *p = ltile + idx;
                    bl = p[0]; br = p[1]; tl = p[LW]; tr = p[LW + 1];
                }
                const float ga = 1.f - fa, gb = 1.f - fb;
                const float c00 = ga * gb, c10 = fa * gb, c01 = ga * fb, c11 = fa * fb;
                acc0 += wv * (bl.x * c00 + tl.x * c10 + br.x * c01 + tr.x * c11);
                acc1 += wv * (bl.y * c00 + tl.y * c10 + br.y * c01 + tr.y * c11);
                acc2 += wv * (bl.z * c00 + tl.z * c10 + br.z * c01 + tr.z * c11);
            }
            asm volatile("" : "+v"(acc0), "+v"(acc1), "+v"(acc2));
            __builtin_amdgcn_sched_barrier(0);
            cur = nxt;
        }
        float *orow = out + (size_t)n * 3 * plane + (size_t)y * g.W;
        orow[(unsigned)x] = acc0;
        (orow + plane)[(unsigned)x] = acc1;
        (orow + 2 * plane)[(unsigned)x] = acc2;
        if (!more) break;
        rowoff += (size_t)WAVES * g.W;
    }
}

// float4 copy / read calibration kernels
__global__ __launch_bounds__(256) void copy_f4(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void read_f4(const f32x4 *__restrict__ src, float *__restrict__ dst, size_t n)
{
    f32x4 a = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a += src[i];
    if (a.x + a.y + a.z + a.w == 1.2345f) dst[0] = a.x;
}
__global__ __launch_bounds__(256) void read_f1(const float *__restrict__ src, float *__restrict__ dst, size_t n)
{
    float a = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a += src[i];
    if (a == 1.2345f) dst[0] = a;
}

template <int R, int TH, int WAVES, int FLAGS>
static void launch(const float *in, const float *w, const float *oy, const float *ox, float *out, Geom g, hipStream_t st)
{
    dim3 grid((g.W + 63) / 64, (g.H + TH - 1) / TH, g.N);
    hipLaunchKernelGGL((exp_fwd<R, TH, WAVES, FLAGS>), grid, dim3(64, WAVES), 0, st, in, w, oy, ox, out, g);
}

extern "C" int exp_run(int variant, const float *in, const float *w, const float *oy, const float *ox,
                       float *out, int N, int H, int W, void *stream)
{
    Geom g{N, 3, H, W, 5, 1, H + 4, W + 4, 0, 0};
    hipStream_t st = (hipStream_t)stream;
    switch (variant) {
    case 0: launch<4, 16, 4, 0>(in, w, oy, ox, out, g, st); break;
    case 1: launch<4, 16, 4, F_NO_LDS>(in, w, oy, ox, out, g, st); break;
    case 2: launch<4, 16, 4, F_STREAM_ONLY>(in, w, oy, ox, out, g, st); break;
    case 3: launch<4, 16, 4, F_NO_FILL>(in, w, oy, ox, out, g, st); break;
    case 4: launch<4, 8, 4, 0>(in, w, oy, ox, out, g, st); break;
    case 5: launch<4, 32, 4, 0>(in, w, oy, ox, out, g, st); break;
    case 6: launch<4, 32, 8, 0>(in, w, oy, ox, out, g, st); break;
    case 7: launch<4, 8, 4, F_STREAM_ONLY>(in, w, oy, ox, out, g, st); break;
    case 8: launch<4, 32, 4, F_STREAM_ONLY>(in, w, oy, ox, out, g, st); break;
    case 9: launch<2, 16, 4, 0>(in, w, oy, ox, out, g, st); break;
    case 10: launch<4, 64, 8, 0>(in, w, oy, ox, out, g, st); break;
    case 11: launch<4, 64, 4, F_STREAM_ONLY>(in, w, oy, ox, out, g, st); break;
    case 100: hipLaunchKernelGGL(copy_f4, dim3(2048), dim3(256), 0, st, (const f32x4 *)w, (f32x4 *)oy, (size_t)N * 25 * H * W / 4); break;
    case 101: hipLaunchKernelGGL(read_f4, dim3(2048), dim3(256), 0, st, (const f32x4 *)w, out, (size_t)N * 25 * H * W / 4); break;
    case 102: hipLaunchKernelGGL(read_f1, dim3(2048), dim3(256), 0, st, w, out, (size_t)N * 25 * H * W); break;
    default: return -1;
    }
    return (int)hipGetLastError();
}
