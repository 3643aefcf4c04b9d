// synth_device.hpp -- device code of the ArTEMIS frame-synthesis op for MI355X (gfx950, CDNA4).
//
// What it computes (reference: /root/reference/cupy_module/synth.py, kernel text :6-201):
//   out[n,c,y,x] = sum_{r,s<k} w[n,rk+s,y,x] * B(in[n,c], y+r*d+trunc(a), x+s*d+trunc(b),
//                                                a-trunc(a), b-trunc(b))
//   a = offset_y[n,rk+s,y,x], b = offset_x[...], trunc = C (int) cast, B = 2x2 blend with the
//   four indices clamped to the padded frame.  Backward: gradWeight, gradOffset_y,
//   gradOffset_x as in the reference's three kernels, fused; gradInput (absent upstream) as the
//   exact adjoint.
//
// Design (DESIGN.md has the numbers): the op is an HBM stream of the 3 x 25 per-pixel planes
// (300 of 324 B/pixel forward, 600 of 624 B/pixel backward) with a 4-corner x 3-channel gather
// from a 12 B/pixel frame.  One lane owns one output pixel (64 lanes = 64 consecutive x, so every
// plane access of a wave is one coalesced 256 B segment) and keeps all channels in registers,
// so the planes are read once (the reference re-reads them per channel).  The frame tile of a
// 64 x TH pixel workgroup, extended by the tap footprint and a displacement radius R, is staged
// once into LDS as float4 {c0,c1,c2,-} texels with the border clamp already applied, so the
// gathers are ds_read_b128 from one address register with immediate offsets and never touch
// the vector memory pipe.  Taps displaced beyond R fall back per lane to clamped global loads.
// No MFMA: there is no dense contraction here.
#ifndef ARTEMIS_SYNTH_DEVICE_HPP
#define ARTEMIS_SYNTH_DEVICE_HPP
// Included by synth_kernels.hip (C ABI, host logic, direct kernels) and by synth_tiles.hip, which
// is compiled once per LDS tile radius (-DSYNTH_TILE_R=2|4|8|14) so the heavy unrolled kernels
// build in parallel.  Everything but the few types and launcher declarations that cross
// translation units lives in an anonymous namespace.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#include "artemis_synth.h"

namespace asynth {

// Problem geometry shared by all kernels. Frame row index of tap (r) at output row y with
// truncated offset ia is  y + oy0 + r*d + ia  (oy0 = ox0 = 0 for the reference layout).
struct Geom {
    int N, C, H, W;
    int k, d;
    int Hin, Win;
    int oy0, ox0;
};

constexpr int kMaxFrames = 8;

// Frame element types: fp32 (the reference's) or bf16 bit patterns (BASELINE configs[4]: "bf16
// frames / fp32 accumulate").  A bf16 texel is widened exactly when it is staged or gathered; all
// arithmetic stays fp32.
typedef unsigned short bf16_t;

// arguments of the fused ChronoSynth head kernels (ignored by the single-frame instantiations)
struct HeadArgs {
    const void *frame[kMaxFrames];   // T frames, each (N, 3, Hin, Win), fp32 or bf16
    const float *occ;                // (N, T, H, W)
    float *gocc;                     // backward only: (N, T, H, W) or null
    int T;
};

// Launchers of the LDS-staged kernels, one explicit instantiation per radius in synth_tiles.hip.
template <int R, class FT>
void launch_fwd_tiled(const FT *in, const float *w, const float *oy, const float *ox, float *out,
                      const Geom &g, bool lean, unsigned long long *reach_stats, hipStream_t st);
template <int R, class FT>
void launch_bwd_tiled(const FT *in, const float *w, const float *oy, const float *ox,
                      const float *gout, float *gw, float *goy, float *gox, const Geom &g, bool lean,
                      unsigned *maxima, hipStream_t st);
template <int R, class FT>
void launch_head_fwd(const float *w, const float *oy, const float *ox, float *out, const Geom &g,
                     const HeadArgs &ha, bool lean, unsigned long long *reach_stats, hipStream_t st);
template <int R, class FT>
void launch_head_bwd(const float *w, const float *oy, const float *ox, const float *gout, float *gw,
                     float *goy, float *gox, const Geom &g, const HeadArgs &ha, bool lean, hipStream_t st);
template <int R, bool FIXED>
void launch_gin_tiled(const float *w, const float *oy, const float *ox, const float *gout, float *gin,
                      unsigned char *ws, const Geom &g, int bits, int guarded, hipStream_t st);

// Grids with fewer 64 x 32 tiles than this run with 64 x 8 tiles (synth_tiles.hip); process-wide
// tuning knob, set through artemis_synth_set_small_grid_tiles (defined in synth_kernels.hip).
long long small_grid_tiles();

inline dim3 pixel_grid(const Geom &g, int rows_per_block)
{
    return dim3((unsigned)((g.W + 63) / 64), (unsigned)((g.H + rows_per_block - 1) / rows_per_block),
                (unsigned)g.N);
}

// XCD-aware tile order (SYNTH_XCD_REMAP, default on): workgroups are dispatched round-robin over
// the 8 XCDs (workgroup id % 8), each with its own L2.  XCD i gets the i-th contiguous band of
// tiles, so neighbouring tiles -- which share their frame halo and the DRAM pages of the plane
// rows they split -- run on the same XCD at about the same time.  The grid is 1-D in x, padded to
// a multiple of 8; a workgroup whose tile index falls past the end exits at once.  Measured at
// 1080p: forward 114.4 -> 111.6 us, backward 251.7 -> 249.0 us at R = 8; 116.2 -> 112.0 and
// 257.8 -> 251.2 us at R = 16.  Correct under any dispatch order.
#ifndef SYNTH_XCD_REMAP
#define SYNTH_XCD_REMAP 1
#endif
// SYNTH_PAIR_SEGMENTS (default off: measured and dropped).  A frame whose width is an odd
// number of 64-column segments (448 = 7 x 64, the Vimeo-90K septuplet) leaves a half-empty workgroup
// at the right edge of every band of 128-wide tiles; pairing consecutive segments of the row-major
// (band, segment) list instead removes those (960 workgroups instead of 1024 at N = 16, 256 x 448) --
// and made that forward 8 % SLOWER (117 vs 109 us; backward +-0): 1024 workgroups are exactly four
// rounds on 256 CUs, the last of them lighter, while 960 full ones leave a quarter of the chip idle
// through a fourth round of full-length workgroups.
#ifndef SYNTH_PAIR_SEGMENTS
#define SYNTH_PAIR_SEGMENTS 0
#endif
// `segments` > 1 (one staged window per 64-column wave column, see segment_origin): a workgroup
// takes `segments` CONSECUTIVE 64-column segments of the row-major list of (band, segment) pairs, so
// a frame whose width is an odd number of segments (448 = 7 x 64, the Vimeo-90K septuplet) leaves no
// half-empty workgroup at the right edge of every band.
inline dim3 tile_grid(const Geom &g, int rows_per_block, int cols_per_block = 64, int segments = 1)
{
    const unsigned tiles_y = (unsigned)((g.H + rows_per_block - 1) / rows_per_block);
#if SYNTH_XCD_REMAP
    unsigned tiles;
    if (segments > 1) {
        const unsigned segs = (unsigned)((g.W + 63) / 64) * tiles_y;
        tiles = (segs + segments - 1) / segments;
    } else {
        tiles = (unsigned)((g.W + cols_per_block - 1) / cols_per_block) * tiles_y;
    }
    return dim3(((tiles + 7) / 8) * 8, 1, (unsigned)g.N);
#else
    const unsigned tiles_x = (unsigned)((g.W + cols_per_block - 1) / cols_per_block);
    return dim3(tiles_x, tiles_y, (unsigned)g.N);
#endif
}

}  // namespace asynth

namespace {

using asynth::Geom;
using asynth::HeadArgs;
using asynth::kMaxFrames;
using asynth::pixel_grid;
using asynth::bf16_t;

// A frame as the gather code sees it: indexable, offsettable, yields fp32.  fp32 frames are plain
// pointers; bf16 frames widen on load (global_load_ushort + shift).
struct Bf16View {
    const bf16_t *p;
    __device__ __forceinline__ float operator[](size_t i) const
    {
        return __uint_as_float((unsigned)p[i] << 16);
    }
    __device__ __forceinline__ Bf16View operator+(size_t o) const { return Bf16View{p + o}; }
};
__device__ __forceinline__ const float *frame_view(const float *p) { return p; }
__device__ __forceinline__ Bf16View frame_view(const bf16_t *p) { return Bf16View{p}; }

// tile origin of this workgroup; false: no tile (only with SYNTH_XCD_REMAP)
template <int TH, int TW>
__device__ __forceinline__ bool tile_origin(const Geom &g, int &tx0, int &ty0)
{
#if SYNTH_XCD_REMAP
    const unsigned tiles_x = (unsigned)(g.W + TW - 1) / TW, tiles_y = (unsigned)(g.H + TH - 1) / TH;
    const unsigned nb = tiles_x * tiles_y, per = (nb + 7) / 8;
    const unsigned t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (t >= nb) return false;
    const unsigned ty = t / tiles_x;
    tx0 = (int)(t - ty * tiles_x) * TW;
    ty0 = (int)ty * TH;
    return true;
#else
    tx0 = blockIdx.x * TW;
    ty0 = blockIdx.y * TH;
    return true;
#endif
}


// Origin of wave column `c` (of WX) of this workgroup when every wave column has its own staged
// window (NWIN == WX): the workgroup owns WX consecutive entries of the row-major list of 64-column
// segments, so its wave columns may sit in different bands (the last segment of one band and the
// first of the next, when a band has an odd number of segments).  For an even number of segments
// per band this is exactly tile_origin's mapping.  false: no workgroup.  A wave column past the end
// of the list gets wx0 = the frame's width rounded up (>= W: inactive, as a right-edge column was).
template <int TH, int WX>
__device__ __forceinline__ bool segment_origin(const Geom &g, int c, int &wx0, int &ty0)
{
#if SYNTH_XCD_REMAP
    const unsigned segs_x = (unsigned)(g.W + 63) / 64, bands = (unsigned)(g.H + TH - 1) / TH;
    const unsigned segs = segs_x * bands, nb = (segs + WX - 1) / WX, per = (nb + 7) / 8;
    const unsigned t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (t >= nb) return false;
    const unsigned l = t * WX + (unsigned)c;
    if (l >= segs) {
        wx0 = (int)segs_x * 64;
        ty0 = (int)(bands - 1) * TH;
        return true;
    }
    const unsigned band = l / segs_x;
    wx0 = (int)(l - band * segs_x) * 64;
    ty0 = (int)band * TH;
    return true;
#else
    int tx0;
    if (!tile_origin<TH, 64 * WX>(g, tx0, ty0)) return false;
    wx0 = tx0 + c * 64;
    return true;
#endif
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return min(max(v, lo), hi); }

// two's-complement wrapping add, as the device's integer add in the reference does
__device__ __forceinline__ int wadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }

// ------------------------------------------------------------------------------------------
// Direct path: any k, d, C.  Gathers go through L1/L2.  One lane per (n, y, x).
// Block (64,4): a wave is 64 consecutive x of one row.  The per-pixel bodies are also the
// fix-up path of the LDS-staged kernels below.
// ------------------------------------------------------------------------------------------

struct TapIdx {
    int bl, tl, br, tr;  // element offsets inside one frame plane
    float fa, fb;
};

__device__ __forceinline__ TapIdx tap_index(const Geom &g, float a, float b, int ybase, int xbase)
{
    TapIdx t;
    const int ia = (int)a, ib = (int)b;  // v_cvt_i32_f32: toward zero, saturating, NaN -> 0
    const int yy = wadd(ybase, ia), xx = wadd(xbase, ib);
    const int bottom = clampi(yy, 0, g.Hin - 1), top = clampi(wadd(yy, 1), 0, g.Hin - 1);
    const int left = clampi(xx, 0, g.Win - 1), right = clampi(wadd(xx, 1), 0, g.Win - 1);
    t.bl = bottom * g.Win + left;
    t.tl = top * g.Win + left;
    t.br = bottom * g.Win + right;
    t.tr = top * g.Win + right;
    t.fa = a - (float)ia;
    t.fb = b - (float)ib;
    return t;
}

// Forward for one pixel, channels [c0, c0 + CH).  wp/ap/bp point at tap 0 of this pixel,
// ip at channel c0 of this sample's frame, op at channel c0 of this pixel's output.
// KS > 0 fixes the filter size at compile time: a tap row's loads and gathers are then issued
// together (the LDS kernels' fix-up runs this with one or two lanes active, so only the length of
// the dependent-latency chain matters: 2 round trips per tap row instead of 2 per tap).
template <int CH, int KS = 0>
__device__ __forceinline__ void fwd_pixel_direct(const float *__restrict__ ip,
                                                 const float *__restrict__ wp,
                                                 const float *__restrict__ ap,
                                                 const float *__restrict__ bp,
                                                 float *__restrict__ op, const Geom &g, int y, int x)
{
    const size_t plane = (size_t)g.H * g.W, iplane = (size_t)g.Hin * g.Win;
    const int k = KS > 0 ? KS : g.k;
    float acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = 0.f;
#pragma unroll 1
    for (int r = 0; r < k; ++r) {
#pragma unroll
        for (int s = 0; s < k; ++s) {
            const size_t t = (size_t)(r * k + s) * plane;
            const float wv = wp[t], a = ap[t], b = bp[t];
            const TapIdx ti = tap_index(g, a, b, y + g.oy0 + r * g.d, x + g.ox0 + s * g.d);
            const float ga = 1.f - ti.fa, gb = 1.f - ti.fb;
            const float c00 = ga * gb, c10 = ti.fa * gb, c01 = ga * ti.fb, c11 = ti.fa * ti.fb;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                const float *p = ip + (size_t)c * iplane;
                acc[c] += wv * (p[ti.bl] * c00 + p[ti.tl] * c10 + p[ti.br] * c01 + p[ti.tr] * c11);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CH; ++c) op[(size_t)c * plane] = acc[c];
}

template <int CH, int KS>
__global__ __launch_bounds__(256) void synth_fwd_direct(const float *__restrict__ in,
                                                         const float *__restrict__ w,
                                                         const float *__restrict__ oy,
                                                         const float *__restrict__ ox,
                                                         float *__restrict__ out, Geom g, int c0)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    const int n = blockIdx.z;
    if (x >= g.W || y >= g.H) return;
    const size_t plane = (size_t)g.H * g.W, iplane = (size_t)g.Hin * g.Win;
    const size_t pix = (size_t)y * g.W + x;
    const size_t tb = (size_t)n * g.k * g.k * plane + pix;
    fwd_pixel_direct<CH, KS>(in + ((size_t)n * g.C + c0) * iplane, w + tb, oy + tb, ox + tb,
                         out + ((size_t)n * g.C + c0) * plane + pix, g, y, x);
}

// Fused gradWeight / gradOffset_y / gradOffset_x for one pixel, all C channels (run-time loop
// unless CH > 0).  gwp/goyp/goxp point at tap 0 of this pixel (each may be null), gp at channel 0
// of gradOutput.  KS as in fwd_pixel_direct.
template <int CH = 0, int KS = 0>
__device__ __forceinline__ void bwd_pixel_direct(const float *__restrict__ ip,
                                                 const float *__restrict__ wp,
                                                 const float *__restrict__ ap,
                                                 const float *__restrict__ bp,
                                                 const float *__restrict__ gp,
                                                 float *__restrict__ gwp, float *__restrict__ goyp,
                                                 float *__restrict__ goxp, const Geom &g, int y,
                                                 int x)
{
    const size_t plane = (size_t)g.H * g.W, iplane = (size_t)g.Hin * g.Win;
    const int k = KS > 0 ? KS : g.k;
    const int C = CH > 0 ? CH : g.C;
#pragma unroll 1
    for (int r = 0; r < k; ++r) {
#pragma unroll
        for (int s = 0; s < k; ++s) {
            const size_t t = (size_t)(r * k + s) * plane;
            const float a = ap[t], b = bp[t];
            const TapIdx ti = tap_index(g, a, b, y + g.oy0 + r * g.d, x + g.ox0 + s * g.d);
            // channel-reduced corner sums: P_q = sum_c delta_c * corner_q(c)
            float pbl = 0.f, ptl = 0.f, pbr = 0.f, ptr_ = 0.f;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float dl = gp[(size_t)c * plane];
                const float *p = ip + (size_t)c * iplane;
                pbl += dl * p[ti.bl];
                ptl += dl * p[ti.tl];
                pbr += dl * p[ti.br];
                ptr_ += dl * p[ti.tr];
            }
            const float fa = ti.fa, fb = ti.fb;
            const float dlt = ptl - pbl, drt = ptr_ - pbr;
            const float lft = pbl + fa * dlt, rgt = pbr + fa * drt, across = rgt - lft;
            if (gwp) gwp[t] = lft + fb * across;
            if (goyp || goxp) {
                const float wv = wp[t];
                if (goyp) goyp[t] = wv * (dlt + fb * (drt - dlt));
                if (goxp) goxp[t] = wv * across;
            }
        }
    }
}

template <int CH, int KS>
__global__ __launch_bounds__(256) void synth_bwd_direct(const float *__restrict__ in,
                                                         const float *__restrict__ w,
                                                         const float *__restrict__ oy,
                                                         const float *__restrict__ ox,
                                                         const float *__restrict__ gout,
                                                         float *__restrict__ gw,
                                                         float *__restrict__ goy,
                                                         float *__restrict__ gox, Geom g)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    const int n = blockIdx.z;
    if (x >= g.W || y >= g.H) return;
    const size_t plane = (size_t)g.H * g.W, iplane = (size_t)g.Hin * g.Win;
    const size_t pix = (size_t)y * g.W + x;
    const size_t tb = (size_t)n * g.k * g.k * plane + pix;
    bwd_pixel_direct<CH, KS>(in + (size_t)n * g.C * iplane, w + tb, oy + tb, ox + tb,
                     gout + (size_t)n * g.C * plane + pix, gw ? gw + tb : nullptr,
                     goy ? goy + tb : nullptr, gox ? gox + tb : nullptr, g, y, x);
}

// gradInput, fp32 atomics straight to global memory (any k, d, C).  gin must be zeroed.
template <int CH, int KS>
__global__ __launch_bounds__(256) void synth_gin_atomic_direct(const float *__restrict__ w,
                                                                const float *__restrict__ oy,
                                                                const float *__restrict__ ox,
                                                                const float *__restrict__ gout,
                                                                float *__restrict__ gin, Geom g)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    const int n = blockIdx.z;
    if (x >= g.W || y >= g.H) return;
    const size_t plane = (size_t)g.H * g.W, iplane = (size_t)g.Hin * g.Win;
    const int k = KS > 0 ? KS : g.k, C = CH > 0 ? CH : g.C;
    const size_t pix = (size_t)y * g.W + x;
    const size_t tbase = (size_t)n * k * k * plane + pix;
    float *op = gin + (size_t)n * C * iplane;
    const float *gp = gout + (size_t)n * C * plane + pix;
#pragma unroll 1
    for (int r = 0; r < k; ++r) {
#pragma unroll
        for (int s = 0; s < k; ++s) {
            const size_t t = tbase + (size_t)(r * k + s) * plane;
            const float wv = w[t], a = oy[t], b = ox[t];
            const TapIdx ti = tap_index(g, a, b, y + g.oy0 + r * g.d, x + g.ox0 + s * g.d);
            const float ga = 1.f - ti.fa, gb = 1.f - ti.fb;
            const float c00 = ga * gb, c10 = ti.fa * gb, c01 = ga * ti.fb, c11 = ti.fa * ti.fb;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float dw = gp[(size_t)c * plane] * wv;
                float *p = op + (size_t)c * iplane;
                atomicAdd(p + ti.bl, dw * c00);
                atomicAdd(p + ti.tl, dw * c10);
                atomicAdd(p + ti.br, dw * c01);
                atomicAdd(p + ti.tr, dw * c11);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Deterministic gradInput: every contribution is rounded to a 64-bit fixed-point integer with
// one scale for the whole call and added with integer atomics.  Integer addition is
// associative, so the sums (and the fp32 result) are bit-identical run to run whatever the
// arrival order, for any offsets.  The scale 2^S is chosen from max|w| * max|gradOutput| so
// that the worst case (every tap of a plane clamped onto one texel) cannot overflow.
// Workspace: [0] max|w| bits, [1] max|gout| bits (uint32, atomicMax on non-negative float
// bits), accumulators (int64, one per gradInput element) from byte 256.
// ------------------------------------------------------------------------------------------

constexpr size_t kDetHeaderBytes = 256;
// Workspace header words beyond the two maxima (synth_gin_gather.hpp): word kGaFlagWord is raised to
// kOneBits (the bits of 1.0f) by gradInput's gather kernel when it meets an offset that does not
// truncate to zero -- it then gives up, and the scatter kernels, launched behind it with `guarded`,
// do the work; while the word is 0 they leave at once.  Words kGaQueueWord .. + 7: the gather
// kernel's per-XCD tile counters.
constexpr unsigned kOneBits = 0x3f800000u;
constexpr int kGaFlagWord = 2, kGaQueueWord = 4;

// max |v| of two tensors in one launch (bit patterns of |v| order like the values), for the
// fixed-point scale of gradInput: slot[0] <- a, slot[1] <- b.  Blocks [0, blocks_a) stream `a`,
// the rest `b`; 16-byte nt loads, four in flight per lane; `vec` = both pointers 16-byte aligned
// (else dword loads).  A 207 MB weight tensor takes ~40 us instead of the 200 us of a plain
// grid-stride dword loop.
__device__ __forceinline__ unsigned absbits(float v) { return __float_as_uint(v) & 0x7fffffffu; }

__global__ __launch_bounds__(256) void absmax2_kernel(const float *__restrict__ a, size_t na,
                                                       const float *__restrict__ b, size_t nb,
                                                       unsigned blocks_a, int vec,
                                                       unsigned *__restrict__ slot, int guarded)
{
    if (guarded && slot[kGaFlagWord] < kOneBits) return;  // gradInput came from the gather kernel: no scale needed
    const bool first = blockIdx.x < blocks_a;
    const float *v = first ? a : b;
    const size_t count = first ? na : nb;
    const size_t nblk = first ? blocks_a : gridDim.x - blocks_a;
    const size_t blk = first ? blockIdx.x : blockIdx.x - blocks_a;
    unsigned m = 0;
    if (vec) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 *v4 = (const f4 *)v;
        const size_t n4 = count / 4, stride = nblk * 256;
        size_t i = blk * 256 + threadIdx.x;
        for (; i + 3 * stride < n4; i += 4 * stride) {
            const f4 x0 = __builtin_nontemporal_load(v4 + i);
            const f4 x1 = __builtin_nontemporal_load(v4 + i + stride);
            const f4 x2 = __builtin_nontemporal_load(v4 + i + 2 * stride);
            const f4 x3 = __builtin_nontemporal_load(v4 + i + 3 * stride);
            m = max(m, max(max(absbits(x0.x), absbits(x0.y)), max(absbits(x0.z), absbits(x0.w))));
            m = max(m, max(max(absbits(x1.x), absbits(x1.y)), max(absbits(x1.z), absbits(x1.w))));
            m = max(m, max(max(absbits(x2.x), absbits(x2.y)), max(absbits(x2.z), absbits(x2.w))));
            m = max(m, max(max(absbits(x3.x), absbits(x3.y)), max(absbits(x3.z), absbits(x3.w))));
        }
        for (; i < n4; i += stride) {
            const f4 x0 = __builtin_nontemporal_load(v4 + i);
            m = max(m, max(max(absbits(x0.x), absbits(x0.y)), max(absbits(x0.z), absbits(x0.w))));
        }
        if (blk == 0 && threadIdx.x < (count & 3)) m = max(m, absbits(v[n4 * 4 + threadIdx.x]));
    } else {
        for (size_t i = blk * 256 + threadIdx.x; i < count; i += nblk * 256) m = max(m, absbits(v[i]));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
    // one atomic per block at most, and only when it can raise the maximum: same-address atomics
    // serialise in L2 (14 K of them cost ~140 us), while a running maximum of n blocks is raised
    // only ~ln n times
    __shared__ unsigned wave_max[4];
    if ((threadIdx.x & 63) == 0) wave_max[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = max(max(wave_max[0], wave_max[1]), max(wave_max[2], wave_max[3]));
        unsigned *s = slot + (first ? 0 : 1);
        if (m > __atomic_load_n(s, __ATOMIC_RELAXED)) atomicMax(s, m);
    }
}

// log2 of the fixed-point scale; a pure function of the two maxima, identical in every lane.
// Every contribution  gradOut * w * coef  is bounded by m = 4 max|w| max|gradOut| < 2^e
// (|coef| < 4: (1 + |fa|)(1 + |fb|) with |fa|, |fb| < 1), so scaled by 2^(30 - e) it rounds to an
// int32 (one v_rndne + v_cvt; its 24-bit mantissa is all the precision the fp32 product has) and
// is added, sign-extended, into a 64-bit integer.  count_bits bounds how many contributions can
// meet in one element so that even that sum stays below 2^62.
__device__ __forceinline__ int det_scale_log2(const unsigned *hdr, int count_bits)
{
    const float mw = __uint_as_float(hdr[0]), mg = __uint_as_float(hdr[1]);
    float m = 4.f * mw * mg;
    if (!(m > 0.f) || !(m < 3.0e38f)) return 0;
    int e;
    frexpf(m, &e);  // m < 2^e
    // 2^S and 2^-S must both be finite, normal fp32 numbers (tiny gradients times tiny weights
    // would otherwise ask for S > 127: scale = inf, every contribution saturates)
    return max(-126, min(126, min(30, 62 - count_bits) - e));
}

__device__ __forceinline__ unsigned long long to_fixed(float contrib_scaled)
{
    return (unsigned long long)(long long)(int)rintf(contrib_scaled);  // saturating, NaN -> 0
}

template <int CH, int KS>
__global__ __launch_bounds__(256) void synth_gin_fixed_direct(const float *__restrict__ w,
                                                               const float *__restrict__ oy,
                                                               const float *__restrict__ ox,
                                                               const float *__restrict__ gout,
                                                               unsigned char *__restrict__ ws,
                                                               Geom g, int count_bits)
{
    const int x = blockIdx.x * 64 + threadIdx.x;
    const int y = blockIdx.y * 4 + threadIdx.y;
    const int n = blockIdx.z;
    if (x >= g.W || y >= g.H) return;
    const float scale = ldexpf(1.f, det_scale_log2((const unsigned *)ws, count_bits));
    unsigned long long *accum = (unsigned long long *)(ws + kDetHeaderBytes);
    const size_t plane = (size_t)g.H * g.W, iplane = (size_t)g.Hin * g.Win;
    const int k = KS > 0 ? KS : g.k, C = CH > 0 ? CH : g.C;
    const size_t pix = (size_t)y * g.W + x;
    const size_t tbase = (size_t)n * k * k * plane + pix;
    unsigned long long *op = accum + (size_t)n * C * iplane;
    const float *gp = gout + (size_t)n * C * plane + pix;
#pragma unroll 1
    for (int r = 0; r < k; ++r) {
#pragma unroll
        for (int s = 0; s < k; ++s) {
            const size_t t = tbase + (size_t)(r * k + s) * plane;
            const float wv = w[t], a = oy[t], b = ox[t];
            const TapIdx ti = tap_index(g, a, b, y + g.oy0 + r * g.d, x + g.ox0 + s * g.d);
            const float ga = 1.f - ti.fa, gb = 1.f - ti.fb;
            const float c00 = ga * gb, c10 = ti.fa * gb, c01 = ga * ti.fb, c11 = ti.fa * ti.fb;
#pragma unroll
            for (int c = 0; c < C; ++c) {
                const float dw = gp[(size_t)c * plane] * wv * scale;
                unsigned long long *p = op + (size_t)c * iplane;
                atomicAdd(p + ti.bl, to_fixed(dw * c00));
                atomicAdd(p + ti.tl, to_fixed(dw * c10));
                atomicAdd(p + ti.br, to_fixed(dw * c01));
                atomicAdd(p + ti.tr, to_fixed(dw * c11));
            }
        }
    }
}

__global__ __launch_bounds__(256) void det_finalize_kernel(const unsigned char *__restrict__ ws,
                                                            float *__restrict__ gin, size_t count,
                                                            int count_bits, int guarded)
{
    if (guarded && ((const unsigned *)ws)[kGaFlagWord] < kOneBits) return;  // the gather kernel wrote gradInput
    const double inv = ldexp(1.0, -det_scale_log2((const unsigned *)ws, count_bits));  // exact
    const long long *accum = (const long long *)(ws + kDetHeaderBytes);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (size_t)gridDim.x * blockDim.x)
        gin[i] = (float)((double)accum[i] * inv);
}

// ------------------------------------------------------------------------------------------
// LDS-staged kernels for k = 5, d = 1, C = 3 (the model's configuration).
// Workgroup = WAVES waves as (64, WAVES): tile of 64 x TH output pixels, TH = WAVES * ROWS; wave w
// owns rows w, w + WAVES, ...
// LDS tile: (TH + 5 + 2R) x (64 + 5 + 2R) float4 texels, texel (ly, lx) = the three channels of
// frame pixel (clamp(gy0 + ly), clamp(gx0 + lx)): the reference's CLAMP is applied at fill
// time, so reading (ly, ly+1) x (lx, lx+1) needs no clamp.  A tap is served from LDS when
// 0 <= ly <= LH-2 and 0 <= lx <= LW-2, i.e. |trunc(offset)| <= R.
// ------------------------------------------------------------------------------------------

constexpr int kTW = 64;
constexpr int kReachSlots = 32;  // 64-bit counters per statistics record (spreads the waves' atomics)
// One wave in kReachSample reports its reach counts (the others' are dropped: the host scales the
// sample size).  With every wave reporting, a regime in which most waves have something to report
// (sigma = 4, a pan, a steep flow) queued 8160 atomics per 1080p launch on the record's two cache
// lines -- one L2 channel -- and the forward ran 15-20 us longer.
constexpr int kReachSample = 8;

// Native 4-float vector and its LDS-qualified form.  Tile reads go through an explicit
// address_space(3) pointer: with a generic pointer hipcc merges the LDS read and the global
// fallback into one flat_load per dword (select of pointers), which kills the design.
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) f32x4 lds_f32x4;

// compile-time loop: the body receives std::integral_constant-like tags so that the tap index
// can be a template argument (immediate LDS offsets)
template <int V>
struct IntTag {
    static constexpr int value = V;
};
template <int B, int E, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (B < E) {
        f(IntTag<B>{});
        static_for<B + 1, E>(f);
    }
}

template <int R, int TH, int TW = kTW>
struct TileShape {
    static constexpr int LW = TW + 5 + 2 * R;
    static constexpr int LH = TH + 5 + 2 * R;
};

// The 75 weight/offset planes and the gradient planes are touched exactly once per launch:
// non-temporal accesses keep them from displacing the (re-read) frame in L2 and, measured on
// MI355X, lift a mixed read/write stream from ~4.9 to ~6.3 TB/s (tools/exp/exp_bw.hip).
#ifndef SYNTH_NT
#define SYNTH_NT 1
#endif
__device__ __forceinline__ float ld_stream(const float *p)
{
#if SYNTH_NT
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
__device__ __forceinline__ void st_stream(float *p, float v)
{
#if SYNTH_NT
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

// ------------------------------------------------------------------------------------------
// Fix-up of taps that leave the staged tile (k = 5, d = 1, C = 3).  The hot loops give such a
// tap weight 0 (forward) or let its stores be overwritten (backward); afterwards only those taps
// are recomputed through clamped global gathers, so the cost grows with the number of far taps,
// not with 25 x the number of pixels that have one.  When few pixels of a wave are affected
// (a few dozen per 1080p frame at sigma = 1) the whole wave handles one pixel at a time, lane
// l < 25 owning tap l: two memory round trips per pixel.  A lane walking its taps alone pays
// up to 50 dependent round trips at the very end of its workgroup, and a launch cannot end before
// its slowest workgroup: measured, that tail cost 8 us (forward) to 50 us (backward) per launch.
// All 64 lanes must be active (the tap-major kernels never retire single lanes).
// ------------------------------------------------------------------------------------------

constexpr int kCoopMaxPixels = 12;  // more flagged lanes than this: every lane fixes its own pixel

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

struct Float3 {
    float x, y, z;
};

// The staged tile is centred on the tile's typical displacement (cy, cx) -- whole pixels, wave-
// uniform -- so that large coherent motion is served from LDS with a small radius.  A tap is
// served from the tile iff its truncated offsets satisfy |trunc(a) - cy| <= R and
// |trunc(b) - cx| <= R (for cy = cx = 0 that is |offset| < R + 1).  The hot loops and the fix-ups
// below use this one predicate, so every tap is computed exactly once.  NaN truncates to 0.
struct Centre {
    float y, x;
};

__device__ __forceinline__ float tap_reach(float a, float b, Centre c)
{
    const float da = (float)(int)a - c.y, db = (float)(int)b - c.x;
    return fmaxf(fabsf(da), fabsf(db));
}

template <int R>
__device__ __forceinline__ bool tap_in_tile(float a, float b, Centre c)
{
    return tap_reach(a, b, c) <= (float)R;
}

#ifndef SYNTH_CENTRE
#define SYNTH_CENTRE 1
#endif
// Centre of one tile for one frame: the rounded mean offset of the 25 taps at the tile's middle
// pixel (one 25-lane gather + a wave reduction per wave; every wave computes the same value).
// Any value is correct; a representative one keeps the taps in the tile.  The tile is moved only
// when that mean exceeds half the radius: for ordinary offsets the centre stays 0 and which taps
// take the fix-up -- hence the last bit of a pixel -- does not depend on how the frame is tiled.
template <int R>
__device__ __forceinline__ Centre tile_centre(const float *__restrict__ ab,
                                              const float *__restrict__ bb, size_t plane,
                                              const Geom &g, int tx0, int ty0, int th, int tw = 64)
{
    Centre c = {0.f, 0.f};
#if SYNTH_CENTRE
    const int l = threadIdx.x & 63;
    const size_t pix = (size_t)min(ty0 + th / 2, g.H - 1) * g.W + min(tx0 + tw / 2, g.W - 1);
    float a = 0.f, b = 0.f;
    if (l < 25) {
        a = ab[(size_t)l * plane + pix];
        b = bb[(size_t)l * plane + pix];
    }
    a = rintf(wave_sum(a) * (1.f / 25.f));
    b = rintf(wave_sum(b) * (1.f / 25.f));
    c.y = (fabsf(a) < 1.0e6f && fabsf(a) > 0.5f * R) ? a : 0.f;  // NaN / inf / absurd: no centring
    c.x = (fabsf(b) < 1.0e6f && fabsf(b) > 0.5f * R) ? b : 0.f;
    c.y = __builtin_amdgcn_readfirstlane(c.y);
    c.x = __builtin_amdgcn_readfirstlane(c.x);
#endif
    return c;
}

// The LEAN shape (PASSES == 1: one window, staged once) does not centre its window at all: it is
// the shape for offsets that stay within its radius of the PIXEL, and leaving the centre out --
// the 25-lane gather in front of every window fill, the subtractions in the tap loop -- is worth
// 3.7 % of the fused head's forward (419 -> 403 us at 1080p; per-frame forward 110 -> 108,
// backward 240 -> 238).  Fetching the next frame's centre while the current frame's taps run did
// not recover that (422 us).  A pan or any other coherent motion beyond the radius sends
// ARTEMIS_SYNTH_RADIUS_AUTO to the radius-14 shape, whose windows follow it: the reach statistic
// that selects the lean shape is measured from the pixel, not from a window centre.
#ifndef SYNTH_LEAN_CENTRE
#define SYNTH_LEAN_CENTRE 0
#endif

// ------------------------------------------------------------------------------------------
// Several windows per tile.  One window follows ONE motion; a tile that straddles a motion
// boundary (object in front of a background) or sits on a steep flow gradient has a second
// population of taps that no single centre serves, and sending all of those through the global
// fix-up is TA-bound (a 1080p frame of smooth +-30 px flow ran at 1/5 of the headline rate).
// So after a pass over the taps the workgroup votes: when enough of its pixels still have taps
// outside every window staged so far, and one of them shows a coherent displacement, the tile is
// re-staged around that displacement and the waves that own such pixels run the (pipelined) tap
// loop again, adding / storing only the taps that the new window serves and no earlier one did.
// Up to kMaxPasses windows; whatever is left after that takes the global fix-up.  Everything is
// a pure function of the operands (the vote picks the wave with the most such pixels, lowest
// index on ties), so results do not vary run to run.
// ------------------------------------------------------------------------------------------
#ifndef SYNTH_MAX_PASSES
#define SYNTH_MAX_PASSES 3
#endif
#ifndef SYNTH_HEAD_SYNC_DEDUPE  // (0: a barrier before every frame's fill, as in round 2 -- for A/B builds)
#define SYNTH_HEAD_SYNC_DEDUPE 1
#endif
#ifndef SYNTH_PASS_MIN_PIXELS
#define SYNTH_PASS_MIN_PIXELS 24
#endif
constexpr int kMaxPasses = SYNTH_MAX_PASSES;
constexpr int kPassMinPixels = SYNTH_PASS_MIN_PIXELS;  // pixels with unserved taps that justify a pass
constexpr int kPassDepth = 3;                          // taps in flight in a further pass of the backward
constexpr int kVoteSlots = 4;                          // floats of LDS per wave for the vote
constexpr float kNoCentre = 1.0e30f;                   // a window that serves nothing

struct Windows {
    Centre c0, c1, c2;  // unused ones at kNoCentre
};

__device__ __forceinline__ Windows one_window(Centre c)
{
    return Windows{c, Centre{kNoCentre, kNoCentre}, Centre{kNoCentre, kNoCentre}};
}

// smallest reach of a tap over all windows: served from LDS by some pass iff <= R
__device__ __forceinline__ float windows_reach(float a, float b, const Windows &w)
{
    return fminf(tap_reach(a, b, w.c0), fminf(tap_reach(a, b, w.c1), tap_reach(a, b, w.c2)));
}

template <int R>
__device__ __forceinline__ bool tap_in_windows(float a, float b, const Windows &w)
{
    return windows_reach(a, b, w) <= (float)R;
}

// The unserved taps of pixel (y, xs), lane l < 25 looking at tap l: truncated displacement and
// whether no window serves the tap yet.
template <int R>
__device__ __forceinline__ bool unserved_tap(const float *__restrict__ ab,
                                             const float *__restrict__ bb, size_t plane,
                                             const Geom &g, int y, int xs, const Windows &w,
                                             float &ia, float &ib)
{
    const int l = threadIdx.x & 63;
    ia = ib = 0.f;
    if (l >= 25) return false;
    const size_t o = (size_t)l * plane + (size_t)y * g.W + xs;
    const float a = ab[o], b = bb[o];
    ia = (float)(int)a;
    ib = (float)(int)b;
    return !(windows_reach(a, b, w) <= (float)R);
}

// Centre for another window, taken from pixel (y, xs): of its unserved taps the first one seeds a
// cluster -- the unserved taps displaced within R of it -- and the centre is the cluster's rounded
// mean displacement.  (A plain mean over all unserved taps fails on a pixel whose taps point at
// two objects.)  ok = false when the pixel has no unserved tap or its displacement is absurd.
// Wave-uniform result.
template <int R>
__device__ __forceinline__ Centre unserved_centre(const float *__restrict__ ab,
                                                  const float *__restrict__ bb, size_t plane,
                                                  const Geom &g, int y, int xs, const Windows &w,
                                                  bool &ok)
{
    float ia, ib;
    const bool far = unserved_tap<R>(ab, bb, plane, g, y, xs, w, ia, ib);
    const unsigned long long fmask = __ballot(far);
    const int seed = fmask ? __ffsll((long long)fmask) - 1 : 0;
    const float sy = __shfl(ia, seed), sx = __shfl(ib, seed);
    const bool member = far && fabsf(ia - sy) <= (float)R && fabsf(ib - sx) <= (float)R;
    const float n = (float)__popcll(__ballot(member));
    const float my = rintf(wave_sum(member ? ia : 0.f) / fmaxf(n, 1.f));
    const float mx = rintf(wave_sum(member ? ib : 0.f) / fmaxf(n, 1.f));
    ok = n > 0.f && fabsf(my) < 1.0e6f && fabsf(mx) < 1.0e6f;
    Centre c;
    c.y = __builtin_amdgcn_readfirstlane(my);
    c.x = __builtin_amdgcn_readfirstlane(mx);
    return c;
}

// The vote after a pass: the waves of a window agree on whether to stage another one and where.
//  1. Every wave reports how many of its pixels still have unserved taps, and the first of them; a
//     window with fewer than kPassMinPixels such pixels is done (the fix-up handles a few pixels
//     faster than a pass).
//  2. Otherwise the candidate centre comes from a pixel of the wave that owns most of them -- the
//     middle of the first run of such pixels in its row that has most (lowest wave on ties; every
//     wave of the window derives it by itself: one 25-lane gather), and every wave counts how many
//     of its pixels the candidate would serve, judged by each pixel's first and last unserved tap.  The window is staged when that is at least kPassMinPixels pixels
//     and a third of all the pixels in question: a steep flow leaves pixels on BOTH sides of the first
//     window (two populations of about half each, one further window apiece), while independent
//     large offsets (sigma = 4 px at R = 8: most pixels have a far tap or two, each in its own
//     direction) are served to a quarter at best by any centre and keep going straight to the
//     fix-up.  A pass costs about as much as the per-pixel fix-up of a few dozen pixels.
// A workgroup holds NWIN windows side by side (wave column c belongs to window c * NWIN / WX);
// barriers are workgroup-wide, so the pass loop goes on while ANY window wants another pass
// (return value) and `mine` says whether this wave's window does (then `next` is its centre).
// flagged[j]: ballot of row j's pixels with unserved taps (owners only); fmask[j]: this lane's
// unserved taps in row j.  Contains one or two barriers: every wave of the workgroup calls it,
// active or not.  slot: WAVES x 4 floats of LDS.
template <int R, int WAVES, int ROWS, int WX, int NWIN>
__device__ __forceinline__ bool vote_next_window(float *slot, int wv, bool active,
                                                 const unsigned long long (&flagged)[ROWS],
                                                 const unsigned (&fmask)[ROWS],
                                                 const int (&rowy)[ROWS], int wx0, int x,
                                                 const float *__restrict__ ab,
                                                 const float *__restrict__ bb, size_t plane,
                                                 const Geom &g, const Windows &w, Centre &next,
                                                 bool &mine)
{
    constexpr int WXW = WX / NWIN;  // wave columns per window
    const int wi = (wv % WX) / WXW;
    int cnt = 0, fy = 0;
    unsigned long long fb = 0;
    if (active) {
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            cnt += __popcll(flagged[j]);
            if (__popcll(flagged[j]) > __popcll(fb)) {  // the row with most such pixels
                fb = flagged[j];
                fy = rowy[j];
            }
        }
    }
    const bool lane0 = (threadIdx.x & 63) == 0;
    if (lane0) {
        slot[wv * kVoteSlots + 0] = (float)cnt;
        slot[wv * kVoteSlots + 1] = (float)fy;
        // the middle of the first run of such pixels: on a steep flow the run's displacements
        // change steadily, and a window centred on its first pixel would spend half its reach on
        // displacements beyond the run
        const int first = fb ? __ffsll((long long)fb) - 1 : 0;
        const unsigned long long rest = ~(fb >> first);
        const int run = rest ? __ffsll((long long)rest) - 1 : 64 - first;
        slot[wv * kVoteSlots + 2] = (float)(wx0 + first + run / 2);
    }
    __syncthreads();
    float total[NWIN];
#pragma unroll
    for (int k = 0; k < NWIN; ++k) total[k] = 0.f;
    float total_mine = 0.f, most = 0.f, py = 0.f, pxl = 0.f;
#pragma unroll
    for (int i = 0; i < WAVES; ++i) {
        const int k = (i % WX) / WXW;
        const float c = slot[i * kVoteSlots + 0];
        total[k] += c;
        if (k == wi && c > most) {
            most = c;
            py = slot[i * kVoteSlots + 1];
            pxl = slot[i * kVoteSlots + 2];
        }
    }
    bool open_any = false;
#pragma unroll
    for (int k = 0; k < NWIN; ++k) {
        open_any |= total[k] >= (float)kPassMinPixels;
        if (k == wi) total_mine = total[k];
    }
    next = Centre{0.f, 0.f};
    mine = false;
    if (!__builtin_amdgcn_readfirstlane((int)open_any)) return false;

    int would = 0;
    if (__builtin_amdgcn_readfirstlane((int)(total_mine >= (float)kPassMinPixels))) {
        bool ok;
        next = unserved_centre<R>(ab, bb, plane, g, __builtin_amdgcn_readfirstlane((int)py),
                                  __builtin_amdgcn_readfirstlane((int)pxl), w, ok);
        if (ok && cnt) {
#pragma unroll
            for (int j = 0; j < ROWS; ++j) {
                if (!flagged[j]) continue;  // wave-uniform
                // (the first AND the last of a pixel's unserved taps: coherent motion passes both,
                // independent far taps rarely do)
                const int t1 = fmask[j] ? __ffs((int)fmask[j]) - 1 : 0;
                const int t2 = fmask[j] ? 31 - __clz((int)fmask[j]) : 0;
                const size_t o = (size_t)rowy[j] * g.W + x;
                const size_t o1 = (size_t)t1 * plane + o, o2 = (size_t)t2 * plane + o;
                const float ia1 = (float)(int)ab[o1], ib1 = (float)(int)bb[o1];
                const float ia2 = (float)(int)ab[o2], ib2 = (float)(int)bb[o2];
                would += __popcll(flagged[j] &
                                  __ballot(fabsf(ia1 - next.y) <= (float)R && fabsf(ib1 - next.x) <= (float)R &&
                                           fabsf(ia2 - next.y) <= (float)R && fabsf(ib2 - next.x) <= (float)R));
            }
        }
        if (!ok) would = -1;  // (the same in every wave of the window)
    }
    if (lane0) slot[wv * kVoteSlots + 3] = (float)would;
    __syncthreads();
    float served[NWIN];
#pragma unroll
    for (int k = 0; k < NWIN; ++k) served[k] = 0.f;
#pragma unroll
    for (int i = 0; i < WAVES; ++i) served[(i % WX) / WXW] += fmaxf(slot[i * kVoteSlots + 3], 0.f);
    bool again_any = false;
#pragma unroll
    for (int k = 0; k < NWIN; ++k) {
        const bool again = total[k] >= (float)kPassMinPixels && served[k] >= (float)kPassMinPixels &&
                           3.f * served[k] >= total[k];
        again_any |= again;
        if (k == wi) mine = again;
    }
    mine = __builtin_amdgcn_readfirstlane((int)mine) != 0;
    return __builtin_amdgcn_readfirstlane((int)again_any) != 0;
}

// Contribution of ONE tap (clamped global gathers), channels 0..2.
template <class FP>
__device__ __forceinline__ Float3 fwd_tap_global(FP ip, const Geom &g, float wv,
                                                 float a, float b, int yb, int xb)
{
    const size_t iplane = (size_t)g.Hin * g.Win;
    const TapIdx ti = tap_index(g, a, b, yb, xb);
    const float ga = 1.f - ti.fa, gb = 1.f - ti.fb;
    const float c00 = ga * gb, c10 = ti.fa * gb, c01 = ga * ti.fb, c11 = ti.fa * ti.fb;
    const FP p0 = ip, p1 = ip + iplane, p2 = ip + 2 * iplane;
    Float3 o;
    o.x = wv * (p0[ti.bl] * c00 + p0[ti.tl] * c10 + p0[ti.br] * c01 + p0[ti.tr] * c11);
    o.y = wv * (p1[ti.bl] * c00 + p1[ti.tl] * c10 + p1[ti.br] * c01 + p1[ti.tr] * c11);
    o.z = wv * (p2[ti.bl] * c00 + p2[ti.tl] * c10 + p2[ti.br] * c01 + p2[ti.tr] * c11);
    return o;
}

// Which taps of a pixel are still unserved is a 25-bit mask kept by the pixel's lane (bit t = tap t),
// set in the first pass over the taps and cleared by later windows that serve them.
constexpr int kMaskMaxTaps = 4;  // at most this many unserved taps per pixel: every lane fixes its own

// Sum over the unserved taps (`mask`, wave-uniform) of pixel (y, xs), computed by the whole wave
// (lane l < 25 owns tap l).  wb/ab/bb: tap 0 of the sample's planes; ip: the sample's frame.
// Result in every lane.
template <class FP>
__device__ __forceinline__ Float3 fwd_outside_coop(FP ip,
                                                   const float *__restrict__ wb,
                                                   const float *__restrict__ ab,
                                                   const float *__restrict__ bb, const Geom &g,
                                                   unsigned mask, int y, int xs)
{
    const int l = threadIdx.x & 63;
    const size_t plane = (size_t)g.H * g.W;
    Float3 c = {0.f, 0.f, 0.f};
    if (l < 25 && ((mask >> l) & 1u)) {
        const size_t o = (size_t)l * plane + (size_t)y * g.W + xs;
        const int r = l / 5, s = l - 5 * r;
        c = fwd_tap_global(ip, g, wb[o], ab[o], bb[o], y + g.oy0 + r, xs + g.ox0 + s);
    }
    c.x = wave_sum(c.x);
    c.y = wave_sum(c.y);
    c.z = wave_sum(c.z);
    return c;
}

// The same sum by one lane for its own pixel (used when many lanes of the wave are flagged and
// some have many unserved taps): a tap row's loads are issued together.
template <class FP>
__device__ __forceinline__ Float3 fwd_outside_lane(FP ip,
                                                   const float *__restrict__ wp,
                                                   const float *__restrict__ ap,
                                                   const float *__restrict__ bp, const Geom &g,
                                                   unsigned mask, int y, int x)
{
    const size_t plane = (size_t)g.H * g.W;
    Float3 c = {0.f, 0.f, 0.f};
#pragma unroll 1
    for (int r = 0; r < 5; ++r) {
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const size_t o = (size_t)(r * 5 + s) * plane;
            if ((mask >> (r * 5 + s)) & 1u) {
                const Float3 t = fwd_tap_global(ip, g, wp[o], ap[o], bp[o], y + g.oy0 + r, x + g.ox0 + s);
                c.x += t.x;
                c.y += t.y;
                c.z += t.z;
            }
        }
    }
    return c;
}

// ... and by every lane for its own pixels one tap per step, lowest unserved tap first, all lanes
// of the wave in parallel and the wave's NR rows together: two memory round trips per step
// whatever the number of pixels, where the whole-wave form pays two per PIXEL.  For rows whose
// pixels have few unserved taps each (independent offsets a little beyond the radius: sigma = 4 px
// at R = 14 leaves one such tap in 1 % of the pixels, i.e. in every other wave row).
// mask[j] = 0 for a row that does not take part.  wp/ap/bp: tap 0 at this lane's column x of row 0
// of the planes (row j at + roff[j]).
template <int NR, class FP>
__device__ __forceinline__ void fwd_outside_steps(FP ip, const float *__restrict__ wp,
                                                  const float *__restrict__ ap,
                                                  const float *__restrict__ bp, const Geom &g,
                                                  unsigned (&mask)[NR], const size_t (&roff)[NR],
                                                  const int (&rowy)[NR], int x, float (&acc0)[NR],
                                                  float (&acc1)[NR], float (&acc2)[NR])
{
    const size_t plane = (size_t)g.H * g.W;
    for (;;) {
        unsigned any = 0;
#pragma unroll
        for (int j = 0; j < NR; ++j) any |= mask[j];
        if (!__ballot(any != 0)) break;  // wave-uniform
        float wv[NR], a[NR], b[NR];
        int t[NR];
        bool on[NR];
        unsigned long long rowon[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) {  // the operands of every row's next tap: one round of loads
            on[j] = mask[j] != 0;
            rowon[j] = __ballot(on[j]);  // rows without work issue nothing (wave-uniform)
            t[j] = on[j] ? __ffs((int)mask[j]) - 1 : 0;
            mask[j] &= mask[j] - (on[j] ? 1u : 0u);
            wv[j] = a[j] = b[j] = 0.f;
            if (rowon[j]) {
                const size_t o = (size_t)t[j] * plane + roff[j];
                wv[j] = wp[o];
                a[j] = ap[o];
                b[j] = bp[o];
            }
        }
#pragma unroll
        for (int j = 0; j < NR; ++j) {  // their gathers
            if (rowon[j]) {
                const int r = (t[j] * 13) >> 6, s = t[j] - 5 * r;  // t / 5 for t < 25
                const Float3 v = fwd_tap_global(ip, g, wv[j], a[j], b[j], rowy[j] + g.oy0 + r,
                                                x + g.ox0 + s);
                acc0[j] += on[j] ? v.x : 0.f;
                acc1[j] += on[j] ? v.y : 0.f;
                acc2[j] += on[j] ? v.z : 0.f;
            }
        }
    }
}

// One tap of the fused backward through clamped global gathers; o = element offset of
// (tap, y, x) in the sample's 25-plane tensors.  The three gradients are stored times `scale`
// (the occlusion weight in the fused head, 1 otherwise); returns w * (unscaled gradWeight), the
// tap's share of the occlusion gradient.
template <class FP>
__device__ __forceinline__ float bwd_tap_global(FP ip, const Geom &g, float wv,
                                                float a, float b, float d0, float d1, float d2, int yb,
                                                int xb, float *__restrict__ gwn,
                                                float *__restrict__ goyn, float *__restrict__ goxn,
                                                size_t o, float scale)
{
    const size_t iplane = (size_t)g.Hin * g.Win;
    const TapIdx ti = tap_index(g, a, b, yb, xb);
    const FP p0 = ip, p1 = ip + iplane, p2 = ip + 2 * iplane;
    const float pbl = d0 * p0[ti.bl] + d1 * p1[ti.bl] + d2 * p2[ti.bl];
    const float ptl = d0 * p0[ti.tl] + d1 * p1[ti.tl] + d2 * p2[ti.tl];
    const float pbr = d0 * p0[ti.br] + d1 * p1[ti.br] + d2 * p2[ti.br];
    const float ptr_ = d0 * p0[ti.tr] + d1 * p1[ti.tr] + d2 * p2[ti.tr];
    const float dlt = ptl - pbl, drt = ptr_ - pbr;
    const float lft = pbl + ti.fa * dlt, rgt = pbr + ti.fa * drt, across = rgt - lft;
    const float gwr = lft + ti.fb * across;
    const float gar = wv * (dlt + ti.fb * (drt - dlt));
    const float gbr = wv * across;
    if (gwn) gwn[o] = scale * gwr;
    if (goyn) goyn[o] = scale * gar;
    if (goxn) goxn[o] = scale * gbr;
    return wv * gwr;
}

// Unserved taps (`mask`, wave-uniform) of pixel (y, xs) recomputed and stored by the whole wave.
// gn: the sample's gradOutput planes; gwn/goyn/goxn: tap 0 of the sample's gradient planes.
// Returns (in every lane) the sum over those taps of w * unscaled gradWeight.
template <class FP>
__device__ __forceinline__ float bwd_outside_coop(FP ip,
                                                  const float *__restrict__ wb,
                                                  const float *__restrict__ ab,
                                                  const float *__restrict__ bb,
                                                  const float *__restrict__ gn, float *__restrict__ gwn,
                                                  float *__restrict__ goyn, float *__restrict__ goxn,
                                                  const Geom &g, unsigned mask, int y, int xs, float scale,
                                                  bool want_sum)
{
    const int l = threadIdx.x & 63;
    float c = 0.f;
    if (l < 25 && ((mask >> l) & 1u)) {
        const size_t plane = (size_t)g.H * g.W;
        const size_t pix = (size_t)y * g.W + xs;
        const size_t o = (size_t)l * plane + pix;
        const int r = l / 5, s = l - 5 * r;
        c = bwd_tap_global(ip, g, wb[o], ab[o], bb[o], gn[pix], gn[plane + pix], gn[2 * plane + pix],
                           y + g.oy0 + r, xs + g.ox0 + s, gwn, goyn, goxn, o, scale);
    }
    return want_sum ? wave_sum(c) : 0.f;
}

template <class FP>
__device__ __forceinline__ float bwd_outside_lane(FP ip,
                                                  const float *__restrict__ wb,
                                                  const float *__restrict__ ab,
                                                  const float *__restrict__ bb, float d0, float d1,
                                                  float d2, float *__restrict__ gwn,
                                                  float *__restrict__ goyn, float *__restrict__ goxn,
                                                  const Geom &g, unsigned mask, int y, int x, float scale)
{
    const size_t plane = (size_t)g.H * g.W;
    const size_t pix = (size_t)y * g.W + x;
    float c = 0.f;
#pragma unroll 1
    for (int r = 0; r < 5; ++r) {
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const size_t o = (size_t)(r * 5 + s) * plane + pix;
            if ((mask >> (r * 5 + s)) & 1u)
                c += bwd_tap_global(ip, g, wb[o], ab[o], bb[o], d0, d1, d2, y + g.oy0 + r, x + g.ox0 + s,
                                    gwn, goyn, goxn, o, scale);
        }
    }
    return c;
}

// one tap per step, all lanes in parallel, NR rows together (see fwd_outside_steps); gocc[j] gets
// the sum over row j's taps of w * unscaled gradWeight
template <int NR, class FP>
__device__ __forceinline__ void bwd_outside_steps(FP ip, const float *__restrict__ wb,
                                                  const float *__restrict__ ab,
                                                  const float *__restrict__ bb,
                                                  const float (&d)[NR][3], float *__restrict__ gwn,
                                                  float *__restrict__ goyn, float *__restrict__ goxn,
                                                  const Geom &g, unsigned (&mask)[NR],
                                                  const size_t (&roff)[NR], const int (&rowy)[NR], int x,
                                                  const float (&scale)[NR], float (&gocc)[NR])
{
    const size_t plane = (size_t)g.H * g.W;
    for (;;) {
        unsigned any = 0;
#pragma unroll
        for (int j = 0; j < NR; ++j) any |= mask[j];
        if (!__ballot(any != 0)) break;  // wave-uniform
        float wv[NR], a[NR], b[NR];
        int t[NR];
        bool on[NR];
        size_t o[NR];
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            on[j] = mask[j] != 0;
            t[j] = on[j] ? __ffs((int)mask[j]) - 1 : 0;
            mask[j] &= mask[j] - (on[j] ? 1u : 0u);
            o[j] = (size_t)t[j] * plane + roff[j] + x;
            wv[j] = a[j] = b[j] = 0.f;
            if (__ballot(on[j])) {  // rows without work issue nothing (wave-uniform)
                wv[j] = wb[o[j]];
                a[j] = ab[o[j]];
                b[j] = bb[o[j]];
            }
        }
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            if (on[j]) {  // (stores: a lane without a tap must not write)
                const int r = (t[j] * 13) >> 6, s = t[j] - 5 * r;  // t / 5 for t < 25
                gocc[j] += bwd_tap_global(ip, g, wv[j], a[j], b[j], d[j][0], d[j][1], d[j][2],
                                          rowy[j] + g.oy0 + r, x + g.ox0 + s, gwn, goyn, goxn, o[j],
                                          scale[j]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// Tap-major variants.  Same tile, same per-tap arithmetic, different loop order: a wave owns
// ROWS pixel rows (tile = 64 x 4*ROWS) and walks the 25 taps in the OUTER loop, visiting its
// rows inside each tap.  All workgroups of a residency round start together and advance through
// the taps at about the same pace, so chip-wide only a few of the 75 (forward) / 150 (backward)
// planes are being streamed at any moment and each plane is swept almost linearly.  HBM3E
// needs that: measured on MI355X (tools/exp), 150 concurrent plane streams run at ~5.1 TB/s,
// a handful at ~6.4-6.6 TB/s, because a DRAM row that is opened for 256 B costs as much as
// one that delivers 1 KB.
// ------------------------------------------------------------------------------------------

#ifndef SYNTH_ROWGROUP
#define SYNTH_ROWGROUP 4
#endif
constexpr int kRowGroup = SYNTH_ROWGROUP;  // rows whose LDS gathers may be in flight together

template <int ROWS>
struct TapCol {
    float w[ROWS], a[ROWS], b[ROWS];
};

// wbase/abase/bbase: wave-uniform pointers to tap 0 of this sample; roff[j]: uniform element
// offsets of the wave's rows.  Uniform base + 32-bit lane offset -> saddr addressing.
template <int ROWS>
__device__ __forceinline__ void load_tap_col(TapCol<ROWS> &t, const float *__restrict__ wbase,
                                             const float *__restrict__ abase,
                                             const float *__restrict__ bbase, size_t plane, int tap,
                                             const size_t (&roff)[ROWS], unsigned x)
{
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        const size_t o = (size_t)tap * plane + roff[j];
        t.w[j] = ld_stream(wbase + o + x);
        t.a[j] = ld_stream(abase + o + x);
        t.b[j] = ld_stream(bbase + o + x);
    }
}

// ------------------------------------------------------------------------------------------
// LDS-DMA operand ring (DMA kernels).  The (w, alpha, beta) rows of the NEXT tap are fetched with
// global_load_lds_dwordx4 -- 16 bytes per lane straight into LDS, no VGPR destination, nt cache
// policy -- instead of one dword per lane into registers: per tap and plane ONE wave instruction
// moves four of the wave's rows x 256 B (lane l: row l / 16, 16-byte chunk l % 16) into a 1 KB slot
// of the wave's own ring, and the lanes read their pixels back with ds_read_b32.  Measured on the
// stream-only stand-in of the forward (tools/exp/exp_glds.hip, profiles/r3_glds_standin.log): 90.6 us
// for the 75 planes of a 1080p frame against 103.8 us for the dword loads (one tap ahead both; with
// the default cache policy, or two / three taps ahead, the DMA form is SLOWER: 105-122 us).
// A wave reads only what it fetched itself, so a counted vmcnt orders the reads: no barrier.
// Needs W % 4 == 0 and 16-byte aligned plane pointers (the launcher checks), ROWS % 4 == 0.
// OFF by default (SYNTH_DMA): inside the real forward the ring LOSES -- lean shape, 1080p, sigma 1,
// same-process A/B (profiles/r3_dma_product_ab.log): per-frame forward 128-130 us against 110-112 us,
// fused head forward 430-439 against 419-430, in both schedules tried (request tap t + 1, then wait
// for tap t at the top of the tap; or one DMA in flight, moved to registers behind the tap's last
// row) -- with parity green.  The stand-in has the LDS to itself; the real kernel's 16 ds_read_b128
// per tap and row already keep it busy, and the DMA's LDS writes queue behind them.
// ------------------------------------------------------------------------------------------
#ifndef SYNTH_DMA
#define SYNTH_DMA 0
#endif
typedef __attribute__((address_space(3))) float lds_float;
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int ROWS>
struct DmaRing {
    static constexpr int K = ROWS / 4;           // wave instructions per plane and tap
    static constexpr int kSlot = 3 * ROWS * 64;  // floats per tap: 3 planes x ROWS rows x 64 pixels
    static constexpr int kPerTap = 3 * K;        // wave instructions per tap
    lds_float *base;                             // this wave's two slots
    size_t src[K];                               // this lane's source offset inside a plane, per instruction
    int xl;                                      // this lane's pixel inside a row
};

// rowy: the wave's pixel rows (clamped to the image); x0: its first column
template <int ROWS>
__device__ __forceinline__ DmaRing<ROWS> dma_ring(float *ring_all, int wv, const int (&rowy)[ROWS],
                                                  int x0, const Geom &g)
{
    DmaRing<ROWS> r;
    r.base = (lds_float *)ring_all + wv * (2 * DmaRing<ROWS>::kSlot);
    const int l = threadIdx.x & 63;
    r.xl = min(l, g.W - 1 - x0);  // lanes past the right edge act as the last column
    const int col = min(x0 + 4 * (l & 15), g.W - 4);
#pragma unroll
    for (int k = 0; k < DmaRing<ROWS>::K; ++k) {
        int y = rowy[4 * k];
#pragma unroll
        for (int q = 1; q < 4; ++q) y = (l >> 4) == q ? rowy[4 * k + q] : y;
        r.src[k] = (size_t)y * g.W + col;
    }
    return r;
}

// One LDS-DMA instruction: 16 bytes per lane from `gsrc` (per lane) to LDS byte address `lds_byte`
// (wave-uniform) + lane * 16, nt.  Inline asm, not __builtin_amdgcn_global_load_lds: with the builtin
// hipcc puts an s_waitcnt vmcnt(0) in front of the first LDS read that follows (it cannot tell the
// window reads from the ring), which waits for the tap just requested and undoes the prefetch.  The
// asm form is absent from the compiler's vmcnt bookkeeping -- its own waits only get stricter -- and
// the reads are ordered by wait_vmcnt above.  M0 (the LDS base of the DMA) is saved and restored
// inside the statement.
__device__ __forceinline__ void glds16_nt(const float *gsrc, unsigned lds_byte)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(lds_byte)
                 : "memory");
}

template <int ROWS>
__device__ __forceinline__ void dma_issue(const DmaRing<ROWS> &r, const float *__restrict__ wb,
                                          const float *__restrict__ ab, const float *__restrict__ bb,
                                          size_t plane, int tap, int slot)
{
    // (an address_space(3) pointer's value is the byte offset inside the workgroup's LDS)
    const unsigned d = (unsigned)__builtin_amdgcn_readfirstlane(
        (int)(unsigned)(size_t)(r.base + slot * DmaRing<ROWS>::kSlot));
#pragma unroll
    for (int k = 0; k < DmaRing<ROWS>::K; ++k) {
        const size_t o = (size_t)tap * plane + r.src[k];
        glds16_nt(wb + o, d + k * 1024);
        glds16_nt(ab + o, d + ROWS * 256 + k * 1024);
        glds16_nt(bb + o, d + 2 * ROWS * 256 + k * 1024);
    }
}

template <int ROWS>
__device__ __forceinline__ void dma_read(const DmaRing<ROWS> &r, TapCol<ROWS> &t, int slot)
{
    const lds_float *s = r.base + slot * DmaRing<ROWS>::kSlot + r.xl;
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        t.w[j] = s[j * 64];
        t.a[j] = s[ROWS * 64 + j * 64];
        t.b[j] = s[2 * ROWS * 64 + j * 64];
    }
}

// Stage the clamped frame window.  Texels are fetched in batches with all of a batch's loads issued
// before the first LDS write: left to itself hipcc waits vmcnt(0) after every texel, which
// serialises ~14 L2 round trips (~10 us) at the head of every workgroup.  The batch size balances
// the rounds: as few rounds as batches of at most kFillBatchMax allow, all of about the same size
// (forward, 256 lanes per window: R = 8 one round of 13 texels per lane, R = 14 one of 19 -- two
// rounds of 10 cost the forward 1.5 %; backward, 128 lanes per window: R = 14 two rounds of 19).
#ifndef SYNTH_FILL_BATCH
#define SYNTH_FILL_BATCH 19
#endif
constexpr int kFillBatchMax = SYNTH_FILL_BATCH;

template <int R, int TH, int THREADS, int TW, class FP>
__device__ __forceinline__ void fill_tile_n(f32x4 *tile, FP ip, const Geom &g,
                                            int gy0, int gx0, int tid)
{
    constexpr int LW = TileShape<R, TH, TW>::LW, LH = TileShape<R, TH, TW>::LH;
    constexpr int N = LH * LW;
    constexpr int ROUNDS = (N + THREADS * kFillBatchMax - 1) / (THREADS * kFillBatchMax);
    constexpr int BATCH = (N + THREADS * ROUNDS - 1) / (THREADS * ROUNDS);
    const int iplane = g.Hin * g.Win;
#pragma unroll 1
    for (int base = tid; base < N; base += THREADS * BATCH) {
        float c0[BATCH], c1[BATCH], c2[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int i = min(base + u * THREADS, N - 1);  // tail lanes re-fetch the last texel
            const int ly = i / LW, lx = i - ly * LW;
            const int gy = clampi(gy0 + ly, 0, g.Hin - 1), gx = clampi(gx0 + lx, 0, g.Win - 1);
            const int o = gy * g.Win + gx;
            c0[u] = ip[o];
            c1[u] = ip[(size_t)iplane + o];
            c2[u] = ip[2 * (size_t)iplane + o];
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
            const int i = min(base + u * THREADS, N - 1);
            tile[i] = f32x4{c0[u], c1[u], c2[u], 0.f};
        }
    }
}

struct Corners {
    f32x4 bl, tl, br, tr;
};

// Byte address (within the tile) of the bottom-left texel of a tap's footprint:
//   ((py + r + R + ia) * LW + (px + s + R + ib)) * 16
//       = ia * (LW*16) + ib * 16 + rowbase + TAPOFF*16
// rowbase = ((py + R) * LW + px + R) * 16 is one VGPR per row, set up once; the tap's own
// displacement TAPOFF = r * LW + s is a compile-time constant that rides in the ds_read
// immediate offset, so a tap costs v_mad_i32_i24 + v_lshl_add_u32 + v_min_u32.
// The unsigned min keeps the four reads in bounds whatever the offsets are (a negative address
// is a huge unsigned); a tap that really is outside is caught by the |offset| test below.
template <int R, int TH, int TAPOFF, int TW>
__device__ __forceinline__ Corners tile_corners_at(const lds_f32x4 *tile, int ia, int ib,
                                                   int rowbase)
{
    constexpr int LW = TileShape<R, TH, TW>::LW, LH = TileShape<R, TH, TW>::LH;
    const unsigned addr = (unsigned)(__mul24(ia, LW * 16) + rowbase + (ib << 4));
    const unsigned safe = min(addr, (unsigned)(((LH - 1) * LW - 2 - TAPOFF) * 16));
    const lds_f32x4 *p =
        (const lds_f32x4 *)((const __attribute__((address_space(3))) char *)tile + safe);
    Corners q;
    q.bl = p[TAPOFF];
    q.br = p[TAPOFF + 1];
    q.tl = p[TAPOFF + LW];
    q.tr = p[TAPOFF + LW + 1];
    return q;
}

// The fused ChronoSynth head (HEAD = true; model/chrono_synth.py:114-129) runs the same body once
// per context frame inside the workgroup: frame f's tile is staged, its 25 taps are applied with
// the weights/offsets of slice [n, f] of the 5-D tensors read in place, and the result is folded
// into  out = sum_f occ[n, f] * S_f  in registers, in the reference's order (product rounded,
// then added: `occ * moduleSynth(...)`, Python `sum`), so the output has the bits of the
// composition without its intermediate tensors.

// Inside the head's frame loop every per-tap address (tap * plane + row offset: 200 wave-uniform
// 64-bit values) is loop-invariant, and LLVM hoists them all out of the loop and then spills
// them.  Passing the two ingredients through an empty asm once per frame makes them opaque, so
// the addresses are formed where they are used, as in the single-frame kernels.
__device__ __forceinline__ size_t per_frame(size_t v)
{
    asm volatile("" : "+s"(v));
    return v;
}

// ... and the rare fix-up's 25 per-tap addresses of each row (lane-dependent) are hoisted to the
// top of the frame iteration and spilled across the hot loop unless the lane's column is made
// opaque inside the fix-up branch.
__device__ __forceinline__ int in_branch(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}


#ifdef SYNTH_DEV_TILE_CLOCK  // developer builds (tools/tile_clock.py): per-wave timeline of the forward
static __device__ unsigned long long *g_tile_log;
#endif

// PASSES: how many windows a tile may stage per frame (kMaxPasses; 1 = the first window only: no vote,
// no further passes -- the lean shape for offsets known to stay within the radius, synth_tiles.hip)
// DMA: the operand planes come through the LDS-DMA ring above (lean shape only: PASSES == 1).
template <int R, int WAVES, int ROWS, int OCC, bool HEAD, class FT, int WX = 1, int NWIN = 1,
          int PASSES = kMaxPasses, bool DMA = false>
__global__ __launch_bounds__(64 * WAVES, OCC) void synth_fwd_tapmajor(
    const FT *__restrict__ in, const float *__restrict__ w, const float *__restrict__ oy,
    const float *__restrict__ ox, float *__restrict__ out, Geom g, HeadArgs ha,
    unsigned long long *__restrict__ reach_stats)
{
    // WX waves side by side: the tile is 64 * WX pixels wide and (WAVES / WX) * ROWS rows high
    constexpr int TW = 64 * WX, WY = WAVES / WX, TH = WY * ROWS;
    // NWIN staged windows side by side, each serving SW = TW / NWIN tile columns (WXW wave columns)
    // with its own centres: the wide tile keeps its 512-byte plane rows, while the motion only has
    // to be coherent over SW pixels for a window to serve it
    constexpr int SW = TW / NWIN, WXW = WX / NWIN;
    constexpr int LW = TileShape<R, TH, SW>::LW, LH = TileShape<R, TH, SW>::LH;
    static_assert(WX % NWIN == 0, "wave columns must split evenly over the windows");
    __shared__ f32x4 tile_all[NWIN * LH * LW];
    __shared__ float vote_slot[WAVES * kVoteSlots];
    static_assert(!DMA || (PASSES == 1 && ROWS % 4 == 0), "the DMA ring serves the single-pass shapes");
    constexpr int DR = ROWS % 4 == 0 ? ROWS : 4;  // (the ring's type must exist for every ROWS)
    __shared__ __attribute__((aligned(16))) float dma_all[DMA ? WAVES * 2 * DmaRing<DR>::kSlot : 4];
#ifdef SYNTH_DEV_TILE_CLOCK
    const unsigned long long dbg_t0 = wall_clock64();
    unsigned long long dbg_t1 = 0, dbg_t2 = 0, dbg_a = 0, dbg_b = 0, dbg_c = 0;
    int dbg_p = 0, dbg_left = 0;
#endif
    const int n = blockIdx.z;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int wy = wv / WX;          // the wave's first row slot
    const int wi = (wv % WX) / WXW;  // its window
    int wx0, ty0, sx0;               // the wave's first column, its band's first row, its window's first column
    if constexpr (SYNTH_PAIR_SEGMENTS && WX > 1 && NWIN == WX) {
        // one window per wave column: each takes its own 64-column segment (segment_origin)
        if (!segment_origin<TH, WX>(g, wv % WX, wx0, ty0)) return;
        sx0 = wx0;
    } else {
        int tx0;
        if (!tile_origin<TH, TW>(g, tx0, ty0)) return;
        wx0 = tx0 + (wv % WX) * 64;
        sx0 = tx0 + wi * SW;
    }
    f32x4 *tile = tile_all + wi * (LH * LW);
    const lds_f32x4 *ltile = (const lds_f32x4 *)tile;
    const int tid = (wy * WXW + (wv % WX) % WXW) * 64 + threadIdx.x;  // thread index among the window's
    const size_t plane = (size_t)g.H * g.W, iplane = (size_t)g.Hin * g.Win;
    // A lane past the right edge acts as the last image column (same operands, same arithmetic,
    // same stores as that column's owner), like the rows below: no lane ever retires on its own,
    // so wave-wide operations in the fix-up see all 64 lanes.
    const int x = min(wx0 + (int)threadIdx.x, g.W - 1), px = x - sx0;
    const bool owner = wx0 + (int)threadIdx.x == x;   // not a right-edge duplicate
    // wave-uniform; a wave without pixels only takes part in the tile fills, barriers and votes
    const bool active = ty0 + wy < g.H && wx0 < g.W;

    // A row past the bottom edge is replaced by the last image row: the wave then recomputes
    // and re-stores exactly what that row's owner stores (same operands, same arithmetic), which
    // keeps the unrolled body free of per-row branches.
    int rowy[ROWS];
    size_t roff[ROWS];
    int rowbase[ROWS];
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        rowy[j] = min(ty0 + wy + WY * j, g.H - 1);
        roff[j] = (size_t)rowy[j] * g.W;
        rowbase[j] = ((rowy[j] - ty0 + R) * LW + px + R) * 16;
    }
    DmaRing<DR> ring;
    if constexpr (DMA) ring = dma_ring<ROWS>(dma_all, wv, rowy, wx0, g);
    float res0[ROWS], res1[ROWS], res2[ROWS];  // HEAD: the running sum over frames
#pragma unroll
    for (int j = 0; j < ROWS; ++j) res0[j] = res1[j] = res2[j] = 0.f;
    // Reach statistics for ARTEMIS_SYNTH_RADIUS_AUTO (synth_kernels.hip): over the taps of the wave's
    // FIRST row (a 1 / ROWS sample of the pixels), how many are displaced beyond 8 and beyond 14 whole
    // pixels from the first window's centre -- low half-word / high half-word (at most 25 x 8 frames
    // per lane, 12 800 per wave: no carry between the halves).
    unsigned reach_cnt = 0u;

    const int T = HEAD ? ha.T : 1;
    bool tile_free = true;  // no wave will read the staged windows again before the next barrier
    constexpr bool kCentred = SYNTH_LEAN_CENTRE || PASSES > 1;  // (the lean shape: see SYNTH_LEAN_CENTRE)
    for (int f = 0; f < T; ++f) {
        size_t pl = plane, rf[ROWS];
#pragma unroll
        for (int j = 0; j < ROWS; ++j) rf[j] = roff[j];
        if constexpr (HEAD) {
            pl = per_frame(pl);
#pragma unroll
            for (int j = 0; j < ROWS; ++j) rf[j] = per_frame(rf[j]);
        }
        const auto ip = frame_view((HEAD ? (const FT *)ha.frame[f] : in) + (size_t)n * 3 * iplane);
        const size_t nb = ((size_t)n * T + f) * 25 * pl;
        const float *wb = w + nb, *ab = oy + nb, *bb = ox + nb;

        // first tap of this lane's pixels: issued before the tile fill so the stream is already
        // in flight while the tile is staged
        TapCol<ROWS> cur;
        float occv[ROWS];
        if (active) {
            if constexpr (DMA) dma_issue<ROWS>(ring, wb, ab, bb, pl, 0, 0);
            else load_tap_col<ROWS>(cur, wb, ab, bb, pl, 0, rf, (unsigned)x);
            if constexpr (HEAD) {
                const float *op = ha.occ + ((size_t)n * T + f) * pl;
#pragma unroll
                for (int j = 0; j < ROWS; ++j) occv[j] = ld_stream(op + rf[j] + (unsigned)x);
            }
        }

        float acc0[ROWS], acc1[ROWS], acc2[ROWS];
        unsigned fmask[ROWS];  // per row: bit t = tap t of this lane's pixel is served by no window yet
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            acc0[j] = acc1[j] = acc2[j] = 0.f;
            fmask[j] = 0u;
        }

        // One pass over the 25 taps with the window centred at `ctr`.  FIRST: the window is the
        // tile's first one (every tap it serves is added); otherwise only taps that it serves and
        // no earlier window (`prev`) did.
        auto run_taps = [&](auto first_tag, const Centre ctr) {
            constexpr bool FIRST = decltype(first_tag)::value != 0;
            const int cbase = (((int)ctr.y) * LW + (int)ctr.x) * 16;  // the window's shift
            if constexpr (DMA) {  // tap 0 (requested before the fill) to registers, tap 1 requested
                wait_vmcnt<0>();
                dma_read<ROWS>(ring, cur, 0);
                dma_issue<ROWS>(ring, wb, ab, bb, pl, 1, 1);
            }
            static_for<0, 25>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                constexpr int r = t / 5, s = t % 5;
                TapCol<ROWS> nxt;
                if constexpr (!DMA) {
                    if (t < 24) load_tap_col<ROWS>(nxt, wb, ab, bb, pl, t + 1, rf, (unsigned)x);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < ROWS; ++j) {
                    if constexpr (DMA) {
                        // Before the tap's last row: tap t + 1 (requested a tap ago) has landed -- it is
                        // the only DMA in flight, which is what the DRAM wants -- so its values move
                        // from the ring to registers behind this row's arithmetic, and tap t + 2 is
                        // requested into the slot tap t came from.
                        if (j == ROWS - 1 && t < 24) {
                            wait_vmcnt<0>();
                            dma_read<ROWS>(ring, nxt, (t + 1) & 1);
                            if (t < 23) dma_issue<ROWS>(ring, wb, ab, bb, pl, t + 2, t & 1);
                        }
                    }
                    const float wv = cur.w[j], a = cur.a[j], b = cur.b[j];
                    const int ia = (int)a, ib = (int)b;
                    const float fa = a - (float)ia, fb = b - (float)ib;
                    const float reach = fmaxf(fabsf((float)ia - ctr.y), fabsf((float)ib - ctr.x));
                    bool near = reach <= (float)R;
                    if constexpr (FIRST) {
                        fmask[j] |= near ? 0u : 1u << t;
                        // (beyond 8: from the PIXEL -- what the lean shape, which has no centre, would
                        // see; beyond 14: from this window's centre -- what radius 14 leaves over)
                        if (j == 0 && R >= 8)
                            reach_cnt += (fmaxf(fabsf((float)ia), fabsf((float)ib)) > 8.f ? 1u : 0u) +
                                         (reach > 14.f ? 0x10000u : 0u);
                    } else {
                        near = near && ((fmask[j] >> t) & 1u);
                        fmask[j] &= near ? ~(1u << t) : ~0u;
                    }
                    const Corners q = tile_corners_at<R, TH, r * LW + s, SW>(ltile, ia, ib, rowbase[j] - cbase);
                    const float ga = 1.f - fa, gb = 1.f - fb;
                    const float c00 = ga * gb, c10 = fa * gb, c01 = ga * fb, c11 = fa * fb;
                    // A tap this window does not serve is added elsewhere.  Its term is dropped by a
                    // select, not by a zero weight: the clamped LDS address of such a tap points at an
                    // unrelated texel, and 0 * Inf / 0 * NaN from there would poison a pixel the
                    // reference leaves finite.
                    const float n0 = acc0[j] + wv * (q.bl.x * c00 + q.tl.x * c10 + q.br.x * c01 + q.tr.x * c11);
                    const float n1 = acc1[j] + wv * (q.bl.y * c00 + q.tl.y * c10 + q.br.y * c01 + q.tr.y * c11);
                    const float n2 = acc2[j] + wv * (q.bl.z * c00 + q.tl.z * c10 + q.br.z * c01 + q.tr.z * c11);
                    acc0[j] = near ? n0 : acc0[j];
                    acc1[j] = near ? n1 : acc1[j];
                    acc2[j] = near ? n2 : acc2[j];
                    // Pin the arithmetic in place: instruction selection otherwise floats the
                    // chain-free FMAs to the end of the unrolled block, keeping every LDS result of
                    // the pixel live (512 VGPRs or spills).  The sched_barrier bounds the LDS results
                    // in flight to kRowGroup rows (16 VGPRs each).
                    asm volatile("" : "+v"(acc0[j]), "+v"(acc1[j]), "+v"(acc2[j]), "+v"(fmask[j]));
                    if (FIRST && j == 0 && R >= 8) asm volatile("" : "+v"(reach_cnt));
                    if (j % kRowGroup == kRowGroup - 1) __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_sched_barrier(0);
                cur = nxt;
            });
        };

        // Rows whose pixels have at most kMaskMaxTaps unserved taps each: every lane fixes its own
        // pixels through clamped global gathers, all such rows together (fwd_outside_steps), and the
        // rows are done (mask cleared).  Called right after the first pass -- before the vote's
        // barrier, so that the gathers of one wave overlap the tap loops of the others, and such
        // pixels do not count in the vote -- and again after the last pass.
        auto fix_few = [&]() {
            unsigned sm[ROWS];
            bool some = false;
#pragma unroll
            for (int j = 0; j < ROWS; ++j) {
                const bool few = __ballot(fmask[j] != 0u) && !__ballot(__popc(fmask[j]) > kMaskMaxTaps);
                sm[j] = few ? fmask[j] : 0u;  // (few is wave-uniform)
                if (few) fmask[j] = 0u;
                some |= few;
            }
            if (some)  // wave-uniform, rare
                fwd_outside_steps<ROWS>(ip, wb + x, ab + x, bb + x, g, sm, rf, rowy, x, acc0, acc1, acc2);
        };

        Windows win = one_window(kCentred ? tile_centre<R>(ab, bb, pl, g, sx0, ty0, TH, SW) : Centre{0.f, 0.f});
        Centre ctr = win.c0;
        unsigned long long flagged[ROWS];
        bool mine = true;  // this wave's window is (re)staged in the coming pass
#pragma unroll 1
        for (int p = 0;; ++p) {
            // (the per-tap addresses are invariant in this loop too: keep them from being hoisted
            // out of it and spilled, as in the head's frame loop)
            pl = per_frame(pl);
#pragma unroll
            for (int j = 0; j < ROWS; ++j) rf[j] = per_frame(rf[j]);
#ifdef SYNTH_DEV_TILE_CLOCK
            if (p == 1) dbg_a = wall_clock64();
#endif
            // every wave is done with the previous window (frame f - 1's last one; within a frame
            // the vote's barrier has already seen to that -- and when frame f - 1 ended on a vote
            // that asked for no further pass, so has that barrier: no wave reads a window after it)
#ifdef SYNTH_DEV_NOFILL  // timing experiment only (wrong results): the head stages frame 0's window alone
            if (!HEAD || f == 0) {
#endif
            if (HEAD && f > 0 && p == 0 && !(SYNTH_HEAD_SYNC_DEDUPE && tile_free)) __syncthreads();
            if (mine)
                fill_tile_n<R, TH, 64 * (WAVES / NWIN), SW>(tile, ip, g, ty0 + g.oy0 - R + (int)ctr.y,
                                                            sx0 + g.ox0 - R + (int)ctr.x, tid);
            __syncthreads();
#ifdef SYNTH_DEV_NOFILL
            }
#endif
#ifdef SYNTH_DEV_TILE_CLOCK
            if (p == 1) dbg_b = wall_clock64();
#endif
            if (p == 0) {
                if (active) {
                    run_taps(IntTag<1>{}, ctr);
                    fix_few();
                }
#ifdef SYNTH_DEV_TILE_CLOCK
                dbg_t1 = wall_clock64();
#endif
            } else {
                bool any = false;
#pragma unroll
                for (int j = 0; j < ROWS; ++j) any |= flagged[j] != 0;
                if (mine && active && any) {  // wave-uniform
                    load_tap_col<ROWS>(cur, wb, ab, bb, pl, 0, rf, (unsigned)x);
                    run_taps(IntTag<0>{}, ctr);
#ifdef SYNTH_DEV_TILE_CLOCK
                    dbg_p += 1;
                    if (p == 1) dbg_c = wall_clock64();
#endif
                }
            }
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
                flagged[j] = active ? __ballot(fmask[j] != 0u && owner) : 0ull;
            tile_free = false;
            if (p + 1 == PASSES) break;
            Centre next;
            if (!vote_next_window<R, WAVES, ROWS, WX, NWIN>(vote_slot, wv, active, flagged, fmask, rowy, wx0, x,
                                                            ab, bb, pl, g, win, next, mine)) {
                tile_free = true;  // (workgroup-uniform: every wave has passed the vote's barrier)
                break;
            }
            if (mine) {  // (else another window of the workgroup goes on: only its barriers matter here)
                if (p == 0) win.c1 = next;
                else win.c2 = next;
                ctr = next;
            }
        }
        if (!active) continue;  // (it has taken part in this frame's fills, barriers and votes)
#ifdef SYNTH_DEV_TILE_CLOCK
        dbg_t2 = wall_clock64();
#endif

        // Taps that no window served are added through clamped global gathers.  Rows whose pixels
        // have few such taps each went first (fix_few, right after the first pass and again here) ...
        fix_few();
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            // ... the others pixel by pixel by the whole wave, or lane by lane when there are many
            unsigned long long todo = __ballot(fmask[j] != 0u && owner);
#ifdef SYNTH_DEV_TILE_CLOCK
            dbg_left += __popcll(todo);
#endif
            if (todo) {  // wave-uniform, rare
                const size_t o = rf[j] + x;
                if (__popcll(todo) <= kCoopMaxPixels) {
                    while (todo) {
                        const int b = __ffsll((long long)todo) - 1;
                        todo &= todo - 1;
                        const Float3 c = fwd_outside_coop(ip, wb, ab, bb, g, (unsigned)__shfl((int)fmask[j], b),
                                                          rowy[j], wx0 + b);
                        if (x == wx0 + b) {  // the owner and its right-edge duplicates
                            acc0[j] += c.x;
                            acc1[j] += c.y;
                            acc2[j] += c.z;
                        }
                    }
                } else if (fmask[j]) {
                    const Float3 c = fwd_outside_lane(ip, wb + o, ab + o, bb + o, g, fmask[j], rowy[j], x);
                    acc0[j] += c.x;
                    acc1[j] += c.y;
                    acc2[j] += c.z;
                }
            }
            if constexpr (HEAD) {
                // occ * S_f rounded to fp32, then added: the reference's two separate ops
                const float p0 = occv[j] * acc0[j], p1 = occv[j] * acc1[j], p2 = occv[j] * acc2[j];
                res0[j] = res0[j] + p0;
                res1[j] = res1[j] + p1;
                res2[j] = res2[j] + p2;
            } else {
                res0[j] = acc0[j];
                res1[j] = acc1[j];
                res2[j] = acc2[j];
            }
        }
    }
    if (!active) return;
    if (R >= 8 && reach_stats && ((blockIdx.x + wv) & (kReachSample - 1)) == 0) {  // wave-uniform
        reach_cnt = owner ? reach_cnt : 0u;
        if (__ballot(reach_cnt != 0u)) {  // (small offsets: nothing to report, no atomic)
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) reach_cnt += (unsigned)__shfl_xor((int)reach_cnt, o);
            if (threadIdx.x == 0)  // one no-return atomic per wave, spread over kReachSlots addresses
                __hip_atomic_fetch_add(reach_stats + (((blockIdx.x + wv) / kReachSample) & (kReachSlots - 1)),
                                       (unsigned long long)(reach_cnt >> 16) << 32 | (reach_cnt & 0xffffu),
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        float *orow = out + (size_t)n * 3 * plane + roff[j];
        st_stream(orow + (unsigned)x, res0[j]);
        st_stream(orow + plane + (unsigned)x, res1[j]);
        st_stream(orow + 2 * plane + (unsigned)x, res2[j]);
    }
#ifdef SYNTH_DEV_TILE_CLOCK
    if (g_tile_log && threadIdx.x == 0) {
        unsigned long long *q =
            g_tile_log + ((size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * WAVES + wv) * 12;
        q[0] = dbg_t0, q[1] = dbg_t1, q[2] = dbg_t2, q[3] = wall_clock64();
        q[4] = dbg_p, q[5] = dbg_left, q[6] = (unsigned long long)(unsigned)sx0 << 32 | (unsigned)ty0;
        q[7] = __smid();
        q[8] = dbg_a, q[9] = dbg_b, q[10] = dbg_c;
    }
#endif
}

// HEAD = true: the fused ChronoSynth head's backward.  gradOutput is the gradient of
// out = sum_f occ_f * S_f; per frame the kernel forms the unscaled tap gradients from it, stores
// them times occ_f into slice [n, f] of the 5-D gradient tensors in place, and accumulates
// gocc[n, f] = sum_c gradOutput_c * S_f,c = sum_t w_t * (unscaled gradWeight_t) in registers.
template <int R, int WAVES, int ROWS, int OCC, bool ALL, bool HEAD, class FT, int WX = 1, int NWIN = 1,
          int PASSES = kMaxPasses>
__global__ __launch_bounds__(64 * WAVES, OCC) void synth_bwd_tapmajor(
    const FT *__restrict__ in, const float *__restrict__ w, const float *__restrict__ oy,
    const float *__restrict__ ox, const float *__restrict__ gout, float *__restrict__ gw,
    float *__restrict__ goy, float *__restrict__ gox, Geom g, HeadArgs ha, unsigned *__restrict__ maxima)
{
    // WX waves side by side: the tile is 64 * WX pixels wide and (WAVES / WX) * ROWS rows high
    constexpr int TW = 64 * WX, WY = WAVES / WX, TH = WY * ROWS;
    // NWIN staged windows side by side, each serving SW = TW / NWIN tile columns (WXW wave columns)
    // with its own centres: the wide tile keeps its 512-byte plane rows, while the motion only has
    // to be coherent over SW pixels for a window to serve it
    constexpr int SW = TW / NWIN, WXW = WX / NWIN;
    constexpr int LW = TileShape<R, TH, SW>::LW, LH = TileShape<R, TH, SW>::LH;
    static_assert(WX % NWIN == 0, "wave columns must split evenly over the windows");
    __shared__ f32x4 tile_all[NWIN * LH * LW];
    __shared__ float vote_slot[WAVES * kVoteSlots];
    const int n = blockIdx.z;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int wy = wv / WX;          // the wave's first row slot
    const int wi = (wv % WX) / WXW;  // its window
    int wx0, ty0, sx0;               // the wave's first column, its band's first row, its window's first column
    if constexpr (SYNTH_PAIR_SEGMENTS && WX > 1 && NWIN == WX) {
        // one window per wave column: each takes its own 64-column segment (segment_origin)
        if (!segment_origin<TH, WX>(g, wv % WX, wx0, ty0)) return;
        sx0 = wx0;
    } else {
        int tx0;
        if (!tile_origin<TH, TW>(g, tx0, ty0)) return;
        wx0 = tx0 + (wv % WX) * 64;
        sx0 = tx0 + wi * SW;
    }
    f32x4 *tile = tile_all + wi * (LH * LW);
    const lds_f32x4 *ltile = (const lds_f32x4 *)tile;
    const int tid = (wy * WXW + (wv % WX) % WXW) * 64 + threadIdx.x;  // thread index among the window's
    const size_t plane = (size_t)g.H * g.W, iplane = (size_t)g.Hin * g.Win;
    // lanes past the right edge duplicate the last image column (see forward)
    const int x = min(wx0 + (int)threadIdx.x, g.W - 1), px = x - sx0;
    const bool owner = wx0 + (int)threadIdx.x == x;
    const bool active = ty0 + wy < g.H && wx0 < g.W;  // wave-uniform (see forward)

    int rowy[ROWS];  // rows past the bottom edge duplicate the last image row (see forward)
    size_t roff[ROWS];
    int rowbase[ROWS];
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        rowy[j] = min(ty0 + wy + WY * j, g.H - 1);
        roff[j] = (size_t)rowy[j] * g.W;
        rowbase[j] = ((rowy[j] - ty0 + R) * LW + px + R) * 16;
    }
    const float *gb_ = gout + (size_t)n * 3 * plane;
    float d[ROWS][3];
    // max |weight| and max |gradOutput| over this lane's pixels, for the caller's gradInput pass
    // (its fixed-point scale; saves that pass a 53 us sweep over the weights): maxima[0], maxima[1]
    unsigned mx_w = 0u, mx_g = 0u;  // max |weight|, max |gradOutput| as bit patterns

    const int T = HEAD ? ha.T : 1;
    bool tile_free = true;  // no wave will read the staged windows again before the next barrier
    constexpr bool kCentred = SYNTH_LEAN_CENTRE || PASSES > 1;  // (the lean shape: see SYNTH_LEAN_CENTRE)
    for (int f = 0; f < T; ++f) {
        size_t pl = plane, rf[ROWS];
#pragma unroll
        for (int j = 0; j < ROWS; ++j) rf[j] = roff[j];
        if constexpr (HEAD) {
            pl = per_frame(pl);
#pragma unroll
            for (int j = 0; j < ROWS; ++j) rf[j] = per_frame(rf[j]);
        }
        const auto ip = frame_view((HEAD ? (const FT *)ha.frame[f] : in) + (size_t)n * 3 * iplane);
        const size_t nb = ((size_t)n * T + f) * 25 * pl;
        const float *wb = w + nb, *ab = oy + nb, *bb = ox + nb;

        TapCol<ROWS> cur;
        float occv[ROWS], gocc[ROWS];
        if (active) {
            load_tap_col<ROWS>(cur, wb, ab, bb, pl, 0, rf, (unsigned)x);
            if constexpr (HEAD) {
                const float *op = ha.occ + ((size_t)n * T + f) * pl;
#pragma unroll
                for (int j = 0; j < ROWS; ++j) {
                    occv[j] = ld_stream(op + rf[j] + (unsigned)x);
                    gocc[j] = 0.f;
                }
            }
        }

        unsigned fmask[ROWS];  // bit t = tap t of this lane's pixel is served by no window yet
#pragma unroll
        for (int j = 0; j < ROWS; ++j) fmask[j] = 0u;

        // One pass over the 25 taps with the window centred at `ctr` (see forward).  FIRST: all
        // three gradients of every tap are stored (those of a tap the window does not serve are
        // overwritten later -- same lane or same wave, program order); otherwise only the taps
        // that this window serves and no earlier one did are stored.
        auto run_taps = [&](auto first_tag, const Centre ctr) {
            constexpr bool FIRST = decltype(first_tag)::value != 0;
            const int cbase = (((int)ctr.y) * LW + (int)ctr.x) * 16;  // the window's shift
            // Taps in flight ahead of the one being computed.  First pass: one -- every CU streams
            // at once and the HBM is the limit.  A further pass runs on a few CUs while the others
            // saturate the memory, and with one tap ahead it exposes most of a ~1.7 us round trip per
            // tap (tools/tile_clock.py: 23-26 us per pass, 8 of them VALU): kPassDepth ahead there
            // (steep smooth flow: 507 -> 495 us; the forward's 256 registers -- two waves per SIMD --
            // hold no more than one).
            constexpr int D = FIRST ? 1 : kPassDepth;
            TapCol<ROWS> ahead[D > 1 ? D - 1 : 1];
            if constexpr (D > 1) {
                static_for<1, D>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    load_tap_col<ROWS>(ahead[i - 1], wb, ab, bb, pl, i, rf, (unsigned)x);
                });
            }
            static_for<0, 25>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                constexpr int r = t / 5, s = t % 5;
                TapCol<ROWS> nxt;
                if (t + D < 25) load_tap_col<ROWS>(nxt, wb, ab, bb, pl, t + D, rf, (unsigned)x);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < ROWS; ++j) {
                    const float wv = cur.w[j], a = cur.a[j], b = cur.b[j];
                    const int ia = (int)a, ib = (int)b;
                    const float fa = a - (float)ia, fb = b - (float)ib;
                    const float reach = fmaxf(fabsf((float)ia - ctr.y), fabsf((float)ib - ctr.x));
                    bool near = reach <= (float)R;
                    if constexpr (FIRST) {
                        fmask[j] |= near ? 0u : 1u << t;
                        if constexpr (!HEAD) mx_w = max(mx_w, absbits(wv));
                    } else {
                        near = near && ((fmask[j] >> t) & 1u);
                        fmask[j] &= near ? ~(1u << t) : ~0u;
                    }
                    const Corners q = tile_corners_at<R, TH, r * LW + s, SW>(ltile, ia, ib, rowbase[j] - cbase);
                    const float pbl = d[j][0] * q.bl.x + d[j][1] * q.bl.y + d[j][2] * q.bl.z;
                    const float ptl = d[j][0] * q.tl.x + d[j][1] * q.tl.y + d[j][2] * q.tl.z;
                    const float pbr = d[j][0] * q.br.x + d[j][1] * q.br.y + d[j][2] * q.br.z;
                    const float ptr_ = d[j][0] * q.tr.x + d[j][1] * q.tr.y + d[j][2] * q.tr.z;
                    const float dl = ptl - pbl, dr = ptr_ - pbr;
                    const float lft = pbl + fa * dl, rgt = pbr + fa * dr;
                    const float across = rgt - lft;
                    const size_t o = nb + (size_t)t * pl + rf[j];
                    float gwr = lft + fb * across;
                    float gar = wv * (dl + fb * (dr - dl));
                    float gbr = wv * across;
                    if constexpr (HEAD) {
                        // a tap this window does not serve gets its share from the pass or fix-up
                        // that computes it (a select of the sum: 0 * NaN from such a tap's
                        // unrelated texel must not enter it)
                        const float ng = gocc[j] + wv * gwr;
                        gocc[j] = near ? ng : gocc[j];
                        gwr = occv[j] * gwr;
                        gar = occv[j] * gar;
                        gbr = occv[j] * gbr;
                        asm volatile("" : "+v"(gocc[j]));
                    }
                    if (FIRST || near) {
                        if (ALL || gw) st_stream(gw + o + (unsigned)x, gwr);
                        if (ALL || goy) st_stream(goy + o + (unsigned)x, gar);
                        if (ALL || gox) st_stream(gox + o + (unsigned)x, gbr);
                    }
                    asm volatile("" : "+v"(fmask[j]));
                    if (j % kRowGroup == kRowGroup - 1) __builtin_amdgcn_sched_barrier(0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (D == 1) {
                    cur = nxt;
                } else {
                    cur = ahead[0];
#pragma unroll
                    for (int i = 0; i + 2 < D; ++i) ahead[i] = ahead[i + 1];
                    ahead[D - 2] = nxt;
                }
            });
        };

        // Rows whose pixels have at most kMaskMaxTaps unserved taps each (see forward): every lane
        // fixes its own pixels, kStepRows rows together; overwrites what the first pass stored for
        // those taps (same lane: program order).
        float *gwn = gw ? gw + nb : nullptr, *goyn = goy ? goy + nb : nullptr,
              *goxn = gox ? gox + nb : nullptr;
        auto fix_few = [&]() {
            constexpr int kStepRows = ROWS < 4 ? ROWS : 4;
#pragma unroll
            for (int j0 = 0; j0 < ROWS; j0 += kStepRows) {
                unsigned sm[kStepRows];
                size_t sr[kStepRows];
                int sy[kStepRows];
                float sd[kStepRows][3], ss[kStepRows], sg[kStepRows];
                bool some = false;
#pragma unroll
                for (int i = 0; i < kStepRows; ++i) {
                    const int j = j0 + i;
                    const bool few = __ballot(fmask[j] != 0u) && !__ballot(__popc(fmask[j]) > kMaskMaxTaps);
                    sm[i] = few ? fmask[j] : 0u;  // (few is wave-uniform)
                    if (few) fmask[j] = 0u;
                    some |= few;
                    sr[i] = rf[j];
                    sy[i] = rowy[j];
                    sd[i][0] = d[j][0], sd[i][1] = d[j][1], sd[i][2] = d[j][2];
                    ss[i] = HEAD ? occv[j] : 1.f;
                    sg[i] = 0.f;
                }
                if (some) {  // wave-uniform, rare
                    bwd_outside_steps<kStepRows>(ip, wb, ab, bb, sd, gwn, goyn, goxn, g, sm, sr, sy,
                                                 HEAD ? in_branch(x) : x, ss, sg);
                    if constexpr (HEAD) {
#pragma unroll
                        for (int i = 0; i < kStepRows; ++i) gocc[j0 + i] += sg[i];
                    }
                }
            }
        };

        Windows win = one_window(kCentred ? tile_centre<R>(ab, bb, pl, g, sx0, ty0, TH, SW) : Centre{0.f, 0.f});
        Centre ctr = win.c0;
        unsigned long long flagged[ROWS];
        bool mine = true;  // this wave's window is (re)staged in the coming pass
#pragma unroll 1
        for (int p = 0;; ++p) {
            pl = per_frame(pl);  // (see forward)
#pragma unroll
            for (int j = 0; j < ROWS; ++j) rf[j] = per_frame(rf[j]);
            // every wave is done with the previous window (see forward)
            if (HEAD && f > 0 && p == 0 && !(SYNTH_HEAD_SYNC_DEDUPE && tile_free)) __syncthreads();
            if (mine)
                fill_tile_n<R, TH, 64 * (WAVES / NWIN), SW>(tile, ip, g, ty0 + g.oy0 - R + (int)ctr.y,
                                                            sx0 + g.ox0 - R + (int)ctr.x, tid);
            __syncthreads();
            if (p == 0) {
                if (active) {
                    if (f == 0) {  // gradOutput is the same for every frame of the head
#pragma unroll
                        for (int j = 0; j < ROWS; ++j) {
                            d[j][0] = ld_stream(gb_ + rf[j] + (unsigned)x);
                            d[j][1] = ld_stream(gb_ + pl + rf[j] + (unsigned)x);
                            d[j][2] = ld_stream(gb_ + 2 * pl + rf[j] + (unsigned)x);
                            if constexpr (!HEAD)
                                mx_g = max(mx_g, max(absbits(d[j][0]), max(absbits(d[j][1]), absbits(d[j][2]))));
                        }
                    }
                    run_taps(IntTag<1>{}, ctr);
                    fix_few();
                }
            } else {
                bool any = false;
#pragma unroll
                for (int j = 0; j < ROWS; ++j) any |= flagged[j] != 0;
                if (mine && active && any) {  // wave-uniform
                    load_tap_col<ROWS>(cur, wb, ab, bb, pl, 0, rf, (unsigned)x);
                    run_taps(IntTag<0>{}, ctr);
                }
            }
#pragma unroll
            for (int j = 0; j < ROWS; ++j)
                flagged[j] = active ? __ballot(fmask[j] != 0u && owner) : 0ull;
            tile_free = false;
            if (p + 1 == PASSES) break;
            Centre next;
            if (!vote_next_window<R, WAVES, ROWS, WX, NWIN>(vote_slot, wv, active, flagged, fmask, rowy, wx0, x,
                                                            ab, bb, pl, g, win, next, mine)) {
                tile_free = true;  // (workgroup-uniform: every wave has passed the vote's barrier)
                break;
            }
            if (mine) {  // (else another window of the workgroup goes on: only its barriers matter here)
                if (p == 0) win.c1 = next;
                else win.c2 = next;
                ctr = next;
            }
        }
        if (!active) continue;  // (it has taken part in this frame's fills, barriers and votes)

        // Taps that no window served: recomputed through clamped global gathers, overwriting what the
        // first pass stored for them (same lane or same wave: program order).  Rows whose pixels have
        // few such taps each went first (fix_few, right after the first pass and again here) ...
        fix_few();
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            // ... the others pixel by pixel by the whole wave, or lane by lane when there are many
            unsigned long long todo = __ballot(fmask[j] != 0u && owner);
            if (todo) {  // wave-uniform, rare
                if (__popcll(todo) <= kCoopMaxPixels) {
                    // lane l < 25 overwrites what the pixel's OWN lane stored for tap l in the first
                    // pass: stores of different lanes of a wave are ordered only once the earlier ones
                    // have been acknowledged
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    while (todo) {
                        const int b = __ffsll((long long)todo) - 1;
                        todo &= todo - 1;
                        const float sc = HEAD ? __shfl(occv[j], b) : 1.f;
                        const float c = bwd_outside_coop(ip, wb, ab, bb, gb_, gwn, goyn, goxn, g,
                                                         (unsigned)__shfl((int)fmask[j], b), rowy[j],
                                                         wx0 + b, sc, HEAD);
                        if (HEAD && x == wx0 + b) gocc[j] += c;
                    }
                } else if (fmask[j]) {
                    const float c = bwd_outside_lane(ip, wb, ab, bb, d[j][0], d[j][1], d[j][2], gwn,
                                                     goyn, goxn, g, fmask[j], rowy[j],
                                                     HEAD ? in_branch(x) : x,
                                                     HEAD ? occv[j] : 1.f);
                    if (HEAD) gocc[j] += c;
                }
            }
            if constexpr (HEAD) {
                if (ha.gocc)
                    st_stream(ha.gocc + ((size_t)n * T + f) * pl + rf[j] + (unsigned)x, gocc[j]);
            }
        }
    }
    if constexpr (!HEAD) {
        if (maxima && active) {  // wave-uniform
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                mx_w = max(mx_w, (unsigned)__shfl_xor((int)mx_w, o));
                mx_g = max(mx_g, (unsigned)__shfl_xor((int)mx_g, o));
            }
            // only when it can raise the running maximum (same-address atomics serialise in L2)
            if (threadIdx.x == 0) {
                if (mx_w > __atomic_load_n(maxima, __ATOMIC_RELAXED)) atomicMax(maxima, mx_w);
                if (mx_g > __atomic_load_n(maxima + 1, __ATOMIC_RELAXED)) atomicMax(maxima + 1, mx_g);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// gradInput for k = 5, d = 1, C = 3, LDS-privatised.  (No reference counterpart: upstream returns
// unfilled zeros, cupy_module/synth.py:333,391.)  The adjoint of the forward scatters, per tap,
// gradOut_c * w * coef_q to the four clamped corners: 300 adds per pixel.  Straight to memory
// that is 2.5 GB of float atomics per 1080p frame (>2 ms at the chip's ~1.3 TB/s atomic rate).
// Here a workgroup accumulates its 64 x TH pixels into an LDS copy of the frame tile with
// ds_add_u64 on fixed-point integers (planar [c][ly][lx]: a wave's lanes hit consecutive banks),
// then flushes the tile with one global atomic per touched texel (~5 per pixel).  Taps displaced
// beyond the tile radius are left out of the LDS pass and sent to memory individually afterwards
// (a per-lane bit mask from the LDS pass says which, so every contribution lands exactly once).
// FIXED (deterministic mode): the flush adds the integers into the int64 image of the workspace,
// so the result does not depend on any order; otherwise each tile sum is converted to fp32 and
// added to gradInput with float atomics.  Same tap-major streaming order as the other kernels.
// ------------------------------------------------------------------------------------------

// LDS accumulation is always 64-bit fixed point: ds_add_f32 turned out to execute lane-serially
// on gfx950 (~170 cycles per wave instruction, measured: 3.1 ms per 1080p frame against 0.4 ms for
// ds_add_u64), so even the non-deterministic mode sums integers on chip and only converts to
// fp32 for the cross-workgroup adds.
template <int R, int WAVES, int ROWS, bool FIXED>
__global__ __launch_bounds__(64 * WAVES) void synth_gin_tiled(const float *__restrict__ w,
                                                              const float *__restrict__ oy,
                                                              const float *__restrict__ ox,
                                                              const float *__restrict__ gout,
                                                              float *__restrict__ gin,
                                                              unsigned char *__restrict__ ws,
                                                              Geom g, int count_bits, int guarded)
{
    if (guarded && ((const unsigned *)ws)[kGaFlagWord] < kOneBits) return;  // the gather kernel took this call
    typedef unsigned long long acc_t;
    constexpr int TH = WAVES * ROWS;
    constexpr int LW = TileShape<R, TH>::LW, LH = TileShape<R, TH>::LH;
    constexpr int CS = LH * LW;  // channel stride in the LDS tile
    __shared__ acc_t acc[3 * CS];
    const int tx0 = blockIdx.x * kTW, ty0 = blockIdx.y * TH, n = blockIdx.z;
    const int wy = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int tid = wy * 64 + threadIdx.x;
    const size_t plane = (size_t)g.H * g.W, iplane = (size_t)g.Hin * g.Win;
    const int xr = tx0 + (int)threadIdx.x;          // this lane's own column
    const bool lane_on = xr < g.W;                   // lanes past the right edge add nothing
    const int x = min(xr, g.W - 1), px = x - tx0;
    const size_t nb = (size_t)n * 25 * plane;
    const float *wb = w + nb, *ab = oy + nb, *bb = ox + nb;
    const float *gb_ = gout + (size_t)n * 3 * plane;

    // ws: header with the two maxima (both modes) followed, in FIXED mode, by the int64 image
    const int slog2 = det_scale_log2((const unsigned *)ws, count_bits);
    const float scale = ldexpf(1.f, slog2);
    acc_t *gfix = (acc_t *)((unsigned char *)ws + kDetHeaderBytes) + (size_t)n * 3 * iplane;
    float *gflt = gin + (size_t)n * 3 * iplane;

    int rowy[ROWS];
    bool rowon[ROWS];  // wave-uniform: rows past the bottom edge add nothing
    size_t roff[ROWS];
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        const int yr = ty0 + wy + WAVES * j;
        rowon[j] = yr < g.H;
        rowy[j] = min(yr, g.H - 1);
        roff[j] = (size_t)rowy[j] * g.W;
    }

    // the accumulator window follows the motion like the forward's tile (tile_centre)
    const Centre ctr = tile_centre<R>(ab, bb, plane, g, tx0, ty0, TH);
    const int cyi = (int)ctr.y, cxi = (int)ctr.x;

    TapCol<ROWS> cur;
    load_tap_col<ROWS>(cur, wb, ab, bb, plane, 0, roff, (unsigned)x);
    for (int i = tid; i < 3 * CS; i += 64 * WAVES) acc[i] = 0;
    __syncthreads();

    float d[ROWS][3];
    unsigned fm[ROWS];  // bit t: tap t of this lane's pixel leaves the window (goes to memory below)
    int rowbase[ROWS];
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        const float on = (lane_on && rowon[j]) ? scale : 0.f;  // masks edge duplicates
        d[j][0] = ld_stream(gb_ + roff[j] + (unsigned)x) * on;
        d[j][1] = ld_stream(gb_ + plane + roff[j] + (unsigned)x) * on;
        d[j][2] = ld_stream(gb_ + 2 * plane + roff[j] + (unsigned)x) * on;
        fm[j] = 0u;
        rowbase[j] = (rowy[j] - ty0 + R - cyi) * LW + px + R - cxi;
    }
    static_for<0, 25>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        constexpr int r = t / 5, s = t % 5;
        TapCol<ROWS> nxt;
        if (t < 24) load_tap_col<ROWS>(nxt, wb, ab, bb, plane, t + 1, roff, (unsigned)x);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            const float wv = cur.w[j], a = cur.a[j], b = cur.b[j];
            const int ia = (int)a, ib = (int)b;
            const float fa = a - (float)ia, fb = b - (float)ib;
            const float reach = fmaxf(fabsf((float)ia - ctr.y), fabsf((float)ib - ctr.x));
            fm[j] |= reach <= (float)R ? 0u : 1u << t;
            if (reach <= (float)R) {  // footprint inside the window
                acc_t *p = acc + (__mul24(ia, LW) + ib + rowbase[j] + (r * LW + s));
                const float ga = 1.f - fa, gb = 1.f - fb;
                const float c00 = ga * gb, c10 = fa * gb, c01 = ga * fb, c11 = fa * fb;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float dw = d[j][c] * wv;
                    atomicAdd(p + c * CS, to_fixed(dw * c00));  // ds_add_u64
                    atomicAdd(p + c * CS + LW, to_fixed(dw * c10));
                    atomicAdd(p + c * CS + 1, to_fixed(dw * c01));
                    atomicAdd(p + c * CS + LW + 1, to_fixed(dw * c11));
                }
            }
            asm volatile("" : "+v"(fm[j]));
        }
        __builtin_amdgcn_sched_barrier(0);
        cur = nxt;
    });
    __syncthreads();

    // flush the tile: texel (ly, lx) is frame pixel (clamp(gy0 + ly), clamp(gx0 + lx))
    const int gy0 = ty0 + g.oy0 - R + cyi, gx0 = tx0 + g.ox0 - R + cxi;
    for (int i = tid; i < CS; i += 64 * WAVES) {
        const int ly = i / LW, lx = i - ly * LW;
        const int gy = clampi(gy0 + ly, 0, g.Hin - 1), gx = clampi(gx0 + lx, 0, g.Win - 1);
        const size_t o = (size_t)gy * g.Win + gx;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const acc_t v = acc[c * CS + i];
            if (v != 0) {
                if constexpr (FIXED)
                    atomicAdd(gfix + c * iplane + o, v);
                else
                    atomicAdd(gflt + c * iplane + o, (float)ldexp((double)(long long)v, -slog2));
            }
        }
    }

    // taps that left the window: straight to memory.  Every lane walks the set bits of its own mask,
    // all lanes of the row together: as many steps as the worst pixel has such taps (sigma = 4 px: one
    // or two; the first version walked all 25 taps of any pixel that had one: 25 dependent round trips)
#pragma unroll
    for (int j = 0; j < ROWS; ++j) {
        unsigned m = (lane_on && rowon[j]) ? fm[j] : 0u;
        const int y = rowy[j];
        const size_t pix = roff[j] + x;
        while (__ballot(m != 0u)) {  // wave-uniform
            if (m) {
                const int t = __ffs((int)m) - 1;
                m &= m - 1;
                const size_t o = (size_t)t * plane + pix;
                const float a = ab[o], b = bb[o], wv = wb[o];
                const int r = t / 5, s2 = t - 5 * r;
                const TapIdx ti = tap_index(g, a, b, y + g.oy0 + r, x + g.ox0 + s2);
                const float ga = 1.f - ti.fa, gb = 1.f - ti.fb;
                const float c00 = ga * gb, c10 = ti.fa * gb, c01 = ga * ti.fb, c11 = ti.fa * ti.fb;
#pragma unroll
                for (int c = 0; c < 3; ++c) {
                    const float dw = d[j][c] * wv;  // d carries the fixed-point scale
                    if constexpr (FIXED) {
                        acc_t *p = gfix + c * iplane;
                        atomicAdd(p + ti.bl, to_fixed(dw * c00));
                        atomicAdd(p + ti.tl, to_fixed(dw * c10));
                        atomicAdd(p + ti.br, to_fixed(dw * c01));
                        atomicAdd(p + ti.tr, to_fixed(dw * c11));
                    } else {
                        float *p = gflt + c * iplane;
                        const float un = dw * ldexpf(1.f, -slog2);
                        atomicAdd(p + ti.bl, un * c00);
                        atomicAdd(p + ti.tl, un * c10);
                        atomicAdd(p + ti.br, un * c01);
                        atomicAdd(p + ti.tr, un * c11);
                    }
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void offset_radius_kernel(const float *__restrict__ oy,
                                                             const float *__restrict__ ox,
                                                             size_t count, int *__restrict__ out)
{
    int m = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (size_t)gridDim.x * blockDim.x) {
        const int ia = (int)oy[i], ib = (int)ox[i];
        // |INT_MIN| does not fit: saturate
        const int aa = ia == INT32_MIN ? INT32_MAX : abs(ia);
        const int ab = ib == INT32_MIN ? INT32_MAX : abs(ib);
        m = max(m, max(aa, ab));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = max(m, __shfl_xor(m, o));
    // only when it can raise the running maximum (same-address atomics serialise, see absmax2_kernel)
    if ((threadIdx.x & 63) == 0 && m > __atomic_load_n(out, __ATOMIC_RELAXED)) atomicMax(out, m);
}

}  // namespace
#endif  // ARTEMIS_SYNTH_DEVICE_HPP
