// synth_kernels.hip -- C ABI, host logic and direct (any k, d, C) kernels of libartemis_synth.so.
// Device code: synth_device.hpp; the LDS-staged kernels are instantiated per tile radius in
// synth_tiles.hip.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <atomic>

#include "synth_device.hpp"

namespace asynth {
static std::atomic<long long> g_small_grid_tiles{768};
long long small_grid_tiles() { return g_small_grid_tiles.load(std::memory_order_relaxed); }
}  // namespace asynth

namespace {

using asynth::launch_bwd_tiled;
using asynth::launch_fwd_tiled;
using asynth::launch_gin_tiled;
using asynth::launch_head_bwd;
using asynth::launch_head_fwd;

// ------------------------------------------------------------------------------------------
// Host side of the C ABI
// ------------------------------------------------------------------------------------------

thread_local char g_last_error[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
    return code;
}

int check_launch(const char *what)
{
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, "%s: %s", what, hipGetErrorString(e));
    return 0;
}

// 0 ok, > 0 invalid, -1 empty problem.  Frame (in_H, in_W) and origin (org_y, org_x): the frame row
// read by tap r of output row y is clamp(y + org_y + r*d + trunc(offset), 0, in_H - 1).
int make_geom_ex(Geom &g, int N, int C, int H, int W, int k, int d, long long in_H, long long in_W,
                 int org_y, int org_x)
{
    if (N < 0 || C < 0 || H < 0 || W < 0)
        return fail(hipErrorInvalidValue, "negative dimension N=%d C=%d H=%d W=%d", N, C, H, W);
    if (k < 1 || d < 1)
        return fail(hipErrorInvalidValue, "filter size k=%d and dilation d=%d must be >= 1", k, d);
    if (N > 65535) return fail(hipErrorInvalidValue, "N=%d exceeds 65535 (grid.z)", N);
    const long long halo = (long long)(k - 1) * d;
    if (halo > (1 << 20) || (long long)H * W >= (1LL << 31) || in_H * in_W >= (1LL << 31) ||
        in_H >= (1LL << 31) || in_W >= (1LL << 31) || (long long)k * k >= (1 << 20))
        return fail(hipErrorInvalidValue,
                    "plane too large: H=%d W=%d k=%d d=%d frame %lld x %lld (each plane must hold "
                    "< 2^31 elements)", H, W, k, d, in_H, in_W);
    if (in_H < 0 || in_W < 0) return fail(hipErrorInvalidValue, "negative frame size");
    if (org_y < -(1 << 24) || org_y > (1 << 24) || org_x < -(1 << 24) || org_x > (1 << 24))
        return fail(hipErrorInvalidValue, "frame origin (%d, %d) out of range", org_y, org_x);
    g.N = N; g.C = C; g.H = H; g.W = W; g.k = k; g.d = d;
    g.Hin = (int)in_H; g.Win = (int)in_W; g.oy0 = org_y; g.ox0 = org_x;
    if (N == 0 || C == 0 || H == 0 || W == 0) return -1;
    if (in_H == 0 || in_W == 0)
        return fail(hipErrorInvalidValue, "empty frame (%lld x %lld) with a non-empty output", in_H, in_W);
    return 0;
}

// the reference layout: frame padded by (k-1)*d, origin 0 (cupy_module/synth.py:277-278)
int make_geom(Geom &g, int N, int C, int H, int W, int k, int d)
{
    const long long halo = (k >= 1 && d >= 1) ? (long long)(k - 1) * d : 0;
    return make_geom_ex(g, N, C, H, W, k, d, H + halo, W + halo, 0, 0);
}

// the largest built radius (two 64-column windows of R = 16 would not fit the 160 KB of LDS)
#ifndef SYNTH_DEFAULT_RADIUS
#define SYNTH_DEFAULT_RADIUS 14
#endif

constexpr int kDefaultRadius = SYNTH_DEFAULT_RADIUS;

int pick_radius(int requested)
{
    if (requested == ARTEMIS_SYNTH_RADIUS_NONE) return 0;
    if (requested < 0) requested = kDefaultRadius;
    if (requested <= 2) return 2;
    if (requested <= 4) return 4;
    if (requested <= 8) return 8;
    if (requested <= 14) return 14;
    return 24;
}

// gradInput's 64-bit accumulator window does not fit the LDS beyond R = 14
int gin_radius(int requested)
{
    const int R = pick_radius(requested);
    return R > 14 ? 14 : R;
}

bool tiled_ok(const Geom &g) { return g.k == 5 && g.d == 1 && g.C == 3; }

template <int KS>
void launch_fwd_direct(int ch, dim3 grid, dim3 block, hipStream_t st, const float *in, const float *w,
                       const float *oy, const float *ox, float *out, const Geom &g, int c0)
{
    switch (ch) {
    case 1: hipLaunchKernelGGL((synth_fwd_direct<1, KS>), grid, block, 0, st, in, w, oy, ox, out, g, c0); break;
    case 2: hipLaunchKernelGGL((synth_fwd_direct<2, KS>), grid, block, 0, st, in, w, oy, ox, out, g, c0); break;
    case 3: hipLaunchKernelGGL((synth_fwd_direct<3, KS>), grid, block, 0, st, in, w, oy, ox, out, g, c0); break;
    default: hipLaunchKernelGGL((synth_fwd_direct<4, KS>), grid, block, 0, st, in, w, oy, ox, out, g, c0); break;
    }
}

// Workspace: deterministic mode = header + int64 image; fp32-atomic mode = header only, and only
// when the LDS-staged kernel (which scales to fixed point on chip) will run.
size_t ws_bytes_needed(const Geom &g, int gin_mode, int R)
{
    if (gin_mode == ARTEMIS_SYNTH_GIN_DETERMINISTIC)
        return kDetHeaderBytes + (size_t)g.N * g.C * g.Hin * g.Win * sizeof(long long);
    return R > 0 ? kDetHeaderBytes : 0;
}

int count_bits_for(const Geom &g)
{
    // contributions that can land on one gradInput element: every tap corner of its plane
    unsigned long long cnt = (unsigned long long)g.H * g.W * g.k * g.k * 4ull;
    int bits = 1;
    while ((1ull << bits) < cnt && bits < 60) ++bits;
    return bits;
}

int reduce_blocks(size_t count)
{
    const size_t b = (count + 255) / 256;
    return (int)(b < 2048 ? (b ? b : 1) : 2048);
}

// frame_dtype checks shared by the entry points that accept bf16 frames
int check_frame_dtype(const Geom &g, int frame_dtype, const char *what)
{
    if (frame_dtype != ARTEMIS_SYNTH_FRAME_F32 && frame_dtype != ARTEMIS_SYNTH_FRAME_BF16)
        return fail(hipErrorInvalidValue, "%s: unknown frame_dtype %d", what, frame_dtype);
    if (frame_dtype == ARTEMIS_SYNTH_FRAME_BF16 && !tiled_ok(g))
        return fail(hipErrorInvalidValue,
                    "%s: bf16 frames are built for k = 5, d = 1, C = 3 only (got k=%d d=%d C=%d)", what,
                    g.k, g.d, g.C);
    return 0;
}

// radius of the LDS-staged kernel to run, 0 = direct kernels (fp32 frames only)
int tiled_radius(const Geom &g, int tile_radius, int frame_dtype)
{
    if (!tiled_ok(g)) return 0;
    int R = pick_radius(tile_radius);
    if (R == 0 && frame_dtype == ARTEMIS_SYNTH_FRAME_BF16) R = pick_radius(ARTEMIS_SYNTH_RADIUS_DEFAULT);
    if (R < 8 && frame_dtype == ARTEMIS_SYNTH_FRAME_BF16) R = 8;  // (smallest radius built for bf16)
    return R;
}

// bf16 frames are built for radii 8, 14 and 24 only (two objects fewer to compile; a small radius
// buys a bf16 caller nothing): their smaller requests run the radius-8 kernels.
template <class FT>
constexpr bool kSmallRadiiBuilt = std::is_same<FT, float>::value;

template <class FT>
void dispatch_fwd_tiled(int R, const void *input, const float *weight, const float *offset_y,
                               const float *offset_x, float *output, const Geom &g, hipStream_t st)
{
    const FT *in = (const FT *)input;
    switch (R) {
    case 2:
        if constexpr (kSmallRadiiBuilt<FT>) {
            launch_fwd_tiled<2, FT>(in, weight, offset_y, offset_x, output, g, st);
            break;
        }
        [[fallthrough]];
    case 4:
        if constexpr (kSmallRadiiBuilt<FT>) {
            launch_fwd_tiled<4, FT>(in, weight, offset_y, offset_x, output, g, st);
            break;
        }
        [[fallthrough]];
    case 8: launch_fwd_tiled<8, FT>(in, weight, offset_y, offset_x, output, g, st); break;
    case 24: launch_fwd_tiled<24, FT>(in, weight, offset_y, offset_x, output, g, st); break;
    default: launch_fwd_tiled<14, FT>(in, weight, offset_y, offset_x, output, g, st); break;
    }
}

template <class FT>
void dispatch_bwd_tiled(int R, const void *input, const float *weight, const float *offset_y,
                               const float *offset_x, const float *grad_output, float *gw, float *goy,
                               float *gox, const Geom &g, hipStream_t st)
{
    const FT *in = (const FT *)input;
    switch (R) {
    case 2:
        if constexpr (kSmallRadiiBuilt<FT>) {
            launch_bwd_tiled<2, FT>(in, weight, offset_y, offset_x, grad_output, gw, goy, gox, g, st);
            break;
        }
        [[fallthrough]];
    case 4:
        if constexpr (kSmallRadiiBuilt<FT>) {
            launch_bwd_tiled<4, FT>(in, weight, offset_y, offset_x, grad_output, gw, goy, gox, g, st);
            break;
        }
        [[fallthrough]];
    case 8: launch_bwd_tiled<8, FT>(in, weight, offset_y, offset_x, grad_output, gw, goy, gox, g, st); break;
    case 24: launch_bwd_tiled<24, FT>(in, weight, offset_y, offset_x, grad_output, gw, goy, gox, g, st); break;
    default: launch_bwd_tiled<14, FT>(in, weight, offset_y, offset_x, grad_output, gw, goy, gox, g, st); break;
    }
}

template <class FT>
void dispatch_head_fwd(int R, const float *w, const float *oy, const float *ox, float *out,
                              const Geom &g, const HeadArgs &ha, hipStream_t st)
{
    switch (R) {
    case 2:
        if constexpr (kSmallRadiiBuilt<FT>) {
            launch_head_fwd<2, FT>(w, oy, ox, out, g, ha, st);
            break;
        }
        [[fallthrough]];
    case 4:
        if constexpr (kSmallRadiiBuilt<FT>) {
            launch_head_fwd<4, FT>(w, oy, ox, out, g, ha, st);
            break;
        }
        [[fallthrough]];
    case 8: launch_head_fwd<8, FT>(w, oy, ox, out, g, ha, st); break;
    case 24: launch_head_fwd<24, FT>(w, oy, ox, out, g, ha, st); break;
    default: launch_head_fwd<14, FT>(w, oy, ox, out, g, ha, st); break;
    }
}

template <class FT>
void dispatch_head_bwd(int R, const float *w, const float *oy, const float *ox,
                              const float *gout, float *gw, float *goy, float *gox, const Geom &g,
                              const HeadArgs &ha, hipStream_t st)
{
    switch (R) {
    case 2:
        if constexpr (kSmallRadiiBuilt<FT>) {
            launch_head_bwd<2, FT>(w, oy, ox, gout, gw, goy, gox, g, ha, st);
            break;
        }
        [[fallthrough]];
    case 4:
        if constexpr (kSmallRadiiBuilt<FT>) {
            launch_head_bwd<4, FT>(w, oy, ox, gout, gw, goy, gox, g, ha, st);
            break;
        }
        [[fallthrough]];
    case 8: launch_head_bwd<8, FT>(w, oy, ox, gout, gw, goy, gox, g, ha, st); break;
    case 24: launch_head_bwd<24, FT>(w, oy, ox, gout, gw, goy, gox, g, ha, st); break;
    default: launch_head_bwd<14, FT>(w, oy, ox, gout, gw, goy, gox, g, ha, st); break;
    }
}

}  // namespace

extern "C" {

const char *artemis_synth_version(void)
{
    return "artemis_synth abi " "1" " gfx950 hip (fwd, fused bwd, grad_input atomic+deterministic, fused head, bf16 frames)";
}

const char *artemis_synth_last_error(void) { return g_last_error; }

static int run_forward(const Geom &g, int rc, const void *input_any, const float *weight,
                       const float *offset_y, const float *offset_x, float *output, int tile_radius,
                       int frame_dtype, void *stream)
{
    if (rc > 0) return rc;
    if (const int e = check_frame_dtype(g, frame_dtype, "artemis_synth_forward")) return e;
    if (rc < 0) return 0;
    if (!input_any || !weight || !offset_y || !offset_x || !output)
        return fail(hipErrorInvalidValue, "artemis_synth_forward: NULL tensor pointer");
    hipStream_t st = (hipStream_t)stream;
    const int R = tiled_radius(g, tile_radius, frame_dtype);
    if (R > 0) {
        if (frame_dtype == ARTEMIS_SYNTH_FRAME_BF16)
            dispatch_fwd_tiled<asynth::bf16_t>(R, input_any, weight, offset_y, offset_x, output, g, st);
        else
            dispatch_fwd_tiled<float>(R, input_any, weight, offset_y, offset_x, output, g, st);
    } else {
        const float *input = (const float *)input_any;
        for (int c0 = 0; c0 < g.C; c0 += 4) {
            const int ch = g.C - c0 < 4 ? g.C - c0 : 4;
            const dim3 grid = pixel_grid(g, 4), block(64, 4);
            if (g.k == 5) launch_fwd_direct<5>(ch, grid, block, st, input, weight, offset_y, offset_x, output, g, c0);
            else launch_fwd_direct<0>(ch, grid, block, st, input, weight, offset_y, offset_x, output, g, c0);
        }
    }
    return check_launch("artemis_synth_forward");
}

int artemis_synth_forward(const float *input, const float *weight, const float *offset_y,
                          const float *offset_x, float *output, int N, int C, int H, int W, int k,
                          int d, int tile_radius, void *stream)
{
    Geom g;
    const int rc = make_geom(g, N, C, H, W, k, d);
    return run_forward(g, rc, input, weight, offset_y, offset_x, output, tile_radius,
                       ARTEMIS_SYNTH_FRAME_F32, stream);
}

int artemis_synth_forward_ex(const void *input, const float *weight, const float *offset_y,
                             const float *offset_x, float *output, int N, int C, int H, int W, int k,
                             int d, int in_H, int in_W, int org_y, int org_x, int frame_dtype,
                             int tile_radius, void *stream)
{
    Geom g;
    const int rc = make_geom_ex(g, N, C, H, W, k, d, in_H, in_W, org_y, org_x);
    return run_forward(g, rc, input, weight, offset_y, offset_x, output, tile_radius, frame_dtype,
                       stream);
}

size_t artemis_synth_backward_workspace_bytes(int N, int C, int H, int W, int k, int d,
                                              int gin_mode)
{
    Geom g;
    if (make_geom(g, N, C, H, W, k, d) != 0) return 0;
    // radius-independent upper bound: any tile_radius may be passed to the backward call
    return ws_bytes_needed(g, gin_mode, tiled_ok(g) ? 1 : 0);
}

size_t artemis_synth_backward_workspace_bytes_ex(int N, int C, int H, int W, int k, int d, int in_H,
                                                 int in_W, int gin_mode)
{
    Geom g;
    if (make_geom_ex(g, N, C, H, W, k, d, in_H, in_W, 0, 0) != 0) return 0;
    return ws_bytes_needed(g, gin_mode, tiled_ok(g) ? 1 : 0);
}

static int run_backward(const Geom &g, int rc, const void *input_any, const float *weight,
                        const float *offset_y, const float *offset_x, const float *grad_output,
                        float *grad_input, float *grad_weight, float *grad_offset_y,
                        float *grad_offset_x, int gin_mode, int tile_radius, int frame_dtype,
                        void *workspace, size_t workspace_bytes, void *stream)
{
    if (rc > 0) return rc;
    if (const int e = check_frame_dtype(g, frame_dtype, "artemis_synth_backward")) return e;
    const float *input = (const float *)input_any;
    const int N = g.N, C = g.C, H = g.H, W = g.W, k = g.k;
    if (gin_mode != ARTEMIS_SYNTH_GIN_ATOMIC && gin_mode != ARTEMIS_SYNTH_GIN_DETERMINISTIC)
        return fail(hipErrorInvalidValue, "artemis_synth_backward: unknown gin_mode %d", gin_mode);
    hipStream_t st = (hipStream_t)stream;
    const size_t gin_count = (size_t)N * C * g.Hin * g.Win;
    if (rc < 0) {
        // no output pixels: gradInput (which may still have elements) is all zero
        if (grad_input && gin_count) {
            const hipError_t e = hipMemsetAsync(grad_input, 0, gin_count * sizeof(float), st);
            if (e != hipSuccess) return fail((int)e, "hipMemsetAsync: %s", hipGetErrorString(e));
        }
        return 0;
    }
    if (!input || !weight || !offset_y || !offset_x || !grad_output)
        return fail(hipErrorInvalidValue, "artemis_synth_backward: NULL tensor pointer");

    if (grad_weight || grad_offset_y || grad_offset_x) {
        const int R = tiled_radius(g, tile_radius, frame_dtype);
        if (R > 0) {
            if (frame_dtype == ARTEMIS_SYNTH_FRAME_BF16)
                dispatch_bwd_tiled<asynth::bf16_t>(R, input_any, weight, offset_y, offset_x, grad_output,
                                                   grad_weight, grad_offset_y, grad_offset_x, g, st);
            else
                dispatch_bwd_tiled<float>(R, input_any, weight, offset_y, offset_x, grad_output,
                                          grad_weight, grad_offset_y, grad_offset_x, g, st);
        } else {
            if (k == 5 && C == 3)
                hipLaunchKernelGGL((synth_bwd_direct<3, 5>), pixel_grid(g, 4), dim3(64, 4), 0, st, input,
                                   weight, offset_y, offset_x, grad_output, grad_weight,
                                   grad_offset_y, grad_offset_x, g);
            else
                hipLaunchKernelGGL((synth_bwd_direct<0, 0>), pixel_grid(g, 4), dim3(64, 4), 0, st, input,
                                   weight, offset_y, offset_x, grad_output, grad_weight,
                                   grad_offset_y, grad_offset_x, g);
        }
        const int e = check_launch("artemis_synth_backward (taps)");
        if (e) return e;
    }

    if (grad_input) {
        const int Rg = tiled_ok(g) ? gin_radius(tile_radius) : 0;
        const int bits = count_bits_for(g);
        const bool fixed = gin_mode == ARTEMIS_SYNTH_GIN_DETERMINISTIC;
        const size_t need = ws_bytes_needed(g, gin_mode, Rg);
        if (need && (!workspace || workspace_bytes < need))
            return fail(hipErrorInvalidValue,
                        "artemis_synth_backward: grad_input needs %zu workspace bytes "
                        "(artemis_synth_backward_workspace_bytes), got %zu", need,
                        workspace ? workspace_bytes : (size_t)0);
        if (need && ((uintptr_t)workspace & 15) != 0)
            return fail(hipErrorInvalidValue, "artemis_synth_backward: workspace not 16-byte aligned");
        unsigned char *ws = (unsigned char *)workspace;
        hipError_t e = hipSuccess;
        if (need) e = hipMemsetAsync(ws, 0, need, st);
        if (e == hipSuccess && !fixed) e = hipMemsetAsync(grad_input, 0, gin_count * sizeof(float), st);
        if (e != hipSuccess) return fail((int)e, "hipMemsetAsync: %s", hipGetErrorString(e));
        if (need) {  // maxima for the fixed-point scale
            const size_t wcount = (size_t)N * k * k * H * W, gcount = (size_t)N * C * H * W;
            // one launch for both maxima; ~16 KB of 16-byte loads per block and pass
            const int vec = (((uintptr_t)weight | (uintptr_t)grad_output) & 15) == 0;
            const size_t per_block = 256 * 4 * 4 * 4;  // elements a block covers per unrolled pass
            auto blocks_for = [&](size_t n) {
                const size_t b = (n + per_block - 1) / per_block;
                return (unsigned)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
            };
            const unsigned ba = blocks_for(wcount), bb = blocks_for(gcount);
            hipLaunchKernelGGL(absmax2_kernel, dim3(ba + bb), dim3(256), 0, st, weight, wcount,
                               grad_output, gcount, ba, vec, (unsigned *)ws);
        }
        if (Rg > 0) {
            switch (Rg * 2 + (fixed ? 1 : 0)) {
            case 4: launch_gin_tiled<2, false>(weight, offset_y, offset_x, grad_output, grad_input, ws, g, bits, st); break;
            case 5: launch_gin_tiled<2, true>(weight, offset_y, offset_x, grad_output, grad_input, ws, g, bits, st); break;
            case 8: launch_gin_tiled<4, false>(weight, offset_y, offset_x, grad_output, grad_input, ws, g, bits, st); break;
            case 9: launch_gin_tiled<4, true>(weight, offset_y, offset_x, grad_output, grad_input, ws, g, bits, st); break;
            case 16: launch_gin_tiled<8, false>(weight, offset_y, offset_x, grad_output, grad_input, ws, g, bits, st); break;
            case 17: launch_gin_tiled<8, true>(weight, offset_y, offset_x, grad_output, grad_input, ws, g, bits, st); break;
            case 28: launch_gin_tiled<14, false>(weight, offset_y, offset_x, grad_output, grad_input, ws, g, bits, st); break;
            default: launch_gin_tiled<14, true>(weight, offset_y, offset_x, grad_output, grad_input, ws, g, bits, st); break;
            }
        } else if (!fixed) {
            if (k == 5 && C == 3)
                hipLaunchKernelGGL((synth_gin_atomic_direct<3, 5>), pixel_grid(g, 4), dim3(64, 4), 0, st,
                                   weight, offset_y, offset_x, grad_output, grad_input, g);
            else
                hipLaunchKernelGGL((synth_gin_atomic_direct<0, 0>), pixel_grid(g, 4), dim3(64, 4), 0, st,
                                   weight, offset_y, offset_x, grad_output, grad_input, g);
        } else {
            if (k == 5 && C == 3)
                hipLaunchKernelGGL((synth_gin_fixed_direct<3, 5>), pixel_grid(g, 4), dim3(64, 4), 0, st,
                                   weight, offset_y, offset_x, grad_output, ws, g, bits);
            else
                hipLaunchKernelGGL((synth_gin_fixed_direct<0, 0>), pixel_grid(g, 4), dim3(64, 4), 0, st,
                                   weight, offset_y, offset_x, grad_output, ws, g, bits);
        }
        if (fixed)
            hipLaunchKernelGGL(det_finalize_kernel, dim3(reduce_blocks(gin_count)), dim3(256), 0, st,
                               ws, grad_input, gin_count, bits);
        const int e2 = check_launch("artemis_synth_backward (grad_input)");
        if (e2) return e2;
    }
    return 0;
}

int artemis_synth_backward(const float *input, const float *weight, const float *offset_y,
                           const float *offset_x, const float *grad_output, float *grad_input,
                           float *grad_weight, float *grad_offset_y, float *grad_offset_x, int N,
                           int C, int H, int W, int k, int d, int gin_mode, int tile_radius,
                           void *workspace, size_t workspace_bytes, void *stream)
{
    Geom g;
    const int rc = make_geom(g, N, C, H, W, k, d);
    return run_backward(g, rc, input, weight, offset_y, offset_x, grad_output, grad_input,
                        grad_weight, grad_offset_y, grad_offset_x, gin_mode, tile_radius,
                        ARTEMIS_SYNTH_FRAME_F32, workspace, workspace_bytes, stream);
}

int artemis_synth_backward_ex(const void *input, const float *weight, const float *offset_y,
                              const float *offset_x, const float *grad_output, float *grad_input,
                              float *grad_weight, float *grad_offset_y, float *grad_offset_x, int N,
                              int C, int H, int W, int k, int d, int in_H, int in_W, int org_y,
                              int org_x, int frame_dtype, int gin_mode, int tile_radius,
                              void *workspace, size_t workspace_bytes, void *stream)
{
    Geom g;
    const int rc = make_geom_ex(g, N, C, H, W, k, d, in_H, in_W, org_y, org_x);
    return run_backward(g, rc, input, weight, offset_y, offset_x, grad_output, grad_input,
                        grad_weight, grad_offset_y, grad_offset_x, gin_mode, tile_radius, frame_dtype,
                        workspace, workspace_bytes, stream);
}

// ---- allocation-free backward ----------------------------------------------------------------
// The workspace of artemis_synth_backward_ex (fixed-point header + int64 image of gradInput) held
// inside the library.  `scratch` is a HOST-side stand-in for the device array: host code never
// passes it anywhere; it resolves each device's own copy once (hipGetSymbolAddress with that device
// current) and uses that pointer.
// (`used`: no kernel names it -- the kernels take the resolved pointer as an argument -- so it must be
// kept in the code object explicitly)
static __device__ __align__(256) __attribute__((used)) unsigned char
    scratch[ARTEMIS_SYNTH_NOALLOC_SCRATCH_BYTES];

namespace {
constexpr int kMaxDevices = 64;
std::atomic<void *> g_scratch_ptr[kMaxDevices];  // per device; racing first calls store one value

int scratch_for_current_device(void **out)
{
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return fail((int)e, "hipGetDevice: %s", hipGetErrorString(e));
    if (dev < 0 || dev >= kMaxDevices)
        return fail(hipErrorInvalidValue, "artemis_synth_backward_noalloc: device %d not supported "
                    "(0..%d)", dev, kMaxDevices - 1);
    void *p = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (!p) {
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || !p)
            return fail(e != hipSuccess ? (int)e : (int)hipErrorInvalidSymbol,
                        "hipGetSymbolAddress(scratch) on device %d: %s", dev, hipGetErrorString(e));
        g_scratch_ptr[dev].store(p, std::memory_order_release);
    }
    *out = p;
    return 0;
}
}  // namespace

int artemis_synth_backward_noalloc(const void *input, const float *weight, const float *offset_y,
                                   const float *offset_x, const float *grad_output,
                                   float *grad_input, float *grad_weight, float *grad_offset_y,
                                   float *grad_offset_x, int N, int C, int H, int W, int k, int d,
                                   int in_H, int in_W, int org_y, int org_x, int frame_dtype,
                                   int gin_mode, int tile_radius, void *stream)
{
    Geom g;
    const int rc = make_geom_ex(g, N, C, H, W, k, d, in_H, in_W, org_y, org_x);
    void *ws = nullptr;
    size_t ws_bytes = 0;
    if (rc == 0 && grad_input &&
        (gin_mode == ARTEMIS_SYNTH_GIN_ATOMIC || gin_mode == ARTEMIS_SYNTH_GIN_DETERMINISTIC)) {
        // the bound that holds for every tile_radius, as the size query reports it
        const size_t need = ws_bytes_needed(g, gin_mode, tiled_ok(g) ? 1 : 0);
        if (need > (size_t)ARTEMIS_SYNTH_NOALLOC_SCRATCH_BYTES)
            return fail(hipErrorInvalidValue,
                        "artemis_synth_backward_noalloc: grad_input of %lld elements in this gin_mode "
                        "needs %zu scratch bytes, the built-in scratch holds %zu "
                        "(ARTEMIS_SYNTH_NOALLOC_SCRATCH_BYTES); use artemis_synth_backward_ex with a "
                        "caller workspace", (long long)N * C * in_H * in_W, need,
                        (size_t)ARTEMIS_SYNTH_NOALLOC_SCRATCH_BYTES);
        if (need) {
            if (const int e = scratch_for_current_device(&ws)) return e;
            ws_bytes = (size_t)ARTEMIS_SYNTH_NOALLOC_SCRATCH_BYTES;
        }
    }
    return run_backward(g, rc, input, weight, offset_y, offset_x, grad_output, grad_input,
                        grad_weight, grad_offset_y, grad_offset_x, gin_mode, tile_radius, frame_dtype,
                        ws, ws_bytes, stream);
}

// ---- fused ChronoSynth head ----------------------------------------------------------------
static int head_setup(Geom &g, HeadArgs &ha, const char *what, const void *const *frames, int T,
                      const float *occlusion, int B, int H, int W, int in_H, int in_W, int org_y,
                      int org_x, int frame_dtype, int tile_radius, int &R)
{
    if (T < 1 || T > kMaxFrames)
        return fail(hipErrorInvalidValue, "%s: T=%d context frames (1..%d supported)", what, T,
                    kMaxFrames);
    const int rc = make_geom_ex(g, B, 3, H, W, 5, 1, in_H, in_W, org_y, org_x);
    if (rc != 0) return rc;
    if ((long long)B * T > 65535LL * 64)
        return fail(hipErrorInvalidValue, "%s: B*T too large", what);
    if (!frames || !occlusion) return fail(hipErrorInvalidValue, "%s: NULL tensor pointer", what);
    ha = HeadArgs{};
    for (int f = 0; f < T; ++f) {
        if (!frames[f]) return fail(hipErrorInvalidValue, "%s: frames[%d] is NULL", what, f);
        ha.frame[f] = frames[f];
    }
    ha.occ = occlusion;
    ha.T = T;
    if (const int e = check_frame_dtype(g, frame_dtype, what)) return e;
    R = pick_radius(tile_radius);
    if (R == 0) R = pick_radius(ARTEMIS_SYNTH_RADIUS_DEFAULT);  // the head has no direct variant
    if (R < 8 && frame_dtype == ARTEMIS_SYNTH_FRAME_BF16) R = 8;  // (smallest radius built for bf16)
    return 0;
}

int artemis_synth_head_forward(const void *const *frames, int T, const float *weights,
                               const float *offsets_y, const float *offsets_x,
                               const float *occlusion, float *output, int B, int H, int W, int in_H,
                               int in_W, int org_y, int org_x, int frame_dtype, int tile_radius,
                               void *stream)
{
    Geom g;
    HeadArgs ha;
    int R = 0;
    const int rc = head_setup(g, ha, "artemis_synth_head_forward", frames, T, occlusion, B, H, W, in_H,
                              in_W, org_y, org_x, frame_dtype, tile_radius, R);
    if (rc > 0) return rc;
    if (rc < 0) return 0;
    if (!weights || !offsets_y || !offsets_x || !output)
        return fail(hipErrorInvalidValue, "artemis_synth_head_forward: NULL tensor pointer");
    hipStream_t st = (hipStream_t)stream;
    if (frame_dtype == ARTEMIS_SYNTH_FRAME_BF16)
        dispatch_head_fwd<asynth::bf16_t>(R, weights, offsets_y, offsets_x, output, g, ha, st);
    else
        dispatch_head_fwd<float>(R, weights, offsets_y, offsets_x, output, g, ha, st);
    return check_launch("artemis_synth_head_forward");
}

int artemis_synth_head_backward(const void *const *frames, int T, const float *weights,
                                const float *offsets_y, const float *offsets_x,
                                const float *occlusion, const float *grad_output,
                                float *grad_weights, float *grad_offsets_y, float *grad_offsets_x,
                                float *grad_occlusion, int B, int H, int W, int in_H, int in_W,
                                int org_y, int org_x, int frame_dtype, int tile_radius, void *stream)
{
    Geom g;
    HeadArgs ha;
    int R = 0;
    const int rc = head_setup(g, ha, "artemis_synth_head_backward", frames, T, occlusion, B, H, W,
                              in_H, in_W, org_y, org_x, frame_dtype, tile_radius, R);
    if (rc > 0) return rc;
    if (rc < 0) return 0;
    if (!weights || !offsets_y || !offsets_x || !grad_output)
        return fail(hipErrorInvalidValue, "artemis_synth_head_backward: NULL tensor pointer");
    if (!grad_weights && !grad_offsets_y && !grad_offsets_x && !grad_occlusion) return 0;
    ha.gocc = grad_occlusion;
    hipStream_t st = (hipStream_t)stream;
    if (frame_dtype == ARTEMIS_SYNTH_FRAME_BF16)
        dispatch_head_bwd<asynth::bf16_t>(R, weights, offsets_y, offsets_x, grad_output, grad_weights,
                                          grad_offsets_y, grad_offsets_x, g, ha, st);
    else
        dispatch_head_bwd<float>(R, weights, offsets_y, offsets_x, grad_output, grad_weights,
                                 grad_offsets_y, grad_offsets_x, g, ha, st);
    return check_launch("artemis_synth_head_backward");
}

long long artemis_synth_set_small_grid_tiles(long long tiles)
{
    if (tiles < 0) return asynth::g_small_grid_tiles.load(std::memory_order_relaxed);
    return asynth::g_small_grid_tiles.exchange(tiles, std::memory_order_relaxed);
}

int artemis_synth_offset_radius(const float *offset_y, const float *offset_x, size_t count,
                                int *out_radius, void *stream)
{
    if (!out_radius) return fail(hipErrorInvalidValue, "artemis_synth_offset_radius: NULL out_radius");
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(out_radius, 0, sizeof(int), st);
    if (e != hipSuccess) return fail((int)e, "hipMemsetAsync: %s", hipGetErrorString(e));
    if (count == 0) return 0;
    if (!offset_y || !offset_x)
        return fail(hipErrorInvalidValue, "artemis_synth_offset_radius: NULL tensor pointer");
    hipLaunchKernelGGL(offset_radius_kernel, dim3(reduce_blocks(count)), dim3(256), 0, st, offset_y,
                       offset_x, count, out_radius);
    return check_launch("artemis_synth_offset_radius");
}

}  // extern "C"
