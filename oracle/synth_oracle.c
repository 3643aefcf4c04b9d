/*
 * synth_oracle.c -- CPU restatement of ArTEMIS's frame-synthesis kernels.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and there
 * only as the checker / the timed CPU baseline.  The product path (artemis_amd/) never links,
 * imports or falls back to it.
 *
 * PARITY UNPINNED.  The reference holds no tests, golden vectors or fixtures for this path
 * (SURVEY.md section 4), its only executable form needs CuPy + NVRTC + an NVIDIA GPU (absent
 * here), and its kernels are CUDA C held in Python strings with textual macros, so no reference
 * build is possible in this image.  This file is therefore pinned only by (1) known answers
 * derived by hand from the reference source (tests/test_oracle.py: one-hot taps, integer
 * shifts, truncation-toward-zero extrapolation, border clamp), (2) an independent vectorised
 * restatement (oracle/synth_oracle.py) and (3) fp64 finite differences of the forward.
 *
 * Each function names the reference lines it follows (paths relative to /root/reference).
 * Arithmetic is fp32 in the reference's operation order; build with -ffp-contract=off so no
 * FMA contraction changes the rounding (oracle/Makefile does).
 *
 * Tensor layout everywhere: NCHW, contiguous, fp32.
 *   input    (N, C, Hin, Win)   Hin = H + (k-1)*d, Win = W + (k-1)*d
 *   weight   (N, k*k, H, W)     offset_y, offset_x likewise
 *   output   (N, C, H, W)       gradLoss likewise
 *
 * The *_ex functions take the frame size (Hin, Win) and the frame coordinate (oy0, ox0) of
 * output pixel (0, 0)'s first tap explicitly (the reference layout is Hin = H + (k-1)*d,
 * oy0 = 0).  They restate two compositions the reference makes around the kernel: the
 * ReplicationPad2d in front of it (model/chrono_synth.py:70,125: an unpadded frame with
 * oy0 = ox0 = -(k-1)*d/2 reads the same texels, because clamping the index to the unpadded
 * frame is the replication) and row strips of one frame (oy0 = first row of the strip).
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* CLAMP(v, lo, hi) -> min(max(v, lo), hi): cupy_module/synth.py:237-250 */
static inline int clampi(int v, int lo, int hi)
{
    int m = v > lo ? v : lo;
    return m < hi ? m : hi;
}

/* (int)alpha as the device executes it (cupy_module/synth.py:29-30): round toward zero,
 * saturate out-of-range values, NaN -> 0 (CUDA cvt.rzi.s32.f32 and CDNA v_cvt_i32_f32 agree;
 * plain C leaves those cases undefined, so they are spelled out here). */
static inline int trunc_to_int(float v)
{
    if (v != v) return 0;
    if (v >= 2147483648.0f) return INT32_MAX;
    if (v <= -2147483648.0f) return INT32_MIN;
    return (int)v;
}

/* int + int with the device's two's-complement wrap (the reference adds without checks). */
static inline int wrap_add(int a, int b)
{
    return (int)((uint32_t)a + (uint32_t)b);
}

typedef struct {
    int bottom, top, left, right;
    float alphaTrunc, betaTrunc;
} tap_geom;

/* Index/fraction computation shared by all four kernels:
 * cupy_module/synth.py:29-38 (fwd), :79-88 (gW), :130-138 (gAlpha), :180-188 (gBeta).
 * y, x: frame coordinates of the pixel's first tap (output coordinate + origin). */
static inline tap_geom tap_geometry(float alpha, float beta, int y, int x, int row, int col,
                                    int d, int Hin, int Win)
{
    tap_geom g;
    int intAlpha = trunc_to_int(alpha);
    int intBeta = trunc_to_int(beta);
    int yy = wrap_add(y + row * d, intAlpha);
    int xx = wrap_add(x + col * d, intBeta);
    g.bottom = clampi(yy, 0, Hin - 1);
    g.left = clampi(xx, 0, Win - 1);
    g.top = clampi(wrap_add(yy, 1), 0, Hin - 1);
    g.right = clampi(wrap_add(xx, 1), 0, Win - 1);
    g.alphaTrunc = alpha - (float)intAlpha;
    g.betaTrunc = beta - (float)intBeta;
    return g;
}

static int g_threads = 1;

/* Threads used by the OpenMP loops (bench.py's cpu_baseline states this count). */
void synth_oracle_set_threads(int n)
{
    g_threads = n > 0 ? n : 1;
}

int synth_oracle_get_threads(void)
{
    return g_threads;
}

int synth_oracle_has_openmp(void)
{
#ifdef _OPENMP
    return 1;
#else
    return 0;
#endif
}

/* kernel_Synth_updateOutput, cupy_module/synth.py:16-50. One iteration per output element
 * (n, c, y, x); taps row-major; corner products summed left to right, then times w. */
void synth_oracle_forward_ex(const float *input, const float *weight, const float *offset_y,
                             const float *offset_x, float *output, int N, int C, int H, int W,
                             int k, int d, int Hin, int Win, int oy0, int ox0)
{
    const size_t plane = (size_t)H * W, iplane = (size_t)Hin * Win;
    const long rows = (long)N * C * H;
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (long rix = 0; rix < rows; ++rix) {
        const int y = (int)(rix % H);
        const int c = (int)((rix / H) % C);
        const int n = (int)(rix / H / C);
        const float *in = input + ((size_t)n * C + c) * iplane;
        for (int x = 0; x < W; ++x) {
            float dblOutput = 0.0f;
            for (int row = 0; row < k; ++row) {
                for (int col = 0; col < k; ++col) {
                    const size_t t = ((size_t)n * k * k + (size_t)(row * k + col)) * plane +
                                     (size_t)y * W + x;
                    const float w = weight[t];
                    const float alpha = offset_y[t];
                    const float beta = offset_x[t];
                    const tap_geom g = tap_geometry(alpha, beta, y + oy0, x + ox0, row, col, d,
                                                    Hin, Win);
                    const float a = g.alphaTrunc, b = g.betaTrunc;
                    dblOutput += w * (in[(size_t)g.bottom * Win + g.left] * (1 - a) * (1 - b) +
                                      in[(size_t)g.top * Win + g.left] * a * (1 - b) +
                                      in[(size_t)g.bottom * Win + g.right] * (1 - a) * b +
                                      in[(size_t)g.top * Win + g.right] * a * b);
                }
            }
            output[(((size_t)n * C + c) * H + y) * W + x] = dblOutput;
        }
    }
}

void synth_oracle_forward(const float *input, const float *weight, const float *offset_y,
                          const float *offset_x, float *output, int N, int C, int H, int W, int k,
                          int d)
{
    synth_oracle_forward_ex(input, weight, offset_y, offset_x, output, N, C, H, W, k, d,
                            H + (k - 1) * d, W + (k - 1) * d, 0, 0);
}

/* which: 0 = kernel_Synth_updateGradWeight (cupy_module/synth.py:64-99),
 *        1 = kernel_Synth_updateGradAlpha  (:114-149),
 *        2 = kernel_Synth_updateGradBeta   (:164-199).
 * One iteration per (n, tap, y, x); channels summed in order 0..C-1.  The reference hard-codes
 * `depth < 3` (:75,:125,:175); C is a parameter here and equals 3 wherever the reference is
 * defined. */
static void grad_taps(int which, const float *gradLoss, const float *input, const float *weight,
                      const float *offset_y, const float *offset_x, float *grad, int N, int C,
                      int H, int W, int k, int d, int Hin, int Win, int oy0, int ox0)
{
    const int K = k * k;
    const size_t plane = (size_t)H * W, iplane = (size_t)Hin * Win;
    const long rows = (long)N * K * H;
#pragma omp parallel for schedule(static) num_threads(g_threads)
    for (long rix = 0; rix < rows; ++rix) {
        const int y = (int)(rix % H);
        const int intDepth = (int)((rix / H) % K);
        const int n = (int)(rix / H / K);
        const int row = intDepth / k, col = intDepth % k;
        for (int x = 0; x < W; ++x) {
            float floatOutput = 0.0f;
            const size_t t = ((size_t)n * K + intDepth) * plane + (size_t)y * W + x;
            for (int depth = 0; depth < C; ++depth) {
                const float delta = gradLoss[(((size_t)n * C + depth) * H + y) * W + x];
                const float alpha = offset_y[t];
                const float beta = offset_x[t];
                const tap_geom g = tap_geometry(alpha, beta, y + oy0, x + ox0, row, col, d, Hin,
                                                Win);
                const float *in = input + ((size_t)n * C + depth) * iplane;
                const float bl = in[(size_t)g.bottom * Win + g.left];
                const float tl = in[(size_t)g.top * Win + g.left];
                const float br = in[(size_t)g.bottom * Win + g.right];
                const float tr = in[(size_t)g.top * Win + g.right];
                const float a = g.alphaTrunc, b = g.betaTrunc;
                if (which == 0) {
                    floatOutput += delta * (bl * (1 - a) * (1 - b) + tl * a * (1 - b) +
                                            br * (1 - a) * b + tr * a * b);
                } else if (which == 1) {
                    const float w = weight[t];
                    floatOutput += delta * w * (-bl * (1 - b) + tl * (1 - b) - br * b + tr * b);
                } else {
                    const float w = weight[t];
                    floatOutput += delta * w * (-bl * (1 - a) - tl * a + br * (1 - a) + tr * a);
                }
            }
            grad[t] = floatOutput;
        }
    }
}

void synth_oracle_grad_weight_ex(const float *gradLoss, const float *input, const float *offset_y,
                                 const float *offset_x, float *gradWeight, int N, int C, int H,
                                 int W, int k, int d, int Hin, int Win, int oy0, int ox0)
{
    grad_taps(0, gradLoss, input, NULL, offset_y, offset_x, gradWeight, N, C, H, W, k, d, Hin, Win,
              oy0, ox0);
}

void synth_oracle_grad_alpha_ex(const float *gradLoss, const float *input, const float *weight,
                                const float *offset_y, const float *offset_x, float *gradOffset_y,
                                int N, int C, int H, int W, int k, int d, int Hin, int Win, int oy0,
                                int ox0)
{
    grad_taps(1, gradLoss, input, weight, offset_y, offset_x, gradOffset_y, N, C, H, W, k, d, Hin,
              Win, oy0, ox0);
}

void synth_oracle_grad_beta_ex(const float *gradLoss, const float *input, const float *weight,
                               const float *offset_y, const float *offset_x, float *gradOffset_x,
                               int N, int C, int H, int W, int k, int d, int Hin, int Win, int oy0,
                               int ox0)
{
    grad_taps(2, gradLoss, input, weight, offset_y, offset_x, gradOffset_x, N, C, H, W, k, d, Hin,
              Win, oy0, ox0);
}

void synth_oracle_grad_weight(const float *gradLoss, const float *input, const float *offset_y,
                              const float *offset_x, float *gradWeight, int N, int C, int H, int W,
                              int k, int d)
{
    synth_oracle_grad_weight_ex(gradLoss, input, offset_y, offset_x, gradWeight, N, C, H, W, k, d,
                                H + (k - 1) * d, W + (k - 1) * d, 0, 0);
}

void synth_oracle_grad_alpha(const float *gradLoss, const float *input, const float *weight,
                             const float *offset_y, const float *offset_x, float *gradOffset_y,
                             int N, int C, int H, int W, int k, int d)
{
    synth_oracle_grad_alpha_ex(gradLoss, input, weight, offset_y, offset_x, gradOffset_y, N, C, H,
                               W, k, d, H + (k - 1) * d, W + (k - 1) * d, 0, 0);
}

void synth_oracle_grad_beta(const float *gradLoss, const float *input, const float *weight,
                            const float *offset_y, const float *offset_x, float *gradOffset_x,
                            int N, int C, int H, int W, int k, int d)
{
    synth_oracle_grad_beta_ex(gradLoss, input, weight, offset_y, offset_x, gradOffset_x, N, C, H,
                              W, k, d, H + (k - 1) * d, W + (k - 1) * d, 0, 0);
}

/* gradInput.  NOT in the reference: cupy_module/synth.py:333 allocates zeros and no kernel
 * writes them (:338-387).  This is the exact adjoint of synth_oracle_forward with respect to
 * `input`: every forward read  in[ci][cj] * coef * w  sends  gradLoss * w * coef  back to
 * (ci, cj), clamped corners included.  Accumulated in fp64 in a fixed order so it can judge
 * both the atomic and the deterministic device paths; checked in tests against autograd through
 * the torch restatement and against finite differences. Serial over pixels within a plane. */
void synth_oracle_grad_input_ex(const float *gradLoss, const float *weight, const float *offset_y,
                                const float *offset_x, float *gradInput, int N, int C, int H,
                                int W, int k, int d, int Hin, int Win, int oy0, int ox0)
{
    const int K = k * k;
    const size_t plane = (size_t)H * W, iplane = (size_t)Hin * Win;
    const long planes = (long)N * C;
#pragma omp parallel for schedule(dynamic, 1) num_threads(g_threads)
    for (long pix = 0; pix < planes; ++pix) {
        const int n = (int)(pix / C);
        double *acc = (double *)calloc(iplane, sizeof(double));
        const float *gl = gradLoss + (size_t)pix * plane;
        for (int y = 0; y < H; ++y) {
            for (int x = 0; x < W; ++x) {
                const double delta = gl[(size_t)y * W + x];
                for (int row = 0; row < k; ++row) {
                    for (int col = 0; col < k; ++col) {
                        const size_t t = ((size_t)n * K + (size_t)(row * k + col)) * plane +
                                         (size_t)y * W + x;
                        const tap_geom g = tap_geometry(offset_y[t], offset_x[t], y + oy0,
                                                        x + ox0, row, col, d, Hin, Win);
                        const double a = g.alphaTrunc, b = g.betaTrunc;
                        const double dw = delta * (double)weight[t];
                        acc[(size_t)g.bottom * Win + g.left] += dw * (1 - a) * (1 - b);
                        acc[(size_t)g.top * Win + g.left] += dw * a * (1 - b);
                        acc[(size_t)g.bottom * Win + g.right] += dw * (1 - a) * b;
                        acc[(size_t)g.top * Win + g.right] += dw * a * b;
                    }
                }
            }
        }
        float *gi = gradInput + (size_t)pix * iplane;
        for (size_t i = 0; i < iplane; ++i) gi[i] = (float)acc[i];
        free(acc);
    }
}

void synth_oracle_grad_input(const float *gradLoss, const float *weight, const float *offset_y,
                             const float *offset_x, float *gradInput, int N, int C, int H, int W,
                             int k, int d)
{
    synth_oracle_grad_input_ex(gradLoss, weight, offset_y, offset_x, gradInput, N, C, H, W, k, d,
                               H + (k - 1) * d, W + (k - 1) * d, 0, 0);
}
