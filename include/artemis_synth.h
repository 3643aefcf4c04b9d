/*
 * artemis_synth.h -- C ABI of libartemis_synth.so: ArTEMIS's frame-synthesis op (per-pixel
 * adaptive deformable convolution) as hand-written CDNA4 (gfx950) HIP kernels.
 *
 * This is the drop-in boundary for the one hot path this repository replaces.  In the reference
 * (paths relative to /root/reference) the boundary is the raw CuPy launch
 *     cupy_launch(name, cupy_kernel(name, k, d, {...}))(grid, block, args=[n, ptrs...], stream)
 * at cupy_module/synth.py:298-309 (forward) and :343-387 (backward), issued from
 * FunctionSynth.forward / .backward (:263-391).  Each entry point below states which of those
 * it replaces.  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *  - plain C: pointers are DEVICE pointers on the current HIP device, sizes are ints, no torch
 *    or C++ types cross the boundary.
 *  - tensors are contiguous NCHW fp32 (the reference asserts contiguity, :280-283, :331):
 *        input     (N, C, H + (k-1)*d, W + (k-1)*d)     the replication-padded frame
 *        weight    (N, k*k, H, W)                        offset_y / offset_x likewise
 *        output    (N, C, H, W)                          grad_output likewise
 *  - the library never allocates, frees or retains caller memory; it only enqueues work on
 *    `stream` (a hipStream_t; NULL = the null stream) and returns.  The reference launches on
 *    torch.cuda.current_stream() (:294-295, :339-340); the caller passes that handle.
 *  - every function returns 0 on success or a hipError_t value (> 0); arguments that fail
 *    validation give hipErrorInvalidValue (1).  artemis_synth_last_error() returns a
 *    thread-local description of the last failure on the calling thread.  No exception crosses
 *    the boundary.
 *  - entry points are reentrant: forward is called from the user's thread, backward from PyTorch's
 *    autograd thread.  The mutable state of the library is two atomic tuning integers
 *    (artemis_synth_set_small_grid_tiles, artemis_synth_set_gin_gather) and, only for callers that pass
 *    ARTEMIS_SYNTH_RADIUS_AUTO, a mutex-guarded table of per-shape radii (see there).
 *  - empty problems (N, H or W equal to 0) are a successful no-op.
 *  - limits: one (H*W) plane and one padded frame plane must each hold fewer than 2^31
 *    elements; N <= 65535; total element counts are 64-bit (the reference's `const int n`
 *    limit, :8, is not inherited).
 */
#ifndef ARTEMIS_SYNTH_H
#define ARTEMIS_SYNTH_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ARTEMIS_SYNTH_ABI_VERSION 1

/* tile_radius: how far (in whole pixels) a tap may be displaced by its offsets and still be
 * served from an LDS-staged window of the frame; larger displacements take a slower clamped global
 * gather, so every value is correct for any offset.  The displacement is measured from the
 * window's own centre: a window (64 x 16 output pixels) is staged around the displacement found at
 * its middle, and when enough of its pixels still have taps outside it that share another coherent
 * displacement (a motion boundary, a steep flow gradient), up to two more windows are staged around
 * those -- so large coherent motion runs from LDS whatever its size.  > 0 picks the smallest built
 * radius (2, 4, 8, 14, 24) that covers it, 24 beyond; ARTEMIS_SYNTH_RADIUS_DEFAULT lets the library
 * choose (14); ARTEMIS_SYNTH_RADIUS_NONE disables LDS staging (direct gathers through L1/L2).
 * Radius 24 is for offsets that may be large or steep (a flow that changes by tens of pixels across
 * a tile, independent displacements of 4-10 px per tap): one window per 64 x 32 tile instead of two
 * per 128 x 16.  Measured, 4 frames x 1080p forward + backward (the final bench line,
 * profiles/r3_bench.json: config.other_offset_regimes): on N(0,1) px offsets radius 14 runs at 0.934
 * and radius 24 at 0.912 of the lean shape (ARTEMIS_SYNTH_RADIUS_SMALL: 1491 Mpix/s; round 1's
 * kernels, which were this shape, 1507); on a smooth +-30 px flow 24 runs at 0.759 and 14 at 0.468.
 * grad_input's accumulator window stays at 14.  Only k = 5,
 * d = 1, C = 3 (the model's configuration: config.py:26-27) has an LDS-staged kernel; other shapes
 * always gather directly. */
#define ARTEMIS_SYNTH_RADIUS_DEFAULT (-1)
#define ARTEMIS_SYNTH_RADIUS_NONE 0
/* ARTEMIS_SYNTH_RADIUS_AUTO: the library moves between the radii 8, 14 and 24 by itself, per (device,
 * H, W, per-frame op | fused head), from what the forward kernels of that shape saw on their earlier
 * launches: over a sample of the pixels they count the taps displaced beyond 8 whole pixels from
 * the pixel itself and beyond 14 from their window's centre; every few launches a one-wave kernel
 * behind the launch publishes the counts to pinned host memory, and the next AUTO call -- forward,
 * backward or head -- reads them with plain host loads and picks the lean radius-8 shape
 * (ARTEMIS_SYNTH_RADIUS_SMALL, below) when fewer than 5e-5 of the taps lie beyond 8, else 14 when
 * fewer than 1.5e-4 lie beyond 14 of a window that follows the motion, else 24 (a factor-two margin
 * before going back down).  No host
 * synchronisation, no extra pass over the offsets.  A new shape starts at the default (14) and
 * follows its offsets with a lag of the launches in flight.  Results are those of the chosen fixed
 * radius: correct for any offsets, equal between radii up to fp32 summation order -- pass a fixed
 * radius where run-to-run bit equality of the forward matters.  This is the library's one piece of
 * mutable state beside the small-grid knob (a mutex-guarded table of 16 shapes per device); its one
 * allocation -- 512 bytes of pinned host memory per device -- happens on the first AUTO FORWARD call
 * on that device (make that call outside stream capture); backward calls only look the radius up.
 * Shapes without LDS kernels (k != 5, d != 1, C != 3) treat AUTO as DEFAULT. */
#define ARTEMIS_SYNTH_RADIUS_AUTO (-2)
/* ARTEMIS_SYNTH_RADIUS_SMALL: radius 8 in the LEAN shape -- the 128 x 16 tile served by one window that is
 * staged once, at the tile's own position: no window centres, no vote, no further windows.  For
 * offsets known to stay within 8 pixels (random initialisation, small motion); anything else -- a
 * pan included: the other shapes' windows follow coherent motion, this one's do not -- is still
 * correct, through the per-pixel global fix-up, only slow.  It is what AUTO runs once its statistics
 * show practically no tap beyond 8 pixels.  grad_input runs its radius-8 kernel. */
#define ARTEMIS_SYNTH_RADIUS_SMALL (-3)

/* grad_input accumulation mode of artemis_synth_backward */
#define ARTEMIS_SYNTH_GIN_ATOMIC 0        /* fp32 atomics across tiles; last-bit run-to-run noise */
#define ARTEMIS_SYNTH_GIN_DETERMINISTIC 1 /* bit-exact run to run; needs the large workspace */

/* frame_dtype of the *_ex and head entry points: element type of the frame(s).  The reference is
 * fp32 only (cupy_module/synth.py:9-13 declares float everywhere); bf16 frames (BASELINE configs[4])
 * are widened exactly to fp32 when staged, all arithmetic and every other tensor stay fp32.  bf16 is
 * built for k = 5, d = 1, C = 3 and tile radii 8, 14 and 24 (RADIUS_NONE means the default radius, a
 * smaller request runs the radius-8 kernels). */
#define ARTEMIS_SYNTH_FRAME_F32 0
#define ARTEMIS_SYNTH_FRAME_BF16 1

/* Replaces the kernel_Synth_updateOutput launch, cupy_module/synth.py:297-309
 * (kernel text :6-52; called from FunctionSynth.forward :265-313).
 *   output[n,c,y,x] = sum over taps (r,s) of weight[n,r*k+s,y,x] * bilinear(input[n,c], ...)
 * with truncation toward zero of the offsets and index clamping exactly as the reference. */
int artemis_synth_forward(const float *input, const float *weight, const float *offset_y,
                          const float *offset_x, float *output, int N, int C, int H, int W, int k,
                          int d, int tile_radius, void *stream);

/* Replaces the three launches kernel_Synth_updateGradWeight / updateGradAlpha / updateGradBeta,
 * cupy_module/synth.py:343-387 (kernel text :54-201; called from FunctionSynth.backward
 * :316-391), fused into one pass.  Each of grad_input, grad_weight, grad_offset_y,
 * grad_offset_x may be NULL (that gradient is not needed: context.needs_input_grad, :333-336);
 * at least the non-NULL ones are fully overwritten.
 *
 * grad_input has no counterpart in the reference, which returns unfilled zeros (:333, :391);
 * here it is the true adjoint of the forward w.r.t. `input`.  gin_mode selects fp32 atomics or
 * the deterministic (integer fixed-point) accumulation.  When grad_input is requested,
 * `workspace` must hold at least artemis_synth_backward_workspace_bytes(...) bytes for that
 * gin_mode (device memory, 16-byte aligned, contents undefined on entry and exit): 256 bytes in
 * atomic mode, 256 + 8 bytes per grad_input element in deterministic mode.  `workspace` may be
 * NULL when grad_input is NULL or the size query returns 0.
 * Channels: the reference hard-codes 3 (:75, :125, :175); any C >= 1 is summed here. */
int artemis_synth_backward(const float *input, const float *weight, const float *offset_y,
                           const float *offset_x, const float *grad_output, float *grad_input,
                           float *grad_weight, float *grad_offset_y, float *grad_offset_x, int N,
                           int C, int H, int W, int k, int d, int gin_mode, int tile_radius,
                           void *workspace, size_t workspace_bytes, void *stream);

/* Bytes of workspace artemis_synth_backward needs for these dimensions and gin_mode when
 * grad_input is requested, valid for every tile_radius (0 when none is needed).  Host-only
 * arithmetic; no reference counterpart (the reference has no grad_input kernel). */
size_t artemis_synth_backward_workspace_bytes(int N, int C, int H, int W, int k, int d,
                                              int gin_mode);

/* ---- explicit frame geometry ------------------------------------------------------------
 * The *_ex entry points are the same kernels with the frame's size and origin passed instead
 * of implied: `input` is (N, C, in_H, in_W) of frame_dtype (grad_input stays fp32) and tap (r, s) of output pixel (y, x) samples around
 * frame coordinate
 *        ( y + org_y + r*d + trunc(offset_y),  x + org_x + s*d + trunc(offset_x) )
 * with all four corner indices clamped to [0, in_H-1] x [0, in_W-1] as in the reference
 * (cupy_module/synth.py:32-35).  artemis_synth_forward/backward are the case
 * in_H = H + (k-1)*d, in_W = W + (k-1)*d, org = 0.  Two compositions the reference makes around
 * the kernel become single calls with no extra copy of the frame:
 *  - the ReplicationPad2d in front of it (model/chrono_synth.py:70 and :125-126,
 *    `moduleSynth(modulePad(frame), ...)`): pass the UNPADDED frame, in_H = H, in_W = W,
 *    org_y = org_x = -((k-1)*d/2).  Clamping to the unpadded frame is the replication, so output
 *    and tap gradients are bit-identical to padding first, and grad_input comes out already
 *    folded onto the unpadded frame (the adjoint of the padding).
 *  - row strips of one frame (4K inputs whose per-call weight/offset tensors should stay small,
 *    or pipelining a frame against the producer of its weights): weight/offset/output tensors hold
 *    rows [y0, y1) only (H = y1 - y0), `input` is the whole frame, org_y = y0 (+ the pad origin
 *    above).  grad_input of a strip call covers the whole frame; strips' grad_inputs add.
 * org_y, org_x may be negative; |org| <= 2^24. */
int artemis_synth_forward_ex(const void *input, const float *weight, const float *offset_y,
                             const float *offset_x, float *output, int N, int C, int H, int W, int k,
                             int d, int in_H, int in_W, int org_y, int org_x, int frame_dtype,
                             int tile_radius, void *stream);

int artemis_synth_backward_ex(const void *input, const float *weight, const float *offset_y,
                              const float *offset_x, const float *grad_output, float *grad_input,
                              float *grad_weight, float *grad_offset_y, float *grad_offset_x, int N,
                              int C, int H, int W, int k, int d, int in_H, int in_W, int org_y,
                              int org_x, int frame_dtype, int gin_mode, int tile_radius,
                              void *workspace, size_t workspace_bytes, void *stream);

/* Workspace bytes for artemis_synth_backward_ex (grad_input is (N, C, in_H, in_W)). */
size_t artemis_synth_backward_workspace_bytes_ex(int N, int C, int H, int W, int k, int d, int in_H,
                                                 int in_W, int gin_mode);

/* ---- allocation-free backward --------------------------------------------------------------
 * NO COUNTERPART IN THE REFERENCE.  For callers that cannot allocate device memory around the
 * call: code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.
 * artemis_synth_backward_ex with its caller-provided `workspace` replaced by a device array that
 * lives inside the library (a file-scope `static __device__` array of
 * ARTEMIS_SYNTH_NOALLOC_SCRATCH_BYTES bytes, held by every GPU the library runs on until the
 * process ends); every other argument, every kernel and every result bit are those of
 * artemis_synth_backward_ex.  The call itself allocates and frees nothing (no hipMalloc /
 * hipFree / allocating library routine); the caller still passes all inputs and outputs.
 *  - limit: the workspace this problem needs (artemis_synth_backward_workspace_bytes_ex: 256 B
 *    in atomic mode, 256 + 8 B per grad_input element in deterministic mode, 0 without
 *    grad_input) must not exceed ARTEMIS_SYNTH_NOALLOC_SCRATCH_BYTES -- i.e. a deterministic
 *    grad_input of up to 8,388,576 elements (one padded 1080p RGB frame is 6,256,848); larger
 *    problems are refused with hipErrorInvalidValue and a message, before any GPU work.
 *  - all calls on one GPU share that GPU's array and it keeps the previous call's contents; each
 *    call clears what it needs (hipMemsetAsync on `stream`), so consecutive calls on ONE stream
 *    are safe, while calls that overlap on one GPU (different streams or threads) and both
 *    request grad_input RACE on it -- serialise them.  Calls without grad_input never touch it.
 *  - the array's device address is resolved once per device, on the first call made with that
 *    device current (hipGetSymbolAddress); make that first call outside graph capture. */
#define ARTEMIS_SYNTH_NOALLOC_SCRATCH_BYTES (64u << 20)
int artemis_synth_backward_noalloc(const void *input, const float *weight, const float *offset_y,
                                   const float *offset_x, const float *grad_output,
                                   float *grad_input, float *grad_weight, float *grad_offset_y,
                                   float *grad_offset_x, int N, int C, int H, int W, int k, int d,
                                   int in_H, int in_W, int org_y, int org_x, int frame_dtype,
                                   int gin_mode, int tile_radius, void *stream);

/* ---- fused ChronoSynth head --------------------------------------------------------------
 * One call for what the reference's only caller does around the op for one pyramid level,
 * model/chrono_synth.py:114-129:
 *     for i in range(T):  warp_i = occlusion[:, i:i+1] *
 *                                  FunctionSynth(ReplicationPad2d(frame_i), weights[:, i].contiguous(),
 *                                                alphas[:, i].contiguous(), betas[:, i].contiguous(), d)
 *     framet = sum(warp)
 * The 5-D tensors are read in place (no .contiguous() copies: 600 B/pixel/frame of memcpy in the
 * reference), the per-frame results never leave registers, and the output has the bits of the
 * composition (product rounded to fp32, then added in frame order).
 *   frames      HOST array of T device pointers, each (B, 3, in_H, in_W) of frame_dtype; geometry as in the
 *               *_ex entry points (unpadded frames: in_H = H, in_W = W, org = -2; padded: in_H = H+4,
 *               org = 0).  1 <= T <= 8.  The array is read during the call only.
 *   weights / offsets_y / offsets_x   (B, T, 25, H, W) contiguous
 *   occlusion   (B, T, H, W) contiguous          output  (B, 3, H, W)
 * Only the model's configuration k = 5, d = 1, C = 3 (config.py:26-27) is built; tile_radius as
 * above except that RADIUS_NONE means the default (the head has no direct-gather variant). */
int artemis_synth_head_forward(const void *const *frames, int T, const float *weights,
                               const float *offsets_y, const float *offsets_x,
                               const float *occlusion, float *output, int B, int H, int W, int in_H,
                               int in_W, int org_y, int org_x, int frame_dtype, int tile_radius,
                               void *stream);

/* Backward of artemis_synth_head_forward w.r.t. weights, offsets and occlusion (what autograd
 * derives for model/chrono_synth.py:114-129 through FunctionSynth.backward, :316-391, the `occ *`
 * product and the sum).  grad_output is (B, 3, H, W); grad_weights / grad_offsets_y /
 * grad_offsets_x are (B, T, 25, H, W), grad_occlusion is (B, T, H, W); each may be NULL, the
 * others are fully overwritten.  Frame gradients are not part of this kernel (ArTEMIS's frames are
 * inputs); a caller that needs them uses artemis_synth_backward_ex per frame with
 * grad_output * occlusion. */
int artemis_synth_head_backward(const void *const *frames, int T, const float *weights,
                                const float *offsets_y, const float *offsets_x,
                                const float *occlusion, const float *grad_output,
                                float *grad_weights, float *grad_offsets_y, float *grad_offsets_x,
                                float *grad_occlusion, int B, int H, int W, int in_H, int in_W,
                                int org_y, int org_x, int frame_dtype, int tile_radius, void *stream);

/* max over both offset tensors of |trunc(offset)| as one int32 written to *out_radius (device
 * memory), enqueued on `stream`: the tile_radius that would serve every tap from LDS.  `count`
 * = elements per tensor.  Helper for callers that want to size tile_radius; no reference
 * counterpart. */
int artemis_synth_offset_radius(const float *offset_y, const float *offset_x, size_t count,
                                int *out_radius, void *stream);

/* What ARTEMIS_SYNTH_RADIUS_AUTO currently holds for shape (H, W) of the per-frame op (head = 0) or
 * the fused head (head = 1) on the current device: the radius the next call would run, the
 * fractions of sampled taps beyond 8 / beyond 14 pixels in the last published window, and the number
 * of AUTO forward launches seen (0 and the default radius for a shape not seen yet).  Any pointer may
 * be NULL.  Host-only, no synchronisation.  artemis_synth_auto_reset forgets every shape of the
 * current device.  No reference counterpart. */
int artemis_synth_auto_query(int H, int W, int head, int *radius, double *frac8, double *frac14,
                             unsigned long long *forward_calls);
void artemis_synth_auto_reset(void);

/* gradInput in gather form.  When artemis_synth_backward* is asked for grad_input (k = 5, d = 1, C = 3,
 * tile_radius != NONE, a frame of at least 32 x 63 texels whose origin lies at most 3 rows / columns
 * outside it) and every offset truncates to zero -- |offset| < 1 everywhere: a freshly initialised
 * model -- grad_input is summed per texel in registers in a fixed order instead of scattered through
 * atomics: bit-reproducible in either gin_mode and faster.  Whether the offsets
 * qualify is found out on the device: the gather kernel runs first, checks every offset it loads and
 * gives up if one reaches 1 (raising word 2 of the workspace header, which is 0 after a
 * call that the gather kernel completed); the scatter kernels are enqueued behind it and leave at
 * once unless it gave up.  No host synchronisation.  Giving up costs 13-18 us per 1080p frame; the
 * kernel's first workgroup reports it through the pinned record of ARTEMIS_SYNTH_RADIUS_AUTO (when a
 * forward call with that radius has set it up on the device), after which the next 1, 2, 4 .. 64
 * backward calls do not enqueue the kernel before it is tried again.
 * artemis_synth_set_gin_gather(0) turns the path off, (1) on (returns the previous setting and
 * forgets the back-off; a negative argument only queries).  No reference counterpart (upstream has
 * no grad_input). */
int artemis_synth_set_gin_gather(int enable);

/* Process-wide tuning knob (one atomic integer): problems
 * with fewer than `tiles` workgroup tiles of 64 x 32 pixels (N * ceil(W/64) * ceil(H/32)) run with
 * 64 x 8 tiles so that small grids (the 64^2 / 128^2 heads of a small-batch training step) still
 * spread over the CUs (the forward switches at a quarter of that count).  Default 768.  0 = never, a huge value = always; results are bit-identical
 * either way.  Returns the previous value; a negative argument only queries.  No reference
 * counterpart. */
long long artemis_synth_set_small_grid_tiles(long long tiles);

/* Library identification: "artemis_synth <abi> gfx950 ..." (static storage). */
const char *artemis_synth_version(void);

/* Thread-local text for the last non-zero return on this thread ("" if none). */
const char *artemis_synth_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* ARTEMIS_SYNTH_H */
