// synth_kernels.hip -- C ABI, host logic and direct (any k, d, C) kernels of libartemis_synth.so.
// Device code: synth_device.hpp; the LDS-staged kernels are instantiated per tile radius in
// synth_tiles.hip.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <atomic>
#include <mutex>

#include "synth_device.hpp"

namespace asynth {
static std::atomic<long long> g_small_grid_tiles{768};
long long small_grid_tiles() { return g_small_grid_tiles.load(std::memory_order_relaxed); }
}  // namespace asynth

namespace {
#include "synth_gin_gather.hpp"
std::atomic<int> g_gin_gather{1};  // artemis_synth_set_gin_gather

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
    if (requested == ARTEMIS_SYNTH_RADIUS_SMALL) return 8;  // (+ the lean shape: is_lean)
    if (requested < 0) requested = kDefaultRadius;  // (AUTO on a shape without LDS kernels included)
    if (requested <= 2) return 2;
    if (requested <= 4) return 4;
    if (requested <= 8) return 8;
    if (requested <= 14) return 14;
    return 24;
}

// RADIUS_SMALL: radius 8 in the lean shape (one window per tile, staged once; synth_tiles.hip LeanCfg)
bool is_lean(int requested) { return requested == ARTEMIS_SYNTH_RADIUS_SMALL; }

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
                               const float *offset_x, float *output, const Geom &g, bool lean,
                               unsigned long long *stats, hipStream_t st)
{
    const FT *in = (const FT *)input;
    switch (R) {
    case 2:
        if constexpr (kSmallRadiiBuilt<FT>) {
            launch_fwd_tiled<2, FT>(in, weight, offset_y, offset_x, output, g, lean, stats, st);
            break;
        }
        [[fallthrough]];
    case 4:
        if constexpr (kSmallRadiiBuilt<FT>) {
            launch_fwd_tiled<4, FT>(in, weight, offset_y, offset_x, output, g, lean, stats, st);
            break;
        }
        [[fallthrough]];
    case 8: launch_fwd_tiled<8, FT>(in, weight, offset_y, offset_x, output, g, lean, stats, st); break;
    case 24: launch_fwd_tiled<24, FT>(in, weight, offset_y, offset_x, output, g, lean, stats, st); break;
    default: launch_fwd_tiled<14, FT>(in, weight, offset_y, offset_x, output, g, lean, stats, st); break;
    }
}

template <class FT>
void dispatch_bwd_tiled(int R, const void *input, const float *weight, const float *offset_y,
                               const float *offset_x, const float *grad_output, float *gw, float *goy,
                               float *gox, const Geom &g, bool lean, unsigned *maxima, hipStream_t st)
{
    const FT *in = (const FT *)input;
    switch (R) {
    case 2:
        if constexpr (kSmallRadiiBuilt<FT>) {
            launch_bwd_tiled<2, FT>(in, weight, offset_y, offset_x, grad_output, gw, goy, gox, g, lean, maxima, st);
            break;
        }
        [[fallthrough]];
    case 4:
        if constexpr (kSmallRadiiBuilt<FT>) {
            launch_bwd_tiled<4, FT>(in, weight, offset_y, offset_x, grad_output, gw, goy, gox, g, lean, maxima, st);
            break;
        }
        [[fallthrough]];
    case 8: launch_bwd_tiled<8, FT>(in, weight, offset_y, offset_x, grad_output, gw, goy, gox, g, lean, maxima, st); break;
    case 24: launch_bwd_tiled<24, FT>(in, weight, offset_y, offset_x, grad_output, gw, goy, gox, g, lean, maxima, st); break;
    default: launch_bwd_tiled<14, FT>(in, weight, offset_y, offset_x, grad_output, gw, goy, gox, g, lean, maxima, st); break;
    }
}

template <class FT>
void dispatch_head_fwd(int R, const float *w, const float *oy, const float *ox, float *out,
                              const Geom &g, const HeadArgs &ha, bool lean, unsigned long long *stats,
                              hipStream_t st)
{
    switch (R) {
    case 2:
        if constexpr (kSmallRadiiBuilt<FT>) {
            launch_head_fwd<2, FT>(w, oy, ox, out, g, ha, lean, stats, st);
            break;
        }
        [[fallthrough]];
    case 4:
        if constexpr (kSmallRadiiBuilt<FT>) {
            launch_head_fwd<4, FT>(w, oy, ox, out, g, ha, lean, stats, st);
            break;
        }
        [[fallthrough]];
    case 8: launch_head_fwd<8, FT>(w, oy, ox, out, g, ha, lean, stats, st); break;
    case 24: launch_head_fwd<24, FT>(w, oy, ox, out, g, ha, lean, stats, st); break;
    default: launch_head_fwd<14, FT>(w, oy, ox, out, g, ha, lean, stats, st); break;
    }
}

template <class FT>
void dispatch_head_bwd(int R, const float *w, const float *oy, const float *ox,
                              const float *gout, float *gw, float *goy, float *gox, const Geom &g,
                              const HeadArgs &ha, bool lean, hipStream_t st)
{
    switch (R) {
    case 2:
        if constexpr (kSmallRadiiBuilt<FT>) {
            launch_head_bwd<2, FT>(w, oy, ox, gout, gw, goy, gox, g, ha, lean, st);
            break;
        }
        [[fallthrough]];
    case 4:
        if constexpr (kSmallRadiiBuilt<FT>) {
            launch_head_bwd<4, FT>(w, oy, ox, gout, gw, goy, gox, g, ha, lean, st);
            break;
        }
        [[fallthrough]];
    case 8: launch_head_bwd<8, FT>(w, oy, ox, gout, gw, goy, gox, g, ha, lean, st); break;
    case 24: launch_head_bwd<24, FT>(w, oy, ox, gout, gw, goy, gox, g, ha, lean, st); break;
    default: launch_head_bwd<14, FT>(w, oy, ox, gout, gw, goy, gox, g, ha, lean, st); break;
    }
}


// ------------------------------------------------------------------------------------------
// ARTEMIS_SYNTH_RADIUS_AUTO: the library picks 8, 14 or 24 per (device, H, W, op | head) from what
// the FORWARD kernels saw on their previous launches -- no host synchronisation anywhere.
//   device side  the forward kernels count, over a 1/ROWS sample of the pixels, the taps displaced
//                beyond 8 whole pixels from the pixel (what the lean shape, which has no window
//                centres, would leave over) and beyond 14 from their window's centre (synth_device.hpp,
//                reach_cnt) and add the counts, one no-return atomic per wave that has any, to the
//                key's record in `artemis_synth_auto_counters`;
//   publish      every few AUTO forward launches a one-wave kernel behind the launch sums the record,
//                clears it and stores {counts, sequence number} into a pinned host mirror;
//   host side    the next AUTO call (forward, backward or head) reads the mirror -- plain loads of
//                host memory -- and moves the key's radius: the smallest of 8 / 14 / 24 that leaves
//                fewer than kAutoFrac8 (8) / kAutoFrac14 (14) of the taps to the global fix-up, with
//                a factor-two margin before it goes back down.
// Any radius is correct for any offsets, so a stale or torn reading costs time, never results
// (bits may differ at rounding level between radii, like between any two tile_radius values).
// ------------------------------------------------------------------------------------------
constexpr int kAutoKeys = 16;
constexpr double kAutoFrac8 = 5.0e-5;    // taps beyond 8 px that make radius 14 the better shape
constexpr double kAutoFrac14 = 1.5e-4;   // taps beyond 14 px that make radius 24 the better shape
constexpr double kAutoMinSamples = 5.0e4;

struct AutoMirror {  // one per key, in pinned host memory the GPU can write
    unsigned long long seq, c8, c14, pad;
};

struct AutoEntry {
    bool used;
    int H, W, kind;             // kind 0: per-frame op, 1: fused head
    int radius;
    unsigned long long calls;   // AUTO forward launches so far
    unsigned long long seq;     // publishes enqueued
    unsigned long long seen;    // last publish applied
    double pending;             // sampled taps launched since the last publish
    double window[8];           // sampled taps covered by publish `seq` (index seq % 8)
    double f8, f14;             // fractions of the last publish applied
};

struct AutoDevice {
    int state;  // 0 untried, 1 ready, -1 unavailable (allocation failed: AUTO = the default radius)
    unsigned long long *counters;  // device: kAutoKeys x kReachSlots
    AutoMirror *mirror;            // host view
    AutoMirror *mirror_dev;        // the same memory as the GPU addresses it
    AutoEntry e[kAutoKeys];
    // gradInput's gather kernel (gather_begin): attempts issued, last give-up report applied
    // (mirror[0].pad holds the id of the last attempt that gave up), calls still to skip, current
    // back-off, attempts since the last report
    unsigned long long ga_seq, ga_seen;
    int ga_skip, ga_backoff, ga_quiet;
};
}  // namespace

// (external linkage and `used`: host code reaches it through hipGetSymbolAddress only; as a `static`
// variable it stayed local to the code object and the lookup aborted inside the HIP runtime)
__device__ __align__(256) __attribute__((used)) unsigned long long
    artemis_synth_auto_counters[kAutoKeys * kReachSlots];

namespace {
constexpr int kMaxDevices = 64;
AutoDevice g_auto[kMaxDevices];
std::mutex g_auto_mutex;

__global__ void reach_publish_kernel(unsigned long long *rec, AutoMirror *mirror, unsigned long long seq)
{
    unsigned long long v = 0;
    if (threadIdx.x < kReachSlots) v = atomicExch(rec + threadIdx.x, 0ull);
    unsigned long long lo = v & 0xffffffffull, hi = v >> 32;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo += (unsigned long long)__shfl_xor((long long)lo, o);
        hi += (unsigned long long)__shfl_xor((long long)hi, o);
    }
    if (threadIdx.x == 0) {
        volatile AutoMirror *m = mirror;
        m->c8 = lo;
        m->c14 = hi;
        __threadfence_system();
        m->seq = seq;
    }
}

int auto_decide(int cur, double f8, double f14)
{
    // 8 stands for ARTEMIS_SYNTH_RADIUS_SMALL: radius 8 in the lean shape (auto_request)
    auto want_for = [&](double margin) {
        return f8 <= kAutoFrac8 * margin ? 8 : (f14 <= kAutoFrac14 * margin ? 14 : 24);
    };
    int want = want_for(1.0);
    if (want < cur) {  // going down needs a margin, so that a borderline workload does not flap
        const int safe = want_for(0.5);
        want = safe < cur ? safe : cur;
    }
    return want;
}

// the current device's record, set up on first use (the only allocation the library ever makes:
// 512 bytes of pinned host memory per device; it fails inside a stream capture, so make the first
// AUTO forward call on a device outside one).  nullptr: unavailable.  Caller holds g_auto_mutex.
AutoDevice *auto_device(bool may_allocate)
{
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
    AutoDevice &d = g_auto[dev];
    if (d.state == 0 && may_allocate) {
        d.state = -1;
        void *sym = nullptr, *host = nullptr, *hdev = nullptr;
        if (hipGetSymbolAddress(&sym, HIP_SYMBOL(artemis_synth_auto_counters)) == hipSuccess && sym &&
            hipHostMalloc(&host, kAutoKeys * sizeof(AutoMirror), hipHostMallocMapped) == hipSuccess && host) {
            memset(host, 0, kAutoKeys * sizeof(AutoMirror));
            if (hipHostGetDevicePointer(&hdev, host, 0) == hipSuccess && hdev) {
                d.counters = (unsigned long long *)sym;
                d.mirror = (AutoMirror *)host;
                d.mirror_dev = (AutoMirror *)hdev;
                d.state = 1;
            } else {
                (void)hipHostFree(host);
            }
        }
        (void)hipGetLastError();  // a failure here is not the caller's launch error
    }
    return d.state == 1 ? &d : nullptr;
}

int auto_find(AutoDevice &d, int H, int W, int kind, bool create)
{
    int free_slot = -1;
    for (int i = 0; i < kAutoKeys; ++i) {
        if (d.e[i].used && d.e[i].H == H && d.e[i].W == W && d.e[i].kind == kind) return i;
        if (!d.e[i].used && free_slot < 0) free_slot = i;
    }
    if (!create || free_slot < 0) return -1;  // (table full: the shape runs at the default radius)
    AutoEntry &e = d.e[free_slot];
    e = AutoEntry{};
    e.used = true;
    e.H = H, e.W = W, e.kind = kind;
    e.radius = kDefaultRadius;
    // publishes of an earlier user of this slot (artemis_synth_auto_reset) must not count
    e.seq = e.seen = ((volatile AutoMirror *)d.mirror)[free_slot].seq;
    return free_slot;
}

// apply what the GPU has published for entry i since the last look
void auto_absorb(AutoDevice &d, int i)
{
    AutoEntry &e = d.e[i];
    volatile AutoMirror *m = d.mirror + i;
    const unsigned long long seq = m->seq;
    if (seq <= e.seen || seq > e.seq) return;
    std::atomic_thread_fence(std::memory_order_acquire);
    const double c8 = (double)m->c8, c14 = (double)m->c14;
    // a publish sums AND clears the record, so the mirror holds the counts of window `seq` alone
    // (with more than 8 publishes in flight the ring entry may be a newer window's: an estimate)
    const double total = e.window[seq % 8];
    e.seen = seq;
    if (total < kAutoMinSamples) return;
    e.f8 = c8 / total;
    e.f14 = c14 / total;
    e.radius = auto_decide(e.radius, e.f8, e.f14);
}

// the tile_radius request an AUTO entry's radius stands for
int auto_request(int radius) { return radius == 8 ? ARTEMIS_SYNTH_RADIUS_SMALL : radius; }

struct AutoLaunch {
    int radius;
    unsigned long long *stats;  // the forward kernel's record, or null
    int dev, key;
};

// Radius (and, for a forward launch, the statistics record) of an AUTO call.
AutoLaunch auto_begin(const Geom &g, int kind, bool forward)
{
    AutoLaunch a{kDefaultRadius, nullptr, -1, -1};
    std::lock_guard<std::mutex> lock(g_auto_mutex);
    AutoDevice *d = auto_device(forward);
    if (!d) return a;
    const int i = auto_find(*d, g.H, g.W, kind, forward);
    if (i < 0) return a;
    auto_absorb(*d, i);
    a.radius = auto_request(d->e[i].radius);
    if (forward) {
        a.stats = d->counters + (size_t)i * kReachSlots;
        a.dev = (int)(d - g_auto);
        a.key = i;
    }
    return a;
}

// after the forward launch: account its sampled taps and, every few launches, publish
void auto_end(const AutoLaunch &a, const Geom &g, int frames, hipStream_t st)
{
    if (!a.stats) return;
    std::lock_guard<std::mutex> lock(g_auto_mutex);
    AutoDevice &d = g_auto[a.dev];
    AutoEntry &e = d.e[a.key];
    // one row in four of one wave in kReachSample is sampled (one row in two on small grids: those
    // fractions read high)
    e.pending += 0.25 / kReachSample * 25.0 * (double)frames * (double)g.N * (double)g.H * (double)g.W;
    e.calls += 1;
    if (e.calls <= 8 || e.calls % 8 == 0) {
        e.seq += 1;
        e.window[e.seq % 8] = e.pending;
        e.pending = 0.0;
        hipLaunchKernelGGL(reach_publish_kernel, dim3(1), dim3(64), 0, st,
                           d.counters + (size_t)a.key * kReachSlots, d.mirror_dev + a.key, e.seq);
    }
}

// Should this backward call enqueue gradInput's gather kernel, which only pays when every offset
// truncates to zero and otherwise gives up after one memory round trip (13-18 us at 1080p)?  Its
// first workgroup reports a give-up to pinned host memory; after a report the next 1, 2, 4 .. 64
// calls do without the kernel, then it is tried again -- so a training run whose offsets have grown
// past a pixel pays for one attempt in 64, and one that is back to small offsets (16 attempts
// without a report) runs it every time.  The channel is the pinned record of
// ARTEMIS_SYNTH_RADIUS_AUTO; before a forward call with that radius has set it up on the device
// (this function never allocates) every call tries.  No synchronisation anywhere: a report may lag
// by the calls in flight, which costs those attempts, nothing else.
struct GatherTry {
    bool attempt;
    unsigned long long id;
    unsigned long long *report;
};

constexpr size_t asynth_header_bytes() { return kDetHeaderBytes; }

// forget the back-off, and whatever reports of attempts issued so far are still on their way
void gather_forget(AutoDevice *d)
{
    d->ga_skip = d->ga_backoff = d->ga_quiet = 0;
    d->ga_seen = d->ga_seq;
}

GatherTry gather_begin()
{
    GatherTry t{true, 0ull, nullptr};
    std::lock_guard<std::mutex> lock(g_auto_mutex);
    AutoDevice *d = auto_device(false);
    if (!d) return t;
    const unsigned long long rep = ((volatile AutoMirror *)d->mirror)[0].pad;
    if (rep > d->ga_seen) {
        d->ga_seen = rep;
        d->ga_backoff = d->ga_backoff ? (d->ga_backoff < 32 ? d->ga_backoff * 2 : 64) : 1;
        d->ga_skip = d->ga_backoff;
        d->ga_quiet = 0;
    }
    if (d->ga_skip > 0) {
        d->ga_skip -= 1;
        t.attempt = false;
        return t;
    }
    if (++d->ga_quiet >= 16) d->ga_backoff = 0;
    t.id = ++d->ga_seq;
    t.report = &d->mirror_dev[0].pad;
    return t;
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
    AutoLaunch au{0, nullptr, -1, -1};
    if (tile_radius == ARTEMIS_SYNTH_RADIUS_AUTO && tiled_ok(g)) {
        au = auto_begin(g, 0, true);
        tile_radius = au.radius;
    }
    const int R = tiled_radius(g, tile_radius, frame_dtype);
    const bool lean = is_lean(tile_radius);
    if (R > 0) {
        if (frame_dtype == ARTEMIS_SYNTH_FRAME_BF16)
            dispatch_fwd_tiled<asynth::bf16_t>(R, input_any, weight, offset_y, offset_x, output, g, lean,
                                               au.stats, st);
        else
            dispatch_fwd_tiled<float>(R, input_any, weight, offset_y, offset_x, output, g, lean, au.stats,
                                      st);
        auto_end(au, g, 1, st);
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
    // AUTO: the radius the forward launches of this shape have settled on (no statistics here)
    if (tile_radius == ARTEMIS_SYNTH_RADIUS_AUTO && tiled_ok(g)) tile_radius = auto_begin(g, 0, false).radius;

    // gradInput's set-up comes first: its workspace header receives max |weight| and max |gradOutput|
    // (the fixed-point scale) from the tap-gradient kernel when that runs anyway -- one sweep over the
    // weights less (53 us at 1080p) -- and from a launch of its own otherwise.
    const int Rg = (grad_input && tiled_ok(g)) ? gin_radius(tile_radius) : 0;
    const bool fixed = gin_mode == ARTEMIS_SYNTH_GIN_DETERMINISTIC;
    const size_t need = grad_input ? ws_bytes_needed(g, gin_mode, Rg) : 0;
    unsigned char *ws = (unsigned char *)workspace;
    // Offsets that all truncate to zero (a freshly initialised model): gradInput in gather form
    // (synth_gin_gather.hpp).  Whether they do is found out on the device, by the gather kernel itself:
    // it and the scatter kernels are both enqueued, and the scatter kernels (`guarded`) leave at once
    // unless the gather kernel gave up.
    if (need && (!workspace || workspace_bytes < need))
        return fail(hipErrorInvalidValue,
                    "artemis_synth_backward: grad_input needs %zu workspace bytes "
                    "(artemis_synth_backward_workspace_bytes), got %zu", need,
                    workspace ? workspace_bytes : (size_t)0);
    if (need && ((uintptr_t)workspace & 15) != 0)
        return fail(hipErrorInvalidValue, "artemis_synth_backward: workspace not 16-byte aligned");
    GatherPlan plan;
    GatherTry gtry{false, 0ull, nullptr};
    // (asked only once the call is known to go ahead: an attempt counts in the back-off)
    if (grad_input && Rg > 0 && need && g_gin_gather.load(std::memory_order_relaxed) && gather_plan(g, plan))
        gtry = gather_begin();
    const int guarded = gtry.attempt ? 1 : 0;
    if (grad_input) {
        hipError_t e = hipSuccess;
        // (with the gather kernel in front, the int64 image and gradInput are cleared only if the
        // scatter kernels get to run: zero_if_scatter_kernel below)
        if (need) e = hipMemsetAsync(ws, 0, guarded ? asynth_header_bytes() : need, st);
        if (e == hipSuccess && !fixed && !guarded)
            e = hipMemsetAsync(grad_input, 0, gin_count * sizeof(float), st);
        if (e != hipSuccess) return fail((int)e, "hipMemsetAsync: %s", hipGetErrorString(e));
    }
    bool have_maxima = false;

    if (grad_weight || grad_offset_y || grad_offset_x) {
        const int R = tiled_radius(g, tile_radius, frame_dtype);
        const bool lean = is_lean(tile_radius);
        if (R > 0) {
            unsigned *maxima = need ? (unsigned *)ws : nullptr;
            have_maxima = maxima != nullptr;
            if (frame_dtype == ARTEMIS_SYNTH_FRAME_BF16)
                dispatch_bwd_tiled<asynth::bf16_t>(R, input_any, weight, offset_y, offset_x, grad_output,
                                                   grad_weight, grad_offset_y, grad_offset_x, g, lean, maxima,
                                                   st);
            else
                dispatch_bwd_tiled<float>(R, input_any, weight, offset_y, offset_x, grad_output,
                                          grad_weight, grad_offset_y, grad_offset_x, g, lean, maxima, st);
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
        const int bits = count_bits_for(g);
        if (guarded) {
            launch_gin_gather(weight, offset_y, offset_x, grad_output, grad_input, g, plan, (unsigned *)ws,
                              gtry.report, gtry.id, st);
            // what the scatter kernels accumulate into starts at zero (the gather kernel may also have
            // stored part of gradInput before it gave up)
            if (!fixed)
                hipLaunchKernelGGL(zero_if_scatter_kernel, dim3(reduce_blocks(gin_count)), dim3(256), 0, st,
                                   grad_input, gin_count, (const unsigned *)ws);
            else if (need > asynth_header_bytes())
                hipLaunchKernelGGL(zero_if_scatter_kernel, dim3(reduce_blocks(gin_count)), dim3(256), 0, st,
                                   (float *)(ws + asynth_header_bytes()),
                                   (need - asynth_header_bytes()) / sizeof(float), (const unsigned *)ws);
        }
        if (need && !have_maxima) {  // maxima for the fixed-point scale
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
                               grad_output, gcount, ba, vec, (unsigned *)ws, guarded);
        }
        if (Rg > 0) {
            switch (Rg * 2 + (fixed ? 1 : 0)) {
            case 4: launch_gin_tiled<2, false>(weight, offset_y, offset_x, grad_output, grad_input, ws, g, bits, guarded, st); break;
            case 5: launch_gin_tiled<2, true>(weight, offset_y, offset_x, grad_output, grad_input, ws, g, bits, guarded, st); break;
            case 8: launch_gin_tiled<4, false>(weight, offset_y, offset_x, grad_output, grad_input, ws, g, bits, guarded, st); break;
            case 9: launch_gin_tiled<4, true>(weight, offset_y, offset_x, grad_output, grad_input, ws, g, bits, guarded, st); break;
            case 16: launch_gin_tiled<8, false>(weight, offset_y, offset_x, grad_output, grad_input, ws, g, bits, guarded, st); break;
            case 17: launch_gin_tiled<8, true>(weight, offset_y, offset_x, grad_output, grad_input, ws, g, bits, guarded, st); break;
            case 28: launch_gin_tiled<14, false>(weight, offset_y, offset_x, grad_output, grad_input, ws, g, bits, guarded, st); break;
            default: launch_gin_tiled<14, true>(weight, offset_y, offset_x, grad_output, grad_input, ws, g, bits, guarded, st); break;
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
                               ws, grad_input, gin_count, bits, guarded);
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
    // argument errors first, so that they read as artemis_synth_backward_ex reports them
    if (rc > 0) return rc;
    if (const int e = check_frame_dtype(g, frame_dtype, "artemis_synth_backward")) return e;
    if (gin_mode != ARTEMIS_SYNTH_GIN_ATOMIC && gin_mode != ARTEMIS_SYNTH_GIN_DETERMINISTIC)
        return fail(hipErrorInvalidValue, "artemis_synth_backward: unknown gin_mode %d", gin_mode);
    if (rc == 0 && grad_input) {
        // what run_backward will ask for at this tile_radius (ATOMIC with RADIUS_NONE needs nothing
        // and never touches -- or resolves -- the array); AUTO resolves to a radius of at most 24
        const int Rg = tiled_ok(g) ? (tile_radius == ARTEMIS_SYNTH_RADIUS_AUTO ? 1 : gin_radius(tile_radius)) : 0;
        const size_t need = ws_bytes_needed(g, gin_mode, Rg);
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
                      int org_x, int frame_dtype, int tile_radius, int &R, bool &lean, bool forward,
                      AutoLaunch &au)
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
    au = AutoLaunch{0, nullptr, -1, -1};
    if (tile_radius == ARTEMIS_SYNTH_RADIUS_AUTO) {
        au = auto_begin(g, 1, forward);
        tile_radius = au.radius;
    }
    R = pick_radius(tile_radius);
    lean = is_lean(tile_radius);
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
    bool lean = false;
    AutoLaunch au;
    const int rc = head_setup(g, ha, "artemis_synth_head_forward", frames, T, occlusion, B, H, W, in_H,
                              in_W, org_y, org_x, frame_dtype, tile_radius, R, lean, true, au);
    if (rc > 0) return rc;
    if (rc < 0) return 0;
    if (!weights || !offsets_y || !offsets_x || !output)
        return fail(hipErrorInvalidValue, "artemis_synth_head_forward: NULL tensor pointer");
    hipStream_t st = (hipStream_t)stream;
    if (frame_dtype == ARTEMIS_SYNTH_FRAME_BF16)
        dispatch_head_fwd<asynth::bf16_t>(R, weights, offsets_y, offsets_x, output, g, ha, lean, au.stats, st);
    else
        dispatch_head_fwd<float>(R, weights, offsets_y, offsets_x, output, g, ha, lean, au.stats, st);
    auto_end(au, g, T, st);
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
    bool lean = false;
    AutoLaunch au;
    const int rc = head_setup(g, ha, "artemis_synth_head_backward", frames, T, occlusion, B, H, W,
                              in_H, in_W, org_y, org_x, frame_dtype, tile_radius, R, lean, false, au);
    if (rc > 0) return rc;
    if (rc < 0) return 0;
    if (!weights || !offsets_y || !offsets_x || !grad_output)
        return fail(hipErrorInvalidValue, "artemis_synth_head_backward: NULL tensor pointer");
    if (!grad_weights && !grad_offsets_y && !grad_offsets_x && !grad_occlusion) return 0;
    ha.gocc = grad_occlusion;
    hipStream_t st = (hipStream_t)stream;
    if (frame_dtype == ARTEMIS_SYNTH_FRAME_BF16)
        dispatch_head_bwd<asynth::bf16_t>(R, weights, offsets_y, offsets_x, grad_output, grad_weights,
                                          grad_offsets_y, grad_offsets_x, g, ha, lean, st);
    else
        dispatch_head_bwd<float>(R, weights, offsets_y, offsets_x, grad_output, grad_weights,
                                 grad_offsets_y, grad_offsets_x, g, ha, lean, st);
    return check_launch("artemis_synth_head_backward");
}

int artemis_synth_auto_query(int H, int W, int head, int *radius, double *frac8, double *frac14,
                             unsigned long long *forward_calls)
{
    std::lock_guard<std::mutex> lock(g_auto_mutex);
    int R = kDefaultRadius;
    double f8 = 0.0, f14 = 0.0;
    unsigned long long calls = 0;
    if (AutoDevice *d = auto_device(false)) {
        const int i = auto_find(*d, H, W, head ? 1 : 0, false);
        if (i >= 0) {
            auto_absorb(*d, i);
            R = d->e[i].radius, f8 = d->e[i].f8, f14 = d->e[i].f14, calls = d->e[i].calls;
        }
    }
    if (radius) *radius = R;
    if (frac8) *frac8 = f8;
    if (frac14) *frac14 = f14;
    if (forward_calls) *forward_calls = calls;
    return 0;
}

void artemis_synth_auto_reset(void)
{
    std::lock_guard<std::mutex> lock(g_auto_mutex);
    if (AutoDevice *d = auto_device(false)) {
        for (int i = 0; i < kAutoKeys; ++i) d->e[i].used = false;
        gather_forget(d);
    }
}

int artemis_synth_set_gin_gather(int enable)
{
    if (enable < 0) return g_gin_gather.load(std::memory_order_relaxed);
    {  // (a change of the setting also forgets the current device's back-off: the next call tries)
        std::lock_guard<std::mutex> lock(g_auto_mutex);
        if (AutoDevice *d = auto_device(false)) gather_forget(d);
    }
    return g_gin_gather.exchange(enable ? 1 : 0, std::memory_order_relaxed);
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
