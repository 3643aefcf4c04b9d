// synth_tiles.hip -- the LDS-staged kernels for ONE tile radius (-DSYNTH_TILE_R=2|4|8|14|24):
// explicit instantiations of the launchers declared in synth_device.hpp.  One object per radius
// lets the heavily unrolled tap-major kernels compile in parallel.
#include "synth_device.hpp"

#ifndef SYNTH_TILE_R
#error "compile with -DSYNTH_TILE_R=<radius>"
#endif
// -DSYNTH_TILE_BF16=1: the bf16-frame instantiations of the same radius (a separate object);
// gradInput does not read the frame and lives in the fp32 object only.
#ifndef SYNTH_TILE_BF16
#define SYNTH_TILE_BF16 0
#endif

namespace {

// Tile configuration (tools/ab.py compares builds in one process; the SYNTH_* macros override the
// shape for such experiments).  Every radius: 8 waves as 2 across x 4 down, 4 rows per wave =
// 128 x 16 pixels, ONE workgroup per CU (the small radii are padded with dynamic LDS so that a second
// workgroup does not fit).  Measured with stream-only stand-ins (tools/exp/exp_stream.hip,
// backward-like, 1080p): 64 x 32 tiles, two workgroups per CU 240 us; one per CU 231 us; 128 x 16,
// one per CU 224 us; 128 x 16, two per CU 241 us.
// The 128 pixels are served by NWIN = 2 staged windows of 64 columns each, every one with its own
// centres (synth_device.hpp, "Several windows per tile"): the plane rows of a workgroup stay 512 B
// long, which the DRAM wants, while a window only has to follow the motion of 64 x 16 pixels.
// LDS: 2 x (16 + 5 + 2R) x (64 + 5 + 2R) x 16 B = 51 / 71 / 101 / 152 KB for R = 2 / 4 / 8 / 14 -- which
// is why the largest radius is 14, not 16: two R = 16 windows would need 171 KB.
#ifndef SYNTH_WAVES
#define SYNTH_WAVES 8
#endif
#ifndef SYNTH_WX
#define SYNTH_WX 2
#endif
#ifndef SYNTH_NWIN
#define SYNTH_NWIN 2
#endif
#ifndef SYNTH_ROWS
#define SYNTH_ROWS 4
#endif
#ifndef SYNTH_OCC
#define SYNTH_OCC 2
#endif
#ifndef SYNTH_ONE_WG_PER_CU
#define SYNTH_ONE_WG_PER_CU 1
#endif

template <int R>
struct TileCfg {
    static constexpr int WAVES = SYNTH_WAVES, WX = SYNTH_WX, ROWS = SYNTH_ROWS, OCC = SYNTH_OCC,
                         NWIN = SYNTH_NWIN, PASSES = kMaxPasses;
    static constexpr bool kSolo = SYNTH_ONE_WG_PER_CU, DMA = false;
};

// Small grids: a problem with fewer than small_grid_tiles() (default 768) tiles of 64 x 32 pixels (the 64^2 and 128^2
// heads of a small-batch training step, BASELINE configs[0] and [2]) cannot fill 256 CUs and is
// bound by one workgroup's serial time.  It runs with 64 x 8 tiles instead (4 waves x 2 rows):
// 4x the workgroups, each a quarter of the work.  Same per-pixel arithmetic, same bits.

// The backward (reads and writes 75 planes each) runs best with still fewer waves in flight: the same
// 128 x 16 tile by 4 waves (2 across x 2 down, 8 rows each), one workgroup per CU: 242 -> 237 us
// at 1080p; the forward loses with that shape (108 -> 127 us) and keeps TileCfg.
#ifndef SYNTH_BWD_WAVES
#define SYNTH_BWD_WAVES 4
#endif
#ifndef SYNTH_BWD_ROWS
#define SYNTH_BWD_ROWS 8
#endif
template <int R>
struct BwdCfg {
    // OCC 1: one workgroup of 4 waves per CU (kSolo) is one wave per SIMD, so the kernel may use the
    // whole 512-register file; with 2 the compiler held the small radii (whose LDS would admit a
    // second workgroup) to 256 registers and spilled 60-240 of them
    static constexpr int WAVES = SYNTH_BWD_WAVES, WX = SYNTH_WX, ROWS = SYNTH_BWD_ROWS, OCC = 1,
                         NWIN = SYNTH_NWIN, PASSES = kMaxPasses;
    static constexpr bool kSolo = true, DMA = false;
};

// The LEAN shape (tile_radius ARTEMIS_SYNTH_RADIUS_SMALL; what ARTEMIS_SYNTH_RADIUS_AUTO runs once the
// forward kernels have seen practically no tap beyond 8 px of its window's centre): radius 8, the same
// 128 x 16 tile served by ONE 128-column window staged once at the tile's own position -- no centres
// (SYNTH_LEAN_CENTRE), no vote, no further windows; a stray tap still takes the per-pixel global
// fix-up, so any offsets are correct.
// It is round 1's kernel shape: the windows-follow-the-motion machinery costs the i.i.d. small-offset
// regimes 1-2 % on the per-frame kernels and 10 % on the fused head's forward (1080p, sigma 1, same
// process, tools/ab.py: head forward 451-466 us with it at radius 8, 441 without the vote, 419-423 with
// one window, 403 without its centre; per-frame forward / backward 113.5 / 242.8 -> 108 / 238 us).
struct LeanCfg {
    static constexpr int WAVES = 8, WX = 2, ROWS = 4, OCC = 2, NWIN = 1, PASSES = 1;
    // the operand planes through the LDS-DMA ring (synth_device.hpp): window 88 KB + ring 48 KB
    static constexpr bool kSolo = true, DMA = SYNTH_DMA;
};
struct LeanBwdCfg {
    static constexpr int WAVES = 4, WX = 2, ROWS = 8, OCC = 1, NWIN = 1, PASSES = 1;
    static constexpr bool kSolo = true, DMA = false;
};

// Radius 24 (tile_radius > 14, for callers whose offsets are large): one window of
// (32 + 5 + 48) x (64 + 5 + 48) texels = 159 KB serves a 64 x 32 tile, 8 waves x 4 rows in both
// directions -- half the plane-row length of the default shape, three times its displacement area.
// 1080p forward / backward us: sigma = 1 118 / 257 (default shape 116 / 246), sigma = 4 118 / 257
// (131 / 268), steep smooth +-30 px flow 148 / 300 (242 / 498), sigma = 16 731 / 1653 (1305 / 1960).
// (64 x 16 tiles with this window: 141 / 266 at sigma = 1; the backward as 4 waves x 8 rows: 314 on the
// steep flow.)
template <>
struct TileCfg<24> {
    static constexpr int WAVES = 8, WX = 1, ROWS = 4, OCC = 2, NWIN = 1, PASSES = kMaxPasses;
    static constexpr bool kSolo = false, DMA = false;
};
template <>
struct BwdCfg<24> {
    static constexpr int WAVES = 8, WX = 1, ROWS = 4, OCC = 2, NWIN = 1, PASSES = kMaxPasses;
    static constexpr bool kSolo = false, DMA = false;
};

struct SmallCfg {
    static constexpr int WAVES = 4, WX = 1, ROWS = 2, OCC = 2, NWIN = 1, PASSES = kMaxPasses;
    static constexpr bool kSolo = false, DMA = false;
};

// dynamic LDS that keeps a second workgroup off the CU (160 KB per CU) when the config wants that
template <int R, class T>
unsigned lds_pad()
{
    constexpr int TH = (T::WAVES / T::WX) * T::ROWS, SW = 64 * T::WX / T::NWIN;
    constexpr int bytes = T::NWIN * TileShape<R, TH, SW>::LW * TileShape<R, TH, SW>::LH * 16;
    return (T::kSolo && bytes < 82 * 1024) ? (unsigned)(82 * 1024 - bytes) : 0u;
}

// The backward (more work per tile, more sensitive to a ragged last round of workgroups) switches
// at the knob, the forward at a quarter of it: measured crossovers on 256 CUs are ~770 and ~160
// tiles (tools/head_sweep.py).
// the DMA ring moves 16-byte pieces of plane rows: rows and planes must start on 16-byte boundaries
inline bool dma_ok(const asynth::Geom &g, const float *w, const float *oy, const float *ox)
{
    return (g.W & 3) == 0 && (((uintptr_t)w | (uintptr_t)oy | (uintptr_t)ox) & 15) == 0;
}

template <class T>
bool small_grid(const asynth::Geom &g, bool backward)
{
    // counted in 64 x 32 units whatever the big tile's shape is
    const long long tiles = (long long)((g.W + 63) / 64) * ((g.H + 31) / 32) * g.N;
    const long long knob = asynth::small_grid_tiles();
    return tiles < (backward ? knob : (knob + 3) / 4);
}

// workgroups of a config: whole tiles, or (SYNTH_PAIR_SEGMENTS builds, one staged window per wave
// column) the paired-segment mapping of segment_origin
template <class T>
dim3 cfg_grid(const asynth::Geom &g)
{
    return asynth::tile_grid(g, (T::WAVES / T::WX) * T::ROWS, 64 * T::WX,
                             (SYNTH_PAIR_SEGMENTS && T::WX > 1 && T::NWIN == T::WX) ? T::WX : 1);
}

template <int R, class T, class FT>
void fwd_cfg(bool head, const FT *in, const float *w, const float *oy, const float *ox, float *out,
             const asynth::Geom &g, const asynth::HeadArgs &ha, unsigned long long *stats, hipStream_t st)
{
    const dim3 grid = cfg_grid<T>(g), block(64, T::WAVES);
    const unsigned pad = lds_pad<R, T>();
    if (head)
        hipLaunchKernelGGL((synth_fwd_tapmajor<R, T::WAVES, T::ROWS, T::OCC, true, FT, T::WX, T::NWIN, T::PASSES, T::DMA>), grid, block,
                           pad, st, in, w, oy, ox, out, g, ha, stats);
    else
        hipLaunchKernelGGL((synth_fwd_tapmajor<R, T::WAVES, T::ROWS, T::OCC, false, FT, T::WX, T::NWIN, T::PASSES, T::DMA>), grid, block,
                           pad, st, in, w, oy, ox, out, g, ha, stats);
}

template <int R, class T, class FT>
void bwd_cfg(bool head, const FT *in, const float *w, const float *oy, const float *ox,
             const float *gout, float *gw, float *goy, float *gox, const asynth::Geom &g,
             const asynth::HeadArgs &ha, unsigned *maxima, hipStream_t st)
{
    const dim3 grid = cfg_grid<T>(g), block(64, T::WAVES);
    const unsigned pad = lds_pad<R, T>();
    const bool all = gw && goy && gox;
    if (head) {
        if (all)
            hipLaunchKernelGGL((synth_bwd_tapmajor<R, T::WAVES, T::ROWS, T::OCC, true, true, FT, T::WX, T::NWIN, T::PASSES>), grid,
                               block, pad, st, in, w, oy, ox, gout, gw, goy, gox, g, ha, maxima);
        else
            hipLaunchKernelGGL((synth_bwd_tapmajor<R, T::WAVES, T::ROWS, T::OCC, false, true, FT, T::WX, T::NWIN, T::PASSES>), grid,
                               block, pad, st, in, w, oy, ox, gout, gw, goy, gox, g, ha, maxima);
    } else {
        if (all)
            hipLaunchKernelGGL((synth_bwd_tapmajor<R, T::WAVES, T::ROWS, T::OCC, true, false, FT, T::WX, T::NWIN, T::PASSES>), grid,
                               block, pad, st, in, w, oy, ox, gout, gw, goy, gox, g, ha, maxima);
        else
            hipLaunchKernelGGL((synth_bwd_tapmajor<R, T::WAVES, T::ROWS, T::OCC, false, false, FT, T::WX, T::NWIN, T::PASSES>), grid,
                               block, pad, st, in, w, oy, ox, gout, gw, goy, gox, g, ha, maxima);
    }
}

}  // namespace

namespace asynth {

template <int R, class FT>
void launch_fwd_tiled(const FT *in, const float *w, const float *oy, const float *ox, float *out,
                      const Geom &g, bool lean, unsigned long long *stats, hipStream_t st)
{
    if (small_grid<TileCfg<R>>(g, false)) {
        fwd_cfg<R, SmallCfg, FT>(false, in, w, oy, ox, out, g, HeadArgs{}, stats, st);
        return;
    }
    if constexpr (R == 8) {  // (the lean shape is built for radius 8 only)
        if (lean && (!LeanCfg::DMA || dma_ok(g, w, oy, ox))) {
            fwd_cfg<R, LeanCfg, FT>(false, in, w, oy, ox, out, g, HeadArgs{}, stats, st);
            return;
        }
    }
    fwd_cfg<R, TileCfg<R>, FT>(false, in, w, oy, ox, out, g, HeadArgs{}, stats, st);
}

template <int R, class FT>
void launch_bwd_tiled(const FT *in, const float *w, const float *oy, const float *ox,
                      const float *gout, float *gw, float *goy, float *gox, const Geom &g, bool lean,
                      unsigned *maxima, hipStream_t st)
{
    if (small_grid<TileCfg<R>>(g, true)) {
        bwd_cfg<R, SmallCfg, FT>(false, in, w, oy, ox, gout, gw, goy, gox, g, HeadArgs{}, maxima, st);
        return;
    }
    if constexpr (R == 8) {
        if (lean) {
            bwd_cfg<R, LeanBwdCfg, FT>(false, in, w, oy, ox, gout, gw, goy, gox, g, HeadArgs{}, maxima, st);
            return;
        }
    }
    bwd_cfg<R, BwdCfg<R>, FT>(false, in, w, oy, ox, gout, gw, goy, gox, g, HeadArgs{}, maxima, st);
}

template <int R, class FT>
void launch_head_fwd(const float *w, const float *oy, const float *ox, float *out, const Geom &g,
                     const HeadArgs &ha, bool lean, unsigned long long *stats, hipStream_t st)
{
    if (small_grid<TileCfg<R>>(g, false)) {
        fwd_cfg<R, SmallCfg, FT>(true, (const FT *)nullptr, w, oy, ox, out, g, ha, stats, st);
        return;
    }
    if constexpr (R == 8) {
        if (lean && (!LeanCfg::DMA || dma_ok(g, w, oy, ox))) {
            fwd_cfg<R, LeanCfg, FT>(true, (const FT *)nullptr, w, oy, ox, out, g, ha, stats, st);
            return;
        }
    }
    fwd_cfg<R, TileCfg<R>, FT>(true, (const FT *)nullptr, w, oy, ox, out, g, ha, stats, st);
}

template <int R, class FT>
void launch_head_bwd(const float *w, const float *oy, const float *ox, const float *gout, float *gw,
                     float *goy, float *gox, const Geom &g, const HeadArgs &ha, bool lean, hipStream_t st)
{
    if (small_grid<TileCfg<R>>(g, true)) {
        bwd_cfg<R, SmallCfg, FT>(true, (const FT *)nullptr, w, oy, ox, gout, gw, goy, gox, g, ha, nullptr, st);
        return;
    }
    if constexpr (R == 8) {
        if (lean) {
            bwd_cfg<R, LeanBwdCfg, FT>(true, (const FT *)nullptr, w, oy, ox, gout, gw, goy, gox, g, ha, nullptr, st);
            return;
        }
    }
    bwd_cfg<R, BwdCfg<R>, FT>(true, (const FT *)nullptr, w, oy, ox, gout, gw, goy, gox, g, ha, nullptr, st);
}

// gradInput tiles: 64 x 16 pixels, 64-bit accumulators (3 x 29 x 77 x 8 B = 53.6 KB at R = 4,
// 114 KB at R = 14: one workgroup per CU)
template <int R, bool FIXED>
void launch_gin_tiled(const float *w, const float *oy, const float *ox, const float *gout, float *gin,
                      unsigned char *ws, const Geom &g, int bits, int guarded, hipStream_t st)
{
    // 64 x 16 tiles; at the largest radius the tile takes most of the LDS (one workgroup per CU), so it
    // gets 8 waves of 2 rows instead of 4 waves of 4 rows (415 -> 385 us at 1080p)
    constexpr int WAVES = R > 8 ? 8 : 4, ROWS = R > 8 ? 2 : 4;
    hipLaunchKernelGGL((synth_gin_tiled<R, WAVES, ROWS, FIXED>), pixel_grid(g, WAVES * ROWS),
                       dim3(64, WAVES), 0, st, w, oy, ox, gout, gin, ws, g, bits, guarded);
}


#if defined(SYNTH_DEV_TILE_CLOCK) && SYNTH_TILE_R == 14 && !SYNTH_TILE_BF16
}  // namespace asynth
extern "C" int artemis_synth_dev_set_tile_log(void *p)
{
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_tile_log), &p, sizeof(p));
}
namespace asynth {
#endif

#if SYNTH_TILE_BF16
typedef bf16_t frame_t;
#else
typedef float frame_t;
#endif

template void launch_fwd_tiled<SYNTH_TILE_R, frame_t>(const frame_t *, const float *, const float *,
                                                      const float *, float *, const Geom &, bool,
                                                      unsigned long long *, hipStream_t);
#ifndef SYNTH_DEV_ONLY_FWD  // (developer builds: compile just the forward to look at its registers)
template void launch_bwd_tiled<SYNTH_TILE_R, frame_t>(const frame_t *, const float *, const float *,
                                                      const float *, const float *, float *, float *,
                                                      float *, const Geom &, bool, unsigned *, hipStream_t);
template void launch_head_fwd<SYNTH_TILE_R, frame_t>(const float *, const float *, const float *,
                                                     float *, const Geom &, const HeadArgs &, bool,
                                                     unsigned long long *, hipStream_t);
template void launch_head_bwd<SYNTH_TILE_R, frame_t>(const float *, const float *, const float *,
                                                     const float *, float *, float *, float *,
                                                     const Geom &, const HeadArgs &, bool, hipStream_t);
#if !SYNTH_TILE_BF16 && SYNTH_TILE_R <= 14  // (gradInput's 64-bit accumulator window tops out at R = 14)
template void launch_gin_tiled<SYNTH_TILE_R, false>(const float *, const float *, const float *,
                                                    const float *, float *, unsigned char *,
                                                    const Geom &, int, int, hipStream_t);
template void launch_gin_tiled<SYNTH_TILE_R, true>(const float *, const float *, const float *,
                                                   const float *, float *, unsigned char *,
                                                   const Geom &, int, int, hipStream_t);
#endif
#endif  // SYNTH_DEV_ONLY_FWD

}  // namespace asynth
