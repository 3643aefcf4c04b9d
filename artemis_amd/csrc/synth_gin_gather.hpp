// synth_gin_gather.hpp -- gradInput in GATHER form (k = 5, d = 1, C = 3), for offsets that all
// truncate to zero.  Included by synth_kernels.hip after synth_device.hpp, inside its anonymous
// namespace.  (No reference counterpart: upstream returns unfilled zeros, cupy_module/synth.py:333,391.)
//
// The adjoint of the forward adds, per pixel (y, x) and tap (r, s), gradOut_c * w * coef to the four
// clamped corners (y + oy0 + r + trunc(a) + dy, x + ox0 + s + trunc(b) + dx).  Where a texel's
// contributions come from therefore depends on the offsets -- except when every offset truncates to
// zero (|a|, |b| < 1: a freshly initialised model, offsets N(0, 0.045)): then texel (vy, vx) receives
// exactly from the pixels (vy - oy0 - r - dy, vx - ox0 - s - dx), 100 (tap, corner) terms, and the
// scatter (300 LDS integer atomics per pixel, synth_gin_tiled) turns into a gather with static
// addresses: a lane owns a texel, sums its terms in registers in a fixed order (bit-reproducible in
// either gin_mode, no fixed-point grid, no memset, no finalize pass) and stores it once.  For other
// offsets the gather form needs (2D + 1)^2 candidate pixels per texel and tap for displacements up
// to D and loses to the scatter.  So this kernel runs OPTIMISTICALLY in front of the
// scatter kernels: it checks every offset it loads, and the first wave that meets one that does not
// truncate to zero raises word kGaFlagWord of the workspace header and gives up; workgroups look at
// that word before they start.  The scatter kernels (and the zero-fill / maxima passes they need),
// enqueued behind it, leave at once while the word is still 0 and redo everything otherwise.  With
// i.i.d. offsets of a pixel or more every workgroup gives up before it stages anything (13-18 us for a
// 1080p frame), and the host, told through pinned memory, enqueues the kernel only once in a while
// as long as that goes on (gather_begin in synth_kernels.hip).
//
// Layout: a workgroup owns 63 WX x 4 (8 / WX) virtual texels (8 waves, 4 rows each; lanes 1..63 are
// texel columns, lane 0 the column left of them; WX = 2: 126 x 16, WX = 1: 63 x 32 for narrow frames),
// stages gradOut for the pixels those can receive from (zero outside the image) in LDS as float4 and
// walks the taps in the outer loop like the other kernels: per tap a wave loads the tap's (w, a, b)
// for 5 pixel rows at column x - s (the dx = 0 corner's pixel) -- 15 coalesced dword loads -- and takes
// column x - s - 1 (dx = 1) from the lane to its left with a DPP wave shift, which is what lane 0 is
// there for; pixel row i feeds texel row i - 1 through its dy = 0 coefficients and texel row i through
// its dy = 1 ones.  The planes come about once from HBM (the 5th row and the lines shared with
// neighbouring waves are L1 / L2 hits): 324 B per texel, the forward's traffic.
// What the kernel's time did and did not depend on (1080p, random init): fetching both
// columns with vector loads (30 per tap and wave) 188 us, one column 168 us; the tap loop in 25 basic blocks, which let the optimizer sink every
// tap's loads to their first use (no tap fetched ahead), same time as one block with the rolling
// refill below; 1, 2 or 3 workgroups per CU, two taps ahead, the window fill in one batch of loads,
// half the arithmetic, no LDS reads: all within 3 us.  Non-temporal loads: +30 us.  Persistent
// workgroups on per-XCD tile queues: +26 us.  4-wave workgroups: -6 % but 8 us more to give up.
//
// Clamping: a corner index is clamped to the frame, so the texels of the frame's border also collect
// what falls on the VIRTUAL rows / columns outside it (one row and column past the padded frame in the
// reference layout, two to three on each side for unpadded frames).  The tiles are laid over the
// virtual range, the first tile starts at its first row / column and the last one ends at its last
// (overlapping its neighbour, which computes the same sums in the same order), so that a border texel
// and the virtual texels that fold onto it sit in ONE wave (rows) and in one wave's lanes (columns)
// and are summed there in a fixed order; only that wave stores the border texel.
#ifndef ARTEMIS_SYNTH_GIN_GATHER_HPP
#define ARTEMIS_SYNTH_GIN_GATHER_HPP

constexpr int kGaRows = 4, kGaWaves = 8;
constexpr int kGaCols = 63;  // texel columns of a wave: lanes 1..63; lane 0 fetches the column left of them
constexpr int kGaMaxFold = 3;  // virtual rows / columns beyond a frame edge (<= kGaRows - 1)
template <int WX>
struct GaShape {
    static constexpr int TX = kGaCols * WX, TY = kGaRows * (kGaWaves / WX);
    static constexpr int LW = TX + 5, LH = TY + 5;
};

// The processed (virtual) texel range and its tiling; host-computed.
struct GatherPlan {
    int ry0, ry1, rx0, rx1;  // rows [ry0, ry1] x columns [rx0, rx1]: the frame and every corner position
    int tiles_y, tiles_x;
    int wx;                  // GaShape<wx>
};

// false: this geometry takes the scatter kernels (tiny frames, origins far outside the frame)
inline bool gather_plan(const Geom &g, GatherPlan &p)
{
    if (g.k != 5 || g.d != 1 || g.C != 3 || g.H < 2 || g.W < 2) return false;
    p.ry0 = g.oy0 < 0 ? g.oy0 : 0;
    p.rx0 = g.ox0 < 0 ? g.ox0 : 0;
    const long long ly = (long long)g.oy0 + g.H + 4, lx = (long long)g.ox0 + g.W + 4;  // last corner row / column
    if (ly > 0x3fffffff || lx > 0x3fffffff) return false;
    p.ry1 = (int)ly > g.Hin - 1 ? (int)ly : g.Hin - 1;
    p.rx1 = (int)lx > g.Win - 1 ? (int)lx : g.Win - 1;
    if (-p.ry0 > kGaMaxFold || -p.rx0 > kGaMaxFold) return false;
    if (p.ry1 - (g.Hin - 1) > kGaMaxFold || p.rx1 - (g.Win - 1) > kGaMaxFold) return false;
    const int nr = p.ry1 - p.ry0 + 1, nc = p.rx1 - p.rx0 + 1;
    p.wx = nc >= GaShape<2>::TX ? 2 : 1;
    const int ty = p.wx == 2 ? GaShape<2>::TY : GaShape<1>::TY, tx = kGaCols * p.wx;
    if (nr < ty || nc < tx) return false;
    p.tiles_y = (nr + ty - 1) / ty;
    p.tiles_x = (nc + tx - 1) / tx;
    if ((long long)p.tiles_x * p.tiles_y * g.N > 0x3fffffff) return false;
    return true;
}

// One tap's operands for 5 pixel rows at column x - s.
struct GaTap {
    float wA[kGaRows + 1], aA[kGaRows + 1], bA[kGaRows + 1];
};

// the value of the lane to the left (lane 0: 0): column x - s - 1 from the loads of column x - s
__device__ __forceinline__ float ga_left(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138 /* wave_shr:1 */,
                                                                 0xf, 0xf, true));
}

template <int WX>
__global__ __launch_bounds__(64 * kGaWaves) void synth_gin_gather(
    const float *__restrict__ w, const float *__restrict__ oy, const float *__restrict__ ox,
    const float *__restrict__ gout, float *__restrict__ gin, Geom g, GatherPlan p,
    unsigned *__restrict__ hdr, unsigned long long *report, unsigned long long attempt)
{
    constexpr int TX = GaShape<WX>::TX, TY = GaShape<WX>::TY, LW = GaShape<WX>::LW, LH = GaShape<WX>::LH;
    __shared__ f32x4 win[LH * LW];

    const int l = threadIdx.x, wv = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int wyv = wv / WX, wxv = wv % WX;  // the wave's row block and column block in the tile
    const int tid = wv * 64 + l;
    const size_t plane = (size_t)g.H * g.W, iplane = (size_t)g.Hin * g.Win;
    unsigned *flag = hdr + kGaFlagWord;
    // XCD i (workgroup id % 8) takes the i-th band of tiles
    const unsigned nb = (unsigned)(p.tiles_x * p.tiles_y), per = (nb + 7) / 8;
    const unsigned t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (t >= nb) return;
    const int n = blockIdx.z;
    {
        const int ty = (int)(t / (unsigned)p.tiles_x), tx = (int)(t - (unsigned)ty * (unsigned)p.tiles_x);
        const int rs = min(p.ry0 + TY * ty, p.ry1 - (TY - 1));  // the last tile ends at the last row
        const int xs = min(p.rx0 + TX * tx, p.rx1 - (TX - 1));
        const int f0 = rs + kGaRows * wyv;  // the wave's first (virtual) texel row
        const int xw = xs + kGaCols * wxv;  // the wave's first (virtual) texel column: lane 1's
        const int vx = xw + l - 1;          // the lane's (lane 0: the column left of the wave's)
        int yb = f0 - g.oy0;                // pixel row of texel row f0 for r = 0, dy = 0
        int xb = vx - g.ox0;                // pixel column for s = 0, dx = 0
        // A wave none of whose texels can receive anything (row strips: the frame's rows far from
        // the strip) stores zeros without streaming a plane.
        const bool live = yb + kGaRows - 1 >= 0 && yb - 5 <= g.H - 1;  // wave-uniform
        const float *wb = w + (size_t)n * 25 * plane, *ab = oy + (size_t)n * 25 * plane,
                    *bb = ox + (size_t)n * 25 * plane;

        auto load_row = [&](GaTap &c, int tap, int i) {
            const int r = tap / 5, s = tap - r * 5;
            const unsigned xa = (unsigned)clampi(xb - s, 0, g.W - 1);
            const size_t ro = (size_t)tap * plane + (size_t)clampi(yb - r - 1 + i, 0, g.H - 1) * g.W;  // wave-uniform
            // (plain loads: the rows a wave shares with the wave above and the lines it shares with
            // its neighbours come from L1 / L2; non-temporal loads cost 30 us)
            c.wA[i] = (wb + ro)[xa];
            c.aA[i] = (ab + ro)[xa];
            c.bA[i] = (bb + ro)[xa];
        };
        auto load_tap = [&](GaTap &c, int tap) {
#pragma unroll
            for (int i = 0; i <= kGaRows; ++i) load_row(c, tap, i);
        };

        // Is this the gather kernel's call at all?  Before anything is staged, every wave looks at
        // the first tap's offsets of its own pixels (fetched here anyway) and thread 0 at the word
        // other workgroups raise; with offsets of a pixel or more the whole grid is gone after one
        // memory round trip.  The verdict is shared through LDS: the barriers below need every wave.
        __shared__ unsigned verdict[kGaWaves + 1];
        GaTap cur;
        {
            unsigned big = 0u;
            if (live) {
                load_tap(cur, 0);
#pragma unroll
                for (int i = 0; i <= kGaRows; ++i) big = max(big, max(absbits(cur.aA[i]), absbits(cur.bA[i])));
            }
            const bool bad = __ballot(big >= kOneBits) != 0ull;
            if (l == 0) verdict[wv] = bad ? kOneBits : 0u;
            if (tid == 64 * kGaWaves - 1)
                verdict[kGaWaves] = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            unsigned own = 0u;
#pragma unroll
            for (int i = 0; i < kGaWaves; ++i) own |= verdict[i];
            if ((own | verdict[kGaWaves]) >= kOneBits) {  // workgroup-uniform
                // one plain store per workgroup, and only while the word still read 0: thousands of
                // atomics on one address serialise in the L2 (40 us when every wave raised it)
                if (tid == 0 && own >= kOneBits && verdict[kGaWaves] < kOneBits)
                    __hip_atomic_store(flag, kOneBits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // ... and the first workgroup tells the host (pinned memory, read without a
                // synchronisation before later calls), which then stops enqueueing this kernel for
                // a while (gather_begin in synth_kernels.hip)
                if (tid == 0 && report && blockIdx.x == 0 && blockIdx.z == 0)
                    __hip_atomic_store(report, attempt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                return;
            }
        }

        // (what the 25 taps' addresses and masks are computed from is opaque up to here: they are
        // speculatable arithmetic, and computed ahead of the verdict they cost every wave of a grid
        // that gives up a thousand instructions -- 23 instead of 18 us for a 1080p frame)
        asm volatile("" : "+s"(yb), "+v"(xb));

        // gradOut of the pixels this tile's texels receive from, zero outside the image
        {
            const int py0 = rs - g.oy0 - 5, px0 = xs - g.ox0 - 5;
            const float *gp = gout + (size_t)n * 3 * plane;
            for (int i = tid; i < LH * LW; i += 64 * kGaWaves) {
                const int ly = i / LW, lx = i - ly * LW;
                const int y = py0 + ly, x = px0 + lx;
                const bool ok = (unsigned)y < (unsigned)g.H && (unsigned)x < (unsigned)g.W;
                const size_t o = (size_t)clampi(y, 0, g.H - 1) * g.W + clampi(x, 0, g.W - 1);
                const float g0 = gp[o], g1 = gp[plane + o], g2 = gp[2 * plane + o];
                win[i] = ok ? f32x4{g0, g1, g2, 0.f} : f32x4{0.f, 0.f, 0.f, 0.f};
            }
        }
        __syncthreads();

        float acc[kGaRows][3];
#pragma unroll
        for (int j = 0; j < kGaRows; ++j) acc[j][0] = acc[j][1] = acc[j][2] = 0.f;
        bool gave_up = false;  // wave-uniform

        if (live) {
            const lds_f32x4 *lwin =
                (const lds_f32x4 *)((const __attribute__((address_space(3))) char *)(const lds_f32x4 *)win +
                                    ((kGaRows * wyv) * LW + kGaCols * wxv + max(l - 1, 0)) * 16);
            // The 25 taps are ONE basic block: no branch, no store, one give-up test after the last.
            // (With a test after every tap the optimizer sank each tap's loads out of the block that
            // issues them into the block that uses them, so that no tap was fetched ahead.  The first
            // tap's offsets were checked above; what only a later tap reveals costs this workgroup its
            // 25 taps and nothing else.)
            unsigned big = 0u;  // largest |offset| of the A columns, as a bit pattern (NaN above all)
            // Fetching ahead is row by row: as soon as a pixel row's operands have been used, its
            // registers receive the same row of the next tap, so one tap's worth of loads is always
            // in flight in 15 registers, not 30 (73 VGPRs: three workgroups per CU).
            static_for<0, 25>([&](auto tc) {
                constexpr int tap = decltype(tc)::value;
                constexpr int r = tap / 5, s = tap % 5;
                const bool okA = (unsigned)(xb - s) < (unsigned)g.W, okB = (unsigned)(xb - s - 1) < (unsigned)g.W;
                static_for<0, kGaRows + 1>([&](auto ic) {
                    constexpr int i = decltype(ic)::value;
                    const bool rowok = (unsigned)(yb - r - 1 + i) < (unsigned)g.H;  // wave-uniform
                    const bool vA = rowok && okA, vB = rowok && okB;
                    // a pixel outside the image contributes nothing: selects, so that whatever sits at
                    // the clamped address (Inf, NaN) cannot reach a texel the reference leaves finite
                    const float wA = vA ? cur.wA[i] : 0.f, aA = vA ? cur.aA[i] : 0.f, bA = vA ? cur.bA[i] : 0.f;
                    // (shifted first, selected afterwards: a DPP move is convergent and would turn a
                    // conditional into a branch)
                    const float wL = ga_left(cur.wA[i]), aL = ga_left(cur.aA[i]), bL = ga_left(cur.bA[i]);
                    const float wB = vB ? wL : 0.f, aB = vB ? aL : 0.f, bB = vB ? bL : 0.f;
                    // (every (pixel, tap) is some texel's A column, so checking that column checks all)
                    big = max(big, max(absbits(aA), absbits(bA)));
                    // trunc = 0: the fractions are the offsets themselves.  Column x - s is this texel's
                    // dx = 0 corner (coefficient 1 - b), column x - s - 1 its dx = 1 corner (b); this
                    // pixel row is dy = 0 (1 - a) for texel row i - 1 and dy = 1 (a) for texel row i.
                    const float tA = wA * (1.f - bA), tB = wB * bB;
                    const float a1 = tA * aA, a0 = tA * (1.f - aA);
                    const float b1 = tB * aB, b0 = tB * (1.f - aB);
                    constexpr int imm = (4 - r + i) * LW + 5 - s;
                    const f32x4 gA = lwin[imm], gB = lwin[imm - 1];
                    if constexpr (i >= 1) {
                        acc[i - 1][0] += a0 * gA.x + b0 * gB.x;
                        acc[i - 1][1] += a0 * gA.y + b0 * gB.y;
                        acc[i - 1][2] += a0 * gA.z + b0 * gB.z;
                    }
                    if constexpr (i < kGaRows) {
                        acc[i][0] += a1 * gA.x + b1 * gB.x;
                        acc[i][1] += a1 * gA.y + b1 * gB.y;
                        acc[i][2] += a1 * gA.z + b1 * gB.z;
                    }
                    // pin this row's arithmetic above, the refill below (else the scheduler hoists
                    // every refill to the top of the block and both taps' operands are live at once)
#pragma unroll
                    for (int j = 0; j < kGaRows; ++j) asm volatile("" : "+v"(acc[j][0]), "+v"(acc[j][1]), "+v"(acc[j][2]));
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (tap < 24) load_row(cur, tap + 1, i);
                    __builtin_amdgcn_sched_barrier(0);
                });
            });
            if (__ballot(big >= kOneBits)) {  // an offset that does not truncate to zero: not this kernel's call
                if (l == 0) __hip_atomic_store(flag, kOneBits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                gave_up = true;
            }

            // Virtual rows above the frame fold onto row 0, those below it onto the last row
            // (wave-uniform, border waves only); fixed order.
            if (f0 < 0) {
                static_for<1, kGaRows>([&](auto jt) {
                    constexpr int j0 = decltype(jt)::value;
                    if (-f0 == j0) {
#pragma unroll
                        for (int j = j0 - 1; j >= 0; --j)
#pragma unroll
                            for (int c = 0; c < 3; ++c) acc[j0][c] += acc[j][c];
                    }
                });
            }
            if (f0 + kGaRows - 1 > g.Hin - 1) {
                static_for<0, kGaRows - 1>([&](auto jt) {
                    constexpr int jb = decltype(jt)::value;
                    if (g.Hin - 1 - f0 == jb) {
#pragma unroll
                        for (int j = jb + 1; j < kGaRows; ++j)
#pragma unroll
                            for (int c = 0; c < 3; ++c) acc[jb][c] += acc[j][c];
                    }
                });
            }
            // ... and virtual columns left / right of the frame onto its first / last column (the
            // lanes in question sit in this wave: the first tile starts at rx0, the last ends at rx1)
            if (xw < 0) {  // wave-uniform
                const int l0 = 1 - xw;  // the lane of column 0; columns xw .. -1 sit in lanes 1 .. l0 - 1
#pragma unroll
                for (int j = 0; j < kGaRows; ++j)
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float v = acc[j][c];
                        float sum = v;
#pragma unroll
                        for (int q = 1; q <= kGaMaxFold; ++q) {
                            const float o = __shfl_up(v, q);
                            sum = (l == l0 && q <= l0 - 1) ? sum + o : sum;
                        }
                        acc[j][c] = sum;
                    }
            }
            if (xw + kGaCols - 1 > g.Win - 1) {
                const int lb = g.Win - xw;  // the lane of the last column
#pragma unroll
                for (int j = 0; j < kGaRows; ++j)
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float v = acc[j][c];
                        float sum = v;
#pragma unroll
                        for (int q = 1; q <= kGaMaxFold; ++q) {
                            const float o = __shfl_down(v, q);
                            sum = (l == lb && lb + q <= 63) ? sum + o : sum;
                        }
                        acc[j][c] = sum;
                    }
            }
        }

        // A border texel is stored by the wave that holds everything folding onto it -- the first
        // tile's first wave / lanes, the last tile's last -- other texels by every wave that computes
        // them (overlapping tiles: the same sum in the same order).  (A wave that gave up stores
        // nothing: the scatter kernels rewrite all of gradInput.)
        const bool col_ok = l >= 1 && vx >= 0 && vx <= g.Win - 1 && (vx != 0 || p.rx0 == 0 || xw == p.rx0) &&
                            (vx != g.Win - 1 || p.rx1 == g.Win - 1 || xw + kGaCols - 1 == p.rx1);
        float *gp = gin + (size_t)n * 3 * iplane;
#pragma unroll
        for (int j = 0; j < kGaRows; ++j) {
            const int vy = f0 + j;
            const bool row_ok = vy >= 0 && vy <= g.Hin - 1 && (vy != 0 || p.ry0 == 0 || f0 == p.ry0) &&
                                (vy != g.Hin - 1 || p.ry1 == g.Hin - 1 || f0 + kGaRows - 1 == p.ry1);
            if (row_ok && col_ok && !gave_up) {
                float *q = gp + (size_t)vy * g.Win + vx;
                q[0] = acc[j][0];
                q[iplane] = acc[j][1];
                q[2 * iplane] = acc[j][2];
            }
        }
    }
}

// gradInput <- 0 for the scatter kernels of mode ATOMIC when they, not the gather kernel, do the work
__global__ __launch_bounds__(256) void zero_if_scatter_kernel(float *__restrict__ gin, size_t count,
                                                               const unsigned *__restrict__ hdr)
{
    if (hdr[kGaFlagWord] < kOneBits) return;
    typedef float f4 __attribute__((ext_vector_type(4)));
    const size_t n4 = (((uintptr_t)gin & 15) == 0) ? count / 4 : 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        ((f4 *)gin)[i] = f4{0.f, 0.f, 0.f, 0.f};
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
         i += (size_t)gridDim.x * blockDim.x)
        gin[i] = 0.f;
}

inline void launch_gin_gather(const float *w, const float *oy, const float *ox, const float *gout,
                              float *gin, const Geom &g, const GatherPlan &p, unsigned *hdr,
                              unsigned long long *report, unsigned long long attempt, hipStream_t st)
{
    const unsigned nb = (unsigned)(p.tiles_x * p.tiles_y);
    const dim3 grid(((nb + 7) / 8) * 8, 1, (unsigned)g.N), block(64, kGaWaves);
    if (p.wx == 2)
        hipLaunchKernelGGL(synth_gin_gather<2>, grid, block, 0, st, w, oy, ox, gout, gin, g, p, hdr, report, attempt);
    else
        hipLaunchKernelGGL(synth_gin_gather<1>, grid, block, 0, st, w, oy, ox, gout, gin, g, p, hdr, report, attempt);
}

#endif
