// Experiment: pure streaming of the 75 planes in the product's access pattern, to find how
// achieved HBM bandwidth depends on bytes in flight (prefetch depth x occupancy) and load width.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// DEPTH = tap rows (15 loads each) in flight ahead of the one being consumed; LDSB = dummy LDS
// bytes per block (caps blocks/CU); each wave streams rows wy, wy+4, ... of a 64 x TH tile.
template <int DEPTH, int LDSB, int TH>
__global__ __launch_bounds__(256) void stream1(const float *__restrict__ w, const float *__restrict__ oy,
                                                const float *__restrict__ ox, float *__restrict__ out,
                                                int H, int W)
{
    __shared__ float dummy[LDSB / 4];
    if (LDSB > 4 && threadIdx.x == 999) dummy[threadIdx.y] = 0;  // keep the allocation
    const int tx0 = blockIdx.x * 64, ty0 = blockIdx.y * TH;
    const int wy = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const unsigned x = tx0 + threadIdx.x;
    if ((int)x >= W) return;
    const size_t plane = (size_t)H * W;
    constexpr int ROWS = TH / 4;       // pixel rows per wave
    constexpr int STEPS = ROWS * 5;    // tap rows per wave
    float buf[DEPTH + 1][15];
    float acc0 = 0, acc1 = 0, acc2 = 0;
    auto issue = [&](int step, float *dst) {
        const int prow = step / 5, r = step % 5;
        const int y = ty0 + wy + 4 * prow;
        const size_t ro = (size_t)min(y, H - 1) * W;
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const size_t o = ro + (size_t)(r * 5 + s) * plane;
            dst[3 * s] = (w + o)[x];
            dst[3 * s + 1] = (oy + o)[x];
            dst[3 * s + 2] = (ox + o)[x];
        }
    };
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) issue(i, buf[i]);
#pragma unroll
    for (int step = 0; step < STEPS; ++step) {
        if (step + DEPTH < STEPS) issue(step + DEPTH, buf[(step + DEPTH) % (DEPTH + 1)]);
        __builtin_amdgcn_sched_barrier(0);
        float *c = buf[step % (DEPTH + 1)];
#pragma unroll
        for (int s = 0; s < 5; ++s) { acc0 += c[3 * s]; acc1 += c[3 * s + 1]; acc2 += c[3 * s + 2]; }
        asm volatile("" : "+v"(acc0), "+v"(acc1), "+v"(acc2));
        __builtin_amdgcn_sched_barrier(0);
        if (step % 5 == 4) {
            const int y = ty0 + wy + 4 * (step / 5);
            if (y < H) {
                float *orow = out + (size_t)y * W;
                orow[x] = acc0; (orow + plane)[x] = acc1; (orow + 2 * plane)[x] = acc2;
            }
            acc0 = acc1 = acc2 = 0;
        }
    }
}

// dwordx4 variant: a lane owns 4 consecutive x; a wave covers 256 x of one row; block (64,4)
template <int DEPTH, int LDSB, int TH>
__global__ __launch_bounds__(256) void stream4(const float *__restrict__ w, const float *__restrict__ oy,
                                                const float *__restrict__ ox, float *__restrict__ out,
                                                int H, int W)
{
    __shared__ float dummy[LDSB / 4];
    if (LDSB > 4 && threadIdx.x == 999) dummy[threadIdx.y] = 0;
    const int tx0 = blockIdx.x * 256, ty0 = blockIdx.y * TH;
    const int wy = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const unsigned x = tx0 + 4 * threadIdx.x;
    if ((int)x >= W) return;
    const size_t plane = (size_t)H * W;
    constexpr int ROWS = TH / 4;
    constexpr int STEPS = ROWS * 5;
    f32x4 buf[DEPTH + 1][15];
    f32x4 acc0 = 0, acc1 = 0, acc2 = 0;
    auto issue = [&](int step, f32x4 *dst) {
        const int prow = step / 5, r = step % 5;
        const int y = ty0 + wy + 4 * prow;
        const size_t ro = (size_t)min(y, H - 1) * W;
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const size_t o = ro + (size_t)(r * 5 + s) * plane;
            dst[3 * s] = *(const f32x4 *)(w + o + x);
            dst[3 * s + 1] = *(const f32x4 *)(oy + o + x);
            dst[3 * s + 2] = *(const f32x4 *)(ox + o + x);
        }
    };
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) issue(i, buf[i]);
#pragma unroll
    for (int step = 0; step < STEPS; ++step) {
        if (step + DEPTH < STEPS) issue(step + DEPTH, buf[(step + DEPTH) % (DEPTH + 1)]);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 *c = buf[step % (DEPTH + 1)];
#pragma unroll
        for (int s = 0; s < 5; ++s) { acc0 += c[3 * s]; acc1 += c[3 * s + 1]; acc2 += c[3 * s + 2]; }
        asm volatile("" : "+v"(acc0), "+v"(acc1), "+v"(acc2));
        __builtin_amdgcn_sched_barrier(0);
        if (step % 5 == 4) {
            const int y = ty0 + wy + 4 * (step / 5);
            if (y < H) {
                float *orow = out + (size_t)y * W + x;
                *(f32x4 *)orow = acc0; *(f32x4 *)(orow + plane) = acc1; *(f32x4 *)(orow + 2 * plane) = acc2;
            }
            acc0 = acc1 = acc2 = 0;
        }
    }
}


// persistent: grid of G blocks; block b walks tiles b, b+G, ... of 64 x 4 (one row per wave),
// so that at any moment the resident blocks read a contiguous band of rows of each plane.
template <int DEPTH, int LDSB>
__global__ __launch_bounds__(256) void stream_persist(const float *__restrict__ w, const float *__restrict__ oy,
                                                       const float *__restrict__ ox, float *__restrict__ out,
                                                       int H, int W)
{
    __shared__ float dummy[LDSB / 4];
    if (LDSB > 4 && threadIdx.x == 999) dummy[threadIdx.y] = 0;
    const int wy = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int tiles_x = (W + 63) / 64, tiles_y = (H + 3) / 4;
    const int ntiles = tiles_x * tiles_y;
    const size_t plane = (size_t)H * W;
    float buf[2][15];
    auto issue = [&](int tile, int r, float *dst) {
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const int y = min(ty * 4 + wy, H - 1);
        const unsigned x = min(tx * 64 + (int)threadIdx.x, W - 1);
        const size_t ro = (size_t)y * W;
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const size_t o = ro + (size_t)(r * 5 + s) * plane;
            dst[3 * s] = (w + o)[x];
            dst[3 * s + 1] = (oy + o)[x];
            dst[3 * s + 2] = (ox + o)[x];
        }
    };
    int tile = blockIdx.x;
    if (tile >= ntiles) return;
    issue(tile, 0, buf[0]);
#pragma unroll 1
    for (; tile < ntiles; tile += gridDim.x) {
        float acc0 = 0, acc1 = 0, acc2 = 0;
#pragma unroll
        for (int r = 0; r < 5; ++r) {
            if (r < 4) issue(tile, r + 1, buf[(r + 1) & 1]);
            else if (tile + (int)gridDim.x < ntiles) issue(tile + gridDim.x, 0, buf[(r + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            float *c = buf[r & 1];
#pragma unroll
            for (int s = 0; s < 5; ++s) { acc0 += c[3 * s]; acc1 += c[3 * s + 1]; acc2 += c[3 * s + 2]; }
            asm volatile("" : "+v"(acc0), "+v"(acc1), "+v"(acc2));
            __builtin_amdgcn_sched_barrier(0);
        }
        // r=4 wrote buf[1]; next tile's r=0 expects buf[0]
#pragma unroll
        for (int i = 0; i < 15; ++i) buf[0][i] = buf[1][i];
        const int ty = tile / tiles_x, tx = tile - ty * tiles_x;
        const int y = ty * 4 + wy, x = tx * 64 + threadIdx.x;
        if (y < H && x < W) {
            float *orow = out + (size_t)y * W;
            orow[x] = acc0; (orow + plane)[x] = acc1; (orow + 2 * plane)[x] = acc2;
        }
    }
}

// backward-like stream: read the 75 planes (+3 gout planes), write 75 planes, trivial math.
template <int LDSB, int TH, int WAVES, bool NT>
__global__ __launch_bounds__(WAVES * 64) void streamb(const float *__restrict__ w, const float *__restrict__ oy,
                                                const float *__restrict__ ox, float *__restrict__ gw,
                                                float *__restrict__ ga, float *__restrict__ gb,
                                                int H, int W)
{
    __shared__ float dummy[LDSB / 4];
    if (LDSB > 4 && threadIdx.x == 999) dummy[threadIdx.y] = 0;
    const int tx0 = blockIdx.x * 64, ty0 = blockIdx.y * TH;
    const int wy = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const unsigned x = tx0 + threadIdx.x;
    if ((int)x >= W) return;
    const size_t plane = (size_t)H * W;
    constexpr int ROWS = TH / WAVES;
    constexpr int STEPS = ROWS * 5;
    float buf[2][15];
    auto issue = [&](int step, float *dst) {
        const int prow = step / 5, r = step % 5;
        const int y = ty0 + wy + WAVES * prow;
        const size_t ro = (size_t)min(y, H - 1) * W;
#pragma unroll
        for (int s = 0; s < 5; ++s) {
            const size_t o = ro + (size_t)(r * 5 + s) * plane;
            dst[3 * s] = NT ? __builtin_nontemporal_load(w + o + x) : (w + o)[x];
            dst[3 * s + 1] = NT ? __builtin_nontemporal_load(oy + o + x) : (oy + o)[x];
            dst[3 * s + 2] = NT ? __builtin_nontemporal_load(ox + o + x) : (ox + o)[x];
        }
    };
    issue(0, buf[0]);
#pragma unroll
    for (int step = 0; step < STEPS; ++step) {
        if (step + 1 < STEPS) issue(step + 1, buf[(step + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        float *c = buf[step & 1];
        const int prow = step / 5, r = step % 5;
        const int y = ty0 + wy + WAVES * prow;
        if (y < H) {
            const size_t ro = (size_t)y * W;
#pragma unroll
            for (int s = 0; s < 5; ++s) {
                const size_t o = ro + (size_t)(r * 5 + s) * plane;
                const float v0 = c[3 * s] * 1.5f, v1 = c[3 * s + 1] + c[3 * s], v2 = c[3 * s + 2] - c[3 * s];
                if (NT) {
                    __builtin_nontemporal_store(v0, gw + o + x);
                    __builtin_nontemporal_store(v1, ga + o + x);
                    __builtin_nontemporal_store(v2, gb + o + x);
                } else {
                    (gw + o)[x] = v0; (ga + o)[x] = v1; (gb + o)[x] = v2;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <int LDSB, int TH, int WAVES, bool NT>
static void lb(const float *w, const float *oy, const float *ox, float *gw, float *ga, float *gb, int H, int W, hipStream_t st)
{
    hipLaunchKernelGGL((streamb<LDSB, TH, WAVES, NT>), dim3((W + 63) / 64, (H + TH - 1) / TH), dim3(64, WAVES), 0, st, w, oy, ox, gw, ga, gb, H, W);
}

// backward-like stream with the tap rows split across grid.z: a block streams only tap row
// blockIdx.z (15 planes in, 15 planes out) for its 64 x TH pixels -> 30 concurrently active
// plane streams chip-wide instead of 150 (blocks dispatch z-major last).
template <int LDSB, int TH, int WAVES, bool NT, int TAPS, int ORDER = 0>
__global__ __launch_bounds__(WAVES * 64) void streamb_split(const float *__restrict__ w, const float *__restrict__ oy,
                                                const float *__restrict__ ox, float *__restrict__ gw,
                                                float *__restrict__ ga, float *__restrict__ gb,
                                                int H, int W)
{
    __shared__ float dummy[LDSB / 4];
    if (LDSB > 4 && threadIdx.x == 999) dummy[threadIdx.y] = 0;
    const int tx0 = (ORDER ? blockIdx.y : blockIdx.x) * 64, ty0 = (ORDER ? blockIdx.z : blockIdx.y) * TH;
    const int t0 = (ORDER ? blockIdx.x : blockIdx.z) * TAPS;
    const int wy = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const unsigned x = tx0 + threadIdx.x;
    if ((int)x >= W) return;
    const size_t plane = (size_t)H * W;
    constexpr int ROWS = TH / WAVES;
    float buf[2][3 * TAPS];
    auto issue = [&](int prow, float *dst) {
        const int y = ty0 + wy + WAVES * prow;
        const size_t ro = (size_t)min(y, H - 1) * W;
#pragma unroll
        for (int s = 0; s < TAPS; ++s) {
            const size_t o = ro + (size_t)(t0 + s) * plane;
            dst[3 * s] = NT ? __builtin_nontemporal_load(w + o + x) : (w + o)[x];
            dst[3 * s + 1] = NT ? __builtin_nontemporal_load(oy + o + x) : (oy + o)[x];
            dst[3 * s + 2] = NT ? __builtin_nontemporal_load(ox + o + x) : (ox + o)[x];
        }
    };
    issue(0, buf[0]);
#pragma unroll
    for (int prow = 0; prow < ROWS; ++prow) {
        if (prow + 1 < ROWS) issue(prow + 1, buf[(prow + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        float *c = buf[prow & 1];
        const int y = ty0 + wy + WAVES * prow;
        if (y < H) {
            const size_t ro = (size_t)y * W;
#pragma unroll
            for (int s = 0; s < TAPS; ++s) {
                const size_t o = ro + (size_t)(t0 + s) * plane;
                const float v0 = c[3 * s] * 1.5f, v1 = c[3 * s + 1] + c[3 * s], v2 = c[3 * s + 2] - c[3 * s];
                if (NT) {
                    __builtin_nontemporal_store(v0, gw + o + x);
                    __builtin_nontemporal_store(v1, ga + o + x);
                    __builtin_nontemporal_store(v2, gb + o + x);
                } else {
                    (gw + o)[x] = v0; (ga + o)[x] = v1; (gb + o)[x] = v2;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}
template <int LDSB, int TH, int WAVES, bool NT, int TAPS>
static void lbs1(const float *w, const float *oy, const float *ox, float *gw, float *ga, float *gb, int H, int W, hipStream_t st)
{
    hipLaunchKernelGGL((streamb_split<LDSB, TH, WAVES, NT, TAPS, 1>), dim3(25 / TAPS, (W + 63) / 64, (H + TH - 1) / TH), dim3(64, WAVES), 0, st, w, oy, ox, gw, ga, gb, H, W);
}
template <int LDSB, int TH, int WAVES, bool NT, int TAPS>
static void lbs(const float *w, const float *oy, const float *ox, float *gw, float *ga, float *gb, int H, int W, hipStream_t st)
{
    hipLaunchKernelGGL((streamb_split<LDSB, TH, WAVES, NT, TAPS>), dim3((W + 63) / 64, (H + TH - 1) / TH, 25 / TAPS), dim3(64, WAVES), 0, st, w, oy, ox, gw, ga, gb, H, W);
}

// tap-major order inside a block: for tap t: for the wave's ROWS rows: stream 3 planes (+ write 3
// planes when BWD).  All blocks of a residency round start together, so chip-wide only a few
// planes are active at any moment.
template <int LDSB, int TH, bool BWD, int NT>
__global__ __launch_bounds__(256) void stream_tapmajor(const float *__restrict__ w, const float *__restrict__ oy,
                                                const float *__restrict__ ox, float *__restrict__ gw,
                                                float *__restrict__ ga, float *__restrict__ gb,
                                                int H, int W)
{
    __shared__ float dummy[LDSB / 4];
    if (LDSB > 4 && threadIdx.x == 999) dummy[threadIdx.y] = 0;
    const int tx0 = blockIdx.x * 64, ty0 = blockIdx.y * TH;
    const int wy = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const unsigned x = tx0 + threadIdx.x;
    if ((int)x >= W) return;
    const size_t plane = (size_t)H * W;
    constexpr int ROWS = TH / 4;
    float buf[2][3 * ROWS];
    float acc[3 * ROWS];
#pragma unroll
    for (int i = 0; i < 3 * ROWS; ++i) acc[i] = 0;
    auto issue = [&](int t, float *dst) {
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            const int y = min(ty0 + wy + 4 * j, H - 1);
            const size_t o = (size_t)y * W + (size_t)t * plane;
            dst[3 * j] = (NT & 1) ? __builtin_nontemporal_load(w + o + x) : (w + o)[x];
            dst[3 * j + 1] = (NT & 1) ? __builtin_nontemporal_load(oy + o + x) : (oy + o)[x];
            dst[3 * j + 2] = (NT & 1) ? __builtin_nontemporal_load(ox + o + x) : (ox + o)[x];
        }
    };
    issue(0, buf[0]);
#pragma unroll
    for (int t = 0; t < 25; ++t) {
        if (t + 1 < 25) issue(t + 1, buf[(t + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        float *c = buf[t & 1];
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            const int y = ty0 + wy + 4 * j;
            if (BWD) {
                if (y < H) {
                    const size_t o = (size_t)y * W + (size_t)t * plane;
                    const float v0 = c[3 * j] * 1.5f, v1 = c[3 * j + 1] + c[3 * j], v2 = c[3 * j + 2] - c[3 * j];
                    if (NT & 2) {
                        __builtin_nontemporal_store(v0, gw + o + x);
                        __builtin_nontemporal_store(v1, ga + o + x);
                        __builtin_nontemporal_store(v2, gb + o + x);
                    } else { (gw + o)[x] = v0; (ga + o)[x] = v1; (gb + o)[x] = v2; }
                }
            } else {
                acc[3 * j] += c[3 * j]; acc[3 * j + 1] += c[3 * j + 1]; acc[3 * j + 2] += c[3 * j + 2];
            }
        }
        if (!BWD) {
#pragma unroll
            for (int i = 0; i < 3 * ROWS; ++i) asm volatile("" : "+v"(acc[i]));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (!BWD) {
#pragma unroll
        for (int j = 0; j < ROWS; ++j) {
            const int y = ty0 + wy + 4 * j;
            if (y < H) {
                float *orow = gw + (size_t)y * W;
                orow[x] = acc[3 * j]; (orow + plane)[x] = acc[3 * j + 1]; (orow + 2 * plane)[x] = acc[3 * j + 2];
            }
        }
    }
}
template <int LDSB, int TH, bool BWD, int NT>
static void ltm(const float *w, const float *oy, const float *ox, float *gw, float *ga, float *gb, int H, int W, hipStream_t st)
{
    hipLaunchKernelGGL((stream_tapmajor<LDSB, TH, BWD, NT>), dim3((W + 63) / 64, (H + TH - 1) / TH), dim3(64, 4), 0, st, w, oy, ox, gw, ga, gb, H, W);
}

// tap-major with a lane owning VEC consecutive pixels: a wave covers 64*VEC pixels of RW rows
// (dwordx2/x4 accesses: 512 B / 1 KB contiguous per wave instruction instead of 256 B), same
// pixels per wave (64 * VEC * RW) as the scalar layout with ROWS = VEC * RW.
template <int LDSB, int VEC, int RW, bool BWD, int NT>
__global__ __launch_bounds__(256) void stream_tapmajor_v(const float *__restrict__ w, const float *__restrict__ oy,
                                                  const float *__restrict__ ox, float *__restrict__ gw,
                                                  float *__restrict__ ga, float *__restrict__ gb,
                                                  int H, int W)
{
    typedef float vecf __attribute__((ext_vector_type(VEC)));
    __shared__ float dummy[LDSB / 4];
    if (LDSB > 4 && threadIdx.x == 999) dummy[threadIdx.y] = 0;
    constexpr int TH = 4 * RW;
    int tx0, ty0;
    if (NT & 4) {  // XCD-aware banded tile order (1-D grid padded to a multiple of 8)
        const unsigned tiles_x = (unsigned)(W + 64 * VEC - 1) / (64 * VEC), tiles_y = (unsigned)(H + TH - 1) / TH;
        const unsigned nb = tiles_x * tiles_y, per = (nb + 7) / 8;
        const unsigned t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
        if (t >= nb) return;
        const unsigned ty = t / tiles_x;
        tx0 = (int)(t - ty * tiles_x) * 64 * VEC;
        ty0 = (int)ty * TH;
    } else {
        tx0 = blockIdx.x * 64 * VEC;
        ty0 = blockIdx.y * TH;
    }
    const int wy = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const unsigned x = tx0 + threadIdx.x * VEC;
    if ((int)x >= W) return;  // W % VEC == 0 assumed
    const size_t plane = (size_t)H * W;
    vecf buf[2][3 * RW];
    vecf acc[3 * RW];
#pragma unroll
    for (int i = 0; i < 3 * RW; ++i) acc[i] = 0;
    auto issue = [&](int t, vecf *dst) {
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            const int y = min(ty0 + wy + 4 * j, H - 1);
            const size_t o = (size_t)y * W + (size_t)t * plane;
            if (NT & 8) {  // write-only experiment: no loads
                dst[3 * j] = (float)(o & 1023);
                dst[3 * j + 1] = (float)t;
                dst[3 * j + 2] = (float)x;
            } else {
                dst[3 * j] = __builtin_nontemporal_load((const vecf *)(w + o + x));
                dst[3 * j + 1] = __builtin_nontemporal_load((const vecf *)(oy + o + x));
                dst[3 * j + 2] = __builtin_nontemporal_load((const vecf *)(ox + o + x));
            }
        }
    };
    issue(0, buf[0]);
#pragma unroll
    for (int t = 0; t < 25; ++t) {
        if (t + 1 < 25) issue(t + 1, buf[(t + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        vecf *c = buf[t & 1];
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            const int y = ty0 + wy + 4 * j;
            if (BWD) {
                if (y < H) {
                    const size_t o = (size_t)y * W + (size_t)t * plane;
                    const vecf v0 = c[3 * j] * 1.5f, v1 = c[3 * j + 1] + c[3 * j], v2 = c[3 * j + 2] - c[3 * j];
                    __builtin_nontemporal_store(v0, (vecf *)(gw + o + x));
                    __builtin_nontemporal_store(v1, (vecf *)(ga + o + x));
                    __builtin_nontemporal_store(v2, (vecf *)(gb + o + x));
                }
            } else {
                acc[3 * j] += c[3 * j]; acc[3 * j + 1] += c[3 * j + 1]; acc[3 * j + 2] += c[3 * j + 2];
            }
        }
        if (!BWD) {
#pragma unroll
            for (int i = 0; i < 3 * RW; ++i) {
#pragma unroll
                for (int c = 0; c < VEC; ++c) {
                    float t = acc[i][c];
                    asm volatile("" : "+v"(t));
                    acc[i][c] = t;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (!BWD) {
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            const int y = ty0 + wy + 4 * j;
            if (y < H) {
                float *orow = gw + (size_t)y * W;
                *(vecf *)(orow + x) = acc[3 * j]; *(vecf *)(orow + plane + x) = acc[3 * j + 1]; *(vecf *)(orow + 2 * plane + x) = acc[3 * j + 2];
            }
        }
    }
}
// as stream_tapmajor_v, but only 32 lanes per row (tile stays 64 pixels wide with VEC = 2): a wave
// instruction covers 2 rows x 256 B.  Separates "wider contiguous pieces" from "wider accesses".
// tap-major with a lane owning VEC consecutive pixels: a wave covers 64*VEC pixels of RW rows
// (dwordx2/x4 accesses: 512 B / 1 KB contiguous per wave instruction instead of 256 B), same
// pixels per wave (64 * VEC * RW) as the scalar layout with ROWS = VEC * RW.
template <int LDSB, int VEC, int RW, bool BWD, int NT>
__global__ __launch_bounds__(256) void stream_tapmajor_h(const float *__restrict__ w, const float *__restrict__ oy,
                                                  const float *__restrict__ ox, float *__restrict__ gw,
                                                  float *__restrict__ ga, float *__restrict__ gb,
                                                  int H, int W)
{
    typedef float vecf __attribute__((ext_vector_type(VEC)));
    __shared__ float dummy[LDSB / 4];
    if (LDSB > 4 && threadIdx.x == 999) dummy[threadIdx.y] = 0;
    constexpr int XW = 64 / VEC, SUB = VEC, TH = 4 * RW * SUB;
    int tx0, ty0;
    if (NT & 4) {  // XCD-aware banded tile order (1-D grid padded to a multiple of 8)
        const unsigned tiles_x = (unsigned)(W + 63) / 64, tiles_y = (unsigned)(H + TH - 1) / TH;
        const unsigned nb = tiles_x * tiles_y, per = (nb + 7) / 8;
        const unsigned t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
        if (t >= nb) return;
        const unsigned ty = t / tiles_x;
        tx0 = (int)(t - ty * tiles_x) * 64;
        ty0 = (int)ty * TH;
    } else {
        tx0 = blockIdx.x * 64;
        ty0 = blockIdx.y * TH;
    }
    const int wy = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const unsigned x = tx0 + (threadIdx.x % XW) * VEC;
    const int sub = threadIdx.x / XW;
    if ((int)x >= W) return;  // W % VEC == 0 assumed
    const size_t plane = (size_t)H * W;
    vecf buf[2][3 * RW];
    vecf acc[3 * RW];
#pragma unroll
    for (int i = 0; i < 3 * RW; ++i) acc[i] = 0;
    auto issue = [&](int t, vecf *dst) {
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            const int y = min(ty0 + wy * SUB + sub + 4 * SUB * j, H - 1);
            const size_t o = (size_t)y * W + (size_t)t * plane;
            dst[3 * j] = __builtin_nontemporal_load((const vecf *)(w + o + x));
            dst[3 * j + 1] = __builtin_nontemporal_load((const vecf *)(oy + o + x));
            dst[3 * j + 2] = __builtin_nontemporal_load((const vecf *)(ox + o + x));
        }
    };
    issue(0, buf[0]);
#pragma unroll
    for (int t = 0; t < 25; ++t) {
        if (t + 1 < 25) issue(t + 1, buf[(t + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        vecf *c = buf[t & 1];
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            const int y = ty0 + wy * SUB + sub + 4 * SUB * j;
            if (BWD) {
                if (y < H) {
                    const size_t o = (size_t)y * W + (size_t)t * plane;
                    const vecf v0 = c[3 * j] * 1.5f, v1 = c[3 * j + 1] + c[3 * j], v2 = c[3 * j + 2] - c[3 * j];
                    __builtin_nontemporal_store(v0, (vecf *)(gw + o + x));
                    __builtin_nontemporal_store(v1, (vecf *)(ga + o + x));
                    __builtin_nontemporal_store(v2, (vecf *)(gb + o + x));
                }
            } else {
                acc[3 * j] += c[3 * j]; acc[3 * j + 1] += c[3 * j + 1]; acc[3 * j + 2] += c[3 * j + 2];
            }
        }
        if (!BWD) {
#pragma unroll
            for (int i = 0; i < 3 * RW; ++i) {
#pragma unroll
                for (int c = 0; c < VEC; ++c) {
                    float t = acc[i][c];
                    asm volatile("" : "+v"(t));
                    acc[i][c] = t;
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (!BWD) {
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            const int y = ty0 + wy * SUB + sub + 4 * SUB * j;
            if (y < H) {
                float *orow = gw + (size_t)y * W;
                *(vecf *)(orow + x) = acc[3 * j]; *(vecf *)(orow + plane + x) = acc[3 * j + 1]; *(vecf *)(orow + 2 * plane + x) = acc[3 * j + 2];
            }
        }
    }
}
template <int LDSB, int VEC, int RW, bool BWD, int NT>
static void ltmv(const float *w, const float *oy, const float *ox, float *gw, float *ga, float *gb, int H, int W, hipStream_t st)
{
    const unsigned tiles_x = (W + 64 * VEC - 1) / (64 * VEC), tiles_y = (H + 4 * RW - 1) / (4 * RW);
    const dim3 grid = (NT & 4) ? dim3(((tiles_x * tiles_y + 7) / 8) * 8) : dim3(tiles_x, tiles_y);
    hipLaunchKernelGGL((stream_tapmajor_v<LDSB, VEC, RW, BWD, NT>), grid, dim3(64, 4), 0, st, w, oy, ox, gw, ga, gb, H, W);
}

template <int LDSB, int VEC, int RW, bool BWD, int NT>
static void ltmh(const float *w, const float *oy, const float *ox, float *gw, float *ga, float *gb, int H, int W, hipStream_t st)
{
    const unsigned tiles_x = (W + 63) / 64, tiles_y = (H + 4 * RW * VEC - 1) / (4 * RW * VEC);
    const dim3 grid = (NT & 4) ? dim3(((tiles_x * tiles_y + 7) / 8) * 8) : dim3(tiles_x, tiles_y);
    hipLaunchKernelGGL((stream_tapmajor_h<LDSB, VEC, RW, BWD, NT>), grid, dim3(64, 4), 0, st, w, oy, ox, gw, ga, gb, H, W);
}

// 128-pixel-wide workgroup tiles with one pixel per lane: 8 waves as 2 (x) x 4 (y), dword accesses.
// Separates "wider tile per workgroup" from "wider access per lane" (stream_tapmajor_v).
template <int LDSB, int RW, bool BWD>
__global__ __launch_bounds__(512) void stream_tapmajor_w(const float *__restrict__ w, const float *__restrict__ oy,
                                                  const float *__restrict__ ox, float *__restrict__ gw,
                                                  float *__restrict__ ga, float *__restrict__ gb,
                                                  int H, int W)
{
    __shared__ float dummy[LDSB / 4];
    if (LDSB > 4 && threadIdx.x == 999) dummy[threadIdx.y] = 0;
    constexpr int TH = 4 * RW;
    const unsigned tiles_x = (unsigned)(W + 127) / 128, tiles_y = (unsigned)(H + TH - 1) / TH;
    const unsigned nb = tiles_x * tiles_y, per = (nb + 7) / 8;
    const unsigned t = (blockIdx.x & 7) * per + (blockIdx.x >> 3);
    if (t >= nb) return;
    const unsigned ty = t / tiles_x;
    const int tx0 = (int)(t - ty * tiles_x) * 128, ty0 = (int)ty * TH;
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int wx = wv & 1, wy = wv >> 1;
    const unsigned x = tx0 + wx * 64 + threadIdx.x;
    if ((int)x >= W) return;
    const size_t plane = (size_t)H * W;
    float buf[2][3 * RW];
    float acc[3 * RW];
#pragma unroll
    for (int i = 0; i < 3 * RW; ++i) acc[i] = 0;
    auto issue = [&](int tt, float *dst) {
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            const int y = min(ty0 + wy + 4 * j, H - 1);
            const size_t o = (size_t)y * W + (size_t)tt * plane;
            dst[3 * j] = __builtin_nontemporal_load(w + o + x);
            dst[3 * j + 1] = __builtin_nontemporal_load(oy + o + x);
            dst[3 * j + 2] = __builtin_nontemporal_load(ox + o + x);
        }
    };
    issue(0, buf[0]);
#pragma unroll
    for (int tt = 0; tt < 25; ++tt) {
        if (tt + 1 < 25) issue(tt + 1, buf[(tt + 1) & 1]);
        __builtin_amdgcn_sched_barrier(0);
        float *c = buf[tt & 1];
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            const int y = ty0 + wy + 4 * j;
            if (BWD) {
                if (y < H) {
                    const size_t o = (size_t)y * W + (size_t)tt * plane;
                    __builtin_nontemporal_store(c[3 * j] * 1.5f, gw + o + x);
                    __builtin_nontemporal_store(c[3 * j + 1] + c[3 * j], ga + o + x);
                    __builtin_nontemporal_store(c[3 * j + 2] - c[3 * j], gb + o + x);
                }
            } else {
                acc[3 * j] += c[3 * j]; acc[3 * j + 1] += c[3 * j + 1]; acc[3 * j + 2] += c[3 * j + 2];
            }
        }
        if (!BWD) {
#pragma unroll
            for (int i = 0; i < 3 * RW; ++i) asm volatile("" : "+v"(acc[i]));
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (!BWD) {
#pragma unroll
        for (int j = 0; j < RW; ++j) {
            const int y = ty0 + wy + 4 * j;
            if (y < H) {
                float *orow = gw + (size_t)y * W;
                orow[x] = acc[3 * j]; (orow + plane)[x] = acc[3 * j + 1]; (orow + 2 * plane)[x] = acc[3 * j + 2];
            }
        }
    }
}
template <int LDSB, int RW, bool BWD>
static void ltmw(const float *w, const float *oy, const float *ox, float *gw, float *ga, float *gb, int H, int W, hipStream_t st)
{
    const unsigned tiles_x = (W + 127) / 128, tiles_y = (H + 4 * RW - 1) / (4 * RW);
    hipLaunchKernelGGL((stream_tapmajor_w<LDSB, RW, BWD>), dim3(((tiles_x * tiles_y + 7) / 8) * 8), dim3(64, 8), 0, st, w, oy, ox, gw, ga, gb, H, W);
}

// persistent tap-major with a soft global pace-maker: a wave does not start tap t+1 until every
// wave of its iteration has finished tap t+1-D (bounded spin: purely a performance hint).
__device__ unsigned g_prog[64 * 25];
__global__ void zero_prog() { for (int i = threadIdx.x; i < 64 * 25; i += blockDim.x) g_prog[i] = 0; }

template <int LDSB, int TH, bool BWD, int D>
__global__ __launch_bounds__(256) void stream_tapsync(const float *__restrict__ w, const float *__restrict__ oy,
                                                const float *__restrict__ ox, float *__restrict__ gw,
                                                float *__restrict__ ga, float *__restrict__ gb,
                                                int H, int W)
{
    __shared__ float dummy[LDSB / 4];
    if (LDSB > 4 && threadIdx.x == 999) dummy[threadIdx.y] = 0;
    const int wy = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int tiles_x = (W + 63) / 64, tiles_y = (H + TH - 1) / TH, ntiles = tiles_x * tiles_y;
    const size_t plane = (size_t)H * W;
    constexpr int ROWS = TH / 4;
    const int G = gridDim.x;
#pragma unroll 1
    for (int iter = 0, tile = blockIdx.x; tile < ntiles; ++iter, tile += G) {
        const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
        const int tx0 = txi * 64, ty0 = tyi * TH;
        const unsigned x = min(tx0 + (int)threadIdx.x, W - 1);
        const unsigned expected = 4u * (unsigned)min(G, ntiles - iter * G);
        float buf[2][3 * ROWS];
        float acc[3 * ROWS];
#pragma unroll
        for (int i = 0; i < 3 * ROWS; ++i) acc[i] = 0;
        auto issue = [&](int t, float *dst) {
#pragma unroll
            for (int j = 0; j < ROWS; ++j) {
                const int y = min(ty0 + wy + 4 * j, H - 1);
                const size_t o = (size_t)y * W + (size_t)t * plane;
                dst[3 * j] = __builtin_nontemporal_load(w + o + x);
                dst[3 * j + 1] = __builtin_nontemporal_load(oy + o + x);
                dst[3 * j + 2] = __builtin_nontemporal_load(ox + o + x);
            }
        };
        issue(0, buf[0]);
#pragma unroll
        for (int t = 0; t < 25; ++t) {
            if (t + 1 < 25) {
                if (D > 0 && t + 1 - D >= 0) {
                    const unsigned *slot = &g_prog[(iter & 63) * 25 + t + 1 - D];
                    int spins = 0;
                    while (__hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < expected && ++spins < 2000)
                        __builtin_amdgcn_s_sleep(4);
                }
                issue(t + 1, buf[(t + 1) & 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            float *c = buf[t & 1];
#pragma unroll
            for (int j = 0; j < ROWS; ++j) {
                const int y = min(ty0 + wy + 4 * j, H - 1);
                if (BWD) {
                    const size_t o = (size_t)y * W + (size_t)t * plane;
                    const float v0 = c[3 * j] * 1.5f, v1 = c[3 * j + 1] + c[3 * j], v2 = c[3 * j + 2] - c[3 * j];
                    __builtin_nontemporal_store(v0, gw + o + x);
                    __builtin_nontemporal_store(v1, ga + o + x);
                    __builtin_nontemporal_store(v2, gb + o + x);
                } else {
                    acc[3 * j] += c[3 * j]; acc[3 * j + 1] += c[3 * j + 1]; acc[3 * j + 2] += c[3 * j + 2];
                }
            }
            if (!BWD) {
#pragma unroll
                for (int i = 0; i < 3 * ROWS; ++i) asm volatile("" : "+v"(acc[i]));
            }
            __builtin_amdgcn_sched_barrier(0);
            if (D > 0 && (threadIdx.x & 63) == 0) atomicAdd(&g_prog[(iter & 63) * 25 + t], 1u);
        }
        if (!BWD) {
#pragma unroll
            for (int j = 0; j < ROWS; ++j) {
                const int y = min(ty0 + wy + 4 * j, H - 1);
                float *orow = gw + (size_t)y * W;
                orow[x] = acc[3 * j]; (orow + plane)[x] = acc[3 * j + 1]; (orow + 2 * plane)[x] = acc[3 * j + 2];
            }
        }
    }
}
// The same with one pace counter per XCD (workgroup id % 8): atomics and loads of workgroups on one
// XCD meet in that XCD's L2.  D >= 2 (a wave's own tap t is counted only after the wait of tap t).
#ifndef PACE_SCOPE
#define PACE_SCOPE __HIP_MEMORY_SCOPE_AGENT
#endif
__device__ unsigned g_progx[8 * 64 * 25];
__global__ void zero_progx() { for (int i = threadIdx.x; i < 8 * 64 * 25; i += blockDim.x) g_progx[i] = 0; }
template <int LDSB, int TH, int D, int SPINS>
__global__ __launch_bounds__(256) void stream_tapsync_xcd(const float *__restrict__ w, const float *__restrict__ oy,
                                                const float *__restrict__ ox, float *__restrict__ gw,
                                                float *__restrict__ ga, float *__restrict__ gb,
                                                int H, int W)
{
    __shared__ float dummy[LDSB / 4];
    if (LDSB > 4 && threadIdx.x == 999) dummy[threadIdx.y] = 0;
    const int wy = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int tiles_x = (W + 63) / 64, tiles_y = (H + TH - 1) / TH, ntiles = tiles_x * tiles_y;
    const size_t plane = (size_t)H * W;
    constexpr int ROWS = TH / 4;
    const int G = gridDim.x, xcd = blockIdx.x & 7, per = (ntiles + 7) / 8;
    unsigned *prog = g_progx + xcd * 64 * 25;
    // XCD i takes the i-th band of tiles, as the product kernels do
#pragma unroll 1
    for (int iter = 0, j = blockIdx.x >> 3; j < per; ++iter, j += G / 8) {
        const int tile = xcd * per + j;
        if (tile >= ntiles) break;
        const int left = min(per, ntiles - xcd * per) - iter * (G / 8);  // this XCD's workgroups in this iteration
        const unsigned expected = 4u * (unsigned)min(G / 8, left);
        const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
        const int tx0 = txi * 64, ty0 = tyi * TH;
        const unsigned x = min(tx0 + (int)threadIdx.x, W - 1);
        float buf[2][3 * ROWS];
        auto issue = [&](int t, float *dst) {
#pragma unroll
            for (int jj = 0; jj < ROWS; ++jj) {
                const int y = min(ty0 + wy + 4 * jj, H - 1);
                const size_t o = (size_t)y * W + (size_t)t * plane;
                dst[3 * jj] = __builtin_nontemporal_load(w + o + x);
                dst[3 * jj + 1] = __builtin_nontemporal_load(oy + o + x);
                dst[3 * jj + 2] = __builtin_nontemporal_load(ox + o + x);
            }
        };
        issue(0, buf[0]);
#pragma unroll
        for (int t = 0; t < 25; ++t) {
            if (t + 1 < 25) {
                if (D > 0 && t + 1 - D >= 0) {
                    const unsigned *slot = &prog[(iter & 63) * 25 + t + 1 - D];
                    int spins = 0;
                    while (__hip_atomic_load(slot, __ATOMIC_RELAXED, PACE_SCOPE) < expected && ++spins < SPINS)
                        __builtin_amdgcn_s_sleep(2);
                    if (spins >= SPINS && (threadIdx.x & 63) == 0) atomicAdd(&g_progx[8 * 64 * 25 - 1], 1u);  // timeouts
                }
                issue(t + 1, buf[(t + 1) & 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            float *c = buf[t & 1];
#pragma unroll
            for (int jj = 0; jj < ROWS; ++jj) {
                const int y = min(ty0 + wy + 4 * jj, H - 1);
                const size_t o = (size_t)y * W + (size_t)t * plane;
                const float v0 = c[3 * jj] * 1.5f, v1 = c[3 * jj + 1] + c[3 * jj], v2 = c[3 * jj + 2] - c[3 * jj];
                __builtin_nontemporal_store(v0, gw + o + x);
                __builtin_nontemporal_store(v1, ga + o + x);
                __builtin_nontemporal_store(v2, gb + o + x);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (D > 0 && (threadIdx.x & 63) == 0)
                __hip_atomic_fetch_add(&prog[(iter & 63) * 25 + t], 1u, __ATOMIC_RELAXED, PACE_SCOPE);
        }
    }
}
extern "C" unsigned exp_pace_timeouts()
{
    unsigned v = 0;
    hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_progx), 4, (8 * 64 * 25 - 1) * 4);
    return v;
}
template <int LDSB, int TH, int D, int SPINS>
static void ltx(int G, const float *w, const float *oy, const float *ox, float *gw, float *ga, float *gb, int H, int W, hipStream_t st)
{
    hipLaunchKernelGGL(zero_progx, dim3(1), dim3(256), 0, st);
    hipLaunchKernelGGL((stream_tapsync_xcd<LDSB, TH, D, SPINS>), dim3(G), dim3(64, 4), 0, st, w, oy, ox, gw, ga, gb, H, W);
}

// Pacing without communication: every workgroup holds tap t back until the chip-wide constant-rate
// clock (wall_clock64, 100 MHz) reaches t0 + (25 * iteration + t) * ticks, t0 written by a one-thread
// kernel in front.  Workgroups that are ahead wait, those behind never do.
__device__ unsigned long long g_t0;
__global__ void set_t0() { g_t0 = wall_clock64() + 300; }
template <int LDSB, int TH>
__global__ __launch_bounds__(256) void stream_clockpace(const float *__restrict__ w, const float *__restrict__ oy,
                                                const float *__restrict__ ox, float *__restrict__ gw,
                                                float *__restrict__ ga, float *__restrict__ gb,
                                                int H, int W, int ticks)
{
    __shared__ float dummy[LDSB / 4];
    if (LDSB > 4 && threadIdx.x == 999) dummy[threadIdx.y] = 0;
    const int wy = __builtin_amdgcn_readfirstlane(threadIdx.y);
    const int tiles_x = (W + 63) / 64, tiles_y = (H + TH - 1) / TH, ntiles = tiles_x * tiles_y;
    const size_t plane = (size_t)H * W;
    constexpr int ROWS = TH / 4;
    const int G = gridDim.x, xcd = blockIdx.x & 7, per = (ntiles + 7) / 8;
    const unsigned long long t0 = __hip_atomic_load(&g_t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll 1
    for (int iter = 0, j = blockIdx.x >> 3; j < per; ++iter, j += G / 8) {
        const int tile = xcd * per + j;
        if (tile >= ntiles) break;
        const int tyi = tile / tiles_x, txi = tile - tyi * tiles_x;
        const int tx0 = txi * 64, ty0 = tyi * TH;
        const unsigned x = min(tx0 + (int)threadIdx.x, W - 1);
        float buf[2][3 * ROWS];
        auto issue = [&](int t, float *dst) {
#pragma unroll
            for (int jj = 0; jj < ROWS; ++jj) {
                const int y = min(ty0 + wy + 4 * jj, H - 1);
                const size_t o = (size_t)y * W + (size_t)t * plane;
                dst[3 * jj] = __builtin_nontemporal_load(w + o + x);
                dst[3 * jj + 1] = __builtin_nontemporal_load(oy + o + x);
                dst[3 * jj + 2] = __builtin_nontemporal_load(ox + o + x);
            }
        };
        issue(0, buf[0]);
#pragma unroll
        for (int t = 0; t < 25; ++t) {
            if (t + 1 < 25) {
                if (ticks > 0) {
                    const unsigned long long due = t0 + (unsigned long long)((25 * iter + t + 1) * ticks);
                    int spins = 0;
                    while (wall_clock64() < due && ++spins < 4000) __builtin_amdgcn_s_sleep(1);
                }
                issue(t + 1, buf[(t + 1) & 1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            float *c = buf[t & 1];
#pragma unroll
            for (int jj = 0; jj < ROWS; ++jj) {
                const int y = min(ty0 + wy + 4 * jj, H - 1);
                const size_t o = (size_t)y * W + (size_t)t * plane;
                const float v0 = c[3 * jj] * 1.5f, v1 = c[3 * jj + 1] + c[3 * jj], v2 = c[3 * jj + 2] - c[3 * jj];
                __builtin_nontemporal_store(v0, gw + o + x);
                __builtin_nontemporal_store(v1, ga + o + x);
                __builtin_nontemporal_store(v2, gb + o + x);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}
template <int LDSB, int TH>
static void lcp(int G, int ticks, const float *w, const float *oy, const float *ox, float *gw, float *ga, float *gb, int H, int W, hipStream_t st)
{
    hipLaunchKernelGGL(set_t0, dim3(1), dim3(1), 0, st);
    hipLaunchKernelGGL((stream_clockpace<LDSB, TH>), dim3(G), dim3(64, 4), 0, st, w, oy, ox, gw, ga, gb, H, W, ticks);
}

template <int LDSB, int TH, bool BWD, int D>
static void lts(int G, const float *w, const float *oy, const float *ox, float *gw, float *ga, float *gb, int H, int W, hipStream_t st)
{
    hipLaunchKernelGGL(zero_prog, dim3(1), dim3(256), 0, st);
    hipLaunchKernelGGL((stream_tapsync<LDSB, TH, BWD, D>), dim3(G), dim3(64, 4), 0, st, w, oy, ox, gw, ga, gb, H, W);
}

template <int DEPTH, int LDSB, int TH>
static void l1(const float *w, const float *oy, const float *ox, float *out, int H, int W, hipStream_t st)
{
    hipLaunchKernelGGL((stream1<DEPTH, LDSB, TH>), dim3((W + 63) / 64, (H + TH - 1) / TH), dim3(64, 4), 0, st, w, oy, ox, out, H, W);
}
template <int DEPTH, int LDSB, int TH>
static void l4(const float *w, const float *oy, const float *ox, float *out, int H, int W, hipStream_t st)
{
    hipLaunchKernelGGL((stream4<DEPTH, LDSB, TH>), dim3((W + 255) / 256, (H + TH - 1) / TH), dim3(64, 4), 0, st, w, oy, ox, out, H, W);
}

extern "C" int exp_runb(int variant, const float *w, const float *oy, const float *ox, float *gw, float *ga, float *gb, int H, int W, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    constexpr int L4 = 40960, L8 = 4;
    switch (variant) {
    case 0: lb<L4, 16, 4, false>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 1: lb<L4, 16, 4, true>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 2: lb<L8, 16, 4, true>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 3: lb<L4, 8, 4, true>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 4: lb<L8, 8, 4, true>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 5: lb<L4, 4, 4, true>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 6: lb<L8, 4, 4, true>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 7: lb<L8, 4, 4, false>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 8: lb<L8, 2, 2, true>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 9: lb<L8, 1, 1, true>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 10: lb<L4, 32, 4, true>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 11: lbs<L4, 16, 4, false, 5>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 12: lbs<L4, 16, 4, true, 5>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 13: lbs<L8, 16, 4, true, 5>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 14: lbs<L4, 32, 4, true, 5>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 15: lbs<L4, 64, 4, true, 5>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 16: lbs<L4, 16, 4, true, 1>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 17: lbs<L4, 64, 4, true, 1>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 18: lbs<L8, 64, 4, false, 1>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 19: lbs1<L4, 16, 4, true, 1>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 20: lbs1<L4, 16, 4, true, 5>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 21: lbs<L4, 16, 4, false, 1>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 22: lbs<L8, 16, 4, true, 1>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 23: lbs<L4, 8, 4, true, 1>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 24: lbs<L4, 32, 4, true, 1>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 30: ltm<L4, 16, true, 3>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 31: ltm<L4, 16, true, 0>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 32: ltm<L4, 32, true, 3>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 33: ltm<L4, 8, true, 3>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 34: ltm<L8, 16, true, 3>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 35: ltm<L4, 16, false, 3>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 36: ltm<L4, 16, false, 0>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 37: ltm<L4, 32, false, 3>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 38: ltm<L4, 8, false, 3>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 39: ltm<L8, 16, false, 3>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 50: lts<70000, 32, true, 0>(512, w, oy, ox, gw, ga, gb, H, W, st); break;
    case 51: lts<70000, 32, true, 1>(512, w, oy, ox, gw, ga, gb, H, W, st); break;
    case 52: lts<70000, 32, true, 2>(512, w, oy, ox, gw, ga, gb, H, W, st); break;
    case 53: lts<70000, 32, true, 4>(512, w, oy, ox, gw, ga, gb, H, W, st); break;
    case 54: lts<70000, 32, false, 0>(512, w, oy, ox, gw, ga, gb, H, W, st); break;
    case 55: lts<70000, 32, false, 1>(512, w, oy, ox, gw, ga, gb, H, W, st); break;
    case 56: lts<70000, 32, false, 2>(512, w, oy, ox, gw, ga, gb, H, W, st); break;
    case 57: lts<36000, 32, true, 2>(1024, w, oy, ox, gw, ga, gb, H, W, st); break;
    case 58: lts<36000, 32, true, 0>(1024, w, oy, ox, gw, ga, gb, H, W, st); break;
    case 300: lcp<140000, 32>(256, 0, w, oy, ox, gw, ga, gb, H, W, st); break;
    case 301: lcp<140000, 32>(256, 180, w, oy, ox, gw, ga, gb, H, W, st); break;
    case 302: lcp<140000, 32>(256, 200, w, oy, ox, gw, ga, gb, H, W, st); break;
    case 303: lcp<140000, 32>(256, 220, w, oy, ox, gw, ga, gb, H, W, st); break;
    case 304: lcp<140000, 32>(256, 240, w, oy, ox, gw, ga, gb, H, W, st); break;
    case 305: lcp<140000, 32>(256, 160, w, oy, ox, gw, ga, gb, H, W, st); break;
    case 310: lcp<70000, 32>(512, 0, w, oy, ox, gw, ga, gb, H, W, st); break;
    case 311: lcp<70000, 32>(512, 360, w, oy, ox, gw, ga, gb, H, W, st); break;
    case 312: lcp<70000, 32>(512, 400, w, oy, ox, gw, ga, gb, H, W, st); break;
    case 313: lcp<70000, 32>(512, 440, w, oy, ox, gw, ga, gb, H, W, st); break;
    case 314: lcp<70000, 32>(512, 480, w, oy, ox, gw, ga, gb, H, W, st); break;
    case 200: ltx<70000, 32, 0, 1>(512, w, oy, ox, gw, ga, gb, H, W, st); break;
    case 201: ltx<70000, 32, 2, 400>(512, w, oy, ox, gw, ga, gb, H, W, st); break;
    case 202: ltx<70000, 32, 3, 400>(512, w, oy, ox, gw, ga, gb, H, W, st); break;
    case 203: ltx<70000, 32, 4, 400>(512, w, oy, ox, gw, ga, gb, H, W, st); break;
    case 204: ltx<140000, 32, 0, 1>(256, w, oy, ox, gw, ga, gb, H, W, st); break;
    case 205: ltx<140000, 32, 2, 400>(256, w, oy, ox, gw, ga, gb, H, W, st); break;
    case 206: ltx<140000, 32, 3, 400>(256, w, oy, ox, gw, ga, gb, H, W, st); break;
    case 207: ltx<140000, 16, 2, 400>(256, w, oy, ox, gw, ga, gb, H, W, st); break;
    case 208: ltx<140000, 16, 0, 1>(256, w, oy, ox, gw, ga, gb, H, W, st); break;
    case 107: ltmw<88000, 4, false>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 108: ltmw<140000, 4, false>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 109: ltmw<140000, 4, true>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 110: ltmv<88000, 1, 8, false, 7>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 100: ltmv<54000, 1, 4, true, 7>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 101: ltmv<88000, 1, 4, true, 7>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 102: ltmv<88000, 1, 8, true, 7>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 103: ltmv<40000, 1, 4, true, 7>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 104: ltmv<40000, 1, 2, true, 7>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 105: ltmw<88000, 2, true>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 106: ltmw<54000, 2, true>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 95: ltmw<126000, 8, true>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 96: ltmw<88000, 4, true>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 97: ltmw<126000, 8, false>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 98: ltmw<72000, 4, true>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 90: ltmv<54000, 1, 8, true, 11>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 91: ltmv<54000, 2, 4, true, 11>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 92: ltmv<54000, 4, 2, true, 11>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 80: ltmh<54000, 2, 4, true, 3>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 81: ltmh<54000, 2, 4, true, 7>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 82: ltmh<54000, 4, 2, true, 3>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 83: ltmh<54000, 2, 4, false, 3>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 70: ltmv<54000, 1, 8, true, 7>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 71: ltmv<54000, 2, 4, true, 7>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 72: ltmv<54000, 4, 2, true, 7>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 73: ltmv<54000, 1, 8, false, 7>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 74: ltmv<54000, 2, 4, false, 7>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 75: ltmv<54000, 2, 8, true, 7>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 76: ltmv<54000, 2, 2, true, 7>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 60: ltmv<54000, 2, 4, true, 3>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 61: ltmv<54000, 4, 2, true, 3>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 62: ltmv<54000, 2, 4, false, 3>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 63: ltmv<54000, 4, 2, false, 3>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 64: ltmv<81920, 4, 2, true, 3>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 65: ltmv<54000, 4, 4, true, 3>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 66: ltmv<54000, 4, 1, true, 3>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 67: ltmv<54000, 1, 8, true, 3>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 40: ltm<L4, 32, true, 0>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 41: ltm<L4, 32, true, 1>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 42: ltm<L4, 32, true, 2>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 43: ltm<81920, 32, true, 3>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 44: ltm<81920, 32, false, 3>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 45: ltm<L4, 32, false, 0>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 46: ltm<54000, 32, false, 3>(w, oy, ox, gw, ga, gb, H, W, st); break;
    case 47: ltm<54000, 32, true, 3>(w, oy, ox, gw, ga, gb, H, W, st); break;
    default: return -1;
    }
    return (int)hipGetLastError();
}

extern "C" int exp_run(int variant, const float *in, const float *w, const float *oy, const float *ox,
                       float *out, int N, int H, int W, void *stream)
{
    hipStream_t st = (hipStream_t)stream;
    constexpr int L4 = 40960, L5 = 32768, L6 = 27000, L8 = 4;  // blocks/CU 4,5,6,8
    switch (variant) {
    case 0: l1<1, L4, 16>(w, oy, ox, out, H, W, st); break;
    case 1: l1<1, L5, 16>(w, oy, ox, out, H, W, st); break;
    case 2: l1<1, L6, 16>(w, oy, ox, out, H, W, st); break;
    case 3: l1<1, L8, 16>(w, oy, ox, out, H, W, st); break;
    case 4: l1<2, L4, 16>(w, oy, ox, out, H, W, st); break;
    case 5: l1<2, L5, 16>(w, oy, ox, out, H, W, st); break;
    case 6: l1<2, L8, 16>(w, oy, ox, out, H, W, st); break;
    case 7: l1<3, L4, 16>(w, oy, ox, out, H, W, st); break;
    case 8: l1<3, L8, 16>(w, oy, ox, out, H, W, st); break;
    case 9: l1<4, L4, 16>(w, oy, ox, out, H, W, st); break;
    case 10: l4<1, L4, 16>(w, oy, ox, out, H, W, st); break;
    case 11: l4<1, L8, 16>(w, oy, ox, out, H, W, st); break;
    case 12: l4<2, L4, 16>(w, oy, ox, out, H, W, st); break;
    case 13: l4<1, L4, 8>(w, oy, ox, out, H, W, st); break;
    case 14: l4<1, L8, 8>(w, oy, ox, out, H, W, st); break;
    case 15: l1<2, L4, 32>(w, oy, ox, out, H, W, st); break;
    case 16: l1<2, L4, 8>(w, oy, ox, out, H, W, st); break;
    case 17: l1<1, L4, 4>(w, oy, ox, out, H, W, st); break;
    case 18: l1<1, L8, 4>(w, oy, ox, out, H, W, st); break;
    case 19: hipLaunchKernelGGL((stream_persist<1, L4>), dim3(1024), dim3(64, 4), 0, st, w, oy, ox, out, H, W); break;
    case 20: hipLaunchKernelGGL((stream_persist<1, L8>), dim3(2048), dim3(64, 4), 0, st, w, oy, ox, out, H, W); break;
    case 21: hipLaunchKernelGGL((stream_persist<1, L4>), dim3(768), dim3(64, 4), 0, st, w, oy, ox, out, H, W); break;
    case 22: hipLaunchKernelGGL((stream_persist<1, L4>), dim3(512), dim3(64, 4), 0, st, w, oy, ox, out, H, W); break;
    default: return -1;
    }
    return (int)hipGetLastError();
}
