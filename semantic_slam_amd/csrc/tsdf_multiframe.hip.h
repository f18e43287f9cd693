// tsdf_multiframe.hip.h -- several frames per pass over the volume.
//
// Integrating frame after frame reads and writes every touched weight once per frame.  When the
// frames are known together (offline labelling replays saved keyframes: ref
// examples/label_instance_rgbd.cpp:78-110; bench sequences), a lane can keep its voxels in
// registers and apply F frames to them in order: the volume is read once and written once per F
// frames, the launch and its fill/drain are paid once per F frames, and the pose-independent part
// of the geometry is shared.  Every voxel sees exactly the same sequence of updates as with F
// separate launches, so the result is bit-identical (tests/test_gpu_multiframe.py).
//
// Structure per wavefront (256(x) x R(y) voxels of one slice, as integrate_tile):
//   flags -> for f in frames { geometry(f), depth gather(f), tests(f); first touching frame loads
//   the quads; update in registers } -> store weights / changed TSDF rows / cleared flag.
// The per-frame arithmetic is the FAST + ELIDE + SUM path of integrate_tile (same helpers).
#pragma once
#include "tsdf_kernels.hip.h"

namespace tsdfk {

constexpr int kMaxFramesPerLaunch = 32;
// fine depth tiles (see "fine tiles for brick-sized boxes" below)
constexpr int kFineTile = 4;
constexpr int kFineLevels = 3;                       // per axis: 1, 2, 4 tiles
__host__ __device__ inline size_t fine_table_elems(int fw, int fh) { return (size_t)kFineLevels * kFineLevels * fw * fh; }

struct FramePose {
    const float *depth;
    const uint8_t *mask;     // may be null
    // as IntegrateParams; ordered so that the pairs the packed operations take -- (rx_k, ry_k), (tx, ty) -- sit in
    // aligned SGPR pairs straight out of the scalar loads (no s_mov shuffles per frame)
    float rx0, ry0, rx1, ry1, rx2, ry2, rz0, rz1, rz2, tz, tx, ty;
    int fast_ok;
    float cz_margin;
    // per-voxel label fusion of the same frame (LABELS kernels; both null = no label evidence in this frame)
    const uint16_t *label_im;
    const float *score_im;
    // depth tile summary of this frame (classify_patch; null = none, e.g. a masked frame) and the camera-frame z
    // below which a patch is too close to the camera plane for its projected bounding box to be trusted
    const float2 *tiles;
    float cz_short;   // host: max(B / 64, error bound / 3.2e-5), B = bound on the camera-frame coordinates of the slab
    float cz_pad;     // host: >= twice the bound on |exact-path cz - affine cz| (and on the corner arithmetic's error)
};

// ---- depth tile summaries ---------------------------------------------------------------------------------
// Per tile of a depth frame (16 x 16 pixels, or 8 x 8: the host chooses per slab, see tile_edge_for): x = the smallest depth
// if EVERY pixel of the tile passes the reference's depth-range test (0 < d <= max_depth, ref: src/tsdf.cu:46), else -inf;
// y = the largest depth among the pixels that pass it, -inf if none does.  A NaN anywhere in the tile (NaN passes the
// reference's tests and updates the voxel) makes the tile claim nothing: (-inf, +inf).
// Finer tiles bound a box's depths more tightly -- S-surf 512^3: 11.6 % of the wavefront-frames are left to the per-voxel
// path instead of 14.9 % -- but their tables are four times as large (640 x 480: 1.6 MB per frame instead of 0.29 MB).  With
// the brick work list that pays for large slabs (512^3 S-surf 0.0313 -> 0.0281 ms per frame, through noise and dropouts
// 0.0408 -> 0.0354, the 1024^3 trajectory 0.161 -> 0.152) and costs on small ones (200^3: 0.0059 -> 0.0070; round 2, before
// the list: 512^3 0.0431 -> 0.0440, 16 x 200^3 masked 0.145 -> 0.277).
// Counters of a classified fused launch, sharded: one word saturates at about 88 returning device-scope atomics per
// microsecond (MI355X_MICROARCH.md, "dequeue"; measured here: one list head for the 262 144 super-bricks of a 1024^3 slab
// made the pre-pass 1.7 ms instead of 0.13), so the brick work list is kListBuckets sub-lists with a head each, 256 bytes
// apart: bucket b = {unsigned int front_entries, back_entries; unsigned long long claims; ...}.
constexpr int kListBuckets = 64;
constexpr int kBucketStride = 256;                          // bytes between buckets
constexpr int kCounterBytes = kListBuckets * kBucketStride;

// On top of the tiles a 2-D sparse table gives the same two quantities for ANY rectangle of tiles in four loads:
// level (i, j) holds, at (ty, tx), the combination over the 2^i x 2^j tiles starting there (min of the x's, max of
// the y's); a rectangle of w x h tiles is the union of four overlapping power-of-two blocks of level
// (floor(log2 h), floor(log2 w)).  Layout per frame: [level i][level j][ty][tx], levels = floor(log2(dim)) + 1.
struct TileSummaryParams {
    const float *depth[kMaxFramesPerLaunch];
    const uint8_t *mask[kMaxFramesPerLaunch];   // instance mask of the frame or null: the summary is of depth * (mask / 255),
                                                // the image Integrate sees (ref: src/Engine.cpp:192-193)
    float2 *tiles;          // n_frames tables
    int H, W, tiles_w, tiles_h;
    float max_depth;
    // (depth_tile_summary<8> only) level (0, 0) of the launch's fine tables, written in the same pass over the frame: an 8-pixel
    // tile is four fine (4-pixel) tiles and the strip kernel already holds their pixels.  Null = none.  Layout: fine_table_elems.
    float2 *fine = nullptr;
    int fw = 0, fh = 0;
};

__device__ __forceinline__ int tile_levels(int n) { return 32 - __clz(n); }   // floor(log2 n) + 1, n >= 1
__device__ __forceinline__ size_t tile_table_elems(int tw, int th) { return (size_t)tile_levels(tw) * tile_levels(th) * tw * th; }

// block = 64 x 4: one wavefront per tile, four tiles per workgroup; grid = (ceil(tiles / 4), frames)
template <int kTile>
__global__ __launch_bounds__(256) void depth_tile_summary_per_tile(TileSummaryParams tp)
{
    static_assert(kTile == 8 || kTile == 16, "one wavefront per tile: 64 lanes x 1 or 4 pixels");
    const int tile = blockIdx.x * 4 + threadIdx.y, f = blockIdx.y;
    if (tile >= tp.tiles_w * tp.tiles_h) return;
    const int ty = tile / tp.tiles_w, tx = tile - ty * tp.tiles_w;
    const int lane = threadIdx.x;
    constexpr int kPxPerLane = kTile * kTile / 64, kLanesPerRow = kTile / kPxPerLane;
    const int py = ty * kTile + lane / kLanesPerRow, px0 = tx * kTile + (lane % kLanesPerRow) * kPxPerLane;
    const float *d = tp.depth[f];
    const uint8_t *m = tp.mask[f];
    const float inf = __builtin_inff();
    float mn = inf, mx = -inf;
    bool all_valid = true, nan = false;
    if (py < tp.H) {
#pragma unroll
        for (int i = 0; i < kPxPerLane; ++i) {
            const int px = px0 + i;
            if (px < tp.W) {
                float v = d[(size_t)py * tp.W + px];
                if (m != nullptr) v = v * (m[(size_t)py * tp.W + px] >= 128 ? 1.0f : 0.0f);   // inf * 0 = NaN, as in the kernel
                nan |= v != v;
                const bool valid = (v > 0.0f) & (v <= tp.max_depth);
                all_valid &= valid;
                if (valid) { mn = fminf(mn, v); mx = fmaxf(mx, v); }
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, off));
        mx = fmaxf(mx, __shfl_xor(mx, off));
    }
    const bool any_nan = __ballot(nan) != 0ull, every_valid = __ballot(!all_valid) == 0ull;
    if (lane == 0) {
        float2 out;
        out.x = (every_valid && !any_nan) ? mn : -inf;
        out.y = any_nan ? inf : mx;
        tp.tiles[(size_t)f * tile_table_elems(tp.tiles_w, tp.tiles_h) + (size_t)ty * tp.tiles_w + tx] = out;
    }
}

// The same values from whole-row reads: one wavefront per STRIP of 64 pixels x kTile rows -- four 16-pixel tiles or eight
// 8-pixel ones: a load instruction covers four rows of 256 contiguous bytes (16 lanes x 16 B per row) instead of 64-byte (or
// 32-byte) row pieces, a lane accumulates its rows x 4 pixels -- all in one tile -- and the lanes of a tile combine by shuffles;
// min / max are exact and order-free, so the table is the one depth_tile_summary_per_tile builds (tsdf_selftest_tile_tables
// compares them; the claim statistics of S-surf are identical).  The two table kernels were 72 us of a 32-frame launch -- a
// quarter of a 200^3 one.  block = 64 x 4 (four strips per workgroup); grid = (ceil(strips / 4), frames).
template <int kTile>
__global__ __launch_bounds__(256) void depth_tile_summary(TileSummaryParams tp)
{
    static_assert(kTile == 8 || kTile == 16, "64-pixel strips of 16- or 8-pixel tiles");
    constexpr int kTilesPerStrip = 64 / kTile;      // 4 or 8
    constexpr int kLanesPerTileRow = kTile / 4;     // lanes (of 4 pixels) side by side in a tile: 4 or 2
    const int strips_w = (tp.tiles_w + kTilesPerStrip - 1) / kTilesPerStrip;
    const int strip = blockIdx.x * 4 + threadIdx.y, f = blockIdx.y;
    if (strip >= strips_w * tp.tiles_h) return;
    const int ty = strip / strips_w, sx = strip - ty * strips_w;
    const int lane = threadIdx.x, c4 = lane & 15, rr = lane >> 4;
    const int px0 = sx * 64 + c4 * 4;
    const float *d = tp.depth[f];
    const uint8_t *m = tp.mask[f];
    const float inf = __builtin_inff();
    float mn = inf, mx = -inf;
    bool all_valid = true, nan = false;
    // the lane's share of its two fine tiles (kTile == 8: rows rr and rr + 4 of the strip belong to the fine tiles 2 ty and 2 ty + 1)
    float fmn[2] = {inf, inf}, fmx[2] = {-inf, -inf};
    bool fbad[2] = {false, false}, fnan[2] = {false, false};
    const bool vec = (tp.W & 3) == 0 && (reinterpret_cast<uintptr_t>(d) & 15) == 0 &&
                     (m == nullptr || (reinterpret_cast<uintptr_t>(m) & 3) == 0);
#pragma unroll
    for (int i = 0; i < kTile / 4; ++i) {
        const int py = ty * kTile + rr + 4 * i;
        if (py >= tp.H || px0 >= tp.W) continue;
        float v[4];
        bool in[4];
        const size_t at = (size_t)py * tp.W + px0;
        if (vec) {      // W % 4 == 0: the four pixels are inside the row together
            const float4 q = *reinterpret_cast<const float4 *>(d + at);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
            in[0] = in[1] = in[2] = in[3] = true;
            if (m != nullptr) {
                const uchar4 k = *reinterpret_cast<const uchar4 *>(m + at);
                v[0] = v[0] * (k.x >= 128 ? 1.0f : 0.0f); v[1] = v[1] * (k.y >= 128 ? 1.0f : 0.0f);   // inf * 0 = NaN, as in the kernel
                v[2] = v[2] * (k.z >= 128 ? 1.0f : 0.0f); v[3] = v[3] * (k.w >= 128 ? 1.0f : 0.0f);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                in[j] = px0 + j < tp.W;
                v[j] = in[j] ? d[at + j] : 0.0f;
                if (in[j] && m != nullptr) v[j] = v[j] * (m[at + j] >= 128 ? 1.0f : 0.0f);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!in[j]) continue;
            nan |= v[j] != v[j];
            const bool valid = (v[j] > 0.0f) & (v[j] <= tp.max_depth);
            all_valid &= valid;
            if (valid) { mn = fminf(mn, v[j]); mx = fmaxf(mx, v[j]); }
            if constexpr (kTile == 8) {
                fnan[i] |= v[j] != v[j];
                fbad[i] |= !valid;
                if (valid) { fmn[i] = fminf(fmn[i], v[j]); fmx[i] = fmaxf(fmx[i], v[j]); }
            }
        }
    }
    if constexpr (kTile == 8) {
        if (tp.fine != nullptr) {      // wave-uniform
            // a fine tile = the lane's four pixels x the four row groups (lane bits 4, 5); every lane of the wavefront takes part
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                float a = fmn[i], b = fmx[i];
                int bad_i = fbad[i] ? 1 : 0, nan_i = fnan[i] ? 1 : 0;
                a = fminf(a, __shfl_xor(a, 16)); b = fmaxf(b, __shfl_xor(b, 16));
                bad_i |= __shfl_xor(bad_i, 16); nan_i |= __shfl_xor(nan_i, 16);
                a = fminf(a, __shfl_xor(a, 32)); b = fmaxf(b, __shfl_xor(b, 32));
                bad_i |= __shfl_xor(bad_i, 32); nan_i |= __shfl_xor(nan_i, 32);
                const int fx = sx * 16 + c4, fy = ty * 2 + i;
                if (rr == 0 && fx < tp.fw && fy < tp.fh) {
                    float2 out;
                    out.x = (bad_i == 0 && nan_i == 0) ? a : -inf;
                    out.y = nan_i != 0 ? inf : b;
                    tp.fine[(size_t)f * fine_table_elems(tp.fw, tp.fh) + (size_t)fy * tp.fw + fx] = out;
                }
            }
        }
    }
    // the lanes of a tile: kLanesPerTileRow neighbours along the row (lane bits 0 [, 1]) x the four row groups (lane bits 4, 5)
    mn = fminf(mn, __shfl_xor(mn, 1));  mx = fmaxf(mx, __shfl_xor(mx, 1));
    if constexpr (kLanesPerTileRow == 4) { mn = fminf(mn, __shfl_xor(mn, 2));  mx = fmaxf(mx, __shfl_xor(mx, 2)); }
    mn = fminf(mn, __shfl_xor(mn, 16)); mx = fmaxf(mx, __shfl_xor(mx, 16));
    mn = fminf(mn, __shfl_xor(mn, 32)); mx = fmaxf(mx, __shfl_xor(mx, 32));
    const unsigned long long bad = __ballot(!all_valid), nans = __ballot(nan);
    if ((lane & (kLanesPerTileRow - 1)) == 0 && rr == 0) {
        const int k = c4 / kLanesPerTileRow, tx = sx * kTilesPerStrip + k;
        if (tx < tp.tiles_w) {
            const unsigned long long row_lanes = kLanesPerTileRow == 4 ? 0x000F000F000F000Full : 0x0003000300030003ull;
            const unsigned long long tile_lanes = row_lanes << (kLanesPerTileRow * k);
            const bool every_valid = (bad & tile_lanes) == 0ull, any_nan = (nans & tile_lanes) != 0ull;
            float2 out;
            out.x = (every_valid && !any_nan) ? mn : -inf;
            out.y = any_nan ? inf : mx;
            tp.tiles[(size_t)f * tile_table_elems(tp.tiles_w, tp.tiles_h) + (size_t)ty * tp.tiles_w + tx] = out;
        }
    }
}

// The upper levels from level (0, 0), one workgroup per (x level j, frame): first level (0, j) straight from the base
// tiles -- the combination over tiles tx .. min(tx + 2^j - 1, tw - 1) of each row (min and max are exact and
// idempotent, so any evaluation order gives the same bits) -- kept in LDS, then, after ONE barrier, the levels (i, j)
// from 2^i rows of it.  6 x n workgroups with one barrier each instead of one workgroup per frame walking through 29
// dependent passes (55 us per launch: a third of a one-frame launch on a 200^3 volume).  Tables of more than
// kTileLdsEntries tiles fall back to reading level (0, j) from memory after the barrier (same values).
constexpr int kTileLdsEntries = 5120;   // float2: 2 x 40 KiB of LDS (a 640 x 480 frame has 1200 16-pixel or 4800 8-pixel tiles)

__global__ __launch_bounds__(256) void tile_sparse_table_scan(float2 *tables, int tw, int th)
{
    const int lj = tile_levels(tw), li = tile_levels(th), n = tw * th;
    const int j = blockIdx.x;                      // this workgroup's x level
    float2 *T = tables + (size_t)blockIdx.y * tile_table_elems(tw, th);
    float2 *row_level = T + (size_t)j * n;         // level (0, j)
    __shared__ float2 lds_base[kTileLdsEntries], lds_row[kTileLdsEntries];
    const bool in_lds = n <= kTileLdsEntries;
    const float2 *base = T;
    if (in_lds) {                                  // the frame's tiles once into LDS: every later read is an LDS read
        for (int k = threadIdx.x; k < n; k += blockDim.x) lds_base[k] = T[k];
        __syncthreads();
        base = lds_base;
    }
    const int span = 1 << j;
    for (int k = threadIdx.x; k < n; k += blockDim.x) {
        const int ty = k / tw, tx = k - ty * tw;
        const int last = min(tx + span - 1, tw - 1);
        float2 acc = base[k];
        for (int x = tx + 1; x <= last; ++x) {
            const float2 b = base[ty * tw + x];
            acc = make_float2(fminf(acc.x, b.x), fmaxf(acc.y, b.y));
        }
        if (in_lds) lds_row[k] = acc;
        if (j > 0) row_level[k] = acc;             // level (0, 0) is the input itself
    }
    if (!in_lds) __threadfence_block();
    __syncthreads();
    const float2 *row = in_lds ? lds_row : (j > 0 ? row_level : T);
    for (int i = 1; i < li; ++i) {
        float2 *dst = T + (size_t)(i * lj + j) * n;
        const int rows = 1 << i;
        for (int k = threadIdx.x; k < n; k += blockDim.x) {
            const int ty = k / tw, tx = k - ty * tw;
            const int last = min(ty + rows - 1, th - 1);
            float2 acc = row[k];
            for (int y = ty + 1; y <= last; ++y) {
                const float2 b = row[y * tw + tx];
                acc = make_float2(fminf(acc.x, b.x), fmaxf(acc.y, b.y));
            }
            dst[k] = acc;
        }
    }
}

// Levels (ky, kx) != (0, 0) of a fine table from its level (0, 0): tile position k reads the up to 4 x 4 base tiles that start
// there (clipped at the table's edge, as the coarse table's levels are) and writes its eight combinations.
__device__ __forceinline__ void fine_levels_at(float2 *__restrict__ T, const int fw, const int fh, const int k)
{
    const int n = fw * fh;
    const int ty = k / fw, tx = k - ty * fw;
    const float inf = __builtin_inff();
    float2 b[4][4];
#pragma unroll
    for (int dy = 0; dy < 4; ++dy)
#pragma unroll
        for (int dx = 0; dx < 4; ++dx)
            b[dy][dx] = (ty + dy < fh && tx + dx < fw) ? T[(size_t)(ty + dy) * fw + tx + dx] : make_float2(inf, -inf);   // neutral
    // rows first: r[dy][kx] = combination of b[dy][0 .. 2^kx - 1]
    float2 r[4][kFineLevels];
#pragma unroll
    for (int dy = 0; dy < 4; ++dy) {
        r[dy][0] = b[dy][0];
        r[dy][1] = make_float2(fminf(b[dy][0].x, b[dy][1].x), fmaxf(b[dy][0].y, b[dy][1].y));
        const float2 hi = make_float2(fminf(b[dy][2].x, b[dy][3].x), fmaxf(b[dy][2].y, b[dy][3].y));
        r[dy][2] = make_float2(fminf(r[dy][1].x, hi.x), fmaxf(r[dy][1].y, hi.y));
    }
#pragma unroll
    for (int kx = 0; kx < kFineLevels; ++kx) {
        const float2 c0 = r[0][kx];
        const float2 c1 = make_float2(fminf(r[0][kx].x, r[1][kx].x), fmaxf(r[0][kx].y, r[1][kx].y));
        const float2 h2 = make_float2(fminf(r[2][kx].x, r[3][kx].x), fmaxf(r[2][kx].y, r[3][kx].y));
        const float2 c2 = make_float2(fminf(c1.x, h2.x), fmaxf(c1.y, h2.y));
        if (kx > 0) T[(size_t)(0 * kFineLevels + kx) * n + k] = c0;
        T[(size_t)(1 * kFineLevels + kx) * n + k] = c1;
        T[(size_t)(2 * kFineLevels + kx) * n + k] = c2;
    }
}

// The same table by doubling, for frames whose tiles fit LDS (n <= kTileLdsEntries; 640 x 480: 1200): a level is the
// combination of two entries of the level below -- (0, s + 1) at tx from (0, s) at tx and at min(tx + 2^s, tw - 1), (i + 1, j)
// from (i, j) at ty and at min(ty + 2^i, th - 1); the clamped partner lies inside the clipped range, and min / max are
// idempotent, so every entry is the exact combination over its clipped block, as the scan version computes it.  One
// workgroup per (x level j, frame): j + levels_y - 1 steps of two LDS reads per entry with a barrier each, instead of
// 2^j + 2^i reads per entry.  zero_me (may be null): the launch's counter block (kCounterBytes: the sharded claim counters and
// work-list lengths, see classify_brick_list), cleared here for the kernels that follow on the stream: saves a memset dispatch.
// fine (may be null): the launch's fine tables, level (0, 0) written by depth_tile_summary<8>; the workgroups past the lj level
// workgroups of a frame compute their upper levels (fine_levels_at), blockDim.x tile positions each -- no launch of their own.
__global__ __launch_bounds__(1024) void tile_sparse_table(float2 *tables, int tw, int th, unsigned long long *zero_me,
                                                          float2 *fine = nullptr, int fw = 0, int fh = 0)
{
    if (zero_me != nullptr && blockIdx.x == 0 && blockIdx.y == 0)
        for (int k = threadIdx.x; k < kCounterBytes / 16; k += blockDim.x) reinterpret_cast<uint4 *>(zero_me)[k] = make_uint4(0u, 0u, 0u, 0u);
    const int lj = tile_levels(tw), li = tile_levels(th), n = tw * th;
    if ((int)blockIdx.x >= lj) {       // workgroup-uniform
        const int k = ((int)blockIdx.x - lj) * (int)blockDim.x + (int)threadIdx.x;
        if (fine != nullptr && k < fw * fh) fine_levels_at(fine + (size_t)blockIdx.y * fine_table_elems(fw, fh), fw, fh, k);
        return;
    }
    const int j = blockIdx.x;
    float2 *T = tables + (size_t)blockIdx.y * tile_table_elems(tw, th);
    __shared__ float2 buf[2][kTileLdsEntries];
    float2 *cur = buf[0], *nxt = buf[1];
    for (int k = threadIdx.x; k < n; k += blockDim.x) cur[k] = T[k];
    __syncthreads();
    for (int s = 0; s < j; ++s) {              // row levels (0, 1) .. (0, j)
        const int step = 1 << s;
        for (int k = threadIdx.x; k < n; k += blockDim.x) {
            const int ty = k / tw, tx = k - ty * tw;
            const float2 a = cur[k], b = cur[ty * tw + min(tx + step, tw - 1)];
            nxt[k] = make_float2(fminf(a.x, b.x), fmaxf(a.y, b.y));
        }
        __syncthreads();
        float2 *t = cur; cur = nxt; nxt = t;
    }
    if (j > 0)
        for (int k = threadIdx.x; k < n; k += blockDim.x) T[(size_t)j * n + k] = cur[k];     // level (0, j); (0, 0) is the input
    for (int i = 1; i < li; ++i) {             // column levels (1, j) .. (li - 1, j)
        const int step = 1 << (i - 1);
        float2 *dst = T + (size_t)(i * lj + j) * n;
        for (int k = threadIdx.x; k < n; k += blockDim.x) {
            const int ty = k / tw;
            const float2 a = cur[k], b = cur[k + (min(ty + step, th - 1) - ty) * tw];
            const float2 c = make_float2(fminf(a.x, b.x), fmaxf(a.y, b.y));
            nxt[k] = c;
            dst[k] = c;
        }
        __syncthreads();
        float2 *t = cur; cur = nxt; nxt = t;
    }
}

// ---- fine tiles for brick-sized boxes (round 4) ------------------------------------------------------------------------------
// A wavefront brick of 8 x 4 x 8 voxels projects onto 5 - 25 pixels, about the size of ONE 8-pixel tile, so the depth range
// the sparse table returns for it is that of the two or three tiles its box touches: on a slanted surface several centimetres
// more than the depths under the brick itself.  tools/claim_sim.py (the claim rule replayed on the CPU, S-surf 512^3): with
// 8-pixel tiles 10.3 % of the bricks stay undecided per frame, with 4-pixel tiles 8.5 % (with per-pixel bounds 7.3 %; 1.5 %
// hold a voxel of the truncation band).  A full sparse table over 4-pixel tiles would be 8.6 MB per frame; a brick-sized box
// only ever spans a few of them, so the fine table holds just the nine levels (2^ky x 2^kx tiles, ky, kx <= 2: boxes of up to
// 7 x 7 tiles = 28 pixels; 1.4 MB per 640 x 480 frame) and classify_patch falls back to the coarse table for larger boxes (the
// super-bricks of the pre-pass, bricks close to the camera).  Same tile semantics as the coarse table (see the top of this file):
// x = smallest depth if every pixel passes the reference's range test else -inf, y = largest passing depth or -inf, a NaN
// poisons its tile to (-inf, +inf); levels combine by min / max -- exact and order-free, so any evaluation order gives the
// table fine_table_reference computes pixel by pixel (tsdf_selftest_tile_tables compares them bit for bit).
struct FineTileParams {
    const float *depth[kMaxFramesPerLaunch];
    const uint8_t *mask[kMaxFramesPerLaunch];
    float2 *fine;           // n_frames tables of fine_table_elems(fw, fh)
    int H, W, fw, fh;
    float max_depth;
};

// Level (0, 0): one thread per tile, four rows of four pixels (consecutive threads read consecutive 16-byte pieces of a row).
// grid = (ceil(fw / 64), fh, frames), block = 64.
__global__ __launch_bounds__(64) void fine_tile_base(FineTileParams tp)
{
    const int tx = blockIdx.x * 64 + threadIdx.x, ty = blockIdx.y, f = blockIdx.z;
    if (tx >= tp.fw) return;
    const float *d = tp.depth[f];
    const uint8_t *m = tp.mask[f];
    const float inf = __builtin_inff();
    float mn = inf, mx = -inf;
    bool all_valid = true, nan = false;
    const int px0 = tx * kFineTile;
    const bool vec = (tp.W & 3) == 0 && (reinterpret_cast<uintptr_t>(d) & 15) == 0 && (m == nullptr || (reinterpret_cast<uintptr_t>(m) & 3) == 0);
#pragma unroll
    for (int r = 0; r < kFineTile; ++r) {
        const int py = ty * kFineTile + r;
        if (py >= tp.H) continue;
        const size_t at = (size_t)py * tp.W + px0;
        float v[4];
        bool in[4];
        if (vec) {                // W % 4 == 0: the four pixels are inside the row together
            const float4 q = *reinterpret_cast<const float4 *>(d + at);
            v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
            in[0] = in[1] = in[2] = in[3] = true;
            if (m != nullptr) {
                const uchar4 k = *reinterpret_cast<const uchar4 *>(m + at);
                v[0] = v[0] * (k.x >= 128 ? 1.0f : 0.0f); v[1] = v[1] * (k.y >= 128 ? 1.0f : 0.0f);   // inf * 0 = NaN, as in the kernel
                v[2] = v[2] * (k.z >= 128 ? 1.0f : 0.0f); v[3] = v[3] * (k.w >= 128 ? 1.0f : 0.0f);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                in[j] = px0 + j < tp.W;
                v[j] = in[j] ? d[at + j] : 0.0f;
                if (in[j] && m != nullptr) v[j] = v[j] * (m[at + j] >= 128 ? 1.0f : 0.0f);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!in[j]) continue;
            nan |= v[j] != v[j];
            const bool valid = (v[j] > 0.0f) & (v[j] <= tp.max_depth);
            all_valid &= valid;
            if (valid) { mn = fminf(mn, v[j]); mx = fmaxf(mx, v[j]); }
        }
    }
    float2 out;
    out.x = (all_valid && !nan) ? mn : -inf;
    out.y = nan ? inf : mx;
    tp.fine[(size_t)f * fine_table_elems(tp.fw, tp.fh) + (size_t)ty * tp.fw + tx] = out;
}

// The fine levels as a launch of their own (self-test; frames whose coarse tables do not fit tile_sparse_table's LDS): one thread per
// tile position.  grid = (ceil(fw * fh / 256), frames), block = 256.
__global__ __launch_bounds__(256) void fine_tile_levels(float2 *fine, int fw, int fh)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= fw * fh) return;
    fine_levels_at(fine + (size_t)blockIdx.y * fine_table_elems(fw, fh), fw, fh, k);
}

// The same table pixel by pixel (self-test only): one thread per (level, tile position).
__global__ __launch_bounds__(256) void fine_table_reference(FineTileParams tp)
{
    const int n = tp.fw * tp.fh;
    const int id = blockIdx.x * 256 + threadIdx.x, f = blockIdx.y;
    if (id >= kFineLevels * kFineLevels * n) return;
    const int level = id / n, k = id - level * n, ky = level / kFineLevels, kx = level - ky * kFineLevels;
    const int ty = k / tp.fw, tx = k - ty * tp.fw;
    const int x0 = tx * kFineTile, x1 = min((tx + (1 << kx)) * kFineTile, tp.W), y0 = ty * kFineTile, y1 = min((ty + (1 << ky)) * kFineTile, tp.H);
    const float inf = __builtin_inff();
    float mn = inf, mx = -inf;
    bool all_valid = true, nan = false;
    for (int y = y0; y < y1; ++y)
        for (int x = x0; x < x1; ++x) {
            float v = tp.depth[f][(size_t)y * tp.W + x];
            if (tp.mask[f] != nullptr) v = v * (tp.mask[f][(size_t)y * tp.W + x] >= 128 ? 1.0f : 0.0f);
            nan |= v != v;
            const bool valid = (v > 0.0f) & (v <= tp.max_depth);
            all_valid &= valid;
            if (valid) { mn = fminf(mn, v); mx = fmaxf(mx, v); }
        }
    // a NaN poisons only its own base tile to (-inf, +inf); min / max over the tiles of the block then give (-inf, +inf) too
    float2 out;
    out.x = (all_valid && !nan) ? mn : -inf;
    out.y = nan ? inf : mx;
    tp.fine[(size_t)f * fine_table_elems(tp.fw, tp.fh) + (size_t)level * n + k] = out;
}

// What classify_patch reads of a frame: 16 dwords, grouped as four 16-byte words so that the 32 frames of a launch can be kept
// structure-of-arrays (ClassPoseTable: lane f reads word k of frame f at w[k][f], 512 contiguous bytes per wave instruction
// instead of 64 lanes picking 72-byte blocks apart).
struct ClassPose {
    float rx0, ry0, rz0, tx;
    float rx1, ry1, rz1, ty;
    float rx2, ry2, rz2, tz;
    const float2 *tiles;
    float cz_short, cz_pad;
};
static_assert(sizeof(ClassPose) == 64, "four 16-byte words");

struct ClassPoseTable {
    float4 w[4][kMaxFramesPerLaunch];
};

__device__ __forceinline__ ClassPose class_pose(const FramePose &q)
{
    ClassPose c;
    c.rx0 = q.rx0; c.ry0 = q.ry0; c.rz0 = q.rz0; c.tx = q.tx;
    c.rx1 = q.rx1; c.ry1 = q.ry1; c.rz1 = q.rz1; c.ty = q.ty;
    c.rx2 = q.rx2; c.ry2 = q.ry2; c.rz2 = q.rz2; c.tz = q.tz;
    c.tiles = q.tiles; c.cz_short = q.cz_short; c.cz_pad = q.cz_pad;
    return c;
}

__device__ __forceinline__ ClassPose class_pose(const ClassPoseTable *__restrict__ t, const int f)
{
    union { ClassPose c; float4 w[4]; } u;
#pragma unroll
    for (int k = 0; k < 4; ++k) u.w[k] = t->w[k][f];
    return u.c;
}

__device__ __forceinline__ void class_pose_store(ClassPoseTable *t, const int f, const ClassPose &c)
{
    union { ClassPose c; float4 w[4]; } u;
    u.c = c;
#pragma unroll
    for (int k = 0; k < 4; ++k) t->w[k][f] = u.w[k];
}

// What one frame does to ALL voxels of a wavefront's patch -- the rectangle x in [xa, xb], y in [ya, yb] of slice
// gz that contains them -- decided from the patch's corners and the depth tile tables, without projecting a
// single voxel:
//   1  every voxel is updated with dist = 1: all of them project inside the image, onto tiles whose pixels are
//      all valid and at least trunc deeper than the farthest corner;
//   2  no voxel is updated: the patch misses the image, or every valid pixel it can reach is more than trunc
//      nearer than the nearest corner (the voxels lie behind the surface, ref: src/tsdf.cu:49);
//   0  no claim -- the per-voxel path decides.
// Evaluated once per workgroup in the kernel prologue with ONE LANE PER FRAME of the launch, so the frame loop only
// tests a bit.
// Why the claims are exact: the patch is planar and, when all corners are in front of the
// camera, projects into the convex hull of its projected corners; camera-frame z is affine over it, so its
// extremes are at the corners.  The corners are projected with ordinary fp32 arithmetic; the widening of the
// pixel box (px_margin_u / _v, derived in host_derive.h, derive_projection_guards: 0.5 px -- a rounded pixel index is within
// half a pixel of its u -- + the projection error of both paths, 3.2e-5 * (|f| + 4 (W + |c|)), valid for cz >= cz_short, +
// 1/16 px for the roundings of u = f * q + c itself) and cz_pad on the z bounds (twice the host's error bound on cz) cover
// the difference to the exact per-voxel values, and rounding is monotone: d - cz >= trunc in the reals implies RN(d - cz) >= trunc.  A NaN or an
// infinity in the corner arithmetic fails a comparison and yields 0.
// PAIRED: the two half-waves share the work on a box of slices -- lane l and lane l ^ 32 are given the same frame and
// the same rectangle, the lower one slice gz (the box's near slice), the upper one slice gz1 (its far slice); each
// projects the four corners of its slice and the six extremes are combined across the halves (min and max are exact,
// so the class is the one the eight-corner evaluation gives).  Every lane of the wavefront must make the call.
// fine: the frame's fine table (fine_tile_levels) or null; consulted when the box spans at most 7 x 7 fine tiles, i.e. when the
// four overlapping blocks of a level it holds cover the box -- the tiles then hug the box more closely than the coarse ones and
// the same comparisons decide more often; any table of correct tile bounds gives a correct claim.
template <bool PAIRED = false>
__device__ __forceinline__ int classify_patch(const IntegrateParams &p, const ClassPose &q, const int xa,
                                              const int xb, const int ya, const int yb, const int gz, const int gz1 = -1,
                                              const float2 *__restrict__ fine = nullptr)
{
    if constexpr (!PAIRED) { if (q.tiles == nullptr) return 0; }
    const float dxa = (p.ox + (float)xa * p.vs) - q.tx, dxb = (p.ox + (float)xb * p.vs) - q.tx;
    const float dya = (p.oy + (float)ya * p.vs) - q.ty, dyb = (p.oy + (float)yb * p.vs) - q.ty;
    const float inf = __builtin_inff();
    float umin = inf, umax = -inf, vmin = inf, vmax = -inf, czmin = inf, czmax = -inf;
    bool finite = true;
    // the four corners of one slice of the box: camera-frame coordinates are affine in x, y and z, so a box of slices
    // gz .. gz1 projects into the convex hull of the corners of its near and far slice (and has its z extremes there)
    auto slice_corners = [&](const int z) {
        const float dz = (p.oz + (float)z * p.vs) - q.tz;
        const float zx = q.rx2 * dz, zy = q.ry2 * dz, zz = q.rz2 * dz;
        auto corner = [&](const float dx, const float dy) {
            const float cx = q.rx0 * dx + q.rx1 * dy + zx;
            const float cy = q.ry0 * dx + q.ry1 * dy + zy;
            const float cz = q.rz0 * dx + q.rz1 * dy + zz;
            const float inv = __builtin_amdgcn_rcpf(cz);
            const float u = p.fx * (cx * inv) + p.cx, v = p.fy * (cy * inv) + p.cy;
            umin = fminf(umin, u); umax = fmaxf(umax, u);
            vmin = fminf(vmin, v); vmax = fmaxf(vmax, v);
            czmin = fminf(czmin, cz); czmax = fmaxf(czmax, cz);
            finite &= (u == u) & (v == v) & (cz == cz);     // fmin/fmax drop a NaN operand: keep it visible
        };
        corner(dxa, dya); corner(dxb, dya);
        corner(dxa, dyb); corner(dxb, dyb);   // the same two again when ya == yb
    };
    if constexpr (PAIRED) {
        slice_corners((threadIdx.x & 32u) ? max(gz1, gz) : gz);
        umin = fminf(umin, __shfl_xor(umin, 32)); umax = fmaxf(umax, __shfl_xor(umax, 32));
        vmin = fminf(vmin, __shfl_xor(vmin, 32)); vmax = fmaxf(vmax, __shfl_xor(vmax, 32));
        czmin = fminf(czmin, __shfl_xor(czmin, 32)); czmax = fmaxf(czmax, __shfl_xor(czmax, 32));
        finite = finite & (__shfl_xor((int)finite, 32) != 0);
        if (q.tiles == nullptr) return 0;
    } else {
        slice_corners(gz);
        if (gz1 > gz) slice_corners(gz1);
    }
    if (!(finite & (czmin > q.cz_short) & (czmax < 3.0e38f))) return 0;
    // pixel box that contains the rounded pixel of every voxel of the patch
    const float u0 = umin - p.px_margin_u, u1 = umax + p.px_margin_u;
    const float v0 = vmin - p.px_margin_v, v1 = vmax + p.px_margin_v;
    const float wmax = (float)(p.W - 1), hmax = (float)(p.H - 1);
    if (!(u1 >= 0.0f) | !(v1 >= 0.0f) | !(u0 <= wmax) | !(v0 <= hmax)) return 2;   // the box misses the image
    const bool inside = (u0 >= 0.0f) & (v0 >= 0.0f) & (u1 <= wmax) & (v1 <= hmax);
    const float uc0 = fmaxf(u0, 0.0f), uc1 = fminf(u1, wmax), vc0 = fmaxf(v0, 0.0f), vc1 = fminf(v1, hmax);
    int tx0 = (int)(uc0 * p.tile_inv), tx1 = (int)(uc1 * p.tile_inv);     // tile_inv: a power of two, exact
    int ty0 = (int)(vc0 * p.tile_inv), ty1 = (int)(vc1 * p.tile_inv);
    // range query: four overlapping power-of-two blocks of level (ky, kx)
    int kx = 31 - __clz(tx1 - tx0 + 1), ky = 31 - __clz(ty1 - ty0 + 1);
    int tw = p.tiles_w;
    const float2 *L = q.tiles + (size_t)(ky * tile_levels(p.tiles_w) + kx) * (p.tiles_w * p.tiles_h);
    if (fine != nullptr) {
        const float finv = 1.0f / (float)kFineTile;
        const int fx0 = (int)(uc0 * finv), fx1 = (int)(uc1 * finv), fy0 = (int)(vc0 * finv), fy1 = (int)(vc1 * finv);
        const int fkx = 31 - __clz(fx1 - fx0 + 1), fky = 31 - __clz(fy1 - fy0 + 1);
        if (fkx < kFineLevels && fky < kFineLevels) {
            tx0 = fx0; tx1 = fx1; ty0 = fy0; ty1 = fy1; kx = fkx; ky = fky; tw = p.fine_w;
            L = fine + (size_t)(fky * kFineLevels + fkx) * (p.fine_w * p.fine_h);
        }
    }
    const int xb2 = tx1 - (1 << kx) + 1, yb2 = ty1 - (1 << ky) + 1;
    const float2 a = L[ty0 * tw + tx0], b = L[ty0 * tw + xb2];
    const float2 c = L[yb2 * tw + tx0], d = L[yb2 * tw + xb2];
    const float dmin = fminf(fminf(a.x, b.x), fminf(c.x, d.x));   // -inf unless every pixel of the box is valid
    const float dmax = fmaxf(fmaxf(a.y, b.y), fmaxf(c.y, d.y));   // the deepest valid pixel of the box
    const float thr_free = (czmax + q.cz_pad) + p.trunc;    // every pixel at least this deep: dist = 1 everywhere
    const float thr_skip = (czmin - q.cz_pad) - p.trunc;    // every valid pixel at most this deep: nothing updated
    if (inside & (dmin >= thr_free)) return 1;
    if (dmax <= thr_skip) return 2;
    return 0;
}

// One masked frame into one volume, with bricks: the class of every wavefront brick, one thread per brick; index =
// (slice group * blocks + workgroup) * 4 + wavefront, blocks = workgroups per slice group (brick_s slices each).
__global__ __launch_bounds__(256) void classify_bricks(IntegrateParams p, FramePose pose, uint8_t *cls, int blocks, int nzg)
{
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= blocks * nzg * 4) return;
    const int wave = id & 3, wg = (id >> 2) % blocks, lz = (id >> 2) / blocks;
    const int brick = wg * 4 + wave;
    const int g = brick / p.bricks_per_group, i = brick - g * p.bricks_per_group;
    int c = 2;
    if (g < p.brick_groups) {
        const int xa = i * p.brick_q * 4, ya = g * p.brick_r;
        const int z0 = lz * p.brick_s, z1 = min(z0 + p.brick_s - 1, p.nz - 1);
        c = classify_patch(p, class_pose(pose), xa, xa + p.brick_q * 4 - 1, ya, min(ya + p.brick_r - 1, p.dim_y - 1), p.z_begin + z0,
                           p.z_begin + z1);
    }
    cls[id] = (uint8_t)c;
}

// The batched form with bricks: one thread per wavefront brick of the launch; index = (slice group of the launch *
// max_blocks + workgroup) * 4 + wavefront.  Each object brings its own brick view (IntegrateParams::brick_*);
// group_map[z] = {object, slice group within it}.
__global__ __launch_bounds__(256) void classify_bricks_batched(const IntegrateParams *__restrict__ params,
                                                               const FramePose *__restrict__ poses,
                                                               const int2 *__restrict__ group_map, uint8_t *cls,
                                                               int max_blocks, int total_groups)
{
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= max_blocks * total_groups * 4) return;
    const int wave = id & 3, wg = (id >> 2) % max_blocks, z = (id >> 2) / max_blocks;
    const int2 m = group_map[z];
    const IntegrateParams p = params[m.x];
    const int brick = wg * 4 + wave;
    const int g = brick / p.bricks_per_group, i = brick - g * p.bricks_per_group;
    int c = 2;                                   // a brick past the end of this object's slice: nothing there
    if (g < p.brick_groups) {
        const int xa = i * p.brick_q * 4, ya = g * p.brick_r;
        const int z0 = m.y * p.brick_s, z1 = min(z0 + p.brick_s - 1, p.nz - 1);
        c = classify_patch(p, class_pose(poses[m.x]), xa, xa + p.brick_q * 4 - 1, ya, min(ya + p.brick_r - 1, p.dim_y - 1), p.z_begin + z0,
                           p.z_begin + z1);
    }
    cls[id] = (uint8_t)c;
}

// Slab arrays of the per-voxel label state (tsdf_labels.hip.h), for the LABELS kernels.
struct LabelState {
    uint16_t *label;
    float *fp, *bp;
    float prob_thd;
};

// The same with the frame blocks inside the kernarg itself: nothing to stage in device memory before the
// launch (a 4 us copy on the stream per launch -- 1 % of a 512^3 pass, 8 % of a 200^3 one).  The kernel
// reads them through the kernarg segment pointer with the loop's wave-uniform index (scalar loads),
// not through the by-value parameter, which the compiler would copy to registers frame by frame.
struct MultiParamsInline {
    IntegrateParams common;
    FramePose frames[kMaxFramesPerLaunch];
    int n_frames;
    LabelState labels;   // read by the LABELS kernels only
    // workgroup order: 0 = memory order (x blocks fastest, then y, then slices), 1 = slices fastest (grid launched as
    // (slices, y blocks, x blocks)): consecutively dispatched workgroups then share their (x, y) footprint and with
    // it the windows of the launch's depth frames they gather from; 2 = slices fastest AND rotated: the workgroup at
    // grid position (z, y, x) takes slice group (z + x + y) mod n.  Workgroups are dealt to the 8 XCDs by their linear
    // index, so with a slice count that is a multiple of 8 order 1 gives every slice to ONE XCD -- and a surface that
    // lies across few slices (a wall facing the camera) then keeps one or two XCDs busy while the others idle
    int z_fastest;
#ifdef TSDF_EXPERIMENTS
    // (integrate_multi_wg) per super-brick (a workgroup's four bricks x kSuperZ consecutive slice groups) the frames that
    // may do something to it (classify_superbricks); a workgroup whose word is 0 leaves before it stages or classifies
    // anything.  Null: no such table.  Index = workgroup index within the slice group * nz_super + slice group / kSuperZ.
    const unsigned int *super_mask;
    int nz_super;
#endif
};

constexpr int kSuperZ = 4;

// FLAT: the lane's quad comes from the linear view of the slice (IntegrateParams::quads_per_slice):
// wavefront = 64 consecutive quads in memory order, whatever dim_x % 4 == 0 is -- rows shorter than or
// not a multiple of 256 voxels no longer leave lanes idle (200-voxel rows: 50 of 64 lanes in the row
// mapping).  b0 = block index within the slice (4 chunks per block).  Needs R == 1.
// !FLAT: the row mapping of integrate_tile (b0, b1 = x-block, y-block), dim_x % 256 == 0.
// LABELS: the label evidence of each frame (tsdf_labels.hip.h: same rule, same voxels) is applied in the same
// pass, from the projection and depth tests Integrate has just made -- the separate label sweep recomputes both.
// MASKS = false: the host promises that no frame of the launch carries an instance mask, and the kernel holds
// no mask bytes, defaults or null tests (2 of 44 VALU instructions per voxel-frame).
// SHORT: the frames of the launch that come with depth tile tables are classified per wavefront in the prologue, one
// lane per frame (classify_patch): all voxels updated with dist = 1, or none updated, without projecting any of them.
// BRICK: a wavefront owns a compact brick of the slice -- brick_q quads of brick_r consecutive rows (IntegrateParams:
// 64 x 4 voxels for 512-voxel rows, 40 x 6 for the reference's 200-voxel rows) -- instead of 256 consecutive voxels; b0 =
// the wavefront's brick index within the slice group, lane l owns quad (l % brick_q) of the brick's row (l / brick_q).  A wave-instruction then touches brick_r row pieces of 16 * brick_q bytes instead of one 1-KiB piece
// (the fused launches are bound by instruction issue, not by HBM), and the wavefront's voxels project onto a compact
// pixel box, so a depth tile table can decide far more wavefront-frames without projecting a voxel: on S-surf an ideal
// classifier claims 38 % of 256 x 1 rows but 77 % of 64 x 4 bricks (a row crosses both image borders and every
// silhouette on its way).  The free-space summary keeps its layout (one word per 256-voxel chunk or row segment, now
// shared by several wavefronts): a set bit was true for the whole chunk at launch start, each lane only changes its own
// voxels -- `ones` is per lane here: "the chunk's bit was set and MY quad is still all ones" -- and clearing is idempotent.
// EAGER_W (the bricks of a work list: some frame touches them): the weights are requested together with the summary word at the
// top instead of by the first frame that touches the lane -- one memory round trip less in a wavefront's chain of dependent
// loads (list entry -> classification tables -> summary word -> quads), which is what the many light wavefronts of a sparse
// launch spend their time in.
template <int R, bool NT, bool FLAT, bool LABELS = false, bool MASKS = true, bool SHORT = false, bool BRICK = false, bool EAGER_W = false>
__device__ __forceinline__ void multi_body(const IntegrateParams &p, const FramePose *__restrict__ frames,
                                           const int n_frames, const int b0, const int b1, const int lz,
                                           const LabelState ls = LabelState(), const unsigned int free_frames = 0u,
                                           const unsigned int skip_frames = 0u)
{
    static_assert(!SHORT || R == 1, "patch classification is written for one row per lane");
    static_assert(!LABELS || R == 1, "label fusion rides on the one-row kernel");
    static_assert(!FLAT || R == 1, "the flat mapping handles one quad per lane");
    static_assert(!BRICK || (R == 1 && !FLAT), "bricks are their own mapping");
    static_assert(!EAGER_W || BRICK, "eager weights are for the bricks of a work list");
    int xg, gy0;
    size_t row0, flag0;
    // BRICK: the lane's quad as (wave-uniform start of the brick) + (32-bit byte offset of the lane within it).  With ordinary
    // (temporal) accesses the two arrays are reached through buffer descriptors in scalar registers -- one offset register per
    // lane instead of two 64-bit addresses held across the frame loop (the kernel sits at its 64-register budget: 5 spilled
    // registers before, none of them now).  brick_shape_ok keeps a brick's span below 4 GiB.
    size_t brick_base = 0;
    unsigned int lane_byte = 0u;
    // (BRICK) the lane's summary word as a 32-bit index, kept across the frame loop for the write-back instead of the three
    // coordinates it is made from (with the row test below, which a one-row lane has already passed, this is what took the
    // masked work-list kernel from 1 VGPR + 4 SGPRs spilled to none at its 64-register budget): a slab has one word per 256
    // voxels, so the index fits while the slab holds fewer than 2^40 voxels (4 TB of volume: no device holds that)
    uint32_t flag32 = 0u;
    constexpr bool kBuf = BRICK && !NT;
    __amdgpu_buffer_rsrc_t t_buf, w_buf;
    auto load_quad = [&](const bool tsdf, const int r) -> float4 {
        if constexpr (kBuf) {
            const v4u v = __builtin_amdgcn_raw_buffer_load_b128(tsdf ? t_buf : w_buf, (int)lane_byte, 0, 0);
            return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
        } else {
            return vol_load<NT>((tsdf ? p.tsdf : p.weight) + row0 + (size_t)r * p.dim_x);
        }
    };
    auto store_quad = [&](const bool tsdf, const int r, const float4 q) {
        if constexpr (kBuf) {
            const v4u v = {__float_as_uint(q.x), __float_as_uint(q.y), __float_as_uint(q.z), __float_as_uint(q.w)};
            __builtin_amdgcn_raw_buffer_store_b128(v, tsdf ? t_buf : w_buf, (int)lane_byte, 0, 0);
        } else {
            vol_store<NT>((tsdf ? p.tsdf : p.weight) + row0 + (size_t)r * p.dim_x, q);
        }
    };
    int lzz = lz;            // the lane's slice of the slab (BRICK: lz counts groups of brick_s slices)
    // the summary word of the lane's quad: row segment (dim_x % 256 == 0) or 256-voxel chunk of the slice's linear view
    auto brick_flag_index = [&](const int gy, const int quad) -> size_t {
        if (p.nseg > 0) return ((size_t)lzz * p.dim_y + gy) * (size_t)p.nseg + (size_t)(quad >> 6);
        return (size_t)lzz * p.chunks_per_slice + (size_t)((gy * p.quads_per_row + quad) >> 6);
    };
    if constexpr (BRICK) {
        const int brick = b0;   // BRICK: b0 is the wavefront's brick within the slice group (wave-uniform)
        const int g = brick / p.bricks_per_group, i = brick - g * p.bricks_per_group;
        // lane -> (slice, row, quad) of the brick: two divisions of a number below 64 by a divisor of at most 64, exact
        // as (x * ceil(2^16 / d)) >> 16 (the host's brick_per_magic / brick_q_magic)
        const int per = p.brick_q * p.brick_r;
        const int zz = (int)(((unsigned)threadIdx.x * (unsigned)p.brick_per_magic) >> 16), rem = (int)threadIdx.x - zz * per;
        const int rr = (int)(((unsigned)rem * (unsigned)p.brick_q_magic) >> 16), qq = rem - rr * p.brick_q;
        xg = i * p.brick_q + qq;
        gy0 = g * p.brick_r + rr;
        lzz = lz * p.brick_s + zz;
        if (g >= p.brick_groups || zz >= p.brick_s || lzz >= p.nz || gy0 >= p.dim_y) return;
        brick_base = ((size_t)(lz * p.brick_s) * p.dim_y + (size_t)(g * p.brick_r)) * (size_t)p.dim_x + (size_t)(i * p.brick_q) * 4;
        lane_byte = (((unsigned)zz * (unsigned)p.dim_y + (unsigned)rr) * (unsigned)p.dim_x + (unsigned)qq * 4u) * 4u;
        row0 = brick_base + (size_t)(lane_byte >> 2);
        if constexpr (kBuf) {   // raw buffers (stride 0, byte offsets, no bound: the host guarantees the span), gfx950 descriptor word
            t_buf = __builtin_amdgcn_make_buffer_rsrc(p.tsdf + brick_base, 0, -1, 0x00020000);
            w_buf = __builtin_amdgcn_make_buffer_rsrc(p.weight + brick_base, 0, -1, 0x00020000);
        }
        flag0 = brick_flag_index(gy0, xg);
        flag32 = (uint32_t)flag0;
    } else if constexpr (FLAT) {
        const int chunk = b0 * 4 + threadIdx.y;
        const int q = chunk * 64 + threadIdx.x;
        if (q >= p.quads_per_slice) return;
        gy0 = q / p.quads_per_row;
        xg = q - gy0 * p.quads_per_row;
        row0 = ((size_t)lz * p.quads_per_slice + q) * 4;
        flag0 = (size_t)lz * p.chunks_per_slice + chunk;
    } else {
        xg = b0 * 64 + threadIdx.x;
        gy0 = (b1 * 4 + threadIdx.y) * R;
        if (xg >= p.xgroups || gy0 >= p.dim_y) return;
        row0 = ((size_t)lz * p.dim_y + gy0) * (size_t)p.dim_x + (size_t)xg * 4;
        flag0 = ((size_t)lz * p.dim_y + gy0) * (size_t)p.nseg + b0;
    }
    const int gz = p.z_begin + lzz;

    // ---- voxel state held in registers across the frames ----------------------------------------
    uint32_t fl[R];
    bool ones[R];            // every TSDF value of the row segment is (still) exactly 1 (wave-uniform); BRICK: see above
    float4 t4[R], w4[R];
    bool touched[R], tchanged[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        fl[r] = (R == 1 || gy0 + r < p.dim_y) ? p.flags[flag0 + (size_t)r * p.nseg] : 0u;
        ones[r] = (fl[r] & 1u) != 0u;
        t4[r] = make_float4(1.f, 1.f, 1.f, 1.f);
        w4[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        touched[r] = tchanged[r] = false;
    }
    bool have_w = false;     // (EAGER_W) the weights are already in registers
    if constexpr (EAGER_W) {
        w4[0] = load_quad(false, 0);
        have_w = true;
    }

    // pose-independent voxel coordinates (ref: src/tsdf.cu:27-29)
    float bxv[4], byv[R];
#pragma unroll
    for (int j = 0; j < 4; ++j) bxv[j] = p.ox + (float)(xg * 4 + j) * p.vs;
#pragma unroll
    for (int r = 0; r < R; ++r) byv[r] = p.oy + (float)(gy0 + r) * p.vs;
    const float bz = p.oz + (float)gz * p.vs;
    const v2f F = {p.fx, p.fy}, C = {p.cx, p.cy};
    const float trunc_r1 = refined_rcp(p.trunc);   // for fast_div_r(diff, trunc): once per kernel
    bool fast_frame = false;                       // wave-uniform: the current frame took the fast projection path

    // What a frame does to the lane's voxels once upd / diff are known (ref: src/tsdf.cu:53-57): the quads are brought in
    // on first touch and updated in registers.  Shared by the per-voxel path and the classified free-space path.
    float diff[R][4];
    bool upd[R][4], rowany[R], bandr[R];
    bool any = false, band = false;
    auto apply_frame = [&]() __attribute__((always_inline)) {
        // ---- first touch: bring the quads in -------------------------------------------------------
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (rowany[r] && !touched[r]) {
                if (!have_w) w4[r] = load_quad(false, r);
                if (!(fl[r] & 1u)) t4[r] = load_quad(true, r);
                touched[r] = true;
            }
        }
        float dist[R][4];
        if (__ballot(band) != 0ull) {
            if (fast_frame && p.trunc_fast != 0) {   // the same quotient from the shared refined reciprocal (fast_div_r)
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) dist[r][j] = fminf(1.0f, fast_div_r(diff[r][j], p.trunc, trunc_r1));
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) dist[r][j] = fminf(1.0f, diff[r][j] / p.trunc);  // ref: :53
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r)
#pragma unroll
                for (int j = 0; j < 4; ++j) dist[r][j] = 1.0f;
        }

        // ---- update in registers (ref: src/tsdf.cu:54-57) ----------------------------------------------
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (__ballot(!(ones[r] && (fl[r] & 2u)) || bandr[r]) == 0ull) {
                // free space (see integrate_tile) for every lane of the wavefront: the TSDF rows stay 1, only the
                // weights move.  (A lane whose row is all ones but whose neighbours' are not takes the general path
                // below on the constant 1 that stands in for its values: the same bits.)
                if (upd[r][0]) w4[r].x += 1.0f;
                if (upd[r][1]) w4[r].y += 1.0f;
                if (upd[r][2]) w4[r].z += 1.0f;
                if (upd[r][3]) w4[r].w += 1.0f;
                continue;
            }
            float tv[4] = {t4[r].x, t4[r].y, t4[r].z, t4[r].w};
            float wv[4] = {w4[r].x, w4[r].y, w4[r].z, w4[r].w};
            float num[4], wn[4];
            // a lane of the wavefront inside the truncation band (dist < 1): its quotient is needed and its value changes, so
            // the tests for "every quotient is exactly 1" and "no value changed" (three + one compares per voxel) are moot --
            // the division runs (x / x = 1 where the shortcut would have applied) and the row is stored
            const bool wave_band = __ballot(bandr[r]) != 0ull;
            bool need = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                wn[j] = wv[j] + 1.0f;
                num[j] = tv[j] * wv[j] + dist[r][j];
            }
            if (!wave_band) {
#pragma unroll
                for (int j = 0; j < 4; ++j) need |= upd[r][j] && !(num[j] == wn[j] && wn[j] < 3.0e38f && wn[j] > 0.0f);
            }
            float nt[4];
            if (wave_band || __ballot(rowany[r] && need) != 0ull) {
#pragma unroll
                for (int j = 0; j < 4; ++j) nt[j] = num[j] / wn[j];
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j) nt[j] = 1.0f;
            }
            bool changed = wave_band, notone = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float newt = upd[r][j] ? nt[j] : tv[j];
                if (!wave_band) changed |= __float_as_uint(newt) != __float_as_uint(tv[j]);
                notone |= upd[r][j] && __float_as_uint(newt) != 0x3f800000u;
                tv[j] = newt;
                wv[j] = upd[r][j] ? wn[j] : wv[j];
            }
            t4[r] = make_float4(tv[0], tv[1], tv[2], tv[3]);
            w4[r] = make_float4(wv[0], wv[1], wv[2], wv[3]);
            tchanged[r] |= changed;
            {   // a value != 1 appeared in the row segment: for every lane of the wavefront that shares the row
                const unsigned long long nb = __ballot(notone);
                if constexpr (BRICK) { if (notone) ones[r] = false; }   // per lane: it is this lane that clears the word
                else { if (nb != 0ull) ones[r] = false; }
            }
        }
    };

    // not unrolled: one frame's temporaries at a time (unrolling interleaves frames: 100 VGPRs)
    // SHORT: the loop visits only the frames that are not skipped (bit scan over the claim word: a wavefront most of
    // whose frames see nothing does not pay a loop iteration of scalar bookkeeping for each of them)
    unsigned int todo = n_frames >= 32 ? 0xffffffffu : ((1u << n_frames) - 1u);
    if constexpr (SHORT) {
        if (p.shortcut_stats != nullptr && (threadIdx.x & 63) == 0 && (skip_frames & todo) != 0u)
            atomicAdd(p.shortcut_stats + 2, (unsigned)__popc(skip_frames & todo));
        todo &= ~skip_frames;
    }
#pragma unroll 1
    for (int f = 0; f < n_frames; ++f) {
        if constexpr (SHORT) {
            if (todo == 0u) break;
            f = __ffs((int)todo) - 1;      // the next frame that does something; wave-uniform
            todo &= todo - 1u;
            const bool all_free = (free_frames >> f) & 1u;
            if (all_free) {
                // every voxel of the wavefront: valid pixel, diff >= trunc (dist = 1), none in the band
                if (__ballot(!((touched[0] || have_w) && ones[0] && (fl[0] & 2u))) == 0ull) {   // steady state of free-space rows: only the weights move
                    touched[0] = true;
                    // ... and a run of such frames moves them by its length at once, when that is the same bits: every
                    // weight an integer below 2^24 - 32, so that each of the run's "+ 1" is exact and so is their sum
                    const unsigned int rest = ~(free_frames >> f);
                    unsigned int run = rest != 0u ? (unsigned)(__ffs((int)rest) - 1) : 32u - (unsigned)f;   // >= 1
                    run = min(run, (unsigned)(n_frames - f));
                    const float4 w = w4[0];
                    const bool whole = w.x == truncf(w.x) && w.y == truncf(w.y) && w.z == truncf(w.z) && w.w == truncf(w.w) &&
                                       fmaxf(fmaxf(w.x, w.y), fmaxf(w.z, w.w)) < 16777184.0f;
                    if (run > 1u && __ballot(!whole) == 0ull) {
                        const float k = (float)run;
                        w4[0].x += k; w4[0].y += k; w4[0].z += k; w4[0].w += k;
                        todo &= ~((run >= 32u ? 0xffffffffu : ((1u << run) - 1u)) << f);
                        if (p.shortcut_stats != nullptr && (threadIdx.x & 63) == 0) atomicAdd(p.shortcut_stats + 1, run);
                        continue;
                    }
                    if (p.shortcut_stats != nullptr && (threadIdx.x & 63) == 0) atomicAdd(p.shortcut_stats + 1, 1u);
                    w4[0].x += 1.0f; w4[0].y += 1.0f; w4[0].z += 1.0f; w4[0].w += 1.0f;
                    continue;
                }
                if (p.shortcut_stats != nullptr && (threadIdx.x & 63) == 0) atomicAdd(p.shortcut_stats + 1, 1u);
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    rowany[r] = true;
                    bandr[r] = false;
#pragma unroll
                    for (int j = 0; j < 4; ++j) { upd[r][j] = true; diff[r][j] = p.trunc; }
                }
                band = false;
                apply_frame();
                continue;
            }
        }
        if constexpr (SHORT) {
            if (p.shortcut_stats != nullptr && (threadIdx.x & 63) == 0) atomicAdd(p.shortcut_stats, 1u);
        }
        any = false;
        band = false;
        {
        const FramePose q = frames[f];   // wave-uniform address: scalar loads (not needed by a classified frame)

        // ---- geometry of frame f (ref: src/tsdf.cu:33-43) ------------------------------------------
        float ax[4], ay[4], az[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float dx = bxv[j] - q.tx;
            ax[j] = q.rx0 * dx; ay[j] = q.ry0 * dx; az[j] = q.rz0 * dx;
        }
        const float dz = bz - q.tz;
        const float x2 = q.rx2 * dz, y2 = q.ry2 * dz, z2 = q.rz2 * dz;
        float pcz[R][4], dval[R][4];
        // The depth sample of a voxel, or 0 when its projection is rejected (ref: src/tsdf.cu:39-43): 0 fails
        // the depth test below exactly as the reference's `continue` does, so no separate validity flag
        // has to survive until then.  The load is skipped (exec-masked), not redirected.
        // The mask byte is fetched with it and applied after all loads of the frame have been issued.
        int mval[MASKS ? R : 1][4];
        uint32_t pixel[LABELS ? R : 1][4];   // (LABELS) the voxel's pixel, for the label and score images
        auto fetch = [&](const int r, const int j, const bool ok, const uint32_t px) {
            if constexpr (LABELS) pixel[r][j] = px;
            float d = 0.0f;
            int m = 255;
            if (ok) {
                d = gather_f32(q.depth, px);
                if constexpr (MASKS) if (q.mask != nullptr) m = q.mask[px];
            }
            dval[r][j] = d;
            if constexpr (MASKS) mval[r][j] = m;
        };
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float z1 = q.rz1 * (byv[r] - q.ty);
#pragma unroll
            for (int j = 0; j < 4; ++j) pcz[r][j] = az[j] + z1 + z2;
        }
        // corner test of the patch against the camera plane, see integrate_tile
        // (compares on the corners themselves: min/max would add two canonicalising moves and treat a NaN corner
        // as absent; a NaN fails every compare here and sends the wavefront down the generic path)
        bool all_front = (pcz[0][0] > q.cz_margin) & (pcz[0][3] > q.cz_margin);
        bool all_behind = (pcz[0][0] < -q.cz_margin) & (pcz[0][3] < -q.cz_margin);
        if constexpr (R > 1) {
            all_front &= (pcz[R - 1][0] > q.cz_margin) & (pcz[R - 1][3] > q.cz_margin);
            all_behind &= (pcz[R - 1][0] < -q.cz_margin) & (pcz[R - 1][3] < -q.cz_margin);
        }
        const bool unsafe = !(all_front | all_behind);
        fast_frame = q.fast_ok != 0 && __ballot(unsafe) == 0ull;
        if (fast_frame) {
            const bool front = all_front;   // the sign of cz over the whole patch (corner test)
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const bool row_ok = R == 1 || gy0 + r < p.dim_y;   // R == 1: a lane whose row is outside has left above
                const float dy = byv[r] - q.ty;
                const v2f XY1 = {q.rx1 * dy, q.ry1 * dy};
                const v2f XY2 = {x2, y2};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const v2f A = {ax[j], ay[j]};
                    const v2f n = A + XY1 + XY2;
                    const float cz = pcz[r][j];
                    const v2f uv = F * fast_div2(n, cz) + C;
                    const int iu = round_half_up_i32(uv.x), iv = round_half_up_i32(uv.y);
                    const bool ok = row_ok & front & (uv.x > -0.5f) & (uv.y > -0.5f) &
                                    ((unsigned)iu < (unsigned)p.W) & ((unsigned)iv < (unsigned)p.H);
                    fetch(r, j, ok, (uint32_t)pixel_index24(iv, p.W, iu));
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const bool row_ok = R == 1 || gy0 + r < p.dim_y;   // R == 1: a lane whose row is outside has left above
                const float dy = byv[r] - q.ty;
                const float x1 = q.rx1 * dy, y1 = q.ry1 * dy;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float cx = ax[j] + x1 + x2;
                    const float cy = ay[j] + y1 + y2;
                    const float cz = pcz[r][j];
                    const float pu = roundf(p.fx * (cx / cz) + p.cx);
                    const float pv = roundf(p.fy * (cy / cz) + p.cy);
                    const bool ok = row_ok && !(cz <= 0.0f) && pu >= 0.0f && pu < (float)p.W && pv >= 0.0f &&
                                    pv < (float)p.H;
                    fetch(r, j, ok, ok ? (uint32_t)((int)pv * p.W + (int)pu) : 0u);
                }
            }
        }

        if constexpr (MASKS) {
            if (q.mask != nullptr) {   // wave-uniform; ref: src/Engine.cpp:192-193
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int j = 0; j < 4; ++j) dval[r][j] = dval[r][j] * (mval[r][j] >= 128 ? 1.0f : 0.0f);
            }
        }

        // ---- depth tests (ref: src/tsdf.cu:46-49) ----------------------------------------------------
#pragma unroll
        for (int r = 0; r < R; ++r) {
            rowany[r] = false;
            bandr[r] = false;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float d = dval[r][j];
                const float df = d - pcz[r][j];
                diff[r][j] = df;
                const bool u = !((d <= 0.0f) | (d > p.max_depth)) & !(df <= -p.trunc);
                upd[r][j] = u;
                rowany[r] |= u;
                bandr[r] |= u & !(df >= p.trunc);
            }
            band |= bandr[r];
            any |= rowany[r];
        }
        if (__ballot(any) == 0ull) continue;   // this frame touches nothing here

        // ---- label evidence of this frame (tsdf_labels.hip.h: voxels observed inside the truncation band) ------
        if constexpr (LABELS) {
            if (q.label_im != nullptr && __ballot(band) != 0ull) {
                uint16_t lab_in[4];
                float sc_in[4];
                bool hit = false;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const bool seen = upd[0][j] & (diff[0][j] < p.trunc);
                    lab_in[j] = 0;
                    sc_in[j] = 0.0f;
                    if (seen) { lab_in[j] = q.label_im[pixel[0][j]]; sc_in[j] = q.score_im[pixel[0][j]]; }
                    hit |= lab_in[j] != 0;
                }
                if (hit) {
                    ushort4 L4 = *reinterpret_cast<const ushort4 *>(ls.label + row0);
                    float4 F4 = *reinterpret_cast<const float4 *>(ls.fp + row0);
                    float4 B4 = *reinterpret_cast<const float4 *>(ls.bp + row0);
                    uint16_t L[4] = {L4.x, L4.y, L4.z, L4.w};
                    float Fv[4] = {F4.x, F4.y, F4.z, F4.w}, Bv[4] = {B4.x, B4.y, B4.z, B4.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const uint16_t l = lab_in[j];
                        const float sc = sc_in[j];
                        if (l == 0) continue;
                        if (L[j] == 0) { L[j] = l; Fv[j] = sc; Bv[j] = 0.0f; }
                        else if (L[j] == l) { Fv[j] = Fv[j] + sc; }
                        else {
                            Bv[j] = Bv[j] + sc;
                            if (Fv[j] / (Fv[j] + Bv[j]) < ls.prob_thd) { L[j] = l; Fv[j] = sc; Bv[j] = 0.0f; }
                        }
                    }
                    ushort4 Lo; Lo.x = L[0]; Lo.y = L[1]; Lo.z = L[2]; Lo.w = L[3];
                    *reinterpret_cast<ushort4 *>(ls.label + row0) = Lo;
                    *reinterpret_cast<float4 *>(ls.fp + row0) = make_float4(Fv[0], Fv[1], Fv[2], Fv[3]);
                    *reinterpret_cast<float4 *>(ls.bp + row0) = make_float4(Bv[0], Bv[1], Bv[2], Bv[3]);
                }
            }
        }

        }

        apply_frame();
    }

    // ---- write back ------------------------------------------------------------------------------------
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const bool store_t = __ballot(touched[r] && tchanged[r]) != 0ull;
        if (touched[r]) {
            if (store_t) store_quad(true, r, t4[r]);
            store_quad(false, r, w4[r]);
        }
        if ((fl[r] & 1u) && !ones[r]) {
            if constexpr (BRICK) {
                p.flags[flag32] = fl[r] & 2u;
            } else {
                p.flags[flag0 + (size_t)r * p.nseg] = fl[r] & 2u;
            }
        }
    }
}

#ifndef TSDF_BRICK_WAVES
#define TSDF_BRICK_WAVES 8   /* waves per SIMD asked of the brick instantiation (A/B builds override it) */
#endif
// The per-voxel fused launch: up to 32 frames per pass, frame blocks read from the kernarg.  (The classified launches run over
// the brick work list: classify_brick_list / integrate_brick_list below.)
template <int R, bool NT, bool FLAT, bool LABELS = false, bool MASKS = true>
__global__ __launch_bounds__(256, R == 2 ? 6 : 1) void integrate_multi_inline(MultiParamsInline mp)
{
    // the single by-value parameter starts the kernarg segment (offset 0)
    typedef const char __attribute__((address_space(4))) *kernarg_ptr;
    kernarg_ptr base = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    typedef const FramePose __attribute__((address_space(4))) *frames_ptr;
    frames_ptr frames = (frames_ptr)(base + offsetof(MultiParamsInline, frames));
    const int wg_x = mp.z_fastest ? (int)blockIdx.z : (int)blockIdx.x, wg_y = (int)blockIdx.y;
    int wg_z = mp.z_fastest ? (int)blockIdx.x : (int)blockIdx.z;
    if (mp.z_fastest == 2) {
        wg_z += (int)((unsigned)(wg_x + wg_y) % gridDim.x);
        if (wg_z >= (int)gridDim.x) wg_z -= (int)gridDim.x;
    }
    multi_body<R, NT, FLAT, LABELS, MASKS>(mp.common, (const FramePose *)frames, mp.n_frames, wg_x, wg_y, wg_z, mp.labels);
}

// ---- live bricks only: a compacted work list -------------------------------------------------------------------------
// Most wavefront bricks of a realistic launch are skipped by every one of its frames (behind the surfaces, outside the
// views: 59 % on S-surf, more on a trajectory), and those that are not are concentrated in the surface band.  Dispatching
// a workgroup per four bricks of the whole slab and letting the dead ones leave (the first version: classify_superbricks
// + a classification in integrate_multi_inline's prologue) kept 5.4 of 8 wavefront slots per SIMD occupied, many of them by
// wavefronts on their way out, and the dispatcher walked through long runs of dead workgroups while finished slots stayed
// empty.  Now the Integrate launch only ever sees the bricks of super-bricks that some frame may touch:
//   classify_brick_list   one WAVEFRONT per super-brick = kSuperBX x kSuperBY x kSuperBZ bricks (32 x 8 x 8 voxels with
//                         the default 8 x 4 x 8-voxel brick), one frame per lane (paired half-waves: classify_patch<true>)
//                         on the super-brick's box.  Skipped by every frame: nothing is emitted.  Otherwise its bricks are
//                         appended to the work list -- ONE atomic per wavefront, one 16-byte entry per brick from as many
//                         lanes -- with what the box proved for each frame (a claim for the box holds for every brick inside
//                         it) and a flag when some frame is left undecided.
//   integrate_brick_list  one wavefront per list entry {brick, slice group, free frames, skipped frames}: no LDS, no barrier,
//                         no workgroup-wide prologue.  A brick whose super-brick left frames undecided classifies itself
//                         first (one frame per lane, as before -- in the shadow of the other wavefronts' per-voxel work) and
//                         leaves if every frame skips it.  The grid is sized for the worst case; the workgroups past the end
//                         of the sub-lists -- all at the END of the dispatch order -- leave on one scalar load.
// Same claims, same per-voxel code, same bits (every classified parity test runs through it).  Measured (rocprofv3, same
// box): the Integrate kernel of a 32-frame launch 1.132 -> 0.93 ms on S-surf 512^3, 5.89 -> 4.84 ms on the fr3 trajectory
// at 1024^3.  (A pre-pass that also classified every brick -- so that no wavefront would ever be launched to leave -- cost
// more than that saves: 0.20 ms and 2.1 ms per launch, sixteen dependent classifications per wavefront.)
// Bricks per super-brick along x, y (row groups), z (slice groups): 4 x 2 x 1 (32 x 8 x 8 voxels).  A smaller box is decided
// more often (fewer bricks left to classify themselves, fewer listed at all) but costs a pre-pass wavefront per box; four along
// x because a workgroup of integrate_brick_list takes four consecutive entries of a sub-list -- then the four x-neighbours of one
// row group and slice group, whose 32-byte row pieces make up whole 128-byte lines.  Measured, same box, ms per frame on the
// all-free-space launch / S-surf 512^3 / the fr3 trajectory 1024^3 / S-surf 200^3, and S-surf's HBM bytes per launch:
//   4x2x2  0.0095 / 0.0321 / 0.1614 / 0.00608   982 MB        4x2x1  0.0101 / 0.0312 / 0.1607 / 0.00585   975 MB
//   4x1x2  0.0101 / 0.0312 / 0.1614 / 0.00570   994 MB        4x1x1  0.0113 / 0.0320 / 0.1676 / 0.00578   932 MB
//   2x2x2  0.0113 / 0.0307 / 0.1600 / 0.00584  1245 MB (two x-neighbours per workgroup: half lines, a quarter more traffic
//   for 1.6 % of S-surf's time);  2x4x2 0.0314 / 0.1594;  2x2x1 0.0312 / 0.1654;  2x1x2 0.0310 / 0.1680;  4x4x2, 4x2x4, 8x2x2
//   0.0347-0.0353 / 0.165;  1x1x1 (every brick classified by the pre-pass) 0.0384 / 0.232.
constexpr int kSuperBX = 4, kSuperBY = 2, kSuperBZ = 1;
constexpr int kSuperBricks = kSuperBX * kSuperBY * kSuperBZ;
static_assert(kSuperBricks <= 32, "one lane of a half-wave per brick of the super-brick");
constexpr unsigned int kBrickUndecided = 0x80000000u;     // list entry flag: the brick has to classify itself

struct BrickListParams {
    uint4 *list;               // kListBuckets sub-lists of bucket_cap entries: a super-brick's bricks go to the sub-list its index hashes to
    unsigned char *counters;   // the launch's counter block (kCounterBytes), zeroed ahead of the pre-pass (tile_sparse_table)
    unsigned int bucket_cap;   // entries per sub-list: every brick of every super-brick it can receive
    ClassPoseTable *poses;     // the launch's frames for classify_patch, structure-of-arrays (written by the pre-pass)
    int nsx, nsy, nsz;         // super-bricks along x, y, z
    // list_bucket: wedge_mode != 0 deals the super-bricks to the XCDs by image rows (see there); the base camera's position in
    // voxel units relative to voxel (0, 0, z_begin) of the slab (rounded: the key only has to be the same on host and device),
    // the super-brick's height and depth in voxels, the focal length in pixels (rounded)
    int wedge_mode, wedge_oy, wedge_oz, super_h, super_d, fy_px;
};

// The sub-list (bucket) of super-brick `id` = (sx, sy, sz): workgroup j of integrate_brick_list takes its entries from sub-list
// j % kListBuckets, and workgroups are dealt to the eight XCDs round robin by their index, so sub-list b is served by XCD b % 8.
// wedge_mode 0 (what ships): a multiplicative hash of the index -- every sub-list, and with it every XCD, samples the whole
// volume: the sub-lists run dry together, but every XCD's L2 ends up fetching every depth frame of the launch (32 x 1.2 MB; an
// XCD has 4 MB).  wedge_mode 1 .. 3 (measurement build only, TSDF_WEDGE_MODE): the low three bits -- the XCD -- come from WHERE
// the super-brick lies in the image: its centre's row under the base camera, in stripes of 16 (mode 3: 8) pixels dealt round
// robin (a stripe is a wedge of the volume through the camera centre, so under any pose of the launch it projects onto a compact
// part of the image); modes 2 and 3 rotate the stripe -> XCD map by three every four super-bricks along x, so that the image's top
// and bottom rows -- where the bricks that straddle the border sit -- go round all XCDs.  Measured, round 4, same box, S-surf 512^3:
// the Integrate kernel's measured traffic 908 -> 558 / 599 / 692 MB per launch (modes 1 / 2 / 3: the depth frames are no longer
// fetched into all eight L2s) but 0.0256 -> 0.0305 / 0.0283 / 0.0276 ms per frame (4.5 - 5.0 resident wavefronts per SIMD instead
// of 5.6: the XCDs' shares of the per-voxel work are no longer alike, and the launch is bound by instruction issue, not by
// bytes); the fr3 trajectory 4.15 -> 3.82 GB, 0.1498 -> 0.1507.  Integer arithmetic only: the host counts the sub-lists'
// capacities with the same function.
__host__ __device__ inline unsigned int list_bucket(const unsigned int id, const int sx, const int sy, const int sz, const BrickListParams &bl)
{
    const unsigned int h = id * 2654435761u;
    if (bl.wedge_mode == 0) return h >> 26;
    const int yc = bl.wedge_oy + (2 * sy + 1) * bl.super_h / 2;
    int zc = bl.wedge_oz + (2 * sz + 1) * bl.super_d / 2;
    if (zc < 1) zc = 1;
    int t = (yc * 256) / zc;                             // 256 * tan of the elevation; |yc| < 2^16
    t = t > 65536 ? 65536 : (t < -65536 ? -65536 : t);   // (a tangent beyond 256 is outside any image; keeps the product below 2^31)
    const int stripe = (t * bl.fy_px) >> (bl.wedge_mode == 3 ? 11 : 12);       // rows / 16 or / 8 (arithmetic shift: floor)
    const int rot = bl.wedge_mode >= 2 ? 3 * (sx >> 2) : 0;
    return ((h >> 29) << 3) | ((unsigned int)(stripe + rot) & 7u);
}

__global__ __launch_bounds__(256) void classify_brick_list(MultiParamsInline mp, BrickListParams bl)
{
    typedef const char __attribute__((address_space(4))) *kernarg_ptr;
    kernarg_ptr base = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    const FramePose *frames = (const FramePose *)(base + offsetof(MultiParamsInline, frames));
    const IntegrateParams &p = mp.common;
    const int id = (int)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)threadIdx.y), lane = threadIdx.x;
    if (id >= bl.nsx * bl.nsy * bl.nsz) return;
    // slice groups fastest, then x, then y: neighbours in the list are neighbours in the volume (and in the depth frames)
    const int sz = id % bl.nsz, t = id / bl.nsz, sx = t % bl.nsx, sy = t / bl.nsx;
    const int nzg = (p.nz + p.brick_s - 1) / p.brick_s;
    const int i0 = sx * kSuperBX, g0 = sy * kSuperBY, zg0 = sz * kSuperBZ;
    const int i1 = min(i0 + kSuperBX, p.bricks_per_group) - 1, g1 = min(g0 + kSuperBY, p.brick_groups) - 1, zg1 = min(zg0 + kSuperBZ, nzg) - 1;
    const unsigned int frames_mask = mp.n_frames >= 32 ? 0xffffffffu : ((1u << mp.n_frames) - 1u);
    static_assert(kMaxFramesPerLaunch == 32, "one frame per lane of a half-wave");
    const int xa = i0 * p.brick_q * 4, xb = (i1 + 1) * p.brick_q * 4 - 1;
    const int ya = g0 * p.brick_r, yb = min((g1 + 1) * p.brick_r, p.dim_y) - 1;
    const int z0 = zg0 * p.brick_s, z1 = min((zg1 + 1) * p.brick_s, p.nz) - 1;
    const ClassPose mine_q = class_pose(frames[lane & 31]);
    // the frames as integrate_brick_list's self-classifying wavefronts read them (the first wavefront of the launch writes)
    if (id == 0 && lane < kMaxFramesPerLaunch) class_pose_store(bl.poses, lane, mine_q);
    const float2 *fine_f = p.fine != nullptr ? p.fine + (size_t)(lane & 31) * fine_table_elems(p.fine_w, p.fine_h) : nullptr;
    const int cls = classify_patch<true>(p, mine_q, xa, xb, ya, yb, p.z_begin + z0, p.z_begin + z1, fine_f);
    const unsigned int super_free = (unsigned int)__ballot(cls == 1) & frames_mask;
    const unsigned int super_skip = (unsigned int)__ballot(cls == 2) & frames_mask;
    const int n_in = (i1 - i0 + 1) * (g1 - g0 + 1) * (zg1 - zg0 + 1);
    // the sub-list of this super-brick: a multiplicative hash of its index, so that every sub-list samples the whole volume
    // (index mod 64 would be the slice group -- the sub-lists of slice groups in free space or behind the walls would run dry
    // long before the others and the launch would end on a quarter of its wavefronts)
    static_assert(kListBuckets == 64, "six bits: list_bucket");
    const unsigned int bkt = list_bucket((unsigned int)id, sx, sy, sz, bl);
    unsigned char *bucket = bl.counters + (size_t)bkt * kBucketStride;
    if (lane == 0) {
        // claims at super-brick granularity (what the per-launch decision is made from; the host adds the buckets up); the
        // diagnostic counters are exact: a super-brick skipped by every frame settles its bricks' wavefront-frames here, all
        // others are counted by the wavefronts of integrate_brick_list
        if (p.claim_counter != nullptr && (super_free | super_skip) != 0u)
            atomicAdd(reinterpret_cast<unsigned long long *>(bucket + 8),
                      ((unsigned long long)(n_in * __popc(super_free)) << 32) | (unsigned long long)(n_in * __popc(super_skip)));
        if (p.shortcut_stats != nullptr) {
            if (super_skip == frames_mask) { atomicAdd(p.shortcut_stats + 2, (unsigned)(n_in * mp.n_frames)); atomicAdd(p.shortcut_stats + 3, 1u); }
            else {
                atomicAdd(p.shortcut_stats + 4, (unsigned)n_in);
                if ((super_free | super_skip) != frames_mask) atomicAdd(p.shortcut_stats + 5, (unsigned)n_in);
            }
        }
    }
    if (super_skip == frames_mask) return;    // nothing in here sees anything
    const int k = lane;                       // lane k < kSuperBricks: brick k of the super-brick
    const int i = i0 + k % kSuperBX, g = g0 + (k / kSuperBX) % kSuperBY, zg = zg0 + k / (kSuperBX * kSuperBY);
    const bool mine = k < kSuperBricks && i <= i1 && g <= g1 && zg <= zg1;
    const unsigned long long live_mask = __ballot(mine);
    // Heavy work first: the bricks of a super-brick with undecided frames (they classify themselves and may take the
    // per-voxel path) grow the sub-list from its front, the bricks whose every frame is decided (a few weight additions
    // each) from its back; integrate_brick_list walks the front first.  A launch of few wavefronts per slot (a 200^3
    // volume: four) ends when its last long wavefront does, so the long ones must not start last.
    const unsigned int undecided = (super_free | super_skip) != frames_mask ? kBrickUndecided : 0u;
    const unsigned int n_live = (unsigned)__popcll(live_mask);
    unsigned int at = 0u;
    if (lane == 0) at = atomicAdd(reinterpret_cast<unsigned int *>(bucket) + (undecided ? 0 : 1), n_live);
    at = __builtin_amdgcn_readfirstlane(at);
    at = bkt * bl.bucket_cap + (undecided ? at : bl.bucket_cap - at - n_live);
    if (mine)
        bl.list[at + (unsigned)__popcll(live_mask & ((1ull << lane) - 1ull))] =
            make_uint4((unsigned)(g * p.bricks_per_group + i) | undecided, (unsigned)zg, super_free, super_skip);
}

template <bool NT, bool LABELS, bool MASKS>
__global__ __launch_bounds__(256, LABELS ? 6 : TSDF_BRICK_WAVES) void integrate_brick_list(MultiParamsInline mp, const uint4 *__restrict__ list,
                                                                                          const unsigned char *__restrict__ counters,
                                                                                          const unsigned int bucket_cap,
                                                                                          const ClassPoseTable *__restrict__ poses)
{
    typedef const char __attribute__((address_space(4))) *kernarg_ptr;
    kernarg_ptr base = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    const FramePose *frames = (const FramePose *)(base + offsetof(MultiParamsInline, frames));
    // workgroup j takes group j / kListBuckets of sub-list j % kListBuckets -- four consecutive entries, one per wavefront (four
    // x-neighbours, see kSuperBX) -- first the groups of the front part (bricks with undecided frames), then those of the back
    // part, walking down from the end; the sub-lists run dry together, at the end of the dispatch order
    const unsigned int b = blockIdx.x % kListBuckets, grp = blockIdx.x / kListBuckets;
    const unsigned int wave = (unsigned)__builtin_amdgcn_readfirstlane((int)threadIdx.y);
    const uint2 n = *reinterpret_cast<const uint2 *>(counters + (size_t)b * kBucketStride);   // {front, back} entries (scalar load)
    const unsigned int front_groups = (n.x + 3u) / 4u;
    unsigned int at;
    if (grp < front_groups) {
        at = grp * 4u + wave;
        if (at >= n.x) return;
    } else {
        const unsigned int k = (grp - front_groups) * 4u + wave;
        if (k >= n.y) return;
        at = bucket_cap - 1u - k;
    }
    const uint4 e = list[(size_t)b * bucket_cap + at];
    const int brick = (int)(e.x & ~kBrickUndecided), zg = (int)e.y;
    unsigned int free_frames = e.z, skip_frames = e.w;
    if (e.x & kBrickUndecided) {
        // the super-brick's box left frames undecided: the brick's own, smaller box may decide them
        const IntegrateParams &p = mp.common;
        const int lane = threadIdx.x;
        const unsigned int frames_mask = mp.n_frames >= 32 ? 0xffffffffu : ((1u << mp.n_frames) - 1u);
        const int g = brick / p.bricks_per_group, i = brick - g * p.bricks_per_group;
        const int xa = i * p.brick_q * 4, ya = g * p.brick_r;
        const int z0 = zg * p.brick_s, z1 = min(z0 + p.brick_s - 1, p.nz - 1);
        const float2 *fine_f = p.fine != nullptr ? p.fine + (size_t)(lane & 31) * fine_table_elems(p.fine_w, p.fine_h) : nullptr;
        const int cls = classify_patch<true>(p, class_pose(poses, lane & 31), xa, xa + p.brick_q * 4 - 1, ya, min(ya + p.brick_r - 1, p.dim_y - 1),
                                             p.z_begin + z0, p.z_begin + z1, fine_f);
        free_frames |= (unsigned int)__ballot(cls == 1) & frames_mask;
        skip_frames |= (unsigned int)__ballot(cls == 2) & frames_mask;
        free_frames = __builtin_amdgcn_readfirstlane(free_frames);
        skip_frames = __builtin_amdgcn_readfirstlane(skip_frames);
        if (skip_frames == frames_mask) {         // every frame skips the brick
            if (p.shortcut_stats != nullptr && lane == 0) { atomicAdd(p.shortcut_stats + 2, (unsigned)mp.n_frames); atomicAdd(p.shortcut_stats + 6, 1u); }
            return;
        }
    }
    multi_body<1, NT, false, LABELS, MASKS, true, true, true>(mp.common, frames, mp.n_frames, brick, 0, zg, mp.labels, free_frames, skip_frames);
}

// One frame, pose by value (no frame block in memory to stage): what a single tsdf_integrate* call
// on a volume served by the flat mapping launches.
template <bool NT, bool FLAT>
__global__ __launch_bounds__(256) void integrate_multi_single(IntegrateParams p, FramePose pose)
{
    multi_body<1, NT, FLAT>(p, &pose, 1, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Many volumes, one frame, one launch (the reference's real usage: one small TSDF per object
// instance, each fed depth x its own instance mask; ref: src/Engine.cpp:172-233, src/Object.cpp:67).
// params[o] / poses[o]: the parameter block and this frame's relative pose + mask of object o (each has
// its own base frame); slice_map[z] = {object, slice within it} for every slice of every object;
// grid = (max blocks per slice, 1, total slices).  Flat mapping: object grids are small and rarely
// 256 wide.  The blocks are read through a wave-uniform index (scalar loads).  (Launches large enough to repay a class
// table run over bricks: integrate_multi_batched_bricks.)
template <bool NT>
__global__ __launch_bounds__(256) void integrate_multi_batched(const IntegrateParams *__restrict__ params,
                                                               const FramePose *__restrict__ poses,
                                                               const int2 *__restrict__ slice_map)
{
    const int2 m = slice_map[blockIdx.z];
    const IntegrateParams p = params[m.x];
    multi_body<1, NT, true>(p, poses + m.x, 1, blockIdx.x, 0, m.y);
}

// One masked frame into one volume over bricks with a class per wavefront (classify_bricks); grid.z = slice groups.
template <bool NT>
__global__ __launch_bounds__(256, 8) void integrate_single_bricks(IntegrateParams p, FramePose pose)
{
    const unsigned c = p.wg_class[(blockIdx.x + gridDim.x * blockIdx.z) * 4u + threadIdx.y];   // wave-uniform
    if (c == 2u) return;
    multi_body<1, NT, false, false, true, true, true>(p, &pose, 1, (int)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)threadIdx.y), 0,
                                                      blockIdx.z, LabelState(), c == 1u ? 1u : 0u, 0u);
}

// The batched launch over bricks with a class per wavefront (classify_bricks_batched): per-object volumes are fed
// depth x their instance mask, so most bricks of most objects see nothing -- by rows, slices AND columns.
// grid.z = slice groups of the launch (group_map[z] = {object, slice group within it}).
template <bool NT>
__global__ __launch_bounds__(256, 8) void integrate_multi_batched_bricks(const IntegrateParams *__restrict__ params,
                                                                        const FramePose *__restrict__ poses,
                                                                        const int2 *__restrict__ group_map,
                                                                        const uint8_t *__restrict__ wg_class)
{
    const unsigned c = wg_class[(blockIdx.x + gridDim.x * blockIdx.z) * 4u + threadIdx.y];   // wave-uniform
    if (c == 2u) return;
    const int2 m = group_map[blockIdx.z];
    const IntegrateParams p = params[m.x];
    multi_body<1, NT, false, false, true, true, true>(p, poses + m.x, 1, (int)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)threadIdx.y), 0,
                                                      m.y, LabelState(), c == 1u ? 1u : 0u, 0u);
}

// The instance masks of one frame for up to kGatherMasks objects, copied into a batch's frame pool by ONE launch
// (tsdf_batch_integrate_device, deferred): grid = (chunks of 4 KiB, objects); a null source is skipped.
constexpr int kGatherMasks = 32;
struct MaskGatherParams {
    const uint8_t *src[kGatherMasks];
    uint8_t *dst[kGatherMasks];
    size_t bytes;      // per mask
};

__global__ __launch_bounds__(256) void gather_masks(MaskGatherParams gp)
{
    const uint8_t *src = gp.src[blockIdx.y];
    uint8_t *dst = gp.dst[blockIdx.y];
    if (src == nullptr) return;
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (i + 16 <= gp.bytes && (((uintptr_t)src | (uintptr_t)dst) & 15) == 0) {
        *reinterpret_cast<uint4 *>(dst + i) = *reinterpret_cast<const uint4 *>(src + i);
    } else {
        for (size_t k = i; k < gp.bytes && k < i + 16; ++k) dst[k] = src[k];
    }
}

}  // namespace tsdfk
