// tsdf_kernels.hip.h -- hand-written gfx950 kernels of the TSDF Integrate path.
//
// What is computed is the per-voxel update of the reference's GpuIntegrate
// (ref: src/tsdf.cu:15-60); how it is computed is new.  The reference runs one thread per
// (y,z) looping over x, so a wavefront touches 64 different rows at once; here a wavefront
// owns 256 consecutive x voxels of ONE row (4 per lane, one 16-byte load/store per array per
// lane = 1 KiB per wave-instruction, fully coalesced), the y/z dependent half of the
// projection is computed once per lane and shared by its 4 voxels, the intrinsics and the
// pose travel as kernel arguments (SGPRs) rather than global pointers, and a wavefront whose
// voxels all fail the geometric tests leaves before it touches the volume at all.
//
// Bit parity: every fp32 operation keeps the reference's order (file compiled with
// -ffp-contract=off, IEEE division, denormals on); the fast paths below only ever change
// how a value the reference would also compute is obtained when that is provably the same
// value (exact elisions, shared-reciprocal projection, one-instruction rounding),
// never the value itself.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tsdfk {

struct IntegrateParams {
    const float *depth;     // H*W fp32 metres, row-major
    const uint8_t *mask;    // H*W {0,255} or nullptr
    float *tsdf;            // slab, x-fastest
    float *weight;          // slab, x-fastest
    // intrinsics (ref: cam_K[0], cam_K[4], cam_K[2], cam_K[5])
    float fx, fy, cx, cy;
    // rotation of cam2base, named by the camera axis they feed (ref: src/tsdf.cu:36-38):
    //   cam_x = rx0*dx + rx1*dy + rx2*dz  with rx = cam2base[0], [4], [8]
    float rx0, rx1, rx2;
    float ry0, ry1, ry2;    // cam2base[1], [5], [9]
    float rz0, rz1, rz2;    // cam2base[2], [6], [10]
    float tx, ty, tz;       // cam2base[3], [7], [11]
    float ox, oy, oz;       // grid origin
    float vs, trunc, max_depth;
    int dim_x, dim_y;       // row length, rows per slice
    int nz, z_begin;        // slices in this slab, global z of the first
    int H, W;
    int xgroups;            // ceil(dim_x / VX)
    // free-space summary: one word per 256-voxel row segment (index row * nseg + segment):
    //   bit 0 = every TSDF value of the segment is exactly 1.0f
    //   bit 1 = every weight of the segment is finite and >= 0 (stays true under += 1)
    // both hold after tsdf_create/tsdf_reset; bit 0 is cleared by the first update that leaves a
    // value != 1; both are rebuilt from the arrays after tsdf_upload (recompute_flags)
    uint32_t *flags;
    int nseg;               // segments per row = dim_x / 256 (row-mapped kernels; dim_x % 256 == 0 only)
    // Linear ("flat") view of a slice, used by the kernels that serve any dim_x % 4 == 0: a slice is
    // quads_per_slice = dim_x*dim_y/4 consecutive quads, a wavefront owns 64 consecutive quads (one
    // "chunk" = 256 voxels, possibly spanning rows), the summary word of chunk c of slice z is
    // flags[z * chunks_per_slice + c].  For dim_x % 256 == 0 a chunk IS a row segment, so both
    // mappings address the same words.
    int quads_per_row, quads_per_slice, chunks_per_slice;
    // Brick view of the slab (tsdf_multiframe.hip.h, BRICK): a wavefront owns brick_q quads (4 * brick_q voxels) of
    // brick_r consecutive rows of brick_s consecutive slices, brick_q * brick_r * brick_s <= 64 lanes (lane = (slice *
    // brick_r + row) * brick_q + quad); a row group holds bricks_per_group = quads_per_row / brick_q bricks, a slice
    // group brick_groups = ceil(dim_y / brick_r) row groups; bricks are numbered group by group, a workgroup takes four
    // consecutive ones, and the launch's z index counts slice groups.  brick_q = 0: no brick view (dim_x % 4 != 0).
    // The shape is the host's choice per volume (tsdf_capi.hip, choose_brick; tsdf_set_brick_shape).
    int brick_q, brick_r, brick_s, bricks_per_group, brick_groups;
    int brick_q_magic, brick_per_magic;   // floor(2^16 / d) + 1 for d = brick_q and d = brick_q * brick_r: x / d for x < 64
    // host-proved magnitude bounds that make the shared-reciprocal projection exact (see fast_div2)
    int fast_ok;
    // 2^-20 <= trunc <= 2^20 and max_depth <= 2^59: diff / trunc may go through the shared refined reciprocal
    // (fast_div_r) on wavefronts that took the fast projection path (see there)
    int trunc_fast;
    // |camera z| below which a lane's patch counts as "near the camera plane": far above the rounding
    // error of cz over this slab (host: 1e-5 x the bound on |cz|), far above TSDF_FAST_D_MIN
    float cz_margin;
    // Depth tile summaries (tsdf_multiframe.hip.h, classify_patch): tiles per image row / column, the half-width
    // in pixels by which a projected patch is widened (host: 1.5 + the projection's error bound), and optional
    // counters {no claim, every voxel updated with dist = 1, no voxel updated} per wavefront-frame (null = off), followed by
    // the brick work list's {super-bricks skipped by every frame, entries, entries left to classify themselves, of those:
    // skipped by every frame} (tsdf_brick_list_stats).
    int tiles_w, tiles_h;
    float tile_inv;           // 1 / (pixels per tile edge): 1/16, or 1/8 for slabs large enough to repay the finer tables
    // Fine tables (tsdf_multiframe.hip.h, fine_tile_levels): tiles of kFineTile x kFineTile pixels with the nine levels (2^ky x 2^kx
    // tiles, ky, kx <= 2) a brick-sized box needs; frame f of the launch at fine + f * 9 * fine_w * fine_h.  Null = none.
    const float2 *fine = nullptr;
    int fine_w = 0, fine_h = 0;
    float px_margin_u, px_margin_v;
    unsigned int *shortcut_stats;
    float cz_short, cz_pad;   // per pose; copied into FramePose (see there)
    // one word per classifying launch: (workgroup-frames claimed as free space << 32) | workgroup-frames skipped; the
    // host reads it back asynchronously and decides whether the next launch classifies at all (null = not counted)
    unsigned long long *claim_counter;
    // One-frame launches of masked / per-object volumes: class of every workgroup's patch for this frame, decided
    // by a small kernel ahead of the launch (tsdf_multiframe.hip.h, classify_workgroups): 0 = per-voxel path,
    // 1 = every voxel updated with dist = 1, 2 = no voxel updated.  Index = linear workgroup id of the launch.
    // Null = not classified.
    const uint8_t *wg_class;
};

// Terms of the camera-frame point that do not depend on x (shared by a lane's voxels).
struct RowTerms {
    float x1, x2, y1, y2, z1, z2;
};

__device__ __forceinline__ RowTerms row_terms(const IntegrateParams &p, int gy, int gz)
{
    // ref: src/tsdf.cu:28-29,34-35 -- base-frame y,z of the row and their offsets from the pose
    float by = p.oy + (float)gy * p.vs;
    float bz = p.oz + (float)gz * p.vs;
    float dy = by - p.ty;
    float dz = bz - p.tz;
    RowTerms r;
    r.x1 = p.rx1 * dy; r.x2 = p.rx2 * dz;
    r.y1 = p.ry1 * dy; r.y2 = p.ry2 * dz;
    r.z1 = p.rz1 * dy; r.z2 = p.rz2 * dz;
    return r;
}

// Geometry + depth test of one voxel.  Returns true and the truncated distance when the
// voxel is to be updated.  Statement for statement ref: src/tsdf.cu:27-53.
template <bool MASKED>
__device__ __forceinline__ bool voxel_dist(const IntegrateParams &p, const RowTerms &r, int gx,
                                           float &dist)
{
    float bx = p.ox + (float)gx * p.vs;
    float dx = bx - p.tx;
    float pcx = p.rx0 * dx + r.x1 + r.x2;
    float pcy = p.ry0 * dx + r.y1 + r.y2;
    float pcz = p.rz0 * dx + r.z1 + r.z2;
    if (pcz <= 0.0f) return false;

    float pu = roundf(p.fx * (pcx / pcz) + p.cx);
    float pv = roundf(p.fy * (pcy / pcz) + p.cy);
    if (!(pu >= 0.0f && pu < (float)p.W && pv >= 0.0f && pv < (float)p.H)) return false;
    int pix = (int)pv * p.W + (int)pu;

    float d = p.depth[pix];
    if (MASKED) d = d * (p.mask[pix] >= 128 ? 1.0f : 0.0f);  // ref: src/Engine.cpp:192-193
    if (d <= 0.0f || d > p.max_depth) return false;

    float diff = d - pcz;
    if (diff <= -p.trunc) return false;
    dist = fminf(1.0f, diff / p.trunc);
    return true;
}

// ------------------------------------------------------------------------------------------
// integrate_scalar: the any-dim_x fallback (rows that are not 16-byte aligned: dim_x % 4 != 0).  One voxel per lane,
// block = 64 x 4 threads, grid = (ceil(dim_x / 64), ceil(dim_y / 4), nz); statement for statement the reference's update.
// ------------------------------------------------------------------------------------------
template <bool MASKED>
__global__ __launch_bounds__(256) void integrate_scalar(IntegrateParams p)
{
    const int gx = blockIdx.x * 64 + threadIdx.x;
    const int gy = blockIdx.y * 4 + threadIdx.y;
    const int lz = blockIdx.z;
    if (gx >= p.dim_x || gy >= p.dim_y) return;
    const int gz = p.z_begin + lz;  // GLOBAL z: a slab must round exactly like the whole grid
    const RowTerms r = row_terms(p, gy, gz);
    float dist;
    const bool upd = voxel_dist<MASKED>(p, r, gx, dist);
    // wavefront early-out: nothing to update in these 64 voxels -> no volume traffic at all
    if (__ballot(upd) == 0ull) return;
    if (!upd) return;
    const size_t at = ((size_t)lz * p.dim_y + gy) * (size_t)p.dim_x + (size_t)gx;
    const float w_old = p.weight[at];     // ref: src/tsdf.cu:54-57
    const float w_new = w_old + 1.0f;
    p.weight[at] = w_new;
    p.tsdf[at] = (p.tsdf[at] * w_old + dist) / w_new;
}

// ------------------------------------------------------------------------------------------
// integrate_tile<R, MASKED>: the one-frame kernel (one launch per tsdf_integrate* call).
//
// block = 64 x 4 threads; a lane owns a 4(x) x R(y) patch of one z slice, a wavefront
// 256(x) x R(y).  grid = (ceil(xgroups/64), ceil(dim_y/(4R)), nz).  Needs dim_x % 256 == 0 (rows that are not take
// the flat mapping, tsdf_multiframe.hip.h).
//
//  * the x-only products (rx0*dx, ry0*dx, rz0*dx) are computed once and shared by the R rows;
//  * geometry is branch-free (rejected voxels read pixel 0): all 4R depth samples of a lane are
//    gathered back to back;
//  * a wavefront with nothing to update neither reads nor writes the volume (__ballot early-out);
//  * exact elisions: arithmetic whose result is known exactly is skipped per wavefront --
//      - diff >= trunc  =>  fmin(1, diff/trunc) == 1: no division unless some lane is inside
//        the truncation band (correctly rounded a/b >= 1 whenever a >= b > 0);
//      - tsdf*w + dist == w + 1 (free space: tsdf 1, dist 1)  =>  the quotient is exactly 1:
//        no division unless some lane differs;
//      - a row whose TSDF values all come out bit-identical to what was loaded is not stored
//        (the weight always changes and is always stored).
//    Every skipped value is the value the full computation would produce, bit for bit.
//  * volume loads/stores carry the non-temporal hint (each byte is touched once per frame).
//  * free-space summary.  A wavefront's row is one 256-voxel segment with one flag word;
//    while the flag says "all TSDF == 1" the TSDF quad is not loaded -- the constant 1 stands
//    in for it and the same arithmetic runs on it -- so free space moves 8 B per voxel, not 12.
//    The first update that leaves a value != 1 stores the row and clears the flag; flags are
//    only ever set by fill_grid / recompute_flags (create, reset, upload).
//  * exact shared-reciprocal projection and one-instruction pixel rounding (below).
// The earlier stages of this ladder (no elision, no summary, R = 1 / 4, speculative volume loads, depth tiles staged in
// LDS) are measurement builds: tsdf_experiments.hip.h, -DTSDF_EXPERIMENTS.
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------
// Exact division, cheaper.  hipcc lowers an IEEE fp32 `n / d` to
//     s_d = div_scale(d), s_n = div_scale(n)          (power-of-two pre-scaling, extreme exponents only)
//     r0 = rcp(s_d); e0 = fma(-s_d, r0, 1); r1 = fma(e0, r0, r0)
//     q0 = s_n*r1;   e1 = fma(-s_d, q0, s_n); q1 = fma(e1, r1, q0)
//     e2 = fma(-s_d, q1, s_n); q = div_fmas(e2, r1, q1); div_fixup(q, d, n)     (11 instructions)
// For operands whose exponents are far from the ends of the range div_scale returns its input,
// div_fmas is a plain fma and div_fixup passes q through, so the seven instructions in the
// middle ARE the correctly rounded quotient.  fast_div2 runs exactly those on a pair of
// numerators that share one denominator -- the reciprocal refinement is done once, the rest is
// two-wide packed math (v_pk_mul_f32 / v_pk_fma_f32, full rate on gfx950): 3 + 5 instructions
// for both quotients instead of 22.  Bit-identity with `/` over the guarded range is checked
// on the device by tsdf_selftest_fastdiv (tests/test_gpu_fastdiv.py); the guard is in the kernel.
// ------------------------------------------------------------------------------------------
typedef float v2f __attribute__((ext_vector_type(2)));

// The refined reciprocal of the sequence above (r1), for a divisor that many quotients share.
__device__ __forceinline__ float refined_rcp(float d)
{
    const float r0 = __builtin_amdgcn_rcpf(d);
    const float e0 = __builtin_fmaf(-d, r0, 1.0f);
    return __builtin_fmaf(e0, r0, r0);
}

// n / d from d's refined reciprocal: the five instructions in the middle of the compiler's expansion.  Used for the
// truncated distance diff / trunc (ref: src/tsdf.cu:53), whose divisor is a launch constant: 5 instructions per voxel
// instead of 11, r1 computed once per kernel.  Bit-identical to `/` when div_scale is the identity, div_fixup passes
// the quotient through and no intermediate is denormal: 2^-20 <= d <= 2^20 (host: IntegrateParams::trunc_fast) and
// n = 0 or 2^-81 <= |n| <= 2^60 -- which holds for n = depth - cz on a wavefront that took the fast projection path:
// cz > cz_margin >= 2^-57 there, depth <= max_depth <= 2^59, |cz| <= 2^59 (fast_ok), and a non-zero difference of two
// floats one of which is at least 2^-57 is at least 2^-81 (Sterbenz: an exact multiple of the smaller operand's ulp
// when they are within a factor 2, at least half the larger one otherwise).  A NaN numerator (NaN depth passes the
// reference's tests) gives NaN either way, and fmin(1, NaN) = 1.  tsdf_selftest_fastdiv_band compares it with the
// compiler's division over exactly this domain on the device.
__device__ __forceinline__ float fast_div_r(float n, float d, float r1)
{
    const float q0 = n * r1;
    const float e1 = __builtin_fmaf(-d, q0, n);
    const float q1 = __builtin_fmaf(e1, r1, q0);
    const float e2 = __builtin_fmaf(-d, q1, n);
    return __builtin_fmaf(e2, r1, q1);
}

__device__ __forceinline__ v2f fast_div2(v2f n, float d)
{
    const float r1 = refined_rcp(d);
    const v2f R = {r1, r1}, D = {d, d};
    const v2f q0 = n * R;
    const v2f e1 = __builtin_elementwise_fma(-D, q0, n);
    const v2f q1 = __builtin_elementwise_fma(e1, R, q0);
    const v2f e2 = __builtin_elementwise_fma(-D, q1, n);
    return __builtin_elementwise_fma(e2, R, q1);
}

// roundf(u) for u > -0.5, as an integer, in ONE instruction: v_cvt_rpi_i32_f32 computes
// (int)floor(u + 0.5) with a single rounding (round to nearest, ties toward +infinity) -- for
// u > -0.5 that is exactly C's round-half-away-from-zero (ref: src/tsdf.cu:41-42 `roundf`), for
// huge u it saturates (and is then rejected by the range test), and u <= -0.5 / NaN are rejected
// by the `u > -0.5` test before the value is used.  It replaces trunc, subtract, compare, select,
// add and convert (6 instructions per coordinate).  There is no builtin, hence the one-line asm;
// tsdf_selftest_round checks it against roundf for EVERY float in (-0.5, 2^24] on the device.
__device__ __forceinline__ int round_half_up_i32(float u)
{
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(u));
    return r;
}

// Denominators the fast path accepts; numerators are bounded by the host (IntegrateParams::fast_ok).
#define TSDF_FAST_D_MIN 8.6736174e-19f   /* 2^-60 */

typedef float v4f __attribute__((ext_vector_type(4)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ float4 vol_load(const float *p)
{
    if constexpr (NT) {
        v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
        return make_float4(v.x, v.y, v.z, v.w);
    } else {
        return *reinterpret_cast<const float4 *>(p);
    }
}
template <bool NT>
__device__ __forceinline__ void vol_store(float *p, float4 v)
{
    if constexpr (NT) {
        v4f q = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(q, reinterpret_cast<v4f *>(p));
    } else {
        *reinterpret_cast<float4 *>(p) = v;
    }
}

// depth[px] through a 32-bit byte offset from the wave-uniform base: the load takes the base from SGPRs
// and the offset from one VGPR (global_load_dword v, v_off, s[base]) instead of a 64-bit address built
// per voxel.  px < 2^30 pixels (tsdf_create refuses larger images), so px * 4 does not wrap.
__device__ __forceinline__ float gather_f32(const float *__restrict__ base, uint32_t px)
{
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + (px << 2));
}

// Pixel index iv * W + iu on the fast path: both factors are below 2^24 there (host: fast_ok), so the
// full-rate 24-bit multiply-add gives the same integer as the quarter-rate 32-bit one.
__device__ __forceinline__ int pixel_index24(int iv, int W, int iu)
{
    return (int)(__umul24((unsigned)iv, (unsigned)W) + (unsigned)iu);
}

// MASKED: 0 = plain depth, 1 = depth * (mask/255) (p.mask must be set).
template <int R, int MASKED>
__device__ __forceinline__ void integrate_tile_body(const IntegrateParams &p, const int bx, const int by, const int lz)
{
    const int gz = p.z_begin + lz;
    const int xg = bx * 64 + threadIdx.x;
    const int gy0 = (by * 4 + threadIdx.y) * R;
    if (xg >= p.xgroups || gy0 >= p.dim_y) return;
    const size_t row0 = ((size_t)lz * p.dim_y + gy0) * (size_t)p.dim_x + (size_t)xg * 4;
    const size_t flag0 = ((size_t)lz * p.dim_y + gy0) * (size_t)p.nseg + bx;

    // ---- phase 0: summary flags ------------------------------------------------------------------
    uint32_t fl[R];
    float4 t4[R], w4[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        fl[r] = gy0 + r < p.dim_y ? p.flags[flag0 + (size_t)r * p.nseg] : 0u;
        t4[r] = make_float4(1.f, 1.f, 1.f, 1.f);
        w4[r] = make_float4(0.f, 0.f, 0.f, 0.f);
    }

    // x-only and z-only terms (ref: src/tsdf.cu:27,29,33,35-38)
    float ax[4], ay[4], az[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float bx = p.ox + (float)(xg * 4 + j) * p.vs;
        float dx = bx - p.tx;
        ax[j] = p.rx0 * dx; ay[j] = p.ry0 * dx; az[j] = p.rz0 * dx;
    }
    const float bz = p.oz + (float)gz * p.vs;
    const float dz = bz - p.tz;
    const float x2 = p.rx2 * dz, y2 = p.ry2 * dz, z2 = p.rz2 * dz;

    // ---- phase 1: geometry of all 4R voxels, depth gathers issued back to back --------------
    float pcz[R][4], dval[R][4];
    bool geo[R][4];
    int pixel[R][4];
    // camera-frame z of every voxel first: it decides which projection path the wavefront takes
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float dy = (p.oy + (float)(gy0 + r) * p.vs) - p.ty;   // ref: src/tsdf.cu:28,34
        const float z1 = p.rz1 * dy;
#pragma unroll
        for (int j = 0; j < 4; ++j) pcz[r][j] = az[j] + z1 + z2;
    }
    // The fast projection needs every cz of the wavefront outside (0, TSDF_FAST_D_MIN).  cz is affine
    // over a lane's 4 x R patch, so its extremes sit at the patch corners (up to rounding, which the
    // margin dwarfs): all corners > margin, or all < -margin (those voxels are rejected whatever the
    // quotient), proves it with 6 instructions instead of two compares per voxel.
    const float cmin = fminf(fminf(pcz[0][0], pcz[0][3]), fminf(pcz[R - 1][0], pcz[R - 1][3]));
    const float cmax = fmaxf(fmaxf(pcz[0][0], pcz[0][3]), fmaxf(pcz[R - 1][0], pcz[R - 1][3]));
    const bool unsafe = !(cmin > p.cz_margin) & !(cmax < -p.cz_margin);
    const bool fast = p.fast_ok != 0 && __ballot(unsafe) == 0ull;   // wave-uniform
    if (fast) {
        // Same values as the generic branch below, obtained with fewer instructions:
        //  - both quotients of a voxel from one refined reciprocal (fast_div2), packed;
        //  - (cx, cy) sums, fx*q + cx / fy*q + cy as two-wide packed operations;
        //  - roundf(u) as one v_cvt_rpi_i32_f32 (round_half_up_i32 above); u <= -0.5 (and NaN) can
        //    only round to a negative pixel, which ref: src/tsdf.cu:43 rejects -- so the lower bound
        //    is tested on u itself, the upper one on the integer (unsigned compare).
        const v2f F = {p.fx, p.fy}, C = {p.cx, p.cy};
        const bool front = cmin > p.cz_margin;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const bool row_ok = gy0 + r < p.dim_y;
            const float dy = (p.oy + (float)(gy0 + r) * p.vs) - p.ty;
            const v2f XY1 = {p.rx1 * dy, p.ry1 * dy};
            const v2f XY2 = {x2, y2};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const v2f A = {ax[j], ay[j]};
                const v2f n = A + XY1 + XY2;                       // (pt_cam_x, pt_cam_y), ref: :36-37
                const float cz = pcz[r][j];
                const v2f uv = F * fast_div2(n, cz) + C;           // ref: :41-42 before rounding
                const int iu = round_half_up_i32(uv.x), iv = round_half_up_i32(uv.y);
                // cz > 0 for the whole patch or for none of it (corner test above): `front`
                const bool ok = row_ok & front & (uv.x > -0.5f) & (uv.y > -0.5f) &
                                ((unsigned)iu < (unsigned)p.W) & ((unsigned)iv < (unsigned)p.H);
                geo[r][j] = ok;
                pixel[r][j] = ok ? pixel_index24(iv, p.W, iu) : 0;
            }
        }
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int gy = gy0 + r;
            const bool row_ok = gy < p.dim_y;
            const float by = p.oy + (float)gy * p.vs;   // ref: src/tsdf.cu:28
            const float dy = by - p.ty;
            const float x1 = p.rx1 * dy, y1 = p.ry1 * dy;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float cx = ax[j] + x1 + x2;
                const float cy = ay[j] + y1 + y2;
                const float cz = pcz[r][j];
                // ref: src/tsdf.cu:39-43.  cz <= 0 is tested before the quotient is used, exactly
                // as the reference's `continue`; the division itself is harmless for any cz.
                const float pu = roundf(p.fx * (cx / cz) + p.cx);
                const float pv = roundf(p.fy * (cy / cz) + p.cy);
                const bool ok = row_ok && !(cz <= 0.0f) && pu >= 0.0f && pu < (float)p.W && pv >= 0.0f &&
                                pv < (float)p.H;
                geo[r][j] = ok;
                pixel[r][j] = ok ? (int)pv * p.W + (int)pu : 0;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // unsigned 32-bit offset from a wave-uniform base: the load takes the base from SGPRs
            const uint32_t px = (uint32_t)pixel[r][j];
            float d = gather_f32(p.depth, px);
            if (MASKED == 1) d = d * (p.mask[px] >= 128 ? 1.0f : 0.0f);  // ref: src/Engine.cpp:192-193
            dval[r][j] = d;
        }
    }

    // ---- phase 2: depth tests (ref: src/tsdf.cu:46-49) ------------------------------------
    float diff[R][4];
    bool upd[R][4], rowany[R], bandr[R];
    bool any = false, band = false;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        rowany[r] = false;
        bandr[r] = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = dval[r][j];
            const float df = d - pcz[r][j];
            diff[r][j] = df;
            // bitwise, not short-circuit: four compares and mask logic, no exec-mask regions
            const bool u = geo[r][j] & !((d <= 0.0f) | (d > p.max_depth)) & !(df <= -p.trunc);
            upd[r][j] = u;
            rowany[r] |= u;
            bandr[r] |= u & !(df >= p.trunc);
        }
        band |= bandr[r];
        any |= rowany[r];
    }
    if (__ballot(any) == 0ull) return;  // wavefront early-out: nothing is read or written
    if (!any) return;

    // ---- phase 3: the volume quads; truncated distance -----------------------------------------
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (rowany[r]) {
            w4[r] = vol_load<true>(p.weight + row0 + (size_t)r * p.dim_x);
            if (!(fl[r] & 1u)) t4[r] = vol_load<true>(p.tsdf + row0 + (size_t)r * p.dim_x);
        }
    }
    float dist[R][4];
    if (__ballot(band) != 0ull) {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) dist[r][j] = fminf(1.0f, diff[r][j] / p.trunc);  // ref: :53
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r)
#pragma unroll
            for (int j = 0; j < 4; ++j) dist[r][j] = 1.0f;
    }

    // ---- phase 4: running weighted mean (ref: src/tsdf.cu:54-57), stores ----------------------
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (fl[r] == 3u && __ballot(bandr[r]) == 0ull) {
            // Free space, wave-uniform: every TSDF value of the segment is 1, every weight is finite
            // and >= 0, and every updated lane has dist == 1.  Then num = fl(1*w + 1) = fl(w + 1) = wn,
            // the quotient is exactly 1, the TSDF row is unchanged: only the weights move.
            if (rowany[r]) {
                const float4 w = w4[r];
                vol_store<true>(p.weight + row0 + (size_t)r * p.dim_x,
                                make_float4(upd[r][0] ? w.x + 1.0f : w.x, upd[r][1] ? w.y + 1.0f : w.y,
                                            upd[r][2] ? w.z + 1.0f : w.z, upd[r][3] ? w.w + 1.0f : w.w));
            }
            continue;
        }
        float tv[4] = {t4[r].x, t4[r].y, t4[r].z, t4[r].w};
        float wv[4] = {w4[r].x, w4[r].y, w4[r].z, w4[r].w};
        float num[4], wn[4];
        bool need = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            wn[j] = wv[j] + 1.0f;
            num[j] = tv[j] * wv[j] + dist[r][j];
            // x / x == 1 exactly for finite non-zero x (wn >= 1 whenever the weights are counts)
            need |= upd[r][j] && !(num[j] == wn[j] && wn[j] < 3.0e38f && wn[j] > 0.0f);
        }
        float nt[4];
        if (__ballot(rowany[r] && need) != 0ull) {
#pragma unroll
            for (int j = 0; j < 4; ++j) nt[j] = num[j] / wn[j];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) nt[j] = 1.0f;
        }
        bool changed = false, notone = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float newt = upd[r][j] ? nt[j] : tv[j];
            changed |= __float_as_uint(newt) != __float_as_uint(tv[j]);
            notone |= upd[r][j] && __float_as_uint(newt) != 0x3f800000u;
            tv[j] = newt;
            wv[j] = upd[r][j] ? wn[j] : wv[j];
        }
        if ((fl[r] & 1u) && __ballot(notone) != 0ull) {
            if (notone) p.flags[flag0 + (size_t)r * p.nseg] = fl[r] & 2u;  // segment no longer all ones
        }
        const bool store_t = __ballot(rowany[r] && changed) != 0ull;
        if (rowany[r]) {
            if (store_t) vol_store<true>(p.tsdf + row0 + (size_t)r * p.dim_x, make_float4(tv[0], tv[1], tv[2], tv[3]));
            vol_store<true>(p.weight + row0 + (size_t)r * p.dim_x, make_float4(wv[0], wv[1], wv[2], wv[3]));
        }
    }
}

template <int R, bool MASKED>
__global__ __launch_bounds__(256) void integrate_tile(IntegrateParams p)
{
    integrate_tile_body<R, MASKED ? 1 : 0>(p, blockIdx.x, blockIdx.y, blockIdx.z);
}

// Device self-test of fast_div2 against the compiler's IEEE division: pseudo-random operands from a
// counter hash, denominators in [2^-60, 2^60], numerators up to 2^60 in magnitude (plus exact
// zeros and structured mantissas).  A sample passes when the quotients are bit-identical, or --
// for quotients below 2^-42, where the unscaled sequence may lose the last bit of a denormal
// residual -- when the pixel the kernel derives from it (rounding of fl(f*q + c) and the u > -0.5 test)
// is identical.
__device__ __forceinline__ uint32_t hash32(uint64_t x)
{
    x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
    return (uint32_t)x;
}

__global__ __launch_bounds__(256) void selftest_fastdiv(uint64_t seed, uint64_t n_samples, float fx, float cx,
                                                        unsigned long long *mismatch, float *first_bad)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_samples; i += stride) {
        const uint32_t h0 = hash32(seed + 3 * i), h1 = hash32(seed + 3 * i + 1), h2 = hash32(seed + 3 * i + 2);
        const uint32_t mode = h2 & 7u;
        // denominator: exponent in [-60, 60], mantissa random / all ones / all zeros
        uint32_t dm = h0 & 0x7fffffu;
        if (mode == 1) dm = 0x7fffffu; else if (mode == 2) dm = 0u; else if (mode == 3) dm &= 0x7u;
        const uint32_t de = 127u - 60u + (h0 >> 23) % 121u;
        const float d = __uint_as_float((de << 23) | dm);
        // numerators: sign, exponent in [-149, 60] for one, near the denominator for the other
        uint32_t ne = (mode >= 6) ? ((h1 >> 23) % 190u) : (de - 12u + (h1 >> 23) % 24u);
        if (ne > 127u + 60u) ne = 127u + 60u;
        uint32_t nm = h1 & 0x7fffffu;
        if (mode == 4) nm = 0x7fffffu; else if (mode == 5) nm = dm;
        float n0 = __uint_as_float(((h2 >> 3) & 1u) << 31 | (ne << 23) | nm);
        float n1 = __uint_as_float(((h2 >> 4) & 1u) << 31 | (((ne + (h2 >> 8) % 5u) % 188u) << 23) | (h2 >> 9));
        if ((h2 >> 5 & 31u) == 0u) n0 = 0.0f;
        const v2f nn = {n0, n1};
        const v2f q = fast_div2(nn, d);
        const float r0 = n0 / d, r1 = n1 / d;
        bool bad0 = __float_as_uint(q.x) != __float_as_uint(r0);
        bool bad1 = __float_as_uint(q.y) != __float_as_uint(r1);
        // what the kernel derives from a quotient: the pixel (round_half_up_i32) and the `u > -0.5` test
        auto same_pixel = [&](float qa, float qb) {
            const float ua = fx * qa + cx, ub = fx * qb + cx;
            return round_half_up_i32(ua) == round_half_up_i32(ub) && (ua > -0.5f) == (ub > -0.5f);
        };
        if (bad0 && fabsf(r0) < 2.2737368e-13f) bad0 = !same_pixel(q.x, r0);
        if (bad1 && fabsf(r1) < 2.2737368e-13f) bad1 = !same_pixel(q.y, r1);
        if (bad0 || bad1) {
            if (atomicAdd(mismatch, 1ull) == 0ull) {
                first_bad[0] = bad0 ? n0 : n1; first_bad[1] = d; first_bad[2] = bad0 ? q.x : q.y;
                first_bad[3] = bad0 ? r0 : r1;
            }
        }
    }
}

// Device self-test of fast_div_r over the domain the truncated distance uses it on: divisors in [2^-20, 2^20],
// numerators 0 or of magnitude in [2^-81, 2^60] (either sign, structured and random mantissas), NaN numerators.
// Every quotient must be bit-identical to the compiler's IEEE division (NaN: both NaN).
__global__ __launch_bounds__(256) void selftest_fastdiv_band(uint64_t seed, uint64_t n_samples, unsigned long long *mismatch,
                                                             float *first_bad)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_samples; i += stride) {
        const uint32_t h0 = hash32(seed + 3 * i), h1 = hash32(seed + 3 * i + 1), h2 = hash32(seed + 3 * i + 2);
        const uint32_t mode = h2 & 7u;
        uint32_t dm = h0 & 0x7fffffu;
        if (mode == 1) dm = 0x7fffffu; else if (mode == 2) dm = 0u; else if (mode == 3) dm &= 0x7u;
        const uint32_t de = 127u - 20u + (h0 >> 23) % 41u;                      // 2^-20 .. 2^20
        const float d = __uint_as_float((de << 23) | dm);
        uint32_t ne = 127u - 81u + (h1 >> 23) % 142u;                           // 2^-81 .. 2^60
        if (mode >= 6) ne = de - 2u + (h1 >> 23) % 5u;                          // quotients around 1 (the clamp's edge)
        if (ne > 127u + 60u) ne = 127u + 60u;
        uint32_t nm = h1 & 0x7fffffu;
        if (mode == 4) nm = 0x7fffffu; else if (mode == 5) nm = dm;
        if (ne == 127u + 60u) nm = 0u;                                          // |n| <= 2^60
        float n = __uint_as_float((((h2 >> 3) & 1u) << 31) | (ne << 23) | nm);
        if (((h2 >> 5) & 63u) == 0u) n = 0.0f;
        if (((h2 >> 5) & 63u) == 1u) n = __uint_as_float(0x7fc00000u | (h2 >> 12));   // NaN
        const float q = fast_div_r(n, d, refined_rcp(d));
        const float r = n / d;
        const bool same = __float_as_uint(q) == __float_as_uint(r) || (q != q && r != r);
        if (!same) {
            if (atomicAdd(mismatch, 1ull) == 0ull) { first_bad[0] = n; first_bad[1] = d; first_bad[2] = q; first_bad[3] = r; }
        }
    }
}

// Exhaustive device check of round_half_up_i32 against roundf: every fp32 bit pattern with
// -0.5 < u <= 2^24 (all of them, not a sample) must convert to the same integer.
__global__ __launch_bounds__(256) void selftest_round(unsigned long long *mismatch, float *first_bad)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    // positive patterns 0 .. 0x4B800000 (2^24), then negative ones 0x80000000 .. 0xBEFFFFFF (-0 .. just above -0.5)
    const uint64_t n_pos = 0x4B800001ull, n_neg = 0x3F000000ull;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_pos + n_neg; i += stride) {
        const uint32_t bits = i < n_pos ? (uint32_t)i : (0x80000000u + (uint32_t)(i - n_pos));
        const float u = __uint_as_float(bits);
        if (!(u > -0.5f)) continue;
        const int want = (int)roundf(u);
        const int got = round_half_up_i32(u);
        if (got != want) {
            if (atomicAdd(mismatch, 1ull) == 0ull) { first_bad[0] = u; first_bad[1] = (float)got; first_bad[2] = (float)want; first_bad[3] = 0.f; }
        }
    }
}

// Ceiling probe: the same 16 B/voxel read-modify-write stream with no geometry at all
// (tsdf *= 1, weight += 0 keeps the grid intact).  What this reaches is what the memory
// system gives this access pattern; Integrate is judged against it and against the 8 TB/s spec.
template <bool NT>
__global__ __launch_bounds__(256) void stream_rmw(float *tsdf, float *weight, size_t n_quads, float one,
                                                  float zero)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < n_quads; q += stride) {
        float4 t = vol_load<NT>(tsdf + 4 * q);
        float4 w = vol_load<NT>(weight + 4 * q);
        t.x *= one; t.y *= one; t.z *= one; t.w *= one;
        w.x += zero; w.y += zero; w.z += zero; w.w += zero;
        vol_store<NT>(tsdf + 4 * q, t);
        vol_store<NT>(weight + 4 * q, w);
    }
}

// Rebuild the free-space summary from the arrays (after tsdf_upload or an external write): one
// wavefront per 256-voxel chunk of a slice.  block = 64 x 4, grid = (ceil(chunks_per_slice/4), 1, nz).
__global__ __launch_bounds__(256) void recompute_flags(const float *tsdf, const float *weight, uint32_t *flags,
                                                       int quads_per_slice, int chunks_per_slice)
{
    const int chunk = blockIdx.x * 4 + threadIdx.y;
    if (chunk >= chunks_per_slice) return;
    const int q = chunk * 64 + threadIdx.x;
    const size_t base = ((size_t)blockIdx.z * quads_per_slice + q) * 4;
    bool ones = true, sane = true;
    if (q < quads_per_slice) {
        const float4 t = *reinterpret_cast<const float4 *>(tsdf + base);
        const float4 w = *reinterpret_cast<const float4 *>(weight + base);
        ones = __float_as_uint(t.x) == 0x3f800000u && __float_as_uint(t.y) == 0x3f800000u &&
               __float_as_uint(t.z) == 0x3f800000u && __float_as_uint(t.w) == 0x3f800000u;
        sane = w.x >= 0.0f && w.x < 3.0e38f && w.y >= 0.0f && w.y < 3.0e38f && w.z >= 0.0f && w.z < 3.0e38f &&
               w.w >= 0.0f && w.w < 3.0e38f;
    }
    const uint32_t f = (__ballot(!ones) == 0ull ? 1u : 0u) | (__ballot(!sane) == 0ull ? 2u : 0u);
    if (threadIdx.x == 0) flags[(size_t)blockIdx.z * chunks_per_slice + chunk] = f;
}

// Raw 16-bit depth -> metres on the device: out = raw * scale where (row % row_step == 0 and
// col % col_step == 0), 0 elsewhere.  scale = 1/DepthMapFactor in fp32 (5000 for TUM,
// ref: config/TUM3.yaml:34); steps (4, 3) reproduce the offline labeller's subsampling
// (ref: examples/label_instance_rgbd.cpp:89-100), steps (1, 1) keep every pixel.  Halves the
// per-frame host->device copy (614 KB instead of 1.2 MB at 640x480).
__global__ __launch_bounds__(256) void depth_u16_to_f32(const uint16_t *raw, float *out, int H, int W,
                                                        float scale, int row_step, int col_step)
{
    const int n = H * W;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int r = i / W, c = i - r * W;
        const bool keep = (r % row_step == 0) && (c % col_step == 0);
        out[i] = (keep ? (float)raw[i] : 0.0f) * scale;
    }
}

// Grid origin of a new object volume: per-axis minimum, over the pixels with (masked) depth > 0, of the
// back-projected point x = (c - cx) * z * (1/fx), y = (r - cy) * z * (1/fy), z -- starting from 1000
// (ref: Object::Object, src/Object.cpp:37-49; the mask is the instance mask of src/Engine.cpp:192-193).
// A minimum does not depend on the order of evaluation, so the device result equals the host loop's bits.
// out[3] must be initialised to 1000.0f; floats with a clear sign bit order like their bit patterns, those with
// the sign bit set in reverse, hence the two atomics -- chosen by the sign BIT, so that -0.0f (pattern INT_MIN)
// does not take the signed-integer branch and displace a genuinely negative minimum.
// NaN samples: the reference's running std::min lets a NaN replace the minimum and the next valid pixel replace
// the NaN, i.e. its result depends on the raster order; this reduction ignores NaN samples (fminf drops them).
__device__ __forceinline__ void atomic_min_float(float *addr, float v)
{
    if (__float_as_int(v) >= 0) atomicMin(reinterpret_cast<int *>(addr), __float_as_int(v));
    else atomicMax(reinterpret_cast<unsigned int *>(addr), __float_as_uint(v));
}

__global__ __launch_bounds__(256) void object_origin(const float *depth, const uint8_t *mask, int H, int W, float fx,
                                                     float fy, float cx, float cy, float *out)
{
    float mx = 1000.0f, my = 1000.0f, mz = 1000.0f;
    const float ifx = 1.0f / fx, ify = 1.0f / fy;
    const int n = H * W;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float z = depth[i];
        if (mask != nullptr) z = z * (mask[i] >= 128 ? 1.0f : 0.0f);
        if (z <= 0.0f) continue;
        const int r = i / W, c = i - r * W;
        const float x = ((float)c - cx) * z * ifx;
        const float y = ((float)r - cy) * z * ify;
        mx = fminf(mx, x); my = fminf(my, y); mz = fminf(mz, z);
    }
    for (int off = 32; off > 0; off >>= 1) {
        mx = fminf(mx, __shfl_down(mx, off)); my = fminf(my, __shfl_down(my, off)); mz = fminf(mz, __shfl_down(mz, off));
    }
    if ((threadIdx.x & 63) == 0) { atomic_min_float(out + 0, mx); atomic_min_float(out + 1, my); atomic_min_float(out + 2, mz); }
}

// TSDF = 1, weight = 0 (ref: src/tsdf.cu:79-81) written at bandwidth on the device.
__global__ __launch_bounds__(256) void fill_grid(float *tsdf, float *weight, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x * 4;
    for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 4 <= n) {
            *reinterpret_cast<float4 *>(tsdf + i) = make_float4(1.f, 1.f, 1.f, 1.f);
            *reinterpret_cast<float4 *>(weight + i) = make_float4(0.f, 0.f, 0.f, 0.f);
        } else {
            for (size_t k = i; k < n; ++k) { tsdf[k] = 1.f; weight[k] = 0.f; }
        }
    }
}

}  // namespace tsdfk
