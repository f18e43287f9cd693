// tsdf_experiments.hip.h -- measurement builds only (-DTSDF_EXPERIMENTS, `make experiments` -> libtsdf_hip_exp.so).
//
// Nothing in here ships in libtsdf_hip.so.  These are the earlier stages and the rejected alternatives DESIGN.md section 4
// quotes numbers for, kept compilable so that the A/B runs can be repeated on one box (TSDF_HIP_LIB=.../libtsdf_hip_exp.so,
// tsdf_set_kernel_variant; tools/sweep.py, tools/probe_sband.py, tools/ab_*.sh):
//   ladder_rows<4>     the first version: one row per wavefront, no elision (variant 2)
//   ladder_tile<...>   the one-frame kernel with every switch of its ladder exposed: R = 1 / 2 / 4, exact elisions, nt hints,
//                      free-space summary, speculative frustum-gated volume loads (EARLY), exact shared-reciprocal projection
//                      (FAST), the workgroup's depth pixels staged in LDS (LDSD: the north-star sketch, measured 1.6x slower),
//                      a class byte per workgroup (CLS) (variants 16 .. 119)
//   integrate_multi / integrate_multi_xcd   fused launches with the frame blocks staged in device memory, R = 2, an XCD-aware
//                      workgroup order (variants 4, 5, 6)
//   integrate_multi_wg the classified fused launch of rounds 1-2: rows / 1024 consecutive voxels classified per workgroup
//                      (variant 11), bricks classified per wavefront in the workgroup's prologue, with and without the
//                      super-brick table of classify_superbricks (variants 13, 12) -- superseded by the brick work list
//   classify_workgroups[_batched], integrate_multi_single<CLS>, integrate_multi_batched<CLS>   one class per workgroup of a
//                      masked one-frame launch (variant 11) -- superseded by classify_bricks
// Every one of them is bit-exact (tests/test_gpu_experiments.py runs them when the experiments build is the loaded library).
#pragma once
#ifndef TSDF_EXPERIMENTS
#error "tsdf_experiments.hip.h is for -DTSDF_EXPERIMENTS builds"
#endif
#include "tsdf_kernels.hip.h"
#include "tsdf_multiframe.hip.h"

namespace tsdfk {

// ------------------------------------------------------------------------------------------
// ladder_rows<VX>: block = 64 x 4 threads; a wavefront = 64 lanes x VX voxels of one row.
// grid = (ceil(xgroups/64), ceil(dim_y/4), nz).
// VX = 4 needs dim_x % 4 == 0 (rows stay 16-byte aligned); VX = 1 takes any dim_x.
// ------------------------------------------------------------------------------------------
template <int VX, bool MASKED>
__global__ __launch_bounds__(256) void ladder_rows(IntegrateParams p)
{
    const int xg = blockIdx.x * 64 + threadIdx.x;
    const int gy = blockIdx.y * 4 + threadIdx.y;
    const int lz = blockIdx.z;
    if (xg >= p.xgroups || gy >= p.dim_y) return;
    const int gz = p.z_begin + lz;  // GLOBAL z: a slab must round exactly like the whole grid

    const RowTerms r = row_terms(p, gy, gz);

    float dist[VX];
    bool upd[VX];
    bool any = false;
#pragma unroll
    for (int j = 0; j < VX; ++j) {
        upd[j] = voxel_dist<MASKED>(p, r, xg * VX + j, dist[j]);
        any |= upd[j];
    }
    // wavefront early-out: nothing to update in these 64*VX voxels -> no volume traffic at all
    if (__ballot(any) == 0ull) return;
    if (!any) return;

    const size_t row = ((size_t)lz * p.dim_y + gy) * (size_t)p.dim_x + (size_t)xg * VX;
    if constexpr (VX == 4) {
        float4 t = *reinterpret_cast<const float4 *>(p.tsdf + row);
        float4 w = *reinterpret_cast<const float4 *>(p.weight + row);
        float tv[4] = {t.x, t.y, t.z, t.w};
        float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (upd[j]) {  // ref: src/tsdf.cu:54-57
                float w_new = wv[j] + 1.0f;
                tv[j] = (tv[j] * wv[j] + dist[j]) / w_new;
                wv[j] = w_new;
            }
        }
        *reinterpret_cast<float4 *>(p.tsdf + row) = make_float4(tv[0], tv[1], tv[2], tv[3]);
        *reinterpret_cast<float4 *>(p.weight + row) = make_float4(wv[0], wv[1], wv[2], wv[3]);
    } else {
#pragma unroll
        for (int j = 0; j < VX; ++j) {
            if (upd[j]) {
                float w_old = p.weight[row + j];
                float w_new = w_old + 1.0f;
                p.weight[row + j] = w_new;
                p.tsdf[row + j] = (p.tsdf[row + j] * w_old + dist[j]) / w_new;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// ladder_tile<R, ELIDE, NT, MASKED, SUM, EARLY, FAST, LDSD, CLS>: every stage of the one-frame kernel's ladder.
//
// block = 64 x 4 threads; a lane owns a 4(x) x R(y) patch of one z slice, a wavefront
// 256(x) x R(y).  grid = (ceil(xgroups/64), ceil(dim_y/(4R)), nz).  Needs dim_x % 4 == 0.
//
//  * the x-only products (rx0*dx, ry0*dx, rz0*dx) are computed once and shared by the R rows;
//  * geometry is branch-free (rejected voxels read pixel 0): all 4R depth samples of a lane are
//    gathered back to back;
//  * a wavefront with nothing to update never writes, and (without EARLY, or when the coarse
//    frustum test rejects its patch) never reads the volume either (__ballot early-out);
//  * ELIDE: arithmetic whose result is known exactly is skipped per wavefront --
//      - diff >= trunc  =>  fmin(1, diff/trunc) == 1: no division unless some lane is inside
//        the truncation band (correctly rounded a/b >= 1 whenever a >= b > 0);
//      - tsdf*w + dist == w + 1 (free space: tsdf 1, dist 1)  =>  the quotient is exactly 1:
//        no division unless some lane differs;
//      - a row whose TSDF values all come out bit-identical to what was loaded is not stored
//        (the weight always changes and is always stored).
//    Every skipped value is the value the full computation would produce, bit for bit.
//  * NT: volume loads/stores carry the non-temporal hint (each byte is touched once per frame).
//  * SUM: free-space summary.  A wavefront's row is one 256-voxel segment with one flag word;
//    while the flag says "all TSDF == 1" the TSDF quad is not loaded -- the constant 1 stands
//    in for it and the same arithmetic runs on it -- so free space moves 8 B per voxel, not 12.
//    The first update that leaves a value != 1 stores the row and clears the flag; flags are
//    only ever set by fill_grid / recompute_flags (create, reset, upload).
//  * EARLY: the kernel was limited by bytes in flight, not by bandwidth or VALU: a wavefront
//    issued its volume loads only after ~1000 cycles of geometry plus a depth-gather round
//    trip.  With EARLY the flag words and the weight quads (and, once the flags are back, the
//    TSDF quads of rows that are not all-ones) are requested at the top of the kernel and
//    arrive while the geometry runs.  The loads are speculative -- a patch may turn out to need
//    nothing -- so they are gated by a coarse, wave-uniform test of the patch's four corners
//    against the image; the gate only decides WHEN a quad is loaded, never what is computed
//    (a lane that has to update a quad that was not pre-loaded loads it then), so it needs no
//    rounding analysis.
// Coarse frustum gate for the speculative loads: true unless the patch's four corners
// (x0|x1, y0|y1 at slice gz) are all behind the camera or all beyond the same image edge by
// more than one pixel.  Lanes 0..3 each project one corner; approximate arithmetic is fine.
__device__ __forceinline__ bool patch_may_be_visible(const IntegrateParams &p, int x0, int x1, int y0,
                                                     int y1, int gz)
{
    const int lane = threadIdx.x & 63;
    const float bx = p.ox + (float)((lane & 1) ? x1 : x0) * p.vs - p.tx;
    const float by = p.oy + (float)((lane & 2) ? y1 : y0) * p.vs - p.ty;
    const float bz = p.oz + (float)gz * p.vs - p.tz;
    const float cx = p.rx0 * bx + p.rx1 * by + p.rx2 * bz;
    const float cy = p.ry0 * bx + p.ry1 * by + p.ry2 * bz;
    const float cz = p.rz0 * bx + p.rz1 * by + p.rz2 * bz;
    const float inv = __builtin_amdgcn_rcpf(cz);
    const float u = p.fx * (cx * inv) + p.cx;
    const float v = p.fy * (cy * inv) + p.cy;
    const bool corner = lane < 4;
    const bool front = cz > 0.0f;
    const unsigned long long m = __ballot(corner);
    const bool all_front = (__ballot(corner && front) == m);
    if (__ballot(corner && !front) == m) return false;          // wholly behind the camera
    if (!all_front) return true;                                 // straddles the camera plane: no claim
    if (__ballot(corner && u < -1.0f) == m) return false;
    if (__ballot(corner && u > (float)p.W) == m) return false;
    if (__ballot(corner && v < -1.0f) == m) return false;
    if (__ballot(corner && v > (float)p.H) == m) return false;
    return true;
}

// MASKED: 0 = plain depth, 1 = depth * (mask/255) (p.mask must be set), 2 = decided per launch
// parameter block (p.mask may be null) -- the batched kernel, where each object brings its own.
// LDSD: stage the depth pixels the workgroup's voxel patch projects onto in LDS and sample from there
// (the north-star sketch).  Kept as a measured experiment: the frame lives in every XCD's L2 and the
// kernel is VALU-issue-bound, so the extra bounding-box / index arithmetic costs more than the L1/L2
// gathers it replaces.
constexpr int kLdsTile = 4096;   // floats: 16 KiB per workgroup, 8+ workgroups per CU still fit

template <int R, bool ELIDE, bool NT, int MASKED, bool SUM, bool EARLY, bool FAST, bool LDSD = false>
__device__ __forceinline__ void ladder_tile_body(const IntegrateParams &p, const int bx, const int by, const int lz)
{
    const int gz = p.z_begin + lz;
    // ---- (LDSD) depth tile of the workgroup's 256 x 4R voxel patch -------------------------------
    __shared__ float lds_depth[LDSD ? kLdsTile : 1];
    int tu0 = 0, tv0 = 0, tw = 0, th = 0;   // tile origin and size in pixels (workgroup-uniform); tw = 0: no tile
    if constexpr (LDSD) {
        const int lane = threadIdx.x & 63;
        const int x0 = bx * 256, x1 = min(x0 + 255, p.dim_x - 1);
        const int y0 = by * 4 * R, y1 = min(y0 + 4 * R - 1, p.dim_y - 1);
        // lanes 0..3 project the four corners (approximate arithmetic: the tile only has to CONTAIN the
        // exact pixels; a voxel whose exact pixel falls outside it reads global memory instead)
        const float qx = p.ox + (float)((lane & 1) ? x1 : x0) * p.vs - p.tx;
        const float qy = p.oy + (float)((lane & 2) ? y1 : y0) * p.vs - p.ty;
        const float qz = p.oz + (float)gz * p.vs - p.tz;
        const float ccx = p.rx0 * qx + p.rx1 * qy + p.rx2 * qz;
        const float ccy = p.ry0 * qx + p.ry1 * qy + p.ry2 * qz;
        const float ccz = p.rz0 * qx + p.rz1 * qy + p.rz2 * qz;
        const float inv = __builtin_amdgcn_rcpf(ccz);
        float u = p.fx * (ccx * inv) + p.cx, v = p.fy * (ccy * inv) + p.cy;
        float umin = u, umax = u, vmin = v, vmax = v, zmin = ccz;
#pragma unroll
        for (int m = 1; m <= 2; m <<= 1) {
            umin = fminf(umin, __shfl_xor(umin, m)); umax = fmaxf(umax, __shfl_xor(umax, m));
            vmin = fminf(vmin, __shfl_xor(vmin, m)); vmax = fmaxf(vmax, __shfl_xor(vmax, m));
            zmin = fminf(zmin, __shfl_xor(zmin, m));
        }
        umin = __shfl(umin, 0); umax = __shfl(umax, 0); vmin = __shfl(vmin, 0); vmax = __shfl(vmax, 0);
        zmin = __shfl(zmin, 0);
        // all corners in front of the camera and a sane box -> clip to the image, 2-pixel safety border
        if (zmin > 0.0f && umax - umin < 4096.0f && vmax - vmin < 4096.0f && umin > -1.0e6f && vmin > -1.0e6f &&
            umax < 1.0e6f && vmax < 1.0e6f) {
            const int a0 = max(0, (int)floorf(umin) - 2), a1 = min(p.W - 1, (int)ceilf(umax) + 2);
            const int b0 = max(0, (int)floorf(vmin) - 2), b1 = min(p.H - 1, (int)ceilf(vmax) + 2);
            const int w_ = a1 - a0 + 1, h_ = b1 - b0 + 1;
            if (w_ > 0 && h_ > 0 && w_ * h_ <= kLdsTile) { tu0 = a0; tv0 = b0; tw = w_; th = h_; }
        }
        for (int ty = threadIdx.y; ty < th; ty += 4)
            for (int tx = lane; tx < tw; tx += 64)
                lds_depth[ty * tw + tx] = p.depth[(size_t)(tv0 + ty) * p.W + (tu0 + tx)];
        __syncthreads();
    }
    const int xg = bx * 64 + threadIdx.x;
    const int gy0 = (by * 4 + threadIdx.y) * R;
    if (xg >= p.xgroups || gy0 >= p.dim_y) return;
    const size_t row0 = ((size_t)lz * p.dim_y + gy0) * (size_t)p.dim_x + (size_t)xg * 4;
    const size_t flag0 = ((size_t)lz * p.dim_y + gy0) * (size_t)p.nseg + bx;

    // ---- phase 0: summary flags, and (EARLY) the speculative volume loads -----------------------
    uint32_t fl[R];
    float4 t4[R], w4[R];
    bool have_w[R], have_t[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        fl[r] = 0u;
        if (SUM && gy0 + r < p.dim_y) fl[r] = p.flags[flag0 + (size_t)r * p.nseg];
        t4[r] = make_float4(1.f, 1.f, 1.f, 1.f);
        w4[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        have_w[r] = have_t[r] = false;
    }
    if (EARLY) {
        const int x_first = bx * 256;
        const int x_last = min(x_first + 255, p.dim_x - 1);
        const int y_last = min(gy0 + R - 1, p.dim_y - 1);
        if (patch_may_be_visible(p, x_first, x_last, gy0, y_last, gz)) {
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (gy0 + r < p.dim_y) {
                    w4[r] = vol_load<NT>(p.weight + row0 + (size_t)r * p.dim_x);
                    have_w[r] = true;
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) {
                if (gy0 + r < p.dim_y && !(SUM && (fl[r] & 1u))) {
                    t4[r] = vol_load<NT>(p.tsdf + row0 + (size_t)r * p.dim_x);
                    have_t[r] = true;
                }
            }
        }
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
    int pix_u[LDSD ? R : 1][4], pix_v[LDSD ? R : 1][4];   // (LDSD) the pixel as column / row
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
    const bool fast = FAST && p.fast_ok != 0 && __ballot(unsafe) == 0ull;   // wave-uniform
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
                if constexpr (LDSD) { pix_u[r][j] = ok ? iu : 0; pix_v[r][j] = ok ? iv : 0; }
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
                if constexpr (LDSD) { pix_u[r][j] = ok ? (int)pu : 0; pix_v[r][j] = ok ? (int)pv : 0; }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            // unsigned 32-bit offset from a wave-uniform base: the load takes the base from SGPRs
            const uint32_t px = (uint32_t)pixel[r][j];
            float d;
            if constexpr (LDSD) {
                const uint32_t du = (uint32_t)(pix_u[r][j] - tu0), dv = (uint32_t)(pix_v[r][j] - tv0);
                if (du < (uint32_t)tw && dv < (uint32_t)th) d = lds_depth[dv * tw + du];
                else d = gather_f32(p.depth, px);   // outside the staged tile (or no tile): the exact pixel from memory
            } else {
                d = gather_f32(p.depth, px);
            }
            if (MASKED == 1 || (MASKED == 2 && p.mask != nullptr))
                d = d * (p.mask[px] >= 128 ? 1.0f : 0.0f);  // ref: src/Engine.cpp:192-193
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
    if (__ballot(any) == 0ull) return;  // wavefront early-out: nothing is written
    if (!any) return;

    // ---- phase 3: whatever was not pre-loaded; truncated distance -------------------------------
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (rowany[r]) {
            if (!have_w[r]) w4[r] = vol_load<NT>(p.weight + row0 + (size_t)r * p.dim_x);
            if (!have_t[r] && !(SUM && (fl[r] & 1u))) t4[r] = vol_load<NT>(p.tsdf + row0 + (size_t)r * p.dim_x);
        }
    }
    float dist[R][4];
    if (!ELIDE || __ballot(band) != 0ull) {
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
        if (SUM && fl[r] == 3u && __ballot(bandr[r]) == 0ull) {
            // Free space, wave-uniform: every TSDF value of the segment is 1, every weight is finite
            // and >= 0, and every updated lane has dist == 1.  Then num = fl(1*w + 1) = fl(w + 1) = wn,
            // the quotient is exactly 1, the TSDF row is unchanged: only the weights move.
            if (rowany[r]) {
                const float4 w = w4[r];
                vol_store<NT>(p.weight + row0 + (size_t)r * p.dim_x,
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
        if (!ELIDE || __ballot(rowany[r] && need) != 0ull) {
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
        if (SUM && (fl[r] & 1u) && __ballot(notone) != 0ull) {
            if (notone) p.flags[flag0 + (size_t)r * p.nseg] = fl[r] & 2u;  // segment no longer all ones
        }
        const bool store_t = !ELIDE || __ballot(rowany[r] && changed) != 0ull;
        if (rowany[r]) {
            if (store_t) vol_store<NT>(p.tsdf + row0 + (size_t)r * p.dim_x, make_float4(tv[0], tv[1], tv[2], tv[3]));
            vol_store<NT>(p.weight + row0 + (size_t)r * p.dim_x, make_float4(wv[0], wv[1], wv[2], wv[3]));
        }
    }
}

// CLS: the launch comes with a workgroup class table (IntegrateParams::wg_class); a workgroup whose whole patch
// the depth tile table proved untouched by this frame leaves at once (masked per-object volumes: most of them).
template <int R, bool ELIDE, bool NT, bool MASKED, bool SUM = false, bool EARLY = false, bool FAST = false,
          bool LDSD = false, bool CLS = false>
__global__ __launch_bounds__(256) void ladder_tile(IntegrateParams p)
{
    if constexpr (CLS) {
        const unsigned id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        if (p.wg_class[id] == 2) return;   // wave-uniform (scalar load): nothing to update anywhere in the patch
    }
    ladder_tile_body<R, ELIDE, NT, MASKED ? 1 : 0, SUM, EARLY, FAST, LDSD>(p, blockIdx.x, blockIdx.y, blockIdx.z);
}

// One-frame launches: the class of every workgroup's patch, one THREAD per workgroup, ahead of the Integrate launch
// (a 200^3 volume has 7 813 workgroups: this kernel is noise).  The Integrate kernel then reads one byte per
// workgroup through a scalar load.  rows_per_wg > 0: row mapping (256 x rows_per_wg voxels per workgroup, grid =
// (x blocks, y blocks, slices)); rows_per_wg == 0: flat mapping (1024 consecutive voxels, grid = (blocks, 1, slices)).
__device__ __forceinline__ int classify_wg_patch(const IntegrateParams &p, const FramePose *pose, int bx, int by, int lz,
                                                 int rows_per_wg)
{
    int xa, xb, ya, yb;
    if (rows_per_wg == 0) {
        const int n_vox = p.quads_per_slice * 4;
        const int i0 = bx * 1024;
        if (i0 >= n_vox) return 2;                       // a block past the end of the slice: nothing there
        const int i1 = min(i0 + 1023, n_vox - 1);
        ya = i0 / p.dim_x;
        yb = i1 / p.dim_x;
        xa = ya == yb ? i0 - ya * p.dim_x : 0;
        xb = ya == yb ? i1 - yb * p.dim_x : p.dim_x - 1;
    } else {
        xa = bx * 256;
        ya = by * rows_per_wg;
        if (xa >= p.dim_x || ya >= p.dim_y) return 2;
        xb = min(xa + 255, p.dim_x - 1);
        yb = min(ya + rows_per_wg - 1, p.dim_y - 1);
    }
    return classify_patch(p, class_pose(*pose), xa, xb, ya, yb, p.z_begin + lz);
}

__global__ __launch_bounds__(256) void classify_workgroups(IntegrateParams p, FramePose pose, uint8_t *cls, int nbx, int nby,
                                                           int nz, int rows_per_wg)
{
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= nbx * nby * nz) return;
    const int bx = id % nbx, t = id / nbx;
    cls[id] = (uint8_t)classify_wg_patch(p, &pose, bx, t % nby, t / nby, rows_per_wg);
}

// the batched form: slice_map[z] = {object, slice}; params / poses per object; flat mapping; grid (max_blocks, 1, total_slices)
__global__ __launch_bounds__(256) void classify_workgroups_batched(const IntegrateParams *__restrict__ params,
                                                                   const FramePose *__restrict__ poses,
                                                                   const int2 *__restrict__ slice_map, uint8_t *cls,
                                                                   int max_blocks, int total_slices)
{
    const int id = blockIdx.x * 256 + threadIdx.x;
    if (id >= max_blocks * total_slices) return;
    const int bx = id % max_blocks, z = id / max_blocks;
    const int2 m = slice_map[z];
    const IntegrateParams p = params[m.x];
    cls[id] = (uint8_t)classify_wg_patch(p, poses + m.x, bx, 0, m.y, 0);
}


struct MultiParams {
    IntegrateParams common;   // grid, intrinsics, volume pointers, summary; its pose fields are unused
    const FramePose *frames;  // n_frames blocks in device memory (indexed in a loop: a by-value array
    int n_frames;             // in the kernarg would be copied to registers and selected per frame)
};


template <int R, bool NT, bool FLAT>
__global__ __launch_bounds__(256, R == 2 ? 6 : 1) void integrate_multi(MultiParams mp)
{
    multi_body<R, NT, FLAT>(mp.common, mp.frames, mp.n_frames, blockIdx.x, blockIdx.y, blockIdx.z);
}


template <int R, bool NT, bool FLAT, bool LABELS = false, bool MASKS = true, bool SHORT = false, bool BRICK = false>
__global__ __launch_bounds__(256, R == 2 ? 6 : (BRICK ? (LABELS ? 6 : TSDF_BRICK_WAVES) : 1)) void integrate_multi_wg(MultiParamsInline mp)
{
    static_assert(!BRICK || SHORT, "the brick mapping exists for the classification's sake");
    // the single by-value parameter starts the kernarg segment (offset 0)
    typedef const char __attribute__((address_space(4))) *kernarg_ptr;
    kernarg_ptr base = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    typedef const FramePose __attribute__((address_space(4))) *frames_ptr;
    frames_ptr frames = (frames_ptr)(base + offsetof(MultiParamsInline, frames));
    unsigned int free_frames = 0u, skip_frames = 0u;
    const int wg_x = mp.z_fastest ? (int)blockIdx.z : (int)blockIdx.x, wg_y = (int)blockIdx.y;
    int wg_z = mp.z_fastest ? (int)blockIdx.x : (int)blockIdx.z;
    if (mp.z_fastest == 2) {
        wg_z += (int)((unsigned)(wg_x + wg_y) % gridDim.x);
        if (wg_z >= (int)gridDim.x) wg_z -= (int)gridDim.x;
    }
    if constexpr (BRICK) {
        if (mp.super_mask != nullptr && mp.super_mask[(size_t)wg_x * mp.nz_super + wg_z / kSuperZ] == 0u) {
            // every frame skips the whole super-brick: the workgroup's wavefront-frames are all "skipped" claims
            const IntegrateParams &p = mp.common;
            if ((p.claim_counter != nullptr || p.shortcut_stats != nullptr) && threadIdx.x == 0 && threadIdx.y == 0) {
                const int in_range = max(0, min(4, p.brick_groups * p.bricks_per_group - wg_x * 4));
                if (p.claim_counter != nullptr) atomicAdd(p.claim_counter, (unsigned long long)(in_range * mp.n_frames));
                if (p.shortcut_stats != nullptr) atomicAdd(p.shortcut_stats + 2, (unsigned)(in_range * mp.n_frames));
            }
            return;
        }
    }
    if constexpr (SHORT) {
        // Patch classification in the prologue.  The first wavefront stages the frame blocks in LDS (coalesced); then
        //   row / flat mapping: it classifies the workgroup's patch (256 x 4 voxels, or 1024 consecutive ones), one frame
        //     per lane, and bit f of two words in LDS tells every wavefront what frame f does to all of its voxels;
        //   BRICK: every wavefront classifies its own brick, one frame per lane, and keeps the two words itself.
        __shared__ FramePose s_frames[kMaxFramesPerLaunch];
        __shared__ unsigned int s_bits[2];
        __shared__ unsigned int s_claims[2];   // BRICK: (free, skipped) wavefront-frames of the workgroup, and
        __shared__ unsigned int s_done;        //        how many of its wavefronts have added theirs
        static_assert(sizeof(FramePose) % 8 == 0, "staged as 8-byte words");
        const IntegrateParams &p = mp.common;
        const int lane = threadIdx.x;
        if (threadIdx.y == 0) {
            constexpr int kWords = (int)(sizeof(FramePose) * kMaxFramesPerLaunch / 8);
            const unsigned long long __attribute__((address_space(4))) *src =
                (const unsigned long long __attribute__((address_space(4))) *)frames;
            unsigned long long *dst = reinterpret_cast<unsigned long long *>(s_frames);
            for (int k = lane; k < kWords; k += 64) dst[k] = src[k];
            if constexpr (BRICK) { if (lane == 0) { s_claims[0] = 0u; s_claims[1] = 0u; s_done = 0u; } }
            if constexpr (!BRICK) {
                int xa, xb, ya, yb;
                if constexpr (FLAT) {
                    const int i0 = wg_x * 1024;
                    const int i1 = min(i0 + 1023, p.quads_per_slice * 4 - 1);
                    ya = i0 / p.dim_x;
                    yb = i1 / p.dim_x;
                    xa = ya == yb ? i0 - ya * p.dim_x : 0;
                    xb = ya == yb ? i1 - yb * p.dim_x : p.dim_x - 1;
                } else {
                    xa = wg_x * 256;
                    xb = min(xa + 255, p.dim_x - 1);
                    ya = wg_y * 4;
                    yb = min(ya + 3, p.dim_y - 1);
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                int cls = 0;
                if (lane < mp.n_frames) cls = classify_patch(p, class_pose(s_frames[lane]), xa, xb, ya, yb, p.z_begin + wg_z);
                const unsigned long long fb = __ballot(cls == 1), sb = __ballot(cls == 2);
                if (lane == 0) {
                    s_bits[0] = (unsigned int)fb;
                    s_bits[1] = (unsigned int)sb;
                    if (p.claim_counter != nullptr && (fb | sb) != 0ull)
                        atomicAdd(p.claim_counter, ((unsigned long long)__popcll(fb) << 32) | (unsigned long long)__popcll(sb));
                }
            }
        }
        __syncthreads();
        if constexpr (BRICK) {
            const int brick = wg_x * 4 + __builtin_amdgcn_readfirstlane((int)threadIdx.y);
            const int g = brick / p.bricks_per_group, i = brick - g * p.bricks_per_group;
            const int xa = i * p.brick_q * 4, ya = g * p.brick_r;
            const int z0 = wg_z * p.brick_s, z1 = min(z0 + p.brick_s - 1, p.nz - 1);
            // frame = lane mod 32; the half-waves share the box's near and far slice (classify_patch<PAIRED>)
            static_assert(kMaxFramesPerLaunch == 32, "one frame per lane of a half-wave");
            int cls = classify_patch<true>(p, class_pose(s_frames[lane & 31]), xa, xa + p.brick_q * 4 - 1, ya,
                                           min(ya + p.brick_r - 1, p.dim_y - 1), p.z_begin + z0, p.z_begin + z1);
            if (g >= p.brick_groups || lane >= mp.n_frames) cls = 0;
            const unsigned long long fb = __ballot(cls == 1), sb = __ballot(cls == 2);
            free_frames = (unsigned int)fb;
            skip_frames = (unsigned int)sb;
            if (p.claim_counter != nullptr && lane == 0) {
                // one global atomic per workgroup: the wavefronts add up in LDS, the last one to arrive passes the sum on
                if (fb != 0ull) atomicAdd(&s_claims[0], (unsigned)__popcll(fb));
                if (sb != 0ull) atomicAdd(&s_claims[1], (unsigned)__popcll(sb));
                __threadfence_block();
                if (atomicAdd(&s_done, 1u) == 3u) {
                    __threadfence_block();
                    const unsigned long long nf = s_claims[0], ns = s_claims[1];
                    if ((nf | ns) != 0ull) atomicAdd(p.claim_counter, (nf << 32) | ns);
                }
            }
        } else {
            free_frames = s_bits[0];
            skip_frames = s_bits[1];
        }
        free_frames = __builtin_amdgcn_readfirstlane(free_frames);
        skip_frames = __builtin_amdgcn_readfirstlane(skip_frames);
    }
    multi_body<R, NT, FLAT, LABELS, MASKS, SHORT, BRICK>(mp.common, (const FramePose *)frames, mp.n_frames,
                                                         BRICK ? wg_x * 4 + __builtin_amdgcn_readfirstlane((int)threadIdx.y) : wg_x,
                                                         wg_y, wg_z, mp.labels, free_frames, skip_frames);
}


// Ahead of a BRICK launch: which frames may do something to each super-brick -- the box of a workgroup's four bricks over
// kSuperZ consecutive slice groups -- one WAVEFRONT per super-brick, one frame per lane (paired half-waves, as in the
// launch's own prologue; the same classify_patch, so the same exactness argument, on a larger box).  Most of a realistic
// launch's workgroups are skipped by every frame (the volume behind the surfaces and outside the views: 45 % of S-surf's
// workgroups, more on a trajectory); with their word 0 they cost a dispatch instead of staging + barrier + four
// classifications.  grid = ceil(super-bricks / 4) x 256 threads.
__global__ __launch_bounds__(256) void classify_superbricks(MultiParamsInline mp, unsigned int *out, int n_wgx)
{
    typedef const char __attribute__((address_space(4))) *kernarg_ptr;
    kernarg_ptr base = (kernarg_ptr)__builtin_amdgcn_kernarg_segment_ptr();
    const FramePose *frames = (const FramePose *)(base + offsetof(MultiParamsInline, frames));
    const IntegrateParams &p = mp.common;
    const int id = (int)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)threadIdx.y), lane = threadIdx.x;
    if (id >= n_wgx * mp.nz_super) return;
    const int wx = id / mp.nz_super, zs = id - wx * mp.nz_super;
    const int total = p.brick_groups * p.bricks_per_group;
    const int b0 = wx * 4, b3 = min(b0 + 3, total - 1);
    if (b0 >= total) {
        if (lane == 0) out[id] = 0u;
        return;
    }
    const int g0 = b0 / p.bricks_per_group, i0 = b0 - g0 * p.bricks_per_group;
    const int g3 = b3 / p.bricks_per_group, i3 = b3 - g3 * p.bricks_per_group;
    // the four bricks lie side by side in one row group, or wrap into the next one (then: the full width of both)
    const int xa = g0 == g3 ? i0 * p.brick_q * 4 : 0, xb = g0 == g3 ? (i3 + 1) * p.brick_q * 4 - 1 : p.dim_x - 1;
    const int ya = g0 * p.brick_r, yb = min((g3 + 1) * p.brick_r - 1, p.dim_y - 1);
    const int nz_groups = (p.nz + p.brick_s - 1) / p.brick_s;
    const int zg0 = zs * kSuperZ, zg1 = min(zg0 + kSuperZ - 1, nz_groups - 1);
    const int z0 = zg0 * p.brick_s, z1 = min((zg1 + 1) * p.brick_s - 1, p.nz - 1);
    const int cls = classify_patch<true>(p, class_pose(frames[lane & 31]), xa, xb, ya, yb, p.z_begin + z0, p.z_begin + z1);
    const unsigned long long work = __ballot(cls != 2 && lane < mp.n_frames);
    if (lane == 0) out[id] = (unsigned int)work;
}


// Experiment (variant 5): XCD-aware block order.  Workgroups are dealt round-robin over the 8 XCDs, so
// ids b and b + 8 share an L2; this remap hands each XCD one contiguous eighth of the slab instead of
// every eighth workgroup.  Measured: no gain -- the only shared data is the 1.2 MB
// depth frame, which every XCD's 4 MiB L2 holds whole either way; volume bytes are touched once.
template <bool NT>
__global__ __launch_bounds__(256) void integrate_multi_xcd(MultiParams mp)
{
    const unsigned nx = gridDim.x, ny = gridDim.y, n = nx * ny * gridDim.z;
    unsigned id = blockIdx.x + nx * (blockIdx.y + ny * blockIdx.z);
    if (n % 8u == 0u) id = (id % 8u) * (n / 8u) + id / 8u;
    const unsigned bx = id % nx, t = id / nx;
    multi_body<1, NT, false>(mp.common, mp.frames, mp.n_frames, (int)bx, (int)(t % ny), (int)(t / ny));
}


// One frame, pose by value (no frame block in memory to stage): what a single tsdf_integrate* call
// on a volume served by the flat mapping launches.
// CLS: with a workgroup class table (IntegrateParams::wg_class, classify_workgroups): skipped workgroups leave at
// once, free-space ones only add to their weights.
template <bool NT, bool FLAT, bool CLS = false>
__global__ __launch_bounds__(256) void integrate_multi_single_cls(IntegrateParams p, FramePose pose)
{
    if constexpr (CLS) {
        const unsigned id = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
        const unsigned c = p.wg_class[id];
        if (c == 2u) return;
        multi_body<1, NT, FLAT, false, true, true>(p, &pose, 1, blockIdx.x, blockIdx.y, blockIdx.z, LabelState(), c == 1u ? 1u : 0u, 0u);
    } else {
        multi_body<1, NT, FLAT>(p, &pose, 1, blockIdx.x, blockIdx.y, blockIdx.z);
    }
}


// Many volumes, one frame, one launch (the reference's real usage: one small TSDF per object
// instance, each fed depth x its own instance mask; ref: src/Engine.cpp:172-233, src/Object.cpp:67).
// params[o] / poses[o]: the parameter block and this frame's relative pose + mask of object o (each has
// its own base frame); slice_map[z] = {object, slice within it} for every slice of every object;
// grid = (max blocks per slice, 1, total slices).  Flat mapping: object grids are small and rarely
// 256 wide.  The blocks are read through a wave-uniform index (scalar loads).
// wg_class (may be null): class of every workgroup of this launch for this frame (classify_workgroups_batched):
// per-object volumes are fed depth x their instance mask, so most of their workgroups see nothing and leave at once.
template <bool NT, bool CLS>
__global__ __launch_bounds__(256) void integrate_multi_batched_cls(const IntegrateParams *__restrict__ params,
                                                               const FramePose *__restrict__ poses,
                                                               const int2 *__restrict__ slice_map,
                                                               const uint8_t *__restrict__ wg_class)
{
    unsigned c = 0u;
    if constexpr (CLS) {
        c = wg_class[blockIdx.x + gridDim.x * blockIdx.z];
        if (c == 2u) return;
    }
    const int2 m = slice_map[blockIdx.z];
    const IntegrateParams p = params[m.x];
    if constexpr (CLS)
        multi_body<1, NT, true, false, true, true>(p, poses + m.x, 1, blockIdx.x, 0, m.y, LabelState(), c == 1u ? 1u : 0u, 0u);
    else
        multi_body<1, NT, true>(p, poses + m.x, 1, blockIdx.x, 0, m.y);
}


}  // namespace tsdfk
