// host_derive.h -- what the host derives from a configuration and a pose before a launch: the wavefront brick, the depth tile
// edge and the guards / margins that make the kernels' shortcuts exact ("Why each shortcut is exact").
// Plain C++ (no HIP, no device types): tsdf_capi.hip includes it for the product, and the CPU sanitizer run compiles it by
// itself with -fsanitize=address,undefined (tests/test_sanitizers.py).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>

#include "tsdf_hip.h"

namespace tsdf_host {

// The wavefront brick of classified launches: q quads (4q voxels) of r rows of s slices, q * r * s <= 64 lanes, q a
// divisor of the row's quads.  A brick is classified as a whole, so what counts is how tightly its box projects
// (few depth tiles, a short range of camera depths) and how it coalesces (16q-byte row pieces).
inline bool brick_shape_ok(const tsdf_config &c, int q, int r, int s)
{
    // (the last condition: a lane's byte offset within its brick is a 32-bit number in the kernels)
    return c.dim_x % 4 == 0 && q >= 1 && r >= 1 && s >= 1 && q * r * s <= 64 && (c.dim_x / 4) % q == 0 &&
           (long long)s * c.dim_x * c.dim_y < (1ll << 30);
}

// The library's choice: the shape that lets the fewest bricks touch a surface band.  A brick of X x Y x Z voxels is
// claimed unless the band (about 10 voxels thick) crosses its box grown by the slack of the depth tiles it is tested
// against (about 8 voxels either way in x and y at the usual 1 - 2 pixels per voxel), so the share of per-voxel
// bricks goes like (X + 8)(Y + 8)(Z + 10) / XYZ; idle lanes and row pieces under 64 bytes (q < 4) cost on top.
// Measured over shapes at 512^3 S-surf (ms per frame, fused): 2,4,8 0.0431; 2,8,4 0.0475; 4,4,4 0.0479; 4,8,2
// 0.0492; 1,8,8 0.0515; 8,8,1 0.0528; 16,4,1 0.0669 -- the same order as this cost; 200^3 @ 4 mm and the 1024^3
// trajectory agree (tools/claim_sim.py replays the claim rule on the CPU: 8 x 4 x 8 voxels leaves the
// fewest bricks undecided of eight shapes).
inline void choose_brick_default(const tsdf_config &c, int &bq, int &br, int &bs)
{
    bq = 0; br = 0; bs = 1;
    if (c.dim_x % 4 != 0) return;
    const int quads = c.dim_x / 4;
    const int nz = c.z_end - c.z_begin;
    double best = 1e300;
    for (int q = 1; q <= 64 && q <= quads; ++q) {
        if (quads % q) continue;
        for (int sl = 1; q * sl <= 64 && sl <= std::max(nz, 1); ++sl) {
            const int r = std::min(64 / (q * sl), std::max(c.dim_y, 1));
            if (!brick_shape_ok(c, q, r, sl)) continue;
            const double X = 4.0 * q, Y = r, Z = sl;
            const double cost = (X + 8.0) * (Y + 8.0) * (Z + 10.0) / (X * Y * Z) * (1.0 + 0.25 / q) * 64.0 / (q * r * sl);
            if (cost < best) { best = cost; bq = q; br = r; bs = sl; }
        }
    }
}

// Guards and margins of one frame: everything below is derived from the configuration and the frame's cam2base pose.
struct ProjectionGuards {
    float cz_margin;    // corner test of a lane's patch against the camera plane (claim 6)
    int fast_ok;        // the shared-reciprocal projection is exact for every operand this slab and pose can produce (claim 4)
    int trunc_fast;     // diff / trunc through the launch-wide refined reciprocal
    float cz_short;     // classify_patch: camera-frame z below which a box's projected corners are not trusted
    float cz_pad;       // classify_patch: widening of the z bounds (twice the error bound on cz)
    float px_margin_u, px_margin_v;   // classify_patch: widening of the projected pixel box
};

// c2b: row-major 4x4 camera-to-base pose of the frame (ref: src/tsdf.cu:142).
inline ProjectionGuards derive_projection_guards(const tsdf_config &c, const float *c2b)
{
    ProjectionGuards g;
    const double fx = c.cam_K[0], fy = c.cam_K[4], cx = c.cam_K[2], cy = c.cam_K[5];
    // The shared-reciprocal projection (tsdf_kernels.hip.h, fast_div2) is exact when no operand
    // needs div_scale's pre-scaling: bound every camera-frame coordinate of the slab by
    // sum_j |R_ij| * max|d_j| and keep it, and the intrinsics, far from the exponent limits.
    // Anything else (including NaN/inf in the pose) takes the generic IEEE-division path.
    const double ext[3] = {(double)(c.dim_x - 1) * c.voxel_size, (double)(c.dim_y - 1) * c.voxel_size,
                           (double)(c.dim_z - 1) * c.voxel_size};
    const double o[3] = {c.origin[0], c.origin[1], c.origin[2]}, t[3] = {c2b[3], c2b[7], c2b[11]};
    double dmax[3];
    for (int k = 0; k < 3; ++k) dmax[k] = std::fmax(std::fabs(o[k] - t[k]), std::fabs(o[k] + ext[k] - t[k])) * 1.001 + 1e-30;
    // rows[i][k]: coefficient of d_k in camera coordinate i (the transposed rotation: columns of the row-major pose)
    const double rows[3][3] = {{c2b[0], c2b[4], c2b[8]}, {c2b[1], c2b[5], c2b[9]}, {c2b[2], c2b[6], c2b[10]}};
    bool ok = true;
    double bz = 0;
    for (int i = 0; i < 3; ++i) {
        double b = 0;
        for (int k = 0; k < 3; ++k) b += std::fabs(rows[i][k]) * dmax[k];
        ok = ok && (b < 5.7e17);  // 2^59; false for NaN/inf
        if (i == 2) bz = b;
    }
    // rounding error of cz is < 4 ulp of bz (2.4e-7 bz): the margin is 40x that, never below 1e-17
    g.cz_margin = ok ? (float)std::fmax(1e-5 * bz, 1e-17) : 3.0e38f;
    ok = ok && std::fabs(fx) < 16384.0 && std::fabs(fy) < 16384.0 &&
         std::fabs(cx) < 1048576.0 && std::fabs(cy) < 1048576.0 &&
         (int64_t)c.im_width * c.im_height <= (1 << 24) &&   // pixel index exact in fp32
         c.im_width < (1 << 24) && c.im_height < (1 << 24);   // and its factors fit the 24-bit multiply
    g.fast_ok = ok ? 1 : 0;
    // diff / trunc through the shared reciprocal (tsdf_kernels.hip.h, fast_div_r): divisor and numerator ranges
    g.trunc_fast = (ok && c.trunc_margin >= 9.5367431640625e-07f && c.trunc_margin <= 1048576.0f &&
                    c.max_depth <= 5.7e17f) ? 1 : 0;   // 2^-20 .. 2^20; max_depth <= 2^59 (false for NaN)
    // Patch classification (tsdf_multiframe.hip.h, classify_patch).  E_k bounds how far a voxel's d_k = (o_k +
    // i*vs) - t_k, as rounded on the per-voxel path, lies from the affine function of the index (two roundings at
    // the magnitude of the coordinate, one at that of the difference); eps bounds the error of a camera-frame
    // coordinate on either path (those, through the rotation, plus five roundings at the magnitude b_i), twice.
    double bmax = 0, eps = 0;
    for (int i = 0; i < 3; ++i) {
        double b = 0, e = 0;
        for (int k = 0; k < 3; ++k) {
            const double Ek = 1.2e-7 * (std::fabs(o[k]) + ext[k] + std::fabs(t[k])) + 6e-8 * dmax[k];
            b += std::fabs(rows[i][k]) * dmax[k];
            e += std::fabs(rows[i][k]) * Ek;
        }
        bmax = std::fmax(bmax, b);
        eps = std::fmax(eps, 2.0 * (e + 3.0e-7 * b));
    }
    const bool sok = ok && bmax > 0 && eps < 1e30;
    g.cz_short = sok ? (float)(std::fmax(bmax / 64.0, eps / 3.2e-5) * 1.0001) : 3.0e38f;
    g.cz_pad = sok ? (float)(std::fmax(2.0 * eps, (double)g.cz_margin) * 1.0001) : 3.0e38f;
    // By how much the pixel box of a patch's projected corners is widened so that it holds the rounded pixel of every voxel of
    // the patch as the per-voxel path computes it:  0.5 (a pixel index is within half a pixel of its u)  +  the projection error
    // of BOTH paths for cz >= cz_short, |fx| * (eps / cz) * (1 + |t|) with eps / cz <= 3.2e-5 (eps is the sum of the two paths'
    // camera-coordinate errors, above) and |t| <= 4 (W + |cx|) / |fx| for every corner that can matter (a corner with a larger
    // tangent projects more than four image widths outside, where an error of a pixel changes nothing; cz >= bmax / 64 keeps
    // it below 1.1 pixels there)  +  1/16 for the roundings of u = fx * q + cx itself (two ulp of a number below 2^13 for such
    // corners: 2e-3).  (Round 2 carried a whole pixel of unexplained slack on top: with 8-pixel tiles that pixel decided one
    // box in twelve -- S-surf 512^3: 11.6 % -> 10.7 % of the wavefront-frames per voxel, 0.0270 -> 0.0257 ms per frame.)
    g.px_margin_u = (float)(0.5625 + 3.2e-5 * (std::fabs(fx) + 4.0 * (c.im_width + std::fabs(cx))));
    g.px_margin_v = (float)(0.5625 + 3.2e-5 * (std::fabs(fy) + 4.0 * (c.im_height + std::fabs(cy))));
    return g;
}

}  // namespace tsdf_host
