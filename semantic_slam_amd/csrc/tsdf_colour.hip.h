// tsdf_colour.hip.h -- per-voxel colour fusion for the TSDFfusion surface.
//
// The reference's second backend hands every RGB-D frame to the third-party package tsdf-fusion-python
// (ref: src/TSDFfusion.py.in:43 `tsdf_vol.integrate(color_image, depth_im, cam_intr, cam_pose, obs_weight=1.)`), which
// fuses colour beside the distance and writes a coloured mesh (ref: src/TSDFfusion.py.in:48-53).  The package is not
// vendored and absent, so its arithmetic cannot be pinned; this file restates its published rule:
// every voxel the frame updates takes, per 8-bit channel, the weighted running mean of its colour and the colour of
// its pixel, rounded and clamped,
//     c' = min(255, round((c * w_old + obs_weight * c_pixel) / w_new)),   obs_weight = 1, w_new = w_old + 1,
// and stores the three channels packed (B << 16 | G << 8 | R; the package keeps B*65536 + G*256 + R in a float).
// The voxels updated, their pixel and the weights are exactly Integrate's (same projection code, FAST path included):
// the colour pass of a frame runs right after its Integrate pass and reads the weights that pass has just written.
// Parity: against this project's own CPU restatement of the same rule only -- "parity unpinned".
#pragma once
#include "tsdf_kernels.hip.h"

namespace tsdfk {

struct ColourParams {
    IntegrateParams g;      // grid, intrinsics, pose, depth, weight (tsdf / flags unused)
    const uint8_t *rgb;     // H*W*3, channel 0 = R (the package packs colour_im[..., 0] into the low byte)
    uint32_t *colour;       // slab array, packed 0x00BBGGRR
};

__device__ __forceinline__ uint32_t blend_channel(uint32_t old_c, uint32_t new_c, float w_old, float w_new)
{
    const float v = fminf(roundf(((float)old_c * w_old + (float)new_c) / w_new), 255.0f);
    return (uint32_t)v;
}

__global__ __launch_bounds__(256) void integrate_colour(ColourParams cp)
{
    const IntegrateParams &p = cp.g;
    const int xg = blockIdx.x * 64 + threadIdx.x;
    const int gy = blockIdx.y * 4 + threadIdx.y;
    const int lz = blockIdx.z;
    if (xg >= p.xgroups || gy >= p.dim_y) return;
    const int gz = p.z_begin + lz;

    // geometry of the lane's 4 voxels (ref: src/tsdf.cu:27-43), as integrate_tile with R = 1
    float ax[4], ay[4], az[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float dx = (p.ox + (float)(xg * 4 + j) * p.vs) - p.tx;
        ax[j] = p.rx0 * dx; ay[j] = p.ry0 * dx; az[j] = p.rz0 * dx;
    }
    const float dy = (p.oy + (float)gy * p.vs) - p.ty;
    const float dz = (p.oz + (float)gz * p.vs) - p.tz;
    const float x1 = p.rx1 * dy, y1 = p.ry1 * dy, z1 = p.rz1 * dy;
    const float x2 = p.rx2 * dz, y2 = p.ry2 * dz, z2 = p.rz2 * dz;
    float pcz[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) pcz[j] = az[j] + z1 + z2;
    const float cmin = fminf(pcz[0], pcz[3]), cmax = fmaxf(pcz[0], pcz[3]);
    const bool unsafe = !(cmin > p.cz_margin) & !(cmax < -p.cz_margin);
    bool geo[4];
    int pixel[4];
    if (p.fast_ok != 0 && __ballot(unsafe) == 0ull) {
        const v2f F = {p.fx, p.fy}, C = {p.cx, p.cy};
        const v2f XY1 = {x1, y1}, XY2 = {x2, y2};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const v2f A = {ax[j], ay[j]};
            const v2f n = A + XY1 + XY2;
            const v2f uv = F * fast_div2(n, pcz[j]) + C;
            const int iu = round_half_up_i32(uv.x), iv = round_half_up_i32(uv.y);
            const bool ok = pcz[j] > 0.0f && uv.x > -0.5f && uv.y > -0.5f && (unsigned)iu < (unsigned)p.W &&
                            (unsigned)iv < (unsigned)p.H;
            geo[j] = ok;
            pixel[j] = ok ? iv * p.W + iu : 0;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float cx = ax[j] + x1 + x2, cy = ay[j] + y1 + y2, cz = pcz[j];
            const float pu = roundf(p.fx * (cx / cz) + p.cx);
            const float pv = roundf(p.fy * (cy / cz) + p.cy);
            const bool ok = !(cz <= 0.0f) && pu >= 0.0f && pu < (float)p.W && pv >= 0.0f && pv < (float)p.H;
            geo[j] = ok;
            pixel[j] = ok ? (int)pv * p.W + (int)pu : 0;
        }
    }
    // the voxels this frame's Integrate pass updated (ref: src/tsdf.cu:46-49)
    bool upd[4];
    bool any = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float d = p.depth[(uint32_t)pixel[j]];
        const float df = d - pcz[j];
        upd[j] = geo[j] & !((d <= 0.0f) | (d > p.max_depth)) & !(df <= -p.trunc);
        any |= upd[j];
    }
    if (__ballot(any) == 0ull) return;
    if (!any) return;

    const size_t row = ((size_t)lz * p.dim_y + gy) * (size_t)p.dim_x + (size_t)xg * 4;
    const float4 w4 = *reinterpret_cast<const float4 *>(p.weight + row);      // already this frame's w_new
    uint4 c4 = *reinterpret_cast<const uint4 *>(cp.colour + row);
    const float wn[4] = {w4.x, w4.y, w4.z, w4.w};
    uint32_t col[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (!upd[j]) continue;
        const uint8_t *px = cp.rgb + (size_t)(uint32_t)pixel[j] * 3u;
        const float w_new = wn[j], w_old = w_new - 1.0f;
        const uint32_t r = blend_channel(col[j] & 255u, px[0], w_old, w_new);
        const uint32_t g = blend_channel((col[j] >> 8) & 255u, px[1], w_old, w_new);
        const uint32_t b = blend_channel((col[j] >> 16) & 255u, px[2], w_old, w_new);
        col[j] = (b << 16) | (g << 8) | r;
    }
    *reinterpret_cast<uint4 *>(cp.colour + row) = make_uint4(col[0], col[1], col[2], col[3]);
}

}  // namespace tsdfk
