// tsdf_labels.hip.h -- per-voxel semantic-label fusion (BASELINE config 5).
//
// Not part of the reference's TSDF.  The reference fuses instance evidence on sparse ObjectPoints:
// Fp += score when the point is observed inside a mask of its own object, Bp += score otherwise,
// P = Fp / (Fp + Bp), the point is dropped when P < threshold (ref: src/ObjectPoint.cpp:190-219,
// :149-154; Engine.mProbThd, config/TUM3.yaml:92).  Here the same evidence rule runs per voxel of
// the dense grid, for the voxels a frame observes inside the truncation band:
//     l = label image at the voxel's pixel (0 = no instance), s = score image there
//     label == 0 -> adopt (label = l, Fp = s, Bp = 0);  label == l -> Fp += s;
//     else Bp += s and, if Fp / (Fp + Bp) < threshold, re-adopt l.
// The voxel's pixel and the depth tests are exactly Integrate's (same projection code, FAST path
// included), so a label pass and an Integrate pass of the same frame agree on which voxels are seen.
// Layout: label uint16, Fp fp32, Bp fp32, three arrays indexed like the TSDF (10 B/voxel resident);
// only voxels in the band are read or written, a wavefront with none leaves before touching them.
#pragma once
#include "tsdf_kernels.hip.h"

namespace tsdfk {

struct LabelParams {
    IntegrateParams g;          // grid, intrinsics, pose, depth (tsdf/weight/flags unused)
    const uint16_t *label_im;   // H*W, 0 = background
    const float *score_im;      // H*W
    uint16_t *label;            // slab arrays
    float *fp, *bp;
    float prob_thd;
};

__global__ __launch_bounds__(256) void integrate_labels(LabelParams lp)
{
    const IntegrateParams &p = lp.g;
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
    // depth tests (ref: src/tsdf.cu:46-49) + inside the truncation band
    bool seen[4];
    bool any = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float d = p.depth[(uint32_t)pixel[j]];
        const float df = d - pcz[j];
        seen[j] = geo[j] & !((d <= 0.0f) | (d > p.max_depth)) & !(df <= -p.trunc) & (df < p.trunc);
        any |= seen[j];
    }
    if (__ballot(any) == 0ull) return;
    uint16_t lab_in[4];
    float sc_in[4];
    bool hit = false;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        lab_in[j] = seen[j] ? lp.label_im[(uint32_t)pixel[j]] : (uint16_t)0;
        sc_in[j] = seen[j] ? lp.score_im[(uint32_t)pixel[j]] : 0.0f;
        hit |= lab_in[j] != 0;
    }
    if (!hit) return;

    const size_t row = ((size_t)lz * p.dim_y + gy) * (size_t)p.dim_x + (size_t)xg * 4;
    ushort4 L4 = *reinterpret_cast<const ushort4 *>(lp.label + row);
    float4 F4 = *reinterpret_cast<const float4 *>(lp.fp + row);
    float4 B4 = *reinterpret_cast<const float4 *>(lp.bp + row);
    uint16_t L[4] = {L4.x, L4.y, L4.z, L4.w};
    float Fv[4] = {F4.x, F4.y, F4.z, F4.w}, Bv[4] = {B4.x, B4.y, B4.z, B4.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const uint16_t l = lab_in[j];
        const float s = sc_in[j];
        if (l == 0) continue;
        if (L[j] == 0) { L[j] = l; Fv[j] = s; Bv[j] = 0.0f; }
        else if (L[j] == l) { Fv[j] = Fv[j] + s; }
        else {
            Bv[j] = Bv[j] + s;
            if (Fv[j] / (Fv[j] + Bv[j]) < lp.prob_thd) { L[j] = l; Fv[j] = s; Bv[j] = 0.0f; }
        }
    }
    ushort4 Lo; Lo.x = L[0]; Lo.y = L[1]; Lo.z = L[2]; Lo.w = L[3];
    *reinterpret_cast<ushort4 *>(lp.label + row) = Lo;
    *reinterpret_cast<float4 *>(lp.fp + row) = make_float4(Fv[0], Fv[1], Fv[2], Fv[3]);
    *reinterpret_cast<float4 *>(lp.bp + row) = make_float4(Bv[0], Bv[1], Bv[2], Bv[3]);
}

// Label / score images from K instance masks (MaskRCNN output format: K x H x W uint8 {0,255},
// label 1..80 and score per instance; ref: src/MaskRCNN.cpp:316-362): per pixel the covering
// instance with the highest score wins, the lower index on ties; uncovered pixels get 0 / 0.
constexpr int kMaxInstances = 128;
struct ComposeParams {
    const uint8_t *masks;
    uint16_t *label_im;
    float *score_im;
    int k, n_pixels;
    uint16_t labels[kMaxInstances];
    float scores[kMaxInstances];
};

__global__ __launch_bounds__(256) void compose_labels(ComposeParams c)
{
    for (int px = blockIdx.x * blockDim.x + threadIdx.x; px < c.n_pixels; px += gridDim.x * blockDim.x) {
        uint16_t l = 0;
        float s = 0.0f;
        for (int m = 0; m < c.k; ++m)
            if (c.masks[(size_t)m * c.n_pixels + px] >= 128 && (l == 0 || c.scores[m] > s)) { l = c.labels[m]; s = c.scores[m]; }
        c.label_im[px] = l;
        c.score_im[px] = s;
    }
}

}  // namespace tsdfk
