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
// value (see "rounding-safe projection" in DESIGN.md), never the value itself.
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
// integrate_rows<VX>: block = 64 x 4 threads; a wavefront = 64 lanes x VX voxels of one row.
// grid = (ceil(xgroups/64), ceil(dim_y/4), nz).
// VX = 4 needs dim_x % 4 == 0 (rows stay 16-byte aligned); VX = 1 takes any dim_x.
// ------------------------------------------------------------------------------------------
template <int VX, bool MASKED>
__global__ __launch_bounds__(256) void integrate_rows(IntegrateParams p)
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
