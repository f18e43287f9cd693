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

// ------------------------------------------------------------------------------------------
// integrate_tile<R, ELIDE, NT, MASKED>: the tuned kernel.
//
// block = 64 x 4 threads; a lane owns a 4(x) x R(y) patch of one z slice, a wavefront
// 256(x) x R(y).  grid = (ceil(xgroups/64), ceil(dim_y/(4R)), nz).  Needs dim_x % 4 == 0.
//
//  * the x-only products (rx0*dx, ry0*dx, rz0*dx) are computed once and shared by the R rows;
//  * all 4R depth samples of a lane are gathered before any is used (branch-free geometry,
//    rejected voxels read pixel 0), then all 2R volume quads are loaded together: two memory
//    round trips per wavefront per R rows instead of two per row;
//  * a wavefront with nothing to update leaves before touching the volume (__ballot);
//  * ELIDE: arithmetic whose result is known exactly is skipped per wavefront --
//      - diff >= trunc  =>  fmin(1, diff/trunc) == 1: no division unless some lane is inside
//        the truncation band (correctly rounded a/b >= 1 whenever a >= b > 0);
//      - tsdf*w + dist == w + 1 (free space: tsdf 1, dist 1)  =>  the quotient is exactly 1:
//        no division unless some lane differs;
//      - a row whose TSDF values all come out bit-identical to what was loaded is not stored
//        (the weight always changes and is always stored).
//    Every skipped value is the value the full computation would produce, bit for bit.
//  * NT: volume loads/stores carry the non-temporal hint (each byte is touched once per frame).
// ------------------------------------------------------------------------------------------
typedef float v4f __attribute__((ext_vector_type(4)));

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

template <int R, bool ELIDE, bool NT, bool MASKED>
__global__ __launch_bounds__(256) void integrate_tile(IntegrateParams p)
{
    const int xg = blockIdx.x * 64 + threadIdx.x;
    const int gy0 = (blockIdx.y * 4 + threadIdx.y) * R;
    const int lz = blockIdx.z;
    if (xg >= p.xgroups || gy0 >= p.dim_y) return;
    const int gz = p.z_begin + lz;

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
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const int gy = gy0 + r;
        const bool row_ok = gy < p.dim_y;
        const float by = p.oy + (float)gy * p.vs;   // ref: src/tsdf.cu:28
        const float dy = by - p.ty;
        const float x1 = p.rx1 * dy, y1 = p.ry1 * dy, z1 = p.rz1 * dy;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float cx = ax[j] + x1 + x2;
            const float cy = ay[j] + y1 + y2;
            const float cz = az[j] + z1 + z2;
            pcz[r][j] = cz;
            // ref: src/tsdf.cu:39-43.  cz <= 0 is tested before the quotient is used, exactly
            // as the reference's `continue`; the division itself is harmless for any cz.
            const float pu = roundf(p.fx * (cx / cz) + p.cx);
            const float pv = roundf(p.fy * (cy / cz) + p.cy);
            const bool ok = row_ok && !(cz <= 0.0f) && pu >= 0.0f && pu < (float)p.W && pv >= 0.0f &&
                            pv < (float)p.H;
            geo[r][j] = ok;
            const int pix = ok ? (int)pv * p.W + (int)pu : 0;
            float d = p.depth[pix];
            if (MASKED) d = d * (p.mask[pix] >= 128 ? 1.0f : 0.0f);  // ref: src/Engine.cpp:192-193
            dval[r][j] = d;
        }
    }

    // ---- phase 2: depth tests (ref: src/tsdf.cu:46-49) ------------------------------------
    float diff[R][4];
    bool upd[R][4], rowany[R];
    bool any = false, band = false;
#pragma unroll
    for (int r = 0; r < R; ++r) {
        rowany[r] = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = dval[r][j];
            const float df = d - pcz[r][j];
            diff[r][j] = df;
            const bool u = geo[r][j] && !(d <= 0.0f || d > p.max_depth) && !(df <= -p.trunc);
            upd[r][j] = u;
            rowany[r] |= u;
            band |= u && !(df >= p.trunc);
        }
        any |= rowany[r];
    }
    if (__ballot(any) == 0ull) return;  // wavefront early-out: no volume traffic at all
    if (!any) return;

    // ---- phase 3: volume quads in, truncated distance ----------------------------------------
    const size_t row0 = ((size_t)lz * p.dim_y + gy0) * (size_t)p.dim_x + (size_t)xg * 4;
    float4 t4[R], w4[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        t4[r] = make_float4(1.f, 1.f, 1.f, 1.f);
        w4[r] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rowany[r]) {
            t4[r] = vol_load<NT>(p.tsdf + row0 + (size_t)r * p.dim_x);
            w4[r] = vol_load<NT>(p.weight + row0 + (size_t)r * p.dim_x);
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
        bool changed = false;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float newt = upd[r][j] ? nt[j] : tv[j];
            changed |= __float_as_uint(newt) != __float_as_uint(tv[j]);
            tv[j] = newt;
            wv[j] = upd[r][j] ? wn[j] : wv[j];
        }
        const bool store_t = !ELIDE || __ballot(rowany[r] && changed) != 0ull;
        if (rowany[r]) {
            if (store_t) vol_store<NT>(p.tsdf + row0 + (size_t)r * p.dim_x, make_float4(tv[0], tv[1], tv[2], tv[3]));
            vol_store<NT>(p.weight + row0 + (size_t)r * p.dim_x, make_float4(wv[0], wv[1], wv[2], wv[3]));
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
