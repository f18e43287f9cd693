// tsdf_extract.hip.h -- surface point extraction on the device.
//
// The reference scans the whole grid twice on the host, once to count and once to write
// (ref: src/tsdf.cu:176-216), keeping voxel i when |tsdf[i]| != 0 and weight[i] > thresh and
// emitting origin + index*voxel_size in grid order.  Here the same predicate is evaluated
// by an order-preserving stream compaction: per-chunk counts -> exclusive scan -> emit, all
// reads coalesced (lane <-> consecutive voxels), ranks inside a wavefront from ballots.
// Output order is grid order, so the point list is byte-comparable with the reference's.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace tsdfx {

constexpr int kPerThread = 16;
constexpr int kChunk = 256 * kPerThread;  // voxels per workgroup

__device__ __forceinline__ bool is_surface_value(float t, float w, float thr)
{
    // ref: src/tsdf.cu:179 (tsdf_thresh is unused there too)
    return fabsf(t) != 0.0f && w > thr;
}

// The four voxels [i0, i0 + 4) of a lane: one 16-byte load per array where the quad lies inside the slab (i0 is a multiple
// of four and the arrays are 16-byte aligned), single loads at the slab's ragged end.  Bit j of the result = voxel i0 + j is
// a surface point.
__device__ __forceinline__ uint32_t surface_quad(const float *tsdf, const float *weight, int64_t i0, int64_t n, float thr)
{
    if (i0 + 3 < n) {
        const float4 t = *reinterpret_cast<const float4 *>(tsdf + i0);
        const float4 w = *reinterpret_cast<const float4 *>(weight + i0);
        return (is_surface_value(t.x, w.x, thr) ? 1u : 0u) | (is_surface_value(t.y, w.y, thr) ? 2u : 0u) |
               (is_surface_value(t.z, w.z, thr) ? 4u : 0u) | (is_surface_value(t.w, w.w, thr) ? 8u : 0u);
    }
    uint32_t bits = 0u;
    for (int j = 0; j < 4; ++j)
        if (i0 + j < n && is_surface_value(tsdf[i0 + j], weight[i0 + j], thr)) bits |= 1u << j;
    return bits;
}

constexpr int kQuadPasses = kChunk / 1024;   // passes of 256 lanes x 4 voxels over a workgroup's chunk

__global__ __launch_bounds__(256) void surface_count(const float *tsdf, const float *weight,
                                                     int64_t n, float thr, uint32_t *counts)
{
    __shared__ uint32_t wave_sum[4];
    const int64_t base = (int64_t)blockIdx.x * kChunk;
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < kQuadPasses; ++k)
        c += (uint32_t)__popc(surface_quad(tsdf, weight, base + k * 1024 + (int64_t)threadIdx.x * 4, n, thr));
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
}

// Exclusive scan of the per-chunk counts by one 1024-thread workgroup (the list is short: 32 Ki entries for a 512^3 grid): per
// tile of 1024 counts, a shuffle scan inside each wavefront, a shuffle scan of the sixteen wavefront totals, two barriers
// (a Hillis-Steele scan over the whole tile took twenty: 61 us per call at 512^3).  A tile sums to <= 2^22.
__global__ __launch_bounds__(1024) void scan_counts(const uint32_t *counts, int64_t n_chunks,
                                                    int64_t *offsets, int64_t *total)
{
    __shared__ uint32_t wave_total[16];
    __shared__ uint32_t tile_total;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int64_t running = 0;                       // the same value in every thread
    for (int64_t base = 0; base < n_chunks; base += 1024) {
        const int64_t i = base + threadIdx.x;
        const uint32_t c = i < n_chunks ? counts[i] : 0u;
        uint32_t incl = c;                     // inclusive scan inside the wavefront
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        if (lane == 63) wave_total[wave] = incl;
        __syncthreads();
        if (wave == 0) {                       // exclusive scan of the sixteen totals, by the first wavefront
            const uint32_t t = lane < 16 ? wave_total[lane] : 0u;
            uint32_t ti = t;
            for (int off = 1; off < 16; off <<= 1) {
                const uint32_t up = __shfl_up(ti, off);
                if (lane >= off) ti += up;
            }
            if (lane < 16) wave_total[lane] = ti - t;
            if (lane == 15) tile_total = ti;
        }
        __syncthreads();
        if (i < n_chunks) offsets[i] = running + (int64_t)(wave_total[wave] + incl - c);
        running += (int64_t)tile_total;
        __syncthreads();                       // the totals are rewritten by the next tile
    }
    if (threadIdx.x == 0) *total = running;
}

__global__ __launch_bounds__(256) void surface_emit(const float *tsdf, const float *weight, int64_t n,
                                                    float thr, const int64_t *offsets, int dim_x,
                                                    int dim_y, int z_begin, float ox, float oy,
                                                    float oz, float vs, float *xyz)
{
    __shared__ uint32_t cnt[kQuadPasses * 4];  // [pass][wave] in output order
    const int64_t base = (int64_t)blockIdx.x * kChunk;
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    uint32_t flags = 0;                         // four bits per pass
#pragma unroll
    for (int k = 0; k < kQuadPasses; ++k) {
        const uint32_t q = surface_quad(tsdf, weight, base + k * 1024 + (int64_t)threadIdx.x * 4, n, thr);
        flags |= q << (4 * k);
        uint32_t c = (uint32_t)__popc(q);
        for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
        if (lane == 0) cnt[k * 4 + wave] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // exclusive scan in output order: pass, then wavefront (a wavefront's 256 voxels are consecutive)
        uint32_t run = 0;
        for (int j = 0; j < kQuadPasses * 4; ++j) { uint32_t c = cnt[j]; cnt[j] = run; run += c; }
    }
    __syncthreads();
    // (no lane leaves before the shuffle scans below: every lane of a wavefront takes part in them)
    const int64_t chunk_off = offsets[blockIdx.x];
    const int64_t slice = (int64_t)dim_x * dim_y;
#pragma unroll
    for (int k = 0; k < kQuadPasses; ++k) {
        const uint32_t q = (flags >> (4 * k)) & 15u;
        // points of the lanes below in this pass: an inclusive shuffle scan of the per-lane counts, minus the own
        const uint32_t mine = (uint32_t)__popc(q);
        uint32_t incl = mine;
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        if (q == 0u) continue;
        int64_t pos = chunk_off + cnt[k * 4 + wave] + (incl - mine);
        const int64_t i0 = base + k * 1024 + (int64_t)threadIdx.x * 4;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (!((q >> j) & 1u)) continue;
            const int64_t i = i0 + j;
            const int lz = (int)(i / slice);
            const int rem = (int)(i - (int64_t)lz * slice);
            const int y = rem / dim_x;
            const int x = rem - y * dim_x;
            // ref: src/tsdf.cu:206-208
            xyz[3 * pos + 0] = ox + (float)x * vs;
            xyz[3 * pos + 1] = oy + (float)y * vs;
            xyz[3 * pos + 2] = oz + (float)(z_begin + lz) * vs;
            ++pos;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Zero-crossing surface vertices (the vertex set of marching cubes): for voxel v and axis a with
// neighbour n = v + e_a, both weights > thr and tsdf(v), tsdf(n) of opposite sign -> the point
// p(v) + (tsdf(v) / (tsdf(v) - tsdf(n))) * voxel_size * e_a.  Not in the reference (its mesh path
// is the absent Python package); the rule is this project's own and is checked bit for bit against
// its CPU restatement in the test suite (tests/test_gpu_crossings.py).  The +z neighbours of a slab's top
// slice come from `halo_*` -- the one-voxel halo a z-slab rank receives from its upper
// neighbour (RCCL send/recv, semantic_slam_amd/sharded.py) -- or are skipped when it is null.
// Same three-pass order-preserving compaction as the point extractor; a voxel emits up to three
// points, in x, y, z edge order.
// ------------------------------------------------------------------------------------------
struct CrossingGrid {
    const float *tsdf, *weight, *halo_tsdf, *halo_weight;
    int64_t n;          // voxels in the slab
    int dim_x, dim_y, nz, z_begin;
    float thr, ox, oy, oz, vs;
    // the free-space summary of the Integrate kernels (one word per 256-voxel row segment, bit 0 = "every TSDF value in here is
    // exactly 1.0"; rows of nseg segments) or nullptr / 0 when the grid has no row segments
    const uint32_t *flags;
    int nseg;
};

// The 256 voxels starting at linear index seg * 256 and every voxel they are compared with -- the segment's +x, +y, +z
// neighbours (and with `cube` the +xy, +xz, +yz, +xyz ones) -- are known to hold the value 1.0: no edge changes sign and no cube
// is cut, so the workgroup skips the segment without reading it.  Most of a fused volume is free or unseen space; the
// summary is kept exact in that direction by every kernel that writes TSDF values (a cleared bit only costs the reading).
// Segments of the slab's top slice depend on the halo slice, which has no summary: they are always read.
__device__ __forceinline__ bool segment_is_flat(const CrossingGrid &g, const int64_t seg, const bool cube)
{
    if (g.flags == nullptr || g.nseg <= 0) return false;
    const int64_t row = seg / g.nseg;                        // lz * dim_y + y
    const int sx = (int)(seg - row * g.nseg);
    const int lz = (int)(row / g.dim_y), y = (int)(row - (int64_t)lz * g.dim_y);
    if (lz >= g.nz || lz + 1 >= g.nz) return false;          // past the slab, or its top slice
    const bool has_x = sx + 1 < g.nseg, has_y = y + 1 < g.dim_y;
    const int64_t dy = g.nseg, dz = (int64_t)g.dim_y * g.nseg;
    uint32_t all = g.flags[seg] & g.flags[seg + dz];
    if (has_x) all &= g.flags[seg + 1] & g.flags[seg + dz + 1];
    if (has_y) all &= g.flags[seg + dy] & g.flags[seg + dz + dy];
    if (cube && has_x && has_y) all &= g.flags[seg + dy + 1] & g.flags[seg + dz + dy + 1];
    return (all & 1u) != 0u;
}

// bit k set = segment k of workgroup `chunk` is flat (lane k of every wavefront looks its segment up: one round trip for all 16)
__device__ __forceinline__ uint32_t flat_segments(const CrossingGrid &g, const int64_t chunk, const bool cube)
{
    const int lane = threadIdx.x & 63;
    const bool f = lane < kPerThread && segment_is_flat(g, chunk * kPerThread + lane, cube);
    return (uint32_t)__ballot(f);
}

// bit a set = the edge from voxel i along axis a crosses zero
__device__ __forceinline__ uint32_t crossing_bits(const CrossingGrid &g, int64_t i, float &t0, float t1[3])
{
    if (i >= g.n) return 0u;
    t0 = g.tsdf[i];
    if (!(g.weight[i] > g.thr)) return 0u;
    const int64_t slice = (int64_t)g.dim_x * g.dim_y;
    const int lz = (int)(i / slice);
    const int rem = (int)(i - (int64_t)lz * slice);
    const int y = rem / g.dim_x, x = rem - y * g.dim_x;
    uint32_t bits = 0u;
    if (x + 1 < g.dim_x) {
        t1[0] = g.tsdf[i + 1];
        if (g.weight[i + 1] > g.thr && ((t0 < 0.0f) != (t1[0] < 0.0f))) bits |= 1u;
    }
    if (y + 1 < g.dim_y) {
        t1[1] = g.tsdf[i + g.dim_x];
        if (g.weight[i + g.dim_x] > g.thr && ((t0 < 0.0f) != (t1[1] < 0.0f))) bits |= 2u;
    }
    if (lz + 1 < g.nz) {
        t1[2] = g.tsdf[i + slice];
        if (g.weight[i + slice] > g.thr && ((t0 < 0.0f) != (t1[2] < 0.0f))) bits |= 4u;
    } else if (g.halo_tsdf != nullptr) {
        t1[2] = g.halo_tsdf[rem];
        if (g.halo_weight[rem] > g.thr && ((t0 < 0.0f) != (t1[2] < 0.0f))) bits |= 4u;
    }
    return bits;
}

__global__ __launch_bounds__(256) void crossing_count(CrossingGrid g, uint32_t *counts)
{
    __shared__ uint32_t wave_sum[4];
    const int64_t base = (int64_t)blockIdx.x * kChunk;
    uint32_t c = 0;
    const uint32_t flat = flat_segments(g, blockIdx.x, false);
    for (int k = 0; k < kPerThread; ++k) {
        if ((flat >> k) & 1u) continue;
        float t0, t1[3];
        c += (uint32_t)__popc(crossing_bits(g, base + k * 256 + threadIdx.x, t0, t1));
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
}

__global__ __launch_bounds__(256) void crossing_emit(CrossingGrid g, const int64_t *offsets, float *xyz)
{
    __shared__ uint32_t cnt[kPerThread * 4];  // [k][wave] in output order
    const int64_t base = (int64_t)blockIdx.x * kChunk;
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const uint32_t flat = flat_segments(g, blockIdx.x, false);      // bit k: segment k of the chunk holds nothing (wave-uniform)
    for (int k = 0; k < kPerThread; ++k) {
        if ((flat >> k) & 1u) {
            if (lane == 0) cnt[k * 4 + wave] = 0u;
            continue;
        }
        float t0, t1[3];
        uint32_t c = (uint32_t)__popc(crossing_bits(g, base + k * 256 + threadIdx.x, t0, t1));
        for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
        if (lane == 0) cnt[k * 4 + wave] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int j = 0; j < kPerThread * 4; ++j) { uint32_t c = cnt[j]; cnt[j] = run; run += c; }
    }
    __syncthreads();
    const int64_t chunk_off = offsets[blockIdx.x];
    const int64_t slice = (int64_t)g.dim_x * g.dim_y;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int k = 0; k < kPerThread; ++k) {
        if ((flat >> k) & 1u) continue;
        const int64_t i = base + k * 256 + threadIdx.x;
        float t0, t1[3];
        const uint32_t bits = crossing_bits(g, i, t0, t1);
        // points emitted by lower lanes of this wavefront for this k (all three axes), then own axes in order
        const unsigned long long bx = __ballot(bits & 1u), by = __ballot(bits & 2u), bz = __ballot(bits & 4u);
        if (bits == 0u) continue;
        int64_t pos = chunk_off + cnt[k * 4 + wave] + __popcll(bx & lt) + __popcll(by & lt) + __popcll(bz & lt);
        const int lz = (int)(i / slice);
        const int rem = (int)(i - (int64_t)lz * slice);
        const int y = rem / g.dim_x, x = rem - y * g.dim_x;
        const float px = g.ox + (float)x * g.vs;
        const float py = g.oy + (float)y * g.vs;
        const float pz = g.oz + (float)(g.z_begin + lz) * g.vs;
        for (int a = 0; a < 3; ++a) {
            if (!(bits & (1u << a))) continue;
            const float s = t0 / (t0 - t1[a]);
            const float d = s * g.vs;
            xyz[3 * pos + 0] = a == 0 ? px + d : px;
            xyz[3 * pos + 1] = a == 1 ? py + d : py;
            xyz[3 * pos + 2] = a == 2 ? pz + d : pz;
            ++pos;
        }
    }
}

// ------------------------------------------------------------------------------------------
// Triangle mesh by marching tetrahedra.  Not in the reference (its SaveMesh needs the absent Python
// package, ref: src/TSDFfusion.py.in:48-53); the rule is this project's own and is checked bit for bit
// against its CPU restatement in the test suite (tests/test_gpu_mesh.py):
//   every cube with base voxel (x, y, z) whose 8 corners c = dx + 2 dy + 4 dz all have weight > thr is cut
//   into the 6 tetrahedra around the diagonal 0-7; inside = tsdf < 0; 1 or 3 corners inside -> one
//   triangle, 2 -> two; an edge vertex between cube corners i < j is p_i + s (p_j - p_i) with
//   s = t_i / (t_i - t_j) (lower corner first: cubes sharing an edge produce the same bits, so the soup is
//   a watertight 2-manifold); winding so that the normal points away from the inside corner.
// Cubes of a slab's top slice take their upper corners from the halo slice (RCCL z halo), or are skipped.
// Output: 9 floats per triangle, cubes in grid order -- same three-pass compaction as above, ranks inside
// a wavefront by a shuffle scan because a cube emits 0..12 triangles.
// ------------------------------------------------------------------------------------------
__constant__ int kTet[6][4] = {{0, 1, 3, 7}, {0, 3, 2, 7}, {0, 2, 6, 7}, {0, 6, 4, 7}, {0, 4, 5, 7}, {0, 5, 1, 7}};

// Loads the 8 corner values of the cube based at slab-local linear index i; false if it is not a cube
// of this slab (border voxel, missing halo, unobserved corner).
__device__ __forceinline__ bool load_cube(const CrossingGrid &g, int64_t i, float t[8], int &x, int &y, int &lz)
{
    if (i >= g.n) return false;
    const int64_t slice = (int64_t)g.dim_x * g.dim_y;
    lz = (int)(i / slice);
    const int rem = (int)(i - (int64_t)lz * slice);
    y = rem / g.dim_x;
    x = rem - y * g.dim_x;
    if (x + 1 >= g.dim_x || y + 1 >= g.dim_y) return false;
    const bool upper_in_slab = lz + 1 < g.nz;
    if (!upper_in_slab && g.halo_tsdf == nullptr) return false;
    bool ok = true;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const int64_t off = (int64_t)(c & 1) + (int64_t)((c >> 1) & 1) * g.dim_x;
        float tv, wv;
        if (!(c >> 2) || upper_in_slab) {
            const int64_t j = i + off + (int64_t)(c >> 2) * slice;
            tv = g.tsdf[j]; wv = g.weight[j];
        } else {
            tv = g.halo_tsdf[rem + off]; wv = g.halo_weight[rem + off];
        }
        ok = ok && (wv > g.thr);
        t[c] = tv;
    }
    return ok;
}

__device__ __forceinline__ uint32_t cube_triangle_count(const float t[8])
{
    uint32_t n = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        int cnt = 0;
#pragma unroll
        for (int a = 0; a < 4; ++a) cnt += t[kTet[k][a]] < 0.0f ? 1 : 0;
        n += (cnt == 1 || cnt == 3) ? 1u : (cnt == 2 ? 2u : 0u);
    }
    return n;
}

__device__ __forceinline__ void mesh_edge(const float p[8][3], const float t[8], int a, int b, float out[3])
{
    const int i = a < b ? a : b, j = a < b ? b : a;
    const float s = t[i] / (t[i] - t[j]);
#pragma unroll
    for (int k = 0; k < 3; ++k) out[k] = p[i][k] + s * (p[j][k] - p[i][k]);
}

__device__ __forceinline__ void mesh_emit(const float P0[3], const float P1[3], const float P2[3], const float q[3],
                                          float *o)
{
    const float e1[3] = {P1[0] - P0[0], P1[1] - P0[1], P1[2] - P0[2]};
    const float e2[3] = {P2[0] - P0[0], P2[1] - P0[1], P2[2] - P0[2]};
    const float nx = e1[1] * e2[2] - e1[2] * e2[1];
    const float ny = e1[2] * e2[0] - e1[0] * e2[2];
    const float nz = e1[0] * e2[1] - e1[1] * e2[0];
    const float d = nx * (P0[0] - q[0]) + ny * (P0[1] - q[1]) + nz * (P0[2] - q[2]);
    const bool keep = d >= 0.0f;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o[k] = P0[k];
        o[3 + k] = keep ? P1[k] : P2[k];
        o[6 + k] = keep ? P2[k] : P1[k];
    }
}

__global__ __launch_bounds__(256) void mesh_count(CrossingGrid g, uint32_t *counts)
{
    __shared__ uint32_t wave_sum[4];
    const int64_t base = (int64_t)blockIdx.x * kChunk;
    uint32_t c = 0;
    const uint32_t flat = flat_segments(g, blockIdx.x, true);
    for (int k = 0; k < kPerThread; ++k) {
        if ((flat >> k) & 1u) continue;
        float t[8];
        int x, y, lz;
        if (load_cube(g, base + k * 256 + threadIdx.x, t, x, y, lz)) c += cube_triangle_count(t);
    }
    for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
}

__global__ __launch_bounds__(256) void mesh_emit_kernel(CrossingGrid g, const int64_t *offsets, float *tri)
{
    __shared__ uint32_t cnt[kPerThread * 4];  // [k][wave] in output order
    const int64_t base = (int64_t)blockIdx.x * kChunk;
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    const uint32_t flat = flat_segments(g, blockIdx.x, true);       // bit k: segment k of the chunk holds nothing (wave-uniform)
    for (int k = 0; k < kPerThread; ++k) {
        if ((flat >> k) & 1u) {
            if (lane == 0) cnt[k * 4 + wave] = 0u;
            continue;
        }
        float t[8];
        int x, y, lz;
        uint32_t c = load_cube(g, base + k * 256 + threadIdx.x, t, x, y, lz) ? cube_triangle_count(t) : 0u;
        for (int off = 32; off > 0; off >>= 1) c += __shfl_down(c, off);
        if (lane == 0) cnt[k * 4 + wave] = c;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int j = 0; j < kPerThread * 4; ++j) { uint32_t c = cnt[j]; cnt[j] = run; run += c; }
    }
    __syncthreads();
    const int64_t chunk_off = offsets[blockIdx.x];
    for (int k = 0; k < kPerThread; ++k) {
        if ((flat >> k) & 1u) continue;
        float t[8];
        int x, y, lz;
        const bool ok = load_cube(g, base + k * 256 + threadIdx.x, t, x, y, lz);
        const uint32_t mine = ok ? cube_triangle_count(t) : 0u;
        uint32_t incl = mine;                       // inclusive scan over the wavefront
        for (int off = 1; off < 64; off <<= 1) {
            const uint32_t up = __shfl_up(incl, off);
            if (lane >= off) incl += up;
        }
        if (mine == 0u) continue;
        int64_t pos = chunk_off + cnt[k * 4 + wave] + (incl - mine);
        float p[8][3];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            p[c][0] = g.ox + (float)(x + (c & 1)) * g.vs;
            p[c][1] = g.oy + (float)(y + ((c >> 1) & 1)) * g.vs;
            p[c][2] = g.oz + (float)(g.z_begin + lz + (c >> 2)) * g.vs;
        }
        for (int kt = 0; kt < 6; ++kt) {
            int v[4], in[4], n_in = 0;
            for (int a = 0; a < 4; ++a) { v[a] = kTet[kt][a]; in[a] = t[v[a]] < 0.0f ? 1 : 0; n_in += in[a]; }
            if (n_in == 0 || n_in == 4) continue;
            int first_in = 0;
            while (!in[first_in]) ++first_in;
            const float *q = p[v[first_in]];
            float E[4][3];
            if (n_in == 1 || n_in == 3) {
                int odd = 0;
                for (int a = 0; a < 4; ++a) if (in[a] == (n_in == 1 ? 1 : 0)) odd = a;
                int m = 0;
                for (int a = 0; a < 4; ++a) if (a != odd) mesh_edge(p, t, v[odd], v[a], E[m++]);
                mesh_emit(E[0], E[1], E[2], q, tri + 9 * pos);
                ++pos;
            } else {
                int A = -1, B = -1, C = -1, D = -1;
                for (int a = 0; a < 4; ++a) {
                    if (in[a]) { if (A < 0) A = a; else B = a; }
                    else { if (C < 0) C = a; else D = a; }
                }
                mesh_edge(p, t, v[A], v[C], E[0]);
                mesh_edge(p, t, v[A], v[D], E[1]);
                mesh_edge(p, t, v[B], v[D], E[2]);
                mesh_edge(p, t, v[B], v[C], E[3]);
                mesh_emit(E[0], E[1], E[2], q, tri + 9 * pos);
                mesh_emit(E[0], E[2], E[3], q, tri + 9 * (pos + 1));
                pos += 2;
            }
        }
    }
}

}  // namespace tsdfx
