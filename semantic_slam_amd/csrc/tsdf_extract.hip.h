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

__device__ __forceinline__ bool is_surface(const float *tsdf, const float *weight, int64_t i,
                                           int64_t n, float thr)
{
    // ref: src/tsdf.cu:179 (tsdf_thresh is unused there too)
    return i < n && fabsf(tsdf[i]) != 0.0f && weight[i] > thr;
}

__global__ __launch_bounds__(256) void surface_count(const float *tsdf, const float *weight,
                                                     int64_t n, float thr, uint32_t *counts)
{
    __shared__ uint32_t wave_sum[4];
    const int64_t base = (int64_t)blockIdx.x * kChunk;
    uint32_t c = 0;
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {
        bool f = is_surface(tsdf, weight, base + k * 256 + threadIdx.x, n, thr);
        c += (uint32_t)__popcll(__ballot(f));
    }
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) counts[blockIdx.x] = wave_sum[0] + wave_sum[1] + wave_sum[2] + wave_sum[3];
}

// Exclusive scan of the per-chunk counts by one 1024-thread workgroup (the list is short:
// 32 Ki entries for a 512^3 grid).
__global__ __launch_bounds__(1024) void scan_counts(const uint32_t *counts, int64_t n_chunks,
                                                    int64_t *offsets, int64_t *total)
{
    __shared__ uint32_t buf[1024];
    __shared__ int64_t running;
    if (threadIdx.x == 0) running = 0;
    __syncthreads();
    for (int64_t base = 0; base < n_chunks; base += 1024) {
        int64_t i = base + threadIdx.x;
        uint32_t c = i < n_chunks ? counts[i] : 0u;
        buf[threadIdx.x] = c;
        __syncthreads();
        for (int d = 1; d < 1024; d <<= 1) {  // inclusive Hillis-Steele; a tile sums to <= 2^22
            uint32_t add = threadIdx.x >= (unsigned)d ? buf[threadIdx.x - d] : 0u;
            __syncthreads();
            buf[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < n_chunks) offsets[i] = running + (int64_t)(buf[threadIdx.x] - c);
        __syncthreads();
        if (threadIdx.x == 0) running += (int64_t)buf[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) *total = running;
}

__global__ __launch_bounds__(256) void surface_emit(const float *tsdf, const float *weight, int64_t n,
                                                    float thr, const int64_t *offsets, int dim_x,
                                                    int dim_y, int z_begin, float ox, float oy,
                                                    float oz, float vs, float *xyz)
{
    __shared__ uint32_t cnt[kPerThread * 4];  // [k][wave] in output order
    const int64_t base = (int64_t)blockIdx.x * kChunk;
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
    uint32_t flags = 0;
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {
        bool f = is_surface(tsdf, weight, base + k * 256 + threadIdx.x, n, thr);
        unsigned long long b = __ballot(f);
        flags |= (uint32_t)f << k;
        if (lane == 0) cnt[k * 4 + wave] = (uint32_t)__popcll(b);
    }
    __syncthreads();
    if (threadIdx.x == 0) {  // 64-entry exclusive scan
        uint32_t run = 0;
        for (int j = 0; j < kPerThread * 4; ++j) { uint32_t c = cnt[j]; cnt[j] = run; run += c; }
    }
    __syncthreads();
    if (flags == 0) return;
    const int64_t chunk_off = offsets[blockIdx.x];
    const int64_t slice = (int64_t)dim_x * dim_y;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
#pragma unroll
    for (int k = 0; k < kPerThread; ++k) {
        bool f = (flags >> k) & 1u;
        unsigned long long b = __ballot(f);
        if (f) {
            int64_t pos = chunk_off + cnt[k * 4 + wave] + __popcll(b & lt);
            int64_t i = base + k * 256 + threadIdx.x;
            int lz = (int)(i / slice);
            int rem = (int)(i - (int64_t)lz * slice);
            int y = rem / dim_x;
            int x = rem - y * dim_x;
            // ref: src/tsdf.cu:206-208
            xyz[3 * pos + 0] = ox + (float)x * vs;
            xyz[3 * pos + 1] = oy + (float)y * vs;
            xyz[3 * pos + 2] = oz + (float)(z_begin + lz) * vs;
        }
    }
}

}  // namespace tsdfx
