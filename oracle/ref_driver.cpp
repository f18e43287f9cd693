/*
 * ref_driver.cpp -- host launcher for the reference kernel body built by `make ref`.
 *
 * TEST INFRASTRUCTURE.  Replays the reference's launch shape <<<dim_z, dim_y>>>
 * (src/tsdf.cu:165: one block per z, one thread per y, each looping over x) as two host
 * loops, optionally spread over OpenMP threads by z (voxels are independent).
 */
#include "ref_shim.h"
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif

thread_local ref_idx3 blockIdx, threadIdx;

/* defined by the sliced reference translation unit (signature of src/tsdf.cu:16-19) */
void GpuIntegrate(float *cam_K, float *cam2base, float *depth_im, int im_height, int im_width,
                  int voxel_grid_dim_x, int voxel_grid_dim_y, int voxel_grid_dim_z,
                  float voxel_grid_origin_x, float voxel_grid_origin_y, float voxel_grid_origin_z,
                  float voxel_size, float trunc_margin, float *voxel_grid_TSDF,
                  float *voxel_grid_weight);

extern "C" void ref_integrate(float *cam_K, float *cam2base, float *depth_im, int im_height,
                              int im_width, int dim_x, int dim_y, int dim_z, float origin_x,
                              float origin_y, float origin_z, float voxel_size,
                              float trunc_margin, float *tsdf, float *weight, int n_threads)
{
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#pragma omp parallel for collapse(2) schedule(static) num_threads(n_threads)
#endif
    for (int z = 0; z < dim_z; ++z) {
        for (int y = 0; y < dim_y; ++y) {
            blockIdx.x = z;   /* one reference "thread" (z, y) at a time per host thread */
            threadIdx.x = y;
            GpuIntegrate(cam_K, cam2base, depth_im, im_height, im_width, dim_x, dim_y, dim_z,
                         origin_x, origin_y, origin_z, voxel_size, trunc_margin, tsdf, weight);
        }
    }
}

extern "C" int ref_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
