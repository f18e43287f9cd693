/*
 * ref_hip_launch.hip -- launcher for the reference's own kernel compiled for gfx950 by `make ref_hip`.
 *
 * TEST INFRASTRUCTURE, never linked into the product.  GpuIntegrate (the reference's only __global__,
 * src/tsdf.cu:15-60) is valid HIP as it stands: hipcc itself supplies __global__, blockIdx, threadIdx,
 * roundf and fmin.  `make -C oracle ref_hip` slices that function, from where it lies, into a temporary file outside the
 * repository (hipcc reads a translation unit twice, so it cannot come from a pipe), compiles it with
 * `-include hip/hip_runtime.h` -- the toolchain's own header, as nvcc force-includes cuda_runtime.h into a .cu file; nothing
 * written by this project, nothing substituted (oracle/Makefile) -- deletes the file, and links the object with this file, which
 * only declares the kernel and launches it with the reference's own shape <<<dim_z, dim_y>>>
 * (src/tsdf.cu:165: one block per z, one thread per y, each looping over x).  The result,
 * oracle/_ref/libtsdf_ref_hip.so, is "the reference compiled here": tests/test_gpu_ref_kernel.py runs it on
 * the GPU next to the CPU restatement and the product kernels.
 *
 * All pointers are device pointers (the reference keeps K and the pose in device memory too,
 * src/tsdf.cu:90-95,161).
 */
#include <hip/hip_runtime.h>

/* signature of src/tsdf.cu:16-19 */
__global__ void GpuIntegrate(float *cam_K, float *cam2base, float *depth_im, int im_height, int im_width,
                             int voxel_grid_dim_x, int voxel_grid_dim_y, int voxel_grid_dim_z,
                             float voxel_grid_origin_x, float voxel_grid_origin_y, float voxel_grid_origin_z,
                             float voxel_size, float trunc_margin, float *voxel_grid_TSDF,
                             float *voxel_grid_weight);

extern "C" int ref_hip_integrate(float *cam_K_dev, float *cam2base_dev, float *depth_dev, int im_height,
                                 int im_width, int dim_x, int dim_y, int dim_z, float origin_x, float origin_y,
                                 float origin_z, float voxel_size, float trunc_margin, float *tsdf_dev,
                                 float *weight_dev)
{
    /* the reference's block size is dim_y (src/tsdf.cu:165): its own hard limit */
    if (dim_y < 1 || dim_y > 1024 || dim_z < 1 || dim_x < 1) return -1;
    GpuIntegrate<<<dim_z, dim_y>>>(cam_K_dev, cam2base_dev, depth_dev, im_height, im_width, dim_x, dim_y, dim_z,
                                   origin_x, origin_y, origin_z, voxel_size, trunc_margin, tsdf_dev, weight_dev);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    return e == hipSuccess ? 0 : -(int)e - 1000;
}
