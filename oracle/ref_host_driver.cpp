/*
 * ref_host_driver.cpp -- C entry points over the reference's own host functions that compile as they stand (`make ref_host`).
 *
 * TEST INFRASTRUCTURE, never linked into the product.  TSDF::multiply_matrix (src/tsdf.cu:253-273), TSDF::invert_matrix
 * (src/tsdf.cu:276-403) and TSDF::SaveVoxelGrid2SurfacePointCloud (src/tsdf.cu:170-218) use nothing but their arguments and
 * the C / C++ standard library; the class they belong to (include/tsdf.hpp:22-93) includes OpenCV and CUDA headers and cannot
 * be compiled here.  `make -C oracle ref_host` streams the three definitions, from where they lie and as they stand, into g++
 * behind ref_host_decl.h -- the standard headers they use and their three prototypes inside a NAMESPACE called TSDF, so that
 * `void TSDF::multiply_matrix(...) {` in the reference's text is the definition of that namespace member.  Nothing of the
 * function bodies is restated, replaced or edited; what this project supplies is the scope they are declared in.  The result,
 * oracle/_ref/libtsdf_ref_host.so, pins
 *   - this project's 4x4 helpers (csrc/pose_math.h, oracle/tsdf_oracle.c) and the pose composition of TSDF::TSDF /
 *     TSDF::Integrate (src/tsdf.cu:74,142: cam2base = invert(base2world) x cam2world), bit for bit (tests/test_pose_math.py);
 *   - the surface rule, point order and bytes of tsdf<id>.ply as ~TSDF writes it (src/tsdf.cu:110-112: thresholds 1.2f, 0.9f),
 *     byte for byte (tests/test_writers_and_adapters.py, tests/test_gpu_dropin.py).
 * Not covered: the .bin dump and the constructor / Integrate glue, which are written inside member functions that touch the
 * class's fields (src/tsdf.cu:62-96, 114-132, 135-168).
 */
#include "ref_host_decl.h"

extern "C" void ref_multiply_matrix(const float *m1, const float *m2, float *out)
{
	TSDF::multiply_matrix(m1, m2, out);
}

extern "C" int ref_invert_matrix(const float *m, float *out)
{
	return TSDF::invert_matrix(m, out) ? 1 : 0;
}

/* the call ~TSDF makes (src/tsdf.cu:110-112) */
extern "C" void ref_save_ply(const char *path, int dim_x, int dim_y, int dim_z, float voxel_size, float ox, float oy, float oz,
                             float *tsdf, float *weight, float tsdf_thresh, float weight_thresh)
{
	TSDF::SaveVoxelGrid2SurfacePointCloud(std::string(path), dim_x, dim_y, dim_z, voxel_size, ox, oy, oz, tsdf, weight, tsdf_thresh,
	                                      weight_thresh);
}
