/* ref_host_decl.h -- see ref_host_driver.cpp.  The standard headers the three functions use (the reference reaches them
 * through include/tsdf.hpp:12-20) and their signatures (include/tsdf.hpp:74-77, :88, :91) at namespace scope. */
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>

namespace TSDF {
void multiply_matrix(const float m1[16], const float m2[16], float mOut[16]);
bool invert_matrix(const float m[16], float invOut[16]);
void SaveVoxelGrid2SurfacePointCloud(const std::string &file_name, int voxel_grid_dim_x, int voxel_grid_dim_y, int voxel_grid_dim_z,
                                     float voxel_size, float voxel_grid_origin_x, float voxel_grid_origin_y, float voxel_grid_origin_z,
                                     float *voxel_grid_TSDF, float *voxel_grid_weight, float tsdf_thresh, float weight_thresh);
}
