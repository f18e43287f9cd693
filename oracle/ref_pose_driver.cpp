/*
 * ref_pose_driver.cpp -- C entry points over the reference's own 4x4 helpers, compiled by `make ref_pose`.
 *
 * TEST INFRASTRUCTURE, never linked into the product.  TSDF::multiply_matrix (src/tsdf.cu:253-273) and
 * TSDF::invert_matrix (src/tsdf.cu:276-403) are plain arithmetic on their arguments; the class they belong to
 * (include/tsdf.hpp:22-93) includes OpenCV and CUDA headers and cannot be compiled here.  `make -C oracle ref_pose`
 * streams the two definitions, from where they lie and as they stand, into g++ behind ref_pose_decl.h -- two prototypes
 * (the signatures of include/tsdf.hpp:88,91) inside a NAMESPACE called TSDF, so that `void TSDF::multiply_matrix(...) {`
 * in the reference's text is the definition of that namespace member.  Nothing of the function bodies is restated,
 * replaced or edited; what this project supplies is the scope they are declared in.  The result,
 * oracle/_ref/libtsdf_ref_pose.so, pins this project's two implementations (csrc/pose_math.h, oracle/tsdf_oracle.c) and the
 * pose composition of TSDF::TSDF / TSDF::Integrate (src/tsdf.cu:74,142: cam2base = invert(base2world) x cam2world) to the
 * reference's own arithmetic, bit for bit (tests/test_pose_math.py).
 */
#include "ref_pose_decl.h"

extern "C" void ref_multiply_matrix(const float *m1, const float *m2, float *out)
{
	TSDF::multiply_matrix(m1, m2, out);
}

extern "C" int ref_invert_matrix(const float *m, float *out)
{
	return TSDF::invert_matrix(m, out) ? 1 : 0;
}
