// asan_host.cpp -- the product's HOST-side arithmetic (no device code) compiled for the CPU sanitizer run:
// semantic_slam_amd/csrc/pose_math.h (4x4 multiply / cofactor inverse, ref: src/tsdf.cu:253-403) and
// semantic_slam_amd/csrc/host_derive.h (wavefront brick choice, guards and margins of the exact shortcuts) and
// semantic_slam_amd/csrc/host_copy.h (the caller's frame into the pinned ring: streaming stores), behind plain C entry points, linked with tsdf_oracle.c into oracle/_asan/liboracle_asan.so by `make -C oracle asan`
// (-fsanitize=address,undefined -fno-sanitize-recover=all).  tests/test_sanitizers.py runs the golden vectors, the pose
// known-answer tests, the writers and these entry points under it.  TEST INFRASTRUCTURE: the headers are
// the product's own files, compiled here a second time; nothing in the product loads this library.
#include "../semantic_slam_amd/csrc/host_derive.h"
#include "../semantic_slam_amd/csrc/pose_math.h"
#include "../semantic_slam_amd/csrc/host_copy.h"

extern "C" {

void asan_multiply_matrix(const float *a, const float *b, float *out) { tsdf_host::multiply_matrix(a, b, out); }

int asan_invert_matrix(const float *m, float *out) { return tsdf_host::invert_matrix(m, out) ? 1 : 0; }

int asan_brick_shape_ok(const tsdf_config *c, int q, int r, int s) { return tsdf_host::brick_shape_ok(*c, q, r, s) ? 1 : 0; }

void asan_default_brick_shape(const tsdf_config *c, int32_t out[3])
{
    int q = 0, r = 0, s = 0;
    tsdf_host::choose_brick_default(*c, q, r, s);
    out[0] = q; out[1] = q ? r : 0; out[2] = q ? s : 0;     // as tsdf_default_brick_shape reports a grid without a brick view
}

// out: cz_margin, fast_ok, trunc_fast, cz_short, cz_pad, px_margin_u, px_margin_v
void asan_projection_guards(const tsdf_config *c, const float *cam2base, float out[7])
{
    const tsdf_host::ProjectionGuards g = tsdf_host::derive_projection_guards(*c, cam2base);
    out[0] = g.cz_margin; out[1] = (float)g.fast_ok; out[2] = (float)g.trunc_fast;
    out[3] = g.cz_short; out[4] = g.cz_pad; out[5] = g.px_margin_u; out[6] = g.px_margin_v;
}

// dst / src: any alignment, any size (the product hands it page-aligned ring slots and whole frames)
void asan_copy_to_pinned(void *dst, const void *src, size_t n) { tsdf_host::copy_to_pinned(dst, src, n); }

}  // extern "C"
