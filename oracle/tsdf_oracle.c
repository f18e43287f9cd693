/*
 * tsdf_oracle.c -- CPU restatement of the reference's dense-grid TSDF path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it; the shipped library (libtsdf_hip.so) never
 * links, loads or calls anything in oracle/.
 *
 * Every function restates, in this project's own words, one function of
 * /root/reference (Tariq-Abuhashim/semantic-slam) and cites the lines it follows.
 * Arithmetic is IEEE fp32 in the reference's operation order; build with
 * -ffp-contract=off (see oracle/Makefile) so no multiply-add is fused.
 *
 * Pinning: the reference has no tests or golden vectors for this path.
 * The voxel update is pinned bit-for-bit against the reference's own kernel body
 * (src/tsdf.cu:15-60): compiled by hipcc for gfx950 exactly as it stands and run on the device
 * (`make -C oracle ref_hip`, tests/test_gpu_ref_kernel.py), compiled for the host from where it lies
 * (`make -C oracle ref`, tests/test_oracle_vs_ref.py), and through the fixtures under tests/golden/ that
 * were generated from that build (tests/golden/make_golden.py).  The 4x4 helpers and the .ply writer are
 * pinned against the reference's own multiply_matrix / invert_matrix / SaveVoxelGrid2SurfacePointCloud
 * (src/tsdf.cu:253-403, :170-218) compiled for the host as they stand (`make -C oracle ref_host`,
 * tests/test_pose_math.py, tests/test_writers_and_adapters.py).  The constructor / Integrate glue and the
 * .bin dump inside ~TSDF touch the class's fields (its header needs OpenCV + CUDA) and cannot be compiled:
 * restated from the text, known-answer tests only.  The label, colour, crossing and mesh rules are not
 * reference functions: parity unpinned, stated where each is defined.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------
 * Voxel update.  Follows GpuIntegrate, src/tsdf.cu:15-60.
 *
 * One call visits global z in [z_begin, z_end) of a dim_x*dim_y*dim_z grid (the
 * reference launches one block per z and one thread per y, each looping over x,
 * tsdf.cu:21-23,165); tsdf/weight point at the first voxel of slice z_begin, so a
 * z-slab owner passes its slab-local arrays.  max_depth is the literal 6 of tsdf.cu:46.
 * Returns the number of voxels whose weight changed.
 * ---------------------------------------------------------------------------------- */
int64_t oracle_integrate(const float *cam_K, const float *cam2base, const float *depth_im,
                         int im_height, int im_width,
                         int dim_x, int dim_y, int dim_z, int z_begin, int z_end,
                         float origin_x, float origin_y, float origin_z,
                         float voxel_size, float trunc_margin, float max_depth,
                         float *tsdf, float *weight, int n_threads)
{
    int64_t n_updated = 0;
    (void)dim_z;
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
    /* rows (z, y) are independent: collapse both loops so that a thin slab still feeds every core */
#pragma omp parallel for collapse(2) schedule(static) num_threads(n_threads) reduction(+ : n_updated)
#else
    (void)n_threads;
#endif
    for (int gz = z_begin; gz < z_end; ++gz) {
        for (int gy = 0; gy < dim_y; ++gy) {
            float *row_t = tsdf + ((int64_t)(gz - z_begin) * dim_y + gy) * dim_x;
            float *row_w = weight + ((int64_t)(gz - z_begin) * dim_y + gy) * dim_x;
            for (int gx = 0; gx < dim_x; ++gx) {
                /* voxel centre in the base camera frame: origin + index*size, tsdf.cu:27-29 */
                float bx = origin_x + (float)gx * voxel_size;
                float by = origin_y + (float)gy * voxel_size;
                float bz = origin_z + (float)gz * voxel_size;

                /* base frame -> current camera frame: subtract the translation column,
                 * then multiply by the transposed rotation, tsdf.cu:33-38 */
                float dx = bx - cam2base[3];
                float dy = by - cam2base[7];
                float dz = bz - cam2base[11];
                float cx = cam2base[0] * dx + cam2base[4] * dy + cam2base[8] * dz;
                float cy = cam2base[1] * dx + cam2base[5] * dy + cam2base[9] * dz;
                float cz = cam2base[2] * dx + cam2base[6] * dy + cam2base[10] * dz;
                if (cz <= 0.0f) continue; /* tsdf.cu:39 */

                /* pinhole projection, round half away from zero, tsdf.cu:41-42.
                 * The reference converts the rounded float to int before the bounds
                 * test; comparing the rounded float instead is the same test for every
                 * value an int can hold and stays defined for those it cannot
                 * (SURVEY.md section 8c caveat 2). */
                float pu = roundf(cam_K[0] * (cx / cz) + cam_K[2]);
                float pv = roundf(cam_K[4] * (cy / cz) + cam_K[5]);
                if (!(pu >= 0.0f && pu < (float)im_width && pv >= 0.0f && pv < (float)im_height))
                    continue; /* tsdf.cu:43 */
                int iu = (int)pu, iv = (int)pv;

                float d = depth_im[iv * im_width + iu]; /* tsdf.cu:45 */
                if (d <= 0.0f || d > max_depth) continue; /* tsdf.cu:46 */

                float diff = d - cz; /* tsdf.cu:48 */
                if (diff <= -trunc_margin) continue; /* tsdf.cu:49 */

                /* running weighted mean with observation weight 1, tsdf.cu:53-57 */
                float dist = fminf(1.0f, diff / trunc_margin);
                float w_old = row_w[gx];
                float w_new = w_old + 1.0f;
                row_w[gx] = w_new;
                row_t[gx] = (row_t[gx] * w_old + dist) / w_new;
                ++n_updated;
            }
        }
    }
    return n_updated;
}

/* Initial grid state: TSDF = 1, weight = 0.  Follows TSDF::TSDF, src/tsdf.cu:79-81. */
void oracle_init_grid(float *tsdf, float *weight, int64_t n)
{
    for (int64_t i = 0; i < n; ++i) tsdf[i] = 1.0f;
    memset(weight, 0, sizeof(float) * (size_t)n);
}

/* ------------------------------------------------------------------------------------
 * 4x4 row-major product.  Follows TSDF::multiply_matrix, src/tsdf.cu:253-273:
 * each entry is a*b + a*b + a*b + a*b, summed left to right in fp32.
 * ---------------------------------------------------------------------------------- */
void oracle_multiply_matrix(const float *a, const float *b, float *out)
{
    float tmp[16];
    for (int r = 0; r < 4; ++r)
        for (int c = 0; c < 4; ++c) {
            float s = a[4 * r + 0] * b[0 + c];
            s = s + a[4 * r + 1] * b[4 + c];
            s = s + a[4 * r + 2] * b[8 + c];
            s = s + a[4 * r + 3] * b[12 + c];
            tmp[4 * r + c] = s;
        }
    memcpy(out, tmp, sizeof tmp);
}

/* ------------------------------------------------------------------------------------
 * 4x4 row-major inverse by cofactors.  Follows TSDF::invert_matrix, src/tsdf.cu:276-403.
 * Each cofactor is six triple products combined left to right with the signs of the
 * reference (a leading minus negates the first factor, which is exact); the determinant
 * is m0*c0 + m1*c4 + m2*c8 + m3*c12 (tsdf.cu:392); its reciprocal is taken in double and
 * stored to float (tsdf.cu:397).  Returns 0 when det == 0 (tsdf.cu:394-395), 1 otherwise.
 * ---------------------------------------------------------------------------------- */
#define T3(a, b, c) (m[a] * m[b] * m[c])
int oracle_invert_matrix(const float *m, float *inv_out)
{
    float c[16];
    c[0]  =  T3(5, 10, 15) - T3(5, 11, 14) - T3(9, 6, 15) + T3(9, 7, 14) + T3(13, 6, 11) - T3(13, 7, 10);
    c[4]  = -m[4] * m[10] * m[15] + T3(4, 11, 14) + T3(8, 6, 15) - T3(8, 7, 14) - T3(12, 6, 11) + T3(12, 7, 10);
    c[8]  =  T3(4, 9, 15) - T3(4, 11, 13) - T3(8, 5, 15) + T3(8, 7, 13) + T3(12, 5, 11) - T3(12, 7, 9);
    c[12] = -m[4] * m[9] * m[14] + T3(4, 10, 13) + T3(8, 5, 14) - T3(8, 6, 13) - T3(12, 5, 10) + T3(12, 6, 9);
    c[1]  = -m[1] * m[10] * m[15] + T3(1, 11, 14) + T3(9, 2, 15) - T3(9, 3, 14) - T3(13, 2, 11) + T3(13, 3, 10);
    c[5]  =  T3(0, 10, 15) - T3(0, 11, 14) - T3(8, 2, 15) + T3(8, 3, 14) + T3(12, 2, 11) - T3(12, 3, 10);
    c[9]  = -m[0] * m[9] * m[15] + T3(0, 11, 13) + T3(8, 1, 15) - T3(8, 3, 13) - T3(12, 1, 11) + T3(12, 3, 9);
    c[13] =  T3(0, 9, 14) - T3(0, 10, 13) - T3(8, 1, 14) + T3(8, 2, 13) + T3(12, 1, 10) - T3(12, 2, 9);
    c[2]  =  T3(1, 6, 15) - T3(1, 7, 14) - T3(5, 2, 15) + T3(5, 3, 14) + T3(13, 2, 7) - T3(13, 3, 6);
    c[6]  = -m[0] * m[6] * m[15] + T3(0, 7, 14) + T3(4, 2, 15) - T3(4, 3, 14) - T3(12, 2, 7) + T3(12, 3, 6);
    c[10] =  T3(0, 5, 15) - T3(0, 7, 13) - T3(4, 1, 15) + T3(4, 3, 13) + T3(12, 1, 7) - T3(12, 3, 5);
    c[14] = -m[0] * m[5] * m[14] + T3(0, 6, 13) + T3(4, 1, 14) - T3(4, 2, 13) - T3(12, 1, 6) + T3(12, 2, 5);
    c[3]  = -m[1] * m[6] * m[11] + T3(1, 7, 10) + T3(5, 2, 11) - T3(5, 3, 10) - T3(9, 2, 7) + T3(9, 3, 6);
    c[7]  =  T3(0, 6, 11) - T3(0, 7, 10) - T3(4, 2, 11) + T3(4, 3, 10) + T3(8, 2, 7) - T3(8, 3, 6);
    c[11] = -m[0] * m[5] * m[11] + T3(0, 7, 9) + T3(4, 1, 11) - T3(4, 3, 9) - T3(8, 1, 7) + T3(8, 3, 5);
    c[15] =  T3(0, 5, 10) - T3(0, 6, 9) - T3(4, 1, 10) + T3(4, 2, 9) + T3(8, 1, 6) - T3(8, 2, 5);

    float det = m[0] * c[0] + m[1] * c[4] + m[2] * c[8] + m[3] * c[12];
    if (det == 0) return 0;
    det = (float)(1.0 / (double)det);
    for (int i = 0; i < 16; ++i) inv_out[i] = c[i] * det;
    return 1;
}
#undef T3

/* cam2base = inverse(base2world) * cam2world.  Follows TSDF::TSDF (tsdf.cu:74) and
 * TSDF::Integrate (tsdf.cu:139-142).  base2world_inv starts zeroed (tsdf.hpp:51) and a
 * failed inversion is ignored by the reference, which this keeps. */
void oracle_cam2base(const float *base2world, const float *cam2world, float *cam2base)
{
    float inv[16] = {0};
    oracle_invert_matrix(base2world, inv);
    oracle_multiply_matrix(inv, cam2world, cam2base);
}

/* ------------------------------------------------------------------------------------
 * Surface point extraction.  Follows TSDF::SaveVoxelGrid2SurfacePointCloud,
 * src/tsdf.cu:170-218: keep voxel i when fabs(tsdf) != 0 and weight > weight_thresh
 * (tsdf_thresh is accepted and unused, as in the reference); point = origin + index*size.
 * xyz may be NULL to count only.  Returns the number of points.
 * ---------------------------------------------------------------------------------- */
int64_t oracle_surface_points(const float *tsdf, const float *weight,
                              int dim_x, int dim_y, int dim_z, float voxel_size,
                              float origin_x, float origin_y, float origin_z,
                              float tsdf_thresh, float weight_thresh, float *xyz)
{
    (void)tsdf_thresh;
    int64_t n = 0;
    for (int z = 0; z < dim_z; ++z)
        for (int y = 0; y < dim_y; ++y)
            for (int x = 0; x < dim_x; ++x) {
                int64_t i = ((int64_t)z * dim_y + y) * dim_x + x;
                if (fabsf(tsdf[i]) != 0.0f && weight[i] > weight_thresh) {
                    if (xyz) {
                        xyz[3 * n + 0] = origin_x + (float)x * voxel_size; /* tsdf.cu:206-208 */
                        xyz[3 * n + 1] = origin_y + (float)y * voxel_size;
                        xyz[3 * n + 2] = origin_z + (float)z * voxel_size;
                    }
                    ++n;
                }
            }
    return n;
}

/* ------------------------------------------------------------------------------------
 * Zero-crossing surface vertices (NOT in the reference: its only extractor is the per-voxel
 * rule above; its mesh path lives in the absent tsdf-fusion-python, ref: src/TSDFfusion.py.in:48-53).
 * Definition used by this project, restated here for the HIP kernel to be checked against --
 * parity unpinned by any reference output:
 *   for every voxel v in grid order and every axis a in x, y, z order, with n = v + e_a inside the
 *   grid (for a = z the slice above the slab comes from `halo_*`, or the edge is skipped when they
 *   are NULL): if weight(v) > thresh and weight(n) > thresh and (tsdf(v) < 0) != (tsdf(n) < 0),
 *   emit  p(v) + s * voxel_size * e_a  with  s = tsdf(v) / (tsdf(v) - tsdf(n))  (fp32, one division),
 *   p(v) = origin + index * voxel_size as in tsdf.cu:206-208, and the a-coordinate computed as
 *   p_a(v) + s * voxel_size (one multiply, one add).
 * tsdf/weight hold slices [z_begin, z_end); halo_* hold slice z_end (dim_x*dim_y floats).
 * ---------------------------------------------------------------------------------- */
int64_t oracle_zero_crossings(const float *tsdf, const float *weight, const float *halo_tsdf,
                              const float *halo_weight, int dim_x, int dim_y, int z_begin, int z_end,
                              float voxel_size, float origin_x, float origin_y, float origin_z,
                              float weight_thresh, float *xyz)
{
    int64_t n = 0;
    const int64_t slice = (int64_t)dim_x * dim_y;
    for (int z = z_begin; z < z_end; ++z)
        for (int y = 0; y < dim_y; ++y)
            for (int x = 0; x < dim_x; ++x) {
                const int64_t i = (int64_t)(z - z_begin) * slice + (int64_t)y * dim_x + x;
                const float t0 = tsdf[i];
                if (!(weight[i] > weight_thresh)) continue;
                const float px = origin_x + (float)x * voxel_size;
                const float py = origin_y + (float)y * voxel_size;
                const float pz = origin_z + (float)z * voxel_size;
                for (int a = 0; a < 3; ++a) {
                    float t1, w1;
                    if (a == 0) { if (x + 1 >= dim_x) continue; t1 = tsdf[i + 1]; w1 = weight[i + 1]; }
                    else if (a == 1) { if (y + 1 >= dim_y) continue; t1 = tsdf[i + dim_x]; w1 = weight[i + dim_x]; }
                    else if (z + 1 < z_end) { t1 = tsdf[i + slice]; w1 = weight[i + slice]; }
                    else { if (!halo_tsdf || !halo_weight) continue; t1 = halo_tsdf[(int64_t)y * dim_x + x]; w1 = halo_weight[(int64_t)y * dim_x + x]; }
                    if (!(w1 > weight_thresh)) continue;
                    if ((t0 < 0.0f) == (t1 < 0.0f)) continue;
                    if (xyz) {
                        const float s = t0 / (t0 - t1);
                        const float d = s * voxel_size;
                        xyz[3 * n + 0] = a == 0 ? px + d : px;
                        xyz[3 * n + 1] = a == 1 ? py + d : py;
                        xyz[3 * n + 2] = a == 2 ? pz + d : pz;
                    }
                    ++n;
                }
            }
    return n;
}

/* ------------------------------------------------------------------------------------
 * Triangle mesh by marching tetrahedra (NOT in the reference: its SaveMesh goes through the absent
 * tsdf-fusion-python, ref: src/TSDFfusion.py.in:48-53).  Project-defined rule, restated here for the
 * HIP kernels to be checked against -- parity unpinned by any reference output:
 *   for every cube with base voxel v = (x, y, z) in grid order whose 8 corners c = dx + 2 dy + 4 dz all
 *   have weight > thresh, for each of the 6 tetrahedra {0,1,3,7} {0,3,2,7} {0,2,6,7} {0,6,4,7} {0,4,5,7}
 *   {0,5,1,7} in that order, with "inside" = tsdf < 0:
 *     1 or 3 corners inside : one triangle on the three edges that leave the odd corner, taken in
 *                             increasing corner position within the tetrahedron;
 *     2 inside (A < B), C < D outside : two triangles (AC, AD, BD) and (AC, BD, BC);
 *   an edge vertex between cube corners i < j is p_i + s (p_j - p_i), s = t_i / (t_i - t_j), per component
 *   (sub, mul, add), p = origin + index * voxel_size (tsdf.cu:206-208): lower corner first, so that
 *   cubes sharing an edge produce the same bits;
 *   orientation: with n = (P1 - P0) x (P2 - P0) and q = the first inside corner of the tetrahedron in
 *   its listed order, the triangle is emitted as (P0, P1, P2) if n . (P0 - p_q) >= 0, else (P0, P2, P1).
 * The slab holds slices [z_begin, z_end); cubes of its top slice take their upper corners from halo_*
 * (slice z_end) or are skipped when those are NULL.  tri: 9 floats per triangle, or NULL to count.
 * ---------------------------------------------------------------------------------- */
static const int kTet[6][4] = {{0, 1, 3, 7}, {0, 3, 2, 7}, {0, 2, 6, 7}, {0, 6, 4, 7}, {0, 4, 5, 7}, {0, 5, 1, 7}};

static void mesh_edge(const float p[8][3], const float t[8], int a, int b, float out[3])
{
    const int i = a < b ? a : b, j = a < b ? b : a;
    const float s = t[i] / (t[i] - t[j]);
    for (int k = 0; k < 3; ++k) out[k] = p[i][k] + s * (p[j][k] - p[i][k]);
}

static int64_t mesh_emit(const float P0[3], const float P1[3], const float P2[3], const float q[3], float *tri, int64_t n)
{
    if (tri) {
        const float e1[3] = {P1[0] - P0[0], P1[1] - P0[1], P1[2] - P0[2]};
        const float e2[3] = {P2[0] - P0[0], P2[1] - P0[1], P2[2] - P0[2]};
        const float nx = e1[1] * e2[2] - e1[2] * e2[1];
        const float ny = e1[2] * e2[0] - e1[0] * e2[2];
        const float nz = e1[0] * e2[1] - e1[1] * e2[0];
        const float d = nx * (P0[0] - q[0]) + ny * (P0[1] - q[1]) + nz * (P0[2] - q[2]);
        const float *A = P1, *B = P2;
        if (!(d >= 0.0f)) { A = P2; B = P1; }
        float *o = tri + 9 * n;
        for (int k = 0; k < 3; ++k) { o[k] = P0[k]; o[3 + k] = A[k]; o[6 + k] = B[k]; }
    }
    return n + 1;
}

int64_t oracle_mesh_triangles(const float *tsdf, const float *weight, const float *halo_tsdf,
                              const float *halo_weight, int dim_x, int dim_y, int z_begin, int z_end,
                              float voxel_size, float origin_x, float origin_y, float origin_z,
                              float weight_thresh, float *tri)
{
    int64_t n = 0;
    const int64_t slice = (int64_t)dim_x * dim_y;
    for (int z = z_begin; z < z_end; ++z) {
        const int upper_in_slab = z + 1 < z_end;
        if (!upper_in_slab && (!halo_tsdf || !halo_weight)) continue;
        for (int y = 0; y + 1 < dim_y; ++y)
            for (int x = 0; x + 1 < dim_x; ++x) {
                float t[8], p[8][3];
                int ok = 1;
                for (int c = 0; c < 8; ++c) {
                    const int cx = x + (c & 1), cy = y + ((c >> 1) & 1), cz = z + (c >> 2);
                    float tv, wv;
                    if (cz < z_end) {
                        const int64_t i = (int64_t)(cz - z_begin) * slice + (int64_t)cy * dim_x + cx;
                        tv = tsdf[i]; wv = weight[i];
                    } else {
                        tv = halo_tsdf[(int64_t)cy * dim_x + cx]; wv = halo_weight[(int64_t)cy * dim_x + cx];
                    }
                    if (!(wv > weight_thresh)) { ok = 0; break; }
                    t[c] = tv;
                    p[c][0] = origin_x + (float)cx * voxel_size;
                    p[c][1] = origin_y + (float)cy * voxel_size;
                    p[c][2] = origin_z + (float)cz * voxel_size;
                }
                if (!ok) continue;
                for (int k = 0; k < 6; ++k) {
                    const int *v = kTet[k];
                    int in[4], cnt = 0;
                    for (int a = 0; a < 4; ++a) { in[a] = t[v[a]] < 0.0f; cnt += in[a]; }
                    if (cnt == 0 || cnt == 4) continue;
                    int first_in = 0;
                    while (!in[first_in]) ++first_in;
                    const float *q = p[v[first_in]];
                    float E[4][3];
                    if (cnt == 1 || cnt == 3) {
                        int odd = 0;
                        for (int a = 0; a < 4; ++a) if (in[a] == (cnt == 1)) odd = a;
                        int m = 0;
                        for (int a = 0; a < 4; ++a) if (a != odd) mesh_edge(p, t, v[odd], v[a], E[m++]);
                        n = mesh_emit(E[0], E[1], E[2], q, tri, n);
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
                        n = mesh_emit(E[0], E[1], E[2], q, tri, n);
                        n = mesh_emit(E[0], E[2], E[3], q, tri, n);
                    }
                }
            }
    }
    return n;
}

/* .ply writer: header text of tsdf.cu:185-192, then 3 floats per point (tsdf.cu:210-212). */
int oracle_save_ply(const char *path, const float *tsdf, const float *weight,
                    int dim_x, int dim_y, int dim_z, float voxel_size,
                    float origin_x, float origin_y, float origin_z,
                    float tsdf_thresh, float weight_thresh)
{
    int64_t n = oracle_surface_points(tsdf, weight, dim_x, dim_y, dim_z, voxel_size, origin_x,
                                      origin_y, origin_z, tsdf_thresh, weight_thresh, NULL);
    float *xyz = (float *)malloc(sizeof(float) * 3 * (size_t)(n ? n : 1));
    if (!xyz) return -1;
    oracle_surface_points(tsdf, weight, dim_x, dim_y, dim_z, voxel_size, origin_x, origin_y,
                          origin_z, tsdf_thresh, weight_thresh, xyz);
    FILE *fp = fopen(path, "w");
    if (!fp) { free(xyz); return -1; }
    fprintf(fp, "ply\n");
    fprintf(fp, "format binary_little_endian 1.0\n");
    fprintf(fp, "element vertex %d\n", (int)n);
    fprintf(fp, "property float x\n");
    fprintf(fp, "property float y\n");
    fprintf(fp, "property float z\n");
    fprintf(fp, "end_header\n");
    fwrite(xyz, sizeof(float), 3 * (size_t)n, fp);
    fclose(fp);
    free(xyz);
    return 0;
}

/* .bin writer: 8-float header {dim_x, dim_y, dim_z, origin xyz, voxel_size, trunc} then the
 * TSDF floats; weights are not saved.  Follows TSDF::~TSDF, src/tsdf.cu:116-132. */
int oracle_save_bin(const char *path, const float *tsdf, int dim_x, int dim_y, int dim_z,
                    float origin_x, float origin_y, float origin_z,
                    float voxel_size, float trunc_margin)
{
    FILE *fp = fopen(path, "wb");
    if (!fp) return -1;
    float hdr[8] = {(float)dim_x, (float)dim_y, (float)dim_z, origin_x, origin_y, origin_z,
                    voxel_size, trunc_margin};
    fwrite(hdr, sizeof(float), 8, fp);
    fwrite(tsdf, sizeof(float), (size_t)dim_x * dim_y * dim_z, fp);
    fclose(fp);
    return 0;
}

/* ------------------------------------------------------------------------------------
 * Caller-side adapters.
 * ---------------------------------------------------------------------------------- */

/* TSDF origin of a new object: per-axis minimum over pixels with depth > 0 of the
 * back-projected point, starting from 1000.  Follows Object::Object, src/Object.cpp:37-49
 * (z is the raw depth value; x and y multiply by the reciprocal focal length). */
void oracle_object_origin(const float *depth, int rows, int cols, const float *K, float *origin)
{
    origin[0] = origin[1] = origin[2] = 1000.0f;
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {
            float z = depth[r * cols + c];
            if (z <= 0.0) continue;
            float x = ((float)c - K[2]) * z * (1.0f / K[0]);
            float y = ((float)r - K[5]) * z * (1.0f / K[4]);
            /* std::min(x, origin) = (origin < x) ? origin : x -- src/Object.cpp:45-47 */
            origin[0] = origin[0] < x ? origin[0] : x;
            origin[1] = origin[1] < y ? origin[1] : y;
            origin[2] = origin[2] < z ? origin[2] : z;
        }
}

/* Depth pre-processing of the offline labeller: keep raw values only at columns 0,3,6,..
 * and rows 0,4,8,.., zero elsewhere, then scale by 1/factor in fp32.
 * Follows examples/label_instance_rgbd.cpp:89-100 (factor 5000, config/TUM3.yaml:34). */
void oracle_depth_prep(const uint16_t *raw, int rows, int cols, float factor, float *out)
{
    float scale = 1.0f / factor;
    for (int r = 0; r < rows; ++r)
        for (int c = 0; c < cols; ++c) {
            float v = (r % 4 == 0 && c % 3 == 0) ? (float)raw[r * cols + c] : 0.0f;
            out[r * cols + c] = v * scale;
        }
}

/* Per-instance masking: depth * (mask/255) with an 8-bit mask whose values are {0,255}
 * (src/MaskRCNN.cpp:354-357).  OpenCV's CV_8U division rounds to nearest, so 255 -> 1 and
 * 0 -> 0 (other values, which the reference never produces, round at 127.5).
 * Follows Engine::Run, src/Engine.cpp:192-193. */
void oracle_mask_depth(const float *depth, const uint8_t *mask, int n, float *out)
{
    for (int i = 0; i < n; ++i) out[i] = depth[i] * (mask[i] >= 128 ? 1.0f : 0.0f);
}

/* ------------------------------------------------------------------------------------
 * Per-voxel semantic-label fusion (BASELINE config 5).  NOT in the reference's TSDF: the
 * reference fuses instance evidence per sparse ObjectPoint -- Fp += score when a point is seen
 * inside a mask of its object, Bp += score otherwise, P = Fp/(Fp+Bp), bad when P < threshold
 * (ref: src/ObjectPoint.cpp:190-219, :149-154; threshold Engine.mProbThd = 0.5, config/TUM3.yaml:92).
 * This project applies the same evidence rule per voxel; parity is against this restatement only.
 *
 * For every voxel that Integrate would update this frame (same geometry and depth tests,
 * src/tsdf.cu:27-49) AND that lies inside the truncation band (diff < trunc), with
 * l = label_im[pixel] (0 = no instance) and s = score_im[pixel]:
 *     l == 0                     -> nothing
 *     label == 0                 -> label = l, Fp = s, Bp = 0            (adopt)
 *     label == l                 -> Fp = Fp + s
 *     otherwise                  -> Bp = Bp + s; if Fp / (Fp + Bp) < thd: label = l, Fp = s, Bp = 0
 * Returns the number of voxels whose label state changed.
 * ---------------------------------------------------------------------------------- */
int64_t oracle_integrate_labels(const float *cam_K, const float *cam2base, const float *depth_im,
                                const uint16_t *label_im, const float *score_im, int im_height, int im_width,
                                int dim_x, int dim_y, int z_begin, int z_end,
                                float origin_x, float origin_y, float origin_z, float voxel_size,
                                float trunc_margin, float max_depth, float prob_thd,
                                uint16_t *label, float *fp, float *bp)
{
    int64_t n = 0;
    /* rows (z, y) are independent (a voxel's label state depends on nothing but its own history): all host
     * threads, so that a whole 2048 x 2048 x 256 slab of BASELINE configs[4] can be checked voxel by voxel */
#ifdef _OPENMP
#pragma omp parallel for collapse(2) schedule(static) reduction(+ : n)
#endif
    for (int gz = z_begin; gz < z_end; ++gz)
        for (int gy = 0; gy < dim_y; ++gy)
            for (int gx = 0; gx < dim_x; ++gx) {
                const int64_t i = ((int64_t)(gz - z_begin) * dim_y + gy) * dim_x + gx;
                float bx = origin_x + (float)gx * voxel_size;
                float by = origin_y + (float)gy * voxel_size;
                float bz = origin_z + (float)gz * voxel_size;
                float dx = bx - cam2base[3], dy = by - cam2base[7], dz = bz - cam2base[11];
                float cx = cam2base[0] * dx + cam2base[4] * dy + cam2base[8] * dz;
                float cy = cam2base[1] * dx + cam2base[5] * dy + cam2base[9] * dz;
                float cz = cam2base[2] * dx + cam2base[6] * dy + cam2base[10] * dz;
                if (cz <= 0.0f) continue;
                float pu = roundf(cam_K[0] * (cx / cz) + cam_K[2]);
                float pv = roundf(cam_K[4] * (cy / cz) + cam_K[5]);
                if (!(pu >= 0.0f && pu < (float)im_width && pv >= 0.0f && pv < (float)im_height)) continue;
                const int pix = (int)pv * im_width + (int)pu;
                float d = depth_im[pix];
                if (d <= 0.0f || d > max_depth) continue;
                float diff = d - cz;
                if (diff <= -trunc_margin) continue;
                if (!(diff < trunc_margin)) continue;      /* free space in front of the surface: no label evidence */
                const uint16_t l = label_im[pix];
                if (l == 0) continue;
                const float s = score_im[pix];
                if (label[i] == 0) { label[i] = l; fp[i] = s; bp[i] = 0.0f; }
                else if (label[i] == l) { fp[i] = fp[i] + s; }
                else {
                    bp[i] = bp[i] + s;
                    if (fp[i] / (fp[i] + bp[i]) < prob_thd) { label[i] = l; fp[i] = s; bp[i] = 0.0f; }
                }
                ++n;
            }
    return n;
}

/* Label / score images from K instance masks in MaskRCNN's output format (uint8 {0,255} per mask,
 * label 1..80, score; ref: src/MaskRCNN.cpp:316-362): per pixel the covering instance with the
 * highest score wins, the lower index on ties; pixels no mask covers get label 0, score 0. */
void oracle_compose_labels(const uint8_t *masks, const uint16_t *labels, const float *scores, int k,
                           int n_pixels, uint16_t *label_im, float *score_im)
{
    for (int p = 0; p < n_pixels; ++p) {
        uint16_t l = 0;
        float s = 0.0f;
        for (int m = 0; m < k; ++m)
            if (masks[(int64_t)m * n_pixels + p] >= 128 && (l == 0 || scores[m] > s)) { l = labels[m]; s = scores[m]; }
        label_im[p] = l;
        score_im[p] = s;
    }
}

/* ----------------------------------------------------------------------------------
 * Per-voxel colour fusion (the TSDFfusion surface).  The rule is tsdf-fusion-python's (the package the reference's
 * Python glue calls, src/TSDFfusion.py.in:43; absent, version unpinned): every voxel the frame updates -- the same
 * tests as oracle_integrate, which must have run for this frame already, so weight[] holds w_new -- takes per 8-bit
 * channel min(255, round((c * w_old + c_pixel) / w_new)) with w_old = w_new - 1; channels packed B << 16 | G << 8 | R,
 * rgb = 3 bytes per pixel, channel 0 = R.  Returns the number of voxels coloured.  Parity unpinned by reference output.
 * ---------------------------------------------------------------------------------- */
static uint32_t blend_channel(uint32_t old_c, uint32_t new_c, float w_old, float w_new)
{
    float v = roundf(((float)old_c * w_old + (float)new_c) / w_new);
    v = fminf(v, 255.0f);
    return (uint32_t)v;
}

int64_t oracle_integrate_colour(const float *cam_K, const float *cam2base, const float *depth_im, const uint8_t *rgb,
                                int im_height, int im_width, int dim_x, int dim_y, int z_begin, int z_end,
                                float origin_x, float origin_y, float origin_z, float voxel_size, float trunc_margin,
                                float max_depth, const float *weight, uint32_t *colour)
{
    int64_t n = 0;
    /* rows (z, y) are independent (a voxel's label state depends on nothing but its own history): all host
     * threads, so that a whole 2048 x 2048 x 256 slab of BASELINE configs[4] can be checked voxel by voxel */
#ifdef _OPENMP
#pragma omp parallel for collapse(2) schedule(static) reduction(+ : n)
#endif
    for (int gz = z_begin; gz < z_end; ++gz)
        for (int gy = 0; gy < dim_y; ++gy)
            for (int gx = 0; gx < dim_x; ++gx) {
                const int64_t i = ((int64_t)(gz - z_begin) * dim_y + gy) * dim_x + gx;
                float bx = origin_x + (float)gx * voxel_size;
                float by = origin_y + (float)gy * voxel_size;
                float bz = origin_z + (float)gz * voxel_size;
                float dx = bx - cam2base[3], dy = by - cam2base[7], dz = bz - cam2base[11];
                float cx = cam2base[0] * dx + cam2base[4] * dy + cam2base[8] * dz;
                float cy = cam2base[1] * dx + cam2base[5] * dy + cam2base[9] * dz;
                float cz = cam2base[2] * dx + cam2base[6] * dy + cam2base[10] * dz;
                if (cz <= 0.0f) continue;
                float pu = roundf(cam_K[0] * (cx / cz) + cam_K[2]);
                float pv = roundf(cam_K[4] * (cy / cz) + cam_K[5]);
                if (!(pu >= 0.0f && pu < (float)im_width && pv >= 0.0f && pv < (float)im_height)) continue;
                const int pix = (int)pv * im_width + (int)pu;
                float d = depth_im[pix];
                if (d <= 0.0f || d > max_depth) continue;
                float diff = d - cz;
                if (diff <= -trunc_margin) continue;
                const float w_new = weight[i], w_old = w_new - 1.0f;
                const uint8_t *px = rgb + (int64_t)pix * 3;
                const uint32_t c = colour[i];
                const uint32_t r = blend_channel(c & 255u, px[0], w_old, w_new);
                const uint32_t g = blend_channel((c >> 8) & 255u, px[1], w_old, w_new);
                const uint32_t b = blend_channel((c >> 16) & 255u, px[2], w_old, w_new);
                colour[i] = (b << 16) | (g << 8) | r;
                ++n;
            }
    return n;
}

int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
