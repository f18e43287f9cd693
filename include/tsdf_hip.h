/*
 * tsdf_hip.h -- C ABI of libtsdf_hip.so, the MI355X (gfx950) TSDF volumetric-fusion library.
 *
 * This is the drop-in boundary for the dense-grid TSDF path of Tariq-Abuhashim/semantic-slam.
 * Every entry point below replaces one piece of the reference's `class TSDF`
 * (include/tsdf.hpp:22-98, src/tsdf.cu) and is what a binding on the reference side would
 * call (include/tsdf.hpp and include/TSDFfusion.hpp in this repository
 * are those bindings for C++).  Citations "ref:" are paths inside the reference repository.
 *
 * Conventions
 *   - plain C: opaque handle, pointers and sizes only; no C++/torch types cross the boundary
 *   - every function returns TSDF_OK (0) or a negative tsdf_status; the message of the last
 *     failure on the calling thread is tsdf_last_error()
 *   - matrices are 16 floats, row-major 4x4 (ref: src/tsdf.cu:253); intrinsics are 9 floats,
 *     row-major 3x3 (ref: include/tsdf.hpp:96)
 *   - grids are x-fastest: index = (z*dim_y + y)*dim_x + x (ref: src/tsdf.cu:52)
 *   - a handle owns one z-slab [z_begin, z_end) of the global grid on one device and one
 *     HIP stream; distinct handles may be driven from distinct threads, one handle may not
 *     (ref: the reference holds one TSDF per Object and never shares it, src/Engine.cpp:170-172)
 *   - "host" pointers are ordinary memory the caller owns and may free as soon as the call
 *     returns; "device" pointers are HBM addresses valid on the handle's device
 */
#ifndef TSDF_HIP_H
#define TSDF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct tsdf_volume tsdf_volume; /* opaque */

typedef enum tsdf_status {
    TSDF_OK = 0,
    TSDF_ERR_INVALID = -1,   /* bad argument (NULL, non-positive size, slab outside the grid) */
    TSDF_ERR_HIP = -2,       /* a HIP runtime call failed; tsdf_last_error() has its name */
    TSDF_ERR_IO = -3,        /* file could not be written / read */
    TSDF_ERR_NO_DEVICE = -4  /* no usable gfx950 device */
} tsdf_status;

/*
 * Everything that is a compile-time member initialiser in the reference, made run-time.
 * tsdf_config_default() fills in exactly the reference's values:
 *   dims 200^3, voxel 0.004 m, trunc 5*voxel   (ref: include/tsdf.hpp:63-67)
 *   K = TUM fr3 {535.4,0,320.1, 0,539.2,247.6, 0,0,1} (ref: include/tsdf.hpp:96)
 *   max_depth 6 m                               (ref: src/tsdf.cu:46)
 *   base2world = identity, origin = 0, slab = whole grid, device 0
 */
typedef struct tsdf_config {
    int32_t im_height, im_width;       /* depth image size (ref ctor args h, w: src/tsdf.cu:62) */
    int32_t dim_x, dim_y, dim_z;       /* GLOBAL grid size in voxels */
    int32_t z_begin, z_end;            /* this handle's slab, global z in [z_begin, z_end) */
    float voxel_size;                  /* metres */
    float trunc_margin;                /* metres */
    float max_depth;                   /* depth samples > max_depth are ignored */
    float origin[3];                   /* grid origin in the base camera frame (ref ctor arg) */
    float cam_K[9];                    /* intrinsics */
    float base2world[16];              /* base camera pose (ref ctor arg base2world_vec) */
    int32_t device;                    /* HIP device ordinal */
    int32_t id;                        /* names tsdf<id>.ply / tsdf<id>.bin (ref: src/tsdf.cu:109,116) */
} tsdf_config;

/* Fill *cfg with the reference defaults for an h x w depth image. */
int tsdf_config_default(tsdf_config *cfg, int32_t im_height, int32_t im_width);

/*
 * Replaces TSDF::TSDF (ref: src/tsdf.cu:62-96): allocates the slab in HBM, sets TSDF = 1 and
 * weight = 0 on the device (no host fill + upload), inverts base2world (a singular
 * base2world leaves the inverse all-zero and is not an error, as in ref: src/tsdf.cu:74).
 */
int tsdf_create(const tsdf_config *cfg, tsdf_volume **out);

/* Frees device memory, the stream and staging buffers.  Writes no files (see tsdf_save_*). */
int tsdf_destroy(tsdf_volume *vol);

/* Back to TSDF = 1, weight = 0 (ref: src/tsdf.cu:79-81), asynchronously on the handle's stream. */
int tsdf_reset(tsdf_volume *vol);

/*
 * Replaces TSDF::Integrate (ref: src/tsdf.cu:135-168).  depth_host: im_height*im_width floats,
 * metres, row-major, borrowed for the call only (copied to a pinned staging slot before
 * returning).  cam2world: current camera pose.  cam2base = inverse(base2world) * cam2world
 * is composed on the host in the reference's fp32 operation order (ref: src/tsdf.cu:142).
 * The call does not wait for the GPU.  DEFERRED INTEGRATION: the reference never reads a result back before its
 * destructor (ref: src/tsdf.cu:101-104), so the frames of successive calls are collected in HBM and applied 32 at a
 * time as one fused sequence (tsdf_integrate_frames_device: same results, bit for bit, several times the throughput);
 * every entry point that observes or changes the volume -- sync, download, extraction, save, reset, any other integrate
 * call -- first applies what has been collected.  tsdf_set_deferral(vol, 0) makes every call launch its own kernel.
 */
int tsdf_integrate(tsdf_volume *vol, const float *depth_host, const float cam2world[16]);
/* Frames tsdf_integrate collects per launch: 0 or 1 = none (one kernel per call), at most 32 (the default). */
int tsdf_set_deferral(tsdf_volume *vol, int32_t n_frames);

/*
 * Same from the sensor's raw 16-bit frame (TUM PNG payload): copies im_height*im_width uint16
 * (half the bytes of the float frame), converts on the device to metres with
 * value * (1.0f / depth_factor), keeping only pixels with row % row_step == 0 and
 * col % col_step == 0 (others become 0), then integrates.  depth_factor 5000, steps (4, 3)
 * reproduce the offline labeller's preparation (ref: examples/label_instance_rgbd.cpp:89-100,
 * config/TUM3.yaml:34); steps (1, 1) keep the whole frame.
 */
int tsdf_integrate_u16(tsdf_volume *vol, const uint16_t *raw_host, float depth_factor, int32_t row_step,
                       int32_t col_step, const float cam2world[16]);

/* The conversion alone, device to device, queued on the handle's stream. */
int tsdf_convert_depth_u16(tsdf_volume *vol, const uint16_t *raw_dev, float *depth_dev, float depth_factor,
                           int32_t row_step, int32_t col_step);

/*
 * Same, with the depth frame already resident in HBM on the handle's device.  Deferred like tsdf_integrate: the frame is
 * copied device to device into the collecting pool on the handle's stream (the ordering its kernel would have had), so
 * depth_dev may be reused under that stream's order as before; with deferral off the kernel reads depth_dev itself.
 * tsdf_integrate_cam2base and tsdf_integrate_masked_device (mask copied with the frame) are collected the same way.
 */
int tsdf_integrate_device(tsdf_volume *vol, const float *depth_dev, const float cam2world[16]);

/*
 * Same as tsdf_integrate_device, for callers that composed the relative pose themselves
 * (cam2base is used as given; base2world is ignored).
 */
int tsdf_integrate_cam2base(tsdf_volume *vol, const float *depth_dev, const float cam2base[16]);

/*
 * A known sequence of frames (offline replay of saved keyframes, ref:
 * examples/label_instance_rgbd.cpp:78-110): exactly n_frames consecutive tsdf_integrate_device /
 * tsdf_integrate_masked_device calls -- same results, bit for bit -- but the library may apply
 * several frames per pass over the volume (weights read and written once per group).
 * depth_dev: n_frames device pointers; masks_dev: NULL or n_frames device pointers (entries may be
 * NULL); cam2world: n_frames x 16 floats.
 * Ordering: every frame (and mask) must be readable once the work queued on the handle's stream before this call has
 * run, and must stay unchanged until the work queued by this call has run.  On slabs below 64 M voxels the library
 * reads the frames of a later pass (their depth tile tables, the brick work list) on a side stream of its own while an
 * earlier pass is still integrating; that stream waits for an event recorded on the handle's stream when the call starts
 * and is joined to the handle's stream before the call returns, so the caller sees one stream as before.
 */
int tsdf_integrate_frames_device(tsdf_volume *vol, const float *const *depth_dev, const uint8_t *const *masks_dev,
                                 const float *cam2world, int32_t n_frames);

/*
 * Per-instance fusion as the reference's caller prepares it: depth * (mask/255) with an
 * 8-bit {0,255} instance mask (ref: src/Engine.cpp:192-193), fused into the depth load
 * instead of materialising the masked image.  mask_dev: im_height*im_width bytes in HBM.
 */
int tsdf_integrate_masked_device(tsdf_volume *vol, const float *depth_dev, const uint8_t *mask_dev,
                                 const float cam2world[16]);

/* Block until everything queued on the handle's stream has finished. */
int tsdf_sync(tsdf_volume *vol);

/*
 * Copy this handle's slab to host arrays of tsdf_slab_voxels() floats each (either may be
 * NULL).  The reference only ever reads results back in its destructor
 * (ref: src/tsdf.cu:101-104); this is the same copy, callable at any time.  Synchronous.
 */
int tsdf_download(tsdf_volume *vol, float *tsdf_host, float *weight_host);

/*
 * Copy n_slices whole z-slices starting at slab-local slice z_local into tsdf_dst / weight_dst
 * (either may be NULL), each n_slices*dim_y*dim_x floats.  The destinations may be host or
 * device addresses (the copy kind is inferred), so a one-slice halo can go straight into a
 * communication buffer in HBM.  Synchronous with respect to the handle's stream.
 */
int tsdf_copy_slices(tsdf_volume *vol, int32_t z_local, int32_t n_slices, void *tsdf_dst, void *weight_dst);

/* Restore a slab from host arrays (resume from a saved state).  Synchronous. */
int tsdf_upload(tsdf_volume *vol, const float *tsdf_host, const float *weight_host);

/*
 * Device addresses of the slab's two arrays (x-fastest, slab-local z).  The library keeps a
 * small free-space summary of the TSDF array; after WRITING TSDF values through
 * this pointer call tsdf_refresh_summary().  Reading needs nothing.
 */
int tsdf_refresh_summary(tsdf_volume *vol);
int tsdf_device_ptrs(tsdf_volume *vol, float **tsdf_dev, float **weight_dev);

/* Number of voxels in this handle's slab: dim_x * dim_y * (z_end - z_begin). */
int64_t tsdf_slab_voxels(const tsdf_volume *vol);
/* Frames tsdf_integrate_frames_device / tsdf_integrate_sequence_timed apply per pass over this slab (1 when
 * the selected kernel variant does not fuse frames). */
int32_t tsdf_frames_per_launch(const tsdf_volume *vol);
/*
 * Diagnostics of the fused path's per-wavefront patch classification (depth tile summaries): reads the
 * counters {wavefront-frames that took the per-voxel path, that were updated as free space without projecting
 * a voxel, that were skipped} accumulated since they were last enabled (counts_out may be NULL), then
 * enables (and zeroes) or disables them.  Off by default; synchronises the stream.
 */
int tsdf_shortcut_stats(tsdf_volume *vol, int32_t enable, uint64_t counts_out[3]);
/*
 * More of the same, for the classified fused launches' brick work list (csrc/tsdf_multiframe.hip.h, classify_brick_list),
 * accumulated while the counters of tsdf_shortcut_stats are enabled: {super-bricks every frame skipped, bricks put on the
 * work list, of those: bricks whose super-brick left frames undecided (they classify themselves), of those: bricks every
 * frame skipped after all}.  Synchronises the stream; does not reset.
 */
int tsdf_brick_list_stats(tsdf_volume *vol, uint64_t counts_out[4]);
/*
 * State of the per-launch decision whether to classify (default kernel variant): info_out[0] = fraction of the
 * workgroup-frames the last counted launch claimed (-1 before the first read-back), info_out[1] = launches that have
 * gone without classification since the last one that classified.  Synchronises the stream.
 */
int tsdf_classification_info(tsdf_volume *vol, double info_out[2]);

/* Copy of the configuration the handle was created with. */
int tsdf_get_config(const tsdf_volume *vol, tsdf_config *out);

/* The relative pose used by the most recent integrate call (16 floats), for parity tests. */
int tsdf_last_cam2base(const tsdf_volume *vol, float out[16]);

/*
 * Run subsequent work of this handle on a caller-owned hipStream_t (passed as void*), e.g.
 * the current PyTorch stream, or back on the handle's own stream when stream == NULL.
 * (Copies of host frames run on the store's copy stream and, for sequence calls, table kernels on a side stream of the
 * handle; both are ordered against this stream by events, so work queued on it after a call sees that call's result.)
 */
int tsdf_set_stream(tsdf_volume *vol, void *hip_stream);
int tsdf_get_stream(tsdf_volume *vol, void **hip_stream);

/*
 * Number of voxels whose weight is > weight_thresh and whose TSDF is non-zero: the surface
 * test of ref: src/tsdf.cu:179, counted on the device.  Synchronous.
 */
int tsdf_count_surface(tsdf_volume *vol, float weight_thresh, int64_t *count);

/*
 * Surface points of the slab in grid order, xyz triples in the base camera frame
 * (ref: src/tsdf.cu:195-212), compacted on the device.  xyz_host receives at most
 * capacity points; *count is the number found.  Synchronous.
 */
int tsdf_extract_surface(tsdf_volume *vol, float weight_thresh, float *xyz_host, int64_t capacity,
                         int64_t *count);

/*
 * Zero-crossing surface vertices (the vertex set of marching cubes), compacted on the device in
 * grid order: for every voxel and every +x, +y, +z neighbour with both weights > weight_thresh and
 * TSDF values of opposite sign, the linearly interpolated crossing point.  Not a reference
 * function (its mesh path is the absent tsdf-fusion-python, ref: src/TSDFfusion.py.in:48-53); the
 * rule is defined in oracle/tsdf_oracle.c (oracle_zero_crossings) and csrc/tsdf_extract.hip.h.
 * halo_tsdf / halo_weight: the dim_y*dim_x values of global slice z_end -- what a z-slab owner
 * receives from its upper neighbour (host or device memory) -- or both NULL on the top slab.
 * Call with xyz_host == NULL to get *count only.  Synchronous.
 */
int tsdf_extract_crossings(tsdf_volume *vol, const float *halo_tsdf, const float *halo_weight, float weight_thresh,
                           float *xyz_host, int64_t capacity, int64_t *count);

/*
 * Triangle mesh of the zero level set by marching tetrahedra, on the device, cubes in grid order,
 * 9 floats (3 vertices) per triangle; a watertight, consistently wound triangle soup (edge vertices of
 * neighbouring cubes are bit-identical).  Stands in for the reference's SaveMesh, which needs the absent
 * tsdf-fusion-python (ref: src/TSDFfusion.py.in:48-53); rule in csrc/tsdf_extract.hip.h, checked against
 * this project's CPU restatement only.  halo_* as for tsdf_extract_crossings.  tsdf_save_mesh_ply writes a
 * binary .ply (vertex + face elements) of a whole-grid handle.
 */
int tsdf_extract_mesh(tsdf_volume *vol, const float *halo_tsdf, const float *halo_weight, float weight_thresh,
                      float *triangles_host, int64_t capacity, int64_t *count);
int tsdf_save_mesh_ply(tsdf_volume *vol, const char *path, float weight_thresh);
/*
 * The same mesh as the reference's Python glue saves it (ref: src/TSDFfusion.py.in:48-53, get_mesh -> verts, faces,
 * norms, colors -> meshwrite): shared (welded) vertices -- the soup's edge vertices are bit-identical between cubes, so
 * welding matches coordinate bits exactly --, an area-weighted normal per vertex, a colour per vertex when colour is
 * enabled, faces as vertex indices in the soup's order and winding.  Binary little-endian .ply.
 */
int tsdf_save_mesh_welded_ply(tsdf_volume *vol, const char *path, float weight_thresh);

/*
 * File writers, byte-compatible with the reference's destructor (ref: src/tsdf.cu:107-132,
 * 170-218).  For a slab handle the .bin header carries the slab's dims and a z-shifted
 * origin is NOT applied: callers that shard gather slabs in z order first (see
 * semantic_slam_amd/sharded.py); a whole-grid handle writes exactly the reference's files.
 *   .ply: binary_little_endian points with |tsdf| != 0 and weight > weight_thresh (0.9 in the ref); the reference
 *         prints the vertex count with %d, so more than 2^31 - 1 points cannot be represented: TSDF_ERR_INVALID
 *   .bin: 8-float header {dim_x, dim_y, dim_z, origin xyz, voxel_size, trunc} + TSDF floats
 */
int tsdf_save_ply(tsdf_volume *vol, const char *path, float weight_thresh);
int tsdf_save_bin(tsdf_volume *vol, const char *path);

/*
 * Checkpoint / resume.  The reference only writes (ref: src/tsdf.cu:114-132) and nothing in it reads a
 * .bin back.  tsdf_load_bin restores the TSDF array from a file in the reference's .bin format (its
 * 8-float header must match the slab; weights are not in that format and are left untouched).
 * tsdf_save_state / tsdf_load_state round-trip the whole slab (configuration, TSDF, weights) in this
 * library's own format, so an interrupted fusion continues bit-exactly.
 */
int tsdf_load_bin(tsdf_volume *vol, const char *path);
int tsdf_save_state(tsdf_volume *vol, const char *path);
int tsdf_load_state(tsdf_volume *vol, const char *path);

/*
 * Timing aid for benchmarks: queue n_frames integrations of one device-resident depth frame
 * with poses cam2world[k*16..] back to back on the handle's stream, bracketed by HIP events
 * on that stream; *elapsed_ms is the device time between the events.  Synchronous.
 */
int tsdf_integrate_sequence_timed(tsdf_volume *vol, const float *depth_dev, const float *cam2world,
                                  int32_t n_frames, float *elapsed_ms);

/*
 * The same for a sequence whose frames each bring their own depth image (and optional instance mask): exactly
 * tsdf_integrate_frames_device, bracketed by HIP events on the handle's stream.  Synchronous.
 */
int tsdf_integrate_frames_timed(tsdf_volume *vol, const float *const *depth_dev, const uint8_t *const *masks_dev,
                                const float *cam2world, int32_t n_frames, float *elapsed_ms);

/*
 * Measurement aid: n_frames one-frame launches (one resident depth frame, poses cam2world[k*16..]) queued call by call
 * against the same launches replayed from a captured hipGraph, iters repetitions each; device milliseconds per
 * repetition.  Applies 2 * iters * n_frames (+ warm-up) frames to the volume.  (small grids.)
 */
int tsdf_probe_graph_replay(tsdf_volume *vol, const float *depth_dev, const float *cam2world, int32_t n_frames,
                            int32_t iters, float *ms_launches, float *ms_graph);

/*
 * Ceiling probe: n_iters passes of a bare 16 B/voxel read-modify-write stream over the slab
 * (values unchanged), timed with HIP events.  non_temporal selects nt loads/stores.
 */
int tsdf_probe_stream(tsdf_volume *vol, int32_t non_temporal, int32_t n_iters, float *elapsed_ms);

/*
 * Device self-test of the kernel's shared-reciprocal division against the compiler's IEEE
 * division on n_samples pseudo-random operand pairs in the range the kernel uses it for
 * (DESIGN.md section 4).  A quotient may differ from the IEEE one only below 2^-42 and only if the
 * pixel coordinate fl(fx*q + cx) it feeds is unchanged for the given fx, cx (any |fx| < 2^14 the
 * kernel's fast path admits).  *mismatches must come back 0; first_bad = {n, d, got, want} otherwise.
 */
int tsdf_selftest_fastdiv(int32_t device, uint64_t seed, uint64_t n_samples, float fx, float cx,
                          uint64_t *mismatches, float first_bad[4]);

/*
 * Device self-test of the truncated distance's division diff / trunc through the shared refined reciprocal (fused
 * kernels, csrc/tsdf_kernels.hip.h: fast_div_r) against the compiler's IEEE division on n_samples operand pairs drawn
 * from the domain the kernel admits: divisor in [2^-20, 2^20], numerator 0, NaN or of magnitude in [2^-81, 2^60].
 * Every quotient must be bit-identical.  *mismatches must come back 0; first_bad = {n, d, got, want} otherwise.
 */
int tsdf_selftest_fastdiv_band(int32_t device, uint64_t seed, uint64_t n_samples, uint64_t *mismatches,
                               float first_bad[4]);

/*
 * Exhaustive device self-test of the kernel's one-instruction pixel rounding (v_cvt_rpi_i32_f32)
 * against roundf for every fp32 value in (-0.5, 2^24].  *mismatches must come back 0;
 * first_bad = {u, got, want, 0} otherwise.
 */
int tsdf_selftest_round(int32_t device, uint64_t *mismatches, float first_bad[4]);

/*
 * Device self-test of the depth tile tables the classified launches consult: builds the table of one frame (depth x mask,
 * mask_dev may be NULL) with the kernels the library launches (whole-row strips, levels by doubling) and with the plain
 * ones (one wavefront per tile, levels by scanning), for both tile sizes the library uses (16 x 16 pixels; 8 x 8 for slabs of
 * 10 M voxels and more), and counts the entries that differ in any bit; *mismatches must come back 0.
 */
int tsdf_selftest_tile_tables(int32_t device, const float *depth_dev, const uint8_t *mask_dev, int32_t im_height,
                              int32_t im_width, float max_depth, uint64_t *mismatches);

/* Select the Integrate kernel variant (0 = default; others are listed in DESIGN.md). */
int tsdf_set_kernel_variant(tsdf_volume *vol, int32_t variant);

/*
 * Tuning knob (no reference counterpart; results never depend on it): the box of voxels one wavefront owns, and
 * classifies as a whole, in the classified launches (bricks): `quads` x 4 voxels of `rows` rows of `slices`
 * slices.  Needs quads * rows * slices <= 64 and quads dividing dim_x / 4; (0, 0, 0) returns to the library's choice
 * for the grid.  tsdf_brick_shape reads the shape in use ({0, 0, 0}: the grid has no brick view, dim_x % 4 != 0).
 */
int tsdf_set_brick_shape(tsdf_volume *vol, int32_t quads, int32_t rows, int32_t slices);
int tsdf_brick_shape(const tsdf_volume *vol, int32_t shape_out[3]);
/* The shape tsdf_create would choose for this grid (host arithmetic only: needs no device). */
int tsdf_default_brick_shape(const tsdf_config *cfg, int32_t shape_out[3]);

/*
 * Integrate AND label fusion of a known sequence of frames in the same passes over the volume: identical to
 * calling tsdf_integrate_device and tsdf_integrate_labels_device for every frame in order, but the label
 * evidence reuses the projection and depth tests Integrate has just made (the separate sweep recomputes
 * them).  Needs dim_x % 4 == 0 and tsdf_labels_enable.  All pointers are device pointers that stay valid
 * until the stream has run the launches.
 */
int tsdf_integrate_frames_labels_device(tsdf_volume *vol, const float *const *depth_dev,
                                        const uint16_t *const *label_im_dev, const float *const *score_im_dev,
                                        const float *cam2world, int32_t n_frames);

/*
 * Per-voxel semantic-label fusion (BASELINE config 5).  Not a function of the reference's TSDF:
 * the reference fuses instance evidence per sparse ObjectPoint -- Fp += score inside a mask of the
 * point's object, Bp += score otherwise, P = Fp/(Fp+Bp), dropped below a threshold
 * (ref: src/ObjectPoint.cpp:190-219,149-154; Engine.mProbThd = 0.5, config/TUM3.yaml:92); the same
 * rule is applied here per voxel of the dense grid (csrc/tsdf_labels.hip.h states it exactly).
 *   tsdf_labels_enable        allocate (or clear) label:uint16, Fp:f32, Bp:f32 arrays for the slab
 *   tsdf_compose_labels       K instance masks (MaskRCNN format, ref: src/MaskRCNN.cpp:316-362:
 *                             K*H*W uint8 {0,255} in device memory, label and score per instance on
 *                             the host) -> one label image + one score image on the device
 *   tsdf_integrate_labels_device  one frame: voxels observed inside the truncation band (same pixel
 *                             and depth tests as tsdf_integrate*) take evidence from the label image
 *   tsdf_download_labels      copy the three arrays out (any pointer may be NULL)
 */
int tsdf_labels_enable(tsdf_volume *vol, float prob_threshold);
int tsdf_compose_labels(tsdf_volume *vol, const uint8_t *masks_dev, const uint16_t *labels_host,
                        const float *scores_host, int32_t k, uint16_t *label_im_dev, float *score_im_dev);
int tsdf_integrate_labels_device(tsdf_volume *vol, const float *depth_dev, const uint16_t *label_im_dev,
                                 const float *score_im_dev, const float cam2world[16]);
int tsdf_download_labels(tsdf_volume *vol, uint16_t *label_host, float *fp_host, float *bp_host);

/*
 * Per-voxel colour fusion for the TSDFfusion surface.  The reference's second backend passes every RGB-D frame to the
 * third-party tsdf-fusion-python (ref: src/TSDFfusion.py.in:43 `integrate(color_image, depth_im, cam_intr, cam_pose,
 * obs_weight=1.)`), which fuses colour beside the distance and colours its mesh (ref: src/TSDFfusion.py.in:48-53).  That
 * package is absent, so this restates its published rule (csrc/tsdf_colour.hip.h): every voxel a frame updates takes per
 * 8-bit channel min(255, round((c * w_old + c_pixel) / w_new)); parity unpinned.
 *   tsdf_colour_enable            allocate (or clear) one packed colour (0x00BBGGRR) per voxel of the slab
 *   tsdf_integrate_colour_device  the colour pass of ONE frame, to be queued right after that frame's
 *                                 tsdf_integrate*_device call (it reads the weights that call has written);
 *                                 rgb_dev: im_height*im_width*3 bytes, channel 0 = R
 *   tsdf_integrate_rgbd           both passes from host images (TSDFfusion::Integrate): copies depth and colour to
 *                                 pinned staging, integrates, then fuses colour; the call does not wait
 *   tsdf_download_colour          copy the packed colours out (tsdf_slab_voxels() uint32)
 * With colour enabled tsdf_save_mesh_ply writes per-vertex red/green/blue (the nearest voxel's colour).
 * The colour pass has no weight of its own: it takes w_new from the weight array the geometry pass has just written and
 * w_old = w_new - 1 (the package's obs_weight = 1).  So (i) a frame integrated WITHOUT a colour image (tsdf_integrate*, or
 * TSDFfusion::Integrate with an empty colour Mat) still raises the weight, and later colour frames are blended as if that
 * frame had confirmed the colour the voxel already had -- the behaviour of the package when every frame brings colour, a
 * documented deviation otherwise; (ii) past 2^24 updates of one voxel w_new - 1 is no longer exact in fp32 (the weights
 * themselves stop counting there, as the reference's do).  The pass re-derives the frame's updated voxels with the same
 * projection code as Integrate; tests/test_gpu_colour.py checks the two sets voxel for voxel on poses that send wavefronts of
 * both kernels down the generic projection.
 */
int tsdf_colour_enable(tsdf_volume *vol);
int tsdf_integrate_colour_device(tsdf_volume *vol, const float *depth_dev, const uint8_t *rgb_dev,
                                 const float cam2world[16]);
int tsdf_integrate_rgbd(tsdf_volume *vol, const float *depth_host, const uint8_t *rgb_host, const float cam2world[16]);
int tsdf_download_colour(tsdf_volume *vol, uint32_t *colour_host);

/*
 * Grid origin of a new object volume from its first (masked) depth frame, on the device: the per-axis
 * minimum over pixels with depth > 0 of the back-projected point, starting from 1000 -- what
 * Object::Object computes on the host before it constructs its TSDF (ref: src/Object.cpp:37-49, with the
 * instance mask of ref: src/Engine.cpp:192-193; mask_dev may be NULL).  Bit-identical to that loop for every
 * frame without NaN samples (a minimum is order-independent); with NaN samples the reference's running std::min
 * depends on the raster order (a NaN replaces the minimum, the next valid pixel replaces the NaN) whereas this
 * reduction ignores them.  The call waits for all work queued on the device (whatever stream produced depth_dev,
 * e.g. tsdf_convert_depth_u16 on a handle's stream) before it reads the frame, and returns when the result is on
 * the host.
 */
int tsdf_object_origin(int32_t device, const float *depth_dev, const uint8_t *mask_dev, int32_t im_height,
                       int32_t im_width, const float cam_K[9], float origin_out[3]);

/*
 * Batched per-object fusion: the reference keeps one small TSDF per object instance and feeds
 * each of them depth * (its instance mask) for every keyframe (ref: src/Engine.cpp:172-233,
 * src/Object.cpp:67,143-166).  A batch owns n volumes (own grid, origin and base pose each; same
 * device and image size; dim_x % 4 == 0) and integrates one frame into ALL of them per call.
 * masks_dev: n device pointers to im_height*im_width {0,255} bytes (entry or whole array may be
 * NULL = unmasked).  Volumes are borrowed with tsdf_batch_volume() for download / save /
 * extraction; they are destroyed with the batch.
 * How the frames reach the volumes is the library's choice, the result is the same bits: either one
 * kernel launch over all members per call, or -- deferral, the default for batches of few or large
 * members, as for a single handle (tsdf_set_deferral on the FIRST member sets the frames per flush, 0 / 1
 * switches it off) -- the frame is copied into the batch's pool in HBM (the depth image once, the masks
 * by one gather launch; the caller's buffers are free again under the batch stream's order) and every
 * 32 collected frames are applied by one fused launch per member.  tsdf_batch_sync and every call that
 * observes or integrates into a member through its borrowed handle apply the collected frames first.
 */
typedef struct tsdf_batch tsdf_batch;
int tsdf_batch_create(const tsdf_config *cfgs, int32_t n, tsdf_batch **out);
int tsdf_batch_destroy(tsdf_batch *batch);
int tsdf_batch_size(const tsdf_batch *batch);
int tsdf_batch_volume(tsdf_batch *batch, int32_t i, tsdf_volume **vol);
int tsdf_batch_integrate_device(tsdf_batch *batch, const float *depth_dev, const uint8_t *const *masks_dev,
                                const float cam2world[16]);
int tsdf_batch_sync(tsdf_batch *batch);

/*
 * One grid over several devices in ONE process.  The reference's host is a C++ program that owns its TSDFs directly
 * (ref: include/tsdf.hpp:22-43; src/Engine.cpp:170-172 drives distinct TSDFs from one loop), so a C++ caller must be
 * able to span the node without a process per GPU.  tsdf_group_create cuts the grid of *cfg (its z_begin / z_end /
 * device fields are ignored) into n_slabs contiguous z-slabs -- slab i holds z in [i*dim_z/n, (i+1)*dim_z/n) and lives
 * on devices[i], with its own stream; ordinals may repeat (several slabs on one device).  Needs n_slabs <= dim_z.
 *   tsdf_group_integrate         TSDF::Integrate for the whole grid (ref: src/tsdf.cu:135-168): the caller's frame is
 *                                copied once into pinned memory; deferred like tsdf_integrate -- 32 collected frames go
 *                                out as one copy and one fused launch per slab (tsdf_group_set_deferral(g, 0): one
 *                                asynchronous copy per slab and its kernel per call); no synchronisation between
 *                                devices and no collective (voxels are independent); returns without waiting
 *   tsdf_group_integrate_frames  a known sequence from host frames: per pass of up to 32 frames, one copy of the
 *                                frames to every device and one fused launch per slab (tsdf_integrate_frames_device)
 *   tsdf_group_download          the whole grid in z order (== one handle's tsdf_download, bit for bit)
 *   tsdf_group_extract_*         lists of the whole grid in grid order; zero crossings and the mesh fetch slice z_end
 *                                from the next slab device to device (hipMemcpyPeerAsync, dim_y*dim_x*8 bytes per
 *                                boundary); one host thread per slab, so the devices extract concurrently
 *   tsdf_group_save_*            the reference's files (ref: src/tsdf.cu:107-132,170-218), byte-identical to a
 *                                whole-grid handle's
 *   tsdf_group_volume            borrow slab i's handle (tsdf_device_ptrs, tsdf_set_kernel_variant, ...); it is
 *                                destroyed with the group
 */
typedef struct tsdf_group tsdf_group;
int tsdf_group_create(const tsdf_config *cfg, const int32_t *devices, int32_t n_slabs, tsdf_group **out);
int tsdf_group_destroy(tsdf_group *group);
int tsdf_group_size(const tsdf_group *group);
int64_t tsdf_group_voxels(const tsdf_group *group);
int tsdf_group_volume(tsdf_group *group, int32_t i, tsdf_volume **vol);
int tsdf_group_integrate(tsdf_group *group, const float *depth_host, const float cam2world[16]);
int tsdf_group_integrate_frames(tsdf_group *group, const float *const *depth_host, const float *cam2world,
                                int32_t n_frames);
/* Frames tsdf_group_integrate collects per pass (deferred integration as for tsdf_integrate; default 32, 0 = none). */
int tsdf_group_set_deferral(tsdf_group *group, int32_t n_frames);
int tsdf_group_sync(tsdf_group *group);
int tsdf_group_reset(tsdf_group *group);
int tsdf_group_download(tsdf_group *group, float *tsdf_host, float *weight_host);
int tsdf_group_extract_surface(tsdf_group *group, float weight_thresh, float *xyz_host, int64_t capacity,
                               int64_t *count);
int tsdf_group_extract_crossings(tsdf_group *group, float weight_thresh, float *xyz_host, int64_t capacity,
                                 int64_t *count);
int tsdf_group_extract_mesh(tsdf_group *group, float weight_thresh, float *triangles_host, int64_t capacity,
                            int64_t *count);
int tsdf_group_save_ply(tsdf_group *group, const char *path, float weight_thresh);
int tsdf_group_save_mesh_ply(tsdf_group *group, const char *path, float weight_thresh);
int tsdf_group_save_bin(tsdf_group *group, const char *path);

/* Message describing the last failure on this thread ("" when none). */
const char *tsdf_last_error(void);

/* Library version string and the offload architecture it was built for ("gfx950"). */
const char *tsdf_version(void);

/* Host-side 4x4 helpers in the reference's exact fp32 operation order
 * (ref: src/tsdf.cu:253-273 and :276-403); exported so bindings can reuse them. */
void tsdf_multiply_matrix(const float a[16], const float b[16], float out[16]);
int tsdf_invert_matrix(const float m[16], float inv_out[16]); /* 1 = ok, 0 = singular */

#ifdef __cplusplus
}
#endif
#endif /* TSDF_HIP_H */
