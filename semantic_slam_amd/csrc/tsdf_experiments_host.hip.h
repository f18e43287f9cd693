// tsdf_experiments_host.hip.h -- host side of the measurement build (-DTSDF_EXPERIMENTS), included by tsdf_capi.hip inside
// its anonymous namespace.  Dispatch of the kernel variants that do not ship (tsdf_experiments.hip.h); the product's launch
// functions call into here through four hooks and otherwise do not know about any of it.
//
//   2        the first version: ladder_rows<4> (one row per wavefront, no elision)
//   4        fused, two rows per lane (integrate_multi<R = 2>)
//   5        fused, XCD-aware workgroup order
//   6        fused, frame blocks staged in device memory instead of the kernarg
//   9        fused per-voxel launches in memory order instead of slices fastest
//   10       slices fastest without the rotation that spreads a slice group over the XCDs
//   11       always classified, with round 1's shapes: 256 x 1 rows / 1024 consecutive voxels, one class per workgroup
//            (fused: in the workgroup's prologue; one masked frame: a class table ahead of the launch)
//   12, 13   always classified, bricks per wavefront classified in a workgroup's prologue over the whole slab (round 2),
//            without / with the super-brick table -- what the brick work list replaced
//   16 + c   ladder: c = (rsel << 2) | (elide << 1) | nt, R = 1, 2, 4 for rsel = 0, 1, 2
//   32 + c   the same with the free-space summary (needs elide = 1)
//   48 + c   summary + early (speculative, frustum-gated) volume loads
//   64 + c   early loads without the summary
//   80 + c   summary + exact shared-reciprocal projection (80 + 7 is the shipped one-frame kernel's configuration)
//   96 + c   shared-reciprocal projection without the summary
//   112 + c  as 80 + c with the depth pixels of each workgroup's patch staged in LDS (c = 3, 7: R = 1, 2)
#pragma once

bool experiment_variant(int variant)
{
    const int c = (variant - 32) & 15;
    const bool sum_ok = (variant >= 32 && variant < 112 && c < 12 && ((c >> 1) & 1)) || variant == 115 || variant == 119;
    return (variant >= 0 && variant <= 13) || (variant >= 16 && variant < 28) || sum_ok;
}

void experiment_adjust(const tsdf_volume *v, bool labels, bool *classify, int *z_fastest)
{
    if (v->variant == 9) *z_fastest = 0;
    if (v->variant == 10) *z_fastest = 1;
    // label launches classify only through bricks (a claimed wavefront-frame carries no label evidence either)
    if (labels && v->variant == 11) *classify = false;
}

template <int R, bool ELIDE, bool NT, bool MASKED, bool SUM, bool EARLY = false, bool FAST = false>
void launch_ladder(const tsdf_volume *v, const tsdfk::IntegrateParams &p)
{
    dim3 block(64, 4, 1);
    dim3 grid((p.xgroups + 63) / 64, (p.dim_y + 4 * R - 1) / (4 * R), p.nz);
    hipLaunchKernelGGL((tsdfk::ladder_tile<R, ELIDE, NT, MASKED, SUM, EARLY, FAST>), grid, block, 0, v->stream, p);
}

// One masked frame into one volume: tile table of depth x mask, then the class of every workgroup of the coming launch
// (grid nbx x nby x nz; rows_per_wg as classify_workgroups takes it).  Sets p.wg_class.
int classify_single(tsdf_volume *v, tsdfk::IntegrateParams &p, int nbx, int nby, int nz, int rows_per_wg)
{
    float2 *tiles = nullptr;
    int rc0 = tables_begin(v, &tiles);   // (released by the caller's tables_end() after the launch that reads the classes)
    if (rc0) return rc0;
    const size_t n_wg = (size_t)nbx * nby * nz;
    if (v->wg_class_bytes < n_wg) {
        if (v->d_wg_class) HIP_TRY(hipFree(v->d_wg_class));
        v->d_wg_class = nullptr;
        v->wg_class_bytes = 0;
        HIP_TRY(hipMalloc((void **)&v->d_wg_class, n_wg));
        v->wg_class_bytes = n_wg;
    }
    const float *d = p.depth;
    const uint8_t *m = p.mask;
    int rc = build_tile_tables(v->stream, v->cfg, p, &d, &m, 1, tiles);
    if (rc) return rc;
    tsdfk::FramePose pose;
    pose_from_params(pose, p);
    pose.tiles = tiles;
    hipLaunchKernelGGL(tsdfk::classify_workgroups, dim3((unsigned)((n_wg + 255) / 256)), dim3(256), 0, v->stream, p, pose,
                       v->d_wg_class, nbx, nby, nz, rows_per_wg);
    HIP_TRY(hipGetLastError());
    p.wg_class = v->d_wg_class;
    return TSDF_OK;
}


// One-frame launches of the ladder (variants 2, 16 .. 119, unmasked) and variant 11's masked one-frame launch.
int launch_integrate_experiment(tsdf_volume *v, const float *depth_dev, const uint8_t *mask_dev, const float *c2b)
{
    const tsdf_config &c = v->cfg;
    const int nz = c.z_end - c.z_begin;
    const int variant = v->variant;
    if (v->flat && variant != 2) return launch_multi(v, &depth_dev, mask_dev ? &mask_dev : nullptr, c2b, 1);   // the flat mapping serves them all
    tsdfk::IntegrateParams p = make_params(v, depth_dev, mask_dev, c2b, 4);
    if (variant == 11) {   // masked, row-mapped, classified per workgroup (256 x 8 voxels)
        v->flags_known_zero = false;
        const dim3 grid((p.xgroups + 63) / 64, (p.dim_y + 7) / 8, p.nz);
        int rc = classify_single(v, p, (int)grid.x, (int)grid.y, (int)grid.z, 8);
        if (rc) return rc;
        hipLaunchKernelGGL((tsdfk::ladder_tile<2, true, true, true, true, false, true, false, true>), grid, dim3(64, 4, 1), 0, v->stream, p);
        HIP_TRY(hipGetLastError());
        return tables_end(v);
    }
    // which launches keep the free-space summary up to date: the SUM kernels; the rows kernel and the plain tile variants do not
    const bool summary = variant != 2 && ((variant >= 32 && variant < 64) || (variant >= 80 && variant < 96) || variant >= 112);
    if (!summary) {
        int rc = drop_summary(v);
        if (rc) return rc;
    } else {
        v->flags_known_zero = false;
    }
    if (variant == 2) {
        dim3 block(64, 4, 1), grid((p.xgroups + 63) / 64, (c.dim_y + 3) / 4, nz);
        if (mask_dev) hipLaunchKernelGGL((tsdfk::ladder_rows<4, true>), grid, block, 0, v->stream, p);
        else hipLaunchKernelGGL((tsdfk::ladder_rows<4, false>), grid, block, 0, v->stream, p);
    } else if (variant >= 32) {
        switch (variant - 32) {
#define SUM_CASE(code, R, N, S, E) case code: launch_ladder<R, true, N, false, S, E>(v, p); break;
#define FAST_CASE(code, R, N, S) case code: launch_ladder<R, true, N, false, S, false, true>(v, p); break;
            FAST_CASE(48 + 2, 1, false, true) FAST_CASE(48 + 3, 1, true, true)
            FAST_CASE(48 + 6, 2, false, true) FAST_CASE(48 + 7, 2, true, true)
            FAST_CASE(48 + 10, 4, false, true) FAST_CASE(48 + 11, 4, true, true)
            FAST_CASE(64 + 2, 1, false, false) FAST_CASE(64 + 3, 1, true, false)
            FAST_CASE(64 + 6, 2, false, false) FAST_CASE(64 + 7, 2, true, false)
            FAST_CASE(64 + 10, 4, false, false) FAST_CASE(64 + 11, 4, true, false)
#undef FAST_CASE
            case 80 + 3: hipLaunchKernelGGL((tsdfk::ladder_tile<1, true, true, false, true, false, true, true>),
                                            dim3((p.xgroups + 63) / 64, (p.dim_y + 3) / 4, p.nz), dim3(64, 4, 1), 0, v->stream, p); break;
            case 80 + 7: hipLaunchKernelGGL((tsdfk::ladder_tile<2, true, true, false, true, false, true, true>),
                                            dim3((p.xgroups + 63) / 64, (p.dim_y + 7) / 8, p.nz), dim3(64, 4, 1), 0, v->stream, p); break;
            SUM_CASE(2, 1, false, true, false) SUM_CASE(3, 1, true, true, false)
            SUM_CASE(6, 2, false, true, false) SUM_CASE(7, 2, true, true, false)
            SUM_CASE(10, 4, false, true, false) SUM_CASE(11, 4, true, true, false)
            SUM_CASE(16 + 2, 1, false, true, true) SUM_CASE(16 + 3, 1, true, true, true)
            SUM_CASE(16 + 6, 2, false, true, true) SUM_CASE(16 + 7, 2, true, true, true)
            SUM_CASE(16 + 10, 4, false, true, true) SUM_CASE(16 + 11, 4, true, true, true)
            SUM_CASE(32 + 2, 1, false, false, true) SUM_CASE(32 + 3, 1, true, false, true)
            SUM_CASE(32 + 6, 2, false, false, true) SUM_CASE(32 + 7, 2, true, false, true)
            SUM_CASE(32 + 10, 4, false, false, true) SUM_CASE(32 + 11, 4, true, false, true)
#undef SUM_CASE
            default: return fail(TSDF_ERR_INVALID, "unknown kernel variant %d", variant);
        }
    } else {
        switch (variant - 16) {
#define TILE_CASE(code, R, E, N) case code: launch_ladder<R, E, N, false, false>(v, p); break;
            TILE_CASE(0, 1, false, false) TILE_CASE(1, 1, false, true)
            TILE_CASE(2, 1, true, false)  TILE_CASE(3, 1, true, true)
            TILE_CASE(4, 2, false, false) TILE_CASE(5, 2, false, true)
            TILE_CASE(6, 2, true, false)  TILE_CASE(7, 2, true, true)
            TILE_CASE(8, 4, false, false) TILE_CASE(9, 4, false, true)
            TILE_CASE(10, 4, true, false) TILE_CASE(11, 4, true, true)
#undef TILE_CASE
            default: return fail(TSDF_ERR_INVALID, "unknown kernel variant %d", variant);
        }
    }
    HIP_TRY(hipGetLastError());
    return TSDF_OK;
}

// Variant 11, one masked frame into a flat-mapped volume: a class per 1024-voxel workgroup patch ahead of the launch.
int launch_single_experiment(tsdf_volume *v, tsdfk::IntegrateParams &common, tsdfk::FramePose &pose, const float *depth_dev,
                             const float *c2b, bool *handled)
{
    *handled = false;
    if (!(v->variant == 11 && v->flat && pose.mask != nullptr && classify_one_frame(v, v->n_vox) && tiles_fit(common))) return TSDF_OK;
    const int nz = v->cfg.z_end - v->cfg.z_begin;
    dim3 block(64, 4, 1), grid((v->chunks_per_slice + 3) / 4, 1, nz);
    tsdfk::IntegrateParams cp = make_params(v, depth_dev, pose.mask, c2b, 4);
    int rc = classify_single(v, cp, (int)grid.x, 1, nz, 0);
    if (rc) return rc;
    common.wg_class = cp.wg_class;
    hipLaunchKernelGGL((tsdfk::integrate_multi_single_cls<true, true, true>), grid, block, 0, v->stream, common, pose);
    HIP_TRY(hipGetLastError());
    *handled = true;
    return tables_end(v);
}

// Fused launches of the measurement build: staged frame blocks (4, 5, 6), rows classified per workgroup (11), brick
// workgroups over the whole slab (12, 13).  *handled = a launch was queued; *claims_total = its wavefront- or
// workgroup-frames when it counts claims.
int launch_multi_experiment(tsdf_volume *v, tsdfk::MultiParamsInline &mi, const float *const *depth_dev, const uint8_t *const *masks_dev,
                            const float *c2b, int n, bool labels, bool any_mask, bool classify, bool *handled, double *claims_total)
{
    *handled = false;
    const tsdf_config &c = v->cfg;
    const int nz = c.z_end - c.z_begin;
    const dim3 block(64, 4, 1);
    mi.super_mask = nullptr;
    mi.nz_super = 1;
    if (!labels && (v->variant == 4 || v->variant == 5 || v->variant == 6)) {
        // frame blocks staged in device memory: pinned host ring -> device ring
        const int s = v->frames_next;
        v->frames_next = (s + 1) % kStageSlots;
        const size_t bytes = tsdfk::kMaxFramesPerLaunch * sizeof(tsdfk::FramePose);
        if (!v->h_frames[s]) {
            HIP_TRY(hipHostMalloc((void **)&v->h_frames[s], bytes, hipHostMallocDefault));
            HIP_TRY(hipMalloc((void **)&v->d_frames[s], bytes));
            HIP_TRY(hipEventCreateWithFlags(&v->frames_done[s], hipEventDisableTiming));
        }
        if (v->frames_used[s]) HIP_TRY(hipEventSynchronize(v->frames_done[s]));
        tsdfk::MultiParams mp;
        mp.common = mi.common;
        mp.common.claim_counter = nullptr;
        mp.frames = v->d_frames[s];
        mp.n_frames = n;
        for (int f = 0; f < n; ++f) { v->h_frames[s][f] = mi.frames[f]; v->h_frames[s][f].tiles = nullptr; }
        HIP_TRY(hipMemcpyAsync(v->d_frames[s], v->h_frames[s], n * sizeof(tsdfk::FramePose), hipMemcpyHostToDevice, v->stream));
        if (v->flat) {
            dim3 grid((v->chunks_per_slice + 3) / 4, 1, nz);
            hipLaunchKernelGGL((tsdfk::integrate_multi<1, true, true>), grid, block, 0, v->stream, mp);
        } else if (v->variant == 5) {   // XCD-aware block order
            dim3 grid((mp.common.xgroups + 63) / 64, (c.dim_y + 3) / 4, nz);
            hipLaunchKernelGGL((tsdfk::integrate_multi_xcd<true>), grid, block, 0, v->stream, mp);
        } else if (v->variant == 4) {   // two rows per lane
            dim3 grid((mp.common.xgroups + 63) / 64, (c.dim_y + 7) / 8, nz);
            hipLaunchKernelGGL((tsdfk::integrate_multi<2, true, false>), grid, block, 0, v->stream, mp);
        } else {                        // variant 6: the default kernel with staged frame blocks (A/B of the kernarg path)
            dim3 grid((mp.common.xgroups + 63) / 64, (c.dim_y + 3) / 4, nz);
            hipLaunchKernelGGL((tsdfk::integrate_multi<1, true, false>), grid, block, 0, v->stream, mp);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(v->frames_done[s], v->stream));
        v->frames_used[s] = true;
        *handled = true;
        *claims_total = 0.0;
        return TSDF_OK;
    }
    if (!classify || !(v->variant >= 11 && v->variant <= 13)) return TSDF_OK;
    dim3 grid_flat((v->chunks_per_slice + 3) / 4, 1, nz), grid_rows((mi.common.xgroups + 63) / 64, (c.dim_y + 3) / 4, nz);
    const int nz_groups = (nz + mi.common.brick_s - 1) / mi.common.brick_s;
    if (v->variant == 11) {   // rows / 1024 consecutive voxels, classified per workgroup in the prologue
        if ((v->flat ? grid_flat.x : grid_rows.x) > 65535u) mi.z_fastest = 0;
        if (mi.z_fastest) { std::swap(grid_flat.x, grid_flat.z); std::swap(grid_rows.x, grid_rows.z); }
        if (v->flat && any_mask)
            hipLaunchKernelGGL((tsdfk::integrate_multi_wg<1, true, true, false, true, true>), grid_flat, block, 0, v->stream, mi);
        else if (any_mask)
            hipLaunchKernelGGL((tsdfk::integrate_multi_wg<1, true, false, false, true, true>), grid_rows, block, 0, v->stream, mi);
        else if (v->flat)
            hipLaunchKernelGGL((tsdfk::integrate_multi_wg<1, true, true, false, false, true>), grid_flat, block, 0, v->stream, mi);
        else
            hipLaunchKernelGGL((tsdfk::integrate_multi_wg<1, true, false, false, false, true>), grid_rows, block, 0, v->stream, mi);
        const dim3 &g = v->flat ? grid_flat : grid_rows;
        *claims_total = (double)g.x * g.y * g.z * n;
    } else {                  // 12, 13: a workgroup per four bricks of the whole slab, each wavefront classifying its brick
        const unsigned wgs = (unsigned)(((int64_t)mi.common.brick_groups * mi.common.bricks_per_group + 3) / 4);
        if (wgs > 65535u) mi.z_fastest = 0;   // the slow grid dimensions hold 65535 at most
        const dim3 grid_bricks = mi.z_fastest ? dim3((unsigned)nz_groups, 1, wgs) : dim3(wgs, 1, (unsigned)nz_groups);
        if (v->variant == 13) {
            mi.nz_super = (nz_groups + tsdfk::kSuperZ - 1) / tsdfk::kSuperZ;
            const size_t words = (size_t)wgs * mi.nz_super;
            if (v->super_words < words) {
                if (v->d_super) HIP_TRY(hipFree(v->d_super));
                v->d_super = nullptr;
                v->super_words = 0;
                HIP_TRY(hipMalloc((void **)&v->d_super, words * sizeof(unsigned int)));
                v->super_words = words;
            }
            hipLaunchKernelGGL(tsdfk::classify_superbricks, dim3((unsigned)((words + 3) / 4)), block, 0, v->stream, mi, v->d_super, (int)wgs);
            mi.super_mask = v->d_super;
        }
        if (labels)
            hipLaunchKernelGGL((tsdfk::integrate_multi_wg<1, true, false, true, false, true, true>), grid_bricks, block, 0, v->stream, mi);
        else if (any_mask)
            hipLaunchKernelGGL((tsdfk::integrate_multi_wg<1, true, false, false, true, true, true>), grid_bricks, block, 0, v->stream, mi);
        else
            hipLaunchKernelGGL((tsdfk::integrate_multi_wg<1, true, false, false, false, true, true>), grid_bricks, block, 0, v->stream, mi);
        *claims_total = (double)grid_bricks.x * grid_bricks.y * grid_bricks.z * n * 4.0;
    }
    HIP_TRY(hipGetLastError());
    *handled = true;
    return TSDF_OK;
}
