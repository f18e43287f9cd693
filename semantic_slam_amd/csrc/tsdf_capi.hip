// tsdf_capi.hip -- implementation of include/tsdf_hip.h (libtsdf_hip.so).
//
// One handle = one z-slab of the voxel grid resident in HBM on one device + one HIP stream.
// Host work per frame is the 4x4 pose composition (pose_math.h) and one kernel launch; the
// depth frame is staged through a small ring of pinned buffers so the caller's pointer can be
// released as soon as tsdf_integrate returns (ref: the reference's blocking cudaMemcpy of the
// caller's buffer, src/tsdf.cu:162).
//
// There is no CPU fallback: without a HIP device every compute entry point fails with
// TSDF_ERR_NO_DEVICE / TSDF_ERR_HIP.
#include "../../include/tsdf_hip.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <unordered_map>
#include <vector>

#include "pose_math.h"
#include "host_derive.h"
#include "host_copy.h"
#include "frame_store.h"
#include "tsdf_kernels.hip.h"
#include "tsdf_multiframe.hip.h"
#include "tsdf_labels.hip.h"
#include "tsdf_colour.hip.h"
#include "tsdf_extract.hip.h"
#ifdef TSDF_EXPERIMENTS
#include "tsdf_experiments.hip.h"
#endif

namespace {

thread_local std::string g_last_error;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return fail(TSDF_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),   \
                        __FILE__, __LINE__);                                                   \
    } while (0)

constexpr int kStageSlots = 3;

// The measurement build (-DTSDF_EXPERIMENTS, `make experiments`) adds the kernel variants of tsdf_experiments.hip.h.
#ifdef TSDF_EXPERIMENTS
constexpr bool kExperiments = true;
#else
constexpr bool kExperiments = false;
#endif

// Volume accesses of the brick kernels are ordinary (temporal) loads and stores, not the non-temporal ones of the row-shaped
// kernels: a brick's row piece is 32 bytes (two lanes), the four x-neighbours a workgroup takes make up a 128-byte line between
// them, and only with the line kept in L2 do their pieces leave as ONE memory request.  With the nt hint every piece went out
// alone: PMC WRITE_SIZE 1 064 MB per 512^3 launch for 537 MB of weights.  Same box, nt -> temporal: the all-free-space launch
// 0.0160 -> 0.0126 ms per frame, S-surf 512^3 0.0387 -> 0.0370, fr3 trajectory 1024^3 0.186 -> 0.178, S-surf 200^3 0.0077 -> 0.0072.
#ifndef TSDF_BRICK_NT
#define TSDF_BRICK_NT 0
#endif
constexpr bool kBrickNT = TSDF_BRICK_NT != 0;
// one launch's counter block: the sharded counters / list heads, then the frames' table for classify_patch
constexpr size_t kClaimBlockBytes = (tsdfk::kCounterBytes + sizeof(tsdfk::ClassPoseTable) + 255) / 256 * 256;
// How classify_brick_list deals super-bricks to the 64 sub-lists (tsdf_multiframe.hip.h, list_bucket): 0 = hashed, 1 = the XCD
// by image-row wedges.
#ifndef TSDF_WEDGE_MODE
#define TSDF_WEDGE_MODE 0
#endif
constexpr int kWedgeMode = TSDF_WEDGE_MODE;

}  // namespace

struct tsdf_volume {
    tsdf_config cfg;
    float base2world_inv[16];
    float last_cam2base[16];
    int64_t n_vox;           // voxels in the slab
    float *d_tsdf;
    float *d_weight;
    hipStream_t own_stream;
    hipStream_t stream;      // the stream work is queued on (own_stream unless overridden)
    // Staging and deferral memory is not the handle's: pinned frames, frame / mask slots in HBM and depth tile tables come from
    // the store all handles of this device and image size share (frame_store.h) and go back to it by stream order.
    tsdf_store::FrameStore *store;
    // the H2D copy of a frame runs on its own stream and overlaps the previous frame's kernel; the kernel waits for it
    hipStream_t copy_stream;
    hipEvent_t flush_done[2];         // recorded on `stream` after every launch that read store slots: what they are released after;
    int flush_parity;                 // two, used in turn, so that a slot of pass k is not held until pass k + 1 has run as well
    int table_slot;                   // the store's table slot of the launch being queued, or -1
    // Deferred integration of host frames (tsdf_integrate): the reference reads results back only in its destructor
    // (ref: src/tsdf.cu:101-104) and this library only at download / extraction / save, so the frames of successive
    // calls are collected -- copied into a pool in HBM, poses composed at call time -- and applied defer_n at a time as
    // ONE fused, classified sequence launch; every entry point that observes or changes the volume flushes first.
    int defer_n;                      // frames per deferred launch (<= 1: every call launches; default kMaxFramesPerLaunch)
    hipEvent_t pend_copied;
    int pend_count;
    int pend_slot[tsdfk::kMaxFramesPerLaunch], pend_mask_slot[tsdfk::kMaxFramesPerLaunch];   // store slots of the collected frames (-1: none)
    const float *pend_depth[tsdfk::kMaxFramesPerLaunch];
    struct tsdf_batch *owner;   // the batch this handle is a member of (its collected frames come first), or null
    struct tsdf_group *group_owner;   // the group this handle is a slab of (likewise), or null
    bool in_flush;
    float pend_c2b[16 * tsdfk::kMaxFramesPerLaunch];
    const uint8_t *pend_mask[tsdfk::kMaxFramesPerLaunch];   // instance masks of collected masked frames (in the store's mask slots)
    int variant;
    // per-launch frame blocks of integrate_multi: pinned host ring -> device ring (allocated on first use)
    tsdfk::FramePose *h_frames[kStageSlots];
    tsdfk::FramePose *d_frames[kStageSlots];
    hipEvent_t frames_done[kStageSlots];
    bool frames_used[kStageSlots];
    int frames_next;
    // per-voxel label fusion (allocated by tsdf_labels_enable)
    uint16_t *d_label;
    float *d_fp, *d_bp;
    float prob_thd;
    // per-voxel colour (allocated by tsdf_colour_enable): packed 0x00BBGGRR; staging of the RGB frames of tsdf_integrate_rgbd
    uint32_t *d_colour;
    uint8_t *d_rgb[kStageSlots];
    uint8_t *h_rgb[kStageSlots];
    hipEvent_t rgb_done[kStageSlots];
    bool rgb_used[kStageSlots];
    int rgb_next;
    // free-space summary (one word per 256-voxel row segment), see tsdf_kernels.hip.h
    uint32_t *d_flags;
    size_t n_flags;
    int nseg;              // chunks per row when dim_x % 256 == 0 (row-mapped kernels), else 0
    int chunks_per_slice;  // ceil(dim_x*dim_y / 256)
    bool flat;             // dim_x % 256 != 0: summary-maintaining launches use the flat mapping
    int brick_q, brick_r, brick_s;   // wavefront brick of the classified launches (choose_brick / tsdf_set_brick_shape); q = 0: none
    int tile;                        // pixels per edge of the depth tiles of this handle's classified launches (tile_edge_for)
    bool fine_tables = false;        // ... and, beside them, 4-pixel tiles for brick-sized boxes (fine_tables_for)
    bool flags_known_zero;
    unsigned int *d_super;       // per super-brick frame words of the current fused brick launch (classify_superbricks)
    size_t super_words;
    uint4 *d_work;               // work list of the current fused brick launch: {brick, slice group, free frames, skipped frames}
    size_t work_entries;         // per live brick (classify_brick_list); capacity = every brick of the slab
    int64_t work_nsuper, work_bucket_supers;   // super-bricks of the shape the list was sized for; most of them in one sub-list
    int work_wedge_mode = -1;                  // ... and the list_bucket mode they were counted under
    // optional diagnostic counters (tsdf_shortcut_stats)
    unsigned int *d_shortcut_stats;
    // adaptive use of the classification: claims of the last classifying launch, read back without blocking
    unsigned long long *d_claims, *h_claims;   // d_claims: TWO counter blocks (+ frame table each), one per list parity
    hipEvent_t claims_done;
    // A sequence call's launches are pipelined: the pre-pass of launch k + 1 (tile tables, classify_brick_list: small grids, 11 % of
    // a 512^3 S-surf launch, reading only the frames and the poses) runs on pre_stream beside launch k's Integrate kernel.  Two work
    // lists, two counter blocks, two table slots (list_parity); pre_done[p]: pre-pass of parity p queued on pre_stream; list_free[p]:
    // the Integrate kernel that read parity p's list has been queued on the handle's stream; seq_ready: a sequence's inputs.
    hipStream_t pre_stream = nullptr, rb_stream = nullptr;   // rb_stream: the counters' read-back of a pipelined launch
    int rb_parity = -1;                                      // parity of the block the read-back in flight on rb_stream reads, or -1
    hipEvent_t pre_done[2] = {nullptr, nullptr}, list_free[2] = {nullptr, nullptr}, seq_ready = nullptr;
    int list_parity = 0;
    bool list_used[2] = {false, false};
    // Pipelining pays where a launch is short enough for its small pre-pass kernels to matter and the chip is not full: measured,
    // same box, sequence path, pre-pass on the handle's stream / beside the previous launch: S-surf 200^3 0.00578 -> 0.00546 ms per
    // frame, 320^3 0.01043 -> 0.01037, 512^3 0.02396 -> 0.02403 (the Integrate kernel runs 76 us longer beside a pre-pass that takes
    // 84 us alone: its latency-bound wavefronts hold slots the Integrate kernel's would use), fr3 trajectory 1024^3 0.1460 -> 0.1458.
    // So: slabs below 64 M voxels; larger ones keep one list (the second would be 134 MB at 1024^3) and one stream.
    bool pipeline_ok = false;
    bool claims_pending, claims_known;
    double claims_total;        // workgroup-frames of the launch the pending read-back belongs to
    double claim_fraction;      // claimed / total of the last launch that was read back
    int launches_unclassified;  // since the last classifying launch
    // scratch for surface extraction (allocated on first use)
    void *d_scratch;
    size_t scratch_bytes;
    // class of every workgroup of a one-frame masked launch (classify_workgroups), grown on demand
    uint8_t *d_wg_class;
    size_t wg_class_bytes;
    // output list of the extraction passes (points / vertices / triangles), grown on demand and kept
    void *d_list;
    size_t list_bytes;
};

constexpr int kBatchSideStreams = 4;

// Many volumes integrated by one launch per frame (include/tsdf_hip.h, tsdf_batch_*).
struct tsdf_batch {
    int device;
    std::vector<tsdf_volume *> vols;
    hipStream_t stream;
    // per-frame parameter blocks: pinned host ring -> device ring
    tsdfk::IntegrateParams *h_params[kStageSlots];
    tsdfk::IntegrateParams *d_params[kStageSlots];
    tsdfk::FramePose *h_poses[kStageSlots];
    tsdfk::FramePose *d_poses[kStageSlots];
    hipEvent_t slot_done[kStageSlots];
    bool slot_used[kStageSlots];
    int slot_next;
    int2 *d_slice_map;
    int total_slices, max_blocks;
    int2 *d_group_map;           // {object, slice group}: the brick launches' z index (a brick spans brick_s slices)
    int total_groups;
    std::vector<int> group_s;    // the brick_s of every member the group map was built with
    // per-object depth tile tables of the current frame and the class of every workgroup of the launch
    float2 *d_tiles;
    size_t tiles_per_object;
    uint8_t *d_wg_class;
    uint8_t *d_brick_class;      // class per wavefront brick of the launch (classify_bricks_batched), on first use
    size_t brick_class_bytes;
    // Deferred integration of a batch of large members (tsdf_batch_integrate_device): frames are collected in HBM -- the
    // depth image once, every member's mask beside it -- and applied by ONE fused launch per member per 32 frames
    // (the volume then moves once per 32 frames instead of once per frame); flushed by anything that observes a member.
    float *d_depth_pool;                     // kMaxFramesPerLaunch frames
    uint8_t *d_mask_pool;                    // kMaxFramesPerLaunch x members masks
    std::vector<float> pend_c2b;             // [member][frame][16], composed at collection time
    std::vector<const uint8_t *> pend_mask;  // [member][frame], null = the member's frame has no mask
    int pend_count;
    bool in_flush;
    // the members' fused launches of a flush are independent of each other: they go out on a few side streams (forked
    // from and joined back into the batch's stream by events), so one member's tail and table kernels overlap the next
    // member's launch
    hipStream_t side[kBatchSideStreams];
    hipEvent_t side_done[kBatchSideStreams];
    hipEvent_t collected;
};

namespace {

int flush_pending(tsdf_volume *v);
int fail(int code, const char *fmt, ...);
int batch_flush(tsdf_batch *b);
int group_flush(tsdf_group *g);   // tsdf_group.hip.h
int batch_collect(tsdf_batch *b, const float *depth_dev, const uint8_t *const *masks_dev, const float cam2world[16]);

// Every entry point starts here: make the handle's device current and, unless the caller is the collecting call itself,
// apply the host frames tsdf_integrate has collected (deferred integration, see struct tsdf_volume).
int bind_device(tsdf_volume *v, bool flush = true)
{
    HIP_TRY(hipSetDevice(v->cfg.device));
    if (v->owner && !v->in_flush) {   // frames its batch has collected were given first
        int rc = batch_flush(v->owner);
        if (rc) return rc;
    }
    if (v->group_owner && !v->in_flush) {   // likewise the frames its group has collected (a slab handle borrowed with tsdf_group_volume)
        int rc = group_flush(v->group_owner);
        if (rc) return rc;
    }
    if (flush && v->pend_count > 0 && !v->in_flush) return flush_pending(v);
    return TSDF_OK;
}

// ---- the shared store (frame_store.h) -----------------------------------------------------------------------------
// A frame (c = frames) or mask (c = masks) slot whose first access is queued on `first_user`.  When the store is at its soft
// cap and every slot is held by collecting handles, this handle applies what IT has collected (never another handle's: that
// one may be in use on another thread) and asks again.
int store_slot(tsdf_volume *v, tsdf_store::SlotClass *c, hipStream_t first_user, int *index, void **dev)
{
    hipError_t e = tsdf_store::slot_acquire(v->store, c, v, first_user, false, index, dev);
    if (e == hipSuccess && *index < 0) {
        if (v->pend_count > 0 && !v->in_flush) {
            int rc = flush_pending(v);
            if (rc) return rc;
        }
        e = tsdf_store::slot_acquire(v->store, c, v, first_user, true, index, dev);
    }
    if (e != hipSuccess) return fail(TSDF_ERR_HIP, "frame store: %s", hipGetErrorString(e));
    return TSDF_OK;
}

// Everything queued on the handle's stream so far has read its store slots: they go back after flush_done.
int store_release(tsdf_volume *v, tsdf_store::SlotClass *c, const int *idx, int n)
{
    hipEvent_t *evt = &v->flush_done[v->flush_parity];
    HIP_TRY(hipEventRecord(*evt, v->stream));
    tsdf_store::slots_release_after(v->store, c, idx, n, evt, v);
    return TSDF_OK;
}

// the pass (a flush of collected frames, or one one-kernel-per-call frame) is queued: the next one records the other event
void store_next_pass(tsdf_volume *v) { v->flush_parity ^= 1; }

// The depth tile tables of the launch being queued (kMaxFramesPerLaunch tables): a table slot of the store, first written on the
// handle's stream; tables_end() after the launch that reads them has been queued.
int tables_begin(tsdf_volume *v, float2 **tiles, hipStream_t first_user = nullptr)
{
    void *dev = nullptr;
    hipError_t e = tsdf_store::slot_acquire(v->store, &v->store->tables, v, first_user ? first_user : v->stream, true, &v->table_slot, &dev);
    if (e != hipSuccess) return fail(TSDF_ERR_HIP, "frame store (tile tables): %s", hipGetErrorString(e));
    *tiles = static_cast<float2 *>(dev);
    return TSDF_OK;
}

int tables_end(tsdf_volume *v)
{
    if (v->table_slot < 0) return TSDF_OK;
    const int slot = v->table_slot;
    v->table_slot = -1;
    return store_release(v, &v->store->tables, &slot, 1);
}

// The wavefront brick of classified launches and the library's choice of it: host arithmetic only (host_derive.h; also behind
// tsdf_default_brick_shape, which needs no device).
using tsdf_host::brick_shape_ok;

void choose_brick_for(const tsdf_config &c, int &bq, int &br, int &bs)
{
#ifdef TSDF_EXPERIMENTS
    if (const char *e = std::getenv("TSDF_BRICK3D")) {      // A/B knob of the measurement build: "q,r,s" (the product: tsdf_set_brick_shape)
        int q = 0, r = 0, sl = 0;
        if (c.dim_x % 4 == 0 && std::sscanf(e, "%d,%d,%d", &q, &r, &sl) == 3 && brick_shape_ok(c, q, r, sl)) {
            bq = q; br = r; bs = sl;
            return;
        }
    }
#endif
    tsdf_host::choose_brick_default(c, bq, br, bs);
}

void choose_brick(tsdf_volume *v) { choose_brick_for(v->cfg, v->brick_q, v->brick_r, v->brick_s); }

tsdfk::IntegrateParams make_params(const tsdf_volume *v, const float *depth_dev,
                                   const uint8_t *mask_dev, const float *c2b, int vx)
{
    const tsdf_config &c = v->cfg;
    tsdfk::IntegrateParams p;
    p.depth = depth_dev;
    p.mask = mask_dev;
    p.tsdf = v->d_tsdf;
    p.weight = v->d_weight;
    p.fx = c.cam_K[0]; p.fy = c.cam_K[4]; p.cx = c.cam_K[2]; p.cy = c.cam_K[5];
    p.rx0 = c2b[0]; p.rx1 = c2b[4]; p.rx2 = c2b[8];
    p.ry0 = c2b[1]; p.ry1 = c2b[5]; p.ry2 = c2b[9];
    p.rz0 = c2b[2]; p.rz1 = c2b[6]; p.rz2 = c2b[10];
    p.tx = c2b[3]; p.ty = c2b[7]; p.tz = c2b[11];
    p.ox = c.origin[0]; p.oy = c.origin[1]; p.oz = c.origin[2];
    p.vs = c.voxel_size; p.trunc = c.trunc_margin; p.max_depth = c.max_depth;
    p.dim_x = c.dim_x; p.dim_y = c.dim_y;
    p.nz = c.z_end - c.z_begin; p.z_begin = c.z_begin;
    p.H = c.im_height; p.W = c.im_width;
    p.xgroups = (c.dim_x + vx - 1) / vx;
    p.flags = v->d_flags;
    p.nseg = v->nseg;
    p.quads_per_row = c.dim_x / 4;
    p.quads_per_slice = (int)((int64_t)c.dim_x * c.dim_y / 4);
    p.chunks_per_slice = v->chunks_per_slice;
    // brick view (tsdf_multiframe.hip.h, BRICK): the volume's wavefront brick, chosen at creation (choose_brick)
    p.brick_q = v->brick_q; p.brick_r = v->brick_r; p.brick_s = v->brick_s > 0 ? v->brick_s : 1;
    p.bricks_per_group = 0; p.brick_groups = 0;
    p.brick_q_magic = 0; p.brick_per_magic = 0;
    if (p.brick_q > 0) {
        p.bricks_per_group = p.quads_per_row / p.brick_q;
        p.brick_groups = (c.dim_y + p.brick_r - 1) / p.brick_r;
        p.brick_q_magic = 65536 / p.brick_q + 1;
        p.brick_per_magic = 65536 / (p.brick_q * p.brick_r) + 1;
    }
    // guards of the exact shortcuts and margins of the box claims: host_derive.h (derive_projection_guards)
    const tsdf_host::ProjectionGuards g = tsdf_host::derive_projection_guards(c, c2b);
    p.cz_margin = g.cz_margin; p.fast_ok = g.fast_ok; p.trunc_fast = g.trunc_fast;
    p.cz_short = g.cz_short; p.cz_pad = g.cz_pad;
    p.tiles_w = (c.im_width + v->tile - 1) / v->tile;
    p.tiles_h = (c.im_height + v->tile - 1) / v->tile;
    p.tile_inv = 1.0f / (float)v->tile;
    p.fine = nullptr;                // set by the launch that builds the fine tables
    p.fine_w = v->fine_tables ? (c.im_width + tsdfk::kFineTile - 1) / tsdfk::kFineTile : 0;
    p.fine_h = v->fine_tables ? (c.im_height + tsdfk::kFineTile - 1) / tsdfk::kFineTile : 0;
    p.px_margin_u = g.px_margin_u; p.px_margin_v = g.px_margin_v;
    p.shortcut_stats = v->d_shortcut_stats;
    p.claim_counter = nullptr;
    p.wg_class = nullptr;
    return p;
}

// The per-frame block of the fused / flat kernels from a full parameter set.
void pose_from_params(tsdfk::FramePose &fp, const tsdfk::IntegrateParams &q)
{
    fp.depth = q.depth; fp.mask = q.mask;
    fp.rx0 = q.rx0; fp.rx1 = q.rx1; fp.rx2 = q.rx2;
    fp.ry0 = q.ry0; fp.ry1 = q.ry1; fp.ry2 = q.ry2;
    fp.rz0 = q.rz0; fp.rz1 = q.rz1; fp.rz2 = q.rz2;
    fp.tx = q.tx; fp.ty = q.ty; fp.tz = q.tz;
    fp.fast_ok = q.fast_ok; fp.cz_margin = q.cz_margin;
    fp.label_im = nullptr; fp.score_im = nullptr;
    fp.tiles = nullptr;
    fp.cz_short = q.cz_short; fp.cz_pad = q.cz_pad;
}

int tile_levels_host(int n) { int l = 0; while (n > 0) { ++l; n >>= 1; } return l; }

size_t tile_table_elems_host(int tiles_w, int tiles_h)
{
    return (size_t)tile_levels_host(tiles_w) * tile_levels_host(tiles_h) * tiles_w * tiles_h;
}

bool tiles_fit(const tsdfk::IntegrateParams &p) { return (int64_t)p.tiles_w * p.tiles_h <= 16384; }

// Pixels per depth tile edge for a slab: 8 where a fused launch is long enough to repay tables four times as large (the finer
// tiles leave a fifth fewer wavefront-frames to the per-voxel path: tsdf_multiframe.hip.h), 16 otherwise and wherever the finer
// grid of tiles would not fit the table kernels.  Measured on S-surf, ms per frame with 16 / 8: 128^3 0.0044 / 0.0051 (before
// the table kernel ran 1024 threads), 200^3 0.00586 / 0.00586, 224^3 0.00642 / 0.00633, 256^3 0.00764 / 0.00741, 288^3 0.0097 /
// 0.0093, 320^3 0.0114 / 0.0106, 512^3 0.0312 / 0.0270: the finer tiles pay from about 10 M voxels.  Members of a batch share one
// table layout and keep 16.
constexpr int64_t kFineTileMinVoxels = 10000000;
thread_local bool g_create_for_batch = false;
int tile_edge_for(const tsdf_config &c)
{
    const int64_t n = (int64_t)c.dim_x * c.dim_y * (int64_t)(c.z_end - c.z_begin);
    const int64_t fine_tiles = (int64_t)((c.im_width + 7) / 8) * ((c.im_height + 7) / 8);
    return (!g_create_for_batch && n >= kFineTileMinVoxels && fine_tiles <= tsdfk::kTileLdsEntries) ? 8 : 16;
}

// Fine (4-pixel) tiles beside the 8-pixel tables: where a fused launch is long enough to repay two more small table kernels and
// 1.4 MB more table per frame (tsdf_multiframe.hip.h, fine_tile_levels).
constexpr int64_t kFineLevelMinVoxels = 64000000;      // measured: 512^3 S-surf 0.0255 -> 0.0240 ms per frame, 320^3 0.0102 -> 0.0105
bool fine_tables_for(const tsdf_config &c, int tile)
{
    const int64_t n = (int64_t)c.dim_x * c.dim_y * (int64_t)(c.z_end - c.z_begin);
    bool on = tile == 8 && n >= kFineLevelMinVoxels;
#ifdef TSDF_EXPERIMENTS
    if (const char *e = std::getenv("TSDF_FINE_TILES")) on = tile == 8 && std::atoi(e) != 0;     // A/B knob of the measurement build
#endif
    return on;
}

// One-frame masked launches are classified per workgroup when the launch is large enough to repay the three small
// dependent dispatches ahead of it (tile summary, sparse table, class table: ~25 us on the stream).  Measured
// (tools/batch_time.py, instance masks over 12 % of the image): 16 x 200^3 batched 0.275 -> 0.157 ms per frame (wavefront
// bricks; 0.201 with 1024-voxel workgroup patches), one 400^3 volume 0.095 -> 0.075; but 4 x 200^3 batched 0.083 -> 0.079
// at best and one 200^3 volume 0.016 -> 0.026.
// Variant 8 classifies regardless (tests), 7 never.
constexpr int64_t kClassifyMinVoxels = 48000000;
bool classify_one_frame(const tsdf_volume *v, int64_t launch_voxels)
{
    if (v->variant == 7) return false;
    return v->variant == 8 || (kExperiments && v->variant >= 11 && v->variant <= 13) || launch_voxels >= kClassifyMinVoxels;
}

// Depth tile tables (summary + sparse table, tsdf_multiframe.hip.h) of n images depth[i] x mask[i] into tables[i], queued
// on `stream`; two small launches per 32 images.
// fine (may be null; 8-pixel tiles only): the launch's fine tables (p.fine_w x p.fine_h tiles of 4 pixels, nine levels per frame),
// built by the same two launches whenever the coarse tables fit tile_sparse_table's LDS, else by two more.
// beside_a_launch: the kernels are queued on a side stream while another launch fills the chip -- the sparse-table kernel then takes
// 256-thread workgroups (a 1024-thread one needs sixteen free wavefront slots on ONE compute unit at once and waited for the other
// launch to drain: 634 us instead of 20).
int build_tile_tables(hipStream_t stream, const tsdf_config &c, const tsdfk::IntegrateParams &p, const float *const *depth,
                      const uint8_t *const *masks, int n, float2 *tables, unsigned long long *zero_me = nullptr, float2 *fine = nullptr,
                      bool beside_a_launch = false)
{
    const size_t per = tile_table_elems_host(p.tiles_w, p.tiles_h);
    if (fine != nullptr && (p.tile_inv != 0.125f || n > tsdfk::kMaxFramesPerLaunch))
        return fail(TSDF_ERR_INVALID, "build_tile_tables: fine tables go with 8-pixel tiles, one launch at a time");
    for (int k = 0; k < n; k += tsdfk::kMaxFramesPerLaunch) {
        const int m = std::min(tsdfk::kMaxFramesPerLaunch, n - k);
        tsdfk::TileSummaryParams tp;
        for (int f = 0; f < tsdfk::kMaxFramesPerLaunch; ++f) {
            tp.depth[f] = depth[k + (f < m ? f : 0)];
            tp.mask[f] = masks ? masks[k + (f < m ? f : 0)] : nullptr;
        }
        tp.tiles = tables + (size_t)k * per;
        tp.H = c.im_height; tp.W = c.im_width; tp.tiles_w = p.tiles_w; tp.tiles_h = p.tiles_h;
        tp.max_depth = c.max_depth;
        tp.fine = fine; tp.fw = p.fine_w; tp.fh = p.fine_h;      // level (0, 0) of the fine tables in the same pass (8-pixel tiles)
        // whole-row reads: one wavefront per strip of 64 pixels (four 16-pixel tiles or eight 8-pixel ones)
        if (p.tile_inv == 0.0625f) {
            const int strips = ((p.tiles_w + 3) / 4) * p.tiles_h;
            hipLaunchKernelGGL(tsdfk::depth_tile_summary<16>, dim3((unsigned)((strips + 3) / 4), m), dim3(64, 4), 0, stream, tp);
        } else {
            const int strips = ((p.tiles_w + 7) / 8) * p.tiles_h;
            hipLaunchKernelGGL(tsdfk::depth_tile_summary<8>, dim3((unsigned)((strips + 3) / 4), m), dim3(64, 4), 0, stream, tp);
        }
        if (p.tiles_w * p.tiles_h <= tsdfk::kTileLdsEntries) {
            // (1024 threads when the frame has thousands of tiles: each of the kernel's ~12 barrier-separated passes visits every tile)
            const unsigned threads = (p.tiles_w * p.tiles_h > 2048 && !beside_a_launch) ? 1024 : 256;
            const unsigned fine_blocks = fine ? ((unsigned)(p.fine_w * p.fine_h) + threads - 1) / threads : 0u;   // the fine tables' upper levels ride along
            hipLaunchKernelGGL(tsdfk::tile_sparse_table, dim3((unsigned)tile_levels_host(p.tiles_w) + fine_blocks, m), dim3(threads), 0, stream, tp.tiles,
                               p.tiles_w, p.tiles_h, k == 0 ? zero_me : (unsigned long long *)nullptr, fine, p.fine_w, p.fine_h);
        } else {
            if (zero_me && k == 0) HIP_TRY(hipMemsetAsync(zero_me, 0, tsdfk::kCounterBytes, stream));
            hipLaunchKernelGGL(tsdfk::tile_sparse_table_scan, dim3((unsigned)tile_levels_host(p.tiles_w), m), dim3(256), 0, stream, tp.tiles,
                               p.tiles_w, p.tiles_h);
            if (fine)
                hipLaunchKernelGGL(tsdfk::fine_tile_levels, dim3((unsigned)((p.fine_w * p.fine_h + 255) / 256), m), dim3(256), 0, stream, fine, p.fine_w, p.fine_h);
        }
    }
    HIP_TRY(hipGetLastError());
    return TSDF_OK;
}

// The same with wavefront bricks (rows that divide into them): class per brick, then the brick kernel.  Queues the launch.
int launch_masked_bricks(tsdf_volume *v, tsdfk::IntegrateParams &p)
{
    const int blocks = (p.brick_groups * p.bricks_per_group + 3) / 4, nz = (p.nz + p.brick_s - 1) / p.brick_s;   // slice groups
    float2 *tiles = nullptr;
    int rc0 = tables_begin(v, &tiles);
    if (rc0) return rc0;
    const size_t n_bricks = (size_t)blocks * nz * 4;
    if (v->wg_class_bytes < n_bricks) {
        if (v->d_wg_class) HIP_TRY(hipFree(v->d_wg_class));
        v->d_wg_class = nullptr;
        v->wg_class_bytes = 0;
        HIP_TRY(hipMalloc((void **)&v->d_wg_class, n_bricks));
        v->wg_class_bytes = n_bricks;
    }
    const float *d = p.depth;
    const uint8_t *m = p.mask;
    int rc = build_tile_tables(v->stream, v->cfg, p, &d, &m, 1, tiles);
    if (rc) return rc;
    tsdfk::FramePose pose;
    pose_from_params(pose, p);
    pose.tiles = tiles;
    hipLaunchKernelGGL(tsdfk::classify_bricks, dim3((unsigned)((n_bricks + 255) / 256)), dim3(256), 0, v->stream, p, pose,
                       v->d_wg_class, blocks, nz);
    p.wg_class = v->d_wg_class;
    hipLaunchKernelGGL((tsdfk::integrate_single_bricks<kBrickNT>), dim3(blocks, 1, nz), dim3(64, 4, 1), 0, v->stream, p, pose);
    HIP_TRY(hipGetLastError());
    return tables_end(v);
}

// Kernel variants (tsdf_set_kernel_variant) of the library as shipped:
//   0   default: one call = one launch of integrate_tile<2> (rows of a multiple of 256 voxels), of the flat kernel (other
//       rows of a multiple of 4 voxels) or of the scalar kernel (any other row); collected frames and frame sequences
//       (tsdf_integrate_frames_device, ..._sequence_timed) are applied up to kMaxFramesPerLaunch (32) per pass over the
//       volume -- over the brick work list when the depth tile tables earn their keep (decided per launch from the previous
//       launch's claims), else by the per-voxel fused kernel
//   3   as 0 but one launch per frame even for sequences
//   7   as 0 but never classified (the per-voxel fused kernel alone)
//   8   as 0 but always classified
//   1   the scalar kernel (any dim_x)
// Every other number belongs to the measurement build (-DTSDF_EXPERIMENTS: tsdf_experiments.hip.h, `make experiments`).

bool shipped_variant(int variant) { return variant == 0 || variant == 1 || variant == 3 || variant == 7 || variant == 8; }

// Kernels that do not maintain the free-space summary must not leave stale "all ones" flags behind.
int drop_summary(tsdf_volume *v)
{
    if (v->flags_known_zero || v->n_flags == 0) return TSDF_OK;
    HIP_TRY(hipMemsetAsync(v->d_flags, 0, v->n_flags * sizeof(uint32_t), v->stream));
    v->flags_known_zero = true;
    return TSDF_OK;
}

int rebuild_summary(tsdf_volume *v)
{
    if (v->n_flags == 0 || v->n_vox == 0) return TSDF_OK;
    const int nz = v->cfg.z_end - v->cfg.z_begin;
    const int qps = (int)((int64_t)v->cfg.dim_x * v->cfg.dim_y / 4);
    dim3 block(64, 4, 1), grid((v->chunks_per_slice + 3) / 4, 1, nz);
    hipLaunchKernelGGL(tsdfk::recompute_flags, grid, block, 0, v->stream, v->d_tsdf, v->d_weight, v->d_flags, qps,
                       v->chunks_per_slice);
    HIP_TRY(hipGetLastError());
    v->flags_known_zero = false;
    return TSDF_OK;
}

int launch_multi(tsdf_volume *v, const float *const *depth_dev, const uint8_t *const *masks_dev, const float *c2b, int n,
                 const uint16_t *const *label_ims = nullptr, const float *const *score_ims = nullptr, hipEvent_t inputs_ready = nullptr);

// Claimed share of the launch whose counter block has arrived in h_claims: (free << 32 | skipped) per bucket, added up.
double claims_read_back(const tsdf_volume *v)
{
    double claimed = 0.0;
    for (int b = 0; b < tsdfk::kListBuckets; ++b) {
        const unsigned long long w = v->h_claims[(size_t)b * (tsdfk::kBucketStride / sizeof(unsigned long long)) + 1];
        claimed += (double)(w >> 32) + (double)(w & 0xffffffffull);
    }
    return v->claims_total > 0 ? claimed / v->claims_total : 0.0;
}

#ifdef TSDF_EXPERIMENTS
bool experiment_variant(int variant);
void experiment_adjust(const tsdf_volume *v, bool labels, bool *classify, int *z_fastest);
int launch_integrate_experiment(tsdf_volume *v, const float *depth_dev, const uint8_t *mask_dev, const float *c2b);
int launch_multi_experiment(tsdf_volume *v, tsdfk::MultiParamsInline &mi, const float *const *depth_dev, const uint8_t *const *masks_dev,
                            const float *c2b, int n, bool labels, bool any_mask, bool classify, bool *handled, double *claims_total);
int launch_single_experiment(tsdf_volume *v, tsdfk::IntegrateParams &common, tsdfk::FramePose &pose, const float *depth_dev,
                             const float *c2b, bool *handled);
#endif

// Queue one Integrate launch.  Shapes are validated at tsdf_create, so the grid covers exactly
// the slab and every access stays inside the two allocations.
int launch_integrate(tsdf_volume *v, const float *depth_dev, const uint8_t *mask_dev,
                     const float *c2b)
{
    const tsdf_config &c = v->cfg;
    const int nz = c.z_end - c.z_begin;
    if (nz == 0) return TSDF_OK;  // empty slab: nothing to do
    std::memcpy(v->last_cam2base, c2b, sizeof v->last_cam2base);
#ifdef TSDF_EXPERIMENTS
    if (c.dim_x % 4 == 0 && (v->variant == 2 || (v->variant >= 16 && !mask_dev))) return launch_integrate_experiment(v, depth_dev, mask_dev, c2b);
#endif
    if (c.dim_x % 4 != 0 || v->variant == 1) {
        // rows that are not 16-byte aligned (or the scalar kernel asked for): one voxel per lane; it does not keep the summary
        int rc = drop_summary(v);
        if (rc) return rc;
        const tsdfk::IntegrateParams p = make_params(v, depth_dev, mask_dev, c2b, 1);
        dim3 block(64, 4, 1), grid((c.dim_x + 63) / 64, (c.dim_y + 3) / 4, nz);
        if (mask_dev) hipLaunchKernelGGL((tsdfk::integrate_scalar<true>), grid, block, 0, v->stream, p);
        else hipLaunchKernelGGL((tsdfk::integrate_scalar<false>), grid, block, 0, v->stream, p);
        HIP_TRY(hipGetLastError());
        return TSDF_OK;
    }
    if (v->flat) {
        // rows that are not a multiple of 256 voxels: the flat mapping (every lane busy, summary kept)
        return launch_multi(v, &depth_dev, mask_dev ? &mask_dev : nullptr, c2b, 1);
    }
    tsdfk::IntegrateParams p = make_params(v, depth_dev, mask_dev, c2b, 4);
    v->flags_known_zero = false;   // integrate_tile keeps the free-space summary up to date
    const dim3 block(64, 4, 1), grid((p.xgroups + 63) / 64, (p.dim_y + 7) / 8, p.nz);
    if (mask_dev) {
        // per-object volumes see their instance only, so the bricks the tile table of depth x mask proves untouched are
        // told to leave (variant 7: never; small launches: not worth the three small dispatches ahead of it)
        if (classify_one_frame(v, v->n_vox) && tiles_fit(p) && p.brick_q > 0 && (!kExperiments || v->variant != 11)) {
            int rc = launch_masked_bricks(v, p);    // per wavefront brick: skipped by rows, slices and columns
            if (rc) return rc;
        } else {
#ifdef TSDF_EXPERIMENTS
            if (v->variant == 11 && classify_one_frame(v, v->n_vox) && tiles_fit(p)) return launch_integrate_experiment(v, depth_dev, mask_dev, c2b);
#endif
            hipLaunchKernelGGL((tsdfk::integrate_tile<2, true>), grid, block, 0, v->stream, p);
        }
    } else {
        hipLaunchKernelGGL((tsdfk::integrate_tile<2, false>), grid, block, 0, v->stream, p);
    }
    HIP_TRY(hipGetLastError());
    return TSDF_OK;
}

void compose_cam2base(const tsdf_volume *v, const float *cam2world, float *c2b)
{
    tsdf_host::multiply_matrix(v->base2world_inv, cam2world, c2b);  // ref: src/tsdf.cu:142
}

// n frames (n <= kMaxFramesPerLaunch) in one pass over the slab.  c2b: n x 16 relative poses.
// label_ims / score_ims (both or neither): the frames' label evidence is fused in the same pass (LABELS kernels).
// inputs_ready (may be null): an event after which the frames (and masks) may be read from ANY stream -- a sequence call records it
// once on the handle's stream at its start; with it the pre-pass of a classified launch runs on the handle's side stream, i.e.
// beside the previous launch's Integrate kernel (see tsdf_volume::pre_stream).  Null: everything on the handle's stream.
int launch_multi(tsdf_volume *v, const float *const *depth_dev, const uint8_t *const *masks_dev,
                 const float *c2b, int n, const uint16_t *const *label_ims, const float *const *score_ims, hipEvent_t inputs_ready)
{
    const tsdf_config &c = v->cfg;
    const int nz = c.z_end - c.z_begin;
    if (nz == 0 || n == 0) return TSDF_OK;
    auto fill_pose = [&](tsdfk::FramePose &fp, int f) {
        pose_from_params(fp, make_params(v, depth_dev[f], masks_dev ? masks_dev[f] : nullptr, c2b + 16 * f, 4));
        fp.label_im = label_ims ? label_ims[f] : nullptr;
        fp.score_im = label_ims ? score_ims[f] : nullptr;
    };
    const dim3 block(64, 4, 1);
    if (n == 1 && !label_ims) {   // pose by value: nothing to stage
        tsdfk::IntegrateParams common = make_params(v, depth_dev[0], nullptr, c2b, 4);
        tsdfk::FramePose pose;
        fill_pose(pose, 0);
        std::memcpy(v->last_cam2base, c2b, sizeof v->last_cam2base);
        v->flags_known_zero = false;
#ifdef TSDF_EXPERIMENTS
        {
            bool handled = false;
            int rc = launch_single_experiment(v, common, pose, depth_dev[0], c2b, &handled);
            if (rc || handled) return rc;
        }
#endif
        if (v->flat && pose.mask != nullptr && classify_one_frame(v, v->n_vox) && tiles_fit(common) && common.brick_q > 0) {
            // one masked frame into a flat-mapped volume (the reference's 200^3 object grids), large enough to repay a class table
            tsdfk::IntegrateParams cp = make_params(v, depth_dev[0], pose.mask, c2b, 4);
            return launch_masked_bricks(v, cp);
        }
        if (v->flat) {
            dim3 grid((v->chunks_per_slice + 3) / 4, 1, nz);
            hipLaunchKernelGGL((tsdfk::integrate_multi_single<true, true>), grid, block, 0, v->stream, common, pose);
        } else {
            dim3 grid((common.xgroups + 63) / 64, (c.dim_y + 3) / 4, nz);
            hipLaunchKernelGGL((tsdfk::integrate_multi_single<true, false>), grid, block, 0, v->stream, common, pose);
        }
        HIP_TRY(hipGetLastError());
        return TSDF_OK;
    }
    std::memcpy(v->last_cam2base, c2b + 16 * (n - 1), sizeof v->last_cam2base);
    v->flags_known_zero = false;   // the fused kernels maintain the summary
    // the frame blocks travel in the kernarg: nothing is staged on the stream ahead of the launch
    tsdfk::MultiParamsInline mi;
    mi.common = make_params(v, depth_dev[0], nullptr, c2b, 4);
    mi.n_frames = n;
    for (int f = 0; f < n; ++f) fill_pose(mi.frames[f], f);
    for (int f = n; f < tsdfk::kMaxFramesPerLaunch; ++f) mi.frames[f] = mi.frames[0];
    mi.labels.label = v->d_label; mi.labels.fp = v->d_fp; mi.labels.bp = v->d_bp; mi.labels.prob_thd = v->prob_thd;
    bool any_mask = false;
    for (int f = 0; f < n && masks_dev; ++f) any_mask = any_mask || masks_dev[f] != nullptr;

    // Patch classification: tables of at most 4 MiB per frame.  Variant 8: always; variant 7: never;
    // default: while it pays -- the first launch classifies, every classifying launch counts its claims, and a launch whose
    // predecessor claimed less than a tenth of its wavefront-frames goes without (the tables and the pre-pass cost more than
    // that saves), with a new probe every eighth launch.  The count is read back asynchronously: a decision never waits
    // for the GPU.
    if (v->claims_pending) {
        const hipError_t qe = hipEventQuery(v->claims_done);
        if (qe == hipSuccess) {
            v->claim_fraction = claims_read_back(v);
            v->claims_pending = false;
            v->claims_known = true;
        } else if (qe == hipErrorNotReady) {
            (void)hipGetLastError();   // "not ready" is an answer, not an error to be found by a later check
        } else {
            // a real failure: forget the stale count (the next launch classifies and counts afresh) and report it
            v->claims_pending = false;
            v->claims_known = false;
            (void)hipGetLastError();
            return fail(TSDF_ERR_HIP, "claims read-back: hipEventQuery failed: %s", hipGetErrorString(qe));
        }
    }
    // classified launches run over wavefront bricks (every grid whose rows are a multiple of 4 voxels has a brick view)
    bool classify = tiles_fit(mi.common) && v->variant != 7 && mi.common.brick_q > 0;
    mi.z_fastest = 2;
#ifdef TSDF_EXPERIMENTS
    experiment_adjust(v, label_ims != nullptr, &classify, &mi.z_fastest);
#endif
    const bool forced = v->variant == 8 || (kExperiments && v->variant >= 11 && v->variant <= 13);
    if (classify && !forced) classify = !v->claims_known || v->claim_fraction >= 0.10 || v->launches_unclassified >= 7;
    v->launches_unclassified = classify ? 0 : v->launches_unclassified + 1;
    const bool count_claims = classify && !v->claims_pending;
    if (classify && !v->d_claims) {
        // the launch's counter block (tsdf_multiframe.hip.h, kListBuckets): per bucket the lengths of its brick sub-list and
        // its share of the claims, cleared by the table kernel below; behind it the frames' table for classify_patch
        // all three or none: a launch that finds d_claims set relies on the host mirror and the event being there too
        unsigned long long *dc = nullptr;
        unsigned char *hc = nullptr;
        hipEvent_t ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
        hipError_t e = hipMalloc((void **)&dc, 2 * kClaimBlockBytes);
        if (e == hipSuccess) e = hipHostMalloc((void **)&hc, tsdfk::kCounterBytes, hipHostMallocDefault);
        for (int i = 0; i < 6 && e == hipSuccess; ++i) e = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming);
        if (e != hipSuccess) {
            for (int i = 0; i < 6; ++i) if (ev[i]) (void)hipEventDestroy(ev[i]);
            if (hc) (void)hipHostFree(hc);
            if (dc) (void)hipFree(dc);
            return fail(TSDF_ERR_HIP, "claim counters: %s", hipGetErrorString(e));
        }
        v->d_claims = reinterpret_cast<decltype(v->d_claims)>(dc);
        v->h_claims = reinterpret_cast<decltype(v->h_claims)>(hc);
        v->claims_done = ev[0];
        v->pre_done[0] = ev[1]; v->pre_done[1] = ev[2]; v->list_free[0] = ev[3]; v->list_free[1] = ev[4]; v->seq_ready = ev[5];
    }
    // this launch's parity: its counter block, its half of the work list, its table slot
    // (slabs too large to be pipelined -- see pipeline_ok -- keep one list and one parity)
    const int P = v->pipeline_ok ? v->list_parity : 0;
    unsigned long long *const claims_p = classify ? v->d_claims + (size_t)P * (kClaimBlockBytes / sizeof(unsigned long long)) : nullptr;
    bool pipelined = classify && inputs_ready != nullptr && v->pipeline_ok;
#ifdef TSDF_EXPERIMENTS
    if (!shipped_variant(v->variant) || std::getenv("TSDF_NO_PIPELINE")) pipelined = false;   // the measurement build's own kernels run on the handle's stream
#endif
    if (pipelined && !v->pre_stream) {
        // the two side streams of a pipelined handle, at its first pipelined launch (a stream costs about 2 MiB of device memory:
        // the per-object handles of a scene, which are fed frame by frame and never pipeline, do not pay for them)
        hipStream_t a = nullptr, b = nullptr;
        int prio = 0;
#ifdef TSDF_EXPERIMENTS
        if (std::getenv("TSDF_PRE_PRIORITY")) { int lo = 0, hi = 0; (void)hipDeviceGetStreamPriorityRange(&lo, &hi); prio = hi; }   // (numerically lower = higher)
#endif
        // (default priority: with the highest one the pre-pass finished 180 us into the launch beside it, which then ran 65 us longer)
        hipError_t e = hipStreamCreateWithPriority(&a, hipStreamNonBlocking, prio);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&b, hipStreamNonBlocking);
        if (e != hipSuccess) {
            if (a) (void)hipStreamDestroy(a);
            return fail(TSDF_ERR_HIP, "side streams: %s", hipGetErrorString(e));
        }
        v->pre_stream = a;
        v->rb_stream = b;
    }
    const hipStream_t ps = pipelined ? v->pre_stream : v->stream;
    if (pipelined) {
        HIP_TRY(hipStreamWaitEvent(ps, inputs_ready, 0));
        if (v->list_used[P]) HIP_TRY(hipStreamWaitEvent(ps, v->list_free[P], 0));   // the Integrate kernel two launches back has read this parity's buffers
        if (v->rb_parity == P) HIP_TRY(hipStreamWaitEvent(ps, v->claims_done, 0)); // ... and its counters have left for the host
    } else if (classify && v->rb_parity == P) {
        HIP_TRY(hipStreamWaitEvent(v->stream, v->claims_done, 0));
    }
    if (count_claims) mi.common.claim_counter = claims_p + 1;   // bucket 0's claims word (the pre-pass adds the bucket's offset)
    if (classify) {
        // depth tile tables of the n frames (two small launches), then the kernels that consult them
        const size_t per_frame = tile_table_elems_host(mi.common.tiles_w, mi.common.tiles_h);
        float2 *tiles = nullptr;
        int rc = tables_begin(v, &tiles, ps);
        if (rc) return rc;
        // (fine_tables) the fine tables of the launch's frames sit behind its kMaxFramesPerLaunch coarse ones in the same slot
        float2 *fine = v->fine_tables ? tiles + (size_t)tsdfk::kMaxFramesPerLaunch * per_frame : nullptr;
        rc = build_tile_tables(ps, c, mi.common, depth_dev, masks_dev, n, tiles, claims_p, fine, pipelined);
        if (rc) return rc;
        mi.common.fine = fine;
        for (int f = 0; f < n; ++f) mi.frames[f].tiles = tiles + (size_t)f * per_frame;
        for (int f = n; f < tsdfk::kMaxFramesPerLaunch; ++f) mi.frames[f] = mi.frames[0];
    }
    const int nz_groups = (nz + mi.common.brick_s - 1) / mi.common.brick_s;   // a brick spans brick_s slices
    double claims_total = 0.0;
    bool launched = false, listed = false;
#ifdef TSDF_EXPERIMENTS
    {
        int rc = launch_multi_experiment(v, mi, depth_dev, masks_dev, c2b, n, label_ims != nullptr, any_mask, classify, &launched, &claims_total);
        if (rc) return rc;
    }
#endif
    if (!launched && classify) {
        // A pre-pass compacts the bricks some frame may touch into a work list and the launch runs one wavefront per entry.
        const int64_t per_group = (int64_t)mi.common.brick_groups * mi.common.bricks_per_group;
        tsdfk::BrickListParams bl;
        bl.nsx = (mi.common.bricks_per_group + tsdfk::kSuperBX - 1) / tsdfk::kSuperBX;
        bl.nsy = (mi.common.brick_groups + tsdfk::kSuperBY - 1) / tsdfk::kSuperBY;
        bl.nsz = (nz_groups + tsdfk::kSuperBZ - 1) / tsdfk::kSuperBZ;
        const int64_t n_super = (int64_t)bl.nsx * bl.nsy * bl.nsz;
        bl.wedge_mode = kWedgeMode;
#ifdef TSDF_EXPERIMENTS
        if (const char *e = std::getenv("TSDF_WEDGE_MODE")) bl.wedge_mode = std::atoi(e);     // A/B knob of the measurement build
#endif
        bl.super_h = tsdfk::kSuperBY * mi.common.brick_r;
        bl.super_d = tsdfk::kSuperBZ * mi.common.brick_s;
        bl.wedge_oy = (int)std::lround(std::fmax(-30000.0, std::fmin(30000.0, (double)c.origin[1] / c.voxel_size)));
        bl.wedge_oz = (int)std::lround(std::fmax(-30000.0, std::fmin(30000.0, (double)c.origin[2] / c.voxel_size + c.z_begin)));
        bl.fy_px = (int)std::lround(std::fmax(1.0, std::fmin(16000.0, std::fabs((double)c.cam_K[4]))));
        if (!(c.voxel_size > 0) || std::isnan(c.origin[1]) || std::isnan(c.origin[2])) bl.wedge_mode = 0;
        // a sub-list must hold every brick of every super-brick list_bucket deals to it: counted exactly, once per grid shape
        if (v->work_nsuper != n_super || v->work_wedge_mode != bl.wedge_mode) {
            std::vector<int64_t> per((size_t)tsdfk::kListBuckets, 0);
            for (int64_t id = 0; id < n_super; ++id) {      // the index order of classify_brick_list: slice groups fastest, then x, then y
                const int sz = (int)(id % bl.nsz), sx = (int)((id / bl.nsz) % bl.nsx), sy = (int)((id / bl.nsz) / bl.nsx);
                ++per[(size_t)tsdfk::list_bucket((unsigned int)id, sx, sy, sz, bl)];
            }
            v->work_bucket_supers = *std::max_element(per.begin(), per.end());
            v->work_nsuper = n_super;
            v->work_wedge_mode = bl.wedge_mode;
        }
        const int64_t cap = v->work_bucket_supers * tsdfk::kSuperBricks;
        const int64_t total = cap * tsdfk::kListBuckets;
        if (total > 0x7fffffffll - 1024) return fail(TSDF_ERR_INVALID, "fused launch: %lld bricks exceed the work list's 32-bit index", (long long)total);
        if (v->work_entries < (size_t)total) {
            if (v->d_work) {      // (a list in flight on either stream is done before its memory goes)
                if (v->pre_stream) HIP_TRY(hipStreamSynchronize(v->pre_stream));
                HIP_TRY(hipStreamSynchronize(v->stream));
                HIP_TRY(hipFree(v->d_work));
            }
            v->d_work = nullptr;
            v->work_entries = 0;
            HIP_TRY(hipMalloc((void **)&v->d_work, (v->pipeline_ok ? 2 : 1) * (size_t)total * sizeof(uint4)));     // one list per parity
            v->work_entries = (size_t)total;
        }
        uint4 *const work_p = v->d_work + (size_t)P * v->work_entries;
        bl.list = work_p;
        bl.counters = reinterpret_cast<unsigned char *>(claims_p);
        bl.bucket_cap = (unsigned int)cap;
        bl.poses = reinterpret_cast<tsdfk::ClassPoseTable *>(bl.counters + tsdfk::kCounterBytes);
        hipLaunchKernelGGL(tsdfk::classify_brick_list, dim3((unsigned)((n_super + 3) / 4)), block, 0, ps, mi, bl);
        if (pipelined) {       // the Integrate kernel waits for its pre-pass; everything before it on the handle's stream does not
            HIP_TRY(hipEventRecord(v->pre_done[P], ps));
            HIP_TRY(hipStreamWaitEvent(v->stream, v->pre_done[P], 0));
        }
        const dim3 grid_list((unsigned)(((cap + 3) / 4 + 1) * tsdfk::kListBuckets));   // front groups + back groups of every sub-list
        const uint4 *wl = work_p;
        const unsigned char *wc = bl.counters;
        const tsdfk::ClassPoseTable *wp = bl.poses;
        if (label_ims)
            hipLaunchKernelGGL((tsdfk::integrate_brick_list<kBrickNT, true, false>), grid_list, block, 0, v->stream, mi, wl, wc, bl.bucket_cap, wp);
        else if (any_mask)
            hipLaunchKernelGGL((tsdfk::integrate_brick_list<kBrickNT, false, true>), grid_list, block, 0, v->stream, mi, wl, wc, bl.bucket_cap, wp);
        else
            hipLaunchKernelGGL((tsdfk::integrate_brick_list<kBrickNT, false, false>), grid_list, block, 0, v->stream, mi, wl, wc, bl.bucket_cap, wp);
        claims_total = (double)per_group * nz_groups * n;   // wavefront-frames = bricks x frames
        launched = true;
        listed = true;
    }
    if (!launched) {
        // The per-voxel fused kernel.  Workgroup order: slices fastest -- consecutively dispatched workgroups share their
        // (x, y) footprint, i.e. the windows of the launch's up to 32 depth frames (39 MB, more than the L2s hold) they gather
        // from (512^3 S-surf with a depth frame per pose: 0.1325 -> 0.1148 ms per frame; 1024^3 fr3 trajectory 0.529 ->
        // 0.376) -- and rotated, (z + x + y) mod n, so that a slice group's workgroups are spread over all eight XCDs.
        dim3 grid_flat((v->chunks_per_slice + 3) / 4, 1, nz), grid_rows((mi.common.xgroups + 63) / 64, (c.dim_y + 3) / 4, nz);
        const dim3 &g0 = v->flat ? grid_flat : grid_rows;
        if (g0.x > 65535u) mi.z_fastest = 0;   // slices fastest puts the blocks of a slice into grid.z: 65535 at most (then: memory order)
        if (mi.z_fastest) {
            std::swap(grid_flat.x, grid_flat.z);
            std::swap(grid_rows.x, grid_rows.z);
        }
        if (label_ims && v->flat)
            hipLaunchKernelGGL((tsdfk::integrate_multi_inline<1, true, true, true, false>), grid_flat, block, 0, v->stream, mi);
        else if (label_ims)
            hipLaunchKernelGGL((tsdfk::integrate_multi_inline<1, true, false, true, false>), grid_rows, block, 0, v->stream, mi);
        else if (v->flat && any_mask)
            hipLaunchKernelGGL((tsdfk::integrate_multi_inline<1, true, true, false, true>), grid_flat, block, 0, v->stream, mi);
        else if (v->flat)
            hipLaunchKernelGGL((tsdfk::integrate_multi_inline<1, true, true, false, false>), grid_flat, block, 0, v->stream, mi);
        else if (any_mask)
            hipLaunchKernelGGL((tsdfk::integrate_multi_inline<1, true, false, false, true>), grid_rows, block, 0, v->stream, mi);
        else
            hipLaunchKernelGGL((tsdfk::integrate_multi_inline<1, true, false, false, false>), grid_rows, block, 0, v->stream, mi);
    }
    if (listed) {
        // after this parity's Integrate kernel: its buffers may be rebuilt (a pre-pass on the side stream waits for it)
        HIP_TRY(hipEventRecord(v->list_free[P], v->stream));
        v->list_used[P] = true;
        v->list_parity = v->pipeline_ok ? P ^ 1 : 0;
    }
    if (count_claims) {
        // the counters' way to the host: on a stream of its own when the launch is pipelined (behind list_free, i.e. after the
        // Integrate kernel), so that the copy sits neither between two Integrate kernels on the handle's stream nor in front of the
        // next pre-pass on the side stream; the pre-pass that clears this block again waits for it (rb_done)
        v->claims_total = claims_total;
        const hipStream_t rs = (pipelined && listed) ? v->rb_stream : v->stream;
        if (rs != v->stream) HIP_TRY(hipStreamWaitEvent(rs, v->list_free[P], 0));
        HIP_TRY(hipMemcpyAsync(v->h_claims, claims_p, tsdfk::kCounterBytes, hipMemcpyDeviceToHost, rs));
        HIP_TRY(hipEventRecord(v->claims_done, rs));
        v->claims_pending = true;
        v->rb_parity = (rs != v->stream) ? P : -1;
    }
    HIP_TRY(hipGetLastError());
    const int rc_end = tables_end(v);
    // (pipelined) the next launch's table slot is released after the OTHER of the handle's two events: a slot taken two launches
    // later then waits for this launch's Integrate kernel, not for the one in between
    if (pipelined) store_next_pass(v);
    return rc_end;
}

// Frames applied per pass over the slab.  Per frame the time is a + b / n: the weights are read and written,
// the summary word and the voxel coordinates set up, the launch filled and drained once per pass (b), and
// 512^3 measures 0.156 / 0.141 / 0.136 / 0.132 ms per frame at n = 4 / 8 / 16 / 32 (a = 0.129).
int frames_per_launch(const tsdf_volume *)
{
    return tsdfk::kMaxFramesPerLaunch;
}

bool can_fuse(const tsdf_volume *v)
{
    // (the measurement build's variants 4 .. 13 are flavours of the fused path)
    return (v->variant == 0 || v->variant == 7 || v->variant == 8 || (kExperiments && v->variant >= 4 && v->variant <= 13)) && v->cfg.dim_x % 4 == 0;
}

// A sequence of frames: fused frames_per_launch() at a time when the default kernel is selected.
int integrate_frames(tsdf_volume *v, const float *const *depth_dev, const uint8_t *const *masks_dev,
                     const float *cam2world, int n_frames)
{
    const bool fuse = can_fuse(v);
    int rc = TSDF_OK;
    // every frame of the sequence is readable once what precedes this call on the handle's stream has run: one event for all its
    // launches, so that their pre-passes may run beside the launches ahead of them (launch_multi, inputs_ready)
    bool seq_event = false;
    for (int k = 0; k < n_frames && rc == TSDF_OK;) {
        const int n = fuse ? std::min(frames_per_launch(v), n_frames - k) : 1;
        float c2b[16 * tsdfk::kMaxFramesPerLaunch];
        for (int i = 0; i < n; ++i) compose_cam2base(v, cam2world + 16 * (k + i), c2b + 16 * i);
        if (fuse && !seq_event && v->seq_ready && v->pipeline_ok && n_frames > n) {
            HIP_TRY(hipEventRecord(v->seq_ready, v->stream));
            seq_event = true;
        }
        if (fuse) rc = launch_multi(v, depth_dev + k, masks_dev ? masks_dev + k : nullptr, c2b, n, nullptr, nullptr, seq_event ? v->seq_ready : nullptr);
        else rc = launch_integrate(v, depth_dev[k], masks_dev ? masks_dev[k] : nullptr, c2b);
        k += n;
    }
    return rc;
}

// Apply the collected frames (poses already composed) as one sequence; their store slots go back after the launch.
int flush_pending(tsdf_volume *v)
{
    if (v->pend_count == 0 || v->in_flush) return TSDF_OK;
    v->in_flush = true;
    const int n = v->pend_count;
    v->pend_count = 0;
    int rc = TSDF_OK;
    hipError_t e = hipSetDevice(v->cfg.device);
    // the frames' copies ran in order on the copy stream: one event after the last of them covers all
    if (e == hipSuccess) e = hipEventRecord(v->pend_copied, v->copy_stream);
    if (e == hipSuccess) e = hipStreamWaitEvent(v->stream, v->pend_copied, 0);
    if (e == hipSuccess) {
        bool any_mask = false;
        for (int f = 0; f < n; ++f) any_mask = any_mask || v->pend_mask[f] != nullptr;
        if (can_fuse(v) && n > 1) {
            rc = launch_multi(v, v->pend_depth, any_mask ? v->pend_mask : nullptr, v->pend_c2b, n);
        } else {   // a single collected frame (or a variant that does not fuse): the one-frame kernels
            for (int f = 0; f < n && rc == TSDF_OK; ++f) rc = launch_integrate(v, v->pend_depth[f], v->pend_mask[f], v->pend_c2b + 16 * f);
        }
        if (rc == TSDF_OK) rc = store_release(v, &v->store->frames, v->pend_slot, n);
        if (rc == TSDF_OK && any_mask) rc = store_release(v, &v->store->masks, v->pend_mask_slot, n);
        store_next_pass(v);
    }
    v->in_flush = false;
    if (e != hipSuccess) return fail(TSDF_ERR_HIP, "deferred integration: %s", hipGetErrorString(e));
    return rc;
}

// Deferred integration, collecting side: a frame slot of the store for the next collected frame; `first_user` is the stream
// the copy into it will be queued on (the copy stream for host frames, the handle's stream for device-resident ones) ...
struct Collected {
    int slot, mask_slot;
    float *dev;
    const uint8_t *mask;
};

int pool_slot_begin(tsdf_volume *v, hipStream_t first_user, Collected *c)
{
    c->slot = c->mask_slot = -1;
    c->mask = nullptr;
    void *dev = nullptr;
    int rc = store_slot(v, &v->store->frames, first_user, &c->slot, &dev);   // (may apply the frames collected so far to make room)
    c->dev = static_cast<float *>(dev);
    return rc;
}

// ... and the frame's pose, composed now (or given as the relative pose itself); the batch is launched when it is full.
int pool_slot_commit(tsdf_volume *v, const Collected &c, const float cam2world[16], const float *cam2base = nullptr)
{
    const int slot = v->pend_count;
    v->pend_slot[slot] = c.slot;
    v->pend_mask_slot[slot] = c.mask_slot;
    v->pend_depth[slot] = c.dev;
    v->pend_mask[slot] = c.mask;
    if (cam2base) std::memcpy(v->pend_c2b + 16 * slot, cam2base, 16 * sizeof(float));
    else compose_cam2base(v, cam2world, v->pend_c2b + 16 * slot);
    std::memcpy(v->last_cam2base, v->pend_c2b + 16 * slot, sizeof v->last_cam2base);
    v->pend_count = slot + 1;
    if (v->pend_count >= std::min(v->defer_n, (int)tsdfk::kMaxFramesPerLaunch)) return flush_pending(v);
    return TSDF_OK;
}

// A device-resident frame (and its instance mask) collected like a host frame: copied device to device into a store slot on
// the handle's stream -- the same ordering the frame's kernel would have had -- so the caller's buffer is free for reuse under
// the stream's order, as before.
int collect_device_frame(tsdf_volume *v, const float *depth_dev, const uint8_t *mask_dev, const float *cam2world,
                         const float *cam2base)
{
    const size_t px = (size_t)v->cfg.im_height * v->cfg.im_width;
    Collected c;
    int rc = pool_slot_begin(v, v->stream, &c);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(c.dev, depth_dev, px * sizeof(float), hipMemcpyDeviceToDevice, v->stream));
    if (mask_dev) {
        void *m = nullptr;
        rc = store_slot(v, &v->store->masks, v->stream, &c.mask_slot, &m);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(m, mask_dev, px, hipMemcpyDeviceToDevice, v->stream));
        c.mask = static_cast<const uint8_t *>(m);
    }
    return pool_slot_commit(v, c, cam2world, cam2base);
}

// One-kernel-per-call staging of a host frame: pinned ring slot -> frame slot of the store (H2D on the copy stream, which the
// handle's stream then waits for).  `fill_host(pinned)` writes the pinned frame; *dev = the frame in HBM; *slot goes to
// stage_end() once the kernels that read it have been queued.
template <typename F>
int stage_begin(tsdf_volume *v, size_t bytes, F fill_host, int *slot, void **dev)
{
    int r = -1;
    hipError_t e = tsdf_store::ring_acquire(v->store, &r);
    if (e != hipSuccess) return fail(TSDF_ERR_HIP, "frame store (pinned ring): %s", hipGetErrorString(e));
    fill_host(v->store->ring[r].host);                     // the caller may free its buffer after we return
    int rc = store_slot(v, &v->store->frames, v->copy_stream, slot, dev);
    if (rc) { (void)tsdf_store::ring_release(v->store, r, v->copy_stream); return rc; }
    e = hipMemcpyAsync(*dev, v->store->ring[r].host, bytes, hipMemcpyHostToDevice, v->copy_stream);
    const hipError_t e2 = tsdf_store::ring_release(v->store, r, v->copy_stream);
    if (e == hipSuccess) e = e2;
    if (e != hipSuccess) return fail(TSDF_ERR_HIP, "staging copy: %s", hipGetErrorString(e));
    return TSDF_OK;
}

// the handle's stream waits for everything queued on the copy stream so far
int stream_waits_for_copies(tsdf_volume *v)
{
    HIP_TRY(hipEventRecord(v->pend_copied, v->copy_stream));
    HIP_TRY(hipStreamWaitEvent(v->stream, v->pend_copied, 0));
    return TSDF_OK;
}

int stage_end(tsdf_volume *v, const int *slots, int n)
{
    const int rc = store_release(v, &v->store->frames, slots, n);
    store_next_pass(v);
    return rc;
}

int fill(tsdf_volume *v)
{
    if (v->n_vox == 0) return TSDF_OK;
    size_t n = (size_t)v->n_vox;
    int blocks = (int)std::min<size_t>((n / 4 + 255) / 256, 256 * 8);
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(tsdfk::fill_grid, dim3(blocks), dim3(256), 0, v->stream, v->d_tsdf, v->d_weight, n);
    HIP_TRY(hipGetLastError());
    if (v->n_flags) {  // every TSDF value is 1: every segment flag is set
        HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)v->d_flags, 3, v->n_flags, v->stream));
        v->flags_known_zero = false;
    }
    return TSDF_OK;
}

int ensure_list(tsdf_volume *v, size_t bytes)
{
    if (v->list_bytes >= bytes) return TSDF_OK;
    if (v->d_list) HIP_TRY(hipFree(v->d_list));
    v->d_list = nullptr;
    v->list_bytes = 0;
    HIP_TRY(hipMalloc(&v->d_list, bytes));
    v->list_bytes = bytes;
    return TSDF_OK;
}

int ensure_scratch(tsdf_volume *v, size_t bytes)
{
    if (v->scratch_bytes >= bytes) return TSDF_OK;
    if (v->d_scratch) HIP_TRY(hipFree(v->d_scratch));
    v->d_scratch = nullptr;
    v->scratch_bytes = 0;
    HIP_TRY(hipMalloc(&v->d_scratch, bytes));
    v->scratch_bytes = bytes;
    return TSDF_OK;
}

// ---- file writers shared by the single-handle and the group entry points ---------------------------------------
// .ply of surface points: header text of ref: src/tsdf.cu:185-192 (the vertex count is printed with %d there)
int write_points_ply(const char *path, const float *xyz, int64_t n, const char *who)
{
    FILE *fp = std::fopen(path, "w");
    if (!fp) return fail(TSDF_ERR_IO, "%s: cannot open %s", who, path);
    std::fprintf(fp, "ply\nformat binary_little_endian 1.0\nelement vertex %d\n", (int)n);
    std::fprintf(fp, "property float x\nproperty float y\nproperty float z\nend_header\n");
    size_t wrote = std::fwrite(xyz, sizeof(float), (size_t)n * 3, fp);
    int bad = std::fclose(fp);
    if (wrote != (size_t)n * 3 || bad) return fail(TSDF_ERR_IO, "%s: short write to %s", who, path);
    return TSDF_OK;
}

// binary .ply with vertex + face elements, three vertices per triangle; rgb (may be null): 3 bytes per vertex
int write_mesh_ply(const char *path, const float *tri, int64_t n, const char *who, const unsigned char *rgb = nullptr)
{
    FILE *fp = std::fopen(path, "wb");
    if (!fp) return fail(TSDF_ERR_IO, "%s: cannot open %s", who, path);
    std::fprintf(fp, "ply\nformat binary_little_endian 1.0\nelement vertex %lld\n", (long long)(3 * n));
    std::fprintf(fp, "property float x\nproperty float y\nproperty float z\n");
    if (rgb) std::fprintf(fp, "property uchar red\nproperty uchar green\nproperty uchar blue\n");
    std::fprintf(fp, "element face %lld\nproperty list uchar int vertex_indices\nend_header\n", (long long)n);
    size_t ok = 0, want = 0;
    if (rgb) {
        std::vector<unsigned char> rec((size_t)(n > 0 ? n : 1) * 3 * 15);
        for (int64_t k = 0; k < 3 * n; ++k) {
            std::memcpy(rec.data() + 15 * k, tri + 3 * k, 12);
            std::memcpy(rec.data() + 15 * k + 12, rgb + 3 * k, 3);
        }
        ok = std::fwrite(rec.data(), 15, (size_t)n * 3, fp);
        want = (size_t)n * 3;
    } else {
        ok = std::fwrite(tri, sizeof(float), (size_t)n * 9, fp);
        want = (size_t)n * 9;
    }
    std::vector<unsigned char> faces((size_t)(n > 0 ? n : 1) * 13);
    for (int64_t f = 0; f < n; ++f) {
        unsigned char *rec = faces.data() + 13 * f;
        rec[0] = 3;
        for (int k = 0; k < 3; ++k) { const int32_t idx = (int32_t)(3 * f + k); std::memcpy(rec + 1 + 4 * k, &idx, 4); }
    }
    ok += std::fwrite(faces.data(), 13, (size_t)n, fp);
    int bad = std::fclose(fp);
    if (ok != want + (size_t)n || bad) return fail(TSDF_ERR_IO, "%s: short write to %s", who, path);
    return TSDF_OK;
}

// ref: src/tsdf.cu:119-129 -- dims as floats, origin, voxel size, truncation margin, then the TSDF values
int write_bin(const char *path, int dx, int dy, int dz, const float origin[3], float vs, float trunc, const float *data,
              int64_t n, const char *who)
{
    FILE *fp = std::fopen(path, "wb");
    if (!fp) return fail(TSDF_ERR_IO, "%s: cannot open %s", who, path);
    float hdr[8] = {(float)dx, (float)dy, (float)dz, origin[0], origin[1], origin[2], vs, trunc};
    size_t ok = std::fwrite(hdr, sizeof(float), 8, fp);
    ok += std::fwrite(data, sizeof(float), (size_t)n, fp);  // one write, not one per float
    int bad = std::fclose(fp);
    if (ok != 8 + (size_t)n || bad) return fail(TSDF_ERR_IO, "%s: short write to %s", who, path);
    return TSDF_OK;
}

// Device memory to an open file at the rate of the slower of PCIe and the file system: pieces of 32 MiB go device -> pinned host
// buffer on the handle's stream while the previous piece is being written (the reference's writers scan and write float by
// float, ref: src/tsdf.cu:130-131,210-212; a whole-array download into a fresh std::vector first costs a zero fill, a pageable
// copy and the write, one after the other: 300 ms for a 512^3 .bin against 150 ms this way).  The two buffers are process-wide
// (files are written rarely and the disk serialises writers anyway); the lock is held for the length of one array.
struct FileStager {
    std::mutex mu;
    void *pin[2] = {nullptr, nullptr};       // portable: any device may copy into them
    static constexpr size_t kPiece = (size_t)32 << 20;
    ~FileStager()
    {
        for (int i = 0; i < 2; ++i)
            if (pin[i]) (void)hipHostFree(pin[i]);
    }
};
FileStager &file_stager() { static FileStager s; return s; }

// bytes of device memory `src` (on v's device) appended to fp; the stream must already hold everything `src` depends on
int stream_device_to_file(tsdf_volume *v, FILE *fp, const void *src, size_t bytes, const char *who, const char *path)
{
    if (bytes == 0) return TSDF_OK;
    FileStager &fs = file_stager();
    std::lock_guard<std::mutex> lk(fs.mu);
    for (int i = 0; i < 2; ++i)
        if (!fs.pin[i]) HIP_TRY(hipHostMalloc(&fs.pin[i], FileStager::kPiece, hipHostMallocPortable));
    // the events belong to the device of v's stream (the current one: every caller has bound it), so they live for the call
    hipEvent_t ev[2] = {nullptr, nullptr};
    hipError_t e = hipEventCreateWithFlags(&ev[0], hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&ev[1], hipEventDisableTiming);
    const size_t pieces = (bytes + FileStager::kPiece - 1) / FileStager::kPiece;
    auto len = [&](size_t k) { return k + 1 < pieces ? FileStager::kPiece : bytes - k * FileStager::kPiece; };
    bool short_write = false;
    for (size_t k = 0; e == hipSuccess && k <= pieces; ++k) {
        if (k < pieces) {     // piece k on its way ...
            e = hipMemcpyAsync(fs.pin[k & 1], (const char *)src + k * FileStager::kPiece, len(k), hipMemcpyDeviceToHost, v->stream);
            if (e == hipSuccess) e = hipEventRecord(ev[k & 1], v->stream);
        }
        if (e == hipSuccess && k > 0) {          // ... while piece k - 1 is written
            e = hipEventSynchronize(ev[(k - 1) & 1]);
            if (e == hipSuccess && !short_write && std::fwrite(fs.pin[(k - 1) & 1], 1, len(k - 1), fp) != len(k - 1)) short_write = true;
        }
    }
    if (e != hipSuccess) (void)hipStreamSynchronize(v->stream);     // nothing may still be writing into the buffers
    for (int i = 0; i < 2; ++i)
        if (ev[i]) (void)hipEventDestroy(ev[i]);
    if (e != hipSuccess) return fail(TSDF_ERR_HIP, "%s: %s", who, hipGetErrorString(e));
    if (short_write) return fail(TSDF_ERR_IO, "%s: short write to %s", who, path);
    return TSDF_OK;
}

#ifdef TSDF_EXPERIMENTS
#include "tsdf_experiments_host.hip.h"
#endif

}  // namespace

extern "C" {

const char *tsdf_last_error(void) { return g_last_error.c_str(); }

const char *tsdf_version(void)
{
    return kExperiments ? "tsdf_hip 0.3 (gfx950) +experiments" : "tsdf_hip 0.3 (gfx950)";
}

void tsdf_multiply_matrix(const float a[16], const float b[16], float out[16])
{
    tsdf_host::multiply_matrix(a, b, out);
}

int tsdf_invert_matrix(const float m[16], float inv_out[16])
{
    return tsdf_host::invert_matrix(m, inv_out) ? 1 : 0;
}

int tsdf_config_default(tsdf_config *cfg, int32_t im_height, int32_t im_width)
{
    if (!cfg) return fail(TSDF_ERR_INVALID, "tsdf_config_default: cfg is NULL");
    std::memset(cfg, 0, sizeof *cfg);
    cfg->im_height = im_height;
    cfg->im_width = im_width;
    cfg->dim_x = cfg->dim_y = cfg->dim_z = 200;  // ref: include/tsdf.hpp:65-67
    cfg->z_begin = 0;
    cfg->z_end = 200;
    cfg->voxel_size = 0.004f;                    // ref: include/tsdf.hpp:63
    cfg->trunc_margin = cfg->voxel_size * 5;     // ref: include/tsdf.hpp:64
    cfg->max_depth = 6.0f;                       // ref: src/tsdf.cu:46
    const float K[9] = {535.4f, 0, 320.1f, 0, 539.2f, 247.6f, 0, 0, 1};  // ref: include/tsdf.hpp:96
    std::memcpy(cfg->cam_K, K, sizeof K);
    for (int i = 0; i < 4; ++i) cfg->base2world[5 * i] = 1.0f;
    return TSDF_OK;
}

int tsdf_create(const tsdf_config *cfg, tsdf_volume **out)
{
    if (!cfg || !out) return fail(TSDF_ERR_INVALID, "tsdf_create: NULL argument");
    *out = nullptr;
    if (cfg->dim_x <= 0 || cfg->dim_y <= 0 || cfg->dim_z <= 0)
        return fail(TSDF_ERR_INVALID, "tsdf_create: grid dims must be positive (%d,%d,%d)",
                    cfg->dim_x, cfg->dim_y, cfg->dim_z);
    if (cfg->z_begin < 0 || cfg->z_end < cfg->z_begin || cfg->z_end > cfg->dim_z)
        return fail(TSDF_ERR_INVALID, "tsdf_create: slab [%d,%d) outside grid z range [0,%d)",
                    cfg->z_begin, cfg->z_end, cfg->dim_z);
    if (cfg->im_height <= 0 || cfg->im_width <= 0 ||
        (int64_t)cfg->im_height * cfg->im_width > (int64_t)1 << 30)
        return fail(TSDF_ERR_INVALID, "tsdf_create: bad image size %dx%d", cfg->im_height, cfg->im_width);
    if (!(cfg->voxel_size > 0.0f) || !(cfg->trunc_margin > 0.0f))
        return fail(TSDF_ERR_INVALID, "tsdf_create: voxel_size and trunc_margin must be > 0");
    if (cfg->dim_y > 65535 * 4 || (cfg->z_end - cfg->z_begin) > 65535)
        return fail(TSDF_ERR_INVALID, "tsdf_create: slab exceeds launch limits (dim_y <= 262140, slices <= 65535)");
    if ((int64_t)cfg->dim_x * cfg->dim_y > ((int64_t)1 << 31) - 1024)
        return fail(TSDF_ERR_INVALID, "tsdf_create: a slice may hold at most 2^31 voxels (dim_x * dim_y)");

    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
        return fail(TSDF_ERR_NO_DEVICE, "tsdf_create: no HIP device visible (this library has no CPU path)");
    if (cfg->device < 0 || cfg->device >= n_dev)
        return fail(TSDF_ERR_INVALID, "tsdf_create: device %d not in [0,%d)", cfg->device, n_dev);

    tsdf_volume *v = new (std::nothrow) tsdf_volume();
    if (!v) return fail(TSDF_ERR_INVALID, "tsdf_create: out of host memory");
    std::memset(v, 0, sizeof *v);
    v->cfg = *cfg;
    v->n_vox = (int64_t)cfg->dim_x * cfg->dim_y * (cfg->z_end - cfg->z_begin);
    // ref: src/tsdf.cu:74 -- the inverse stays zero when base2world is singular, silently
    std::memset(v->base2world_inv, 0, sizeof v->base2world_inv);
    tsdf_host::invert_matrix(cfg->base2world, v->base2world_inv);

    int rc = TSDF_OK;
    auto cleanup = [&](int code) { tsdf_destroy(v); return code; };
    if (hipSetDevice(cfg->device) != hipSuccess)
        return cleanup(fail(TSDF_ERR_HIP, "tsdf_create: hipSetDevice(%d) failed", cfg->device));
    hipError_t e;
    if ((e = hipStreamCreateWithFlags(&v->own_stream, hipStreamNonBlocking)) != hipSuccess)
        return cleanup(fail(TSDF_ERR_HIP, "tsdf_create: hipStreamCreate: %s", hipGetErrorString(e)));
    v->stream = v->own_stream;
    v->defer_n = tsdfk::kMaxFramesPerLaunch;
    size_t bytes = (size_t)(v->n_vox > 0 ? v->n_vox : 1) * sizeof(float);
    if ((e = hipMalloc((void **)&v->d_tsdf, bytes)) != hipSuccess ||
        (e = hipMalloc((void **)&v->d_weight, bytes)) != hipSuccess)
        return cleanup(fail(TSDF_ERR_HIP, "tsdf_create: hipMalloc of %zu bytes x2: %s", bytes, hipGetErrorString(e)));
    v->flat = cfg->dim_x % 256 != 0;
    v->nseg = v->flat ? 0 : cfg->dim_x / 256;
    v->chunks_per_slice = (int)(((int64_t)cfg->dim_x * cfg->dim_y + 255) / 256);
    choose_brick(v);
    v->n_flags = cfg->dim_x % 4 == 0 ? (size_t)v->chunks_per_slice * (size_t)(cfg->z_end - cfg->z_begin) : 0;
    if ((e = hipMalloc((void **)&v->d_flags, (v->n_flags ? v->n_flags : 1) * sizeof(uint32_t))) != hipSuccess)
        return cleanup(fail(TSDF_ERR_HIP, "tsdf_create: hipMalloc of the summary: %s", hipGetErrorString(e)));
    // staging, deferral and tile-table memory: the store shared by every handle of this device and image size (nothing is
    // allocated until a frame arrives)
    v->tile = tile_edge_for(*cfg);
    v->fine_tables = fine_tables_for(*cfg, v->tile);
    v->pipeline_ok = (int64_t)cfg->dim_x * cfg->dim_y * (int64_t)(cfg->z_end - cfg->z_begin) < kFineLevelMinVoxels;
#ifdef TSDF_EXPERIMENTS
    if (const char *e = std::getenv("TSDF_PIPELINE")) v->pipeline_ok = std::atoi(e) != 0;     // A/B knob of the measurement build
#endif
    {
        const int tw = (cfg->im_width + v->tile - 1) / v->tile, th = (cfg->im_height + v->tile - 1) / v->tile;
        const int fw = (cfg->im_width + tsdfk::kFineTile - 1) / tsdfk::kFineTile, fh = (cfg->im_height + tsdfk::kFineTile - 1) / tsdfk::kFineTile;
        // a launch's tables: kMaxFramesPerLaunch coarse sparse tables, then (fine_tables) as many fine ones
        const size_t table_bytes = (int64_t)tw * th <= 16384
            ? tsdfk::kMaxFramesPerLaunch * (tile_table_elems_host(tw, th) + (v->fine_tables ? tsdfk::fine_table_elems(fw, fh) : 0)) * sizeof(float2) : 0;
        if ((e = tsdf_store::store_ref(cfg->device, (size_t)cfg->im_height * cfg->im_width, table_bytes, &v->store)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&v->flush_done[0], hipEventDisableTiming)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&v->flush_done[1], hipEventDisableTiming)) != hipSuccess ||
            (e = hipEventCreateWithFlags(&v->pend_copied, hipEventDisableTiming)) != hipSuccess)
            return cleanup(fail(TSDF_ERR_HIP, "tsdf_create: frame store: %s", hipGetErrorString(e)));
        v->table_slot = -1;
        v->copy_stream = v->store->copy_stream;   // the store's: shared by its handles (thread-safe; copies share one PCIe pipe anyway)
    }
    if ((rc = fill(v)) != TSDF_OK) return cleanup(rc);
    *out = v;
    return TSDF_OK;
}

int tsdf_destroy(tsdf_volume *v)
{
    if (!v) return TSDF_OK;
    (void)hipSetDevice(v->cfg.device);
    if (v->own_stream) (void)hipStreamSynchronize(v->own_stream);
    if (v->stream && v->stream != v->own_stream) (void)hipStreamSynchronize(v->stream);
    if (v->copy_stream) (void)hipStreamSynchronize(v->copy_stream);
    if (v->pre_stream) (void)hipStreamSynchronize(v->pre_stream);
    if (v->rb_stream) (void)hipStreamSynchronize(v->rb_stream);
    v->pend_count = 0;   // frames collected but never observed: nothing can tell whether they were applied
    if (v->store) {      // (the streams are idle: whatever this handle held, or others were waiting on its events for, is free)
        tsdf_store::slots_drop_owner(v->store, v);
        tsdf_store::store_unref(v->store);
        v->store = nullptr;
    }
    if (v->pend_copied) (void)hipEventDestroy(v->pend_copied);
    for (int i = 0; i < 2; ++i) if (v->flush_done[i]) (void)hipEventDestroy(v->flush_done[i]);
    for (int i = 0; i < kStageSlots; ++i) {
        if (v->h_frames[i]) (void)hipHostFree(v->h_frames[i]);
        if (v->d_frames[i]) (void)hipFree(v->d_frames[i]);
        if (v->frames_done[i]) (void)hipEventDestroy(v->frames_done[i]);
    }
    if (v->d_colour) (void)hipFree(v->d_colour);
    for (int i = 0; i < kStageSlots; ++i) {
        if (v->d_rgb[i]) (void)hipFree(v->d_rgb[i]);
        if (v->h_rgb[i]) (void)hipHostFree(v->h_rgb[i]);
        if (v->rgb_done[i]) (void)hipEventDestroy(v->rgb_done[i]);
    }
    if (v->d_label) (void)hipFree(v->d_label);
    if (v->d_fp) (void)hipFree(v->d_fp);
    if (v->d_bp) (void)hipFree(v->d_bp);
    if (v->d_scratch) (void)hipFree(v->d_scratch);
    if (v->d_list) (void)hipFree(v->d_list);
    if (v->d_wg_class) (void)hipFree(v->d_wg_class);
    if (v->d_flags) (void)hipFree(v->d_flags);
    if (v->d_super) (void)hipFree(v->d_super);
    if (v->d_work) (void)hipFree(v->d_work);
    if (v->d_claims) (void)hipFree(v->d_claims);
    if (v->h_claims) (void)hipHostFree(v->h_claims);
    if (v->claims_done) (void)hipEventDestroy(v->claims_done);
    for (int i = 0; i < 2; ++i) {
        if (v->pre_done[i]) (void)hipEventDestroy(v->pre_done[i]);
        if (v->list_free[i]) (void)hipEventDestroy(v->list_free[i]);
    }
    if (v->seq_ready) (void)hipEventDestroy(v->seq_ready);
    if (v->pre_stream) (void)hipStreamDestroy(v->pre_stream);
    if (v->rb_stream) (void)hipStreamDestroy(v->rb_stream);
    if (v->d_shortcut_stats) (void)hipFree(v->d_shortcut_stats);
    if (v->d_tsdf) (void)hipFree(v->d_tsdf);
    if (v->d_weight) (void)hipFree(v->d_weight);
    if (v->own_stream) (void)hipStreamDestroy(v->own_stream);
    delete v;
    return TSDF_OK;
}

int tsdf_reset(tsdf_volume *v)
{
    if (!v) return fail(TSDF_ERR_INVALID, "tsdf_reset: NULL handle");
    int rc = bind_device(v);
    if (rc) return rc;
    return fill(v);
}

int tsdf_integrate(tsdf_volume *v, const float *depth_host, const float cam2world[16])
{
    if (!v || !depth_host || !cam2world) return fail(TSDF_ERR_INVALID, "tsdf_integrate: NULL argument");
    const bool defer = v->defer_n > 1;
    int rc = bind_device(v, !defer);
    if (rc) return rc;
    const size_t img = (size_t)v->cfg.im_height * v->cfg.im_width * sizeof(float);
    // caller's frame -> pinned ring -> a frame slot in HBM (copy stream); the caller may free depth_host when we return
    int slot = -1;
    void *dev = nullptr;
    rc = stage_begin(v, img, [&](float *pinned) { tsdf_host::copy_to_pinned(pinned, depth_host, img); }, &slot, &dev);
    if (rc) return rc;
    if (defer) {
        // collect: pose composed now; launched defer_n at a time (or at the next call that observes the volume) as one fused
        // sequence
        Collected c;
        c.slot = slot; c.mask_slot = -1; c.dev = static_cast<float *>(dev); c.mask = nullptr;
        return pool_slot_commit(v, c, cam2world);
    }
    // one kernel per call: it waits for the copy (which overlapped the previous frame's kernel)
    rc = stream_waits_for_copies(v);
    if (rc) return rc;
    float c2b[16];
    compose_cam2base(v, cam2world, c2b);
    rc = launch_integrate(v, static_cast<const float *>(dev), nullptr, c2b);
    if (rc) return rc;
    return stage_end(v, &slot, 1);
}

int tsdf_set_deferral(tsdf_volume *v, int32_t n_frames)
{
    if (!v) return fail(TSDF_ERR_INVALID, "tsdf_set_deferral: NULL handle");
    if (n_frames < 0 || n_frames > tsdfk::kMaxFramesPerLaunch)
        return fail(TSDF_ERR_INVALID, "tsdf_set_deferral: n_frames must be in [0, %d]", tsdfk::kMaxFramesPerLaunch);
    int rc = bind_device(v);   // applies what has been collected under the old setting
    if (rc) return rc;
    v->defer_n = n_frames;
    return TSDF_OK;
}

int tsdf_convert_depth_u16(tsdf_volume *v, const uint16_t *raw_dev, float *depth_dev, float depth_factor,
                           int32_t row_step, int32_t col_step)
{
    if (!v || !raw_dev || !depth_dev) return fail(TSDF_ERR_INVALID, "tsdf_convert_depth_u16: NULL argument");
    if (!(depth_factor > 0.0f) || row_step < 1 || col_step < 1)
        return fail(TSDF_ERR_INVALID, "tsdf_convert_depth_u16: depth_factor must be > 0 and steps >= 1");
    int rc = bind_device(v);
    if (rc) return rc;
    const int n = v->cfg.im_height * v->cfg.im_width;
    const float scale = 1.0f / depth_factor;  // ref: examples/label_instance_rgbd.cpp:99-100 (fp32 reciprocal)
    hipLaunchKernelGGL(tsdfk::depth_u16_to_f32, dim3((n + 255) / 256), dim3(256), 0, v->stream, raw_dev, depth_dev,
                       v->cfg.im_height, v->cfg.im_width, scale, row_step, col_step);
    HIP_TRY(hipGetLastError());
    return TSDF_OK;
}

int tsdf_integrate_u16(tsdf_volume *v, const uint16_t *raw_host, float depth_factor, int32_t row_step,
                       int32_t col_step, const float cam2world[16])
{
    if (!v || !raw_host || !cam2world) return fail(TSDF_ERR_INVALID, "tsdf_integrate_u16: NULL argument");
    if (!(depth_factor > 0.0f) || row_step < 1 || col_step < 1)
        return fail(TSDF_ERR_INVALID, "tsdf_integrate_u16: depth_factor must be > 0 and steps >= 1");
    const bool defer = v->defer_n > 1;
    int rc = bind_device(v, !defer);
    if (rc) return rc;
    const size_t px = (size_t)v->cfg.im_height * v->cfg.im_width;
    // the raw frame (half the bytes of the float frame) -> pinned ring -> a frame slot; converted on the copy stream into a
    // second slot, which is the frame Integrate reads
    int raw_slot = -1, slot = -1;
    void *raw_dev = nullptr, *dev = nullptr;
    rc = stage_begin(v, px * sizeof(uint16_t), [&](float *pinned) { tsdf_host::copy_to_pinned(pinned, raw_host, px * sizeof(uint16_t)); }, &raw_slot, &raw_dev);
    if (rc) return rc;
    // a failure from here on hands the slots it holds back to the (shared, per-device) store: they would otherwise stay held until
    // the handle is destroyed and count against every other handle's share
    auto give_back = [&](int failed) {
        int held[2], n = 0;
        if (raw_slot >= 0) held[n++] = raw_slot;
        if (slot >= 0) held[n++] = slot;
        if (n > 0) {
            (void)hipEventRecord(v->pend_copied, v->copy_stream);      // whatever was queued on the copy stream has read them by then
            tsdf_store::slots_release_after(v->store, &v->store->frames, held, n, &v->pend_copied, v);
        }
        return failed;
    };
    rc = store_slot(v, &v->store->frames, v->copy_stream, &slot, &dev);
    if (rc) { slot = -1; return give_back(rc); }
    hipLaunchKernelGGL(tsdfk::depth_u16_to_f32, dim3((unsigned)((px + 255) / 256)), dim3(256), 0, v->copy_stream,
                       static_cast<const uint16_t *>(raw_dev), static_cast<float *>(dev), v->cfg.im_height, v->cfg.im_width,
                       1.0f / depth_factor, row_step, col_step);   // ref: examples/label_instance_rgbd.cpp:99-100 (fp32 reciprocal)
    {
        const hipError_t le = hipGetLastError();
        if (le != hipSuccess) return give_back(fail(TSDF_ERR_HIP, "depth_u16_to_f32: %s", hipGetErrorString(le)));
    }
    // the raw slot is free once the conversion has run: released after an event of the copy stream
    {
        const hipError_t ee = hipEventRecord(v->pend_copied, v->copy_stream);
        if (ee != hipSuccess) return give_back(fail(TSDF_ERR_HIP, "hipEventRecord: %s", hipGetErrorString(ee)));
    }
    tsdf_store::slots_release_after(v->store, &v->store->frames, &raw_slot, 1, &v->pend_copied, v);
    raw_slot = -1;
    if (defer) {
        Collected c;
        c.slot = slot; c.mask_slot = -1; c.dev = static_cast<float *>(dev); c.mask = nullptr;
        return pool_slot_commit(v, c, cam2world);
    }
    {
        const hipError_t we = hipStreamWaitEvent(v->stream, v->pend_copied, 0);
        if (we != hipSuccess) return give_back(fail(TSDF_ERR_HIP, "hipStreamWaitEvent: %s", hipGetErrorString(we)));
    }
    float c2b[16];
    compose_cam2base(v, cam2world, c2b);
    rc = launch_integrate(v, static_cast<const float *>(dev), nullptr, c2b);
    if (rc) return give_back(rc);
    return stage_end(v, &slot, 1);
}

int tsdf_integrate_device(tsdf_volume *v, const float *depth_dev, const float cam2world[16])
{
    if (!v || !depth_dev || !cam2world) return fail(TSDF_ERR_INVALID, "tsdf_integrate_device: NULL argument");
    const bool defer = v->defer_n > 1 && can_fuse(v);
    int rc = bind_device(v, !defer);
    if (rc) return rc;
    if (defer) return collect_device_frame(v, depth_dev, nullptr, cam2world, nullptr);
    float c2b[16];
    compose_cam2base(v, cam2world, c2b);
    return launch_integrate(v, depth_dev, nullptr, c2b);
}

int tsdf_integrate_cam2base(tsdf_volume *v, const float *depth_dev, const float cam2base[16])
{
    if (!v || !depth_dev || !cam2base) return fail(TSDF_ERR_INVALID, "tsdf_integrate_cam2base: NULL argument");
    const bool defer = v->defer_n > 1 && can_fuse(v);
    int rc = bind_device(v, !defer);
    if (rc) return rc;
    if (defer) return collect_device_frame(v, depth_dev, nullptr, nullptr, cam2base);
    return launch_integrate(v, depth_dev, nullptr, cam2base);
}

int tsdf_integrate_frames_device(tsdf_volume *v, const float *const *depth_dev, const uint8_t *const *masks_dev,
                                 const float *cam2world, int32_t n_frames)
{
    if (!v || !depth_dev || !cam2world || n_frames < 0)
        return fail(TSDF_ERR_INVALID, "tsdf_integrate_frames_device: bad argument");
    for (int k = 0; k < n_frames; ++k)
        if (!depth_dev[k]) return fail(TSDF_ERR_INVALID, "tsdf_integrate_frames_device: depth_dev[%d] is NULL", k);
    int rc = bind_device(v);
    if (rc) return rc;
    return integrate_frames(v, depth_dev, masks_dev, cam2world, n_frames);
}

int tsdf_integrate_frames_labels_device(tsdf_volume *v, const float *const *depth_dev, const uint16_t *const *label_im_dev,
                                        const float *const *score_im_dev, const float *cam2world, int32_t n_frames)
{
    if (!v || !depth_dev || !label_im_dev || !score_im_dev || !cam2world || n_frames < 0)
        return fail(TSDF_ERR_INVALID, "tsdf_integrate_frames_labels_device: bad argument");
    if (!v->d_label) return fail(TSDF_ERR_INVALID, "tsdf_integrate_frames_labels_device: call tsdf_labels_enable first");
    for (int k = 0; k < n_frames; ++k)
        if (!depth_dev[k] || !label_im_dev[k] || !score_im_dev[k])
            return fail(TSDF_ERR_INVALID, "tsdf_integrate_frames_labels_device: frame %d has a NULL image", k);
    int rc = bind_device(v);
    if (rc) return rc;
    bool seq_event = false;      // as integrate_frames: one "inputs readable" event for all launches of the sequence
    for (int k = 0; k < n_frames && rc == TSDF_OK;) {
        const int n = std::min(frames_per_launch(v), n_frames - k);
        float c2b[16 * tsdfk::kMaxFramesPerLaunch];
        for (int i = 0; i < n; ++i) compose_cam2base(v, cam2world + 16 * (k + i), c2b + 16 * i);
        if (!seq_event && v->seq_ready && v->pipeline_ok && n_frames > n) {
            HIP_TRY(hipEventRecord(v->seq_ready, v->stream));
            seq_event = true;
        }
        rc = launch_multi(v, depth_dev + k, nullptr, c2b, n, label_im_dev + k, score_im_dev + k, seq_event ? v->seq_ready : nullptr);
        k += n;
    }
    return rc;
}

int tsdf_integrate_masked_device(tsdf_volume *v, const float *depth_dev, const uint8_t *mask_dev,
                                 const float cam2world[16])
{
    if (!v || !depth_dev || !mask_dev || !cam2world)
        return fail(TSDF_ERR_INVALID, "tsdf_integrate_masked_device: NULL argument");
    const bool defer = v->defer_n > 1 && can_fuse(v);
    int rc = bind_device(v, !defer);
    if (rc) return rc;
    if (defer) return collect_device_frame(v, depth_dev, mask_dev, cam2world, nullptr);
    float c2b[16];
    compose_cam2base(v, cam2world, c2b);
    return launch_integrate(v, depth_dev, mask_dev, c2b);
}

int tsdf_sync(tsdf_volume *v)
{
    if (!v) return fail(TSDF_ERR_INVALID, "tsdf_sync: NULL handle");
    int rc = bind_device(v);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(v->stream));
    return TSDF_OK;
}

int tsdf_download(tsdf_volume *v, float *tsdf_host, float *weight_host)
{
    if (!v) return fail(TSDF_ERR_INVALID, "tsdf_download: NULL handle");
    int rc = bind_device(v);
    if (rc) return rc;
    size_t bytes = (size_t)v->n_vox * sizeof(float);
    HIP_TRY(hipStreamSynchronize(v->stream));
    if (bytes == 0) return TSDF_OK;
    if (tsdf_host) HIP_TRY(hipMemcpy(tsdf_host, v->d_tsdf, bytes, hipMemcpyDeviceToHost));
    if (weight_host) HIP_TRY(hipMemcpy(weight_host, v->d_weight, bytes, hipMemcpyDeviceToHost));
    return TSDF_OK;
}

int tsdf_copy_slices(tsdf_volume *v, int32_t z_local, int32_t n_slices, void *tsdf_dst, void *weight_dst)
{
    if (!v) return fail(TSDF_ERR_INVALID, "tsdf_copy_slices: NULL handle");
    const int nz = v->cfg.z_end - v->cfg.z_begin;
    if (z_local < 0 || n_slices < 0 || z_local + n_slices > nz)
        return fail(TSDF_ERR_INVALID, "tsdf_copy_slices: slices [%d,%d) outside the slab's %d", z_local, z_local + n_slices, nz);
    int rc = bind_device(v);
    if (rc) return rc;
    const size_t slice = (size_t)v->cfg.dim_x * v->cfg.dim_y;
    const size_t bytes = slice * (size_t)n_slices * sizeof(float);
    if (bytes == 0) return TSDF_OK;
    if (tsdf_dst) HIP_TRY(hipMemcpyAsync(tsdf_dst, v->d_tsdf + slice * z_local, bytes, hipMemcpyDefault, v->stream));
    if (weight_dst) HIP_TRY(hipMemcpyAsync(weight_dst, v->d_weight + slice * z_local, bytes, hipMemcpyDefault, v->stream));
    HIP_TRY(hipStreamSynchronize(v->stream));
    return TSDF_OK;
}

int tsdf_upload(tsdf_volume *v, const float *tsdf_host, const float *weight_host)
{
    if (!v) return fail(TSDF_ERR_INVALID, "tsdf_upload: NULL handle");
    int rc = bind_device(v);
    if (rc) return rc;
    size_t bytes = (size_t)v->n_vox * sizeof(float);
    HIP_TRY(hipStreamSynchronize(v->stream));
    if (bytes == 0) return TSDF_OK;
    if (tsdf_host) HIP_TRY(hipMemcpy(v->d_tsdf, tsdf_host, bytes, hipMemcpyHostToDevice));
    if (weight_host) HIP_TRY(hipMemcpy(v->d_weight, weight_host, bytes, hipMemcpyHostToDevice));
    return rebuild_summary(v);
}

int tsdf_refresh_summary(tsdf_volume *v)
{
    if (!v) return fail(TSDF_ERR_INVALID, "tsdf_refresh_summary: NULL handle");
    int rc = bind_device(v);
    if (rc) return rc;
    return rebuild_summary(v);
}

int tsdf_device_ptrs(tsdf_volume *v, float **tsdf_dev, float **weight_dev)
{
    if (!v) return fail(TSDF_ERR_INVALID, "tsdf_device_ptrs: NULL handle");
    int rc = bind_device(v);   // what the caller reads through the pointers must include every frame handed over so far
    if (rc) return rc;
    if (tsdf_dev) *tsdf_dev = v->d_tsdf;
    if (weight_dev) *weight_dev = v->d_weight;
    return TSDF_OK;
}

int64_t tsdf_slab_voxels(const tsdf_volume *v) { return v ? v->n_vox : 0; }

int tsdf_shortcut_stats(tsdf_volume *v, int32_t enable, uint64_t counts_out[3])
{
    if (!v) return fail(TSDF_ERR_INVALID, "tsdf_shortcut_stats: NULL handle");
    int rc = bind_device(v);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(v->stream));
    if (counts_out) {
        unsigned int h[3] = {0, 0, 0};
        if (v->d_shortcut_stats) HIP_TRY(hipMemcpy(h, v->d_shortcut_stats, sizeof h, hipMemcpyDeviceToHost));
        for (int i = 0; i < 3; ++i) counts_out[i] = h[i];
    }
    if (enable && !v->d_shortcut_stats) HIP_TRY(hipMalloc((void **)&v->d_shortcut_stats, 8 * sizeof(unsigned int)));
    if (v->d_shortcut_stats) {
        if (enable) HIP_TRY(hipMemset(v->d_shortcut_stats, 0, 8 * sizeof(unsigned int)));
        else { (void)hipFree(v->d_shortcut_stats); v->d_shortcut_stats = nullptr; }
    }
    return TSDF_OK;
}

int tsdf_brick_list_stats(tsdf_volume *v, uint64_t counts_out[4])
{
    if (!v || !counts_out) return fail(TSDF_ERR_INVALID, "tsdf_brick_list_stats: NULL argument");
    int rc = bind_device(v);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(v->stream));
    unsigned int h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if (v->d_shortcut_stats) HIP_TRY(hipMemcpy(h, v->d_shortcut_stats, sizeof h, hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; ++i) counts_out[i] = h[3 + i];
    return TSDF_OK;
}

int tsdf_classification_info(tsdf_volume *v, double info_out[2])
{
    if (!v || !info_out) return fail(TSDF_ERR_INVALID, "tsdf_classification_info: NULL argument");
    int rc = bind_device(v);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(v->stream));
    if (v->rb_stream) HIP_TRY(hipStreamSynchronize(v->rb_stream));     // (a pipelined launch's counters travel on their own stream)
    if (v->claims_pending && hipEventQuery(v->claims_done) == hipSuccess) {
        v->claim_fraction = claims_read_back(v);
        v->claims_pending = false;
        v->claims_known = true;
    }
    info_out[0] = v->claims_known ? v->claim_fraction : -1.0;
    info_out[1] = (double)v->launches_unclassified;
    return TSDF_OK;
}

int32_t tsdf_frames_per_launch(const tsdf_volume *v)
{
    if (!v) return 0;
    return can_fuse(v) ? frames_per_launch(v) : 1;
}

int tsdf_get_config(const tsdf_volume *v, tsdf_config *out)
{
    if (!v || !out) return fail(TSDF_ERR_INVALID, "tsdf_get_config: NULL argument");
    *out = v->cfg;
    return TSDF_OK;
}

int tsdf_last_cam2base(const tsdf_volume *v, float out[16])
{
    if (!v || !out) return fail(TSDF_ERR_INVALID, "tsdf_last_cam2base: NULL argument");
    std::memcpy(out, v->last_cam2base, sizeof v->last_cam2base);
    return TSDF_OK;
}

int tsdf_set_stream(tsdf_volume *v, void *hip_stream)
{
    if (!v) return fail(TSDF_ERR_INVALID, "tsdf_set_stream: NULL handle");
    int rc = bind_device(v);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(v->stream));  // do not reorder against work already queued
    v->stream = hip_stream ? (hipStream_t)hip_stream : v->own_stream;
    return TSDF_OK;
}

int tsdf_get_stream(tsdf_volume *v, void **hip_stream)
{
    if (!v || !hip_stream) return fail(TSDF_ERR_INVALID, "tsdf_get_stream: NULL argument");
    *hip_stream = (void *)v->stream;
    return TSDF_OK;
}

int tsdf_set_kernel_variant(tsdf_volume *v, int32_t variant)
{
    if (!v) return fail(TSDF_ERR_INVALID, "tsdf_set_kernel_variant: NULL handle");
    bool ok = shipped_variant(variant);
#ifdef TSDF_EXPERIMENTS
    ok = ok || experiment_variant(variant);
#endif
    if (!ok)
        return fail(TSDF_ERR_INVALID, "tsdf_set_kernel_variant: unknown variant %d (this build knows 0, 1, 3, 7, 8%s)", variant,
                    kExperiments ? " and the experiments" : "; the others live in the -DTSDF_EXPERIMENTS build");
    v->variant = variant;
    return TSDF_OK;
}

int tsdf_set_brick_shape(tsdf_volume *v, int32_t quads, int32_t rows, int32_t slices)
{
    if (!v) return fail(TSDF_ERR_INVALID, "tsdf_set_brick_shape: NULL handle");
    if (quads == 0 && rows == 0 && slices == 0) { choose_brick(v); return TSDF_OK; }
    if (!brick_shape_ok(v->cfg, quads, rows, slices))
        return fail(TSDF_ERR_INVALID, "tsdf_set_brick_shape: %d quads x %d rows x %d slices: needs quads * rows * slices <= 64 and "
                    "quads dividing dim_x / 4 = %d", quads, rows, slices, v->cfg.dim_x / 4);
    v->brick_q = quads; v->brick_r = rows; v->brick_s = slices;
    return TSDF_OK;
}

int tsdf_default_brick_shape(const tsdf_config *cfg, int32_t shape_out[3])
{
    if (!cfg || !shape_out) return fail(TSDF_ERR_INVALID, "tsdf_default_brick_shape: NULL argument");
    if (cfg->dim_x <= 0 || cfg->dim_y <= 0 || cfg->z_end < cfg->z_begin)
        return fail(TSDF_ERR_INVALID, "tsdf_default_brick_shape: bad grid");
    int q, r, s;
    choose_brick_for(*cfg, q, r, s);
    shape_out[0] = q; shape_out[1] = q ? r : 0; shape_out[2] = q ? s : 0;
    return TSDF_OK;
}

int tsdf_brick_shape(const tsdf_volume *v, int32_t shape_out[3])
{
    if (!v || !shape_out) return fail(TSDF_ERR_INVALID, "tsdf_brick_shape: NULL argument");
    shape_out[0] = v->brick_q; shape_out[1] = v->brick_q ? v->brick_r : 0; shape_out[2] = v->brick_q ? v->brick_s : 0;
    return TSDF_OK;
}

int tsdf_selftest_fastdiv(int32_t device, uint64_t seed, uint64_t n_samples, float fx, float cx,
                          uint64_t *mismatches, float first_bad[4])
{
    if (!mismatches || !first_bad) return fail(TSDF_ERR_INVALID, "tsdf_selftest_fastdiv: NULL argument");
    HIP_TRY(hipSetDevice(device));
    unsigned long long *d_cnt = nullptr;
    float *d_bad = nullptr;
    HIP_TRY(hipMalloc((void **)&d_cnt, sizeof *d_cnt));
    HIP_TRY(hipMalloc((void **)&d_bad, 4 * sizeof(float)));
    HIP_TRY(hipMemset(d_cnt, 0, sizeof *d_cnt));
    HIP_TRY(hipMemset(d_bad, 0, 4 * sizeof(float)));
    hipLaunchKernelGGL(tsdfk::selftest_fastdiv, dim3(256 * 8), dim3(256), 0, 0, seed, n_samples, fx, cx, d_cnt, d_bad);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    unsigned long long cnt = 0;
    if (e == hipSuccess) e = hipMemcpy(&cnt, d_cnt, sizeof cnt, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(first_bad, d_bad, 4 * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d_cnt);
    (void)hipFree(d_bad);
    if (e != hipSuccess) return fail(TSDF_ERR_HIP, "tsdf_selftest_fastdiv: %s", hipGetErrorString(e));
    *mismatches = cnt;
    return TSDF_OK;
}

int tsdf_selftest_fastdiv_band(int32_t device, uint64_t seed, uint64_t n_samples, uint64_t *mismatches, float first_bad[4])
{
    if (!mismatches || !first_bad) return fail(TSDF_ERR_INVALID, "tsdf_selftest_fastdiv_band: NULL argument");
    HIP_TRY(hipSetDevice(device));
    unsigned long long *d_cnt = nullptr;
    float *d_bad = nullptr;
    HIP_TRY(hipMalloc((void **)&d_cnt, sizeof *d_cnt));
    HIP_TRY(hipMalloc((void **)&d_bad, 4 * sizeof(float)));
    HIP_TRY(hipMemset(d_cnt, 0, sizeof *d_cnt));
    HIP_TRY(hipMemset(d_bad, 0, 4 * sizeof(float)));
    hipLaunchKernelGGL(tsdfk::selftest_fastdiv_band, dim3(256 * 8), dim3(256), 0, 0, seed, n_samples, d_cnt, d_bad);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    unsigned long long cnt = 0;
    if (e == hipSuccess) e = hipMemcpy(&cnt, d_cnt, sizeof cnt, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(first_bad, d_bad, 4 * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d_cnt);
    (void)hipFree(d_bad);
    if (e != hipSuccess) return fail(TSDF_ERR_HIP, "tsdf_selftest_fastdiv_band: %s", hipGetErrorString(e));
    *mismatches = cnt;
    return TSDF_OK;
}

int tsdf_object_origin(int32_t device, const float *depth_dev, const uint8_t *mask_dev, int32_t im_height,
                       int32_t im_width, const float cam_K[9], float origin_out[3])
{
    if (!depth_dev || !cam_K || !origin_out || im_height <= 0 || im_width <= 0)
        return fail(TSDF_ERR_INVALID, "tsdf_object_origin: bad argument");
    HIP_TRY(hipSetDevice(device));
    // the frame may have been produced on a handle's (non-blocking) stream, which the null stream does not order against
    HIP_TRY(hipDeviceSynchronize());
    float *d_out = nullptr;
    const float init[3] = {1000.0f, 1000.0f, 1000.0f};   // ref: src/Object.cpp:37
    HIP_TRY(hipMalloc((void **)&d_out, sizeof init));
    hipError_t e = hipMemcpy(d_out, init, sizeof init, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const int n = im_height * im_width;
        hipLaunchKernelGGL(tsdfk::object_origin, dim3(std::min((n + 255) / 256, 1024)), dim3(256), 0, 0, depth_dev, mask_dev,
                           im_height, im_width, cam_K[0], cam_K[4], cam_K[2], cam_K[5], d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(origin_out, d_out, sizeof init, hipMemcpyDeviceToHost);
    (void)hipFree(d_out);
    if (e != hipSuccess) return fail(TSDF_ERR_HIP, "tsdf_object_origin: %s", hipGetErrorString(e));
    return TSDF_OK;
}

int tsdf_selftest_round(int32_t device, uint64_t *mismatches, float first_bad[4])
{
    if (!mismatches || !first_bad) return fail(TSDF_ERR_INVALID, "tsdf_selftest_round: NULL argument");
    HIP_TRY(hipSetDevice(device));
    unsigned long long *d_cnt = nullptr;
    float *d_bad = nullptr;
    HIP_TRY(hipMalloc((void **)&d_cnt, sizeof *d_cnt));
    HIP_TRY(hipMalloc((void **)&d_bad, 4 * sizeof(float)));
    HIP_TRY(hipMemset(d_cnt, 0, sizeof *d_cnt));
    HIP_TRY(hipMemset(d_bad, 0, 4 * sizeof(float)));
    hipLaunchKernelGGL(tsdfk::selftest_round, dim3(256 * 8), dim3(256), 0, 0, d_cnt, d_bad);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    unsigned long long cnt = 0;
    if (e == hipSuccess) e = hipMemcpy(&cnt, d_cnt, sizeof cnt, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(first_bad, d_bad, 4 * sizeof(float), hipMemcpyDeviceToHost);
    (void)hipFree(d_cnt);
    (void)hipFree(d_bad);
    if (e != hipSuccess) return fail(TSDF_ERR_HIP, "tsdf_selftest_round: %s", hipGetErrorString(e));
    *mismatches = cnt;
    return TSDF_OK;
}

int tsdf_selftest_tile_tables(int32_t device, const float *depth_dev, const uint8_t *mask_dev, int32_t im_height,
                              int32_t im_width, float max_depth, uint64_t *mismatches)
{
    if (!depth_dev || !mismatches || im_height <= 0 || im_width <= 0)
        return fail(TSDF_ERR_INVALID, "tsdf_selftest_tile_tables: bad argument");
    HIP_TRY(hipSetDevice(device));
    uint64_t bad = 0;
    for (const int tile : {16, 8}) {      // both tile sizes the library uses (tile_edge_for)
        const int tw = (im_width + tile - 1) / tile, th = (im_height + tile - 1) / tile;
        if ((int64_t)tw * th > 16384) continue;
        const size_t per = tile_table_elems_host(tw, th);
        float2 *d_a = nullptr, *d_b = nullptr;
        HIP_TRY(hipMalloc((void **)&d_a, per * sizeof(float2)));
        if (hipMalloc((void **)&d_b, per * sizeof(float2)) != hipSuccess) { (void)hipFree(d_a); return fail(TSDF_ERR_HIP, "tsdf_selftest_tile_tables: hipMalloc"); }
        (void)hipMemset(d_a, 0xff, per * sizeof(float2));
        (void)hipMemset(d_b, 0x7f, per * sizeof(float2));
        tsdfk::TileSummaryParams tp;
        for (int f = 0; f < tsdfk::kMaxFramesPerLaunch; ++f) { tp.depth[f] = depth_dev; tp.mask[f] = mask_dev; }
        tp.H = im_height; tp.W = im_width; tp.tiles_w = tw; tp.tiles_h = th; tp.max_depth = max_depth;
        const unsigned lj = (unsigned)tile_levels_host(tw);
        // a: the kernels the library launches (strips of 64 pixels; doubling in LDS when the frame's tiles fit)
        tp.tiles = d_a;
        if (tile == 16)
            hipLaunchKernelGGL(tsdfk::depth_tile_summary<16>, dim3((unsigned)((((tw + 3) / 4) * th + 3) / 4), 1), dim3(64, 4), 0, 0, tp);
        else
            hipLaunchKernelGGL(tsdfk::depth_tile_summary<8>, dim3((unsigned)((((tw + 7) / 8) * th + 3) / 4), 1), dim3(64, 4), 0, 0, tp);
        if (tw * th <= tsdfk::kTileLdsEntries)
            hipLaunchKernelGGL(tsdfk::tile_sparse_table, dim3(lj, 1), dim3(tw * th > 2048 ? 1024 : 256), 0, 0, d_a, tw, th, (unsigned long long *)nullptr);
        else
            hipLaunchKernelGGL(tsdfk::tile_sparse_table_scan, dim3(lj, 1), dim3(256), 0, 0, d_a, tw, th);
        // b: one wavefront per tile, levels by scanning
        tp.tiles = d_b;
        if (tile == 16)
            hipLaunchKernelGGL(tsdfk::depth_tile_summary_per_tile<16>, dim3((unsigned)((tw * th + 3) / 4), 1), dim3(64, 4), 0, 0, tp);
        else
            hipLaunchKernelGGL(tsdfk::depth_tile_summary_per_tile<8>, dim3((unsigned)((tw * th + 3) / 4), 1), dim3(64, 4), 0, 0, tp);
        hipLaunchKernelGGL(tsdfk::tile_sparse_table_scan, dim3(lj, 1), dim3(256), 0, 0, d_b, tw, th);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();
        std::vector<float2> a(per), b(per);
        if (e == hipSuccess) e = hipMemcpy(a.data(), d_a, per * sizeof(float2), hipMemcpyDeviceToHost);
        if (e == hipSuccess) e = hipMemcpy(b.data(), d_b, per * sizeof(float2), hipMemcpyDeviceToHost);
        (void)hipFree(d_a);
        (void)hipFree(d_b);
        if (e != hipSuccess) return fail(TSDF_ERR_HIP, "tsdf_selftest_tile_tables: %s", hipGetErrorString(e));
        for (size_t i = 0; i < per; ++i) bad += std::memcmp(&a[i], &b[i], sizeof(float2)) != 0;
    }
    {   // the fine table (4-pixel tiles, nine levels): the two kernels the library launches against the pixel-by-pixel one
        const int fw = (im_width + tsdfk::kFineTile - 1) / tsdfk::kFineTile, fh = (im_height + tsdfk::kFineTile - 1) / tsdfk::kFineTile;
        const size_t per = tsdfk::fine_table_elems(fw, fh);
        float2 *d_a = nullptr, *d_b = nullptr;
        HIP_TRY(hipMalloc((void **)&d_a, per * sizeof(float2)));
        if (hipMalloc((void **)&d_b, per * sizeof(float2)) != hipSuccess) { (void)hipFree(d_a); return fail(TSDF_ERR_HIP, "tsdf_selftest_tile_tables: hipMalloc"); }
        (void)hipMemset(d_a, 0xff, per * sizeof(float2));
        (void)hipMemset(d_b, 0x7f, per * sizeof(float2));
        tsdfk::FineTileParams fp;
        for (int f = 0; f < tsdfk::kMaxFramesPerLaunch; ++f) { fp.depth[f] = depth_dev; fp.mask[f] = mask_dev; }
        fp.H = im_height; fp.W = im_width; fp.fw = fw; fp.fh = fh; fp.max_depth = max_depth;
        // b: pixel by pixel
        fp.fine = d_b;
        hipLaunchKernelGGL(tsdfk::fine_table_reference, dim3((unsigned)((per + 255) / 256), 1), dim3(256), 0, 0, fp);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();
        std::vector<float2> a(per), b(per);
        if (e == hipSuccess) e = hipMemcpy(b.data(), d_b, per * sizeof(float2), hipMemcpyDeviceToHost);
        // a: the standalone kernels (base tiles, then levels), then what a launch runs -- level (0, 0) out of the 8-pixel strip
        // kernel's pass, the upper levels by the extra workgroups of the sparse-table kernel
        for (int how = 0; how < 2 && e == hipSuccess; ++how) {
            (void)hipMemset(d_a, 0xff, per * sizeof(float2));
            if (how == 0) {
                fp.fine = d_a;
                hipLaunchKernelGGL(tsdfk::fine_tile_base, dim3((unsigned)((fw + 63) / 64), (unsigned)fh, 1), dim3(64), 0, 0, fp);
                hipLaunchKernelGGL(tsdfk::fine_tile_levels, dim3((unsigned)((fw * fh + 255) / 256), 1), dim3(256), 0, 0, d_a, fw, fh);
            } else {
                const int tw = (im_width + 7) / 8, th = (im_height + 7) / 8;
                if ((int64_t)tw * th > tsdfk::kTileLdsEntries) break;
                float2 *d_c = nullptr;
                if (hipMalloc((void **)&d_c, tile_table_elems_host(tw, th) * sizeof(float2)) != hipSuccess) { e = hipErrorOutOfMemory; break; }
                tsdfk::TileSummaryParams tp;
                for (int f = 0; f < tsdfk::kMaxFramesPerLaunch; ++f) { tp.depth[f] = depth_dev; tp.mask[f] = mask_dev; }
                tp.H = im_height; tp.W = im_width; tp.tiles_w = tw; tp.tiles_h = th; tp.max_depth = max_depth;
                tp.tiles = d_c; tp.fine = d_a; tp.fw = fw; tp.fh = fh;
                hipLaunchKernelGGL(tsdfk::depth_tile_summary<8>, dim3((unsigned)((((tw + 7) / 8) * th + 3) / 4), 1), dim3(64, 4), 0, 0, tp);
                const unsigned threads = tw * th > 2048 ? 1024 : 256;
                hipLaunchKernelGGL(tsdfk::tile_sparse_table, dim3((unsigned)tile_levels_host(tw) + ((unsigned)(fw * fh) + threads - 1) / threads, 1), dim3(threads), 0, 0,
                                   d_c, tw, th, (unsigned long long *)nullptr, d_a, fw, fh);
                e = hipGetLastError();
                if (e == hipSuccess) e = hipDeviceSynchronize();
                (void)hipFree(d_c);
            }
            if (e == hipSuccess) e = hipGetLastError();
            if (e == hipSuccess) e = hipDeviceSynchronize();
            if (e == hipSuccess) e = hipMemcpy(a.data(), d_a, per * sizeof(float2), hipMemcpyDeviceToHost);
            if (e == hipSuccess) for (size_t i = 0; i < per; ++i) bad += std::memcmp(&a[i], &b[i], sizeof(float2)) != 0;
        }
        (void)hipFree(d_a);
        (void)hipFree(d_b);
        if (e != hipSuccess) return fail(TSDF_ERR_HIP, "tsdf_selftest_tile_tables (fine): %s", hipGetErrorString(e));
    }
    *mismatches = bad;
    return TSDF_OK;
}

int tsdf_probe_stream(tsdf_volume *v, int32_t non_temporal, int32_t n_iters, float *elapsed_ms)
{
    if (!v || n_iters <= 0 || !elapsed_ms) return fail(TSDF_ERR_INVALID, "tsdf_probe_stream: bad argument");
    int rc = bind_device(v);
    if (rc) return rc;
    if (v->n_vox % 4 != 0 || v->n_vox == 0) return fail(TSDF_ERR_INVALID, "tsdf_probe_stream: slab voxels must be a positive multiple of 4");
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    const size_t nq = (size_t)v->n_vox / 4;
    const int blocks = (int)std::min<size_t>((nq + 255) / 256, (size_t)256 * 8);
    HIP_TRY(hipEventRecord(e0, v->stream));
    for (int i = 0; i < n_iters; ++i) {
        if (non_temporal) hipLaunchKernelGGL(tsdfk::stream_rmw<true>, dim3(blocks), dim3(256), 0, v->stream, v->d_tsdf, v->d_weight, nq, 1.0f, 0.0f);
        else hipLaunchKernelGGL(tsdfk::stream_rmw<false>, dim3(blocks), dim3(256), 0, v->stream, v->d_tsdf, v->d_weight, nq, 1.0f, 0.0f);
    }
    hipError_t er = hipEventRecord(e1, v->stream);
    hipError_t es = hipEventSynchronize(e1);
    float ms = 0.f;
    hipError_t et = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (er != hipSuccess || es != hipSuccess || et != hipSuccess)
        return fail(TSDF_ERR_HIP, "tsdf_probe_stream: event timing failed");
    *elapsed_ms = ms;
    return TSDF_OK;
}

static int frames_timed(tsdf_volume *v, const float *const *depth_dev, const uint8_t *const *masks_dev,
                        const float *cam2world, int32_t n_frames, float *elapsed_ms, const char *who)
{
    int rc = bind_device(v);
    if (rc) return rc;
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    HIP_TRY(hipEventRecord(e0, v->stream));
    rc = integrate_frames(v, depth_dev, masks_dev, cam2world, n_frames);
    hipError_t er = hipEventRecord(e1, v->stream);
    hipError_t es = hipEventSynchronize(e1);
    float ms = 0.f;
    hipError_t et = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc) return rc;
    if (er != hipSuccess || es != hipSuccess || et != hipSuccess)
        return fail(TSDF_ERR_HIP, "%s: event timing failed", who);
    *elapsed_ms = ms;
    return TSDF_OK;
}

int tsdf_integrate_sequence_timed(tsdf_volume *v, const float *depth_dev, const float *cam2world,
                                  int32_t n_frames, float *elapsed_ms)
{
    if (!v || !depth_dev || !cam2world || n_frames <= 0 || !elapsed_ms)
        return fail(TSDF_ERR_INVALID, "tsdf_integrate_sequence_timed: bad argument");
    std::vector<const float *> depths((size_t)n_frames, depth_dev);
    return frames_timed(v, depths.data(), nullptr, cam2world, n_frames, elapsed_ms, "tsdf_integrate_sequence_timed");
}

// Measurement aid ("hipGraph"): the same n one-frame launches queued call by call and replayed from a
// captured hipGraph, `iters` times each; device milliseconds per repetition by HIP events.  The volume ends up with
// 2 * iters * n more frames applied than before (both forms run).
int tsdf_probe_graph_replay(tsdf_volume *v, const float *depth_dev, const float *cam2world, int32_t n_frames, int32_t iters,
                            float *ms_launches, float *ms_graph)
{
    if (!v || !depth_dev || !cam2world || n_frames <= 0 || iters <= 0 || !ms_launches || !ms_graph)
        return fail(TSDF_ERR_INVALID, "tsdf_probe_graph_replay: bad argument");
    int rc = bind_device(v);
    if (rc) return rc;
    std::vector<float> c2b((size_t)n_frames * 16);
    for (int k = 0; k < n_frames; ++k) compose_cam2base(v, cam2world + 16 * k, c2b.data() + 16 * k);
    auto queue_all = [&]() -> int {
        for (int k = 0; k < n_frames; ++k) {
            int r = launch_integrate(v, depth_dev, nullptr, c2b.data() + 16 * k);
            if (r) return r;
        }
        return TSDF_OK;
    };
    hipEvent_t e0, e1;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    rc = queue_all();                                   // warm-up (and the summary's one-off work)
    if (rc) return rc;
    HIP_TRY(hipEventRecord(e0, v->stream));
    for (int i = 0; i < iters && rc == TSDF_OK; ++i) rc = queue_all();
    if (rc) return rc;
    HIP_TRY(hipEventRecord(e1, v->stream));
    HIP_TRY(hipEventSynchronize(e1));
    HIP_TRY(hipEventElapsedTime(ms_launches, e0, e1));
    *ms_launches /= (float)iters;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    HIP_TRY(hipStreamBeginCapture(v->stream, hipStreamCaptureModeThreadLocal));
    rc = queue_all();
    hipError_t ce = hipStreamEndCapture(v->stream, &graph);
    if (rc) return rc;
    if (ce != hipSuccess) return fail(TSDF_ERR_HIP, "tsdf_probe_graph_replay: capture failed: %s", hipGetErrorString(ce));
    HIP_TRY(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
    HIP_TRY(hipGraphLaunch(exec, v->stream));           // warm-up
    HIP_TRY(hipEventRecord(e0, v->stream));
    for (int i = 0; i < iters; ++i) HIP_TRY(hipGraphLaunch(exec, v->stream));
    HIP_TRY(hipEventRecord(e1, v->stream));
    HIP_TRY(hipEventSynchronize(e1));
    HIP_TRY(hipEventElapsedTime(ms_graph, e0, e1));
    *ms_graph /= (float)iters;
    (void)hipGraphExecDestroy(exec);
    (void)hipGraphDestroy(graph);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return TSDF_OK;
}

int tsdf_integrate_frames_timed(tsdf_volume *v, const float *const *depth_dev, const uint8_t *const *masks_dev,
                                const float *cam2world, int32_t n_frames, float *elapsed_ms)
{
    if (!v || !depth_dev || !cam2world || n_frames <= 0 || !elapsed_ms)
        return fail(TSDF_ERR_INVALID, "tsdf_integrate_frames_timed: bad argument");
    for (int k = 0; k < n_frames; ++k)
        if (!depth_dev[k]) return fail(TSDF_ERR_INVALID, "tsdf_integrate_frames_timed: depth_dev[%d] is NULL", k);
    return frames_timed(v, depth_dev, masks_dev, cam2world, n_frames, elapsed_ms, "tsdf_integrate_frames_timed");
}

// ---------------------------------------------------------------------------------------------
// per-voxel label fusion (csrc/tsdf_labels.hip.h)
// ---------------------------------------------------------------------------------------------
int tsdf_labels_enable(tsdf_volume *v, float prob_threshold)
{
    if (!v) return fail(TSDF_ERR_INVALID, "tsdf_labels_enable: NULL handle");
    if (v->cfg.dim_x % 4 != 0) return fail(TSDF_ERR_INVALID, "tsdf_labels_enable: dim_x must be a multiple of 4");
    int rc = bind_device(v);
    if (rc) return rc;
    v->prob_thd = prob_threshold;
    const size_t n = (size_t)(v->n_vox > 0 ? v->n_vox : 1);
    if (!v->d_label) {
        HIP_TRY(hipMalloc((void **)&v->d_label, n * sizeof(uint16_t)));
        HIP_TRY(hipMalloc((void **)&v->d_fp, n * sizeof(float)));
        HIP_TRY(hipMalloc((void **)&v->d_bp, n * sizeof(float)));
    }
    HIP_TRY(hipMemsetAsync(v->d_label, 0, n * sizeof(uint16_t), v->stream));
    HIP_TRY(hipMemsetAsync(v->d_fp, 0, n * sizeof(float), v->stream));
    HIP_TRY(hipMemsetAsync(v->d_bp, 0, n * sizeof(float), v->stream));
    return TSDF_OK;
}

int tsdf_integrate_labels_device(tsdf_volume *v, const float *depth_dev, const uint16_t *label_im_dev,
                                 const float *score_im_dev, const float cam2world[16])
{
    if (!v || !depth_dev || !label_im_dev || !score_im_dev || !cam2world)
        return fail(TSDF_ERR_INVALID, "tsdf_integrate_labels_device: NULL argument");
    if (!v->d_label) return fail(TSDF_ERR_INVALID, "tsdf_integrate_labels_device: call tsdf_labels_enable first");
    int rc = bind_device(v);
    if (rc) return rc;
    const int nz = v->cfg.z_end - v->cfg.z_begin;
    if (nz == 0) return TSDF_OK;
    float c2b[16];
    compose_cam2base(v, cam2world, c2b);
    tsdfk::LabelParams lp;
    lp.g = make_params(v, depth_dev, nullptr, c2b, 4);
    lp.label_im = label_im_dev; lp.score_im = score_im_dev;
    lp.label = v->d_label; lp.fp = v->d_fp; lp.bp = v->d_bp; lp.prob_thd = v->prob_thd;
    dim3 block(64, 4, 1), grid((lp.g.xgroups + 63) / 64, (v->cfg.dim_y + 3) / 4, nz);
    hipLaunchKernelGGL(tsdfk::integrate_labels, grid, block, 0, v->stream, lp);
    HIP_TRY(hipGetLastError());
    return TSDF_OK;
}

int tsdf_download_labels(tsdf_volume *v, uint16_t *label_host, float *fp_host, float *bp_host)
{
    if (!v) return fail(TSDF_ERR_INVALID, "tsdf_download_labels: NULL handle");
    if (!v->d_label) return fail(TSDF_ERR_INVALID, "tsdf_download_labels: labels not enabled");
    int rc = bind_device(v);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(v->stream));
    const size_t n = (size_t)v->n_vox;
    if (n == 0) return TSDF_OK;
    if (label_host) HIP_TRY(hipMemcpy(label_host, v->d_label, n * sizeof(uint16_t), hipMemcpyDeviceToHost));
    if (fp_host) HIP_TRY(hipMemcpy(fp_host, v->d_fp, n * sizeof(float), hipMemcpyDeviceToHost));
    if (bp_host) HIP_TRY(hipMemcpy(bp_host, v->d_bp, n * sizeof(float), hipMemcpyDeviceToHost));
    return TSDF_OK;
}

int tsdf_compose_labels(tsdf_volume *v, const uint8_t *masks_dev, const uint16_t *labels_host, const float *scores_host,
                        int32_t k, uint16_t *label_im_dev, float *score_im_dev)
{
    if (!v || !label_im_dev || !score_im_dev || k < 0 || (k > 0 && (!masks_dev || !labels_host || !scores_host)))
        return fail(TSDF_ERR_INVALID, "tsdf_compose_labels: bad argument");
    if (k > tsdfk::kMaxInstances) return fail(TSDF_ERR_INVALID, "tsdf_compose_labels: at most %d instances per frame", tsdfk::kMaxInstances);
    int rc = bind_device(v);
    if (rc) return rc;
    tsdfk::ComposeParams c;
    c.masks = masks_dev; c.label_im = label_im_dev; c.score_im = score_im_dev; c.k = k;
    c.n_pixels = v->cfg.im_height * v->cfg.im_width;
    for (int i = 0; i < tsdfk::kMaxInstances; ++i) { c.labels[i] = i < k ? labels_host[i] : 0; c.scores[i] = i < k ? scores_host[i] : 0.0f; }
    hipLaunchKernelGGL(tsdfk::compose_labels, dim3((c.n_pixels + 255) / 256), dim3(256), 0, v->stream, c);
    HIP_TRY(hipGetLastError());
    return TSDF_OK;
}

// ---------------------------------------------------------------------------------------------
// per-voxel colour fusion (csrc/tsdf_colour.hip.h)
// ---------------------------------------------------------------------------------------------
int tsdf_colour_enable(tsdf_volume *v)
{
    if (!v) return fail(TSDF_ERR_INVALID, "tsdf_colour_enable: NULL handle");
    if (v->cfg.dim_x % 4 != 0) return fail(TSDF_ERR_INVALID, "tsdf_colour_enable: dim_x must be a multiple of 4");
    int rc = bind_device(v);
    if (rc) return rc;
    const size_t n = (size_t)(v->n_vox > 0 ? v->n_vox : 1);
    if (!v->d_colour) HIP_TRY(hipMalloc((void **)&v->d_colour, n * sizeof(uint32_t)));
    HIP_TRY(hipMemsetAsync(v->d_colour, 0, n * sizeof(uint32_t), v->stream));
    return TSDF_OK;
}

static int launch_colour(tsdf_volume *v, const float *depth_dev, const uint8_t *rgb_dev, const float *c2b)
{
    const int nz = v->cfg.z_end - v->cfg.z_begin;
    if (nz == 0) return TSDF_OK;
    tsdfk::ColourParams cp;
    cp.g = make_params(v, depth_dev, nullptr, c2b, 4);
    cp.rgb = rgb_dev;
    cp.colour = v->d_colour;
    dim3 block(64, 4, 1), grid((cp.g.xgroups + 63) / 64, (v->cfg.dim_y + 3) / 4, nz);
    hipLaunchKernelGGL(tsdfk::integrate_colour, grid, block, 0, v->stream, cp);
    HIP_TRY(hipGetLastError());
    return TSDF_OK;
}

int tsdf_integrate_colour_device(tsdf_volume *v, const float *depth_dev, const uint8_t *rgb_dev, const float cam2world[16])
{
    if (!v || !depth_dev || !rgb_dev || !cam2world) return fail(TSDF_ERR_INVALID, "tsdf_integrate_colour_device: NULL argument");
    if (!v->d_colour) return fail(TSDF_ERR_INVALID, "tsdf_integrate_colour_device: call tsdf_colour_enable first");
    int rc = bind_device(v);
    if (rc) return rc;
    float c2b[16];
    compose_cam2base(v, cam2world, c2b);
    return launch_colour(v, depth_dev, rgb_dev, c2b);
}

int tsdf_integrate_rgbd(tsdf_volume *v, const float *depth_host, const uint8_t *rgb_host, const float cam2world[16])
{
    if (!v || !depth_host || !rgb_host || !cam2world) return fail(TSDF_ERR_INVALID, "tsdf_integrate_rgbd: NULL argument");
    if (!v->d_colour) return fail(TSDF_ERR_INVALID, "tsdf_integrate_rgbd: call tsdf_colour_enable first");
    int rc = bind_device(v);
    if (rc) return rc;
    const size_t px = (size_t)v->cfg.im_height * v->cfg.im_width, img = px * sizeof(float);
    // the colour image has a small ring of its own (pinned + HBM, allocated on first use; one event per slot: the colour
    // kernel that read it), the depth frame goes through the shared store like any other host frame
    const int s = v->rgb_next;
    v->rgb_next = (s + 1) % kStageSlots;
    if (!v->d_rgb[s]) {
        HIP_TRY(hipMalloc((void **)&v->d_rgb[s], px * 3));
        HIP_TRY(hipHostMalloc((void **)&v->h_rgb[s], px * 3, hipHostMallocDefault));
        HIP_TRY(hipEventCreateWithFlags(&v->rgb_done[s], hipEventDisableTiming));
    }
    if (v->rgb_used[s]) HIP_TRY(hipEventSynchronize(v->rgb_done[s]));
    std::memcpy(v->h_rgb[s], rgb_host, px * 3);           // the caller may free both images after we return
    int slot = -1;
    void *dev = nullptr;
    rc = stage_begin(v, img, [&](float *pinned) { tsdf_host::copy_to_pinned(pinned, depth_host, img); }, &slot, &dev);
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(v->d_rgb[s], v->h_rgb[s], px * 3, hipMemcpyHostToDevice, v->copy_stream));
    rc = stream_waits_for_copies(v);
    if (rc) return rc;
    float c2b[16];
    compose_cam2base(v, cam2world, c2b);
    rc = launch_integrate(v, static_cast<const float *>(dev), nullptr, c2b);
    if (rc == TSDF_OK) rc = launch_colour(v, static_cast<const float *>(dev), v->d_rgb[s], c2b);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(v->rgb_done[s], v->stream));
    v->rgb_used[s] = true;
    return stage_end(v, &slot, 1);
}

int tsdf_download_colour(tsdf_volume *v, uint32_t *colour_host)
{
    if (!v || !colour_host) return fail(TSDF_ERR_INVALID, "tsdf_download_colour: NULL argument");
    if (!v->d_colour) return fail(TSDF_ERR_INVALID, "tsdf_download_colour: colour not enabled");
    int rc = bind_device(v);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(v->stream));
    if (v->n_vox > 0) HIP_TRY(hipMemcpy(colour_host, v->d_colour, (size_t)v->n_vox * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return TSDF_OK;
}

// ---------------------------------------------------------------------------------------------
// batched per-object volumes
// ---------------------------------------------------------------------------------------------
namespace {


// Apply the frames the batch has collected: one fused launch per member, in member order, on the batch's stream.
int batch_flush(tsdf_batch *b)
{
    if (b->pend_count == 0 || b->in_flush) return TSDF_OK;
    b->in_flush = true;
    const int n = b->pend_count, members = (int)b->vols.size();
    b->pend_count = 0;
    const size_t px = (size_t)b->vols[0]->cfg.im_height * b->vols[0]->cfg.im_width;
    const float *ptrs[tsdfk::kMaxFramesPerLaunch];
    for (int f = 0; f < n; ++f) ptrs[f] = b->d_depth_pool + (size_t)f * px;
    int rc = TSDF_OK;
    if (hipSetDevice(b->device) != hipSuccess) rc = fail(TSDF_ERR_HIP, "tsdf_batch: hipSetDevice failed");
    // fork: the side streams start when everything queued on the batch's stream so far (the frames' copies) is done
    // (measured, 200^3 members with instance masks, ms per frame, one stream -> four: 16 members 0.081 -> 0.062, 8 members
    // 0.047 -> 0.045, 4 members 0.027 -> 0.034, 2 members 0.016 -> 0.023: few members fill the GPU one after the other)
    const int lanes = members < 8 ? 1 : kBatchSideStreams;
    hipError_t e = hipSuccess;
    if (rc == TSDF_OK && lanes > 1) {
        if (!b->collected) {
            e = hipEventCreateWithFlags(&b->collected, hipEventDisableTiming);
            for (int k = 0; k < kBatchSideStreams && e == hipSuccess; ++k) {
                e = hipStreamCreateWithFlags(&b->side[k], hipStreamNonBlocking);
                if (e == hipSuccess) e = hipEventCreateWithFlags(&b->side_done[k], hipEventDisableTiming);
            }
        }
        if (e == hipSuccess) e = hipEventRecord(b->collected, b->stream);
        for (int k = 0; k < lanes && e == hipSuccess; ++k) e = hipStreamWaitEvent(b->side[k], b->collected, 0);
        if (e != hipSuccess) rc = fail(TSDF_ERR_HIP, "tsdf_batch: side streams: %s", hipGetErrorString(e));
    }
    for (int i = 0; i < members && rc == TSDF_OK; ++i) {
        tsdf_volume *v = b->vols[i];
        const uint8_t *const *masks = b->pend_mask.data() + (size_t)i * tsdfk::kMaxFramesPerLaunch;
        const float *c2b = b->pend_c2b.data() + (size_t)i * tsdfk::kMaxFramesPerLaunch * 16;
        bool any_mask = false;
        for (int f = 0; f < n; ++f) any_mask = any_mask || masks[f] != nullptr;
        if (lanes > 1) v->stream = b->side[i % lanes];
        if (can_fuse(v) && n > 1) {
            rc = launch_multi(v, ptrs, any_mask ? masks : nullptr, c2b, n);
        } else {
            for (int f = 0; f < n && rc == TSDF_OK; ++f) rc = launch_integrate(v, ptrs[f], masks[f], c2b + 16 * f);
        }
        v->stream = b->stream;
    }
    // join: whatever follows on the batch's stream (the next frames' copies into the pool, a download) comes after them
    for (int k = 0; k < lanes && lanes > 1; ++k) {
        if (hipEventRecord(b->side_done[k], b->side[k]) != hipSuccess || hipStreamWaitEvent(b->stream, b->side_done[k], 0) != hipSuccess) {
            (void)hipDeviceSynchronize();   // never leave the streams unordered
            if (rc == TSDF_OK) rc = fail(TSDF_ERR_HIP, "tsdf_batch: joining the side streams failed");
        }
    }
    b->in_flush = false;
    return rc;
}

// Collect one frame for every member: the depth image once, the masks by one gather launch, the poses composed now.
// Everything is queued on the batch's stream, so the caller's buffers are free for reuse under that stream's order and the
// pool is not overwritten before the launches that read it have run.
int batch_collect(tsdf_batch *b, const float *depth_dev, const uint8_t *const *masks_dev, const float cam2world[16])
{
    const int members = (int)b->vols.size(), slot = b->pend_count;
    const size_t px = (size_t)b->vols[0]->cfg.im_height * b->vols[0]->cfg.im_width;
    if (!b->d_depth_pool) {
        HIP_TRY(hipMalloc((void **)&b->d_depth_pool, (size_t)tsdfk::kMaxFramesPerLaunch * px * sizeof(float)));
        b->pend_c2b.assign((size_t)members * tsdfk::kMaxFramesPerLaunch * 16, 0.0f);
        b->pend_mask.assign((size_t)members * tsdfk::kMaxFramesPerLaunch, nullptr);
    }
    bool any_mask = false;
    for (int i = 0; i < members && masks_dev; ++i) any_mask = any_mask || masks_dev[i] != nullptr;
    if (any_mask && !b->d_mask_pool)
        HIP_TRY(hipMalloc((void **)&b->d_mask_pool, (size_t)tsdfk::kMaxFramesPerLaunch * members * px));
    HIP_TRY(hipMemcpyAsync(b->d_depth_pool + (size_t)slot * px, depth_dev, px * sizeof(float), hipMemcpyDeviceToDevice, b->stream));
    for (int i0 = 0; i0 < members && any_mask; i0 += tsdfk::kGatherMasks) {
        tsdfk::MaskGatherParams gp;
        const int m = std::min(tsdfk::kGatherMasks, members - i0);
        bool some = false;
        for (int k = 0; k < tsdfk::kGatherMasks; ++k) {
            gp.src[k] = k < m ? masks_dev[i0 + k] : nullptr;
            gp.dst[k] = k < m ? b->d_mask_pool + ((size_t)slot * members + i0 + k) * px : nullptr;
            some = some || gp.src[k] != nullptr;
        }
        gp.bytes = px;
        if (some) hipLaunchKernelGGL(tsdfk::gather_masks, dim3((unsigned)((px + 4095) / 4096), (unsigned)m), dim3(256), 0, b->stream, gp);
    }
    HIP_TRY(hipGetLastError());
    for (int i = 0; i < members; ++i) {
        tsdf_volume *v = b->vols[i];
        float *c2b = b->pend_c2b.data() + ((size_t)i * tsdfk::kMaxFramesPerLaunch + slot) * 16;
        compose_cam2base(v, cam2world, c2b);   // each object has its own base frame (ref: src/Object.cpp:23-29)
        std::memcpy(v->last_cam2base, c2b, 16 * sizeof(float));
        b->pend_mask[(size_t)i * tsdfk::kMaxFramesPerLaunch + slot] =
            (masks_dev && masks_dev[i]) ? b->d_mask_pool + ((size_t)slot * members + i) * px : nullptr;
    }
    b->pend_count = slot + 1;
    if (b->pend_count >= std::min(b->vols[0]->defer_n, (int)tsdfk::kMaxFramesPerLaunch)) return batch_flush(b);
    return TSDF_OK;
}

}  // namespace

int tsdf_batch_destroy(tsdf_batch *b)
{
    if (!b) return TSDF_OK;
    (void)hipSetDevice(b->device);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    for (tsdf_volume *v : b->vols) {
        if (v) { v->stream = v->own_stream; v->owner = nullptr; tsdf_destroy(v); }
    }
    for (int i = 0; i < kStageSlots; ++i) {
        if (b->h_params[i]) (void)hipHostFree(b->h_params[i]);
        if (b->d_params[i]) (void)hipFree(b->d_params[i]);
        if (b->h_poses[i]) (void)hipHostFree(b->h_poses[i]);
        if (b->d_poses[i]) (void)hipFree(b->d_poses[i]);
        if (b->slot_done[i]) (void)hipEventDestroy(b->slot_done[i]);
    }
    if (b->d_slice_map) (void)hipFree(b->d_slice_map);
    if (b->d_group_map) (void)hipFree(b->d_group_map);
    if (b->d_tiles) (void)hipFree(b->d_tiles);
    if (b->d_wg_class) (void)hipFree(b->d_wg_class);
    if (b->d_brick_class) (void)hipFree(b->d_brick_class);
    for (int i = 0; i < kBatchSideStreams; ++i) {
        if (b->side[i]) { (void)hipStreamSynchronize(b->side[i]); (void)hipStreamDestroy(b->side[i]); }
        if (b->side_done[i]) (void)hipEventDestroy(b->side_done[i]);
    }
    if (b->collected) (void)hipEventDestroy(b->collected);
    if (b->d_depth_pool) (void)hipFree(b->d_depth_pool);
    if (b->d_mask_pool) (void)hipFree(b->d_mask_pool);
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
    return TSDF_OK;
}

int tsdf_batch_create(const tsdf_config *cfgs, int32_t n, tsdf_batch **out)
{
    if (!cfgs || !out || n <= 0) return fail(TSDF_ERR_INVALID, "tsdf_batch_create: bad argument");
    *out = nullptr;
    for (int i = 0; i < n; ++i) {
        if (cfgs[i].device != cfgs[0].device || cfgs[i].im_height != cfgs[0].im_height ||
            cfgs[i].im_width != cfgs[0].im_width)
            return fail(TSDF_ERR_INVALID, "tsdf_batch_create: volume %d differs in device or image size", i);
        if (cfgs[i].dim_x % 4 != 0)
            return fail(TSDF_ERR_INVALID, "tsdf_batch_create: volume %d: dim_x must be a multiple of 4", i);
    }
    tsdf_batch *b = new (std::nothrow) tsdf_batch();
    if (!b) return fail(TSDF_ERR_INVALID, "tsdf_batch_create: out of host memory");
    b->device = cfgs[0].device;
    b->stream = nullptr; b->d_slice_map = nullptr; b->d_group_map = nullptr; b->total_groups = 0; b->slot_next = 0;
    b->d_depth_pool = nullptr; b->d_mask_pool = nullptr; b->pend_count = 0; b->in_flush = false;
    for (int i = 0; i < kBatchSideStreams; ++i) { b->side[i] = nullptr; b->side_done[i] = nullptr; }
    b->collected = nullptr;
    b->d_tiles = nullptr; b->tiles_per_object = 0; b->d_wg_class = nullptr; b->d_brick_class = nullptr; b->brick_class_bytes = 0;
    b->total_slices = b->max_blocks = 0;
    for (int i = 0; i < kStageSlots; ++i) {
        b->h_params[i] = nullptr; b->d_params[i] = nullptr; b->h_poses[i] = nullptr; b->d_poses[i] = nullptr;
        b->slot_done[i] = nullptr; b->slot_used[i] = false;
    }
    auto cleanup = [&](int code) { tsdf_batch_destroy(b); return code; };
    std::vector<int2> map;
    for (int i = 0; i < n; ++i) {
        tsdf_volume *v = nullptr;
        g_create_for_batch = true;        // one table layout for all members (tile_edge_for)
        int rc = tsdf_create(&cfgs[i], &v);
        g_create_for_batch = false;
        if (rc) return cleanup(rc);
        b->vols.push_back(v);
        const int nz = cfgs[i].z_end - cfgs[i].z_begin;
        for (int z = 0; z < nz; ++z) map.push_back(make_int2(i, z));
        b->max_blocks = std::max(b->max_blocks, (v->chunks_per_slice + 3) / 4);
    }
    b->total_slices = (int)map.size();
    if (b->total_slices > 65535) return cleanup(fail(TSDF_ERR_INVALID, "tsdf_batch_create: %d slices in total exceed the launch limit 65535", b->total_slices));
    hipError_t e = hipSetDevice(b->device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking);
    if (e == hipSuccess && !map.empty()) e = hipMalloc((void **)&b->d_slice_map, map.size() * sizeof(int2));
    if (e == hipSuccess && !map.empty()) e = hipMemcpy(b->d_slice_map, map.data(), map.size() * sizeof(int2), hipMemcpyHostToDevice);
    for (int i = 0; i < kStageSlots && e == hipSuccess; ++i) {
        e = hipHostMalloc((void **)&b->h_params[i], n * sizeof(tsdfk::IntegrateParams), hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc((void **)&b->d_params[i], n * sizeof(tsdfk::IntegrateParams));
        if (e == hipSuccess) e = hipHostMalloc((void **)&b->h_poses[i], n * sizeof(tsdfk::FramePose), hipHostMallocDefault);
        if (e == hipSuccess) e = hipMalloc((void **)&b->d_poses[i], n * sizeof(tsdfk::FramePose));
        if (e == hipSuccess) e = hipEventCreateWithFlags(&b->slot_done[i], hipEventDisableTiming);
    }
    if (e != hipSuccess) return cleanup(fail(TSDF_ERR_HIP, "tsdf_batch_create: %s", hipGetErrorString(e)));
    for (tsdf_volume *v : b->vols) {   // creation fills ran on each volume's own stream: finish them, then share ours
        if (hipStreamSynchronize(v->stream) != hipSuccess) return cleanup(fail(TSDF_ERR_HIP, "tsdf_batch_create: sync failed"));
        v->stream = b->stream;
        v->owner = b;
    }
    *out = b;
    return TSDF_OK;
}

int tsdf_batch_size(const tsdf_batch *b) { return b ? (int)b->vols.size() : 0; }

int tsdf_batch_volume(tsdf_batch *b, int32_t i, tsdf_volume **vol)
{
    if (!b || !vol || i < 0 || i >= (int)b->vols.size()) return fail(TSDF_ERR_INVALID, "tsdf_batch_volume: bad argument");
    *vol = b->vols[i];
    return TSDF_OK;
}

int tsdf_batch_sync(tsdf_batch *b)
{
    if (!b) return fail(TSDF_ERR_INVALID, "tsdf_batch_sync: NULL handle");
    HIP_TRY(hipSetDevice(b->device));
    int rc = batch_flush(b);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(b->stream));
    return TSDF_OK;
}

int tsdf_batch_integrate_device(tsdf_batch *b, const float *depth_dev, const uint8_t *const *masks_dev,
                                const float cam2world[16])
{
    if (!b || !depth_dev || !cam2world) return fail(TSDF_ERR_INVALID, "tsdf_batch_integrate_device: NULL argument");
    HIP_TRY(hipSetDevice(b->device));
    if (b->total_slices == 0) return TSDF_OK;
    const int n = (int)b->vols.size();
    for (tsdf_volume *v : b->vols)
        if (v->pend_count > 0) { int rc = flush_pending(v); if (rc) return rc; }   // frames given to a borrowed handle come first
    // Deferral as for a single handle (tsdf_set_deferral on the FIRST member switches it): collect the frame and apply 32 at
    // a time with one fused launch per member -- the volumes then move once per 32 frames, but every member costs a launch
    // with its own tile tables per flush, so many small members stay with the one batched launch per frame.  Fitted to
    // tools/batch_time.py (instance masks, ms per frame, batched -> deferred): 1 x 200^3 0.035 -> 0.013, 4 x 200^3 0.082
    // -> 0.031, 16 x 200^3 0.145 -> 0.100, 2 x 400^3 0.122 -> 0.046, 8 x 128^3 0.059 -> 0.047; but 16 x 64^3 0.035 ->
    // 0.070, 64 x 100^3 0.229 -> 0.307: deferred costs about 4 us per member + 0.8 us per M voxels, batched 20 us + 2.8.
    {
        bool defer = b->vols[0]->defer_n > 1;
        int64_t total = 0;
        for (tsdf_volume *v : b->vols) { defer = defer && can_fuse(v); total += v->n_vox; }
        defer = defer && 2 * (int64_t)n < 10 + total / 1000000;
        if (defer) return batch_collect(b, depth_dev, masks_dev, cam2world);
        int rc = batch_flush(b);   // the policy changed between calls (tsdf_set_deferral, a kernel variant)
        if (rc) return rc;
    }
    const int s = b->slot_next;
    b->slot_next = (s + 1) % kStageSlots;
    if (b->slot_used[s]) HIP_TRY(hipEventSynchronize(b->slot_done[s]));
    for (int i = 0; i < n; ++i) {
        tsdf_volume *v = b->vols[i];
        float c2b[16];
        compose_cam2base(v, cam2world, c2b);   // each object has its own base frame (ref: src/Object.cpp:23-29)
        std::memcpy(v->last_cam2base, c2b, sizeof c2b);
        const tsdfk::IntegrateParams q = make_params(v, depth_dev, masks_dev ? masks_dev[i] : nullptr, c2b, 4);
        b->h_params[s][i] = q;
        pose_from_params(b->h_poses[s][i], q);
        v->flags_known_zero = false;           // the batched kernel maintains the summary
    }
    // Instance masks given: every object sees only its own instance (ref: src/Engine.cpp:192-193), so most workgroups of
    // most objects see nothing.  Tile tables of depth x mask per object, the class of every workgroup of the launch
    // (one thread each), and the launch's untouched workgroups leave at once.  (First volume on variant 7: never.)
    bool any_mask = false;
    for (int i = 0; i < n && masks_dev; ++i) any_mask = any_mask || masks_dev[i] != nullptr;
    bool same_range = true;   // one tile table per object, but all made with one depth-range test
    for (int i = 1; i < n; ++i) same_range = same_range && b->vols[i]->cfg.max_depth == b->vols[0]->cfg.max_depth;
    int64_t launch_voxels = 0;
    for (tsdf_volume *v : b->vols) launch_voxels += v->n_vox;
    // ... and not for many small volumes: one tile table per object has to be built per frame (64 x 100^3: 0.231 -> 0.262 ms)
    const bool big_enough = b->vols[0]->variant == 8 || (kExperiments && b->vols[0]->variant >= 11 && b->vols[0]->variant <= 13) || launch_voxels >= (int64_t)n * 2000000;
    const bool classify = any_mask && same_range && big_enough && classify_one_frame(b->vols[0], launch_voxels) &&
                          tiles_fit(b->h_params[s][0]);
    if (classify) {
        const size_t per = tile_table_elems_host(b->h_params[s][0].tiles_w, b->h_params[s][0].tiles_h);
        if (!b->d_tiles) {
            HIP_TRY(hipMalloc((void **)&b->d_tiles, (size_t)n * per * sizeof(float2)));
            if (kExperiments) HIP_TRY(hipMalloc((void **)&b->d_wg_class, (size_t)b->max_blocks * b->total_slices));
            b->tiles_per_object = per;
        }
        std::vector<const float *> depths((size_t)n, depth_dev);
        int rc = build_tile_tables(b->stream, b->vols[0]->cfg, b->h_params[s][0], depths.data(), masks_dev, n, b->d_tiles);
        if (rc) return rc;
        for (int i = 0; i < n; ++i) b->h_poses[s][i].tiles = b->d_tiles + (size_t)i * per;
    }
    HIP_TRY(hipMemcpyAsync(b->d_params[s], b->h_params[s], n * sizeof(tsdfk::IntegrateParams), hipMemcpyHostToDevice, b->stream));
    HIP_TRY(hipMemcpyAsync(b->d_poses[s], b->h_poses[s], n * sizeof(tsdfk::FramePose), hipMemcpyHostToDevice, b->stream));
    dim3 block(64, 4, 1), grid(b->max_blocks, 1, b->total_slices);
    // a class per wavefront brick (every member has a brick view: dim_x % 4 == 0 is a condition of tsdf_batch_create)
    int brick_blocks = 0;
    bool bricks = classify && !(kExperiments && b->vols[0]->variant == 11);
    for (int i = 0; i < n && bricks; ++i) {
        const tsdfk::IntegrateParams &q = b->h_params[s][i];
        bricks = q.brick_q > 0;
        brick_blocks = std::max(brick_blocks, (q.brick_groups * q.bricks_per_group + 3) / 4);
    }
    if (bricks) {
        // z index of the brick launches: slice groups (rebuilt when a member's brick depth changed: tsdf_set_brick_shape)
        bool same = b->d_group_map != nullptr && (int)b->group_s.size() == n;
        for (int i = 0; i < n && same; ++i) same = b->group_s[i] == b->h_params[s][i].brick_s;
        if (!same) {
            std::vector<int2> gmap;
            b->group_s.assign((size_t)n, 1);
            for (int i = 0; i < n; ++i) {
                const tsdfk::IntegrateParams &q = b->h_params[s][i];
                b->group_s[i] = q.brick_s;
                for (int g = 0; g < (q.nz + q.brick_s - 1) / q.brick_s; ++g) gmap.push_back(make_int2(i, g));
            }
            HIP_TRY(hipStreamSynchronize(b->stream));
            if (b->d_group_map) HIP_TRY(hipFree(b->d_group_map));
            b->d_group_map = nullptr;
            HIP_TRY(hipMalloc((void **)&b->d_group_map, std::max<size_t>(gmap.size(), 1) * sizeof(int2)));
            HIP_TRY(hipMemcpy(b->d_group_map, gmap.data(), gmap.size() * sizeof(int2), hipMemcpyHostToDevice));
            b->total_groups = (int)gmap.size();
        }
        const size_t n_bricks = (size_t)brick_blocks * b->total_groups * 4;
        if (b->brick_class_bytes < n_bricks) {
            if (b->d_brick_class) HIP_TRY(hipFree(b->d_brick_class));
            b->d_brick_class = nullptr;
            b->brick_class_bytes = 0;
            HIP_TRY(hipMalloc((void **)&b->d_brick_class, n_bricks));
            b->brick_class_bytes = n_bricks;
        }
        hipLaunchKernelGGL(tsdfk::classify_bricks_batched, dim3((unsigned)((n_bricks + 255) / 256)), dim3(256), 0, b->stream,
                           b->d_params[s], b->d_poses[s], b->d_group_map, b->d_brick_class, brick_blocks, b->total_groups);
        hipLaunchKernelGGL((tsdfk::integrate_multi_batched_bricks<kBrickNT>), dim3(brick_blocks, 1, b->total_groups), block, 0, b->stream,
                           b->d_params[s], b->d_poses[s], b->d_group_map, b->d_brick_class);
#ifdef TSDF_EXPERIMENTS
    } else if (classify) {   // variant 11: a class per 1024-voxel workgroup patch (round 2a's shape)
        const size_t n_wg = (size_t)b->max_blocks * b->total_slices;
        hipLaunchKernelGGL(tsdfk::classify_workgroups_batched, dim3((unsigned)((n_wg + 255) / 256)), dim3(256), 0, b->stream,
                           b->d_params[s], b->d_poses[s], b->d_slice_map, b->d_wg_class, b->max_blocks, b->total_slices);
        hipLaunchKernelGGL((tsdfk::integrate_multi_batched_cls<true, true>), grid, block, 0, b->stream, b->d_params[s], b->d_poses[s],
                           b->d_slice_map, b->d_wg_class);
#endif
    } else {
        hipLaunchKernelGGL((tsdfk::integrate_multi_batched<true>), grid, block, 0, b->stream, b->d_params[s], b->d_poses[s],
                           b->d_slice_map);
    }
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipEventRecord(b->slot_done[s], b->stream));
    b->slot_used[s] = true;
    return TSDF_OK;
}

// ---------------------------------------------------------------------------------------------
// surface extraction (ref: src/tsdf.cu:170-218), on the device
// ---------------------------------------------------------------------------------------------
// xyz_host == nullptr with keep_on_device: the list is left in v->d_list (the file writers stream it from there)
static int surface_pass(tsdf_volume *v, float weight_thresh, float *xyz_host, int64_t capacity,
                        int64_t *count, bool keep_on_device = false)
{
    int rc = bind_device(v);
    if (rc) return rc;
    *count = 0;
    if (v->n_vox == 0) return TSDF_OK;
    const int64_t n = v->n_vox;
    const int64_t n_chunks = (n + tsdfx::kChunk - 1) / tsdfx::kChunk;
    if (n_chunks > 0x7fffffff) return fail(TSDF_ERR_INVALID, "surface extraction: slab too large");
    // scratch: per-chunk counts (u32) then per-chunk offsets (i64) then total (i64)
    size_t off_counts = 0;
    size_t off_offsets = ((size_t)n_chunks * sizeof(uint32_t) + 255) & ~(size_t)255;
    size_t off_total = off_offsets + (size_t)n_chunks * sizeof(int64_t);
    size_t need = off_total + 256;
    rc = ensure_scratch(v, need);
    if (rc) return rc;
    char *s = (char *)v->d_scratch;
    uint32_t *d_counts = (uint32_t *)(s + off_counts);
    int64_t *d_offsets = (int64_t *)(s + off_offsets);
    int64_t *d_total = (int64_t *)(s + off_total);

    hipLaunchKernelGGL(tsdfx::surface_count, dim3((unsigned)n_chunks), dim3(256), 0, v->stream,
                       v->d_tsdf, v->d_weight, n, weight_thresh, d_counts);
    hipLaunchKernelGGL(tsdfx::scan_counts, dim3(1), dim3(1024), 0, v->stream, d_counts, n_chunks,
                       d_offsets, d_total);
    HIP_TRY(hipGetLastError());
    int64_t total = 0;
    HIP_TRY(hipMemcpyAsync(&total, d_total, sizeof total, hipMemcpyDeviceToHost, v->stream));
    HIP_TRY(hipStreamSynchronize(v->stream));
    *count = total;
    if ((!xyz_host && !keep_on_device) || capacity <= 0 || total == 0) return TSDF_OK;

    int64_t n_out = total < capacity ? total : capacity;
    rc = ensure_list(v, (size_t)total * 3 * sizeof(float));
    if (rc) return rc;
    float *d_xyz = (float *)v->d_list;
    const tsdf_config &c = v->cfg;
    hipLaunchKernelGGL(tsdfx::surface_emit, dim3((unsigned)n_chunks), dim3(256), 0, v->stream,
                       v->d_tsdf, v->d_weight, n, weight_thresh, d_offsets, c.dim_x, c.dim_y,
                       c.z_begin, c.origin[0], c.origin[1], c.origin[2], c.voxel_size, d_xyz);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && xyz_host)
        e = hipMemcpyAsync(xyz_host, d_xyz, (size_t)n_out * 3 * sizeof(float), hipMemcpyDeviceToHost, v->stream);
    if (e == hipSuccess && xyz_host) e = hipStreamSynchronize(v->stream);
    if (e != hipSuccess) return fail(TSDF_ERR_HIP, "surface extraction: %s", hipGetErrorString(e));
    return TSDF_OK;
}

int tsdf_count_surface(tsdf_volume *v, float weight_thresh, int64_t *count)
{
    if (!v || !count) return fail(TSDF_ERR_INVALID, "tsdf_count_surface: NULL argument");
    return surface_pass(v, weight_thresh, nullptr, 0, count);
}

int tsdf_extract_surface(tsdf_volume *v, float weight_thresh, float *xyz_host, int64_t capacity,
                         int64_t *count)
{
    if (!v || !count) return fail(TSDF_ERR_INVALID, "tsdf_extract_surface: NULL argument");
    return surface_pass(v, weight_thresh, xyz_host, capacity, count);
}

// Zero-crossing vertices.  halo_*: slice z_end from the upper neighbour (host or device memory) or NULL.
static int crossing_pass(tsdf_volume *v, const float *halo_tsdf, const float *halo_weight, float weight_thresh,
                         float *xyz_host, int64_t capacity, int64_t *count, bool mesh = false)
{
    const size_t item_floats = mesh ? 9 : 3;   // triangle = 3 vertices, crossing = 1 vertex
    int rc = bind_device(v);
    if (rc) return rc;
    *count = 0;
    if (v->n_vox == 0) return TSDF_OK;
    if ((halo_tsdf == nullptr) != (halo_weight == nullptr))
        return fail(TSDF_ERR_INVALID, "zero crossings: give both halo arrays or neither");
    const tsdf_config &c = v->cfg;
    const int64_t n = v->n_vox;
    const int64_t n_chunks = (n + tsdfx::kChunk - 1) / tsdfx::kChunk;
    if (n_chunks > 0x7fffffff) return fail(TSDF_ERR_INVALID, "zero crossings: slab too large");
    const size_t slice = (size_t)c.dim_x * c.dim_y;
    // scratch: counts (u32) | offsets (i64) | total (i64) | halo copy (2 slices of floats)
    size_t off_offsets = ((size_t)n_chunks * sizeof(uint32_t) + 255) & ~(size_t)255;
    size_t off_total = off_offsets + (size_t)n_chunks * sizeof(int64_t);
    size_t off_halo = (off_total + 256 + 255) & ~(size_t)255;
    rc = ensure_scratch(v, off_halo + 2 * slice * sizeof(float));
    if (rc) return rc;
    char *s = (char *)v->d_scratch;
    uint32_t *d_counts = (uint32_t *)s;
    int64_t *d_offsets = (int64_t *)(s + off_offsets);
    int64_t *d_total = (int64_t *)(s + off_total);
    float *d_halo = (float *)(s + off_halo);
    tsdfx::CrossingGrid g;
    g.tsdf = v->d_tsdf; g.weight = v->d_weight; g.halo_tsdf = nullptr; g.halo_weight = nullptr;
    if (halo_tsdf) {   // host or device source: stage both slices in our scratch
        HIP_TRY(hipMemcpyAsync(d_halo, halo_tsdf, slice * sizeof(float), hipMemcpyDefault, v->stream));
        HIP_TRY(hipMemcpyAsync(d_halo + slice, halo_weight, slice * sizeof(float), hipMemcpyDefault, v->stream));
        g.halo_tsdf = d_halo; g.halo_weight = d_halo + slice;
    }
    g.n = n; g.dim_x = c.dim_x; g.dim_y = c.dim_y; g.nz = c.z_end - c.z_begin; g.z_begin = c.z_begin;
    g.thr = weight_thresh; g.ox = c.origin[0]; g.oy = c.origin[1]; g.oz = c.origin[2]; g.vs = c.voxel_size;
    g.flags = v->nseg > 0 ? v->d_flags : nullptr; g.nseg = v->nseg;     // segments that are all free / unseen space are skipped
    if (mesh) hipLaunchKernelGGL(tsdfx::mesh_count, dim3((unsigned)n_chunks), dim3(256), 0, v->stream, g, d_counts);
    else hipLaunchKernelGGL(tsdfx::crossing_count, dim3((unsigned)n_chunks), dim3(256), 0, v->stream, g, d_counts);
    hipLaunchKernelGGL(tsdfx::scan_counts, dim3(1), dim3(1024), 0, v->stream, d_counts, n_chunks, d_offsets, d_total);
    HIP_TRY(hipGetLastError());
    int64_t total = 0;
    HIP_TRY(hipMemcpyAsync(&total, d_total, sizeof total, hipMemcpyDeviceToHost, v->stream));
    HIP_TRY(hipStreamSynchronize(v->stream));
    *count = total;
    if (!xyz_host || capacity <= 0 || total == 0) return TSDF_OK;
    const int64_t n_out = total < capacity ? total : capacity;
    rc = ensure_list(v, (size_t)total * item_floats * sizeof(float));
    if (rc) return rc;
    float *d_xyz = (float *)v->d_list;
    if (mesh) hipLaunchKernelGGL(tsdfx::mesh_emit_kernel, dim3((unsigned)n_chunks), dim3(256), 0, v->stream, g, d_offsets, d_xyz);
    else hipLaunchKernelGGL(tsdfx::crossing_emit, dim3((unsigned)n_chunks), dim3(256), 0, v->stream, g, d_offsets, d_xyz);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess)
        e = hipMemcpyAsync(xyz_host, d_xyz, (size_t)n_out * item_floats * sizeof(float), hipMemcpyDeviceToHost, v->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(v->stream);
    if (e != hipSuccess) return fail(TSDF_ERR_HIP, "zero crossings: %s", hipGetErrorString(e));
    return TSDF_OK;
}

int tsdf_extract_crossings(tsdf_volume *v, const float *halo_tsdf, const float *halo_weight, float weight_thresh,
                           float *xyz_host, int64_t capacity, int64_t *count)
{
    if (!v || !count) return fail(TSDF_ERR_INVALID, "tsdf_extract_crossings: NULL argument");
    return crossing_pass(v, halo_tsdf, halo_weight, weight_thresh, xyz_host, capacity, count);
}

int tsdf_extract_mesh(tsdf_volume *v, const float *halo_tsdf, const float *halo_weight, float weight_thresh,
                      float *triangles_host, int64_t capacity, int64_t *count)
{
    if (!v || !count) return fail(TSDF_ERR_INVALID, "tsdf_extract_mesh: NULL argument");
    return crossing_pass(v, halo_tsdf, halo_weight, weight_thresh, triangles_host, capacity, count, true);
}

int tsdf_save_mesh_ply(tsdf_volume *v, const char *path, float weight_thresh)
{
    if (!v || !path) return fail(TSDF_ERR_INVALID, "tsdf_save_mesh_ply: NULL argument");
    int64_t n = 0;
    int rc = crossing_pass(v, nullptr, nullptr, weight_thresh, nullptr, 0, &n, true);
    if (rc) return rc;
    std::vector<float> tri((size_t)(n > 0 ? n : 1) * 9);
    if (n > 0) {
        rc = crossing_pass(v, nullptr, nullptr, weight_thresh, tri.data(), n, &n, true);
        if (rc) return rc;
    }
    if (v->d_colour) {
        // vertex colour = colour of the nearest voxel (what the Python glue's get_mesh does with the rounded vertex indices)
        std::vector<uint32_t> col((size_t)(v->n_vox > 0 ? v->n_vox : 1));
        rc = tsdf_download_colour(v, col.data());
        if (rc) return rc;
        const tsdf_config &c = v->cfg;
        const int nz = c.z_end - c.z_begin;
        std::vector<unsigned char> rgb((size_t)(n > 0 ? n : 1) * 9);
        for (int64_t k = 0; k < 3 * n; ++k) {
            const float *p = tri.data() + 3 * k;
            long ix = std::lround((p[0] - c.origin[0]) / c.voxel_size), iy = std::lround((p[1] - c.origin[1]) / c.voxel_size);
            long iz = std::lround((p[2] - c.origin[2]) / c.voxel_size) - c.z_begin;
            ix = std::min<long>(std::max<long>(ix, 0), c.dim_x - 1);
            iy = std::min<long>(std::max<long>(iy, 0), c.dim_y - 1);
            iz = std::min<long>(std::max<long>(iz, 0), nz - 1);
            const uint32_t q = col[((size_t)iz * c.dim_y + (size_t)iy) * c.dim_x + (size_t)ix];
            rgb[3 * k] = (unsigned char)(q & 255u); rgb[3 * k + 1] = (unsigned char)((q >> 8) & 255u);
            rgb[3 * k + 2] = (unsigned char)((q >> 16) & 255u);
        }
        return write_mesh_ply(path, tri.data(), n, "tsdf_save_mesh_ply", rgb.data());
    }
    return write_mesh_ply(path, tri.data(), n, "tsdf_save_mesh_ply");
}

// The mesh as the reference's Python glue saves it (ref: src/TSDFfusion.py.in:48-53: get_mesh -> verts, faces, norms,
// colors -> meshwrite): shared vertices, a normal per vertex, a colour per vertex when colour is enabled.  The triangle soup's
// edge vertices are bit-identical between neighbouring cubes (tsdf_extract.hip.h), so welding is an exact match on the three
// coordinates' bits; normals are the area-weighted sums of the face normals; faces keep the soup's order and winding.
int tsdf_save_mesh_welded_ply(tsdf_volume *v, const char *path, float weight_thresh)
{
    if (!v || !path) return fail(TSDF_ERR_INVALID, "tsdf_save_mesh_welded_ply: NULL argument");
    int64_t n = 0;
    int rc = crossing_pass(v, nullptr, nullptr, weight_thresh, nullptr, 0, &n, true);
    if (rc) return rc;
    if (3 * n > 0x7fffffffll) return fail(TSDF_ERR_INVALID, "tsdf_save_mesh_welded_ply: %lld triangles exceed 32-bit vertex indices", (long long)n);
    std::vector<float> tri((size_t)(n > 0 ? n : 1) * 9);
    if (n > 0 && (rc = crossing_pass(v, nullptr, nullptr, weight_thresh, tri.data(), n, &n, true)) != TSDF_OK) return rc;
    struct Key { uint32_t x, y, z; bool operator==(const Key &o) const { return x == o.x && y == o.y && z == o.z; } };
    struct Hash { size_t operator()(const Key &k) const { uint64_t h = k.x * 0x9E3779B97F4A7C15ull; h ^= (h >> 29) + k.y * 0xBF58476D1CE4E5B9ull; h ^= (h >> 31) + k.z * 0x94D049BB133111EBull; return (size_t)(h ^ (h >> 32)); } };
    std::unordered_map<Key, int32_t, Hash> ids;
    ids.reserve((size_t)n);
    std::vector<float> verts;
    std::vector<int32_t> faces((size_t)(n > 0 ? n : 1) * 3);
    for (int64_t k = 0; k < 3 * n; ++k) {
        Key key;
        std::memcpy(&key, tri.data() + 3 * k, 12);
        auto it = ids.find(key);
        if (it == ids.end()) {
            it = ids.emplace(key, (int32_t)(verts.size() / 3)).first;
            verts.insert(verts.end(), tri.data() + 3 * k, tri.data() + 3 * k + 3);
        }
        faces[(size_t)k] = it->second;
    }
    const size_t nv = verts.size() / 3;
    std::vector<double> acc(nv * 3 + 3, 0.0);
    for (int64_t f = 0; f < n; ++f) {
        const float *a = tri.data() + 9 * f, *b = a + 3, *c = a + 6;
        const double ux = (double)b[0] - a[0], uy = (double)b[1] - a[1], uz = (double)b[2] - a[2];
        const double wx = (double)c[0] - a[0], wy = (double)c[1] - a[1], wz = (double)c[2] - a[2];
        const double nx = uy * wz - uz * wy, ny = uz * wx - ux * wz, nz = ux * wy - uy * wx;   // 2 x area x unit normal
        for (int k = 0; k < 3; ++k) { double *q = acc.data() + 3 * (size_t)faces[(size_t)(3 * f + k)]; q[0] += nx; q[1] += ny; q[2] += nz; }
    }
    std::vector<uint32_t> col;
    if (v->d_colour) {
        col.resize((size_t)(v->n_vox > 0 ? v->n_vox : 1));
        rc = tsdf_download_colour(v, col.data());
        if (rc) return rc;
    }
    const tsdf_config &c = v->cfg;
    const int nz_ = c.z_end - c.z_begin;
    const size_t rec_bytes = 24 + (v->d_colour ? 3 : 0);
    std::vector<unsigned char> rec((nv > 0 ? nv : 1) * rec_bytes);
    for (size_t i = 0; i < nv; ++i) {
        const float *p = verts.data() + 3 * i;
        const double *q = acc.data() + 3 * i;
        const double len = std::sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
        const float nrm[3] = {len > 0 ? (float)(q[0] / len) : 0.0f, len > 0 ? (float)(q[1] / len) : 0.0f, len > 0 ? (float)(q[2] / len) : 0.0f};
        unsigned char *r = rec.data() + i * rec_bytes;
        std::memcpy(r, p, 12);
        std::memcpy(r + 12, nrm, 12);
        if (v->d_colour) {   // the nearest voxel's colour (the Python glue indexes its colour volume with the rounded vertex)
            long ix = std::lround((p[0] - c.origin[0]) / c.voxel_size), iy = std::lround((p[1] - c.origin[1]) / c.voxel_size);
            long iz = std::lround((p[2] - c.origin[2]) / c.voxel_size) - c.z_begin;
            ix = std::min<long>(std::max<long>(ix, 0), c.dim_x - 1);
            iy = std::min<long>(std::max<long>(iy, 0), c.dim_y - 1);
            iz = std::min<long>(std::max<long>(iz, 0), nz_ - 1);
            const uint32_t u = col[((size_t)iz * c.dim_y + (size_t)iy) * c.dim_x + (size_t)ix];
            r[24] = (unsigned char)(u & 255u); r[25] = (unsigned char)((u >> 8) & 255u); r[26] = (unsigned char)((u >> 16) & 255u);
        }
    }
    FILE *fp = std::fopen(path, "wb");
    if (!fp) return fail(TSDF_ERR_IO, "tsdf_save_mesh_welded_ply: cannot open %s", path);
    std::fprintf(fp, "ply\nformat binary_little_endian 1.0\nelement vertex %zu\n", nv);
    std::fprintf(fp, "property float x\nproperty float y\nproperty float z\nproperty float nx\nproperty float ny\nproperty float nz\n");
    if (v->d_colour) std::fprintf(fp, "property uchar red\nproperty uchar green\nproperty uchar blue\n");
    std::fprintf(fp, "element face %lld\nproperty list uchar int vertex_index\nend_header\n", (long long)n);
    size_t ok = std::fwrite(rec.data(), rec_bytes, nv, fp);
    std::vector<unsigned char> fr((size_t)(n > 0 ? n : 1) * 13);
    for (int64_t f = 0; f < n; ++f) { fr[13 * f] = 3; std::memcpy(fr.data() + 13 * f + 1, faces.data() + 3 * f, 12); }
    ok += std::fwrite(fr.data(), 13, (size_t)n, fp);
    const int bad = std::fclose(fp);
    if (ok != nv + (size_t)n || bad) return fail(TSDF_ERR_IO, "tsdf_save_mesh_welded_ply: short write to %s", path);
    return TSDF_OK;
}

int tsdf_save_ply(tsdf_volume *v, const char *path, float weight_thresh)
{
    if (!v || !path) return fail(TSDF_ERR_INVALID, "tsdf_save_ply: NULL argument");
    // one counting + emitting pass that leaves the points in device memory, then header + list streamed to the file
    int64_t n = 0;
    int rc = surface_pass(v, weight_thresh, nullptr, INT64_MAX, &n, true);
    if (rc) return rc;
    if (n > 0x7fffffffll)   // the header's "element vertex %d" (ref: src/tsdf.cu:188) cannot hold it
        return fail(TSDF_ERR_INVALID, "tsdf_save_ply: %lld surface points exceed the format's 2^31 - 1 (write slabs separately)", (long long)n);
    FILE *fp = std::fopen(path, "w");
    if (!fp) return fail(TSDF_ERR_IO, "tsdf_save_ply: cannot open %s", path);
    bool ok = std::fprintf(fp, "ply\nformat binary_little_endian 1.0\nelement vertex %d\n", (int)n) > 0 &&
              std::fprintf(fp, "property float x\nproperty float y\nproperty float z\nend_header\n") > 0;
    rc = ok ? stream_device_to_file(v, fp, v->d_list, (size_t)n * 3 * sizeof(float), "tsdf_save_ply", path) : TSDF_OK;
    const int bad = std::fclose(fp);
    if (rc) return rc;
    if (!ok || bad) return fail(TSDF_ERR_IO, "tsdf_save_ply: short write to %s", path);
    return TSDF_OK;
}

int tsdf_save_bin(tsdf_volume *v, const char *path)
{
    if (!v || !path) return fail(TSDF_ERR_INVALID, "tsdf_save_bin: NULL argument");
    int rc = bind_device(v);
    if (rc) return rc;
    const tsdf_config &c = v->cfg;
    FILE *fp = std::fopen(path, "wb");
    if (!fp) return fail(TSDF_ERR_IO, "tsdf_save_bin: cannot open %s", path);
    const float hdr[8] = {(float)c.dim_x, (float)c.dim_y, (float)(c.z_end - c.z_begin), c.origin[0], c.origin[1], c.origin[2],
                          c.voxel_size, c.trunc_margin};       // ref: src/tsdf.cu:118-129
    const bool ok = std::fwrite(hdr, sizeof(float), 8, fp) == 8;
    rc = ok ? stream_device_to_file(v, fp, v->d_tsdf, (size_t)v->n_vox * sizeof(float), "tsdf_save_bin", path) : TSDF_OK;
    const int bad = std::fclose(fp);
    if (rc) return rc;
    if (!ok || bad) return fail(TSDF_ERR_IO, "tsdf_save_bin: short write to %s", path);
    return TSDF_OK;
}

// ---------------------------------------------------------------------------------------------
// checkpoint / resume (the reference only ever writes: ref src/tsdf.cu:114-132; nothing reads a .bin back)
// ---------------------------------------------------------------------------------------------
int tsdf_load_bin(tsdf_volume *v, const char *path)
{
    if (!v || !path) return fail(TSDF_ERR_INVALID, "tsdf_load_bin: NULL argument");
    FILE *fp = std::fopen(path, "rb");
    if (!fp) return fail(TSDF_ERR_IO, "tsdf_load_bin: cannot open %s", path);
    float hdr[8];
    const tsdf_config &c = v->cfg;
    const int nz = c.z_end - c.z_begin;
    bool ok = std::fread(hdr, sizeof(float), 8, fp) == 8 && hdr[0] == (float)c.dim_x && hdr[1] == (float)c.dim_y &&
              hdr[2] == (float)nz;
    std::vector<float> host((size_t)(v->n_vox > 0 ? v->n_vox : 1));
    ok = ok && std::fread(host.data(), sizeof(float), (size_t)v->n_vox, fp) == (size_t)v->n_vox;
    std::fclose(fp);
    if (!ok) return fail(TSDF_ERR_IO, "tsdf_load_bin: %s is not a %dx%dx%d TSDF dump", path, c.dim_x, c.dim_y, nz);
    return tsdf_upload(v, host.data(), nullptr);   // weights are not in the reference's format
}

static const char kStateMagic[8] = {'T', 'S', 'D', 'F', 'H', 'I', 'P', '1'};

int tsdf_save_state(tsdf_volume *v, const char *path)
{
    if (!v || !path) return fail(TSDF_ERR_INVALID, "tsdf_save_state: NULL argument");
    int rc = bind_device(v);
    if (rc) return rc;
    FILE *fp = std::fopen(path, "wb");
    if (!fp) return fail(TSDF_ERR_IO, "tsdf_save_state: cannot open %s", path);
    const bool ok = std::fwrite(kStateMagic, 1, 8, fp) == 8 && std::fwrite(&v->cfg, sizeof(tsdf_config), 1, fp) == 1;
    rc = ok ? stream_device_to_file(v, fp, v->d_tsdf, (size_t)v->n_vox * sizeof(float), "tsdf_save_state", path) : TSDF_OK;
    if (ok && rc == TSDF_OK) rc = stream_device_to_file(v, fp, v->d_weight, (size_t)v->n_vox * sizeof(float), "tsdf_save_state", path);
    const int bad = std::fclose(fp);
    if (rc) return rc;
    if (!ok || bad) return fail(TSDF_ERR_IO, "tsdf_save_state: short write to %s", path);
    return TSDF_OK;
}

int tsdf_load_state(tsdf_volume *v, const char *path)
{
    if (!v || !path) return fail(TSDF_ERR_INVALID, "tsdf_load_state: NULL argument");
    FILE *fp = std::fopen(path, "rb");
    if (!fp) return fail(TSDF_ERR_IO, "tsdf_load_state: cannot open %s", path);
    char magic[8];
    tsdf_config c;
    bool ok = std::fread(magic, 1, 8, fp) == 8 && std::memcmp(magic, kStateMagic, 8) == 0 &&
              std::fread(&c, sizeof c, 1, fp) == 1;
    ok = ok && c.dim_x == v->cfg.dim_x && c.dim_y == v->cfg.dim_y && c.dim_z == v->cfg.dim_z &&
         c.z_begin == v->cfg.z_begin && c.z_end == v->cfg.z_end;
    std::vector<float> t((size_t)(v->n_vox > 0 ? v->n_vox : 1)), w(t.size());
    ok = ok && std::fread(t.data(), sizeof(float), (size_t)v->n_vox, fp) == (size_t)v->n_vox &&
         std::fread(w.data(), sizeof(float), (size_t)v->n_vox, fp) == (size_t)v->n_vox;
    std::fclose(fp);
    if (!ok) return fail(TSDF_ERR_IO, "tsdf_load_state: %s does not hold the state of this slab", path);
    return tsdf_upload(v, t.data(), w.data());
}

}  // extern "C"

#include "tsdf_group.hip.h"
