// tsdf_group.hip.h -- one grid over several devices in ONE process (include/tsdf_hip.h, tsdf_group_*).
//
// Included at the end of tsdf_capi.hip (it drives the handles' internals: staging buffers, streams, arrays).
//
// The reference's host is a C++ program that owns its TSDFs directly (ref: include/tsdf.hpp:22-43; src/Engine.cpp:170-172
// drives distinct TSDFs from one loop), so spanning the node must not need a process per GPU: a group cuts the grid
// into n contiguous z-slabs (the layout is z-major, ref: src/tsdf.cu:52), one `tsdf_volume` handle each, slab i
// on devices[i] with its own stream.
//   * Integrate: the caller's depth frame is copied once into a pinned buffer every device can read, then fanned out
//     with one hipMemcpyAsync per slab on that slab's copy stream (it overlaps the slab's previous kernel), followed
//     by that slab's kernel -- no host
//     synchronisation between devices, no collective; every slab uses the GLOBAL z index, so the result is
//     bit-identical to one handle holding the whole grid.
//   * Extraction: the reference's surface rule is per voxel; zero crossings and the mesh take slice z_end from the
//     next slab -- a device-to-device hipMemcpyPeerAsync of dim_y*dim_x*8 bytes over xGMI -- and run one host thread
//     per slab so the devices work concurrently.  Lists are concatenated in z order: identical to the unsharded list.
//   * Writers: the gathered grid / list written once, byte-identical to a whole-grid handle's files.
#pragma once
#include <functional>
#include <thread>

struct tsdf_group {
    tsdf_config cfg;                      // the GLOBAL grid (z_begin = 0, z_end = dim_z)
    std::vector<tsdf_volume *> slabs;     // slab i: z in [i*dim_z/n, (i+1)*dim_z/n), on devices[i]
    std::vector<int> devices;
    // pinned frames every device can read (hipHostMallocPortable): a ring for single frames, a pool for sequences
    float *h_ring[kStageSlots];
    int ring_next;
    std::vector<hipEvent_t> ring_copied;  // [ring slot][slab]: that slab's copy out of the slot
    std::vector<bool> ring_used;
    float *h_pool;                        // kMaxFramesPerLaunch frames, allocated on first tsdf_group_integrate_frames
    std::vector<float *> d_pool;          // per slab: the same frames in its device's memory
    std::vector<hipEvent_t> pool_done;    // per slab: the copy of its last pass out of h_pool has run
    std::vector<bool> pool_used;
    // deferred integration (as tsdf_integrate on one handle): tsdf_group_integrate collects frames in h_pool and applies
    // defer_n of them per pass as one fused launch per slab; every other group entry point flushes first
    int defer_n, pend_count;
    float pend_poses[16 * tsdfk::kMaxFramesPerLaunch];
    std::vector<float *> d_halo;          // per slab: slice z_end of the next slab (tsdf, then weight), on first use
};

namespace {

// Run fn(i) for every slab, one host thread each (the calls block on their device), and hand the first failure --
// code and message -- back to the calling thread (tsdf_last_error() is thread-local).
int for_each_slab(tsdf_group *g, const std::function<int(int)> &fn)
{
    const int n = (int)g->slabs.size();
    std::vector<int> rc((size_t)n, TSDF_OK);
    std::vector<std::string> msg((size_t)n);
    if (n == 1) {
        return fn(0);
    }
    std::vector<std::thread> th;
    for (int i = 0; i < n; ++i)
        th.emplace_back([&, i]() {
            rc[(size_t)i] = fn(i);
            if (rc[(size_t)i] != TSDF_OK) msg[(size_t)i] = g_last_error;
        });
    for (auto &t : th) t.join();
    for (int i = 0; i < n; ++i)
        if (rc[(size_t)i] != TSDF_OK) return fail(rc[(size_t)i], "slab %d (device %d): %s", i, g->devices[(size_t)i], msg[(size_t)i].c_str());
    return TSDF_OK;
}

// Slice z_end of slab i from slab i + 1, device to device, queued on slab i's stream (which the extraction then uses).
int fetch_halo(tsdf_group *g, int i, const float **ht, const float **hw)
{
    *ht = *hw = nullptr;
    if (i + 1 >= (int)g->slabs.size()) return TSDF_OK;
    tsdf_volume *lo = g->slabs[(size_t)i], *hi = g->slabs[(size_t)i + 1];
    const size_t slice = (size_t)g->cfg.dim_x * g->cfg.dim_y, bytes = slice * sizeof(float);
    HIP_TRY(hipSetDevice(hi->cfg.device));
    HIP_TRY(hipStreamSynchronize(hi->stream));          // the neighbour's integrations have finished
    HIP_TRY(hipSetDevice(lo->cfg.device));
    if (!g->d_halo[(size_t)i]) HIP_TRY(hipMalloc((void **)&g->d_halo[(size_t)i], 2 * bytes));
    float *buf = g->d_halo[(size_t)i];
    HIP_TRY(hipMemcpyPeerAsync(buf, lo->cfg.device, hi->d_tsdf, hi->cfg.device, bytes, lo->stream));
    HIP_TRY(hipMemcpyPeerAsync(buf + slice, lo->cfg.device, hi->d_weight, hi->cfg.device, bytes, lo->stream));
    *ht = buf;
    *hw = buf + slice;
    return TSDF_OK;
}

enum class ListKind { Surface, Crossings, Mesh };

// count pass on every slab (concurrently), then -- when a destination is given -- the emit passes into the right offsets
int group_list(tsdf_group *g, ListKind kind, float weight_thresh, float *out_host, int64_t capacity, int64_t *count)
{
    int rc_flush = group_flush(g);
    if (rc_flush) return rc_flush;
    const int n = (int)g->slabs.size();
    const size_t item = kind == ListKind::Mesh ? 9 : 3;
    std::vector<int64_t> cnt((size_t)n, 0);
    std::vector<const float *> ht((size_t)n, nullptr), hw((size_t)n, nullptr);
    auto pass = [&](int i, float *dst, int64_t cap, int64_t *c) -> int {
        tsdf_volume *v = g->slabs[(size_t)i];
        if (kind == ListKind::Surface) return surface_pass(v, weight_thresh, dst, cap, c);
        return crossing_pass(v, ht[(size_t)i], hw[(size_t)i], weight_thresh, dst, cap, c, kind == ListKind::Mesh);
    };
    int rc = for_each_slab(g, [&](int i) -> int {
        if (kind != ListKind::Surface) {
            int r = fetch_halo(g, i, &ht[(size_t)i], &hw[(size_t)i]);
            if (r) return r;
        }
        return pass(i, nullptr, 0, &cnt[(size_t)i]);
    });
    if (rc) return rc;
    int64_t total = 0;
    std::vector<int64_t> off((size_t)n, 0);
    for (int i = 0; i < n; ++i) { off[(size_t)i] = total; total += cnt[(size_t)i]; }
    *count = total;
    if (!out_host || capacity <= 0 || total == 0) return TSDF_OK;
    return for_each_slab(g, [&](int i) -> int {
        const int64_t room = capacity - off[(size_t)i];
        if (cnt[(size_t)i] == 0 || room <= 0) return TSDF_OK;
        int64_t c = 0;
        return pass(i, out_host + (size_t)off[(size_t)i] * item, room, &c);
    });
}

// The first n frames of h_pool (already there) into every slab: one copy and one fused launch per slab.
int group_pass_from_pool(tsdf_group *g, const float *cam2world, int n)
{
    const size_t px = (size_t)g->cfg.im_height * g->cfg.im_width, img = px * sizeof(float);
    for (size_t i = 0; i < g->slabs.size(); ++i) {
        tsdf_volume *v = g->slabs[i];
        int rc0 = bind_device(v);
        if (rc0) return rc0;
        if (!g->d_pool[i]) {
            HIP_TRY(hipMalloc((void **)&g->d_pool[i], (size_t)tsdfk::kMaxFramesPerLaunch * img));
            HIP_TRY(hipEventCreateWithFlags(&g->pool_done[i], hipEventDisableTiming));
        }
        // on the slab's stream: ordered after the launch that read d_pool in the previous pass
        HIP_TRY(hipMemcpyAsync(g->d_pool[i], g->h_pool, (size_t)n * img, hipMemcpyHostToDevice, v->stream));
        HIP_TRY(hipEventRecord(g->pool_done[i], v->stream));
        g->pool_used[i] = true;
        const float *ptrs[tsdfk::kMaxFramesPerLaunch];
        for (int f = 0; f < n; ++f) ptrs[f] = g->d_pool[i] + (size_t)f * px;
        int rc = integrate_frames(v, ptrs, nullptr, cam2world, n);
        if (rc) return rc;
    }
    return TSDF_OK;
}

// h_pool may be overwritten when every slab's copy out of it has run.
int group_wait_pool(tsdf_group *g)
{
    for (size_t i = 0; i < g->slabs.size(); ++i) {
        if (g->pool_used[i]) {
            HIP_TRY(hipSetDevice(g->slabs[i]->cfg.device));
            HIP_TRY(hipEventSynchronize(g->pool_done[i]));
        }
    }
    return TSDF_OK;
}

// (called by every group entry point and, through bind_device, by every entry point of a borrowed slab handle)
int group_flush(tsdf_group *g)
{
    if (g->pend_count == 0) return TSDF_OK;
    const int n = g->pend_count;
    g->pend_count = 0;
    return group_pass_from_pool(g, g->pend_poses, n);
}

}  // namespace

extern "C" {

int tsdf_group_destroy(tsdf_group *g)
{
    if (!g) return TSDF_OK;
    for (size_t i = 0; i < g->slabs.size(); ++i) {
        tsdf_volume *v = g->slabs[i];
        if (!v) continue;
        v->group_owner = nullptr;     // frames collected but never observed go with the group
        (void)hipSetDevice(v->cfg.device);
        (void)hipStreamSynchronize(v->stream);
        if (i < g->d_pool.size() && g->d_pool[i]) (void)hipFree(g->d_pool[i]);
        if (i < g->pool_done.size() && g->pool_done[i]) (void)hipEventDestroy(g->pool_done[i]);
        if (i < g->d_halo.size() && g->d_halo[i]) (void)hipFree(g->d_halo[i]);
        tsdf_destroy(v);
    }
    for (hipEvent_t e : g->ring_copied)
        if (e) (void)hipEventDestroy(e);
    for (int s = 0; s < kStageSlots; ++s)
        if (g->h_ring[s]) (void)hipHostFree(g->h_ring[s]);
    if (g->h_pool) (void)hipHostFree(g->h_pool);
    delete g;
    return TSDF_OK;
}

int tsdf_group_create(const tsdf_config *cfg, const int32_t *devices, int32_t n_slabs, tsdf_group **out)
{
    if (!cfg || !devices || !out || n_slabs <= 0) return fail(TSDF_ERR_INVALID, "tsdf_group_create: bad argument");
    *out = nullptr;
    if (cfg->dim_z <= 0 || n_slabs > cfg->dim_z)
        return fail(TSDF_ERR_INVALID, "tsdf_group_create: %d slabs for %d slices (every slab needs at least one)", n_slabs, cfg->dim_z);
    tsdf_group *g = new (std::nothrow) tsdf_group();
    if (!g) return fail(TSDF_ERR_INVALID, "tsdf_group_create: out of host memory");
    g->cfg = *cfg;
    g->cfg.z_begin = 0;
    g->cfg.z_end = cfg->dim_z;
    g->cfg.device = devices[0];
    g->ring_next = 0;
    g->h_pool = nullptr;
    g->defer_n = tsdfk::kMaxFramesPerLaunch;
    g->pend_count = 0;
    for (int s = 0; s < kStageSlots; ++s) g->h_ring[s] = nullptr;
    auto cleanup = [&](int code) { const std::string keep = g_last_error; tsdf_group_destroy(g); g_last_error = keep; return code; };
    for (int i = 0; i < n_slabs; ++i) {
        tsdf_config c = *cfg;
        c.z_begin = (int32_t)((int64_t)i * cfg->dim_z / n_slabs);
        c.z_end = (int32_t)((int64_t)(i + 1) * cfg->dim_z / n_slabs);
        c.device = devices[i];
        tsdf_volume *v = nullptr;
        int rc = tsdf_create(&c, &v);
        if (rc) return cleanup(rc);
        v->group_owner = g;           // every entry point of the slab's handle applies the group's collected frames first
        g->slabs.push_back(v);
        g->devices.push_back(devices[i]);
    }
    g->d_pool.assign((size_t)n_slabs, nullptr);
    g->pool_done.assign((size_t)n_slabs, nullptr);
    g->pool_used.assign((size_t)n_slabs, false);
    g->d_halo.assign((size_t)n_slabs, nullptr);
    g->ring_copied.assign((size_t)n_slabs * kStageSlots, nullptr);
    g->ring_used.assign((size_t)n_slabs * kStageSlots, false);
    const size_t img = (size_t)cfg->im_height * cfg->im_width * sizeof(float);
    for (int s = 0; s < kStageSlots; ++s) {
        hipError_t e = hipHostMalloc((void **)&g->h_ring[s], img, hipHostMallocPortable);
        if (e != hipSuccess) return cleanup(fail(TSDF_ERR_HIP, "tsdf_group_create: pinned frame: %s", hipGetErrorString(e)));
    }
    // neighbouring slabs on different devices exchange one slice at extraction: let the copy go directly over xGMI
    for (int i = 0; i + 1 < n_slabs; ++i) {
        const int a = devices[i], b = devices[i + 1];
        if (a == b) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, a, b) == hipSuccess && can) {
            if (hipSetDevice(a) == hipSuccess) {
                const hipError_t e = hipDeviceEnablePeerAccess(b, 0);
                if (e != hipSuccess) (void)hipGetLastError();   // already enabled (or unavailable: the copy is then staged)
            }
        }
    }
    *out = g;
    return TSDF_OK;
}

int tsdf_group_size(const tsdf_group *g) { return g ? (int)g->slabs.size() : 0; }

int64_t tsdf_group_voxels(const tsdf_group *g)
{
    return g ? (int64_t)g->cfg.dim_x * g->cfg.dim_y * g->cfg.dim_z : 0;
}

int tsdf_group_volume(tsdf_group *g, int32_t i, tsdf_volume **vol)
{
    if (!g || !vol || i < 0 || i >= (int)g->slabs.size()) return fail(TSDF_ERR_INVALID, "tsdf_group_volume: bad argument");
    int rc = group_flush(g);       // (the handle's own entry points do this too: frames given to the group later are applied
    if (rc) return rc;             //  before anything observes or integrates through the borrowed handle)
    *vol = g->slabs[(size_t)i];
    return TSDF_OK;
}

int tsdf_group_integrate(tsdf_group *g, const float *depth_host, const float cam2world[16])
{
    if (!g || !depth_host || !cam2world) return fail(TSDF_ERR_INVALID, "tsdf_group_integrate: NULL argument");
    if (g->defer_n > 1) {
        // deferred (see tsdf_integrate): the frame into the pinned pool, launched defer_n at a time as one fused sequence
        const size_t px = (size_t)g->cfg.im_height * g->cfg.im_width, bytes = px * sizeof(float);
        if (!g->h_pool) HIP_TRY(hipHostMalloc((void **)&g->h_pool, (size_t)tsdfk::kMaxFramesPerLaunch * bytes, hipHostMallocPortable));
        if (g->pend_count == 0) { int rc = group_wait_pool(g); if (rc) return rc; }
        tsdf_host::copy_to_pinned(g->h_pool + (size_t)g->pend_count * px, depth_host, bytes);
        std::memcpy(g->pend_poses + 16 * g->pend_count, cam2world, 16 * sizeof(float));
        if (++g->pend_count >= std::min(g->defer_n, (int)tsdfk::kMaxFramesPerLaunch)) return group_flush(g);
        return TSDF_OK;
    }
    const int s = g->ring_next;
    g->ring_next = (s + 1) % kStageSlots;
    const size_t img = (size_t)g->cfg.im_height * g->cfg.im_width * sizeof(float);
    const size_t n = g->slabs.size();
    // the pinned slot is free again when every slab's copy out of it has run
    for (size_t i = 0; i < n; ++i) {
        if (g->ring_used[(size_t)s * n + i]) {
            HIP_TRY(hipSetDevice(g->slabs[i]->cfg.device));
            HIP_TRY(hipEventSynchronize(g->ring_copied[(size_t)s * n + i]));
        }
    }
    tsdf_host::copy_to_pinned(g->h_ring[s], depth_host, img);          // the caller may free depth_host after we return
    for (size_t i = 0; i < n; ++i) {
        tsdf_volume *v = g->slabs[i];
        int rc0 = bind_device(v);      // device current; frames given to a borrowed slab handle come first
        if (rc0) return rc0;
        // a frame slot of the slab's store, filled on its copy stream (overlapping the slab's previous kernel)
        int slot = -1;
        void *dev = nullptr;
        int rc = store_slot(v, &v->store->frames, v->copy_stream, &slot, &dev);
        if (rc) return rc;
        HIP_TRY(hipMemcpyAsync(dev, g->h_ring[s], img, hipMemcpyHostToDevice, v->copy_stream));
        if (!g->ring_copied[(size_t)s * n + i]) HIP_TRY(hipEventCreateWithFlags(&g->ring_copied[(size_t)s * n + i], hipEventDisableTiming));
        HIP_TRY(hipEventRecord(g->ring_copied[(size_t)s * n + i], v->copy_stream));
        g->ring_used[(size_t)s * n + i] = true;
        HIP_TRY(hipStreamWaitEvent(v->stream, g->ring_copied[(size_t)s * n + i], 0));
        float c2b[16];
        compose_cam2base(v, cam2world, c2b);
        rc = launch_integrate(v, static_cast<const float *>(dev), nullptr, c2b);
        if (rc) return rc;
        rc = stage_end(v, &slot, 1);
        if (rc) return rc;
    }
    return TSDF_OK;
}

int tsdf_group_integrate_frames(tsdf_group *g, const float *const *depth_host, const float *cam2world, int32_t n_frames)
{
    if (!g || !depth_host || !cam2world || n_frames < 0) return fail(TSDF_ERR_INVALID, "tsdf_group_integrate_frames: bad argument");
    for (int k = 0; k < n_frames; ++k)
        if (!depth_host[k]) return fail(TSDF_ERR_INVALID, "tsdf_group_integrate_frames: depth_host[%d] is NULL", k);
    int rc = group_flush(g);
    if (rc) return rc;
    const size_t px = (size_t)g->cfg.im_height * g->cfg.im_width, img = px * sizeof(float);
    const int fpl = tsdfk::kMaxFramesPerLaunch;
    if (!g->h_pool) HIP_TRY(hipHostMalloc((void **)&g->h_pool, (size_t)fpl * img, hipHostMallocPortable));
    for (int k = 0; k < n_frames; k += fpl) {
        const int n = std::min(fpl, n_frames - k);
        rc = group_wait_pool(g);     // the pool is reused per pass
        if (rc) return rc;
        for (int f = 0; f < n; ++f) tsdf_host::copy_to_pinned(g->h_pool + (size_t)f * px, depth_host[k + f], img);
        rc = group_pass_from_pool(g, cam2world + 16 * (size_t)k, n);
        if (rc) return rc;
    }
    return TSDF_OK;
}

int tsdf_group_set_deferral(tsdf_group *g, int32_t n_frames)
{
    if (!g) return fail(TSDF_ERR_INVALID, "tsdf_group_set_deferral: NULL handle");
    if (n_frames < 0 || n_frames > tsdfk::kMaxFramesPerLaunch)
        return fail(TSDF_ERR_INVALID, "tsdf_group_set_deferral: n_frames must be in [0, %d]", tsdfk::kMaxFramesPerLaunch);
    int rc = group_flush(g);
    if (rc) return rc;
    g->defer_n = n_frames;
    return TSDF_OK;
}

int tsdf_group_sync(tsdf_group *g)
{
    if (!g) return fail(TSDF_ERR_INVALID, "tsdf_group_sync: NULL handle");
    int rc0 = group_flush(g);
    if (rc0) return rc0;
    for (tsdf_volume *v : g->slabs) {
        int rc = tsdf_sync(v);
        if (rc) return rc;
    }
    return TSDF_OK;
}

int tsdf_group_reset(tsdf_group *g)
{
    if (!g) return fail(TSDF_ERR_INVALID, "tsdf_group_reset: NULL handle");
    int rc0 = group_flush(g);
    if (rc0) return rc0;
    for (tsdf_volume *v : g->slabs) {
        int rc = tsdf_reset(v);
        if (rc) return rc;
    }
    return TSDF_OK;
}

int tsdf_group_download(tsdf_group *g, float *tsdf_host, float *weight_host)
{
    if (!g) return fail(TSDF_ERR_INVALID, "tsdf_group_download: NULL handle");
    int rc0 = group_flush(g);
    if (rc0) return rc0;
    const size_t slice = (size_t)g->cfg.dim_x * g->cfg.dim_y;
    return for_each_slab(g, [&](int i) -> int {
        tsdf_volume *v = g->slabs[(size_t)i];
        const size_t off = slice * (size_t)v->cfg.z_begin;
        return tsdf_download(v, tsdf_host ? tsdf_host + off : nullptr, weight_host ? weight_host + off : nullptr);
    });
}

int tsdf_group_extract_surface(tsdf_group *g, float weight_thresh, float *xyz_host, int64_t capacity, int64_t *count)
{
    if (!g || !count) return fail(TSDF_ERR_INVALID, "tsdf_group_extract_surface: NULL argument");
    return group_list(g, ListKind::Surface, weight_thresh, xyz_host, capacity, count);
}

int tsdf_group_extract_crossings(tsdf_group *g, float weight_thresh, float *xyz_host, int64_t capacity, int64_t *count)
{
    if (!g || !count) return fail(TSDF_ERR_INVALID, "tsdf_group_extract_crossings: NULL argument");
    return group_list(g, ListKind::Crossings, weight_thresh, xyz_host, capacity, count);
}

int tsdf_group_extract_mesh(tsdf_group *g, float weight_thresh, float *triangles_host, int64_t capacity, int64_t *count)
{
    if (!g || !count) return fail(TSDF_ERR_INVALID, "tsdf_group_extract_mesh: NULL argument");
    return group_list(g, ListKind::Mesh, weight_thresh, triangles_host, capacity, count);
}

int tsdf_group_save_ply(tsdf_group *g, const char *path, float weight_thresh)
{
    if (!g || !path) return fail(TSDF_ERR_INVALID, "tsdf_group_save_ply: NULL argument");
    int64_t n = 0;
    int rc = group_list(g, ListKind::Surface, weight_thresh, nullptr, 0, &n);
    if (rc) return rc;
    if (n > 0x7fffffffll)
        return fail(TSDF_ERR_INVALID, "tsdf_group_save_ply: %lld surface points exceed the format's 2^31 - 1", (long long)n);
    std::vector<float> xyz((size_t)(n > 0 ? n : 1) * 3);
    if (n > 0 && (rc = group_list(g, ListKind::Surface, weight_thresh, xyz.data(), n, &n)) != TSDF_OK) return rc;
    return write_points_ply(path, xyz.data(), n, "tsdf_group_save_ply");
}

int tsdf_group_save_mesh_ply(tsdf_group *g, const char *path, float weight_thresh)
{
    if (!g || !path) return fail(TSDF_ERR_INVALID, "tsdf_group_save_mesh_ply: NULL argument");
    int64_t n = 0;
    int rc = group_list(g, ListKind::Mesh, weight_thresh, nullptr, 0, &n);
    if (rc) return rc;
    std::vector<float> tri((size_t)(n > 0 ? n : 1) * 9);
    if (n > 0 && (rc = group_list(g, ListKind::Mesh, weight_thresh, tri.data(), n, &n)) != TSDF_OK) return rc;
    return write_mesh_ply(path, tri.data(), n, "tsdf_group_save_mesh_ply");
}

int tsdf_group_save_bin(tsdf_group *g, const char *path)
{
    if (!g || !path) return fail(TSDF_ERR_INVALID, "tsdf_group_save_bin: NULL argument");
    int rc = group_flush(g);
    if (rc) return rc;
    const tsdf_config &c = g->cfg;
    FILE *fp = std::fopen(path, "wb");
    if (!fp) return fail(TSDF_ERR_IO, "tsdf_group_save_bin: cannot open %s", path);
    const float hdr[8] = {(float)c.dim_x, (float)c.dim_y, (float)c.dim_z, c.origin[0], c.origin[1], c.origin[2], c.voxel_size, c.trunc_margin};
    const bool ok = std::fwrite(hdr, sizeof(float), 8, fp) == 8;
    // the slabs in z order, each streamed from its own device (the file is the whole grid's: ref src/tsdf.cu:118-131)
    for (size_t i = 0; ok && rc == TSDF_OK && i < g->slabs.size(); ++i) {
        tsdf_volume *v = g->slabs[i];
        if ((rc = bind_device(v)) != TSDF_OK) break;
        rc = stream_device_to_file(v, fp, v->d_tsdf, (size_t)v->n_vox * sizeof(float), "tsdf_group_save_bin", path);
    }
    const int bad = std::fclose(fp);
    if (rc) return rc;
    if (!ok || bad) return fail(TSDF_ERR_IO, "tsdf_group_save_bin: short write to %s", path);
    return TSDF_OK;
}

}  // extern "C"
