// frame_store.h -- staging and deferral memory shared by all handles of one device and image size.
//
// The reference keeps one TSDF per object instance and creates them by the dozen (ref: src/Engine.cpp:172-233,
// src/Object.cpp:67).  Round 2 gave every handle its own pinned ring (3 x 1.2 MB + 3 x 1.2 MB in HBM at creation), its own two
// 32-frame deferral pools (78.6 MB on the first deferred call, + 19.7 MB of mask pools) and its own depth tile tables (9.4 MB on
// the first classified launch): more than a 200^3 volume (64 MB) itself.  Now one FrameStore per (device, image size), shared and
// reference-counted, holds
//   * a ring of pinned host frames (the caller's buffer is copied there so that it may be freed when the call returns),
//   * frame slots in HBM (one depth image each): the frames a handle has collected but not yet launched, and the staging
//     of a one-kernel-per-call frame,
//   * mask slots (one instance mask each) and table slots (the depth tile tables of one fused launch),
// handed out per frame and taken back by stream order, never by a host wait: a slot released "after event E" may be handed out
// again at once -- its next user's stream is made to wait for E (hipStreamWaitEvent) before it writes.  A handle may be driven
// from its own thread (include/tsdf_hip.h), so the store is mutex-guarded; a handle only ever flushes ITSELF to make room.
// 64 handles of 640 x 480 frames cost their volumes plus one store (a few frame slots each while they collect, bounded by
// kSoftCapFrames: beyond it a handle that asks for a slot applies what it has collected first).
#pragma once
#include <hip/hip_runtime.h>

#include <mutex>
#include <thread>
#include <vector>

namespace tsdf_store {

constexpr int kRingSlots = 4;
constexpr int kSoftCapFrames = 160;    // frame slots (197 MB at 640 x 480) before collecting handles are asked to flush themselves
constexpr int kSoftCapTables = 16;     // table slots (9.2 MB each at 640 x 480): launches of that many handles may be in flight before one waits for another's

struct Slot {
    void *dev;
    hipEvent_t *release;     // state 2: free again once this event (owned by the handle that used the slot) has happened
    const void *owner;       // the handle holding it (state 1) or whose event it waits for (state 2)
    int state;               // 0 free, 1 held, 2 in flight
    unsigned long long seq;  // state 2: when it was released (the class's counter): the oldest is reused first
};

struct RingSlot {
    float *host;             // pinned, one float frame (also holds a 16-bit frame)
    hipEvent_t copied;       // the last copy out of it
    bool used, busy;
};

struct SlotClass {
    std::vector<Slot> slots;
    size_t bytes;
    int soft_cap;
    int quota;               // slots one handle may take before it reuses its own (stream-ordered) instead of growing the class
    unsigned long long seq;
};

struct FrameStore {
    int device;
    size_t px;               // pixels per frame
    int refs;
    std::mutex mu;
    RingSlot ring[kRingSlots];
    int ring_next;
    SlotClass frames, masks, tables;
    hipStream_t copy_stream;   // host -> device copies of every handle of the store (one PCIe pipe; a stream per handle cost ~2 MiB each)
};

inline std::mutex &registry_mutex() { static std::mutex m; return m; }
inline std::vector<FrameStore *> &registry() { static std::vector<FrameStore *> r; return r; }

// Get (or create) the store of this device and image size; table_bytes = the tile tables of one fused launch.
inline hipError_t store_ref(int device, size_t px, size_t table_bytes, FrameStore **out)
{
    std::lock_guard<std::mutex> lk(registry_mutex());
    for (FrameStore *s : registry())
        if (s->device == device && s->px == px && s->tables.bytes == table_bytes) { ++s->refs; *out = s; return hipSuccess; }
    FrameStore *s = new FrameStore();
    s->device = device; s->px = px; s->refs = 1; s->ring_next = 0;
    s->frames.seq = s->masks.seq = s->tables.seq = 0;
    s->frames.bytes = px * sizeof(float); s->frames.soft_cap = kSoftCapFrames; s->frames.quota = 64;
    s->masks.bytes = px; s->masks.soft_cap = kSoftCapFrames; s->masks.quota = 64;
    s->tables.bytes = table_bytes; s->tables.soft_cap = kSoftCapTables; s->tables.quota = 2;
    s->copy_stream = nullptr;
    for (int i = 0; i < kRingSlots; ++i) { s->ring[i].host = nullptr; s->ring[i].copied = nullptr; s->ring[i].used = s->ring[i].busy = false; }
    (void)hipSetDevice(device);
    const hipError_t e = hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete s; return e; }
    registry().push_back(s);
    *out = s;
    return hipSuccess;
}

inline void store_unref(FrameStore *s)
{
    if (!s) return;
    std::lock_guard<std::mutex> lk(registry_mutex());
    if (--s->refs > 0) return;
    (void)hipSetDevice(s->device);
    if (s->copy_stream) { (void)hipStreamSynchronize(s->copy_stream); (void)hipStreamDestroy(s->copy_stream); }
    for (int i = 0; i < kRingSlots; ++i) {
        if (s->ring[i].copied) { (void)hipEventSynchronize(s->ring[i].copied); (void)hipEventDestroy(s->ring[i].copied); }
        if (s->ring[i].host) (void)hipHostFree(s->ring[i].host);
    }
    for (SlotClass *c : {&s->frames, &s->masks, &s->tables})
        for (Slot &x : c->slots) if (x.dev) (void)hipFree(x.dev);
    auto &r = registry();
    for (size_t i = 0; i < r.size(); ++i) if (r[i] == s) { r.erase(r.begin() + (long)i); break; }
    delete s;
}

// A pinned frame of the ring, ready to be written by the host (its previous copy has run).  Release with ring_release.
inline hipError_t ring_acquire(FrameStore *s, int *index)
{
    int i = -1;
    for (;;) {
        {
            std::lock_guard<std::mutex> lk(s->mu);
            for (int k = 0; k < kRingSlots; ++k) {
                const int j = (s->ring_next + k) % kRingSlots;
                if (!s->ring[j].busy) { i = j; s->ring[j].busy = true; s->ring_next = (j + 1) % kRingSlots; break; }
            }
        }
        if (i >= 0) break;
        std::this_thread::yield();      // more threads than ring slots: someone is in the middle of a memcpy
    }
    RingSlot &r = s->ring[i];
    hipError_t e = hipSuccess;
    if (!r.host) {
        e = hipHostMalloc((void **)&r.host, s->px * sizeof(float), hipHostMallocPortable);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&r.copied, hipEventDisableTiming);
    }
    if (e == hipSuccess && r.used) e = hipEventSynchronize(r.copied);
    if (e != hipSuccess) { std::lock_guard<std::mutex> lk(s->mu); r.busy = false; return e; }
    *index = i;
    return hipSuccess;
}

// The copies out of ring slot i have been queued on `stream`: the slot is free for the next writer once they have run.
inline hipError_t ring_release(FrameStore *s, int i, hipStream_t stream)
{
    const hipError_t e = hipEventRecord(s->ring[i].copied, stream);
    std::lock_guard<std::mutex> lk(s->mu);
    s->ring[i].used = true;
    s->ring[i].busy = false;
    return e;
}

// A slot of class c for `owner`, whose first access will be queued on `first_user`.  In order of preference: a free slot; once
// the handle has its quota (two passes' worth: one being filled while the other is read), the slot it released longest ago --
// its stream waits for its own earlier launch; a new slot while the class is below its soft cap (rather than make one handle's
// launch wait for another's); the slot released longest ago by anyone -- the next user's stream waits for that reader, the host
// never does (and no hipEventQuery: it costs microseconds per call on this runtime).  *index = -1 with hipSuccess: at the cap
// with every slot HELD by handles that are still collecting -- the caller should apply its own collected frames and ask again
// with force = true.
inline hipError_t slot_acquire(FrameStore *s, SlotClass *c, const void *owner, hipStream_t first_user, bool force, int *index, void **dev)
{
    std::lock_guard<std::mutex> lk(s->mu);
    *index = -1;
    int pick = -1, mine = -1, oldest = -1, own = 0;
    for (size_t i = 0; i < c->slots.size(); ++i) {
        const Slot &x = c->slots[i];
        if (x.state == 0) { if (pick < 0) pick = (int)i; continue; }
        if (x.owner == owner) ++own;
        if (x.state != 2) continue;
        if (x.owner == owner && (mine < 0 || x.seq < c->slots[(size_t)mine].seq)) mine = (int)i;
        if (oldest < 0 || x.seq < c->slots[(size_t)oldest].seq) oldest = (int)i;
    }
    const bool at_cap = (int)c->slots.size() >= c->soft_cap;
    int wait_for = -1;
    if (pick < 0 && mine >= 0 && (own >= c->quota || at_cap)) wait_for = mine;
    else if (pick < 0 && at_cap && oldest >= 0) wait_for = oldest;
    if (wait_for >= 0) {
        const hipError_t e = hipStreamWaitEvent(first_user, *c->slots[(size_t)wait_for].release, 0);
        if (e != hipSuccess) return e;
        pick = wait_for;
    }
    if (pick < 0) {
        if (at_cap && !force) return hipSuccess;
        Slot x;
        x.dev = nullptr; x.release = nullptr; x.owner = nullptr; x.state = 0; x.seq = 0;
        const hipError_t e = hipMalloc(&x.dev, c->bytes ? c->bytes : 1);
        if (e != hipSuccess) return e;
        c->slots.push_back(x);
        pick = (int)c->slots.size() - 1;
    }
    Slot &x = c->slots[(size_t)pick];
    x.state = 1; x.owner = owner; x.release = nullptr;
    *index = pick;
    *dev = x.dev;
    return hipSuccess;
}

// The slots' last reader has been queued: they are free again once *release (an event of `owner`, recorded after that reader)
// has happened.
inline void slots_release_after(FrameStore *s, SlotClass *c, const int *idx, int n, hipEvent_t *release, const void *owner)
{
    std::lock_guard<std::mutex> lk(s->mu);
    for (int k = 0; k < n; ++k) {
        if (idx[k] < 0) continue;
        Slot &x = c->slots[(size_t)idx[k]];
        x.state = 2; x.release = release; x.owner = owner; x.seq = ++c->seq;
    }
}

// `owner` goes away (its streams have been synchronised): everything it holds or that waits for one of its events is free.
inline void slots_drop_owner(FrameStore *s, const void *owner)
{
    std::lock_guard<std::mutex> lk(s->mu);
    for (SlotClass *c : {&s->frames, &s->masks, &s->tables})
        for (Slot &x : c->slots)
            if (x.state != 0 && x.owner == owner) { x.state = 0; x.release = nullptr; x.owner = nullptr; }
}

// bytes of device memory the store holds (diagnostics / tests)
inline size_t store_device_bytes(FrameStore *s)
{
    std::lock_guard<std::mutex> lk(s->mu);
    size_t b = 0;
    for (SlotClass *c : {&s->frames, &s->masks, &s->tables}) b += c->slots.size() * c->bytes;
    return b;
}

}  // namespace tsdf_store
