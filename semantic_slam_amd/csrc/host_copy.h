// host_copy.h -- the caller's frame into the pinned ring (TSDF::Integrate's host pointer, ref: src/tsdf.cu:161-162).
// The reference's only call shape hands over 1.2 MB of pageable memory per frame; the library copies it into a pinned ring slot so
// that the caller may free its buffer when the call returns, and that copy -- one host thread, 1.2 MB -- is what bounds the
// deferred host-pointer path (round 3: 0.051 ms per call on the realistic scene against 0.0255 from HBM).  memcpy stores through
// the cache (every destination line is first read for ownership: a frame of this size stays below glibc's non-temporal
// threshold); the pinned slot is written once and next read by the DMA engine, so streaming stores are the right kind:
// 32-byte non-temporal stores (AVX2), no read of the destination, nothing of the ring kept in the cache.
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace tsdf_host {

#if defined(__x86_64__)
__attribute__((target("avx2"))) inline void copy_streaming_avx2(void *dst, const void *src, size_t n)
{
    unsigned char *d = static_cast<unsigned char *>(dst);
    const unsigned char *s = static_cast<const unsigned char *>(src);
    // head: up to the destination's next 32-byte boundary (pinned slots are page-aligned: normally nothing)
    size_t head = (32 - (reinterpret_cast<uintptr_t>(d) & 31)) & 31;
    if (head > n) head = n;
    if (head) { std::memcpy(d, s, head); d += head; s += head; n -= head; }
    size_t i = 0;
    for (; i + 128 <= n; i += 128) {
        const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + i));
        const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + i + 32));
        const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + i + 64));
        const __m256i e = _mm256_loadu_si256(reinterpret_cast<const __m256i *>(s + i + 96));
        _mm256_stream_si256(reinterpret_cast<__m256i *>(d + i), a);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(d + i + 32), b);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(d + i + 64), c);
        _mm256_stream_si256(reinterpret_cast<__m256i *>(d + i + 96), e);
    }
    _mm_sfence();      // the streamed lines are globally visible before the DMA is queued
    if (i < n) std::memcpy(d + i, s + i, n - i);
}
#endif

// dst: a pinned ring slot; src: the caller's buffer (any alignment).
inline void copy_to_pinned(void *dst, const void *src, size_t n)
{
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
#ifdef TSDF_EXPERIMENTS
    static const bool plain = std::getenv("TSDF_PLAIN_MEMCPY") != nullptr;      // A/B knob of the measurement build
    if (plain) { std::memcpy(dst, src, n); return; }
#endif
    if (avx2 && n >= 65536) { copy_streaming_avx2(dst, src, n); return; }
#endif
    std::memcpy(dst, src, n);
}

}  // namespace tsdf_host
