// jsnoop_bytes.h -- marker searches of the staging code (host side), sixteen bytes per step.
// Every byte of every file passes through one of them when it is added to a batch (the pass the reference's SOS handler makes over the
// entropy-coded segment, source/JfifDecode.cpp:5207-5265); a byte-at-a-time loop took 0.6 of the 1.9 ms a call on a 2.2 MB file cost.
#pragma once
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <stddef.h>
#include <stdint.h>

// first index >= q with f[index] == FF, or n
static inline size_t js_next_ff(const uint8_t* f, size_t q, size_t n)
{
#if defined(__SSE2__)                                             // (hosts without SSE2 -- aarch64 nodes -- keep the byte loop)
    const __m128i ff = _mm_set1_epi8((char)0xFF);
    while (q + 16 <= n) {
        const unsigned m = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(f + q)), ff));
        if (m) return q + (unsigned)__builtin_ctz(m);
        q += 16;
    }
#endif
    while (q < n && f[q] != 0xFF) q++;
    return q;
}
// First offset q >= start with f[q] == FF and f[q + 1] neither 00 nor RSTn (q + 1 < len), else len: where the entropy-coded data ends.
static inline uint32_t js_scan_end(const uint8_t* f, uint32_t q, size_t len)
{
#if defined(__SSE2__)
    const __m128i ff = _mm_set1_epi8((char)0xFF);
#endif
    while ((size_t)q + 1 < len) {
#if defined(__SSE2__)
        if ((size_t)q + 17 <= len) {
            unsigned m = (unsigned)_mm_movemask_epi8(_mm_cmpeq_epi8(_mm_loadu_si128(reinterpret_cast<const __m128i*>(f + q)), ff));
            while (m) { const unsigned j = (unsigned)__builtin_ctz(m); m &= m - 1; const uint8_t nx = f[q + j + 1]; if (nx != 0 && !(nx >= 0xD0 && nx <= 0xD7)) return q + j; }
            q += 16; continue;
        }
#endif
        if (f[q] == 0xFF && f[q + 1] != 0 && !(f[q + 1] >= 0xD0 && f[q + 1] <= 0xD7)) return q;
        q++;
    }
    return (uint32_t)len;
}
