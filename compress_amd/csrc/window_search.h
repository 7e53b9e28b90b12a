// window_search.h - the smallest-rank occurrence of a match inside a window (gfx950, device code).
//
// Second half of Index.LookupLongest (lzss/internal/suffixarray/suffixarray.go:59-152) in closed form: the
// reference's binary search ends on the occurrence whose suffix has the smallest rank among those that start
// inside the window, and that occurrence fixes the address bits of the back-reference (backref.go:66).  The match
// LENGTH does not depend on it, and the phrase walk (compress.go:176-257) reads lengths only - so match_kernel may
// leave this search to emit_kernel, which runs it for the phrases that are actually written.
#pragma once
#include "kernels.h"

#ifndef MP_ARG  // section clocks of the MATCH_PROFILE build of match.hip; nothing elsewhere
#define MP_ARG
#define MP(i)
#endif

namespace lzss {

// first j in [j0, end) with P[j] < x (LESS) or P[j] >= x (!LESS); `end` if there is none.  m1 / m2 hold the
// min (LESS) or max (!LESS) of P over blocks of 64 and 4096 entries, so whole blocks without a hit are
// skipped: at most ~64 steps per level whatever the distance.
// mask: the bits of an entry of P that hold the value (all for plain arrays; the summaries are over masked values)
template <bool LESS>
__device__ inline int next_thr(const int32_t *P, const int32_t *m1, const int32_t *m2, int j0, int end, int x, int mask = -1) {
    int j = j0;
    // rest of the current 64-block, four entries per load once j is a multiple of 4
    while (j < end && (j & 3)) {
        const int v = P[j] & mask;
        if (LESS ? v < x : v >= x) return j;
        ++j;
    }
    while (j < end && (j & 63)) {
        const I4 q = ld_i4(P + j);  // entries past `end` belong to the padded allocation and are ignored
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (j + t < end && (LESS ? (q.v[t] & mask) < x : (q.v[t] & mask) >= x)) return j + t;
        j += 4;
    }
    if (j >= end) return end;
    int b = j >> 6;
    const int bend = (end + 63) >> 6;
    while (b < bend) {
        if ((b & 63) == 0) {
            const int v2 = m2[b >> 6];
            if (LESS ? v2 >= x : v2 < x) { b += 64; continue; }
        }
        if ((b & 3) == 0 && b + 4 <= bend) {  // four block summaries per load
            const I4 q = ld_i4(m1 + b);
            int t = 0;
            for (; t < 4; ++t)
                if (LESS ? q.v[t] < x : q.v[t] >= x) break;
            b += t;
            if (t < 4) break;
            continue;
        }
        const int v1 = m1[b];
        if (LESS ? v1 < x : v1 >= x) break;
        ++b;
    }
    if (b >= bend) return end;
    for (j = b << 6; j < end; j += 4) {
        const I4 q = ld_i4(P + j);
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (j + t < end && (LESS ? (q.v[t] & mask) < x : (q.v[t] & mask) >= x)) return j + t;
    }
    return end;
}

struct SegIndex {  // one segment's view of BatchIndex
    const int32_t *SA;
    const uint16_t *LCP, *b1, *b2;
    const int32_t *KS, *PS, *pmin1, *pmax1, *pmin2, *pmax2, *samin1, *samin2;
    const uint16_t *cdir;
    int nch;
    int n;
    int wsh;  // 14 when the window index is packed (an entry of KS = rank << 14 | position inside the chunk; PS == KS), else 0
};

#ifndef MATCH_LINEAR_SPAN
#define MATCH_LINEAR_SPAN 24
#endif
constexpr int LINEAR_SPAN = MATCH_LINEAR_SPAN;  // intervals up to this many ranks are scanned directly

// first entry j in [a, end) of a chunk's rank-ordered list with KS[j] >= lo, a coming from the chunk directory
// (so at most 64 entries away, a chunk's share of 64 ranks)
__device__ inline int chunk_lower_bound(const int32_t *KS, int a, int end, int lo, int sh) {
    while (a < end && (a & 3)) {
        if ((int)((uint32_t)KS[a] >> sh) >= lo) return a;
        ++a;
    }
    while (a < end) {
        const I4 q = ld_i4(KS + a);  // entries past `end` belong to the padded allocation and are ignored
#pragma unroll
        for (int t = 0; t < 4; ++t)
            if (a + t >= end || (int)((uint32_t)q.v[t] >> sh) >= lo) return (a + t < end) ? a + t : end;
        a += 4;
    }
    return end;
}

// Smallest-rank occurrence inside the window [ws, i) among the suffixes sharing >= L bytes with rank kdn (which is
// itself in the window, so the answer is <= kdn).  SHORT window: through the window index (two chunks, one-sided
// condition in each).  LONG window: ws == 0 whenever i <= 2^21, and then "pos < i" over the suffix array itself is
// one-sided.  A lane needs this for ONE of its two windows only (the one whose reference is emitted), so both share
// the interval search.
__device__ inline int leftmost(const SegIndex &S, int kdn, int L, int ws, int i, bool is_short MP_ARG) {
    const int lo = prev_lower(S.LCP, S.b1, S.b2, kdn, L);
    MP(8)
    int res = -1;
    if (kdn - lo <= LINEAR_SPAN || (!is_short && ws != 0)) {
        if (kdn - lo <= LINEAR_SPAN) {
            // the whole interval in one round trip: all loads are issued before the first is looked at (rank kdn is
            // inside the window, so the scan ends there at the latest)
            constexpr int NQ = LINEAR_SPAN / 4 + 1;
            I4 q[NQ];
#pragma unroll
            for (int t = 0; t < NQ; ++t)
                if (lo + 4 * t <= kdn) q[t] = ld_i4(S.SA + lo + 4 * t);  // entries past kdn are ignored
#pragma unroll
            for (int t = 0; t < NQ; ++t)
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int p = q[t].v[u];
                    if (res < 0 && lo + 4 * t + u <= kdn && p >= ws && p < i) res = p;
                }
        } else {
            for (int k = lo; k < kdn && res < 0; ++k) {
                const int p = S.SA[k];
                if (p >= ws && p < i) res = p;
            }
            if (res < 0) res = S.SA[kdn];
        }
        MP(9)
    } else if (is_short) {
        int best = kdn;
        const int c2 = (i - 1) >> LZSS_SHORT_ADDR_BITS;  // chunk of the window's last position: entries with pos < i
        const int lo1 = lo < 1 ? 1 : lo;                 // rank 0 is the sentinel, in no chunk
        const uint16_t *dir = S.cdir + (size_t)((lo1 - 1) >> 6) * S.nch;
        const int pm = S.wsh ? (1 << LZSS_SHORT_ADDR_BITS) - 1 : -1;  // packed: positions are relative to their chunk
        {
            const int beg = c2 << LZSS_SHORT_ADDR_BITS;
            const int end = (beg + LZSS_SHORT_WINDOW) < S.n ? (beg + LZSS_SHORT_WINDOW) : S.n;
            const int a = chunk_lower_bound(S.KS, beg + (int)dir[c2], end, lo1, S.wsh);
            const int j = next_thr<true>(S.PS, S.pmin1, S.pmin2, a, end, S.wsh ? i - beg : i, pm);
            if (j < end) { const int k = (int)((uint32_t)S.KS[j] >> S.wsh); best = k < best ? k : best; }
        }
        if (c2 >= 1 && ws < (c2 << LZSS_SHORT_ADDR_BITS)) {  // previous chunk: entries with pos >= ws
            const int beg = (c2 - 1) << LZSS_SHORT_ADDR_BITS;
            const int end = beg + LZSS_SHORT_WINDOW;
            const int a = chunk_lower_bound(S.KS, beg + (int)dir[c2 - 1], end, lo1, S.wsh);
            const int j = next_thr<false>(S.PS, S.pmax1, S.pmax2, a, end, S.wsh ? ws - beg : ws, pm);
            if (j < end) { const int k = (int)((uint32_t)S.KS[j] >> S.wsh); best = k < best ? k : best; }
        }
        res = S.SA[best];
        MP(10)
    } else {
        const int k = next_thr<true>(S.SA, S.samin1, S.samin2, lo, kdn + 1, i);
        res = S.SA[k <= kdn ? k : kdn];
        MP(11)
    }
    return res;
}

// one segment's view of the batch index
__device__ inline SegIndex seg_index(const BatchIndex &ix, const Seg &sg) {
    SegIndex S;
    S.SA = ix.sa + sg.off; S.LCP = ix.lcp + sg.off;
    S.b1 = ix.bmin1 + sg.off / 64; S.b2 = ix.bmin2 + sg.off / 64;
    S.KS = ix.ks(sg); S.PS = ix.win_packed ? ix.ks(sg) : ix.ps(sg);
    S.wsh = ix.win_packed ? LZSS_SHORT_ADDR_BITS : 0;
    S.pmin1 = ix.pmin1 + sg.off / 64; S.pmax1 = ix.pmax1 + sg.off / 64;
    S.pmin2 = ix.pmin2 + sg.off / 64; S.pmax2 = ix.pmax2 + sg.off / 64;
    S.samin1 = ix.samin1 + sg.off / 64; S.samin2 = ix.samin2 + sg.off / 64;
    S.cdir = ix.cdir + (sg.off / 64) * ix.nch; S.nch = (int)ix.nch;
    S.n = (int)sg.n;
    return S;
}

}  // namespace lzss
