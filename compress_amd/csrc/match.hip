// match.hip - the per-position longest-back-reference search (gfx950).
//
// Replaces, for EVERY position at once, the bestBackref closure of lzss/compress.go:146-173:
// findBackRef (compress.go:435-470) over Index.LookupLongest
// (lzss/internal/suffixarray/suffixarray.go:59-152) for the short window [at-2^14, at), the long
// window [at-2^21, at) and the dictionary.
//
// LookupLongest in closed form (cross-checked against the literal transcription in
// tests/test_oracle.py): the result length is the largest L in [minLen, maxLen] such that s[:L]
// occurs at an in-range offset, and the result offset is the in-range occurrence whose suffix has
// the SMALLEST RANK in the suffix array of the indexed text.
//
// Search in rank order, one lane per rank r (position i = SA[r]): the suffixes sharing the longest
// prefix with suffix i are its neighbours in rank, and lcp(i, SA[k]) is the running minimum of the
// LCP array between r and k.  So
//   1. walk up and down from r to the nearest in-window neighbour on each side (Lup, Ldn);
//   2. L = max(Lup, Ldn); if only the upper side attains L, the nearest upper neighbour IS the
//      smallest-rank occurrence (every rank between is out of the window);
//   3. otherwise the occurrences of s[:L] are the rank interval [lo, ..] with lo = the previous rank
//      whose LCP drops below L (two-level block minima make that O(1)-ish), and the answer is the
//      first in-window entry from lo upward.
// Positions below HEAD are done by brute force over their (tiny) window instead, because the expected
// walk length from r to an in-window neighbour grows like n / i.
#include "kernels.h"

namespace lzss {

constexpr int LCP_THREADS = 256;
constexpr int MATCH_THREADS = 256;

// ---- LCP array and its block minima -------------------------------------------------------------
__global__ __launch_bounds__(LCP_THREADS) void lcp_kernel(const Seg *segs, BatchIndex ix) {
    const Seg sg = segs[blockIdx.y];
    const uint32_t N = sg.n + 1;
    if (blockIdx.x * LCP_THREADS >= N) return;
    const uint32_t k = blockIdx.x * LCP_THREADS + threadIdx.x;
    const uint8_t *T = ix.text + sg.off;
    const int32_t *SA = ix.sa + sg.off;
    uint32_t l = 0xFFFFu;
    if (k < N) {
        l = 0;
        if (k > 0) {
            const uint32_t a = (uint32_t)SA[k - 1], b = (uint32_t)SA[k];
            const uint32_t hi = a > b ? a : b;
            int lim = (int)(sg.n - hi);  // the shorter suffix ends at the sentinel
            if (lim > LZSS_MAX_LEN) lim = LZSS_MAX_LEN;
            l = (uint32_t)lcp_bytes(T + a, T + b, 0, lim);
        }
        ix.lcp[sg.off + k] = (uint16_t)l;
    }
    // minimum over the 64 ranks of this wave
    uint32_t m = l;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t t = (uint32_t)__shfl_xor((int)m, o);
        m = t < m ? t : m;
    }
    if ((threadIdx.x & 63) == 0) ix.bmin1[sg.off / 64 + (k >> 6)] = (uint16_t)m;
}

__global__ __launch_bounds__(64) void bmin2_kernel(const Seg *segs, BatchIndex ix) {
    const Seg sg = segs[blockIdx.y];
    const uint32_t N = sg.n + 1;
    const uint32_t nb1 = (N + 63) / 64;
    const uint32_t c = blockIdx.x;
    if (c * 64 >= nb1) return;
    const uint32_t b = c * 64 + threadIdx.x;
    uint32_t m = b < nb1 ? (uint32_t)ix.bmin1[sg.off / 64 + b] : 0xFFFFu;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        uint32_t t = (uint32_t)__shfl_xor((int)m, o);
        m = t < m ? t : m;
    }
    if (threadIdx.x == 0) ix.bmin2[sg.off / 64 + c] = (uint16_t)m;
}

void launch_lcp(const Seg *d_segs, uint32_t n_segs, uint32_t max_n, const BatchIndex &ix, hipStream_t stream) {
    if (n_segs == 0) return;
    const uint32_t N = max_n + 1;
    dim3 g1((N + LCP_THREADS - 1) / LCP_THREADS, n_segs);
    hipLaunchKernelGGL(lcp_kernel, g1, dim3(LCP_THREADS), 0, stream, d_segs, ix);
    const uint32_t nb1 = (N + 63) / 64;
    dim3 g2((nb1 + 63) / 64, n_segs);
    hipLaunchKernelGGL(bmin2_kernel, g2, dim3(64), 0, stream, d_segs, ix);
}

// ---- dictionary lookup (compress.go:463: LookupLongest over [0, dictLen)) -----------------------
// Longest prefix of pat[0:maxLen) occurring in the dictionary, and its smallest-rank occurrence.
__device__ inline void dict_lookup(const DictView &D, const uint8_t *pat, int maxLen, int &Ld, int &pd) {
    // invariant: suffix(lo) < pat <= suffix(hi); rank 0 is the sentinel, rank n+1 is virtual
    int lo = 0, hi = (int)D.n + 1;
    int llo = 0, lhi = 0;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        const int q = D.SA[mid];
        const int avail = (int)D.n - q;
        const int lim = maxLen < avail ? maxLen : avail;
        int o = llo < lhi ? llo : lhi;
        o = lcp_bytes(pat, D.T + q, o, lim);
        bool suffix_less;
        if (o == maxLen) suffix_less = false;          // pat is a prefix of the suffix
        else if (o == avail) suffix_less = true;       // the suffix is a proper prefix of pat
        else suffix_less = D.T[q + o] < pat[o];
        if (suffix_less) { lo = mid; llo = o; } else { hi = mid; lhi = o; }
    }
    Ld = llo > lhi ? llo : lhi;
    pd = -1;
    if (Ld <= 0) return;
    if (lhi > llo) {
        pd = D.SA[hi];  // nothing below hi shares Ld bytes
    } else {
        const int lo2 = prev_lower(D.LCP, D.bmin1, D.bmin2, lo, Ld);
        pd = D.SA[lo2];
    }
}

// Combine the data-window results with the dictionary and apply the selection rule of
// compress.go:160-169.  Ls/ps: short window, LL/pL: long window (0 / -1 = none).
__device__ inline uint32_t select_backref(const DictView &D, const uint8_t *T, int i, int maxLen, int minD, bool dynOK,
                                          int Ls, int ps, int LL, int pL) {
    int Ldyn = 0;
    uint32_t fdyn = 0;
    if (dynOK) {
        if (LL > 0) {
            Ldyn = LL;
            fdyn = (uint32_t)(i - pL - 1);
        }
        if (LL < maxLen) {  // compress.go:460: the dictionary is consulted only then
            int Ld, pd;
            dict_lookup(D, T + i, maxLen, Ld, pd);
            if (Ld >= minD && Ld > LL) {  // compress.go:464: strictly longer
                Ldyn = Ld;
                fdyn = (uint32_t)(i + (int)D.n - pd - 1);  // backref.go:66, masked to 21 bits by res_pack
            }
        }
    }
    // short wins iff found and 8*Ls-30 > 8*Ldyn-37, i.e. Ls >= Ldyn (compress.go:165)
    if (Ls > 0 && Ls >= Ldyn) return res_pack(RES_SHORT, (uint32_t)Ls, (uint32_t)(i - ps - 1));
    if (Ldyn > 0) return res_pack(RES_DYN, (uint32_t)Ldyn, fdyn);
    return res_pack(RES_NONE, 0, 0);
}

// first in-window entry of the rank interval of suffixes sharing >= L bytes with rank kdn, scanning
// upward from the interval's first rank; kdn itself is in the window, so the scan ends there at the latest
__device__ inline int leftmost_in_window(const int32_t *SA, const uint16_t *LCP, const uint16_t *b1, const uint16_t *b2,
                                         int kdn, int L, int ws, int i) {
    int k = prev_lower(LCP, b1, b2, kdn, L);
    for (; k < kdn; ++k) {
        const int p = SA[k];
        if (p >= ws && p < i) return p;
    }
    return SA[kdn];
}

__global__ __launch_bounds__(MATCH_THREADS) void match_kernel(const Seg *segs, BatchIndex ix, DictView D, uint32_t *res) {
    const Seg sg = segs[blockIdx.y];
    const int n = (int)sg.n;
    const int N = n + 1;
    if ((int)(blockIdx.x * MATCH_THREADS) >= N) return;
    const int r = (int)(blockIdx.x * MATCH_THREADS + threadIdx.x);
    if (r < 1 || r >= N) return;
    const uint8_t *T = ix.text + sg.off;
    const int32_t *SA = ix.sa + sg.off;
    const uint16_t *LCP = ix.lcp + sg.off;
    const uint16_t *b1 = ix.bmin1 + sg.off / 64;
    const uint16_t *b2 = ix.bmin2 + sg.off / 64;

    const int i = SA[r];
    if (i < (int)sg.start || i < HEAD) return;

    const int maxLen = (n - i) < LZSS_MAX_LEN ? (n - i) : LZSS_MAX_LEN;
    const bool reserved = T[i] >= LZSS_SYMBOL_SHORT;
    const int minS = reserved ? 1 : LZSS_SHORT_MIN_LEN;
    const int minD = reserved ? 1 : LZSS_DYN_MIN_LEN;
    const bool shortOK = maxLen >= minS, dynOK = maxLen >= minD;
    const int wsS = i > LZSS_SHORT_WINDOW ? i - LZSS_SHORT_WINDOW : 0;
    const int wsL = i > LZSS_DYN_WINDOW ? i - LZSS_DYN_WINDOW : 0;

    int LupS = 0, pupS = -1, LupL = 0, pupL = -1;
    int LdnS = 0, kdnS = -1, LdnL = 0, kdnL = -1;
    if (shortOK) {
        // nearest in-window neighbours above r
        int m = 1 << 20;
        bool fL = false;
        for (int k = r + 1; k < N; ++k) {
            const int l = LCP[k];
            m = l < m ? l : m;
            if (m < minS) break;
            const int p = SA[k];
            if (p < i) {
                if (!fL && p >= wsL) { fL = true; LupL = m; pupL = p; }
                if (p >= wsS) { LupS = m; pupS = p; break; }
            }
        }
        // nearest in-window neighbours below r (rank 0 is the sentinel: lcp 0)
        m = 1 << 20;
        fL = false;
        for (int k = r - 1; k >= 1; --k) {
            const int l = LCP[k + 1];
            m = l < m ? l : m;
            if (m < minS) break;
            const int p = SA[k];
            if (p < i) {
                if (!fL && p >= wsL) { fL = true; LdnL = m; kdnL = k; }
                if (p >= wsS) { LdnS = m; kdnS = k; break; }
            }
        }
    }
    // short window
    int Ls = 0, ps = -1;
    {
        const int ld = LdnS >= minS ? LdnS : 0, lu = LupS >= minS ? LupS : 0;
        Ls = ld > lu ? ld : lu;
        if (Ls > 0) ps = (lu > ld) ? pupS : leftmost_in_window(SA, LCP, b1, b2, kdnS, Ls, wsS, i);
    }
    // long window
    int LL = 0, pL = -1;
    if (dynOK) {
        const int ld = LdnL >= minD ? LdnL : 0, lu = LupL >= minD ? LupL : 0;
        LL = ld > lu ? ld : lu;
        if (LL > 0) pL = (lu > ld) ? pupL : leftmost_in_window(SA, LCP, b1, b2, kdnL, LL, wsL, i);
    }
    res[sg.off + i] = select_backref(D, T, i, maxLen, minD, dynOK, Ls, ps, LL, pL);
}

// positions i < HEAD: brute force over the window [0, i) (identical for both window sizes there)
__global__ __launch_bounds__(64) void match_head_kernel(const Seg *segs, BatchIndex ix, DictView D, uint32_t *res) {
    const Seg sg = segs[blockIdx.y];
    const int n = (int)sg.n;
    const int i = (int)(blockIdx.x * 64 + threadIdx.x);
    if (i >= n || i >= HEAD || i < (int)sg.start) return;
    const uint8_t *T = ix.text + sg.off;
    const int32_t *ISA = ix.isa + sg.off;

    const int maxLen = (n - i) < LZSS_MAX_LEN ? (n - i) : LZSS_MAX_LEN;
    const bool reserved = T[i] >= LZSS_SYMBOL_SHORT;
    const int minS = reserved ? 1 : LZSS_SHORT_MIN_LEN;
    const int minD = reserved ? 1 : LZSS_DYN_MIN_LEN;
    const bool shortOK = maxLen >= minS, dynOK = maxLen >= minD;

    int best = 0, bp = -1, brank = 0x7FFFFFFF;
    if (shortOK) {
        for (int p = 0; p < i; ++p) {
            const int l = lcp_bytes(T + p, T + i, 0, maxLen);
            if (l < minS || l < best) continue;
            const int rk = ISA[p];
            if (l > best || rk < brank) { best = l; bp = p; brank = rk; }
        }
    }
    const int Ls = best, ps = bp;
    const int LL = (dynOK && best >= minD) ? best : 0, pL = bp;
    res[sg.off + i] = select_backref(D, T, i, maxLen, minD, dynOK, Ls, ps, LL, pL);
}

void launch_match(const Seg *d_segs, uint32_t n_segs, uint32_t max_n, const BatchIndex &ix, const DictView &dict,
                  uint32_t *d_res, hipStream_t stream) {
    if (n_segs == 0 || max_n == 0) return;
    const uint32_t head = max_n < (uint32_t)HEAD ? max_n : (uint32_t)HEAD;
    dim3 gh((head + 63) / 64, n_segs);
    hipLaunchKernelGGL(match_head_kernel, gh, dim3(64), 0, stream, d_segs, ix, dict, d_res);
    if (max_n > (uint32_t)HEAD) {
        dim3 g((max_n + 1 + MATCH_THREADS - 1) / MATCH_THREADS, n_segs);
        hipLaunchKernelGGL(match_kernel, g, dim3(MATCH_THREADS), 0, stream, d_segs, ix, dict, d_res);
    }
}

}  // namespace lzss
