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
#include <atomic>
#include <cstdlib>

#include "kernels.h"

#ifdef MATCH_PROFILE  // per-section clocks summed over all waves (debug builds only)
namespace lzss { __device__ unsigned long long g_mp[16]; }
#define MP_DECL long long mp_acc[15] = {}; long long mp_last = clock64();
#define MP(i) { __builtin_amdgcn_s_waitcnt(0); const long long t_ = clock64(); mp_acc[i] += t_ - mp_last; mp_last = t_; }
#define MP_PRINT if (threadIdx.x == 0 && (blockIdx.x & 31) == 0) { for (int q_ = 0; q_ < 15; ++q_) atomicAdd(&g_mp[q_], (unsigned long long)mp_acc[q_]); atomicAdd(&g_mp[15], 1ull); }
#define MP_ARG , long long *mp_acc, long long &mp_last
#define MP_PASS , mp_acc, mp_last
#define MP_NONE , mp_none_acc, mp_none_last
#else
#define MP_DECL
#define MP(i)
#define MP_PRINT
#define MP_ARG
#define MP_PASS
#define MP_NONE
#endif
#include "window_search.h"

#ifdef DICT_STATS  // event counts of the dictionary lookups of every 64th block (debug builds only)
namespace lzss { __device__ unsigned long long g_ds[16]; }
#define DS(k, v) { if ((blockIdx.x & 63) == 0) atomicAdd(&lzss::g_ds[k], (unsigned long long)(v)); }
#else
#define DS(k, v)
#endif

namespace lzss {

constexpr int LCP_THREADS = 256;
#ifndef MATCH_BLOCK
#define MATCH_BLOCK 64
#endif
constexpr int MATCH_THREADS = MATCH_BLOCK;

// ---- dictionary lookup (compress.go:463: LookupLongest over [0, dictLen)) -----------------------
// Run-aware 8-byte key of the string s[0:A), A >= 1 (readable 8 bytes past A, zeros there): first byte c (8 bits),
// run code (9 bits), the 5 bytes after the run of c (40 bits, zero padded).  Run code: rl = length of the run of c
// the string starts with; strings whose run ends in a smaller byte (or ends the string) come first, shorter runs
// first: code rl (1..253); then runs of >= DICT_RUN_CAP: code 255, no tail; then strings whose run ends in a larger
// byte, longer runs first: code 511 - rl.  The key order is the string order (a proper prefix is smaller), and two
// different keys also give the exact common prefix length (key_lcp) - so on zero-heavy data, where a plain 8-byte
// prefix ties on every probe and every tie costs a text compare, almost every probe is decided by one load.
constexpr int DICT_RUN_CAP = 254;
// run: the length of the run of s[0] at s if the caller knows it (the text's `rlen`, capped at 255; it may exceed A), else < 0.
// w0: the first 8 bytes of s (little-endian load); a run of up to 3 leaves the whole tail inside it, so the common
// case costs no load of its own
// (ld_at(o): the 8 bytes at s + o; rl: the length of the run of s[0] at s, capped at 255, it may exceed A)
template <class LdAt>
__device__ inline uint64_t run_key_t(int A, int &rlc, int run, uint64_t w0, LdAt ld_at) {
    const uint32_t c = (uint32_t)(w0 & 0xFFull);
    const int lim = (A - 1) < DICT_RUN_CAP ? (A - 1) : DICT_RUN_CAP;
    const int rl = run < lim + 1 ? run : lim + 1;
    if (rl >= DICT_RUN_CAP) {
        rlc = DICT_RUN_CAP;
        return ((uint64_t)c << 49) | (255ull << 40);
    }
    rlc = rl;
    uint64_t tail = 0;
    bool falling = true;
    if (rl < A) {
        const int avail = A - rl;
        const uint64_t w = rl <= 3 ? (w0 >> (8 * rl)) : ld_at(rl);
        tail = (__builtin_bswap64(w) >> 24) & ((1ull << 40) - 1ull);
        if (avail < 5) tail &= ~((1ull << (8 * (5 - avail))) - 1ull);
        falling = (uint32_t)(w & 0xFFull) < c;
    }
    const uint64_t code = falling ? (uint64_t)rl : (uint64_t)(511 - rl);
    return ((uint64_t)c << 49) | (code << 40) | tail;
}
__device__ inline uint64_t run_key_w(const uint8_t *s, int A, int &rlc, int run, uint64_t w0) {
    if (run < 0) {
        const int lim = (A - 1) < DICT_RUN_CAP ? (A - 1) : DICT_RUN_CAP;
        run = 1 + lcp_bytes(s, s + 1, 0, lim);
    }
    return run_key_t(A, rlc, run, w0, [&](int o) { return ld8(s + o); });
}
// the run length a key vouches for (what run_key_t returns as rlc)
__device__ inline int key_run(uint64_t k) {
    const int code = (int)(k >> 40) & 511;
    return code == 255 ? DICT_RUN_CAP : (code < 255 ? code : 511 - code);
}
// What match_kernel needs to know about the pattern at a rank, written by the LCP kernels (which have the suffix's
// first bytes and its run length at hand, the LDS one at LDS cost) so that match_kernel reads one coalesced word per
// rank instead of gathering text and run lengths by position: the pattern's run-aware key (57 bits; its first byte
// tells a reserved symbol), and in bit 63 whether the phrase walk can never ask about the position - it, the two
// positions before it and at least LZSS_MIN_REPEAT bytes behind it lie in one run (compress.go:184: the RLE branch is
// taken at i, i - 1 and i - 2; match_kernel used to read that off the run length at i - 2).
constexpr uint64_t PKEY_SKIP = 1ull << 63, PKEY_KEY_MASK = (1ull << 57) - 1ull;
template <class LdAt>
__device__ inline uint64_t pattern_word(uint32_t n, uint32_t pos, uint64_t w0, int run, LdAt ld_at /* 8 bytes at pos + o, o >= -2 */) {
    if (pos >= n) return 0;  // the sentinel
    const int A = (int)(n - pos) < LZSS_MAX_LEN ? (int)(n - pos) : LZSS_MAX_LEN;
    int rlc;
    uint64_t k = run_key_t(A, rlc, run, w0, ld_at);
    if (run >= LZSS_MIN_REPEAT && pos >= 2) {
        const uint64_t b = ld_at(-2);
        const uint32_t c = (uint32_t)(w0 & 0xFFull);
        if ((uint32_t)(b & 0xFFull) == c && (uint32_t)((b >> 8) & 0xFFull) == c) k |= PKEY_SKIP;
    }
    return k;
}
__device__ inline uint64_t run_key(const uint8_t *s, int A, int &rlc) { return run_key_w(s, A, rlc, -1, ld8(s)); }
// common prefix length of two strings from their (different) keys; the caller caps it by the strings' lengths
__device__ inline int key_lcp(uint64_t a, uint64_t b) {
    if ((a ^ b) >> 49) return 0;
    const int ca = (int)(a >> 40) & 511, cb = (int)(b >> 40) & 511;
    const int ra = ca == 255 ? DICT_RUN_CAP : (ca < 255 ? ca : 511 - ca);
    const int rb = cb == 255 ? DICT_RUN_CAP : (cb < 255 ? cb : 511 - cb);
    if (ca != cb) return ra < rb ? ra : rb;
    const uint64_t x = (a ^ b) & ((1ull << 40) - 1ull);
    return ra + (((int)__builtin_clzll(x) - 24) >> 3);
}

// ---- LCP array and its block minima -------------------------------------------------------------
__global__ __launch_bounds__(LCP_THREADS) void lcp_kernel(const Seg *segs, uint32_t n_segs, uint32_t gx, BatchIndex ix) {
    const uint32_t vb = xcd_block(blockIdx.x, gridDim.x);  // a segment's text and run lengths meet in one XCD's L2
    if (vb >= gx * n_segs) return;
    const Seg sg = segs[vb / gx];
    const uint32_t bx = vb % gx;
    const uint32_t N = sg.n + 1;
    if (bx * LCP_THREADS >= N) return;
    const uint32_t k = bx * LCP_THREADS + threadIdx.x;
    const uint8_t *T = ix.text + sg.off;
    const uint8_t *RL = ix.rlen + sg.off;
    const int32_t *SA = ix.sa + sg.off;
    uint32_t l = 0xFFFFu;
    int sak = INT_MAX;
    uint64_t wb = 0;
    int rb = 0;
    // the run length at a suffix: read off its first 8 bytes when the run ends inside them (no scattered load of its own)
    auto run_at = [&](int pos, uint64_t wd) -> int {
        const uint64_t x = wd ^ ((wd & 0xFFull) * 0x0101010101010101ull);
        if (x == 0) return (int)RL[pos];  // 8 equal bytes (or the zero padding): the suffix sort's count
        const int r8 = (int)(__builtin_ctzll(x) >> 3), rest = (int)sg.n - pos;
        return r8 < rest ? r8 : rest;     // a run of zeros must not continue into the padding
    };
    if (k < N) {  // the sentinel's "text" is the zero padding
        sak = SA[k];
        wb = ld8(T + sak);
        rb = run_at(sak, wb);
    }
    // the upper suffix of the pair is the neighbour lane's; the first lane of a wave fetches it
    int a = lane_below(sak, 0);
    uint64_t wa = lane_below(wb, 0ull);
    int ra = lane_below(rb, 0);
    if ((threadIdx.x & 63) == 0 && k > 0 && k < N) {
        a = SA[k - 1];
        wa = ld8(T + a);
        ra = run_at(a, wa);
    }
    if (k < N) {
        l = 0;
        if (k > 0) {
            const uint32_t hi = (uint32_t)(a > sak ? a : sak);
            int lim = (int)(sg.n - hi);  // the shorter suffix ends at the sentinel
            if (lim > LZSS_MAX_LEN) lim = LZSS_MAX_LEN;
            l = (uint32_t)lcp_run_rank(T, (uint32_t)a, (uint32_t)sak, wa, wb, ra, rb, lim);
        }
        ix.lcp[sg.off + k] = (uint16_t)l;
        if (k == N - 1)  // the upward walk of match_kernel reads up to three ranks past N and must die there
            for (uint32_t e = 0; e < 4; ++e) ix.lcp[sg.off + N + e] = 0;
        ix.pkey[sg.off + k] = pattern_word(sg.n, (uint32_t)sak, wb, rb, [&](int o) { return ld8(T + sak + o); });
    }
    // minimum of LCP and of SA over the 64 ranks of this wave
    const uint32_t m = (uint32_t)wave_min((int)l);  // (l <= 0xFFFF: the signed minimum is the unsigned one)
    const int ms = wave_min(sak);
    if ((threadIdx.x & 63) == 0) {
        ix.bmin1[sg.off / 64 + (k >> 6)] = (uint16_t)m;
        ix.samin1[sg.off / 64 + (k >> 6)] = ms;
    }
}

// The same for texts that fit LDS (up to LCP_LDS_MAX_N bytes): one 1024-thread workgroup per segment stages the whole
// text in LDS once (coalesced), and every scattered text read of the kernel - a suffix's first 8 bytes, the compare
// loop behind them - becomes an LDS read.  The global version is bound by the texture addresser (TA_BUSY 85 %: a
// wave-wide load whose lanes hit 64 different lines costs ~250 cycles of it); here the only global traffic left is the
// suffix array in (prefetched one step ahead) and the LCP array out, both streams.
constexpr int LCPL_THREADS = 1024;
#ifndef LCP_WIDE
#define LCP_WIDE 1   // compare steps of 16 bytes (0: 8 bytes)
#endif
constexpr uint32_t LCP_LDS_MAX_N = 150000;
__global__ __launch_bounds__(LCPL_THREADS) void lcp_lds_kernel(const Seg *segs, BatchIndex ix) {
    extern __shared__ uint32_t s_t32[];  // the text as words, zero padded (the 64 zero bytes behind it come along)
    const Seg sg = segs[blockIdx.x];
    const uint32_t n = sg.n, N = n + 1;
    const int tid = threadIdx.x, lane = tid & 63;
    const uint8_t *T = ix.text + sg.off;
    const uint8_t *RL = ix.rlen + sg.off;
    const int32_t *SA = ix.sa + sg.off;
    const uint32_t words = (n + 32) / 4 + 1;  // every 8-byte read below starts at or before n + 8 (compares stop at lim)
    for (uint32_t x = tid; x < words; x += LCPL_THREADS) s_t32[x] = reinterpret_cast<const uint32_t *>(T)[x];
    __syncthreads();
    auto ld8s = [&](uint32_t p) -> uint64_t {  // 8 bytes at any byte offset of the staged text
        const uint32_t i0 = p >> 2, sh = p & 3u;
        const uint32_t a0 = s_t32[i0], a1 = s_t32[i0 + 1], a2 = s_t32[i0 + 2];
        const uint32_t lo = __builtin_amdgcn_alignbyte(a1, a0, sh), hi = __builtin_amdgcn_alignbyte(a2, a1, sh);
        return (uint64_t)lo | ((uint64_t)hi << 32);
    };
    auto run_at = [&](int pos, uint64_t wd) -> int {
        const uint64_t x = wd ^ ((wd & 0xFFull) * 0x0101010101010101ull);
        if (x == 0) return (int)RL[pos];
        const int r8 = (int)(__builtin_ctzll(x) >> 3), rest = (int)n - pos;
        return r8 < rest ? r8 : rest;
    };
    int sak_next = (uint32_t)tid < N ? SA[tid] : INT_MAX;
    for (uint32_t base = 0; base < N; base += LCPL_THREADS) {
        const uint32_t k = base + (uint32_t)tid;
        const int sak = sak_next;
        sak_next = k + LCPL_THREADS < N ? SA[k + LCPL_THREADS] : INT_MAX;  // in flight while this step computes
        uint32_t l = 0xFFFFu;
        uint64_t wb = 0;
        int rb = 0;
        if (k < N) {
            wb = ld8s((uint32_t)sak);
            rb = run_at(sak, wb);
        }
        int a = lane_below(sak, 0);
        uint64_t wa = lane_below(wb, 0ull);
        int ra = lane_below(rb, 0);
        if (lane == 0 && k > 0 && k < N) {
            a = SA[k - 1];
            wa = ld8s((uint32_t)a);
            ra = run_at(a, wa);
        }
        if (k < N) {
            l = 0;
            if (k > 0) {
                const uint32_t hi = (uint32_t)(a > sak ? a : sak);
                int lim = (int)(n - hi);
                if (lim > LZSS_MAX_LEN) lim = LZSS_MAX_LEN;
                // lcp_run_rank with the compare loop on the staged text
                const uint64_t x = wa ^ wb;
                if (lim <= 0 || (x & 0xFFull)) l = 0;
                else {
                    const int r = ra < rb ? ra : rb;
                    if (ra != rb) l = (uint32_t)(r < lim ? r : lim);
                    else if (x) {
                        const int q = (int)(__builtin_ctzll(x) >> 3);
                        l = (uint32_t)(q < lim ? q : lim);
                    } else {
                        int o = r > 8 ? r : 8;
#if LCP_WIDE
                        while (o < lim) {  // 16 bytes per round trip: five words of each suffix (12.9 vs 13.7 ms; 32 bytes: 13.2)
                            const uint32_t pa = (uint32_t)a + (uint32_t)o, pb = (uint32_t)sak + (uint32_t)o;
                            const uint32_t ia = pa >> 2, sa_ = pa & 3u, ib = pb >> 2, sb_ = pb & 3u;
                            const uint32_t a0 = s_t32[ia], a1 = s_t32[ia + 1], a2 = s_t32[ia + 2], a3 = s_t32[ia + 3], a4 = s_t32[ia + 4];
                            const uint32_t b0 = s_t32[ib], b1 = s_t32[ib + 1], b2 = s_t32[ib + 2], b3 = s_t32[ib + 3], b4 = s_t32[ib + 4];
                            const uint32_t x0 = __builtin_amdgcn_alignbyte(a1, a0, sa_) ^ __builtin_amdgcn_alignbyte(b1, b0, sb_);
                            const uint32_t x1 = __builtin_amdgcn_alignbyte(a2, a1, sa_) ^ __builtin_amdgcn_alignbyte(b2, b1, sb_);
                            const uint32_t x2 = __builtin_amdgcn_alignbyte(a3, a2, sa_) ^ __builtin_amdgcn_alignbyte(b3, b2, sb_);
                            const uint32_t x3 = __builtin_amdgcn_alignbyte(a4, a3, sa_) ^ __builtin_amdgcn_alignbyte(b4, b3, sb_);
                            if (x0 | x1 | x2 | x3) {
                                if (x0) o += (int)(__builtin_ctz(x0) >> 3);
                                else if (x1) o += 4 + (int)(__builtin_ctz(x1) >> 3);
                                else if (x2) o += 8 + (int)(__builtin_ctz(x2) >> 3);
                                else o += 12 + (int)(__builtin_ctz(x3) >> 3);
                                break;
                            }
                            o += 16;
                        }
#else
                        while (o < lim) {
                            const uint64_t y = ld8s((uint32_t)a + (uint32_t)o) ^ ld8s((uint32_t)sak + (uint32_t)o);
                            if (y) {
                                o += (int)(__builtin_ctzll(y) >> 3);
                                break;
                            }
                            o += 8;
                        }
#endif
                        l = (uint32_t)(o < lim ? o : lim);
                    }
                }
            }
            ix.lcp[sg.off + k] = (uint16_t)l;
            if (k == N - 1)
                for (uint32_t e = 0; e < 4; ++e) ix.lcp[sg.off + N + e] = 0;
            ix.pkey[sg.off + k] = pattern_word(n, (uint32_t)sak, wb, rb, [&](int o) { return ld8s((uint32_t)(sak + o)); });
        }
        const uint32_t m = (uint32_t)wave_min((int)l);  // (l <= 0xFFFF: the signed minimum is the unsigned one)
        const int ms = wave_min(sak);
        if (lane == 0 && k < N) {  // (a block past the last rank would land in the next segment's summaries)
            ix.bmin1[sg.off / 64 + (k >> 6)] = (uint16_t)m;
            ix.samin1[sg.off / 64 + (k >> 6)] = ms;
        }
    }
}

__global__ __launch_bounds__(64) void bmin2_kernel(const Seg *segs, BatchIndex ix) {
    const Seg sg = segs[blockIdx.y];
    const uint32_t N = sg.n + 1;
    const uint32_t nb1 = (N + 63) / 64;
    const uint32_t c = blockIdx.x;
    if (c * 64 >= nb1) return;
    const uint32_t b = c * 64 + threadIdx.x;
    uint32_t m = b < nb1 ? (uint32_t)ix.bmin1[sg.off / 64 + b] : 0xFFFFu;
    m = (uint32_t)wave_min((int)m);
    if (threadIdx.x == 0) ix.bmin2[sg.off / 64 + c] = (uint16_t)m;
}

void launch_lcp(const Seg *d_segs, uint32_t n_segs, uint32_t max_n, const BatchIndex &ix, hipStream_t stream) {
    if (n_segs == 0) return;
    const uint32_t N = max_n + 1;
    const uint32_t gx = (N + LCP_THREADS - 1) / LCP_THREADS;
    if (max_n <= LCP_LDS_MAX_N) {
        static std::atomic<uint64_t> attr_devs{0};  // once per device (see launch_sa_build)
        int dev = 0;
        (void)hipGetDevice(&dev);
        const uint64_t dbit = 1ull << (dev & 63);
        if (!(attr_devs.load(std::memory_order_acquire) & dbit)) {
            (void)hipFuncSetAttribute((const void *)lcp_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)(((LCP_LDS_MAX_N + 32) / 4 + 1) * 4));
            attr_devs.fetch_or(dbit, std::memory_order_release);
        }
        hipLaunchKernelGGL(lcp_lds_kernel, dim3(n_segs), dim3(LCPL_THREADS), ((max_n + 32) / 4 + 1) * 4, stream, d_segs, ix);
    } else {
        hipLaunchKernelGGL(lcp_kernel, dim3(xcd_grid((uint64_t)gx * n_segs)), dim3(LCP_THREADS), 0, stream, d_segs, n_segs, gx, ix);
    }
    const uint32_t nb1 = (N + 63) / 64;
    dim3 g2((nb1 + 63) / 64, n_segs);
    hipLaunchKernelGGL(bmin2_kernel, g2, dim3(64), 0, stream, d_segs, ix);
}

// Auxiliary arrays of the dictionary index: a run-aware 8-byte key per rank (one load decides almost every probe
// of the binary search) and the rank range of every two-byte prefix (skips the first probes altogether).
__global__ __launch_bounds__(256) void dict_aux_kernel(const uint8_t *T, uint32_t n, const int32_t *SA, uint64_t *key,
                                                       int32_t *bkt) {
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;  // rank, 0 = sentinel
    if (k > n) return;
    const uint32_t q = (uint32_t)SA[k];
    int rlc;
    const uint64_t kk = k == 0 ? 0ull : run_key(T + q, (int)(n - q), rlc);  // the text is zero padded past n
    key[k] = kk;
    if (k == 0) return;
    {   // rank range of every (first byte, run code) = the top 17 key bits, behind the two-byte table
        const uint32_t kt = (uint32_t)(kk >> 40);
        bool first = k == 1, last = k == n;
        if (!first) {
            const uint32_t qp = (uint32_t)SA[k - 1];
            first = (uint32_t)(run_key(T + qp, (int)(n - qp), rlc) >> 40) != kt;
        }
        if (!last) {
            const uint32_t qn = (uint32_t)SA[k + 1];
            last = (uint32_t)(run_key(T + qn, (int)(n - qn), rlc) >> 40) != kt;
        }
        if (first) bkt[2 * (DICT_PAIR_BUCKETS + kt)] = (int32_t)k;
        if (last) bkt[2 * (DICT_PAIR_BUCKETS + kt) + 1] = (int32_t)k + 1;
    }
    if (q + 2 > n) return;
    const uint32_t b = ((uint32_t)T[q] << 8) | T[q + 1];
    const uint32_t qp = (uint32_t)SA[k - 1];
    const bool first = (k == 1) || qp + 2 > n || ((((uint32_t)T[qp] << 8) | T[qp + 1]) != b);
    if (first) bkt[2 * b] = (int32_t)k;
    bool last = (k == n);
    if (!last) {
        const uint32_t qn = (uint32_t)SA[k + 1];
        last = qn + 2 > n || ((((uint32_t)T[qn] << 8) | T[qn + 1]) != b);
    }
    if (last) bkt[2 * b + 1] = (int32_t)k + 1;
}

void launch_dict_aux(const uint8_t *d_text, uint32_t n, const int32_t *d_sa, uint64_t *d_key, int32_t *d_bkt, hipStream_t stream) {
    hipLaunchKernelGGL(dict_aux_kernel, dim3((n + 1 + 255) / 256), dim3(256), 0, stream, d_text, n, d_sa, d_key, d_bkt);
}

// exact common prefix of pat[0:maxLen) and the dictionary suffix of rank k (0 for the sentinel / virtual ranks):
// from the keys when they differ, else by comparing text past the run the equal keys vouch for
__device__ inline int dict_lcp(const DictView &D, const uint8_t *pat, int maxLen, uint64_t pk, int prl, int k, int q, uint64_t kq,
                               int known = -1) {
    if (k < 1 || k > (int)D.n) return 0;  // q = SA[k], kq = key[k] (whatever the arrays hold for the virtual rank n + 1)
    const int avail = (int)D.n - q;
    const int lim = maxLen < avail ? maxLen : avail;
    if (kq != pk) {
        const int l = key_lcp(pk, kq);
        return l < lim ? l : lim;
    }
    if (known >= 0) return known;  // the search compared this very suffix (dict_search_in)
    const int l_ = lcp_bytes(pat, D.T + q, prl < lim ? prl : lim, lim);
    DS(5, 1) DS(6, (l_ - (prl < lim ? prl : lim)) / 8 + 1)
    return l_;
}

// Binary search of pat[0:maxLen) among the dictionary suffixes.  Invariant: suffix(lo) < pat <= suffix(hi); rank 0 is
// the sentinel, rank n+1 is virtual.  Narrows [lo, hi] until hi == lo + 1; hi is then the lower bound of pat.
// pk / prl: run_key of the pattern and the run length it vouches for.  elo / ehi: the exact lcp(pat, suffix(lo / hi))
// when the text compare of a probe with an equal key has produced it, else -1 (ends with a different key get
// theirs from the keys, dict_lcp) - so the final step does not repeat the compares of the search.
template <class KeyAt, class SaAt>
__device__ inline void dict_search_in(const DictView &D, const uint8_t *pat, int maxLen, uint64_t pk, int prl, int &lo, int &hi,
                                      int &elo, int &ehi, KeyAt key_at, SaAt sa_at) {
    elo = -1;
    ehi = -1;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        const uint64_t kq = key_at(mid);
        bool suffix_less;
        int o = -1;
        DS(2, 1)
        if (kq != pk) {
            suffix_less = kq < pk;  // the key order is the string order
        } else {
            const int q = sa_at(mid);
            const int avail = (int)D.n - q;
            const int lim = maxLen < avail ? maxLen : avail;
            o = elo < ehi ? elo : ehi;  // both ends share that much with pat, so everything between does
            const int o1 = prl < lim ? prl : lim;
            DS(3, 1) DS(4, -(o > o1 ? o : o1))
            // (Taking the two bytes out of the words the compare loop loaded saves this round trip, but every form of it
            // tried - a bool or the two byte values leaving the loop - gave results that changed from run to run in
            // ~0.1 % of the blobs of a 3000-blob batch on hipcc / ROCm 7.2; the two byte loads are deterministic.)
            o = lcp_bytes(pat, D.T + q, o > o1 ? o : o1, lim);
            const bool b_less = o < lim && D.T[q + o] < pat[o];
            DS(4, o + 8)
            if (o == maxLen) suffix_less = false;          // pat is a prefix of the suffix
            else if (o == avail) suffix_less = true;       // the suffix is a proper prefix of pat
            else suffix_less = b_less;
        }
        if (suffix_less) { lo = mid; elo = o; } else { hi = mid; ehi = o; }
    }
}
// The same search in two phases, for a bracket staged in LDS.  The lanes of a wave step through their searches
// together, and in nearly every step of the interleaved search above SOME lane meets an equal key and sends the whole
// wave through the round trips of a text compare (measured: 6 % of the probes tie, 5.4 probes per lane).  The suffixes
// with a key equal to pk are consecutive in rank, so phase A finds the two ends of that run by the keys alone (LDS
// reads only) and phase B searches inside it, where every probe is a text compare - one or two steps instead of six.
template <class KeyAt, class SaAt>
__device__ inline void dict_search_staged(const DictView &D, const uint8_t *pat, int maxLen, uint64_t pk, int prl, int &lo, int &hi,
                                          int &elo, int &ehi, KeyAt key_at, SaAt sa_at MP_ARG) {
    elo = -1;
    ehi = -1;
    {   // phase A: the first rank whose key is >= pk, then the first whose key is > pk (hi itself is never read)
        int l = lo, h = hi;
        while (h - l > 1) {
            const int mid = (l + h) >> 1;
            DS(2, 1)
            if (key_at(mid) < pk) l = mid; else h = mid;
        }
        lo = l;
        h = hi;
        while (h - l > 1) {
            const int mid = (l + h) >> 1;
            DS(2, 1)
            if (key_at(mid) <= pk) l = mid; else h = mid;
        }
        hi = h;
    }
    MP(12)
    while (hi - lo > 1) {  // phase B: every rank between has the key pk
        const int mid = (lo + hi) >> 1;
        const int q = sa_at(mid);
        const int avail = (int)D.n - q;
        const int lim = maxLen < avail ? maxLen : avail;
        int o = elo < ehi ? elo : ehi;  // both ends share that much with pat, so everything between does
        const int o1 = prl < lim ? prl : lim;
        DS(3, 1) DS(4, -(o > o1 ? o : o1))
        o = lcp_bytes16(pat, D.T + q, o > o1 ? o : o1, lim);
        const bool b_less = o < lim && D.T[q + o] < pat[o];  // (two byte loads: see dict_search_in)
        DS(4, o + 8)
        bool suffix_less;
        if (o == maxLen) suffix_less = false;          // pat is a prefix of the suffix
        else if (o == avail) suffix_less = true;       // the suffix is a proper prefix of pat
        else suffix_less = b_less;
        if (suffix_less) { lo = mid; elo = o; } else { hi = mid; ehi = o; }
    }
}
__device__ inline void dict_search(const DictView &D, const uint8_t *pat, int maxLen, uint64_t pk, int prl, int &lo, int &hi,
                                   int &elo, int &ehi) {
    dict_search_in(D, pat, maxLen, pk, prl, lo, hi, elo, ehi, [&](int k) { return D.key[k]; }, [&](int k) { return D.SA[k]; });
}

// the bracket of every suffix sharing two bytes with pat; false if there is none
__device__ inline bool dict_bucket(const DictView &D, uint64_t w0, int &lo, int &hi) {
    const uint32_t b = ((uint32_t)(w0 & 0xFFull) << 8) | (uint32_t)((w0 >> 8) & 0xFFull);
    const int2 se = D.bkt[b];  // one load for both ends
    const int s = se.x, e = se.y;
    if (s >= e) return false;
    lo = lo > s - 1 ? lo : s - 1;
    hi = hi < e ? hi : e;
    return true;
}

// Longest prefix of pat[0:maxLen) occurring in the dictionary (Ld) and a rank kd that attains it, searched inside the
// bracket [lo, hi] (0, n + 1 for the whole dictionary).  minLen: shorter results are of no use to the caller (lets an
// empty two-byte bucket end the search).  run: the text's run length at pat, if known; w0: the first 8 bytes of pat.  The occurrence itself
// (dict_occurrence) is only needed for a reference that is written.
__device__ inline void dict_lookup(const DictView &D, const uint8_t *pat, int maxLen, int minLen, int lo, int hi, int &Ld,
                                   int &kd, int run, uint64_t w0 MP_ARG) {
    Ld = 0;
    kd = 0;
    DS(0, 1)
    if (maxLen >= 2 && !dict_bucket(D, w0, lo, hi) && minLen >= 2) { DS(1, 1) return; }  // nothing shares two bytes with pat
    DS(8, hi - lo)
    int prl;
    const uint64_t pk = run_key_w(pat, maxLen, prl, run, w0);
    if (prl >= 2) {  // inside a run the two-byte bucket is huge (zero-heavy data): narrow to the suffixes with the same run code
        const uint32_t kt = (uint32_t)(pk >> 40);
        const int2 se = D.bkt[DICT_PAIR_BUCKETS + kt];
        const int s = se.x, e = se.y;
        if (s < e) {  // the key order is the suffix order: the pattern's lower bound lies in [s, e]
            lo = lo > s - 1 ? lo : s - 1;
            hi = hi < e ? hi : e;
        }
    }
    DS(10, hi - lo)
    MP(12)
    int elo, ehi;
    dict_search(D, pat, maxLen, pk, prl, lo, hi, elo, ehi);
    MP(13)
    // hi == lo + 1: the suffix-array entries and the keys of both ends come with one load each
    int q2[2];
    uint64_t k2[2];
    __builtin_memcpy(q2, D.SA + lo, 8);
    __builtin_memcpy(k2, D.key + lo, 16);
    const int llo = dict_lcp(D, pat, maxLen, pk, prl, lo, q2[0], k2[0], elo);
    const int lhi = dict_lcp(D, pat, maxLen, pk, prl, hi, q2[1], k2[1], ehi);
    Ld = llo > lhi ? llo : lhi;
    MP(14)
    DS(9, Ld)
    // the smallest-rank occurrence is the first rank of the interval of suffixes sharing Ld bytes with rank kd
    // (dict_occurrence); when hi alone attains Ld that is hi itself (LCP[hi] = lcp with lo < Ld)
    kd = lhi > llo ? hi : lo;
}
// Lower bound of pat[0:maxLen) among the dictionary suffixes (the `hi` the binary search ends on).
__device__ inline int dict_lower_bound(const DictView &D, const uint8_t *pat, int maxLen, int run, uint64_t w0) {
    int lo = 0, hi = (int)D.n + 1, prl;
    const uint64_t pk = run_key_w(pat, maxLen, prl, run, w0);
    if (maxLen >= 2) (void)dict_bucket(D, w0, lo, hi);
    if (prl >= 2) {
        const int2 se = D.bkt[DICT_PAIR_BUCKETS + (uint32_t)(pk >> 40)];
        if (se.x < se.y) {
            lo = lo > se.x - 1 ? lo : se.x - 1;
            hi = hi < se.y ? hi : se.y;
        }
    }
    int elo, ehi;
    dict_search(D, pat, maxLen, pk, prl, lo, hi, elo, ehi);
    return hi;
}

// The patterns of a text are SORTED in suffix-rank order (a pattern is the suffix cut at maxLen <= 256, which keeps
// the order), so their lower bounds in the dictionary are monotone in the rank.  dict_bounds_kernel searches the
// dictionary for every 64th rank only (one lane per block of 64 ranks: dense waves, 1 / 64 of the probes);
// match_kernel's wave of ranks [64 b, 64 b + 64) then knows that all its lower bounds lie between dlb[b] and
// dlb[b + 1], stages the keys and suffix-array entries of that stretch of the dictionary in LDS with coalesced loads
// and every lane finishes its search there - no scattered probes at all.  Brackets wider than DICT_WIN ranks (a
// block in a stretch where the dictionary is much denser than the text) search global memory inside the bracket.
constexpr int DICT_WIN = 256;
__global__ __launch_bounds__(64) void dict_bounds_kernel(const Seg *segs, uint32_t nblk_max, BatchIndex ix, DictView D, int32_t *dlb) {
    const Seg sg = segs[blockIdx.y];
    const uint32_t N = sg.n + 1, nblk = (N + 63) / 64;
    const uint32_t b = blockIdx.x * 64 + threadIdx.x;
    if (b >= nblk) return;
    int lb = 1;  // rank 0 is the text's end sentinel: the empty pattern
    if (b > 0) {
        const uint32_t i = (uint32_t)ix.sa[sg.off + 64 * b];
        const int maxLen = (int)(sg.n - i) < LZSS_MAX_LEN ? (int)(sg.n - i) : LZSS_MAX_LEN;
        const uint8_t *pat = ix.text + sg.off + i;
        lb = dict_lower_bound(D, pat, maxLen, (int)ix.rlen[sg.off + i], ld8(pat));
    }
    dlb[sg.off / 64 + b] = lb;
}

// hF / hL: lower bounds of the block's first rank and of the next block's first rank (D.n + 1 past the last block)
__device__ inline void dict_lookup_wave(const DictView &D, const uint64_t *s_key, const int32_t *s_dsa, bool staged, int LB, int hL,
                                        bool need, const uint8_t *pat, int maxLen, int &Ld, int &kd, uint64_t pk MP_ARG) {
    Ld = 0;
    kd = 0;
    if (!need) return;
    const int prl = key_run(pk);  // pk: run_key of pat[0:maxLen) (pattern_word)
    int lo = LB, hi = hL, elo, ehi;
    int q2[2];
    uint64_t k2[2];
    if (staged) {
        dict_search_staged(D, pat, maxLen, pk, prl, lo, hi, elo, ehi, [&](int k) { return s_key[k - LB]; }, [&](int k) { return s_dsa[k - LB]; } MP_PASS);
        q2[0] = s_dsa[lo - LB]; q2[1] = s_dsa[lo - LB + 1];
        k2[0] = s_key[lo - LB]; k2[1] = s_key[lo - LB + 1];
    } else {
        dict_search(D, pat, maxLen, pk, prl, lo, hi, elo, ehi);
        __builtin_memcpy(q2, D.SA + lo, 8);
        __builtin_memcpy(k2, D.key + lo, 16);
    }
    MP(13)
    const int llo = dict_lcp(D, pat, maxLen, pk, prl, lo, q2[0], k2[0], elo);
    const int lhi = dict_lcp(D, pat, maxLen, pk, prl, hi, q2[1], k2[1], ehi);
    Ld = llo > lhi ? llo : lhi;
    kd = lhi > llo ? hi : lo;
    MP(14)
}
__device__ inline int dict_occurrence(const DictView &D, int kd, int Ld) {
    return D.SA[prev_lower(D.LCP, D.bmin1, D.bmin2, kd, Ld)];
}

// Apply the selection rule of compress.go:160-169 to the data-window results (Ls/ps: short window, LL/pL: long
// window, 0 / -1 = none) and the dictionary result (Ld/pd; the caller consults the dictionary only when
// dynOK && LL < maxLen, compress.go:460).
__device__ inline uint32_t select_backref(const DictView &D, int i, int minD, bool dynOK, int Ls, int ps, int LL, int pL,
                                          int Ld, int pd) {
    int Ldyn = 0;
    uint32_t fdyn = 0;
    if (dynOK) {
        if (LL > 0) {
            Ldyn = LL;
            fdyn = (uint32_t)(i - pL - 1);
        }
        if (Ld >= minD && Ld > LL) {  // compress.go:464: strictly longer
            Ldyn = Ld;
            fdyn = (uint32_t)(i + (int)D.n - pd - 1);  // backref.go:66, masked to 21 bits by res_pack
        }
    }
    // short wins iff found and 8*Ls-30 > 8*Ldyn-37, i.e. Ls >= Ldyn (compress.go:165)
    if (Ls > 0 && Ls >= Ldyn) return res_pack(RES_SHORT, (uint32_t)Ls, (uint32_t)(i - ps - 1));
    if (Ldyn > 0) return res_pack(RES_DYN, (uint32_t)Ldyn, fdyn);
    return res_pack(RES_NONE, 0, 0);
}

#ifndef MATCH_SCAN_CAP
#define MATCH_SCAN_CAP 7   // (measured per 10 000 blobs: 4 55.2, 5 53.7, 6 53.2, 7 52.9, 8 53.9, 12 56.6 ms)
#endif
constexpr int SCAN_CAP = MATCH_SCAN_CAP;  // x4 ranks a lane walks alone before its wave takes the walk over
// ranks staged in LDS on each side of the wave's 64: what the lane-private walks reach (4 * SCAN_CAP + 4).  A wider
// window also serves the first 64-rank step of the walks the wave takes over (they check for it), but staging 96 ranks
// per side so that every such first step is an LDS read gained nothing: 48.8 vs 48.6 ms (64 with SCAN_CAP 5: 48.9)
#ifndef MATCH_STAGE_HALO
#define MATCH_STAGE_HALO (4 * MATCH_SCAN_CAP + 4)
#endif
constexpr int WIN_HALO = MATCH_STAGE_HALO;
static_assert(WIN_HALO >= 4 * SCAN_CAP + 4 && WIN_HALO % 4 == 0, "the lane-private walks read the staged window only");
constexpr int WIN_TOTAL = 64 + 2 * WIN_HALO;

__device__ inline uint32_t min3u(uint32_t a, uint32_t b, uint32_t c) {
    const uint32_t t = a < b ? a : b;
    return t < c ? t : c;
}
__device__ inline uint32_t max3u(uint32_t a, uint32_t b, uint32_t c) {
    const uint32_t t = a > b ? a : b;
    return t > c ? t : c;
}

__device__ inline int wave_incl_min(int v, int lane) {
    (void)lane;
    return wave_incl_min_dpp(v);
}

// One lane per rank.  The walks to the nearest in-window neighbours are short for most ranks and very long
// for a few (heavy tail), so a lane walks SCAN_CAP x 4 ranks alone; whatever is left is finished by the
// whole wave, 64 ranks per step (coalesced loads, prefix-min of LCP across lanes, ballot for the first hit).
#ifndef MATCH_WAVES_PER_EU
#define MATCH_WAVES_PER_EU 8
#endif
__global__ __launch_bounds__(MATCH_THREADS) __attribute__((amdgpu_waves_per_eu(MATCH_WAVES_PER_EU, MATCH_WAVES_PER_EU))) void match_kernel(const Seg *segs, uint32_t n_segs, uint32_t gx, BatchIndex ix,
                                                              DictView D, const int32_t *dlb, uint32_t *res, int skip_unqueried, int defer,
                                                              int ablate) {
    const uint32_t vb = xcd_block(blockIdx.x, gridDim.x);
    if (vb >= gx * n_segs) return;
    const uint32_t bx = vb % gx;
    const Seg sg = segs[vb / gx];
    const int n = (int)sg.n;
    const int N = n + 1;
    if ((int)(bx * MATCH_THREADS) >= N) return;
    const int lane = threadIdx.x & 63;
    const int r = (int)(bx * MATCH_THREADS + threadIdx.x);
    MP_DECL
    const uint8_t *T = ix.text + sg.off;
    const int32_t *SA = ix.sa + sg.off;
    const uint16_t *LCP = ix.lcp + sg.off;

    // the wave's neighbourhood in rank, [R0 - WIN_HALO, R0 + 64 + WIN_HALO), goes to LDS once (coalesced): the
    // lane-private part of the walks then reads LDS instead of issuing 64 overlapping 16-byte loads per step
    static_assert(MATCH_THREADS == 64, "one wave per block: the window belongs to the wave");
    __shared__ int32_t s_sa[WIN_TOTAL + 4];
    __shared__ uint16_t s_lc[WIN_TOTAL + 4];  // (lcp <= 256)
    __shared__ uint64_t s_dkey[DICT_WIN + 2];  // the wave's bracket of the dictionary (dict_lookup_wave)
    __shared__ int32_t s_dsa[DICT_WIN + 2];
    const int wbase = (int)(bx * MATCH_THREADS) - WIN_HALO;
    // the lane's own position comes straight from memory, so that the scattered reads that depend on it (below) are in
    // flight while the windows are staged
    bool work = r >= 1 && r < N;
    int i = work ? SA[r] : 0;
    work = work && !(i < (int)sg.start || i < HEAD);
    // what the LCP kernel left for this rank (pattern_word): the pattern's dictionary key and the RLE skip flag - one
    // coalesced load instead of gathers of the text and the run lengths at the lane's position
    const uint64_t pkw = (r >= 1 && r < N) ? ix.pkey[sg.off + r] : 0ull;
    for (int j = lane; j < WIN_TOTAL; j += 64) {
        int g = wbase + j;
        g = g < 0 ? 0 : (g > N + 3 ? N + 3 : g);  // LCP is zero at rank 0 and from N to N + 3: walks die there
        s_sa[j] = SA[g];
        s_lc[j] = LCP[g];
    }
    if (lane < 4) {  // what a lane that has finished reads: outside every window, an lcp that changes nothing
        s_sa[WIN_TOTAL + lane] = 0x7FFFFFFF;
        s_lc[WIN_TOTAL + lane] = 0xFFFFu;
    }
    // the block's stretch of the dictionary (dict_bounds_kernel): ranks dLB .. dhL hold every lane's (lo, hi) pair
    const int nblk = (N + 63) >> 6;
    const int dLB = dlb[sg.off / 64 + bx] - 1;
    const int dhL = (int)bx + 1 < nblk ? dlb[sg.off / 64 + bx + 1] : (int)D.n + 1;
    const bool dstaged = dhL - dLB + 1 <= DICT_WIN;
    if (dstaged) {
        for (int j = lane; j <= dhL - dLB; j += 64) {
            s_dkey[j] = D.key[dLB + j];
            s_dsa[j] = D.SA[dLB + j];
        }
    }
    __syncthreads();
    // the walk can never call bestBackref(i) when i, i-1 and i-2 lie in one run of equal bytes and at least 160 of them
    // follow i: the RLE branch (compress.go:184) is taken at i, i-1 and i-2 (run lengths capped at 255)
    if (work && skip_unqueried && (pkw & PKEY_SKIP)) {
        res[sg.off + i] = res_pack(RES_NONE, 0, 0);
        work = false;
    }
    const int maxLen = (n - i) < LZSS_MAX_LEN ? (n - i) : LZSS_MAX_LEN;
    const bool reserved = work && (uint32_t)((pkw >> 49) & 0xFFull) >= LZSS_SYMBOL_SHORT;
    const int minS = reserved ? 1 : LZSS_SHORT_MIN_LEN;
    const int minD = reserved ? 1 : LZSS_DYN_MIN_LEN;
    const bool shortOK = maxLen >= minS, dynOK = maxLen >= minD;
    const int wsS = i > LZSS_SHORT_WINDOW ? i - LZSS_SHORT_WINDOW : 0;
    const int wsL = i > LZSS_DYN_WINDOW ? i - LZSS_DYN_WINDOW : 0;
    const bool scan = work && shortOK && !(ablate & 4);

    // nearest in-window neighbour per side and window as one word, rank << 9 | lcp with it (lcp <= 256, rank <= 2^22):
    // walking up the nearest is the smallest such word, walking down the largest, so a rank costs one min / max per
    // window instead of a chain of mask operations on the scalar unit
    constexpr uint32_t NONE_UP = 0xFFFFFFFFu, NONE_DN = 0u;
    uint32_t qupS = NONE_UP, qupL = NONE_UP, qdnS = NONE_DN, qdnL = NONE_DN;
    // p lies in a window <=> (uint)(i - 1 - p) < span.  A walk does not test whether it is still alive (running lcp
    // >= minS) rank by rank: the lcp never grows along a walk, so the first in-window rank is found regardless and a
    // dead one carries an lcp below the minimum, which the length selection further down discards anyway
    const uint32_t im1 = (uint32_t)(i - 1), spanS = (uint32_t)(i - wsS), spanL = (uint32_t)(i - wsL);

    MP(0)
    // ---- nearest in-window neighbours above r (LCP is zero for four ranks past N: the walk dies there)
    {
        int m = 1 << 20, k = r + 1;
        bool done = !scan || k >= N;
        for (int it = 0; it < SCAN_CAP && __ballot(!done); ++it) {
            const int j = done ? WIN_TOTAL : k - wbase;
            uint32_t cS[4], cL[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int l = (int)s_lc[j + t];
                m = l < m ? l : m;
                const uint32_t d = im1 - (uint32_t)s_sa[j + t];  // offset - 1; wraps to >= 2^31 for positions >= i
                const uint32_t key = ((uint32_t)(k + t) << 9) | (uint32_t)m;
                cS[t] = d < spanS ? key : NONE_UP;
                cL[t] = d < spanL ? key : NONE_UP;
            }
            qupS = min3u(qupS, min3u(cS[0], cS[1], cS[2]), cS[3]);
            qupL = min3u(qupL, min3u(cL[0], cL[1], cL[2]), cL[3]);
            done = done || m < minS || qupS != NONE_UP || k + 4 >= N;
            // a long-window hit of a useful length ends the search for a short-window hit as soon as the running lcp has
            // dropped below it: the short window lies inside the long one, and a short reference only wins with
            // Ls >= LL (compress.go:165), so a shorter short-window match on this side can never be selected
            const int lf = qupL != NONE_UP ? (int)(qupL & 511u) : 0;
            done = done || (lf >= minD && m < lf);
            k = done ? k : k + 4;
        }
        MP(1)
        for (uint64_t hm = (ablate & 16) ? 0ull : __ballot(!done); hm; hm &= hm - 1) {  // the wave finishes lane t's walk
            const int t = (int)__builtin_ctzll(hm);
            const int ci = lane_value(i, t), cws = lane_value(wsS, t), cwl = lane_value(wsL, t), cminD = lane_value(minD, t);
            int cmin = lane_value(minS, t);  // the lcp below which the walk ends; raised by a useful long-window hit (see above)
            int cm = lane_value(m, t);
            bool cfL = lane_value((int)(qupL != NONE_UP), t) != 0;
            {
                const int lv = lane_value((int)(qupL & 511u), t);
                if (cfL && lv >= cminD && lv > cmin) cmin = lv;
            }
            uint32_t rL = NONE_UP, rS = NONE_UP;
            DS(11, 1)
            for (int base = lane_value(k, t); base < N; base += 64) {
                DS(12, 1)
                const int kk = base + lane;
                const bool valid = kk < N;
                int l = 0, p = 0x7FFFFFFF;
                if (base - wbase + 64 <= WIN_TOTAL) {  // (base >= wbase: the walk starts above the wave's first rank)
                    if (valid) { l = (int)s_lc[kk - wbase]; p = s_sa[kk - wbase]; }
                } else if (valid) {
                    l = (int)LCP[kk];
                    p = SA[kk];
                }
                int pm = wave_incl_min(l, lane);
                pm = pm < cm ? pm : cm;
                if (!cfL) {
                    const uint64_t bL = __ballot(pm >= cmin && p < ci && p >= cwl);
                    if (bL) {
                        const int t2 = (int)__builtin_ctzll(bL);
                        const int lv = lane_value(pm, t2);
                        rL = ((uint32_t)(base + t2) << 9) | (uint32_t)lv;
                        cfL = true;
                        if (lv >= cminD && lv > cmin) cmin = lv;
                    }
                }
                const bool alive = pm >= cmin;
                const uint64_t bS = __ballot(alive && p < ci && p >= cws);
                const uint64_t bDead = __ballot(!alive);
                if (bS) {
                    const int t2 = (int)__builtin_ctzll(bS);
                    rS = ((uint32_t)(base + t2) << 9) | (uint32_t)lane_value(pm, t2);
                    break;
                }
                if (bDead) break;
                cm = lane_value(pm, 63);
            }
            if (lane == t) {
                if (rL != NONE_UP) qupL = rL;
                qupS = rS;
            }
        }
    }
    MP(2)
    // ---- nearest in-window neighbours below r: lcp(i, SA[k]) = min LCP[k+1 .. r]; rank 0 is the sentinel
    {
        int m = 1 << 20, k = r - 1;
        bool done = !scan;
        // the same bound from the other side: a useful long-window match above r makes every shorter match below r
        // irrelevant in both windows (LL is the larger of the two sides, and Ls must reach LL to be selected)
        const int upL0 = qupL != NONE_UP ? (int)(qupL & 511u) : 0;
        const int floor0 = (upL0 >= minD && upL0 > minS) ? upL0 : minS;
        for (int it = 0; it < SCAN_CAP && __ballot(!done && k >= 4); ++it) {  // ranks k, k-1, k-2, k-3
            const bool go = !done && k >= 4;
            const int j = go ? k - 3 - wbase : WIN_TOTAL;  // rank kk's lcp with its upper neighbour is LCP[kk + 1]
            uint32_t cS[4], cL[4];
#pragma unroll
            for (int t = 3; t >= 0; --t) {
                const int l = (int)s_lc[go ? j + t + 1 : j + t];
                m = l < m ? l : m;
                const uint32_t d = im1 - (uint32_t)s_sa[j + t];
                const uint32_t key = ((uint32_t)(k - 3 + t) << 9) | (uint32_t)m;
                cS[t] = d < spanS ? key : NONE_DN;
                cL[t] = d < spanL ? key : NONE_DN;
            }
            qdnS = max3u(qdnS, max3u(cS[0], cS[1], cS[2]), cS[3]);
            qdnL = max3u(qdnL, max3u(cL[0], cL[1], cL[2]), cL[3]);
            const int lf = (int)(qdnL & 511u);  // 0 while there is no long-window hit on this side
            const bool stop = m < floor0 || qdnS != NONE_DN || (lf >= minD && m < lf);
            done = done || (go && stop);
            k = (go && !stop) ? k - 4 : k;
        }
        if (!done && k < 4) {
            for (; k >= 1; --k) {
                const int l = LCP[k + 1];
                m = l < m ? l : m;
                if (m < floor0) break;
                const int p = SA[k];
                if (p < i) {
                    const uint32_t key = ((uint32_t)k << 9) | (uint32_t)m;
                    if (qdnL == NONE_DN && p >= wsL) qdnL = key;
                    if (p >= wsS) { qdnS = key; break; }
                }
            }
            done = true;
        }
        // A short-window match of the full length above r settles everything but the address bits: the short
        // reference wins whatever lies below r, and its address is the smallest-rank in-window occurrence of the
        // interval, which the interval search finds from the upper neighbour just as well.  Without this the walk
        // below r runs through every shorter suffix of a periodic text (all of them full-length matches at positions
        // past i): 1.7 s for 2 MiB of "abc".
        if (!done && qupS != NONE_UP && (int)(qupS & 511u) == maxLen) {
            qdnS = qupS;
            done = true;
        }
        MP(3)
        for (uint64_t hm = (ablate & 16) ? 0ull : __ballot(!done); hm; hm &= hm - 1) {
            const int t = (int)__builtin_ctzll(hm);
            const int ci = lane_value(i, t), cws = lane_value(wsS, t), cwl = lane_value(wsL, t), cminD = lane_value(minD, t);
            int cmin = lane_value(floor0, t);
            int cm = lane_value(m, t);
            bool cfL = lane_value((int)(qdnL != NONE_DN), t) != 0;
            {
                const int lv = lane_value((int)(qdnL & 511u), t);
                if (cfL && lv >= cminD && lv > cmin) cmin = lv;
            }
            uint32_t rL = NONE_DN, rS = NONE_DN;
            DS(13, 1)
            for (int base = lane_value(k, t); base >= 1; base -= 64) {
                DS(14, 1)
                const int kk = base - lane;
                const bool valid = kk >= 1;
                int l = 0, p = 0x7FFFFFFF;
                if (base - 63 >= wbase) {  // (base + 1 < wbase + WIN_TOTAL: the walk starts below the wave's last rank)
                    if (valid) { l = (int)s_lc[kk + 1 - wbase]; p = s_sa[kk - wbase]; }
                } else if (valid) {
                    l = (int)LCP[kk + 1];
                    p = SA[kk];
                }
                int pm = wave_incl_min(l, lane);
                pm = pm < cm ? pm : cm;
                if (!cfL) {
                    const uint64_t bL = __ballot(pm >= cmin && p < ci && p >= cwl);
                    if (bL) {
                        const int t2 = (int)__builtin_ctzll(bL);
                        const int lv = lane_value(pm, t2);
                        rL = ((uint32_t)(base - t2) << 9) | (uint32_t)lv;
                        cfL = true;
                        if (lv >= cminD && lv > cmin) cmin = lv;
                    }
                }
                const bool alive = pm >= cmin;
                const uint64_t bS = __ballot(alive && p < ci && p >= cws);
                const uint64_t bDead = __ballot(!alive);
                if (bS) {
                    const int t2 = (int)__builtin_ctzll(bS);
                    rS = ((uint32_t)(base - t2) << 9) | (uint32_t)lane_value(pm, t2);
                    break;
                }
                if (bDead) break;
                cm = lane_value(pm, 63);
            }
            if (lane == t) {
                if (rL != NONE_DN) qdnL = rL;
                qdnS = rS;
            }
        }
    }
    const int LupS = qupS != NONE_UP ? (int)(qupS & 511u) : 0, kupS = (int)(qupS >> 9);
    const int LupL = qupL != NONE_UP ? (int)(qupL & 511u) : 0, kupL = (int)(qupL >> 9);
    const int LdnS = (int)(qdnS & 511u), kdnS = (int)(qdnS >> 9);
    const int LdnL = (int)(qdnL & 511u), kdnL = (int)(qdnL >> 9);
    MP(4)
    const SegIndex S = seg_index(ix, sg);
    // lengths in both windows (compress.go:146-159)
    int Ls = 0, LL = 0;
    bool upS = false, upL = false;  // the upper neighbour alone attains the length: it IS the smallest-rank occurrence
    if (work) {
        const int ld = LdnS >= minS ? LdnS : 0, lu = LupS >= minS ? LupS : 0;
        Ls = ld > lu ? ld : lu;
        upS = lu > ld;
    }
    if (work && dynOK) {
        const int ld = LdnL >= minD ? LdnL : 0, lu = LupL >= minD ? LupL : 0;
        LL = ld > lu ? ld : lu;
        upL = lu > ld;
    }
    MP(5)
    // dictionary (compress.go:460-466): the lanes' patterns are sorted, so the wave brackets them between the lower
    // bounds of its first and last lane and finishes inside LDS (dict_lookup_wave)
    int Ld = 0, kd = 0;
    dict_lookup_wave(D, s_dkey, s_dsa, dstaged, dLB, dhL, work && dynOK && LL < maxLen && !(ablate & 1), T + i, maxLen, Ld, kd,
                     pkw & PKEY_KEY_MASK MP_PASS);
    MP(7)
    // selection (compress.go:160-169) BEFORE the offsets: only the window whose reference is emitted needs its
    // smallest-rank occurrence, so every lane runs at most one interval search
    const bool from_dict = dynOK && Ld >= minD && Ld > LL;  // compress.go:464: strictly longer
    const int Ldyn = dynOK ? (from_dict ? Ld : LL) : 0;
    const bool short_wins = work && Ls > 0 && Ls >= Ldyn;   // 8*Ls-30 > 8*Ldyn-37 (compress.go:165)
    const bool long_wins = work && !short_wins && Ldyn > 0 && !from_dict;
    // the smallest-rank occurrence (the address bits).  When the lower neighbour attains the length it takes an
    // interval search; with `defer` that is left to emit_kernel, which runs it for the phrases it writes only
    int p = -1;
    bool pending = false;
    if (short_wins || long_wins) {
        const bool up = short_wins ? upS : upL;
        const int kdn = short_wins ? kdnS : kdnL;
        if (up) {  // mostly still inside the staged window
            const int ku = short_wins ? kupS : kupL;
            p = (ku - wbase < WIN_TOTAL) ? s_sa[ku - wbase] : SA[ku];
        }
        else if (kdn >= wbase && kdn - wbase < WIN_TOTAL && (int)s_lc[kdn - wbase] < (short_wins ? Ls : LL)) {
            p = s_sa[kdn - wbase];  // rank kdn - 1 shares fewer bytes: kdn is the first rank of the interval, hence the answer
        }
        else if (defer) { pending = true; p = kdn; }
        else if (ablate & 2) p = SA[kdn];
        else p = leftmost(S, kdn, short_wins ? Ls : LL, short_wins ? wsS : wsL, i, short_wins MP_PASS);
    }
    MP(6)
    MP_PRINT
    if (!work) return;
    uint32_t out = res_pack(RES_NONE, 0, 0);
    if (pending) out = res_pack(short_wins ? RES_SHORT : RES_DYN, (uint32_t)(short_wins ? Ls : LL), (uint32_t)p) | RES_PENDING;
    else if (short_wins) out = res_pack(RES_SHORT, (uint32_t)Ls, (uint32_t)(i - p - 1));
    else if (long_wins) out = res_pack(RES_DYN, (uint32_t)LL, (uint32_t)(i - p - 1));
    else if (Ldyn > 0) {  // the dictionary reference: address bits i + DictLen - address - 1 masked to 21 (backref.go:66)
        if (defer && D.n + 1 < RES_DEFER_MAX_N) out = res_pack(RES_DICT, (uint32_t)Ld, (uint32_t)kd);
        else out = res_pack(RES_DYN, (uint32_t)Ld, (uint32_t)(i + (int)D.n - dict_occurrence(D, kd, Ld) - 1));
    }
    res[sg.off + i] = out;
}

// positions i < HEAD: brute force over the window [0, i) (identical for both window sizes there)
__global__ __launch_bounds__(64) void match_head_kernel(const Seg *segs, BatchIndex ix, DictView D, uint32_t *res) {
    const Seg sg = segs[blockIdx.y];
    const int n = (int)sg.n;
    const int i = (int)(blockIdx.x * 64 + threadIdx.x);
    if (i >= n || i >= HEAD || i < (int)sg.start) return;
    const uint8_t *T = ix.text + sg.off;
    const uint8_t *RL = ix.rlen + sg.off;
    const int32_t *ISA = ix.isa + sg.off;
#ifdef MATCH_PROFILE
    long long mp_none_acc[15], mp_none_last = 0;
#endif

    const int maxLen = (n - i) < LZSS_MAX_LEN ? (n - i) : LZSS_MAX_LEN;
    const bool reserved = T[i] >= LZSS_SYMBOL_SHORT;
    const int minS = reserved ? 1 : LZSS_SHORT_MIN_LEN;
    const int minD = reserved ? 1 : LZSS_DYN_MIN_LEN;
    const bool shortOK = maxLen >= minS, dynOK = maxLen >= minD;

    int best = 0, bp = -1, brank = 0x7FFFFFFF;
    if (shortOK) {
        for (int p = 0; p < i; ++p) {
            const int l = lcp_run(T, RL, (uint32_t)p, (uint32_t)i, maxLen);
            if (l < minS || l < best) continue;
            const int rk = ISA[p];
            if (l > best || rk < brank) { best = l; bp = p; brank = rk; }
        }
    }
    const int Ls = best, ps = bp;
    const int LL = (dynOK && best >= minD) ? best : 0, pL = bp;
    int Ld = 0, kd = 0, pd = -1;
    if (dynOK && LL < maxLen) dict_lookup(D, T + i, maxLen, minD, 0, (int)D.n + 1, Ld, kd, (int)RL[i], ld8(T + i) MP_NONE);  // compress.go:460
    if (Ld >= minD && Ld > LL) pd = dict_occurrence(D, kd, Ld);
    res[sg.off + i] = select_backref(D, i, minD, dynOK, Ls, ps, LL, pL, Ld, pd);
}

void launch_match_head(const Seg *d_segs, uint32_t n_segs, uint32_t max_n, const BatchIndex &ix, const DictView &dict,
                       uint32_t *d_res, hipStream_t stream) {
    if (n_segs == 0 || max_n == 0) return;
    const uint32_t head = max_n < (uint32_t)HEAD ? max_n : (uint32_t)HEAD;
    dim3 gh((head + 63) / 64, n_segs);
    hipLaunchKernelGGL(match_head_kernel, gh, dim3(64), 0, stream, d_segs, ix, dict, d_res);
    // the dictionary brackets of match_kernel's blocks ride in this stage, so that the "match" stage of
    // mi355_lzss_batch_stage_times is match_kernel alone (what the roofline prices and rocprofv3 lists)
    if (max_n > (uint32_t)HEAD) {
        const uint32_t gx = (max_n + 1 + MATCH_THREADS - 1) / MATCH_THREADS;
        hipLaunchKernelGGL(dict_bounds_kernel, dim3((gx + 63) / 64, n_segs), dim3(64), 0, stream, d_segs, gx, ix, dict, ix.dlb);
    }
}

void launch_match(const Seg *d_segs, uint32_t n_segs, uint32_t max_n, const BatchIndex &ix, const DictView &dict,
                  uint32_t *d_res, bool skip_unqueried, hipStream_t stream) {
    if (n_segs == 0 || max_n <= (uint32_t)HEAD) return;
#ifdef LZSS_DEBUG_KNOBS  // timing experiments only (scripts/build_variants.sh); the shipped library ignores the environment
    static const int ablate = getenv("MI355_LZSS_ABLATE") ? atoi(getenv("MI355_LZSS_ABLATE")) : 0;
    static const bool no_defer = getenv("MI355_LZSS_NO_DEFER") != nullptr;
#else
    constexpr int ablate = 0;
    constexpr bool no_defer = false;
#endif
    const uint32_t gx = (max_n + 1 + MATCH_THREADS - 1) / MATCH_THREADS;  // blocks per segment
    // for_emit: the results feed the phrase walk and the emitter only (not the per-position debug view)
    const int defer = (skip_unqueried && !no_defer && max_n + 1 < RES_DEFER_MAX_N) ? 1 : 0;
    hipLaunchKernelGGL(match_kernel, dim3(xcd_grid((uint64_t)gx * n_segs)), dim3(MATCH_THREADS), 0, stream, d_segs, n_segs, gx,
                       ix, dict, ix.dlb, d_res, skip_unqueried ? 1 : 0, defer, ablate);
}

}  // namespace lzss

#ifdef DICT_STATS
extern "C" int mi355_lzss_debug_dict_stats(unsigned long long *out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(lzss::g_ds), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(lzss::g_ds), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

#ifdef MATCH_PROFILE
extern "C" int mi355_lzss_debug_match_profile(unsigned long long *out16, int reset) {
    if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(lzss::g_mp), 16 * sizeof(unsigned long long)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(lzss::g_mp), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
