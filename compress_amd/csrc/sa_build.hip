// sa_build.hip - suffix array + inverse of every segment of a batch, one 1024-thread workgroup per
// segment (gfx950).  Replaces suffixarray.New (lzss/internal/suffixarray/suffixarray.go:32-45, which
// runs SA-IS, sais.go:128-203) for both the input index (lzss/compress.go:116) and the dictionary
// index (compress.go:64).  The suffix array of a text is a unique object, so the construction is
// free to be GPU-shaped: prefix doubling (Manber-Myers bucket induction), every round one stable
// LSD radix sort of the rank keys, stability from wave-level ballot matching, 16 wave-private
// histograms in LDS.
//
// N-world: the text gets an explicit smallest sentinel at index n, N = n + 1 suffixes, SA[0] == n.
//
// Round with sorted depth h (suffixes are ordered by their first h symbols, RK[p] = first index of
// p's group): walk the current order j = 0..N-1 and emit p = SA[j] - h; a stable sort of that
// sequence by RK[p] orders every group by its second half.  Slots with SA[j] < h stand for the h
// suffixes p >= N - h, whose first h symbols contain the sentinel and which are therefore already
// alone in their group, so their place in the sequence does not matter: p = N - h + SA[j] keeps the
// sequence a permutation of 0..N-1.
#include <cstdlib>

#include "kernels.h"

namespace lzss {

constexpr int SA_THREADS = 1024;
constexpr int SA_WAVES = SA_THREADS / 64;
constexpr int SA_MAXDB = 9;              // radix digit width (bits) at most
constexpr int SA_BINS = 1 << SA_MAXDB;

struct SaShared {
    int32_t hist[SA_WAVES][SA_BINS];
    int32_t colsum[SA_BINS];
    int32_t wlast[SA_WAVES];
    int32_t wcount[SA_WAVES];
};

__device__ inline uint64_t lanemask_lt(int lane) { return (1ull << lane) - 1ull; }

// lanes of `amask` holding the same digit as this lane
__device__ inline uint64_t match_digit(int d, uint64_t amask) {
    uint64_t peers = amask;
#pragma unroll
    for (int bit = 0; bit < SA_MAXDB; ++bit) {
        const bool b = (d >> bit) & 1;
        const uint64_t bal = __ballot(b);
        peers &= b ? bal : ~bal;
    }
    return peers;
}

__device__ inline int wave_incl_add(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(v, o);
        if (lane >= o) v += t;
    }
    return v;
}
__device__ inline int wave_incl_max(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        int t = __shfl_up(v, o);
        if (lane >= o) v = v > t ? v : t;
    }
    return v;
}

__device__ inline void hist_clear(SaShared &sh, int tid) {
    for (int x = tid; x < SA_WAVES * SA_BINS; x += SA_THREADS) (&sh.hist[0][0])[x] = 0;
}

// one wave adds the digits of its 64 keys to its private histogram row
__device__ inline void hist_add(int32_t *hrow, int d, bool active, int lane) {
    const uint64_t amask = __ballot(active);
    const uint64_t peers = match_digit(d, amask);
    const int cnt = __popcll(peers);
    const int rank = __popcll(peers & lanemask_lt(lane));
    if (active && rank == cnt - 1) hrow[d] += cnt;
}

// hist[w][d] (counts) -> first output index of wave w's keys with digit d (digit-major, wave-minor)
__device__ inline void hist_scan(SaShared &sh, int tid, int nbins) {
    __syncthreads();
    if (tid < nbins) {
        int s = 0;
        for (int w = 0; w < SA_WAVES; ++w) {
            int t = sh.hist[w][tid];
            sh.hist[w][tid] = s;
            s += t;
        }
        sh.colsum[tid] = s;
    }
    __syncthreads();
    if (tid < 64) {
        int carry = 0;
        for (int base = 0; base < nbins; base += 64) {
            int v = sh.colsum[base + tid];
            int inc = wave_incl_add(v, tid);
            sh.colsum[base + tid] = carry + inc - v;
            carry += __shfl(inc, 63);
        }
    }
    __syncthreads();
    if (tid < nbins) {
        int base = sh.colsum[tid];
        for (int w = 0; w < SA_WAVES; ++w) sh.hist[w][tid] += base;
    }
    __syncthreads();
}

// stable scatter of this wave's chunk [cbeg, cend) by digit (key >> shift) & mask
__device__ inline void scatter_chunk(SaShared &sh, const int32_t *kin, const int32_t *vin, int32_t *kout,
                                     int32_t *vout, uint32_t cbeg, uint32_t cend, int shift, int mask, int w,
                                     int lane) {
    int32_t *hrow = sh.hist[w];
    for (uint32_t j0 = cbeg; j0 < cend; j0 += 64) {
        const uint32_t j = j0 + lane;
        const bool active = j < cend;
        int key = 0, val = 0;
        if (active) {
            key = kin[j];
            val = vin[j];
        }
        const int d = (key >> shift) & mask;
        const uint64_t amask = __ballot(active);
        const uint64_t peers = match_digit(d, amask);
        const int cnt = __popcll(peers);
        const int rank = __popcll(peers & lanemask_lt(lane));
        const int base = hrow[d];
        if (active) {
            kout[base + rank] = key;
            vout[base + rank] = val;
        }
        __builtin_amdgcn_wave_barrier();
        if (active && rank == cnt - 1) hrow[d] = base + cnt;
        __builtin_amdgcn_wave_barrier();
    }
}

__device__ inline void count_chunk(SaShared &sh, const int32_t *kin, uint32_t cbeg, uint32_t cend, int shift,
                                   int mask, int w, int lane) {
    int32_t *hrow = sh.hist[w];
    for (uint32_t j0 = cbeg; j0 < cend; j0 += 64) {
        const uint32_t j = j0 + lane;
        const bool active = j < cend;
        const int key = active ? kin[j] : 0;
        hist_add(hrow, (key >> shift) & mask, active, lane);
    }
}

// Fallback for texts whose activity bitmap does not fit LDS: every round sorts ALL suffixes.
__global__ __launch_bounds__(SA_THREADS) void sa_build_full_kernel(const Seg *segs, const uint8_t *text, SaWork wk) {
    __shared__ SaShared sh;
    const Seg sg = segs[blockIdx.x];
    const uint8_t *T = text + sg.off;
    const uint32_t n = sg.n;
    const uint32_t N = n + 1;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;

    int32_t *keyA = wk.keyA + sg.off, *valA = wk.valA + sg.off;
    int32_t *keyB = wk.keyB + sg.off, *valB = wk.valB + sg.off;
    int32_t *rk = wk.rkA + sg.off, *rknew = wk.rkB + sg.off;
    uint8_t *flags = wk.flags + sg.off;
    int32_t *sa_out = wk.sa + sg.off, *isa_out = wk.isa + sg.off;

    if (N == 1) {
        if (tid == 0) {
            sa_out[0] = 0;
            isa_out[0] = 0;
        }
        return;
    }

    const uint32_t C = (((N + SA_WAVES - 1) / SA_WAVES) + 63u) & ~63u;  // per-wave chunk of the sequence
    const uint32_t cbeg = min((uint32_t)w * C, N), cend = min(cbeg + C, N);

    const int KB = 32 - __clz((int)(N - 1));  // bits of the largest rank key
    const int P = (KB + SA_MAXDB - 1) / SA_MAXDB;
    const int DB = (KB + P - 1) / P;

    // ---- initial order: by first symbol (0 = sentinel, byte + 1 otherwise), one 9-bit pass
    hist_clear(sh, tid);
    __syncthreads();
    for (uint32_t j0 = cbeg; j0 < cend; j0 += 64) {
        const uint32_t j = j0 + lane;
        const bool active = j < cend;
        int key = 0;
        if (active) {
            key = j < n ? (int)T[j] + 1 : 0;
            keyA[j] = key;
            valA[j] = (int)j;
        }
        hist_add(sh.hist[w], key, active, lane);
    }
    hist_scan(sh, tid, SA_BINS);
    scatter_chunk(sh, keyA, valA, keyB, valB, cbeg, cend, 0, SA_BINS - 1, w, lane);
    __syncthreads();
    int32_t *kcur = keyB, *vcur = valB;  // sorted keys / current suffix order

    uint32_t h = 0;  // sorted depth; 0 = "rank by the key alone"
    for (;;) {
        // ---- ranks of the new order: RK'[p] = first index of the group of equal (key, second key)
        {
            int lastf = -1, cntf = 0;
            int c1 = 0, c2 = 0;
            for (uint32_t j0 = cbeg; j0 < cend; j0 += 64) {
                const uint32_t j = j0 + lane;
                const bool active = j < cend;
                int p = 0, k1 = -2, k2 = -1;
                if (active) {
                    p = vcur[j];
                    k1 = kcur[j];
                    if (h != 0 && (uint32_t)p + h < N) k2 = rk[(uint32_t)p + h];
                }
                int q1 = __shfl_up(k1, 1), q2 = __shfl_up(k2, 1);
                if (lane == 0) {
                    if (j0 == cbeg) {
                        if (j > 0) {
                            const int pp = vcur[j - 1];
                            q1 = kcur[j - 1];
                            q2 = (h != 0 && (uint32_t)pp + h < N) ? rk[(uint32_t)pp + h] : -1;
                        } else {
                            q1 = -3;
                        }
                    } else {
                        q1 = c1;
                        q2 = c2;
                    }
                }
                const bool flag = active && (k1 != q1 || k2 != q2);
                c1 = __shfl(k1, 63);
                c2 = __shfl(k2, 63);
                if (active) flags[j] = flag ? 1 : 0;
                const uint64_t fb = __ballot(flag);
                cntf += __popcll(fb);
                if (fb) lastf = (int)j0 + 63 - __clzll(fb);
            }
            if (lane == 0) {
                sh.wlast[w] = lastf;
                sh.wcount[w] = cntf;
            }
            __syncthreads();
            int carry = -1, total = 0;
            for (int ww = 0; ww < SA_WAVES; ++ww) {
                if (ww < w) carry = max(carry, sh.wlast[ww]);
                total += sh.wcount[ww];
            }
            for (uint32_t j0 = cbeg; j0 < cend; j0 += 64) {
                const uint32_t j = j0 + lane;
                const bool active = j < cend;
                int g = -1, p = 0;
                if (active) {
                    p = vcur[j];
                    if (flags[j]) g = (int)j;
                }
                g = wave_incl_max(g, lane);
                g = max(g, carry);
                carry = __shfl(g, 63);
                if (active) rknew[p] = g;
            }
            __syncthreads();
            int32_t *t = rk;
            rk = rknew;
            rknew = t;
            if ((uint32_t)total == N || h >= N) break;  // every suffix alone in its group: done (h >= N cannot happen)
        }
        h = (h == 0) ? 1 : h * 2;

        // ---- sequence of this round into (keyA, valA), fused with the count of digit 0
        hist_clear(sh, tid);
        __syncthreads();
        const int mask = (1 << DB) - 1;
        for (uint32_t j0 = cbeg; j0 < cend; j0 += 64) {
            const uint32_t j = j0 + lane;
            const bool active = j < cend;
            int key = 0;
            if (active) {
                const uint32_t s = (uint32_t)vcur[j];
                const uint32_t p = s >= h ? s - h : N - h + s;
                key = rk[p];
                keyA[j] = key;
                valA[j] = (int)p;
            }
            hist_add(sh.hist[w], key & mask, active, lane);
        }
        // ---- P stable passes, least significant digit first
        int32_t *kin = keyA, *vin = valA, *kout = keyB, *vout = valB;
        for (int pass = 0; pass < P; ++pass) {
            if (pass > 0) {
                hist_clear(sh, tid);
                __syncthreads();
                count_chunk(sh, kin, cbeg, cend, pass * DB, mask, w, lane);
            }
            hist_scan(sh, tid, 1 << DB);
            scatter_chunk(sh, kin, vin, kout, vout, cbeg, cend, pass * DB, mask, w, lane);
            __syncthreads();
            int32_t *t;
            t = kin; kin = kout; kout = t;
            t = vin; vin = vout; vout = t;
        }
        kcur = kin;
        vcur = vin;
    }

    // ---- final order and ranks (rk holds the exact rank of every suffix now)
    for (uint32_t j0 = cbeg; j0 < cend; j0 += 64) {
        const uint32_t j = j0 + lane;
        if (j < cend) {
            const int p = vcur[j];
            sa_out[j] = p;
            isa_out[p] = (int)j;
        }
    }
}


// ---- main kernel: rounds touch only the suffixes that still share their group ----------------------------
// A bitmap in LDS (one bit per suffix) marks the suffixes whose group has more than one member.  Per round:
//   A  one coalesced sweep over the current order keeps the marked members of the induction sequence
//      (p = SA[j] - h) in wave-private compact lists;
//   B  the lists are concatenated into a dense array with their rank keys (one gather per ACTIVE suffix);
//   C  the same stable LSD radix passes as the fallback, over the dense array only;
//   D  boundaries of (old group, second key) give every active suffix its new group start and its slot in
//      the suffix array (old group start + index inside the group); suffixes now alone leave the bitmap.
// Suffixes that are alone in their group never move again, so SA and the ranks are updated in place.
constexpr uint32_t SA_BITMAP_MAX_N = 300000;
constexpr int SA_INIT_SYMS = 4;  // depth of the initial sort (symbols compared straight from the text)

// the first SA_INIT_SYMS symbols of suffix p as two integers (high half, low half), sentinel-aware
__device__ inline void init_keys(const uint8_t *T, uint32_t n, uint32_t p, int &k1, int &k2) {
    uint32_t s[SA_INIT_SYMS];
#pragma unroll
    for (int t = 0; t < SA_INIT_SYMS; ++t) s[t] = (p + t < n) ? (uint32_t)T[p + t] + 1u : 0u;
    uint32_t hi = 0, lo = 0;
#pragma unroll
    for (int t = 0; t < SA_INIT_SYMS / 2; ++t) hi = (hi << 9) | s[t];
#pragma unroll
    for (int t = SA_INIT_SYMS / 2; t < SA_INIT_SYMS; ++t) lo = (lo << 9) | s[t];
    k1 = (int)hi;
    k2 = (int)lo;
}  // bitmap words live in dynamic LDS

struct SaShared2 {
    SaShared s;
    int32_t wbase[SA_WAVES + 1];
    int32_t wlastS[SA_WAVES];
};

__global__ __launch_bounds__(SA_THREADS) void sa_build_kernel(const Seg *segs, const uint8_t *text, SaWork wk, int max_rounds) {
    __shared__ SaShared2 sh2;
    extern __shared__ uint32_t bits[];
    SaShared &sh = sh2.s;
    const Seg sg = segs[blockIdx.x];
    const uint8_t *T = text + sg.off;
    const uint32_t n = sg.n;
    const uint32_t N = n + 1;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;

    int32_t *keyA = wk.keyA + sg.off, *valA = wk.valA + sg.off;
    int32_t *keyB = wk.keyB + sg.off, *valB = wk.valB + sg.off;
    uint8_t *flags = wk.flags + sg.off;
    int32_t *sa = wk.sa + sg.off, *rk = wk.isa + sg.off;

    if (N == 1) {
        if (tid == 0) {
            sa[0] = 0;
            rk[0] = 0;
        }
        return;
    }
    const uint32_t C = (((N + SA_WAVES - 1) / SA_WAVES) + 63u) & ~63u;
    const uint32_t cbeg = min((uint32_t)w * C, N), cend = min(cbeg + C, N);
    const int KB = 32 - __clz((int)(N - 1));
    const int P = (KB + SA_MAXDB - 1) / SA_MAXDB;
    const int DB = (KB + P - 1) / P;
    const int mask = (1 << DB) - 1;

    for (uint32_t x = tid; x < (N + 31) / 32; x += SA_THREADS) bits[x] = 0;

    // ---- initial order by the first SA_INIT_SYMS symbols (0 = sentinel, byte + 1 otherwise): LSD passes
    //      whose digit is read from the text, so only the suffix numbers move (4 bytes per element)
    {
        const int32_t *vin = nullptr;  // pass 0 reads the identity
        int32_t *vout = valA;
        for (int pass = 0; pass < SA_INIT_SYMS; ++pass) {
            const uint32_t d = (uint32_t)(SA_INIT_SYMS - 1 - pass);  // symbol index used by this pass
            int32_t *hrow = sh.hist[w];
            hist_clear(sh, tid);
            __syncthreads();
            for (uint32_t j0 = cbeg; j0 < cend; j0 += 64) {
                const uint32_t j = j0 + lane;
                const bool active = j < cend;
                int key = 0;
                if (active) {
                    const uint32_t q = (vin ? (uint32_t)vin[j] : j) + d;
                    key = q < n ? (int)T[q] + 1 : 0;
                }
                hist_add(hrow, key, active, lane);
            }
            hist_scan(sh, tid, SA_BINS);
            for (uint32_t j0 = cbeg; j0 < cend; j0 += 64) {
                const uint32_t j = j0 + lane;
                const bool active = j < cend;
                int key = 0, val = 0;
                if (active) {
                    val = vin ? vin[j] : (int)j;
                    const uint32_t q = (uint32_t)val + d;
                    key = q < n ? (int)T[q] + 1 : 0;
                }
                const uint64_t amask = __ballot(active);
                const uint64_t peers = match_digit(key, amask);
                const int cnt = __popcll(peers);
                const int rank = __popcll(peers & lanemask_lt(lane));
                const int base = hrow[key];
                if (active) vout[base + rank] = val;
                __builtin_amdgcn_wave_barrier();
                if (active && rank == cnt - 1) hrow[key] = base + cnt;
                __builtin_amdgcn_wave_barrier();
            }
            __syncthreads();
            vin = vout;
            vout = (vout == valA) ? valB : valA;
        }
    }
    const int32_t *kcur = keyB, *vcur = (SA_INIT_SYMS & 1) ? valA : valB;
    uint32_t m = N;                            // its length
    uint32_t dbeg = cbeg, dend = cend;         // this wave's chunk of it
    uint32_t h = 0;
    bool initial = true;
    for (;;) {
        // ---- D: boundaries, new group starts, slots
        {
            int lastK = -1, lastS = -1;
            int c1 = 0, c2 = 0;
            for (uint32_t u0 = dbeg; u0 < dend; u0 += 64) {
                const uint32_t u = u0 + lane;
                const bool act = u < dend;
                int p = 0, k1 = -2, k2 = -1;
                if (act) {
                    p = vcur[u];
                    if (initial) {
                        init_keys(T, n, (uint32_t)p, k1, k2);
                    } else {
                        k1 = kcur[u];
                        if ((uint32_t)p + h < N) k2 = rk[(uint32_t)p + h];
                    }
                }
                int q1 = __shfl_up(k1, 1), q2 = __shfl_up(k2, 1);
                if (lane == 0) {
                    if (u0 == dbeg) {
                        if (u > 0) {
                            const int pp = vcur[u - 1];
                            if (initial) {
                                init_keys(T, n, (uint32_t)pp, q1, q2);
                            } else {
                                q1 = kcur[u - 1];
                                q2 = ((uint32_t)pp + h < N) ? rk[(uint32_t)pp + h] : -1;
                            }
                        } else {
                            q1 = -3;
                        }
                    } else {
                        q1 = c1;
                        q2 = c2;
                    }
                }
                const bool fK = act && k1 != q1;
                const bool fS = act && (fK || k2 != q2);
                c1 = __shfl(k1, 63);
                c2 = __shfl(k2, 63);
                if (act) flags[u] = (uint8_t)((fK ? 1 : 0) | (fS ? 2 : 0));
                const uint64_t bK = __ballot(fK), bS = __ballot(fS);
                if (bK) lastK = (int)u0 + 63 - __clzll(bK);
                if (bS) lastS = (int)u0 + 63 - __clzll(bS);
            }
            if (lane == 0) {
                sh.wlast[w] = lastK;
                sh2.wlastS[w] = lastS;
            }
            __syncthreads();
            int carryK = -1, carryS = -1;
            for (int ww = 0; ww < w; ++ww) {
                carryK = max(carryK, sh.wlast[ww]);
                carryS = max(carryS, sh2.wlastS[ww]);
            }
            int rem = 0;
            for (uint32_t u0 = dbeg; u0 < dend; u0 += 64) {
                const uint32_t u = u0 + lane;
                const bool act = u < dend;
                int f = 0, fn = 2, p = 0, k1 = 0;
                if (act) {
                    f = flags[u];
                    if (u + 1 < m) fn = flags[u + 1];
                    p = vcur[u];
                    if (!initial) k1 = kcur[u];
                }
                int gK = wave_incl_max((f & 1) ? (int)u : -1, lane);
                int gS = wave_incl_max((f & 2) ? (int)u : -1, lane);
                gK = max(gK, carryK);
                gS = max(gS, carryS);
                carryK = __shfl(gK, 63);
                carryS = __shfl(gS, 63);
                const bool single = (f & 2) && (fn & 2);
                if (act) {
                    const int slot = initial ? (int)u : k1 + ((int)u - gK);
                    const int nr = initial ? gS : k1 + (gS - gK);
                    sa[slot] = p;
                    rk[p] = nr;
                    if (initial) {
                        if (!single) atomicOr(&bits[(uint32_t)p >> 5], 1u << (p & 31));
                    } else if (single) {
                        atomicAnd(&bits[(uint32_t)p >> 5], ~(1u << (p & 31)));
                    }
                }
                rem += __popcll(__ballot(act && !single));
            }
            if (lane == 0) sh.wcount[w] = rem;
            __syncthreads();
            int remaining = 0;
            for (int ww = 0; ww < SA_WAVES; ++ww) remaining += sh.wcount[ww];
            if (remaining == 0 || h >= N || --max_rounds < 0) break;
        }
        initial = false;
        h = (h == 0) ? (uint32_t)SA_INIT_SYMS : h * 2;

        // ---- A: the marked members of the sequence p = SA[j] - h, compacted per wave into valB[cbeg..]
        {
            uint32_t wrun = 0;
            for (uint32_t j0 = cbeg; j0 < cend; j0 += 64) {
                const uint32_t j = j0 + lane;
                bool a = false;
                uint32_t p = 0;
                if (j < cend) {
                    const uint32_t s = (uint32_t)sa[j];
                    p = s >= h ? s - h : N - h + s;
                    a = (bits[p >> 5] >> (p & 31)) & 1u;
                }
                const uint64_t bal = __ballot(a);
                if (a) valB[cbeg + wrun + (uint32_t)__popcll(bal & lanemask_lt(lane))] = (int)p;
                wrun += (uint32_t)__popcll(bal);
            }
            if (lane == 0) sh.wcount[w] = (int)wrun;
            __syncthreads();
            if (tid == 0) {
                int acc = 0;
                for (int ww = 0; ww < SA_WAVES; ++ww) {
                    sh2.wbase[ww] = acc;
                    acc += sh.wcount[ww];
                }
                sh2.wbase[SA_WAVES] = acc;
            }
            hist_clear(sh, tid);
            __syncthreads();
            m = (uint32_t)sh2.wbase[SA_WAVES];
        }
        const uint32_t Cm = (((m + SA_WAVES - 1) / SA_WAVES) + 63u) & ~63u;
        dbeg = min((uint32_t)w * Cm, m);
        dend = min(dbeg + Cm, m);

        // ---- B: dense (rank key, suffix) array in (keyA, valA), fused with the count of digit 0
        for (uint32_t u0 = dbeg; u0 < dend; u0 += 64) {
            const uint32_t u = u0 + lane;
            const bool act = u < dend;
            int key = 0;
            if (act) {
                int sw = 0;  // source wave: wbase[sw] <= u < wbase[sw + 1]
#pragma unroll
                for (int ww = 1; ww < SA_WAVES; ++ww) sw += ((uint32_t)sh2.wbase[ww] <= u) ? 1 : 0;
                const uint32_t sbeg = min((uint32_t)sw * C, N);
                const int p = valB[sbeg + (u - (uint32_t)sh2.wbase[sw])];
                key = rk[p];
                keyA[u] = key;
                valA[u] = p;
            }
            hist_add(sh.hist[w], key & mask, act, lane);
        }
        // ---- C: P stable passes over the dense array
        int32_t *kin = keyA, *vin = valA, *kout = keyB, *vout = valB;
        for (int pass = 0; pass < P; ++pass) {
            if (pass > 0) {
                hist_clear(sh, tid);
                __syncthreads();
                count_chunk(sh, kin, dbeg, dend, pass * DB, mask, w, lane);
            }
            hist_scan(sh, tid, 1 << DB);
            scatter_chunk(sh, kin, vin, kout, vout, dbeg, dend, pass * DB, mask, w, lane);
            __syncthreads();
            int32_t *t;
            t = kin; kin = kout; kout = t;
            t = vin; vin = vout; vout = t;
        }
        kcur = kin;
        vcur = vin;
    }
}

// ---- window index: ranks regrouped by the 16 KiB chunk of their position (stable in rank order) ------
// Lets the match kernel answer "first rank >= lo whose position lies in [i - 2^14, i)" without walking
// the suffix array: the window meets at most two chunks, and inside a chunk the condition is one-sided.
constexpr int WIN_SHIFT = LZSS_SHORT_ADDR_BITS;  // chunk = 2^14 positions

__global__ __launch_bounds__(SA_THREADS) void window_index_kernel(const Seg *segs, BatchIndex ix) {
    __shared__ SaShared sh;
    const Seg sg = segs[blockIdx.x];
    const uint32_t n = sg.n;
    if (n == 0) return;
    const uint32_t N = n + 1;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int32_t *SA = ix.sa + sg.off;
    int32_t *KS = ix.ks + sg.off, *PS = ix.ps + sg.off;
    const int nch = (int)((n + (1u << WIN_SHIFT) - 1) >> WIN_SHIFT);  // <= 256 for n <= 4 MiB

    // ranks 1..n (rank 0 is the sentinel), one chunk of the rank range per wave
    const uint32_t C = (((n + SA_WAVES - 1) / SA_WAVES) + 63u) & ~63u;
    const uint32_t cbeg = 1 + min((uint32_t)w * C, n), cend = min(cbeg + C, N);

    hist_clear(sh, tid);
    __syncthreads();
    for (uint32_t k0 = cbeg; k0 < cend; k0 += 64) {
        const uint32_t k = k0 + lane;
        const bool active = k < cend;
        const int c = active ? (SA[k] >> WIN_SHIFT) : 0;
        hist_add(sh.hist[w], c, active, lane);
    }
    __syncthreads();
    if (tid < nch) {  // first slot of wave w inside chunk tid
        int s = tid << WIN_SHIFT;
        for (int ww = 0; ww < SA_WAVES; ++ww) {
            const int t = sh.hist[ww][tid];
            sh.hist[ww][tid] = s;
            s += t;
        }
    }
    __syncthreads();
    int32_t *hrow = sh.hist[w];
    for (uint32_t k0 = cbeg; k0 < cend; k0 += 64) {
        const uint32_t k = k0 + lane;
        const bool active = k < cend;
        const int p = active ? SA[k] : 0;
        const int c = p >> WIN_SHIFT;
        const uint64_t amask = __ballot(active);
        const uint64_t peers = match_digit(c, amask);
        const int cnt = __popcll(peers);
        const int rank = __popcll(peers & lanemask_lt(lane));
        const int base = hrow[c];
        if (active) {
            KS[base + rank] = (int)k;
            PS[base + rank] = p;
        }
        __builtin_amdgcn_wave_barrier();
        if (active && rank == cnt - 1) hrow[c] = base + cnt;
        __builtin_amdgcn_wave_barrier();
    }
}

// min / max over blocks of 64 entries of a per-segment int32 array with `count(seg)` entries
template <bool IS_MAX>
__global__ __launch_bounds__(256) void block64_kernel(const Seg *segs, const int32_t *in, int32_t *out, int level, int extra) {
    const Seg sg = segs[blockIdx.y];
    // level 0: entries of the array itself ([off + j], count n + extra); level 1: the level-0 summaries
    uint32_t count = sg.n + (uint32_t)extra;
    const int32_t *src = in + sg.off;
    if (level == 1) {
        count = (count + 63) / 64;
        src = in + sg.off / 64;
    }
    if (blockIdx.x * 256 >= count) return;
    const uint32_t j = blockIdx.x * 256 + threadIdx.x;
    int v = IS_MAX ? INT_MIN : INT_MAX;
    if (j < count) v = src[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int t = __shfl_xor(v, o);
        v = IS_MAX ? (t > v ? t : v) : (t < v ? t : v);
    }
    if ((threadIdx.x & 63) == 0) out[sg.off / 64 + (j >> 6)] = v;
}

void launch_window_index(const Seg *d_segs, uint32_t n_segs, uint32_t max_n, const BatchIndex &ix, hipStream_t stream) {
    if (n_segs == 0 || max_n == 0) return;
    hipLaunchKernelGGL(window_index_kernel, dim3(n_segs), dim3(SA_THREADS), 0, stream, d_segs, ix);
    const uint32_t c0 = max_n + 1, c1 = (c0 + 63) / 64;
    dim3 g0((c0 + 255) / 256, n_segs), g1((c1 + 255) / 256, n_segs);
    hipLaunchKernelGGL(block64_kernel<false>, g0, dim3(256), 0, stream, d_segs, ix.ps, ix.pmin1, 0, 0);
    hipLaunchKernelGGL(block64_kernel<true>, g0, dim3(256), 0, stream, d_segs, ix.ps, ix.pmax1, 0, 0);
    hipLaunchKernelGGL(block64_kernel<false>, g0, dim3(256), 0, stream, d_segs, ix.sa, ix.samin1, 0, 1);
    hipLaunchKernelGGL(block64_kernel<false>, g1, dim3(256), 0, stream, d_segs, ix.pmin1, ix.pmin2, 1, 0);
    hipLaunchKernelGGL(block64_kernel<true>, g1, dim3(256), 0, stream, d_segs, ix.pmax1, ix.pmax2, 1, 0);
    hipLaunchKernelGGL(block64_kernel<false>, g1, dim3(256), 0, stream, d_segs, ix.samin1, ix.samin2, 1, 1);
}

void launch_sa_build(const Seg *d_segs, uint32_t n_segs, uint32_t max_n, const uint8_t *d_text, const SaWork &wk,
                     hipStream_t stream) {
    if (n_segs == 0) return;
    if (max_n + 1 <= SA_BITMAP_MAX_N) {
        const size_t lds = ((size_t)max_n + 1 + 31) / 32 * 4;
        static const int max_rounds = getenv("MI355_LZSS_SA_ROUNDS") ? atoi(getenv("MI355_LZSS_SA_ROUNDS")) : 1000;  // timing experiments only
        hipLaunchKernelGGL(sa_build_kernel, dim3(n_segs), dim3(SA_THREADS), lds, stream, d_segs, d_text, wk, max_rounds);
    } else {
        hipLaunchKernelGGL(sa_build_full_kernel, dim3(n_segs), dim3(SA_THREADS), 0, stream, d_segs, d_text, wk);
    }
}

}  // namespace lzss
