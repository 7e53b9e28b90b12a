// sa_build.hip - suffix array + inverse of every segment of a batch, one 512-thread workgroup per segment, two
// workgroups per CU (gfx950); texts of more than 2^19 suffixes go to sa_large.hip.  Replaces suffixarray.New
// (lzss/internal/suffixarray/suffixarray.go:32-45, which runs SA-IS, sais.go:128-203) for both the input index
// (lzss/compress.go:116) and the dictionary index (compress.go:64).  The suffix array of a text is a unique object, so
// the construction is free to be GPU-shaped: prefix doubling (Manber-Myers bucket induction), every round one stable
// LSD radix sort of the rank keys, stability from wave-level peer matching, wave-private histograms in LDS.
//
// N-world: the text gets an explicit smallest sentinel at index n, N = n + 1 suffixes, SA[0] == n.
//
// Round with sorted depth h (suffixes are ordered by their first h symbols, RK[p] = first index of
// p's group): walk the current order j = 0..N-1 and emit p = SA[j] - h; a stable sort of that
// sequence by RK[p] orders every group by its second half.  Slots with SA[j] < h stand for the h
// suffixes p >= N - h, whose first h symbols contain the sentinel and which are therefore already
// alone in their group, so their place in the sequence does not matter: p = N - h + SA[j] keeps the
// sequence a permutation of 0..N-1.
#include <atomic>
#include <cstdlib>

#include "kernels.h"

namespace lzss {

constexpr int SA_THREADS = 1024;         // window_index_kernel (the suffix sort proper: SAM_THREADS below)
constexpr int SA_WAVES = SA_THREADS / 64;
constexpr int SA_BINS = 512;             // chunks of 2^14 positions a text of MaxInputSize has at most: 256

struct SaShared {
    int32_t hist[SA_WAVES][SA_BINS];
    int32_t colsum[SA_BINS];
    int32_t wlast[SA_WAVES];
    int32_t wcount[SA_WAVES];
};

__device__ inline uint64_t lanemask_lt(int lane) { return (1ull << lane) - 1ull; }

// lanes of `amask` holding the same digit as this lane; the digit has `nbits` bits (wave-uniform)
__device__ inline uint64_t match_digit_n(int d, uint64_t amask, int nbits) {
    uint64_t peers = amask;
    for (int bit = 0; bit < nbits; ++bit) {
        const bool b = (d >> bit) & 1;
        const uint64_t bal = __ballot(b);
        peers &= b ? bal : ~bal;
    }
    return peers;
}


__device__ inline void hist_clear(SaShared &sh, int tid) {
    for (int x = tid; x < SA_WAVES * SA_BINS; x += SA_THREADS) (&sh.hist[0][0])[x] = 0;
}

// ---- main kernel --------------------------------------------------------------------------------------
// Rounds touch only the suffixes that still share their group, and every sort pass is a WORKGROUP-wide
// stable scatter over tiles of SA_TILE sequence entries (the waves rank their 64-entry steps against
// per-wave LDS counters, one scan per tile turns them into offsets behind running per-bin bases), so a
// pass keeps one output stream per bin and needs no counting sweep of its own: the bin totals are known
// beforehand.
//
//   key sweep:     run-aware 36-bit keys (first symbol, run code, two symbols after the run) as records
//                  (key << 20 | suffix) in text order, the run lengths (rlen) and the counts of the four digits
//   initial order: SA_INIT_DIGITS LSD passes over the records, least significant digit first
//   regroup (D):   boundaries of (group, second key) in the sorted records give every active suffix its slot
//                  in the suffix array and its new group start; suffixes now alone leave the `act` bitmap;
//                  the digits of the new group starts are counted for the next round's passes
//   S1:            one sweep over the suffix array: the active members of the induction sequence
//                  p = SA[j] - h, with their group start k1 = RK[p] (the one gather of a round) and the group
//                  start k2 of slot j (the top bit of SA[j] marks the first slot of a shared group), packed
//                  into 8-byte records and scattered by the low digit of k1
//   S2:            scatter of the records by the high digit of k1 -> sorted by (k1, sequence order)
//   late rounds:   when few suffixes are left, the records (RK[p], RK[p + h], p) come straight from the `act`
//                  bitmap and are sorted by both keys (four passes over m records, no sweep over N slots)
// Suffixes alone in their group never move again, so SA and the ranks are updated in place.
#ifndef SA_KEY_DIGITS
#define SA_KEY_DIGITS 4
#endif
// 9-bit digits of the initial keys: (c, run code, x, y); optionally without y (3), or - for texts of up to 2^18 suffixes,
// whose index leaves room for a fifth digit in the 8-byte record - with z, the third symbol after the run (5).  Measured
// with the small-group finisher in place (sa_finish): 4 digits 85.9 ms, 5 digits 89.9 ms per 10 000 blobs.
template <int DB> struct SaInit {
    static constexpr int DIGITS = (SA_KEY_DIGITS == 5 && DB > 9) ? 4 : SA_KEY_DIGITS;
    static constexpr bool Z = DIGITS == 5;
    static constexpr int SHIFT = Z ? 19 : 20;          // record = key << SHIFT | suffix
    static constexpr uint32_t H0 = DIGITS - 1;         // symbols every key group shares at least
};
#ifndef SA_MAIN_THREADS
#define SA_MAIN_THREADS 512
#endif
#ifndef SA_MAIN_K
#define SA_MAIN_K 8
#endif
#ifndef SA_USE_MTAB
#define SA_USE_MTAB 1
#endif
#ifdef SA_PROFILE  // per-phase clocks of workgroup 0 (debug builds only)
__device__ long long g_rg[8];
#define RG(i) { if (INITIAL) { __builtin_amdgcn_s_waitcnt(0); const long long t_ = clock64(); if (blockIdx.x == 0 && tid == 0) g_rg[i] += t_ - rg_last; rg_last = t_; } }
#define RG_DECL long long rg_last = clock64();
#define SA_PROF_DECL long long prof_acc[24] = {}; long long prof_last = clock64();
#define SA_PROF(i) { __builtin_amdgcn_s_waitcnt(0); const long long t_ = clock64(); prof_acc[i] += t_ - prof_last; prof_last = t_; }
#define SA_PROF_PRINT if (blockIdx.x == 0 && (threadIdx.x & 63) == 0 && threadIdx.x / 64 < 2) { printf("w%d init: ld %lld rank %lld prep %lld put %lld | regI %lld | S1: ld %lld gath %lld rank %lld prep %lld put %lld | S2: ld %lld rank %lld prep %lld put %lld | D %lld | other %lld | late build %lld passes %lld | finish init %lld rounds %lld | setup %lld sweep %lld\n", threadIdx.x / 64, prof_acc[0], prof_acc[1], prof_acc[2], prof_acc[3], prof_acc[4], prof_acc[5], prof_acc[6], prof_acc[7], prof_acc[8], prof_acc[9], prof_acc[10], prof_acc[11], prof_acc[12], prof_acc[13], prof_acc[14], prof_acc[15], prof_acc[16], prof_acc[17], prof_acc[19], prof_acc[18], prof_acc[20], prof_acc[21]); if (threadIdx.x == 0) printf("regI: loop %lld load %lld flags %lld barrier %lld lists %lld write %lld\n", g_rg[0], g_rg[1], g_rg[2], g_rg[3], g_rg[4], g_rg[5]); }
#else
#define RG(i)
#define RG_DECL
#define SA_PROF_DECL
#define SA_PROF(i)
#define SA_PROF_PRINT
#endif
#ifndef SA_MTAB_BITS
#define SA_MTAB_BITS 10
#endif
constexpr int SA_MTB = SA_MTAB_BITS;     // digit bits resolved by the table (the rest by ballots)
constexpr int SA_MTB_WIDE = 8;           // the same for the 10-bit-digit instantiation
#ifndef SA_WGS_PER_CU
#define SA_WGS_PER_CU 2   // (3 with SA_MAIN_K 4 and SA_MTAB_BITS 7: 80 VGPRs; slower once the finisher was in: 85.9 vs 81.9 ms)
#endif
constexpr bool SA_MTAB = SA_USE_MTAB;    // lanes with equal digits find each other through a bit table in LDS
constexpr int SAM_THREADS = SA_MAIN_THREADS;
constexpr int SAM_WAVES = SAM_THREADS / 64;
constexpr int SA_K = SA_MAIN_K;  // sequence entries per thread and tile
constexpr int SA_WSPAN = 64 * SA_K;            // consecutive entries of one wave inside a tile
constexpr int SA_TILE = SAM_THREADS * SA_K;

#ifndef SA_SWEEP_AHEAD
#define SA_SWEEP_AHEAD 2   // steps between the loads behind a block's runs and the block's records (key sweep)
#endif
#ifndef SA_PEER_ROUNDS
#define SA_PEER_ROUNDS 1    // leader votes of wave_peers before the bit table
#endif
#ifndef SA_COUNT_ROUNDS
#define SA_COUNT_ROUNDS 1   // leader votes of digit_count before the per-lane atomics
#endif
#ifndef SA_ISA_BINNED
#define SA_ISA_BINNED 1
#endif
// The initial regroup knows the rank of every suffix in suffix order and has to store it in position order.  Written
// directly that is one random 4-byte store per suffix, and a random store costs a whole 32-byte sector of HBM write
// (a third of the kernel's WRITE_SIZE).  Instead the (rank, position) pairs are first appended to one stream per chunk
// of 8192 positions (4 bytes per pair: rank << 13 | position inside the chunk; a chunk receives exactly 8192 pairs, so
// the streams need no counting pass), and every chunk is then spread out in LDS and written as one 32 KiB run.
constexpr int SA_ISA_CH = 13;
#ifndef SA_LATE_DIVISOR
#define SA_LATE_DIVISOR 4
#endif
constexpr uint32_t SA_LATE_DIV = SA_LATE_DIVISOR;  // rounds with at most N / this active suffixes skip the sweep
constexpr int SA_RUN_CAP = 254;  // runs of equal symbols are told apart by length up to here
// run-compressed sort (sa_meta_kernel): item i of a text starts at byte mpos[i] & MP_POS_MASK; bits 20-27 hold that
// byte, bit 28 says that the item is one byte of a run with at least SA_RUN_CAP bytes to go ("deep")
constexpr uint32_t MP_POS_MASK = (1u << 20) - 1u;
constexpr int MP_BYTE_SHIFT = 20, MP_DEEP_SHIFT = 28, MP_CONT_SHIFT = 29;  // (bit 29: the item is not the first byte of its run)

// Length of the run of equal bytes that starts at position e of the text (1 at and past its end), exact up to 257:
// the lanes of a wave hold 64 consecutive positions and read the run off five ballots of "byte == next byte".
__device__ inline int run_length_64(const uint8_t *T, uint32_t n, uint32_t e, int lane) {
    uint64_t E[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
        const uint32_t q = e + 64 * t;
        E[t] = __ballot(q + 1 < n && T[q] == T[q + 1]);
    }
    const uint64_t x0 = E[0] >> lane;  // the top `lane` bits are zero
    int rl = (x0 == ~0ull) ? 64 : (int)__builtin_ctzll(~x0);  // ones from this lane's bit on, <= 64 - lane
    if (rl == 64 - lane) {
#pragma unroll
        for (int t = 1; t < 5; ++t) {
            const int rt = E[t] == ~0ull ? 64 : (int)__builtin_ctzll(~E[t]);
            rl += rt;
            if (rt < 64) break;
        }
    }
    return rl + 1;
}

// run lengths for the texts the fallback kernel sorts (the main kernel writes them in its first sweep)
__global__ __launch_bounds__(256) void rlen_kernel(const Seg *segs, const uint8_t *text, uint8_t *rlen) {
    const Seg sg = segs[blockIdx.y];
    if (blockIdx.x * 256 > sg.n) return;
    const uint32_t e = blockIdx.x * 256 + threadIdx.x;
    const int rl = run_length_64(text + sg.off, sg.n, e, threadIdx.x & 63);
    if (e <= sg.n) rlen[sg.off + e] = (uint8_t)(rl < 255 ? rl : 255);
}

template <int DB>
__device__ inline uint64_t match_bits(int d, uint64_t amask) {
    uint64_t peers = amask;
#pragma unroll
    for (int bit = 0; bit < DB; ++bit) {
        const bool b = (d >> bit) & 1;
        const uint64_t bal = __ballot(b);
        peers &= b ? bal : ~bal;
    }
    return peers;
}

// bitmap words of a text of N suffixes (even; at least two digit histograms fit: they live there during the initial sort)
template <int DB>
__host__ __device__ inline uint32_t sa_bitmap_words(uint32_t N) {
    const uint32_t wds = ((N + 63) / 64) * 2;
    return wds < (2u << DB) ? (2u << DB) : wds;
}

template <int DB>
struct SaLds {
    static constexpr int NB = 1 << DB;
    // digit bits the peer table resolves: texts above 2^18 need a 64 KiB bitmap, their table is smaller
    static constexpr int MTB = DB <= 9 ? SA_MTB : (SA_MTB < SA_MTB_WIDE ? SA_MTB : SA_MTB_WIDE);
    static constexpr int mtab_entries() { return 1 << (DB < MTB ? DB : MTB); }
    uint32_t *act;     // bit p: suffix p still shares its group
    int32_t *base;     // [NB] running output index of bin b
    int32_t *delta;    // [NB] output index of the current tile's first entry of bin b
    int32_t *tpos;     // [NB] scratch: digit counts of a late round's second keys (low digit)
    int32_t *hl, *hh;  // [NB] digit counts of the next round's keys (low / high digit)
    int32_t *hy;       // [NB] digit counts of a late round's second keys (high digit; the low digit's sit in tpos)
    int32_t *hsec;     // [NB] sa_meta_kernel: entries per section of the expansion (items by their byte + 1)
    int32_t *wtot;     // [16] tile entries per wave of bin threads
    uint16_t *whist;   // [SAM_WAVES / 2][NB][2] entries of the current tile per bin and wave (zero between tiles): waves
                       // 2k and 2k + 1 share a 32-bit word, a row holds the NB bins, so a wave's lanes with different
                       // digits reach different banks (a [NB][SAM_WAVES] layout put them 16 bytes apart: 16 banks only)
    uint16_t *woff;    // the same layout: offset of wave w's entries inside the tile's bin b
    uint64_t *mtab;    // [SAM_WAVES][mtab_entries()] per wave: lanes holding (the low bits of) digit d (zero between steps)
    __device__ SaLds(uint32_t *mem, uint32_t words) {
        act = mem;
        base = (int32_t *)(act + words);
        delta = base + NB;
        tpos = delta + NB;
        hl = tpos + NB;
        hh = hl + NB;
        hy = hh + NB;
        hsec = hy + NB;
        wtot = hsec + NB;
        whist = (uint16_t *)(wtot + 16);
        woff = whist + SAM_WAVES * NB;
        mtab = (uint64_t *)(woff + SAM_WAVES * NB);
    }
    static __device__ inline int cell(int d, int w) { return ((((w >> 1) * NB) + d) << 1) | (w & 1); }
    // words: bitmap words (even)
    static size_t bytes(uint32_t words) {
        return (size_t)words * 4 + ((size_t)NB * 7 + 16) * 4 + (size_t)SAM_WAVES * NB * 4 +
               (SA_MTAB ? (size_t)SAM_WAVES * mtab_entries() * 8 : 0);
    }
};

struct alignas(16) SaCarry {   // (its size sets the offset of the dynamic LDS behind it, whose peer table holds 64-bit words)
    int32_t a[2][SAM_WAVES];
    int32_t b[2][SAM_WAVES];
    int32_t wcount[SAM_WAVES];
    int32_t fin_cnt[8];   // small groups listed for sa_finish, by size (2 .. 8 at [0 .. 6]); [7]: suffixes it has finished
    int32_t bin_cur[64];  // initial regroup: (position, rank) pairs written so far per chunk of 2^SA_ISA_CH positions
    int32_t next_sec;     // sa_meta_kernel: next section of the expansion a wave may take
};

// lanes of the wave (among `valid`) holding the same digit as this lane.  The most frequent digits are found
// by two leader votes; the rest meet in the wave's bit table (one 64-bit word per digit).
template <int RB, int MTB>
__device__ inline uint64_t wave_peers(uint64_t *mrow, int d, bool valid, int lane) {
    if (!SA_MTAB) return match_bits<RB>(d, __ballot(valid));
    uint64_t peers = 0;
    bool todo = valid;
#pragma unroll
    for (int r = 0; r < SA_PEER_ROUNDS; ++r) {
        const uint64_t rem = __ballot(todo);
        if (rem) {
            const int dl = lane_value(d, (int)__builtin_ctzll(rem));
            const bool eq = todo && d == dl;
            const uint64_t e = __ballot(eq);
            if (eq) {
                peers = e;
                todo = false;
            }
        }
    }
    if (__ballot(todo)) {
        // the LDS unit executes one wave's instructions in order, so the three accesses need no waits between
        // them; the empty asm statements only keep the compiler from reordering them
        constexpr int TB = RB < MTB ? RB : MTB;
        uint64_t *cell = mrow + (d & ((1 << TB) - 1));
        if (todo) atomicOr((unsigned long long *)cell, 1ull << lane);
        asm volatile("" ::: "memory");
        if (todo) peers = *(volatile uint64_t *)cell;
        asm volatile("" ::: "memory");
        if (todo) *cell = 0;
        asm volatile("" ::: "memory");
#pragma unroll
        for (int bit = TB; bit < RB; ++bit) {  // digit bits beyond the table: ballots
            const bool b = (d >> bit) & 1;
            const uint64_t bal = __ballot(b);
            peers &= b ? bal : ~bal;
        }
    }
    return peers;
}

// rank of this lane's entry among the entries of its wave's tile part with the same digit (stable)
template <int RB, int DB>
__device__ inline int tile_rank(const SaLds<DB> &L, int w, int d, bool valid, int lane) {
    uint16_t *cell = L.whist + SaLds<DB>::cell(d, w);
    const uint64_t peers = wave_peers<RB, SaLds<DB>::MTB>(L.mtab + w * SaLds<DB>::mtab_entries(), d, valid, lane);
    const int cnt = __popcll(peers);
    const int rank = __popcll(peers & lanemask_lt(lane));
    const int lh = valid ? (int)*cell : 0;
    __builtin_amdgcn_wave_barrier();
    if (valid && rank == cnt - 1) *cell = (uint16_t)(lh + cnt);
    __builtin_amdgcn_wave_barrier();
    return lh + rank;
}

// After every wave has ranked its part of the tile: per-wave counts -> offsets inside the tile's bins, and every bin's
// output position for this tile (the running bases advance).  Two barriers.
template <int DB>
__device__ inline void tile_prepare(const SaLds<DB> &L, int tid) {
    constexpr int NB = 1 << DB;
    constexpr int BPT = NB > SAM_THREADS ? NB / SAM_THREADS : 1;  // consecutive bins per bin thread
    constexpr int NBT = NB / BPT;
    __syncthreads();
    if (tid < NBT) {
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            const int b = tid * BPT + i;
            // the bin's counts of two waves per load, their running sums likewise
            uint32_t *hp = reinterpret_cast<uint32_t *>(L.whist) + b;
            uint32_t *op = reinterpret_cast<uint32_t *>(L.woff) + b;
            int r = 0;
#pragma unroll
            for (int wp = 0; wp < SAM_WAVES / 2; ++wp) {
                const uint32_t c2 = hp[wp * NB];
                const uint32_t o2 = (uint32_t)r | ((uint32_t)(r + (int)(c2 & 0xFFFFu)) << 16);
                r += (int)(c2 & 0xFFFFu) + (int)(c2 >> 16);
                hp[wp * NB] = 0u;
                op[wp * NB] = o2;
            }
            const int gs = L.base[b];
            L.delta[b] = gs;
            L.base[b] = gs + r;
        }
    }
    __syncthreads();
}

// one record of the tile to its place in the output: the running base of its bin (delta) + the records of the bin that
// precede it in the tile (staging tiles sorted in LDS so that the scatter writes whole runs was measured without gain
// and removed)
template <int DB>
__device__ inline void tile_put(const SaLds<DB> &L, uint64_t *out, int d, int w, int loc, uint64_t r) {
    out[L.delta[d] + (int)L.woff[SaLds<DB>::cell(d, w)] + loc] = r;
}

// histogram update of one 64-entry step: two leader votes take the most frequent digits, the rest adds itself
__device__ inline void digit_count(int32_t *hist, int d, bool valid, int lane) {
    bool todo = valid;
#pragma unroll
    for (int r = 0; r < SA_COUNT_ROUNDS; ++r) {
        const uint64_t rem = __ballot(todo);
        if (rem) {
            const int first = (int)__builtin_ctzll(rem);
            const int dl = lane_value(d, first);
            const bool eq = todo && d == dl;
            const uint64_t em = __ballot(eq);
            if (lane == first) atomicAdd(&hist[dl], __popcll(em));
            todo = todo && !eq;
        }
    }
    if (todo) atomicAdd(&hist[d], 1);
}

// base[b] = sum of src[0..b) (ends with a barrier; src must be complete and visible: barrier before)
__device__ inline void bins_scan(const int32_t *src, int32_t *base, int nb, int tid) {
    if (tid < 64) {
        int carry = 0;
        for (int b0 = 0; b0 < nb; b0 += 64) {
            const int v = src[b0 + tid];
            const int inc = wave_incl_add(v, tid);
            base[b0 + tid] = carry + inc - v;
            carry += lane_value(inc, 63);
        }
    }
    __syncthreads();
}

// one stable scatter pass over m records by the digit (rec >> sh) & (NB - 1); `hist` holds the digit counts
template <int DB>
__device__ inline void scatter_pass(const SaLds<DB> &L, const uint64_t *in, uint64_t *out, uint32_t m, int sh,
                                    const int32_t *hist, int tid) {
    constexpr int NB = 1 << DB;
    constexpr int MASK = NB - 1;
    const int lane = tid & 63, w = tid >> 6;
    bins_scan(hist, L.base, NB, tid);
    for (uint32_t t0 = 0; t0 < m; t0 += SA_TILE) {
        const uint32_t wb = t0 + (uint32_t)w * SA_WSPAN;
        uint64_t r[SA_K];
        int loc[SA_K];
#pragma unroll
        for (int s = 0; s < SA_K; ++s) {
            const uint32_t u = wb + s * 64 + lane;
            r[s] = (u < m) ? in[u] : 0;
        }
#pragma unroll
        for (int s = 0; s < SA_K; ++s) {
            const uint32_t u = wb + s * 64 + lane;
            loc[s] = tile_rank<DB, DB>(L, w, (int)(r[s] >> sh) & MASK, u < m, lane);
        }
        tile_prepare<DB>(L, tid);
#pragma unroll
        for (int s = 0; s < SA_K; ++s) {
            const uint32_t u = wb + s * 64 + lane;
            if (u < m) tile_put<DB>(L, out, (int)(r[s] >> sh) & MASK, w, loc[s], r[s]);
        }
    }
    __syncthreads();
}

// ---- small groups are finished by comparing their suffixes directly -------------------------------------------
// Real repeats in the data this path sees are long (32-byte words): after the initial keys most shared groups hold 2-4
// suffixes that stay together through the rounds with h = 4, 8, 16 and part at depth 32-64, and every such round costs
// them a rank gather, two scatter passes and a regroup.  The regroup step therefore LISTS the groups of 2 .. SA_FIN_G
// suffixes that lie inside one 64-entry step (one list per size, in the spare record buffer), and sa_finish then gives
// every listed group to one lane: it compares the members' suffixes pairwise from the depth they are known to share,
// up to SA_FIN_SPAN bytes further; if every pair is decided the members get their final slots and leave the active
// set (the digit counts the regroup step made for them are taken back), otherwise the group goes on through the
// rounds as before.  (Comparing inside the regroup sweep itself was 1.7x slower than no finishing at all: a step
// waited for its slowest lane's chain of dependent loads while most lanes had no group.)
#ifndef SA_FIN_G
#define SA_FIN_G 5   // 1: no finishing; at most 8 (measured: 4 82.0, 5 81.0, 6 81.0, 8 92.3 ms per 10 000 blobs)
#endif
static_assert(SA_FIN_G <= 6, "finish_group is instantiated for 2 .. 6 members");  // (entries that also carry the members' positions, so that sa_finish
// need not read them back from the suffix array, cost more in the regroup sweep than they saved: 83.3 vs 80.9 ms)
// first entry of the list of the groups of z suffixes: a text of N suffixes has at most N / z of them (all lists
// together, sizes 2 .. 8: 1.72 N four-byte entries - they fit the spare 8-byte-per-suffix record buffer)
__device__ inline uint32_t fin_list_base(int z, uint32_t N) {
    uint32_t b = 0;
    for (int y = 2; y < z; ++y) b += N / (uint32_t)y + 1;
    return b;
}
#ifndef SA_FIN_SPAN
#define SA_FIN_SPAN 128
#endif
// order of the suffixes at pa and pb (pa != pb), which share their first d0 symbols: -1 (pa's is smaller), +1, or 0 when
// they are still equal SA_FIN_SPAN bytes further.  The suffix that ends first is the smaller one (the sentinel).
__device__ inline int suffix_cmp(const uint8_t *T, uint32_t n, uint32_t pa, uint32_t pb, int d0) {
    const int lim = (int)(n - (pa > pb ? pa : pb));  // length of the shorter suffix
    const int cap = d0 + SA_FIN_SPAN;
    int o = d0;
#ifndef SA_FIN_WIDE
#define SA_FIN_WIDE 1
#endif
#if !SA_FIN_WIDE
    while (o < cap && o < lim) {
        const uint64_t x = ld8(T + pa + o) ^ ld8(T + pb + o);
        if (x) {
            o += (int)(__builtin_ctzll(x) >> 3);
            break;
        }
        o += 8;
    }
#else
    while (o < cap && o < lim) {  // 32 bytes per round trip (the text is readable 64 bytes past its end)
        const uint64_t x0 = ld8(T + pa + o) ^ ld8(T + pb + o), x1 = ld8(T + pa + o + 8) ^ ld8(T + pb + o + 8);
        const uint64_t x2 = ld8(T + pa + o + 16) ^ ld8(T + pb + o + 16), x3 = ld8(T + pa + o + 24) ^ ld8(T + pb + o + 24);
        if (x0 | x1 | x2 | x3) {
            if (x0) o += (int)(__builtin_ctzll(x0) >> 3);
            else if (x1) o += 8 + (int)(__builtin_ctzll(x1) >> 3);
            else if (x2) o += 16 + (int)(__builtin_ctzll(x2) >> 3);
            else o += 24 + (int)(__builtin_ctzll(x3) >> 3);
            break;
        }
        o += 32;
    }
#endif
    if (o >= lim) return pa > pb ? -1 : 1;
    const int ca = T[pa + o], cb = T[pb + o];  // (plain loads after the loop: see the note in match.hip, dict_search_in)
    if (ca == cb) return 0;                    // the loop ran into the cap
    return ca < cb ? -1 : 1;
}

// does the initial regroup store the ranks through the chunk streams?  (needs the staging area: woff + the peer table)
template <int DB>
constexpr bool isa_binned() {
    return SA_ISA_BINNED && ((size_t)SAM_WAVES * (1 << DB) * 2 + (SA_MTAB ? (size_t)SAM_WAVES * SaLds<DB>::mtab_entries() * 8 : 0)) >= ((size_t)4 << SA_ISA_CH);
}

// Second half of the binned rank store (see SA_ISA_CH): every chunk's pairs -> LDS -> one run of the rank array.
template <int DB>
__device__ inline void isa_flush(const SaLds<DB> &L, const uint32_t *pairs, uint32_t m, int32_t *rk, int tid) {
    constexpr uint32_t CH = 1u << SA_ISA_CH;
    constexpr int PER = (int)(CH / SAM_THREADS);
    uint32_t *stage = reinterpret_cast<uint32_t *>(L.woff);
    __syncthreads();  // every pair is written
    for (uint32_t c0 = 0; c0 < m; c0 += CH) {
        const uint32_t cnt = m - c0 < CH ? m - c0 : CH;
        uint32_t v[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const uint32_t x = (uint32_t)i * SAM_THREADS + (uint32_t)tid;
            v[i] = x < cnt ? pairs[c0 + x] : 0u;
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const uint32_t x = (uint32_t)i * SAM_THREADS + (uint32_t)tid;
            if (x < cnt) stage[v[i] & (CH - 1u)] = v[i] >> SA_ISA_CH;
        }
        __syncthreads();
        for (uint32_t x = (uint32_t)tid * 4; x < cnt; x += SAM_THREADS * 4) {
            if (x + 4 <= cnt) {
                const int2 lo = *reinterpret_cast<const int2 *>(stage + x), hi = *reinterpret_cast<const int2 *>(stage + x + 2);
                *reinterpret_cast<int4 *>(rk + c0 + x) = make_int4(lo.x, lo.y, hi.x, hi.y);
            } else {
                for (uint32_t y = x; y < cnt; ++y) rk[c0 + y] = (int)stage[y];
            }
        }
        __syncthreads();
    }
    if (SA_MTAB)
        for (int x = tid; x < SAM_WAVES * SaLds<DB>::mtab_entries(); x += SAM_THREADS) L.mtab[x] = 0;
    __syncthreads();
}

// Regroup: `m` sorted records (INITIAL: (initial key, suffix) in key order, one group;
// otherwise (k1, k2, suffix) sorted by (k1, sequence order)) -> slots, new group starts, bitmaps, next digit counts.
// Returns the number of suffixes that still share their group.
template <bool INITIAL, int DB>
__device__ inline uint32_t sa_regroup(const SaLds<DB> &L, SaCarry &cy, const uint64_t *rec, uint32_t m, int32_t *sa,
                                      int32_t *rk, int tid, uint32_t *fin_list, uint32_t fin_N, uint32_t *pairs = nullptr) {
    constexpr int NB = 1 << DB;
    constexpr int RS = INITIAL ? SaInit<DB>::SHIFT : 20;  // bits of the suffix index below the key(s)
    // the chunk staging area of the binned rank store: woff and the peer table (both idle here; the table is zeroed again)
    constexpr bool BINNED = INITIAL && isa_binned<DB>();
    const int lane = tid & 63, w = tid >> 6;
    int tileK = -1, tileS = -1;
    uint32_t rem = 0;
    int par = 0;
    RG_DECL
    uint64_t nx[SA_K];
#pragma unroll
    for (int s = 0; s < SA_K; ++s) {
        const uint32_t u = (uint32_t)w * SA_WSPAN + s * 64 + lane;
        nx[s] = (INITIAL && u < m) ? rec[u] : 0;
    }
    for (uint32_t t0 = 0; t0 < m; t0 += SA_TILE, par ^= 1) {
        RG(0)
        const uint32_t wb = t0 + (uint32_t)w * SA_WSPAN;
        uint64_t key[SA_K];
        int p[SA_K];
#pragma unroll
        for (int s = 0; s < SA_K; ++s) {
            const uint32_t u = wb + s * 64 + lane;
            key[s] = 0;
            p[s] = 0;
            if (u < m) {
                const uint64_t r = INITIAL ? nx[s] : rec[u];
                p[s] = (int)(r & ((1u << RS) - 1u));
                key[s] = r >> RS;
            }
        }
        RG(1)
        // neighbours across the wave's span (uniform loads)
        uint64_t pkey = 0, nkey = 0;
        const uint32_t un = wb + SA_WSPAN;
        if (wb > 0 && wb < m) pkey = rec[wb - 1] >> RS;
        if (un < m) nkey = rec[un] >> RS;

        // per entry: bit 0 = first of its group, bit 1 = first of its (group, second key) run, bit 2 = alone
        int fl[SA_K];
#pragma unroll
        for (int s = 0; s < SA_K; ++s) {
            const uint32_t u = wb + s * 64 + lane;
            const uint64_t q63 = lane_value(key[s > 0 ? s - 1 : 0], 63);
            const uint64_t q = lane_below(key[s], (s == 0) ? pkey : q63);
            const bool valid = u < m;
            const bool first = (u == 0);
            const bool fS = valid && (first || key[s] != q);
            const bool fK = valid && (first || (!INITIAL && (key[s] >> 20) != (q >> 20)));
            fl[s] = (fK ? 1 : 0) | (fS ? 2 : 0);
        }
        int gKl[SA_K], gSl[SA_K];
        int runK = -1, runS = -1;
        uint64_t bS[SA_K];
        bool nfend[SA_K];
#pragma unroll
        for (int s = 0; s < SA_K; ++s) bS[s] = __ballot(fl[s] & 2);
        const uint64_t lem = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
#pragma unroll
        for (int s = 0; s < SA_K; ++s) {
            const uint32_t u = wb + s * 64 + lane;
            // does the next entry start a run?  (from the ballots; the wave's last entry looks at its neighbour)
            bool nf;
            if (lane < 63) nf = (bS[s] >> (lane + 1)) & 1;
            else if (s + 1 < SA_K) nf = bS[s + 1 < SA_K ? s + 1 : s] & 1;
            else nf = (un < m) ? (nkey != key[s]) : true;
            if (u + 1 >= m) nf = true;
            nfend[s] = lane_value((int)nf, 63) != 0;  // the entry behind this step starts a run (or there is none)
            if ((fl[s] & 2) && nf) fl[s] |= 4;
            // index of the last marked entry at or before this one: from the ballots, no scans
            const int u0 = (int)(wb + s * 64);
            const uint64_t bK = __ballot(fl[s] & 1);
            const uint64_t lK = bK & lem, lS = bS[s] & lem;
            gKl[s] = lK ? u0 + 63 - (int)__builtin_clzll(lK) : runK;
            gSl[s] = lS ? u0 + 63 - (int)__builtin_clzll(lS) : runS;
            if (bK) runK = u0 + 63 - (int)__builtin_clzll(bK);
            if (bS[s]) runS = u0 + 63 - (int)__builtin_clzll(bS[s]);
        }
        if (lane == 0) {
            cy.a[par][w] = runK;
            cy.b[par][w] = runS;
        }
        RG(2)
        __syncthreads();
        RG(3)
        int cinK = tileK, cinS = tileS;
        for (int ww = 0; ww < SAM_WAVES; ++ww) {
            const int a = cy.a[par][ww], b = cy.b[par][ww];
            if (ww < w) {
                cinK = max(cinK, a);
                cinS = max(cinS, b);
            }
            tileK = max(tileK, a);
            tileS = max(tileS, b);
        }
        // small groups that lie inside a step: their first entry lists them for sa_finish (fsz = the group's size there)
        int fsz[SA_K];
#pragma unroll
        for (int s = 0; s < SA_K; ++s) {
            fsz[s] = 0;
            if (SA_FIN_G < 2) continue;
            const uint32_t u = wb + s * 64 + lane;
            const bool valid = u < m;
            const int nvalid = __popcll(__ballot(valid));
            const uint64_t above = lane == 63 ? 0ull : (bS[s] >> (lane + 1));
            int b = above ? lane + 1 + (int)__builtin_ctzll(above) : 64;     // lane behind the run's last entry
            if (b > nvalid) b = nvalid;
            const bool closed = b < nvalid || nvalid < 64 || nfend[s];
            const int size = b - lane;
            if (valid && (fl[s] & 2) && closed && size >= 2 && size <= SA_FIN_G) fsz[s] = size;
            if (BINNED) {  // members of a listed group: sa_finish stores their ranks (final, or the group's)
                const uint64_t heads = __ballot(fsz[s] > 0);
                const uint64_t lS = bS[s] & lem;  // group starts of this step at or below this lane
                if (lS && ((heads >> (63 - (int)__builtin_clzll(lS))) & 1)) fl[s] |= 8;
            }
        }
        RG(4)
        // the next tile's records, in flight while this tile is written (a tile used to start by waiting ~10 us for
        // them; the keys are dead by now - issued any earlier the loads cost 26 spilled registers: 64.5 vs 61.7 ms)
        if (INITIAL && t0 + SA_TILE < m) {
#pragma unroll
            for (int s = 0; s < SA_K; ++s) {
                const uint32_t u = wb + SA_TILE + s * 64 + lane;
                nx[s] = u < m ? rec[u] : 0;
            }
        }
#pragma unroll
        for (int s = 0; s < SA_K; ++s) {
            const uint32_t u = wb + s * 64 + lane;
            const bool valid = u < m;
            const int gK = max(gKl[s], cinK), gS = max(gSl[s], cinS);
            const int k1 = INITIAL ? 0 : (int)(key[s] >> 20);
            const int slot = k1 + ((int)u - gK);
            const int nr = k1 + (gS - gK);
            const bool fS = fl[s] & 2, sgl = fl[s] & 4;
            if (SA_FIN_G >= 2) {
#pragma unroll
                for (int z = 2; z <= SA_FIN_G; ++z) {
                    const uint64_t em = __ballot(fsz[s] == z);
                    if (em == 0) continue;
                    int base = 0;
                    if (lane == (int)__builtin_ctzll(em)) base = atomicAdd(&cy.fin_cnt[z - 2], __popcll(em));
                    base = lane_value(base, (int)__builtin_ctzll(em));
                    if (fsz[s] == z) fin_list[fin_list_base(z, fin_N) + (uint32_t)base + (uint32_t)__popcll(em & lanemask_lt(lane))] = (uint32_t)slot;
                }
            }
            if (valid) {
                sa[slot] = p[s] | ((fS && !sgl) ? (int)0x80000000 : 0);  // top bit: first slot of a shared group
                if (BINNED && (fl[s] & 8)) {
                } else if (BINNED) {
                    const uint32_t bin = (uint32_t)p[s] >> SA_ISA_CH;
                    const uint32_t at = (uint32_t)atomicAdd(&cy.bin_cur[bin], 1);
                    pairs[(bin << SA_ISA_CH) + at] = ((uint32_t)nr << SA_ISA_CH) | ((uint32_t)p[s] & ((1u << SA_ISA_CH) - 1u));
                } else {
                    rk[p[s]] = nr;
                }
                if (INITIAL) {
                    if (!sgl) atomicOr(&L.act[(uint32_t)p[s] >> 5], 1u << (p[s] & 31));
                } else if (sgl) {
                    atomicAnd(&L.act[(uint32_t)p[s] >> 5], ~(1u << (p[s] & 31)));
                }
            }
            // digit counts of the keys of the next round, one update per run of equal group starts
            const uint64_t vmask = __ballot(valid);
            const bool head = valid && (fS || lane == 0);
            const uint64_t heads = __ballot(head);
            if (head && !sgl) {
                const uint64_t rest = (lane == 63) ? 0ull : (heads >> (lane + 1));
                const int nxt = rest ? lane + 1 + (int)__builtin_ctzll(rest) : __popcll(vmask);
                const int len = nxt - lane;
                atomicAdd(&L.hl[nr & (NB - 1)], len);
                atomicAdd(&L.hh[nr >> DB], len);
            }
            rem += (uint32_t)__popcll(__ballot(valid && !sgl));
        }
        RG(5)
    }
    if (lane == 0) cy.wcount[w] = (int)rem;
    __syncthreads();
    uint32_t total = 0;
    for (int ww = 0; ww < SAM_WAVES; ++ww) total += (uint32_t)cy.wcount[ww];
    __syncthreads();
    return total;
}

// The listed groups (sa_regroup), one lane each: order the members by direct comparison; a group whose pairs are all
// decided gets its final slots, leaves the active set and the next round's digit counts.  Returns the number of
// suffixes finished.  d0: symbols the members of a group are known to share.
// One listed group of Z suffixes, ordered by one lane.  The members' texts are read ONCE per 32-byte block (all loads
// of a block in flight together) and every pair is compared from registers; a pair is settled by the first byte that
// differs, or by length when the shorter suffix ends first (the sentinel is the smallest symbol).  The round trips of
// a group are then those of its slowest pair, whatever Z - comparing the pairs one after the other made a group of
// five cost ten chains of dependent loads.  Returns false when a pair is still equal `span` bytes past d0.
#ifndef SA_FIN_SHARED
#define SA_FIN_SHARED 1
#endif
template <int Z>
__device__ inline int finish_group(const uint8_t *T, uint32_t n, const uint32_t *p, int d0, int span, int *place) {
    constexpr int NP = Z * (Z - 1) / 2;
    int mo[NP];  // per pair: offset of the first difference (>= the shorter length: settled by length), -1 = not yet known
    int rest[Z];
#pragma unroll
    for (int j = 0; j < Z; ++j) rest[j] = (int)(n - p[j]);
#pragma unroll
    for (int q = 0; q < NP; ++q) mo[q] = -1;
    const int cap = d0 + span;
    int o = d0;
    bool open = true;
    while (open && o < cap) {
        uint64_t w[Z][4];
#pragma unroll
        for (int j = 0; j < Z; ++j) {
            const int oj = o < rest[j] ? o : rest[j];  // past its end a suffix's pairs are settled: read the padding
#pragma unroll
            for (int t = 0; t < 4; ++t) w[j][t] = ld8(T + p[j] + oj + 8 * t);
        }
        open = false;
        int q = 0;
#pragma unroll
        for (int i = 0; i < Z; ++i)
#pragma unroll
            for (int j = i + 1; j < Z; ++j, ++q) {
                if (mo[q] >= 0) continue;
                const int lim = rest[i] < rest[j] ? rest[i] : rest[j];
                const uint64_t x0 = w[i][0] ^ w[j][0], x1 = w[i][1] ^ w[j][1], x2 = w[i][2] ^ w[j][2], x3 = w[i][3] ^ w[j][3];
                int m = -1;
                if (x0) m = o + (int)(__builtin_ctzll(x0) >> 3);
                else if (x1) m = o + 8 + (int)(__builtin_ctzll(x1) >> 3);
                else if (x2) m = o + 16 + (int)(__builtin_ctzll(x2) >> 3);
                else if (x3) m = o + 24 + (int)(__builtin_ctzll(x3) >> 3);
                else if (o + 32 >= lim) m = lim;
                mo[q] = m;
                open = open || m < 0;
            }
        o += 32;
    }
    if (open) {  // 2: no pair is decided - the members share their first d0 + span symbols
        bool none = true;
#pragma unroll
        for (int q = 0; q < NP; ++q) none = none && mo[q] < 0;
        return none ? 2 : 0;
    }
    // settle the pairs: by length, or by the two bytes at the difference (plain loads after the loop: see the note in
    // match.hip, dict_search_in); all pairs' loads are in flight together
    int ca[NP], cb[NP];
    {
        int q = 0;
#pragma unroll
        for (int i = 0; i < Z; ++i)
#pragma unroll
            for (int j = i + 1; j < Z; ++j, ++q) {
                const int lim = rest[i] < rest[j] ? rest[i] : rest[j];
                const int at = mo[q] < lim ? mo[q] : 0;
                ca[q] = T[p[i] + at];
                cb[q] = T[p[j] + at];
            }
    }
#pragma unroll
    for (int j = 0; j < Z; ++j) place[j] = 0;
    {
        int q = 0;
#pragma unroll
        for (int i = 0; i < Z; ++i)
#pragma unroll
            for (int j = i + 1; j < Z; ++j, ++q) {
                const int lim = rest[i] < rest[j] ? rest[i] : rest[j];
                // i's suffix is the larger one: it is the longer one when the shorter ended, else its byte is larger
                const bool i_larger = mo[q] >= lim ? p[i] < p[j] : ca[q] > cb[q];
                if (i_larger) ++place[i];
                else ++place[j];
            }
    }
    return true;
}

// The listed groups (sa_regroup), one lane each: order the members by direct comparison; a group whose pairs are all
// decided gets its final slots, leaves the active set and the next round's digit counts.  Returns the number of
// suffixes finished.  d0: symbols the members of a group are known to share.
template <int DB>
__device__ inline uint32_t sa_finish(const SaLds<DB> &L, SaCarry &cy, const uint32_t *fin_list, uint32_t fin_N, int32_t *sa,
                                     int32_t *rk, const uint8_t *T, uint32_t n, int d0, int tid, uint32_t *pairs = nullptr,
                                     const uint32_t *mpos = nullptr) {
    // mpos: the suffixes are those of the run-compressed text (sa_meta_kernel): members are item indices, item i starts at
    // byte mpos[i] & MP_POS_MASK, d0 counts items; equal items cover equal byte counts, so the bytes the members share are
    // those of the first member's d0 items, and the byte compare decides the items' order
    // pairs: after the initial regroup (binned rank store) - the ranks of ALL listed members go to the chunk streams
    constexpr int NB = 1 << DB;
    if (SA_FIN_G < 2) return 0;
    __syncthreads();  // the lists, the slots and the counters are complete
#pragma unroll
    for (int z = 2; z <= SA_FIN_G; ++z) {
        const uint32_t cnt = (uint32_t)cy.fin_cnt[z - 2];
        for (uint32_t g = tid; g < cnt; g += SAM_THREADS) {
            const uint32_t slot0 = fin_list[fin_list_base(z, fin_N) + g];
            uint32_t p[SA_FIN_G];
#pragma unroll
            for (int j = 0; j < z; ++j) p[j] = (uint32_t)sa[slot0 + j] & 0x7FFFFFFFu;
            int place[SA_FIN_G];
#if SA_FIN_SHARED
            uint32_t pb_[SA_FIN_G];   // byte positions of the members
            int d0b = d0;             // bytes the members share
            if (mpos) {
#pragma unroll
                for (int j = 0; j < z; ++j) pb_[j] = mpos[p[j]] & MP_POS_MASK;
                d0b = (int)((mpos[p[0] + (uint32_t)d0] & MP_POS_MASK) - pb_[0]);
            } else {
#pragma unroll
                for (int j = 0; j < z; ++j) pb_[j] = p[j];
            }
            const uint32_t *pb = pb_;
            int ok = 0;
            if (z == 2) ok = finish_group<2>(T, n, pb, d0b, SA_FIN_SPAN, place);
            else if (z == 3) ok = finish_group<3>(T, n, pb, d0b, SA_FIN_SPAN, place);
            else if (z == 4) ok = finish_group<(SA_FIN_G >= 4 ? 4 : 2)>(T, n, pb, d0b, SA_FIN_SPAN, place);
            else if (z == 5) ok = finish_group<(SA_FIN_G >= 5 ? 5 : 2)>(T, n, pb, d0b, SA_FIN_SPAN, place);
            else if (z == 6) ok = finish_group<(SA_FIN_G >= 6 ? 6 : 2)>(T, n, pb, d0b, SA_FIN_SPAN, place);
#ifdef SA_PROFILE_ROUNDS
            if (ok != 1) atomicAdd(&cy.fin_cnt[ok == 2 ? 6 : 5], z);
#endif
            if (ok != 1) {
                if (pairs) {
#pragma unroll
                    for (int j = 0; j < z; ++j) {
                        const uint32_t bin = p[j] >> SA_ISA_CH;
                        const uint32_t at = (uint32_t)atomicAdd(&cy.bin_cur[bin], 1);
                        pairs[(bin << SA_ISA_CH) + at] = (slot0 << SA_ISA_CH) | (p[j] & ((1u << SA_ISA_CH) - 1u));
                    }
                }
                continue;
            }
#else
#pragma unroll
            for (int j = 0; j < z; ++j) place[j] = 0;
            bool und = false;
#pragma unroll
            for (int i = 0; i < z; ++i)
#pragma unroll
                for (int j = i + 1; j < z; ++j) {
                    const int c = suffix_cmp(T, n, p[i], p[j], d0);
                    und = und || c == 0;
                    if (c > 0) ++place[i];
                    else ++place[j];
                }
            if (und) continue;
#endif
#pragma unroll
            for (int j = 0; j < z; ++j) {
                const uint32_t slot = slot0 + (uint32_t)place[j];
                sa[slot] = (int)p[j];
                if (pairs) {
                    const uint32_t bin = p[j] >> SA_ISA_CH;
                    const uint32_t at = (uint32_t)atomicAdd(&cy.bin_cur[bin], 1);
                    pairs[(bin << SA_ISA_CH) + at] = (slot << SA_ISA_CH) | (p[j] & ((1u << SA_ISA_CH) - 1u));
                } else {
                    rk[p[j]] = (int)slot;
                }
                atomicAnd(&L.act[p[j] >> 5], ~(1u << (p[j] & 31)));
            }
            atomicSub(&L.hl[slot0 & (NB - 1)], z);
            atomicSub(&L.hh[slot0 >> DB], z);
            atomicAdd(&cy.fin_cnt[7], z);
        }
    }
    __syncthreads();
    const uint32_t done = (uint32_t)cy.fin_cnt[7];
#ifdef SA_PROFILE_ROUNDS
    if (blockIdx.x == 0 && tid == 0) printf("  finish d0=%d: listed groups %d %d %d %d, finished %u, failed with a decided pair %d, failed all equal %d\n", d0, cy.fin_cnt[0], cy.fin_cnt[1], cy.fin_cnt[2], cy.fin_cnt[3], done, cy.fin_cnt[5], cy.fin_cnt[6]);
#endif
    __syncthreads();
    if (tid < 8) cy.fin_cnt[tid] = 0;
    __syncthreads();
    return done;
}

// ---- initial order: SaInit<DB>::DIGITS LSD passes over the NN records (36- or 45-bit key << SHIFT | index) in recB, least
//      significant digit first, ping-pong with recA; the digit counts come from the sweep that built the records
template <int DB>
__device__ inline void sa_initial_sort(const SaLds<DB> &L, uint64_t *recB, uint64_t *recA, uint32_t N, const int32_t *hZ,
                                       const int32_t *hY, const int32_t *hX, const int32_t *hC, const int32_t *hS, int tid) {
    constexpr int NB = 1 << DB;
    const int lane = tid & 63, w = tid >> 6;
    SA_PROF_DECL
    {
        const uint64_t *rin = recB;
        uint64_t *rout = recA;
        for (int pass = 0; pass < SaInit<DB>::DIGITS; ++pass) {
            const int shift = SaInit<DB>::SHIFT + 9 * pass;
            const int dg = pass + 5 - SaInit<DB>::DIGITS;  // 0: z, 1: y, 2: x, 3: run code, 4: first symbol
            bins_scan(dg == 0 ? hZ : dg == 1 ? hY : dg == 2 ? hX : dg == 3 ? hC : hS, L.base, NB, tid);
            SA_PROF(15)
            // the records of the next tile are loaded while the current one is being placed
            auto load_rec = [&](uint32_t e) -> uint64_t {
                return e < N ? rin[e] : 0;
            };
            uint64_t nx[SA_K];
#pragma unroll
            for (int s = 0; s < SA_K; ++s) nx[s] = load_rec((uint32_t)w * SA_WSPAN + s * 64 + lane);
            for (uint32_t t0 = 0; t0 < N; t0 += SA_TILE) {
                const uint32_t wb = t0 + (uint32_t)w * SA_WSPAN;
                uint64_t r[SA_K];
                int loc[SA_K];
#pragma unroll
                for (int s = 0; s < SA_K; ++s) r[s] = nx[s];
                SA_PROF(0)
#pragma unroll
                for (int s = 0; s < SA_K; ++s) {
                    const uint32_t e = wb + s * 64 + lane;
                    loc[s] = tile_rank<9, DB>(L, w, (int)(r[s] >> shift) & 511, e < N, lane);
                }
                SA_PROF(1)
                if (t0 + SA_TILE < N) {
#pragma unroll
                    for (int s = 0; s < SA_K; ++s) nx[s] = load_rec(wb + SA_TILE + s * 64 + lane);
                }
                tile_prepare<DB>(L, tid);
                SA_PROF(2)
#pragma unroll
                for (int s = 0; s < SA_K; ++s) {
                    const uint32_t e = wb + s * 64 + lane;
                    const int sym = (int)(r[s] >> shift) & 511;
                    if (e < N) tile_put<DB>(L, rout, sym, w, loc[s], r[s]);
                }
                SA_PROF(3)
            }
            __syncthreads();
            rin = rout;
            rout = (rout == recA) ? recB : recA;
        }
    }
}

// ---- the doubling rounds: NN suffixes (of the text itself, or - mpos != nullptr - of the run-compressed text whose
//      item i starts at byte mpos[i] & MP_POS_MASK of T), m of them still sharing their group, which share h symbols
template <int DB>
__device__ inline void sa_rounds(const SaLds<DB> &L, SaCarry &cy, uint64_t *recA, uint64_t *recB, int32_t *sa, int32_t *rk,
                                 uint32_t N, uint32_t words, uint32_t m, uint32_t h, const uint8_t *T, uint32_t n,
                                 const uint32_t *mpos, int tid, int max_rounds) {
    constexpr int NB = 1 << DB;
    constexpr int MASK = NB - 1;
    const int lane = tid & 63, w = tid >> 6;
    SA_PROF_DECL
    while (m != 0 && h < N && --max_rounds >= 0) {
#ifdef SA_PROFILE_ROUNDS
        if (blockIdx.x == 0 && tid == 0) printf("round h=%u m=%u late=%d\n", h, m, (int)((uint64_t)m * SA_LATE_DIV <= N));
#endif
        if ((uint64_t)m * SA_LATE_DIV <= N) {
            // ---- late round: few suffixes are left, so instead of sweeping the whole suffix array the records
            //      (k1 = RK[p], k2 = RK[p + h], p) are built straight from the `act` bitmap and sorted by (k1, k2):
            //      four passes over m records instead of one sweep over N slots and two passes
            int32_t *hx = L.tpos;
            for (int x = tid; x < NB; x += SAM_THREADS) {
                hx[x] = 0;
                L.hy[x] = 0;
            }
            if (tid == 0) cy.wcount[0] = 0;
            __syncthreads();
            // the active positions first, as a list (in the part of recA the passes below do not reach) ...
            uint32_t *plist = (uint32_t *)(recA + (N + 1) / 2);
            for (uint32_t x = tid; x < words; x += SAM_THREADS) {
                uint32_t wbits = L.act[x];
                if (wbits) {
                    int pos = atomicAdd(&cy.wcount[0], __popc(wbits));
                    while (wbits) {
                        plist[pos++] = x * 32 + (uint32_t)__builtin_ctz(wbits);
                        wbits &= wbits - 1;
                    }
                }
            }
            __syncthreads();
            // ... then one record per list entry: the two rank gathers of an entry depend on nothing but the list, so
            // a thread has those of several entries in flight (gathering inside the bit loop above made every entry
            // wait for the one before: 26 % of the kernel for 5 % of the suffixes)
            for (uint32_t q0 = tid; q0 < m; q0 += 4 * SAM_THREADS) {
                uint32_t p[4], k1[4], k2[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t q = q0 + (uint32_t)u * SAM_THREADS;
                    p[u] = q < m ? plist[q] : 0u;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    k1[u] = (uint32_t)rk[p[u]];
                    k2[u] = (p[u] + h < N) ? (uint32_t)rk[p[u] + h] : 0u;
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint32_t q = q0 + (uint32_t)u * SAM_THREADS;
                    if (q < m) {
                        recB[q] = ((uint64_t)k1[u] << 40) | ((uint64_t)k2[u] << 20) | (uint64_t)p[u];
                        atomicAdd(&hx[k2[u] & MASK], 1);
                        atomicAdd(&L.hy[k2[u] >> DB], 1);
                    }
                }
            }
            __syncthreads();
            SA_PROF(16)
            scatter_pass<DB>(L, recB, recA, m, 20, hx, tid);
            scatter_pass<DB>(L, recA, recB, m, 20 + DB, L.hy, tid);
            scatter_pass<DB>(L, recB, recA, m, 40, L.hl, tid);
            scatter_pass<DB>(L, recA, recB, m, 40 + DB, L.hh, tid);
            SA_PROF(17)
        } else {
        // ---- S1: active members of the sequence p = SA[j] - h -> records, scattered by the low digit of k1
        bins_scan(L.hl, L.base, NB, tid);
        SA_PROF(15)
        {
            int tileG = 0;  // slot 0 always starts a group
            int par = 0;
            int nsv[SA_K];
#pragma unroll
            for (int s = 0; s < SA_K; ++s) {
                const uint32_t j = (uint32_t)w * SA_WSPAN + s * 64 + lane;
                nsv[s] = (j < N) ? sa[j] : 0;
            }
            for (uint32_t t0 = 0; t0 < N; t0 += SA_TILE, par ^= 1) {
                const uint32_t wb = t0 + (uint32_t)w * SA_WSPAN;
                int sv[SA_K];
#pragma unroll
                for (int s = 0; s < SA_K; ++s) sv[s] = nsv[s];
                SA_PROF(5)
                int p[SA_K], k1[SA_K], k2[SA_K], loc[SA_K];
                bool a[SA_K];
#pragma unroll
                for (int s = 0; s < SA_K; ++s) {
                    const uint32_t j = wb + s * 64 + lane;
                    const uint32_t sx = (uint32_t)sv[s] & 0x7FFFFFFFu;
                    const uint32_t pp = sx >= h ? sx - h : N - h + sx;
                    p[s] = (int)pp;
                    a[s] = (j < N) && ((L.act[pp >> 5] >> (pp & 31)) & 1u);
                    k1[s] = a[s] ? rk[pp] : 0;
                }
                if (t0 + SA_TILE < N) {  // the next tile's slots, loaded while this tile is being placed
#pragma unroll
                    for (int s = 0; s < SA_K; ++s) {
                        const uint32_t j = wb + SA_TILE + s * 64 + lane;
                        nsv[s] = (j < N) ? sa[j] : 0;
                    }
                }
                SA_PROF(6)
                int runG = -1;
#pragma unroll
                for (int s = 0; s < SA_K; ++s) {
                    const uint32_t j = wb + s * 64 + lane;
                    const uint32_t sx = (uint32_t)sv[s] & 0x7FFFFFFFu;
                    // slot j starts a group: marked, or its suffix is alone in its group
                    const bool gstart = (j < N) && ((sv[s] < 0) || !((L.act[sx >> 5] >> (sx & 31)) & 1u));
                    const uint64_t gw = __ballot(gstart);
                    const uint64_t le = gw & ((lane == 63) ? ~0ull : ((2ull << lane) - 1ull));
                    const int j0 = (int)(wb + s * 64);
                    k2[s] = le ? j0 + 63 - (int)__builtin_clzll(le) : runG;
                    if (gw) runG = j0 + 63 - (int)__builtin_clzll(gw);
                    loc[s] = 0;
                    if (__ballot(a[s])) loc[s] = tile_rank<DB, DB>(L, w, k1[s] & MASK, a[s], lane);
                }
                if (lane == 0) cy.a[par][w] = runG;
                SA_PROF(7)
                tile_prepare<DB>(L, tid);
                SA_PROF(8)
                int cin = tileG;
                for (int ww = 0; ww < SAM_WAVES; ++ww) {
                    const int g = cy.a[par][ww];
                    if (ww < w) cin = max(cin, g);
                    tileG = max(tileG, g);
                }
#pragma unroll
                for (int s = 0; s < SA_K; ++s) {
                    if (a[s]) {
                        const int d = k1[s] & MASK;
                        const int g = k2[s] < 0 ? cin : k2[s];
                        tile_put<DB>(L, recA, d, w, loc[s],
                                     ((uint64_t)(uint32_t)k1[s] << 40) | ((uint64_t)(uint32_t)g << 20) | (uint64_t)(uint32_t)p[s]);
                    }
                }
                SA_PROF(9)
            }
        }
        __syncthreads();
        // ---- S2: records by the high digit of k1
        scatter_pass<DB>(L, recA, recB, m, 40 + DB, L.hh, tid);
        SA_PROF(13)
        }
        for (int x = tid; x < 2 * NB; x += SAM_THREADS) L.hl[x] = 0;  // hl, hh: refilled by the regroup below
        __syncthreads();
        // ---- D: new groups
        m = sa_regroup<false, DB>(L, cy, recB, m, sa, rk, tid, (uint32_t *)recA, N);
        SA_PROF(14)
        m -= sa_finish<DB>(L, cy, (const uint32_t *)recA, N, sa, rk, T, n, (int)(2 * h < (1u << 20) ? 2 * h : (1u << 20)), tid, nullptr, mpos);
        SA_PROF(18)
        h *= 2;
    }
    if (m != 0) {  // rounds cut short (timing experiments): drop the group marks so that SA stays a permutation
        __syncthreads();
        for (uint32_t j = tid; j < N; j += SAM_THREADS) sa[j] &= 0x7FFFFFFF;
    }
}

template <int DB>
__global__ __launch_bounds__(SAM_THREADS) __attribute__((amdgpu_waves_per_eu(SAM_WAVES * SA_WGS_PER_CU / 4, SAM_WAVES * SA_WGS_PER_CU / 4))) void sa_build_kernel(const Seg *segs, const uint8_t *text, SaWork wk, int max_rounds) {
    constexpr int NB = 1 << DB;
    constexpr int MASK = NB - 1;
    __shared__ SaCarry cy;
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn_lds[];
    const Seg sg = segs[blockIdx.x];
    const uint8_t *T = text + sg.off;
    const uint32_t n = sg.n;
    const uint32_t N = n + 1;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int32_t *sa = wk.sa + sg.off, *rk = wk.isa + sg.off;

    if (N == 1) {
        if (tid == 0) {
            sa[0] = 0;
            rk[0] = 0;
        }
        return;
    }
    const uint32_t words = sa_bitmap_words<DB>(N);
    const SaLds<DB> L(dyn_lds, words);
    uint64_t *recA = wk.recA + sg.off, *recB = wk.recB + sg.off;

    SA_PROF_DECL
    if (tid < 8) cy.fin_cnt[tid] = 0;
    if (tid < 64) cy.bin_cur[tid] = 0;
    for (uint32_t x = tid; x < words; x += SAM_THREADS) L.act[x] = 0;
    for (int x = tid; x < 7 * NB; x += SAM_THREADS) L.base[x] = 0;  // base, delta, tpos, hl, hh, hy, hsec
    for (int x = tid; x < SAM_WAVES * NB; x += SAM_THREADS) L.whist[x] = 0;
    if (SA_MTAB)
        for (int x = tid; x < SAM_WAVES * SaLds<DB>::mtab_entries(); x += SAM_THREADS) L.mtab[x] = 0;
    __syncthreads();
    SA_PROF(20)

    // ---- initial keys.  36 bits per suffix, 4 digits of 9: (first symbol c, run code, x, y) where the run code
    //      encodes the length rl of the run of c the suffix starts with and x, y are the two symbols after the
    //      run (0 = sentinel, byte + 1 otherwise).  Among suffixes starting with c, those whose run ends in a
    //      smaller symbol come first, shorter runs first (codes 1..253), then runs of >= SA_RUN_CAP (code 255, x
    //      and y left out), then those whose run ends in a larger symbol, longer runs first (codes 511 - rl).  The
    //      key order is the suffix order, every key group shares at least 3 symbols, and suffixes inside long
    //      runs - most of the deep groups of zero-heavy inputs - start out almost sorted (measured on the bench
    //      blobs: the doubling rounds move 0.98 N suffixes in total instead of 2.44 N after a plain 4-symbol key).
    //      One sweep builds the records (key << 20 | suffix) in text order and counts the four digits.
    int32_t *hY = (int32_t *)L.act, *hX = hY + NB, *hC = L.hl, *hS = L.hh;  // act is free until the first regroup
    int32_t *hZ = L.hy;                                                       // (a late round's array)
    {
        // Every wave sweeps a contiguous range of 64-position blocks, so the "byte == next byte" ballots that give the
        // run lengths slide: a block's run lengths read the flags of itself and the next four blocks (exact up to 257),
        // and only the fifth is new in each step - one two-byte load per lane and step instead of ten byte loads.
        const uint32_t nblk = (N + 63) / 64, per = (nblk + SAM_WAVES - 1) / SAM_WAVES;
        // (the wave's index as an SGPR: with bounds the compiler takes for divergent, this loop was compiled with exec
        // masking, its ballots in VGPRs and its loads waited for where they are issued)
        const uint32_t wu = (uint32_t)wave_index(tid);
        const uint32_t bb = wu * per, be = bb + per < nblk ? bb + per : nblk;
        // The loads of a step are issued ahead of their use (4 waves per SIMD hide little latency by themselves, and a
        // step that waited for its flag load and then for the load behind its run took 2.6 us): the flag bytes of a
        // block arrive two steps before their ballot, and the symbols behind the runs of a block are requested
        // SA_SWEEP_AHEAD steps before the block's records are built.
        constexpr int PD = SA_SWEEP_AHEAD;
        auto raw_load = [&](uint32_t blk) -> uint32_t {
            const uint32_t q = blk * 64 + (uint32_t)lane;
            uint16_t v = 0;
            if (q < N) __builtin_memcpy(&v, T + q, 2);  // (zero padding follows the text)
            return v;
        };
        auto flags_of = [&](uint32_t blk, uint32_t v) -> uint64_t {
            const uint32_t q = blk * 64 + (uint32_t)lane;
            return __ballot(q + 1 < n && (v & 0xFFu) == (v >> 8));
        };
        uint64_t E1 = 0, E2 = 0, E3 = 0, E4 = 0;
        uint32_t rawA = 0, rawB = 0;
        if (bb < be) {
            const uint32_t r0 = raw_load(bb), r1 = raw_load(bb + 1), r2 = raw_load(bb + 2), r3 = raw_load(bb + 3);
            rawA = raw_load(bb + 4);
            rawB = raw_load(bb + 5);
            E1 = flags_of(bb, r0);
            E2 = flags_of(bb + 1, r1);
            E3 = flags_of(bb + 2, r2);
            E4 = flags_of(bb + 3, r3);
        }
        int p_rl[PD];
        uint32_t p_v[PD], p_s[PD];
#pragma unroll
        for (int t = 0; t < PD; ++t) {
            p_rl[t] = 1;
            p_v[t] = p_s[t] = 0;
        }
        for (uint32_t c = bb; c < be + PD && bb < be; ++c) {
            // ---- block c: run lengths from the sliding flags, loads of the symbols behind the runs
            int n_rl = 1;
            uint32_t n_v = 0, n_s = 0;
            if (c < be) {
                const uint64_t E0 = E1;
                E1 = E2; E2 = E3; E3 = E4;
                E4 = flags_of(c + 4, rawA);
                rawA = rawB;
                rawB = raw_load(c + 6);
                const uint32_t e = c * 64 + (uint32_t)lane;
                // ones from this lane's flag on (run_length_64 on the sliding flags)
                const uint64_t x0 = E0 >> lane;
                int rl = (x0 == ~0ull) ? 64 : (int)__builtin_ctzll(~x0);
                if (rl == 64 - lane) {
                    const uint64_t Et[4] = {E1, E2, E3, E4};
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const int rt = Et[t] == ~0ull ? 64 : (int)__builtin_ctzll(~Et[t]);
                        rl += rt;
                        if (rt < 64) break;
                    }
                }
                n_rl = rl + 1;
                if (e < N) {
                    n_s = T[e];
                    if (n_rl < SA_RUN_CAP) __builtin_memcpy(&n_v, T + e + (uint32_t)n_rl, 4);  // (the text is zero padded)
                }
            }
            // ---- block c - PD: records and digit counts
            if (c >= bb + PD) {
                const uint32_t e = (c - PD) * 64 + (uint32_t)lane;
                const int rl = p_rl[0];
                const uint32_t v = p_v[0];
                const bool valid = e < N;
                if (valid) wk.rlen[sg.off + e] = (uint8_t)(rl < 255 ? rl : 255);
                int dZ = 0, dY = 0, dX = 0, dC = 0, dS = 0;
                if (valid) {
                    dS = e < n ? (int)p_s[0] + 1 : 0;
                    if (rl >= SA_RUN_CAP) {
                        dC = 255;
                    } else {
                        const uint32_t q = e + (uint32_t)rl;
                        dX = q < n ? (int)(v & 0xFFu) + 1 : 0;
                        dY = q + 1 < n ? (int)((v >> 8) & 0xFFu) + 1 : 0;
                        dZ = q + 2 < n ? (int)((v >> 16) & 0xFFu) + 1 : 0;
                        dC = dX < dS ? rl : 511 - rl;
                    }
                    uint64_t rec = ((((uint64_t)dS << 9) | (uint64_t)dC) << 9) | (uint64_t)dX;
                    if (SaInit<DB>::DIGITS >= 4) rec = (rec << 9) | (uint64_t)dY;
                    if (SaInit<DB>::Z) rec = (rec << 9) | (uint64_t)dZ;
                    rec = (rec << SaInit<DB>::SHIFT) | (uint64_t)e;
                    recB[e] = rec;
                }
                if (SaInit<DB>::Z) digit_count(hZ, dZ, valid, lane);
                if (SaInit<DB>::DIGITS >= 4) digit_count(hY, dY, valid, lane);
                digit_count(hX, dX, valid, lane);
                digit_count(hC, dC, valid, lane);
                digit_count(hS, dS, valid, lane);
            }
#pragma unroll
            for (int t = 0; t + 1 < PD; ++t) {
                p_rl[t] = p_rl[t + 1];
                p_v[t] = p_v[t + 1];
                p_s[t] = p_s[t + 1];
            }
            p_rl[PD - 1] = n_rl;
            p_v[PD - 1] = n_v;
            p_s[PD - 1] = n_s;
        }
    }
    __syncthreads();
    SA_PROF(21)

    sa_initial_sort<DB>(L, recB, recA, N, hZ, hY, hX, hC, hS, tid);
    const uint64_t *rcur = (SaInit<DB>::DIGITS & 1) ? recA : recB;
    for (int x = tid; x < 2 * NB; x += SAM_THREADS) L.hl[x] = 0;  // hl, hh
    for (uint32_t x = tid; x < words; x += SAM_THREADS) L.act[x] = 0;  // held two digit histograms so far
    __syncthreads();

    SA_PROF(15)
    uint32_t m = sa_regroup<true, DB>(L, cy, rcur, N, sa, rk, tid, (uint32_t *)(rcur == recA ? recB : recA), N, wk.pairs + sg.off);
    SA_PROF(4)
    m -= sa_finish<DB>(L, cy, (const uint32_t *)(rcur == recA ? recB : recA), N, sa, rk, T, n, (int)SaInit<DB>::H0, tid,
                       isa_binned<DB>() ? wk.pairs + sg.off : nullptr);
    if (isa_binned<DB>()) isa_flush<DB>(L, wk.pairs + sg.off, N, rk, tid);
    SA_PROF(19)
    uint32_t h = SaInit<DB>::H0;  // every group of the initial keys shares at least that many symbols
#ifdef SA_PROFILE_ROUNDS  // (a device printf costs far more than a round: not in the clocked build)
    if (blockIdx.x == 0 && tid == 0) printf("after init: N=%u m=%u\n", N, m);
#endif
    sa_rounds<DB>(L, cy, recA, recB, sa, rk, N, words, m, h, T, n, nullptr, tid, max_rounds);
    SA_PROF(15)
    SA_PROF_PRINT
}

// ---- run-compressed suffix sort ------------------------------------------------------------------------------------
// 60 % of the positions of calldata-like texts lie inside a run of equal bytes behind its first byte.  Their suffixes
// need no sorting of their own: the suffix at q inside a run of c that ends at e is c^(e - q) followed by the suffix at e,
// so among the suffixes that start with the same number of c's (and end their run the same way) it sorts by the rank of
// the suffix at e.  sa_meta_kernel therefore sorts ITEMS only: the first position of every run, and - runs of more than
// 253 bytes - every position with at least 253 bytes of the run to go (a run is cut into single "deep" bytes and a final
// chunk of 253).  An item's symbol is (byte + 1, run code of its chunk: length l if the chunk is followed by a smaller
// byte or the end, 255 for a deep byte, 511 - l if followed by a larger byte); equal symbols cover equal byte counts and
// the lexicographic order of the symbol sequences is the order of the suffixes (the run code orders a shorter run
// before or behind a longer one exactly as the byte behind it does), so the items are sorted by the machinery above -
// prefix doubling in ITEM space: keys of two symbols, h counted in items, texts of ~0.4 N symbols without any runs -
// with the finisher comparing bytes at the items' positions.  The EXPANSION then writes the whole suffix array in one
// sweep over the sorted items: in rank order, item u is the tail of the chunk before it (entry: tail position, chunk
// length l, type), entries are scattered by the chunk's byte (sections), and inside a section the records
// (tail - k, k = 1 .. l) of run code k are the entries with l >= k in rank order - level k is the previous level
// filtered, dense in every lane, and lands in the suffix array as one contiguous run.
#ifndef SA_META
#define SA_META 1
#endif
#ifdef SA_META_CHECK  // debug builds: every index of the expansion is checked, a violation is reported and skipped
__device__ int g_meta_bad = 0;
#define MCHK(cond, site, a_, b_) ((cond) ? true : (atomicAdd(&g_meta_bad, 1) < 40 ? (printf("META CHECK site %d blk %d tid %d: %lld %lld\n", site, (int)blockIdx.x, (int)threadIdx.x, (long long)(a_), (long long)(b_)), false) : false))
#else
#define MCHK(cond, site, a_, b_) true
#endif
#ifdef SA_META_PROFILE  // per-phase clocks of a few workgroups (debug builds only)
#define MPH(i) { __builtin_amdgcn_s_waitcnt(0); __syncthreads(); const long long t_ = clock64(); mph[i] = t_ - mph_last; mph_last = t_; }
#define MPH_DECL long long mph[16] = {}; long long mph_last = clock64();
#define MPH_PRINT if (tid == 0 && (blockIdx.x % 997) == 0) printf("blk %d N %u items %u | init %lld sweep %lld passB %lld sort %lld regroup %lld finish %lld flush %lld rounds %lld P %lld scan %lld Xbig %lld Xsmall %lld flush2 %lld\n", (int)blockIdx.x, N, NM, mph[0], mph[1], mph[2], mph[3], mph[4], mph[5], mph[6], mph[7], mph[8], mph[9], mph[10], mph[11], mph[12]);
#else
#define MPH(i)
#define MPH_DECL
#define MPH_PRINT
#endif
constexpr int XP_W = 512;            // entries a wave expands at a time (its private list in LDS)
constexpr uint32_t XP_BIG = 1024;    // sections of at least this many entries are expanded by all waves together
constexpr int XP_TAB = 512;          // level table: [k] chunks followed by a smaller byte, [255] deep bytes, [256 + k] by a larger one

// histogram of (type, level count) of entries [a, b) into tab (zeroed); one wave.  Returns the largest level count.
__device__ inline int xp_count(const uint32_t *ent, uint32_t a, uint32_t b, int32_t *tab, int lane) {
    int mx = 1;
    for (uint32_t x = a + (uint32_t)lane; x < b; x += 64) {
        const uint32_t e = ent[x];
        const uint32_t ty = (e >> 28) & 1u, ell = (e >> 20) & 0xFFu;
        atomicAdd(&tab[ty * 256u + ell], 1);
        mx = (int)ell > mx ? (int)ell : mx;
    }
    __builtin_amdgcn_wave_barrier();
    return wave_max(mx);
}
// tab[k] = number of entries of that type with at least k levels (in place, k = 1 .. 253 of both types); one wave.
// qmax: no entry has more than 64 qmax + 63 levels (the 64-entry stretches of the tables above that are zero and stay so)
__device__ inline void xp_levels(int32_t *tab, int lane, int qmax) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        int carry = 0;
        for (int q = qmax; q >= 0; --q) {
            const int idx = q * 64 + lane;
            const bool in = idx >= 1 && idx <= 253;
            const int v = in ? tab[half * 256 + idx] : 0;
            const int pre = wave_incl_add(v, lane);
            const int tot = lane_value(pre, 63);
            if (in) tab[half * 256 + idx] = carry + tot - pre + v;
            carry += tot;
        }
    }
    __builtin_amdgcn_wave_barrier();
}
// out[.] = first slot of every level's records in the section that starts at slot s0 and holds secb bytes, from the
// section's level counts cnt.  The suffixes of a section (first byte c) are ordered by the length of the run of c they
// start with: runs followed by a smaller byte (or the end of the text) first, shorter ones first - levels 1, 2, ... of
// type 0 ascending -, then runs followed by a larger byte, longer ones first - the levels of type 1 descending.  The
// levels beyond 253 (runs longer than that, rare) sit between the two halves of the table: out[255] = their first slot,
// out[254] = how many records they are (xp_long places them).  One wave; out may be cnt.
__device__ inline void xp_offsets(const int32_t *cnt, int32_t *out, int s0, int secb, int lane, int qmax) {
    int c0[4], c1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int idx = q * 64 + lane;
        const bool in = q <= qmax && idx >= 1 && idx <= 253;
        c0[q] = in ? cnt[idx] : 0;
        c1[q] = in ? cnt[256 + idx] : 0;
    }
    __builtin_amdgcn_wave_barrier();
    int carry = s0, tot1 = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // type 0: ascending k
        if (q > qmax) continue;
        const int idx = q * 64 + lane;
        const int pre = wave_incl_add(c0[q], lane);
        if (idx >= 1 && idx <= 253) out[idx] = carry + pre - c0[q];
        carry += lane_value(pre, 63);
        tot1 += lane_value(wave_incl_add(c1[q], lane), 63);
    }
    const int nlong = secb - (carry - s0) - tot1;  // records of the levels beyond 253
    if (lane == 0) {
        out[255] = carry;
        out[254] = nlong;
    }
    carry += nlong;
#pragma unroll
    for (int q = 3; q >= 0; --q) {  // type 1: descending k
        if (q > qmax) continue;
        const int idx = q * 64 + lane;
        const int pre = wave_incl_add(c1[q], lane);
        const int tot = lane_value(pre, 63);
        if (idx >= 1 && idx <= 253) out[256 + idx] = carry + tot - pre;   // the levels above k come first
        carry += tot;
    }
    __builtin_amdgcn_wave_barrier();
}
// Entries [a, b) of one section, level by level: level k writes the records of the entries with at least k levels (run[]
// holds the next free slot of every level and moves on), then drops the entries with exactly k from the wave's list.
// Four steps of 64 entries at a time, so that their LDS and global round trips overlap.  One wave.
__device__ inline void xp_expand(const uint32_t *ent, uint32_t a, uint32_t b, int32_t *run, uint32_t *buf, int32_t *sa,
                                 int32_t *rk, int lane, uint32_t N) {
    (void)N;
    const uint64_t lt = lanemask_lt(lane);
    for (uint32_t a0 = a; a0 < b; a0 += XP_W) {
        uint32_t cur = b - a0 < (uint32_t)XP_W ? b - a0 : (uint32_t)XP_W;
        for (uint32_t x = lane; x < cur; x += 64) buf[x] = ent[a0 + x];
        __builtin_amdgcn_wave_barrier();
        for (uint32_t k = 1; cur > 0; ++k) {
            int r0 = run[k], r1 = run[256 + k];
            uint32_t keep = 0;
            for (uint32_t c0 = 0; c0 < cur; c0 += 256) {
                uint32_t e[4];
                bool kp[4];
                uint64_t kb[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t x = c0 + (uint32_t)q * 64 + (uint32_t)lane;
                    e[q] = x < cur ? buf[x] : 0u;   // (a valid entry is never 0: it has at least one level)
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool valid = e[q] != 0u;
                    const uint32_t ty = (e[q] >> 28) & 1u, ell = (e[q] >> 20) & 0xFFu;
                    const uint64_t b0 = __ballot(valid && ty == 0u), b1 = __ballot(valid && ty == 1u);
                    if (valid && MCHK((e[q] & MP_POS_MASK) >= k && (e[q] & MP_POS_MASK) - k < N && k <= 253u, 10, e[q], k)) {
                        const int slot = ty == 0u ? r0 + __popcll(b0 & lt) : r1 + __popcll(b1 & lt);
                        const uint32_t pos = (e[q] & MP_POS_MASK) - k;
                        if (MCHK(slot >= 1 && (uint32_t)slot < N, 11, slot, e[q])) {
                            sa[slot] = (int)pos;
                            if (pos < (uint32_t)HEAD) rk[pos] = slot;  // (all of the inverse array that the match kernels read)
                        }
                    }
                    r0 += __popcll(b0);
                    r1 += __popcll(b1);
                    kp[q] = valid && ell > k;
                    kb[q] = __ballot(kp[q]);
                }
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (kp[q]) buf[keep + (uint32_t)__popcll(kb[q] & lt)] = e[q];   // (in front of every entry still to be read)
                    keep += (uint32_t)__popcll(kb[q]);
                }
                __builtin_amdgcn_wave_barrier();
            }
            if (lane == 0) {
                run[k] = r0;
                run[256 + k] = r1;
            }
            __builtin_amdgcn_wave_barrier();
            cur = keep;
        }
    }
}

// A section of at most 4 x 64 entries with fewer than 64 levels each, by one wave without tables: the entries stay in
// registers (entry q * 64 + lane of the section: rank order), lane k holds the counts and then the first slots of level
// k, the levels' records are placed by ballots.  Returns false (nothing written) when an entry has too many levels.
__device__ inline bool xp_small(const uint32_t *ent, uint32_t a, uint32_t cnt, int s0, int32_t *sa, int32_t *rk, int lane, uint32_t N) {
    (void)N;
    const uint64_t lt = lanemask_lt(lane);
    uint32_t e[4];
    bool v[4];
    int mx = 1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const uint32_t x = (uint32_t)q * 64 + (uint32_t)lane;
        v[q] = x < cnt;
        e[q] = v[q] ? ent[a + x] : 0u;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int ell = (int)((e[q] >> 20) & 0xFFu);
        mx = ell > mx ? ell : mx;
    }
    const int kmax = wave_max(mx);
    if (kmax > 63) return false;
    int cS = 0, cL = 0;  // lane k: entries of level k by type
    for (int k = 1; k <= kmax; ++k) {
        int s = 0, l = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t ty = (e[q] >> 28) & 1u;
            const bool in = v[q] && (int)((e[q] >> 20) & 0xFFu) >= k;
            s += __popcll(__ballot(in && ty == 0u));
            l += __popcll(__ballot(in && ty == 1u));
        }
        if (lane == k) {
            cS = s;
            cL = l;
        }
    }
    // first slots: the levels of type 0 ascending, then the levels of type 1 descending
    const int iS = wave_incl_add(cS, lane), iL = wave_incl_add(cL, lane);
    const int oS = s0 + iS - cS;
    const int oL = s0 + lane_value(iS, 63) + lane_value(iL, 63) - iL;
    for (int k = 1; k <= kmax; ++k) {
        int rS = lane_value(oS, k), rL = lane_value(oL, k);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t ty = (e[q] >> 28) & 1u;
            const bool in = v[q] && (int)((e[q] >> 20) & 0xFFu) >= k;
            const uint64_t b0 = __ballot(in && ty == 0u), b1 = __ballot(in && ty == 1u);
            if (in) {
                const int slot = ty == 0u ? rS + __popcll(b0 & lt) : rL + __popcll(b1 & lt);
                const uint32_t pos = (e[q] & MP_POS_MASK) - (uint32_t)k;
                if (MCHK(slot >= 1 && (uint32_t)slot < N && pos < N, 12, slot, e[q])) {
                    sa[slot] = (int)pos;
                    if (pos < (uint32_t)HEAD) rk[pos] = slot;
                }
            }
            rS += __popcll(b0);
            rL += __popcll(b1);
        }
    }
    return true;
}

// The levels beyond 253 of a section: its long runs (entries with the long bit, their lengths in runlen[tail]) in rank
// order, nlong records from slot `first` on - levels 254, 255, ... of the runs of type 0 ascending, then the levels of
// the runs of type 1 descending.  The long entries are first listed (tail << 32 | length << 1 | type, in `list`, which has
// room for one word per entry of the section); then 64 levels at a time, lane l the level k0 + l: a pass over the list
// counts every level's records, a second one places them.  One wave.
__device__ inline void xp_long(const uint32_t *ent, uint32_t a, uint32_t b, const uint32_t *runlen, uint64_t *list, int first,
                               int nlong, int32_t *sa, int32_t *rk, int lane, uint32_t N) {
    (void)N;
    (void)nlong;
    const uint64_t lt = lanemask_lt(lane);
    uint32_t J = 0;
    for (uint32_t x0 = a; x0 < b; x0 += 64) {
        const uint32_t x = x0 + (uint32_t)lane;
        const uint32_t e = x < b ? ent[x] : 0u;
        const bool lg = (e >> 30) & 1u;
        const uint64_t bm = __ballot(lg);
        if (lg) list[J + (uint32_t)__popcll(bm & lt)] = ((uint64_t)(e & MP_POS_MASK) << 32) | ((uint64_t)runlen[e & MP_POS_MASK] << 1) | (uint64_t)((e >> 28) & 1u);
        J += (uint32_t)__popcll(bm);
    }
    __builtin_amdgcn_s_waitcnt(0);  // the list is read back below (by this wave)
    __builtin_amdgcn_wave_barrier();
    // totals per type and the longest run
    uint32_t tS = 0, tL = 0, lmax = 0;
    for (uint32_t j0 = 0; j0 < J; j0 += 64) {
        const uint32_t j = j0 + (uint32_t)lane;
        const uint64_t it = j < J ? list[j] : 0ull;
        const uint32_t len = (uint32_t)(it >> 1) & 0x7FFFFFFFu;
        const uint32_t extra = j < J ? len - (uint32_t)(SA_RUN_CAP - 1) : 0u;
        tS += (uint32_t)lane_value(wave_incl_add((it & 1ull) ? 0 : (int)extra, lane), 63);
        tL += (uint32_t)lane_value(wave_incl_add((it & 1ull) ? (int)extra : 0, lane), 63);
        lmax = (uint32_t)wave_max((int)len) > lmax ? (uint32_t)wave_max((int)len) : lmax;
    }
    uint32_t fS = 0, pL = 0;  // records of type 0 in the levels below the block; of type 1
    for (uint32_t k0 = (uint32_t)SA_RUN_CAP; k0 <= lmax; k0 += 64) {
        const uint32_t k = k0 + (uint32_t)lane;
        int cS = 0, cL = 0;
        for (uint32_t j0 = 0; j0 < J; j0 += 64) {
            const uint32_t j = j0 + (uint32_t)lane;
            const uint64_t it = j < J ? list[j] : 0ull;
            const int len = (int)((uint32_t)(it >> 1) & 0x7FFFFFFFu), ty = (int)(it & 1ull);
            const int nn = (int)(J - j0 < 64u ? J - j0 : 64u);
            for (int i = 0; i < nn; ++i) {
                const bool in = (uint32_t)lane_value(len, i) >= k;
                if (lane_value(ty, i)) cL += in ? 1 : 0;
                else cS += in ? 1 : 0;
            }
        }
        const int iS = wave_incl_add(cS, lane), iL = wave_incl_add(cL, lane);
        int rS = first + (int)fS + iS - cS;                         // the levels below k of type 0 come first
        int rL = first + (int)tS + (int)tL - (int)pL - iL;            // the levels above k of type 1 come first
        for (uint32_t j0 = 0; j0 < J; j0 += 64) {
            const uint32_t j = j0 + (uint32_t)lane;
            const uint64_t it = j < J ? list[j] : 0ull;
            const int len = (int)((uint32_t)(it >> 1) & 0x7FFFFFFFu), ty = (int)(it & 1ull), tail = (int)(it >> 32);
            const int nn = (int)(J - j0 < 64u ? J - j0 : 64u);
            for (int i = 0; i < nn; ++i) {
                const bool in = (uint32_t)lane_value(len, i) >= k;
                const int t_ = lane_value(ty, i);
                const uint32_t pos = (uint32_t)lane_value(tail, i) - k;
                if (in) {
                    const int slot = t_ ? rL : rS;
                    if (MCHK(slot >= 1 && (uint32_t)slot < N && pos < N, 13, slot, pos)) {
                        sa[slot] = (int)pos;
                        if (pos < (uint32_t)HEAD) rk[pos] = slot;
                    }
                    if (t_) ++rL;
                    else ++rS;
                }
            }
        }
        fS += (uint32_t)lane_value(iS, 63);
        pL += (uint32_t)lane_value(iL, 63);
    }
}

template <int DB>
__global__ __launch_bounds__(SAM_THREADS) __attribute__((amdgpu_waves_per_eu(SAM_WAVES * SA_WGS_PER_CU / 4, SAM_WAVES * SA_WGS_PER_CU / 4))) void sa_meta_kernel(const Seg *segs, const uint8_t *text, SaWork wk, int max_rounds) {
    constexpr int NB = 1 << DB;
    static_assert(SaInit<DB>::DIGITS == 4 && SaInit<DB>::SHIFT == 20, "keys of two 18-bit symbols");
    static_assert(isa_binned<DB>(), "the expansion stores the ranks through the chunk streams");
    static_assert(SAM_THREADS == 512 && NB >= XP_TAB, "level tables: one entry per thread");
    __shared__ SaCarry cy;
    extern __shared__ __attribute__((aligned(16))) uint32_t dyn_lds[];
    const Seg sg = segs[blockIdx.x];
    const uint8_t *T = text + sg.off;
    const uint32_t n = sg.n;
    const uint32_t N = n + 1;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    int32_t *sa = wk.sa + sg.off, *rk = wk.isa + sg.off;
    uint32_t *mpos = wk.mpos + sg.off;
    uint32_t *pairs = wk.pairs + sg.off;

    if (N == 1) {
        if (tid == 0) {
            sa[0] = 0;
            rk[0] = 0;
        }
        return;
    }
    MPH_DECL
    const uint32_t words = sa_bitmap_words<DB>(N);
    const SaLds<DB> L(dyn_lds, words);
    uint64_t *recA = wk.recA + sg.off, *recB = wk.recB + sg.off;

    if (tid < 8) cy.fin_cnt[tid] = 0;
    if (tid < 64) cy.bin_cur[tid] = 0;
    for (uint32_t x = tid; x < words; x += SAM_THREADS) L.act[x] = 0;
    for (int x = tid; x < 7 * NB; x += SAM_THREADS) L.base[x] = 0;  // base, delta, tpos, hl, hh, hy, hsec
    for (int x = tid; x < SAM_WAVES * NB; x += SAM_THREADS) L.whist[x] = 0;
    if (SA_MTAB)
        for (int x = tid; x < SAM_WAVES * SaLds<DB>::mtab_entries(); x += SAM_THREADS) L.mtab[x] = 0;
    __syncthreads();
    MPH(0)

    // ---- sweep: run lengths (rlen) and the items, every wave over a contiguous range of 64-position blocks.  A wave
    //      appends its items (symbol << 32 | position) in text order to its own stretch of recA.
    int32_t *hY = (int32_t *)L.act, *hX = hY + NB, *hC = L.hl, *hS = L.hh;  // act is free until the first regroup
    const uint32_t nblk = (N + 63) / 64, per = (nblk + SAM_WAVES - 1) / SAM_WAVES;
    {
        const uint32_t wu = (uint32_t)wave_index(tid);
        const uint32_t bb = wu * per, be = bb + per < nblk ? bb + per : nblk;
        constexpr int G = 4;  // blocks per step: their scattered loads are in flight together (a block alone waited a latency)
        auto raw_load = [&](uint32_t blk) -> uint32_t {
            const uint32_t q = blk * 64 + (uint32_t)lane;
            uint16_t v = 0;
            if (q < N) __builtin_memcpy(&v, T + q, 2);  // (zero padding follows the text)
            return v;
        };
        auto flags_of = [&](uint32_t blk, uint32_t v) -> uint64_t {
            const uint32_t q = blk * 64 + (uint32_t)lane;
            return __ballot(q + 1 < n && (v & 0xFFu) == (v >> 8));
        };
        uint64_t *stage = recA + (uint64_t)bb * 64;
        uint32_t cntw = 0;
        if (bb < be) {
            // the position before the wave's first: is it equal to it, and how far is it from the first byte of its run
            uint32_t dcar = 0;
            bool eqcar = false;
            if (bb > 0) {
                const uint32_t q0 = bb * 64 - 1;
                eqcar = q0 + 1 < n && T[q0] == T[q0 + 1];
                for (uint32_t q = q0;;) {  // lane l: is position q - l equal to the one before it?
                    const bool eq = q >= (uint32_t)lane + 1 && T[q - (uint32_t)lane - 1] == T[q - (uint32_t)lane];
                    const uint64_t bm = __ballot(eq);
                    const int c = bm == ~0ull ? 64 : (int)__builtin_ctzll(~bm);
                    dcar += (uint32_t)c;
                    if (c < 64 || q < 64) break;
                    q -= 64;
                }
            }
            const uint64_t le = lane == 63 ? ~0ull : ((2ull << lane) - 1ull);
            uint32_t rw[2 * G], rn[G];
            uint64_t E[2 * G];
#pragma unroll
            for (int i = 0; i < 2 * G; ++i) rw[i] = raw_load(bb + i);
#pragma unroll
            for (int i = 0; i < G; ++i) rn[i] = raw_load(bb + 2 * G + i);
#pragma unroll
            for (int i = 0; i < 2 * G; ++i) E[i] = flags_of(bb + i, rw[i]);
            for (uint32_t c = bb; c < be; c += G) {
                int rl_[G];
                bool it_[G], hd_[G];
                uint32_t xv[G];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const uint32_t e = (c + g) * 64 + (uint32_t)lane;
                    const uint64_t E0 = E[g];
                    const uint64_t x0 = E0 >> lane;
                    int rl = (x0 == ~0ull) ? 64 : (int)__builtin_ctzll(~x0);
                    if (rl == 64 - lane) {
#pragma unroll
                        for (int t = 1; t <= 4; ++t) {
                            const int rt = E[g + t] == ~0ull ? 64 : (int)__builtin_ctzll(~E[g + t]);
                            rl += rt;
                            if (rt < 64) break;
                        }
                    }
                    rl_[g] = rl + 1;  // exact up to 257
                    // bytes of the run before this position: from the flags of the block, or carried over from the block before
                    const uint64_t P = (E0 << 1) | (eqcar ? 1ull : 0ull);  // bit l: position e equals the one before it
                    const uint64_t zl = ~P & le;
                    const uint32_t d = zl ? (uint32_t)lane - (uint32_t)(63 - (int)__builtin_clzll(zl)) : dcar + 1u + (uint32_t)lane;
                    dcar = (uint32_t)lane_value((int)d, 63);
                    eqcar = (E0 >> 63) != 0;
                    // an item: the first byte of a run and every 253rd byte behind it
                    it_[g] = c + g < be && e < N && d % (uint32_t)(SA_RUN_CAP - 1) == 0u;
                    hd_[g] = d == 0u;
                    xv[g] = 0;
                    if (it_[g] && rl_[g] < SA_RUN_CAP) xv[g] = T[e + (uint32_t)rl_[g]];  // (the text is zero padded)
                }
                uint64_t En[G];
                uint32_t rq[G];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    En[g] = flags_of(c + 2 * G + g, rn[g]);
                    rq[g] = rn[g];
                    rn[g] = raw_load(c + 3 * G + g);
                }
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    if (c + g >= be) break;
                    const uint32_t e = (c + g) * 64 + (uint32_t)lane;
                    const int rl = rl_[g];
                    const bool valid = e < N;
                    if (valid) wk.rlen[sg.off + e] = (uint8_t)(rl < 255 ? rl : 255);
                    uint32_t sym = 0;  // the sentinel's
                    if (valid && e < n) {
                        const uint32_t dS = (rw[g] & 0xFFu) + 1u;
                        uint32_t dC = 255u;  // 253 bytes of a run that goes on
                        if (rl < SA_RUN_CAP) {
                            const uint32_t q = e + (uint32_t)rl;
                            const uint32_t dX = q < n ? xv[g] + 1u : 0u;
                            dC = dX < dS ? (uint32_t)rl : (uint32_t)(511 - rl);
                        }
                        sym = (dS << 9) | dC;
                    }
                    const uint64_t bm = __ballot(it_[g]);
                    if (it_[g]) stage[cntw + (uint32_t)__popcll(bm & lanemask_lt(lane))] = ((uint64_t)sym << 32) | (uint64_t)e | (hd_[g] ? 0ull : 1ull << 31);
                    cntw += (uint32_t)__popcll(bm);
                }
#pragma unroll
                for (int i = 0; i < G; ++i) {
                    E[i] = E[G + i];
                    rw[i] = rw[G + i];
                    E[G + i] = En[i];
                    rw[G + i] = rq[i];
                }
            }
        }
        if (lane == 0) cy.wcount[w] = (int32_t)cntw;
    }
    __syncthreads();
    MPH(1)

    // ---- the items get their index (text order), their position word and their record: key = the item's symbol and the
    //      next item's (36 bits) << 20 | index; the digits of the keys are counted for the four passes
    uint32_t base_w = 0, NM = 0;
    for (int ww = 0; ww < SAM_WAVES; ++ww) {
        const uint32_t c = (uint32_t)cy.wcount[ww];
        if (ww < w) base_w += c;
        NM += c;
    }
    {
        const uint32_t wu = (uint32_t)wave_index(tid);
        const uint64_t *stage = recA + (uint64_t)wu * per * 64;
        const uint32_t cntw = (uint32_t)cy.wcount[wu];
        uint64_t wnext = 0;  // the first item behind this wave's last one (none: the last item is the sentinel)
        for (uint32_t ww = wu + 1; ww < (uint32_t)SAM_WAVES; ++ww)
            if (cy.wcount[ww] > 0) {
                wnext = recA[(uint64_t)ww * per * 64];
                break;
            }
        for (uint32_t j00 = 0; j00 < cntw; j00 += 256) {  // four steps' loads in flight (a step alone waited a memory latency)
            uint64_t itq[4], scq[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t j = j00 + (uint32_t)q * 64 + (uint32_t)lane;
                itq[q] = j < cntw ? stage[j] : 0ull;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t jn = j00 + (uint32_t)(q + 1) * 64;
                scq[q] = q < 3 ? lane_value(itq[q < 3 ? q + 1 : q], 0) : (jn < cntw ? stage[jn] : wnext);
                if (jn >= cntw) scq[q] = wnext;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const uint32_t j0 = j00 + (uint32_t)q * 64;
                if (j0 >= cntw) break;
                const uint32_t j = j0 + (uint32_t)lane;
                const bool valid = j < cntw;
                const uint64_t it = itq[q], succ = scq[q];
                uint64_t nx;
                {
                    const uint32_t lo = (uint32_t)lane_above((int)(uint32_t)it, (int)(uint32_t)succ);
                    const uint32_t hi = (uint32_t)lane_above((int)(uint32_t)(it >> 32), (int)(uint32_t)(succ >> 32));
                    nx = (uint64_t)lo | ((uint64_t)hi << 32);
                }
                if (j + 1 == cntw) nx = wnext;
                const uint32_t symA = (uint32_t)(it >> 32), e = (uint32_t)it & MP_POS_MASK, symB = (uint32_t)(nx >> 32);
                const uint32_t cont = ((uint32_t)it >> 31) & 1u;  // the item continues a run (behind a full chunk)
                const uint32_t i = base_w + j;
                if (valid) {
                    mpos[i] = e | ((((symA >> 9) - 1u) & 0xFFu) << MP_BYTE_SHIFT) | ((symA & 511u) == 255u ? 1u << MP_DEEP_SHIFT : 0u) | (cont << MP_CONT_SHIFT);
                    recB[i] = ((((uint64_t)symA << 18) | (uint64_t)symB) << 20) | (uint64_t)i;
                }
                // the digits of the item's own symbol are counted; the next item's (the low half of the key) follow from
                // them below.  (Five leader-vote counts per step were 8 % of the kernel: LDS atomics of many lanes on one word.)
                digit_count(hC, (int)(symA & 511u), valid, lane);
                digit_count(hS, (int)(symA >> 9), valid, lane);
                // full chunks (253 bytes of a run that goes on): they end no run, so they stand for no entry of the expansion
                const bool full = valid && (symA & 511u) == 255u;
                if (__ballot(full)) digit_count(L.hsec, (int)(symA >> 9), full, lane);
            }
        }
    }
    __syncthreads();
    {   // the low half of item i's key is the symbol of item i + 1 (nothing for the last item, the sentinel): its digits
        // are the symbols' digits without the first item's and with one more zero.  Sections of the expansion: one entry
        // per run in the section of its byte + 1 - every item but the sentinel (digit 0) and the full chunks.
        const uint32_t sym0 = (uint32_t)(recA[0] >> 32);  // (position 0 is an item, the first of wave 0's)
        for (int x = tid; x < NB; x += SAM_THREADS) {
            hX[x] = hS[x] + (x == 0 ? 1 : 0) - (x == (int)(sym0 >> 9) ? 1 : 0);
            hY[x] = hC[x] + (x == 0 ? 1 : 0) - (x == (int)(sym0 & 511u) ? 1 : 0);
            L.hsec[x] = x >= 1 ? hS[x] - L.hsec[x] : 0;
        }
    }
    __syncthreads();
    MPH(2)

    // ---- the items' suffixes: initial order by the keys, regroup, small groups by direct comparison, doubling rounds
    sa_initial_sort<DB>(L, recB, recA, NM, hY, hY, hX, hC, hS, tid);
    MPH(3)
    const uint64_t *rcur = recB;  // four passes
    for (int x = tid; x < 2 * NB; x += SAM_THREADS) L.hl[x] = 0;  // hl, hh
    for (uint32_t x = tid; x < words; x += SAM_THREADS) L.act[x] = 0;  // held two digit histograms so far
    __syncthreads();
    uint32_t m = sa_regroup<true, DB>(L, cy, rcur, NM, sa, rk, tid, (uint32_t *)recA, NM, pairs);
    MPH(4)
#ifdef SA_PROFILE_ROUNDS
    if (blockIdx.x == 0 && tid == 0) printf("meta: N=%u items=%u shared after the keys=%u\n", N, NM, m);
#endif
    m -= sa_finish<DB>(L, cy, (const uint32_t *)recA, NM, sa, rk, T, n, 2, tid, pairs, mpos);
#ifdef SA_PROFILE_ROUNDS
    if (blockIdx.x == 0 && tid == 0) printf("meta: shared after the finisher=%u\n", m);
#endif
    MPH(5)
    isa_flush<DB>(L, pairs, NM, rk, tid);
    MPH(6)
    sa_rounds<DB>(L, cy, recA, recB, sa, rk, NM, words, m, 2u, T, n, mpos, tid, max_rounds);
    __syncthreads();
    MPH(7)

    // ---- expansion, pass P: in rank order, an item u that starts a run (or is the sentinel) is the tail of the run before
    //      it: the entries  tail position | min(run length, 253) << 20 | type << 28 (0: the tail's byte is smaller than the
    //      run's, or the text ends; 1: larger) | long << 30 (a run of more than 253 bytes: its length goes to
    //      runlen[tail]), scattered by the run's byte + 1; the bytes of every section are summed
    const uint32_t m1 = NM - 1;            // the sentinel's item
    uint32_t *ent = (uint32_t *)recA;
    uint32_t *runlen = pairs;              // (idle since the items' ranks were flushed)
    int32_t *secbytes = L.tpos;
    for (int x = tid; x < NB; x += SAM_THREADS) secbytes[x] = 0;
    if (tid == 0) cy.next_sec = 1;
    __syncthreads();
    bins_scan(L.hsec, L.base, NB, tid);
    uint32_t un[SA_K];
#pragma unroll
    for (int s_ = 0; s_ < SA_K; ++s_) {
        const uint32_t r = (uint32_t)w * SA_WSPAN + s_ * 64 + lane;
        un[s_] = r < NM ? (uint32_t)sa[r] & 0x7FFFFFFFu : 0u;
    }
    for (uint32_t t0 = 0; t0 < NM; t0 += SA_TILE) {
        const uint32_t wb = t0 + (uint32_t)w * SA_WSPAN;
        uint32_t eb[SA_K];
        int db[SA_K], lb[SA_K];
        bool vb[SA_K];
        uint32_t u[SA_K], w0[SA_K], w1[SA_K];
#pragma unroll
        for (int s_ = 0; s_ < SA_K; ++s_) {  // the position words of the item and of the item before it (all in flight)
            const uint32_t r = wb + s_ * 64 + lane;
            u[s_] = un[s_];
            w0[s_] = 0;
            w1[s_] = 0;
            if (r < NM && MCHK(u[s_] < NM, 20, u[s_], r)) {
                w1[s_] = mpos[u[s_]];
                if (u[s_] >= 1) w0[s_] = mpos[u[s_] - 1];
            }
        }
        if (t0 + SA_TILE < NM) {  // the next tile's ranks, loaded while this tile is placed
#pragma unroll
            for (int s_ = 0; s_ < SA_K; ++s_) {
                const uint32_t r = wb + SA_TILE + s_ * 64 + lane;
                un[s_] = r < NM ? (uint32_t)sa[r] & 0x7FFFFFFFu : 0u;
            }
        }
#pragma unroll
        for (int s_ = 0; s_ < SA_K; ++s_) {
            const uint32_t r = wb + s_ * 64 + lane;
            const bool ok = r < NM;
            const uint32_t pos_u = w1[s_] & MP_POS_MASK, c_u = (w1[s_] >> MP_BYTE_SHIFT) & 0xFFu, c_p = (w0[s_] >> MP_BYTE_SHIFT) & 0xFFu;
            const bool sentinel = u[s_] == m1;
            // the item before u ends its run (a chunk of 253 bytes that is followed by the same byte does not)
            vb[s_] = ok && u[s_] >= 1 && !((w0[s_] >> MP_DEEP_SHIFT) & 1u);
            uint32_t ell = pos_u - (w0[s_] & MP_POS_MASK);   // the run's last chunk
            uint32_t lng = 0;
            if (vb[s_] && ((w0[s_] >> MP_CONT_SHIFT) & 1u)) {  // ... behind full chunks: a long run (rare)
                uint32_t len = ell + (uint32_t)(SA_RUN_CAP - 1);
                for (uint32_t v = u[s_] - 2; v >= 1 && ((mpos[v - 1] >> MP_DEEP_SHIFT) & 1u); --v) len += (uint32_t)(SA_RUN_CAP - 1);
                runlen[pos_u] = len;
                lng = 1u << 30;
                atomicAdd(&secbytes[(int)c_p + 1], (int)len);
                ell = (uint32_t)(SA_RUN_CAP - 1);
            } else if (vb[s_]) {
                atomicAdd(&secbytes[(int)c_p + 1], (int)ell);
            }
            db[s_] = (int)c_p + 1;
            eb[s_] = pos_u | (ell << 20) | ((sentinel || c_u < c_p) ? 0u : 1u << 28) | lng;
            lb[s_] = 0;
            if (__ballot(vb[s_])) lb[s_] = tile_rank<9, DB>(L, w, db[s_], vb[s_], lane);
        }
        tile_prepare<DB>(L, tid);
#pragma unroll
        for (int s_ = 0; s_ < SA_K; ++s_) {
            if (vb[s_]) {
                const int at = L.delta[db[s_]] + (int)L.woff[SaLds<DB>::cell(db[s_], w)] + lb[s_];
                if (MCHK(at >= 0 && (uint32_t)at < m1, 21, at, eb[s_])) ent[at] = eb[s_];
            }
        }
    }
    __syncthreads();
    MPH(8)
    // L.base[d] is now the END of section d's entries; the sections' first slots: 1 + the bytes of the sections before
    bins_scan(secbytes, L.delta, NB, tid);

    // ---- expansion, pass X.  Scratch: the scatter tables (whist, woff, the peer table: 48 KiB, idle from here on) hold
    //      the level tables of a section expanded by all waves (sh[wave][XP_TAB]) and, per wave, a table of its own and
    //      its list of XP_W entries
    int32_t *sh = (int32_t *)L.whist;
    int32_t *run = sh + SAM_WAVES * XP_TAB + w * (XP_TAB + XP_W);
    uint32_t *buf = (uint32_t *)(run + XP_TAB);
    int32_t *tot = L.hl;   // [XP_TAB] (hl, hh: idle)
    if (tid == 0) {
        sa[0] = (int)n;
        if (n < (uint32_t)HEAD) rk[n] = 0;
    }
    MPH(9)
    // large sections: every wave a slice
    for (int d = 1; d <= 256; ++d) {
        const uint32_t cnt = (uint32_t)L.hsec[d];
        if (cnt < XP_BIG) continue;
        const uint32_t a0 = (uint32_t)L.base[d] - cnt;
        const uint32_t pw = ((cnt + SAM_WAVES - 1) / SAM_WAVES + 63u) & ~63u;
        const uint32_t a = a0 + min((uint32_t)w * pw, cnt), b = a0 + min((uint32_t)(w + 1) * pw, cnt);
        int32_t *mine = sh + w * XP_TAB;
        for (int x = lane; x < XP_TAB; x += 64) mine[x] = 0;
        __builtin_amdgcn_wave_barrier();
        (void)xp_count(ent, a, b, mine, lane);
        xp_levels(mine, lane, 3);
        __syncthreads();
        {   // per level: the waves' counts -> their offsets inside the level, and the level's total
            int r = 0;
            for (int ww = 0; ww < SAM_WAVES; ++ww) {
                const int c = sh[ww * XP_TAB + tid];
                sh[ww * XP_TAB + tid] = r;
                r += c;
            }
            tot[tid] = r;
        }
        __syncthreads();
        if (w == 0) xp_offsets(tot, tot, 1 + L.delta[d], secbytes[d], lane, 3);
        __syncthreads();
        for (int x = lane; x < XP_TAB; x += 64) run[x] = tot[x] + mine[x];
        __builtin_amdgcn_wave_barrier();
        xp_expand(ent, a, b, run, buf, sa, rk, lane, N);
        if (w == 0 && tot[254] > 0) xp_long(ent, a0, a0 + cnt, runlen, recB + a0, tot[255], tot[254], sa, rk, lane, N);
        __syncthreads();
    }
    MPH(10)
    // the other sections: a wave takes the next one when it is free
    for (;;) {
        int d = 0;
        if (lane == 0) d = atomicAdd(&cy.next_sec, 1);
        d = lane_value(d, 0);
        if (d > 256) break;
        const uint32_t cnt = (uint32_t)L.hsec[d];
        if (cnt == 0 || cnt >= XP_BIG) continue;
        const uint32_t a = (uint32_t)L.base[d] - cnt, b = a + cnt;
        if (cnt <= 256u && xp_small(ent, a, cnt, 1 + L.delta[d], sa, rk, lane, N)) continue;
        for (int x = lane; x < XP_TAB; x += 64) run[x] = 0;
        __builtin_amdgcn_wave_barrier();
        const int qmax = xp_count(ent, a, b, run, lane) >> 6;
        xp_levels(run, lane, qmax);
        xp_offsets(run, run, 1 + L.delta[d], secbytes[d], lane, qmax);
        const int lfirst = run[255], nlong = run[254];
        xp_expand(ent, a, b, run, buf, sa, rk, lane, N);
        if (nlong > 0) xp_long(ent, a, b, runlen, recB + a, lfirst, nlong, sa, rk, lane, N);
    }
    MPH(11)
    // (no inverse array: the items' ranks have used its buffer, and nothing behind the suffix sort reads more of it than
    // the ranks of the first HEAD positions, stored above; launch_isa_from_sa fills it for the index views of the tests)
    MPH_PRINT
}

// ---- window index: ranks regrouped by the 16 KiB chunk of their position (stable in rank order) ------
// Lets the match kernel answer "first rank >= lo whose position lies in [i - 2^14, i)" without walking
// the suffix array: the window meets at most two chunks, and inside a chunk the condition is one-sided.
constexpr int WIN_SHIFT = LZSS_SHORT_ADDR_BITS;  // chunk = 2^14 positions

// lanes of `amask` whose chunk number equals this lane's: unrolled for the common widths (8 chunks = texts of 128 KiB)
__device__ inline uint64_t chunk_peers(int c, uint64_t amask, int cbits) {
    if (cbits <= 3) return match_bits<3>(c, amask);
    if (cbits <= 5) return match_bits<5>(c, amask);
    return match_digit_n(c, amask, cbits);
}

__global__ __launch_bounds__(SA_THREADS) void window_index_kernel(const Seg *segs, BatchIndex ix, uint32_t skip_above) {
    __shared__ SaShared sh;
    const Seg sg = segs[blockIdx.x];
    const uint32_t n = sg.n;
    if (n == 0 || n + 1 > skip_above) return;  // (texts of more suffixes than skip_above: launch_window_large)
    const uint32_t N = n + 1;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int32_t *SA = ix.sa + sg.off;
    int32_t *KS = ix.ks(sg), *PS = ix.ps(sg);
    const int nch = (int)((n + (1u << WIN_SHIFT) - 1) >> WIN_SHIFT);  // <= 256 for n <= 4 MiB

    // ranks 1..n (rank 0 is the sentinel), one chunk of the rank range per wave
    const uint32_t C = (((n + SA_WAVES - 1) / SA_WAVES) + 63u) & ~63u;
    const uint32_t cbeg = 1 + min((uint32_t)w * C, n), cend = min(cbeg + C, N);

    const int cbits = nch > 1 ? 32 - __clz(nch - 1) : 0;  // bits of a chunk number
    hist_clear(sh, tid);
    __syncthreads();
    // (both sweeps load a step's suffix-array entries one step ahead: a wave's steps were one memory latency each)
    int sa_next = cbeg + lane < cend ? SA[cbeg + lane] : 0;
    for (uint32_t k0 = cbeg; k0 < cend; k0 += 64) {
        const uint32_t k = k0 + lane;
        const bool active = k < cend;
        const int c = active ? (sa_next >> WIN_SHIFT) : 0;
        sa_next = k + 64 < cend ? SA[k + 64] : 0;
        const uint64_t peers = chunk_peers(c, __ballot(active), cbits);
        if (active && (peers & lanemask_lt(lane)) == 0) sh.hist[w][c] += __popcll(peers);
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
    uint16_t *cdir = ix.cdir + (sg.off / 64) * ix.nch;
    sa_next = cbeg + lane < cend ? SA[cbeg + lane] : 0;
    for (uint32_t k0 = cbeg; k0 < cend; k0 += 64) {
        const uint32_t k = k0 + lane;
        const bool active = k < cend;
        const int p = active ? sa_next : 0;
        sa_next = k + 64 < cend ? SA[k + 64] : 0;
        const int c = p >> WIN_SHIFT;
        // directory: entries of every chunk before rank k0 (k0 == 1 + 64 b; the waves' rank ranges are whole blocks)
        for (int cc = lane; cc < nch; cc += 64) cdir[((k0 - 1) >> 6) * ix.nch + cc] = (uint16_t)(hrow[cc] - (cc << WIN_SHIFT));
        __builtin_amdgcn_wave_barrier();
        const uint64_t amask = __ballot(active);
        const uint64_t peers = chunk_peers(c, amask, cbits);
        const int cnt = __popcll(peers);
        const int rank = __popcll(peers & lanemask_lt(lane));
        const int base = hrow[c];
        if (active) {
            if (ix.win_packed) KS[base + rank] = (int)((k << WIN_SHIFT) | ((uint32_t)p & ((1u << WIN_SHIFT) - 1u)));
            else {
                KS[base + rank] = (int)k;
                PS[base + rank] = p;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (active && rank == cnt - 1) hrow[c] = base + cnt;
        __builtin_amdgcn_wave_barrier();
    }
    // min / max of the positions over blocks of 64 and of 4096 entries, from the entries this workgroup has just
    // written (they are still in its L2): a wave per block of 64, then a wave per 64 summaries
    __syncthreads();
    const int32_t *E = ix.win_packed ? KS : PS;
    const int emask = ix.win_packed ? (1 << WIN_SHIFT) - 1 : -1;
    int32_t *pmin1 = ix.pmin1 + sg.off / 64, *pmax1 = ix.pmax1 + sg.off / 64;
    int32_t *pmin2 = ix.pmin2 + sg.off / 64, *pmax2 = ix.pmax2 + sg.off / 64;
    const uint32_t nb1 = (n + 63) / 64, nb2 = (nb1 + 63) / 64;
    int e_next = (uint32_t)w * 64 + lane < n ? E[(uint32_t)w * 64 + lane] : 0;
    for (uint32_t b = w; b < nb1; b += SA_WAVES) {
        const uint32_t j = b * 64 + lane;
        int lo = INT_MAX, hi = INT_MIN;
        if (j < n) lo = hi = e_next & emask;
        e_next = j + 64 * SA_WAVES < n ? E[j + 64 * SA_WAVES] : 0;
        lo = wave_min(lo);
        hi = wave_max(hi);
        if (lane == 0) {
            pmin1[b] = lo;
            pmax1[b] = hi;
        }
    }
    __syncthreads();
    for (uint32_t b = w; b < nb2; b += SA_WAVES) {
        const uint32_t j = b * 64 + lane;
        int lo = INT_MAX, hi = INT_MIN;
        if (j < nb1) {
            lo = pmin1[j];
            hi = pmax1[j];
        }
        lo = wave_min(lo);
        hi = wave_max(hi);
        if (lane == 0) {
            pmin2[b] = lo;
            pmax2[b] = hi;
        }
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
    v = IS_MAX ? wave_max(v) : wave_min(v);
    if ((threadIdx.x & 63) == 0) out[sg.off / 64 + (j >> 6)] = v;
}

// The inverse suffix array of whole segments, isa[sa[k]] = k.  The run-compressed sort leaves only the ranks of the first
// HEAD positions behind (all that the match kernels read); the index views of the C ABI (mi355_lzss_batch_index,
// tests) call this for the segment they hand out.
__global__ __launch_bounds__(256) void isa_from_sa_kernel(const Seg *segs, const int32_t *sa, int32_t *isa) {
    const Seg sg = segs[blockIdx.y];
    const uint32_t k = blockIdx.x * 256 + threadIdx.x;
    if (k <= sg.n) isa[sg.off + (uint32_t)sa[sg.off + k]] = (int32_t)k;
}
void launch_isa_from_sa(const Seg *d_segs, uint32_t n_segs, uint32_t max_n, const int32_t *d_sa, int32_t *d_isa, hipStream_t stream) {
    if (n_segs == 0) return;
    hipLaunchKernelGGL(isa_from_sa_kernel, dim3((max_n + 256) / 256, n_segs), dim3(256), 0, stream, d_segs, d_sa, d_isa);
}

void launch_window_index(const Seg *d_segs, uint32_t n_segs, uint32_t max_n, const BatchIndex &ix, hipStream_t stream,
                         bool index, bool summaries, const Seg *h_segs, const SaWork *wk) {
    if (n_segs == 0 || max_n == 0) return;
    if (index) {
        const bool large = h_segs && wk && wk->tmp && max_n + 1 > SA_BITMAP_MAX_N;
        hipLaunchKernelGGL(window_index_kernel, dim3(n_segs), dim3(SA_THREADS), 0, stream, d_segs, ix,
                           large ? SA_BITMAP_MAX_N : 0xFFFFFFFFu);
        if (large)
            for (uint32_t k = 0; k < n_segs; ++k)
                if (h_segs[k].n + 1 > SA_BITMAP_MAX_N && launch_window_large(h_segs[k], ix, *wk, stream) != 0)
                    hipLaunchKernelGGL(window_index_kernel, dim3(1), dim3(SA_THREADS), 0, stream, d_segs + k, ix, 0xFFFFFFFFu);
    }
    if (!summaries) return;
    const uint32_t c0 = max_n + 1, c1 = (c0 + 63) / 64;
    dim3 g1((c1 + 255) / 256, n_segs);
    hipLaunchKernelGGL(block64_kernel<false>, g1, dim3(256), 0, stream, d_segs, ix.samin1, ix.samin2, 1, 1);  // samin1: lcp_kernel
}

static void launch_sa_small(const Seg *d_segs, uint32_t n_segs, uint32_t max_n, const uint8_t *d_text, const SaWork &wk,
                            hipStream_t stream) {
#ifdef LZSS_DEBUG_KNOBS  // timing experiments only (scripts/build_variants.sh); the shipped library ignores the environment
    static const int max_rounds = getenv("MI355_LZSS_SA_ROUNDS") ? atoi(getenv("MI355_LZSS_SA_ROUNDS")) : 1000;
#else
    constexpr int max_rounds = 1000;
#endif
    const uint32_t N = max_n + 1;
    // both instantiations may use more than the default 64 KiB of dynamic LDS; the attribute belongs to the function on
    // ONE device, so it is set once per device (several contexts on several GPUs may live in one process)
    static std::atomic<uint64_t> attr_devs{0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t dbit = 1ull << (dev & 63);
    if (!(attr_devs.load(std::memory_order_acquire) & dbit)) {
        (void)hipFuncSetAttribute((const void *)sa_build_kernel<9>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)SaLds<9>::bytes(sa_bitmap_words<9>(1u << 18)));
        (void)hipFuncSetAttribute((const void *)sa_build_kernel<10>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)SaLds<10>::bytes(sa_bitmap_words<10>(SA_BITMAP_MAX_N)));
        (void)hipFuncSetAttribute((const void *)sa_meta_kernel<9>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)SaLds<9>::bytes(sa_bitmap_words<9>(1u << 18)));
        (void)hipFuncSetAttribute((const void *)sa_meta_kernel<10>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)SaLds<10>::bytes(sa_bitmap_words<10>(SA_BITMAP_MAX_N)));
        attr_devs.fetch_or(dbit, std::memory_order_release);
    }
#ifdef LZSS_DEBUG_KNOBS
    static const bool plain = getenv("MI355_LZSS_SA_PLAIN") != nullptr;  // the sort of every suffix, for comparisons
#else
    constexpr bool plain = false;
#endif
    if (SA_META && wk.mpos && !plain) {  // run-compressed sort
        if (N <= (1u << 18)) {
            hipLaunchKernelGGL(sa_meta_kernel<9>, dim3(n_segs), dim3(SAM_THREADS), SaLds<9>::bytes(sa_bitmap_words<9>(N)), stream,
                               d_segs, d_text, wk, max_rounds);
        } else {
            hipLaunchKernelGGL(sa_meta_kernel<10>, dim3(n_segs), dim3(SAM_THREADS), SaLds<10>::bytes(sa_bitmap_words<10>(N)), stream,
                               d_segs, d_text, wk, max_rounds);
        }
        return;
    }
    if (N <= (1u << 18)) {
        hipLaunchKernelGGL(sa_build_kernel<9>, dim3(n_segs), dim3(SAM_THREADS), SaLds<9>::bytes(sa_bitmap_words<9>(N)), stream,
                           d_segs, d_text, wk, max_rounds);
    } else {
        hipLaunchKernelGGL(sa_build_kernel<10>, dim3(n_segs), dim3(SAM_THREADS), SaLds<10>::bytes(sa_bitmap_words<10>(N)), stream,
                           d_segs, d_text, wk, max_rounds);
    }
}

int launch_sa_build(const Seg *d_segs, const Seg *h_segs, uint32_t n_segs, uint32_t max_n, const uint8_t *d_text, const SaWork &wk,
                    hipStream_t stream) {
    if (n_segs == 0) return 0;
    if (max_n + 1 <= SA_BITMAP_MAX_N) {
        launch_sa_small(d_segs, n_segs, max_n, d_text, wk, stream);
        return 0;
    }
    // a batch with large texts: runs of smaller ones go to the one-workgroup kernels, every large one to the device-wide sort
    if (!h_segs || !wk.tmp) return (int)hipErrorInvalidValue;
    for (uint32_t k = 0; k < n_segs;) {
        if (h_segs[k].n + 1 > SA_BITMAP_MAX_N) {
            const int rc = launch_sa_large(h_segs[k], d_text, wk, stream);
            if (rc != 0) return rc;
            hipLaunchKernelGGL(rlen_kernel, dim3((h_segs[k].n + 256) / 256, 1), dim3(256), 0, stream, d_segs + k, d_text, wk.rlen);
            ++k;
            continue;
        }
        uint32_t e = k, mx = 0;
        while (e < n_segs && h_segs[e].n + 1 <= SA_BITMAP_MAX_N) {
            mx = h_segs[e].n > mx ? h_segs[e].n : mx;
            ++e;
        }
        launch_sa_small(d_segs + k, e - k, mx, d_text, wk, stream);
        k = e;
    }
    return 0;
}

}  // namespace lzss
