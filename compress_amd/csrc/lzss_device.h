// lzss_device.h - data layout shared by the HIP kernels and the C-ABI host code (gfx950 only).
//
// A *segment* is one indexed text: a blob's whole inBuf (lzss/compress.go:113-116) or the augmented
// dictionary (compress.go:64).  All per-position arrays of a segment live at the same element offset
// `off` inside batch-wide arrays, so one launch covers every blob of a batch.
//
// Index arrays are in the "N-world": N = n + 1 entries, rank 0 is the implicit end sentinel of
// suffixarray.New (a suffix that is a prefix of another sorts first), i.e. SA[0] == n and the
// reference's suffix array is SA[1..N).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/lzss_format.h"

namespace lzss {

struct Seg {
    uint64_t off;    // element offset of this segment in every per-position array (multiple of 256)
    uint32_t n;      // text length in bytes
    uint32_t start;  // first position to encode (Write's lastInLen, compress.go:118)
};

constexpr uint32_t SEG_ALIGN = 256;   // segment stride granularity; >= 16 bytes of zero padding follow each text
constexpr uint32_t SEG_PAD = 64;      // stride = align(n + SEG_PAD)
#ifndef LZSS_HEAD
#define LZSS_HEAD 256
#endif
__host__ __device__ constexpr inline uint64_t seg_stride(uint64_t n) { return (n + SEG_PAD + SEG_ALIGN - 1) / SEG_ALIGN * SEG_ALIGN; }
constexpr int HEAD = LZSS_HEAD;         // positions below HEAD are matched by brute force (match_head_kernel)

// packed per-position result of bestBackref (compress.go:146-173)
constexpr uint32_t RES_NONE = 0u;
constexpr uint32_t RES_SHORT = 1u;
constexpr uint32_t RES_DYN = 2u;
constexpr uint32_t RES_DICT = 3u;  // a dynamic reference into the dictionary whose address bits are pending: field = a dictionary rank
// type:2 | pending:1 | length-1:8 | field:21.  field = the address bits as written (backref.go:66), or, while the
// result is PENDING, the rank the search for the smallest-rank occurrence starts from (window_search.h): emit_kernel
// finishes pending results for the phrases it writes.  Ranks need 21 bits, so only texts below 2 MiB are deferred.
constexpr uint32_t RES_PENDING = 1u << 29;
constexpr uint32_t RES_DEFER_MAX_N = 1u << 21;  // N = n + 1 must stay below
__host__ __device__ inline uint32_t res_pack(uint32_t type, uint32_t len, uint32_t field) {
    return type == RES_NONE ? 0u : (type << 30) | (((len - 1u) & 0xFFu) << 21) | (field & 0x1FFFFFu);
}
__host__ __device__ inline uint32_t res_type(uint32_t r) { return r >> 30; }
__host__ __device__ inline uint32_t res_len(uint32_t r) { return (r >> 30) ? ((r >> 21) & 0xFFu) + 1u : 0u; }
__host__ __device__ inline bool res_pending(uint32_t r) { return (r & RES_PENDING) != 0; }
__host__ __device__ inline uint32_t res_field(uint32_t r) { return r & 0x1FFFFFu; }

// phrase kinds chosen by the walk (compress.go:176-257)
enum : uint8_t {
    K_LIT = 0,        // one literal                               8 bits
    K_LIT2 = 1,       // two literals (lookahead at i+2)           16 bits
    K_SHORT = 2,      // short back-reference                      30 bits
    K_DYN = 3,        // long back-reference                       37 bits
    K_RLE_CONT = 4,   // RLE, previous byte equal: short ref off 0 30 bits
    K_RLE_LIT = 5,    // RLE: literal + short ref                  38 bits
    K_RLE_RES = 6,    // RLE of a reserved symbol: dict ref + short ref  67 bits
};
__host__ __device__ inline uint32_t kind_bits(uint8_t k) {
    switch (k) {
        case K_LIT: return 8;
        case K_LIT2: return 16;
        case K_SHORT: return LZSS_SHORT_BITS;
        case K_DYN: return LZSS_DYN_BITS;
        case K_RLE_CONT: return LZSS_SHORT_BITS;
        case K_RLE_LIT: return 8 + LZSS_SHORT_BITS;
        default: return LZSS_DYN_BITS + LZSS_SHORT_BITS;
    }
}

// suffix-array index of one text as the match kernels see it
struct IndexView {
    const uint8_t *T;       // text, readable 16 bytes past n
    uint32_t n;             // text length; N = n + 1
    const int32_t *SA;      // N entries
    const int32_t *ISA;     // N entries (rank of every suffix)
    const uint16_t *LCP;    // N entries, LCP[k] = lcp(SA[k-1], SA[k]) capped at 256; LCP[0] = 0
    const uint16_t *bmin1;  // min of LCP over blocks of 64 ranks
    const uint16_t *bmin2;  // min of bmin1 over blocks of 64 blocks
};

// XCD-aware block order.  Blocks b and b + 8 of a launch land on the same XCD (MI355X_MICROARCH.md, workgroup
// dispatch), each XCD has its own 4 MiB L2, and the index arrays of one 128 KiB segment are ~2.5 MB.  Launching a
// 1-D grid of nblk (multiple of 8) blocks and working on virtual block (b % 8) * (nblk / 8) + b / 8 lets every XCD
// walk a contiguous eighth of the (segment, block) space, so a segment's random accesses meet in ONE L2.
__device__ inline uint32_t xcd_block(uint32_t b, uint32_t nblk) { return (b & 7u) * (nblk >> 3) + (b >> 3); }
inline uint32_t xcd_grid(uint64_t blocks) { return (uint32_t)((blocks + 7) / 8 * 8); }

// ---- wave-level data movement without the LDS crossbar ------------------------------------------------
// A thread's wave inside its workgroup as a value the compiler KNOWS to be wave-uniform (an SGPR): derived from
// threadIdx alone it counts as divergent, and every loop bounded by it (a wave's share of a sweep) is compiled with
// exec masking, its ballots kept in VGPRs and its loads waited for at once.
__device__ inline int wave_index(int tid) { return __builtin_amdgcn_readfirstlane(tid >> 6); }
// The value of lane `l` (wave-uniform) in every lane: one v_readlane instead of a trip through the LDS crossbar
// (__shfl compiles to ds_bpermute_b32 and a wait for it).
__device__ inline int lane_value(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ inline uint64_t lane_value(uint64_t v, int l) {
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)v, l);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(v >> 32), l);
    return (uint64_t)lo | ((uint64_t)hi << 32);
}
// the value of the lane below (lane 0 keeps `first`): a DPP move, wave_shr:1
__device__ inline int lane_below(int v, int first) { return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xF, 0xF, false); }
__device__ inline uint64_t lane_below(uint64_t v, uint64_t first) {
    const uint32_t lo = (uint32_t)lane_below((int)(uint32_t)v, (int)(uint32_t)first);
    const uint32_t hi = (uint32_t)lane_below((int)(uint32_t)(v >> 32), (int)(uint32_t)(first >> 32));
    return (uint64_t)lo | ((uint64_t)hi << 32);
}
// the value of the lane above (lane 63 keeps `last`): wave_shl:1
__device__ inline int lane_above(int v, int last) { return __builtin_amdgcn_update_dpp(last, v, 0x130, 0xF, 0xF, false); }

// inclusive prefix sum over the wave by DPP moves (no LDS crossbar): four row shifts scan each row of 16 lanes, two row
// broadcasts carry the row totals on (rows 1 and 3 take lane 15 of the row below them, then rows 2 and 3 take lane 31)
__device__ inline int wave_incl_add(int v, int lane) {
    (void)lane;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);  // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);  // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);  // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);  // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false); // row_bcast:15 -> rows 1, 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false); // row_bcast:31 -> rows 2, 3
    return v;
}

// inclusive prefix minimum over the wave, the same way (a lane without a source keeps its own value)
__device__ inline int wave_incl_min_dpp(int v) {
#define LZSS_DPP_MIN(ctrl, rows)                                                        \
    {                                                                                  \
        const int t_ = __builtin_amdgcn_update_dpp(v, v, ctrl, rows, 0xF, false);      \
        v = t_ < v ? t_ : v;                                                           \
    }
    LZSS_DPP_MIN(0x111, 0xF) LZSS_DPP_MIN(0x112, 0xF) LZSS_DPP_MIN(0x114, 0xF) LZSS_DPP_MIN(0x118, 0xF)
    LZSS_DPP_MIN(0x142, 0xA) LZSS_DPP_MIN(0x143, 0xC)
#undef LZSS_DPP_MIN
    return v;
}
// minimum / maximum over the wave, in every lane
__device__ inline int wave_min(int v) { return __builtin_amdgcn_readlane(wave_incl_min_dpp(v), 63); }
__device__ inline int wave_max(int v) {
#define LZSS_DPP_MAX(ctrl, rows)                                                        \
    {                                                                                  \
        const int t_ = __builtin_amdgcn_update_dpp(v, v, ctrl, rows, 0xF, false);      \
        v = t_ > v ? t_ : v;                                                           \
    }
    LZSS_DPP_MAX(0x111, 0xF) LZSS_DPP_MAX(0x112, 0xF) LZSS_DPP_MAX(0x114, 0xF) LZSS_DPP_MAX(0x118, 0xF)
    LZSS_DPP_MAX(0x142, 0xA) LZSS_DPP_MAX(0x143, 0xC)
#undef LZSS_DPP_MAX
    return __builtin_amdgcn_readlane(v, 63);
}

// 8 bytes at any byte address (amdhsa runs with unaligned access enabled)
__device__ inline uint64_t ld8(const uint8_t *p) {
    uint64_t v;
    __builtin_memcpy(&v, p, 8);
    return v;
}

// four consecutive int32 / uint16 at any (element-aligned) address in one load
typedef int32_t i32x4_t __attribute__((ext_vector_type(4)));
typedef uint16_t u16x4_t __attribute__((ext_vector_type(4)));
struct I4 { i32x4_t v; };
struct U4 { u16x4_t v; };
__device__ inline I4 ld_i4(const int32_t *p) {
    typedef i32x4_t unaligned_t __attribute__((aligned(4)));
    I4 r;
    r.v = *reinterpret_cast<const unaligned_t *>(p);
    return r;
}
__device__ inline U4 ld_u4(const uint16_t *p) {
    typedef u16x4_t unaligned_t __attribute__((aligned(2)));
    U4 r;
    r.v = *reinterpret_cast<const unaligned_t *>(p);
    return r;
}

// common prefix length of a and b starting from `from`, at most lim; both readable 8 bytes past lim
__device__ inline int lcp_bytes(const uint8_t *a, const uint8_t *b, int from, int lim) {
    int o = from;
    while (o < lim) {
        uint64_t x = ld8(a + o) ^ ld8(b + o);
        if (x) {
            o += (int)(__builtin_ctzll(x) >> 3);
            break;
        }
        o += 8;
    }
    return o < lim ? o : lim;
}

// the same, 16 bytes per round trip (for compares that are expected to run a few words); both readable 16 bytes past lim
__device__ inline int lcp_bytes16(const uint8_t *a, const uint8_t *b, int from, int lim) {
    int o = from;
    while (o < lim) {
        const uint64_t x0 = ld8(a + o) ^ ld8(b + o), x1 = ld8(a + o + 8) ^ ld8(b + o + 8);
        if (x0) {
            o += (int)(__builtin_ctzll(x0) >> 3);
            break;
        }
        if (x1) {
            o += 8 + (int)(__builtin_ctzll(x1) >> 3);
            break;
        }
        o += 16;
    }
    return o < lim ? o : lim;
}

// the same with ONE 16-byte load per side and step (one pass through the texture addresser instead of two)
typedef uint64_t u64x2_t __attribute__((ext_vector_type(2)));
__device__ inline u64x2_t ld16(const uint8_t *p) {
    typedef u64x2_t unaligned_t __attribute__((aligned(1)));
    return *reinterpret_cast<const unaligned_t *>(p);
}
__device__ inline int lcp_bytes_x16(const uint8_t *a, const uint8_t *b, int from, int lim) {
    int o = from;
    while (o < lim) {
        const u64x2_t x = ld16(a + o) ^ ld16(b + o);
        if (x[0]) {
            o += (int)(__builtin_ctzll(x[0]) >> 3);
            break;
        }
        if (x[1]) {
            o += 8 + (int)(__builtin_ctzll(x[1]) >> 3);
            break;
        }
        o += 16;
    }
    return o < lim ? o : lim;
}

// common prefix length of the suffixes at a and b (a != b), at most lim, with the run lengths RL of the text (length
// of the run of equal bytes starting at every position, capped at 255): different run lengths settle the answer
// without a text compare, equal ones let it start behind the run (long runs are what makes prefixes long here)
__device__ inline int lcp_run(const uint8_t *T, const uint8_t *RL, uint32_t a, uint32_t b, int lim) {
    if (lim <= 0 || T[a] != T[b]) return 0;
    const int ra = RL[a], rb = RL[b];
    const int r = ra < rb ? ra : rb;
    if (ra != rb) return r < lim ? r : lim;
    return lcp_bytes(T + a, T + b, r < lim ? r : lim, lim);
}

// lcp_run for the LCP array (neighbours in rank: the first bytes are nearly always equal): the first 8 text bytes
// and the run lengths of both suffixes are fetched in ONE round trip; they settle the first-byte test, unequal runs
// and every prefix shorter than 8, so only longer prefixes pay for a compare loop; and rank k's upper suffix is rank
// k - 1's lower one, so a lane fetches its own suffix only and takes the other from its neighbour lane.  (32 bytes per compare step was
// measured slower: this kernel is bound by the number of scattered loads, not by the length of their chain.)
// wa / wb: the first 8 text bytes at a / b, ra / rb: the run lengths there (the caller has them: a lane's a is its
// neighbour's b)
__device__ inline int lcp_run_rank(const uint8_t *T, uint32_t a, uint32_t b, uint64_t wa, uint64_t wb, int ra, int rb, int lim) {
    if (lim <= 0) return 0;
    const uint64_t x = wa ^ wb;
    if (x & 0xFFull) return 0;
    const int r = ra < rb ? ra : rb;
    if (ra != rb) return r < lim ? r : lim;
    if (x) {
        const int l = (int)(__builtin_ctzll(x) >> 3);
        return l < lim ? l : lim;
    }
    const int from = r > 8 ? r : 8;
#ifndef LCP_NARROW
    return lcp_bytes_x16(T + a, T + b, from < lim ? from : lim, lim);
#else
    return lcp_bytes(T + a, T + b, from < lim ? from : lim, lim);
#endif
}

typedef uint16_t u16x8_t __attribute__((ext_vector_type(8)));

// largest index <= j inside the 8-aligned group of j whose value is < L, or -1 (one 16-byte load)
__device__ inline int scan_down8(const uint16_t *A, int j, int L) {
    const int g = j & ~7;
    typedef u16x8_t unaligned_t __attribute__((aligned(2)));
    const u16x8_t q = *reinterpret_cast<const unaligned_t *>(A + g);
    int found = -1;
#pragma unroll
    for (int t = 0; t < 8; ++t)
        if (t <= j - g && (int)q[t] < L) found = g + t;
    return found;
}

// largest j <= k with LCP[j] < L (L >= 1; LCP[0] == 0 guarantees one exists): the first rank of
// the interval of suffixes sharing >= L bytes with the suffix of rank k.  Two levels of block minima
// (64 and 4096 ranks) and 8 entries per load keep this at a handful of dependent loads.
__device__ inline int prev_lower(const uint16_t *LCP, const uint16_t *bmin1, const uint16_t *bmin2, int k, int L) {
    int j = k;
    for (;;) {  // rest of k's 64-block
        const int f = scan_down8(LCP, j, L);
        if (f >= 0) return f;
        j = (j & ~7) - 1;
        if (((j + 1) & 63) == 0) break;
    }
    if (j < 0) return 0;  // cannot happen (LCP[0] == 0 < L); keeps the loops below bounded
    int b = j >> 6;       // last block not yet examined
    for (;;) {
        const int f = scan_down8(bmin1, b, L);
        if (f >= 0) { b = f; break; }
        b = (b & ~7) - 1;
        if (b < 0) { b = 0; break; }
        if (((b + 1) & 63) == 0) {  // a whole superblock boundary: skip superblocks without a hit
            int c = b >> 6;
            while (c > 0 && (int)bmin2[c] >= L) --c;
            b = c * 64 + 63;
        }
    }
    j = b * 64 + 63;
    for (;;) {
        const int f = scan_down8(LCP, j, L);
        if (f >= 0) return f;
        j = (j & ~7) - 1;
        if (j < 0) return 0;
    }
}

}  // namespace lzss
