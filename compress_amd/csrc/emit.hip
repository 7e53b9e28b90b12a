// emit.hip - phrase selection walk and MSB-first bit emission (gfx950).
//
// Replaces the main loop of compressor.write (lzss/compress.go:176-257) and backref.writeTo
// (lzss/backref.go:63-68) / TryWriteByte.
//
// The loop `for i := start; i < n; i += advance(i)` is a chain, but advance(i) and the phrase chosen
// at i are pure functions of the position (the RLE test compress.go:180-184, bestBackref at i, i+1,
// i+2).  So:
//   decide_kernel   every position in parallel: phrase kind + advance (<= 256); then, per 256-position tile and
//                   by pointer doubling, "enter at x -> land e positions into the next tile, having written b
//                   bits" for every x (because advance <= 256 the chain never skips a tile);
//   walk_kernel     one workgroup per blob: one lane hops tile to tile through those tables, a scan of the
//                   visited tiles' bits gives each tile its bit offset;
//   emit_kernel     one wave per tile: binary lifting over the tile's chain gives every lane its phrases, a scan
//                   their bit offsets; the bits meet in an LDS image of the tile and leave as coalesced bytes
//                   (the byte shared with a neighbouring tile by a word atomic OR into the zeroed output).
#include <algorithm>

#include "kernels.h"
#include "window_search.h"

namespace lzss {

constexpr int DECIDE_THREADS = 256;
constexpr int WALK_THREADS = 256;
constexpr int TILE = 256;  // must be >= the largest advance (1 << maxBackrefLenLog2)

__device__ inline int savings_of(uint32_t r) {  // backref.savings, lzss/backref.go:88-93
    const uint32_t t = res_type(r);
    if (t == RES_NONE) return INT_MIN / 2;
    return 8 * (int)res_len(r) - (t == RES_SHORT ? LZSS_SHORT_BITS : LZSS_DYN_BITS);
}

// One WAVE per 256-position tile, four positions per lane (four tiles per workgroup; the waves never meet, so there
// is no workgroup barrier).  The tile's rows come in and leave as whole words - text, run lengths and the four output
// rows one instruction each, the results as one 16-byte load per lane - and the neighbours a position looks at
// (i - 1, i + 1, i + 2) are the next bytes / words of the same lane or of the lane beside it.  After the phrase at every
// position is known, the tile also resolves, for every position x, where the chain x -> x + advance(x) first lands
// past the tile and how many phrase bits it writes on the way (pointer doubling in LDS: <= 8 rounds, fewer when every
// chain has left the tile) - what walk_kernel needs to hop from tile to tile without ever walking inside one.
__device__ inline void decide_one(int i, int n, int tbase, int tend, bool valid, uint32_t c, uint32_t prev, uint32_t next, int count,
                                  uint32_t r0, uint32_t r1, uint32_t r2, const uint8_t *T, uint32_t &adv, uint32_t &kind_o, uint32_t &p,
                                  uint32_t &bits) {
    int advance = 1;
    uint8_t kind = K_LIT;
    if (valid) {
        const bool reserved = c >= LZSS_SYMBOL_SHORT;
        // run of equal bytes starting at i, capped at 256 and at the end of the buffer (compress.go:180-183)
        const int lim = (n - i) < LZSS_MAX_LEN ? (n - i) : LZSS_MAX_LEN;
        if (count == 255 && lim > 255 && T[i + 255] == c) count = 256;  // (the suffix sort's run lengths are capped at 255)
        if (count > lim) count = lim;
        if (count >= LZSS_MIN_REPEAT) {  // compress.go:184-218
            advance = count;
            if (i > 0 && prev == c) kind = K_RLE_CONT;
            else kind = reserved ? K_RLE_RES : K_RLE_LIT;
        } else {
            const int s0 = savings_of(r0);
            const uint8_t refkind = res_type(r0) == RES_SHORT ? K_SHORT : K_DYN;
            if (reserved) {  // compress.go:221-226 (a dictionary reference of length 1 always exists)
                kind = refkind;
                advance = (int)res_len(r0);
                if (res_type(r0) == RES_NONE) { kind = K_LIT; advance = 1; }  // unreachable; keeps the chain finite
            } else if (s0 < 0) {  // compress.go:227-232
                kind = K_LIT;
                advance = 1;
            } else if (i + 1 < n && savings_of(r1) > s0 + 1) {  // compress.go:235-242
                kind = K_LIT;
                advance = 1;
            } else if (i + 2 < n && next < LZSS_SYMBOL_SHORT && savings_of(r2) > s0 + 2) {  // :243-253
                kind = K_LIT2;
                advance = 2;
            } else {  // compress.go:255-256
                kind = refkind;
                advance = (int)res_len(r0);
            }
        }
    }
    // chain state: < TILE = position inside the tile; TILE + e = left the tile, landing e positions into the next one
    // (e = 0 also stands for "past the end of the text": the hop loop of walk_kernel stops there on its own)
    p = TILE;
    bits = 0;
    if (valid) {
        const int nx = i + advance;
        p = nx >= tend ? (uint32_t)(TILE + nx - tend) : (nx >= n ? (uint32_t)TILE : (uint32_t)(nx - tbase));
        bits = kind_bits(kind);
    }
    adv = (uint32_t)(advance - 1);
    kind_o = kind;
}

__global__ __launch_bounds__(DECIDE_THREADS) void decide_kernel(const Seg *segs, const uint8_t *text, const uint8_t *rlen,
                                                                const uint32_t *res, uint8_t *adv_out, uint8_t *kind_out,
                                                                uint8_t *exit_out, uint16_t *bits_out) {
    __shared__ uint32_t s_pb_all[DECIDE_THREADS / 64][2][TILE];
    const Seg sg = segs[blockIdx.y];
    const int n = (int)sg.n;
    const int lane = threadIdx.x & 63, w = wave_index((int)threadIdx.x);
    const int tbase = (int)((blockIdx.x * (DECIDE_THREADS / 64) + w) * TILE), tend = tbase + TILE;
    if (tbase >= n || tend <= (int)sg.start) return;
    uint32_t (*s_pb)[TILE] = s_pb_all[w];
    const uint8_t *T = text + sg.off;
    const uint32_t *R = res + sg.off;
    const size_t gbase = (size_t)sg.off + (size_t)tbase;  // a multiple of 256: the rows are word aligned, and the arrays
                                                          // are padded to whole tiles, so the whole rows are readable
    const uint32_t tw = reinterpret_cast<const uint32_t *>(text + gbase)[lane];
    const uint32_t rw = reinterpret_cast<const uint32_t *>(rlen + gbase)[lane];
    const uint4 rr = reinterpret_cast<const uint4 *>(res + gbase)[lane];
    // what the lane's last positions see in the next lane / the next tile, and its first one in the previous
    uint32_t tn = (uint32_t)lane_above((int)tw, 0), tp = (uint32_t)lane_below((int)tw, 0) >> 24;
    uint32_t rn0 = (uint32_t)lane_above((int)rr.x, 0), rn1 = (uint32_t)lane_above((int)rr.y, 0);
    if (lane == 63) {
        tn = tend < n ? (uint32_t)T[tend] : 0u;
        rn0 = tend < n ? R[tend] : 0u;
        rn1 = tend + 1 < n ? R[tend + 1] : 0u;
    }
    if (lane == 0) tp = tbase > 0 ? (uint32_t)T[tbase - 1] : 0u;
    const uint32_t r[6] = {rr.x, rr.y, rr.z, rr.w, rn0, rn1};
    const uint32_t tb[6] = {tp, tw & 0xFFu, (tw >> 8) & 0xFFu, (tw >> 16) & 0xFFu, tw >> 24, tn & 0xFFu};
    uint32_t adv[4], kind[4], p[4], bits[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int x = 4 * lane + j, i = tbase + x;
        const bool valid = i < n && i >= (int)sg.start;
        decide_one(i, n, tbase, tend, valid, tb[j + 1], tb[j], tb[j + 2], (int)((rw >> (8 * j)) & 0xFFu), r[j], r[j + 1], r[j + 2], T,
                   adv[j], kind[j], p[j], bits[j]);
    }
    // one word per position: chain state in the low half, bits so far in the high half (<= 9539 bits per tile)
    int cur = 0;
    {
        uint4 v;
        v.x = p[0] | (bits[0] << 16); v.y = p[1] | (bits[1] << 16); v.z = p[2] | (bits[2] << 16); v.w = p[3] | (bits[3] << 16);
        reinterpret_cast<uint4 *>(s_pb[0])[lane] = v;
    }
    __builtin_amdgcn_wave_barrier();
    for (int k = 0; k < 8; ++k) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (p[j] < (uint32_t)TILE) {
                const uint32_t v = s_pb[cur][p[j]];
                bits[j] += v >> 16;
                p[j] = v & 0xFFFFu;
            }
        }
        uint4 v;
        v.x = p[0] | (bits[0] << 16); v.y = p[1] | (bits[1] << 16); v.z = p[2] | (bits[2] << 16); v.w = p[3] | (bits[3] << 16);
        reinterpret_cast<uint4 *>(s_pb[cur ^ 1])[lane] = v;
        cur ^= 1;
        __builtin_amdgcn_wave_barrier();
        const bool more = p[0] < (uint32_t)TILE || p[1] < (uint32_t)TILE || p[2] < (uint32_t)TILE || p[3] < (uint32_t)TILE;
        if (__ballot(more) == 0) break;
    }
    // rows out (positions outside [start, n) of a tile that meets it get the values of an unvisited position; nobody
    // reads them: chains only move forward from `start`, and the emitter stops at n)
    reinterpret_cast<uint32_t *>(adv_out + gbase)[lane] = adv[0] | (adv[1] << 8) | (adv[2] << 16) | (adv[3] << 24);
    reinterpret_cast<uint32_t *>(kind_out + gbase)[lane] = kind[0] | (kind[1] << 8) | (kind[2] << 16) | (kind[3] << 24);
    reinterpret_cast<uint32_t *>(exit_out + gbase)[lane] =
        ((p[0] - TILE) & 0xFFu) | (((p[1] - TILE) & 0xFFu) << 8) | (((p[2] - TILE) & 0xFFu) << 16) | (((p[3] - TILE) & 0xFFu) << 24);
    uint2 bo;
    bo.x = (bits[0] & 0xFFFFu) | (bits[1] << 16);
    bo.y = (bits[2] & 0xFFFFu) | (bits[3] << 16);
    reinterpret_cast<uint2 *>(bits_out + gbase)[lane] = bo;
}

void launch_decide(const Seg *d_segs, uint32_t n_segs, uint32_t max_n, const uint8_t *d_text, const uint8_t *d_rlen,
                   const uint32_t *d_res, const WalkWork &ww, hipStream_t stream) {
    if (n_segs == 0 || max_n == 0) return;
    const uint32_t tiles = (max_n + TILE - 1) / TILE, per_wg = DECIDE_THREADS / 64;
    dim3 g((tiles + per_wg - 1) / per_wg, n_segs);
    hipLaunchKernelGGL(decide_kernel, g, dim3(DECIDE_THREADS), 0, stream, d_segs, d_text, d_rlen, d_res, ww.adv, ww.kind,
                       ww.exitoff, ww.tbits);
}

// ---- walk: visited tiles, per-tile bit offsets, per-blob size -------------------------------------
// One workgroup per blob.  decide_kernel left, for every position, where its chain leaves its tile (exitoff) and the
// phrase bits up to there (tbits).  Groups of WG_TILES tiles stage their exit rows in LDS with coalesced 16-byte
// loads; one lane hops tile to tile through them (the only serial part, one LDS read per tile), every visited tile
// then fetches the bits of its entry and a scan gives each tile its bit offset.  Tile tl's byte x lives at dword
// ((x / 4 + tl) mod 64) of its 256-byte row, so consecutive hops do not meet in one bank.
constexpr int WG_TILES = 128;

__device__ inline uint32_t tile_idx(int tl, int x) { return (uint32_t)(tl * 256 + ((((x >> 2) + tl) & 63) << 2) + (x & 3)); }

__global__ __launch_bounds__(WALK_THREADS) void walk_kernel(const Seg *segs, WalkWork ww, uint32_t header_bytes,
                                                            uint32_t bit_off) {
    __shared__ uint32_t s_exit32[WG_TILES * 64];
    __shared__ uint16_t s_entry[WG_TILES];
    __shared__ uint64_t s_wave[WALK_THREADS / 64];
    __shared__ uint64_t s_carry;
    __shared__ int s_e;
    const uint8_t *s_exit = (const uint8_t *)s_exit32;
    const Seg sg = segs[blockIdx.x];
    const int n = (int)sg.n, start = (int)sg.start;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const uint8_t *exitoff = ww.exitoff + sg.off;
    const uint16_t *tbits = ww.tbits + sg.off;
    uint16_t *entry = ww.entry + sg.off / TILE;
    uint64_t *tilebit = ww.tilebit + sg.off / TILE;
    const int ntiles = (n + TILE - 1) / TILE;
    const int t0 = start / TILE;
    if (tid == 0) {
        s_carry = 0;
        s_e = start;
    }
    for (int g0 = t0; g0 < ntiles; g0 += WG_TILES) {
        const int gt = (ntiles - g0) < WG_TILES ? (ntiles - g0) : WG_TILES;
        __syncthreads();
        // stage the exit rows of the group's tiles (the array is padded, reading past n is harmless)
        for (int c = tid; c < gt * 16; c += WALK_THREADS) {
            const int tl = c >> 4, x0 = (c & 15) * 16;
            const uint4 ve = *reinterpret_cast<const uint4 *>(exitoff + (size_t)(g0 + tl) * TILE + x0);
            const uint32_t e4[4] = {ve.x, ve.y, ve.z, ve.w};
#pragma unroll
            for (int d = 0; d < 4; ++d) s_exit32[(uint32_t)(tl * 64 + (((x0 >> 2) + d + tl) & 63))] = e4[d];
        }
        __syncthreads();
        // hop from tile to tile
        if (tid == 0) {
            int e = s_e;
            for (int tl = 0; tl < gt; ++tl) {
                const int t = g0 + tl;
                if (e >= n || (e >> 8) != t) {
                    s_entry[tl] = 0xFFFFu;
                } else {
                    s_entry[tl] = (uint16_t)(e & 255);
                    e = (t + 1) * TILE + (int)s_exit[tile_idx(tl, e & 255)];
                }
            }
            s_e = e;
        }
        __syncthreads();
        // bits per visited tile, exclusive scan in tile order
        uint64_t bits = 0;
        uint32_t ent = 0xFFFFu;
        if (tid < gt) {
            ent = s_entry[tid];
            if (ent != 0xFFFFu) bits = tbits[(size_t)(g0 + tid) * TILE + ent];
        }
        uint64_t inc = bits;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint64_t v = __shfl_up(inc, o);
            if (lane >= o) inc += v;
        }
        if (lane == 63) s_wave[w] = inc;
        __syncthreads();
        uint64_t base = s_carry;
        for (int ww2 = 0; ww2 < w; ++ww2) base += s_wave[ww2];
        if (tid < gt) {
            tilebit[g0 + tid] = (uint64_t)bit_off + base + inc - bits;
            entry[g0 + tid] = (uint16_t)ent;
        }
        __syncthreads();
        if (tid == WALK_THREADS - 1) s_carry = base + inc;
    }
    __syncthreads();
    if (tid == 0) {
        const uint64_t total = s_carry;
        ww.nbits[blockIdx.x] = total;
        ww.out_len[blockIdx.x] = header_bytes + (uint32_t)(((uint64_t)bit_off + total + 7) / 8);
    }
}

void launch_walk(const Seg *d_segs, uint32_t n_segs, const WalkWork &ww, uint32_t header_bytes, uint32_t bit_off,
                 hipStream_t stream) {
    if (n_segs == 0) return;
    hipLaunchKernelGGL(walk_kernel, dim3(n_segs), dim3(WALK_THREADS), 0, stream, d_segs, ww, header_bytes, bit_off);
}

// ---- output offsets: exclusive scan of the 16-byte aligned blob sizes -----------------------------
__global__ __launch_bounds__(1024) void out_offsets_kernel(uint32_t n_segs, WalkWork ww) {
    __shared__ uint64_t s_wave[16];
    __shared__ uint64_t s_carry;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (uint32_t b0 = 0; b0 < n_segs; b0 += 1024) {
        const uint32_t b = b0 + tid;
        const uint64_t v = b < n_segs ? (((uint64_t)ww.out_len[b] + 15) & ~15ull) : 0;
        uint64_t inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint64_t x = __shfl_up(inc, o);
            if (lane >= o) inc += x;
        }
        if (lane == 63) s_wave[w] = inc;
        __syncthreads();
        uint64_t base = s_carry;
        for (int k = 0; k < w; ++k) base += s_wave[k];
        if (b < n_segs) ww.out_off[b] = base + inc - v;
        __syncthreads();
        if (tid == 1023) s_carry = base + inc;
        __syncthreads();
    }
    if (tid == 0) ww.out_off[n_segs] = s_carry;
}

void launch_out_offsets(uint32_t n_segs, const WalkWork &ww, hipStream_t stream) {
    hipLaunchKernelGGL(out_offsets_kernel, dim3(1), dim3(1024), 0, stream, n_segs, ww);
}

// ---- emit -------------------------------------------------------------------------------------------
// One wave per visited tile.  The chain inside a tile is a pointer chase (x -> x + advance(x)); binary lifting turns
// it into lookups: s_jump[k][x] = the position 2^k phrases after x (TILE = out of the tile), so lane l finds phrases
// l, l + 64, l + 128, l + 192 of the tile directly, wave scans of their bit counts place them, and every phrase ORs its bits into
// the tile's LDS image, which leaves for HBM as coalesced bytes.  The byte a tile shares with a neighbour (its
// stream is not byte aligned) is ORed into the zeroed output with a word atomic; all others are plain stores.
constexpr int EMIT_WORDS = 304;  // 256 one-byte dynamic references (37 bits each) behind a 7-bit lead-in, + 2 words slack

// address bits of a result match_kernel left pending: the smallest-rank occurrence of the match inside the window
// (compress.go:444: windowStart = max(0, i - maxAddress)), searched from the rank stored in the field
__device__ inline uint32_t resolve_field(const SegIndex &S, uint32_t r, int i, bool is_short) {
    const int span = is_short ? LZSS_SHORT_WINDOW : LZSS_DYN_WINDOW;
    const int ws = i > span ? i - span : 0;
    const int p = leftmost(S, (int)res_field(r), (int)res_len(r), ws, i, is_short);
    return (uint32_t)(i - p - 1);
}

// address bits of a pending long reference: into the dictionary (backref.go:66 with the dictionary's smallest-rank
// occurrence) or into the window
__device__ inline uint32_t pending_dyn_field(const SegIndex &S, const DictView &D, uint32_t r, int i) {
    if (res_type(r) == RES_DICT) {
        const int kd = (int)res_field(r), Ld = (int)res_len(r);
        const int pd = D.SA[prev_lower(D.LCP, D.bmin1, D.bmin2, kd, Ld)];
        return (uint32_t)(i + (int)D.n - pd - 1) & 0x1FFFFFu;
    }
    return resolve_field(S, r, i, false) & 0x1FFFFFu;
}

// OR the nb (1..37) low bits of v into the MSB-first bit image w at bit offset o
__device__ inline void image_put(uint32_t *w, uint32_t o, uint64_t v, int nb) {
    const uint64_t x = v << (64 - nb);
    const uint32_t sh = o & 31u, wi = o >> 5;
    const uint64_t y = x >> sh;
    const uint32_t w0 = (uint32_t)(y >> 32), w1 = (uint32_t)y;
    const uint32_t w2 = sh ? (uint32_t)((x << (64 - sh)) >> 32) : 0u;
    if (w0) atomicOr(&w[wi], w0);
    if (w1) atomicOr(&w[wi + 1], w1);
    if (w2) atomicOr(&w[wi + 2], w2);
}

#ifndef EMIT_WAVES_PER_EU
#define EMIT_WAVES_PER_EU 7   // (8 costs 112 spilled SGPRs for nothing: 11.87 vs 11.84 ms)
#endif
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(EMIT_WAVES_PER_EU, EMIT_WAVES_PER_EU))) void emit_kernel(const Seg *segs, uint32_t n_segs, uint32_t gx, const uint8_t *text,
                                                  const uint32_t *res, BatchIndex ix, DictView D, WalkWork ww, uint32_t header_bytes,
                                                  uint32_t bit_off, uint32_t dict_len, int32_t res_idx_fe, int32_t res_idx_ff,
                                                  uint8_t *out_buf, uint64_t out_cap, int32_t *status, int split) {
    // 5.1 KB of LDS per wave, so that eight waves per SIMD fit a CU's 160 KB (with the match results staged as well and a
    // seventh jump level it was 6.6 KB: six waves)
    __shared__ uint32_t s_kind32[64], s_adv32[64], s_text32[65], s_out[EMIT_WORDS];
    __shared__ uint16_t s_jump[6][TILE + 2];
    const uint32_t vb = xcd_block(blockIdx.x, gridDim.x);
    if (vb >= gx * n_segs) return;
    const uint32_t si = vb / gx;
    const int t = (int)(vb % gx);
    const Seg sg = segs[si];
    const int n = (int)sg.n, start = (int)sg.start;
    const int lane = threadIdx.x;
    const uint64_t ooff = ww.out_off[si];
    const uint32_t olen = ww.out_len[si];
    const bool fits = ooff + olen <= out_cap;
    uint8_t *out = out_buf + ooff;
    if (t == 0 && lane == 0) {
        status[si] = fits ? 0 : -5;  // MI355_LZSS_E_CAPACITY
        if (fits && header_bytes == LZSS_HEADER_SIZE) {  // lzss/header.go:22-32, NoCompression = false
            out[0] = (uint8_t)(LZSS_VERSION >> 8);
            out[1] = (uint8_t)(LZSS_VERSION & 0xFF);
            out[2] = 0;
        }
    }
    const int ntiles = (n + TILE - 1) / TILE;
    if (!fits || start >= n || t < start / TILE || t >= ntiles) return;  // an empty Write leaves the zeroed partial byte
    const uint32_t e = ww.entry[sg.off / TILE + t];
    if (e == 0xFFFFu) return;
    out += header_bytes;
    const uint64_t b0 = ww.tilebit[sg.off / TILE + t];
    const int tbase = t * TILE;
    const int hi = (n - tbase) < TILE ? (n - tbase) : TILE;
    const size_t gbase = (size_t)sg.off + (size_t)tbase;

    // stage the tile (the arrays are padded to whole tiles; the byte after the tile belongs to the next tile, if any)
    s_kind32[lane] = reinterpret_cast<const uint32_t *>(ww.kind + gbase)[lane] & 0x7F7F7F7Fu;  // (bit 7: resolve_kernel's mark)
    s_adv32[lane] = reinterpret_cast<const uint32_t *>(ww.adv + gbase)[lane];
    s_text32[lane] = reinterpret_cast<const uint32_t *>(text + gbase)[lane];
    if (lane == 0) s_text32[64] = (t + 1 < ntiles) ? text[gbase + TILE] : 0u;
    const uint32_t *tres = res + gbase;  // the match results of the tile's phrases are read where they lie (a few per lane)
    for (int j = lane; j < EMIT_WORDS; j += 64) s_out[j] = 0;
    if (lane < 6) s_jump[lane][TILE] = TILE;
    __syncthreads();
    const uint8_t *s_kind = (const uint8_t *)s_kind32, *s_adv = (const uint8_t *)s_adv32, *s_text = (const uint8_t *)s_text32;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int x = lane + 64 * q;
        const int nx = x + (int)s_adv[x] + 1;
        s_jump[0][x] = (uint16_t)((x < hi && nx < hi) ? nx : TILE);
    }
    __syncthreads();
    for (int k = 1; k < 6; ++k) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int x = lane + 64 * q;
            s_jump[k][x] = s_jump[k - 1][s_jump[k - 1][x]];
        }
        __syncthreads();
    }
    // my phrases: numbers lane, lane + 64, lane + 128, lane + 192 of the tile (most tiles hold fewer than 64, so the
    // phrases - and the offset searches some of them still need - spread over the lanes)
    int xs[4];
    {
        int x = (int)e;
#pragma unroll
        for (int k = 0; k < 6; ++k)
            if ((lane >> k) & 1) x = s_jump[k][x];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            xs[j] = x;
            x = s_jump[5][s_jump[5][x]];  // 64 phrases on
        }
    }
    const uint32_t lead = (uint32_t)(b0 & 7);
    uint32_t nb[4], off[4], total = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        nb[j] = xs[j] < TILE ? kind_bits(s_kind[xs[j]]) : 0u;
        const uint32_t inc = (uint32_t)wave_incl_add((int)nb[j], lane);
        off[j] = lead + total + inc - nb[j];
        total += (uint32_t)lane_value((int)inc, 63);
    }
    const SegIndex S = seg_index(ix, sg);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int x = xs[j];
        if (x >= TILE) continue;
        const int i = tbase + x;
        const uint8_t k = s_kind[x];
        const uint32_t a = (uint32_t)s_adv[x] + 1u;
        uint64_t v1, v2 = 0;
        int n1, n2 = 0;
        bool later = false;  // the address bits are left to resolve_kernel
        switch (k) {
            case K_LIT: v1 = s_text[x]; n1 = 8; break;
            case K_LIT2: v1 = ((uint64_t)s_text[x] << 8) | s_text[x + 1]; n1 = 16; break;
            case K_SHORT: {
                const uint32_t r = tres[x];
                uint32_t f = res_field(r);
                if (res_pending(r)) {
                    if (split) { f = 0; later = true; }
                    else f = resolve_field(S, r, i, true);
                }
                v1 = ((uint64_t)LZSS_SYMBOL_SHORT << 22) | ((uint64_t)(res_len(r) - 1) << 14) | (f & 0x3FFFu);
                n1 = LZSS_SHORT_BITS;
                break;
            }
            case K_DYN: {
                const uint32_t r = tres[x];
                uint32_t f = res_field(r);
                if (res_type(r) == RES_DICT || res_pending(r)) {
                    if (split) { f = 0; later = true; }
                    else f = pending_dyn_field(S, D, r, i);
                }
                v1 = ((uint64_t)LZSS_SYMBOL_DYNAMIC << 29) | ((uint64_t)(res_len(r) - 1) << 21) | f;
                n1 = LZSS_DYN_BITS;
                break;
            }
            case K_RLE_CONT:  // short ref at address i-1, length count (compress.go:209-212)
                v1 = ((uint64_t)LZSS_SYMBOL_SHORT << 22) | ((uint64_t)(a - 1) << 14);
                n1 = LZSS_SHORT_BITS;
                break;
            case K_RLE_LIT:  // the byte itself, then a short ref of length count-1 (compress.go:201-212)
                v1 = s_text[x]; n1 = 8;
                v2 = ((uint64_t)LZSS_SYMBOL_SHORT << 22) | ((uint64_t)(a - 2) << 14); n2 = LZSS_SHORT_BITS;
                break;
            default: {  // K_RLE_RES: length-1 reference to the reserved symbol in the dictionary (compress.go:192-200)
                const int idx = s_text[x] == LZSS_SYMBOL_DYNAMIC ? res_idx_ff : res_idx_fe;
                v1 = ((uint64_t)LZSS_SYMBOL_DYNAMIC << 29) | (uint64_t)((uint32_t)(i + (int)dict_len - idx - 1) & 0x1FFFFFu);
                n1 = LZSS_DYN_BITS;
                v2 = ((uint64_t)LZSS_SYMBOL_SHORT << 22) | ((uint64_t)(a - 2) << 14); n2 = LZSS_SHORT_BITS;
                break;
            }
        }
        image_put(s_out, off[j], v1, n1);
        if (n2) image_put(s_out, off[j] + (uint32_t)n1, v2, n2);
        if (later) {  // mark the position and leave the bit at which its address field ends (relative to the stream's first byte)
            const uint32_t fe = (uint32_t)((b0 >> 3) << 3) + off[j] + (uint32_t)n1;
            ww.kind[gbase + x] = (uint8_t)(k | 0x80u);
            ww.tbits[gbase + x] = (uint16_t)fe;
            ww.exitoff[gbase + x] = (uint8_t)(fe >> 16);
        }
    }
    __syncthreads();
    const uint32_t nbytes = (lead + total + 7) >> 3, tailbits = (lead + total) & 7u;
    uint8_t *dst = out + (b0 >> 3);
    for (uint32_t j = lane; j < nbytes; j += 64) {
        const uint32_t byte = (s_out[j >> 2] >> (24 - 8 * (j & 3))) & 0xFFu;
        const bool shared = (j == 0 && lead) || (j + 1 == nbytes && tailbits);
        uint8_t *pb = dst + j;
        if (!shared) {
            *pb = (uint8_t)byte;
        } else if (byte) {
            const uintptr_t ad = (uintptr_t)pb;
            atomicOr(reinterpret_cast<uint32_t *>(ad & ~(uintptr_t)3), byte << (8 * (ad & 3)));
        }
    }
}

// ---- the address bits emit_kernel left open ---------------------------------------------------------------------
// In emit_kernel a wave owns one tile, and the handful of its phrases whose smallest-rank occurrence is still pending
// (match_kernel's deferral) kept the whole wave waiting through their searches - chains of dependent loads with six
// busy lanes in 64, half of that kernel's time.  In split mode emit_kernel writes those phrases with a zero address
// field, marks their positions (bit 7 of `kind`) and leaves the bit at which the field ends (24 bits in `tbits` /
// `exitoff`, which nobody reads after walk_kernel); this kernel then scans the marks of 8192 positions per workgroup into
// an LDS queue and every lane takes one pending phrase: dense searches, and the field goes into the stream with byte-wide
// atomic ORs (the emitter stored zeros there).
constexpr int RESOLVE_THREADS = 256;
constexpr int RESOLVE_CHUNK = 8192;
constexpr uint32_t RESOLVE_MAX_N = 400000;  // 37 bits per input byte at worst: the field-end bit must fit 24 bits

__global__ __launch_bounds__(RESOLVE_THREADS) void resolve_kernel(const Seg *segs, uint32_t n_segs, uint32_t gx, const uint32_t *res,
                                                                  BatchIndex ix, DictView D, WalkWork ww, uint32_t header_bytes,
                                                                  uint8_t *out_buf, uint64_t out_cap) {
    __shared__ uint16_t s_q[RESOLVE_CHUNK];
    __shared__ int s_cnt;
    const uint32_t vb = xcd_block(blockIdx.x, gridDim.x);
    if (vb >= gx * n_segs) return;
    const uint32_t si = vb / gx;
    const Seg sg = segs[si];
    const int n = (int)sg.n;
    const int base = (int)(vb % gx) * RESOLVE_CHUNK;
    if (base >= n) return;
    const uint64_t ooff = ww.out_off[si];
    if (ooff + ww.out_len[si] > out_cap) return;  // the emitter wrote nothing for this blob (MI355_LZSS_E_CAPACITY)
    const int tid = threadIdx.x;
    if (tid == 0) s_cnt = 0;
    __syncthreads();
    const size_t gbase = (size_t)sg.off + (size_t)base;  // a multiple of 256; the rows are padded to whole tiles
    const int words = ((n - base < RESOLVE_CHUNK ? n - base : RESOLVE_CHUNK) + 3) / 4;
    for (int w = tid; w < words; w += RESOLVE_THREADS) {
        uint32_t mk = reinterpret_cast<const uint32_t *>(ww.kind + gbase)[w] & 0x80808080u;
        while (mk) {
            const int b = (int)(__builtin_ctz(mk) >> 3);
            mk &= mk - 1;
            const int x = 4 * w + b;
            if (base + x < n && base + x >= (int)sg.start) s_q[atomicAdd(&s_cnt, 1)] = (uint16_t)x;
        }
    }
    __syncthreads();
    const int cnt = s_cnt;
    const SegIndex S = seg_index(ix, sg);
    uint8_t *outp = out_buf + ooff + header_bytes;
    for (int q = tid; q < cnt; q += RESOLVE_THREADS) {
        const int x = (int)s_q[q], i = base + x;
        const uint32_t r = res[sg.off + i];
        const bool is_short = (ww.kind[gbase + x] & 0x7Fu) == K_SHORT;
        const uint32_t fe = (uint32_t)ww.tbits[gbase + x] | ((uint32_t)ww.exitoff[gbase + x] << 16);
        const int nbf = is_short ? LZSS_SHORT_ADDR_BITS : LZSS_DYN_ADDR_BITS;
        const uint32_t f = is_short ? (resolve_field(S, r, i, true) & 0x3FFFu) : pending_dyn_field(S, D, r, i);
        const uint32_t bp = fe - (uint32_t)nbf;                                  // first bit of the field
        const uint64_t y = (uint64_t)f << (64 - nbf - (int)(bp & 7u));           // byte (bp >> 3) is y's top byte
        uint8_t *pb = outp + (bp >> 3);
#pragma unroll
        for (int t = 0; t < 4; ++t) {  // 21 bits behind up to 7 lead bits: at most 4 bytes
            const uint32_t byte = (uint32_t)(y >> (56 - 8 * t)) & 0xFFu;
            if (byte) {
                const uintptr_t ad = (uintptr_t)(pb + t);
                atomicOr(reinterpret_cast<uint32_t *>(ad & ~(uintptr_t)3), byte << (8 * (ad & 3)));
            }
        }
    }
}

void launch_emit(const Seg *d_segs, uint32_t n_segs, uint32_t max_n, const uint8_t *d_text, const uint32_t *d_res, const BatchIndex &ix,
                 const DictView &dict, const WalkWork &ww,
                 uint32_t header_bytes, uint32_t bit_off, uint32_t dict_len, int32_t res_idx_fe, int32_t res_idx_ff,
                 uint8_t *d_out, uint64_t out_cap, int32_t *d_status, hipStream_t stream) {
    if (n_segs == 0) return;
    const uint32_t gx = std::max<uint32_t>(1u, (max_n + TILE - 1) / TILE);  // tiles per segment; tile 0 also reports the status
    const int split = max_n <= RESOLVE_MAX_N ? 1 : 0;
    hipLaunchKernelGGL(emit_kernel, dim3(xcd_grid((uint64_t)gx * n_segs)), dim3(64), 0, stream, d_segs, n_segs, gx, d_text, d_res,
                       ix, dict, ww, header_bytes, bit_off, dict_len, res_idx_fe, res_idx_ff, d_out, out_cap, d_status, split);
    if (split && max_n > 0) {
        const uint32_t gr = (max_n + RESOLVE_CHUNK - 1) / RESOLVE_CHUNK;
        hipLaunchKernelGGL(resolve_kernel, dim3(xcd_grid((uint64_t)gr * n_segs)), dim3(RESOLVE_THREADS), 0, stream, d_segs, n_segs, gr,
                           d_res, ix, dict, ww, header_bytes, d_out, out_cap);
    }
}

}  // namespace lzss
