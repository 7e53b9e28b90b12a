// decompress.hip - batched decompressor: one 64-lane workgroup (one wavefront) per blob (gfx950).
//
// Replaces Decompress (lzss/decompress.go:17-84) and backref.readFrom (lzss/backref.go:70-86) for a
// batch of independent streams that share one dictionary.
//
// The phrase stream is bit-unaligned and strictly sequential (decompress.go:44-81); what is wide inside one blob:
//   * the stream is staged in LDS (2 KiB window, refilled with coalesced 16-byte loads), so the serial chain of a
//     step is LDS latency, not a global round trip;
//   * one step takes a literal run AND the back-reference behind it: every lane k extracts the byte at bit b + 8k, a
//     ballot finds the first reserved symbol (0xFE / 0xFF) or the end of the stream, the literals before it (a
//     prefix of the lanes) are stored at once, and the fields of the reference are the bytes of the next lanes
//     (the same bit alignment), read with v_readlane - no further memory access to decode it;
//   * the copy of up to 256 bytes is done by all lanes, four bytes per lane when source and destination do not
//     overlap; an overlapping copy (offset < length: the run-length encodings of compress.go:175-218) is resolved in
//     order by reading source byte (k mod offset), which only touches bytes written before this phrase.
// Phrases are handled in stream order by the whole wave, so every copy sees its sources complete; a
// workgroup-scope fence is issued only when a copy can reach bytes stored since the last fence.
#include "kernels.h"

namespace lzss {

constexpr int32_t ST_FORMAT = -6;    // MI355_LZSS_E_FORMAT
constexpr int32_t ST_CAPACITY = -5;  // MI355_LZSS_E_CAPACITY

constexpr int DEC_WIN = 2048;        // bytes of the stream staged in LDS
constexpr int DEC_NEED = 64 + 16;    // bytes one step may look at from its first byte on (the fast loop reads three words per lane)

typedef uint32_t dec_u32x4_t __attribute__((ext_vector_type(4)));

__device__ inline uint32_t ld4u(const uint8_t *p) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
}
__device__ inline void st4u(uint8_t *p, uint32_t v) { __builtin_memcpy(p, &v, 4); }

__global__ __launch_bounds__(64) void decompress_kernel(const uint8_t *in, const uint64_t *in_off, uint8_t *out,
                                                        const uint64_t *out_off, const uint8_t *dict, uint32_t dict_len,
                                                        uint32_t *out_len, int32_t *status) {
    __shared__ __attribute__((aligned(16))) uint8_t s_in[DEC_WIN + 16];
    const uint32_t blob = blockIdx.x;
    const int lane = threadIdx.x;
    const uint8_t *src = in + in_off[blob];
    const uint64_t n = in_off[blob + 1] - in_off[blob];
    uint8_t *dst = out + out_off[blob];
    const uint64_t cap = out_off[blob + 1] - out_off[blob];

    // header (lzss/header.go:34-44, decompress.go:22-31)
    if (n < LZSS_HEADER_SIZE || (((uint32_t)src[0] << 8) | src[1]) != LZSS_VERSION || src[2] > 1) {
        if (lane == 0) { status[blob] = ST_FORMAT; out_len[blob] = 0; }
        return;
    }
    if (src[2] == 1) {  // NoCompression: the payload is the data
        const uint64_t m = n - LZSS_HEADER_SIZE;
        if (m > cap) {
            if (lane == 0) { status[blob] = ST_CAPACITY; out_len[blob] = 0; }
            return;
        }
        for (uint64_t k = lane; k < m; k += 64) dst[k] = src[LZSS_HEADER_SIZE + k];
        if (lane == 0) { status[blob] = 0; out_len[blob] = (uint32_t)m; }
        return;
    }

    // The wave-uniform state lives on the scalar unit, which all waves of a CU share: with a whole batch resident that
    // unit, not memory, bounds the kernel, so the state is 32 bits wide (a stream of MaxInputSize bytes has < 2^28
    // bits; capacities are clamped) and the checks of a step are folded.
    if (n >= (1ull << 28)) {
        if (lane == 0) { status[blob] = ST_FORMAT; out_len[blob] = 0; }
        return;
    }
    const uint32_t total = (uint32_t)n * 8u;   // bits in the stream (the input buffer is readable 16 bytes past n)
    const uint32_t cap32 = cap > 0xFFFFFF00ull ? 0xFFFFFF00u : (uint32_t)cap;
    uint32_t b = LZSS_HEADER_SIZE * 8;   // current bit position (wave-uniform)
    uint32_t opos = 0;                   // bytes produced so far (wave-uniform)
    uint32_t fenced = 0;                 // output position of the last fence: bytes from there on may not be visible yet
    uint32_t wend = 0;                   // first stream byte past the LDS window (0: no window yet)
    uint32_t wbase = 0;
    int32_t st = 0;

    // ---- fast loop: while a whole step (64 symbols and a reference behind any of the first 60) lies inside the stream
    //      and its output (64 literals and 256 copied bytes) inside the capacity, nothing has to be checked against the
    //      ends.  With a whole batch resident the scalar unit, which the waves of a CU share, bounds this kernel, so the
    //      step is written for few scalar instructions: every lane takes the 40 bits at its bit position from three
    //      words of the window (funnel shifts) and decodes them as if a reference started there (vector work, the same
    //      for all lanes); after the ballot for the first reserved symbol ONE v_readlane brings the fields of the
    //      reference, and the copy addresses are 32-bit lane offsets from a scalar base.
    const uint32_t *s_w = reinterpret_cast<const uint32_t *>(s_in);
    const uint32_t k4 = 4u * (uint32_t)lane;
    // (scalar compares feeding branches directly, rare cases out of line: the step's control flow is what the scalar
    // unit spends its time on)
    const uint32_t b_lim = total >= 512u ? total - 512u : 0u;   // the fast loop runs while b <= b_lim (and total >= 512)
    const uint32_t o_lim = cap32 >= 320u ? cap32 - 320u : 0u;   // ... and opos <= o_lim (and cap32 >= 320)
    if (total >= 512u && cap32 >= 320u) {
        for (;;) {
            // the window serves the steps that start at bits [.., b_stop]
            if (b > b_lim || opos > o_lim || st != 0) break;
            wbase = (b >> 3) & ~15u;
            wend = wbase + DEC_WIN;
            for (int j = lane; j < DEC_WIN / 16; j += 64) {
                const uint32_t o = wbase + 16u * (uint32_t)j;
                dec_u32x4_t v = {0u, 0u, 0u, 0u};
                if (o <= (uint32_t)n) {
                    typedef dec_u32x4_t unaligned_t __attribute__((aligned(1)));
                    v = *reinterpret_cast<const unaligned_t *>(src + o);
                }
                *reinterpret_cast<dec_u32x4_t *>(s_in + 16 * j) = v;
            }
            __syncthreads();
            const uint32_t wlast = (wend - DEC_NEED) * 8u + 7u;  // last bit position whose step the window serves
            uint32_t b_stop = wlast < b_lim ? wlast : b_lim;
            // one exit, at the top: a step that must leave (capacity, malformed reference) clears b_stop
            while (b <= b_stop) {
                if (__builtin_expect(opos > o_lim, 0)) { b_stop = 0; continue; }
                const uint32_t bi = (b >> 3) - wbase + (uint32_t)lane;
                const uint32_t di = bi >> 2, rr = bi & 3u;
                const uint32_t w0 = s_w[di], w1 = s_w[di + 1], w2 = s_w[di + 2];
                const uint32_t lo = __builtin_amdgcn_alignbyte(w1, w0, rr), hi = __builtin_amdgcn_alignbyte(w2, w1, rr);
                const uint64_t be = (((uint64_t)__builtin_bswap32(lo) << 32) | (uint64_t)__builtin_bswap32(hi)) << (b & 7u);
                const uint32_t T = (uint32_t)(be >> 32), U = (uint32_t)be;  // the 64 bits from bit b + 8 lane on, MSB first
                const uint32_t sv = T >> 24;
                // literals in front of the first reserved symbol (lane 63 counts as one: a step takes 63 literals at most)
                uint32_t f;  // (as asm: the compiler carries the count of trailing zeros as a 64-bit value into every compare)
                asm("s_ff1_i32_b64 %0, %1" : "=s"(f) : "s"(__ballot(sv >= LZSS_SYMBOL_SHORT) | (1ull << 63)));
                if ((uint32_t)lane < f) dst[opos + (uint32_t)lane] = (uint8_t)sv;
                opos += f;
                b += 8u * f;
                if (f > 59u) continue;  // all literals, or the reference's 37 bits are not all inside the lanes' words
                // backref.go:70-86 for the reference that would start at this lane: delimiter(8) | length-1 (8) | offset-1
                // (14 | 21); and whether it is one of the plain ones: source inside the output, in memory, not
                // overlapping its destination.  opos - off wraps for a source in the dictionary, which fails the first test
                const uint32_t lng = sv & 1u;  // 0xFF: the long form (selects are written as bit operations: no lane branches)
                const uint32_t lmask = 0u - lng;
                const uint32_t offm1 = (__builtin_amdgcn_alignbit(T, U, 27) & 0x1FFFFFu & lmask) | ((T >> 2) & 0x3FFFu & ~lmask);
                const uint32_t lenm1 = (T >> 16) & 0xFFu;
                const uint32_t sol = opos - offm1 - 1u;
                const uint32_t special = (uint32_t)(sol + lenm1 >= fenced) | (uint32_t)(offm1 < lenm1);
                const uint32_t pk = offm1 | (lenm1 << 21) | (lng << 29) | (special << 30);
                const uint32_t P = (uint32_t)__builtin_amdgcn_readlane((int)pk, (int)f);
                const uint32_t len = ((P >> 21) & 0xFFu) + 1;
                const uint32_t nb = (P & (1u << 29)) ? LZSS_DYN_BITS : LZSS_SHORT_BITS;
                if (__builtin_expect((P & (1u << 30)) == 0, 1)) {
                    const uint32_t so = opos + ~(P & 0x1FFFFFu);  // opos - off
                    if (len <= 64u) {
                        if ((uint32_t)lane < len) dst[opos + (uint32_t)lane] = dst[so + (uint32_t)lane];
                    } else if (k4 < len) {  // four bytes per lane; the last lane's word ends with the copy (it overlaps its neighbour's)
                        const uint32_t o = k4 + 4 <= len ? k4 : len - 4;
                        st4u(dst + (opos + o), ld4u(dst + (so + o)));
                    }
                } else {
                    const uint32_t off = (P & 0x1FFFFFu) + 1;
                    if (off > opos) {
                        // the reference starts in the dictionary and must end there (decompress.go:65-70); a short one cannot (:53-55)
                        const uint32_t back = off - opos;
                        if (!(P & (1u << 29)) || back > dict_len || len > back) { st = ST_FORMAT; b_stop = 0; continue; }
                        const uint32_t so = dict_len - back;
                        for (uint32_t k = lane; k < len; k += 64) dst[opos + k] = dict[so + k];
                    } else {
                        // the source reaches bytes that may still be on their way - the wait in front of every copy's store
                        // also ends all stores before it, so those are the previous copy's and the literals since,
                        // [fenced, opos) -, or overlaps the destination (offset < length, the run-length encodings of
                        // compress.go:175-218: source byte k mod offset)
                        const uint32_t so = opos - off;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                        if (off >= len) {
                            for (uint32_t k = lane; k < len; k += 64) dst[opos + k] = dst[so + k];
                        } else {
                            for (uint32_t k = lane; k < len; k += 64) dst[opos + k] = dst[so + k % off];
                        }
                    }
                }
                fenced = opos;
                opos += len;
                b += nb;
            }
        }
    }

    // ---- the ends of the stream and of the capacity: the same step with every check
    if (st == 0) for (;;) {
        // ---- the window must hold the bytes this step may look at
        const uint32_t byte0 = b >> 3;
        if (byte0 + DEC_NEED > wend) {
            wbase = byte0 & ~15u;
            wend = wbase + DEC_WIN;
            for (int j = lane; j < DEC_WIN / 16; j += 64) {
                const uint32_t o = wbase + 16u * (uint32_t)j;
                dec_u32x4_t v = {0u, 0u, 0u, 0u};
                if (o <= (uint32_t)n) {
                    typedef dec_u32x4_t unaligned_t __attribute__((aligned(1)));
                    v = *reinterpret_cast<const unaligned_t *>(src + o);
                }
                *reinterpret_cast<dec_u32x4_t *>(s_in + 16 * j) = v;
            }
            __syncthreads();
        }
        // ---- the byte at bit b + 8k for every lane
        const uint32_t bk = b + 8u * (uint32_t)lane;
        const bool valid = bk + 8 <= total;
        const uint32_t bi = (bk >> 3) - wbase;
        const uint32_t sh = 8u - (b & 7u);
        const uint32_t two = ((uint32_t)s_in[bi] << 8) | (uint32_t)s_in[bi + 1];
        // (also for lanes whose eight bits end past the stream: a reference at the very end needs 30 / 37 bits, not
        // whole bytes, so its last field byte may come from such a lane; the bits past the stream are shifted out)
        const uint32_t sym = (two >> sh) & 0xFFu;
        const bool stop = !valid || sym >= LZSS_SYMBOL_SHORT;
        const uint64_t mask = __ballot(stop);
        const uint32_t f = mask ? (uint32_t)__builtin_ctzll(mask) : 64u;  // number of literals in front of the stop
        if (opos + f > cap32) { st = ST_CAPACITY; break; }
        if ((uint32_t)lane < f) dst[opos + (uint32_t)lane] = (uint8_t)sym;
        opos += f;
        b += 8u * f;
        if (f > 59) continue;                 // all literals, or the fields of the reference are not all in this step's lanes
        if (b + 8 > total) break;             // fewer than 8 bits left: the reader's EOF ends the loop (decompress.go:45)

        // ---- back-reference at bit b (backref.go:70-86): delimiter(8) | length-1 (8) | offset-1 (14 | 21); its bytes
        //      are the symbols of lanes f .. f + 4
        const uint32_t s0 = (uint32_t)__builtin_amdgcn_readlane((int)sym, (int)f);
        const uint32_t s1 = (uint32_t)__builtin_amdgcn_readlane((int)sym, (int)f + 1);
        const uint32_t s2 = (uint32_t)__builtin_amdgcn_readlane((int)sym, (int)f + 2);
        const uint32_t s3 = (uint32_t)__builtin_amdgcn_readlane((int)sym, (int)f + 3);
        const uint32_t s4 = (uint32_t)__builtin_amdgcn_readlane((int)sym, (int)f + 4);
        const bool is_short = s0 == LZSS_SYMBOL_SHORT;
        const uint32_t nb = is_short ? LZSS_SHORT_BITS : LZSS_DYN_BITS;
        const uint32_t len = s1 + 1;
        const uint32_t off = (is_short ? ((s2 << 6) | (s3 >> 2)) : ((s2 << 13) | (s3 << 5) | (s4 >> 3))) + 1;
        if (b + nb > total) { st = ST_FORMAT; break; }  // truncated reference: readFrom returns the EOF
        if (opos + len > cap32) { st = ST_CAPACITY; break; }
        uint8_t *d0 = dst + opos;
        if (off > opos) {
            // the reference starts in the dictionary and must end there (decompress.go:65-70); a short one cannot (:53-55)
            const uint32_t back = off - opos;  // bytes before the end of the dictionary
            if (is_short || back > dict_len || len > back) { st = ST_FORMAT; break; }
            const uint8_t *sp = dict + (dict_len - back);
            if (k4 + 4 <= len) st4u(d0 + k4, ld4u(sp + k4));
            else
                for (uint32_t t = k4; t < len; ++t) d0[t] = sp[t];
        } else {
            const uint32_t span = off < len ? off : len;  // the copy reads [opos - off, opos - off + span)
            if (opos - off + span > fenced) {  // ... which reaches bytes this wave stored after its last fence
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                fenced = opos;
            }
            const uint8_t *sp = d0 - off;
            if (off >= len) {  // source and destination apart: four bytes per lane
                if (k4 + 4 <= len) st4u(d0 + k4, ld4u(sp + k4));
                else
                    for (uint32_t t = k4; t < len; ++t) d0[t] = sp[t];
            } else {
                for (uint32_t k = lane; k < len; k += 64) d0[k] = sp[k % off];
            }
        }
        opos += len;
        b += nb;
    }
    if (lane == 0) {
        status[blob] = st;
        out_len[blob] = st ? 0u : (uint32_t)opos;
    }
}

void launch_decompress(const uint8_t *d_in, const uint64_t *d_in_off, uint32_t n_blobs, uint8_t *d_out,
                       const uint64_t *d_out_off, const uint8_t *d_dict, uint32_t dict_len, uint32_t *d_out_len,
                       int32_t *d_status, hipStream_t stream) {
    if (n_blobs == 0) return;
    hipLaunchKernelGGL(decompress_kernel, dim3(n_blobs), dim3(64), 0, stream, d_in, d_in_off, d_out, d_out_off, d_dict,
                       dict_len, d_out_len, d_status);
}

}  // namespace lzss
