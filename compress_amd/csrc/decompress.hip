// decompress.hip - batched decompressor: one 64-lane workgroup (one wavefront) per blob (gfx950).
//
// Replaces Decompress (lzss/decompress.go:17-84) and backref.readFrom (lzss/backref.go:70-86) for a
// batch of independent streams that share one dictionary.
//
// The phrase stream is bit-unaligned and strictly sequential (decompress.go:44-81), but two things
// are wide inside one blob:
//   * literals: from the current bit position every lane k extracts the byte at bit b + 8k; a ballot
//     finds the first reserved symbol (0xFE / 0xFF) or the end of the stream, and all literals before
//     it (a prefix count of the ballot mask) are stored at once at consecutive output positions;
//   * back-references: the copy of up to 256 bytes is done by all lanes; an overlapping copy
//     (offset < length: the run-length encodings of compress.go:175-218) is resolved in order by
//     reading source byte (k mod offset), which only touches bytes written before this phrase.
// Phrases are handled in stream order by the whole wave, so every copy sees its sources complete; a
// workgroup-scope fence is issued only when a copy can reach bytes stored since the last fence.
#include "kernels.h"

namespace lzss {

constexpr int32_t ST_FORMAT = -6;    // MI355_LZSS_E_FORMAT
constexpr int32_t ST_CAPACITY = -5;  // MI355_LZSS_E_CAPACITY

__device__ inline uint64_t load_be64(const uint8_t *p) { return __builtin_bswap64(ld8(p)); }

__global__ __launch_bounds__(64) void decompress_kernel(const uint8_t *in, const uint64_t *in_off, uint8_t *out,
                                                        const uint64_t *out_off, const uint8_t *dict, uint32_t dict_len,
                                                        uint32_t *out_len, int32_t *status) {
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

    const uint64_t total = n * 8;        // bits in the stream (the input buffer is readable 16 bytes past n)
    uint64_t b = LZSS_HEADER_SIZE * 8;   // current bit position (wave-uniform)
    uint64_t opos = 0;                   // bytes produced so far (wave-uniform)
    uint64_t unfenced = 0;               // bytes stored since the last fence
    int32_t st = 0;

    for (;;) {
        // ---- literals: the byte at bit b + 8k for every lane
        const uint64_t bk = b + 8ull * (uint64_t)lane;
        const bool valid = bk + 8 <= total;
        uint32_t sym = 0;
        if (valid) {
            const uint8_t *p = src + (bk >> 3);
            const uint32_t two = ((uint32_t)p[0] << 8) | p[1];
            sym = (two >> (8 - (uint32_t)(bk & 7))) & 0xFFu;
        }
        const bool stop = !valid || sym >= LZSS_SYMBOL_SHORT;
        const uint64_t mask = __ballot(stop);
        const int f = mask ? (int)__builtin_ctzll(mask) : 64;  // number of literals in front of the stop
        if (opos + (uint64_t)f > cap) { st = ST_CAPACITY; break; }
        if (lane < f) dst[opos + (uint64_t)lane] = (uint8_t)sym;
        opos += (uint64_t)f;
        unfenced += (uint64_t)f;
        b += 8ull * (uint64_t)f;
        if (f == 64) continue;
        const uint32_t symf = (uint32_t)__shfl((int)sym, f);
        const bool validf = (__ballot(valid) >> f) & 1ull;
        if (!validf) break;  // fewer than 8 bits left: the reader's EOF ends the loop (decompress.go:45)

        // ---- back-reference at bit b (backref.go:70-86): delimiter(8) | length-1 (8) | offset-1 (14 | 21)
        const bool is_short = symf == LZSS_SYMBOL_SHORT;
        const int nb = is_short ? LZSS_SHORT_BITS : LZSS_DYN_BITS;
        if (b + (uint64_t)nb > total) { st = ST_FORMAT; break; }  // truncated reference: readFrom returns the EOF
        const uint64_t v = load_be64(src + (b >> 3)) << (b & 7);
        const uint64_t field = v >> (64 - nb);
        const int abits = is_short ? LZSS_SHORT_ADDR_BITS : LZSS_DYN_ADDR_BITS;
        const uint32_t len = (uint32_t)((field >> abits) & 0xFFu) + 1;
        const uint64_t off = (field & ((1ull << abits) - 1)) + 1;
        if (opos + len > cap) { st = ST_CAPACITY; break; }
        if (off > opos) {
            if (is_short) { st = ST_FORMAT; break; }  // decompress.go:53-55
            // the reference starts in the dictionary and must end there (decompress.go:65-70)
            const int64_t dict_start = (int64_t)dict_len - (int64_t)(off - opos);
            if (dict_start < 0 || dict_start + (int64_t)len > (int64_t)dict_len) { st = ST_FORMAT; break; }
            for (uint32_t k = lane; k < len; k += 64) dst[opos + k] = dict[dict_start + k];
        } else {
            const uint64_t span = off < len ? off : len;  // the copy reads [opos - off, opos - off + span)
            if (unfenced > off - span) {  // ... which reaches bytes this wave stored after its last fence
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                unfenced = 0;
            }
            const uint8_t *s0 = dst + (opos - off);
            const uint32_t o32 = (uint32_t)off;
            for (uint32_t k = lane; k < len; k += 64) dst[opos + k] = s0[off >= len ? k : k % o32];
        }
        opos += len;
        unfenced += len;
        b += (uint64_t)nb;
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
