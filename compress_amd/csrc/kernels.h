// kernels.h - launchers of the HIP kernels (internal to libmi355_lzss.so).
#pragma once
#include "lzss_device.h"

namespace lzss {

// ---- sa_build.hip -----------------------------------------------------------------------------
struct SaWork {       // batch-wide arrays, indexed [seg.off + k], k in [0, n]
    uint64_t *recA, *recB;               // sort ping-pong (8-byte records, [seg.off + k])
    int32_t *rkA, *rkB;                  // large texts only (sa_large.hip): second suffix buffer, group starts
    int32_t *sa, *isa;                   // outputs (isa: only the ranks of the first HEAD positions are guaranteed, see launch_isa_from_sa)
    uint8_t *rlen;                       // output: length of the run of equal bytes starting at every position, capped at 255
    uint32_t *pairs;                     // scratch of the main kernel, 4 bytes per suffix ([seg.off + k]; the match results'
                                         // buffer, which nothing uses before the segment is sorted)
    uint32_t *mpos;                      // scratch of the run-compressed sort (sa_meta_kernel): position word of every item,
                                         // 4 bytes per suffix ([seg.off + i]); null: every suffix is sorted (sa_build_kernel)
    uint8_t *tmp;                        // scratch of the large-text sort (sa_large.hip), tmp_bytes long; null without large texts
    size_t tmp_bytes;
};
// texts of more than this many suffixes are sorted one at a time by the device-wide sort of sa_large.hip
constexpr uint32_t SA_BITMAP_MAX_N = 1u << 19;
// h_segs: host copy of the segments; needed (and read) only when max_n + 1 > SA_BITMAP_MAX_N - the large texts are then
// sorted one after the other by launch_sa_large, which synchronises the stream.  Returns 0 or a hipError_t.
int launch_sa_build(const Seg *d_segs, const Seg *h_segs, uint32_t n_segs, uint32_t max_n, const uint8_t *d_text, const SaWork &wk,
                    hipStream_t stream);
// isa[sa[k]] = k for whole segments: launch_sa_build leaves only the ranks of the first HEAD positions in SaWork::isa
// (the run-compressed sort uses the buffer for its items); the index views of the C ABI call this
void launch_isa_from_sa(const Seg *d_segs, uint32_t n_segs, uint32_t max_n, const int32_t *d_sa, int32_t *d_isa, hipStream_t stream);
// sa_large.hip: one text of more than SA_BITMAP_MAX_N suffixes on the whole device (rocPRIM radix sort per doubling round)
size_t sa_large_temp_bytes(uint32_t N);
int launch_sa_large(const Seg &sg, const uint8_t *d_text, const SaWork &wk, hipStream_t stream);

// ---- match.hip --------------------------------------------------------------------------------
struct DictView {     // the dictionary index of a ctx (device pointers)
    const uint8_t *T;
    uint32_t n;
    const int32_t *SA;
    const uint16_t *LCP, *bmin1, *bmin2;
    const uint64_t *key;       // [rank] first 8 bytes of the suffix, big-endian, zero padded
    const int2 *bkt;           // rank range [x, y) of the suffixes of a bucket (DICT_BUCKETS entries, see below)
};
// rank range tables of the dictionary: [0, 65536) by the first two bytes, then 2^17 entries by the top 17 bits of the
// run-aware key (first byte, run code)
constexpr uint32_t DICT_PAIR_BUCKETS = 65536, DICT_BUCKETS = 65536 + (1u << 17);
// fills key / bkt (DICT_BUCKETS pairs, must be zeroed) from the dictionary's text and suffix array
void launch_dict_aux(const uint8_t *d_text, uint32_t n, const int32_t *d_sa, uint64_t *d_key, int32_t *d_bkt, hipStream_t stream);
struct BatchIndex {   // batch-wide index arrays
    const uint8_t *text;
    int32_t *sa, *isa;
    uint16_t *lcp;    // [seg.off + k]
    const uint8_t *rlen;  // [seg.off + i] run of equal bytes starting at position i, capped at 255 (sa_build)
    // [seg.off + k] what match_kernel needs to know about the pattern at rank k (match.hip: pattern_word), written by the
    // LCP kernels.  It lives in the suffix sort's SECOND record buffer, which nothing uses once the segment is sorted.
    uint64_t *pkey;
    uint16_t *bmin1;  // [seg.off/64 + k/64]
    uint16_t *bmin2;  // [seg.off/64 + k/4096]
    // window index (sa_build.hip: window_index_kernel): the ranks 1..n regrouped by the 16 KiB chunk of their
    // position, rank order kept inside a chunk.  It lives in the segment's OWN part of the suffix sort's record
    // buffer (free once the segment is sorted; other segments may still be sorting theirs on another stream):
    // win + seg.off holds seg_stride(n) 8-byte records = 2 * seg_stride(n) int32: first the ranks, then the positions.
    uint64_t *win;
    __host__ __device__ int32_t *ks(const Seg &sg) const { return (int32_t *)(win + sg.off); }                   // rank of entry j
    __host__ __device__ int32_t *ps(const Seg &sg) const { return (int32_t *)(win + sg.off) + seg_stride(sg.n); }  // its position
    // entry j of chunk c sits at index c * 16384 + j of both
    int32_t *pmin1, *pmax1;  // min / max of ps over blocks of 64 entries   [seg.off/64 + j/64]
    int32_t *pmin2, *pmax2;  // ... over blocks of 4096 entries             [seg.off/64 + j/4096]
    int32_t *samin1, *samin2;  // min of sa over blocks of 64 / 4096 ranks  [seg.off/64 + k/64], [.. + k/4096]
    int32_t *dlb;     // [seg.off/64 + b] lower bound in the DICTIONARY's suffix array of the pattern at rank 64 b (match.hip)
    // chunk directory: cdir[(seg.off/64 + b) * nch + c] = entries of chunk c whose rank is < 1 + 64 b, so the first
    // entry of a chunk with rank >= lo is a few entries past cdir[(lo - 1) / 64] instead of a binary search away
    uint16_t *cdir;
    uint32_t nch;     // chunks per segment the directory is laid out for (>= chunks of the largest segment)
    // texts of up to 2^18 suffixes: an entry of the window index is ONE word, rank << 14 | position inside its chunk
    // (in ks; ps is not written), which halves what window_index_kernel writes and what the summaries read
    uint32_t win_packed;
};
// index: the window index proper; summaries: the second level of the suffix array's block minima (the first is the LCP
// kernel's), which the long-window search reads
// h_segs / wk: with them, the texts of more than SA_BITMAP_MAX_N suffixes get their index from the device-wide routine of
// sa_large.hip (launch_window_large; scratch: wk->rkA, rkB, pairs, tmp) instead of the one-workgroup kernel
void launch_window_index(const Seg *d_segs, uint32_t n_segs, uint32_t max_n, const BatchIndex &ix, hipStream_t stream,
                         bool index = true, bool summaries = true, const Seg *h_segs = nullptr, const SaWork *wk = nullptr);
size_t window_large_temp_bytes(uint32_t n);
int launch_window_large(const Seg &sg, const BatchIndex &ix, const SaWork &wk, hipStream_t stream);
void launch_lcp(const Seg *d_segs, uint32_t n_segs, uint32_t max_n, const BatchIndex &ix, hipStream_t stream);
// skip_unqueried: leave out positions the walk provably never asks about (deep inside a run of >= 160
// equal bytes, compress.go:180-218); their result is then RES_NONE
// (launch_match_head also fills ix.dlb, the dictionary brackets launch_match reads: call it first)
void launch_match_head(const Seg *d_segs, uint32_t n_segs, uint32_t max_n, const BatchIndex &ix, const DictView &dict,
                       uint32_t *d_res, hipStream_t stream);
void launch_match(const Seg *d_segs, uint32_t n_segs, uint32_t max_n, const BatchIndex &ix, const DictView &dict,
                  uint32_t *d_res, bool skip_unqueried, hipStream_t stream);

// ---- emit.hip ---------------------------------------------------------------------------------
struct WalkWork {
    uint8_t *adv;       // [seg.off + i]  advance - 1 of the phrase chosen at i
    uint8_t *kind;      // [seg.off + i]  phrase kind
    uint8_t *exitoff;   // [seg.off + i]  where the chain from i first lands past i's 256-tile
    uint16_t *entry;    // [seg.off/256 + t] offset of the first visited position in tile t, 0xFFFF = none
    uint64_t *tilebit;  // [seg.off/256 + t] bit offset of tile t's first phrase in the blob's stream
    uint16_t *tbits;    // [seg.off + i]  phrase bits the chain from i writes before it leaves i's tile
    uint64_t *nbits;    // [seg] phrase bits of the blob
    uint64_t *out_off;  // [seg], n_segs+1 entries: byte offset of the blob's output in the batch output buffer
    uint32_t *out_len;  // [seg] bytes written (header + ceil(bits/8))
};
// d_rlen: run lengths of the suffix sort (SaWork::rlen)
void launch_decide(const Seg *d_segs, uint32_t n_segs, uint32_t max_n, const uint8_t *d_text, const uint8_t *d_rlen,
                   const uint32_t *d_res, const WalkWork &ww, hipStream_t stream);
// header_bytes: 3 (one-shot Compress) or 0; bit_off: first bit of the stream inside its first byte (streaming Write)
void launch_walk(const Seg *d_segs, uint32_t n_segs, const WalkWork &ww, uint32_t header_bytes, uint32_t bit_off,
                 hipStream_t stream);
void launch_out_offsets(uint32_t n_segs, const WalkWork &ww, hipStream_t stream);
void launch_emit(const Seg *d_segs, uint32_t n_segs, uint32_t max_n, const uint8_t *d_text, const uint32_t *d_res, const BatchIndex &ix,
                 const DictView &dict, const WalkWork &ww,
                 uint32_t header_bytes, uint32_t bit_off, uint32_t dict_len, int32_t res_idx_fe, int32_t res_idx_ff,
                 uint8_t *d_out, uint64_t out_cap, int32_t *d_status, hipStream_t stream);

// ---- decompress.hip ---------------------------------------------------------------------------
// in / out: CSR offsets (n_blobs + 1 entries, device); the input buffer must be readable 16 bytes past its end
void launch_decompress(const uint8_t *d_in, const uint64_t *d_in_off, uint32_t n_blobs, uint8_t *d_out,
                       const uint64_t *d_out_off, const uint8_t *d_dict, uint32_t dict_len, uint32_t *d_out_len,
                       int32_t *d_status, hipStream_t stream);

}  // namespace lzss
