/* mi355_lzss.h - C ABI of the MI355X-native lzss match-finding path.
 *
 * This is the drop-in boundary.  The reference (Consensys/compress, pure Go) has no FFI seam;
 * the seam is its Go package API (lzss/compress.go, lzss/decompress.go).  A Go shim keeps that
 * API (go/lzss/) and binds the entry points below through cgo.  Every entry
 * point cites the reference code it replaces (paths relative to the reference repository root).
 *
 * Conventions
 *  - plain pointers and sizes only; the caller owns every host buffer and the library never keeps
 *    a host pointer after the call returns (cgo pointer rules);
 *  - return value: 0 = MI355_LZSS_OK, negative = error; mi355_lzss_last_error() gives the text.
 *    Where the reference has an error text, the same text is used
 *    ("dict size must be <= 4194304", "input size must be <= 4194304");
 *  - a ctx / batch is NOT thread-safe (like lzss.Compressor); use one per thread / goroutine - except
 *    mi355_lzss_compressed_size(_batch), which is (like CompressedSize256k, compress.go:381-383).
 *    Objects created on different devices are independent;
 *  - there is no CPU fallback: without a usable HIP device every entry point fails with
 *    MI355_LZSS_E_DEVICE.
 */
#ifndef MI355_LZSS_H
#define MI355_LZSS_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MI355_LZSS_OK            0
#define MI355_LZSS_E_ARG        -1   /* bad argument */
#define MI355_LZSS_E_DICT_SIZE  -2   /* "dict size must be <= %d"  lzss/compress.go:39-41 */
#define MI355_LZSS_E_INPUT_SIZE -3   /* "input size must be <= %d" lzss/compress.go:473-475 */
#define MI355_LZSS_E_DEVICE     -4   /* HIP runtime error / no device */
#define MI355_LZSS_E_CAPACITY   -5   /* caller's output buffer too small */
#define MI355_LZSS_E_FORMAT     -6   /* malformed compressed stream (per blob: see status[]) */

typedef struct mi355_lzss_ctx mi355_lzss_ctx;     /* one dictionary on one device */
typedef struct mi355_lzss_batch mi355_lzss_batch; /* device workspace for batches of blobs */

/* ---- dictionary / context -------------------------------------------------------------------
 * Replaces the dictionary half of NewCompressor (lzss/compress.go:37-67): AugmentDict
 * (compress.go:70-88), the size check (:39-41), dictReservedIdx (:47-59) and
 * suffixarray.New(dict) (:64) - the dictionary index is built on the device. */
int  mi355_lzss_ctx_create(const uint8_t *dict, size_t dict_len, int device, mi355_lzss_ctx **out);
void mi355_lzss_ctx_destroy(mi355_lzss_ctx *ctx);
/* A context for Decompress only (lzss/decompress.go:34: the decoder augments the dictionary and builds
 * no index): the augmented dictionary's text goes to the device, nothing else is built or allocated
 * until a decompress entry point runs.  Every compress entry point fails with MI355_LZSS_E_ARG on it. */
int  mi355_lzss_ctx_create_decoder(const uint8_t *dict, size_t dict_len, int device, mi355_lzss_ctx **out);
/* last error text of this thread (valid until the next failing call on the thread) */
const char *mi355_lzss_last_error(void);
/* Tuning knob with no counterpart in the reference: batches of this context run their index / match stages as
 * `parts` (1..4) sub-batches on concurrent streams, so that the tail of one part's suffix sort overlaps the next
 * stages of another.  Results do not depend on it; per-stage times (batch_stage_times) are only meaningful with
 * parts == 1.  Default: 2 (batches of fewer than 8 blobs per part, or with a text of 512 KiB or more, always run as
 * one part), or the environment variable MI355_LZSS_SPLIT when the context is created. */
int  mi355_lzss_ctx_set_overlap(mi355_lzss_ctx *ctx, int parts);
/* Tuning knob with no counterpart in the reference: the host-buffer batch entry points
 * (mi355_lzss_compress_batch, ..._multi) cut a batch into chunks of about this many input bytes and run
 * them as a pipeline - chunk k+1 goes up and chunk k-1 comes down while chunk k computes.  Default
 * 200 MiB; a chunk that is not the last takes a whole number of rounds of the suffix sort's resident
 * workgroups (2 per CU: 1536 blobs of 128 KiB on an MI355X).  Results do not depend on it. */
int  mi355_lzss_ctx_set_chunk_bytes(mi355_lzss_ctx *ctx, uint64_t bytes);
/* Pinned host memory (hipHostMalloc) for callers that want their batch buffers to move by DMA: with
 * `in` / `out` of the batch entry points allocated here the copies overlap with the kernels.  Pageable
 * buffers work too (the HIP runtime stages them).  No counterpart in the reference. */
void *mi355_lzss_host_alloc(size_t bytes);
void  mi355_lzss_host_free(void *p);
/* length of the augmented dictionary and the two reserved-symbol indices ([0]=0xFE, [1]=0xFF) */
int  mi355_lzss_dict_info(const mi355_lzss_ctx *ctx, uint64_t *aug_len, int32_t reserved_idx[2]);

/* ---- the hot path, one input buffer ----------------------------------------------------------
 * Per-position results of the bestBackref closure (lzss/compress.go:146-173 = findBackRef
 * :435-470 over suffixarray.LookupLongest, suffixarray.go:59-152) for every position of
 * d[start:n); the index covers all of d[0:n) exactly as in Write (compress.go:113-118).
 * type[k] = 0xFE / 0xFF, or 0 when there is no back-reference; addr = backref.address
 * (dictionary-prefixed for 0xFF), len = backref.length; arrays hold n-start entries. */
int mi355_lzss_best_backrefs(mi355_lzss_ctx *ctx, const uint8_t *d, size_t n, size_t start,
                             int32_t *addr, int32_t *len, uint8_t *type);

/* The phrase stream of one Write: compressor.write (lzss/compress.go:138-260) with the bit
 * emission of backref.writeTo (lzss/backref.go:63-68) for inbuf[start:n), index over inbuf[0:n).
 * Bits are packed MSB-first starting at bit `bit_off` (0..7) of out[0]; the bit_off high bits of
 * out[0] are written as zero (the caller ORs its pending partial byte in), the pad bits after
 * the last phrase are zero.  *nbits = number of phrase bits (without bit_off).
 * Needs out_cap >= ceil((bit_off + *nbits)/8); on MI355_LZSS_E_CAPACITY *nbits is still set. */
int mi355_lzss_write(mi355_lzss_ctx *ctx, const uint8_t *inbuf, size_t n, size_t start,
                     unsigned bit_off, uint8_t *out, size_t out_cap, uint64_t *nbits);

/* The same Write for callers that cannot allocate (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc).  NO COUNTERPART IN THE REFERENCE.  It calls no hipMalloc /
 * hipFree / allocating routine and does not synchronise: the kernels of mi355_lzss_write are enqueued on
 * `stream` (a hipStream_t; NULL = the default stream) with their ~37 bytes per position of workspace in a
 * static device array of the library (5.8 MB held by every GPU the library runs on until the process
 * ends).  d_inbuf, d_out, d_nbits, d_status are DEVICE pointers: inbuf[0:n) in, the packed bits in
 * d_out (zero padded, layout as above), *d_nbits = phrase bits, *d_status = 0 or MI355_LZSS_E_CAPACITY
 * (then d_out holds nothing useful); all valid once the stream has reached the end of the call's work.
 * Limit: n <= MI355_LZSS_NOALLOC_MAX_INPUT, larger inputs are rejected with MI355_LZSS_E_INPUT_SIZE.
 * All calls on one GPU share that GPU's array (whatever the ctx): calls that overlap on one GPU race. */
#define MI355_LZSS_NOALLOC_MAX_INPUT 131072u
int mi355_lzss_write_noalloc(mi355_lzss_ctx *ctx, const uint8_t *d_inbuf, size_t n, size_t start,
                             unsigned bit_off, uint8_t *d_out, size_t out_cap, uint64_t *d_nbits,
                             int32_t *d_status, void *stream);

/* CompressedSize256k (lzss/compress.go:384-407): 3 + ceil(bits/8) of write(d, 0); n <= 256 KiB.
 * Like the reference's ("state less and thread-safe", compress.go:381-383) this entry point - and only
 * this one and its batch form - may be called on one ctx from several threads at once: every call
 * works in a workspace of its own (own stream, own buffers) taken from a pool of the ctx. */
int mi355_lzss_compressed_size(mi355_lzss_ctx *ctx, const uint8_t *d, size_t n, uint64_t *size);
/* the same for n_blobs independent inputs in one launch sequence (CSR offsets, n_blobs+1 entries);
 * sizes[b] = CompressedSize256k(blob b); status[b] = 0 or MI355_LZSS_E_INPUT_SIZE (blob > 256 KiB;
 * with status == NULL such a blob fails the call) */
int mi355_lzss_compressed_size_batch(mi355_lzss_ctx *ctx, const uint8_t *in, const uint64_t *in_off,
                                     uint32_t n_blobs, uint64_t *sizes, int32_t *status);

/* ---- batches of independent blobs ------------------------------------------------------------
 * One-shot Compress (lzss/compress.go:375-379 = Reset + Write + Bytes) of n_blobs independent
 * inputs that share the ctx dictionary.  in_off / out_off are CSR-style byte offsets with
 * n_blobs+1 entries; blob b is in[in_off[b]:in_off[b+1]) and may use out[out_off[b]:out_off[b+1]).
 * out_len[b] = compressed size including the 3-byte header; status[b] = 0 or a negative code. */
int mi355_lzss_compress_batch(mi355_lzss_ctx *ctx, const uint8_t *in, const uint64_t *in_off,
                              uint32_t n_blobs, uint8_t *out, const uint64_t *out_off,
                              uint32_t *out_len, int32_t *status);

/* Decompress (lzss/decompress.go:17-84) of n_blobs independent streams with the ctx dictionary.
 * out_len[b] = decompressed size; status[b] = 0, MI355_LZSS_E_FORMAT or MI355_LZSS_E_CAPACITY.
 * The bytes of out[out_off[b]:out_off[b+1]) beyond out_len[b] are unspecified after the call. */
int mi355_lzss_decompress_batch(mi355_lzss_ctx *ctx, const uint8_t *in, const uint64_t *in_off,
                                uint32_t n_blobs, uint8_t *out, const uint64_t *out_off,
                                uint32_t *out_len, int32_t *status);

/* ---- one call, several GPUs -------------------------------------------------------------------
 * The two batch operations over n_ctx contexts (one per device of the node, all with the same
 * dictionary): the blobs are cut into n_ctx contiguous ranges balanced by input bytes
 * (mi355_lzss_partition), every range runs on its context from a host thread of its own, results land
 * in the caller's CSR buffers; no collective, no peer traffic - blobs are independent (each Compress
 * starts from Reset, compress.go:375-376).  No counterpart in the reference (single-threaded). */
int mi355_lzss_compress_batch_multi(mi355_lzss_ctx *const *ctxs, int n_ctx, const uint8_t *in,
                                    const uint64_t *in_off, uint32_t n_blobs, uint8_t *out,
                                    const uint64_t *out_off, uint32_t *out_len, int32_t *status);
int mi355_lzss_decompress_batch_multi(mi355_lzss_ctx *const *ctxs, int n_ctx, const uint8_t *in,
                                      const uint64_t *in_off, uint32_t n_blobs, uint8_t *out,
                                      const uint64_t *out_off, uint32_t *out_len, int32_t *status);
/* the partition they use (host only, no device needed): first[t] = first blob of part t, n_parts+1
 * entries, first[n_parts] = n_blobs; part t starts at the first blob whose offset reaches t/n_parts of
 * the bytes */
int mi355_lzss_partition(const uint64_t *off, uint32_t n_blobs, int n_parts, uint32_t *first);

/* Filling a size-capped blob from a stream: the Write / Revert loop of the reference's callers
 * (lzss/compress_test.go:337-353: Write chunk after chunk while Len() < out_max, Revert the chunk
 * that overshot).  Every Write re-indexes the whole inBuf (compress.go:113-116), so on a GPU the
 * loop is one short job per chunk; the bits a chunk adds depend only on the blob's bytes before and
 * inside it, so this call runs the Writes of `lookahead` candidate chunk counts as one batch and
 * replays the loop's decisions on their bit counts.  data[0, n) is the rest of the stream.  Out: the
 * blob (header included), byte for byte what the loop leaves in Bytes() - the pad bits a Revert
 * leaves behind (outBuf.Truncate, compress.go:341) included -, and the number of chunks it holds.
 * *needs_loop = 1 (nothing written): ConsiderBypassing would fire in this blob (compress_test.go:
 * 343-346); run the loop itself for it (mi355_lzss_write and the host mirror's state machine).
 * Thread-safe per ctx (pooled workspaces).  compress_amd/stream.py is the same in Python. */
int mi355_lzss_pack_blob(mi355_lzss_ctx *ctx, const uint8_t *data, uint64_t n, uint32_t chunk,
                         uint64_t out_max, uint32_t lookahead, uint8_t *out, uint64_t out_cap,
                         uint64_t *out_len, uint64_t *chunks, int *needs_loop);

/* ---- device-resident batches (what bench.py times) -------------------------------------------
 * The same two operations split into upload / run / download so that a caller can keep inputs
 * and outputs resident in HBM and time the device work alone.  A batch owns all device memory
 * for up to max_blobs blobs and max_total_bytes input bytes (compress) or output bytes
 * (decompress).  No counterpart in the reference (it has no device). */
int  mi355_lzss_batch_create(mi355_lzss_ctx *ctx, uint32_t max_blobs, uint64_t max_total_bytes,
                             mi355_lzss_batch **out);
void mi355_lzss_batch_destroy(mi355_lzss_batch *b);
/* host -> HBM: the blobs to compress (starts[b] = first position to encode, NULL = all zero) */
int  mi355_lzss_batch_upload(mi355_lzss_batch *b, const uint8_t *in, const uint64_t *in_off,
                             uint32_t n_blobs, const uint32_t *starts);
/* device only: index build, match search, phrase walk, bit emission; returns after the stream
 * has drained.  with_header != 0 writes the 3-byte header in front of every stream. */
int  mi355_lzss_batch_compress(mi355_lzss_batch *b, int with_header);
/* sizes of the last run: out_len[b] bytes (incl. header if requested), total = sum */
int  mi355_lzss_batch_sizes(mi355_lzss_batch *b, uint32_t *out_len, uint64_t *nbits, uint64_t *total);
/* HBM -> host: compressed blobs into out[out_off[b]:out_off[b+1]) */
int  mi355_lzss_batch_download(mi355_lzss_batch *b, uint8_t *out, const uint64_t *out_off,
                               int32_t *status);
/* RAW per-position results of the last run for blob `blob`, n entries (diagnostic).  Packed:
 * bits 30-31 type (0 none, 1 short, 2 long, 3 long into the dictionary), bit 29 PENDING,
 * bits 21-28 length - 1, bits 0-20 field.  field = the address bits as written (backref.go:66)
 * ONLY for a type 1 / 2 entry without the PENDING bit.  batch_compress leaves the address search
 * of most entries to the emitter: a PENDING entry's field is the suffix RANK that search starts
 * from, a type 3 entry's field is a DICTIONARY rank, and positions the phrase walk never asks
 * about (inside runs of 160+ equal bytes, compress.go:180-218) read as type 0.  Lengths and types
 * 1 / 2 are final.  For the reference's (type, address, length) of every position use
 * mi355_lzss_best_backrefs. */
int  mi355_lzss_batch_results(mi355_lzss_batch *b, uint32_t blob, uint32_t *packed, size_t n);
/* suffix array / inverse / lcp of blob `blob` from the last run (diagnostics for parity tests;
 * n+1 entries each: rank 0 is the end sentinel) */
int  mi355_lzss_batch_index(mi355_lzss_batch *b, uint32_t blob, int32_t *sa, int32_t *isa,
                            uint16_t *lcp, size_t n_plus_1);

/* batched decompression, device resident */
int  mi355_lzss_batch_upload_compressed(mi355_lzss_batch *b, const uint8_t *in, const uint64_t *in_off,
                                        uint32_t n_blobs, const uint64_t *out_cap_off);
int  mi355_lzss_batch_decompress(mi355_lzss_batch *b);
int  mi355_lzss_batch_download_decompressed(mi355_lzss_batch *b, uint8_t *out, const uint64_t *out_off,
                                            uint32_t *out_len, int32_t *status);

/* HIP-event time (ms) the stages of the last batch_compress / batch_decompress spent on the
 * batch's stream; names[k] points to static strings.  Returns the number of stages. */
int  mi355_lzss_batch_stage_times(const mi355_lzss_batch *b, const char **names, float *ms, int cap);

#ifdef __cplusplus
}
#endif
#endif
