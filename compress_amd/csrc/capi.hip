// capi.hip - host side of libmi355_lzss.so: the C ABI of include/mi355_lzss.h over the HIP kernels.
// There is no CPU fallback in here: every data-path entry point launches kernels or fails.
#include "../../include/mi355_lzss.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "kernels.h"

using namespace lzss;

// ---- errors ---------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail(MI355_LZSS_E_DEVICE, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

extern "C" const char *mi355_lzss_last_error(void) { return g_err; }

// ---- memory helpers ---------------------------------------------------------------------------------
struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {  // grows (never shrinks); contents are NOT preserved
        if (need <= bytes) return 0;
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
        HIP_TRY(hipMalloc(&p, need));
        bytes = need;
        return 0;
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T>
    T *as() const { return (T *)p; }
};

struct PinBuf {
    void *p = nullptr;
    size_t bytes = 0;
    int ensure(size_t need) {
        if (need <= bytes) return 0;
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
        HIP_TRY(hipHostMalloc(&p, need, hipHostMallocDefault));
        bytes = need;
        return 0;
    }
    void release() {
        if (p) (void)hipHostFree(p);
        p = nullptr;
        bytes = 0;
    }
    template <class T>
    T *as() const { return (T *)p; }
};

// pinned host memory for callers: transfers from / to such buffers are true DMA and overlap with the kernels
extern "C" void *mi355_lzss_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        (void)fail(MI355_LZSS_E_DEVICE, "hipHostMalloc of %zu bytes failed", bytes);
        return nullptr;
    }
    return p;
}
extern "C" void mi355_lzss_host_free(void *p) {
    if (p) (void)hipHostFree(p);
}

// ---- objects ----------------------------------------------------------------------------------------
enum { ST_SA = 0, ST_LCP, ST_WINDEX, ST_MHEAD, ST_MATCH, ST_DECIDE, ST_WALK, ST_EMIT, ST_DECOMP, ST_COUNT };
static const char *const kStageNames[ST_COUNT] = {"sa_build", "lcp", "window_index", "match_head", "match",
                                                  "decide", "walk", "emit", "decompress"};

struct Pipeline;

struct mi355_lzss_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool decode_only = false;   // ctx_create_decoder: the augmented dictionary's text only, no index
    std::vector<uint8_t> dict;  // augmented
    int32_t res_idx[2] = {0, 0};
    // dictionary index on the device
    DevBuf d_text, d_sa, d_isa, d_lcp, d_b1, d_b2, d_key, d_bkt;
    DictView view{};
    mi355_lzss_batch *scratch = nullptr;  // for the single-buffer entry points (created on first use)
    mi355_lzss_batch *na = nullptr;       // view of this device's static scratch array (mi355_lzss_write_noalloc)
    // side streams + events for running the index / match stages of sub-batches concurrently
    static constexpr int kMaxSplit = 4;
    hipStream_t side[kMaxSplit] = {};
    hipEvent_t fork_ev = nullptr, join_ev[kMaxSplit] = {};
    // per part: the head positions' brute-force search and the dictionary brackets (no LDS, few registers) run on a stream
    // of their own beside the LCP kernel, which fills a CU's LDS with half of its wave slots
    hipStream_t aux[kMaxSplit] = {};
    hipEvent_t sa_ev[kMaxSplit] = {}, head_ev[kMaxSplit] = {}, win_ev[kMaxSplit] = {};
    int split = 2;
    // workspaces of the entry points that may be called concurrently (compressed_size: compress.go:381-383)
    std::mutex pool_mu;
    std::vector<mi355_lzss_batch *> pool_free, pool_all;
    // host-buffer batches: chunks of about this many input bytes in a copy / compute / copy pipeline
    Pipeline *pipe = nullptr;
    uint64_t chunk_bytes = 200ull << 20;
    uint32_t wg_slots = 512;  // resident suffix-sort workgroups of the device: 2 x its CUs (set at creation)
};

struct mi355_lzss_batch {
    mi355_lzss_ctx *ctx = nullptr;
    hipStream_t stream = nullptr;  // the ctx's stream, or the batch's own (pool workspaces)
    bool own_stream = false;
    uint32_t max_blobs = 0;
    uint64_t cap_elems = 0;  // element capacity of the per-position arrays
    // per-position arrays
    DevBuf text, sortA, sortB, rkA, rkB, sa_tmp, rlen, sa, isa, lcp, b1, b2, res, adv, kind, exitoff, mpos;
    DevBuf pmin1, pmax1, pmin2, pmax2, samin1, samin2, dlb;  // per 64 entries
    DevBuf cdir;                                        // per 64 ranks and chunk
    DevBuf tbits;                   // per position
    DevBuf entry, tilebit;          // per 256-tile
    DevBuf segs, src_off, nbits, out_off, out_len, status;  // per blob
    DevBuf flat;                                   // the blobs as the caller holds them (back to back), before layout
    DevBuf out;                                    // compressed bytes, blob b at out_off[b]
    PinBuf h_stage;                                // pinned staging for downloads
    PinBuf h_meta;                                 // pinned: Seg[n] + source offsets[n] of the upload in flight
    PinBuf h_res;                                  // pinned: out_off[n+1], nbits[n], out_len[n], status[n] of the last run
    std::vector<Seg> h_segs;
    uint64_t *h_out_off = nullptr, *h_nbits = nullptr;  // views of h_res
    uint32_t *h_out_len = nullptr;
    int32_t *h_status = nullptr;
    std::vector<int32_t> h_pre_status;
    uint32_t n_blobs = 0, max_n = 0;
    uint64_t used_elems = 0, flat_bytes = 0;
    bool have_sizes = false;
    uint32_t header_bytes = 0, bit_off = 0;
    hipEvent_t ev[ST_COUNT + 1] = {};
    bool ev_valid[ST_COUNT + 1] = {};
    float stage_ms[ST_COUNT] = {};
    // decompression side
    DevBuf d_cin, d_cin_off, d_dout, d_dout_off, d_dlen, d_dstatus;
    std::vector<uint64_t> h_cin_off, h_dout_off;
    uint32_t n_dblobs = 0;
};

static int set_device(const mi355_lzss_ctx *ctx) {
    HIP_TRY(hipSetDevice(ctx->device));
    return 0;
}

// ---- batch workspace --------------------------------------------------------------------------------
static int batch_reserve(mi355_lzss_batch *b, uint32_t max_blobs, uint64_t elems) {
    int rc;
    if (elems > b->cap_elems) {
        const uint64_t e = elems;
#define R(buf, bytes_per) if ((rc = b->buf.ensure((size_t)(e * (bytes_per)))) != 0) return rc;
        R(text, 1) R(sortA, 8) R(sortB, 8) R(rlen, 1)  // rkA, rkB, sa_tmp: only with texts for the device-wide suffix sort (upload)
        R(sa, 4) R(isa, 4) R(mpos, 4) R(lcp, 2) R(res, 4) R(adv, 1) R(kind, 1) R(exitoff, 1) R(tbits, 2)
#undef R
        const uint64_t e64 = e / 64 + 1, e256 = e / 256 + 1;
        if ((rc = b->b1.ensure(e64 * 2)) != 0) return rc;
        if ((rc = b->b2.ensure(e64 * 2)) != 0) return rc;
        for (DevBuf *d : {&b->pmin1, &b->pmax1, &b->pmin2, &b->pmax2, &b->samin1, &b->samin2, &b->dlb})
            if ((rc = d->ensure(e64 * 4)) != 0) return rc;
        if ((rc = b->entry.ensure(e256 * 2)) != 0) return rc;
        if ((rc = b->tilebit.ensure(e256 * 8)) != 0) return rc;
        b->cap_elems = elems;
    }
    if (max_blobs > b->max_blobs) {
        const size_t m = max_blobs;
        if ((rc = b->segs.ensure(m * sizeof(Seg))) != 0) return rc;
        if ((rc = b->src_off.ensure(m * 8)) != 0) return rc;
        if ((rc = b->nbits.ensure(m * 8)) != 0) return rc;
        if ((rc = b->out_off.ensure((m + 1) * 8)) != 0) return rc;
        if ((rc = b->out_len.ensure(m * 4)) != 0) return rc;
        if ((rc = b->status.ensure(m * 4)) != 0) return rc;
        b->max_blobs = max_blobs;
    }
    return 0;
}

static SaWork sa_work(mi355_lzss_batch *b) {
    SaWork w;
    w.recA = b->sortA.as<uint64_t>(); w.recB = b->sortB.as<uint64_t>();
    w.rkA = b->rkA.as<int32_t>(); w.rkB = b->rkB.as<int32_t>();
    w.rlen = b->rlen.as<uint8_t>();
    w.sa = b->sa.as<int32_t>(); w.isa = b->isa.as<int32_t>();
    w.pairs = b->res.as<uint32_t>();
    w.mpos = b->mpos.as<uint32_t>();
    w.tmp = b->sa_tmp.as<uint8_t>(); w.tmp_bytes = b->sa_tmp.bytes;
    return w;
}
static BatchIndex batch_index(mi355_lzss_batch *b) {
    BatchIndex ix;
    ix.text = b->text.as<uint8_t>();
    ix.sa = b->sa.as<int32_t>(); ix.isa = b->isa.as<int32_t>();
    ix.rlen = b->rlen.as<uint8_t>();
    ix.lcp = b->lcp.as<uint16_t>(); ix.bmin1 = b->b1.as<uint16_t>(); ix.bmin2 = b->b2.as<uint16_t>();
    ix.win = b->sortA.as<uint64_t>();  // a segment's part of the sort buffer is free once the segment is sorted
    ix.pkey = b->sortB.as<uint64_t>();  // ... and so is its part of the second one
    ix.pmin1 = b->pmin1.as<int32_t>(); ix.pmax1 = b->pmax1.as<int32_t>();
    ix.pmin2 = b->pmin2.as<int32_t>(); ix.pmax2 = b->pmax2.as<int32_t>();
    ix.samin1 = b->samin1.as<int32_t>(); ix.samin2 = b->samin2.as<int32_t>();
    ix.dlb = b->dlb.as<int32_t>();
    ix.cdir = b->cdir.as<uint16_t>();
    ix.nch = std::max<uint32_t>(1, (b->max_n + LZSS_SHORT_WINDOW - 1) / LZSS_SHORT_WINDOW);
    ix.win_packed = (uint64_t)b->max_n + 1 <= (1u << 18) ? 1u : 0u;
    return ix;
}
static WalkWork walk_work(mi355_lzss_batch *b) {
    WalkWork w;
    w.adv = b->adv.as<uint8_t>(); w.kind = b->kind.as<uint8_t>(); w.exitoff = b->exitoff.as<uint8_t>();
    w.entry = b->entry.as<uint16_t>(); w.tilebit = b->tilebit.as<uint64_t>();
    w.tbits = b->tbits.as<uint16_t>();
    w.nbits = b->nbits.as<uint64_t>(); w.out_off = b->out_off.as<uint64_t>(); w.out_len = b->out_len.as<uint32_t>();
    return w;
}

static int mark(mi355_lzss_batch *b, int k) {
    if (!b->ev_valid[k]) {
        HIP_TRY(hipEventCreate(&b->ev[k]));
        b->ev_valid[k] = true;
    }
    HIP_TRY(hipEventRecord(b->ev[k], b->stream));
    return 0;
}

static int batch_new(mi355_lzss_ctx *ctx, uint32_t max_blobs, uint64_t max_total_bytes, bool own_stream, mi355_lzss_batch **out) {
    int rc;
    if ((rc = set_device(ctx)) != 0) return rc;
    mi355_lzss_batch *b = new mi355_lzss_batch();
    b->ctx = ctx;
    b->stream = ctx->stream;
    if (own_stream) {
        if (hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking) != hipSuccess) {
            delete b;
            return fail(MI355_LZSS_E_DEVICE, "hipStreamCreate failed");
        }
        b->own_stream = true;
    }
    if (max_blobs == 0) max_blobs = 1;
    const uint64_t elems = max_total_bytes + (uint64_t)max_blobs * (SEG_PAD + SEG_ALIGN);
    if ((rc = batch_reserve(b, max_blobs, elems)) != 0) {
        mi355_lzss_batch_destroy(b);
        return rc;
    }
    *out = b;
    return 0;
}

extern "C" int mi355_lzss_batch_create(mi355_lzss_ctx *ctx, uint32_t max_blobs, uint64_t max_total_bytes,
                                       mi355_lzss_batch **out) {
    if (!ctx || !out) return fail(MI355_LZSS_E_ARG, "null argument");
    return batch_new(ctx, max_blobs, max_total_bytes, false, out);
}

extern "C" void mi355_lzss_batch_destroy(mi355_lzss_batch *b) {
    if (!b) return;
    (void)hipSetDevice(b->ctx->device);
    DevBuf *bufs[] = {&b->text, &b->sortA, &b->sortB, &b->rkA, &b->rkB, &b->sa_tmp, &b->rlen, &b->sa, &b->isa,
                      &b->lcp, &b->mpos, &b->b1, &b->b2, &b->pmin1, &b->pmax1, &b->pmin2, &b->pmax2, &b->samin1, &b->samin2, &b->dlb, &b->cdir, &b->res, &b->adv, &b->kind, &b->exitoff, &b->entry, &b->tilebit,
                      &b->tbits, &b->segs, &b->src_off, &b->nbits, &b->out_off, &b->out_len, &b->status, &b->flat, &b->out,
                      &b->d_cin, &b->d_cin_off, &b->d_dout, &b->d_dout_off, &b->d_dlen, &b->d_dstatus};
    for (DevBuf *d : bufs) d->release();
    b->h_stage.release();
    b->h_meta.release();
    b->h_res.release();
    for (int k = 0; k <= ST_COUNT; ++k)
        if (b->ev_valid[k]) (void)hipEventDestroy(b->ev[k]);
    if (b->own_stream && b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
}

// ---- upload: the caller's bytes go to the device as they are (one copy), a kernel lays them out ------------
namespace lzss {
// text[seg.off + x] = flat[src[blob] + x] for x < n, zero up to the segment's stride.  16 bytes per thread: the
// source is read at any byte alignment (flat is readable 16 bytes past its end), the stores are aligned.
__global__ __launch_bounds__(256) void layout_text_kernel(const Seg *segs, const uint64_t *src, const uint8_t *flat, uint8_t *text) {
    const Seg sg = segs[blockIdx.y];
    const uint64_t stride = seg_stride(sg.n);
    const uint64_t x = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 16;
    if (x >= stride) return;
    typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
    typedef u32x4_t unaligned_t __attribute__((aligned(1)));
    u32x4_t v = {0u, 0u, 0u, 0u};
    if (x < sg.n) {
        v = *reinterpret_cast<const unaligned_t *>(flat + src[blockIdx.y] + x);
        const uint64_t left = sg.n - x;  // bytes of the text in this piece
        if (left < 16) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int64_t keep = (int64_t)left - 4 * q;  // bytes of word q that belong to the text
                if (keep <= 0) v[q] = 0u;
                else if (keep < 4) v[q] &= (1u << (8 * keep)) - 1u;
            }
        }
    }
    *reinterpret_cast<u32x4_t *>(text + sg.off + x) = v;
}
// the first min(*total, cap) bytes of out (16-byte granules) become zero
__global__ __launch_bounds__(256) void zero_out_kernel(uint8_t *out, const uint64_t *total, uint64_t cap) {
    const uint64_t t = *total < cap ? *total : cap;
    const uint64_t words = (t + 15) / 16;
    uint4 *o = reinterpret_cast<uint4 *>(out);
    for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < words; j += (uint64_t)gridDim.x * 256) o[j] = make_uint4(0, 0, 0, 0);
}
}  // namespace lzss

// Enqueues on `s`: the blobs' bytes, their segment table, the layout kernel.  Host-side state of the batch is set
// at once; the device side is ready when `s` has drained.  limit: blobs longer than this get E_INPUT_SIZE.
// lens (optional): blob k is in[in_off[k], in_off[k] + lens[k]) - the ranges may then overlap (the candidate Writes of
// mi355_lzss_pack_blob are prefixes of one another: their bytes go up once) and in_off has n_blobs entries.
static int upload_async(mi355_lzss_batch *b, const uint8_t *in, const uint64_t *in_off, uint32_t n_blobs,
                        const uint32_t *starts, uint64_t limit, hipStream_t s, const uint64_t *lens = nullptr) {
    int rc;
    uint64_t elems = 0;
    uint32_t max_n = 0;
    b->h_segs.assign(n_blobs, Seg{});
    b->h_pre_status.assign(n_blobs, 0);
    if ((rc = b->h_meta.ensure((size_t)std::max<uint32_t>(n_blobs, 1) * (sizeof(Seg) + 8))) != 0) return rc;
    Seg *hs = b->h_meta.as<Seg>();
    uint64_t *hsrc = reinterpret_cast<uint64_t *>(hs + n_blobs);
    uint64_t base = n_blobs ? in_off[0] : 0, top = n_blobs && !lens ? in_off[n_blobs] : 0, coded_bytes = 0;
    if (lens)
        for (uint32_t k = 0; k < n_blobs; ++k) {
            base = std::min(base, in_off[k]);
            top = std::max(top, in_off[k] + lens[k]);
        }
    for (uint32_t k = 0; k < n_blobs; ++k) {
        if (!lens && in_off[k + 1] < in_off[k]) return fail(MI355_LZSS_E_ARG, "in_off must be non-decreasing");
        uint64_t n = lens ? lens[k] : in_off[k + 1] - in_off[k];
        if (n > limit) {  // compress.go:473-475 (compress.go:390 for CompressedSize256k)
            b->h_pre_status[k] = MI355_LZSS_E_INPUT_SIZE;
            n = 0;
        }
        uint32_t st = starts ? starts[k] : 0;
        if (st > n) st = (uint32_t)n;
        coded_bytes += n - st;
        b->h_segs[k] = hs[k] = Seg{elems, (uint32_t)n, st};
        hsrc[k] = in_off[k] - base;
        elems += seg_stride(n);
        max_n = std::max<uint32_t>(max_n, (uint32_t)n);
    }
    const uint64_t flat_bytes = n_blobs ? top - base : 0;
    if ((rc = batch_reserve(b, std::max<uint32_t>(n_blobs, 1), std::max<uint64_t>(elems, SEG_ALIGN))) != 0) return rc;
    if ((rc = b->flat.ensure((size_t)flat_bytes + 16)) != 0) return rc;
    if (max_n + 1 > SA_BITMAP_MAX_N) {  // the device-wide sort of large texts (sa_large.hip) keeps two more arrays in HBM
        const uint64_t e = b->cap_elems;
        if ((rc = b->rkA.ensure((size_t)(e * 4))) != 0 || (rc = b->rkB.ensure((size_t)(e * 4))) != 0 ||
            (rc = b->sa_tmp.ensure(sa_large_temp_bytes(max_n + 1))) != 0)
            return rc;
    }
    {
        const uint64_t nch = std::max<uint32_t>(1, (max_n + LZSS_SHORT_WINDOW - 1) / LZSS_SHORT_WINDOW);
        if ((rc = b->cdir.ensure((size_t)((std::max<uint64_t>(elems, SEG_ALIGN) / 64 + 1) * nch * 2))) != 0) return rc;
    }
    if (flat_bytes) HIP_TRY(hipMemcpyAsync(b->flat.p, in + base, (size_t)flat_bytes, hipMemcpyHostToDevice, s));
    if (n_blobs) {
        HIP_TRY(hipMemcpyAsync(b->segs.p, hs, n_blobs * sizeof(Seg), hipMemcpyHostToDevice, s));
        HIP_TRY(hipMemcpyAsync(b->src_off.p, hsrc, n_blobs * 8, hipMemcpyHostToDevice, s));
        const uint64_t stride = seg_stride(max_n);
        dim3 g((uint32_t)((stride / 16 + 255) / 256), n_blobs);
        hipLaunchKernelGGL(layout_text_kernel, g, dim3(256), 0, s, b->segs.as<Seg>(), b->src_off.as<uint64_t>(), b->flat.as<uint8_t>(),
                           b->text.as<uint8_t>());
    }
    b->n_blobs = n_blobs;
    b->max_n = max_n;
    b->used_elems = elems;
    b->flat_bytes = lens ? std::max(flat_bytes, coded_bytes) : flat_bytes;  // (sizes the output buffer: what gets encoded)
    b->have_sizes = false;
    return 0;
}

extern "C" int mi355_lzss_batch_upload(mi355_lzss_batch *b, const uint8_t *in, const uint64_t *in_off, uint32_t n_blobs,
                                       const uint32_t *starts) {
    if (!b || !in_off || (n_blobs && !in)) return fail(MI355_LZSS_E_ARG, "null argument");
    if (b->ctx->decode_only) return fail(MI355_LZSS_E_ARG, "the context was created for decompression only");
    int rc;
    if ((rc = set_device(b->ctx)) != 0) return rc;
    if ((rc = upload_async(b, in, in_off, n_blobs, starts, LZSS_MAX_INPUT_SIZE, b->stream)) != 0) return rc;
    HIP_TRY(hipStreamSynchronize(b->stream));
    HIP_TRY(hipGetLastError());
    return 0;
}

// index build + match search for the uploaded blobs
static int run_index_and_match(mi355_lzss_batch *b, bool skip_unqueried) {
    mi355_lzss_ctx *ctx = b->ctx;
    hipStream_t sm = b->stream;
    int rc;
    const Seg *d_segs = b->segs.as<Seg>();
    int split = ctx->split;
    // the side streams belong to the ctx: only batches on the ctx's own stream (one at a time) may use them
    if (b->n_blobs < 8u * (uint32_t)split || b->max_n + 1 > SA_BITMAP_MAX_N || sm != ctx->stream) split = 1;
    if (split == 1) {
        if ((rc = mark(b, ST_SA)) != 0) return rc;
        HIP_TRY((hipError_t)launch_sa_build(d_segs, b->h_segs.data(), b->n_blobs, b->max_n, b->text.as<uint8_t>(), sa_work(b), sm));
        if ((rc = mark(b, ST_LCP)) != 0) return rc;
        launch_lcp(d_segs, b->n_blobs, b->max_n, batch_index(b), sm);
        if ((rc = mark(b, ST_WINDEX)) != 0) return rc;
        {
            const SaWork wk = sa_work(b);  // (scratch for the device-wide index of texts above 2^19 suffixes)
            launch_window_index(d_segs, b->n_blobs, b->max_n, batch_index(b), sm, true, true, b->h_segs.data(), &wk);
        }
        if ((rc = mark(b, ST_MHEAD)) != 0) return rc;
        launch_match_head(d_segs, b->n_blobs, b->max_n, batch_index(b), ctx->view, b->res.as<uint32_t>(), sm);
        if ((rc = mark(b, ST_MATCH)) != 0) return rc;
        launch_match(d_segs, b->n_blobs, b->max_n, batch_index(b), ctx->view, b->res.as<uint32_t>(), skip_unqueried, sm);
        if ((rc = mark(b, ST_DECIDE)) != 0) return rc;
        HIP_TRY(hipGetLastError());
        return 0;
    }
    // sub-batches on side streams: the tail of one sub-batch's suffix sort (a last wave of workgroups that fills
    // only part of the CUs) overlaps the next stages of another; per-stage events lose their meaning here
    if (!ctx->fork_ev) HIP_TRY(hipEventCreateWithFlags(&ctx->fork_ev, hipEventDisableTiming));
    for (int k = 1; k < split; ++k) {
        if (!ctx->side[k]) HIP_TRY(hipStreamCreateWithFlags(&ctx->side[k], hipStreamNonBlocking));
        if (!ctx->join_ev[k]) HIP_TRY(hipEventCreateWithFlags(&ctx->join_ev[k], hipEventDisableTiming));
    }
    for (int k = 0; k < split; ++k) {
        if (!ctx->aux[k]) HIP_TRY(hipStreamCreateWithFlags(&ctx->aux[k], hipStreamNonBlocking));
        if (!ctx->sa_ev[k]) HIP_TRY(hipEventCreateWithFlags(&ctx->sa_ev[k], hipEventDisableTiming));
        if (!ctx->head_ev[k]) HIP_TRY(hipEventCreateWithFlags(&ctx->head_ev[k], hipEventDisableTiming));
        if (!ctx->win_ev[k]) HIP_TRY(hipEventCreateWithFlags(&ctx->win_ev[k], hipEventDisableTiming));
    }
    // match_kernel leaves the smallest-rank searches to the emitter (launch_match: defer), so nothing reads the window
    // index before the emitter does
    bool deferred = skip_unqueried && b->max_n + 1 < RES_DEFER_MAX_N;
#ifdef LZSS_DEBUG_KNOBS
    if (getenv("MI355_LZSS_NO_DEFER")) deferred = false;
#endif
    if ((rc = mark(b, ST_SA)) != 0) return rc;
    HIP_TRY(hipEventRecord(ctx->fork_ev, sm));
    for (int k = 0; k < split; ++k) {
        hipStream_t st = k == 0 ? sm : ctx->side[k];
        const uint32_t lo = (uint32_t)((uint64_t)b->n_blobs * k / split), hi = (uint32_t)((uint64_t)b->n_blobs * (k + 1) / split);
        const Seg *sg = d_segs + lo;
        const uint32_t cnt = hi - lo;
        if (k) HIP_TRY(hipStreamWaitEvent(st, ctx->fork_ev, 0));
        HIP_TRY((hipError_t)launch_sa_build(sg, b->h_segs.data() + lo, cnt, b->max_n, b->text.as<uint8_t>(), sa_work(b), st));
        HIP_TRY(hipEventRecord(ctx->sa_ev[k], st));
        HIP_TRY(hipStreamWaitEvent(ctx->aux[k], ctx->sa_ev[k], 0));
        launch_match_head(sg, cnt, b->max_n, batch_index(b), ctx->view, b->res.as<uint32_t>(), ctx->aux[k]);
        HIP_TRY(hipEventRecord(ctx->head_ev[k], ctx->aux[k]));
        // the window index (only the emitter's searches read it) on the side stream, beside the LCP and match kernels
        launch_window_index(sg, cnt, b->max_n, batch_index(b), ctx->aux[k], true, false);
        HIP_TRY(hipEventRecord(ctx->win_ev[k], ctx->aux[k]));
        launch_lcp(sg, cnt, b->max_n, batch_index(b), st);
        launch_window_index(sg, cnt, b->max_n, batch_index(b), st, false, true);  // block minima of the suffix array, second level
        HIP_TRY(hipStreamWaitEvent(st, ctx->head_ev[k], 0));
        if (!deferred) HIP_TRY(hipStreamWaitEvent(st, ctx->win_ev[k], 0));  // match_kernel itself searches the window index
        launch_match(sg, cnt, b->max_n, batch_index(b), ctx->view, b->res.as<uint32_t>(), skip_unqueried, st);
        if (k) HIP_TRY(hipEventRecord(ctx->join_ev[k], st));
    }
    for (int k = 1; k < split; ++k) HIP_TRY(hipStreamWaitEvent(sm, ctx->join_ev[k], 0));
    for (int k = 0; k < split; ++k) HIP_TRY(hipStreamWaitEvent(sm, ctx->win_ev[k], 0));
    for (int k = ST_LCP; k <= ST_DECIDE; ++k)
        if ((rc = mark(b, k)) != 0) return rc;
    HIP_TRY(hipGetLastError());
    return 0;
}

static int collect_stage_times(mi355_lzss_batch *b, int first, int last) {
    for (int k = 0; k < ST_COUNT; ++k) b->stage_ms[k] = 0.f;
    for (int k = first; k < last; ++k) {
        float ms = 0.f;
        if (b->ev_valid[k] && b->ev_valid[k + 1]) HIP_TRY(hipEventElapsedTime(&ms, b->ev[k], b->ev[k + 1]));
        b->stage_ms[k] = ms;
    }
    return 0;
}

static int enqueue_emit(mi355_lzss_batch *b) {
    mi355_lzss_ctx *ctx = b->ctx;
    WalkWork ww = walk_work(b);
    // emit ORs the bytes that tiles share: the stream bytes must start out zero (their total is known on the device only)
    hipLaunchKernelGGL(zero_out_kernel, dim3(512), dim3(256), 0, b->stream, b->out.as<uint8_t>(), ww.out_off + b->n_blobs, (uint64_t)b->out.bytes);
    launch_emit(b->segs.as<Seg>(), b->n_blobs, b->max_n, b->text.as<uint8_t>(), b->res.as<uint32_t>(), batch_index(b), ctx->view, ww,
                b->header_bytes, b->bit_off, (uint32_t)ctx->dict.size(), ctx->res_idx[0], ctx->res_idx[1], b->out.as<uint8_t>(),
                b->out.bytes, b->status.as<int32_t>(), b->stream);
    return 0;
}

// decide + walk (+ emit) and the read-back of the sizes, all enqueued on the batch's stream WITHOUT a host
// synchronisation: the output buffer is sized from the input (ratio >= 0.8) and finish_walk_emit repairs the rare
// batch that expands beyond it.  bit_off: first bit inside the first stream byte.
static int enqueue_walk_emit(mi355_lzss_batch *b, uint32_t header_bytes, uint32_t bit_off, bool emit) {
    hipStream_t sm = b->stream;
    int rc;
    const Seg *d_segs = b->segs.as<Seg>();
    WalkWork ww = walk_work(b);
    b->header_bytes = header_bytes;
    b->bit_off = bit_off;
    launch_decide(d_segs, b->n_blobs, b->max_n, b->text.as<uint8_t>(), b->rlen.as<uint8_t>(), b->res.as<uint32_t>(), ww, sm);
    if ((rc = mark(b, ST_WALK)) != 0) return rc;
    launch_walk(d_segs, b->n_blobs, ww, header_bytes, bit_off, sm);
    launch_out_offsets(b->n_blobs, ww, sm);
    if ((rc = mark(b, ST_EMIT)) != 0) return rc;
    const uint32_t nb = b->n_blobs;
    if ((rc = b->h_res.ensure((size_t)(nb + 1) * 8 + (size_t)nb * 16 + 64)) != 0) return rc;
    b->h_out_off = b->h_res.as<uint64_t>();
    b->h_nbits = b->h_out_off + nb + 1;
    b->h_out_len = reinterpret_cast<uint32_t *>(b->h_nbits + nb);
    b->h_status = reinterpret_cast<int32_t *>(b->h_out_len + nb);
    b->h_out_off[nb] = 0;
    for (uint32_t k = 0; k < nb; ++k) b->h_status[k] = 0;
    if (emit) {
        const uint64_t guess = b->flat_bytes + b->flat_bytes / 4 + (uint64_t)nb * (16 + header_bytes) + 4096;
        if ((rc = b->out.ensure((size_t)guess)) != 0) return rc;
        if ((rc = enqueue_emit(b)) != 0) return rc;
    }
    if ((rc = mark(b, ST_EMIT + 1)) != 0) return rc;
    if (nb) {
        HIP_TRY(hipMemcpyAsync(b->h_out_off, ww.out_off, (nb + 1) * 8, hipMemcpyDeviceToHost, sm));
        HIP_TRY(hipMemcpyAsync(b->h_out_len, ww.out_len, nb * 4, hipMemcpyDeviceToHost, sm));
        HIP_TRY(hipMemcpyAsync(b->h_nbits, ww.nbits, nb * 8, hipMemcpyDeviceToHost, sm));
        if (emit) HIP_TRY(hipMemcpyAsync(b->h_status, b->status.p, nb * 4, hipMemcpyDeviceToHost, sm));
    }
    HIP_TRY(hipGetLastError());
    return 0;
}

// after the stream has drained: the sizes are on the host; a batch that outgrew the output buffer is emitted again
static int finish_walk_emit(mi355_lzss_batch *b, bool emit) {
    int rc;
    const uint32_t nb = b->n_blobs;
    const uint64_t total = nb ? b->h_out_off[nb] : 0;
    if (emit && total > b->out.bytes) {
        if ((rc = b->out.ensure((size_t)total)) != 0) return rc;
        if ((rc = enqueue_emit(b)) != 0) return rc;
        HIP_TRY(hipMemcpyAsync(b->h_status, b->status.p, nb * 4, hipMemcpyDeviceToHost, b->stream));
        HIP_TRY(hipStreamSynchronize(b->stream));
        HIP_TRY(hipGetLastError());
    }
    b->have_sizes = true;
    return 0;
}

static int run_walk_emit(mi355_lzss_batch *b, uint32_t header_bytes, uint32_t bit_off, bool emit) {
    int rc;
    if ((rc = enqueue_walk_emit(b, header_bytes, bit_off, emit)) != 0) return rc;
    HIP_TRY(hipStreamSynchronize(b->stream));
    HIP_TRY(hipGetLastError());
    return finish_walk_emit(b, emit);
}

extern "C" int mi355_lzss_batch_compress(mi355_lzss_batch *b, int with_header) {
    if (!b) return fail(MI355_LZSS_E_ARG, "null argument");
    if (b->ctx->decode_only) return fail(MI355_LZSS_E_ARG, "the context was created for decompression only");
    int rc;
    if ((rc = set_device(b->ctx)) != 0) return rc;
    if ((rc = run_index_and_match(b, true)) != 0) return rc;
    if ((rc = run_walk_emit(b, with_header ? LZSS_HEADER_SIZE : 0, 0, true)) != 0) return rc;
    return collect_stage_times(b, ST_SA, ST_EMIT + 1);
}

extern "C" int mi355_lzss_batch_sizes(mi355_lzss_batch *b, uint32_t *out_len, uint64_t *nbits, uint64_t *total) {
    if (!b || !b->have_sizes) return fail(MI355_LZSS_E_ARG, "no compressed batch");
    uint64_t t = 0;
    for (uint32_t k = 0; k < b->n_blobs; ++k) {
        if (out_len) out_len[k] = b->h_out_len[k];
        if (nbits) nbits[k] = b->h_nbits[k];
        t += b->h_out_len[k];
    }
    if (total) *total = t;
    return 0;
}

// the blobs of the last run from the pinned image `img` of the device output into the caller's layout
#ifndef PIPE_GRADED
#define PIPE_GRADED 1
#endif
#ifndef SCATTER_THREADS
#define SCATTER_THREADS 4
#endif
static int scatter_outputs(mi355_lzss_batch *b, const uint8_t *img, uint8_t *out, const uint64_t *out_off, uint32_t *out_len,
                           int32_t *status) {
    const uint32_t nb = b->n_blobs;
    auto range = [&](uint32_t lo, uint32_t hi) -> int {
        int worst = 0;
        for (uint32_t k = lo; k < hi; ++k) {  // also with nothing to copy: every blob gets its status
            int32_t st = b->h_pre_status[k] ? b->h_pre_status[k] : b->h_status[k];
            const uint64_t cap = out_off[k + 1] - out_off[k];
            if (st == 0 && b->h_out_len[k] > cap) st = MI355_LZSS_E_CAPACITY;
            if (st == 0 && b->h_out_len[k]) memcpy(out + out_off[k], img + b->h_out_off[k], b->h_out_len[k]);
            if (out_len) out_len[k] = st == 0 ? b->h_out_len[k] : 0;  // a failed blob has no length
            if (status) status[k] = st;
            if (st && !worst) worst = st;
        }
        return worst;
    };
    // a chunk of the pipeline is tens of megabytes: one thread copies ~10 GB/s, and the last chunk's copy overlaps nothing
    constexpr int NT = SCATTER_THREADS;
    const uint64_t total = nb ? b->h_out_off[nb] : 0;
    if (NT < 2 || total < (8ull << 20) || nb < 4 * NT) return range(0, nb);
    int worst[NT] = {};
    std::thread th[NT > 1 ? NT - 1 : 1];
    for (int t = 1; t < NT; ++t)
        th[t - 1] = std::thread([&, t] { worst[t] = range((uint32_t)((uint64_t)nb * t / NT), (uint32_t)((uint64_t)nb * (t + 1) / NT)); });
    worst[0] = range(0, (uint32_t)((uint64_t)nb / NT));
    for (int t = 1; t < NT; ++t) th[t - 1].join();
    for (int t = 0; t < NT; ++t)
        if (worst[t]) return worst[t];
    return 0;
}

extern "C" int mi355_lzss_batch_download(mi355_lzss_batch *b, uint8_t *out, const uint64_t *out_off, int32_t *status) {
    if (!b || !b->have_sizes || !out_off || (b->n_blobs && !out)) return fail(MI355_LZSS_E_ARG, "bad argument");
    int rc;
    if ((rc = set_device(b->ctx)) != 0) return rc;
    const uint64_t total = b->n_blobs ? b->h_out_off[b->n_blobs] : 0;
    if (total) {
        if ((rc = b->h_stage.ensure((size_t)total)) != 0) return rc;
        HIP_TRY(hipMemcpyAsync(b->h_stage.p, b->out.p, (size_t)total, hipMemcpyDeviceToHost, b->stream));
        HIP_TRY(hipStreamSynchronize(b->stream));
    }
    const int worst = scatter_outputs(b, b->h_stage.as<uint8_t>(), out, out_off, nullptr, status);
    if (worst && !status) return fail(worst, "blob failed with status %d", worst);
    return 0;
}

extern "C" int mi355_lzss_batch_results(mi355_lzss_batch *b, uint32_t blob, uint32_t *packed, size_t n) {
    if (!b || blob >= b->n_blobs || !packed) return fail(MI355_LZSS_E_ARG, "bad argument");
    int rc;
    if ((rc = set_device(b->ctx)) != 0) return rc;
    const Seg &s = b->h_segs[blob];
    if (n > s.n) n = s.n;
    if (n) HIP_TRY(hipMemcpy(packed, b->res.as<uint32_t>() + s.off, n * 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int mi355_lzss_batch_index(mi355_lzss_batch *b, uint32_t blob, int32_t *sa, int32_t *isa, uint16_t *lcp,
                                      size_t n_plus_1) {
    if (!b || blob >= b->n_blobs) return fail(MI355_LZSS_E_ARG, "bad argument");
    int rc;
    if ((rc = set_device(b->ctx)) != 0) return rc;
    const Seg &s = b->h_segs[blob];
    size_t N = std::min<size_t>(n_plus_1, (size_t)s.n + 1);
    if (sa) HIP_TRY(hipMemcpy(sa, b->sa.as<int32_t>() + s.off, N * 4, hipMemcpyDeviceToHost));
    if (isa) {  // the suffix sort leaves only the head of the inverse array behind: this view fills the segment's in
        launch_isa_from_sa(b->segs.as<Seg>() + blob, 1, s.n, b->sa.as<int32_t>(), b->isa.as<int32_t>(), b->stream);
        HIP_TRY(hipStreamSynchronize(b->stream));
        HIP_TRY(hipMemcpy(isa, b->isa.as<int32_t>() + s.off, N * 4, hipMemcpyDeviceToHost));
    }
    if (lcp) HIP_TRY(hipMemcpy(lcp, b->lcp.as<uint16_t>() + s.off, N * 2, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int mi355_lzss_batch_stage_times(const mi355_lzss_batch *b, const char **names, float *ms, int cap) {
    if (!b) return 0;
    int k = 0;
    for (; k < ST_COUNT && k < cap; ++k) {
        if (names) names[k] = kStageNames[k];
        if (ms) ms[k] = b->stage_ms[k];
    }
    return k;
}

// ---- allocation-free Write ---------------------------------------------------------------------------------
// mi355_lzss_write_noalloc: the kernels of mi355_lzss_write on a workspace that lives in a file-scope device array
// instead of hipMalloc'ed buffers, inputs and outputs in the caller's device memory, everything enqueued on the
// caller's stream - for callers that cannot allocate (HIP graph capture, serving loops).  Every GPU the library runs on
// holds one copy of the array until the process ends; all calls on a GPU share it, so calls that overlap on one GPU race.
constexpr uint32_t NOALLOC_MAX_N = MI355_LZSS_NOALLOC_MAX_INPUT;
constexpr uint64_t NOALLOC_ELEMS = seg_stride(NOALLOC_MAX_N);
constexpr size_t NOALLOC_SCRATCH_BYTES = (size_t)NOALLOC_ELEMS * 44 + (64u << 10);  // ~41 B per position + summaries (checked in noalloc_bind)
static __device__ __align__(256) unsigned char scratch[NOALLOC_SCRATCH_BYTES];

// The array's name is a host-side stand-in, not a device address: the address on each device comes from
// hipGetSymbolAddress with that device current, once per device, and only that pointer is ever used.
constexpr int NOALLOC_MAX_DEVICES = 64;
static std::mutex g_na_mu;
static unsigned char *g_na_ptr[NOALLOC_MAX_DEVICES] = {};

static int noalloc_scratch_ptr(int device, unsigned char **out) {  // the caller has made `device` current
    if (device < 0 || device >= NOALLOC_MAX_DEVICES) return fail(MI355_LZSS_E_DEVICE, "device %d beyond the scratch table", device);
    std::lock_guard<std::mutex> lk(g_na_mu);
    if (!g_na_ptr[device]) {
        unsigned char *ptr = nullptr;
        HIP_TRY(hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch)));
        if (!ptr) return fail(MI355_LZSS_E_DEVICE, "hipGetSymbolAddress returned a null scratch address");
        g_na_ptr[device] = ptr;
    }
    *out = g_na_ptr[device];
    return 0;
}

// lays the arrays of a one-blob batch of up to NOALLOC_MAX_N bytes out inside [base, base + NOALLOC_SCRATCH_BYTES)
static int noalloc_bind(mi355_lzss_ctx *ctx) {
    unsigned char *base = nullptr;
    int rc;
    if ((rc = noalloc_scratch_ptr(ctx->device, &base)) != 0) return rc;
    mi355_lzss_batch *b = new mi355_lzss_batch();
    b->ctx = ctx;
    size_t used = 0;
    auto take = [&](DevBuf &d, size_t bytes) {
        d.p = base + used;
        d.bytes = bytes;
        used += (bytes + 255) & ~(size_t)255;
    };
    const uint64_t e = NOALLOC_ELEMS, e64 = e / 64 + 1, e256 = e / 256 + 1;
    const uint64_t nch = (NOALLOC_MAX_N + LZSS_SHORT_WINDOW - 1) / LZSS_SHORT_WINDOW;
    take(b->text, e); take(b->sortA, e * 8); take(b->sortB, e * 8); take(b->rlen, e);
    take(b->sa, e * 4); take(b->isa, e * 4); take(b->mpos, e * 4); take(b->lcp, e * 2); take(b->res, e * 4);
    take(b->adv, e); take(b->kind, e); take(b->exitoff, e); take(b->tbits, e * 2);
    take(b->b1, e64 * 2); take(b->b2, e64 * 2);
    for (DevBuf *d : {&b->pmin1, &b->pmax1, &b->pmin2, &b->pmax2, &b->samin1, &b->samin2, &b->dlb}) take(*d, e64 * 4);
    take(b->cdir, e64 * nch * 2);
    take(b->entry, e256 * 2); take(b->tilebit, e256 * 8);
    take(b->segs, sizeof(Seg)); take(b->nbits, 8); take(b->out_off, 16); take(b->out_len, 4); take(b->status, 4);
    b->cap_elems = e;
    b->max_blobs = 1;
    if (used > NOALLOC_SCRATCH_BYTES) {
        delete b;
        return fail(MI355_LZSS_E_DEVICE, "scratch layout needs %zu bytes, the array has %zu", used, (size_t)NOALLOC_SCRATCH_BYTES);
    }
    ctx->na = b;
    return 0;
}

namespace lzss {
__global__ void noalloc_seg_kernel(Seg *s, uint32_t n, uint32_t start) { s[0] = Seg{0, n, start}; }
}

extern "C" int mi355_lzss_write_noalloc(mi355_lzss_ctx *ctx, const uint8_t *d_inbuf, size_t n, size_t start, unsigned bit_off,
                                        uint8_t *d_out, size_t out_cap, uint64_t *d_nbits, int32_t *d_status, void *stream) {
    if (!ctx || !ctx->na || (n && !d_inbuf) || !d_out || !d_nbits || !d_status) return fail(MI355_LZSS_E_ARG, "null argument");
    if (n > NOALLOC_MAX_N)
        return fail(MI355_LZSS_E_INPUT_SIZE, "input size must be <= %u for the allocation-free entry point", NOALLOC_MAX_N);
    if (start > n) return fail(MI355_LZSS_E_ARG, "start beyond the end of the buffer");
    if (bit_off > 7) return fail(MI355_LZSS_E_ARG, "bit_off must be 0..7");
    if (out_cap == 0) return fail(MI355_LZSS_E_CAPACITY, "the output buffer needs at least one byte");
    int rc;
    if ((rc = set_device(ctx)) != 0) return rc;
    mi355_lzss_batch *b = ctx->na;
    hipStream_t sm = (hipStream_t)stream;
    const uint64_t stride = seg_stride(n);
    // the blob's text with its zero padding; everything else of the last call's contents is rewritten before it is read
    if (n) HIP_TRY(hipMemcpyAsync(b->text.p, d_inbuf, n, hipMemcpyDeviceToDevice, sm));
    HIP_TRY(hipMemsetAsync((uint8_t *)b->text.p + n, 0, (size_t)(stride - n), sm));
    hipLaunchKernelGGL(noalloc_seg_kernel, dim3(1), dim3(1), 0, sm, b->segs.as<Seg>(), (uint32_t)n, (uint32_t)start);
    b->n_blobs = 1;
    b->max_n = (uint32_t)n;
    const Seg *d_segs = b->segs.as<Seg>();
    HIP_TRY((hipError_t)launch_sa_build(d_segs, nullptr, 1, b->max_n, b->text.as<uint8_t>(), sa_work(b), sm));
    launch_lcp(d_segs, 1, b->max_n, batch_index(b), sm);
    launch_window_index(d_segs, 1, b->max_n, batch_index(b), sm);
    launch_match_head(d_segs, 1, b->max_n, batch_index(b), ctx->view, b->res.as<uint32_t>(), sm);
    launch_match(d_segs, 1, b->max_n, batch_index(b), ctx->view, b->res.as<uint32_t>(), true, sm);
    WalkWork ww = walk_work(b);
    launch_decide(d_segs, 1, b->max_n, b->text.as<uint8_t>(), b->rlen.as<uint8_t>(), b->res.as<uint32_t>(), ww, sm);
    launch_walk(d_segs, 1, ww, 0, bit_off, sm);
    launch_out_offsets(1, ww, sm);
    // emit ORs the bytes that tiles share into a zeroed output; worst case 37 bits per position
    const size_t worst = (size_t)((bit_off + (uint64_t)LZSS_DYN_BITS * (n - start) + 7) / 8);
    HIP_TRY(hipMemsetAsync(d_out, 0, std::max<size_t>(1, std::min(out_cap, worst)), sm));
    launch_emit(d_segs, 1, b->max_n, b->text.as<uint8_t>(), b->res.as<uint32_t>(), batch_index(b), ctx->view, ww, 0, bit_off,
                (uint32_t)ctx->dict.size(), ctx->res_idx[0], ctx->res_idx[1], d_out, out_cap, b->status.as<int32_t>(), sm);
    HIP_TRY(hipMemcpyAsync(d_nbits, ww.nbits, 8, hipMemcpyDeviceToDevice, sm));
    HIP_TRY(hipMemcpyAsync(d_status, b->status.p, 4, hipMemcpyDeviceToDevice, sm));
    HIP_TRY(hipGetLastError());
    return 0;
}

// ---- context ----------------------------------------------------------------------------------------
static void augment_dict(const uint8_t *dict, size_t dict_len, std::vector<uint8_t> &aug) {  // lzss/compress.go:70-88
    aug.assign(dict, dict + dict_len);
    uint8_t found = 0;
    bool both = false;
    for (size_t i = 0; i < dict_len && !both; ++i) {
        if (dict[i] == LZSS_SYMBOL_SHORT) found |= 2;
        else if (dict[i] == LZSS_SYMBOL_DYNAMIC) found |= 4;
        else continue;
        both = (found == 6);
    }
    if (!both) {
        aug.push_back(LZSS_SYMBOL_SHORT);
        aug.push_back(LZSS_SYMBOL_DYNAMIC);
    }
}

static int ensure_scratch(mi355_lzss_ctx *ctx) {
    if (ctx->scratch) return 0;
    return batch_new(ctx, 1, 0, false, &ctx->scratch);
}

static int ctx_create(const uint8_t *dict, size_t dict_len, int device, bool decode_only, mi355_lzss_ctx **out) {
    if (!out || (dict_len && !dict)) return fail(MI355_LZSS_E_ARG, "null argument");
    *out = nullptr;
    std::vector<uint8_t> aug;
    augment_dict(dict, dict_len, aug);
    if (aug.size() > LZSS_MAX_DICT_SIZE)  // compress.go:39-41
        return fail(MI355_LZSS_E_DICT_SIZE, "dict size must be <= %d", LZSS_MAX_DICT_SIZE);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(MI355_LZSS_E_DEVICE, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= ndev) return fail(MI355_LZSS_E_DEVICE, "device %d out of range (%d devices)", device, ndev);

    mi355_lzss_ctx *ctx = new mi355_lzss_ctx();
    ctx->device = device;
    ctx->dict = aug;
    ctx->decode_only = decode_only;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0)
            ctx->wg_slots = 2u * (uint32_t)cus;
    }
    // dictReservedIdx, compress.go:47-59: last occurrence of each symbol up to the point both were seen
    {
        bool fe = false, ff = false;
        for (size_t i = 0; i < aug.size(); ++i) {
            if (aug[i] == LZSS_SYMBOL_DYNAMIC) { ctx->res_idx[1] = (int32_t)i; ff = true; }
            else if (aug[i] == LZSS_SYMBOL_SHORT) { ctx->res_idx[0] = (int32_t)i; fe = true; }
            else continue;
            if (fe && ff) break;
        }
    }
    int rc = 0;
    auto bail = [&](int code) {
        mi355_lzss_ctx_destroy(ctx);
        return code;
    };
    if ((rc = set_device(ctx)) != 0) return bail(rc);
    if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess)
        return bail(fail(MI355_LZSS_E_DEVICE, "hipStreamCreate failed"));
    hipStream_t sm = ctx->stream;
    const size_t stride = (size_t)seg_stride(aug.size());

    if (decode_only) {  // Decompress (decompress.go:34) only augments the dictionary: text on the device, nothing else
        if ((rc = ctx->d_text.ensure(stride)) != 0) return bail(rc);
        if (hipMemsetAsync(ctx->d_text.p, 0, stride, sm) != hipSuccess ||
            hipMemcpyAsync(ctx->d_text.p, aug.data(), aug.size(), hipMemcpyHostToDevice, sm) != hipSuccess ||
            hipStreamSynchronize(sm) != hipSuccess)
            return bail(fail(MI355_LZSS_E_DEVICE, "uploading the dictionary failed"));
        ctx->view.T = ctx->d_text.as<uint8_t>();
        ctx->view.n = (uint32_t)aug.size();
        *out = ctx;
        return 0;
    }

    if (const char *e = getenv("MI355_LZSS_SPLIT")) {
        const int v = atoi(e);
        ctx->split = v < 1 ? 1 : (v > mi355_lzss_ctx::kMaxSplit ? mi355_lzss_ctx::kMaxSplit : v);
    }
    // dictionary index (compress.go:64) built with the same kernels as the input index
    if ((rc = ensure_scratch(ctx)) != 0) return bail(rc);
    mi355_lzss_batch *b = ctx->scratch;
    const uint64_t off2[2] = {0, aug.size()};
    if ((rc = upload_async(b, aug.data(), off2, 1, nullptr, LZSS_MAX_DICT_SIZE, sm)) != 0) return bail(rc);
    HIP_TRY((hipError_t)launch_sa_build(b->segs.as<Seg>(), b->h_segs.data(), 1, b->max_n, b->text.as<uint8_t>(), sa_work(b), sm));
    launch_lcp(b->segs.as<Seg>(), 1, b->max_n, batch_index(b), sm);
    if ((rc = ctx->d_text.ensure(stride)) != 0) return bail(rc);
    if ((rc = ctx->d_sa.ensure(stride * 4)) != 0) return bail(rc);
    if ((rc = ctx->d_isa.ensure(stride * 4)) != 0) return bail(rc);
    if ((rc = ctx->d_lcp.ensure(stride * 2)) != 0) return bail(rc);
    if ((rc = ctx->d_b1.ensure((stride / 64 + 1) * 2)) != 0) return bail(rc);
    if ((rc = ctx->d_b2.ensure((stride / 64 + 1) * 2)) != 0) return bail(rc);
#define CP(dst, src, bytes)                                                                              \
    if (hipMemcpyAsync(ctx->dst.p, b->src.p, (bytes), hipMemcpyDeviceToDevice, sm) != hipSuccess)         \
        return bail(fail(MI355_LZSS_E_DEVICE, "device copy of the dictionary index failed"));
    CP(d_text, text, stride) CP(d_sa, sa, stride * 4) CP(d_isa, isa, stride * 4) CP(d_lcp, lcp, stride * 2)
    CP(d_b1, b1, (stride / 64 + 1) * 2) CP(d_b2, b2, (stride / 64 + 1) * 2)
#undef CP
    if ((rc = ctx->d_key.ensure(stride * 8)) != 0) return bail(rc);
    if ((rc = ctx->d_bkt.ensure(DICT_BUCKETS * 8)) != 0) return bail(rc);
    if (hipMemsetAsync(ctx->d_bkt.p, 0, DICT_BUCKETS * 8, sm) != hipSuccess)
        return bail(fail(MI355_LZSS_E_DEVICE, "hipMemsetAsync failed"));
    launch_dict_aux(ctx->d_text.as<uint8_t>(), (uint32_t)aug.size(), ctx->d_sa.as<int32_t>(), ctx->d_key.as<uint64_t>(),
                    ctx->d_bkt.as<int32_t>(), sm);
    if (hipStreamSynchronize(sm) != hipSuccess || hipGetLastError() != hipSuccess)
        return bail(fail(MI355_LZSS_E_DEVICE, "building the dictionary index failed: %s", hipGetErrorString(hipGetLastError())));
    ctx->view.T = ctx->d_text.as<uint8_t>();
    ctx->view.n = (uint32_t)aug.size();
    ctx->view.SA = ctx->d_sa.as<int32_t>();
    ctx->view.LCP = ctx->d_lcp.as<uint16_t>();
    ctx->view.bmin1 = ctx->d_b1.as<uint16_t>();
    ctx->view.bmin2 = ctx->d_b2.as<uint16_t>();
    ctx->view.key = ctx->d_key.as<uint64_t>();
    ctx->view.bkt = ctx->d_bkt.as<int2>();
    if ((rc = noalloc_bind(ctx)) != 0) return bail(rc);
    *out = ctx;
    return 0;
}

extern "C" int mi355_lzss_ctx_create(const uint8_t *dict, size_t dict_len, int device, mi355_lzss_ctx **out) {
    return ctx_create(dict, dict_len, device, false, out);
}
extern "C" int mi355_lzss_ctx_create_decoder(const uint8_t *dict, size_t dict_len, int device, mi355_lzss_ctx **out) {
    return ctx_create(dict, dict_len, device, true, out);
}

extern "C" int mi355_lzss_ctx_set_overlap(mi355_lzss_ctx *ctx, int parts) {
    if (!ctx || parts < 1 || parts > mi355_lzss_ctx::kMaxSplit) return fail(MI355_LZSS_E_ARG, "overlap parts must be 1..%d", mi355_lzss_ctx::kMaxSplit);
    ctx->split = parts;
    return 0;
}

extern "C" int mi355_lzss_ctx_set_chunk_bytes(mi355_lzss_ctx *ctx, uint64_t bytes) {
    if (!ctx || bytes == 0) return fail(MI355_LZSS_E_ARG, "chunk size must be positive");
    ctx->chunk_bytes = bytes;
    return 0;
}

static void pipeline_destroy(Pipeline *p);

extern "C" void mi355_lzss_ctx_destroy(mi355_lzss_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    pipeline_destroy(ctx->pipe);
    for (mi355_lzss_batch *b : ctx->pool_all) mi355_lzss_batch_destroy(b);
    if (ctx->scratch) mi355_lzss_batch_destroy(ctx->scratch);
    delete ctx->na;  // its buffers are views of the static scratch array: nothing to free
    DevBuf *bufs[] = {&ctx->d_text, &ctx->d_sa, &ctx->d_isa, &ctx->d_lcp, &ctx->d_b1, &ctx->d_b2, &ctx->d_key, &ctx->d_bkt};
    for (DevBuf *d : bufs) d->release();
    for (int k = 0; k < mi355_lzss_ctx::kMaxSplit; ++k) {
        if (ctx->side[k]) (void)hipStreamDestroy(ctx->side[k]);
        if (ctx->join_ev[k]) (void)hipEventDestroy(ctx->join_ev[k]);
        if (ctx->aux[k]) (void)hipStreamDestroy(ctx->aux[k]);
        if (ctx->sa_ev[k]) (void)hipEventDestroy(ctx->sa_ev[k]);
        if (ctx->head_ev[k]) (void)hipEventDestroy(ctx->head_ev[k]);
        if (ctx->win_ev[k]) (void)hipEventDestroy(ctx->win_ev[k]);
    }
    if (ctx->fork_ev) (void)hipEventDestroy(ctx->fork_ev);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int mi355_lzss_dict_info(const mi355_lzss_ctx *ctx, uint64_t *aug_len, int32_t reserved_idx[2]) {
    if (!ctx) return fail(MI355_LZSS_E_ARG, "null argument");
    if (aug_len) *aug_len = ctx->dict.size();
    if (reserved_idx) {
        reserved_idx[0] = ctx->res_idx[0];
        reserved_idx[1] = ctx->res_idx[1];
    }
    return 0;
}

// ---- single-buffer entry points -----------------------------------------------------------------------
static int upload_one(mi355_lzss_batch *b, const uint8_t *d, size_t n, size_t start) {
    if (n && !d) return fail(MI355_LZSS_E_ARG, "null argument");
    if (n > LZSS_MAX_INPUT_SIZE) return fail(MI355_LZSS_E_INPUT_SIZE, "input size must be <= %d", LZSS_MAX_INPUT_SIZE);
    if (start > n) return fail(MI355_LZSS_E_ARG, "start beyond the end of the buffer");
    const uint64_t off2[2] = {0, n};
    const uint32_t st = (uint32_t)start;
    return upload_async(b, d ? d : (const uint8_t *)"", off2, 1, &st, LZSS_MAX_INPUT_SIZE, b->stream);
}

static int compress_ctx(mi355_lzss_ctx *ctx) {  // a ctx that can compress, made current, with its scratch batch
    if (!ctx) return fail(MI355_LZSS_E_ARG, "null argument");
    if (ctx->decode_only) return fail(MI355_LZSS_E_ARG, "the context was created for decompression only");
    int rc;
    if ((rc = set_device(ctx)) != 0) return rc;
    return ensure_scratch(ctx);
}

extern "C" int mi355_lzss_best_backrefs(mi355_lzss_ctx *ctx, const uint8_t *d, size_t n, size_t start, int32_t *addr,
                                        int32_t *len, uint8_t *type) {
    int rc;
    if ((rc = compress_ctx(ctx)) != 0) return rc;
    mi355_lzss_batch *b = ctx->scratch;
    if ((rc = upload_one(b, d, n, start)) != 0) return rc;
    if ((rc = run_index_and_match(b, false)) != 0) return rc;
    std::vector<uint32_t> packed(n ? n : 1);
    HIP_TRY(hipStreamSynchronize(b->stream));
    if ((rc = mi355_lzss_batch_results(b, 0, packed.data(), n)) != 0) return rc;
    const int dict_len = (int)ctx->dict.size();
    for (size_t i = start; i < n; ++i) {
        const uint32_t r = packed[i];
        const size_t o = i - start;
        const uint32_t t = res_type(r);
        if (t == RES_NONE) {
            if (type) type[o] = 0;
            if (addr) addr[o] = -1;
            if (len) len[o] = -1;
            continue;
        }
        const int field = (int)res_field(r);
        int a;
        if (t == RES_SHORT) a = (int)i - field - 1;
        else a = (int)i + dict_len - field - 1;  // backref.go:66 inverted (dictionary-prefixed address)
        if (type) type[o] = t == RES_SHORT ? LZSS_SYMBOL_SHORT : LZSS_SYMBOL_DYNAMIC;
        if (addr) addr[o] = a;
        if (len) len[o] = (int)res_len(r);
    }
    return 0;
}

extern "C" int mi355_lzss_write(mi355_lzss_ctx *ctx, const uint8_t *inbuf, size_t n, size_t start, unsigned bit_off,
                                uint8_t *out, size_t out_cap, uint64_t *nbits) {
    if (bit_off > 7) return fail(MI355_LZSS_E_ARG, "bit_off must be 0..7");
    int rc;
    if ((rc = compress_ctx(ctx)) != 0) return rc;
    mi355_lzss_batch *b = ctx->scratch;
    if ((rc = upload_one(b, inbuf, n, start)) != 0) return rc;
    if ((rc = run_index_and_match(b, true)) != 0) return rc;
    if ((rc = run_walk_emit(b, 0, bit_off, true)) != 0) return rc;
    if (nbits) *nbits = b->h_nbits[0];
    const size_t need = b->h_out_len[0];
    if (need > out_cap) return fail(MI355_LZSS_E_CAPACITY, "output needs %zu bytes, capacity is %zu", need, out_cap);
    if (b->h_status[0] != 0) return fail(b->h_status[0], "emit failed with status %d", b->h_status[0]);
    if (need) HIP_TRY(hipMemcpy(out, b->out.as<uint8_t>() + b->h_out_off[0], need, hipMemcpyDeviceToHost));
    return 0;
}

// ---- CompressedSize256k: stateless and thread-safe (compress.go:381-383) -----------------------------------
// Every call takes a workspace of its own (own stream, own buffers) from the ctx's pool, so calls from several
// threads share nothing but the read-only dictionary index.
static int pool_acquire(mi355_lzss_ctx *ctx, mi355_lzss_batch **out) {
    {
        std::lock_guard<std::mutex> lk(ctx->pool_mu);
        if (!ctx->pool_free.empty()) {
            *out = ctx->pool_free.back();
            ctx->pool_free.pop_back();
            return 0;
        }
    }
    mi355_lzss_batch *b = nullptr;
    int rc;
    if ((rc = batch_new(ctx, 1, 0, true, &b)) != 0) return rc;
    std::lock_guard<std::mutex> lk(ctx->pool_mu);
    ctx->pool_all.push_back(b);
    *out = b;
    return 0;
}
// a pooled workspace keeps what it grew to (~40 B per element); one that a single large call blew up beyond this many
// elements (~13 GB) is given back to the device when it returns to the pool instead of staying until ctx_destroy
constexpr uint64_t POOL_KEEP_ELEMS = 320ull << 20;
static void pool_release(mi355_lzss_ctx *ctx, mi355_lzss_batch *b) {
    if (b->cap_elems > POOL_KEEP_ELEMS) {
        {
            std::lock_guard<std::mutex> lk(ctx->pool_mu);
            for (size_t k = 0; k < ctx->pool_all.size(); ++k)
                if (ctx->pool_all[k] == b) {
                    ctx->pool_all.erase(ctx->pool_all.begin() + (long)k);
                    break;
                }
        }
        mi355_lzss_batch_destroy(b);
        return;
    }
    std::lock_guard<std::mutex> lk(ctx->pool_mu);
    ctx->pool_free.push_back(b);
}

extern "C" int mi355_lzss_compressed_size_batch(mi355_lzss_ctx *ctx, const uint8_t *in, const uint64_t *in_off, uint32_t n_blobs,
                                                uint64_t *sizes, int32_t *status) {
    if (!ctx || !in_off || (n_blobs && !in) || !sizes) return fail(MI355_LZSS_E_ARG, "null argument");
    if (ctx->decode_only) return fail(MI355_LZSS_E_ARG, "the context was created for decompression only");
    int rc;
    if ((rc = set_device(ctx)) != 0) return rc;
    mi355_lzss_batch *b = nullptr;
    if ((rc = pool_acquire(ctx, &b)) != 0) return rc;
    auto run = [&]() -> int {
        int r;
        if ((r = upload_async(b, in, in_off, n_blobs, nullptr, LZSS_SIZE256K_MAX, b->stream)) != 0) return r;
        if ((r = run_index_and_match(b, true)) != 0) return r;
        if ((r = run_walk_emit(b, LZSS_HEADER_SIZE, 0, false)) != 0) return r;
        int worst = 0;
        for (uint32_t k = 0; k < n_blobs; ++k) {
            const int32_t st = b->h_pre_status[k];
            sizes[k] = st ? 0 : b->h_out_len[k];
            if (status) status[k] = st;
            if (st && !worst) worst = st;
        }
        if (worst && !status) return fail(worst, "input size must be <= %d", LZSS_SIZE256K_MAX);
        return 0;
    };
    rc = run();
    pool_release(ctx, b);
    return rc;
}

extern "C" int mi355_lzss_compressed_size(mi355_lzss_ctx *ctx, const uint8_t *d, size_t n, uint64_t *size) {
    if (n > LZSS_SIZE256K_MAX) return fail(MI355_LZSS_E_INPUT_SIZE, "input size must be <= %d", LZSS_SIZE256K_MAX);
    if (n && !d) return fail(MI355_LZSS_E_ARG, "null argument");
    const uint64_t off2[2] = {0, n};
    uint64_t sz = 0;
    const int rc = mi355_lzss_compressed_size_batch(ctx, d ? d : (const uint8_t *)"", off2, 1, &sz, nullptr);
    if (rc == 0 && size) *size = sz;
    return rc;
}

// ---- one size-capped blob from a stream, its candidate Writes as batches -------------------------
// bits [0, nbits) of src (most significant bit first) appended to dst at bit position pos
static void append_bits(std::vector<uint8_t> &dst, uint64_t &pos, const uint8_t *src, uint64_t nbits) {
    if (nbits == 0) return;
    dst.resize((size_t)((pos + nbits + 7) / 8), 0);
    const unsigned sh = (unsigned)(pos & 7);
    const uint64_t byte = pos >> 3, nbytes = (nbits + 7) / 8;
    for (uint64_t k = 0; k < nbytes; ++k) {
        uint8_t v = src[k];
        if (k == nbytes - 1 && (nbits & 7)) v &= (uint8_t)(0xFF00u >> (nbits & 7));
        dst[(size_t)(byte + k)] |= (uint8_t)(v >> sh);
        if (sh && byte + k + 1 < dst.size()) dst[(size_t)(byte + k + 1)] |= (uint8_t)(v << (8 - sh));
    }
    pos += nbits;
}

constexpr uint64_t PACK_BATCH_BYTES = 256ull << 20;  // candidate prefixes of one mi355_lzss_pack_blob batch (~10 GB of workspace)

extern "C" int mi355_lzss_pack_blob(mi355_lzss_ctx *ctx, const uint8_t *data, uint64_t n, uint32_t chunk, uint64_t out_max,
                                    uint32_t lookahead, uint8_t *out, uint64_t out_cap, uint64_t *out_len, uint64_t *chunks,
                                    int *needs_loop) {
    if (!ctx || (n && !data) || !out_len || !chunks || !needs_loop || (out_cap && !out))
        return fail(MI355_LZSS_E_ARG, "null argument");
    if (chunk == 0 || lookahead == 0 || out_max < LZSS_HEADER_SIZE) return fail(MI355_LZSS_E_ARG, "chunk, lookahead and out_max must be positive");
    if (ctx->decode_only) return fail(MI355_LZSS_E_ARG, "the context was created for decompression only");
    *out_len = 0;
    *chunks = 0;
    *needs_loop = 0;
    int rc;
    if ((rc = set_device(ctx)) != 0) return rc;
    mi355_lzss_batch *b = nullptr;
    if ((rc = pool_acquire(ctx, &b)) != 0) return rc;
    std::vector<uint8_t> acc;  // the blob's phrase bits so far
    uint64_t pos = 0;          // bits in acc
    uint64_t total = 0;        // accepted phrase bits (pos, but for the pad bits a Revert leaves behind)
    uint64_t j = 0;            // accepted chunks
    auto finish = [&](uint64_t k) -> int {
        const uint64_t len = LZSS_HEADER_SIZE + (pos + 7) / 8;
        *out_len = len;
        *chunks = k;
        if (len > out_cap) return fail(MI355_LZSS_E_CAPACITY, "the blob needs %llu bytes", (unsigned long long)len);
        out[0] = (uint8_t)(LZSS_VERSION >> 8);
        out[1] = (uint8_t)(LZSS_VERSION & 0xFF);
        out[2] = 0;
        if (len > LZSS_HEADER_SIZE) memcpy(out + LZSS_HEADER_SIZE, acc.data(), (size_t)(len - LZSS_HEADER_SIZE));
        return 0;
    };
    auto run = [&]() -> int {
        std::vector<uint8_t> streams;
        std::vector<uint64_t> off, plen, soff, nbits;
        std::vector<uint32_t> starts, lens;
        std::vector<int32_t> status;
        for (;;) {
            // candidate Writes j + 1 .. j + look (the loop starts a Write only while bytes are left).  After the first
            // batch the blob's own bits per chunk predict where the cap falls: the next batch reaches just past it
            uint64_t look = lookahead;
            if (j && total) {
                const double left = (double)(((int64_t)out_max - LZSS_HEADER_SIZE) * 8 - (int64_t)total);
                const int64_t est = (int64_t)(left / ((double)total / (double)j) * 1.04) + 3;
                look = (uint64_t)std::max<int64_t>(4, std::min<int64_t>(4 * (int64_t)lookahead, est));
            }
            off.clear();
            plen.clear();
            starts.clear();
            // a candidate is a whole prefix of the stream's rest and costs ~40 B of workspace per position: the batch is
            // bounded by the SUM of the prefix lengths too (small chunks with a large out_max would otherwise reserve
            // tens of GB), at least one candidate per batch
            uint64_t sum_len = 0;
            for (uint64_t t = 0; t < look; ++t) {
                const uint64_t s = (j + t) * (uint64_t)chunk;
                if (s >= n) break;
                const uint64_t e = std::min<uint64_t>(s + chunk, n);
                if (e > LZSS_MAX_INPUT_SIZE) break;  // Write would fail there (compress.go:473); the loop ends before it
                if (t > 0 && sum_len + e > PACK_BATCH_BYTES) break;
                sum_len += e;
                off.push_back(0);  // every candidate's inBuf is a prefix of the stream's rest: the bytes go up once
                plen.push_back(e);
                starts.push_back((uint32_t)s);
            }
            const uint32_t cnt = (uint32_t)starts.size();
            if (cnt == 0) return finish(j);
            int r;
            if ((r = upload_async(b, data, off.data(), cnt, starts.data(), LZSS_MAX_INPUT_SIZE, b->stream, plen.data())) != 0) return r;
            if ((r = mi355_lzss_batch_compress(b, 0)) != 0) return r;
            lens.assign(cnt, 0);
            nbits.assign(cnt, 0);
            uint64_t tot = 0;
            if ((r = mi355_lzss_batch_sizes(b, lens.data(), nbits.data(), &tot)) != 0) return r;
            bool have = false;
            for (uint32_t t = 0; t < cnt; ++t) {
                const uint64_t new_total = total + nbits[t];
                const uint64_t new_len = LZSS_HEADER_SIZE + (new_total + 7) / 8;
                const uint64_t uncompressed = (j + t + 1) * (uint64_t)chunk + 3;
                if (new_len >= out_max && uncompressed <= out_max && new_len > uncompressed) {
                    *needs_loop = 1;  // the loop would call ConsiderBypassing here (compress_test.go:343-346)
                    return 0;
                }
                if (!have) {
                    soff.assign(cnt + 1, 0);
                    for (uint32_t k = 0; k < cnt; ++k) soff[k + 1] = soff[k] + lens[k];
                    streams.assign((size_t)soff[cnt] + 16, 0);
                    status.assign(cnt, 0);
                    if ((r = mi355_lzss_batch_download(b, streams.data(), soff.data(), status.data())) != 0) return r;
                    for (uint32_t k = 0; k < cnt; ++k)
                        if (status[k] != 0) return fail(status[k], "candidate write %u failed", k);
                    have = true;
                }
                const uint8_t *st = streams.data() + soff[t];
                if (new_len > out_max) {
                    // Revert: the blob ends before this chunk.  outBuf.Truncate(lastOutLen) (compress.go:341) keeps the
                    // re-opened last byte as the reverted Write left it, so the pad bits of the blob are the first
                    // bits of the reverted chunk's phrases, not zeros
                    const uint64_t pad = std::min<uint64_t>((8 - (total & 7)) & 7, nbits[t]);
                    append_bits(acc, pos, st, pad);
                    return finish(j + t);
                }
                append_bits(acc, pos, st, nbits[t]);
                total = new_total;
                if (new_len >= out_max) return finish(j + t + 1);
            }
            j += cnt;
        }
    };
    rc = run();
    pool_release(ctx, b);
    return rc;
}

// ---- host-buffer batches: chunks in a copy-in / compute / copy-out pipeline ---------------------------------------
// Chunk k's bytes go up on the copy-in stream while chunk k-1 computes and chunk k-2's streams come down: two
// workspaces, three streams, no host synchronisation inside a chunk's kernel sequence.  The caller's buffers are used
// as they are: pinned ones (mi355_lzss_host_alloc) make both copies true DMA.
struct Pipeline {
    mi355_lzss_batch *slot[2] = {nullptr, nullptr};
    hipStream_t s_in = nullptr, s_out = nullptr;
    hipEvent_t ev_up[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
    PinBuf h_out[2];
};

static void pipeline_destroy(Pipeline *p) {
    if (!p) return;
    for (int k = 0; k < 2; ++k) {
        if (p->slot[k]) mi355_lzss_batch_destroy(p->slot[k]);
        if (p->ev_up[k]) (void)hipEventDestroy(p->ev_up[k]);
        if (p->ev_done[k]) (void)hipEventDestroy(p->ev_done[k]);
        p->h_out[k].release();
    }
    if (p->s_in) (void)hipStreamDestroy(p->s_in);
    if (p->s_out) (void)hipStreamDestroy(p->s_out);
    delete p;
}

static int pipeline_get(mi355_lzss_ctx *ctx, Pipeline **out) {
    if (!ctx->pipe) {
        Pipeline *p = new Pipeline();
        ctx->pipe = p;  // owned by the ctx from here on (a partial one is destroyed with it)
        HIP_TRY(hipStreamCreateWithFlags(&p->s_in, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&p->s_out, hipStreamNonBlocking));
        for (int k = 0; k < 2; ++k) {
            int rc;
            if ((rc = batch_new(ctx, 1, 0, false, &p->slot[k])) != 0) return rc;
            HIP_TRY(hipEventCreateWithFlags(&p->ev_up[k], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&p->ev_done[k], hipEventDisableTiming));
        }
    }
    if (!ctx->pipe->slot[1] || !ctx->pipe->ev_done[1]) return fail(MI355_LZSS_E_DEVICE, "the transfer pipeline could not be set up");
    *out = ctx->pipe;
    return 0;
}

// consecutive blobs [first, last) whose bytes stay within `budget` (at least one blob)
// slots: workgroups of the suffix sort the device holds at once (two per CU, one per blob): a chunk that is not the
// batch's last takes a whole number of such rounds (1250 blobs on 512 slots ran 2.44 rounds, the third 44 % full:
// transfer-inclusive 0.88 x the resident rate with chunks of 1250 blobs, 0.93 x with 1536)
static uint32_t chunk_end(const uint64_t *off, uint32_t first, uint32_t n_blobs, uint64_t budget, uint32_t slots) {
    uint32_t last = first + 1;
    while (last < n_blobs && off[last + 1] - off[first] <= budget && last - first < (1u << 20)) ++last;
    if (last < n_blobs && slots && last - first > slots) last = first + (last - first) / slots * slots;
    return last;
}

static int compress_pipelined(mi355_lzss_ctx *ctx, const uint8_t *in, const uint64_t *in_off, uint32_t n_blobs, uint8_t *out,
                              const uint64_t *out_off, uint32_t *out_len, int32_t *status) {
    int rc;
    if ((rc = set_device(ctx)) != 0) return rc;
    Pipeline *p = nullptr;
    if ((rc = pipeline_get(ctx, &p)) != 0) return rc;
    struct Chunk { uint32_t first, last; };
    std::vector<Chunk> chunks;
    {
        // The first chunk's upload and the last chunk's download overlap no kernels: on a long batch both chunks are one
        // round of the suffix sort's resident workgroups only (150.4 -> 144.0 ms per 10 000 blobs with chunks of 2560)
        const uint32_t slots = ctx->wg_slots;
        uint32_t f = 0, tail_first = n_blobs;
        if (PIPE_GRADED && slots && n_blobs >= 6 * slots && in_off[slots] - in_off[0] <= ctx->chunk_bytes &&
            in_off[n_blobs] - in_off[n_blobs - slots] <= ctx->chunk_bytes) {
            chunks.push_back(Chunk{0, slots});
            f = slots;
            tail_first = n_blobs - slots;
        }
        while (f < tail_first) {
            const uint32_t l = chunk_end(in_off, f, tail_first, ctx->chunk_bytes, slots);
            chunks.push_back(Chunk{f, l});
            f = l;
        }
        if (tail_first < n_blobs) chunks.push_back(Chunk{tail_first, n_blobs});
    }
    int worst = 0;
    // chunk k's streams come down (and go to the caller's layout) while chunk k+1 computes
    auto finish = [&](size_t k) -> int {
        const int s = (int)(k & 1);
        mi355_lzss_batch *b = p->slot[s];
        const Chunk &c = chunks[k];
        int r;
        HIP_TRY(hipEventSynchronize(p->ev_done[s]));
        if ((r = finish_walk_emit(b, true)) != 0) return r;
        const uint64_t total = b->n_blobs ? b->h_out_off[b->n_blobs] : 0;
        if (total) {
            if ((r = p->h_out[s].ensure((size_t)total)) != 0) return r;
            HIP_TRY(hipMemcpyAsync(p->h_out[s].p, b->out.p, (size_t)total, hipMemcpyDeviceToHost, p->s_out));
            HIP_TRY(hipStreamSynchronize(p->s_out));
        }
        const int w = scatter_outputs(b, p->h_out[s].as<uint8_t>(), out, out_off + c.first, out_len ? out_len + c.first : nullptr,
                                      status ? status + c.first : nullptr);
        if (w && !worst) worst = w;
        return 0;
    };
    auto run = [&]() -> int {
        int r;
        for (size_t k = 0; k < chunks.size(); ++k) {
            const int s = (int)(k & 1);
            mi355_lzss_batch *b = p->slot[s];
            const Chunk &c = chunks[k];
            // slot s is free: chunk k-2 was finished (synchronised) in the previous iteration
            if ((r = upload_async(b, in, in_off + c.first, c.last - c.first, nullptr, LZSS_MAX_INPUT_SIZE, p->s_in)) != 0) return r;
            HIP_TRY(hipEventRecord(p->ev_up[s], p->s_in));
            HIP_TRY(hipStreamWaitEvent(ctx->stream, p->ev_up[s], 0));
            if ((r = run_index_and_match(b, true)) != 0) return r;
            if ((r = enqueue_walk_emit(b, LZSS_HEADER_SIZE, 0, true)) != 0) return r;
            HIP_TRY(hipEventRecord(p->ev_done[s], ctx->stream));
            if (k >= 1 && (r = finish(k - 1)) != 0) return r;
        }
        if (!chunks.empty() && (r = finish(chunks.size() - 1)) != 0) return r;
        return 0;
    };
    rc = run();
    if (rc != 0) {
        // an error exit leaves copies that read the caller's `in`, kernels of one or two chunks and possibly a download
        // in flight: drain all three streams (and the side streams the kernels fork onto, which join ctx->stream) before
        // the caller may release its buffers or the next call reuses the two slots.  The first error is the one reported.
        (void)hipStreamSynchronize(p->s_in);
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipStreamSynchronize(p->s_out);
        (void)hipDeviceSynchronize();
        return rc;
    }
    if (worst && !status) return fail(worst, "blob failed with status %d", worst);
    return 0;
}

extern "C" int mi355_lzss_compress_batch(mi355_lzss_ctx *ctx, const uint8_t *in, const uint64_t *in_off, uint32_t n_blobs,
                                         uint8_t *out, const uint64_t *out_off, uint32_t *out_len, int32_t *status) {
    if (!ctx || !in_off || !out_off || (n_blobs && (!in || !out))) return fail(MI355_LZSS_E_ARG, "null argument");
    if (ctx->decode_only) return fail(MI355_LZSS_E_ARG, "the context was created for decompression only");
    return compress_pipelined(ctx, in, in_off, n_blobs, out, out_off, out_len, status);
}

// ---- one call for several devices: contiguous blob ranges balanced by bytes, one host thread per context ----------
extern "C" int mi355_lzss_partition(const uint64_t *off, uint32_t n_blobs, int n_parts, uint32_t *first) {
    if (!off || !first || n_parts < 1) return fail(MI355_LZSS_E_ARG, "bad argument");
    const uint64_t base = off[0], total = off[n_blobs] - base;
    uint32_t b = 0;
    first[0] = 0;
    for (int t = 1; t < n_parts; ++t) {
        // part t starts at the first blob whose start offset reaches t / n_parts of the bytes; never before part t-1
        const uint64_t target = (uint64_t)((unsigned __int128)total * (unsigned)t / (unsigned)n_parts);
        while (b < n_blobs && off[b] - base < target) ++b;
        first[t] = b;
    }
    first[n_parts] = n_blobs;
    return 0;
}

typedef int (*batch_fn)(mi355_lzss_ctx *, const uint8_t *, const uint64_t *, uint32_t, uint8_t *, const uint64_t *, uint32_t *, int32_t *);

static int run_multi(batch_fn fn, mi355_lzss_ctx *const *ctxs, int n_ctx, const uint8_t *in, const uint64_t *in_off, uint32_t n_blobs,
                     uint8_t *out, const uint64_t *out_off, uint32_t *out_len, int32_t *status) {
    if (!ctxs || n_ctx < 1 || !in_off || !out_off) return fail(MI355_LZSS_E_ARG, "null argument");
    for (int t = 0; t < n_ctx; ++t)
        if (!ctxs[t]) return fail(MI355_LZSS_E_ARG, "null context");
    std::vector<uint32_t> first((size_t)n_ctx + 1);
    int rc;
    if ((rc = mi355_lzss_partition(in_off, n_blobs, n_ctx, first.data())) != 0) return rc;
    std::vector<int> rcs((size_t)n_ctx, 0);
    std::vector<std::string> errs((size_t)n_ctx);
    auto work = [&](int t) {
        const uint32_t f = first[(size_t)t], cnt = first[(size_t)t + 1] - f;
        if (cnt == 0) return;
        rcs[(size_t)t] = fn(ctxs[t], in, in_off + f, cnt, out, out_off + f, out_len ? out_len + f : nullptr, status ? status + f : nullptr);
        if (rcs[(size_t)t] != 0) errs[(size_t)t] = g_err;  // the error text is per thread
    };
    std::vector<std::thread> th;
    for (int t = 1; t < n_ctx; ++t) th.emplace_back(work, t);
    work(0);
    for (std::thread &x : th) x.join();
    for (int t = 0; t < n_ctx; ++t)
        if (rcs[(size_t)t] != 0) return fail(rcs[(size_t)t], "context %d: %s", t, errs[(size_t)t].c_str());
    return 0;
}

extern "C" int mi355_lzss_compress_batch_multi(mi355_lzss_ctx *const *ctxs, int n_ctx, const uint8_t *in, const uint64_t *in_off,
                                               uint32_t n_blobs, uint8_t *out, const uint64_t *out_off, uint32_t *out_len,
                                               int32_t *status) {
    return run_multi(mi355_lzss_compress_batch, ctxs, n_ctx, in, in_off, n_blobs, out, out_off, out_len, status);
}

// ---- decompression: see decompress.hip ----------------------------------------------------------------------
extern "C" int mi355_lzss_batch_upload_compressed(mi355_lzss_batch *b, const uint8_t *in, const uint64_t *in_off,
                                                  uint32_t n_blobs, const uint64_t *out_cap_off) {
    if (!b || !in_off || !out_cap_off || (n_blobs && !in)) return fail(MI355_LZSS_E_ARG, "null argument");
    int rc;
    if ((rc = set_device(b->ctx)) != 0) return rc;
    for (uint32_t k = 0; k < n_blobs; ++k)
        if (in_off[k + 1] < in_off[k] || out_cap_off[k + 1] < out_cap_off[k])
            return fail(MI355_LZSS_E_ARG, "offsets must be non-decreasing");
    // rebase both offset arrays to 0
    b->h_cin_off.assign(n_blobs + 1, 0);
    b->h_dout_off.assign(n_blobs + 1, 0);
    for (uint32_t k = 0; k <= n_blobs; ++k) {
        b->h_cin_off[k] = in_off[k] - in_off[0];
        b->h_dout_off[k] = out_cap_off[k] - out_cap_off[0];
    }
    const uint64_t cin = b->h_cin_off[n_blobs], dout = b->h_dout_off[n_blobs];
    if ((rc = b->d_cin.ensure((size_t)cin + 16)) != 0) return rc;
    if ((rc = b->d_dout.ensure((size_t)std::max<uint64_t>(dout, 16))) != 0) return rc;
    if ((rc = b->d_cin_off.ensure((n_blobs + 1) * 8)) != 0) return rc;
    if ((rc = b->d_dout_off.ensure((n_blobs + 1) * 8)) != 0) return rc;
    if ((rc = b->d_dlen.ensure(std::max<uint32_t>(n_blobs, 1) * 4)) != 0) return rc;
    if ((rc = b->d_dstatus.ensure(std::max<uint32_t>(n_blobs, 1) * 4)) != 0) return rc;
    hipStream_t sm = b->stream;
    HIP_TRY(hipMemsetAsync((uint8_t *)b->d_cin.p + cin, 0, 16, sm));
    if (cin) HIP_TRY(hipMemcpyAsync(b->d_cin.p, in + in_off[0], (size_t)cin, hipMemcpyHostToDevice, sm));
    HIP_TRY(hipMemcpyAsync(b->d_cin_off.p, b->h_cin_off.data(), (n_blobs + 1) * 8, hipMemcpyHostToDevice, sm));
    HIP_TRY(hipMemcpyAsync(b->d_dout_off.p, b->h_dout_off.data(), (n_blobs + 1) * 8, hipMemcpyHostToDevice, sm));
    HIP_TRY(hipStreamSynchronize(sm));
    b->n_dblobs = n_blobs;
    return 0;
}

extern "C" int mi355_lzss_batch_decompress(mi355_lzss_batch *b) {
    if (!b) return fail(MI355_LZSS_E_ARG, "null argument");
    int rc;
    mi355_lzss_ctx *ctx = b->ctx;
    if ((rc = set_device(ctx)) != 0) return rc;
    if ((rc = mark(b, ST_DECOMP)) != 0) return rc;
    launch_decompress(b->d_cin.as<uint8_t>(), b->d_cin_off.as<uint64_t>(), b->n_dblobs, b->d_dout.as<uint8_t>(),
                      b->d_dout_off.as<uint64_t>(), ctx->view.T, ctx->view.n, b->d_dlen.as<uint32_t>(),
                      b->d_dstatus.as<int32_t>(), b->stream);
    if ((rc = mark(b, ST_DECOMP + 1)) != 0) return rc;
    HIP_TRY(hipStreamSynchronize(b->stream));
    HIP_TRY(hipGetLastError());
    return collect_stage_times(b, ST_DECOMP, ST_DECOMP + 1);
}

// The device output is laid out like the caller's (blob k at out_off[k] - out_off[0], capacity up to out_off[k+1]),
// so when the capacities are mostly used ONE copy moves everything straight into the caller's buffer (the bytes of
// a blob's region beyond out_len[k] are unspecified then); sparse layouts are copied blob by blob.
extern "C" int mi355_lzss_batch_download_decompressed(mi355_lzss_batch *b, uint8_t *out, const uint64_t *out_off,
                                                      uint32_t *out_len, int32_t *status) {
    if (!b || !out_off || (b->n_dblobs && !out)) return fail(MI355_LZSS_E_ARG, "null argument");
    int rc;
    if ((rc = set_device(b->ctx)) != 0) return rc;
    const uint32_t nb = b->n_dblobs;
    if (nb == 0) return 0;
    std::vector<uint32_t> lens(nb);
    std::vector<int32_t> sts(nb);
    hipStream_t sm = b->stream;
    HIP_TRY(hipMemcpyAsync(lens.data(), b->d_dlen.p, nb * 4, hipMemcpyDeviceToHost, sm));
    HIP_TRY(hipMemcpyAsync(sts.data(), b->d_dstatus.p, nb * 4, hipMemcpyDeviceToHost, sm));
    HIP_TRY(hipStreamSynchronize(sm));
    const uint64_t total = b->h_dout_off[nb];
    uint64_t used = 0;
    bool same_layout = true;
    for (uint32_t k = 0; k < nb; ++k) {
        const uint64_t cap = out_off[k + 1] - out_off[k];
        if (sts[k] == 0 && lens[k] > cap) sts[k] = MI355_LZSS_E_CAPACITY;
        if (sts[k] == 0) used += lens[k];
        if (out_off[k] - out_off[0] != b->h_dout_off[k] || cap != b->h_dout_off[k + 1] - b->h_dout_off[k]) same_layout = false;
    }
    if (same_layout && total && used * 2 >= total) {
        HIP_TRY(hipMemcpyAsync(out + out_off[0], b->d_dout.p, (size_t)total, hipMemcpyDeviceToHost, sm));
    } else {
        for (uint32_t k = 0; k < nb; ++k)
            if (sts[k] == 0 && lens[k])
                HIP_TRY(hipMemcpyAsync(out + out_off[k], b->d_dout.as<uint8_t>() + b->h_dout_off[k], lens[k], hipMemcpyDeviceToHost, sm));
    }
    HIP_TRY(hipStreamSynchronize(sm));
    int worst = 0;
    for (uint32_t k = 0; k < nb; ++k) {
        if (out_len) out_len[k] = sts[k] == 0 ? lens[k] : 0;
        if (status) status[k] = sts[k];
        if (sts[k] && !worst) worst = sts[k];
    }
    if (worst && !status) return fail(worst, "blob failed with status %d", worst);
    return 0;
}

extern "C" int mi355_lzss_decompress_batch(mi355_lzss_ctx *ctx, const uint8_t *in, const uint64_t *in_off, uint32_t n_blobs,
                                           uint8_t *out, const uint64_t *out_off, uint32_t *out_len, int32_t *status) {
    if (!ctx || !in_off || !out_off) return fail(MI355_LZSS_E_ARG, "null argument");
    int rc;
    if ((rc = set_device(ctx)) != 0) return rc;
    if ((rc = ensure_scratch(ctx)) != 0) return rc;
    mi355_lzss_batch *b = ctx->scratch;
    if ((rc = mi355_lzss_batch_upload_compressed(b, in, in_off, n_blobs, out_off)) != 0) return rc;
    if ((rc = mi355_lzss_batch_decompress(b)) != 0) return rc;
    return mi355_lzss_batch_download_decompressed(b, out, out_off, out_len, status);
}

extern "C" int mi355_lzss_decompress_batch_multi(mi355_lzss_ctx *const *ctxs, int n_ctx, const uint8_t *in, const uint64_t *in_off,
                                                 uint32_t n_blobs, uint8_t *out, const uint64_t *out_off, uint32_t *out_len,
                                                 int32_t *status) {
    return run_multi(mi355_lzss_decompress_batch, ctxs, n_ctx, in, in_off, n_blobs, out, out_off, out_len, status);
}
