/* lzss_oracle.c - CPU restatement of the reference lzss compressor.  See lzss_oracle.h:
 * TEST INFRASTRUCTURE ONLY, never linked or loaded by the product path.
 *
 * Every function cites the reference file:line it follows (paths relative to the
 * reference repository root).  The suffix-array *construction* is free (the array is a
 * unique mathematical object); it is a textbook SA-IS written for this file.  The
 * *query* (LookupLongest) and everything above it are literal restatements.
 *
 * Third-party arithmetic not in the reference tree: github.com/icza/bitio v1.1.0
 * (go.mod:6).  Restated from its documented contract: MSB-first packing, WriteBits
 * keeps the n low bits of the value, Align zero-pads to a byte and returns the number
 * of pad bits, the reader latches the first error (EOF) and returns 0 afterwards.
 */
#include "lzss_oracle.h"
#include "../include/lzss_format.h"

#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------ */
/* Suffix array: textbook SA-IS over an int32 text with an implicit end sentinel.       */
/* ------------------------------------------------------------------------------------ */

static void sais_induce(const int32_t *T, int32_t *SA, int32_t n, int32_t K,
                        const uint8_t *isS, const int32_t *cnt, int32_t *bkt)
{
    int32_t i, j, sum;
    /* L-type: bucket heads, left to right */
    for (i = 0, sum = 0; i < K; i++) { bkt[i] = sum; sum += cnt[i]; }
    SA[bkt[T[n - 1]]++] = n - 1; /* induced by the sentinel suffix */
    for (i = 0; i < n; i++) {
        j = SA[i];
        if (j > 0 && !isS[j - 1]) SA[bkt[T[j - 1]]++] = j - 1;
    }
    /* S-type: bucket tails, right to left */
    for (i = 0, sum = 0; i < K; i++) { sum += cnt[i]; bkt[i] = sum; }
    for (i = n - 1; i >= 0; i--) {
        j = SA[i];
        if (j > 0 && isS[j - 1]) SA[--bkt[T[j - 1]]] = j - 1;
    }
}

static void sais_core(const int32_t *T, int32_t *SA, int32_t n, int32_t K)
{
    if (n <= 0) return;
    if (n == 1) { SA[0] = 0; return; }

    uint8_t *isS = (uint8_t *)malloc((size_t)n + 1);
    int32_t *cnt = (int32_t *)calloc((size_t)K, sizeof(int32_t));
    int32_t *bkt = (int32_t *)malloc((size_t)K * sizeof(int32_t));
    int32_t i, sum, m = 0;

    isS[n] = 1; /* sentinel */
    isS[n - 1] = 0;
    for (i = n - 2; i >= 0; i--)
        isS[i] = (uint8_t)(T[i] < T[i + 1] || (T[i] == T[i + 1] && isS[i + 1]));
#define IS_LMS(p) ((p) > 0 && isS[(p)] && !isS[(p) - 1])
    for (i = 0; i < n; i++) cnt[T[i]]++;

    /* stage 1: LMS positions at their bucket tails in any order, then induce */
    for (i = 0; i < n; i++) SA[i] = -1;
    for (i = 0, sum = 0; i < K; i++) { sum += cnt[i]; bkt[i] = sum; }
    for (i = 1; i < n; i++)
        if (IS_LMS(i)) { SA[--bkt[T[i]]] = i; m++; }
    sais_induce(T, SA, n, K, isS, cnt, bkt);

    if (m > 0) {
        /* LMS suffixes, now ordered by their LMS substrings, compacted to SA[0:m) */
        int32_t *lms_sorted = (int32_t *)malloc((size_t)m * sizeof(int32_t));
        int32_t *name_of = (int32_t *)malloc((size_t)n * sizeof(int32_t)); /* by text position */
        int32_t k = 0, names = 0, prev = -1;
        for (i = 0; i < n; i++)
            if (IS_LMS(SA[i])) lms_sorted[k++] = SA[i];
        for (i = 0; i < m; i++) {
            int32_t cur = lms_sorted[i];
            int diff = 1;
            if (prev >= 0) {
                int32_t d;
                diff = 0;
                for (d = 0;; d++) {
                    int ea = (prev + d == n), eb = (cur + d == n);
                    if (ea || eb) { diff = !(ea && eb); break; }
                    if (T[prev + d] != T[cur + d] || isS[prev + d] != isS[cur + d]) { diff = 1; break; }
                    if (d > 0) {
                        int la = IS_LMS(prev + d), lb = IS_LMS(cur + d);
                        if (la || lb) { diff = !(la && lb); break; }
                    }
                }
            }
            if (diff) names++;
            name_of[cur] = names - 1;
            prev = cur;
        }
        /* reduced text in position order */
        int32_t *lms_pos = (int32_t *)malloc((size_t)m * sizeof(int32_t));
        int32_t *T1 = (int32_t *)malloc((size_t)m * sizeof(int32_t));
        int32_t *SA1 = (int32_t *)malloc((size_t)m * sizeof(int32_t));
        for (i = 1, k = 0; i < n; i++)
            if (IS_LMS(i)) { lms_pos[k] = i; T1[k] = name_of[i]; k++; }
        if (names < m) {
            sais_core(T1, SA1, m, names);
        } else {
            for (i = 0; i < m; i++) SA1[T1[i]] = i;
        }
        /* stage 2: exactly sorted LMS suffixes at bucket tails, then induce */
        for (i = 0; i < n; i++) SA[i] = -1;
        for (i = 0, sum = 0; i < K; i++) { sum += cnt[i]; bkt[i] = sum; }
        for (i = m - 1; i >= 0; i--) {
            int32_t p = lms_pos[SA1[i]];
            SA[--bkt[T[p]]] = p;
        }
        sais_induce(T, SA, n, K, isS, cnt, bkt);
        free(lms_sorted); free(name_of); free(lms_pos); free(T1); free(SA1);
    }
#undef IS_LMS
    free(isS); free(cnt); free(bkt);
}

void ora_suffix_array(const uint8_t *data, int32_t n, int32_t *sa)
{
    if (n <= 0) return;
    int32_t *T = (int32_t *)malloc((size_t)n * sizeof(int32_t));
    for (int32_t i = 0; i < n; i++) T[i] = data[i];
    sais_core(T, sa, n, 256);
    free(T);
}

/* ------------------------------------------------------------------------------------ */
/* Index query: lzss/internal/suffixarray/suffixarray.go                                 */
/* ------------------------------------------------------------------------------------ */

typedef struct {
    const uint8_t *data;
    int32_t n;
    const int32_t *sa;
} ora_index;

/* bytes.Compare(x.at(i), s)  (suffixarray.go:53-55,125,145) */
static int idx_compare(const ora_index *x, int32_t i, const uint8_t *s, int32_t slen)
{
    int32_t off = x->sa[i], alen = x->n - off;
    int32_t m = alen < slen ? alen : slen;
    int c = memcmp(x->data + off, s, (size_t)m);
    if (c != 0) return c < 0 ? -1 : 1;
    if (alen < slen) return -1;
    if (alen > slen) return 1;
    return 0;
}

/* bytes.HasPrefix(x.at(i), s) */
static int idx_has_prefix(const ora_index *x, int32_t i, const uint8_t *s, int32_t slen)
{
    int32_t off = x->sa[i];
    if (x->n - off < slen) return 0;
    return memcmp(x->data + off, s, (size_t)slen) == 0;
}

/* suffixarray.go:144-152 */
static void idx_lookup_longest_initial(const ora_index *x, const uint8_t *s, int32_t slen,
                                       int32_t *r_start, int32_t *r_end)
{
    int32_t lo = 0, hi = x->n; /* sort.Search(len(x.sa), compare >= 0) */
    while (lo < hi) {
        int32_t h = (int32_t)(((uint32_t)lo + (uint32_t)hi) >> 1);
        if (!(idx_compare(x, h, s, slen) >= 0)) lo = h + 1; else hi = h;
    }
    int32_t i = lo;
    if (i == x->n || !idx_has_prefix(x, i, s, slen)) { *r_start = -1; *r_end = -1; return; }
    lo = 0; hi = x->n - i; /* sort.Search(len(x.sa)-i, !HasPrefix(at(k+i))) */
    while (lo < hi) {
        int32_t h = (int32_t)(((uint32_t)lo + (uint32_t)hi) >> 1);
        if (!(!idx_has_prefix(x, h + i, s, slen))) lo = h + 1; else hi = h;
    }
    *r_start = i;
    *r_end = i + lo;
}

/* suffixarray.go:121-142 */
static void idx_lookup_longest_step(const ora_index *x, const uint8_t *s, int32_t slen,
                                    int32_t range_start, int32_t range_end,
                                    int32_t s_start, int32_t s_end,
                                    int32_t *r_start, int32_t *offset_out)
{
    *r_start = s_start;
    int32_t lo = 0, hi = s_end - s_start;
    while (lo < hi) {
        int32_t h = (int32_t)(((uint32_t)lo + (uint32_t)hi) >> 1);
        if (!(idx_compare(x, h + s_start, s, slen) >= 0)) lo = h + 1; else hi = h;
    }
    int32_t i = lo + s_start;
    if (i == s_end || !idx_has_prefix(x, i, s, slen)) { *offset_out = -1; return; }
    *r_start = i;
    while (i < s_end && idx_has_prefix(x, i, s, slen)) {
        int32_t offset = x->sa[i];
        if (offset >= range_start && offset < range_end) { *offset_out = offset; return; }
        i++;
    }
    *offset_out = -1;
}

/* suffixarray.go:59-117 */
static void idx_lookup_longest(const ora_index *x, const uint8_t *s, int32_t min_end, int32_t max_end,
                               int32_t range_start, int32_t range_end,
                               int32_t *index, int32_t *length)
{
    *index = -1; *length = -1;
    int32_t s_start, s_end;
    idx_lookup_longest_initial(x, s, min_end, &s_start, &s_end);
    if (s_start == -1) return;
    if (s_start == s_end) {
        int32_t offset = x->sa[s_start];
        if (offset >= range_start && offset < range_end) { *index = offset; *length = min_end; }
        return;
    }
    for (int32_t i = s_start; i < s_end; i++) {
        int32_t offset = x->sa[i];
        if (offset >= range_start && offset < range_end) { *index = offset; *length = min_end; break; }
    }
    if (*length == -1) return;
    int32_t low = min_end, high = max_end;
    while (low <= high) {
        int32_t mid = low + (high - low) / 2;
        int32_t new_start, offset;
        idx_lookup_longest_step(x, s, mid, range_start, range_end, s_start, s_end, &new_start, &offset);
        if (offset != -1) {
            s_start = new_start;
            *index = offset;
            *length = mid;
            low = mid + 1;
            continue;
        }
        high = mid - 1;
    }
}

void ora_lookup_longest(const uint8_t *data, int32_t n, const int32_t *sa,
                        const uint8_t *s, int32_t min_end, int32_t max_end,
                        int32_t range_start, int32_t range_end,
                        int32_t *index, int32_t *length)
{
    ora_index x = { data, n, sa };
    idx_lookup_longest(&x, s, min_end, max_end, range_start, range_end, index, length);
}

/* ------------------------------------------------------------------------------------ */
/* Back-reference types and cost model: lzss/backref.go                                  */
/* ------------------------------------------------------------------------------------ */

typedef struct {
    uint8_t delimiter;
    int nb_bits_address, nb_bits_length, nb_bits_backref;
    int nb_bytes_backref, max_address, max_length, dict_len;
} backref_type;

/* backref.go:42-53 */
static backref_type new_backref_type(uint8_t symbol, int nb_bits_address, int nb_bits_length, int dict_len)
{
    backref_type t;
    t.delimiter = symbol;
    t.nb_bits_address = nb_bits_address;
    t.nb_bits_length = nb_bits_length;
    t.nb_bits_backref = 8 + nb_bits_address + nb_bits_length;
    t.nb_bytes_backref = (8 + nb_bits_address + nb_bits_length + 7) / 8;
    t.max_address = 1 << nb_bits_address;
    t.max_length = 1 << nb_bits_length;
    t.dict_len = dict_len;
    return t;
}
/* backref.go:32-35 */
static backref_type new_short_type(void)
{ return new_backref_type(LZSS_SYMBOL_SHORT, LZSS_SHORT_ADDR_BITS, LZSS_MAX_LEN_LOG2, 0); }
/* backref.go:37-40: the bound is fixed at 21 bits, addressableBytes is ignored */
static backref_type new_dynamic_type(int dict_len, int addressable_bytes)
{ (void)addressable_bytes; return new_backref_type(LZSS_SYMBOL_DYNAMIC, LZSS_DYN_ADDR_BITS, LZSS_MAX_LEN_LOG2, dict_len); }

typedef struct { int address, length; backref_type type; } backref;

#define SAVINGS_NONE LONG_MIN
/* backref.go:88-93 */
static long backref_savings(const backref *b)
{
    if (b->length == -1) return SAVINGS_NONE;
    return 8L * b->length - b->type.nb_bits_backref;
}

/* ------------------------------------------------------------------------------------ */
/* Bit writers (bitio.Writer over outBuf; bitCounterWriter compress.go:409-425)          */
/* ------------------------------------------------------------------------------------ */

typedef struct {
    uint8_t *buf; size_t len, cap;
} bytebuf;

static void bb_reserve(bytebuf *b, size_t extra)
{
    if (b->len + extra <= b->cap) return;
    size_t nc = b->cap ? b->cap : 256;
    while (nc < b->len + extra) nc *= 2;
    b->buf = (uint8_t *)realloc(b->buf, nc);
    b->cap = nc;
}
static void bb_push(bytebuf *b, uint8_t v) { bb_reserve(b, 1); b->buf[b->len++] = v; }
static void bb_append(bytebuf *b, const uint8_t *p, size_t n)
{ if (n) { bb_reserve(b, n); memmove(b->buf + b->len, p, n); b->len += n; } }

typedef struct {
    bytebuf *out;      /* NULL => counting only */
    uint64_t acc;      /* pending bits, right aligned */
    int nacc;          /* number of pending bits, < 8 between calls */
    uint64_t nb_bits;  /* total bits written (counter mode) */
} bitwriter;

/* bitio WriteBits: the n low bits of v, most significant first */
static void bw_write_bits(bitwriter *w, uint64_t v, int n)
{
    w->nb_bits += (uint64_t)n;
    if (!w->out) return;
    if (n < 64) v &= (((uint64_t)1) << n) - 1;
    w->acc = (w->acc << n) | v;
    w->nacc += n;
    while (w->nacc >= 8) {
        bb_push(w->out, (uint8_t)(w->acc >> (w->nacc - 8)));
        w->nacc -= 8;
    }
    w->acc &= (((uint64_t)1) << w->nacc) - 1;
}
static void bw_write_byte(bitwriter *w, uint8_t b) { bw_write_bits(w, b, 8); }
/* bitio Align: zero-pad to the byte boundary, return the number of pad bits */
static int bw_align(bitwriter *w)
{
    if (!w->out || w->nacc == 0) return 0;
    int skipped = 8 - w->nacc;
    bb_push(w->out, (uint8_t)(w->acc << skipped));
    w->acc = 0; w->nacc = 0;
    return skipped;
}

/* backref.go:63-68 */
static void backref_write_to(const backref *b, bitwriter *w, int i)
{
    bw_write_byte(w, b->type.delimiter);
    bw_write_bits(w, (uint64_t)(int64_t)(b->length - 1), b->type.nb_bits_length);
    int64_t addr_to_write = ((int64_t)i + b->type.dict_len) - b->address - 1;
    bw_write_bits(w, (uint64_t)addr_to_write, b->type.nb_bits_address);
}

/* ------------------------------------------------------------------------------------ */
/* Compressor: lzss/compress.go                                                          */
/* ------------------------------------------------------------------------------------ */

struct ora_compressor {
    bytebuf out;
    bitwriter bw;                 /* invariant: no pending bits between calls */
    uint8_t nb_skipped_bits;
    bytebuf in;
    long last_out_len;
    uint8_t last_nb_skipped_bits;
    long last_in_len;
    int32_t *input_sa; size_t input_sa_cap;
    uint8_t *dict; size_t dict_len;
    int32_t *dict_sa;
    int dict_reserved_idx[2];     /* [0] for 0xFE, [1] for 0xFF (compress.go:47-59) */
    int no_compression;
    uint64_t stats[6];
};

static void set_err(char *err, size_t cap, const char *msg)
{ if (err && cap) { snprintf(err, cap, "%s", msg); } }

/* compress.go:70-88 */
size_t ora_augment_dict(const uint8_t *dict, size_t dict_len, uint8_t *out)
{
    uint8_t found = 0;
    const uint8_t mask = 0x6;
    if (dict_len) memmove(out, dict, dict_len);
    for (size_t i = 0; i < dict_len; i++) {
        uint8_t b = dict[i];
        if (b == LZSS_SYMBOL_SHORT) found |= 0x2;
        else if (b == LZSS_SYMBOL_DYNAMIC) found |= 0x4;
        else continue;
        if (found == mask) return dict_len;
    }
    out[dict_len] = LZSS_SYMBOL_SHORT;
    out[dict_len + 1] = LZSS_SYMBOL_DYNAMIC;
    return dict_len + 2;
}

/* compress.go:289-304 */
void ora_reset(ora_compressor *c)
{
    c->no_compression = 0;
    c->out.len = 0;
    bb_push(&c->out, (uint8_t)(LZSS_VERSION >> 8));  /* header.go:22-32 */
    bb_push(&c->out, (uint8_t)(LZSS_VERSION & 0xFF));
    bb_push(&c->out, 0);
    c->in.len = 0;
    c->last_out_len = (long)c->out.len;
    c->last_nb_skipped_bits = 0;
    c->nb_skipped_bits = 0;
    c->last_in_len = 0;
}

/* compress.go:37-67 */
ora_compressor *ora_new_compressor(const uint8_t *dict, size_t dict_len, char *err, size_t err_cap)
{
    uint8_t *aug = (uint8_t *)malloc(dict_len + 2);
    size_t alen = ora_augment_dict(dict, dict_len, aug);
    if (alen > LZSS_MAX_DICT_SIZE) {
        char msg[64];
        snprintf(msg, sizeof msg, "dict size must be <= %d", LZSS_MAX_DICT_SIZE);
        set_err(err, err_cap, msg);
        free(aug);
        return NULL;
    }
    ora_compressor *c = (ora_compressor *)calloc(1, sizeof *c);
    c->dict = aug; c->dict_len = alen;
    int have_fe = 0, have_ff = 0;
    c->dict_reserved_idx[0] = c->dict_reserved_idx[1] = 0;
    for (size_t i = 0; i < alen; i++) {
        uint8_t b = aug[i];
        if (b == LZSS_SYMBOL_DYNAMIC) { c->dict_reserved_idx[1] = (int)i; have_ff = 1; }
        else if (b == LZSS_SYMBOL_SHORT) { c->dict_reserved_idx[0] = (int)i; have_fe = 1; }
        else continue;
        if (have_fe && have_ff) break;
    }
    c->bw.out = &c->out;
    c->dict_sa = (int32_t *)malloc((alen ? alen : 1) * sizeof(int32_t));
    ora_suffix_array(c->dict, (int32_t)alen, c->dict_sa);
    ora_reset(c);
    return c;
}

void ora_free_compressor(ora_compressor *c)
{
    if (!c) return;
    free(c->out.buf); free(c->in.buf); free(c->input_sa); free(c->dict); free(c->dict_sa);
    free(c);
}

/* compress.go:428-430 */
static int can_encode_symbol(uint8_t b) { return b != LZSS_SYMBOL_DYNAMIC && b != LZSS_SYMBOL_SHORT; }

/* compress.go:435-470 */
static void find_backref(const uint8_t *data, int n, int i, const backref_type *bt, int min_length,
                         const ora_index *data_index, const ora_index *dict_index, int dict_len,
                         int *addr, int *length)
{
    if (min_length == -1) min_length = bt->nb_bytes_backref;
    if (i + min_length > n) { *addr = -1; *length = -1; return; }
    int window_start = i - bt->max_address; if (window_start < 0) window_start = 0;
    int max_length = 1 << LZSS_MAX_LEN_LOG2;
    if (i + max_length > n) max_length = n - i;
    if (min_length > max_length) { *addr = -1; *length = -1; return; }

    int32_t a, l;
    idx_lookup_longest(data_index, data + i, min_length, max_length, window_start, i, &a, &l);
    *addr = a; *length = l;
    if (bt->delimiter == LZSS_SYMBOL_DYNAMIC) *addr += dict_len;
    if (*length < max_length && bt->delimiter == LZSS_SYMBOL_DYNAMIC) {
        int32_t da, dl;
        idx_lookup_longest(dict_index, data + i, min_length, max_length, 0, dict_len, &da, &dl);
        if (dl > *length) { *addr = da; *length = dl; }
    }
}

/* the bestBackref closure of compress.go:146-173 without its memo (circularBuffer,
 * compress.go:262-287, only caches a pure function of (d, at, dict)) */
static backref best_backref(const ora_compressor *c, const uint8_t *d, int n, int at,
                            const ora_index *input_index, const ora_index *dict_index)
{
    int dict_len = (int)c->dict_len;
    backref b_dyn, b_short;
    b_dyn.type = new_dynamic_type(dict_len, at); b_dyn.length = -1; b_dyn.address = -1;
    b_short.type = new_short_type(); b_short.length = -1; b_short.address = -1;
    int min_len = -1;
    if (!can_encode_symbol(d[at])) min_len = 1;
    find_backref(d, n, at, &b_short.type, min_len, input_index, dict_index, dict_len, &b_short.address, &b_short.length);
    find_backref(d, n, at, &b_dyn.type, min_len, input_index, dict_index, dict_len, &b_dyn.address, &b_dyn.length);
    if (b_short.length != -1 && backref_savings(&b_short) > backref_savings(&b_dyn)) return b_short;
    return b_dyn;
}

typedef struct { int key[3]; backref b[3]; int k; } circ_buf; /* compress.go:262-287 */

static backref best_backref_memo(const ora_compressor *c, circ_buf *cb, const uint8_t *d, int n, int at,
                                 const ora_index *input_index, const ora_index *dict_index, long *savings)
{
    for (int i = 0; i < 3; i++)
        if (cb->key[i] == at) { *savings = backref_savings(&cb->b[i]); return cb->b[i]; }
    backref b = best_backref(c, d, n, at, input_index, dict_index);
    cb->key[cb->k] = at; cb->b[cb->k] = b; cb->k = (cb->k + 1) % 3;
    *savings = backref_savings(&b);
    return b;
}

static void count_ref(ora_compressor *c, const backref *b, int forced)
{
    if (forced) c->stats[4]++;
    if (b->type.delimiter == LZSS_SYMBOL_SHORT) c->stats[1]++;
    else if (b->address < b->type.dict_len) c->stats[3]++;
    else c->stats[2]++;
}

/* compress.go:138-260 */
static int compressor_write_phrases(ora_compressor *c, bitwriter *w, const uint8_t *d, int n, int start_index,
                                    const ora_index *input_index)
{
    int dict_len = (int)c->dict_len;
    ora_index dict_index = { c->dict, (int32_t)c->dict_len, c->dict_sa };
    backref_type short_type = new_short_type();
    circ_buf cb; cb.key[0] = cb.key[1] = cb.key[2] = -1; cb.k = 0;
    memset(c->stats, 0, sizeof c->stats);

    const int min_repeating_bytes = LZSS_MIN_REPEAT;
    for (int i = start_index; i < n;) {
        int count = 0;
        while (i + count < n && count < (1 << LZSS_MAX_LEN_LOG2) && d[i] == d[i + count]) count++;
        if (count >= min_repeating_bytes) {
            c->stats[5]++;
            if (!(i > 0 && d[i - 1] == d[i])) {
                if (!can_encode_symbol(d[i])) {
                    backref b_dict;
                    b_dict.type = new_dynamic_type(dict_len, i);
                    b_dict.address = c->dict_reserved_idx[d[i] == LZSS_SYMBOL_DYNAMIC ? 1 : 0];
                    b_dict.length = 1;
                    backref_write_to(&b_dict, w, i);
                    count_ref(c, &b_dict, 1);
                } else {
                    bw_write_byte(w, d[i]);
                    c->stats[0]++;
                }
                i++;
                count--;
            }
            backref b_short, b_dyn;
            b_short.type = short_type; b_short.address = i - 1; b_short.length = count;
            b_dyn.type = new_dynamic_type(dict_len, i); b_dyn.address = dict_len + i - 1; b_dyn.length = count;
            if (backref_savings(&b_short) > backref_savings(&b_dyn)) { backref_write_to(&b_short, w, i); count_ref(c, &b_short, 0); }
            else { backref_write_to(&b_dyn, w, i); count_ref(c, &b_dyn, 0); }
            i += count;
            continue;
        }

        long best_savings;
        backref best_at_i = best_backref_memo(c, &cb, d, n, i, input_index, &dict_index, &best_savings);
        if (!can_encode_symbol(d[i])) {
            backref_write_to(&best_at_i, w, i);
            count_ref(c, &best_at_i, 1);
            i += best_at_i.length;
            continue;
        }
        if (best_savings < 0) {
            bw_write_byte(w, d[i]);
            c->stats[0]++;
            i++;
            continue;
        }
        if (i + 1 < n) {
            long new_savings;
            (void)best_backref_memo(c, &cb, d, n, i + 1, input_index, &dict_index, &new_savings);
            if (new_savings > best_savings + 1) {
                bw_write_byte(w, d[i]);
                c->stats[0]++;
                i++;
                continue;
            }
        }
        if (i + 2 < n && can_encode_symbol(d[i + 1])) {
            long new_savings;
            (void)best_backref_memo(c, &cb, d, n, i + 2, input_index, &dict_index, &new_savings);
            if (new_savings > best_savings + 2) {
                bw_write_byte(w, d[i]);
                bw_write_byte(w, d[i + 1]);
                c->stats[0] += 2;
                i += 2;
                continue;
            }
        }
        backref_write_to(&best_at_i, w, i);
        count_ref(c, &best_at_i, 0);
        i += best_at_i.length;
    }
    return n - start_index;
}

static int32_t *ensure_input_sa(ora_compressor *c, size_t n)
{
    if (n > c->input_sa_cap) {
        free(c->input_sa);
        c->input_sa = (int32_t *)malloc((n ? n : 1) * sizeof(int32_t));
        c->input_sa_cap = n;
    }
    return c->input_sa;
}

/* compress.go:91-129 */
long ora_write(ora_compressor *c, const uint8_t *d, size_t n, char *err, size_t err_cap)
{
    /* reconstruct bit writer cache (compress.go:94-100) */
    c->last_out_len = (long)c->out.len;
    uint8_t last_byte = c->out.buf[c->out.len - 1];
    c->out.len -= 1;
    last_byte = (uint8_t)(last_byte >> c->nb_skipped_bits);
    bw_write_bits(&c->bw, last_byte, 8 - c->nb_skipped_bits);

    c->last_nb_skipped_bits = c->nb_skipped_bits;
    /* appendInput compress.go:472-479 */
    if (c->in.len + n > LZSS_MAX_INPUT_SIZE) {
        char msg[64];
        snprintf(msg, sizeof msg, "input size must be <= %d", LZSS_MAX_INPUT_SIZE);
        set_err(err, err_cap, msg);
        return -1;
    }
    c->last_in_len = (long)c->in.len;
    bb_append(&c->in, d, n);

    if (c->no_compression) {
        bb_append(&c->out, d, n);
        return (long)n;
    }
    const uint8_t *all = c->in.buf;
    int total = (int)c->in.len;
    int32_t *sa = ensure_input_sa(c, (size_t)total);
    ora_suffix_array(all, total, sa);
    ora_index input_index = { all, total, sa };
    int written = compressor_write_phrases(c, &c->bw, all, total, (int)c->last_in_len, &input_index);
    c->nb_skipped_bits = (uint8_t)bw_align(&c->bw);
    return written;
}

size_t ora_len(const ora_compressor *c) { return c->out.len; }
size_t ora_written(const ora_compressor *c) { return c->in.len; }
const uint8_t *ora_written_bytes(const ora_compressor *c) { return c->in.buf; }
const uint8_t *ora_bytes(const ora_compressor *c) { return c->out.buf; }

/* compress.go:348-367 */
int ora_consider_bypassing(ora_compressor *c)
{
    if (c->out.len > c->in.len + LZSS_HEADER_SIZE) {
        c->no_compression = 1;
        c->nb_skipped_bits = 0;
        c->last_out_len = c->last_in_len + LZSS_HEADER_SIZE;
        c->last_nb_skipped_bits = 0;
        c->out.len = 0;
        bb_push(&c->out, (uint8_t)(LZSS_VERSION >> 8));
        bb_push(&c->out, (uint8_t)(LZSS_VERSION & 0xFF));
        bb_push(&c->out, 1);
        bb_append(&c->out, c->in.buf, c->in.len);
        return 1;
    }
    return 0;
}

/* compress.go:324-345 */
int ora_revert(ora_compressor *c, char *err, size_t err_cap)
{
    if (c->last_in_len == -1) { set_err(err, err_cap, "cannot revert twice in a row"); return -1; }
    c->in.len = (size_t)c->last_in_len;
    c->last_in_len = -1;
    if (c->no_compression) {
        size_t n = c->in.len;
        uint8_t *copy = (uint8_t *)malloc(n ? n : 1);
        if (n) memcpy(copy, c->in.buf, n);
        ora_reset(c);
        long r = ora_write(c, copy, n, err, err_cap);
        free(copy);
        if (r < 0) return -1;
        ora_consider_bypassing(c);
        return 0;
    }
    c->out.len = (size_t)c->last_out_len;
    c->nb_skipped_bits = c->last_nb_skipped_bits;
    return 0;
}

/* compress.go:375-379 */
int ora_compress(ora_compressor *c, const uint8_t *d, size_t n, char *err, size_t err_cap)
{
    ora_reset(c);
    return ora_write(c, d, n, err, err_cap) < 0 ? -1 : 0;
}

/* compress.go:384-407 */
long ora_compressed_size_256k(ora_compressor *c, const uint8_t *d, size_t n, char *err, size_t err_cap)
{
    long size = LZSS_HEADER_SIZE;
    if (c->no_compression) return size + (long)n;
    if (n > LZSS_SIZE256K_MAX) {
        char msg[64];
        snprintf(msg, sizeof msg, "input size must be <= %d", LZSS_SIZE256K_MAX);
        set_err(err, err_cap, msg);
        return -1;
    }
    int32_t *sa = (int32_t *)malloc((n ? n : 1) * sizeof(int32_t));
    ora_suffix_array(d, (int32_t)n, sa);
    ora_index index = { d, (int32_t)n, sa };
    bitwriter counter; memset(&counter, 0, sizeof counter);
    compressor_write_phrases(c, &counter, d, (int)n, 0, &index);
    free(sa);
    return size + (long)((counter.nb_bits + 7) / 8);
}

int ora_best_backrefs(ora_compressor *c, const uint8_t *d, size_t n, size_t start,
                      int32_t *addr, int32_t *len, uint8_t *type)
{
    int32_t *sa = (int32_t *)malloc((n ? n : 1) * sizeof(int32_t));
    ora_suffix_array(d, (int32_t)n, sa);
    ora_index index = { d, (int32_t)n, sa };
    ora_index dict_index = { c->dict, (int32_t)c->dict_len, c->dict_sa };
    for (size_t at = start; at < n; at++) {
        backref b = best_backref(c, d, (int)n, (int)at, &index, &dict_index);
        size_t o = at - start;
        if (b.length == -1) { type[o] = 0; addr[o] = -1; len[o] = -1; }
        else { type[o] = b.type.delimiter; addr[o] = b.address; len[o] = b.length; }
    }
    free(sa);
    return 0;
}

void ora_phrase_stats(const ora_compressor *c, uint64_t stats[6])
{ memcpy(stats, c->stats, sizeof c->stats); }

/* ------------------------------------------------------------------------------------ */
/* Decompress: lzss/decompress.go:17-84                                                   */
/* ------------------------------------------------------------------------------------ */

typedef struct { const uint8_t *p; size_t n; size_t bitpos; int err; } bitreader;

/* bitio TryReadBits: MSB first; on EOF latch the error and return 0 */
static uint64_t br_read_bits(bitreader *r, int nbits)
{
    if (r->err) return 0;
    if (r->bitpos + (size_t)nbits > r->n * 8) { r->err = 1; r->bitpos = r->n * 8; return 0; }
    uint64_t v = 0;
    for (int k = 0; k < nbits; k++) {
        size_t b = r->bitpos++;
        v = (v << 1) | ((r->p[b >> 3] >> (7 - (b & 7))) & 1);
    }
    return v;
}

/* backref.go:70-86 */
static int backref_read_from(backref *b, bitreader *r)
{
    uint64_t v = br_read_bits(r, b->type.nb_bits_length);
    b->length = (int)v + 1;
    v = br_read_bits(r, b->type.nb_bits_address);
    b->address = (int)v + 1;
    if (r->err) return -1;
    if (b->length <= 0 || b->address < 0) return -2;
    return 0;
}

int ora_decompress(const uint8_t *data, size_t n, const uint8_t *dict_in, size_t dict_in_len,
                   uint8_t **out_p, size_t *out_len, char *err, size_t err_cap)
{
    *out_p = NULL; *out_len = 0;
    /* header.go:34-44 */
    if (n < LZSS_HEADER_SIZE) { set_err(err, err_cap, "failed to read header: unexpected EOF"); return -1; }
    unsigned version = ((unsigned)data[0] << 8) | data[1];
    if (data[2] > 1) { set_err(err, err_cap, "failed to read header: expected 0 or 1"); return -1; }
    if (version != LZSS_VERSION) { set_err(err, err_cap, "unsupported compressor version"); return -1; }
    if (data[2] == 1) {
        size_t m = n - LZSS_HEADER_SIZE;
        uint8_t *o = (uint8_t *)malloc(m ? m : 1);
        if (m) memcpy(o, data + LZSS_HEADER_SIZE, m);
        *out_p = o; *out_len = m;
        return 0;
    }
    uint8_t *dict = (uint8_t *)malloc(dict_in_len + 2);
    size_t dict_len = ora_augment_dict(dict_in, dict_in_len, dict);
    backref b_short; b_short.type = new_short_type();
    bytebuf out; memset(&out, 0, sizeof out);
    bb_reserve(&out, n * 7 + 16);
    bitreader r = { data, n, LZSS_HEADER_SIZE * 8, 0 };
    int rc = 0;
    uint8_t s = (uint8_t)br_read_bits(&r, 8);
    while (!r.err) {
        if (s == LZSS_SYMBOL_SHORT) {
            if (backref_read_from(&b_short, &r) != 0) { set_err(err, err_cap, "EOF"); rc = -1; break; }
            for (int i = 0; i < b_short.length; i++) {
                if ((size_t)b_short.address > out.len) { set_err(err, err_cap, "invalid short backref"); rc = -1; break; }
                bb_push(&out, out.buf[out.len - (size_t)b_short.address]);
            }
            if (rc) break;
        } else if (s == LZSS_SYMBOL_DYNAMIC) {
            backref b_dyn; b_dyn.type = new_dynamic_type((int)dict_len, (int)out.len);
            if (backref_read_from(&b_dyn, &r) != 0) { set_err(err, err_cap, "EOF"); rc = -1; break; }
            if ((size_t)b_dyn.address > out.len) {
                long dict_start = (long)dict_len - ((long)b_dyn.address - (long)out.len);
                if (dict_start < 0 || dict_start > (long)dict_len || dict_start + b_dyn.length > (long)dict_len) {
                    set_err(err, err_cap, "invalid dynamic backref"); rc = -1; break;
                }
                bb_append(&out, dict + dict_start, (size_t)b_dyn.length);
            } else {
                for (int i = 0; i < b_dyn.length; i++)
                    bb_push(&out, out.buf[out.len - (size_t)b_dyn.address]);
            }
        } else {
            bb_push(&out, s);
        }
        s = (uint8_t)br_read_bits(&r, 8);
    }
    free(dict);
    if (rc) { free(out.buf); return rc; }
    if (!out.buf) out.buf = (uint8_t *)malloc(1);
    *out_p = out.buf; *out_len = out.len;
    return 0;
}

void ora_free(void *p) { free(p); }
