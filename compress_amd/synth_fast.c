// synth_fast.c - the generator of compress_amd/synth.py in C, byte for byte the same
// blobs: bench.py needs 10 000 of them per run and the Python loop takes ~20 ms per blob (a pool of forked
// workers, which must not run under rocprofv3).  Host only, no GPU code; tests/test_synth_fast.py checks it
// against synth.py.  Built by compress_amd/csrc/Makefile into compress_amd/libsynth_fast.so.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    uint64_t seed;
    uint64_t k;  // outputs drawn so far
} rng_t;

static uint64_t rng_u64(rng_t *r) {
    r->k += 1;
    uint64_t z = r->seed + r->k * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static uint64_t rng_below(rng_t *r, uint64_t n) { return rng_u64(r) % n; }

typedef struct {
    uint8_t *p;
    size_t len, cap;
} buf_t;

static void buf_need(buf_t *b, size_t more) {
    if (b->len + more <= b->cap) return;
    while (b->len + more > b->cap) b->cap *= 2;
    b->p = (uint8_t *)realloc(b->p, b->cap);
}
static void put(buf_t *b, const uint8_t *s, size_t n) {
    buf_need(b, n);
    memcpy(b->p + b->len, s, n);
    b->len += n;
}
static void put_fill(buf_t *b, int c, size_t n) {
    buf_need(b, n);
    memset(b->p + b->len, c, n);
    b->len += n;
}
// rbytes(n): whole 64-bit outputs, little endian, cut to n bytes
static void put_random(buf_t *b, rng_t *r, size_t n) {
    buf_need(b, n + 8);
    size_t done = 0;
    while (done < n) {
        const uint64_t v = rng_u64(r);
        for (int i = 0; i < 8; ++i) b->p[b->len + done + i] = (uint8_t)(v >> (8 * i));
        done += 8;
    }
    b->len += n;
}

static uint8_t g_sel[64][4], g_addr[512][20], g_words[128][32];
static int g_pools_ready = 0;

static void pools(void) {
    if (g_pools_ready) return;
    rng_t r = {0xC0FFEEull, 0};
    buf_t b = {(uint8_t *)malloc(64), 0, 64};
    for (int i = 0; i < 64; ++i) { b.len = 0; put_random(&b, &r, 4); memcpy(g_sel[i], b.p, 4); }
    for (int i = 0; i < 512; ++i) { b.len = 0; put_random(&b, &r, 20); memcpy(g_addr[i], b.p, 20); }
    for (int i = 0; i < 128; ++i) { b.len = 0; put_random(&b, &r, 32); memcpy(g_words[i], b.p, 32); }
    free(b.p);
    g_pools_ready = 1;
}

static void abi_word(rng_t *r, buf_t *out) {
    const uint64_t u = rng_below(r, 100);
    if (u < 40) {
        const size_t k = 1 + (size_t)rng_below(r, 4);
        put_fill(out, 0, 32 - k);
        put_random(out, r, k);
    } else if (u < 63) {
        put_fill(out, 0, 12);
        // synth.py: `addr[r.below(512)] if r.below(10) < 8 else r.rbytes(20)` draws the coin first
        if (rng_below(r, 10) < 8) put(out, g_addr[rng_below(r, 512)], 20);
        else put_random(out, r, 20);
    } else if (u < 77) {
        put_fill(out, 0, 32);
    } else if (u < 87) {
        put(out, g_words[rng_below(r, 128)], 32);
    } else if (u < 92) {
        put_random(out, r, 32);
    } else if (u < 93) {
        put_fill(out, 0xFF, 32);
    } else if (u < 94) {
        put_fill(out, 0xFF, 30);
        put_fill(out, 0xFE, 1);
        put_random(out, r, 1);
    } else {
        const uint64_t ln = 1 + rng_below(r, 100);
        put_fill(out, 0, 28);
        const uint8_t be[4] = {(uint8_t)(ln >> 24), (uint8_t)(ln >> 16), (uint8_t)(ln >> 8), (uint8_t)ln};
        put(out, be, 4);
        put_random(out, r, (size_t)ln);
        put_fill(out, 0, (size_t)((32 - (ln % 32)) % 32));
    }
}

static void transaction(rng_t *r, buf_t *out) {
    put_random(out, r, 3 + (size_t)rng_below(r, 7));
    const uint64_t a = rng_below(r, 64), b = rng_below(r, 64);
    put(out, g_sel[a < b ? a : b], 4);
    const uint64_t words = 1 + rng_below(r, 12);
    for (uint64_t i = 0; i < words; ++i) abi_word(r, out);
    if (rng_below(r, 64) == 0) put_fill(out, 0, 160 + (size_t)rng_below(r, 865));
}

// blob number b (seed 0x5EED0000 + b), `size` bytes into dst
void synth_fast_blob(uint64_t b, uint64_t size, uint8_t *dst) {
    pools();
    rng_t r = {0x5EED0000ull + b, 0};
    buf_t out = {(uint8_t *)malloc(size + 4096), 0, size + 4096};
    while (out.len < size) transaction(&r, &out);
    memcpy(dst, out.p, size);
    free(out.p);
}

// blobs first, first + stride, ... (count of them), each `size` bytes, back to back into dst
void synth_fast_blobs(uint64_t first, uint64_t stride, uint64_t count, uint64_t size, uint8_t *dst) {
    for (uint64_t k = 0; k < count; ++k) synth_fast_blob(first + stride * k, size, dst + k * size);
}
