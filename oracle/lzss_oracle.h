/* lzss_oracle.h - CPU restatement of the reference lzss compressor (Consensys/compress).
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under compress_amd/ (the product) may include,
 * link or load this; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg use it, and only as the checker / reported baseline.
 *
 * Parity status: the reference is Go and no Go toolchain exists in the build image, so
 * the reference itself cannot be run.  This restatement is pinned by the reference's own
 * tests (round trips on lzss/testdata, the five compression-ratio windows of
 * lzss/regress_test.go:16-32, the empty-input and TestInvalidBackref known answers of
 * lzss/compress_test.go:365-405, CompressedSize256k == len(Compress)).  The reference
 * holds NO golden compressed-byte vectors, so exact bytes / offset tie-breaks are
 * "parity unpinned": they rest on this file following the cited source line by line.
 */
#ifndef LZSS_ORACLE_H
#define LZSS_ORACLE_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ora_compressor ora_compressor;

/* suffix array of data[0:n) with an implicit smallest end sentinel
 * (lzss/internal/suffixarray/suffixarray.go:32-45; the array is a unique object) */
void ora_suffix_array(const uint8_t *data, int32_t n, int32_t *sa);

/* lzss/internal/suffixarray/suffixarray.go:59-117, literal transcription */
void ora_lookup_longest(const uint8_t *data, int32_t n, const int32_t *sa,
                        const uint8_t *s, int32_t min_end, int32_t max_end,
                        int32_t range_start, int32_t range_end,
                        int32_t *index, int32_t *length);

/* lzss/compress.go:70-88.  out must hold dict_len+2 bytes; returns the augmented length */
size_t ora_augment_dict(const uint8_t *dict, size_t dict_len, uint8_t *out);

/* lzss/compress.go:37-67; NULL + err text on failure */
ora_compressor *ora_new_compressor(const uint8_t *dict, size_t dict_len, char *err, size_t err_cap);
void ora_free_compressor(ora_compressor *c);

/* lzss/compress.go:91-129; returns number of bytes consumed or -1 (err filled) */
long ora_write(ora_compressor *c, const uint8_t *d, size_t n, char *err, size_t err_cap);
void ora_reset(ora_compressor *c);                         /* compress.go:289 */
size_t ora_len(const ora_compressor *c);                   /* compress.go:307 */
size_t ora_written(const ora_compressor *c);               /* compress.go:312 */
const uint8_t *ora_written_bytes(const ora_compressor *c); /* compress.go:318 */
int ora_revert(ora_compressor *c, char *err, size_t err_cap); /* compress.go:324; 0 ok, -1 err */
int ora_consider_bypassing(ora_compressor *c);             /* compress.go:348 */
const uint8_t *ora_bytes(const ora_compressor *c);         /* compress.go:370 */
/* compress.go:375: Reset+Write; result is ora_bytes()/ora_len(); 0 ok, -1 err */
int ora_compress(ora_compressor *c, const uint8_t *d, size_t n, char *err, size_t err_cap);
/* compress.go:384-407; returns size or -1 */
long ora_compressed_size_256k(ora_compressor *c, const uint8_t *d, size_t n, char *err, size_t err_cap);

/* lzss/decompress.go:17-84.  *out is malloc'd (free with ora_free); 0 ok, -1 err */
int ora_decompress(const uint8_t *data, size_t n, const uint8_t *dict, size_t dict_len,
                   uint8_t **out, size_t *out_len, char *err, size_t err_cap);
void ora_free(void *p);

/* Diagnostics for parity tests: the bestBackref closure of lzss/compress.go:146-173 evaluated
 * at every position of d[start:n) (the index covers all of d, as in Write).
 * type[i] = 0xFE / 0xFF, or 0 when no back-reference exists; addr = backref.address
 * (dictionary-prefixed for 0xFF), len = backref.length. Arrays have n-start entries. */
int ora_best_backrefs(ora_compressor *c, const uint8_t *d, size_t n, size_t start,
                      int32_t *addr, int32_t *len, uint8_t *type);

/* phrase statistics of the last ora_write/ora_compress/ora_compressed_size_256k call:
 * [0] literals, [1] short refs, [2] long refs into data, [3] long refs into dict,
 * [4] refs forced by a reserved symbol, [5] RLE phrases (compress.go:184) */
void ora_phrase_stats(const ora_compressor *c, uint64_t stats[6]);

#ifdef __cplusplus
}
#endif
#endif
