/* lzss_format.h - constants of the lzss ("linzip") stream format.
 *
 * One header shared by the CPU oracle (oracle/), the HIP kernels and the C-ABI
 * (compress_amd/csrc/).  Every value restates a constant of the reference;
 * citations are file:line relative to the reference repository root.
 */
#ifndef LZSS_FORMAT_H
#define LZSS_FORMAT_H

#define LZSS_MAX_INPUT_SIZE   (1 << 22)   /* lzss/backref.go:10  MaxInputSize  */
#define LZSS_MAX_DICT_SIZE    (1 << 22)   /* lzss/backref.go:11  MaxDictSize   */
#define LZSS_SYMBOL_DYNAMIC   0xFF        /* lzss/backref.go:15  long/dynamic  */
#define LZSS_SYMBOL_SHORT     0xFE        /* lzss/backref.go:16  short         */
#define LZSS_MAX_LEN_LOG2     8           /* lzss/backref.go:17  maxBackrefLenLog2 */
#define LZSS_MAX_LEN          (1 << LZSS_MAX_LEN_LOG2)
#define LZSS_SHORT_ADDR_BITS  14          /* lzss/backref.go:18  shortAddrBits */
#define LZSS_DYN_ADDR_BITS    21          /* lzss/backref.go:38  fixed bound   */
#define LZSS_SHORT_BITS       (8 + LZSS_MAX_LEN_LOG2 + LZSS_SHORT_ADDR_BITS) /* 30, backref.go:47 */
#define LZSS_DYN_BITS         (8 + LZSS_MAX_LEN_LOG2 + LZSS_DYN_ADDR_BITS)   /* 37 */
#define LZSS_SHORT_MIN_LEN    ((LZSS_SHORT_BITS + 7) / 8)  /* 4, nbBytesBackRef backref.go:48 */
#define LZSS_DYN_MIN_LEN      ((LZSS_DYN_BITS + 7) / 8)    /* 5 */
#define LZSS_SHORT_WINDOW     (1 << LZSS_SHORT_ADDR_BITS)  /* maxAddress backref.go:49 */
#define LZSS_DYN_WINDOW       (1 << LZSS_DYN_ADDR_BITS)
#define LZSS_MIN_REPEAT       160         /* lzss/compress.go:175 minRepeatingBytes */
#define LZSS_VERSION          1           /* lzss/header.go:11 */
#define LZSS_HEADER_SIZE      3           /* lzss/header.go:12 */
#define LZSS_SIZE256K_MAX     (1 << 18)   /* lzss/compress.go:390 */

#endif
