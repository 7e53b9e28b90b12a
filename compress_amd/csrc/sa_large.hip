// sa_large.hip - suffix array + inverse of ONE text that is too large for the one-workgroup kernels of sa_build.hip
// (more than SA_BITMAP_MAX_N = 2^19 suffixes, up to MaxInputSize = 4 MiB, lzss/compress.go:473), on the whole GPU.
// Replaces suffixarray.New (lzss/internal/suffixarray/suffixarray.go:32-45) for the reference's single-file use
// (main.go:81-87).  Plain prefix doubling, every round one device-wide radix sort (rocPRIM) of all N suffixes by
// (group rank, group rank h symbols further): ~1 ms per round on 4 Mi suffixes, log2(longest repeat / 7) rounds.
// The host reads one word per round (number of groups) to know when to stop.
//
// N-world as in sa_build.hip: the text gets an explicit smallest sentinel at index n, N = n + 1 suffixes, SA[0] == n.
#include <cstring>

#include <rocprim/rocprim.hpp>

#include "kernels.h"

namespace lzss {

constexpr int SAL_THREADS = 256;
constexpr int SAL_SYMS = 7;  // symbols of the initial key, 9 bits each (0 = sentinel / past the end, byte + 1 otherwise)

__global__ __launch_bounds__(SAL_THREADS) void sal_init_kernel(const uint8_t *T, uint32_t n, uint64_t *keys, uint32_t *idx) {
    const uint32_t i = blockIdx.x * SAL_THREADS + threadIdx.x;
    if (i > n) return;
    const uint64_t w = ld8(T + i);  // (zero padding follows the text)
    uint64_t key = 0;
#pragma unroll
    for (int t = 0; t < SAL_SYMS; ++t) {
        const uint64_t s = i + (uint32_t)t < n ? ((w >> (8 * t)) & 0xFFull) + 1ull : 0ull;
        key = (key << 9) | s;
    }
    keys[i] = key;
    idx[i] = i;
}

// hd[j] = j where slot j starts a group (its key differs from the slot before), else 0; *n_groups counts the starts
__global__ __launch_bounds__(SAL_THREADS) void sal_heads_kernel(const uint64_t *keys, uint32_t N, uint32_t *hd, uint32_t *n_groups) {
    const uint32_t j = blockIdx.x * SAL_THREADS + threadIdx.x;
    bool head = false;
    if (j < N) {
        head = j == 0 || keys[j] != keys[j - 1];
        hd[j] = head ? j : 0u;
    }
    const uint64_t b = __ballot(head);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(n_groups, (uint32_t)__popcll(b));
}

// rk[suffix at slot j] = first slot of its group
__global__ __launch_bounds__(SAL_THREADS) void sal_rank_kernel(const uint32_t *idx, const uint32_t *grp, uint32_t N, int32_t *rk) {
    const uint32_t j = blockIdx.x * SAL_THREADS + threadIdx.x;
    if (j < N) rk[idx[j]] = (int32_t)grp[j];
}

// next round's key of the suffix p at slot j: (rank of p) << 23 | rank of p + h.  A suffix with p + h >= N holds the
// sentinel among its first h symbols, so it is alone in its group and its second half never decides anything.
__global__ __launch_bounds__(SAL_THREADS) void sal_keys_kernel(const uint32_t *idx, const uint32_t *grp, const int32_t *rk, uint32_t N,
                                                               uint32_t h, uint64_t *keys) {
    const uint32_t j = blockIdx.x * SAL_THREADS + threadIdx.x;
    if (j >= N) return;
    const uint32_t p = idx[j];
    const uint32_t k2 = (uint64_t)p + h < N ? (uint32_t)rk[p + h] : 0u;
    keys[j] = ((uint64_t)grp[j] << 23) | (uint64_t)k2;
}

__global__ __launch_bounds__(SAL_THREADS) void sal_copy_kernel(const uint32_t *src, uint32_t N, int32_t *dst) {
    const uint32_t j = blockIdx.x * SAL_THREADS + threadIdx.x;
    if (j < N) dst[j] = (int32_t)src[j];
}

size_t sa_large_temp_bytes(uint32_t N) {
    size_t a = 0, b = 0;
    rocprim::double_buffer<uint64_t> dk(nullptr, nullptr);
    rocprim::double_buffer<uint32_t> dv(nullptr, nullptr);
    (void)rocprim::radix_sort_pairs(nullptr, a, dk, dv, (size_t)N, 0u, 64u, (hipStream_t)0);
    (void)rocprim::inclusive_scan(nullptr, b, (uint32_t *)nullptr, (uint32_t *)nullptr, (size_t)N, rocprim::maximum<uint32_t>(),
                                  (hipStream_t)0);
    return (a > b ? a : b) + 256;
}

// Returns hipSuccess, or the error of the first call that failed.  Synchronises the stream once per round.
int launch_sa_large(const Seg &sg, const uint8_t *d_text, const SaWork &wk, hipStream_t stream) {
    const uint32_t n = sg.n, N = n + 1;
    if (N > (1u << 23)) return (int)hipErrorInvalidValue;  // the keys hold 23-bit ranks
    const uint8_t *T = d_text + sg.off;
    uint64_t *kA = wk.recA + sg.off, *kB = wk.recB + sg.off;
    uint32_t *iA = reinterpret_cast<uint32_t *>(wk.sa + sg.off), *iB = reinterpret_cast<uint32_t *>(wk.rkA + sg.off);
    uint32_t *grp = reinterpret_cast<uint32_t *>(wk.rkB + sg.off), *hd = wk.pairs + sg.off;
    int32_t *rk = wk.isa + sg.off;
    // the first 256 bytes of the scratch: the group counter; the rest is rocPRIM's
    uint32_t *d_cnt = reinterpret_cast<uint32_t *>(wk.tmp);
    void *tmp = wk.tmp + 256;
    size_t tmp_bytes = wk.tmp_bytes - 256;
    const uint32_t blocks = (N + SAL_THREADS - 1) / SAL_THREADS;
    hipError_t e;
#define SAL_TRY(x) do { e = (x); if (e != hipSuccess) return (int)e; } while (0)

    hipLaunchKernelGGL(sal_init_kernel, dim3(blocks), dim3(SAL_THREADS), 0, stream, T, n, kA, iA);
    rocprim::double_buffer<uint64_t> dk(kA, kB);
    rocprim::double_buffer<uint32_t> dv(iA, iB);
    unsigned end_bit = 9 * SAL_SYMS;
    for (uint32_t h = SAL_SYMS;; h *= 2) {
        SAL_TRY(rocprim::radix_sort_pairs(tmp, tmp_bytes, dk, dv, (size_t)N, 0u, end_bit, stream));
        SAL_TRY(hipMemsetAsync(d_cnt, 0, 4, stream));
        hipLaunchKernelGGL(sal_heads_kernel, dim3(blocks), dim3(SAL_THREADS), 0, stream, dk.current(), N, hd, d_cnt);
        SAL_TRY(rocprim::inclusive_scan(tmp, tmp_bytes, hd, grp, (size_t)N, rocprim::maximum<uint32_t>(), stream));
        hipLaunchKernelGGL(sal_rank_kernel, dim3(blocks), dim3(SAL_THREADS), 0, stream, dv.current(), grp, N, rk);
        uint32_t groups = 0;
        SAL_TRY(hipMemcpyAsync(&groups, d_cnt, 4, hipMemcpyDeviceToHost, stream));
        SAL_TRY(hipStreamSynchronize(stream));
        if (groups == N || h >= N) break;
        hipLaunchKernelGGL(sal_keys_kernel, dim3(blocks), dim3(SAL_THREADS), 0, stream, dv.current(), grp, rk, N, h, dk.alternate());
        dk.swap();
        end_bit = 46;
    }
    if (dv.current() != iA) hipLaunchKernelGGL(sal_copy_kernel, dim3(blocks), dim3(SAL_THREADS), 0, stream, dv.current(), N, wk.sa + sg.off);
    SAL_TRY(hipGetLastError());
#undef SAL_TRY
    return 0;
}

// ---- the window index of ONE large text on the whole device --------------------------------------------------
// (sa_build.hip: window_index_kernel is one workgroup per text - 9 to 12 of the ~30 ms of a 4 MiB text.)  The index is
// the ranks 1..n grouped by the 16 KiB chunk of their position, rank order kept inside a chunk: a stable sort by the chunk
// number, i.e. ONE 8-bit radix pass over (chunk, rank) pairs.  Every chunk but the last holds exactly 16 384 positions and
// every position has one rank, so chunk c's list starts at entry c * 16 384 by itself - the layout the searches expect
// (window_search.h).  The chunk directory (entries of chunk c with a rank below 1 + 64 b, for every block b of 64 ranks)
// follows from the places where the block of the rank changes along a chunk's list: entry j of the list writes its
// own number into the rows of the blocks between its predecessor's and its own.
constexpr int WIL_THREADS = 256;

__global__ __launch_bounds__(WIL_THREADS) void wil_keys_kernel(const int32_t *SA, uint32_t n, uint32_t *keys, uint32_t *vals) {
    const uint32_t j = blockIdx.x * WIL_THREADS + threadIdx.x;  // entry j of the unsorted list: rank j + 1
    if (j >= n) return;
    keys[j] = (uint32_t)SA[j + 1] >> LZSS_SHORT_ADDR_BITS;
    vals[j] = j + 1;
}

// PS[j] = position of the suffix of rank KS[j]; min / max of the positions over blocks of 64 entries
__global__ __launch_bounds__(WIL_THREADS) void wil_positions_kernel(const int32_t *SA, const int32_t *KS, uint32_t n, int32_t *PS,
                                                                    int32_t *pmin1, int32_t *pmax1) {
    const uint32_t j = blockIdx.x * WIL_THREADS + threadIdx.x;
    int lo = INT_MAX, hi = INT_MIN;
    if (j < n) {
        const int p = SA[KS[j]];
        PS[j] = p;
        lo = hi = p;
    }
    lo = wave_min(lo);
    hi = wave_max(hi);
    if ((threadIdx.x & 63) == 0 && j < n) {
        pmin1[j >> 6] = lo;
        pmax1[j >> 6] = hi;
    }
}

// cdir[b * row + c] for every block b of 64 ranks, b < nb1, and every chunk c of the text
__global__ __launch_bounds__(WIL_THREADS) void wil_directory_kernel(const int32_t *KS, uint32_t n, uint32_t nb1, uint32_t row,
                                                                    uint16_t *cdir) {
    const uint32_t j = blockIdx.x * WIL_THREADS + threadIdx.x;
    if (j >= n) return;
    const uint32_t c = j >> LZSS_SHORT_ADDR_BITS, jj = j & (LZSS_SHORT_WINDOW - 1);
    const uint32_t first = c << LZSS_SHORT_ADDR_BITS;
    const uint32_t cnt = n - first < (uint32_t)LZSS_SHORT_WINDOW ? n - first : (uint32_t)LZSS_SHORT_WINDOW;  // entries of chunk c
    const uint32_t bj = ((uint32_t)KS[j] - 1u) >> 6;                       // block of this entry's rank (ranks start at 1)
    uint32_t b = jj == 0 ? 0u : (((uint32_t)KS[j - 1] - 1u) >> 6) + 1u;    // first block behind the predecessor's
    for (; b <= bj && b < nb1; ++b) cdir[(size_t)b * row + c] = (uint16_t)jj;   // the entries before this one have smaller ranks
    if (jj == cnt - 1)
        for (b = bj + 1; b < nb1; ++b) cdir[(size_t)b * row + c] = (uint16_t)cnt;
}

__global__ __launch_bounds__(WIL_THREADS) void wil_level2_kernel(const int32_t *pmin1, const int32_t *pmax1, uint32_t nb1, int32_t *pmin2,
                                                                 int32_t *pmax2) {
    const uint32_t j = blockIdx.x * WIL_THREADS + threadIdx.x;
    int lo = INT_MAX, hi = INT_MIN;
    if (j < nb1) {
        lo = pmin1[j];
        hi = pmax1[j];
    }
    lo = wave_min(lo);
    hi = wave_max(hi);
    if ((threadIdx.x & 63) == 0 && j < nb1) {
        pmin2[j >> 6] = lo;
        pmax2[j >> 6] = hi;
    }
}

size_t window_large_temp_bytes(uint32_t n) {
    size_t a = 0;
    rocprim::double_buffer<uint32_t> dk(nullptr, nullptr), dv(nullptr, nullptr);
    (void)rocprim::radix_sort_pairs(nullptr, a, dk, dv, (size_t)n, 0u, 8u, (hipStream_t)0);
    return a + 256;
}

// scratch: wk.rkA / wk.rkB (the chunk numbers, in and out), wk.pairs (the ranks before the sort), wk.tmp (rocPRIM's);
// the suffix array and its first-level block minima must be complete.  Returns hipSuccess or the first error.
int launch_window_large(const Seg &sg, const BatchIndex &ix, const SaWork &wk, hipStream_t stream) {
    const uint32_t n = sg.n;
    if (n == 0) return 0;
    if (!wk.tmp || !wk.rkA || !wk.rkB || !wk.pairs || wk.tmp_bytes < window_large_temp_bytes(n)) return (int)hipErrorInvalidValue;
    const int32_t *SA = ix.sa + sg.off;
    int32_t *KS = ix.ks(sg), *PS = ix.ps(sg);
    uint32_t *kA = reinterpret_cast<uint32_t *>(wk.rkA + sg.off), *kB = reinterpret_cast<uint32_t *>(wk.rkB + sg.off);
    uint32_t *vA = wk.pairs + sg.off, *vB = reinterpret_cast<uint32_t *>(KS);
    const uint32_t blocks = (n + WIL_THREADS - 1) / WIL_THREADS;
    hipError_t e;
    hipLaunchKernelGGL(wil_keys_kernel, dim3(blocks), dim3(WIL_THREADS), 0, stream, SA, n, kA, vA);
    rocprim::double_buffer<uint32_t> dk(kA, kB), dv(vA, vB);
    void *tmp = wk.tmp + 256;
    size_t tmp_bytes = wk.tmp_bytes - 256;
    e = rocprim::radix_sort_pairs(tmp, tmp_bytes, dk, dv, (size_t)n, 0u, 8u, stream);  // stable: rank order inside a chunk
    if (e != hipSuccess) return (int)e;
    if (dv.current() != vB) hipLaunchKernelGGL(sal_copy_kernel, dim3(blocks), dim3(WIL_THREADS), 0, stream, dv.current(), n, KS);
    const uint32_t nb1 = (n + 63) / 64;
    hipLaunchKernelGGL(wil_positions_kernel, dim3(blocks), dim3(WIL_THREADS), 0, stream, SA, KS, n, PS, ix.pmin1 + sg.off / 64,
                       ix.pmax1 + sg.off / 64);
    hipLaunchKernelGGL(wil_directory_kernel, dim3(blocks), dim3(WIL_THREADS), 0, stream, KS, n, nb1, ix.nch,
                       ix.cdir + (size_t)(sg.off / 64) * ix.nch);
    // second level of the min / max summaries: blocks of 64 first-level entries
    hipLaunchKernelGGL(wil_level2_kernel, dim3((nb1 + WIL_THREADS - 1) / WIL_THREADS), dim3(WIL_THREADS), 0, stream,
                       ix.pmin1 + sg.off / 64, ix.pmax1 + sg.off / 64, nb1, ix.pmin2 + sg.off / 64, ix.pmax2 + sg.off / 64);
    e = hipGetLastError();
    return (int)e;
}

}  // namespace lzss
