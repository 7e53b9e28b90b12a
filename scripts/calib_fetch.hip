// calib_fetch.hip - known-byte-count kernels in this path's access widths, to calibrate rocprofv3's FETCH_SIZE /
// WRITE_SIZE on gfx950 (MI355X_MICROARCH.md, HBM section: "calibrate on a known byte count in your own access
// pattern before trusting an absolute").  Build: hipcc --offload-arch=gfx950 -O2 scripts/calib_fetch.hip -o
// scripts/calib_fetch ; run under rocprofv3 --pmc FETCH_SIZE (and WRITE_SIZE) by scripts/calib_fetch.sh.
// Every kernel moves BYTES (1 GiB, four times the 256 MiB Infinity Cache) exactly once.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>

constexpr size_t BYTES = 1ull << 30;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <typename T>
__device__ inline void read_stream(const T *in, size_t n, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const T v = in[i];
        const unsigned char *b = reinterpret_cast<const unsigned char *>(&v);
        acc += b[0];
    }
    if (acc == 0xFFFFFFFFu) *sink = acc;  // never true for the zeroed buffer's sums; keeps the loads alive
}
__global__ __launch_bounds__(256) void read_x4(const u32x4 *in, size_t n, uint32_t *sink) { read_stream<u32x4>(in, n, sink); }
__global__ __launch_bounds__(256) void read_x2(const uint64_t *in, size_t n, uint32_t *sink) { read_stream<uint64_t>(in, n, sink); }
__global__ __launch_bounds__(256) void read_x1(const uint32_t *in, size_t n, uint32_t *sink) { read_stream<uint32_t>(in, n, sink); }
__global__ __launch_bounds__(256) void read_b1(const uint8_t *in, size_t n, uint32_t *sink) { read_stream<uint8_t>(in, n, sink); }

// the rank gather of a doubling round: 4-byte loads at pseudo-random indices of a 1 GiB table, one per lane;
// n_access accesses of 4 bytes each (useful bytes = 4 * n_access, lines touched = 64 or 128 B each)
__global__ __launch_bounds__(256) void gather_x1(const uint32_t *in, size_t n_words, size_t n_access, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n_access; i += (size_t)gridDim.x * 256) {
        const size_t j = (i * 0x9E3779B97F4A7C15ull >> 20) % n_words;
        acc += in[j];
    }
    if (acc == 0xFFFFFFFFu) *sink = acc;
}
// the same inside a 512 KiB neighbourhood (one text's rank array): the lines are shared by the lanes of a workgroup
__global__ __launch_bounds__(256) void gather_local_x1(const uint32_t *in, size_t n_words, size_t n_access, uint32_t *sink) {
    uint32_t acc = 0;
    const size_t per = 131072;  // words of one text's rank array
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n_access; i += (size_t)gridDim.x * 256) {
        const size_t seg = (i / per) % (n_words / per);
        const size_t j = seg * per + ((i * 0x9E3779B97F4A7C15ull >> 20) % per);
        acc += in[j];
    }
    if (acc == 0xFFFFFFFFu) *sink = acc;
}

template <typename T>
__device__ inline void write_stream(T *out, size_t n) {
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        T v;
        __builtin_memset(&v, 1, sizeof(T));
        out[i] = v;
    }
}
__global__ __launch_bounds__(256) void write_x4(u32x4 *out, size_t n) { write_stream<u32x4>(out, n); }
__global__ __launch_bounds__(256) void write_x2(uint64_t *out, size_t n) { write_stream<uint64_t>(out, n); }
__global__ __launch_bounds__(256) void write_x1(uint32_t *out, size_t n) { write_stream<uint32_t>(out, n); }

// the radix scatter of the suffix sorter: 8-byte records to 512 output streams per 1 MiB segment, runs of ~8 records
__global__ __launch_bounds__(256) void scatter_x2(uint64_t *out, size_t n) {
    const size_t per = 131072;  // records of one text
    for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const size_t seg = i / per, u = i % per;
        const size_t tile = u / 4096, t = u % 4096;
        const size_t bin = t / 8, k = t % 8;              // 512 bins x 8 records per tile
        const size_t dstu = bin * (per / 512) + tile * 8 + k;  // bin-major: each bin's stream advances 8 records per tile
        out[seg * per + dstu] = i;
    }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main() {
    void *buf = nullptr, *buf2 = nullptr;
    uint32_t *sink = nullptr;
    CK(hipMalloc(&buf, BYTES));
    CK(hipMalloc(&buf2, BYTES));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(buf, 0, BYTES));
    CK(hipMemset(buf2, 0, BYTES));
    const int grid = 256 * 16;
    for (int rep = 0; rep < 2; ++rep) {
        // every kernel streams 1 GiB, four times the Infinity Cache, so none finds the previous one's lines there
        hipLaunchKernelGGL(read_x4, dim3(grid), dim3(256), 0, 0, (const u32x4 *)buf, BYTES / 16, sink);
        hipLaunchKernelGGL(read_x2, dim3(grid), dim3(256), 0, 0, (const uint64_t *)buf, BYTES / 8, sink);
        hipLaunchKernelGGL(read_x1, dim3(grid), dim3(256), 0, 0, (const uint32_t *)buf, BYTES / 4, sink);
        hipLaunchKernelGGL(read_b1, dim3(grid), dim3(256), 0, 0, (const uint8_t *)buf, BYTES / 4, sink);  // 256 MiB of bytes
        hipLaunchKernelGGL(gather_x1, dim3(grid), dim3(256), 0, 0, (const uint32_t *)buf, BYTES / 4, BYTES / 64, sink);
        hipLaunchKernelGGL(gather_local_x1, dim3(grid), dim3(256), 0, 0, (const uint32_t *)buf, BYTES / 4, BYTES / 4, sink);
        hipLaunchKernelGGL(write_x4, dim3(grid), dim3(256), 0, 0, (u32x4 *)buf2, BYTES / 16);
        hipLaunchKernelGGL(write_x2, dim3(grid), dim3(256), 0, 0, (uint64_t *)buf2, BYTES / 8);
        hipLaunchKernelGGL(write_x1, dim3(grid), dim3(256), 0, 0, (uint32_t *)buf2, BYTES / 4);
        hipLaunchKernelGGL(scatter_x2, dim3(grid), dim3(256), 0, 0, (uint64_t *)buf2, BYTES / 8);
    }
    CK(hipDeviceSynchronize());
    printf("known bytes per launch: read_x4/x2/x1 %zu, read_b1 %zu, gather_x1 %zu accesses x 4 B (%zu B of 64-B lines), "
           "gather_local_x1 %zu accesses x 4 B over a %zu B table, write_x4/x2/x1 %zu, scatter_x2 %zu\n",
           BYTES, BYTES / 4, BYTES / 64, (BYTES / 64) * 64, BYTES / 4, BYTES, BYTES, BYTES);
    hipFree(buf);
    hipFree(buf2);
    hipFree(sink);
    return 0;
}
