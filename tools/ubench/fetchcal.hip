// Calibration of the TCC FETCH_SIZE / WRITE_SIZE counters by access width on gfx950: each kernel
// streams a 256 MiB buffer ONCE from HBM (no reuse: the algorithmic read bytes are exactly the buffer size) with 2-, 4-, 8- or
// 16-byte loads per lane, contiguous across the wave, and writes a 1/64 digest.  Run under
//     rocprofv3 --pmc FETCH_SIZE --kernel-trace ... ./fetchcal        and        --pmc WRITE_SIZE ...
// and compare counter x 1024 (KB units) with 268435456 bytes: the ratio is the factor to apply to that counter for kernels
// whose loads have that width (MI355X_MICROARCH.md prescribes x2 for FETCH_SIZE of wide streams: 64 B counted per 128 B request).
// build: hipcc --offload-arch=gfx950 -O3 -o fetchcal fetchcal.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <typename V>
__global__ __launch_bounds__(256) void stream_read(const V* __restrict__ src, unsigned* __restrict__ dig, size_t n) {
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const V v = src[i];
        if constexpr (sizeof(V) <= 4) acc += (unsigned)v;
        else if constexpr (sizeof(V) == 8) acc += v[0] ^ v[1];
        else acc += v[0] ^ v[1] ^ v[2] ^ v[3];
    }
    if ((threadIdx.x & 63) == 0) dig[blockIdx.x * 4 + (threadIdx.x >> 6)] = acc;
}
template <typename V>
__global__ __launch_bounds__(256) void stream_copy(const V* __restrict__ src, V* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
// the weight-resident kernel's access shape: 2-byte loads, lanes along pixels, one channel row (128 B) per wave instruction,
// rows 2 * HW bytes apart
__global__ __launch_bounds__(256) void stream_read_rows_u16(const unsigned short* __restrict__ src, unsigned* __restrict__ dig,
                                                            int rows, int row_elems) {
    unsigned acc = 0;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, nw = gridDim.x * 4;
    const int chunks = row_elems / 64;
    for (long long j = wave; j < (long long)rows * chunks; j += nw) {
        const int r = (int)(j % rows), c = (int)(j / rows);          // consecutive waves walk DIFFERENT rows (channel-major order)
        acc += src[(size_t)r * row_elems + c * 64 + lane];
    }
    if (lane == 0) dig[wave] = acc;
}

int main() {
    const size_t bytes = (size_t)256 << 20;
    unsigned char *a, *b; unsigned* dig;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&dig, 1 << 20);
    hipMemset(a, 3, bytes); hipMemset(b, 0, bytes); hipDeviceSynchronize();
    const int grid = 256 * 8;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(stream_read<unsigned short>, dim3(grid), dim3(256), 0, 0, (const unsigned short*)a, dig, bytes / 2);
        hipLaunchKernelGGL(stream_read<unsigned>, dim3(grid), dim3(256), 0, 0, (const unsigned*)a, dig, bytes / 4);
        hipLaunchKernelGGL(stream_read<u32x2>, dim3(grid), dim3(256), 0, 0, (const u32x2*)a, dig, bytes / 8);
        hipLaunchKernelGGL(stream_read<u32x4>, dim3(grid), dim3(256), 0, 0, (const u32x4*)a, dig, bytes / 16);
        hipLaunchKernelGGL(stream_read_rows_u16, dim3(grid), dim3(256), 0, 0, (const unsigned short*)a, dig, 1024, (int)(bytes / 2 / 1024));
        hipLaunchKernelGGL(stream_copy<unsigned short>, dim3(grid), dim3(256), 0, 0, (const unsigned short*)a, (unsigned short*)b, bytes / 2);
        hipLaunchKernelGGL(stream_copy<unsigned>, dim3(grid), dim3(256), 0, 0, (const unsigned*)a, (unsigned*)b, bytes / 4);
        hipLaunchKernelGGL(stream_copy<u32x4>, dim3(grid), dim3(256), 0, 0, (const u32x4*)a, (u32x4*)b, bytes / 16);
    }
    hipDeviceSynchronize();
    printf("fetchcal: every kernel reads %zu bytes once (copies also write them)\n", bytes);
    return 0;
}
