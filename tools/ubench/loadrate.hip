// Micro-benchmark: global-load instruction throughput per CU by access width (L2-resident data).
// build: hipcc --offload-arch=gfx950 -O3 -o loadrate loadrate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

template <typename V, int UNROLL>
__global__ __launch_bounds__(256) void k_load(const unsigned char* __restrict__ base, unsigned* out, int iters, unsigned span_mask, int lane_stride_bytes) {
    const int tid = threadIdx.x;
    const unsigned wave_off = (blockIdx.x * 4 + (tid >> 6)) * 4096u;
    unsigned acc = 0;
    unsigned off = wave_off + (tid & 63) * lane_stride_bytes;
    for (int it = 0; it < iters; ++it) {
        V v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = *(const V*)(base + ((off + u * 8192u) & span_mask));
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            if constexpr (sizeof(V) == 2) acc += (unsigned)v[u];
            else if constexpr (sizeof(V) == 4) acc += v[u];
            else if constexpr (sizeof(V) == 8) acc += v[u][0] ^ v[u][1];
            else acc += v[u][0] ^ v[u][3];
        }
        off += 64 * 1024u + 128u;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <typename V>
static void run(const char* name, const unsigned char* buf, unsigned* out, unsigned span_mask, int lane_stride, int blocks) {
    const int iters = 2000, UN = 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_load<V, UN>), dim3(blocks), dim3(256), 0, 0, buf, out, 10, span_mask, lane_stride);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_load<V, UN>), dim3(blocks), dim3(256), 0, 0, buf, out, iters, span_mask, lane_stride);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double wave_loads = (double)blocks * 4 * iters * UN;
    const double per_cu_per_us = wave_loads / 256.0 / (ms * 1e3);
    printf("%-28s lane_stride %3d B  blocks %5d: %8.3f ms  %7.2f wave-loads/us/CU  %8.1f GB/s useful\n", name, lane_stride, blocks, ms,
           per_cu_per_us, wave_loads * 64 * sizeof(V) / (ms * 1e6));
}

int main() {
    const size_t bytes = 1u << 26;   // 64 MiB window (fits MALL/L2 partially); mask variants below restrict it
    unsigned char* buf; unsigned* out;
    hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes); hipMalloc(&out, 64);
    for (int pass = 0; pass < 2; ++pass) {
        const unsigned mask = pass == 0 ? (1u << 21) - 16 : (1u << 26) - 16;     // 2 MiB (L2 resident) / 64 MiB
        printf("--- window %u KiB\n", (mask + 16) >> 10);
        for (int blocks : {256, 512, 1024, 2048}) {
            run<unsigned short>("u16 contiguous", buf, out, mask, 2, blocks);
            run<unsigned>("u32 contiguous", buf, out, mask, 4, blocks);
            run<u32x2>("u64 contiguous", buf, out, mask, 8, blocks);
            run<u32x4>("u128 contiguous", buf, out, mask, 16, blocks);
        }
        run<unsigned short>("u16 stride 4B", buf, out, mask, 4, 1024);
        run<unsigned short>("u16 stride 64B", buf, out, mask, 64, 1024);
        run<unsigned>("u32 stride 64B", buf, out, mask, 64, 1024);
    }
    return 0;
}
