// LPIPS (net-lin, AlexNet taps) glue kernels.  Reference: src/loss/perceptual_similarity/perceptual_loss.py:26-46,
// networks_basic.py:61-108.  The AlexNet convolutions themselves run on the gconv engine; these kernels are
// the HBM-bound pieces: input scaling, per-pixel channel unit-normalisation + squared difference + 1x1 "lin"
// + spatial mean (one fused pass per tap), and its backward with respect to the second ("pred") image.
#include "common.h"

#define EW_GRID(total) dim3((unsigned)((((total) + 255) / 256) > 16384 ? 16384 : (((total) + 255) / 256)))
#define EW_LOOP(i, total) \
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (total); i += (long long)gridDim.x * blockDim.x)

// out[n, c] = ((a*x + b) - shift_c)/scale_c, n in [0,B): src0 (target), n in [B,2B): src1 (pred); a=2,b=-1 when
// `normalize` (images in [0,1]), else a=1,b=0
template <typename T>
__global__ void lpips_prep_kernel(const void* __restrict__ src0, int s0_f32, const void* __restrict__ src1, int s1_f32,
                                  T* __restrict__ out, int B, int HW, float a, float b) {
    const float shift[3] = {-.030f, -.088f, -.188f};
    const float scale[3] = {.458f, .448f, .450f};
    const long long half = (long long)B * 3 * HW;
    EW_LOOP(i, 2 * half) {
        const bool second = i >= half;
        const long long j = second ? i - half : i;
        const int c = (int)((j / HW) % 3);
        const void* s = second ? src1 : src0;
        const int f32 = second ? s1_f32 : s0_f32;
        const float v = f32 ? ((const float*)s)[j] : bf2f(((const bf16_t*)s)[j]);
        DT<T>::st(out + i, ((a * v + b) - shift[c]) / scale[c]);
    }
}
// dsrc1[j] = dgen[j] * a / scale_c      (dgen = gradient w.r.t. the pred half of the prepped batch, [B,3,HW])
template <typename T, typename TO>
__global__ void lpips_prep_bwd_kernel(const T* __restrict__ dgen, TO* __restrict__ dsrc1, int B, int HW, float a) {
    const float scale[3] = {.458f, .448f, .450f};
    const long long half = (long long)B * 3 * HW;
    EW_LOOP(j, half) {
        const int c = (int)((j / HW) % 3);
        DT<TO>::st(dsrc1 + j, DT<T>::ld(dgen + j) * a / scale[c]);
    }
}

// f: [2B, C, HW]; part[b][chunk] = sum over the chunk's 64 pixels of sum_c w_c (f0/|f0| - f1/|f1|)^2
// Channel loops run LP_U channels per trip with all 2*LP_U loads in flight (clamped indices, masked contributions):
// a one-channel-per-iteration loop is a chain of memory round trips (up to 96 of them for 384 channels).
#define LP_U 8
template <typename T>
__global__ __launch_bounds__(256) void lpips_tap_fwd_kernel(const T* __restrict__ f, const float* __restrict__ w,
                                                            float* __restrict__ part, int B, int C, int HW, float eps,
                                                            unsigned* ticket, float* val, float inv_hw, int accumulate) {
    __shared__ float r0[4][64], r1[4][64];
    const int px = threadIdx.x & 63, cg = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int hw0 = blockIdx.x * 64 + px;
    const bool ok = hw0 < HW;
    const int hw = ok ? hw0 : HW - 1;
    const T* f0 = f + (size_t)b * C * HW + hw;
    const T* f1 = f + (size_t)(B + b) * C * HW + hw;
    float s0 = 0.f, s1 = 0.f;
    for (int c0 = cg; c0 < C; c0 += 4 * LP_U) {
        float a[LP_U], bb[LP_U];
#pragma unroll
        for (int j = 0; j < LP_U; ++j) {
            const int c = c0 + 4 * j; const size_t o = (size_t)(c < C ? c : C - 1) * HW;
            a[j] = DT<T>::ld(f0 + o); bb[j] = DT<T>::ld(f1 + o);
        }
#pragma unroll
        for (int j = 0; j < LP_U; ++j)
            if (ok && c0 + 4 * j < C) { s0 += a[j] * a[j]; s1 += bb[j] * bb[j]; }
    }
    r0[cg][px] = s0; r1[cg][px] = s1;
    __syncthreads();
    const float n0 = sqrtf(r0[0][px] + r0[1][px] + r0[2][px] + r0[3][px] + eps);
    const float n1 = sqrtf(r1[0][px] + r1[1][px] + r1[2][px] + r1[3][px] + eps);
    __syncthreads();
    float d = 0.f;
    for (int c0 = cg; c0 < C; c0 += 4 * LP_U) {
        float a[LP_U], bb[LP_U], wv[LP_U];
#pragma unroll
        for (int j = 0; j < LP_U; ++j) {
            const int c = c0 + 4 * j; const int ci = c < C ? c : C - 1; const size_t o = (size_t)ci * HW;
            a[j] = DT<T>::ld(f0 + o); bb[j] = DT<T>::ld(f1 + o); wv[j] = w[ci];
        }
#pragma unroll
        for (int j = 0; j < LP_U; ++j)
            if (ok && c0 + 4 * j < C) { const float u = a[j] / n0, v = bb[j] / n1; d += wv[j] * (u - v) * (u - v); }
    }
    d = wave_sum(d);
    if (px == 0) r0[cg][0] = d;
    __syncthreads();
    if (threadIdx.x == 0) hific_st_agent(part + (size_t)b * gridDim.x + blockIdx.x, r0[0][0] + r0[1][0] + r0[2][0] + r0[3][0]);
    // lpips_tap_reduce_kernel's sum for image pair b, by the last-arriving of its gridDim.x workgroups (one ticket per b)
    if (ticket && hific_last_block(ticket + b, gridDim.x)) {
        const int nchunk = (int)gridDim.x;
        float s = 0.f;
        for (int i = threadIdx.x; i < nchunk; i += 256) s += hific_ld_agent(part + (size_t)b * nchunk + i);
        s = wave_sum(s);
        if (px == 0) r1[cg][0] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            const float t = (r1[0][0] + r1[1][0] + r1[2][0] + r1[3][0]) * inv_hw;
            if (accumulate) val[b] += t; else val[b] = t;
        }
    }
}
// val[b] (=|+=) sum_chunks part[b][chunk] / HW
__global__ __launch_bounds__(256) void lpips_tap_reduce_kernel(const float* __restrict__ part, float* __restrict__ val,
                                                               int nchunk, float inv_hw, int accumulate) {
    __shared__ float sh[4];
    const int b = blockIdx.x;
    float s = 0.f;
    for (int i = threadIdx.x; i < nchunk; i += 256) s += part[(size_t)b * nchunk + i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float t = (sh[0] + sh[1] + sh[2] + sh[3]) * inv_hw;
        if (accumulate) val[b] += t; else val[b] = t;
    }
}
// df1[b,c,hw] (=|+=) (1/n1) (gv_c - v_c sum_c' gv_c' v_c'),  gv_c = -2 w_c (u_c - v_c) * gval[b] / HW
template <typename T>
__global__ __launch_bounds__(256) void lpips_tap_bwd_kernel(const T* __restrict__ f, const float* __restrict__ w,
                                                            const float* __restrict__ gval, T* __restrict__ df1, int B,
                                                            int C, int HW, float eps, int accumulate) {
    __shared__ float r0[4][64], r1[4][64];
    const int px = threadIdx.x & 63, cg = threadIdx.x >> 6;
    const int b = blockIdx.y;
    const int hw0 = blockIdx.x * 64 + px;
    const bool ok = hw0 < HW;
    const int hw = ok ? hw0 : HW - 1;
    const T* f0 = f + (size_t)b * C * HW + hw;
    const T* f1 = f + (size_t)(B + b) * C * HW + hw;
    const float g = gval[b] / (float)HW;
    float s0 = 0.f, s1 = 0.f;
    for (int c0 = cg; c0 < C; c0 += 4 * LP_U) {
        float a[LP_U], bb[LP_U];
#pragma unroll
        for (int j = 0; j < LP_U; ++j) {
            const int c = c0 + 4 * j; const size_t o = (size_t)(c < C ? c : C - 1) * HW;
            a[j] = DT<T>::ld(f0 + o); bb[j] = DT<T>::ld(f1 + o);
        }
#pragma unroll
        for (int j = 0; j < LP_U; ++j)
            if (ok && c0 + 4 * j < C) { s0 += a[j] * a[j]; s1 += bb[j] * bb[j]; }
    }
    r0[cg][px] = s0; r1[cg][px] = s1;
    __syncthreads();
    const float n0 = sqrtf(r0[0][px] + r0[1][px] + r0[2][px] + r0[3][px] + eps);
    const float n1 = sqrtf(r1[0][px] + r1[1][px] + r1[2][px] + r1[3][px] + eps);
    __syncthreads();
    float dot = 0.f;
    for (int c0 = cg; c0 < C; c0 += 4 * LP_U) {
        float a[LP_U], bb[LP_U], wv[LP_U];
#pragma unroll
        for (int j = 0; j < LP_U; ++j) {
            const int c = c0 + 4 * j; const int ci = c < C ? c : C - 1; const size_t o = (size_t)ci * HW;
            a[j] = DT<T>::ld(f0 + o); bb[j] = DT<T>::ld(f1 + o); wv[j] = w[ci];
        }
#pragma unroll
        for (int j = 0; j < LP_U; ++j)
            if (ok && c0 + 4 * j < C) { const float u = a[j] / n0, v = bb[j] / n1; dot += -2.f * wv[j] * (u - v) * g * v; }
    }
    r0[cg][px] = dot;
    __syncthreads();
    const float S = r0[0][px] + r0[1][px] + r0[2][px] + r0[3][px];
    T* dp = df1 + (size_t)b * C * HW + hw;
    for (int c0 = cg; c0 < C; c0 += 4 * LP_U) {
        float a[LP_U], bb[LP_U], wv[LP_U], old[LP_U];
#pragma unroll
        for (int j = 0; j < LP_U; ++j) {
            const int c = c0 + 4 * j; const int ci = c < C ? c : C - 1; const size_t o = (size_t)ci * HW;
            a[j] = DT<T>::ld(f0 + o); bb[j] = DT<T>::ld(f1 + o); wv[j] = w[ci];
            old[j] = accumulate ? DT<T>::ld(dp + o) : 0.f;
        }
#pragma unroll
        for (int j = 0; j < LP_U; ++j) {
            const int c = c0 + 4 * j;
            if (ok && c < C) {
                const float u = a[j] / n0, v = bb[j] / n1;
                const float gv = -2.f * wv[j] * (u - v) * g;
                DT<T>::st(dp + (size_t)c * HW, (gv - v * S) / n1 + old[j]);
            }
        }
    }
}

extern "C" {

int hific_lpips_prep(const void* src0, int s0_f32, const void* src1, int s1_f32, void* out, int B, int HW,
                     int normalize, int dtype, hipStream_t st) {
    const float a = normalize ? 2.f : 1.f, b = normalize ? -1.f : 0.f;
    const long long total = 2LL * B * 3 * HW;
    if (dtype == HIFIC_F32)
        hipLaunchKernelGGL(lpips_prep_kernel<float>, EW_GRID(total), dim3(256), 0, st, src0, s0_f32, src1, s1_f32, (float*)out, B, HW, a, b);
    else if (dtype == HIFIC_BF16)
        hipLaunchKernelGGL(lpips_prep_kernel<bf16_t>, EW_GRID(total), dim3(256), 0, st, src0, s0_f32, src1, s1_f32, (bf16_t*)out, B, HW, a, b);
    else return HIFIC_ERR_ARG;
    return hific_launch_status();
}
// dgen: [B,3,HW] dtype (gradient of the pred half only; the target half carries none); dsrc1: [B,3,HW]
// (f32 when out_f32 else dtype)
int hific_lpips_prep_bwd(const void* dout, void* dsrc1, int B, int HW, int normalize, int dtype, int out_f32,
                         hipStream_t st) {
    const float a = normalize ? 2.f : 1.f;
    const long long total = (long long)B * 3 * HW;
    if (dtype == HIFIC_F32)
        hipLaunchKernelGGL((lpips_prep_bwd_kernel<float, float>), EW_GRID(total), dim3(256), 0, st, (const float*)dout, (float*)dsrc1, B, HW, a);
    else if (dtype == HIFIC_BF16 && out_f32)
        hipLaunchKernelGGL((lpips_prep_bwd_kernel<bf16_t, float>), EW_GRID(total), dim3(256), 0, st, (const bf16_t*)dout, (float*)dsrc1, B, HW, a);
    else if (dtype == HIFIC_BF16)
        hipLaunchKernelGGL((lpips_prep_bwd_kernel<bf16_t, bf16_t>), EW_GRID(total), dim3(256), 0, st, (const bf16_t*)dout, (bf16_t*)dsrc1, B, HW, a);
    else return HIFIC_ERR_ARG;
    return hific_launch_status();
}

// val[b] (=|+=) spatial-mean of lin(|normalize(f0) - normalize(f1)|^2); ws >= B*ceil(HW/64) floats
int hific_lpips_tap_fwd(const void* f, const float* w, float* val, int B, int C, int HW, int accumulate, int dtype,
                        void* ws, size_t ws_bytes, hipStream_t st) {
    const int nchunk = cdiv(HW, 64);
    if ((size_t)B * nchunk * sizeof(float) > ws_bytes) return HIFIC_ERR_WS;
    float* part = (float*)ws;
    unsigned* tk = hific_tickets(st, B);
    const float inv_hw = 1.f / (float)HW;
    if (dtype == HIFIC_F32)
        hipLaunchKernelGGL(lpips_tap_fwd_kernel<float>, dim3(nchunk, B), dim3(256), 0, st, (const float*)f, w, part, B, C, HW, 1e-10f, tk, val, inv_hw, accumulate);
    else if (dtype == HIFIC_BF16)
        hipLaunchKernelGGL(lpips_tap_fwd_kernel<bf16_t>, dim3(nchunk, B), dim3(256), 0, st, (const bf16_t*)f, w, part, B, C, HW, 1e-10f, tk, val, inv_hw, accumulate);
    else return HIFIC_ERR_ARG;
    if (!tk) hipLaunchKernelGGL(lpips_tap_reduce_kernel, dim3(B), dim3(256), 0, st, part, val, nchunk, inv_hw, accumulate);
    return hific_launch_status();
}
// df1: [B,C,HW] gradient wrt the pred-half features; gval: [B] f32 (d loss / d val[b])
int hific_lpips_tap_bwd(const void* f, const float* w, const float* gval, void* df1, int B, int C, int HW,
                        int accumulate, int dtype, hipStream_t st) {
    const int nchunk = cdiv(HW, 64);
    if (dtype == HIFIC_F32)
        hipLaunchKernelGGL(lpips_tap_bwd_kernel<float>, dim3(nchunk, B), dim3(256), 0, st, (const float*)f, w, gval, (float*)df1, B, C, HW, 1e-10f, accumulate);
    else if (dtype == HIFIC_BF16)
        hipLaunchKernelGGL(lpips_tap_bwd_kernel<bf16_t>, dim3(nchunk, B), dim3(256), 0, st, (const bf16_t*)f, w, gval, (bf16_t*)df1, B, C, HW, 1e-10f, accumulate);
    else return HIFIC_ERR_ARG;
    return hific_launch_status();
}

}  // extern "C"
