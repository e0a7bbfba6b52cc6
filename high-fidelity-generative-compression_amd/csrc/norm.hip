// ChannelNorm2D (reference: src/normalisation/channel.py:29-59) forward/backward on NCHW tensors.
// Per pixel (n,h,w): mu = mean_c x, var = sum_c (x-mu)^2/(C-1) (unbiased), r = rsqrt(var+eps),
// y = gamma_c*(x-mu)*r + beta_c, optionally fused with the ReLU that follows it in the Encoder/Generator.
// HBM-bound: a workgroup owns 64 consecutive pixels x all C channels; every channel row it touches is a
// 128 B (bf16) / 256 B (f32) coalesced run; the three channel passes after the first hit L2.
#include "common.h"

template <typename T>
__global__ __launch_bounds__(256) void cn_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, T* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     int C, int HW, float eps, int relu) {
    __shared__ float red[4][64];
    const int px = threadIdx.x & 63, cg = threadIdx.x >> 6;
    const int n = blockIdx.y;
    const int hw = blockIdx.x * 64 + px;
    const bool ok = hw < HW;
    const T* xp = x + (size_t)n * C * HW + hw;
    float s = 0.f;
    if (ok) for (int c = cg; c < C; c += 4) s += DT<T>::ld(xp + (size_t)c * HW);
    red[cg][px] = s;
    __syncthreads();
    const float mu = (red[0][px] + red[1][px] + red[2][px] + red[3][px]) / (float)C;
    __syncthreads();
    float v = 0.f;
    if (ok) for (int c = cg; c < C; c += 4) { float d = DT<T>::ld(xp + (size_t)c * HW) - mu; v += d * d; }
    red[cg][px] = v;
    __syncthreads();
    const float var = (red[0][px] + red[1][px] + red[2][px] + red[3][px]) / (float)(C - 1);
    const float r = rsqrtf(var + eps);
    if (ok) {
        if (cg == 0) { mean_out[(size_t)n * HW + hw] = mu; rstd_out[(size_t)n * HW + hw] = r; }
        T* yp = y + (size_t)n * C * HW + hw;
        for (int c = cg; c < C; c += 4) {
            float o = gamma[c] * ((DT<T>::ld(xp + (size_t)c * HW) - mu) * r) + beta[c];
            if (relu) o = o > 0.f ? o : 0.f;
            DT<T>::st(yp + (size_t)c * HW, o);
        }
    }
}

// dx = r*(g - mean_c g) - d * (sum_c g*d) * r^3/(C-1),  g = dy*gamma (dy masked by the fused ReLU)
template <typename T>
__global__ __launch_bounds__(256) void cn_bwd_dx_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        T* __restrict__ dx, int C, int HW, int relu) {
    __shared__ float red1[4][64], red2[4][64];
    const int px = threadIdx.x & 63, cg = threadIdx.x >> 6;
    const int n = blockIdx.y;
    const int hw = blockIdx.x * 64 + px;
    const bool ok = hw < HW;
    const size_t off = (size_t)n * C * HW + hw;
    const float mu = ok ? mean[(size_t)n * HW + hw] : 0.f;
    const float r = ok ? rstd[(size_t)n * HW + hw] : 0.f;
    float s1 = 0.f, s2 = 0.f;
    if (ok) for (int c = cg; c < C; c += 4) {
        const float d = DT<T>::ld(x + off + (size_t)c * HW) - mu;
        float g = DT<T>::ld(dy + off + (size_t)c * HW);
        if (relu && !(gamma[c] * (d * r) + beta[c] > 0.f)) g = 0.f;
        g *= gamma[c];
        s1 += g; s2 += g * d;
    }
    red1[cg][px] = s1; red2[cg][px] = s2;
    __syncthreads();
    const float S1 = (red1[0][px] + red1[1][px] + red1[2][px] + red1[3][px]) / (float)C;
    const float S2 = (red2[0][px] + red2[1][px] + red2[2][px] + red2[3][px]) * r * r * r / (float)(C - 1);
    if (ok) for (int c = cg; c < C; c += 4) {
        const float d = DT<T>::ld(x + off + (size_t)c * HW) - mu;
        float g = DT<T>::ld(dy + off + (size_t)c * HW);
        if (relu && !(gamma[c] * (d * r) + beta[c] > 0.f)) g = 0.f;
        g *= gamma[c];
        DT<T>::st(dx + off + (size_t)c * HW, r * (g - S1) - d * S2);
    }
}

// per-channel partial sums: part[split][0][c] = sum dy'*xhat, part[split][1][c] = sum dy'
template <typename T>
__global__ __launch_bounds__(256) void cn_bwd_param_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           float* __restrict__ part, int N, int C, int HW, int relu,
                                                           int nsplit) {
    __shared__ float r1[4], r2[4];
    const int c = blockIdx.x, split = blockIdx.y;
    const long long total = (long long)N * HW;
    const long long per = (total + nsplit - 1) / nsplit;
    const long long lo = split * per;
    long long hi = lo + per; if (hi > total) hi = total;
    const float gm = gamma[c], bt = beta[c];
    float sg = 0.f, sb = 0.f;
    for (long long i = lo + threadIdx.x; i < hi; i += 256) {
        const int n = (int)(i / HW);
        const int hw = (int)(i - (long long)n * HW);
        const size_t off = ((size_t)n * C + c) * HW + hw;
        const float xh = (DT<T>::ld(x + off) - mean[i]) * rstd[i];
        float g = DT<T>::ld(dy + off);
        if (relu && !(gm * xh + bt > 0.f)) g = 0.f;
        sg += g * xh; sb += g;
    }
    sg = wave_sum(sg); sb = wave_sum(sb);
    if ((threadIdx.x & 63) == 0) { r1[threadIdx.x >> 6] = sg; r2[threadIdx.x >> 6] = sb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        part[((size_t)split * 2 + 0) * C + c] = r1[0] + r1[1] + r1[2] + r1[3];
        part[((size_t)split * 2 + 1) * C + c] = r2[0] + r2[1] + r2[2] + r2[3];
    }
}

__global__ void cn_bwd_param_reduce_kernel(const float* __restrict__ part, float* __restrict__ dgamma,
                                           float* __restrict__ dbeta, int C, int nsplit, int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * C) return;
    const int which = i / C, c = i - which * C;
    float s = 0.f;
    for (int sp = 0; sp < nsplit; ++sp) s += part[((size_t)sp * 2 + which) * C + c];
    float* d = which == 0 ? dgamma : dbeta;
    if (accumulate) d[c] += s; else d[c] = s;
}

extern "C" {

// x,y: [N,C,H*W] dtype; gamma,beta: [C] f32; mean,rstd: [N,H*W] f32 (saved for backward)
int hific_channelnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                          int N, int C, int HW, float eps, int relu, int dtype, hipStream_t st) {
    if (C < 2 || N <= 0 || HW <= 0) return HIFIC_ERR_ARG;
    dim3 grid(cdiv(HW, 64), N);
    if (dtype == HIFIC_F32)
        hipLaunchKernelGGL(cn_fwd_kernel<float>, grid, dim3(256), 0, st, (const float*)x, gamma, beta, (float*)y,
                           mean, rstd, C, HW, eps, relu);
    else if (dtype == HIFIC_BF16)
        hipLaunchKernelGGL(cn_fwd_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, gamma, beta, (bf16_t*)y,
                           mean, rstd, C, HW, eps, relu);
    else return HIFIC_ERR_ARG;
    return hific_launch_status();
}

// ws: at least hific_channelnorm_bwd_ws_bytes(); dgamma/dbeta f32 [C]
size_t hific_channelnorm_bwd_ws_bytes(int N, int C, int HW) {
    (void)N; (void)HW;
    return (size_t)64 * 2 * C * sizeof(float);
}

int hific_channelnorm_bwd(const void* x, const void* dy, const float* gamma, const float* beta, const float* mean,
                          const float* rstd, void* dx, float* dgamma, float* dbeta, int N, int C, int HW, int relu,
                          int accumulate, int dtype, void* ws, size_t ws_bytes, hipStream_t st) {
    if (C < 2 || N <= 0 || HW <= 0) return HIFIC_ERR_ARG;
    long long total = (long long)N * HW;
    int nsplit = cdiv(1024, C);
    if (nsplit > 64) nsplit = 64;
    if ((long long)nsplit * 256 > total) nsplit = (int)((total + 255) / 256);
    if (nsplit < 1) nsplit = 1;
    if ((size_t)nsplit * 2 * C * sizeof(float) > ws_bytes) return HIFIC_ERR_WS;
    float* part = (float*)ws;
    dim3 grid(cdiv(HW, 64), N), pgrid(C, nsplit);
    if (dtype == HIFIC_F32) {
        hipLaunchKernelGGL(cn_bwd_dx_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (const float*)dy, gamma,
                           beta, mean, rstd, (float*)dx, C, HW, relu);
        hipLaunchKernelGGL(cn_bwd_param_kernel<float>, pgrid, dim3(256), 0, st, (const float*)x, (const float*)dy,
                           gamma, beta, mean, rstd, part, N, C, HW, relu, nsplit);
    } else if (dtype == HIFIC_BF16) {
        hipLaunchKernelGGL(cn_bwd_dx_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, gamma,
                           beta, mean, rstd, (bf16_t*)dx, C, HW, relu);
        hipLaunchKernelGGL(cn_bwd_param_kernel<bf16_t>, pgrid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy,
                           gamma, beta, mean, rstd, part, N, C, HW, relu, nsplit);
    } else return HIFIC_ERR_ARG;
    hipLaunchKernelGGL(cn_bwd_param_reduce_kernel, dim3(cdiv(2 * C, 256)), dim3(256), 0, st, part, dgamma, dbeta, C,
                       nsplit, accumulate);
    return hific_launch_status();
}

}  // extern "C"
