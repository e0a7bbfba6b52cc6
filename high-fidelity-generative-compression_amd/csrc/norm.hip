// ChannelNorm2D (reference: src/normalisation/channel.py:29-59) forward/backward on NCHW tensors.
// Per pixel (n,h,w): mu = mean_c x, var = sum_c (x-mu)^2/(C-1) (unbiased), r = rsqrt(var+eps),
// y = gamma_c*(x-mu)*r + beta_c, optionally fused with the ReLU that follows it in the Encoder/Generator.
// HBM-bound: a workgroup owns 64 consecutive pixels x all C channels; every channel row it touches is a
// 128 B (bf16) / 256 B (f32) coalesced run; the three channel passes after the first hit L2.
#include "common.h"

// Workgroup = 64 consecutive pixels x all C channels, NW waves: wave w owns channels w, w+NW, ...  Loads are issued
// 8 at a time per thread (clamped index + select instead of branches) so that the three channel passes are
// bandwidth- rather than latency-bound; NW is chosen from C so small planes (16x16 x 960 ch) still fill the chip.
#define CN_U 8
template <typename T, int NW>
__global__ __launch_bounds__(NW * 64) void cn_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, T* __restrict__ y,
                                                         float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                         int C, int HW, float eps, int relu) {
    __shared__ float red[NW][64];
    const int px = threadIdx.x & 63, cg = threadIdx.x >> 6;
    const int n = blockIdx.y;
    const int hw0 = blockIdx.x * 64 + px;
    const bool ok = hw0 < HW;
    const int hw = ok ? hw0 : HW - 1;
    const T* xp = x + (size_t)n * C * HW + hw;
    float s = 0.f;
    for (int c = cg; c < C; c += NW * CN_U) {
        float v[CN_U];
#pragma unroll
        for (int k = 0; k < CN_U; ++k) { const int cc = c + k * NW; v[k] = DT<T>::ld(xp + (size_t)(cc < C ? cc : C - 1) * HW); }
#pragma unroll
        for (int k = 0; k < CN_U; ++k) s += (c + k * NW < C) ? v[k] : 0.f;
    }
    red[cg][px] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) tot += red[w][px];
    const float mu = tot / (float)C;
    __syncthreads();
    float q = 0.f;
    for (int c = cg; c < C; c += NW * CN_U) {
        float v[CN_U];
#pragma unroll
        for (int k = 0; k < CN_U; ++k) { const int cc = c + k * NW; v[k] = DT<T>::ld(xp + (size_t)(cc < C ? cc : C - 1) * HW); }
#pragma unroll
        for (int k = 0; k < CN_U; ++k) { const float d = v[k] - mu; q += (c + k * NW < C) ? d * d : 0.f; }
    }
    red[cg][px] = q;
    __syncthreads();
    float tq = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) tq += red[w][px];
    const float var = tq / (float)(C - 1);
    const float r = rsqrtf(var + eps);
    if (ok && cg == 0) { mean_out[(size_t)n * HW + hw] = mu; rstd_out[(size_t)n * HW + hw] = r; }
    T* yp = y + (size_t)n * C * HW + hw;
    for (int c = cg; c < C; c += NW * CN_U) {
        float v[CN_U];
#pragma unroll
        for (int k = 0; k < CN_U; ++k) { const int cc = c + k * NW; v[k] = DT<T>::ld(xp + (size_t)(cc < C ? cc : C - 1) * HW); }
#pragma unroll
        for (int k = 0; k < CN_U; ++k) {
            const int cc = c + k * NW;
            if (ok && cc < C) {
                float o = gamma[cc] * ((v[k] - mu) * r) + beta[cc];
                if (relu) o = o > 0.f ? o : 0.f;
                DT<T>::st(yp + (size_t)cc * HW, o);
            }
        }
    }
}

// dx = r*(g - mean_c g) - d * (sum_c g*d) * r^3/(C-1),  g = dy*gamma (dy masked by the fused ReLU)
template <typename T, int NW>
__global__ __launch_bounds__(NW * 64) void cn_bwd_dx_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ mean, const float* __restrict__ rstd,
                                                            T* __restrict__ dx, int C, int HW, int relu) {
    __shared__ float red1[NW][64], red2[NW][64];
    const int px = threadIdx.x & 63, cg = threadIdx.x >> 6;
    const int n = blockIdx.y;
    const int hw0 = blockIdx.x * 64 + px;
    const bool ok = hw0 < HW;
    const int hw = ok ? hw0 : HW - 1;
    const size_t off = (size_t)n * C * HW + hw;
    const float mu = mean[(size_t)n * HW + hw];
    const float r = rstd[(size_t)n * HW + hw];
    float s1 = 0.f, s2 = 0.f;
    for (int c = cg; c < C; c += NW * CN_U) {
        float xv[CN_U], gv[CN_U];
#pragma unroll
        for (int k = 0; k < CN_U; ++k) {
            const int cc = c + k * NW; const size_t o = off + (size_t)(cc < C ? cc : C - 1) * HW;
            xv[k] = DT<T>::ld(x + o); gv[k] = DT<T>::ld(dy + o);
        }
#pragma unroll
        for (int k = 0; k < CN_U; ++k) {
            const int cc = c + k * NW; const int ci = cc < C ? cc : C - 1;
            const float d = xv[k] - mu;
            float g = gv[k];
            if (relu && !(gamma[ci] * (d * r) + beta[ci] > 0.f)) g = 0.f;
            g *= gamma[ci];
            if (cc < C) { s1 += g; s2 += g * d; }
        }
    }
    red1[cg][px] = s1; red2[cg][px] = s2;
    __syncthreads();
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) { t1 += red1[w][px]; t2 += red2[w][px]; }
    const float S1 = t1 / (float)C;
    const float S2 = t2 * r * r * r / (float)(C - 1);
    for (int c = cg; c < C; c += NW * CN_U) {
        float xv[CN_U], gv[CN_U];
#pragma unroll
        for (int k = 0; k < CN_U; ++k) {
            const int cc = c + k * NW; const size_t o = off + (size_t)(cc < C ? cc : C - 1) * HW;
            xv[k] = DT<T>::ld(x + o); gv[k] = DT<T>::ld(dy + o);
        }
#pragma unroll
        for (int k = 0; k < CN_U; ++k) {
            const int cc = c + k * NW;
            if (ok && cc < C) {
                const float d = xv[k] - mu;
                float g = gv[k];
                if (relu && !(gamma[cc] * (d * r) + beta[cc] > 0.f)) g = 0.f;
                g *= gamma[cc];
                DT<T>::st(dx + off + (size_t)cc * HW, r * (g - S1) - d * S2);
            }
        }
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
    const int per = (HW + nsplit - 1) / nsplit;
    const int lo = split * per;
    int hi = lo + per; if (hi > HW) hi = HW;
    const float gm = gamma[c], bt = beta[c];
    float sg = 0.f, sb = 0.f;
    for (int n = 0; n < N; ++n) {
        const size_t off = ((size_t)n * C + c) * HW;
        const float* mp = mean + (size_t)n * HW;
        const float* rp = rstd + (size_t)n * HW;
        for (int i = lo + threadIdx.x; i < hi; i += 256) {
            const float xh = (DT<T>::ld(x + off + i) - mp[i]) * rp[i];
            float g = DT<T>::ld(dy + off + i);
            if (relu && !(gm * xh + bt > 0.f)) g = 0.f;
            sg += g * xh; sb += g;
        }
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
    const int nw = C >= 480 ? 16 : (C >= 200 ? 8 : 4);
#define CN_FWD(TT, NWV) hipLaunchKernelGGL((cn_fwd_kernel<TT, NWV>), grid, dim3(NWV * 64), 0, st, (const TT*)x, gamma, \
                                           beta, (TT*)y, mean, rstd, C, HW, eps, relu)
    if (dtype == HIFIC_F32) { if (nw == 16) CN_FWD(float, 16); else if (nw == 8) CN_FWD(float, 8); else CN_FWD(float, 4); }
    else if (dtype == HIFIC_BF16) { if (nw == 16) CN_FWD(bf16_t, 16); else if (nw == 8) CN_FWD(bf16_t, 8); else CN_FWD(bf16_t, 4); }
    else return HIFIC_ERR_ARG;
#undef CN_FWD
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
    int nsplit = cdiv(1024, C);
    if (nsplit > 64) nsplit = 64;
    if (nsplit * 256 > HW) nsplit = cdiv(HW, 256);
    if (nsplit < 1) nsplit = 1;
    if ((size_t)nsplit * 2 * C * sizeof(float) > ws_bytes) return HIFIC_ERR_WS;
    float* part = (float*)ws;
    dim3 grid(cdiv(HW, 64), N), pgrid(C, nsplit);
    const int nw = C >= 480 ? 16 : (C >= 200 ? 8 : 4);
#define CN_BWD(TT, NWV) hipLaunchKernelGGL((cn_bwd_dx_kernel<TT, NWV>), grid, dim3(NWV * 64), 0, st, (const TT*)x, \
                                           (const TT*)dy, gamma, beta, mean, rstd, (TT*)dx, C, HW, relu)
    if (dtype == HIFIC_F32) {
        if (nw == 16) CN_BWD(float, 16); else if (nw == 8) CN_BWD(float, 8); else CN_BWD(float, 4);
        hipLaunchKernelGGL(cn_bwd_param_kernel<float>, pgrid, dim3(256), 0, st, (const float*)x, (const float*)dy,
                           gamma, beta, mean, rstd, part, N, C, HW, relu, nsplit);
    } else if (dtype == HIFIC_BF16) {
        if (nw == 16) CN_BWD(bf16_t, 16); else if (nw == 8) CN_BWD(bf16_t, 8); else CN_BWD(bf16_t, 4);
        hipLaunchKernelGGL(cn_bwd_param_kernel<bf16_t>, pgrid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy,
                           gamma, beta, mean, rstd, part, N, C, HW, relu, nsplit);
    } else return HIFIC_ERR_ARG;
#undef CN_BWD
    hipLaunchKernelGGL(cn_bwd_param_reduce_kernel, dim3(cdiv(2 * C, 256)), dim3(256), 0, st, part, dgamma, dbeta, C,
                       nsplit, accumulate);
    return hific_launch_status();
}

}  // extern "C"
