// Hyperprior entropy-model arithmetic (all f32, elementwise / per-channel, HBM-bound and tiny):
//   * factorised hyperlatent density  (reference src/compression/hyperprior_model.py:305-326,349-384)
//   * Gaussian / logistic latent likelihood (src/hyperprior.py:124-139, src/helpers/maths.py:102-109)
//   * LowerBoundToward (src/helpers/maths.py:87-100), rounding about a mean (src/hyperprior.py:68-74),
//   * entropy estimate sum(log(p+1e-9)) (src/hyperprior.py:80-93)
#include "common.h"

#define EW_GRID(total) dim3((unsigned)((((total) + 255) / 256) > 16384 ? 16384 : (((total) + 255) / 256)))
#define EW_LOOP(i, total) \
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (total); i += (long long)gridDim.x * blockDim.x)

__device__ __forceinline__ float block_sum_256e(float v, float* sh) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return r;
}

__device__ __forceinline__ float softplus_t(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoid_t(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float sign_t(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }

// ---- generic pieces ---------------------------------------------------------------------------------
__global__ void round_kernel(const float* __restrict__ x, const float* __restrict__ mean, float* __restrict__ o,
                             long long n) {
    EW_LOOP(i, n) {
        if (mean) { const float m = mean[i]; o[i] = floorf((x[i] - m) + 0.5f) + m; }
        else o[i] = floorf(x[i] + 0.5f);
    }
}
__global__ void lower_bound_fwd_kernel(const float* __restrict__ x, float bound, float* __restrict__ o, long long n) {
    EW_LOOP(i, n) o[i] = fmaxf(x[i], bound);
}
__global__ void lower_bound_bwd_kernel(const float* __restrict__ x, const float* __restrict__ dy, float bound,
                                       float* __restrict__ dx, long long n) {
    EW_LOOP(i, n) { const float g = dy[i]; dx[i] = (x[i] >= bound || g < 0.f) ? g : 0.f; }
}
__global__ __launch_bounds__(256) void logsum_partial_kernel(const float* __restrict__ p, float eps,
                                                             float* __restrict__ part, long long n, unsigned* ticket, float tk_mul, float* tk_out) {
    __shared__ float sh[4];
    float s = 0.f;
    EW_LOOP(i, n) s += logf(p[i] + eps);
    s = block_sum_256e(s, sh);
    if (threadIdx.x == 0) hific_st_agent(part + blockIdx.x, s);
    if (ticket && hific_last_block(ticket, gridDim.x)) {          // final_sum_kernel_e's sums by the last-arriving workgroup
        float t = 0.f;
        for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) t += hific_ld_agent(part + i);
        t = block_sum_256e(t, sh);
        if (threadIdx.x == 0) *tk_out = t * tk_mul;
    }
}
__global__ __launch_bounds__(256) void final_sum_kernel_e(const float* __restrict__ part, int n, float mul,
                                                          float* __restrict__ out) {
    __shared__ float sh[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += part[i];
    s = block_sum_256e(s, sh);
    if (threadIdx.x == 0) *out = s * mul;
}
__global__ void logsum_bwd_kernel(const float* __restrict__ p, const float* __restrict__ g, float eps, float mul,
                                  float* __restrict__ dp, long long n, int accumulate) {
    const float gg = *g * mul;
    EW_LOOP(i, n) { const float v = gg / (p[i] + eps); if (accumulate) dp[i] += v; else dp[i] = v; }
}

// ---- Gaussian / logistic latent likelihood ------------------------------------------------------------
__device__ __forceinline__ float std_cdf(float t, int logistic) {
    return logistic ? sigmoid_t(t) : 0.5f * erfcf(t * -0.70710678118654752440f);
}
__device__ __forceinline__ float std_pdf(float t, int logistic) {
    if (logistic) { const float s = sigmoid_t(-fabsf(t)); return s * (1.f - s); }   // small-side sigmoid: no 1-s cancellation
    return 0.39894228040143267794f * expf(-0.5f * t * t);
}
__global__ void gauss_lik_fwd_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                     const float* __restrict__ scale, float* __restrict__ lik, long long n,
                                     float min_lik, int logistic) {
    EW_LOOP(i, n) {
        const float a = fabsf(x[i] - mean[i]);
        const float s = scale[i];
        const float p = std_cdf((0.5f - a) / s, logistic) - std_cdf(-(0.5f + a) / s, logistic);
        lik[i] = fmaxf(p, min_lik);
    }
}
__global__ void gauss_lik_bwd_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                     const float* __restrict__ scale, const float* __restrict__ dlik,
                                     float* __restrict__ dx, float* __restrict__ dmean, float* __restrict__ dscale,
                                     long long n, float min_lik, int logistic, int acc_mean, int acc_scale) {
    EW_LOOP(i, n) {
        const float d = x[i] - mean[i];
        const float a = fabsf(d);
        const float s = scale[i];
        const float t1 = (0.5f - a) / s, t2 = (0.5f + a) / s;
        const float p = std_cdf(t1, logistic) - std_cdf(-t2, logistic);
        float g = dlik[i];
        if (!(p >= min_lik || g < 0.f)) g = 0.f;
        const float f1 = std_pdf(t1, logistic), f2 = std_pdf(t2, logistic);
        const float dpa = (f2 - f1) / s;
        const float dps = -(f1 * (0.5f - a) + f2 * (0.5f + a)) / (s * s);
        const float gx = g * dpa * sign_t(d);
        if (dx) dx[i] = gx;
        if (dmean) { if (acc_mean) dmean[i] -= gx; else dmean[i] = -gx; }
        if (dscale) { if (acc_scale) dscale[i] += g * dps; else dscale[i] = g * dps; }
    }
}

// ---- factorised prior ---------------------------------------------------------------------------------
struct FpPtrs { const float* H[4]; const float* a[4]; const float* b[4]; };
struct FpGrads { float* H[4]; float* a[4]; float* b[4]; };
// per-channel parameter block, transformed: sp(H) (24), tanh(a) (10), b (10)  = 44 floats
// layout: [H0(3) | H1(9) | H2(9) | H3(3) | a0(3) a1(3) a2(3) a3(1) | b0(3) b1(3) b2(3) b3(1)]
#define FP_NP 44
__device__ __forceinline__ void fp_load(const FpPtrs& P, int c, float* q) {
    for (int i = 0; i < 3; ++i) q[i] = softplus_t(P.H[0][c * 3 + i]);
    for (int i = 0; i < 9; ++i) q[3 + i] = softplus_t(P.H[1][c * 9 + i]);
    for (int i = 0; i < 9; ++i) q[12 + i] = softplus_t(P.H[2][c * 9 + i]);
    for (int i = 0; i < 3; ++i) q[21 + i] = softplus_t(P.H[3][c * 3 + i]);
    for (int i = 0; i < 3; ++i) q[24 + i] = tanhf(P.a[0][c * 3 + i]);
    for (int i = 0; i < 3; ++i) q[27 + i] = tanhf(P.a[1][c * 3 + i]);
    for (int i = 0; i < 3; ++i) q[30 + i] = tanhf(P.a[2][c * 3 + i]);
    q[33] = tanhf(P.a[3][c]);
    for (int i = 0; i < 3; ++i) q[34 + i] = P.b[0][c * 3 + i];
    for (int i = 0; i < 3; ++i) q[37 + i] = P.b[1][c * 3 + i];
    for (int i = 0; i < 3; ++i) q[40 + i] = P.b[2][c * 3 + i];
    q[43] = P.b[3][c];
}
// forward logits; z* keep the pre-gate values (after the affine map, before + tanh(a) tanh(.))
__device__ __forceinline__ float fp_logits(const float* q, float v, float* z0, float* z1, float* z2, float* z3) {
    float h0[3], h1[3], h2[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { z0[i] = q[i] * v + q[34 + i]; h0[i] = z0[i] + q[24 + i] * tanhf(z0[i]); }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        z1[i] = q[3 + i * 3] * h0[0] + q[4 + i * 3] * h0[1] + q[5 + i * 3] * h0[2] + q[37 + i];
        h1[i] = z1[i] + q[27 + i] * tanhf(z1[i]);
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        z2[i] = q[12 + i * 3] * h1[0] + q[13 + i * 3] * h1[1] + q[14 + i * 3] * h1[2] + q[40 + i];
        h2[i] = z2[i] + q[30 + i] * tanhf(z2[i]);
    }
    *z3 = q[21] * h2[0] + q[22] * h2[1] + q[23] * h2[2] + q[43];
    return *z3 + q[33] * tanhf(*z3);
}
// backward of one logits evaluation: accumulates d(transformed params) into dq, returns d/dv
__device__ __forceinline__ float fp_logits_bwd(const float* q, float v, float dout, float* dq) {
    float z0[3], z1[3], z2[3], z3;
    fp_logits(q, v, z0, z1, z2, &z3);
    float h0[3], h1[3], h2[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        h0[i] = z0[i] + q[24 + i] * tanhf(z0[i]);
        h1[i] = z1[i] + q[27 + i] * tanhf(z1[i]);
        h2[i] = z2[i] + q[30 + i] * tanhf(z2[i]);
    }
    // layer 3
    float t3 = tanhf(z3);
    dq[33] += dout * t3;
    float dz3 = dout * (1.f + q[33] * (1.f - t3 * t3));
    dq[43] += dz3;
    float dh2[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { dq[21 + j] += dz3 * h2[j]; dh2[j] = dz3 * q[21 + j]; }
    // layer 2
    float dh1[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float t = tanhf(z2[i]);
        dq[30 + i] += dh2[i] * t;
        const float dz = dh2[i] * (1.f + q[30 + i] * (1.f - t * t));
        dq[40 + i] += dz;
#pragma unroll
        for (int j = 0; j < 3; ++j) { dq[12 + i * 3 + j] += dz * h1[j]; dh1[j] += dz * q[12 + i * 3 + j]; }
    }
    // layer 1
    float dh0[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float t = tanhf(z1[i]);
        dq[27 + i] += dh1[i] * t;
        const float dz = dh1[i] * (1.f + q[27 + i] * (1.f - t * t));
        dq[37 + i] += dz;
#pragma unroll
        for (int j = 0; j < 3; ++j) { dq[3 + i * 3 + j] += dz * h0[j]; dh0[j] += dz * q[3 + i * 3 + j]; }
    }
    // layer 0
    float dv = 0.f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float t = tanhf(z0[i]);
        dq[24 + i] += dh0[i] * t;
        const float dz = dh0[i] * (1.f + q[24 + i] * (1.f - t * t));
        dq[34 + i] += dz;
        dq[i] += dz * v;
        dv += dz * q[i];
    }
    return dv;
}

__global__ __launch_bounds__(256) void fp_lik_fwd_kernel(const float* __restrict__ x, FpPtrs P, float* __restrict__ lik,
                                                         int N, int C, int HW, float min_lik) {
    const int c = blockIdx.x;
    float q[FP_NP];
    fp_load(P, c, q);
    const long long total = (long long)N * HW;
    for (long long e = (long long)blockIdx.y * 256 + threadIdx.x; e < total; e += (long long)gridDim.y * 256) {
        const int n = (int)(e / HW);
        const int hw = (int)(e - (long long)n * HW);
        const size_t idx = ((size_t)n * C + c) * HW + hw;
        const float v = x[idx];
        float z0[3], z1[3], z2[3], z3;
        const float up = fp_logits(q, v + 0.5f, z0, z1, z2, &z3);
        const float lo = fp_logits(q, v - 0.5f, z0, z1, z2, &z3);
        const float sg = -sign_t(up + lo);
        const float p = fabsf(sigmoid_t(sg * up) - sigmoid_t(sg * lo));
        lik[idx] = fmaxf(p, min_lik);
    }
}

// part[split][c][44]: grads wrt the transformed parameters
__global__ __launch_bounds__(256) void fp_lik_bwd_kernel(const float* __restrict__ x, FpPtrs P,
                                                         const float* __restrict__ dlik, float* __restrict__ dx,
                                                         float* __restrict__ part, int N, int C, int HW, float min_lik) {
    __shared__ float sh[4][FP_NP];
    const int c = blockIdx.x;
    float q[FP_NP], dq[FP_NP];
    fp_load(P, c, q);
#pragma unroll
    for (int i = 0; i < FP_NP; ++i) dq[i] = 0.f;
    const long long total = (long long)N * HW;
    for (long long e = (long long)blockIdx.y * 256 + threadIdx.x; e < total; e += (long long)gridDim.y * 256) {
        const int n = (int)(e / HW);
        const int hw = (int)(e - (long long)n * HW);
        const size_t idx = ((size_t)n * C + c) * HW + hw;
        const float v = x[idx];
        float z0[3], z1[3], z2[3], z3;
        const float up = fp_logits(q, v + 0.5f, z0, z1, z2, &z3);
        const float lo = fp_logits(q, v - 0.5f, z0, z1, z2, &z3);
        const float sg = -sign_t(up + lo);
        const float su = sigmoid_t(sg * up), sl = sigmoid_t(sg * lo);
        const float diff = su - sl;
        const float p = fabsf(diff);
        float g = dlik[idx];
        if (!(p >= min_lik || g < 0.f)) g = 0.f;
        const float gs = g * sign_t(diff) * sg;
        const float dup = gs * su * (1.f - su);
        const float dlo = -gs * sl * (1.f - sl);
        float dv = fp_logits_bwd(q, v + 0.5f, dup, dq);
        dv += fp_logits_bwd(q, v - 0.5f, dlo, dq);
        if (dx) dx[idx] = dv;
    }
#pragma unroll
    for (int i = 0; i < FP_NP; ++i) {
        const float s = wave_sum(dq[i]);
        if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < FP_NP)
        part[((size_t)blockIdx.y * C + c) * FP_NP + threadIdx.x] =
            sh[0][threadIdx.x] + sh[1][threadIdx.x] + sh[2][threadIdx.x] + sh[3][threadIdx.x];
}

// chain through softplus / tanh and scatter into the 12 parameter-gradient tensors
__global__ void fp_param_reduce_kernel(const float* __restrict__ part, FpPtrs P, FpGrads G, int C, int nsplit,
                                       int accumulate) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= C * FP_NP) return;
    const int c = i / FP_NP, k = i - c * FP_NP;
    float s = 0.f;
    for (int sp = 0; sp < nsplit; ++sp) s += part[((size_t)sp * C + c) * FP_NP + k];
    float* dst; float raw; int kind;   // kind 0: softplus, 1: tanh, 2: identity
    if (k < 3)       { dst = G.H[0] + c * 3 + k;        raw = P.H[0][c * 3 + k];        kind = 0; }
    else if (k < 12) { dst = G.H[1] + c * 9 + (k - 3);  raw = P.H[1][c * 9 + (k - 3)];  kind = 0; }
    else if (k < 21) { dst = G.H[2] + c * 9 + (k - 12); raw = P.H[2][c * 9 + (k - 12)]; kind = 0; }
    else if (k < 24) { dst = G.H[3] + c * 3 + (k - 21); raw = P.H[3][c * 3 + (k - 21)]; kind = 0; }
    else if (k < 27) { dst = G.a[0] + c * 3 + (k - 24); raw = P.a[0][c * 3 + (k - 24)]; kind = 1; }
    else if (k < 30) { dst = G.a[1] + c * 3 + (k - 27); raw = P.a[1][c * 3 + (k - 27)]; kind = 1; }
    else if (k < 33) { dst = G.a[2] + c * 3 + (k - 30); raw = P.a[2][c * 3 + (k - 30)]; kind = 1; }
    else if (k < 34) { dst = G.a[3] + c;                raw = P.a[3][c];                kind = 1; }
    else if (k < 37) { dst = G.b[0] + c * 3 + (k - 34); raw = 0.f; kind = 2; }
    else if (k < 40) { dst = G.b[1] + c * 3 + (k - 37); raw = 0.f; kind = 2; }
    else if (k < 43) { dst = G.b[2] + c * 3 + (k - 40); raw = 0.f; kind = 2; }
    else             { dst = G.b[3] + c;                raw = 0.f; kind = 2; }
    if (kind == 0) s *= (raw > 20.f ? 1.f : sigmoid_t(raw));
    else if (kind == 1) { const float t = tanhf(raw); s *= (1.f - t * t); }
    if (accumulate) *dst += s; else *dst = s;
}

// ---- EVALUATION path, device half of `compress`: int32 symbols + table indices --------------
// prior_model.py:148-156 (compute_indices: the table entry of each predicted scale = n_table-1 minus the number of
// table[:-1] entries >= the lower-bounded scale) and :180-181 (symbols = floor(y + 0.5 - mean)), fused; the rANS
// coder on the host then needs only these two int32 tensors instead of three float tensors.
__global__ __launch_bounds__(256) void prior_symbols_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ table, int n_table,
                                                            float scales_min, int* __restrict__ symbols,
                                                            int* __restrict__ indices, long long n) {
    __shared__ float tab[256];
    for (int t = threadIdx.x; t < n_table; t += 256) tab[t] = table[t];
    __syncthreads();
    EW_LOOP(i, n) {
        const float s = fmaxf(scale[i], scales_min);               // LowerBoundToward forward (maths.py:87-95)
        int idx = n_table - 1;
        for (int t = 0; t < n_table - 1; ++t) idx -= (s <= tab[t]) ? 1 : 0;
        indices[i] = idx;
        symbols[i] = (int)floorf((x[i] + 0.5f) - mean[i]);
    }
}
// hyperprior_model.py:135-139 (indices = channel) and :169 (symbols = floor(z + 0.5))
__global__ __launch_bounds__(256) void hyper_symbols_kernel(const float* __restrict__ z, int* __restrict__ symbols,
                                                            int* __restrict__ indices, int C, int HW, long long n) {
    EW_LOOP(i, n) {
        symbols[i] = (int)floorf(z[i] + 0.5f);
        indices[i] = (int)((i / HW) % C);
    }
}

extern "C" {

int hific_prior_symbols(const float* x, const float* mean, const float* scale, const float* table, int n_table,
                        float scales_min, int* symbols, int* indices, long long n, hipStream_t st) {
    if (n_table < 1 || n_table > 256 || n < 0) return HIFIC_ERR_ARG;
    if (n == 0) return HIFIC_OK;
    hipLaunchKernelGGL(prior_symbols_kernel, EW_GRID(n), dim3(256), 0, st, x, mean, scale, table, n_table, scales_min,
                       symbols, indices, n);
    return hific_launch_status();
}
int hific_hyper_symbols(const float* z, int* symbols, int* indices, int N, int C, int HW, hipStream_t st) {
    if (N < 0 || C <= 0 || HW <= 0) return HIFIC_ERR_ARG;
    const long long n = (long long)N * C * HW;
    if (n == 0) return HIFIC_OK;
    hipLaunchKernelGGL(hyper_symbols_kernel, EW_GRID(n), dim3(256), 0, st, z, symbols, indices, C, HW, n);
    return hific_launch_status();
}

int hific_round_f32(const float* x, const float* mean, float* o, long long n, hipStream_t st) {
    hipLaunchKernelGGL(round_kernel, EW_GRID(n), dim3(256), 0, st, x, mean, o, n);
    return hific_launch_status();
}
int hific_lower_bound_fwd(const float* x, float bound, float* o, long long n, hipStream_t st) {
    hipLaunchKernelGGL(lower_bound_fwd_kernel, EW_GRID(n), dim3(256), 0, st, x, bound, o, n);
    return hific_launch_status();
}
int hific_lower_bound_bwd(const float* x, const float* dy, float bound, float* dx, long long n, hipStream_t st) {
    hipLaunchKernelGGL(lower_bound_bwd_kernel, EW_GRID(n), dim3(256), 0, st, x, dy, bound, dx, n);
    return hific_launch_status();
}
// out[0] = mul * sum(log(p + eps)); ws >= 256 floats
int hific_logsum_fwd(const float* p, float* out, long long n, float eps, float mul, void* ws, size_t ws_bytes,
                     hipStream_t st) {
    const int nb = 256;
    if (ws_bytes < nb * sizeof(float)) return HIFIC_ERR_WS;
    float* part = (float*)ws;
    unsigned* tk = hific_tickets(st, 1);
    hipLaunchKernelGGL(logsum_partial_kernel, dim3(nb), dim3(256), 0, st, p, eps, part, n, tk, mul, out);
    if (!tk) hipLaunchKernelGGL(final_sum_kernel_e, dim3(1), dim3(256), 0, st, part, nb, mul, out);
    return hific_launch_status();
}
// dp (=|+=) (*g) * mul / (p + eps)
int hific_logsum_bwd(const float* p, const float* g, float* dp, long long n, float eps, float mul, int accumulate,
                     hipStream_t st) {
    hipLaunchKernelGGL(logsum_bwd_kernel, EW_GRID(n), dim3(256), 0, st, p, g, eps, mul, dp, n, accumulate);
    return hific_launch_status();
}
int hific_gauss_lik_fwd(const float* x, const float* mean, const float* scale, float* lik, long long n, float min_lik,
                        int logistic, hipStream_t st) {
    hipLaunchKernelGGL(gauss_lik_fwd_kernel, EW_GRID(n), dim3(256), 0, st, x, mean, scale, lik, n, min_lik, logistic);
    return hific_launch_status();
}
// any of dx/dmean/dscale may be null; acc_* selects += for dmean/dscale
int hific_gauss_lik_bwd(const float* x, const float* mean, const float* scale, const float* dlik, float* dx,
                        float* dmean, float* dscale, long long n, float min_lik, int logistic, int acc_mean,
                        int acc_scale, hipStream_t st) {
    hipLaunchKernelGGL(gauss_lik_bwd_kernel, EW_GRID(n), dim3(256), 0, st, x, mean, scale, dlik, dx, dmean, dscale, n,
                       min_lik, logistic, acc_mean, acc_scale);
    return hific_launch_status();
}

// params: 12 pointers in the order H_0..H_3, a_0..a_3, b_0..b_3 (reference parameter names)
int hific_factorized_lik_fwd(const float* x, const float* const* params, float* lik, int N, int C, int HW,
                             float min_lik, hipStream_t st) {
    FpPtrs P;
    for (int k = 0; k < 4; ++k) { P.H[k] = params[k]; P.a[k] = params[4 + k]; P.b[k] = params[8 + k]; }
    long long total = (long long)N * HW;
    int ny = (int)((total + 255) / 256); if (ny > 64) ny = 64;
    hipLaunchKernelGGL(fp_lik_fwd_kernel, dim3(C, ny), dim3(256), 0, st, x, P, lik, N, C, HW, min_lik);
    return hific_launch_status();
}
// dparams: 12 gradient pointers (same order); dx may be null; ws >= 64*C*44 floats
int hific_factorized_lik_bwd(const float* x, const float* const* params, const float* dlik, float* dx,
                             float* const* dparams, int N, int C, int HW, float min_lik, int accumulate, void* ws,
                             size_t ws_bytes, hipStream_t st) {
    FpPtrs P; FpGrads G;
    for (int k = 0; k < 4; ++k) {
        P.H[k] = params[k]; P.a[k] = params[4 + k]; P.b[k] = params[8 + k];
        G.H[k] = dparams[k]; G.a[k] = dparams[4 + k]; G.b[k] = dparams[8 + k];
    }
    long long total = (long long)N * HW;
    int ny = (int)((total + 255) / 256); if (ny > 64) ny = 64;
    if ((size_t)ny * C * FP_NP * sizeof(float) > ws_bytes) return HIFIC_ERR_WS;
    float* part = (float*)ws;
    hipLaunchKernelGGL(fp_lik_bwd_kernel, dim3(C, ny), dim3(256), 0, st, x, P, dlik, dx, part, N, C, HW, min_lik);
    hipLaunchKernelGGL(fp_param_reduce_kernel, dim3(cdiv(C * FP_NP, 256)), dim3(256), 0, st, part, P, G, C, ny,
                       accumulate);
    return hific_launch_status();
}

}  // extern "C"
