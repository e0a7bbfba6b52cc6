// ChannelNorm2D (reference: src/normalisation/channel.py:29-59) forward/backward on NCHW tensors.
// Per pixel (n,h,w): mu = mean_c x, var = sum_c (x-mu)^2/(C-1) (unbiased), r = rsqrt(var+eps),
// y = gamma_c*(x-mu)*r + beta_c, optionally fused with the ReLU that follows it in the Encoder/Generator.
// HBM-bound: a workgroup owns 64 consecutive pixels x all C channels; every channel row it touches is a
// 128 B (bf16) / 256 B (f32) coalesced run; the three channel passes after the first hit L2.
#include "common.h"
#include <stdlib.h>
#include <type_traits>

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

// ---------------------------------------------------------------------------------------------------
// Register-resident variants (the ones the C-ABI dispatches to whenever a configuration fits).
// A workgroup owns PXB consecutive pixels x all C channels.  Lane = (pixel px = lane % PXB, channel sub-group
// lane / PXB); channel group g = wave * (64/PXB) + sub of G = NW * 64/PXB groups owns channels g, g+G, ... (at most
// CPT of them), which the thread keeps in registers: x (and dy) are read from memory exactly once, with all CPT
// (2 CPT) loads of a thread in flight together, instead of three latency-bound passes.  PXB < 64 trades the length
// of the coalesced runs (PXB * 2 bytes) for workgroups: a 16x16 plane with 960 channels and batch 16 gives 256
// workgroups at PXB = 16 instead of 64 at PXB = 64.
// The backward kernel also produces the gamma/beta gradient partials (sum over its pixels, via lane shuffles), so x
// and dy are not read a second time for the parameter gradients.
// ---------------------------------------------------------------------------------------------------
template <int PXB>
__device__ __forceinline__ float cn_sum_subs(float v) {       // sum over the channel sub-groups of a wave (same pixel)
#pragma unroll
    for (int o = PXB; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <int PXB>
__device__ __forceinline__ float cn_sum_px(float v) {         // sum over the PXB pixels of one channel sub-group
#pragma unroll
    for (int o = PXB >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Workgroups are dealt round-robin to the 8 XCDs (own L2 each).  With PXB < 64 the 64/PXB workgroups that share every
// 128-byte line of a channel row would land on different XCDs and each fetch the line from MALL/HBM (measured: the
// 16x16x960 backward ran at the MALL rate of an 8x over-fetch).  Remap so that consecutive pixel groups share an XCD.
__device__ __forceinline__ void cn_block_remap(int& bx, int& by, int remap) {
    const unsigned total = gridDim.x * gridDim.y;
    unsigned bid = blockIdx.y * gridDim.x + blockIdx.x;
    if (remap && (total & 7u) == 0) bid = (bid & 7u) * (total >> 3) + (bid >> 3);
    by = (int)(bid / gridDim.x); bx = (int)(bid - (unsigned)by * gridDim.x);
}

// RES: y = act(norm(x)) + resid - the residual add of a ResidualBlock (generator.py:44: `torch.add(res, identity_map)`)
// folded into the block's second norm (one 8 MB read-modify-write pass and a launch less per block and forward).
template <typename T, int PXB, int NW, int CPT, bool RES = false>
__global__ __launch_bounds__(NW * 64) void cn_fwd_reg_kernel(const T* __restrict__ x, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, T* __restrict__ y,
                                                             float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                             int C, int HW, float eps, int relu, int remap,
                                                             const T* __restrict__ resid = nullptr) {
    constexpr int SUBS = 64 / PXB, G = NW * SUBS;
    __shared__ float red[2][NW][PXB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = lane % PXB, g = wave * SUBS + lane / PXB;
    int bx, n;
    cn_block_remap(bx, n, remap);
    const int hw0 = bx * PXB + px;
    const bool ok = hw0 < HW;
    const int hw = ok ? hw0 : HW - 1;
    // wave-uniform base + 32-bit per-lane offsets (one VGPR per load address; C*HW < 2^32 elements per image)
    const T* xb = x + (size_t)n * C * HW;
    T* yb = y + (size_t)n * C * HW;
    // gamma/beta are loaded here, unconditionally, with everything else: a load inside the (predicated) output
    // loop costs one memory round trip per channel
    float v[CPT], gm[CPT], bt[CPT], rs[RES ? CPT : 1];
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int c = g + G * k; const int ci = c < C ? c : C - 1;
        v[k] = DT<T>::ld(xb + ((unsigned)ci * (unsigned)HW + (unsigned)hw));
        gm[k] = gamma[ci]; bt[k] = beta[ci];
        if constexpr (RES) rs[k] = DT<T>::ld(resid + (size_t)n * C * HW + ((unsigned)ci * (unsigned)HW + (unsigned)hw));
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < CPT; ++k) s += (g + G * k < C) ? v[k] : 0.f;
    s = cn_sum_subs<PXB>(s);
    if (lane < PXB) red[0][wave][px] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) tot += red[0][w][px];
    const float mu = tot / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < CPT; ++k) { const float d = v[k] - mu; q += (g + G * k < C) ? d * d : 0.f; }
    q = cn_sum_subs<PXB>(q);
    if (lane < PXB) red[1][wave][px] = q;
    __syncthreads();
    float tq = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) tq += red[1][w][px];
    const float r = rsqrtf(tq / (float)(C - 1) + eps);
    if (ok && threadIdx.x < PXB) { mean_out[(size_t)n * HW + hw] = mu; rstd_out[(size_t)n * HW + hw] = r; }
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int c = g + G * k;
        if (ok && c < C) {
            float o = gm[k] * ((v[k] - mu) * r) + bt[k];
            if (relu) o = o > 0.f ? o : 0.f;
            if constexpr (RES) {
                // the sum the separate add kernel would form: both terms rounded to the storage type first
                T on; DT<T>::st(&on, o);
                o = DT<T>::ld(&on) + rs[k];
            }
            DT<T>::st(yb + ((unsigned)c * (unsigned)HW + (unsigned)hw), o);
        }
    }
}

// Exact-index chain: the norm between two split-bf16 convolutions.  Input = the float32 output z of
// the exact convolution; from the float32 result y = relu?(gamma * (z - mu) * rstd + beta) it writes
//   y   bf16 [N,C,HW]   - the NOMINAL activation: what the (plain bf16) backward pass of the next layer reads,
//   x3  bf16 [N,3C,HW]  - (hi, lo, hi) of y, hi = bf16(y), lo = bf16(y - hi): the next exact convolution's operand,
//   zb  bf16 [N,C,HW]   - bf16(z), saved for this norm's own backward (cn_bwd_reg_kernel<bf16>),
// so the autograd graph of the chain is the plain bf16 one and no float32 activation is kept or re-read
// (float32 norm + separate split pass: 18 bytes per element and a float32 backward; this: 14 and a bf16 backward).
// RES (exact Generator chain, ResidualBlock generator.py:44 and nothing else): y = norm(z) + (rh + rl), the residual being read
// from the (hi, lo, ..) split image r3 of the block's input (layout rpair: 0 = 3C planes, 1 = pair groups), so the float32-accurate
// trunk value never exists as a float32 tensor.
template <int PXB, int NW, int CPT, bool RES = false>
__global__ __launch_bounds__(NW * 64) void cn_fwd_exact_kernel(const float* __restrict__ z, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, bf16_t* __restrict__ zb,
                                                               bf16_t* __restrict__ y, bf16_t* __restrict__ x3,
                                                               float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                               int C, int HW, float eps, int relu, int remap, int pair,
                                                               const bf16_t* __restrict__ r3 = nullptr, int rpair = 0) {
    constexpr int SUBS = 64 / PXB, G = NW * SUBS;
    __shared__ float red[2][NW][PXB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = lane % PXB, g = wave * SUBS + lane / PXB;
    int bx, n;
    cn_block_remap(bx, n, remap);
    const int hw0 = bx * PXB + px;
    const bool ok = hw0 < HW;
    const int hw = ok ? hw0 : HW - 1;
    const float* xb = z + (size_t)n * C * HW;
    bf16_t* zbb = zb + (size_t)n * C * HW;
    bf16_t* yb = y + (size_t)n * C * HW;
    const int C16 = (C + 15) & ~15;
    bf16_t* x3b = x3 + (size_t)n * (pair ? 2 * C16 : 3 * C) * HW;
    float v[CPT], gm[CPT], bt[CPT], rs[RES ? CPT : 1];
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int c = g + G * k; const int ci = c < C ? c : C - 1;
        v[k] = xb[(unsigned)ci * (unsigned)HW + (unsigned)hw];
        gm[k] = gamma[ci]; bt[k] = beta[ci];
        if constexpr (RES) {
            const bf16_t* rb = r3 + (size_t)n * (rpair ? 2 * C16 : 3 * C) * HW;
            const unsigned oh = rpair ? (unsigned)(32 * (ci >> 4) + (ci & 15)) * (unsigned)HW + (unsigned)hw
                                      : (unsigned)ci * (unsigned)HW + (unsigned)hw;
            const unsigned ol = oh + (rpair ? 16u * (unsigned)HW : (unsigned)C * (unsigned)HW);
            rs[k] = bf2f(rb[oh]) + bf2f(rb[ol]);
        }
    }
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < CPT; ++k) s += (g + G * k < C) ? v[k] : 0.f;
    s = cn_sum_subs<PXB>(s);
    if (lane < PXB) red[0][wave][px] = s;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) tot += red[0][w][px];
    const float mu = tot / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < CPT; ++k) { const float d = v[k] - mu; q += (g + G * k < C) ? d * d : 0.f; }
    q = cn_sum_subs<PXB>(q);
    if (lane < PXB) red[1][wave][px] = q;
    __syncthreads();
    float tq = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) tq += red[1][w][px];
    const float r = rsqrtf(tq / (float)(C - 1) + eps);
    if (ok && threadIdx.x < PXB) { mean_out[(size_t)n * HW + hw] = mu; rstd_out[(size_t)n * HW + hw] = r; }
    const unsigned chw = (unsigned)C * (unsigned)HW;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int c = g + G * k;
        if (ok && c < C) {
            float o = gm[k] * ((v[k] - mu) * r) + bt[k];
            if (relu) o = o > 0.f ? o : 0.f;
            if constexpr (RES) o += rs[k];
            const unsigned off = (unsigned)c * (unsigned)HW + (unsigned)hw;
            const bf16_t h = f2bf(o);
            const bf16_t l = f2bf(o - bf2f(h));
            zbb[off] = f2bf(v[k]);
            yb[off] = h;
            if (!pair) { x3b[off] = h; x3b[off + chw] = l; x3b[off + 2u * chw] = h; }
            else {
                // pair layout of the native split kernels (hific_split3 which = 2): [32 g + j] = hi, [32 g + 16 + j] = lo
                const unsigned o2 = (unsigned)(32 * (c >> 4) + (c & 15)) * (unsigned)HW + (unsigned)hw;
                x3b[o2] = h; x3b[o2 + 16u * (unsigned)HW] = l;
            }
        } else if (pair && ok && c < C16) {        // zero padding channels of the last 16-group
            const unsigned o2 = (unsigned)(32 * (c >> 4) + (c & 15)) * (unsigned)HW + (unsigned)hw;
            x3b[o2] = 0; x3b[o2 + 16u * (unsigned)HW] = 0;
        }
    }
}

// dx as in cn_bwd_dx_kernel; part[blk][0][c] = sum_px dy'*xhat, part[blk][1][c] = sum_px dy' over the PIT pixel groups
// of workgroup blk (dy' = dy masked by the fused ReLU).
// DB: also part[blk][2][c] = sum_px dx (as stored, i.e. rounded to T): the bias gradient of the convolution whose output
// is this norm's input (every conv -> ChannelNorm pair of the Encoder / Generator), so that layer needs no separate
// channel-sum pass over its 8 MB gradient tensor (two launches per layer, src/network/generator.py:28-42).
template <typename T, int PXB, int NW, int CPT, bool DB>
__global__ __launch_bounds__(NW * 64) void cn_bwd_reg_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                             const float* __restrict__ gamma, const float* __restrict__ beta,
                                                             const float* __restrict__ mean, const float* __restrict__ rstd,
                                                             T* __restrict__ dx, float* __restrict__ part,
                                                             int C, int HW, int relu, int pit, int remap) {
    constexpr int SUBS = 64 / PXB, G = NW * SUBS;
    __shared__ float red[2][NW][PXB];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int px = lane % PXB, g = wave * SUBS + lane / PXB;
    int bx, n;
    cn_block_remap(bx, n, remap);
    float pg[CPT], pb[CPT], gm[CPT], bt[CPT];
    float pd[DB ? CPT : 1];
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int c = g + G * k; const int ci = c < C ? c : C - 1;
        pg[k] = 0.f; pb[k] = 0.f; gm[k] = gamma[ci]; bt[k] = beta[ci];      // loaded once, unconditionally
        if constexpr (DB) pd[k] = 0.f;
    }
    for (int it = 0; it < pit; ++it) {
        const int hw0 = (bx * pit + it) * PXB + px;
        const bool ok = hw0 < HW;
        const int hw = ok ? hw0 : HW - 1;
        const T* xb = x + (size_t)n * C * HW;
        const T* gb = dy + (size_t)n * C * HW;
        T* dxb = dx + (size_t)n * C * HW;
        const float mu = mean[(size_t)n * HW + hw];
        const float r = rstd[(size_t)n * HW + hw];
        float xv[CPT], gv[CPT];
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int c = g + G * k; const unsigned o = (unsigned)(c < C ? c : C - 1) * (unsigned)HW + (unsigned)hw;
            xv[k] = DT<T>::ld(xb + o); gv[k] = DT<T>::ld(gb + o);
        }
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int c = g + G * k;
            const float gmk = gm[k];
            const float d = xv[k] - mu, xh = d * r;
            float g0 = gv[k];
            if (relu && !(gmk * xh + bt[k] > 0.f)) g0 = 0.f;
            if (!(ok && c < C)) g0 = 0.f;
            pg[k] += g0 * xh; pb[k] += g0;
            const float gg = g0 * gmk;
            s1 += gg; s2 += gg * d;
            xv[k] = d; gv[k] = gg;                      // keep d and g for the dx pass
        }
        s1 = cn_sum_subs<PXB>(s1); s2 = cn_sum_subs<PXB>(s2);
        if (it > 0) __syncthreads();                    // previous iteration's readers are done with red
        if (lane < PXB) { red[0][wave][px] = s1; red[1][wave][px] = s2; }
        __syncthreads();
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) { t1 += red[0][w][px]; t2 += red[1][w][px]; }
        const float S1 = t1 / (float)C;
        const float S2 = t2 * r * r * r / (float)(C - 1);
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const int c = g + G * k;
            const float dxv = r * (gv[k] - S1) - xv[k] * S2;
            if (ok && c < C) {
                DT<T>::st(dxb + ((unsigned)c * (unsigned)HW + (unsigned)hw), dxv);
                if constexpr (DB) pd[k] += std::is_same<T, float>::value ? dxv : bf2f(f2bf(dxv));   // what the tensor holds
            }
        }
    }
    constexpr int NR = DB ? 3 : 2;
    const size_t blk = (size_t)n * gridDim.x + bx;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const float a = cn_sum_px<PXB>(pg[k]), b = cn_sum_px<PXB>(pb[k]);
        const int c = g + G * k;
        if (px == 0 && c < C) { part[(blk * NR + 0) * C + c] = a; part[(blk * NR + 1) * C + c] = b; }
        if constexpr (DB) {
            const float d = cn_sum_px<PXB>(pd[k]);
            if (px == 0 && c < C) part[(blk * NR + 2) * C + c] = d;
        }
    }
}

// Narrow planes with many channels (the 960-channel 16x16 residual blocks): cn_bwd_reg_kernel reaches its channel groups
// with 8-pixel runs, i.e. 16 useful bytes per channel row and one 2-byte load per element (48 memory instructions per
// thread and pixel group; 25 us for 24 MB).  Here a LANE owns 8 consecutive pixels of a channel - one 16-byte load per
// tensor, one 16-byte store - and a workgroup of 256 threads owns all channels (c = tid + 256 k, k < CPT) of ONE 8-pixel
// group: the per-channel sums (dgamma, dbeta, producer bias) stay in the thread, only the 2 x 8 per-pixel sums over the
// channels cross lanes (DPP/permute butterfly + one LDS hop between the four waves).  bf16, HW % 8 == 0.
// part[] layout as cn_bwd_reg_kernel with PXB = 8, pit = 1 (same workspace size, same column-sum pass).
// Measured (round 3, 16 x 960 x 16x16): 25.2 -> 15.4 us.  (The same lane mapping for the FORWARD kernel - two block-wide
// reductions of 8 values instead of one pass per pixel column - ran 10.5 us against cn_fwd_reg_kernel's 7.5 and was dropped.)
template <int CPT, bool DB>
__global__ __launch_bounds__(256) void cn_bwd_v8_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        const float* __restrict__ mean, const float* __restrict__ rstd,
                                                        bf16_t* __restrict__ dx, float* __restrict__ part,
                                                        int C, int HW, int relu, int remap, int pit) {
    // pit > 1 (round 4: the 120 / 240 / 480-channel planes of the Encoder and the up-convolutions, where cn_bwd_reg_kernel
    // reaches its channel groups with 16-32-pixel runs at 1.1-1.6 TB/s): the workgroup walks `pit` consecutive 8-pixel
    // groups - whole 128-byte lines of every channel row - and keeps the per-channel sums in its threads across them, so the
    // partial rows are written once per workgroup.
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    __shared__ float red[2][4][16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int bx, n;
    cn_block_remap(bx, n, remap);                          // the 8 groups of a 128-byte line on one XCD
    float pgs[CPT], pbs[CPT], pds[DB ? CPT : 1];
#pragma unroll
    for (int k = 0; k < CPT; ++k) { pgs[k] = 0.f; pbs[k] = 0.f; if constexpr (DB) pds[k] = 0.f; }
    for (int it = 0; it < pit; ++it) {
    const int hw0 = (bx * pit + it) * 8;
    if (hw0 >= HW) break;
    float (*redp)[16] = red[it & 1];
    float mu[8], r[8];
    {
        const float4* mp = (const float4*)(mean + (size_t)n * HW + hw0);
        const float4* rp = (const float4*)(rstd + (size_t)n * HW + hw0);
        const float4 m0 = mp[0], m1 = mp[1], r0 = rp[0], r1 = rp[1];
        mu[0] = m0.x; mu[1] = m0.y; mu[2] = m0.z; mu[3] = m0.w; mu[4] = m1.x; mu[5] = m1.y; mu[6] = m1.z; mu[7] = m1.w;
        r[0] = r0.x; r[1] = r0.y; r[2] = r0.z; r[3] = r0.w; r[4] = r1.x; r[5] = r1.y; r[6] = r1.z; r[7] = r1.w;
    }
    const size_t img = (size_t)n * C * HW + hw0;
    u32x4_t xr[CPT], gr[CPT];
    float gm[CPT], bt[CPT];
#pragma unroll
    for (int k = 0; k < CPT; ++k) {                       // every load issued up front, clamped index instead of a branch
        const int c = tid + 256 * k; const int ci = c < C ? c : C - 1;
        xr[k] = *(const u32x4_t*)(x + img + (size_t)ci * HW);
        gr[k] = *(const u32x4_t*)(dy + img + (size_t)ci * HW);
        gm[k] = gamma[ci]; bt[k] = beta[ci];
    }
    float dv[CPT][8], gv[CPT][8];                          // x - mu and gamma * dy' of this thread's elements
    float pg[CPT], pb[CPT];
    float s1[8], s2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const bool okc = tid + 256 * k < C;
        pg[k] = 0.f; pb[k] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned xw = xr[k][j >> 1], gw = gr[k][j >> 1];
            const float xv = __uint_as_float((j & 1) ? (xw & 0xffff0000u) : (xw << 16));
            float g0 = __uint_as_float((j & 1) ? (gw & 0xffff0000u) : (gw << 16));
            const float d = xv - mu[j], xh = d * r[j];
            if (relu && !(gm[k] * xh + bt[k] > 0.f)) g0 = 0.f;
            if (!okc) g0 = 0.f;
            pg[k] += g0 * xh; pb[k] += g0;
            const float gg = g0 * gm[k];
            s1[j] += gg; s2[j] += gg * d;
            dv[k][j] = d; gv[k][j] = gg;
        }
    }
    // per-pixel sums over all channels: butterfly inside the wave, then across the four waves through LDS
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { s1[j] += __shfl_xor(s1[j], m); s2[j] += __shfl_xor(s2[j], m); }
    }
    if (lane == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { redp[wave][j] = s1[j]; redp[wave][8 + j] = s2[j]; }
    }
    __syncthreads();                                       // (the LDS buffer alternates per group: one barrier per group)
    float S1[8], S2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float t1 = (redp[0][j] + redp[1][j]) + (redp[2][j] + redp[3][j]);
        const float t2 = (redp[0][8 + j] + redp[1][8 + j]) + (redp[2][8 + j] + redp[3][8 + j]);
        S1[j] = t1 / (float)C;
        S2[j] = t2 * r[j] * r[j] * r[j] / (float)(C - 1);
    }
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int c = tid + 256 * k;
        if (c < C) {
            u32x4_t o;
            float pd = 0.f;
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                o[j >> 1] = f2bf2(r[j] * (gv[k][j] - S1[j]) - dv[k][j] * S2[j],
                                  r[j + 1] * (gv[k][j + 1] - S1[j + 1]) - dv[k][j + 1] * S2[j + 1]);
                if constexpr (DB) pd += __uint_as_float(o[j >> 1] << 16) + __uint_as_float(o[j >> 1] & 0xffff0000u);   // what the tensor holds
            }
            *(u32x4_t*)(dx + img + (size_t)c * HW) = o;
            pgs[k] += pg[k]; pbs[k] += pb[k];
            if constexpr (DB) pds[k] += pd;
        }
    }
    }   // pixel groups of this workgroup
    constexpr int NR = DB ? 3 : 2;
    const size_t blk = (size_t)n * gridDim.x + bx;
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int c = tid + 256 * k;
        if (c < C) {
            part[(blk * NR + 0) * C + c] = pgs[k];
            part[(blk * NR + 1) * C + c] = pbs[k];
            if constexpr (DB) part[(blk * NR + 2) * C + c] = pds[k];
        }
    }
}

// dgamma/dbeta (=|+=) column sums of part[nblk][2][C]: 64 columns x 16 row lanes per workgroup, coalesced rows
// (nr = 3: third row block = the producing convolution's bias gradient -> dprev, with its own accumulate flag)
__global__ __launch_bounds__(1024) void cn_param_colsum_kernel(const float* __restrict__ part, float* __restrict__ dgamma,
                                                               float* __restrict__ dbeta, int C, int nblk, int accumulate,
                                                               int nr, float* __restrict__ dprev, int accumulate_prev) {
    __shared__ float red[16][64];
    const int col = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
    const int W2 = nr * C;
    float s = 0.f;
    if (col < W2) {
        int r = rl;
        for (; r + 7 * 16 < nblk; r += 8 * 16) {          // 8 independent loads per trip
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = part[(size_t)(r + 16 * j) * W2 + col];
            s += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        }
        for (; r < nblk; r += 16) s += part[(size_t)r * W2 + col];
    }
    red[rl][threadIdx.x & 63] = s;
    __syncthreads();
    if (rl == 0 && col < W2) {
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) t += red[i][threadIdx.x & 63];
        float* d = col < C ? dgamma + col : (col < 2 * C ? dbeta + (col - C) : dprev + (col - 2 * C));
        if (col < 2 * C ? accumulate : accumulate_prev) *d += t; else *d = t;
    }
}

// configuration: fewest channel groups G with <= 32 channels per thread, then more groups (shorter pixel runs)
// while the grid would leave CUs idle
static int cn_remap_flag(int bit) { const char* e = getenv("HIFIC_CN_REMAP"); return ((e ? atoi(e) : 3) & bit) ? 1 : 0; }
struct CnCfg { int pxb, nw, cpt, G; };
// The backward kernel holds 4 values per channel in registers: it stays at <= 512 threads (256 VGPRs) and gets its
// channel groups from narrower pixel runs instead of more waves.
static bool cn_pick(int N, int C, int HW, CnCfg& cfg, bool bwd = false) {
    static const int PXF[5] = {64, 64, 64, 32, 16}, NWF[5] = {4, 8, 16, 16, 16};
    static const int PXR[5] = {64, 64, 32, 16, 8}, NWR[5] = {4, 8, 8, 8, 8};
    const int* PX = bwd ? PXR : PXF; const int* NWS = bwd ? NWR : NWF;
    int sel = -1;
    for (int lim = 16; lim <= 32 && sel < 0; lim += 16)        // <= 16 channels per thread keeps the backward kernel
        for (int i = 0; i < 5; ++i) {                          // spill-free at 1024 threads; 32 only for C > 1024
            const int G = NWS[i] * (64 / PX[i]);
            if (cdiv(C, G) <= lim) { sel = i; break; }
        }
    if (sel < 0) return false;
    while (sel < 4) {
        const long long blocks = (long long)cdiv(HW, PX[sel]) * N;
        const int Gn = NWS[sel + 1] * (64 / PX[sel + 1]);
        if (blocks >= 256 || cdiv(C, Gn) < 4) break;
        ++sel;
    }
    {   // experiment knob: force a table entry (the caller still needs C / G <= 32)
        const char* e = getenv(bwd ? "HIFIC_CN_BWD_SEL" : "HIFIC_CN_FWD_SEL");
        if (e && atoi(e) >= 0 && atoi(e) < 5 && cdiv(C, NWS[atoi(e)] * (64 / PX[atoi(e)])) <= 32) sel = atoi(e);
    }
    cfg.pxb = PX[sel]; cfg.nw = NWS[sel]; cfg.G = NWS[sel] * (64 / PX[sel]);
    cfg.cpt = cdiv(C, cfg.G) <= 16 ? 16 : 32;
    return true;
}
static int cn_pit(int N, int HW, const CnCfg& cfg) {   // pixel groups per workgroup in the backward kernel
    const long long groups = (long long)cdiv(HW, cfg.pxb);
    int pit = (int)(groups * N / 2048);
    if (pit < 1) pit = 1; if (pit > 16) pit = 16;
    return pit;
}

extern "C" {

// x,y: [N,C,H*W] dtype; gamma,beta: [C] f32; mean,rstd: [N,H*W] f32 (saved for backward)
// y = act(norm(x)) + resid (resid: same shape / dtype as x; the residual add of generator.py:44 folded into the block's second
// norm).  HIFIC_ERR_UNSUPPORTED when the shape has no register-resident configuration: the caller adds separately.
int hific_channelnorm_fwd_res(const void* x, const float* gamma, const float* beta, const void* resid, void* y, float* mean,
                              float* rstd, int N, int C, int HW, float eps, int relu, int dtype, hipStream_t st) {
    if (C < 2 || N <= 0 || HW <= 0 || !resid) return HIFIC_ERR_ARG;
    if (dtype != HIFIC_F32 && dtype != HIFIC_BF16) return HIFIC_ERR_ARG;
    CnCfg cfg;
    if (!cn_pick(N, C, HW, cfg) || cfg.cpt != 16) return HIFIC_ERR_UNSUPPORTED;
    dim3 rgrid(cdiv(HW, cfg.pxb), N);
#define CN_FWD_RR(TT, PXB, NWV) hipLaunchKernelGGL((cn_fwd_reg_kernel<TT, PXB, NWV, 16, true>), rgrid, dim3(NWV * 64), 0, st, \
                                       (const TT*)x, gamma, beta, (TT*)y, mean, rstd, C, HW, eps, relu, cn_remap_flag(1), (const TT*)resid)
#define CN_FWD_RC(TT)                                                                         \
    do {                                                                                      \
        if (cfg.pxb == 64 && cfg.nw == 4) CN_FWD_RR(TT, 64, 4);                               \
        else if (cfg.pxb == 64 && cfg.nw == 8) CN_FWD_RR(TT, 64, 8);                          \
        else if (cfg.pxb == 64) CN_FWD_RR(TT, 64, 16);                                        \
        else if (cfg.pxb == 32) CN_FWD_RR(TT, 32, 16);                                        \
        else CN_FWD_RR(TT, 16, 16);                                                           \
    } while (0)
    if (dtype == HIFIC_F32) CN_FWD_RC(float); else CN_FWD_RC(bf16_t);
#undef CN_FWD_RC
#undef CN_FWD_RR
    return hific_launch_status();
}

int hific_channelnorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* rstd,
                          int N, int C, int HW, float eps, int relu, int dtype, hipStream_t st) {
    if (C < 2 || N <= 0 || HW <= 0) return HIFIC_ERR_ARG;
    if (dtype != HIFIC_F32 && dtype != HIFIC_BF16) return HIFIC_ERR_ARG;
    CnCfg cfg;
    if (cn_pick(N, C, HW, cfg)) {
        dim3 rgrid(cdiv(HW, cfg.pxb), N);
#define CN_FWD_R(TT, PXB, NWV, CPT) hipLaunchKernelGGL((cn_fwd_reg_kernel<TT, PXB, NWV, CPT>), rgrid, dim3(NWV * 64), 0, st, \
                                           (const TT*)x, gamma, beta, (TT*)y, mean, rstd, C, HW, eps, relu, cn_remap_flag(1))
#define CN_FWD_C(TT, CPT)                                                                     \
        do {                                                                                  \
            if (cfg.pxb == 64 && cfg.nw == 4) CN_FWD_R(TT, 64, 4, CPT);                       \
            else if (cfg.pxb == 64 && cfg.nw == 8) CN_FWD_R(TT, 64, 8, CPT);                  \
            else if (cfg.pxb == 64) CN_FWD_R(TT, 64, 16, CPT);                                \
            else if (cfg.pxb == 32) CN_FWD_R(TT, 32, 16, CPT);                                \
            else CN_FWD_R(TT, 16, 16, CPT);                                                   \
        } while (0)
        if (dtype == HIFIC_F32) { if (cfg.cpt == 16) CN_FWD_C(float, 16); else CN_FWD_C(float, 32); }
        else { if (cfg.cpt == 16) CN_FWD_C(bf16_t, 16); else CN_FWD_C(bf16_t, 32); }
#undef CN_FWD_C
#undef CN_FWD_R
        return hific_launch_status();
    }
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

// Exact-index chain norm (cn_fwd_exact_kernel): z f32 [N,C,HW] -> zb, y bf16 [N,C,HW], x3 bf16 [N,3C,HW], mean/rstd f32.
// split_layout 0: x3 = (hi, lo, hi) over 3C channels; 2: the pair layout [N, 2 * C16, HW] of the native split kernels.
// HIFIC_ERR_UNSUPPORTED when the shape has no register-resident configuration (the caller then runs the float32 norm and
// hific_split3 instead).
static int cn_fwd_exact_launch(const float* z, const float* gamma, const float* beta, void* zb, void* y, void* x3,
                               float* mean, float* rstd, int N, int C, int HW, float eps, int relu, int split_layout,
                               const void* resid3, int resid_layout, hipStream_t st) {
    if (C < 2 || N <= 0 || HW <= 0 || !z || !zb || !y || !x3 || (split_layout != 0 && split_layout != 2)) return HIFIC_ERR_ARG;
    if (resid3 && resid_layout != 0 && resid_layout != 2) return HIFIC_ERR_ARG;
    if ((long long)3 * C * HW >= (1ll << 31)) return HIFIC_ERR_UNSUPPORTED;
    const int pair = split_layout == 2, rpair = resid_layout == 2;
    if (pair && (long long)2 * ((C + 15) / 16 * 16) * HW >= (1ll << 31)) return HIFIC_ERR_UNSUPPORTED;
    CnCfg cfg;
    if (!cn_pick(N, C, HW, cfg)) return HIFIC_ERR_UNSUPPORTED;
    dim3 rgrid(cdiv(HW, cfg.pxb), N);
#define CN_FWD_X(PXB, NWV, CPT)                                                                                                   \
    do {                                                                                                                          \
        if (resid3) hipLaunchKernelGGL((cn_fwd_exact_kernel<PXB, NWV, CPT, true>), rgrid, dim3(NWV * 64), 0, st, z, gamma, beta,   \
                                       (bf16_t*)zb, (bf16_t*)y, (bf16_t*)x3, mean, rstd, C, HW, eps, relu, cn_remap_flag(1), pair, \
                                       (const bf16_t*)resid3, rpair);                                                             \
        else hipLaunchKernelGGL((cn_fwd_exact_kernel<PXB, NWV, CPT, false>), rgrid, dim3(NWV * 64), 0, st, z, gamma, beta,         \
                                (bf16_t*)zb, (bf16_t*)y, (bf16_t*)x3, mean, rstd, C, HW, eps, relu, cn_remap_flag(1), pair,        \
                                (const bf16_t*)nullptr, 0);                                                                       \
    } while (0)
#define CN_FWD_XC(CPT)                                                                    \
    do {                                                                                  \
        if (cfg.pxb == 64 && cfg.nw == 4) CN_FWD_X(64, 4, CPT);                           \
        else if (cfg.pxb == 64 && cfg.nw == 8) CN_FWD_X(64, 8, CPT);                      \
        else if (cfg.pxb == 64) CN_FWD_X(64, 16, CPT);                                    \
        else if (cfg.pxb == 32) CN_FWD_X(32, 16, CPT);                                    \
        else CN_FWD_X(16, 16, CPT);                                                       \
    } while (0)
    if (cfg.cpt == 16) CN_FWD_XC(16); else CN_FWD_XC(32);
#undef CN_FWD_XC
#undef CN_FWD_X
    return hific_launch_status();
}

int hific_channelnorm_fwd_exact(const float* z, const float* gamma, const float* beta, void* zb, void* y, void* x3,
                                float* mean, float* rstd, int N, int C, int HW, float eps, int relu, int split_layout,
                                hipStream_t st) {
    return cn_fwd_exact_launch(z, gamma, beta, zb, y, x3, mean, rstd, N, C, HW, eps, relu, split_layout, nullptr, 0, st);
}

// The second norm of a ResidualBlock in the exact Generator chain: y = norm(z) + resid, resid given as the split image of the
// block's input (resid_layout 0: (hi, lo, hi) over 3C channels, 2: pair layout); outputs as hific_channelnorm_fwd_exact.
int hific_channelnorm_fwd_exact_res(const float* z, const float* gamma, const float* beta, const void* resid3, int resid_layout,
                                    void* zb, void* y, void* x3, float* mean, float* rstd, int N, int C, int HW, float eps,
                                    int relu, int split_layout, hipStream_t st) {
    if (!resid3) return HIFIC_ERR_ARG;
    return cn_fwd_exact_launch(z, gamma, beta, zb, y, x3, mean, rstd, N, C, HW, eps, relu, split_layout, resid3, resid_layout, st);
}

// ws: at least hific_channelnorm_bwd_ws_bytes(); dgamma/dbeta f32 [C]
size_t hific_channelnorm_bwd_ws_bytes(int N, int C, int HW) {
    size_t b = (size_t)64 * 2 * C * sizeof(float);
    CnCfg cfg;
    if (C >= 2 && N > 0 && HW > 0 && cn_pick(N, C, HW, cfg, true)) {
        const size_t nblk = (size_t)cdiv(cdiv(HW, cfg.pxb), cn_pit(N, HW, cfg)) * N;
        const size_t r = nblk * 3 * C * sizeof(float);
        if (r > b) b = r;
        if (HW % 8 == 0) {                                  // the 16-byte-per-lane kernel's partial rows (<= 1024 + N workgroups)
            const int groups = HW / 8;
            int pitv = (int)((long long)groups * N / 1024);
            if (pitv < 1) pitv = 1; if (pitv > 16) pitv = 16;
            while (groups % pitv) --pitv;
            const size_t rv = (size_t)(groups / pitv) * N * 3 * C * sizeof(float);
            if (rv > b) b = rv;
        }
    }
    return b;
}

extern "C" int hific_channel_sum(const void* x, float* out, int N, int C, int HW, int accumulate, int dtype, void* ws,
                                 size_t ws_bytes, hipStream_t st);

// dprev_bias (may be null): float32 [C], receives (=|+= by accumulate_prev) sum_{n,hw} dx - the bias gradient of the
// convolution that produced x (its only consumer is this norm)
int hific_channelnorm_bwd(const void* x, const void* dy, const float* gamma, const float* beta, const float* mean,
                          const float* rstd, void* dx, float* dgamma, float* dbeta, int N, int C, int HW, int relu,
                          int accumulate, int dtype, void* ws, size_t ws_bytes, float* dprev_bias, int accumulate_prev,
                          hipStream_t st) {
    if (C < 2 || N <= 0 || HW <= 0) return HIFIC_ERR_ARG;
    if (dtype != HIFIC_F32 && dtype != HIFIC_BF16) return HIFIC_ERR_ARG;
    CnCfg cfg;
    if (cn_pick(N, C, HW, cfg, true)) {
        const int pit = cn_pit(N, HW, cfg);
        dim3 rgrid(cdiv(cdiv(HW, cfg.pxb), pit), N);
        const size_t nblk = (size_t)rgrid.x * rgrid.y;
        const int nr = dprev_bias ? 3 : 2;
        if (nblk * nr * C * sizeof(float) > ws_bytes) return HIFIC_ERR_WS;
        float* rpart = (float*)ws;
        // 8-pixel runs per lane: the 16-byte-per-lane kernel (cn_bwd_v8_kernel), same partial layout.  Round 3: the 960-channel
        // 16x16 planes (pxb 8).  Round 4: every plane with more than 256 channels (HIFIC_CN_BWD_V8=2: round-3 rule), `pitv`
        // 8-pixel groups per workgroup so that ~1000 workgroups write partial rows: 480 channels @32x32 43.9 -> 27.5 us.  With one
        // channel per thread (C <= 256: 240 @64x64 88.6 -> 103 us, 120 @128x128 123 -> 186 us) only two 16-byte loads are in
        // flight per thread and group - not taken there.
        static const int v8 = getenv("HIFIC_CN_BWD_V8") ? atoi(getenv("HIFIC_CN_BWD_V8")) : 1;
        const bool v8_shape = cfg.pxb == 8 ? pit == 1 : (v8 != 2 && C > 256);
        if (v8 && dtype == HIFIC_BF16 && v8_shape && HW % 8 == 0 && C <= 1024 &&
            (((size_t)x | (size_t)dy | (size_t)dx | (size_t)mean | (size_t)rstd) & 15) == 0) {
            const int groups = HW / 8;
            int pitv = (int)((long long)groups * N / 1024);
            if (pitv < 1) pitv = 1; if (pitv > 16) pitv = 16;
            while (groups % pitv) --pitv;
            dim3 vgrid(groups / pitv, N);
            const size_t nblk_v = (size_t)vgrid.x * vgrid.y;
            if (nblk_v * nr * C * sizeof(float) > ws_bytes) return HIFIC_ERR_WS;
#define CN_BWD_V8(CPT)                                                                                                  \
            do {                                                                                                        \
                if (dprev_bias) hipLaunchKernelGGL((cn_bwd_v8_kernel<CPT, true>), vgrid, dim3(256), 0, st, (const bf16_t*)x,  \
                    (const bf16_t*)dy, gamma, beta, mean, rstd, (bf16_t*)dx, rpart, C, HW, relu, cn_remap_flag(2), pitv);    \
                else hipLaunchKernelGGL((cn_bwd_v8_kernel<CPT, false>), vgrid, dim3(256), 0, st, (const bf16_t*)x,      \
                    (const bf16_t*)dy, gamma, beta, mean, rstd, (bf16_t*)dx, rpart, C, HW, relu, cn_remap_flag(2), pitv);    \
            } while (0)
            if (C <= 256) CN_BWD_V8(1); else if (C <= 512) CN_BWD_V8(2); else if (C <= 768) CN_BWD_V8(3); else CN_BWD_V8(4);
#undef CN_BWD_V8
            hipLaunchKernelGGL(cn_param_colsum_kernel, dim3(cdiv(nr * C, 64)), dim3(1024), 0, st, rpart, dgamma, dbeta, C,
                               (int)nblk_v, accumulate, nr, dprev_bias, accumulate_prev);
            return hific_launch_status();
        }
#define CN_BWD_R(TT, PXB, NWV, CPT)                                                                                    \
        do {                                                                                                           \
            if (dprev_bias) hipLaunchKernelGGL((cn_bwd_reg_kernel<TT, PXB, NWV, CPT, true>), rgrid, dim3(NWV * 64), 0, st, \
                (const TT*)x, (const TT*)dy, gamma, beta, mean, rstd, (TT*)dx, rpart, C, HW, relu, pit, cn_remap_flag(2)); \
            else hipLaunchKernelGGL((cn_bwd_reg_kernel<TT, PXB, NWV, CPT, false>), rgrid, dim3(NWV * 64), 0, st,       \
                (const TT*)x, (const TT*)dy, gamma, beta, mean, rstd, (TT*)dx, rpart, C, HW, relu, pit, cn_remap_flag(2)); \
        } while (0)
#define CN_BWD_C(TT, CPT)                                                                     \
        do {                                                                                  \
            if (cfg.pxb == 64 && cfg.nw == 4) CN_BWD_R(TT, 64, 4, CPT);                       \
            else if (cfg.pxb == 64) CN_BWD_R(TT, 64, 8, CPT);                                 \
            else if (cfg.pxb == 32) CN_BWD_R(TT, 32, 8, CPT);                                 \
            else if (cfg.pxb == 16) CN_BWD_R(TT, 16, 8, CPT);                                 \
            else CN_BWD_R(TT, 8, 8, CPT);                                                     \
        } while (0)
        if (dtype == HIFIC_F32) { if (cfg.cpt == 16) CN_BWD_C(float, 16); else CN_BWD_C(float, 32); }
        else { if (cfg.cpt == 16) CN_BWD_C(bf16_t, 16); else CN_BWD_C(bf16_t, 32); }
#undef CN_BWD_C
#undef CN_BWD_R
        hipLaunchKernelGGL(cn_param_colsum_kernel, dim3(cdiv(nr * C, 64)), dim3(1024), 0, st, rpart, dgamma, dbeta, C,
                           (int)nblk, accumulate, nr, dprev_bias, accumulate_prev);
        return hific_launch_status();
    }
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
    if (dprev_bias)        // shapes outside the register-resident kernel: the plain channel sum over dx (stream-ordered re-use of ws)
        return hific_channel_sum(dx, dprev_bias, N, C, HW, accumulate_prev, dtype, ws, ws_bytes, st);
    return hific_launch_status();
}

}  // extern "C"
