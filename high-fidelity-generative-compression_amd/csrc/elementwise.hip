// Small HBM-bound kernels of the HiFIC step: activations' backward, residual adds, per-channel sums (bias
// gradients), casts, max-pool (AlexNet), MSE distortion, BCE-with-logits GAN losses, nearest-upsample+concat
// (Discriminator input), spectral norm, fused Adam.  All grid-stride, vectorisable, one pass over HBM each.
// Reference call sites: src/network/generator.py:44,161 (adds), src/network/hyper.py:59-60,91-92 (ReLU),
// src/network/discriminator.py:36,44,74-84, src/model.py:190-194 (MSE), src/loss/losses.py:30-41 (BCE),
// torch.nn.utils.spectral_norm (discriminator.py:46-62), train.py:287-301 (Adam).
#include "common.h"
#include <math.h>

#define EW_GRID(total) dim3((unsigned)((((total) + 255) / 256) > 16384 ? 16384 : (((total) + 255) / 256)))
#define EW_LOOP(i, total) \
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (total); i += (long long)gridDim.x * blockDim.x)

// block-wide sum of one float per thread (256 threads); result valid on thread 0
__device__ __forceinline__ float block_sum_256(float v, float* sh) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = sh[0] + sh[1] + sh[2] + sh[3];
    __syncthreads();
    return r;
}

// second stage of a scalar reduction, run by the last-arriving workgroup of the partial kernel (common.h "tickets"): the same
// summation order as final_sum_kernel
__device__ __forceinline__ void final_sum_tail(const float* part, int n, float mul, float* out, float* sh) {
    float s = 0.f;
    for (int i0 = threadIdx.x; i0 < n; i0 += 4 * 256) {            // independent loads (n <= 1024: one batch)
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = hific_ld_agent(part + (i0 + 256 * j < n ? i0 + 256 * j : 0));
#pragma unroll
        for (int j = 0; j < 4; ++j) s += i0 + 256 * j < n ? v[j] : 0.f;
    }
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) *out = s * mul;
}
#define TICKET_TAIL_ARGS unsigned* ticket, float tk_mul, float* tk_out
#define TICKET_TAIL(part, sh) \
    do { if (ticket && hific_last_block(ticket, gridDim.x)) final_sum_tail(part, (int)gridDim.x, tk_mul, tk_out, sh); } while (0)

template <typename T>
__global__ void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dx, long long n,
                               float slope) {
    EW_LOOP(i, n) {
        const float g = DT<T>::ld(dy + i);
        DT<T>::st(dx + i, DT<T>::ld(y + i) > 0.f ? g : slope * g);
    }
}

template <typename T>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ o, long long n) {
    EW_LOOP(i, n) DT<T>::st(o + i, DT<T>::ld(a + i) + DT<T>::ld(b + i));
}

// bf16 forms with 8 elements per thread (16-byte loads / stores; same per-element arithmetic, so the same bits): the
// residual adds and ReLU backward passes over the 7.8 MB residual-block tensors were ~10 us launches of two-byte accesses
typedef unsigned int ew_u32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ unsigned ew_pack2(float lo, float hi) { return f2bf2(lo, hi); }
__global__ void add_bf16x8_kernel(const ew_u32x4_t* __restrict__ a, const ew_u32x4_t* __restrict__ b,
                                  ew_u32x4_t* __restrict__ o, long long n8) {
    EW_LOOP(i, n8) {
        const ew_u32x4_t va = a[i], vb = b[i];
        ew_u32x4_t r;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            r[k] = ew_pack2(bf2f((bf16_t)(va[k] & 0xffffu)) + bf2f((bf16_t)(vb[k] & 0xffffu)),
                            bf2f((bf16_t)(va[k] >> 16)) + bf2f((bf16_t)(vb[k] >> 16)));
        o[i] = r;
    }
}
__global__ void act_bwd_bf16x8_kernel(const ew_u32x4_t* __restrict__ dy, const ew_u32x4_t* __restrict__ y,
                                      ew_u32x4_t* __restrict__ dx, long long n8, float slope) {
    EW_LOOP(i, n8) {
        const ew_u32x4_t g = dy[i], v = y[i];
        ew_u32x4_t r;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float g0 = bf2f((bf16_t)(g[k] & 0xffffu)), g1 = bf2f((bf16_t)(g[k] >> 16));
            const float y0 = bf2f((bf16_t)(v[k] & 0xffffu)), y1 = bf2f((bf16_t)(v[k] >> 16));
            r[k] = ew_pack2(y0 > 0.f ? g0 : slope * g0, y1 > 0.f ? g1 : slope * g1);
        }
        dx[i] = r;
    }
}

template <typename TI, typename TO>
__global__ void cast_kernel(const TI* __restrict__ a, TO* __restrict__ o, long long n) {
    EW_LOOP(i, n) DT<TO>::st(o + i, DT<TI>::ld(a + i));
}

__global__ void axpby_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o,
                             float alpha, float beta, long long n) {
    EW_LOOP(i, n) o[i] = alpha * a[i] + beta * b[i];
}

// part[split][c] = sum over a slice of (n, hw) of x[n][c][hw]
template <typename T>
__global__ __launch_bounds__(256) void chan_sum_kernel(const T* __restrict__ x, float* __restrict__ part, int N, int C,
                                                       int HW, int nsplit, unsigned* ticket, float* out, int accumulate) {
    // block (c, split): the split owns a contiguous slice of the HW axis of every image (no per-element division)
    __shared__ float sh[4];
    const int c = blockIdx.x, split = blockIdx.y;
    const int per = (HW + nsplit - 1) / nsplit;
    const int lo = split * per;
    int hi = lo + per; if (hi > HW) hi = HW;
    // 8 images per trip with all 8 loads in flight (a one-load-per-iteration loop is a chain of memory round trips:
    // 16 of them for a batch-16 16x16 plane)
    float s = 0.f;
    for (int i = lo + threadIdx.x; i < hi; i += 256) {
        for (int n0 = 0; n0 < N; n0 += 8) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int n = n0 + j < N ? n0 + j : N - 1;
                v[j] = DT<T>::ld(x + ((size_t)n * C + c) * HW + i);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) s += (n0 + j < N) ? v[j] : 0.f;
        }
    }
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) hific_st_agent(part + (size_t)split * C + c, s);
    if (ticket && hific_last_block(ticket, gridDim.x * gridDim.y)) {
        // second stage (chan_sum_reduce_kernel's sums, same order) by the last-arriving workgroup
        // (loads in independent batches of 8: a dependent chain of agent-scope loads costs a memory round trip each)
        for (int cc = threadIdx.x; cc < C; cc += 256) {
            float t = 0.f;
            for (int sp0 = 0; sp0 < nsplit; sp0 += 8) {
                float v[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = hific_ld_agent(part + (size_t)(sp0 + j < nsplit ? sp0 + j : 0) * C + cc);
#pragma unroll
                for (int j = 0; j < 8; ++j) t += sp0 + j < nsplit ? v[j] : 0.f;
            }
            if (accumulate) out[cc] += t; else out[cc] = t;
        }
    }
}
__global__ void chan_sum_reduce_kernel(const float* __restrict__ part, float* __restrict__ out, int C, int nsplit,
                                       int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float s = 0.f;
#pragma unroll 8
    for (int sp = 0; sp < nsplit; ++sp) s += part[(size_t)sp * C + c];
    if (accumulate) out[c] += s; else out[c] = s;
}

// ---- MaxPool2d(kernel 3, stride 2, no padding) -----------------------------------------------------
template <typename T>
__global__ void maxpool3s2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long long planes, int H, int W,
                                      int OH, int OW) {
    const long long total = planes * OH * OW;
    EW_LOOP(i, total) {
        const int ox = (int)(i % OW), oy = (int)((i / OW) % OH);
        const long long pc = i / ((long long)OW * OH);
        const T* xp = x + pc * H * W;
        float m = -INFINITY;
        for (int r = 0; r < 3; ++r)
            for (int s = 0; s < 3; ++s) m = fmaxf(m, DT<T>::ld(xp + (size_t)(oy * 2 + r) * W + ox * 2 + s));
        DT<T>::st(y + i, m);
    }
}
// gather form of the adjoint: dx[iy,ix] = sum over windows containing (iy,ix) whose arg-max (first max in
// row-major window order, torch semantics) is (iy,ix)
template <typename T>
__global__ void maxpool3s2_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx,
                                      long long planes, int H, int W, int OH, int OW) {
    const long long total = planes * H * W;
    EW_LOOP(i, total) {
        const int ix = (int)(i % W), iy = (int)((i / W) % H);
        const long long pc = i / ((long long)W * H);
        const T* xp = x + pc * H * W;
        const T* gp = dy + pc * OH * OW;
        float acc = 0.f;
        int oy_lo = (iy - 2 + 1) / 2; if (iy - 2 < 0) oy_lo = 0;
        int ox_lo = (ix - 2 + 1) / 2; if (ix - 2 < 0) ox_lo = 0;
        for (int oy = oy_lo; oy <= iy / 2 && oy < OH; ++oy)
            for (int ox = ox_lo; ox <= ix / 2 && ox < OW; ++ox) {
                float m = -INFINITY; int am = -1;
                for (int r = 0; r < 3; ++r)
                    for (int s = 0; s < 3; ++s) {
                        const float v = DT<T>::ld(xp + (size_t)(oy * 2 + r) * W + ox * 2 + s);
                        if (v > m || am < 0) { m = v; am = r * 3 + s; }
                    }
                if (am == (iy - oy * 2) * 3 + (ix - ox * 2)) acc += DT<T>::ld(gp + (size_t)oy * OW + ox);
            }
        DT<T>::st(dx + i, acc);
    }
}

// 2x2 stride-2 max pooling (torchvision VGG16.features: nn.MaxPool2d(2, 2); LPIPS net='vgg') and its adjoint in gather
// form: windows do not overlap, an input pixel receives its window's gradient iff it is the window's first maximum in
// row-major order (torch semantics); rows / columns past 2*OH / 2*OW belong to no window.
template <typename T>
__global__ void maxpool2s2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long long planes, int H, int W,
                                      int OH, int OW) {
    const long long total = planes * OH * OW;
    EW_LOOP(i, total) {
        const int ox = (int)(i % OW), oy = (int)((i / OW) % OH);
        const long long pc = i / ((long long)OW * OH);
        const T* xp = x + pc * H * W + (size_t)(oy * 2) * W + ox * 2;
        const float a = DT<T>::ld(xp), b = DT<T>::ld(xp + 1), c = DT<T>::ld(xp + W), d = DT<T>::ld(xp + W + 1);
        DT<T>::st(y + i, fmaxf(fmaxf(a, b), fmaxf(c, d)));
    }
}
template <typename T>
__global__ void maxpool2s2_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx,
                                      long long planes, int H, int W, int OH, int OW) {
    const long long total = planes * H * W;
    EW_LOOP(i, total) {
        const int ix = (int)(i % W), iy = (int)((i / W) % H);
        const long long pc = i / ((long long)W * H);
        const int oy = iy >> 1, ox = ix >> 1;
        float g = 0.f;
        if (oy < OH && ox < OW) {
            const T* xp = x + pc * H * W + (size_t)(oy * 2) * W + ox * 2;
            const float v[4] = {DT<T>::ld(xp), DT<T>::ld(xp + 1), DT<T>::ld(xp + W), DT<T>::ld(xp + W + 1)};
            int am = 0; float m = v[0];
#pragma unroll
            for (int k = 1; k < 4; ++k) if (v[k] > m) { m = v[k]; am = k; }
            if (am == (iy & 1) * 2 + (ix & 1)) g = DT<T>::ld(dy + pc * OH * OW + (size_t)oy * OW + ox);
        }
        DT<T>::st(dx + i, g);
    }
}

// ---- tanh output activation and the [-1,1] -> [0,1] map of `normalize_input_image` (src/model.py:155-156, 206-209) ----
template <typename T>
__global__ void tanh_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long long n) {
    EW_LOOP(i, n) DT<T>::st(y + i, tanhf(DT<T>::ld(x + i)));
}
template <typename T>
__global__ void tanh_bwd_kernel(const T* __restrict__ y, const T* __restrict__ dy, T* __restrict__ dx, long long n) {
    EW_LOOP(i, n) { const float t = DT<T>::ld(y + i); DT<T>::st(dx + i, DT<T>::ld(dy + i) * (1.f - t * t)); }
}
template <typename T>
__global__ void scale_shift_kernel(const T* __restrict__ x, T* __restrict__ y, long long n, float a, float b) {
    EW_LOOP(i, n) DT<T>::st(y + i, a * DT<T>::ld(x + i) + b);
}

// ---- squared-error sum: part[b] = sum (s*a - s*b)^2 ; final = sum(part) -----------------------------
template <typename TA>
__global__ __launch_bounds__(256) void sqdiff_partial_kernel(const TA* __restrict__ a, const float* __restrict__ b,
                                                             float* __restrict__ part, long long n, float scale, TICKET_TAIL_ARGS) {
    __shared__ float sh[4];
    float s = 0.f;
    EW_LOOP(i, n) { const float d = scale * DT<TA>::ld(a + i) - scale * b[i]; s += d * d; }
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) hific_st_agent(part + blockIdx.x, s);
    TICKET_TAIL(part, sh);
}
__global__ __launch_bounds__(256) void final_sum_kernel(const float* __restrict__ part, int n, float mul,
                                                        float* __restrict__ out) {
    __shared__ float sh[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += part[i];
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) *out = s * mul;
}
// da = g * 2*scale^2*(a-b)/n
template <typename TA>
__global__ void sqdiff_bwd_kernel(const TA* __restrict__ a, const float* __restrict__ b, const float* __restrict__ g,
                                  TA* __restrict__ da, long long n, float coef) {
    const float gg = *g * coef;
    EW_LOOP(i, n) DT<TA>::st(da + i, gg * (DT<TA>::ld(a + i) - b[i]));
}

// ---- BCE with logits against a constant target, mean over n ----------------------------------------
__global__ __launch_bounds__(256) void bce_partial_kernel(const float* __restrict__ z, float target,
                                                          float* __restrict__ part, long long n, TICKET_TAIL_ARGS) {
    __shared__ float sh[4];
    float s = 0.f;
    EW_LOOP(i, n) {
        const float x = z[i];
        s += fmaxf(x, 0.f) - x * target + log1pf(expf(-fabsf(x)));
    }
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) hific_st_agent(part + blockIdx.x, s);
    TICKET_TAIL(part, sh);
}
// dz (=|+=) g * (sigmoid(z) - target)/n
__global__ void bce_bwd_kernel(const float* __restrict__ z, float target, const float* __restrict__ g,
                               float* __restrict__ dz, long long n, float inv_n, int accumulate) {
    const float gg = *g * inv_n;
    EW_LOOP(i, n) {
        const float v = gg * (1.f / (1.f + expf(-z[i])) - target);
        if (accumulate) dz[i] += v; else dz[i] = v;
    }
}
// ---- least-squares GAN term on the sigmoid output: mean over n of (sigmoid(z) - target)^2 ----------
__global__ __launch_bounds__(256) void lsq_partial_kernel(const float* __restrict__ z, float target,
                                                          float* __restrict__ part, long long n, TICKET_TAIL_ARGS) {
    __shared__ float sh[4];
    float s = 0.f;
    EW_LOOP(i, n) { const float d = 1.f / (1.f + expf(-z[i])) - target; s += d * d; }
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) hific_st_agent(part + blockIdx.x, s);
    TICKET_TAIL(part, sh);
}
// dz (=|+=) g * 2 (s - target) s (1 - s) / n,  s = sigmoid(z)
__global__ void lsq_bwd_kernel(const float* __restrict__ z, float target, const float* __restrict__ g,
                               float* __restrict__ dz, long long n, float inv_n, int accumulate) {
    const float gg = *g * 2.f * inv_n;
    EW_LOOP(i, n) {
        const float sg = 1.f / (1.f + expf(-z[i]));
        const float v = gg * (sg - target) * sg * (1.f - sg);
        if (accumulate) dz[i] += v; else dz[i] = v;
    }
}
__global__ void sigmoid_kernel(const float* __restrict__ z, float* __restrict__ o, long long n) {
    EW_LOOP(i, n) o[i] = 1.f / (1.f + expf(-z[i]));
}

// ---- Discriminator input: out[n, 0:Ci] = img[n], out[n, Ci:Ci+Cc] = nearest-upsample(ctx[n], x f) ---
template <typename T>
__global__ void upcat_fwd_kernel(const T* __restrict__ img, const T* __restrict__ ctx, T* __restrict__ out, int N,
                                 int Ci, int Cc, int H, int W, int f) {
    const int C = Ci + Cc, h = H / f, w = W / f;
    const long long total = (long long)N * C * H * W;
    EW_LOOP(i, total) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const int c = (int)((i / ((long long)W * H)) % C);
        const int n = (int)(i / ((long long)W * H * C));
        T v;
        if (c < Ci) v = img[(((size_t)n * Ci + c) * H + y) * W + x];
        else v = ctx[(((size_t)n * Cc + (c - Ci)) * h + y / f) * w + x / f];
        out[i] = v;
    }
}
// bf16, W % 8 == 0, f % 8 == 0: one thread = 8 consecutive pixels of a row (one 16-byte store; the image part is one
// 16-byte load, the context part one scalar broadcast) - the element form above is four 64-bit div/mods and a two-byte
// store per element (115 us for the 63 MB Discriminator input).
__global__ void upcat_fwd_vec_kernel(const bf16_t* __restrict__ img, const bf16_t* __restrict__ ctx, bf16_t* __restrict__ out,
                                     unsigned N, int Ci, int Cc, int H, int W, int f) {
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    const int C = Ci + Cc, h = H / f, w = W / f, W8 = W / 8;
    const unsigned total = N * (unsigned)(C * H * W8);
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int x8 = (int)(i % (unsigned)W8);
        unsigned t = i / (unsigned)W8;
        const int y = (int)(t % (unsigned)H); t /= (unsigned)H;
        const int c = (int)(t % (unsigned)C);
        const unsigned n = t / (unsigned)C;
        u32x4_t v;
        if (c < Ci) {
            v = *(const u32x4_t*)(img + (((size_t)n * Ci + c) * H + y) * W + x8 * 8);
        } else {
            const unsigned e = ctx[(((size_t)n * Cc + (c - Ci)) * h + y / f) * w + (x8 * 8) / f];
            const unsigned d = e | (e << 16);
            v = (u32x4_t){d, d, d, d};
        }
        *(u32x4_t*)(out + (size_t)i * 8) = v;
    }
}
// dimg = dout[:, :Ci] (nimg leading images only; others have no grad), dctx = block sums of dout[:, Ci:]
template <typename T>
__global__ void upcat_bwd_img_kernel(const T* __restrict__ dout, T* __restrict__ dimg, int n0, int N, int Ci, int Cc,
                                     int H, int W) {
    const int C = Ci + Cc;
    const long long total = (long long)N * Ci * H * W;
    EW_LOOP(i, total) {
        const long long hw = i % ((long long)H * W);
        const int c = (int)((i / ((long long)H * W)) % Ci);
        const int n = (int)(i / ((long long)H * W * Ci));
        dimg[i] = dout[((size_t)(n0 + n) * C + c) * H * W + hw];
    }
}
template <typename T>
__global__ void upcat_bwd_ctx_kernel(const T* __restrict__ dout, T* __restrict__ dctx, int N, int Ci, int Cc, int H,
                                     int W, int f) {
    const int C = Ci + Cc, h = H / f, w = W / f;
    const long long total = (long long)N * Cc * h * w * 64;   // one wave per output element
    const int lane = threadIdx.x & 63;
    for (long long gi = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; gi < total / 64;
         gi += ((long long)gridDim.x * blockDim.x) >> 6) {
        const int x = (int)(gi % w), y = (int)((gi / w) % h);
        const int c = (int)((gi / ((long long)w * h)) % Cc);
        const int n = (int)(gi / ((long long)w * h * Cc));
        const T* p = dout + (((size_t)n * C + Ci + c) * H + (size_t)y * f) * W + (size_t)x * f;
        float s = 0.f;
        for (int k = lane; k < f * f; k += 64) s += DT<T>::ld(p + (size_t)(k / f) * W + (k % f));
        s = wave_sum(s);
        if (lane == 0) DT<T>::st(dctx + gi, s);
    }
}

// ---- Discriminator input from its three sources (src/model.py:176-179 + src/network/discriminator.py:36,75-77): the reference
//      builds cat([real, gen]) and repeat_interleave(latents, 2) and runs the context conv on the 2B repeated latents.  Here
//      image n < B is real[n], image n >= B is gen[n - B], and image n takes the context map of latent n >> 1 (the pairing
//      quirk: repeat_interleave duplicates neighbours, cat stacks halves) - the context conv runs on the B latents once.
__global__ void upcat_pair_fwd_kernel(const bf16_t* __restrict__ real, const bf16_t* __restrict__ gen,
                                      const bf16_t* __restrict__ ctx, bf16_t* __restrict__ out, unsigned B, int Ci, int Cc,
                                      int H, int W, int f) {
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    const int C = Ci + Cc, h = H / f, w = W / f, W8 = W / 8;
    const unsigned total = 2u * B * (unsigned)(C * H * W8);
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int x8 = (int)(i % (unsigned)W8);
        unsigned t = i / (unsigned)W8;
        const int y = (int)(t % (unsigned)H); t /= (unsigned)H;
        const int c = (int)(t % (unsigned)C);
        const unsigned n = t / (unsigned)C;
        u32x4_t v;
        if (c < Ci) {
            const bf16_t* src = n < B ? real + (size_t)n * Ci * H * W : gen + (size_t)(n - B) * Ci * H * W;
            v = *(const u32x4_t*)(src + ((size_t)c * H + y) * W + x8 * 8);
        } else {
            const unsigned e = ctx[(((size_t)(n >> 1) * Cc + (c - Ci)) * h + y / f) * w + (x8 * 8) / f];
            const unsigned d = e | (e << 16);
            v = (u32x4_t){d, d, d, d};
        }
        *(u32x4_t*)(out + (size_t)i * 8) = v;
    }
}
template <typename T>
__global__ void upcat_pair_fwd_elem_kernel(const T* __restrict__ real, const T* __restrict__ gen, const T* __restrict__ ctx,
                                           T* __restrict__ out, int B, int Ci, int Cc, int H, int W, int f) {
    const int C = Ci + Cc, h = H / f, w = W / f;
    const long long total = 2ll * B * C * H * W;
    EW_LOOP(i, total) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const int c = (int)((i / ((long long)W * H)) % C);
        const int n = (int)(i / ((long long)W * H * C));
        T v;
        if (c < Ci) v = (n < B ? real : gen)[(((size_t)(n < B ? n : n - B) * Ci + c) * H + y) * W + x];
        else v = ctx[(((size_t)(n >> 1) * Cc + (c - Ci)) * h + y / f) * w + x / f];
        out[i] = v;
    }
}
// dctx[k] = block sums of dout[2k, Ci:] + dout[2k+1, Ci:] (the two images that read latent k's context map)
template <typename T>
__global__ void upcat_pair_bwd_ctx_kernel(const T* __restrict__ dout, T* __restrict__ dctx, int B, int Ci, int Cc, int H,
                                          int W, int f) {
    const int C = Ci + Cc, h = H / f, w = W / f;
    const long long total = (long long)B * Cc * h * w;          // one wave per output element
    const int lane = threadIdx.x & 63;
    for (long long gi = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6; gi < total;
         gi += ((long long)gridDim.x * blockDim.x) >> 6) {
        const int x = (int)(gi % w), y = (int)((gi / w) % h);
        const int c = (int)((gi / ((long long)w * h)) % Cc);
        const int k = (int)(gi / ((long long)w * h * Cc));
        float s = 0.f;
        for (int j = 0; j < 2; ++j) {
            const T* p = dout + (((size_t)(2 * k + j) * C + Ci + c) * H + (size_t)y * f) * W + (size_t)x * f;
            for (int q = lane; q < f * f; q += 64) s += DT<T>::ld(p + (size_t)(q / f) * W + (q % f));
        }
        s = wave_sum(s);
        if (lane == 0) DT<T>::st(dctx + gi, s);
    }
}

// ---- spectral norm (one power iteration, torch.nn.utils.spectral_norm semantics) --------------------
// W: [K, M] row-major f32.  step 1: vraw[m] = sum_k W[k,m] u[k]
// The K rows are cut into slices of SN_ROWS (blockIdx.y): all loads of a thread are independent and issued together,
// and the grid has K/16 times more blocks (a per-column loop over all K rows was one latency chain per thread on
// M/256 <= 32 blocks: 94-200 us for a 2-8 MB matrix).  part[slice][m]; sn_colsum_kernel adds the slices in order.
#define SN_ROWS 16
__global__ __launch_bounds__(256) void sn_wtu_kernel(const float* __restrict__ W, const float* __restrict__ u,
                                                     float* __restrict__ part, int K, int M) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const int k0 = blockIdx.y * SN_ROWS;
    float w[SN_ROWS];
#pragma unroll
    for (int j = 0; j < SN_ROWS; ++j) {
        const int k = k0 + j < K ? k0 + j : K - 1;
        w[j] = W[(size_t)k * M + m];
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < SN_ROWS; ++j) s += (k0 + j < K) ? w[j] * u[k0 + j] : 0.f;
    part[(size_t)blockIdx.y * M + m] = s;
}
__global__ __launch_bounds__(256) void sn_colsum_kernel(const float* __restrict__ part, float* __restrict__ vraw,
                                                        int nslice, int M) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    float s = 0.f;
    int j = 0;
    for (; j + 8 <= nslice; j += 8) {
        float t[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) t[q] = part[(size_t)(j + q) * M + m];
#pragma unroll
        for (int q = 0; q < 8; ++q) s += t[q];
    }
    for (; j < nslice; ++j) s += part[(size_t)j * M + m];
    vraw[m] = s;
}
// x <- x / max(||x||, eps)   (single block)
__global__ __launch_bounds__(256) void sn_normalize_kernel(const float* __restrict__ x, float* __restrict__ o, int n,
                                                           float eps) {
    __shared__ float sh[4];
    __shared__ float inv;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += x[i] * x[i];
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) inv = 1.f / fmaxf(sqrtf(s), eps);
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) o[i] = x[i] * inv;
}
// step 2: wv[k] = sum_m W[k,m] v[m]   (one block per row)
__global__ __launch_bounds__(256) void sn_wv_kernel(const float* __restrict__ W, const float* __restrict__ v,
                                                    float* __restrict__ wv, int K, int M) {
    __shared__ float sh[4];
    const int k = blockIdx.x;
    float s = 0.f;
    for (int m = threadIdx.x; m < M; m += 256) s += W[(size_t)k * M + m] * v[m];
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) wv[k] = s;
}
// sigma = u . wv ; inv_sigma = 1/sigma  (single block)
__global__ __launch_bounds__(256) void sn_sigma_kernel(const float* __restrict__ u, const float* __restrict__ wv,
                                                       float* __restrict__ sigma_out, int K) {
    __shared__ float sh[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < K; i += 256) s += u[i] * wv[i];
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) { sigma_out[0] = s; sigma_out[1] = 1.f / s; }
}
// ---- the same chain for several layers per launch (the Discriminator's four spectral-norm convolutions: their power
// iterations depend only on the weights and run at the top of Discriminator.forward - 6 launches instead of 24) --------------
#define SN_MAXJOBS 8
struct SnJob { const float* W; float* u; float* v; float* sig; float* tmpM; float* tmpK; float* part; int K, M; };
struct SnJobs { SnJob j[SN_MAXJOBS]; };
__global__ __launch_bounds__(256) void sn_wtu_batch_kernel(const SnJobs J) {
    const SnJob& q = J.j[blockIdx.z];
    const int m = blockIdx.x * 256 + threadIdx.x;
    const int k0 = blockIdx.y * SN_ROWS;
    if (m >= q.M || k0 >= q.K) return;
    float w[SN_ROWS];
#pragma unroll
    for (int j = 0; j < SN_ROWS; ++j) {
        const int k = k0 + j < q.K ? k0 + j : q.K - 1;
        w[j] = q.W[(size_t)k * q.M + m];
    }
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < SN_ROWS; ++j) s += (k0 + j < q.K) ? w[j] * q.u[k0 + j] : 0.f;
    q.part[(size_t)blockIdx.y * q.M + m] = s;
}
__global__ __launch_bounds__(256) void sn_colsum_batch_kernel(const SnJobs J) {
    const SnJob& q = J.j[blockIdx.y];
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= q.M) return;
    const int nslice = (q.K + SN_ROWS - 1) / SN_ROWS;
    float s = 0.f;
    int j = 0;
    for (; j + 8 <= nslice; j += 8) {
        float t[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) t[r] = q.part[(size_t)(j + r) * q.M + m];
#pragma unroll
        for (int r = 0; r < 8; ++r) s += t[r];
    }
    for (; j < nslice; ++j) s += q.part[(size_t)j * q.M + m];
    q.tmpM[m] = s;
}
// which 0: v <- normalize(tmpM); 1: u <- normalize(tmpK)
__global__ __launch_bounds__(256) void sn_normalize_batch_kernel(const SnJobs J, int which, float eps) {
    const SnJob& q = J.j[blockIdx.x];
    const float* x = which ? q.tmpK : q.tmpM;
    float* o = which ? q.u : q.v;
    const int n = which ? q.K : q.M;
    __shared__ float sh[4];
    __shared__ float inv;
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += x[i] * x[i];
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) inv = 1.f / fmaxf(sqrtf(s), eps);
    __syncthreads();
    for (int i = threadIdx.x; i < n; i += 256) o[i] = x[i] * inv;
}
__global__ __launch_bounds__(256) void sn_wv_batch_kernel(const SnJobs J) {
    const SnJob& q = J.j[blockIdx.y];
    const int k = blockIdx.x;
    if (k >= q.K) return;
    __shared__ float sh[4];
    float s = 0.f;
    for (int m = threadIdx.x; m < q.M; m += 256) s += q.W[(size_t)k * q.M + m] * q.v[m];
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) q.tmpK[k] = s;
}
__global__ __launch_bounds__(256) void sn_sigma_batch_kernel(const SnJobs J, int snap) {
    const SnJob& q = J.j[blockIdx.x];
    __shared__ float sh[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < q.K; i += 256) s += q.u[i] * q.tmpK[i];
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) { q.sig[0] = s; q.sig[1] = 1.f / s; }
    if (snap) {
        // the post-iteration (u, v) this forward's backward needs, behind sigma: the next forward iterates u, v in place
        // (torch's spectral_norm clones them for the same reason); 8 small device copies per Discriminator forward otherwise
        for (int i = threadIdx.x; i < q.K; i += 256) q.sig[2 + i] = q.u[i];
        for (int i = threadIdx.x; i < q.M; i += 256) q.sig[2 + q.K + i] = q.v[i];
    }
}
// backward: dWorig = (dW - (sum dW*Worig)/sigma * u v^T) / sigma
__global__ __launch_bounds__(256) void sn_dot_partial_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             float* __restrict__ part, long long n, TICKET_TAIL_ARGS) {
    __shared__ float sh[4];
    float s = 0.f;
    EW_LOOP(i, n) s += a[i] * b[i];
    s = block_sum_256(s, sh);
    if (threadIdx.x == 0) hific_st_agent(part + blockIdx.x, s);
    TICKET_TAIL(part, sh);
}
__global__ void sn_bwd_kernel(const float* __restrict__ dW, const float* __restrict__ u, const float* __restrict__ v,
                              const float* __restrict__ sigma, const float* __restrict__ dot,
                              float* __restrict__ dWorig, int K, int M, int accumulate) {
    const float inv = sigma[1];
    const float coef = *dot * inv * inv;   // (sum dW*Worig)/sigma^2
    const long long total = (long long)K * M;
    EW_LOOP(i, total) {
        const int m = (int)(i % M), k = (int)(i / M);
        const float val = dW[i] * inv - coef * u[k] * v[m];
        if (accumulate) dWorig[i] += val; else dWorig[i] = val;
    }
}

// ---- fused Adam over a flat parameter arena (torch.optim.Adam, amsgrad=False, weight_decay=0) --------
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, float lr, float b1, float b2, float eps, float bc1,
                            float bc2_sqrt, float grad_scale) {
    EW_LOOP(i, n) {
        const float gi = g[i] * grad_scale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] -= (lr / bc1) * (mi / denom);
    }
}

// Device-side step counter (hipGraph replays cannot change kernel arguments): adam_prep_kernel advances the step and
// publishes the two bias-correction factors, adam_dev_kernel reads them.  Same arithmetic as the host path.
__global__ void adam_prep_kernel(int* __restrict__ step, float* __restrict__ bc, float b1, float b2) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const int t = *step + 1;
        *step = t;
        bc[0] = (float)(1.0 - pow((double)b1, (double)t));
        bc[1] = (float)sqrt(1.0 - pow((double)b2, (double)t));
    }
}
__device__ __forceinline__ void adam_elem(float& p, float g, float& m, float& v, float lr, float b1, float b2, float eps,
                                          float bc1, float bc2_sqrt, float grad_scale) {
    const float gi = g * grad_scale;
    const float mi = b1 * m + (1.f - b1) * gi;
    const float vi = b2 * v + (1.f - b2) * gi * gi;
    m = mi; v = vi;
    const float denom = sqrtf(vi) / bc2_sqrt + eps;
    p -= (lr / bc1) * (mi / denom);
}
#ifndef ADAM_V
#define ADAM_V 2     // 0 = one 16-byte group per thread and trip (1067-1099 us for the 181 M-parameter arena), 1 = two groups (1043-1063),
                     // 2 = + non-temporal accesses for g / m / v, which are streamed once per step (1000-1035 us = 5.0 TB/s; same arithmetic)
#endif
typedef float adam_f4 __attribute__((ext_vector_type(4)));
__global__ void adam_dev4_kernel(float4* __restrict__ p, const float4* __restrict__ g, float4* __restrict__ m,
                                 float4* __restrict__ v, long long n4, float lr, float b1, float b2, float eps,
                                 const float* __restrict__ bc, float grad_scale) {
    const float bc1 = bc[0], bc2_sqrt = bc[1];
#if ADAM_V == 0
    EW_LOOP(i, n4) {
        float4 pi = p[i], mi = m[i], vi = v[i];
        const float4 gi = g[i];
        adam_elem(pi.x, gi.x, mi.x, vi.x, lr, b1, b2, eps, bc1, bc2_sqrt, grad_scale);
        adam_elem(pi.y, gi.y, mi.y, vi.y, lr, b1, b2, eps, bc1, bc2_sqrt, grad_scale);
        adam_elem(pi.z, gi.z, mi.z, vi.z, lr, b1, b2, eps, bc1, bc2_sqrt, grad_scale);
        adam_elem(pi.w, gi.w, mi.w, vi.w, lr, b1, b2, eps, bc1, bc2_sqrt, grad_scale);
        m[i] = mi; v[i] = vi; p[i] = pi;
    }
#else
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += 2 * stride) {
        const long long j = i + stride < n4 ? i + stride : i;      // second group of the trip (the tail re-reads the first)
        adam_f4 pa = ((adam_f4*)p)[i], pb = ((adam_f4*)p)[j];
#if ADAM_V == 2
        adam_f4 ma = __builtin_nontemporal_load((adam_f4*)m + i), mb = __builtin_nontemporal_load((adam_f4*)m + j);
        adam_f4 va = __builtin_nontemporal_load((adam_f4*)v + i), vb = __builtin_nontemporal_load((adam_f4*)v + j);
        const adam_f4 ga = __builtin_nontemporal_load((const adam_f4*)g + i), gb = __builtin_nontemporal_load((const adam_f4*)g + j);
#else
        adam_f4 ma = ((adam_f4*)m)[i], mb = ((adam_f4*)m)[j];
        adam_f4 va = ((adam_f4*)v)[i], vb = ((adam_f4*)v)[j];
        const adam_f4 ga = ((const adam_f4*)g)[i], gb = ((const adam_f4*)g)[j];
#endif
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float x = pa[e], mm = ma[e], vv = va[e];
            adam_elem(x, ga[e], mm, vv, lr, b1, b2, eps, bc1, bc2_sqrt, grad_scale);
            pa[e] = x; ma[e] = mm; va[e] = vv;
            float y = pb[e], m2 = mb[e], v2 = vb[e];
            adam_elem(y, gb[e], m2, v2, lr, b1, b2, eps, bc1, bc2_sqrt, grad_scale);
            pb[e] = y; mb[e] = m2; vb[e] = v2;
        }
#if ADAM_V == 2
        __builtin_nontemporal_store(ma, (adam_f4*)m + i); __builtin_nontemporal_store(va, (adam_f4*)v + i);
#else
        ((adam_f4*)m)[i] = ma; ((adam_f4*)v)[i] = va;
#endif
        ((adam_f4*)p)[i] = pa;
        if (j != i) {
#if ADAM_V == 2
            __builtin_nontemporal_store(mb, (adam_f4*)m + j); __builtin_nontemporal_store(vb, (adam_f4*)v + j);
#else
            ((adam_f4*)m)[j] = mb; ((adam_f4*)v)[j] = vb;
#endif
            ((adam_f4*)p)[j] = pb;
        }
    }
#endif
}
__global__ void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                float* __restrict__ v, long long n, float lr, float b1, float b2, float eps,
                                const float* __restrict__ bc, float grad_scale) {
    const float bc1 = bc[0], bc2_sqrt = bc[1];
    EW_LOOP(i, n) {
        const float gi = g[i] * grad_scale;
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] -= (lr / bc1) * (mi / denom);
    }
}

// ---------------------------------------------------------------------------------------------------
// Split-bf16 operands (exact-index mode).  A float32 value v is carried as hi = bf16(v) and
// lo = bf16(v - hi) (v - hi is exact in f32); x*w ~= xh*wh + xl*wh + xh*wl (relative error ~2^-17 per product instead
// of 2^-9), three bf16 MFMA products accumulated in f32.  The three terms are laid out as 3C reduction channels so the
// ordinary bf16 contraction kernels compute the sum: activations (hi, lo, hi), weights (hi, hi, lo).
// src f32 [outer][C][inner] -> dst [outer][3C][inner].
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ void split_hl(float v, float& h, float& l) {
    h = bf2f(f2bf(v));
    l = bf2f(f2bf(v - h));
}
template <typename TO, int WHICH>
__global__ void split3_kernel(const float* __restrict__ src, TO* __restrict__ dst, long long total, unsigned C,
                              unsigned inner) {
    const unsigned ci = C * inner;
    EW_LOOP(i, total) {
        const unsigned long long o = (unsigned long long)i / ci;
        const unsigned r = (unsigned)((unsigned long long)i - o * ci);
        float h, l;
        split_hl(src[i], h, l);
        TO* d = dst + o * 3ull * ci + r;
        DT<TO>::st(d, h);
        DT<TO>::st(d + ci, WHICH == 0 ? l : h);
        DT<TO>::st(d + 2ull * ci, WHICH == 0 ? h : l);
    }
}
// 4 consecutive elements per thread (inner % 4 == 0, 16-byte aligned): one 16-byte load, three 8-byte bf16 stores
template <int WHICH>
__global__ void split3_bf16x4_kernel(const float4* __restrict__ src, uint2* __restrict__ dst, long long total4, unsigned C,
                                     unsigned inner4) {
    const unsigned ci = C * inner4;
    EW_LOOP(i, total4) {
        const unsigned long long o = (unsigned long long)i / ci;
        const unsigned r = (unsigned)((unsigned long long)i - o * ci);
        const float4 v = src[i];
        float h0, l0, h1, l1, h2, l2, h3, l3;
        split_hl(v.x, h0, l0); split_hl(v.y, h1, l1); split_hl(v.z, h2, l2); split_hl(v.w, h3, l3);
        uint2 H, L;
        H.x = ew_pack2(h0, h1); H.y = ew_pack2(h2, h3);
        L.x = ew_pack2(l0, l1); L.y = ew_pack2(l2, l3);
        uint2* d = dst + o * 3ull * ci + r;
        d[0] = H;
        d[ci] = WHICH == 0 ? L : H;
        d[2ull * ci] = WHICH == 0 ? H : L;
    }
}
// Pair layout of the native split-bf16 kernels (gconv_kernel SPLIT): channel c = 16 g + j of the source becomes channels
// 32 g + j (hi) and 32 g + 16 + j (lo) of dst [outer][2 * C16][inner], C16 = C rounded up to 16; the padding channels of the
// last group are written as zeros.  The same layout serves activations and weights: the kernel forms the cross terms.
template <typename TO>
__global__ void split_pair_kernel(const float* __restrict__ src, TO* __restrict__ dst, long long total, unsigned C,
                                  unsigned C16, unsigned inner) {
    const unsigned ci16 = C16 * inner;
    EW_LOOP(i, total) {
        const unsigned long long o = (unsigned long long)i / ci16;
        const unsigned r = (unsigned)((unsigned long long)i - o * ci16);
        const unsigned c = r / inner, q = r - c * inner;
        float h = 0.f, l = 0.f;
        if (c < C) split_hl(src[(o * C + c) * inner + q], h, l);
        TO* d = dst + (o * 2ull * C16 + 32u * (c >> 4) + (c & 15u)) * inner + q;
        DT<TO>::st(d, h);
        DT<TO>::st(d + 16ull * inner, l);
    }
}

// ---------------------------------------------------------------------------------------------------
// Gradient of the Discriminator's context maps THROUGH its first convolution, without the data gradient of that convolution
// (src/network/discriminator.py:75-78: x = cat(x, upsample16(context)); conv1 = 4x4, stride 2, reflect pad 1).
// d ctx[m][c][Y][X] = sum over the f x f block of the input plane of dx[i][Ci + c] for the images i in {2m, 2m+1} (model.py:
// 176-179: repeat_interleave pairs them), and dx = conv1^T dz.  Both are linear and the block sum commutes with the taps:
//   d ctx = (1/sigma) sum_i sum_k sum_{r,s} w[k][Ci+c][r][s] * sum_{q in Q_r(Y)} sum_{q' in Q_s(X)} dz[i][k][q][q']
// where Q_r(b) = {q : P0(b) <= 2q + r <= P1(b)} and [P0, P1] is the block's range of PADDED rows, the first / last block taking
// the mirror row with it (reflect pad 1: padded row 0 is input row 1, padded row H+1 is input row H-2).  One pass over dz (67 MB
// at the benchmark shape) instead of a 15-channel data gradient on the 258 x 258 padded plane + its block-sum kernel.
// Two launches.  d1_ctx_win_kernel, grid (H / f) x B x column segments, 512 threads: a workgroup walks the K output channels of
// its two images four at a time; thread (channel, column) sums its column over each tap row's window (rows requested one step
// ahead, registers) -> LDS -> thread (channel, tap row, tap column, block) forms the column window (one base window + edge
// corrections), adds the pair's two images and writes T[(m, Y, X)][(k, r, s)] (float32, 16 MB at the benchmark shape).
// d1_ctx_dot_kernel: d ctx[(m, Y, X)][c] = (1/sigma) sum_j T[.][j] w[j][c], the weights in LDS.
// ---------------------------------------------------------------------------------------------------
#define D1C_MAXW 128      // columns of dz per workgroup (blockIdx.z = column segment)
#define D1C_NK 8          // output channels per step: loading thread = (channel, column pair)
// NR = rows a block's taps can draw from (f / 2 + 2); HALO: the plane is wider than one segment (the segment's edge threads also
// carry the column outside it).  Every load is unconditional from a clamped address and masked afterwards (loads under a
// condition make hipcc wait for each one); a thread loads two neighbouring columns per instruction.
template <typename TI> struct D1Pair;
template <> struct D1Pair<bf16_t> {
    typedef unsigned type;
    __device__ static __forceinline__ float lo(unsigned v) { return __uint_as_float(v << 16); }
    __device__ static __forceinline__ float hi(unsigned v) { return __uint_as_float(v & 0xffff0000u); }
};
template <> struct D1Pair<float> {
    typedef float2 type;
    __device__ static __forceinline__ float lo(float2 v) { return v.x; }
    __device__ static __forceinline__ float hi(float2 v) { return v.y; }
};
template <typename TI, int NR, bool HALO>
__global__ __launch_bounds__(512) void d1_ctx_win_kernel(const TI* __restrict__ dz, float* __restrict__ T, int B, int K, int OH,
                                                         int OW, int f) {
    // vertical window sums of the channels in flight, per tap row: index 0 = the column left of the segment, 1 .. 128 = the
    // segment, 129 = the column right of it (a block's outermost tap windows reach one column past its own)
    __shared__ float V[D1C_NK][4][D1C_MAXW + 2];
    typedef typename D1Pair<TI>::type PT;
    const int tid = threadIdx.x;
    const int Y = blockIdx.x, m = blockIdx.y;
    const int H = OH * 2, NB = H / f, NBX = (OW * 2) / f;
    const int bps = D1C_MAXW / (f / 2);                   // blocks per column segment (16)
    const int X0 = blockIdx.z * bps, c0 = X0 * (f / 2);   // first block / first column of this segment
    const int nbx = NBX - X0 < bps ? NBX - X0 : bps;
    // padded row range of block Y and the output rows each tap row draws from
    const int P0 = Y == 0 ? 0 : f * Y + 1, P1 = Y == NB - 1 ? H + 1 : f * Y + f;
    int qlo[4], qhi[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int lo = (P0 - r + 1) >> 1, hi = (P1 - r) >> 1;   // ceil((P0 - r) / 2), floor((P1 - r) / 2)  (arithmetic shifts)
        qlo[r] = lo < 0 ? 0 : lo; qhi[r] = hi > OH - 1 ? OH - 1 : hi;
    }
    const int row0 = qlo[3], row1 = qhi[0];               // tap row 3 starts lowest, tap row 0 ends highest
    // window threads: value u = tid + 512 e (e < 4) of the 16 x 128 values of a step -> block column (tid >> 7) + 4 e of the
    // segment, j = tid & 127 = (channel of the step, tap row, tap column).  Column windows as indices into V (column - c0 + 1):
    // the window of tap columns 1 and 2 is exactly f / 2 = NR - 2 columns for every block; tap columns 0 and 3 differ from it by
    // at most one column per end: `ia` / `ib` = index added at the low / high end (or -1), `sl` / `sh` = the base window's first /
    // last column is subtracted
    int wlo[4], ia[4], ib[4];
    bool sl[4], sh[4], wok[4];
    const int jj = tid & 127, jk = jj >> 4, jr = (jj >> 2) & 3, js = jj & 3;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int xl = (tid >> 7) + 4 * e;                 // block column inside the segment
        wok[e] = xl < nbx;
        const int cX = X0 + (wok[e] ? xl : 0);
        const int W2 = OW * 2;
        const int PX0 = cX == 0 ? 0 : f * cX + 1, PX1 = cX == NBX - 1 ? W2 + 1 : f * cX + f;
        int slo[4], shi[4];
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2) {
            const int lo = (PX0 - s2 + 1) >> 1, hi = (PX1 - s2) >> 1;
            slo[s2] = (lo < 0 ? 0 : lo) - c0 + 1; shi[s2] = (hi > OW - 1 ? OW - 1 : hi) - c0 + 1;
        }
        wlo[e] = slo[1];
        ia[e] = slo[js] < slo[1] ? slo[js] : -1; sl[e] = slo[js] > slo[1];
        ib[e] = shi[js] > shi[1] ? shi[js] : -1; sh[e] = shi[js] < shi[1];
    }
    const int kk = tid >> 6, lp = tid & 63;               // loading threads: channel of the step, column pair of the segment
    const int col = c0 + 2 * lp;                          // (OW is even: a pair is inside or outside as a whole)
    // the segment's first / last pair thread also carries the column outside it (-1: none)
    const int hcol = lp == 0 ? (c0 > 0 ? c0 - 1 : -1) : (lp == 63 ? (c0 + D1C_MAXW < OW ? c0 + D1C_MAXW : -1) : -1);
    const int hidx = lp == 0 ? 0 : D1C_MAXW + 1;
    const size_t plane = (size_t)OH * OW;
    const int nstep = 2 * ((K + D1C_NK - 1) / D1C_NK);    // (channel group, image) steps
    // row weights of the four tap rows (1 inside the tap row's window, 0 outside: uniform over the workgroup) and this thread's
    // clamped element offsets inside a channel plane
    unsigned roff[NR], hoff[NR];
    float mk[NR][4];
#pragma unroll
    for (int j = 0; j < NR; ++j) {
        const int q = row0 + j, qc = q <= row1 ? q : row1;
        roff[j] = (unsigned)(qc * OW + (col < OW ? col : OW - 2));
        hoff[j] = (unsigned)(qc * OW + (hcol >= 0 ? hcol : 0));
#pragma unroll
        for (int r = 0; r < 4; ++r) mk[j][r] = (q <= row1 && q >= qlo[r] && q <= qhi[r]) ? 1.f : 0.f;
    }
    const bool cok = col < OW, hok = hcol >= 0;
    PT xr[NR];
    typename std::conditional<std::is_same<TI, float>::value, float, unsigned short>::type xh[NR];
    // rows of step st_ -> xr / xh
#define D1C_LOAD(st_)                                                                                   \
    do {                                                                                                \
        const int img_ = (st_) & 1, k_ = ((st_) >> 1) * D1C_NK + kk;                                    \
        const TI* zk_ = dz + ((size_t)(2 * m + img_) * K + (k_ < K ? k_ : K - 1)) * plane;              \
        _Pragma("unroll") for (int j = 0; j < NR; ++j) {                                                \
            xr[j] = *(const PT*)(zk_ + roff[j]);                                                        \
            if constexpr (HALO) xh[j] = zk_[hoff[j]];                                                   \
        }                                                                                               \
    } while (0)
    D1C_LOAD(0);
    float tacc[4] = {0.f, 0.f, 0.f, 0.f};
    const size_t J = (size_t)K * 16;
    float* Trow = T + (((size_t)m * NB + Y) * NBX + X0) * J;
    for (int st = 0; st < nstep; ++st) {
        float v0[4] = {0.f, 0.f, 0.f, 0.f}, v1[4] = {0.f, 0.f, 0.f, 0.f}, vh[4] = {0.f, 0.f, 0.f, 0.f};
        const bool kok = (st >> 1) * D1C_NK + kk < K;
#pragma unroll
        for (int j = 0; j < NR; ++j) {
            const float x0 = D1Pair<TI>::lo(xr[j]), x1 = D1Pair<TI>::hi(xr[j]);
            float xhv = 0.f;
            if constexpr (HALO) xhv = DT<TI>::ld((const TI*)&xh[j]);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v0[r] += mk[j][r] * x0; v1[r] += mk[j][r] * x1;
                if constexpr (HALO) vh[r] += mk[j][r] * xhv;
            }
        }
        __syncthreads();                                   // the previous step's V has been consumed
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            V[kk][r][2 * lp + 1] = (cok && kok) ? v0[r] : 0.f;
            V[kk][r][2 * lp + 2] = (cok && kok) ? v1[r] : 0.f;
        }
        if (HALO && (lp == 0 || lp == 63)) {
#pragma unroll
            for (int r = 0; r < 4; ++r) V[kk][r][hidx] = (hok && kok) ? vh[r] : 0.f;
        }
        if (st + 1 < nstep) D1C_LOAD(st + 1);              // in flight under the window sums below
        __syncthreads();
        const float* vr = V[jk][jr];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float wv[NR - 2];
#pragma unroll
            for (int q = 0; q < NR - 2; ++q) wv[q] = vr[wlo[e] + q];
            const float ea = vr[ia[e] >= 0 ? ia[e] : 0], eb = vr[ib[e] >= 0 ? ib[e] : 0];
            float base = 0.f;
#pragma unroll
            for (int q = 0; q < NR - 2; ++q) base += wv[q];
            tacc[e] += base + (ia[e] >= 0 ? ea : 0.f) - (sl[e] ? wv[0] : 0.f) + (ib[e] >= 0 ? eb : 0.f) - (sh[e] ? wv[NR - 3] : 0.f);
        }
        if (st & 1) {                                      // both images of the pair are in: T[(m, Y, X)][(k, r, s)]
            const int kq = (st >> 1) * D1C_NK;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (wok[e] && kq + jk < K) Trow[(size_t)((tid >> 7) + 4 * e) * J + (size_t)kq * 16 + jj] = tacc[e];
                tacc[e] = 0.f;
            }
        }
    }
#undef D1C_LOAD
}

// d ctx[(m, Y, X)][c] = sc * sum_j T[(m, Y, X)][j] * w[(j >> 4) * Ct + Ci + c][j & 15].  16 rows per workgroup of 256 threads;
// thread = (row, one of 16 slices of 64 j): its 64 T values in flight as 16 float4 loads, the weights of its slice from LDS
// ([slice][64 j][CC] + 4 floats of padding per slice: the 16 slices of a wave sit in different banks, the rows broadcast),
// slice sums by lane shuffles.  J = 1024, CC = 12 instantiated.
template <typename TO, int CC>
__global__ __launch_bounds__(256) void d1_ctx_dot_kernel(const float* __restrict__ T, const float* __restrict__ w,
                                                         const float* __restrict__ inv_sigma, TO* __restrict__ dctx, int rows,
                                                         int Ci, int NBNBX) {
    constexpr int JS = 64, NS = 16, J = JS * NS, SST = JS * CC + 4;
    static_assert(CC % 4 == 0, "16-byte weight reads");
    __shared__ __attribute__((aligned(16))) float wl[NS * SST];
    const int tid = threadIdx.x;
    const int Ct = Ci + CC;
    for (int i = tid; i < J * CC; i += 256) {
        const int j = i / CC, c = i - j * CC;
        wl[(j >> 6) * SST + (j & 63) * CC + c] = w[((size_t)(j >> 4) * Ct + Ci + c) * 16 + (j & 15)];
    }
    const int sl = tid & 15, rl = tid >> 4;
    const int row = blockIdx.x * 16 + rl;
    const float* tr = T + (size_t)(row < rows ? row : rows - 1) * J + sl * JS;
    float4 t[JS / 4];
#pragma unroll
    for (int q = 0; q < JS / 4; ++q) t[q] = *(const float4*)(tr + 4 * q);
    __syncthreads();
    float acc[CC];
#pragma unroll
    for (int c = 0; c < CC; ++c) acc[c] = 0.f;
    const float* ws_ = wl + sl * SST;
#pragma unroll
    for (int q = 0; q < JS / 4; ++q) {
        const float tv[4] = {t[q].x, t[q].y, t[q].z, t[q].w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int c4 = 0; c4 < CC / 4; ++c4) {
                const float4 wv = *(const float4*)(ws_ + (4 * q + u) * CC + 4 * c4);
                acc[4 * c4 + 0] += tv[u] * wv.x; acc[4 * c4 + 1] += tv[u] * wv.y;
                acc[4 * c4 + 2] += tv[u] * wv.z; acc[4 * c4 + 3] += tv[u] * wv.w;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CC; ++c) {
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) acc[c] += __shfl_xor(acc[c], o, 16);
    }
    if (sl == 0 && row < rows) {
        const float sc = inv_sigma ? *inv_sigma : 1.f;
        const int mm = row / NBNBX, yx = row - mm * NBNBX;
#pragma unroll
        for (int c = 0; c < CC; ++c) DT<TO>::st(dctx + ((size_t)mm * CC + c) * NBNBX + yx, acc[c] * sc);
    }
}

// Exact Generator chain: the sum of two float32-accurate activations that exist only as split-bf16 images (the head skip
// `x + head` after the residual trunk, generator.py:161).  a3 / b3: (hi, lo, ..) images in layout la / lb (0 = 3C planes,
// 2 = pair groups); writes the nominal bf16 sum y [N,C,HW] and its split image y3 in layout lo.
__device__ __forceinline__ void split_offs(unsigned c, unsigned hw, unsigned C, unsigned C16, unsigned HW, int lay,
                                           unsigned& oh, unsigned& ol, unsigned& per_n) {
    if (lay == 2) { oh = (32u * (c >> 4) + (c & 15u)) * HW + hw; ol = oh + 16u * HW; per_n = 2u * C16 * HW; }
    else { oh = c * HW + hw; ol = oh + C * HW; per_n = 3u * C * HW; }
}
__global__ __launch_bounds__(256) void add_split_kernel(const bf16_t* __restrict__ a3, const bf16_t* __restrict__ b3,
                                                        bf16_t* __restrict__ y, bf16_t* __restrict__ y3, long long total,
                                                        unsigned C, unsigned C16, unsigned HW, int la, int lb, int lo) {
    const unsigned chw = C16 * HW;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long n = i / chw;
        const unsigned rem = (unsigned)(i - n * chw), c = rem / HW, hw = rem - c * HW;
        unsigned oh, ol, pn;
        if (c >= C) {                                   // zero padding channels of the pair layout's last 16-group
            if (lo == 2) { split_offs(c, hw, C, C16, HW, 2, oh, ol, pn); y3[n * pn + oh] = 0; y3[n * pn + ol] = 0; }
            continue;
        }
        split_offs(c, hw, C, C16, HW, la, oh, ol, pn);
        float v = bf2f(a3[n * pn + oh]) + bf2f(a3[n * pn + ol]);
        split_offs(c, hw, C, C16, HW, lb, oh, ol, pn);
        v += bf2f(b3[n * pn + oh]) + bf2f(b3[n * pn + ol]);
        const bf16_t h = f2bf(v), l = f2bf(v - bf2f(h));
        y[(n * C + c) * HW + hw] = h;
        split_offs(c, hw, C, C16, HW, lo, oh, ol, pn);
        y3[n * pn + oh] = h; y3[n * pn + ol] = l;
        if (lo != 2) y3[n * pn + oh + 2u * C * HW] = h;
    }
}

// ---- the scalar loss composition of one training forward as ONE launch (src/model.py:211-220 + :373-376, src/loss/losses.py:8-28)
//   perceptual = mean_b lp[b];  penalty = q > target ? lambda_A : lambda_B
//   total = ((penalty * nbpp + k_M * mse) + k_P * perceptual) [+ beta * g_loss]          (the reference's order of additions)
// aux[0..3] = perceptual, penalty, penalty * nbpp, k_M * mse (logging + backward).  ~12 zero-dimensional ATen launches
// (mean, mul, add, gt, where, full_like) and as many autograd nodes per forward otherwise.
__global__ void loss_combine_fwd_kernel(const float* __restrict__ mse, const float* __restrict__ lp, int B,
                                        const float* __restrict__ nbpp, const float* __restrict__ q,
                                        const float* __restrict__ g_loss, float kM, float kP, float lamA, float lamB,
                                        float target, float beta, float* __restrict__ total, float* __restrict__ aux) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float sp = 0.f;
    for (int b = 0; b < B; ++b) sp += lp[b];
    const float perc = sp / (float)B;
    const float pen = *q > target ? lamA : lamB;
    const float wr = pen * *nbpp, wd = kM * *mse;
    float t = (wr + wd) + kP * perc;
    if (g_loss) t = t + beta * *g_loss;
    *total = t;
    aux[0] = perc; aux[1] = pen; aux[2] = wr; aux[3] = wd;
}
// grads[0] = d total / d mse * g, [1] = .. / d nbpp, [2] = .. / d g_loss, [3 + b] = .. / d lp[b]
__global__ void loss_combine_bwd_kernel(const float* __restrict__ g, const float* __restrict__ aux, int B, float kM, float kP,
                                        float beta, float* __restrict__ grads) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const float gg = *g;
    if (i == 0) grads[0] = kM * gg;
    else if (i == 1) grads[1] = aux[1] * gg;
    else if (i == 2) grads[2] = beta * gg;
    else if (i < 3 + B) grads[i] = kP * gg / (float)B;
}

// Reflect (ReflectionPad2d semantics) or zero padding of [planes, H, W] -> [planes, H + pt + pb, W + pl + pr]: the EVALUATION
// path's pad-to-a-multiple-of-16 (src/helpers/utils.py:50-62) - the training path never materialises a padded tensor.
template <typename T>
__global__ void pad2d_kernel(const T* __restrict__ x, T* __restrict__ y, long long planes, int H, int W, int pt, int pl, int Ho,
                             int Wo, int reflect) {
    const long long total = planes * Ho * Wo;
    EW_LOOP(i, total) {
        const int xo = (int)(i % Wo);
        const long long j = i / Wo;
        const int yo = (int)(j % Ho);
        const long long pc = j / Ho;
        int yi = yo - pt, xi = xo - pl;
        bool in = (unsigned)yi < (unsigned)H && (unsigned)xi < (unsigned)W;
        if (reflect) { yi = reflect_idx(yi, H); xi = reflect_idx(xi, W); in = true; }
        y[i] = in ? x[(pc * H + yi) * W + xi] : (T)0;
    }
}

extern "C" {

#define DISPATCH_T(dtype, CALL_F32, CALL_BF16) \
    do { if ((dtype) == HIFIC_F32) { CALL_F32; } else if ((dtype) == HIFIC_BF16) { CALL_BF16; } else return HIFIC_ERR_ARG; } while (0)

int hific_act_bwd(const void* dy, const void* y, void* dx, long long n, float slope, int dtype, hipStream_t st) {
    if (dtype == HIFIC_BF16 && n % 8 == 0 && (((size_t)dy | (size_t)y | (size_t)dx) & 15) == 0) {
        hipLaunchKernelGGL(act_bwd_bf16x8_kernel, EW_GRID(n / 8), dim3(256), 0, st, (const ew_u32x4_t*)dy, (const ew_u32x4_t*)y,
                           (ew_u32x4_t*)dx, n / 8, slope);
        return hific_launch_status();
    }
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(act_bwd_kernel<float>, EW_GRID(n), dim3(256), 0, st, (const float*)dy, (const float*)y, (float*)dx, n, slope),
        hipLaunchKernelGGL(act_bwd_kernel<bf16_t>, EW_GRID(n), dim3(256), 0, st, (const bf16_t*)dy, (const bf16_t*)y, (bf16_t*)dx, n, slope));
    return hific_launch_status();
}

int hific_tanh_fwd(const void* x, void* y, long long n, int dtype, hipStream_t st) {
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(tanh_fwd_kernel<float>, EW_GRID(n), dim3(256), 0, st, (const float*)x, (float*)y, n),
        hipLaunchKernelGGL(tanh_fwd_kernel<bf16_t>, EW_GRID(n), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, n));
    return hific_launch_status();
}
int hific_tanh_bwd(const void* y, const void* dy, void* dx, long long n, int dtype, hipStream_t st) {
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(tanh_bwd_kernel<float>, EW_GRID(n), dim3(256), 0, st, (const float*)y, (const float*)dy, (float*)dx, n),
        hipLaunchKernelGGL(tanh_bwd_kernel<bf16_t>, EW_GRID(n), dim3(256), 0, st, (const bf16_t*)y, (const bf16_t*)dy, (bf16_t*)dx, n));
    return hific_launch_status();
}
int hific_scale_shift(const void* x, void* y, long long n, float a, float b, int dtype, hipStream_t st) {
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(scale_shift_kernel<float>, EW_GRID(n), dim3(256), 0, st, (const float*)x, (float*)y, n, a, b),
        hipLaunchKernelGGL(scale_shift_kernel<bf16_t>, EW_GRID(n), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, n, a, b));
    return hific_launch_status();
}

int hific_add(const void* a, const void* b, void* o, long long n, int dtype, hipStream_t st) {
    if (dtype == HIFIC_BF16 && n % 8 == 0 && (((size_t)a | (size_t)b | (size_t)o) & 15) == 0) {
        hipLaunchKernelGGL(add_bf16x8_kernel, EW_GRID(n / 8), dim3(256), 0, st, (const ew_u32x4_t*)a, (const ew_u32x4_t*)b,
                           (ew_u32x4_t*)o, n / 8);
        return hific_launch_status();
    }
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(add_kernel<float>, EW_GRID(n), dim3(256), 0, st, (const float*)a, (const float*)b, (float*)o, n),
        hipLaunchKernelGGL(add_kernel<bf16_t>, EW_GRID(n), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)o, n));
    return hific_launch_status();
}

int hific_cast(const void* a, int src_dtype, void* o, int dst_dtype, long long n, hipStream_t st) {
    if (src_dtype == HIFIC_F32 && dst_dtype == HIFIC_BF16)
        hipLaunchKernelGGL((cast_kernel<float, bf16_t>), EW_GRID(n), dim3(256), 0, st, (const float*)a, (bf16_t*)o, n);
    else if (src_dtype == HIFIC_BF16 && dst_dtype == HIFIC_F32)
        hipLaunchKernelGGL((cast_kernel<bf16_t, float>), EW_GRID(n), dim3(256), 0, st, (const bf16_t*)a, (float*)o, n);
    else if (src_dtype == HIFIC_F32 && dst_dtype == HIFIC_F32)
        hipLaunchKernelGGL((cast_kernel<float, float>), EW_GRID(n), dim3(256), 0, st, (const float*)a, (float*)o, n);
    else if (src_dtype == HIFIC_BF16 && dst_dtype == HIFIC_BF16)
        hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), EW_GRID(n), dim3(256), 0, st, (const bf16_t*)a, (bf16_t*)o, n);
    else return HIFIC_ERR_ARG;
    return hific_launch_status();
}

// which 0: activation layout (hi, lo, hi); 1: weight layout (hi, hi, lo); 2: pair layout [outer][2 * C16][inner] of the
// native split kernels (split_pair_kernel; activations and weights alike).  dst_dtype bf16 (activations) or f32 (a
// derived weight tensor whose values are exactly bf16-representable, fed to the ordinary weight-pack path).
int hific_split3(const float* src, void* dst, long long outer, int C, long long inner, int which, int dst_dtype,
                 hipStream_t st) {
    if (!src || !dst || outer <= 0 || C <= 0 || inner <= 0 || which < 0 || which > 2) return HIFIC_ERR_ARG;
    if ((long long)C * inner >= (1ll << 31)) return HIFIC_ERR_UNSUPPORTED;
    if (which == 2) {       // pair layout: dst [outer][2 * C16][inner]
        const long long C16 = ((long long)C + 15) / 16 * 16;
        if (2 * C16 * inner >= (1ll << 31)) return HIFIC_ERR_UNSUPPORTED;
        const long long tot = outer * C16 * inner;
        if (dst_dtype == HIFIC_BF16)
            hipLaunchKernelGGL(split_pair_kernel<bf16_t>, EW_GRID(tot), dim3(256), 0, st, src, (bf16_t*)dst, tot, (unsigned)C, (unsigned)C16, (unsigned)inner);
        else if (dst_dtype == HIFIC_F32)
            hipLaunchKernelGGL(split_pair_kernel<float>, EW_GRID(tot), dim3(256), 0, st, src, (float*)dst, tot, (unsigned)C, (unsigned)C16, (unsigned)inner);
        else return HIFIC_ERR_ARG;
        return hific_launch_status();
    }
    const long long total = outer * C * inner;
    if (dst_dtype == HIFIC_BF16 && inner % 4 == 0 && (((size_t)src & 15) | ((size_t)dst & 7)) == 0) {
        if (which == 0) hipLaunchKernelGGL(split3_bf16x4_kernel<0>, EW_GRID(total / 4), dim3(256), 0, st, (const float4*)src, (uint2*)dst, total / 4, (unsigned)C, (unsigned)(inner / 4));
        else hipLaunchKernelGGL(split3_bf16x4_kernel<1>, EW_GRID(total / 4), dim3(256), 0, st, (const float4*)src, (uint2*)dst, total / 4, (unsigned)C, (unsigned)(inner / 4));
        return hific_launch_status();
    }
    if (dst_dtype == HIFIC_BF16) {
        if (which == 0) hipLaunchKernelGGL((split3_kernel<bf16_t, 0>), EW_GRID(total), dim3(256), 0, st, src, (bf16_t*)dst, total, (unsigned)C, (unsigned)inner);
        else hipLaunchKernelGGL((split3_kernel<bf16_t, 1>), EW_GRID(total), dim3(256), 0, st, src, (bf16_t*)dst, total, (unsigned)C, (unsigned)inner);
    } else if (dst_dtype == HIFIC_F32) {
        if (which == 0) hipLaunchKernelGGL((split3_kernel<float, 0>), EW_GRID(total), dim3(256), 0, st, src, (float*)dst, total, (unsigned)C, (unsigned)inner);
        else hipLaunchKernelGGL((split3_kernel<float, 1>), EW_GRID(total), dim3(256), 0, st, src, (float*)dst, total, (unsigned)C, (unsigned)inner);
    } else return HIFIC_ERR_ARG;
    return hific_launch_status();
}

int hific_add_split(const void* a3, int la, const void* b3, int lb, void* y, void* y3, int lo, int N, int C, int HW,
                    hipStream_t st) {
    if (!a3 || !b3 || !y || !y3 || N <= 0 || C <= 0 || HW <= 0) return HIFIC_ERR_ARG;
    if ((la != 0 && la != 2) || (lb != 0 && lb != 2) || (lo != 0 && lo != 2)) return HIFIC_ERR_ARG;
    const long long C16 = ((long long)C + 15) / 16 * 16;
    if ((long long)3 * C * HW >= (1ll << 31) || 2 * C16 * HW >= (1ll << 31)) return HIFIC_ERR_UNSUPPORTED;
    const long long tot = (long long)N * C16 * HW;
    hipLaunchKernelGGL(add_split_kernel, EW_GRID(tot), dim3(256), 0, st, (const bf16_t*)a3, (const bf16_t*)b3, (bf16_t*)y,
                       (bf16_t*)y3, tot, (unsigned)C, (unsigned)C16, (unsigned)HW, la, lb, lo);
    return hific_launch_status();
}

int hific_loss_combine_fwd(const float* mse, const float* lp, int B, const float* nbpp, const float* q, const float* g_loss,
                           float kM, float kP, float lamA, float lamB, float target, float beta, float* total, float* aux,
                           hipStream_t st) {
    if (!mse || !lp || B <= 0 || !nbpp || !q || !total || !aux) return HIFIC_ERR_ARG;
    hipLaunchKernelGGL(loss_combine_fwd_kernel, dim3(1), dim3(64), 0, st, mse, lp, B, nbpp, q, g_loss, kM, kP, lamA, lamB, target,
                       beta, total, aux);
    return hific_launch_status();
}
int hific_loss_combine_bwd(const float* g, const float* aux, int B, float kM, float kP, float beta, float* grads,
                           hipStream_t st) {
    if (!g || !aux || B <= 0 || !grads) return HIFIC_ERR_ARG;
    hipLaunchKernelGGL(loss_combine_bwd_kernel, dim3(cdiv(3 + B, 64)), dim3(64), 0, st, g, aux, B, kM, kP, beta, grads);
    return hific_launch_status();
}

int hific_pad2d(const void* x, void* y, long long planes, int H, int W, int pt, int pl, int pb, int pr, int reflect, int dtype,
                hipStream_t st) {
    if (!x || !y || planes <= 0 || H <= 0 || W <= 0 || pt < 0 || pl < 0 || pb < 0 || pr < 0) return HIFIC_ERR_ARG;
    if (reflect && (pt >= H || pb >= H || pl >= W || pr >= W)) return HIFIC_ERR_ARG;
    const int Ho = H + pt + pb, Wo = W + pl + pr;
    const long long total = planes * Ho * Wo;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(pad2d_kernel<float>, EW_GRID(total), dim3(256), 0, st, (const float*)x, (float*)y, planes, H, W, pt, pl, Ho, Wo, reflect),
        hipLaunchKernelGGL(pad2d_kernel<bf16_t>, EW_GRID(total), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, planes, H, W, pt, pl, Ho, Wo, reflect));
    return hific_launch_status();
}

int hific_axpby_f32(const float* a, const float* b, float* o, float alpha, float beta, long long n, hipStream_t st) {
    hipLaunchKernelGGL(axpby_kernel, EW_GRID(n), dim3(256), 0, st, a, b, o, alpha, beta, n);
    return hific_launch_status();
}

// out[c] (=|+=) sum_{n,hw} x[n,c,hw]; ws >= 64*C floats
int hific_channel_sum(const void* x, float* out, int N, int C, int HW, int accumulate, int dtype, void* ws,
                      size_t ws_bytes, hipStream_t st) {
    int nsplit = cdiv(1024, C); if (nsplit > 64) nsplit = 64;
    if (nsplit * 256 > HW) nsplit = cdiv(HW, 256);
    if (nsplit < 1) nsplit = 1;
    if ((size_t)nsplit * C * sizeof(float) > ws_bytes) return HIFIC_ERR_WS;
    float* part = (float*)ws;
    unsigned* tk = hific_tickets(st, 1);
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(chan_sum_kernel<float>, dim3(C, nsplit), dim3(256), 0, st, (const float*)x, part, N, C, HW, nsplit, tk, out, accumulate),
        hipLaunchKernelGGL(chan_sum_kernel<bf16_t>, dim3(C, nsplit), dim3(256), 0, st, (const bf16_t*)x, part, N, C, HW, nsplit, tk, out, accumulate));
    if (!tk) hipLaunchKernelGGL(chan_sum_reduce_kernel, dim3(cdiv(C, 256)), dim3(256), 0, st, part, out, C, nsplit, accumulate);
    return hific_launch_status();
}

int hific_maxpool3s2_fwd(const void* x, void* y, long long planes, int H, int W, int dtype, hipStream_t st) {
    const int OH = (H - 3) / 2 + 1, OW = (W - 3) / 2 + 1;
    const long long total = planes * OH * OW;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(maxpool3s2_fwd_kernel<float>, EW_GRID(total), dim3(256), 0, st, (const float*)x, (float*)y, planes, H, W, OH, OW),
        hipLaunchKernelGGL(maxpool3s2_fwd_kernel<bf16_t>, EW_GRID(total), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, planes, H, W, OH, OW));
    return hific_launch_status();
}
int hific_maxpool2s2_fwd(const void* x, void* y, long long planes, int H, int W, int dtype, hipStream_t st) {
    if (H < 2 || W < 2 || planes < 0) return HIFIC_ERR_ARG;
    const int OH = H / 2, OW = W / 2;
    const long long total = planes * OH * OW;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(maxpool2s2_fwd_kernel<float>, EW_GRID(total), dim3(256), 0, st, (const float*)x, (float*)y, planes, H, W, OH, OW),
        hipLaunchKernelGGL(maxpool2s2_fwd_kernel<bf16_t>, EW_GRID(total), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, planes, H, W, OH, OW));
    return hific_launch_status();
}
int hific_maxpool2s2_bwd(const void* x, const void* dy, void* dx, long long planes, int H, int W, int dtype,
                         hipStream_t st) {
    if (H < 2 || W < 2 || planes < 0) return HIFIC_ERR_ARG;
    const int OH = H / 2, OW = W / 2;
    const long long total = planes * H * W;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(maxpool2s2_bwd_kernel<float>, EW_GRID(total), dim3(256), 0, st, (const float*)x, (const float*)dy, (float*)dx, planes, H, W, OH, OW),
        hipLaunchKernelGGL(maxpool2s2_bwd_kernel<bf16_t>, EW_GRID(total), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, planes, H, W, OH, OW));
    return hific_launch_status();
}
int hific_maxpool3s2_bwd(const void* x, const void* dy, void* dx, long long planes, int H, int W, int dtype,
                         hipStream_t st) {
    const int OH = (H - 3) / 2 + 1, OW = (W - 3) / 2 + 1;
    const long long total = planes * H * W;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(maxpool3s2_bwd_kernel<float>, EW_GRID(total), dim3(256), 0, st, (const float*)x, (const float*)dy, (float*)dx, planes, H, W, OH, OW),
        hipLaunchKernelGGL(maxpool3s2_bwd_kernel<bf16_t>, EW_GRID(total), dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, planes, H, W, OH, OW));
    return hific_launch_status();
}

// out[0] = mean((scale*a - scale*b)^2); a dtype, b f32 (the input image); ws >= 1024 floats
int hific_mse_fwd(const void* a, const float* b, float* out, long long n, float scale, int dtype, void* ws,
                  size_t ws_bytes, hipStream_t st) {
    const int nb = 1024;
    if (ws_bytes < nb * sizeof(float)) return HIFIC_ERR_WS;
    float* part = (float*)ws;
    unsigned* tk = hific_tickets(st, 1);
    const float mul = 1.f / (float)n;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(sqdiff_partial_kernel<float>, dim3(nb), dim3(256), 0, st, (const float*)a, b, part, n, scale, tk, mul, out),
        hipLaunchKernelGGL(sqdiff_partial_kernel<bf16_t>, dim3(nb), dim3(256), 0, st, (const bf16_t*)a, b, part, n, scale, tk, mul, out));
    if (!tk) hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, st, part, nb, mul, out);
    return hific_launch_status();
}
// da = (*g) * 2*scale^2*(a-b)/n
int hific_mse_bwd(const void* a, const float* b, const float* g, void* da, long long n, float scale, int dtype,
                  hipStream_t st) {
    const float coef = 2.f * scale * scale / (float)n;
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(sqdiff_bwd_kernel<float>, EW_GRID(n), dim3(256), 0, st, (const float*)a, b, g, (float*)da, n, coef),
        hipLaunchKernelGGL(sqdiff_bwd_kernel<bf16_t>, EW_GRID(n), dim3(256), 0, st, (const bf16_t*)a, b, g, (bf16_t*)da, n, coef));
    return hific_launch_status();
}

// out[0] = mean BCEWithLogits(z, target); ws >= 256 floats
int hific_bce_fwd(const float* z, float target, float* out, long long n, void* ws, size_t ws_bytes, hipStream_t st) {
    const int nb = 256;
    if (ws_bytes < nb * sizeof(float)) return HIFIC_ERR_WS;
    float* part = (float*)ws;
    unsigned* tk = hific_tickets(st, 1);
    hipLaunchKernelGGL(bce_partial_kernel, dim3(nb), dim3(256), 0, st, z, target, part, n, tk, 1.f / (float)n, out);
    if (!tk) hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, st, part, nb, 1.f / (float)n, out);
    return hific_launch_status();
}
int hific_bce_bwd(const float* z, float target, const float* g, float* dz, long long n, int accumulate,
                  hipStream_t st) {
    hipLaunchKernelGGL(bce_bwd_kernel, EW_GRID(n), dim3(256), 0, st, z, target, g, dz, n, 1.f / (float)n, accumulate);
    return hific_launch_status();
}
// out[0] = mean (sigmoid(z) - target)^2 (losses.py:43-50 on D_real / D_gen = sigmoid(logits)); ws >= 256 floats
int hific_lsq_sigmoid_fwd(const float* z, float target, float* out, long long n, void* ws, size_t ws_bytes,
                          hipStream_t st) {
    const int nb = 256;
    if (ws_bytes < nb * sizeof(float)) return HIFIC_ERR_WS;
    float* part = (float*)ws;
    unsigned* tk = hific_tickets(st, 1);
    hipLaunchKernelGGL(lsq_partial_kernel, dim3(nb), dim3(256), 0, st, z, target, part, n, tk, 1.f / (float)n, out);
    if (!tk) hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, st, part, nb, 1.f / (float)n, out);
    return hific_launch_status();
}
int hific_lsq_sigmoid_bwd(const float* z, float target, const float* g, float* dz, long long n, int accumulate,
                          hipStream_t st) {
    hipLaunchKernelGGL(lsq_bwd_kernel, EW_GRID(n), dim3(256), 0, st, z, target, g, dz, n, 1.f / (float)n, accumulate);
    return hific_launch_status();
}
int hific_sigmoid_f32(const float* z, float* o, long long n, hipStream_t st) {
    hipLaunchKernelGGL(sigmoid_kernel, EW_GRID(n), dim3(256), 0, st, z, o, n);
    return hific_launch_status();
}

int hific_upcat_fwd(const void* img, const void* ctx, void* out, int N, int Ci, int Cc, int H, int W, int f, int dtype,
                    hipStream_t st) {
    if (H % f || W % f) return HIFIC_ERR_ARG;
    const long long total = (long long)N * (Ci + Cc) * H * W;
    if (dtype == HIFIC_BF16 && W % 8 == 0 && f % 8 == 0 && total / 8 < (1ll << 31) &&
        (((size_t)img | (size_t)out) & 15) == 0) {
        hipLaunchKernelGGL(upcat_fwd_vec_kernel, EW_GRID(total / 8), dim3(256), 0, st, (const bf16_t*)img, (const bf16_t*)ctx,
                           (bf16_t*)out, (unsigned)N, Ci, Cc, H, W, f);
        return hific_launch_status();
    }
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(upcat_fwd_kernel<float>, EW_GRID(total), dim3(256), 0, st, (const float*)img, (const float*)ctx, (float*)out, N, Ci, Cc, H, W, f),
        hipLaunchKernelGGL(upcat_fwd_kernel<bf16_t>, EW_GRID(total), dim3(256), 0, st, (const bf16_t*)img, (const bf16_t*)ctx, (bf16_t*)out, N, Ci, Cc, H, W, f));
    return hific_launch_status();
}
// dimg: [nimg,Ci,H,W] taken from images [n0, n0+nimg) of dout (pass nimg=0 to skip); dctx: [N,Cc,H/f,W/f] or null
int hific_upcat_bwd(const void* dout, void* dimg, int n0, int nimg, void* dctx, int N, int Ci, int Cc, int H, int W,
                    int f, int dtype, hipStream_t st) {
    if (dimg && nimg > 0) {
        const long long total = (long long)nimg * Ci * H * W;
        DISPATCH_T(dtype,
            hipLaunchKernelGGL(upcat_bwd_img_kernel<float>, EW_GRID(total), dim3(256), 0, st, (const float*)dout, (float*)dimg, n0, nimg, Ci, Cc, H, W),
            hipLaunchKernelGGL(upcat_bwd_img_kernel<bf16_t>, EW_GRID(total), dim3(256), 0, st, (const bf16_t*)dout, (bf16_t*)dimg, n0, nimg, Ci, Cc, H, W));
    }
    if (dctx) {
        const long long total = (long long)N * Cc * (H / f) * (W / f) * 64;
        DISPATCH_T(dtype,
            hipLaunchKernelGGL(upcat_bwd_ctx_kernel<float>, EW_GRID(total), dim3(256), 0, st, (const float*)dout, (float*)dctx, N, Ci, Cc, H, W, f),
            hipLaunchKernelGGL(upcat_bwd_ctx_kernel<bf16_t>, EW_GRID(total), dim3(256), 0, st, (const bf16_t*)dout, (bf16_t*)dctx, N, Ci, Cc, H, W, f));
    }
    return hific_launch_status();
}

// out [2B, Ci+Cc, H, W] from real [B,Ci,H,W], gen [B,Ci,H,W] and ctx [B,Cc,H/f,W/f] (image n reads ctx[n >> 1])
int hific_upcat_pair_fwd(const void* real, const void* gen, const void* ctx, void* out, int B, int Ci, int Cc, int H, int W,
                         int f, int dtype, hipStream_t st) {
    if (!real || !gen || !ctx || !out || B <= 0 || f <= 0 || H % f || W % f) return HIFIC_ERR_ARG;
    const long long total = 2ll * B * (Ci + Cc) * H * W;
    if (dtype == HIFIC_BF16 && W % 8 == 0 && f % 8 == 0 && total / 8 < (1ll << 32) &&
        !((((size_t)real) | ((size_t)gen) | ((size_t)out)) & 15)) {
        hipLaunchKernelGGL(upcat_pair_fwd_kernel, EW_GRID(total / 8), dim3(256), 0, st, (const bf16_t*)real, (const bf16_t*)gen,
                           (const bf16_t*)ctx, (bf16_t*)out, (unsigned)B, Ci, Cc, H, W, f);
        return hific_launch_status();
    }
    DISPATCH_T(dtype,
        hipLaunchKernelGGL(upcat_pair_fwd_elem_kernel<float>, EW_GRID(total), dim3(256), 0, st, (const float*)real, (const float*)gen, (const float*)ctx, (float*)out, B, Ci, Cc, H, W, f),
        hipLaunchKernelGGL(upcat_pair_fwd_elem_kernel<bf16_t>, EW_GRID(total), dim3(256), 0, st, (const bf16_t*)real, (const bf16_t*)gen, (const bf16_t*)ctx, (bf16_t*)out, B, Ci, Cc, H, W, f));
    return hific_launch_status();
}
// dgen [B,Ci,H,W] = dout[B:, :Ci] (null: skipped - the D-turn detaches the generated images), dctx [B,Cc,H/f,W/f] (null: skipped)
int hific_upcat_pair_bwd(const void* dout, void* dgen, void* dctx, int B, int Ci, int Cc, int H, int W, int f, int dtype,
                         hipStream_t st) {
    if (!dout || B <= 0 || f <= 0) return HIFIC_ERR_ARG;
    if (dgen) {
        const long long total = (long long)B * Ci * H * W;
        DISPATCH_T(dtype,
            hipLaunchKernelGGL(upcat_bwd_img_kernel<float>, EW_GRID(total), dim3(256), 0, st, (const float*)dout, (float*)dgen, B, B, Ci, Cc, H, W),
            hipLaunchKernelGGL(upcat_bwd_img_kernel<bf16_t>, EW_GRID(total), dim3(256), 0, st, (const bf16_t*)dout, (bf16_t*)dgen, B, B, Ci, Cc, H, W));
    }
    if (dctx) {
        const long long total = (long long)B * Cc * (H / f) * (W / f) * 64;
        DISPATCH_T(dtype,
            hipLaunchKernelGGL(upcat_pair_bwd_ctx_kernel<float>, EW_GRID(total), dim3(256), 0, st, (const float*)dout, (float*)dctx, B, Ci, Cc, H, W, f),
            hipLaunchKernelGGL(upcat_pair_bwd_ctx_kernel<bf16_t>, EW_GRID(total), dim3(256), 0, st, (const bf16_t*)dout, (bf16_t*)dctx, B, Ci, Cc, H, W, f));
    }
    return hific_launch_status();
}

// dz [2B, K, H/2, W/2] (gradient at conv1's pre-activation), w [K, Ci+Cc, 4, 4] float32 (weight_orig), inv_sigma: device scalar
// or null, dctx [B, Cc, H/f, W/f].  Geometry fixed by the reference: 4x4 window, stride 2, reflect pad 1; f even, W/2 <= 128.
int hific_d1_ctx_grad(const void* dz, const float* w, const float* inv_sigma, void* dctx, int B, int K, int Ci, int Cc, int H,
                      int W, int f, int dtype, void* ws, size_t ws_bytes, hipStream_t st) {
    if (!dz || !w || !dctx || B <= 0 || K <= 0 || Cc <= 0 || f < 4 || (f & 1) || H % f || W % f || (H & 1) || (W & 1))
        return HIFIC_ERR_ARG;
    const int OH = H / 2, OW = W / 2;
    if (f != 16 || 16 * Cc > 256 || H / f < 2 || W / f < 2) return HIFIC_ERR_UNSUPPORTED;     // instantiated: NR = f / 2 + 2 = 10
    const size_t J = (size_t)K * 16, rows = (size_t)B * (H / f) * (W / f);
    if (K != 64 || Cc != 12) return HIFIC_ERR_UNSUPPORTED;    // d1_ctx_dot_kernel is instantiated for the reference's layer
    if (!ws || ws_bytes < rows * J * sizeof(float)) return HIFIC_ERR_WS;
    float* T = (float*)ws;
    const dim3 grid(H / f, B, cdiv(OW, D1C_MAXW));
    const bool halo = OW > D1C_MAXW;
#define D1C_GO(TI_, NR_, HALO_)                                                                                             \
    hipLaunchKernelGGL((d1_ctx_win_kernel<TI_, NR_, HALO_>), grid, dim3(512), 0, st, (const TI_*)dz, T, B, K, OH, OW, f)
#define D1C_PICK(TI_)                                                                                                       \
    do { if (halo) D1C_GO(TI_, 10, true); else D1C_GO(TI_, 10, false); } while (0)
    DISPATCH_T(dtype, D1C_PICK(float), D1C_PICK(bf16_t));
#undef D1C_PICK
#undef D1C_GO
    const int nb = (int)((rows + 15) / 16);
    DISPATCH_T(dtype,
        hipLaunchKernelGGL((d1_ctx_dot_kernel<float, 12>), dim3(nb), dim3(256), 0, st, T, w, inv_sigma, (float*)dctx, (int)rows, Ci, (H / f) * (W / f)),
        hipLaunchKernelGGL((d1_ctx_dot_kernel<bf16_t, 12>), dim3(nb), dim3(256), 0, st, T, w, inv_sigma, (bf16_t*)dctx, (int)rows, Ci, (H / f) * (W / f)));
    return hific_launch_status();
}

// n <= 8 layers per call: W[i] f32 [K[i], M[i]], u[i] [K[i]], v[i] [M[i]] updated in place (do_iter bit 0), sig[i] = [sigma, 1/sigma].
// do_iter bit 1: sig[i] has 2 + K[i] + M[i] floats and also receives copies of the post-iteration u[i], v[i].
// Same arithmetic per layer as hific_spectral_norm_fwd (bit-identical), 6 launches for the whole set.
int hific_spectral_norm_fwd_batch(const float* const* W, float* const* u, float* const* v, float* const* sig, const int* K,
                                  const int* M, int n, int do_iter, float eps, void* ws, size_t ws_bytes, hipStream_t st) {
    if (n <= 0 || n > SN_MAXJOBS || !W || !u || !v || !sig || !K || !M) return HIFIC_ERR_ARG;
    SnJobs J;
    float* wp = (float*)ws;
    size_t used = 0;
    int maxK = 0, maxM = 0;
    for (int i = 0; i < n; ++i) {
        if (!W[i] || !u[i] || !v[i] || !sig[i] || K[i] <= 0 || M[i] <= 0) return HIFIC_ERR_ARG;
        const int nslice = cdiv(K[i], SN_ROWS);
        const size_t need = (size_t)M[i] + K[i] + (size_t)nslice * M[i];
        if ((used + need) * sizeof(float) > ws_bytes) return HIFIC_ERR_WS;
        J.j[i].W = W[i]; J.j[i].u = u[i]; J.j[i].v = v[i]; J.j[i].sig = sig[i]; J.j[i].K = K[i]; J.j[i].M = M[i];
        J.j[i].tmpM = wp + used; J.j[i].tmpK = J.j[i].tmpM + M[i]; J.j[i].part = J.j[i].tmpK + K[i];
        used += need;
        if (K[i] > maxK) maxK = K[i];
        if (M[i] > maxM) maxM = M[i];
    }
    for (int i = n; i < SN_MAXJOBS; ++i) J.j[i] = J.j[0];
    const int snap = (do_iter >> 1) & 1;
    do_iter &= 1;
    if (do_iter) {
        hipLaunchKernelGGL(sn_wtu_batch_kernel, dim3(cdiv(maxM, 256), cdiv(maxK, SN_ROWS), n), dim3(256), 0, st, J);
        hipLaunchKernelGGL(sn_colsum_batch_kernel, dim3(cdiv(maxM, 256), n), dim3(256), 0, st, J);
        hipLaunchKernelGGL(sn_normalize_batch_kernel, dim3(n), dim3(256), 0, st, J, 0, eps);
        hipLaunchKernelGGL(sn_wv_batch_kernel, dim3(maxK, n), dim3(256), 0, st, J);
        hipLaunchKernelGGL(sn_normalize_batch_kernel, dim3(n), dim3(256), 0, st, J, 1, eps);
    } else {
        hipLaunchKernelGGL(sn_wv_batch_kernel, dim3(maxK, n), dim3(256), 0, st, J);
    }
    hipLaunchKernelGGL(sn_sigma_batch_kernel, dim3(n), dim3(256), 0, st, J, snap);
    return hific_launch_status();
}

// One power iteration in place on (u [K], v [M]); sigma_out[0]=sigma, [1]=1/sigma. ws >= (K+M) floats.
// do_iter=0 (eval mode): only sigma from the stored u, v.
int hific_spectral_norm_fwd(const float* W, float* u, float* v, float* sigma_out, int K, int M, int do_iter, float eps,
                            void* ws, size_t ws_bytes, hipStream_t st) {
    if (ws_bytes < (size_t)(K + M) * sizeof(float)) return HIFIC_ERR_WS;
    float* tmpM = (float*)ws;
    float* tmpK = tmpM + M;
    if (do_iter) {
        const int nslice = cdiv(K, SN_ROWS);
        float* part = tmpK + K;
        if (((size_t)M + K + (size_t)nslice * M) * sizeof(float) > ws_bytes) return HIFIC_ERR_WS;
        hipLaunchKernelGGL(sn_wtu_kernel, dim3(cdiv(M, 256), nslice), dim3(256), 0, st, W, u, part, K, M);
        hipLaunchKernelGGL(sn_colsum_kernel, dim3(cdiv(M, 256)), dim3(256), 0, st, part, tmpM, nslice, M);
        hipLaunchKernelGGL(sn_normalize_kernel, dim3(1), dim3(256), 0, st, tmpM, v, M, eps);
        hipLaunchKernelGGL(sn_wv_kernel, dim3(K), dim3(256), 0, st, W, v, tmpK, K, M);
        hipLaunchKernelGGL(sn_normalize_kernel, dim3(1), dim3(256), 0, st, tmpK, u, K, eps);
    } else {
        hipLaunchKernelGGL(sn_wv_kernel, dim3(K), dim3(256), 0, st, W, v, tmpK, K, M);
    }
    hipLaunchKernelGGL(sn_sigma_kernel, dim3(1), dim3(256), 0, st, u, tmpK, sigma_out, K);
    return hific_launch_status();
}
// dWorig (=|+=) (dW - (sum dW*Worig)/sigma * u v^T)/sigma ; ws >= 257 floats
int hific_spectral_norm_bwd(const float* dW, const float* Worig, const float* u, const float* v, const float* sigma,
                            float* dWorig, int K, int M, int accumulate, void* ws, size_t ws_bytes, hipStream_t st) {
    const int nb = 256;
    if (ws_bytes < (nb + 1) * sizeof(float)) return HIFIC_ERR_WS;
    float* part = (float*)ws;
    float* dot = part + nb;
    const long long n = (long long)K * M;
    unsigned* tk = hific_tickets(st, 1);
    hipLaunchKernelGGL(sn_dot_partial_kernel, dim3(nb), dim3(256), 0, st, dW, Worig, part, n, tk, 1.f, dot);
    if (!tk) hipLaunchKernelGGL(final_sum_kernel, dim3(1), dim3(256), 0, st, part, nb, 1.f, dot);
    hipLaunchKernelGGL(sn_bwd_kernel, EW_GRID(n), dim3(256), 0, st, dW, u, v, sigma, dot, dWorig, K, M, accumulate);
    return hific_launch_status();
}

// torch.optim.Adam step t (1-based) over flat f32 buffers; grad_scale multiplies g first (1/world for DDP means)
int hific_adam_step(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                    float eps, int step, float grad_scale, hipStream_t st) {
    if (step < 1) return HIFIC_ERR_ARG;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    hipLaunchKernelGGL(adam_kernel, EW_GRID(n), dim3(256), 0, st, p, g, m, v, n, lr, beta1, beta2, eps, (float)bc1,
                       (float)sqrt(bc2), grad_scale);
    return hific_launch_status();
}

// Graph-replayable form: the step count lives in device memory.  hific_adam_prepare: ++*step_dev, bc_dev[0] = 1 - b1^t,
// bc_dev[1] = sqrt(1 - b2^t); hific_adam_apply: the update of one parameter range with those factors (a group may be
// updated in several ranges / on several streams after ONE prepare).
int hific_adam_prepare(int* step_dev, float* bc_dev, float beta1, float beta2, hipStream_t st) {
    if (!step_dev || !bc_dev) return HIFIC_ERR_ARG;
    hipLaunchKernelGGL(adam_prep_kernel, dim3(1), dim3(64), 0, st, step_dev, bc_dev, beta1, beta2);
    return hific_launch_status();
}
int hific_adam_apply(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2,
                     float eps, const float* bc_dev, float grad_scale, hipStream_t st) {
    if (!p || !g || !m || !v || !bc_dev || n <= 0) return HIFIC_ERR_ARG;
    // 16-byte accesses when the range allows it (ParamArena slices are 256-byte aligned multiples of 64 elements): the same
    // per-element arithmetic in the same order, four elements per thread and trip
    if (n % 4 == 0 && ((((size_t)p | (size_t)g | (size_t)m | (size_t)v) & 15) == 0)) {
        hipLaunchKernelGGL(adam_dev4_kernel, EW_GRID(n / 4), dim3(256), 0, st, (float4*)p, (const float4*)g, (float4*)m,
                           (float4*)v, n / 4, lr, beta1, beta2, eps, bc_dev, grad_scale);
        return hific_launch_status();
    }
    hipLaunchKernelGGL(adam_dev_kernel, EW_GRID(n), dim3(256), 0, st, p, g, m, v, n, lr, beta1, beta2, eps, bc_dev,
                       grad_scale);
    return hific_launch_status();
}

}  // extern "C"
