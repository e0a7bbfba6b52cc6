// Weight gradient of a stride-1 convolution with <= 4 INPUT channels on a big plane (gfx950): the first Encoder layer, 3 -> 60,
// 7x7, reflect pad 3 on 256 x 256 (reference: src/network/encoder.py:56-62 through autograd).  It is the LAST kernel of the G-turn's
// backward pass - the optimizer waits for it (tools/r05/tail.py) - and wgrad_im2col_kernel spends 168-205 us on it: per 128-pixel
// tile a synchronous stage of dY, an im2col image gathered element by element through LDS, four barrier-separated groups of 8 MFMAs.
//
// Here both operands stay in their natural NCHW order (pixels contiguous = the MFMA reduction dimension):
//   dW[k][c][r][s] = sum_{n, y, x} dY[n][k][y][x] * xpad[n][c][y - pt + r][x - pl + s]
//   * rows of the MFMA = k (<= 64), columns = (c, r) (C R <= 32), one accumulator per kernel column s: the B fragment of column s
//     is the x row (c, y - pt + r) shifted by s pixels - two aligned 16-byte LDS reads per lane shared by all s, then a funnel
//     shift by a compile-time amount per s (no im2col image, no gather);
//   * a workgroup is persistent over image rows (n, y); waves 4-7 stage the next row's dY (64 x OW, 16-byte loads) and its C R
//     padded x rows (padding rule applied here) while waves 0-3 = (row block, half of the s range) run this row's MFMAs;
//   * partial sums per workgroup to the workspace; wgrad_c3_finalize_kernel adds them into dW[K][C][R][S].
#include "gconv.h"
#include "gconv_dev.h"
#include <stdio.h>
#include <string.h>

struct WgC3Params {
    const bf16_t* dy;        // [N, K, OH, OW] bf16
    const void* x;           // [N, C, H, W] float32 or bf16
    float* ws;               // [nwg][S][64][32] partial sums
    int N, K, C, H, W, OH, OW, R, S, pt, pl, bmode;
    int rows_per_wg, nrows;  // image rows (n, y) per workgroup / in total
    int apitch, xpitch;      // bytes per LDS row of the dY tile / of an x row
    int xcols;               // columns of an LDS x row (image column j - pl)
    int dbg;                 // timing ablations (HIFIC_DBG): 1 = no staging after the first row, 2 = no MFMA loop
};

typedef unsigned int c3_u32x4_t __attribute__((ext_vector_type(4)));

// B fragment of kernel column S_ from the 16 row elements (w8: 8 dwords) that start at the fragment's aligned pixel group
template <int S_>
__device__ __forceinline__ bf16x8_t c3_frag(const unsigned (&w8)[8]) {
    constexpr int sd = S_ >> 1;
    c3_u32x4_t f;
#pragma unroll
    for (int d = 0; d < 4; ++d)
        f[d] = (S_ & 1) ? __builtin_amdgcn_alignbit(w8[(sd + d + 1) & 7], w8[sd + d], 16) : w8[sd + d];
    return __builtin_bit_cast(bf16x8_t, f);
}

template <bool XF32>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void wgrad_c3_kernel(const WgC3Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const bool loader = threadIdx.x >= 256;
    const int tid = loader ? (int)threadIdx.x - 256 : (int)threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    // LDS: two dY tiles [64 rows][apitch] + ONE ring of padded x rows: per channel 24 slots (16 used) = 2 image parities x 8 logical rows
    // (logical row t = y - pt + r of image n lives in slot (n & 1) * 8 + (t & 7): a tile reads R <= 7 of the 8 slots of its image's
    // half, the loader writes the eighth - or, at an image boundary, the other half), + one all-zero row for the unused MFMA columns
    const unsigned abytes = 64u * (unsigned)p.apitch;
    unsigned char* xring = smem + 2 * abytes;
    const int zrow = p.C * 24;
    const int row_lo = (int)blockIdx.x * p.rows_per_wg;
    const int row_hi = row_lo + p.rows_per_wg < p.nrows ? row_lo + p.rows_per_wg : p.nrows;
    if (row_lo >= row_hi) return;
    const int CR = p.C * p.R;
    for (unsigned i = threadIdx.x; i < (unsigned)p.xpitch / 4u; i += 512) *(unsigned*)(xring + (size_t)zrow * p.xpitch + i * 4u) = 0u;

    if (loader) {
        const int npc = p.OW >> 3;                          // 16-byte pieces per dY row
        const int npieces = 64 * npc;
        const size_t aplane = (size_t)p.OH * p.OW;
        const size_t xplane = (size_t)p.H * p.W;
        // static per thread (the same for every row): its 8 dY pieces (row k, 16-byte column piece) and its <= 4 elements of the one
        // new x row per channel - no integer division in the steady-state path
        unsigned a_g[8], a_l[8];
        bool a_ok[8], a_in[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int pc = tid + 256 * j;
            const int k = pc / npc, c8 = pc - k * npc;
            a_in[j] = pc < npieces; a_ok[j] = a_in[j] && k < p.K;
            a_g[j] = a_ok[j] ? (unsigned)((size_t)k * aplane + c8 * 8) : 0u;
            a_l[j] = (unsigned)(k * p.apitch + c8 * 16);
        }
        int x_c[4], x_ix[4];
        unsigned x_l[4];
        bool x_in[4], x_ok[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int e = tid + 256 * j;
            const int c = e / p.xcols, col = e - c * p.xcols;
            int ix = col - p.pl;
            if (p.bmode == PAD_REFLECT) ix = reflect_idx(ix, p.W);
            x_in[j] = e < p.C * p.xcols;
            x_ok[j] = x_in[j] && col < p.OW + p.S - 1 && (unsigned)ix < (unsigned)p.W;
            x_c[j] = x_in[j] ? c : 0; x_ix[j] = x_ok[j] ? ix : 0;
            x_l[j] = (unsigned)(col * 2);
        }
        const bool big = npieces > 256 * 8 || p.C * p.xcols > 256 * 4;     // (planes wider than 256: the generic path below)
        // steady state: the dY tile of row (n, y) into buffer `buf_` and the ONE new logical x row t = y - pt + R - 1 per channel
#define C3_STAGE_FAST(row_, buf_)                                                                                       \
    do {                                                                                                                \
        const int n_ = (row_) / p.OH, y_ = (row_) - n_ * p.OH;                                                          \
        unsigned char* ab_ = smem + (size_t)(buf_) * abytes;                                                            \
        const bf16_t* dyb_ = p.dy + ((size_t)n_ * p.K) * aplane + (size_t)y_ * p.OW;                                    \
        const int t_ = y_ - p.pt + p.R - 1;                                                                             \
        int iy_ = t_;                                                                                                   \
        if (p.bmode == PAD_REFLECT) iy_ = reflect_idx(iy_, p.H);                                                        \
        const bool rowok_ = (unsigned)iy_ < (unsigned)p.H;                                                              \
        const size_t xb_ = (size_t)n_ * p.C * xplane + (size_t)(rowok_ ? iy_ : 0) * p.W;                                \
        c3_u32x4_t v_[8];                                                                                               \
        float xv_[4];                                                                                                   \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) v_[j] = *(const c3_u32x4_t*)(dyb_ + a_g[j]);                      \
        _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                                 \
            const size_t off_ = xb_ + (size_t)x_c[j] * xplane + x_ix[j];                                                \
            if constexpr (XF32) xv_[j] = ((const float*)p.x)[off_];                                                     \
            else xv_[j] = bf2f(((const bf16_t*)p.x)[off_]);                                                             \
        }                                                                                                               \
        const c3_u32x4_t z_ = {0u, 0u, 0u, 0u};                                                                         \
        _Pragma("unroll") for (int j = 0; j < 8; ++j) if (a_in[j]) *(c3_u32x4_t*)(ab_ + a_l[j]) = a_ok[j] ? v_[j] : z_; \
        const unsigned sl_ = (unsigned)((((n_ & 1) << 3) + (t_ & 7)) * p.xpitch);                                       \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                                   \
            if (x_in[j]) *(bf16_t*)(xring + (size_t)x_c[j] * 24 * p.xpitch + sl_ + x_l[j]) = f2bf((x_ok[j] && rowok_) ? xv_[j] : 0.f); \
    } while (0)
        // generic: all R logical rows when `full_` (first row of the workgroup or of an image), else the one new row
#define C3_STAGE(row_, buf_, full_)                                                                                     \
    do {                                                                                                                \
        const int n_ = (row_) / p.OH, y_ = (row_) - n_ * p.OH;                                                          \
        unsigned char* ab_ = smem + (size_t)(buf_) * abytes;                                                            \
        const bf16_t* dyb_ = p.dy + ((size_t)n_ * p.K) * aplane + (size_t)y_ * p.OW;                                    \
        const int r0_ = (full_) ? 0 : p.R - 1;                                                                          \
        const int nxr_ = p.R - r0_;                          /* logical rows to stage per channel */                    \
        const int nel_ = p.C * nxr_ * p.xcols;                                                                          \
        for (int it_ = 0; it_ * 256 * 8 < npieces || it_ * 256 * 4 < nel_; ++it_) {                                     \
            c3_u32x4_t v_[8];                                                                                           \
            float xv_[4];                                                                                               \
            bool okx_[4];                                                                                               \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                             \
                const int pc_ = tid + 256 * (8 * it_ + j);                                                              \
                const int k_ = pc_ / npc, c8_ = pc_ - k_ * npc;                                                         \
                const bool ok_ = pc_ < npieces && k_ < p.K;                                                             \
                v_[j] = *(const c3_u32x4_t*)(dyb_ + (size_t)(ok_ ? k_ : 0) * aplane + (ok_ ? c8_ * 8 : 0));             \
            }                                                                                                           \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                             \
                const int e_ = tid + 256 * (4 * it_ + j);                                                               \
                const int l_ = e_ / p.xcols, col_ = e_ - l_ * p.xcols;                                                  \
                const int c_ = l_ / nxr_, r_ = r0_ + (l_ - c_ * nxr_);                                                  \
                int iy_ = y_ - p.pt + r_, ix_ = col_ - p.pl;                                                            \
                if (p.bmode == PAD_REFLECT) { iy_ = reflect_idx(iy_, p.H); ix_ = reflect_idx(ix_, p.W); }               \
                okx_[j] = e_ < nel_ && col_ < p.OW + p.S - 1 && (unsigned)iy_ < (unsigned)p.H && (unsigned)ix_ < (unsigned)p.W; \
                const size_t off_ = okx_[j] ? ((size_t)(n_ * p.C + c_) * xplane + (size_t)iy_ * p.W + ix_) : 0;         \
                if constexpr (XF32) xv_[j] = ((const float*)p.x)[off_];                                                 \
                else xv_[j] = bf2f(((const bf16_t*)p.x)[off_]);                                                         \
            }                                                                                                           \
            _Pragma("unroll") for (int j = 0; j < 8; ++j) {                                                             \
                const int pc_ = tid + 256 * (8 * it_ + j);                                                              \
                const int k_ = pc_ / npc, c8_ = pc_ - k_ * npc;                                                         \
                const bool ok_ = pc_ < npieces && k_ < p.K;                                                             \
                const c3_u32x4_t z_ = {0u, 0u, 0u, 0u};                                                                 \
                if (pc_ < npieces) *(c3_u32x4_t*)(ab_ + (size_t)k_ * p.apitch + c8_ * 16) = ok_ ? v_[j] : z_;           \
            }                                                                                                           \
            _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                             \
                const int e_ = tid + 256 * (4 * it_ + j);                                                               \
                const int l_ = e_ / p.xcols, col_ = e_ - l_ * p.xcols;                                                  \
                const int c_ = l_ / nxr_, r_ = r0_ + (l_ - c_ * nxr_);                                                  \
                const int slot_ = c_ * 24 + ((n_ & 1) << 3) + ((y_ - p.pt + r_) & 7);                                   \
                if (e_ < nel_) *(bf16_t*)(xring + (size_t)slot_ * p.xpitch + col_ * 2) = f2bf(okx_[j] ? xv_[j] : 0.f);  \
            }                                                                                                           \
        }                                                                                                               \
    } while (0)
        C3_STAGE(row_lo, 0, true);
        __syncthreads();
        for (int row = row_lo; row < row_hi; ++row) {
            if (row + 1 < row_hi && !(p.dbg & 1)) {
                const bool full = (row + 1) % p.OH == 0;
                if (full) C3_STAGE(row + 1, (row + 1 - row_lo) & 1, true);
                else if (big) C3_STAGE(row + 1, (row + 1 - row_lo) & 1, false);
                else C3_STAGE_FAST(row + 1, (row + 1 - row_lo) & 1);
            }
            __syncthreads();
        }
#undef C3_STAGE
#undef C3_STAGE_FAST
        return;
    }

    // ===================== compute role: wave = (row block mi, half sh of the s range: s = 4 sh .. 4 sh + 3) =====================
    const int mi = wave >> 1, sh = wave & 1;
    f32x16_t acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const unsigned aoff = (unsigned)((mi * 32 + l31) * p.apitch + lhi * 16);
    const int bc = l31 / p.R, br = l31 - bc * p.R;          // this lane's MFMA column = (channel, kernel row)
    const int nks = p.OW >> 4;
    // the MFMAs of one reduction step for the s range [4 SH, 4 SH + 4): element shift s = dword shift s >> 1 (+ 16 bits when odd)
#define C3_STEP(SH)                                                                                                     \
    do {                                                                                                                \
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, c3_frag<4 * (SH) + 0>(w8), acc[0], 0, 0, 0);                \
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, c3_frag<4 * (SH) + 1>(w8), acc[1], 0, 0, 0);                \
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, c3_frag<4 * (SH) + 2>(w8), acc[2], 0, 0, 0);                \
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, c3_frag<4 * (SH) + 3>(w8), acc[3], 0, 0, 0);                \
    } while (0)
    // (the s-half branch OUTSIDE the loops: inside, the accumulators were copied between register sets at every step)
#define C3_ROWS(SH)                                                                                                     \
    do {                                                                                                                \
    for (int row = row_lo; row < row_hi; ++row) {                                                                       \
        const int n = row / p.OH, y = row - n * p.OH;                                                                   \
        const unsigned char* ab = smem + (size_t)((row - row_lo) & 1) * abytes + aoff;                                  \
        const int slot = l31 < CR ? bc * 24 + ((n & 1) << 3) + ((y - p.pt + br) & 7) : zrow;                            \
        const unsigned char* xb = xring + (size_t)slot * p.xpitch + lhi * 16;                                           \
        bf16x8_t an = *(const bf16x8_t*)(ab);                                                                           \
        c3_u32x4_t g0n = *(const c3_u32x4_t*)(xb), g1n = *(const c3_u32x4_t*)(xb + 16);                                 \
        for (int j = 0; j < ((p.dbg & 2) ? 0 : nks); ++j) {                                                             \
            const bf16x8_t a = an;                                                                                      \
            const unsigned w8[8] = {g0n[0], g0n[1], g0n[2], g0n[3], g1n[0], g1n[1], g1n[2], g1n[3]};                    \
            const int jn = j + 1 < nks ? j + 1 : j;                                                                     \
            an = *(const bf16x8_t*)(ab + jn * 32);                                                                      \
            g0n = *(const c3_u32x4_t*)(xb + jn * 32); g1n = *(const c3_u32x4_t*)(xb + jn * 32 + 16);                    \
            __builtin_amdgcn_sched_barrier(0);                                                                          \
            C3_STEP(SH);                                                                                                \
        }                                                                                                               \
        __syncthreads();                                                                                                \
    }                                                                                                                   \
    } while (0)
    __syncthreads();
    if (sh) C3_ROWS(1); else C3_ROWS(0);
#undef C3_ROWS
#undef C3_STEP
    // partial sums of this workgroup: ws[wg][s][m][n]
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int s = 4 * sh + i;
        if (s < p.S) {
            float* o = p.ws + (((size_t)blockIdx.x * p.S + s) * 64) * 32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                o[m * 32 + l31] = acc[i][r];
            }
        }
    }
}

// dW[k][c][r][s] (=|+=) sum_wg ws[wg][s][k][c R + r].  Block = 64 elements x 4 slices of the workgroup range.
__global__ __launch_bounds__(256) void wgrad_c3_finalize_kernel(const float* __restrict__ ws, int nwg, float* __restrict__ dw, int K,
                                                                int C, int R, int S, int accumulate) {
    __shared__ float part[4][64];
    const int total = S * 64 * 32;
    const int e = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + e;
    const int per = (nwg + 3) / 4;
    const int w0 = q * per, w1 = w0 + per < nwg ? w0 + per : nwg;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    if (i < total) {
        int wg = w0;
        for (; wg + 4 <= w1; wg += 4) {
            a0 += ws[(size_t)(wg + 0) * total + i]; a1 += ws[(size_t)(wg + 1) * total + i];
            a2 += ws[(size_t)(wg + 2) * total + i]; a3 += ws[(size_t)(wg + 3) * total + i];
        }
        for (; wg < w1; ++wg) a0 += ws[(size_t)wg * total + i];
    }
    part[q][e] = (a0 + a1) + (a2 + a3);
    __syncthreads();
    if (q != 0 || i >= total) return;
    const float v = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
    const int n = i & 31, k = (i >> 5) & 63, s = i >> 11;
    if (k >= K || n >= C * R) return;
    const int c = n / R, r = n - c * R;
    float* d = dw + (((size_t)k * C + c) * R + r) * S + s;
    *d = accumulate ? *d + v : v;
}

// HIFIC_ERR_UNSUPPORTED: not this layer (nothing launched)
int gc_launch_wgrad_c3(const ConvGeom& g, const void* x, const void* dy, float* dw, int accumulate, int x_f32, int dy_f32,
                       WsAlloc& ws, hipStream_t st) {
    if (!gc_env_int("HIFIC_WGRAD_C3", 1)) return HIFIC_ERR_UNSUPPORTED;
    const int OH = g.OH(), OW = g.OW();
    if (g.stride != 1 || dy_f32 || g.C > 4 || g.C * g.R > 32 || g.R > 7 || g.S > 8 || g.K > 64 || g.K <= 4 || OW % 16 != 0 || OW > 1024 ||
        g.pl >= 8 || g.N * OH < 1024)
        return HIFIC_ERR_UNSUPPORTED;
    if (((size_t)dy & 15) != 0) return HIFIC_ERR_UNSUPPORTED;
    if (g.pad_mode == PAD_REFLECT && (g.pt >= g.H || g.pb >= g.H || g.pl >= g.W || g.pr >= g.W)) return HIFIC_ERR_UNSUPPORTED;
    WgC3Params p; memset(&p, 0, sizeof(p));
    p.dy = (const bf16_t*)dy; p.x = x;
    p.N = g.N; p.K = g.K; p.C = g.C; p.H = g.H; p.W = g.W; p.OH = OH; p.OW = OW; p.R = g.R; p.S = g.S; p.pt = g.pt; p.pl = g.pl;
    p.bmode = g.pad_mode;
    p.dbg = gc_env_int("HIFIC_DBG", 0);
    p.nrows = g.N * OH;
    int nwg = p.nrows < 256 ? p.nrows : 256;
    p.rows_per_wg = cdiv(p.nrows, nwg);
    nwg = cdiv(p.nrows, p.rows_per_wg);
    p.xcols = ((OW + 8 + 7) & ~7) + 8;                       // the shifted windows of the last reduction step reach column OW + 14
    p.apitch = (OW + 8) * 2;
    p.xpitch = ((p.xcols * 2 + 15) / 16) * 16;
    if (((p.xpitch / 16) & 1) == 0) p.xpitch += 16;          // odd multiple of 16 bytes: consecutive ring slots in different banks
    const size_t lds = 2 * (size_t)64 * p.apitch + (size_t)(g.C * 24 + 1) * p.xpitch;
    if (lds > (size_t)160 * 1024) return HIFIC_ERR_UNSUPPORTED;
    p.ws = (float*)ws.take((size_t)nwg * g.S * 64 * 32 * sizeof(float));
    if (!p.ws) return HIFIC_ERR_UNSUPPORTED;
    char ptag[112];
    snprintf(ptag, sizeof(ptag), "wgrad_c3 K%d C%d N%d out%dx%d taps%dx%d grid%d", g.K, g.C, g.N, OH, OW, g.R, g.S, nwg);
    const int pslot = gc_prof_open("wgrad_c3_kernel", 2.0 * g.K * g.C * g.R * g.S * (double)g.N * OH * OW, st, ptag);
    gc_prof_bytes(pslot, (double)g.N * g.K * OH * OW * 2.0 + (double)g.N * g.C * g.H * g.W * (x_f32 ? 4.0 : 2.0) +
                         (double)g.K * g.C * g.R * g.S * 4.0);
    if (x_f32) {
        gc_set_max_lds((const void*)wgrad_c3_kernel<true>, (int)lds);
        hipLaunchKernelGGL(wgrad_c3_kernel<true>, dim3(nwg), dim3(512), lds, st, p);
    } else {
        gc_set_max_lds((const void*)wgrad_c3_kernel<false>, (int)lds);
        hipLaunchKernelGGL(wgrad_c3_kernel<false>, dim3(nwg), dim3(512), lds, st, p);
    }
    const int total = g.S * 64 * 32;
    hipLaunchKernelGGL(wgrad_c3_finalize_kernel, dim3(cdiv(total, 64)), dim3(256), 0, st, p.ws, nwg, dw, g.K, g.C, g.R, g.S, accumulate);
    gc_prof_close(pslot, st);
    return hific_launch_status();
}
