// Weight-gradient kernels of the conv engine and their planner: dW[m][tap][c] = sum_{n,u,v} A[n,m,u,v] B[n,c,u*ist+dy,v*ist+dx]
// (A = dY, B = x; roles swapped for conv-transpose).  wgrad_s1_kernel / wgrad_s2_kernel: both operands in natural NCHW order
// (stride-1 3x3 on 16-pixel planes; stride-2 3x3 / 4x4, phase-decomposed); wgrad_pipe_kernel, wgrad_kernel: transposed halo
// patch + ds_read_b64_tr_b16 (odd planes, 5x5, 1x1, float32); wgrad_im2col_kernel: few channels on one side.  Reference: the
// autograd of nn.Conv2d / nn.ConvTranspose2d at every call site listed in gconv.h.
#include "gconv.h"
#include "gconv_dev.h"
#include "gconv_stage.h"
#include <type_traits>
#include <string.h>
#include <stdio.h>

// ---------------------------------------------------------------------------------------------------
// Weight-gradient kernel
// ---------------------------------------------------------------------------------------------------
template <typename T> struct WgCfg;
template <> struct WgCfg<bf16_t> { static constexpr int DWR = 32, PITCH = 144, KS = 16; };
template <> struct WgCfg<float>  { static constexpr int DWR = 64, PITCH = 260, KS = 2; };

// QBW as in gconv_kernel: -1 = wide-load staging variant (operands with wstage_a / wstage_b set use stage_W)
template <typename T, int QBW>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(QBW > 1 ? 1 : 2, QBW > 1 ? 1 : 8)))
void wgrad_kernel(const WgParams p) {
    constexpr bool WIDE = QBW < 0;
    constexpr int QB = QBW < 0 ? 1 : QBW;
    static_assert(!WIDE || std::is_same<T, bf16_t>::value, "wide staging: bf16");
    using Cfg = WgCfg<T>;
    constexpr int PITCH = Cfg::PITCH, KS = Cfg::KS, DWR = Cfg::DWR;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;

    const GcPhase& gp = p.grp[blockIdx.y];
    const int ctiles = p.Cpad / 64;
    const int m0 = (blockIdx.x / ctiles) * 64;
    const int c0 = (blockIdx.x % ctiles) * 64;
    const int split = blockIdx.z;
    const int PH = gp.PH, PW = gp.PW, npp = PH * PW;
    const int npix = p.NI * p.TH * p.TW;          // multiple of 16
    const int npatch = p.NI * npp;

    int* qtab = (int*)smem;                                   // [128]
    unsigned char* at = smem + 512;                           // [npix][PITCH]
    unsigned char* patch = at + (size_t)GC_NPIX * PITCH;      // [npatch][PITCH]

    const int thw = p.TH * p.TW;
    if (tid < GC_NPIX) {
        const int img = tid / thw;
        const int rem = tid - img * thw;
        const int ty_ = rem / p.TW, tx_ = rem - ty_ * p.TW;
        qtab[tid] = (tid < npix) ? (img * npp + ty_ * p.ist * PW + tx_ * p.ist) : 0;
    }
    int toffs[GC_TG];
#pragma unroll
    for (int t = 0; t < GC_TG; ++t) {
        const int tt = t < gp.ntaps ? t : 0;
        toffs[t] = ((int)p.tap_dy[gp.tap0 + tt] - gp.dy_min) * PW + ((int)p.tap_dx[gp.tap0 + tt] - gp.dx_min);
    }

    f32x16_t acc[GC_TG];
#pragma unroll
    for (int t = 0; t < GC_TG; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int tile_lo = split * p.tiles_per_split;
    int tile_hi = tile_lo + p.tiles_per_split;
    if (tile_hi > p.ntiles) tile_hi = p.ntiles;

    for (int tile = tile_lo; tile < tile_hi; ++tile) {
        const int tx = tile % p.tiles_x;
        const int ty = (tile / p.tiles_x) % p.tiles_y;
        const int tn = tile / (p.tiles_x * p.tiles_y);
        const int u0 = ty * p.TH, v0 = tx * p.TW, n0 = tn * p.NI;
        __syncthreads();
        bool wide_a = false, wide_b = false;
        if constexpr (WIDE) { wide_a = p.wstage_a != 0; wide_b = p.wstage_b != 0; }
        if (!(p.dbg & 1)) {
            if (wide_a) {
                if constexpr (WIDE)
                    stage_W<PITCH, GC_WSTAGE_WB_WG>(at, (const bf16_t*)p.a, p.N, p.M, p.AH, p.AW, PAD_ZERO, n0, p.NI, u0, v0,
                                                 p.TH, p.TW, p.TW, m0, tid, GC_NPIX + npatch);
            } else
            stage_T<T, DWR, PITCH, QB>(at, p.a, p.a_f32, p.N, p.M, p.AH, p.AW, PAD_ZERO,
                                       n0, p.NI, u0, v0, 0, p.TH, p.TW, m0, tid, 256);
        }
        if (!(p.dbg & 2)) {
            if (wide_b) {
                if constexpr (WIDE)
                    stage_W<PITCH, GC_WSTAGE_WB_WG>(patch, (const bf16_t*)p.b, p.N, p.C, p.BH, p.BW, p.bmode, n0, p.NI,
                                                 u0 * p.ist + gp.dy_min, v0 * p.ist + gp.dx_min, PH, PW, PW, c0, tid, npatch);
            } else
            stage_T<T, DWR, PITCH, QB>(patch, p.b, p.b_f32, p.N, p.C, p.BH, p.BW, p.bmode,
                                       n0, p.NI, u0 * p.ist + gp.dy_min, v0 * p.ist + gp.dx_min, 0, PH, PW, c0, tid, 256);
        }
        __syncthreads();
        for (int ks = 0; ks < ((p.dbg & 4) ? 0 : npix / KS); ++ks) {
            if constexpr (std::is_same<T, float>::value) {
                const int r = ks * 2 + lhi;
                const float a = *(const float*)(at + (size_t)r * PITCH + (wm * 32 + l31) * 4);
                const unsigned char* brow = patch + (size_t)qtab[r] * PITCH + (wn * 32 + l31) * 4;
                // all GC_TG taps unconditionally (taps beyond ntaps alias tap 0 and are dropped in the epilogue)
                float bb[GC_TG];
#pragma unroll
                for (int t = 0; t < GC_TG; ++t) bb[t] = *(const float*)(brow + (size_t)toffs[t] * PITCH);
#pragma unroll
                for (int t = 0; t < GC_TG; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bb[t], acc[t], 0, 0, 0);
            } else {
                // ds_read_b64_tr_b16: each 16-lane group reads a [4 rows][16 cols] bf16 block; lane i supplies the
                // address of row (i>>2), col chunk (i&3)*4 and receives column i of the 4 rows.
                const int g = lane >> 4, i16 = lane & 15;
                const int rb = ks * 16 + (g >> 1) * 8 + (i16 >> 2);
                const int colb = ((g & 1) * 16 + (i16 & 3) * 4) * 2;
                typedef __attribute__((address_space(3))) short4_t* lds_s4;
                const unsigned char* a0p = at + (size_t)rb * PITCH + wm * 64 + colb;
                const unsigned char* a1p = at + (size_t)(rb + 4) * PITCH + wm * 64 + colb;
                short4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)a0p);
                short4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)a1p);
                bf16x8_t a;
                {
                    typedef __attribute__((ext_vector_type(8))) short short8_t;
                    short8_t av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                    a = __builtin_bit_cast(bf16x8_t, av);
                }
                const int q0 = qtab[rb], q1 = qtab[rb + 4];
                const unsigned char* b0row = patch + (size_t)q0 * PITCH + wn * 64 + colb;
                const unsigned char* b1row = patch + (size_t)q1 * PITCH + wn * 64 + colb;
                typedef __attribute__((ext_vector_type(8))) short short8_t;
                short4_t b0[GC_TG], b1[GC_TG];
#pragma unroll
                for (int t = 0; t < GC_TG; ++t) {
                    b0[t] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(b0row + (size_t)toffs[t] * PITCH));
                    b1[t] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(b1row + (size_t)toffs[t] * PITCH));
                }
#pragma unroll
                for (int t = 0; t < GC_TG; ++t) {
                    short8_t bv = {b0[t][0], b0[t][1], b0[t][2], b0[t][3], b1[t][0], b1[t][1], b1[t][2], b1[t][3]};
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8_t, bv), acc[t], 0, 0, 0);
                }
            }
        }
    }

    // single split: scatter straight into the PyTorch weight-gradient layout; else partials ws[split][m][tap][c]
#pragma unroll
    for (int t = 0; t < GC_TG; ++t) {
        if (t < gp.ntaps) {
            const int tg = gp.tap0 + t;
            const long long toff_w = p.direct ? (long long)p.tap_r[tg] * p.sr + (long long)p.tap_s[tg] * p.ss : 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                const int c = c0 + wn * 32 + l31;
                if (p.direct) {
                    if (m < p.M && c < p.C) {
                        float* d = p.dw + m * p.sm + c * p.sc + toff_w;
                        if (p.accumulate) *d += acc[t][r]; else *d = acc[t][r];
                    }
                } else {
                    p.ws[(((size_t)split * p.Mpad + m) * p.ntaps + tg) * p.Cpad + c] = acc[t][r];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Pipelined bf16 weight-gradient kernel (3x3 stride-1 class layers: TW % 8 == 0, AW % 8 == 0, patch <= 192 pixels).
//   * dY operand in its natural NCHW order: [64 m][128 tile pixels] LDS image filled by 16-byte global loads
//     (4 per thread per tile instead of 32 two-byte loads); its MFMA fragment is a plain ds_read_b128
//   * x operand: transposed halo patch + ds_read_b64_tr_b16 as in wgrad_kernel
//   * both LDS images are double-buffered; the next tile's data is prefetched into registers while the current
//     tile's 72 MFMAs per wave run (patch in two halves to keep the prefetch at 24+16 VGPRs); one barrier per tile
// ---------------------------------------------------------------------------------------------------
// TS = 2: 8 waves; waves 4-7 mirror waves 0-3 on the same (m, c) tile and the same staged operands but own taps 5..8
// (waves 0-3: taps 0..4).  80 instead of 144 accumulator registers per wave, so two waves fit per SIMD and hide each
// other's LDS / barrier latency; no exchange at the end (different taps are different outputs); the staging work of a
// tile is spread over 512 threads (half the prefetch registers per thread).  Needs a 9-tap group.
// SH3 (3x3 window, taps ordered (dy, dx) with dx ascending): the B fragment of tap (dy, dx+1) is the fragment of tap
// (dy, dx) shifted by one pixel along the reduction index, so the three fragments of a kernel row are built from 10
// consecutive patch pixels (3 transpose reads + 4 v_alignbit) instead of 3 x 2 transpose reads: 9 instead of 18 LDS
// reads per 9 MFMAs (the kernel is LDS-read bound: 10 KB of fragment reads per wave per 16-deep slice).  With TS = 2
// the waves split by kernel row (rows 0-1 | row 2) instead of 5 | 4 taps.
template <int TS, bool SH3>
__global__ __launch_bounds__(256 * TS) __attribute__((amdgpu_waves_per_eu(TS == 2 ? 2 : 1, TS == 2 ? 2 : 8)))
void wgrad_pipe_kernel(const WgParams p) {
    constexpr int PITCH = 144, NDW = 8 / TS, QI = 3, HALF = NDW / 2;
    constexpr int NTH = 256 * TS;                                  // threads
    constexpr int NAP = 4 / TS;                                    // 16-byte A pieces per thread per tile
    constexpr int NACC = TS == 2 ? (SH3 ? 6 : 5) : GC_TG;          // accumulator sets per wave
    constexpr int TSPLIT = SH3 ? 6 : 5;                            // first tap of the second wave set
    constexpr int APITCH = GC_NPIX * 2 + 16;                       // 272 B per m row
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wset = wave >> 2;                                    // tap set of this wave (TS == 2)
    const int wm = (wave & 3) >> 1, wn = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int pwv = tid >> 6;

    const GcPhase& gp = p.grp[blockIdx.y];
    const int ctiles = p.Cpad / 64;
    const int m0 = (blockIdx.x / ctiles) * 64;
    const int c0 = (blockIdx.x % ctiles) * 64;
    const int split = blockIdx.z;
    const int PH = gp.PH, PW = gp.PW, npp = PH * PW;
    const int npatch = p.NI * npp;
    const size_t patch_bytes = ((size_t)(npatch + 3) * PITCH + 15) & ~(size_t)15;   // + dump row for lanes past the patch, + 2 rows read (unused) by SH3
    constexpr int ABYTES = 64 * APITCH;

    int* qtab = (int*)smem;                                         // [128]
    unsigned char* abuf = smem + 512;                               // 2 x ABYTES
    unsigned char* pbuf = abuf + 2 * ABYTES;                        // 2 x patch_bytes

    const int thw = p.TH * p.TW;
    if (tid < GC_NPIX) {
        const int img = tid / thw;
        const int rem = tid - img * thw;
        const int ty_ = rem / p.TW, tx_ = rem - ty_ * p.TW;
        qtab[tid] = img * npp + ty_ * p.ist * PW + tx_ * p.ist;
    }
    int toffs[GC_TG];
#pragma unroll
    for (int t = 0; t < GC_TG; ++t) {
        const int tt = t < gp.ntaps ? t : 0;
        toffs[t] = ((int)p.tap_dy[gp.tap0 + tt] - gp.dy_min) * PW + ((int)p.tap_dx[gp.tap0 + tt] - gp.dx_min);
    }
    int toffb[GC_TG];
#pragma unroll
    for (int t = 0; t < GC_TG; ++t) toffb[t] = toffs[t] * PITCH;
    // A pieces of this thread: piece = tid + 256*i -> (row m, 8-pixel segment); tile-independent part of the address
    int a_img[NAP], a_ty[NAP], a_tx[NAP], a_row[NAP], a_seg[NAP];
    unsigned a_rel[NAP];
    const unsigned aplane = (unsigned)(p.AH * p.AW);
#pragma unroll
    for (int i = 0; i < NAP; ++i) {
        const int piece = tid + NTH * i;
        a_row[i] = piece >> 4; a_seg[i] = piece & 15;
        const int r0 = a_seg[i] * 8;
        a_img[i] = r0 / thw;
        const int rem = r0 - a_img[i] * thw;
        a_ty[i] = rem / p.TW; a_tx[i] = rem - a_ty[i] * p.TW;
        a_rel[i] = (unsigned)(a_img[i] * p.M + m0 + a_row[i]) * aplane + (unsigned)(a_ty[i] * p.AW + a_tx[i]);
    }
    const bf16_t* asrc = (const bf16_t*)p.a;
    const float inv_npp = 1.0f / (float)npp, inv_pw = 1.0f / (float)PW;
    const unsigned bplane = (unsigned)(p.BH * p.BW);
    const bool cfull = c0 + 64 <= p.C;

    f32x16_t acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int tile_lo = split * p.tiles_per_split;
    int tile_hi = tile_lo + p.tiles_per_split;
    if (tile_hi > p.ntiles) tile_hi = p.ntiles;

    u32x4_t areg[NAP]; unsigned aokm = 0;
    unsigned short plo[QI][NDW], phi[QI][NDW];                      // raw 16-bit loads, untouched until the store
    unsigned qoff[QI]; unsigned qokm = 0;

#define WG_TILE_ORIGIN(tile_, n0_, u0_, v0_)                        \
    const int tx_t = (tile_) % p.tiles_x;                           \
    const int ty_t = ((tile_) / p.tiles_x) % p.tiles_y;             \
    const int tn_t = (tile_) / (p.tiles_x * p.tiles_y);             \
    const int u0_ = ty_t * p.TH, v0_ = tx_t * p.TW, n0_ = tn_t * p.NI;
#define WG_LOAD_A(n0_, u0_, v0_)                                                                            \
    do {                                                                                                    \
        const unsigned tbase = (unsigned)(n0_ * p.M) * aplane + (unsigned)(u0_ * p.AW + v0_);               \
        _Pragma("unroll") for (int i = 0; i < NAP; ++i) {                                                   \
            const bool ok_ = (n0_ + a_img[i] < p.N) && (u0_ + a_ty[i] < p.AH) && (v0_ + a_tx[i] < p.AW) &&   \
                     (m0 + a_row[i] < p.M) && (a_img[i] < p.NI);                                            \
            aokm = (aokm & ~(1u << i)) | ((ok_ ? 1u : 0u) << i);                                            \
            areg[i] = *(const u32x4_t*)(asrc + (ok_ ? tbase + a_rel[i] : 0u));                              \
        }                                                                                                   \
    } while (0)
#define WG_STORE_A(buf_)                                                                                    \
    do {                                                                                                    \
        _Pragma("unroll") for (int i = 0; i < NAP; ++i) {                                                   \
            u32x4_t v = areg[i];                                                                            \
            if (!((aokm >> i) & 1u)) { v.x = 0; v.y = 0; v.z = 0; v.w = 0; }                                            \
            *(u32x4_t*)((buf_) + a_row[i] * APITCH + a_seg[i] * 16) = v;                                    \
        }                                                                                                   \
    } while (0)
#define WG_DECODE_P(n0_, u0_, v0_)                                                                          \
    do {                                                                                                    \
        qokm = 0;                                                                                           \
        _Pragma("unroll") for (int j = 0; j < QI; ++j) {                                                    \
            bool ok_;                                                                                       \
            int qs_;                                                                                        \
            px_decode(lane + 64 * j, npatch, npp, PW, inv_npp, inv_pw, n0_, u0_ * p.ist + gp.dy_min,        \
                      v0_ * p.ist + gp.dx_min, p.N, p.C, p.BH, p.BW, p.bmode, qoff[j], ok_, PW, qs_);       \
            qokm |= (ok_ ? 1u : 0u) << j;                                                                   \
        }                                                                                                   \
    } while (0)
    // half h of the patch dwords: i in [h*HALF, h*HALF+HALF)
#define WG_LOAD_P(h)                                                                                        \
    do {                                                                                                    \
        const bf16_t* sp = (const bf16_t*)p.b;                                                              \
        _Pragma("unroll") for (int j = 0; j < QI; ++j) {                                                    \
            _Pragma("unroll") for (int ii = 0; ii < HALF; ++ii) {                                           \
                const int i = (h) * HALF + ii;                                                              \
                const int c = c0 + 2 * (pwv + 4 * TS * i);                                                       \
                const unsigned off = qoff[j] + (unsigned)c * bplane;                                        \
                const bool k0 = cfull || c < p.C, k1 = cfull || c + 1 < p.C;                                \
                plo[j][i] = sp[k0 ? off : 0u];                                                              \
                phi[j][i] = sp[k1 ? off + bplane : 0u];                                                     \
            }                                                                                               \
        }                                                                                                   \
    } while (0)
#define WG_STORE_P(buf_, h)                                                                                 \
    do {                                                                                                    \
        _Pragma("unroll") for (int j = 0; j < QI; ++j) {                                                    \
            const int q = lane + 64 * j;                                                                    \
            unsigned char* row = (buf_) + (size_t)(q < npatch ? q : npatch) * PITCH + pwv * 4;              \
            _Pragma("unroll") for (int ii = 0; ii < HALF; ++ii) {                                           \
                const int i = (h) * HALF + ii;                                                              \
                const int c = c0 + 2 * (pwv + 4 * TS * i);                                                       \
                const unsigned l = (((qokm >> j) & 1u) && (cfull || c < p.C)) ? (unsigned)plo[j][i] : 0u;   \
                const unsigned hh = (((qokm >> j) & 1u) && (cfull || c + 1 < p.C)) ? (unsigned)phi[j][i] : 0u; \
                *(unsigned*)(row + i * 16 * TS) = l | (hh << 16);                                                \
            }                                                                                               \
        }                                                                                                   \
    } while (0)
#define WG_COMPUTE(ab_, pb_, ks_lo, ks_hi, T0, NT)                                                                  \
    do {                                                                                                    \
        const int g = lane >> 4, i16 = lane & 15;                                                           \
        const int colb = ((g & 1) * 16 + (i16 & 3) * 4) * 2;                                                \
        typedef __attribute__((address_space(3))) short4_t* lds_s4;                                         \
        typedef __attribute__((ext_vector_type(8))) short short8_t;                                         \
        _Pragma("unroll") for (int ks = (ks_lo); ks < (ks_hi); ++ks) {                                       \
            const bf16x8_t a = *(const bf16x8_t*)((ab_) + (wm * 32 + l31) * APITCH + (ks * 16 + lhi * 8) * 2); \
            const int rb = ks * 16 + (g >> 1) * 8 + (i16 >> 2);                                             \
            const unsigned char* b0row = (pb_) + (size_t)qtab[rb] * PITCH + wn * 64 + colb;                 \
            const unsigned char* b1row = (pb_) + (size_t)qtab[rb + 4] * PITCH + wn * 64 + colb;             \
            /* all GC_TG taps unconditionally (taps beyond ntaps alias tap 0, their accumulators are dropped): */ \
            /* straight-line code lets the compiler issue the 18 LDS reads ahead of the 9 independent MFMAs */   \
            short4_t b0[NT], b1[NT];                                                                        \
            _Pragma("unroll") for (int t = 0; t < (NT); ++t) {                                              \
                b0[t] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(b0row + toffb[(T0) + t]));         \
                b1[t] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(b1row + toffb[(T0) + t]));         \
            }                                                                                               \
            _Pragma("unroll") for (int t = 0; t < (NT); ++t) {                                              \
                short8_t bv = {b0[t][0], b0[t][1], b0[t][2], b0[t][3], b1[t][0], b1[t][1], b1[t][2], b1[t][3]}; \
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8_t, bv), acc[t], 0, 0, 0); \
            }                                                                                               \
        }                                                                                                   \
    } while (0)

    // SH3 form: kernel rows [D0, D0+ND) of the 3x3 window; accumulator of tap (d, j) = acc[(d - D0) * 3 + j]
#define WG_COMPUTE_SH3(ab_, pb_, ks_lo, ks_hi, D0, ND)                                                      \
    do {                                                                                                    \
        const int g = lane >> 4, i16 = lane & 15;                                                           \
        const int colb = ((g & 1) * 16 + (i16 & 3) * 4) * 2;                                                \
        typedef __attribute__((address_space(3))) short4_t* lds_s4;                                         \
        typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));                                   \
        typedef unsigned int u32x4b_t __attribute__((ext_vector_type(4)));                                  \
        _Pragma("unroll") for (int ks = (ks_lo); ks < (ks_hi); ++ks) {                                       \
            const bf16x8_t a = *(const bf16x8_t*)((ab_) + (wm * 32 + l31) * APITCH + (ks * 16 + lhi * 8) * 2); \
            const int rb = ks * 16 + (g >> 1) * 8 + (i16 >> 2);                                             \
            const unsigned char* brow = (pb_) + (size_t)qtab[rb] * PITCH + wn * 64 + colb;                  \
            u32x2_t P[ND][3];                                                                               \
            _Pragma("unroll") for (int d = 0; d < (ND); ++d) {                                              \
                const unsigned char* r0 = brow + toffb[((D0) + d) * 3];                                     \
                P[d][0] = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(r0)));             \
                P[d][1] = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(r0 + 4 * PITCH))); \
                P[d][2] = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(r0 + 8 * PITCH))); \
            }                                                                                               \
            _Pragma("unroll") for (int d = 0; d < (ND); ++d) {                                              \
                const unsigned R0 = P[d][0].x, R1 = P[d][0].y, R2 = P[d][1].x, R3 = P[d][1].y, R4 = P[d][2].x; \
                const u32x4b_t f0 = {R0, R1, R2, R3};                                                       \
                const u32x4b_t f1 = {__builtin_amdgcn_alignbit(R1, R0, 16), __builtin_amdgcn_alignbit(R2, R1, 16), \
                                     __builtin_amdgcn_alignbit(R3, R2, 16), __builtin_amdgcn_alignbit(R4, R3, 16)}; \
                const u32x4b_t f2 = {R1, R2, R3, R4};                                                       \
                acc[d * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8_t, f0), acc[d * 3 + 0], 0, 0, 0); \
                acc[d * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8_t, f1), acc[d * 3 + 1], 0, 0, 0); \
                acc[d * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8_t, f2), acc[d * 3 + 2], 0, 0, 0); \
            }                                                                                               \
        }                                                                                                   \
    } while (0)

    // Register prefetch at a distance of one full tile, branch-free: iteration t stores tile t+1 (loaded during
    // iteration t-1) into the other LDS buffers, requests tile t+2 and then runs the 72 MFMAs of tile t, so every
    // load has a whole tile of compute to land and the loop body is one basic block with constant load counts.
    // Tiles past the end re-load the last tile (never consumed).
    if (tile_lo < tile_hi) {
        const int tile_last = tile_hi - 1;
        {
            WG_TILE_ORIGIN(tile_lo, n0, u0, v0)
            WG_LOAD_A(n0, u0, v0);
            WG_DECODE_P(n0, u0, v0);
            WG_LOAD_P(0); WG_LOAD_P(1);
            WG_STORE_A(abuf);
            WG_STORE_P(pbuf, 0); WG_STORE_P(pbuf, 1);
        }
        {
            const int t1 = tile_lo + 1 < tile_hi ? tile_lo + 1 : tile_last;
            WG_TILE_ORIGIN(t1, n0, u0, v0)
            WG_LOAD_A(n0, u0, v0);
            WG_DECODE_P(n0, u0, v0);
            WG_LOAD_P(0); WG_LOAD_P(1);
        }
        for (int tile = tile_lo; tile < tile_hi; ++tile) {
            const int cur = (tile - tile_lo) & 1;
            const unsigned char* ab = abuf + cur * ABYTES;
            const unsigned char* pb = pbuf + cur * patch_bytes;
            unsigned char* abn = abuf + (cur ^ 1) * ABYTES;
            unsigned char* pbn = pbuf + (cur ^ 1) * patch_bytes;
            __syncthreads();
            WG_STORE_A(abn);
            WG_STORE_P(pbn, 0); WG_STORE_P(pbn, 1);
            {
                const int t2 = tile + 2 < tile_hi ? tile + 2 : tile_last;
                WG_TILE_ORIGIN(t2, n0, u0, v0)
                WG_LOAD_A(n0, u0, v0);
                WG_DECODE_P(n0, u0, v0);
                WG_LOAD_P(0); WG_LOAD_P(1);
            }
            if constexpr (SH3) {
                if constexpr (TS == 2) {
                    if (wset == 0) WG_COMPUTE_SH3(ab, pb, 0, GC_NPIX / 16, 0, 2);
                    else WG_COMPUTE_SH3(ab, pb, 0, GC_NPIX / 16, 2, 1);
                } else {
                    WG_COMPUTE_SH3(ab, pb, 0, GC_NPIX / 16, 0, 3);
                }
            } else if constexpr (TS == 2) {
                if (wset == 0) WG_COMPUTE(ab, pb, 0, GC_NPIX / 16, 0, 5);
                else WG_COMPUTE(ab, pb, 0, GC_NPIX / 16, 5, 4);
            } else {
                WG_COMPUTE(ab, pb, 0, GC_NPIX / 16, 0, GC_TG);
            }
        }
    }
#undef WG_COMPUTE_SH3
#undef WG_COMPUTE
#undef WG_STORE_P
#undef WG_LOAD_P
#undef WG_DECODE_P
#undef WG_STORE_A
#undef WG_LOAD_A
#undef WG_TILE_ORIGIN

    // Epilogue.  Direct mode with the whole kernel window in this group (3x3 layers): the per-lane scatter
    // (4-byte stores at a 36-byte stride) costs 8x write amplification (rocprofv3 WRITE_SIZE 275 MB for a 33 MB
    // gradient), so the tile is transposed through LDS and each m row leaves as one contiguous run of 64c x 9 taps.
    if (p.direct && p.ngroups == 1 && p.sc == gp.ntaps && p.ss == 1 && gp.ntaps == GC_TG) {
        constexpr int RP = 32 * GC_TG + 1;                      // floats per staged row (odd: conflict-free)
        float* stg = (float*)smem + (size_t)(wave & 3) * 16 * RP;   // region of the (wm, wn) quadrant: 16 rows
        const int tbase = (TS == 2 && wset == 1) ? TSPLIT : 0;  // first tap of this wave's accumulators
        __syncthreads();                                        // all waves done with the operand buffers
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int t = 0; t < NACC; ++t)
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int r = h * 8 + rr;
                    const int rowl = (r & 3) + 8 * ((r >> 2) & 1) + 4 * lhi;      // 0..15 within the half
                    if (tbase + t < GC_TG) stg[rowl * RP + l31 * GC_TG + tbase + t] = acc[t][r];
                }
            if constexpr (TS == 2) __syncthreads();   // the quadrant's two waves filled disjoint taps of the region
            else __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's LDS writes landed (wave-private region)
            for (int rowl = (TS == 2 ? wset * 8 : 0); rowl < (TS == 2 ? wset * 8 + 8 : 16); ++rowl) {
                const int m = m0 + wm * 32 + h * 16 + rowl;
                if (m >= p.M) break;                                   // (no barrier inside this loop)
                float* drow = p.dw + (long long)m * p.sm + (long long)(c0 + wn * 32) * p.sc;
                int nvalid = (p.C - (c0 + wn * 32)) * GC_TG; if (nvalid > 32 * GC_TG) nvalid = 32 * GC_TG;
                for (int j = lane; j < nvalid; j += 64) {
                    const float v = stg[rowl * RP + j];
                    if (p.accumulate) drow[j] += v; else drow[j] = v;
                }
            }
            if constexpr (TS == 2) __syncthreads();
            else __builtin_amdgcn_s_waitcnt(0xc07f);
        }
        return;
    }
    const int tb2 = (TS == 2 && wset == 1) ? TSPLIT : 0;
#pragma unroll
    for (int ta = 0; ta < NACC; ++ta) {
        const int t = tb2 + ta;
        if (t < gp.ntaps) {
            const int tg = gp.tap0 + t;
            const long long toff_w = p.direct ? (long long)p.tap_r[tg] * p.sr + (long long)p.tap_s[tg] * p.ss : 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                const int c = c0 + wn * 32 + l31;
                if (p.direct) {
                    if (m < p.M && c < p.C) {
                        float* d = p.dw + m * p.sm + c * p.sc + toff_w;
                        if (p.accumulate) *d += acc[ta][r]; else *d = acc[ta][r];
                    }
                } else {
                    p.ws[(((size_t)split * p.Mpad + m) * p.ntaps + tg) * p.Cpad + c] = acc[ta][r];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Small-channel weight gradient (one operand has <= 4 channels: the first Encoder conv 3->60 and the last Generator
// conv 60->3, both 7x7).  Padding 3 channels to a 64-wide MFMA tile wastes 95% of the work, so the taps are folded
// into the GEMM column dimension instead: column j = tap*4 + c, B image [tile pixel][64 columns] is an im2col slice
// gathered straight from global memory (the small operand is L2 resident), one accumulator tile per wave.
//   normal  : D[m][(t,c)]  = sum_pix A[m][pix] * B[c][pix + tap_t]         (A = dY, B = x with reflect/zero pad)
//   swapped : D[c][(t,m)]  = sum_pix' A'[c][pix'] * B'[m][pix' - tap_t]    (A' = padded x over the padded domain,
//             B' = dY zero outside) -- used when dY is the small operand
// ---------------------------------------------------------------------------------------------------
#ifndef IM2COL_QB
#define IM2COL_QB 2
#endif
template <typename T, bool BF32>
__global__ __launch_bounds__(256) void wgrad_im2col_kernel(const WgParams p) {
    using Cfg = WgCfg<T>;
    constexpr int PITCH = Cfg::PITCH, KS = Cfg::KS, DWR = Cfg::DWR, NDW = DWR / 4;
    constexpr int MAXCT = 4;                                   // 64-column tiles per workgroup (<= 256 virtual columns)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int wv = tid >> 6;
    const int nct = p.Cpad / 64;
    const int m0 = blockIdx.x * 64;
    const int split = blockIdx.z;
    const int npix = p.NI * p.TH * p.TW;
    int4* ctab = (int4*)smem;                                  // [256] per virtual column: (offset, dy, dx, channel | valid<<8)
    unsigned char* at = smem + 4096;                           // [128][PITCH]  A^T image
    unsigned char* bt = at + (size_t)GC_NPIX * PITCH;          // [128][PITCH]  im2col image of one 64-column tile
    // halo patch of the SMALL operand (<= 4 channels) for the current pixel tile, padding rule applied: [img][c][PHh][PWw].
    // The im2col image is gathered from here (round 4); it used to be gathered from global memory element by element - 196
    // two-byte loads per pixel, issue-bound: 410 us for an 18.5 GFLOP layer.
    typedef typename std::conditional<std::is_same<T, float>::value, unsigned, unsigned short>::type PE;
    PE* pb = (PE*)(bt + (size_t)GC_NPIX * PITCH);
    const int thw = p.TH * p.TW;
    const float inv_thw = 1.0f / (float)thw, inv_tw = 1.0f / (float)p.TW;
    const unsigned bplane = (unsigned)(p.BH * p.BW);

    // The workgroup owns ALL virtual columns (tap*4 + channel) of its 64 rows: the big operand's tile (A) is staged once
    // per pixel tile and re-used by every 64-column tile (it used to be re-read by one workgroup per column tile: 4x the
    // HBM traffic of the layer's dominant tensor).
    const int tdy_min = p.grp[0].dy_min, tdx_min = p.grp[0].dx_min, tdy_max = p.grp[0].PH, tdx_max = p.grp[0].PW;
    const int PHh = (p.TH - 1) * p.ist + 1 + tdy_max - tdy_min, PWw = (p.TW - 1) * p.ist + 1 + tdx_max - tdx_min;
    const int npl = PHh * PWw;
    if (tid < 256) {
        const int col = tid;
        const int t = col >> p.cqs, cc = col & ((1 << p.cqs) - 1);
        const bool v = col < p.Cpad && t < p.ntaps_real && cc < p.creal;
        const int tt = t < p.ntaps_real ? t : 0;
        const int dy = p.tsign * (int)p.tap_dy[tt], dx = p.tsign * (int)p.tap_dx[tt];
        // .x: offset of (channel, tap) inside one image's patch
        ctab[col] = make_int4(v ? (cc * npl + (dy - tdy_min) * PWw + (dx - tdx_min)) : 0, dy, dx, (v ? cc : 0) | ((v ? 1 : 0) << 8));
    }

    f32x16_t acc[MAXCT];
#pragma unroll
    for (int c = 0; c < MAXCT; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    const int tile_lo = split * p.tiles_per_split;
    int tile_hi = tile_lo + p.tiles_per_split;
    if (tile_hi > p.ntiles) tile_hi = p.ntiles;
    constexpr int NCOL = std::is_same<T, float>::value ? NDW : 2 * NDW;
    const float inv_npl = 1.0f / (float)npl, inv_pww = 1.0f / (float)PWw;
    const int npatch = p.NI * p.creal * npl;

    for (int tile = tile_lo; tile < tile_hi; ++tile) {
        const int tx = tile % p.tiles_x;
        const int ty = (tile / p.tiles_x) % p.tiles_y;
        const int tn = tile / (p.tiles_x * p.tiles_y);
        const int u0 = ty * p.TH, v0 = tx * p.TW, n0 = tn * p.NI;
        __syncthreads();
        // A^T: the tile itself, sampled at (u + a_y0, v + a_x0) of the source tensor [N, M, a_h, a_w].  Pixels of
        // the tile that lie outside the (padded) domain must contribute nothing: they are zeroed via the B image.
        // (both 64-pixel rounds of the tile in flight at once: IM2COL_QB, A/B in tools/r05)
        stage_T<T, DWR, PITCH, IM2COL_QB>(at, p.a, p.a_f32, p.N, p.M, p.a_h, p.a_w, p.a_bmode, n0, p.NI, u0 + p.a_y0,
                                          v0 + p.a_x0, 0, p.TH, p.TW, m0, tid, 256);
        // the small operand's halo patch: rows y0 .. y0 + PHh - 1, columns x0 .. x0 + PWw - 1 of B (padding rule applied here,
        // so the gather below needs no bounds tests); consecutive threads take consecutive columns
        {
            const int y0 = u0 * p.ist + p.b_y0 + tdy_min, x0 = v0 * p.ist + p.b_x0 + tdx_min;
            // (eight loads in flight per thread: one per loop trip was a memory round trip per 256 elements)
            constexpr int SB = 8;
            for (int base = tid; base < npatch; base += 256 * SB) {
                unsigned off[SB], v[SB];
                bool ok[SB];
#pragma unroll
                for (int b = 0; b < SB; ++b) {
                    const int idx = base + 256 * b;
                    const int ci = (int)(((float)idx + 0.5f) * inv_npl);       // exact for idx < 2^22
                    const int r = idx - ci * npl;
                    const int yy = (int)(((float)r + 0.5f) * inv_pww);
                    const int xx = r - yy * PWw;
                    const int img = ci / p.creal, c = ci - img * p.creal;
                    int yb = y0 + yy, xb = x0 + xx;
                    if (p.bmode == PAD_REFLECT) { yb = reflect_idx(yb, p.BH); xb = reflect_idx(xb, p.BW); }
                    const int n = n0 + img;
                    ok[b] = idx < npatch && n < p.N && (unsigned)yb < (unsigned)p.BH && (unsigned)xb < (unsigned)p.BW;
                    off[b] = ok[b] ? ((unsigned)(n * p.creal + c) * bplane + (unsigned)(yb * p.BW + xb)) : 0u;
                }
#pragma unroll
                for (int b = 0; b < SB; ++b) {
                    if constexpr (BF32) v[b] = __float_as_uint(((const float*)p.b)[off[b]]);
                    else v[b] = ((const bf16_t*)p.b)[off[b]];
                }
#pragma unroll
                for (int b = 0; b < SB; ++b) {
                    const int idx = base + 256 * b;
                    unsigned x = v[b];
                    if constexpr (!std::is_same<T, float>::value && BF32) x = f2bf(__uint_as_float(x));
                    if (idx < npatch) pb[idx] = (PE)(ok[b] ? x : 0u);
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int ct = 0; ct < MAXCT; ++ct) {
            if (ct < nct) {
                if (ct > 0) __syncthreads();                   // the previous column tile's MFMAs are done with `bt`
                // im2col slice of column tile ct: rows = tile pixels, dword dw = wv + 4*i covers columns
                // (64*ct + 2*dw, + 1); the thread's column descriptors come from the LDS table
                int4 cd[NCOL];
#pragma unroll
                for (int k = 0; k < NCOL; ++k) {
                    const int col = ct * 64 + (std::is_same<T, float>::value ? (wv + 4 * k) : 2 * (wv + 4 * (k >> 1)) + (k & 1));
                    cd[k] = ctab[col];
                }
                unsigned colv = 0;
#pragma unroll
                for (int k = 0; k < NCOL; ++k) colv |= (unsigned)((cd[k].w >> 8) & 1) << k;
                for (int q = lane; q < npix; q += 64) {
                    const int img = (int)(((float)q + 0.5f) * inv_thw);
                    const int rem = q - img * thw;
                    const int tyy = (int)(((float)rem + 0.5f) * inv_tw);
                    const int txx = rem - tyy * p.TW;
                    const int n = n0 + img, ud = u0 + tyy, vd = v0 + txx;
                    unsigned raw[NCOL];
                    const bool pix_ok = (n < p.N) && (ud < p.AH) && (vd < p.AW);
                    const unsigned okm = pix_ok ? colv : 0u;
                    const int pixl = img * p.creal * npl + tyy * p.ist * PWw + txx * p.ist;
#pragma unroll
                    for (int k = 0; k < NCOL; ++k) raw[k] = pb[pixl + cd[k].x];
                    unsigned char* row = bt + (size_t)q * PITCH + wv * 4;
#pragma unroll
                    for (int i = 0; i < NDW; ++i) {
                        unsigned w;
                        if constexpr (std::is_same<T, float>::value) {
                            w = ((okm >> i) & 1u) ? raw[i] : 0u;
                        } else {
                            const unsigned l = raw[2 * i], h = raw[2 * i + 1];       // bf16 bits (converted at staging)
                            w = (((okm >> (2 * i)) & 1u) ? l : 0u) | ((((okm >> (2 * i + 1)) & 1u) ? h : 0u) << 16);
                        }
                        *(unsigned*)(row + i * 16) = w;
                    }
                }
                __syncthreads();
                for (int ks = 0; ks < npix / KS; ++ks) {
                    if constexpr (std::is_same<T, float>::value) {
                        const int r = ks * 2 + lhi;
                        const float a = *(const float*)(at + (size_t)r * PITCH + (wm * 32 + l31) * 4);
                        const float b = *(const float*)(bt + (size_t)r * PITCH + (wn * 32 + l31) * 4);
                        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[ct], 0, 0, 0);
                    } else {
                        const int g = lane >> 4, i16 = lane & 15;
                        const int rb = ks * 16 + (g >> 1) * 8 + (i16 >> 2);
                        const int colb = ((g & 1) * 16 + (i16 & 3) * 4) * 2;
                        typedef __attribute__((address_space(3))) short4_t* lds_s4;
                        typedef __attribute__((ext_vector_type(8))) short short8_t;
                        short4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(at + (size_t)rb * PITCH + wm * 64 + colb));
                        short4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(at + (size_t)(rb + 4) * PITCH + wm * 64 + colb));
                        short4_t b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(bt + (size_t)rb * PITCH + wn * 64 + colb));
                        short4_t b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4)(bt + (size_t)(rb + 4) * PITCH + wn * 64 + colb));
                        short8_t av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
                        short8_t bv = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
                        acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, av),
                                                                          __builtin_bit_cast(bf16x8_t, bv), acc[ct], 0, 0, 0);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int ct = 0; ct < MAXCT; ++ct) {
        if (ct < nct) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                const int c = ct * 64 + wn * 32 + l31;
                p.ws[((size_t)split * p.Mpad + m) * p.Cpad + c] = acc[ct][r];
            }
        }
    }
}

// im2col-mode finalize: dw[md*sm + cd*sc + r*sr + s*ss] (=|+=) sum_split ws[split][row][t*4 + col4]
//   normal: row = md (dY channel), col4 = cd;  swapped: row = cd (x channel), col4 = md
__global__ void wgrad_im2col_finalize_kernel(const WgParams p, int Md, int Cd) {
    const long long total = (long long)Md * Cd * p.ntaps_real;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int t = (int)(i % p.ntaps_real);
        const long long j = i / p.ntaps_real;
        const int cd = (int)(j % Cd);
        const int md = (int)(j / Cd);
        const int row = p.swap_out ? cd : md, col4 = p.swap_out ? md : cd;
        float s = 0.f;
        for (int sp = 0; sp < p.nsplit; ++sp) s += p.ws[((size_t)sp * p.Mpad + row) * p.Cpad + (t << p.cqs) + col4];
        const long long o = md * p.sm + cd * p.sc + p.tap_r[t] * p.sr + p.tap_s[t] * p.ss;
        if (p.accumulate) p.dw[o] += s; else p.dw[o] = s;
    }
}

// First reduction stage for many pixel splits: out[g][l] = sum over the splits of group g of ws[split][l], on the
// flat partial images (L4 float4 per split).  Fully coalesced; leaves <= 16 groups for the layout-changing finalize.
__global__ void wgrad_reduce_kernel(const float4* __restrict__ ws, float4* __restrict__ out, unsigned L4, int nsplit,
                                    int per_group) {
    const unsigned l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L4) return;
    const int sp0 = blockIdx.y * per_group;
    int sp1 = sp0 + per_group; if (sp1 > nsplit) sp1 = nsplit;
    float4 a = {0.f, 0.f, 0.f, 0.f};
    int sp = sp0;
    for (; sp + 4 <= sp1; sp += 4) {
        const float4 v0 = ws[(size_t)sp * L4 + l], v1 = ws[(size_t)(sp + 1) * L4 + l];
        const float4 v2 = ws[(size_t)(sp + 2) * L4 + l], v3 = ws[(size_t)(sp + 3) * L4 + l];
        a.x += (v0.x + v1.x) + (v2.x + v3.x); a.y += (v0.y + v1.y) + (v2.y + v3.y);
        a.z += (v0.z + v1.z) + (v2.z + v3.z); a.w += (v0.w + v1.w) + (v2.w + v3.w);
    }
    for (; sp < sp1; ++sp) {
        const float4 v = ws[(size_t)sp * L4 + l];
        a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
    out[(size_t)blockIdx.y * L4 + l] = a;
}

// Same sums for the plain conv layout (dw[m][c][t] with the taps in raster order, sc == ntaps): block (64 c, one m)
// reads the partial rows [t][c] coalesced along c, transposes through LDS and writes the 64 x ntaps run contiguously.
// (The generic kernel below reads with a Cpad*4-byte stride between neighbouring threads.)
__global__ __launch_bounds__(256) void wgrad_finalize_t_kernel(const WgParams p, float* __restrict__ dw, long long sm,
                                                               int accumulate) {
    extern __shared__ float fin_lds[];                     // [64][nt | 1]
    const int nt = p.ntaps, pitch = nt | 1;
    const int c0 = blockIdx.x * 64, m = blockIdx.y;
    const size_t sstride = (size_t)p.Mpad * nt * p.Cpad;
    for (int idx = threadIdx.x; idx < 64 * nt; idx += 256) {
        const int t = idx >> 6, c = idx & 63;
        const float* wp_ = p.ws + ((size_t)m * nt + t) * p.Cpad + c0 + c;
        float s = 0.f;
        int sp = 0;
        for (; sp + 4 <= p.nsplit; sp += 4) {
            const float v0 = wp_[(size_t)sp * sstride], v1 = wp_[(size_t)(sp + 1) * sstride];
            const float v2 = wp_[(size_t)(sp + 2) * sstride], v3 = wp_[(size_t)(sp + 3) * sstride];
            s += (v0 + v1) + (v2 + v3);
        }
        for (; sp < p.nsplit; ++sp) s += wp_[(size_t)sp * sstride];
        fin_lds[c * pitch + t] = s;
    }
    __syncthreads();
    const int cn = p.C - c0 < 64 ? p.C - c0 : 64;
    float* d = dw + (long long)m * sm + (long long)c0 * nt;
    for (int idx = threadIdx.x; idx < cn * nt; idx += 256) {
        const int c = idx / nt, t = idx - c * nt;
        const float v = fin_lds[c * pitch + t];
        if (accumulate) d[idx] += v; else d[idx] = v;
    }
}

// dw[m*sm + c*sc + r*sr + s*ss] (=|+=) sum_split ws[split][m][t][c]
__global__ void wgrad_finalize_kernel(const WgParams p, float* __restrict__ dw, long long sm, long long sc,
                                      long long sr, long long ss, int accumulate) {
    // thread per (m, c, t), t fastest: contiguous writes; 32-bit index math
    const unsigned nt = (unsigned)p.ntaps, C = (unsigned)p.C;
    const unsigned total = (unsigned)p.M * C * nt;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned j = i / nt, t = i - j * nt;
        const unsigned m = j / C, c = j - m * C;
        float s = 0.f;
        const size_t sstride = (size_t)p.Mpad * nt * p.Cpad;
        const float* wp_ = p.ws + ((size_t)m * nt + t) * p.Cpad + c;
        int sp = 0;
        for (; sp + 4 <= p.nsplit; sp += 4) {                  // 4 independent loads per trip
            const float v0 = wp_[(size_t)sp * sstride], v1 = wp_[(size_t)(sp + 1) * sstride];
            const float v2 = wp_[(size_t)(sp + 2) * sstride], v3 = wp_[(size_t)(sp + 3) * sstride];
            s += (v0 + v1) + (v2 + v3);
        }
        for (; sp < p.nsplit; ++sp) s += wp_[(size_t)sp * sstride];
        const long long o = m * sm + c * sc + p.tap_r[t] * sr + p.tap_s[t] * ss;
        if (accumulate) dw[o] += s; else dw[o] = s;
    }
}

// partial planes ws[split][m][tap][c] -> the weight-gradient layout (shared by the generic, pipelined and stride-2 kernels)
int gc_wgrad_finish(WgParams& p, float* dw, long long sm, long long sc, long long sr, long long ss, int accumulate,
                        WsAlloc& ws, hipStream_t st) {
    if (p.nsplit > 24) {
        // two-stage reduction: 16 coalesced group sums first, then the (strided) finalize over 16 partials
        const int ngrp = 16;
        const int per_group = cdiv(p.nsplit, ngrp);
        const int groups = cdiv(p.nsplit, per_group);
        const size_t L = (size_t)p.Mpad * p.ntaps * p.Cpad;          // multiple of 4 (Cpad % 64 == 0)
        float* ws2 = (float*)ws.take(groups * L * sizeof(float));
        if (ws2) {
            const unsigned L4 = (unsigned)(L / 4);
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv((int)L4, 256), groups), dim3(256), 0, st,
                               (const float4*)p.ws, (float4*)ws2, L4, p.nsplit, per_group);
            p.ws = ws2; p.nsplit = groups;
        }
    }
    bool raster = (sc == p.ntaps && ss == 1 && p.ntaps <= 32);
    for (int t = 0; t < p.ntaps && raster; ++t) raster = (p.tap_r[t] * sr + p.tap_s[t] * ss == t);
    if (raster) {
        const size_t lb = (size_t)64 * (p.ntaps | 1) * sizeof(float);
        hipLaunchKernelGGL(wgrad_finalize_t_kernel, dim3(cdiv(p.C, 64), p.M), dim3(256), lb, st, p, dw, sm, accumulate);
        return hific_launch_status();
    }
    long long total = (long long)p.M * p.C * p.ntaps;
    int gx = (int)((total + 255) / 256); if (gx > 16384) gx = 16384;
    hipLaunchKernelGGL(wgrad_finalize_kernel, dim3(gx), dim3(256), 0, st, p, dw, sm, sc, sr, ss, accumulate);
    return hific_launch_status();
}


// ---------------------------------------------------------------------------------------------------
template <typename T>
static int launch_wgrad_t(WgParams& p, float* dw, long long sm, long long sc, long long sr, long long ss,
                          int accumulate, WsAlloc& ws, hipStream_t st) {
    using Cfg = WgCfg<T>;
    if constexpr (std::is_same<T, bf16_t>::value) {
        int rc = gc_launch_wgrad_s2(p, dw, sm, sc, sr, ss, accumulate, ws, st);
        if (rc != HIFIC_ERR_UNSUPPORTED) return rc;
        rc = gc_launch_wgrad_s1(p, dw, sm, sc, sr, ss, accumulate, ws, st);
        if (rc != HIFIC_ERR_UNSUPPORTED) return rc;
    }
    p.Mpad = cdiv(p.M, 64) * 64; p.Cpad = cdiv(p.C, 64) * 64;
    p.dbg = env_int("HIFIC_DBG", 0);
    // tap groups of <= GC_TG consecutive taps
    p.ngroups = cdiv(p.ntaps, GC_TG);
    if (p.ngroups > GC_MAXPH) return HIFIC_ERR_UNSUPPORTED;
    int span_y = 1, span_x = 1;
    for (int gi = 0; gi < p.ngroups; ++gi) {
        GcPhase& gp = p.grp[gi];
        gp.tap0 = gi * GC_TG;
        gp.ntaps = p.ntaps - gp.tap0 < GC_TG ? p.ntaps - gp.tap0 : GC_TG;
        int dymin = 0, dymax = 0, dxmin = 0, dxmax = 0;
        for (int t = 0; t < gp.ntaps; ++t) {
            int dy = p.tap_dy[gp.tap0 + t], dx = p.tap_dx[gp.tap0 + t];
            if (t == 0) { dymin = dymax = dy; dxmin = dxmax = dx; }
            if (dy < dymin) dymin = dy; if (dy > dymax) dymax = dy;
            if (dx < dxmin) dxmin = dx; if (dx > dxmax) dxmax = dx;
        }
        gp.dy_min = dymin; gp.dx_min = dxmin;
        gp.PH = dymax - dymin + 1; gp.PW = dxmax - dxmin + 1;
        if (gp.PH > span_y) span_y = gp.PH;
        if (gp.PW > span_x) span_x = gp.PW;
    }
    const int fixed = 512 + GC_NPIX * Cfg::PITCH;
    // Two co-resident workgroups on half-LDS tiles (the kernels are register-capped at 256 for it) beat whole-LDS
    // tiles with every load of a tile in flight (stage_T QB = 12): 4.3 vs 6.3 ms per GAN cycle over the strided layers.
    const bool bigstage = std::is_same<T, bf16_t>::value && env_int("HIFIC_BIGSTAGE", 0);       // opt-in: measured slower
    if (!gc_choose_tile(p.N, p.AH, p.AW, p.ist, span_y, span_x, Cfg::PITCH, fixed, bigstage ? kLdsBudget : 72 * 1024,
                     p.TH, p.TW, p.NI, p.ntaps < GC_TG ? p.ntaps : GC_TG, true)) {
        // tiny odd planes: fall back to a (masked) 16-pixel-multiple tile wider than the plane
        p.TW = 16; p.TH = p.AH < 8 ? p.AH : 8; p.NI = 1;
        while ((p.TH * p.TW) % 16 != 0) ++p.TH;
    }
    size_t lds = 0;
    for (int gi = 0; gi < p.ngroups; ++gi) {
        GcPhase& gp = p.grp[gi];
        gp.PH = (p.TH - 1) * p.ist + gp.PH;
        gp.PW = (p.TW - 1) * p.ist + gp.PW;
        size_t b = (size_t)fixed + (size_t)p.NI * gp.PH * gp.PW * Cfg::PITCH;
        if (b > lds) lds = b;
    }
    // wide-load staging (stage_W) per operand: bf16 rows that are 16-byte aligned; + one shared dump row of LDS
    p.wstage_a = p.wstage_b = 0;
    if constexpr (std::is_same<T, bf16_t>::value) {
        const int wst = env_int("HIFIC_WSTAGE", 1);
        if ((wst == 2 || (wst == 1 && p.ist >= 2)) && lds + Cfg::PITCH <= (size_t)kLdsBudget) {
            // stride-2 layers only, and not the narrowest planes (512<-256 @16x16: 191 -> 204 us)
            p.wstage_a = !p.a_f32 && p.AW % 8 == 0 && p.AW >= env_int("HIFIC_WSTAGE_MINW", 32) && p.TW % 8 == 0 && ((size_t)p.a & 15) == 0;
            p.wstage_b = !p.b_f32 && p.BW % 8 == 0 && p.BW >= 2 * env_int("HIFIC_WSTAGE_MINW", 32) && ((size_t)p.b & 15) == 0;
            if (p.wstage_a || p.wstage_b) lds += Cfg::PITCH;
        }
    }
    if (lds > (size_t)kLdsBudget) return HIFIC_ERR_UNSUPPORTED;
    p.tiles_y = cdiv(p.AH, p.TH); p.tiles_x = cdiv(p.AW, p.TW); p.tiles_n = cdiv(p.N, p.NI);
    p.ntiles = p.tiles_n * p.tiles_y * p.tiles_x;
    const int base_blocks = (p.Mpad / 64) * (p.Cpad / 64) * p.ngroups;
    // enough (m,c,tap-group) tiles to fill the chip: no pixel split, epilogue writes the final layout directly
    int nsplit = 1;
    if (base_blocks < env_int("HIFIC_WG_NOSPLIT", 160)) {
        const int target = env_int("HIFIC_WG_TARGET", 0);
        if (target > 0) {
            nsplit = cdiv(target, base_blocks);
        } else {
            // Two workgroups co-reside per CU (512 slots): a launch of 746 workgroups runs as two rounds, the second one
            // half empty (60<-120 stride 2: 227 us at 746 workgroups, 187 us at exactly 512).  Pick the split that
            // minimises rounds x tiles per workgroup; every split also costs one partial tile of HBM traffic.
            double best = 1e30;
            const int nmax = p.ntiles < 4096 / base_blocks ? p.ntiles : 4096 / base_blocks;
            for (int n = 1; n <= nmax; ++n) {
                const int rounds = cdiv(base_blocks * n, 512);
                const int tps_ = cdiv(p.ntiles, n);
                const double cost = (double)rounds * tps_ + 0.02 * n;
                if (cost < best - 1e-9) { best = cost; nsplit = n; }
            }
        }
    }
    if (nsplit > p.ntiles) nsplit = p.ntiles;
    if (nsplit < 1) nsplit = 1;
    p.tiles_per_split = cdiv(p.ntiles, nsplit);
    p.nsplit = cdiv(p.ntiles, p.tiles_per_split);
    p.direct = p.nsplit == 1;
    p.dw = dw; p.sm = sm; p.sc = sc; p.sr = sr; p.ss = ss; p.accumulate = accumulate;
    if (!p.direct) {
        const size_t wsb = (size_t)p.nsplit * p.Mpad * p.ntaps * p.Cpad * sizeof(float);
        p.ws = (float*)ws.take(wsb);
        if (!p.ws) return HIFIC_ERR_WS;
    }
    dim3 grid((p.Mpad / 64) * (p.Cpad / 64), p.ngroups, p.nsplit);
    char ptag[112];
    snprintf(ptag, sizeof(ptag), "wgrad M%d C%d N%d a%dx%d taps%d ist%d tile%dx%dx%d split%d grid%d",
             p.M, p.C, p.N, p.AH, p.AW, p.ntaps, p.ist, p.NI, p.TH, p.TW, p.nsplit,
             (p.Mpad / 64) * (p.Cpad / 64) * p.ngroups * p.nsplit);
    bool pipe = false;
    if constexpr (std::is_same<T, bf16_t>::value) {
        int npatch_max = 0;
        for (int gi = 0; gi < p.ngroups; ++gi) {
            int np_ = p.NI * p.grp[gi].PH * p.grp[gi].PW;
            if (np_ > npatch_max) npatch_max = np_;
        }
        const size_t lds_pipe = 512 + 2 * (size_t)64 * (GC_NPIX * 2 + 16) + 2 * (((size_t)(npatch_max + 3) * 144 + 15) & ~(size_t)15);
        pipe = !p.a_f32 && !p.b_f32 && p.NI * p.TH * p.TW == GC_NPIX && p.TW % 8 == 0 && p.AW % 8 == 0 &&
               npatch_max <= 192 && lds_pipe <= (size_t)kLdsBudget && !env_int("HIFIC_NO_WGPIPE", 0);
    }
    const int pslot = gc_prof_open(pipe ? "wgrad_pipe_kernel" : (std::is_same<T, float>::value ? "wgrad_kernel<f32>" : "wgrad_kernel<bf16>"),
                                2.0 * p.M * p.C * p.ntaps * (double)p.N * p.AH * p.AW, st, ptag);
    gc_prof_bytes(pslot, (double)p.N * p.M * p.AH * p.AW * (p.a_f32 ? 4.0 : 2.0) + (double)p.N * p.C * p.BH * p.BW * (p.b_f32 ? 4.0 : 2.0) + (double)p.M * p.C * p.ntaps * 4.0);
    if constexpr (std::is_same<T, bf16_t>::value) {
        int npatch_max = 0;
        for (int gi = 0; gi < p.ngroups; ++gi) {
            int np_ = p.NI * p.grp[gi].PH * p.grp[gi].PW;
            if (np_ > npatch_max) npatch_max = np_;
        }
        const size_t lds_pipe = 512 + 2 * (size_t)64 * (GC_NPIX * 2 + 16) + 2 * (((size_t)(npatch_max + 3) * 144 + 15) & ~(size_t)15);
        if (pipe) {
            // tap-split 8-wave variant for a full 9-tap group (all 3x3 layers)
            const bool ts2 = p.ngroups == 1 && p.ntaps == GC_TG && env_int("HIFIC_WGPIPE_TS", 2) == 2;
            // shifted-fragment form: 3x3 window in (dy, dx) order with dx ascending by one patch pixel
            bool sh3 = p.ngroups == 1 && p.ntaps == 9 && p.ist == 1 && env_int("HIFIC_WGPIPE_SH3", 1);
            for (int d = 0; d < 3 && sh3; ++d)
                for (int j = 0; j < 3; ++j)
                    sh3 = sh3 && p.tap_dy[3 * d + j] == p.tap_dy[3 * d] && p.tap_dx[3 * d + j] == p.tap_dx[3 * d] + j;
#define WGP_LAUNCH(TS_, SH_)                                                                                       \
    do {                                                                                                           \
        gc_set_max_lds((const void*)wgrad_pipe_kernel<TS_, SH_>, (int)lds_pipe); \
        hipLaunchKernelGGL((wgrad_pipe_kernel<TS_, SH_>), grid, dim3(256 * TS_), lds_pipe, st, p);                 \
    } while (0)
            if (ts2) { if (sh3) WGP_LAUNCH(2, true); else WGP_LAUNCH(2, false); }
            else { if (sh3) WGP_LAUNCH(1, true); else WGP_LAUNCH(1, false); }
#undef WGP_LAUNCH
        }
    }
    if (!pipe) {
        void (*kfn)(const WgParams) = wgrad_kernel<T, 1>;
        if constexpr (std::is_same<T, bf16_t>::value) {
            if (bigstage) kfn = wgrad_kernel<T, 12>;
            else if (p.wstage_a || p.wstage_b) kfn = wgrad_kernel<T, -1>;
        }
        if (lds > 48 * 1024)
            gc_set_max_lds((const void*)kfn, (int)lds);
        hipLaunchKernelGGL(kfn, grid, dim3(256), lds, st, p);
    }
    gc_prof_close(pslot, st);
    int rc = hific_launch_status();
    if (rc != HIFIC_OK || p.direct) return rc;
    return gc_wgrad_finish(p, dw, sm, sc, sr, ss, accumulate, ws, st);
}

static int launch_wgrad(WgParams& p, int dtype, float* dw, long long sm, long long sc, long long sr, long long ss,
                        int accumulate, WsAlloc& ws, hipStream_t st) {
    if (dtype == HIFIC_F32) { p.a_f32 = 1; p.b_f32 = 1; return launch_wgrad_t<float>(p, dw, sm, sc, sr, ss, accumulate, ws, st); }
    if (dtype == HIFIC_BF16) return launch_wgrad_t<bf16_t>(p, dw, sm, sc, sr, ss, accumulate, ws, st);
    return HIFIC_ERR_ARG;
}

// small-channel path for nn.Conv2d weight gradients with stride 1 (see wgrad_im2col_kernel)
template <typename T>
static int launch_wgrad_im2col_t(const ConvGeom& g, const void* x, const void* dy, float* dw, int accumulate,
                                 int x_f32, int dy_f32, WsAlloc& ws, hipStream_t st) {
    using Cfg = WgCfg<T>;
    WgParams p; memset(&p, 0, sizeof(p));
    const bool swap = g.K <= 4 && g.C > 4;          // dY is the small operand
    const int small = swap ? g.K : g.C;
    const int cqs = small <= 4 ? 2 : 4;             // 4 or 16 channel slots per tap
    int nt = 0;
    for (int r = 0; r < g.R; ++r) for (int s = 0; s < g.S; ++s) {
        p.tap_dy[nt] = (short)(r - g.pt); p.tap_dx[nt] = (short)(s - g.pl); p.tap_r[nt] = (short)r; p.tap_s[nt] = (short)s; ++nt;
    }
    p.im2col = 1; p.ntaps_real = nt; p.ntaps = 1; p.ngroups = 1; p.ist = g.stride; p.N = g.N; p.cqs = cqs;
    if (swap && g.stride != 1) return HIFIC_ERR_UNSUPPORTED;
    p.swap_out = swap ? 1 : 0;
    const int Hp = g.H + g.pt + g.pb, Wp = g.W + g.pl + g.pr;
    if (!swap) {
        // A = dY [N,K,OH,OW] over its own domain; B = x with the conv's padding rule
        p.a = dy; p.a_f32 = dy_f32; p.M = g.K; p.a_h = g.OH(); p.a_w = g.OW(); p.a_bmode = PAD_ZERO; p.a_y0 = 0; p.a_x0 = 0;
        p.AH = g.OH(); p.AW = g.OW();
        p.b = x; p.b_f32 = x_f32; p.creal = g.C; p.BH = g.H; p.BW = g.W; p.bmode = g.pad_mode; p.b_y0 = 0; p.b_x0 = 0;
        p.tsign = 1;
    } else {
        // A' = padded x over the padded domain (origin -pt,-pl, conv's padding rule); B' = dY, zero outside
        p.a = x; p.a_f32 = x_f32; p.M = g.C; p.a_h = g.H; p.a_w = g.W; p.a_bmode = g.pad_mode; p.a_y0 = -g.pt; p.a_x0 = -g.pl;
        p.AH = Hp; p.AW = Wp;
        p.b = dy; p.b_f32 = dy_f32; p.creal = g.K; p.BH = g.OH(); p.BW = g.OW(); p.bmode = PAD_ZERO;
        p.b_y0 = -g.pt; p.b_x0 = -g.pl; p.tsign = -1;
    }
    p.C = nt << cqs;                                // virtual columns
    p.Mpad = cdiv(p.M, 64) * 64; p.Cpad = cdiv(p.C, 64) * 64;
    p.dbg = 0;
    {   // extent of the signed tap offsets (interior-tile test of the kernel's fast path)
        int ymin = 0, ymax = 0, xmin = 0, xmax = 0;
        for (int t = 0; t < nt; ++t) {
            const int dyv = p.tsign * p.tap_dy[t], dxv = p.tsign * p.tap_dx[t];
            if (t == 0) { ymin = ymax = dyv; xmin = xmax = dxv; }
            if (dyv < ymin) ymin = dyv; if (dyv > ymax) ymax = dyv;
            if (dxv < xmin) xmin = dxv; if (dxv > xmax) xmax = dxv;
        }
        p.grp[0].dy_min = ymin; p.grp[0].dx_min = xmin; p.grp[0].PH = ymax; p.grp[0].PW = xmax;
    }
    // pixel tile: 2 x 64 on wide planes (a 64-pixel bf16 row segment is a whole 128-byte line: 8x16 tiles made four
    // neighbouring tiles share every line of the big operand and thrashed L2: FETCH_SIZE 1.14 GB per launch for a
    // 126 MB tensor); 128 pixels = the whole K extent of a tile, multiple of 16 for the bf16 MFMA
    // (TW is 16 or 64 also on planes narrower than 16: TW = AW there needed TH rounded UP to a multiple-of-16 pixel count,
    // which could pass the GC_NPIX rows of the LDS images - 12-wide plane: 12 x 12 = 144 pixels; the columns past AW are masked)
    p.TW = p.AW < 64 ? 16 : 64; p.TH = GC_NPIX / p.TW; if (p.TH > p.AH) p.TH = p.AH; p.NI = 1;
    p.tiles_y = cdiv(p.AH, p.TH); p.tiles_x = cdiv(p.AW, p.TW); p.tiles_n = cdiv(p.N, p.NI);
    p.ntiles = p.tiles_n * p.tiles_y * p.tiles_x;
    if (p.Cpad / 64 > 4) return HIFIC_ERR_UNSUPPORTED;             // one workgroup covers all (<= 256) virtual columns
    const int base_blocks = p.Mpad / 64;
    // three workgroups per CU co-reside (103 registers, ~48 KB of LDS): one full round of 768 (round 5: 512 -> 768: 194 -> 168 us
    // on the first Encoder layer's weight gradient, 243 -> 207 us on the output layer's; 1024-2048 are slower again)
    int nsplit = cdiv(env_int("HIFIC_IM2COL_TARGET", 768), base_blocks);
    if (nsplit > p.ntiles) nsplit = p.ntiles;
    p.tiles_per_split = cdiv(p.ntiles, nsplit);
    p.nsplit = cdiv(p.ntiles, p.tiles_per_split);
    p.ws = (float*)ws.take((size_t)p.nsplit * p.Mpad * p.Cpad * sizeof(float));
    if (!p.ws) return HIFIC_ERR_WS;
    const long long RS = (long long)g.R * g.S;
    p.dw = dw; p.sm = (long long)g.C * RS; p.sc = RS; p.sr = g.S; p.ss = 1; p.accumulate = accumulate;
    // + the small operand's halo patch [NI][creal][TH + span_y - 1][TW + span_x - 1]
    const size_t lds = 4096 + 2 * (size_t)GC_NPIX * Cfg::PITCH +
                       (((size_t)p.NI * p.creal * ((p.TH - 1) * p.ist + 1 + p.grp[0].PH - p.grp[0].dy_min) *
                         ((p.TW - 1) * p.ist + 1 + p.grp[0].PW - p.grp[0].dx_min) * sizeof(T) + 15) & ~(size_t)15);
    if (lds > 160 * 1024) return HIFIC_ERR_UNSUPPORTED;
    dim3 grid(base_blocks, 1, p.nsplit);
    void (*kfn)(const WgParams) = (std::is_same<T, float>::value || p.b_f32) ? wgrad_im2col_kernel<T, true>
                                                                              : wgrad_im2col_kernel<T, false>;
    if (lds > 48 * 1024)
        gc_set_max_lds((const void*)kfn, (int)lds);
    char ptag[112];
    snprintf(ptag, sizeof(ptag), "wgrad_im2col K%d C%d N%d out%dx%d taps%d split%d", g.K, g.C, g.N, g.OH(), g.OW(), nt, p.nsplit);
    const int pslot = gc_prof_open(std::is_same<T, float>::value ? "wgrad_im2col_kernel<f32>" : "wgrad_im2col_kernel<bf16>",
                                2.0 * g.K * g.C * nt * (double)g.N * g.OH() * g.OW(), st, ptag);
    gc_prof_bytes(pslot, (double)g.N * g.K * g.OH() * g.OW() * (dy_f32 ? 4.0 : 2.0) + (double)g.N * g.C * g.H * g.W * (x_f32 ? 4.0 : 2.0) +
                         (double)g.K * g.C * nt * 4.0);
    hipLaunchKernelGGL(kfn, grid, dim3(256), lds, st, p);
    gc_prof_close(pslot, st);
    int rc = hific_launch_status();
    if (rc != HIFIC_OK) return rc;
    if (p.nsplit > 24) {
        const int per_group = cdiv(p.nsplit, 16), groups = cdiv(p.nsplit, per_group);
        const size_t L = (size_t)p.Mpad * p.Cpad;
        float* ws2 = (float*)ws.take(groups * L * sizeof(float));
        if (ws2) {
            const unsigned L4 = (unsigned)(L / 4);
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(cdiv((int)L4, 256), groups), dim3(256), 0, st,
                               (const float4*)p.ws, (float4*)ws2, L4, p.nsplit, per_group);
            p.ws = ws2; p.nsplit = groups;
        }
    }
    long long total = (long long)g.K * g.C * nt;
    int gx = (int)((total + 255) / 256); if (gx > 8192) gx = 8192;
    hipLaunchKernelGGL(wgrad_im2col_finalize_kernel, dim3(gx), dim3(256), 0, st, p, g.K, g.C);
    return hific_launch_status();
}

int gc_conv_bwd_weight(const ConvGeom& g, const void* x, const void* dy, float* dw, int accumulate,
                       int dtype, int x_f32, int dy_f32, WsAlloc& ws, hipStream_t st) {
    if (g.R * g.S > GC_MAXTAPS) return HIFIC_ERR_UNSUPPORTED;
    // im2col path: 7x7 layers with <= 4 channels on one side (stride 1), and (round 4) <= 16 INPUT channels with up to 16 taps at
    // stride 1 or 2 - the Discriminator's first layer (15 -> 64, 4x4 stride 2: 240 of its 256 virtual columns are real, where the
    // 64 x 64 tile of the generic kernel pads 15 channels to 64)
    const bool im2col4 = g.stride == 1 && g.R * g.S >= 9 && g.R * g.S * 4 <= 256 && (g.C <= 4 || g.K <= 4) && (g.C > 4 || g.K > 4);
    const bool im2col16 = !im2col4 && (g.stride == 1 || g.stride == 2) && g.C > 4 && g.C <= 16 && g.K > 16 && g.R * g.S >= 9 &&
                          g.R * g.S * 16 <= 256 && env_int("HIFIC_IM2COL16", 1);
    if (im2col16 && g.stride == 2 && dtype == HIFIC_BF16 && env_int("HIFIC_S2_FEWC", 1)) {
        // a stride-2 3x3 / 4x4 layer on a 16-pixel-multiple plane: the phase-decomposed kernel, although 15 of its 64 channel
        // columns are real - the layer is bound by its operand bytes, not by MFMA slots (Discriminator conv1: 199 us on the
        // im2col kernel, whose halo-patch staging is six dependent batches of two-byte loads per tile)
        WgParams q; memset(&q, 0, sizeof(q));
        q.a = dy; q.b = x; q.N = g.N; q.M = g.K; q.C = g.C; q.AH = g.OH(); q.AW = g.OW(); q.BH = g.H; q.BW = g.W;
        q.ist = g.stride; q.bmode = g.pad_mode; q.a_f32 = dy_f32; q.b_f32 = x_f32;
        int nq = 0;
        for (int r = 0; r < g.R; ++r) for (int s = 0; s < g.S; ++s) {
            q.tap_dy[nq] = (short)(r - g.pt); q.tap_dx[nq] = (short)(s - g.pl); q.tap_r[nq] = (short)r; q.tap_s[nq] = (short)s; ++nq;
        }
        q.ntaps = nq;
        const long long RSq = (long long)g.R * g.S;
        const int rc = gc_launch_wgrad_s2(q, dw, (long long)g.C * RSq, RSq, g.S, 1, accumulate, ws, st);
        if (rc != HIFIC_ERR_UNSUPPORTED) return rc;
    }
    if (im2col4 && g.C <= 4 && dtype == HIFIC_BF16) {
        // few INPUT channels on a big plane (the first Encoder layer): natural-order kernel (gconv_wgrad_c3.hip)
        const int rc = gc_launch_wgrad_c3(g, x, dy, dw, accumulate, x_f32, dy_f32, ws, st);
        if (rc != HIFIC_ERR_UNSUPPORTED) return rc;
    }
    if ((im2col4 || im2col16) && !env_int("HIFIC_NO_IM2COL", 0)) {
        int rc = HIFIC_ERR_ARG;
        if (dtype == HIFIC_F32) rc = launch_wgrad_im2col_t<float>(g, x, dy, dw, accumulate, 1, 1, ws, st);
        else if (dtype == HIFIC_BF16) rc = launch_wgrad_im2col_t<bf16_t>(g, x, dy, dw, accumulate, x_f32, dy_f32, ws, st);
        if (rc != HIFIC_ERR_UNSUPPORTED) return rc;          // (nothing was launched: the generic kernel takes it)
    }
    WgParams p; memset(&p, 0, sizeof(p));
    p.a = dy; p.b = x; p.N = g.N; p.M = g.K; p.C = g.C; p.AH = g.OH(); p.AW = g.OW(); p.BH = g.H; p.BW = g.W;
    p.ist = g.stride; p.bmode = g.pad_mode; p.a_f32 = dy_f32; p.b_f32 = x_f32;
    int nt = 0;
    for (int r = 0; r < g.R; ++r) for (int s = 0; s < g.S; ++s) {
        p.tap_dy[nt] = (short)(r - g.pt); p.tap_dx[nt] = (short)(s - g.pl); p.tap_r[nt] = (short)r; p.tap_s[nt] = (short)s; ++nt;
    }
    p.ntaps = nt;
    const long long RS = (long long)g.R * g.S;
    return launch_wgrad(p, dtype, dw, (long long)g.C * RS, RS, g.S, 1, accumulate, ws, st);
}

int gc_convT_bwd_weight(const ConvTGeom& g, const void* x, const void* dy, float* dw, int accumulate,
                        int dtype, int x_f32, int dy_f32, WsAlloc& ws, hipStream_t st) {
    if (g.R * g.S > GC_MAXTAPS) return HIFIC_ERR_UNSUPPORTED;
    WgParams p; memset(&p, 0, sizeof(p));
    // dw[ci,co,r,s] = sum x[ci,i] * dOut[co, st*i - pad + r]
    p.a = x; p.b = dy; p.N = g.N; p.M = g.Ci; p.C = g.Co; p.AH = g.H; p.AW = g.W; p.BH = g.OH(); p.BW = g.OW();
    p.ist = g.stride; p.bmode = PAD_ZERO; p.a_f32 = x_f32; p.b_f32 = dy_f32;
    int nt = 0;
    for (int r = 0; r < g.R; ++r) for (int s = 0; s < g.S; ++s) {
        p.tap_dy[nt] = (short)(r - g.pad); p.tap_dx[nt] = (short)(s - g.pad); p.tap_r[nt] = (short)r; p.tap_s[nt] = (short)s; ++nt;
    }
    p.ntaps = nt;
    const long long RS = (long long)g.R * g.S;
    return launch_wgrad(p, dtype, dw, (long long)g.Co * RS, RS, g.S, 1, accumulate, ws, st);
}

