// Two special-structure forward-type kernels of the conv engine and their launchers:
//   gconv_mp_kernel - the stride-2 TRANSPOSED structure (conv-transpose forward, stride-2 data gradients) with the two column
//                     phases of an output-row parity merged in one workgroup (src/network/generator.py:115-137, the data gradients
//                     of encoder.py:64-93 and discriminator.py:53-62);
//   gconv_vc_kernel - few input channels with many taps on the dense virtual column c * R*S + tap (LPIPS / AlexNet conv1:
//                     src/loss/perceptual_similarity/pretrained_networks.py:59-75).
#include "gconv.h"
#include "gconv_dev.h"
#include "gconv_stage.h"
#include <type_traits>
#include <string.h>
#include <stdio.h>

// ---------------------------------------------------------------------------------------------------
// Merged-phase kernel for the stride-2 TRANSPOSED structure (round 4): conv-transpose forward and the data gradient of a
// stride-2 convolution are four sub-pixel phases (py, px) over the same (u, v) input domain.  gconv_kernel runs them as four
// workgroups per tile (blockIdx.z): each stages the SAME halo patch, and each writes every other pixel of every other
// output row - 2-byte stores at a 4-byte stride.  Timing ablation of 60 <- 120 @128 -> 256 (tools/micro_conv.py, HIFIC_DBG):
// of 273 us, staging 88, epilogue 74, launch + barriers of the 8192 workgroups 54, MFMA + fragment reads 29, weights 12.
// Here ONE workgroup owns a (u, v) tile for the two column phases of an output row parity: the union halo patch is staged once
// per channel chunk for both, their taps stream through the same weight ring as one step sequence (accumulator set chosen by a
// uniform branch per step: the set index must be a compile-time constant or the accumulators go to scratch), and the epilogue
// writes the two column phases of a pixel as ONE 4-byte (bf16) / 8-byte (f32) store: 32 lanes = 128 contiguous bytes.  64-row
// tiles, 64-channel chunks, two workgroups per CU, grid.z = 2 (row parity).
// p.ph[0..3]: the phases (py-major, as the planners emit them); p.ph[4]: union patch / tile grid (host); p.epi_wide == 2:
// pair stores are legal (no fold / residual, even output width, both column phases of a row have the same domain).
// SPLIT: operands in the pair layout of the exact-index chain (see gconv_kernel).
template <bool F32>
__device__ __forceinline__ void mp_store_pair(const GcParams& p, const GcPhase& phA, const f32x16_t a, const f32x16_t b,
                                              int mbase, int lhi, int pu_, int pv_, int pn_, bool pvalid_, bool hb,
                                              const float* bp, float slope) {
    float bv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = mbase + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        bv[r] = bp[(hb && m < p.K) ? m : 0];
    }
    const float osc = p.oscale ? *p.oscale : 1.f;
    const int oy = pu_ * 2 + phA.ooy, ox = pv_ * 2 + phA.oox;
    const bool okp = pvalid_ && pn_ < p.N && pu_ < phA.OHt && pv_ < phA.OWt && (unsigned)oy < (unsigned)p.OHf &&
                     (unsigned)(ox + 1) < (unsigned)p.OWf;
    if (!okp) return;
    // 32-bit element offsets (the plan checks N K OH OW < 2^31); the row bound is tested once per wave when the whole 32-row
    // block is inside (per-row branches with 64-bit index arithmetic were most of this kernel's VALU work)
    const unsigned plane = (unsigned)(p.OHf * p.OWf);
    const unsigned pb32 = (unsigned)(pn_ * p.K + mbase + 4 * lhi) * plane + (unsigned)(oy * p.OWf + ox);
    const bool full = mbase + 32 <= p.K;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int kr = (r & 3) + 8 * (r >> 2);
        const int m = mbase + kr + 4 * lhi;
        const float b_ = (hb && (full || m < p.K)) ? bv[r] : 0.f;
        float x0 = a[r] * osc + b_, x1 = b[r] * osc + b_;
        x0 = x0 > 0.f ? x0 : x0 * slope;
        x1 = x1 > 0.f ? x1 : x1 * slope;
        if (full || m < p.K) {
            const unsigned idx = pb32 + (unsigned)kr * plane;
            if constexpr (F32) *(float2*)((float*)p.out + idx) = make_float2(x0, x1);
            else *(unsigned*)((bf16_t*)p.out + idx) = f2bf2(x0, x1);
        }
    }
}

// Pair stores for a REFLECT-FOLD data gradient (p.fold_h: interior pixels of the padded plane go straight to dx = out2, the rim
// to the padded float32 buffer `out`) when the left pad is even: column phase 0 of pixel v is padded column 2v, phase 1 is
// 2v + 1 - an aligned pair that lies entirely inside or entirely outside the interior (even origin, even width).  The last
// padded column (odd plane width) has no partner and is stored alone.
template <bool F32O2>
__device__ __forceinline__ void mp_store_pair_fold(const GcParams& p, const GcPhase& phA, const GcPhase& phB, const f32x16_t a,
                                                   const f32x16_t b, int mbase, int lhi, int pu_, int pv_, int pn_,
                                                   bool pvalid_) {
    const float osc = p.oscale ? *p.oscale : 1.f;
    const int oy = pu_ * 2 + phA.ooy, ox = pv_ * 2 + phA.oox;
    const bool okn = pvalid_ && pn_ < p.N && (unsigned)oy < (unsigned)p.OHf;
    const bool inA = okn && pu_ < phA.OHt && pv_ < phA.OWt && (unsigned)ox < (unsigned)p.OWf;
    const bool inB = okn && pu_ < phB.OHt && pv_ < phB.OWt && (unsigned)(ox + 1) < (unsigned)p.OWf;
    if (!inA && !inB) return;
    const int iy = oy - p.fold_pt, ix = ox - p.fold_pl;
    const bool rowi = (unsigned)iy < (unsigned)p.fold_h;
    const bool intA = inA && rowi && (unsigned)ix < (unsigned)p.fold_w;
    const bool intB = inB && rowi && (unsigned)(ix + 1) < (unsigned)p.fold_w;
    const size_t plane_p = (size_t)p.OHf * p.OWf, plane_i = (size_t)p.fold_h * p.fold_w;
    const size_t base_p = (size_t)pn_ * p.K * plane_p + (size_t)oy * p.OWf + ox;
    const size_t base_i = (size_t)pn_ * p.K * plane_i + (size_t)(rowi ? iy : 0) * p.fold_w + (intA || intB ? ix : 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = mbase + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (m >= p.K) continue;
        const float x0 = a[r] * osc, x1 = b[r] * osc;
        if (intA && intB) {
            const size_t idx = base_i + (size_t)m * plane_i;
            if constexpr (F32O2) *(float2*)((float*)p.out2 + idx) = make_float2(x0, x1);
            else *(unsigned*)((bf16_t*)p.out2 + idx) = f2bf2(x0, x1);
        } else {
            if (inA) {
                if (intA) { if constexpr (F32O2) ((float*)p.out2)[base_i + (size_t)m * plane_i] = x0;
                            else ((bf16_t*)p.out2)[base_i + (size_t)m * plane_i] = f2bf(x0); }
                else ((float*)p.out)[base_p + (size_t)m * plane_p] = x0;
            }
            if (inB) {
                if (intB) { if constexpr (F32O2) ((float*)p.out2)[base_i + (size_t)m * plane_i + 1] = x1;
                            else ((bf16_t*)p.out2)[base_i + (size_t)m * plane_i + 1] = f2bf(x1); }
                else ((float*)p.out)[base_p + (size_t)m * plane_p + 1] = x1;
            }
        }
    }
}

template <bool SPLIT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void gconv_mp_kernel(const GcParams p) {
    // A workgroup owns the two COLUMN phases (py, 0), (py, 1) of its tile (py = blockIdx.z): 4 waves as 2 x 2, two pixel
    // fragments per wave and phase = 64 accumulator registers.  (All four phases in one workgroup - 128 accumulator registers
    // on four waves, or 64 on eight waves at four waves per SIMD - spilled an accumulator fragment in every step on this
    // compiler; the row-pair form stages the patch twice instead of four times and keeps the pair stores.)
    constexpr int BC = 64, KS = 16, PITCH = 144, DWR = 32, PPR = 8, BM = 64, WN = 2, WGN = 2;
    constexpr int WBYTES = BM * PITCH;
    constexpr int NWP = 2;                          // 16-byte weight pieces per thread per step (64 rows x 8 pieces / 256)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int l31 = lane & 31, lhi = lane >> 5;
    const GcPhase& U = p.ph[4];
    const int py2 = (int)blockIdx.z * 2;
    const GcPhase& PA = p.ph[py2];
    const GcPhase& PB = p.ph[py2 + 1];
    const int ntile = p.tiles_n * U.tiles_y * U.tiles_x;
    int tile, mtile;
    {
        const int nwg = gridDim.x;
        const int q8 = nwg >> 3, r8 = nwg & 7;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int q = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
        mtile = q / p.max_tiles;
        tile = q - mtile * p.max_tiles;
    }
    if (tile >= ntile) return;
    const int tx = tile % U.tiles_x;
    const int ty = (tile / U.tiles_x) % U.tiles_y;
    const int tn = tile / (U.tiles_x * U.tiles_y);
    const int u0 = ty * p.TH, v0 = tx * p.TW, n0 = tn * p.NI;
    const int m0 = mtile * BM;
    const int PH = U.PH, PW = U.PW, PWs = U.PWs;
    const int npps = PH * PWs;
    const int iy0 = u0 + U.dy_min, ix0 = v0 + U.dx_min;
    // the taps of the two phases are contiguous in the tap table: [gt0, gt0 + T), the first t1 of them belong to phase A
    const int gt0 = PA.tap0, t1 = PA.ntaps, T = PA.ntaps + PB.ntaps;

    int* toffs = (int*)smem;
    unsigned char* wbuf = smem + 512;
    unsigned char* patch = wbuf + 2 * WBYTES;
    if (tid < T) toffs[tid] = ((int)p.tap_dy[gt0 + tid] - U.dy_min) * PWs + ((int)p.tap_dx[gt0 + tid] - U.dx_min);

    int qb[WN], pu[WN], pv[WN], pn[WN];
    bool pvalid[WN];
    const int thw = p.TH * p.TW;
#pragma unroll
    for (int ni = 0; ni < WN; ++ni) {
        const int pt = (wn * WN + ni) * 32 + l31;
        const int img = pt / thw;
        const int rem = pt - img * thw;
        const int ty_ = rem / p.TW;
        const int tx_ = rem - ty_ * p.TW;
        const bool v = img < p.NI;
        pvalid[ni] = v;
        qb[ni] = v ? (img * npps + ty_ * PWs + tx_) : 0;
        pu[ni] = u0 + ty_; pv[ni] = v0 + tx_; pn[ni] = n0 + img;
    }
    f32x16_t acc0[WN], acc1[WN];
#pragma unroll
    for (int ni = 0; ni < WN; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[ni][r] = 0.f; acc1[ni][r] = 0.f; }

    const int nchunks = p.Cpad / BC;
    const int nsteps = nchunks * T;
    // per-phase weight images [Kpad][ntaps_ph][Cpad]: base and row pitch in scalars; 32-bit byte offsets (far below 4 GB)
    const unsigned char* wpb = (const unsigned char*)p.wp;
    const unsigned b0 = (unsigned)PA.wp_off * 2u, b1 = (unsigned)PB.wp_off * 2u;
    const unsigned r0 = (unsigned)(PA.ntaps * p.Cpad * 2), r1 = (unsigned)(PB.ntaps * p.Cpad * 2);
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t wA[NWP], wB[NWP];
    // piece i of a thread: row tid / 8 + 32 i, 16-byte part tid % 8
    const unsigned wp16 = (unsigned)((tid & 7) * 16);
    const unsigned wlds0 = (unsigned)((tid >> 3) * PITCH) + wp16;
    unsigned wmrow[NWP];
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
        const int wrow = (tid >> 3) + 32 * i;
        wmrow[i] = (unsigned)(m0 + wrow < p.K ? m0 + wrow : p.K - 1);           // padded rows re-read row K-1 (never stored)
    }
#define MP_WLOAD(R, S_)                                                                     \
    do {                                                                                    \
        int s_ = (S_) < nsteps ? (S_) : nsteps - 1;                                         \
        const int c_ = s_ / T;                                                              \
        const int g_ = s_ - c_ * T;                                                         \
        const unsigned base_ = g_ < t1 ? b0 : b1;                                           \
        const unsigned rowb_ = g_ < t1 ? r0 : r1;                                           \
        const int tl_ = g_ < t1 ? g_ : g_ - t1;                                             \
        const unsigned off_ = base_ + (unsigned)(tl_ * p.Cpad + c_ * BC) * 2u;              \
        _Pragma("unroll") for (int i = 0; i < NWP; ++i)                                     \
            R[i] = *(const u32x4_t*)(wpb + (off_ + wmrow[i] * rowb_ + wp16));               \
    } while (0)
#define MP_WSTORE(R, BUF)                                                                   \
    do {                                                                                    \
        _Pragma("unroll") for (int i = 0; i < NWP; ++i)                                     \
            *(u32x4_t*)((BUF) + wlds0 + i * 32 * PITCH) = R[i];                             \
    } while (0)
#define MP_COMPUTE(ACC, wb, toff)                                                                               \
    do {                                                                                                        \
        const unsigned char* arow = (wb) + (wm * 32 + l31) * PITCH;                                             \
        if constexpr (!SPLIT) {                                                                                 \
            _Pragma("unroll") for (int kk = 0; kk < BC / KS; ++kk) {                                            \
                const bf16x8_t a = *(const bf16x8_t*)(arow + kk * 32 + lhi * 16);                               \
                _Pragma("unroll") for (int ni = 0; ni < WN; ++ni) {                                             \
                    const bf16x8_t b = *(const bf16x8_t*)(patch + (size_t)(qb[ni] + (toff)) * PITCH + kk * 32 + lhi * 16); \
                    ACC[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, ACC[ni], 0, 0, 0);                  \
                }                                                                                               \
            }                                                                                                   \
        } else {                                                                                                \
            _Pragma("unroll") for (int kk = 0; kk < BC / KS; kk += 2) {                                         \
                const bf16x8_t a = *(const bf16x8_t*)(arow + kk * 32 + lhi * 16);                               \
                const bf16x8_t al = *(const bf16x8_t*)(arow + (kk + 1) * 32 + lhi * 16);                        \
                _Pragma("unroll") for (int ni = 0; ni < WN; ++ni) {                                             \
                    const bf16x8_t b = *(const bf16x8_t*)(patch + (size_t)(qb[ni] + (toff)) * PITCH + kk * 32 + lhi * 16); \
                    const bf16x8_t bl = *(const bf16x8_t*)(patch + (size_t)(qb[ni] + (toff)) * PITCH + (kk + 1) * 32 + lhi * 16); \
                    ACC[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, b, ACC[ni], 0, 0, 0);                 \
                    ACC[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, bl, ACC[ni], 0, 0, 0);                 \
                    ACC[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, ACC[ni], 0, 0, 0);                  \
                }                                                                                               \
            }                                                                                                   \
        }                                                                                                       \
    } while (0)
#define MP_STEP(s, RL, RS)                                                                  \
    do {                                                                                    \
        if (g == 0) {                                                                       \
            __syncthreads();                                                                \
            stage_T<bf16_t, DWR, PITCH, 2>(patch, p.in, 0, p.N, p.C, p.IH, p.IW, p.bmode, n0, p.NI, iy0, ix0, PWs, \
                                           PH, PW, chunk * BC, tid, 256);                   \
        }                                                                                   \
        __syncthreads();                                                                    \
        MP_WLOAD(RL, (s) + 2);                                                              \
        const int toff = toffs[g];                                                          \
        const unsigned char* wb_ = wbuf + ((s) & 1) * WBYTES;                               \
        if (g < t1) MP_COMPUTE(acc0, wb_, toff);                                            \
        else MP_COMPUTE(acc1, wb_, toff);                                                   \
        MP_WSTORE(RS, wbuf + (((s) + 1) & 1) * WBYTES);                                     \
        if (++g == T) { g = 0; ++chunk; }                                                   \
    } while (0)

    if (nsteps > 0) {
        MP_WLOAD(wA, 0);
        MP_WLOAD(wB, 1);
        MP_WSTORE(wA, wbuf);
        int chunk = 0, g = 0, s = 0;
        for (; s + 1 < nsteps; s += 2) {
            MP_STEP(s, wA, wB);
            MP_STEP(s + 1, wB, wA);
        }
        if (s < nsteps) MP_STEP(s, wA, wB);
    }
#undef MP_STEP
#undef MP_COMPUTE
#undef MP_WSTORE
#undef MP_WLOAD
    const int mbase = m0 + wm * 32;
    if (p.epi_wide == 3) {
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) {
            if (p.out2_f32) mp_store_pair_fold<true>(p, PA, PB, acc0[ni], acc1[ni], mbase, lhi, pu[ni], pv[ni], pn[ni], pvalid[ni]);
            else mp_store_pair_fold<false>(p, PA, PB, acc0[ni], acc1[ni], mbase, lhi, pu[ni], pv[ni], pn[ni], pvalid[ni]);
        }
        return;
    }
    if (p.epi_wide == 4) {
        // Wide pair stores (round 6): the pair store above is ONE 4-byte store per (row, pixel) - 32 stores per lane and tile,
        // 67 of the 169 us of the 60 <- 120 @128 -> 256 layer in the round-4 ablation.  Here every wave interleaves the two column
        // phases of its 32 rows x 64 pixels through a private LDS region ([row][64 px x 2 phases] bf16, pitch 288 B: the two
        // half-waves land on disjoint banks) and stores 16-byte pieces = 4 pixels x 2 phases of one row: 8 stores per lane.
        // Planner guarantees: bf16 output, TW % 4 == 0, OWt % 4 == 0 (a piece never straddles a tile row or the image edge),
        // OWf % 8 == 0 and a 16-byte aligned output (every piece is 16-byte aligned), column phases at oox = 0 / 1.
        constexpr int EP = 288;
        const bool hb = p.bias != nullptr;
        const float* bp = hb ? p.bias : (const float*)p.in;
        const float slope = p.act == ACT_RELU ? 0.f : (p.act == ACT_LEAKY ? 0.2f : 1.f);
        const float osc = p.oscale ? *p.oscale : 1.f;
        __syncthreads();                                   // every wave is done with the patch / weight ring
        unsigned char* reg = smem + wave * (32 * EP);
        float bv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mbase + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            bv[r] = (hb && m < p.K) ? bp[m] : 0.f;
        }
#pragma unroll
        for (int ni = 0; ni < WN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rl = (r & 3) + 8 * (r >> 2) + 4 * lhi;
                float x0 = acc0[ni][r] * osc + bv[r], x1 = acc1[ni][r] * osc + bv[r];
                x0 = x0 > 0.f ? x0 : x0 * slope;
                x1 = x1 > 0.f ? x1 : x1 * slope;
                *(unsigned*)(reg + rl * EP + (ni * 32 + l31) * 4) = f2bf2(x0, x1);
            }
        __syncthreads();
        // lane -> pixel group j = lane % 16 (4 consecutive pixels of one tile row), rows lane / 16 + 4 i
        const int j = lane & 15;
        const int pt = (wn * WN) * 32 + 4 * j;
        const int img = pt / thw;
        const int rem = pt - img * thw;
        const int ty_ = rem / p.TW, tx_ = rem - ty_ * p.TW;
        const int u_ = u0 + ty_, v_ = v0 + tx_, n_ = n0 + img;
        const bool okp = img < p.NI && n_ < p.N && u_ < PA.OHt && v_ < PA.OWt;
        const int oy = u_ * 2 + PA.ooy, ox = v_ * 2;
        const unsigned plane = (unsigned)(p.OHf * p.OWf);
        if (okp && (unsigned)oy < (unsigned)p.OHf) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int rl = (lane >> 4) + 4 * i;
                const int m = mbase + rl;
                if (m < p.K) {
                    const u32x4_t v = *(const u32x4_t*)(reg + rl * EP + j * 16);
                    const unsigned idx = (unsigned)(n_ * p.K + m) * plane + (unsigned)(oy * p.OWf + ox);
                    *(u32x4_t*)((bf16_t*)p.out + idx) = v;
                }
            }
        }
        return;
    }
    if (p.epi_wide == 2) {
        const bool hb = p.bias != nullptr;
        const float* bp = hb ? p.bias : (const float*)p.in;
        const float slope = p.act == ACT_RELU ? 0.f : (p.act == ACT_LEAKY ? 0.2f : 1.f);
#pragma unroll
        for (int ni = 0; ni < WN; ++ni) {
            if (p.out_f32)
                mp_store_pair<true>(p, PA, acc0[ni], acc1[ni], mbase, lhi, pu[ni], pv[ni], pn[ni], pvalid[ni], hb, bp, slope);
            else
                mp_store_pair<false>(p, PA, acc0[ni], acc1[ni], mbase, lhi, pu[ni], pv[ni], pn[ni], pvalid[ni], hb, bp, slope);
        }
        return;
    }
    gc_epilogue<false, 1, WN, -1>(p, PA, acc0[0], acc0[1], acc0[0], acc0[1], mbase, lhi, pu, pv, pn, pvalid);
    gc_epilogue<false, 1, WN, -1>(p, PB, acc1[0], acc1[1], acc1[0], acc1[1], mbase, lhi, pu, pv, pn, pvalid);
}


// ---------------------------------------------------------------------------------------------------
// Virtual-column forward kernel for FEW-CHANNEL inputs (round 4): LPIPS/AlexNet conv1 (3 -> 64, 11x11 stride 4), the first
// Encoder layer in the exact-index chain (9 = 3 x 3 split channels -> 60, 7x7), the Discriminator's first layer (15 -> 64, 4x4
// stride 2).  An implicit GEMM spends one 16-deep MFMA slice per TAP on C useful channels (3/16 .. 9/16 of the work, 49-121
// barrier steps): 248 us for the 5.9 GFLOP of AlexNet conv1, 291 us for the first Encoder layer.  Here the reduction index is
// the dense virtual column j = c * R*S + tap - the weight tensor's own memory order, so the packed operand is just the weight
// matrix [K][C*R*S] in bf16 (packed by the ordinary 1x1 pack path) - and per 64-column chunk every thread GATHERS its part of
// the im2col image [128 pixels][64 columns] from an LDS-resident halo patch of the input tile ([img][c][rows][cols], padding
// rule applied while staging); the MFMAs then run exactly like gconv_kernel's on a one-tap "patch".  ceil(C*R*S / 64) steps of
// 4 K-slices instead of R*S steps of one mostly-empty slice.
// p.ph[0]: the phase (PH/PW = halo patch of the tile); p.Cpad = padded C*R*S; taps in (r, s) order; ost = 1.
template <bool F32SRC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void gconv_vc_kernel(const GcParams p) {
    constexpr int PITCH = 144, BM = 64, WN = 2, WGN = 2, PPR = 8;
    constexpr int WBYTES = BM * PITCH;
    constexpr int NWP = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int wv = tid >> 6;
    const GcPhase& ph = p.ph[0];
    const int ntile = p.tiles_n * ph.tiles_y * ph.tiles_x;
    int tile, mtile;
    {
        const int nwg = gridDim.x;
        const int q8 = nwg >> 3, r8 = nwg & 7;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int q = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
        mtile = q / p.max_tiles;
        tile = q - mtile * p.max_tiles;
    }
    if (tile >= ntile) return;
    const int tx = tile % ph.tiles_x;
    const int ty = (tile / ph.tiles_x) % ph.tiles_y;
    const int tn = tile / (ph.tiles_x * ph.tiles_y);
    const int u0 = ty * p.TH, v0 = tx * p.TW, n0 = tn * p.NI;
    const int m0 = mtile * BM;
    const int PHh = ph.PH, PWw = ph.PW;
    const int npl = PHh * PWw;
    const int T = ph.ntaps;
    const int J = p.C * T;                                     // real virtual columns; p.Cpad = J rounded up to 64
    const int nch = p.Cpad / 64;

    int* jtab = (int*)smem;                                    // [Cpad] patch offset of column j, -1 for the padding
    unsigned char* wt = smem + (((size_t)p.Cpad * 4 + 15) & ~(size_t)15);      // 2 x WBYTES
    unsigned char* bt = wt + 2 * WBYTES;                       // [128][PITCH] im2col chunk
    unsigned short* pb = (unsigned short*)(bt + (size_t)GC_NPIX * PITCH);
    for (int j = tid; j < p.Cpad; j += 256) {
        const int c = j / T, t = j - c * T;
        jtab[j] = j < J ? (c * npl + ((int)p.tap_dy[t] - ph.dy_min) * PWw + ((int)p.tap_dx[t] - ph.dx_min)) : -1;
    }
    // halo patch of the tile, padding rule applied, bf16: rows y0 .., columns x0 ..
    {
        const int y0 = u0 * p.ist + ph.dy_min, x0 = v0 * p.ist + ph.dx_min;
        const int npatch = p.NI * p.C * npl;
        const float inv_npl = 1.0f / (float)npl, inv_pww = 1.0f / (float)PWw;
        const unsigned plane = (unsigned)(p.IH * p.IW);
        // eight loads in flight per thread (one load per loop trip was one memory round trip per 256 elements: 20 serialised
        // round trips for the 5040-element patch of the first Encoder layer, 300 us of the launch)
        constexpr int SB = 8;
        for (int base = tid; base < npatch && !(p.dbg & 64); base += 256 * SB) {
            unsigned off[SB], v[SB];
            bool ok[SB];
#pragma unroll
            for (int b = 0; b < SB; ++b) {
                const int idx = base + 256 * b;
                const int ci = (int)(((float)idx + 0.5f) * inv_npl);           // exact for idx < 2^22
                const int r = idx - ci * npl;
                const int yy = (int)(((float)r + 0.5f) * inv_pww);
                const int xx = r - yy * PWw;
                const int img = ci / p.C, c = ci - img * p.C;
                int yb = y0 + yy, xb = x0 + xx;
                if (p.bmode == PAD_REFLECT) { yb = reflect_idx(yb, p.IH); xb = reflect_idx(xb, p.IW); }
                const int n = n0 + img;
                ok[b] = idx < npatch && n < p.N && (unsigned)yb < (unsigned)p.IH && (unsigned)xb < (unsigned)p.IW;
                off[b] = ok[b] ? ((unsigned)(n * p.C + c) * plane + (unsigned)(yb * p.IW + xb)) : 0u;
            }
#pragma unroll
            for (int b = 0; b < SB; ++b) {
                if constexpr (F32SRC) v[b] = __float_as_uint(((const float*)p.in)[off[b]]);
                else v[b] = ((const bf16_t*)p.in)[off[b]];
            }
#pragma unroll
            for (int b = 0; b < SB; ++b) {
                const int idx = base + 256 * b;
                unsigned x = v[b];
                if constexpr (F32SRC) x = f2bf(__uint_as_float(x));
                if (idx < npatch) pb[idx] = (unsigned short)(ok[b] ? x : 0u);
            }
        }
    }
    // gather role: pixels q = lane, lane + 64 of the tile; dword columns wv + 4 i of a chunk
    const int thw = p.TH * p.TW;
    int pixl[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int q = lane + 64 * k;
        const int img = q / thw;
        const int rem = q - img * thw;
        const int tyy = rem / p.TW, txx = rem - tyy * p.TW;
        pixl[k] = img < p.NI ? (img * p.C * npl + tyy * p.ist * PWw + txx * p.ist) : 0;
    }
    // MFMA role
    int pu[WN], pv[WN], pn[WN];
    bool pvalid[WN];
#pragma unroll
    for (int ni = 0; ni < WN; ++ni) {
        const int pt = (wn * WN + ni) * 32 + l31;
        const int img = pt / thw;
        const int rem = pt - img * thw;
        const int ty_ = rem / p.TW;
        const int tx_ = rem - ty_ * p.TW;
        pvalid[ni] = img < p.NI;
        pu[ni] = u0 + ty_; pv[ni] = v0 + tx_; pn[ni] = n0 + img;
    }
    f32x16_t acc[WN];
#pragma unroll
    for (int ni = 0; ni < WN; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ni][r] = 0.f;

    // weight tile of chunk ch: rows m0 .. m0 + 63, columns 64 ch .. of wp[Kpad][Cpad]; one chunk ahead in registers
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    const unsigned char* wpb = (const unsigned char*)p.wp;
    const unsigned wp16 = (unsigned)((tid & 7) * 16);
    const unsigned wlds0 = (unsigned)((tid >> 3) * PITCH) + wp16;
    unsigned wrowoff[NWP];
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
        const int wrow = (tid >> 3) + 32 * i;
        wrowoff[i] = (unsigned)(m0 + wrow < p.K ? m0 + wrow : p.K - 1) * (unsigned)(p.Cpad * 2) + wp16;
    }
    u32x4_t wr[NWP];
#pragma unroll
    for (int i = 0; i < NWP; ++i) wr[i] = *(const u32x4_t*)(wpb + wrowoff[i]);
    __syncthreads();                                           // jtab and the patch are in place

    for (int ch = 0; ch < nch; ++ch) {
        unsigned char* wcur = wt + (ch & 1) * WBYTES;
#pragma unroll
        for (int i = 0; i < NWP; ++i) *(u32x4_t*)(wcur + wlds0 + i * 32 * PITCH) = wr[i];
        {
            const int chn = ch + 1 < nch ? ch + 1 : ch;
#pragma unroll
            for (int i = 0; i < NWP; ++i) wr[i] = *(const u32x4_t*)(wpb + wrowoff[i] + (unsigned)chn * 128u);
        }
        // im2col chunk: columns 64 ch + 2 (wv + 4 i) + {0, 1}
        int jt[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) jt[k] = jtab[ch * 64 + 2 * (wv + 4 * (k >> 1)) + (k & 1)];
        if (!(p.dbg & 1))
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            unsigned raw[16];
#pragma unroll
            for (int c2 = 0; c2 < 16; ++c2) raw[c2] = pb[pixl[k] + (jt[c2] >= 0 ? jt[c2] : 0)];
            unsigned char* row = bt + (size_t)(lane + 64 * k) * PITCH + wv * 4;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                *(unsigned*)(row + i * 16) = (jt[2 * i] >= 0 ? raw[2 * i] : 0u) | ((jt[2 * i + 1] >= 0 ? raw[2 * i + 1] : 0u) << 16);
        }
        __syncthreads();
        if (!(p.dbg & 2)) {
            const unsigned char* arow = wcur + (wm * 32 + l31) * PITCH;
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const bf16x8_t a = *(const bf16x8_t*)(arow + kk * 32 + lhi * 16);
#pragma unroll
                for (int ni = 0; ni < WN; ++ni) {
                    const bf16x8_t b = *(const bf16x8_t*)(bt + (size_t)((wn * WN + ni) * 32 + l31) * PITCH + kk * 32 + lhi * 16);
                    acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[ni], 0, 0, 0);
                }
            }
        }
        __syncthreads();                                       // every wave is done with bt before the next gather
    }
    const int mbase = m0 + wm * 32;
    if (p.dbg & 32) { if (acc[0][0] == 12345.678f) ((float*)p.out)[0] = acc[1][1]; return; }
    if (p.epi_wide) {
        gc_epilogue_wide<1, WN, -1>(p, ph, acc[0], acc[1], acc[0], acc[1], mbase, lane, wn, u0, v0, n0,
                                    smem + (size_t)wave * 32 * (WN * 64 + 16));
        return;
    }
    gc_epilogue<false, 1, WN, -1>(p, ph, acc[0], acc[1], acc[0], acc[1], mbase, lhi, pu, pv, pn, pvalid);
}

// Few-channel forward convolutions on the virtual-column kernel (gconv_vc_kernel).  HIFIC_ERR_UNSUPPORTED: not this layer.
int gc_launch_vc(GcParams& p, const float* w, const float* w_scale, long long sm, long long sc,
                           long long sr, long long ss, WsAlloc& ws, hipStream_t st) {
    GcPhase& ph = p.ph[0];
    const int T = ph.ntaps;
    // Measured (round 4, batch 16-32 x 256^2): AlexNet conv1 (C 3, 121 taps, stride 4) 245 -> 50 us.  NOT taken for the 9- and
    // 15-channel layers on 256 x 256 planes (first exact Encoder layer 288 -> 306 us, Discriminator conv1 121 -> 154 us): with
    // 4 000-8 000 tiles their time is the per-tile staging of the raw patch (73-93 us), the gather (21-69 us) and the per-tile
    // base cost, which the saved MFMA slices do not pay for (HIFIC_VC_MAXC widens the rule for experiments).
    if (p.nphase != 1 || p.ost != 1 || p.C > env_int("HIFIC_VC_MAXC", 4) || p.C > 16 || p.C * T < 96 || T < env_int("HIFIC_VC_MINTAPS", 64) ||
        p.K <= 4 || p.K > 256 || p.csplit || p.msplit || p.rfx || p.fold_h || p.resid || p.split || ph.tap0 != 0 || env_int("HIFIC_NO_VC", 0))
        return HIFIC_ERR_UNSUPPORTED;
    // the virtual column order c * T + t is the weight tensor's own order: taps must be (r, s)-major and contiguous
    if (!(ss == 1 && sr > 0 && sc == (long long)T)) return HIFIC_ERR_UNSUPPORTED;
    for (int t = 0; t < T; ++t) if (p.tap_r[t] * (int)sr + p.tap_s[t] != t) return HIFIC_ERR_UNSUPPORTED;
    const int J = p.C * T;
    p.Kpad = cdiv(p.K, 64) * 64;
    p.Cpad = cdiv(J, 64) * 64;
    p.dbg = env_int("HIFIC_DBG", 0); p.afrag = 0; p.ksplit = 1; p.kchunks = 0; p.kpart = nullptr; p.wstage = 0;
    // pixel tile: 2 x 64 on wide planes (whole 128-byte lines of the output), 8 x 16 otherwise
    p.TW = ph.OWt < 64 ? (ph.OWt < 16 ? ph.OWt : 16) : 64;
    p.TH = GC_NPIX / p.TW; if (p.TH > ph.OHt) p.TH = ph.OHt;
    p.NI = 1;
    const int sy = ph.PH, sx = ph.PW;                          // tap spans (finish_phase)
    ph.PH = (p.TH - 1) * p.ist + sy; ph.PW = (p.TW - 1) * p.ist + sx; ph.PWs = ph.PW;
    ph.tiles_y = cdiv(ph.OHt, p.TH); ph.tiles_x = cdiv(ph.OWt, p.TW); p.tiles_n = cdiv(p.N, p.NI);
    ph.wp_off = 0;
    p.max_tiles = p.tiles_n * ph.tiles_y * ph.tiles_x;
    const size_t patch_b = (((size_t)p.NI * p.C * ph.PH * ph.PW * 2) + 15) & ~(size_t)15;
    size_t lds = (((size_t)p.Cpad * 4 + 15) & ~(size_t)15) + 2 * (size_t)64 * 144 + (size_t)GC_NPIX * 144 + patch_b;
    if (lds > (size_t)76 * 1024) { ph.PH = sy; ph.PW = sx; return HIFIC_ERR_UNSUPPORTED; }
    p.epi_wide = 0;
    if (!p.out_f32 && p.TW % 8 == 0 && p.OWf % 8 == 0 && ph.OWt % 8 == 0 && ph.ooy == 0 && ph.oox == 0 &&
        !env_int("HIFIC_NO_WIDE_EPI", 0)) {
        p.epi_wide = 1;
        const size_t need = (size_t)4 * 32 * (2 * 64 + 16);
        if (need > lds) lds = need;
    }
    // the packed operand = the weight matrix [K][J] in bf16, rows padded to Cpad: the 1x1 pack plan over J "channels"
    const size_t wp_bytes = (size_t)p.Kpad * p.Cpad * sizeof(bf16_t);
    PackJob job; memset(&job, 0, sizeof(job));
    GcParams& q = job.p;
    q.K = p.K; q.C = J; q.Kpad = p.Kpad; q.Cpad = p.Cpad; q.nphase = 1; q.tap_sw = 1;
    q.ph[0].ntaps = 1; q.ph[0].tap0 = 0; q.ph[0].wp_off = 0;
    job.sm = sm; job.sc = 1; job.sr = 1; job.ss = 1; job.RS = 1; job.dtype = HIFIC_BF16; job.wp_bytes = (long long)wp_bytes;
    job.mode = 0; job.MB = 16; job.gx = p.Cpad / 64; job.gy = cdiv(p.Kpad, job.MB);
    job.lds_bytes = (int)((size_t)64 * ((job.MB * 1) | 1) * sizeof(float));
    if (ws.plan_out) { *ws.plan_out = job; return HIFIC_OK; }
    void* wp;
    if (ws.wcache_state != 0) {
        if (!ws.wcache || ws.wcache_bytes < wp_bytes) return HIFIC_ERR_WS;
        wp = ws.wcache;
    } else {
        wp = ws.take(wp_bytes);
        if (!wp) return HIFIC_ERR_WS;
    }
    p.wp = wp;
    if (ws.wcache_state != 2) {
        q.wp = wp;
        gc_pack_launch_bf16(job, w, w_scale, st);
    }
    dim3 grid(p.max_tiles * (p.Kpad / 64), 1, 1);
    char ptag[112];
    snprintf(ptag, sizeof(ptag), "gconv_vc K%d C%d N%d in%dx%d out%dx%d taps%d ist%d tile%dx%dx%d J%d grid%d", p.K, p.C, p.N, p.IH,
             p.IW, p.OHf, p.OWf, T, p.ist, p.NI, p.TH, p.TW, J, (int)grid.x);
    const int pslot = gc_prof_open("gconv_vc_kernel", p.aflops, st, ptag);
    gc_prof_bytes(pslot, gc_algo_bytes(p));
    if (p.in_f32) {
        if (lds > 48 * 1024) gc_set_max_lds((const void*)gconv_vc_kernel<true>, (int)lds);
        hipLaunchKernelGGL(gconv_vc_kernel<true>, grid, dim3(256), lds, st, p);
    } else {
        if (lds > 48 * 1024) gc_set_max_lds((const void*)gconv_vc_kernel<false>, (int)lds);
        hipLaunchKernelGGL(gconv_vc_kernel<false>, grid, dim3(256), lds, st, p);
    }
    gc_prof_close(pslot, st);
    return hific_launch_status();
}


// gconv_mp_kernel launch for the plan of gconv.hip's launch_gconv_tb (grid.z = 2: output-row parity)
void gc_launch_mp(const GcParams& p, dim3 grid, size_t lds, hipStream_t st) {
    if (p.split) {
        if (lds > 48 * 1024) gc_set_max_lds((const void*)gconv_mp_kernel<true>, (int)lds);
        hipLaunchKernelGGL(gconv_mp_kernel<true>, grid, dim3(256), lds, st, p);
    } else {
        if (lds > 48 * 1024) gc_set_max_lds((const void*)gconv_mp_kernel<false>, (int)lds);
        hipLaunchKernelGGL(gconv_mp_kernel<false>, grid, dim3(256), lds, st, p);
    }
}
