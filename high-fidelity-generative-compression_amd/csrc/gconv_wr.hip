// Weight-resident persistent kernel for the FEW-CHANNEL layers on big planes (gfx950): the first Encoder convolution in its
// split-operand form (60 <- 9 channels, 7x7, 256 x 256), the virtual-row / virtual-channel forms of the Generator's 60 -> 3 output
// convolution and its data gradient (21 <- 60 and 60 <- 21 channels, 7 taps), the Discriminator's first convolution
// (reference: src/network/encoder.py:56-62, src/network/generator.py:139-142, src/network/discriminator.py:53).
//
// Why.  These layers have ONE channel chunk, <= 64 output rows and 4 000-8 000 pixel tiles.  gconv_kernel gives every tile its
// own workgroup: launch, weight-tile ring from L2 (the same 30-100 KB for every tile), synchronous patch staging, MFMAs,
// epilogue - all serial, 3-15 % of the HBM rate.  Here
//   * a workgroup is PERSISTENT (one per CU) and keeps the whole packed weight tensor in LDS, written once in MFMA A-fragment
//     order ([tap][16-channel slice][32-row block][lane][16 B]);
//   * the 8 waves have two ROLES: waves 4-7 stage the halo patch of tile i+1 (stage_T / stage_W: the loaders of gconv_kernel)
//     into the other patch buffer while waves 0-3 run the MFMAs of tile i (64 pixels x all rows per wave, fragment reads one
//     step ahead of the MFMAs) - one barrier per tile, and the stores of tile i are issued after it, when the loaders are
//     already on tile i+2;
//   * 256-pixel tiles (8 x 32, 4 x 64 or 16 x 16): the halo is re-read less often than with the 128-pixel tiles.
// The epilogue is gc_store_block (bias, output scale, activation, residual, reflect fold, any output stride).
#include "gconv.h"
#include "gconv_dev.h"
#include "gconv_stage.h"
#include <stdio.h>
#include <string.h>

#define WR_NPIX 256

// NLW = loader waves (4 or 8): the 64-channel layers stage 50-60 KB per tile with two-byte loads and are bound by the loads one
// wave keeps in flight; eight loader waves need the 3-waves-per-SIMD register budget (WM = 1 only).
template <int BC, int WM, int NLW>
__global__ __launch_bounds__(256 + 64 * NLW) __attribute__((amdgpu_waves_per_eu(NLW == 8 ? 3 : 2, NLW == 8 ? 3 : 2)))
void gconv_wr_kernel(const GcParams p) {
    static_assert(NLW == 4 || NLW == 8, "loader waves");
    constexpr int NTHR = 256 + 64 * NLW;
    constexpr int KK = BC / 16;                    // 16-deep reduction slices per tap
    constexpr int PITCH = BC * 2 + 16;             // bytes of one patch pixel row (BC channels + pad: conflict-free 16-byte reads)
    constexpr int DWR = BC / 2;
    constexpr int TU = KK == 1 ? 2 : 1;            // taps per trip of the tap loop (the fragment register sets alternate per slice)
    constexpr int QB = BC == 16 ? 8 : (BC == 32 ? 6 : 4);
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const bool loader = threadIdx.x >= 256;
    const int tid = loader ? (int)threadIdx.x - 256 : (int)threadIdx.x;      // role-local
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;

    const GcPhase& ph = p.ph[0];
    const int nt = ph.ntaps;
    const int nt2 = (nt + 1) & ~1;                 // (an odd tap count gets one all-zero tap)
    const int PH = ph.PH, PW = ph.PW;
    const int npatch = PH * PW;
    const unsigned patch_bytes = (unsigned)(((size_t)(npatch + 1) * PITCH + 15) & ~(size_t)15);       // + the dump row of stage_W
    int* toffs = (int*)smem;                                   // [GC_MAXTAPS] byte offset of each tap inside the patch
    float* bias_l = (float*)(smem + 512);                      // [64] bias of the row tile (wide-store epilogue)
    unsigned char* wl = smem + 1024;                           // nt2 x KK x WM x 1 KB
    unsigned char* pbuf = wl + (size_t)nt2 * KK * WM * 1024;   // 2 x patch_bytes
    unsigned char* epi_l = pbuf + 2 * (size_t)patch_bytes;     // p.epi_wide: 4 x 2 KB, the compute waves' store-transposition regions

    // ---- this workgroup's tiles [t_lo, t_hi): contiguous, and neighbouring ranges on the same XCD (their halos share that L2)
    int t_lo, t_hi;
    {
        const int nwg = gridDim.x;
        const int q8 = nwg >> 3, r8 = nwg & 7;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int q = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
        t_lo = q * p.pl_tpw;
        t_hi = t_lo + p.pl_tpw < p.max_tiles ? t_lo + p.pl_tpw : p.max_tiles;
        if (t_lo >= t_hi) return;
    }
    const int tiles_xy = ph.tiles_x * ph.tiles_y;

    // ---- resident weights + tap table (all threads)
    if (threadIdx.x < GC_MAXTAPS) {
        const int t = threadIdx.x;
        toffs[t] = t < nt ? (((int)p.tap_dy[ph.tap0 + t] - ph.dy_min) * PW + ((int)p.tap_dx[ph.tap0 + t] - ph.dx_min)) * PITCH : 0;
    }
    if (threadIdx.x >= 256 && threadIdx.x < 320) {
        const int m = threadIdx.x - 256;
        bias_l[m] = (p.bias && m < p.K) ? p.bias[m] : 0.f;
    }
    {
        const bf16_t* wp = (const bf16_t*)p.wp + ph.wp_off;
        const int nfr = nt2 * KK * WM * 64;                    // 16-byte fragment pieces
        for (int i = threadIdx.x; i < nfr; i += NTHR) {
            const int ln = i & 63;
            int r = i >> 6;
            const int mi = r % WM; r /= WM;
            const int kk = r % KK;
            const int t = r / KK;
            int m = mi * 32 + (ln & 31);
            if (m >= p.K) m = p.K - 1;                         // padded rows repeat the last row (never stored)
            const int c = kk * 16 + (ln >> 5) * 8;
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (t < nt) v = *(const u32x4_t*)(wp + ((size_t)m * nt + t) * p.Cpad + c);
            *(u32x4_t*)(wl + (size_t)i * 16) = v;
        }
    }

#define WR_TILE_ORIGIN(t_, n_, u0_, v0_)                                        \
    const int n_ = (t_) / tiles_xy;                                             \
    const int u0_ = (((t_) - n_ * tiles_xy) / ph.tiles_x) * p.TH;               \
    const int v0_ = (((t_) - n_ * tiles_xy) % ph.tiles_x) * p.TW;

    if (loader) {
        // ===================== loader role: the patch of the NEXT tile into the other buffer =====================
#define WR_STAGE(t_, buf_)                                                                                              \
    do {                                                                                                                \
        WR_TILE_ORIGIN(t_, n_, u0_, v0_)                                                                                \
        unsigned char* d_ = pbuf + (size_t)(buf_) * patch_bytes;                                                        \
        const int iy0_ = u0_ * p.ist + ph.dy_min, ix0_ = v0_ * p.ist + ph.dx_min;                                       \
        if constexpr (BC >= 32 && NLW == 4) {                                                                           \
            if (p.wstage) stage_W<PITCH, 6, BC / 2>(d_, (const bf16_t*)p.in, p.N, p.C, p.IH, p.IW, p.bmode, n_, 1,      \
                                                    iy0_, ix0_, PH, PW, PW, 0, tid, npatch);                            \
            else stage_T<bf16_t, DWR, PITCH, QB>(d_, p.in, p.in_f32, p.N, p.C, p.IH, p.IW, p.bmode, n_, 1, iy0_, ix0_,  \
                                                 PW, PH, PW, 0, tid, 256);                                              \
        } else {                                                                                                        \
            stage_T<bf16_t, DWR, PITCH, QB, NLW == 8>(d_, p.in, p.in_f32, p.N, p.C, p.IH, p.IW, p.bmode, n_, 1, iy0_,   \
                                                      ix0_, PW, PH, PW, 0, tid, 64 * NLW);                              \
        }                                                                                                               \
    } while (0)
        WR_STAGE(t_lo, 0);
        __syncthreads();
        for (int t = t_lo; t < t_hi; ++t) {
            if (t + 1 < t_hi && !(p.dbg & 1)) WR_STAGE(t + 1, (t + 1 - t_lo) & 1);      // (dbg: timing ablations, HIFIC_DBG)
            __syncthreads();
        }
#undef WR_STAGE
        return;
    }

    // ===================== compute role: 4 waves x 64 pixels x all rows =====================
    int ty_[2], tx_[2];
    unsigned qb[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int pt = (wave * 2 + ni) * 32 + l31;
        ty_[ni] = pt / p.TW; tx_[ni] = pt - ty_[ni] * p.TW;
        qb[ni] = (unsigned)((ty_[ni] * p.ist * PW + tx_[ni] * p.ist) * PITCH + lhi * 16);
    }
    f32x16_t acc[WM][2];
#pragma unroll
    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // fragments of step (tap tt_, slice kk_) -> register set S_; `toff_` = that tap's patch offset
#define WR_FRAGS(S_, tt_, kk_, toff_)                                                                               \
    do {                                                                                                            \
        const unsigned char* a_ = wl + ((size_t)((tt_) * KK + (kk_)) * WM) * 1024 + lane * 16;                      \
        _Pragma("unroll") for (int mi = 0; mi < WM; ++mi) fa[S_][mi] = *(const bf16x8_t*)(a_ + mi * 1024);          \
        _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) fb[S_][ni] = *(const bf16x8_t*)(pc + qb[ni] + (toff_) + (kk_) * 32); \
    } while (0)

    __syncthreads();                                           // weights, tap table, first patch
    for (int t = t_lo; t < t_hi; ++t) {
        const unsigned char* pc = pbuf + (size_t)((t - t_lo) & 1) * patch_bytes;
        bf16x8_t fa[2][WM], fb[2][2];
        // tap offsets are read from the LDS table TWO taps ahead: the fragment reads of the next step never wait for one
        unsigned tcur = (unsigned)toffs[0], tnx = (unsigned)toffs[1];
        WR_FRAGS(0, 0, 0, tcur);
        for (int tt0 = 0; tt0 < ((p.dbg & 2) ? 0 : nt2); tt0 += TU) {
#pragma unroll
            for (int u = 0; u < TU; ++u) {
                const int tt = tt0 + u;
                const int tn = tt + 1 < nt2 ? tt + 1 : 0;      // (the last tap prefetches tap 0 again: never consumed)
                const unsigned tnn = (unsigned)toffs[tt + 2 < nt2 ? tt + 2 : 0];
#pragma unroll
                for (int kk = 0; kk < KK; ++kk) {
                    const int cur = (u * KK + kk) & 1;
                    if (kk + 1 < KK) WR_FRAGS(cur ^ 1, tt, kk + 1, tcur);
                    else WR_FRAGS(cur ^ 1, tn, 0, tnx);
                    __builtin_amdgcn_sched_barrier(0);         // (keeps the reads of the NEXT step ahead of this step's MFMAs)
#pragma unroll
                    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[cur][mi], fb[cur][ni], acc[mi][ni], 0, 0, 0);
                }
                tcur = tnx; tnx = tnn;
            }
        }
        __syncthreads();                                       // the loaders move on to tile t + 2; this buffer is theirs again
        if (p.dbg & 32) {
            if (acc[0][0][0] == 12345.678f) ((float*)p.out)[0] = acc[WM - 1][1][1];
        } else {
            WR_TILE_ORIGIN(t, n_, u0_, v0_)
            // (opaque row base: with a loop-invariant one hipcc hoists the epilogue's address arithmetic out of the tile loop
            //  and carries it through the MFMA loop in scratch - seen on gconv_pl_kernel)
            int mb = 0;
            asm volatile("" : "+v"(mb));
            if (p.epi_wide) {
                // LDS-transposed 16-byte stores (pl_store_wide): legality checked by the plan
                unsigned char* wreg = epi_l + wave * 2048;
                const int tws = p.TW == 64 ? 6 : (p.TW == 32 ? 5 : 4);
                const float osc = p.oscale ? *p.oscale : 1.f;
                const float slope = p.act == ACT_RELU ? 0.f : (p.act == ACT_LEAKY ? 0.2f : 1.f);
#pragma unroll
                for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        if (p.out_f32) pl_store_wide<true>(p, ph, acc[mi][ni], mi, 0, mb, lane, wave * 2 + ni, u0_, v0_, n_, tws, wreg, 1024, bias_l, osc, slope);
                        else pl_store_wide<false>(p, ph, acc[mi][ni], mi, 0, mb, lane, wave * 2 + ni, u0_, v0_, n_, tws, wreg, 1024, bias_l, osc, slope);
                    }
            } else {
                int pu[2], pv[2], pn[2];
                bool pvalid[2];
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) { pu[ni] = u0_ + ty_[ni]; pv[ni] = v0_ + tx_[ni]; pn[ni] = n_; pvalid[ni] = true; }
                gc_epilogue<false, WM, 2, -1>(p, ph, acc[0][0], acc[0][1], acc[WM - 1][0], acc[WM - 1][1], mb, lhi, pu, pv, pn, pvalid);
            }
        }
#pragma unroll
        for (int mi = 0; mi < WM; ++mi)
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
    }
#undef WR_FRAGS
#undef WR_TILE_ORIGIN
}

// ---------------------------------------------------------------------------------------------------------------------------
// Host side: does the plan qualify, tile shape, weight packing (the [row][tap][channel] image of gconv_kernel), launch.
// HIFIC_ERR_UNSUPPORTED: not this layer (the plan is left for the caller to restore).
// ---------------------------------------------------------------------------------------------------------------------------
int launch_gconv_wr(GcParams& p, const float* w, const float* w_scale, long long sm, long long sc, long long sr, long long ss,
                    WsAlloc& ws, hipStream_t st) {
    if (!gc_env_int("HIFIC_WR", 1)) return HIFIC_ERR_UNSUPPORTED;
    if (p.nphase != 1 || p.rfx || p.split || p.K > 64 || p.C > 64 || p.ist < 1 || p.ist > 2) return HIFIC_ERR_UNSUPPORTED;
    // reflect-fold data gradients (interior pixels to dx, rim to the plane buffer) keep the generic kernel: their stores cannot take
    // the 16-byte path (the interior starts `pad` pixels into the padded row) and the 2-byte form measured 173 -> 181 us on the
    // 60 <- 21-channel virtual-channel gradient of the output convolution
    if (p.fold_h && !gc_env_int("HIFIC_WR_FOLD", 0)) return HIFIC_ERR_UNSUPPORTED;
    if (p.ist == 2 && !gc_env_int("HIFIC_WR_S2", 1)) return HIFIC_ERR_UNSUPPORTED;
    GcPhase& ph = p.ph[0];
    const int nt = ph.ntaps, nt2 = (nt + 1) & ~1;
    if (nt < 1 || nt2 > GC_MAXTAPS) return HIFIC_ERR_UNSUPPORTED;
    if ((long long)p.N * p.C * p.IH * p.IW >= (1ll << 31)) return HIFIC_ERR_UNSUPPORTED;          // 32-bit element offsets (stage_T)
    const int BC = p.C <= 16 ? 16 : (p.C <= 32 ? 32 : 64);
    const int WM = p.K <= 32 ? 1 : 2;
    const int PITCH = BC * 2 + 16;
    const size_t wbytes = 1024 + (size_t)nt2 * (BC / 16) * WM * 1024;
    // wide-store epilogue (pl_store_wide): output stride 1 on the (u, v) domain itself, 16-byte aligned rows, pieces of 8 bf16 / 4
    // f32 pixels never straddle the domain edge, no fold / residual; + 4 x 2 KB of LDS
    const int pe = p.out_f32 ? 4 : 8;
    const bool wide_ok = p.ost == 1 && ph.ooy == 0 && ph.oox == 0 && !p.fold_h && !p.resid && p.OWf % pe == 0 && ph.OWt % pe == 0 &&
                         !gc_env_int("HIFIC_NO_WIDE_EPI", 0);
    // tile: 256 pixels of one image; the shape with the least staged pixels over the whole domain that fits next to the weights
    // (first with the store regions of the wide epilogue, then without)
    const int span_y = ph.PH, span_x = ph.PW;
    static const int shapes[3][2] = {{8, 32}, {4, 64}, {16, 16}};
    int TH = 0, TW = 0;
    bool wide = false;
    for (int pass = wide_ok ? 0 : 1; pass < 2 && !TH; ++pass) {
        double best = 1e300;
        for (int i = 0; i < 3; ++i) {
            const int th = shapes[i][0], tw = shapes[i][1];
            if (tw > ph.OWt && i != 2) continue;
            const long long phh = (long long)(th - 1) * p.ist + span_y, pww = (long long)(tw - 1) * p.ist + span_x;
            const size_t pb = (((size_t)(phh * pww + 1) * PITCH) + 15) & ~(size_t)15;
            if (wbytes + 2 * pb + (pass == 0 ? 8192 : 0) > (size_t)160 * 1024) continue;
            const double cost = (double)cdiv(ph.OHt, th) * cdiv(ph.OWt, tw) * (double)phh * ((double)pww + 16.0);
            if (cost < best) { best = cost; TH = th; TW = tw; wide = pass == 0; }
        }
    }
    if (!TH) return HIFIC_ERR_UNSUPPORTED;
    const int tiles_y = cdiv(ph.OHt, TH), tiles_x = cdiv(ph.OWt, TW);
    const long long ntile = (long long)p.N * tiles_y * tiles_x;
    if (ntile < gc_env_int("HIFIC_WR_MIN_TILES", 1024) || ntile >= (1ll << 30)) return HIFIC_ERR_UNSUPPORTED;

    p.TH = TH; p.TW = TW; p.NI = 1; p.tiles_n = p.N;
    p.Kpad = WM * 32;
    p.Cpad = BC;
    p.dbg = gc_env_int("HIFIC_DBG", 0);
    p.tap_sw = (int)sr;
    p.afrag = 0; p.ksplit = 1; p.kchunks = 0; p.kpart = nullptr; p.kpart_stride = 0; p.epi_wide = 0;
    ph.PH = (TH - 1) * p.ist + span_y; ph.PW = (TW - 1) * p.ist + span_x; ph.PWs = ph.PW;
    ph.tiles_y = tiles_y; ph.tiles_x = tiles_x;
    ph.wp_off = 0;
    const long long wp_elems = (long long)p.Kpad * nt * p.Cpad;
    const size_t pb = (((size_t)(ph.PH * ph.PW + 1) * PITCH) + 15) & ~(size_t)15;
    size_t lds = wbytes + 2 * pb;
    if (wide && ((size_t)p.out & 15) == 0) { p.epi_wide = 1; lds += 8192; }
    // wide-load staging (16-byte loads of aligned 8-pixel groups): bf16 source with 16-byte aligned rows.  HIFIC_WR_WSTAGE:
    // 0 = never, 1 = the stride-2 layers (default), 2 = whenever legal
    p.wstage = 0;
    {
        const int wst = gc_env_int("HIFIC_WR_WSTAGE", 1);
        if (BC >= 32 && !p.in_f32 && p.IW % 8 == 0 && p.IW >= 32 && ((size_t)p.in & 15) == 0 && (wst == 2 || (wst == 1 && p.ist == 2)))
            p.wstage = 1;
    }
    const int ncu = 256;
    int grid = ntile < ncu ? (int)ntile : ncu;
    p.pl_tpw = (int)cdivl(ntile, grid);
    grid = (int)cdivl(ntile, p.pl_tpw);
    p.max_tiles = (int)ntile;

    bool plan_only = false;
    const int rcp = gc_pack_weights_bf16(p, wp_elems, w, w_scale, sm, sc, sr, ss, ws, st, &plan_only);
    if (rcp != HIFIC_OK || plan_only) return rcp;

    const int nlw = (BC == 64 && WM == 1 && !p.wstage && gc_env_int("HIFIC_WR_LW8", 1)) ? 8 : 4;
    char ptag[112], kname[64];
    snprintf(ptag, sizeof(ptag), "gconv_wr K%d C%d N%d in%dx%d out%dx%d taps%d ist%d ost%d tile%dx%d tpw%d grid%d%s%s", p.K, p.C, p.N,
             p.IH, p.IW, p.OHf, p.OWf, nt, p.ist, p.ost, TH, TW, p.pl_tpw, grid, p.wstage ? " wload" : "",
             p.epi_wide ? " wstore" : "");
    snprintf(kname, sizeof(kname), "gconv_wr_kernel<%d,%d%s>", BC, WM, nlw == 8 ? ",lw8" : "");
    const int pslot = gc_prof_open(kname, p.aflops, st, ptag);
    gc_prof_bytes(pslot, gc_algo_bytes(p));
#define WR_LAUNCH(BC_, WM_, NLW_)                                                                     \
    do {                                                                                              \
        gc_set_max_lds((const void*)gconv_wr_kernel<BC_, WM_, NLW_>, (int)lds);                       \
        hipLaunchKernelGGL((gconv_wr_kernel<BC_, WM_, NLW_>), dim3(grid), dim3(256 + 64 * NLW_), lds, st, p); \
    } while (0)
    if (BC == 16) { if (WM == 1) WR_LAUNCH(16, 1, 4); else WR_LAUNCH(16, 2, 4); }
    else if (BC == 32) { if (WM == 1) WR_LAUNCH(32, 1, 4); else WR_LAUNCH(32, 2, 4); }
    else if (WM == 2) WR_LAUNCH(64, 2, 4);
    else if (nlw == 8) WR_LAUNCH(64, 1, 8);
    else WR_LAUNCH(64, 1, 4);
#undef WR_LAUNCH
    gc_prof_close(pslot, st);
    return hific_launch_status();
}
