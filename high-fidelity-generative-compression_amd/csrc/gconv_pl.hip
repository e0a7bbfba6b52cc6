// Pipelined, persistent, wave-specialised forward-type kernel for the STRIDE-2 layers (gfx950): Encoder blocks 2-5 (plain and
// native split-bf16 form), the Discriminator's 4x4 convolutions, and the data gradients of the Generator's up-convolutions - every
// one of them a one-phase tap-table contraction with input stride 2 (reference: src/network/encoder.py:64-93,
// src/network/discriminator.py:53-62, src/network/generator.py:115-137 backward).
//
// Why a second kernel.  gconv_kernel runs stage -> barrier -> MFMA -> barrier per channel chunk with nothing in flight across a
// barrier: on these layers (four input pixels per output pixel, 2000-4000 tiles) every workgroup starts its staging at the same
// time, HBM is saturated for the staging phase and idle for the rest (counters: VALU 49 %, LDS 26 %, MFMA 13 %).  Here
//   * a workgroup is PERSISTENT: it walks a contiguous range of pixel tiles of one 128-row tile and treats (tile, 32-channel
//     chunk) pairs as one stream of items;
//   * the 16 waves have two ROLES.  Waves 8-15 LOAD: the halo patch of item i+1 is requested (16-byte loads, global -> registers)
//     at the start of item i and written to the OTHER patch buffer in the last tap group of item i; the packed weights
//     (GcParams::afrag = 2: 8 KB per tap in MFMA A-fragment order) run in groups of TG taps through a two-slot LDS ring and are
//     requested TWO groups ahead into rotating register sets.  Every load count is static (hipcc's vmcnt stays exact) and
//     vector loads return in order, so the A request goes out first and the patch request second.  Waves 0-7 COMPUTE: 2 row
//     halves x 4 pixel quarters (64 rows x 32 pixels each), fragment reads of tap t+1 under the MFMAs of tap t, and the epilogue
//     of a tile (LDS-transposed 16-byte stores through a wave-private 2 KB region) in the first group of the NEXT tile, when
//     the loaders' requests for that tile are already in flight.  Both role loops execute the same barrier sequence: one
//     barrier per tap group;
//   * a patch load instruction reads ONE channel (wave-uniform) for 64 (patch row, aligned 8-pixel group) units: neighbouring
//     lanes read neighbouring 16 bytes (8-20 cache lines per instruction; with lane = channel pair it was 48-64 and the kernel
//     ran at the texture addresser's line rate);
//   * the patch image in LDS is PARITY-PLANAR in planes of 4 channels: patch pixel (y, x) lives at pixel index
//     ((y & 1, x & 1), y >> 1, x >> 1) of each of the 8 planes, 8 bytes per pixel, so that the B fragment of any tap - 32 pixels
//     two patch pixels apart - is 32 CONSECUTIVE pixels: conflict-free ds_read_b64 x 4 per tap and ds_write_b64 on the way
//     in (v_perm interleaves the two channels of a pair), no padding bytes; the mirror column of a reflect boundary is written
//     from the registers that already hold its source column.
// SPLIT: operands in the pair layout of the exact-index chain (hific_split3 which = 2: slice 0 = hi, slice 1 = lo of the same
// 16 real channels); a tap issues lo*hi + hi*lo + hi*hi per row block, as gconv_kernel<..., SPLIT> does.
// Measured history (combined-role 512-thread form, transposed-tile form, spread requests, ...): docs/ENGINEERING_LOG.md round 5.
#include "gconv.h"
#include "gconv_dev.h"
#include "gconv_stage.h"
#include <stdio.h>
#include <string.h>

// timing ablations (WRONG RESULTS; tools/r05 only): 1 = no patch loads, 2 = no A loads, 4 = no patch LDS writes, 16 = no
// epilogue, 32 = no A LDS writes, 64 = epilogue without its global stores, 128 = epilogue stores of a dummy register instead
// of the LDS-transposed data
#ifndef PL_ABL
#define PL_ABL 0
#endif
#define PL_QI 4          // patch lane-items per thread and item: (channel pair, unit = (patch row, aligned 8-pixel group)), 2 x 16 bytes
#define PL_SLOT_TAP 8192 // bytes of one tap's A operands: 4 row blocks x 2 slices x 1 KB

template <int NT, int TG, bool SPLIT>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4)))
void gconv_pl_kernel(const GcParams p) {
    static_assert(NT % TG == 0, "taps per group");
    constexpr int NG = NT / TG;
    constexpr int SLOT = TG * PL_SLOT_TAP;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    // 16 waves: 0-7 compute (LDS fragment reads, MFMAs, epilogue), 8-15 load (global requests, LDS writes of both operands).
    // Each role has its own loop with the SAME barrier sequence; `tid` / `wave` below are role-local (0..511 / 0..7).
    const bool loader = threadIdx.x >= 512;
    const int tid = threadIdx.x & 511;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int l31 = lane & 31, lhi = lane >> 5;

    const GcPhase& ph = p.ph[0];
    const int PH = ph.PH, PW = ph.PW;
    const int RH = (PH + 1) >> 1, RW = (PW + 1) >> 1;
    const int npix = 4 * RH * RW;                          // pixels of one patch buffer (+ 1 dump pixel)
    const unsigned plb = (unsigned)(npix + 1) * 8u;        // bytes of one 4-channel plane: [pixel][4 channels]
    const unsigned pbytes = 8u * plb;                      // one patch buffer: 8 planes of 4 channels
    unsigned char* aring = smem;                           // 2 x SLOT
    unsigned char* pbuf = smem + 2 * SLOT;                 // 2 x pbytes
    float* bias_l = (float*)(pbuf + 2 * pbytes);           // 128 floats: bias of this workgroup's row tile
    unsigned char* epi_l = (unsigned char*)(bias_l + 128); // 8 x 2 KB: the compute waves' store-transposition regions

    // ---- work of this workgroup: pixel tiles [t_lo, t_hi) of row tile `mtile` ------------------------------------------
    int mtile, t_lo, t_hi;
    {
        const int nwg = gridDim.x;
        const int q8 = nwg >> 3, r8 = nwg & 7;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int q = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;      // bijective, XCD-contiguous
        const int mtiles = p.Kpad >> 7;
        const int ntile = p.tiles_n * ph.tiles_y * ph.tiles_x;
        const int npg = nwg / mtiles;
        int pg;
        if (p.pl_wshare) {
            // big weight tensors: an XCD works on ONE row tile (its packed weights stay in that L2: the A requests have one tap
            // group of lead) and streams the pixel tiles
            mtile = q / npg; pg = q - mtile * npg;
        } else {
            // neighbours in q (same XCD) share their pixel tiles and differ in the row tile: the patch is served by that L2
            mtile = q % mtiles; pg = q / mtiles;
        }
        t_lo = pg * p.pl_tpw;
        t_hi = t_lo + p.pl_tpw < ntile ? t_lo + p.pl_tpw : ntile;
        if (t_lo >= t_hi) return;
    }
    const int m0 = mtile << 7;
    const int nch = p.Cpad >> 5;
    const unsigned plane = (unsigned)(p.IH * p.IW);
    const bf16_t* inb = (const bf16_t*)p.in;
    const bool refl = p.bmode == PAD_REFLECT;
    const int tiles_xy = ph.tiles_x * ph.tiles_y;
    const bool hb = p.bias != nullptr;
    if (!loader && tid < 128) bias_l[tid] = (hb && m0 + tid < p.K) ? p.bias[m0 + tid] : 0.f;
    const float osc = p.oscale ? *p.oscale : 1.f;
    const float slope = p.act == ACT_RELU ? 0.f : (p.act == ACT_LEAKY ? 0.2f : 1.f);

    // ---- this thread's patch lane-items.  A wave instruction loads ONE channel (wave-uniform) for 64 (patch row, aligned
    //      8-pixel group) units, groups fastest: neighbouring lanes read neighbouring 16 bytes of a row, so an instruction
    //      touches ~8-20 cache lines (with lane = channel pair it was 48-64 lines and the kernel ran at the texture
    //      addresser's one-line-per-clock rate: 30 of 94 us on 480 -> 960 @32x32).  Wave w owns channel pairs 2w, 2w+1;
    //      item j = (pair j >> 1, half j & 1 of the <= 128 units).  Static per thread (the same for every item: (TW * ist)
    //      % 8 == 0, so the position of the patch inside its aligned groups never changes): unit -> (row, group) and the
    //      eight LDS dword offsets of its pixels. -------------------------------------------------------------------------
    const int acol = ((ph.dx_min % 8) + 8) & 7;            // patch column 0 = element `acol` of aligned group 0
    const int NGR = (acol + PW + 7) >> 3;                  // aligned groups per patch row
    const unsigned dump_off = (unsigned)npix * 8u;
    int un_py[2], un_g[2];
    bool un_ok[2];
    unsigned un_dst[2][4];                                 // plane-relative byte offsets of elements (2i, 2i+1), 16 bits each
    {
        const float inv_ngr = 1.0f / (float)NGR;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int u = lane + 64 * h;
            const int py = (int)(((float)u + 0.5f) * inv_ngr);
            const int g = u - py * NGR;
            un_ok[h] = py < PH;
            un_py[h] = py; un_g[h] = g;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int pc = g * 8 + e - acol;
                const bool in = un_ok[h] && pc >= 0 && pc < PW;
                const int pix = (((py & 1) * 2 + (pc & 1)) * RH + (py >> 1)) * RW + (pc >> 1);
                const unsigned d = in ? (unsigned)pix * 8u : dump_off;
                if (e & 1) un_dst[h][e >> 1] |= d << 16; else un_dst[h][e >> 1] = d;        // (the plan keeps a plane < 64 KB)
            }
        }
    }

    // ---- B side: this lane's pixel inside the tile (TH * TW == 128, one image per tile) -------------------------------------
    int lt_y, lt_x;
    unsigned bpix0;
    {
        const int pt = wn * 32 + l31;
        lt_y = pt / p.TW; lt_x = pt - lt_y * p.TW;
        bpix0 = (unsigned)(lt_y * RW + lt_x);
    }
    // patch-pixel displacement of every tap: plane (dy & 1, dx & 1), row / column shift (dy >> 1, dx >> 1); wave-uniform
    unsigned toffs[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int dyp = (int)p.tap_dy[t] - ph.dy_min, dxp = (int)p.tap_dx[t] - ph.dx_min;
        toffs[t] = (unsigned)((((dyp & 1) * 2 + (dxp & 1)) * RH + (dyp >> 1)) * RW + (dxp >> 1));
    }

    f32x16_t acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }

    // A operands are requested TWO tap groups ahead (one group of MFMAs is 0.3-0.6 us, an L2 round trip under load more than that:
    // with one group of lead every group ended in a wait - timing ablation: the loop without any memory instruction 27 us, with
    // the A path alone +12-20 us on 480 -> 960 @32x32).  Register sets rotate with a period that divides the groups per item, so
    // every index stays a compile-time constant; the LDS ring keeps its two slots.
    constexpr int NSET = (NG % 2 == 0) ? 2 : 3;
    static_assert(NG % NSET == 0 && NG >= 2, "register sets per item");
    pl_u32x4_t aN[NSET][TG] = {};
    pl_u32x4_t pva[PL_QI] = {}, pvb[PL_QI] = {};           // item j: channels 2 cp, 2 cp + 1 of unit (j & 1), cp = 2 wave + (j >> 1)
    unsigned pmask[2];                                     // per unit half: all ones / zero (row, image column range)
    bool pwr[2];
    unsigned poff[2];                                      // per unit half: element offset of its group in channel 0 (0 when masked)
    int prl[2], prr[2];                                    // reflect rim: patch column fed by element 1 / 6 of this unit, or -1

    const unsigned char* wp_m = (const unsigned char*)p.wp + (size_t)mtile * nch * NT * PL_SLOT_TAP + (size_t)tid * 16;

    // A operands of tap group g of chunk c -> registers
#define PL_ISSUE_A(set_, c_, g_)                                                                                   \
    do {                                                                                                           \
        const unsigned char* s_ = wp_m + ((size_t)(c_) * NT + (g_) * TG) * PL_SLOT_TAP;                            \
        if constexpr (!(PL_ABL & 2)) {                                                                             \
        _Pragma("unroll") for (int j = 0; j < TG; ++j) aN[set_][j] = *(const pl_u32x4_t*)(s_ + (size_t)j * PL_SLOT_TAP); } \
    } while (0)
#define PL_WRITE_A(set_, slot_)                                                                                    \
    do {                                                                                                           \
        unsigned char* d_ = aring + (slot_) * SLOT + tid * 16;                                                     \
        if constexpr (!(PL_ABL & 32)) {                                                                            \
        _Pragma("unroll") for (int j = 0; j < TG; ++j) *(pl_u32x4_t*)(d_ + j * PL_SLOT_TAP) = aN[set_][j]; }       \
    } while (0)
    // halo patch of (tile t_, chunk c_) -> registers: unconditional loads from clamped addresses, masks applied at the LDS write
#define PL_P_ADDR(t_)                                                                                              \
    do {                                                                                                           \
        const int tn_ = (t_) / tiles_xy, trem_ = (t_) - tn_ * tiles_xy;                                            \
        const int ty_ = trem_ / ph.tiles_x, tx_ = trem_ - ty_ * ph.tiles_x;                                        \
        const int iy0_ = ty_ * p.TH * p.ist + ph.dy_min, ix0_ = tx_ * p.TW * p.ist + ph.dx_min;                    \
        const int gx0_ = ix0_ - acol;                                                                              \
        const unsigned nbase_ = (unsigned)tn_ * (unsigned)p.C * plane;                                             \
        _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                            \
            int iy_ = iy0_ + un_py[h];                                                                             \
            if (refl) iy_ = reflect_idx(iy_, p.IH);                                                                \
            const int gx_ = gx0_ + un_g[h] * 8;                                                                    \
            const bool col_in_ = gx_ >= 0 && gx_ + 8 <= p.IW;                                                      \
            const bool ok_ = un_ok[h] && (unsigned)iy_ < (unsigned)p.IH && col_in_;                                \
            poff[h] = ok_ ? (nbase_ + (unsigned)iy_ * (unsigned)p.IW + (unsigned)gx_) : 0u;                        \
            pmask[h] = ok_ ? 0xffffffffu : 0u;                                                                     \
            /* reflect mode: out-of-image groups are filled from their mirror columns (rim, below), not with zeros */ \
            pwr[h] = !(refl && !col_in_);                                                                          \
            /* the image's first / last group also feeds the patch column of image column -1 / W (mirror of column 1 / W-2) */ \
            const int pl_ = -1 - ix0_, pr_ = p.IW - ix0_;                                                          \
            prl[h] = (refl && ok_ && gx_ == 0 && pl_ >= 0 && pl_ < PW) ? pl_ : -1;                                 \
            prr[h] = (refl && ok_ && gx_ == p.IW - 8 && pr_ >= 0 && pr_ < PW) ? pr_ : -1;                          \
        }                                                                                                          \
    } while (0)
    // the two 16-byte loads of item j (channel pair 2 wave + (j >> 1) of chunk c_, unit j & 1)
#define PL_P_LOAD(j, c_)                                                                                           \
    do {                                                                                                           \
        const int ca_ = (c_) * 32 + 2 * (2 * wave + ((j) >> 1));                                                   \
        const unsigned offa_ = (ca_ < p.C ? (unsigned)ca_ : 0u) * plane;                                           \
        const unsigned offb_ = (ca_ + 1 < p.C ? (unsigned)(ca_ + 1) : 0u) * plane;                                 \
        if constexpr (!(PL_ABL & 1)) {                                                                             \
        pva[j] = *(const pl_u32x4_t*)(inb + poff[(j) & 1] + offa_);                                                \
        pvb[j] = *(const pl_u32x4_t*)(inb + poff[(j) & 1] + offb_); }                                              \
    } while (0)
    // registers -> the other patch buffer; `c_` is the chunk the registers were loaded for (channel tail mask).  Items h and
    // h + 2 hold the channels (4 wave .. 4 wave + 3) of unit h: one 8-byte store per pixel into plane `wave`
#define PL_P_WRITE(buf_, c_, h)                                                                                    \
    do {                                                                                                           \
        const int ca_ = (c_) * 32 + 4 * wave;                                                                      \
        const unsigned cm0_ = (ca_ < p.C ? 0xffffu : 0u) | (ca_ + 1 < p.C ? 0xffff0000u : 0u);                     \
        const unsigned cm1_ = (ca_ + 2 < p.C ? 0xffffu : 0u) | (ca_ + 3 < p.C ? 0xffff0000u : 0u);                 \
        unsigned char* pb_ = pbuf + (buf_) * pbytes + wave * plb;                                                  \
        {                                                                                                          \
            const unsigned m0_ = pmask[h] & cm0_, m1_ = pmask[h] & cm1_;                                           \
            if (!(PL_ABL & 4) && pwr[h]) {                                                                         \
                _Pragma("unroll") for (int e = 0; e < 8; ++e) {                                                    \
                    const unsigned sel_ = (e & 1) ? 0x07060302u : 0x05040100u;                                     \
                    uint2 v_;                                                                                      \
                    v_.x = __builtin_amdgcn_perm(pvb[h][e >> 1], pva[h][e >> 1], sel_) & m0_;                      \
                    v_.y = __builtin_amdgcn_perm(pvb[h + 2][e >> 1], pva[h + 2][e >> 1], sel_) & m1_;              \
                    *(uint2*)(pb_ + ((e & 1) ? (un_dst[h][e >> 1] >> 16) : (un_dst[h][e >> 1] & 0xffffu))) = v_;   \
                }                                                                                                  \
            }                                                                                                      \
            if (!(PL_ABL & 4) && prl[h] >= 0) {       /* image column 1 (element 1) -> patch column of image column -1 */ \
                const int pix_ = (((un_py[h] & 1) * 2 + (prl[h] & 1)) * RH + (un_py[h] >> 1)) * RW + (prl[h] >> 1); \
                uint2 v_;                                                                                          \
                v_.x = __builtin_amdgcn_perm(pvb[h][0], pva[h][0], 0x07060302u) & m0_;                             \
                v_.y = __builtin_amdgcn_perm(pvb[h + 2][0], pva[h + 2][0], 0x07060302u) & m1_;                     \
                *(uint2*)(pb_ + pix_ * 8) = v_;                                                                    \
            }                                                                                                      \
            if (!(PL_ABL & 4) && prr[h] >= 0) {       /* image column W-2 (element 6) -> patch column of image column W */ \
                const int pix_ = (((un_py[h] & 1) * 2 + (prr[h] & 1)) * RH + (un_py[h] >> 1)) * RW + (prr[h] >> 1); \
                uint2 v_;                                                                                          \
                v_.x = __builtin_amdgcn_perm(pvb[h][3], pva[h][3], 0x05040100u) & m0_;                             \
                v_.y = __builtin_amdgcn_perm(pvb[h + 2][3], pva[h + 2][3], 0x05040100u) & m1_;                     \
                *(uint2*)(pb_ + pix_ * 8) = v_;                                                                    \
            }                                                                                                      \
        }                                                                                                          \
    } while (0)
    // fragment reads of tap t_ (the tt-th of its group) into a register set: A operands from ring slot `as_`, B fragments
    // (2 x 8 bytes per slice: planes 4 s + 2 lhi + {0, 1}) from patch buffer `pc_`; PL_MFMAS: that tap's MFMAs
#define PL_FRAGS(FA, FB, as_, pc_, tt, t_)                                                                         \
    do {                                                                                                           \
        const unsigned char* bp_ = (pc_) + (unsigned)(lhi * 2) * plb + (bpix0 + toffs[t_]) * 8u;                   \
        const uint2 q00_ = *(const uint2*)(bp_), q01_ = *(const uint2*)(bp_ + plb);                                \
        const uint2 q10_ = *(const uint2*)(bp_ + 4u * plb), q11_ = *(const uint2*)(bp_ + 5u * plb);                \
        const pl_u32x4_t bq0_ = {q00_.x, q00_.y, q01_.x, q01_.y}, bq1_ = {q10_.x, q10_.y, q11_.x, q11_.y};         \
        FB[0] = __builtin_bit_cast(bf16x8_t, bq0_); FB[1] = __builtin_bit_cast(bf16x8_t, bq1_);                    \
        const unsigned char* ab_ = (as_) + (((tt) * 4 + wm * 2) * 2) * 1024 + lane * 16;                           \
        FA[0] = *(const bf16x8_t*)(ab_);        FA[1] = *(const bf16x8_t*)(ab_ + 1024);                            \
        FA[2] = *(const bf16x8_t*)(ab_ + 2048); FA[3] = *(const bf16x8_t*)(ab_ + 3072);                            \
    } while (0)
#define PL_MFMAS(FA, FB)                                                                                           \
    do {                                                                                                           \
        if constexpr (SPLIT) {                                                                                     \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[1], FB[0], acc0, 0, 0, 0);                           \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[3], FB[0], acc1, 0, 0, 0);                           \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0], FB[1], acc0, 0, 0, 0);                           \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[2], FB[1], acc1, 0, 0, 0);                           \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0], FB[0], acc0, 0, 0, 0);                           \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[2], FB[0], acc1, 0, 0, 0);                           \
        } else {                                                                                                   \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[0], FB[0], acc0, 0, 0, 0);                           \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[2], FB[0], acc1, 0, 0, 0);                           \
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[1], FB[1], acc0, 0, 0, 0);                           \
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(FA[3], FB[1], acc1, 0, 0, 0);                           \
        }                                                                                                          \
    } while (0)
    // bias / activation / store of this wave's 64 rows x 32 pixels of tile `t_`; the accumulators are cleared for the next tile
#define PL_EPILOGUE(t_)                                                                                            \
    do {                                                                                                           \
        const int tn_ = (t_) / tiles_xy, trem_ = (t_) - tn_ * tiles_xy;                                            \
        const int ty_ = trem_ / ph.tiles_x, tx_ = trem_ - ty_ * ph.tiles_x;                                        \
        /* (opaque to the optimiser: with a loop-invariant row base hipcc hoists the epilogue's address arithmetic out of the \
            persistent loop and carries ~140 registers of it through the loop - in scratch) */                    \
        int mrel_ = wm * 64;                                                                                       \
        asm volatile("" : "+v"(mrel_));                                                                            \
        if constexpr ((PL_ABL & 16) != 0) { if (acc0[0] == 12345.678f) ((float*)p.out)[0] = acc1[1]; }             \
        else {                                                                                                     \
            unsigned char* wreg_ = epi_l + wave * 2048;                                                            \
            const int tws_ = p.TW == 32 ? 5 : 4;                                                                   \
            if (p.out_f32) {                                                                                       \
                pl_store_wide<true>(p, ph, acc0, 0, m0, mrel_, lane, wn, ty_ * p.TH, tx_ * p.TW, tn_, tws_, wreg_, 1024, bias_l, osc, slope); \
                pl_store_wide<true>(p, ph, acc1, 1, m0, mrel_, lane, wn, ty_ * p.TH, tx_ * p.TW, tn_, tws_, wreg_, 1024, bias_l, osc, slope); \
            } else {                                                                                               \
                pl_store_wide<false>(p, ph, acc0, 0, m0, mrel_, lane, wn, ty_ * p.TH, tx_ * p.TW, tn_, tws_, wreg_, 1024, bias_l, osc, slope); \
                pl_store_wide<false>(p, ph, acc1, 1, m0, mrel_, lane, wn, ty_ * p.TH, tx_ * p.TW, tn_, tws_, wreg_, 1024, bias_l, osc, slope); \
            }                                                                                                      \
        }                                                                                                          \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }                           \
    } while (0)

    if (loader) {
        // ================= loader role: both operands, one tap group / one item ahead of the compute waves =================
        int tile = t_lo, chunk = 0;
        PL_ISSUE_A(0, 0, 0);
        PL_P_ADDR(tile);
        PL_P_LOAD(0, 0); PL_P_LOAD(1, 0); PL_P_LOAD(2, 0); PL_P_LOAD(3, 0);
        PL_WRITE_A(0, 0);
        PL_P_WRITE(0, 0, 0); PL_P_WRITE(0, 0, 1);
        PL_ISSUE_A(1 % NSET, 0, 1);
        __syncthreads();
        unsigned gcnt = 0;
        int pb = 0;
        for (;;) {
            int ntile = tile, nchunk = chunk + 1;
            if (nchunk == nch) { nchunk = 0; ntile = tile + 1; }
            const bool last = ntile >= t_hi;
            const int ptile = last ? tile : ntile, pchunk = last ? chunk : nchunk;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + 2 < NG) PL_ISSUE_A((g + 2) % NSET, chunk, g + 2); else PL_ISSUE_A((g + 2) % NSET, pchunk, g + 2 - NG);
                if (g == 0) {
                    PL_P_ADDR(ptile);
                    PL_P_LOAD(0, pchunk); PL_P_LOAD(1, pchunk); PL_P_LOAD(2, pchunk); PL_P_LOAD(3, pchunk);
                }
                __builtin_amdgcn_sched_barrier(0);
                PL_WRITE_A((g + 1) % NSET, (gcnt + 1u) & 1u);
                if (g == NG - 1) { PL_P_WRITE(pb ^ 1, pchunk, 0); PL_P_WRITE(pb ^ 1, pchunk, 1); }
                __syncthreads();
                ++gcnt;
            }
            pb ^= 1;
            if (last) break;
            tile = ntile; chunk = nchunk;
        }
        return;
    }
    // ===================== compute role: fragment reads + MFMAs, epilogue of the previous tile in the first group =====================
    int tile = t_lo, chunk = 0;
    __syncthreads();                                       // (the loader's prologue)
    unsigned gcnt = 0;
    int pb = 0;
    int epi_tile = -1;
    for (;;) {
        int ntile = tile, nchunk = chunk + 1;
        if (nchunk == nch) { nchunk = 0; ntile = tile + 1; }
        const bool last = ntile >= t_hi;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            if (g == 0 && epi_tile >= 0) {
                PL_EPILOGUE(epi_tile);
                epi_tile = -1;
            }
            // fragments of tap tt + 1 are read while the MFMAs of tap tt run (two register sets): with the reads of a tap
            // directly ahead of its MFMAs every tap began with an LDS round trip (37 % of the wave cycles parked in waits)
            {
                const unsigned char* as_ = aring + (gcnt & 1u) * SLOT;
                const unsigned char* pc_ = pbuf + pb * pbytes;
                bf16x8_t fa[2][4], fb[2][2];
                PL_FRAGS(fa[0], fb[0], as_, pc_, 0, g * TG);
#pragma unroll
                for (int tt = 0; tt < TG; ++tt) {
                    if (tt + 1 < TG) PL_FRAGS(fa[(tt + 1) & 1], fb[(tt + 1) & 1], as_, pc_, tt + 1, g * TG + tt + 1);
                    PL_MFMAS(fa[tt & 1], fb[tt & 1]);
                }
            }
            __syncthreads();
            ++gcnt;
        }
        pb ^= 1;
        if (nchunk == 0) epi_tile = tile;
        if (last) break;
        tile = ntile; chunk = nchunk;
    }
    if (epi_tile >= 0) PL_EPILOGUE(epi_tile);
#undef PL_MFMAS
#undef PL_FRAGS
#undef PL_EPILOGUE
#undef PL_P_WRITE
#undef PL_P_LOAD
#undef PL_P_ADDR
#undef PL_WRITE_A
#undef PL_ISSUE_A
}

// ---------------------------------------------------------------------------------------------------------------------------
// Host side: does the plan qualify, tile / grid choice, weight packing (through gconv.hip's pack path), launch.
// ---------------------------------------------------------------------------------------------------------------------------
int launch_gconv_pl(GcParams& p, const float* w, const float* w_scale, long long sm, long long sc, long long sr, long long ss,
                    WsAlloc& ws, hipStream_t st) {
    if (!gc_env_int("HIFIC_PL", 1)) return HIFIC_ERR_UNSUPPORTED;
    if (p.nphase != 1 || p.ist != 2 || p.ost != 1 || p.in_f32 || p.rfx || p.resid || p.csplit || p.msplit || p.fold_h)
        return HIFIC_ERR_UNSUPPORTED;
    GcPhase& ph = p.ph[0];
    const int nt = ph.ntaps;
    if (nt != 9 && nt != 16) return HIFIC_ERR_UNSUPPORTED;
    if (p.K < gc_env_int("HIFIC_PL_MINK", 48)) return HIFIC_ERR_UNSUPPORTED;
    if (p.IW % 8 != 0 || ph.OWt % 16 != 0 || ph.OHt < 1) return HIFIC_ERR_UNSUPPORTED;
    if (ph.ooy != 0 || ph.oox != 0 || p.OWf != ph.OWt || p.OHf != ph.OHt) return HIFIC_ERR_UNSUPPORTED;     // 16-byte row pieces
    if (p.split && (p.C % 32) != 0) return HIFIC_ERR_UNSUPPORTED;
    if ((long long)p.N * p.C * p.IH * p.IW >= (1ll << 31)) return HIFIC_ERR_UNSUPPORTED;          // 32-bit element offsets
    // spans of the tap table (finish_phase left them in PH / PW)
    const int span_y = ph.PH, span_x = ph.PW;
    if (p.bmode == PAD_REFLECT) {
        // the rim pass covers ONE column left and right of the image; rows are mirrored per load
        int dxmax = 0;
        for (int t = 0; t < nt; ++t) if (t == 0 || p.tap_dx[ph.tap0 + t] > dxmax) dxmax = p.tap_dx[ph.tap0 + t];
        if (ph.dx_min < -1 || (ph.OWt - 1) * 2 + dxmax > p.IW) return HIFIC_ERR_UNSUPPORTED;
        if (p.IW < 2 || p.IH < 2) return HIFIC_ERR_UNSUPPORTED;
    }
    // tile: 128 output pixels of one image, rows of 32 (wide planes) or 16 pixels
    int TW = ph.OWt >= 32 && ph.OWt % 32 == 0 ? 32 : 16;
    if (gc_env_int("HIFIC_PL_TW", 0) == 16) TW = 16;                        // (A/B knob: 16-pixel tile rows everywhere)
    const int TH = 128 / TW;
    const int PH = (TH - 1) * 2 + span_y, PW = (TW - 1) * 2 + span_x;
    const int RH = (PH + 1) / 2, RW = (PW + 1) / 2;
    const int acol = ((ph.dx_min % 8) + 8) & 7;
    const int NGR = (acol + PW + 7) / 8;
    if (PH * NGR > 128) return HIFIC_ERR_UNSUPPORTED;             // (patch row, aligned group) units: two per lane
    const size_t pbytes = (size_t)(4 * RH * RW + 1) * 64;         // 16 channel-pair planes of 4 RH RW + 1 dwords
    // LDS: two ring slots + two patch buffers + bias + the eight compute waves' 2 KB store regions (160 KB per CU)
    int tg = nt == 9 ? 3 : (gc_env_int("HIFIC_PL_TG", 4) == 2 ? 2 : 4);
    if (2 * (size_t)tg * PL_SLOT_TAP + 2 * pbytes + 512 + 16384 > (size_t)160 * 1024) {
        if (nt == 16) tg = 2;
        if (2 * (size_t)tg * PL_SLOT_TAP + 2 * pbytes + 512 + 16384 > (size_t)160 * 1024) return HIFIC_ERR_UNSUPPORTED;
    }
    const size_t lds = 2 * (size_t)tg * PL_SLOT_TAP + 2 * pbytes + 512 + 16384;

    p.TH = TH; p.TW = TW; p.NI = 1; p.tiles_n = p.N;
    p.Kpad = cdiv(p.K, 128) * 128;
    p.Cpad = cdiv(p.C, 32) * 32;
    p.dbg = gc_env_int("HIFIC_DBG", 0);
    p.tap_sw = (int)sr;
    p.afrag = 2; p.ksplit = 1; p.kchunks = 0; p.kpart = nullptr; p.kpart_stride = 0; p.wstage = 0; p.epi_wide = 0;
    ph.PH = PH; ph.PW = PW; ph.PWs = PW;
    ph.tiles_y = cdiv(ph.OHt, TH); ph.tiles_x = cdiv(ph.OWt, TW);
    ph.wp_off = 0;
    const long long wp_elems = (long long)p.Kpad * nt * p.Cpad;
    const int ntile = p.tiles_n * ph.tiles_y * ph.tiles_x;
    const int mtiles = p.Kpad / 128;
    int npg = 256 / mtiles; if (npg < 1) npg = 1; if (npg > ntile) npg = ntile;
    p.pl_tpw = cdiv(ntile, npg);
    npg = cdiv(ntile, p.pl_tpw);
    p.max_tiles = ntile;

    bool plan_only = false;
    const int rcp = gc_pack_weights_bf16(p, wp_elems, w, w_scale, sm, sc, sr, ss, ws, st, &plan_only);
    if (rcp != HIFIC_OK || plan_only) return rcp;
    if (((size_t)p.in & 15) != 0) return HIFIC_ERR_ARG;          // 16-byte loads (the pack layout is already committed: loud)
    // 16-byte stores: aligned output rows (OWf = OWt is a multiple of 16 here), N K OH OW in 32-bit reach is not needed (64-bit math)
    if (((size_t)p.out & 15) != 0) return HIFIC_ERR_ARG;         // 16-byte stores (as above)
    p.epi_wide = 1;
    p.pl_wshare = ((long long)p.Kpad * p.Cpad * nt * 2 > (long long)gc_env_int("HIFIC_PL_WSHARE_KB", 2048) * 1024 && mtiles > 1) ? 1 : 0;

    char ptag[112], kname[64];
    snprintf(ptag, sizeof(ptag), "gconv_pl K%d C%d N%d in%dx%d out%dx%d taps%d tg%d tile%dx%d tpw%d grid%d%s", p.K, p.C, p.N,
             p.IH, p.IW, p.OHf, p.OWf, nt, tg, TH, TW, p.pl_tpw, npg * mtiles, p.split ? " split" : "");
    snprintf(kname, sizeof(kname), "gconv_pl_kernel<%d,%d%s>", nt, tg, p.split ? ",split" : "");
    const int pslot = gc_prof_open(kname, p.aflops, st, ptag);
    gc_prof_bytes(pslot, gc_algo_bytes(p));
    const dim3 grid(npg * mtiles);
#define PL_LAUNCH(NT_, TG_, SP_)                                                                      \
    do {                                                                                              \
        gc_set_max_lds((const void*)gconv_pl_kernel<NT_, TG_, SP_>, (int)lds);                        \
        hipLaunchKernelGGL((gconv_pl_kernel<NT_, TG_, SP_>), grid, dim3(1024), lds, st, p);           \
    } while (0)
    if (nt == 9) { if (p.split) PL_LAUNCH(9, 3, true); else PL_LAUNCH(9, 3, false); }
    else if (tg == 4) { if (p.split) PL_LAUNCH(16, 4, true); else PL_LAUNCH(16, 4, false); }
    else { if (p.split) PL_LAUNCH(16, 2, true); else PL_LAUNCH(16, 2, false); }
#undef PL_LAUNCH
    gc_prof_close(pslot, st);
    return hific_launch_status();
}
