// Device helpers shared by the kernels of the conv engine: NCHW -> LDS transposing stages (stage_T, stage_W) and the epilogues
// of the forward-type kernels.  Included by gconv.hip, gconv_sp9.hip, gconv_wgrad.hip.
#pragma once
#include "gconv.h"
#include "gconv_dev.h"
#include <type_traits>

// K-slice of one MFMA and LDS row padding per element type; BC (channels per chunk) is a kernel template parameter
template <typename T> struct GcCfg;
template <> struct GcCfg<bf16_t> { static constexpr int KS = 16, PAD = 16; };
template <> struct GcCfg<float>  { static constexpr int KS = 2,  PAD = 4; };

// ---------------------------------------------------------------------------------------------------
// Transposing stage: NCHW global -> LDS image [NI*PH*PW rows][DWR dwords], row pitch PITCH bytes.
// bf16: one dword = channels (c0+2*dw, c0+2*dw+1); f32: one dword = channel c0+dw.
// Consecutive threads take consecutive patch pixels => coalesced global reads along W.
// ---------------------------------------------------------------------------------------------------
// Element loads with 32-bit element offsets from a wave-uniform base (tensors on this path are < 2^31 elements):
// one v_add per load instead of 64-bit address arithmetic + selects.
template <bool F32SRC> struct SrcT;
template <> struct SrcT<true>  { typedef float type; };
template <> struct SrcT<false> { typedef bf16_t type; };

// Loads the NDW dwords (channel pairs for bf16, single channels for f32) of one patch pixel into raw registers.
//   full: the whole chunk [c0, c0+BC) is inside [0, C) -> no per-channel predicate
template <typename T, int NDW, bool SF32>
__device__ __forceinline__ void px_load(unsigned (&lo)[NDW], unsigned (&hi)[NDW], const void* src, unsigned qoff,
                                        unsigned plane, int C, int c0, int wv, bool full) {
    if constexpr (std::is_same<T, float>::value) {
        const float* sp = (const float*)src;
        unsigned off = qoff + (unsigned)(c0 + wv) * plane;
#pragma unroll
        for (int i = 0; i < NDW; ++i) {
            const bool okc = full || (c0 + wv + 4 * i < C);
            lo[i] = __float_as_uint(sp[okc ? off : 0u]);
            hi[i] = okc ? 1u : 0u;
            off += 4u * plane;
        }
    } else {
        typedef typename SrcT<SF32>::type S;
        const S* sp = (const S*)src;
        unsigned off = qoff + (unsigned)(c0 + 2 * wv) * plane;
        // Channels past C read element 0 instead (clamped address, unconditional load) and are zeroed at store
        // time: a select on the loaded value right here made the compiler wait for every pair of loads
        // (8 serial round trips per pixel on every partial chunk, i.e. on all of a 60-channel layer).
#pragma unroll
        for (int i = 0; i < NDW; ++i) {
            const int c = c0 + 2 * (wv + 4 * i);
            const unsigned o0 = (full || c < C) ? off : 0u, o1 = (full || c + 1 < C) ? off + plane : 0u;
            if constexpr (SF32) { lo[i] = __float_as_uint(sp[o0]); hi[i] = __float_as_uint(sp[o1]); }
            else { lo[i] = sp[o0]; hi[i] = sp[o1]; }
            off += 8u * plane;
        }
    }
}
// Packs and writes one pixel row; ok=false writes zeros (padding / masked pixels); channels >= C are zeroed
template <typename T, int NDW, bool SF32>
__device__ __forceinline__ void px_store(unsigned char* row, const unsigned (&lo)[NDW], const unsigned (&hi)[NDW],
                                         bool ok, int C, int c0, int wv, bool full) {
#pragma unroll
    for (int i = 0; i < NDW; ++i) {
        unsigned v;
        if constexpr (std::is_same<T, float>::value) {
            v = (ok && hi[i]) ? lo[i] : 0u;
        } else {
            unsigned l = lo[i], h = hi[i];
            if constexpr (SF32) { l = f2bf(__uint_as_float(l)); h = f2bf(__uint_as_float(h)); }
            const int c = c0 + 2 * (wv + 4 * i);
            const bool okl = ok && (full || c < C), okh = ok && (full || c + 1 < C);
            v = (okl ? l : 0u) | ((okh ? h : 0u) << 16);
        }
        *(unsigned*)(row + i * 16) = v;
    }
}
// Decode patch pixel q -> element offset of channel 0 (0 when outside) and validity
__device__ __forceinline__ void px_decode(int q, int npatch, int npp, int PW, float inv_npp, float inv_pw, int n0, int y0,
                                          int x0, int N, int C, int H, int W, int bmode, unsigned& qoff, bool& ok,
                                          int PWs, int& qs) {
    const int img = (int)(((float)q + 0.5f) * inv_npp);          // exact for q < 2^22
    const int r = q - img * npp;
    const int py = (int)(((float)r + 0.5f) * inv_pw);
    const int px = r - py * PW;
    qs = q + (img * (npp / PW) + py) * (PWs - PW);              // storage index: rows are PWs wide
    int iy = y0 + py, ix = x0 + px;
    const int n = n0 + img;
    if (bmode == PAD_REFLECT) { iy = reflect_idx(iy, H); ix = reflect_idx(ix, W); }
    ok = (q < npatch) && (n < N) && ((unsigned)iy < (unsigned)H) && ((unsigned)ix < (unsigned)W);
    qoff = ok ? ((unsigned)n * (unsigned)C * (unsigned)(H * W) + (unsigned)iy * (unsigned)W + (unsigned)ix) : 0u;
}

// QB = patch pixels per lane whose loads are issued back-to-back before the first LDS store.  QB = 1 is one memory
// round trip per 64 pixels (a 645-pixel stride-2 halo patch = 11 serialised round trips per channel chunk: measured
// 166 of the 198 us of the 60->120 stride-2 layer).  Kernels that run one workgroup per CU anyway (full-LDS tiles)
// have 512 VGPRs per lane to spend and use QB = 12: up to 768 pixels x 16 loads in flight, one round trip per chunk.
// W8: 512-thread workgroups (gconv_mp_kernel) - waves 4..7 take the odd 64-pixel groups with the dword split of waves 0..3.
template <typename T, int DWR, int PITCH, bool SF32, int QB, bool W8 = false>
__device__ __forceinline__ void stage_T_impl(unsigned char* lds, const void* src, int N, int C, int H, int W, int bmode,
                                             int n0, int NI, int y0, int x0, int PH, int PW, int c0, int tid, int PWs) {
    // Thread (lane, wave) handles patch pixels q = lane + 64*j and dwords dw = wave + 4*i: the pixel is decoded once
    // and all DWR/4 channel loads of it are issued back-to-back; lanes run along W so every channel row is a
    // coalesced run.
    static_assert(DWR % 4 == 0, "DWR");
    constexpr int NDW = DWR / 4;
    constexpr int BCH = std::is_same<T, float>::value ? DWR : DWR * 2;
    const int npp = PH * PW;
    const int npatch = NI * npp;
    const unsigned plane = (unsigned)(H * W);
    const int lane = tid & 63, wv = W8 ? ((tid >> 6) & 3) : (tid >> 6);
    const float inv_npp = 1.0f / (float)npp, inv_pw = 1.0f / (float)PW;
    const bool full = c0 + BCH <= C;
    for (int q0 = lane + (W8 ? 64 * QB * (tid >> 8) : 0); q0 < npatch; q0 += 64 * QB * (W8 ? 2 : 1)) {
        unsigned qoff[QB]; bool ok[QB]; int qs[QB];
        unsigned lo[QB][NDW], hi[QB][NDW];
#pragma unroll
        for (int b = 0; b < QB; ++b)
            px_decode(q0 + 64 * b, npatch, npp, PW, inv_npp, inv_pw, n0, y0, x0, N, C, H, W, bmode, qoff[b], ok[b], PWs, qs[b]);
#pragma unroll
        for (int b = 0; b < QB; ++b) px_load<T, NDW, SF32>(lo[b], hi[b], src, qoff[b], plane, C, c0, wv, full);
#pragma unroll
        for (int b = 0; b < QB; ++b)
            if (QB == 1 || q0 + 64 * b < npatch)
                px_store<T, NDW, SF32>(lds + (size_t)qs[b] * PITCH + wv * 4, lo[b], hi[b], ok[b], C, c0, wv, full);
    }
}
template <typename T, int DWR, int PITCH, int QB = 1, bool W8 = false>
__device__ __forceinline__ void stage_T(unsigned char* lds, const void* src, int src_f32,
                                        int N, int C, int H, int W, int bmode,
                                        int n0, int NI, int y0, int x0, int PWs, int PH, int PW,
                                        int c0, int tid, int nthreads) {
    if (PWs < PW) PWs = PW;
    if (std::is_same<T, float>::value || src_f32)
        stage_T_impl<T, DWR, PITCH, true, QB, W8>(lds, src, N, C, H, W, bmode, n0, NI, y0, x0, PH, PW, c0, tid, PWs);
    else
        stage_T_impl<T, DWR, PITCH, false, QB, W8>(lds, src, N, C, H, W, bmode, n0, NI, y0, x0, PH, PW, c0, tid, PWs);
}

// ---------------------------------------------------------------------------------------------------
// Wide-load transposing stage (bf16 source, 64-channel chunk, W % 8 == 0, 16-byte aligned tensor): same LDS image
// as stage_T, but every global load is 16 bytes = 8 consecutive pixels of ONE channel (an aligned group of the source
// row), so a 645-pixel x 64-channel stride-2 halo patch is 21 loads per thread instead of 161 two-byte loads and all
// of a chunk's data is in flight in 2-3 batches (stage_T: one memory round trip per 64 patch pixels = 11 serialised
// round trips per chunk, which is what the strided / transposed layers spent their time on).
//   * a wave item = (patch row, pair of aligned 8-pixel groups); lane = (channel pair cp = lane & 31, group lane >> 5):
//     two loads per lane (channels c0+2cp, c0+2cp+1), eight v_perm to interleave them, eight ds_write_b32 to rows
//     px .. px+7, dword column cp: 32 consecutive dwords per half wave and the two halves 8 rows = 288 dwords apart
//     -> all 64 banks, conflict-free.
//   * group elements outside the patch columns, and whole items past the end, are redirected to a dump row (qdump).
//   * zero padding: out-of-image rows / groups store zeros.  Reflect padding: rows are mirrored; the <= pad columns per
//     side that lie outside the image (only on border tiles) are filled by a second, two-byte pass.
// ---------------------------------------------------------------------------------------------------
#ifndef GC_WSTAGE_WB
#define GC_WSTAGE_WB 6
#endif
#ifndef GC_WSTAGE_WB_WG
#define GC_WSTAGE_WB_WG 3
#endif
// NCP = channel pairs per chunk (32: 64-channel chunks; 16: 32-channel chunks, four groups per wave item, whose ds_writes
// are 2-way bank conflicted at the 80-byte pitch).
template <int PITCH, int WB, int NCP = 32>
__device__ __forceinline__ void stage_W(unsigned char* lds, const bf16_t* __restrict__ src, int N, int C, int H, int W,
                                        int bmode, int n0, int NI, int y0, int x0, int PH, int PW, int PWs, int c0,
                                        int tid, int qdump) {
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    constexpr int GPI = 64 / NCP;                              // groups per wave item
    const int lane = tid & 63, wv = tid >> 6;
    const int cp = lane % NCP, gg = lane / NCP;
    const int g_lo = x0 >> 3;                                  // floor(x0 / 8), x0 may be negative
    const int NG = ((x0 + PW - 1) >> 3) - g_lo + 1;            // aligned groups that intersect [x0, x0 + PW)
    const int NG2 = (NG + GPI - 1) / GPI;
    const int NR = NI * PH;
    const int nitems = NR * NG2;
    const unsigned plane = (unsigned)(H * W);
    const int ca = c0 + 2 * cp;
    const bool oka = ca < C, okb = ca + 1 < C;
    const unsigned offa = (oka ? (unsigned)ca : 0u) * plane, offb = (okb ? (unsigned)(ca + 1) : 0u) * plane;
    const unsigned cmask = (oka ? 0xffffu : 0u) | (okb ? 0xffff0000u : 0u);
    const float inv_ng2 = 1.0f / (float)NG2, inv_ph = 1.0f / (float)PH;
    const bool refl = bmode == PAD_REFLECT;
    for (int it0 = wv; it0 < nitems; it0 += 4 * WB) {
        u32x4_t va[WB], vb[WB];
        int qrow[WB], px0[WB];
        unsigned vm[WB];
#pragma unroll
        for (int b = 0; b < WB; ++b) {
            const int it = it0 + 4 * b;
            const int row = (int)(((float)it + 0.5f) * inv_ng2);
            const int g = (it - row * NG2) * GPI + gg;
            const int img = (int)(((float)row + 0.5f) * inv_ph);
            const int py = row - img * PH;
            int iy = y0 + py;
            if (refl) iy = reflect_idx(iy, H);
            const int n = n0 + img;
            const int gx = (g_lo + g) * 8;
            const bool in_patch = it < nitems && g < NG;
            const bool col_in = gx >= 0 && gx + 8 <= W;
            const bool ok = in_patch && n < N && (unsigned)iy < (unsigned)H && col_in;
            const unsigned off = ok ? ((unsigned)n * (unsigned)C * plane + (unsigned)iy * (unsigned)W + (unsigned)gx) : 0u;
            va[b] = *(const u32x4_t*)(src + off + offa);
            vb[b] = *(const u32x4_t*)(src + off + offb);
            // reflect mode leaves the out-of-image columns of valid images to the rim pass
            const bool wr = in_patch && !(refl && !col_in && n < N);
            qrow[b] = wr ? row * PWs : -0x40000000;
            px0[b] = gx - x0;
            vm[b] = ok ? cmask : 0u;
        }
#pragma unroll
        for (int b = 0; b < WB; ++b) {
            // every lane's group entirely inside the patch (all but the first / last group of a row and the items past the end):
            // one base address, the eight rows at immediate offsets - 2 VALU instructions per element instead of 8 (the
            // per-element column test + dump-row select + address multiply made this stage ~57 VALU instructions per 16-byte
            // load, half the VALU time of the strided forward layers)
            const bool whole = qrow[b] >= 0 && px0[b] >= 0 && px0[b] + 8 <= PW;
            if (__builtin_amdgcn_ballot_w64(!whole) == 0) {
                unsigned char* d = lds + (size_t)(qrow[b] + px0[b]) * PITCH + cp * 4;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned a_dw = va[b][e >> 1], b_dw = vb[b][e >> 1];
                    *(unsigned*)(d + e * PITCH) = __builtin_amdgcn_perm(b_dw, a_dw, (e & 1) ? 0x07060302u : 0x05040100u) & vm[b];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const unsigned a_dw = va[b][e >> 1], b_dw = vb[b][e >> 1];
                    const unsigned v = __builtin_amdgcn_perm(b_dw, a_dw, (e & 1) ? 0x07060302u : 0x05040100u) & vm[b];
                    const int px = px0[b] + e;
                    const int q = ((unsigned)px < (unsigned)PW && qrow[b] >= 0) ? qrow[b] + px : qdump;
                    *(unsigned*)(lds + (size_t)q * PITCH + cp * 4) = v;
                }
            }
        }
    }
    if (refl) {
        const int nl = x0 < 0 ? (-x0 < PW ? -x0 : PW) : 0;
        const int ovr = x0 + PW - W;
        const int nr = ovr > 0 ? (ovr < PW ? ovr : PW) : 0;
        const int nrim = nl + nr;
        if (nrim > 0) {
            const unsigned short* sp = (const unsigned short*)src;
            const int cpr = tid % NCP, car = c0 + 2 * cpr;                 // this pass: thread = (channel pair, rim item)
            const bool okar = car < C, okbr = car + 1 < C;
            const unsigned offa_r = (okar ? (unsigned)car : 0u) * plane, offb_r = (okbr ? (unsigned)(car + 1) : 0u) * plane;
            const unsigned cmask_r = (okar ? 0xffffu : 0u) | (okbr ? 0xffff0000u : 0u);
            for (int rr = tid / NCP; rr < NR * nrim; rr += 256 / NCP) {
                const int row = rr / nrim, rc = rr - row * nrim;
                const int img = row / PH, py = row - img * PH;
                const int px = rc < nl ? rc : PW - nr + (rc - nl);
                const int iy = reflect_idx(y0 + py, H), ix = reflect_idx(x0 + px, W);
                const int n = n0 + img;
                const bool ok = n < N;
                const unsigned off = ok ? ((unsigned)n * (unsigned)C * plane + (unsigned)iy * (unsigned)W + (unsigned)ix) : 0u;
                const unsigned lo = sp[off + offa_r], hi = sp[off + offb_r];
                const unsigned v = (lo | (hi << 16)) & (ok ? cmask_r : 0u);
                *(unsigned*)(lds + (size_t)(row * PWs + px) * PITCH + cpr * 4) = v;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Epilogue shared by the forward-type kernels: bias + residual + activation, NCHW store (32 consecutive pixels per
// store instruction).  Everything that needs a LOAD is issued unconditionally up front: a load inside a (even
// wave-uniform) branch makes hipcc wait `vmcnt(0)` right behind it, and the per-element `if (p.bias) v += p.bias[m]`
// this replaces was 16*WM*WN serialised L2 round trips at the end of every workgroup (~10 us on a 90 us launch).
// ---------------------------------------------------------------------------------------------------
// NI_ONLY >= 0: this wave writes only that pixel fragment (K-split kernels); -1: all.
template <bool TF32, int WM, int WN, int NI_ONLY>
__device__ __forceinline__ void gc_epilogue(const GcParams& p, const GcPhase& ph, const f32x16_t a00, const f32x16_t a01,
                                            const f32x16_t a10, const f32x16_t a11, int mbase, int lhi,
                                            const int (&pu)[WN], const int (&pv)[WN], const int (&pn)[WN],
                                            const bool (&pvalid)[WN]) {
    const bool hb = p.bias != nullptr && p.ksplit <= 1;          // split-K partials: bias and activation in the reduce pass
    const float* bp = hb ? p.bias : (const float*)p.in;          // always a readable address; masked in the block
    const float slope = p.ksplit > 1 ? 1.f : (p.act == ACT_RELU ? 0.f : (p.act == ACT_LEAKY ? 0.2f : 1.f));
    if constexpr (NI_ONLY < 0 || NI_ONLY == 0) {
        gc_store_block<TF32>(p, ph, a00, 0, mbase, lhi, pu[0], pv[0], pn[0], pvalid[0], hb, bp, slope);
        if constexpr (WM == 2) gc_store_block<TF32>(p, ph, a10, 1, mbase, lhi, pu[0], pv[0], pn[0], pvalid[0], hb, bp, slope);
    }
    if constexpr (WN == 2 && (NI_ONLY < 0 || NI_ONLY == 1)) {
        gc_store_block<TF32>(p, ph, a01, 0, mbase, lhi, pu[WN - 1], pv[WN - 1], pn[WN - 1], pvalid[WN - 1], hb, bp, slope);
        if constexpr (WM == 2) gc_store_block<TF32>(p, ph, a11, 1, mbase, lhi, pu[WN - 1], pv[WN - 1], pn[WN - 1], pvalid[WN - 1], hb, bp, slope);
    }
}

// Wide-store epilogue (p.epi_wide, set by the plan when it is legal): the per-element NCHW stores above are 2 bytes
// per lane - 16*WM*WN store instructions per thread, store-ISSUE bound (58 of the 216 us of the 60->120 stride-2
// layer, 1.1 TB/s).  Here each wave transposes its (WM*32 rows) x (NIW*32 pixels) bf16 tile through a private LDS
// region and writes it back as 16-byte pieces (8 consecutive pixels of one row): 8x fewer store instructions.
// Requirements checked on the host: output stride 1, bf16 output, no fold / residual, TW % 8 == 0, OWf % 8 == 0,
// OWt % 8 == 0 (a piece is entirely inside or entirely outside the image).
template <int WN, int NI_ONLY>
__device__ __forceinline__ void gc_wide_rows(const GcParams& p, const f32x16_t a0, const f32x16_t a1, int mi, int mbase,
                                             int lhi, int l31, bool hb, const float* bp, float slope, int ni0, int rowb,
                                             unsigned char* wave_lds) {
    float bv[16];                                           // the 16 bias loads of a row block in flight together
    const float osc = p.oscale ? *p.oscale : 1.f;
#pragma unroll 16
    for (int r = 0; r < 16; ++r) {
        const int m = mbase + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        bv[r] = bp[(hb && m < p.K) ? m : 0];
    }
    float va[16], vb[16];
#pragma unroll 16
    for (int r = 0; r < 16; ++r) {
        const int ml = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        const float b_ = (hb && mbase + ml < p.K) ? bv[r] : 0.f;
        const float x0 = a0[r] * osc + b_, x1 = a1[r] * osc + b_;
        va[r] = x0 > 0.f ? x0 : x0 * slope;
        vb[r] = x1 > 0.f ? x1 : x1 * slope;
    }
    if constexpr (NI_ONLY < 0 || NI_ONLY == 0) {
#pragma unroll 16
        for (int r = 0; r < 16; ++r) {
            const int ml = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            *(bf16_t*)(wave_lds + ml * rowb + ((0 - ni0) * 32 + l31) * 2) = f2bf(va[r]);
        }
    }
    if constexpr (WN >= 2 && (NI_ONLY < 0 || NI_ONLY == 1)) {
#pragma unroll 16
        for (int r = 0; r < 16; ++r) {
            const int ml = mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            *(bf16_t*)(wave_lds + ml * rowb + ((1 - ni0) * 32 + l31) * 2) = f2bf(vb[r]);
        }
    }
}

// NI_ONLY is a template parameter on purpose: with a runtime fragment index hipcc turns `for ni: if (ni == k)` into a
// dynamically indexed accumulator access and moves the whole accumulator array to scratch (seen: 320 B/lane, every
// MFMA step re-loading its accumulators).
// The accumulators arrive BY VALUE, one vector per (row block, pixel fragment): a reference to the accumulator array
// kept it in scratch on the WM = 2 kernels.
template <int WM, int WN, int NI_ONLY>
__device__ __forceinline__ void gc_epilogue_wide(const GcParams& p, const GcPhase& ph, const f32x16_t a00, const f32x16_t a01,
                                                 const f32x16_t a10, const f32x16_t a11, int mbase,
                                                 int lane, int wn, int u0, int v0, int n0, unsigned char* wave_lds) {
    constexpr int ni_only = NI_ONLY;
    const int l31 = lane & 31, lhi = lane >> 5;
    const bool hb = p.bias != nullptr;
    const float* bp = hb ? p.bias : (const float*)p.in;
    const float slope = p.act == ACT_RELU ? 0.f : (p.act == ACT_LEAKY ? 0.2f : 1.f);
    const int niw = ni_only >= 0 ? 1 : WN;                  // pixel fragments written by this wave
    const int ni0 = ni_only >= 0 ? ni_only : 0;
    const int rowb = niw * 64 + 16;                         // bytes per LDS row (padding: conflict-free 16-byte reads)
    // one 32-row block at a time through a helper with compile-time accumulator indices (an `mi` loop left the
    // accumulators dynamically indexed on the WM = 2 kernels)
    gc_wide_rows<WN, NI_ONLY>(p, a00, a01, 0, mbase, lhi, l31, hb, bp, slope, ni0, rowb, wave_lds);
    if constexpr (WM == 2) gc_wide_rows<WN, NI_ONLY>(p, a10, a11, 1, mbase, lhi, l31, hb, bp, slope, ni0, rowb, wave_lds);
    __builtin_amdgcn_s_waitcnt(0xc07f);                     // lgkmcnt(0): this wave's LDS writes landed (private region)
    const int gpr = niw * 4;                                // 16-byte pieces per row
    const int npieces = WM * 32 * gpr;
    const int thw = p.TH * p.TW;
    const size_t plane = (size_t)p.OHf * p.OWf;
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    for (int q = lane; q < npieces; q += 64) {
        const int ml = q / gpr, g = q - ml * gpr;
        const int pt = (wn * WN + ni0) * 32 + g * 8;        // first pixel of the piece inside the 128-pixel tile
        const int img = pt / thw;
        const int rem = pt - img * thw;
        const int ty_ = rem / p.TW, tx_ = rem - ty_ * p.TW;
        const int m = mbase + ml, n = n0 + img, oy = u0 + ty_ + ph.ooy, ox = v0 + tx_ + ph.oox;
        if (m < p.K && img < p.NI && n < p.N && u0 + ty_ < ph.OHt && v0 + tx_ < ph.OWt) {
            const u32x4_t v = *(const u32x4_t*)(wave_lds + ml * rowb + g * 16);
            *(u32x4_t*)((bf16_t*)p.out + ((size_t)n * p.K + m) * plane + (size_t)oy * p.OWf + ox) = v;
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Wide-store epilogue of the persistent kernels (gconv_pl_kernel, gconv_wr_kernel)
// ---------------------------------------------------------------------------------------------------
#ifndef PL_ABL
#define PL_ABL 0           // timing ablations of gconv_pl.hip (bits 64 / 128 act here)
#endif
typedef unsigned int pl_u32x4_t __attribute__((ext_vector_type(4)));

// Wide-store epilogue of one 32-row block (mi) of a wave's 64 x 32 tile: the accumulator fragment (lane = pixel, 16 rows per
// lane) is transposed through a WAVE-PRIVATE LDS region and leaves as 16-byte pieces (8 bf16 / 4 f32 pixels of one row) - 4 (bf16)
// or 8 (f32) store instructions per block instead of 16 two- / four-byte ones.  The region is two `cst`-byte chunks at `wreg`, private
// to the wave (no barrier).
template <bool F32O>
__device__ __forceinline__ void pl_store_wide(const GcParams& p, const GcPhase& ph, const f32x16_t a, int mi, int m0, int mrel,
                                              int lane, int wn, int u0, int v0, int n, int tw_shift, unsigned char* wreg,
                                              int cst, const float* bias_l, float osc, float slope) {
    const int l31 = lane & 31, lhi = lane >> 5;
    const int mbase = m0 + mrel;
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ml = mrel + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;     // row inside the 128-row tile: bias from LDS (no
        const float x = a[r] * osc + bias_l[ml];                              // vector-memory load between the prefetch requests)
        v[r] = x > 0.f ? x : x * slope;
    }
    const size_t plane = (size_t)p.OHf * p.OWf;
    if constexpr (!F32O) {
        // [32 rows][64 bytes] = 2 chunks of 16 rows
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ml = (r & 3) + 8 * (r >> 2) + 4 * lhi;
            *(bf16_t*)(wreg + (ml >> 4) * cst + (ml & 15) * 64 + l31 * 2) = f2bf(v[r]);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);                 // lgkmcnt(0): this wave's LDS writes have landed
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int q = lane + 64 * i;                    // piece: row q >> 2, pixels (q & 3) * 8 .. + 7 of the wave's 32
            const int row = q >> 2, ptl = wn * 32 + (q & 3) * 8;
            const int ty = ptl >> tw_shift, tx = ptl & ((1 << tw_shift) - 1);
            const int m = mbase + mi * 32 + row;
            pl_u32x4_t d = *(const pl_u32x4_t*)(wreg + (row >> 4) * cst + (row & 15) * 64 + (q & 3) * 16);
            if constexpr ((PL_ABL & 128) != 0) { d[0] = (unsigned)q; d[1] = d[2] = d[3] = 0u; }
            if (!(PL_ABL & 64) && m < p.K && u0 + ty < ph.OHt && v0 + tx < ph.OWt)
                *(pl_u32x4_t*)((bf16_t*)p.out + ((size_t)n * p.K + m) * plane + (size_t)(u0 + ty) * p.OWf + (v0 + tx)) = d;
        }
    } else {
        // two halves of 16 rows: [16 rows][128 bytes] = 2 chunks of 8 rows (accumulator registers 8h .. 8h+7 hold rows 16h .. 16h+15)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) {
                const int r = h * 8 + r8;
                const int ml = (r & 3) + 8 * ((r >> 2) & 1) + 4 * lhi;           // row inside the half
                *(float*)(wreg + (ml >> 3) * cst + (ml & 7) * 128 + l31 * 4) = v[r];
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int q = lane + 64 * i;                // piece: row q >> 3 of the half, pixels (q & 7) * 4 .. + 3
                const int row = q >> 3, ptl = wn * 32 + (q & 7) * 4;
                const int ty = ptl >> tw_shift, tx = ptl & ((1 << tw_shift) - 1);
                const int m = mbase + mi * 32 + h * 16 + row;
                const pl_u32x4_t d = *(const pl_u32x4_t*)(wreg + (row >> 3) * cst + (row & 7) * 128 + (q & 7) * 16);
                if (!(PL_ABL & 64) && m < p.K && u0 + ty < ph.OHt && v0 + tx < ph.OWt)
                    *(pl_u32x4_t*)((float*)p.out + ((size_t)n * p.K + m) * plane + (size_t)(u0 + ty) * p.OWf + (v0 + tx)) = d;
            }
            if (h == 0) __builtin_amdgcn_s_waitcnt(0xc07f); // the reads of half 0 are done before half 1 overwrites the region
        }
    }
}

