// Tap-table implicit-GEMM convolution kernels (gfx950): see gconv.h for the op coverage.
//
// GEMM view per workgroup:  D[BM out-channels][128 pixels] += W[BM][tap,c] * X[tap,c][128 pixels]
//   * pixels tile = NI images x TH rows x TW cols of the (u,v) output domain of one stride phase
//   * the NCHW input halo patch of a c-chunk is read once from HBM (coalesced along W), transposed
//     through registers into LDS as [patch pixel][c] and re-used by every tap (im2col-free)
//   * packed weights [k][tap][c] stream through a double-buffered LDS tile, one (tap, c-chunk) per step
//   * bf16: v_mfma_f32_32x32x16_bf16, f32 parity mode: v_mfma_f32_32x32x2_f32 (exact f32 fma chain)
// Weight-gradient kernel: D[64 m][64 c] per tap += A^T[pixels][m] * Xpatch^T[pixels + tap][c], reduction over
// pixels; both operands use the same transposed LDS image, bf16 fragments come from ds_read_b64_tr_b16.
#include "gconv.h"
#include "gconv_dev.h"
#include <type_traits>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

// build-time A/B switches (csrc/build.sh EXTRA=-D...; lib.py loads $HIFIC_LIB_PATH when set)
#ifndef GC_TOFF_EARLY
#define GC_TOFF_EARLY 0     // gconv_kernel: tap offsets of a step read from LDS before the step's barriers - measured
                            // (round 3, A/B libraries in one run): 1-3 % SLOWER on every generic instantiation: off
#endif
#ifndef SP9_ABL
#define SP9_ABL 0           // timing ablations of gconv_sp9_kernel's A-from-global loop (WRONG RESULTS; tools/micro_sp9.py only):
#endif                      // 1 = no patch loads / LDS writes, 2 = no A loads, 4 = no B fragment reads, 8 = no chunk barrier
#ifndef SP9_TOFF_ARG
#define SP9_TOFF_ARG 1      // gconv_sp9_kernel: tap offsets from the kernel arguments instead of the LDS table (-1..2 %)
#endif

// Dynamic-LDS opt-in per kernel function, raised monotonically (never lowered): a launch recorded in a hipGraph is
// replayed later, when another layer's launch of the same function may have asked for less; and the attribute call
// leaves the per-launch host path once a function has reached its maximum.
#include <mutex>
#include <unordered_map>
void gc_set_max_lds(const void* fn, int bytes) {
    // the attribute belongs to the CURRENT device's function object: one record per (device, function)
    static std::mutex mu;
    static std::unordered_map<unsigned long long, int> cur;
    int dev = 0;
    hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    int& c = cur[(unsigned long long)(uintptr_t)fn * 64u + (unsigned)dev];
    if (bytes > c) {
        hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        c = bytes;
    }
}

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
// Forward-type kernel
// ---------------------------------------------------------------------------------------------------
// TPS = taps per barrier step.  Layers with few channels and many taps (7x7 / 11x11 with 3 channels, 5x5 hyperprior
// convs, 3-channel outputs) run 2-4 MFMAs per wave per tap: with one tap per step the kernel is bound by the barrier and
// the weight-tile latency of 49-121 steps (measured 610 us for the 60->3 7x7 layer whose MFMA time is ~80 us).  TPS
// weight tiles are fetched, stored and consumed per step instead.
// QBW = patch pixels per lane per staging batch (stage_T QB); -1 = the wide-load staging variant (stage_W).  A separate
// instantiation on purpose: with both loaders behind a runtime flag the 64-row kernel went from 3 to 2 waves per SIMD
// (189 VGPRs) and every stride-1 layer on it lost 15-35 %.
// SPLIT (bf16, BC >= 32): native split-bf16 reduction of the exact-index chain.  Both operands arrive in the pair layout
// (hific_split3 which = 2): K-slices (2j, 2j+1) of a chunk are the hi and lo halves of the same 16 real channels, and a
// step issues hi*hi + hi*lo + lo*hi per slice pair - 3 MFMAs on 2 + 2 staged fragments, where the (hi, lo, hi) x (hi, hi, lo)
// form over 3C channels of the plain kernel stages 3 + 3 (a third more patch staging, weight tiles, LDS reads and barrier
// steps for the same MFMAs).
template <typename T, int BC, int WGM, int WGN, int WM, int WN, int QBW, int TPS, bool SPLIT = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(QBW > 1 ? 1 : 2, QBW > 1 ? 1 : 8)))
void gconv_kernel(const GcParams p) {
    constexpr bool WIDE = QBW < 0;
    static_assert(!SPLIT || (std::is_same<T, bf16_t>::value && BC >= 32), "split reduction: bf16, slice pairs");
    constexpr int QB = QBW < 0 ? 1 : QBW;
    static_assert(!WIDE || (std::is_same<T, bf16_t>::value && BC >= 32), "wide staging: bf16, 32/64-channel chunks");
    using Cfg = GcCfg<T>;
    constexpr int KS = Cfg::KS;
    constexpr int ROWB = BC * (int)sizeof(T);          // bytes of one LDS row (BC channels)
    constexpr int PITCH = ROWB + Cfg::PAD;
    constexpr int DWR = ROWB / 4;
    constexpr int PPR = ROWB / 16;                     // 16-byte pieces per weight row
    constexpr int BM = WGM * WM * 32;
    static_assert(WGM * WGN == 4, "4 waves");
    static_assert(WGN * WN * 32 == GC_NPIX, "128 pixels per tile");
    static_assert(ROWB % 16 == 0, "row bytes");
    constexpr int WBYTES = BM * PITCH;
    constexpr int NWP = (BM * PPR + 255) / 256;   // 16-byte weight pieces per thread per step

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int l31 = lane & 31, lhi = lane >> 5;

    const GcPhase& ph = p.ph[blockIdx.z];
    const int ntile_ph = p.tiles_n * ph.tiles_y * ph.tiles_x;
    // XCD-aware mapping: the dispatcher places block b on XCD b % 8 (speed only, never correctness).  Blocks are
    // renumbered so that each XCD works on a contiguous range of the m-major (m-tile, pixel-tile) list: an XCD then
    // streams only ~1/8 of the packed weights through its private 4 MiB L2 instead of all of them.
    int tile, mtile;
    {
        const int nwg = gridDim.x;                       // = max_tiles * mtiles (host)
        const int q8 = nwg >> 3, r8 = nwg & 7;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int q = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;   // bijective
        mtile = q / p.max_tiles;
        tile = q - mtile * p.max_tiles;
    }
    if (tile >= ntile_ph) return;
    const int tx = tile % ph.tiles_x;
    const int ty = (tile / ph.tiles_x) % ph.tiles_y;
    const int tn = tile / (ph.tiles_x * ph.tiles_y);
    const int u0 = ty * p.TH, v0 = tx * p.TW, n0 = tn * p.NI;
    const int m0 = mtile * BM;
    const int PH = ph.PH, PW = ph.PW, PWs = ph.PWs;
    const int npp = PH * PW;
    const int npps = PH * PWs;                    // storage pixels per image
    const int npatch = p.NI * npp;
    const int iy0 = u0 * p.ist + ph.dy_min, ix0 = v0 * p.ist + ph.dx_min;

    int* toffs = (int*)smem;                          // [GC_MAXTAPS] tap -> patch row offset
    unsigned char* wbuf = smem + 512;                 // 2 x TPS x WBYTES
    unsigned char* patch = wbuf + 2 * TPS * WBYTES;   // npatch x PITCH
    if (tid < ph.ntaps)
        toffs[tid] = ((int)p.tap_dy[ph.tap0 + tid] - ph.dy_min) * PWs + ((int)p.tap_dx[ph.tap0 + tid] - ph.dx_min);
    if (GC_TOFF_EARLY) __syncthreads();               // GC_STEP reads the table before its own barriers

    // per-lane pixel decode for the B (pixel) operand and the epilogue
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
        qb[ni] = v ? (img * npps + ty_ * p.ist * PWs + tx_ * p.ist) : 0;
        pu[ni] = u0 + ty_; pv[ni] = v0 + tx_; pn[ni] = n0 + img;
    }

    f32x16_t acc[WM][WN];
#pragma unroll
    for (int mi = 0; mi < WM; ++mi)
#pragma unroll
        for (int ni = 0; ni < WN; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    const int nt = ph.ntaps;
    const int ng = (nt + TPS - 1) / TPS;               // tap groups (steps) per channel chunk
    const int nchunks = p.Cpad / BC;
    // split-K: this workgroup reduces over chunks [chunk_lo, chunk_hi); steps are numbered from 0 inside that range
    int chunk_lo = 0, chunk_hi = nchunks;
    if (p.ksplit > 1) {
        chunk_lo = (int)blockIdx.y * p.kchunks;
        chunk_hi = chunk_lo + p.kchunks < nchunks ? chunk_lo + p.kchunks : nchunks;
    }
    const int s_lo = chunk_lo * ng;
    const int nsteps = (chunk_hi - chunk_lo) * ng;
    const unsigned char* wp_ph = (const unsigned char*)p.wp + (size_t)ph.wp_off * sizeof(T);
    const size_t wrow_bytes = (size_t)nt * p.Cpad * sizeof(T);   // one m-row of this phase

    // Weight tiles stream through a 2-deep LDS ring with a distance-2 register prefetch: at step s the loads of the
    // tiles of step s+2 are issued right after the barrier and those of step s+1 (loaded during step s-1) are written to
    // the other LDS buffer after the MFMAs of step s, so every weight load has two full steps to land.
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t wA[TPS][NWP], wB[TPS][NWP];
    int wrow[NWP], wpart[NWP];
    const unsigned char* wsrc[NWP];
#pragma unroll
    for (int i = 0; i < NWP; ++i) {
        int piece = tid + i * 256;
        if (piece >= BM * PPR) piece = BM * PPR - 1;          // clamp (duplicate load) instead of a divergent branch
        wrow[i] = piece / PPR; wpart[i] = piece % PPR;
        // rows past K (zero padding of the M tile) re-read row K-1 instead: same cache line for every such lane, no L2
        // traffic, and their outputs are never stored.  (K = 3 padded to 32 rows made every workgroup of the 60->3 7x7
        // layer stream 200 KB of zeros: 349 of its 640 us.)
        const int mrow = m0 + wrow[i] < p.K ? m0 + wrow[i] : p.K - 1;
        wsrc[i] = wp_ph + (size_t)mrow * wrow_bytes + wpart[i] * 16;
    }
    // byte offset of tile j of step s = (tap (s % ng) * TPS + j, chunk s / ng); clamped to the last step / last tap
    auto tile_off = [&](int s_, int j_) -> size_t {
        if (s_ >= nsteps) s_ = nsteps - 1;
        s_ += s_lo;
        const int c_ = s_ / ng;
        int t_ = (s_ - c_ * ng) * TPS + j_;
        if (t_ >= nt) t_ = nt - 1;
        return ((size_t)t_ * p.Cpad + (size_t)c_ * BC) * sizeof(T);
    };
#define GC_WLOAD(R, S_)                                                                     \
    do {                                                                                    \
        _Pragma("unroll") for (int j = 0; j < TPS; ++j) {                                   \
            const size_t off_ = tile_off(S_, j);                                            \
            _Pragma("unroll") for (int i = 0; i < NWP; ++i) R[j][i] = *(const u32x4_t*)(wsrc[i] + off_); \
        }                                                                                   \
    } while (0)
#define GC_WSTORE(R, BUF)                                                                   \
    do {                                                                                    \
        _Pragma("unroll") for (int j = 0; j < TPS; ++j)                                     \
        _Pragma("unroll") for (int i = 0; i < NWP; ++i) {                                   \
            if (tid + i * 256 < BM * PPR) {                                                 \
                unsigned char* d = (BUF) + j * WBYTES + wrow[i] * PITCH + wpart[i] * 16;    \
                if constexpr (PITCH % 16 == 0) { *(u32x4_t*)d = R[j][i]; }                  \
                else { ((unsigned*)d)[0] = R[j][i].x; ((unsigned*)d)[1] = R[j][i].y;        \
                       ((unsigned*)d)[2] = R[j][i].z; ((unsigned*)d)[3] = R[j][i].w; }      \
            }                                                                               \
        }                                                                                   \
    } while (0)
    // one GEMM step on the tile in `wb` with tap `t`
#define GC_COMPUTE(wb, t, TOFFV)                                                                                \
    do {                                                                                                        \
        const int toff = GC_TOFF_EARLY ? (TOFFV) : toffs[t];                                                    \
        const unsigned char* arow = (wb) + (wm * WM * 32 + l31) * PITCH;                                        \
        _Pragma("unroll") for (int kk = 0; kk < BC / KS; ++kk) {                                                \
            if constexpr (std::is_same<T, float>::value) {                                                      \
                float a[WM], b[WN];                                                                             \
                _Pragma("unroll") for (int mi = 0; mi < WM; ++mi)                                               \
                    a[mi] = *(const float*)(arow + mi * 32 * PITCH + (kk * 2 + lhi) * 4);                       \
                _Pragma("unroll") for (int ni = 0; ni < WN; ++ni)                                               \
                    b[ni] = *(const float*)(patch + (size_t)(qb[ni] + toff) * PITCH + (kk * 2 + lhi) * 4);      \
                _Pragma("unroll") for (int mi = 0; mi < WM; ++mi)                                               \
                    _Pragma("unroll") for (int ni = 0; ni < WN; ++ni)                                           \
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0); \
            } else if constexpr (SPLIT) {                                                                       \
                if (kk & 1) continue;                                                                           \
                bf16x8_t a[WM], al[WM], b[WN], bl[WN];                                                          \
                _Pragma("unroll") for (int mi = 0; mi < WM; ++mi) {                                             \
                    a[mi] = *(const bf16x8_t*)(arow + mi * 32 * PITCH + kk * 32 + lhi * 16);                    \
                    al[mi] = *(const bf16x8_t*)(arow + mi * 32 * PITCH + (kk + 1) * 32 + lhi * 16);             \
                }                                                                                               \
                _Pragma("unroll") for (int ni = 0; ni < WN; ++ni) {                                             \
                    b[ni] = *(const bf16x8_t*)(patch + (size_t)(qb[ni] + toff) * PITCH + kk * 32 + lhi * 16);   \
                    bl[ni] = *(const bf16x8_t*)(patch + (size_t)(qb[ni] + toff) * PITCH + (kk + 1) * 32 + lhi * 16); \
                }                                                                                               \
                _Pragma("unroll") for (int mi = 0; mi < WM; ++mi)                                               \
                    _Pragma("unroll") for (int ni = 0; ni < WN; ++ni) {                                         \
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mi], b[ni], acc[mi][ni], 0, 0, 0); \
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], bl[ni], acc[mi][ni], 0, 0, 0); \
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0);  \
                    }                                                                                           \
            } else {                                                                                            \
                bf16x8_t a[WM], b[WN];                                                                          \
                if (!(p.dbg & 8)) {                                                                             \
                _Pragma("unroll") for (int mi = 0; mi < WM; ++mi)                                               \
                    a[mi] = *(const bf16x8_t*)(arow + mi * 32 * PITCH + kk * 32 + lhi * 16);                    \
                _Pragma("unroll") for (int ni = 0; ni < WN; ++ni)                                               \
                    b[ni] = *(const bf16x8_t*)(patch + (size_t)(qb[ni] + toff) * PITCH + kk * 32 + lhi * 16);   \
                } else {                                                                                        \
                _Pragma("unroll") for (int mi = 0; mi < WM; ++mi) a[mi] = __builtin_bit_cast(bf16x8_t, dbgv);   \
                _Pragma("unroll") for (int ni = 0; ni < WN; ++ni) b[ni] = __builtin_bit_cast(bf16x8_t, dbgv);   \
                }                                                                                               \
                _Pragma("unroll") for (int mi = 0; mi < WM; ++mi)                                               \
                    _Pragma("unroll") for (int ni = 0; ni < WN; ++ni)                                           \
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi], b[ni], acc[mi][ni], 0, 0, 0); \
            }                                                                                                   \
        }                                                                                                       \
    } while (0)
    // step s: RL = register set that receives the tiles of step s+2, RS = register set holding those of step s+1
#define GC_STEP(s, RL, RS)                                                                  \
    do {                                                                                    \
        /* this step's tap offsets: requested before the barriers / weight loads instead of in front of each tap's   \
           fragment reads (one exposed LDS round trip per tap otherwise; the table never changes) */                \
        int toffv[TPS];                                                                     \
        if (GC_TOFF_EARLY) {                                                                \
            _Pragma("unroll") for (int j = 0; j < TPS; ++j) {                               \
                const int t_ = g * TPS + j;                                                 \
                toffv[j] = toffs[t_ < nt ? t_ : nt - 1];                                    \
            }                                                                               \
        }                                                                                   \
        if (g == 0 && !((p.dbg & 1) && chunk > 0) && !(p.dbg & 64)) {                       \
            __syncthreads();                                                                \
            if constexpr (WIDE)                                                             \
                stage_W<PITCH, GC_WSTAGE_WB, BC / 2>(patch, (const bf16_t*)p.in, p.N, p.C, p.IH, p.IW, p.bmode, n0, p.NI, \
                                                     iy0, ix0, PH, PW, PWs, chunk * BC, tid, p.NI * npps); \
            else                                                                            \
            stage_T<T, DWR, PITCH, QB>(patch, p.in, p.in_f32, p.N, p.C, p.IH, p.IW, p.bmode, \
                                       n0, p.NI, iy0, ix0, PWs, PH, PW, chunk * BC, tid, 256);  \
        }                                                                                   \
        if (!(p.dbg & 16)) __syncthreads();                                                 \
        if (!(p.dbg & 4)) GC_WLOAD(RL, (s) + 2);                                            \
        if (!(p.dbg & 2)) {                                                                 \
            _Pragma("unroll") for (int j = 0; j < TPS; ++j) {                               \
                const int t = g * TPS + j;                                                  \
                if (TPS == 1 || t < nt) GC_COMPUTE(wbuf + (((s) & 1) * TPS + j) * WBYTES, t, toffv[j]); \
            }                                                                               \
        }                                                                                   \
        if (!(p.dbg & 4)) GC_WSTORE(RS, wbuf + (((s) + 1) & 1) * TPS * WBYTES);             \
        if (++g == ng) { g = 0; ++chunk; }                                                  \
    } while (0)

    const u32x4_t dbgv = {(unsigned)tid, 1u, 2u, 3u};
    if (nsteps > 0) {
        GC_WLOAD(wA, 0);
        GC_WLOAD(wB, 1);
        GC_WSTORE(wA, wbuf);
        int chunk = chunk_lo, g = 0, s = 0;
        for (; s + 1 < nsteps; s += 2) {
            GC_STEP(s, wA, wB);
            GC_STEP(s + 1, wB, wA);
        }
        if (s < nsteps) GC_STEP(s, wA, wB);
    }
#undef GC_STEP
#undef GC_COMPUTE
#undef GC_WSTORE
#undef GC_WLOAD

    if (p.dbg & 32) {      // ablation: no epilogue (one never-taken store keeps the accumulators alive)
        if (acc[0][0][0] == 12345.678f) ((float*)p.out)[0] = acc[0][0][1];
        return;
    }
    if constexpr (std::is_same<T, bf16_t>::value) {
        if (p.epi_wide) {
            __syncthreads();                       // every wave is done with the operand buffers
            gc_epilogue_wide<WM, WN, -1>(p, ph, acc[0][0], acc[0][WN - 1], acc[WM - 1][0], acc[WM - 1][WN - 1],
                                         m0 + wm * WM * 32, lane, wn, u0, v0, n0,
                                         smem + (size_t)wave * (WM * 32) * (WN * 64 + 16));
            return;
        }
    }
    gc_epilogue<std::is_same<T, float>::value, WM, WN, -1>(p, ph, acc[0][0], acc[0][WN - 1], acc[WM - 1][0], acc[WM - 1][WN - 1],
                                                        m0 + wm * WM * 32, lhi, pu, pv, pn, pvalid);
}


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

// ---------------------------------------------------------------------------------------------------
// Software-pipelined kernel for the stride-1 3x3 layers (bf16, 9 taps in one phase, 64-channel chunks, halo patch of
// <= 192 pixels).  Same tiling as gconv_kernel; the 9 GEMM steps of a channel chunk are ONE straight-line block in
// which every wave also
//   * loads weight tile s+3 (retired into a 3-deep LDS ring two steps later), and
//   * in steps 0..3 loads 2 of the 8 dword columns of the NEXT chunk's halo patch (retired two steps later into the
//     other patch buffer).
// Everything about a thread's patch items is static: its pixels (lane + 64 j) are decoded once into registers, the
// channel of an item is wave-uniform (scalar base pointer + 32-bit vector offset addressing), LDS offsets are
// immediates.  The number of loads per step is a compile-time constant, so vmcnt accounting stays exact (conditional
// or table-driven loads made the compiler drain with vmcnt(0): measured 1.3-2x slower) and the loads / ds_writes are
// interleaved between the 16 MFMAs of a step.  One barrier per step, no separate staging phase.
// ---------------------------------------------------------------------------------------------------
// KSP = 2: 8 waves; waves 4-7 mirror waves 0-3 on the same output tile but take the upper half of every 64-channel
// chunk's reduction (two waves per SIMD hide each other's LDS/barrier latency at unchanged LDS bytes per MFMA); the
// two partial accumulators are exchanged through LDS at the end and each half writes half of the tile.
// RFX = true: data gradient of a REFLECT-padded 3x3 stride-1 convolution in gather form, on the un-padded output
// domain.  `in` is the extended gradient E[N,K,H+2,W+2] (reflect_extend_kernel: rows/columns 1..H are dY, row 0 =
// dY[0]+dY[2], row H+1 = dY[H-3]+dY[H-1], likewise columns); tap (r,s) of output pixel (i,j) reads E[i+2-r][j+2-s],
// except that the pixels of rows/columns 1 and H-2 take the summed border line for the outermost tap and the pixels
// of rows/columns 0 and H-1 read zero there (the reflection's adjoint folded into per-lane tap offsets).  No padded
// 18x18 domain (27 % extra MFMA work, 1.5 waves of workgroups), no rim buffer, no fold kernel.
// PHS != 0: the 9 (phase, tap) pairs of a kernel-3 stride-2 TRANSPOSED structure - conv-transpose forward (PHS = 1:
// phases hold 1,2,2,4 taps) and the data gradient of a stride-2 conv (PHS = 2: 4,2,2,1) - in one pass over the INPUT
// domain: all four sub-pixel phases read the same halo patch, so it is staged once per channel chunk (the phase-per-
// launch-slice form staged it four times, 3 memory round trips for as little as ONE MFMA step) and each tap's MFMAs
// accumulate into the accumulator set of its phase.  Four accumulator sets => 64-row tiles, one workgroup per CU.
__host__ __device__ constexpr int sp9_phase(int phs, int t) {
    return phs == 1 ? (t < 1 ? 0 : t < 3 ? 1 : t < 5 ? 2 : 3) : phs == 2 ? (t < 4 ? 0 : t < 6 ? 1 : t < 8 ? 2 : 3) : 0;
}
__host__ __device__ constexpr int sp9_tap_in_phase(int phs, int t) {
    return phs == 1 ? (t < 1 ? t : t < 3 ? t - 1 : t < 5 ? t - 3 : t - 5)
                    : phs == 2 ? (t < 4 ? t : t < 6 ? t - 4 : t < 8 ? t - 6 : t - 8) : t;
}
// DS = true (16-pixel tile rows, taps ordered (dy, dx) with dx ascending by one patch pixel): the B fragment of tap
// (dy, dx+1) for pixel n is the fragment of tap (dy, dx) for pixel n+1, i.e. the neighbouring lane's registers.  A tile
// row is exactly one 16-lane DPP row, so taps dx = 1, 2 of a kernel row take their fragments with `row_shl:1` from the
// previous tap's and only the last pixel of each tile row (lanes 15, 31, 47, 63: the halo column) reads LDS.  B-side LDS
// reads per kernel row: 3 KB -> 1.1 KB per fragment slice (the kernel is LDS-read bound).
// AG = true (128-row tiles, K-split): the packed weights are in MFMA A-fragment order (GcParams::afrag, gc_wp_index) and
// every wave loads its own A operands global -> registers (one contiguous 1 KB per operand, two steps ahead) instead of
// the workgroup staging a weight tile through LDS.  Per step and CU that removes the 16 KB tile write and 32 KB of A-fragment
// reads from LDS (of 83 KB: the kernel was LDS-issue bound), frees the 55 KB weight ring, and leaves the
// patch double buffer as the only shared state: ONE barrier per 64-channel chunk instead of one per tap.
// AG levels (HIFIC_SP9_AG): 1 = as described; 2 = + s_setprio(1) around each step's MFMA cluster (the waves of a workgroup
// are no longer in lockstep, so the CU scheduler has something to arbitrate); 3 = + the B fragments of the next tap are read
// from LDS before the current tap's MFMAs are issued (register double buffer; taps of one chunk share the patch buffer);
// 4 = level 3 without the priority hints.
// KSP = 4 (with AG): four reduction quarters - 8 waves as 2 row positions x 4 quarters; a wave owns one 16-deep slice of every
// 64-channel chunk and a 64-row slab that spans all 128 PIXELS of the tile (WN = 4, one wave column): 2 A + 4 B operands per
// 8 MFMAs, and no two waves load the same A operand (the 64x64 wave tiles of KSP = 2 stream every A operand through the
// vector-memory path twice, once per wave column).  The four partial accumulators meet in a two-stage tree through LDS (each
// stage halves the pixel fragments a wave keeps) and every wave writes one pixel fragment of its two row blocks.
// (Also measured: 128x128 slabs on four waves, one per SIMD with 512 registers and the B register double buffer - half the
//  operand bytes per MFMA on both sides, but nothing hides a wave's own waits: 89 vs 66 us on 960->960 @16x16x16.)
__host__ __device__ constexpr int sp9_threads(int ksp) { return ksp == 4 ? 512 : 256 * ksp; }
template <int WM, int KSP, bool RFX, int PHS, bool DS = false, int AG = 0>
__global__ __launch_bounds__(sp9_threads(KSP)) __attribute__((amdgpu_waves_per_eu(PHS ? 1 : 2, PHS ? 1 : 2)))
void gconv_sp9_kernel(const GcParams p) {
    typedef bf16_t T;
    static_assert(!(PHS && (RFX || KSP != 1)), "phase-merged mode: 4 waves, no reflect gather");
    static_assert(!(DS && (RFX || PHS)), "shifted fragments: plain 3x3 stride-1 forward type only");
    static_assert(AG == 0 || (WM == 2 && (KSP == 2 || KSP == 4) && PHS == 0 && !DS), "A-from-global: 128-row K-split tiles");
    static_assert(KSP != 4 || AG == 1, "four reduction quarters: A-from-global form only");
    constexpr bool W4 = KSP == 4;
    constexpr int NPH = PHS ? 4 : 1;
    constexpr int BC = 64, KS = 16, PITCH = 144, PPR = 8, WGN = W4 ? 1 : 2, WN = W4 ? 4 : 2, NT = 9, QJ = 3;
    constexpr int BM = 2 * WM * 32;
    constexpr int NPOS = 2 * WGN;              // wave positions inside the tile (2 row positions x WGN pixel positions)
    constexpr int NWAVES = NPOS * KSP;
    constexpr int NTHR = 64 * NWAVES;
    constexpr int WBYTES = BM * PITCH, NWP = BM * PPR / NTHR;
    static_assert(BM * PPR % NTHR == 0 && NWP >= 1, "weight pieces per thread");
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kgrp = wave / NPOS, tw = wave % NPOS;             // reduction part, wave position inside the tile
    const int wm = tw / WGN, wn = tw % WGN;
    const int l31 = lane & 31, lhi = lane >> 5;

    const GcPhase& ph = p.ph[PHS ? 4 : 0];            // PHS: slot 4 = the union of the four phases (plan: merged patch)
    const int ntile_ph = p.tiles_n * ph.tiles_y * ph.tiles_x;
    int tile, mtile;
    {
        const int nwg = gridDim.x;
        const int q8 = nwg >> 3, r8 = nwg & 7;
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        const int q = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
        mtile = q / p.max_tiles;
        tile = q - mtile * p.max_tiles;
    }
    if (tile >= ntile_ph) return;
    const int tx = tile % ph.tiles_x;
    const int ty = (tile / ph.tiles_x) % ph.tiles_y;
    const int tn = tile / (ph.tiles_x * ph.tiles_y);
    const int u0 = ty * p.TH, v0 = tx * p.TW, n0 = tn * p.NI;
    const int m0 = mtile * BM;
    const int PH = ph.PH, PW = ph.PW;
    const int npp = PH * PW;
    const int npatch = p.NI * npp;
    const int iy0 = u0 * p.ist + ph.dy_min, ix0 = v0 * p.ist + ph.dx_min;
    const unsigned patch_bytes = (unsigned)(((size_t)(npatch + 2) * PITCH + 15) & ~(size_t)15);   // + dump row + zero row

    int* toffs = (int*)smem;                                   // [16] byte offset of each tap inside the patch
    unsigned char* wbuf = smem + 64;                           // 3 x WBYTES
    unsigned char* pbuf = wbuf + (AG ? 0 : 3 * WBYTES);        // 2 x patch_bytes (AG: no weight ring)
    if (tid < NT)
        toffs[tid] = (((int)p.tap_dy[tid] - ph.dy_min) * PW + ((int)p.tap_dx[tid] - ph.dx_min)) * PITCH;
    if (RFX && tid < 2 * (PITCH / 4))                          // the all-zero pixel row of both patch buffers
        *(unsigned*)(pbuf + (tid / (PITCH / 4)) * patch_bytes + (size_t)(npatch + 1) * PITCH + (tid % (PITCH / 4)) * 4) = 0u;

    // static patch pixels of this thread
    unsigned qoff[QJ], pdst[QJ];
    bool qok[QJ];
    {
        const float inv_npp = 1.0f / (float)npp, inv_pw = 1.0f / (float)PW;
#pragma unroll
        for (int j = 0; j < QJ; ++j) {
            const int q = lane + 64 * j;
            int qs;
            px_decode(q, npatch, npp, PW, inv_npp, inv_pw, n0, iy0, ix0, p.N, p.C, p.IH, p.IW, p.bmode, qoff[j], qok[j], PW, qs);
            pdst[j] = (unsigned)((q < npatch ? q : npatch) * PITCH + wave * 4);
        }
    }

    unsigned brow[WN];                                          // byte offset of this lane's B rows inside a patch buffer
    int pu[WN], pv[WN], pn[WN];
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
        brow[ni] = (unsigned)((v ? (img * npp + ty_ * PW + tx_) : 0) * PITCH + lhi * 16);
        pu[ni] = u0 + ty_; pv[ni] = v0 + tx_; pn[ni] = n0 + img;
    }
    const unsigned arow = (unsigned)((wm * WM * 32 + l31) * PITCH + lhi * 16);
    const int kgrp_k0 = kgrp * (BC / KS / KSP);                 // first 16-deep reduction slice of this wave's half
    // RFX: per-lane byte displacement of the patch row / column read by tap row r / tap column s (RFX_ZERO: reads 0)
    constexpr int RFX_ZERO = -(1 << 28);
    int rfx_r[WN][3], rfx_c[WN][3];
    const unsigned rfx_zrow = (unsigned)((npatch + 1) * PITCH + lhi * 16);
#pragma unroll
    for (int ni = 0; ni < WN; ++ni) {
        const int i = pu[ni], j = pv[ni], H = p.OHf, W = p.OWf;
        rfx_r[ni][0] = !RFX ? 0 : (i == 1 ? -3 * PW * PITCH : (i == H - 1 ? RFX_ZERO : 0));
        rfx_r[ni][1] = 0;
        rfx_r[ni][2] = !RFX ? 0 : (i == H - 2 ? 3 * PW * PITCH : (i == 0 ? RFX_ZERO : 0));
        rfx_c[ni][0] = !RFX ? 0 : (j == 1 ? -3 * PITCH : (j == W - 1 ? RFX_ZERO : 0));
        rfx_c[ni][1] = 0;
        rfx_c[ni][2] = !RFX ? 0 : (j == W - 2 ? 3 * PITCH : (j == 0 ? RFX_ZERO : 0));
    }

    f32x16_t acc[NPH][WM][WN];
#pragma unroll
    for (int f = 0; f < NPH; ++f)
#pragma unroll
        for (int mi = 0; mi < WM; ++mi)
#pragma unroll
            for (int ni = 0; ni < WN; ++ni)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[f][mi][ni][r] = 0.f;

    const int nchunks_all = p.Cpad / BC;
    int chunk_lo = 0, chunk_hi = nchunks_all;          // split-K (GcParams::ksplit): this workgroup's channel chunks
    if (p.ksplit > 1) {
        chunk_lo = (int)blockIdx.y * p.kchunks;
        chunk_hi = chunk_lo + p.kchunks < nchunks_all ? chunk_lo + p.kchunks : nchunks_all;
    }
    const int nchunks = chunk_hi;                       // bound used by the prefetch clamps below
    const unsigned plane = (unsigned)(p.IH * p.IW);
    const bf16_t* inb = (const bf16_t*)p.in;

    // packed weights are per phase [Kpad][taps of the phase][Cpad]: one source pointer set per phase
    unsigned wdst[NWP];
    const unsigned char* wsrc[NPH][NWP];
#pragma unroll
    for (int f = 0; f < NPH; ++f) {
        const GcPhase& pf = p.ph[f];
        const unsigned char* wp_ph = (const unsigned char*)p.wp + (size_t)pf.wp_off * sizeof(T);
        const size_t wrow_bytes = (size_t)pf.ntaps * p.Cpad * sizeof(T);
#pragma unroll
        for (int i = 0; i < NWP; ++i) {
            const int piece = tid + i * NTHR;
            wdst[i] = (unsigned)((piece / PPR) * PITCH + (piece % PPR) * 16);
            const int mrow = m0 + piece / PPR < p.K ? m0 + piece / PPR : p.K - 1;      // padded rows: see gconv_kernel
            wsrc[f][i] = wp_ph + (size_t)mrow * wrow_bytes + (piece % PPR) * 16;
        }
    }

    // AG: this wave's A operands in the fragment-ordered image: [(32-row block * NT + tap) * nchunks + chunk][4 slices][1 KB]
    const unsigned char* abase[WM];
#pragma unroll
    for (int mi = 0; mi < WM; ++mi)
        abase[mi] = (const unsigned char*)p.wp + ((size_t)((m0 >> 5) + wm * WM + mi) * NT * nchunks_all) * 4096 + lane * 16;
    // A operands are requested two steps ahead (ring of three register sets); patch rows are written to LDS PDS steps after
    // their request - one step in the 64x128-slab form, whose 128 accumulator registers leave room for two patch sets only.
    // (Measured, round 3: patch distances 3 and 4, patch requests ahead of the A requests, and A operands by LDS-DMA into a
    //  wave-private ring all ran within +-1.5 % of this schedule; a ring of four A sets needs the chunk loop unrolled x4 and
    //  spills.  Vector loads return in order, so a wait for one load is a wait for every older one.)
    constexpr int PDS = (AG == 1 && W4) ? 1 : 2, PSETS = PDS + 1;
    u32x4_t aS[3][WM][BC / KS / KSP];
    u32x4_t bS[2][WN][BC / KS / KSP];                           // AG >= 3: B fragments of the current / next tap
    constexpr int PD = 8 / NWAVES;                              // patch dword columns issued per step (steps 0..3)
    u32x4_t wS[3][NWP];
    unsigned short rlo[3][PD * QJ], rhi[3][PD * QJ];
    u32x4_t bsh[WN][BC / KS / KSP];                             // DS: B fragments carried from tap to tap of a kernel row
    const bool edge_lane = (l31 & 15) == 15;                    // last pixel of a 16-pixel tile row

    // prologue: patch of chunk 0 staged synchronously (by the first four waves: stage_T's thread map is 4 waves wide),
    // weight tiles 0..2 requested, tile 0 in ring slot 0
    if (KSP == 1 || tid < 256)
        stage_T<T, 32, PITCH>(pbuf, p.in, 0, p.N, p.C, p.IH, p.IW, p.bmode, n0, p.NI, iy0, ix0, 0, PH, PW, chunk_lo * BC, tid, 256);

    // weight tile (chunk cc, tap tt); tiles past the end re-read the last chunk (never consumed)
#define SP_WISSUE(SET, cc, tt)                                                                     \
    do { const int c_ = (cc) < nchunks ? (cc) : nchunks - 1;                                       \
         const size_t off_ = ((size_t)sp9_tap_in_phase(PHS, tt) * p.Cpad + (size_t)c_ * BC) * sizeof(T); \
         _Pragma("unroll") for (int i = 0; i < NWP; ++i)                                           \
             wS[SET][i] = *(const u32x4_t*)(wsrc[sp9_phase(PHS, tt)][i] + off_); } while (0)
#define SP_AISSUE(SET, cc, tt)                                                                     \
    do { const int c_ = (cc) < nchunks ? (cc) : nchunks - 1;                                       \
         const size_t off_ = (((size_t)(tt) * nchunks_all + c_) * 4 + kgrp_k0) * 1024;             \
         _Pragma("unroll") for (int mi = 0; mi < WM; ++mi)                                         \
             _Pragma("unroll") for (int kq = 0; kq < BC / KS / KSP; ++kq)                          \
                 aS[SET][mi][kq] = *(const u32x4_t*)(abase[mi] + off_ + kq * 1024); } while (0)
#define SP_WRETIRE(SET, SLOT)                                                                      \
    do { _Pragma("unroll") for (int i = 0; i < NWP; ++i)                                           \
             *(u32x4_t*)(wbuf + (SLOT) * WBYTES + wdst[i]) = wS[SET][i]; } while (0)
    // dword columns of the next chunk's patch taken by this wave in step tt (PD per step, 32 columns per chunk over
    // NWAVES waves and 4 steps): column = wave + NWAVES*(PD*tt + d); channel c = c0n + 2*column is wave-uniform
#define SP_PISSUE(SET, tt)                                                                                  \
    do {                                                                                                    \
        _Pragma("unroll") for (int d = 0; d < PD; ++d) {                                                    \
            const int c = c0n + 2 * (wave + NWAVES * (PD * (tt) + d));                                      \
            const bf16_t* pl0 = inb + (size_t)(c < p.C ? c : 0) * plane;                                    \
            const bf16_t* pl1 = inb + (size_t)(c + 1 < p.C ? c + 1 : 0) * plane;                            \
            _Pragma("unroll") for (int j = 0; j < QJ; ++j) {                                                \
                rlo[SET][d * QJ + j] = pl0[qoff[j]];                                                        \
                rhi[SET][d * QJ + j] = pl1[qoff[j]];                                                        \
            }                                                                                               \
        }                                                                                                   \
    } while (0)
#define SP_PRETIRE(SET, tt)                                                                                 \
    do {                                                                                                    \
        _Pragma("unroll") for (int d = 0; d < PD; ++d) {                                                    \
            const int c = c0n + 2 * (wave + NWAVES * (PD * (tt) + d));                                      \
            const bool c0ok = c < p.C, c1ok = c + 1 < p.C;                                                  \
            _Pragma("unroll") for (int j = 0; j < QJ; ++j) {                                                \
                const unsigned lo_ = (qok[j] && c0ok) ? (unsigned)rlo[SET][d * QJ + j] : 0u;                \
                const unsigned hi_ = (qok[j] && c1ok) ? (unsigned)rhi[SET][d * QJ + j] : 0u;                \
                *(unsigned*)(pnext + pdst[j] + (PD * (tt) + d) * 4 * NWAVES) = lo_ | (hi_ << 16);           \
            }                                                                                               \
        }                                                                                                   \
    } while (0)
#define SP_COMPUTE(SLOT, tt)                                                                                    \
    do {                                                                                                        \
        const unsigned toff = SP9_TOFF_ARG                                                                      \
            ? (unsigned)((((int)p.tap_dy[tt] - ph.dy_min) * PW + ((int)p.tap_dx[tt] - ph.dx_min)) * PITCH)      \
            : (unsigned)toffs[tt];                                                                              \
        const unsigned char* ab = wbuf + (SLOT) * WBYTES + arow;                                                \
        unsigned bo[WN];                                                                                        \
        _Pragma("unroll") for (int ni = 0; ni < WN; ++ni) {                                                     \
            if constexpr (RFX) {   /* taps are enumerated r-major: r = tt / 3, s = tt % 3 */                    \
                const int d_ = rfx_r[ni][(tt) / 3] + rfx_c[ni][(tt) % 3];                                       \
                bo[ni] = d_ < RFX_ZERO / 2 ? rfx_zrow : (unsigned)((int)(brow[ni] + toff) + d_);                \
            } else bo[ni] = brow[ni] + toff;                                                                    \
        }                                                                                                       \
        _Pragma("unroll") for (int kq = 0; kq < BC / KS / KSP; ++kq) {                                          \
            const int kk = kq + kgrp_k0;                                                                        \
            bf16x8_t a[WM], b[WN];                                                                              \
            _Pragma("unroll") for (int mi = 0; mi < WM; ++mi) {                                                 \
                if constexpr (AG != 0) a[mi] = __builtin_bit_cast(bf16x8_t, aS[SLOT][mi][kq]);                       \
                else a[mi] = *(const bf16x8_t*)(ab + mi * 32 * PITCH + kk * 32);                                \
            }                                                                                                   \
            _Pragma("unroll") for (int ni = 0; ni < WN; ++ni) {                                                 \
                if constexpr (DS) {                                                                             \
                    if ((tt) % 3 == 0) {                                                                        \
                        bsh[ni][kq] = *(const u32x4_t*)(pcur + bo[ni] + kk * 32);                               \
                    } else {                                                                                    \
                        u32x4_t edge_ = {0u, 0u, 0u, 0u};                                                       \
                        if (edge_lane) edge_ = *(const u32x4_t*)(pcur + bo[ni] + kk * 32);                      \
                        _Pragma("unroll") for (int d = 0; d < 4; ++d)                                           \
                            bsh[ni][kq][d] = (unsigned)__builtin_amdgcn_update_dpp((int)edge_[d], (int)bsh[ni][kq][d], \
                                                                                   0x101, 0xf, 0xf, false);     \
                    }                                                                                           \
                    b[ni] = __builtin_bit_cast(bf16x8_t, bsh[ni][kq]);                                          \
                } else if constexpr ((SP9_ABL & 4) && AG != 0) b[ni] = __builtin_bit_cast(bf16x8_t, aS[SLOT][0][kq]); \
                else b[ni] = *(const bf16x8_t*)(pcur + bo[ni] + kk * 32);                                       \
            }                                                                                                   \
            _Pragma("unroll") for (int mi = 0; mi < WM; ++mi)                                                   \
                _Pragma("unroll") for (int ni = 0; ni < WN; ++ni)                                               \
                    acc[sp9_phase(PHS, tt)][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                 \
                        a[mi], b[ni], acc[sp9_phase(PHS, tt)][mi][ni], 0, 0, 0);                                \
        }                                                                                                       \
    } while (0)
    // step tt of the current chunk: tile (chunk, tt) sits in ring slot tt%3; issue into register set tt%3,
    // retire the set issued two steps ago ((tt+1)%3) = tile tt+1 -> slot (tt+1)%3
    // AG >= 3: B fragments of tap tt into register set SETB (same address rule as SP_COMPUTE)
#define SP_BLOAD(SETB, tt)                                                                                      \
    do {                                                                                                        \
        /* tap offset from the kernel arguments (constant index: scalar loads hoisted out of the loop), not from the  \
           LDS table: that read sat in front of every step's fragment reads as one more LDS round trip */       \
        const unsigned toff = (unsigned)((((int)p.tap_dy[tt] - ph.dy_min) * PW + ((int)p.tap_dx[tt] - ph.dx_min)) * PITCH); \
        _Pragma("unroll") for (int ni = 0; ni < WN; ++ni) {                                                     \
            unsigned bo_;                                                                                       \
            if constexpr (RFX) {                                                                                \
                const int d_ = rfx_r[ni][(tt) / 3] + rfx_c[ni][(tt) % 3];                                       \
                bo_ = d_ < RFX_ZERO / 2 ? rfx_zrow : (unsigned)((int)(brow[ni] + toff) + d_);                   \
            } else bo_ = brow[ni] + toff;                                                                       \
            _Pragma("unroll") for (int kq = 0; kq < BC / KS / KSP; ++kq)                                        \
                bS[SETB][ni][kq] = *(const u32x4_t*)(pcur + bo_ + (kq + kgrp_k0) * 32);                         \
        }                                                                                                       \
    } while (0)
#define SP_MFMA_REG(SLOT, SETB)                                                                                 \
    do {                                                                                                        \
        if constexpr (AG == 3) __builtin_amdgcn_s_setprio(1);                                                   \
        _Pragma("unroll") for (int kq = 0; kq < BC / KS / KSP; ++kq)                                            \
            _Pragma("unroll") for (int mi = 0; mi < WM; ++mi)                                                   \
                _Pragma("unroll") for (int ni = 0; ni < WN; ++ni)                                               \
                    acc[0][mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(                                   \
                        __builtin_bit_cast(bf16x8_t, aS[SLOT][mi][kq]), __builtin_bit_cast(bf16x8_t, bS[SETB][ni][kq]), \
                        acc[0][mi][ni], 0, 0, 0);                                                               \
        if constexpr (AG == 3) __builtin_amdgcn_s_setprio(0);                                                   \
    } while (0)
    // AG: operands of step tt sit in register set tt%3 (requested two steps earlier); the patch double buffer is the only
    // shared state - next chunk's rows are written in steps 2..5 and first read after the barrier of the next chunk's step 0,
    // which also orders the last reads of the buffer that becomes `pnext` there before its first overwrite (step 2)
    // AG == 1: operands of step tt sit in register set tt % 3 (requested two steps earlier)
#define SP_STEP1(tt)                                                                        \
    do {                                                                                    \
        if constexpr (!(SP9_ABL & 8)) { if ((tt) == 0) __syncthreads(); }                   \
        if constexpr (!(SP9_ABL & 2)) {                                                     \
            if ((tt) + 2 < NT) SP_AISSUE(((tt) + 2) % 3, chunk, (tt) + 2);                  \
            else SP_AISSUE(((tt) + 2) % 3, chunk + 1, (tt) + 2 - NT);                       \
        }                                                                                   \
        if constexpr (!(SP9_ABL & 1)) { if ((tt) < 4) SP_PISSUE((tt) % PSETS, tt); }        \
        /* keep the requests HERE: the scheduler otherwise sinks each load to just above its MFMA two steps later */ \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        SP_COMPUTE((tt) % 3, tt);                                                           \
        if constexpr (!(SP9_ABL & 1)) {                                                     \
            if ((tt) >= PDS && (tt) < 4 + PDS) SP_PRETIRE(((tt) - PDS) % PSETS, (tt) - PDS); \
        }                                                                                   \
    } while (0)
#define SP_STEP(tt)                                                                         \
    do {                                                                                    \
        if constexpr (AG == 1) { SP_STEP1(tt); }                                            \
        else if constexpr (AG != 0) {                                                       \
            if constexpr (!(SP9_ABL & 8)) { if ((tt) == 0) __syncthreads(); }               \
            if constexpr (!(SP9_ABL & 2)) {                                                 \
                if ((tt) + 2 < NT) SP_AISSUE(((tt) + 2) % 3, chunk, (tt) + 2);              \
                else SP_AISSUE(((tt) + 2) % 3, chunk + 1, (tt) + 2 - NT);                   \
            }                                                                               \
            if constexpr (SP9_ABL & 1) { }                                                  \
            else if constexpr (W4 && WM == 2) { if ((tt) < 4) SP_PISSUE((tt) & 1, tt); }    \
            else { if ((tt) < 4) SP_PISSUE((tt) % 3, tt); }                                 \
            /* keep the requests HERE: the scheduler otherwise sinks each load to just above its MFMA two steps later */ \
            __builtin_amdgcn_sched_barrier(0);                                              \
            if constexpr (AG >= 3) {                                                        \
                if ((tt) == 0) SP_BLOAD(0, 0);                                              \
                if ((tt) + 1 < NT) SP_BLOAD(((tt) + 1) & 1, ((tt) + 1 < NT ? (tt) + 1 : 0));  \
                SP_MFMA_REG((tt) % 3, (tt) & 1);                                            \
            } else {                                                                        \
                if constexpr (AG == 2) __builtin_amdgcn_s_setprio(1);                       \
                SP_COMPUTE((tt) % 3, tt);                                                   \
                if constexpr (AG == 2) __builtin_amdgcn_s_setprio(0);                       \
            }                                                                               \
            /* W4 (128 accumulator registers): patch rows retire one step after their request, two register sets */ \
            if constexpr (SP9_ABL & 1) { }                                                  \
            else if constexpr (W4 && WM == 2) { if ((tt) >= 1 && (tt) < 5) SP_PRETIRE(((tt) + 1) & 1, (tt) - 1); }  \
            else { if ((tt) >= 2 && (tt) < 6) SP_PRETIRE(((tt) + 1) % 3, (tt) - 2); }       \
        } else {                                                                            \
            __syncthreads();                                                                \
            if ((tt) + 3 < NT) SP_WISSUE((tt) % 3, chunk, (tt) + 3);                        \
            else SP_WISSUE((tt) % 3, chunk + 1, (tt) + 3 - NT);                             \
            if ((tt) < 4) SP_PISSUE((tt) % 3, tt);                                          \
            SP_COMPUTE((tt) % 3, tt);                                                       \
            SP_WRETIRE(((tt) + 1) % 3, ((tt) + 1) % 3);                                     \
            if ((tt) >= 2 && (tt) < 6) SP_PRETIRE(((tt) + 1) % 3, (tt) - 2);                \
        }                                                                                   \
    } while (0)

    if constexpr (AG != 0) {
        SP_AISSUE(0, chunk_lo, 0);
        SP_AISSUE(1, chunk_lo, 1);
    } else {
        SP_WISSUE(0, chunk_lo, 0);
        SP_WRETIRE(0, 0);
        SP_WISSUE(1, chunk_lo, 1);
        SP_WISSUE(2, chunk_lo, 2);
    }
    for (int chunk = chunk_lo; chunk < chunk_hi; ++chunk) {
        const unsigned char* pcur = pbuf + ((chunk - chunk_lo) & 1) * patch_bytes;
        unsigned char* pnext = pbuf + ((chunk - chunk_lo + 1) & 1) * patch_bytes;
        const int c0n = (chunk + 1 < nchunks ? chunk + 1 : chunk) * BC;     // last chunk: harmless re-load
        SP_STEP(0); SP_STEP(1); SP_STEP(2); SP_STEP(3); SP_STEP(4); SP_STEP(5); SP_STEP(6); SP_STEP(7); SP_STEP(8);
    }
#undef SP_STEP1
#undef SP_STEP
#undef SP_COMPUTE
#undef SP_PRETIRE
#undef SP_PISSUE
#undef SP_WRETIRE
#undef SP_MFMA_REG
#undef SP_BLOAD
#undef SP_AISSUE
#undef SP_WISSUE

    if constexpr (KSP == 2) {
        // Each half keeps one pixel fragment (kgrp 0: ni = 0, kgrp 1: ni = 1): it sends its partial sums of the other
        // fragment through LDS and adds the partner's partial sums of its own.  Region per tile wave: WM*16 floats x 64
        // lanes per direction (operand buffers are free after the barrier).
        float* xch = (float*)smem;
        __syncthreads();
        float* mine = xch + ((size_t)(tw * 2 + kgrp) * WM * 16) * 64 + lane;                // what this wave sends
        const float* theirs = xch + ((size_t)(tw * 2 + (1 - kgrp)) * WM * 16) * 64 + lane;  // what the partner sent
        // accumulator indices must stay compile-time constants (a runtime index would move acc[] to scratch)
        if (kgrp == 0) {
#pragma unroll
            for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) mine[(mi * 16 + r) * 64] = acc[0][mi][1][r];
        } else {
#pragma unroll
            for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) mine[(mi * 16 + r) * 64] = acc[0][mi][0][r];
        }
        __syncthreads();
        if (kgrp == 0) {
#pragma unroll
            for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][mi][0][r] += theirs[(mi * 16 + r) * 64];
        } else {
#pragma unroll
            for (int mi = 0; mi < WM; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[0][mi][1][r] += theirs[(mi * 16 + r) * 64];
        }
    }
    if constexpr (KSP == 4) {
        // Two-stage tree over the four reduction quarters.  Region per wave: 2*WM fragments x 16 floats x 64 lanes (128 KB for
        // the workgroup either way; operand buffers are free after the barrier).  Stage 1, partner kgrp ^ 1: even quarters keep
        // pixel fragments {0,1} and send {2,3}, odd ones the reverse.  Stage 2, partner kgrp ^ 2: of the pair it kept, the lower
        // quarter keeps the first fragment.  Owner of fragment ni: quarters 0, 2, 1, 3.  Accumulator indices stay compile-time
        // constants (see above).
        constexpr int XREG = 2 * WM * 1024;                                     // floats per wave region
        float* xch = (float*)smem;
        float* mine = xch + (size_t)wave * XREG + lane;
        const float* th1 = xch + (size_t)((kgrp ^ 1) * NPOS + tw) * XREG + lane;
        const float* th2 = xch + (size_t)((kgrp ^ 2) * NPOS + tw) * XREG + lane;
        // slot of (row block mi, k-th fragment of the message): mi * FPM + k, FPM fragments per row block in the message
#define X4_SEND(FPM, K, FR)                                                                         \
    _Pragma("unroll") for (int mi = 0; mi < WM; ++mi)                                               \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) mine[((mi * (FPM) + (K)) * 16 + r) * 64] = acc[0][mi][FR][r];
#define X4_RECV(TH, FPM, K, FR)                                                                     \
    _Pragma("unroll") for (int mi = 0; mi < WM; ++mi)                                               \
        _Pragma("unroll") for (int r = 0; r < 16; ++r) acc[0][mi][FR][r] += TH[((mi * (FPM) + (K)) * 16 + r) * 64];
        __syncthreads();
        if ((kgrp & 1) == 0) { X4_SEND(2, 0, 2) X4_SEND(2, 1, 3) }
        else                 { X4_SEND(2, 0, 0) X4_SEND(2, 1, 1) }
        __syncthreads();
        if ((kgrp & 1) == 0) { X4_RECV(th1, 2, 0, 0) X4_RECV(th1, 2, 1, 1) }
        else                 { X4_RECV(th1, 2, 0, 2) X4_RECV(th1, 2, 1, 3) }
        __syncthreads();
        if (kgrp == 0)      { X4_SEND(1, 0, 1) }
        else if (kgrp == 1) { X4_SEND(1, 0, 3) }
        else if (kgrp == 2) { X4_SEND(1, 0, 0) }
        else                { X4_SEND(1, 0, 2) }
        __syncthreads();
        const bool hb = p.bias != nullptr && p.ksplit <= 1;          // as gc_epilogue
        const float* bp = hb ? p.bias : (const float*)p.in;
        const float slope = p.ksplit > 1 ? 1.f : (p.act == ACT_RELU ? 0.f : (p.act == ACT_LEAKY ? 0.2f : 1.f));
        const int mbase = m0 + wm * WM * 32;
        // (the fragment's pixel is decoded again here instead of keeping pu/pv/pn of all four fragments live across the loop)
#define X4_STORE(FR)                                                                                                      \
    do {                                                                                                                  \
        const int pt_ = (FR) * 32 + l31;                                                                                  \
        const int img_ = pt_ / thw, rem_ = pt_ - img_ * thw;                                                              \
        const int ty_ = rem_ / p.TW, tx_ = rem_ - ty_ * p.TW;                                                             \
        _Pragma("unroll") for (int mi = 0; mi < WM; ++mi)                                                                 \
            gc_store_block<false>(p, ph, acc[0][mi][FR], mi, mbase, lhi, u0 + ty_, v0 + tx_, n0 + img_, img_ < p.NI, hb, bp, slope); \
    } while (0)
        if (kgrp == 0)      { X4_RECV(th2, 1, 0, 0) X4_STORE(0); }
        else if (kgrp == 1) { X4_RECV(th2, 1, 0, 2) X4_STORE(2); }
        else if (kgrp == 2) { X4_RECV(th2, 1, 0, 1) X4_STORE(1); }
        else                { X4_RECV(th2, 1, 0, 3) X4_STORE(3); }
#undef X4_STORE
#undef X4_RECV
#undef X4_SEND
    } else if constexpr (PHS != 0) {
#pragma unroll
        for (int f = 0; f < NPH; ++f)
            gc_epilogue<false, WM, WN, -1>(p, p.ph[f], acc[f][0][0], acc[f][0][WN - 1], acc[f][WM - 1][0], acc[f][WM - 1][WN - 1],
                                           m0 + wm * WM * 32, lhi, pu, pv, pn, pvalid);
    } else {
        // (WM == 2 instantiations sit at the 256-register cap: with this path compiled in, the allocator spilled the
        //  accumulators across the chunk loop - 320 B/lane of scratch; their 7.8 MB outputs are not store-bound anyway)
        if (WM == 1 && p.epi_wide) {
            __syncthreads();                       // operand buffers / exchange slots are free
            unsigned char* wl = smem + (size_t)wave * (WM * 32) * ((KSP == 2 ? 1 : WN) * 64 + 16);
            if constexpr (KSP == 2) {
                if (kgrp == 0) gc_epilogue_wide<WM, WN, 0>(p, ph, acc[0][0][0], acc[0][0][1], acc[0][WM - 1][0], acc[0][WM - 1][1],
                                                           m0 + wm * WM * 32, lane, wn, u0, v0, n0, wl);
                else gc_epilogue_wide<WM, WN, 1>(p, ph, acc[0][0][0], acc[0][0][1], acc[0][WM - 1][0], acc[0][WM - 1][1],
                                                 m0 + wm * WM * 32, lane, wn, u0, v0, n0, wl);
            } else {
                gc_epilogue_wide<WM, WN, -1>(p, ph, acc[0][0][0], acc[0][0][1], acc[0][WM - 1][0], acc[0][WM - 1][1],
                                             m0 + wm * WM * 32, lane, wn, u0, v0, n0, wl);
            }
            return;
        }
        if constexpr (KSP == 2) {
            if (kgrp == 0) gc_epilogue<false, WM, WN, 0>(p, ph, acc[0][0][0], acc[0][0][1], acc[0][WM - 1][0], acc[0][WM - 1][1],
                                                         m0 + wm * WM * 32, lhi, pu, pv, pn, pvalid);
            else gc_epilogue<false, WM, WN, 1>(p, ph, acc[0][0][0], acc[0][0][1], acc[0][WM - 1][0], acc[0][WM - 1][1],
                                               m0 + wm * WM * 32, lhi, pu, pv, pn, pvalid);
        } else {
            gc_epilogue<false, WM, WN, -1>(p, ph, acc[0][0][0], acc[0][0][1], acc[0][WM - 1][0], acc[0][WM - 1][1],
                                           m0 + wm * WM * 32, lhi, pu, pv, pn, pvalid);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Weight packing: wp[phase][m][t][c] = w[m*sm + c*sc + r_t*sr + s_t*ss] * scale   (zero padded)
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_w_kernel(const GcParams p, const float* __restrict__ w, const float* scale,
                              long long sm, long long sc, long long sr, long long ss) {
    const GcPhase& ph = p.ph[blockIdx.y];
    const long long total = (long long)p.Kpad * ph.ntaps * p.Cpad;
    const float sc_ = scale ? *scale : 1.f;
    T* dst = (T*)p.wp + ph.wp_off;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(i % p.Cpad);
        const long long j = i / p.Cpad;
        const int t = (int)(j % ph.ntaps);
        const int m = (int)(j / ph.ntaps);
        float v = 0.f;
        if (m < p.K && c < p.C) {
            const int r = p.tap_r[ph.tap0 + t], s = p.tap_s[ph.tap0 + t];
            v = w[gc_weight_index(p, m, c, r, s, sm, sc, sr, ss)] * sc_;
        }
        DT<T>::st(dst + gc_wp_index(p, ph, m, t, c), v);
    }
}

// Coalesced packing through LDS.  The source keeps the R*S taps of one (m, c) pair contiguous; which of m / c is the
// neighbouring dimension (stride R*S) decides the tiling:
//   MODE 0 (c adjacent: conv fwd, conv-transpose bwd-data): block = (one m, 64 c)   -> 64*RS contiguous floats
//   MODE 1 (m adjacent: conv bwd-data, conv-transpose fwd): block = (MB m, 64 c)    -> 64 runs of MB*RS floats
// Output rows wp[phase][m][t][c0..c0+63] are 128-byte (bf16) contiguous stores.  One launch covers all phases.
template <typename T, int MODE>
__device__ __forceinline__ void pack_w2_body(const GcParams& p, const float* __restrict__ w, const float* scale,
                                             long long sm, long long sc, int RS, int MB, int bx, int by) {
    extern __shared__ float pk_lds[];
    const float sc_ = scale ? *scale : 1.f;
    const int c0 = bx * 64;
    const int mb = MB;
    const int m0 = by * mb;
    const int run = mb * RS;                 // floats per c row of the LDS image: [c][ml][rs]
    const int pitch = (run | 1);             // odd pitch: conflict-free column reads
    if (MODE == 0) {
        // MB chunks (one per m row) of 64*RS contiguous floats starting at (m0 + ml, c0); LDS index c*pitch + ml*RS + rs.
        // (One m row per block was 2.3 KB of work per block: 300k blocks per step and 1.7 TB/s; 16 rows per block with
        // four independent loads per trip.)
        const int n = 64 * RS;
        const int cvalid = (p.C - c0 < 64 ? p.C - c0 : 64) * RS;
        const float inv_rs = 1.0f / (float)RS, inv_n = 1.0f / (float)n;
        const int total = mb * n;
        const float* wrow = w + (long long)c0 * sc;
        // 16-byte loads when every m row of the block is 16-byte aligned (C * RS % 4 == 0: all layers but the 3-channel ones):
        // four floats per lane and request instead of one (the pack ran at 2.4-3.1 TB/s with 4-byte loads; Adam streams at 4.7)
        const bool vec4 = ((sm & 3) == 0) && ((((size_t)wrow) & 15) == 0);
        if (vec4) {
            const int total4 = total >> 2;                        // n = 64 * RS is a multiple of 4
            for (int q0 = threadIdx.x; q0 < total4; q0 += 256 * 4) {
                float4 v4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = (q0 + 256 * u) << 2;
                    const int ml = (int)(((float)j + 0.5f) * inv_n);
                    const int i = j - ml * n;
                    const bool ok = j < total && m0 + ml < p.K && i + 3 < cvalid;
                    v4[u] = *(const float4*)(wrow + (ok ? (long long)(m0 + ml) * sm + i : 0));
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = (q0 + 256 * u) << 2;
                    if (j < total) {
                        const int ml = (int)(((float)j + 0.5f) * inv_n);
                        const int i = j - ml * n;
                        const bool rowok = m0 + ml < p.K;
                        const bool whole = i + 3 < cvalid;
                        const float vv[4] = {v4[u].x, v4[u].y, v4[u].z, v4[u].w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int ie = i + e;
                            const int c = (int)(((float)ie + 0.5f) * inv_rs);
                            float x = 0.f;
                            if (rowok && whole) x = vv[e] * sc_;
                            else if (rowok && ie < cvalid) x = wrow[(long long)(m0 + ml) * sm + ie] * sc_;   // (channel tail of the last block)
                            pk_lds[c * pitch + ml * RS + (ie - c * RS)] = x;
                        }
                    }
                }
            }
        } else
        for (int j0 = threadIdx.x; j0 < total; j0 += 256 * 8) {       // 8 independent loads per trip
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + 256 * u;
                const int ml = (int)(((float)j + 0.5f) * inv_n);          // exact for j < 2^22
                const int i = j - ml * n;
                const bool ok = j < total && m0 + ml < p.K && i < cvalid;
                v[u] = wrow[ok ? (long long)(m0 + ml) * sm + i : 0];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int j = j0 + 256 * u;
                if (j < total) {
                    const int ml = (int)(((float)j + 0.5f) * inv_n);
                    const int i = j - ml * n;
                    const int c = (int)(((float)i + 0.5f) * inv_rs);
                    const bool ok = m0 + ml < p.K && i < cvalid;
                    pk_lds[c * pitch + ml * RS + (i - c * RS)] = ok ? v[u] * sc_ : 0.f;
                }
            }
        }
    } else {
        // 64 rows (c) of `run` contiguous floats each: wave w takes rows w, w+4, ...
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        int jmax = (p.K - m0) * RS; if (jmax > run) jmax = run; if (jmax < 0) jmax = 0;
        // 8 rows per trip, unconditional clamped loads: 8 independent 256-byte wave loads in flight per thread
        const float* wm = w + (long long)(m0 < p.K ? m0 : 0) * sm;      // padded m rows: any valid address, zeroed below
        const bool vec4 = ((run & 3) == 0) && ((sc & 3) == 0) && ((((size_t)wm) & 15) == 0) && ((jmax & 3) == 0);
        if (vec4) {
            for (int j4 = lane; j4 < (run >> 2); j4 += 64) {
                const int j = j4 << 2;
                const bool jok = j < jmax;
                const int jc = jok ? j : 0;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    float4 v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int c = wv + 4 * (half * 8 + u);
                        v[u] = *(const float4*)(wm + (long long)(c0 + c < p.C ? c0 + c : 0) * sc + jc);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int c = wv + 4 * (half * 8 + u);
                        const bool ok = jok && c0 + c < p.C;
                        float* d = pk_lds + c * pitch + j;
                        d[0] = ok ? v[u].x * sc_ : 0.f; d[1] = ok ? v[u].y * sc_ : 0.f;
                        d[2] = ok ? v[u].z * sc_ : 0.f; d[3] = ok ? v[u].w * sc_ : 0.f;
                    }
                }
            }
        } else
        for (int j = lane; j < run; j += 64) {
            const bool jok = j < jmax;
            const int jc = jok ? j : 0;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = wv + 4 * (half * 8 + u);
                    v[u] = wm[(long long)(c0 + c < p.C ? c0 + c : 0) * sc + jc];
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int c = wv + 4 * (half * 8 + u);
                    pk_lds[c * pitch + j] = (jok && c0 + c < p.C) ? v[u] * sc_ : 0.f;
                }
            }
        }
    }
    __syncthreads();
    // write: thread = (4 consecutive c, row slot); rows enumerate (ml, t); 8-byte (bf16) / 16-byte (f32) stores
    const int cq = (threadIdx.x & 15) * 4, rslot = threadIdx.x >> 4;
    for (int phi = 0; phi < p.nphase; ++phi) {
        const GcPhase& ph = p.ph[phi];
        T* dst = (T*)p.wp + ph.wp_off;
        const int nrows = mb * ph.ntaps;
        for (int r = rslot; r < nrows; r += 16) {
            const int ml = r / ph.ntaps, t = r - ml * ph.ntaps;
            const int m = m0 + ml;
            if (m >= p.Kpad || c0 + cq >= p.Cpad) continue;
            const int rs = (int)p.tap_r[ph.tap0 + t] * p.tap_sw + (int)p.tap_s[ph.tap0 + t];
            const float* lp = pk_lds + ml * RS + rs;
            const float v0 = lp[(cq + 0) * pitch], v1 = lp[(cq + 1) * pitch], v2 = lp[(cq + 2) * pitch], v3 = lp[(cq + 3) * pitch];
            T* d = dst + gc_wp_index(p, ph, m, t, c0 + cq);        // cq % 4 == 0: the 4 channels stay in one 8-group
            if constexpr (std::is_same<T, float>::value) {
                *(float4*)d = make_float4(v0, v1, v2, v3);
            } else {
                uint2 o;
                o.x = f2bf2(v0, v1);
                o.y = f2bf2(v2, v3);
                *(uint2*)d = o;
            }
        }
    }
}
template <typename T, int MODE>
__global__ __launch_bounds__(256) void pack_w2_kernel(const GcParams p, const float* __restrict__ w,
                                                      const float* scale, long long sm, long long sc, int RS, int MB) {
    pack_w2_body<T, MODE>(p, w, scale, sm, sc, RS, MB, blockIdx.x, blockIdx.y);
}

// Batched packing: ONE launch re-packs every (layer, direction) whose weights changed (after an optimizer step), instead
// of one ~14 us launch per use of every layer (87 launches / 1.25 ms per compression step, 150 / 2.1 ms per GAN cycle).
// Block b serves job j = last job with prefix[j] <= b; jobs and prefix live in device memory.
template <typename T>
__global__ __launch_bounds__(256) void pack_batch_kernel(const PackJob* __restrict__ jobs, const int* __restrict__ prefix,
                                                         int njobs) {
    int lo = 0, hi = njobs - 1;
    const int b = blockIdx.x;
    while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (prefix[mid] <= b) lo = mid; else hi = mid - 1; }
    const PackJob& J = jobs[lo];
    const int lb = b - prefix[lo];
    if (J.mode == 0) pack_w2_body<T, 0>(J.p, J.w, J.scale, J.sm, J.sc, J.RS, J.MB, lb % J.gx, lb / J.gx);
    else if (J.mode == 1) pack_w2_body<T, 1>(J.p, J.w, J.scale, J.sm, J.sc, J.RS, J.MB, lb % J.gx, lb / J.gx);
    else {
        // generic element-wise pack of phase lb / gx (rare layouts), grid-stride over the phase's elements
        const GcParams& p = J.p;
        const int phi = lb / J.gx, bxx = lb % J.gx;
        const GcPhase& ph = p.ph[phi];
        const long long total = (long long)p.Kpad * ph.ntaps * p.Cpad;
        const float sc_ = J.scale ? *J.scale : 1.f;
        T* dst = (T*)p.wp + ph.wp_off;
        for (long long i = (long long)bxx * 256 + threadIdx.x; i < total; i += (long long)J.gx * 256) {
            const int c = (int)(i % p.Cpad);
            const long long j2 = i / p.Cpad;
            const int t = (int)(j2 % ph.ntaps);
            const int m = (int)(j2 / ph.ntaps);
            float v = 0.f;
            if (m < p.K && c < p.C) {
                const int r = p.tap_r[ph.tap0 + t], s2 = p.tap_s[ph.tap0 + t];
                v = J.w[gc_weight_index(p, m, c, r, s2, J.sm, J.sc, J.sr, J.ss)] * sc_;
            }
            DT<T>::st(dst + gc_wp_index(p, ph, m, t, c), v);
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Reflect fold: dx[y,x] = sum over padded positions that the reflection pad maps onto (y,x)
// (adjoint of ReflectionPad2d, torch reflection_pad2d_backward)
// ---------------------------------------------------------------------------------------------------
template <typename TO>
__global__ void reflect_fold_kernel(const float* __restrict__ src, TO* __restrict__ dst, long long planes,
                                    int H, int W, int pt, int pl, int pb, int pr) {
    // one thread per output element, 32-bit index math (planes*H*W < 2^31 on this path)
    const int Hp = H + pt + pb, Wp = W + pl + pr;
    const unsigned total = (unsigned)planes * (unsigned)(H * W);
    const unsigned hw = (unsigned)(H * W);
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned pc = i / hw;
        const unsigned rem = i - pc * hw;
        const int y = (int)(rem / (unsigned)W);
        const int x = (int)(rem - (unsigned)y * (unsigned)W);
        int ys[3], xs[3], ny = 0, nx = 0;
        ys[ny++] = y + pt;
        if (y >= 1 && y <= pt) ys[ny++] = pt - y;
        if (y <= H - 2 && y >= H - 1 - pb) ys[ny++] = pt + 2 * (H - 1) - y;
        xs[nx++] = x + pl;
        if (x >= 1 && x <= pl) xs[nx++] = pl - x;
        if (x <= W - 2 && x >= W - 1 - pr) xs[nx++] = pl + 2 * (W - 1) - x;
        const float* s = src + (size_t)pc * Hp * Wp;
        float acc = 0.f;
        for (int a = 0; a < ny; ++a)
            for (int b2 = 0; b2 < nx; ++b2) acc += s[ys[a] * Wp + xs[b2]];
        DT<TO>::st(dst + i, acc);
    }
}

// Rim variant: dst already holds the interior term dxp[y+pt][x+pl] (written by the conv epilogue); add the other
// padded positions that reflect onto (y,x) - they all lie on the rim of the padded plane, the only part of src the
// epilogue wrote.  Touches only the few rows/columns next to the border.
template <typename TO>
__global__ void reflect_rim_add_kernel(const float* __restrict__ src, TO* __restrict__ dst, long long planes,
                                       int H, int W, int pt, int pl, int pb, int pr) {
    // Only elements in rows {1..pt} u {H-1-pb..H-2} or columns {1..pl} u {W-1-pr..W-2} receive reflected terms:
    // enumerate exactly those (R full rows, then the nc columns of the remaining rows) instead of the whole plane.
    const int Hp = H + pt + pb, Wp = W + pl + pr;
    const int R = pt + pb, nc = pl + pr;
    // tiny planes (top and bottom bands overlap): enumerate the whole plane instead
    const bool whole = (H < R + 3) || (W < nc + 3);
    const unsigned nrim = whole ? (unsigned)(H * W) : (unsigned)(R * W + (H - R) * nc);
    const unsigned total = (unsigned)planes * nrim;
    for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const unsigned pc = i / nrim;
        const int k = (int)(i - pc * nrim);
        int y, x;
        if (whole) {
            y = k / W; x = k - y * W;
        } else if (k < R * W) {
            const int ri = k / W; x = k - ri * W;
            y = ri < pt ? 1 + ri : H - 1 - pb + (ri - pt);
        } else {
            const int k2 = k - R * W;
            const int yi = k2 / nc, ci = k2 - yi * nc;
            y = yi == 0 ? 0 : (yi <= H - R - 2 ? pt + yi : H - 1);
            x = ci < pl ? 1 + ci : W - 1 - pr + (ci - pl);
        }
        int ys[3], xs[3], ny = 0, nx = 0;
        ys[ny++] = y + pt;
        if (y >= 1 && y <= pt) ys[ny++] = pt - y;
        if (y <= H - 2 && y >= H - 1 - pb) ys[ny++] = pt + 2 * (H - 1) - y;
        xs[nx++] = x + pl;
        if (x >= 1 && x <= pl) xs[nx++] = pl - x;
        if (x <= W - 2 && x >= W - 1 - pr) xs[nx++] = pl + 2 * (W - 1) - x;
        if (ny == 1 && nx == 1) continue;
        const float* s = src + (size_t)pc * Hp * Wp;
        float acc = 0.f;
        for (int a = 0; a < ny; ++a)
            for (int b2 = 0; b2 < nx; ++b2)
                if (a | b2) acc += s[ys[a] * Wp + xs[b2]];
        TO* d = dst + (size_t)pc * H * W + y * W + x;
        DT<TO>::st(d, DT<TO>::ld(d) + acc);
    }
}

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
// Stride-2 bf16 weight gradient, phase-decomposed (round 4): 3x3 / 4x4 windows with pad <= 1, AW % 16 == 0.
//   dw[m][c][r][s] = sum_pixels a[m][u][v] * b[c][2u + r - PT][2v + s - PL]
// The four parity phases b_pq[c][u'][v'] = b[c][2u' + p][2v' + q] turn every tap into a STRIDE-1 shift of one phase plane
// (dy = r - PT = 2 sy + p), so the operands need no transposing gather at all:
//   * a tile = 4 rows x 16 pixels of `a` (64 reduction pixels), in natural NCHW order in LDS ([64 m][64 px], one 16-byte
//     global load per thread and tile); its MFMA fragment is one ds_read_b128 (as in wgrad_pipe_kernel)
//   * the b halo patch (9-10 input rows x 32 pixels + edge pixels, 64 channels) arrives as 16-byte row segments (4-5 per
//     thread and tile, against ~70 two-byte loads of the transposing generic stage), is de-interleaved in registers (2
//     v_perm per 8 pixels) and stored as two 8-byte pieces into the even / odd column phase rows: [c][phase][row][16 (+margin) px];
//     a B fragment of tap (r, s) at reduction slice ks (= tile row ks) is the 16-byte run [8*lhi, 8*lhi + 8) of phase row
//     ks + sy, shifted by one pixel (v_alignbit with one extra dword) for the second tap of a phase
//   * all R x S taps of the (64 m x 64 c) tile accumulate in the workgroup: 8 waves = 4 quadrants x 2 kernel-row sets
//   * both LDS images double-buffered, the next tile prefetched through registers, one barrier per tile
//   * workgroups that read the same b tiles (same channel block and pixel range, all m blocks) are mapped to one XCD
// The generic kernel ran these layers at 140-170 us (34 GFLOP each): 70 two-byte loads per thread and tile, transposed
// 2-byte LDS writes, 517 MB of HBM traffic per launch.
template <int R_, int S_, int PT, int PL>
struct S2Cfg {
    static constexpr int TH = 4;
    static constexpr int par(int d) { return d & 1; }
    static constexpr int shf(int d) { return (d - (d & 1)) / 2; }
    static constexpr int smin(int n, int pad, int q) {
        int m = 99;
        for (int t = 0; t < n; ++t) if (par(t - pad) == q && shf(t - pad) < m) m = shf(t - pad);
        return m;
    }
    static constexpr int smax(int n, int pad, int q) {
        int m = -99;
        for (int t = 0; t < n; ++t) if (par(t - pad) == q && shf(t - pad) > m) m = shf(t - pad);
        return m;
    }
    static constexpr int symin(int q) { return smin(R_, PT, q); }
    static constexpr int symax(int q) { return smax(R_, PT, q); }
    static constexpr int sxmin(int q) { return smin(S_, PL, q); }
    static constexpr int sxmax(int q) { return smax(S_, PL, q); }
    static constexpr int rows(int q) { return TH + symax(q) - symin(q); }
    static constexpr int lm(int q) { return sxmin(q) < 0 ? 16 : 0; }                   // left margin bytes of a phase row
    static constexpr int pitchx(int q) { return lm(q) + 32 + (sxmax(q) > 0 ? 16 : 0); }
    static constexpr int base(int py, int px) {                                        // phases in (0,0) (0,1) (1,0) (1,1) order
        int b = 0;
        for (int i = 0; i < py * 2 + px; ++i) b += rows(i >> 1) * pitchx(i & 1);
        return b;
    }
    static constexpr int cp() {                                                        // bytes per channel: odd multiple of 16 (conflict-free b128 reads)
        const int c = base(1, 1) + rows(1) * pitchx(1);
        return ((c / 16) & 1) ? c : c + 16;
    }
    static constexpr int nry() { return 2 * (TH - 1) + R_; }                           // input rows of a tile
};

// Wave set 1 owns kernel rows 2.. : the same row parities as rows 0.. of set 0, one phase row further down - one code path,
// `wset` enters the addresses only (3x3: set 1 has the single row 2 and skips d = 1).
template <int R_, int S_, int PT, int PL, int NACC>
__device__ __forceinline__ void s2_compute(f32x16_t (&acc)[NACC], const unsigned char* ab, const unsigned char* pb,
                                           int wm, int wn, int l31, int lhi, int wset, int k0, int k1) {
    using G = S2Cfg<R_, S_, PT, PL>;
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    constexpr int CP = G::cp();
    constexpr int SYMIN[2] = {G::symin(0), G::symin(1)}, SXMIN[2] = {G::sxmin(0), G::sxmin(1)}, SXMAX[2] = {G::sxmax(0), G::sxmax(1)};
    constexpr int PITCHX[2] = {G::pitchx(0), G::pitchx(1)}, LM[2] = {G::lm(0), G::lm(1)};
    constexpr int BASE[2][2] = {{G::base(0, 0), G::base(0, 1)}, {G::base(1, 0), G::base(1, 1)}};
    const unsigned char* arow = ab + (wm * 32 + l31) * 144 + lhi * 16;
    const unsigned char* bch = pb + (wn * 32 + l31) * CP + lhi * 16;
#ifndef S2_KU4
#define S2_KU4 4
#endif
#ifndef S2_KU3
#define S2_KU3 4
#endif
    constexpr int KU = R_ == 4 ? S2_KU4 : S2_KU3;    // (slices unrolled together when the whole tile is one call)
#pragma unroll KU
    for (int ks = k0; ks < k1; ++ks) {
        const bf16x8_t a = *(const bf16x8_t*)(arow + ks * 32);
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            if (R_ == 3 && d == 1 && wset) continue;           // (wave-uniform)
            const int dyv = d - PT;
            const int py = dyv & 1, sy = (dyv - py) / 2;
            const int irow = ks + sy - SYMIN[py] + wset;
#pragma unroll
            for (int px = 0; px < 2; ++px) {
                const unsigned char* rowp = bch + BASE[py][px] + irow * PITCHX[px] + LM[px];
                const u32x4_t Rv = *(const u32x4_t*)rowp;
                unsigned Rm = 0, Rp = 0;
                if (SXMIN[px] < 0) Rm = *(const unsigned*)(rowp - 4);
                if (SXMAX[px] > 0) Rp = *(const unsigned*)(rowp + 16);
#pragma unroll
                for (int s = 0; s < S_; ++s) {
                    const int dxv = s - PL;
                    if ((dxv & 1) != px) continue;
                    const int sx = (dxv - px) / 2;
                    u32x4_t f;
                    if (sx == 0) f = Rv;
                    else if (sx < 0) f = u32x4_t{__builtin_amdgcn_alignbit(Rv.x, Rm, 16), __builtin_amdgcn_alignbit(Rv.y, Rv.x, 16),
                                                 __builtin_amdgcn_alignbit(Rv.z, Rv.y, 16), __builtin_amdgcn_alignbit(Rv.w, Rv.z, 16)};
                    else f = u32x4_t{__builtin_amdgcn_alignbit(Rv.y, Rv.x, 16), __builtin_amdgcn_alignbit(Rv.z, Rv.y, 16),
                                     __builtin_amdgcn_alignbit(Rv.w, Rv.z, 16), __builtin_amdgcn_alignbit(Rp, Rv.w, 16)};
                    acc[d * S_ + s] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8_t, f), acc[d * S_ + s], 0, 0, 0);
                }
            }
        }
    }
}

template <int R_, int S_, int PT, int PL>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void wgrad_s2_kernel(const WgParams p) {
    using G = S2Cfg<R_, S_, PT, PL>;
    constexpr int TH = G::TH, NRY = G::nry(), CP = G::cp();
    static_assert(G::sxmin(0) >= 0 && G::sxmin(1) >= -1 && G::sxmax(0) <= 1 && G::sxmax(1) <= 0, "window");
    static_assert(G::symin(0) <= G::symax(0) && G::symin(1) <= G::symax(1), "both row parities need a tap");
    constexpr int SYMIN0 = G::symin(0), SYMIN1 = G::symin(1), PX0 = G::pitchx(0), PX1 = G::pitchx(1), LM0 = G::lm(0), LM1 = G::lm(1);
    constexpr int B00 = G::base(0, 0), B01 = G::base(0, 1), B10 = G::base(1, 0), B11 = G::base(1, 1);
    constexpr int APITCH = 144, ABYTES = 64 * APITCH, PBYTES = 64 * CP;
    constexpr int NBI = 64 * NRY * 4, NB = (NBI + 511) / 512;       // 16-byte b segments per tile / per thread
    constexpr int NEI = 64 * NRY, NE = (NEI + 511) / 512;           // (channel, row) pairs: edge pixels
    constexpr bool HASL = G::sxmin(1) < 0, HASR = G::sxmax(0) > 0;
    constexpr int RSET = 2;                                         // kernel rows of wave set 0 (set 1: rows 2 .. R_-1)
    static_assert(R_ == 3 || R_ == 4, "window");
    constexpr int NACC = RSET * S_;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));
    unsigned char* abuf = smem;
    unsigned char* pbuf = smem + 2 * ABYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wset = wave >> 2;
    const int wm = (wave & 3) >> 1, wn = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;

    const int ctiles = p.Cpad / 64, mtiles = p.Mpad / 64;
    int mblk, grp;
    if (p.xcd_remap) { const int x = blockIdx.x & 7, k = blockIdx.x >> 3; grp = x + 8 * (k / mtiles); mblk = k % mtiles; }
    else { mblk = blockIdx.x % mtiles; grp = blockIdx.x / mtiles; }
    const int m0 = mblk * 64, c0 = (grp % ctiles) * 64, split = grp / ctiles;

    const unsigned aplane = (unsigned)(p.AH * p.AW), bplane = (unsigned)(p.BH * p.BW);
    const bf16_t* asrc = (const bf16_t*)p.a;
    const bf16_t* bsrc = (const bf16_t*)p.b;
    // a piece of this thread: row m = tid >> 3, tile row (tid >> 1) & 3, 8-pixel half tid & 1
    const int a_m = tid >> 3, a_ty = (tid >> 1) & 3, a_h = tid & 1;
    const unsigned a_rel = (unsigned)(m0 + a_m) * aplane + (unsigned)(a_ty * p.AW + a_h * 8);
    const int a_dst = a_m * APITCH + a_ty * 32 + a_h * 16;
    // b segments of this thread: segment e = tid + 512 i -> (channel, input row, 16-byte piece); the two LDS destinations
    // (even / odd pixel phase) are tile-independent and kept packed (16 + 16 bits), the rest is re-derived per tile
    unsigned b_dst[NB], b_rel[NB], b_okc = 0;
    int b_r[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int e = tid + 512 * i;
        const int sg = e & 3, rc = e >> 2;
        const int c = rc / NRY, r = rc - c * NRY;
        const int t = r - PT, py = t & 1, hy = (t - py) >> 1;
        const int irow = hy - (py ? SYMIN1 : SYMIN0);
        const int cb = c * CP + 8 * sg;
        const int d0 = cb + (py ? B10 : B00) + irow * PX0 + LM0, d1 = cb + (py ? B11 : B01) + irow * PX1 + LM1;
        b_dst[i] = e < NBI ? (unsigned)d0 | ((unsigned)d1 << 16) : 0u;
        b_r[i] = r;
        b_rel[i] = (unsigned)(c0 + c) * bplane + (unsigned)(8 * sg);
        b_okc |= ((e < NBI && c0 + c < p.C) ? 1u : 0u) << i;
    }
    static_assert(PBYTES <= 65536, "packed LDS offsets");
    unsigned e_dst[NE], e_rel[NE], e_okc = 0;
    int e_r[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = tid + 512 * i;
        const int c = e / NRY, r = e - c * NRY;
        const int t = r - PT, py = t & 1, hy = (t - py) >> 1;
        const int irow = hy - (py ? SYMIN1 : SYMIN0);
        const int dl = c * CP + (py ? B11 : B01) + irow * PX1 + LM1 - 2, dr = c * CP + (py ? B10 : B00) + irow * PX0 + LM0 + 32;
        e_dst[i] = e < NEI ? (unsigned)(dl & 0xffff) | ((unsigned)dr << 16) : 0u;
        e_r[i] = r;
        e_rel[i] = (unsigned)(c0 + c) * bplane;
        e_okc |= ((e < NEI && c0 + c < p.C) ? 1u : 0u) << i;
    }

    f32x16_t acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int tile_lo = split * p.tiles_per_split;
    int tile_hi = tile_lo + p.tiles_per_split;
    if (tile_hi > p.ntiles) tile_hi = p.ntiles;

    u32x4_t areg; bool aok = false;
    u32x4_t breg[NB]; unsigned bokm = 0;
    unsigned short el[NE], er[NE]; unsigned eokm = 0;

#define S2_LOAD_HEAD(tile_)                                                                                         \
        const int tx_t = (tile_) % p.tiles_x;                                                                       \
        const int ty_t = ((tile_) / p.tiles_x) % p.tiles_y;                                                         \
        const int n0 = (tile_) / (p.tiles_x * p.tiles_y);                                                           \
        const int u0 = ty_t * TH, v0 = tx_t * 16;                                                                   \
        /* rows: iy = 2 u0 - PT + r, one reflection = abs, then min(iy, 2 (BH - 1) - iy); a row outside ends up negative */ \
        const unsigned bbase = (unsigned)(n0 * p.C) * bplane + (unsigned)(2 * v0);                                  \
        const int ybase = 2 * u0 - PT, ytop = 2 * (p.BH - 1);                                                       \
        const bool refl = p.bmode == PAD_REFLECT;                                                                   \
        bokm = 0; eokm = 0;
#define S2_LOAD_A()                                                                                                 \
    do {                                                                                                            \
        aok = (m0 + a_m < p.M) && (u0 + a_ty < p.AH);                                                               \
        areg = *(const u32x4_t*)(asrc + (aok ? (unsigned)(n0 * p.M) * aplane + (unsigned)(u0 * p.AW + v0) + a_rel : 0u)); \
    } while (0)
#define S2_LOAD_B(i0_, i1_)                                                                                         \
    do {                                                                                                            \
        _Pragma("unroll") for (int i = (i0_); i < ((i1_) < NB ? (i1_) : NB); ++i) {                                 \
            int iy = ybase + b_r[i];                                                                                \
            if (refl) { iy = iy < 0 ? -iy : iy; const int m_ = ytop - iy; iy = iy < m_ ? iy : m_; }                 \
            const bool ok_ = ((b_okc >> i) & 1u) && (unsigned)iy < (unsigned)p.BH;                                  \
            bokm |= (ok_ ? 1u : 0u) << i;                                                                           \
            breg[i] = *(const u32x4_t*)(bsrc + (ok_ ? bbase + b_rel[i] + (unsigned)(iy * p.BW) : 0u));             \
        }                                                                                                           \
    } while (0)
#define S2_LOAD_E()                                                                                                 \
    do {                                                                                                            \
        int xl = 2 * v0 - 1, xr = 2 * v0 + 32;                                                                      \
        if (refl) { xl = xl < 0 ? -xl : xl; xr = xr > p.BW - 1 ? 2 * (p.BW - 1) - xr : xr; }                        \
        const bool okl_ = (unsigned)xl < (unsigned)p.BW, okr_ = (unsigned)xr < (unsigned)p.BW;                      \
        _Pragma("unroll") for (int i = 0; i < NE; ++i) {                                                            \
            int iy = ybase + e_r[i];                                                                                \
            if (refl) { iy = iy < 0 ? -iy : iy; const int m_ = ytop - iy; iy = iy < m_ ? iy : m_; }                 \
            const bool oky = ((e_okc >> i) & 1u) && (unsigned)iy < (unsigned)p.BH;                                  \
            const unsigned rb_ = (unsigned)(n0 * p.C) * bplane + e_rel[i] + (unsigned)(iy * p.BW);                  \
            if (HASL) { const bool ok_ = oky && okl_; eokm |= (ok_ ? 1u : 0u) << (2 * i);                           \
                        el[i] = bsrc[ok_ ? rb_ + (unsigned)xl : 0u]; }                                              \
            if (HASR) { const bool ok_ = oky && okr_; eokm |= (ok_ ? 1u : 0u) << (2 * i + 1);                       \
                        er[i] = bsrc[ok_ ? rb_ + (unsigned)xr : 0u]; }                                              \
        }                                                                                                           \
    } while (0)
#define S2_LOAD(tile_) do { S2_LOAD_HEAD(tile_) S2_LOAD_A(); S2_LOAD_B(0, NB); S2_LOAD_E(); } while (0)
#define S2_STORE(ab_, pb_)                                                                                          \
    do {                                                                                                            \
        u32x4_t av = areg;                                                                                          \
        if (!aok) { av.x = 0; av.y = 0; av.z = 0; av.w = 0; }                                                       \
        *(u32x4_t*)((ab_) + a_dst) = av;                                                                            \
        _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                                            \
            u32x4_t v = breg[i];                                                                                    \
            if (!((bokm >> i) & 1u)) { v.x = 0; v.y = 0; v.z = 0; v.w = 0; }                                        \
            const u32x2_t ev = {__builtin_amdgcn_perm(v.y, v.x, 0x05040100u), __builtin_amdgcn_perm(v.w, v.z, 0x05040100u)}; \
            const u32x2_t od = {__builtin_amdgcn_perm(v.y, v.x, 0x07060302u), __builtin_amdgcn_perm(v.w, v.z, 0x07060302u)}; \
            if (tid + 512 * i < NBI) {                                                                              \
                *(u32x2_t*)((pb_) + (b_dst[i] & 0xffffu)) = ev;                                                     \
                *(u32x2_t*)((pb_) + (b_dst[i] >> 16)) = od;                                                         \
            }                                                                                                       \
        }                                                                                                           \
        _Pragma("unroll") for (int i = 0; i < NE; ++i) {                                                            \
            if (tid + 512 * i < NEI) {                                                                              \
                if (HASL) *(unsigned short*)((pb_) + (e_dst[i] & 0xffffu)) = ((eokm >> (2 * i)) & 1u) ? el[i] : (unsigned short)0; \
                if (HASR) *(unsigned short*)((pb_) + (e_dst[i] >> 16)) = ((eokm >> (2 * i + 1)) & 1u) ? er[i] : (unsigned short)0; \
            }                                                                                                       \
        }                                                                                                           \
    } while (0)

    if (tile_lo < tile_hi) {
        const int tile_last = tile_hi - 1;
        S2_LOAD(tile_lo);
        S2_STORE(abuf, pbuf);
        { const int t1 = tile_lo + 1 < tile_hi ? tile_lo + 1 : tile_last; S2_LOAD(t1); }
        for (int tile = tile_lo; tile < tile_hi; ++tile) {
            const int cur = (tile - tile_lo) & 1;
            const unsigned char* ab = abuf + cur * ABYTES;
            const unsigned char* pb = pbuf + cur * PBYTES;
            __syncthreads();
            S2_STORE(abuf + (cur ^ 1) * ABYTES, pbuf + (cur ^ 1) * PBYTES);
#ifndef S2_ILV
#define S2_ILV 1
#endif
            const int t2 = tile + 2 < tile_hi ? tile + 2 : tile_last;
#if S2_ILV
            // the address arithmetic and requests of tile t+2 in three pieces between the reduction slices of tile t: the
            // wave's VALU work runs while its MFMAs execute (all 8 waves leave the barrier together, so without this
            // every SIMD first sits through two waves' staging code and only then starts its matrix pipe)
            S2_LOAD_HEAD(t2)
            s2_compute<R_, S_, PT, PL, NACC>(acc, ab, pb, wm, wn, l31, lhi, wset, 0, 1);
            S2_LOAD_A(); S2_LOAD_B(0, 2);
            s2_compute<R_, S_, PT, PL, NACC>(acc, ab, pb, wm, wn, l31, lhi, wset, 1, 2);
            S2_LOAD_B(2, 4);
            s2_compute<R_, S_, PT, PL, NACC>(acc, ab, pb, wm, wn, l31, lhi, wset, 2, 3);
            S2_LOAD_B(4, NB); S2_LOAD_E();
            s2_compute<R_, S_, PT, PL, NACC>(acc, ab, pb, wm, wn, l31, lhi, wset, 3, 4);
#else
            S2_LOAD(t2);
            s2_compute<R_, S_, PT, PL, NACC>(acc, ab, pb, wm, wn, l31, lhi, wset, 0, 4);
#endif
        }
    }
#undef S2_STORE
#undef S2_LOAD
#undef S2_LOAD_E
#undef S2_LOAD_B
#undef S2_LOAD_A
#undef S2_LOAD_HEAD

    const int r0 = wset ? RSET : 0, nr = wset ? R_ - RSET : RSET;
#pragma unroll
    for (int ta = 0; ta < NACC; ++ta) {
        if (ta < nr * S_) {
            const int tg = r0 * S_ + ta;
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
// Stride-1 3x3 (pad 1) bf16 weight gradient on natural-order operands (late round 4; the residual-block layers and the other
// 16 x 16-plane 3x3 layers).  wgrad_pipe_kernel keeps the input-side operand as a TRANSPOSED halo patch ([pixel][channel], filled
// by two-byte loads) and reads its fragments with ds_read_b64_tr_b16: 10 KB of LDS reads per wave and 16-deep slice - it is
// LDS-read bound at 0.26 of the MFMA peak.  A weight gradient reduces over PIXELS, so both operands are already in MFMA order in
// NCHW (see wgrad_s2_kernel): here
//   * a tile = 8 rows x 16 pixels; a [64 m][128 px] and the b patch [64 c][10 rows][16 px] in natural order, 16-byte loads only
//     (2 + 3 per thread and tile); the two edge pixels of a patch row (columns -1 and 16, padding rule applied) live in a
//     separate [row][channel] array, one dword {left, right} per (row, channel): a conflict-free 4-byte read
//   * the three taps of a kernel row at reduction slice ks (= tile row ks) are ONE ds_read_b128 of patch row ks + r plus that
//     edge dword: the run shifted by -1 / +1 pixel is built with four v_alignbit from the lane's own four dwords and ONE dword
//     of the other half-wave (v_permlane32_swap: lanes 32-63 hold pixels 8-15 of the same channel) or the edge pixel
//   -> 3.5 KB of LDS reads per wave and slice instead of 10
//   * 8 waves = 4 quadrants x 2 kernel-row sets (rows 0-1 | row 2), one code path (`wset` enters the addresses only)
//   * both operand images double-buffered, register prefetch one tile ahead in pieces between the slices, one barrier per tile
//   * epilogue as wgrad_pipe_kernel (tile transposed through LDS, contiguous 64 c x 9 tap runs per m row)
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void wgrad_s1_kernel(const WgParams p) {
    constexpr int TH = 8, NPR = TH + 2;
    constexpr int APITCH = 272, ABYTES = 64 * APITCH;
    constexpr int CP = NPR * 32 + 16, PBYTES = 64 * CP;
    constexpr int MBYTES = NPR * 64 * 4;
    constexpr int NAI = 64 * 16, NA = NAI / 512;
    constexpr int NBI = 64 * NPR * 2, NB = (NBI + 511) / 512;
    constexpr int NEI = 64 * NPR, NE = (NEI + 511) / 512;
    constexpr int NACC = 6, NTAP = 9;
    static_assert((CP / 16) & 1, "odd multiple of 16 bytes");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    unsigned char* abuf = smem;
    unsigned char* pbuf = abuf + 2 * ABYTES;
    unsigned char* mbuf = pbuf + 2 * PBYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wset = wave >> 2;
    const int wm = (wave & 3) >> 1, wn = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;

    const int ctiles = p.Cpad / 64, mtiles = p.Mpad / 64;
    int mblk, grp;
    if (p.xcd_remap) { const int x = blockIdx.x & 7, k = blockIdx.x >> 3; grp = x + 8 * (k / mtiles); mblk = k % mtiles; }
    else { mblk = blockIdx.x % mtiles; grp = blockIdx.x / mtiles; }
    const int m0 = mblk * 64, c0 = (grp % ctiles) * 64, split = grp / ctiles;

    const unsigned aplane = (unsigned)(p.AH * p.AW), bplane = (unsigned)(p.BH * p.BW);
    const bf16_t* asrc = (const bf16_t*)p.a;
    const bf16_t* bsrc = (const bf16_t*)p.b;
    const int PT = -(int)p.tap_dy[0];
    // a pieces: piece = tid + 512 i -> (row m = piece >> 4, 8-pixel segment piece & 15 = tile row * 2 + half)
    unsigned a_rel[NA]; int a_dst[NA], a_m[NA], a_ty[NA];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int piece = tid + 512 * i;
        const int m = piece >> 4, seg = piece & 15;
        a_m[i] = m; a_ty[i] = seg >> 1;
        a_rel[i] = (unsigned)(m0 + m) * aplane + (unsigned)((seg >> 1) * p.AW + (seg & 1) * 8);
        a_dst[i] = m * APITCH + seg * 16;
    }
    // b segments: e = tid + 512 i -> (channel, patch row, half)
    unsigned b_rel[NB]; int b_dst[NB], b_r[NB]; unsigned b_okc = 0;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int e = tid + 512 * i;
        const int sg = e & 1, rc = e >> 1;
        const int c = rc / NPR, r = rc - c * NPR;
        b_r[i] = r;
        b_rel[i] = (unsigned)(c0 + c) * bplane + (unsigned)(8 * sg);
        b_dst[i] = c * CP + r * 32 + 16 * sg;
        b_okc |= ((e < NBI && c0 + c < p.C) ? 1u : 0u) << i;
    }
    unsigned e_rel[NE]; int e_dst[NE], e_r[NE]; unsigned e_okc = 0;
#pragma unroll
    for (int i = 0; i < NE; ++i) {
        const int e = tid + 512 * i;
        const int c = e / NPR, r = e - c * NPR;
        e_r[i] = r;
        e_rel[i] = (unsigned)(c0 + c) * bplane;
        e_dst[i] = (r * 64 + c) * 4;
        e_okc |= ((e < NEI && c0 + c < p.C) ? 1u : 0u) << i;
    }

    f32x16_t acc[NACC];
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    const int tile_lo = split * p.tiles_per_split;
    int tile_hi = tile_lo + p.tiles_per_split;
    if (tile_hi > p.ntiles) tile_hi = p.ntiles;

    u32x4_t areg[NA]; unsigned aokm = 0;
    u32x4_t breg[NB]; unsigned bokm = 0;
    unsigned short el[NE], er[NE]; unsigned eokm = 0;

#define S1_LOAD_HEAD(tile_)                                                                                         \
        const int tx_t = (tile_) % p.tiles_x;                                                                       \
        const int ty_t = ((tile_) / p.tiles_x) % p.tiles_y;                                                         \
        const int n0 = (tile_) / (p.tiles_x * p.tiles_y);                                                           \
        const int u0 = ty_t * TH, v0 = tx_t * 16;                                                                   \
        const unsigned bbase = (unsigned)(n0 * p.C) * bplane + (unsigned)v0;                                        \
        const int ybase = u0 - PT, ytop = 2 * (p.BH - 1);                                                           \
        const bool refl = p.bmode == PAD_REFLECT;                                                                   \
        aokm = 0; bokm = 0; eokm = 0;
#define S1_LOAD_A()                                                                                                 \
    do {                                                                                                            \
        const unsigned tb_ = (unsigned)(n0 * p.M) * aplane + (unsigned)(u0 * p.AW + v0);                            \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                                            \
            const bool ok_ = (m0 + a_m[i] < p.M) && (u0 + a_ty[i] < p.AH);                                          \
            aokm |= (ok_ ? 1u : 0u) << i;                                                                           \
            areg[i] = *(const u32x4_t*)(asrc + (ok_ ? tb_ + a_rel[i] : 0u));                                        \
        }                                                                                                           \
    } while (0)
#define S1_LOAD_B(i0_, i1_)                                                                                         \
    do {                                                                                                            \
        _Pragma("unroll") for (int i = (i0_); i < ((i1_) < NB ? (i1_) : NB); ++i) {                                 \
            int iy = ybase + b_r[i];                                                                                \
            if (refl) { iy = iy < 0 ? -iy : iy; const int m_ = ytop - iy; iy = iy < m_ ? iy : m_; }                 \
            const bool ok_ = ((b_okc >> i) & 1u) && (unsigned)iy < (unsigned)p.BH;                                  \
            bokm |= (ok_ ? 1u : 0u) << i;                                                                           \
            breg[i] = *(const u32x4_t*)(bsrc + (ok_ ? bbase + b_rel[i] + (unsigned)(iy * p.BW) : 0u));             \
        }                                                                                                           \
    } while (0)
#define S1_LOAD_E()                                                                                                 \
    do {                                                                                                            \
        int xl = v0 - 1, xr = v0 + 16;                                                                              \
        if (refl) { xl = xl < 0 ? -xl : xl; xr = xr > p.BW - 1 ? 2 * (p.BW - 1) - xr : xr; }                        \
        const bool okl_ = (unsigned)xl < (unsigned)p.BW, okr_ = (unsigned)xr < (unsigned)p.BW;                      \
        _Pragma("unroll") for (int i = 0; i < NE; ++i) {                                                            \
            int iy = ybase + e_r[i];                                                                                \
            if (refl) { iy = iy < 0 ? -iy : iy; const int m_ = ytop - iy; iy = iy < m_ ? iy : m_; }                 \
            const bool oky = ((e_okc >> i) & 1u) && (unsigned)iy < (unsigned)p.BH;                                  \
            const unsigned rb_ = (unsigned)(n0 * p.C) * bplane + e_rel[i] + (unsigned)(iy * p.BW);                  \
            const bool ol_ = oky && okl_, or_ = oky && okr_;                                                        \
            eokm |= ((ol_ ? 1u : 0u) << (2 * i)) | ((or_ ? 1u : 0u) << (2 * i + 1));                                \
            el[i] = bsrc[ol_ ? rb_ + (unsigned)xl : 0u];                                                            \
            er[i] = bsrc[or_ ? rb_ + (unsigned)xr : 0u];                                                            \
        }                                                                                                           \
    } while (0)
#define S1_STORE(ab_, pb_, mb_)                                                                                     \
    do {                                                                                                            \
        _Pragma("unroll") for (int i = 0; i < NA; ++i) {                                                            \
            u32x4_t v = areg[i];                                                                                    \
            if (!((aokm >> i) & 1u)) { v.x = 0; v.y = 0; v.z = 0; v.w = 0; }                                        \
            *(u32x4_t*)((ab_) + a_dst[i]) = v;                                                                      \
        }                                                                                                           \
        _Pragma("unroll") for (int i = 0; i < NB; ++i) {                                                            \
            u32x4_t v = breg[i];                                                                                    \
            if (!((bokm >> i) & 1u)) { v.x = 0; v.y = 0; v.z = 0; v.w = 0; }                                        \
            if (tid + 512 * i < NBI) *(u32x4_t*)((pb_) + b_dst[i]) = v;                                             \
        }                                                                                                           \
        _Pragma("unroll") for (int i = 0; i < NE; ++i) {                                                            \
            const unsigned l_ = ((eokm >> (2 * i)) & 1u) ? (unsigned)el[i] : 0u;                                    \
            const unsigned r_ = ((eokm >> (2 * i + 1)) & 1u) ? (unsigned)er[i] : 0u;                                \
            if (tid + 512 * i < NEI) *(unsigned*)((mb_) + e_dst[i]) = l_ | (r_ << 16);                              \
        }                                                                                                           \
    } while (0)
    // One tile of a wave set: kernel rows [D0, D0 + ND) at all 8 reduction slices, as ONE straight-line block per wave set (a
    // per-row `if (wset)` split every row into its own basic block: LDS read -> wait -> MFMA with nothing scheduled across).
    // The operands of unit u + 1 (patch row run + edge dword, A fragment at a slice change) are read before the MFMAs of unit u;
    // the three request pieces of tile t + 2 sit after slices 1, 3 and 5.
#define S1_READ_B(RV_, M_, u_)                                                                                      \
    do {                                                                                                            \
        const int irow_ = (u_) / (ND_) + (D0_) + (u_) % (ND_);                                                      \
        RV_ = *(const u32x4_t*)(bch_ + irow_ * 32);                                                                 \
        M_ = *(const unsigned*)(mch_ + irow_ * 256);                                                                \
    } while (0)
#define S1_TILE(ab_, pb_, mb_, D0__, ND__)                                                                          \
    do {                                                                                                            \
        constexpr int D0_ = (D0__), ND_ = (ND__), NU_ = 8 * ND_;                                                    \
        const unsigned char* arow_ = (ab_) + (wm * 32 + l31) * APITCH + lhi * 16;                                   \
        const unsigned char* bch_ = (pb_) + (wn * 32 + l31) * CP + lhi * 16;                                        \
        const unsigned char* mch_ = (mb_) + (wn * 32 + l31) * 4;                                                    \
        u32x4_t Rv, Rn; unsigned Mv, Mn;                                                                            \
        bf16x8_t a = *(const bf16x8_t*)(arow_), an = a;                                                             \
        S1_READ_B(Rv, Mv, 0);                                                                                       \
        _Pragma("unroll") for (int u = 0; u < NU_; ++u) {                                                           \
            const int ks = u / ND_, d = u % ND_;                                                                    \
            if (u + 1 < NU_) {                                                                                      \
                S1_READ_B(Rn, Mn, u + 1);                                                                           \
                if ((u + 1) % ND_ == 0) an = *(const bf16x8_t*)(arow_ + ((u + 1) / ND_) * 32);                      \
            }                                                                                                       \
            /* Rm: the dword before the run - lanes 32-63: the last dword of lanes 0-31; lanes 0-31: the left edge pixel */ \
            const unsigned Rm = __builtin_amdgcn_permlane32_swap(Mv << 16, Rv.w, false, false)[0];                  \
            /* Rp: the dword after the run - lanes 0-31: the first dword of lanes 32-63; lanes 32-63: the right edge pixel */ \
            const unsigned Rp = __builtin_amdgcn_permlane32_swap(Rv.x, Mv >> 16, false, false)[1];                  \
            const unsigned s1_ = __builtin_amdgcn_alignbit(Rv.y, Rv.x, 16), s2_ = __builtin_amdgcn_alignbit(Rv.z, Rv.y, 16), \
                           s3_ = __builtin_amdgcn_alignbit(Rv.w, Rv.z, 16);                                         \
            const u32x4_t fm = {__builtin_amdgcn_alignbit(Rv.x, Rm, 16), s1_, s2_, s3_};                            \
            const u32x4_t fp = {s1_, s2_, s3_, __builtin_amdgcn_alignbit(Rp, Rv.w, 16)};                            \
            acc[d * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8_t, Rv), acc[d * 3 + 1], 0, 0, 0); \
            acc[d * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8_t, fm), acc[d * 3 + 0], 0, 0, 0); \
            acc[d * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8_t, fp), acc[d * 3 + 2], 0, 0, 0); \
            if (u == 2 * ND_ - 1) S1_LOAD_A();                                                                      \
            if (u == 4 * ND_ - 1) S1_LOAD_B(0, 2);                                                                  \
            if (u == 6 * ND_ - 1) { S1_LOAD_B(2, NB); S1_LOAD_E(); }                                                \
            Rv = Rn; Mv = Mn;                                                                                       \
            if ((u + 1) % ND_ == 0) a = an;                                                                         \
            (void)ks;                                                                                               \
        }                                                                                                           \
    } while (0)

    if (tile_lo < tile_hi) {
        const int tile_last = tile_hi - 1;
        { S1_LOAD_HEAD(tile_lo) S1_LOAD_A(); S1_LOAD_B(0, NB); S1_LOAD_E(); }
        S1_STORE(abuf, pbuf, mbuf);
        { const int t1 = tile_lo + 1 < tile_hi ? tile_lo + 1 : tile_last; S1_LOAD_HEAD(t1) S1_LOAD_A(); S1_LOAD_B(0, NB); S1_LOAD_E(); }
        for (int tile = tile_lo; tile < tile_hi; ++tile) {
            const int cur = (tile - tile_lo) & 1;
            const unsigned char* ab = abuf + cur * ABYTES;
            const unsigned char* pb = pbuf + cur * PBYTES;
            const unsigned char* mb = mbuf + cur * MBYTES;
            __syncthreads();
            S1_STORE(abuf + (cur ^ 1) * ABYTES, pbuf + (cur ^ 1) * PBYTES, mbuf + (cur ^ 1) * MBYTES);
            const int t2 = tile + 2 < tile_hi ? tile + 2 : tile_last;
            S1_LOAD_HEAD(t2)
            if (wset == 0) S1_TILE(ab, pb, mb, 0, 2);
            else S1_TILE(ab, pb, mb, 2, 1);
        }
    }
#undef S1_TILE
#undef S1_READ_B
#undef S1_STORE
#undef S1_LOAD_E
#undef S1_LOAD_B
#undef S1_LOAD_A
#undef S1_LOAD_HEAD

    const int tbase = wset ? 6 : 0;                         // first tap of this wave's accumulators
    if (p.direct && p.sc == NTAP && p.ss == 1) {
        // each m row leaves as one contiguous run of 64 c x 9 taps (see wgrad_pipe_kernel)
        constexpr int RP = 32 * NTAP + 1;
        float* stg = (float*)smem + (size_t)(wave & 3) * 16 * RP;
        __syncthreads();
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
            for (int t = 0; t < NACC; ++t)
#pragma unroll
                for (int rr = 0; rr < 8; ++rr) {
                    const int r = h * 8 + rr;
                    const int rowl = (r & 3) + 8 * ((r >> 2) & 1) + 4 * lhi;
                    if (tbase + t < NTAP) stg[rowl * RP + l31 * NTAP + tbase + t] = acc[t][r];
                }
            __syncthreads();
            for (int rowl = wset * 8; rowl < wset * 8 + 8; ++rowl) {
                const int m = m0 + wm * 32 + h * 16 + rowl;
                if (m >= p.M) break;
                float* drow = p.dw + (long long)m * p.sm + (long long)(c0 + wn * 32) * p.sc;
                int nvalid = (p.C - (c0 + wn * 32)) * NTAP; if (nvalid > 32 * NTAP) nvalid = 32 * NTAP;
                for (int j = lane; j < nvalid; j += 64) {
                    const float v = stg[rowl * RP + j];
                    if (p.accumulate) drow[j] += v; else drow[j] = v;
                }
            }
            __syncthreads();
        }
        return;
    }
#pragma unroll
    for (int ta = 0; ta < NACC; ++ta) {
        const int tg = tbase + ta;
        if (tg < NTAP) {
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
        stage_T<T, DWR, PITCH>(at, p.a, p.a_f32, p.N, p.M, p.a_h, p.a_w, p.a_bmode, n0, p.NI, u0 + p.a_y0,
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

// ---------------------------------------------------------------------------------------------------
// Extended gradient for the gather-form reflect data gradient (gconv_sp9_kernel RFX): E[pc][e][f], e in [0, H+2),
// f in [0, W+2); row sets {0,2}, {e-1}, {H-3,H-1}; column sets likewise; E = sum over the row set x column set.
// ---------------------------------------------------------------------------------------------------
template <typename TI>
__global__ void reflect_extend_kernel(const TI* __restrict__ dy, bf16_t* __restrict__ E, unsigned planes, int H, int W) {
    const int He = H + 2, We = W + 2;
    const unsigned total = planes * (unsigned)(He * We);
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const unsigned pc = idx / (unsigned)(He * We);
        const int rem = (int)(idx - pc * (unsigned)(He * We));
        const int e = rem / We, f = rem - e * We;
        const int y0 = e == 0 ? 0 : (e == H + 1 ? H - 3 : e - 1), y1 = e == 0 ? 2 : (e == H + 1 ? H - 1 : y0);
        const int x0 = f == 0 ? 0 : (f == W + 1 ? W - 3 : f - 1), x1 = f == 0 ? 2 : (f == W + 1 ? W - 1 : x0);
        const TI* s = dy + (size_t)pc * H * W;
        // all four loads unconditional (duplicates where a set has one element), masked in the sum
        const float a = DT<TI>::ld(s + y0 * W + x0), b = DT<TI>::ld(s + y0 * W + x1);
        const float c = DT<TI>::ld(s + y1 * W + x0), d = DT<TI>::ld(s + y1 * W + x1);
        float v = a;
        if (x1 != x0) v += b;
        if (y1 != y0) { v += c; if (x1 != x0) v += d; }
        E[idx] = f2bf(v);
    }
}

// Row form of reflect_extend_kernel for bf16 planes with W % 8 == 0 (the 16x16 residual-block planes): one thread builds
// one row of E from one or two rows of dY read as 16-byte pieces (the element form is 4 two-byte loads, 2 divisions and a
// two-byte store per element: 16 us for a 10 MB tensor).
template <int WMAX>
__global__ void reflect_extend_rows_kernel(const bf16_t* __restrict__ dy, bf16_t* __restrict__ E, unsigned planes, int H, int W) {
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    const int He = H + 2, We = W + 2;
    const unsigned nrows = planes * (unsigned)He;
    for (unsigned row = blockIdx.x * blockDim.x + threadIdx.x; row < nrows; row += gridDim.x * blockDim.x) {
        const unsigned pc = row / (unsigned)He;
        const int e = (int)(row - pc * (unsigned)He);
        const int y0 = e == 0 ? 0 : (e == H + 1 ? H - 3 : e - 1), y1 = e == 0 ? 2 : (e == H + 1 ? H - 1 : y0);
        const bf16_t* s0 = dy + ((size_t)pc * H + y0) * W;
        const bf16_t* s1 = dy + ((size_t)pc * H + y1) * W;
        float v[WMAX];
#pragma unroll
        for (int j = 0; j < WMAX / 8; ++j) {
            if (j * 8 < W) {
                const u32x4_t a = *(const u32x4_t*)(s0 + j * 8), b = *(const u32x4_t*)(s1 + j * 8);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float a0 = bf2f((bf16_t)(a[k] & 0xffffu)), a1 = bf2f((bf16_t)(a[k] >> 16));
                    const float b0 = bf2f((bf16_t)(b[k] & 0xffffu)), b1 = bf2f((bf16_t)(b[k] >> 16));
                    v[j * 8 + 2 * k] = y1 != y0 ? a0 + b0 : a0;
                    v[j * 8 + 2 * k + 1] = y1 != y0 ? a1 + b1 : a1;
                }
            }
        }
        // E row: [v0 + v2, v0 .. v(W-1), v(W-3) + v(W-1)]; We = W + 2 is even: stored as dwords (rows are 4-byte aligned)
        unsigned* d = (unsigned*)(E + (size_t)row * We);
        bf16_t o[WMAX + 2];
        o[0] = f2bf(v[0] + v[2]);
#pragma unroll
        for (int x = 0; x < WMAX; ++x) if (x < W) o[x + 1] = f2bf(v[x]);
#pragma unroll
        for (int x = 0; x < WMAX; ++x) if (x == W - 1) o[x + 2] = f2bf(v[x - 2] + v[x]);
#pragma unroll
        for (int x = 0; x < (WMAX + 2) / 2; ++x) if (2 * x < We) d[x] = (unsigned)o[2 * x] | ((unsigned)o[2 * x + 1] << 16);
    }
}

// ---------------------------------------------------------------------------------------------------
// Optional in-library profiler: HIP event pairs around every GEMM-class launch (on the launch stream), keyed by
// kernel kind, with the algorithmic FLOPs of each launch.  Used by bench.py for the live roofline figure.
// ---------------------------------------------------------------------------------------------------
#define PROF_MAX 16384
#define PROF_MAXKINDS 32
#define PROF_NAMELEN 64
// kinds are kernel functions, registered by name on first use (the table only grows while profiling is on)
static bool g_prof_on = false;
static int g_prof_n = 0;
static int g_prof_nk = 0;
static char g_prof_kname[PROF_MAXKINDS][PROF_NAMELEN];
static hipEvent_t g_prof_ev[PROF_MAX][2];
static bool g_prof_ev_made[PROF_MAX];
static int g_prof_kind[PROF_MAX];
static double g_prof_flops[PROF_MAX];
static char g_prof_tag[PROF_MAX][112];     // launch shape, printed per launch when HIFIC_PROF_DUMP=1

static int prof_open(const char* kname, double flops, hipStream_t st, const char* tag = "") {
    if (!g_prof_on || g_prof_n >= PROF_MAX) return -1;
    int k = 0;
    while (k < g_prof_nk && strcmp(g_prof_kname[k], kname) != 0) ++k;
    if (k == g_prof_nk) {
        if (g_prof_nk >= PROF_MAXKINDS) return -1;
        strncpy(g_prof_kname[k], kname, PROF_NAMELEN - 1); g_prof_kname[k][PROF_NAMELEN - 1] = 0;
        ++g_prof_nk;
    }
    const int i = g_prof_n++;
    if (!g_prof_ev_made[i]) {
        hipEventCreate(&g_prof_ev[i][0]); hipEventCreate(&g_prof_ev[i][1]); g_prof_ev_made[i] = true;
    }
    g_prof_kind[i] = k; g_prof_flops[i] = flops;
    strncpy(g_prof_tag[i], tag, sizeof(g_prof_tag[i]) - 1); g_prof_tag[i][sizeof(g_prof_tag[i]) - 1] = 0;
    hipEventRecord(g_prof_ev[i][0], st);
    return i;
}
static void prof_close(int i, hipStream_t st) { if (i >= 0) hipEventRecord(g_prof_ev[i][1], st); }
int gc_prof_open(const char* kname, double flops, hipStream_t st, const char* tag) { return prof_open(kname, flops, st, tag); }
void gc_prof_close(int slot, hipStream_t st) { prof_close(slot, st); }

extern "C" int hific_prof_begin(void) { g_prof_on = true; g_prof_n = 0; g_prof_nk = 0; return HIFIC_OK; }
// Synchronises the recorded events.  Fills, for up to max_kinds kernel functions: total ms, total algorithmic FLOPs,
// launch count and the kernel name (names: max_kinds x 64 chars).  Returns the number of kinds (<0: error code).
extern "C" int hific_prof_end(int max_kinds, double* ms, double* flops, int* count, char* names) {
    g_prof_on = false;
    const char* dump_e = getenv("HIFIC_PROF_DUMP");
    const bool dump = dump_e && atoi(dump_e) != 0;
    const int nk = g_prof_nk < max_kinds ? g_prof_nk : max_kinds;
    for (int k = 0; k < nk; ++k) {
        ms[k] = 0; flops[k] = 0; count[k] = 0;
        strncpy(names + (size_t)k * PROF_NAMELEN, g_prof_kname[k], PROF_NAMELEN);
    }
    for (int i = 0; i < g_prof_n; ++i) {
        if (hipEventSynchronize(g_prof_ev[i][1]) != hipSuccess) return HIFIC_ERR_LAUNCH;
        float t = 0.f;
        hipEventElapsedTime(&t, g_prof_ev[i][0], g_prof_ev[i][1]);
        const int k = g_prof_kind[i];
        if (k < nk) { ms[k] += t; flops[k] += g_prof_flops[i]; count[k]++; }
        if (dump) fprintf(stderr, "HIFIC_PROF %s %.3f %.4g %s\n", g_prof_kname[k], t * 1e3, g_prof_flops[i], g_prof_tag[i]);
    }
    g_prof_n = 0;
    return nk;
}

extern "C" int hific_pack_batch(const void* jobs_dev, const int* prefix_dev, int njobs, int total_blocks, size_t lds_bytes,
                                int dtype, hipStream_t st) {
    if (!jobs_dev || !prefix_dev || njobs <= 0 || total_blocks <= 0) return HIFIC_ERR_ARG;
    if (dtype == HIFIC_BF16) {
        if (lds_bytes > 48 * 1024)
            gc_set_max_lds((const void*)pack_batch_kernel<bf16_t>, (int)lds_bytes);
        hipLaunchKernelGGL(pack_batch_kernel<bf16_t>, dim3(total_blocks), dim3(256), lds_bytes, st, (const PackJob*)jobs_dev, prefix_dev, njobs);
    } else if (dtype == HIFIC_F32) {
        if (lds_bytes > 48 * 1024)
            gc_set_max_lds((const void*)pack_batch_kernel<float>, (int)lds_bytes);
        hipLaunchKernelGGL(pack_batch_kernel<float>, dim3(total_blocks), dim3(256), lds_bytes, st, (const PackJob*)jobs_dev, prefix_dev, njobs);
    } else return HIFIC_ERR_ARG;
    return hific_launch_status();
}

// ===================================================================================================
// Host-side planning
// ===================================================================================================
// Split-K epilogue of the forward-type kernels: out = act(sum_s part[s] + bias[k]) over [N, K, plane]
template <typename TO>
__global__ void ksplit_reduce_kernel(const float* __restrict__ part, long long stride, int nsplit, long long total, int K,
                                     int plane, const float* __restrict__ bias, int act, TO* __restrict__ out) {
    const float slope = act == ACT_RELU ? 0.f : (act == ACT_LEAKY ? 0.2f : 1.f);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        float v[16];
#pragma unroll
        for (int s = 0; s < 16; ++s) v[s] = part[(s < nsplit ? s : 0) * stride + i];     // independent loads in flight
        float acc = bias ? bias[(int)((i / plane) % K)] : 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) acc += s < nsplit ? v[s] : 0.f;
        DT<TO>::st(out + i, acc > 0.f ? acc : acc * slope);
    }
}

// Environment knobs are read ONCE per process and call site (the planner asks for dozens per launch: getenv walks the
// whole environment block each time).  Key = the address of the name literal.  hific_env_refresh() drops the cache (tests and
// tools that flip a knob inside one process).
static std::mutex g_env_mu;
static std::unordered_map<const void*, int> g_env_cache;
int gc_env_int(const char* name, int dflt) {
    std::lock_guard<std::mutex> lock(g_env_mu);
    auto it = g_env_cache.find((const void*)name);
    if (it != g_env_cache.end()) return it->second;
    const char* s = getenv(name);
    const int v = s ? atoi(s) : dflt;
    g_env_cache.emplace((const void*)name, v);
    return v;
}
extern "C" int hific_env_refresh(void) {
    std::lock_guard<std::mutex> lock(g_env_mu);
    g_env_cache.clear();
    return HIFIC_OK;
}
static inline int env_int(const char* name, int dflt) { return gc_env_int(name, dflt); }

static const int kLdsBudget = 150 * 1024;

// choose (TH, TW, NI) for a (u,v) domain: exhaustive search over tile shapes (not only powers of two, so padded
// data-gradient domains such as 18x18 tile as 7x18 instead of 8x16) minimising a cost model
//   tiles * (128 pixels * ntaps MFMA work + staging of the halo patch)
// subject to the LDS budget.  need16: pixels per tile must be a multiple of 16 (weight-gradient K-slice).
static bool choose_tile(int N, int OHt, int OWt, int ist, int span_y, int span_x, int pitch, int fixed_bytes,
                        int pref_budget, int& TH, int& TW, int& NI, int ntaps = 9, bool need16 = false) {
    for (int pass = 0; pass < 2; ++pass) {
        const long long budget = pass == 0 ? pref_budget : kLdsBudget;
        double best = 1e300;
        bool found = false;
        const int twmax = OWt < GC_NPIX ? OWt : GC_NPIX;
        // every patch row is a separate run of cache lines: charge ~one extra 128-byte line per row so that big
        // planes get wide tiles (narrow tiles re-fetch each line once per tile that touches it)
        const double line_px = (double)env_int("HIFIC_LINE_PX", 48);
        for (int tw = twmax; tw >= 1; --tw) {
            if (tw < 4 && tw != twmax) break;
            int thmax = GC_NPIX / tw; if (thmax > OHt) thmax = OHt;
            for (int th = thmax; th >= 1; --th) {
                const bool whole = (th >= OHt && tw >= OWt);
                int nimax = whole ? GC_NPIX / (th * tw) : 1;       // several images per tile only for whole planes
                if (nimax > N) nimax = N;
                for (int ni = nimax; ni >= 1; --ni) {
                    if (need16 && (ni * th * tw) % 16 != 0) continue;
                    const long long ph = (long long)(th - 1) * ist + span_y, pw = (long long)(tw - 1) * ist + span_x;
                    const long long bytes = (long long)ni * ph * pw * pitch + fixed_bytes;
                    if (bytes > budget) continue;
                    const double tiles = (double)cdiv(OHt, th) * cdiv(OWt, tw) * cdiv(N, ni);
                    const double cost = tiles * (128.0 * ntaps + 3.0 * (double)(ni * ph) * ((double)pw + line_px));
                    // prefer wider tiles on ties (longer coalesced runs)
                    if (cost < best * (1.0 - 1e-9)) { best = cost; TH = th; TW = tw; NI = ni; found = true; }
                }
            }
        }
        if (found) return true;
    }
    return false;
}

struct TapList { int n; short dy[GC_MAXTAPS], dx[GC_MAXTAPS], r[GC_MAXTAPS], s[GC_MAXTAPS]; };

static void finish_phase(GcPhase& ph, const GcParams& p) {
    int dymin = 0, dymax = 0, dxmin = 0, dxmax = 0;
    for (int t = 0; t < ph.ntaps; ++t) {
        int dy = p.tap_dy[ph.tap0 + t], dx = p.tap_dx[ph.tap0 + t];
        if (t == 0) { dymin = dymax = dy; dxmin = dxmax = dx; }
        if (dy < dymin) dymin = dy; if (dy > dymax) dymax = dy;
        if (dx < dxmin) dxmin = dx; if (dx > dxmax) dxmax = dx;
    }
    ph.dy_min = dymin; ph.dx_min = dxmin;
    ph.PH = dymax - dymin + 1; ph.PW = dxmax - dxmin + 1;   // spans; converted to patch extents later
}

// Weight packing for plan `p` (destination layout = gc_wp_index): pack tiling choice, plan-only hand-over of the job
// (hific_*_pack_plan), destination in the caller's cache or the workspace, and the pack launch unless the cache is current.
template <typename T>
static int gc_pack_weights(GcParams& p, long long wp_elems, const float* w, const float* w_scale, long long sm, long long sc,
                           long long sr, long long ss, WsAlloc& ws, hipStream_t st, bool* plan_only) {
    *plan_only = false;
    const size_t wp_bytes = (size_t)(wp_elems > 0 ? wp_elems : 1) * sizeof(T);
    // pack tiling (coalesced LDS-transposing kernel when the taps of a (m,c) pair are contiguous and one of m/c is adjacent)
    PackJob job; memset(&job, 0, sizeof(job));
    {
        int RS = 0;
        for (int i = 0; i < p.nphase; ++i) RS += p.ph[i].ntaps;       // phases partition the R*S taps
        const bool contiguous = (sr == p.tap_sw && ss == 1 && RS > 0);
        job.sm = sm; job.sc = sc; job.sr = sr; job.ss = ss; job.RS = RS; job.MB = 1; job.dtype = DT<T>::code;
        job.wp_bytes = (long long)wp_bytes;
        if (contiguous && sc == RS && !p.csplit && !p.msplit && !env_int("HIFIC_OLD_PACK", 0)) {
            int MB = 40960 / (64 * RS * 4); if (MB > 16) MB = 16; if (MB < 1) MB = 1;
            job.mode = 0; job.MB = MB; job.gx = p.Cpad / 64 + (p.Cpad % 64 ? 1 : 0); job.gy = cdiv(p.Kpad, MB);
            job.lds_bytes = (int)((size_t)64 * ((MB * RS) | 1) * sizeof(float));
        } else if (contiguous && sm == RS && !p.csplit && !p.msplit && !env_int("HIFIC_OLD_PACK", 0)) {
            int MB = env_int("HIFIC_PACK_MB", 144) / RS; if (MB > 32) MB = 32; if (MB < 1) MB = 1;
            job.mode = 1; job.MB = MB; job.gx = p.Cpad / 64 + (p.Cpad % 64 ? 1 : 0); job.gy = cdiv(p.Kpad, MB);
            job.lds_bytes = (int)((size_t)64 * ((MB * RS) | 1) * sizeof(float));
        } else {
            long long mx = 0;
            for (int i = 0; i < p.nphase; ++i) {
                long long e = (long long)p.Kpad * p.ph[i].ntaps * p.Cpad;
                if (e > mx) mx = e;
            }
            int gx = (int)((mx + 255) / 256); if (gx > 4096) gx = 4096; if (gx < 1) gx = 1;
            job.mode = 2; job.gx = gx; job.gy = p.nphase; job.lds_bytes = 0;
        }
    }
    if (ws.plan_out) {          // plan-only call (hific_conv_pack_plan): hand the job to the caller, launch nothing
        job.p = p; job.p.wp = nullptr;
        *ws.plan_out = job;
        *plan_only = true;
        return HIFIC_OK;
    }
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
        if (job.mode == 0) {
            if (job.lds_bytes > 48 * 1024)
                gc_set_max_lds((const void*)pack_w2_kernel<T, 0>, job.lds_bytes);
            hipLaunchKernelGGL((pack_w2_kernel<T, 0>), dim3(job.gx, job.gy), dim3(256), job.lds_bytes, st, p, w, w_scale, sm, sc, job.RS, job.MB);
        } else if (job.mode == 1) {
            if (job.lds_bytes > 48 * 1024)
                gc_set_max_lds((const void*)pack_w2_kernel<T, 1>, job.lds_bytes);
            hipLaunchKernelGGL((pack_w2_kernel<T, 1>), dim3(job.gx, job.gy), dim3(256), job.lds_bytes, st, p, w, w_scale, sm, sc, job.RS, job.MB);
        } else {
            hipLaunchKernelGGL(pack_w_kernel<T>, dim3(job.gx, p.nphase), dim3(256), 0, st, p, w, w_scale, sm, sc, sr, ss);
        }
    }
    return HIFIC_OK;
}
int gc_pack_weights_bf16(GcParams& p, long long wp_elems, const float* w, const float* w_scale, long long sm, long long sc,
                         long long sr, long long ss, WsAlloc& ws, hipStream_t st, bool* plan_only) {
    return gc_pack_weights<bf16_t>(p, wp_elems, w, w_scale, sm, sc, sr, ss, ws, st, plan_only);
}

template <typename T, int BC>
static int launch_gconv_tb(GcParams& p, const float* w, const float* w_scale, long long sm, long long sc,
                           long long sr, long long ss, WsAlloc& ws, hipStream_t st, int tile_budget = kLdsBudget) {
    using Cfg = GcCfg<T>;
    constexpr int PITCH = BC * (int)sizeof(T) + Cfg::PAD;
    // Phase-merged software-pipelined kernel (gconv_sp9_kernel PHS) for the kernel-3 stride-2 transposed structure:
    // four phases with 1,2,2,4 (conv-transpose forward) or 4,2,2,1 (stride-2 conv data gradient) taps over a stride-1
    // bf16 input.  Decided here because it fixes the M tile (four accumulator sets => 64 rows).
    int phs = 0;
    int u_dymin = 0, u_dymax = 0, u_dxmin = 0, u_dxmax = 0;
    if constexpr (std::is_same<T, bf16_t>::value && BC == 64) {
        // >= 4 channel chunks: the software pipeline needs chunks to overlap; single-chunk large-plane layers are
        // HBM/epilogue-bound and did better with two co-resident generic workgroups (K60 C120 @128x128: 205 -> 230 us,
        // K480 C960 @16x16: 134 -> 89 us)
        if (p.nphase == 4 && p.ist == 1 && !p.in_f32 && p.K > 32 && p.C >= 256 && !p.rfx && !p.split && !p.oscale &&
            !env_int("HIFIC_NO_PHS", 0)) {
            const int n0_ = p.ph[0].ntaps, n1_ = p.ph[1].ntaps, n2_ = p.ph[2].ntaps, n3_ = p.ph[3].ntaps;
            if (n0_ == 1 && n1_ == 2 && n2_ == 2 && n3_ == 4) phs = 1;
            else if (n0_ == 4 && n1_ == 2 && n2_ == 2 && n3_ == 1) phs = 2;
            if (phs) {
                for (int t = 0; t < 9; ++t) {
                    const int dy = p.tap_dy[t], dx = p.tap_dx[t];
                    if (t == 0) { u_dymin = u_dymax = dy; u_dxmin = u_dxmax = dx; }
                    if (dy < u_dymin) u_dymin = dy; if (dy > u_dymax) u_dymax = dy;
                    if (dx < u_dxmin) u_dxmin = dx; if (dx > u_dxmax) u_dxmax = dx;
                }
                for (int i = 0; i < 4; ++i) if (p.ph[i].tap0 != (i == 0 ? 0 : p.ph[i - 1].tap0 + p.ph[i - 1].ntaps)) phs = 0;
            }
        }
    }
    // Merged-phase kernel (gconv_mp_kernel): the four phases of a stride-2 transposed structure in one workgroup per (u, v)
    // tile.  Where the software-pipelined phase-merged kernel above applies (>= 4 channel chunks, small planes) that one stays.
    bool mp = false;
    if constexpr (std::is_same<T, bf16_t>::value && BC == 64) {
        // (K <= 32: a 64-row tile would be mostly padding - the 15-channel data gradient of the Discriminator's first layer ran
        //  252 -> 291 us on it)
        if (!phs && p.nphase == 4 && p.ist == 1 && p.ost == 2 && p.K > 32 && !p.rfx && !p.resid && !p.csplit && !p.msplit &&
            !p.in_f32 && (long long)p.N * p.K * p.OHf * p.OWf < (1ll << 31) && env_int("HIFIC_MP", 1)) {   // (32-bit store offsets)
            mp = true;
            int nt_ = 0;
            for (int i = 0; i < 4; ++i) { if (p.ph[i].tap0 != nt_ || p.ph[i].ntaps < 1) mp = false; nt_ += p.ph[i].ntaps; }
            if (nt_ > GC_MAXTAPS) mp = false;
            for (int t = 0; t < nt_ && mp; ++t) {
                const int dy = p.tap_dy[t], dx = p.tap_dx[t];
                if (t == 0) { u_dymin = u_dymax = dy; u_dxmin = u_dxmax = dx; }
                if (dy < u_dymin) u_dymin = dy; if (dy > u_dymax) u_dymax = dy;
                if (dx < u_dxmin) u_dxmin = dx; if (dx > u_dxmax) u_dxmax = dx;
            }
        }
    }
    // M tile
    int bm;
    {
        long long w128 = cdivl(p.K, 128) * 128, w64 = cdivl(p.K, 64) * 64;
        if (p.K <= 32) bm = 32;
        else if (w128 * 100 > w64 * 110) bm = 64;
        else bm = 128;
        int e = env_int("HIFIC_BM", 0);
        if ((e == 64 || e == 128) && p.K > 32) bm = e;
        if (phs || mp) bm = 64;
    }
    p.Kpad = cdiv(p.K, bm) * bm;
    p.Cpad = cdiv(p.C, BC) * BC;
    p.dbg = env_int("HIFIC_DBG", 0);
    p.tap_sw = (int)sr;
    // tile shape: common to all phases (largest span decides)
    int span_y = 1, span_x = 1, OHt = 1, OWt = 1;
    for (int i = 0; i < p.nphase; ++i) {
        if (p.ph[i].PH > span_y) span_y = p.ph[i].PH;
        if (p.ph[i].PW > span_x) span_x = p.ph[i].PW;
        if (p.ph[i].OHt > OHt) OHt = p.ph[i].OHt;
        if (p.ph[i].OWt > OWt) OWt = p.ph[i].OWt;
    }
    int maxtaps = 1;
    for (int i = 0; i < p.nphase; ++i) if (p.ph[i].ntaps > maxtaps) maxtaps = p.ph[i].ntaps;
    // taps per barrier step (gconv_kernel TPS): 7 for >= 49 taps, 4 for >= 16, when the per-thread weight prefetch stays
    // within 8 x 16 bytes per register set (few-row / few-channel tiles: exactly the layers that are barrier-bound)
    // Launches that cannot even give every CU one workgroup (the hyperprior's 4x4 .. 16x16 planes: 30-160 workgroups of
    // 45-125 serial steps) are bound by the latency of ONE 8 KB weight tile per step (1 us per step, 240 GB/s chip-wide):
    // several tiles in flight per step from 4 taps up, and no co-residency constraint on the ring size.
    long long est_grid = 0;
    for (int i = 0; i < p.nphase; ++i)
        est_grid += cdivl((long long)p.N * p.ph[i].OHt * p.ph[i].OWt, GC_NPIX) * cdiv(p.K, 64);
    const bool small_grid = est_grid < 256 && !env_int("HIFIC_NO_TPS_SMALL", 0);
    // ... and with >= 25 taps, 32-row tiles: twice the workgroups, and a 4 KB weight tile per tap lets 7 taps share a step
    if (small_grid && maxtaps >= 25 && bm == 64 && !phs && !mp && std::is_same<T, bf16_t>::value && BC == 64 &&
        !env_int("HIFIC_NO_TPS", 0) && !env_int("HIFIC_BM", 0)) {
        bm = 32; p.Kpad = cdiv(p.K, bm) * bm;
    }
    auto pick_tps = [&](int bm_) -> int {
        if (!std::is_same<T, bf16_t>::value || env_int("HIFIC_NO_TPS", 0)) return 1;
        const int nwp = cdiv(bm_ * (BC * (int)sizeof(T) / 16), 256);
        // (7 taps per step on the 32-row 5x5 layers measured 190 vs 80 us with 4: not taken)
        const int cand = maxtaps >= 49 ? 7 : (maxtaps >= 16 ? 4 : ((small_grid && maxtaps >= 4) ? 4 : 1));
        // ... and the weight ring must leave room for two co-resident workgroups (7 taps x 4.6 KB x 2 next to a 55 KB patch
        // put the 60->3 layer at one workgroup per CU: 640 -> 790 us)
        const int ring_cap = small_grid ? 96 * 1024 : 44 * 1024;
        return (cand > 1 && nwp * cand <= 8 && 2 * cand * bm_ * PITCH <= ring_cap) ? cand : 1;
    };
    int tps = pick_tps(bm);
    int wbytes = 512 + 2 * tps * bm * PITCH;
    // Full-LDS tiles (1 workgroup per CU, fewer halo re-reads) when they still give >= one workgroup per CU;
    // otherwise tiles small enough for two co-resident workgroups.
    bool tiled = false;
    if (phs) {
        // one merged patch for the four phases: union of the tap offsets, at most 192 pixels (3 per lane)
        span_y = u_dymax - u_dymin + 1; span_x = u_dxmax - u_dxmin + 1;
        if (choose_tile(p.N, OHt, OWt, 1, span_y, span_x, PITCH, 0, 192 * PITCH, p.TH, p.TW, p.NI, 9) && p.NI == 1) {
            tiled = true; tps = 1; wbytes = 512 + 2 * bm * PITCH;
        } else {
            phs = 0;      // no tile fits: the generic kernel with the 64-row tiles already chosen
            span_y = 1; span_x = 1;
            for (int i = 0; i < p.nphase; ++i) { if (p.ph[i].PH > span_y) span_y = p.ph[i].PH; if (p.ph[i].PW > span_x) span_x = p.ph[i].PW; }
        }
    }
    if (mp) {
        // one union patch, one tap per step, tiles small enough for two co-resident workgroups
        tps = 1; wbytes = 512 + 2 * bm * PITCH;
        span_y = u_dymax - u_dymin + 1; span_x = u_dxmax - u_dxmin + 1;
        int nt_all = 0;
        for (int i = 0; i < 4; ++i) nt_all += p.ph[i].ntaps;
        // (launches that cannot give every CU a workgroup - the hyperprior's 4x4 .. 16x16 planes - keep the per-phase grid and its
        //  split-K: a quarter of the workgroups with four times the serial steps is the wrong trade there)
        if (choose_tile(p.N, OHt, OWt, 1, span_y, span_x, PITCH, wbytes, 76 * 1024, p.TH, p.TW, p.NI, nt_all) &&
            2ll * cdiv(OHt, p.TH) * cdiv(OWt, p.TW) * cdiv(p.N, p.NI) * (p.Kpad / bm) >= env_int("HIFIC_MP_MIN_GRID", 512))
            tiled = true;
        else {
            mp = false;
            tps = pick_tps(bm); wbytes = 512 + 2 * tps * bm * PITCH;
            span_y = 1; span_x = 1;
            for (int i = 0; i < p.nphase; ++i) { if (p.ph[i].PH > span_y) span_y = p.ph[i].PH; if (p.ph[i].PW > span_x) span_x = p.ph[i].PW; }
        }
    }
    if (!phs && !mp && choose_tile(p.N, OHt, OWt, p.ist, span_y, span_x, PITCH, wbytes, tile_budget, p.TH, p.TW, p.NI, maxtaps)) {
        const long long g = (long long)cdiv(OHt, p.TH) * cdiv(OWt, p.TW) * cdiv(p.N, p.NI) * (p.Kpad / bm) * p.nphase;
        tiled = g >= env_int("HIFIC_GC_BIGTILE_MIN_GRID", 256);
    }
    // Layers whose 64-channel halo patch only fits as a whole-LDS tile (stride-2 convs: four input pixels per output
    // pixel) run ONE workgroup per CU, and its weight-load / staging / MFMA / bias / store-drain latencies are all
    // exposed back to back (60->120 stride 2 @256x256: staging 40 + epilogue 42 + MFMA 33 + weights 19 + launch 43 us
    // of 183).  32-channel chunks halve the patch and the weight ring, so two workgroups co-reside and overlap.
    if constexpr (std::is_same<T, bf16_t>::value && BC == 64) {
        // ... when the launch has at least two workgroups per CU to co-reside (a 256-workgroup launch only gets the
        // doubled step count: 960<-480 @16x16 138 -> 191 us)
        const long long g64 = (long long)cdiv(OHt, p.TH) * cdiv(OWt, p.TW) * cdiv(p.N, p.NI) * (p.Kpad / bm) * p.nphase;
        if (tiled && !phs && !mp && g64 >= 512 && env_int("HIFIC_BC32", 1)) {
            size_t need = 0;
            for (int i = 0; i < p.nphase; ++i) {
                const size_t b = (size_t)wbytes + (size_t)p.NI * ((p.TH - 1) * p.ist + p.ph[i].PH) * ((p.TW - 1) * p.ist + p.ph[i].PW) * PITCH;
                if (b > need) need = b;
            }
            if (need > (size_t)80 * 1024) {
                const GcParams saved = p;
                const int rc32 = launch_gconv_tb<bf16_t, 32>(p, w, w_scale, sm, sc, sr, ss, ws, st, 72 * 1024);
                if (rc32 != HIFIC_ERR_UNSUPPORTED) return rc32;
                p = saved;                                 // no 72 KB tile at 32 channels either: whole-LDS tile it is
            }
        }
    }
    if (!tiled && !choose_tile(p.N, OHt, OWt, p.ist, span_y, span_x, PITCH, wbytes, 72 * 1024, p.TH, p.TW, p.NI, maxtaps))
        return HIFIC_ERR_UNSUPPORTED;
    p.tiles_n = cdiv(p.N, p.NI);
    if (bm == 128 && !phs && !mp && !env_int("HIFIC_NO_BM_TAIL", 0)) {
        // 128-row tiles run one workgroup per CU: a grid of e.g. 1.5 x 256 workgroups (18x18 padded-gradient
        // domain of the 16x16x960 layers) leaves half the chip idle in its second wave.  64-row tiles co-reside two
        // per CU, so the same launch quantises at 512 slots.
        long long tiles = 0;
        for (int i = 0; i < p.nphase; ++i)
            tiles += (long long)cdiv(p.ph[i].OHt, p.TH) * cdiv(p.ph[i].OWt, p.TW) * p.tiles_n;
        const long long g128 = tiles * cdiv(p.K, 128), g64 = tiles * cdiv(p.K, 64);
        const double eff128 = (double)p.K / (cdiv(p.K, 128) * 128.0) * (double)g128 / (double)(cdivl(g128, 256) * 256);
        const double eff64 = (double)p.K / (cdiv(p.K, 64) * 64.0) * (double)g64 / (double)(cdivl(g64, 512) * 512);
        if (g128 > 256 && g128 <= 512 && eff64 >= eff128 - 0.02 && eff128 < 0.8) {
            bm = 64; p.Kpad = cdiv(p.K, bm) * bm; tps = pick_tps(bm); wbytes = 512 + 2 * tps * bm * PITCH;
        }
    }
    long long wp_elems = 0;
    int max_tiles = 0;
    size_t lds = 0;
    for (int i = 0; i < p.nphase; ++i) {
        GcPhase& ph = p.ph[i];
        const int sy = ph.PH, sx = ph.PW;
        ph.PH = (p.TH - 1) * p.ist + sy;
        ph.PW = (p.TW - 1) * p.ist + sx;
        ph.PWs = ph.PW;
        if (std::is_same<T, bf16_t>::value && BC == 64 && (ph.PW % 16) != 0 && env_int("HIFIC_PWS", 0)) {   // measured: no gain
            const int pws = (ph.PW + 15) / 16 * 16;
            if ((size_t)wbytes + (size_t)p.NI * ph.PH * pws * PITCH <= (size_t)96 * 1024) ph.PWs = pws;
        }
        ph.tiles_y = cdiv(ph.OHt, p.TH);
        ph.tiles_x = cdiv(ph.OWt, p.TW);
        ph.wp_off = wp_elems;
        wp_elems += (long long)p.Kpad * ph.ntaps * p.Cpad;
        int nt = p.tiles_n * ph.tiles_y * ph.tiles_x;
        if (nt > max_tiles) max_tiles = nt;
        size_t b = (size_t)wbytes + (size_t)p.NI * ph.PH * ph.PWs * PITCH;
        if (b > lds) lds = b;
    }
    // wide-load staging (stage_W): bf16 NCHW source whose rows are 16-byte aligned; + one dump row of LDS
    p.wstage = 0;
    if constexpr (std::is_same<T, bf16_t>::value && BC >= 32) {
        // measured: pays on the stride-2 layers (halo patch = 4 input pixels per output pixel, 11 -> 2-3 round trips:
        // 60->120 @256x256 222 -> 180 us); stride-1 patches are 2-3 round trips either way and the 36-pixel rows of the
        // 60->3 virtual-row layer waste half of every aligned group (191 -> 221 us)
        const int wst = env_int("HIFIC_WSTAGE", 1);
        if (!p.in_f32 && p.IW % 8 == 0 && p.IW >= env_int("HIFIC_WSTAGE_MINW", 32) && ((size_t)p.in & 15) == 0 &&
            lds + PITCH <= (size_t)kLdsBudget && tps == 1 && (wst == 2 || (wst == 1 && p.ist >= 2))) {
            p.wstage = 1;
            lds += PITCH;
        }
    }
    if (lds > (size_t)kLdsBudget) return HIFIC_ERR_UNSUPPORTED;
    // wide-store epilogue (gc_epilogue_wide): one phase, output stride 1, bf16 output written straight to `out`, pieces of
    // 8 pixels never straddle a tile row / the image edge, no residual
    p.epi_wide = 0;
    if constexpr (std::is_same<T, bf16_t>::value) {
        if (p.nphase == 1 && p.ost == 1 && !p.out_f32 && !p.fold_h && !p.resid && p.TW % 8 == 0 && p.OWf % 8 == 0 &&
            p.ph[0].OWt % 8 == 0 && p.ph[0].ooy == 0 && p.ph[0].oox == 0 && !env_int("HIFIC_NO_WIDE_EPI", 0)) {
            p.epi_wide = 1;
            const size_t need = (size_t)4 * (bm / 2 < 32 ? 32 : bm / 2) * (2 * 64 + 16);     // 4 waves x rows x row bytes
            if (need > lds) lds = need;
        }
    }
    // software-pipelined kernel: one phase of exactly 9 taps, input stride 1, bf16 input, halo patch <= 192 pixels
    bool use_sp9 = false;
    if constexpr (std::is_same<T, bf16_t>::value && BC == 64) {
        use_sp9 = bm >= 64 && p.nphase == 1 && p.ph[0].ntaps == 9 && p.ist == 1 && !p.in_f32 && !p.split && !p.oscale &&
                  p.NI * p.ph[0].PH * p.ph[0].PW <= 192 && !env_int("HIFIC_NO_SP", 0) &&
                  64 + 3 * (size_t)bm * PITCH + 2 * (((size_t)(p.NI * p.ph[0].PH * p.ph[0].PW + 2) * PITCH + 15) & ~(size_t)15) <= (size_t)kLdsBudget;
    }
    if (mp) {
        GcPhase& u = p.ph[4];
        memset(&u, 0, sizeof(u));
        u.tap0 = 0; u.dy_min = u_dymin; u.dx_min = u_dxmin;
        for (int i = 0; i < 4; ++i) u.ntaps += p.ph[i].ntaps;
        u.PH = (p.TH - 1) + (u_dymax - u_dymin + 1); u.PW = (p.TW - 1) + (u_dxmax - u_dxmin + 1); u.PWs = u.PW;
        for (int i = 0; i < 4; ++i) {
            if (p.ph[i].tiles_y > u.tiles_y) u.tiles_y = p.ph[i].tiles_y;
            if (p.ph[i].tiles_x > u.tiles_x) u.tiles_x = p.ph[i].tiles_x;
            if (p.ph[i].OHt > u.OHt) u.OHt = p.ph[i].OHt;
            if (p.ph[i].OWt > u.OWt) u.OWt = p.ph[i].OWt;
        }
        max_tiles = p.tiles_n * u.tiles_y * u.tiles_x;
        lds = (size_t)wbytes + (size_t)p.NI * u.PH * u.PWs * PITCH;
        use_sp9 = false;
        p.wstage = 0;
        // pair stores: the two column phases of an output row as one 4 / 8-byte store per pixel
        p.epi_wide = 0;
        if (!p.fold_h && p.OWf % 2 == 0 && p.ph[0].ooy == p.ph[1].ooy && p.ph[2].ooy == p.ph[3].ooy &&
            p.ph[0].oox == 0 && p.ph[1].oox == 1 && p.ph[2].oox == 0 && p.ph[3].oox == 1 &&
            p.ph[0].OHt == p.ph[1].OHt && p.ph[0].OWt == p.ph[1].OWt && p.ph[2].OHt == p.ph[3].OHt &&
            p.ph[2].OWt == p.ph[3].OWt && (((size_t)p.out) & 7) == 0 && !env_int("HIFIC_MP_NO_PAIR", 0))
            p.epi_wide = 2;
        // reflect-fold data gradients with an even left pad (the Encoder's asymmetric pad (1, 0, 0, 1)): pair stores into dx
        if (p.fold_h && p.out2 && !p.bias && p.act == ACT_NONE && p.fold_pl % 2 == 0 && p.fold_w % 2 == 0 &&
            p.ph[0].ooy == p.ph[1].ooy && p.ph[2].ooy == p.ph[3].ooy && p.ph[0].oox == 0 && p.ph[1].oox == 1 &&
            p.ph[2].oox == 0 && p.ph[3].oox == 1 && p.ph[0].OHt == p.ph[1].OHt && p.ph[2].OHt == p.ph[3].OHt &&
            (((size_t)p.out2) & 7) == 0 && !env_int("HIFIC_MP_NO_PAIR", 0))
            p.epi_wide = 3;
    }
    if (phs) {
        GcPhase& u = p.ph[4];
        memset(&u, 0, sizeof(u));
        u.ntaps = 9; u.tap0 = 0; u.dy_min = u_dymin; u.dx_min = u_dxmin;
        u.PH = (p.TH - 1) + (u_dymax - u_dymin + 1); u.PW = (p.TW - 1) + (u_dxmax - u_dxmin + 1); u.PWs = u.PW;
        for (int i = 0; i < 4; ++i) {
            if (p.ph[i].tiles_y > u.tiles_y) u.tiles_y = p.ph[i].tiles_y;
            if (p.ph[i].tiles_x > u.tiles_x) u.tiles_x = p.ph[i].tiles_x;
            if (p.ph[i].OHt > u.OHt) u.OHt = p.ph[i].OHt;
            if (p.ph[i].OWt > u.OWt) u.OWt = p.ph[i].OWt;
        }
        max_tiles = p.tiles_n * u.tiles_y * u.tiles_x;
        const int npatch = p.NI * u.PH * u.PW;
        use_sp9 = npatch <= 192 &&
                  64 + 3 * (size_t)bm * PITCH + 2 * (((size_t)(npatch + 2) * PITCH + 15) & ~(size_t)15) <= (size_t)kLdsBudget;
        if (!use_sp9) phs = 0;
    }
    if (p.rfx) {
        // gather-form reflect gradient: only gconv_sp9_kernel implements the per-lane tap displacements, and the border
        // lines a pixel needs must lie in its own tile's patch: rows/cols {0,1} and {H-2,H-1} never split across tiles
        const bool ok = use_sp9 && p.NI == 1 && p.TH >= 2 && p.TW >= 2 && (p.OHf - 1) % p.TH != 0 && (p.OWf - 1) % p.TW != 0;
        if (!ok) return HIFIC_ERR_UNSUPPORTED;
    }
    // A operands streamed global -> registers from a fragment-ordered packed image (gconv_sp9_kernel AG): the 128-row K-split
    // instantiations (residual-block forward and gather-form data gradient); decided before the pack job is built
    p.afrag = 0;
    if constexpr (std::is_same<T, bf16_t>::value && BC == 64) {
        if (use_sp9 && !phs && bm == 128 && env_int("HIFIC_SP9_KSPLIT", 2) == 2 && !env_int("HIFIC_SP9_DS", 0) &&
            env_int("HIFIC_SP9_AG", 1))
            p.afrag = 1;
    }
    // Split-K for launches that cannot fill the chip (see GcParams::ksplit): only where one workgroup's serial chain is
    // long enough to pay for the float32 partials and the reduce pass (~10-15 us).  Measured (round 3, batch 16): 320<-960
    // 5x5 s2 @8x8 (80 workgroups x 105 steps) 201 -> 59 us, its transposed sibling 162 -> 49, 220<-2880 3x3 @16x16 (64
    // workgroups x 405 steps) 178 -> 108; launches of 128-256 workgroups or < ~60 us of chain got 10-40 % SLOWER.
    p.ksplit = 1; p.kchunks = 0; p.kpart = nullptr; p.kpart_stride = 0;
    if constexpr (std::is_same<T, bf16_t>::value) {
        const long long g0 = mp ? (1ll << 40) : (long long)max_tiles * (p.Kpad / bm) * (phs ? 1 : p.nphase);
        const int nch = p.Cpad / BC;
        const double chain_us = (double)nch * (use_sp9 ? 9 * 0.5 : cdiv(maxtaps, tps) * 1.5);
        // (the generic kernel's step is a barrier + synchronous staging, ~1.5-2 us; the software-pipelined one ~0.5 us)
        // (up to 400 workgroups of 32-row tiles - several co-reside per CU: 320<-960 5x5 s2 @16x16, 320 workgroups, 232 -> 170 us;
        //  256 workgroups of 128-row tiles are one per CU and lose 15-35 % when split)
        const long long gmax = use_sp9 ? env_int("HIFIC_KSPLIT_MAXGRID_SP", 160)
                                       : (bm <= 32 ? env_int("HIFIC_KSPLIT_MAXGRID_32", 400) : env_int("HIFIC_KSPLIT_MAXGRID", 160));
        if (g0 <= gmax && nch >= 4 &&
            chain_us >= (use_sp9 ? env_int("HIFIC_KSPLIT_MIN_US_SP", 60) : env_int("HIFIC_KSPLIT_MIN_US", 20)) &&
            !p.fold_h && !p.resid && !p.msplit && !p.csplit && !p.oscale && env_int("HIFIC_KSPLIT", 1)) {
            int ks = (int)cdivl(env_int("HIFIC_KSPLIT_TARGET", 800), g0);
            if (ks > nch / 2) ks = nch / 2;
            if (ks > 16) ks = 16;
            if (ks >= 2) {
                p.kchunks = cdiv(nch, ks);
                p.ksplit = cdiv(nch, p.kchunks);
                p.kpart_stride = (long long)p.N * p.K * p.OHf * p.OWf;
                p.epi_wide = 0;
            }
        }
    }
    {
        bool plan_only = false;
        const int rcp = gc_pack_weights<T>(p, wp_elems, w, w_scale, sm, sc, sr, ss, ws, st, &plan_only);
        if (rcp != HIFIC_OK || plan_only) return rcp;
    }
    p.max_tiles = max_tiles;
    if (p.ksplit > 1) {
        p.kpart = (float*)ws.take((size_t)p.ksplit * (size_t)p.kpart_stride * sizeof(float));
        if (!p.kpart) { p.ksplit = 1; p.kchunks = 0; }          // no room: one pass (epi_wide stays off: harmless)
    }
    dim3 grid(max_tiles * (p.Kpad / bm), p.ksplit, phs ? 1 : (mp ? 2 : p.nphase));
    // algorithmic FLOPs of the op (set by the caller on the op's REAL output domain: a reflect-padded data gradient
    // computes on the padded plane, which is extra work, not extra useful FLOPs)
    const double aflops = p.aflops;
    char ptag[112];
    snprintf(ptag, sizeof(ptag), "gconv K%d C%d N%d in%dx%d out%dx%d ph%d taps%d ist%d ost%d tile%dx%dx%d bm%d tps%d grid%d",
             p.K, p.C, p.N, p.IH, p.IW, p.OHf, p.OWf, p.nphase, maxtaps, p.ist, p.ost, p.NI, p.TH, p.TW, bm, tps,
             max_tiles * (p.Kpad / bm) * p.nphase * p.ksplit);
    char kname[PROF_NAMELEN];
    // four-quarter form (gconv_sp9_kernel KSP = 4).  Measured (round 3, 960->960 @16x16x16, in the training cycle):
    // 68.7 -> 66.4 us forward, 76.1 -> 74.3 us gather-form data gradient against the two-half form (HIFIC_SP9_W4=0)
    const int sp9_w4 = (use_sp9 && p.afrag) ? (env_int("HIFIC_SP9_W4", 1) != 0) : 0;
    if (use_sp9) snprintf(kname, sizeof(kname), "gconv_sp9_kernel<%d,%d%s>", bm / 64,
                          sp9_w4 ? 4 : phs ? 1 : p.rfx ? (bm == 128 ? 2 : 1)
                                : ((env_int("HIFIC_SP9_KSPLIT", 2) == 2 && (bm == 128 || env_int("HIFIC_SP9_KSPLIT64", 0))) ? 2 : 1),
                          phs ? (phs == 1 ? ",phs1" : ",phs2") : (p.rfx ? ",rfx" : ""));
    // the profiler keeps the residual-block trunk (>= 512 x 512 channels, one workgroup per CU) apart from the 220 / 320-channel
    // launches of the same instantiation (K-split grids, a third of the rows): roofline.frac of the trunk is a property of the
    // kernel, the average over both classes was a property of the layer mix
    if (use_sp9) { if (!(p.K >= 512 && p.C >= 512) && strlen(kname) + 8 < sizeof(kname)) strcat(kname, " narrow"); }
    else if (mp) snprintf(kname, sizeof(kname), "gconv_mp_kernel%s", p.split ? "<split>" : "");
    else snprintf(kname, sizeof(kname), "gconv_kernel<%s,%d,%s>", std::is_same<T, float>::value ? "f32" : "bf16", BC,
                  bm == 128 ? "2,2,2,2" : (bm == 64 ? "2,2,1,2" : "1,4,1,1"));
    const int pslot = prof_open(kname, aflops, st, ptag);
    // full-LDS tiles run one workgroup per CU: the staging variant with all of a chunk's loads in flight (QB = 12)
    constexpr bool kBigStage = std::is_same<T, bf16_t>::value && BC == 64;
    // measured (round 2): slower than two co-resident workgroups with one round trip per 64 pixels (the phases of one
    // workgroup overlap the other's) - kept as an opt-in experiment
    const bool bigstage = kBigStage && tiled && lds > 80 * 1024 && env_int("HIFIC_BIGSTAGE", 0);
#define GC_LAUNCH(WGM, WGN, WM, WN)                                                                      \
    do {                                                                                                 \
        void (*kfn)(const GcParams) = gconv_kernel<T, BC, WGM, WGN, WM, WN, 1, 1>;                       \
        if constexpr (kBigStage) { if (bigstage) kfn = gconv_kernel<T, BC, WGM, WGN, WM, WN, 12, 1>; }   \
        if constexpr (std::is_same<T, bf16_t>::value && BC >= 32) {                                      \
            if (p.wstage) kfn = gconv_kernel<T, BC, WGM, WGN, WM, WN, -1, 1>;                            \
        }                                                                                                \
        if constexpr (std::is_same<T, bf16_t>::value) {                                                  \
            constexpr int nwp_ = (WGM * WM * 32 * (BC * (int)sizeof(T) / 16) + 255) / 256;               \
            if constexpr (nwp_ * 4 <= 8) { if (tps == 4) kfn = gconv_kernel<T, BC, WGM, WGN, WM, WN, 1, 4>; } \
            if constexpr (nwp_ * 7 <= 8) { if (tps == 7) kfn = gconv_kernel<T, BC, WGM, WGN, WM, WN, 1, 7>; } \
        }                                                                                                \
        if constexpr (!(std::is_same<T, bf16_t>::value && BC >= 32)) {                                   \
            if (p.split) kfn = nullptr;      /* pair layout: only the SPLIT instantiations (bf16, >= 32-channel chunks) */ \
        }                                                                                                \
        if constexpr (std::is_same<T, bf16_t>::value && BC >= 32) {                                      \
            if (p.split) {                                                                               \
                constexpr int nwp_ = (WGM * WM * 32 * (BC * (int)sizeof(T) / 16) + 255) / 256;           \
                kfn = p.wstage ? gconv_kernel<T, BC, WGM, WGN, WM, WN, -1, 1, true>                      \
                               : gconv_kernel<T, BC, WGM, WGN, WM, WN, 1, 1, true>;                      \
                if constexpr (nwp_ * 4 <= 8) { if (tps == 4) kfn = gconv_kernel<T, BC, WGM, WGN, WM, WN, 1, 4, true>; } \
                if (tps == 7) kfn = nullptr;                                                             \
            }                                                                                            \
        }                                                                                                \
        if (!kfn) { prof_close(pslot, st); return HIFIC_ERR_UNSUPPORTED; }                               \
        if (lds > 48 * 1024)                                                                             \
            gc_set_max_lds((const void*)kfn, (int)lds); \
        hipLaunchKernelGGL(kfn, grid, dim3(256), lds, st, p);                                            \
    } while (0)
    bool sp_done = false;
    if constexpr (std::is_same<T, bf16_t>::value && BC == 64) {
        // software-pipelined kernel: one phase of exactly 9 taps, input stride 1, patch <= 192 pixels
        if (use_sp9) {
            const int npatch = phs ? p.NI * p.ph[4].PH * p.ph[4].PW : p.NI * p.ph[0].PH * p.ph[0].PW;
            const size_t lds_sp = 64 + 3 * (size_t)bm * PITCH + 2 * (((size_t)(npatch + 2) * PITCH + 15) & ~(size_t)15);
            if (lds_sp <= (size_t)kLdsBudget) {
                const bool ks2 = env_int("HIFIC_SP9_KSPLIT", 2) == 2;
#define SP9_LAUNCH(WM_, KSP_, RFX_, PHS_)                                                                           \
    do {                                                                                                            \
        gc_set_max_lds((const void*)gconv_sp9_kernel<WM_, KSP_, RFX_, PHS_>, (int)lds_sp); \
        hipLaunchKernelGGL((gconv_sp9_kernel<WM_, KSP_, RFX_, PHS_>), grid, dim3(256 * KSP_), lds_sp, st, p);       \
    } while (0)
                // shifted B fragments (DS): 16-pixel tile rows, taps (dy, dx) with dx ascending by one pixel
                // measured (round 2): 115 -> 128 us on 960x960 @16x16x16 - the exec-masked edge reads and the DPP -> MFMA
                // dependency cost more than the saved LDS bytes: opt-in experiment
                bool ds = !phs && !p.rfx && p.TW == 16 && p.ph[0].ntaps == 9 && env_int("HIFIC_SP9_DS", 0);
                for (int r = 0; r < 3 && ds; ++r)
                    for (int c = 0; c < 3; ++c)
                        ds = ds && p.tap_dy[3 * r + c] == p.tap_dy[3 * r] && p.tap_dx[3 * r + c] == p.tap_dx[3 * r] + c;
                if (p.afrag) {
                    // no weight ring; the K-split exchange (64 KB) is the larger LDS use for the usual 180-pixel patch
                    size_t lds_ag = 64 + 2 * (((size_t)(npatch + 2) * PITCH + 15) & ~(size_t)15);
                    // four reduction quarters on 64x128 wave slabs (gconv_sp9_kernel KSP = 4): half the A operand loads per MFMA
                    const int w4 = sp9_w4;
                    if (lds_ag < (w4 ? 131072u : 65536u)) lds_ag = w4 ? 131072 : 65536;
                    const int agl = env_int("HIFIC_SP9_AG", 1);
#define SP9_AG_LAUNCH(RFX_, L_)                                                                                         \
    do {                                                                                                                \
        gc_set_max_lds((const void*)gconv_sp9_kernel<2, 2, RFX_, 0, false, L_>, (int)lds_ag);                           \
        hipLaunchKernelGGL((gconv_sp9_kernel<2, 2, RFX_, 0, false, L_>), grid, dim3(512), lds_ag, st, p);               \
    } while (0)
#define SP9_W4_LAUNCH(WM_, RFX_, L_)                                                                                    \
    do {                                                                                                                \
        gc_set_max_lds((const void*)gconv_sp9_kernel<WM_, 4, RFX_, 0, false, L_>, (int)lds_ag);                         \
        hipLaunchKernelGGL((gconv_sp9_kernel<WM_, 4, RFX_, 0, false, L_>), grid, dim3(sp9_threads(4)), lds_ag, st, p);      \
    } while (0)
                    if (w4) { if (p.rfx) SP9_W4_LAUNCH(2, true, 1); else SP9_W4_LAUNCH(2, false, 1); }
                    else if (p.rfx) { if (agl >= 4) SP9_AG_LAUNCH(true, 4); else if (agl == 3) SP9_AG_LAUNCH(true, 3); else if (agl == 2) SP9_AG_LAUNCH(true, 2); else SP9_AG_LAUNCH(true, 1); }
                    else { if (agl >= 4) SP9_AG_LAUNCH(false, 4); else if (agl == 3) SP9_AG_LAUNCH(false, 3); else if (agl == 2) SP9_AG_LAUNCH(false, 2); else SP9_AG_LAUNCH(false, 1); }
#undef SP9_AG_LAUNCH
#undef SP9_W4_LAUNCH
                }
                else if (phs == 1) SP9_LAUNCH(1, 1, false, 1);
                else if (phs == 2) SP9_LAUNCH(1, 1, false, 2);
                else if (p.rfx) { if (bm == 128) SP9_LAUNCH(2, 2, true, 0); else SP9_LAUNCH(1, 1, true, 0); }
                else if (bm == 128 && ks2 && ds) {
                    gc_set_max_lds((const void*)gconv_sp9_kernel<2, 2, false, 0, true>, (int)lds_sp);
                    hipLaunchKernelGGL((gconv_sp9_kernel<2, 2, false, 0, true>), grid, dim3(512), lds_sp, st, p);
                }
                else if (bm == 128) { if (ks2) SP9_LAUNCH(2, 2, false, 0); else SP9_LAUNCH(2, 1, false, 0); }
                else { if (ks2 && env_int("HIFIC_SP9_KSPLIT64", 0)) SP9_LAUNCH(1, 2, false, 0); else SP9_LAUNCH(1, 1, false, 0); }
#undef SP9_LAUNCH
                sp_done = true;
            }
        }
    }
    bool mp_done = false;
    if constexpr (std::is_same<T, bf16_t>::value && BC == 64) {
        if (mp) {
            if (p.split) {
                if (lds > 48 * 1024) gc_set_max_lds((const void*)gconv_mp_kernel<true>, (int)lds);
                hipLaunchKernelGGL(gconv_mp_kernel<true>, grid, dim3(256), lds, st, p);
            } else {
                if (lds > 48 * 1024) gc_set_max_lds((const void*)gconv_mp_kernel<false>, (int)lds);
                hipLaunchKernelGGL(gconv_mp_kernel<false>, grid, dim3(256), lds, st, p);
            }
            mp_done = true;
        }
    }
    if (sp_done || mp_done) { /* launched */ }
    else if (bm == 128) GC_LAUNCH(2, 2, 2, 2);
    else if (bm == 64) GC_LAUNCH(2, 2, 1, 2);
    else GC_LAUNCH(1, 4, 1, 1);
#undef GC_LAUNCH
    if (p.ksplit > 1) {
        const long long total = p.kpart_stride;
        int gx = (int)cdivl(total, 256); if (gx > 8192) gx = 8192;
        if (p.out_f32)
            hipLaunchKernelGGL(ksplit_reduce_kernel<float>, dim3(gx), dim3(256), 0, st, p.kpart, p.kpart_stride, p.ksplit, total,
                               p.K, p.OHf * p.OWf, p.bias, p.act, (float*)p.out);
        else
            hipLaunchKernelGGL(ksplit_reduce_kernel<bf16_t>, dim3(gx), dim3(256), 0, st, p.kpart, p.kpart_stride, p.ksplit, total,
                               p.K, p.OHf * p.OWf, p.bias, p.act, (bf16_t*)p.out);
    }
    prof_close(pslot, st);
    return hific_launch_status();
}

template <typename T>
static int launch_gconv_t(GcParams& p, const float* w, const float* w_scale, long long sm, long long sc,
                          long long sr, long long ss, WsAlloc& ws, hipStream_t st) {
    if constexpr (std::is_same<T, float>::value) {
        return launch_gconv_tb<float, 16>(p, w, w_scale, sm, sc, sr, ss, ws, st);
    } else {
        {   // pipelined persistent kernel for the stride-2 layers (gconv_pl.hip); it leaves the plan untouched when it declines
            const GcParams saved = p;
            const int rcp = launch_gconv_pl(p, w, w_scale, sm, sc, sr, ss, ws, st);
            if (rcp != HIFIC_ERR_UNSUPPORTED) return rcp;
            p = saved;
        }
        if (p.C <= 16 || (p.csplit && p.C <= 32)) return launch_gconv_tb<bf16_t, 16>(p, w, w_scale, sm, sc, sr, ss, ws, st);
        return launch_gconv_tb<bf16_t, 64>(p, w, w_scale, sm, sc, sr, ss, ws, st);
    }
}

// ---------------------------------------------------------------------------------------------------
// Few input channels with many taps (the 7x7 convs on 3-channel tensors): an implicit GEMM spends one MFMA K-slice
// (16 deep) per TAP on 3 useful channels.  The horizontal taps are folded into VIRTUAL CHANNELS instead:
//   X'[n][c*nd + j][iy][v] = in[n][c][iy][v + dx_j]     (horizontal pad rule applied here; bf16)
// and the conv runs over X' with one tap per kernel ROW (dy_i, dx = 0): 7 taps x 21 channels instead of 49 taps x 3.
// The expansion costs one pass (44 MB written for a 16 x 3 x 256 x 256 input).
// ---------------------------------------------------------------------------------------------------
template <typename TI>
__global__ void vchan_expand_kernel(const TI* __restrict__ in, bf16_t* __restrict__ X, unsigned N, int C, int IH, int IW,
                                    int OW, int nd, int dx0, int dxs /* dx_j = dx0 + j*dxs */, int bmode) {
    const unsigned total = N * (unsigned)(C * nd * IH * OW);
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int v = (int)(idx % (unsigned)OW);
        unsigned t = idx / (unsigned)OW;
        const int iy = (int)(t % (unsigned)IH); t /= (unsigned)IH;
        const int j = (int)(t % (unsigned)nd); t /= (unsigned)nd;
        const int c = (int)(t % (unsigned)C);
        const unsigned n = t / (unsigned)C;
        int ix = v + dx0 + j * dxs;
        if (bmode == PAD_REFLECT) ix = reflect_idx(ix, IW);
        const bool ok = (unsigned)ix < (unsigned)IW;
        const float val = DT<TI>::ld(in + ((size_t)(n * C + c) * IH + iy) * IW + (ok ? ix : 0));
        X[idx] = f2bf(ok ? val : 0.f);
    }
}

static int launch_gconv_t_bf16(GcParams& p, const float* w, const float* w_scale, long long sm, long long sc,
                               long long sr, long long ss, WsAlloc& ws, hipStream_t st);

// returns HIFIC_ERR_UNSUPPORTED when the layer does not qualify (caller continues with the plain plan)
static int launch_gconv_fewc(GcParams& p, const float* w, const float* w_scale, long long sm, long long sc,
                             long long sr, long long ss, WsAlloc& ws, hipStream_t st) {
    const GcPhase& ph = p.ph[0];
    if (p.C > 4 || p.nphase != 1 || p.ist != 1 || ph.ntaps < 25 || p.csplit || p.msplit || p.split ||      // (pair layout:
        env_int("HIFIC_NO_FEWC", 0))                                                // only the SPLIT kernels form the cross terms)
        return HIFIC_ERR_UNSUPPORTED;
    // the taps must form a full (dy_i) x (dx_j) grid with equally spaced dx
    int dys[16], dxv[16], rs_[16], ss_[16], ny = 0, nx = 0;
    for (int t = 0; t < ph.ntaps; ++t) {
        const int dy = p.tap_dy[t], dx = p.tap_dx[t];
        int i = 0; while (i < ny && dys[i] != dy) ++i;
        if (i == ny) { if (ny == 16) return HIFIC_ERR_UNSUPPORTED; dys[ny] = dy; rs_[ny] = p.tap_r[t]; ++ny; }
        int j = 0; while (j < nx && dxv[j] != dx) ++j;
        if (j == nx) { if (nx == 16) return HIFIC_ERR_UNSUPPORTED; dxv[nx] = dx; ss_[nx] = p.tap_s[t]; ++nx; }
    }
    if (ny * nx != ph.ntaps || p.C * nx > 32 || nx < 2) return HIFIC_ERR_UNSUPPORTED;
    // every (dy_i, dx_j) present with consistent (r, s)
    for (int t = 0; t < ph.ntaps; ++t) {
        int i = 0; while (dys[i] != p.tap_dy[t]) ++i;
        int j = 0; while (dxv[j] != p.tap_dx[t]) ++j;
        if (p.tap_r[t] != rs_[i] || p.tap_s[t] != ss_[j]) return HIFIC_ERR_UNSUPPORTED;
    }
    // sort dx ascending (with their kernel columns) and require a constant step
    for (int a = 0; a < nx; ++a) for (int b = a + 1; b < nx; ++b)
        if (dxv[b] < dxv[a]) { int t1 = dxv[a]; dxv[a] = dxv[b]; dxv[b] = t1; t1 = ss_[a]; ss_[a] = ss_[b]; ss_[b] = t1; }
    const int step = dxv[1] - dxv[0];
    for (int j = 2; j < nx; ++j) if (dxv[j] - dxv[j - 1] != step) return HIFIC_ERR_UNSUPPORTED;
    const int OWv = ph.OWt;                                  // width of the (u, v) output domain
    const size_t xe = (size_t)p.N * p.C * nx * p.IH * OWv;
    const size_t mark = ws.off;
    bf16_t* X = (bf16_t*)ws.take(xe * sizeof(bf16_t));
    if (!X) { ws.off = mark; return HIFIC_ERR_UNSUPPORTED; }
    if (xe >= ((size_t)1 << 31)) { ws.off = mark; return HIFIC_ERR_UNSUPPORTED; }
    if (!ws.plan_out) {
        int gx = (int)((xe + 255) / 256); if (gx > 32768) gx = 32768;
        if (p.in_f32) hipLaunchKernelGGL(vchan_expand_kernel<float>, dim3(gx), dim3(256), 0, st, (const float*)p.in, X, (unsigned)p.N,
                                         p.C, p.IH, p.IW, OWv, nx, dxv[0], step, p.bmode);
        else hipLaunchKernelGGL(vchan_expand_kernel<bf16_t>, dim3(gx), dim3(256), 0, st, (const bf16_t*)p.in, X, (unsigned)p.N,
                                p.C, p.IH, p.IW, OWv, nx, dxv[0], step, p.bmode);
    }
    GcParams q = p;
    q.in = X; q.in_f32 = 0; q.C = p.C * nx; q.IW = OWv; q.csplit = nx;
    for (int j = 0; j < nx; ++j) q.vcol_s[j] = (short)ss_[j];
    int nt = 0;
    for (int i = 0; i < ny; ++i) { q.tap_dy[nt] = (short)dys[i]; q.tap_dx[nt] = 0; q.tap_r[nt] = (short)rs_[i]; q.tap_s[nt] = 0; ++nt; }
    GcPhase& qh = q.ph[0];
    qh.ntaps = nt; qh.tap0 = 0;
    {   // spans of the new tap set (finish_phase semantics)
        int dymin = dys[0], dymax = dys[0];
        for (int i = 1; i < ny; ++i) { if (dys[i] < dymin) dymin = dys[i]; if (dys[i] > dymax) dymax = dys[i]; }
        qh.dy_min = dymin; qh.dx_min = 0; qh.PH = dymax - dymin + 1; qh.PW = 1;
    }
    const int rc = launch_gconv_t_bf16(q, w, w_scale, sm, sc, sr, ss, ws, st);
    if (rc == HIFIC_ERR_UNSUPPORTED) ws.off = mark;
    return rc;
}

// ---------------------------------------------------------------------------------------------------
// Few OUTPUT channels with many taps (the 60->3 7x7 output conv): 3 of the 32 MFMA rows carry work.  The horizontal taps
// become VIRTUAL OUTPUT ROWS: P[n][k*nd + j][u][v'] = sum_{c, i} w[k,c,r_i,s_j] in[c][u + dy_i][v' + dx_0]  (7 taps, 21
// rows, on a domain nd-1 columns wider), then out[n][k][u][v] = act(bias[k] + sum_j P[n][k*nd + j][u][v + j]).
// ---------------------------------------------------------------------------------------------------
template <typename TO>
__global__ void vrow_shift_add_kernel(const float* __restrict__ P, const float* __restrict__ bias, TO* __restrict__ out,
                                      unsigned N, int K, int OH, int OW, int nd, int act) {
    const int PW = OW + nd - 1;
    const unsigned total = N * (unsigned)(K * OH * OW);
    const float slope = act == ACT_RELU ? 0.f : (act == ACT_LEAKY ? 0.2f : 1.f);
    for (unsigned idx = blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += gridDim.x * blockDim.x) {
        const int v = (int)(idx % (unsigned)OW);
        unsigned t = idx / (unsigned)OW;
        const int u = (int)(t % (unsigned)OH); t /= (unsigned)OH;
        const int k = (int)(t % (unsigned)K);
        const unsigned n = t / (unsigned)K;
        const float* src = P + (((size_t)n * K + k) * nd * OH + u) * PW + v;
        float vals[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) vals[j] = src[j < nd ? (size_t)j * OH * PW + j : 0];
        float s = bias ? bias[k] : 0.f;
#pragma unroll
        for (int j = 0; j < 16; ++j) s += j < nd ? vals[j] : 0.f;
        s = s > 0.f ? s : s * slope;
        DT<TO>::st(out + idx, s);
    }
}

static int launch_gconv_fewk(GcParams& p, const float* w, const float* w_scale, long long sm, long long sc,
                             long long sr, long long ss, WsAlloc& ws, hipStream_t st) {
    const GcPhase& ph = p.ph[0];
    if (p.K > 4 || p.C < 16 || p.nphase != 1 || p.ist != 1 || p.ost != 1 || ph.ntaps < 25 || p.csplit || p.msplit || p.split ||
        p.fold_h || p.resid || ph.ooy || ph.oox || ph.OHt != p.OHf || ph.OWt != p.OWf || env_int("HIFIC_NO_FEWK", 0))
        return HIFIC_ERR_UNSUPPORTED;
    int dys[16], dxv[16], rs_[16], ss_[16], ny = 0, nx = 0;
    for (int t = 0; t < ph.ntaps; ++t) {
        const int dy = p.tap_dy[t], dx = p.tap_dx[t];
        int i = 0; while (i < ny && dys[i] != dy) ++i;
        if (i == ny) { if (ny == 16) return HIFIC_ERR_UNSUPPORTED; dys[ny] = dy; rs_[ny] = p.tap_r[t]; ++ny; }
        int j = 0; while (j < nx && dxv[j] != dx) ++j;
        if (j == nx) { if (nx == 16) return HIFIC_ERR_UNSUPPORTED; dxv[nx] = dx; ss_[nx] = p.tap_s[t]; ++nx; }
    }
    if (ny * nx != ph.ntaps || p.K * nx > 32 || nx < 2) return HIFIC_ERR_UNSUPPORTED;
    for (int t = 0; t < ph.ntaps; ++t) {
        int i = 0; while (dys[i] != p.tap_dy[t]) ++i;
        int j = 0; while (dxv[j] != p.tap_dx[t]) ++j;
        if (p.tap_r[t] != rs_[i] || p.tap_s[t] != ss_[j]) return HIFIC_ERR_UNSUPPORTED;
    }
    for (int a = 0; a < nx; ++a) for (int b = a + 1; b < nx; ++b)
        if (dxv[b] < dxv[a]) { int t1 = dxv[a]; dxv[a] = dxv[b]; dxv[b] = t1; t1 = ss_[a]; ss_[a] = ss_[b]; ss_[b] = t1; }
    for (int j = 1; j < nx; ++j) if (dxv[j] - dxv[j - 1] != 1) return HIFIC_ERR_UNSUPPORTED;
    const int PWd = p.OWf + nx - 1;
    const size_t pe = (size_t)p.N * p.K * nx * p.OHf * PWd;
    const size_t mark = ws.off;
    float* P = (float*)ws.take(pe * sizeof(float));
    if (!P || pe >= ((size_t)1 << 31)) { ws.off = mark; return HIFIC_ERR_UNSUPPORTED; }
    GcParams q = p;
    q.K = p.K * nx; q.msplit = nx; q.out = P; q.out_f32 = 1; q.bias = nullptr; q.act = ACT_NONE;
    q.OWf = PWd;
    for (int j = 0; j < nx; ++j) q.vcol_s[j] = (short)ss_[j];
    int nt = 0;
    for (int i = 0; i < ny; ++i) { q.tap_dy[nt] = (short)dys[i]; q.tap_dx[nt] = (short)dxv[0]; q.tap_r[nt] = (short)rs_[i]; q.tap_s[nt] = 0; ++nt; }
    GcPhase& qh = q.ph[0];
    qh.ntaps = nt; qh.tap0 = 0; qh.OWt = PWd;
    {
        int dymin = dys[0], dymax = dys[0];
        for (int i = 1; i < ny; ++i) { if (dys[i] < dymin) dymin = dys[i]; if (dys[i] > dymax) dymax = dys[i]; }
        qh.dy_min = dymin; qh.dx_min = dxv[0]; qh.PH = dymax - dymin + 1; qh.PW = 1;
    }
    const int rc = launch_gconv_t_bf16(q, w, w_scale, sm, sc, sr, ss, ws, st);
    if (rc != HIFIC_OK) { if (rc == HIFIC_ERR_UNSUPPORTED) ws.off = mark; return rc; }
    if (ws.plan_out) return rc;
    const size_t total = (size_t)p.N * p.K * p.OHf * p.OWf;
    int gx = (int)((total + 255) / 256); if (gx > 32768) gx = 32768;
    if (p.out_f32) hipLaunchKernelGGL(vrow_shift_add_kernel<float>, dim3(gx), dim3(256), 0, st, P, p.bias, (float*)p.out, (unsigned)p.N,
                                      p.K, p.OHf, p.OWf, nx, p.act);
    else hipLaunchKernelGGL(vrow_shift_add_kernel<bf16_t>, dim3(gx), dim3(256), 0, st, P, p.bias, (bf16_t*)p.out, (unsigned)p.N,
                            p.K, p.OHf, p.OWf, nx, p.act);
    return hific_launch_status();
}

static int launch_gconv_t_bf16(GcParams& p, const float* w, const float* w_scale, long long sm, long long sc,
                               long long sr, long long ss, WsAlloc& ws, hipStream_t st) {
    return launch_gconv_t<bf16_t>(p, w, w_scale, sm, sc, sr, ss, ws, st);
}


// Few-channel forward convolutions on the virtual-column kernel (gconv_vc_kernel).  HIFIC_ERR_UNSUPPORTED: not this layer.
static int launch_gconv_vc(GcParams& p, const float* w, const float* w_scale, long long sm, long long sc,
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
        hipLaunchKernelGGL((pack_w2_kernel<bf16_t, 0>), dim3(job.gx, job.gy), dim3(256), job.lds_bytes, st, q, w, w_scale, job.sm,
                           job.sc, job.RS, job.MB);
    }
    dim3 grid(p.max_tiles * (p.Kpad / 64), 1, 1);
    char ptag[112];
    snprintf(ptag, sizeof(ptag), "gconv_vc K%d C%d N%d in%dx%d out%dx%d taps%d ist%d tile%dx%dx%d J%d grid%d", p.K, p.C, p.N, p.IH,
             p.IW, p.OHf, p.OWf, T, p.ist, p.NI, p.TH, p.TW, J, (int)grid.x);
    const int pslot = prof_open("gconv_vc_kernel", p.aflops, st, ptag);
    if (p.in_f32) {
        if (lds > 48 * 1024) gc_set_max_lds((const void*)gconv_vc_kernel<true>, (int)lds);
        hipLaunchKernelGGL(gconv_vc_kernel<true>, grid, dim3(256), lds, st, p);
    } else {
        if (lds > 48 * 1024) gc_set_max_lds((const void*)gconv_vc_kernel<false>, (int)lds);
        hipLaunchKernelGGL(gconv_vc_kernel<false>, grid, dim3(256), lds, st, p);
    }
    prof_close(pslot, st);
    return hific_launch_status();
}

static int launch_gconv(GcParams& p, int dtype, const float* w, const float* w_scale, long long sm, long long sc,
                        long long sr, long long ss, WsAlloc& ws, hipStream_t st) {
    if (dtype == HIFIC_BF16 && !p.oscale) {
        int rcv = launch_gconv_vc(p, w, w_scale, sm, sc, sr, ss, ws, st);
        if (rcv != HIFIC_ERR_UNSUPPORTED) return rcv;
        int rc = launch_gconv_fewc(p, w, w_scale, sm, sc, sr, ss, ws, st);
        if (rc != HIFIC_ERR_UNSUPPORTED) return rc;
        rc = launch_gconv_fewk(p, w, w_scale, sm, sc, sr, ss, ws, st);
        if (rc != HIFIC_ERR_UNSUPPORTED) return rc;
    }
    if (dtype == HIFIC_F32) { p.in_f32 = 1; p.out_f32 = 1; return launch_gconv_t<float>(p, w, w_scale, sm, sc, sr, ss, ws, st); }
    if (dtype == HIFIC_BF16) return launch_gconv_t<bf16_t>(p, w, w_scale, sm, sc, sr, ss, ws, st);
    return HIFIC_ERR_ARG;
}

static void add_tap(GcParams& p, int& nt, int dy, int dx, int r, int s) {
    p.tap_dy[nt] = (short)dy; p.tap_dx[nt] = (short)dx; p.tap_r[nt] = (short)r; p.tap_s[nt] = (short)s; ++nt;
}

int gc_conv_fwd(const ConvGeom& g, const void* x, const float* w, const float* w_scale, const float* bias,
                void* y, const void* resid, int act, int dtype, int in_f32, int out_f32,
                WsAlloc& ws, hipStream_t st) {
    if (g.R * g.S > GC_MAXTAPS) return HIFIC_ERR_UNSUPPORTED;
    GcParams p; memset(&p, 0, sizeof(p));
    p.in = x; p.bias = bias; p.out = y; p.resid = resid;
    p.N = g.N; p.C = g.C; p.IH = g.H; p.IW = g.W; p.K = g.K; p.OHf = g.OH(); p.OWf = g.OW();
    p.ist = g.stride; p.ost = 1; p.bmode = g.pad_mode; p.act = act; p.in_f32 = in_f32; p.out_f32 = out_f32;
    p.nphase = 1;
    int nt = 0;
    for (int r = 0; r < g.R; ++r) for (int s = 0; s < g.S; ++s) add_tap(p, nt, r - g.pt, s - g.pl, r, s);
    GcPhase& ph = p.ph[0];
    ph.ntaps = nt; ph.tap0 = 0; ph.ooy = 0; ph.oox = 0; ph.OHt = p.OHf; ph.OWt = p.OWf;
    finish_phase(ph, p);
    const long long RS = (long long)g.R * g.S;
    p.split = g.red_split == 2;
    p.oscale = g.oscale;
    const double cred = g.red_split == 2 ? (g.red_C > 0 ? g.red_C : g.C / 2) : (g.red_split ? g.C / 3.0 : g.C);
    p.aflops = 2.0 * g.K * cred * (double)RS * g.N * g.OH() * g.OW();
    return launch_gconv(p, dtype, w, w_scale, (long long)g.C * RS, RS, g.S, 1, ws, st);
}

// data gradient of a strided conv: stride-phase decomposition over the padded input domain
int gc_conv_bwd_data(const ConvGeom& g, const void* dy, const float* w, const float* w_scale, void* dx,
                     int dtype, int in_f32, int out_f32, WsAlloc& ws, hipStream_t st) {
    const int stv = g.stride;
    if (stv * stv > GC_MAXPH || g.R * g.S > GC_MAXTAPS) return HIFIC_ERR_UNSUPPORTED;
    const bool has_pad = (g.pt | g.pl | g.pb | g.pr) != 0;
    const bool fold = has_pad && g.pad_mode == PAD_REFLECT;
    const int Hp = g.H + g.pt + g.pb, Wp = g.W + g.pl + g.pr;
    // Reflect-padded 3x3 stride-1 layers (the residual blocks: 69 % of the model's MACs): gather form on the un-padded
    // domain through the software-pipelined kernel (see gconv_sp9_kernel RFX).  Falls through to the padded-domain
    // route when the plan cannot use that kernel.
    if (fold && dtype == HIFIC_BF16 && g.R == 3 && g.S == 3 && stv == 1 && g.pt == 1 && g.pl == 1 && g.pb == 1 && g.pr == 1 &&
        g.H >= 4 && g.W >= 4 && g.C > 32 && g.K > 16 && !g.oscale && !env_int("HIFIC_NO_RFX", 0)) {
        const size_t ws_mark = ws.off;
        const size_t e_elems = (size_t)g.N * g.K * (g.H + 2) * (g.W + 2);
        bf16_t* E = (bf16_t*)ws.take(e_elems * sizeof(bf16_t));
        if (E) {
            const unsigned planes = (unsigned)(g.N * g.K);
            int gx = (int)((e_elems + 255) / 256); if (gx > 16384) gx = 16384;
            if (ws.plan_out) { /* plan-only: nothing is launched */ }
            else if (in_f32) hipLaunchKernelGGL(reflect_extend_kernel<float>, dim3(gx), dim3(256), 0, st, (const float*)dy, E, planes, g.H, g.W);
            else if (g.W == 16 && ((size_t)dy & 15) == 0 && !env_int("HIFIC_NO_EXTEND_ROWS", 0)) {
                const unsigned nrows = planes * (unsigned)(g.H + 2);
                hipLaunchKernelGGL(reflect_extend_rows_kernel<16>, dim3((nrows + 255) / 256), dim3(256), 0, st, (const bf16_t*)dy, E, planes, g.H, g.W);
            }
            else hipLaunchKernelGGL(reflect_extend_kernel<bf16_t>, dim3(gx), dim3(256), 0, st, (const bf16_t*)dy, E, planes, g.H, g.W);
            GcParams q; memset(&q, 0, sizeof(q));
            q.in = E; q.out = dx; q.rfx = 1;
            q.N = g.N; q.C = g.K; q.IH = g.H + 2; q.IW = g.W + 2; q.K = g.C; q.OHf = g.H; q.OWf = g.W;
            q.ist = 1; q.ost = 1; q.bmode = PAD_ZERO; q.act = ACT_NONE; q.in_f32 = 0; q.out_f32 = out_f32;
            q.nphase = 1;
            int nt = 0;
            for (int r = 0; r < 3; ++r) for (int s2 = 0; s2 < 3; ++s2) add_tap(q, nt, 2 - r, 2 - s2, r, s2);   // r-major: the kernel decodes r = t/3, s = t%3
            GcPhase& ph = q.ph[0];
            ph.ntaps = nt; ph.tap0 = 0; ph.ooy = 0; ph.oox = 0; ph.OHt = g.H; ph.OWt = g.W;
            finish_phase(ph, q);
            q.aflops = 2.0 * g.K * g.C * 9.0 * g.N * g.H * g.W;
            const int rc = launch_gconv(q, dtype, w, w_scale, 9, (long long)g.C * 9, 3, 1, ws, st);
            if (rc != HIFIC_ERR_UNSUPPORTED) return rc;
        }
        ws.off = ws_mark;
    }
    GcParams p; memset(&p, 0, sizeof(p));
    p.in = dy; p.bias = nullptr; p.resid = nullptr; p.oscale = g.oscale;
    p.N = g.N; p.C = g.K; p.IH = g.OH(); p.IW = g.OW(); p.K = g.C;
    p.ist = 1; p.ost = stv; p.bmode = PAD_ZERO; p.act = ACT_NONE; p.in_f32 = in_f32;
    float* padbuf = nullptr;
    if (fold) {
        padbuf = (float*)ws.take((size_t)g.N * g.C * Hp * Wp * sizeof(float));
        if (!padbuf) return HIFIC_ERR_WS;
        p.out = padbuf; p.out_f32 = 1; p.OHf = Hp; p.OWf = Wp;
        if (!env_int("HIFIC_NO_FOLD_ROUTE", 0)) {
            p.out2 = dx; p.out2_f32 = (dtype == HIFIC_F32 || out_f32) ? 1 : 0;
            p.fold_pt = g.pt; p.fold_pl = g.pl; p.fold_h = g.H; p.fold_w = g.W;
        }
    } else {
        p.out = dx; p.out_f32 = out_f32; p.OHf = g.H; p.OWf = g.W;
    }
    int nt = 0, np = 0;
    for (int py = 0; py < stv; ++py) for (int px = 0; px < stv; ++px) {
        GcPhase& ph = p.ph[np];
        ph.tap0 = nt;
        for (int r = 0; r < g.R; ++r) {
            if ((r - py) % stv != 0) continue;
            for (int s = 0; s < g.S; ++s) {
                if ((s - px) % stv != 0) continue;
                add_tap(p, nt, (py - r) / stv, (px - s) / stv, r, s);
            }
        }
        ph.ntaps = nt - ph.tap0;
        ph.OHt = (Hp - py + stv - 1) / stv; ph.OWt = (Wp - px + stv - 1) / stv;
        if (ph.OHt <= 0 || ph.OWt <= 0) { nt = ph.tap0; continue; }
        ph.ooy = fold ? py : py - g.pt; ph.oox = fold ? px : px - g.pl;
        finish_phase(ph, p);
        ++np;
    }
    p.nphase = np;
    const long long RS = (long long)g.R * g.S;
    p.aflops = 2.0 * g.K * g.C * (double)RS * g.N * g.OH() * g.OW();       // same MACs as the forward op
    // out-channel m = c (stride RS), reduction channel = k (stride C*RS)
    int rc = launch_gconv(p, dtype, w, w_scale, RS, (long long)g.C * RS, g.S, 1, ws, st);
    if (rc != HIFIC_OK || ws.plan_out) return rc;
    if (fold) {
        const long long planes = (long long)g.N * g.C;
        long long total = planes * g.H * g.W;
        int gx = (int)((total + 255) / 256); if (gx > 16384) gx = 16384;
        const bool of32 = (dtype == HIFIC_F32) || out_f32;
        if (p.fold_h) {
            const bool whole = (g.H < g.pt + g.pb + 3) || (g.W < g.pl + g.pr + 3);
            const long long nrim = whole ? (long long)g.H * g.W
                                         : (long long)(g.pt + g.pb) * g.W + (long long)(g.H - g.pt - g.pb) * (g.pl + g.pr);
            gx = (int)((planes * nrim + 255) / 256); if (gx > 16384) gx = 16384; if (gx < 1) gx = 1;
            if (of32)
                hipLaunchKernelGGL(reflect_rim_add_kernel<float>, dim3(gx), dim3(256), 0, st, padbuf, (float*)dx, planes,
                                   g.H, g.W, g.pt, g.pl, g.pb, g.pr);
            else
                hipLaunchKernelGGL(reflect_rim_add_kernel<bf16_t>, dim3(gx), dim3(256), 0, st, padbuf, (bf16_t*)dx, planes,
                                   g.H, g.W, g.pt, g.pl, g.pb, g.pr);
        } else if (of32)
            hipLaunchKernelGGL(reflect_fold_kernel<float>, dim3(gx), dim3(256), 0, st, padbuf, (float*)dx, planes,
                               g.H, g.W, g.pt, g.pl, g.pb, g.pr);
        else
            hipLaunchKernelGGL(reflect_fold_kernel<bf16_t>, dim3(gx), dim3(256), 0, st, padbuf, (bf16_t*)dx, planes,
                               g.H, g.W, g.pt, g.pl, g.pb, g.pr);
        return hific_launch_status();
    }
    return HIFIC_OK;
}

int gc_convT_fwd(const ConvTGeom& g, const void* x, const float* w, const float* bias, void* y, int act,
                 int dtype, int in_f32, int out_f32, WsAlloc& ws, hipStream_t st) {
    const int stv = g.stride;
    if (stv * stv > GC_MAXPH || g.R * g.S > GC_MAXTAPS) return HIFIC_ERR_UNSUPPORTED;
    GcParams p; memset(&p, 0, sizeof(p));
    p.in = x; p.bias = bias; p.out = y; p.resid = nullptr;
    p.N = g.N; p.C = g.Ci; p.IH = g.H; p.IW = g.W; p.K = g.Co; p.OHf = g.OH(); p.OWf = g.OW();
    p.ist = 1; p.ost = stv; p.bmode = PAD_ZERO; p.act = act; p.in_f32 = in_f32; p.out_f32 = out_f32;
    int nt = 0, np = 0;
    for (int py = 0; py < stv; ++py) for (int px = 0; px < stv; ++px) {
        GcPhase& ph = p.ph[np];
        ph.tap0 = nt;
        for (int r = 0; r < g.R; ++r) {
            if ((py + g.pad - r) % stv != 0) continue;
            for (int s = 0; s < g.S; ++s) {
                if ((px + g.pad - s) % stv != 0) continue;
                // floor-exact because the numerator is a multiple of stride
                add_tap(p, nt, (py + g.pad - r) / stv, (px + g.pad - s) / stv, r, s);
            }
        }
        ph.ntaps = nt - ph.tap0;
        ph.OHt = (p.OHf - py + stv - 1) / stv; ph.OWt = (p.OWf - px + stv - 1) / stv;
        if (ph.OHt <= 0 || ph.OWt <= 0) { nt = ph.tap0; continue; }
        ph.ooy = py; ph.oox = px;
        finish_phase(ph, p);
        ++np;
    }
    p.nphase = np;
    const long long RS = (long long)g.R * g.S;
    p.split = g.red_split == 2;
    const double cred = g.red_split == 2 ? (g.red_C > 0 ? g.red_C : g.Ci / 2) : (g.red_split ? g.Ci / 3.0 : g.Ci);
    p.aflops = 2.0 * cred * g.Co * (double)RS * g.N * g.H * g.W;   // every (input pixel, tap) pair once
    // w[ci][co][r][s]: m = co (stride RS), reduction channel ci (stride Co*RS)
    return launch_gconv(p, dtype, w, nullptr, RS, (long long)g.Co * RS, g.S, 1, ws, st);
}

int gc_convT_bwd_data(const ConvTGeom& g, const void* dy, const float* w, void* dx, int dtype, int in_f32,
                      int out_f32, WsAlloc& ws, hipStream_t st) {
    if (g.R * g.S > GC_MAXTAPS) return HIFIC_ERR_UNSUPPORTED;
    GcParams p; memset(&p, 0, sizeof(p));
    p.in = dy; p.out = dx;
    p.N = g.N; p.C = g.Co; p.IH = g.OH(); p.IW = g.OW(); p.K = g.Ci; p.OHf = g.H; p.OWf = g.W;
    p.ist = g.stride; p.ost = 1; p.bmode = PAD_ZERO; p.act = ACT_NONE; p.in_f32 = in_f32; p.out_f32 = out_f32;
    p.nphase = 1;
    int nt = 0;
    for (int r = 0; r < g.R; ++r) for (int s = 0; s < g.S; ++s) add_tap(p, nt, r - g.pad, s - g.pad, r, s);
    GcPhase& ph = p.ph[0];
    ph.ntaps = nt; ph.tap0 = 0; ph.OHt = g.H; ph.OWt = g.W;
    finish_phase(ph, p);
    const long long RS = (long long)g.R * g.S;
    p.aflops = 2.0 * g.Ci * g.Co * (double)RS * g.N * g.H * g.W;
    // m = ci (stride Co*RS), reduction channel co (stride RS)
    return launch_gconv(p, dtype, w, nullptr, (long long)g.Co * RS, RS, g.S, 1, ws, st);
}

// partial planes ws[split][m][tap][c] -> the weight-gradient layout (shared by the generic, pipelined and stride-2 kernels)
static int wgrad_finish(WgParams& p, float* dw, long long sm, long long sc, long long sr, long long ss, int accumulate,
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


// Stride-1 3x3 pad-1 bf16 layers on 16-pixel-multiple planes -> wgrad_s1_kernel; HIFIC_ERR_UNSUPPORTED when the layer does not qualify
static int launch_wgrad_s1(WgParams& p, float* dw, long long sm, long long sc, long long sr, long long ss, int accumulate,
                           WsAlloc& ws, hipStream_t st) {
    if (p.ist != 1 || p.a_f32 || p.b_f32 || p.ntaps != 9 || !env_int("HIFIC_WGRAD_S1", 1)) return HIFIC_ERR_UNSUPPORTED;
    if (p.AW % 16 != 0 || p.BW != p.AW || p.BH != p.AH || p.BH < 2 || (((size_t)p.a | (size_t)p.b) & 15) != 0) return HIFIC_ERR_UNSUPPORTED;
    if ((long long)p.N * p.M * p.AH * p.AW >= (1ll << 32) || (long long)p.N * p.C * p.BH * p.BW >= (1ll << 32)) return HIFIC_ERR_UNSUPPORTED;
    for (int t = 0; t < 9; ++t)
        if (p.tap_r[t] != t / 3 || p.tap_s[t] != t % 3 || p.tap_dy[t] != t / 3 - 1 || p.tap_dx[t] != t % 3 - 1) return HIFIC_ERR_UNSUPPORTED;
    p.Mpad = cdiv(p.M, 64) * 64; p.Cpad = cdiv(p.C, 64) * 64;
    p.ngroups = 1; p.TH = 8; p.TW = 16; p.NI = 1;
    p.tiles_y = cdiv(p.AH, p.TH); p.tiles_x = p.AW / 16; p.tiles_n = p.N;
    p.ntiles = p.tiles_n * p.tiles_y * p.tiles_x;
    const int mt = p.Mpad / 64, ct = p.Cpad / 64, base_blocks = mt * ct;
    int nsplit = 1;
    if (base_blocks < env_int("HIFIC_WG_NOSPLIT", 160)) nsplit = env_int("HIFIC_WGS1_TARGET", 256) / base_blocks;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > p.ntiles) nsplit = p.ntiles;
    p.tiles_per_split = cdiv(p.ntiles, nsplit);
    p.nsplit = cdiv(p.ntiles, p.tiles_per_split);
    p.direct = p.nsplit == 1;
    p.dw = dw; p.sm = sm; p.sc = sc; p.sr = sr; p.ss = ss; p.accumulate = accumulate;
    if (!p.direct) {
        // the split-partial planes must fit the caller's workspace: fewer splits first, and if even two do not fit the launcher
        // declines (nothing has been launched or written yet) and the generic weight-gradient path takes the layer
        const size_t ws_mark = ws.off;
        for (;;) {
            p.ws = (float*)ws.take((size_t)p.nsplit * p.Mpad * p.ntaps * p.Cpad * sizeof(float));
            if (p.ws || p.nsplit <= 2) break;
            ws.off = ws_mark;
            p.tiles_per_split = cdiv(p.ntiles, p.nsplit / 2);
            p.nsplit = cdiv(p.ntiles, p.tiles_per_split);
        }
        if (!p.ws) { ws.off = ws_mark; return HIFIC_ERR_UNSUPPORTED; }
    }
    p.xcd_remap = ((ct * p.nsplit) % 8 == 0) && env_int("HIFIC_WGS1_XCD", 1);
    const int grid = base_blocks * p.nsplit;
    char ptag[112];
    snprintf(ptag, sizeof(ptag), "wgrad_s1 M%d C%d N%d a%dx%d split%d grid%d", p.M, p.C, p.N, p.AH, p.AW, p.nsplit, grid);
    const int pslot = prof_open("wgrad_s1_kernel", 2.0 * p.M * p.C * 9 * (double)p.N * p.AH * p.AW, st, ptag);
    size_t lds = 2 * (size_t)64 * 272 + 2 * (size_t)64 * (10 * 32 + 16) + 2 * (size_t)10 * 64 * 4;
    const size_t epi = (size_t)4 * 16 * (32 * 9 + 1) * sizeof(float);
    if (lds < epi) lds = epi;
    gc_set_max_lds((const void*)wgrad_s1_kernel, (int)lds);
    hipLaunchKernelGGL(wgrad_s1_kernel, dim3(grid), dim3(512), lds, st, p);
    prof_close(pslot, st);
    int rc = hific_launch_status();
    if (rc != HIFIC_OK || p.direct) return rc;
    return wgrad_finish(p, dw, sm, sc, sr, ss, accumulate, ws, st);
}

// Stride-2 3x3 / 4x4 bf16 layers -> wgrad_s2_kernel; HIFIC_ERR_UNSUPPORTED (nothing launched) when the layer does not qualify
static int launch_wgrad_s2(WgParams& p, float* dw, long long sm, long long sc, long long sr, long long ss, int accumulate,
                           WsAlloc& ws, hipStream_t st) {
    if (p.ist != 2 || p.a_f32 || p.b_f32 || !env_int("HIFIC_WGRAD_S2", 1)) return HIFIC_ERR_UNSUPPORTED;
    if (p.AW % 16 != 0 || p.BW < 2 * p.AW || p.BW % 8 != 0 || (((size_t)p.a | (size_t)p.b) & 15) != 0) return HIFIC_ERR_UNSUPPORTED;
    // (the kernel addresses both operands with 32-bit element offsets)
    if ((long long)p.N * p.M * p.AH * p.AW >= (1ll << 32) || (long long)p.N * p.C * p.BH * p.BW >= (1ll << 32)) return HIFIC_ERR_UNSUPPORTED;
    int R = 0, S = 0;
    for (int t = 0; t < p.ntaps; ++t) { if (p.tap_r[t] + 1 > R) R = p.tap_r[t] + 1; if (p.tap_s[t] + 1 > S) S = p.tap_s[t] + 1; }
    if (R * S != p.ntaps) return HIFIC_ERR_UNSUPPORTED;
    const int PT = -p.tap_dy[0], PL = -p.tap_dx[0];
    for (int t = 0; t < p.ntaps; ++t)
        if (p.tap_r[t] != t / S || p.tap_s[t] != t % S || p.tap_dy[t] != p.tap_r[t] - PT || p.tap_dx[t] != p.tap_s[t] - PL)
            return HIFIC_ERR_UNSUPPORTED;
    int kind = -1;
    if (R == 3 && S == 3 && PT == 1 && PL == 0) kind = 0;            // Encoder convs (reflect pad top 1 / right 1)
    else if (R == 3 && S == 3 && PT == 1 && PL == 1) kind = 1;       // Generator conv-transposes
    else if (R == 4 && S == 4 && PT == 1 && PL == 1) kind = 2;       // Discriminator convs
    if (kind < 0) return HIFIC_ERR_UNSUPPORTED;
    // the last input row / column a tile touches must exist or be produced by the padding rule: reflect needs index <= 2(H-1)
    if (p.bmode == PAD_REFLECT && (2 * p.AH + R - 2 - PT > 2 * (p.BH - 1) || 2 * p.AW + S - 2 - PL > 2 * (p.BW - 1) || PT > p.BH - 1))
        return HIFIC_ERR_UNSUPPORTED;
    p.Mpad = cdiv(p.M, 64) * 64; p.Cpad = cdiv(p.C, 64) * 64;
    p.ngroups = 1; p.TH = 4; p.TW = 16; p.NI = 1;
    p.tiles_y = cdiv(p.AH, p.TH); p.tiles_x = p.AW / 16; p.tiles_n = p.N;
    p.ntiles = p.tiles_n * p.tiles_y * p.tiles_x;
    const int mt = p.Mpad / 64, ct = p.Cpad / 64, base_blocks = mt * ct;
    // one workgroup (8 waves, 110-140 KB of LDS) per CU: split the pixels until the grid covers the 256 CUs once
    int nsplit = env_int("HIFIC_WGS2_TARGET", 256) / base_blocks;
    if (nsplit < 1) nsplit = 1;
    if (nsplit > p.ntiles) nsplit = p.ntiles;
    p.tiles_per_split = cdiv(p.ntiles, nsplit);
    p.nsplit = cdiv(p.ntiles, p.tiles_per_split);
    p.direct = p.nsplit == 1;
    p.dw = dw; p.sm = sm; p.sc = sc; p.sr = sr; p.ss = ss; p.accumulate = accumulate;
    if (!p.direct) {
        // the split-partial planes must fit the caller's workspace: fewer splits first, and if even two do not fit the launcher
        // declines (nothing has been launched or written yet) and the generic weight-gradient path takes the layer
        const size_t ws_mark = ws.off;
        for (;;) {
            p.ws = (float*)ws.take((size_t)p.nsplit * p.Mpad * p.ntaps * p.Cpad * sizeof(float));
            if (p.ws || p.nsplit <= 2) break;
            ws.off = ws_mark;
            p.tiles_per_split = cdiv(p.ntiles, p.nsplit / 2);
            p.nsplit = cdiv(p.ntiles, p.tiles_per_split);
        }
        if (!p.ws) { ws.off = ws_mark; return HIFIC_ERR_UNSUPPORTED; }
    }
    p.xcd_remap = ((ct * p.nsplit) % 8 == 0) && env_int("HIFIC_WGS2_XCD", 1);
    const int grid = base_blocks * p.nsplit;
    char ptag[112];
    snprintf(ptag, sizeof(ptag), "wgrad_s2 M%d C%d N%d a%dx%d taps%d split%d grid%d", p.M, p.C, p.N, p.AH, p.AW, p.ntaps, p.nsplit, grid);
    const int pslot = prof_open("wgrad_s2_kernel", 2.0 * p.M * p.C * p.ntaps * (double)p.N * p.AH * p.AW, st, ptag);
#define WGS2_LAUNCH(R_, S_, PT_, PL_)                                                                          \
    do {                                                                                                       \
        const size_t lds = 2 * (size_t)64 * 144 + 2 * (size_t)64 * S2Cfg<R_, S_, PT_, PL_>::cp();              \
        gc_set_max_lds((const void*)wgrad_s2_kernel<R_, S_, PT_, PL_>, (int)lds);                              \
        hipLaunchKernelGGL((wgrad_s2_kernel<R_, S_, PT_, PL_>), dim3(grid), dim3(512), lds, st, p);            \
    } while (0)
    if (kind == 0) WGS2_LAUNCH(3, 3, 1, 0);
    else if (kind == 1) WGS2_LAUNCH(3, 3, 1, 1);
    else WGS2_LAUNCH(4, 4, 1, 1);
#undef WGS2_LAUNCH
    prof_close(pslot, st);
    int rc = hific_launch_status();
    if (rc != HIFIC_OK || p.direct) return rc;
    return wgrad_finish(p, dw, sm, sc, sr, ss, accumulate, ws, st);
}

// ---------------------------------------------------------------------------------------------------
template <typename T>
static int launch_wgrad_t(WgParams& p, float* dw, long long sm, long long sc, long long sr, long long ss,
                          int accumulate, WsAlloc& ws, hipStream_t st) {
    using Cfg = WgCfg<T>;
    if constexpr (std::is_same<T, bf16_t>::value) {
        int rc = launch_wgrad_s2(p, dw, sm, sc, sr, ss, accumulate, ws, st);
        if (rc != HIFIC_ERR_UNSUPPORTED) return rc;
        rc = launch_wgrad_s1(p, dw, sm, sc, sr, ss, accumulate, ws, st);
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
    if (!choose_tile(p.N, p.AH, p.AW, p.ist, span_y, span_x, Cfg::PITCH, fixed, bigstage ? kLdsBudget : 72 * 1024,
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
    const int pslot = prof_open(pipe ? "wgrad_pipe_kernel" : (std::is_same<T, float>::value ? "wgrad_kernel<f32>" : "wgrad_kernel<bf16>"),
                                2.0 * p.M * p.C * p.ntaps * (double)p.N * p.AH * p.AW, st, ptag);
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
    prof_close(pslot, st);
    int rc = hific_launch_status();
    if (rc != HIFIC_OK || p.direct) return rc;
    return wgrad_finish(p, dw, sm, sc, sr, ss, accumulate, ws, st);
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
    int nsplit = cdiv(env_int("HIFIC_IM2COL_TARGET", 512), base_blocks);       // two workgroups per CU: one full round
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
    const int pslot = prof_open(std::is_same<T, float>::value ? "wgrad_im2col_kernel<f32>" : "wgrad_im2col_kernel<bf16>",
                                2.0 * g.K * g.C * nt * (double)g.N * g.OH() * g.OW(), st, ptag);
    hipLaunchKernelGGL(kfn, grid, dim3(256), lds, st, p);
    prof_close(pslot, st);
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
        const int rc = launch_wgrad_s2(q, dw, (long long)g.C * RSq, RSq, g.S, 1, accumulate, ws, st);
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

// conservative workspace bound for any op on this layer (packed weights + padded-grad buffer + wgrad partials)
size_t gc_ws_bytes_conv(const ConvGeom& g, int dtype) {
    const size_t es = dtype == HIFIC_F32 ? 4 : 2;
    const size_t kp = (size_t)cdiv(g.K, 128) * 128 + 128, cp = (size_t)cdiv(g.C, 64) * 64 + 64;
    size_t packed = kp * cp * g.R * g.S * es;
    size_t padbuf = (size_t)g.N * g.C * (g.H + g.pt + g.pb) * (g.W + g.pl + g.pr) * 4;
    size_t part = kp * cp * g.R * g.S * 4 * 32 + (size_t)1100 * 64 * 64 * 9 * 4;   // stage-2 groups + <= ~1100 block partials
    return packed + padbuf + part + 4096;
}
size_t gc_ws_bytes_convT(const ConvTGeom& g, int dtype) {
    const size_t es = dtype == HIFIC_F32 ? 4 : 2;
    const size_t kp = (size_t)cdiv(g.Co, 128) * 128 + 128, cp = (size_t)cdiv(g.Ci, 64) * 64 + 64;
    size_t packed = kp * cp * g.R * g.S * es;
    size_t part = kp * cp * g.R * g.S * 4 * 32 + (size_t)1100 * 64 * 64 * 9 * 4;
    return packed + part + 4096;
}
