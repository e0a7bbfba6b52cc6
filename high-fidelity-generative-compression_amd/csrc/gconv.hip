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
#include "gconv_stage.h"
#include <type_traits>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>

// build-time A/B switches (csrc/build.sh EXTRA=-D...; lib.py loads $HIFIC_LIB_PATH when set)
#ifndef GC_TOFF_EARLY
#define GC_TOFF_EARLY 0     // gconv_kernel: tap offsets of a step read from LDS before the step's barriers - measured
                            // (round 3, A/B libraries in one run): 1-3 % SLOWER on every generic instantiation: off
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
#define PROF_MAXKINDS 48
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
static double g_prof_bytes[PROF_MAX];    // algorithmic HBM bytes of the launch (operands read once + result written once); 0 = not given
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
    g_prof_kind[i] = k; g_prof_flops[i] = flops; g_prof_bytes[i] = 0.0;
    strncpy(g_prof_tag[i], tag, sizeof(g_prof_tag[i]) - 1); g_prof_tag[i][sizeof(g_prof_tag[i]) - 1] = 0;
    hipEventRecord(g_prof_ev[i][0], st);
    return i;
}
static void prof_close(int i, hipStream_t st) { if (i >= 0) hipEventRecord(g_prof_ev[i][1], st); }
int gc_prof_open(const char* kname, double flops, hipStream_t st, const char* tag) { return prof_open(kname, flops, st, tag); }
void gc_prof_close(int slot, hipStream_t st) { prof_close(slot, st); }
void gc_prof_bytes(int slot, double bytes) { if (slot >= 0) g_prof_bytes[slot] = bytes; }
double gc_algo_bytes(const GcParams& p) {
    int nt = 0;
    for (int i = 0; i < p.nphase; ++i) nt += p.ph[i].ntaps;
    return (double)p.N * p.C * p.IH * p.IW * (p.in_f32 ? 4.0 : 2.0) + (double)p.N * p.K * p.OHf * p.OWf * (p.out_f32 ? 4.0 : 2.0) +
           (double)p.K * p.C * nt * 2.0;
}
// Algorithmic bytes per kernel function of the profile in progress (same order as hific_prof_end, which must be called AFTER this)
extern "C" int hific_prof_bytes(int max_kinds, double* bytes) {
    const int nk = g_prof_nk < max_kinds ? g_prof_nk : max_kinds;
    for (int k = 0; k < nk; ++k) bytes[k] = 0;
    for (int i = 0; i < g_prof_n; ++i) if (g_prof_kind[i] < nk) bytes[g_prof_kind[i]] += g_prof_bytes[i];
    return nk;
}

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


// choose (TH, TW, NI) for a (u,v) domain: exhaustive search over tile shapes (not only powers of two, so padded
// data-gradient domains such as 18x18 tile as 7x18 instead of 8x16) minimising a cost model
//   tiles * (128 pixels * ntaps MFMA work + staging of the halo patch)
// subject to the LDS budget.  need16: pixels per tile must be a multiple of 16 (weight-gradient K-slice).
static bool choose_tile(int N, int OHt, int OWt, int ist, int span_y, int span_x, int pitch, int fixed_bytes,
                        int pref_budget, int& TH, int& TW, int& NI, int ntaps = 9) {
    return gc_choose_tile(N, OHt, OWt, ist, span_y, span_x, pitch, fixed_bytes, pref_budget, TH, TW, NI, ntaps, false);
}
bool gc_choose_tile(int N, int OHt, int OWt, int ist, int span_y, int span_x, int pitch, int fixed_bytes,
                    int pref_budget, int& TH, int& TW, int& NI, int ntaps, bool need16) {
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
        tiled = g >= env_int("HIFIC_GC_BIGTILE_MIN_GRID", 64);     // (round 5: 256 -> 64, the hyperprior 5x5 layers: 127 -> 94, 77 -> 71 us)
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
        // ... as 16-byte pieces through LDS (gconv_mp_kernel epi_wide 4): bf16 output, 4-pixel groups inside a tile row and the
        // image, 16-byte aligned rows; needs 4 x 32 x 288 bytes of LDS after the main loop
        if (p.epi_wide == 2 && !p.out_f32 && p.TW % 4 == 0 && p.OWf % 8 == 0 && (((size_t)p.out) & 15) == 0 &&
            p.ph[0].OWt % 4 == 0 && p.ph[2].OWt % 4 == 0 && (long long)p.N * p.K * p.OHf * p.OWf < (1ll << 31) &&
            env_int("HIFIC_MP_WIDE", 1)) {
            p.epi_wide = 4;
            const size_t need = (size_t)4 * 32 * 288;
            if (need > lds) lds = need;
        }
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
        if (use_sp9 && !phs && bm == 128 && env_int("HIFIC_SP9_KSPLIT", 2) == 2 && env_int("HIFIC_SP9_AG", 1))
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
            // 128-row software-pipelined tiles run one workgroup per CU: one full wave of 256 workgroups (measured, round 5:
            // 220 <- 2880 @16x16 100.6 -> 81.0 us, 220 <- 960 57.0 -> 41.1 us against the 800-slot target of the co-resident kernels)
            const int target = (use_sp9 && bm == 128) ? env_int("HIFIC_KSPLIT_TARGET_SP128", 256) : env_int("HIFIC_KSPLIT_TARGET", 800);
            int ks = (use_sp9 && bm == 128) ? (int)(target / g0) : (int)cdivl(target, g0);       // (one wave: round down)
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
        int rcp;
        if constexpr (std::is_same<T, float>::value) rcp = gc_pack_weights_f32(p, wp_elems, w, w_scale, sm, sc, sr, ss, ws, st, &plan_only);
        else rcp = gc_pack_weights_bf16(p, wp_elems, w, w_scale, sm, sc, sr, ss, ws, st, &plan_only);
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
                                : ((env_int("HIFIC_SP9_KSPLIT", 2) == 2 && bm == 128) ? 2 : 1),
                          phs ? (phs == 1 ? ",phs1" : ",phs2") : (p.rfx ? ",rfx" : ""));
    // the profiler keeps the residual-block trunk (>= 512 x 512 channels, one workgroup per CU) apart from the 220 / 320-channel
    // launches of the same instantiation (K-split grids, a third of the rows): roofline.frac of the trunk is a property of the
    // kernel, the average over both classes was a property of the layer mix
    if (use_sp9) { if (!(p.K >= 512 && p.C >= 512) && strlen(kname) + 8 < sizeof(kname)) strcat(kname, " narrow"); }
    else if (mp) snprintf(kname, sizeof(kname), "gconv_mp_kernel%s", p.split ? "<split>" : "");
    else snprintf(kname, sizeof(kname), "gconv_kernel<%s,%d,%s>", std::is_same<T, float>::value ? "f32" : "bf16", BC,
                  bm == 128 ? "2,2,2,2" : (bm == 64 ? "2,2,1,2" : "1,4,1,1"));
    const int pslot = prof_open(kname, aflops, st, ptag);
    gc_prof_bytes(pslot, gc_algo_bytes(p));
#define GC_LAUNCH(WGM, WGN, WM, WN)                                                                      \
    do {                                                                                                 \
        void (*kfn)(const GcParams) = gconv_kernel<T, BC, WGM, WGN, WM, WN, 1, 1>;                       \
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
        // software-pipelined kernel family (gconv_sp9.hip): one phase of exactly 9 taps, input stride 1, patch <= 192 pixels
        if (use_sp9) sp_done = gc_launch_sp9(p, grid, st, bm, phs, sp9_w4);
    }
    bool mp_done = false;
    if constexpr (std::is_same<T, bf16_t>::value && BC == 64) {
        if (mp) {
            gc_launch_mp(p, grid, lds, st);
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
        {   // weight-resident persistent kernel for the few-channel layers on big planes (gconv_wr.hip)
            const GcParams saved = p;
            const int rcw = launch_gconv_wr(p, w, w_scale, sm, sc, sr, ss, ws, st);
            if (rcw != HIFIC_ERR_UNSUPPORTED) return rcw;
            p = saved;
        }
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
                                      unsigned N, int K, int OH, int OW, int nd, int PW /* row pitch of P */, int act) {
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
    // (row pitch of P rounded up to 16 bytes: the weight-resident kernel then writes it with 16-byte stores; the extra columns
    //  are computed like any other and never read)
    const int PWd = (p.OWf + nx - 1 + 3) & ~3;
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
                                      p.K, p.OHf, p.OWf, nx, PWd, p.act);
    else hipLaunchKernelGGL(vrow_shift_add_kernel<bf16_t>, dim3(gx), dim3(256), 0, st, P, p.bias, (bf16_t*)p.out, (unsigned)p.N,
                            p.K, p.OHf, p.OWf, nx, PWd, p.act);
    return hific_launch_status();
}

static int launch_gconv_t_bf16(GcParams& p, const float* w, const float* w_scale, long long sm, long long sc,
                               long long sr, long long ss, WsAlloc& ws, hipStream_t st) {
    return launch_gconv_t<bf16_t>(p, w, w_scale, sm, sc, sr, ss, ws, st);
}


static int launch_gconv(GcParams& p, int dtype, const float* w, const float* w_scale, long long sm, long long sc,
                        long long sr, long long ss, WsAlloc& ws, hipStream_t st) {
    if (dtype == HIFIC_BF16 && !p.oscale) {
        int rcv = gc_launch_vc(p, w, w_scale, sm, sc, sr, ss, ws, st);
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
    if (g.wsplit) {          // `w` has C / 3 channels per row: the pack reads it three times (hi, hi, lo)
        p.wsplit_C = g.C / 3;
        return launch_gconv(p, dtype, w, w_scale, (long long)(g.C / 3) * RS, RS, g.S, 1, ws, st);
    }
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
