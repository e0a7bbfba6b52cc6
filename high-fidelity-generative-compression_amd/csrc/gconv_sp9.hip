// gconv_sp9_kernel: the software-pipelined kernel family of the stride-1 3x3 layers (residual-block trunk forward and gather-form
// data gradient, the 220 / 320-channel 3x3 layers, the phase-merged kernel-3 stride-2 transposed layers on small planes) and its
// launcher.  Reference call sites: src/network/generator.py:9-44 (ResidualBlock), 98-137.
#include "gconv.h"
#include "gconv_dev.h"
#include "gconv_stage.h"

// build-time A/B switches (csrc/build.sh EXTRA=-D...; lib.py loads $HIFIC_LIB_PATH when set)
#ifndef SP9_ABL
#define SP9_ABL 0           // timing ablations of gconv_sp9_kernel's A-from-global loop (WRONG RESULTS; tools/micro_sp9.py only):
#endif                      // 1 = no patch loads / LDS writes, 2 = no A loads, 4 = no B fragment reads, 8 = no chunk barrier
#ifndef SP9_EXP
#define SP9_EXP 0           // bit 0: static wave priority, bit 1: pixel-shared XCD mapping (round-5 experiments, see docs/ENGINEERING_LOG.md)
#endif
#ifndef SP9_TOFF_ARG
#define SP9_TOFF_ARG 1      // gconv_sp9_kernel: tap offsets from the kernel arguments instead of the LDS table (-1..2 %)
#endif

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
#if SP9_EXP & 1
    // (experiment: static priority for the second-dispatched half of an 8-wave workgroup - MI355X_MICROARCH "two waves per SIMD" item 4)
    if (NWAVES == 8 && wave >= 4) __builtin_amdgcn_s_setprio(1);
#endif
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
#if SP9_EXP & 2
        // (experiment: XCD-contiguous q shares the PIXEL tile, the row tile varies fastest)
        { const int mtiles_ = nwg / p.max_tiles; mtile = q % mtiles_; tile = q / mtiles_; }
#else
        mtile = q / p.max_tiles;
        tile = q - mtile * p.max_tiles;
#endif
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
// Launcher: the instantiation the plan (gconv.hip launch_gconv_tb) asks for.  Returns false when the LDS image does not fit
// (the caller then falls back to the generic kernel).  Knobs that lost their A/B three rounds running are gone with their
// instantiations (A-from-global levels 2-4, the DPP-shifted fragments, the K-split of 64-row tiles).
// ---------------------------------------------------------------------------------------------------
bool gc_launch_sp9(const GcParams& p, dim3 grid, hipStream_t st, int bm, int phs, int sp9_w4) {
    constexpr int PITCH = 144;
    const int npatch = phs ? p.NI * p.ph[4].PH * p.ph[4].PW : p.NI * p.ph[0].PH * p.ph[0].PW;
    const size_t lds_sp = 64 + 3 * (size_t)bm * PITCH + 2 * (((size_t)(npatch + 2) * PITCH + 15) & ~(size_t)15);
    if (lds_sp > (size_t)kLdsBudget) return false;
    const bool ks2 = env_int("HIFIC_SP9_KSPLIT", 2) == 2;
#define SP9_LAUNCH(WM_, KSP_, RFX_, PHS_)                                                                           \
    do {                                                                                                            \
        gc_set_max_lds((const void*)gconv_sp9_kernel<WM_, KSP_, RFX_, PHS_>, (int)lds_sp);                          \
        hipLaunchKernelGGL((gconv_sp9_kernel<WM_, KSP_, RFX_, PHS_>), grid, dim3(256 * KSP_), lds_sp, st, p);       \
    } while (0)
    if (p.afrag) {
        // no weight ring; the K-split exchange (64 / 128 KB) is the larger LDS use for the usual 180-pixel patch
        size_t lds_ag = 64 + 2 * (((size_t)(npatch + 2) * PITCH + 15) & ~(size_t)15);
        // four reduction quarters on 64x128 wave slabs (gconv_sp9_kernel KSP = 4): half the A operand loads per MFMA
        if (lds_ag < (sp9_w4 ? 131072u : 65536u)) lds_ag = sp9_w4 ? 131072 : 65536;
#define SP9_AG_LAUNCH(KSP_, RFX_)                                                                                   \
    do {                                                                                                            \
        gc_set_max_lds((const void*)gconv_sp9_kernel<2, KSP_, RFX_, 0, false, 1>, (int)lds_ag);                     \
        hipLaunchKernelGGL((gconv_sp9_kernel<2, KSP_, RFX_, 0, false, 1>), grid, dim3(sp9_threads(KSP_)), lds_ag, st, p); \
    } while (0)
        if (sp9_w4) { if (p.rfx) SP9_AG_LAUNCH(4, true); else SP9_AG_LAUNCH(4, false); }
        else { if (p.rfx) SP9_AG_LAUNCH(2, true); else SP9_AG_LAUNCH(2, false); }
#undef SP9_AG_LAUNCH
    }
    else if (phs == 1) SP9_LAUNCH(1, 1, false, 1);
    else if (phs == 2) SP9_LAUNCH(1, 1, false, 2);
    else if (p.rfx) { if (bm == 128) SP9_LAUNCH(2, 2, true, 0); else SP9_LAUNCH(1, 1, true, 0); }
    else if (bm == 128) { if (ks2) SP9_LAUNCH(2, 2, false, 0); else SP9_LAUNCH(2, 1, false, 0); }
    else SP9_LAUNCH(1, 1, false, 0);
#undef SP9_LAUNCH
    return true;
}
