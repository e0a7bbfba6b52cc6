// Natural-order weight-gradient kernels: a weight gradient REDUCES over pixels, so NCHW is already the MFMA's order.
// wgrad_s2_kernel (stride-2 3x3 / 4x4, conv and conv-transpose form: the four parity phases of the input turn every tap into a
// stride-1 shift of one phase plane) and wgrad_s1_kernel (stride-1 3x3 on 16-pixel planes: the three taps of a kernel row from
// one 16-byte LDS read), both filled by 16-byte loads only, and their launchers.
#include "gconv.h"
#include "gconv_dev.h"
#include "gconv_stage.h"
#include <type_traits>
#include <string.h>
#include <stdio.h>

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

// Stride-1 3x3 pad-1 bf16 layers on 16-pixel-multiple planes -> wgrad_s1_kernel; HIFIC_ERR_UNSUPPORTED when the layer does not qualify
int gc_launch_wgrad_s1(WgParams& p, float* dw, long long sm, long long sc, long long sr, long long ss, int accumulate,
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
    const int pslot = gc_prof_open("wgrad_s1_kernel", 2.0 * p.M * p.C * 9 * (double)p.N * p.AH * p.AW, st, ptag);
    gc_prof_bytes(pslot, (double)p.N * p.M * p.AH * p.AW * (p.a_f32 ? 4.0 : 2.0) + (double)p.N * p.C * p.BH * p.BW * (p.b_f32 ? 4.0 : 2.0) + (double)p.M * p.C * 9 * 4.0);
    size_t lds = 2 * (size_t)64 * 272 + 2 * (size_t)64 * (10 * 32 + 16) + 2 * (size_t)10 * 64 * 4;
    const size_t epi = (size_t)4 * 16 * (32 * 9 + 1) * sizeof(float);
    if (lds < epi) lds = epi;
    gc_set_max_lds((const void*)wgrad_s1_kernel, (int)lds);
    hipLaunchKernelGGL(wgrad_s1_kernel, dim3(grid), dim3(512), lds, st, p);
    gc_prof_close(pslot, st);
    int rc = hific_launch_status();
    if (rc != HIFIC_OK || p.direct) return rc;
    return gc_wgrad_finish(p, dw, sm, sc, sr, ss, accumulate, ws, st);
}

// Stride-2 3x3 / 4x4 bf16 layers -> wgrad_s2_kernel; HIFIC_ERR_UNSUPPORTED (nothing launched) when the layer does not qualify
int gc_launch_wgrad_s2(WgParams& p, float* dw, long long sm, long long sc, long long sr, long long ss, int accumulate,
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
    const int pslot = gc_prof_open("wgrad_s2_kernel", 2.0 * p.M * p.C * p.ntaps * (double)p.N * p.AH * p.AW, st, ptag);
    gc_prof_bytes(pslot, (double)p.N * p.M * p.AH * p.AW * (p.a_f32 ? 4.0 : 2.0) + (double)p.N * p.C * p.BH * p.BW * (p.b_f32 ? 4.0 : 2.0) + (double)p.M * p.C * p.ntaps * 4.0);
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
    gc_prof_close(pslot, st);
    int rc = hific_launch_status();
    if (rc != HIFIC_OK || p.direct) return rc;
    return gc_wgrad_finish(p, dw, sm, sc, sr, ss, accumulate, ws, st);
}

