// Generic tap-table implicit-GEMM convolution engine for gfx950 (MFMA, LDS-staged NCHW tiles).
//
// One kernel family covers every dense contraction on the HiFIC hot path:
//   conv2d fwd, conv2d bwd-data (as stride-phase sub-convs + reflect fold), conv-transpose fwd
//   (sub-pixel phases, no zero insertion), conv-transpose bwd-data, and the weight-gradient GEMM.
// Reference call sites this replaces (all nn.Conv2d / nn.ConvTranspose2d -> ATen/cuDNN there):
//   src/network/encoder.py:56-101, src/network/generator.py:28-42,98-142, src/network/hyper.py:52-54,83-85,
//   src/network/discriminator.py:35,53-64, src/loss/perceptual_similarity/pretrained_networks.py:59-75.
#pragma once
#include "common.h"

#define GC_MAXPH 16
#define GC_MAXTAPS 128
#define GC_NPIX 128      // pixels per tile (GEMM N per workgroup)
#define GC_TG 9          // taps accumulated per workgroup in the weight-gradient kernel

struct GcPhase {
    int ntaps, tap0;     // taps [tap0, tap0+ntaps) of the tap table
    int ooy, oox;        // output position = u*ost + ooy (masked to the full plane)
    int OHt, OWt;        // extent of the (u,v) tile domain
    int dy_min, dx_min;  // min tap offset (patch origin)
    int PH, PW;          // LDS patch extent per image (logical)
    int PWs;             // storage row width of the LDS patch (>= PW; multiple of 16 avoids B-fragment bank conflicts)
    int tiles_x, tiles_y;
    long long wp_off;    // element offset of this phase's packed weights
};

struct GcParams {
    const void* in;      // [N, C, IH, IW]
    const void* wp;      // packed weights, per phase [Kpad][ntaps][Cpad]
    const float* bias;   // [K] or null
    void* out;           // [N, K, OHf, OWf]
    const void* resid;   // same shape as out, or null
    int N, C, IH, IW, K, OHf, OWf, Cpad, Kpad;
    int ist, ost, bmode, act, in_f32, out_f32;
    int TH, TW, NI, tiles_n, max_tiles;
    int nphase;
    int tap_sw;          // kernel width S (weight tap index = r*S + s)
    int dbg;             // ablation flags for micro-benchmarks (HIFIC_DBG; 0 in production)
    double aflops;       // algorithmic FLOPs of the op on its real output domain (profiler only)
    // virtual channels (few-channel layers, see launch_gconv_fewc): reduction channel cc = c * csplit + j reads weight
    // column vcol_s[j]; output row mm = k * msplit + j reads weight column vrow_s[j] (0: off)
    int csplit, msplit;
    // split-in-pack (round 6): the weight tensor has wsplit_C real reduction channels and the packed image 3 * wsplit_C =
    // (hi, hi, lo) of them (hific_split3 which = 1), formed by the pack pass itself - no float32 (hi, hi, lo) weight image is
    // materialised (16 B written + 12 B re-read per weight and optimizer step).  wsplit_C % 64 == 0; 0 = off.
    int wsplit_C;
    short vcol_s[16];
    int epi_wide;        // wide-store epilogue through LDS (gc_epilogue_wide): legality checked by the plan
    int wstage;          // wide-load staging (stage_W): bf16 NCHW input, IW % 8 == 0, 16-byte aligned (set by the plan)
    int rfx;             // gather-form reflect data gradient (gconv_sp9_kernel RFX): `in` is the extended gradient
    // split-K of the forward-type kernels (small-grid layers: the hyperprior's 4x4..16x16 planes give 40-320 workgroups of
    // 50-400 serial steps): blockIdx.y takes `kchunks` channel chunks and writes raw float32 partial sums to
    // kpart[blockIdx.y][N,K,OHf,OWf]; ksplit_reduce_kernel adds them, the bias and the activation
    int ksplit, kchunks;
    float* kpart;
    long long kpart_stride;
    const float* oscale; // device scalar multiplied into the accumulator before bias / activation (spectral norm's 1/sigma with
                         // UNSCALED packed weights: the pack then depends only on the optimizer step and is cached)
    int split;           // native split-bf16 reduction (gconv_kernel SPLIT): both operands in the pair layout of
                         // hific_split3 which = 2 - every 32 reduction channels are (hi 16 | lo 16) of 16 real channels and
                         // a step issues hi*hi + hi*lo + lo*hi
    int afrag;           // packed weights in MFMA A-fragment order (gc_wp_index): gconv_sp9_kernel AG streams them
                         // global -> registers, bypassing LDS
                         // (2: the 8 KB per (row tile, 32-channel chunk, tap) blocks of gconv_pl_kernel)
    int pl_tpw;          // gconv_pl_kernel: pixel tiles per workgroup (persistent loop)
    int pl_wshare;       // gconv_pl_kernel: workgroups of one XCD share a row tile (big weights) instead of their pixel tiles
    // reflect-padded data gradient: `out` is the f32 padded plane buffer (only its rim is written); pixels inside
    // [fold_pt, fold_pt+fold_h) x [fold_pl, fold_pl+fold_w) go straight to out2 = dx[N,K,fold_h,fold_w]
    void* out2;
    int fold_pt, fold_pl, fold_h, fold_w, out2_f32;
    GcPhase ph[GC_MAXPH];
    short tap_dy[GC_MAXTAPS], tap_dx[GC_MAXTAPS];
    short tap_r[GC_MAXTAPS], tap_s[GC_MAXTAPS];
};

struct WgParams {
    const void* a;       // [N, M, AH, AW]   output-side operand (dY, or x for conv-transpose)
    const void* b;       // [N, C, BH, BW]   input-side operand, sampled at (u*ist+dy, v*ist+dx)
    float* ws;           // partials [nsplit][Mpad][ntaps][Cpad]
    float* dw;           // final gradient (direct epilogue when nsplit == 1)
    long long sm, sc, sr, ss;
    int accumulate, direct;
    int N, M, C, AH, AW, BH, BW, Mpad, Cpad;
    int ist, bmode, a_f32, b_f32;
    int TH, TW, NI, tiles_y, tiles_x, tiles_n, ntiles, tiles_per_split, nsplit;
    int ntaps, ngroups;
    int dbg;
    int xcd_remap;            // wgrad_s2_kernel: workgroups of one (channel block, pixel split) share an XCD
    int wstage_a, wstage_b;   // wide-load staging (stage_W) of the a / b operand: legality checked by the plan
    // small-channel (im2col) mode: virtual columns j = tap*4 + c; see wgrad_im2col_kernel
    int im2col, creal, ntaps_real, tsign, swap_out, a_bmode, a_y0, a_x0, a_h, a_w, b_y0, b_x0;
    int cqs;                  // im2col mode: log2 of the channel slots per tap (virtual column = (tap << cqs) + channel): 2 or 4
    GcPhase grp[GC_MAXPH];
    short tap_dy[GC_MAXTAPS], tap_dx[GC_MAXTAPS];
    short tap_r[GC_MAXTAPS], tap_s[GC_MAXTAPS];
};

// geometry of a conv layer as the reference constructs it (nn.Conv2d semantics)
struct ConvGeom {
    int N, C, H, W, K, R, S, stride, pt, pl, pb, pr, pad_mode;
    int red_split = 0;   // 1: C is 3x the layer's channels (split-bf16 operands, hific_split3): count 1/3 of the FLOPs
                         // 2: C is the pair layout (2 * C16) of the native split kernels; red_C = the layer's real channels
    int red_C = 0;
    int wsplit = 0;                  // 1: `w` is the layer's real [K, C / 3, R, S] weight, the pack forms (hi, hi, lo) (flags bit 5)
    const float* oscale = nullptr;   // see GcParams::oscale (conv2d fwd / bwd-data flags bit 4)
    int OH() const { return (H + pt + pb - R) / stride + 1; }
    int OW() const { return (W + pl + pr - S) / stride + 1; }
};
// nn.ConvTranspose2d semantics: x[N,Ci,H,W], w[Ci,Co,R,S]
struct ConvTGeom {
    int N, Ci, H, W, Co, R, S, stride, pad, outpad;
    int red_split = 0, red_C = 0;
    int OH() const { return (H - 1) * stride - 2 * pad + R + outpad; }
    int OW() const { return (W - 1) * stride - 2 * pad + S + outpad; }
};

// One weight-packing job of the batched pack kernel (hific_pack_batch): the conv plan (phases, taps, padded sizes,
// destination `p.wp`) plus the source tensor and the tiling of the pack grid.  Built on the host by
// hific_conv_pack_plan(), pointers set by hific_pack_job_set_ptrs(), uploaded to the device by the caller.
struct PackJob {
    GcParams p;
    const float* w;
    const float* scale;
    long long sm, sc, sr, ss;
    int RS, MB, mode;        // mode 0 / 1: pack_w2 tilings (c adjacent / m adjacent), 2: generic element-wise pack
    int gx, gy;              // pack grid of this job (blocks = gx * gy)
    int lds_bytes;
    int dtype;
    long long wp_bytes;      // size of the packed image
};

struct WsAlloc {   // bump allocator over the caller-provided workspace (+ the per-call weight-cache controls)
    char* base; size_t cap; size_t off;
    // persistent packed-weight cache of the caller (hific_hip.h: `wcache`): state 0 = none (pack into the workspace),
    // 1 = pack into wcache now, 2 = wcache already holds the packed image of the current weights (skip the pack)
    void* wcache = nullptr; size_t wcache_bytes = 0; int wcache_state = 0;
    PackJob* plan_out = nullptr;     // plan-only call: fill the job, launch nothing
    void* take(size_t bytes) {
        size_t a = (off + 255) & ~(size_t)255;
        if (a + bytes > cap) return nullptr;
        off = a + bytes;
        return base + a;
    }
};

int gc_conv_fwd(const ConvGeom& g, const void* x, const float* w, const float* w_scale, const float* bias,
                void* y, const void* resid, int act, int dtype, int in_f32, int out_f32,
                WsAlloc& ws, hipStream_t st);
int gc_conv_bwd_data(const ConvGeom& g, const void* dy, const float* w, const float* w_scale, void* dx,
                     int dtype, int in_f32, int out_f32, WsAlloc& ws, hipStream_t st);
int gc_conv_bwd_weight(const ConvGeom& g, const void* x, const void* dy, float* dw, int accumulate,
                       int dtype, int x_f32, int dy_f32, WsAlloc& ws, hipStream_t st);
int gc_convT_fwd(const ConvTGeom& g, const void* x, const float* w, const float* bias, void* y, int act,
                 int dtype, int in_f32, int out_f32, WsAlloc& ws, hipStream_t st);
int gc_convT_bwd_data(const ConvTGeom& g, const void* dy, const float* w, void* dx, int dtype, int in_f32,
                      int out_f32, WsAlloc& ws, hipStream_t st);
int gc_convT_bwd_weight(const ConvTGeom& g, const void* x, const void* dy, float* dw, int accumulate,
                        int dtype, int x_f32, int dy_f32, WsAlloc& ws, hipStream_t st);
size_t gc_ws_bytes_conv(const ConvGeom& g, int dtype);
size_t gc_ws_bytes_convT(const ConvTGeom& g, int dtype);
