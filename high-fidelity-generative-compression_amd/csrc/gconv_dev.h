// Device / host helpers shared by the translation units of the conv engine (gconv.hip, gconv_pl.hip).
#pragma once
#include "gconv.h"

// host-side services of gconv.hip used by the other translation units of the engine
int gc_env_int(const char* name, int dflt);                 // environment knob, read once per process and cached
static inline int env_int(const char* name, int dflt) { return gc_env_int(name, dflt); }
static const int kLdsBudget = 150 * 1024;
// (TH, TW, NI) of a (u, v) tile domain under an LDS budget: exhaustive search over tile shapes (gconv.hip)
bool gc_choose_tile(int N, int OHt, int OWt, int ist, int span_y, int span_x, int pitch, int fixed_bytes, int pref_budget,
                    int& TH, int& TW, int& NI, int ntaps, bool need16);
// gconv_sp9_kernel family (gconv_sp9.hip): launches the instantiation the plan asks for; false when its LDS image does not fit
bool gc_launch_sp9(const GcParams& p, dim3 grid, hipStream_t st, int bm, int phs, int sp9_w4);
// launches the pack kernel of an already planned bf16 job (job.p.wp set)
void gc_pack_launch_bf16(const PackJob& job, const float* w, const float* w_scale, hipStream_t st);
int gc_pack_weights_f32(GcParams& p, long long wp_elems, const float* w, const float* w_scale, long long sm, long long sc,
                        long long sr, long long ss, WsAlloc& ws, hipStream_t st, bool* plan_only);
void gc_set_max_lds(const void* fn, int bytes);             // dynamic-LDS opt-in per (device, kernel function), raised monotonically
int gc_prof_open(const char* kname, double flops, hipStream_t st, const char* tag);
void gc_prof_close(int slot, hipStream_t st);
void gc_prof_bytes(int slot, double bytes);                 // algorithmic HBM bytes of the launch in `slot`
double gc_algo_bytes(const GcParams& p);                    // input + output + weights of a forward-type launch, each once
// Packs the weights for plan `p` (layout = gc_wp_index) into the caller's cache / the workspace as WsAlloc says, or - plan-only
// call (ws.plan_out) - fills the pack job and sets *plan_only.  Sets p.wp.
int gc_pack_weights_bf16(GcParams& p, long long wp_elems, const float* w, const float* w_scale, long long sm, long long sc,
                         long long sr, long long ss, WsAlloc& ws, hipStream_t st, bool* plan_only);
// Pipelined stride-2 forward-type kernel (gconv_pl.hip); HIFIC_ERR_UNSUPPORTED when the plan does not qualify
int launch_gconv_pl(GcParams& p, const float* w, const float* w_scale, long long sm, long long sc, long long sr, long long ss,
                    WsAlloc& ws, hipStream_t st);
// Weight-resident persistent kernel for the few-channel layers on big planes (gconv_wr.hip); same contract
int launch_gconv_wr(GcParams& p, const float* w, const float* w_scale, long long sm, long long sc, long long sr, long long ss,
                    WsAlloc& ws, hipStream_t st);

// Source index of packed element (row m, reduction channel c, tap (r, s)).  With virtual channels (csplit) the packed
// channel cc = c * csplit + j stands for channel c at kernel column vcol_s[j]; with virtual rows (msplit) the packed row
// mm = k * msplit + j stands for output channel k at kernel column vcol_s[j] (the tap then only carries the kernel row).
__host__ __device__ __forceinline__ long long gc_weight_index(const GcParams& p, int m, int c, int r, int s, long long sm,
                                                              long long sc, long long sr, long long ss) {
    if (p.csplit) { s = p.vcol_s[c % p.csplit]; c = c / p.csplit; }
    if (p.msplit) { s = p.vcol_s[m % p.msplit]; m = m / p.msplit; }
    return m * sm + c * sc + r * sr + s * ss;
}
// Destination element index of packed weight (row m, tap t of phase ph, reduction channel c).  Default: [m][t][c].
// afrag (gconv_sp9_kernel AG): MFMA A-fragment order - for every (32-row block, tap, 64-channel chunk, 16-deep slice) the
// 64 lanes' 16-byte operands are contiguous (lane = (c / 8 % 2) * 32 + m % 32 holds channels c..c+7 of row m), so one
// wave-wide 16-byte-per-lane load brings a whole v_mfma_f32_32x32x16_bf16 A operand as 1 KB of consecutive bytes.
__host__ __device__ __forceinline__ long long gc_wp_index(const GcParams& p, const GcPhase& ph, int m, int t, int c) {
    if (!p.afrag) return ((long long)m * ph.ntaps + t) * p.Cpad + c;
    if (p.afrag == 2) {
        // gconv_pl_kernel: per (128-row tile, 32-channel chunk, tap) one 8 KB block = [4 row blocks][2 slices][1 KB operand], so
        // the TG taps of a step group are TG x 8 KB of consecutive bytes that 512 threads copy with one 16-byte load each per tap
        const int nch = p.Cpad >> 5;
        const long long blk = ((((long long)(m >> 7) * nch + (c >> 5)) * ph.ntaps + t) * 4 + ((m >> 5) & 3)) * 2 + ((c >> 4) & 1);
        const int lane = ((c >> 3) & 1) * 32 + (m & 31);
        return (blk * 64 + lane) * 8 + (c & 7);
    }
    const int nch = p.Cpad >> 6;
    const long long blk = (((long long)(m >> 5) * ph.ntaps + t) * nch + (c >> 6)) * 4 + ((c >> 4) & 3);   // 1 KB operand
    const int lane = ((c >> 3) & 1) * 32 + (m & 31);
    return (blk * 64 + lane) * 8 + (c & 7);
}

// One (32-row block mi, pixel fragment ni) of the tile; the accumulator vector arrives BY VALUE and every index is a
// compile-time constant (references to the accumulator array / runtime fragment indices made hipcc keep the whole
// accumulator array in scratch on some instantiations).
template <bool TF32>
__device__ __forceinline__ void gc_store_block(const GcParams& p, const GcPhase& ph, const f32x16_t a, int mi, int mbase,
                                               int lhi, int pu_, int pv_, int pn_, bool pvalid_, bool hb, const float* bp,
                                               float slope) {
    const bool out_f32 = TF32 || p.out_f32;
    float bv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = mbase + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        bv[r] = bp[(hb && m < p.K) ? m : 0];
    }
    const float osc = p.oscale ? *p.oscale : 1.f;
    const int oy = pu_ * p.ost + ph.ooy, ox = pv_ * p.ost + ph.oox;
    const bool okp = pvalid_ && pn_ < p.N && pu_ < ph.OHt && pv_ < ph.OWt &&
                     (unsigned)oy < (unsigned)p.OHf && (unsigned)ox < (unsigned)p.OWf;
    if (!okp) return;
    size_t plane = (size_t)p.OHf * p.OWf;
    size_t pbase = (size_t)pn_ * p.K * plane + (size_t)oy * p.OWf + ox;
    // split-K partial sums (GcParams::ksplit): the same store path with the destination redirected to this split's float32
    // plane set; the caller passes no bias and slope 1.  (A separate store loop here spilled the accumulators of the
    // 128-row sp9 instantiations to scratch: 320 B/lane, 12x slower.)
    const bool part = p.ksplit > 1;
    void* optr = part ? (void*)(p.kpart + (size_t)blockIdx.y * (size_t)p.kpart_stride) : p.out;
    bool of32 = out_f32 || part;
    if (p.fold_h) {      // reflect-pad data gradient: interior pixels straight to dx, only the rim to the plane buffer
        const int iy = oy - p.fold_pt, ix = ox - p.fold_pl;
        if ((unsigned)iy < (unsigned)p.fold_h && (unsigned)ix < (unsigned)p.fold_w) {
            plane = (size_t)p.fold_h * p.fold_w;
            pbase = (size_t)pn_ * p.K * plane + (size_t)iy * p.fold_w + ix;
            optr = p.out2; of32 = TF32 || p.out2_f32;
        }
    }
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = mbase + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        v[r] = a[r] * osc + ((hb && m < p.K) ? bv[r] : 0.f);
    }
    if (p.resid) {          // rare path (no caller on the HiFIC graph fuses a residual): loads batched per fragment
        float rv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = mbase + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            const size_t idx = pbase + (size_t)(m < p.K ? m : 0) * plane;
            rv[r] = out_f32 ? ((const float*)p.resid)[idx] : bf2f(((const bf16_t*)p.resid)[idx]);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] += rv[r];
    }
    // one 64-bit row base per fragment, 32-bit row strides (kr * plane < 2^32), the row bound tested once per wave when the
    // whole 32-row block is inside
    const size_t rowbase = pbase + (size_t)(mbase + mi * 32 + 4 * lhi) * plane;
    const unsigned plane32 = (unsigned)plane;
    const bool full = mbase + mi * 32 + 32 <= p.K;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int kr = (r & 3) + 8 * (r >> 2);
        const float y = v[r] > 0.f ? v[r] : v[r] * slope;
        if (full || mbase + mi * 32 + kr + 4 * lhi < p.K) {
            const size_t idx = rowbase + (size_t)((unsigned)kr * plane32);
            if (of32) ((float*)optr)[idx] = y; else ((bf16_t*)optr)[idx] = f2bf(y);
        }
    }
}

// weight-gradient family (gconv_wgrad.hip, gconv_wgrad_nat.hip)
int gc_wgrad_finish(WgParams& p, float* dw, long long sm, long long sc, long long sr, long long ss, int accumulate, WsAlloc& ws,
                    hipStream_t st);
int gc_launch_wgrad_s1(WgParams& p, float* dw, long long sm, long long sc, long long sr, long long ss, int accumulate, WsAlloc& ws,
                       hipStream_t st);
int gc_launch_wgrad_s2(WgParams& p, float* dw, long long sm, long long sc, long long sr, long long ss, int accumulate, WsAlloc& ws,
                       hipStream_t st);

// gconv_wgrad_c3.hip: stride-1 weight gradient with <= 4 input channels on big planes, operands in natural order
int gc_launch_wgrad_c3(const ConvGeom& g, const void* x, const void* dy, float* dw, int accumulate, int x_f32, int dy_f32,
                       WsAlloc& ws, hipStream_t st);

// gconv_mpvc.hip
void gc_launch_mp(const GcParams& p, dim3 grid, size_t lds, hipStream_t st);
int gc_launch_vc(GcParams& p, const float* w, const float* w_scale, long long sm, long long sc, long long sr, long long ss,
                 WsAlloc& ws, hipStream_t st);
