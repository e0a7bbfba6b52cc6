// C-ABI entry points for the convolution family (see include/hific_hip.h for the contract).
#include "gconv.h"
#include <string.h>

#include <mutex>
#include <stdlib.h>

// ---- ticket buffers (common.h "tickets"): stream -> caller-owned zeroed counters ---------------------------------------------
namespace {
struct TicketSlot { hipStream_t st; int dev; unsigned* buf; int n; };     // (device, stream): the null stream is handle 0 on every device
TicketSlot g_tickets[64];
int g_ntickets = 0;
std::mutex g_ticket_mu;
int g_tickets_on = -1;
}
unsigned* hific_tickets(hipStream_t st, int need) {
    std::lock_guard<std::mutex> lk(g_ticket_mu);
    if (g_tickets_on < 0) { const char* e = getenv("HIFIC_TICKETS"); g_tickets_on = (e && e[0] == '0') ? 0 : 1; }
    if (!g_tickets_on) return nullptr;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    for (int i = 0; i < g_ntickets; ++i)
        if (g_tickets[i].st == st && g_tickets[i].dev == dev) return g_tickets[i].n >= need ? g_tickets[i].buf : nullptr;
    return nullptr;
}

extern "C" {

int hific_version(void) { return 100; }

// Registers (buf != NULL) or removes (buf == NULL) the ticket counters of `stream`: `bytes` / 4 unsigned counters in device
// memory, ZERO when registered, owned by the caller and valid until removed.  See include/hific_hip.h.
int hific_set_ticket_buffer(hipStream_t stream, void* buf, size_t bytes) {
    std::lock_guard<std::mutex> lk(g_ticket_mu);
    g_tickets_on = -1;                                   // re-read HIFIC_TICKETS (tests flip it)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return HIFIC_ERR_ARG;     // the buffer belongs to the CURRENT device
    int at = -1;
    for (int i = 0; i < g_ntickets; ++i) if (g_tickets[i].st == stream && g_tickets[i].dev == dev) at = i;
    if (!buf) {
        if (at >= 0) g_tickets[at] = g_tickets[--g_ntickets];
        return HIFIC_OK;
    }
    if (((size_t)buf & 3) != 0 || bytes < 4) return HIFIC_ERR_ARG;
    if (at < 0) {
        if (g_ntickets >= 64) return HIFIC_ERR_UNSUPPORTED;
        at = g_ntickets++;
    }
    g_tickets[at] = TicketSlot{stream, dev, (unsigned*)buf, (int)(bytes / 4)};
    return HIFIC_OK;
}

// Returns 0 and fills (name[<=63], CU count, LDS bytes/CU) for the given device; never throws.
int hific_device_info(int device, char* name, int* cus, int* lds_per_cu) {
    hipDeviceProp_t pr;
    if (hipGetDeviceProperties(&pr, device) != hipSuccess) return HIFIC_ERR_ARG;
    if (name) { strncpy(name, pr.gcnArchName, 63); name[63] = 0; }
    if (cus) *cus = pr.multiProcessorCount;
    if (lds_per_cu) *lds_per_cu = (int)pr.maxSharedMemoryPerMultiProcessor;
    return HIFIC_OK;
}

size_t hific_conv2d_ws_bytes(int N, int C, int H, int W, int K, int R, int S, int stride, int pt, int pl, int pb,
                             int pr, int dtype) {
    ConvGeom g{N, C, H, W, K, R, S, stride, pt, pl, pb, pr, PAD_ZERO};
    return gc_ws_bytes_conv(g, dtype);
}
size_t hific_conv_transpose2d_ws_bytes(int N, int Ci, int H, int W, int Co, int R, int S, int stride, int pad,
                                       int outpad, int dtype) {
    ConvTGeom g{N, Ci, H, W, Co, R, S, stride, pad, outpad};
    return gc_ws_bytes_convT(g, dtype);
}

static bool geom_ok(const ConvGeom& g) {
    return g.N > 0 && g.C > 0 && g.K > 0 && g.H > 0 && g.W > 0 && g.R > 0 && g.S > 0 && g.stride > 0 &&
           g.pt >= 0 && g.pl >= 0 && g.pb >= 0 && g.pr >= 0 && g.OH() > 0 && g.OW() > 0 &&
           (g.pad_mode == PAD_ZERO || (g.pad_mode == PAD_REFLECT && g.pt < g.H && g.pb < g.H && g.pl < g.W && g.pr < g.W));
}
static bool geomT_ok(const ConvTGeom& g) {
    return g.N > 0 && g.Ci > 0 && g.Co > 0 && g.H > 0 && g.W > 0 && g.R > 0 && g.S > 0 && g.stride > 0 &&
           g.pad >= 0 && g.outpad >= 0 && g.OH() > 0 && g.OW() > 0;
}

// y = act(conv2d(pad(x), w*w_scale) + bias [+ resid]);  x [N,C,H,W], w f32 [K,C,R,S], y [N,K,OH,OW]
// flags: bit0 = x is f32 although dtype is bf16, bit1 = y (and resid) are f32 although dtype is bf16, bit2 = C is the
// 3x split-bf16 reduction of a C/3-channel layer (hific_split3; only the profiler's FLOP count changes), bit3 = x and w
// are in the PAIR layout (hific_split3 which = 2, C = 2 * C16): the native split kernel forms hi*hi + hi*lo + lo*hi; bit4 =
// w_scale (a device scalar: spectral norm's 1/sigma) multiplies the ACCUMULATOR in the epilogue instead of the weights in the
// pack pass, so the packed image depends only on the weights and can live in the caller's cache across forwards; bits
// 8.. = the layer's real channel count (profiler FLOPs)
int hific_conv2d_fwd(const void* x, const float* w, const float* w_scale, const float* bias, const void* resid, void* y,
                     int N, int C, int H, int W, int K, int R, int S, int stride, int pt, int pl, int pb, int pr,
                     int pad_mode, int act, int dtype, int flags, void* ws, size_t ws_bytes, void* wcache,
                     size_t wcache_bytes, int wcache_state, hipStream_t stream) {
    ConvGeom g{N, C, H, W, K, R, S, stride, pt, pl, pb, pr, pad_mode};
    if (!geom_ok(g) || !x || !w || !y) return HIFIC_ERR_ARG;
    WsAlloc a{(char*)ws, ws_bytes, 0};
    a.wcache = wcache; a.wcache_bytes = wcache_bytes; a.wcache_state = wcache_state;
    g.red_split = (flags & 8) ? 2 : ((flags >> 2) & 1);
    g.red_C = flags >> 8;
    if ((flags & 8) && (dtype != HIFIC_BF16 || (flags & 1) || C % 32 != 0 || w_scale)) return HIFIC_ERR_ARG;
    if (flags & 16) {            // w_scale multiplies the accumulator in the epilogue; the weights are packed unscaled
        if (!w_scale || resid) return HIFIC_ERR_ARG;
        g.oscale = w_scale; w_scale = nullptr;
    }
    if (flags & 32) {            // split-in-pack: w is [K, C / 3, R, S]; the pack forms the (hi, hi, lo) image (bit 2 layout)
        if (!(flags & 4) || (flags & 8) || dtype != HIFIC_BF16 || C % 3 != 0 || (C / 3) % 64 != 0 || w_scale) return HIFIC_ERR_ARG;
        g.wsplit = 1;
    }
    return gc_conv_fwd(g, x, w, w_scale, bias, y, resid, act, dtype, flags & 1, (flags >> 1) & 1, a, stream);
}

// dx = adjoint of the (padded, strided) convolution applied to dy.  flags: bit0 dy is f32, bit1 dx is f32
int hific_conv2d_bwd_data(const void* dy, const float* w, const float* w_scale, void* dx, int N, int C, int H, int W,
                          int K, int R, int S, int stride, int pt, int pl, int pb, int pr, int pad_mode, int dtype,
                          int flags, void* ws, size_t ws_bytes, void* wcache, size_t wcache_bytes, int wcache_state,
                          hipStream_t stream) {
    ConvGeom g{N, C, H, W, K, R, S, stride, pt, pl, pb, pr, pad_mode};
    if (!geom_ok(g) || !dy || !w || !dx) return HIFIC_ERR_ARG;
    WsAlloc a{(char*)ws, ws_bytes, 0};
    a.wcache = wcache; a.wcache_bytes = wcache_bytes; a.wcache_state = wcache_state;
    if (flags & 16) {            // as in hific_conv2d_fwd
        if (!w_scale) return HIFIC_ERR_ARG;
        g.oscale = w_scale; w_scale = nullptr;
    }
    return gc_conv_bwd_data(g, dy, w, w_scale, dx, dtype, flags & 1, (flags >> 1) & 1, a, stream);
}

// dw f32 [K,C,R,S] (=|+=) sum_n,oy,ox dy * pad(x).  flags: bit0 x is f32, bit1 dy is f32
int hific_conv2d_bwd_weight(const void* x, const void* dy, float* dw, int N, int C, int H, int W, int K, int R, int S,
                            int stride, int pt, int pl, int pb, int pr, int pad_mode, int accumulate, int dtype,
                            int flags, void* ws, size_t ws_bytes, hipStream_t stream) {
    ConvGeom g{N, C, H, W, K, R, S, stride, pt, pl, pb, pr, pad_mode};
    if (!geom_ok(g) || !x || !dy || !dw) return HIFIC_ERR_ARG;
    WsAlloc a{(char*)ws, ws_bytes, 0};
    return gc_conv_bwd_weight(g, x, dy, dw, accumulate, dtype, flags & 1, (flags >> 1) & 1, a, stream);
}

// nn.ConvTranspose2d: x [N,Ci,H,W], w f32 [Ci,Co,R,S], y [N,Co,OH,OW]
int hific_conv_transpose2d_fwd(const void* x, const float* w, const float* bias, void* y, int N, int Ci, int H, int W,
                               int Co, int R, int S, int stride, int pad, int outpad, int act, int dtype, int flags,
                               void* ws, size_t ws_bytes, void* wcache, size_t wcache_bytes, int wcache_state,
                               hipStream_t stream) {
    ConvTGeom g{N, Ci, H, W, Co, R, S, stride, pad, outpad};
    if (!geomT_ok(g) || !x || !w || !y) return HIFIC_ERR_ARG;
    WsAlloc a{(char*)ws, ws_bytes, 0};
    a.wcache = wcache; a.wcache_bytes = wcache_bytes; a.wcache_state = wcache_state;
    g.red_split = (flags & 8) ? 2 : ((flags >> 2) & 1);
    g.red_C = flags >> 8;
    if ((flags & 8) && (dtype != HIFIC_BF16 || (flags & 1) || Ci % 32 != 0)) return HIFIC_ERR_ARG;
    return gc_convT_fwd(g, x, w, bias, y, act, dtype, flags & 1, (flags >> 1) & 1, a, stream);
}
int hific_conv_transpose2d_bwd_data(const void* dy, const float* w, void* dx, int N, int Ci, int H, int W, int Co,
                                    int R, int S, int stride, int pad, int outpad, int dtype, int flags, void* ws,
                                    size_t ws_bytes, void* wcache, size_t wcache_bytes, int wcache_state,
                                    hipStream_t stream) {
    ConvTGeom g{N, Ci, H, W, Co, R, S, stride, pad, outpad};
    if (!geomT_ok(g) || !dy || !w || !dx) return HIFIC_ERR_ARG;
    WsAlloc a{(char*)ws, ws_bytes, 0};
    a.wcache = wcache; a.wcache_bytes = wcache_bytes; a.wcache_state = wcache_state;
    return gc_convT_bwd_data(g, dy, w, dx, dtype, flags & 1, (flags >> 1) & 1, a, stream);
}
int hific_conv_transpose2d_bwd_weight(const void* x, const void* dy, float* dw, int N, int Ci, int H, int W, int Co,
                                      int R, int S, int stride, int pad, int outpad, int accumulate, int dtype,
                                      int flags, void* ws, size_t ws_bytes, hipStream_t stream) {
    ConvTGeom g{N, Ci, H, W, Co, R, S, stride, pad, outpad};
    if (!geomT_ok(g) || !x || !dy || !dw) return HIFIC_ERR_ARG;
    WsAlloc a{(char*)ws, ws_bytes, 0};
    return gc_convT_bwd_weight(g, x, dy, dw, accumulate, dtype, flags & 1, (flags >> 1) & 1, a, stream);
}

// ---- persistent packed-weight cache: plan / batched re-pack (see include/hific_hip.h) ---------------------------
size_t hific_pack_job_bytes(void) { return sizeof(PackJob); }

// kind 0: the packing hific_conv2d_fwd uses, 1: hific_conv2d_bwd_data.  Fills *job (host memory, hific_pack_job_bytes()).
int hific_conv2d_pack_plan(int kind, int N, int C, int H, int W, int K, int R, int S, int stride, int pt, int pl, int pb,
                           int pr, int pad_mode, int dtype, int flags, void* job, size_t job_bytes) {
    ConvGeom g{N, C, H, W, K, R, S, stride, pt, pl, pb, pr, pad_mode};
    if (!geom_ok(g) || !job || job_bytes < sizeof(PackJob) || (kind != 0 && kind != 1)) return HIFIC_ERR_ARG;
    WsAlloc a{(char*)0x100000, (size_t)1 << 60, 0};            // plan-only: nothing is dereferenced or launched
    a.plan_out = (PackJob*)job;
    const float* fake_w = (const float*)job;       // never dereferenced in a plan-only call
    g.red_split = (flags & 8) ? 2 : 0;                          // (the plan - and so the pack layout - depends on it)
    if (flags & 16) g.oscale = fake_w;                          // ... and on the epilogue-scale form (kernel choice)
    if (flags & 32) {                                           // ... and on split-in-pack (source strides of the job)
        if (kind != 0 || (flags & 8) || C % 3 != 0 || (C / 3) % 64 != 0) return HIFIC_ERR_ARG;
        g.wsplit = 1;
    }
    if (kind == 0) return gc_conv_fwd(g, job, fake_w, nullptr, nullptr, job, nullptr, ACT_NONE, dtype, flags & 1, (flags >> 1) & 1, a, nullptr);
    return gc_conv_bwd_data(g, job, fake_w, nullptr, job, dtype, flags & 1, (flags >> 1) & 1, a, nullptr);
}
int hific_conv_transpose2d_pack_plan(int kind, int N, int Ci, int H, int W, int Co, int R, int S, int stride, int pad,
                                     int outpad, int dtype, int flags, void* job, size_t job_bytes) {
    ConvTGeom g{N, Ci, H, W, Co, R, S, stride, pad, outpad};
    if (!geomT_ok(g) || !job || job_bytes < sizeof(PackJob) || (kind != 0 && kind != 1)) return HIFIC_ERR_ARG;
    WsAlloc a{(char*)0x100000, (size_t)1 << 60, 0};            // plan-only: nothing is dereferenced or launched
    a.plan_out = (PackJob*)job;
    const float* fake_w = (const float*)job;
    g.red_split = (flags & 8) ? 2 : 0;
    if (kind == 0) return gc_convT_fwd(g, job, fake_w, nullptr, job, ACT_NONE, dtype, flags & 1, (flags >> 1) & 1, a, nullptr);
    return gc_convT_bwd_data(g, job, fake_w, job, dtype, flags & 1, (flags >> 1) & 1, a, nullptr);
}
int hific_pack_job_set_ptrs(void* job, void* wpack, const float* w, const float* w_scale) {
    if (!job) return HIFIC_ERR_ARG;
    PackJob* j = (PackJob*)job;
    j->p.wp = wpack; j->w = w; j->scale = w_scale;
    return HIFIC_OK;
}
int hific_pack_job_info(const void* job, int* nblocks, int* lds_bytes, long long* wpack_bytes, int* dtype) {
    if (!job) return HIFIC_ERR_ARG;
    const PackJob* j = (const PackJob*)job;
    if (nblocks) *nblocks = j->gx * j->gy;
    if (lds_bytes) *lds_bytes = j->lds_bytes;
    if (wpack_bytes) *wpack_bytes = j->wp_bytes;
    if (dtype) *dtype = j->dtype;
    return HIFIC_OK;
}

}  // extern "C"
