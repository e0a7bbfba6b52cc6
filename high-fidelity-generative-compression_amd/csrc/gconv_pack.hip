// Weight packing of the conv engine: float32 [K, C, R, S] (any strides) -> the packed bf16 / f32 operand image of a plan
// (layout = gc_wp_index: [m][tap][c], or the MFMA-fragment orders of gconv_sp9_kernel AG / gconv_pl_kernel), per launch, into the
// caller's persistent cache, or for every stale layer of a step in ONE batched launch (hific_pack_batch).
#include "gconv.h"
#include "gconv_dev.h"
#include <type_traits>
#include <string.h>

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
        // split-in-pack (GcParams::wsplit_C): the 64 packed channels c0.. are channels c0 % Cr.. of the real weight, as hi (parts
        // 0, 1) or lo (part 2) of their bf16 split; Cr % 64 == 0, so a block never straddles two parts
        const int Cr = p.wsplit_C;
        const int wpart = Cr ? c0 / Cr : 0;
        const float* wrow = w + (long long)(Cr ? c0 - wpart * Cr : c0) * sc;
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
                            if (Cr) { const float h_ = bf2f(f2bf(x)); x = wpart < 2 ? h_ : x - h_; }
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
                    float x = ok ? v[u] * sc_ : 0.f;
                    if (Cr) { const float h_ = bf2f(f2bf(x)); x = wpart < 2 ? h_ : x - h_; }
                    pk_lds[c * pitch + ml * RS + (i - c * RS)] = x;
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
        if (p.wsplit_C && !(contiguous && sc == RS && !p.csplit && !p.msplit) ) return HIFIC_ERR_UNSUPPORTED;   // mode 0 only
        if (contiguous && sc == RS && !p.csplit && !p.msplit && (p.wsplit_C || !env_int("HIFIC_OLD_PACK", 0))) {
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
int gc_pack_weights_f32(GcParams& p, long long wp_elems, const float* w, const float* w_scale, long long sm, long long sc,
                        long long sr, long long ss, WsAlloc& ws, hipStream_t st, bool* plan_only) {
    return gc_pack_weights<float>(p, wp_elems, w, w_scale, sm, sc, sr, ss, ws, st, plan_only);
}


void gc_pack_launch_bf16(const PackJob& job, const float* w, const float* w_scale, hipStream_t st) {
    const GcParams& q = job.p;
    if (job.mode == 0) {
        if (job.lds_bytes > 48 * 1024) gc_set_max_lds((const void*)pack_w2_kernel<bf16_t, 0>, job.lds_bytes);
        hipLaunchKernelGGL((pack_w2_kernel<bf16_t, 0>), dim3(job.gx, job.gy), dim3(256), job.lds_bytes, st, q, w, w_scale, job.sm,
                           job.sc, job.RS, job.MB);
    } else if (job.mode == 1) {
        if (job.lds_bytes > 48 * 1024) gc_set_max_lds((const void*)pack_w2_kernel<bf16_t, 1>, job.lds_bytes);
        hipLaunchKernelGGL((pack_w2_kernel<bf16_t, 1>), dim3(job.gx, job.gy), dim3(256), job.lds_bytes, st, q, w, w_scale, job.sm,
                           job.sc, job.RS, job.MB);
    } else {
        hipLaunchKernelGGL(pack_w_kernel<bf16_t>, dim3(job.gx, q.nphase), dim3(256), 0, st, q, w, w_scale, job.sm, job.sc, job.sr,
                           job.ss);
    }
}
