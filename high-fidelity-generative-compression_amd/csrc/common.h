// Shared device/host helpers for libhific_hip.so (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>

#define HIFIC_OK 0
#define HIFIC_ERR_ARG (-1)
#define HIFIC_ERR_WS (-2)
#define HIFIC_ERR_LAUNCH (-3)
#define HIFIC_ERR_UNSUPPORTED (-4)

enum { HIFIC_F32 = 0, HIFIC_BF16 = 1 };
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_LEAKY = 2 };
enum { PAD_ZERO = 0, PAD_REFLECT = 1 };

typedef unsigned short bf16_t;  // raw bfloat16 bits

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short short4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__device__ __forceinline__ float bf2f(bf16_t h) {
    return __uint_as_float(((unsigned)h) << 16);
}
// round-to-nearest-even, NaN quieted (matches torch's float->bfloat16): gfx950's v_cvt_pk_bf16_f32, one instruction per PAIR
// (the integer sequence - NaN test, +0x7fff + lsb, shift, pack - was ~14 VALU instructions per pair, a third of the VALU work of
// the forward epilogues)
typedef __bf16 hw_bf16x2_t __attribute__((ext_vector_type(2)));
typedef float hw_f32x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ bf16_t f2bf(float f) {
    const __bf16 r = (__bf16)f;
    return __builtin_bit_cast(bf16_t, r);
}
// (lo, hi) -> packed pair, lo in bits 0..15
__device__ __forceinline__ unsigned f2bf2(float lo, float hi) {
    const hw_f32x2_t v = {lo, hi};
    const hw_bf16x2_t r = __builtin_convertvector(v, hw_bf16x2_t);
    return __builtin_bit_cast(unsigned, r);
}

template <typename T> struct DT;
template <> struct DT<float> {
    static constexpr int code = HIFIC_F32;
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct DT<bf16_t> {
    static constexpr int code = HIFIC_BF16;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf2f(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f2bf(v); }
};

// single-reflection index (ReflectionPad2d semantics, no edge repeat); requires pad < n
__host__ __device__ __forceinline__ int reflect_idx(int i, int n) {
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return i;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---- tickets: the last-arriving workgroup of a launch finishes a two-stage reduction inside the same launch ----------------
// A two-stage reduction (partials per workgroup -> one tiny reduce launch) costs a second launch of 4-7 us for a few KB of
// work: ~120 such launches per training cycle.  With a ticket - an unsigned counter in device memory, zero between launches -
// every workgroup publishes its partials, takes a ticket, and the workgroup that draws the last one performs the second
// stage (same summation order as the separate kernel where noted: bit-identical results).
// Memory model on gfx950 (8 XCDs, one L2 each, not coherent with each other for ordinary accesses): an agent-scope
// release / acquire FENCE (`__threadfence()`) is `buffer_wbl2` / `buffer_inv` - a write-back / invalidate of the XCD's WHOLE L2;
// issued once per workgroup next to kernels that stream hundreds of MB it cost the training cycle +9 ms (measured, round 6).
// So no fences: the PARTIALS travel by agent-scope atomic stores and loads (hific_st_agent / hific_ld_agent: write-through /
// read-through at the device coherence point, nothing else is flushed), each wave waits for its stores' acknowledgements
// (s_waitcnt vmcnt(0)) before the barrier, and only then does thread 0 take the ticket (agent-scope atomic).  Nothing else crosses workgroups inside the launch; the results the last workgroup writes become visible at the end
// of the kernel like any other output.
// Counters live in a caller-owned, zero-initialised buffer registered per stream (hific_set_ticket_buffer): kernels of one
// stream run one after another, so one buffer per stream suffices; without a registered buffer every entry point keeps its
// two-launch form.  A kernel resets the counters it used (the last workgroup stores 0) - the buffer stays zero between launches.
__device__ __forceinline__ float hific_ld_agent(const float* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void hific_st_agent(float* p, float v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bool hific_last_block(unsigned* ticket, unsigned nblocks) {
    __shared__ unsigned hific_tk_last;
    // this wave's partial stores are acknowledged by the coherence point before anyone takes the ticket: a workgroup-scope
    // release fence emits no vmcnt wait on gfx950 (checked in the ISA), so the wait is explicit: vmcnt(0), other counters free
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned last = (t + 1u == nblocks) ? 1u : 0u;
        if (last) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        hific_tk_last = last;
    }
    __syncthreads();
    return hific_tk_last != 0u;
}
// host side (capi.hip): the ticket counters registered for `st`, or nullptr (then: two launches); `need` counters
unsigned* hific_tickets(hipStream_t st, int need);

static inline int hific_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? HIFIC_OK : HIFIC_ERR_LAUNCH;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline long long cdivl(long long a, long long b) { return (a + b - 1) / b; }
