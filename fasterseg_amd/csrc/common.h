// Shared device/host helpers for libfasterseg_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/fasterseg_hip.h"

namespace fs {

typedef uint16_t bf16_t;   // raw bfloat16 storage
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

// ---- error reporting (thread-local, never aborts) ---------------------------------------------
void set_error(const char* fmt, ...);
#define FS_REQUIRE(cond, code, ...)            \
    do {                                       \
        if (!(cond)) {                         \
            fs::set_error(__VA_ARGS__);        \
            return code;                       \
        }                                      \
    } while (0)

inline fs_status check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        set_error("%s: launch failed: %s", what, hipGetErrorString(e));
        return FS_ERR_LAUNCH;
    }
    return FS_OK;
}

// Bit-reproducible mode (api.cpp; FS_DETERMINISTIC=1 or fs_set_deterministic): cross-block reductions (wgrad pixel slabs, BatchNorm
// statistics / parameter gradients of maps above 512 pixels) go through ordered partial sums in the caller's workspace instead of
// float atomics.  Off by default: on MI355X the ordered form costs 2x on those kernels (per-XCD L2: the partials travel through HBM).
extern int g_deterministic;

// launch census (census.hip).  g_census_on: 0 off, 1 count conv launches by geometry, 2 count AND time every kernel launch:
// FS_LAUNCH then goes through hipExtLaunchKernelGGL with a start/stop event pair, i.e. the dispatch's own begin/end timestamps
// (what rocprofv3's kernel trace reports), on the stream the kernel is launched on.
extern int g_census_on;
struct CensusScope {          // every FS_LAUNCH issued while the scope lives is attributed to this (family, geometry) entry
    bool live;
    CensusScope(int family, const fs_conv_desc* d);
    ~CensusScope();
};
struct CensusGroupScope {     // a grouped launch: n entries counted, the launch's measured time shared out by `share`
    bool live;
    CensusGroupScope(int family, const fs_conv_desc* const* d, const double* share, int n);
    ~CensusGroupScope();
};
bool census_events(const char* kernel, hipStream_t stream, hipEvent_t* start, hipEvent_t* stop);
// ALGORITHMIC HBM bytes of the NEXT timed launch of this thread (the HBM-bound families: BatchNorm passes, resamples, weighted sums;
// summed over the problems of a grouped launch): fs_census_read_kernels reports them per kernel beside launches and device time, so the
// train-step roofline can price every family, not only the convolutions
void census_note_bytes(double bytes);
#define FS_NOTE_BYTES(expr)                                              \
    do {                                                                 \
        if (fs::g_census_on > 1) fs::census_note_bytes((double)(expr)); \
    } while (0)
#define FS_CENSUS(family, d) fs::CensusScope fs_census_scope_((family), (d))
#define FS_LAUNCH(kernel, grid, block, shmem, stream, ...)                                                      \
    do {                                                                                                         \
        hipEvent_t fs_e0_, fs_e1_;                                                                               \
        if (fs::g_census_on > 1 && fs::census_events(#kernel, (stream), &fs_e0_, &fs_e1_)) {                    \
            hipExtLaunchKernelGGL(kernel, grid, block, shmem, stream, fs_e0_, fs_e1_, 0, __VA_ARGS__);           \
        } else {                                                                                                 \
            hipLaunchKernelGGL(kernel, grid, block, shmem, stream, __VA_ARGS__);                                 \
        }                                                                                                        \
    } while (0)

// fs_conv2d_fwd_ws without its split-K reduction launch: when the kernel splits K across blocks the partial slabs stay in
// `workspace` ([*slices][M][Cout] fp32, y untouched) and the caller sums them (fs_bn_group_fwd does); *slices == 1: y is written.
fs_status conv_fwd_deferred(void* stream, const fs_conv_desc* d, const void* x, const void* w_packed, void* y, void* workspace,
                            long long workspace_bytes, int* slices);

// wgrad.hip: n fs_conv2d_wgrad_ws calls as one launch (program.hip's lockstep executor)
fs_status wgrad_launch_group(void* stream, int n, const fs_conv_desc* const* d, const void* const* x, const void* const* dy, float* const* dw,
                             const long long* o_stride, const long long* i_stride, const long long* t_stride, void* workspace,
                             long long workspace_bytes);

// units.hip: the arguments of fs_conv_bn_act_train_fwd / _bwd as records, and n of them executed with grouped convolution launches
struct UnitFwdCall {
    const fs_conv_desc* d; const void* x; const void* w; const float* gamma; const float* beta; float* running_mean; float* running_var;
    long long* num_batches_tracked; float eps, momentum; float* stats; float* saved; void* z; void* y; void* ws; long long ws_bytes;
};
struct UnitBwdCall {
    const fs_conv_desc* d; const void* x; const void* w_flip; const void* z; const void* y; const void* dy; int dy_cs; const float* saved;
    const float* gamma; float* red; float* dgamma_acc; float* dbeta_acc; void* dz; float* dw; long long o_stride, i_stride, t_stride;
    void* dx; int dx_cs, wf_os, wf_ts; void* ws; long long ws_bytes;
};
fs_status unit_fwd_group(void* stream, const UnitFwdCall* u, int n);
// Deferred weight gradients (round 6): nothing in a layer call reads a weight gradient, and the operands of one - the unit's saved input
// and its dz - live until the call returns (the launch programs' arenas never reuse a slot).  The layer executor collects the weight
// gradients of ALL its scheduler rounds here and issues them at the end, FS_MAX_GROUP problems per launch: a round's five to ten
// gradients no longer pay a launch of their own (~20 us of launch / block start-up floor each, profiles/r06_wgrad_ablation_in_step.txt).
struct WgradDeferred {
    fs_conv_desc d; const void* x; const void* dy; float* dw; long long o_stride, i_stride, t_stride;
};
struct WgradSink {
    static constexpr int CAP = 192;
    WgradDeferred q[CAP];
    int n;
    void* ws; long long ws_bytes;
};
fs_status wgrad_sink_flush(void* stream, WgradSink* sink);          // wgrad.hip: issues and empties the sink
fs_status unit_bwd_group(void* stream, const UnitBwdCall* u, int n, WgradSink* sink = nullptr);

// `relu` argument of the BatchNorm kernels: bit 0 = apply ReLU, bits 8.. = the first channel it applies to (0: every channel);
// forward kernels: bit 1 = num_batches_tracked points at TWO adjacent counters (the two BatchNorm modules of a fused pair).
// A fused pair of conv->BN units whose first member has its ReLU after a later up-sample ('conv_downup', operations.py:271-276)
// and whose second has it right after the BN ('conv_2x_downup'.bn1, :438-440) is one BN launch over both channel ranges.
__host__ __device__ inline bool relu_at(int relu, int c) { return (relu & 1) && c >= (relu >> 8); }
__device__ inline void bump_batches_tracked(long long* counters, int relu, int by) {
    if (!counters) return;
    counters[0] += by;
    if (relu & 2) counters[1] += by;
}

// ---- cross-block reductions without float atomics ("last block finishes") -----------------------------------------------------
// MI355X has one L2 per XCD: a whole-cache fence pair per block (__threadfence) writes back and invalidates the L2 for every block of
// a launch - measured 3x the kernel time on the supernet's wgrad launches.  Instead the partial results are stored and re-loaded with
// agent-scope (sc1) accesses, which are coherent per location across XCDs, every thread waits for its stores (vmcnt 0), and after the
// block barrier ONE thread bumps the arrival counter with release semantics (one L2 write-back, no invalidate).
__device__ __forceinline__ void store_coherent(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float load_coherent(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// returns true in every thread of the LAST of `expected` blocks to arrive at *counter (call from all threads of the block, after the
// block's store_coherent calls); `flag` is a __shared__ int
__device__ __forceinline__ bool arrive_last(unsigned int* counter, unsigned int expected, int* flag) {
    __builtin_amdgcn_s_waitcnt(0);                 // this thread's stores have left the CU
    __syncthreads();
    if (threadIdx.x == 0)
        // acquire as well as release: the last block's reads of the other blocks' partials are ordered behind this arrival by the
        // memory model, not only by the sc1 accesses of today's hardware
        *flag = (__hip_atomic_fetch_add(counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == expected - 1u) ? 1 : 0;
    __syncthreads();
    return *flag != 0;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
inline int elem_size(int dtype) { return dtype == FS_BF16 ? 2 : 4; }
inline int vec_elems(int dtype) { return 16 / elem_size(dtype); }

// ---- bf16 <-> f32 ------------------------------------------------------------------------------
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even through the hardware converter (v_cvt_pk_bf16_f32 on gfx950): one instruction per PAIR instead of
// the ~6-instruction integer sequence - the epilogues and resamplers are issue-bound on these conversions
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
__device__ __forceinline__ uint32_t pack2_bf16(float lo, float hi) {
    bf16x2_t v;
    v[0] = (__bf16)lo;
    v[1] = (__bf16)hi;
    return __builtin_bit_cast(uint32_t, v);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int VEC = 4;
    __device__ static __forceinline__ float load(const float* p) { return *p; }
    __device__ static __forceinline__ void store(float* p, float v) { *p = v; }
    // unpack a 16-byte vector into VEC floats
    __device__ static __forceinline__ void unpack(const u32x4& v, float* out) {
#pragma unroll
        for (int i = 0; i < 4; ++i) out[i] = __uint_as_float(v[i]);
    }
    __device__ static __forceinline__ u32x4 pack(const float* in) {
        u32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = __float_as_uint(in[i]);
        return v;
    }
};
template <> struct Elem<bf16_t> {
    static constexpr int VEC = 8;
    __device__ static __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(*p); }
    __device__ static __forceinline__ void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
    __device__ static __forceinline__ void unpack(const u32x4& v, float* out) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            out[2 * i] = __uint_as_float(v[i] << 16);
            out[2 * i + 1] = __uint_as_float(v[i] & 0xffff0000u);
        }
    }
    __device__ static __forceinline__ u32x4 pack(const float* in) {
        u32x4 v;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = pack2_bf16(in[2 * i], in[2 * i + 1]);
        return v;
    }
};

// An int divisor that carries its magic numbers (round 6).  The element-wise kernels of a train step - BatchNorm passes, weighted sums,
// resamples - decode every 16-byte vector's (pixel, channel vector) from a linear index; with a `long long` index that was a 64-bit
// SOFTWARE division (~150 VALU instructions) per 16 bytes moved, three to six of them in the resamples: the "HBM-bound" families ran at
// 0.5-1.5 TB/s, bound by their index arithmetic.  fast_div(n, d) = n / d by one v_mul_hi + shift for 0 <= n < 2^31 (Granlund-Montgomery
// round-up form: magic = floor(2^(31 + l) / d) + 1 with l = ceil(log2 d) fits 32 bits and is exact below 2^31), the plain division above.
// Constructed from an int on the host (the args structs are aggregate-initialised with ints), reads as an int everywhere else.
struct DivInt {
    int v;
    uint32_t magic;
    int shift;
    DivInt() = default;
    __host__ __device__ DivInt(int d) : v(d), magic(0), shift(0) {
        if (d > 1) {
            int l = 0;
            while ((1u << l) < (unsigned)d) ++l;
            shift = l - 1;
            magic = (uint32_t)(((1ull << (31 + l)) / (unsigned)d) + 1ull);
        }
    }
    __host__ __device__ operator int() const { return v; }
};
__device__ __forceinline__ long long fast_div(long long n, const DivInt& d) {
    if ((unsigned long long)n < 0x80000000ull) return (long long)(d.v == 1 ? (uint32_t)n : (__umulhi((uint32_t)n, d.magic) >> d.shift));
    return n / d.v;
}

// r = t % d; t /= d with a 32-bit division while t fits (kernels whose divisors arrive as plain ints: ~35 instead of ~150 instructions)
__device__ __forceinline__ int divmod32(long long& t, int d) {
    if ((unsigned long long)t < 0x100000000ull) {
        const uint32_t q = (uint32_t)t / (uint32_t)d;
        const int r = (int)((uint32_t)t - q * (uint32_t)d);
        t = (long long)q;
        return r;
    }
    const long long q = t / d;
    const int r = (int)(t - q * d);
    t = q;
    return r;
}

__device__ __forceinline__ u32x4 ldg16(const void* p) { return *reinterpret_cast<const u32x4*>(p); }
__device__ __forceinline__ void stg16(void* p, const u32x4& v) { *reinterpret_cast<u32x4*>(p) = v; }

// wave64 all-lane sum
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

}  // namespace fs

// ---- bilinear taps, align_corners=True (ATen index arithmetic: see resize.hip) ------------------------------------
namespace fs {
struct Tap {
    int i0, i1;
    float l0, l1;
};
__device__ __forceinline__ Tap make_tap(float scale, int dst, int in_size) {
    Tap t;
    const float src = scale * (float)dst;
    t.i0 = (int)src;
    t.i1 = t.i0 + ((t.i0 < in_size - 1) ? 1 : 0);
    t.l1 = src - (float)t.i0;
    t.l0 = 1.f - t.l1;
    return t;
}
}  // namespace fs
