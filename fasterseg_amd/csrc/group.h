// Grouped launches: n <= FS_MAX_GROUP independent problems of ONE kernel in one launch (round 6).
//
// The supernet step (reference search/model_search.py:310-333) evaluates up to twelve MixedOps per layer that only depend on the
// previous layer; every MixedOp is ~17 launches forward and ~25 backward of 4-15 us on a fraction of the 256 CUs, and the step is the
// SUM of its kernel durations.  Round 4 grouped the convolutions; this header carries the same mechanism for
// everything else a MixedOp launches (BatchNorm passes, bilinear resamples, weighted sums): the problems' arguments travel by value in
// the kernel-argument segment, a workgroup finds its problem with a scalar search over the block prefix and runs the single-problem
// body on its local block id.  Each kernel is written once as a __device__ body over (args, block, blocks); its single and grouped
// __global__ wrappers are two lines each.
#pragma once
#include "common.h"

namespace fs {

constexpr int FS_MAX_GROUP = 12;         // problems per grouped launch (their arguments travel as kernel arguments: < 4 KB, asserted per struct;
                                         //   round 6: 8 -> 12, two supernet passes of five MixedOps each share a layer call)
constexpr int FS_KERNARG_MAX = 4096;     // bytes of kernel arguments a HIP launch can carry

template <typename A> struct GroupOf {
    int n;
    int blk_start[FS_MAX_GROUP + 1];     // first workgroup of every problem, [n] = grid size
    A p[FS_MAX_GROUP];
};
#define FS_ASSERT_KERNARG(S) static_assert(sizeof(S) <= fs::FS_KERNARG_MAX, #S " does not fit the kernel-argument segment")

// index of the problem workgroup `bid` belongs to (wave-uniform: scalar compares)
template <typename G> __device__ __forceinline__ int group_locate(const G& g, int bid) {
    static_assert(sizeof(G) <= FS_KERNARG_MAX, "grouped launch arguments do not fit the kernel-argument segment");
    int i = 0;
#pragma unroll
    for (int k = 1; k < FS_MAX_GROUP; ++k) i += (k < g.n && bid >= g.blk_start[k]) ? 1 : 0;
    return i;
}

// Host side: partitions calls 0..n-1 into buckets of equal key (first-occurrence order, at most FS_MAX_GROUP calls each) and hands
// every bucket's index list to run(idx, m).  n may exceed FS_MAX_GROUP.
template <typename KeyFn, typename RunFn> inline fs_status for_each_bucket(int n, KeyFn key, RunFn run) {
    constexpr int CAP = 4 * FS_MAX_GROUP;
    FS_REQUIRE(n >= 0 && n <= CAP, FS_ERR_INVALID, "grouped launch: %d problems (at most %d)", n, CAP);
    long long keys[CAP];
    bool done[CAP];
    for (int i = 0; i < n; ++i) { keys[i] = key(i); done[i] = false; }
    for (int i = 0; i < n; ++i) {
        if (done[i]) continue;
        int idx[FS_MAX_GROUP], m = 0;
        for (int j = i; j < n && m < FS_MAX_GROUP; ++j)
            if (!done[j] && keys[j] == keys[i]) { idx[m++] = j; done[j] = true; }
        const fs_status s = run(idx, m);
        if (s != FS_OK) return s;
    }
    return FS_OK;
}

// FS_GROUP_EW=0: the grouped forms of this header's clients are off (every problem its own launch, the round-5 behaviour)
inline bool group_ew_enabled() {
    static const bool on = [] { const char* e = getenv("FS_GROUP_EW"); return !(e && e[0] == '0'); }();
    return on;
}

// ---- call records of the grouped entry points (same fields as the extern "C" functions they batch) -----------------------------------
struct BnFwdCall {       // fs_bn_act_train_fwd (+ `stats_ready`: the producing convolution's epilogue already left (sum, sumsq) in stats)
    long long pixels; int C, groups; void* z; int z_cs; const float* gamma; const float* beta; float eps, momentum;
    float* running_mean; float* running_var; long long* num_batches_tracked; float* stats; float* saved; void* y; int y_cs; int dtype;
    int relu; void* ws; long long ws_bytes; int stats_ready;
};
struct BnBwdCall {       // fs_bn_act_train_bwd / the BatchNorm half of fs_conv_bn_act_train_bwd (`split_red`: red holds [groups + 1][2][C])
    long long pixels; int C, groups; const void* z; int z_cs; const void* dy; int dy_cs; const void* y; int y_cs; const float* saved;
    const float* gamma; float* red; int dtype; int relu; void* dz; int dz_cs; float* dgamma_acc; float* dbeta_acc; void* ws;
    long long ws_bytes;
};
struct ResizeCall {      // fs_bilinear_fwd (x -> y) / fs_bilinear_bwd (dy, y_out -> dx)
    const fs_resize_desc* d; const void* a; const void* b; void* out;
};
struct WsumCall {        // fs_weighted_sum / _bwd / _dots
    long long pixels; int C, n; const void* const* ptrs; const int* cs; const void* t; int t_cs; const float* coef; float* out; int dtype;
};
struct AxpyCall {        // fs_axpy_channels
    long long pixels; int C; const void* x; int x_cs; const float* alpha; void* y; int y_cs; int dtype; int accumulate;
};

// units.hip
fs_status bn_fwd_group(void* stream, const BnFwdCall* c, int n);
fs_status bn_bwd_group(void* stream, const BnBwdCall* c, int n);
// bn_col.hip: the one-launch BatchNorm of maps <= 512 pixels per group; idx selects the calls
fs_status bn_col_fwd_group(void* stream, const BnFwdCall* c, const int* idx, int n);
fs_status bn_col_bwd_group(void* stream, const BnBwdCall* c, const int* idx, int n);
// elementwise.hip: the grid-wide passes of larger maps
fs_status bn_stats_group(void* stream, const BnFwdCall* c, const int* idx, int n);
fs_status bn_apply_group(void* stream, const BnFwdCall* c, const int* idx, int n);
fs_status bn_bwd_reduce_group(void* stream, const BnBwdCall* c, const int* idx, int n);
fs_status bn_bwd_apply_group(void* stream, const BnBwdCall* c, const int* idx, int n);
// ... and the MIXED launches: problems of one launch take different bodies (kind[i]: 0 normalisation, 1 column kernel, 3 statistics /
// reduction pass), so the BatchNorm forward of all pending units is one launch and the backward two
fs_status bn_fwd_mixed_group(void* stream, const BnFwdCall* c, const int* idx, const int* kind, int n);
fs_status bn_bwd_mixed_group(void* stream, const BnBwdCall* c, const int* idx, const int* kind, int n);
fs_status wsum_group(void* stream, const WsumCall* c, int n);            // out = sum_k coef[k] * ptrs[k]          (t = out map)
fs_status wsum_bwd_group(void* stream, const WsumCall* c, int n);        // ptrs[k] = coef[k] * t                  (t = dy)
fs_status wsum_dots_group(void* stream, const WsumCall* c, int n);       // out[k] += <t, ptrs[k]>                 (t = dy)
fs_status axpy_group(void* stream, const AxpyCall* c, int n);
// resize.hip
fs_status bilinear_fwd_group(void* stream, const ResizeCall* c, int n);
fs_status bilinear_bwd_group(void* stream, const ResizeCall* c, int n);

}  // namespace fs
