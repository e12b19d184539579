// OHEM cross-entropy on the full-resolution logits.
//
// Replaces, on the GPU, the softmax / gather / log_softmax / nll_loss chain of ProbOhemCrossEntropy2d (reference
// tools/seg_opr/loss_opr.py:63-93) in the student distillation step (train/train.py:256-259): at 12 x 19 x 512 x 1024
// logits are 478 MB per head, and the ATen chain moves that tensor ~10 times (softmax out, transposed copy, log_softmax out,
// nll, and their backwards).  Here the forward reads the logits once and emits two per-pixel vectors (probability of the
// true class, log-sum-exp); the hard-example threshold is still found with a device sort of that vector; the backward reads
// the logits once more and writes d logits = kept * (softmax - onehot) * scale directly.
#include "common.h"

namespace fs {

__global__ __launch_bounds__(256) void ohem_ce_fwd_kernel(const float* __restrict__ logits, const long long* __restrict__ target,
                                                          int C, long long HW, long long P, int ignore,
                                                          float* __restrict__ true_prob, float* __restrict__ nll,
                                                          float* __restrict__ lse_out) {
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < P; p += (long long)gridDim.x * blockDim.x) {
        const long long b = p / HW, hw = p - b * HW;
        const float* base = logits + b * C * HW + hw;          // class planes: consecutive threads read consecutive floats
        float m = -INFINITY;
        for (int c = 0; c < C; ++c) m = fmaxf(m, base[(long long)c * HW]);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += expf(base[(long long)c * HW] - m);
        const float lse = m + logf(s);
        const long long t = target[p];
        const bool valid = t != (long long)ignore && t >= 0 && t < C;
        const float xt = valid ? base[t * HW] : 0.f;
        true_prob[p] = valid ? expf(xt - lse) : 1.f;            // ignored pixels never count as hard examples
        nll[p] = valid ? lse - xt : 0.f;
        lse_out[p] = lse;
    }
}

__global__ __launch_bounds__(256) void ohem_ce_bwd_kernel(const float* __restrict__ logits, const long long* __restrict__ target,
                                                          const float* __restrict__ lse, const unsigned char* __restrict__ kept,
                                                          const float* __restrict__ scale, int C, long long HW, long long P,
                                                          float* __restrict__ dlogits) {
    const float g = *scale;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < P; p += (long long)gridDim.x * blockDim.x) {
        const long long b = p / HW, hw = p - b * HW;
        const long long off = b * C * HW + hw;
        const bool k = kept[p] != 0;
        const float l = lse[p];
        const long long t = target[p];
        for (int c = 0; c < C; ++c) {
            float d = 0.f;
            if (k) d = (expf(logits[off + (long long)c * HW] - l) - (c == t ? 1.f : 0.f)) * g;
            __builtin_nontemporal_store(d, dlogits + off + (long long)c * HW);
        }
    }
}

// KL distillation term: nn.KLDivLoss()(log_softmax(student), softmax(teacher)) with reduction 'mean' (train/train.py:64,260):
// sum over all B*C*H*W elements of p_t * (log p_t - log p_s), divided by the element count.  Forward: one read of both logit
// tensors -> per-pixel KL and the two log-sum-exps; backward: d student = (p_s - p_t) * scale.
__global__ __launch_bounds__(256) void kl_fwd_kernel(const float* __restrict__ s_logits, const float* __restrict__ t_logits, int C,
                                                     long long HW, long long P, float* __restrict__ kl, float* __restrict__ lse_s,
                                                     float* __restrict__ lse_t) {
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < P; p += (long long)gridDim.x * blockDim.x) {
        const long long b = p / HW, hw = p - b * HW;
        const long long off = b * C * HW + hw;
        float ms = -INFINITY, mt = -INFINITY;
        for (int c = 0; c < C; ++c) {
            ms = fmaxf(ms, s_logits[off + (long long)c * HW]);
            mt = fmaxf(mt, t_logits[off + (long long)c * HW]);
        }
        float ss = 0.f, st = 0.f;
        for (int c = 0; c < C; ++c) {
            ss += expf(s_logits[off + (long long)c * HW] - ms);
            st += expf(t_logits[off + (long long)c * HW] - mt);
        }
        const float ls = ms + logf(ss), lt = mt + logf(st);
        float acc = 0.f;
        for (int c = 0; c < C; ++c) {
            const float lps = s_logits[off + (long long)c * HW] - ls, lpt = t_logits[off + (long long)c * HW] - lt;
            const float pt = expf(lpt);
            acc += pt > 0.f ? pt * (lpt - lps) : 0.f;          // xlogy convention of F.kl_div: 0 * log 0 = 0
        }
        kl[p] = acc;
        lse_s[p] = ls;
        lse_t[p] = lt;
    }
}

__global__ __launch_bounds__(256) void kl_bwd_kernel(const float* __restrict__ s_logits, const float* __restrict__ t_logits,
                                                     const float* __restrict__ lse_s, const float* __restrict__ lse_t,
                                                     const float* __restrict__ scale, int C, long long HW, long long P,
                                                     float* __restrict__ d_s) {
    const float g = *scale;
    for (long long p = blockIdx.x * (long long)blockDim.x + threadIdx.x; p < P; p += (long long)gridDim.x * blockDim.x) {
        const long long b = p / HW, hw = p - b * HW;
        const long long off = b * C * HW + hw;
        const float ls = lse_s[p], lt = lse_t[p];
        for (int c = 0; c < C; ++c) {
            const long long i = off + (long long)c * HW;
            __builtin_nontemporal_store((expf(s_logits[i] - ls) - expf(t_logits[i] - lt)) * g, d_s + i);
        }
    }
}

}  // namespace fs

using namespace fs;

extern "C" fs_status fs_kl_distill_fwd(void* stream, const float* student, const float* teacher, long long B, int C, long long HW,
                                       float* kl, float* lse_s, float* lse_t) {
    FS_REQUIRE(student && teacher && kl && lse_s && lse_t && B > 0 && C > 0 && HW > 0, FS_ERR_INVALID, "fs_kl_distill_fwd: bad argument");
    const long long P = B * HW;
    long long blocks = (P + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    FS_LAUNCH(kl_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, student, teacher, C, HW, P, kl, lse_s,
                       lse_t);
    return check_launch("fs_kl_distill_fwd");
}

extern "C" fs_status fs_kl_distill_bwd(void* stream, const float* student, const float* teacher, const float* lse_s,
                                       const float* lse_t, const float* scale, long long B, int C, long long HW, float* d_student) {
    FS_REQUIRE(student && teacher && lse_s && lse_t && scale && d_student && B > 0 && C > 0 && HW > 0, FS_ERR_INVALID,
               "fs_kl_distill_bwd: bad argument");
    const long long P = B * HW;
    long long blocks = (P + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    FS_LAUNCH(kl_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, student, teacher, lse_s, lse_t, scale,
                       C, HW, P, d_student);
    return check_launch("fs_kl_distill_bwd");
}

extern "C" fs_status fs_ohem_ce_fwd(void* stream, const float* logits, const long long* target, long long B, int C, long long HW,
                                    int ignore, float* true_prob, float* nll, float* lse) {
    FS_REQUIRE(logits && target && true_prob && nll && lse && B > 0 && C > 0 && HW > 0, FS_ERR_INVALID, "fs_ohem_ce_fwd: bad argument");
    const long long P = B * HW;
    long long blocks = (P + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    FS_LAUNCH(ohem_ce_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, logits, target, C, HW, P, ignore,
                       true_prob, nll, lse);
    return check_launch("fs_ohem_ce_fwd");
}

extern "C" fs_status fs_ohem_ce_bwd(void* stream, const float* logits, const long long* target, const float* lse,
                                    const unsigned char* kept, const float* scale, long long B, int C, long long HW,
                                    float* dlogits) {
    FS_REQUIRE(logits && target && lse && kept && scale && dlogits && B > 0 && C > 0 && HW > 0, FS_ERR_INVALID,
               "fs_ohem_ce_bwd: bad argument");
    const long long P = B * HW;
    long long blocks = (P + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    FS_LAUNCH(ohem_ce_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, logits, target, lse, kept, scale,
                       C, HW, P, dlogits);
    return check_launch("fs_ohem_ce_bwd");
}
