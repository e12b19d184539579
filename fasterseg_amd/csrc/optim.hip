// Multi-tensor SGD (momentum + weight decay) with a fused global-norm clip factor, one launch for every parameter.
//
// Replaces, for the train steps, torch.nn.utils.clip_grad_norm_'s scaling pass + torch.optim.SGD.step over the supernet's
// ~40 k parameter tensors (reference search/train_search.py:94-98,248-250; train/train.py:173-176): the gradients already
// live in ONE flat fp32 buffer (fasterseg_amd.parallel.FlatGradientSync), so the update is a single streaming pass
// p -= lr * (buf = momentum*buf + clip*g + wd*p) driven by a static device-side table of tensors and chunks.
// Conv filters keep their OIHW parameter storage while their gradient slice is stored [O][R][S][I] (coalesced wgrad
// atomics); the kernel walks the gradient order and maps each element to its OIHW parameter address.
#include "common.h"

namespace fs {

constexpr int SGD_CHUNK = 4096;

__global__ __launch_bounds__(256) void sgd_multi_kernel(const fs_sgd_tensor* __restrict__ tensors, const int* __restrict__ chunks,
                                                        const unsigned char* __restrict__ touched,
                                                        const float* __restrict__ grads, float* __restrict__ mom,
                                                        const float* __restrict__ grad_scale, float lr, float momentum,
                                                        float weight_decay) {
    const int t = chunks[2 * blockIdx.x];
    if (touched && !touched[t]) return;            // no gradient this step: torch leaves such a parameter untouched
    const fs_sgd_tensor d = tensors[t];
    const long long begin = (long long)chunks[2 * blockIdx.x + 1] * SGD_CHUNK;
    long long end = begin + SGD_CHUNK;
    if (end > d.numel) end = d.numel;
    const float clip = grad_scale ? *grad_scale : 1.f;
    const long long RSI = (long long)d.taps * d.I;
    for (long long e = begin + threadIdx.x; e < end; e += blockDim.x) {
        // e indexes the gradient slice ([O][taps][I]); the parameter is [O][I][taps]
        long long pe = e;
        if (d.taps > 1) {
            const long long o = e / RSI;
            const int rem = (int)(e - o * RSI);
            const int tap = rem / d.I, i = rem - tap * d.I;
            pe = o * RSI + (long long)i * d.taps + tap;
        }
        const float p = d.p[pe];
        const float g = grads[d.g_off + e] * clip + weight_decay * p;
        const float b = momentum * mom[d.g_off + e] + g;
        mom[d.g_off + e] = b;
        d.p[pe] = p - lr * b;
    }
}

}  // namespace fs

using namespace fs;

extern "C" int fs_sgd_chunk_elems(void) { return SGD_CHUNK; }

extern "C" fs_status fs_sgd_momentum_multi(void* stream, const fs_sgd_tensor* tensors, const int* chunks, int n_chunks,
                                           const unsigned char* touched, const float* grads, float* momentum_buf,
                                           const float* grad_scale, float lr, float momentum, float weight_decay) {
    FS_REQUIRE(tensors && chunks && grads && momentum_buf && n_chunks > 0, FS_ERR_INVALID, "fs_sgd_momentum_multi: bad argument");
    hipLaunchKernelGGL(sgd_multi_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, tensors, chunks, touched, grads,
                       momentum_buf, grad_scale, lr, momentum, weight_decay);
    return check_launch("fs_sgd_momentum_multi");
}
