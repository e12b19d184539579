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

template <typename PT>
__global__ __launch_bounds__(256) void sgd_multi_kernel(const fs_sgd_tensor* __restrict__ tensors, const int* __restrict__ chunks,
                                                        const unsigned char* __restrict__ touched,
                                                        const float* __restrict__ grads, float* __restrict__ mom,
                                                        const float* __restrict__ grad_scale, float lr, float momentum,
                                                        float weight_decay, int pack_only) {
    const int t = chunks[2 * blockIdx.x];
    const fs_sgd_tensor d = tensors[t];
    // no gradient this step: torch leaves such a parameter untouched (its resident packs are still current)
    if (!pack_only && touched && !touched[t]) return;
    if (pack_only && !d.pack_fwd && !d.pack_flip) return;
    const long long begin = (long long)chunks[2 * blockIdx.x + 1] * SGD_CHUNK;
    long long end = begin + SGD_CHUNK;
    if (end > d.numel) end = d.numel;
    const float clip = (grad_scale && !pack_only) ? *grad_scale : 1.f;
    const long long RSI = (long long)d.taps * d.I;
    const long long O = d.numel / RSI;
    PT* const fwd = (PT*)d.pack_fwd;
    PT* const flip = (PT*)d.pack_flip;
    for (long long e = begin + threadIdx.x; e < end; e += blockDim.x) {
        // e indexes the gradient slice ([O][taps][I]); the parameter is [O][I][taps]
        const long long o = e / RSI;
        const int rem = (int)(e - o * RSI);
        const int tap = rem / d.I, i = rem - tap * d.I;
        const long long pe = o * RSI + (long long)i * d.taps + tap;
        float p = d.p[pe];
        if (!pack_only) {
            const float g = grads[d.g_off + e] * clip + weight_decay * p;
            const float b = momentum * mom[d.g_off + e] + g;
            mom[d.g_off + e] = b;
            p -= lr * b;
            d.p[pe] = p;
        }
        // resident packed copies in the compute dtype: the gradient order IS the forward pack [O][R][S][I]; the data-gradient
        // pack is [I][R][S][O] with the taps rotated by 180 degrees
        if (fwd) Elem<PT>::store(fwd + e, p);
        if (flip) Elem<PT>::store(flip + ((long long)i * d.taps + (d.taps - 1 - tap)) * O + o, p);
    }
}

}  // namespace fs

using namespace fs;

extern "C" int fs_sgd_chunk_elems(void) { return SGD_CHUNK; }

extern "C" fs_status fs_sgd_momentum_multi(void* stream, const fs_sgd_tensor* tensors, const int* chunks, int n_chunks,
                                           const unsigned char* touched, const float* grads, float* momentum_buf,
                                           const float* grad_scale, float lr, float momentum, float weight_decay,
                                           int pack_dtype, int pack_only) {
    FS_REQUIRE(tensors && chunks && n_chunks > 0, FS_ERR_INVALID, "fs_sgd_momentum_multi: bad argument");
    FS_REQUIRE(pack_only || (grads && momentum_buf), FS_ERR_INVALID, "fs_sgd_momentum_multi: null gradient / momentum buffer");
    FS_REQUIRE(pack_dtype == FS_F32 || pack_dtype == FS_BF16, FS_ERR_INVALID, "fs_sgd_momentum_multi: bad pack dtype %d", pack_dtype);
    if (pack_dtype == FS_F32)
        hipLaunchKernelGGL((sgd_multi_kernel<float>), dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, tensors, chunks, touched,
                           grads, momentum_buf, grad_scale, lr, momentum, weight_decay, pack_only);
    else
        hipLaunchKernelGGL((sgd_multi_kernel<bf16_t>), dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, tensors, chunks, touched,
                           grads, momentum_buf, grad_scale, lr, momentum, weight_decay, pack_only);
    return check_launch("fs_sgd_momentum_multi");
}
