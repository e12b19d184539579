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

constexpr int SGD_CHUNK = 4096;           // elements per block of a tensor walked linearly
constexpr int SGD_TO = 16;                // tiled tensors: output channels per block ...
constexpr int SGD_POS = 256;              // ... x parameter-order positions (input channel, tap) per block

// A tensor is walked in 2-D tiles when its gradient order differs from its parameter order (taps > 1) or it keeps resident
// packs (a 1x1 filter's rotated pack is its transpose); everything else (BN vectors, biases, 2-D weights) linearly.
__host__ __device__ inline bool sgd_tiled(int taps, const void* pack_fwd, const void* pack_flip) {
    return taps > 1 || pack_fwd != nullptr || pack_flip != nullptr;
}
__host__ __device__ inline int sgd_tile_i(int taps) { return taps >= SGD_POS ? 1 : SGD_POS / taps; }       // input channels per tile

// One block = one chunk of one tensor.
//   linear: SGD_CHUNK consecutive elements, gradient order == parameter order.
//   tiled:  SGD_TO output channels x (IT input channels x taps) of a filter.  The parameter rows are read and written as
//           contiguous runs ([O][I][taps] order), the gradient / momentum / forward pack in THEIR order ([O][taps][I]: runs of IT
//           floats), the rotated pack [I][taps][O] as runs of SGD_TO output channels - each side coalesced, exchanged through
//           LDS.  (The first version walked the gradient order and scattered single 4-byte parameter and 2-byte rotated-pack
//           accesses: 7.5 ms for the supernet's 252 M parameters, ~0.8 TB/s.)
template <typename PT>
__global__ __launch_bounds__(256) void sgd_multi_kernel(const fs_sgd_tensor* __restrict__ tensors, const int* __restrict__ chunks,
                                                        const unsigned char* __restrict__ touched,
                                                        const float* __restrict__ grads, float* __restrict__ mom,
                                                        const float* __restrict__ grad_scale, float lr, float momentum,
                                                        float weight_decay, int pack_only) {
    __shared__ float tile[SGD_TO][SGD_POS + 1];
    const int t = chunks[2 * blockIdx.x];
    const int c = chunks[2 * blockIdx.x + 1];
    const fs_sgd_tensor d = tensors[t];
    // no gradient this step: torch leaves such a parameter untouched (its resident packs are still current)
    if (!pack_only && touched && !touched[t]) return;
    if (pack_only && !d.pack_fwd && !d.pack_flip) return;
    const float clip = (grad_scale && !pack_only) ? *grad_scale : 1.f;
    PT* const fwd = (PT*)d.pack_fwd;
    PT* const flip = (PT*)d.pack_flip;
    const int tid = threadIdx.x;
    if (!sgd_tiled(d.taps, d.pack_fwd, d.pack_flip)) {
        const long long begin = (long long)c * SGD_CHUNK;
        long long end = begin + SGD_CHUNK;
        if (end > d.numel) end = d.numel;
        for (long long e = begin + tid; e < end; e += 256) {
            float p = d.p[e];
            const float g = grads[d.g_off + e] * clip + weight_decay * p;
            const float b = momentum * mom[d.g_off + e] + g;
            mom[d.g_off + e] = b;
            d.p[e] = p - lr * b;
        }
        return;
    }
    const int taps = d.taps, I = d.I;
    const long long RSI = (long long)taps * I;
    const int O = (int)(d.numel / RSI);
    const int IT = sgd_tile_i(taps);
    const int npt = (I + IT - 1) / IT;
    const int o0 = (c / npt) * SGD_TO, i0 = (c % npt) * IT;
    const int rows = O - o0 < SGD_TO ? O - o0 : SGD_TO;
    const int ni = I - i0 < IT ? I - i0 : IT;
    const int npos = ni * taps;                         // contiguous parameter positions i0*taps .. of every row
    // 1: parameter rows -> LDS (runs of npos floats)
    if (tid < npos)
        for (int r = 0; r < rows; ++r) tile[r][tid] = d.p[(long long)(o0 + r) * RSI + (long long)i0 * taps + tid];
    __syncthreads();
    // 2: the update in gradient order (input channel fastest)
    const int per_row = npos, total = rows * per_row;
    for (int idx = tid; idx < total; idx += 256) {
        const int r = idx / per_row, rem = idx - r * per_row;
        const int tap = rem / ni, ii = rem - tap * ni;
        const long long e = (long long)(o0 + r) * RSI + (long long)tap * I + i0 + ii;
        const int pos = ii * taps + tap;
        float p = tile[r][pos];
        if (!pack_only) {
            const float g = grads[d.g_off + e] * clip + weight_decay * p;
            const float b = momentum * mom[d.g_off + e] + g;
            mom[d.g_off + e] = b;
            p -= lr * b;
            tile[r][pos] = p;
        }
        if (fwd) Elem<PT>::store(fwd + e, p);           // the gradient order IS the forward pack [O][R][S][I]
    }
    __syncthreads();
    // 3: parameters back; the data-gradient pack [I][R][S][O], taps rotated by 180 degrees, SGD_TO output channels per run
    if (!pack_only && tid < npos)
        for (int r = 0; r < rows; ++r) d.p[(long long)(o0 + r) * RSI + (long long)i0 * taps + tid] = tile[r][tid];
    if (flip) {
        for (int idx = tid; idx < npos * SGD_TO; idx += 256) {
            const int r = idx % SGD_TO, pos = idx / SGD_TO;
            if (r < rows) {
                const int ii = pos / taps, tap = pos - ii * taps;
                Elem<PT>::store(flip + ((long long)(i0 + ii) * taps + (taps - 1 - tap)) * O + o0 + r, tile[r][pos]);
            }
        }
    }
}

}  // namespace fs

using namespace fs;

extern "C" int fs_sgd_chunk_elems(void) { return SGD_CHUNK; }

extern "C" long long fs_sgd_tensor_chunks(long long numel, int I, int taps, int has_packs) {
    if (numel <= 0 || I <= 0 || taps <= 0) return 0;
    if (!sgd_tiled(taps, has_packs ? (const void*)1 : nullptr, nullptr)) return (numel + SGD_CHUNK - 1) / SGD_CHUNK;
    const long long O = numel / ((long long)taps * I);
    const int IT = sgd_tile_i(taps);
    return ((O + SGD_TO - 1) / SGD_TO) * ((I + IT - 1) / IT);
}

extern "C" fs_status fs_sgd_momentum_multi(void* stream, const fs_sgd_tensor* tensors, const int* chunks, int n_chunks,
                                           const unsigned char* touched, const float* grads, float* momentum_buf,
                                           const float* grad_scale, float lr, float momentum, float weight_decay,
                                           int pack_dtype, int pack_only) {
    FS_REQUIRE(tensors && chunks && n_chunks > 0, FS_ERR_INVALID, "fs_sgd_momentum_multi: bad argument");
    FS_REQUIRE(pack_only || (grads && momentum_buf), FS_ERR_INVALID, "fs_sgd_momentum_multi: null gradient / momentum buffer");
    FS_REQUIRE(pack_dtype == FS_F32 || pack_dtype == FS_BF16, FS_ERR_INVALID, "fs_sgd_momentum_multi: bad pack dtype %d", pack_dtype);
    if (pack_dtype == FS_F32)
        FS_LAUNCH((sgd_multi_kernel<float>), dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, tensors, chunks, touched,
                           grads, momentum_buf, grad_scale, lr, momentum, weight_decay, pack_only);
    else
        FS_LAUNCH((sgd_multi_kernel<bf16_t>), dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, tensors, chunks, touched,
                           grads, momentum_buf, grad_scale, lr, momentum, weight_decay, pack_only);
    return check_launch("fs_sgd_momentum_multi");
}
