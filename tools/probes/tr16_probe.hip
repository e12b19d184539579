// Semantics probe of ds_read_b64_tr_b16 (__builtin_amdgcn_ds_read_tr16_b64_v4i16) as wgrad.hip uses it: within a 16-lane group, lane t
// supplies the address of row (t >> 2), column quad (t & 3) of a [4][16] block of 16-bit elements (any row stride) and receives
// COLUMN t of that block.   hipcc --offload-arch=gfx950 tools/probes/tr16_probe.hip -o /tmp/tr16 && /tmp/tr16
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) short s16x4;
constexpr int PITCH = 96;
__global__ void k3(const unsigned short* x, int* y) {
    __shared__ __attribute__((aligned(16))) unsigned short sm[128 * PITCH];
    for (int i = threadIdx.x; i < 128 * PITCH; i += 64) sm[i] = x[i];
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, t = l & 15;
    const unsigned short* p = sm + ((g >> 1) * 8 + (t >> 2)) * PITCH + (g & 1) * 16 + (t & 3) * 4;
    s16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    s16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p + 4 * PITCH));
    for (int j = 0; j < 4; ++j) { y[l * 8 + j] = (unsigned short)a0[j]; y[l * 8 + 4 + j] = (unsigned short)a1[j]; }
}
int main() {
    const int n = 128 * PITCH;
    unsigned short* hx = new unsigned short[n];
    for (int i = 0; i < n; ++i) hx[i] = (unsigned short)i;
    unsigned short* dx; int* dy; int hy[512];
    hipMalloc(&dx, n * 2); hipMalloc(&dy, sizeof(hy));
    hipMemcpy(dx, hx, n * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k3, dim3(1), dim3(64), 0, 0, dx, dy);
    hipMemcpy(hy, dy, sizeof(hy), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int g = l >> 4, t = l & 15;
        for (int e = 0; e < 8; ++e) {
            const int row = (g >> 1) * 8 + e, col = (g & 1) * 16 + t;      // pixel k, channel
            const int want = row * PITCH + col;
            if (hy[l * 8 + e] != want) { if (bad < 8) printf("lane %d elem %d: got %d want %d\n", l, e, hy[l * 8 + e], want); ++bad; }
        }
    }
    printf("tr16 probe: %d mismatches of 512\n", bad);
    printf("lane 0: %d %d %d %d | %d %d %d %d   lane 17: %d %d %d %d\n", hy[0], hy[1], hy[2], hy[3], hy[4], hy[5], hy[6], hy[7], hy[17 * 8], hy[17 * 8 + 1], hy[17 * 8 + 2], hy[17 * 8 + 3]);
    return bad != 0;
}
