// Operand-fill microbenchmark behind DESIGN.md §7.1: how fast can ONE CU pull bytes L2 -> LDS, as a function of how many waves issue the
// loads and how many each keeps in flight?  conv_igemm2.hip measured 30-35 B/clk/CU with 4 waves x ~12 KB in flight; this probe says
// whether that is the path's limit or latency x bytes-in-flight.
//   MODE 0: buffer_load_dwordx4 ... lds (LDS-DMA, no VGPRs)      MODE 1: global_load_dwordx4 -> VGPR -> ds_write_b128
// Every wave streams 1 KB per load instruction (64 lanes x 16 B) round-robin through a source window (private per block or shared by all
// blocks), keeping INFLIGHT loads outstanding with counted s_waitcnt vmcnt.  One block per CU (LDS request > 80 KB).
//   hipcc -O3 --offload-arch=gfx950 tools/probes/lds_dma_fill.hip -o tools/probes/lds_dma_fill.bin && tools/probes/lds_dma_fill.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

template <int WAVES, int INFLIGHT, int MODE>
__global__ __launch_bounds__(WAVES * 64) void fill_kernel(const unsigned char* src, unsigned window, int private_window, int iters, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned char* base = src + (private_window ? (size_t)blockIdx.x * window : 0);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    unsigned char* dst = smem + wave * (INFLIGHT * 1024);
    const unsigned mask = window - 1;
    unsigned off = (unsigned)(wave * 1024 + lane * 16) & mask;
    const unsigned step = WAVES * 1024;
    uint4 r[INFLIGHT];
#pragma unroll
    for (int s = 0; s < INFLIGHT; ++s) {
        const unsigned o = off;
        if (MODE == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + s * 1024), 16, o, 0, 0, 0);
        else r[s] = *(const uint4*)(base + o);
        off = (off + step) & mask;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < INFLIGHT; ++s) {
            wait_vm<INFLIGHT - 1>();                         // the oldest outstanding load has landed
            const unsigned o = off;
            if (MODE == 0) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + s * 1024), 16, o, 0, 0, 0);
            } else {
                *(uint4*)(dst + s * 1024 + lane * 16) = r[s];
                r[s] = *(const uint4*)(base + o);
            }
            off = (off + step) & mask;
        }
    }
    wait_vm<0>();
    if (MODE == 1) {
#pragma unroll
        for (int s = 0; s < INFLIGHT; ++s) *(uint4*)(dst + s * 1024 + lane * 16) = r[s];
    }
    __syncthreads();
    if (iters < 0) sink[threadIdx.x] = *(unsigned*)(smem + threadIdx.x * 4);      // keeps the LDS image alive for the compiler
}

// The K loop of conv_igemm2.hip in miniature: a wave-instruction gathers 8 ROWS of 128 bytes (8 lanes x 16 B each; rows `pitch` bytes
// apart, first byte `misalign` past a 128-byte boundary), a "step" is 4 such instructions per wave (one 64 x 64 bf16 operand stage per
// 4-wave block), AHEAD steps stay in flight, and BARRIER puts the block-wide s_barrier of the real loop after the landing wait.
template <int WAVES, int AHEAD, int BARRIER>
__global__ __launch_bounds__(WAVES * 64) void gather_kernel(const unsigned char* src, unsigned window, int private_window, unsigned pitch, unsigned misalign,
                                                           unsigned rowlen, int steps, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned char* base = src + (private_window ? (size_t)blockIdx.x * window : 0);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    constexpr int NSTAGE = AHEAD + 1;
    const unsigned rows = (window - misalign - 128) / pitch / (WAVES * 32) * (WAVES * 32);     // a whole number of block-wide row groups
    unsigned row = (unsigned)(wave * 32 + (lane >> 3)), k = 0;           // 4 instructions x 8 rows per wave and step
    auto issue = [&](int stage) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned o = (row + i * 8) * pitch + misalign + k + (lane & 7) * 16;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(smem + (stage * WAVES + wave) * 4096 + i * 1024), 16, o, 0, 0, 0);
        }
        k += 128;
        if (k + 128 > rowlen) { k = 0; row += WAVES * 32; if (row >= rows) row -= rows; }
    };
#pragma unroll
    for (int s = 0; s < AHEAD; ++s) issue(s);
    for (int t = 0; t < steps; t += NSTAGE) {                            // (straight-line: the LDS address of an LDS-DMA is an M0 value per site)
#pragma unroll
        for (int u = 0; u < NSTAGE; ++u) {
            wait_vm<4 * (AHEAD - 1)>();
            if (BARRIER) __builtin_amdgcn_s_barrier();
            issue((AHEAD + u) % NSTAGE);
        }
    }
    wait_vm<0>();
    __syncthreads();
    if (steps < 0) sink[threadIdx.x] = *(unsigned*)(smem + threadIdx.x * 4);
}

// The same gather addresses, but issued like fill_kernel: one load whenever the wave's oldest has landed (INFLIGHT outstanding per wave)
// instead of a burst of 4 per step - separates "the address pattern" from "the step-granular issue" as the cause of a lower rate.
template <int WAVES, int INFLIGHT>
__global__ __launch_bounds__(WAVES * 64) void gather1_kernel(const unsigned char* src, unsigned window, int private_window, unsigned pitch, unsigned misalign,
                                                            unsigned rowlen, int steps, unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned char* base = src + (private_window ? (size_t)blockIdx.x * window : 0);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    const unsigned rows = (window - misalign - 128) / pitch / (WAVES * 32) * (WAVES * 32);
    unsigned row = (unsigned)(wave * 32 + (lane >> 3)), k = 0;
    unsigned char* dst = smem + wave * (INFLIGHT * 1024);
    static_assert(INFLIGHT % 4 == 0, "four loads per step");
    auto one = [&](int s) {
        const unsigned o = (row + (s & 3) * 8) * pitch + misalign + k + (lane & 7) * 16;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + s * 1024), 16, o, 0, 0, 0);
        if ((s & 3) == 3) {
            k += 128;
            if (k + 128 > rowlen) { k = 0; row += WAVES * 32; if (row >= rows) row -= rows; }
        }
    };
#pragma unroll
    for (int s = 0; s < INFLIGHT; ++s) one(s);
    for (int t = 0; t < steps * 4 / INFLIGHT; ++t) {
#pragma unroll
        for (int s = 0; s < INFLIGHT; ++s) {
            wait_vm<INFLIGHT - 1>();
            one(s);
        }
    }
    wait_vm<0>();
    __syncthreads();
    if (steps < 0) sink[threadIdx.x] = *(unsigned*)(smem + threadIdx.x * 4);
}

// Wave-private K-interleaved pipelines (the design DESIGN.md section 7.1 proposes): every wave owns a ring of 2 stages of LOADS 1-KB loads
// (a whole 64 x 64 bf16 operand stage = 16), waits for ITS oldest stage only, "computes" for SLEEP x 64 clocks, re-issues the stage;
// no block barrier anywhere.
template <int WAVES, int LOADS, int SLEEP>
__global__ __launch_bounds__(WAVES * 64) void private_kernel(const unsigned char* src, unsigned window, int private_window, unsigned pitch, int steps,
                                                            unsigned* sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const unsigned char* base = src + (private_window ? (size_t)blockIdx.x * window : 0);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, 0x7fffffff, 0x00020000);
    const unsigned rows = (window - 128) / pitch / (LOADS * 8) * (LOADS * 8);
    unsigned row = (unsigned)(lane >> 3), k = (unsigned)wave * 128;            // wave w takes the K steps t = w (mod WAVES)
    unsigned char* dst = smem + wave * (2 * LOADS * 1024);
    auto issue = [&](int stage) {
#pragma unroll
        for (int i = 0; i < LOADS; ++i) {
            const unsigned o = (row + i * 8) * pitch + k + (lane & 7) * 16;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + (stage * LOADS + i) * 1024), 16, o, 0, 0, 0);
        }
        k += WAVES * 128;
        if (k + 128 > pitch) { k = (unsigned)wave * 128; row += LOADS * 8; if (row >= rows) row -= rows; }
    };
    issue(0);
    issue(1);
    for (int t = 0; t < steps; t += 2) {
        wait_vm<LOADS>();
        if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
        issue(0);
        wait_vm<LOADS>();
        if (SLEEP) __builtin_amdgcn_s_sleep(SLEEP);
        issue(1);
    }
    wait_vm<0>();
    __syncthreads();
    if (steps < 0) sink[threadIdx.x] = *(unsigned*)(smem + threadIdx.x * 4);
}

struct Result { int waves, inflight, mode, grid, priv; unsigned window; double us, gbs_cu, bclk_cu; };

struct GResult { int waves, ahead, barrier, grid; unsigned pitch, misalign, rowlen; double us, bclk_cu; };

template <int WAVES, int AHEAD, int BARRIER>
static int run_gather(const unsigned char* src, unsigned pitch, unsigned misalign, unsigned rowlen, int grid, unsigned* sink, std::vector<GResult>& out) {
    const unsigned window = 256u << 10;                                  // grid 64: private (16 MB in all, L2-resident); grid 256: one shared window
    const int priv = grid <= 64;
    const int steps = (int)((8u << 20) / (WAVES * 4096)) - AHEAD;
    const size_t lds = 96 * 1024;
    auto k = gather_kernel<WAVES, AHEAD, BARRIER>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k, dim3(grid), dim3(WAVES * 64), lds, 0, src, window, priv, pitch, misalign, rowlen, steps, sink);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    CK(hipGetLastError());
    const double bytes = (double)WAVES * 4096 * ((steps + AHEAD) / (AHEAD + 1) * (AHEAD + 1) + AHEAD), us = best * 1e3 - 4.0;
    GResult r = {WAVES, AHEAD, BARRIER, grid, pitch, misalign, rowlen, best * 1e3, bytes / (us * 1e-6) / 2.4e9};
    out.push_back(r);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return 0;
}

template <int WAVES, int INFLIGHT, int MODE>
static int run(const unsigned char* src, unsigned window, int priv, int grid, unsigned* sink, std::vector<Result>& out) {
    const size_t per_block = 8u << 20;                                   // bytes every block pulls
    const int iters = (int)(per_block / ((size_t)WAVES * INFLIGHT * 1024)) - 1;
    const size_t lds = 96 * 1024;                                        // one block per CU
    auto k = fill_kernel<WAVES, INFLIGHT, MODE>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k, dim3(grid), dim3(WAVES * 64), lds, 0, src, window, priv, iters, sink);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    CK(hipGetLastError());
    const double bytes = (double)WAVES * INFLIGHT * 1024 * (iters + 1);
    const double us = best * 1e3 - 4.0;                                   // minus the launch floor measured for these grids
    Result r = {WAVES, INFLIGHT, MODE, grid, priv, window, best * 1e3, bytes / (us * 1e-6) / 1e9, bytes / (us * 1e-6) / 2.4e9};
    out.push_back(r);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return 0;
}

template <int WAVES, int MODE>
static int sweep_inflight(const unsigned char* src, unsigned window, int priv, int grid, unsigned* sink, std::vector<Result>& out) {
    if (run<WAVES, 1, MODE>(src, window, priv, grid, sink, out)) return 1;
    if (run<WAVES, 2, MODE>(src, window, priv, grid, sink, out)) return 1;
    if (run<WAVES, 4, MODE>(src, window, priv, grid, sink, out)) return 1;
    if (WAVES * 8 <= 96 && run<WAVES, (WAVES * 8 <= 96 ? 8 : 1), MODE>(src, window, priv, grid, sink, out)) return 1;
    if (WAVES * 16 <= 96 && MODE == 0 && run<WAVES, (WAVES * 16 <= 96 ? 16 : 1), MODE>(src, window, priv, grid, sink, out)) return 1;
    return 0;
}

template <int WAVES, int INFLIGHT>
static int run_gather1(const unsigned char* src, unsigned pitch, unsigned misalign, unsigned rowlen, int grid, unsigned* sink, std::vector<GResult>& out) {
    const unsigned window = 256u << 10;
    const int priv = grid <= 64;
    const int steps = (int)((8u << 20) / (WAVES * 4096)) - INFLIGHT / 4;
    const size_t lds = 96 * 1024;
    auto k = gather1_kernel<WAVES, INFLIGHT>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k, dim3(grid), dim3(WAVES * 64), lds, 0, src, window, priv, pitch, misalign, rowlen, steps, sink);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    CK(hipGetLastError());
    const double bytes = (double)WAVES * 1024 * INFLIGHT * (steps * 4 / INFLIGHT + 1), us = best * 1e3 - 4.0;
    GResult r = {WAVES, -INFLIGHT, 0, grid, pitch, misalign, rowlen, best * 1e3, bytes / (us * 1e-6) / 2.4e9};      // ahead < 0: loads in flight per wave
    out.push_back(r);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return 0;
}

template <int WAVES, int LOADS, int SLEEP>
static int run_private(const unsigned char* src, unsigned pitch, int grid, unsigned* sink) {
    const unsigned window = 256u << 10;
    const int priv = grid <= 64;
    const int steps = (int)((8u << 20) / (WAVES * LOADS * 1024)) - 2;
    const size_t lds = (size_t)WAVES * 2 * LOADS * 1024 > 96 * 1024 ? (size_t)WAVES * 2 * LOADS * 1024 : 96 * 1024;
    auto k = private_kernel<WAVES, LOADS, SLEEP>;
    CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k, dim3(grid), dim3(WAVES * 64), lds, 0, src, window, priv, pitch, steps, sink);
        CK(hipEventRecord(e1, 0));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
    }
    CK(hipGetLastError());
    const double bytes = (double)WAVES * LOADS * 1024 * (steps / 2 * 2 + 2), us = best * 1e3 - 4.0;
    printf("private,%d,%d,%d,%d,%u,%.1f,%.2f\n", grid, WAVES, LOADS, SLEEP, pitch, best * 1e3, bytes / (us * 1e-6) / 2.4e9);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return 0;
}

static int private_main(const unsigned char* src, unsigned* sink) {
    printf("private: grid,waves,loads_per_stage,sleep_x64clk,pitch,us,B_per_clk_per_cu\n");
    for (int gi = 0; gi < 2; ++gi) {
        const int grid = gi == 0 ? 64 : 256;
        for (unsigned pitch = 768; pitch <= 1536; pitch += 768) {
            if (run_private<4, 16, 0>(src, pitch, grid, sink)) return 1;
            if (run_private<4, 16, 8>(src, pitch, grid, sink)) return 1;
            if (run_private<4, 16, 15>(src, pitch, grid, sink)) return 1;
            if (run_private<4, 8, 0>(src, pitch, grid, sink)) return 1;
            if (run_private<4, 8, 8>(src, pitch, grid, sink)) return 1;
            if (run_private<4, 4, 0>(src, pitch, grid, sink)) return 1;
            if (run_private<4, 4, 4>(src, pitch, grid, sink)) return 1;
            if (run_private<8, 8, 0>(src, pitch, grid, sink)) return 1;
            if (run_private<8, 8, 15>(src, pitch, grid, sink)) return 1;
        }
    }
    return 0;
}

static int gather_main(const unsigned char* src, unsigned* sink) {
    std::vector<GResult> out;
    const unsigned pat[7][3] = {{128, 0, 128}, {256, 0, 256}, {256, 64, 256}, {768, 0, 768}, {768, 64, 768}, {192, 0, 192}, {832, 0, 768}};
    for (int gi = 0; gi < 2; ++gi) {
        const int grid = gi == 0 ? 64 : 256;
        for (int pi = 0; pi < 7; ++pi) {
            const unsigned p = pat[pi][0], m = pat[pi][1], rl = pat[pi][2];
            if (run_gather<4, 1, 0>(src, p, m, rl, grid, sink, out)) return 1;
            if (run_gather<4, 2, 0>(src, p, m, rl, grid, sink, out)) return 1;
            if (run_gather<4, 3, 0>(src, p, m, rl, grid, sink, out)) return 1;
            if (run_gather<4, 4, 0>(src, p, m, rl, grid, sink, out)) return 1;
            if (run_gather<4, 3, 1>(src, p, m, rl, grid, sink, out)) return 1;
            if (run_gather<8, 3, 1>(src, p, m, rl, grid, sink, out)) return 1;
            if (run_gather1<4, 4>(src, p, m, rl, grid, sink, out)) return 1;
            if (run_gather1<4, 8>(src, p, m, rl, grid, sink, out)) return 1;
            if (run_gather1<4, 16>(src, p, m, rl, grid, sink, out)) return 1;
            if (run_gather1<8, 8>(src, p, m, rl, grid, sink, out)) return 1;
        }
    }
    printf("gather: grid,waves,steps_ahead (negative: single loads in flight per wave),barrier,pitch,misalign,rowlen,us,B_per_clk_per_cu\n");
    for (const GResult& r : out)
        printf("gather,%d,%d,%d,%d,%u,%u,%u,%.1f,%.2f\n", r.grid, r.waves, r.ahead, r.barrier, r.pitch, r.misalign, r.rowlen, r.us, r.bclk_cu);
    return 0;
}

int main(int argc, char** argv) {
    const size_t total = 512u << 20;
    unsigned char* src; unsigned* sink;
    CK(hipMalloc(&src, total)); CK(hipMemset(src, 1, total)); CK(hipMalloc(&sink, 4096 * 4));
    if (argc > 1 && argv[1][0] == 'g') return gather_main(src, sink);
    if (argc > 1 && argv[1][0] == 'p') return private_main(src, sink);
    std::vector<Result> out;
    const unsigned windows[3] = {64u << 10, 1u << 20, 4u << 20};          // private 64 KB tiles (16 MB in all: L2-resident); private 1 MB (256 MB in all: MALL / HBM); one SHARED 4 MB bank (a filter bank every block reads)
    for (int wi = 0; wi < 3; ++wi) {
        const unsigned window = windows[wi];
        const int priv = wi < 2;
        for (int gi = 0; gi < 2; ++gi) {
            const int grid = gi == 0 ? 64 : 256;
            if (sweep_inflight<1, 0>(src, window, priv, grid, sink, out)) return 1;
            if (sweep_inflight<2, 0>(src, window, priv, grid, sink, out)) return 1;
            if (sweep_inflight<4, 0>(src, window, priv, grid, sink, out)) return 1;
            if (sweep_inflight<8, 0>(src, window, priv, grid, sink, out)) return 1;
            if (sweep_inflight<16, 0>(src, window, priv, grid, sink, out)) return 1;
            if (wi == 0) {
                if (sweep_inflight<4, 1>(src, window, priv, grid, sink, out)) return 1;
                if (sweep_inflight<8, 1>(src, window, priv, grid, sink, out)) return 1;
            }
        }
    }
    printf("mode,window_kb,private,grid,waves,inflight_per_wave,kb_in_flight_per_cu,us,GBps_per_cu,B_per_clk_per_cu\n");
    for (const Result& r : out)
        printf("%s,%u,%d,%d,%d,%d,%d,%.1f,%.1f,%.2f\n", r.mode ? "vgpr" : "ldsdma", r.window >> 10, r.priv, r.grid, r.waves, r.inflight,
               r.waves * r.inflight, r.us, r.gbs_cu, r.bclk_cu);
    return 0;
}
