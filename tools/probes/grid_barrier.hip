// Probe behind the one-launch conv -> BatchNorm(train) -> ReLU unit: what does a barrier among the workgroups of
// ONE launch cost on MI355X (8 XCDs, one L2 each), and when is it safe?
//
//   phase A  every block adds 128 floats into a statistics array with agent-scope float atomics (what the conv epilogue does)
//   barrier  one thread per block bumps an arrival counter (release, agent scope) and polls it with agent-scope loads until all blocks
//            of its PANEL (consecutive `panel` blocks) have arrived - BOUNDED: a poll budget turns a missing co-resident block into a
//            counted timeout instead of a hang
//   phase B  every block reads the 128 totals back (agent-scope loads) and writes one value
//
// Reported per (grid, panel, LDS bytes per block): microseconds per launch with and without the barrier (HIP events over back-to-back
// launches on one stream), then the same with 4 streams issuing concurrently (the supernet step's eager passes run on 4 lanes), and the
// number of blocks whose poll budget ran out.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/grid_barrier.hip -o tools/probes/grid_barrier.bin && tools/probes/grid_barrier.bin
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Args {
    float* stats;            // [panels][128]
    unsigned* counters;      // [panels][2]: arrivals, departures (left zero by the last block to leave)
    unsigned* spread;        // variant 2: [panels][8][32] arrival counters
    unsigned* timeouts;      // blocks that gave up
    float* out;
    int panel;               // blocks per panel
    int barrier;             // 0: phases A and B only
    int work;                // s_sleep units between start and phase A (stands in for the K loop; blocks then arrive spread out)
    int variant;             // 0: release arrival + acquire polls (buffer_wbl2 / buffer_inv sc1 per access: the first version of this probe)
                             // 1: relaxed agent-scope arrival and polls (the totals are atomics and are read with sc1 loads: nothing to flush)
                             // 2: variant 1 with the arrivals spread over 8 counters on separate cache lines (same-address atomics serialise)
};

__global__ __launch_bounds__(256) void barrier_kernel(Args a) {
    extern __shared__ unsigned char smem[];
    __shared__ int flag;
    const int tid = threadIdx.x;
    const int p = blockIdx.x / a.panel;
    float* st = a.stats + p * 128;
    for (int i = 0; i < a.work; ++i) __builtin_amdgcn_s_sleep(64);
    if (tid < 128) __hip_atomic_fetch_add(st + tid, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a.barrier) {
        __builtin_amdgcn_s_waitcnt(0);
        __syncthreads();
        if (tid == 0) {
            unsigned* c = a.counters + 2 * p;
            int ok = 0;
            if (a.variant == 0) {
                __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                for (int it = 0; it < (1 << 13); ++it) {          // bounded: 8 k polls of ~1 us each
                    if (__hip_atomic_load(c, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)a.panel) { ok = 1; break; }
                    __builtin_amdgcn_s_sleep(8);
                }
            } else if (a.variant == 1) {
                __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int it = 0; it < (1 << 14); ++it) {
                    if (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)a.panel) { ok = 1; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
            } else {
                unsigned* sub = a.spread + (size_t)p * 8 * 32;     // 8 counters, 128 bytes apart
                __hip_atomic_fetch_add(sub + (blockIdx.x & 7) * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                for (int it = 0; it < (1 << 14); ++it) {
                    unsigned tot = 0;
#pragma unroll
                    for (int q = 0; q < 8; ++q) tot += __hip_atomic_load(sub + q * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (tot >= (unsigned)a.panel) { ok = 1; break; }
                    __builtin_amdgcn_s_sleep(2);
                }
            }
            if (!ok) atomicAdd(a.timeouts, 1u);
            flag = ok;
        }
        __syncthreads();
    }
    float v = 0.f;
    if (tid < 128) v = __hip_atomic_load(st + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (a.barrier && tid < 128 && flag && v != (float)a.panel) atomicAdd(a.timeouts + 1, 1u);      // a total that is not complete after the barrier
    if (tid == 0) a.out[blockIdx.x] = v + (smem[0] ? 0.f : 0.f);
    if (a.barrier) {
        __syncthreads();                                           // every thread of the block has read its totals
        if (tid == 0) {
            unsigned* c = a.counters + 2 * p;
            if (__hip_atomic_fetch_add(c + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)a.panel - 1u) {
                // last block of the panel to leave: everything goes back to zero for the next launch on this stream
                for (int i = 0; i < 128; ++i) __hip_atomic_store(st + i, 0.f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (a.variant == 2)
                    for (int q = 0; q < 8; ++q) __hip_atomic_store(a.spread + ((size_t)p * 8 + q) * 32, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(c + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

__global__ void zero_stats(float* s, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) s[i] = 0.f;
}

__global__ void empty_kernel(float* p) { if (p == nullptr && threadIdx.x == 999) *p = 0.f; }

int main(int argc, char** argv) {
    const int MAXP = 4096;
    const int NS = 4;
    hipStream_t st[NS];
    Args base[NS];
    for (int s = 0; s < NS; ++s) {
        CK(hipStreamCreate(&st[s]));
        CK(hipMalloc(&base[s].stats, MAXP * 128 * 4));
        CK(hipMalloc(&base[s].counters, MAXP * 2 * 4));
        CK(hipMalloc(&base[s].timeouts, 8));
        CK(hipMalloc(&base[s].spread, MAXP * 8 * 32 * 4));
        CK(hipMemset(base[s].spread, 0, MAXP * 8 * 32 * 4));
        CK(hipMalloc(&base[s].out, 8192 * 4));
        CK(hipMemset(base[s].stats, 0, MAXP * 128 * 4));
        CK(hipMemset(base[s].counters, 0, MAXP * 2 * 4));
        CK(hipMemset(base[s].timeouts, 0, 8));
    }
    CK(hipFuncSetAttribute((const void*)barrier_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int REP = 200;
    // launch floor of this box
    {
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(empty_kernel, dim3(64), dim3(256), 0, st[0], base[0].out);
        CK(hipStreamSynchronize(st[0]));
        CK(hipEventRecord(e0, st[0]));
        for (int i = 0; i < REP; ++i) hipLaunchKernelGGL(empty_kernel, dim3(64), dim3(256), 0, st[0], base[0].out);
        CK(hipEventRecord(e1, st[0]));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("empty kernel, 64 blocks, back to back on one stream: %.2f us per launch\n", ms * 1e3 / REP);
    }
    printf("%6s %6s %8s %5s | %10s %10s %10s | %12s %12s | %s\n", "grid", "panel", "lds_KB", "work", "plain_us", "barrier_us", "delta_us", "4str_plain", "4str_barrier",
           "timeouts/incomplete");
    const int full = argc > 1 && argv[1][0] == 'f';              // "full": the wide sweep of the first run (variant 0 included)
    std::vector<int> grids = {32, 64, 96, 128, 192, 256, 512};
    std::vector<int> ldss = {66};
    std::vector<int> works = {0, 40};
    std::vector<int> variants = {1, 2};
    if (full) { grids = {32, 64, 96, 128, 192, 256, 384, 512, 768, 1024}; ldss = {32, 66}; variants = {0, 1, 2}; }
    for (int variant : variants)
    for (int lds : ldss)
        for (int work : works)
            for (int grid : grids) {
                int panels[3] = {grid, grid >= 64 ? grid / 4 : grid, 16};
                for (int pi = 0; pi < 3; ++pi) {
                    const int panel = panels[pi];
                    if (pi > 0 && panel == panels[pi - 1]) continue;
                    if (grid % panel) continue;
                    float res[4];
                    for (int mode = 0; mode < 4; ++mode) {           // 0: 1 stream plain, 1: 1 stream barrier, 2: 4 streams plain, 3: 4 streams barrier
                        const int ns = mode >= 2 ? NS : 1;
                        const int bar = mode & 1;
                        for (int s = 0; s < ns; ++s) {
                            Args a = base[s];
                            a.panel = panel; a.barrier = bar; a.work = work; a.variant = variant;
                            for (int i = 0; i < 5; ++i) {
                                hipLaunchKernelGGL(barrier_kernel, dim3(grid), dim3(256), lds * 1024, st[s], a);
                                if (!bar) hipLaunchKernelGGL(zero_stats, dim3((grid / panel * 128 + 255) / 256), dim3(256), 0, st[s], a.stats, grid / panel * 128);
                            }
                        }
                        CK(hipDeviceSynchronize());
                        CK(hipEventRecord(e0, st[0]));
                        for (int s = 1; s < ns; ++s) CK(hipStreamWaitEvent(st[s], e0, 0));
                        for (int i = 0; i < REP; ++i)
                            for (int s = 0; s < ns; ++s) {
                                Args a = base[s];
                                a.panel = panel; a.barrier = bar; a.work = work; a.variant = variant;
                                hipLaunchKernelGGL(barrier_kernel, dim3(grid), dim3(256), lds * 1024, st[s], a);
                            }
                        hipEvent_t done[NS];
                        for (int s = 1; s < ns; ++s) {
                            CK(hipEventCreate(&done[s]));
                            CK(hipEventRecord(done[s], st[s]));
                            CK(hipStreamWaitEvent(st[0], done[s], 0));
                        }
                        CK(hipEventRecord(e1, st[0]));
                        CK(hipEventSynchronize(e1));
                        CK(hipDeviceSynchronize());
                        for (int s = 1; s < ns; ++s) CK(hipEventDestroy(done[s]));
                        float ms;
                        CK(hipEventElapsedTime(&ms, e0, e1));
                        res[mode] = ms * 1e3 / REP;                  // per launch ROUND (4 concurrent launches in modes 2, 3)
                        if (!bar)
                            for (int s = 0; s < ns; ++s) CK(hipMemsetAsync(base[s].stats, 0, MAXP * 128 * 4, st[s]));
                        CK(hipDeviceSynchronize());
                    }
                    unsigned to[2] = {0, 0}, tsum[2] = {0, 0};
                    for (int s = 0; s < NS; ++s) {
                        CK(hipMemcpy(to, base[s].timeouts, 8, hipMemcpyDeviceToHost));
                        tsum[0] += to[0]; tsum[1] += to[1];
                        CK(hipMemset(base[s].timeouts, 0, 8));
                        CK(hipMemset(base[s].counters, 0, MAXP * 2 * 4));      // a timed-out panel leaves its counters dirty
                        CK(hipMemset(base[s].spread, 0, MAXP * 8 * 32 * 4));
                        CK(hipMemset(base[s].stats, 0, MAXP * 128 * 4));
                    }
                    printf("v%d %6d %6d %8d %5d | %10.2f %10.2f %10.2f | %12.2f %12.2f | %u/%u\n", variant, grid, panel, lds, work, res[0], res[1], res[1] - res[0], res[2], res[3],
                           tsum[0], tsum[1]);
                    fflush(stdout);
                }
            }
    return 0;
}
