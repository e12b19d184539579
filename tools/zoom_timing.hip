// Developer tool: where does the time of one fs_zoom_cell_fwd launch go?  Builds zoom_cell.hip with shader-clock stamps at the
// phase boundaries (block 0, wave 0) and prints the cycle count of each phase next to the HIP-event time of the launch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DFS_ZOOM_TIMING tools/zoom_timing.hip fasterseg_amd/csrc/api.cpp -o scratch/zoom_timing
#include "../fasterseg_amd/csrc/zoom_cell.hip"
#include <vector>

static ZoomArgs make(int N, int H, int W, int Cin, int C, int down, int up, int dtype) {
    ZoomArgs a{};
    a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cmid = C; a.Cout = C;
    a.h = down ? H / 2 : H; a.w = down ? W / 2 : W;
    a.Ho = up ? 2 * a.h : a.h; a.Wo = up ? 2 * a.w : a.w;
    a.x_cs = Cin; a.y_cs = C; a.down = down; a.up = up;
    const int TH = up ? ZR2 - 2 : ZR2, TW = up ? 12 : 14;
    a.tiles_y = (a.h + TH - 1) / TH; a.tiles_x = (a.w + TW - 1) / TW;
    const int ck = 4 * vec_elems(dtype);
    a.nch1 = (Cin + ck - 1) / ck; a.nch2 = (C + ck - 1) / ck;
    a.rh_dn = zoom_scale(H, a.h); a.rw_dn = zoom_scale(W, a.w); a.rh_up = zoom_scale(a.h, a.Ho); a.rw_up = zoom_scale(a.w, a.Wo);
    return a;
}

template <int NT> static void run(const char* name, int N, int H, int W, int Cin, int C, int down, int up) {
    ZoomArgs a = make(N, H, W, Cin, C, down, up, FS_BF16);
    size_t xb = (size_t)N * H * W * Cin * 2, yb = (size_t)N * a.Ho * a.Wo * C * 2;
    size_t w1b = (size_t)8 * a.nch1 * 18 * 1024, w2b = (size_t)8 * a.nch2 * 18 * 1024;
    void *x, *y, *w1, *w2; unsigned long long* dbg;
    hipMalloc(&x, xb); hipMalloc(&y, yb); hipMalloc(&w1, w1b); hipMalloc(&w2, w2b); hipMalloc(&dbg, 64);
    hipMemset(x, 0x3c, xb); hipMemset(w1, 0x3a, w1b); hipMemset(w2, 0x3a, w2b);
    a.x = (const unsigned char*)x; a.y = (unsigned char*)y; a.w1 = (const unsigned char*)w1; a.w2 = (const unsigned char*)w2; a.dbg = dbg;
    hipStream_t st; hipStreamCreate(&st);
    for (int i = 0; i < 5; ++i) launch_zoom<bf16_t, NT>(st, a);
    hipStreamSynchronize(st);
    unsigned long long h[8];
    hipMemcpy(h, dbg, 64, hipMemcpyDeviceToHost);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, st);
    for (int i = 0; i < 200; ++i) launch_zoom<bf16_t, NT>(st, a);
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-34s blocks %4d  %6.2f us/launch | cycles: prologue %5llu  stage0 %5llu  conv1 %6llu  epi1 %5llu  conv2 %6llu  epi2 %5llu  out %5llu  total %6llu\n",
           name, a.N * a.tiles_x * a.tiles_y, ms * 1e3 / 200, h[0] - h[7], h[1] - h[0], h[2] - h[1], h[3] - h[2], h[4] - h[3], h[5] - h[4],
           h[6] - h[5], h[6] - h[7]);
    hipFree(x); hipFree(y); hipFree(w1); hipFree(w2); hipFree(dbg);
}

int main() {
    run<1>("dn up 32->32->32 conv@64x128", 1, 128, 256, 32, 32, 1, 1);
    run<1>("plain 32->32->32 conv@128x256", 1, 128, 256, 32, 32, 0, 0);
    run<2>("dn up 64->64->64 conv@32x64", 1, 64, 128, 64, 64, 1, 1);
    run<2>("dn up 128->64->64 conv@32x64", 1, 64, 128, 128, 64, 1, 1);
    run<4>("dn 64->128->128 conv@32x64", 1, 64, 128, 64, 128, 1, 0);
    run<4>("dn up 128->128->128 conv@16x32", 1, 32, 64, 128, 128, 1, 1);
    run<6>("dn up 64->192->192 conv@32x64", 1, 64, 128, 64, 192, 1, 1);
    run<8>("dn up 128->256->256 conv@16x32", 1, 32, 64, 128, 256, 1, 1);
    return 0;
}
