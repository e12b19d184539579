// hipcc (ROCm 7.2) gives the closure types of immediately-invoked lambdas that initialise EXTERNAL-linkage namespace-scope variables of one
// translation unit the same mangled name: the second variable takes the first one's initialiser.  This is how libfasterseg_hip's
// fs::g_fp32x3 silently read FS_DETERMINISTIC instead of FS_FP32_X3 (round 6, csrc/api.cpp).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probes/hipcc_lambda_init.hip -o /tmp/p && /tmp/p     prints "1 2 3 3 56" (expected 1 2 3 4 56)
//   g++ -x c++ -O2 -std=c++17 tools/probes/hipcc_lambda_init.hip -o /tmp/q && /tmp/q                       prints "1 2 3 4 56"
// `static` variables (a, b) and function-local statics (f) are not affected.
#include <stdio.h>
#include <stdlib.h>
static int a = [] { const char* e = getenv("AA"); return e ? atoi(e) : 1; }();
static int b = [] { const char* e = getenv("BB"); return e ? atoi(e) : 2; }();
namespace fs { int c = [] { const char* e = getenv("CC"); return e ? atoi(e) : 3; }(); }
namespace fs { int d = [] { const char* e = getenv("DD"); return e ? atoi(e) : 4; }(); }
static int f() { static const int v = [] { return 5; }(); static const int w = [] { return 6; }(); return v * 10 + w; }
int main() { printf("%d %d %d %d %d\n", a, b, fs::c, fs::d, f()); return 0; }
