// Error channel + version of libfasterseg_hip (host only).
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../include/fasterseg_hip.h"

namespace fs {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace fs

namespace fs {
int g_deterministic = [] { const char* e = getenv("FS_DETERMINISTIC"); return (e && atoi(e) > 0) ? 1 : 0; }();
}  // namespace fs

extern "C" void fs_set_deterministic(int on) { fs::g_deterministic = on ? 1 : 0; }
extern "C" int fs_get_deterministic(void) { return fs::g_deterministic; }
extern "C" const char* fs_last_error(void) { return fs::g_err; }
// (a named function, not a second immediately-invoked lambda: hipcc gave both namespace-scope initialiser lambdas of this file ONE mangled
//  name and this flag silently took FS_DETERMINISTIC's initialiser - it read 0 whatever FS_FP32_X3 said, round 6)
static int env_fp32_split() { const char* e = getenv("FS_FP32_X3"); return e ? (atoi(e) != 0) : 1; }
namespace fs { int g_fp32x3 = env_fp32_split(); }
extern "C" void fs_set_fp32_split(int on) { fs::g_fp32x3 = on ? 1 : 0; }
extern "C" int fs_get_fp32_split(void) { return fs::g_fp32x3; }
extern "C" int fs_version(void) { return FS_ABI_VERSION; }
/* sizeof of the descriptor structs this library was compiled with (0: conv, 1: resize, 2: zoom, 3: sgd tensor): a binding whose
 * struct layout differs from the header it was written against must fail at load, not read garbage fields. */
extern "C" int fs_struct_size(int which) {
    switch (which) {
        case 0: return (int)sizeof(fs_conv_desc);
        case 1: return (int)sizeof(fs_resize_desc);
        case 2: return (int)sizeof(fs_zoom_desc);
        case 3: return (int)sizeof(fs_sgd_tensor);
        case 4: return (int)sizeof(fs_logits_desc);
        default: return -1;
    }
}

