// Error channel + version of libfasterseg_hip (host only).
#include <stdarg.h>
#include <stdio.h>
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

extern "C" const char* fs_last_error(void) { return fs::g_err; }
extern "C" int fs_version(void) { return 130; }
