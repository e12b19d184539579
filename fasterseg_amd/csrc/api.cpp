// Error channel + version of libfasterseg_hip (host only).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <map>
#include <mutex>
#include <string>
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

// ---- launch census ---------------------------------------------------------------------------------------------------
// Counts the convolution launches issued through the C ABI by geometry (also while a hipGraph is being captured: what is
// counted then is what every replay runs).  bench.py uses it to know WHICH conv shapes a train step is made of, times each
// shape alone with HIP events, and derives the roofline line of the step's dominant kernel from launches x duration.
namespace fs {
int g_census_on = 0;
static std::mutex g_census_mutex;                       // autograd runs backward on its own thread
static std::map<std::string, fs_census_entry> g_census;
void census_conv(int family, const fs_conv_desc* d) {
    fs_census_entry e;
    memset(&e, 0, sizeof(e));
    e.family = family;
    e.desc = *d;
    std::string key((const char*)&e, sizeof(int) + sizeof(fs_conv_desc));
    std::lock_guard<std::mutex> lock(g_census_mutex);
    auto it = g_census.find(key);
    if (it == g_census.end()) {
        e.count = 1;
        g_census.emplace(key, e);
    } else {
        it->second.count += 1;
    }
}
}  // namespace fs

extern "C" void fs_census_enable(int on) {
    std::lock_guard<std::mutex> lock(fs::g_census_mutex);
    if (on) fs::g_census.clear();
    fs::g_census_on = on ? 1 : 0;
}

extern "C" int fs_census_read(fs_census_entry* out, int max_entries) {
    std::lock_guard<std::mutex> lock(fs::g_census_mutex);
    int n = 0;
    for (auto& kv : fs::g_census) {
        if (out && n < max_entries) out[n] = kv.second;
        ++n;
    }
    return n;
}
