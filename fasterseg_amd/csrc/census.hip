// Launch census and in-step kernel timing (measurement support; host only).
//
// Level 1 counts the convolution launches issued through the C ABI by geometry (also while a hipGraph is being captured:
// what is counted then is what every replay runs).  Level 2 additionally TIMES every kernel this library launches: FS_LAUNCH
// (common.h) hands hipExtLaunchKernelGGL a start/stop event pair, so the elapsed time of a pair is the dispatch's own
// begin -> end interval as the command processor stamps it - the quantity rocprofv3's kernel trace reports - measured on
// the stream the kernel runs on, inside the step that is being priced (every launch, in its real cache state, not isolated
// warm replays).  bench.py turns {family, geometry} x {launches, FLOPs, measured time} into the `roofline` object and prints
// the per-kernel table; profiles/ holds the rocprofv3 summary of the same command for cross-checking.
// Launches issued while a stream is capturing cannot carry events and are left untimed (the census step runs eagerly).
#include <string.h>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>
#include "common.h"

namespace fs {

int g_census_on = 0;

namespace {

struct KernelAcc {
    long long count = 0;
    double ms = 0.0;
    double bytes = 0.0;          // algorithmic HBM bytes the launchers noted (census_note_bytes)
};
struct Pending {
    hipEvent_t e0, e1;
    int kernel;                  // index into g_names
    int tag;                     // caller's tag at launch time (fs_census_tag), -1: none
    fs_census_entry* entry;      // geometry entry the launch belongs to (or null)
    // grouped launch: its time is shared out over several entries.  Held by value (shared with the scope that made it), never as an index
    // into a container a harvest can clear: a live CensusGroupScope outlives the harvests census_events / fs_census_read trigger
    std::shared_ptr<const std::vector<std::pair<fs_census_entry*, double>>> group;
};

std::recursive_mutex g_mutex;                        // autograd runs backward on its own thread
std::map<std::string, fs_census_entry> g_census;     // node-based: entry addresses are stable
std::vector<std::string> g_names;
std::map<std::string, int> g_name_index;
std::vector<KernelAcc> g_kernels;
std::vector<Pending> g_pending;
std::vector<std::pair<hipEvent_t, hipEvent_t>> g_free;
thread_local fs_census_entry* t_scope = nullptr;
thread_local double t_bytes = 0.0;                   // census_note_bytes: consumed by the next timed launch of this thread
thread_local std::shared_ptr<const std::vector<std::pair<fs_census_entry*, double>>> t_group;     // (entry, share of the launch's duration)
std::map<int, KernelAcc> g_tags;
int g_tag = -1;
constexpr size_t MAX_PENDING = 8192;                 // harvest (one stream sync) when this many launches are outstanding

std::string clean_name(const char* raw) {            // "(conv_igemm_kernel<T, WAVES_M, ...>)" -> "conv_igemm_kernel"
    std::string s(raw);
    size_t b = 0;
    while (b < s.size() && (s[b] == '(' || s[b] == ' ')) ++b;
    size_t e = b;
    while (e < s.size() && s[e] != '<' && s[e] != ')' && s[e] != ' ') ++e;
    return s.substr(b, e - b);
}

void harvest() {
    for (Pending& p : g_pending) {
        float ms = 0.f;
        if (hipEventSynchronize(p.e1) == hipSuccess && hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
            g_kernels[p.kernel].ms += ms;
            if (p.entry) p.entry->ms += ms;
            if (p.group)
                for (auto& es : *p.group) es.first->ms += ms * es.second;
            if (p.tag >= 0) {
                KernelAcc& t = g_tags[p.tag];
                t.count += 1;
                t.ms += ms;
            }
        } else {
            (void)hipGetLastError();
        }
        g_free.emplace_back(p.e0, p.e1);
    }
    g_pending.clear();
}

}  // namespace

// One launch that serves several (family, geometry) entries: each is counted once and receives `share[i]` of the measured duration
CensusGroupScope::CensusGroupScope(int family, const fs_conv_desc* const* d, const double* share, int n) : live(false) {
    if (!g_census_on) return;
    std::lock_guard<std::recursive_mutex> lock(g_mutex);
    auto grp = std::make_shared<std::vector<std::pair<fs_census_entry*, double>>>();
    for (int i = 0; i < n; ++i) {
        fs_census_entry e;
        memset(&e, 0, sizeof(e));
        e.family = family;
        e.desc = *d[i];
        std::string key((const char*)&e, sizeof(int) + sizeof(fs_conv_desc));
        auto it = g_census.find(key);
        if (it == g_census.end()) it = g_census.emplace(key, e).first;
        it->second.count += 1;
        grp->emplace_back(&it->second, share[i]);
    }
    t_group = grp;
    live = true;
}

CensusGroupScope::~CensusGroupScope() {
    if (live) t_group.reset();
}

CensusScope::CensusScope(int family, const fs_conv_desc* d) : live(false) {
    if (!g_census_on) return;
    fs_census_entry e;
    memset(&e, 0, sizeof(e));
    e.family = family;
    e.desc = *d;
    std::string key((const char*)&e, sizeof(int) + sizeof(fs_conv_desc));
    std::lock_guard<std::recursive_mutex> lock(g_mutex);
    auto it = g_census.find(key);
    if (it == g_census.end()) it = g_census.emplace(key, e).first;
    it->second.count += 1;
    t_scope = &it->second;
    live = true;
}

CensusScope::~CensusScope() {
    if (live) t_scope = nullptr;
}

void census_note_bytes(double bytes) { t_bytes = bytes; }

bool census_events(const char* kernel, hipStream_t stream, hipEvent_t* start, hipEvent_t* stop) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        return false;
    }
    std::lock_guard<std::recursive_mutex> lock(g_mutex);
    if (g_census_on < 2) return false;
    if (g_pending.size() >= MAX_PENDING) harvest();
    std::pair<hipEvent_t, hipEvent_t> ev;
    if (!g_free.empty()) {
        ev = g_free.back();
        g_free.pop_back();
    } else {
        if (hipEventCreate(&ev.first) != hipSuccess || hipEventCreate(&ev.second) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
    }
    const std::string name = clean_name(kernel);
    auto it = g_name_index.find(name);
    int k;
    if (it == g_name_index.end()) {
        k = (int)g_names.size();
        g_names.push_back(name);
        g_kernels.emplace_back();
        g_name_index.emplace(name, k);
    } else {
        k = it->second;
    }
    g_kernels[k].count += 1;
    g_kernels[k].bytes += t_bytes;
    t_bytes = 0.0;
    g_pending.push_back(Pending{ev.first, ev.second, k, g_tag, t_scope, t_group});
    *start = ev.first;
    *stop = ev.second;
    return true;
}

}  // namespace fs

extern "C" void fs_census_enable(int level) {
    std::lock_guard<std::recursive_mutex> lock(fs::g_mutex);
    if (level) {
        fs::harvest();
        fs::g_census.clear();
        fs::g_names.clear();
        fs::g_name_index.clear();
        fs::g_kernels.clear();
        fs::g_tags.clear();
        fs::g_tag = -1;
    } else {
        fs::harvest();              // waits for the outstanding timed launches
    }
    fs::g_census_on = level < 0 ? 0 : (level > 2 ? 2 : level);
}

extern "C" int fs_census_read(fs_census_entry* out, int max_entries) {
    std::lock_guard<std::recursive_mutex> lock(fs::g_mutex);
    fs::harvest();
    int n = 0;
    for (auto& kv : fs::g_census) {
        if (out && n < max_entries) out[n] = kv.second;
        ++n;
    }
    return n;
}

extern "C" int fs_census_read_kernels(fs_kernel_time* out, int max_entries) {
    std::lock_guard<std::recursive_mutex> lock(fs::g_mutex);
    fs::harvest();
    const int n = (int)fs::g_names.size();
    for (int k = 0; k < n && out && k < max_entries; ++k) {
        memset(&out[k], 0, sizeof(out[k]));
        strncpy(out[k].name, fs::g_names[k].c_str(), sizeof(out[k].name) - 1);
        out[k].count = fs::g_kernels[k].count;
        out[k].ms = fs::g_kernels[k].ms;
        out[k].bytes = fs::g_kernels[k].bytes;
    }
    return n;
}

extern "C" void fs_census_tag(int tag) {
    std::lock_guard<std::recursive_mutex> lock(fs::g_mutex);
    fs::g_tag = tag;
}

extern "C" int fs_census_read_tags(int n_tags, long long* counts, double* ms) {
    std::lock_guard<std::recursive_mutex> lock(fs::g_mutex);
    fs::harvest();
    for (int t = 0; t < n_tags; ++t) {
        auto it = fs::g_tags.find(t);
        counts[t] = it == fs::g_tags.end() ? 0 : it->second.count;
        ms[t] = it == fs::g_tags.end() ? 0.0 : it->second.ms;
    }
    return (int)fs::g_tags.size();
}
