// libdiqt_hip.so: version + thread-local error string.
#include "common.h"
#include <string.h>
#include <atomic>

namespace diqt {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- launch census: tags are string literals, so the pointer identifies the launch site ----
namespace {
constexpr int kCensusSlots = 512;
struct CensusSlot { std::atomic<const char*> tag{nullptr}; std::atomic<long long> n{0}; };
CensusSlot g_census[kCensusSlots];
std::atomic<int> g_census_on{0};
thread_local const char* g_last_launch = "";
}  // namespace
void census_note(const char* what) {
    g_last_launch = what;
    if (!g_census_on.load(std::memory_order_relaxed)) return;
    unsigned h = (unsigned)((reinterpret_cast<uintptr_t>(what) >> 3) * 2654435761u) % kCensusSlots;
    for (int probe = 0; probe < kCensusSlots; ++probe, h = (h + 1) % kCensusSlots) {
        const char* cur = g_census[h].tag.load();
        if (cur == nullptr && g_census[h].tag.compare_exchange_strong(cur, what)) cur = what;
        if (cur == what) { g_census[h].n.fetch_add(1, std::memory_order_relaxed); return; }
    }
}
}  // namespace diqt

// diqt_census_enable(1): zero the counters and start counting launches per tag; (0): stop.  Returns the previous state.
extern "C" int diqt_census_enable(int on) {
    const int was = diqt::g_census_on.exchange(on ? 1 : 0);
    if (on) for (auto& s : diqt::g_census) s.n.store(0);
    return was;
}
// launches counted since diqt_census_enable(1) whose tag contains `substr` (the tags are the names check_launch reports, e.g.
// "conv3d_fwd_h(persistent)", "temporal_attention_h", "conv3d_fwd(v9)")
extern "C" long long diqt_census_count(const char* substr) {
    long long n = 0;
    for (auto& s : diqt::g_census) {
        const char* t = s.tag.load();
        if (t && (!substr || strstr(t, substr))) n += s.n.load();
    }
    return n;
}
// tag of the calling thread's most recent launch ("" before the first)
extern "C" const char* diqt_get_last_launch(void) { return diqt::g_last_launch; }

extern "C" int diqt_version(void) { return 100; }   // 0.1.0
extern "C" const char* diqt_last_error(void) { return diqt::g_err; }
