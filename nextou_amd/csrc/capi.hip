// Error plumbing and version query of libnextou_hip.so.
#include "common.h"
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

namespace nextou {

char* error_buffer() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

// ---- launch profiler ------------------------------------------------------------------------
namespace {
struct ProfRecord {
    hipEvent_t a, b;
    int bound;
    double work;
    char name[96];
};
std::mutex g_prof_mutex;
std::vector<ProfRecord> g_prof;
size_t g_prof_used = 0, g_prof_dropped = 0;
bool g_prof_on = false;
}  // namespace

ProfScope::ProfScope(hipStream_t s, int bound, double work, const char* fmt, ...) : slot(-1), stream(s) {
    if (!g_prof_on) return;
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    if (g_prof_used >= g_prof.size()) { ++g_prof_dropped; return; }  // pool exhausted: count it, never allocate here
    ProfRecord& r = g_prof[g_prof_used];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(r.name, sizeof(r.name), fmt, ap);
    va_end(ap);
    r.bound = bound;
    r.work = work;
    if (hipEventRecord(r.a, s) != hipSuccess) return;
    slot = (int)g_prof_used++;
}

ProfScope::~ProfScope() {
    if (slot < 0) return;
    (void)hipEventRecord(g_prof[slot].b, stream);
}

}  // namespace nextou

extern "C" int nextou_profile_enable(int max_records) {
    using namespace nextou;
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    g_prof_used = 0;
    g_prof_dropped = 0;
    g_prof_on = max_records > 0;
    while ((int)g_prof.size() < max_records) {
        ProfRecord r{};
        if (hipEventCreate(&r.a) != hipSuccess || hipEventCreate(&r.b) != hipSuccess)
            return fail(NEXTOU_EINVAL, "profile_enable: hipEventCreate failed after %zu records", g_prof.size());
        g_prof.push_back(r);
    }
    return 0;
}

extern "C" int nextou_profile_dropped(void) {
    using namespace nextou;
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    return (int)g_prof_dropped;
}

// JSON array, one object per distinct launch label, aggregated over the recorded launches.
// The caller must have synchronised the device.  Returns the number of bytes written (0 if the
// buffer is too small).
extern "C" size_t nextou_profile_report(char* buf, size_t cap) {
    using namespace nextou;
    std::lock_guard<std::mutex> lock(g_prof_mutex);
    struct Agg { std::string name; int bound; double work, ms; int launches; };
    std::vector<Agg> aggs;
    for (size_t i = 0; i < g_prof_used; ++i) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, g_prof[i].a, g_prof[i].b) != hipSuccess) continue;
        Agg* hit = nullptr;
        for (auto& a : aggs)
            if (a.name == g_prof[i].name) { hit = &a; break; }
        if (!hit) { aggs.push_back({g_prof[i].name, g_prof[i].bound, 0.0, 0.0, 0}); hit = &aggs.back(); }
        hit->work += g_prof[i].work;
        hit->ms += ms;
        hit->launches += 1;
    }
    std::string out = "[";
    for (size_t i = 0; i < aggs.size(); ++i) {
        char line[256];
        snprintf(line, sizeof(line), "%s{\"kernel\": \"%s\", \"bound\": \"%s\", \"launches\": %d, \"ms\": %.6f, \"work\": %.6e}",
                 i ? ", " : "", aggs[i].name.c_str(), aggs[i].bound == kBoundMfma ? "mfma" : "hbm",
                 aggs[i].launches, aggs[i].ms, aggs[i].work);
        out += line;
    }
    out += "]";
    if (out.size() + 1 > cap) return 0;
    memcpy(buf, out.c_str(), out.size() + 1);
    return out.size();
}

extern "C" int nextou_abi_version(void) { return NEXTOU_ABI_VERSION; }
extern "C" const char* nextou_last_error(void) { return nextou::error_buffer(); }
