// runtime.hip -- ABI version, per-thread error message, and the optional per-kernel timer of
// libtdk_hip.so.
#include <string.h>

#include <map>
#include <mutex>
#include <set>
#include <string>
#include <utility>
#include <vector>

#include "tdk_common.h"

static thread_local char g_last_error[512] = "";

int tdk_device_cus() {
  static int cus = 0;  // one device type per process; a stale value only changes launch shapes, never results
  if (cus == 0) {
    int dev = 0, n = 256;
    if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
    cus = n > 0 ? n : 256;
  }
  return cus;
}

void tdk_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
  va_end(ap);
}

int tdk_raise_lds_limit(const void* func, int bytes, const char* what) {
  static std::mutex mu;
  static std::set<std::pair<int, const void*>> done;  // (device, kernel function)
  int dev = 0;
  TDK_HIP_CALL(hipGetDevice(&dev), what);
  std::lock_guard<std::mutex> lk(mu);
  if (done.count({dev, func})) return TDK_OK;
  TDK_HIP_CALL(hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes), what);
  done.insert({dev, func});
  return TDK_OK;
}

TDK_EXPORT int tdk_abi_version(void) { return TDK_ABI_VERSION; }
TDK_EXPORT const char* tdk_last_error(void) { return g_last_error; }

// ---------------------------------------------------------------- per-kernel event timer
// When enabled, every TDK_LAUNCH (or, with tdk_profile_filter, every launch whose name contains the
// filter string) is bracketed by two hipEvents recorded on the launch stream.
// tdk_profile_report() synchronises those events and returns, per kernel name, the launch
// count and the summed device time.  Off by default: the launch path then costs one branch.
bool g_tdk_profile_on = false;

namespace {
struct Rec {
  const char* name;
  hipEvent_t a, b;
};
std::mutex g_mu;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;
std::string g_filter;

hipEvent_t get_event() {
  if (!g_pool.empty()) {
    hipEvent_t e = g_pool.back();
    g_pool.pop_back();
    return e;
  }
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return nullptr;
  return e;
}
}  // namespace

bool tdk_timer_begin(const char* name, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_filter.empty() && strstr(name, g_filter.c_str()) == nullptr) return false;
  Rec r{name, get_event(), get_event()};
  if (r.a) (void)hipEventRecord(r.a, s);
  g_recs.push_back(r);
  return true;
}

void tdk_timer_end(hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_recs.empty() && g_recs.back().b) (void)hipEventRecord(g_recs.back().b, s);
}

TDK_EXPORT int tdk_profile_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  for (auto& r : g_recs) {
    if (r.a) g_pool.push_back(r.a);
    if (r.b) g_pool.push_back(r.b);
  }
  g_recs.clear();
  g_tdk_profile_on = on != 0;
  return TDK_OK;
}

// Restrict the timer to launches whose name contains `substr` (NULL or "" = every launch), so that
// timing one kernel inside a throughput measurement does not put events between all the others.
TDK_EXPORT int tdk_profile_filter(const char* substr) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_filter = substr ? substr : "";
  return TDK_OK;
}

// Writes lines "name count total_ms\n" into buf (NUL-terminated, truncated to cap) and returns
// the number of bytes needed.  Blocks until the recorded events have completed.
TDK_EXPORT int64_t tdk_profile_report(char* buf, int64_t cap) {
  std::lock_guard<std::mutex> lk(g_mu);
  std::map<std::string, std::pair<long, double>> agg;
  for (auto& r : g_recs) {
    if (!r.a || !r.b) continue;
    if (hipEventSynchronize(r.b) != hipSuccess) continue;
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, r.a, r.b) != hipSuccess) continue;
    auto& e = agg[r.name];
    e.first += 1;
    e.second += ms;
  }
  std::string out;
  char line[256];
  for (auto& kv : agg) {
    snprintf(line, sizeof(line), "%s %ld %.6f\n", kv.first.c_str(), kv.second.first, kv.second.second);
    out += line;
  }
  if (buf && cap > 0) {
    const size_t n = out.size() < (size_t)(cap - 1) ? out.size() : (size_t)(cap - 1);
    memcpy(buf, out.data(), n);
    buf[n] = 0;
  }
  return (int64_t)out.size() + 1;
}
