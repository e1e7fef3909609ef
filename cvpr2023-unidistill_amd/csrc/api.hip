// Library identity, error strings and the optional per-kernel HIP-event profiler.
#include "ud_common.h"
#include "ud_prof.h"
#include <map>
#include <mutex>
#include <string>
#include <vector>

#define UD_ABI_VERSION 1

extern "C" const char* ud_version(void) { return "unidistill_hip 1 gfx950"; }
extern "C" int ud_abi_version(void) { return UD_ABI_VERSION; }
extern "C" const char* ud_error_string(int code) {
  switch (code) {
    case UD_OK: return "ok";
    case UD_ERR_INVALID_ARG: return "invalid argument";
    case UD_ERR_WORKSPACE: return "workspace missing or too small";
    case UD_ERR_HIP: return "HIP runtime / launch failure";
    case UD_ERR_UNSUPPORTED: return "unsupported configuration";
    default: return "unknown error";
  }
}

// ---- profiler: hipEvent pairs recorded on the launch stream around named kernels -------------
namespace {
struct Pair {
  hipEvent_t a, b;
};
std::mutex g_mu;
bool g_on = false;
std::map<std::string, std::vector<Pair>> g_live;
std::vector<Pair> g_pool;

Pair get_pair() {
  if (!g_pool.empty()) {
    Pair p = g_pool.back();
    g_pool.pop_back();
    return p;
  }
  Pair p;
  (void)hipEventCreate(&p.a);
  (void)hipEventCreate(&p.b);
  return p;
}
}  // namespace

bool ud_prof_on() { return g_on; }

void* ud_prof_begin(const char* name, hipStream_t stream) {
  std::lock_guard<std::mutex> lk(g_mu);
  Pair p = get_pair();
  (void)hipEventRecord(p.a, stream);
  auto& v = g_live[name];
  v.push_back(p);
  return (void*)p.b;
}

void ud_prof_end(void* token, hipStream_t stream) { (void)hipEventRecord((hipEvent_t)token, stream); }

extern "C" void ud_prof_enable(int on) {
  std::lock_guard<std::mutex> lk(g_mu);
  g_on = on != 0;
}

extern "C" int ud_prof_read(const char* name, double* total_ms, int* calls, int reset) {
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_live.find(name);
  double tot = 0.0;
  int n = 0;
  if (it != g_live.end()) {
    for (Pair& p : it->second) {
      if (hipEventSynchronize(p.b) != hipSuccess) return UD_ERR_HIP;
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, p.a, p.b) != hipSuccess) return UD_ERR_HIP;
      tot += ms;
      ++n;
    }
    if (reset) {
      for (Pair& p : it->second) g_pool.push_back(p);
      it->second.clear();
    }
  }
  if (total_ms) *total_ms = tot;
  if (calls) *calls = n;
  return UD_OK;
}
