// The launch log of launch_log.h.
#include <cxxabi.h>
#include <mutex>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "common.h"

std::atomic<int> g_sdf_launch_log_on{0};

namespace {
struct Rec {
  const void* fn;
  hipStream_t stream;
  uint32_t wgs, threads, lds;
  hipEvent_t e0, e1;
};
std::mutex g_mu;
std::vector<Rec> g_recs;
std::vector<hipEvent_t> g_pool;                      // events are kept and reused across logs
size_t g_pool_used = 0;

hipEvent_t take_event() {
  if (g_pool_used == g_pool.size()) {
    hipEvent_t e = nullptr;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    g_pool.push_back(e);
  }
  return g_pool[g_pool_used++];
}
}  // namespace

void sdf_launch_log_begin(const void* fn, dim3 grid, dim3 block, size_t lds, hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_mu);
  Rec r;
  r.fn = fn; r.stream = s;
  r.wgs = grid.x * grid.y * grid.z; r.threads = block.x * block.y * block.z; r.lds = (uint32_t)lds;
  r.e0 = take_event(); r.e1 = take_event();
  if (r.e0) (void)hipEventRecord(r.e0, s);
  g_recs.push_back(r);
}

void sdf_launch_log_end(hipStream_t s) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!g_recs.empty() && g_recs.back().e1) (void)hipEventRecord(g_recs.back().e1, s);
}

extern "C" void sdf_launch_log(int enable) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (enable) { g_recs.clear(); g_pool_used = 0; }
  g_sdf_launch_log_on.store(enable ? 1 : 0, std::memory_order_relaxed);
}

extern "C" int sdf_launch_log_read(SdfLaunchRecord* out, int max_records) {
  std::lock_guard<std::mutex> lk(g_mu);
  if (!out && max_records > 0) return SDF_E_NULL;
  const int n = (int)g_recs.size();
  for (int i = 0; i < n && i < max_records; ++i) {
    const Rec& r = g_recs[i];
    SdfLaunchRecord& o = out[i];
    memset(&o, 0, sizeof(o));
    o.workgroups = r.wgs; o.threads = r.threads; o.lds_bytes = r.lds; o.us = -1.f;
    if (r.e0 && r.e1 && hipEventSynchronize(r.e1) == hipSuccess) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) o.us = ms * 1e3f;
    }
    const char* mangled = hipKernelNameRefByPtr(r.fn, r.stream);
    if (mangled) {
      int st = 0;
      char* dem = abi::__cxa_demangle(mangled, nullptr, nullptr, &st);
      const char* name = (st == 0 && dem) ? dem : mangled;
      strncpy(o.kernel, name, sizeof(o.kernel) - 1);
      free(dem);
    }
  }
  return n;
}
