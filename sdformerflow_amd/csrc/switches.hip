// The switch table of switches.h: one getenv() per switch at the first use (or at sdf_switches_reload()), none afterwards.
#include <atomic>
#include <mutex>
#include <stdlib.h>
#include <string.h>

#include "common.h"
#include "switches.h"

namespace {
struct Table {
  bool set[SW_COUNT];
  char val[SW_COUNT][24];
};
Table g_tab[2];                                   // double-buffered: a reload fills the other copy, then publishes it
std::atomic<int> g_cur{-1};                       // -1: not built yet
std::mutex g_mu;
const char* const kNames[SW_COUNT] = {
#define X(n) "SDF_" #n,
    SDF_SWITCH_LIST(X)
#undef X
};
void fill(Table& t) {
  for (int i = 0; i < SW_COUNT; ++i) {
    const char* e = getenv(kNames[i]);
    t.set[i] = e != nullptr;
    t.val[i][0] = 0;
    if (e) { strncpy(t.val[i], e, sizeof(t.val[i]) - 1); t.val[i][sizeof(t.val[i]) - 1] = 0; }
  }
}
}  // namespace

const char* sdf_sw(SdfSwitch s) {
  int c = g_cur.load(std::memory_order_acquire);
  if (c < 0) {
    std::lock_guard<std::mutex> lk(g_mu);
    c = g_cur.load(std::memory_order_relaxed);
    if (c < 0) { fill(g_tab[0]); c = 0; g_cur.store(0, std::memory_order_release); }
  }
  return g_tab[c].set[s] ? g_tab[c].val[s] : nullptr;
}

extern "C" void sdf_switches_reload(void) {
  std::lock_guard<std::mutex> lk(g_mu);
  const int c = g_cur.load(std::memory_order_relaxed), n = c == 0 ? 1 : 0;
  fill(g_tab[n]);
  g_cur.store(n, std::memory_order_release);
}
