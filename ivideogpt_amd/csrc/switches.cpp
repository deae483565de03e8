// The switch table of libivg (switches.h): read from the environment, published through one atomic pointer.
#include "switches.h"

#include <atomic>
#include <cstdlib>
#include <mutex>

namespace ivg {

static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && v[0]) ? atoi(v) : dflt;
}

static Switches read_env() {
  Switches s;
  s.conv3x3 = env_int("IVG_CONV3X3", 1) != 0;
  s.gemm256 = env_int("IVG_GEMM256", 1) != 0;
  s.dg3 = env_int("IVG_DG3", 1) != 0;
  s.flash_prefill = env_int("IVG_FLASH_PREFILL", 1) != 0;
  s.flash_xatt = env_int("IVG_FLASH_XATT", 1) != 0;
  s.gn_fuse = env_int("IVG_GN_FUSE", 1) != 0;
  s.gn_apply_fuse = env_int("IVG_GN_APPLY_FUSE", 1) != 0;
  s.x3 = env_int("IVG_X3", 1) != 0;
  s.conv_wide = env_int("IVG_CONV_WIDE", 1) != 0;
  s.conv_wide_grid = env_int("IVG_CONV_WIDE_GRID", 0);
  s.graph = env_int("IVG_GRAPH", 0) == 1;
  s.dg3_warm = env_int("IVG_DG3_WARM", 1) != 0;
  s.conv_cap = env_int("IVG_CONV_CAP", 0) == 1;
  s.decode_lds_kb = env_int("IVG_DECODE_LDS_KB", 160);
  if (s.decode_lds_kb < 16 || s.decode_lds_kb > 160) s.decode_lds_kb = 160;
  return s;
}

// A reload publishes a NEW table and leaves the old one alive (a launcher on another host thread may still be reading it): a few
// dozen bytes per ivg_create, bounded by a small ring that is only recycled after 64 further reloads.
static std::atomic<const Switches*> g_cur{nullptr};
static std::mutex g_mu;
static Switches g_ring[64];
static unsigned g_next = 0;

void reload_switches() {
  std::lock_guard<std::mutex> lk(g_mu);
  Switches& slot = g_ring[g_next++ & 63];
  slot = read_env();
  g_cur.store(&slot, std::memory_order_release);
}

const Switches& sw() {
  const Switches* p = g_cur.load(std::memory_order_acquire);
  if (!p) { reload_switches(); p = g_cur.load(std::memory_order_acquire); }
  return *p;
}

}  // namespace ivg
