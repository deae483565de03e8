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
  s.tail_fuse = env_int("IVG_TAIL_FUSE", 1) != 0;
  s.shortcut_gemm256 = env_int("IVG_SHORTCUT_GEMM256", 1) != 0;
  s.conv_wide = env_int("IVG_CONV_WIDE", 0);
  if (s.conv_wide < 0 || s.conv_wide > 2) s.conv_wide = 0;
  s.conv_wide_grid = env_int("IVG_CONV_WIDE_GRID", 0);
  s.conv_wide_pf = env_int("IVG_CONV_WIDE_PF", 1) != 0;
  s.conv_wide_probe = env_int("IVG_CONV_WIDE_PROBE", 0);
  s.conv_wide_stagger = env_int("IVG_CONV_WIDE_STAGGER", -1);
  if (s.conv_wide_stagger > 64) s.conv_wide_stagger = -1;
  s.graph = env_int("IVG_GRAPH", 0) == 1;
  s.dg3_warm = env_int("IVG_DG3_WARM", 1) != 0;
  s.conv_cap = env_int("IVG_CONV_CAP", 0) == 1;
  s.decode_lds_kb = env_int("IVG_DECODE_LDS_KB", 160);
  s.decode_w_shared = env_int("IVG_DECODE_W_SHARED", 1) != 0;
  s.inflight_warm = env_int("IVG_INFLIGHT_WARM", 0) != 0;
  if (s.decode_lds_kb < 16 || s.decode_lds_kb > 160) s.decode_lds_kb = 160;
  return s;
}

// Every reload publishes a NEW immutable table through one atomic pointer and never frees or rewrites an old one (a launcher on
// another host thread may hold a reference for the length of a launch; ~60 bytes per ivg_create / ivg_reload_switches, deliberately
// leaked).  getenv() is only called here, under the mutex -- a Python thread changing os.environ while another thread reloads is the
// caller's race (ivideogpt_amd/switches.py: set / override are not to be called while batches are in flight).
static std::atomic<const Switches*> g_cur{nullptr};
static std::atomic<unsigned> g_gen{0};
static std::mutex g_mu;

void reload_switches() {
  std::lock_guard<std::mutex> lk(g_mu);
  g_cur.store(new Switches(read_env()), std::memory_order_release);
  g_gen.fetch_add(1, std::memory_order_release);
}

const Switches& sw() {
  const Switches* p = g_cur.load(std::memory_order_acquire);
  if (!p) { reload_switches(); p = g_cur.load(std::memory_order_acquire); }
  return *p;
}

unsigned switches_generation() { return g_gen.load(std::memory_order_acquire); }

}  // namespace ivg
