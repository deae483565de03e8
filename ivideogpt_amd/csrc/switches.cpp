// The switch table of libivg (switches.h): read from the environment, published through one atomic pointer.
#include "switches.h"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <mutex>

namespace ivg {

static int env_int(const char* name, int dflt) {
  const char* v = getenv(name);
  return (v && v[0]) ? atoi(v) : dflt;
}

static Switches read_env() {
  Switches s;
  const bool dev = env_int("IVG_DEV", 0) == 1;   // the A/B switches are development tools: inert without it
  auto dv = [&](const char* name, int dflt) { return dev ? env_int(name, dflt) : dflt; };
  s.conv3x3 = dv("IVG_CONV3X3", 1) != 0;
  s.subpixel = dv("IVG_SUBPIXEL", 1) != 0;
  s.gemm256 = dv("IVG_GEMM256", 1) != 0;
  s.gemm256x3 = dv("IVG_GEMM256X3", 1) != 0;
  s.kv24 = dv("IVG_KV24", 1) != 0;
  s.dg3 = dv("IVG_DG3", 1) != 0;
  s.flash_prefill = dv("IVG_FLASH_PREFILL", 1) != 0;
  s.flash_xatt = dv("IVG_FLASH_XATT", 1) != 0;
  s.gn_fuse = dv("IVG_GN_FUSE", 1) != 0;
  s.gn_apply_fuse = dv("IVG_GN_APPLY_FUSE", 1) != 0;
  s.x3 = dv("IVG_X3", 1) != 0;
  s.dg3_warm = dv("IVG_DG3_WARM", 1) != 0;
  s.warm_gate_up = dv("IVG_WARM_GATE_UP", 0) != 0;
  s.conv_cap = dv("IVG_CONV_CAP", 0) == 1;
  s.decode_w_shared = dv("IVG_DECODE_W_SHARED", 1) != 0;
  s.inflight_warm = dv("IVG_INFLIGHT_WARM", 0) != 0;
  s.dg2_mf_cap = dv("IVG_DG2_MF_CAP", 0);
  if (s.dg2_mf_cap != 1 && s.dg2_mf_cap != 2) s.dg2_mf_cap = 0;
  s.inflight_gemm256 = dv("IVG_INFLIGHT_GEMM256", 1) != 0;
  if (const char* v = dev ? getenv("IVG_INFLIGHT_KB") : nullptr) {
    int k[5] = {0, 0, 0, 0, 0};
    if (sscanf(v, "%d,%d,%d,%d,%d", &k[0], &k[1], &k[2], &k[3], &k[4]) == 5)
      for (int i = 0; i < 5; ++i) s.inflight_kb[i] = (k[i] >= 16 && k[i] <= 160) ? k[i] : 0;
  }
  s.graph = env_int("IVG_GRAPH", 0) == 1;
  s.decode_lds_kb = env_int("IVG_DECODE_LDS_KB", 160);
  if (s.decode_lds_kb < 16 || s.decode_lds_kb > 160) s.decode_lds_kb = 160;
  return s;
}

bool Switches::operator==(const Switches& o) const {
  return conv3x3 == o.conv3x3 && subpixel == o.subpixel && gemm256 == o.gemm256 && dg3 == o.dg3 && flash_prefill == o.flash_prefill &&
         flash_xatt == o.flash_xatt && gn_fuse == o.gn_fuse && gn_apply_fuse == o.gn_apply_fuse && x3 == o.x3 && gemm256x3 == o.gemm256x3 && kv24 == o.kv24 && graph == o.graph &&
         dg3_warm == o.dg3_warm && warm_gate_up == o.warm_gate_up && conv_cap == o.conv_cap && decode_lds_kb == o.decode_lds_kb && decode_w_shared == o.decode_w_shared &&
         inflight_warm == o.inflight_warm && dg2_mf_cap == o.dg2_mf_cap && inflight_gemm256 == o.inflight_gemm256 &&
         std::equal(inflight_kb, inflight_kb + 5, o.inflight_kb);
}

// A reload that CHANGES the table publishes a new immutable one through one atomic pointer and never frees or rewrites the old
// (a launcher on another host thread may hold a reference for the length of a launch; ~64 bytes per change, deliberately leaked).
// A reload that reads the same values publishes nothing and leaves the generation alone: ivg_create re-reads the environment, and
// with the generation in the key of captured step graphs every engine construction (replicas, capacity growth) would otherwise
// orphan the graphs of all live engines (advice, round 5).  getenv() is only called here, under the mutex -- a Python thread
// changing os.environ while another thread reloads is the caller's race (ivideogpt_amd/switches.py).
static std::atomic<const Switches*> g_cur{nullptr};
static std::atomic<unsigned> g_gen{0};
static std::mutex g_mu;

void reload_switches() {
  std::lock_guard<std::mutex> lk(g_mu);
  const Switches now = read_env();
  const Switches* cur = g_cur.load(std::memory_order_acquire);
  if (cur && *cur == now) return;
  g_cur.store(new Switches(now), std::memory_order_release);
  g_gen.fetch_add(1, std::memory_order_release);
}

const Switches& sw() {
  const Switches* p = g_cur.load(std::memory_order_acquire);
  if (!p) { reload_switches(); p = g_cur.load(std::memory_order_acquire); }
  return *p;
}

unsigned switches_generation() { return g_gen.load(std::memory_order_acquire); }

}  // namespace ivg
