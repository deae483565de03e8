// Every run-time switch of libivg, in ONE place.  The table is read from the environment when the library is loaded, again at
// every ivg_create and on ivg_reload_switches() (tests flip a variable, then call it); the kernels' launchers only ever read the
// published table -- no getenv() and no lazily initialised static on a launch path, so engines driven from several host threads
// (bench.py --lanes, replica()) share nothing that is written after start-up.
//
// Two classes (round 6: the settled A/B switches left the product surface):
//   * LAUNCH POLICY -- honoured always: how a deployment wants the same kernels launched.
//   * DEVELOPMENT -- honoured only when IVG_DEV=1 is set in the environment as well: which of two kernels runs a shape.  They exist
//     so that a test can put the alternative path (the one the fp32 parity mode or an uncovered shape takes anyway) beside the default
//     on the same inputs; without IVG_DEV=1 a stray variable changes nothing.  tests/conftest.py's `switches` fixture sets it.
//
//   variable                 default  meaning
//   ---- launch policy
//   IVG_GRAPH                0        1: decode steps replayed from hipGraphs (8 steps per launch) instead of eager launches
//   IVG_DECODE_LDS_KB        160      process-wide default of ivg_config.decode_lds_kb: LDS budget of a decode-GEMM workgroup in KiB; a
//                                        budget below 160 is the batches-in-flight profile (see include/ivg.h)
//   ---- development (IVG_DEV=1)
//   IVG_CONV3X3              1        0: every 3x3 convolution on the generic implicit GEMM (igemm.hip)
//   IVG_SUBPIXEL             1        0: nearest-x2 upsampling convolutions as nine taps over the upsampled grid (1: four 2x2 phase
//                                        convolutions over the low-resolution input with pre-summed weights, conv3x3.hip SUBPIX)
//   IVG_GEMM256              1        0: large dense GEMMs on the generic implicit GEMM
//   IVG_KV24                 1        0: the x3 rollout keeps an fp32 K / V cache (1: 24 bits per element in two planes, llama_ops.hip)
//   IVG_GEMM256X3            1        0: large dense split-bf16 ("x3") GEMMs on the generic implicit GEMM's X3 instance (1: 256 x 256 tiles)
//   IVG_DG3                  1        0: decode GEMMs on the second-generation kernel (dgemm.hip)
//   IVG_FLASH_PREFILL        1        0: prompt attention as score GEMM + softmax + P.V GEMM (what the fp32 engine mode runs)
//   IVG_FLASH_XATT           1        0: tokenizer attention as score GEMM + softmax + P.V GEMM
//   IVG_GN_FUSE              1        0: every GroupNorm computes its own statistics (1: reduced by the producing conv3x3's epilogue)
//   IVG_GN_APPLY_FUSE        1        0: GroupNorm + SiLU as a separate apply pass (1: inside the consuming conv3x3's halo staging)
//   IVG_X3                   1        0: the fp32 decode path of the tokenizer on f32-input MFMAs (1: split-bf16 "x3" convolutions)
//   IVG_DG3_WARM             1        0: decode GEMMs do not pull the next launch's weights toward the chip
//   IVG_WARM_GATE_UP         0        1: o-proj also warms the gate/up matrix (round 5's behaviour; measured neutral in time, 1.6 x the traffic)
//   IVG_CONV_CAP             0        1: conv3x3 grids at ONE workgroup per CU (LDS padded past half a CU's 160 KiB)
//   IVG_DECODE_W_SHARED      1        engines with the batches-in-flight profile: 0 = non-temporal weight requests as for one batch alone
//   IVG_INFLIGHT_WARM        0        the same engines: 1 = keep warming the next launch's weights
//   IVG_DG2_MF_CAP           0        the same engines: > 0 caps the row tiles per workgroup of the second-generation decode GEMMs (smaller
//                                        workgroups, more of them co-resident per CU; row grouping only -- never a summation order)
//   IVG_INFLIGHT_KB          -        "q,o,g,d,l": per-kind LDS budgets (KiB, 0 = the engine's) of the same engines' decode GEMMs (q/k/v, o-proj,
//                                        gate/up, down, lm_head): which GEMMs may keep their whole K range in flight (one memory round trip)
//   IVG_INFLIGHT_GEMM256     1        the same engines: 0 = prompt-pass GEMMs on the 256-thread implicit-GEMM kernel instead of the 1024-thread /
//                                        128 KiB gemm256l workgroups (which wait for a whole CU to drain while other batches are in flight)
#pragma once

namespace ivg {

struct Switches {
  int conv3x3 = 1, subpixel = 1, gemm256 = 1, dg3 = 1, flash_prefill = 1, flash_xatt = 1, gn_fuse = 1, gn_apply_fuse = 1, x3 = 1, gemm256x3 = 1, kv24 = 1;
  int graph = 0, dg3_warm = 1, warm_gate_up = 0, conv_cap = 0, decode_lds_kb = 160, decode_w_shared = 1, inflight_warm = 0, dg2_mf_cap = 0, inflight_gemm256 = 1;
  int inflight_kb[5] = {0, 0, 0, 0, 0};
  bool operator==(const Switches& o) const;
};

const Switches& sw();             // the published table: immutable, never freed or rewritten (a reload that changes it publishes a NEW one)
void reload_switches();           // re-read the environment; publishes a new table (and a new generation) only when a field changed
unsigned switches_generation();   // incremented by every reload that changed the table (part of the key of captured step graphs)

}  // namespace ivg
