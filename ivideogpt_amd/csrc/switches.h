// Every run-time switch of libivg, in ONE place.  The table is read from the environment when the library is loaded, again at
// every ivg_create and on ivg_reload_switches() (tests flip a variable, then call it); the kernels' launchers only ever read the
// published table -- no getenv() and no lazily initialised static on a launch path, so engines driven from several host threads
// (bench.py --lanes, replica()) share nothing that is written after start-up.
//
//   variable                 default  meaning
//   ---- which kernel runs a shape (A/B runs and the tests of the alternative paths)
//   IVG_CONV3X3              1        0: every 3x3 convolution on the generic implicit GEMM (igemm.hip)
//   IVG_GEMM256              1        0: large dense GEMMs on the generic implicit GEMM
//   IVG_DG3                  1        0: decode GEMMs on the second-generation kernel (dgemm.hip)
//   IVG_FLASH_PREFILL        1        0: prompt attention as score GEMM + softmax + P.V GEMM (what the fp32 engine mode runs)
//   IVG_FLASH_XATT           1        0: tokenizer attention as score GEMM + softmax + P.V GEMM
//   IVG_GN_FUSE              1        0: every GroupNorm computes its own statistics (1: reduced by the producing conv3x3's epilogue)
//   IVG_GN_APPLY_FUSE        1        0: GroupNorm + SiLU as a separate apply pass (1: inside the consuming conv3x3's halo staging)
//   IVG_CONV_WIDE            0        0: bf16 3x3 convolutions on conv3x3.hip's 256-pixel kernel only; 1: the persistent two-tile kernel of
//                                        conv3x3w.hip for the shapes it is faster on IN ISOLATION (one N tile, upsampling); 2: wherever it covers the
//                                        shape.  Off by default: inside the decode stage (residuals, output statistics) it is 1 % behind. Development:
//                                        IVG_CONV_WIDE_GRID its grid size, IVG_CONV_WIDE_PF=0 no fragment prefetch across the step barrier,
//                                        IVG_CONV_WIDE_STAGGER start phases (-1 = by items per workgroup), IVG_CONV_WIDE_PROBE timing probes with
//                                        WRONG results (1: no epilogue, 2: no input normalisation)
//   IVG_TAIL_FUSE            1        0: the decoders' tail as GroupNorm apply pass + implicit-GEMM conv_out (1: one conv3x3 launch with the
//                                        normalisation inside its staging; bf16 decode path)
//   IVG_SHORTCUT_GEMM256     1        0: 1x1 convolutions always on the implicit GEMM (1: on gemm256l where Cout % 256 == 0)
//   IVG_X3                   1        0: the fp32 decode path of the tokenizer on f32-input MFMAs (1: split-bf16 "x3" convolutions)
//   ---- launch policy
//   IVG_GRAPH                0        1: decode steps replayed from hipGraphs (8 steps per launch) instead of eager launches
//   IVG_DG3_WARM             1        0: decode GEMMs do not pull the next launch's weights toward the chip
//   IVG_CONV_CAP             0        1: conv3x3 grids at ONE workgroup per CU (LDS padded past half a CU's 160 KiB): leaves half of
//                                        every CU's LDS, wave slots and registers to the kernels of another batch in flight
//   IVG_DECODE_W_SHARED      1        engines with a batches-in-flight budget (decode_lds_kb > 0): 0 = non-temporal weight requests as for one batch alone
//   IVG_INFLIGHT_WARM        0        the same engines: 1 = keep warming the next launch's weights
//   IVG_DECODE_LDS_KB        160      LDS budget of a decode-GEMM workgroup in KiB (<= 78: it fits beside a capped conv3x3 workgroup)
#pragma once

namespace ivg {

struct Switches {
  int conv3x3 = 1, gemm256 = 1, dg3 = 1, flash_prefill = 1, flash_xatt = 1, gn_fuse = 1, gn_apply_fuse = 1, x3 = 1, tail_fuse = 1, shortcut_gemm256 = 1, conv_wide = 0, conv_wide_grid = 0, conv_wide_pf = 1, conv_wide_probe = 0, conv_wide_stagger = -1;
  int graph = 0, dg3_warm = 1, conv_cap = 0, decode_lds_kb = 160, decode_w_shared = 1, inflight_warm = 0;
};

const Switches& sw();             // the published table: immutable, never freed or rewritten (every reload publishes a NEW one)
void reload_switches();           // re-read the environment and publish a new table
unsigned switches_generation();   // incremented by every reload (part of the key of captured step graphs)

}  // namespace ivg
