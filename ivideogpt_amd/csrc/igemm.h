// Implicit-GEMM convolution / batched GEMM on MFMA (gfx950).  See igemm.hip.
#pragma once
#include "common.h"

namespace ivg {

enum IgemmFlags : int {
  IG_BIAS_N = 1,     // + bias[n]           (fp32)
  IG_BIAS_M = 2,     // + bias[m]           (fp32; used by the transposed V projection)
  IG_RESIDUAL = 4,   // + R[m][n]           (element type T, same addressing as the output)
  IG_SILU = 8,       // silu(.) after bias/residual
  IG_GLU = 16,       // weight rows packed [16 gate | 16 up] per 32: out[m][j] = silu(gate) * up, N_out = N/2
  IG_OUT_F32 = 32,   // output is fp32 regardless of T (attention scores, logits, final pixels)
  SK_NORM = 64,      // skinny GEMM only: scale row m by rsqrt(mean_k X[m][k]^2 + eps) (RMSNorm with the weight folded into W)
  IG_CLAMP01 = 128,  // clamp(., 0, 1) last (the final conv of the decoders when the caller wants displayable frames: predict.py:73)
};

// Y[z](m, n) = epi( alpha * sum_k A[z](m, k) * W[z][n][k] )
//   A(m, k): gathered from an NHWC activation tensor  X[img][ih][iw][c]  (pixel stride ldx):
//            m -> (img, oh, ow);  k -> (kh, kw, c);  ih = oh*stride + kh - pad  (ups: ih = (oh + kh - 1) >> 1)
//   W[n][k]: weights, K contiguous (row stride ldw), K ordered (kh, kw, c)
//   Y addressing: base + (img / c_grp) * c_grp_stride + (img % c_grp) * c_img + pix * c_pix + n * c_ch
// A plain row-major GEMM is the 1x1 case with Hin = Hout = 1, Win = Wout = M.
struct IgemmArgs {
  const void* X = nullptr;
  const void* W = nullptr;
  void* Y = nullptr;
  const void* R = nullptr;
  const float* bias = nullptr;
  int Nimg = 1, Hin = 1, Win = 1, Cin = 0, ldx = 0;
  int Hout = 1, Wout = 1;
  int KH = 1, KW = 1, stride = 1, pad = 0, ups = 0;
  int N = 0, ldw = 0;
  long c_img = 0, c_pix = 0, c_ch = 1;
  int c_grp = 1;
  long c_grp_stride = 0;
  int flags = 0;
  float alpha = 1.0f;
  // GroupNorm statistics of the OUTPUT produced in the epilogue (conv3x3.hip only; null = off): per-(image, chunk, group)
  // (sum, sum of squares) of the stored values, double2 [Nimg][gn_chunks][gn_groups] in the layout norm.hip's gn_apply reads.
  // The launcher sets gn_chunks (out); the buffer must hold at least Nimg * gn_chunks_bound(...) * gn_groups entries.
  void* gn_part = nullptr;
  int gn_groups = 0;
  mutable int gn_chunks = 0;
  // GroupNorm + SiLU of the INPUT fused into the conv3x3 staging (null = off): float2 (scale, shift) [Nimg][Cin] with
  // x_normalised = silu(x * scale + shift)  (norm.hip: launch_gn_coef); the conv then reads the RAW tensor
  const void* gn_in_coef = nullptr;
  // conv3x3.hip, fp32 tensors only: the weights pre-split into bf16 (hi, lo) pairs in the kernel's slot layout (packing.py pack_x3;
  // same bytes per row as W) -- selects the split-bf16 ("x3") arithmetic of the 1e-3-compliant decode mode.  null: off
  const void* W_x3 = nullptr;
  // conv3x3.hip, ups = 1 only: the upsampling convolution's weights pre-summed per output-pixel parity for the SUB-PIXEL form
  // (packing.py pack_subpixel: [4 phases][N][4 taps x Cin], element type of X; W_sub_x3: the same pre-split for the x3 arithmetic).
  // null: nine taps over the upsampled grid
  const void* W_sub = nullptr;
  const void* W_sub_x3 = nullptr;
  // igemm.hip, fp32 tensors only: split-bf16 arithmetic with both operands split in registers (no pre-split weights needed)
  bool x3 = false;
  // batch z = (z0 * nb1 + z1) * nb2 + z2 ; element strides per operand
  int nb0 = 1, nb1 = 1, nb2 = 1;
  long sa[3] = {0, 0, 0}, sw[3] = {0, 0, 0}, sy[3] = {0, 0, 0};
};

// dtype = element type of X / W / R (and of Y unless IG_OUT_F32).  Returns hipError_t as int.
int launch_igemm(const IgemmArgs& a, DType dtype, hipStream_t stream);
// 256 x 256-tile plain GEMM for tens of thousands of rows, bf16 (gemm256.hip); -1 when the shape is not covered
int launch_gemm256(const IgemmArgs& a, DType dtype, hipStream_t stream);
// LDS-halo 3x3 stride-1 kernel (conv3x3.hip); returns -1 when the shape is not covered (use launch_igemm then)
int launch_conv3x3(const IgemmArgs& a, DType dtype, hipStream_t stream);
long long gemm256x3_launches();         // gemm256.hip: launches of the 256 x 256-tile split-bf16 GEMM since load (test hook)
long long conv3x3_subpixel_launches();   // conv3x3.hip: upsampling convolutions launched in sub-pixel form since load (test hook)
long long decode_gemm_launches(int generation);   // dgemm.hip: decode GEMMs the dispatcher sent to generation 3 / 2 since load (test hook)
// upper bound of the GroupNorm statistics chunks a conv3x3 launch with this output geometry writes per image
int conv3x3_gn_chunks_bound(int Hout, int Wout, int N);

// Skinny GEMM for the autoregressive decode steps (M <= 128 rows, weights streamed once):
//   Y[m][n] = epi( sum_k X[m][k] * W[n][k] ),  X row stride ldx, W row stride ldw.
struct SkinnyArgs {
  const void* X = nullptr;
  const void* W = nullptr;
  void* Y = nullptr;       // T, or fp32 when IG_OUT_F32
  int M = 0, N = 0, K = 0, ldx = 0, ldw = 0, ldy = 0;
  int flags = 0;           // IG_GLU | IG_OUT_F32 | IG_RESIDUAL (Y += ..., in place) | SK_NORM
  float eps = 1e-6f;       // SK_NORM
  int* bump = nullptr;     // optional pair of device ints incremented once at the end (StepState advance)
  // measurement hook (bench.py): launch window on the 100 MHz wall clock, [IVG_GEMM_PROF_SLOTS][starts prof_ld | ends prof_ld],
  // indexed by the decode position *pos (the launch runs inside a replayed graph: HIP events cannot bracket it)
  unsigned long long* prof = nullptr;
  const int* pos = nullptr;
  int prof_ld = 0;
  bool x3 = false;            // fp32 tensors: split-bf16 arithmetic (dgemm3.hip only; the other generations ignore it)
  bool w_shared = false;      // the weight matrix is read by OTHER engines' launches too within the same few microseconds (replicas over one
                              // copy of the weights, batches in flight): default cache policy instead of non-temporal requests
  int lds_kb = 0;             // LDS budget of a workgroup in KiB (the calling engine's policy); 0: the process default (IVG_DECODE_LDS_KB)
  int kind = -1;              // which GEMM of the decode step this is (0 q/k/v, 1 o-proj, 2 gate/up, 3 down, 4 lm_head; -1: unknown) -- selects the
                              // per-kind budget of the batches-in-flight profile (switches.h: inflight_kb)
  long long* dbg = nullptr;   // development: phase stamps of the second / third-generation kernel (tools/ubench/dgemm_phase.hip)
  // cache warm-up (dgemm3.hip): the weight matrix the NEXT launch of the chain streams, as next_tiles contiguous tiles of
  // next_tile_bytes (= rows one workgroup of that launch owns x K bytes); tile t is pulled by a workgroup of XCD t % 8
  const void* next_W = nullptr;
  long next_tile_bytes = 0;
  int next_tiles = 0;
};
#define IVG_GEMM_PROF_SLOTS 8
int launch_skinny(const SkinnyArgs& a, DType dtype, hipStream_t stream);   // dispatcher: dgemm3.hip, else dgemm.hip (error: neither covers the shape)
// dgemm.hip: second-generation kernel, activations staged as whole cache lines; -1 when the shape is not covered
int launch_dgemm(const SkinnyArgs& a, DType dtype, hipStream_t stream);
// dgemm3.hip: third generation (K over up to 16 waves, one barrier, cache warm-up of the next launch's weights); -1 when not covered
int launch_dgemm3(const SkinnyArgs& a, DType dtype, hipStream_t stream);
int dgemm3_w_rows_per_block(const SkinnyArgs& a, DType dtype);   // rows of W per workgroup of that plan (0: not covered)
int dgemm_w_rows_per_block(const SkinnyArgs& a, DType dtype);    // the same for the second-generation kernel (0: not covered)

}  // namespace ivg
