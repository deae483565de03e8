// Large plain GEMM for gfx950 (bf16):  Y[m][n] = epi( sum_k X[m][k] * W[n][k] ),  M in the tens of thousands.
// The prompt pass of the transformer (SURVEY.md 2.4 K14 / K17 at L = 514: 32,896 rows) and any dense 1x1 layer large enough.
//
// Why a third MFMA kernel: the generic implicit GEMM (igemm.hip, 128 x 128 tiles, 64 FLOP per byte pulled into the CU) sits
// at the per-CU ingest limit measured on MI355X (~17 B/clk from L2): 510-570 TFLOP/s on these shapes, MFMA pipe 16-23 %
// busy (profiles/r01_pmc_mfma_lds.txt).  Only a larger tile changes that ratio:
//   * a workgroup owns a 256 x 256 output tile (128 FLOP/B), 16 waves = 4 (M) x 4 (N), each a 64 x 64 sub-tile of
//     v_mfma_f32_16x16x32_bf16 fragments with swapped operands (a lane owns 4 consecutive output columns);
//   * K advances 32 elements (one 64-byte LDS row per matrix row) per step; the A and W slabs of a step (16 KiB each)
//     arrive by LDS-DMA (global_load_lds, 16 B/lane) into a ring of four stages issued three steps ahead and retired by
//     COUNTED s_waitcnt vmcnt(n) + raw s_barrier, so the DMA queue never drains inside the loop;
//   * 64-byte rows with the XOR swizzle on the SOURCE chunk (conflict-free ds_read_b128 of 16 consecutive rows);
//   * epilogue in registers (bias, residual, SiLU, SiLU(gate) * up on the interleaved weight packing), then staged through
//     the (now free) ring so that global stores are whole 16-byte runs of an output row;
//   * XCD-aware block order: the N tiles of one M tile run on one XCD and share its L2 copy of the A slab.
#include <algorithm>
#include <atomic>
#include <cstdlib>

#include "igemm.h"
#include "switches.h"

namespace ivg {

struct G256Dev {
  const bf16_t* X; const bf16_t* W; bf16_t* Y; const bf16_t* R; const float* bias;
  int M, N, K, ldx, ldw, ldy;
  int tiles_n;
  int flags;
  int tiles_m, gn;   // tile order: groups of gn N tiles, inside a group M outer / N inner (see g256_tile)
};

// Work item v -> (tile_m, tile_n).  Each XCD walks a contiguous run of items.  N fastest over ALL N tiles (round 2) keeps the A slab of
// an M tile in the XCD's L2 but streams the whole weight matrix through it once per M tile: gate/up of the prompt pass is 9.4 MB of
// W against 4 MB of L2 -> 129 x 9.4 MB = 1.2 GB from beyond the L2 per GEMM.  Blocked: the N tiles are taken in groups whose W
// slabs fit the L2 (gn x 256 rows x K), every M tile is walked inside a group before the next group starts: W is fetched once
// per XCD and group, A once per group (gate/up: 3-4 x 50 MB instead of 1.2 GB).
template <typename Dev>
__device__ __forceinline__ void g256_tile(const Dev& p, int v, int& tile_m, int& tile_n) {
  const int per_group = p.gn * p.tiles_m;
  const int g = v / per_group, rem = v - g * per_group;
  const int gsize = min(p.gn, p.tiles_n - g * p.gn);
  tile_m = rem / gsize;
  tile_n = g * p.gn + (rem - tile_m * gsize);
}

__device__ __attribute__((aligned(16))) unsigned char g_zero_chunk_g256[16];

constexpr int G256_PITCH = 256 * 2 + 16;          // staged output row (bytes): + 16 spreads the 16 rows of a fragment over banks

// epilogue shared by both kernels: lane holds 4 consecutive columns n of row (wm*64 + b*16 + lr); bias, residual, SiLU, SiLU(gate) * up
// in registers, then staged through LDS (the ring is free) so that global stores are whole 16-byte runs of an output row
template <int BN>
__device__ __forceinline__ void g256_epilogue(const G256Dev& p, f32x4 (&acc)[BN / 64][4], unsigned char* smem, int m0, int n0, int wm, int wn, int lr, int lg,
                                              int tid) {
  constexpr int FN = BN / 64, WNC = BN / 4;           // N fragments per wave, columns per wave (4 waves along N)
  const int flags = p.flags;
  const bool glu = flags & IG_GLU;
  const int out_cols = glu ? BN / 2 : BN;             // columns this tile writes
  const int out_n0 = glu ? (n0 >> 1) : n0;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int row = wm * 64 + b * 16 + lr;
    const int m = m0 + row;
#pragma unroll
    for (int a = 0; a < FN; ++a) {
      if (glu && (a & 1)) continue;
      float v4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v4[r] = acc[a][b][r];
      const int ncol = wn * WNC + a * 16 + lg * 4;    // column inside the BN-wide tile (of the PACKED weight rows for GLU)
      if (flags & IG_BIAS_N) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v4[r] += p.bias[n0 + ncol + r];
      }
      int ocol = ncol;
      if (glu) {  // rows [16 gate | 16 up] per 32 packed weight rows: fragment a = gate, a + 1 = up of the same 16 outputs
        const int a1 = a + 1 < FN ? a + 1 : a;
#pragma unroll
        for (int r = 0; r < 4; ++r) v4[r] = silu_t<bf16_t>(v4[r]) * acc[a1][b][r];
        ocol = (wn * WNC + a * 16) / 2 + lg * 4;
      }
      if ((flags & IG_RESIDUAL) && m < p.M) {
        const bf16x4 rv = *(const bf16x4*)(p.R + (long)m * p.ldy + out_n0 + ocol);
#pragma unroll
        for (int r = 0; r < 4; ++r) v4[r] += (float)rv[r];
      }
      if (flags & IG_SILU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v4[r] = silu_t<bf16_t>(v4[r]);
      }
      *(bf16x4*)(smem + row * G256_PITCH + ocol * 2) = bf16x4{(bf16_t)v4[0], (bf16_t)v4[1], (bf16_t)v4[2], (bf16_t)v4[3]};
    }
  }
  __syncthreads();
  const int cpr = out_cols / 8;                        // 16-byte chunks per output row
  for (int q = tid; q < 256 * cpr; q += 1024) {
    const int row = q / cpr, ch = q - row * cpr;
    if (m0 + row >= p.M) continue;
    const Chunk16 val = *(const Chunk16*)(smem + row * G256_PITCH + ch * 16);
    *(Chunk16*)(p.Y + (long)(m0 + row) * p.ldy + out_n0 + ch * 8) = val;
  }
}

// (Round 2's kernel of this file -- 64-byte rows, K steps of 32 elements, four 32 KiB stages -- was removed in round 4: every dense GEMM of
// the released models has K % 64 == 0 and runs on the whole-line kernel below, which replaced it at +7 % on the class,
// profiles/r03_gemm256_line_ab.txt; other K fall to the generic implicit GEMM.)

// ---- whole-line kernel (round 3).  Its predecessor requested 64-byte rows (K steps of 32 elements): half a cache line per row and
// request, the shape the round-2 decode micro-benchmarks measured at 12-16 B/clk/CU out of L2 against 25-49 for whole 128-byte
// lines -- and 32 KiB per 1,030 MFMA clocks is exactly what this kernel was getting (11 B/clk/CU, MFMA pipe 25-41 % busy).  Here a
// step is 64 elements of K: every LDS-DMA instruction fetches 8 rows x one whole line, with the source-side chunk permutation of
// dgemm.hip (slot jj of row r holds source chunk jj ^ ((r >> 1) & 7)): the lane-linear [16 rows][8 chunks] image is read back as
// MFMA fragments by conflict-free ds_read_b128.  Two stages of 64 KiB (A | W), one barrier per 32 MFMAs per wave; the address
// is split the way the hardware takes it (scalar base advanced per step + a 32-bit lane offset set up once).
constexpr int G256L_HALF = 256 * 128;             // one operand of a stage: 256 rows x 128 bytes = 16 tiles of 2 KiB
constexpr int G256L_STAGE = 2 * G256L_HALF;
__device__ __forceinline__ void g256l_dma16(const void* sbase, unsigned voff, unsigned lds_wave_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_wave_base), "v"(voff), "s"(sbase) : "memory");
}
// BN = 128 (round 5): the same 256 rows against 128 weight rows -- 16 waves = 4 (M) x 4 (N) of 64 x 32 sub-tiles, waves 0-7 fetch the
// 16 KiB weight half of a stage.  For the dense 1x1 layers with 128 output channels over millions of pixels (the 64 x 64 / 256 x 256
// level's resnet shortcuts, 256 -> 128): HBM-bound shapes (85 FLOP per byte) the 128 x 128 implicit GEMM ran at 8.7 % MFMA-busy.
template <int BN>
__global__ __launch_bounds__(1024) void gemm256l_kernel(const G256Dev p) {
  constexpr int FN = BN / 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int wm = wave & 3, wn = wave >> 2;
  const int nwg = gridDim.x;
  int v;
  {
    const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  int tile_m, tile_n;
  g256_tile(p, v, tile_m, tile_n);
  const int m0 = tile_m * 256, n0 = tile_n * BN;
  const int steps = p.K >> 6;
  // wave w fills tile w (16 rows) of both operands: two requests of 8 rows x 128 bytes each
  const int r8 = lane >> 3, jj = lane & 7;
  unsigned aoff[2], woff[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r = h * 8 + r8;
    const unsigned sw = (unsigned)(jj ^ ((r >> 1) & 7)) * 16u;
    aoff[h] = (unsigned)min(m0 + wave * 16 + r, p.M - 1) * (unsigned)(p.ldx * 2) + sw;   // rows beyond M re-read row M - 1 (never stored)
    woff[h] = (unsigned)(n0 + min(wave * 16 + r, BN - 1)) * (unsigned)(p.ldw * 2) + sw;
  }
  const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) const void*)smem;
  auto issue = [&](int step) {
    const unsigned base = lds0 + (step & 1) * G256L_STAGE + wave * 2048;
    const char* xs = (const char*)p.X + (size_t)step * 128;
    const char* ws = (const char*)p.W + (size_t)step * 128;
#pragma unroll
    for (int h = 0; h < 2; ++h) g256l_dma16(xs, aoff[h], base + h * 1024);
    if (BN == 256 || wave < BN / 16) {
#pragma unroll
      for (int h = 0; h < 2; ++h) g256l_dma16(ws, woff[h], base + G256L_HALF + h * 1024);
    }
  };
  f32x4 acc[FN][4];   // [a: N fragment][b: M fragment]
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fslot0 = lr * 8, fkey = (lr >> 1) & 7;
  issue(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int s = 0; s < steps; ++s) {
    if (s + 1 < steps) issue(s + 1);   // the other stage: last read in step s - 1, before the barrier that ended it
    const unsigned char* sa = smem + (s & 1) * G256L_STAGE;
    const unsigned char* sw_ = sa + G256L_HALF;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int slot = fslot0 + ((tt * 4 + lg) ^ fkey);
      Chunk16 xa[4], wv[FN];
#pragma unroll
      for (int b = 0; b < 4; ++b) xa[b] = *(const Chunk16*)(sa + ((wm * 4 + b) * 128 + slot) * 16);
#pragma unroll
      for (int a = 0; a < FN; ++a) wv[a] = *(const Chunk16*)(sw_ + ((wn * FN + a) * 128 + slot) * 16);
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b0 = 0; b0 < 4; ++b0) {
          const int b = (a & 1) ? 3 - b0 : b0;   // snake order: one operand changes per MFMA (conv3x3.hip has the measurement)
          acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wv[a]), __builtin_bit_cast(bf16x8, xa[b]),
                                                              acc[a][b], 0, 0, 0);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  g256_epilogue<BN>(p, acc, smem, m0, n0, wm, wn, lr, lg, tid);
}

// ---- split-bf16 ("x3") instance (round 6): fp32 operands in HBM and in LDS, 256 x 256 output tile.  The prompt pass and the dense 1 x 1
// layers of the 1e-3-compliant mode (IVG_F32X3) ran on the 128 x 128 implicit GEMM (igemm_kernel<float, ..., X3>); on this tile the
// mode's rollout takes 285.4 instead of 292.5 ms and its decode stage 118.1 instead of 120.6 (ABAB, NOTES_r06.md 9b); the launches run
// at ~1.05 PFLOP/s of bf16 MFMA work.  Same whole-line staging as gemm256l_kernel -- a step is one 128-byte line per row = 32 fp32 elements of K, two stages of
// 64 KiB (A | W) -- and the arithmetic of the other X3 kernels: a fragment slot (4 fp32 of K) is split in registers into
// [bf16 hi(4) | bf16 lo(4)], the weight slot is duplicated into [w_hi | w_hi] and [w_lo | w_lo], and two K = 32 bf16 MFMAs produce all
// four partial products with fp32 accumulation.  64 MFMAs per wave and stage (four per fragment pair) against the bf16 kernel's 32:
// the per-CU ingest that bounds gemm256l (64 KiB per 2,048 matrix clocks) is halved here (64 KiB per 4,096).  fp32 output straight
// from the registers (a lane owns 4 consecutive columns = one 16-byte store); epilogues as gemm256l.
struct G256XDev {
  const float* X; const float* W; float* Y; const float* R; const float* bias;
  int M, N, K, ldx, ldw, ldy;
  int tiles_n;
  int flags;
  int tiles_m, gn;
};

__device__ __forceinline__ Chunk16 g256_split_hi_lo(const Chunk16 raw) {
  const f32x4 x = __builtin_bit_cast(f32x4, raw);
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) { const bf16_t hi = (bf16_t)x[j]; o[j] = hi; o[4 + j] = (bf16_t)(x[j] - (float)hi); }
  return __builtin_bit_cast(Chunk16, o);
}

__global__ __launch_bounds__(1024) void gemm256x3_kernel(const G256XDev p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int wm = wave & 3, wn = wave >> 2;
  const int nwg = gridDim.x;
  int v;
  {
    const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  int tile_m, tile_n;
  g256_tile(p, v, tile_m, tile_n);
  const int m0 = tile_m * 256, n0 = tile_n * 256;
  const int steps = p.K >> 5;          // 32 fp32 of K = one 128-byte line per row
  const int r8 = lane >> 3, jj = lane & 7;
  unsigned aoff[2], woff[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r = h * 8 + r8;
    const unsigned sw = (unsigned)(jj ^ ((r >> 1) & 7)) * 16u;
    aoff[h] = (unsigned)min(m0 + wave * 16 + r, p.M - 1) * (unsigned)(p.ldx * 4) + sw;   // rows beyond M re-read row M - 1 (never stored)
    woff[h] = (unsigned)(n0 + wave * 16 + r) * (unsigned)(p.ldw * 4) + sw;
  }
  const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) const void*)smem;
  auto issue = [&](int step) {
    const unsigned base = lds0 + (step & 1) * G256L_STAGE + wave * 2048;
    const char* xs = (const char*)p.X + (size_t)step * 128;
    const char* ws = (const char*)p.W + (size_t)step * 128;
#pragma unroll
    for (int h = 0; h < 2; ++h) g256l_dma16(xs, aoff[h], base + h * 1024);
#pragma unroll
    for (int h = 0; h < 2; ++h) g256l_dma16(ws, woff[h], base + G256L_HALF + h * 1024);
  };
  f32x4 acc[4][4];   // [a: N fragment][b: M fragment]
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fslot0 = lr * 8, fkey = (lr >> 1) & 7;
  issue(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int s = 0; s < steps; ++s) {
    if (s + 1 < steps) issue(s + 1);
    const unsigned char* sa = smem + (s & 1) * G256L_STAGE;
    const unsigned char* sw_ = sa + G256L_HALF;
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int slot = fslot0 + ((tt * 4 + lg) ^ fkey);
      Chunk16 xs[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) xs[b] = g256_split_hi_lo(*(const Chunk16*)(sa + ((wm * 4 + b) * 128 + slot) * 16));
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const Chunk16 ws = g256_split_hi_lo(*(const Chunk16*)(sw_ + ((wn * 4 + a) * 128 + slot) * 16));
#pragma unroll
        for (int h = 0; h < 2; ++h) {   // one duplicated half live at a time (registers); an accumulator is revisited after 4 MFMAs
          Chunk16 wd = Chunk16{ws[2 * h], ws[2 * h + 1], ws[2 * h], ws[2 * h + 1]};
          asm volatile("" : "+v"(wd));
#pragma unroll
          for (int b0 = 0; b0 < 4; ++b0) {
            const int b = ((a * 2 + h) & 1) ? 3 - b0 : b0;   // snake order: the activation fragment stays when the weight half changes
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wd), __builtin_bit_cast(bf16x8, xs[b]),
                                                                acc[a][b], 0, 0, 0);
          }
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  // ---- epilogue in registers: lane holds columns ncol .. ncol + 3 of row (wm * 64 + b * 16 + lr)
  const int flags = p.flags;
  const bool glu = flags & IG_GLU;
  const int out_n0 = glu ? (n0 >> 1) : n0;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int m = m0 + wm * 64 + b * 16 + lr;
    if (m >= p.M) continue;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      if (glu && (a & 1)) continue;
      float v4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v4[r] = acc[a][b][r];
      const int ncol = wn * 64 + a * 16 + lg * 4;
      if (flags & IG_BIAS_N) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v4[r] += p.bias[n0 + ncol + r];
      }
      int ocol = ncol;
      if (glu) {  // rows [16 gate | 16 up] per 32 packed weight rows: fragment a = gate, a + 1 = up of the same 16 outputs
        const int a1 = a + 1 < 4 ? a + 1 : a;
#pragma unroll
        for (int r = 0; r < 4; ++r) v4[r] = silu_t<float>(v4[r]) * acc[a1][b][r];
        ocol = (wn * 64 + a * 16) / 2 + lg * 4;
      }
      float* dst = p.Y + (long)m * p.ldy + out_n0 + ocol;
      if (flags & IG_RESIDUAL) {
        const f32x4 rv = *(const f32x4*)(p.R + (long)m * p.ldy + out_n0 + ocol);
#pragma unroll
        for (int r = 0; r < 4; ++r) v4[r] += rv[r];
      }
      if (flags & IG_SILU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v4[r] = silu_t<float>(v4[r]);
      }
      *(f32x4*)dst = f32x4{v4[0], v4[1], v4[2], v4[3]};
    }
  }
}

static std::atomic<long long> g_gemm256x3_launches{0};
long long gemm256x3_launches() { return g_gemm256x3_launches.load(); }

// fp32 tensors, split-bf16 arithmetic (a.x3): -1 when the shape is not covered (caller: igemm_kernel<float, ..., X3>)
static int launch_gemm256_x3(const IgemmArgs& a, hipStream_t stream) {
  if (!sw().gemm256) return -1;
  if (a.KH != 1 || a.KW != 1 || a.stride != 1 || a.pad != 0 || a.ups) return -1;
  if (a.nb0 * a.nb1 * a.nb2 != 1 || a.alpha != 1.0f) return -1;
  if (a.flags & ~(IG_BIAS_N | IG_RESIDUAL | IG_SILU | IG_GLU | IG_OUT_F32)) return -1;
  const long M = (long)a.Nimg * a.Hout * a.Wout;
  if (a.Hout != a.Hin || a.Wout != a.Win || a.ldx != a.Cin) return -1;   // dense rows
  if (a.c_ch != 1 || (a.c_grp > 1)) return -1;
  if (a.Nimg > 1 && a.c_img != (long)a.Hout * a.Wout * a.c_pix) return -1;
  const bool glu = a.flags & IG_GLU;
  if (M < 4096 || a.N % 256 != 0 || a.Cin % 32 != 0 || a.Cin < 64 || a.ldw % 4 != 0 || a.c_pix % 4 != 0) return -1;
  if (((uintptr_t)a.X & 15) || ((uintptr_t)a.W & 15) || ((uintptr_t)a.Y & 15) || ((a.flags & IG_RESIDUAL) && ((uintptr_t)a.R & 15))) return -1;
  if (glu && (a.flags & IG_BIAS_N)) return -1;
  if ((long)a.N * a.ldw * 4 >= (1L << 32)) return -1;
  const long row_bytes = 4L * std::max<long>(a.ldx, a.c_pix);
  const long rows_max = ((1L << 32) - 1) / row_bytes / 256 * 256;   // 32-bit byte offsets of the operand rows
  if (rows_max < 4096) return -1;
  static DynLdsOnce once_x;
  for (long r0 = 0; r0 < M; r0 += rows_max) {
    const long rows = std::min(rows_max, M - r0);
    G256XDev d;
    d.X = (const float*)a.X + r0 * a.ldx; d.W = (const float*)a.W; d.Y = (float*)a.Y + r0 * a.c_pix;
    d.R = a.R ? (const float*)a.R + r0 * a.c_pix : nullptr; d.bias = a.bias;
    d.M = (int)rows; d.N = a.N; d.K = a.Cin; d.ldx = a.ldx; d.ldw = a.ldw; d.ldy = (int)a.c_pix;
    d.tiles_n = a.N / 256;
    d.flags = a.flags;
    d.tiles_m = cdiv(rows, 256);
    {
      const long slab = 256L * d.K * 4;
      d.gn = (int)std::max(1L, std::min<long>((5L << 19) / slab, d.tiles_n));
    }
    const long tiles = (long)cdiv(rows, 256) * d.tiles_n;
    if (hipError_t e = ensure_dyn_lds(once_x, (const void*)gemm256x3_kernel, 160 * 1024); e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(gemm256x3_kernel, dim3((unsigned)tiles), dim3(1024), 2 * G256L_STAGE, stream, d);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return (int)e;
    g_gemm256x3_launches.fetch_add(1);
  }
  return 0;
}

// Returns -1 when the shape is not covered (caller uses launch_igemm).
int launch_gemm256(const IgemmArgs& a, DType dtype, hipStream_t stream) {
  if (dtype == F32 && a.x3 && sw().gemm256x3) return launch_gemm256_x3(a, stream);
  if (dtype != BF16 || !sw().gemm256) return -1;
  if (a.KH != 1 || a.KW != 1 || a.stride != 1 || a.pad != 0 || a.ups) return -1;
  if (a.nb0 * a.nb1 * a.nb2 != 1 || a.alpha != 1.0f) return -1;
  if (a.flags & ~(IG_BIAS_N | IG_RESIDUAL | IG_SILU | IG_GLU)) return -1;
  const long M = (long)a.Nimg * a.Hout * a.Wout;
  if (a.Hout != a.Hin || a.Wout != a.Win || a.ldx != a.Cin) return -1;   // dense rows
  if (a.c_ch != 1 || (a.c_grp > 1)) return -1;
  if (a.Nimg > 1 && a.c_img != (long)a.Hout * a.Wout * a.c_pix) return -1;
  const bool glu = a.flags & IG_GLU;
  const int BN = a.N % 256 == 0 ? 256 : 128;
  if (BN == 128 && (glu || M < (1L << 18))) return -1;   // the half-width tile: plain epilogues, HBM-bound 1x1 layers over >= 2^18 rows
  if (M < 4096 || M > 0x7fffffffL || a.N % 128 != 0 || a.Cin % 32 != 0 || a.Cin < 64 || a.ldw % 8 != 0 || a.c_pix % 8 != 0) return -1;
  if (((uintptr_t)a.X & 15) || ((uintptr_t)a.W & 15) || ((uintptr_t)a.Y & 15) || ((a.flags & IG_RESIDUAL) && ((uintptr_t)a.R & 7))) return -1;
  if (glu && (a.flags & IG_BIAS_N)) return -1;
  // whole-line requests: K steps of 64 elements; other K (no released shape) -> generic implicit GEMM
  if (a.Cin % 64 != 0 || (long)a.N * a.ldw * 2 >= (1L << 31)) return -1;
  // The kernel addresses its operands with 32-bit byte offsets: a launch covers at most 2 GiB of X and of Y.  Larger row counts
  // (256 x 256 frames: 14.7 M rows of the top decoder level's 1 x 1 shortcuts, 7.5 GB of activations -- round 6: they fell back to the
  // implicit GEMM, 4.8 ms instead of ~2.4) run as consecutive launches over row ranges; rows are dense, so a range is just an offset.
  const long row_bytes = 2L * std::max<long>(a.ldx, a.c_pix);
  long rows_max = ((1L << 31) - 1) / row_bytes / 256 * 256;
  if (rows_max < 4096) return -1;
  static DynLdsOnce once_l, once_h;
  const int smem_l = 256 * G256_PITCH > 2 * G256L_STAGE ? 256 * G256_PITCH : 2 * G256L_STAGE;
  for (long r0 = 0; r0 < M; r0 += rows_max) {
    const long rows = std::min(rows_max, M - r0);
    G256Dev d;
    d.X = (const bf16_t*)a.X + r0 * a.ldx; d.W = (const bf16_t*)a.W; d.Y = (bf16_t*)a.Y + r0 * a.c_pix;
    d.R = a.R ? (const bf16_t*)a.R + r0 * a.c_pix : nullptr; d.bias = a.bias;
    d.M = (int)rows; d.N = a.N; d.K = a.Cin; d.ldx = a.ldx; d.ldw = a.ldw; d.ldy = (int)a.c_pix;
    d.tiles_n = a.N / BN;
    d.flags = a.flags;
    d.tiles_m = cdiv(rows, 256);
    {  // N-tile groups whose weight slabs (gn x 256 rows x K bf16) stay within ~2.5 MB of the 4 MB L2 of an XCD
      const long slab = (long)BN * d.K * 2;
      d.gn = (int)std::max(1L, std::min<long>((5L << 19) / slab, d.tiles_n));
    }
    const long tiles = (long)cdiv(rows, 256) * d.tiles_n;
    if (BN == 256) {
      if (hipError_t e = ensure_dyn_lds(once_l, (const void*)gemm256l_kernel<256>, 160 * 1024); e != hipSuccess) return (int)e;
      hipLaunchKernelGGL(gemm256l_kernel<256>, dim3((unsigned)tiles), dim3(1024), smem_l, stream, d);
    } else {
      if (hipError_t e = ensure_dyn_lds(once_h, (const void*)gemm256l_kernel<128>, 160 * 1024); e != hipSuccess) return (int)e;
      hipLaunchKernelGGL(gemm256l_kernel<128>, dim3((unsigned)tiles), dim3(1024), smem_l, stream, d);
    }
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return (int)e;
  }
  return 0;
}

}  // namespace ivg
