// Decode-step GEMM, second generation (SURVEY.md 2.4 K14/K17 at L = 1):  Y[m][n] = epi( sum_k X[m][k] * W[n][k] ),  M <= 128.
//
// What the round-2 micro-benchmarks (tools/ubench/decode_ubench.hip, profiles/r02_decode_ubench.txt) showed about the
// first-generation kernel (removed in round 4): it pulled the ACTIVATION rows in MFMA-fragment shape -- 16 rows x 64 bytes per
// wave instruction, half a cache line per row -- and that shape runs at 12-16 B/clk/CU out of L2, while whole 128-byte
// lines run at 25-49 B/clk/CU (3x).  The activations are 60 % of the bytes a workgroup ingests, so here:
//   * activations arrive by LDS-DMA (global_load_lds, 16 B per lane) as WHOLE LINES: 8 consecutive lanes fetch the 8 chunks
//     of one line; the chunk order inside the line is permuted on the SOURCE side (chunk ^ ((row >> 1) & 7)) so that the
//     lane-linear LDS image [16 rows][8 chunks] is read back as MFMA fragments by conflict-free ds_read_b128;
//   * the staging area is private to a wave (its K slice of the rows): no workgroup barrier before the MFMAs;
//   * weights take the same road (whole lines by LDS-DMA, non-temporal: streamed once per step): with NO register-destination
//     load in flight the compiler has nothing to drain with vmcnt(0) (it does so whenever ordinary loads and LDS-DMA share the
//     counter), so the hand-counted waits can release the lines group by group and the MFMAs / norm sums of the first
//     128 bytes of K run while the rest of the wave's slice is still arriving;
//   * the residual rows the epilogue updates are requested at kernel start instead of after the K reduction.
// Waves split K (fixed partition per (K, dtype): a trajectory's result does not depend on its batch-mates), combine
// through LDS in a fixed order; epilogues: RMSNorm row scale (weight folded into W), residual, SiLU(gate)*up,
// step-counter advance.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include <atomic>
#include "igemm.h"
#include "switches.h"

namespace ivg {

struct DgDev {
  const void* X; const void* W; void* Y;
  int M, N, K, ldx, ldw, ldy, flags;
  float eps;
  int* bump;
  int nburst;   // bursts of 8 * LG chunks per wave
  unsigned long long* prof; const int* pos; int prof_ld;
  long long* dbg;   // development (tools/ubench/dgemm_phase.hip): per (workgroup, wave) phase stamps, null in production
  // cache warm-up of the next launch's weights (same scheme as dgemm3.hip): tile t of pf_tile_bytes is read by XCD t % 8
  const char* pf_base; unsigned pf_tile_bytes; int pf_tiles; int pf_per_wave;
  int w_nt;     // weights by non-temporal requests (one batch: every byte is read once per token) or default-policy ones (SkinnyArgs.w_shared)
};

// LDS-DMA with the address split the way the hardware takes it: wave-uniform 64-bit base (scalar registers, advanced per burst /
// line group by the scalar unit) + 32-bit byte offset per lane (set up once).  __builtin_amdgcn_global_load_lds always gets a
// 64-bit per-lane address: a 64-bit vector add and a v_readfirstlane for M0 per transfer -- with 36 transfers per burst that
// was a third of the vector instructions of this kernel, which has ONE wave per SIMD and so nothing to hide them under.
// lds_wave_base: LDS byte address the wave's 64 x 16 B land at (M0; reserved register, the compiler has no use of its own here).
__device__ __forceinline__ void dg_dma16(const void* sbase, unsigned voff, unsigned lds_wave_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_wave_base), "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void dg_dma16_nt(const void* sbase, unsigned voff, unsigned lds_wave_base) {   // nt: streamed-once weights
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt" ::"s"(lds_wave_base), "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ unsigned dg_lds_addr(const void* p) { return (unsigned)(unsigned long)(__attribute__((address_space(3))) const void*)p; }

template <typename T> struct Vec4T;
template <> struct Vec4T<bf16_t> { typedef bf16x4 type; };
template <> struct Vec4T<float> { typedef f32x4 type; };

// MF: 16-row tiles of X per workgroup, FN: 16-row tiles of W, LG: 128-byte lines per row per burst, WMAX: launch bound (waves)
template <typename T, int MF, int FN, int LG, int WMAX>
__global__ __launch_bounds__(WMAX * 64) void dgemm_kernel(const DgDev p) {
  constexpr int KS = 2 * LG;                       // MFMA K-steps per burst (4 chunks = 64 bytes of a row each)
  constexpr int NFRAG = FN * MF;
  constexpr bool PREFETCH_RES = NFRAG <= 4;
  typedef typename Vec4T<T>::type V4;
  const unsigned long long t_start = p.prof ? (unsigned long long)wall_clock64() : 0ull;
  const int prof_pos = p.prof ? *p.pos : 0;   // read up front: the lm_head launch advances the counter at its end
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, waves = (int)blockDim.x >> 6;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform values stay in scalar registers
  long long* dbg = p.dbg ? p.dbg + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + wave) * 16 : nullptr;
  auto stamp = [&](int i) { if (dbg && lane == 0) dbg[i] = (long long)__builtin_readcyclecounter(); };
  if (dbg && lane == 0) dbg[8] = (long long)wall_clock64();
  stamp(0);
  const int lr = lane & 15, lg = lane >> 4;
  const int n_tile = blockIdx.x * 16 * FN, m_tile = blockIdx.y * 16 * MF;
  unsigned char* stage = smem + (size_t)wave * (LG * (MF + FN) * 2048);   // [LG][MF] activation tiles, then [LG][FN] weight tiles
  unsigned char* stage_w = stage + LG * MF * 2048;
  const long cbeg = (long)wave * p.nburst * (8 * LG);   // first 16-byte chunk (along K) of this wave
  const char* X = (const char*)p.X;
  const char* W = (const char*)p.W;
  const bool glu = p.flags & IG_GLU;
  const bool do_norm = p.flags & SK_NORM;

  // ---- residual rows of the fragments this wave will finalise: requested now, consumed after the K reduction
  V4 res[PREFETCH_RES ? NFRAG : 1];
  if constexpr (PREFETCH_RES) {
    if ((p.flags & IG_RESIDUAL) && !(p.flags & IG_OUT_F32)) {
#pragma unroll
      for (int i = 0; i < NFRAG; ++i) {
        const int f = wave + i * waves;
        if (f < NFRAG) {
          const int a = f / MF, b = f - a * MF;
          const int m = m_tile + b * 16 + lr, n0 = n_tile + a * 16 + lg * 4;
          if (m < p.M && n0 + 3 < p.N) res[i] = *(const V4*)((const T*)p.Y + (long)m * p.ldy + n0);
        }
      }
    }
  }

  // ---- per-lane source addresses
  // (32-bit byte offsets from X / W: the launcher checks that the operands are smaller than 2 GiB)
  const int r8 = lane >> 3, jj = lane & 7;
  unsigned xoff[MF][2], woff[FN][2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r = h * 8 + r8;
    const unsigned sw = (unsigned)(jj ^ ((r >> 1) & 7)) * 16u;   // chunk of the line this lane fetches
#pragma unroll
    for (int b = 0; b < MF; ++b) {
      const int m = min(m_tile + b * 16 + r, p.M - 1);
      xoff[b][h] = (unsigned)m * (unsigned)p.ldx * (unsigned)sizeof(T) + sw;
    }
#pragma unroll
    for (int a = 0; a < FN; ++a) {
      const int n = min(n_tile + a * 16 + r, p.N - 1);
      woff[a][h] = (unsigned)n * (unsigned)p.ldw * (unsigned)sizeof(T) + sw;
    }
  }
  const unsigned stage_lds = dg_lds_addr(stage), stage_w_lds = dg_lds_addr(stage_w);

  f32x4 acc[FN][MF];
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < MF; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  float ssq[MF];
#pragma unroll
  for (int b = 0; b < MF; ++b) ssq[b] = 0.f;

  for (int burst = 0; burst < p.nburst; ++burst) {
    const long c0 = cbeg + (long)burst * (8 * LG);
    if (burst) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the previous burst's fragment reads have left the staging area
    // Issue order = consumption order: [activation lines | weight lines] of group g (128 bytes of K) for g = 0 .. LG-1; memory
    // returns in order, so group g has landed once at most (LG - 1 - g) * (2 MF + 2 FN) younger DMA requests are outstanding.
#pragma unroll
    for (int g = 0; g < LG; ++g) {
#pragma unroll
      for (int b = 0; b < MF; ++b)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          dg_dma16(X + (c0 + g * 8) * 16, xoff[b][h], stage_lds + ((g * MF + b) * 2 + h) * 1024);
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int h = 0; h < 2; ++h)
          if (p.w_nt) dg_dma16_nt(W + (c0 + g * 8) * 16, woff[a][h], stage_w_lds + ((g * FN + a) * 2 + h) * 1024);
          else dg_dma16(W + (c0 + g * 8) * 16, woff[a][h], stage_w_lds + ((g * FN + a) * 2 + h) * 1024);
    }
    if (burst == 0) stamp(1);
#pragma unroll
    for (int g = 0; g < LG; ++g) {
      constexpr int PER = 2 * MF + 2 * FN;
      if (g == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LG - 1) * PER) : "memory");
      else if (g == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((LG > 1 ? LG - 2 : 0) * PER) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (burst == 0 && g == 0) stamp(2);
      if (burst == p.nburst - 1 && g == LG - 1) stamp(3);
      Chunk16 xa[2][MF], wa[2][FN];
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        const int j = tt * 4 + lg;   // chunk of the line
        const int slot = lr * 8 + (j ^ ((lr >> 1) & 7));
#pragma unroll
        for (int b = 0; b < MF; ++b) xa[tt][b] = *(const Chunk16*)(stage + ((g * MF + b) * 128 + slot) * 16);
#pragma unroll
        for (int a = 0; a < FN; ++a) wa[tt][a] = *(const Chunk16*)(stage_w + ((g * FN + a) * 128 + slot) * 16);
      }
      if (do_norm) {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
          for (int b = 0; b < MF; ++b) {
            if constexpr (sizeof(T) == 2) {
              // v_dot2c_f32_bf16: two squares per instruction, no unpacking (4 instead of 12 instructions per fragment)
              const bf16x8 xx = __builtin_bit_cast(bf16x8, xa[tt][b]);
#pragma unroll
              for (int u = 0; u < 4; ++u) {
                const bf16x2 pr = bf16x2{xx[2 * u], xx[2 * u + 1]};
                ssq[b] = __builtin_amdgcn_fdot2_f32_bf16(pr, pr, ssq[b], false);
              }
            } else {
              const f32x4 xx = __builtin_bit_cast(f32x4, xa[tt][b]);
#pragma unroll
              for (int u = 0; u < 4; ++u) ssq[b] = fmaf(xx[u], xx[u], ssq[b]);
            }
          }
      }
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int a = 0; a < FN; ++a)
#pragma unroll
          for (int b = 0; b < MF; ++b) {
            if constexpr (sizeof(T) == 2) {
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wa[tt][a]),
                                                                  __builtin_bit_cast(bf16x8, xa[tt][b]), acc[a][b], 0, 0, 0);
            } else {
              const f32x4 wf = __builtin_bit_cast(f32x4, wa[tt][a]), xf = __builtin_bit_cast(f32x4, xa[tt][b]);
#pragma unroll
              for (int u = 0; u < 4; ++u) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u], xf[u], acc[a][b], 0, 0, 0);
            }
          }
    }
  }

  // ---- cache warm-up of the next launch's weight tiles (this wave's share of the tiles its XCD will read): the requests travel
  // while this launch combines and stores and are waited for at the very end (round 3; see dgemm3.hip)
  Chunk16 pf_sink = Chunk16{0u, 0u, 0u, 0u};
  if (p.pf_per_wave > 0) {
    const int Lb = blockIdx.y * gridDim.x + blockIdx.x, x = Lb & 7, q = Lb >> 3;
    const unsigned upt = p.pf_tile_bytes >> 10;
    const unsigned total = (unsigned)((p.pf_tiles - x + 7) >> 3) * upt;
    unsigned v = ((unsigned)q * (unsigned)waves + (unsigned)wave) * (unsigned)p.pf_per_wave;
    for (int i = 0; i < p.pf_per_wave; ++i, ++v) {
      if (v >= total) break;
      const unsigned j = v / upt, off = v - j * upt;
      const char* src = p.pf_base + (size_t)(x + 8 * j) * p.pf_tile_bytes + (size_t)off * 1024 + lane * 16;
      asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(pf_sink) : "v"(src) : "memory");
    }
  }
  // ---- combine the waves' K slices (fixed order w = 0 .. waves-1); the combine area aliases the staging areas
  stamp(4);
  __syncthreads();
  stamp(5);
  f32x4* red = (f32x4*)smem;                                  // [waves][FN][MF][64 lanes]
  float* s_ss = (float*)(red + (size_t)waves * NFRAG * 64);   // [waves][MF * 16 rows]
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < MF; ++b) red[((wave * FN + a) * MF + b) * 64 + lane] = acc[a][b];
  if (do_norm) {
#pragma unroll
    for (int b = 0; b < MF; ++b) {
      float v = ssq[b];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      if (lg == 0) s_ss[wave * (MF * 16) + b * 16 + lr] = v;
    }
  }
  __syncthreads();
  stamp(6);

  const bool f32out = p.flags & IG_OUT_F32;
#pragma unroll
  for (int i = 0; i < NFRAG; ++i) {
    const int f = wave + i * waves;
    if (f >= NFRAG) break;
    const int a = f / MF, b = f - a * MF;
    if (glu && (a & 1)) continue;
    f32x4 v = red[((0 * FN + a) * MF + b) * 64 + lane];
    for (int w = 1; w < waves; ++w) v += red[((w * FN + a) * MF + b) * 64 + lane];
    const int m = m_tile + b * 16 + lr;
    int n0 = n_tile + a * 16 + lg * 4;
    if (m >= p.M || n0 >= p.N) continue;
    int nlim = p.N;
    float rs = 1.0f;
    if (do_norm) {
      float tot = 0.f;
      for (int w = 0; w < waves; ++w) tot += s_ss[w * (MF * 16) + b * 16 + lr];
      rs = rsqrtf(tot / (float)p.K + p.eps);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= rs;
    }
    if (glu) {
      if constexpr (FN >= 2) {
        const int a1 = a + 1 < FN ? a + 1 : a;
        f32x4 u = red[((0 * FN + a1) * MF + b) * 64 + lane];
        for (int w = 1; w < waves; ++w) u += red[((w * FN + a1) * MF + b) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = silu_t<T>(v[r]) * (u[r] * rs);
      }
      n0 = (n_tile >> 1) + (a >> 1) * 16 + lg * 4;
      nlim = p.N >> 1;
    }
    if (f32out) {
      float* Y = (float*)p.Y + (long)m * p.ldy + n0;
      if (n0 + 3 < nlim && ((p.ldy & 3) == 0)) *(f32x4*)Y = v;
      else {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (n0 + r < nlim) Y[r] = v[r];
      }
    } else {
      T* Y = (T*)p.Y + (long)m * p.ldy + n0;
      const bool whole = n0 + 3 < nlim;
      if (p.flags & IG_RESIDUAL) {  // in-place residual-stream update: each element is read and written by one thread
        if (whole) {
          V4 o;
          if constexpr (PREFETCH_RES) o = res[i]; else o = *(const V4*)Y;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = from_f32<T>(to_f32(o[r]) + v[r]);
          *(V4*)Y = o;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (n0 + r < nlim) Y[r] = from_f32<T>(to_f32(Y[r]) + v[r]);
        }
      } else {
        if (whole) {
          V4 o;
#pragma unroll
          for (int r = 0; r < 4; ++r) o[r] = from_f32<T>(v[r]);
          *(V4*)Y = o;
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (n0 + r < nlim) Y[r] = from_f32<T>(v[r]);
        }
      }
    }
  }
  stamp(7);
  if (p.pf_per_wave > 0) {   // the warm-up requests name a register: it stays reserved until they have all returned
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" ::"v"(pf_sink));
  }
  if (dbg && lane == 0) dbg[9] = (long long)wall_clock64();
  if (p.prof && tid == 0) {
    unsigned long long* slot = p.prof + (size_t)((blockIdx.x * 7 + blockIdx.y) % IVG_GEMM_PROF_SLOTS) * 2 * p.prof_ld;
    atomicMax(slot + prof_pos, ~t_start);
    atomicMax(slot + p.prof_ld + prof_pos, (unsigned long long)wall_clock64());
  }
  if (p.bump && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0) { p.bump[0] += 1; p.bump[1] += 1; }
}

template <typename T, int MF, int FN, int LG, int WMAX>
static int launch_dg(const DgDev& d, int waves, hipStream_t stream) {
  const int nfrag = FN * MF;
  const int stage = waves * LG * (MF + FN) * 2048;
  const int comb = waves * nfrag * 64 * 16 + waves * MF * 16 * 4;
  const int smem = std::max(stage, comb);
  if (smem > 160 * 1024 || waves > WMAX) return -1;
  static DynLdsOnce once;
  auto kfn = dgemm_kernel<T, MF, FN, LG, WMAX>;
  if (hipError_t e = ensure_dyn_lds(once, (const void*)kfn, 160 * 1024); e != hipSuccess) return (int)e;
  dim3 grid((unsigned)cdiv(d.N, 16 * FN), (unsigned)cdiv(d.M, 16 * MF), 1);
  hipLaunchKernelGGL(kfn, grid, dim3(waves * 64), smem, stream, d);
  return (int)hipGetLastError();
}

// K partition: a function of (K in bytes) only -- never of the batch -- so a row's sum order does not depend on its batch-mates.
// chunks = K * sizeof(T) / 16;  waves * nburst * 8 * LG = chunks.
struct DgSplit { int waves, lg, nburst; };
static bool dg_split(long chunks, DgSplit& s) {
  if (chunks <= 0 || chunks % 8 != 0) return false;
  const long lines = chunks / 8;   // 128-byte lines per row
  // preference: bursts of three lines per wave (24 chunks: all of a wave's operands in flight at once), then two, then one;
  // as many waves as divide the K range (fewest bursts per wave)
  for (int lg : {3, 2, 1}) {
    if (lines % lg != 0) continue;
    const long units = lines / lg;           // (wave, burst) units
    for (int waves : {16, 12, 8, 6, 4, 2, 1}) {
      if (units % waves != 0) continue;
      const long nb = units / waves;
      if (nb > 8) continue;
      s = DgSplit{waves, lg, (int)nb};
      return true;
    }
  }
  return false;
}


// Measured picks (tools/dgemm_sweep.py on MI355X, profiles/r02_dgemm_sweep_*.txt) for the GEMMs of the released transformers,
// keyed by (K bytes, N) -- never by the batch, so the K partition of a GEMM is fixed.  mf caps the row tiles per workgroup.
struct DgPick { int kbytes, N, mf, fn, waves, lg; };
static const DgPick kDgPicks[] = {
    {1536, 2304, 2, 2, 4, 3},    // small: q/k/v            5.0 us per launch incl. the boundary (first generation 6.5)
    {1536, 768, 1, 1, 4, 3},     // small: o-proj           3.4 (4.8)
    {1536, 6144, 4, 2, 4, 3},    // small: gate/up          7.0 (9.6)
    {6144, 768, 1, 1, 8, 3},     // small: down             5.8 (8.9)
    {1536, 16386, 4, 2, 2, 2},   // small: lm_head         12.9 (19.2)
    {2048, 3072, 2, 2, 4, 2},    // medium (hidden 1024, intermediate 4096): q/k/v   6.5 (8.2)
    {2048, 1024, 1, 1, 4, 2},    // medium: o-proj          4.2 (5.3)
    {2048, 8192, 4, 2, 4, 2},    // medium: gate/up         8.6 (11.0)
    {8192, 1024, 1, 1, 4, 2},    // medium: down            8.8 (11.3)
    {2048, 16386, 4, 2, 2, 1},   // medium: lm_head        16.3 (21.0)
};

template <typename T, int LG>
static int launch_dg_t(const DgDev& d, int MF, int FN, int waves, hipStream_t st) {
  // launch bound = the smallest class that holds the waves (register budget: 4 waves -> 512, 8 -> 256, 16 -> 128 per lane)
#define IVG_DG(mf, fn) if (MF == mf && FN == fn) { \
    if (waves <= 4) return launch_dg<T, mf, fn, LG, 4>(d, waves, st); \
    if constexpr (mf * fn <= 8) { if (waves <= 8) return launch_dg<T, mf, fn, LG, 8>(d, waves, st); } \
    if constexpr (mf * fn <= 2) return launch_dg<T, mf, fn, LG, 16>(d, waves, st); \
    return -1; }
  IVG_DG(1, 1) IVG_DG(1, 2) IVG_DG(1, 4)
  IVG_DG(2, 1) IVG_DG(2, 2) IVG_DG(2, 4)
  IVG_DG(4, 1) IVG_DG(4, 2) IVG_DG(4, 4)
#undef IVG_DG
  return -1;
}

// rows of W one workgroup owns for this GEMM (what a predecessor's cache warm-up mirrors): the pick table, else the default tile
int dgemm_w_rows_per_block(const SkinnyArgs& a, DType dtype) {
  const int es = dtype == BF16 ? 2 : 4;
  if (((long)a.K * es) % 128 != 0) return 0;
  for (const DgPick& k : kDgPicks) if (k.kbytes == a.K * es && k.N == a.N) return 16 * k.fn;
  return 16 * ((a.flags & IG_GLU) ? 2 : 1);
}

// -1: shape not covered (launch_skinny below answers hipErrorInvalidValue); otherwise a hipError_t
int launch_dgemm(const SkinnyArgs& a, DType dtype, hipStream_t stream) {
  const int es = dtype == BF16 ? 2 : 4;
  if (a.M <= 0 || a.N <= 0 || a.M > 128) return -1;
  if (((long)a.K * es) % 128 != 0 || ((long)a.ldx * es) % 16 != 0 || ((long)a.ldw * es) % 16 != 0) return -1;
  if (((uintptr_t)a.X & 15) || ((uintptr_t)a.W & 15)) return -1;
  if ((long)a.N * a.ldw * es >= (1L << 31) || (long)a.M * a.ldx * es >= (1L << 31)) return -1;   // 32-bit per-lane offsets
  const bool glu = a.flags & IG_GLU;
  if (glu && a.N % 32 != 0) return -1;
  if ((a.flags & IG_RESIDUAL) && !(a.flags & IG_OUT_F32) && ((a.ldy & 3) != 0 || ((uintptr_t)a.Y & (4 * es - 1)))) return -1;
  DgSplit sp;
  if (!dg_split((long)a.K * es / 16, sp)) return -1;
  // tile: all rows of the batch in one workgroup when that still leaves >= ~half the CUs busy, W tiles as narrow as the
  // epilogue allows -- activations are cheap now (whole lines out of L2), weight bytes per CU are what is left to balance
  const int mt = cdiv(a.M, 16);
  int MF = mt >= 4 ? 4 : (mt >= 2 ? 2 : 1);
  int FN = glu ? 2 : 1;
  {
    auto wgs = [&](int mf, int fn) { return (long)cdiv(a.M, 16 * mf) * cdiv(a.N, 16 * fn); };
    while (MF > 1 && wgs(MF, FN) < 128) MF >>= 1;                 // narrow GEMMs: split the rows to fill the chip
    while (FN < 4 && wgs(MF, FN) > 512) FN <<= 1;                 // wide GEMMs (lm_head): fatter W tiles, fewer rounds
  }
  int waves = sp.waves, lgv = sp.lg, nburst = sp.nburst;
  for (const DgPick& k : kDgPicks) {
    if (k.kbytes != a.K * es || k.N != a.N) continue;
    const long chunks = (long)a.K * es / 16;
    if (chunks % ((long)k.waves * 8 * k.lg) != 0) break;
    MF = std::min(k.mf, mt >= 4 ? 4 : (mt >= 2 ? 2 : 1));
    FN = k.fn; waves = k.waves; lgv = k.lg;
    nburst = (int)(chunks / ((long)waves * 8 * lgv));
    break;
  }
  // The wave count (the K partition) is a function of (K bytes, N) only; the register class that holds it bounds the fragments a
  // workgroup may own (launch_dg_t: <= 4 waves any tile, <= 8 waves MF * FN <= 8, 16 waves MF * FN <= 2).  A tile the batch size
  // asked for that does not fit is CLAMPED here -- never answered with -1: a different kernel for some batch sizes only
  // would give a trajectory a different K-summation order depending on its batch-mates (fp32 parity mode: M = 64 vs a 16-row shard).
  {
    const int allowed = waves <= 4 ? 16 : (waves <= 8 ? 8 : 2);
    while (MF * FN > allowed && MF > 1) MF >>= 1;
    while (MF * FN > allowed && FN > (glu ? 2 : 1)) FN >>= 1;
  }
  // staging budget: waves * LG * (MF + FN) * 2 KiB of LDS.  Over budget: first more (shorter) bursts per wave -- the K partition over
  // the waves, and so every sum order, stays what it is -- then fewer row tiles per workgroup.  BEST EFFORT: the wave count of a shape is
  // never changed for the budget (it is the K partition), so a pick whose smallest form (one line group, one row tile) is still larger
  // launches with that form -- 16 waves x FN = 2 is 96 KiB whatever the budget says (lm_head and the medium transformer's o-proj /
  // gate-up; the q/k/v, gate-up and down GEMMs of the small transformer do fit 40 KiB)
  const int budget = (a.lds_kb > 0 ? a.lds_kb : sw().decode_lds_kb) * 1024;   // (IVG_DECODE_LDS_KB: room left for another batch's conv3x3 workgroup on the CU)
  while (lgv > 1 && waves * lgv * (MF + FN) * 2048 > budget) {
    const int total = lgv * nburst;           // lines per wave
    --lgv;
    while (total % lgv != 0) --lgv;
    nburst = total / lgv;
  }
  while (MF > 1 && waves * lgv * (MF + FN) * 2048 > budget) MF >>= 1;
  if (a.w_shared && sw().dg2_mf_cap > 0) MF = std::min(MF, sw().dg2_mf_cap);   // development A/B (IVG_DG2_MF_CAP)
  DgDev d{a.X, a.W, a.Y, a.M, a.N, a.K, a.ldx, a.ldw, a.ldy, a.flags, a.eps, a.bump, nburst, a.pos ? a.prof : nullptr, a.pos, a.prof_ld, a.dbg,
          nullptr, 0u, 0, 0, a.w_shared ? 0 : 1};
  if (a.next_W && a.next_tile_bytes >= 1024 && a.next_tiles > 0 && sw().dg3_warm) {
    const long grid = (long)cdiv(a.N, 16 * FN) * cdiv(a.M, 16 * MF);
    const long waves_per_xcd = std::max(1L, grid / 8) * waves;
    const long units_per_xcd = (long)cdiv(a.next_tiles, 8) * (a.next_tile_bytes >> 10);
    long per = (units_per_xcd + waves_per_xcd - 1) / waves_per_xcd;
    if (per > 8) per = 8;
    d.pf_base = (const char*)a.next_W; d.pf_tile_bytes = (unsigned)a.next_tile_bytes; d.pf_tiles = a.next_tiles; d.pf_per_wave = (int)per;
  }
  int rc;
  if (dtype == BF16) rc = lgv == 3 ? launch_dg_t<bf16_t, 3>(d, MF, FN, waves, stream) : lgv == 2 ? launch_dg_t<bf16_t, 2>(d, MF, FN, waves, stream)
                                                                                     : launch_dg_t<bf16_t, 1>(d, MF, FN, waves, stream);
  else rc = lgv == 3 ? launch_dg_t<float, 3>(d, MF, FN, waves, stream) : lgv == 2 ? launch_dg_t<float, 2>(d, MF, FN, waves, stream)
                                                                                 : launch_dg_t<float, 1>(d, MF, FN, waves, stream);
  return rc;
}

// Decode-step GEMM dispatcher (the name survives from the first-generation kernel, removed in round 4: no shape of a released model
// reached it any more): third generation (dgemm3.hip) wherever it covers the shape, else the second (this file).  Coverage is a
// function of (K, N, dtype, flags) only, so a GEMM of the model runs on the same kernel -- the same K-summation order -- whatever
// the batch.  A shape neither covers (K bytes not a multiple of 128, M > 128, unaligned operands) fails loudly.
static std::atomic<long long> g_gen3_launches{0}, g_gen2_launches{0};
long long decode_gemm_launches(int generation) { return (generation == 3 ? g_gen3_launches : g_gen2_launches).load(std::memory_order_relaxed); }

int launch_skinny(const SkinnyArgs& a, DType dtype, hipStream_t stream) {
  if (a.M <= 0 || a.N <= 0) return 0;
  int rc = launch_dgemm3(a, dtype, stream);
  if (rc != -1) { g_gen3_launches.fetch_add(1, std::memory_order_relaxed); return rc; }
  rc = launch_dgemm(a, dtype, stream);
  if (rc != -1) { g_gen2_launches.fetch_add(1, std::memory_order_relaxed); return rc; }
  return (int)hipErrorInvalidValue;
}

}  // namespace ivg
