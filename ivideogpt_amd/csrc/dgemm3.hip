// Decode-step GEMM, third generation (SURVEY.md 2.4 K14/K17 at L = 1):  Y[m][n] = epi( sum_k X[m][k] * W[n][k] ),  M <= 128.
//
// What the round-3 phase stamps of the second-generation kernel (tools/ubench/dgemm_phase.hip, profiles/r03_dgemm_phase_v2.txt)
// showed: a launch is bound by what ONE CU can ingest through LDS-DMA -- a wave gets one 1 KiB global_load_lds through every
// ~100-150 clocks whatever it does around it, so four waves pull ~30 B/clk/CU and the 24-36 requests of a wave's K slice take
// 3,300-4,900 clocks to ISSUE, before the HBM latency of the last one even starts; then a runtime-length combine (two barriers,
// ds_bpermute reductions, dependent LDS read loops) and an epilogue with an IEEE division added 2,000-4,000 clocks.  So:
//   * K is split over up to 16 waves (compile-time count): a wave owns a few 128-byte lines of K for all rows of the tile,
//     issues its 8-24 requests in a third of the time, and every SIMD keeps requests in flight (38-45 B/clk/CU);
//   * long K (down-proj) streams through a two-slot ring PRIVATE to the wave: a line is requested again as soon as the wave's
//     own fragment reads have left the slot (no workgroup barrier, the request queue never drains);
//   * ONE barrier: a wave parks its partial tile in its own (now idle) staging region, fixed-order combine fully unrolled,
//     RMSNorm partial sums ride along as plain LDS rows (no cross-lane shuffles), 1/K comes from the host;
//   * cache warm-up: the weight tiles of the NEXT launch of the chain are requested while this launch is busy with its own
//     combine / epilogue, each tile by a workgroup of the XCD that will consume it (block b -> XCD b % 8: verified on every launch
//     with HW_REG_XCC_ID, tools/ubench/dgemm_phase).  The dependent launch then starts on cache hits instead of a cold HBM stream
//     (consume-only micro-benchmark, profiles/r03_l2warm.txt: 9.4 MB of weights 3.7 -> 2.4 us, also across the 124 MB non-temporal
//     KV stream of the attention in between; layer chain 41.3 -> 40.1 us).  WHICH cache: the PMC pass (profiles/r03_warmup_pmc.txt)
//     shows the consumer's FETCH_SIZE unchanged -- its requests still leave the L2, with non-temporal and default-policy loads
//     alike -- so the lines do not survive the kernel boundary in the XCD's L2; what the consumer hits is the memory-side
//     Infinity Cache.  The counter therefore shows every weight byte twice (2.4 x the algorithmic bytes of the class): once
//     from HBM under the predecessor, once from the Infinity Cache on the critical path.
// Same contracts as dgemm.hip: fixed K partition per (K bytes, N) -- never per batch --, fixed-order sums, epilogues RMSNorm row
// scale (weight folded into W), in-place residual, SiLU(gate)*up, fp32 out, step-counter advance; whole-line LDS-DMA operands
// with the source-side chunk permutation that makes the lane-linear LDS image conflict-free to read back as MFMA fragments.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "igemm.h"
#include "switches.h"

namespace ivg {

struct Dg3Dev {
  const void* X; const void* W; void* Y;
  int M, N;
  unsigned ldxb, ldwb;      // row strides in bytes
  int ldy;                  // elements
  int wr;                   // rows of W one workgroup owns (<= 16 FN; a multiple of 4): the W tile pitch along N
  int klw;                  // 128-byte lines of K per wave
  int ring;                 // slots of the wave's staging ring (1: klw == 1 or no room; 2: continuous stream)
  unsigned wave_bytes;      // LDS bytes per wave (ring * line bytes)
  int flags;
  int w_nt;                  // weights by non-temporal requests (streamed once per step) or default-policy ones
  int x3;                    // fp32 tensors only: split-bf16 arithmetic (both fragments split into bf16 hi | lo in registers, two K = 32
                             // bf16 MFMAs per fragment pair instead of four f32-input ones: IVG_F32X3 rollout mode)
  float inv_k, eps;
  int* bump;
  unsigned long long* prof; const int* pos; int prof_ld;
  // cache warm-up of the next launch's weights: [tile][rows][K] contiguous tiles of pf_tile_bytes, tile t is read by XCD t % 8
  const char* pf_base; unsigned pf_tile_bytes; int pf_tiles; int pf_per_wave;
  long long* dbg;
};

__device__ __forceinline__ void dg3_dma16(const void* sbase, unsigned voff, unsigned lds_wave_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_wave_base), "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ void dg3_dma16_nt(const void* sbase, unsigned voff, unsigned lds_wave_base) {   // nt: streamed-once weights
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt" ::"s"(lds_wave_base), "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ bf16x8 dg3_split_hi_lo(const f32x4 x) {   // 4 fp32 -> [bf16 hi(4) | bf16 lo(4)], lo = bf16(x - hi)
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) { const bf16_t hi = (bf16_t)x[j]; o[j] = hi; o[4 + j] = (bf16_t)(x[j] - (float)hi); }
  return o;
}
__device__ __forceinline__ unsigned dg3_lds_addr(const void* p) { return (unsigned)(unsigned long)(__attribute__((address_space(3))) const void*)p; }

template <typename T> struct Vec4T3;
template <> struct Vec4T3<bf16_t> { typedef bf16x4 type; };
template <> struct Vec4T3<float> { typedef f32x4 type; };

// MF: 16-row tiles of X per workgroup, FN: 16-row tiles of W, WAVES: waves splitting K
// A workgroup owns p.wr <= 16 FN rows of W (o-proj / down of the small transformer: 12 of 16), so that the N tiles x M tiles of
// a GEMM come out at ~256 workgroups: every CU takes part and ingests fewer weight rows next to its activation rows (196 -> 172
// KB per CU for down).  Rows beyond wr are neither requested (lanes masked off) nor stored; wr > 16 FN - 8, so every half tile
// still has a valid row and the request count per line is a compile-time constant.
// X3 (T = float only, p.x3): split-bf16 arithmetic -- a template parameter, not a run-time branch: with both paths in one kernel the
// f32-input instances with four row tiles grew from 124-150 to 168-184 registers and spilled.
template <typename T, int MF, int FN, int WAVES, bool X3 = false>
__global__ __launch_bounds__(WAVES * 64) void dg3_kernel(const Dg3Dev p) {
  static_assert(!X3 || sizeof(T) == 4, "split-bf16 arithmetic reads fp32 tensors");
  constexpr int PER = 2 * (MF + FN);             // LDS-DMA requests per line (two 8-row halves per 16-row tile)
  constexpr int NFRAG = FN * MF;
  constexpr int LINE = (MF + FN) * 2048;         // staged bytes of one 128-byte line of K: [MF activation tiles | FN weight tiles] x 2 KiB
  constexpr int NFIN = (NFRAG + WAVES - 1) / WAVES;   // output fragments a wave finalises
  typedef typename Vec4T3<T>::type V4;
  const unsigned long long t_start = p.prof ? (unsigned long long)wall_clock64() : 0ull;
  const int prof_pos = p.prof ? *p.pos : 0;      // read up front: the lm_head launch advances the counter at its end
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  long long* dbg = p.dbg ? p.dbg + ((size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + wave) * 16 : nullptr;
  auto stamp = [&](int i) { if (dbg && lane == 0) dbg[i] = (long long)__builtin_readcyclecounter(); };
  if (dbg && lane == 0) {
    dbg[8] = (long long)wall_clock64();
    dbg[10] = (long long)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7);   // HW_REG_XCC_ID: the XCD this workgroup really runs on
  }
  stamp(0);
  const int lr = lane & 15, lg = lane >> 4;
  const int n_tile = blockIdx.x * p.wr, m_tile = blockIdx.y * 16 * MF;
  unsigned char* my = smem + (size_t)wave * p.wave_bytes;
  const unsigned my_lds = dg3_lds_addr(my);
  const bool glu = p.flags & IG_GLU;
  const bool do_norm = p.flags & SK_NORM;
  const bool res_bf = (p.flags & IG_RESIDUAL) && !(p.flags & IG_OUT_F32);

  // ---- per-lane source offsets (32-bit: the launcher checks that the operands are smaller than 2 GiB)
  const int r8 = lane >> 3, jj = lane & 7;
  unsigned xoff[MF][2], woff[FN][2];
  bool wreq[FN][2];   // this lane's row of the half tile is one of the wr rows the workgroup owns
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r = h * 8 + r8;
    const unsigned sw = (unsigned)(jj ^ ((r >> 1) & 7)) * 16u;   // chunk of the line this lane fetches
#pragma unroll
    for (int b = 0; b < MF; ++b) xoff[b][h] = (unsigned)min(m_tile + b * 16 + r, p.M - 1) * p.ldxb + sw;
#pragma unroll
    for (int a = 0; a < FN; ++a) {
      woff[a][h] = (unsigned)min(n_tile + a * 16 + r, p.N - 1) * p.ldwb + sw;   // (rows beyond N re-read row N - 1: never stored)
      wreq[a][h] = a * 16 + r < p.wr;
    }
  }
  const char* Xw = (const char*)p.X + (size_t)wave * p.klw * 128;   // this wave's K slice
  const char* Ww = (const char*)p.W + (size_t)wave * p.klw * 128;
  auto issue = [&](int line, int slot) {
    const unsigned base = my_lds + (unsigned)slot * LINE;
#pragma unroll
    for (int b = 0; b < MF; ++b)
#pragma unroll
      for (int h = 0; h < 2; ++h) dg3_dma16(Xw + (size_t)line * 128, xoff[b][h], base + (b * 2 + h) * 1024);
    // (every half tile has at least one valid row -- 16 FN - 8 < wr by construction -- so the request count per line is exactly
    // PER in every wave: the counted waits below depend on it)
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (wreq[a][h]) {
          if (p.w_nt) dg3_dma16_nt(Ww + (size_t)line * 128, woff[a][h], base + ((MF + a) * 2 + h) * 1024);
          else dg3_dma16(Ww + (size_t)line * 128, woff[a][h], base + ((MF + a) * 2 + h) * 1024);
        }
      }
  };
  // ---- residual rows of the fragments this wave will finalise: requested FIRST, consumed after the K reduction.  Register-
  // destination loads share the counter with the LDS-DMA queue and requests return in order: being the OLDEST entries they never
  // enter the counted waits below (behind the prologue's lines they would be older than the lines requested later in the loop and
  // the counts would be wrong for those).  Inline asm: a plain load may be scheduled anywhere by the compiler.
  V4 res[NFIN];
  if (res_bf) {
#pragma unroll
    for (int i = 0; i < NFIN; ++i) {
      const int f = min(wave + i * WAVES, NFRAG - 1);
      const int a = f / MF, b = f - a * MF;
      const int m = min(m_tile + b * 16 + lr, p.M - 1), n0 = min(n_tile + a * 16 + lg * 4, p.N - 4);
      const T* src = (const T*)p.Y + (long)m * p.ldy + n0;
      if constexpr (sizeof(V4) == 8) asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(res[i]) : "v"(src) : "memory");
      else asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(res[i]) : "v"(src) : "memory");
    }
  }

  int issued = 0;
  {
    const int first = min(p.ring, p.klw);
    for (; issued < first; ++issued) issue(issued, issued);
  }
  stamp(1);

  f32x4 acc[FN][MF];
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < MF; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  float ssq[MF];
#pragma unroll
  for (int b = 0; b < MF; ++b) ssq[b] = 0.f;

  for (int i = 0; i < p.klw; ++i) {
    // requests return in order: line i has landed once at most (lines issued after it) * PER requests are outstanding
    const int younger = issued - 1 - i;
    if (younger >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (i == 0) stamp(2);
    if (i == p.klw - 1) stamp(3);
    const unsigned char* stage = my + (size_t)(p.ring == 2 ? (i & 1) : 0) * LINE;
    Chunk16 xa[2][MF], wa[2][FN];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      const int j = tt * 4 + lg;   // chunk of the line
      const int slot = lr * 8 + (j ^ ((lr >> 1) & 7));
#pragma unroll
      for (int b = 0; b < MF; ++b) xa[tt][b] = *(const Chunk16*)(stage + (b * 128 + slot) * 16);
#pragma unroll
      for (int a = 0; a < FN; ++a) wa[tt][a] = *(const Chunk16*)(stage + ((MF + a) * 128 + slot) * 16);
    }
    if (issued < p.klw) {   // stream on: the slot is free once this wave's own fragment reads have returned
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      issue(issued, p.ring == 2 ? (i & 1) : 0);
      ++issued;
    }
    if (do_norm) {
#pragma unroll
      for (int tt = 0; tt < 2; ++tt)
#pragma unroll
        for (int b = 0; b < MF; ++b) {
          if constexpr (sizeof(T) == 2) {
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
    if constexpr (X3) {
      {
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
          bf16x8 xs[MF];
#pragma unroll
          for (int b = 0; b < MF; ++b) xs[b] = dg3_split_hi_lo(__builtin_bit_cast(f32x4, xa[tt][b]));
#pragma unroll
          for (int a = 0; a < FN; ++a) {
            const Chunk16 ws = __builtin_bit_cast(Chunk16, dg3_split_hi_lo(__builtin_bit_cast(f32x4, wa[tt][a])));
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              Chunk16 wd = Chunk16{ws[2 * h], ws[2 * h + 1], ws[2 * h], ws[2 * h + 1]};   // [w_hi | w_hi], then [w_lo | w_lo]
              asm volatile("" : "+v"(wd));
#pragma unroll
              for (int b = 0; b < MF; ++b)
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wd), xs[b], acc[a][b], 0, 0, 0);
            }
          }
        }
      }
    } else {
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
  stamp(4);   // (the last line's wait was vmcnt(0): the residual rows are here as well)

  // ---- cache warm-up of the next launch's weight tiles (this wave's share of the tiles its XCD will read); the requests travel
  // while this launch combines and stores, and are waited for at the very end
  Chunk16 pf_sink = Chunk16{0u, 0u, 0u, 0u};
  if (p.pf_per_wave > 0) {
    const int L = blockIdx.y * gridDim.x + blockIdx.x, x = L & 7, q = L >> 3;
    const unsigned upt = p.pf_tile_bytes >> 10;                       // 1 KiB units per tile
    const int tiles_x = (p.pf_tiles - x + 7) >> 3;                    // tiles t = x + 8 j of this XCD
    const unsigned total = (unsigned)tiles_x * upt;
    unsigned v = ((unsigned)q * WAVES + (unsigned)wave) * (unsigned)p.pf_per_wave;
    for (int i = 0; i < p.pf_per_wave; ++i, ++v) {
      if (v >= total) break;
      const unsigned j = v / upt, off = v - j * upt;
      const char* src = p.pf_base + (size_t)(x + 8 * j) * p.pf_tile_bytes + (size_t)off * 1024 + lane * 16;
      asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(pf_sink) : "v"(src) : "memory");
    }
  }

  // ---- combine the waves' K slices: every wave parks its partial tile (+ partial row sums of squares) in its own staging
  // region -- nobody else touches it, and its own fragment reads have returned -- then one barrier, then fixed order w = 0 .. WAVES-1
  f32x4* red = (f32x4*)my;                                    // [NFRAG][64 lanes]
  float* ss = (float*)(my + NFRAG * 1024);                    // [MF][16 rows][4 lane groups]
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < MF; ++b) red[(a * MF + b) * 64 + lane] = acc[a][b];
  if (do_norm) {
#pragma unroll
    for (int b = 0; b < MF; ++b) ss[(b * 16 + lr) * 4 + lg] = ssq[b];
  }
  stamp(5);
  __syncthreads();
  stamp(6);

  const bool f32out = p.flags & IG_OUT_F32;
#pragma unroll
  for (int i = 0; i < NFIN; ++i) {
    const int f = wave + i * WAVES;
    if (f >= NFRAG) break;
    const int a = f / MF, b = f - a * MF;
    if (glu && (a & 1)) continue;
    const unsigned char* base = smem + f * 1024 + lane * 16;
    f32x4 part[WAVES];
#pragma unroll
    for (int w = 0; w < WAVES; ++w) part[w] = *(const f32x4*)(base + (size_t)w * p.wave_bytes);
    f32x4 v = part[0];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) v += part[w];
    const int m = m_tile + b * 16 + lr;
    int n0 = n_tile + a * 16 + lg * 4;
    const bool owned = a * 16 + lg * 4 < p.wr;   // (wr is a multiple of 4: a lane's four columns are owned together)
    float rs = 1.0f;
    if (do_norm) {
      const unsigned char* sb = smem + NFRAG * 1024 + (b * 16 + lr) * 16;
      f32x4 sp[WAVES];
#pragma unroll
      for (int w = 0; w < WAVES; ++w) sp[w] = *(const f32x4*)(sb + (size_t)w * p.wave_bytes);
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) tot += (sp[w][0] + sp[w][1]) + (sp[w][2] + sp[w][3]);
      rs = rsqrtf(tot * p.inv_k + p.eps);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= rs;
    }
    int nlim = p.N;
    if (glu) {
      if constexpr (FN >= 2) {
        const unsigned char* ub = base + MF * 1024;           // fragment (a + 1, b)
        f32x4 u = *(const f32x4*)ub;
#pragma unroll
        for (int w = 1; w < WAVES; ++w) u += *(const f32x4*)(ub + (size_t)w * p.wave_bytes);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = silu_t<T>(v[r]) * (u[r] * rs);
      }
      n0 = (n_tile >> 1) + (a >> 1) * 16 + lg * 4;
      nlim = p.N >> 1;
    }
    if (m >= p.M || n0 >= nlim || !owned) continue;
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
          V4 o = res[i];
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

template <typename T, int MF, int FN, int WAVES, bool X3 = false>
static int launch_dg3(const Dg3Dev& d, hipStream_t stream) {
  if constexpr (sizeof(T) == 4 && !X3) {
    if (d.x3) return launch_dg3<T, MF, FN, WAVES, true>(d, stream);
  }
  const int smem = (int)d.wave_bytes * WAVES;
  if (smem > 160 * 1024) return -1;
  static DynLdsOnce once;
  auto kfn = dg3_kernel<T, MF, FN, WAVES, X3>;
  if (hipError_t e = ensure_dyn_lds(once, (const void*)kfn, 160 * 1024); e != hipSuccess) return (int)e;
  dim3 grid((unsigned)cdiv(d.N, d.wr), (unsigned)cdiv(d.M, 16 * MF), 1);
  hipLaunchKernelGGL(kfn, grid, dim3(WAVES * 64), smem, stream, d);
  return (int)hipGetLastError();
}

template <typename T, int WAVES>
static int launch_dg3_w(const Dg3Dev& d, int MF, int FN, hipStream_t st) {
#define IVG_DG3(mf, fn) if (MF == mf && FN == fn) return launch_dg3<T, mf, fn, WAVES>(d, st);
  IVG_DG3(1, 1) IVG_DG3(2, 1) IVG_DG3(4, 1)
  IVG_DG3(1, 2) IVG_DG3(2, 2) IVG_DG3(4, 2)
#undef IVG_DG3
  return -1;
}

// K partition: a function of the K bytes only.  lines = K bytes / 128 over the largest wave count of {16, 12, 8, 4} that divides it.
static int dg3_waves(long lines) {
  for (int w : {16, 12, 8, 4}) if (lines % w == 0) return w;
  return 0;
}

// Measured picks (tools/ubench/dgemm_phase, profiles/r03_dgemm3_sweep.txt) for the GEMMs of the released transformers, keyed by
// (K bytes, N) -- never by the batch.  mf caps the row tiles per workgroup.
struct Dg3Pick { int kbytes, N, mf, fn, wr; };   // wr: rows of W per workgroup (0: the full 16 fn)
static const Dg3Pick kDg3Picks[] = {
    {1536, 2304, 2, 2, 0},    // small: q/k/v      72 N tiles x 2 row halves = 144 workgroups x 98 KB
    {1536, 768, 1, 1, 12},    // small: o-proj      64 x 4 = 256 workgroups, 25 + 18 KB (16 rows: 192 x 49 KB)
    {1536, 6144, 4, 2, 0},    // small: gate/up    the [16 gate | 16 up] packing keeps the 32-row tile
    {6144, 768, 1, 1, 12},    // small: down        64 x 4 = 256 workgroups, 98 + 74 KB (16 rows: 192 x 196 KB)
    {2048, 3072, 2, 2, 0},    // medium: q/k/v
    {2048, 1024, 1, 1, 0},    // medium: o-proj    (on the skip list)
    {2048, 8192, 2, 2, 0},    // medium: gate/up   (on the skip list)
    {8192, 1024, 1, 1, 0},    // medium: down      64 x 4 = 256 workgroups already
};

// GEMMs of the released transformers that measure FASTER on the second-generation kernel inside the layer chain
// (tools/ubench/dgemm_phase with GENMASK, profiles/r03_dgemm_generation_per_gemm.txt: medium 50.98 vs 53.54 us per layer with
// o-proj and gate/up on dgemm.hip; the small transformer's o-proj measured 40.03 vs 40.75 there in the harness but 51.6 vs 49.6 ms
// of decode-GEMM time per step in the engine -- it stays on this kernel): keyed by (K bytes, N) like every other decision
struct Dg3Skip { int kbytes, N; };
static const Dg3Skip kDg3Skip[] = {
    {2048, 1024},    // medium: o-proj
    {2048, 8192},    // medium: gate/up (does not fit one round of workgroups with everything in flight: 16 waves x 12 KiB)
};

struct Dg3Plan { int mf, fn, waves, klw, ring; unsigned wave_bytes; int wr; };

// shape -> launch plan; false: not covered (launch_skinny falls back to dgemm.hip).  Coverage and the K partition
// depend on (K, N, dtype, flags) only; the batch size only picks MF (which rows share a workgroup -- never a sum order).
static bool dg3_plan(const SkinnyArgs& a, DType dtype, Dg3Plan& pl) {
  const int es = dtype == BF16 ? 2 : 4;
  if (a.M <= 0 || a.N <= 0 || a.M > 128) return false;
  if (((long)a.K * es) % 128 != 0 || ((long)a.ldx * es) % 16 != 0 || ((long)a.ldw * es) % 16 != 0) return false;
  if (((uintptr_t)a.X & 15) || ((uintptr_t)a.W & 15)) return false;
  if ((long)a.N * a.ldw * es >= (1L << 31) || (long)a.M * a.ldx * es >= (1L << 31)) return false;   // 32-bit per-lane offsets
  const bool glu = a.flags & IG_GLU;
  if (glu && a.N % 32 != 0) return false;
  if (a.N < 4) return false;
  if ((a.flags & IG_RESIDUAL) && !(a.flags & IG_OUT_F32) && ((a.ldy & 3) != 0 || ((uintptr_t)a.Y & (4 * es - 1)) || (a.N & 3))) return false;
  for (const Dg3Skip& k : kDg3Skip) if (k.kbytes == a.K * es && k.N == a.N) return false;
  const long lines = (long)a.K * es / 128;
  int waves = dg3_waves(lines);
  if (!waves) return false;
  const int mt = cdiv(a.M, 16);
  int MF = mt >= 4 ? 4 : (mt >= 2 ? 2 : 1);
  int FN = (glu || a.N > 16 * 256) ? 2 : 1;                       // (a function of N only: it decides the coverage below)
  {
    auto wgs = [&](int mf, int fn) { return (long)cdiv(a.M, 16 * mf) * cdiv(a.N, 16 * fn); };
    while (MF > 1 && wgs(MF, FN) < 128) MF >>= 1;                 // narrow GEMMs: split the rows to fill the chip
  }
  int WR = 0;
  for (const Dg3Pick& k : kDg3Picks) {
    if (k.kbytes != a.K * es || k.N != a.N) continue;
    MF = std::min(k.mf, mt >= 4 ? 4 : (mt >= 2 ? 2 : 1));
    FN = k.fn;
    if (!glu && k.wr % 4 == 0 && k.wr > 16 * FN - 8 && k.wr <= 16 * FN) WR = k.wr;
    break;
  }
  // one workgroup per CU (the staging fills most of the LDS): a GEMM whose W tiles outnumber the CUs would run in rounds, each
  // paying the full request latency -- lm_head (513 tiles) stays on the second-generation kernel, whose small workgroups are all
  // resident at once.  (A function of N and the pick only, like the rest of the coverage.)
  if (WR == 0) WR = 16 * FN;
  if (cdiv(a.N, WR) > 256) return false;
  const int klw = (int)(lines / waves);
  // everything in flight at once needs waves * (MF + FN) * 2 KiB per line of K; over budget: fewer row tiles per workgroup (never
  // another K partition: MF only picks which rows share a workgroup).  The budget is the whole LDS of a CU unless the caller wants
  // these workgroups to fit BESIDE a capped conv3x3 workgroup of another batch in flight (IVG_DECODE_LDS_KB, switches.h).
  const int budget = (a.lds_kb > 0 ? a.lds_kb : sw().decode_lds_kb) * 1024;
  while (MF > 1 && waves * (MF + FN) * 2048 > budget) MF >>= 1;
  if (waves * (MF + FN) * 2048 > budget) return false;
  const int ring = klw >= 2 && waves * 2 * (MF + FN) * 2048 <= budget ? 2 : 1;
  pl = Dg3Plan{MF, FN, waves, klw, ring, (unsigned)(ring * (MF + FN) * 2048), WR};
  return true;
}

// rows of W one workgroup of the plan for this GEMM owns (the L2 warm-up of a predecessor launch mirrors that tiling); 0: not covered
int dgemm3_w_rows_per_block(const SkinnyArgs& a, DType dtype) {
  Dg3Plan pl;
  return dg3_plan(a, dtype, pl) ? pl.wr : 0;
}

// -1: shape not covered; otherwise a hipError_t
int launch_dgemm3(const SkinnyArgs& a, DType dtype, hipStream_t stream) {
  if (!sw().dg3) return -1;   // IVG_DG3=0: second-generation kernel (A/B runs, tests of the older generations)
  Dg3Plan pl;
  if (!dg3_plan(a, dtype, pl)) return -1;
  const int es = dtype == BF16 ? 2 : 4;
  Dg3Dev d{};
  d.X = a.X; d.W = a.W; d.Y = a.Y; d.M = a.M; d.N = a.N;
  d.ldxb = (unsigned)a.ldx * es; d.ldwb = (unsigned)a.ldw * es; d.ldy = a.ldy;
  d.wr = pl.wr; d.klw = pl.klw; d.ring = pl.ring; d.wave_bytes = pl.wave_bytes; d.flags = a.flags;
  d.inv_k = 1.0f / (float)a.K; d.eps = a.eps; d.bump = a.bump;
  d.w_nt = a.w_shared ? 0 : 1;
  d.x3 = (a.x3 && dtype == F32) ? 1 : 0;
  d.prof = a.pos ? a.prof : nullptr; d.pos = a.pos; d.prof_ld = a.prof_ld;
  d.dbg = a.dbg;
  if (a.next_W && a.next_tile_bytes >= 1024 && a.next_tiles > 0 && sw().dg3_warm) {   // IVG_DG3_WARM=0: no warm-up of the next launch's weights
    const long grid = (long)cdiv(a.N, pl.wr) * cdiv(a.M, 16 * pl.mf);
    const long waves_per_xcd = std::max(1L, grid / 8) * pl.waves;
    const long units_per_xcd = (long)cdiv(a.next_tiles, 8) * (a.next_tile_bytes >> 10);
    long per = (units_per_xcd + waves_per_xcd - 1) / waves_per_xcd;
    if (per > 8) per = 8;   // 1 KiB requests per wave at most
    d.pf_base = (const char*)a.next_W; d.pf_tile_bytes = (unsigned)a.next_tile_bytes; d.pf_tiles = a.next_tiles; d.pf_per_wave = (int)per;
  }
  int rc;
#define IVG_DG3_W(w) if (pl.waves == w) { rc = dtype == BF16 ? launch_dg3_w<bf16_t, w>(d, pl.mf, pl.fn, stream) : launch_dg3_w<float, w>(d, pl.mf, pl.fn, stream); return rc; }
  IVG_DG3_W(16) IVG_DG3_W(12) IVG_DG3_W(8) IVG_DG3_W(4)
#undef IVG_DG3_W
  return -1;
}

}  // namespace ivg
