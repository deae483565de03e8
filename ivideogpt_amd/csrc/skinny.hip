// Skinny GEMM for the autoregressive decode steps (SURVEY.md 2.4 K14/K17 at L = 1):
//   Y[m][n] = sum_k X[m][k] * W[n][k],   M = batch of trajectories (<= 128), W streamed once from HBM.
//
// Measured regime on MI355X (profiles/): a step's GEMMs are bound by what ONE CU can ingest
// (~10 B/clk ~ 25 GB/s per CU for anything that misses its L1, weights from HBM and the activation matrix from
// the other XCDs alike) plus a fixed ~3 us per launch.  So the decomposition minimises bytes per CU:
//   * a workgroup owns a (16*MF rows of X) x (16*FN rows of W) output tile over the FULL K; its WAVES waves split
//     K and combine through LDS in a fixed order (deterministic);  the tile shape is picked per GEMM by a small
//     cost model (launch_skinny): narrow GEMMs split the batch rows across workgroups instead of re-reading all of
//     X in every workgroup, wide ones (lm_head) take fat W tiles so X is amortised;
//   * weights go straight from global memory into MFMA A-fragments (no LDS staging: every byte is used once),
//     all loads of a burst are issued before the first MFMA;
//   * epilogues fuse what would otherwise be extra latency-bound launches: RMSNorm of the input rows (row scale
//     from the activations the wave streams anyway; norm weight pre-folded into W), in-place residual add,
//     SiLU(gate)*up, and the advance of the device-side step counter.
#include <cstdio>
#include <cstdlib>

#include "igemm.h"

namespace ivg {

struct SkinnyDev {
  const void* X; const void* W; void* Y;
  int M, N, K, ldx, ldw, ldy, splits, flags;
  float eps;       // SK_NORM: y = rsqrt(mean_k x^2 + eps) * (x . W)   (RMSNorm weight pre-folded into W)
  int* bump;       // optional: two ints incremented by one thread at the end (device-side step state)
  int tail_split;  // the ragged last column tile is computed by one extra workgroup PER ROW TILE (see launch_sk)
};

template <typename T, int MF, int FN, int WAVES>
__global__ __launch_bounds__(WAVES * 64) void skinny_kernel(const SkinnyDev p) {
  constexpr int VEC = Traits<T>::VEC;
  constexpr int KSTEP = 4 * VEC;  // 4 lane groups x one 16-byte chunk
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  f32x4* red = (f32x4*)smem;  // [WAVES][FN][MF][64 lanes]

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int lr = lane & 15, lg = lane >> 4;
  // tail_split: N = q * (16 FN) + r.  Workgroups [0, q) own the full column tiles; the r ragged columns go to MF extra
  // workgroups, one per 16-row tile, instead of one workgroup that would pull all of X for a sliver of W (lm_head:
  // 16386 = 256 * 64 + 2 -> the 257th workgroup doubled the kernel's time on 256 CUs)
  const int main_tiles = p.tail_split ? p.N / (16 * FN) : (int)gridDim.x;
  const bool tail = (int)blockIdx.x >= main_tiles;
  const int tail_row = tail ? (int)blockIdx.x - main_tiles : -1;
  const int n_tile = (tail ? main_tiles : (int)blockIdx.x) * (16 * FN);
  const int m_tile = blockIdx.y * (16 * MF);
  const int s = blockIdx.z;
  const int ks = p.K / p.splits, kw = ks / WAVES;
  const int kbeg = s * ks + wave * kw, kend = kbeg + kw;
  const T* X = (const T*)p.X;
  const T* W = (const T*)p.W;

  f32x4 acc[FN][MF];
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < MF; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_norm = p.flags & SK_NORM;
  float ssq[MF];
#pragma unroll
  for (int b = 0; b < MF; ++b) ssq[b] = 0.f;

  long woff[FN];
  bool wok[FN];
#pragma unroll
  for (int a = 0; a < FN; ++a) {
    const int n = n_tile + a * 16 + lr;
    wok[a] = n < p.N;
    woff[a] = (long)(wok[a] ? n : 0) * p.ldw + lg * VEC;
  }
  long xoff[MF];
  bool xok[MF];
#pragma unroll
  for (int b = 0; b < MF; ++b) {
    const int m = m_tile + b * 16 + lr;
    xok[b] = m < p.M && (!tail || b == tail_row);
    xoff[b] = (long)(xok[b] ? m : 0) * p.ldx + lg * VEC;
  }

  // bursts of S K-steps: every load of a burst is issued before the first MFMA
  constexpr int OPS = FN + MF;  // 16-byte operand chunks per K-step per lane (4 VGPRs each)
  constexpr int S = WAVES >= 16 ? (OPS > 6 ? 2 : 4) : (OPS > 6 ? 2 : 6);  // VGPR budget (1024-thread blocks: 128 / lane)
  for (int k = kbeg; k < kend; k += KSTEP * S) {
    Chunk16 wv[S][FN], xv[S][MF];
#pragma unroll
    for (int t = 0; t < S; ++t) {
      const int kk = k + t * KSTEP;
      const bool in = kk < kend;
#pragma unroll
      for (int a = 0; a < FN; ++a) wv[t][a] = (in && wok[a]) ? *(const Chunk16*)(W + woff[a] + kk) : Chunk16{0u, 0u, 0u, 0u};
#pragma unroll
      for (int b = 0; b < MF; ++b) xv[t][b] = (in && xok[b]) ? *(const Chunk16*)(X + xoff[b] + kk) : Chunk16{0u, 0u, 0u, 0u};
    }
    if (do_norm) {  // row sums of squares of the activations this wave streams anyway (its K slice)
#pragma unroll
      for (int t = 0; t < S; ++t)
#pragma unroll
        for (int b = 0; b < MF; ++b) {
          if constexpr (sizeof(T) == 2) {
            const bf16x8 xx = __builtin_bit_cast(bf16x8, xv[t][b]);
#pragma unroll
            for (int u = 0; u < 8; ++u) { const float f = (float)xx[u]; ssq[b] = fmaf(f, f, ssq[b]); }
          } else {
            const f32x4 xx = __builtin_bit_cast(f32x4, xv[t][b]);
#pragma unroll
            for (int u = 0; u < 4; ++u) ssq[b] = fmaf(xx[u], xx[u], ssq[b]);
          }
        }
    }
#pragma unroll
    for (int t = 0; t < S; ++t)
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < MF; ++b) {
          if constexpr (sizeof(T) == 2) {
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wv[t][a]),
                                                                __builtin_bit_cast(bf16x8, xv[t][b]), acc[a][b], 0, 0, 0);
          } else {
            const f32x4 wf = __builtin_bit_cast(f32x4, wv[t][a]), xf = __builtin_bit_cast(f32x4, xv[t][b]);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u], xf[u], acc[a][b], 0, 0, 0);
          }
        }
  }

  // ---- combine the waves' K slices (fixed order w = 0..WAVES-1)
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < MF; ++b) red[((wave * FN + a) * MF + b) * 64 + lane] = acc[a][b];
  float* s_ss = (float*)(red + WAVES * FN * MF * 64);  // [WAVES][MF*16 rows]
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

  const bool glu = p.flags & IG_GLU;
  const bool f32out = (p.flags & IG_OUT_F32) || p.splits > 1;
  for (int f = wave; f < FN * MF; f += WAVES) {
    const int a = f / MF, b = f - a * MF;
    if (glu && (a & 1)) continue;
    f32x4 v = red[((0 * FN + a) * MF + b) * 64 + lane];
#pragma unroll
    for (int w = 1; w < WAVES; ++w) v += red[((w * FN + a) * MF + b) * 64 + lane];
    const int m = m_tile + b * 16 + lr;
    int n0 = n_tile + a * 16 + lg * 4;
    if (m >= p.M || n0 >= p.N || (tail && b != tail_row)) continue;
    int nlim = p.N;
    float rs = 1.0f;
    if (do_norm) {
      float tot = 0.f;
#pragma unroll
      for (int w = 0; w < WAVES; ++w) tot += s_ss[w * (MF * 16) + b * 16 + lr];
      rs = rsqrtf(tot / (float)p.K + p.eps);
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] *= rs;
    }
    if (glu) {
      if constexpr (FN >= 2) {
        const int a1 = a + 1 < FN ? a + 1 : a;
        f32x4 u = red[((0 * FN + a1) * MF + b) * 64 + lane];
#pragma unroll
        for (int w = 1; w < WAVES; ++w) u += red[((w * FN + a1) * MF + b) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = silu_t<T>(v[r]) * (u[r] * rs);
      }
      n0 = (n_tile >> 1) + (a >> 1) * 16 + lg * 4;
      nlim = p.N >> 1;
    }
    if (f32out) {
      float* Y = (float*)p.Y + ((long)s * p.M + m) * (p.splits > 1 ? p.N : p.ldy) + n0;
#pragma unroll
      for (int r = 0; r < 4; ++r) if (n0 + r < nlim) Y[r] = v[r];
    } else {
      T* Y = (T*)p.Y + (long)m * p.ldy + n0;
      if (p.flags & IG_RESIDUAL) {  // in-place residual-stream update: each element is read and written by one thread
#pragma unroll
        for (int r = 0; r < 4; ++r) if (n0 + r < nlim) Y[r] = from_f32<T>(to_f32(Y[r]) + v[r]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (n0 + r < nlim) Y[r] = from_f32<T>(v[r]);
      }
    }
  }
  if (p.bump && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && tid == 0) { p.bump[0] += 1; p.bump[1] += 1; }
}

template <typename T, int MF, int FN, int WAVES>
static int launch_sk(const SkinnyDev& d, hipStream_t stream) {
  constexpr int smem = WAVES * FN * MF * 64 * 16 + WAVES * MF * 16 * 4;
  static DynLdsOnce once;
  auto kfn = skinny_kernel<T, MF, FN, WAVES>;
  if (hipError_t e = ensure_dyn_lds(once, (const void*)kfn, smem); e != hipSuccess) return (int)e;
  SkinnyDev dd = d;
  dd.tail_split = (MF > 1 && d.M <= 16 * MF && d.splits == 1 && d.N % (16 * FN) != 0 && d.N > 16 * FN && !(d.flags & IG_GLU)) ? 1 : 0;
  const unsigned gx = dd.tail_split ? (unsigned)(d.N / (16 * FN) + cdiv(d.M, 16)) : (unsigned)cdiv(d.N, 16 * FN);
  dim3 grid(gx, (unsigned)cdiv(d.M, 16 * MF), (unsigned)d.splits);
  hipLaunchKernelGGL(kfn, grid, dim3(WAVES * 64), smem, stream, dd);
  return (int)hipGetLastError();
}

// waves per workgroup: as many K slices as divide K into whole MFMA K-steps (one burst of loads per wave)
static int pick_waves(int K, int splits, DType dt) {
  const int kstep = (dt == BF16) ? 32 : 16;
  const int ks = K / splits;
  if (ks >= 2048 && ks % (16 * kstep) == 0) return 16;
  if (ks % (8 * kstep) == 0) return 8;
  return 4;
}

// (MF, FN) by a per-CU ingest model: a workgroup pulls 16*(MF+FN) rows of K elements; workgroups beyond the 256
// CUs queue behind each other.  Lower bound on the kernel's time ~ ceil(workgroups / 256) * bytes per workgroup.
static void pick_tile(int M, int N, int K, bool glu, int& MF, int& FN) {
  const int mf_all = cdiv(M, 16);
  const int mf_opts[4] = {1, 2, 4, 8};
  const int fn_opts[3] = {1, 2, 4};
  double best = 1e30;
  MF = mf_all <= 1 ? 1 : (mf_all <= 2 ? 2 : (mf_all <= 4 ? 4 : 8));
  FN = glu ? 2 : 1;
  for (int mi = 0; mi < 4; ++mi)
    for (int fi = 0; fi < 3; ++fi) {
      const int mf = mf_opts[mi], fn = fn_opts[fi];
      if (mf > 1 && 16 * (mf / 2) >= M) continue;      // tile twice the batch: pure waste
      if (glu && fn < 2) continue;
      if (mf * fn > 16 || (mf + fn) > 10) continue;     // accumulator / operand register budget
      const long wgs = (long)cdiv(M, 16 * mf) * cdiv(N, 16 * fn);
      const double per = 16.0 * (mf + fn) * K;
      const double rounds = (double)((wgs + 255) / 256);
      const double cost = rounds * per + 0.02 * per;   // tie-break towards fewer bytes per workgroup
      if (cost < best) { best = cost; MF = mf; FN = fn; }
    }
}

template <typename T, int MF, int FN>
static int launch_sk_w(const SkinnyDev& d, hipStream_t stream) {
  const int w = pick_waves(d.K, d.splits, Traits<T>::dtype);
  if constexpr (MF + FN <= 5 && MF * FN <= 4) {  // 1024-thread workgroups only where the 128-VGPR budget holds the burst
    if (w == 16) return launch_sk<T, MF, FN, 16>(d, stream);
  }
  if (w >= 8) return launch_sk<T, MF, FN, 8>(d, stream);
  return launch_sk<T, MF, FN, 4>(d, stream);
}

template <typename T>
static int launch_sk_t(const SkinnyDev& d, int MF, int FN, hipStream_t stream) {
#define IVG_SK(mf, fn) if (MF == mf && FN == fn) return launch_sk_w<T, mf, fn>(d, stream)
  IVG_SK(1, 1); IVG_SK(1, 2); IVG_SK(1, 4);
  IVG_SK(2, 1); IVG_SK(2, 2); IVG_SK(2, 4);
  IVG_SK(4, 1); IVG_SK(4, 2); IVG_SK(4, 4);
  IVG_SK(8, 1); IVG_SK(8, 2);
#undef IVG_SK
  return (int)hipErrorInvalidValue;
}

int launch_skinny(const SkinnyArgs& a, DType dtype, hipStream_t stream) {
  SkinnyDev d{a.X, a.W, a.Y, a.M, a.N, a.K, a.ldx, a.ldw, a.ldy, a.splits < 1 ? 1 : a.splits, a.flags, a.eps, a.bump, 0};
  const int kstep = (dtype == BF16) ? 32 : 16, vec = (dtype == BF16) ? 8 : 4;
  if (a.M <= 0 || a.N <= 0) return 0;
  if (a.M > 128 || a.K % (d.splits * 4 * kstep) != 0 || a.ldx % vec != 0 || a.ldw % vec != 0)
    return (int)hipErrorInvalidValue;
  const bool glu = a.flags & IG_GLU;
  if (glu && (a.N % 32 != 0 || d.splits != 1)) return (int)hipErrorInvalidValue;
  if ((a.flags & (SK_NORM | IG_RESIDUAL)) && d.splits != 1) return (int)hipErrorInvalidValue;
  {  // third-generation kernel (dgemm3.hip), then the second (dgemm.hip), wherever they cover the shape -- coverage is a function
     // of (K, N, dtype, flags) only, so a GEMM of the model always runs on the same kernel whatever the batch
    int rc = launch_dgemm3(a, dtype, stream);
    if (rc != -1) return rc;
    rc = launch_dgemm(a, dtype, stream);
    if (rc != -1) return rc;
  }
  int MF, FN;
  pick_tile(a.M, a.N, a.K, glu, MF, FN);
  if (MF == 8 && FN == 4) FN = 2;
  return dtype == BF16 ? launch_sk_t<bf16_t>(d, MF, FN, stream) : launch_sk_t<float>(d, MF, FN, stream);
}

}  // namespace ivg
