// Skinny GEMM for the autoregressive decode steps (SURVEY.md 2.4 K14/K17 at L = 1):
//   Y[m][n] = sum_k X[m][k] * W[n][k],   M = batch of trajectories (<= 128), W streamed once from HBM.
//
// HBM-bound weight streaming, so no LDS staging of W: every wave loads its MFMA A-fragments (weights)
// straight from global memory with 16-byte loads (cdna_hip_programming.md, "GEMV / M <= 16 decode
// weights: load straight to VGPRs"), the small activation matrix is re-read through L1/L2.
// A workgroup = 4 waves that share one 16*FN-row slice of W and split its K range four ways; partial
// accumulators are combined through LDS in a fixed order (deterministic).  grid.y = split-K slices
// whose fp32 partials are summed, again in fixed order, by the consumer kernel (add_rmsnorm).
#include "igemm.h"

namespace ivg {

struct SkinnyDev {
  const void* X; const void* W; void* Y;
  int M, N, K, ldx, ldw, ldy, splits, flags;
};

template <typename T, int MF, int FN>
__global__ __launch_bounds__(256) void skinny_kernel(const SkinnyDev p) {
  constexpr int VEC = Traits<T>::VEC;
  constexpr int KSTEP = 4 * VEC;  // 4 lane groups x one 16-byte chunk
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  f32x4* red = (f32x4*)smem;  // [4 waves][FN][MF][64 lanes]

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int lr = lane & 15, lg = lane >> 4;
  const int n_tile = blockIdx.x * (16 * FN);
  const int s = blockIdx.y;
  const int ks = p.K / p.splits, kw = ks / 4;
  const int kbeg = s * ks + wave * kw, kend = kbeg + kw;
  const T* X = (const T*)p.X;
  const T* W = (const T*)p.W;

  f32x4 acc[FN][MF];
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < MF; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

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
    const int m = b * 16 + lr;
    xok[b] = m < p.M;
    xoff[b] = (long)(xok[b] ? m : 0) * p.ldx + lg * VEC;
  }

  // bursts of S K-steps: every load of a burst is issued before the first MFMA (the step is latency-bound, so
  // memory-level parallelism matters more than anything else here)
  constexpr int S = MF >= 8 ? 2 : 4;
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

  // ---- combine the 4 waves' K slices (fixed order w = 0..3)
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < MF; ++b) red[((wave * FN + a) * MF + b) * 64 + lane] = acc[a][b];
  __syncthreads();

  const bool glu = p.flags & IG_GLU;
  const bool f32out = (p.flags & IG_OUT_F32) || p.splits > 1;
  for (int f = wave; f < FN * MF; f += 4) {
    const int a = f / MF, b = f - a * MF;
    if (glu && a == 1) continue;
    f32x4 v = red[((0 * FN + a) * MF + b) * 64 + lane];
#pragma unroll
    for (int w = 1; w < 4; ++w) v += red[((w * FN + a) * MF + b) * 64 + lane];
    const int m = b * 16 + lr;
    int n0 = n_tile + a * 16 + lg * 4;
    if (m >= p.M || n0 >= p.N) continue;
    int nlim = p.N;
    if (glu) {
      if constexpr (FN == 2) {
        f32x4 u = red[((0 * FN + 1) * MF + b) * 64 + lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) u += red[((w * FN + 1) * MF + b) * 64 + lane];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]) * u[r];
      }
      n0 = (n_tile >> 1) + lg * 4;
      nlim = p.N >> 1;
    }
    if (f32out) {
      float* Y = (float*)p.Y + ((long)s * p.M + m) * (p.splits > 1 ? p.N : p.ldy) + n0;
#pragma unroll
      for (int r = 0; r < 4; ++r) if (n0 + r < nlim) Y[r] = v[r];
    } else {
      T* Y = (T*)p.Y + (long)m * p.ldy + n0;
#pragma unroll
      for (int r = 0; r < 4; ++r) if (n0 + r < nlim) Y[r] = from_f32<T>(v[r]);
    }
  }
}

template <typename T, int MF, int FN>
static int launch_sk(const SkinnyDev& d, hipStream_t stream) {
  constexpr int smem = 4 * FN * MF * 64 * 16;
  static bool attr_set = false;
  auto kfn = skinny_kernel<T, MF, FN>;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  dim3 grid((unsigned)cdiv(d.N, 16 * FN), (unsigned)d.splits, 1);
  hipLaunchKernelGGL(kfn, grid, dim3(256), smem, stream, d);
  return (int)hipGetLastError();
}

template <typename T, int FN>
static int launch_sk_m(const SkinnyDev& d, hipStream_t stream) {
  if (d.M <= 16) return launch_sk<T, 1, FN>(d, stream);
  if (d.M <= 32) return launch_sk<T, 2, FN>(d, stream);
  if (d.M <= 64) return launch_sk<T, 4, FN>(d, stream);
  return launch_sk<T, 8, FN>(d, stream);
}

int skinny_pick_splits(int N, int K, DType dtype) {
  // enough workgroups to fill 256 CUs; each wave's K slice must be a multiple of one MFMA K-step
  const int kstep = (dtype == BF16) ? 32 : 16;
  const int tiles = cdiv(N, 16);
  int best = 1;
  for (int s = 1; s <= 16; ++s) {
    if (K % (s * 4 * kstep) != 0) continue;
    if (K / (s * 4) < 2 * kstep && s > 1) break;
    best = s;
    if ((long)tiles * s >= 256) break;
  }
  return best;
}

int launch_skinny(const SkinnyArgs& a, DType dtype, hipStream_t stream) {
  SkinnyDev d{a.X, a.W, a.Y, a.M, a.N, a.K, a.ldx, a.ldw, a.ldy, a.splits < 1 ? 1 : a.splits, a.flags};
  const int kstep = (dtype == BF16) ? 32 : 16, vec = (dtype == BF16) ? 8 : 4;
  if (a.M <= 0 || a.N <= 0) return 0;
  if (a.M > 128 || a.K % (d.splits * 4 * kstep) != 0 || a.ldx % vec != 0 || a.ldw % vec != 0)
    return (int)hipErrorInvalidValue;
  const bool glu = a.flags & IG_GLU;
  if (glu && (a.N % 32 != 0 || d.splits != 1)) return (int)hipErrorInvalidValue;
  if (dtype == BF16) return glu ? launch_sk_m<bf16_t, 2>(d, stream) : launch_sk_m<bf16_t, 1>(d, stream);
  return glu ? launch_sk_m<float, 2>(d, stream) : launch_sk_m<float, 1>(d, stream);
}

}  // namespace ivg
