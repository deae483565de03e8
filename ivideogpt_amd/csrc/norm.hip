// Normalisation / softmax kernels (HBM-bound, vectorised 16-byte accesses, wave64 reductions).
//   gn_partial / gn_apply : GroupNorm(32 groups) [+ SiLU] [+ position embedding] on NHWC tensors
//                           (SURVEY.md 2.4 K4, and the q/kv norms of K8).  Deterministic two-stage
//                           statistics: per-chunk fp32 sums -> fp64 combine in fixed order.
//   add_rmsnorm           : residual-stream update (sum of split-K partials) + Llama RMSNorm (K13).
//   softmax_rows          : row softmax of fp32 scores with optional causal mask (K7/K8/K16).
#include "ops.h"

namespace ivg {

// ------------------------------------------------------------------------------------------------ GroupNorm
// X [N][P][C] (C contiguous).  partial[n][chunk][g] = (sum, sumsq) over the chunk's pixels, fp64.
template <typename T>
__global__ __launch_bounds__(256) void gn_partial_kernel(const T* __restrict__ X, double2* __restrict__ part, int P, int C,
                                                         int groups, int chunk_px) {
  constexpr int VEC = Traits<T>::VEC;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* s_sum = (float*)smem;  // [PL][C]
  const int vpp = C / VEC;
  const int pl_n = 256 / vpp;  // pixel lanes
  const int tid = threadIdx.x;
  const int n = blockIdx.y, chunk = blockIdx.x;
  const int p0 = chunk * chunk_px, p1 = min(P, p0 + chunk_px);
  float* s_sq = s_sum + pl_n * C;
  float sum[VEC], sq[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { sum[j] = 0.f; sq[j] = 0.f; }
  const int v = tid % vpp, pl = tid / vpp;
  if (pl < pl_n) {
    const T* base = X + ((long)n * P) * C + v * VEC;
    for (int p = p0 + pl; p < p1; p += pl_n) {
      const Chunk16 raw = *(const Chunk16*)(base + (long)p * C);
      if constexpr (sizeof(T) == 2) {
        const bf16x8 x = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
        for (int j = 0; j < VEC; ++j) { const float f = (float)x[j]; sum[j] += f; sq[j] = fmaf(f, f, sq[j]); }
      } else {
        const f32x4 x = __builtin_bit_cast(f32x4, raw);
#pragma unroll
        for (int j = 0; j < VEC; ++j) { const float f = x[j]; sum[j] += f; sq[j] = fmaf(f, f, sq[j]); }
      }
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) { s_sum[pl * C + v * VEC + j] = sum[j]; s_sq[pl * C + v * VEC + j] = sq[j]; }
  }
  __syncthreads();
  if (tid < groups) {
    const int cpg = C / groups;
    double a = 0.0, b = 0.0;
    for (int c = tid * cpg; c < (tid + 1) * cpg; ++c)
      for (int q = 0; q < pl_n; ++q) { a += (double)s_sum[q * C + c]; b += (double)s_sq[q * C + c]; }
    part[((long)n * gridDim.x + chunk) * groups + tid] = double2{a, b};
  }
}

// Y = act(GN(X)) (+ pos[p][c]);  statistics from `part` (nchunks chunks of image n).
template <typename T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const T* __restrict__ X, T* __restrict__ Y,
                                                       const double2* __restrict__ part, int nchunks,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       const float* __restrict__ pos, int P, int C, int groups, float eps,
                                                       int silu, int px_per_block) {
  constexpr int VEC = Traits<T>::VEC;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  f32x2* coef = (f32x2*)smem;  // [C] (scale, shift)
  const int n = blockIdx.y, tid = threadIdx.x;
  const int cpg = C / groups;
  for (int c = tid; c < C; c += 256) {
    const int g = c / cpg;
    double a = 0.0, b = 0.0;
    for (int q = 0; q < nchunks; ++q) { const double2 t = part[((long)n * nchunks + q) * groups + g]; a += t.x; b += t.y; }
    const double cnt = (double)P * cpg;
    const double mean = a / cnt;
    double var = b / cnt - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * rstd;
    coef[c] = f32x2{sc, beta[c] - (float)mean * sc};
  }
  __syncthreads();
  const int vpp = C / VEC;
  const int p0 = blockIdx.x * px_per_block, p1 = min(P, p0 + px_per_block);
  const long nvec = (long)(p1 - p0) * vpp;
  const T* xb = X + ((long)n * P + p0) * C;
  T* yb = Y + ((long)n * P + p0) * C;
  for (long i = tid; i < nvec; i += 256) {
    const int c0 = (int)(i % vpp) * VEC;
    const int p = p0 + (int)(i / vpp);
    const Chunk16 raw = *(const Chunk16*)(xb + i * VEC);
    float f[VEC];
    if constexpr (sizeof(T) == 2) {
      const bf16x8 x = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
      for (int j = 0; j < VEC; ++j) f[j] = (float)x[j];
    } else {
      const f32x4 x = __builtin_bit_cast(f32x4, raw);
#pragma unroll
      for (int j = 0; j < VEC; ++j) f[j] = x[j];
    }
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const f32x2 cf = coef[c0 + j];
      float y = fmaf(f[j], cf[0], cf[1]);
      if (silu) y = silu_t<T>(y);
      if (pos) y += pos[(long)p * C + c0 + j];
      f[j] = y;
    }
    if constexpr (sizeof(T) == 2) {
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < VEC; ++j) o[j] = (bf16_t)f[j];
      *(bf16x8*)(yb + i * VEC) = o;
    } else {
      *(f32x4*)(yb + i * VEC) = f32x4{f[0], f[1], f[2], f[3]};
    }
  }
}

static int gn_chunk_px(int P) { return P > 1024 ? 1024 : P; }

int gn_num_chunks(int P) { return cdiv(P, gn_chunk_px(P)); }

int launch_groupnorm(const void* X, void* Y, void* part_ws, const float* gamma, const float* beta, const float* pos,
                     int N, int P, int C, int groups, float eps, int silu, DType dt, hipStream_t st) {
  const int vec = dt == BF16 ? 8 : 4;
  if (C % vec != 0 || C % groups != 0 || C / vec > 256 || groups > 256) return (int)hipErrorInvalidValue;
  const int chunk_px = gn_chunk_px(P), nchunks = cdiv(P, chunk_px);
  const int pl_n = 256 / (C / vec);
  const size_t smem1 = (size_t)2 * pl_n * C * sizeof(float);
  dim3 g1(nchunks, N);
  const int px_per_block = 256;
  dim3 g2(cdiv(P, px_per_block), N);
  const size_t smem2 = (size_t)C * sizeof(f32x2);
  if (dt == BF16) {
    hipLaunchKernelGGL(gn_partial_kernel<bf16_t>, g1, dim3(256), smem1, st, (const bf16_t*)X, (double2*)part_ws, P, C, groups, chunk_px);
    hipLaunchKernelGGL(gn_apply_kernel<bf16_t>, g2, dim3(256), smem2, st, (const bf16_t*)X, (bf16_t*)Y, (const double2*)part_ws,
                       nchunks, gamma, beta, pos, P, C, groups, eps, silu, px_per_block);
  } else {
    hipLaunchKernelGGL(gn_partial_kernel<float>, g1, dim3(256), smem1, st, (const float*)X, (double2*)part_ws, P, C, groups, chunk_px);
    hipLaunchKernelGGL(gn_apply_kernel<float>, g2, dim3(256), smem2, st, (const float*)X, (float*)Y, (const double2*)part_ws,
                       nchunks, gamma, beta, pos, P, C, groups, eps, silu, px_per_block);
  }
  return (int)hipGetLastError();
}

// (scale, shift) per (image, channel) from the reduced statistics: y = x * scale + shift is GroupNorm's affine output.  The
// conv3x3 kernel applies it (+ SiLU) to its input while staging it (IgemmArgs::gn_in_coef).  Same arithmetic as gn_apply.
__global__ __launch_bounds__(256) void gn_coef_kernel(const double2* __restrict__ part, int nchunks, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, int P, int C, int groups, float eps,
                                                      f32x2* __restrict__ coef) {
  const int n = blockIdx.x, cpg = C / groups;
  for (int c = threadIdx.x; c < C; c += 256) {
    const int g = c / cpg;
    double a = 0.0, b = 0.0;
    for (int q = 0; q < nchunks; ++q) { const double2 t = part[((long)n * nchunks + q) * groups + g]; a += t.x; b += t.y; }
    const double cnt = (double)P * cpg;
    const double mean = a / cnt;
    double var = b / cnt - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float sc = gamma[c] * rstd;
    coef[(long)n * C + c] = f32x2{sc, beta[c] - (float)mean * sc};
  }
}

int launch_gn_coef(const void* part, int nchunks, const float* gamma, const float* beta, int N, int P, int C, int groups, float eps, void* coef,
                   hipStream_t st) {
  if (N <= 0 || nchunks <= 0 || C % groups != 0) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(gn_coef_kernel, dim3((unsigned)N), dim3(256), 0, st, (const double2*)part, nchunks, gamma, beta, P, C, groups, eps, (f32x2*)coef);
  return (int)hipGetLastError();
}

// statistics only (the first half of launch_groupnorm): double2 [N][gn_num_chunks(P)][groups]
int launch_groupnorm_partial(const void* X, void* part_ws, int N, int P, int C, int groups, DType dt, hipStream_t st) {
  const int vec = dt == BF16 ? 8 : 4;
  if (C % vec != 0 || C % groups != 0 || C / vec > 256 || groups > 256) return (int)hipErrorInvalidValue;
  const int chunk_px = gn_chunk_px(P), nchunks = cdiv(P, chunk_px);
  const int pl_n = 256 / (C / vec);
  const size_t smem1 = (size_t)2 * pl_n * C * sizeof(float);
  dim3 g1(nchunks, N);
  if (dt == BF16) hipLaunchKernelGGL(gn_partial_kernel<bf16_t>, g1, dim3(256), smem1, st, (const bf16_t*)X, (double2*)part_ws, P, C, groups, chunk_px);
  else hipLaunchKernelGGL(gn_partial_kernel<float>, g1, dim3(256), smem1, st, (const float*)X, (double2*)part_ws, P, C, groups, chunk_px);
  return (int)hipGetLastError();
}

// apply only: the statistics were produced elsewhere (conv3x3 epilogue) as double2 [N][nchunks][groups]
int launch_groupnorm_apply(const void* X, void* Y, const void* part, int nchunks, const float* gamma, const float* beta, const float* pos,
                           int N, int P, int C, int groups, float eps, int silu, DType dt, hipStream_t st) {
  const int vec = dt == BF16 ? 8 : 4;
  if (C % vec != 0 || C % groups != 0 || C / vec > 256 || groups > 256 || nchunks <= 0) return (int)hipErrorInvalidValue;
  const int px_per_block = 256;
  dim3 g2(cdiv(P, px_per_block), N);
  const size_t smem2 = (size_t)C * sizeof(f32x2);
  if (dt == BF16)
    hipLaunchKernelGGL(gn_apply_kernel<bf16_t>, g2, dim3(256), smem2, st, (const bf16_t*)X, (bf16_t*)Y, (const double2*)part, nchunks, gamma, beta,
                       pos, P, C, groups, eps, silu, px_per_block);
  else
    hipLaunchKernelGGL(gn_apply_kernel<float>, g2, dim3(256), smem2, st, (const float*)X, (float*)Y, (const double2*)part, nchunks, gamma, beta, pos,
                       P, C, groups, eps, silu, px_per_block);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ RMSNorm
// One workgroup per row (H <= 2048), every thread owns up to 8 strided elements so all loads of a pass are
// independent (the decode step is latency-bound: M = batch rows only).  x (T) is updated in place when split-K
// partials are given (fixed order s = 0..S-1), out = rmsnorm(x) * w.
template <typename T>
__global__ __launch_bounds__(256) void add_rmsnorm_kernel(T* __restrict__ x, long xs, const float* __restrict__ w, T* __restrict__ out,
                                                          int M, int H, float eps) {
  __shared__ float red[4];
  const int row = blockIdx.x, tid = threadIdx.x;
  T* xr = x + (long)row * xs;
  float f[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { const int c = tid + 256 * i; f[i] = c < H ? to_f32(xr[c]) : 0.f; }
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) ss = fmaf(f[i], f[i], ss);
  ss = wave_sum(ss);
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  __syncthreads();
  ss = (red[0] + red[1]) + (red[2] + red[3]);
  const float inv = rsqrtf(ss / (float)H + eps);
  if (out) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = tid + 256 * i;
      if (c < H) {
        const float nrm = to_f32(from_f32<T>(f[i] * inv));  // HF casts back to the input dtype before * weight
        out[(long)row * H + c] = from_f32<T>(w[c] * nrm);
      }
    }
  }
}

int launch_add_rmsnorm(void* x, long x_stride, const float* w, void* out, int M, int H, float eps, DType dt, hipStream_t st) {
  if (H > 2048) return (int)hipErrorInvalidValue;
  if (M <= 0) return 0;
  dim3 g(M);
  if (dt == BF16)
    hipLaunchKernelGGL(add_rmsnorm_kernel<bf16_t>, g, dim3(256), 0, st, (bf16_t*)x, x_stride, w, (bf16_t*)out, M, H, eps);
  else
    hipLaunchKernelGGL(add_rmsnorm_kernel<float>, g, dim3(256), 0, st, (float*)x, x_stride, w, (float*)out, M, H, eps);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ softmax
// S fp32 [rows][lds] -> Pm (T) [rows][ldp].  Row r belongs to query q = r % Lq; causal: keys j <= q + (Lk - Lq).
// Columns [Lk, ldp) of Pm are written as zero (the P.V GEMM runs over the padded K).
template <typename T>
__global__ __launch_bounds__(256) void softmax_kernel(const float* __restrict__ S, T* __restrict__ Pm, int Lq, int Lk, int lds,
                                                      int ldp, int causal) {
  __shared__ float red[8];
  const long r = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int q = (int)(r % Lq);
  const int lim = causal ? min(Lk, q + (Lk - Lq) + 1) : Lk;
  const float* s = S + r * lds;
  float v[8];  // Lk <= 2048
  float mx = -INFINITY;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int j = tid + i * 256;
    v[i] = (j < lim) ? s[j] : -INFINITY;
    mx = fmaxf(mx, v[i]);
  }
  mx = wave_max(mx);
  if (lane == 0) red[wv] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int j = tid + i * 256;
    v[i] = (j < lim) ? expf(v[i] - mx) : 0.f;
    sum += v[i];
  }
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wv] = sum;
  __syncthreads();
  sum = (red[4] + red[5]) + (red[6] + red[7]);
  const float inv = 1.0f / sum;
  T* o = Pm + r * ldp;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int j = tid + i * 256;
    if (j < ldp) o[j] = from_f32<T>(v[i] * inv);
  }
}

int launch_softmax(const float* S, void* Pm, long rows, int Lq, int Lk, int lds, int ldp, int causal, DType dt, hipStream_t st) {
  if (Lk > 2048 || rows <= 0) return rows <= 0 ? 0 : (int)hipErrorInvalidValue;
  dim3 g((unsigned)rows);
  if (dt == BF16)
    hipLaunchKernelGGL(softmax_kernel<bf16_t>, g, dim3(256), 0, st, S, (bf16_t*)Pm, Lq, Lk, lds, ldp, causal);
  else
    hipLaunchKernelGGL(softmax_kernel<float>, g, dim3(256), 0, st, S, (float*)Pm, Lq, Lk, lds, ldp, causal);
  return (int)hipGetLastError();
}

}  // namespace ivg
