// Vector-quantisation kernels.
//
// vq_argmin restates diffusers' VectorQuantizer (called at compressive_vq_model.py:199,202):
//   idx = argmin_j cdist(z, E)[., j],  cdist in its fp32 GEMM form
//   d_j = sqrt(max(||z||^2 + ||e_j||^2 - 2 z.e_j, 0)),  lowest index among equal minima.
// The dot products run on the exact-fp32 matrix cores (v_mfma_f32_16x16x4_f32 = an fp32 fma chain),
// z fragments stay in registers for the whole codebook sweep, codebook tiles (2 MiB, L2-resident) are
// read with 16-byte loads, and the (min, index) reduction is a wavefront shuffle + one LDS hop.
#include "ops.h"

namespace ivg {

__global__ __launch_bounds__(256) void sqnorm_rows_kernel(const float* __restrict__ E, float* __restrict__ out, int rows, int dim) {
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  float s = 0.f;
  for (int k = 0; k < dim; ++k) { const float v = E[(long)r * dim + k]; s = fmaf(v, v, s); }
  out[r] = s;
}

int launch_sqnorm_rows(const float* E, float* out, int rows, int dim, hipStream_t st) {
  hipLaunchKernelGGL(sqnorm_rows_kernel, dim3(cdiv(rows, 256)), dim3(256), 0, st, E, out, rows, dim);
  return (int)hipGetLastError();
}

__device__ __forceinline__ long tok_addr(const TokMap& m, int r) {
  const int g = r / m.tpf, i = r - g * m.tpf;
  const int b = g / m.nf, f = g - b * m.nf;
  return (long)b * m.stride + m.start + (long)f * m.fstride + i;
}

// 64 z rows per workgroup (4 m-fragments), 4 waves each sweep a quarter of the codebook.  dim = 64.
__global__ __launch_bounds__(256) void vq_argmin_kernel(const float* __restrict__ z, const float* __restrict__ E,
                                                        const float* __restrict__ ee, int64_t* __restrict__ out, TokMap map,
                                                        int64_t offset, int R, int n_e) {
  __shared__ float s_val[4][64];
  __shared__ int s_idx[4][64];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lg = lane >> 4;
  const int r0 = blockIdx.x * 64;
  // z fragments (MFMA B operand): row r0 + fm*16 + lr, chunk c = kk*4 + lg  (16 chunks of 4 floats = 64 dims)
  f32x4 zf[4][4];
  float zz[4];
#pragma unroll
  for (int fm = 0; fm < 4; ++fm) {
    const int r = r0 + fm * 16 + lr;
    float part = 0.f;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (r < R) v = *(const f32x4*)(z + (long)r * 64 + (kk * 4 + lg) * 4);
      zf[fm][kk] = v;
      part += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    // ||z||^2 of row lr: combine the 4 lane groups holding the other chunks of the same row
    part += __shfl_xor(part, 16, 64);
    part += __shfl_xor(part, 32, 64);
    zz[fm] = part;
  }
  float best[4];
  int bidx[4];
#pragma unroll
  for (int fm = 0; fm < 4; ++fm) { best[fm] = INFINITY; bidx[fm] = 0x7fffffff; }

  const int per_wave = (n_e + 3) / 4;
  const int n_beg = wave * per_wave, n_end = min(n_e, n_beg + per_wave);
  for (int n0 = n_beg; n0 < n_end; n0 += 16) {
    // codebook fragment (MFMA A operand): code n0 + lr, chunks kk*4 + lg
    f32x4 ef[4];
    const int code = n0 + lr;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
      if (code < n_end) v = *(const f32x4*)(E + (long)code * 64 + (kk * 4 + lg) * 4);
      ef[kk] = v;
    }
    f32x4 een = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int r = 0; r < 4; ++r) { const int c = n0 + lg * 4 + r; een[r] = c < n_end ? ee[c] : 0.f; }
#pragma unroll
    for (int fm = 0; fm < 4; ++fm) {
      f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
#pragma unroll
        for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(ef[kk][s], zf[fm][kk][s], acc, 0, 0, 0);
      // acc[r] = z_{row lr} . e_{n0 + lg*4 + r}
      const float zzm = zz[fm];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int c = n0 + lg * 4 + r;
        const float d2 = fmaf(-2.0f, acc[r], zzm) + een[r];
        const float d = sqrtf(fmaxf(d2, 0.f));
        if (c < n_end && d < best[fm]) { best[fm] = d; bidx[fm] = c; }  // strict <: keeps the lowest index
      }
    }
  }
  // combine the 4 lane groups of each row, then the 4 waves (ties -> lowest index)
#pragma unroll
  for (int fm = 0; fm < 4; ++fm) {
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {
      const float ov = __shfl_xor(best[fm], o, 64);
      const int oi = __shfl_xor(bidx[fm], o, 64);
      if (ov < best[fm] || (ov == best[fm] && oi < bidx[fm])) { best[fm] = ov; bidx[fm] = oi; }
    }
    if (lg == 0) { s_val[wave][fm * 16 + lr] = best[fm]; s_idx[wave][fm * 16 + lr] = bidx[fm]; }
  }
  __syncthreads();
  if (tid < 64) {
    const int r = r0 + tid;
    if (r < R) {
      float bv = s_val[0][tid];
      int bi = s_idx[0][tid];
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const float ov = s_val[w][tid];
        const int oi = s_idx[w][tid];
        if (ov < bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
      }
      out[tok_addr(map, r)] = (int64_t)bi + offset;
    }
  }
}

int launch_vq_argmin(const float* z, const float* E, const float* ee, int64_t* out, const TokMap& map, int64_t offset, int R,
                     int n_e, hipStream_t st) {
  if (R <= 0) return 0;
  hipLaunchKernelGGL(vq_argmin_kernel, dim3(cdiv(R, 64)), dim3(256), 0, st, z, E, ee, out, map, offset, R, n_e);
  return (int)hipGetLastError();
}

// Y[r][:] = E[clamp(ids[addr(r)] - sub, 0, n_e-1)][:]   (codebook fp32 -> T), dim % 4 == 0
template <typename T>
__global__ __launch_bounds__(256) void gather_rows_kernel(const int64_t* __restrict__ ids, TokMap map, const float* __restrict__ E,
                                                          T* __restrict__ Y, int R, int dim, int64_t sub, int n_e) {
  const int q = dim / 4;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)R * q) return;
  const int r = (int)(i / q), c = (int)(i % q) * 4;
  long id = ids[tok_addr(map, r)] - sub;
  id = id < 0 ? 0 : (id > n_e - 1 ? n_e - 1 : id);
  const f32x4 v = *(const f32x4*)(E + id * dim + c);
  if constexpr (sizeof(T) == 2) {
    *(bf16x4*)(Y + (long)r * dim + c) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
  } else {
    *(f32x4*)(Y + (long)r * dim + c) = v;
  }
}

int launch_gather_rows(const int64_t* ids, const TokMap& map, const float* E, void* Y, DType dt, int R, int dim, int64_t sub,
                       int n_e, hipStream_t st) {
  if (R <= 0) return 0;
  dim3 g(cdiv((long)R * (dim / 4), 256));
  if (dt == BF16)
    hipLaunchKernelGGL(gather_rows_kernel<bf16_t>, g, dim3(256), 0, st, ids, map, E, (bf16_t*)Y, R, dim, sub, n_e);
  else
    hipLaunchKernelGGL(gather_rows_kernel<float>, g, dim3(256), 0, st, ids, map, E, (float*)Y, R, dim, sub, n_e);
  return (int)hipGetLastError();
}

// q[M*np*np][p*p*C], feature order (ph, pw, c)  ->  NHWC [M][np*p][np*p][C]     (compressive_vq_model.py:247-250)
template <typename T>
__global__ __launch_bounds__(256) void unpatchify_kernel(const T* __restrict__ q, T* __restrict__ out, int M, int np, int C, int p) {
  const long total = (long)M * np * np * p * p * C;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int c = (int)(i % C);
  long t = i / C;
  const int pw = (int)(t % p); t /= p;
  const int ph = (int)(t % p); t /= p;
  const int wp = (int)(t % np); t /= np;
  const int hp = (int)(t % np);
  const int m = (int)(t / np);
  const int side = np * p;
  out[(((long)m * side + hp * p + ph) * side + wp * p + pw) * C + c] = q[i];
}

int launch_unpatchify(const void* q, void* out, DType dt, int M, int np, int C, int p, hipStream_t st) {
  const long total = (long)M * np * np * p * p * C;
  if (total <= 0) return 0;
  dim3 g(cdiv(total, 256));
  if (dt == BF16)
    hipLaunchKernelGGL(unpatchify_kernel<bf16_t>, g, dim3(256), 0, st, (const bf16_t*)q, (bf16_t*)out, M, np, C, p);
  else
    hipLaunchKernelGGL(unpatchify_kernel<float>, g, dim3(256), 0, st, (const float*)q, (float*)out, M, np, C, p);
  return (int)hipGetLastError();
}

// special tokens + labels   (compressive_vq_model.py:205-218); ids row stride `stride`, labels dense [B][L]
__global__ __launch_bounds__(256) void finish_tokens_kernel(int64_t* __restrict__ ids, long stride, int64_t* __restrict__ labels,
                                                            int B, int L, int ctx, int64_t scf, int64_t sdf) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * L) return;
  const int pos = (int)(i % L);
  const long at = (i / L) * stride + pos;
  const int nctx = ctx * 257;  // context part incl. the first sdf
  if (pos < nctx - 1) {
    if (pos % 257 == 256) ids[at] = scf;
  } else if ((pos - (nctx - 1)) % 17 == 0) {
    ids[at] = sdf;
  }
  if (labels) labels[i] = pos < nctx ? (int64_t)-100 : ids[at];
}

int launch_finish_tokens(int64_t* ids, long stride, int64_t* labels, int B, int L, int ctx, int64_t scf, int64_t sdf, hipStream_t st) {
  hipLaunchKernelGGL(finish_tokens_kernel, dim3(cdiv((long)B * L, 256)), dim3(256), 0, st, ids, stride, labels, B, L, ctx, scf, sdf);
  return (int)hipGetLastError();
}

}  // namespace ivg
