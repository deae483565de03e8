// First convolution of the encoders (3 -> C0 channels, 3x3, pad 1): reads the (B, T, 3, H, W) clip
// directly (planar rows are contiguous along W -> coalesced), writes NHWC activations.
// 0.03 GFLOP/frame at 64x64: bandwidth/latency bound, so a direct VALU kernel (no MFMA: K = 27).
// Replaces `conv_in` of Encoder / ConditionalEncoder (ivideogpt/vq_model/vae.py:86-92).
#include "ops.h"

namespace ivg {

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void conv_in_kernel(const TI* __restrict__ video, const float* __restrict__ w,
                                                      const float* __restrict__ bias, TO* __restrict__ Y, int N, int per,
                                                      int T_total, int t0, int H, int W, int C0) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* sw = (float*)smem;          // [27][C0]  (k-major so a thread's 8 couts are contiguous)
  float* sb = sw + 27 * C0;          // [C0]
  for (int i = threadIdx.x; i < 27 * C0; i += 256) {
    const int co = i / 27, k = i - co * 27;  // source layout [C0][3][3][3] = [co][ci*9 + kh*3 + kw]
    sw[k * C0 + co] = w[i];
  }
  for (int i = threadIdx.x; i < C0; i += 256) sb[i] = bias[i];
  __syncthreads();
  const int cgn = C0 / 8;
  const long total = (long)N * H * W * cgn;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const int cg = (int)(idx % cgn);
    long t = idx / cgn;
    const int ow = (int)(t % W); t /= W;
    const int oh = (int)(t % H);
    const int n = (int)(t / H);
    const long fr = (long)(n / per) * T_total + t0 + (n % per);
    const TI* src = video + fr * 3 * H * W;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = sb[cg * 8 + j];
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int ih = oh + kh - 1, iw = ow + kw - 1;
          float x = 0.f;
          if (ih >= 0 && ih < H && iw >= 0 && iw < W) x = to_f32(src[((long)ci * H + ih) * W + iw]);
          const float* wk = sw + (ci * 9 + kh * 3 + kw) * C0 + cg * 8;
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = fmaf(x, wk[j], acc[j]);
        }
    TO* o = Y + (((long)n * H + oh) * W + ow) * C0 + cg * 8;
    if constexpr (sizeof(TO) == 2) {
      bf16x8 v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = (bf16_t)acc[j];
      *(bf16x8*)o = v;
    } else {
      *(f32x4*)o = f32x4{acc[0], acc[1], acc[2], acc[3]};
      *(f32x4*)(o + 4) = f32x4{acc[4], acc[5], acc[6], acc[7]};
    }
  }
}

int launch_conv_in(const void* video, DType video_dt, const float* w, const float* bias, void* Y, DType dt, int N, int per,
                   int T_total, int t0, int H, int W, int C0, hipStream_t st) {
  if (C0 % 8 != 0 || N <= 0) return N <= 0 ? 0 : (int)hipErrorInvalidValue;
  const long total = (long)N * H * W * (C0 / 8);
  const int blocks = (int)(total / 256 > 16384 ? 16384 : cdiv(total, 256));
  const size_t smem = (size_t)(28 * C0) * sizeof(float);
  dim3 g(blocks), b(256);
#define IVG_CI(TI, TO) hipLaunchKernelGGL((conv_in_kernel<TI, TO>), g, b, smem, st, (const TI*)video, w, bias, (TO*)Y, N, per, T_total, t0, H, W, C0)
  if (video_dt == F32 && dt == F32) IVG_CI(float, float);
  else if (video_dt == F32 && dt == BF16) IVG_CI(float, bf16_t);
  else if (video_dt == BF16 && dt == F32) IVG_CI(bf16_t, float);
  else IVG_CI(bf16_t, bf16_t);
#undef IVG_CI
  return (int)hipGetLastError();
}

}  // namespace ivg
