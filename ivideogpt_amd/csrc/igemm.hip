// Implicit-GEMM convolution / batched GEMM for gfx950 (MI355X), hand-written MFMA kernel.
//
// Replaces the library conv2d / linear / bmm ops under the reference's tokenizer and transformer
// (K1 conv3x3, K2 stride-2 conv with right/bottom zero pad, K3 1x1 conv, K5 nearest
// x2 upsample folded into the gather, K6 residual add, K7/K8 attention GEMMs, K9/K14/K17 linears).
//
// Design (wave64, 256 threads = 4 waves per workgroup):
//   * activations are NHWC so the K axis (kh, kw, c) is contiguous per tap: an A-tile row is one
//     128-byte run of channels of one input pixel -> coalesced 16-byte global loads, no im2col buffer;
//   * A (pixels) and W (output channels) tiles are staged through LDS in 128-byte rows with an XOR
//     swizzle (chunk ^ (row>>1)&7) so both the ds_write_b128 staging and the ds_read_b128 fragment
//     reads are bank-conflict free; double-buffered, global loads for tile k+1 are in flight while
//     tile k is on the matrix cores;
//   * MFMA operands are swapped (A-operand = weights): the accumulator of a lane then holds 4
//     CONSECUTIVE output channels of one pixel, so the NHWC epilogue stores/loads 8-16 bytes per lane;
//   * bf16: v_mfma_f32_16x16x32_bf16 (fp32 accumulate).  fp32: v_mfma_f32_16x16x4_f32, an exact
//     fp32 fma chain -- used by the tokenize path where VQ indices must match the fp32 reference.
#include <cstdlib>

#include "igemm.h"

namespace ivg {

struct IgemmDev {
  const void* X; const void* W; void* Y; const void* R; const float* bias;
  int Hin, Win, Cin, ldx, Hout, Wout, KW, stride, pad, ups;
  int M, N, K, ldw, HWo, single_tap;
  long c_img, c_pix, c_ch, c_grp_stride;
  int c_grp, flags;
  float alpha;
  int nb1, nb2;
  long sa[3], sw[3], sy[3];
};

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// source of every out-of-image / out-of-range 16-byte chunk of the LDS-DMA path (zero-initialised device global)
__device__ __attribute__((aligned(16))) unsigned char g_zero_chunk[16];

__device__ __forceinline__ void glds16(const void* gsrc, unsigned char* lds_wave_base) {
  // global -> LDS DMA, 16 B per lane; LDS destination = wave-uniform base + lane * 16 (cdna_hip_programming.md 5)
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// 4 fp32 values -> [bf16 hi(4) | bf16 lo(4)] with hi = bf16(x), lo = bf16(x - hi): the operand format of the split-bf16 ("x3") mode
__device__ __forceinline__ bf16x8 split_hi_lo(const f32x4 x) {
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 4; ++j) { const bf16_t hi = (bf16_t)x[j]; o[j] = hi; o[4 + j] = (bf16_t)(x[j] - (float)hi); }
  return o;
}

// X3 (T = float only): split-bf16 arithmetic of the 1e-3-compliant decode / rollout mode.  Both operands stay fp32 in HBM and in
// LDS; a fragment (4 fp32 of K per lane) is split into [hi | lo] bf16 in registers, and two K = 32 bf16 MFMAs -- the activation
// slot against [w_hi | w_hi], then against [w_lo | w_lo] -- produce all four partial products with fp32 accumulation: 2 x 16 MFMA
// clocks per 16 elements of K where the f32-input MFMA path needs 4 x 32.  Operand error 2^-17 instead of 2^-24.
template <typename T, int BM, int BN, int WM, int WN, bool X3 = false>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmDev p) {
  static_assert(!X3 || sizeof(T) == 4, "split-bf16 arithmetic reads fp32 tensors");
  constexpr int VEC = Traits<T>::VEC;
  constexpr int BK = 8 * VEC;  // one 128-byte row
  constexpr int FM = WM / 16, FN = WN / 16;
  constexpr int WAVES_M = BM / WM;
  static_assert((BM / WM) * (BN / WN) == 4, "4 waves per workgroup");
  constexpr int A_IT = BM / 32;
  constexpr int B_IT = (BN + 31) / 32;
  constexpr int A_BYTES = BM * 128, B_BYTES = BN * 128, STAGE = A_BYTES + B_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x;
  const int chunk = tid & 7, row0 = tid >> 3;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tile_n = blockIdx.x % tiles_n, tile_m = blockIdx.x / tiles_n;
  const int z = blockIdx.y;
  const int z2 = z % p.nb2, z1 = (z / p.nb2) % p.nb1, z0 = z / (p.nb2 * p.nb1);
  const T* X = (const T*)p.X + (z0 * p.sa[0] + z1 * p.sa[1] + z2 * p.sa[2]);
  const T* Wt = (const T*)p.W + (z0 * p.sw[0] + z1 * p.sw[1] + z2 * p.sw[2]);
  const long ybase = z0 * p.sy[0] + z1 * p.sy[1] + z2 * p.sy[2];

  // ---- per-thread gather bookkeeping (A_IT pixel rows, fixed for the whole K loop)
  int a_base[A_IT], a_h[A_IT], a_w[A_IT];
#pragma unroll
  for (int i = 0; i < A_IT; ++i) {
    const int m = tile_m * BM + row0 + 32 * i;
    if (m < p.M) {
      const int img = m / p.HWo, rem = m - img * p.HWo;
      const int oh = rem / p.Wout, ow = rem - oh * p.Wout;
      a_base[i] = img * p.Hin * p.Win;
      a_h[i] = oh * p.stride - p.pad;
      a_w[i] = ow * p.stride - p.pad;
    } else {
      a_base[i] = 0; a_h[i] = -(1 << 24); a_w[i] = 0;
    }
  }
  const int h_lim = p.ups ? 2 * p.Hin : p.Hin, w_lim = p.ups ? 2 * p.Win : p.Win;

  const int wave = tid >> 6, lane = tid & 63;
  // LDS-DMA staging: lane (r = lane>>3, slot = lane&7) of instruction i lands at row 8*wave + 32*i + r, slot `slot`
  // (lane-linear destination); the XOR swizzle is applied on the SOURCE chunk index, reads are unchanged.
  auto issue_tile = [&](int kt, int buf) {
    const int k0 = kt * BK;
    const int tap = p.single_tap ? 0 : k0 / p.Cin;
    const int c0 = k0 - tap * p.Cin;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    unsigned char* sA = smem + buf * STAGE;
    unsigned char* sB = sA + A_BYTES;
#pragma unroll
    for (int i = 0; i < A_IT; ++i) {
      const int row = row0 + 32 * i;
      const int c = chunk ^ ((row >> 1) & 7);
      int ih = a_h[i] + kh, iw = a_w[i] + kw;
      const bool ok = (ih >= 0) & (ih < h_lim) & (iw >= 0) & (iw < w_lim) & ((k0 + c * VEC) < p.K);
      if (p.ups) { ih >>= 1; iw >>= 1; }
      const void* src = ok ? (const void*)(X + ((long)(a_base[i] + ih * p.Win + iw) * p.ldx + c0 + c * VEC)) : (const void*)g_zero_chunk;
      glds16(src, sA + (8 * wave + 32 * i) * 128);
    }
#pragma unroll
    for (int i = 0; i < B_IT; ++i) {
      if (8 * wave + 32 * i < BN) {
        const int r = row0 + 32 * i;
        const int c = chunk ^ ((r >> 1) & 7);
        const int n = tile_n * BN + r;
        const bool ok = (n < p.N) & ((k0 + c * VEC) < p.K);
        const void* src = ok ? (const void*)(Wt + ((long)n * p.ldw + k0 + c * VEC)) : (const void*)g_zero_chunk;
        glds16(src, sB + (8 * wave + 32 * i) * 128);
      }
    }
  };
  const int wm = wave % WAVES_M, wn = wave / WAVES_M;
  const int lr = lane & 15, lg = lane >> 4;

  f32x4 acc[FN][FM];
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < FM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (p.K + BK - 1) / BK;
  issue_tile(0, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) issue_tile(kt + 1, buf ^ 1);
    const unsigned char* sA = smem + buf * STAGE;
    const unsigned char* sB = sA + A_BYTES;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int c = kk * 4 + lg;
      Chunk16 xa[FM], wb[FN];
#pragma unroll
      for (int b = 0; b < FM; ++b) xa[b] = *(const Chunk16*)(sA + lds_off(wm * WM + b * 16 + lr, c));
#pragma unroll
      for (int a = 0; a < FN; ++a) wb[a] = *(const Chunk16*)(sB + lds_off(wn * WN + a * 16 + lr, c));
      if constexpr (X3) {
        bf16x8 xs[FM];
#pragma unroll
        for (int b = 0; b < FM; ++b) xs[b] = split_hi_lo(__builtin_bit_cast(f32x4, xa[b]));
#pragma unroll
        for (int a = 0; a < FN; ++a) {
          const Chunk16 ws = __builtin_bit_cast(Chunk16, split_hi_lo(__builtin_bit_cast(f32x4, wb[a])));
#pragma unroll
          for (int h = 0; h < 2; ++h) {   // one duplicated half live at a time (registers), an accumulator revisited after FM MFMAs
            Chunk16 wd = Chunk16{ws[2 * h], ws[2 * h + 1], ws[2 * h], ws[2 * h + 1]};
            asm volatile("" : "+v"(wd));
#pragma unroll
            for (int b = 0; b < FM; ++b)
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wd), xs[b], acc[a][b], 0, 0, 0);
          }
        }
      } else {
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b) {
          if constexpr (sizeof(T) == 2) {
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wb[a]), __builtin_bit_cast(bf16x8, xa[b]),
                                                                acc[a][b], 0, 0, 0);
          } else {
            const f32x4 wv = __builtin_bit_cast(f32x4, wb[a]), xv = __builtin_bit_cast(f32x4, xa[b]);
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[s], xv[s], acc[a][b], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();  // (the compiler drains the DMA queue, vmcnt(0), in front of the barrier)
  }

  // ---- epilogue: lane holds, per fragment, 4 consecutive n (= lg*4 + r) of pixel m (= lr)
  const int flags = p.flags;
  const bool glu = flags & IG_GLU;
  const int n_valid = p.N;
#pragma unroll
  for (int b = 0; b < FM; ++b) {
    const int m = tile_m * BM + wm * WM + b * 16 + lr;
    if (m >= p.M) continue;
    const int img = m / p.HWo, pix = m - img * p.HWo;
    const long obase = ybase + (long)(img / p.c_grp) * p.c_grp_stride + (long)(img % p.c_grp) * p.c_img + (long)pix * p.c_pix;
    const float bm = (flags & IG_BIAS_M) ? p.bias[m] : 0.f;
#pragma unroll
    for (int a = 0; a < FN; ++a) {
      if (glu && (a & 1)) continue;
      const int n0 = tile_n * BN + wn * WN + a * 16 + lg * 4;
      if (n0 >= n_valid) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[a][b][r] * p.alpha + bm;
      if (flags & IG_BIAS_N) {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (n0 + r < n_valid) v[r] += p.bias[n0 + r];
      }
      int no = n0;
      if (glu) {
        if constexpr (FN >= 2) {
          const int a1 = (a + 1 < FN) ? a + 1 : a;
          float u[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) u[r] = acc[a1][b][r] * p.alpha + bm;
          if (flags & IG_BIAS_N) {
#pragma unroll
            for (int r = 0; r < 4; ++r) u[r] += p.bias[n0 + 16 + r];
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = silu_t<T>(v[r]) * u[r];
        }
        no = (n0 >> 5) * 16 + (n0 & 15);
      }
      const bool vec_ok = (p.c_ch == 1) && (n0 + 3 < n_valid) && ((p.c_pix & 3) == 0);
      const long o = obase + (long)no * p.c_ch;
      if (flags & IG_RESIDUAL) {
        const T* R = (const T*)p.R;
#pragma unroll
        for (int r = 0; r < 4; ++r) if (n0 + r < n_valid) v[r] += to_f32(R[o + (long)r * p.c_ch]);
      }
      if (flags & IG_SILU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = silu_t<T>(v[r]);
      }
      if (flags & IG_CLAMP01) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fminf(fmaxf(v[r], 0.0f), 1.0f);
      }
      if (flags & IG_OUT_F32) {
        float* Y = (float*)p.Y;
        if (vec_ok && ((o & 3) == 0)) {
          *(f32x4*)(Y + o) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (n0 + r < n_valid) Y[o + (long)r * p.c_ch] = v[r];
        }
      } else {
        T* Y = (T*)p.Y;
        if (vec_ok && ((o & 3) == 0)) {
          if constexpr (sizeof(T) == 2) {
            *(bf16x4*)(Y + o) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
          } else {
            *(f32x4*)(Y + o) = f32x4{v[0], v[1], v[2], v[3]};
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (n0 + r < n_valid) Y[o + (long)r * p.c_ch] = from_f32<T>(v[r]);
        }
      }
    }
  }
}

template <typename T, int BM, int BN, int WM, int WN, bool X3 = false>
static int launch_cfg(const IgemmDev& d, int nbatch, hipStream_t stream) {
  constexpr int smem = 2 * (BM + BN) * 128;
  static DynLdsOnce once;
  auto kfn = igemm_kernel<T, BM, BN, WM, WN, X3>;
  if (hipError_t e = ensure_dyn_lds(once, (const void*)kfn, smem); e != hipSuccess) return (int)e;
  const long tiles = (long)cdiv(d.M, BM) * cdiv(d.N, BN);
  dim3 grid((unsigned)tiles, (unsigned)nbatch, 1);
  hipLaunchKernelGGL(kfn, grid, dim3(256), smem, stream, d);
  return (int)hipGetLastError();
}

template <typename T, bool X3 = false>
static int launch_typed(const IgemmDev& d, int nbatch, hipStream_t stream) {
  if (d.flags & IG_GLU) return launch_cfg<T, 128, 128, 64, 64, X3>(d, nbatch, stream);
  if (d.N > 64) return launch_cfg<T, 128, 128, 64, 64, X3>(d, nbatch, stream);
  if (d.N > 16) return launch_cfg<T, 128, 64, 32, 64, X3>(d, nbatch, stream);
  return launch_cfg<T, 128, 16, 32, 16, X3>(d, nbatch, stream);
}

int launch_igemm(const IgemmArgs& a, DType dtype, hipStream_t stream) {
  IgemmDev d;
  d.X = a.X; d.W = a.W; d.Y = a.Y; d.R = a.R; d.bias = a.bias;
  d.Hin = a.Hin; d.Win = a.Win; d.Cin = a.Cin; d.ldx = a.ldx; d.Hout = a.Hout; d.Wout = a.Wout;
  d.KW = a.KW; d.stride = a.stride; d.pad = a.pad; d.ups = a.ups;
  d.HWo = a.Hout * a.Wout;
  d.M = a.Nimg * d.HWo; d.N = a.N; d.K = a.KH * a.KW * a.Cin; d.ldw = a.ldw;
  d.single_tap = (a.KH * a.KW == 1) ? 1 : 0;
  d.c_img = a.c_img; d.c_pix = a.c_pix; d.c_ch = a.c_ch; d.c_grp = a.c_grp > 0 ? a.c_grp : 1;
  d.c_grp_stride = a.c_grp_stride; d.flags = a.flags; d.alpha = a.alpha;
  if (a.c_grp <= 1 && a.c_grp_stride == 0) d.c_grp_stride = a.c_img;  // no frame grouping given: image n starts at n * c_img
  d.nb1 = a.nb1; d.nb2 = a.nb2;
  for (int i = 0; i < 3; ++i) { d.sa[i] = a.sa[i]; d.sw[i] = a.sw[i]; d.sy[i] = a.sy[i]; }
  const int nbatch = a.nb0 * a.nb1 * a.nb2;
  const int bk = (dtype == BF16) ? 64 : 32, vec = (dtype == BF16) ? 8 : 4;
  if (d.M <= 0 || d.N <= 0 || nbatch <= 0) return 0;
  if (a.Cin % (d.single_tap ? vec : bk) != 0 || a.ldx % vec != 0 || a.ldw % vec != 0) return (int)hipErrorInvalidValue;
  if (((uintptr_t)a.X & 15) || ((uintptr_t)a.W & 15)) return (int)hipErrorInvalidValue;
  if ((a.flags & IG_GLU) && (a.N % 32 != 0)) return (int)hipErrorInvalidValue;
  if (a.x3 && dtype == F32) return launch_typed<float, true>(d, nbatch, stream);   // split-bf16 arithmetic on fp32 tensors
  return dtype == BF16 ? launch_typed<bf16_t>(d, nbatch, stream) : launch_typed<float>(d, nbatch, stream);
}

}  // namespace ivg
