// Clip ingest on the device: uint8 episode frames [T][H][W][3] -> planar float clip [T][3][R][R] in [0, 1],
// i.e. the reference's preprocessing (inference/utils.py:12-16, ivideogpt/data/simple_dataloader.py:512-516):
//     images / 255  ->  (optional centre crop to the short side)  ->  torchvision F.resize(images, [R, R])
// where the tensor path of torchvision 0.17's resize is interpolate(mode='bilinear', align_corners=False, antialias=True).
// The antialiased bilinear filter is ATen's separable one (aten/src/ATen/native/cpu/UpSampleKernel.cpp,
// _compute_indices_min_size_weights_aa): per output index i, scale = in / out, support = max(scale, 1), centre = scale (i + 0.5),
// taps [int(centre - support + 0.5), int(centre + support + 0.5)) clipped to the image, triangle weights
// max(0, 1 - |(j - centre + 0.5) / max(scale, 1)|) normalised to 1; the width axis is filtered first, then the height axis,
// all in fp32 -- restated here with the same operation order.
//
// One workgroup = one frame x a block of output rows.  HBM-bound, every source byte is read once: the source rows the block
// needs arrive as aligned 32-bit words (an interleaved RGB row is one contiguous run), the horizontal pass goes LDS -> LDS,
// the vertical pass LDS -> planar output with the x index fastest (coalesced stores).
#include <cmath>

#include "ops.h"

namespace ivg {

constexpr int ING_MAXT = 32;    // taps per output index: 2 * support + 1 <= 32  ->  downscale factors up to 15

struct IngestDev {
  const unsigned char* src; void* dst;
  int T, H, W;                 // source frames
  int x_off, y_off, cw, ch;    // crop window
  int R;                       // output side
  int rows_per_wg, blocks_per_frame;
  int in_rows_cap;             // source rows a block may need (LDS sizing)
  int pitch;                   // LDS bytes per staged source row (multiple of 4)
  size_t total_bytes;          // of the source buffer (the aligned word loads never leave it)
};

__device__ __forceinline__ void aa_taps(int i, int in_size, float scale, float support, int& xmin, int& xsize, float* w) {
  // same types as ATen: centre and the weights are fp32, (i + 0.5) and the filter argument are formed in double
  const float center = (float)((double)scale * ((double)i + 0.5));
  const float invscale = scale >= 1.0f ? 1.0f / scale : 1.0f;
  xmin = max((int)((double)(center - support) + 0.5), 0);
  xsize = min((int)((double)(center + support) + 0.5), in_size) - xmin;
  if (xsize > ING_MAXT) xsize = ING_MAXT;
  float total = 0.f;
  for (int j = 0; j < xsize; ++j) {
    float x = (float)(((double)((float)(j + xmin) - center) + 0.5) * (double)invscale);
    x = fabsf(x);
    const float v = x < 1.0f ? 1.0f - x : 0.0f;
    w[j] = v;
    total += v;
  }
  if (total != 0.f)
    for (int j = 0; j < xsize; ++j) w[j] /= total;
}

template <typename TO>
__global__ __launch_bounds__(256) void ingest_kernel(const IngestDev p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  // layout: wx [R][MAXT] f32 | xmin [R] | xsize [R] | wy [rows][MAXT] | ymin [rows] | ysize [rows] | hbuf [in_rows][R][3] f32 | bytes
  float* wx = (float*)smem;
  int* xmin = (int*)(wx + (size_t)p.R * ING_MAXT);
  int* xsize = xmin + p.R;
  float* wy = (float*)(xsize + p.R);
  int* ymin = (int*)(wy + (size_t)p.rows_per_wg * ING_MAXT);
  int* ysize = ymin + p.rows_per_wg;
  float* hbuf = (float*)(ysize + p.rows_per_wg);
  unsigned char* bytes = (unsigned char*)(hbuf + (size_t)p.in_rows_cap * p.R * 3);
  __shared__ int s_y0, s_y1;
  const int tid = threadIdx.x;
  const int t = blockIdx.x / p.blocks_per_frame, blk = blockIdx.x % p.blocks_per_frame;
  const int r0 = blk * p.rows_per_wg, r1 = min(r0 + p.rows_per_wg, p.R);
  const float sx = (float)p.cw / (float)p.R, sy = (float)p.ch / (float)p.R;
  const float supx = sx >= 1.0f ? sx : 1.0f, supy = sy >= 1.0f ? sy : 1.0f;
  for (int i = tid; i < p.R; i += 256) aa_taps(i, p.cw, sx, supx, xmin[i], xsize[i], wx + (size_t)i * ING_MAXT);
  for (int i = tid; i < r1 - r0; i += 256) aa_taps(r0 + i, p.ch, sy, supy, ymin[i], ysize[i], wy + (size_t)i * ING_MAXT);
  __syncthreads();
  if (tid == 0) {
    int lo = p.ch, hi = 0;
    for (int i = 0; i < r1 - r0; ++i) { lo = min(lo, ymin[i]); hi = max(hi, ymin[i] + ysize[i]); }
    s_y0 = lo; s_y1 = min(hi, lo + p.in_rows_cap);
  }
  __syncthreads();
  const int y0 = s_y0, ny = s_y1 - s_y0;
  // ---- stage the source rows: aligned 32-bit words covering bytes [row start, row start + cw * 3)
  const unsigned char* frame = p.src + (size_t)t * p.H * p.W * 3;
  for (int y = 0; y < ny; ++y) {
    const size_t b0 = ((size_t)(p.y_off + y0 + y) * p.W + p.x_off) * 3;
    const unsigned char* rp = frame + b0;
    const int mis = (int)((uintptr_t)rp & 3);
    const unsigned* wsrc = (const unsigned*)(rp - mis);
    const int nw = (mis + p.cw * 3 + 3) >> 2;
    unsigned* wdst = (unsigned*)(bytes + (size_t)y * p.pitch);
    const size_t w0 = (size_t)(rp - mis - p.src);   // may wrap for the very first word of a misaligned buffer: checked below
    for (int i = tid; i < nw; i += 256) {
      const bool inside = (rp - mis >= p.src || i > 0) && w0 + 4 * (size_t)(i + 1) <= p.total_bytes;
      if (inside) wdst[i] = wsrc[i];
      else {   // first / last word of the buffer: byte by byte
        unsigned v = 0;
        for (int k = 0; k < 4; ++k) {
          const unsigned char* bp = rp - mis + 4 * i + k;
          if (bp >= p.src && (size_t)(bp - p.src) < p.total_bytes) v |= (unsigned)*bp << (8 * k);
        }
        wdst[i] = v;
      }
    }
  }
  __syncthreads();
  // ---- horizontal pass: hbuf[y][x][c] = sum_j wx[x][j] * (u8 / 255)
  for (int i = tid; i < ny * p.R * 3; i += 256) {
    const int c = i % 3, x = (i / 3) % p.R, y = i / (3 * p.R);
    const size_t b0 = ((size_t)(p.y_off + y0 + y) * p.W + p.x_off) * 3;
    const int mis = (int)((uintptr_t)(frame + b0) & 3);
    const unsigned char* row = bytes + (size_t)y * p.pitch + mis;
    const float* w = wx + (size_t)x * ING_MAXT;
    const int xm = xmin[x], n = xsize[x];
    float acc = ((float)row[(xm)*3 + c] / 255.0f) * w[0];
    for (int j = 1; j < n; ++j) acc += ((float)row[(xm + j) * 3 + c] / 255.0f) * w[j];
    hbuf[i] = n > 0 ? acc : 0.f;
  }
  __syncthreads();
  // ---- vertical pass -> planar output [t][c][r][x]
  TO* out = (TO*)p.dst + (size_t)t * 3 * p.R * p.R;
  for (int i = tid; i < (r1 - r0) * p.R * 3; i += 256) {
    const int x = i % p.R, r = (i / p.R) % (r1 - r0), c = i / (p.R * (r1 - r0));
    const float* w = wy + (size_t)r * ING_MAXT;
    const int ym = ymin[r] - y0, n = ysize[r];
    float acc = 0.f;
    if (n > 0 && ym >= 0 && ym + n <= ny) {
      acc = hbuf[((size_t)ym * p.R + x) * 3 + c] * w[0];
      for (int j = 1; j < n; ++j) acc += hbuf[((size_t)(ym + j) * p.R + x) * 3 + c] * w[j];
    }
    out[((size_t)c * p.R + (r0 + r)) * p.R + x] = from_f32<TO>(acc);
  }
}

// crop: 0 = none (aspect ratio squashed, inference/utils.py:12-16), 1 = centre crop to the short side first
// (simple_dataloader.py:512-516 for tfds_robonet; torchvision center_crop: top = round((H - s) / 2), left likewise)
int launch_ingest(const unsigned char* src, int T, int H, int W, int crop, void* dst, DType dst_dt, int R, hipStream_t st) {
  if (T <= 0 || H <= 0 || W <= 0 || R <= 0 || R > 1024) return (int)hipErrorInvalidValue;
  IngestDev d;
  d.src = src; d.dst = dst; d.T = T; d.H = H; d.W = W; d.R = R;
  d.x_off = 0; d.y_off = 0; d.cw = W; d.ch = H;
  if (crop) {
    const int s = H < W ? H : W;
    d.y_off = (int)nearbyint((H - s) / 2.0); d.x_off = (int)nearbyint((W - s) / 2.0);   // Python round(): half to even
    d.cw = s; d.ch = s;
  }
  const double sx = (double)d.cw / R, sy = (double)d.ch / R;
  const double supx = sx >= 1 ? sx : 1, supy = sy >= 1 ? sy : 1;
  if (2 * supx + 1 > ING_MAXT || 2 * supy + 1 > ING_MAXT) return (int)hipErrorInvalidValue;   // downscale factor beyond 15
  d.pitch = ((d.cw * 3 + 3 + 3) / 4) * 4 + 4;
  // rows per workgroup: as many as keep the staged source rows + the horizontally filtered rows inside ~96 KB of LDS
  int rows = 16;
  auto smem_for = [&](int rpw, int& cap) {
    cap = (int)(rpw * sy + 2 * supy + 4);
    if (cap > d.ch) cap = d.ch;
    return (size_t)R * ING_MAXT * 4 + (size_t)R * 8 + (size_t)rpw * ING_MAXT * 4 + (size_t)rpw * 8 + (size_t)cap * R * 3 * 4 + (size_t)cap * d.pitch + 16;
  };
  int cap = 0;
  while (rows > 1 && smem_for(rows, cap) > 96 * 1024) rows >>= 1;
  const size_t smem = smem_for(rows, cap);
  d.rows_per_wg = rows; d.in_rows_cap = cap; d.blocks_per_frame = cdiv(R, rows);
  d.total_bytes = (size_t)T * H * W * 3;
  // (the kernel also has 8 bytes of static LDS: the dynamic part may take 160 KiB minus that)
  constexpr int kMaxDyn = 160 * 1024 - 64;
  if (smem > (size_t)kMaxDyn) return (int)hipErrorInvalidValue;
  static DynLdsOnce once_f, once_b;
  if (dst_dt == BF16) {
    if (hipError_t e = ensure_dyn_lds(once_b, (const void*)ingest_kernel<bf16_t>, kMaxDyn); e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(ingest_kernel<bf16_t>, dim3((unsigned)(T * d.blocks_per_frame)), dim3(256), smem, st, d);
  } else {
    if (hipError_t e = ensure_dyn_lds(once_f, (const void*)ingest_kernel<float>, kMaxDyn); e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(ingest_kernel<float>, dim3((unsigned)(T * d.blocks_per_frame)), dim3(256), smem, st, d);
  }
  return (int)hipGetLastError();
}

}  // namespace ivg
