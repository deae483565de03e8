// Frame metrics on the device: what the reference's Evaluator.forward computes per predicted clip
// (/root/reference/ivideogpt/utils/video_metric.py:63-100) minus LPIPS (external VGG weights):
//   mse_f  = mean over (3, H, W) of (x - y)^2                                   nn.MSELoss(reduction='none').mean([1, 2, 3])
//   psnr_f = 10 log10(1 / (mse_f + 1e-8))                                       piqa.PSNR(epsilon=1e-8, value_range=1)
//   ssim_f = mean over (3, H-10, W-10) of the SSIM map with an 11-tap Gaussian (sigma 1.5, separable, no padding),
//            c1 = 0.01^2, c2 = 0.03^2                                           piqa.SSIM(window_size=11, sigma=1.5, n_channels=3)
// then per trajectory the mean over its frames and the best of the t samples drawn for it (min mse, max psnr, max ssim).
// The rows (B, 3) are what the one collective of the multi-GPU path ships (train_gpt.py:476-479).
//
// HBM-bound: every pixel of both clips is read once.  Kernel 1: a workgroup owns a 32 x 32 tile of the SSIM map of one
// (sample, frame, channel): the (42 x 42) patches of x and y go through LDS, the five filtered moments (x, y, xx, yy, xy) are
// produced by a horizontal pass into LDS and a vertical pass in registers; it also sums the squared error of the input pixels
// it owns.  Partials land in a workspace in a fixed layout; kernel 2 reduces them in a fixed order (deterministic, no atomics).
#include <cmath>

#include "ops.h"

namespace ivg {

constexpr int MT = 32;          // SSIM-map tile side
constexpr int MW = 11;          // Gaussian window
constexpr int MP = MT + MW - 1; // input patch side (42)

struct MetricDev {
  const void* gt; const float* pred; float* part;
  int B, n_samples, T, H, W;
  long gt_bstride, gt_t0_off, pr_bstride, pr_t0_off;   // elements
  int tiles_y, tiles_x;
  float g[MW];
};

template <typename TG>
__global__ __launch_bounds__(256) void metric_tile_kernel(const MetricDev p) {
  __shared__ float sx[MP * MP], sy[MP * MP];
  __shared__ float hm[5][MP * MT];    // horizontally filtered moments: [moment][row][col]
  __shared__ float red[2][4];
  const int tid = threadIdx.x;
  int bid = blockIdx.x;
  const int tx = bid % p.tiles_x; bid /= p.tiles_x;
  const int ty = bid % p.tiles_y; bid /= p.tiles_y;
  const int c = bid % 3; bid /= 3;
  const int f = bid % p.T; bid /= p.T;
  const int s = bid;                         // sample row k * B + b
  const int b = s % p.B;
  const long plane = (long)p.H * p.W;
  const TG* X = (const TG*)p.gt + (long)b * p.gt_bstride + p.gt_t0_off + ((long)f * 3 + c) * plane;
  const float* Y = p.pred + (long)s * p.pr_bstride + p.pr_t0_off + ((long)f * 3 + c) * plane;
  const int y0 = ty * MT, x0 = tx * MT;
  // ---- patches (zero beyond the image: those taps only feed SSIM outputs that are masked below)
  for (int i = tid; i < MP * MP; i += 256) {
    const int r = i / MP, q = i - r * MP;
    const int yy = y0 + r, xx = x0 + q;
    const bool in = yy < p.H && xx < p.W;
    sx[i] = in ? to_f32(X[(long)yy * p.W + xx]) : 0.f;
    sy[i] = in ? Y[(long)yy * p.W + xx] : 0.f;
  }
  __syncthreads();
  // ---- squared error of the owned input pixels: rows [y0, y0 + 32) (the last tile row owns up to H), same for columns
  float se = 0.f;
  {
    const int oy1 = ty == p.tiles_y - 1 ? p.H : y0 + MT, ox1 = tx == p.tiles_x - 1 ? p.W : x0 + MT;
    // pixels beyond the 42-wide patch (only when H or W is not of the form 32 k + 10 .. ) are read from global memory
    for (int yy = y0 + tid / 64; yy < oy1; yy += 4)
      for (int xx = x0 + (tid & 63); xx < ox1; xx += 64) {
        float d;
        if (yy - y0 < MP && xx - x0 < MP) d = sx[(yy - y0) * MP + (xx - x0)] - sy[(yy - y0) * MP + (xx - x0)];
        else d = to_f32(X[(long)yy * p.W + xx]) - Y[(long)yy * p.W + xx];
        se = fmaf(d, d, se);
      }
  }
  // ---- horizontal pass: MP rows x MT columns x 5 moments
  for (int i = tid; i < MP * MT; i += 256) {
    const int r = i / MT, q = i - r * MT;
    float mx = 0.f, my = 0.f, mxx = 0.f, myy = 0.f, mxy = 0.f;
#pragma unroll
    for (int k = 0; k < MW; ++k) {
      const float a = sx[r * MP + q + k], bb = sy[r * MP + q + k], w = p.g[k];
      mx = fmaf(w, a, mx); my = fmaf(w, bb, my); mxx = fmaf(w, a * a, mxx); myy = fmaf(w, bb * bb, myy); mxy = fmaf(w, a * bb, mxy);
    }
    hm[0][i] = mx; hm[1][i] = my; hm[2][i] = mxx; hm[3][i] = myy; hm[4][i] = mxy;
  }
  __syncthreads();
  // ---- vertical pass + SSIM map
  const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
  const int oh = p.H - (MW - 1), ow = p.W - (MW - 1);
  float ss = 0.f;
  for (int i = tid; i < MT * MT; i += 256) {
    const int r = i / MT, q = i - r * MT;
    if (y0 + r >= oh || x0 + q >= ow) continue;
    float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < MW; ++k) {
      const float w = p.g[k];
#pragma unroll
      for (int u = 0; u < 5; ++u) m[u] = fmaf(w, hm[u][(r + k) * MT + q], m[u]);
    }
    const float mu_xx = m[0] * m[0], mu_yy = m[1] * m[1], mu_xy = m[0] * m[1];
    const float s_xx = m[2] - mu_xx, s_yy = m[3] - mu_yy, s_xy = m[4] - mu_xy;
    const float cs = (2.f * s_xy + c2) / (s_xx + s_yy + c2);
    ss += (2.f * mu_xy + c1) / (mu_xx + mu_yy + c1) * cs;
  }
  se = wave_sum(se); ss = wave_sum(ss);
  if ((tid & 63) == 0) { red[0][tid >> 6] = se; red[1][tid >> 6] = ss; }
  __syncthreads();
  if (tid == 0) {
    p.part[2 * (long)blockIdx.x] = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    p.part[2 * (long)blockIdx.x + 1] = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  }
}

// one workgroup per trajectory b: frames of sample k in fixed order -> per-frame metrics -> mean over frames -> best of t
__global__ __launch_bounds__(64) void metric_reduce_kernel(const float* __restrict__ part, float* __restrict__ rows, int B, int n_samples,
                                                          int T, int H, int W, int tiles) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int t = n_samples / B;
  const double n_px = 3.0 * H * W, n_ss = 3.0 * (H - 10) * (W - 10);
  float best_mse = INFINITY, best_psnr = -INFINITY, best_ssim = -INFINITY;
  for (int k = 0; k < t; ++k) {
    const int s = k * B + b;
    double a_mse = 0.0, a_psnr = 0.0, a_ssim = 0.0;
    for (int f = lane; f < T; f += 64) {
      const float* pp = part + 2 * ((long)(s * T + f) * 3 * tiles);
      double se = 0.0, ss = 0.0;
      for (int i = 0; i < 3 * tiles; ++i) { se += pp[2 * i]; ss += pp[2 * i + 1]; }
      const float mse = (float)(se / n_px);
      a_mse += mse;
      a_psnr += 10.f * log10f(1.f / (mse + 1e-8f));
      a_ssim += (float)(ss / n_ss);
    }
    a_mse = wave_sum(a_mse); a_psnr = wave_sum(a_psnr); a_ssim = wave_sum(a_ssim);
    const float m1 = (float)(a_mse / T), m2 = (float)(a_psnr / T), m3 = (float)(a_ssim / T);
    best_mse = fminf(best_mse, m1); best_psnr = fmaxf(best_psnr, m2); best_ssim = fmaxf(best_ssim, m3);
  }
  if (lane == 0) { rows[3 * b] = best_mse; rows[3 * b + 1] = best_psnr; rows[3 * b + 2] = best_ssim; }
}

size_t frame_metrics_ws_bytes(int n_samples, int T, int H, int W) {
  const int ty = cdiv(std::max(1, H - (MW - 1)), MT), tx = cdiv(std::max(1, W - (MW - 1)), MT);
  return (size_t)n_samples * T * 3 * ty * tx * 2 * sizeof(float);
}

int launch_frame_metrics(const void* gt, DType gt_dt, int B, int T_gt, int gt_t0, const float* pred, int n_samples, int T_pr, int pr_t0, int T,
                         int H, int W, float* rows, void* ws, hipStream_t st) {
  if (B <= 0 || n_samples <= 0 || n_samples % B != 0 || T <= 0 || H < MW || W < MW) return (int)hipErrorInvalidValue;
  MetricDev d;
  d.gt = gt; d.pred = pred; d.part = (float*)ws;
  d.B = B; d.n_samples = n_samples; d.T = T; d.H = H; d.W = W;
  const long frame = 3L * H * W;
  d.gt_bstride = (long)T_gt * frame; d.gt_t0_off = (long)gt_t0 * frame;
  d.pr_bstride = (long)T_pr * frame; d.pr_t0_off = (long)pr_t0 * frame;
  d.tiles_y = cdiv(H - (MW - 1), MT); d.tiles_x = cdiv(W - (MW - 1), MT);
  {  // piqa.gaussian_kernel(11, 1.5): exp(-(i - 5)^2 / (2 sigma^2)) normalised, in fp32 like the reference
    float sum = 0.f;
    for (int i = 0; i < MW; ++i) { const float x = (float)i - (MW - 1) / 2.0f; d.g[i] = expf(-(x * x) / (2.f * 1.5f * 1.5f)); sum += d.g[i]; }
    for (int i = 0; i < MW; ++i) d.g[i] /= sum;
  }
  const long blocks = (long)n_samples * T * 3 * d.tiles_y * d.tiles_x;
  if (gt_dt == BF16) hipLaunchKernelGGL(metric_tile_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, d);
  else hipLaunchKernelGGL(metric_tile_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, d);
  int rc = (int)hipGetLastError();
  if (rc) return rc;
  hipLaunchKernelGGL(metric_reduce_kernel, dim3((unsigned)B), dim3(64), 0, st, (const float*)ws, rows, B, n_samples, T, H, W, d.tiles_y * d.tiles_x);
  return (int)hipGetLastError();
}

}  // namespace ivg
