// 3x3 / stride-1 / pad-1 convolution (optionally over a nearest-x2 upsampled input) for gfx950 -- the FLOP majority
// of the tokenizer (DF ResnetBlock2D.conv1/conv2, Upsample2D.conv, conv_in of the decoders).
//
// Why a second conv kernel: the generic implicit GEMM (igemm.hip) re-gathers the A tile from global memory for every
// tap, and measured on MI355X it runs at exactly the per-CU ingest limit (~10 B/clk/CU from HBM, ~17 from L2/MALL:
// 395 TF at Cout = 128, 690 TF at Cout = 512; LDS-DMA staging changes nothing).  Here a workgroup
//   * owns a TH x TW = 256-pixel spatial tile (16x16 or 8x32) of one image and BN output channels,
//   * stages the (TH+2) x (TW+2) input HALO tile of one 32-channel (bf16; 16 fp32) chunk in LDS ONCE and runs all
//     nine taps out of it (A traffic / 9, no im2col), double-buffered across channel chunks, the next chunk's halo
//     arriving in pieces under the current chunk's taps,
//   * streams the [BN][64 B] weight tile of each (tap, chunk) through a ring of three buffers, issued two steps ahead
//     and retired by COUNTED s_waitcnt vmcnt(n) + raw s_barrier (the DMA queue is never drained inside the loop),
//   * both by LDS-DMA (global_load_lds, 16 B/lane; out-of-image pixels source a zero page), 64-byte LDS rows with an
//     XOR swizzle on the SOURCE chunk (conflict-free ds_read_b128 from any start row),
//   * 72 KiB of LDS per workgroup -> TWO workgroups (16 waves) per CU: measured with cycle stamps, one workgroup per CU
//     spends 25 % of every step in the barrier and 22 % of its life in an un-overlapped prologue / epilogue; a second
//     resident workgroup fills exactly those holes,
//   * 8 waves (2 per SIMD), each a 64 x 64 (or 64 x 32) sub-tile of MFMA fragments, swapped operands so a lane owns
//     4 consecutive output channels;  the XCD-aware block order keeps the N tiles of one spatial tile on one L2.
// Bytes per (tap, chunk) step: 8 KiB of weights + 1/9 of a ~24 KiB halo for 2.1 MFLOP -> ~195 FLOP/B (igemm: 64).
#include <cstdio>
#include <cstdlib>

#include "igemm.h"

namespace ivg {

struct Conv3Dev {
  const void* X; const void* W; void* Y; const void* R; const float* bias;
  int H, Wd, Cin, Ho, Wo;            // input H x W (before upsampling), output Ho x Wo
  int TH, TW, tw_shift;              // output tile, TW = 1 << tw_shift, TH * TW = 256
  int HTH, HTW;                      // halo tile (input pixels)
  int tiles_x, tiles_per_img, n_sp;  // spatial tiles
  int N, ldw, tiles_n;
  long c_img, c_pix, c_ch, c_grp_stride;
  int c_grp, flags;
  int hb_bytes;                      // one halo buffer
  int stage_ok;                      // the 256 x BN staging tile of the epilogue fits in the workgroup's LDS
  const f32x2* in_coef;              // GroupNorm + SiLU of the INPUT applied while it is staged (ABL bit 16): (scale, shift) [img][Cin]
  int coef_off;                      // byte offset of the two per-chunk coefficient rows in LDS
  double2* gn_part;                  // GroupNorm statistics of the output (null: off): [img][chunk = spatial tile x N tile][group]
  int gn_groups, gn_off;             // gn_off: byte offset of the per-channel partial sums in LDS (behind everything else)
  long long* dbg;                    // development: cycle stamps of workgroup 0 / wave 0 (null in production)
};

// source of every out-of-image 16-byte chunk (zero-initialised device global; one per translation unit, no RDC needed)
__device__ __attribute__((aligned(16))) unsigned char g_zero_chunk3[16];

__device__ __forceinline__ void glds16b(const void* gsrc, unsigned char* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}
// XOR key: ds_read_b128 of 16 CONSECUTIVE rows is bank-conflict free from ANY start row (the halo fragments of tap
// (kh, kw) start at arbitrary rows)
__device__ __forceinline__ int swz_key(int row) { return (row >> 1) & 3; }   // 64-byte rows, 4 chunks: found by exhaustive search
__device__ __forceinline__ int swz(int row, int chunk) { return row * 64 + ((chunk ^ swz_key(row)) << 4); }

// ABL (development ablations, 0 in production): 1 no MFMA, 2 no LDS reads + no MFMA, 4 no halo DMA, 8 no weight DMA
// ABL bit 16 (production): GroupNorm(+SiLU) of the input fused into the staging -- the halo chunk is normalised IN PLACE in LDS
// (y = silu(x * scale[c] + shift[c]), out-of-image padding stays zero) between its arrival and its first tap, piece by piece
// under the taps of the previous chunk, so the normalised tensor never exists in HBM
// NW = 8: 256-pixel tile, two workgroups per CU.  NW = 16: 512-pixel tile, one 1024-thread workgroup per CU whose 16 waves
// share ONE weight ring -- half the weight bytes per pixel, for the short-K layers that re-stream the whole weight matrix
// for every tile (launch_conv3x3 picks).
template <typename T, int BN, bool UPS, int ABL = 0, int NW = 8>
__global__ __launch_bounds__(NW * 64, NW == 8 ? 4 : 1) void conv3x3_kernel(const Conv3Dev p) {
  constexpr int VEC = Traits<T>::VEC;
  constexpr int NT = NW * 64;              // threads
  constexpr int PT = NW * 32;              // output pixels per tile
  constexpr int CK = 4 * VEC;              // channels per chunk: one 64-byte LDS row per halo pixel (one MFMA K-step)
  constexpr int WN = BN / 2;               // NW waves = NW/2 (M) x 2 (N)
  constexpr int FM = 4, FN = WN / 16;
  constexpr int W_BYTES = BN * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* hbuf0 = smem;
  unsigned char* wbuf0 = smem + 2 * p.hb_bytes;   // three weight buffers

  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int lr = lane & 15, lg = lane >> 4;
  const int wm = wave % (NW / 2), wn = wave / (NW / 2);

  // ---- XCD-aware block order (blocks b, b+8, ... share an XCD/L2): give each XCD a contiguous run of work items
  // with the N tile fastest, so the N tiles of one spatial tile hit the same L2 (bijective for any grid size)
  const int nwg = gridDim.x;
  int v;
  {
    const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int tile_n = v % p.tiles_n;
  const int sp = v / p.tiles_n;
  const int img = sp / p.tiles_per_img;
  const int t_in = sp - img * p.tiles_per_img;
  const int ty = t_in / p.tiles_x, tx = t_in - ty * p.tiles_x;
  const int y0 = ty * p.TH, x0 = tx * p.TW;            // output tile origin
  const int iy0 = UPS ? ((y0 - 1) >> 1) : (y0 - 1);    // halo origin in input pixels
  const int ix0 = UPS ? ((x0 - 1) >> 1) : (x0 - 1);
  const T* X = (const T*)p.X + (long)img * p.H * p.Wd * p.Cin;
  const T* Wt = (const T*)p.W;
  const int n_base = tile_n * BN;
  const int halo_rows = p.HTH * p.HTW;
  const int halo_iters = (halo_rows * 4 + NT - 1) / NT;

  // one piece of a halo tile: NT lanes x 16 B, lane-linear in LDS
  auto issue_halo_piece = [&](int chunk, int it, unsigned char* hb) {
    const int q = it * NT + tid;
    const int row = q >> 2, slot = q & 3;
    const int c = slot ^ swz_key(row);
    const int hy = row / p.HTW, hx = row - hy * p.HTW;
    const int iy = iy0 + hy, ix = ix0 + hx;
    const bool ok = (row < halo_rows) & (iy >= 0) & (iy < p.H) & (ix >= 0) & (ix < p.Wd);
    const void* src = ok ? (const void*)(X + ((long)(iy * p.Wd + ix) * p.Cin + chunk * CK + c * VEC)) : (const void*)g_zero_chunk3;
    glds16b(src, hb + (it * NT + wave * 64) * 16);
  };
  auto issue_w = [&](int step, unsigned char* wb) {
    const int chunk = step / 9, tap = step - chunk * 9;
    if (wave * 64 < BN * 4) {   // BN rows x 4 chunks: all 8 waves for BN = 128, the first 4 for BN = 64 (wave-uniform)
      const int q = tid;
      const int n = q >> 2, slot = q & 3;
      const int c = slot ^ swz_key(n);
      const bool ok = (n_base + n) < p.N;
      const void* src = ok ? (const void*)(Wt + ((long)(n_base + n) * p.ldw + tap * p.Cin + chunk * CK + c * VEC)) : (const void*)g_zero_chunk3;
      glds16b(src, wb + (wave * 64) * 16);
    }
  };

  constexpr bool GNA = (ABL & 16) != 0;
  f32x2* s_coef = (f32x2*)(smem + p.coef_off);   // [2][CK] (scale, shift) of the channels of the chunk being staged
  // the chunk's coefficient row (CK x 8 bytes) by one 4-byte-per-lane LDS-DMA of wave 0: no register-destination load may sit
  // beside the DMA queue (the compiler would drain it with vmcnt(0)); returns the DMA instructions this wave issued
  auto load_coef = [&](int chunk) -> int {
    if constexpr (GNA) {
      if (wave == 0) {
        if (lane < 2 * CK)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)((const float*)(p.in_coef + (long)img * p.Cin + chunk * CK) + lane),
                                           (__attribute__((address_space(3))) void*)(s_coef + (chunk & 1) * CK), 4, 0, 0);
        return 1;
      }
    }
    return 0;
  };
  // normalise one piece of a staged halo chunk in place (same lane -> (row, slot) map as issue_halo_piece)
  auto transform_piece = [&](int chunk, int it, unsigned char* hb) {
    if constexpr (GNA) {
      const int q = it * NT + tid;
      const int row = q >> 2, slot = q & 3;
      const int c = slot ^ swz_key(row);
      const int hy = row / p.HTW, hx = row - hy * p.HTW;
      const int iy = iy0 + hy, ix = ix0 + hx;
      const bool ok = (row < halo_rows) & (iy >= 0) & (iy < p.H) & (ix >= 0) & (ix < p.Wd);
      if (ok) {
        Chunk16* ptr = (Chunk16*)(hb + (size_t)q * 16);
        const f32x2* cf = s_coef + (chunk & 1) * CK + c * VEC;
        const Chunk16 raw = *ptr;
        if constexpr (sizeof(T) == 2) {
          const bf16x8 x = __builtin_bit_cast(bf16x8, raw);
          bf16x8 o;
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = (bf16_t)silu_f(fmaf((float)x[j], cf[j][0], cf[j][1]));
          *ptr = __builtin_bit_cast(Chunk16, o);
        } else {
          const f32x4 x = __builtin_bit_cast(f32x4, raw);
          f32x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = silu_f(fmaf(x[j], cf[j][0], cf[j][1]));
          *ptr = __builtin_bit_cast(Chunk16, o);
        }
      }
    }
  };

  // ---- per-lane pixel bookkeeping: fragment fm covers pixels wm*64 + fm*16 + lr of the tile
  int py[FM], px[FM];
#pragma unroll
  for (int b = 0; b < FM; ++b) {
    const int pl = wm * 64 + b * 16 + lr;
    py[b] = pl >> p.tw_shift;
    px[b] = pl & (p.TW - 1);
  }

  f32x4 acc[FN][FM];
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < FM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nchunks = p.Cin / CK;
  const int steps = nchunks * 9;
  const bool dbg = p.dbg && blockIdx.x == 8 && tid == 0;
  int dbg_n = 0;
  auto stamp = [&]() { if (dbg) p.dbg[dbg_n++] = (long long)__builtin_readcyclecounter(); };
  stamp();
  // Pipeline: the weight tile of step s+2 and one piece of the next chunk's halo are issued at the top of step s;
  // the end-of-step wait is COUNTED (all DMA except what this step just issued), so a transfer has two full steps
  // to land and the barrier never drains the queue (cdna_hip_programming.md 5, "Pipelining across barriers").
  const int W_IT = (wave * 64 < BN * 4) ? 1 : 0;   // DMA instructions this wave issues per weight tile
  for (int it = 0; it < halo_iters; ++it) issue_halo_piece(0, it, hbuf0);
  issue_w(0, wbuf0);
  if (steps > 1) issue_w(1, wbuf0 + W_BYTES);
  (void)load_coef(0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if constexpr (GNA) {   // the first chunk is normalised before its first tap; later chunks under the taps of their predecessor
    for (int it = 0; it < halo_iters; ++it) transform_piece(0, it, hbuf0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  stamp();
  for (int s = 0; s < steps; ++s) {
    const int chunk = s / 9, tap = s - chunk * 9;
    int issued = 0;
    // the two waves of a SIMD (w, w + 4) issue their DMA at different points of the step, so one of them is always
    // feeding the matrix pipe (an in-order wave cannot issue MFMAs while it is issuing LDS-DMA)
    auto issue_dma = [&]() {
      if constexpr (!(ABL & 8)) {
        if (s + 2 < steps) { issue_w(s + 2, wbuf0 + ((s + 2) % 3) * W_BYTES); issued += W_IT; }
      }
      if constexpr (!(ABL & 4)) {
        if (chunk + 1 < nchunks && tap < halo_iters) { issue_halo_piece(chunk + 1, tap, hbuf0 + ((chunk + 1) & 1) * p.hb_bytes); issued += 1; }
      }
    };
    const bool early = __builtin_amdgcn_readfirstlane(wave) < NW / 2;
    if (early) issue_dma();
    if constexpr (GNA) {
      if (chunk + 1 < nchunks) {
        if (tap == 0) issued += load_coef(chunk + 1);
        // piece `it` of the next chunk was requested at tap `it` and has landed by the end of tap `it + 1`: normalise it at
        // tap 4 + it (visible to everyone after that step's barrier, long before the chunk's first tap)
        if (tap >= 4 && tap - 4 < halo_iters) transform_piece(chunk + 1, tap - 4, hbuf0 + ((chunk + 1) & 1) * p.hb_bytes);
      }
    }
    const unsigned char* hb = hbuf0 + (chunk & 1) * p.hb_bytes;
    const unsigned char* wb = wbuf0 + (s % 3) * W_BYTES;
    const int kh = tap / 3, kw = tap - kh * 3;
    int hr[FM];
#pragma unroll
    for (int b = 0; b < FM; ++b) {
      if constexpr (UPS) hr[b] = (((y0 + py[b] + kh - 1) >> 1) - iy0) * p.HTW + (((x0 + px[b] + kw - 1) >> 1) - ix0);
      else hr[b] = (py[b] + kh) * p.HTW + (px[b] + kw);
    }
    {
      if (!early) issue_dma();
      const int c = lg;
      Chunk16 xa[FM], wv[FN];
#pragma unroll
      for (int b = 0; b < FM; ++b) xa[b] = *(const Chunk16*)(hb + swz(hr[b], c));
#pragma unroll
      for (int a = 0; a < FN; ++a) wv[a] = *(const Chunk16*)(wb + swz(wn * WN + a * 16 + lr, c));
      if constexpr (ABL & 1) {
#pragma unroll
        for (int b = 0; b < FM; ++b) asm volatile("" :: "v"(xa[b]));
#pragma unroll
        for (int a = 0; a < FN; ++a) asm volatile("" :: "v"(wv[a]));
      }
      if constexpr (!(ABL & 1))
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b) {
          if constexpr (sizeof(T) == 2) {
            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wv[a]), __builtin_bit_cast(bf16x8, xa[b]),
                                                                acc[a][b], 0, 0, 0);
          } else {
            const f32x4 wf = __builtin_bit_cast(f32x4, wv[a]), xf = __builtin_bit_cast(f32x4, xa[b]);
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u], xf[u], acc[a][b], 0, 0, 0);
          }
        }
    }
    if (s < 16) stamp();
    // everything issued BEFORE this step has landed once at most `issued` transfers are still in flight
    if (issued == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    else if (issued == 1) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
    else if (issued == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (s < 16) stamp();
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  stamp();

  // ---- epilogue (same contract as igemm.hip): lane holds 4 consecutive n of pixel (py, px)
  const int flags = p.flags;
  // dense NHWC output of element type T: stage the PT x BN tile through LDS (the halo buffers are free now) and store
  // whole pixel rows, 16 B per lane, instead of 8-byte pieces at a 256-byte stride (store-issue bound otherwise)
  const bool staged = !(flags & IG_OUT_F32) && p.c_ch == 1 && p.c_pix == p.N && (p.N % BN) == 0 && p.stage_ok;
  constexpr int PITCH = BN * (int)sizeof(T) + 16;   // bytes per staged pixel row (+16: spreads the 16 pixel rows of a fragment over banks)
  // GroupNorm statistics of what this workgroup stores (the consumer's GroupNorm then needs no pass of its own over the tensor):
  // per lane the sums over its FM pixels of every channel it owns, reduced over the 16 pixel lanes, the NW / 2 pixel waves and
  // finally the channels of a group -- all in a fixed order
  const bool gn = p.gn_part != nullptr;
  float gs[FN][4], gq[FN][4];
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) { gs[a][r] = 0.f; gq[a][r] = 0.f; }
#pragma unroll
  for (int b = 0; b < FM; ++b) {
    const int pix = (y0 + py[b]) * p.Wo + (x0 + px[b]);
    const long obase = (long)(img / p.c_grp) * p.c_grp_stride + (long)(img % p.c_grp) * p.c_img + (long)pix * p.c_pix;
#pragma unroll
    for (int a = 0; a < FN; ++a) {
      const int n0 = n_base + wn * WN + a * 16 + lg * 4;
      if (n0 >= p.N) continue;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[a][b][r];
      if (flags & IG_BIAS_N) {
#pragma unroll
        for (int r = 0; r < 4; ++r) if (n0 + r < p.N) v[r] += p.bias[n0 + r];
      }
      const bool vec_ok = (p.c_ch == 1) && (n0 + 3 < p.N) && ((p.c_pix & 3) == 0);
      const long o = obase + (long)n0 * p.c_ch;
      if (flags & IG_RESIDUAL) {
        const T* R = (const T*)p.R;
        if (vec_ok && ((o & 3) == 0)) {
          if constexpr (sizeof(T) == 2) {
            const bf16x4 rv = *(const bf16x4*)(R + o);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
          } else {
            const f32x4 rv = *(const f32x4*)(R + o);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += rv[r];
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (n0 + r < p.N) v[r] += to_f32(R[o + (long)r * p.c_ch]);
        }
      }
      if (flags & IG_SILU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = silu_f(v[r]);
      }
      if (gn) {   // statistics of the STORED values (rounded to the output type, as a separate pass over the tensor would see them)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float f = (flags & IG_OUT_F32) ? v[r] : to_f32(from_f32<T>(v[r]));
          if (n0 + r < p.N) { gs[a][r] += f; gq[a][r] = fmaf(f, f, gq[a][r]); }
        }
      }
      if (staged) {
        unsigned char* dst = smem + (wm * 64 + b * 16 + lr) * PITCH + (wn * WN + a * 16 + lg * 4) * (int)sizeof(T);
        if constexpr (sizeof(T) == 2) *(bf16x4*)dst = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        else *(f32x4*)dst = f32x4{v[0], v[1], v[2], v[3]};
      } else if (flags & IG_OUT_F32) {
        float* Y = (float*)p.Y;
        if (vec_ok && ((o & 3) == 0)) *(f32x4*)(Y + o) = f32x4{v[0], v[1], v[2], v[3]};
        else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (n0 + r < p.N) Y[o + (long)r * p.c_ch] = v[r];
        }
      } else {
        T* Y = (T*)p.Y;
        if (vec_ok && ((o & 3) == 0)) {
          if constexpr (sizeof(T) == 2) *(bf16x4*)(Y + o) = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
          else *(f32x4*)(Y + o) = f32x4{v[0], v[1], v[2], v[3]};
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) if (n0 + r < p.N) Y[o + (long)r * p.c_ch] = from_f32<T>(v[r]);
        }
      }
    }
  }
  if (gn) {
    float* ch_s = (float*)(smem + p.gn_off);          // [NW / 2 pixel waves][BN channels]
    float* ch_q = ch_s + (NW / 2) * BN;
#pragma unroll
    for (int a = 0; a < FN; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s1 = row16_sum(gs[a][r]), s2 = row16_sum(gq[a][r]);   // over the 16 pixel lanes (DPP: no LDS traffic)
        if (lr == 0) { ch_s[wm * BN + wn * WN + a * 16 + lg * 4 + r] = s1; ch_q[wm * BN + wn * WN + a * 16 + lg * 4 + r] = s2; }
      }
  }
  if (staged || gn) __syncthreads();
  if (gn && tid < p.gn_groups) {
    const float* ch_s = (const float*)(smem + p.gn_off);
    const float* ch_q = ch_s + (NW / 2) * BN;
    const int cpg = p.N / p.gn_groups;
    const int c0 = max(tid * cpg, n_base), c1 = min(min((tid + 1) * cpg, n_base + BN), p.N);
    double a1 = 0.0, a2 = 0.0;
    for (int c = c0; c < c1; ++c)
      for (int w = 0; w < NW / 2; ++w) { a1 += (double)ch_s[w * BN + c - n_base]; a2 += (double)ch_q[w * BN + c - n_base]; }
    const long chunk = (long)t_in * p.tiles_n + tile_n;
    p.gn_part[((long)img * p.tiles_per_img * p.tiles_n + chunk) * p.gn_groups + tid] = double2{a1, a2};
  }
  if (staged) {
    constexpr int CPR = BN * (int)sizeof(T) / 16;   // 16-byte chunks per staged pixel row
    T* Y = (T*)p.Y;
    const long ibase = (long)(img / p.c_grp) * p.c_grp_stride + (long)(img % p.c_grp) * p.c_img;
    for (int q = tid; q < PT * CPR; q += NT) {
      const int pl = q / CPR, ch = q - pl * CPR;
      const int oy = y0 + (pl >> p.tw_shift), ox = x0 + (pl & (p.TW - 1));
      const Chunk16 val = *(const Chunk16*)(smem + pl * PITCH + ch * 16);
      *(Chunk16*)(Y + ibase + (long)(oy * p.Wo + ox) * p.c_pix + n_base + ch * (16 / (int)sizeof(T))) = val;
    }
  }
  if (dbg) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(); p.dbg[63] = dbg_n; }
}

template <typename T, int BN, bool UPS, int ABL = 0, int NW = 8>
static int launch_c3(const Conv3Dev& d, int nimg, hipStream_t stream) {
  int smem = 2 * d.hb_bytes + 3 * BN * 64;
  const int stage = NW * 32 * (BN * (int)sizeof(T) + 16);   // LDS-staged epilogue tile
  Conv3Dev dd = d;
  dd.stage_ok = stage <= (NW == 8 ? 80 : 160) * 1024;        // NW = 8 keeps two workgroups per CU (fp32 x 128 channels stores directly)
  if (dd.stage_ok && smem < stage) smem = stage;
  if (d.gn_part) { dd.gn_off = (smem + 15) & ~15; smem = dd.gn_off + 2 * (NW / 2) * BN * 4; }
  if constexpr ((ABL & 16) != 0) { dd.coef_off = (smem + 15) & ~15; smem = dd.coef_off + 2 * (4 * Traits<T>::VEC) * 8; }
  static unsigned long long attr_set = 0;
  auto kfn = conv3x3_kernel<T, BN, UPS, ABL, NW>;
  if (first_time_on_device(attr_set)) {
    hipError_t e = hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return (int)e;
  }
  const long blocks = (long)nimg * d.tiles_per_img * d.tiles_n;
  hipLaunchKernelGGL(kfn, dim3((unsigned)blocks), dim3(NW * 64), smem, stream, dd);
  return (int)hipGetLastError();
}

int conv3x3_gn_chunks_bound(int Hout, int Wout, int N) { return cdiv((long)Hout * Wout, 256) * cdiv(N, 64); }

bool conv3x3_enabled() {
  static int v = -1;
  if (v < 0) { const char* e = getenv("IVG_CONV3X3"); v = (e && e[0] == '0') ? 0 : 1; }
  return v == 1;
}

// Measured choice between the two tile sizes (tools/conv_bench.py, IVG_C3_NW): filled in from the sweep.
static bool conv3x3_prefers_16(int cin, int ho, bool ups) {
  (void)cin; (void)ho; (void)ups;
  return false;
}

// Returns -1 when the shape is not covered (caller falls back to the generic implicit GEMM).
int launch_conv3x3(const IgemmArgs& a, DType dtype, hipStream_t stream) {
  if (!conv3x3_enabled()) return -1;
  if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1) return -1;
  if (a.nb0 * a.nb1 * a.nb2 != 1 || a.alpha != 1.0f || (a.flags & (IG_GLU | IG_BIAS_M))) return -1;
  const int ck = dtype == BF16 ? 32 : 16;
  if (a.Cin % ck != 0 || a.ldx != a.Cin) return -1;
  const int Ho = a.Hout, Wo = a.Wout;
  if (a.ups ? (Ho != 2 * a.Hin || Wo != 2 * a.Win) : (Ho != a.Hin || Wo != a.Win)) return -1;
  int TW = Wo >= 32 ? 32 : Wo;   // 16x16 or 8x32 output tiles: halo <= 10 x 34 pixels = 49 KiB per buffer
  if (TW != 16 && TW != 32) return -1;
  const int bn = a.N > 64 ? 128 : 64;
  // 16-wave / 512-pixel (16 x 32) tiles: bf16, 128-channel N tiles, images at least 16 x 32 (IVG_C3_NW=8 / 16 forces)
  static int nw_env = -1;
  if (nw_env < 0) { const char* e = getenv("IVG_C3_NW"); nw_env = e ? atoi(e) : 0; }
  const bool can16 = dtype == BF16 && bn == 128 && TW == 32 && Ho % 16 == 0 && Wo % 32 == 0;
  const bool nw16 = can16 && (nw_env == 16 || (nw_env == 0 && conv3x3_prefers_16(a.Cin, Ho, a.ups != 0)));
  const int TH = (nw16 ? 512 : 256) / TW;
  if (Wo % TW != 0 || Ho % TH != 0) return -1;
  Conv3Dev d;
  d.X = a.X; d.W = a.W; d.Y = a.Y; d.R = a.R; d.bias = a.bias;
  d.H = a.Hin; d.Wd = a.Win; d.Cin = a.Cin; d.Ho = Ho; d.Wo = Wo;
  d.TH = TH; d.TW = TW; d.tw_shift = TW == 32 ? 5 : 4;
  if (a.ups) { d.HTH = TH / 2 + 2; d.HTW = TW / 2 + 2; }
  else { d.HTH = TH + 2; d.HTW = TW + 2; }
  d.tiles_x = Wo / TW; d.tiles_per_img = d.tiles_x * (Ho / TH); d.n_sp = a.Nimg * d.tiles_per_img;
  d.N = a.N; d.ldw = a.ldw;
  d.tiles_n = cdiv(a.N, bn);
  d.c_img = a.c_img; d.c_pix = a.c_pix; d.c_ch = a.c_ch; d.c_grp = a.c_grp > 0 ? a.c_grp : 1; d.c_grp_stride = a.c_grp_stride;
  if (a.c_grp <= 1 && a.c_grp_stride == 0) d.c_grp_stride = a.c_img;
  d.flags = a.flags;
  d.gn_part = nullptr; d.gn_groups = 0; d.gn_off = 0;
  d.in_coef = nullptr; d.coef_off = 0;
  const bool gna = a.gn_in_coef != nullptr;
  if (gna && (a.ups || nw16)) return -1;   // (the upsampling convs take un-normalised inputs; the 16-wave variant is not instantiated)
  d.in_coef = (const f32x2*)a.gn_in_coef;
  if (a.gn_part && a.gn_groups > 0 && a.gn_groups <= 64 && a.N % a.gn_groups == 0 && !nw16 && (a.c_grp <= 1)) {
    d.gn_part = (double2*)a.gn_part; d.gn_groups = a.gn_groups;
    a.gn_chunks = d.tiles_per_img * d.tiles_n;
  } else {
    a.gn_chunks = 0;
  }
  d.hb_bytes = nw16 ? cdiv(d.HTH * d.HTW * 4, 1024) * 16384 : cdiv(d.HTH * d.HTW * 4, 512) * 8192;
  {
    static long long* dbg_buf = nullptr;
    static int want = -1;
    if (want < 0) { const char* e = getenv("IVG_C3_DEBUG"); want = (e && e[0] == '1') ? 1 : 0; if (want) (void)hipMalloc((void**)&dbg_buf, 64 * 8); }
    d.dbg = dbg_buf;
    if (want) {
      static int calls = 0;
      if (++calls == 8) {  // after warm-up: dump the previous launch's stamps
        long long h[64]; (void)hipDeviceSynchronize(); (void)hipMemcpy(h, dbg_buf, sizeof(h), hipMemcpyDeviceToHost);
        const int n = (int)h[63];
        fprintf(stderr, "[c3 dbg] stamps=%d total=%lld cycles: prologue %lld", n, h[n - 1] - h[0], h[1] - h[0]);
        for (int i = 2; i + 1 < n - 2; i += 2) fprintf(stderr, " | step %d: compute %lld wait+bar %lld", (i - 2) / 2, h[i] - h[i - 1], h[i + 1] - h[i]);
        fprintf(stderr, " | drain %lld epilogue %lld\n", h[n - 2] - h[n - 3], h[n - 1] - h[n - 2]);
      }
    }
  }
  if (2 * d.hb_bytes + 3 * bn * 64 > (nw16 ? 160 : 80) * 1024) { a.gn_chunks = 0; return -1; }
  if (((uintptr_t)a.X & 15) || ((uintptr_t)a.W & 15)) { a.gn_chunks = 0; return -1; }
  {  // development ablations of the bf16 / BN = 128 / no-upsample instance (IVG_C3_ABLATE=<mask>)
    static int abl = -1;
    if (abl < 0) { const char* e = getenv("IVG_C3_ABLATE"); abl = e ? atoi(e) : 0; }
    if (abl && dtype == BF16 && bn == 128 && !a.ups) {
      switch (abl) {
        case 1: return launch_c3<bf16_t, 128, false, 1>(d, a.Nimg, stream);
        case 2: return launch_c3<bf16_t, 128, false, 2>(d, a.Nimg, stream);
        case 4: return launch_c3<bf16_t, 128, false, 4>(d, a.Nimg, stream);
        case 8: return launch_c3<bf16_t, 128, false, 8>(d, a.Nimg, stream);
        case 12: return launch_c3<bf16_t, 128, false, 12>(d, a.Nimg, stream);
        case 14: return launch_c3<bf16_t, 128, false, 14>(d, a.Nimg, stream);
        default: break;
      }
    }
  }
  if (nw16) return a.ups ? launch_c3<bf16_t, 128, true, 0, 16>(d, a.Nimg, stream) : launch_c3<bf16_t, 128, false, 0, 16>(d, a.Nimg, stream);
  if (gna) {
    if (dtype == BF16) return bn == 128 ? launch_c3<bf16_t, 128, false, 16>(d, a.Nimg, stream) : launch_c3<bf16_t, 64, false, 16>(d, a.Nimg, stream);
    return bn == 128 ? launch_c3<float, 128, false, 16>(d, a.Nimg, stream) : launch_c3<float, 64, false, 16>(d, a.Nimg, stream);
  }
#define IVG_C3(T, BNv) (a.ups ? launch_c3<T, BNv, true>(d, a.Nimg, stream) : launch_c3<T, BNv, false>(d, a.Nimg, stream))
  if (dtype == BF16) return bn == 128 ? IVG_C3(bf16_t, 128) : IVG_C3(bf16_t, 64);
  return bn == 128 ? IVG_C3(float, 128) : IVG_C3(float, 64);
#undef IVG_C3
}

}  // namespace ivg
