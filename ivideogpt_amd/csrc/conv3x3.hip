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
//     and retired by COUNTED s_waitcnt vmcnt(n) + raw s_barrier (the DMA queue is never drained inside the loop); the plain
//     bf16 instances run TWO steps per barrier on a ring of two double slots requested one pair ahead,
//   * both by LDS-DMA (global_load_lds, 16 B/lane; out-of-image pixels are never requested, their LDS slots are zeroed
//     once), 64-byte LDS rows with an XOR swizzle on the SOURCE chunk (conflict-free ds_read_b128 from any start row),
//   * 72-80 KiB of LDS per workgroup -> TWO workgroups (16 waves) per CU: measured with cycle stamps, one workgroup per CU
//     spends 25 % of every step in the barrier and 22 % of its life in an un-overlapped prologue / epilogue; a second
//     resident workgroup fills exactly those holes,
//   * 8 waves (2 per SIMD), each a 64 x 64 (or 64 x 32) sub-tile of MFMA fragments, swapped operands so a lane owns
//     4 consecutive output channels;  the XCD-aware block order keeps the N tiles of one spatial tile on one L2,
//   * the step loop issues (almost) nothing but MFMAs, LDS reads and DMA: a wave64 VALU instruction occupies its SIMD for
//     4 clocks and a bf16 step is only 16 MFMAs x 16 clocks per wave, so the ~110 address / predicate instructions per step
//     of the first version of this loop (integer division by the halo width, swizzle keys, bounds tests, 64-bit pointer
//     arithmetic, all per lane and per step) kept the vector pipe busier than the matrix pipe (the fp32 instances, 2048 MFMA
//     clocks per step, hid the same overhead and ran at 70-80 % of their peak).  Now the tile geometry is a template
//     parameter and the nine taps are unrolled: every LDS fragment address is one per-lane base register (per kw, set up
//     once) plus an immediate offset, and every DMA source is a per-lane 32-bit offset (set up once) from a base the scalar
//     unit advances.  Register allocation of this loop is tight (acc 64 + fragments 32 + ~10 addresses of 128): check
//     `-Rpass-analysis=kernel-resource-usage` for spills after any change -- a reload shares the counter with the LDS-DMA
//     queue and drains it.
// Bytes per (tap, chunk) step: 8 KiB of weights + 1/9 of a ~24 KiB halo for 2.1 MFLOP -> ~195 FLOP/B (igemm: 64).
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "conv3x3_common.h"
#include "switches.h"

namespace ivg {

// GNA: GroupNorm(+SiLU) of the input fused into the staging -- the halo chunk is normalised IN PLACE in LDS
// (y = silu(x * scale[c] + shift[c]), out-of-image padding stays zero) between its arrival and its first tap, piece by piece
// under the taps of the previous chunk, so the normalised tensor never exists in HBM.  ON by default
// (tokenizer.cpp: norm_conv; IVG_GN_APPLY_FUSE=0 restores the separate apply pass): -2.2 ms per config-2 step with the
// rebuilt loop (profiles/r02_gn_apply_fusion_ab.txt; the same fusion cost +2.5 ms while the loop was instruction-bound).
// Round 3 measured two re-arrangements of the transform and kept neither: pair-by-pair / recomputed lane maps to get rid of
// the five spilled registers of the bf16 x 128-channel instances (the compiler spills elsewhere: 20 -> 20 / 272 bytes), and a
// schedule staggered over the taps and the two waves of a SIMD (waves 0-3 at taps 2, 4, 6, waves 4-7 at 3, 5, 7: 39.4-39.9 vs
// 38.6 ms of conv3x3 per step, profiles/r03_conv3x3_stagger_ab.txt).
// TPB2: two (tap, chunk) steps per workgroup barrier -- the weight ring holds two slots of two tiles and is refilled one PAIR of
// steps ahead, the fragment registers are reused by the second step; 80 KiB of LDS, still two workgroups per CU.
// (Reading the halo fragments of step s + 1 under the MFMAs of step s was measured: +-0.5 %, removed.)
// X3 (T = float only): "split-bf16" arithmetic for the 1e-3-compliant decode path.  Activations and weights stay fp32 in HBM;
// a staged halo slot (4 fp32 channels) is rewritten in place as [hi(4) | lo(4)] bf16 with hi = bf16(x), lo = bf16(x - hi) -- the same
// 16 bytes -- and the weights arrive pre-split in the same slot layout (packing.py: pack_x3).  One K = 32 bf16 MFMA then multiplies
// a slot's [a_hi | a_lo] by [w_hi | w_hi], a second one by [w_lo | w_lo]: all four partial products, fp32 accumulate, for 2 x 16
// MFMA clocks per 16 channels where the f32-input MFMA path needs 4 x 32 -- 2^-17 relative per operand instead of 2^-24 (far inside
// the 1e-3 bar on pixels; NOT used by tokenize, whose bit-exact ids need the exact fp32 chain).  Same LDS geometry as the fp32 instances.
// SUBPIX (round 6): the nearest-x2 upsampling convolution in SUB-PIXEL form.  A 3x3 convolution over a nearest-x2 upsampled image is
// exactly four 2x2 convolutions over the low-resolution image, one per output-pixel parity (py, px), with pre-summed weights: output
// row 2 iy + py reads upsampled rows 2 iy + py - 1 .. 2 iy + py + 1 = input rows {iy - 1, iy, iy} (py = 0) or {iy, iy, iy + 1}
// (py = 1), so the row taps collapse to {W0, W1 + W2} resp. {W0 + W1, W2}; columns alike; the zero padding of the upsampled image
// maps onto the zero padding of the input.  16 instead of 36 tap-GEMMs per input pixel: 2.25 x fewer multiplies for the same
// result (the weights are pre-summed in fp32 at pack time, packing.py pack_subpixel: [phase][N][tap (kh2, kw2)][Cin]).  A workgroup
// owns a TH x TW tile of INPUT pixels, ONE phase and BN channels: the plain kernel's halo tile and fragment addressing with the tap
// set {(py + kh2, px + kw2)} -- four steps per channel chunk on a ring of four weight slots -- and an epilogue that stores to the
// pixels (2 y + py, 2 x + px).  The four phases of a spatial tile are adjacent in the XCD-aware block order (one halo in L2).
template <typename T, int BN, bool UPS, int TW, bool GNA, bool TPB2, bool X3 = false, bool SUBPIX = false>
__global__ __launch_bounds__(512, 4) void conv3x3_kernel(const Conv3Dev p) {
  static_assert(!X3 || sizeof(T) == 4, "split-bf16 arithmetic reads fp32 tensors");
  static_assert(!SUBPIX || (!UPS && !GNA && !TPB2), "the sub-pixel instances are plain-geometry, one step per barrier, un-normalised input");
  constexpr int NT = SUBPIX ? 4 : 9;       // taps (steps) per channel chunk
  constexpr int RING = SUBPIX ? 4 : 3;     // weight ring slots of the one-step loop (slot of step s = s % RING must be a function of the tap)
  constexpr int VEC = Traits<T>::VEC;
  constexpr int CK = 4 * VEC;              // channels per chunk: one 64-byte LDS row per halo pixel (one MFMA K-step)
  // 8 waves = 4 (pixel groups of 64) x 2 (channel halves); BN = 16 (round 6: the decoders' tail, 3 output channels -- the 64-channel
  // instance computed 61 channels nobody stores): 8 pixel groups of 32 x one 16-channel fragment
  constexpr int PG = BN == 16 ? 8 : 4;     // pixel groups (waves along the tile's pixels)
  constexpr int WN = BN == 16 ? 16 : BN / 2;
  constexpr int FM = 256 / PG / 16, FN = WN / 16;
  constexpr int W_BYTES = BN * 64;
  // ---- tile geometry
  constexpr int TH = 256 / TW, TWS = TW == 32 ? 5 : 4;
  constexpr int HTW = UPS ? TW / 2 + 2 : TW + 2, HTH = UPS ? TH / 2 + 2 : TH + 2;   // halo tile (input pixels)
  constexpr int HROWS = HTH * HTW;
  constexpr int HI = (HROWS * 4 + 511) / 512;      // DMA pieces (512 lanes x 16 B) per halo chunk
  constexpr int HB = HI * 8192;                    // one halo buffer
  // fragment b of a wave covers tile row pyw + BR(b), columns PXO(b) + lr
  auto BR = [](int b) constexpr { return TW == 16 ? b : (b >> 1); };
  auto PXO = [](int b) constexpr { return TW == 16 ? 0 : (b & 1) * 16; };
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* hbuf0 = smem;
  unsigned char* wbuf0 = smem + 2 * HB;   // three weight buffers (four: sub-pixel instances and the two-step loop)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform values stay in scalar registers
  const int lr = lane & 15, lg = lane >> 4;
  const int wm = BN == 16 ? wave : (wave & 3), wn = BN == 16 ? 0 : (wave >> 2);
  const int pyw = wm * (256 / PG / TW);    // first tile row of this wave's pixels (64 / TW rows per group; 32 / TW for BN = 16)

  // ---- XCD-aware block order (blocks b, b+8, ... share an XCD/L2): give each XCD a contiguous run of work items
  // with the N tile fastest, so the N tiles of one spatial tile hit the same L2 (bijective for any grid size)
  const int nwg = gridDim.x;
  int v;
  {
    const int b = blockIdx.x, xcd = b & 7, q = nwg >> 3, r = nwg & 7;
    v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
  }
  const int tile_n = v % p.tiles_n;
  const int phase = SUBPIX ? (v / p.tiles_n) & 3 : 0;   // (py, px) = (phase >> 1, phase & 1): parity of the output pixels this workgroup writes
  const int sp = SUBPIX ? (v / p.tiles_n) >> 2 : v / p.tiles_n;
  const int img = sp / p.tiles_per_img;
  const int t_in = sp - img * p.tiles_per_img;
  const int ty = t_in / p.tiles_x, tx = t_in - ty * p.tiles_x;
  const int y0 = ty * TH, x0 = tx * TW;                // output tile origin
  const int iy0 = UPS ? ((y0 - 1) >> 1) : (y0 - 1);    // halo origin in input pixels
  const int ix0 = UPS ? ((x0 - 1) >> 1) : (x0 - 1);
  const unsigned char* Xb = (const unsigned char*)((const T*)p.X + (long)img * p.H * p.Wd * p.Cin);   // this image, chunk 0
  const int n_base = tile_n * BN;

  // ---- per-lane DMA sources, set up once: a 32-bit byte offset per lane from a wave-uniform base that the scalar unit
  // advances (chunk / tap), so a transfer costs no vector instruction beyond the load itself.
  // Halo piece `it`: lane -> (halo pixel, 16-byte slot).  Out-of-image pixels (and the tail of the last piece) are never
  // requested: their LDS slots -- the same in every chunk -- are zeroed once here, in both buffers.
  unsigned hoff[HI];
  bool hok[HI];
  int h_any[HI];     // this wave requests anything in piece `it` (wave-uniform; the counted waits need the exact number)
#pragma unroll
  for (int it = 0; it < HI; ++it) {
    const int q = it * 512 + tid;
    const int row = q >> 2, slot = q & 3;
    const int hy = row / HTW, hx = row - hy * HTW;
    const int c = slot ^ halo_key<UPS>(hx);
    const int iy = iy0 + hy, ix = ix0 + hx;
    hok[it] = (row < HROWS) & (iy >= 0) & (iy < p.H) & (ix >= 0) & (ix < p.Wd);
    // (24-bit multiply: pixel index < 2^24, bytes per pixel < 2^24 -- keeps the offset a single 32-bit register; the generic
    // form is selected as a 64-bit multiply-add whose register pair stays allocated through the loop)
    hoff[it] = hok[it] ? (unsigned)__umul24((unsigned)(iy * p.Wd + ix), (unsigned)(p.Cin * (int)sizeof(T))) + (unsigned)(c * VEC * (int)sizeof(T)) : 0u;
    h_any[it] = __builtin_amdgcn_readfirstlane(__builtin_amdgcn_ballot_w64(hok[it]) != 0 ? 1 : 0);
    if (!hok[it]) {
      *(Chunk16*)(hbuf0 + (size_t)q * 16) = Chunk16{0, 0, 0, 0};
      *(Chunk16*)(hbuf0 + HB + (size_t)q * 16) = Chunk16{0, 0, 0, 0};
    }
  }
  auto issue_halo_piece = [&](int it, int chunk, unsigned char* hb) {
    if (hok[it]) glds16s(Xb + (size_t)chunk * (CK * sizeof(T)), hoff[it], lds_addr(hb) + (it * 512 + wave * 64) * 16);
  };
  // weight tile of one (tap, chunk): BN rows x 4 slots; rows beyond N re-read row N - 1 (their outputs are never stored).
  // BN = 64 is 256 lanes: waves 4..7 repeat the transfers of waves 0..3 (same bytes to the same place) so that every wave
  // counts the same number of outstanding transfers.
  unsigned woff;
  {
    const int q = BN == 128 ? tid : (BN == 64 ? (tid & 255) : (tid & 63));   // (BN = 16: one 1 KiB transfer, repeated by every wave)
    const int n = q >> 2, slot = q & 3;
    const int c = slot ^ swz_key(n);
    const int nrow = min(n_base + n, p.N - 1);
    woff = (unsigned)((SUBPIX ? phase * p.N + nrow : nrow) * p.ldw) * (unsigned)sizeof(T) + (unsigned)(c * VEC * (int)sizeof(T));
  }
  const int w_dst = (BN == 128 ? wave : (BN == 64 ? (wave & 3) : 0)) * 1024;
  auto issue_w = [&](int tap, int chunk, int ring) {
    // (the 64-byte-per-row shape of these requests is not what bounds the kernel: requesting the same bytes as whole 128-byte
    // lines -- a probe with wrong results -- changed nothing, profiles/r02_conv3x3_wline_probe.txt)
    glds16s((const unsigned char*)p.W + (size_t)(tap * p.Cin + chunk * CK) * sizeof(T), woff, lds_addr(wbuf0) + ring * W_BYTES + w_dst);
  };

  f32x2* s_coef = (f32x2*)(smem + p.coef_off);   // [2][CK] (scale, shift) of the channels of the chunk being staged
  // the chunk's coefficient row (CK x 8 bytes) by one 4-byte-per-lane LDS-DMA of wave 0: no register-destination load may sit
  // beside the DMA queue (the compiler would drain it with vmcnt(0)); returns the DMA instructions this wave issued
  auto load_coef = [&](int chunk) -> int {
    if constexpr (GNA) {
      if (wave == 0) {
        if (lane < 2 * CK) glds4s(p.in_coef + ((long)img * p.Cin + chunk * CK), (unsigned)lane * 4u, lds_addr(s_coef + (chunk & 1) * CK));
        return 1;
      }
    }
    return 0;
  };
  // normalise one piece of a staged halo chunk in place (same lane -> (pixel, slot) map as the DMA)
  auto transform_piece = [&](int chunk, int it, unsigned char* hb) {
    if constexpr (X3 && !GNA) {   // split only: out-of-image slots hold zeros, whose split is zeros -- no bounds test needed
      Chunk16* ptr = (Chunk16*)(hb + (size_t)(it * 512 + tid) * 16);
      const f32x4 x = __builtin_bit_cast(f32x4, *ptr);
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 4; ++j) { const bf16_t hi = (bf16_t)x[j]; o[j] = hi; o[4 + j] = (bf16_t)(x[j] - (float)hi); }
      *ptr = __builtin_bit_cast(Chunk16, o);
    }
    if constexpr (GNA) {
      const int q = it * 512 + tid;
      const int row = q >> 2, slot = q & 3;
      const int hy = row / HTW, hx = row - hy * HTW;
      const int c = slot ^ halo_key<UPS>(hx);
      const int iy = iy0 + hy, ix = ix0 + hx;
      const bool ok = (row < HROWS) & (iy >= 0) & (iy < p.H) & (ix >= 0) & (ix < p.Wd);
      if (ok) {
        Chunk16* ptr = (Chunk16*)(hb + (size_t)q * 16);
        const f32x2* cf = s_coef + (chunk & 1) * CK + c * VEC;
        const Chunk16 raw = *ptr;
        if constexpr (sizeof(T) == 2) {
          const bf16x8 x = __builtin_bit_cast(bf16x8, raw);
          bf16x8 o;
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] = (bf16_t)silu_t<T>(fmaf((float)x[j], cf[j][0], cf[j][1]));
          *ptr = __builtin_bit_cast(Chunk16, o);
        } else {
          const f32x4 x = __builtin_bit_cast(f32x4, raw);
          f32x4 o;
#pragma unroll
          for (int j = 0; j < 4; ++j) o[j] = silu_t<T>(fmaf(x[j], cf[j][0], cf[j][1]));
          if constexpr (X3) {
            bf16x8 s8;
#pragma unroll
            for (int j = 0; j < 4; ++j) { const bf16_t hi = (bf16_t)o[j]; s8[j] = hi; s8[4 + j] = (bf16_t)(o[j] - (float)hi); }
            *ptr = __builtin_bit_cast(Chunk16, s8);
          } else {
            *ptr = __builtin_bit_cast(Chunk16, o);
          }
        }
      }
    }
  };

  // ---- per-lane LDS fragment addresses, set up once.  Halo: one base per kw (and per column half where the key differs);
  // the pixel row of fragment b and the kh tap are constant byte offsets.  Weights: one base; ring slot and the FN
  // 16-row groups are constant offsets.
  // Upsampling instances with 32-pixel rows: the second column half of a wave's fragments sits 8 halo columns further, where
  // the key differs in its low bit (bitrev2 of (column >> 2) + 2): its address is (base + 8 * 64) ^ 16 -- two instructions at
  // the read instead of three more registers that the loop cannot spare.
  constexpr bool XHALF = UPS && TW == 32;
  int a_base[3];   // byte offsets into halo buffer 0 (buffer 1: + HB, an immediate -- the chunk loop is unrolled by two)
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    int hx, rowbase;
    if constexpr (UPS) { hx = ((lr + kw - 1) >> 1) + 1; rowbase = (pyw >> 1) * HTW; }
    else if constexpr (SUBPIX) { hx = lr + kw + (phase & 1); rowbase = (pyw + (phase >> 1)) * HTW; }   // taps (py + kh2, px + kw2), kw < 2 used
    else { hx = lr + kw; rowbase = pyw * HTW; }
    a_base[kw] = (rowbase + hx) * 64 + ((lg ^ halo_key<UPS>(hx)) << 4);
  }
  auto a_imm = [&](int b, int kh) constexpr -> int {   // folds to an immediate after unrolling
    if constexpr (UPS) return ((((BR(b) + kh - 1) >> 1) + 1) * HTW) * 64;
    else return ((BR(b) + kh) * HTW + PXO(b)) * 64;
  };
  auto read_a = [&](int buf, int tap, int b) -> Chunk16 {
    const int kh = SUBPIX ? tap >> 1 : tap / 3, kw = SUBPIX ? tap & 1 : tap - kh * 3;
    int base = a_base[kw];
    if constexpr (XHALF) {
      if (b & 1) { asm volatile("" : "+v"(base)); base = (base + 8 * 64) ^ 16; }   // (opaque: or the compiler hoists the three results back into registers)
    }
    return *(const Chunk16*)(hbuf0 + base + (buf * HB + a_imm(b, kh)));
  };
  const int w_base = swz(wn * WN + lr, lg);

  f32x4 acc[FN][FM];
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int b = 0; b < FM; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nchunks = p.Cin / CK;
  // Pipeline: the weight tile of step s+2 and one piece of the next chunk's halo are issued at the top of step s;
  // the end-of-step wait is COUNTED (all DMA except what this step just issued), so a transfer has two full steps
  // to land and the barrier never drains the queue (cdna_hip_programming.md 5, "Pipelining across barriers").
#pragma unroll
  for (int it = 0; it < HI; ++it) issue_halo_piece(it, 0, hbuf0);
  issue_w(0, 0, 0);
  issue_w(1, 0, 1);   // (slot 1 of the three-slot ring = second half of slot 0 of the pair ring)
  (void)load_coef(0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if constexpr (GNA || X3) {   // the first chunk is normalised / split before its first tap; later chunks under the taps of their predecessor
    for (int it = 0; it < HI; ++it) transform_piece(0, it, hbuf0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  Chunk16 xa[FM];
  // the two waves of a SIMD (w, w + 4) issue their DMA at different points of the step, so one of them is always
  // feeding the matrix pipe (an in-order wave cannot issue MFMAs while it is issuing LDS-DMA)
  const bool early = __builtin_amdgcn_readfirstlane(wave) < 4;
  // one 32-channel chunk = nine steps; PAR (the halo buffer it is consumed from) is a compile-time value, so that buffer
  // addresses are immediates: the chunk loop below alternates the two instances
  auto run_chunk = [&](int chunk, auto par) {
    constexpr int PAR = decltype(par)::value;
    const bool more = chunk + 1 < nchunks;                       // another chunk follows: its halo is staged under this one
    unsigned char* hb_next = hbuf0 + (1 - PAR) * HB;
#pragma unroll
    for (int tap = 0; tap < NT; ++tap) {
      const int tap2 = (tap + 2) % NT, chunk2 = chunk + (tap + 2) / NT;   // the step whose weights are requested now
      const bool w_more = tap < NT - 2 || more;
      int issued = 0;
      auto issue_dma = [&]() {
        if (w_more) { issue_w(tap2, chunk2, (tap + 2) % RING); issued += 1; }
        if constexpr (SUBPIX) {
          // four steps per chunk: pieces 0 and 1 of the next chunk's halo at tap 0, piece 2 at tap 1 -- each has landed by the end of
          // the following step, so the split-bf16 instances can rewrite them at taps 2 and 3, before the chunk's last barrier
          static_assert(!SUBPIX || HI == 3, "halo pieces of the plain geometry");
          if (more) {
            if (tap == 0) { issue_halo_piece(0, chunk + 1, hb_next); issue_halo_piece(1, chunk + 1, hb_next); issued += h_any[0] + h_any[1]; }
            if (tap == 1) { issue_halo_piece(2, chunk + 1, hb_next); issued += h_any[2]; }
          }
        } else {
          if (tap < HI) { if (more) { issue_halo_piece(tap, chunk + 1, hb_next); issued += h_any[tap < HI ? tap : 0]; } }
        }
      };
      if (GNA || X3 || early) issue_dma();   // (the instances that transform the staged halo issue at the top in every wave: one code path less, no spills)
      if constexpr (GNA || X3) {
        if (more) {
          if (tap == 0) issued += load_coef(chunk + 1);
          // piece `it` of the next chunk was requested at tap `it` and has landed by the end of tap `it + 1`: normalise it at
          // tap 4 + it (visible to everyone after that step's barrier, long before the chunk's first tap)
          if constexpr (SUBPIX) {
            if (tap == 2) { transform_piece(chunk + 1, 0, hb_next); transform_piece(chunk + 1, 1, hb_next); }
            if (tap == 3) transform_piece(chunk + 1, 2, hb_next);
          } else {
            if (tap >= 4 && tap - 4 < HI) transform_piece(chunk + 1, tap - 4, hb_next);
          }
        }
      }
      if (!GNA && !X3 && !early) issue_dma();
      __builtin_amdgcn_sched_barrier(0);   // (as in the two-step loop: bounds the register pressure of the unrolled taps)
      Chunk16 wv[FN];
#pragma unroll
      for (int b = 0; b < FM; ++b) xa[b] = read_a(PAR, tap, b);
#pragma unroll
      for (int a = 0; a < FN; ++a) wv[a] = *(const Chunk16*)(wbuf0 + w_base + ((tap % RING) * W_BYTES + a * 1024));
      if constexpr (X3) {
        // W fragment outermost: one duplicated half ([w_hi | w_hi], then [w_lo | w_lo]) is live at a time -- built for all FN fragments
        // up front (the order the compiler prefers) it costs 32 registers the 128-channel instances do not have (13-31 spilled, and
        // a spill reload drains the DMA queue); an accumulator is revisited after FM MFMAs, beyond the dependent-issue latency
#pragma unroll
        for (int a = 0; a < FN; ++a) {
#pragma unroll
          for (int h = 0; h < 2; ++h) {
            Chunk16 wd = Chunk16{wv[a][2 * h], wv[a][2 * h + 1], wv[a][2 * h], wv[a][2 * h + 1]};
            asm volatile("" : "+v"(wd));   // (keeps the duplicate from being hoisted out of its four MFMAs)
#pragma unroll
            for (int b0 = 0; b0 < FM; ++b0) {
              const int b = ((a * 2 + h) & 1) ? FM - 1 - b0 : b0;   // snake order: the halo fragment stays when the weight half changes
              acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wd), __builtin_bit_cast(bf16x8, xa[b]), acc[a][b], 0, 0, 0);
            }
          }
        }
      } else {
      // SNAKE order over the (b, a) fragment tile: exactly ONE operand changes between consecutive MFMAs (row-major changes both at
      // every row end).  Same products into the same accumulators in the same K order -- bit-identical -- but the matrix pipe's input
      // toggling is what the socket's power limit prices: a pure random-operand stream sustains 2,061 TFLOP/s walked this way against
      // 2,023 row-major and 1,973 with both operands changing every time (tools/ubench/mfma_power.hip, profiles/r05_mfma_power.txt)
#pragma unroll
      for (int b = 0; b < FM; ++b) {
#pragma unroll
        for (int a0 = 0; a0 < FN; ++a0) {
          const int a = (b & 1) ? FN - 1 - a0 : a0;
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
      }
      // everything issued BEFORE this step has landed once at most `issued` transfers are still in flight
      if (issued == 0) wait_dma_keep<0>();
      else if (issued == 1) wait_dma_keep<1>();
      else if (issued == 2) wait_dma_keep<2>();
      else wait_dma_keep<3>();
      __builtin_amdgcn_s_barrier();
    }
  };
  if constexpr (!TPB2) {
    for (int chunk = 0; chunk < nchunks; chunk += 2) {
      run_chunk(chunk, std::integral_constant<int, 0>{});
      if (chunk + 1 < nchunks) run_chunk(chunk + 1, std::integral_constant<int, 1>{});
    }
  } else {
    static_assert(!((GNA || X3) && TPB2), "the instances that transform the staged halo run on the one-step-per-barrier loop");
    // Two chunks = 18 steps = nine PAIRS of steps, unrolled: step s of the group is (chunk cg + s / 9, tap s % 9) and reads halo
    // buffer (s / 9) & 1.  Weight ring: two slots of two tiles; the pair after this one is requested at the top of this pair
    // into the other slot (free since the barrier that ended the previous pair) and has landed at this pair's end -- the
    // end-of-pair wait lets only the halo piece requested in this pair stay in flight.  Halo of chunk cg + 1 (buffer 1, last
    // read in the previous group's last pair): pieces at pairs 0 .. HI-1, first needed in pair 4; halo of chunk cg + 2
    // (buffer 0, last read in the first half of pair 4): pieces at pairs 5 .. 5+HI-1, first needed in the next group's pair 0.
    int pair = 0;
    for (int cg = 0; cg < nchunks; cg += 2) {
      const bool has_c1 = cg + 1 < nchunks, has_c2 = cg + 2 < nchunks;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const int s0 = 2 * k, s1 = 2 * k + 1;
        if (s0 >= 9) { if (!has_c1) break; }
        const bool second = s1 < 9 || has_c1;
        const int slot = pair & 1;
        int issued_h = 0;
        auto issue_dma = [&]() {
          const int n0 = s0 + 2, n1 = s1 + 2;   // the next pair's steps (>= 18: the next group)
          const bool e0 = n0 < 9 ? true : (n0 < 18 ? has_c1 : has_c2);
          const bool e1 = n1 < 9 ? true : (n1 < 18 ? has_c1 : has_c2);
          const int nslot = (slot ^ 1) * 2;
          if (e0) issue_w(n0 % 9, cg + n0 / 9, nslot);
          if (e1) issue_w(n1 % 9, cg + n1 / 9, nslot + 1);
          if (k < HI) { if (has_c1) { issue_halo_piece(k, cg + 1, hbuf0 + HB); issued_h += h_any[k < HI ? k : 0]; } }
          if (k >= 5 && k - 5 < HI) { if (has_c2) { issue_halo_piece(k - 5, cg + 2, hbuf0); issued_h += h_any[(k >= 5 && k - 5 < HI) ? k - 5 : 0]; } }
        };
        const int w_cur = w_base + slot * 2 * W_BYTES;
        auto half_step = [&](int s, int half) {
          __builtin_amdgcn_sched_barrier(0);   // keep the next step's fragment reads from being hoisted over this step's MFMAs (registers)
          Chunk16 wv[FN];
#pragma unroll
          for (int b = 0; b < FM; ++b) xa[b] = read_a((s / 9) & 1, s % 9, b);
#pragma unroll
          for (int a = 0; a < FN; ++a) wv[a] = *(const Chunk16*)(wbuf0 + w_cur + (half * W_BYTES + a * 1024));
#pragma unroll
          for (int b = 0; b < FM; ++b)
#pragma unroll
            for (int a0 = 0; a0 < FN; ++a0) {
              const int a = (b & 1) ? FN - 1 - a0 : a0;   // snake order (see the one-step loop)
              if constexpr (sizeof(T) == 2) {
                acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wv[a]), __builtin_bit_cast(bf16x8, xa[b]),
                                                                    acc[a][b], 0, 0, 0);
              } else {
                const f32x4 wf = __builtin_bit_cast(f32x4, wv[a]), xf = __builtin_bit_cast(f32x4, xa[b]);
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[u], xf[u], acc[a][b], 0, 0, 0);
              }
            }
        };
        issue_dma();
        half_step(s0, 0);
        if (second) half_step(s1, 1);
        if (issued_h == 0) wait_dma_keep<0>(); else wait_dma_keep<1>();
        __builtin_amdgcn_s_barrier();
        ++pair;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

  // ---- epilogue (same contract as igemm.hip): lane holds 4 consecutive n of pixel (py, px)
  const int flags = p.flags;
  // dense NHWC output of element type T: stage the 256 x BN tile through LDS (the halo buffers are free now) and store
  // whole pixel rows, 16 B per lane, instead of 8-byte pieces at a 256-byte stride (store-issue bound otherwise)
  const bool staged = !(flags & IG_OUT_F32) && p.c_ch == 1 && p.c_pix == p.N && (p.N % BN) == 0 && p.stage_ok;
  constexpr int PITCH = BN * (int)sizeof(T) + 16;   // bytes per staged pixel row (+16: spreads the 16 pixel rows of a fragment over banks)
  // GroupNorm statistics of what this workgroup stores (the consumer's GroupNorm then needs no pass of its own over the tensor):
  // per lane the sums over its FM pixels of every channel it owns, reduced over the 16 pixel lanes, the 4 pixel waves and
  // finally the channels of a group -- all in a fixed order
  const bool gn = p.gn_part != nullptr;
  float gs[FN][4], gq[FN][4];
#pragma unroll
  for (int a = 0; a < FN; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) { gs[a][r] = 0.f; gq[a][r] = 0.f; }
#pragma unroll
  for (int b = 0; b < FM; ++b) {
    const int pix = SUBPIX ? (2 * (y0 + pyw + BR(b)) + (phase >> 1)) * p.Wo + 2 * (x0 + PXO(b) + lr) + (phase & 1)
                           : (y0 + pyw + BR(b)) * p.Wo + (x0 + PXO(b) + lr);
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
        for (int r = 0; r < 4; ++r) v[r] = silu_t<T>(v[r]);
      }
      if (flags & IG_CLAMP01) {   // the decoders' last convolution when the caller wants displayable frames (predict.py:73)
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = fminf(fmaxf(v[r], 0.f), 1.f);
      }
      if (gn) {   // statistics of the STORED values (rounded to the output type, as a separate pass over the tensor would see them)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float f = (flags & IG_OUT_F32) ? v[r] : to_f32(from_f32<T>(v[r]));
          if (n0 + r < p.N) { gs[a][r] += f; gq[a][r] = fmaf(f, f, gq[a][r]); }
        }
      }
      if (staged) {
        unsigned char* dst = smem + (wm * (FM * 16) + b * 16 + lr) * PITCH + (wn * WN + a * 16 + lg * 4) * (int)sizeof(T);
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
    float* ch_s = (float*)(smem + p.gn_off);          // [PG pixel waves][BN channels]
    float* ch_q = ch_s + (PG) * BN;
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
    const float* ch_q = ch_s + (PG) * BN;
    const int cpg = p.N / p.gn_groups;
    const int c0 = max(tid * cpg, n_base), c1 = min(min((tid + 1) * cpg, n_base + BN), p.N);
    double a1 = 0.0, a2 = 0.0;
    for (int c = c0; c < c1; ++c)
      for (int w = 0; w < PG; ++w) { a1 += (double)ch_s[w * BN + c - n_base]; a2 += (double)ch_q[w * BN + c - n_base]; }
    const long chunk = (long)(SUBPIX ? t_in * 4 + phase : t_in) * p.tiles_n + tile_n;
    p.gn_part[((long)img * p.tiles_per_img * (SUBPIX ? 4 : 1) * p.tiles_n + chunk) * p.gn_groups + tid] = double2{a1, a2};
  }
  if (staged) {
    constexpr int CPR = BN * (int)sizeof(T) / 16;   // 16-byte chunks per staged pixel row
    T* Y = (T*)p.Y;
    const long ibase = (long)(img / p.c_grp) * p.c_grp_stride + (long)(img % p.c_grp) * p.c_img;
    for (int q = tid; q < 256 * CPR; q += 512) {
      const int pl = q / CPR, ch = q - pl * CPR;
      const int oy = SUBPIX ? 2 * (y0 + (pl >> TWS)) + (phase >> 1) : y0 + (pl >> TWS);
      const int ox = SUBPIX ? 2 * (x0 + (pl & (TW - 1))) + (phase & 1) : x0 + (pl & (TW - 1));
      const Chunk16 val = *(const Chunk16*)(smem + pl * PITCH + ch * 16);
      *(Chunk16*)(Y + ibase + (long)(oy * p.Wo + ox) * p.c_pix + n_base + ch * (16 / (int)sizeof(T))) = val;
    }
  }
}

template <typename T, int BN, bool UPS, int TW, bool GNA, bool TPB2, bool X3 = false, bool SUBPIX = false>
static int launch_c3(const Conv3Dev& d, int nimg, hipStream_t stream) {
  constexpr int TH = 256 / TW;
  constexpr int HROWS = (UPS ? TH / 2 + 2 : TH + 2) * (UPS ? TW / 2 + 2 : TW + 2);
  constexpr int HB = (HROWS * 4 + 511) / 512 * 8192;
  constexpr int MAIN = 2 * HB + (TPB2 || SUBPIX ? 4 : 3) * BN * 64;   // halo double buffer + weight ring
  static_assert(MAIN <= 80 * 1024, "two workgroups per CU");
  int smem = MAIN;
  const int stage = 256 * (BN * (int)sizeof(T) + 16);   // LDS-staged epilogue tile
  Conv3Dev dd = d;
  dd.stage_ok = stage <= 80 * 1024;                      // keeps two workgroups per CU (fp32 x 128 channels stores directly)
  if (dd.stage_ok && smem < stage) smem = stage;
  // the statistics partials of the epilogue sit behind the staging tile (the main-loop buffers are dead by then)
  if (d.gn_part) { dd.gn_off = dd.stage_ok ? ((stage + 15) & ~15) : 0; smem = std::max(smem, dd.gn_off + 2 * (BN == 16 ? 8 : 4) * BN * 4); }
  if constexpr (GNA) { dd.coef_off = (smem + 15) & ~15; smem = dd.coef_off + 2 * (4 * Traits<T>::VEC) * 8; }
  // IVG_CONV_CAP=1: ONE workgroup per CU -- the request is padded past half of a CU's 160 KiB, so a second workgroup of this grid
  // never fits beside the first and half of the LDS, of the wave slots (8 of 16 per SIMD pair) and of the registers stay free for the
  // short kernels of ANOTHER batch in flight (decode attention, decode GEMMs planned under IVG_DECODE_LDS_KB): MFMA-bound waves
  // beside HBM- / latency-bound ones instead of a grid that holds every CU until it drains.
  if (sw().conv_cap) smem = std::max(smem, 82 * 1024);
  static DynLdsOnce once;
  auto kfn = conv3x3_kernel<T, BN, UPS, TW, GNA, TPB2, X3, SUBPIX>;
  if (hipError_t e = ensure_dyn_lds(once, (const void*)kfn, 160 * 1024); e != hipSuccess) return (int)e;
  const long blocks = (long)nimg * d.tiles_per_img * d.tiles_n * (SUBPIX ? 4 : 1);
  hipLaunchKernelGGL(kfn, dim3((unsigned)blocks), dim3(512), smem, stream, dd);
  return (int)hipGetLastError();
}

int conv3x3_gn_chunks_bound(int Hout, int Wout, int N) { return cdiv((long)Hout * Wout, 256) * cdiv(N, 64); }

bool conv3x3_enabled() { return sw().conv3x3 != 0; }

static std::atomic<long long> g_subpix_launches{0};
long long conv3x3_subpixel_launches() { return g_subpix_launches.load(std::memory_order_relaxed); }

// Returns -1 when the shape is not covered (caller falls back to the generic implicit GEMM).
int launch_conv3x3(const IgemmArgs& a, DType dtype, hipStream_t stream) {
  a.gn_chunks = 0;
  if (!conv3x3_enabled()) return -1;
  if (a.KH != 3 || a.KW != 3 || a.stride != 1 || a.pad != 1) return -1;
  if (a.nb0 * a.nb1 * a.nb2 != 1 || a.alpha != 1.0f || (a.flags & (IG_GLU | IG_BIAS_M))) return -1;
  const int ck = dtype == BF16 ? 32 : 16;
  if (a.Cin % ck != 0 || a.ldx != a.Cin || a.N < 1) return -1;
  const int Ho = a.Hout, Wo = a.Wout;
  if (a.ups ? (Ho != 2 * a.Hin || Wo != 2 * a.Win) : (Ho != a.Hin || Wo != a.Win)) return -1;
  // sub-pixel form of an upsampling convolution (pre-summed phase weights at hand, input tileable): tiles are laid over the INPUT
  const void* w_sub = a.W_x3 ? a.W_sub_x3 : a.W_sub;
  bool subpix = a.ups && w_sub && sw().subpixel && !a.gn_in_coef;
  if (subpix) {
    const int tw = a.Win >= 32 ? 32 : a.Win;
    if ((tw != 16 && tw != 32) || a.Win % tw != 0 || a.Hin % (256 / tw) != 0 || ((uintptr_t)w_sub & 15)) subpix = false;
  }
  const int Ht = subpix ? a.Hin : Ho, Wt = subpix ? a.Win : Wo;   // the grid the 256-pixel tiles cover
  const int TW = Wt >= 32 ? 32 : Wt;   // 16x16 or 8x32 tiles: halo <= 10 x 34 pixels = 24 KiB per buffer
  if (TW != 16 && TW != 32) return -1;
  // (16: the decoders' fused tail -- GroupNorm inside the staging, <= 16 output channels, bf16: one 16-channel fragment per wave)
  const int bn = a.N > 64 ? 128 : ((a.N <= 16 && a.gn_in_coef && dtype == BF16 && !a.W_x3 && !a.ups) ? 16 : 64);
  const int TH = 256 / TW;
  if (Wt % TW != 0 || Ht % TH != 0) return -1;
  if (((uintptr_t)a.X & 15) || ((uintptr_t)a.W & 15)) return -1;
  Conv3Dev d;
  d.X = a.X; d.W = a.W; d.Y = a.Y; d.R = a.R; d.bias = a.bias;
  d.H = a.Hin; d.Wd = a.Win; d.Cin = a.Cin; d.Ho = Ho; d.Wo = Wo;
  d.tiles_x = Wt / TW; d.tiles_per_img = d.tiles_x * (Ht / TH);
  d.N = a.N; d.ldw = a.ldw;
  d.tiles_n = cdiv(a.N, bn);
  d.c_img = a.c_img; d.c_pix = a.c_pix; d.c_ch = a.c_ch; d.c_grp = a.c_grp > 0 ? a.c_grp : 1; d.c_grp_stride = a.c_grp_stride;
  if (a.c_grp <= 1 && a.c_grp_stride == 0) d.c_grp_stride = a.c_img;
  d.flags = a.flags; d.stage_ok = 0;
  d.gn_part = nullptr; d.gn_groups = 0; d.gn_off = 0;
  d.in_coef = nullptr; d.coef_off = 0;
  const bool gna = a.gn_in_coef != nullptr;
  if (gna && a.ups) return -1;   // (the upsampling convs take un-normalised inputs)
  d.in_coef = (const f32x2*)a.gn_in_coef;
  if (a.gn_part && a.gn_groups > 0 && a.gn_groups <= 64 && a.N % a.gn_groups == 0 && (a.c_grp <= 1)) {
    d.gn_part = (double2*)a.gn_part; d.gn_groups = a.gn_groups;
    a.gn_chunks = d.tiles_per_img * d.tiles_n * (subpix ? 4 : 1);
  }
  if (subpix) {   // [phase][N][4 taps x Cin]: four 2x2 convolutions over the input (see the kernel's SUBPIX note)
    d.W = w_sub; d.ldw = 4 * a.Cin;
    g_subpix_launches.fetch_add(1, std::memory_order_relaxed);
#define IVG_C3S_TW(T, BNv, X) (TW == 16 ? launch_c3<T, BNv, false, 16, false, false, X, true>(d, a.Nimg, stream) : launch_c3<T, BNv, false, 32, false, false, X, true>(d, a.Nimg, stream))
#define IVG_C3S_BN(T, X) (bn == 128 ? IVG_C3S_TW(T, 128, X) : IVG_C3S_TW(T, 64, X))
    if (a.W_x3) { if (dtype != F32) return (int)hipErrorInvalidValue; return IVG_C3S_BN(float, true); }
    return dtype == BF16 ? IVG_C3S_BN(bf16_t, false) : IVG_C3S_BN(float, false);
#undef IVG_C3S_BN
#undef IVG_C3S_TW
  }
#define IVG_C3_TW(T, BNv, U, G, PR) (TW == 16 ? launch_c3<T, BNv, U, 16, G, PR>(d, a.Nimg, stream) : launch_c3<T, BNv, U, 32, G, PR>(d, a.Nimg, stream))
#define IVG_C3_BN(T, U, G, PR) (bn == 128 ? IVG_C3_TW(T, 128, U, G, PR) : IVG_C3_TW(T, 64, U, G, PR))
  if (a.W_x3) {   // split-bf16 arithmetic on fp32 tensors (the launcher swaps in the pre-split weights: same bytes per row)
    if (dtype != F32) return (int)hipErrorInvalidValue;
    d.W = a.W_x3;
#define IVG_C3X_TW(BNv, U, G) (TW == 16 ? launch_c3<float, BNv, U, 16, G, false, true>(d, a.Nimg, stream) : launch_c3<float, BNv, U, 32, G, false, true>(d, a.Nimg, stream))
#define IVG_C3X_BN(U, G) (bn == 128 ? IVG_C3X_TW(128, U, G) : IVG_C3X_TW(64, U, G))
    if (gna) return IVG_C3X_BN(false, true);
    return a.ups ? IVG_C3X_BN(true, false) : IVG_C3X_BN(false, false);
#undef IVG_C3X_BN
#undef IVG_C3X_TW
  }
  if (gna && bn == 16) return IVG_C3_TW(bf16_t, 16, false, true, false);
  if (gna) return dtype == BF16 ? IVG_C3_BN(bf16_t, false, true, false) : IVG_C3_BN(float, false, true, false);
  // (measured per shape, profiles/r02_conv3x3_tpb.txt: +2 ... +3.5 % on the plain convolutions, -0.8 % on the upsampling ones,
  // whose 32-pixel-row instance also spills registers in the two-step form: those keep one step per barrier)
  if (dtype == BF16 && !a.ups) return IVG_C3_BN(bf16_t, false, false, true);   // two steps per barrier
  if (a.ups) return dtype == BF16 ? IVG_C3_BN(bf16_t, true, false, false) : IVG_C3_BN(float, true, false, false);
  return dtype == BF16 ? IVG_C3_BN(bf16_t, false, false, false) : IVG_C3_BN(float, false, false, false);
#undef IVG_C3_BN
#undef IVG_C3_TW
}

}  // namespace ivg
