// RECORD, not a build target (round 6): this file was ivideogpt_amd/csrc/conv3x3w.hip in round 5 -- built into libivg behind IVG_CONV_WIDE
// (default off), bit-identical to conv3x3.hip, +13 % in isolation on the 64x64 layers and 1 % behind inside the decode stage.  It is the
// evidence behind DESIGN.md 6.1 (the convolutions sit at the socket's power limit: profiles/r05_conv_ab_*.txt, r05_conv_power.txt).  The
// round-5 review asked for measurement scaffolding (the IVG_CONV_WIDE_PROBE wrong-result probes) to leave the product library; the
// kernel left with it.  To rebuild: check out 683c1c8 (tests/test_gpu_conv_wide.py and tools/conv_ab.py live there).
// Persistent two-tile 3x3 convolution for gfx950 (bf16, 128 output channels per workgroup) -- an ALTERNATIVE to conv3x3.hip's kernel for
// the decoder trunks (vae.py:298-371, conditional_vae.py:186-212), selected by IVG_CONV_WIDE (default OFF).
//
// What it is.  conv3x3.hip: 256 pixels x 128 channels per workgroup, two workgroups per CU, both streaming the SAME weight tiles into
// private rings.  Here ONE workgroup per CU owns TWO spatial tiles (2 x 256 pixels, any two consecutive tiles of the launch) against ONE
// weight ring:
//   * weight bytes per FLOP halved, LDS fragment reads per MFMA down by a quarter (a wave reads 4 weight + 8 halo fragments for 32 MFMAs),
//   * 8 waves x 256 registers: a wave holds a 2 x (64 pixels x 64 channels) accumulator tile (128 registers), MFMAs written as in-place
//     inline asm (the register allocator otherwise rotates the accumulators through ~190 registers and spills addresses),
//   * PERSISTENT: the grid is one workgroup per CU, each walks its items (tile pair x N tile; the N tile is fixed per workgroup so the
//     weight stream simply wraps around) -- the first halo chunk, the first two weight tiles and (fused GroupNorm) the first chunk's
//     normalisation of item i + 1 are requested under the LAST chunk of item i exactly like any next chunk: the main loop never drains,
//   * the halo fragments of step s + 1 are read under the MFMAs of step s (PF), into the registers step s has just finished with,
//   * epilogue without workgroup barriers around the stores: every wave stages its own 64 x 64 sub-tile in a private LDS region (the
//     parity-1 halo buffers, dead by then) and stores whole 128-byte runs; one barrier per item.
// Everything else is conv3x3.hip's design: halo tile of a 32-channel chunk staged once by LDS-DMA and reused by nine taps, source-side
// XOR swizzle, three-slot weight ring requested two steps ahead, counted s_waitcnt vmcnt(n) + raw s_barrier, unrolled taps with
// immediate offsets, GroupNorm + SiLU of the input in place in LDS (GNA), output statistics from the epilogue (fixed order, no atomics).
// Same bf16 products in the same fp32 summation order: bit-identical to conv3x3.hip (tests/test_gpu_conv_wide.py).
//
// What round 5 measured with it (profiles/r05_conv_*.txt, r05_mfma_power.txt; DESIGN.md "what bounds the convolutions"):
//   * in isolation (tools/conv_ab.py, no residual / statistics): +13 % on the 64 x 64 fused-GroupNorm layers, +1 ... +3 % on the
//     upsampling ones, level or -3 % elsewhere; probes with the epilogue / the in-place normalisation switched off price them at
//     0.05-0.46 ms and 0.19-0.43 ms per launch -- both exposed here (one workgroup per CU: nothing overlaps them), both hidden by the
//     co-resident workgroup in conv3x3.hip;
//   * inside the decode stage (residual reads, statistics, kernel trace of tools/quick_bench.py): 38.4 ms of conv3x3 per decode
//     against 37.9 -- 1 % BEHIND, hence off by default;
//   * why no structure moves these kernels: they run at the socket's power limit.  tools/conv_power.py: the 256-pixel kernel on random
//     data draws 1,400 W (the cap) at 1,770 MHz for 1,473 TFLOP/s; the SAME launch on all-zero data 1,125 W at 2,383 MHz for 1,965; this
//     kernel 1,381 W at 2,089 MHz for 1,438 (fewer LDS / DMA bytes, more stalls: the firmware trades them for clock).
//     tools/ubench/mfma_power.hip: a pure stream of these MFMAs reaches 2,417 TFLOP/s on zeros (790 W), 1,950-2,000 on random operands
//     (1,305 W, sclk 2.03 GHz), 1,670-1,770 with the fragment reads of either kernel beside it.  MFMA-busy x clock, not MFMA-busy, is what
//     a launch delivers, and the product is set by the energy the launch spends.
#include <algorithm>
#include <type_traits>

#include "conv3x3_common.h"
#include "switches.h"

namespace ivg {

// acc += W fragment x A fragment, IN PLACE.  Written out because with 256 registers to spend the register allocator gives the
// builtin's result a fresh register quad (dst != srcC is legal), rotates the 128 accumulator registers through ~190 and then spills
// loop-carried addresses into scratch -- reloaded between the MFMAs, on the counter the LDS-DMA queue is waited on.  The accumulators
// are only ever written by these instructions, and the fragment registers only by LDS reads, so no software wait states are due inside
// the loop; the epilogue's first read of an accumulator sits behind a barrier and the item's address arithmetic.
__device__ __forceinline__ void mfma_bf16_acc(f32x4& c, const Chunk16& w, const Chunk16& x) {
  asm("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(c) : "v"(w), "v"(x));
}

template <int N> __device__ __forceinline__ void wait_dma_keep4() {   // as wait_dma_keep, but the newest FOUR LDS operations may stay in flight
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(4)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(4)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(3) lgkmcnt(4)" ::: "memory");
}

template <bool UPS, int TW, bool GNA, bool PF>
__global__ __launch_bounds__(512, 2) void conv3x3w_kernel(const Conv3Dev p) {
  using T = bf16_t;
  constexpr int BN = 128, CK = 32, WN = 64, FM = 4, FN = 4, NT = 2;
  constexpr int W_BYTES = BN * 64;
  constexpr int TH = 256 / TW;
  constexpr int HTW = UPS ? TW / 2 + 2 : TW + 2, HTH = UPS ? TH / 2 + 2 : TH + 2;
  constexpr int HROWS = HTH * HTW;
  constexpr int HI = (HROWS * 4 + 511) / 512;      // DMA pieces (512 lanes x 16 B) per halo chunk and tile
  constexpr int HB = HI * 8192;                    // one halo buffer
  constexpr int NP = NT * HI;                      // halo pieces per chunk: piece pc = (tile pc / HI, part pc % HI)
  static_assert(NP <= 6, "pieces are requested at taps 0 .. NP-1 and normalised at taps 3 .. NP+2");
  // ---- LDS map
  //   [0, STG)                 per-wave epilogue staging (8 x 64 rows x 144 B); its first 2 x HB bytes are the PARITY-1 halo buffers
  //   [STG, STG + 2 HB)        parity-0 halo buffers (tile 0, tile 1): hold the NEXT item's first chunk during an epilogue
  //   [.., + 3 x 8 KiB)        weight ring
  //   [.., + 8 KiB)            GroupNorm statistics partials of the epilogue: [tile][sum | sumsq][4 pixel waves][128 channels]
  //   [.., + 1 KiB)            coefficient rows of the fused input GroupNorm: [tile][parity][32] (scale, shift)
  constexpr int SPITCH = 64 * 2 + 16;              // staged row: a wave's 64 channels + 16 B (spreads the rows over banks)
  constexpr int STG = 8 * 64 * SPITCH;
  static_assert(2 * HB <= STG, "parity-1 halo buffers live under the staging region");
  constexpr int OFF_H0 = STG, OFF_W = STG + 2 * HB, OFF_GN = OFF_W + 3 * W_BYTES, OFF_COEF = OFF_GN + 8192;
  auto BR = [](int b) constexpr { return TW == 16 ? b : (b >> 1); };
  auto PXO = [](int b) constexpr { return TW == 16 ? 0 : (b & 1) * 16; };
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* wbuf0 = smem + OFF_W;
  auto hbuf = [&](int t, int par) -> unsigned char* { return smem + (par ? 0 : OFF_H0) + t * HB; };

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lr = lane & 15, lg = lane >> 4;
  const int wm = wave & 3, wn = wave >> 2;
  const int pyw = TW == 16 ? wm * 4 : wm * 2;

  // ---- work mapping.  Blocks b, b + 8, ... share an XCD / L2: within an XCD consecutive blocks take the N tiles of ONE tile
  // pair (their halos are the same lines), the N tile of a workgroup never changes (its weight stream wraps around).
  const int G = gridDim.x, per_xcd = G >> 3;
  const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
  const int tile_n = slot % p.tiles_n;
  const int L = G / p.tiles_n;                                    // tile pairs in flight on the chip
  int sp2 = xcd * (per_xcd / p.tiles_n) + slot / p.tiles_n;       // this workgroup's first pair; then sp2 += L
  if (sp2 >= p.sp_pairs) return;
  const int n_base = tile_n * BN;
  const int cin_b = p.Cin * (int)sizeof(T);

  struct TileGeo { int img, y0, x0, valid; };
  auto tile_geo = [&](int pair, int t) -> TileGeo {
    int s = 2 * pair + t;
    const int valid = s < p.sp_total;
    s = min(s, p.sp_total - 1);
    const int img = s / p.tiles_per_img, t_in = s - img * p.tiles_per_img;
    const int ty = t_in / p.tiles_x, tx = t_in - ty * p.tiles_x;
    return TileGeo{img, ty * TH, tx * TW, valid};
  };

  // ---- halo source state H: per-lane DMA sources of the item whose chunks are being PREFETCHED (the current item, or -- during
  // an item's last chunk -- the next one).  Out-of-image pixels (and the tail of the last piece) are never requested: their LDS
  // slots are zeroed whenever H changes.
  constexpr unsigned NOREQ = 0xffffffffu;   // hoff of a slot that is never requested (kept in the offset itself: no mask registers)
  unsigned hoff[NT][HI];
  unsigned h_any = 0;                       // bit pc: this WAVE requests anything in piece pc (the counted waits need the exact number)
  const unsigned char* Xb[NT];
  int himg[NT];
  auto setup_H = [&](int pair) {
    h_any = 0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const TileGeo g = tile_geo(pair, t);
      const int iy0 = UPS ? ((g.y0 - 1) >> 1) : (g.y0 - 1), ix0 = UPS ? ((g.x0 - 1) >> 1) : (g.x0 - 1);
      Xb[t] = (const unsigned char*)((const T*)p.X + (long)g.img * p.H * p.Wd * p.Cin);
      himg[t] = g.img;
#pragma unroll
      for (int it = 0; it < HI; ++it) {
        const int q = it * 512 + tid;
        const int row = q >> 2, slot4 = q & 3;
        const int hy = row / HTW, hx = row - hy * HTW;
        const int c = slot4 ^ halo_key<UPS>(hx);
        const int iy = iy0 + hy, ix = ix0 + hx;
        const bool ok = (row < HROWS) & (iy >= 0) & (iy < p.H) & (ix >= 0) & (ix < p.Wd);
        hoff[t][it] = ok ? (unsigned)__umul24((unsigned)(iy * p.Wd + ix), (unsigned)cin_b) + (unsigned)(c * 16) : NOREQ;
        h_any |= (unsigned)__builtin_amdgcn_readfirstlane(__builtin_amdgcn_ballot_w64(ok) != 0 ? 1 : 0) << (t * HI + it);
      }
    }
  };
  auto zero_fill = [&](int par) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int it = 0; it < HI; ++it)
        if (hoff[t][it] == NOREQ) *(Chunk16*)(hbuf(t, par) + (size_t)(it * 512 + tid) * 16) = Chunk16{0, 0, 0, 0};
  };
  const int wave_q = __builtin_amdgcn_readfirstlane(wave * 1024);   // a wave's 64 x 16 B of a 512-lane transfer
  auto issue_halo_piece = [&](int pc, int chunk, int par) {
    const int t = pc / HI, it = pc % HI;
    if (hoff[t][it] != NOREQ)
      glds16s(Xb[t] + (size_t)chunk * (CK * sizeof(T)), hoff[t][it], __builtin_amdgcn_readfirstlane(lds_addr(hbuf(t, par)) + it * 8192 + wave_q));
  };
  auto piece_any = [&](int pc) -> int { return (int)((h_any >> pc) & 1u); };

  // weight tile of one (tap, chunk): 128 rows x 4 slots of 16 B, one transfer per lane
  unsigned woff;
  {
    const int n = tid >> 2, slot4 = tid & 3;
    const int c = slot4 ^ swz_key(n);
    woff = (unsigned)((n_base + n) * p.ldw) * (unsigned)sizeof(T) + (unsigned)(c * 16);
  }
  auto issue_w = [&](int tap, int chunk, int ring) {
    glds16s((const unsigned char*)p.W + (size_t)(tap * p.Cin + chunk * CK) * sizeof(T), woff, __builtin_amdgcn_readfirstlane(lds_addr(wbuf0) + ring * W_BYTES + wave_q));
  };

  f32x2* s_coef = (f32x2*)(smem + OFF_COEF);   // [tile][parity][CK]
  // coefficient rows of the chunk being prefetched: wave t loads tile t's row (64 floats = one 4-byte-per-lane LDS-DMA)
  auto load_coef = [&](int chunk) -> int {
    if constexpr (GNA) {
      if (wave < NT) {
        const int img = wave == 0 ? himg[0] : himg[1];
        glds4s(p.in_coef + ((long)img * p.Cin + chunk * CK), (unsigned)lane * 4u, __builtin_amdgcn_readfirstlane(lds_addr(s_coef + (wave * 2 + (chunk & 1)) * CK)));
        return 1;
      }
    }
    return 0;
  };
  // normalise one piece of a staged halo chunk in place (same lane -> (pixel, slot) map as the DMA).  What the lane needs per piece
  // is its byte offset in the piece (tid * 16) and the channel slot its 16 bytes hold, c = slot ^ key(column): the three parts'
  // slots are packed into ONE register, and both are handed to every call as opaque copies -- left visible, the compiler hoists the
  // six pieces' pointers and predicates out of the chunk loop, where they live in scratch and are reloaded between the MFMAs
  // (a reload shares the counter of the LDS-DMA queue)
  unsigned cpack = 0;
  if constexpr (GNA) {
#pragma unroll
    for (int it = 0; it < HI; ++it) {
      const int row = (it * 512 + tid) >> 2;
      cpack |= (unsigned)((tid & 3) ^ halo_key<UPS>(row % HTW)) << (2 * it);
    }
  }
  auto transform_piece = [&](int chunk, int pc, int par) {
    if constexpr (GNA) {
      const int t = pc / HI, it = pc % HI;
      unsigned off = hoff[t][it], cp = cpack, t16 = (unsigned)tid * 16u;
      asm volatile("" : "+v"(off), "+v"(cp), "+v"(t16));
      if (off != NOREQ) {
        Chunk16* ptr = (Chunk16*)(hbuf(t, par) + it * 8192 + t16);
        const unsigned char* cf0 = (const unsigned char*)(s_coef + (t * 2 + (chunk & 1)) * CK) + ((cp >> (2 * it)) & 3u) * 64u;
        const f32x2* cf = (const f32x2*)cf0;
        const bf16x8 x = __builtin_bit_cast(bf16x8, *ptr);
        bf16x8 o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (bf16_t)silu_t<T>(fmaf((float)x[j], cf[j][0], cf[j][1]));
        *ptr = __builtin_bit_cast(Chunk16, o);
      }
    }
  };

  // ---- per-lane LDS fragment addresses (conv3x3.hip): one base per kw; fragment row, kh tap, tile and parity are immediates
  constexpr bool XHALF = UPS && TW == 32;
  int a_base[3];
#pragma unroll
  for (int kw = 0; kw < 3; ++kw) {
    int hx, rowbase;
    if constexpr (UPS) { hx = ((lr + kw - 1) >> 1) + 1; rowbase = (pyw >> 1) * HTW; }
    else { hx = lr + kw; rowbase = pyw * HTW; }
    a_base[kw] = (rowbase + hx) * 64 + ((lg ^ halo_key<UPS>(hx)) << 4);
  }
  auto a_imm = [&](int b, int kh) constexpr -> int {
    if constexpr (UPS) return ((((BR(b) + kh - 1) >> 1) + 1) * HTW) * 64;
    else return ((BR(b) + kh) * HTW + PXO(b)) * 64;
  };
  auto read_a = [&](int par, int t, int tap, int b) -> Chunk16 {
    const int kh = tap / 3, kw = tap - kh * 3;
    int base = a_base[kw];
    if constexpr (XHALF) {
      if (b & 1) { asm volatile("" : "+v"(base)); base = (base + 8 * 64) ^ 16; }
    }
    return *(const Chunk16*)(smem + base + ((par ? 0 : OFF_H0) + t * HB + a_imm(b, kh)));
  };
  const int w_base = swz(wn * WN + lr, lg);

  f32x4 acc[NT][FN][FM];
  auto clear_acc = [&]() {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int b = 0; b < FM; ++b) acc[t][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
  };

  const int nchunks = p.Cin / CK;   // even (launcher)
  // Every item of a launch costs the same, so workgroups that start together reach their epilogues together: 256 CUs x 128 KiB of
  // stores (+ the residual reads) hit the memory system in one burst while it idles during the main loops.  p.stagger > 0: the
  // workgroups start in eight phases, p.stagger x ~0.75 us apart, and keep that offset for the whole launch.
  for (int i = (slot & 7) * p.stagger; i > 0; --i) __builtin_amdgcn_s_sleep(24);
  // ---- prologue of the workgroup's FIRST item (every later item's first chunk arrives under its predecessor's last one)
  setup_H(sp2);
  zero_fill(0);
  zero_fill(1);
#pragma unroll
  for (int pc = 0; pc < NP; ++pc) issue_halo_piece(pc, 0, 0);
  issue_w(0, 0, 0);
  issue_w(1, 0, 1);
  (void)load_coef(0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  if constexpr (GNA) {
#pragma unroll
    for (int pc = 0; pc < NP; ++pc) transform_piece(0, pc, 0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  const bool early = wave < 4;   // the two waves of a SIMD (w, w + 4) issue their DMA at different points of a step

  // PF: the halo fragments of step s + 1 are read UNDER the MFMAs of step s, into the registers step s has just finished with (tile 0's
  // after its 16 MFMAs, tile 1's before the barrier): two waves per SIMD that meet at every barrier cannot hide each other's LDS
  // latency, and without this a step starts with twelve fragment reads in front of its first MFMA (only the four weight fragments are
  // left there: the weight tile of a step is only known to have landed at the barrier in front of it).
  Chunk16 xa[NT][FM];
  auto load_xa = [&](int par, int t, int tap) {
#pragma unroll
    for (int b = 0; b < FM; ++b) xa[t][b] = par ? read_a(1, t, tap, b) : read_a(0, t, tap, b);
  };
  // one 32-channel chunk = nine steps, consumed from the halo buffers of parity PAR (compile time: buffer addresses are immediates).
  // cnext: the chunk whose halo is staged under this one -- chunk + 1 of this item, or chunk 0 of the next item (H was switched);
  // item_end: this is the item's last chunk (its last step prefetches nothing: the epilogue comes first)
  auto run_chunk = [&](int chunk, int cnext, bool more, bool item_end, auto par) {
    constexpr int PAR = decltype(par)::value;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int tap2 = (tap + 2) % 9;
      const int chunk2 = tap < 7 ? chunk : cnext;
      const bool w_more = tap < 7 || more;
      int issued = 0;
      auto issue_dma = [&]() {
        if (w_more) { issue_w(tap2, chunk2, (tap + 2) % 3); issued += 1; }
        if (tap < NP) { if (more) { issue_halo_piece(tap, cnext, 1 - PAR); issued += piece_any(tap < NP ? tap : 0); } }
      };
      if (GNA || early) issue_dma();
      if constexpr (GNA) {
        if (more) {
          if (tap == 0) issued += load_coef(cnext);
          // piece pc was requested at tap pc and has landed by the end of tap pc + 1: normalised at tap pc + 2 (the last one at tap 7:
          // tap 8 already reads the next chunk's fragments)
          // (normalising in the second wave of every SIMD AFTER its MFMAs instead -- exponentials and reciprocals of one wave beside the
          // MFMAs of the other -- changes nothing: the launch is bound by the energy it spends, not by what overlaps what; header)
          if (tap >= 2 && tap - 2 < NP && !(p.probe & 2)) transform_piece(cnext, tap - 2, 1 - PAR);
        }
      }
      if (!GNA && !early) issue_dma();
      __builtin_amdgcn_sched_barrier(0);
      Chunk16 wv[FN];
#pragma unroll
      for (int a = 0; a < FN; ++a) wv[a] = *(const Chunk16*)(wbuf0 + w_base + ((tap % 3) * W_BYTES + a * 1024));
      if constexpr (!PF) {
        load_xa(PAR, 0, tap);
        load_xa(PAR, 1, tap);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int b = 0; b < FM; ++b)
#pragma unroll
          for (int a0 = 0; a0 < FN; ++a0) {
            const int a = (b & 1) ? FN - 1 - a0 : a0;   // snake order: one operand changes per MFMA (conv3x3.hip has the measurement)
            mfma_bf16_acc(acc[t][a][b], wv[a], xa[t][b]);
          }
        if constexpr (PF) {
          __builtin_amdgcn_sched_barrier(0);   // the reads below reuse xa[t]: not before its last MFMA has been issued
          if (tap < 8) load_xa(PAR, t, tap + 1);
          else if (!item_end) load_xa(1 - PAR, t, 0);
        }
      }
      // everything issued BEFORE this step has landed once at most `issued` transfers are still in flight.  LDS: this step's weight
      // fragments were consumed by its MFMAs and an in-place normalisation was issued before them (LDS operations of a wave complete in
      // order) -- only the four fragment reads of tile 1's NEXT step may stay in flight across the barrier (PF)
      if (PF && !(tap == 8 && item_end)) {
        if (issued == 0) wait_dma_keep4<0>();
        else if (issued == 1) wait_dma_keep4<1>();
        else if (issued == 2) wait_dma_keep4<2>();
        else wait_dma_keep4<3>();
      } else {
        if (issued == 0) wait_dma_keep<0>();
        else if (issued == 1) wait_dma_keep<1>();
        else if (issued == 2) wait_dma_keep<2>();
        else wait_dma_keep<3>();
      }
      __builtin_amdgcn_s_barrier();
    }
  };

  const int flags = p.flags;
  const bool gn = p.gn_part != nullptr;
  unsigned char* stg = smem + wave * (64 * SPITCH);       // this wave's private staging region
  float* ch_part = (float*)(smem + OFF_GN);               // [tile][2][4][BN]

  for (;;) {
    const int sp2n = sp2 + L;
    const bool has_next = sp2n < p.sp_pairs;
    clear_acc();
    if constexpr (PF) { load_xa(0, 0, 0); load_xa(0, 1, 0); }   // step 0's fragments (every later step's arrive under its predecessor)
    for (int chunk = 0; chunk < nchunks; chunk += 2) {
      run_chunk(chunk, chunk + 1, true, false, std::integral_constant<int, 0>{});
      const bool last = chunk + 2 >= nchunks;
      if (last && has_next) {
        // the halo source state moves on to the next item: its chunk 0 arrives in the parity-0 buffers (free since the barrier
        // that closed the chunk above) under this item's last chunk
        setup_H(sp2n);
        zero_fill(0);
      }
      run_chunk(chunk + 1, last ? 0 : chunk + 2, !last || has_next, last, std::integral_constant<int, 1>{});
    }

    // ---- epilogue of item sp2: bias, residual, SiLU, GroupNorm statistics, per-wave staged stores
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");   // (the last MFMAs' results, read by the VALU below: see mfma_bf16_acc)
    // (opaque per-item copies of the lane ids and of the bias pointer: nothing of the epilogue's address arithmetic or of its
    // invariant loads may be hoisted out of the persistent loop -- it would live in registers the main loop needs, or in scratch)
    int lr_e = lr, lg_e = lg, lane_e = lane;
    const float* bias_e = p.bias;
    asm volatile("" : "+v"(lr_e), "+v"(lg_e), "+v"(lane_e), "+s"(bias_e));
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const TileGeo g = tile_geo(sp2, t);
      if (!g.valid || (p.probe & 1)) continue;   // (odd number of spatial tiles: the last pair's second tile repeats the first and stores nothing)
      const long ibase = (long)g.img * p.c_grp_stride;
      float gs[FN][4], gq[FN][4];
#pragma unroll
      for (int a = 0; a < FN; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) { gs[a][r] = 0.f; gq[a][r] = 0.f; }
#pragma unroll
      for (int b = 0; b < FM; ++b) {
        const int pix = (g.y0 + pyw + BR(b)) * p.Wo + (g.x0 + PXO(b) + lr_e);
        const long obase = ibase + (long)pix * p.N + n_base + wn * WN + lg_e * 4;
#pragma unroll
        for (int a = 0; a < FN; ++a) {
          float v[4];
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = acc[t][a][b][r];
          if (flags & IG_BIAS_N) {
            const f32x4 bv = *(const f32x4*)(bias_e + n_base + wn * WN + a * 16 + lg_e * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += bv[r];
          }
          if (flags & IG_RESIDUAL) {
            const bf16x4 rv = *(const bf16x4*)((const T*)p.R + obase + a * 16);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += (float)rv[r];
          }
          if (flags & IG_SILU) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = silu_t<T>(v[r]);
          }
          const bf16x4 o4 = bf16x4{(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
          if (gn) {   // statistics of the STORED values
#pragma unroll
            for (int r = 0; r < 4; ++r) { const float f = (float)o4[r]; gs[a][r] += f; gq[a][r] = fmaf(f, f, gq[a][r]); }
          }
          *(bf16x4*)(stg + (b * 16 + lr_e) * SPITCH + (a * 16 + lg_e * 4) * 2) = o4;
        }
      }
      if (gn) {
        float* ch_s = ch_part + t * (2 * 4 * BN);
        float* ch_q = ch_s + 4 * BN;
#pragma unroll
        for (int a = 0; a < FN; ++a)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float s1 = row16_sum(gs[a][r]), s2 = row16_sum(gq[a][r]);
            if (lr_e == 0) { ch_s[wm * BN + wn * WN + a * 16 + lg_e * 4 + r] = s1; ch_q[wm * BN + wn * WN + a * 16 + lg_e * 4 + r] = s2; }
          }
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own staged rows (no other wave touches this region)
      T* Y = (T*)p.Y;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int b = j >> 1, lrr = (j & 1) * 8 + (lane_e >> 3), c16 = lane_e & 7;
        const Chunk16 val = *(const Chunk16*)(stg + (b * 16 + lrr) * SPITCH + c16 * 16);
        const int pix = (g.y0 + pyw + BR(b)) * p.Wo + (g.x0 + PXO(b) + lrr);
        *(Chunk16*)(Y + ibase + (long)pix * p.N + n_base + wn * WN + c16 * 8) = val;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // staged rows read before the next tile overwrites them
    }
    __builtin_amdgcn_s_barrier();   // every wave is done with its staging region (= the parity-1 halo buffers) and the partials are written
    if (gn) {
      const int t = tid >> 6, grp = tid & 63;
      if (t < NT && grp < p.gn_groups) {
        const TileGeo g = tile_geo(sp2, t);
        if (g.valid) {
          const float* ch_s = ch_part + t * (2 * 4 * BN);
          const float* ch_q = ch_s + 4 * BN;
          const int cpg = p.N / p.gn_groups;
          const int c0 = max(grp * cpg, n_base), c1 = min(min((grp + 1) * cpg, n_base + BN), p.N);
          double a1 = 0.0, a2 = 0.0;
          for (int c = c0; c < c1; ++c)
            for (int w = 0; w < 4; ++w) { a1 += (double)ch_s[w * BN + c - n_base]; a2 += (double)ch_q[w * BN + c - n_base]; }
          const int s = 2 * sp2 + t;
          const int t_in = s - g.img * p.tiles_per_img;
          const long chunk = (long)t_in * p.tiles_n + tile_n;
          p.gn_part[((long)g.img * p.tiles_per_img * p.tiles_n + chunk) * p.gn_groups + grp] = double2{a1, a2};
        }
      }
    }
    if (!has_next) break;
    zero_fill(1);   // H already describes the next item; its parity-1 slots were overwritten by the staging
    sp2 = sp2n;
  }
}

template <bool UPS, int TW, bool GNA, bool PF>
static int launch_c3w(const Conv3Dev& d, int grid, hipStream_t stream) {
  constexpr int TH = 256 / TW;
  constexpr int HROWS = (UPS ? TH / 2 + 2 : TH + 2) * (UPS ? TW / 2 + 2 : TW + 2);
  constexpr int HB = (HROWS * 4 + 511) / 512 * 8192;
  constexpr int SMEM = 8 * 64 * (64 * 2 + 16) + 2 * HB + 3 * 128 * 64 + 8192 + 1024;
  static_assert(SMEM <= 160 * 1024, "one workgroup per CU");
  static DynLdsOnce once;
  auto kfn = conv3x3w_kernel<UPS, TW, GNA, PF>;
  if (hipError_t e = ensure_dyn_lds(once, (const void*)kfn, 160 * 1024); e != hipSuccess) return (int)e;
  // (always more than half of a CU's LDS: a second workgroup of this grid never shares the CU)
  hipLaunchKernelGGL(kfn, dim3((unsigned)grid), dim3(512), std::max(SMEM, 82 * 1024), stream, d);
  return (int)hipGetLastError();
}

static std::atomic<long long> g_wide_launches{0};
long long conv3x3_wide_launches() { return g_wide_launches.load(std::memory_order_relaxed); }

// CUs of the current device (one workgroup per CU), cached per device
static int device_cus() {
  static std::atomic<int> cache[64];
  int dev = 0;
  (void)hipGetDevice(&dev);
  int v = cache[dev & 63].load(std::memory_order_relaxed);
  if (v > 0) return v;
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
  cache[dev & 63].store(n, std::memory_order_relaxed);
  return n;
}

int launch_conv3x3_wide(const Conv3Dev& d0, int nimg, bool ups, int TW, bool gna, hipStream_t stream) {
  if (!sw().conv_wide) return -1;
  if (d0.N % 128 != 0 || d0.c_ch != 1 || d0.c_pix != d0.N || d0.c_grp > 1) return -1;
  if (d0.flags & ~(IG_BIAS_N | IG_RESIDUAL | IG_SILU)) return -1;
  if (d0.Cin % 64 != 0) return -1;                         // an even number of 32-channel chunks (the halo parity of an item's first chunk)
  if (((uintptr_t)d0.Y & 15) || ((d0.flags & IG_RESIDUAL) && ((uintptr_t)d0.R & 7)) || ((d0.flags & IG_BIAS_N) && ((uintptr_t)d0.bias & 15))) return -1;
  if ((long)d0.c_grp_stride % 8 != 0) return -1;
  if (gna && ups) return -1;
  // IVG_CONV_WIDE=1: where it is the faster of the two IN ISOLATION (tools/conv_ab.py, profiles/r05_conv_ab_*.txt): ONE N tile (the
  // 64 x 64 / 256 x 256 levels, Cout = 128: +9 ... +16 %) and the upsampling convolutions up to four N tiles (+1 ... +3 %).  =2: wherever
  // it covers the shape (fused-GroupNorm layers with two and four N tiles are level or 1-3 % behind the 256-pixel kernel: every N tile
  // normalises the halo again, and nothing overlaps a workgroup's epilogue; 768-channel upsampling 6 % behind).
  if (sw().conv_wide != 2 && !(d0.tiles_n == 1 || (ups && d0.tiles_n <= 4))) return -1;
  Conv3Dev d = d0;
  d.sp_total = nimg * d.tiles_per_img;
  d.sp_pairs = (d.sp_total + 1) / 2;
  d.probe = sw().conv_wide_probe;
  // grid: one workgroup per CU, a multiple of 8 XCDs x N tiles; launches that would leave CUs without an item stay on the
  // two-workgroups-per-CU kernel
  const int unit = 8 * d.tiles_n;
  int grid = device_cus() / unit * unit;
  if (sw().conv_wide_grid > 0 && sw().conv_wide_grid % unit == 0) grid = sw().conv_wide_grid;   // development: another grid size
  if (grid <= 0 || (long)d.sp_pairs * d.tiles_n < grid) return -1;
  // start phases (see the kernel): worth +5 % where a workgroup walks many short items (64 x 64, 36 steps, 28 items), a loss on launches
  // of a few items per workgroup (the context decoder's 128 frames)
  const long items_per_wg = (long)d.sp_pairs * d.tiles_n / grid;
  d.stagger = sw().conv_wide_stagger >= 0 ? sw().conv_wide_stagger : (items_per_wg >= 12 ? 2 : 0);
  g_wide_launches.fetch_add(1, std::memory_order_relaxed);
#define IVG_C3W(U, W_, G_) (sw().conv_wide_pf ? launch_c3w<U, W_, G_, true>(d, grid, stream) : launch_c3w<U, W_, G_, false>(d, grid, stream))
  if (gna) return TW == 16 ? IVG_C3W(false, 16, true) : IVG_C3W(false, 32, true);
  if (ups) return TW == 16 ? IVG_C3W(true, 16, false) : IVG_C3W(true, 32, false);
  return TW == 16 ? IVG_C3W(false, 16, false) : IVG_C3W(false, 32, false);
#undef IVG_C3W
}

}  // namespace ivg
