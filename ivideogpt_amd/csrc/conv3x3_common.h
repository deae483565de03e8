// Shared pieces of the LDS-halo 3x3 convolution kernel (conv3x3.hip: 256-pixel tiles, two workgroups per CU).  gfx950 only.
// (Round 5's persistent two-tile variant -- bit-identical, 1 % behind inside the decode stage, the evidence behind DESIGN.md 6.1 --
// left the product library in round 6: tools/ubench/conv3x3w.hip keeps the source as a record.)
#pragma once
#include "igemm.h"

namespace ivg {

struct Conv3Dev {
  const void* X; const void* W; void* Y; const void* R; const float* bias;
  int H, Wd, Cin, Ho, Wo;            // input H x W (before upsampling), output Ho x Wo
  int tiles_x, tiles_per_img;        // spatial tiles (TH x TW output pixels each, TH * TW = 256)
  int N, ldw, tiles_n;
  long c_img, c_pix, c_ch, c_grp_stride;
  int c_grp, flags;
  int stage_ok;                      // the 256 x BN staging tile of the epilogue fits in the workgroup's LDS
  const f32x2* in_coef;              // GroupNorm + SiLU of the INPUT applied while it is staged (GNA): (scale, shift) [img][Cin]
  int coef_off;                      // byte offset of the two per-chunk coefficient rows in LDS
  double2* gn_part;                  // GroupNorm statistics of the output (null: off): [img][chunk = spatial tile x N tile][group]
  int gn_groups, gn_off;             // gn_off: byte offset of the per-channel partial sums in LDS (behind everything else)
};


// LDS-DMA (global_load_lds, 16 B per lane, lane-linear in LDS) with the address split the way the hardware takes it: a
// wave-uniform 64-bit base in scalar registers plus a 32-bit byte offset per lane.  __builtin_amdgcn_global_load_lds is always
// selected with a 64-bit per-lane address, which costs a 64-bit vector add per transfer and two registers per source; written
// out, the scalar unit advances the base (chunk / tap) and the per-lane offsets never change.  lds_wave_base: LDS byte address the wave's 64 x 16 B land at (M0).
__device__ __forceinline__ void glds16s(const void* sbase, unsigned voff, unsigned lds_wave_base) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_wave_base), "v"(voff), "s"(sbase) : "memory");
}
// (M0 is a reserved register: the compiler sets it itself right before each of its own uses -- this file leaves it none, every
// LDS-DMA goes through these two helpers)
__device__ __forceinline__ void glds4s(const void* sbase, unsigned voff, unsigned lds_wave_base) {   // 4 bytes per lane
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(lds_wave_base), "v"(voff), "s"(sbase) : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) {
  return (unsigned)(unsigned long)(__attribute__((address_space(3))) const void*)p;
}
// 64-byte LDS rows, 4 chunks of 16 B; the chunk a lane wants sits in slot (chunk ^ key).  Weight rows (16 consecutive rows per
// fragment) and the halo rows of the plain convolutions (16 consecutive pixels of one halo line) use key = (x >> 1) & 3:
// ds_read_b128 of 16 CONSECUTIVE rows is then bank-conflict free from ANY start row (found by enumeration over the lane groups
// of ds_read_b128, tools/lds_swizzle_check.py).  The halo key is a function of the pixel's COLUMN in the halo tile, so the
// fragments of the three kh taps and of a wave's four pixel rows differ by a constant byte offset.
__device__ __forceinline__ int swz_key(int row) { return (row >> 1) & 3; }
__device__ __forceinline__ int swz(int row, int chunk) { return row * 64 + ((chunk ^ swz_key(row)) << 4); }
// Halo rows of the nearest-x2 upsampling convs are read in PAIRS (two output pixels share an input pixel): the 16 lanes of a
// fragment touch 8 consecutive rows (tap kw = 1) or 9 (kw = 0, 2), and with the key above rows r and r + 8 of the 9-row case meet
// on the same banks (2-way conflict on two taps of three: SQ_LDS_BANK_CONFLICT 26 % in round 1).  No single key serves both
// access shapes, so the halo tiles of the upsampling instances use their own: the 2-bit reversal of (column >> 2),
// conflict-free for both pair alignments.
template <bool UPS>
__device__ __forceinline__ int halo_key(int hx) {
  if constexpr (UPS) { const int j = hx >> 2; return ((j & 1) << 1) | ((j >> 1) & 1); }
  else return swz_key(hx);
}

template <int N> __device__ __forceinline__ void wait_dma_keep() {   // all DMA but the newest N transfers, and every LDS read
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
}

}  // namespace ivg
