// Transformer-side kernels that are not GEMMs (SURVEY.md 2.4 K11, K15, K16 decode, K18, K19).
// All step-dependent scalars (cache length, index of the token being decided) live in a device-side
// StepState so the whole per-token kernel sequence has constant arguments and can be captured once
// into a hipGraph and replayed for every generated token.
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "ops.h"
#include <atomic>
#include "switches.h"

namespace ivg {

// ------------------------------------------------------------------------------------------------ embedding
template <typename T>
__global__ __launch_bounds__(256) void embed_kernel(const int64_t* __restrict__ ids, long id_stride, const T* __restrict__ E,
                                                    T* __restrict__ x, long x_bstride, int L, int H, int V) {
  constexpr int VEC = Traits<T>::VEC;
  const int row = blockIdx.x;  // b * L + l
  const int b = row / L, l = row - b * L;
  long id = ids[(long)b * id_stride + l];
  id = id < 0 ? 0 : (id >= V ? V - 1 : id);  // memory safety only: the mirror's callers pass ids < vocab_size
  const T* src = E + id * H;
  T* dst = x + (long)b * x_bstride + (long)l * H;
  for (int c = threadIdx.x * VEC; c < H; c += 256 * VEC) *(Chunk16*)(dst + c) = *(const Chunk16*)(src + c);
}

int launch_embed(const int64_t* ids, long id_stride, const void* E, void* x, DType dt, int B, int L, int H, int V, hipStream_t st,
                 long x_bstride) {
  if (B * L <= 0) return 0;
  if (x_bstride <= 0) x_bstride = (long)L * H;
  if (dt == BF16) hipLaunchKernelGGL(embed_kernel<bf16_t>, dim3(B * L), dim3(256), 0, st, ids, id_stride, (const bf16_t*)E, (bf16_t*)x, x_bstride, L, H, V);
  else hipLaunchKernelGGL(embed_kernel<float>, dim3(B * L), dim3(256), 0, st, ids, id_stride, (const float*)E, (float*)x, x_bstride, L, H, V);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ RoPE + KV append
// thread = (row m = b*L + l, head h, pair i < hd/2).  HF rotate_half convention:
//   out[i] = x[i]*cos - x[i+hd/2]*sin ;  out[i+hd/2] = x[i+hd/2]*cos + x[i]*sin
template <typename T>
__global__ __launch_bounds__(256) void rope_kv_kernel(T* __restrict__ qkv, T* __restrict__ kc, T* __restrict__ vc,
                                                      T* __restrict__ vt, int ldvt, const float* __restrict__ cosT,
                                                      const float* __restrict__ sinT, int B, int L, int heads, int hd, int Lmax,
                                                      const StepState* __restrict__ state, int pos0) {
  const int half = hd / 2;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long total = (long)B * L * heads * half;
  if (idx >= total) return;
  const int i = (int)(idx % half);
  long t = idx / half;
  const int h = (int)(t % heads); t /= heads;
  const int l = (int)(t % L);
  const int b = (int)(t / L);
  const int base_pos = state ? state->pos : pos0;
  const int pos = base_pos + l;
  const int H = heads * hd;
  T* row = qkv + ((long)b * L + l) * 3 * H;
  const float c = cosT[(long)pos * half + i], s = sinT[(long)pos * half + i];
  T* q = row + h * hd;
  const float q1 = to_f32(q[i]), q2 = to_f32(q[i + half]);
  q[i] = from_f32<T>(q1 * c - q2 * s);
  q[i + half] = from_f32<T>(q2 * c + q1 * s);
  const T* k = row + H + h * hd;
  const float k1 = to_f32(k[i]), k2 = to_f32(k[i + half]);
  T* kdst = kc + (((long)b * heads + h) * Lmax + pos) * hd;
  kdst[i] = from_f32<T>(k1 * c - k2 * s);
  kdst[i + half] = from_f32<T>(k2 * c + k1 * s);
  const T* v = row + 2 * H + h * hd;
  T* vdst = vc + (((long)b * heads + h) * Lmax + pos) * hd;
  const T v1 = v[i], v2 = v[i + half];
  vdst[i] = v1;
  vdst[i + half] = v2;
  if (vt) {
    T* tb = vt + ((long)b * heads + h) * hd * ldvt;
    tb[(long)i * ldvt + l] = v1;
    tb[(long)(i + half) * ldvt + l] = v2;
  }
}

// Vectorised form for head_dim 64 prompts (pos0 known on the host): a workgroup owns ROWS consecutive positions of one
// (trajectory, head); 16-byte loads / stores throughout, and V^T leaves through an LDS transpose as whole 32-byte runs
// along the position axis (the scalar kernel above writes V^T 2 bytes at a stride of ldvt).
template <typename T>
__global__ __launch_bounds__(256) void rope_kv64_kernel(T* __restrict__ qkv, T* __restrict__ kc, T* __restrict__ vc,
                                                        T* __restrict__ vt, int ldvt, const float* __restrict__ cosT,
                                                        const float* __restrict__ sinT, int L, int heads, int Lmax, int pos0) {
  constexpr int VEC = Traits<T>::VEC, HD = 64, HALF = 32;
  constexpr int CH = HALF / VEC;        // threads per position (each owns chunk c of both halves)
  constexpr int ROWS = 256 / CH;        // positions per workgroup: 64 (bf16) / 32 (fp32)
  constexpr int PITCH = HD * (int)sizeof(T) / 4 + 1;   // dwords per staged V row: odd -> transposed reads spread over banks
  __shared__ unsigned sv[ROWS * PITCH];
  const int tid = threadIdx.x;
  const int bh = blockIdx.y, b = bh / heads, h = bh - b * heads;
  const int H = heads * HD;
  const int l0 = blockIdx.x * ROWS;
  {
    const int lr_ = tid / CH, c = tid - lr_ * CH;
    const int l = l0 + lr_;
    unsigned* dst = sv + lr_ * PITCH;
    constexpr int DW = 16 / 4;           // dwords per 16-byte chunk
    if (l < L) {
      const int pos = pos0 + l;
      T* row = qkv + ((long)b * L + l) * 3 * H + h * HD;
      float cs[VEC], sn[VEC];
#pragma unroll
      for (int j = 0; j < VEC; j += 4) {
        const f32x4 cv = *(const f32x4*)(cosT + (long)pos * HALF + c * VEC + j), sv4 = *(const f32x4*)(sinT + (long)pos * HALF + c * VEC + j);
#pragma unroll
        for (int r = 0; r < 4; ++r) { cs[j + r] = cv[r]; sn[j + r] = sv4[r]; }
      }
      auto rope = [&](T* lo_p, T* hi_p, T* lo_dst, T* hi_dst) {
        const Chunk16 lo = *(const Chunk16*)lo_p, hi = *(const Chunk16*)hi_p;
        Chunk16 olo, ohi;
        if constexpr (sizeof(T) == 2) {
          const bf16x8 a = __builtin_bit_cast(bf16x8, lo), bb = __builtin_bit_cast(bf16x8, hi);
          bf16x8 oa, ob;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float x1 = (float)a[j], x2 = (float)bb[j];
            oa[j] = (bf16_t)(x1 * cs[j] - x2 * sn[j]);
            ob[j] = (bf16_t)(x2 * cs[j] + x1 * sn[j]);
          }
          olo = __builtin_bit_cast(Chunk16, oa); ohi = __builtin_bit_cast(Chunk16, ob);
        } else {
          const f32x4 a = __builtin_bit_cast(f32x4, lo), bb = __builtin_bit_cast(f32x4, hi);
          f32x4 oa, ob;
#pragma unroll
          for (int j = 0; j < 4; ++j) { oa[j] = a[j] * cs[j] - bb[j] * sn[j]; ob[j] = bb[j] * cs[j] + a[j] * sn[j]; }
          olo = __builtin_bit_cast(Chunk16, oa); ohi = __builtin_bit_cast(Chunk16, ob);
        }
        *(Chunk16*)lo_dst = olo; *(Chunk16*)hi_dst = ohi;
      };
      T* q = row + c * VEC;
      rope(q, q + HALF, q, q + HALF);
      T* k = row + H + c * VEC;
      T* kd = kc + ((long)bh * Lmax + pos) * HD + c * VEC;
      rope(k, k + HALF, kd, kd + HALF);
      const T* v = row + 2 * H + c * VEC;
      const Chunk16 vlo = *(const Chunk16*)v, vhi = *(const Chunk16*)(v + HALF);
      T* vd = vc + ((long)bh * Lmax + pos) * HD + c * VEC;
      *(Chunk16*)vd = vlo; *(Chunk16*)(vd + HALF) = vhi;
#pragma unroll
      for (int r = 0; r < DW; ++r) { dst[c * DW + r] = vlo[r]; dst[HALF * (int)sizeof(T) / 4 + c * DW + r] = vhi[r]; }
    } else {
#pragma unroll
      for (int r = 0; r < DW; ++r) { dst[c * DW + r] = 0u; dst[HALF * (int)sizeof(T) / 4 + c * DW + r] = 0u; }   // zero padding of V^T
    }
  }
  __syncthreads();
  if (vt) {   // thread (d, lc): ROWS / 4 consecutive positions of channel d
    constexpr int PER = ROWS / 4;
    const int d = tid >> 2, lc = tid & 3;
    const T* svT = (const T*)sv;
    T* out = vt + ((long)bh * HD + d) * ldvt + l0 + lc * PER;
    T vals[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) vals[j] = svT[(long)(lc * PER + j) * (PITCH * 4 / (int)sizeof(T)) + d];
#pragma unroll
    for (int j = 0; j < PER; j += VEC) {
      Chunk16 o;
      if constexpr (sizeof(T) == 2) { bf16x8 t8; for (int r = 0; r < 8; ++r) t8[r] = vals[j + r]; o = __builtin_bit_cast(Chunk16, t8); }
      else { f32x4 t4; for (int r = 0; r < 4; ++r) t4[r] = vals[j + r]; o = __builtin_bit_cast(Chunk16, t4); }
      *(Chunk16*)(out + j) = o;
    }
  }
}

int launch_rope_kv(void* qkv, void* kc, void* vc, void* vt, int ldvt, const float* cosT, const float* sinT, int B, int L,
                   int heads, int hd, int Lmax, const StepState* state, int pos0, DType dt, hipStream_t st) {
  const long total = (long)B * L * heads * (hd / 2);
  if (total <= 0) return 0;
  const int rows = dt == BF16 ? 64 : 32;
  if (hd == 64 && !state && (!vt || (ldvt % rows == 0 && ldvt >= cdiv(L, rows) * rows)) && ((uintptr_t)qkv & 15) == 0) {
    dim3 g2((unsigned)cdiv(L, rows), (unsigned)(B * heads));
    if (dt == BF16)
      hipLaunchKernelGGL(rope_kv64_kernel<bf16_t>, g2, dim3(256), 0, st, (bf16_t*)qkv, (bf16_t*)kc, (bf16_t*)vc, (bf16_t*)vt, ldvt, cosT, sinT,
                         L, heads, Lmax, pos0);
    else
      hipLaunchKernelGGL(rope_kv64_kernel<float>, g2, dim3(256), 0, st, (float*)qkv, (float*)kc, (float*)vc, (float*)vt, ldvt, cosT, sinT, L,
                         heads, Lmax, pos0);
    return (int)hipGetLastError();
  }
  dim3 g(cdiv(total, 256));
  if (dt == BF16)
    hipLaunchKernelGGL(rope_kv_kernel<bf16_t>, g, dim3(256), 0, st, (bf16_t*)qkv, (bf16_t*)kc, (bf16_t*)vc, (bf16_t*)vt, ldvt, cosT,
                       sinT, B, L, heads, hd, Lmax, state, pos0);
  else
    hipLaunchKernelGGL(rope_kv_kernel<float>, g, dim3(256), 0, st, (float*)qkv, (float*)kc, (float*)vc, (float*)vt, ldvt, cosT, sinT,
                       B, L, heads, hd, Lmax, state, pos0);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ V^T tile of the one-pass attention kernels
// The P.V MFMA of flash_prefill_kernel / xattn_kernel enumerates the keys of a 32-key step as (4*lg + r, 16 + 4*lg + r) -- the order a
// lane's own S registers have -- so its V^T fragment is two groups of four consecutive keys, 16 apart.  Read as two 8-byte halves
// (rounds 2-4) the compiler merged them into ds_read2_b64: two accesses of 4 x 16 CONTIGUOUS lanes with banks mod 32, half the rate
// of ds_read_b128, and 2-way conflicted for 16 consecutive rows of the 144-byte pitch (the 32-38 % bank-conflict cycles of the PMC
// tables).  Now the keys of every 32-key block are PERMUTED when the tile is staged -- position lg*8 + half*4 + r holds key
// half*16 + lg*4 + r -- so a fragment is ONE conflict-free ds_read_b128.  Rows of 64 keys carry 32 bytes of pad; rows of 32 keys
// (64 bytes: the single-head 512- / 768-channel instances) use the XOR key of conv3x3's 64-byte rows instead, without pad (the
// 512-channel instance keeps two workgroups per CU).  The staging stores become two ds_write_b64 per 16-byte chunk, odd rows of the
// padded form in the other order (16 contiguous lanes = two rows then touch 32 distinct banks).  Model: tools/lds_vt_layout_check.py
// (function of the permutation + bank cycles of both access shapes).
template <int KT>
struct VtTile {
  static constexpr bool SWZ = KT == 32;
  static constexpr int PITCH = SWZ ? KT : KT + 16;   // elements
  static __device__ __forceinline__ int quad_off(int row, int quad) {   // element offset of 16-byte quad `quad` of row `row`
    if constexpr (SWZ) quad ^= (row >> 1) & 3;
    return row * PITCH + (quad << 3);
  }
  // source chunk cc of a row = its keys 8cc .. 8cc+7
  static __device__ __forceinline__ void store(bf16_t* sV, int row, int cc, Chunk16 c) {
    typedef uint32_t U2 __attribute__((ext_vector_type(2)));
    const int qa = (cc >> 2) * 4 + 2 * (cc & 1), e = ((cc >> 1) & 1) * 4;
    bf16_t* a = sV + quad_off(row, qa) + e;
    bf16_t* b = sV + quad_off(row, qa + 1) + e;
    const U2 lo = {c.x, c.y}, hi = {c.z, c.w};
    const bool odd = !SWZ && (row & 1);
    bf16_t* p0 = odd ? b : a;
    bf16_t* p1 = odd ? a : b;
    *(U2*)p0 = odd ? hi : lo;
    *(U2*)p1 = odd ? lo : hi;
  }
  // fragment of K-step pr for lane (lr, lg) of the 16-row block d: frag(sV, d * 16 + lr, pr * 4 + lg)
  static __device__ __forceinline__ bf16x8 frag(const bf16_t* sV, int row, int quad) { return *(const bf16x8*)(sV + quad_off(row, quad)); }
};

// ------------------------------------------------------------------------------------------------ prefill attention
// Causal self-attention of a prompt (positions 0 .. L-1) in one pass, bf16 / head_dim 64: replaces the score GEMM, the row
// softmax and the P.V GEMM of the prefill (and their fp32 score matrix in HBM: 910 MB per layer at config 2).
// Workgroup = 64 query rows of one (trajectory, head); wave = 16 query rows against every key tile up to the diagonal:
//   S = K Q^T      a = K rows (cache, roped), b = Q rows (roped in place by rope_kv)  -> lane: S[key = 4*lg + r][q = lr]
//   online softmax in fp32 (running max / sum per query row; a row lives in the 4 lanes lr, lr+16, lr+32, lr+48)
//   O += V^T P     a = rows of V^T (vt, [d][key]), b = P cast to bf16: the lane's own S registers ARE its P fragment when
//                  the K dimension of this MFMA enumerates the keys as (4*lg + r, 16 + 4*lg + r) -- V^T is read in the
//                  same order, so no cross-lane movement is needed          -> lane: O[d = 4*lg + r][q = lr]
// Key rows beyond the diagonal are masked by select (never by arithmetic: cache rows past L may hold anything).
__global__ __launch_bounds__(256) void flash_prefill_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ kc,
                                                            const bf16_t* __restrict__ vt, bf16_t* __restrict__ out, int L, int Lp,
                                                            int heads, int Lmax, float scale) {
  constexpr int HD = 64;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lg = lane >> 4;
  const int qt = gridDim.x - 1 - blockIdx.x;   // long rows (many key tiles) are scheduled first
  const int bh = blockIdx.y, b = bh / heads, h = bh - b * heads;
  const int H = heads * HD;
  const int q = qt * 64 + wave * 16 + lr;
  const int qc = q < L ? q : L - 1;
  const bf16_t* qrow = qkv + ((long)b * L + qc) * 3 * H + h * HD;
  const bf16x8 qf0 = *(const bf16x8*)(qrow + lg * 8), qf1 = *(const bf16x8*)(qrow + 32 + lg * 8);
  const bf16_t* kb = kc + (long)bh * Lmax * HD;
  const bf16_t* vb = vt + (long)bh * HD * Lp;
  f32x4 o[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  float m = -INFINITY, lsum = 0.f;
  // The four waves of the workgroup need the same 64-key tile of K and V^T: it is staged once in LDS instead of being pulled
  // from L2 by every wave, and the next tile travels from global memory into registers while the current one is consumed.
  // Row pitches per access shape (bank model of MI355X_MICROARCH.md, tools/lds_swizzle_check.py): the K rows are read back as
  // 16-byte MFMA fragments (ds_read_b128: 4 groups of 16 lanes) -- a pad of 32 bytes is conflict-free, the 16 bytes of round 2
  // cost every read a second LDS cycle (PMC: 41 % bank-conflict cycles); the V^T rows are staged key-permuted (VtTile above) and
  // read back the same way.
  using VT = VtTile<64>;
  constexpr int PITCHK = HD + 16;     // K rows
  __shared__ __attribute__((aligned(16))) bf16_t sK[64 * PITCHK];
  __shared__ __attribute__((aligned(16))) bf16_t sV[64 * VT::PITCH];
  const int srow0 = tid >> 3, scol = (tid & 7) * 8;   // this thread stages chunks (srow0, scol) and (srow0 + 32, scol)
  Chunk16 pk[2], pv[2];
  auto fetch = [&](int kt) {
    const int k0 = kt * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = srow0 + i * 32;
      const int kr = k0 + row;
      pk[i] = *(const Chunk16*)(kb + (long)(kr < Lmax ? kr : Lmax - 1) * HD + scol);   // rows past the diagonal are masked below
      pv[i] = *(const Chunk16*)(vb + (long)row * Lp + k0 + scol);
    }
  };
  fetch(0);
  for (int kt = 0; kt <= qt; ++kt) {
    const int k0 = kt * 64;
    __syncthreads();   // every wave is done with the previous tile
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *(Chunk16*)(sK + (srow0 + i * 32) * PITCHK + scol) = pk[i];
      VT::store(sV, srow0 + i * 32, tid & 7, pv[i]);
    }
    __syncthreads();
    if (kt < qt) fetch(kt + 1);
    bf16x8 kf[4][2];
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      const bf16_t* krow = sK + (sub * 16 + lr) * PITCHK;
      kf[sub][0] = *(const bf16x8*)(krow + lg * 8);
      kf[sub][1] = *(const bf16x8*)(krow + 32 + lg * 8);
    }
    bf16x8 vf[4][2];
#pragma unroll
    for (int d = 0; d < 4; ++d)
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) vf[d][pr] = VT::frag(sV, d * 16 + lr, pr * 4 + lg);
    f32x4 sc[4];
    float mt = -INFINITY;
#pragma unroll
    for (int sub = 0; sub < 4; ++sub) {
      sc[sub] = f32x4{0.f, 0.f, 0.f, 0.f};
      sc[sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[sub][0], qf0, sc[sub], 0, 0, 0);
      sc[sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf[sub][1], qf1, sc[sub], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = k0 + sub * 16 + lg * 4 + r;
        sc[sub][r] = key <= q ? sc[sub][r] * scale : -INFINITY;
        mt = fmaxf(mt, sc[sub][r]);
      }
    }
    mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float mn = fmaxf(m, mt);            // finite: key 0 <= q is in the first tile of every row
    const float alpha = expf(m - mn);         // first tile: exp(-inf) = 0
    m = mn;
    float ps = 0.f;
    bf16x8 pf[2];
#pragma unroll
    for (int sub = 0; sub < 4; ++sub)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pv = expf(sc[sub][r] - mn);
        ps += pv;
        pf[sub >> 1][(sub & 1) * 4 + r] = (bf16_t)pv;
      }
    lsum = lsum * alpha + ps;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
#pragma unroll
      for (int r = 0; r < 4; ++r) o[d][r] *= alpha;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf[d][pr], pf[pr], o[d], 0, 0, 0);
    }
  }
  lsum += __shfl_xor(lsum, 16, 64);
  lsum += __shfl_xor(lsum, 32, 64);
  if (q < L) {
    const float inv = 1.0f / lsum;
    bf16_t* orow = out + ((long)b * L + q) * H + h * HD + lg * 4;
#pragma unroll
    for (int d = 0; d < 4; ++d)
      *(bf16x4*)(orow + d * 16) = bf16x4{(bf16_t)(o[d][0] * inv), (bf16_t)(o[d][1] * inv), (bf16_t)(o[d][2] * inv), (bf16_t)(o[d][3] * inv)};
  }
}

// qkv: [B*L][3H] with q already roped in place; kc: roped keys [B][heads][Lmax][64]; vt: V^T [B][heads][64][Lp] (finite
// beyond L); out [B*L][H].  bf16, head_dim 64 only (returns -1 otherwise: the caller keeps the three-kernel path).
int launch_flash_prefill(const void* qkv, const void* kc, const void* vt, void* out, int B, int L, int Lp, int heads, int hd, int Lmax,
                         DType dt, hipStream_t st) {
  if (dt != BF16 || hd != 64 || L <= 0 || Lp % 64 != 0 || Lp < L) return -1;
  dim3 grid((unsigned)cdiv(L, 64), (unsigned)(B * heads));
  hipLaunchKernelGGL(flash_prefill_kernel, grid, dim3(256), 0, st, (const bf16_t*)qkv, (const bf16_t*)kc, (const bf16_t*)vt, (bf16_t*)out, L,
                     Lp, heads, Lmax, 1.0f / sqrtf((float)hd));
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ tokenizer cross-attention
// CrossAttentionBlock's multi-head attention (ivideogpt/vq_model/conditional_vae.py:38-55) in one pass, bf16:
// every future frame's P = side^2 query tokens attend to the kv = ctx * P projected context tokens of ITS trajectory.  Replaces the
// batched score GEMM, the row softmax over an fp32 score matrix in HBM (1.9 GB at config 2) and the batched P.V GEMM.
// Same dataflow as flash_prefill_kernel (S = K Q^T, online softmax in fp32, O += V^T P with the lane's own S registers as its P
// fragment), without a mask, for head dims 128 / 192 (C = 512 / 768, four heads):
//   q   [M][P][C]   frame m = b * F + f, head h reads channels [h * HD, (h + 1) * HD)
//   Kp  [B][kv][C]  projected keys of trajectory b (shared by its F frames)
//   VpT [B][C][kv]  projected values, transposed
//   out [M][P][C]
template <int HD, int KT, bool PRE = true>   // KT: keys per staged tile (64; 32 for the single-head self-attention of 512 / 768 channels: LDS);
                                            // PRE: the next tile travels into registers under the current one (off at 768 channels: registers)
__global__ __launch_bounds__(256) void xattn_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ Kp, const bf16_t* __restrict__ VpT,
                                                    bf16_t* __restrict__ out, int P, int kv, int C, int F, int nh, float scale) {
  constexpr int KQ = HD / 32;   // MFMA K-steps of the score product
  constexpr int DO = HD / 16;   // 16-row tiles of V^T / output channels
  constexpr int SUB = KT / 16;  // 16-key sub-tiles of the score tile
  constexpr int NPF = KT / 32;  // 32-key P fragments of the P.V product
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, lr = lane & 15, lg = lane >> 4;
  const int qt = blockIdx.x;
  const int mh = blockIdx.y, m = mh / nh, h = mh - m * nh, b = m / F;
  const int qi = qt * 64 + wave * 16 + lr;               // P is a multiple of 64 (16 x 16 or 32 x 32 tokens)
  const bf16_t* qrow = q + ((long)m * P + qi) * C + h * HD;
  bf16x8 qf[KQ];
#pragma unroll
  for (int k = 0; k < KQ; ++k) qf[k] = *(const bf16x8*)(qrow + k * 32 + lg * 8);
  const bf16_t* kb = Kp + (long)b * kv * C + h * HD;     // row stride C
  const bf16_t* vb = VpT + ((long)b * C + h * HD) * kv;  // row stride kv
  f32x4 o[DO];
#pragma unroll
  for (int d = 0; d < DO; ++d) o[d] = f32x4{0.f, 0.f, 0.f, 0.f};
  float mx = -INFINITY, lsum = 0.f;
  constexpr int PK = HD + 16;   // staged K row (elements), read back as 16-byte fragments: 32 bytes of pad are conflict-free for
                                // ds_read_b128's lane groups (16 bytes cost every read a second cycle: 42 % conflict cycles in round 2)
  using VT = VtTile<KT>;        // staged V^T rows: key-permuted, one ds_read_b128 per fragment (above)
  __shared__ __attribute__((aligned(16))) bf16_t sK[KT * PK];
  __shared__ __attribute__((aligned(16))) bf16_t sV[HD * VT::PITCH];
  constexpr int KC = KT * HD / 8 / 256;   // 16-byte chunks of the K tile per thread
  constexpr int VC = HD * KT / 8 / 256;   // ... of the V^T tile
  constexpr int VCH = KT / 8;             // chunks per V^T row
  static_assert(KC >= 1 && VC >= 1, "tile too small for 256 threads");
  Chunk16 pk[KC], pvv[VC];
  auto fetch = [&](int kt) {
    const int k0 = kt * KT;
#pragma unroll
    for (int i = 0; i < KC; ++i) {
      const int c = i * 256 + tid, row = c / (HD / 8), col = (c % (HD / 8)) * 8;
      pk[i] = *(const Chunk16*)(kb + (long)(k0 + row) * C + col);
    }
#pragma unroll
    for (int i = 0; i < VC; ++i) {
      const int c = i * 256 + tid, row = c / VCH, col = (c % VCH) * 8;
      pvv[i] = *(const Chunk16*)(vb + (long)row * kv + k0 + col);
    }
  };
  const int ntiles = kv / KT;
  if constexpr (PRE) fetch(0);
  for (int kt = 0; kt < ntiles; ++kt) {
    __syncthreads();   // every wave is done with the previous tile
    if constexpr (!PRE) fetch(kt);
#pragma unroll
    for (int i = 0; i < KC; ++i) { const int c = i * 256 + tid; *(Chunk16*)(sK + (c / (HD / 8)) * PK + (c % (HD / 8)) * 8) = pk[i]; }
#pragma unroll
    for (int i = 0; i < VC; ++i) { const int c = i * 256 + tid; VT::store(sV, c / VCH, c % VCH, pvv[i]); }
    __syncthreads();
    if constexpr (PRE) { if (kt + 1 < ntiles) fetch(kt + 1); }
    f32x4 sc[SUB];
    float mt = -INFINITY;
#pragma unroll
    for (int sub = 0; sub < SUB; ++sub) {
      sc[sub] = f32x4{0.f, 0.f, 0.f, 0.f};
      const bf16_t* krow = sK + (sub * 16 + lr) * PK + lg * 8;
#pragma unroll
      for (int k = 0; k < KQ; ++k) sc[sub] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(*(const bf16x8*)(krow + k * 32), qf[k], sc[sub], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; ++r) { sc[sub][r] *= scale; mt = fmaxf(mt, sc[sub][r]); }
    }
    mt = fmaxf(mt, __shfl_xor(mt, 16, 64));
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float mn = fmaxf(mx, mt);
    const float alpha = expf(mx - mn);        // first tile: exp(-inf) = 0
    mx = mn;
    float ps = 0.f;
    bf16x8 pf[NPF];
#pragma unroll
    for (int sub = 0; sub < SUB; ++sub)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float pe = expf(sc[sub][r] - mn);
        ps += pe;
        pf[sub >> 1][(sub & 1) * 4 + r] = (bf16_t)pe;
      }
    lsum = lsum * alpha + ps;
#pragma unroll
    for (int d = 0; d < DO; ++d) {
#pragma unroll
      for (int r = 0; r < 4; ++r) o[d][r] *= alpha;
#pragma unroll
      for (int pr = 0; pr < NPF; ++pr)
        o[d] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(VT::frag(sV, d * 16 + lr, pr * 4 + lg), pf[pr], o[d], 0, 0, 0);
    }
  }
  lsum += __shfl_xor(lsum, 16, 64);
  lsum += __shfl_xor(lsum, 32, 64);
  const float inv = 1.0f / lsum;
  bf16_t* orow = out + ((long)m * P + qi) * C + h * HD + lg * 4;
#pragma unroll
  for (int d = 0; d < DO; ++d)
    *(bf16x4*)(orow + d * 16) = bf16x4{(bf16_t)(o[d][0] * inv), (bf16_t)(o[d][1] * inv), (bf16_t)(o[d][2] * inv), (bf16_t)(o[d][3] * inv)};
}

// one predicate for the planner and the launcher: which (dtype, shape) the one-pass kernel covers
bool xattn_covers(int P, int kv, int C, int nh, DType dt) {
  if (!sw().flash_xatt || dt != BF16 || nh <= 0 || C % nh != 0 || P % 64 != 0 || kv % 64 != 0 || (C & 7)) return false;
  const int hd = C / nh;
  return hd == 32 || hd == 64 || hd == 128 || hd == 192 || hd == 512 || hd == 768;
}

// -1: shape / dtype not covered (the caller keeps the score GEMM + softmax + P.V GEMM path)
int launch_xattn(const void* q, const void* Kp, const void* VpT, void* out, int M, int F, int P, int kv, int C, int nh, DType dt, hipStream_t st) {
  if (!xattn_covers(P, kv, C, nh, dt) || M <= 0 || F <= 0 || M % F != 0) return -1;
  if (((uintptr_t)q & 15) || ((uintptr_t)Kp & 15) || ((uintptr_t)VpT & 15) || ((uintptr_t)out & 7)) return -1;
  const int hd = C / nh;
  dim3 grid((unsigned)(P / 64), (unsigned)(M * nh));
  const float scale = 1.0f / sqrtf((float)hd);
#define IVG_XATTN(HDv, KTv, PREv) hipLaunchKernelGGL((xattn_kernel<HDv, KTv, PREv>), grid, dim3(256), 0, st, (const bf16_t*)q, (const bf16_t*)Kp, (const bf16_t*)VpT, (bf16_t*)out, P, kv, C, F, nh, scale)
  if (hd == 128) IVG_XATTN(128, 64, true);
  else if (hd == 192) IVG_XATTN(192, 64, true);
  else if (hd == 64) IVG_XATTN(64, 64, true);
  else if (hd == 32) IVG_XATTN(32, 64, true);
  else if (hd == 512) IVG_XATTN(512, 32, true);   // diffusers Attention of the conditional mid blocks: one head of 512 / 768 channels
  else if (hd == 768) IVG_XATTN(768, 32, false);
  else return -1;
#undef IVG_XATTN
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ decode attention
// One workgroup per (b, head): RoPE of the new q/k, KV-cache append and single-token attention in one launch.
// HBM-bound stream of the K and V rows of the cache (coalesced: LPK lanes share one row, 16 bytes each, four
// independent row loads in flight per lane).  Two passes over LDS-held scores -> deterministic reduction order.
// q of one lane for the key dot products: fp32 values (fp32 caches) or packed bf16 pairs (bf16 caches)
template <typename T> struct QPack;
template <> struct QPack<float> {
  __device__ __forceinline__ void set(const float*) {}
  __device__ __forceinline__ float dot(const float* qf, Chunk16 raw) const {
    const f32x4 kk = __builtin_bit_cast(f32x4, raw);
    float d = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) d = fmaf(qf[j], kk[j], d);
    return d;
  }
};
template <> struct QPack<bf16_t> {
  bf16x2 q[4];
  __device__ __forceinline__ void set(const float* qf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) q[j] = bf16x2{(bf16_t)qf[2 * j], (bf16_t)qf[2 * j + 1]};
  }
  __device__ __forceinline__ float dot(const float*, Chunk16 raw) const {
    const bf16x8 kk = __builtin_bit_cast(bf16x8, raw);
    float d = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) d = __builtin_amdgcn_fdot2_f32_bf16(q[j], bf16x2{kk[2 * j], kk[2 * j + 1]}, d, false);
    return d;
  }
};
template <typename T> __device__ __forceinline__ float dot_chunk(const float* qf, Chunk16 raw) {
  constexpr int VEC = Traits<T>::VEC;
  float d = 0.f;
  if constexpr (sizeof(T) == 2) {
    const bf16x8 kk = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
    for (int j = 0; j < VEC; ++j) d = fmaf(qf[j], (float)kk[j], d);
  } else {
    const f32x4 kk = __builtin_bit_cast(f32x4, raw);
#pragma unroll
    for (int j = 0; j < VEC; ++j) d = fmaf(qf[j], kk[j], d);
  }
  return d;
}
template <typename T> __device__ __forceinline__ void axpy_chunk(float* of, float pw, Chunk16 raw) {
  constexpr int VEC = Traits<T>::VEC;
  if constexpr (sizeof(T) == 2) {
    const bf16x8 vv = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
    for (int j = 0; j < VEC; ++j) of[j] = fmaf(pw, (float)vv[j], of[j]);
  } else {
    const f32x4 vv = __builtin_bit_cast(f32x4, raw);
#pragma unroll
    for (int j = 0; j < VEC; ++j) of[j] = fmaf(pw, vv[j], of[j]);
  }
}

// Sum over the lpk (8 or 16) adjacent lanes that share one key row, every lane receiving the total.  DPP lane permutes
// (VALU latency) instead of __shfl_xor (ds_bpermute: an LDS round trip per step, three or four dependent ones per key row
// -- that latency, not HBM, was pacing the key pass).  Same pairing tree as the xor butterfly: bitwise the same sums.
template <int CTRL> __device__ __forceinline__ float dpp_add(float v) {
  return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float group_sum(float d, int lpk) {
  if (lpk >= 2) d = dpp_add<0xB1>(d);    // quad_perm [1,0,3,2]: lane ^ 1
  if (lpk >= 4) d = dpp_add<0x4E>(d);    // quad_perm [2,3,0,1]: lane ^ 2
  if (lpk >= 8) d = dpp_add<0x141>(d);   // row_half_mirror: lane i <-> 7 - i of its 8-lane half (the other quad's total)
  if (lpk >= 16) d = dpp_add<0x140>(d);  // row_mirror: lane i <-> 15 - i (the other half's total)
  for (int o = 16; o < lpk; o <<= 1) d += __shfl_xor(d, o, 64);   // head_dim > 128 (bf16): beyond a DPP row
  return d;
}

// NT: non-temporal loads of the cache rows (streamed once per step: keep them from evicting the weights)
// HD: head dimension as a compile-time value (64 for the released transformers; 0 = generic, read from the argument) -- with it
// the lane-group reductions, the group counts and the row strides are constants instead of chains of scalar branches per chunk
// (Two value blocks in flight across the softmax statistics were built and measured slower in round 3 -- 64.3 vs 61.2 ms of
// attention per step, profiles/r03_attn_pre2_ab.txt: one block per workgroup already keeps 25 MB in flight chip-wide -- removed.)
// SHARED (round 6, shared-context rollouts: ivg_generate_shared): the G rows of a group were given ONE prompt -- predict.py's
// repeat_times samples of a clip, train_gpt.generate_multiple_times, VP2's candidate action sequences over the same two frames.  The
// prompt was prefilled once per group and its K / V rows [0, P) live ONCE, in cache row `slot = (b - row0) / G` of the chunk (the rows
// a prefill of the group's prompt wrote); a trajectory's own rows -- positions >= P: the prompt's last token, which carries the row's
// action, and everything generated -- live in its own cache row b.  The two position ranges never overlap, so no second buffer
// exists: a key row t is read from the group's cache row when t < P and from the trajectory's otherwise.  The same arithmetic in the
// same order as the un-shared kernel on the same bytes; the prefix rows use default-policy loads (the other rows of the group ask for
// the same lines: they are served from L2 / Infinity Cache instead of HBM).
template <typename T, bool NT, int HD, bool SHARED = false>
__global__ __launch_bounds__(256) void decode_attn_kernel(const T* __restrict__ qkv, T* __restrict__ kc, T* __restrict__ vc,
                                                          T* __restrict__ out, const float* __restrict__ cosT,
                                                          const float* __restrict__ sinT, int heads, int hd_arg, int Lmax,
                                                          const StepState* __restrict__ state, unsigned long long* prof,
                                                          int sh_P = 0, int sh_G = 1, int sh_row0 = 0) {
  constexpr int VEC = Traits<T>::VEC;
  const int hd = HD > 0 ? HD : hd_arg;
  constexpr int UNR = 8;   // 16-byte loads in flight per lane: 3 workgroups x 256 lanes x 8 x 16 B = 96 KiB per CU
  // measurement hook (bench.py roofline): launch window = [min start, max end] over workgroups on the 100 MHz wall clock,
  // reduced per slot here (min kept as max of the complement so that 0 = not stamped) and over slots by the host
  const unsigned long long t_start = prof ? wall_clock64() : 0ull;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lpk = hd / VEC;          // lanes per key row (8 for bf16 hd=64, 16 for fp32)
  const int gpb = 256 / lpk;         // key groups per workgroup
  float* sq = (float*)smem;          // [hd] roped q   | [hd] roped new k | [hd] new v  (values already rounded to T)
  float* sk = sq + hd;
  float* sv = sk + hd;
  float* sc = sv + hd;               // [Lmax] scores
  float* red = sc + Lmax;            // [gpb][hd] partial outputs
  __shared__ float sred[8];
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int sub = tid % lpk, grp = tid / lpk;
  const int H = heads * hd, half = hd / 2;
  const float scale = rsqrtf((float)hd);
  T* kb = kc + ((long)b * heads + h) * Lmax * hd;
  T* vb = vc + ((long)b * heads + h) * Lmax * hd;
  // SHARED: byte distance from this trajectory's head base to its group's (the prefix rows t < sh_P are read there); 0 otherwise
  long sh_delta = 0;
  if constexpr (SHARED) sh_delta = ((long)((b - sh_row0) / sh_G) - b) * heads * Lmax * hd * (long)sizeof(T);
  const int step = gpb * UNR;
  // rows beyond the cached keys are not requested (on average half of the last block of 256: re-reading a clamped row instead
  // measured +1 us per launch); 32-bit byte offsets from the (wave-uniform) head base
  const unsigned lane_off = (unsigned)(sub * VEC) * (unsigned)sizeof(T);
  // `limit`: rows below it are requested.  Every call but the first passes the number of cached keys; the FIRST round of key rows goes
  // out before the step counter is even read (limit = Lmax: the rows exist in the cache buffer whatever they hold; rows >= pos are
  // never used -- the score and value passes test t < pos) -- the counter's load, a dependent memory round trip at the head of every
  // launch, then overlaps that round instead of preceding it (round 6)
  auto load_rows = [&](Chunk16 (&dst)[UNR], const T* base, int t0, int limit) {
    const char* bb = (const char*)base;
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int t = t0 + u * gpb + grp;
      const char* rb = bb;
      if constexpr (SHARED) rb = t < sh_P ? bb + sh_delta : bb;
      const Chunk16* src = (const Chunk16*)(rb + ((unsigned)(t * hd) * (unsigned)sizeof(T) + lane_off));
      dst[u] = t < limit ? ((NT && !SHARED) ? __builtin_nontemporal_load(src) : *src) : Chunk16{0u, 0u, 0u, 0u};
    }
  };
  Chunk16 cur[UNR], nxt[UNR];
  // the few loads of the RoPE step go out BEFORE the first round of key rows: memory returns in order, so issued after them
  // they (and the barrier behind them) would wait for the whole 32 KiB round
  float rc = 0.f, rs = 0.f, q1 = 0.f, q2 = 0.f, k1 = 0.f, k2 = 0.f;
  T va = from_f32<T>(0.f), vb2 = from_f32<T>(0.f);
  if (tid < half) {
    const T* row = qkv + (long)b * 3 * H + h * hd;
    q1 = to_f32(row[tid]); q2 = to_f32(row[tid + half]);
    k1 = to_f32(row[H + tid]); k2 = to_f32(row[H + tid + half]);
    va = row[2 * H + tid]; vb2 = row[2 * H + tid + half];
  }
  load_rows(cur, kb, 0, Lmax);  // the first key rows are in flight while the step counter arrives and q is roped
  const int pos = state->pos;        // position of the token being fed = number of cached keys
  const int n_keys = pos + 1;
  if (tid < half) { rc = cosT[(long)pos * half + tid]; rs = sinT[(long)pos * half + tid]; }
  if (tid < half) {  // RoPE (HF rotate_half) of q and the new k; append k, v to the cache
    const T qa = from_f32<T>(q1 * rc - q2 * rs), qb = from_f32<T>(q2 * rc + q1 * rs);
    const T ka = from_f32<T>(k1 * rc - k2 * rs), kb2 = from_f32<T>(k2 * rc + k1 * rs);
    sq[tid] = to_f32(qa); sq[tid + half] = to_f32(qb);
    sk[tid] = to_f32(ka); sk[tid + half] = to_f32(kb2);
    sv[tid] = to_f32(va); sv[tid + half] = to_f32(vb2);
    kb[(long)pos * hd + tid] = ka; kb[(long)pos * hd + tid + half] = kb2;
    vb[(long)pos * hd + tid] = va; vb[(long)pos * hd + tid + half] = vb2;
  }
  __syncthreads();
  float qf[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) qf[j] = sq[sub * VEC + j];
  // bf16: q (already rounded to bf16) as packed pairs for v_dot2c_f32_bf16 -- two products per instruction, no unpacking of
  // the key chunk (the products are exact either way; 4 instead of 16 instructions per 16-byte chunk)
  QPack<T> qp;
  qp.set(qf);
  // pass A: scores of the cached keys (the new key comes from LDS); the next rows are requested before the current
  // ones are consumed, and the first value rows before the softmax statistics.  The two row buffers swap roles every
  // iteration (unrolled by two: no register copies); the value rows requested last end up in `cur` for pass C.
  auto scores = [&](Chunk16 (&rows)[UNR], int t0) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int t = t0 + u * gpb + grp;
      float d = qp.dot(qf, rows[u]);
      d = group_sum(d, lpk);
      if (t < pos && sub == 0) sc[t] = d * scale;
    }
  };
  {
    int t0 = 0;
    bool in_nxt = false;   // the rows to consume next are in `nxt`
    while (t0 < pos) {
      if (t0 + step < pos) load_rows(nxt, kb, t0 + step, pos); else load_rows(nxt, vb, 0, pos);
      scores(cur, t0);
      t0 += step;
      in_nxt = true;
      if (t0 >= pos) break;
      if (t0 + step < pos) load_rows(cur, kb, t0 + step, pos); else load_rows(cur, vb, 0, pos);
      scores(nxt, t0);
      t0 += step;
      in_nxt = false;
    }
    if (in_nxt) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) cur[u] = nxt[u];
    }
  }
  if (grp == 0) {
    float d = 0.f;
#pragma unroll
    for (int j = 0; j < VEC; ++j) d = fmaf(qf[j], sk[sub * VEC + j], d);
    d = group_sum(d, lpk);
    if (sub == 0) sc[pos] = d * scale;
  }
  __syncthreads();
  // pass B: softmax statistics
  float mx = -INFINITY;
  for (int t = tid; t < n_keys; t += 256) mx = fmaxf(mx, sc[t]);
  mx = wave_max(mx);
  if (lane == 0) sred[wv] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(sred[0], sred[1]), fmaxf(sred[2], sred[3]));
  float sum = 0.f;
  for (int t = tid; t < n_keys; t += 256) { const float e = expf(sc[t] - mx); sc[t] = e; sum += e; }
  sum = wave_sum(sum);
  if (lane == 0) sred[4 + wv] = sum;
  __syncthreads();
  sum = (sred[4] + sred[5]) + (sred[6] + sred[7]);
  // pass C: weighted V sum; group `grp` takes keys grp, grp+gpb, ... (fixed order), the new token's v from LDS
  float of[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) of[j] = 0.f;
  auto weighted = [&](Chunk16 (&rows)[UNR], int t0) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int t = t0 + u * gpb + grp;
      if (t < pos) axpy_chunk<T>(of, sc[t], rows[u]);
    }
  };
    for (int t0 = 0; t0 < pos; t0 += 2 * step) {  // cur holds value rows [t0, t0 + step) (requested during pass A / the last iteration)
      if (t0 + step < pos) load_rows(nxt, vb, t0 + step, pos);
      weighted(cur, t0);
      if (t0 + step >= pos) break;
      if (t0 + 2 * step < pos) load_rows(cur, vb, t0 + 2 * step, pos);
      weighted(nxt, t0 + step);
    }
  if (grp == 0) {
    const float pw = sc[pos];
#pragma unroll
    for (int j = 0; j < VEC; ++j) of[j] = fmaf(pw, sv[sub * VEC + j], of[j]);
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) red[grp * hd + sub * VEC + j] = of[j];
  __syncthreads();
  if (tid < hd) {
    float a = 0.f;
    for (int g = 0; g < gpb; ++g) a += red[g * hd + tid];
    out[(long)b * H + h * hd + tid] = from_f32<T>(a / sum);
  }
  if (prof && tid == 0) {
    unsigned long long* slot = prof + (size_t)((blockIdx.x * 7 + blockIdx.y) % IVG_ATTN_PROF_SLOTS) * 2 * Lmax;
    atomicMax(slot + pos, ~t_start);
    atomicMax(slot + Lmax + pos, (unsigned long long)wall_clock64());
  }
}

int launch_decode_attn(const void* qkv, void* kc, void* vc, void* out, const float* cosT, const float* sinT, int B, int heads, int hd,
                       int Lmax, const StepState* state, unsigned long long* prof, DType dt, hipStream_t st, int sh_P, int sh_G, int sh_row0) {
  const int vec = dt == BF16 ? 8 : 4;
  if (hd % vec != 0 || hd > 256 || 256 % (hd / vec) != 0 || (hd & 1)) return (int)hipErrorInvalidValue;
  const int gpb = 256 / (hd / vec);
  const size_t smem = (size_t)(3 * hd + Lmax + gpb * hd) * sizeof(float);
  dim3 g(B * heads);
  // non-temporal cache-row loads (bf16): 5.57 -> 6.28 TB/s on a pure stream, 203 -> 191 ms per rollout
#define IVG_DA(T, NTv, HDv) \
  hipLaunchKernelGGL((decode_attn_kernel<T, NTv, HDv>), g, dim3(256), smem, st, (const T*)qkv, (T*)kc, (T*)vc, (T*)out, cosT, sinT, heads, hd, Lmax, state, prof, 0, 1, 0)
  if (sh_G > 1) {   // shared-context rollout: prefix rows from the group's cache row
    if (sh_P < 0 || sh_row0 > 0) return (int)hipErrorInvalidValue;
#define IVG_DAS(T, NTv, HDv) \
  hipLaunchKernelGGL((decode_attn_kernel<T, NTv, HDv, true>), g, dim3(256), smem, st, (const T*)qkv, (T*)kc, (T*)vc, (T*)out, cosT, sinT, heads, hd, Lmax, state, prof, sh_P, sh_G, sh_row0)
    if (dt == BF16) { if (hd == 64) IVG_DAS(bf16_t, true, 64); else IVG_DAS(bf16_t, true, 0); }
    else { if (hd == 64) IVG_DAS(float, false, 64); else IVG_DAS(float, false, 0); }
#undef IVG_DAS
    return (int)hipGetLastError();
  }
  if (dt == BF16) { if (hd == 64) IVG_DA(bf16_t, true, 64); else IVG_DA(bf16_t, true, 0); }
  else { if (hd == 64) IVG_DA(float, false, 64); else IVG_DA(float, false, 0); }
#undef IVG_DA
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ 24-bit K / V cache of the x3 rollout
// The 1e-3-compliant rollout mode (IVG_F32X3) keeps fp32 tensors, and its decode attention streamed an fp32 cache: 249 MB per layer
// and step at config 2, 130 of the mode's 425 ms per step (45.8 us per launch; over the planes below: 32.8 us, 93 ms).  The arithmetic of that mode carries 2^-17 per operand anyway (bf16 hi +
// bf16 lo), so the cache keeps 24 of the 32 bits: sign, exponent and 15 mantissa bits, rounded to nearest even -- 2^-17 relative --
// as TWO PLANES per (trajectory, head): the upper 16 bits of every element ([Lmax][64] uint16, a bf16 image of the row) followed by
// the next 8 bits ([Lmax][64] uint8).  A key row is 128 + 64 bytes instead of 256; a lane reads 16 + 8 bytes (8 elements) and
// rebuilds each fp32 value with one v_perm_b32.  Everything else is decode_attn_kernel's structure (8 lanes per key row, as its bf16
// instance): first round of key rows before the step counter, scores -> LDS, softmax statistics, weighted value sum in fixed order.
// The appended k / v are rounded BEFORE they are used for this step's own score and output, so a later step reads exactly what
// this one computed with.  The prefill keeps fp32 K / V of one layer in scratch (its score GEMM reads K as a matrix) and packs the
// rows into the planes afterwards (kv24_pack_kernel).
__device__ __forceinline__ unsigned f24_round(float x) {   // fp32 bits rounded to 24 (RNE), low byte zero
  unsigned u = __float_as_uint(x);
  u += 0x7fu + ((u >> 8) & 1u);
  return u & 0xffffff00u;
}
struct Row24 { Chunk16 hi; unsigned lo0, lo1; };
__device__ __forceinline__ void row24_unpack(const Row24& r, float (&f)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const unsigned h = r.hi[j >> 1];
    const unsigned l = j < 4 ? r.lo0 : r.lo1;
    // result bytes: [0] = 0, [1] = byte (j & 3) of the low plane's dword, [2..3] = the element's 16 upper bits
    const unsigned sel = (j & 1) ? (0x07060000u | ((unsigned)(j & 3) << 8) | 0x0cu) : (0x05040000u | ((unsigned)(j & 3) << 8) | 0x0cu);
    f[j] = __uint_as_float(__builtin_amdgcn_perm(h, l, sel));
  }
}

template <bool SHARED>
__global__ __launch_bounds__(256) void decode_attn24_kernel(const float* __restrict__ qkv, unsigned char* __restrict__ kc, unsigned char* __restrict__ vc,
                                                            float* __restrict__ out, const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                            int heads, int Lmax, const StepState* __restrict__ state, unsigned long long* prof,
                                                            int sh_P, int sh_G, int sh_row0) {
  constexpr int HD = 64, HALF = 32, LPK = 8, GPB = 32, UNR = 8;
  const unsigned long long t_start = prof ? wall_clock64() : 0ull;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* sq = (float*)smem;
  float* sk = sq + HD;
  float* sv = sk + HD;
  float* sc = sv + HD;               // [Lmax] scores
  float* red = sc + Lmax;            // [GPB][HD] partial outputs
  __shared__ float sred[8];
  const int b = blockIdx.x / heads, h = blockIdx.x % heads;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int sub = tid % LPK, grp = tid / LPK;
  const int H = heads * HD;
  const float scale = rsqrtf((float)HD);
  const long blk = (long)Lmax * (HD * 3);                          // bytes of one (trajectory, head): [Lmax][64] u16 | [Lmax][64] u8
  unsigned char* kb = kc + ((long)b * heads + h) * blk;
  unsigned char* vb = vc + ((long)b * heads + h) * blk;
  const unsigned lo_plane = (unsigned)Lmax * (HD * 2);
  long sh_delta = 0;
  if constexpr (SHARED) sh_delta = ((long)((b - sh_row0) / sh_G) - b) * heads * blk;
  const int step = GPB * UNR;
  auto load_rows = [&](Row24 (&dst)[UNR], const unsigned char* base, int t0, int limit) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int t = t0 + u * GPB + grp;
      const unsigned char* rb = base;
      if constexpr (SHARED) rb = t < sh_P ? base + sh_delta : base;
      const Chunk16* ph = (const Chunk16*)(rb + ((unsigned)t * (HD * 2) + (unsigned)sub * 16u));
      typedef unsigned int U2 __attribute__((ext_vector_type(2)));
      const U2* pl = (const U2*)(rb + (lo_plane + (unsigned)t * HD + (unsigned)sub * 8u));
      if (t < limit) {
        if constexpr (SHARED) { dst[u].hi = *ph; const U2 l2 = *pl; dst[u].lo0 = l2[0]; dst[u].lo1 = l2[1]; }
        else { dst[u].hi = __builtin_nontemporal_load(ph); const U2 l2 = __builtin_nontemporal_load(pl); dst[u].lo0 = l2[0]; dst[u].lo1 = l2[1]; }
      } else {
        dst[u].hi = Chunk16{0u, 0u, 0u, 0u}; dst[u].lo0 = 0u; dst[u].lo1 = 0u;
      }
    }
  };
  Row24 cur[UNR], nxt[UNR];
  float rc = 0.f, rs = 0.f, q1 = 0.f, q2 = 0.f, k1 = 0.f, k2 = 0.f, va = 0.f, vb2 = 0.f;
  if (tid < HALF) {
    const float* row = qkv + (long)b * 3 * H + h * HD;
    q1 = row[tid]; q2 = row[tid + HALF];
    k1 = row[H + tid]; k2 = row[H + tid + HALF];
    va = row[2 * H + tid]; vb2 = row[2 * H + tid + HALF];
  }
  load_rows(cur, kb, 0, Lmax);       // the first key rows are in flight while the step counter arrives and q is roped
  const int pos = state->pos;
  const int n_keys = pos + 1;
  if (tid < HALF) { rc = cosT[(long)pos * HALF + tid]; rs = sinT[(long)pos * HALF + tid]; }
  if (tid < HALF) {
    const unsigned ka = f24_round(k1 * rc - k2 * rs), kb2 = f24_round(k2 * rc + k1 * rs);
    const unsigned v1 = f24_round(va), v2 = f24_round(vb2);
    sq[tid] = q1 * rc - q2 * rs; sq[tid + HALF] = q2 * rc + q1 * rs;
    sk[tid] = __uint_as_float(ka); sk[tid + HALF] = __uint_as_float(kb2);
    sv[tid] = __uint_as_float(v1); sv[tid + HALF] = __uint_as_float(v2);
    unsigned short* kh = (unsigned short*)kb + (long)pos * HD;
    unsigned short* vh = (unsigned short*)vb + (long)pos * HD;
    unsigned char* kl = kb + lo_plane + (long)pos * HD;
    unsigned char* vl = vb + lo_plane + (long)pos * HD;
    kh[tid] = (unsigned short)(ka >> 16); kh[tid + HALF] = (unsigned short)(kb2 >> 16);
    kl[tid] = (unsigned char)(ka >> 8); kl[tid + HALF] = (unsigned char)(kb2 >> 8);
    vh[tid] = (unsigned short)(v1 >> 16); vh[tid + HALF] = (unsigned short)(v2 >> 16);
    vl[tid] = (unsigned char)(v1 >> 8); vl[tid + HALF] = (unsigned char)(v2 >> 8);
  }
  __syncthreads();
  float qf[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) qf[j] = sq[sub * 8 + j];
  auto scores = [&](Row24 (&rows)[UNR], int t0) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int t = t0 + u * GPB + grp;
      float kf[8];
      row24_unpack(rows[u], kf);
      float d = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) d = fmaf(qf[j], kf[j], d);
      d = group_sum(d, LPK);
      if (t < pos && sub == 0) sc[t] = d * scale;
    }
  };
  {
    int t0 = 0;
    bool in_nxt = false;
    while (t0 < pos) {
      if (t0 + step < pos) load_rows(nxt, kb, t0 + step, pos); else load_rows(nxt, vb, 0, pos);
      scores(cur, t0);
      t0 += step;
      in_nxt = true;
      if (t0 >= pos) break;
      if (t0 + step < pos) load_rows(cur, kb, t0 + step, pos); else load_rows(cur, vb, 0, pos);
      scores(nxt, t0);
      t0 += step;
      in_nxt = false;
    }
    if (in_nxt) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) cur[u] = nxt[u];
    }
  }
  if (grp == 0) {
    float d = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) d = fmaf(qf[j], sk[sub * 8 + j], d);
    d = group_sum(d, LPK);
    if (sub == 0) sc[pos] = d * scale;
  }
  __syncthreads();
  float mx = -INFINITY;
  for (int t = tid; t < n_keys; t += 256) mx = fmaxf(mx, sc[t]);
  mx = wave_max(mx);
  if (lane == 0) sred[wv] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(sred[0], sred[1]), fmaxf(sred[2], sred[3]));
  float sum = 0.f;
  for (int t = tid; t < n_keys; t += 256) { const float e = expf(sc[t] - mx); sc[t] = e; sum += e; }
  sum = wave_sum(sum);
  if (lane == 0) sred[4 + wv] = sum;
  __syncthreads();
  sum = (sred[4] + sred[5]) + (sred[6] + sred[7]);
  float of[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) of[j] = 0.f;
  auto weighted = [&](Row24 (&rows)[UNR], int t0) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int t = t0 + u * GPB + grp;
      if (t < pos) {
        float vf[8];
        row24_unpack(rows[u], vf);
        const float pw = sc[t];
#pragma unroll
        for (int j = 0; j < 8; ++j) of[j] = fmaf(pw, vf[j], of[j]);
      }
    }
  };
  for (int t0 = 0; t0 < pos; t0 += 2 * step) {
    if (t0 + step < pos) load_rows(nxt, vb, t0 + step, pos);
    weighted(cur, t0);
    if (t0 + step >= pos) break;
    if (t0 + 2 * step < pos) load_rows(cur, vb, t0 + 2 * step, pos);
    weighted(nxt, t0 + step);
  }
  if (grp == 0) {
    const float pw = sc[pos];
#pragma unroll
    for (int j = 0; j < 8; ++j) of[j] = fmaf(pw, sv[sub * 8 + j], of[j]);
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[grp * HD + sub * 8 + j] = of[j];
  __syncthreads();
  if (tid < HD) {
    float a = 0.f;
    for (int g = 0; g < GPB; ++g) a += red[g * HD + tid];
    out[(long)b * H + h * HD + tid] = a / sum;
  }
  if (prof && tid == 0) {
    unsigned long long* slot = prof + (size_t)((blockIdx.x * 7 + blockIdx.y) % IVG_ATTN_PROF_SLOTS) * 2 * Lmax;
    atomicMax(slot + pos, ~t_start);
    atomicMax(slot + Lmax + pos, (unsigned long long)wall_clock64());
  }
}

static std::atomic<long long> g_attn24_launches{0};
long long decode_attn24_launches() { return g_attn24_launches.load(); }

int launch_decode_attn24(const void* qkv, void* kc, void* vc, void* out, const float* cosT, const float* sinT, int B, int heads, int Lmax,
                         const StepState* state, unsigned long long* prof, hipStream_t st, int sh_P, int sh_G, int sh_row0) {
  g_attn24_launches.fetch_add(1, std::memory_order_relaxed);
  const size_t smem = (size_t)(3 * 64 + Lmax + 32 * 64) * sizeof(float);
  dim3 g(B * heads);
  if (sh_G > 1) {
    if (sh_P < 0 || sh_row0 > 0) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(decode_attn24_kernel<true>, g, dim3(256), smem, st, (const float*)qkv, (unsigned char*)kc, (unsigned char*)vc, (float*)out, cosT, sinT,
                       heads, Lmax, state, prof, sh_P, sh_G, sh_row0);
  } else {
    hipLaunchKernelGGL(decode_attn24_kernel<false>, g, dim3(256), smem, st, (const float*)qkv, (unsigned char*)kc, (unsigned char*)vc, (float*)out, cosT, sinT,
                       heads, Lmax, state, prof, 0, 1, 0);
  }
  return (int)hipGetLastError();
}

// fp32 rows [0, L) of K / V ([B * heads][Lmax][64], what rope_kv wrote for the prompt) -> the 24-bit planes of the cache
__global__ __launch_bounds__(256) void kv24_pack_kernel(const float* __restrict__ k32, const float* __restrict__ v32, unsigned char* __restrict__ kc,
                                                        unsigned char* __restrict__ vc, int L, int Lmax) {
  const int bh = blockIdx.y, tid = threadIdx.x;
  const int row = blockIdx.x * 32 + (tid >> 3), sub = tid & 7;
  if (row >= L) return;
  const long blk = (long)Lmax * 192;
  const unsigned lo_plane = (unsigned)Lmax * 128u;
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    const float* src = (which ? v32 : k32) + ((long)bh * Lmax + row) * 64 + sub * 8;
    unsigned char* dst = (which ? vc : kc) + (long)bh * blk;
    const f32x4 a = *(const f32x4*)src, c = *(const f32x4*)(src + 4);
    unsigned u[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) { u[j] = f24_round(a[j]); u[4 + j] = f24_round(c[j]); }
    Chunk16 hi;
#pragma unroll
    for (int j = 0; j < 4; ++j) hi[j] = (u[2 * j] >> 16) | (u[2 * j + 1] & 0xffff0000u);
    uint2 lo;
    lo.x = ((u[0] >> 8) & 0xffu) | (u[1] & 0xff00u) | ((u[2] << 8) & 0xff0000u) | ((u[3] << 16) & 0xff000000u);
    lo.y = ((u[4] >> 8) & 0xffu) | (u[5] & 0xff00u) | ((u[6] << 8) & 0xff0000u) | ((u[7] << 16) & 0xff000000u);
    *(Chunk16*)(dst + (unsigned)row * 128u + (unsigned)sub * 16u) = hi;
    *(uint2*)(dst + lo_plane + (unsigned)row * 64u + (unsigned)sub * 8u) = lo;
  }
}

int launch_kv24_pack(const void* k32, const void* v32, void* kc, void* vc, int BH, int L, int Lmax, hipStream_t st) {
  if (L <= 0) return 0;
  hipLaunchKernelGGL(kv24_pack_kernel, dim3((unsigned)cdiv(L, 32), (unsigned)BH), dim3(256), 0, st, (const float*)k32, (const float*)v32, (unsigned char*)kc,
                     (unsigned char*)vc, L, Lmax);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ sampling
// One workgroup per trajectory.  Restates HF TopKLogitsWarper(top_k) + softmax + one draw as an
// explicit-uniform inverse CDF over the kept tokens in ascending id order (oracle/llama.py
// sample_from_logits); uniforms == null -> greedy argmax (lowest id on ties).  Then embeds the decided
// token as the next input row and (forced sdf slots) adds the action embedding.
__device__ __forceinline__ unsigned f2key(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // monotone: larger float -> larger key
}

constexpr int SAMPLE_CAP = 1024;  // kept-token list (ids + values) of the fast inverse-CDF path

// exclusive suffix sum over the 256 threads of a workgroup (thread t receives the sum of threads t+1 .. 255)
__device__ __forceinline__ int block_suffix_excl(int x, int* s4, int lane, int wv) {
  int incl = x;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int dn = __shfl_down(incl, o, 64);
    if (lane + o < 64) incl += dn;
  }
  if (lane == 0) s4[wv] = incl;  // wave total
  __syncthreads();
  int above = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) if (w > wv) above += s4[w];
  __syncthreads();
  return incl - x + above;
}

template <typename T, int KPT>  // KPT logits per thread (contiguous segment): vocab <= 256 * KPT
__global__ __launch_bounds__(256) void sample_embed_kernel(SampleArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* lg = (float*)smem;  // [V] staging: coalesced global read, then each thread takes a contiguous segment
  __shared__ float s_f[4];
  __shared__ int s_i[4];
  __shared__ int s4[4];
  __shared__ int s_sel[2];
  __shared__ __attribute__((aligned(16))) int hist[2048];
  __shared__ int kid[SAMPLE_CAP];
  __shared__ float cval[SAMPLE_CAP];
  __shared__ double s_w[4];
  __shared__ long s_tok;
  __shared__ double s_target;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int j = a.state->j;
  const int V = a.V;
  const bool forced = a.forced_period > 0 && (j % a.forced_period) == 0;
  long tok = 0;
  if (forced) {
    tok = a.forced_token;
  } else if (j == 0) {
    // step 0 only ever FEEDS the prompt's last token (kept-cache and shared-context rollouts: the cache holds positions [0, L0 - 1)):
    // nothing is decided, the token is read back from the id row (the store below rewrites it with itself)
    tok = a.ids_out[(long)b * a.ids_stride + a.L0 - 1];
  } else {
    const float* src = a.logits + (long)b * V;
    if ((V & 1) == 0 && (((uintptr_t)src) & 7) == 0) {
      // 8-byte loads, 16 in flight per lane (one workgroup per trajectory: only B CUs pull the logits, so the row must
      // arrive in as few round trips as possible; a scalar loop here was most of the kernel's time)
      const float2* src2 = (const float2*)src;
      float2* lg2 = (float2*)lg;
      const int n2 = V >> 1;
      for (int i0b = 0; i0b < n2; i0b += 256 * 16) {
        float2 tmp[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) { const int i = i0b + r * 256 + tid; tmp[r] = i < n2 ? src2[i] : float2{0.f, 0.f}; }
#pragma unroll
        for (int r = 0; r < 16; ++r) { const int i = i0b + r * 256 + tid; if (i < n2) lg2[i] = tmp[r]; }
      }
    } else {
      for (int i = tid; i < V; i += 256) lg[i] = src[i];
    }
    if (tid == 0) s_tok = -1;
    __syncthreads();
    // HF TemperatureLogitsWarper runs BEFORE the top-k filter: scores = scores / temperature in fp32 (IEEE division, as torch's);
    // every later read of the row (candidates, exponentials) sees the scaled values.  1.0 (every caller of the reference): untouched.
    if (a.temperature != 1.0f) {
      for (int i = tid; i < V; i += 256) lg[i] = lg[i] / a.temperature;
      __syncthreads();
    }
    const int seg = (V + 255) / 256;
    const int i0 = tid * seg;
    const int nvalid = max(0, min(seg, V - i0));   // this thread owns ids [i0, i0 + nvalid)
    float v[KPT];
    float mx = -INFINITY;
    int mi = 0x7fffffff;
#pragma unroll
    for (int q = 0; q < KPT; ++q) v[q] = lg[min(i0 + q, V - 1)];   // unconditional: all reads are issued back to back
#pragma unroll
    for (int q = 0; q < KPT; ++q) {
      if (q >= nvalid) v[q] = -INFINITY;
      if (v[q] > mx) { mx = v[q]; mi = i0 + q; }  // ascending index: the first maximum is kept
    }
    const float tmax = mx;   // this thread's own maximum (pivot pre-filter of the sampler below)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {  // (max, lowest index) reduction
      const float ov = __shfl_xor(mx, o, 64);
      const int oi = __shfl_xor(mi, o, 64);
      if (ov > mx || (ov == mx && oi < mi)) { mx = ov; mi = oi; }
    }
    if (lane == 0) { s_f[wv] = mx; s_i[wv] = mi; }
    __syncthreads();
    mx = s_f[0]; mi = s_i[0];
#pragma unroll
    for (int w = 1; w < 4; ++w) if (s_f[w] > mx || (s_f[w] == mx && s_i[w] < mi)) { mx = s_f[w]; mi = s_i[w]; }
    if (a.uniforms == nullptr) {
      tok = mi;
    } else {
      // ---- key of the k-th largest logit (= largest T with #{key >= T} >= k).  General method: radix select, digits of
      //      11 / 11 / 10 bits, one LDS histogram per digit over the keys that match the digits found so far (exact counts)
      unsigned key[KPT];   // filled by the radix path only
      int kk = a.top_k < V ? a.top_k : V;
      unsigned thr = 0u;
      bool have_thr = false;
      int n_list = -1;                              // >= 0: (key, id, value) lists in LDS, ascending id order
      unsigned* s_ck = (unsigned*)hist + 256;       // [SAMPLE_CAP] keys of the listed tokens (hist is free when they are written)
      if (kk <= 256) {
        // Pre-filter (k <= 256): in every wave take the ceil(k/4)-th largest of its 64 per-thread maxima; the smallest of the
        // four is a lower bound of the k-th largest logit (at least k distinct elements are >= it), so only the tokens >=
        // that pivot -- a few more than k of them -- can hold the threshold.  They are listed once (key, id, value;
        // ascending id) and ranked against each other by counting: no histogram atomics and no second sweep over the
        // vocabulary on the hot path.
        const unsigned mine = f2key(tmax);
        int rank = 0;
#pragma unroll
        for (int jl = 0; jl < 64; ++jl) {
          const unsigned o = (unsigned)__builtin_amdgcn_readlane((int)mine, jl);
          rank += (o > mine || (o == mine && jl < lane)) ? 1 : 0;
        }
        if (rank == (kk + 3) / 4 - 1) s4[wv] = (int)mine;   // ranks are a permutation of 0..63: one writer per wave
        __syncthreads();
        unsigned pivot = (unsigned)s4[0];
#pragma unroll
        for (int w = 1; w < 4; ++w) pivot = (unsigned)s4[w] < pivot ? (unsigned)s4[w] : pivot;
        const float pivf = __uint_as_float((pivot & 0x80000000u) ? (pivot & 0x7fffffffu) : ~pivot);   // inverse of f2key
        unsigned cm[(KPT + 31) / 32];
#pragma unroll
        for (int w = 0; w < (KPT + 31) / 32; ++w) cm[w] = 0u;
#pragma unroll
        for (int q = 0; q < KPT; ++q) cm[q >> 5] |= (q < nvalid && v[q] >= pivf) ? (1u << (q & 31)) : 0u;
        int cc = 0;
#pragma unroll
        for (int w = 0; w < (KPT + 31) / 32; ++w) cc += __builtin_popcount(cm[w]);
        __syncthreads();                             // every thread has read s4
        const int hi = block_suffix_excl(cc, s4, lane, wv);
        if (tid == 0) s_sel[1] = hi + cc;           // number of candidates (>= k)
        __syncthreads();
        const int n_c = s_sel[1];
        if (n_c <= SAMPLE_CAP) {
          int off = n_c - hi - cc;                   // candidates in lower threads = lower ids
#pragma unroll
          for (int w = 0; w < (KPT + 31) / 32; ++w) {
            unsigned mm = cm[w];
            while (mm) {                             // a thread holds 0-3 candidates: read them back from the staged row
              const int id = i0 + w * 32 + __builtin_ctz(mm);
              mm &= mm - 1;
              const float val = lg[id];
              s_ck[off] = f2key(val); kid[off] = id; cval[off] = val; ++off;
            }
          }
          if (tid < 4) s_ck[n_c + tid] = 0u;           // pad to a multiple of 4 (hist has room beyond CAP)
          __syncthreads();
          unsigned best = 0xffffffffu;               // smallest candidate with fewer than k strictly larger ones = k-th largest
          const uint4* c4 = (const uint4*)s_ck;
          const int n4 = (n_c + 3) >> 2;
          for (int i = tid; i < n_c; i += 256) {
            const unsigned ci = s_ck[i];
            int g = 0;
#pragma unroll 8
            for (int j4 = 0; j4 < n4; ++j4) {
              const uint4 o = c4[j4];
              g += (o.x > ci ? 1 : 0) + (o.y > ci ? 1 : 0) + (o.z > ci ? 1 : 0) + (o.w > ci ? 1 : 0);
            }
            if (g < kk && ci < best) best = ci;
          }
#pragma unroll
          for (int o = 32; o > 0; o >>= 1) { const unsigned ob = __shfl_xor(best, o, 64); best = ob < best ? ob : best; }
          if (lane == 0) s4[wv] = (int)best;
          __syncthreads();
          thr = (unsigned)s4[0];
#pragma unroll
          for (int w = 1; w < 4; ++w) thr = (unsigned)s4[w] < thr ? (unsigned)s4[w] : thr;
          have_thr = true;
          n_list = n_c;
          __syncthreads();                           // s4 is reused below
        }
      }
      if (!have_thr) {   // k > 256 or a flood of candidates (massive ties): radix select over all keys
#pragma unroll
      for (int q = 0; q < KPT; ++q) key[q] = f2key(v[q]);
      unsigned prefix = 0u, pmask = 0u;
#pragma unroll
      for (int pass = 0; pass < 3; ++pass) {
        const int shift = pass == 0 ? 21 : (pass == 1 ? 10 : 0);
        const int nb = pass == 2 ? 1024 : 2048, per = nb / 256;
        for (int i = tid; i < nb; i += 256) hist[i] = 0;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < KPT; ++q)
          if (q < seg && i0 + q < V && (key[q] & pmask) == prefix) atomicAdd(&hist[(key[q] >> shift) & (nb - 1)], 1);
        __syncthreads();
        int mine = 0;
        for (int r = 0; r < per; ++r) mine += hist[tid * per + r];
        const int above = block_suffix_excl(mine, s4, lane, wv);  // keys in bins above this thread's bins
        if (above < kk && kk <= above + mine) {                    // exactly one thread: the digit is in its bins
          int acc = above, d = tid * per + per - 1;
          for (; d > tid * per; --d) { if (acc + hist[d] >= kk) break; acc += hist[d]; }
          s_sel[0] = d; s_sel[1] = kk - acc;
        }
        __syncthreads();
        prefix |= (unsigned)s_sel[0] << shift;
        pmask |= (unsigned)(nb - 1) << shift;
        kk = s_sel[1];
      }
      thr = prefix;
      }
      // everything >= thr is kept (ties at the threshold included, as HF's masked_fill)
      // ---- inverse CDF in ascending id order, fp64 (oracle: double cumsum of exp(logit - max) over the kept ids)
      if (n_list < 0) {   // radix path: list the kept tokens now
        int cnt = 0;
#pragma unroll
        for (int q = 0; q < KPT; ++q) cnt += (key[q] >= thr && v[q] > -INFINITY) ? 1 : 0;
        const int after = block_suffix_excl(cnt, s4, lane, wv);
        if (tid == 0) s_sel[0] = after + cnt;  // number of kept tokens
        __syncthreads();
        const int n_kept = s_sel[0];
        if (n_kept <= SAMPLE_CAP) {
          int off = n_kept - after - cnt;  // kept tokens in lower threads = lower ids
#pragma unroll
          for (int q = 0; q < KPT; ++q)
            if (key[q] >= thr && v[q] > -INFINITY) { s_ck[off] = key[q]; cval[off] = v[q]; kid[off] = i0 + q; ++off; }
          __syncthreads();
          n_list = n_kept;
        }
      }
      if (n_list >= 0) {
        // the listed tokens (about top_k of them) are spread over the threads, so the fp64 exponentials run once each, in
        // parallel; listed tokens below the threshold (pre-filter path) weigh zero, exactly as in the oracle's masked sum
        const int per = (n_list + 255) / 256;  // <= 4
        const int e0 = tid * per, e1 = min(n_list, e0 + per);
        double ex[SAMPLE_CAP / 256];
        double part = 0.0;
#pragma unroll
        for (int r = 0; r < SAMPLE_CAP / 256; ++r) {
          ex[r] = 0.0;
          if (r < per) {   // uniform across the workgroup
            if (e0 + r < e1 && s_ck[e0 + r] >= thr && cval[e0 + r] > -INFINITY) ex[r] = exp((double)(cval[e0 + r] - mx));
            part += ex[r];
          }
        }
        double incl = part;  // inclusive scan over threads (thread order == id order)
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const double up = __shfl_up(incl, o, 64);
          if (lane >= o) incl += up;
        }
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        double base = 0.0, total = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { if (w < wv) base += s_w[w]; total += s_w[w]; }
        incl += base;
        double excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = base;
        const double target = (double)a.uniforms[(long)b * a.n_uni + (j - 1)] * total;
        // intervals [excl, incl) tile [0, total) exactly (incl of thread t IS excl of thread t+1)
        if (part > 0.0 && excl <= target && target < incl) {
          double run = excl;
          int found = -1, last = -1;
#pragma unroll
          for (int r = 0; r < SAMPLE_CAP / 256; ++r)
            if (r < per && ex[r] > 0.0) { run += ex[r]; last = kid[e0 + r]; if (found < 0 && run > target) found = last; }
          s_tok = found >= 0 ? found : last;  // rounding at the chunk edge: the chunk's last kept token
        }
        __syncthreads();
        tok = s_tok;
        if (tok < 0) tok = mi;  // unreachable for u in [0, 1): defensive
      } else {
        // general path (massive ties at the threshold): every thread walks its own id segment
        double part = 0.0;
#pragma unroll
        for (int q = 0; q < KPT; ++q)
          if (key[q] >= thr && v[q] > -INFINITY) part += exp((double)(v[q] - mx));
        double incl = part;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const double up = __shfl_up(incl, o, 64);
          if (lane >= o) incl += up;
        }
        if (lane == 63) s_w[wv] = incl;
        __syncthreads();
        double base = 0.0, total = 0.0;
#pragma unroll
        for (int w = 0; w < 4; ++w) { if (w < wv) base += s_w[w]; total += s_w[w]; }
        incl += base;
        double excl = __shfl_up(incl, 1, 64);
        if (lane == 0) excl = base;
        if (tid == 0) s_target = (double)a.uniforms[(long)b * a.n_uni + (j - 1)] * total;
        __syncthreads();
        const double target = s_target;
        if (part > 0.0 && excl <= target && target < incl) {
          double run = excl;
          long found = -1, last = -1;
#pragma unroll
          for (int q = 0; q < KPT; ++q) {
            if (key[q] >= thr && v[q] > -INFINITY) {
              run += exp((double)(v[q] - mx));
              last = i0 + q;
              if (found < 0 && run > target) found = i0 + q;
            }
          }
          s_tok = found >= 0 ? found : last;
        }
        __syncthreads();
        tok = s_tok;
        if (tok < 0) tok = mi;
      }
    }
  }
  // a row of NaN / -inf logits leaves the arg-max index at its initial value: never let it reach the id buffer or the
  // embedding gather (out-of-bounds read inside a replayed graph); such a row decides token 0
  if (tok < 0 || tok >= V) tok = 0;
  if (tid == 0) a.ids_out[(long)b * a.ids_stride + a.L0 + (j - 1)] = (int64_t)tok;
  // ---- next input embedding
  const T* src = (const T*)a.E + tok * a.H;
  T* dst = (T*)a.x + (long)b * a.H;
  const T* act = nullptr;
  if (forced && a.act) act = (const T*)a.act + ((long)b * a.act_T + (a.slot0 + j / a.forced_period + a.ctx - 1)) * a.H;
  for (int c = tid; c < a.H; c += 256) {
    float val = to_f32(src[c]);
    if (act) val += to_f32(act[c]);
    dst[c] = from_f32<T>(val);
  }
}

template <typename T, int KPT>
static void launch_sample_t(const SampleArgs& a, int B, size_t smem, hipStream_t st) {
  static DynLdsOnce once;
  (void)ensure_dyn_lds(once, (const void*)sample_embed_kernel<T, KPT>, 96 * 1024);
  hipLaunchKernelGGL((sample_embed_kernel<T, KPT>), dim3(B), dim3(256), smem, st, a);
}

int launch_sample_embed(const SampleArgs& a, int B, DType dt, hipStream_t st) {
  const size_t smem = (size_t)std::max(a.V, SAMPLE_CAP) * sizeof(float);
  if (a.V > 256 * 72 || smem > 96 * 1024) return (int)hipErrorInvalidValue;
  const bool small = a.V <= 256 * 36;
  if (dt == BF16) { if (small) launch_sample_t<bf16_t, 36>(a, B, smem, st); else launch_sample_t<bf16_t, 72>(a, B, smem, st); }
  else { if (small) launch_sample_t<float, 36>(a, B, smem, st); else launch_sample_t<float, 72>(a, B, smem, st); }
  return (int)hipGetLastError();
}

__global__ void state_set_kernel(StepState* s, int pos, int j) { s->pos = pos; s->j = j; }

int launch_state_set(StepState* state, int pos, int j, hipStream_t st) {
  hipLaunchKernelGGL(state_set_kernel, dim3(1), dim3(1), 0, st, state, pos, j);
  return (int)hipGetLastError();
}

// ids[r][0 .. L) = prompts[(b0 + r) / G][0 .. L): every trajectory of a group starts from a copy of its group's prompt
__global__ void expand_prompt_rows_kernel(const int64_t* __restrict__ prompts, long pstride, int64_t* __restrict__ ids, long ids_ld, int L,
                                          int G, int b0) {
  const int r = blockIdx.y;
  const int64_t* src = prompts + (long)((b0 + r) / G) * pstride;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < L; c += gridDim.x * blockDim.x) ids[(long)r * ids_ld + c] = src[c];
}

int launch_expand_prompt_rows(const int64_t* prompts, long pstride, int64_t* ids, long ids_ld, int rows, int L, int G, int b0, hipStream_t st) {
  if (rows <= 0 || L <= 0) return 0;
  hipLaunchKernelGGL(expand_prompt_rows_kernel, dim3((unsigned)((L + 255) / 256), (unsigned)rows), dim3(256), 0, st, prompts, pstride, ids, ids_ld, L, G, b0);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ small heads
template <typename T>
__global__ __launch_bounds__(256) void action_embed_kernel(const float* __restrict__ act, const float* __restrict__ W,
                                                           const float* __restrict__ bias, T* __restrict__ out, int A, int H) {
  const int row = blockIdx.x;
  for (int c = threadIdx.x; c < H; c += 256) {
    float s = 0.f;
    for (int k = 0; k < A; ++k) s = fmaf(act[(long)row * A + k], W[(long)c * A + k], s);
    out[(long)row * H + c] = from_f32<T>(s + bias[c]);
  }
}

int launch_action_embed(const float* act, const float* W, const float* bias, void* out, DType dt, int BT, int A, int H,
                        hipStream_t st) {
  if (BT <= 0) return 0;
  if (dt == BF16) hipLaunchKernelGGL(action_embed_kernel<bf16_t>, dim3(BT), dim3(256), 0, st, act, W, bias, (bf16_t*)out, A, H);
  else hipLaunchKernelGGL(action_embed_kernel<float>, dim3(BT), dim3(256), 0, st, act, W, bias, (float*)out, A, H);
  return (int)hipGetLastError();
}

template <typename T>
__global__ __launch_bounds__(256) void add_rows_kernel(T* __restrict__ x, long xs, const T* __restrict__ add, long as, int H) {
  const int b = blockIdx.x;
  for (int c = threadIdx.x; c < H; c += 256) x[b * xs + c] = from_f32<T>(to_f32(x[b * xs + c]) + to_f32(add[b * as + c]));
}

int launch_add_rows(void* x, long x_stride, const void* add, long add_stride, int B, int H, DType dt, hipStream_t st) {
  if (B <= 0) return 0;
  if (dt == BF16) hipLaunchKernelGGL(add_rows_kernel<bf16_t>, dim3(B), dim3(256), 0, st, (bf16_t*)x, x_stride, (const bf16_t*)add, add_stride, H);
  else hipLaunchKernelGGL(add_rows_kernel<float>, dim3(B), dim3(256), 0, st, (float*)x, x_stride, (const float*)add, add_stride, H);
  return (int)hipGetLastError();
}

// r[b] = rsqrt(mean(h^2) + eps) * dot(h[b], w) + bias: reward head on the final-normed hidden state (norm weight folded into w)
template <typename T>
__global__ __launch_bounds__(64) void rowdot_kernel(const T* __restrict__ h, const float* __restrict__ w,
                                                    const float* __restrict__ bias, float* __restrict__ out, int H, float eps) {
  const int b = blockIdx.x;
  float s = 0.f, ss = 0.f;
  for (int c = threadIdx.x; c < H; c += 64) { const float v = to_f32(h[(long)b * H + c]); s = fmaf(v, w[c], s); ss = fmaf(v, v, ss); }
  s = wave_sum(s);
  ss = wave_sum(ss);
  if (threadIdx.x == 0) out[b] = (eps >= 0.f ? s * rsqrtf(ss / (float)H + eps) : s) + bias[0];   // eps < 0: h is already normed
}

int launch_rowdot(const void* h, const float* w, const float* bias, float* out, int B, int H, float eps, DType dt, hipStream_t st) {
  if (B <= 0) return 0;
  if (dt == BF16) hipLaunchKernelGGL(rowdot_kernel<bf16_t>, dim3(B), dim3(64), 0, st, (const bf16_t*)h, w, bias, out, H, eps);
  else hipLaunchKernelGGL(rowdot_kernel<float>, dim3(B), dim3(64), 0, st, (const float*)h, w, bias, out, H, eps);
  return (int)hipGetLastError();
}

// out[b][:] = w * T(x[b][:] * rsqrt(mean x^2 + eps))  -- HF LlamaRMSNorm: the normalised row is rounded to the model dtype before
// the weight multiplies it.  This is `hidden_states[-1]` of the last layer as HF reports it (post final norm).
template <typename T>
__global__ __launch_bounds__(256) void final_hidden_kernel(const T* __restrict__ x, const float* __restrict__ w, T* __restrict__ out,
                                                           int H, float eps) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  float ss = 0.f;
  for (int c = tid; c < H; c += 256) { const float v = to_f32(x[(long)b * H + c]); ss = fmaf(v, v, ss); }
  ss = wave_sum(ss);
  if ((tid & 63) == 0) red[tid >> 6] = ss;
  __syncthreads();
  const float rs = rsqrtf(((red[0] + red[1]) + (red[2] + red[3])) / (float)H + eps);
  for (int c = tid; c < H; c += 256) {
    const T n = from_f32<T>(to_f32(x[(long)b * H + c]) * rs);
    out[(long)b * H + c] = from_f32<T>(to_f32(from_f32<T>(w[c])) * to_f32(n));
  }
}

int launch_final_hidden(const void* x, const float* w, void* out, int B, int H, float eps, DType dt, hipStream_t st) {
  if (B <= 0) return 0;
  if (dt == BF16) hipLaunchKernelGGL(final_hidden_kernel<bf16_t>, dim3(B), dim3(256), 0, st, (const bf16_t*)x, w, (bf16_t*)out, H, eps);
  else hipLaunchKernelGGL(final_hidden_kernel<float>, dim3(B), dim3(256), 0, st, (const float*)x, w, (float*)out, H, eps);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ eval heads
// Shifted cross-entropy of teacher-forced logits (HF ForCausalLM loss, train_gpt.py:356-376): row r = (b, l) of a chunk of logits
// [rows][V] fp32 predicts labels[b][l + 1]; nll[r] = logsumexp(logits[r]) - logits[r][target], 0 where the target is -100 (or l is
// the last position).  One workgroup per row, fixed-order reductions.
__global__ __launch_bounds__(256) void ce_rows_kernel(const float* __restrict__ logits, const int64_t* __restrict__ labels, long row0,
                                                      int L, int V, float* __restrict__ nll) {
  __shared__ float red[8];
  const long r = row0 + blockIdx.x;
  const int l = (int)(r % L), tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const long tgt = l + 1 < L ? labels[r + 1] : -100;
  if (tgt < 0 || tgt >= V) { if (tid == 0) nll[r] = 0.f; return; }
  const float* row = logits + (long)blockIdx.x * V;
  float mx = -INFINITY;
  for (int i = tid; i < V; i += 256) mx = fmaxf(mx, row[i]);
  mx = wave_max(mx);
  if (lane == 0) red[wv] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int i = tid; i < V; i += 256) sum += expf(row[i] - mx);
  sum = wave_sum(sum);
  if (lane == 0) red[4 + wv] = sum;
  __syncthreads();
  if (tid == 0) nll[r] = logf((red[4] + red[5]) + (red[6] + red[7])) + mx - row[tgt];
}

int launch_ce_rows(const float* logits, const int64_t* labels, long row0, int rows, int L, int V, float* nll, hipStream_t st) {
  if (rows <= 0) return 0;
  hipLaunchKernelGGL(ce_rows_kernel, dim3((unsigned)rows), dim3(256), 0, st, logits, labels, row0, L, V, nll);
  return (int)hipGetLastError();
}

// out[b] = { sum_l nll[b][l], number of positions l whose target labels[b][l + 1] is not ignored }
__global__ __launch_bounds__(256) void ce_reduce_kernel(const float* __restrict__ nll, const int64_t* __restrict__ labels, int L, int V,
                                                        float* __restrict__ out) {
  __shared__ float rs[4], rc[4];
  const int b = blockIdx.x, tid = threadIdx.x;
  float s = 0.f, c = 0.f;
  for (int l = tid; l + 1 < L; l += 256) {
    const long t = labels[(long)b * L + l + 1];
    if (t >= 0 && t < V) { s += nll[(long)b * L + l]; c += 1.f; }
  }
  s = wave_sum(s); c = wave_sum(c);
  if ((tid & 63) == 0) { rs[tid >> 6] = s; rc[tid >> 6] = c; }
  __syncthreads();
  if (tid == 0) { out[2 * b] = (rs[0] + rs[1]) + (rs[2] + rs[3]); out[2 * b + 1] = (rc[0] + rc[1]) + (rc[2] + rc[3]); }
}

int launch_ce_reduce(const float* nll, const int64_t* labels, int B, int L, int V, float* out, hipStream_t st) {
  if (B <= 0) return 0;
  hipLaunchKernelGGL(ce_reduce_kernel, dim3((unsigned)B), dim3(256), 0, st, nll, labels, L, V, out);
  return (int)hipGetLastError();
}

// action reconstruction error (action_model.py:187-196): every position p >= prelude of trajectory b predicts the action of its
// frame slot i = (p - prelude) / 17, a_hat = W h[b][p] + bias; out[b] = sum over positions and action dims of (a_hat - a)^2.
template <typename T>
__global__ __launch_bounds__(256) void action_recon_kernel(const T* __restrict__ hidden, const float* __restrict__ W, const float* __restrict__ bias,
                                                           const float* __restrict__ act, int L, int H, int A, int act_T, int ctx, int prelude,
                                                           float* __restrict__ out) {
  __shared__ float red[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float err = 0.f;   // lane 0 of every wave accumulates the positions that wave handles (fixed order)
  for (int p = prelude + wv; p < L; p += 4) {
    const int slot = (p - prelude) / 17;
    const T* h = hidden + ((long)b * L + p) * H;
    for (int a = 0; a < A; ++a) {
      float s = 0.f;
      for (int c = lane; c < H; c += 64) s = fmaf(to_f32(h[c]), W[(long)a * H + c], s);
      s = wave_sum(s);
      const float d = s + bias[a] - act[((long)b * act_T + ctx - 1 + slot) * A + a];
      err = fmaf(d, d, err);
    }
  }
  if (lane == 0) red[wv] = err;
  __syncthreads();
  if (tid == 0) out[b] = (red[0] + red[1]) + (red[2] + red[3]);
}

int launch_action_recon(const void* hidden, const float* W, const float* bias, const float* act, int B, int L, int H, int A, int act_T,
                        int ctx, int prelude, float* out, DType dt, hipStream_t st) {
  if (B <= 0) return 0;
  if (dt == BF16) hipLaunchKernelGGL(action_recon_kernel<bf16_t>, dim3((unsigned)B), dim3(256), 0, st, (const bf16_t*)hidden, W, bias, act, L, H, A, act_T, ctx, prelude, out);
  else hipLaunchKernelGGL(action_recon_kernel<float>, dim3((unsigned)B), dim3(256), 0, st, (const float*)hidden, W, bias, act, L, H, A, act_T, ctx, prelude, out);
  return (int)hipGetLastError();
}

// flag[0] += number of 4-byte words that differ between rows of a and b (row r at a + r * a_stride bytes; row_bytes % 4 == 0).
// Used to verify that a kept KV cache was built from exactly the prefix a step-wise caller presents again.
__global__ __launch_bounds__(256) void compare_rows_kernel(const unsigned* __restrict__ a, long a_stride, const unsigned* __restrict__ b,
                                                           long b_stride, long row_words, int* __restrict__ flag) {
  const int r = blockIdx.y;
  const unsigned* pa = (const unsigned*)((const char*)a + (long)r * a_stride);
  const unsigned* pb = (const unsigned*)((const char*)b + (long)r * b_stride);
  int bad = 0;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < row_words; i += (long)gridDim.x * 256) bad += pa[i] != pb[i] ? 1 : 0;
  bad = (int)wave_sum((float)bad);
  if ((threadIdx.x & 63) == 0 && bad) atomicAdd(flag, bad);
}

int launch_compare_rows(const void* a, long a_stride_bytes, const void* b, long b_stride_bytes, int rows, long row_bytes, int* flag,
                        hipStream_t st) {
  if (rows <= 0 || row_bytes <= 0) return 0;
  if (row_bytes % 4 != 0 || a_stride_bytes % 4 != 0 || b_stride_bytes % 4 != 0) return (int)hipErrorInvalidValue;
  const long words = row_bytes / 4;
  const unsigned gx = (unsigned)std::min<long>(64, (words + 255) / 256);
  hipLaunchKernelGGL(compare_rows_kernel, dim3(gx, (unsigned)rows), dim3(256), 0, st, (const unsigned*)a, a_stride_bytes,
                     (const unsigned*)b, b_stride_bytes, words, flag);
  return (int)hipGetLastError();
}

}  // namespace ivg
