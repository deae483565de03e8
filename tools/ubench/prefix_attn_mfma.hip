// RECORD, not a build target (round 6).  Shared-prefix attention of a decode step on the matrix cores: built, op-tested against fp64
// (tests at commit time: 5 shapes x {every row reads the group's copy, this kernel + merge} green on the MI355X), measured, NOT kept.
//
// Design: ivg_generate_shared keeps a group's prompt K / V rows once; decode_attn_kernel SHARED lets every trajectory read them from
// the group's cache row (the product path).  This kernel instead attends the prompt rows [0, P) ONCE per 16 trajectories of a group
// with v_mfma_f32_16x16x32_bf16 (rows = M dimension of QK^T / PV) and hands (o[64], m, l) per (row, head) to decode_attn_kernel, which
// then covers only the row's own tail [P, pos] and merges.
//
// Measured (profiles/r06_shared_prefix_mfma_ab.txt, VP2-shaped call, bf16, small transformer; decode-attention launch window / whole call):
//   200 candidates: plain 34.1 us / 497 ms | rows read the group's copy 20.8 us / 348 ms | this kernel + merge 13.1 us / 365 ms
//    64 candidates: plain 21.1 us / 159 ms |                            13.4 us / 120 ms |                      8.3 us / 134 ms
//    16 candidates: plain 11.5 us /  97 ms |                             9.4 us /  91 ms |                      5.7 us / 107 ms
// The attention launch does get shorter (-5 .. -8 us), but the extra launch per layer (its dependent-launch boundary + a 4-tile
// latency chain per wave) costs more than that: every call is SLOWER than with the indirection alone.  A decode step at these batch
// sizes is bound by launch latency, not by the L2 reads the indirection leaves.  Kept here as the record of that measurement.
//
// To rebuild: paste prefix_attn_kernel / launch_prefix_attn into csrc/llama_ops.hip, give decode_attn_kernel the sh_part merge
// (pos_off = sh_P; kb / vb offset; mx = max(mx, m_p); sum += l_p * exp(m_p - mx); a += o_p[tid] * exp(m_p - mx)).
// Shared-prefix attention of a decode step on the matrix cores (round 6; bf16, head_dim 64): the G trajectories of a group attend to
// the SAME prompt rows [0, P) -- the group's rows are the M dimension of QK^T and PV, so the prefix is read ONCE per 16 trajectories
// instead of once per trajectory (VP2: 200 candidates over one context; train_gpt.generate_multiple_times: t samples per clip).
//   grid (row blocks of 16, heads, group slots of the chunk); 4 waves split the prefix's 32-key tiles, each with its own online
//   softmax state; combined in LDS in wave order.  Output: per (trajectory, head) the UN-normalised weighted value sum o[64], the
//   running maximum m and the sum l of exp(s - m) -- 80 floats -- which decode_attn_kernel (SHARED, sh_part) merges with the
//   trajectory's own rows [P, pos].
//   S[key][row] = K_tile . Q^T by v_mfma_f32_16x16x32_bf16 (lane: 4 keys of one row), softmax statistics per row with two
//   cross-lane maxima per tile, P as the B operand straight from the score registers (the MFMA's K index is permuted the same way
//   in the V operand: slot (lg, e) = key lg*4 + e, resp. 16 + lg*4 + e - 4), V^T tiles transposed through a per-wave LDS region.
// RoPE of q as decode_attn_kernel applies it (rotate_half, result rounded to bf16).
__global__ __launch_bounds__(256) void prefix_attn_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ kc, const bf16_t* __restrict__ vc,
                                                          float* __restrict__ part, const float* __restrict__ cosT, const float* __restrict__ sinT,
                                                          int heads, int Lmax, const StepState* __restrict__ state, int P, int G, int row0, int B) {
  constexpr int HD = 64, VT_PITCH = 80;                 // bytes per d-row of the transposed V tile (32 keys x 2 B + pad)
  __shared__ __attribute__((aligned(16))) unsigned char s_vt[4][HD * VT_PITCH];   // per wave; reused for the combine
  __shared__ float s_m[4][16], s_l[4][16];
  const int rb = blockIdx.x, h = blockIdx.y, slot = blockIdx.z;
  const int lo = max(0, row0 + slot * G), hi = min(B, row0 + (slot + 1) * G);
  const int r_first = lo + rb * 16;
  if (r_first >= hi) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lg = lane >> 4;
  const int pos = state->pos, H = heads * HD;
  // ---- Q fragments of row r_first + lr: dims lg*8 .. +8 (MFMA k-step 0) and 32 + lg*8 .. +8 (k-step 1) = the rotate_half pairs
  const int row = min(r_first + lr, hi - 1);
  const bf16_t* qrow = qkv + (long)row * 3 * H + h * HD;
  const bf16x8 qa = *(const bf16x8*)(qrow + lg * 8), qb = *(const bf16x8*)(qrow + 32 + lg * 8);
  bf16x8 q0, q1;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float c = cosT[(long)pos * 32 + lg * 8 + j], sn = sinT[(long)pos * 32 + lg * 8 + j];
    q0[j] = (bf16_t)((float)qa[j] * c - (float)qb[j] * sn);
    q1[j] = (bf16_t)((float)qb[j] * c + (float)qa[j] * sn);
  }
  const bf16_t* kbase = kc + ((long)slot * heads + h) * Lmax * HD;   // the group's cache row: where the prefill of its prompt wrote
  const bf16_t* vbase = vc + ((long)slot * heads + h) * Lmax * HD;
  float m_run = -INFINITY, l_run = 0.f;
  f32x4 O[4];
#pragma unroll
  for (int db = 0; db < 4; ++db) O[db] = f32x4{0.f, 0.f, 0.f, 0.f};
  unsigned char* vt = s_vt[wave];
  for (int t0 = wave * 32; t0 < P; t0 += 128) {
    // ---- scores of 32 keys: two 16-key blocks x two k-steps
    f32x4 S[2];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      const int key = min(t0 + kb * 16 + lr, P - 1);
      const bf16x8 k0 = *(const bf16x8*)(kbase + (long)key * HD + lg * 8), k1 = *(const bf16x8*)(kbase + (long)key * HD + 32 + lg * 8);
      S[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, q0, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      S[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, q1, S[kb], 0, 0, 0);
    }
    // ---- V tile -> LDS, transposed and key-permuted: Vt[d][slot position lg' * 8 + e] with key = (e < 4 ? lg'*4 + e : 16 + lg'*4 + e - 4)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = lane + 64 * i, key = c >> 3, dch = c & 7;
      const bf16x8 vv = *(const bf16x8*)(vbase + (long)min(t0 + key, P - 1) * HD + dch * 8);
      const int kk = key & 15, pp = (kk >> 2) * 8 + (kk & 3) + 4 * (key >> 4);
#pragma unroll
      for (int j = 0; j < 8; ++j) *(bf16_t*)(vt + (dch * 8 + j) * VT_PITCH + pp * 2) = vv[j];
    }
    // ---- online softmax: the tile's row maximum over this lane's 8 keys and the 4 lanes that share the row
    float sv[2][4];
    float tmax = -INFINITY;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = t0 + kb * 16 + lg * 4 + r;
        sv[kb][r] = key < P ? S[kb][r] * 0.125f : -INFINITY;
        tmax = fmaxf(tmax, sv[kb][r]);
      }
    tmax = fmaxf(tmax, __shfl_xor(tmax, 16, 64));
    tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
    const float m_new = fmaxf(m_run, tmax);            // (every tile holds at least one valid key: finite)
    const float alpha = expf(m_run - m_new);           // exp(-inf) = 0 on the first tile
    bf16x8 pf;
    float psum = 0.f;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const bf16_t pb = (bf16_t)expf(sv[kb][r] - m_new);   // rounded once: the sum below uses the value the matrix core multiplies
        pf[kb * 4 + r] = pb;
        psum += (float)pb;
      }
    l_run = l_run * alpha + psum;
    m_run = m_new;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's transposed V tile is in LDS
#pragma unroll
    for (int db = 0; db < 4; ++db) {
#pragma unroll
      for (int r = 0; r < 4; ++r) O[db][r] *= alpha;
      const bf16x8 vf = *(const bf16x8*)(vt + (db * 16 + lr) * VT_PITCH + lg * 16);
      O[db] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, O[db], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the fragment reads are done before the next tile overwrites the region
  }
  // ---- combine the four waves (fixed order): per-row sums first (the 4 lanes of a row), then m / l / O through LDS
  l_run += __shfl_xor(l_run, 16, 64);
  l_run += __shfl_xor(l_run, 32, 64);
  __syncthreads();                                       // every wave is done with its V region
  float* wo = (float*)s_vt[wave];                        // [64 d][16 rows] of this wave
  if (lg == 0) { s_m[wave][lr] = m_run; s_l[wave][lr] = l_run; }
#pragma unroll
  for (int db = 0; db < 4; ++db)
#pragma unroll
    for (int r = 0; r < 4; ++r) wo[(db * 16 + lg * 4 + r) * 16 + lr] = O[db][r];
  __syncthreads();
  for (int idx = tid; idx < 16 * 64; idx += 256) {
    const int r = idx & 15, d = idx >> 4;
    if (r_first + r >= hi) continue;
    const float M = fmaxf(fmaxf(s_m[0][r], s_m[1][r]), fmaxf(s_m[2][r], s_m[3][r]));
    float o = 0.f, l = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float wgt = s_l[w][r] > 0.f ? expf(s_m[w][r] - M) : 0.f;
      o = fmaf(((const float*)s_vt[w])[d * 16 + r], wgt, o);
      l = fmaf(s_l[w][r], wgt, l);
    }
    float* dst = part + ((long)(r_first + r) * heads + h) * 80;
    dst[d] = o;
    if (d == 0) { dst[64] = M; dst[65] = l; }
  }
}

int launch_prefix_attn(const void* qkv, const void* kc, const void* vc, float* part, const float* cosT, const float* sinT, int B, int heads, int hd,
                       int Lmax, const StepState* state, int P, int G, int row0, int n_slots, DType dt, hipStream_t st) {
  if (dt != BF16 || hd != 64 || P < 1 || G < 1 || row0 > 0 || n_slots < 1) return -1;
  const int rows = std::min(G, B);
  dim3 g((unsigned)((rows + 15) / 16), (unsigned)heads, (unsigned)n_slots);
  hipLaunchKernelGGL(prefix_attn_kernel, g, dim3(256), 0, st, (const bf16_t*)qkv, (const bf16_t*)kc, (const bf16_t*)vc, part, cosT, sinT, heads, Lmax, state, P, G,
                     row0, B);
  return (int)hipGetLastError();
}

