// Host-side launchers of every hand-written kernel (all return hipError_t as int, 0 = ok).
#pragma once
#include "common.h"
#include "igemm.h"

namespace ivg {

// ---- norm.hip
int gn_num_chunks(int P);
// GroupNorm over [N][P][C] (stats per (n, group) over P*C/groups values), optional SiLU and position embedding
// pos[P][C] (fp32).  part_ws: N * gn_num_chunks(P) * groups double2 of scratch.
int launch_groupnorm(const void* X, void* Y, void* part_ws, const float* gamma, const float* beta, const float* pos,
                     int N, int P, int C, int groups, float eps, int silu, DType dt, hipStream_t st);
// the statistics half alone, and the (scale, shift) table [N][C] float2 the conv3x3 kernel applies to its input while staging it
int launch_groupnorm_partial(const void* X, void* part_ws, int N, int P, int C, int groups, DType dt, hipStream_t st);
int launch_gn_coef(const void* part, int nchunks, const float* gamma, const float* beta, int N, int P, int C, int groups, float eps, void* coef,
                   hipStream_t st);
// the apply half alone, with statistics produced by a conv3x3 epilogue (IgemmArgs::gn_part): double2 [N][nchunks][groups]
int launch_groupnorm_apply(const void* X, void* Y, const void* part, int nchunks, const float* gamma, const float* beta, const float* pos,
                           int N, int P, int C, int groups, float eps, int silu, DType dt, hipStream_t st);
// out = rmsnorm(x) * w  (rows of x at stride x_stride; HF semantics: normalised row rounded to T before the weight)
int launch_add_rmsnorm(void* x, long x_stride, const float* w, void* out, int M, int H, float eps, DType dt, hipStream_t st);
int launch_softmax(const float* S, void* Pm, long rows, int Lq, int Lk, int lds, int ldp, int causal, DType dt, hipStream_t st);

// ---- conv_small.hip
// First conv of the encoders: reads frames straight from the (B, T, 3, H, W) video (planar, fp32 or bf16),
// writes NHWC.  Image n of the launch is frame (n / per) * T_total + t0 + (n % per) of the clip tensor.
int launch_conv_in(const void* video, DType video_dt, const float* w /*[C0][3][3][3]*/, const float* bias, void* Y, DType dt,
                   int N, int per, int T_total, int t0, int H, int W, int C0, hipStream_t st);

// ---- vq.hip
// token row r = (g, i): g = r / tpf is a global frame, i = r % tpf;  b = g / nf, f = g % nf
//   -> element  b * stride + start + f * fstride + i  of the int64 token matrix (compressive_vq_model.py:205-215)
struct TokMap {
  int tpf, nf;
  long stride;
  int start, fstride;
};
int launch_sqnorm_rows(const float* E, float* out, int rows, int dim, hipStream_t st);
// out[map(r)] = argmin_j dist(z_r, e_j) + offset;  z fp32 [R][64];  lowest index on ties
int launch_vq_argmin(const float* z, const float* E, const float* ee, int64_t* out, const TokMap& map, int64_t offset, int R,
                     int n_e, hipStream_t st);
// Y[r][:] = E[clamp(ids[map(r)] - sub, 0, n_e - 1)][:]  (codebook fp32 -> T)
int launch_gather_rows(const int64_t* ids, const TokMap& map, const float* E, void* Y, DType dt, int R, int dim, int64_t sub,
                       int n_e, hipStream_t st);
// un-patchify: q[M*np*np][p*p*C] with feature order (ph, pw, c) -> NHWC [M][np*p][np*p][C]   (compressive_vq_model.py:247-250)
int launch_unpatchify(const void* q, void* out, DType dt, int M, int np, int C, int p, hipStream_t st);
// special tokens (scf / sdf slots) and labels (-100 over the context part)                    (compressive_vq_model.py:205-218)
int launch_finish_tokens(int64_t* ids, long stride, int64_t* labels, int B, int L, int ctx, int64_t scf, int64_t sdf, hipStream_t st);

// ---- llama_ops.hip
struct StepState {  // device-resident per-generate state (so one captured step graph can be replayed)
  int pos;          // number of tokens already in the KV cache (= position of the token being fed)
  int j;            // 1-based index of the NEW token being decided at this step
};
// x[b][:] = E[tok[b]] (+ act[b][slot][:] when add_act), T
// (x_bstride: elements between the outputs of consecutive trajectories; 0 = dense [B][L][H])
int launch_embed(const int64_t* ids, long id_stride, const void* E, void* x, DType dt, int B, int L, int H, int V, hipStream_t st,
                 long x_bstride = 0);
// RoPE on q,k of qkv[M][3H] (in place) and append k,v to the cache [B][heads][Lmax][hd]; position of row (b, l) = pos0 + l
// (pos0 from *state when state != null).  vt (optional): transposed V scratch [B][heads][hd][ldvt] for the prefill P.V GEMM.
int launch_rope_kv(void* qkv, void* kc, void* vc, void* vt, int ldvt, const float* cosT, const float* sinT, int B, int L,
                   int heads, int hd, int Lmax, const StepState* state, int pos0, DType dt, hipStream_t st);
// single-token step: RoPE of q / new k at position *pos, append k, v to the cache, then
// out[b][h*hd..] = softmax(q.K^T / sqrt(hd)) V over positions [0, *pos]
// causal prompt attention in one kernel (bf16, head_dim 64); -1 = not covered, use the score GEMM / softmax / P.V path
int launch_flash_prefill(const void* qkv, const void* kc, const void* vt, void* out, int B, int L, int Lp, int heads, int hd, int Lmax,
                         DType dt, hipStream_t st);
// tokenizer attention in one pass (bf16): cross-attention, 4 heads of 32 / 64 / 128 / 192 channels, and the single-head
// self-attention of the conditional mid blocks (512 / 768 channels; M frames attending to themselves: F = 1, B = M); -1 = not covered
int launch_xattn(const void* q, const void* Kp, const void* VpT, void* out, int M, int F, int P, int kv, int C, int nh, DType dt, hipStream_t st);
bool xattn_covers(int P, int kv, int C, int nh, DType dt);   // pure predicate (shape, dtype): usable while planning the workspace
// prof (nullable): [IVG_ATTN_PROF_SLOTS][Lmax starts | Lmax ends] wall-clock stamps (100 MHz) of the launch at each cache
// position; workgroups spread over the slots so the atomics do not serialise on one address
#define IVG_ATTN_PROF_SLOTS 32
// sh_G > 1 (shared-context rollout): chunk row b belongs to group slot (b - sh_row0) / sh_G (sh_row0 <= 0); key rows t < sh_P are read
// from cache row `slot` (where the prefill of the group's prompt wrote them), rows t >= sh_P from the trajectory's own cache row b
int launch_decode_attn(const void* qkv, void* kc, void* vc, void* out, const float* cosT, const float* sinT, int B, int heads, int hd,
                       int Lmax, const StepState* state, unsigned long long* prof, DType dt, hipStream_t st, int sh_P = 0, int sh_G = 1,
                       int sh_row0 = 0);
// 24-bit K / V cache of the x3 rollout (llama_ops.hip): per (trajectory, head) [Lmax][64] uint16 (upper halves) | [Lmax][64] uint8 (next byte);
// head_dim 64, fp32 q / output.  launch_kv24_pack: fp32 rows [0, L) of [BH][Lmax][64] K and V (the prefill's scratch) -> the planes
int launch_decode_attn24(const void* qkv, void* kc, void* vc, void* out, const float* cosT, const float* sinT, int B, int heads, int Lmax,
                         const StepState* state, unsigned long long* prof, hipStream_t st, int sh_P = 0, int sh_G = 1, int sh_row0 = 0);
long long decode_attn24_launches();   // launches over the 24-bit cache since load (test hook: which cache format an x3 engine really ran)
int launch_kv24_pack(const void* k32, const void* v32, void* kc, void* vc, int BH, int L, int Lmax, hipStream_t st);
int launch_expand_prompt_rows(const int64_t* prompts, long pstride, int64_t* ids, long ids_ld, int rows, int L, int G, int b0, hipStream_t st);
// token decision + embedding of the decided token (+ action embedding on forced sdf slots) + state advance
struct SampleArgs {
  const float* logits; int V;            // [B][V] fp32 (row stride V)
  const float* uniforms; int n_uni;      // [B][n_uni] or null (greedy)
  int top_k;
  int64_t* ids_out; long ids_stride; int L0;   // ids_out[b*ids_stride + L0 + j - 1] = token j
  int forced_period; int64_t forced_token;     // j % period == 0 -> forced token (0 = never)
  const void* E; void* x; int H;               // next input embedding x[b][:] (T)
  const void* act; int act_T; int ctx;         // act[B][act_T][H] (T) or null: added on forced slots, index slot0 + j/period + ctx - 1
  int slot0;                                   // sdf slots already inside the prompt beyond the first: (L0 - 257*ctx) / 17
  StepState* state;
  float temperature;                           // logits / temperature before the top-k filter (HF TemperatureLogitsWarper); 1.0: none
};
int launch_sample_embed(const SampleArgs& a, int B, DType dt, hipStream_t st);
int launch_state_set(StepState* state, int pos, int j, hipStream_t st);
// y[b][t][:] = W[H][A] a[b][t][:] + bias  (tiny; fp32 in, T out)
int launch_action_embed(const float* act, const float* W, const float* bias, void* out, DType dt, int BT, int A, int H,
                        hipStream_t st);
int launch_add_rows(void* x, long x_stride, const void* add, long add_stride, int B, int H, DType dt, hipStream_t st);
// r[b] = rsqrt(mean(h[b]^2) + eps) * dot(h[b], w) + bias   (reward head on the RMS-normed hidden state);  eps < 0: plain dot + bias
int launch_rowdot(const void* h, const float* w, const float* bias, float* out, int B, int H, float eps, DType dt, hipStream_t st);
// out[b] = final RMSNorm of the residual rows x[b] as HF reports them in hidden_states[-1] (normalised row rounded to T, times w)
int launch_final_hidden(const void* x, const float* w, void* out, int B, int H, float eps, DType dt, hipStream_t st);
// eval heads: shifted cross-entropy per row of a logits chunk, per-trajectory (sum, count), action reconstruction squared error
int launch_ce_rows(const float* logits, const int64_t* labels, long row0, int rows, int L, int V, float* nll, hipStream_t st);
int launch_ce_reduce(const float* nll, const int64_t* labels, int B, int L, int V, float* out, hipStream_t st);
int launch_action_recon(const void* hidden, const float* W, const float* bias, const float* act, int B, int L, int H, int A, int act_T,
                        int ctx, int prelude, float* out, DType dt, hipStream_t st);
// *flag += number of differing 32-bit words between `rows` rows of row_bytes bytes (strides in bytes)
int launch_compare_rows(const void* a, long a_stride_bytes, const void* b, long b_stride_bytes, int rows, long row_bytes, int* flag,
                        hipStream_t st);

// ---- ingest.hip
// uint8 frames [T][H][W][3] -> planar [T][3][R][R] in [0, 1]: / 255, optional centre crop, antialiased bilinear resize (ATen semantics)
int launch_ingest(const unsigned char* src, int T, int H, int W, int crop, void* dst, DType dst_dt, int R, hipStream_t st);

// ---- metrics.hip
// per-trajectory (mse, psnr, ssim) of predicted frames, mean over frames, best of the n_samples / B samples per trajectory
size_t frame_metrics_ws_bytes(int n_samples, int T, int H, int W);
int launch_frame_metrics(const void* gt, DType gt_dt, int B, int T_gt, int gt_t0, const float* pred, int n_samples, int T_pr, int pr_t0, int T,
                         int H, int W, float* rows, void* ws, hipStream_t st);

}  // namespace ivg
