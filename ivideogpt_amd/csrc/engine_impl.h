// Per-call execution context shared by tokenizer.cpp / transformer.cpp / api.cpp.
#pragma once
#include <algorithm>
#include <cmath>
#include <string>
#include <vector>

#include "engine.h"

namespace ivg {

#define IVG_TRY(x) do { int _r = (x); if (_r != 0) return _r; } while (0)

// GroupNorm statistics of an activation tensor that its PRODUCER (a conv3x3 epilogue) already reduced: [N][chunks][groups] double2
struct GnStats { void* part = nullptr; int chunks = 0; };

struct Run {
  ivg_engine* e;
  hipStream_t st;
  bool planning;  // true: only walk the allocation plan (no launches) to size the workspace
  bool x3 = false;   // the entry point running now computes its fp32 matrix products in split-bf16 arithmetic (IVG_F32X3 path)
  int kv_group = 1;  // decoder_trunk: trajectories per shared context (cross-attention K / V projected once per group; detokenize sets it)

  // ---- primitives (tokenizer.cpp)
  int conv(DType dt, const void* X, int N, int H, int W, const ConvW& c, void* Y, int stride, int ups, const void* Rres, int flags,
           int out_f32, GnStats* out_stats = nullptr /* in: .part = buffer; out: .chunks (0: the statistics were not produced) */,
           const void* in_coef = nullptr /* GroupNorm + SiLU of X applied inside the conv3x3 staging; returns -100 when not covered */);
  // conv(silu(GroupNorm(x))) with the normalisation fused into the convolution's input staging where the 3x3 kernel covers the
  // shape (the normalised tensor never reaches HBM); otherwise GroupNorm into `scratch`, then the convolution
  int norm_conv(DType dt, const void* x, int N, int H, int W, const NormW& n, float eps, const GnStats* x_stats, const ConvW& c, void* Y,
                const void* Rres, void* scratch, GnStats* out_stats);
  int gemm(DType dt, const IgemmArgs& a, double flops, double bytes);
  int linear(DType dt, const void* X, long rows, const ConvW& c, void* Y, const void* Rres, int flags, int out_f32);
  int gnorm(DType dt, const void* X, void* Y, int N, int P, int C, const NormW& n, float eps, int silu, const float* pos,
            const GnStats* stats = nullptr /* statistics of X from its producer: skips the statistics pass */);
  int resnet(DType dt, const void* x, int N, int H, int W, const ResnetW& r, void* out, const GnStats* x_stats = nullptr,
             GnStats* out_stats = nullptr);
  size_t gn_stats_bytes(int N, int H, int W, int C) const;
  int self_attention(DType dt, const void* x, int N, int P, int C, const AttnW& a, void* out);
  int xatt_project_kv(DType dt, const void* feat, int B, const XAttW& x, void* Kp, void* VpT);
  int cross_attention(DType dt, const void* z, int B, int F, const XAttW& x, const void* Kp, const void* VpT, void* out);
  int encoder_trunk(const TrunkW& w, const void* pixels, DType pix_dt, int B, int per, int T_total, int t0,
                    std::vector<Feature>* keep, const std::vector<Feature>* cond, void* latent);
  int decoder_trunk(const TrunkW& w, const void* z, int B, int per, int T_total, int t0, std::vector<Feature>* keep,
                    const std::vector<Feature>* cond, void* out_pixels, DType out_dt);
  int tokenize(const void* pixels, DType pix_dt, int B, int T, int64_t* ids, int64_t ids_stride, int64_t* labels, bool ctx_only);
  int detokenize(const int64_t* ids, int B, int F, void* out_pixels, DType out_dt, ivg_cache* cache, int cache_mode,
                 int group = 1 /* > 1: consecutive rows share their context (decoded / projected once per group) */);

  // ---- transformer (transformer.cpp)
  int prefill(const int64_t* ids, int64_t ids_stride, int B, int L, const void* act_emb, int act_T, int ctx, bool all_slots,
              float* logits_all /* [B][L][V] or null */, float* logits_last /* [B][V] or null */, void* hidden_last,
              const void* embeds = nullptr /* [B][L][H] llm dtype: used instead of the embedding of ids */,
              void* hidden_all = nullptr /* [B][L][H] llm dtype: post-final-norm hidden states (eval heads) */,
              const int64_t* labels = nullptr, float* token_nll = nullptr /* [B][L]: shifted cross-entropy per position */);
  // embeds != null: llm.generate(inputs_embeds=...) -- the prompt is given as input embeddings, only the n_new tokens are
  // returned (new_ids_out [B][n_new]); hidden_out [B][H]: post-norm hidden state of the last forward pass
  int generate(const int64_t* prompt, int64_t prompt_stride, int B, int L0, int n_new, const float* actions, int act_T, int ctx,
               const float* uniforms, int top_k, int64_t* ids_out, float* reward_out, bool reuse_kv = false,
               const void* embeds = nullptr, int64_t* new_ids_out = nullptr, void* hidden_out = nullptr,
               bool force_sdf = false /* every 17th new token is the forced sdf even without actions (generate_without_action) */,
               int group = 1 /* > 1: shared-context rollout, `prompt` holds one row per group of `group` consecutive trajectories */);

  // ---- measurement
  void prof_begin(DType dt, double flops, double bytes, int base = 0);   // base 0: igemm classes, 2: conv3x3 classes
  void prof_end(DType dt, int base = 0);
  void prof_cancel(DType dt, int base = 0);
};

size_t dtype_size(DType d);
int build_tokenizer(ivg_engine* e);
int build_transformer(ivg_engine* e);
size_t gen_buffer_bytes(const ivg_engine* e);
// device-side verification (one stream synchronisation) that the kept KV cache was built from this very prefix
int kv_prefix_matches_ids(ivg_engine* e, const int64_t* prompt, int64_t prompt_stride, int B, int L0, const float* actions, int act_T,
                          int ctx, hipStream_t st, bool* ok);
int kv_prefix_matches_embeds(ivg_engine* e, const void* embeds, int B, int L0, hipStream_t st, bool* ok);

}  // namespace ivg
