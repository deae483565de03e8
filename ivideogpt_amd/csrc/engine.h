// Engine: owns the workspace arena, KV cache and the layer graphs of the tokenizer and transformer.
#pragma once
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/ivg.h"
#include "ops.h"

namespace ivg {

struct Arena {
  char* base = nullptr;
  size_t cap = 0, off = 0, high = 0;
  bool planning = true;  // planning pass: no memory, only the high-water mark
  bool overflow = false;
  void* alloc(size_t bytes) {
    off = (off + 255) & ~(size_t)255;
    void* p = planning ? (void*)(uintptr_t)(0x1000 + off) : (void*)(base + off);
    off += bytes;
    if (off > high) high = off;
    if (!planning && off > cap) { overflow = true; return base; }  // never hand out memory past the arena (callers plan first)
    return p;
  }
  size_t mark() const { return off; }
  void reset(size_t m) { off = m; }
};

struct ConvW { const void* w = nullptr; const float* b = nullptr; int cin = 0, cout = 0, k = 1;
               const void* w3 = nullptr; /* fp32 3x3 convs of the "x3" decode mode: weights pre-split into bf16 (hi, lo) pairs */
               const void* wsub = nullptr; const void* wsub3 = nullptr; /* upsampler convs: pre-summed sub-pixel phase weights (+ x3 split) */ };
struct NormW { const float* g = nullptr; const float* b = nullptr; };
struct ResnetW { NormW n1, n2; ConvW c1, c2, sc; bool has_sc = false; int cin = 0, cout = 0; };
struct AttnW { NormW gn; ConvW q, k, v, o; };
struct XAttW { NormW kvn, qn; const float* kv_pos = nullptr; const float* q_pos = nullptr; int kv_rows = 0;
               ConvW q, k, v, o; int C = 0, side = 0; };
struct TrunkW {  // encoder or decoder
  const float* conv_in_raw_w = nullptr;  // encoders: fp32 [C0][3][3][3]
  ConvW conv_in;                          // decoders: packed (latent -> C)
  std::vector<std::vector<ResnetW>> blocks;
  std::vector<ConvW> resample;            // down/up-sampler conv of each level (cin = 0: none)
  ResnetW mid0, mid1;
  bool has_attn = false;
  AttnW attn;
  NormW norm_out;
  ConvW conv_out;
  std::vector<XAttW> xatt;
};
struct LayerW { const void* wqkv; const void* wo; const void* wgu; const void* wdown; };  // RMSNorm weights are folded into wqkv / wgu

struct Feature { void* p = nullptr; int side = 0, C = 0; };

struct ProfSlot { hipEvent_t a, b; double flops, bytes; };
struct ProfClass { bool enabled = false; std::vector<ProfSlot> used; std::vector<ProfSlot> pool; };

}  // namespace ivg

struct ivg_cache {
  int B = 0;
  float* ctx_pixels = nullptr;             // [B][ctx][3][H][W] (sized for float32; holds pix_dt elements)
  int pix_dt = 0;                           // element type of the kept context pixels (the output type of the call that filled the cache)
  std::vector<void*> feat;                  // un-repeated per-trajectory context decoder features (NHWC)
  bool filled = false;
  bool clamped = false;                     // the kept context pixels were written with the output clamp on (ivg_set_output_clamp at fill time)
};

struct ivg_engine {
  ivg_config cfg;
  int device = 0;
  std::string err;
  std::unordered_map<std::string, ivg_tensor> wmap;
  ivg::Arena ws;
  int ctx = 1;  // current context length (set_context_length)
  bool clamp_out = false;   // detokenize writes clamp(frames, 0, 1) (conv_out epilogue) instead of the raw decoder output (ivg_set_output_clamp)
  ivg::DType enc_dt, dec_dt, llm_dt;   // element types in HBM
  bool dec_x3 = false, llm_x3 = false;  // IVG_F32X3: fp32 tensors, split-bf16 matrix arithmetic on that path
  bool kv24 = false;         // x3 rollout, head_dim 64: the K / V cache keeps 24 of the 32 bits in two planes (llama_ops.hip: decode_attn24_kernel)
  size_t kv_elem_bytes() const { return kv24 ? 3 : (llm_dt == ivg::BF16 ? 2 : 4); }
  // tokenizer
  ivg::TrunkW enc, cenc, dec, cdec;
  ivg::ConvW quant_conv, post_quant_conv, quant_linear, post_quant_linear;
  const float* cb_c = nullptr; const float* cb_d = nullptr;
  float* ee_c = nullptr; float* ee_d = nullptr;
  // transformer
  std::vector<ivg::LayerW> layers;
  const void* embed = nullptr; const void* lm_head = nullptr;  // final norm weight folded into lm_head
  float* ones = nullptr;     // [hidden] of 1.0f: the prefill's stand-alone RMSNorm has no weight left to apply
  const float* rope_cos = nullptr; const float* rope_sin = nullptr;
  const float* act_w = nullptr; const float* act_b = nullptr; const float* rew_w = nullptr; const float* rew_b = nullptr;
  int heads = 0, hd = 0, Lmax = 0;
  char* kv = nullptr;        // [layers][2][Bmax][heads][Lmax][hd]  (kv24: [layers][2][Bmax][heads]{[Lmax][hd] u16 | [Lmax][hd] u8})
  char* vt = nullptr;        // [Bmax][heads][hd][Lmax] transposed V scratch for the prefill
  char* gen_buf = nullptr;   // persistent decode-step buffers (fixed addresses -> graph replay)
  size_t gen_bytes = 0;
  std::unordered_map<std::string, hipGraphExec_t> graphs;
  bool use_graph = false;     // IVG_GRAPH=1 at ivg_create (switches.h)
  unsigned graphs_gen = 0;    // switches_generation() the captured graphs belong to (a changed table drops them all)
  int decode_lds_kb = 0;      // LDS budget of this engine's decode GEMMs (ivg_config.decode_lds_kb / ivg_set_decode_lds_kb; 0: process default)
  // the budget in force (this engine's, else the process default IVG_DECODE_LDS_KB) and what it implies: below a whole CU's 160 KiB the
  // engine runs the BATCHES-IN-FLIGHT profile (shared-weight cache policy, no warm-up of the next launch) -- the same whichever of
  // the two routes set the budget, and an explicit 160 is the one-batch profile like 0 (advice, round 5)
  int effective_lds_kb() const;
  bool in_flight() const { return effective_lds_kb() < 160; }
  float temperature = 1.0f;   // sampling temperature of the rollout (ivg_set_temperature; HF TemperatureLogitsWarper semantics)
  ivg::ProfClass prof[IVG_K_COUNT];
  unsigned long long* attn_prof = nullptr;  // [layers][IVG_ATTN_PROF_SLOTS][2][Lmax] wall-clock stamps of the decode attention
  bool attn_prof_on = false;                // ivg_profile_enable(IVG_K_DECODE_ATTN): part of the step-graph key
  unsigned long long* gemm_prof = nullptr;  // [layers * 4 + 1][IVG_GEMM_PROF_SLOTS][2][Lmax] stamps of the decode-step GEMMs (allocated on first use)
  bool gemm_prof_on = false;                // ivg_profile_enable(IVG_K_DECODE_GEMM): part of the step-graph key
  int gemm_prof_B = 0;
  double gemm_kind_ms[5] = {0, 0, 0, 0, 0};   // last ivg_profile_read(IVG_K_DECODE_GEMM) by kind: q/k/v, o-proj, gate/up, down, lm_head
  long long gemm_kind_n[5] = {0, 0, 0, 0, 0};
  int kv_len = 0, kv_B = 0;                 // the KV cache holds positions [0, kv_len) of kv_B trajectories (last generate call)
  // what that cache was built from, kept so that a step-wise caller's "same prefix" claim can be VERIFIED on the device:
  //   token path: the ids are in the persistent id buffer (gen_buf), the action table of the call in last_act;
  //   embeds path (ivg_generate_embeds): the input embeddings fed so far in emb_snap
  const float* final_norm = nullptr;        // model.norm.weight (fp32), for the hidden state handed to the caller
  const float* rew_w_raw = nullptr;         // reward_linear.weight as stored (applies to the post-norm hidden state)
  const float* ar_w = nullptr; const float* ar_b = nullptr;   // action_recon_linear (optional, eval loss term only)
  char* emb_snap = nullptr;                 // [Bc][Lmax][H] llm dtype, allocated on the first embeds call
  bool snap_valid = false;                  // emb_snap holds the inputs of positions [0, kv_len)
  bool ids_valid = false;                   // gen_buf ids hold the tokens of positions [0, kv_len)
  int last_act_T = 0;                       // rows per trajectory of last_act (0: the cache was built without actions)
  int last_ctx = 0;                         // context length of the call that built the kept cache (action slot positions depend on it)
  int* h_flag = nullptr;                    // pinned host word for the verification result
  int attn_prof_B = 0;
  double attn_fit_fixed_us = 0, attn_fit_gbps = 0;   // line fit of the last ivg_profile_read(IVG_K_DECODE_ATTN)

  int fail(int code, const std::string& msg) { err = msg; return code; }
};
