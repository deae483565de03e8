// Llama-style autoregressive transformer over video tokens on the hand-written kernels.
//
// Replaces (structure only; arithmetic is in the .hip kernels):
//   HF LlamaForCausalLM.forward / GenerationMixin._sample          
//   HeadModelWithAction.generate / .forward                        ivideogpt/transformer/action_model.py:56-121,154-205
// MI355X-first differences from the reference's op sequence (token-identical):
//   * ONE prefill + KV-cached single-token steps even in the action-conditioned mode (the reference
//     re-prefills the whole prefix for every future frame, action_model.py:101-110);
//   * a decode step is a fixed sequence of 62 kernels (sampler, 12 x [QKV GEMM with the input RMSNorm folded in, attention,
//     o-proj + residual, gate/up GEMM with the post-attention norm + SiLU(gate)*up, down-proj + residual], lm_head with the
//     final norm) whose step-dependent scalars live in device memory, captured once into a hipGraph and replayed for every
//     generated token -- no host sync, no per-step Python;
//   * decode GEMMs stream each weight matrix once; their K reduction happens inside the workgroup in a fixed order
//     (deterministic, no partials in HBM);
//   * the prompt pass uses the 256 x 256-tile GEMM (gemm256.hip), a vectorised RoPE + cache append and, in bf16, a one-pass
//     causal attention kernel; step-wise callers (MBRL) keep the KV cache across calls (ivg_generate_continue).
#include <cstring>
#include "engine_impl.h"
#include "switches.h"

namespace ivg {

#define CK(x) do { int _e = (x); if (_e != 0) return e->fail(IVG_ERR_HIP, std::string(#x) + " failed: hip error " + std::to_string(_e)); } while (0)

static size_t esz(DType d) { return d == BF16 ? 2 : 4; }
static size_t rup(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct GenBuf {  // persistent decode-step buffers (fixed addresses so the captured step graph can be replayed)
  StepState* state;
  char* x; char* qkv; char* attn; char* act; float* logits;
  int64_t* ids; float* uni; char* act_emb; int Bc, ids_ld;
  float* last_act;   // [Bc][max_frames][action_dim]: the action table of the call that built the kept KV cache
  int* flag;         // mismatch counter of the prefix verification
  // shared-context rollout (ivg_generate_shared; set per chunk by Run::generate): rows of the chunk in groups of sh_G sharing the cache
  // rows of their prompt -- see decode_attn_kernel SHARED.  sh_G = 1: off
  int sh_P = 0, sh_G = 1, sh_row0 = 0;
};

static int gen_chunk(const ivg_engine* e) { return std::min(e->cfg.max_batch, 128); }

static void gen_layout(const ivg_engine* e, GenBuf& g, char* base, size_t* total) {
  const ivg_config& c = e->cfg;
  const DType dt = e->llm_dt;
  const int Bc = gen_chunk(e), H = c.hidden_size, I = c.intermediate_size, V = c.vocab_size;
  g.Bc = Bc;
  g.ids_ld = e->Lmax;
  size_t off = 0;
  auto take = [&](size_t bytes) { char* p = base ? base + off : nullptr; off = rup(off + bytes, 256); return p; };
  g.state = (StepState*)take(256);
  g.x = take((size_t)Bc * H * esz(dt));
  g.qkv = take((size_t)Bc * 3 * H * esz(dt));
  g.attn = take((size_t)Bc * H * esz(dt));
  g.act = take((size_t)Bc * I * esz(dt));
  g.logits = (float*)take((size_t)Bc * V * 4);
  g.ids = (int64_t*)take((size_t)Bc * g.ids_ld * 8);
  g.uni = (float*)take((size_t)Bc * g.ids_ld * 4);
  g.act_emb = take((size_t)Bc * std::max(1, c.max_frames) * H * esz(dt));
  g.last_act = (float*)take((size_t)Bc * std::max(1, c.max_frames) * std::max(1, c.action_dim) * 4);
  g.flag = (int*)take(256);
  *total = off;
}

size_t gen_buffer_bytes(const ivg_engine* e) {
  GenBuf g;
  size_t t = 0;
  gen_layout(e, g, nullptr, &t);
  return t;
}

static char* kc_ptr(const ivg_engine* e, int layer, int which) {
  const size_t per = (size_t)gen_chunk(e) * e->heads * e->Lmax * e->hd * e->kv_elem_bytes();
  return e->kv + ((size_t)layer * 2 + which) * per;
}

// -------------------------------------------------------------------------------------------- prefill
static bool flash_prefill_covers(DType dt, int hd) {
  return sw().flash_prefill && dt == BF16 && hd == 64;   // IVG_FLASH_PREFILL=0: score GEMM + softmax + P.V GEMM (A/B tests)
}

int Run::prefill(const int64_t* ids, int64_t ids_stride, int B, int L, const void* act_emb, int act_T, int ctx, bool all_slots,
                 float* logits_all, float* logits_last, void* hidden_last, const void* embeds, void* hidden_all, const int64_t* labels,
                 float* token_nll) {
  const ivg_config& c = e->cfg;
  const DType dt = e->llm_dt;
  x3 = e->llm_x3;
  const int H = c.hidden_size, I = c.intermediate_size, V = c.vocab_size, heads = e->heads, hd = e->hd, Lmax = e->Lmax;
  const long M = (long)B * L;
  const int Lp = (int)rup(L, 64);
  const size_t m = e->ws.mark();
  char* x = (char*)e->ws.alloc((size_t)M * H * esz(dt));
  char* xn = (char*)e->ws.alloc((size_t)M * H * esz(dt));
  char* qkv = (char*)e->ws.alloc((size_t)M * 3 * H * esz(dt));
  char* attn = (char*)e->ws.alloc((size_t)M * H * esz(dt));
  char* act = (char*)e->ws.alloc((size_t)M * I * esz(dt));
  const bool flash = flash_prefill_covers(dt, hd);   // one-pass causal attention: no score matrix in HBM
  float* S = flash ? nullptr : (float*)e->ws.alloc((size_t)B * heads * L * Lp * 4);
  // 24-bit cache (x3 rollout): RoPE writes one layer's fp32 K / V rows here -- the score GEMM reads K as a matrix -- and
  // kv24_pack_kernel moves them into the cache's planes
  const bool kv24 = e->kv24;
  char* k32 = kv24 ? (char*)e->ws.alloc((size_t)B * heads * Lmax * hd * 4) : nullptr;
  char* v32 = kv24 ? (char*)e->ws.alloc((size_t)B * heads * Lmax * hd * 4) : nullptr;
  char* Pm = flash ? nullptr : (char*)e->ws.alloc((size_t)B * heads * L * Lp * esz(dt));
  if (!planning) {
    e->kv_len = 0; e->kv_B = 0;   // the cache rows are about to be overwritten (ivg_generate re-validates them at its end)
    if (embeds) CK((int)hipMemcpyAsync(x, embeds, (size_t)M * H * esz(dt), hipMemcpyDeviceToDevice, st));
    else CK(launch_embed(ids, ids_stride, e->embed, x, dt, B, L, H, V, st));
    if (act_emb) {  // action embedding on the sdf slot(s): slot i (position 257*ctx - 1 + 17*i) gets action i + ctx - 1
      for (int i = 0;; ++i) {
        const int pos = 257 * ctx - 1 + 17 * i;
        if (pos >= L || i + ctx - 1 >= act_T) break;
        CK(launch_add_rows(x + (size_t)pos * H * esz(dt), (long)L * H, (const char*)act_emb + (size_t)(i + ctx - 1) * H * esz(dt),
                           (long)act_T * H, B, H, dt, st));
        if (!all_slots) break;
      }
    }
  }
  for (int l = 0; l < c.num_layers; ++l) {
    const LayerW& w = e->layers[l];
    if (!planning) CK(launch_add_rmsnorm(x, H, e->ones, xn, (int)M, H, c.rms_norm_eps, dt, st));
    ConvW wq; wq.w = w.wqkv; wq.cin = H; wq.cout = 3 * H;
    IVG_TRY(linear(dt, xn, M, wq, qkv, nullptr, 0, 0));
    char* kl = kv24 ? k32 : kc_ptr(e, l, 0);
    if (!planning) {
      CK(launch_rope_kv(qkv, kl, kv24 ? v32 : kc_ptr(e, l, 1), e->vt, Lp, e->rope_cos, e->rope_sin, B, L, heads, hd, Lmax, nullptr, 0, dt, st));
      if (kv24) CK(launch_kv24_pack(k32, v32, kc_ptr(e, l, 0), kc_ptr(e, l, 1), B * heads, L, Lmax, st));
    }
    if (flash) {
      if (!planning) CK(launch_flash_prefill(qkv, kl, e->vt, attn, B, L, Lp, heads, hd, Lmax, dt, st));
    } else {
    {  // S[b][h] = Q K^T / sqrt(hd)
      IgemmArgs g;
      g.X = qkv; g.W = kl; g.Y = S;
      g.Nimg = 1; g.Hin = 1; g.Win = L; g.Cin = hd; g.ldx = 3 * H; g.Hout = 1; g.Wout = L;
      g.N = L; g.ldw = hd; g.c_pix = Lp; g.c_ch = 1; g.flags = IG_OUT_F32; g.alpha = 1.0f / sqrtf((float)hd);
      g.nb0 = B; g.nb1 = heads;
      g.sa[0] = (long)L * 3 * H; g.sa[1] = hd;
      g.sw[0] = (long)heads * Lmax * hd; g.sw[1] = (long)Lmax * hd;
      g.sy[0] = (long)heads * L * Lp; g.sy[1] = (long)L * Lp;
      IVG_TRY(gemm(dt, g, 2.0 * B * heads * (double)L * L * hd, (double)esz(dt) * 2.0 * M * H + 4.0 * B * heads * L * Lp));
    }
    if (!planning) CK(launch_softmax(S, Pm, (long)B * heads * L, L, L, Lp, Lp, 1, dt, st));
    {  // attn[b][:, h*hd..] = P V
      IgemmArgs g;
      g.X = Pm; g.W = e->vt; g.Y = attn;
      g.Nimg = 1; g.Hin = 1; g.Win = L; g.Cin = Lp; g.ldx = Lp; g.Hout = 1; g.Wout = L;
      g.N = hd; g.ldw = Lp; g.c_pix = H; g.c_ch = 1;
      g.nb0 = B; g.nb1 = heads;
      g.sa[0] = (long)heads * L * Lp; g.sa[1] = (long)L * Lp;
      g.sw[0] = (long)heads * hd * Lp; g.sw[1] = (long)hd * Lp;
      g.sy[0] = (long)L * H; g.sy[1] = hd;
      IVG_TRY(gemm(dt, g, 2.0 * B * heads * (double)L * Lp * hd, (double)esz(dt) * ((double)B * heads * L * Lp + 2.0 * M * H)));
    }
    }
    ConvW wo; wo.w = w.wo; wo.cin = H; wo.cout = H;
    IVG_TRY(linear(dt, attn, M, wo, x, x, 0, 0));  // in-place residual
    if (!planning) CK(launch_add_rmsnorm(x, H, e->ones, xn, (int)M, H, c.rms_norm_eps, dt, st));
    {  // act = silu(gate) * up   (weights packed [16 gate | 16 up] per 32 rows)
      IgemmArgs g;
      g.X = xn; g.W = w.wgu; g.Y = act;
      g.Nimg = 1; g.Hin = 1; g.Win = (int)M; g.Cin = H; g.ldx = H; g.Hout = 1; g.Wout = (int)M;
      g.N = 2 * I; g.ldw = H; g.c_pix = I; g.c_ch = 1; g.flags = IG_GLU;
      IVG_TRY(gemm(dt, g, 2.0 * M * (double)H * 2 * I, (double)esz(dt) * ((double)M * H + 2.0 * I * H + (double)M * I)));
    }
    ConvW wd; wd.w = w.wdown; wd.cin = I; wd.cout = H;
    IVG_TRY(linear(dt, act, M, wd, x, x, 0, 0));
  }
  if (hidden_all && !planning) {
    if (!e->final_norm) return e->fail(IVG_ERR_MISSING, "hidden states requested but 'llm.norm' is not in the weight table");
    CK(launch_final_hidden(x, e->final_norm, hidden_all, (int)M, H, c.rms_norm_eps, dt, st));
  }
  if (logits_all) {
    if (!planning) CK(launch_add_rmsnorm(x, H, e->ones, xn, (int)M, H, c.rms_norm_eps, dt, st));
    ConvW wl; wl.w = e->lm_head; wl.cin = H; wl.cout = V;
    IVG_TRY(linear(dt, xn, M, wl, logits_all, nullptr, 0, 1));
  }
  if (token_nll) {
    // loss without the [B][L][V] fp32 logits tensor (3.1 GB at B = 64, L = 751): lm_head over chunks of rows, each chunk reduced
    // to its per-position cross-entropy before the next one overwrites it
    const long Rc = std::min<long>(M, 4096);
    float* chunk = (float*)e->ws.alloc((size_t)Rc * V * 4);
    if (!planning && !logits_all) CK(launch_add_rmsnorm(x, H, e->ones, xn, (int)M, H, c.rms_norm_eps, dt, st));
    ConvW wl; wl.w = e->lm_head; wl.cin = H; wl.cout = V;
    for (long r0 = 0; r0 < M; r0 += Rc) {
      const long rows = std::min(Rc, M - r0);
      IVG_TRY(linear(dt, xn + (size_t)r0 * H * esz(dt), rows, wl, chunk, nullptr, 0, 1));
      if (!planning) CK(launch_ce_rows(chunk, labels, r0, (int)rows, L, V, token_nll, st));
    }
  }
  if (logits_last && !planning) {
    // last position of every sequence -> residual rows hidden_last [B][H]; final RMSNorm is fused into the lm_head GEMM
    CK((int)hipMemcpy2DAsync(hidden_last, (size_t)H * esz(dt), x + (size_t)(L - 1) * H * esz(dt), (size_t)L * H * esz(dt),
                             (size_t)H * esz(dt), B, hipMemcpyDeviceToDevice, st));
    SkinnyArgs s;
    s.X = hidden_last; s.W = e->lm_head; s.Y = logits_last; s.M = B; s.N = V; s.K = H; s.ldx = H; s.ldw = H; s.ldy = V;
    s.flags = IG_OUT_F32 | SK_NORM; s.eps = c.rms_norm_eps; s.lds_kb = e->decode_lds_kb;
    s.w_shared = e->in_flight() && sw().decode_w_shared;   // (the same policy as the steps' lm_head: one copy of the weights under several engines)
    CK(launch_skinny(s, dt, st));
  }
  e->ws.reset(m);
  return 0;
}

// -------------------------------------------------------------------------------------------- one decode step
// decide token j (sample / forced), embed it, run it through the layers against the KV cache, produce the
// logits for token j+1, advance the device-side state.
// (Splitting the rows into several concurrent chains on side streams was measured twice -- rounds 2 and 3, also on CU-masked
// streams -- without gain: every launch is bound by what one CU ingests, half-batch GEMMs cost what full-batch ones do.  Removed.)
static int step_body(ivg_engine* e, hipStream_t st, const GenBuf& g, int B, const SampleArgs& sa0, bool forward, bool skip_sample = false) {
  constexpr int b0 = 0;
  const ivg_config& c = e->cfg;
  const DType dt = e->llm_dt;
  const int H = c.hidden_size, I = c.intermediate_size, V = c.vocab_size;
  const size_t es = esz(dt);
  StepState* state = g.state;
  char* x = g.x + (size_t)b0 * H * es;
  char* qkv = g.qkv + (size_t)b0 * 3 * H * es;
  char* attn = g.attn + (size_t)b0 * H * es;
  char* act = g.act + (size_t)b0 * I * es;
  float* logits = g.logits + (size_t)b0 * V;
  const size_t kv_off = (size_t)b0 * e->heads * e->Lmax * e->hd * e->kv_elem_bytes();
  SampleArgs sa = sa0;
  sa.logits = logits;
  if (sa.uniforms) sa.uniforms += (size_t)b0 * sa.n_uni;
  sa.ids_out += (size_t)b0 * sa.ids_stride;
  sa.x = x;
  if (sa.act) sa.act = (const char*)sa.act + (size_t)b0 * sa.act_T * H * es;
  sa.state = state;
  if (!skip_sample) CK(launch_sample_embed(sa, B, dt, st));   // skip: x already holds the input row (embeds path, kept KV cache)
  if (!forward) return 0;
  // 5 launches per layer: RMSNorms are fused into the consuming GEMMs (weights pre-multiplied by the norm weight,
  // row scale computed from the activations the GEMM streams anyway), residual adds into the producing GEMMs.
  const size_t gp_ld = (size_t)IVG_GEMM_PROF_SLOTS * 2 * e->Lmax;
  auto gprof = [&](SkinnyArgs& a, int idx) {
    if (!e->gemm_prof_on || !e->gemm_prof) return;
    a.prof = e->gemm_prof + (size_t)idx * gp_ld; a.pos = (const int*)state; a.prof_ld = e->Lmax;
  };
  // The batches-in-flight profile of an engine (an LDS budget below a whole CU, set per engine -- ivg_config.decode_lds_kb -- or process-wide
  // -- IVG_DECODE_LDS_KB: it shares the GPU -- and ONE copy of the weights -- with other engines, bench.py --lanes): beside the LDS budget, (1) the weight requests of the decode GEMMs use the default cache policy
  // instead of non-temporal ones -- the other engines ask for the same lines within microseconds (+1.7 %, IVG_DECODE_W_SHARED=0 for
  // A/B) -- and (2) no launch warms the next launch's weights: with (1) the other lanes' launches already do, and the extra requests
  // only compete with three attention streams (+2.0 % on top, IVG_INFLIGHT_WARM=1 for A/B; profiles/r05_lanes_policy.txt).  One batch
  // alone keeps non-temporal weights + warm-up (rounds 2 / 3: each byte is read once per token, the warm-up hides its first touch).
  const bool in_flight = e->in_flight();
  const bool w_shared = in_flight && sw().decode_w_shared;
  const bool warm = !in_flight || sw().inflight_warm;
  // every launch also pulls the weight tiles of the NEXT launch of the chain toward the CUs that will consume them (dgemm3.hip): the
  // dependent GEMM then starts on (Infinity-)cache hits instead of a cold HBM stream
  // Conditional per shape (round 6, profiles/r06_warmup_by_kind.txt): the gate/up matrix -- half of a layer's weight bytes -- is NOT warmed:
  // requesting its 9.4 MB under o-proj lengthens o-proj by what it then saves gate/up (3.36 / 4.91 us warmed vs 2.9 / 5.35 cold), and
  // those requests were most of the class's counted traffic (2.4 x the algorithmic bytes; Infinity-Cache re-reads).  q/k/v, o-proj, down
  // and lm_head keep their warm-up (0.4 - 1.8 us per launch each).
  auto link_next = [&](SkinnyArgs& cur, const SkinnyArgs& nxt) {
    if (!warm) return;
    if ((nxt.flags & IG_GLU) && !sw().warm_gate_up) return;
    int rows = dgemm3_w_rows_per_block(nxt, dt);
    if (rows <= 0) rows = dgemm_w_rows_per_block(nxt, dt);   // the next launch runs on the second-generation kernel
    if (rows <= 0 || nxt.ldw != nxt.K) return;
    cur.next_W = nxt.W; cur.next_tile_bytes = (long)rows * nxt.K * (long)es; cur.next_tiles = nxt.N / rows;
  };
  auto layer_args = [&](int l, SkinnyArgs* g) {
    const LayerW& w = e->layers[l];
    for (int k = 0; k < 4; ++k) {
      g[k].x3 = e->llm_x3 && sw().x3; g[k].lds_kb = e->decode_lds_kb; g[k].w_shared = w_shared; g[k].kind = k;
      if (in_flight && sw().inflight_kb[k] > 0) g[k].lds_kb = sw().inflight_kb[k];
    }
    SkinnyArgs& s = g[0];
    s.X = x; s.W = w.wqkv; s.Y = qkv; s.M = B; s.N = 3 * H; s.K = H; s.ldx = H; s.ldw = H; s.ldy = 3 * H;
    s.flags = SK_NORM; s.eps = c.rms_norm_eps;
    SkinnyArgs& o = g[1];
    o.X = attn; o.W = w.wo; o.Y = x; o.M = B; o.N = H; o.K = H; o.ldx = H; o.ldw = H; o.ldy = H; o.flags = IG_RESIDUAL;
    SkinnyArgs& u = g[2];
    u.X = x; u.W = w.wgu; u.Y = act; u.M = B; u.N = 2 * I; u.K = H; u.ldx = H; u.ldw = H; u.ldy = I;
    u.flags = IG_GLU | SK_NORM; u.eps = c.rms_norm_eps;
    SkinnyArgs& d = g[3];
    d.X = act; d.W = w.wdown; d.Y = x; d.M = B; d.N = H; d.K = I; d.ldx = I; d.ldw = I; d.ldy = H; d.flags = IG_RESIDUAL;
  };
  SkinnyArgs lm;
  lm.X = x; lm.W = e->lm_head; lm.Y = logits; lm.M = B; lm.N = V; lm.K = H; lm.ldx = H; lm.ldw = H; lm.ldy = V;
  lm.flags = IG_OUT_F32 | SK_NORM; lm.eps = c.rms_norm_eps; lm.lds_kb = e->decode_lds_kb; lm.w_shared = w_shared; lm.kind = 4;
  if (in_flight && sw().inflight_kb[4] > 0) lm.lds_kb = sw().inflight_kb[4];
  lm.bump = (int*)state;  // pos += 1, j += 1 once the last reader of this chain's state (its last attention) is done
  SkinnyArgs cur[4], nxt[4];
  layer_args(0, cur);
  for (int l = 0; l < c.num_layers; ++l) {
    const bool last = l + 1 == c.num_layers;
    if (!last) layer_args(l + 1, nxt);
    link_next(cur[0], cur[1]); link_next(cur[1], cur[2]); link_next(cur[2], cur[3]); link_next(cur[3], last ? lm : nxt[0]);
    gprof(cur[0], 4 * l + 0);
    CK(launch_skinny(cur[0], dt, st));
    unsigned long long* aprof = e->attn_prof_on ? e->attn_prof + (size_t)l * IVG_ATTN_PROF_SLOTS * 2 * e->Lmax : nullptr;
    if (e->kv24)
      CK(launch_decode_attn24(qkv, kc_ptr(e, l, 0) + kv_off, kc_ptr(e, l, 1) + kv_off, attn, e->rope_cos, e->rope_sin, B, e->heads, e->Lmax, state,
                              aprof, st, g.sh_P, g.sh_G, g.sh_row0));
    else
      CK(launch_decode_attn(qkv, kc_ptr(e, l, 0) + kv_off, kc_ptr(e, l, 1) + kv_off, attn, e->rope_cos, e->rope_sin, B, e->heads, e->hd,
                            e->Lmax, state, aprof, dt, st, g.sh_P, g.sh_G, g.sh_row0));
    gprof(cur[1], 4 * l + 1);
    CK(launch_skinny(cur[1], dt, st));
    gprof(cur[2], 4 * l + 2);
    CK(launch_skinny(cur[2], dt, st));
    gprof(cur[3], 4 * l + 3);
    CK(launch_skinny(cur[3], dt, st));
    if (!last) for (int k = 0; k < 4; ++k) cur[k] = nxt[k];
  }
  {   // the first GEMM of the next step's chain: q/k/v of layer 0
    SkinnyArgs first[4];
    layer_args(0, first);
    link_next(lm, first[0]);
  }
  gprof(lm, 4 * c.num_layers);
  CK(launch_skinny(lm, dt, st));
  return 0;
}

// group > 1 (ivg_generate_shared): `prompt` holds one row per GROUP of `group` consecutive trajectories (B = groups x group rows of
// actions / uniforms / ids_out).  Per chunk: the prompts of the groups the chunk's rows belong to are prefilled ONCE each -- positions
// [0, L0 - 1), into cache rows 0 .. n_groups-1 -- and every trajectory then feeds the prompt's last token itself (step j = 0, exactly the
// kept-cache entry of the step-wise callers: that slot carries the row's own action), appending from position L0 - 1 on in its own
// cache row; the decode attention reads key rows < L0 - 1 from the group's row (decode_attn_kernel SHARED).
int Run::generate(const int64_t* prompt, int64_t prompt_stride, int B, int L0, int n_new, const float* actions, int act_T, int ctx,
                  const float* uniforms, int top_k, int64_t* ids_out, float* reward_out, bool reuse_kv, const void* embeds,
                  int64_t* new_ids_out, void* hidden_out, bool force_sdf, int group) {
  const ivg_config& c = e->cfg;
  const DType dt = e->llm_dt;
  const int H = c.hidden_size, V = c.vocab_size;
  GenBuf g;
  size_t tot = 0;
  gen_layout(e, g, e->gen_buf, &tot);
  const long Ltot = (long)L0 + n_new;
  const bool shared = group > 1;
  if (planning) {
    if (shared) return prefill(nullptr, 0, std::min((std::min(B, g.Bc) + group - 1) / group + 1, std::min(B, g.Bc)), L0 - 1, nullptr, 0, ctx, false, nullptr, nullptr, nullptr);
    return reuse_kv ? 0 : prefill(nullptr, 0, std::min(B, g.Bc), L0, nullptr, 0, ctx, false, nullptr, nullptr, nullptr);
  }
  e->kv_len = 0; e->kv_B = 0;   // set again once every launch of this call is queued
  const size_t es = esz(dt);
  if (embeds && B > g.Bc) return e->fail(IVG_ERR_CAPACITY, "generate (inputs_embeds): batch exceeds the KV-cache chunk");
  if (embeds && !e->emb_snap) {   // one-time: copy of the inputs the cache is built from (verifies later "same prefix" claims)
    if (hipMalloc((void**)&e->emb_snap, (size_t)g.Bc * e->Lmax * H * es) != hipSuccess) return e->fail(IVG_ERR_HIP, "hipMalloc of the embeddings snapshot failed");
  }
  e->snap_valid = false; e->ids_valid = false;
  for (int b0 = 0; b0 < B; b0 += g.Bc) {
    const int Bc = std::min(g.Bc, B - b0);
    if (e->gemm_prof_on && e->gemm_prof) {
      CK((int)hipMemsetAsync(e->gemm_prof, 0, (size_t)(4 * c.num_layers + 1) * IVG_GEMM_PROF_SLOTS * 2 * e->Lmax * 8, st));
      e->gemm_prof_B = Bc;
    }
    if (e->attn_prof_on) {  // fresh launch windows for this call: every stamp slot back to 0 (= not stamped)
      CK((int)hipMemsetAsync(e->attn_prof, 0, (size_t)c.num_layers * IVG_ATTN_PROF_SLOTS * 2 * e->Lmax * 8, st));
      e->attn_prof_B = Bc;
    }
    if (shared) {   // every trajectory starts from a copy of its group's prompt; groups g_lo .. g_hi have rows in this chunk
      const int g_lo = b0 / group, g_hi = (b0 + Bc - 1) / group;
      g.sh_P = L0 - 1; g.sh_G = group; g.sh_row0 = g_lo * group - b0;
      // (attending the prompt rows once per 16 trajectories on the matrix cores -- tools/ubench/prefix_attn_mfma.hip -- shortens the
      // attention launch but needs a launch of its own per layer: every call measured slower, profiles/r06_shared_prefix_mfma_ab.txt)
      CK(launch_expand_prompt_rows(prompt, prompt_stride, g.ids, g.ids_ld, Bc, L0, group, b0, st));
      IVG_TRY(prefill(prompt + (long)g_lo * prompt_stride, prompt_stride, g_hi - g_lo + 1, L0 - 1, nullptr, 0, ctx, false, nullptr, nullptr, nullptr));
    } else if (!embeds)
      CK((int)hipMemcpy2DAsync(g.ids, (size_t)g.ids_ld * 8, prompt + (long)b0 * prompt_stride, (size_t)prompt_stride * 8, (size_t)L0 * 8, Bc,
                               hipMemcpyDeviceToDevice, st));
    if (uniforms)
      CK((int)hipMemcpy2DAsync(g.uni, (size_t)g.ids_ld * 4, uniforms + (long)b0 * n_new, (size_t)n_new * 4, (size_t)n_new * 4, Bc,
                               hipMemcpyDeviceToDevice, st));
    if (actions) {
      CK(launch_action_embed(actions + (long)b0 * act_T * c.action_dim, e->act_w, e->act_b, g.act_emb, dt, Bc * act_T, c.action_dim, H, st));
      if (B <= g.Bc) CK((int)hipMemcpyAsync(g.last_act, actions, (size_t)B * act_T * c.action_dim * 4, hipMemcpyDeviceToDevice, st));
    }
    if (!reuse_kv && !shared) IVG_TRY(prefill(g.ids, g.ids_ld, Bc, L0, actions ? g.act_emb : nullptr, act_T, ctx, true, nullptr, g.logits, g.x, embeds));
    if (embeds) {   // keep what the cache is (being) built from: the whole prompt after a prefill, its last row on the kept-cache path
      const int p0 = reuse_kv ? L0 - 1 : 0;
      CK((int)hipMemcpy2DAsync(e->emb_snap + (size_t)p0 * H * es, (size_t)e->Lmax * H * es, (const char*)embeds + (size_t)p0 * H * es,
                               (size_t)L0 * H * es, (size_t)(L0 - p0) * H * es, Bc, hipMemcpyDeviceToDevice, st));
      if (reuse_kv)   // the input row of the first forward pass comes straight from the caller
        CK((int)hipMemcpy2DAsync(g.x, (size_t)H * es, (const char*)embeds + (size_t)(L0 - 1) * H * es, (size_t)L0 * H * es, (size_t)H * es, Bc,
                                 hipMemcpyDeviceToDevice, st));
    }
    // reuse_kv: the cache already holds positions [0, L0 - 1); the step counter starts at j = 0, whose "decision" is the
    // forced sdf the prompt ends with (0 % 17 == 0): the sampler re-embeds it with the new action and the forward pass of
    // that step appends position L0 - 1 and yields the logits of new token 1 -- exactly what the prefill would have left
    const bool feed_last = reuse_kv || shared;   // the cache holds [0, L0 - 1): step j = 0 feeds the prompt's last token
    CK(launch_state_set(g.state, feed_last ? L0 - 1 : L0, feed_last ? 0 : 1, st));
    SampleArgs sa{};
    sa.logits = g.logits; sa.V = V;
    sa.uniforms = uniforms ? g.uni : nullptr; sa.n_uni = g.ids_ld;
    sa.top_k = top_k;
    sa.ids_out = g.ids; sa.ids_stride = g.ids_ld; sa.L0 = L0;
    sa.forced_period = (actions || force_sdf) ? 17 : 0; sa.forced_token = V - 1;   // (no action embedding is added without actions: sa.act == null)
    sa.E = e->embed; sa.x = g.x; sa.H = H;
    sa.act = actions ? g.act_emb : nullptr; sa.act_T = act_T; sa.ctx = ctx;
    sa.slot0 = actions ? (L0 - 257 * ctx) / 17 : 0;  // a prompt that already holds t generated frames (MBRL step-wise rollout)
    sa.state = g.state;
    sa.temperature = e->temperature;
    // step 1 eagerly (also performs every kernel's one-time attribute setup), then replay a captured step graph
    // (everything a captured step bakes in: the temperature by its BIT PATTERN -- to_string keeps six decimals --, this engine's LDS
    // budget and the generation of the switch table, whose kernel-selection switches a replayed graph would otherwise keep ignoring)
    uint32_t t_bits; memcpy(&t_bits, &e->temperature, 4);
    const std::string key = std::to_string(Bc) + ":" + std::to_string(t_bits) + ":" + std::to_string(e->decode_lds_kb) + ":" + std::to_string(switches_generation()) +
                            ":" + (uniforms ? "s" : "g") + ":" + std::to_string(top_k) + ":" +
                            std::to_string(sa.forced_period) + ":" + std::to_string(ctx) + ":" + std::to_string(act_T) + ":" +
                            std::to_string(L0) + (e->attn_prof_on ? ":p" : "") + (e->gemm_prof_on ? ":q" : "") +   // (the same step graph serves both entry modes)
                            (shared ? ":sh" + std::to_string(group) + ":" + std::to_string(g.sh_row0) : "");
    // reward head: reads the residual stream left by the LAST forward pass, i.e. before the final decide-only step
    // overwrites it with the embedding of the last token (mbrl/video_predictor.py:311-313: hidden state of the last step)
    auto reward = [&]() -> int {
      if (hidden_out) {   // hidden_states[-1][-1] of HF generate: the last forward pass, after the final norm
        if (!e->final_norm) return e->fail(IVG_ERR_MISSING, "generate: hidden state requested but 'llm.norm' is not in the weight table");
        CK(launch_final_hidden(g.x, e->final_norm, (char*)hidden_out + (size_t)b0 * H * es, Bc, H, c.rms_norm_eps, dt, st));
      }
      if (!reward_out) return 0;
      if (!e->rew_w) return e->fail(IVG_ERR_MISSING, "generate: reward requested but reward_linear is not loaded");
      CK(launch_rowdot(g.x, e->rew_w, e->rew_b, reward_out + b0, Bc, H, c.rms_norm_eps, dt, st));
      return 0;
    };
    int j = 1;
    if (feed_last) IVG_TRY(step_body(e, st, g, Bc, sa, true, embeds != nullptr));   // j = 0: feed the prompt's last token
    if (n_new == 1) IVG_TRY(reward());
    if (n_new >= 1) { IVG_TRY(step_body(e, st, g, Bc, sa, j < n_new)); ++j; }
    // the step sequence is position-independent (all step-dependent scalars live in StepState): it is captured once as a graph of
    // ONE step and once as a graph of `multi` consecutive steps -- the long rollouts replay the multi-step graph (a graph launch
    // costs the host ~10-16 us and leaves a bubble on the device; 8 steps per launch amortise it), the tail the single-step one
    auto get_graph = [&](int n_steps, hipGraphExec_t* out) -> int {
      *out = nullptr;
      if (!(e->use_graph && st != nullptr)) return 0;
      if (e->graphs_gen != switches_generation()) {   // the switch table changed: no captured step of the old table can be replayed again
        if (!e->graphs.empty()) {
          CK((int)hipStreamSynchronize(st));
          for (auto& kv : e->graphs) (void)hipGraphExecDestroy(kv.second);
          e->graphs.clear();
        }
        e->graphs_gen = switches_generation();
      }
      const std::string k = key + (n_steps > 1 ? ":x" + std::to_string(n_steps) : "");
      auto it = e->graphs.find(k);
      if (it != e->graphs.end()) { *out = it->second; return 0; }
      hipGraph_t graph = nullptr;
      hipGraphExec_t ex = nullptr;
      if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        int rc = 0;
        for (int i = 0; i < n_steps && rc == 0; ++i) rc = step_body(e, st, g, Bc, sa, true);
        const hipError_t ce = hipStreamEndCapture(st, &graph);
        if (rc == 0 && ce == hipSuccess && graph && hipGraphInstantiate(&ex, graph, nullptr, nullptr, 0) == hipSuccess) {
          if (e->graphs.size() >= 64) {   // bound the cache (a server fed ever new prompt lengths): drop all, recapture on demand
            CK((int)hipStreamSynchronize(st));
            for (auto& kv : e->graphs) (void)hipGraphExecDestroy(kv.second);
            e->graphs.clear();
          }
          e->graphs[k] = ex;
          *out = ex;
        } else {
          (void)hipGetLastError();
        }
        if (graph) (void)hipGraphDestroy(graph);
      } else {
        (void)hipGetLastError();
      }
      return 0;
    };
    constexpr int multi = 8;
    hipGraphExec_t exec = nullptr, exec_multi = nullptr;
    if (j < n_new) IVG_TRY(get_graph(1, &exec));
    if (exec && multi > 1 && n_new - j >= 2 * multi) IVG_TRY(get_graph(multi, &exec_multi));
    while (j < n_new) {
      if (exec_multi && n_new - j >= multi) { CK((int)hipGraphLaunch(exec_multi, st)); j += multi; }
      else if (exec) { CK((int)hipGraphLaunch(exec, st)); ++j; }
      else { IVG_TRY(step_body(e, st, g, Bc, sa, true)); ++j; }
    }
    if (j == n_new && n_new > 1) {
      IVG_TRY(reward());
      IVG_TRY(step_body(e, st, g, Bc, sa, false));  // decide the last token (no forward)
    }
    if (embeds) {
      CK((int)hipMemcpy2DAsync(new_ids_out + (long)b0 * n_new, (size_t)n_new * 8, g.ids + L0, (size_t)g.ids_ld * 8, (size_t)n_new * 8, Bc,
                               hipMemcpyDeviceToDevice, st));
      if (n_new > 1)   // inputs of the positions the steps appended: the embeddings of the fed new tokens
        CK(launch_embed(g.ids + L0, g.ids_ld, e->embed, e->emb_snap + (size_t)L0 * H * es, dt, Bc, n_new - 1, H, V, st, (long)e->Lmax * H));
    } else {
      CK((int)hipMemcpy2DAsync(ids_out + (long)b0 * Ltot, (size_t)Ltot * 8, g.ids, (size_t)g.ids_ld * 8, (size_t)Ltot * 8, Bc,
                               hipMemcpyDeviceToDevice, st));
    }
  }
  if (B <= g.Bc && !shared) {   // the last new token is decided but never fed (a shared-context cache is not a per-trajectory cache: never kept)
    e->kv_len = L0 + n_new - 1; e->kv_B = B;
    e->snap_valid = embeds != nullptr; e->ids_valid = embeds == nullptr;
    e->last_act_T = actions ? act_T : 0;
    e->last_ctx = ctx;
  }
  return 0;
}

// ---------------------------------------------------------------------------------------- kept-cache verification
static int read_flag(ivg_engine* e, const GenBuf& g, hipStream_t st, bool* ok) {
  if (!e->h_flag && hipHostMalloc((void**)&e->h_flag, sizeof(int), hipHostMallocDefault) != hipSuccess) return e->fail(IVG_ERR_HIP, "hipHostMalloc failed");
  CK((int)hipMemcpyAsync(e->h_flag, g.flag, sizeof(int), hipMemcpyDeviceToHost, st));
  CK((int)hipStreamSynchronize(st));
  *ok = *e->h_flag == 0;
  return 0;
}

int kv_prefix_matches_ids(ivg_engine* e, const int64_t* prompt, int64_t prompt_stride, int B, int L0, const float* actions, int act_T,
                          int ctx, hipStream_t st, bool* ok) {
  *ok = false;
  GenBuf g;
  size_t tot = 0;
  gen_layout(e, g, e->gen_buf, &tot);
  if (!e->ids_valid || e->kv_B != B || e->kv_len != L0 - 1 || B > g.Bc) return 0;
  // the cache must have been built the same way: with / without actions, the same context length, the same action-table shape;
  // whatever cannot be compared row for row is a mismatch, never a silent "ok"
  if ((actions != nullptr) != (e->last_act_T > 0)) return 0;
  if (ctx != e->last_ctx) return 0;
  if (actions && (e->last_act_T != act_T)) return 0;
  CK((int)hipMemsetAsync(g.flag, 0, sizeof(int), st));
  CK(launch_compare_rows(prompt, prompt_stride * 8, g.ids, (long)g.ids_ld * 8, B, (long)(L0 - 1) * 8, g.flag, st));
  if (actions) {   // the action rows already baked into the cached sdf slots: slot i (position 257*ctx - 1 + 17*i < kv_len) used row i + ctx - 1
    const int A = e->cfg.action_dim;
    const int slots = std::max(0, (e->kv_len - (257 * ctx - 1) + 16) / 17);
    if (slots > 0 && ctx - 1 + slots > act_T) return 0;   // the cached slots used action rows the presented table does not have
    if (slots > 0)
      CK(launch_compare_rows(actions + (long)(ctx - 1) * A, (long)act_T * A * 4, g.last_act + (long)(ctx - 1) * A, (long)act_T * A * 4, B,
                             (long)slots * A * 4, g.flag, st));
  }
  return read_flag(e, g, st, ok);
}

int kv_prefix_matches_embeds(ivg_engine* e, const void* embeds, int B, int L0, hipStream_t st, bool* ok) {
  *ok = false;
  GenBuf g;
  size_t tot = 0;
  gen_layout(e, g, e->gen_buf, &tot);
  if (!e->snap_valid || !e->emb_snap || e->kv_B != B || e->kv_len != L0 - 1 || B > g.Bc) return 0;
  const size_t es = esz(e->llm_dt);
  const long H = e->cfg.hidden_size;
  CK((int)hipMemsetAsync(g.flag, 0, sizeof(int), st));
  CK(launch_compare_rows(embeds, (long)L0 * H * es, e->emb_snap, (long)e->Lmax * H * es, B, (long)(L0 - 1) * H * es, g.flag, st));
  return read_flag(e, g, st, ok);
}

}  // namespace ivg
