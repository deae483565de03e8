// C ABI of libivg (include/ivg.h): engine construction from a named weight table, workspace planning,
// the tokenize / generate / detokenize entry points and the op-level hooks used by the parity tests.
#include <cstdlib>
#include <cstring>

#include <cstdio>
#include "engine_impl.h"
#include "switches.h"

using namespace ivg;

static thread_local std::string g_create_err;

namespace ivg {

size_t dtype_size(DType d) { return d == BF16 ? 2 : 4; }

struct Lookup {
  ivg_engine* e;
  bool ok = true;
  const ivg_tensor* find(const std::string& name, int dtype, int64_t numel) {
    auto it = e->wmap.find(name);
    if (it == e->wmap.end()) { if (ok) e->err = "missing weight tensor '" + name + "'"; ok = false; return nullptr; }
    const ivg_tensor& t = it->second;
    int64_t n = 1;
    for (int i = 0; i < t.ndim; ++i) n *= t.shape[i];
    if (t.dtype != dtype || n != numel) {
      if (ok) e->err = "weight tensor '" + name + "' has dtype " + std::to_string(t.dtype) + " / " + std::to_string(n) +
                       " elements, expected dtype " + std::to_string(dtype) + " / " + std::to_string(numel);
      ok = false;
      return nullptr;
    }
    return &t;
  }
  const void* data(const std::string& name, int dtype, int64_t numel) {
    const ivg_tensor* t = find(name, dtype, numel);
    return t ? t->data : nullptr;
  }
  const float* f32(const std::string& name, int64_t numel) { return (const float*)data(name, IVG_F32, numel); }
  ConvW conv(const std::string& n, int cin, int cout, int k, DType dt) {
    ConvW c; c.cin = cin; c.cout = cout; c.k = k;
    c.w = data(n + ".weight", (int)dt, (int64_t)cout * k * k * cin);
    c.b = f32(n + ".bias", cout);
    // optional: the same matrix pre-split into bf16 (hi, lo) pairs for the split-bf16 3x3 kernel (packing.py: pack_x3)
    if (dt == F32 && k == 3 && e->wmap.count(n + ".weight.x3")) c.w3 = data(n + ".weight.x3", IVG_BF16, (int64_t)2 * cout * k * k * cin);
    // optional (upsampler convs): the sub-pixel phase weights [4][cout][4 * cin] (packing.py: pack_subpixel), and their x3 split
    if (k == 3 && e->wmap.count(n + ".weight.subpix")) c.wsub = data(n + ".weight.subpix", (int)dt, (int64_t)16 * cout * cin);
    if (dt == F32 && k == 3 && e->wmap.count(n + ".weight.subpix.x3")) c.wsub3 = data(n + ".weight.subpix.x3", IVG_BF16, (int64_t)32 * cout * cin);
    return c;
  }
  NormW norm(const std::string& n, int C) { NormW r; r.g = f32(n + ".weight", C); r.b = f32(n + ".bias", C); return r; }
  ResnetW resnet(const std::string& n, int cin, int cout, DType dt) {
    ResnetW r; r.cin = cin; r.cout = cout;
    r.n1 = norm(n + ".norm1", cin); r.c1 = conv(n + ".conv1", cin, cout, 3, dt);
    r.n2 = norm(n + ".norm2", cout); r.c2 = conv(n + ".conv2", cout, cout, 3, dt);
    r.has_sc = cin != cout;
    if (r.has_sc) r.sc = conv(n + ".conv_shortcut", cin, cout, 1, dt);
    return r;
  }
  AttnW attn(const std::string& n, int C, DType dt) {
    AttnW a; a.gn = norm(n + ".group_norm", C);
    a.q = conv(n + ".to_q", C, C, 1, dt); a.k = conv(n + ".to_k", C, C, 1, dt);
    a.v = conv(n + ".to_v", C, C, 1, dt); a.o = conv(n + ".to_out.0", C, C, 1, dt);
    return a;
  }
  XAttW xatt(const std::string& n, int C, int side, int ctx0, DType dt) {
    XAttW x; x.C = C; x.side = side; x.kv_rows = ctx0 * side * side;
    x.kvn = norm(n + ".kv_norm", C); x.qn = norm(n + ".q_norm", C);
    x.kv_pos = f32(n + ".kv_pos_emb", (int64_t)x.kv_rows * C);
    x.q_pos = f32(n + ".q_pos_emb", (int64_t)side * side * C);
    const char* w = (const char*)data(n + ".att.in_proj_weight", (int)dt, (int64_t)3 * C * C);
    const float* b = f32(n + ".att.in_proj_bias", 3 * C);
    auto part = [&](int i) { ConvW c; c.cin = C; c.cout = C; c.k = 1; c.w = w ? w + (size_t)i * C * C * dtype_size(dt) : nullptr; c.b = b ? b + i * C : nullptr; return c; };
    x.q = part(0); x.k = part(1); x.v = part(2);
    x.o = conv(n + ".att.out_proj", C, C, 1, dt);
    return x;
  }
};

int build_tokenizer(ivg_engine* e) {
  const ivg_config& c = e->cfg;
  const int nl = c.n_levels, lpb = c.layers_per_block, lat = c.latent_channels, dim = c.vq_embed_dim, p = c.patch_size;
  const int* ch = c.block_out_channels;
  Lookup L{e};
  auto encoder = [&](const std::string& n, bool attn, bool cond, TrunkW& t) {
    const DType dt = e->enc_dt;
    t.conv_in_raw_w = L.f32(n + ".conv_in.weight", (int64_t)ch[0] * 27);
    t.conv_in.b = L.f32(n + ".conv_in.bias", ch[0]);
    int prev = ch[0], res = c.resolution;
    for (int i = 0; i < nl; ++i) {
      std::vector<ResnetW> blk;
      for (int j = 0; j < lpb; ++j) blk.push_back(L.resnet(n + ".down_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), j == 0 ? prev : ch[i], ch[i], dt));
      t.blocks.push_back(blk);
      ConvW ds;
      if (i != nl - 1) { ds = L.conv(n + ".down_blocks." + std::to_string(i) + ".downsamplers.0.conv", ch[i], ch[i], 3, dt); res /= 2; }
      t.resample.push_back(ds);
      if (cond && res <= c.max_att_resolution)
        t.xatt.push_back(L.xatt(n + ".cross_att_blocks." + std::to_string(t.xatt.size()), ch[i], res, c.context_length, dt));
      prev = ch[i];
    }
    t.mid0 = L.resnet(n + ".mid_block.resnets.0", ch[nl - 1], ch[nl - 1], dt);
    t.mid1 = L.resnet(n + ".mid_block.resnets.1", ch[nl - 1], ch[nl - 1], dt);
    t.has_attn = attn;
    if (attn) t.attn = L.attn(n + ".mid_block.attentions.0", ch[nl - 1], dt);
    t.norm_out = L.norm(n + ".conv_norm_out", ch[nl - 1]);
    t.conv_out = L.conv(n + ".conv_out", ch[nl - 1], lat, 3, dt);
  };
  auto decoder = [&](const std::string& n, bool attn, bool cond, TrunkW& t) {
    const DType dt = e->dec_dt;
    const int top = ch[nl - 1];
    t.conv_in = L.conv(n + ".conv_in", lat, top, 3, dt);
    t.mid0 = L.resnet(n + ".mid_block.resnets.0", top, top, dt);
    t.mid1 = L.resnet(n + ".mid_block.resnets.1", top, top, dt);
    t.has_attn = attn;
    if (attn) t.attn = L.attn(n + ".mid_block.attentions.0", top, dt);
    int prev = top, res = 16;
    if (cond) t.xatt.push_back(L.xatt(n + ".cross_att_blocks.0", top, 16, c.context_length, dt));
    for (int i = 0; i < nl; ++i) {
      const int co = ch[nl - 1 - i];
      std::vector<ResnetW> blk;
      for (int j = 0; j < lpb + 1; ++j) blk.push_back(L.resnet(n + ".up_blocks." + std::to_string(i) + ".resnets." + std::to_string(j), j == 0 ? prev : co, co, dt));
      t.blocks.push_back(blk);
      ConvW us;
      if (i != nl - 1) { us = L.conv(n + ".up_blocks." + std::to_string(i) + ".upsamplers.0.conv", co, co, 3, dt); res *= 2; }
      t.resample.push_back(us);
      if (cond && res <= c.max_att_resolution)
        t.xatt.push_back(L.xatt(n + ".cross_att_blocks." + std::to_string(t.xatt.size()), co, res, c.context_length, dt));
      prev = co;
    }
    t.norm_out = L.norm(n + ".conv_norm_out", ch[0]);
    t.conv_out = L.conv(n + ".conv_out", ch[0], 3, 3, dt);
  };
  encoder("encoder", c.mid_block_add_attention != 0, false, e->enc);
  encoder("cond_encoder", true, true, e->cenc);
  decoder("decoder", c.mid_block_add_attention != 0, false, e->dec);
  decoder("cond_decoder", true, true, e->cdec);
  e->quant_conv = L.conv("quant_conv", lat, dim, 1, e->enc_dt);
  e->quant_linear = L.conv("quant_linear", lat, dim, p, e->enc_dt);  // [dim][p*p*lat] with (ph, pw, c) feature order
  e->post_quant_conv = L.conv("post_quant_conv", dim, lat, 1, e->dec_dt);
  e->post_quant_linear = L.conv("post_quant_linear", dim, lat * p * p, 1, e->dec_dt);
  e->cb_c = L.f32("quantize.embedding.weight", (int64_t)c.num_vq_embeddings * dim);
  e->cb_d = L.f32("dynamics_quantize.embedding.weight", (int64_t)c.num_dyn_embeddings * dim);
  return L.ok ? 0 : IVG_ERR_MISSING;
}

int build_transformer(ivg_engine* e) {
  const ivg_config& c = e->cfg;
  const DType dt = e->llm_dt;
  const int H = c.hidden_size, I = c.intermediate_size, V = c.vocab_size;
  Lookup L{e};
  e->heads = c.num_heads; e->hd = H / c.num_heads;
  e->Lmax = c.max_seq > 0 ? c.max_seq : c.max_position_embeddings;
  e->layers.clear();
  for (int l = 0; l < c.num_layers; ++l) {
    const std::string b = "llm.layers." + std::to_string(l) + ".";
    LayerW w;
    w.wqkv = L.data(b + "wqkv", (int)dt, (int64_t)3 * H * H);
    w.wo = L.data(b + "wo", (int)dt, (int64_t)H * H);
    w.wgu = L.data(b + "wgu", (int)dt, (int64_t)2 * I * H);
    w.wdown = L.data(b + "wdown", (int)dt, (int64_t)H * I);
    e->layers.push_back(w);
  }
  e->embed = L.data("llm.embed", (int)dt, (int64_t)V * H);
  e->lm_head = L.data("llm.lm_head", (int)dt, (int64_t)V * H);
  e->rope_cos = L.f32("llm.rope_cos", (int64_t)c.max_position_embeddings * (e->hd / 2));
  e->rope_sin = L.f32("llm.rope_sin", (int64_t)c.max_position_embeddings * (e->hd / 2));
  if (c.action_dim > 0) {
    e->act_w = L.f32("llm.action_linear.weight", (int64_t)H * c.action_dim);
    e->act_b = L.f32("llm.action_linear.bias", H);
  }
  if (c.reward_head) {
    e->rew_w = L.f32("llm.reward_linear.weight", H);
    e->rew_b = L.f32("llm.reward_linear.bias", 1);
    if (e->wmap.count("llm.reward_linear.raw")) e->rew_w_raw = L.f32("llm.reward_linear.raw", H);
  }
  if (e->wmap.count("llm.norm")) e->final_norm = L.f32("llm.norm", H);   // optional: only the hidden-state outputs need it
  if (c.action_dim > 0 && e->wmap.count("llm.action_recon_linear.weight")) {
    e->ar_w = L.f32("llm.action_recon_linear.weight", (int64_t)c.action_dim * H);
    e->ar_b = L.f32("llm.action_recon_linear.bias", c.action_dim);
  }
  return L.ok ? 0 : IVG_ERR_MISSING;
}

// ---- measurement hooks
void Run::prof_begin(DType dt, double flops, double bytes, int base) {
  ProfClass& pc = e->prof[base + (dt == BF16 ? 0 : 1)];
  if (!pc.enabled) return;
  ProfSlot s;
  if (!pc.pool.empty()) { s = pc.pool.back(); pc.pool.pop_back(); }
  else { if (hipEventCreate(&s.a) != hipSuccess || hipEventCreate(&s.b) != hipSuccess) return; }
  s.flops = flops; s.bytes = bytes;
  (void)hipEventRecord(s.a, st);
  pc.used.push_back(s);
}
void Run::prof_end(DType dt, int base) {
  ProfClass& pc = e->prof[base + (dt == BF16 ? 0 : 1)];
  if (!pc.enabled || pc.used.empty()) return;
  (void)hipEventRecord(pc.used.back().b, st);
}
void Run::prof_cancel(DType dt, int base) {
  ProfClass& pc = e->prof[base + (dt == BF16 ? 0 : 1)];
  if (!pc.enabled || pc.used.empty()) return;
  pc.pool.push_back(pc.used.back());
  pc.used.pop_back();
}

}  // namespace ivg

#define API_CK(x) do { hipError_t _e = (x); if (_e != hipSuccess) { e->err = std::string(#x) + ": " + hipGetErrorString(_e); return IVG_ERR_HIP; } } while (0)

// Every entry point first walks its own allocation plan (host arithmetic only) and grows the arena if this call
// needs more than any earlier one; then runs for real.  f(Run&) must be the same call in both passes.
template <typename F>
static int plan_then_run(ivg_engine* e, hipStream_t st, F&& f) {
  e->ws.planning = true; e->ws.off = 0; e->ws.high = 0; e->ws.overflow = false;
  Run p{e, nullptr, true};
  int rc = f(p);
  const size_t need = e->ws.high + (1 << 20);
  e->ws.planning = false; e->ws.off = 0;
  if (rc) return rc;
  if (need > e->ws.cap) {
    API_CK(hipDeviceSynchronize());
    if (e->ws.base) API_CK(hipFree(e->ws.base));
    e->ws.base = nullptr; e->ws.cap = 0;
    API_CK(hipMalloc((void**)&e->ws.base, need));
    e->ws.cap = need;
  }
  Run r{e, st, false};
  rc = f(r);
  if (rc == 0 && e->ws.overflow) return e->fail(IVG_ERR_CAPACITY, "internal: workspace plan was smaller than the run");
  return rc;
}

static int plan_and_allocate(ivg_engine* e) {
  const ivg_config& c = e->cfg;
  e->ws.planning = true; e->ws.off = 0; e->ws.high = 0;
  Run r{e, nullptr, true};
  const int B = c.max_batch, T = c.max_frames;
  if (c.n_levels > 0) {
    const int saved = e->ctx;
    for (int ctx = 1; ctx <= c.context_length; ++ctx) {  // any context length set_context_length may select later
      e->ctx = ctx;
      if (T > ctx) {
        int rc = r.tokenize(nullptr, F32, B, T, nullptr, 257 * ctx - 1 + 17 * (T - ctx), nullptr, false); if (rc) return rc;
        rc = r.detokenize(nullptr, B, T - ctx, nullptr, F32, nullptr, 0); if (rc) return rc;
      }
    }
    e->ctx = saved;
  }
  if (c.num_layers > 0) {
    const int Lpre = std::min(e->Lmax - 1, 514);  // typical prompt (2 context frames); larger calls grow the arena on demand
    int rc = r.generate(nullptr, 0, B, Lpre, 1, nullptr, 0, 1, nullptr, 0, nullptr, nullptr); if (rc) return rc;
  }
  e->ws.cap = e->ws.high + (1 << 20);
  API_CK(hipMalloc((void**)&e->ws.base, e->ws.cap));
  e->ws.planning = false; e->ws.off = 0;
  return 0;
}

extern "C" {

const char* ivg_version(void) { return "libivg 0.1 (gfx950)"; }

const char* ivg_last_error(const ivg_engine* e) { return e ? e->err.c_str() : g_create_err.c_str(); }

void ivg_destroy(ivg_engine* e) {
  if (!e) return;
  (void)hipSetDevice(e->device);
  (void)hipDeviceSynchronize();
  for (auto& kv : e->graphs) (void)hipGraphExecDestroy(kv.second);
  for (int k = 0; k < IVG_K_COUNT; ++k) {
    for (auto& s : e->prof[k].used) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
    for (auto& s : e->prof[k].pool) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
  }
  if (e->ws.base) (void)hipFree(e->ws.base);
  if (e->ee_c) (void)hipFree(e->ee_c);
  if (e->ee_d) (void)hipFree(e->ee_d);
  if (e->kv) (void)hipFree(e->kv);
  if (e->vt) (void)hipFree(e->vt);
  if (e->gen_buf) (void)hipFree(e->gen_buf);
  if (e->ones) (void)hipFree(e->ones);
  if (e->attn_prof) (void)hipFree(e->attn_prof);
  if (e->emb_snap) (void)hipFree(e->emb_snap);
  if (e->gemm_prof) (void)hipFree(e->gemm_prof);
  if (e->h_flag) (void)hipHostFree(e->h_flag);
  delete e;
}

int ivg_create(const ivg_config* cfg, const ivg_tensor* weights, int n_weights, int device, ivg_engine** out) {
  if (!cfg || !out) { g_create_err = "ivg_create: null argument"; return IVG_ERR_INVALID; }
  ivg_engine* e = new ivg_engine();
  e->cfg = *cfg; e->device = device;
  auto bail = [&](int code) { g_create_err = e->err; ivg_destroy(e); return code; };
  if (hipSetDevice(device) != hipSuccess) { e->err = "hipSetDevice failed (no MI355X visible?)"; return bail(IVG_ERR_HIP); }
  reload_switches();   // the one place the environment is read (switches.h)
  // Decode steps are launched eagerly by default: on ROCm 7.2 a replayed hipGraph leaves ~1 us MORE between two dependent kernel
  // nodes than the same kernels launched one by one on the stream (config-2 rollout: 166 ms replayed, 149 ms eager; the host
  // issues a launch in ~4 us against ~10 us of device time per kernel, so it stays ahead).  IVG_GRAPH=1 captures the step into
  // a hipGraph (8 steps per launch) for callers that need the host thread back early.
  e->use_graph = sw().graph != 0;
  if (cfg->decode_lds_kb != 0 && (cfg->decode_lds_kb < 16 || cfg->decode_lds_kb > 160)) { e->err = "ivg_create: decode_lds_kb must be 0 (process default) or 16 .. 160"; return bail(IVG_ERR_INVALID); }
  e->decode_lds_kb = cfg->decode_lds_kb;
  if (cfg->encode_dtype == IVG_F32X3) { e->err = "ivg_create: encode_dtype cannot be IVG_F32X3 (bit-exact VQ indices need the exact fp32 chain)"; return bail(IVG_ERR_INVALID); }
  // IVG_F32X3: fp32 tensors, split-bf16 matrix arithmetic (conv3x3.hip / igemm.hip X3 instances)
  e->dec_x3 = cfg->decode_dtype == IVG_F32X3; e->llm_x3 = cfg->llm_dtype == IVG_F32X3;
  e->enc_dt = (DType)cfg->encode_dtype; e->dec_dt = e->dec_x3 ? F32 : (DType)cfg->decode_dtype; e->llm_dt = e->llm_x3 ? F32 : (DType)cfg->llm_dtype;
  e->ctx = cfg->context_length > 0 ? cfg->context_length : 1;
  for (int i = 0; i < n_weights; ++i) e->wmap[weights[i].name] = weights[i];
  if (cfg->max_batch <= 0 || cfg->max_frames <= 0) { e->err = "ivg_create: max_batch / max_frames must be positive"; return bail(IVG_ERR_INVALID); }
  if (cfg->n_levels > 0) {
    const ivg_config& c = *cfg;
    if (c.n_levels > 8 || c.vq_embed_dim != 64 || c.patch_size != 4 || (c.resolution >> (c.n_levels - 1)) != 16 || c.latent_channels % 32 != 0) {
      e->err = "ivg_create: unsupported tokenizer geometry (needs vq_embed_dim 64, patch 4, 16x16 latent grid)"; return bail(IVG_ERR_INVALID);
    }
    int rc = build_tokenizer(e); if (rc) return bail(rc);
    if (hipMalloc((void**)&e->ee_c, (size_t)c.num_vq_embeddings * 4) != hipSuccess || hipMalloc((void**)&e->ee_d, (size_t)c.num_dyn_embeddings * 4) != hipSuccess) {
      e->err = "hipMalloc failed"; return bail(IVG_ERR_HIP);
    }
    if (launch_sqnorm_rows(e->cb_c, e->ee_c, c.num_vq_embeddings, c.vq_embed_dim, nullptr) || launch_sqnorm_rows(e->cb_d, e->ee_d, c.num_dyn_embeddings, c.vq_embed_dim, nullptr)) {
      e->err = "codebook norm kernel failed to launch (is this a gfx950 device?)"; return bail(IVG_ERR_HIP);
    }
  }
  if (cfg->num_layers > 0) {
    int rc = build_transformer(e); if (rc) return bail(rc);
    const int kb = std::min(cfg->max_batch, 128);
    e->kv24 = e->llm_x3 && e->hd == 64 && sw().x3 && sw().kv24;   // (IVG_X3=0: an x3 engine IS the fp32 engine)
    const size_t kvb = (size_t)cfg->num_layers * 2 * kb * e->heads * e->Lmax * e->hd * e->kv_elem_bytes();
    const size_t vtb = (size_t)kb * e->heads * e->hd * ((e->Lmax + 63) / 64 * 64) * dtype_size(e->llm_dt);
    e->gen_bytes = gen_buffer_bytes(e);
    if (hipMalloc((void**)&e->kv, kvb) != hipSuccess || hipMalloc((void**)&e->vt, vtb) != hipSuccess || hipMalloc((void**)&e->gen_buf, e->gen_bytes) != hipSuccess) {
      e->err = "hipMalloc of the KV cache failed"; return bail(IVG_ERR_HIP);
    }
    {
      std::vector<float> one((size_t)cfg->hidden_size, 1.0f);
      if (hipMalloc((void**)&e->ones, one.size() * 4) != hipSuccess || hipMemcpy(e->ones, one.data(), one.size() * 4, hipMemcpyHostToDevice) != hipSuccess) {
        e->err = "hipMalloc failed"; return bail(IVG_ERR_HIP);
      }
    }
    if (hipMalloc((void**)&e->attn_prof, (size_t)cfg->num_layers * IVG_ATTN_PROF_SLOTS * 2 * e->Lmax * 8) != hipSuccess) { e->err = "hipMalloc failed"; return bail(IVG_ERR_HIP); }
    (void)hipMemset(e->attn_prof, 0, (size_t)cfg->num_layers * IVG_ATTN_PROF_SLOTS * 2 * e->Lmax * 8);
    (void)hipMemset(e->vt, 0, vtb);
    (void)hipMemset(e->gen_buf, 0, e->gen_bytes);
  }
  int rc = plan_and_allocate(e);
  if (rc) return bail(rc);
  if (hipDeviceSynchronize() != hipSuccess) { e->err = "device error during engine construction"; return bail(IVG_ERR_HIP); }
  *out = e;
  return IVG_OK;
}

int ivg_set_context_length(ivg_engine* e, int k) {
  if (!e) return IVG_ERR_INVALID;
  if (k < 1 || k > e->cfg.context_length) return e->fail(IVG_ERR_INVALID, "set_context_length: k must be in [1, pretrained context_length]");
  e->ctx = k;
  return IVG_OK;
}

static int check_tok(ivg_engine* e, int B, int T, const char* who) {
  if (e->cfg.n_levels <= 0) return e->fail(IVG_ERR_INVALID, std::string(who) + ": engine was created without a tokenizer");
  if (B <= 0 || B > e->cfg.max_batch || T > e->cfg.max_frames)
    return e->fail(IVG_ERR_CAPACITY, std::string(who) + ": batch " + std::to_string(B) + " x " + std::to_string(T) + " frames exceeds the capacity the engine was created with (" +
                   std::to_string(e->cfg.max_batch) + " x " + std::to_string(e->cfg.max_frames) + ")");
  return 0;
}

int ivg_tokenize(ivg_engine* e, const void* pixels, int pixel_dtype, int B, int T, int64_t* ids_out, int64_t* labels_out, ivg_stream stream) {
  if (!e) return IVG_ERR_INVALID;
  IVG_TRY(check_tok(e, B, T, "tokenize"));
  if (T < e->ctx + 1) return e->fail(IVG_ERR_INVALID, "tokenize: needs at least one future frame (T >= context_length + 1)");
  return plan_then_run(e, (hipStream_t)stream, [&](Run& r) {
    return r.tokenize(pixels, (DType)pixel_dtype, B, T, ids_out, 257L * e->ctx - 1 + 17L * (T - e->ctx), labels_out, false); });
}

int ivg_encode_context(ivg_engine* e, const void* pixels, int pixel_dtype, int B, int T, int64_t* ids_out, int64_t ids_stride, ivg_stream stream) {
  if (!e) return IVG_ERR_INVALID;
  IVG_TRY(check_tok(e, B, std::min(T, e->cfg.max_frames), "encode_context"));
  if (T < e->ctx || ids_stride < 257L * e->ctx) return e->fail(IVG_ERR_INVALID, "encode_context: T < context_length or ids_stride < 257*ctx");
  return plan_then_run(e, (hipStream_t)stream, [&](Run& r) {
    return r.tokenize(pixels, (DType)pixel_dtype, B, T, ids_out, ids_stride, nullptr, true); });
}

void ivg_reload_switches(void) { reload_switches(); }

int ivg_set_temperature(ivg_engine* e, float temperature) {
  if (!e) return IVG_ERR_INVALID;
  if (!(temperature > 0.0f) || !std::isfinite(temperature)) return e->fail(IVG_ERR_INVALID, "temperature must be a strictly positive float");   // HF raises the same
  e->temperature = temperature;
  return IVG_OK;
}

int ivg_engine::effective_lds_kb() const { return decode_lds_kb > 0 ? decode_lds_kb : ivg::sw().decode_lds_kb; }

int ivg_set_decode_lds_kb(ivg_engine* e, int kb) {
  if (!e) return IVG_ERR_INVALID;
  if (kb != 0 && (kb < 16 || kb > 160)) return e->fail(IVG_ERR_INVALID, "decode_lds_kb must be 0 (process default) or 16 .. 160");
  e->decode_lds_kb = kb;
  return IVG_OK;
}

int ivg_set_output_clamp(ivg_engine* e, int on) {
  if (!e) return IVG_ERR_INVALID;
  e->clamp_out = on != 0;
  return IVG_OK;
}

int ivg_detokenize_to(ivg_engine* e, const int64_t* ids, int B, int F, void* pixels_out, int pixel_dtype, ivg_cache* cache, int cache_mode,
                      ivg_stream stream) {
  if (!e) return IVG_ERR_INVALID;
  IVG_TRY(check_tok(e, B, e->ctx + F, "detokenize"));
  if (F < 0) return e->fail(IVG_ERR_INVALID, "detokenize: token count does not match 257*ctx - 1 + 17*F");
  if (pixel_dtype != IVG_F32 && pixel_dtype != IVG_BF16) return e->fail(IVG_ERR_INVALID, "detokenize: pixels are float32 or bfloat16");
  return plan_then_run(e, (hipStream_t)stream, [&](Run& r) { return r.detokenize(ids, B, F, pixels_out, (DType)pixel_dtype, cache, cache ? cache_mode : 0); });
}

int ivg_detokenize_shared(ivg_engine* e, const int64_t* ids, int n_groups, int group_size, int F, void* pixels_out, int pixel_dtype, ivg_stream stream) {
  if (!e) return IVG_ERR_INVALID;
  if (n_groups <= 0 || group_size <= 0) return e->fail(IVG_ERR_INVALID, "detokenize_shared: n_groups and group_size must be positive");
  const int B = n_groups * group_size;
  IVG_TRY(check_tok(e, B, e->ctx + F, "detokenize"));
  if (F < 0) return e->fail(IVG_ERR_INVALID, "detokenize: token count does not match 257*ctx - 1 + 17*F");
  if (pixel_dtype != IVG_F32 && pixel_dtype != IVG_BF16) return e->fail(IVG_ERR_INVALID, "detokenize: pixels are float32 or bfloat16");
  return plan_then_run(e, (hipStream_t)stream, [&](Run& r) { return r.detokenize(ids, B, F, pixels_out, (DType)pixel_dtype, nullptr, 0, group_size); });
}

int ivg_detokenize(ivg_engine* e, const int64_t* ids, int B, int F, float* pixels_out, ivg_cache* cache, int cache_mode, ivg_stream stream) {
  return ivg_detokenize_to(e, ids, B, F, pixels_out, IVG_F32, cache, cache_mode, stream);
}

int ivg_cache_create(ivg_engine* e, int B, ivg_cache** out) {
  if (!e || !out || e->cfg.n_levels <= 0) return IVG_ERR_INVALID;
  const ivg_config& c = e->cfg;
  if (B <= 0 || B > c.max_batch) return e->fail(IVG_ERR_CAPACITY, "cache_create: batch " + std::to_string(B) + " exceeds the engine's max_batch");
  ivg_cache* k = new ivg_cache();
  k->B = B;
  const int ctx = c.context_length, res = c.resolution, nl = c.n_levels;
  bool ok = hipMalloc((void**)&k->ctx_pixels, (size_t)B * ctx * 3 * res * res * 4) == hipSuccess;
  // same order as decoder_feature_plan: [1] then every up level whose side <= max_att
  auto add = [&](int side, int C) {
    void* p = nullptr;
    if (ok && hipMalloc(&p, (size_t)B * ctx * side * side * C * dtype_size(e->dec_dt)) == hipSuccess) k->feat.push_back(p);
    else ok = false;
  };
  add(16, c.block_out_channels[nl - 1]);
  int s = 16;
  for (int i = 0; i < nl; ++i) { if (i != nl - 1) s *= 2; if (s <= c.max_att_resolution) add(s, c.block_out_channels[nl - 1 - i]); }
  if (!ok) {   // a partial cache would be overrun by the feature writes of detokenize: release everything
    (void)hipGetLastError();
    if (k->ctx_pixels) (void)hipFree(k->ctx_pixels);
    for (void* p : k->feat) (void)hipFree(p);
    delete k;
    return e->fail(IVG_ERR_HIP, "cache_create: hipMalloc failed");
  }
  *out = k;
  return IVG_OK;
}

void ivg_cache_destroy(ivg_engine* e, ivg_cache* c) {
  if (!c) return;
  if (e) (void)hipSetDevice(e->device);
  (void)hipDeviceSynchronize();
  if (c->ctx_pixels) (void)hipFree(c->ctx_pixels);
  for (void* p : c->feat) (void)hipFree(p);
  delete c;
}

int ivg_generate(ivg_engine* e, const int64_t* prompt, int64_t prompt_stride, int B, int L0, int n_new, const float* actions, int act_T,
                 int ctx, const float* uniforms, int top_k, int64_t* ids_out, float* reward_out, ivg_stream stream) {
  if (!e) return IVG_ERR_INVALID;
  if (e->cfg.num_layers <= 0) return e->fail(IVG_ERR_INVALID, "generate: engine was created without a transformer");
  if (B <= 0 || n_new < 1 || L0 < 1 || L0 + n_new > e->Lmax)
    return e->fail(IVG_ERR_CAPACITY, "generate: sequence of " + std::to_string(L0 + n_new) + " tokens exceeds the KV cache (" + std::to_string(e->Lmax) + ")");
  if (actions && (e->cfg.action_dim <= 0 || !e->act_w)) return e->fail(IVG_ERR_INVALID, "generate: actions given but the model is action-free");
  if (actions) {
    if (L0 < 257 * ctx || (L0 - 257 * ctx) % 17 != 0) return e->fail(IVG_ERR_INVALID, "generate: action-conditioned prompt must hold 257*ctx + 17*t tokens");
    const int last = (L0 - 257 * ctx) / 17 + n_new / 17 + ctx - 1;  // highest action row read (prompt slots + forced sdf slots)
    if (last >= act_T || act_T > e->cfg.max_frames) return e->fail(IVG_ERR_INVALID, "generate: action tensor too short (or longer than max_frames)");
  }
  return plan_then_run(e, (hipStream_t)stream, [&](Run& r) {
    return r.generate(prompt, prompt_stride, B, L0, n_new, actions, act_T, ctx, uniforms, top_k, ids_out, reward_out); });
}

int ivg_generate_shared(ivg_engine* e, const int64_t* prompts, int64_t prompt_stride, int n_groups, int group_size, int L0, int n_new,
                        const float* actions, int act_T, int ctx, const float* uniforms, int top_k, int force_sdf, int64_t* ids_out, float* reward_out,
                        ivg_stream stream) {
  if (!e) return IVG_ERR_INVALID;
  if (e->cfg.num_layers <= 0) return e->fail(IVG_ERR_INVALID, "generate: engine was created without a transformer");
  if (n_groups <= 0 || group_size <= 0) return e->fail(IVG_ERR_INVALID, "generate_shared: n_groups and group_size must be positive");
  if (n_new < 1 || L0 < 2 || L0 + n_new > e->Lmax)
    return e->fail(IVG_ERR_CAPACITY, "generate: sequence of " + std::to_string(L0 + n_new) + " tokens exceeds the KV cache (" + std::to_string(e->Lmax) + ")");
  if (actions && (e->cfg.action_dim <= 0 || !e->act_w)) return e->fail(IVG_ERR_INVALID, "generate: actions given but the model is action-free");
  if (actions) {
    // the shared prefix is [0, L0 - 1): it may not contain an action slot (those carry per-trajectory actions), so the prompt is the
    // context alone -- its last token, the first sdf slot, is fed per trajectory
    if (L0 != 257 * ctx) return e->fail(IVG_ERR_INVALID, "generate_shared: an action-conditioned shared prompt must hold exactly 257*ctx tokens");
    const int last = n_new / 17 + ctx - 1;
    if (last >= act_T || act_T > e->cfg.max_frames) return e->fail(IVG_ERR_INVALID, "generate: action tensor too short (or longer than max_frames)");
  }
  const int B = n_groups * group_size;
  if (group_size == 1)   // nothing to share: the plain entry (one prefill over all rows)
    return plan_then_run(e, (hipStream_t)stream, [&](Run& r) {
      return r.generate(prompts, prompt_stride, B, L0, n_new, actions, act_T, ctx, uniforms, top_k, ids_out, reward_out, false, nullptr, nullptr, nullptr,
                        force_sdf != 0); });
  return plan_then_run(e, (hipStream_t)stream, [&](Run& r) {
    return r.generate(prompts, prompt_stride, B, L0, n_new, actions, act_T, ctx, uniforms, top_k, ids_out, reward_out, false, nullptr, nullptr, nullptr,
                      force_sdf != 0, group_size); });
}

int ivg_generate_forced_sdf(ivg_engine* e, const int64_t* prompt, int64_t prompt_stride, int B, int L0, int n_new, int ctx, const float* uniforms,
                            int top_k, int64_t* ids_out, ivg_stream stream) {
  if (!e) return IVG_ERR_INVALID;
  if (e->cfg.num_layers <= 0) return e->fail(IVG_ERR_INVALID, "generate: engine was created without a transformer");
  if (B <= 0 || n_new < 1 || L0 < 1 || L0 + n_new > e->Lmax)
    return e->fail(IVG_ERR_CAPACITY, "generate: sequence of " + std::to_string(L0 + n_new) + " tokens exceeds the KV cache (" + std::to_string(e->Lmax) + ")");
  return plan_then_run(e, (hipStream_t)stream, [&](Run& r) {
    return r.generate(prompt, prompt_stride, B, L0, n_new, nullptr, 0, ctx, uniforms, top_k, ids_out, nullptr, false, nullptr, nullptr, nullptr, true); });
}

int ivg_generate_continue(ivg_engine* e, const int64_t* prompt, int64_t prompt_stride, int B, int L0, int n_new, const float* actions,
                          int act_T, int ctx, const float* uniforms, int top_k, int64_t* ids_out, float* reward_out, ivg_stream stream) {
  if (!e) return IVG_ERR_INVALID;
  if (e->cfg.num_layers <= 0) return e->fail(IVG_ERR_INVALID, "generate: engine was created without a transformer");
  if (!actions || e->cfg.action_dim <= 0 || !e->act_w) return e->fail(IVG_ERR_INVALID, "generate_continue: action-conditioned models only");
  if (B <= 0 || n_new < 1 || L0 < 2 || L0 + n_new > e->Lmax)
    return e->fail(IVG_ERR_CAPACITY, "generate: sequence of " + std::to_string(L0 + n_new) + " tokens exceeds the KV cache (" + std::to_string(e->Lmax) + ")");
  if (L0 < 257 * ctx || (L0 - 257 * ctx) % 17 != 0) return e->fail(IVG_ERR_INVALID, "generate: action-conditioned prompt must hold 257*ctx + 17*t tokens");
  const int last = (L0 - 257 * ctx) / 17 + n_new / 17 + ctx - 1;
  if (last >= act_T || act_T > e->cfg.max_frames) return e->fail(IVG_ERR_INVALID, "generate: action tensor too short (or longer than max_frames)");
  if (e->kv_B != B || e->kv_len != L0 - 1)
    return e->fail(IVG_ERR_INVALID, "generate_continue: the KV cache holds " + std::to_string(e->kv_len) + " positions of " + std::to_string(e->kv_B) +
                                        " trajectories, the call needs " + std::to_string(L0 - 1) + " of " + std::to_string(B));
  {  // same (batch, length) is not enough: another caller may have used the model in between.  Compare the cached prefix
     // (token ids + the action rows baked into its sdf slots) with the prompt on the device (one stream synchronisation).
    bool same = false;
    IVG_TRY(kv_prefix_matches_ids(e, prompt, prompt_stride, B, L0, actions, act_T, ctx, (hipStream_t)stream, &same));
    if (!same) return e->fail(IVG_ERR_INVALID, "generate_continue: the KV cache was built from a different prefix (tokens or actions differ)");
  }
  return plan_then_run(e, (hipStream_t)stream, [&](Run& r) {
    return r.generate(prompt, prompt_stride, B, L0, n_new, actions, act_T, ctx, uniforms, top_k, ids_out, reward_out, true); });
}

int ivg_generate_embeds(ivg_engine* e, const void* embeds, int B, int L0, int n_new, const float* uniforms, int top_k, int64_t* new_ids_out,
                        void* hidden_out, int allow_reuse, int* reused_out, ivg_stream stream) {
  if (!e) return IVG_ERR_INVALID;
  if (reused_out) *reused_out = 0;
  if (e->cfg.num_layers <= 0) return e->fail(IVG_ERR_INVALID, "generate: engine was created without a transformer");
  if (!embeds || !new_ids_out) return e->fail(IVG_ERR_INVALID, "generate_embeds: null argument");
  if (B <= 0 || B > std::min(e->cfg.max_batch, 128) || n_new < 1 || L0 < 1 || L0 + n_new > e->Lmax)
    return e->fail(IVG_ERR_CAPACITY, "generate_embeds: batch " + std::to_string(B) + " / sequence of " + std::to_string(L0 + n_new) +
                                         " tokens exceeds the capacity (" + std::to_string(std::min(e->cfg.max_batch, 128)) + " x " + std::to_string(e->Lmax) + ")");
  bool reuse = false;
  if (allow_reuse && L0 >= 2) IVG_TRY(kv_prefix_matches_embeds(e, embeds, B, L0, (hipStream_t)stream, &reuse));
  if (reused_out) *reused_out = reuse ? 1 : 0;
  return plan_then_run(e, (hipStream_t)stream, [&](Run& r) {
    return r.generate(nullptr, 0, B, L0, n_new, nullptr, 0, 1, uniforms, top_k, nullptr, nullptr, reuse, embeds, new_ids_out, hidden_out); });
}

int ivg_embed_tokens(ivg_engine* e, const int64_t* ids, int64_t ids_stride, int B, int L, void* out, ivg_stream stream) {
  if (!e) return IVG_ERR_INVALID;
  if (e->cfg.num_layers <= 0 || !e->embed) return e->fail(IVG_ERR_INVALID, "embed_tokens: engine was created without a transformer");
  if (B <= 0 || L <= 0) return IVG_OK;
  if (launch_embed(ids, ids_stride, e->embed, out, e->llm_dt, B, L, e->cfg.hidden_size, e->cfg.vocab_size, (hipStream_t)stream))
    return e->fail(IVG_ERR_HIP, "embed_tokens: launch failed");
  return IVG_OK;
}

int ivg_action_linear(ivg_engine* e, const float* actions, int rows, void* out, ivg_stream stream) {
  if (!e) return IVG_ERR_INVALID;
  if (e->cfg.action_dim <= 0 || !e->act_w) return e->fail(IVG_ERR_INVALID, "action_linear: the model is action-free");
  if (launch_action_embed(actions, e->act_w, e->act_b, out, e->llm_dt, rows, e->cfg.action_dim, e->cfg.hidden_size, (hipStream_t)stream))
    return e->fail(IVG_ERR_HIP, "action_linear: launch failed");
  return IVG_OK;
}

int ivg_reward_linear(ivg_engine* e, const void* hidden, int rows, float* out, ivg_stream stream) {
  if (!e) return IVG_ERR_INVALID;
  if (!e->rew_w_raw || !e->rew_b) return e->fail(IVG_ERR_MISSING, "reward_linear: the model has no reward head");
  if (launch_rowdot(hidden, e->rew_w_raw, e->rew_b, out, rows, e->cfg.hidden_size, -1.0f, e->llm_dt, (hipStream_t)stream))
    return e->fail(IVG_ERR_HIP, "reward_linear: launch failed");
  return IVG_OK;
}

int ivg_logits(ivg_engine* e, const int64_t* ids, int B, int L, const float* actions, int act_T, int ctx, float* logits_out, ivg_stream stream) {
  if (!e) return IVG_ERR_INVALID;
  if (e->cfg.num_layers <= 0) return e->fail(IVG_ERR_INVALID, "logits: engine was created without a transformer");
  if (B <= 0 || B > std::min(e->cfg.max_batch, 128) || L > e->Lmax) return e->fail(IVG_ERR_CAPACITY, "logits: batch or length exceeds capacity");
  if (actions && (e->cfg.action_dim <= 0 || !e->act_w)) return e->fail(IVG_ERR_INVALID, "logits: actions given but the model is action-free");
  return plan_then_run(e, (hipStream_t)stream, [&](Run& r) {
    const void* act_emb = nullptr;
    if (actions) {
      char* buf = (char*)e->ws.alloc((size_t)B * act_T * e->cfg.hidden_size * dtype_size(e->llm_dt));
      if (!r.planning) {
        int rc = launch_action_embed(actions, e->act_w, e->act_b, buf, e->llm_dt, B * act_T, e->cfg.action_dim, e->cfg.hidden_size, r.st);
        if (rc) return e->fail(IVG_ERR_HIP, "action_embed launch failed");
      }
      act_emb = buf;
    }
    return r.prefill(ids, L, B, L, act_emb, act_T, ctx, true, logits_out, nullptr, nullptr);
  });
}

int ivg_eval_forward(ivg_engine* e, const int64_t* ids, const int64_t* labels, int B, int L, const float* actions, int act_T, int ctx,
                     float* token_nll_out, float* loss_rows_out, void* hidden_out, ivg_stream stream) {
  if (!e) return IVG_ERR_INVALID;
  if (e->cfg.num_layers <= 0) return e->fail(IVG_ERR_INVALID, "eval_forward: engine was created without a transformer");
  if (!ids || !labels || !token_nll_out || !loss_rows_out) return e->fail(IVG_ERR_INVALID, "eval_forward: null argument");
  if (B <= 0 || B > std::min(e->cfg.max_batch, 128) || L < 2 || L > e->Lmax) return e->fail(IVG_ERR_CAPACITY, "eval_forward: batch or length exceeds capacity");
  if (actions && (e->cfg.action_dim <= 0 || !e->act_w)) return e->fail(IVG_ERR_INVALID, "eval_forward: actions given but the model is action-free");
  return plan_then_run(e, (hipStream_t)stream, [&](Run& r) {
    const void* act_emb = nullptr;
    if (actions) {
      char* buf = (char*)e->ws.alloc((size_t)B * act_T * e->cfg.hidden_size * dtype_size(e->llm_dt));
      if (!r.planning && launch_action_embed(actions, e->act_w, e->act_b, buf, e->llm_dt, B * act_T, e->cfg.action_dim, e->cfg.hidden_size, r.st))
        return e->fail(IVG_ERR_HIP, "action_embed launch failed");
      act_emb = buf;
    }
    int rc = r.prefill(ids, L, B, L, act_emb, act_T, ctx, true, nullptr, nullptr, nullptr, nullptr, hidden_out, labels, token_nll_out);
    if (rc) return rc;
    if (!r.planning && launch_ce_reduce(token_nll_out, labels, B, L, e->cfg.vocab_size, loss_rows_out, r.st)) return e->fail(IVG_ERR_HIP, "ce_reduce launch failed");
    return 0;
  });
}

int ivg_action_recon_sqerr(ivg_engine* e, const void* hidden, const float* actions, int B, int L, int act_T, int ctx, int prelude, float* out,
                           ivg_stream stream) {
  if (!e) return IVG_ERR_INVALID;
  if (!e->ar_w || !e->ar_b) return e->fail(IVG_ERR_MISSING, "action_recon: 'action_recon_linear' is not in the weight table");
  if (!hidden || !actions || !out || B <= 0 || prelude < 0 || prelude >= L) return e->fail(IVG_ERR_INVALID, "action_recon: bad argument");
  if (ctx - 1 + (L - 1 - prelude) / 17 >= act_T) return e->fail(IVG_ERR_INVALID, "action_recon: action tensor too short");
  if (launch_action_recon(hidden, e->ar_w, e->ar_b, actions, B, L, e->cfg.hidden_size, e->cfg.action_dim, act_T, ctx, prelude, out, e->llm_dt,
                          (hipStream_t)stream))
    return e->fail(IVG_ERR_HIP, "action_recon: launch failed");
  return IVG_OK;
}

int ivg_ingest_frames(const uint8_t* frames, int T, int H, int W, int center_crop, void* clip_out, int out_dtype, int resolution, ivg_stream stream) {
  if (!frames || !clip_out || (out_dtype != IVG_F32 && out_dtype != IVG_BF16)) return IVG_ERR_INVALID;
  const int rc = launch_ingest(frames, T, H, W, center_crop, clip_out, (DType)out_dtype, resolution, (hipStream_t)stream);
  return rc == 0 ? IVG_OK : (rc == (int)hipErrorInvalidValue ? IVG_ERR_INVALID : IVG_ERR_HIP);
}

size_t ivg_frame_metrics_ws_bytes(int n_samples, int T, int H, int W) { return frame_metrics_ws_bytes(n_samples, T, H, W); }

int ivg_frame_metrics(const void* gt, int gt_dtype, int B, int T_gt, int gt_t0, const float* pred, int n_samples, int T_pr, int pr_t0, int T, int H,
                      int W, float* rows_out, void* ws, size_t ws_bytes, ivg_stream stream) {
  if (!gt || !pred || !rows_out || !ws) return IVG_ERR_INVALID;
  if (B <= 0 || n_samples <= 0 || n_samples % B != 0 || T <= 0 || gt_t0 < 0 || pr_t0 < 0 || gt_t0 + T > T_gt || pr_t0 + T > T_pr || H < 11 || W < 11)
    return IVG_ERR_INVALID;
  if (ws_bytes < frame_metrics_ws_bytes(n_samples, T, H, W)) return IVG_ERR_CAPACITY;
  return launch_frame_metrics(gt, (DType)gt_dtype, B, T_gt, gt_t0, pred, n_samples, T_pr, pr_t0, T, H, W, rows_out, ws, (hipStream_t)stream) ? IVG_ERR_HIP
                                                                                                                                                : IVG_OK;
}

int ivg_profile_attn_fit(ivg_engine* e, double* fixed_us, double* gbps) {
  if (!e || !fixed_us || !gbps) return IVG_ERR_INVALID;
  *fixed_us = e->attn_fit_fixed_us; *gbps = e->attn_fit_gbps;
  return IVG_OK;
}

int ivg_profile_enable(ivg_engine* e, int k, int enable) {
  if (!e || k < 0 || k >= IVG_K_COUNT) return IVG_ERR_INVALID;
  if (k == IVG_K_DECODE_ATTN) { e->attn_prof_on = enable != 0 && e->attn_prof; return IVG_OK; }
  if (k == IVG_K_DECODE_GEMM) {
    if (enable && !e->gemm_prof && e->cfg.num_layers > 0) {
      const size_t n = (size_t)(4 * e->cfg.num_layers + 1) * IVG_GEMM_PROF_SLOTS * 2 * e->Lmax * 8;
      API_CK(hipMalloc((void**)&e->gemm_prof, n));
      API_CK(hipMemset(e->gemm_prof, 0, n));
    }
    e->gemm_prof_on = enable != 0 && e->gemm_prof;
    return IVG_OK;
  }
  e->prof[k].enabled = enable != 0;
  return IVG_OK;
}

int ivg_profile_read(ivg_engine* e, int k, ivg_profile_stats* out) {
  if (!e || !out || k < 0 || k >= IVG_K_COUNT) return IVG_ERR_INVALID;
  API_CK(hipDeviceSynchronize());
  if (k == IVG_K_DECODE_ATTN) {
    // launch windows stamped by the kernel itself (it runs inside a replayed hipGraph, where HIP events cannot bracket
    // single launches): last ivg_generate call only; bytes = K and V rows read per launch
    out->launches = 0; out->total_ms = 0; out->total_flops = 0; out->total_bytes = 0;
    if (!e->attn_prof) return IVG_OK;
    const int L = e->Lmax, nl = e->cfg.num_layers, NS = IVG_ATTN_PROF_SLOTS;
    std::vector<unsigned long long> h((size_t)nl * NS * 2 * L);
    API_CK(hipMemcpy(h.data(), e->attn_prof, h.size() * 8, hipMemcpyDeviceToHost));
    // least-squares line  duration = fixed + bytes / rate  over the launches (they differ in cache length)
    double sn = 0, sx = 0, sy = 0, sxx = 0, sxy = 0;
    for (int l = 0; l < nl; ++l)
      for (int p = 0; p < L; ++p) {
        unsigned long long s = ~0ull, t = 0;
        for (int k = 0; k < NS; ++k) {
          const unsigned long long cs = h[((size_t)(l * NS + k) * 2) * L + p], ct = h[((size_t)(l * NS + k) * 2 + 1) * L + p];
          if (cs) s = std::min(s, ~cs);
          t = std::max(t, ct);
        }
        if (t == 0 || s == ~0ull || t < s) continue;
        const double ms = (double)(t - s) * 1e-5;   // 100 MHz wall clock -> ms
        const double bytes = 2.0 * e->attn_prof_B * e->heads * (double)(p + 1) * e->hd * (double)e->kv_elem_bytes();
        out->launches++;
        out->total_ms += ms;
        out->total_bytes += bytes;
        out->total_flops += 4.0 * e->attn_prof_B * e->heads * (double)(p + 1) * e->hd;
        sn += 1; sx += bytes; sy += ms; sxx += bytes * bytes; sxy += bytes * ms;
      }
    const double den = sn * sxx - sx * sx;
    e->attn_fit_fixed_us = 0; e->attn_fit_gbps = 0;
    if (sn > 2 && den > 0) {
      const double slope = (sn * sxy - sx * sy) / den;            // ms per byte
      e->attn_fit_fixed_us = (sy - slope * sx) / sn * 1e3;
      e->attn_fit_gbps = slope > 0 ? 1e-6 / slope : 0;             // bytes/ms -> GB/s
    }
    return IVG_OK;
  }
  if (k == IVG_K_DECODE_GEMM) {
    // launch windows stamped by the GEMM kernels of the last ivg_generate call; bytes = the weight matrix a launch streams
    out->launches = 0; out->total_ms = 0; out->total_flops = 0; out->total_bytes = 0;
    if (!e->gemm_prof) return IVG_OK;
    const int L = e->Lmax, nl = e->cfg.num_layers, NS = IVG_GEMM_PROF_SLOTS, ng = 4 * nl + 1;
    const double H = e->cfg.hidden_size, I = e->cfg.intermediate_size, V = e->cfg.vocab_size, es = (double)dtype_size(e->llm_dt);
    const double wbytes[5] = {3 * H * H * es, H * H * es, 2 * I * H * es, H * I * es, V * H * es};
    std::vector<unsigned long long> h((size_t)ng * NS * 2 * L);
    API_CK(hipMemcpy(h.data(), e->gemm_prof, h.size() * 8, hipMemcpyDeviceToHost));
    for (int i = 0; i < 5; ++i) { e->gemm_kind_ms[i] = 0; e->gemm_kind_n[i] = 0; }
    for (int g = 0; g < ng; ++g)
      for (int p = 0; p < L; ++p) {
        unsigned long long s = ~0ull, t = 0;
        for (int q = 0; q < NS; ++q) {
          const unsigned long long cs = h[((size_t)(g * NS + q) * 2) * L + p], ct = h[((size_t)(g * NS + q) * 2 + 1) * L + p];
          if (cs) s = std::min(s, ~cs);
          t = std::max(t, ct);
        }
        if (t == 0 || s == ~0ull || t < s) continue;
        const double wb = g == 4 * nl ? wbytes[4] : wbytes[g & 3];
        out->launches++;
        out->total_ms += (double)(t - s) * 1e-5;
        e->gemm_kind_ms[g == 4 * nl ? 4 : (g & 3)] += (double)(t - s) * 1e-5;
        e->gemm_kind_n[g == 4 * nl ? 4 : (g & 3)] += 1;
        out->total_bytes += wb;
        out->total_flops += wb / es * 2.0 * e->gemm_prof_B;
      }
    return IVG_OK;
  }
  ProfClass& pc = e->prof[k];
  out->launches = 0; out->total_ms = 0; out->total_flops = 0; out->total_bytes = 0;
  for (auto& s : pc.used) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) { out->launches++; out->total_ms += ms; out->total_flops += s.flops; out->total_bytes += s.bytes; }
    pc.pool.push_back(s);
  }
  pc.used.clear();
  return IVG_OK;
}

int ivg_profile_gemm_kinds(ivg_engine* e, double* mean_us, int64_t* launches) {
  if (!e || !mean_us || !launches) return IVG_ERR_INVALID;
  for (int i = 0; i < 5; ++i) { launches[i] = e->gemm_kind_n[i]; mean_us[i] = e->gemm_kind_n[i] ? 1e3 * e->gemm_kind_ms[i] / (double)e->gemm_kind_n[i] : 0.0; }
  return IVG_OK;
}

// ---------------------------------------------------------------------------------------------- op-level hooks
int ivg_op_igemm(const ivg_igemm_args* a, int dtype, ivg_stream stream) {
  IgemmArgs g;
  g.X = a->X; g.W = a->W; g.Y = a->Y; g.R = a->R; g.bias = a->bias;
  g.Nimg = a->Nimg; g.Hin = a->Hin; g.Win = a->Win; g.Cin = a->Cin; g.ldx = a->ldx; g.Hout = a->Hout; g.Wout = a->Wout;
  g.KH = a->KH; g.KW = a->KW; g.stride = a->stride; g.pad = a->pad; g.ups = a->ups; g.N = a->N; g.ldw = a->ldw;
  g.c_img = a->c_img; g.c_pix = a->c_pix; g.c_ch = a->c_ch; g.c_grp = a->c_grp; g.c_grp_stride = a->c_grp_stride;
  g.flags = a->flags; g.alpha = a->alpha; g.nb0 = a->nb0; g.nb1 = a->nb1; g.nb2 = a->nb2;
  for (int i = 0; i < 3; ++i) { g.sa[i] = a->sa[i]; g.sw[i] = a->sw[i]; g.sy[i] = a->sy[i]; }
  if (dtype == IVG_F32X3) { g.x3 = true; dtype = IVG_F32; }   // fp32 tensors, split-bf16 arithmetic: what Run::gemm sets for an x3 engine
  if (g.KH == 3 && g.KW == 3 && g.stride == 1 && !g.x3) {  // same dispatch as the engine: LDS-halo kernel first
    const int rc = launch_conv3x3(g, (DType)dtype, (hipStream_t)stream);
    if (rc == 0) return IVG_OK;
    if (rc > 0) return IVG_ERR_HIP;
  }
  {  // large dense GEMMs: the 256 x 256-tile kernel first, as the engine does
    const int rc = launch_gemm256(g, (DType)dtype, (hipStream_t)stream);
    if (rc == 0) return IVG_OK;
    if (rc > 0) return IVG_ERR_HIP;
  }
  return launch_igemm(g, (DType)dtype, (hipStream_t)stream) ? IVG_ERR_HIP : IVG_OK;
}

int ivg_op_conv_gn(const ivg_igemm_args* a, int dtype, void* gn_part, int groups, const float* gamma, const float* beta, void* gn_out, float eps,
                   int silu, ivg_stream stream) {
  // unit-test hook of the fused path: 3x3 convolution whose epilogue reduces the GroupNorm statistics of its output, followed by
  // the apply-only GroupNorm that consumes them.  Returns the number of statistics chunks per image (> 0) or a negative status.
  IgemmArgs g;
  g.X = a->X; g.W = a->W; g.Y = a->Y; g.R = a->R; g.bias = a->bias;
  g.Nimg = a->Nimg; g.Hin = a->Hin; g.Win = a->Win; g.Cin = a->Cin; g.ldx = a->ldx; g.Hout = a->Hout; g.Wout = a->Wout;
  g.KH = a->KH; g.KW = a->KW; g.stride = a->stride; g.pad = a->pad; g.ups = a->ups; g.N = a->N; g.ldw = a->ldw;
  g.c_img = a->c_img; g.c_pix = a->c_pix; g.c_ch = a->c_ch; g.c_grp = a->c_grp; g.c_grp_stride = a->c_grp_stride;
  g.flags = a->flags; g.alpha = a->alpha;
  g.gn_part = gn_part; g.gn_groups = groups;
  const int rc = launch_conv3x3(g, (DType)dtype, (hipStream_t)stream);
  if (rc != 0 || g.gn_chunks <= 0) return rc > 0 ? IVG_ERR_HIP : IVG_ERR_INVALID;
  if (launch_groupnorm_apply(a->Y, gn_out, gn_part, g.gn_chunks, gamma, beta, nullptr, a->Nimg, a->Hout * a->Wout, a->N, groups, eps, silu,
                             (DType)dtype, (hipStream_t)stream))
    return IVG_ERR_HIP;
  return g.gn_chunks;
}

int ivg_op_conv_subpixel(const ivg_igemm_args* a, int dtype, const void* w_sub, const void* w_x3, const void* w_sub_x3, void* gn_part, int groups,
                         ivg_stream stream) {
  // unit-test hook: the nearest-x2 upsampling convolution in sub-pixel form (a->ups must be 1; a->W = the plain [N][9 Cin] matrix,
  // unused unless the shape falls back).  w_x3 / w_sub_x3 != NULL: split-bf16 arithmetic on fp32 tensors.  gn_part != NULL: output
  // statistics from the epilogue; returns the chunks per image then (0 without), IVG_ERR_INVALID when the sub-pixel kernel did not run.
  if (!a->ups || !w_sub) return IVG_ERR_INVALID;
  IgemmArgs g;
  g.X = a->X; g.W = a->W; g.Y = a->Y; g.R = a->R; g.bias = a->bias;
  g.Nimg = a->Nimg; g.Hin = a->Hin; g.Win = a->Win; g.Cin = a->Cin; g.ldx = a->ldx; g.Hout = a->Hout; g.Wout = a->Wout;
  g.KH = a->KH; g.KW = a->KW; g.stride = a->stride; g.pad = a->pad; g.ups = a->ups; g.N = a->N; g.ldw = a->ldw;
  g.c_img = a->c_img; g.c_pix = a->c_pix; g.c_ch = a->c_ch; g.c_grp = a->c_grp; g.c_grp_stride = a->c_grp_stride;
  g.flags = a->flags; g.alpha = a->alpha;
  g.W_sub = w_sub; g.W_x3 = w_x3; g.W_sub_x3 = w_sub_x3;
  g.gn_part = gn_part; g.gn_groups = groups;
  const long long before = conv3x3_subpixel_launches();
  const int rc = launch_conv3x3(g, (DType)dtype, (hipStream_t)stream);
  if (rc != 0) return rc > 0 ? IVG_ERR_HIP : IVG_ERR_INVALID;
  if (conv3x3_subpixel_launches() == before) return IVG_ERR_INVALID;
  return gn_part ? g.gn_chunks : 0;
}

int ivg_op_xattn(const void* q, const void* Kp, const void* VpT, void* out, int M, int F, int P, int kv, int C, int nh, int dtype, ivg_stream stream) {
  const int rc = launch_xattn(q, Kp, VpT, out, M, F, P, kv, C, nh, (DType)dtype, (hipStream_t)stream);
  return rc == 0 ? IVG_OK : (rc > 0 ? IVG_ERR_HIP : IVG_ERR_INVALID);
}

int ivg_op_gn_conv(const ivg_igemm_args* a, int dtype, int groups, const float* gamma, const float* beta, float eps, void* ws, ivg_stream stream) {
  // unit-test hook: y = conv3x3(silu(GroupNorm(x))) with the normalisation applied inside the convolution's input staging.
  // ws: scratch of at least Nimg * (ceil(H*W/1024) * groups * 16 + Cin * 8) bytes.  Returns IVG_ERR_INVALID when the fused kernel
  // does not cover the shape.
  IgemmArgs g;
  g.X = a->X; g.W = a->W; g.Y = a->Y; g.R = a->R; g.bias = a->bias;
  g.Nimg = a->Nimg; g.Hin = a->Hin; g.Win = a->Win; g.Cin = a->Cin; g.ldx = a->ldx; g.Hout = a->Hout; g.Wout = a->Wout;
  g.KH = a->KH; g.KW = a->KW; g.stride = a->stride; g.pad = a->pad; g.ups = a->ups; g.N = a->N; g.ldw = a->ldw;
  g.c_img = a->c_img; g.c_pix = a->c_pix; g.c_ch = a->c_ch; g.c_grp = a->c_grp; g.c_grp_stride = a->c_grp_stride;
  g.flags = a->flags; g.alpha = a->alpha;
  const int P = a->Hin * a->Win, nch = gn_num_chunks(P);
  char* part = (char*)ws;
  char* coef = part + (size_t)a->Nimg * nch * groups * 16;
  if (launch_groupnorm_partial(a->X, part, a->Nimg, P, a->Cin, groups, (DType)dtype, (hipStream_t)stream)) return IVG_ERR_HIP;
  if (launch_gn_coef(part, nch, gamma, beta, a->Nimg, P, a->Cin, groups, eps, coef, (hipStream_t)stream)) return IVG_ERR_HIP;
  g.gn_in_coef = coef;
  const int rc = launch_conv3x3(g, (DType)dtype, (hipStream_t)stream);
  return rc == 0 ? IVG_OK : (rc > 0 ? IVG_ERR_HIP : IVG_ERR_INVALID);
}

int ivg_op_conv_x3(const ivg_igemm_args* a, const void* w_x3, int groups, const float* gamma, const float* beta, float eps, void* ws, ivg_stream stream) {
  // unit-test hook of the split-bf16 3x3 convolution (fp32 tensors, weights pre-split by packing.py pack_x3); gamma != NULL: with
  // GroupNorm + SiLU of the input applied (and the result split) inside the staging, ws as in ivg_op_gn_conv
  IgemmArgs g;
  g.X = a->X; g.W = a->W; g.Y = a->Y; g.R = a->R; g.bias = a->bias;
  g.Nimg = a->Nimg; g.Hin = a->Hin; g.Win = a->Win; g.Cin = a->Cin; g.ldx = a->ldx; g.Hout = a->Hout; g.Wout = a->Wout;
  g.KH = a->KH; g.KW = a->KW; g.stride = a->stride; g.pad = a->pad; g.ups = a->ups; g.N = a->N; g.ldw = a->ldw;
  g.c_img = a->c_img; g.c_pix = a->c_pix; g.c_ch = a->c_ch; g.c_grp = a->c_grp; g.c_grp_stride = a->c_grp_stride;
  g.flags = a->flags; g.alpha = a->alpha;
  g.W_x3 = w_x3;
  if (!w_x3) return IVG_ERR_INVALID;
  if (gamma) {
    const int P = a->Hin * a->Win, nch = gn_num_chunks(P);
    char* part = (char*)ws;
    char* coef = part + (size_t)a->Nimg * nch * groups * 16;
    if (launch_groupnorm_partial(a->X, part, a->Nimg, P, a->Cin, groups, F32, (hipStream_t)stream)) return IVG_ERR_HIP;
    if (launch_gn_coef(part, nch, gamma, beta, a->Nimg, P, a->Cin, groups, eps, coef, (hipStream_t)stream)) return IVG_ERR_HIP;
    g.gn_in_coef = coef;
  }
  const int rc = launch_conv3x3(g, F32, (hipStream_t)stream);
  return rc == 0 ? IVG_OK : (rc > 0 ? IVG_ERR_HIP : IVG_ERR_INVALID);
}

int64_t ivg_debug_counter(const char* name) {
  if (name && !strcmp(name, "conv3x3_subpixel")) return conv3x3_subpixel_launches();
  if (name && !strcmp(name, "gemm256x3")) return gemm256x3_launches();
  if (name && !strcmp(name, "decode_attn24")) return decode_attn24_launches();
  if (name && !strcmp(name, "decode_gemm_gen3")) return decode_gemm_launches(3);
  if (name && !strcmp(name, "decode_gemm_gen2")) return decode_gemm_launches(2);
  return -1;
}

int ivg_op_skinny(const void* X, const void* W, void* Y, int M, int N, int K, int ldx, int ldw, int ldy, int flags, int dtype, ivg_stream stream) {
  return ivg_op_skinny_policy(X, W, Y, M, N, K, ldx, ldw, ldy, flags, dtype, 0, 0, stream);
}

int ivg_op_skinny_policy(const void* X, const void* W, void* Y, int M, int N, int K, int ldx, int ldw, int ldy, int flags, int dtype, int lds_kb,
                         int w_shared, ivg_stream stream) {
  if (lds_kb != 0 && (lds_kb < 16 || lds_kb > 160)) return IVG_ERR_INVALID;
  SkinnyArgs s; s.X = X; s.W = W; s.Y = Y; s.M = M; s.N = N; s.K = K; s.ldx = ldx; s.ldw = ldw; s.ldy = ldy; s.flags = flags;
  s.lds_kb = lds_kb; s.w_shared = w_shared != 0;
  return launch_skinny(s, (DType)dtype, (hipStream_t)stream) ? IVG_ERR_HIP : IVG_OK;
}

int ivg_op_groupnorm(const void* X, void* Y, void* ws, const float* gamma, const float* beta, const float* pos, int N, int P, int C, int groups,
                     float eps, int silu, int dtype, ivg_stream stream) {
  return launch_groupnorm(X, Y, ws, gamma, beta, pos, N, P, C, groups, eps, silu, (DType)dtype, (hipStream_t)stream) ? IVG_ERR_HIP : IVG_OK;
}

int ivg_op_softmax(const float* S, void* P, int64_t rows, int Lq, int Lk, int lds, int ldp, int causal, int dtype, ivg_stream stream) {
  return launch_softmax(S, P, rows, Lq, Lk, lds, ldp, causal, (DType)dtype, (hipStream_t)stream) ? IVG_ERR_HIP : IVG_OK;
}

int ivg_op_vq_argmin(const float* z, const float* codebook, float* ee_ws, int64_t* out, int R, int n_e, ivg_stream stream) {
  if (launch_sqnorm_rows(codebook, ee_ws, n_e, 64, (hipStream_t)stream)) return IVG_ERR_HIP;
  TokMap mp{1, 1, 1, 0, 1};  // identity: row r -> out[r]
  return launch_vq_argmin(z, codebook, ee_ws, out, mp, 0, R, n_e, (hipStream_t)stream) ? IVG_ERR_HIP : IVG_OK;
}

int ivg_op_add_rmsnorm(void* x, const float* w, void* out, int M, int H, float eps, int dtype, ivg_stream stream) {
  return launch_add_rmsnorm(x, H, w, out, M, H, eps, (DType)dtype, (hipStream_t)stream) ? IVG_ERR_HIP : IVG_OK;
}

int ivg_op_conv_in(const void* video, int video_dtype, const float* w, const float* bias, void* Y, int dtype, int N, int per, int T_total, int t0,
                   int H, int W, int C0, ivg_stream stream) {
  return launch_conv_in(video, (DType)video_dtype, w, bias, Y, (DType)dtype, N, per, T_total, t0, H, W, C0, (hipStream_t)stream) ? IVG_ERR_HIP : IVG_OK;
}

int ivg_op_shared_decode_attn(const void* qkv, void* kc, void* vc, void* out, const float* cos_t, const float* sin_t, int B, int heads, int hd, int Lmax,
                              int pos, int P, int G, int row0, int dtype, ivg_stream stream) {
  // unit-test hook of one decode-attention step of a shared-context rollout (decode_attn_kernel SHARED)
  if (B <= 0 || G < 1 || P < 0 || P > pos || pos >= Lmax || row0 > 0) return IVG_ERR_INVALID;
  StepState* state = nullptr;
  if (hipMalloc((void**)&state, sizeof(StepState)) != hipSuccess) return IVG_ERR_HIP;
  hipStream_t st = (hipStream_t)stream;
  int rc = launch_state_set(state, pos, 1, st);
  if (!rc) rc = launch_decode_attn(qkv, kc, vc, out, cos_t, sin_t, B, heads, hd, Lmax, state, nullptr, (DType)dtype, st, P, G, row0);
  (void)hipStreamSynchronize(st);
  (void)hipFree(state);
  return rc == 0 ? IVG_OK : (rc > 0 ? IVG_ERR_HIP : IVG_ERR_INVALID);
}

int ivg_op_kv24_pack(const float* k32, const float* v32, void* kc, void* vc, int BH, int L, int Lmax, ivg_stream stream) {
  if (BH <= 0 || L < 0 || L > Lmax) return IVG_ERR_INVALID;
  return launch_kv24_pack(k32, v32, kc, vc, BH, L, Lmax, (hipStream_t)stream) ? IVG_ERR_HIP : IVG_OK;
}

int ivg_op_decode_attn24(const float* qkv, void* kc, void* vc, float* out, const float* cos_t, const float* sin_t, int B, int heads, int Lmax, int pos,
                         int P, int G, int row0, ivg_stream stream) {
  // unit-test hook of one decode-attention step over the 24-bit K / V cache of the x3 rollout (decode_attn24_kernel; G > 1: SHARED)
  if (B <= 0 || G < 1 || P < 0 || P > pos || pos >= Lmax || row0 > 0) return IVG_ERR_INVALID;
  StepState* state = nullptr;
  if (hipMalloc((void**)&state, sizeof(StepState)) != hipSuccess) return IVG_ERR_HIP;
  hipStream_t st = (hipStream_t)stream;
  int rc = launch_state_set(state, pos, 1, st);
  if (!rc) rc = launch_decode_attn24(qkv, kc, vc, out, cos_t, sin_t, B, heads, Lmax, state, nullptr, st, P, G, row0);
  (void)hipStreamSynchronize(st);
  (void)hipFree(state);
  return rc == 0 ? IVG_OK : (rc > 0 ? IVG_ERR_HIP : IVG_ERR_INVALID);
}

int ivg_op_sample(const float* logits, int B, int V, int top_k, float temperature, const float* uniforms, int64_t* out, ivg_stream stream) {
  if (!(temperature > 0.0f) || !std::isfinite(temperature)) return IVG_ERR_INVALID;   // as ivg_set_temperature
  // one draw per row through the rollout's sampler kernel (token j = 1 of a prompt of length 0; no embedding: H = 0)
  StepState* state = nullptr;
  if (hipMalloc((void**)&state, sizeof(StepState)) != hipSuccess) return IVG_ERR_HIP;
  hipStream_t st = (hipStream_t)stream;
  int rc = launch_state_set(state, 0, 1, st);
  SampleArgs sa{};
  sa.logits = logits; sa.V = V; sa.uniforms = uniforms; sa.n_uni = 1; sa.top_k = top_k;
  sa.ids_out = out; sa.ids_stride = 1; sa.L0 = 0; sa.forced_period = 0; sa.forced_token = 0;
  sa.E = logits; sa.x = out; sa.H = 0; sa.act = nullptr; sa.act_T = 0; sa.ctx = 1; sa.slot0 = 0; sa.state = state;
  sa.temperature = temperature;
  if (!rc) rc = launch_sample_embed(sa, B, F32, st);
  (void)hipStreamSynchronize(st);
  (void)hipFree(state);
  return rc ? IVG_ERR_HIP : IVG_OK;
}

}  // extern "C"
