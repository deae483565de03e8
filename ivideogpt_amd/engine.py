"""Thin Python handle on a libivg engine (one per process per GPU).  All compute happens in the HIP
library; PyTorch only owns the device buffers (weights, inputs, outputs) and the stream."""
import ctypes as C

import torch

from . import _lib
from .packing import config_code, dtype_code


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class Engine:
    def __init__(self, device, tensors, tok_cfg=None, llm_cfg=None, action_dim=0, reward_head=False,
                 encode_dtype="fp32", decode_dtype="bf16", llm_dtype="bf16", max_batch=1, max_frames=16, max_seq=0,
                 decode_lds_kb=0):
        self.lib = _lib.load()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("ivideogpt_amd runs on an MI355X only (device must be 'cuda'); there is no CPU path")
        self.tensors = tensors  # keep the packed weights alive: the engine borrows their device memory
        cfg = _lib.IvgConfig()
        if tok_cfg is not None:
            ch = list(tok_cfg["block_out_channels"])
            cfg.n_levels = len(ch)
            for i, c in enumerate(ch):
                cfg.block_out_channels[i] = c
            for k in ("layers_per_block", "latent_channels", "vq_embed_dim", "num_vq_embeddings", "num_dyn_embeddings",
                      "norm_num_groups", "context_length", "max_att_resolution", "resolution", "patch_size"):
                setattr(cfg, k, int(tok_cfg[k]))
            cfg.mid_block_add_attention = int(bool(tok_cfg["mid_block_add_attention"]))
        if llm_cfg is not None:
            cfg.hidden_size, cfg.intermediate_size = llm_cfg["hidden_size"], llm_cfg["intermediate_size"]
            cfg.num_layers, cfg.num_heads = llm_cfg["num_hidden_layers"], llm_cfg["num_attention_heads"]
            cfg.vocab_size, cfg.max_position_embeddings = llm_cfg["vocab_size"], llm_cfg["max_position_embeddings"]
            cfg.rms_norm_eps = llm_cfg["rms_norm_eps"]
            cfg.action_dim, cfg.reward_head = int(action_dim or 0), int(bool(reward_head))
        cfg.encode_dtype, cfg.decode_dtype, cfg.llm_dtype = config_code(encode_dtype), config_code(decode_dtype), config_code(llm_dtype)
        cfg.max_batch, cfg.max_frames, cfg.max_seq = int(max_batch), int(max_frames), int(max_seq)
        cfg.decode_lds_kb = int(decode_lds_kb or 0)   # this engine's launch policy (include/ivg.h): 0 = the process default
        self.cfg = cfg
        names = [n.encode() for n in tensors]
        table = (_lib.IvgTensor * len(tensors))()
        for i, (n, t) in enumerate(tensors.items()):
            assert t.is_cuda and t.is_contiguous(), n
            table[i].name = names[i]
            table[i].data = t.data_ptr()
            table[i].dtype = dtype_code(t.dtype)
            table[i].ndim = min(t.dim(), 4)
            shp = list(t.shape) if t.dim() <= 4 else [t.numel()]
            for j, s in enumerate(shp[:4]):
                table[i].shape[j] = s
            if t.dim() > 4:
                table[i].ndim = 1
        self._names = names
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = self.lib.ivg_create(C.byref(cfg), table, len(tensors), self.device.index or 0, C.byref(h))
        _lib.check(rc, None, "ivg_create")
        self.h = h
        self.max_batch, self.max_frames = int(max_batch), int(max_frames)
        # dedicated non-default stream for callers on the legacy default stream (graph capture needs one) -- created on first use: a
        # stream nobody runs on still takes a slot in the round-robin over the (4) hardware queues, and two batches in flight whose
        # streams land on the SAME hardware queue do not overlap at all (bench.py --lanes: 4,230 instead of 5,050 frames/s)
        self._stream = None
        self._caches = set()   # live detokenize caches (device memory owned here: released with the engine)
        self._clamp_out = False
        self._temperature = 1.0
        self._run = None       # the stream of the last call (the dedicated one, or the caller's own)

    def close(self):
        if getattr(self, "h", None):
            for c in list(getattr(self, "_caches", ())):
                self.lib.ivg_cache_destroy(self.h, C.c_void_p(c))
            self._caches = set()
            self.lib.ivg_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- stream plumbing.  A caller that made a stream of its own current (`with torch.cuda.stream(s):`, one per batch in flight) gets
    # the launches on THAT stream, like any PyTorch op; under the legacy default stream the engine runs on its dedicated stream,
    # ordered after the caller's current stream and before its future work.  The engine's workspace belongs to one stream at a time:
    # a call on another stream than the previous call's first waits for it.
    class _On:
        def __init__(self, eng):
            self.eng = eng

        def __enter__(self):
            eng = self.eng
            cur = torch.cuda.current_stream(eng.device)
            if cur == torch.cuda.default_stream(eng.device):
                if eng._stream is None:
                    eng._stream = torch.cuda.Stream(device=eng.device)
                run = eng._stream
            else:
                run = cur
            if eng._run is not None and eng._run != run:
                run.wait_stream(eng._run)
            if run != cur:
                run.wait_stream(cur)
            eng._run, self.cur = run, cur
            return C.c_void_p(run.cuda_stream)

        def __exit__(self, *exc):
            if self.eng._run != self.cur:
                self.cur.wait_stream(self.eng._run)
            return False

    def stream(self):
        return Engine._On(self)

    def check(self, rc, what):
        _lib.check(rc, self.h, what)

    # ---- entry points
    def set_temperature(self, t):
        """``temperature`` of the reference's generate calls (engine state: rarely anything but 1.0)."""
        t = float(t)
        if t != self._temperature:
            self.check(self.lib.ivg_set_temperature(self.h, t), "set_temperature")
            self._temperature = t
        return self

    def set_decode_lds_kb(self, kb):
        """LDS budget of this engine's decode-step GEMMs (``ivg_config.decode_lds_kb``; 0 = process default, 16 .. 160 KiB)."""
        self.check(self.lib.ivg_set_decode_lds_kb(self.h, int(kb or 0)), "set_decode_lds_kb")
        return self

    def set_context_length(self, k):
        self.check(self.lib.ivg_set_context_length(self.h, int(k)), "set_context_length")

    def tokenize(self, pixels, ids, labels):
        B, T = pixels.shape[:2]
        with self.stream() as s:
            self.check(self.lib.ivg_tokenize(self.h, _ptr(pixels), dtype_code(pixels.dtype), B, T, _ptr(ids), _ptr(labels), s), "tokenize")
            for t in (pixels, ids, labels):
                if t is not None:
                    t.record_stream(self._run)

    def encode_context(self, pixels, ids):
        B, T = pixels.shape[:2]
        with self.stream() as s:
            self.check(self.lib.ivg_encode_context(self.h, _ptr(pixels), dtype_code(pixels.dtype), B, T, _ptr(ids), ids.stride(0), s),
                       "encode_context")
            pixels.record_stream(self._run); ids.record_stream(self._run)

    def detokenize(self, ids, F, out, cache=None, cache_mode=0, clamp=False):
        if clamp != self._clamp_out:   # clamp(0, 1) in the epilogue of the decoders' last convolution (engine state, rarely toggled)
            self.check(self.lib.ivg_set_output_clamp(self.h, int(bool(clamp))), "set_output_clamp")
            self._clamp_out = bool(clamp)
        with self.stream() as s:
            self.check(self.lib.ivg_detokenize_to(self.h, _ptr(ids), ids.shape[0], int(F), _ptr(out), dtype_code(out.dtype), cache, int(cache_mode), s),
                       "detokenize")
            ids.record_stream(self._run); out.record_stream(self._run)

    def detokenize_shared(self, ids, group_size, F, out, clamp=False):
        """ivg_detokenize_shared: rows of ``ids`` / ``out`` in groups of ``group_size`` consecutive trajectories with the same context tokens."""
        if clamp != self._clamp_out:
            self.check(self.lib.ivg_set_output_clamp(self.h, int(bool(clamp))), "set_output_clamp")
            self._clamp_out = bool(clamp)
        with self.stream() as s:
            self.check(self.lib.ivg_detokenize_shared(self.h, _ptr(ids), ids.shape[0] // int(group_size), int(group_size), int(F), _ptr(out),
                                                      dtype_code(out.dtype), s), "detokenize_shared")
            ids.record_stream(self._run); out.record_stream(self._run)

    def cache_create(self, B):
        h = C.c_void_p()
        self.check(self.lib.ivg_cache_create(self.h, int(B), C.byref(h)), "cache_create")
        self._caches.add(h.value)
        return h

    def cache_destroy(self, h):
        if self.h is not None and h.value in self._caches:   # (already released when the engine was closed first)
            self._caches.discard(h.value)
            self.lib.ivg_cache_destroy(self.h, h)

    def generate(self, prompt, n_new, out, actions=None, ctx=1, uniforms=None, top_k=100, reward=None, reuse_kv=False):
        B, L0 = prompt.shape
        act_T = actions.shape[1] if actions is not None else 0
        fn = self.lib.ivg_generate_continue if reuse_kv else self.lib.ivg_generate
        with self.stream() as s:
            self.check(fn(self.h, _ptr(prompt), prompt.stride(0), B, L0, int(n_new), _ptr(actions), act_T, int(ctx),
                                             _ptr(uniforms), int(top_k), _ptr(out), _ptr(reward), s), "generate")
            for t in (prompt, out, actions, uniforms, reward):
                if t is not None:
                    t.record_stream(self._run)

    def generate_shared(self, prompts, group_size, n_new, out, actions=None, ctx=1, uniforms=None, top_k=100, reward=None, force_sdf=False):
        """ivg_generate_shared: ``prompts`` (n_groups, L0), one row per group of ``group_size`` consecutive trajectories; actions /
        uniforms / out / reward have n_groups * group_size rows (row g * group_size + k = sample k of prompt g)."""
        n_groups, L0 = prompts.shape
        act_T = actions.shape[1] if actions is not None else 0
        with self.stream() as s:
            self.check(self.lib.ivg_generate_shared(self.h, _ptr(prompts), prompts.stride(0), n_groups, int(group_size), L0, int(n_new), _ptr(actions),
                                                    act_T, int(ctx), _ptr(uniforms), int(top_k), int(bool(force_sdf)), _ptr(out), _ptr(reward), s),
                       "generate_shared")
            for t in (prompts, out, actions, uniforms, reward):
                if t is not None:
                    t.record_stream(self._run)

    def generate_forced_sdf(self, prompt, n_new, out, ctx=1, uniforms=None, top_k=100):
        B, L0 = prompt.shape
        with self.stream() as s:
            self.check(self.lib.ivg_generate_forced_sdf(self.h, _ptr(prompt), prompt.stride(0), B, L0, int(n_new), int(ctx), _ptr(uniforms),
                                                        int(top_k), _ptr(out), s), "generate_forced_sdf")
            for t in (prompt, out, uniforms):
                if t is not None:
                    t.record_stream(self._run)

    def embed_tokens(self, ids, out):
        B, L = ids.shape
        with self.stream() as s:
            self.check(self.lib.ivg_embed_tokens(self.h, _ptr(ids), ids.stride(0), B, L, _ptr(out), s), "embed_tokens")
            ids.record_stream(self._run); out.record_stream(self._run)

    def action_linear(self, actions, out):
        with self.stream() as s:
            self.check(self.lib.ivg_action_linear(self.h, _ptr(actions), actions.numel() // actions.shape[-1], _ptr(out), s), "action_linear")
            actions.record_stream(self._run); out.record_stream(self._run)

    def reward_linear(self, hidden, out):
        with self.stream() as s:
            self.check(self.lib.ivg_reward_linear(self.h, _ptr(hidden), hidden.numel() // hidden.shape[-1], _ptr(out), s), "reward_linear")
            hidden.record_stream(self._run); out.record_stream(self._run)

    def generate_embeds(self, embeds, n_new, out, hidden=None, uniforms=None, top_k=100, allow_reuse=True):
        """-> True when the kept KV cache was reused (only the last row of ``embeds`` was fed)."""
        B, L0 = embeds.shape[:2]
        reused = C.c_int(0)
        with self.stream() as s:
            self.check(self.lib.ivg_generate_embeds(self.h, _ptr(embeds), B, L0, int(n_new), _ptr(uniforms), int(top_k), _ptr(out),
                                                    _ptr(hidden), int(bool(allow_reuse)), C.byref(reused), s), "generate_embeds")
            for t in (embeds, out, hidden, uniforms):
                if t is not None:
                    t.record_stream(self._run)
        return bool(reused.value)

    def logits(self, ids, out, actions=None, ctx=1):
        B, L = ids.shape
        act_T = actions.shape[1] if actions is not None else 0
        with self.stream() as s:
            self.check(self.lib.ivg_logits(self.h, _ptr(ids), B, L, _ptr(actions), act_T, int(ctx), _ptr(out), s), "logits")
            for t in (ids, out, actions):
                if t is not None:
                    t.record_stream(self._run)

    def eval_forward(self, ids, labels, token_nll, loss_rows, actions=None, ctx=1, hidden=None):
        B, L = ids.shape
        act_T = actions.shape[1] if actions is not None else 0
        with self.stream() as s:
            self.check(self.lib.ivg_eval_forward(self.h, _ptr(ids), _ptr(labels), B, L, _ptr(actions), act_T, int(ctx), _ptr(token_nll),
                                                 _ptr(loss_rows), _ptr(hidden), s), "eval_forward")
            for t in (ids, labels, token_nll, loss_rows, actions, hidden):
                if t is not None:
                    t.record_stream(self._run)

    def action_recon_sqerr(self, hidden, actions, ctx, prelude, out):
        B, L = hidden.shape[:2]
        with self.stream() as s:
            self.check(self.lib.ivg_action_recon_sqerr(self.h, _ptr(hidden), _ptr(actions), B, L, actions.shape[1], int(ctx), int(prelude),
                                                       _ptr(out), s), "action_recon_sqerr")
            for t in (hidden, actions, out):
                t.record_stream(self._run)

    def profile_enable(self, kclass, on=True):
        self.check(self.lib.ivg_profile_enable(self.h, kclass, int(on)), "profile_enable")

    def profile_attn_fit(self):
        """(fixed microseconds per launch, streaming GB/s) of the decode attention, after profile_read(IVG_K_DECODE_ATTN)."""
        f, r = C.c_double(0), C.c_double(0)
        self.check(self.lib.ivg_profile_attn_fit(self.h, C.byref(f), C.byref(r)), "profile_attn_fit")
        return f.value, r.value

    def profile_gemm_kinds(self):
        """{kind: (mean launch window in us, launches)} of the decode-step GEMMs, after profile_read(IVG_K_DECODE_GEMM)."""
        us, n = (C.c_double * 5)(), (C.c_int64 * 5)()
        self.check(self.lib.ivg_profile_gemm_kinds(self.h, us, n), "profile_gemm_kinds")
        return {k: (us[i], n[i]) for i, k in enumerate(("qkv", "o_proj", "gate_up", "down", "lm_head"))}

    def profile_read(self, kclass):
        st = _lib.IvgProfileStats()
        self.check(self.lib.ivg_profile_read(self.h, kclass, C.byref(st)), "profile_read")
        return dict(launches=st.launches, total_ms=st.total_ms, total_flops=st.total_flops, total_bytes=st.total_bytes)
