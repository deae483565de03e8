"""Drop-in mirrors of the transformer-side objects the reference's callers use:

  * ``LlamaForCausalLM``   -- what ``AutoModelForCausalLM.from_pretrained(path, subfolder='transformer')``
    returns in inference/predict.py:111-113 (``.generate(input_ids, do_sample, temperature, top_k,
    max_new_tokens, pad_token_id)`` -> prompt + new tokens; ``.config.vocab_size``).
  * ``HeadModelWithAction`` -- /root/reference/ivideogpt/transformer/action_model.py:8-121, same
    constructor signature, ``load_state_dict(load_file(...), strict=True)``, ``generate(..., action=...)``,
    ``token_for_sdf``.

Sampling: ``torch.multinomial`` streams are device- and version-specific even inside the reference, so the
engine draws by inverse CDF from explicit uniforms (``torch.rand`` on the model's device, default or
supplied generator) over the top-k kept tokens in ascending id order; ``do_sample=False`` is greedy argmax.
All compute is in libivg (HIP); this file is tensor plumbing and checkpoint I/O.
"""
from types import SimpleNamespace

import torch

from . import weights as W
from .engine import Engine
from .packing import dtype_code, pack_llama, torch_dtype


class _Embedding:
    """What ``llm.get_input_embeddings()`` returns (an ``nn.Embedding`` in HF): callable on int64 ids -> (…, hidden) rows of
    ``model.embed_tokens.weight`` in the engine's transformer dtype (libivg ``ivg_embed_tokens``)."""

    def __init__(self, llm):
        self._llm = llm

    def __call__(self, input_ids):
        llm = self._llm
        ids = input_ids.to(device=llm.device, dtype=torch.int64)
        shape = ids.shape
        ids2 = ids.reshape(-1, shape[-1]).contiguous() if ids.dim() > 1 else ids.reshape(1, -1).contiguous()
        out = torch.empty(*ids2.shape, llm._cfg["hidden_size"], dtype=llm.torch_dtype, device=llm.device)
        llm._ensure(ids2.shape[0]).embed_tokens(ids2, out)
        return out.reshape(*shape, -1)

    forward = __call__


class _ActionLinear:
    """``HeadModelWithAction.action_linear`` (action_model.py:36): float (…, action_dim) -> (…, hidden), transformer dtype."""

    def __init__(self, llm):
        self._llm = llm

    def __call__(self, action):
        llm = self._llm
        a = action.to(device=llm.device, dtype=torch.float32).contiguous()
        out = torch.empty(*a.shape[:-1], llm._cfg["hidden_size"], dtype=llm.torch_dtype, device=llm.device)
        llm._ensure(max(1, a.shape[0] if a.dim() > 1 else 1)).action_linear(a, out)
        return out

    forward = __call__


class _RewardLinear:
    """``HeadModelWithAction.reward_linear`` (action_model.py:41): post-norm hidden states (…, hidden) -> float32 (…, 1)."""

    def __init__(self, llm):
        self._llm = llm

    def __call__(self, hidden):
        llm = self._llm
        h = hidden.to(device=llm.device, dtype=llm.torch_dtype).contiguous()
        out = torch.empty(*h.shape[:-1], 1, dtype=torch.float32, device=llm.device)
        llm._ensure(max(1, h.shape[0] if h.dim() > 1 else 1)).reward_linear(h, out)
        return out

    forward = __call__


def shared_prompt_groups(ids, repeat_times):
    """The reference's multi-sample callers build their prompts as ``gen_input.repeat(t, 1)`` (inference/predict.py:65,
    train_gpt.py:170,179): row ``k * B0 + b`` is sample k of prompt b.  ``repeat_times``:
      * an int t > 1 -- the caller says so; checked on the device (one comparison kernel + a host read: a generate call is >= 100 ms);
      * ``"auto"`` -- detect it: the largest t dividing B for which rows b, b + B0, ... are equal (1: nothing shared).
    -> (t, B0).  Raises ValueError when an explicit t does not describe ``ids``."""
    B = ids.shape[0]
    if repeat_times == "auto":
        for t in range(B, 1, -1):
            if B % t == 0 and torch.equal(ids.reshape(t, B // t, -1), ids[:B // t].unsqueeze(0).expand(t, -1, -1)):
                return t, B // t
        return 1, B
    t = int(repeat_times)
    if t <= 1:
        return 1, B
    if B % t != 0 or not torch.equal(ids.reshape(t, B // t, -1), ids[:B // t].unsqueeze(0).expand(t, -1, -1)):
        raise ValueError(f"shared_context={t}: input_ids is not `prompts.repeat({t}, 1)` (rows b, b + B/{t}, ... must be identical)")
    return t, B // t


def _to_group_major(x, t, B0):
    """rows (k * B0 + b) -> rows (b * t + k): the engine keeps the t samples of a prompt in consecutive rows"""
    if x is None or t == 1 or B0 == 1:
        return x
    return x.view(t, B0, *x.shape[1:]).transpose(0, 1).reshape(x.shape).contiguous()


def _from_group_major(x, t, B0):
    if x is None or t == 1 or B0 == 1:
        return x
    return x.view(B0, t, *x.shape[1:]).transpose(0, 1).reshape(x.shape).contiguous()


class LlamaForCausalLM:
    supports_shared_context = True   # generate / detokenize accept shared_context= (libivg ivg_generate_shared / ivg_detokenize_shared)
    def __init__(self, config, state_dict=None, dtype="bf16", prefix="", action_dim=None, reward_prediction=False, decode_lds_kb=0):
        self._decode_lds_kb = int(decode_lds_kb or 0)   # launch policy of THIS model's engine (set_decode_lds_kb)
        self._cfg = dict(W.LLAMA_SMALL)
        self._cfg.update({k: v for k, v in dict(config).items() if k in self._cfg})
        self.config = SimpleNamespace(**self._cfg)
        self.config.n_embd = self._cfg["hidden_size"]
        self._sd, self._prefix = state_dict, prefix
        self._action_dim, self._reward = action_dim, reward_prediction
        self.dtype = dtype
        self.torch_dtype = torch_dtype(dtype_code(dtype))
        self.device = torch.device("cpu")
        self._engine = None
        # the packed weights in HBM, kept across engine rebuilds and shared by replicas.  Validity is an explicit version counter
        # bumped by every load_state_dict (never id(dict): a reloaded or in-place mutated dict keeps its id, a collected one's is reused)
        self._packed, self._packed_key, self._sd_version = None, None, 0

    def _pack_key(self):
        return (self._sd_version, self._prefix, str(self.device), self.dtype)

    def _invalidate_pack(self):
        """Called whenever the weights, their key prefix or the device change: the old pack is released with the engine that used it."""
        self._sd_version += 1
        self._packed, self._packed_key = None, None

    def _packed_weights(self):
        if self._packed is None or self._packed_key != self._pack_key():
            self._packed = pack_llama(self._sd, self._cfg, self.device, dtype_code(self.dtype), prefix=self._prefix)
            self._packed_key = self._pack_key()
        return self._packed

    def replica(self):
        """A second model object over the SAME weights in HBM (the engine only reads them): its own engine -- KV cache, workspace --
        for a second batch in flight on another stream / host thread (bench.py --lanes; INTEGRATION.md, streams)."""
        if self.device.type != "cuda":
            raise RuntimeError("replica(): call .to('cuda') first")
        r = LlamaForCausalLM(self._cfg, self._sd, dtype=self.dtype, prefix=self._prefix, action_dim=self._action_dim,
                             reward_prediction=self._reward, decode_lds_kb=self._decode_lds_kb)
        r.device = self.device
        r._sd_version = self._sd_version
        if hasattr(self, "_wrapper_heads"):
            r._wrapper_heads = self._wrapper_heads
        r._packed, r._packed_key = self._packed_weights(), self._pack_key()
        return r

    # HF keyword arguments of from_config / from_pretrained that have no meaning for an inference engine (eval semantics, one
    # attention implementation, local files only): accepted and ignored -- anything else raises TypeError
    _IGNORED_HF_KWARGS = frozenset({"trust_remote_code", "attn_implementation", "torch_dtype", "attention_dropout", "use_cache", "revision",
                                    "cache_dir", "local_files_only", "device_map", "use_safetensors", "token"})

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder="transformer", low_cpu_mem_usage=False, dtype="bf16",
                        **unused):
        cfg, sd = W.load_transformer_checkpoint(pretrained_model_name_or_path, subfolder)
        W.validate_state_dict(sd, W.llama_param_shapes(cfg), "transformer")
        return cls(cfg, sd, dtype=dtype)

    @classmethod
    def from_config(cls, config, seed=None, dtype="bf16", **hf_kwargs):
        """``AutoModelForCausalLM.from_config(config)`` (mbrl/video_predictor.py:72, train_gpt.py:593): ``config`` = a dict, an object
        with the HF field names, or a path to a ``config.json`` / its directory (what ``AutoConfig.from_pretrained`` takes).
        seed = None: no weights yet (load_state_dict follows); otherwise seeded random weights in the checkpoint schema.
        Of HF's keyword arguments only the ones with no meaning here are accepted (and ignored): a misspelt name raises."""
        unknown = set(hf_kwargs) - cls._IGNORED_HF_KWARGS
        if unknown:
            raise TypeError(f"from_config() got unexpected keyword argument(s) {sorted(unknown)}")
        if isinstance(config, (str, bytes)) or hasattr(config, "__fspath__"):
            config = W.load_llama_config(config)
        cfg = dict(W.LLAMA_SMALL)
        cfg.update({k: v for k, v in (vars(config) if not isinstance(config, dict) else config).items() if k in cfg})
        sd = W.random_llama_state_dict(cfg, seed) if seed is not None else None
        return cls(cfg, sd, dtype=dtype)

    def state_dict(self):
        return self._sd

    def load_state_dict(self, sd, strict=True):
        if strict:
            W.validate_state_dict(sd, W.llama_param_shapes(self._cfg), "transformer")
        heads = getattr(self, "_wrapper_heads", None)
        if heads is not None:   # ``model.llm.load_state_dict(...)`` of a HeadModelWithAction (mbrl/video_predictor.py:82-83, load_internal_llm):
            # only the transformer's weights change, the wrapper's heads stay what they are
            old = self._sd if (self._sd is not None and self._prefix == "llm.") else None
            self._sd = HeadModelWithAction._with_fresh_heads(sd, self._cfg["hidden_size"], heads[0], heads[1], old=old)
            self._prefix = "llm."
        else:
            self._sd, self._prefix = sd, ""
        self._drop_engine()
        self._invalidate_pack()

    def save_pretrained(self, path, subfolder="transformer"):
        W.save_transformer_checkpoint(path, self._cfg, self._sd, subfolder)

    def to(self, device=None, *a, **k):
        if device is not None and not isinstance(device, torch.dtype):
            dev = torch.device(device)
            if dev.type == "cuda" and dev.index is None:
                dev = torch.device("cuda", torch.cuda.current_device())
            if dev != self.device:
                self.device = dev
                self._drop_engine()
                self._packed, self._packed_key = None, None   # the pack lives on the old device
        return self

    def cuda(self, index=None):
        return self.to(torch.device("cuda", index if index is not None else torch.cuda.current_device()))

    def eval(self):
        return self

    def _drop_engine(self):
        if self._engine is not None:
            self._engine.close()
        self._engine = None

    def _ensure(self, B, frames=32):
        e = self._engine
        if e is not None and B <= e.max_batch and frames <= e.max_frames:
            return e
        if self.device.type != "cuda":
            raise RuntimeError("call .to('cuda') first -- the engine runs on an MI355X only (no CPU path)")
        if self._sd is None:
            raise RuntimeError("model has no weights: use from_pretrained / load_state_dict")
        cap_b = max(B, e.max_batch if e else 0)
        cap_t = max(frames, e.max_frames if e else 0)
        self._drop_engine()
        self._engine = Engine(self.device, self._packed_weights(), llm_cfg=self._cfg, action_dim=self._action_dim or 0,
                              reward_head=self._reward, llm_dtype=self.dtype, max_batch=cap_b, max_frames=cap_t,
                              decode_lds_kb=self._decode_lds_kb)
        return self._engine

    # LDS budget (KiB) of a decode-step GEMM workgroup for a model whose batch shares the GPU with other batches in flight (bench.py
    # --lanes, INTEGRATION.md "streams"): with a whole CU's LDS per workgroup (the default, fastest for one batch alone) the decode
    # GEMMs of one batch lock the other batches' kernels out of the CU for their whole duration; at <= 52 KiB three or four
    # workgroups of different engines fit, and four batches in flight reach 5,680 instead of 5,350 predicted frames/s
    # (profiles/r04_lanes.txt).  A property of the ENGINE (ivg_config.decode_lds_kb), not of the process: a latency-bound model
    # (MBRL step-wise rollout) and throughput lanes can live side by side.
    BATCHES_IN_FLIGHT_LDS_KB = 40

    def set_decode_lds_kb(self, kb):
        """``kb`` = 0: the process default (``IVG_DECODE_LDS_KB``, 160); 16 .. 160 otherwise.  Applies to the live engine and to every
        engine this model builds later; the budget picks the kernel generation, so tokens of two budgets are each deterministic
        but not bit-comparable with one another."""
        self._decode_lds_kb = int(kb or 0)
        if self._engine is not None:
            self._engine.set_decode_lds_kb(self._decode_lds_kb)
        return self

    # ------------------------------------------------------------------ hot path
    def _uniforms(self, B, n, do_sample, generator):
        if not do_sample:
            return None
        return torch.rand(B, n, device=self.device, dtype=torch.float32, generator=generator)

    def get_input_embeddings(self):
        return _Embedding(self)

    @torch.no_grad()
    def generate(self, input_ids=None, do_sample=True, temperature=1.0, top_k=100, max_new_tokens=None, pad_token_id=None,
                 generator=None, uniforms=None, inputs_embeds=None, return_dict_in_generate=False, output_hidden_states=False,
                 use_cache=True, shared_context=None, **unused):
        """``input_ids`` prompt -> int64 (B, L0 + max_new_tokens), prompt included (HF convention).
        ``shared_context`` (not in HF; round 6): ``t`` or ``"auto"`` when ``input_ids`` is ``prompts.repeat(t, 1)`` -- what
        inference/predict.py:65 and train_gpt.py:170-184 pass.  The prompt is then prefilled ONCE per distinct row, its K / V rows are kept
        once and shared by the t samples at every decode step (libivg ``ivg_generate_shared``); row order, uniforms and results are those
        of the plain call (same tokens up to near-ties of the sampler: the prompt's last position goes through the decode-step kernels).
        ``inputs_embeds`` prompt (B, L0, hidden) -> only the new tokens (B, max_new_tokens), as HF does for embeddings prompts
        (action_model.py:101-110, mbrl/video_predictor.py:298-313).  With ``return_dict_in_generate`` the result has
        ``.sequences`` and, with ``output_hidden_states``, ``.hidden_states`` of which only what the callers read exists:
        ``hidden_states[-1][-1]`` = last layer (post final norm) of the LAST forward pass, (B, 1, hidden).
        When the engine's KV cache was built from exactly ``inputs_embeds[:, :-1]`` (the step-wise rollout: previous prompt +
        the embeddings of the tokens it generated), only the last row is fed -- verified on the device, never assumed."""
        if not (isinstance(temperature, (int, float)) and temperature > 0):   # HF's TemperatureLogitsWarper raises the same way
            raise ValueError(f"`temperature` (={temperature}) has to be a strictly positive float")
        if inputs_embeds is not None:
            emb = inputs_embeds.to(device=self.device, dtype=self.torch_dtype).contiguous()
            B, L0, _ = emb.shape
            out = torch.empty(B, max_new_tokens, dtype=torch.int64, device=self.device)
            hidden = torch.empty(B, 1, emb.shape[-1], dtype=self.torch_dtype, device=self.device) if output_hidden_states else None
            u = uniforms if uniforms is not None else self._uniforms(B, max_new_tokens, do_sample, generator)
            self.last_generate_reused_cache = self._ensure(B).set_temperature(temperature).generate_embeds(
                emb, max_new_tokens, out, hidden=hidden, uniforms=u, top_k=top_k or self._cfg["vocab_size"], allow_reuse=use_cache)
            if not return_dict_in_generate:
                return out
            return SimpleNamespace(sequences=out, hidden_states=((hidden,),) if output_hidden_states else None)
        ids = input_ids.to(device=self.device, dtype=torch.int64).contiguous()
        B, L0 = ids.shape
        out = torch.empty(B, L0 + max_new_tokens, dtype=torch.int64, device=self.device)
        u = uniforms if uniforms is not None else self._uniforms(B, max_new_tokens, do_sample, generator)
        t, B0 = shared_prompt_groups(ids, shared_context) if shared_context else (1, B)
        if t > 1 and L0 >= 2:
            self._ensure(B).set_temperature(temperature).generate_shared(ids[:B0].contiguous(), t, max_new_tokens, out, uniforms=_to_group_major(u, t, B0),
                                                                         top_k=top_k or self._cfg["vocab_size"])
            return _from_group_major(out, t, B0)
        self._ensure(B).set_temperature(temperature).generate(ids, max_new_tokens, out, uniforms=u, top_k=top_k or self._cfg["vocab_size"])
        return out

    @torch.no_grad()
    def logits(self, input_ids):
        """Teacher-forced logits, float32 (B, L, vocab)  (``model(input_ids).logits`` in the reference)."""
        ids = input_ids.to(device=self.device, dtype=torch.int64).contiguous()
        B, L = ids.shape
        out = torch.empty(B, L, self._cfg["vocab_size"], dtype=torch.float32, device=self.device)
        self._ensure(B).logits(ids, out)
        return out

    @torch.no_grad()
    def _eval_forward(self, input_ids, labels, action=None, ctx=1, want_hidden=False, frames=32):
        """-> namespace(loss, token_nll (B, L), sample_loss (B), sample_perplexity (B), hidden_states or None): HF shifted
        cross-entropy (ignore_index -100; mean over the batch's valid targets) computed by libivg ``ivg_eval_forward`` without the
        (B, L, vocab) logits tensor."""
        ids = input_ids.to(device=self.device, dtype=torch.int64).contiguous()
        lab = labels.to(device=self.device, dtype=torch.int64).contiguous()
        B, L = ids.shape
        nll = torch.empty(B, L, dtype=torch.float32, device=self.device)
        rows = torch.empty(B, 2, dtype=torch.float32, device=self.device)
        hidden = torch.empty(B, L, self._cfg["hidden_size"], dtype=self.torch_dtype, device=self.device) if want_hidden else None
        act = action.to(device=self.device, dtype=torch.float32).contiguous() if action is not None else None
        self._ensure(B, frames).eval_forward(ids, lab, nll, rows, actions=act, ctx=ctx, hidden=hidden)
        sums, counts = rows[:, 0], rows[:, 1]
        per = sums / counts.clamp_min(1.0)
        return SimpleNamespace(loss=sums.sum() / counts.sum().clamp_min(1.0), token_nll=nll, sample_loss=per,
                               sample_perplexity=torch.exp(per), hidden_states=(hidden,) if want_hidden else None, logits=None)

    def __call__(self, input_ids=None, labels=None, output_hidden_states=False, **unused):
        """``model(input_ids=tokens, labels=labels)`` of the eval loop (train_gpt.py:356-376): ``.loss`` (+ per-sample loss /
        perplexity); without labels: ``.logits`` (B, L, vocab) fp32."""
        if labels is None:
            return SimpleNamespace(logits=self.logits(input_ids), loss=None)
        return self._eval_forward(input_ids, labels, want_hidden=output_hidden_states)

    forward = __call__


class HeadModelWithAction:
    """action_model.py:8-45: wraps an ``llm`` and adds ``action_linear`` (+ optional ``reward_linear``)."""
    supports_shared_context = True   # generate / detokenize accept shared_context= (libivg ivg_generate_shared / ivg_detokenize_shared)

    def __init__(self, llm, action_dim, prelude_tokens_num, tokens_num_per_dyna, context, segment_length, model_type="llama",
                 reward_prediction=False, action_recon=None, **kwargs):
        if model_type != "llama":
            raise ValueError(f"model_type {model_type} is not supported.")
        self.llm = llm
        self.action_dim = action_dim
        self.prelude_tokens_num = prelude_tokens_num
        self.tokens_num_per_dyna = tokens_num_per_dyna
        self.context = context
        self.segment_length = segment_length
        self.model_type = model_type
        self.token_for_sdf = llm.config.vocab_size - 1
        self.reward_prediction = reward_prediction
        self.action_recon = action_recon
        if (llm._action_dim, llm._reward, llm._prefix) != (action_dim, reward_prediction, "llm."):
            if llm._sd is not None and llm._prefix == "":
                # wrapping an llm that already holds weights (AutoModelForCausalLM.from_config / from_pretrained, then
                # HeadModelWithAction(model, ...): mbrl/video_predictor.py:74-79): its keys move under "llm." and the new heads get the
                # reference constructor's initial values (action_model.py:36-42)
                llm._sd = self._with_fresh_heads(llm._sd, llm._cfg["hidden_size"], action_dim, reward_prediction)
            llm._action_dim, llm._reward, llm._prefix = action_dim, reward_prediction, "llm."
            llm._invalidate_pack()   # other key prefix / extra heads: whatever was packed for the bare llm is stale
        llm._wrapper_heads = (action_dim, reward_prediction)   # llm.load_state_dict(...) under this wrapper keeps the heads (load_internal_llm)
        llm._drop_engine()   # an engine built for the bare llm has no action / reward head
        self.device = llm.device
        self.action_linear = _ActionLinear(llm)
        if reward_prediction:
            self.reward_linear = _RewardLinear(llm)

    @staticmethod
    def _with_fresh_heads(llm_sd, hidden, action_dim, reward_prediction, old=None, seed=0):
        """{"llm." + k: v} plus the heads: kept from ``old`` (a previous wrapper state dict) when present, else as the reference's
        constructor leaves them -- ``action_linear`` zero-initialised (action_model.py:36-39), ``reward_linear`` with nn.Linear's
        default initialiser (:41-42; seeded here, the reference draws from the global generator)."""
        sd = {"llm." + k: v for k, v in llm_sd.items()}
        old = old or {}
        heads = {"action_linear.weight": torch.zeros(hidden, action_dim), "action_linear.bias": torch.zeros(hidden)}
        if reward_prediction:
            lin = torch.nn.Linear(hidden, 1)
            g = torch.Generator().manual_seed(seed)
            bound = 1.0 / hidden ** 0.5
            with torch.no_grad():
                lin.weight.uniform_(-bound, bound, generator=g)
                lin.bias.uniform_(-bound, bound, generator=g)
            heads["reward_linear.weight"], heads["reward_linear.bias"] = lin.weight.detach(), lin.bias.detach()
        for k, v in heads.items():
            sd[k] = old.get(k, v)
        for k, v in old.items():
            if k.startswith("action_recon_linear"):
                sd[k] = v
        return sd

    def get_input_embeddings(self, input_ids):
        """action_model.py:47-54."""
        return self.llm.get_input_embeddings()(input_ids)

    def set_decode_lds_kb(self, kb):
        self.llm.set_decode_lds_kb(kb)
        return self

    def replica(self):
        """As LlamaForCausalLM.replica: a second wrapper (own engine) over the same weights in HBM."""
        llm = self.llm.replica()   # (carries action_dim / reward / prefix and the shared pack: the wrapper below changes none of them)
        return HeadModelWithAction(llm, self.action_dim, self.prelude_tokens_num, self.tokens_num_per_dyna, self.context, self.segment_length,
                                   model_type=self.model_type, reward_prediction=self.reward_prediction, action_recon=self.action_recon)

    def load_state_dict(self, sd, strict=True):
        if strict:
            W.validate_state_dict({k: v for k, v in sd.items() if not k.startswith("action_recon_linear")},
                                  W.llama_param_shapes(self.llm._cfg, self.action_dim, self.reward_prediction), "HeadModelWithAction")
        self.llm._sd, self.llm._prefix = sd, "llm."
        self.llm._drop_engine()
        self.llm._invalidate_pack()

    def state_dict(self):
        return self.llm._sd

    def to(self, device=None, *a, **k):
        self.llm.to(device)
        self.device = self.llm.device
        return self

    def eval(self):
        return self

    @torch.no_grad()
    def generate(self, inputs_token, do_sample=True, temperature=1.0, top_k=100, max_new_tokens=None, pad_token_id=50256,
                 action=None, generator=None, uniforms=None, return_reward=False, reuse_cache=False, shared_context=None):
        """action_model.py:56-121: action (B, T, D); new token j is the forced sdf when j % 17 == 0; the i-th sdf slot's
        embedding gets ``action_linear(action[:, i + context - 1])``.  -> int64 (B, L0 + max_new_tokens).
        ``shared_context``: as ``LlamaForCausalLM.generate`` -- ``inputs_token`` is ``prompts.repeat(t, 1)`` (train_gpt.py:170, VP2's
        candidate action sequences over one context: vp/ivideogpt_interface.py:155-202); the ACTIONS stay per row (the shared prefix ends
        before the first action slot).
        ``reuse_cache=True`` (step-wise rollouts, mbrl/video_predictor.py:286-317): the prompt is the previous call's full
        output plus the forced ``sdf``; the engine keeps the KV cache of that call and feeds only the last prompt token
        instead of prefilling the grown prompt again (raises AssertionError when the cache holds something else)."""
        if not (isinstance(temperature, (int, float)) and temperature > 0):
            raise ValueError(f"`temperature` (={temperature}) has to be a strictly positive float")
        llm = self.llm
        ids = inputs_token.to(device=llm.device, dtype=torch.int64).contiguous()
        B, L0 = ids.shape
        act = action.to(device=llm.device, dtype=torch.float32).contiguous()
        out = torch.empty(B, L0 + max_new_tokens, dtype=torch.int64, device=llm.device)
        u = uniforms if uniforms is not None else llm._uniforms(B, max_new_tokens, do_sample, generator)
        reward = torch.empty(B, dtype=torch.float32, device=llm.device) if return_reward else None
        t, B0 = shared_prompt_groups(ids, shared_context) if (shared_context and not reuse_cache) else (1, B)
        if t > 1 and L0 == 257 * self.context:
            llm._ensure(B, act.shape[1]).set_temperature(temperature).generate_shared(
                ids[:B0].contiguous(), t, max_new_tokens, out, actions=_to_group_major(act, t, B0), ctx=self.context, uniforms=_to_group_major(u, t, B0),
                top_k=top_k or llm._cfg["vocab_size"], reward=reward)
            out, reward = _from_group_major(out, t, B0), _from_group_major(reward, t, B0)
            return (out, reward) if return_reward else out
        llm._ensure(B, act.shape[1]).set_temperature(temperature).generate(ids, max_new_tokens, out, actions=act, ctx=self.context, uniforms=u,
                                              top_k=top_k or llm._cfg["vocab_size"], reward=reward, reuse_kv=reuse_cache)
        return (out, reward) if return_reward else out

    @torch.no_grad()
    def generate_without_action(self, inputs_token, do_sample=True, temperature=1.0, top_k=100, max_new_tokens=None, generator=None,
                                uniforms=None):
        """action_model.py:123-152 (no caller in the reference): per future frame 16 sampled tokens, then the forced ``sdf`` -- the
        schedule of ``generate`` without any action embedding; the last forced ``sdf`` is dropped.  -> int64 (B, L0 + max_new_tokens).
        One prefill + cached steps instead of the reference's per-frame re-prefill (token-identical: same argument as ``generate``)."""
        if not (isinstance(temperature, (int, float)) and temperature > 0):
            raise ValueError(f"`temperature` (={temperature}) has to be a strictly positive float")
        llm = self.llm
        ids = inputs_token.to(device=llm.device, dtype=torch.int64).contiguous()
        B, L0 = ids.shape
        assert (max_new_tokens + 1) % (self.segment_length - self.context) == 0, "max_new_tokens must be (tokens_per_dyna + 1) * frames - 1"
        out = torch.empty(B, L0 + max_new_tokens, dtype=torch.int64, device=llm.device)
        u = uniforms if uniforms is not None else llm._uniforms(B, max_new_tokens, do_sample, generator)
        llm._ensure(B).set_temperature(temperature).generate_forced_sdf(ids, max_new_tokens, out, ctx=self.context, uniforms=u, top_k=top_k or llm._cfg["vocab_size"])
        return out

    @torch.no_grad()
    def __call__(self, input_ids=None, attention_mask=None, labels=None, position_ids=None, action=None):
        """``HeadModelWithAction.forward`` (action_model.py:154-205) as the eval loop calls it (train_gpt.py:356-376):
        ``x.loss`` = HF shifted cross-entropy (+ ``action_recon`` * MSE of the reconstructed actions, :187-196); with
        ``reward_prediction`` returns ``(x, reward_pred)`` with ``reward_pred`` (B, segment - context, 1) read from the hidden
        state of the last token of every predicted frame (:198-204).  No logits tensor is materialised (``x.logits`` is None;
        ``self.logits(ids, action)`` returns them when needed)."""
        assert attention_mask is None and position_ids is None, "the reference's callers never pass masks / position ids"
        llm = self.llm
        F = self.segment_length - self.context
        need_hidden = bool(self.reward_prediction or self.action_recon)
        if labels is None:
            return SimpleNamespace(logits=self.logits(input_ids, action), loss=None)
        x = llm._eval_forward(input_ids, labels, action=action, ctx=self.context, want_hidden=need_hidden, frames=action.shape[1])
        hidden = x.hidden_states[-1] if need_hidden else None
        if self.action_recon:
            B, L = hidden.shape[:2]
            act = action.to(device=llm.device, dtype=torch.float32).contiguous()
            err = torch.empty(B, dtype=torch.float32, device=llm.device)
            llm._engine.action_recon_sqerr(hidden, act, self.context, self.prelude_tokens_num, err)
            self.action_recon_loss = err.sum() / (B * (L - self.prelude_tokens_num) * self.action_dim)
            x.loss = x.loss + self.action_recon * self.action_recon_loss
        if self.reward_prediction:
            start = self.prelude_tokens_num + torch.arange(F, device=llm.device) * (self.tokens_num_per_dyna + 1)
            reward_pred = self.reward_linear(hidden[:, start + self.tokens_num_per_dyna])   # (B, F, 1)
            return x, reward_pred
        return x

    forward = __call__

    @torch.no_grad()
    def logits(self, input_ids, action):
        """Teacher-forced logits with the action embeddings added on every sdf slot (action_model.py:154-185)."""
        llm = self.llm
        ids = input_ids.to(device=llm.device, dtype=torch.int64).contiguous()
        act = action.to(device=llm.device, dtype=torch.float32).contiguous()
        out = torch.empty(ids.shape[0], ids.shape[1], llm._cfg["vocab_size"], dtype=torch.float32, device=llm.device)
        llm._ensure(ids.shape[0], act.shape[1]).logits(ids, out, actions=act, ctx=self.context)
        return out
