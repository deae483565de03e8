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
from .packing import dtype_code, pack_llama


class LlamaForCausalLM:
    def __init__(self, config, state_dict=None, dtype="bf16", prefix="", action_dim=None, reward_prediction=False):
        self._cfg = dict(W.LLAMA_SMALL)
        self._cfg.update({k: v for k, v in dict(config).items() if k in self._cfg})
        self.config = SimpleNamespace(**self._cfg)
        self.config.n_embd = self._cfg["hidden_size"]
        self._sd, self._prefix = state_dict, prefix
        self._action_dim, self._reward = action_dim, reward_prediction
        self.dtype = dtype
        self.device = torch.device("cpu")
        self._engine = None

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder="transformer", low_cpu_mem_usage=False, dtype="bf16",
                        **unused):
        cfg, sd = W.load_transformer_checkpoint(pretrained_model_name_or_path, subfolder)
        W.validate_state_dict(sd, W.llama_param_shapes(cfg), "transformer")
        return cls(cfg, sd, dtype=dtype)

    @classmethod
    def from_config(cls, config, seed=None, dtype="bf16"):
        cfg = dict(W.LLAMA_SMALL)
        cfg.update({k: v for k, v in (vars(config) if not isinstance(config, dict) else config).items() if k in cfg})
        sd = W.random_llama_state_dict(cfg, seed) if seed is not None else None
        return cls(cfg, sd, dtype=dtype)

    def state_dict(self):
        return self._sd

    def load_state_dict(self, sd, strict=True):
        if strict:
            W.validate_state_dict(sd, W.llama_param_shapes(self._cfg), "transformer")
        self._sd, self._prefix = sd, ""
        self._drop_engine()

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
        tensors = pack_llama(self._sd, self._cfg, self.device, dtype_code(self.dtype), prefix=self._prefix)
        self._engine = Engine(self.device, tensors, llm_cfg=self._cfg, action_dim=self._action_dim or 0,
                              reward_head=self._reward, llm_dtype=self.dtype, max_batch=cap_b, max_frames=cap_t)
        return self._engine

    # ------------------------------------------------------------------ hot path
    def _uniforms(self, B, n, do_sample, generator):
        if not do_sample:
            return None
        return torch.rand(B, n, device=self.device, dtype=torch.float32, generator=generator)

    @torch.no_grad()
    def generate(self, input_ids=None, do_sample=True, temperature=1.0, top_k=100, max_new_tokens=None, pad_token_id=None,
                 generator=None, uniforms=None, **unused):
        """-> int64 (B, L0 + max_new_tokens), prompt included (HF convention for ``input_ids`` prompts)."""
        assert temperature == 1.0, "the reference always samples at temperature 1.0"
        ids = input_ids.to(device=self.device, dtype=torch.int64).contiguous()
        B, L0 = ids.shape
        out = torch.empty(B, L0 + max_new_tokens, dtype=torch.int64, device=self.device)
        u = uniforms if uniforms is not None else self._uniforms(B, max_new_tokens, do_sample, generator)
        self._ensure(B).generate(ids, max_new_tokens, out, uniforms=u, top_k=top_k or self._cfg["vocab_size"])
        return out

    @torch.no_grad()
    def logits(self, input_ids):
        """Teacher-forced logits, float32 (B, L, vocab)  (``model(input_ids).logits`` in the reference)."""
        ids = input_ids.to(device=self.device, dtype=torch.int64).contiguous()
        B, L = ids.shape
        out = torch.empty(B, L, self._cfg["vocab_size"], dtype=torch.float32, device=self.device)
        self._ensure(B).logits(ids, out)
        return out

    def __call__(self, input_ids=None, labels=None, **unused):
        lg = self.logits(input_ids)
        loss = None
        if labels is not None:  # HF shifted cross-entropy, ignore_index -100 (train_gpt.py:364-376)
            loss = torch.nn.functional.cross_entropy(lg[:, :-1].reshape(-1, lg.shape[-1]), labels[:, 1:].reshape(-1).to(lg.device),
                                                     ignore_index=-100)
        return SimpleNamespace(logits=lg, loss=loss)


class HeadModelWithAction:
    """action_model.py:8-45: wraps an ``llm`` and adds ``action_linear`` (+ optional ``reward_linear``)."""

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
        llm._action_dim, llm._reward, llm._prefix = action_dim, reward_prediction, "llm."
        self.device = llm.device

    def load_state_dict(self, sd, strict=True):
        if strict:
            W.validate_state_dict({k: v for k, v in sd.items() if not k.startswith("action_recon_linear")},
                                  W.llama_param_shapes(self.llm._cfg, self.action_dim, self.reward_prediction), "HeadModelWithAction")
        self.llm._sd, self.llm._prefix = sd, "llm."
        self.llm._drop_engine()

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
                 action=None, generator=None, uniforms=None, return_reward=False, reuse_cache=False):
        """action_model.py:56-121: action (B, T, D); new token j is the forced sdf when j % 17 == 0; the i-th sdf slot's
        embedding gets ``action_linear(action[:, i + context - 1])``.  -> int64 (B, L0 + max_new_tokens).
        ``reuse_cache=True`` (step-wise rollouts, mbrl/video_predictor.py:286-317): the prompt is the previous call's full
        output plus the forced ``sdf``; the engine keeps the KV cache of that call and feeds only the last prompt token
        instead of prefilling the grown prompt again (raises AssertionError when the cache holds something else)."""
        assert temperature == 1.0
        llm = self.llm
        ids = inputs_token.to(device=llm.device, dtype=torch.int64).contiguous()
        B, L0 = ids.shape
        act = action.to(device=llm.device, dtype=torch.float32).contiguous()
        out = torch.empty(B, L0 + max_new_tokens, dtype=torch.int64, device=llm.device)
        u = uniforms if uniforms is not None else llm._uniforms(B, max_new_tokens, do_sample, generator)
        reward = torch.empty(B, dtype=torch.float32, device=llm.device) if return_reward else None
        llm._ensure(B, act.shape[1]).generate(ids, max_new_tokens, out, actions=act, ctx=self.context, uniforms=u,
                                              top_k=top_k or llm._cfg["vocab_size"], reward=reward, reuse_kv=reuse_cache)
        return (out, reward) if return_reward else out

    @torch.no_grad()
    def logits(self, input_ids, action):
        """Teacher-forced logits with the action embeddings added on every sdf slot (action_model.py:154-185)."""
        llm = self.llm
        ids = input_ids.to(device=llm.device, dtype=torch.int64).contiguous()
        act = action.to(device=llm.device, dtype=torch.float32).contiguous()
        out = torch.empty(ids.shape[0], ids.shape[1], llm._cfg["vocab_size"], dtype=torch.float32, device=llm.device)
        llm._ensure(ids.shape[0], act.shape[1]).logits(ids, out, actions=act, ctx=self.context)
        return out
