"""Drop-in mirror of the reference's ``ivideogpt.vq_model.CompressiveVQModel`` inference API
(/root/reference/ivideogpt/vq_model/compressive_vq_model.py:33-277) on the MI355X engine.

Same names, argument meaning and error behaviour as the reference class:
``from_pretrained`` (diffusers checkpoint layout), ``.to(device)``, ``.eval()``, ``.config[...]``,
``.context_length``, ``.num_vq_embeddings``, ``.num_dyn_embeddings``, ``set_context_length``,
``tokenize(pixel_values, context_length) -> (indices, labels)``,
``detokenize(indices, context_length, cache=None, return_cache=False)``.
``encode_context`` is the one addition: the context-only path every prediction caller wants
(predict.py:53-54 tokenizes the whole clip and then drops all future-frame tokens).

Only tensor plumbing and checkpoint I/O happen here; every FLOP is in libivg (HIP).
"""
import torch

from . import weights as W
from .engine import Engine
from .packing import pack_tokenizer

CTX_TOKENS = 16 * 16 + 1   # 256 context tokens + separator   (compressive_vq_model.py:225)
DYN_TOKENS = 4 * 4 + 1     # 16 dynamics tokens + separator


class DetokenizeCache:
    """Opaque cache handed back by ``detokenize(..., return_cache=True)`` (mbrl/video_predictor.py:320-321).
    Unlike the reference's dict (valid only for F == 1 calls) it holds the un-repeated
    per-trajectory context features, so it can be reused with any number of future frames."""

    def __init__(self, engine, B):
        self.engine, self.B = engine, B
        self.handle = engine.cache_create(B)   # the engine owns the device memory: Engine.close releases what is still alive

    def __del__(self):
        try:
            self.engine.cache_destroy(self.handle)
        except Exception:
            pass


class CompressiveVQModel:
    supports_shared_context = True   # generate / detokenize accept shared_context= (libivg ivg_generate_shared / ivg_detokenize_shared)
    def __init__(self, config=None, state_dict=None, encode_dtype="fp32", decode_dtype="bf16", **kwargs):
        cfg = dict(config or {})
        cfg.update(kwargs)
        self.config = W.tokenizer_config(**{k: v for k, v in cfg.items() if k in W.TOKENIZER_DEFAULTS})
        self.context_length = self.config["context_length"]
        self.num_vq_embeddings = self.config["num_vq_embeddings"]
        self.num_dyn_embeddings = self.config["num_dyn_embeddings"]
        self.patch_size = self.config["patch_size"]
        self.latent_channels = self.config["latent_channels"]
        self.vq_embed_dim = self.config["vq_embed_dim"]
        self._pretrained_context = self.context_length
        self._sd = state_dict
        self.encode_dtype, self.decode_dtype = encode_dtype, decode_dtype
        self.device = torch.device("cpu")
        self._engine = None
        # the packed weights in HBM, kept across engine rebuilds and shared by replicas; valid for one (weights version, device,
        # dtypes) -- the version is bumped by load_state_dict (never id(dict): it survives in-place edits and is reused after gc)
        self._packed, self._packed_key, self._sd_version = None, None, 0
        self.training = False

    def _pack_key(self):
        return (self._sd_version, str(self.device), self.encode_dtype, self.decode_dtype, self._pretrained_context)

    def _packed_weights(self, cfg):
        from .packing import dtype_code, is_x3
        if self._packed is None or self._packed_key != self._pack_key():
            self._packed = pack_tokenizer(self._sd, cfg, self.device, dtype_code(self.encode_dtype), dtype_code(self.decode_dtype),
                                          dec_x3=is_x3(self.decode_dtype))
            self._packed_key = self._pack_key()
        return self._packed

    def replica(self):
        """A second tokenizer object over the SAME weights in HBM (the engine only reads them), with its own engine (workspace,
        detokenize caches): a second batch in flight on another stream / host thread (bench.py --lanes)."""
        if self.device.type != "cuda":
            raise RuntimeError("replica(): call .to('cuda') first")
        cfg = dict(self.config)
        cfg["context_length"] = self._pretrained_context
        r = CompressiveVQModel(cfg, self._sd, encode_dtype=self.encode_dtype, decode_dtype=self.decode_dtype)
        r.device = self.device
        r._sd_version = self._sd_version
        r._packed, r._packed_key = self._packed_weights(cfg), self._pack_key()
        if self.context_length != self._pretrained_context:
            r.set_context_length(self.context_length)
        return r

    # ------------------------------------------------------------------ construction / plumbing
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, subfolder=None, low_cpu_mem_usage=False, encode_dtype="fp32",
                        decode_dtype="bf16", **unused):
        cfg, sd = W.load_tokenizer_checkpoint(pretrained_model_name_or_path, subfolder)
        return cls(cfg, sd, encode_dtype=encode_dtype, decode_dtype=decode_dtype)

    @classmethod
    def from_config(cls, config, seed=0, codebook_std=None, **kw):
        """Seeded random weights in the real checkpoint schema (no pretrained files are available offline).  ``config``: a dict, or --
        as the reference's MBRL loader passes it (mbrl/video_predictor.py:43: ``CompressiveVQModel.from_config(path)``) -- a
        directory / file holding ``config.json``."""
        if isinstance(config, (str, bytes)) or hasattr(config, "__fspath__"):
            import json
            import os
            p = os.fspath(config)
            with open(os.path.join(p, "config.json") if os.path.isdir(p) else p) as f:
                config = json.load(f)
        cfg = W.tokenizer_config(**{k: v for k, v in dict(config).items() if k in W.TOKENIZER_DEFAULTS})
        return cls(cfg, W.random_tokenizer_state_dict(cfg, seed, codebook_std), **kw)

    def state_dict(self):
        return self._sd

    def load_state_dict(self, sd, strict=True):
        sd = W.remap_legacy_attention_keys(sd)
        if strict:
            W.validate_state_dict(sd, W.tokenizer_param_shapes(self.config), "tokenizer")
        self._sd = sd
        self._drop_engine()
        self._sd_version += 1
        self._packed, self._packed_key = None, None

    def save_pretrained(self, path, subfolder=None):
        cfg = dict(self.config)
        cfg["context_length"] = self._pretrained_context
        W.save_tokenizer_checkpoint(path, cfg, self._sd, subfolder)

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

    def requires_grad_(self, flag=False):
        return self

    def _drop_engine(self):
        if self._engine is not None:
            self._engine.close()
        self._engine = None

    def _ensure(self, B, T):
        e = self._engine
        if e is not None and B <= e.max_batch and T <= e.max_frames:
            return e
        if self.device.type != "cuda":
            raise RuntimeError("CompressiveVQModel: call .to('cuda') first -- the engine runs on an MI355X only (no CPU path)")
        if self._sd is None:
            raise RuntimeError("CompressiveVQModel has no weights: use from_pretrained / from_config / load_state_dict")
        cap_b = max(B, e.max_batch if e else 0)
        cap_t = max(T, e.max_frames if e else 0)
        self._drop_engine()
        cfg = dict(self.config)
        cfg["context_length"] = self._pretrained_context
        self._engine = Engine(self.device, self._packed_weights(cfg), tok_cfg=cfg, encode_dtype=self.encode_dtype, decode_dtype=self.decode_dtype,
                              max_batch=cap_b, max_frames=cap_t)
        if self.context_length != self._pretrained_context:
            self._engine.set_context_length(self.context_length)
        return self._engine

    def set_context_length(self, context_length):
        """compressive_vq_model.py:154-158 (cross-attention keeps the LAST k frames of kv_pos_emb)."""
        if not 1 <= context_length <= self._pretrained_context:
            raise AssertionError("context_length must be in [1, pretrained context_length]")
        self.context_length = context_length
        self.config["context_length"] = context_length
        if self._engine is not None:
            self._engine.set_context_length(context_length)

    # ------------------------------------------------------------------ the hot path
    def _pixels(self, pixel_values):
        if pixel_values.dim() != 5 or pixel_values.shape[2] != 3:
            raise AssertionError("pixel_values must be (B, T, 3, H, W)")
        res = self.config["resolution"]
        if pixel_values.shape[-1] != res or pixel_values.shape[-2] != res:
            raise AssertionError(f"pixel_values must be {res}x{res}")
        px = pixel_values.to(self.device)
        if px.dtype not in (torch.float32, torch.bfloat16):
            px = px.float()
        return px.contiguous()

    @torch.no_grad()
    def tokenize(self, pixel_values, context_length=0):
        assert context_length == self.context_length  # same contract as the reference (:166)
        px = self._pixels(pixel_values)
        B, T = px.shape[:2]
        assert T >= context_length + 1, "tokenize needs at least one future frame (callers pad with zero frames)"
        L = CTX_TOKENS * context_length - 1 + DYN_TOKENS * (T - context_length)
        ids = torch.empty(B, L, dtype=torch.int64, device=self.device)
        labels = torch.empty(B, L, dtype=torch.int64, device=self.device)
        self._ensure(B, T).tokenize(px, ids, labels)
        return ids, labels

    @torch.no_grad()
    def encode_context(self, pixel_values, context_length=0):
        """(B, >=ctx, 3, H, W) -> int64 (B, 257*ctx): context tokens, scf separators and the trailing sdf --
        exactly ``tokenize(...)[0][:, :257*ctx]`` without encoding any future frame."""
        assert context_length == self.context_length
        px = self._pixels(pixel_values)
        B, T = px.shape[:2]
        assert T >= context_length
        ids = torch.empty(B, CTX_TOKENS * context_length, dtype=torch.int64, device=self.device)
        self._ensure(B, max(T, context_length + 1)).encode_context(px, ids)
        return ids

    @torch.no_grad()
    def detokenize(self, indices, context_length=0, cache=None, return_cache=False, clamp=False, out_dtype=torch.float32, shared_context=None):
        """``shared_context`` (not in the reference's signature; round 6): ``t`` or ``"auto"`` when the rows of ``indices`` are the t
        samples of B / t clips in ``repeat(t, 1)`` order (row k * B0 + b = sample k of clip b: what predict.py:65-72 and
        train_gpt.py:170-184 produce, VP2's candidates with B0 = 1) -- their CONTEXT tokens are identical, so the context frames are
        decoded once per clip and the predicted frames' cross-attention K / V are projected once per clip (libivg
        ``ivg_detokenize_shared``); same pixels as the plain call.
        ``clamp=True`` (not in the reference's signature): the frames come back as ``clamp(0, 1)`` -- the post-processing every
        caller applies (predict.py:73) -- written by the epilogue of the decoders' last convolution instead of a pass over the clip.
        ``out_dtype=torch.bfloat16`` (bf16 decode mode only): the clip in bfloat16, as the reference returns it under
        ``torch.autocast(bfloat16)`` (vp/ivideogpt_interface.py:180, mbrl/video_predictor.py:269) -- half the bytes."""
        assert context_length == self.context_length
        assert (indices.shape[1] + 1 - CTX_TOKENS * context_length) % DYN_TOKENS == 0
        F = (indices.shape[1] + 1 - CTX_TOKENS * context_length) // DYN_TOKENS
        B = indices.shape[0]
        ids = indices.to(device=self.device, dtype=torch.int64).contiguous()
        res = self.config["resolution"]
        out = torch.empty(B, context_length + F, 3, res, res, dtype=out_dtype, device=self.device)
        eng = self._ensure(B, context_length + F)
        if shared_context and cache is None and not return_cache and F > 0:
            from .transformer import _from_group_major, _to_group_major, shared_prompt_groups
            t, B0 = shared_prompt_groups(ids[:, :CTX_TOKENS * context_length - 1], shared_context)
            if t > 1:
                eng.detokenize_shared(_to_group_major(ids, t, B0), t, F, out, clamp=clamp)
                return _from_group_major(out, t, B0)
        handle, mode = None, 0
        if cache is not None and (cache.engine is not eng or cache.engine.h is None):
            # the engine was rebuilt since the cache was filled (batch or clip length grew): the cached context features are
            # gone with it -- decode the context again into a fresh cache instead of failing mid-rollout
            if cache.B != B:
                raise AssertionError("detokenize: cache was filled for another batch size")
            cache = DetokenizeCache(eng, B)
            handle, mode = cache.handle, 1
        elif cache is not None:
            if cache.B != B:
                raise AssertionError("detokenize: cache was filled for another batch size")
            handle, mode = cache.handle, 2
        elif return_cache:
            cache = DetokenizeCache(eng, B)
            handle, mode = cache.handle, 1
        eng.detokenize(ids, F, out, handle, mode, clamp=clamp)
        return (out, cache) if return_cache else out

    # ------------------------------------------------------------------ measurement
    def profile(self, kclass, on=True):
        self._engine.profile_enable(kclass, on)

    def profile_read(self, kclass):
        return self._engine.profile_read(kclass)
