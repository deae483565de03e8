"""Inference half of the reference's MBRL world model (/root/reference/mbrl/video_predictor.py): construction from the hydra
``world_model`` block -- ``get_tokenizer`` (:40-56), ``load_models`` (:59-89), ``VideoPredictor(device, args)`` (:100-110; what
``mbrl/train_metaworld_mbpo.py:41-42`` calls) -- and ``VideoPredictor.rollout`` (:267-339).
Model / tokenizer TRAINING (``update_*``, the optimisers and LPIPS of the constructor, :112-265) is out of scope.

The rollout runs the reference's own per-step op sequence against the mirror objects, at the embeddings level (:286-317):
``get_input_embeddings`` of the context tokens once; then per environment step ``action_linear(action)`` added to the last
embedding (the step's ``sdf`` slot), ``llm.generate(inputs_embeds=..., max_new_tokens=17, return_dict_in_generate=True,
output_hidden_states=True)``, reward = ``reward_linear(hidden_states[-1][-1])``, the 16 predicted tokens + a forced ``sdf``
embedded and appended, the new frame decoded with the detokenizer cache and pushed onto the 3-frame stack.  What differs from
the reference is inside the engine: from the second step on ``generate`` recognises (device-side comparison with the inputs it
kept) that the KV cache already holds everything but the last embedding and feeds only that row -- 17 cached decode steps per
environment step instead of a prefill of the grown prompt."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

TOKENS_PER_DYN = 16


def symexp(x):
    return torch.sign(x) * (torch.exp(torch.abs(x)) - 1)


def _arg(args, name, default=None):
    """hydra DictConfig, argparse Namespace or plain dict."""
    if isinstance(args, dict):
        return args.get(name, default)
    return getattr(args, name, default) if not hasattr(args, "get") else args.get(name, default)


def get_tokenizer(args):
    """mbrl/video_predictor.py:40-56: the compressive tokenizer from ``pretrained_model_name_or_path`` (random weights of that
    config when ``load_pretrained_model`` is off), ``set_context_length`` with the reference's warning when the checkpoint's context
    length differs from ``args.context_length``; -> (tokenizer, vocab_size = context codes + dynamics codes + 2 special tokens)."""
    from ivideogpt_amd import CompressiveVQModel
    if _arg(args, "vqgan_type") != "ctx_vqgan":
        raise NotImplementedError
    path = _arg(args, "pretrained_model_name_or_path")
    dt = dict(encode_dtype=_arg(args, "encode_dtype", "fp32"), decode_dtype=_arg(args, "decode_dtype", "bf16"))   # (the reference runs under bf16 autocast, :269)
    if not _arg(args, "load_pretrained_model"):
        vq_model = CompressiveVQModel.from_config(path, **dt)
    else:
        vq_model = CompressiveVQModel.from_pretrained(path, subfolder=None, revision=None, variant=None, use_safetensor=True,
                                                      low_cpu_mem_usage=False, device_map=None, **dt)
    if _arg(args, "context_length") != vq_model.context_length:
        print(f"[Warning] pretrained context length of vq_model mismatch, change from {vq_model.context_length} to {_arg(args, 'context_length')}")
        vq_model.set_context_length(_arg(args, "context_length"))
    return vq_model, vq_model.num_vq_embeddings + vq_model.num_dyn_embeddings + 2


def load_models(args):
    """mbrl/video_predictor.py:59-89: tokenizer + ``HeadModelWithAction(AutoModelForCausalLM.from_config(config), action_dim,
    prelude = 257 * context - 1, 16 tokens per frame, context, segment_length, model_type = parent directory of config_name,
    reward_prediction=True)``; with ``load_pretrained_model`` the transformer weights come from
    ``pretrained_transformer_path/model.safetensors`` -- into ``model.llm`` only when ``load_internal_llm`` (an action-free
    pretrained transformer under freshly initialised action / reward heads), else into the whole wrapper, strictly.
    (``llama_attn_drop`` configures training-time dropout: no effect on the inference path, accepted and ignored.)"""
    from safetensors.torch import load_file
    from ivideogpt_amd import HeadModelWithAction, LlamaForCausalLM
    from ivideogpt_amd import weights as W
    tokenizer, vocab_size = get_tokenizer(args)
    config_name = _arg(args, "config_name")
    assert config_name, "world_model.config_name is required"
    config = W.load_llama_config(config_name)
    config["vocab_size"] = vocab_size
    load = bool(_arg(args, "load_pretrained_model"))
    model = LlamaForCausalLM.from_config(config, seed=None if load else _arg(args, "seed", 0), dtype=_arg(args, "llm_dtype", "bf16"))
    ctx = _arg(args, "context_length")
    model = HeadModelWithAction(model, action_dim=_arg(args, "action_dim"), prelude_tokens_num=(256 + 1) * ctx - 1, tokens_num_per_dyna=16,
                                context=ctx, segment_length=_arg(args, "segment_length"), model_type=os.path.normpath(config_name).split(os.sep)[-2],
                                reward_prediction=True)
    if load:
        state_dict = load_file(os.path.join(_arg(args, "pretrained_transformer_path"), "model.safetensors"))
        if _arg(args, "load_internal_llm"):
            model.llm.load_state_dict(state_dict, strict=True)
        else:
            model.load_state_dict(state_dict, strict=True)
    return model, tokenizer


class VideoPredictor:
    def __init__(self, device, args=None, reuse_cache=True):
        """``VideoPredictor('cuda', cfg.world_model)`` -- the reference's constructor (:100-110): models built by ``load_models(args)``
        and moved to ``device``.  ``reuse_cache=False`` forces a prefill of the whole prompt at every step (the reference's behaviour;
        A/B and tests).  ``VideoPredictor.from_models(tokenizer, model, ...)`` wraps objects that already exist."""
        self.args, self.device = args, torch.device(device)
        self.model, self.tokenizer = load_models(args)
        self.model = self.model.to(self.device)
        self.tokenizer = self.tokenizer.to(self.device)
        self.context_length, self.symlog, self.reuse_cache = _arg(args, "context_length"), bool(_arg(args, "symlog", True)), reuse_cache
        self.steps_with_kept_cache = 0

    @classmethod
    def from_models(cls, tokenizer, model, context_length=2, symlog=True, device="cuda", reuse_cache=True):
        """tokenizer: ivideogpt_amd.CompressiveVQModel; model: ivideogpt_amd.HeadModelWithAction(reward_prediction=True)."""
        self = cls.__new__(cls)
        self.args = None
        self.tokenizer, self.model, self.device = tokenizer, model, torch.device(device)
        self.context_length, self.symlog, self.reuse_cache = context_length, symlog, reuse_cache
        self.steps_with_kept_cache = 0
        return self

    @torch.no_grad()
    def rollout(self, obs, policy, horizon):
        """obs [B, 9, H, W] in 0..255 (3 stacked RGB frames); policy(obs, t) -> [B, A].
        -> (obss [B, horizon+1, 9, H, W], actions [B, horizon+1, A], rewards [B, horizon+1, 1])"""
        ctx, model, llm = self.context_length, self.model, self.model.llm
        B = obs.shape[0]
        obs = obs.to(self.device).float() / 255.
        first_obs = obs
        stack = list(torch.chunk(obs, 3, dim=1))                               # frame_stack = 3
        prompt = self.tokenizer.encode_context(torch.stack(stack[-ctx:], dim=1), ctx)   # [B, 257*ctx], ends with the first sdf
        embeds = model.get_input_embeddings(prompt)
        sdf_col = torch.full((B, 1), model.token_for_sdf, dtype=prompt.dtype, device=self.device)
        cache, trace = None, {"obs": [], "act": [], "rew": []}
        self.steps_with_kept_cache = 0
        for t in range(horizon):
            action = policy(obs, t).to(self.device).float()
            embeds[:, -1] += model.action_linear(action)                       # this step's sdf slot carries the action
            result = llm.generate(inputs_embeds=embeds, do_sample=True, temperature=1.0, top_k=100, pad_token_id=50256,
                                  use_cache=self.reuse_cache, max_new_tokens=TOKENS_PER_DYN + 1, return_dict_in_generate=True,
                                  output_hidden_states=True)
            self.steps_with_kept_cache += int(llm.last_generate_reused_cache)
            predicted = result.sequences[:, :-1]                               # the 17th token is replaced by the forced sdf
            reward = model.reward_linear(result.hidden_states[-1][-1]).squeeze(-2)   # last layer, last forward pass
            embeds = torch.cat([embeds, model.get_input_embeddings(torch.cat([predicted, sdf_col], 1))], 1)
            fmap, cache = self.tokenizer.detokenize(torch.cat([prompt, predicted], 1), ctx, cache=cache, return_cache=True)
            stack = stack[1:] + [fmap.clamp(0.0, 1.0)[:, -1]]
            obs = torch.cat(stack, dim=1)
            trace["obs"].append(obs); trace["act"].append(action); trace["rew"].append(reward)
        # dummy step 0: the initial observation with a zero action / reward
        obss = torch.stack([first_obs] + trace["obs"], 1).float()
        actions = torch.stack([torch.zeros_like(trace["act"][0])] + trace["act"], 1).float()
        rewards = [torch.zeros_like(trace["rew"][0])] + trace["rew"]
        if self.symlog:
            rewards = [symexp(r) for r in rewards]
        return obss, actions, torch.stack(rewards, 1).float()
