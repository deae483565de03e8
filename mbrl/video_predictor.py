"""Inference half of the reference's MBRL world model (/root/reference/mbrl/video_predictor.py:267-339): ``VideoPredictor.rollout``.
Model / tokenizer TRAINING (``update_*``, :152-265) is out of scope.

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


class VideoPredictor:
    def __init__(self, tokenizer, model, context_length=2, symlog=True, device="cuda", reuse_cache=True):
        """tokenizer: ivideogpt_amd.CompressiveVQModel; model: ivideogpt_amd.HeadModelWithAction(reward_prediction=True).
        reuse_cache=False forces a prefill of the whole prompt at every step (the reference's behaviour; A/B and tests)."""
        self.tokenizer, self.model, self.device = tokenizer, model, torch.device(device)
        self.context_length, self.symlog, self.reuse_cache = context_length, symlog, reuse_cache
        self.steps_with_kept_cache = 0

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
