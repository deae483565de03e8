"""Inference half of the reference's MBRL world model (/root/reference/mbrl/video_predictor.py:267-339): ``VideoPredictor.rollout``.
Model / tokenizer TRAINING (``update_*``, :152-265) is out of scope.

Step t: action_t is added to the embedding of the current last token (the t-th ``sdf`` slot, :295-296), 16 dynamics tokens are
sampled, reward = ``reward_linear`` of the last layer's hidden state at the last generation step (:311-313), the predicted
tokens plus a forced ``sdf`` extend the sequence (:315-317), the new frame is decoded with the detokenizer cache (:320-321) and
pushed onto the 3-frame stack (:323-325).  The engine keeps tokens (not embeddings); from the second step on it also keeps the KV
cache of the previous step (``reuse_cache``), so a step costs 17 cached decode steps instead of a prefill of the grown prompt
(the reference re-runs the whole prefix through ``llm.generate(inputs_embeds=...)`` every step)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def symexp(x):
    return torch.sign(x) * (torch.exp(torch.abs(x)) - 1)


class VideoPredictor:
    def __init__(self, tokenizer, model, context_length=2, symlog=True, device="cuda", reuse_cache=True):
        """tokenizer: ivideogpt_amd.CompressiveVQModel; model: ivideogpt_amd.HeadModelWithAction(reward_prediction=True)."""
        self.tokenizer, self.model, self.device = tokenizer, model, torch.device(device)
        self.context_length, self.symlog, self.reuse_cache = context_length, symlog, reuse_cache

    @torch.no_grad()
    def rollout(self, obs, policy, horizon):
        """obs [B, 9, H, W] in 0..255 (3 stacked RGB frames); policy(obs, t) -> [B, A].
        -> (obss [B, horizon+1, 9, H, W], actions [B, horizon+1, A], rewards [B, horizon+1, 1])"""
        ctx = self.context_length
        B = obs.shape[0]
        obs = obs.to(self.device).float() / 255.
        init_obs = obs
        frames = list(torch.chunk(obs, 3, dim=1))                              # frame_stack = 3
        context = torch.stack(frames[-ctx:], dim=1)
        tokens = self.tokenizer.encode_context(context, ctx)                   # [B, 257*ctx] incl. the trailing sdf
        init_tokens, cache = tokens, None
        sdf = self.model.token_for_sdf
        obss, actions, rewards = [], [], []
        act = None
        for t in range(horizon):
            action = policy(obs, t).to(self.device).float()
            if act is None:  # fixed-size action table: slot i of the sequence reads row i + ctx - 1 (+1 never-fed row at the end)
                act = torch.zeros(B, ctx - 1 + horizon + 1, action.shape[-1], device=self.device)
            act[:, ctx - 1 + t] = action
            kw = dict(do_sample=True, temperature=1.0, top_k=100, max_new_tokens=17, pad_token_id=50256, action=act, return_reward=True)
            try:
                out, reward = self.model.generate(tokens, reuse_cache=self.reuse_cache and t > 0, **kw)
            except AssertionError:    # the policy (or anyone else) used the transformer in between: prefill again
                out, reward = self.model.generate(tokens, **kw)
            predicted = out[:, tokens.shape[1]:tokens.shape[1] + 16]
            tokens = torch.cat([tokens, predicted, torch.full((B, 1), sdf, dtype=tokens.dtype, device=self.device)], 1)
            fmap, cache = self.tokenizer.detokenize(torch.cat([init_tokens, predicted], 1), ctx, cache=cache, return_cache=True)
            fmap = fmap.clamp(0.0, 1.0)
            frames.append(fmap[:, -1])
            frames.pop(0)
            obs = torch.cat(frames, dim=1)
            obss.append(obs); actions.append(action); rewards.append(reward[:, None])
        obss = [init_obs] + obss                                               # dummy step
        actions = [torch.zeros_like(actions[0])] + actions
        rewards = [torch.zeros_like(rewards[0])] + rewards
        if self.symlog:
            rewards = [symexp(r) for r in rewards]
        return torch.stack(obss, 1).float(), torch.stack(actions, 1).float(), torch.stack(rewards, 1).float()
