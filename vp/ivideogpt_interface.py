"""Drop-in for the reference's VP2 planner interface (/root/reference/vp/ivideogpt_interface.py:73-206):
``iVideoGPTPredictor(config_name, seed, vqgan_type, pretrained_vqgan_name_or_path, pretrained_transformer_path, action_dim,
generate_max_batchsize, decode_max_batchsize, action_recon, lora, ...)`` with ``num_context = 2``,
``base_prediction_modality = "rgb"``, ``close()`` and ``__call__({"video": [B,2,64,64,3], "actions": [B,>=11,A]}) ->
{"rgb": float32 ndarray [B,11,64,64,3]}``.  LoRA checkpoints are out of scope (fine-tuning feature)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ivideogpt_amd import CompressiveVQModel, HeadModelWithAction, LlamaForCausalLM, weights as W  # noqa: E402


class iVideoGPTPredictor:
    def __init__(self, config_name, seed, vqgan_type, pretrained_vqgan_name_or_path, pretrained_transformer_path, action_dim,
                 generate_max_batchsize, decode_max_batchsize, action_recon=None, lora=False, lora_r=8, lora_alpha=32,
                 lora_dropout=0.0, epoch=None, dtype="bf16"):
        assert vqgan_type == 'ctx_vqgan', "we only have CompressiveVQModel now"
        assert not lora, "LoRA checkpoints are not supported by the MI355X engine (fine-tuning feature, out of scope)"
        self.video_predictor_config = {"context_length": 2, "segment_length": 12, "generate_max_batchsize": generate_max_batchsize,
                                       "decode_max_batchsize": decode_max_batchsize}
        if seed is not None:
            torch.manual_seed(seed)
            np.random.seed(seed)
        self.num_context = 2                       # needed by vp2
        self.base_prediction_modality = "rgb"
        self.tokenizer = CompressiveVQModel.from_pretrained(pretrained_vqgan_name_or_path, subfolder=None, decode_dtype=dtype)
        if self.tokenizer.context_length != 2:
            self.tokenizer.set_context_length(2)   # vp/ivideogpt_interface.py:22-27
        cfg_path = config_name if os.path.isfile(config_name) else os.path.join(config_name, "config.json")
        with open(cfg_path) as f:
            raw = json.load(f)
        cfg = dict(W.LLAMA_SMALL)
        cfg.update({k: raw[k] for k in cfg if k in raw})
        cfg["vocab_size"] = self.tokenizer.num_vq_embeddings + self.tokenizer.num_dyn_embeddings + 2
        from safetensors.torch import load_file
        self.model = HeadModelWithAction(LlamaForCausalLM(cfg, None, dtype=dtype), action_dim=action_dim, prelude_tokens_num=257 * 2 - 1,
                                         tokens_num_per_dyna=16, context=2, segment_length=12, action_recon=action_recon)
        self.model.load_state_dict(load_file(os.path.join(pretrained_transformer_path, 'model.safetensors')), strict=True)
        self.model = self.model.to('cuda')
        self.tokenizer = self.tokenizer.to('cuda')

    def close(self):
        pass

    @torch.no_grad()
    def __call__(self, batch):
        ctx, T = 2, 12
        gmax, dmax = self.video_predictor_config["generate_max_batchsize"], self.video_predictor_config["decode_max_batchsize"]
        video = torch.as_tensor(batch["video"]).float().cuda()          # [B, 2, 64, 64, 3] in [0, 1]
        actions = torch.as_tensor(batch["actions"]).float().cuda()       # [B, >= T-1, A]
        B = video.shape[0]
        pixels = video.permute(0, 1, 4, 2, 3).contiguous()
        act = actions                                                      # slot i reads row i + ctx - 1 (rows 1..10), as in the reference
        outs = []
        for s in range(0, B, gmax):                                      # the reference chunks by generate_max_batchsize (:155-202)
            px, a = pixels[s:s + gmax], act[s:s + gmax]
            n = px.shape[0]
            # VP2's planner scores its candidate action sequences from ONE observation: the rows of `video` are copies of the same two
            # frames (vp/ivideogpt_interface.py:155-169 tokenizes every copy).  When they are, the context is encoded, prefilled and
            # decoded ONCE and its K / V rows are shared by all candidates (shared_context); the actions stay per row.
            same = n > 1 and bool((px == px[:1]).all())
            prompt = self.tokenizer.encode_context(px[:1], ctx).repeat(n, 1) if same else self.tokenizer.encode_context(px, ctx)
            tokens = self.model.generate(prompt, do_sample=True, temperature=1.0, top_k=100, max_new_tokens=17 * (T - ctx) - 1,
                                         pad_token_id=50256, action=a, shared_context=n if same else None)
            for d in range(0, tokens.shape[0], dmax):
                chunk = tokens[d:d + dmax]
                outs.append(self.tokenizer.detokenize(chunk, ctx, shared_context=chunk.shape[0] if same and chunk.shape[0] > 1 else None).clamp(0.0, 1.0))
        rec = torch.cat(outs, 0)[:, 1:]                                  # 11 frames: last context frame + 10 predictions (:199-205)
        return {"rgb": rec.permute(0, 1, 3, 4, 2).float().cpu().numpy()}
