#!/usr/bin/env python
"""Command-line video prediction on the MI355X engine: the drop-in for the reference's ``inference/predict.py``
(/root/reference/inference/predict.py:25-122 -- same flag names, same flow: load tokenizer + transformer, read one clip from
an ``.npz`` file, sample ``--repeat_times`` futures from the context frames, decode, save).

Differences from the reference script: the context tokens come from ``encode_context`` (the reference tokenizes every frame
and then drops the future tokens, :53-54); the result is saved as ``pred-samples.npz`` (uint8 ``[repeat, T, H, 2W, 3]``,
ground truth and prediction side by side, plus the token ids) and additionally as GIFs when ``imageio`` is importable."""
import argparse
import os
import random
import sys

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from ivideogpt_amd import CompressiveVQModel, HeadModelWithAction, LlamaForCausalLM, weights as W  # noqa: E402
from ivideogpt_amd.data import NPZParser  # noqa: E402

GPU = "cuda"
TOKENS_PER_FRAME = 4 * 4 + 1        # 16 dynamics tokens + the sdf separator
CTX_TOKENS_PER_FRAME = 16 * 16 + 1

# (flag, kwargs) -- names and defaults of the reference CLI (predict.py:76-91) + --dtype
CLI = [
    ("--pretrained_model_name_or_path", dict(type=str, required=True, help="checkpoint directory (tokenizer/ + transformer/)")),
    ("--input_path", dict(type=str, required=True, help=".npz clip")),
    ("--dataset_name", dict(type=str, required=True, help="selects the .npz key layout (ivideogpt_amd/data.py)")),
    ("--output_path", dict(type=str, default="outputs", help="directory for pred-samples.*")),
    ("--context_length", dict(type=int, default=2, help="frames the prediction is conditioned on")),
    ("--segment_length", dict(type=int, default=16, help="context + predicted frames")),
    ("--resolution", dict(type=int, default=64, help="frame side in pixels")),
    ("--goal_conditioned", dict(action="store_true", help="the clip's last frame is moved to the front as a goal image")),
    ("--action_conditioned", dict(action="store_true", help="HeadModelWithAction checkpoint + actions from the .npz")),
    ("--action_dim", dict(type=int, default=4, help="width of one action vector")),
    ("--repeat_times", dict(type=int, default=5, help="independent samples of the future")),
    ("--seed", dict(type=int, default=0, help="seeds python / numpy / torch")),
    ("--dtype", dict(default="bf16", choices=["bf16", "fp32"], help="arithmetic of decode + rollout (tokenize is always fp32)")),
]


def set_seed(seed):
    for seeder in (random.seed, np.random.seed, torch.manual_seed, torch.cuda.manual_seed_all):
        seeder(seed)


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    for flag, kw in CLI:
        parser.add_argument(flag, **kw)
    return parser.parse_args(argv)


def save_outputs(out_dir, clip, recon, tokens):
    """clip (T, 3, H, W) in [0, 1]; recon (R, T, 3, H, W); tokens (R, L)."""
    os.makedirs(out_dir, exist_ok=True)
    truth = (clip.permute(0, 2, 3, 1).float().cpu().numpy() * 255).astype(np.uint8)
    pred = (recon.permute(0, 1, 3, 4, 2).cpu().numpy() * 255).astype(np.uint8)
    side_by_side = np.concatenate([np.broadcast_to(truth[None], pred.shape), pred], axis=3)
    np.savez_compressed(os.path.join(out_dir, "pred-samples.npz"), frames=side_by_side, tokens=tokens.cpu().numpy())
    try:
        import imageio
    except ImportError:
        return
    for k, sample in enumerate(side_by_side):
        imageio.mimsave(os.path.join(out_dir, f"pred-samples-{k}.gif"), list(sample), fps=4, loop=0)


@torch.no_grad()
def predict(args, tokenizer, model, input, actions=None):
    """One clip -> ``repeat_times`` sampled continuations, decoded and clamped to [0, 1]  (predict.py:47-74)."""
    ctx, reps = args.context_length, args.repeat_times
    clip = input.to(GPU, non_blocking=True)[None]
    prompt = tokenizer.encode_context(clip, ctx).repeat(reps, 1)          # == tokenize(...)[0][:, :257 * ctx], repeated
    extra = {}
    if actions is not None:
        extra["action"] = actions.to(GPU, non_blocking=True)[None].repeat(reps, 1, 1)
    n_new = TOKENS_PER_FRAME * (args.segment_length - ctx) - 1
    # the `reps` rows share ONE context (predict.py:65 repeats it): prefilled / decoded once, its K / V rows kept once (shared_context)
    tokens = model.generate(prompt, do_sample=True, temperature=1.0, top_k=100, max_new_tokens=n_new, pad_token_id=50256, shared_context=reps,
                            **extra)
    recon = tokenizer.detokenize(tokens, ctx, shared_context=reps).clamp(0.0, 1.0)
    save_outputs(args.output_path, clip[0], recon, tokens)
    return recon


def load_models(args):
    root = args.pretrained_model_name_or_path
    tokenizer = CompressiveVQModel.from_pretrained(root, subfolder="tokenizer", low_cpu_mem_usage=False, decode_dtype=args.dtype).to(GPU)
    assert args.context_length == tokenizer.context_length                                       # predict.py:96
    vocab = tokenizer.num_vq_embeddings + tokenizer.num_dyn_embeddings + 2
    if not args.action_conditioned:
        model = LlamaForCausalLM.from_pretrained(root, subfolder="transformer", dtype=args.dtype).to(GPU)
        assert model.config.vocab_size == vocab                                                   # :113
        return tokenizer, model
    cfg, state = W.load_transformer_checkpoint(root, "transformer")
    model = HeadModelWithAction(LlamaForCausalLM.from_config(cfg, dtype=args.dtype), action_dim=args.action_dim,
                                prelude_tokens_num=CTX_TOKENS_PER_FRAME * args.context_length - 1, tokens_num_per_dyna=16,
                                context=args.context_length, segment_length=args.segment_length).to(GPU)
    model.load_state_dict(state, strict=True)                                                     # :108
    assert model.llm.config.vocab_size == vocab
    return tokenizer, model


def main(argv=None):
    args = parse_args(argv)
    assert not (args.goal_conditioned and args.action_conditioned), "goal- and action-conditioning are exclusive"
    if args.seed is not None:
        set_seed(args.seed)
    tokenizer, model = load_models(args)
    clip, actions = NPZParser(args.segment_length, args.resolution).parse(args.input_path, args.dataset_name,
                                                                           load_action=args.action_conditioned)
    if args.goal_conditioned:
        clip = torch.cat([clip[-1:], clip[:-1]], dim=0)
    return predict(args, tokenizer, model, clip, actions)


if __name__ == "__main__":
    main()
