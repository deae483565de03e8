#!/usr/bin/env python
"""Drop-in for the reference's ``inference/predict.py`` (same flags, same flow: /root/reference/inference/predict.py:25-122)
on the MI355X engine.  Differences: the context tokens come from ``encode_context`` (the reference tokenizes all frames and
drops the future tokens, :53-54); predictions are saved as ``pred-samples.npz`` (uint8 ``[repeat, T, H, 2W, 3]``: ground
truth | prediction side by side) and, when ``imageio`` is installed, as the reference's GIFs."""
import argparse
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ivideogpt_amd import CompressiveVQModel, HeadModelWithAction, LlamaForCausalLM, weights as W  # noqa: E402
from ivideogpt_amd.data import NPZParser  # noqa: E402

device = 'cuda'


def set_seed(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)


def parse_args(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument('--pretrained_model_name_or_path', type=str, required=True, help="path to pretrained model")
    p.add_argument('--input_path', type=str, required=True, help="path to input npz file")
    p.add_argument('--dataset_name', type=str, required=True, help="dataset name")
    p.add_argument('--output_path', type=str, default='outputs', help="path to save predicted video")
    p.add_argument("--context_length", type=int, default=2, help="number of init context frames")
    p.add_argument("--segment_length", type=int, default=16, help="number of frames in total, including context and future frames")
    p.add_argument('--resolution', type=int, default=64, help="resolution of frames")
    p.add_argument('--goal_conditioned', default=False, action='store_true', help="goal-conditioned prediction")
    p.add_argument('--action_conditioned', default=False, action='store_true', help="action-conditioned prediction")
    p.add_argument('--action_dim', default=4, type=int)
    p.add_argument('--repeat_times', default=5, type=int, help="number of times to repeat prediction")
    p.add_argument("--seed", type=int, default=0, help="random seed")
    p.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"], help="arithmetic of decode + rollout (tokenize is always fp32)")
    return p.parse_args(argv)


@torch.no_grad()
def predict(args, tokenizer, model, input, actions=None):
    pixel_values = input.to(device, non_blocking=True).unsqueeze(0)
    actions = actions.to(device, non_blocking=True).unsqueeze(0) if actions is not None else None
    gen_input = tokenizer.encode_context(pixel_values, args.context_length)           # == tokenize(...)[0][:, :ctx*257]
    max_new_tokens = (1 + 4 * 4) * (args.segment_length - args.context_length) - 1
    generated_tokens = model.generate(
        gen_input.repeat(args.repeat_times, 1), do_sample=True, temperature=1.0, top_k=100, max_new_tokens=max_new_tokens,
        pad_token_id=50256, **({'action': actions.repeat(args.repeat_times, 1, 1)} if actions is not None else {}))
    recon_output = tokenizer.detokenize(generated_tokens, args.context_length).clamp(0.0, 1.0)
    os.makedirs(args.output_path, exist_ok=True)
    gt = (pixel_values[0].permute(0, 2, 3, 1).float().cpu().numpy() * 255).astype(np.uint8)
    rec = (recon_output.permute(0, 1, 3, 4, 2).cpu().numpy() * 255).astype(np.uint8)
    frames = np.concatenate([np.broadcast_to(gt[None], rec.shape), rec], axis=3)
    np.savez_compressed(os.path.join(args.output_path, "pred-samples.npz"), frames=frames, tokens=generated_tokens.cpu().numpy())
    try:
        import imageio
        for j in range(args.repeat_times):
            imageio.mimsave(f"{args.output_path}/pred-samples-{j}.gif", list(frames[j]), fps=4, loop=0)
    except ImportError:
        pass
    return recon_output


def load_models(args):
    dt = args.dtype
    tokenizer = CompressiveVQModel.from_pretrained(args.pretrained_model_name_or_path, subfolder='tokenizer', low_cpu_mem_usage=False,
                                                   decode_dtype=dt).to(device)
    assert args.context_length == tokenizer.context_length
    if args.action_conditioned:
        cfg, sd = W.load_transformer_checkpoint(args.pretrained_model_name_or_path, 'transformer')
        prelude_tokens_num, tokens_per_dyna = (256 + 1) * args.context_length - 1, 16
        model = HeadModelWithAction(LlamaForCausalLM.from_config(cfg, dtype=dt), action_dim=args.action_dim,
                                    prelude_tokens_num=prelude_tokens_num, tokens_num_per_dyna=tokens_per_dyna,
                                    context=args.context_length, segment_length=args.segment_length).to(device)
        model.load_state_dict(sd, strict=True)
        assert model.llm.config.vocab_size == tokenizer.num_vq_embeddings + tokenizer.num_dyn_embeddings + 2
    else:
        model = LlamaForCausalLM.from_pretrained(args.pretrained_model_name_or_path, subfolder='transformer', dtype=dt).to(device)
        assert model.config.vocab_size == tokenizer.num_vq_embeddings + tokenizer.num_dyn_embeddings + 2
    return tokenizer, model


def main(argv=None):
    args = parse_args(argv)
    if args.seed is not None:
        set_seed(args.seed)
    assert not (args.goal_conditioned and args.action_conditioned), "Cannot be both goal and action conditioned"
    tokenizer, model = load_models(args)
    input, actions = NPZParser(args.segment_length, args.resolution).parse(args.input_path, args.dataset_name, load_action=args.action_conditioned)
    if args.goal_conditioned:
        input = torch.concat([input[-1:], input[:-1]], dim=0)
    return predict(args, tokenizer, model, input, actions)


if __name__ == "__main__":
    main()
