"""Evaluation half of the reference's ``train_gpt.py`` on the MI355X engine (the training loop is out of scope):

    get_tokenizer             /root/reference/train_gpt.py:128-149
    generate_multiple_times   :152-191   t samples per trajectory, chunked by ``max_generate_batchsize``
    batch_forward             :194-195   chunked detokenize
    evaluate                  :321-512   full-clip tokenize -> ``model(**input).loss`` -> gather -> t x B repeated generation ->
                                         chunked detokenize -> clamp -> best-of-t frame metrics -> gather -> eval logs

Same names, arguments and op sequence as the reference, so the loop reads like the original; what differs:
  * ``accelerator`` is any object with ``device / num_processes / is_main_process / is_local_main_process / gather / unwrap_model /
    log`` -- ``ivideogpt_amd.parallel.LocalAccelerator`` (torch.distributed over RCCL, no ``accelerate`` dependency) or HF's
    ``Accelerator`` itself;
  * FVD and the LPIPS column need network weights that do not ship (``args.use_fvd`` raises; ``eval/lpips`` is NaN);
  * GIF dumps (``imageio``) are not written; the ``eval/mse`` fallback of that branch is kept (:447-449).
The only collectives are the all-gathers of per-sample loss / metric rows (train_gpt.py:376, 476-479).
"""
import argparse
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def get_tokenizer(args):
    """train_gpt.py:128-149 (``ctx_vqgan`` only, as the reference: the plain ``vqgan`` branch raises there too).
    -> (tokenizer, size of the transformer's vocabulary: static + dynamic codes, + 2 with the frame-separator tokens)."""
    from ivideogpt_amd import CompressiveVQModel
    if args.vqgan_type != "ctx_vqgan":
        raise NotImplementedError(f"vqgan_type {args.vqgan_type!r}: only the compressive tokenizer is on the path")
    tok = CompressiveVQModel.from_pretrained(args.pretrained_model_name_or_path, subfolder=None, low_cpu_mem_usage=False).eval()
    had = tok.context_length
    if had != args.context_length:
        print(f"[Warning] pretrained context length of vq_model mismatch, change from {had} to {args.context_length}")
        tok.set_context_length(args.context_length)
    n_codes = tok.num_vq_embeddings + tok.num_dyn_embeddings
    return tok, n_codes + (2 if args.special_token else 0)


def generate_multiple_times(gen_times, accelerator, model, gen_input, actions, gen_kwargs, max_batch_size=None, verbose=False,
                            reward_prediction=False):
    """train_gpt.py:152-191: ``gen_times`` samples of every prompt row; one ``generate`` call carries ``max_batch_size // B`` copies
    of the B prompts (``max_batch_size`` None: one copy), so ``gen_times`` must be a multiple of that and ``max_batch_size`` of B
    (AssertionError otherwise, as the reference).  -> tokens [t * B, L], sample k of trajectory b at row k * B + b (the layout
    ``Evaluator`` expects); with ``reward_prediction`` also the rewards, concatenated the same way."""
    B = gen_input.shape[0]
    cap = B if max_batch_size is None or max_batch_size == 0 else max_batch_size
    copies, rest = divmod(cap, B)
    assert rest == 0, f"max_batch_size {cap} is not a multiple of the batch {B}"
    calls, rest = divmod(gen_times, copies)
    assert rest == 0, f"gen_times {gen_times} is not a multiple of the {copies} copies one call carries"
    net = accelerator.unwrap_model(model)
    prompts = gen_input.repeat(copies, 1)
    extra = dict(gen_kwargs, pad_token_id=50256)
    if actions is not None:
        extra["action"] = actions.repeat(copies, 1, 1)
    if reward_prediction:
        extra["return_reward"] = True
    if copies > 1 and getattr(net, "supports_shared_context", False):
        extra["shared_context"] = copies   # rows k * B + b repeat prompt b: prefilled once, its K / V rows kept once (ivg_generate_shared)
    tokens, rewards = [], []
    for _ in range(calls):
        out = net.generate(prompts, **extra)
        if reward_prediction:
            tokens.append(out[0]); rewards.append(out[1])
        else:
            tokens.append(out)
    tokens = torch.cat(tokens, dim=0)
    return (tokens, torch.cat(rewards, dim=0)) if reward_prediction else tokens


def batch_forward(batch_size, input, forward, verbose=False):
    """train_gpt.py:194-195: ``forward`` over consecutive slices of ``batch_size`` rows, results concatenated."""
    pieces = [forward(input[lo: lo + batch_size]) for lo in range(0, input.shape[0], batch_size)]
    return torch.cat(pieces, dim=0)


def _prompt_and_budget(args, tokens):
    """The context part of a tokenized clip and the number of tokens a rollout adds (train_gpt.py:392-403): 256 tokens per context
    frame and 16 per predicted frame, one more of each with the frame-separator tokens (the last separator is not generated)."""
    sep = 1 if args.special_token else 0
    n_future = args.segment_length - args.context_length
    return tokens[:, :args.context_length * (256 + sep)], (16 + sep) * n_future - sep


@torch.no_grad()
def evaluate(args, accelerator, tokenizer, model, eval_dataloader, evaluator, completed_steps):
    """train_gpt.py:321-512.  ``eval_dataloader`` yields ``pixel_values [B, T, 3, H, W]`` in [0, 1] (or ``(pixel_values, actions)``
    when ``args.action_conditioned``).  Returns the eval logs on the main process, None elsewhere."""
    if getattr(args, "use_fvd", False):
        raise NotImplementedError("FVD needs the I3D detector weights, which do not ship with the reference (out of scope)")
    tok, net = accelerator.unwrap_model(tokenizer), accelerator.unwrap_model(model)
    dev, ctx = accelerator.device, args.context_length
    rows = {k: [] for k in ("loss", "mse", "psnr", "ssim", "lpips")}   # per-sample values, gathered over the ranks batch by batch

    def keep(name, value, n):
        rows[name].append(accelerator.gather(value.repeat(n)))

    for it, batch in enumerate(eval_dataloader):
        if it == args.max_eval_iters:
            break
        frames, actions = batch if args.action_conditioned else (batch, None)
        frames = frames.to(dev, non_blocking=True)
        actions = actions.to(dev, non_blocking=True) if actions is not None else None
        n = frames.shape[0]

        # teacher-forced loss over the whole clip (:356-376)
        tokens, labels = tok.tokenize(frames, ctx)
        fwd = dict(input_ids=tokens, labels=labels)
        if actions is not None:
            fwd["action"] = actions
        out = net(**fwd)
        keep("loss", (out[0] if args.reward_prediction else out).loss, n)

        # rollout from the context frames, t samples per trajectory, decoded back to pixels (:390-430)
        show = it % args.log_gif_interval == 0 and accelerator.is_main_process
        recon = None
        if show or args.use_frame_metrics:
            prompt, budget = _prompt_and_budget(args, tokens)
            sampled = generate_multiple_times(args.eval_generate_times, accelerator, model, prompt, actions,
                                              gen_kwargs=dict(do_sample=True, temperature=1.0, top_k=100, max_new_tokens=budget),
                                              max_batch_size=args.max_generate_batchsize, reward_prediction=args.reward_prediction)
            sampled = sampled[0] if args.reward_prediction else sampled       # prompt included
            if args.eval_generate_times > 1 and getattr(tok, "supports_shared_context", False):
                decode = lambda ids: tok.detokenize(ids, ctx, shared_context="auto")   # the samples of a clip share its context: decoded once per clip
            else:
                decode = lambda ids: tok.detokenize(ids, ctx)
            chunk = args.max_decode_batchsize
            recon = batch_forward(chunk, sampled, decode) if chunk is not None and sampled.shape[0] > chunk else decode(sampled)
            recon = recon.clamp(0.0, 1.0)

        if args.use_frame_metrics:
            # the ground truth is clamped too: resized frames overshoot 1.0 by an ulp (:470-471)
            for name, value in zip(("mse", "psnr", "ssim", "lpips"), evaluator(frames.clamp(0.0, 1.0), recon)):
                keep(name, value, n)
        elif show:   # no metric pass: the plain pixel MSE of the shown batch stands in for eval/mse (:447-449), not gathered
            assert recon.shape[0] == n, "the fallback compares one sample per trajectory"
            rows["mse"].append(((frames.float() - recon) ** 2).mean().repeat(n))

    if not accelerator.is_main_process:
        return None
    mean = lambda name: torch.cat(rows[name], 0).mean().item() if rows[name] else float("nan")
    eval_loss = mean("loss")
    try:
        perplexity = math.exp(eval_loss)
    except OverflowError:
        perplexity = float("inf")
    eval_logs = {"eval/eval_loss": eval_loss, "eval/perplexity": perplexity, "eval/mse": mean("mse")}
    if args.use_frame_metrics:
        eval_logs.update({f"eval/{k}": mean(k) for k in ("psnr", "ssim", "lpips")})
    accelerator.log(eval_logs, step=completed_steps)
    return eval_logs


def eval_args(**overrides):
    """The subset of the reference's ``parse_args`` the eval loop reads, with its defaults (train_gpt.py:198-318)."""
    d = dict(action_conditioned=False, reward_prediction=False, use_fvd=False, use_frame_metrics=True, special_token=True, context_length=2,
             segment_length=16, eval_generate_times=1, max_generate_batchsize=None, max_decode_batchsize=None, max_eval_iters=100,
             log_gif_interval=10, vqgan_type="ctx_vqgan", pretrained_model_name_or_path=None, output_dir="eval-out")
    d.update(overrides)
    return argparse.Namespace(**d)


def main(argv=None):
    """``python train_gpt.py --eval_only``-style entry on synthetic clips and seeded random weights (no dataset / checkpoint ships):
    every rank evaluates its shard of the batches; rank 0 prints the logs."""
    import json
    from ivideogpt_amd import CompressiveVQModel, LlamaForCausalLM, parallel, weights as W
    from ivideogpt_amd.metrics import Evaluator
    ap = argparse.ArgumentParser()
    ap.add_argument("--pretrained_model_name_or_path", default=None, help="checkpoint directory (tokenizer/ + transformer/); default: seeded random weights")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--iters", type=int, default=2)
    ap.add_argument("--segment_length", type=int, default=16)
    ap.add_argument("--context_length", type=int, default=2)
    ap.add_argument("--eval_generate_times", type=int, default=2)
    ap.add_argument("--max_generate_batchsize", type=int, default=None)
    ap.add_argument("--max_decode_batchsize", type=int, default=None)
    a = ap.parse_args(argv)
    rank, world, local = parallel.init_from_env()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    if a.pretrained_model_name_or_path:
        tok = CompressiveVQModel.from_pretrained(a.pretrained_model_name_or_path, subfolder="tokenizer").to(dev)
        llm = LlamaForCausalLM.from_pretrained(a.pretrained_model_name_or_path, subfolder="transformer").to(dev)
    else:
        tcfg = W.tokenizer_config(**W.CTX_VAE64)
        tok = CompressiveVQModel(tcfg, W.random_tokenizer_state_dict(tcfg, 0, codebook_std=0.4)).to(dev)
        llm = LlamaForCausalLM(dict(W.LLAMA_SMALL), W.random_llama_state_dict(dict(W.LLAMA_SMALL), 0)).to(dev)
    if a.context_length != tok.context_length:
        tok.set_context_length(a.context_length)
    res = tok.config["resolution"]
    g = torch.Generator().manual_seed(1234)
    batches = [torch.rand(a.batch, a.segment_length, 3, res, res, generator=g) for _ in range(a.iters * world)][rank::world]
    args = eval_args(context_length=a.context_length, segment_length=a.segment_length, eval_generate_times=a.eval_generate_times,
                     max_generate_batchsize=a.max_generate_batchsize, max_decode_batchsize=a.max_decode_batchsize)
    logs = evaluate(args, parallel.LocalAccelerator(dev), tok, llm, batches, Evaluator(), 0)
    if logs is not None:
        print(json.dumps(logs))
    return logs


if __name__ == "__main__":
    main()
