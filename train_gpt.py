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
    """train_gpt.py:128-149 (``ctx_vqgan`` only, as the reference: the plain ``vqgan`` branch raises there too)."""
    from ivideogpt_amd import CompressiveVQModel
    if args.vqgan_type != "ctx_vqgan":
        raise NotImplementedError
    vq_model = CompressiveVQModel.from_pretrained(args.pretrained_model_name_or_path, subfolder=None, low_cpu_mem_usage=False).eval()
    if args.context_length != vq_model.context_length:
        print(f"[Warning] pretrained context length of vq_model mismatch, change from {vq_model.context_length} to {args.context_length}")
        vq_model.set_context_length(args.context_length)
    vocab_size = vq_model.num_vq_embeddings + vq_model.num_dyn_embeddings
    if args.special_token:
        vocab_size += 2
    return vq_model, vocab_size


def generate_multiple_times(gen_times, accelerator, model, gen_input, actions, gen_kwargs, max_batch_size=None, verbose=False,
                            reward_prediction=False):
    """train_gpt.py:152-191: ``gen_times`` samples of every prompt row, ``max_batch_size // B`` repeats per ``generate`` call.
    -> tokens [t * B, L] with sample k of trajectory b at row k * B + b (what ``Evaluator`` expects)."""
    max_batch_size = max_batch_size or gen_input.shape[0]
    assert max_batch_size % gen_input.shape[0] == 0
    repeat_times = max_batch_size // gen_input.shape[0]
    assert gen_times % (max_batch_size // gen_input.shape[0]) == 0
    repeat_iters = gen_times // (max_batch_size // gen_input.shape[0])
    results, rewards = [], []
    m = accelerator.unwrap_model(model)
    for _ in range(repeat_iters):
        kw = dict(gen_kwargs)
        if actions is not None:
            kw["action"] = actions.repeat(repeat_times, 1, 1)
        if reward_prediction:
            generated_tokens, reward = m.generate(gen_input.repeat(repeat_times, 1), **kw, pad_token_id=50256, return_reward=True)
            rewards.append(reward)
        else:
            generated_tokens = m.generate(gen_input.repeat(repeat_times, 1), **kw, pad_token_id=50256)
        results.append(generated_tokens)
    if reward_prediction:
        return torch.cat(results, dim=0), torch.cat(rewards, dim=0)
    return torch.cat(results, dim=0)   # [t*B, ...] where t means number of generation times


def batch_forward(batch_size, input, forward, verbose=False):
    """train_gpt.py:194-195."""
    return torch.cat([forward(input[i: i + batch_size]) for i in range(0, input.shape[0], batch_size)], dim=0)


@torch.no_grad()
def evaluate(args, accelerator, tokenizer, model, eval_dataloader, evaluator, completed_steps):
    """train_gpt.py:321-512.  ``eval_dataloader`` yields ``pixel_values [B, T, 3, H, W]`` in [0, 1] (or ``(pixel_values, actions)``
    when ``args.action_conditioned``).  Returns the eval logs on the main process, None elsewhere."""
    if getattr(args, "use_fvd", False):
        raise NotImplementedError("FVD needs the I3D detector weights, which do not ship with the reference (out of scope)")
    losses = []
    mse_values, psnr_values, ssim_values, lpips_values = [], [], [], []
    tok, mdl = accelerator.unwrap_model(tokenizer), accelerator.unwrap_model(model)

    for i, batch in enumerate(eval_dataloader):
        if i == args.max_eval_iters:
            break
        if args.action_conditioned:
            pixel_values, actions = batch
            actions = actions.to(accelerator.device, non_blocking=True)
            pixel_values = pixel_values.to(accelerator.device, non_blocking=True)
        else:
            pixel_values, actions = batch.to(accelerator.device, non_blocking=True), None
        batch_size = pixel_values.shape[0]

        tokens, labels = tok.tokenize(pixel_values, args.context_length)
        model_input = {"input_ids": tokens, "labels": labels}
        if args.action_conditioned:
            model_input["action"] = actions
        if args.reward_prediction:
            outputs, rewards = mdl(**model_input)
        else:
            outputs = mdl(**model_input)
        loss = outputs.loss
        losses.append(accelerator.gather(loss.repeat(batch_size)))

        # predict next frames
        recon_output = None
        if (i % args.log_gif_interval == 0 and accelerator.is_main_process) or args.use_frame_metrics:
            if args.special_token:
                gen_input = tokens[:, :args.context_length * (256 + 1)]
                max_new_tokens = (1 + 16) * (args.segment_length - args.context_length) - 1
            else:
                gen_input = tokens[:, :args.context_length * 256]
                max_new_tokens = 16 * (args.segment_length - args.context_length)
            gen_kwargs = {"do_sample": True, "temperature": 1.0, "top_k": 100, "max_new_tokens": max_new_tokens}
            out = generate_multiple_times(args.eval_generate_times, accelerator, model, gen_input, actions if args.action_conditioned else None,
                                          gen_kwargs=gen_kwargs, max_batch_size=args.max_generate_batchsize, verbose=False,
                                          reward_prediction=args.reward_prediction)
            generated_tokens = out[0] if args.reward_prediction else out
            if args.max_decode_batchsize is not None and generated_tokens.shape[0] > args.max_decode_batchsize:
                recon_output = batch_forward(args.max_decode_batchsize, generated_tokens, lambda x: tok.detokenize(x, args.context_length))
            else:
                recon_output = tok.detokenize(generated_tokens, args.context_length)   # generated_tokens include gen_input
            recon_output = recon_output.clamp(0.0, 1.0)

        if i % args.log_gif_interval == 0 and accelerator.is_main_process and not args.use_frame_metrics:
            assert pixel_values.shape[0] == recon_output.shape[0]
            mse_values.append(torch.mean((pixel_values.float() - recon_output) ** 2).repeat(batch_size))

        if args.use_frame_metrics:
            # pixel_values can be 1.0000001192092896 numerically (train_gpt.py:470-471)
            mse_value, psnr_value, ssim_value, lpips_value = evaluator(pixel_values.clamp(0.0, 1.0), recon_output)
            mse_values.append(accelerator.gather(mse_value.repeat(batch_size)))
            psnr_values.append(accelerator.gather(psnr_value.repeat(batch_size)))
            ssim_values.append(accelerator.gather(ssim_value.repeat(batch_size)))
            lpips_values.append(accelerator.gather(lpips_value.repeat(batch_size)))

    if not accelerator.is_main_process:
        return None
    eval_loss = torch.cat(losses, 0).mean().item()
    try:
        perplexity = math.exp(eval_loss)
    except OverflowError:
        perplexity = float("inf")
    eval_logs = {"eval/eval_loss": eval_loss, "eval/perplexity": perplexity,
                 "eval/mse": torch.cat(mse_values, 0).mean().item() if mse_values else float("nan")}
    if args.use_frame_metrics:
        eval_logs.update({"eval/psnr": torch.cat(psnr_values, 0).mean().item(), "eval/ssim": torch.cat(ssim_values, 0).mean().item(),
                          "eval/lpips": torch.cat(lpips_values, 0).mean().item()})
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
