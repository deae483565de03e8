"""The prediction hot path as one call:  context frames -> tokens -> autoregressive rollout -> frames.

Same sequence as ``predict()`` in /root/reference/inference/predict.py:47-73 (tokenize, keep the 257*ctx
context tokens, ``generate`` 17*F - 1 new tokens, ``detokenize``, ``clamp(0, 1)``), minus the work the
reference does and then throws away (it cond-encodes every future frame only to drop those tokens, :53-54).
"""
import torch


@torch.no_grad()
def predict_frames(tokenizer, model, pixel_values, context_length, future_length, actions=None, do_sample=True, top_k=100,
                   generator=None, uniforms=None, return_tokens=False):
    """pixel_values (B, >=ctx, 3, H, W) on the GPU (fp32 or bf16, [0,1]).  -> float32 (B, ctx+F, 3, H, W) in [0,1]."""
    prompt = tokenizer.encode_context(pixel_values, context_length)
    n_new = 17 * future_length - 1
    if actions is not None:
        tokens = model.generate(prompt, do_sample=do_sample, top_k=top_k, max_new_tokens=n_new, action=actions, generator=generator,
                                uniforms=uniforms)
    else:
        tokens = model.generate(prompt, do_sample=do_sample, top_k=top_k, max_new_tokens=n_new, generator=generator, uniforms=uniforms)
    frames = tokenizer.detokenize(tokens, context_length).clamp_(0.0, 1.0)
    return (frames, tokens) if return_tokens else frames


@torch.no_grad()
def frame_metrics(pred, target):
    """Per-trajectory rows [B, 4]: (mse, psnr, mean abs err, max abs err) over the predicted frames -- the payload of the
    one collective of the multi-GPU path (the reference gathers mse/psnr/ssim/lpips, train_gpt.py:476-479)."""
    d = (pred.float() - target.float())
    mse = d.pow(2).flatten(1).mean(1)
    psnr = -10.0 * torch.log10(mse.clamp_min(1e-12))
    return torch.stack([mse, psnr, d.abs().flatten(1).mean(1), d.abs().flatten(1).amax(1)], 1)
