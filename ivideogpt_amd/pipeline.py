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
    frames = tokenizer.detokenize(tokens, context_length, clamp=True)   # clamp(0, 1) in the epilogue of the decoders' last convolution
    return (frames, tokens) if return_tokens else frames


@torch.no_grad()
def frame_metrics(pred, target, first_frame=0):
    """Per-trajectory rows [B, 3] = (mse, psnr, ssim) over the frames from ``first_frame`` on, best of the t = pred.shape[0] /
    target.shape[0] samples per trajectory -- the reference's Evaluator semantics (ivideogpt/utils/video_metric.py:63-100, piqa
    PSNR / SSIM) computed by libivg ``ivg_frame_metrics``; the payload of the one collective of the multi-GPU path
    (the reference gathers mse / psnr / ssim / lpips, train_gpt.py:476-479)."""
    from .metrics import frame_metric_rows
    return frame_metric_rows(target, pred, gt_t0=first_frame, pred_t0=first_frame)
