"""The prediction hot path as one call:  context frames -> tokens -> autoregressive rollout -> frames.

Same sequence as ``predict()`` in /root/reference/inference/predict.py:47-73 (tokenize, keep the 257*ctx
context tokens, ``generate`` 17*F - 1 new tokens, ``detokenize``, ``clamp(0, 1)``), minus the work the
reference does and then throws away (it cond-encodes every future frame only to drop those tokens, :53-54).
"""
import contextlib

import torch


@torch.no_grad()
def predict_frames(tokenizer, model, pixel_values, context_length, future_length, actions=None, do_sample=True, top_k=100,
                   generator=None, uniforms=None, return_tokens=False, temperature=1.0, conv_gate=None, metrics_of=None, rollout_stream=None,
                   shared_context=None):
    """pixel_values (B, >=ctx, 3, H, W) on the GPU (fp32 or bf16, [0,1]).  -> float32 (B, ctx+F, 3, H, W) in [0,1].
    ``shared_context=t``: t samples of every clip (predict.py's ``repeat_times``, VP2's candidates: ``actions`` then has t * B rows,
    row k * B + b = sample k of clip b, like the result): the context is encoded, prefilled and decoded once per clip.
    ``conv_gate`` (parallel.PhaseGate, several batches in flight on one GPU): the two convolution phases -- context encode, frame
    decode -- are ordered against the other lanes' convolution phases; the rollout between them is not.  ``metrics_of`` (ground-truth
    clip): the per-trajectory metric rows of the predicted frames are computed inside the decode phase and returned beside the frames.
    ``rollout_stream``: the rollout runs on this stream (ordered after the encode and before the decode of the current stream) -- e.g. a
    CU-masked stream, so that rollouts and convolution phases of different lanes own disjoint compute units."""
    stream = torch.cuda.current_stream(pixel_values.device)
    gated = (lambda: conv_gate.phase(stream)) if conv_gate is not None else contextlib.nullcontext
    with gated():
        prompt = tokenizer.encode_context(pixel_values, context_length)
    n_new = 17 * future_length - 1
    kw = {} if temperature == 1.0 else {"temperature": temperature}
    dkw = {}
    if shared_context and int(shared_context) > 1:
        prompt = prompt.repeat(int(shared_context), 1)
        kw["shared_context"] = dkw["shared_context"] = int(shared_context)
    if actions is not None:
        kw["action"] = actions
    if do_sample and uniforms is None:   # drawn on the caller's stream (the generator's state is not tied to a stream)
        uniforms = torch.rand(prompt.shape[0], n_new, device=prompt.device, dtype=torch.float32, generator=generator)
    if rollout_stream is not None:
        rollout_stream.wait_stream(stream)
        with torch.cuda.stream(rollout_stream):
            tokens = model.generate(prompt, do_sample=do_sample, top_k=top_k, max_new_tokens=n_new, uniforms=uniforms, **kw)
        stream.wait_stream(rollout_stream)
        for t in (prompt, uniforms, tokens):
            if t is not None:
                t.record_stream(rollout_stream)
    else:
        tokens = model.generate(prompt, do_sample=do_sample, top_k=top_k, max_new_tokens=n_new, uniforms=uniforms, **kw)
    with gated():
        frames = tokenizer.detokenize(tokens, context_length, clamp=True, **dkw)   # clamp(0, 1) in the epilogue of the decoders' last convolution
        rows = frame_metrics(frames, metrics_of, first_frame=context_length) if metrics_of is not None else None
    out = (frames, tokens) if return_tokens else (frames,)
    if metrics_of is not None:
        out = out + (rows,)
    return out if len(out) > 1 else out[0]


@torch.no_grad()
def frame_metrics(pred, target, first_frame=0):
    """Per-trajectory rows [B, 3] = (mse, psnr, ssim) over the frames from ``first_frame`` on, best of the t = pred.shape[0] /
    target.shape[0] samples per trajectory -- the reference's Evaluator semantics (ivideogpt/utils/video_metric.py:63-100, piqa
    PSNR / SSIM) computed by libivg ``ivg_frame_metrics``; the payload of the one collective of the multi-GPU path
    (the reference gathers mse / psnr / ssim / lpips, train_gpt.py:476-479)."""
    from .metrics import frame_metric_rows
    return frame_metric_rows(target, pred, gt_t0=first_frame, pred_t0=first_frame)
