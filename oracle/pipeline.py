"""The reference's CPU prediction path, restated op for op -- what the ``cpu_baseline`` leg of bench.py times.
TEST INFRASTRUCTURE (see oracle/__init__.py).

Follows ``predict()`` of inference/predict.py:47-73: ``tokenize`` of the WHOLE clip (all T frames go through the
encoders, :53), slice the 257*ctx context tokens (:54), ``generate`` 17F-1 tokens with top-k sampling (one prefill +
KV-cached steps for the action-free model, per-frame re-prefill for HeadModelWithAction), ``detokenize`` with the
F-times repeated context features, ``clamp(0, 1)`` (:72-73).
"""
import torch

from .llama import generate_reference_algorithm


@torch.no_grad()
def predict_reference_algorithm(tok_ref, llama_ref, pixel_values, ctx, uniforms=None, top_k=100, action_embeds=None, sdf_token=None):
    T = pixel_values.shape[1]
    F = T - ctx
    tokens, _ = tok_ref.tokenize(pixel_values, ctx)
    gen_input = tokens[:, :ctx * 257]
    n_new = 17 * F - 1
    out = generate_reference_algorithm(llama_ref, gen_input, n_new, top_k=top_k, uniforms=uniforms, action_embeds=action_embeds,
                                       ctx=ctx, sdf_token=sdf_token)
    return tok_ref.detokenize(out, ctx).clamp(0.0, 1.0), out
