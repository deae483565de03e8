"""Shared helpers for the parity tests: rebuild the seeded weights a golden fixture was made with."""
import json
import os

import numpy as np
import torch

from ivideogpt_amd import weights as W

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    d = np.load(os.path.join(GOLDEN, name))
    return {k: d[k] for k in d.files}


def tokenizer_fixture(name):
    """-> (cfg, state dict, ctx, pixels [B,T,3,H,W] fp32, golden arrays)"""
    g = load_golden(name)
    cfg = W.tokenizer_config(**json.loads(str(g["config"])))
    sd = W.random_tokenizer_state_dict(cfg, int(g["seed"]), float(g["codebook_std"]))
    px = torch.from_numpy(g["pixels_u8"]).float() / 255.0
    return cfg, sd, int(g["context_length"]), px, g


def oracle_tokenizer(cfg, sd, ctx):
    from oracle.vq_tokenizer import CompressiveVQRef
    m = CompressiveVQRef(**cfg).eval()
    m.load_state_dict(sd, strict=True)
    if ctx != cfg["context_length"]:
        m.set_context_length(ctx)
    return m


def llama_fixture(name):
    g = load_golden(name)
    cfg = json.loads(str(g["config"]))
    adim = int(g["action_dim"]) if "action_dim" in g else None
    sd = W.random_llama_state_dict(cfg, int(g["seed"]), action_dim=adim)
    return cfg, sd, g


def oracle_llama(cfg, sd, prefix="model."):
    from oracle.llama import LlamaRef
    return LlamaRef(sd, cfg["num_hidden_layers"], cfg["num_attention_heads"], cfg["rms_norm_eps"],
                    cfg["rope_theta"], cfg["max_position_embeddings"], prefix=prefix)


def assert_sampled_rollout_matches(out, ref, oracle_model, uniforms, top_k, L0, what="rollout", tie=3e-3):
    """Sampled rollouts of two fp32 implementations agree token for token EXCEPT where a uniform lands on a boundary of the
    inverse CDF closer than what the logits bar allows: logits within 1e-3 (the parity bar) move every kept probability by up to
    0.1 % and a CDF boundary by up to ~0.2 % of the mass; a draw closer than that to a boundary may fall to the neighbouring kept
    token, and the rest of that row then legitimately diverges.  Every row must therefore equal the oracle's up to its first
    mismatch, and that mismatch must be such a near-tie under the ORACLE's own logits (|u * total - cdf boundary| / total < `tie`,
    engine token = the adjacent kept token).  Returns the number of rows that diverged at a near-tie."""
    out, ref = out.cpu(), ref.cpu()
    assert out.shape == ref.shape
    diverged = 0
    for b in range(out.shape[0]):
        bad = (out[b] != ref[b]).nonzero().flatten()
        if len(bad) == 0:
            continue
        p = int(bad[0])
        j = p - L0                                             # index of the new token (0-based) -> uniform column j
        assert j >= 0, f"{what}: row {b} differs inside the prompt"
        logits = oracle_model.logits(ref[b:b + 1, :p])[0, -1].double()
        kth = torch.topk(logits, min(top_k, logits.numel())).values[-1]
        keep = logits >= kth
        e = torch.where(keep, torch.exp(logits - logits.max()), torch.zeros((), dtype=torch.double))
        cdf = torch.cumsum(e, 0)
        total = cdf[-1]
        target = uniforms[b, j].double() * total
        kept_ids = keep.nonzero().flatten()
        k_ref = int((kept_ids == ref[b, p]).nonzero())
        k_out = (kept_ids == out[b, p]).nonzero()
        assert len(k_out) == 1, f"{what}: row {b}, new token {j + 1}: the engine drew a token outside the oracle's top-{top_k} set"
        k_out = int(k_out)
        assert abs(k_out - k_ref) == 1, f"{what}: row {b}, new token {j + 1}: tokens {int(out[b, p])} vs {int(ref[b, p])} are not neighbours in the kept set"
        boundary = cdf[kept_ids[min(k_out, k_ref)]]
        margin = float((target - boundary).abs() / total)
        assert margin < tie, f"{what}: row {b}, new token {j + 1}: differs from the oracle with a CDF margin of {margin:.2e} (not a near-tie)"
        diverged += 1
    return diverged
