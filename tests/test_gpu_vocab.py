"""End-to-end VQ token ids at the REAL vocabulary (8192 context + 8192 dynamics codes, configs/ctx_vae64 and configs/ctx_vae of the
reference; /root/reference/ivideogpt/vq_model/compressive_vq_model.py:102-123,199-220) through the full-width 114 M and 310 M
tokenizers, HIP fp32 engine vs the CPU oracle, with the near-tie audit of SURVEY.md section 7 contract (iii):

  * every id equal to the oracle's, EXCEPT positions where the oracle's own fp64 top-2 distance margin is below eps = 1e-4
    (relative to the best distance) and the engine's id is within eps of the best -- those are counted, printed and recorded
    (gpurun_out/r03_parity_margins.jsonl -> profiles/r03_parity_margins.txt);
  * codebooks drawn twice: N(mean, std) of the latents, and diffusers' default uniform initialiser scaled to the same std
    (a codebook far off the latent scale collapses the assignment onto a few codes and tests nothing).
"""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN, cache_oracle_stages, matched_codebooks, oracle_tokenizer, vq_near_tie_audit

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def make_tok(cfg, sd, ctx):
    from ivideogpt_amd import CompressiveVQModel
    m = CompressiveVQModel(cfg, sd, encode_dtype="fp32", decode_dtype="fp32").to(DEV)
    if ctx != cfg["context_length"]:
        m.set_context_length(ctx)
    return m


def clips_64(T, ctx):
    """16 clips: uniform noise, smooth ramps, the reference's real sample episode (tests/golden/fractal_clip_seed0.npz = the clip
    the reference's NPZParser makes of inference/samples/fractal_sample.npz), flat and saturated frames, zero-padded futures
    (what the reference's callers feed: vp/ivideogpt_interface.py:158-169)."""
    g = torch.Generator().manual_seed(2024)
    out = [torch.rand(4, T, 3, 64, 64, generator=g)]
    ramp = torch.linspace(0, 1, 64)[None, :].expand(64, 64)
    smooth = torch.stack([ramp, ramp.T, 1 - ramp], 0)[None, None].expand(3, T, 3, 64, 64).clone()
    smooth = smooth + 0.05 * torch.rand(3, T, 3, 64, 64, generator=g)
    for t in range(T):
        smooth[:, t] = torch.roll(smooth[:, t], shifts=3 * t, dims=-1)              # motion
    out.append(smooth.clamp(0, 1))
    real = torch.from_numpy(np.load(os.path.join(GOLDEN, "fractal_clip_seed0.npz"))["clip"])   # [16, 3, 64, 64]
    out.append(torch.stack([real[s:s + T] for s in (0, 3, 6, 16 - T)], 0))
    flat = torch.rand(1, 1, 3, 1, 1, generator=g).expand(1, T, 3, 64, 64).clone()
    sat = (torch.rand(1, T, 3, 64, 64, generator=g) > 0.5).float()
    out += [flat, sat]
    padded = torch.cat([torch.rand(2, T, 3, 64, 64, generator=g), torch.stack([real[:T], real[8:8 + T]], 0)], 0)[:3]
    padded[:, ctx:] = 0
    out.append(padded)
    px = torch.cat(out, 0)
    assert px.shape[0] == 16
    return px.contiguous()


def clips_256(T, ctx):
    """4 clips at 256 x 256: noise, a moving ramp, the real episode (antialiased bilinear resize of the uint8 frames, the
    reference's preprocessing: inference/utils.py:12-16), the real context with zero-padded futures."""
    g = torch.Generator().manual_seed(2025)
    noise = torch.rand(1, T, 3, 256, 256, generator=g)
    ramp = torch.linspace(0, 1, 256)[None, :].expand(256, 256)
    smooth = torch.stack([ramp, ramp.T, 1 - ramp], 0)[None, None].expand(1, T, 3, 256, 256).clone()
    for t in range(T):
        smooth[:, t] = torch.roll(smooth[:, t], shifts=9 * t, dims=-1)
    ep = torch.from_numpy(np.load(os.path.join(GOLDEN, "fractal_sample.npz"))["image"][:8 * T:8]).permute(0, 3, 1, 2).float() / 255
    real = torch.nn.functional.interpolate(ep, size=(256, 256), mode="bilinear", antialias=True).clamp(0, 1)[None]
    padded = real.clone()
    padded[:, ctx:] = 0
    return torch.cat([noise, smooth, real, padded], 0).contiguous()


def run_audit(cfg_base, seed, px, ctx, what):
    from ivideogpt_amd import weights as W
    cfg = W.tokenizer_config(**cfg_base)
    assert cfg["num_vq_embeddings"] == 8192 and cfg["num_dyn_embeddings"] == 8192          # the released vocabulary, uncut
    sd = W.random_tokenizer_state_dict(cfg, seed, codebook_std=0.4)
    ora = oracle_tokenizer(cfg, sd, ctx)
    cache_oracle_stages(ora, px, ctx)
    total = tol = 0
    for kind in ("gauss", "uniform"):
        stds = matched_codebooks(ora, sd, px, ctx, kind, seed + 1)
        ids_ref, labels_ref = ora.tokenize(px, ctx)
        used = (len(torch.unique(ids_ref[ids_ref < 8192])), len(torch.unique(ids_ref[(ids_ref >= 8192) & (ids_ref < 16384)])))
        m = make_tok(cfg, sd, ctx)
        ids, labels = m.tokenize(px.to(DEV), ctx)
        st = vq_near_tie_audit(ora, px, ctx, ids, ids_ref, eps=1e-4, what=f"{what}, {kind} codebooks (latent std {stds[0]:.3f} / {stds[1]:.3f}, "
                               f"{used[0]} + {used[1]} distinct codes used)")
        # labels: -100 over the context, the future part equal to the ids (compressive_vq_model.py:216-218)
        lab = labels.cpu()
        assert torch.equal(lab == -100, labels_ref == -100) and torch.equal(lab[lab != -100], ids.cpu()[lab != -100])
        n_dyn = px.shape[0] * (px.shape[1] - ctx) * 16
        assert used[0] > 500 and used[1] > min(100, n_dyn // 2), f"degenerate assignment: {used} distinct codes"
        total += st["tokens"]; tol += st["mismatches_tolerated_as_near_ties"]
        del m
    print(f"{what}: {total} code tokens audited at 8192 + 8192 codes, {tol} near-tie mismatches tolerated, 0 others")
    assert tol <= 2 + total // 500, f"{tol} near-tie flips in {total} tokens: more than the margin distribution explains"


def test_full_vocab_64_tokenizer_16_clips_near_tie_audit():
    from ivideogpt_amd import weights as W
    run_audit(W.CTX_VAE64, 131, clips_64(6, 2), 2, "ctx_vae64 (114 M), 16 clips x (2 + 4) frames")


def test_full_vocab_64_tokenizer_ctx1():
    """set_context_length(1) (the BAIR / config-3 shape) at the real vocabulary."""
    from ivideogpt_amd import weights as W
    run_audit(W.CTX_VAE64, 137, clips_64(3, 1)[:8], 1, "ctx_vae64 ctx 1, 8 clips x (1 + 2) frames")


def test_full_vocab_256_tokenizer_4_clips_near_tie_audit():
    from ivideogpt_amd import weights as W
    run_audit(W.CTX_VAE256, 133, clips_256(3, 2), 2, "ctx_vae256 (310 M), 4 clips x (2 + 1) frames")
