"""The benchmarked arithmetic (bf16 decode + bf16 rollout) against the fp32 reference, MEASURED AND RECORDED:
mini models (reference-generated golden vectors) and the full-width released shapes (CPU oracle).  bf16 storage cannot meet the
1e-3 bar of the fp32 mode (bf16 eps = 3.9e-3); the bars below are what the reference's own bf16 paths deliver (autocast decode,
HF bf16 weights: tests/golden/bf16_mini64_ctx2.npz), and every measured deviation is appended to
gpurun_out/r03_bf16_deviations.jsonl (copied into profiles/r03_parity_margins.txt by the builder)."""
import json
import os

import numpy as np
import pytest
import torch

from helpers import llama_fixture, load_golden, oracle_llama, oracle_tokenizer, tokenizer_fixture

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def record(**kw):
    print("bf16 deviation:", json.dumps(kw))
    try:
        out = os.path.join(ROOT, "gpurun_out")
        os.makedirs(out, exist_ok=True)
        with open(os.path.join(out, "r03_bf16_deviations.jsonl"), "a") as f:
            f.write(json.dumps(kw) + "\n")
    except OSError:
        pass


def test_bf16_deviations_mini_models_recorded():
    from ivideogpt_amd import CompressiveVQModel, LlamaForCausalLM
    gb = load_golden("bf16_mini64_ctx2.npz")
    cfg, sd, ctx, px, g = tokenizer_fixture("tok_mini64_ctx2.npz")
    m = CompressiveVQModel(cfg, sd, encode_dtype="fp32", decode_dtype="bf16").to(DEV)
    rec = m.detokenize(torch.from_numpy(g["indices"]).to(DEV), ctx).cpu().numpy()
    d32, dac = np.abs(rec - g["recon"]), np.abs(rec - gb["pixels_autocast"])
    ref_dev = gb["autocast_pixel_dev"]
    record(what="mini tokenizer, decoded pixels", engine_bf16_vs_reference_fp32_max=float(d32.max()), engine_bf16_vs_reference_fp32_mean=float(d32.mean()),
           engine_bf16_vs_reference_autocast_max=float(dac.max()), engine_bf16_vs_reference_autocast_mean=float(dac.mean()),
           reference_autocast_vs_its_fp32_max=float(ref_dev[0]), reference_autocast_vs_its_fp32_mean=float(ref_dev[1]))
    assert d32.max() <= 1.25 * ref_dev[0] and d32.mean() <= 1.25 * ref_dev[1]
    lcfg, lsd, gl = llama_fixture("llama_tiny_ctx2_free.npz")
    ids = torch.from_numpy(gl["teacher_ids"])
    ref = oracle_llama(lcfg, lsd).logits(ids)
    lg = LlamaForCausalLM(lcfg, lsd, dtype="bf16").to(DEV).logits(ids.to(DEV)).cpu()
    d = (lg - ref).abs()
    top1 = float((lg.argmax(-1) == ref.argmax(-1)).float().mean())
    record(what="mini transformer, teacher-forced logits", engine_bf16_vs_fp32_max=float(d.max()), engine_bf16_vs_fp32_mean=float(d.mean()),
           logit_scale=float(ref.abs().max()), top1_agreement=top1)
    assert d.max() < 0.30


def test_bf16_deviations_full_width_recorded():
    """ivideogpt-oxe-64-act-free shapes at full width (114 M tokenizer, 138 M transformer, seeded random weights)."""
    from ivideogpt_amd import CompressiveVQModel, LlamaForCausalLM, weights as W
    tcfg = W.tokenizer_config(**W.CTX_VAE64)
    tsd = W.random_tokenizer_state_dict(tcfg, 31, codebook_std=0.4)
    px = torch.randint(0, 256, (1, 4, 3, 64, 64), generator=torch.Generator().manual_seed(2)).float() / 255
    ora = oracle_tokenizer(tcfg, tsd, 2)
    ids_ref, _ = ora.tokenize(px, 2)
    ref = ora.detokenize(ids_ref, 2)
    m16 = CompressiveVQModel(tcfg, tsd, encode_dtype="fp32", decode_dtype="bf16").to(DEV)
    m32 = CompressiveVQModel(tcfg, tsd, encode_dtype="fp32", decode_dtype="fp32").to(DEV)
    d16 = (m16.detokenize(ids_ref.to(DEV), 2).cpu() - ref).abs()
    d32 = (m32.detokenize(ids_ref.to(DEV), 2).cpu() - ref).abs()
    with torch.autocast("cpu", dtype=torch.bfloat16):
        auto = ora.detokenize(ids_ref, 2).float()
    dau = (auto - ref).abs()
    scale = float(ref.abs().mean())
    record(what="ctx_vae64 full width, decoded pixels (random weights: pixel range is not [0, 1])", pixel_abs_mean_of_reference=scale,
           engine_bf16_max=float(d16.max()), engine_bf16_mean=float(d16.mean()), engine_fp32_max=float(d32.max()),
           oracle_autocast_max=float(dau.max()), oracle_autocast_mean=float(dau.mean()))
    assert d32.max() < 1e-3
    assert d16.mean() <= 1.5 * dau.mean() + 1e-3 and d16.max() <= 2.0 * dau.max() + 1e-2, "bf16 decode further from fp32 than the reference-style autocast path"
    lcfg = dict(W.LLAMA_SMALL)
    lsd = W.random_llama_state_dict(lcfg, 41)
    ids = torch.randint(0, 16386, (2, 300), generator=torch.Generator().manual_seed(3))
    lref = oracle_llama(lcfg, lsd).logits(ids)
    l16 = LlamaForCausalLM(lcfg, lsd, dtype="bf16").to(DEV).logits(ids.to(DEV)).cpu()
    l32 = LlamaForCausalLM(lcfg, lsd, dtype="fp32").to(DEV).logits(ids.to(DEV)).cpu()
    # the reference's bf16 route: HF weights cast to bf16 (vp/ivideogpt_interface.py, mbrl/video_predictor.py load bf16 checkpoints)
    lsd16 = {k: v.to(torch.bfloat16).float() for k, v in lsd.items()}
    lw16 = oracle_llama(lcfg, lsd16).logits(ids)
    d16, d32, dw = (l16 - lref).abs(), (l32 - lref).abs(), (lw16 - lref).abs()
    record(what="Llama small (12 layers) full width, teacher-forced logits, 2 x 300 tokens", logit_scale=float(lref.abs().max()),
           engine_bf16_max=float(d16.max()), engine_bf16_mean=float(d16.mean()), engine_fp32_max=float(d32.max()),
           fp32_arithmetic_with_bf16_rounded_weights_max=float(dw.max()), fp32_arithmetic_with_bf16_rounded_weights_mean=float(dw.mean()),
           top1_agreement_bf16=float((l16.argmax(-1) == lref.argmax(-1)).float().mean()),
           top100_overlap_bf16=float(np.mean([len(set(a.tolist()) & set(b.tolist())) / 100.0
                                              for a, b in zip(l16[0, ::25].topk(100).indices, lref[0, ::25].topk(100).indices)])))
    assert d32.max() < 1e-3
    assert d16.mean() <= 3.0 * dw.mean() + 1e-3, "bf16 engine logits further from fp32 than bf16 weight rounding alone explains (x3)"
