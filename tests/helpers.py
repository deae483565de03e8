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


# ------------------------------------------------------------------------------------------------ VQ near-tie audit
def _margins_fp64(z, E, chunk=2048):
    """z [R, D] fp32 latents (the ORACLE's), E [n_e, D] codebook -> fp64 (best id, best distance, runner-up distance, all distances fn)."""
    best_i, d1, d2 = [], [], []
    Ed = E.double()
    for r0 in range(0, z.shape[0], chunk):
        d = torch.cdist(z[r0:r0 + chunk].double(), Ed)
        t = torch.topk(d, 2, dim=1, largest=False)
        best_i.append(t.indices[:, 0]); d1.append(t.values[:, 0]); d2.append(t.values[:, 1])
    return torch.cat(best_i), torch.cat(d1), torch.cat(d2)


def vq_near_tie_audit(ora, px, ctx, ids, ids_ref, eps=1e-4, what="tokenize", record=True):
    """SURVEY.md section 7, contract (iii): end-to-end token ids of the engine (`ids`) against the oracle's (`ids_ref`) at the
    REAL vocabulary.  A differing id is tolerated only if it is a NEAR-TIE under the oracle's own latents: the oracle's fp64
    top-2 distance margin at that position, relative to the best distance, is below `eps`, and the id the engine chose is within
    `eps` (relative) of the best distance.  Returns a dict of statistics (token count, mismatches, the margin distribution), prints
    it and appends it to gpurun_out/r03_parity_margins.jsonl (copied into profiles/ by the builder)."""
    ids, ids_ref = ids.cpu(), ids_ref.cpu()
    assert ids.shape == ids_ref.shape
    B, T = px.shape[:2]
    fut = T - ctx
    st = ora.encode_stages(px, ctx)
    n_vq = ora.num_vq_embeddings
    zc = st["hq"].permute(0, 2, 3, 1).reshape(B, ctx * 256, -1)            # rows (frame, i) of trajectory b
    zd = st["dq"].reshape(B, fut * 16, -1)
    pos_c = torch.tensor([f * 257 + i for f in range(ctx) for i in range(256)])
    pos_d = torch.tensor([257 * ctx + 17 * f + i for f in range(fut) for i in range(16)])
    special = torch.ones(ids.shape[1], dtype=torch.bool)
    special[pos_c] = False; special[pos_d] = False
    assert torch.equal(ids[:, special], ids_ref[:, special]), f"{what}: separator tokens differ"
    stats = {"what": what, "eps": eps, "tokens": int(B * (len(pos_c) + len(pos_d)))}
    n_bad = n_tol = 0
    worst = 0.0
    for kind, z, pos, E, off in (("ctx", zc, pos_c, ora.quantize.embedding.weight, 0),
                                 ("dyn", zd, pos_d, ora.dynamics_quantize.embedding.weight, n_vq)):
        zf = z.reshape(-1, z.shape[-1])
        bi, d1, d2 = _margins_fp64(zf, E.detach())
        rel = ((d2 - d1) / d1.clamp_min(1e-30)).reshape(B, -1)
        got = (ids[:, pos] - off).reshape(-1)
        ref = (ids_ref[:, pos] - off).reshape(-1)
        stats[f"{kind}_oracle_fp32_vs_fp64_argmin_differ"] = int((ref != bi).sum())
        q = torch.tensor([0.0, 1e-4, 1e-3, 1e-2, 0.5], dtype=torch.double)
        stats[f"{kind}_rel_margin_min_q1e-4_q1e-3_q1e-2_median"] = [float(v) for v in torch.quantile(rel.reshape(-1), q)]
        stats[f"{kind}_tokens_with_margin_below_eps"] = int((rel < eps).sum())
        bad = (got != ref).nonzero().flatten()
        stats[f"{kind}_mismatches"] = int(len(bad))
        for r in bad.tolist():
            m = float(rel.reshape(-1)[r])
            dg = float(torch.cdist(zf[r:r + 1].double(), E.detach()[int(got[r].clamp(0, E.shape[0] - 1))][None].double()))
            excess = (dg - float(d1[r])) / max(float(d1[r]), 1e-30)
            worst = max(worst, m)
            if m < eps and excess < eps and 0 <= int(got[r]) < E.shape[0]:
                n_tol += 1
            else:
                n_bad += 1
                print(f"{what}: {kind} row {r}: engine id {int(got[r])} vs oracle {int(ref[r])}: oracle top-2 relative margin {m:.3e}, "
                      f"engine id's excess distance {excess:.3e} -- NOT a near-tie")
    stats["mismatches_tolerated_as_near_ties"] = n_tol
    stats["mismatches_not_near_ties"] = n_bad
    stats["worst_tolerated_margin"] = worst
    print("VQ near-tie audit:", json.dumps(stats))
    if record:
        try:
            out = os.path.join(os.path.dirname(GOLDEN.rstrip("/")), os.pardir, "gpurun_out")
            os.makedirs(out, exist_ok=True)
            with open(os.path.join(out, "r03_parity_margins.jsonl"), "a") as f:
                f.write(json.dumps(stats) + "\n")
        except OSError:
            pass
    assert n_bad == 0, f"{what}: {n_bad} token ids differ from the oracle at positions that are not near-ties (see the lines above)"
    return stats


def cache_oracle_stages(ora, px, ctx):
    """Run the oracle's encoder trunks ONCE for (px, ctx) and make ``ora.encode_stages`` / ``ora.tokenize`` answer from that run
    (only the two codebook look-ups are redone): the audits below swap codebooks under fixed latents."""
    st = ora.encode_stages(px, ctx)

    def stages(pixel_values, context_length):
        assert pixel_values is px and context_length == ctx
        with torch.no_grad():
            st["idx_c"] = ora.quantize(st["hq"])[2][2]
            st["idx_d"] = ora.dynamics_quantize(st["dq"].transpose(-1, -2).unsqueeze(-1))[2][2]
        return st
    ora.encode_stages = stages
    return st


def matched_codebooks(ora, sd, px, ctx, kind, seed):
    """Codebooks at the REAL size redrawn to the statistics of the latents they quantise (random conv weights put the latents at
    an arbitrary scale; a codebook far off that scale makes every row pick the same few codes): kind 'gauss' = N(0, std of the
    latents) around their per-channel mean, kind 'uniform' = diffusers' default initialiser U(-a, a) (VectorQuantizer: U(+-1/n_e)) with a scaled to the same
    std.  Writes them into `sd` and into the oracle; returns the (ctx, dyn) latent stds."""
    st = ora.encode_stages(px, ctx)
    g = torch.Generator().manual_seed(seed)
    stds = []
    for key, mod, z in (("quantize.embedding.weight", ora.quantize, st["hq"].permute(0, 2, 3, 1).reshape(-1, st["hq"].shape[1])),
                        ("dynamics_quantize.embedding.weight", ora.dynamics_quantize, st["dq"].reshape(-1, st["dq"].shape[-1]))):
        mu = z.mean(0, keepdim=True)
        s = float((z - mu).std())
        shape = sd[key].shape
        if kind == "gauss":
            w = mu + torch.randn(shape, generator=g) * s
        else:
            w = mu + (torch.rand(shape, generator=g) * 2 - 1) * (s * 3 ** 0.5)
        sd[key] = w.contiguous()
        with torch.no_grad():
            mod.embedding.weight.copy_(w)
        stds.append(s)
    return stds


# ------------------------------------------------------------------------------------------------ MBRL world-model checkpoint tree
def world_model_files(tmp_path, load_internal_llm):
    """A checkpoint directory tree laid out as the reference's mbrl/cfgs/mbpo_config.yaml expects it."""
    import json
    from safetensors.torch import save_file
    from ivideogpt_amd import weights as W
    tcfg = W.tokenizer_config(block_out_channels=(64, 64, 64), layers_per_block=1, latent_channels=64, num_vq_embeddings=64,
                              num_dyn_embeddings=64, mid_block_add_attention=False, context_length=2, resolution=64, max_att_resolution=16)
    tsd = W.random_tokenizer_state_dict(tcfg, 3, codebook_std=0.4)
    W.save_tokenizer_checkpoint(str(tmp_path / "tokenizer"), tcfg, tsd)
    lcfg = dict(W.LLAMA_SMALL, hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2, vocab_size=130)
    (tmp_path / "configs" / "llama").mkdir(parents=True)
    with open(tmp_path / "configs" / "llama" / "config.json", "w") as f:
        json.dump(dict(lcfg, model_type="llama", vocab_size=32000), f)      # the shipped config's vocabulary is overwritten by load_models
    full = W.random_llama_state_dict(lcfg, 5, action_dim=4, reward_prediction=True)
    (tmp_path / "transformer").mkdir()
    if load_internal_llm:    # an action-free pretrained transformer: bare HF keys
        sd = {k[len("llm."):]: v for k, v in full.items() if k.startswith("llm.")}
    else:
        sd = full
    save_file({k: v.contiguous() for k, v in sd.items()}, str(tmp_path / "transformer" / "model.safetensors"))
    args = dict(load_pretrained_model=True, config_name=str(tmp_path / "configs" / "llama" / "config.json"), vqgan_type="ctx_vqgan",
                pretrained_model_name_or_path=str(tmp_path / "tokenizer"), pretrained_transformer_path=str(tmp_path / "transformer"),
                load_internal_llm=load_internal_llm, llama_attn_drop=0.1, symlog=True, context_length=2, segment_length=12, action_dim=4)
    return args, tcfg, tsd, lcfg, full
