#!/usr/bin/env python
"""Pin the oracle against the reference and (re)generate tests/golden/*.  BUILD CONTAINER ONLY.

Run from the repo root:   python oracle/pin/pin_against_reference.py

It imports the reference from /root/reference (which never travels to the GPU box):

  * ``ivideogpt.transformer.HeadModelWithAction`` + HF ``LlamaForCausalLM``  -> pins oracle/llama.py
  * ``ivideogpt.vq_model.CompressiveVQModel`` (the reference's unmodified repo-owned code),
    executed over a throw-away ``diffusers`` shim written to /tmp that re-exports
    ``oracle.df_blocks``  -> pins oracle/vq_tokenizer.py (everything except the DF blocks).

Every fixture holds only inputs + outputs produced BY THE REFERENCE CLASSES; weights are
re-derived from (config, seed) by ``ivideogpt_amd.weights.random_*_state_dict``.
"""
import json
import os
import sys
import textwrap

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF = "/root/reference"
SHIM = "/tmp/ivg_df_shim"
GOLD = os.path.join(ROOT, "tests", "golden")

from ivideogpt_amd import weights as W  # noqa: E402
from oracle import llama as OL          # noqa: E402
from oracle import vq_tokenizer as OT   # noqa: E402


def write_shim():
    files = {
        "diffusers/__init__.py": "",
        "diffusers/utils/__init__.py": """
            from collections import OrderedDict
            class BaseOutput(OrderedDict):
                def __post_init__(self):
                    for k, v in self.__dict__.items():
                        self[k] = v
            def is_torch_version(op, v):
                return True
        """,
        "diffusers/utils/torch_utils.py": "def randn_tensor(*a, **k):\n    raise NotImplementedError\n",
        "diffusers/utils/accelerate_utils.py": "def apply_forward_hook(f):\n    return f\n",
        "diffusers/configuration_utils.py": """
            import functools, inspect
            class ConfigMixin:
                pass
            def register_to_config(init):
                @functools.wraps(init)
                def inner(self, *args, **kwargs):
                    sig = inspect.signature(init)
                    ba = sig.bind(self, *args, **kwargs); ba.apply_defaults()
                    self.config = {k: v for k, v in ba.arguments.items() if k != 'self'}
                    init(self, *args, **kwargs)
                return inner
        """,
        "diffusers/models/__init__.py": "",
        "diffusers/models/modeling_utils.py": "import torch.nn as nn\nclass ModelMixin(nn.Module):\n    pass\n",
        "diffusers/models/activations.py": """
            import torch.nn as nn
            def get_activation(name):
                assert name in ('silu', 'swish')
                return nn.SiLU()
        """,
        "diffusers/models/attention_processor.py": "class SpatialNorm:\n    pass\n",
        "diffusers/models/unets/__init__.py": "",
        "diffusers/models/unets/unet_2d_blocks.py": """
            from oracle.df_blocks import UNetMidBlock2D, get_down_block, get_up_block
            class AutoencoderTinyBlock:
                pass
        """,
        "diffusers/models/autoencoders/__init__.py": "",
        "diffusers/models/autoencoders/vae.py": "from oracle.df_blocks import VectorQuantizer\n",
        "torchvision_stub/torchvision/__init__.py": "",
        "torchvision_stub/torchvision/models.py": "",
        "torchvision_stub/torchvision/transforms/__init__.py": "",
        # torchvision 0.17 tensor path of resize / center_crop (functional.py: interpolate(..., antialias=True); crop by slicing)
        "torchvision_stub/torchvision/transforms/functional.py": """
            import torch
            def resize(img, size, *a, **k):
                return torch.nn.functional.interpolate(img, size=list(size), mode='bilinear', align_corners=False, antialias=True)
            def center_crop(img, output_size):
                s = output_size if isinstance(output_size, int) else output_size[0]
                h, w = img.shape[-2:]
                top, left = int(round((h - s) / 2.0)), int(round((w - s) / 2.0))
                return img[..., top:top + s, left:left + s]
        """,
    }
    for rel, body in files.items():
        p = os.path.join(SHIM, rel)
        os.makedirs(os.path.dirname(p), exist_ok=True)
        with open(p, "w") as f:
            f.write(textwrap.dedent(body))


def import_reference():
    import transformers  # noqa: F401  resolve Llama BEFORE the torchvision stub is visible (SURVEY 8c trap)
    from transformers import LlamaConfig, LlamaForCausalLM  # noqa: F401
    write_shim()
    sys.path.insert(0, SHIM)
    sys.path.insert(0, REF)
    from ivideogpt.transformer import HeadModelWithAction
    sys.path.insert(0, os.path.join(SHIM, "torchvision_stub"))
    from ivideogpt.vq_model import CompressiveVQModel
    return CompressiveVQModel, HeadModelWithAction


def save(name, **arrays):
    os.makedirs(GOLD, exist_ok=True)
    path = os.path.join(GOLD, name)
    np.savez_compressed(path, **{k: (v.numpy() if torch.is_tensor(v) else v) for k, v in arrays.items()})
    print(f"  wrote {os.path.relpath(path, ROOT)}  ({os.path.getsize(path) / 1024:.0f} KiB)")


# ----------------------------------------------------------------------------- tokenizer
def seeded_pixels(seed, shape):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, shape, generator=g).float() / 255.0   # u8-exact values, like real clips


def pin_tokenizer(CompressiveVQModel, name, cfg, seed, B, T, codebook_std, ctx_override=None, subsample=1):
    print(f"[tokenizer] {name}: {cfg}")
    full = W.tokenizer_config(**cfg)
    sd = W.random_tokenizer_state_dict(full, seed, codebook_std)
    kw = {k: (list(v) if isinstance(v, tuple) else v) for k, v in full.items()}
    nlev = len(full["block_out_channels"])
    kw["down_block_types"], kw["up_block_types"] = ["DownEncoderBlock2D"] * nlev, ["UpDecoderBlock2D"] * nlev
    kw["vq_embed_dim"] = None if cfg.get("vq_embed_dim") is None else cfg["vq_embed_dim"]
    ref = CompressiveVQModel(**kw).eval()
    ref.load_state_dict(sd, strict=True)            # proves the key schema == the reference's
    n_ref = sum(p.numel() for p in ref.parameters())
    assert n_ref == W.count_params(W.tokenizer_param_shapes(full)), "parameter count mismatch"
    ora = OT.CompressiveVQRef(**full).eval()
    ora.load_state_dict(sd, strict=True)
    ctx = full["context_length"]
    if ctx_override is not None:
        ref.set_context_length(ctx_override); ora.set_context_length(ctx_override); ctx = ctx_override
    res = full["resolution"]
    px = seeded_pixels(seed + 1, (B, T, 3, res, res))
    with torch.no_grad():
        ids_ref, lab_ref = ref.tokenize(px, ctx)
        ids_ora, lab_ora = ora.tokenize(px, ctx)
        assert torch.equal(ids_ref, ids_ora) and torch.equal(lab_ref, lab_ora), "tokenize mismatch vs reference"
        # detokenize a *perturbed* token sequence too (out-of-range dyn ids exercise the clamp, :234-236)
        g = torch.Generator().manual_seed(seed + 2)
        ids2 = ids_ref.clone()
        vocab = full["num_vq_embeddings"] + full["num_dyn_embeddings"] + 2
        flip = torch.rand(ids2.shape, generator=g) < 0.2
        flip[:, :ctx * 257] = False                 # the reference does not clamp context ids (IndexError)
        ids2[flip] = torch.randint(0, vocab, ids2.shape, generator=g)[flip]
        rec_ref, rec_ora = ref.detokenize(ids_ref, ctx), ora.detokenize(ids_ref, ctx)
        rec2_ref, rec2_ora = ref.detokenize(ids2, ctx), ora.detokenize(ids2, ctx)
        assert torch.equal(rec_ref, rec_ora) and torch.equal(rec2_ref, rec2_ora), "detokenize mismatch vs reference"
        # the F=1 cache path (mbrl/video_predictor.py:320-321)
        one = ids_ref[:, :ctx * 257 + 16]
        r1, cache = ref.detokenize(one, ctx, return_cache=True)
        r1c = ref.detokenize(one, ctx, cache=cache)
        assert torch.equal(r1, r1c)
        st = ora.encode_stages(px, ctx)
        zc = st["hq"].permute(0, 2, 3, 1).reshape(-1, full["vq_embed_dim"])
        b0, b1, _ = OT.vq_margin(zc, sd["quantize.embedding.weight"], st["idx_c"])
        print(f"  params {n_ref / 1e6:.3f} M, tokens {tuple(ids_ref.shape)}, latent std {zc.std():.3f}, "
              f"min VQ margin {(b1 - b0).min():.2e}, median {(b1 - b0).median():.2e}")
    save(f"tok_{name}.npz", config=json.dumps({k: (list(v) if isinstance(v, tuple) else v) for k, v in full.items()}),
         seed=seed, codebook_std=codebook_std, context_length=ctx, pixels_u8=(px * 255).round().to(torch.uint8),
         indices=ids_ref, labels=lab_ref, recon=rec_ref[..., ::subsample, ::subsample],
         indices_perturbed=ids2, recon_perturbed=rec2_ref[..., ::subsample, ::subsample],
         latent_ctx=st["hq"], latent_dyn=st["dq"], subsample=subsample)


# ----------------------------------------------------------------------------- transformer
def pin_llama(HeadModelWithAction, name, cfg, seed, B, ctx, F, action_dim):
    from transformers import LlamaConfig, LlamaForCausalLM
    print(f"[llama] {name}: {cfg}")
    hf_cfg = LlamaConfig(**{**cfg, "hidden_act": "silu", "tie_word_embeddings": False, "attention_bias": False,
                            "bos_token_id": 50256, "eos_token_id": 50256})
    V, per = cfg["vocab_size"], 17
    g = torch.Generator().manual_seed(seed + 10)
    L0 = 257 * ctx
    n_new = per * F - 1
    prompt = torch.randint(0, 8192, (B, L0), generator=g)
    prompt[:, -1] = V - 1
    for c in range(1, ctx):
        prompt[:, 257 * c - 1] = V - 2

    # ---- action-free: HF LlamaForCausalLM
    sd = W.random_llama_state_dict(cfg, seed)
    hf = LlamaForCausalLM(hf_cfg).to(torch.float32).eval()
    hf.load_state_dict(sd, strict=True)
    ora = OL.LlamaRef(sd, cfg["num_hidden_layers"], cfg["num_attention_heads"], cfg["rms_norm_eps"],
                      cfg["rope_theta"], cfg["max_position_embeddings"])
    with torch.no_grad():
        full = torch.cat([prompt, torch.randint(0, V, (B, 40), generator=g)], 1)
        lg_ref = hf(input_ids=full).logits.float()
        lg_ora = ora.logits(full)
        err = (lg_ref - lg_ora).abs().max().item()
        print(f"  teacher-forced logits: max|HF - oracle| = {err:.2e} (scale {lg_ref.abs().max():.2f})")
        assert err < 2e-4
        gen_ref = hf.generate(prompt, do_sample=False, max_new_tokens=n_new, pad_token_id=50256)
        gen_ora = OL.generate_cached(ora, prompt, n_new)
        assert torch.equal(gen_ref, gen_ora), "greedy action-free rollout mismatch vs HF generate"
    save(f"llama_{name}_free.npz", config=json.dumps(cfg), seed=seed, prompt=prompt, teacher_ids=full,
         teacher_logits_last=lg_ref[:, -2:], teacher_logits_sub=lg_ref[:, ::37, ::101], greedy=gen_ref)

    # ---- action-conditioned: the reference's HeadModelWithAction (per-frame re-prefill)
    sda = W.random_llama_state_dict(cfg, seed + 1, action_dim=action_dim)
    llm = LlamaForCausalLM(hf_cfg).to(torch.float32).eval()
    T = ctx + F
    head = HeadModelWithAction(llm, action_dim=action_dim, prelude_tokens_num=L0 - 1, tokens_num_per_dyna=16,
                               context=ctx, segment_length=T).eval()
    head.load_state_dict(sda, strict=True)
    action = torch.randn(B, T, action_dim, generator=g)
    oraa = OL.LlamaRef(sda, cfg["num_hidden_layers"], cfg["num_attention_heads"], cfg["rms_norm_eps"],
                       cfg["rope_theta"], cfg["max_position_embeddings"], prefix="llm.model.")
    with torch.no_grad():
        out_ref = head.generate(prompt, do_sample=False, max_new_tokens=n_new, action=action)
        ae = torch.nn.functional.linear(action, sda["action_linear.weight"], sda["action_linear.bias"])
        out_a = OL.generate_reference_algorithm(oraa, prompt, n_new, action_embeds=ae, ctx=ctx, sdf_token=V - 1)
        out_b = OL.generate_cached(oraa, prompt, n_new, action_embeds=ae, ctx=ctx, sdf_token=V - 1)
        assert torch.equal(out_ref, out_a), "oracle re-prefill algorithm != reference HeadModelWithAction.generate"
        assert torch.equal(out_ref, out_b), "single-prefill cached algorithm != reference"
        print(f"  action-conditioned greedy: {tuple(out_ref.shape)} tokens identical (re-prefill and cached)")
        # teacher-forced forward of the reference head (action_model.py:154-185): action embeddings on every sdf slot
        fw_ref = head(input_ids=out_ref, action=action).logits.float()
        x = oraa.embed(out_ref).clone()
        start = (L0 - 1) + torch.arange(F) * per                      # start_index, :176-178
        x[:, start] += ae[:, ctx - 1:-1]
        fw_ora = oraa.logits(embeds=x)
        err = (fw_ref - fw_ora).abs().max().item()
        print(f"  HeadModelWithAction.forward logits: max|reference - oracle| = {err:.2e}")
        assert err < 2e-4
    save(f"llama_{name}_act.npz", config=json.dumps(cfg), seed=seed + 1, action_dim=action_dim, prompt=prompt,
         action=action, greedy=out_ref, ctx=ctx, forward_logits_last=fw_ref[:, -2:], forward_logits_sub=fw_ref[:, ::37, ::101])


def pin_mbrl_step(HeadModelWithAction, name, cfg, seed, B, ctx, action_dim, n_steps=2):
    """The per-step op sequence of the reference's MBRL rollout (mbrl/video_predictor.py:293-317), greedy so that no RNG is
    involved: action added to the embedding of the last token (an sdf slot), ``llm.generate(inputs_embeds=..., max_new_tokens=17,
    output_hidden_states=True)``, reward = ``reward_linear`` of the last layer's hidden state of the LAST generation step,
    16 predicted tokens + a forced sdf appended.  Pins the oracle's ``generate_cached(..., return_last_hidden=True)`` with a
    growing prompt (``slot0``) and writes the tokens / rewards of every step."""
    from transformers import LlamaConfig, LlamaForCausalLM
    print(f"[mbrl step] {name}")
    hf_cfg = LlamaConfig(**{**cfg, "hidden_act": "silu", "tie_word_embeddings": False, "attention_bias": False,
                            "bos_token_id": 50256, "eos_token_id": 50256})
    V = cfg["vocab_size"]
    g = torch.Generator().manual_seed(seed + 20)
    sd = W.random_llama_state_dict(cfg, seed, action_dim=action_dim, reward_prediction=True)
    llm = LlamaForCausalLM(hf_cfg).to(torch.float32).eval()
    head = HeadModelWithAction(llm, action_dim=action_dim, prelude_tokens_num=257 * ctx - 1, tokens_num_per_dyna=16, context=ctx,
                               segment_length=ctx + n_steps + 1, reward_prediction=True).eval()
    head.load_state_dict(sd, strict=True)
    ora = OL.LlamaRef(sd, cfg["num_hidden_layers"], cfg["num_attention_heads"], cfg["rms_norm_eps"], cfg["rope_theta"],
                      cfg["max_position_embeddings"], prefix="llm.model.")
    prompt = torch.randint(0, 8192, (B, 257 * ctx), generator=g)
    prompt[:, -1] = V - 1
    for c in range(1, ctx):
        prompt[:, 257 * c - 1] = V - 2
    actions = torch.randn(n_steps, B, action_dim, generator=g)
    act_table = torch.zeros(B, ctx - 1 + n_steps + 1, action_dim)
    tokens = prompt
    step_tokens, step_rewards = [], []
    with torch.no_grad():
        embeds = head.get_input_embeddings(prompt)
        for t in range(n_steps):
            embeds = embeds.clone()
            embeds[:, -1] += head.action_linear(actions[t])                                              # :295-296
            res = head.llm.generate(inputs_embeds=embeds, do_sample=False, pad_token_id=50256, use_cache=True, max_new_tokens=17,
                                    return_dict_in_generate=True, output_hidden_states=True)              # :298-308
            pred = res.sequences[:, :-1]                                                                  # :310
            reward = head.reward_linear(res.hidden_states[-1][-1]).squeeze(-2)                            # :311-313
            cat = torch.cat([pred, torch.full((B, 1), V - 1, dtype=pred.dtype)], 1)
            embeds = torch.cat([embeds, head.get_input_embeddings(cat)], 1)                               # :315-316
            # oracle: same step on token ids + the action table the engine's callers keep
            act_table[:, ctx - 1 + t] = actions[t]
            ae = torch.nn.functional.linear(act_table, sd["action_linear.weight"], sd["action_linear.bias"])
            out, hid = OL.generate_cached(ora, tokens, 17, uniforms=None, action_embeds=ae, ctx=ctx, sdf_token=V - 1,
                                          return_last_hidden=True)
            r_ora = torch.nn.functional.linear(hid, sd["reward_linear.weight"], sd["reward_linear.bias"])
            assert torch.equal(out[:, tokens.shape[1]:tokens.shape[1] + 16], pred), f"step {t}: tokens differ from the reference"
            err = (r_ora - reward).abs().max().item()
            print(f"  step {t}: 16 tokens identical, reward max|reference - oracle| = {err:.2e}")
            assert err < 1e-4
            tokens = torch.cat([tokens, pred, torch.full((B, 1), V - 1, dtype=tokens.dtype)], 1)
            step_tokens.append(pred)
            step_rewards.append(reward.reshape(B))
    save(f"llama_{name}_mbrl.npz", config=json.dumps(cfg), seed=seed, action_dim=action_dim, ctx=ctx, prompt=prompt,
         actions=actions, step_tokens=torch.stack(step_tokens), step_rewards=torch.stack(step_rewards))


def pin_eval_forward(HeadModelWithAction, name, cfg, seed, B, ctx, F, action_dim):
    """The eval forward (train_gpt.py:356-376): ``LlamaForCausalLM(input_ids, labels).loss`` and the reference's
    ``HeadModelWithAction(reward_prediction=True, action_recon=0.5).forward(input_ids, labels, action)`` -> loss (cross-entropy
    + action-reconstruction term, action_model.py:187-196) and ``reward_pred`` (:198-204)."""
    from transformers import LlamaConfig, LlamaForCausalLM
    print(f"[eval forward] {name}")
    hf_cfg = LlamaConfig(**{**cfg, "hidden_act": "silu", "tie_word_embeddings": False, "attention_bias": False,
                            "bos_token_id": 50256, "eos_token_id": 50256})
    V, per = cfg["vocab_size"], 17
    g = torch.Generator().manual_seed(seed + 30)
    L = 257 * ctx - 1 + per * F
    ids = torch.randint(0, V - 2, (B, L), generator=g)
    for i in range(F):
        ids[:, 257 * ctx - 1 + per * i] = V - 1
    for c in range(1, ctx):
        ids[:, 257 * c - 1] = V - 2
    labels = ids.clone()
    labels[:, :257 * ctx] = -100                       # context tokens carry no loss (compressive_vq_model.py:216-218)
    labels[0, 257 * ctx + 5] = -100                    # an ignored target in the middle
    args = dict(num_layers=cfg["num_hidden_layers"], heads=cfg["num_attention_heads"])
    # ---- action-free
    sd = W.random_llama_state_dict(cfg, seed)
    hf = LlamaForCausalLM(hf_cfg).to(torch.float32).eval()
    hf.load_state_dict(sd, strict=True)
    ora = OL.LlamaRef(sd, args["num_layers"], args["heads"], cfg["rms_norm_eps"], cfg["rope_theta"], cfg["max_position_embeddings"])
    with torch.no_grad():
        loss_ref = hf(input_ids=ids, labels=labels).loss.float()
        o = OL.eval_forward(ora, ids, labels)
        print(f"  LlamaForCausalLM loss: reference {loss_ref.item():.6f}, oracle {o['loss'].item():.6f}")
        assert abs(loss_ref.item() - o["loss"].item()) < 1e-4
    # ---- HeadModelWithAction with reward head and action reconstruction
    T = ctx + F
    sda = W.random_llama_state_dict(cfg, seed + 1, action_dim=action_dim, reward_prediction=True, action_recon=True)
    llm = LlamaForCausalLM(hf_cfg).to(torch.float32).eval()
    head = HeadModelWithAction(llm, action_dim=action_dim, prelude_tokens_num=257 * ctx - 1, tokens_num_per_dyna=16, context=ctx,
                               segment_length=T, reward_prediction=True, action_recon=0.5).eval()
    head.load_state_dict(sda, strict=True)
    action = torch.randn(B, T, action_dim, generator=g)
    oraa = OL.LlamaRef(sda, args["num_layers"], args["heads"], cfg["rms_norm_eps"], cfg["rope_theta"], cfg["max_position_embeddings"],
                       prefix="llm.model.")
    with torch.no_grad():
        x, reward_pred = head(input_ids=ids, labels=labels, action=action)
        loss_a = x.loss.float()
        ar = head.action_recon_loss.float()
        ae = torch.nn.functional.linear(action, sda["action_linear.weight"], sda["action_linear.bias"])
        o = OL.eval_forward(oraa, ids, labels, action_embeds=ae, ctx=ctx, n_future=F)
        hid = o["hidden"]
        rec = torch.nn.functional.linear(hid[:, 257 * ctx - 1:], sda["action_recon_linear.weight"], sda["action_recon_linear.bias"])
        tgt = action[:, ctx - 1:-1].unsqueeze(-2).repeat(1, 1, per, 1)
        ar_o = torch.nn.functional.mse_loss(rec.reshape(-1, F, per, action_dim), tgt)
        start = (257 * ctx - 1) + torch.arange(F) * per
        rp_o = torch.nn.functional.linear(hid[:, start + 16], sda["reward_linear.weight"], sda["reward_linear.bias"])
        loss_o = o["loss"] + 0.5 * ar_o
        print(f"  HeadModelWithAction loss: reference {loss_a.item():.6f} (action_recon {ar.item():.6f}), oracle {loss_o.item():.6f}; "
              f"reward_pred max diff {(rp_o - reward_pred).abs().max().item():.2e}")
        assert abs(loss_a.item() - loss_o.item()) < 1e-4 and abs(ar.item() - ar_o.item()) < 1e-5
        assert (rp_o - reward_pred).abs().max().item() < 1e-4
    save(f"llama_{name}_eval.npz", config=json.dumps(cfg), seed=seed, action_dim=action_dim, ctx=ctx, n_future=F, ids=ids, labels=labels,
         action=action, loss_free=loss_ref, loss_act=loss_a, action_recon_loss=ar, action_recon_weight=0.5, reward_pred=reward_pred.float())


def pin_bf16(CompressiveVQModel, name, tcfg, tseed, lcfg, lseed, codebook_std):
    """The reference's OWN bf16 path (``torch.autocast(dtype=bfloat16)``, vp/ivideogpt_interface.py:180, mbrl/video_predictor.py:269)
    run here on the CPU: tokenizer decode of the fp32 fixture's tokens and HF Llama teacher-forced logits under autocast, plus the
    same Llama with bf16 weights (``model.to(bfloat16)``).  These are the vectors the engine's bf16 mode is tested against."""
    from transformers import LlamaConfig, LlamaForCausalLM
    print(f"[bf16] {name}")
    g0 = np.load(os.path.join(GOLD, f"tok_{name}.npz"))
    cfg = W.tokenizer_config(**tcfg)
    kw = {k: (list(v) if isinstance(v, tuple) else v) for k, v in cfg.items()}
    nlev = len(cfg["block_out_channels"])
    kw["down_block_types"], kw["up_block_types"] = ["DownEncoderBlock2D"] * nlev, ["UpDecoderBlock2D"] * nlev
    kw["vq_embed_dim"] = None if tcfg.get("vq_embed_dim") is None else tcfg["vq_embed_dim"]
    ref = CompressiveVQModel(**kw).eval()
    ref.load_state_dict(W.random_tokenizer_state_dict(cfg, tseed, codebook_std), strict=True)
    ids = torch.from_numpy(g0["indices"])
    ctx = int(g0["context_length"])
    with torch.no_grad():
        rec32 = ref.detokenize(ids, ctx).float()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            rec16 = ref.detokenize(ids, ctx).float()
    d = (rec16 - rec32).abs()
    print(f"  detokenize under autocast vs fp32: max {d.max().item():.3e} mean {d.mean().item():.3e}")
    hf_cfg = LlamaConfig(**{**lcfg, "hidden_act": "silu", "tie_word_embeddings": False, "attention_bias": False,
                            "bos_token_id": 50256, "eos_token_id": 50256})
    sd = W.random_llama_state_dict(lcfg, lseed)
    hf = LlamaForCausalLM(hf_cfg).to(torch.float32).eval()
    hf.load_state_dict(sd, strict=True)
    gl = np.load(os.path.join(GOLD, "llama_tiny_ctx2_free.npz"))
    full = torch.from_numpy(gl["teacher_ids"])
    with torch.no_grad():
        lg32 = hf(input_ids=full).logits.float()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            lg_ac = hf(input_ids=full).logits.float()
        lg_bf = hf.to(torch.bfloat16)(input_ids=full).logits.float()
    print(f"  logits: autocast vs fp32 max {(lg_ac - lg32).abs().max().item():.3e}; bf16 weights vs fp32 max {(lg_bf - lg32).abs().max().item():.3e} "
          f"(scale {lg32.abs().max().item():.2f})")
    save(f"bf16_{name}.npz", pixels_autocast=rec16, pixels_fp32_sub=rec32[:, :, :, ::4, ::4], logits_autocast_last=lg_ac[:, -2:],
         logits_autocast_sub=lg_ac[:, ::37, ::101], logits_bf16_last=lg_bf[:, -2:], logits_bf16_sub=lg_bf[:, ::37, ::101],
         autocast_pixel_dev=np.array([d.max().item(), d.mean().item()]),
         logits_dev=np.array([(lg_ac - lg32).abs().max().item(), (lg_bf - lg32).abs().max().item()]))


def pin_fractal_clip():
    """BASELINE config 1: the reference's clip ingest (inference/utils.py:12-39, executed unmodified; torchvision's tensor
    ``resize`` is supplied by the shim as ``interpolate(mode='bilinear', antialias=True)``, its definition in torchvision 0.17)
    on inference/samples/fractal_sample.npz, seed 0 -> the [16, 3, 64, 64] clip the CLI feeds the tokenizer.  Pins
    ivideogpt_amd.data.NPZParser and is the input of the config-1 GPU test.  The episode itself (a data file of the reference)
    is committed next to it so the CLI test can run on the real file."""
    import importlib.util
    import shutil
    print("[fractal clip]")
    spec = importlib.util.spec_from_file_location("ref_inference_utils", os.path.join(REF, "inference", "utils.py"))
    ru = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ru)
    src = os.path.join(REF, "inference", "samples", "fractal_sample.npz")
    np.random.seed(0)
    clip_ref, _ = ru.NPZParser(16, 64).parse(src, "fractal20220817_data")
    from ivideogpt_amd.data import NPZParser, DATASETS
    np.random.seed(0)
    clip_own, _ = NPZParser(16, 64).parse(src, "fractal20220817_data")
    assert clip_ref.shape == (16, 3, 64, 64) and torch.equal(clip_ref, clip_own), "NPZParser restatement differs from the reference"
    for k, v in ru.BASE_STEPSIZE.items():
        assert DATASETS[k][0] == v, f"stride table differs for {k}"
    for k, v in ru.DISPLAY_KEY.items():
        assert DATASETS[k][1] == v, f"display key differs for {k}"
    for k in DATASETS:
        assert k in ru.BASE_STEPSIZE or k in ru.DISPLAY_KEY, f"unknown dataset {k}"
    shutil.copyfile(src, os.path.join(GOLD, "fractal_sample.npz"))
    save("fractal_clip_seed0.npz", clip=clip_ref, seed=0, dataset_name="fractal20220817_data", segment_length=16, resolution=64)


def pin_sampler():
    """oracle.llama.sample_from_logits against HF's own logits processors (the reference passes temperature / top_k straight to HF
    generate: inference/predict.py:57-69, action_model.py:101-110): TemperatureLogitsWarper -> TopKLogitsWarper -> softmax.  HF
    then calls torch.multinomial (stream not reproducible); the draw is pinned as the inverse CDF of HF's probabilities in
    ascending id order.  Writes tests/golden/sampler_temperature.npz (seeded logits are re-derived by the test)."""
    from transformers.generation.logits_process import TemperatureLogitsWarper, TopKLogitsWarper
    B, V, k = 48, 16386, 100
    g = torch.Generator().manual_seed(777)
    logits = torch.randn(B, V, generator=g) * 3
    u = torch.rand(B, generator=g)
    out = {}
    for T in (0.7, 1.0, 1.3):
        scores = logits.clone()
        if T != 1.0:
            scores = TemperatureLogitsWarper(T)(None, scores)
        scores = TopKLogitsWarper(top_k=k, filter_value=-float("inf"))(None, scores)
        probs = torch.softmax(scores, -1)
        assert int((probs > 0).sum(-1).min()) >= k
        cdf = torch.cumsum(probs.double(), -1)
        tok_hf = (cdf > (u.double() * cdf[:, -1]).view(-1, 1)).double().argmax(-1)
        tok = OL.sample_from_logits(logits, k, u, temperature=T)
        kept = OL.sample_from_logits(logits, k, None)   # greedy is temperature-free
        assert torch.equal(kept, logits.argmax(-1))
        assert torch.equal(tok, tok_hf), f"sampler at temperature {T}: oracle != HF processors"
        out[f"tok_T{T}"] = tok
    assert not torch.equal(out["tok_T0.7"], out["tok_T1.3"])
    save("sampler_temperature.npz", seed=np.int64(777), B=np.int64(B), V=np.int64(V), top_k=np.int64(k), u=u, **out)
    print("[sampler] oracle == HF TemperatureLogitsWarper + TopKLogitsWarper + softmax inverse CDF at T = 0.7 / 1.0 / 1.3")


def pin_param_counts():
    n64 = W.count_params(W.tokenizer_param_shapes(W.CTX_VAE64))
    n256 = W.count_params(W.tokenizer_param_shapes(W.CTX_VAE256))
    ns = W.count_params(W.llama_param_shapes(W.LLAMA_SMALL))
    nm = W.count_params(W.llama_param_shapes(W.LLAMA_MEDIUM))
    print(f"[params] tokenizer64 {n64 / 1e6:.3f} M (README 114 M), tokenizer256 {n256 / 1e6:.3f} M (README 310 M), "
          f"llama small {ns / 1e6:.2f} M (138 M), medium {nm / 1e6:.2f} M (436 M)")
    assert abs(n64 / 1e6 - 114.16) < 0.01 and abs(n256 / 1e6 - 310.47) < 0.01 and abs(ns / 1e6 - 138.43) < 0.01


def main():
    torch.manual_seed(0)
    CompressiveVQModel, HeadModelWithAction = import_reference()
    pin_param_counts()
    mini64 = dict(block_out_channels=(64, 128, 128), layers_per_block=1, latent_channels=64, num_vq_embeddings=512,
                  num_dyn_embeddings=512, mid_block_add_attention=False, context_length=2, resolution=64,
                  max_att_resolution=16)
    pin_tokenizer(CompressiveVQModel, "mini64_ctx2", mini64, seed=11, B=2, T=4, codebook_std=0.4)
    pin_tokenizer(CompressiveVQModel, "mini64_ctx1", mini64, seed=12, B=2, T=3, codebook_std=0.4, ctx_override=1)
    mini256 = dict(block_out_channels=(64, 64, 64, 128, 192), layers_per_block=1, latent_channels=64,
                   num_vq_embeddings=512, num_dyn_embeddings=512, mid_block_add_attention=False, context_length=2)
    pin_tokenizer(CompressiveVQModel, "mini256_ctx2", mini256, seed=13, B=1, T=3, codebook_std=0.4, subsample=4)
    tiny = dict(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=2,
                num_key_value_heads=2, rms_norm_eps=1e-6, rope_theta=10000.0, max_position_embeddings=1024,
                vocab_size=16386)
    pin_llama(HeadModelWithAction, "tiny_ctx2", tiny, seed=21, B=2, ctx=2, F=3, action_dim=4)
    pin_llama(HeadModelWithAction, "tiny_ctx1", tiny, seed=23, B=3, ctx=1, F=4, action_dim=7)
    pin_mbrl_step(HeadModelWithAction, "tiny_ctx2", tiny, seed=31, B=2, ctx=2, action_dim=4, n_steps=3)
    pin_eval_forward(HeadModelWithAction, "tiny_ctx2", tiny, seed=41, B=3, ctx=2, F=3, action_dim=4)
    pin_bf16(CompressiveVQModel, "mini64_ctx2", mini64, 11, tiny, 21, 0.4)
    pin_fractal_clip()
    pin_sampler()
    print("all pins passed")


if __name__ == "__main__":
    main()
