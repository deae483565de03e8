"""Checkpoint schema of the iVideoGPT models (host side: shapes, validation, seeded random init).

The key schema is the reference's on-disk layout:
  tokenizer   = diffusers ``ModelMixin`` state dict of ``CompressiveVQModel``
                (/root/reference/ivideogpt/vq_model/compressive_vq_model.py:33-152),
  transformer = HF ``LlamaForCausalLM`` state dict, optionally wrapped by ``HeadModelWithAction``
                (``llm.`` prefix + ``action_linear`` / ``reward_linear``;
                /root/reference/ivideogpt/transformer/action_model.py:9-45).

No pretrained checkpoint is available offline, so benchmarks and tests use ``random_*_state_dict``
(seeded, CPU generator => identical tensors on every machine with the same torch build).
"""
from collections import OrderedDict
import json
import math
import os

import torch

TOKENIZER_DEFAULTS = dict(
    in_channels=3, out_channels=3, block_out_channels=(64,), layers_per_block=1, act_fn="silu",
    latent_channels=3, sample_size=32, num_vq_embeddings=256, norm_num_groups=32, vq_embed_dim=None,
    scaling_factor=0.18215, norm_type="group", mid_block_add_attention=True, lookup_from_codebook=False,
    force_upcast=False, num_dyn_embeddings=256, context_length=1, max_att_resolution=32, resolution=256,
    patch_size=4,
)  # CompressiveVQModel.__init__ defaults, compressive_vq_model.py:36-60

# configs/ctx_vae64/config.json and configs/ctx_vae/config.json of the reference (shape facts only)
CTX_VAE64 = dict(block_out_channels=(128, 256, 512), layers_per_block=2, latent_channels=64,
                 num_vq_embeddings=8192, num_dyn_embeddings=8192, norm_num_groups=32,
                 mid_block_add_attention=False, context_length=2, resolution=64, max_att_resolution=16)
CTX_VAE256 = dict(block_out_channels=(128, 256, 256, 512, 768), layers_per_block=2, latent_channels=64,
                  num_vq_embeddings=8192, num_dyn_embeddings=8192, norm_num_groups=32,
                  mid_block_add_attention=False, context_length=2)
# configs/llama/config.json, config_medium.json (vocab overwritten to 8192+8192+2, train_gpt.py:144-146)
LLAMA_SMALL = dict(hidden_size=768, intermediate_size=3072, num_hidden_layers=12, num_attention_heads=12,
                   num_key_value_heads=12, rms_norm_eps=1e-6, rope_theta=10000.0, max_position_embeddings=1024,
                   vocab_size=16386)
LLAMA_MEDIUM = dict(hidden_size=1024, intermediate_size=4096, num_hidden_layers=24, num_attention_heads=16,
                    num_key_value_heads=16, rms_norm_eps=1e-6, rope_theta=10000.0, max_position_embeddings=1024,
                    vocab_size=16386)


def tokenizer_config(**overrides):
    cfg = dict(TOKENIZER_DEFAULTS)
    cfg.update(overrides)
    cfg["block_out_channels"] = tuple(cfg["block_out_channels"])
    if cfg["vq_embed_dim"] is None:
        cfg["vq_embed_dim"] = cfg["latent_channels"]
    return cfg


def cross_attention_sites(cfg):
    """(encoder sites, decoder sites); each site = (channels, side).  conditional_vae.py:86-102,163-181."""
    chans = list(cfg["block_out_channels"])
    res, enc = cfg["resolution"], []
    for i, c in enumerate(chans):
        if i != len(chans) - 1:
            res //= 2
        if res <= cfg["max_att_resolution"]:
            enc.append((c, res))
    rev = chans[::-1]
    res, dec = 16, [(rev[0], 16)]
    for i, c in enumerate(rev):
        if i != len(rev) - 1:
            res *= 2
        if res <= cfg["max_att_resolution"]:
            dec.append((c, res))
    return enc, dec


def tokenizer_param_shapes(cfg):
    cfg = tokenizer_config(**cfg)
    chans, lpb = list(cfg["block_out_channels"]), cfg["layers_per_block"]
    lat, dim, ctx, p = cfg["latent_channels"], cfg["vq_embed_dim"], cfg["context_length"], cfg["patch_size"]
    S = OrderedDict()

    def conv(n, cin, cout, k):
        S[n + ".weight"], S[n + ".bias"] = (cout, cin, k, k), (cout,)

    def lin(n, cin, cout):
        S[n + ".weight"], S[n + ".bias"] = (cout, cin), (cout,)

    def norm(n, c):
        S[n + ".weight"], S[n + ".bias"] = (c,), (c,)

    def resnet(n, cin, cout):
        norm(n + ".norm1", cin); conv(n + ".conv1", cin, cout, 3)
        norm(n + ".norm2", cout); conv(n + ".conv2", cout, cout, 3)
        if cin != cout:
            conv(n + ".conv_shortcut", cin, cout, 1)

    def mid(n, c, attn):
        if attn:
            norm(n + ".attentions.0.group_norm", c)
            for t in ("to_q", "to_k", "to_v", "to_out.0"):
                lin(n + ".attentions.0." + t, c, c)
        resnet(n + ".resnets.0", c, c); resnet(n + ".resnets.1", c, c)

    def xatt(n, c, side):
        S[n + ".kv_pos_emb"], S[n + ".q_pos_emb"] = (ctx * side * side, c), (side * side, c)
        S[n + ".att.in_proj_weight"], S[n + ".att.in_proj_bias"] = (3 * c, c), (3 * c,)
        lin(n + ".att.out_proj", c, c)
        norm(n + ".kv_norm", c); norm(n + ".q_norm", c)

    def encoder(n, attn):
        conv(n + ".conv_in", cfg["in_channels"], chans[0], 3)
        prev = chans[0]
        for i, c in enumerate(chans):
            for j in range(lpb):
                resnet(f"{n}.down_blocks.{i}.resnets.{j}", prev if j == 0 else c, c)
            if i != len(chans) - 1:
                conv(f"{n}.down_blocks.{i}.downsamplers.0.conv", c, c, 3)
            prev = c
        mid(n + ".mid_block", chans[-1], attn)
        norm(n + ".conv_norm_out", chans[-1]); conv(n + ".conv_out", chans[-1], lat, 3)

    def decoder(n, attn):
        rev = chans[::-1]
        conv(n + ".conv_in", lat, rev[0], 3)
        mid(n + ".mid_block", rev[0], attn)
        prev = rev[0]
        for i, c in enumerate(rev):
            for j in range(lpb + 1):
                resnet(f"{n}.up_blocks.{i}.resnets.{j}", prev if j == 0 else c, c)
            if i != len(rev) - 1:
                conv(f"{n}.up_blocks.{i}.upsamplers.0.conv", c, c, 3)
            prev = c
        norm(n + ".conv_norm_out", chans[0]); conv(n + ".conv_out", chans[0], cfg["out_channels"], 3)

    enc_sites, dec_sites = cross_attention_sites(cfg)
    encoder("cond_encoder", True)
    for k, (c, side) in enumerate(enc_sites):
        xatt(f"cond_encoder.cross_att_blocks.{k}", c, side)
    encoder("encoder", cfg["mid_block_add_attention"])
    conv("quant_conv", lat, dim, 1)
    S["quantize.embedding.weight"] = (cfg["num_vq_embeddings"], dim)
    conv("post_quant_conv", dim, lat, 1)
    lin("quant_linear", lat * p * p, dim)
    S["dynamics_quantize.embedding.weight"] = (cfg["num_dyn_embeddings"], dim)
    lin("post_quant_linear", dim, lat * p * p)
    decoder("cond_decoder", True)
    for k, (c, side) in enumerate(dec_sites):
        xatt(f"cond_decoder.cross_att_blocks.{k}", c, side)
    decoder("decoder", cfg["mid_block_add_attention"])
    return S


def llama_param_shapes(cfg, action_dim=None, reward_prediction=False, action_recon=False):
    """HF Llama keys; with ``action_dim`` the HeadModelWithAction wrapper keys (``llm.`` prefix)."""
    H, I, V = cfg["hidden_size"], cfg["intermediate_size"], cfg["vocab_size"]
    pre = "llm." if action_dim is not None else ""
    S = OrderedDict()
    S[pre + "model.embed_tokens.weight"] = (V, H)
    for l in range(cfg["num_hidden_layers"]):
        b = f"{pre}model.layers.{l}."
        for t in ("q_proj", "k_proj", "v_proj", "o_proj"):
            S[b + f"self_attn.{t}.weight"] = (H, H)
        S[b + "mlp.gate_proj.weight"], S[b + "mlp.up_proj.weight"], S[b + "mlp.down_proj.weight"] = (I, H), (I, H), (H, I)
        S[b + "input_layernorm.weight"], S[b + "post_attention_layernorm.weight"] = (H,), (H,)
    S[pre + "model.norm.weight"] = (H,)
    S[pre + "lm_head.weight"] = (V, H)
    if action_dim is not None:
        S["action_linear.weight"], S["action_linear.bias"] = (H, action_dim), (H,)
        if reward_prediction:
            S["reward_linear.weight"], S["reward_linear.bias"] = (1, H), (1,)
        if action_recon:
            S["action_recon_linear.weight"], S["action_recon_linear.bias"] = (action_dim, H), (action_dim,)
    return S


def count_params(shapes):
    return sum(math.prod(s) for s in shapes.values())


def _draw(g, name, shape, codebook_std):
    if name.endswith("embedding.weight") and "quantize" in name:
        if codebook_std is None:  # diffusers default: U(-1/n_e, 1/n_e)
            return (torch.rand(shape, generator=g) * 2 - 1) / shape[0]
        return torch.randn(shape, generator=g) * codebook_std
    if name.endswith("pos_emb"):
        return torch.randn(shape, generator=g) * 0.05
    is_norm = any(t in name for t in ("norm1.", "norm2.", "group_norm.", "kv_norm.", "q_norm.", "conv_norm_out.",
                                      "layernorm.", "model.norm."))
    if is_norm:
        if name.endswith(".weight"):
            return 1.0 + 0.1 * torch.randn(shape, generator=g)
        return 0.1 * torch.randn(shape, generator=g)
    if name.endswith("embed_tokens.weight"):
        return torch.randn(shape, generator=g) * 0.5
    if len(shape) >= 2:
        fan_in = math.prod(shape[1:])
        b = 1.0 / math.sqrt(fan_in)
        return (torch.rand(shape, generator=g) * 2 - 1) * b * (2.0 if "lm_head" not in name else 4.0)
    return (torch.rand(shape, generator=g) * 2 - 1) * 0.05  # biases


def random_state_dict(shapes, seed, codebook_std=None):
    """Seeded fp32 CPU tensors for every key, drawn in key order from one generator.
    (Non-trivial norm affine / bias / position-embedding / action weights so parity tests see them.)"""
    g = torch.Generator(device="cpu")
    g.manual_seed(seed)
    return OrderedDict((n, _draw(g, n, tuple(s), codebook_std).contiguous()) for n, s in shapes.items())


def random_tokenizer_state_dict(cfg, seed=0, codebook_std=None):
    return random_state_dict(tokenizer_param_shapes(cfg), seed, codebook_std)


def random_llama_state_dict(cfg, seed=0, action_dim=None, reward_prediction=False, action_recon=False):
    return random_state_dict(llama_param_shapes(cfg, action_dim, reward_prediction, action_recon), seed)


def validate_state_dict(sd, shapes, what):
    missing = [k for k in shapes if k not in sd]
    unexpected = [k for k in sd if k not in shapes]
    bad = [k for k in shapes if k in sd and tuple(sd[k].shape) != tuple(shapes[k])]
    if missing or unexpected or bad:
        raise RuntimeError(f"{what}: state dict does not match the schema: missing={missing[:5]} "
                           f"unexpected={unexpected[:5]} shape-mismatch={bad[:5]}")


# --------------------------------------------------------------------------- checkpoint I/O
_DF_LEGACY_ATTN = {"query": "to_q", "key": "to_k", "value": "to_v", "proj_attn": "to_out.0"}


def remap_legacy_attention_keys(sd):
    """diffusers remaps the deprecated AttentionBlock names on load."""
    out = OrderedDict()
    for k, v in sd.items():
        parts = k.split(".")
        if "attentions" in parts:
            i = parts.index("attentions")
            if len(parts) > i + 2 and parts[i + 2] in _DF_LEGACY_ATTN:
                parts[i + 2:i + 3] = _DF_LEGACY_ATTN[parts[i + 2]].split(".")
                k = ".".join(parts)
        out[k] = v
    return out


def load_tokenizer_checkpoint(path, subfolder=None):
    """<path>/<subfolder>/{config.json, diffusion_pytorch_model.safetensors} -> (config, state dict)."""
    from safetensors.torch import load_file
    d = os.path.join(path, subfolder) if subfolder else path
    with open(os.path.join(d, "config.json")) as f:
        raw = json.load(f)
    cfg = tokenizer_config(**{k: v for k, v in raw.items() if k in TOKENIZER_DEFAULTS})
    sd = remap_legacy_attention_keys(load_file(os.path.join(d, "diffusion_pytorch_model.safetensors")))
    validate_state_dict(sd, tokenizer_param_shapes(cfg), "tokenizer")
    return cfg, sd


def save_tokenizer_checkpoint(path, cfg, sd, subfolder=None):
    from safetensors.torch import save_file
    d = os.path.join(path, subfolder) if subfolder else path
    os.makedirs(d, exist_ok=True)
    out = dict(cfg)
    out["block_out_channels"] = list(out["block_out_channels"])
    out.update(_class_name="CompressiveVQModel", _diffusers_version="0.27.0")
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(out, f, indent=2)
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(d, "diffusion_pytorch_model.safetensors"))


def load_llama_config(path):
    """``AutoConfig.from_pretrained(path)`` for the Llama configs the reference ships (configs/llama/config.json,
    config_medium.json): a ``config.json`` file or the directory holding one -> config dict (unknown keys dropped)."""
    p = os.fspath(path)
    with open(os.path.join(p, "config.json") if os.path.isdir(p) else p) as f:
        raw = json.load(f)
    cfg = dict(LLAMA_SMALL)
    cfg.update({k: raw[k] for k in cfg if k in raw})
    if "rope_theta" not in raw and isinstance(raw.get("rope_parameters"), dict):
        cfg["rope_theta"] = raw["rope_parameters"].get("rope_theta", cfg["rope_theta"])
    return cfg


def load_transformer_checkpoint(path, subfolder="transformer"):
    """<path>/<subfolder>/{config.json, model.safetensors} -> (llama config dict, state dict)."""
    from safetensors.torch import load_file
    d = os.path.join(path, subfolder) if subfolder else path
    with open(os.path.join(d, "config.json")) as f:
        raw = json.load(f)
    cfg = dict(LLAMA_SMALL)
    cfg.update({k: raw[k] for k in cfg if k in raw})
    if "rope_theta" not in raw and isinstance(raw.get("rope_parameters"), dict):
        cfg["rope_theta"] = raw["rope_parameters"].get("rope_theta", cfg["rope_theta"])
    return cfg, load_file(os.path.join(d, "model.safetensors"))


def save_transformer_checkpoint(path, cfg, sd, subfolder="transformer"):
    from safetensors.torch import save_file
    d = os.path.join(path, subfolder) if subfolder else path
    os.makedirs(d, exist_ok=True)
    out = dict(cfg)
    out.update(model_type="llama", architectures=["LlamaForCausalLM"], hidden_act="silu", tie_word_embeddings=False)
    with open(os.path.join(d, "config.json"), "w") as f:
        json.dump(out, f, indent=2)
    save_file({k: v.contiguous() for k, v in sd.items()}, os.path.join(d, "model.safetensors"))
