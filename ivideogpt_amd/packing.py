"""Checkpoint tensors -> the device layouts libivg consumes (host-side tensor plumbing, PyTorch-ROCm).

Layouts (see csrc/igemm.hip): activations are NHWC, so a conv weight [Cout, Cin, kh, kw] becomes
[Cout, kh*kw*Cin] (K contiguous, ordered (kh, kw, c)); linears stay [N, K]; q/k/v of the Llama layers are
fused into one [3H, H] matrix; gate/up are interleaved [16 gate | 16 up] per 32 rows so the GEMM epilogue can
apply SiLU(gate)*up in registers.  Norm affine parameters, biases, position embeddings and the codebooks stay fp32.
"""
import torch

from . import _lib

_TORCH_DT = {_lib.IVG_F32: torch.float32, _lib.IVG_BF16: torch.bfloat16}


def torch_dtype(code):
    return _TORCH_DT[code]


X3 = "x3"   # arithmetic mode name: fp32 tensors in HBM, split-bf16 (hi + lo) MFMAs with fp32 accumulation (csrc/conv3x3.hip, X3)


def is_x3(dt):
    return isinstance(dt, str) and dt.lower() in ("x3", "fp32x3", "f32x3")


def config_code(dt):
    """ivg_config.{encode,decode,llm}_dtype: like dtype_code, but "x3" is IVG_F32X3 (fp32 tensors, split-bf16 matrix arithmetic)."""
    return _lib.IVG_F32X3 if is_x3(dt) else dtype_code(dt)


def dtype_code(dt):
    """Element type of the tensors in HBM.  "x3" stores fp32 (only the matrix arithmetic differs)."""
    if is_x3(dt):
        return _lib.IVG_F32
    if dt in (torch.float32, "fp32", "float32", "f32", _lib.IVG_F32):
        return _lib.IVG_F32
    if dt in (torch.bfloat16, "bf16", "bfloat16", _lib.IVG_BF16):
        return _lib.IVG_BF16
    raise ValueError(f"unsupported dtype {dt!r} (float32 / bfloat16)")


def _conv(w, dt):
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).to(dt).contiguous()


def pack_x3(w):
    """fp32 [N, K] (K % 4 == 0) -> bfloat16 [N, 2K]: every 4 consecutive K elements become one 16-byte slot [hi(4) | lo(4)] with
    hi = bf16(w), lo = bf16(w - hi) -- the slot layout the X3 kernels multiply against an activation slot split the same way
    (hi * hi + hi * lo + lo * hi + lo * lo by two K = 32 bf16 MFMAs).  Same bytes per row as the fp32 matrix."""
    n, k = w.shape
    assert k % 4 == 0
    w = w.float()
    hi = w.to(torch.bfloat16)
    lo = (w - hi.float()).to(torch.bfloat16)
    return torch.stack([hi.view(n, k // 4, 4), lo.view(n, k // 4, 4)], 2).reshape(n, 2 * k).contiguous()


def pack_subpixel(w):
    """Upsampler conv weight fp32 [Cout, Cin, 3, 3] -> fp32 [4 * Cout, 4 * Cin]: the SUB-PIXEL form of ``conv3x3(nearest_x2(x))``
    (diffusers Upsample2D, /root/reference/ivideogpt/vq_model/vae.py:271-284 builds it through get_up_block).  Output pixel
    (2 iy + py, 2 ix + px) reads upsampled rows 2 iy + py - 1 .. + 1, i.e. input rows {iy - 1, iy, iy} for py = 0 and {iy, iy, iy + 1}
    for py = 1: the three row taps collapse to two, {W0, W1 + W2} resp. {W0 + W1, W2}; the columns alike.  Row block ``phase = 2 py +
    px`` holds that phase's 2 x 2 convolution as [Cout][(kh2, kw2, c)] (K contiguous like every packed conv), where tap (kh2, kw2)
    reads input pixel (iy + py + kh2 - 1, ix + px + kw2 - 1).  Summed in fp32 (rounded to the storage type by the caller)."""
    w = w.float()
    rows = ((w[:, :, 0], w[:, :, 1] + w[:, :, 2]), (w[:, :, 0] + w[:, :, 1], w[:, :, 2]))      # [py][kh2] -> [Cout, Cin, 3 (kw)]
    out = []
    for py in range(2):
        for px in range(2):
            taps = []
            for kh2 in range(2):
                r = rows[py][kh2]
                cols = (r[:, :, 0], r[:, :, 1] + r[:, :, 2]) if px == 0 else (r[:, :, 0] + r[:, :, 1], r[:, :, 2])
                taps += [cols[0], cols[1]]                                               # (kh2, kw2) order, each [Cout, Cin]
            out.append(torch.stack(taps, 1).reshape(w.shape[0], -1))                        # [Cout, 4 * Cin]
    return torch.cat(out, 0).contiguous()


def pack_tokenizer(sd, cfg, device, enc_code, dec_code, dec_x3=False):
    """DF state dict of CompressiveVQModel -> {name: device tensor} for ivg_create.  dec_x3: the 3x3 convolutions of the two
    decoders also get their weights pre-split for the split-bf16 kernels (``<name>.x3``, beside the fp32 matrix other shapes use)."""
    enc_dt, dec_dt = torch_dtype(enc_code), torch_dtype(dec_code)
    out = {}
    for k, v in sd.items():
        v = v.detach().to(device=device, dtype=torch.float32)
        top = k.split(".")[0]
        dt = enc_dt if top in ("encoder", "cond_encoder", "quant_conv", "quant_linear") else dec_dt
        if k.endswith(("pos_emb", "embedding.weight", ".bias", "in_proj_bias")) or v.dim() == 1:
            out[k] = v.contiguous()
        elif k in ("encoder.conv_in.weight", "cond_encoder.conv_in.weight"):
            out[k] = v.contiguous()                       # raw [C0, 3, 3, 3] fp32: direct first-layer kernel
        elif v.dim() == 4:
            out[k] = _conv(v, dt)
            if dec_x3 and top in ("decoder", "cond_decoder") and v.shape[2] == 3 and v.shape[1] % 16 == 0:
                out[k + ".x3"] = pack_x3(out[k])
            if top in ("decoder", "cond_decoder") and ".upsamplers." in k and v.shape[2] == 3:   # sub-pixel phase weights (conv3x3.hip SUBPIX)
                sub = pack_subpixel(v)
                out[k + ".subpix"] = sub.to(dt).contiguous()
                if dec_x3 and v.shape[1] % 16 == 0:
                    out[k + ".subpix.x3"] = pack_x3(sub)
        elif v.dim() == 2:
            out[k] = v.to(dt).contiguous()                # Linear / MHA in_proj / out_proj; quant_linear is already (ph, pw, c)
        else:
            raise ValueError(f"unexpected tensor {k} {tuple(v.shape)}")
    return out


def pack_llama(sd, cfg, device, code, prefix=""):
    """HF Llama (optionally HeadModelWithAction, prefix 'llm.') state dict -> engine tensors."""
    dt = torch_dtype(code)
    H, I, nl = cfg["hidden_size"], cfg["intermediate_size"], cfg["num_hidden_layers"]
    heads = cfg["num_attention_heads"]
    hd = H // heads

    def g(name):
        return sd[prefix + name].detach().to(device=device, dtype=torch.float32)

    def vec(t):  # HF keeps norm weights in the model dtype: round through it so the products match
        return t.to(dt).float().contiguous()

    out = {}
    for l in range(nl):
        b = f"model.layers.{l}."
        ln1, ln2 = g(b + "input_layernorm.weight"), g(b + "post_attention_layernorm.weight")
        # RMSNorm weights are folded into the consuming projection (W' = W diag(w)): the engine fuses the norm's row
        # scale into the GEMM, so no separate normalisation pass exists in the decode step
        out[f"llm.layers.{l}.wqkv"] = (torch.cat([g(b + "self_attn.q_proj.weight"), g(b + "self_attn.k_proj.weight"),
                                                  g(b + "self_attn.v_proj.weight")], 0) * ln1[None, :]).to(dt).contiguous()
        out[f"llm.layers.{l}.wo"] = g(b + "self_attn.o_proj.weight").to(dt).contiguous()
        gate, up = g(b + "mlp.gate_proj.weight"), g(b + "mlp.up_proj.weight")
        assert I % 16 == 0
        out[f"llm.layers.{l}.wgu"] = (torch.stack([gate.view(I // 16, 16, H), up.view(I // 16, 16, H)], 1)
                                      .reshape(2 * I, H) * ln2[None, :]).to(dt).contiguous()
        out[f"llm.layers.{l}.wdown"] = g(b + "mlp.down_proj.weight").to(dt).contiguous()
    out["llm.embed"] = g("model.embed_tokens.weight").to(dt).contiguous()
    fnorm = g("model.norm.weight")
    out["llm.lm_head"] = (g("lm_head.weight") * fnorm[None, :]).to(dt).contiguous()
    out["llm.norm"] = fnorm.contiguous()                  # raw: the hidden states handed back to callers are post-norm (HF)
    # RoPE tables exactly as HF builds them (fp32 inv_freq, fp32 outer product, cos/sin, cast to the model dtype)
    inv_freq = 1.0 / (cfg.get("rope_theta", 10000.0) ** (torch.arange(0, hd, 2, dtype=torch.float32) / hd))
    freqs = torch.arange(cfg["max_position_embeddings"], dtype=torch.float32)[:, None] * inv_freq[None, :]
    out["llm.rope_cos"] = vec(freqs.cos().to(device))
    out["llm.rope_sin"] = vec(freqs.sin().to(device))
    if prefix and "action_linear.weight" in sd:
        out["llm.action_linear.weight"] = sd["action_linear.weight"].detach().to(device=device, dtype=torch.float32).contiguous()
        out["llm.action_linear.bias"] = sd["action_linear.bias"].detach().to(device=device, dtype=torch.float32).contiguous()
    if prefix and "action_recon_linear.weight" in sd:
        out["llm.action_recon_linear.weight"] = sd["action_recon_linear.weight"].detach().to(device=device, dtype=torch.float32).contiguous()
        out["llm.action_recon_linear.bias"] = sd["action_recon_linear.bias"].detach().to(device=device, dtype=torch.float32).contiguous()
    if "reward_linear.weight" in sd:
        out["llm.reward_linear.weight"] = (sd["reward_linear.weight"].detach().to(device=device, dtype=torch.float32).reshape(-1)
                                           * fnorm).contiguous()   # reward head reads the final-normed hidden state
        out["llm.reward_linear.bias"] = sd["reward_linear.bias"].detach().to(device=device, dtype=torch.float32).contiguous()
        out["llm.reward_linear.raw"] = sd["reward_linear.weight"].detach().to(device=device, dtype=torch.float32).reshape(-1).contiguous()
    return out
