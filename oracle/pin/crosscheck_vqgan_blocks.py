#!/usr/bin/env python
"""Cross-check ``oracle/df_blocks.py`` against an INDEPENDENT third-party implementation of the same taming-VQGAN blocks that this
image does hold: ``transformers.models.chameleon.modeling_chameleon`` (HF's port of Chameleon's VQGAN encoder).  TEST INFRASTRUCTURE.

``diffusers==0.27.0`` (reference ``requirements.txt:6``) is absent, so ``oracle/df_blocks.py`` restates its blocks from the published
algorithm and stays "parity unpinned" (its header).  The blocks the reference assembles its tokenizer from
(``/root/reference/ivideogpt/vq_model/vae.py:104-130,250-284``, ``compressive_vq_model.py:102-123``) are the taming-transformers
VQGAN blocks; HF's Chameleon module is written by other people from the same lineage:

    oracle.df_blocks.ResnetBlock2D (GN(32, 1e-6) -> SiLU -> conv3x3 -> GN -> SiLU -> conv3x3, 1x1 skip)  <->  ChameleonVQVAEEncoderResnetBlock
    oracle.df_blocks.Downsample2D  (pad (0,1,0,1), conv3x3 stride 2 pad 0)                               <->  ChameleonVQVAEEncoderConvDownsample
    oracle.df_blocks.Attention     (GN, q/k/v/out Linear with bias, one head, softmax(QK^T/sqrt(C))V, +x) <->  ChameleonVQVAEEncoderAttnBlock (1x1 convs)
    oracle.df_blocks.VectorQuantizer (argmin of cdist)                                                   <->  ChameleonVQVAEVectorQuantizer (argmin of z^2+e^2-2ze)
    oracle.vq_tokenizer.EncoderRef (conv_in, DownEncoderBlock2D x L, UNetMidBlock2D, GN-SiLU-conv_out)   <->  ChameleonVQVAEEncoder (whole trunk)

and, for the DECODER side (Chameleon ships no decoder), a second independent implementation of the same lineage,
``transformers.models.janus.modeling_janus`` (HF's port of Janus' VQGAN):

    oracle.df_blocks.Upsample2D    (nearest x2, conv3x3 pad 1)                                            <->  JanusVQVAEConvUpsample
    oracle.vq_tokenizer.DecoderRef (conv_in, UNetMidBlock2D, UpDecoderBlock2D x L with layers_per_block+1
                                    resnets and the upsamplers on all but the last level, GN-SiLU-conv_out) <->  JanusVQVAEDecoder
    (Janus puts an attention block behind every resnet of its lowest-resolution level, which diffusers' UpDecoderBlock2D does not
    have: those blocks get a zero output projection, i.e. become the identity -- everything else of the trunk is compared.)

The same tensors go into both; outputs must agree to 1e-5 relative (fp32 summation orders differ: Linear vs 1x1 conv, SDPA vs bmm) and
the VQ assignment must be identical.  This is NOT a pin against diffusers (oracle/pin/pin_df_blocks.py does that wherever the wheel
exists); it shrinks the unpinned surface to "three restatements by different people share a misunderstanding".  Runs on CPU in
seconds; part of the CPU suite (tests/test_oracle_crosscheck.py).
"""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import df_blocks as DF          # noqa: E402
from oracle import vq_tokenizer as OT       # noqa: E402


def _cham():
    from transformers.models.chameleon import modeling_chameleon as MC
    return MC


def _cfg(**kw):
    d = dict(dropout=0.0, num_embeddings=512, embed_dim=64, beta=1.0, base_channels=64, channel_multiplier=(1, 2, 4), num_res_blocks=2,
             resolution=32, in_channels=3, double_latent=False, latent_channels=64, attn_resolutions=None, attn_type="vanilla")
    d.update(kw)
    return types.SimpleNamespace(**d)


def _rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


def _copy_resnet(dst, src):
    """oracle ResnetBlock2D -> ChameleonVQVAEEncoderResnetBlock"""
    for n in ("norm1", "conv1", "norm2", "conv2"):
        getattr(dst, n).load_state_dict(getattr(src, n).state_dict())
    if src.conv_shortcut is not None:
        dst.nin_shortcut.load_state_dict(src.conv_shortcut.state_dict())


def _copy_attn(dst, src):
    """oracle Attention (Linear [C, C]) -> ChameleonVQVAEEncoderAttnBlock (1x1 conv [C, C, 1, 1])"""
    dst.norm.load_state_dict(src.group_norm.state_dict())
    for d, s in ((dst.q, src.to_q), (dst.k, src.to_k), (dst.v, src.to_v), (dst.proj_out, src.to_out[0])):
        d.weight.data.copy_(s.weight.data[:, :, None, None])
        d.bias.data.copy_(s.bias.data)


def _randomise(m, g):
    """weights AND biases / affine parameters away from their initial values (a zero bias or unit gamma would hide a mix-up)"""
    for p in m.parameters():
        p.data.copy_(torch.randn(p.shape, generator=g) * (1.0 / float(p[0].numel()) ** 0.5 if p.dim() > 1 else 0.3))
    for mod in m.modules():
        if isinstance(mod, torch.nn.GroupNorm):
            mod.weight.data.copy_(1.0 + 0.3 * torch.randn(mod.weight.shape, generator=g))


@torch.no_grad()
def check_resnet(g):
    MC = _cham()
    out = {}
    for cin, cout in ((64, 64), (64, 128), (128, 96)):
        a = DF.ResnetBlock2D(cin, cout, eps=1e-6, groups=32).eval()
        _randomise(a, g)
        b = MC.ChameleonVQVAEEncoderResnetBlock(_cfg(), cin, cout).eval()
        _copy_resnet(b, a)
        x = torch.randn(2, cin, 12, 10, generator=g)
        out[f"resnet_{cin}_{cout}"] = _rel(a(x), b(x.clone()))
    return out


@torch.no_grad()
def check_downsample(g):
    MC = _cham()
    a = DF.Downsample2D(64).eval()
    _randomise(a, g)
    b = MC.ChameleonVQVAEEncoderConvDownsample(64).eval()
    b.conv.load_state_dict(a.conv.state_dict())
    out = {}
    for hw in ((16, 16), (9, 11)):   # odd sizes: the asymmetric (0,1,0,1) pad decides the last row / column
        x = torch.randn(2, 64, *hw, generator=g)
        ya, yb = a(x), b(x)
        assert ya.shape == yb.shape, (ya.shape, yb.shape)
        out[f"downsample_{hw[0]}x{hw[1]}"] = _rel(ya, yb)
    return out


@torch.no_grad()
def check_attention(g):
    MC = _cham()
    out = {}
    for c in (64, 128):
        a = DF.Attention(c, dim_head=c, eps=1e-6, groups=32).eval()   # heads = 1, as UNetMidBlock2D(attention_head_dim=C) builds it (vae.py:126,256)
        _randomise(a, g)
        b = MC.ChameleonVQVAEEncoderAttnBlock(c).eval()
        _copy_attn(b, a)
        x = torch.randn(2, c, 8, 8, generator=g)
        out[f"attention_{c}"] = _rel(a(x), b(x.clone()))
    return out


@torch.no_grad()
def check_vq(g):
    MC = _cham()
    out = {}
    for n_e, scale in ((512, 1.0), (8192, 0.05)):
        a = DF.VectorQuantizer(n_e, 64, beta=1.0, legacy=False)
        a.embedding.weight.data.copy_(torch.randn(n_e, 64, generator=g) * scale)
        b = MC.ChameleonVQVAEVectorQuantizer(_cfg(num_embeddings=n_e))
        b.embedding.load_state_dict(a.embedding.state_dict())
        z = torch.randn(3, 64, 16, 16, generator=g) * scale
        zq_a, _, (_, _, idx_a) = a(z)
        zq_b, _, idx_b = b(z)
        out[f"vq_{n_e}_ids_differ"] = int((idx_a != idx_b).sum())
        out[f"vq_{n_e}_zq"] = _rel(zq_a, zq_b)
    return out


@torch.no_grad()
def check_encoder_trunk(g):
    """the whole encoder as vae.py:47-195 composes it (levels, where the downsamplers sit, mid block with / without attention, tail)"""
    MC = _cham()
    out = {}
    for mid_attention in (False, True):   # `encoder` (configs/*/config.json: mid_block_add_attention=false) / `cond_encoder` trunk (compressive_vq_model.py:79)
        chans = (64, 128, 256)
        a = OT.EncoderRef(3, 64, chans, 2, 32, mid_attention).eval()
        _randomise(a, g)
        b = MC.ChameleonVQVAEEncoder(_cfg(attn_type="vanilla" if mid_attention else "none")).eval()
        b.conv_in.load_state_dict(a.conv_in.state_dict())
        for lvl, blk in enumerate(a.down_blocks):
            for j, r in enumerate(blk.resnets):
                _copy_resnet(b.down[lvl].block[j], r)
            if blk.downsamplers is not None:
                b.down[lvl].downsample.conv.load_state_dict(blk.downsamplers[0].conv.state_dict())
            else:
                assert not hasattr(b.down[lvl], "downsample"), "both put no downsampler on the last level"
        _copy_resnet(b.mid.block_1, a.mid_block.resnets[0])
        _copy_resnet(b.mid.block_2, a.mid_block.resnets[1])
        if mid_attention:
            _copy_attn(b.mid.attn_1, a.mid_block.attentions[0])
        b.norm_out.load_state_dict(a.conv_norm_out.state_dict())
        b.conv_out.load_state_dict(a.conv_out.state_dict())
        n_a = sum(p.numel() for p in a.parameters())
        n_b = sum(p.numel() for p in b.parameters())
        assert n_a == n_b, (n_a, n_b)
        x = torch.rand(2, 3, 32, 32, generator=g)
        za, feats = a(x)
        zb = b(x)
        assert za.shape == zb.shape == (2, 64, 8, 8)
        out[f"encoder_trunk_mid_attention={int(mid_attention)}"] = _rel(za, zb)
    return out


def _janus():
    from transformers.models.janus import modeling_janus as MJ
    return MJ


@torch.no_grad()
def check_upsample(g):
    MJ = _janus()
    a = DF.Upsample2D(64).eval()
    _randomise(a, g)
    b = MJ.JanusVQVAEConvUpsample(64).eval()
    b.conv.load_state_dict(a.conv.state_dict())
    out = {}
    for hw in ((8, 8), (5, 7)):
        x = torch.randn(2, 64, *hw, generator=g)
        ya, yb = a(x), b(x)
        assert ya.shape == yb.shape == (2, 64, 2 * hw[0], 2 * hw[1])
        out[f"upsample_{hw[0]}x{hw[1]}"] = _rel(ya, yb)
    return out


@torch.no_grad()
def check_decoder_trunk(g):
    """the whole decoder as vae.py:198-371 composes it: conv_in, mid block (with attention: the cond_decoder's; Janus always has it),
    up levels of layers_per_block + 1 resnets, where the upsamplers sit, tail"""
    MJ = _janus()
    out = {}
    for chans, lpb in (((64, 128, 256), 2), ((64, 64), 1)):
        a = OT.DecoderRef(64, 3, chans, lpb, 32, True).eval()
        _randomise(a, g)
        mult = tuple(c // chans[0] for c in chans)
        b = MJ.JanusVQVAEDecoder(_cfg(base_channels=chans[0], channel_multiplier=mult, num_res_blocks=lpb, out_channels=3)).eval()
        _randomise(b, g)                                             # (so that nothing left un-copied could agree by accident)
        b.conv_in.load_state_dict(a.conv_in.state_dict())
        _copy_resnet(b.mid.block_1, a.mid_block.resnets[0])
        _copy_attn(b.mid.attn_1, a.mid_block.attentions[0])
        _copy_resnet(b.mid.block_2, a.mid_block.resnets[1])
        assert len(b.up) == len(a.up_blocks)
        for lvl, blk in enumerate(a.up_blocks):
            assert len(b.up[lvl].block) == len(blk.resnets) == lpb + 1
            for j, r in enumerate(blk.resnets):
                _copy_resnet(b.up[lvl].block[j], r)
            for att in b.up[lvl].attn:                                # Janus-only blocks -> identity
                att.proj_out.weight.data.zero_()
                att.proj_out.bias.data.zero_()
            if blk.upsamplers is not None:
                b.up[lvl].upsample.conv.load_state_dict(blk.upsamplers[0].conv.state_dict())
            else:
                assert not hasattr(b.up[lvl], "upsample"), "both put no upsampler on the last level"
        b.norm_out.load_state_dict(a.conv_norm_out.state_dict())
        b.conv_out.load_state_dict(a.conv_out.state_dict())
        z = torch.randn(2, 64, 4, 4, generator=g)
        ya, _ = a(z)
        yb = b(z)
        side = 4 * 2 ** (len(chans) - 1)
        assert ya.shape == yb.shape == (2, 3, side, side)
        out[f"decoder_trunk_{len(chans)}_levels"] = _rel(ya, yb)
    return out


def run(verbose=True):
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(1234)
    res = {}
    for fn in (check_resnet, check_downsample, check_attention, check_vq, check_encoder_trunk, check_upsample, check_decoder_trunk):
        res.update(fn(g))
    if verbose:
        for k, v in res.items():
            print(f"{k:44s} {v:.3e}" if isinstance(v, float) else f"{k:44s} {v}")
    bad = {k: v for k, v in res.items() if (k.endswith("ids_differ") and v != 0) or (not k.endswith("ids_differ") and v > 1e-5)}
    return res, bad


if __name__ == "__main__":
    res, bad = run()
    print("CROSS-CHECK", "FAILED: " + str(bad) if bad else "ok: oracle/df_blocks.py agrees with transformers' Chameleon (encoder side) and Janus (decoder side) VQGAN blocks")
    sys.exit(1 if bad else 0)
