"""CPU restatement of the reference's compressive tokenizer (ctx_vqgan).  TEST INFRASTRUCTURE.

Follows, function by function:

  * ``Encoder.forward(return_features=True)``       ivideogpt/vq_model/vae.py:141-195
  * ``Decoder.forward(return_features=True)``       ivideogpt/vq_model/vae.py:298-371
  * ``CrossAttentionBlock.forward``                 ivideogpt/vq_model/conditional_vae.py:38-55
  * ``ConditionalEncoder.forward``                  ivideogpt/vq_model/conditional_vae.py:108-132
  * ``ConditionalDecoder.forward``                  ivideogpt/vq_model/conditional_vae.py:186-212
  * ``CompressiveVQModel.tokenize``                 ivideogpt/vq_model/compressive_vq_model.py:164-220
  * ``CompressiveVQModel.detokenize``               ivideogpt/vq_model/compressive_vq_model.py:222-277
  * ``CompressiveVQModel.set_context_length``       ivideogpt/vq_model/compressive_vq_model.py:154-158

Pinned by ``oracle/pin/pin_against_reference.py``: the reference's own (unmodified) classes run on
the same ``oracle.df_blocks`` with the same state dict give bit-identical tokens / pixels.  The DF
blocks underneath remain unpinned (oracle/df_blocks.py).

Module/parameter names equal the reference's, so ``state_dict()`` is the DF checkpoint schema
(SURVEY.md Appendix C).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .df_blocks import DownEncoderBlock2D, UpDecoderBlock2D, UNetMidBlock2D, VectorQuantizer

CTX_RES = 16   # context token grid is 16x16 at both resolutions   (compressive_vq_model.py:225)
DYN_RES = 4    # dynamics token grid is 4x4                        (compressive_vq_model.py:226)


class EncoderRef(nn.Module):
    def __init__(self, in_channels, out_channels, chans, layers_per_block, groups, mid_attention):
        super().__init__()
        self.conv_in = nn.Conv2d(in_channels, chans[0], 3, padding=1)
        self.down_blocks = nn.ModuleList()
        prev = chans[0]
        for i, c in enumerate(chans):
            self.down_blocks.append(DownEncoderBlock2D(layers_per_block, prev, c, i != len(chans) - 1, 1e-6, groups))
            prev = c
        self.mid_block = UNetMidBlock2D(chans[-1], resnet_eps=1e-6, attention_head_dim=chans[-1],
                                        resnet_groups=groups, add_attention=mid_attention)
        self.conv_norm_out = nn.GroupNorm(groups, chans[-1], eps=1e-6)
        self.conv_out = nn.Conv2d(chans[-1], out_channels, 3, padding=1)

    def tail(self, x):
        return self.conv_out(F.silu(self.conv_norm_out(x)))

    def forward(self, x):
        """-> (latent, [conv_in out, each down level out, mid out])   vae.py:148-150,181,185"""
        x = self.conv_in(x)
        feats = [x]
        for blk in self.down_blocks:
            x = blk(x)
            feats.append(x)
        x = self.mid_block(x)
        feats.append(x)
        return self.tail(x), feats


class DecoderRef(nn.Module):
    def __init__(self, in_channels, out_channels, chans, layers_per_block, groups, mid_attention):
        super().__init__()
        rev = list(reversed(chans))
        self.conv_in = nn.Conv2d(in_channels, rev[0], 3, padding=1)
        self.mid_block = UNetMidBlock2D(rev[0], resnet_eps=1e-6, attention_head_dim=rev[0],
                                        resnet_groups=groups, add_attention=mid_attention)
        self.up_blocks = nn.ModuleList()
        prev = rev[0]
        for i, c in enumerate(rev):
            self.up_blocks.append(UpDecoderBlock2D(layers_per_block + 1, prev, c, i != len(rev) - 1, 1e-6, groups))
            prev = c
        self.conv_norm_out = nn.GroupNorm(groups, chans[0], eps=1e-6)
        self.conv_out = nn.Conv2d(chans[0], out_channels, 3, padding=1)

    def tail(self, x):
        return self.conv_out(F.silu(self.conv_norm_out(x)))

    def forward(self, x):
        """-> (pixels, [conv_in out, mid out, each up level out])   vae.py:306-309,353,358"""
        x = self.conv_in(x)
        feats = [x]
        x = self.mid_block(x)
        feats.append(x)
        for blk in self.up_blocks:
            x = blk(x)
            feats.append(x)
        return self.tail(x), feats


class CrossAttentionRef(nn.Module):
    """conditional_vae.py:10-55.  GroupNorm default eps (1e-5) on q and kv, learned position
    embeddings, 4-head ``nn.MultiheadAttention`` (q != kv), ``silu(z + out)``; dropout inactive."""

    def __init__(self, channels, resolution, kv_frames):
        super().__init__()
        self.att = nn.MultiheadAttention(channels, 4, dropout=0.1, batch_first=True)
        self.kv_norm = nn.GroupNorm(32, channels)
        self.q_norm = nn.GroupNorm(32, channels)
        self.kv_frames = kv_frames
        self.kv_pos_emb = nn.Parameter(torch.zeros(kv_frames * resolution * resolution, channels))
        self.q_pos_emb = nn.Parameter(torch.zeros(resolution * resolution, channels))

    def set_kv_frames(self, k):
        # keeps the LAST k frames' rows  (conditional_vae.py:34-36)
        self.kv_pos_emb.data = self.kv_pos_emb.data[-k * self.kv_pos_emb.shape[0] // self.kv_frames:]
        self.kv_frames = k

    def forward(self, z, addin):
        if self.kv_frames > 1:  # [M,t,C,H,W] -> [M,C,t*H,W]: the norm statistics span all context frames
            addin = addin.permute(0, 2, 1, 3, 4).reshape(addin.shape[0], addin.shape[2], -1, addin.shape[3])
        kv = self.kv_norm(addin).permute(0, 2, 3, 1).reshape(addin.shape[0], -1, addin.shape[1]) + self.kv_pos_emb
        q = self.q_norm(z).permute(0, 2, 3, 1).reshape(z.shape[0], -1, z.shape[1]) + self.q_pos_emb
        out, _ = self.att(q, kv, kv)
        return F.silu(z + out.permute(0, 2, 1).reshape(z.shape))


class CondEncoderRef(EncoderRef):
    def __init__(self, in_channels, out_channels, chans, layers_per_block, groups, max_att, init_res, ctx):
        super().__init__(in_channels, out_channels, chans, layers_per_block, groups, True)
        self.max_att = max_att
        self.cross_att_blocks = nn.ModuleList()
        res = init_res
        for i, c in enumerate(chans):
            if i != len(chans) - 1:
                res //= 2
            if res <= max_att:
                self.cross_att_blocks.append(CrossAttentionRef(c, res, ctx))

    def forward(self, x, cond):
        x = self.conv_in(x)
        k = 0
        for i, blk in enumerate(self.down_blocks):
            x = blk(x)
            if x.shape[-1] <= self.max_att:
                x = self.cross_att_blocks[k](x, cond[i + 1])
                k += 1
        return self.tail(self.mid_block(x))


class CondDecoderRef(DecoderRef):
    def __init__(self, in_channels, out_channels, chans, layers_per_block, groups, max_att, init_res, ctx):
        super().__init__(in_channels, out_channels, chans, layers_per_block, groups, True)
        self.max_att = max_att
        rev = list(reversed(chans))
        res = init_res
        self.cross_att_blocks = nn.ModuleList([CrossAttentionRef(rev[0], res, ctx)])
        for i, c in enumerate(rev):
            if i != len(rev) - 1:
                res *= 2
            if res <= max_att:
                self.cross_att_blocks.append(CrossAttentionRef(c, res, ctx))

    def forward(self, x, cond):
        x = self.mid_block(self.conv_in(x))
        x = self.cross_att_blocks[0](x, cond[1])
        for i, blk in enumerate(self.up_blocks):
            x = blk(x)
            if x.shape[-1] <= self.max_att:
                x = self.cross_att_blocks[i + 1](x, cond[i + 2])
        return self.tail(x)


def _repeat_features(feats, n_traj, ctx, fut):
    """compressive_vq_model.py:176-187 / :257-266  (expand == repeat numerically)."""
    out = []
    for f in feats:
        if ctx > 1:
            g = f.reshape(n_traj, ctx, *f.shape[-3:]).unsqueeze(1).expand(-1, fut, -1, -1, -1, -1)
            out.append(g.reshape(-1, ctx, *f.shape[-3:]))
        else:
            out.append(f.unsqueeze(1).expand(-1, fut, -1, -1, -1).reshape(-1, *f.shape[-3:]))
    return out


class CompressiveVQRef(nn.Module):
    def __init__(self, in_channels=3, out_channels=3, block_out_channels=(64,), layers_per_block=1,
                 latent_channels=3, num_vq_embeddings=256, norm_num_groups=32, vq_embed_dim=None,
                 mid_block_add_attention=True, num_dyn_embeddings=256, context_length=1,
                 max_att_resolution=32, resolution=256, patch_size=4, **unused):
        super().__init__()
        chans = tuple(block_out_channels)
        self.context_length = context_length
        self.num_vq_embeddings, self.num_dyn_embeddings = num_vq_embeddings, num_dyn_embeddings
        self.patch_size, self.latent_channels = patch_size, latent_channels
        self.vq_embed_dim = vq_embed_dim or latent_channels
        self.cond_encoder = CondEncoderRef(in_channels, latent_channels, chans, layers_per_block, norm_num_groups,
                                           max_att_resolution, resolution, context_length)
        self.encoder = EncoderRef(in_channels, latent_channels, chans, layers_per_block, norm_num_groups,
                                  mid_block_add_attention)
        self.quant_conv = nn.Conv2d(latent_channels, self.vq_embed_dim, 1)
        self.quantize = VectorQuantizer(num_vq_embeddings, self.vq_embed_dim)
        self.post_quant_conv = nn.Conv2d(self.vq_embed_dim, latent_channels, 1)
        self.quant_linear = nn.Linear(latent_channels * patch_size * patch_size, self.vq_embed_dim)
        self.dynamics_quantize = VectorQuantizer(num_dyn_embeddings, self.vq_embed_dim)
        self.post_quant_linear = nn.Linear(self.vq_embed_dim, latent_channels * patch_size * patch_size)
        self.cond_decoder = CondDecoderRef(latent_channels, out_channels, chans, layers_per_block, norm_num_groups,
                                           max_att_resolution, 16, context_length)
        self.decoder = DecoderRef(latent_channels, out_channels, chans, layers_per_block, norm_num_groups,
                                  mid_block_add_attention)

    def set_context_length(self, k):
        self.context_length = k
        for m in list(self.cond_encoder.cross_att_blocks) + list(self.cond_decoder.cross_att_blocks):
            m.set_kv_frames(k)

    # ------------------------------------------------------------------ tokenize
    @torch.no_grad()
    def encode_stages(self, pixel_values, context_length):
        """All intermediates of tokenize (for per-stage parity tests)."""
        assert context_length == self.context_length
        B, T, C, H, W = pixel_values.shape
        ctx, fut = context_length, T - context_length
        context = pixel_values[:, :ctx].reshape(-1, C, H, W)
        future = pixel_values[:, ctx:].reshape(-1, C, H, W)
        h, feats = self.encoder(context)
        hq = self.quant_conv(h)
        d = self.cond_encoder(future, _repeat_features(feats, B, ctx, fut))
        p = self.patch_size
        dp = d.permute(0, 2, 3, 1).unfold(1, p, p).unfold(2, p, p).permute(0, 1, 2, 4, 5, 3)
        dp = dp.reshape(dp.shape[0], dp.shape[1] * dp.shape[2], -1)      # [M, 16, p*p*C], feature order (ph, pw, c)
        dq = self.quant_linear(dp)                                         # [M, 16, 64]
        idx_c = self.quantize(hq)[2][2]
        idx_d = self.dynamics_quantize(dq.transpose(-1, -2).unsqueeze(-1))[2][2]
        return dict(features=feats, h=h, hq=hq, d=d, dq=dq, idx_c=idx_c, idx_d=idx_d)

    @torch.no_grad()
    def tokenize(self, pixel_values, context_length=0):
        st = self.encode_stages(pixel_values, context_length)
        B, T = pixel_values.shape[:2]
        ctx, fut = context_length, T - context_length
        idx_c = st["idx_c"].reshape(B, ctx, -1)
        scf = self.num_vq_embeddings + self.num_dyn_embeddings
        idx_c = torch.cat([torch.full((B, ctx, 1), scf, dtype=idx_c.dtype), idx_c], 2).reshape(B, -1)[:, 1:]
        idx_d = st["idx_d"].reshape(B, fut, -1) + self.num_vq_embeddings
        idx_d = torch.cat([torch.full((B, fut, 1), scf + 1, dtype=idx_d.dtype), idx_d], 2).reshape(B, -1)
        indices = torch.cat([idx_c, idx_d], 1)
        labels = torch.cat([torch.full((B, idx_c.shape[1] + 1), -100, dtype=indices.dtype), idx_d[:, 1:]], 1)
        return indices, labels

    # ---------------------------------------------------------------- detokenize
    @torch.no_grad()
    def split_tokens(self, indices, context_length):
        B = indices.shape[0]
        per_c, per_d = 1 + CTX_RES * CTX_RES, 1 + DYN_RES * DYN_RES
        assert (indices.shape[1] + 1 - per_c * context_length) % per_d == 0
        fut = (indices.shape[1] + 1 - per_c * context_length) // per_d
        ids = torch.cat([torch.ones(B, 1, dtype=indices.dtype), indices], 1)
        n_c = context_length * per_c
        idx_c = ids[:, :n_c].reshape(B, context_length, -1)[:, :, 1:].reshape(B, -1)
        idx_d = ids[:, n_c:].reshape(B, fut, -1)[:, :, 1:].reshape(B, -1)
        idx_d = (idx_d - self.num_vq_embeddings).clamp(min=0, max=self.num_dyn_embeddings - 1)
        return idx_c, idx_d, fut

    @torch.no_grad()
    def decode_stages(self, indices, context_length):
        assert context_length == self.context_length
        B = indices.shape[0]
        idx_c, idx_d, fut = self.split_tokens(indices, context_length)
        quant = self.quantize.embedding(idx_c).reshape(B * context_length, CTX_RES, CTX_RES, self.vq_embed_dim)
        quant2 = self.post_quant_conv(quant.permute(0, 3, 1, 2))
        qd = self.dynamics_quantize.embedding(idx_d).reshape(-1, DYN_RES * DYN_RES, self.vq_embed_dim)
        q2d = self.post_quant_linear(qd)
        hw, p, c = quant2.shape[-1], self.patch_size, self.latent_channels
        q2d = q2d.reshape(q2d.shape[0], hw // p, hw // p, p, p, c)
        q2d = torch.einsum("nhwpqc->nchpwq", q2d).reshape(q2d.shape[0], c, hw, hw)
        ctx_dec, feats = self.decoder(quant2)
        dec = self.cond_decoder(q2d, _repeat_features(feats, B, context_length, fut))
        return dict(quant2=quant2, quant2_d=q2d, features=feats, context_dec=ctx_dec, dec=dec, fut=fut)

    @torch.no_grad()
    def detokenize(self, indices, context_length=0):
        st = self.decode_stages(indices, context_length)
        B = indices.shape[0]
        c = st["context_dec"].reshape(B, context_length, *st["context_dec"].shape[-3:])
        d = st["dec"].reshape(B, st["fut"], *st["dec"].shape[-3:])
        return torch.cat([c, d], 1)


def vq_margin(z_flat, codebook, idx):
    """fp64 audit of an index assignment: (best distance, runner-up distance, distance of idx)."""
    d = torch.cdist(z_flat.double(), codebook.double())
    top2 = torch.topk(d, 2, dim=1, largest=False).values
    return top2[:, 0], top2[:, 1], d.gather(1, idx.view(-1, 1)).squeeze(1)
