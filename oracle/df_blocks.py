"""Restatement of the ``diffusers==0.27.0`` blocks the reference tokenizer is assembled from.

TEST INFRASTRUCTURE (see oracle/__init__.py).  ``diffusers`` is a third-party dependency pinned
at ``requirements.txt:6`` of the reference and is absent from /root/reference and from the build
image, so its published algorithm is restated here.  The reference
call sites that fix the parameterisation are

  * ``ivideogpt/vq_model/vae.py:104-116``  get_down_block(..., resnet_eps=1e-6, downsample_padding=0)
  * ``ivideogpt/vq_model/vae.py:120-130,250-260``  UNetMidBlock2D(..., attention_head_dim=C, add_attention=...)
  * ``ivideogpt/vq_model/vae.py:271-284``  get_up_block(..., num_layers=layers_per_block+1)
  * ``ivideogpt/vq_model/compressive_vq_model.py:102-123``  VectorQuantizer(n_e, dim, beta=1.0, legacy=False)

Parameter/attribute names follow the DF state-dict schema so a reference
checkpoint loads unchanged.  **Parity unpinned** for this file: no DF source or golden vector is
available offline.  What narrows that: parameter counts (114.16 M / 310.47 M = the reference README's), and -- round 5 --
``oracle/pin/crosscheck_vqgan_blocks.py`` (in the CPU suite): ResnetBlock2D, Downsample2D, Attention, VectorQuantizer and the whole
encoder trunk agree (<= 1.5e-6 relative, VQ ids identical) with an INDEPENDENT implementation of the same taming-VQGAN blocks that
the image does hold, ``transformers.models.chameleon.modeling_chameleon``; Upsample2D and the whole decoder trunk (mid block with
attention, layers_per_block + 1 resnets per up level, where the upsamplers sit, tail) agree (<= 2.7e-6) with a second one,
``transformers.models.janus.modeling_janus``.  ``oracle/pin/pin_df_blocks.py`` pins this file against
real ``diffusers`` (and oracle/metrics.py against real ``piqa``) wherever those wheels import; here it prints SKIPPED.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class ResnetBlock2D(nn.Module):
    """GN -> SiLU -> conv3x3 -> GN -> SiLU -> conv3x3, plus (1x1-projected) skip.  temb is None."""

    def __init__(self, in_channels, out_channels, eps=1e-6, groups=32, output_scale_factor=1.0):
        super().__init__()
        self.norm1 = nn.GroupNorm(groups, in_channels, eps=eps, affine=True)
        self.conv1 = nn.Conv2d(in_channels, out_channels, 3, stride=1, padding=1)
        self.norm2 = nn.GroupNorm(groups, out_channels, eps=eps, affine=True)
        self.conv2 = nn.Conv2d(out_channels, out_channels, 3, stride=1, padding=1)
        self.conv_shortcut = nn.Conv2d(in_channels, out_channels, 1) if in_channels != out_channels else None
        self.output_scale_factor = output_scale_factor

    def forward(self, x, temb=None):
        h = self.conv1(F.silu(self.norm1(x)))
        h = self.conv2(F.silu(self.norm2(h)))  # dropout p=0
        if self.conv_shortcut is not None:
            x = self.conv_shortcut(x)
        return (x + h) / self.output_scale_factor


class Downsample2D(nn.Module):
    """padding=0 variant: zero-pad right/bottom by one, then conv3x3 stride 2."""

    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, stride=2, padding=0)

    def forward(self, x):
        return self.conv(F.pad(x, (0, 1, 0, 1), mode="constant", value=0))


class Upsample2D(nn.Module):
    """nearest x2 (computed in fp32 when the input is bf16), then conv3x3 pad 1."""

    def __init__(self, channels):
        super().__init__()
        self.conv = nn.Conv2d(channels, channels, 3, padding=1)

    def forward(self, x):
        dt = x.dtype
        if dt == torch.bfloat16:
            x = x.float()
        x = F.interpolate(x, scale_factor=2.0, mode="nearest")
        return self.conv(x.to(dt))


class DownEncoderBlock2D(nn.Module):
    def __init__(self, num_layers, in_channels, out_channels, add_downsample, eps, groups):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(in_channels if i == 0 else out_channels, out_channels, eps, groups) for i in range(num_layers)]
        )
        self.downsamplers = nn.ModuleList([Downsample2D(out_channels)]) if add_downsample else None

    def forward(self, x):
        for r in self.resnets:
            x = r(x)
        if self.downsamplers is not None:
            x = self.downsamplers[0](x)
        return x


class UpDecoderBlock2D(nn.Module):
    def __init__(self, num_layers, in_channels, out_channels, add_upsample, eps, groups):
        super().__init__()
        self.resnets = nn.ModuleList(
            [ResnetBlock2D(in_channels if i == 0 else out_channels, out_channels, eps, groups) for i in range(num_layers)]
        )
        self.upsamplers = nn.ModuleList([Upsample2D(out_channels)]) if add_upsample else None

    def forward(self, x, temb=None):
        for r in self.resnets:
            x = r(x)
        if self.upsamplers is not None:
            x = self.upsamplers[0](x)
        return x


class Attention(nn.Module):
    """DF ``Attention`` as built by UNetMidBlock2D: heads = C / attention_head_dim (= 1 here),
    GroupNorm on the (B, C, HW) view, biased q/k/v/out Linear, softmax(QK^T / sqrt(d)) V, residual."""

    def __init__(self, channels, dim_head, eps=1e-6, groups=32, rescale_output_factor=1.0):
        super().__init__()
        self.heads = channels // dim_head
        self.group_norm = nn.GroupNorm(groups, channels, eps=eps, affine=True)
        self.to_q = nn.Linear(channels, channels, bias=True)
        self.to_k = nn.Linear(channels, channels, bias=True)
        self.to_v = nn.Linear(channels, channels, bias=True)
        self.to_out = nn.ModuleList([nn.Linear(channels, channels, bias=True), nn.Dropout(0.0)])
        self.rescale_output_factor = rescale_output_factor

    def forward(self, x, temb=None):
        b, c, hh, ww = x.shape
        res = x
        t = x.view(b, c, hh * ww)
        t = self.group_norm(t).transpose(1, 2)  # [B, HW, C]
        q, k, v = self.to_q(t), self.to_k(t), self.to_v(t)
        d = c // self.heads
        q, k, v = (u.view(b, -1, self.heads, d).transpose(1, 2) for u in (q, k, v))
        o = F.scaled_dot_product_attention(q, k, v)  # scale 1/sqrt(d)
        o = o.transpose(1, 2).reshape(b, -1, c)
        o = self.to_out[0](o)
        o = o.transpose(-1, -2).reshape(b, c, hh, ww)
        return (o + res) / self.rescale_output_factor


class UNetMidBlock2D(nn.Module):
    def __init__(self, in_channels, resnet_eps=1e-6, resnet_act_fn="silu", output_scale_factor=1.0,
                 resnet_time_scale_shift="default", attention_head_dim=1, resnet_groups=32,
                 temb_channels=None, add_attention=True):
        super().__init__()
        assert temb_channels is None and resnet_act_fn in ("silu", "swish")
        self.resnets = nn.ModuleList([
            ResnetBlock2D(in_channels, in_channels, resnet_eps, resnet_groups, output_scale_factor),
            ResnetBlock2D(in_channels, in_channels, resnet_eps, resnet_groups, output_scale_factor),
        ])
        self.attentions = nn.ModuleList([
            Attention(in_channels, attention_head_dim, resnet_eps, resnet_groups, output_scale_factor)
            if add_attention else None
        ])

    def forward(self, x, temb=None):
        x = self.resnets[0](x)
        if self.attentions[0] is not None:
            x = self.attentions[0](x)
        return self.resnets[1](x)


def get_down_block(down_block_type, num_layers, in_channels, out_channels, add_downsample, resnet_eps,
                   downsample_padding, resnet_act_fn, resnet_groups, attention_head_dim=None, temb_channels=None,
                   **kw):
    assert down_block_type == "DownEncoderBlock2D" and downsample_padding == 0 and temb_channels is None
    return DownEncoderBlock2D(num_layers, in_channels, out_channels, add_downsample, resnet_eps, resnet_groups)


def get_up_block(up_block_type, num_layers, in_channels, out_channels, prev_output_channel, add_upsample,
                 resnet_eps, resnet_act_fn, resnet_groups, attention_head_dim=None, temb_channels=None,
                 resnet_time_scale_shift="default", **kw):
    assert up_block_type == "UpDecoderBlock2D" and temb_channels is None
    return UpDecoderBlock2D(num_layers, in_channels, out_channels, add_upsample, resnet_eps, resnet_groups)


class VectorQuantizer(nn.Module):
    """argmin_j || z - e_j ||  via ``torch.cdist`` (fp32 GEMM form), lowest index on ties."""

    def __init__(self, n_e, vq_embed_dim, beta=1.0, remap=None, sane_index_shape=False, legacy=False, **kw):
        super().__init__()
        assert remap is None and not sane_index_shape
        self.n_e, self.vq_embed_dim, self.beta, self.legacy = n_e, vq_embed_dim, beta, legacy
        self.embedding = nn.Embedding(n_e, vq_embed_dim)
        self.embedding.weight.data.uniform_(-1.0 / n_e, 1.0 / n_e)

    def forward(self, z):
        z = z.permute(0, 2, 3, 1).contiguous()
        zf = z.view(-1, self.vq_embed_dim)
        idx = torch.argmin(torch.cdist(zf, self.embedding.weight), dim=1)
        z_q = self.embedding(idx).view(z.shape)
        if not self.legacy:
            loss = self.beta * torch.mean((z_q.detach() - z) ** 2) + torch.mean((z_q - z.detach()) ** 2)
        else:
            loss = torch.mean((z_q.detach() - z) ** 2) + self.beta * torch.mean((z_q - z.detach()) ** 2)
        z_q = z + (z_q - z).detach()
        return z_q.permute(0, 3, 1, 2).contiguous(), loss, (None, None, idx)
