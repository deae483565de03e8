#!/usr/bin/env python
"""Pin ``oracle/df_blocks.py`` against the REAL ``diffusers`` and ``oracle/metrics.py`` against the REAL ``piqa`` -- wherever those
wheels are importable.  TEST INFRASTRUCTURE.   python oracle/pin/pin_df_blocks.py

The build image holds neither (no network), so here every section prints ``SKIPPED`` and the two files stay "parity unpinned"; anyone
with the reference's environment (``pip install diffusers==0.27.0 piqa``, reference ``requirements.txt:6``) closes SURVEY.md 8(c) in one
command.  What it does when they import:

  * builds the blocks through the SAME factory calls the reference makes -- ``get_down_block`` (``vae.py:104-116``), ``UNetMidBlock2D``
    (``vae.py:120-130,250-260``), ``get_up_block`` (``vae.py:271-284``), ``VectorQuantizer`` (``compressive_vq_model.py:102-123``) --
    loads the oracle block's ``state_dict()`` into them with ``strict=True`` (pins the key schema) and asserts equal outputs
    (<= 1e-6 relative; VQ ids identical),
  * runs ``piqa.SSIM(window_size=11, sigma=1.5, n_channels=3, reduction='none')`` / ``piqa.PSNR(epsilon=1e-8, value_range=1.0,
    reduction='none')`` (``ivideogpt/utils/video_metric.py:23-24``) against ``oracle.metrics``.

Exit status: 0 = every section that could run passed (or was skipped), 1 = a mismatch.
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import df_blocks as DF   # noqa: E402
from oracle import metrics as OM     # noqa: E402


def _rel(a, b):
    return ((a - b).abs().max() / b.abs().max().clamp_min(1e-12)).item()


def _randomise(m, g):
    for p in m.parameters():
        p.data.copy_(torch.randn(p.shape, generator=g) * (1.0 / float(p[0].numel()) ** 0.5 if p.dim() > 1 else 0.3))
    for mod in m.modules():
        if isinstance(mod, torch.nn.GroupNorm):
            mod.weight.data.copy_(1.0 + 0.3 * torch.randn(mod.weight.shape, generator=g))


@torch.no_grad()
def pin_diffusers():
    try:
        import diffusers
        from diffusers.models.unets.unet_2d_blocks import UNetMidBlock2D, get_down_block, get_up_block
        from diffusers.models.autoencoders.vae import VectorQuantizer
    except Exception as e:   # noqa: BLE001 -- any import failure means "not available here"
        print(f"diffusers: SKIPPED (not importable: {type(e).__name__}: {e})")
        return None
    print(f"diffusers {diffusers.__version__} (the reference pins 0.27.0)")
    g = torch.Generator().manual_seed(99)
    res = {}
    # DownEncoderBlock2D incl. Downsample2D -- vae.py:104-116
    for cin, cout, down in ((64, 128, True), (128, 128, False)):
        mine = DF.get_down_block("DownEncoderBlock2D", num_layers=2, in_channels=cin, out_channels=cout, add_downsample=down, resnet_eps=1e-6,
                                 downsample_padding=0, resnet_act_fn="silu", resnet_groups=32, attention_head_dim=cout, temb_channels=None).eval()
        _randomise(mine, g)
        ref = get_down_block("DownEncoderBlock2D", num_layers=2, in_channels=cin, out_channels=cout, add_downsample=down, resnet_eps=1e-6,
                             downsample_padding=0, resnet_act_fn="silu", resnet_groups=32, attention_head_dim=cout, temb_channels=None).eval()
        ref.load_state_dict(mine.state_dict(), strict=True)
        x = torch.randn(2, cin, 11, 16, generator=g)
        res[f"down_block_{cin}_{cout}_down={int(down)}"] = _rel(mine(x), ref(x))
    # UNetMidBlock2D with / without attention -- vae.py:120-130,250-260
    for c, att in ((128, False), (128, True), (64, True)):
        mine = DF.UNetMidBlock2D(in_channels=c, resnet_eps=1e-6, resnet_act_fn="silu", output_scale_factor=1, resnet_time_scale_shift="default",
                                 attention_head_dim=c, resnet_groups=32, temb_channels=None, add_attention=att).eval()
        _randomise(mine, g)
        ref = UNetMidBlock2D(in_channels=c, resnet_eps=1e-6, resnet_act_fn="silu", output_scale_factor=1, resnet_time_scale_shift="default",
                             attention_head_dim=c, resnet_groups=32, temb_channels=None, add_attention=att).eval()
        sd = {k: v for k, v in mine.state_dict().items()}
        ref.load_state_dict(sd, strict=True)
        x = torch.randn(2, c, 8, 8, generator=g)
        res[f"mid_block_{c}_attention={int(att)}"] = _rel(mine(x), ref(x))
    # UpDecoderBlock2D incl. Upsample2D -- vae.py:271-284 (resnet_time_scale_shift = norm_type = "group")
    for cin, cout, up in ((128, 64, True), (64, 64, False)):
        mine = DF.get_up_block("UpDecoderBlock2D", num_layers=3, in_channels=cin, out_channels=cout, prev_output_channel=None, add_upsample=up,
                               resnet_eps=1e-6, resnet_act_fn="silu", resnet_groups=32, attention_head_dim=cout, temb_channels=None,
                               resnet_time_scale_shift="group").eval()
        _randomise(mine, g)
        ref = get_up_block("UpDecoderBlock2D", num_layers=3, in_channels=cin, out_channels=cout, prev_output_channel=None, add_upsample=up,
                           resnet_eps=1e-6, resnet_act_fn="silu", resnet_groups=32, attention_head_dim=cout, temb_channels=None,
                           resnet_time_scale_shift="group").eval()
        ref.load_state_dict(mine.state_dict(), strict=True)
        x = torch.randn(2, cin, 8, 8, generator=g)
        res[f"up_block_{cin}_{cout}_up={int(up)}"] = _rel(mine(x), ref(x))
        xb = x.to(torch.bfloat16)   # Upsample2D's bf16 -> fp32 -> bf16 detour around F.interpolate
        res[f"up_block_{cin}_{cout}_up={int(up)}_bf16"] = _rel(mine.to(torch.bfloat16)(xb).float(), ref.to(torch.bfloat16)(xb).float())
    # VectorQuantizer -- compressive_vq_model.py:102-123
    for n_e in (512, 8192):
        mine = DF.VectorQuantizer(n_e, 64, beta=1.0, remap=None, sane_index_shape=False, legacy=False)
        ref = VectorQuantizer(n_e, 64, beta=1.0, remap=None, sane_index_shape=False, legacy=False)
        mine.embedding.weight.data.copy_(torch.randn(n_e, 64, generator=g) * 0.05)
        ref.load_state_dict(mine.state_dict(), strict=True)
        z = torch.randn(3, 64, 16, 16, generator=g) * 0.05
        zq_a, loss_a, (_, _, idx_a) = mine(z)
        zq_b, loss_b, (_, _, idx_b) = ref(z)
        res[f"vq_{n_e}_ids_differ"] = int((idx_a.flatten() != idx_b.flatten()).sum())
        res[f"vq_{n_e}_zq"] = _rel(zq_a, zq_b)
        res[f"vq_{n_e}_loss"] = abs(float(loss_a) - float(loss_b)) / max(abs(float(loss_b)), 1e-12)
    return res


@torch.no_grad()
def pin_piqa():
    try:
        import piqa
    except Exception as e:   # noqa: BLE001
        print(f"piqa: SKIPPED (not importable: {type(e).__name__}: {e})")
        return None
    print(f"piqa {getattr(piqa, '__version__', '?')}")
    g = torch.Generator().manual_seed(7)
    ssim = piqa.SSIM(window_size=11, sigma=1.5, n_channels=3, reduction="none")   # video_metric.py:24
    psnr = piqa.PSNR(epsilon=1e-08, value_range=1.0, reduction="none")            # video_metric.py:23
    res = {}
    for (n, h, w) in ((6, 64, 64), (2, 256, 256), (3, 80, 107)):
        a = torch.rand(n, 3, h, w, generator=g)
        b = (a + 0.1 * torch.randn(n, 3, h, w, generator=g)).clamp(0, 1)
        res[f"ssim_{h}x{w}"] = _rel(OM.ssim_per_image(a, b), ssim(a, b))
        mse = ((a - b) ** 2).mean([1, 2, 3])
        res[f"psnr_{h}x{w}"] = _rel(10 * torch.log10(1.0 / (mse + 1e-8)), psnr(a, b))
    return res


def main():
    bad = {}
    for name, fn in (("diffusers", pin_diffusers), ("piqa", pin_piqa)):
        res = fn()
        if res is None:
            continue
        for k, v in res.items():
            print(f"  {k:40s} {v:.3e}" if isinstance(v, float) else f"  {k:40s} {v}")
            if (k.endswith("ids_differ") and v != 0) or (not k.endswith("ids_differ") and v > (2e-2 if k.endswith("bf16") else 1e-6 if "ssim" not in k else 1e-5)):
                bad[k] = v
        print(f"{name}: {'MISMATCH ' + str({k: v for k, v in bad.items()}) if bad else 'PINNED'}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
