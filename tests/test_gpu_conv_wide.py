"""The persistent two-tile 3x3 convolution (csrc/conv3x3w.hip) through the C ABI: against torch conv2d in fp64 at small sizes with a
small forced grid (several items per workgroup: the item switch, the prefetch of the next item's first chunk under the last chunk of
the current one, an ODD number of spatial tiles -- the last pair's second tile repeats the first and stores nothing), and bit for bit
against the 256-pixel kernel of conv3x3.hip at the decoder's real shapes with the default one-workgroup-per-CU grid (same bf16
products, same fp32 summation order: chunk-major, nine taps per chunk).  Reference semantics: DF ResnetBlock2D.conv1 / conv2 and
Upsample2D.conv as parameterised by /root/reference/ivideogpt/vq_model/vae.py:250-284."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from test_gpu_ops import DEV, P, code, lib, q, rel_err, stream, tdt, TOL

pytestmark = pytest.mark.gpu


def _wide_launches(l):
    return l.ivg_debug_counter(b"conv3x3_wide")


def _args(L, X, Wp, Y, R, bias, Nb, H, Cin, Cout, ups, flags):
    Ho = 2 * H if ups else H
    a = L.IvgIgemmArgs()
    a.X, a.W, a.Y, a.R, a.bias = X.data_ptr(), Wp.data_ptr(), Y.data_ptr(), (R.data_ptr() if R is not None else None), \
        (bias.data_ptr() if bias is not None else None)
    for k, v in dict(Nimg=Nb, Hin=H, Win=H, Cin=Cin, ldx=Cin, Hout=Ho, Wout=Ho, KH=3, KW=3, stride=1, pad=1, ups=ups, N=Cout, ldw=9 * Cin,
                     c_img=Ho * Ho * Cout, c_pix=Cout, c_ch=1, c_grp=1, c_grp_stride=0, flags=flags, alpha=1.0, nb0=1, nb1=1, nb2=1).items():
        setattr(a, k, v)
    return a


# (H, Cin, Cout, ups, Nb, grid): spatial tiles = Nb * Ho * Ho / 256, pairs = ceil(tiles / 2), items = pairs * Cout / 128 >= grid
SMALL = [
    (64, 128, 128, 0, 3, 8),     # 8 x 32 tiles, 24 pairs on 8 workgroups: three items each
    (16, 128, 256, 0, 19, 16),   # 16 x 16 tiles, 19 tiles (odd): 10 pairs x 2 N tiles on 16 workgroups, two of the 8 pair lanes run twice
    (16, 64, 128, 1, 6, 8),      # upsampling, 8 x 32 output tiles, two chunks: every chunk is an item's first or last
    (8, 64, 128, 1, 21, 8),      # upsampling, 16 x 16 output tiles, odd tile count
    (32, 256, 512, 0, 5, 32),    # four N tiles, 10 pairs on 8 pair lanes
    (16, 512, 128, 0, 40, 8),    # sixteen chunks
]


@pytest.mark.parametrize("epi", ["bias", "bias_res", "res_silu"])
@pytest.mark.parametrize("H,Cin,Cout,ups,Nb,grid", SMALL)
def test_wide_conv3x3_against_fp64(H, Cin, Cout, ups, Nb, grid, epi, switches):
    L, l = lib()
    switches(IVG_CONV_WIDE="2", IVG_CONV_WIDE_GRID=str(grid))
    dt = "bf16"
    g = torch.Generator().manual_seed(H + Cin + Cout + Nb)
    x = q(torch.randn(Nb, Cin, H, H, generator=g), dt)
    w = q(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5, dt)
    b = torch.randn(Cout, generator=g)
    Ho = 2 * H if ups else H
    res, silu, bias = "res" in epi, "silu" in epi, "bias" in epi
    xin = F.interpolate(x.double(), scale_factor=2.0, mode="nearest") if ups else x.double()
    ref = F.conv2d(xin, w.double(), b.double() if bias else None, padding=1)
    r = q(torch.randn(Nb, Cout, Ho, Ho, generator=g), dt) if res else None
    if res:
        ref = ref + r.double()
    if silu:
        ref = F.silu(ref)
    X = x.permute(0, 2, 3, 1).contiguous().to(DEV, tdt(dt))
    Wp = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(DEV, tdt(dt))
    Y = torch.full((Nb, Ho, Ho, Cout), float("nan"), device=DEV, dtype=tdt(dt))
    if res:
        Y.copy_(r.permute(0, 2, 3, 1))   # in-place residual: R == Y
    bd = b.to(DEV)
    a = _args(L, X, Wp, Y, Y if res else None, bd if bias else None, Nb, H, Cin, Cout, ups, (1 if bias else 0) | (4 if res else 0) | (8 if silu else 0))
    n0 = _wide_launches(l)
    assert l.ivg_op_igemm(C.byref(a), code(dt), stream()) == 0
    torch.cuda.synchronize()
    assert _wide_launches(l) == n0 + 1, "the persistent kernel must be the one that ran"
    assert torch.isfinite(Y.float()).all()
    assert rel_err(Y.float().permute(0, 3, 1, 2), ref) < TOL[dt]
    # and the same bits as the 256-pixel kernel
    Y2 = torch.full_like(Y, float("nan"))
    if res:
        Y2.copy_(r.permute(0, 2, 3, 1))
    switches(IVG_CONV_WIDE="0")
    a2 = _args(L, X, Wp, Y2, Y2 if res else None, bd if bias else None, Nb, H, Cin, Cout, ups, a.flags)
    assert l.ivg_op_igemm(C.byref(a2), code(dt), stream()) == 0
    torch.cuda.synchronize()
    assert _wide_launches(l) == n0 + 1
    assert torch.equal(Y, Y2)


@pytest.mark.parametrize("H,Cin,Cout,Nb,grid,res", [(64, 128, 128, 3, 8, 1), (16, 128, 256, 19, 16, 0), (32, 256, 512, 5, 32, 1), (16, 512, 128, 40, 8, 0)])
def test_wide_conv3x3_with_fused_input_groupnorm(H, Cin, Cout, Nb, grid, res, switches):
    """conv3x3(silu(GroupNorm(x))) with the normalisation applied in place on the staged halo chunks of BOTH tiles (coefficient rows
    per tile: the two tiles of a pair may belong to two images), zero padding staying zero after it."""
    L, l = lib()
    switches(IVG_CONV_WIDE="2", IVG_CONV_WIDE_GRID=str(grid))
    dt, groups = "bf16", 32
    g = torch.Generator().manual_seed(H * 3 + Cin + Cout)
    x = q(torch.randn(Nb, Cin, H, H, generator=g) * (1 + 0.5 * torch.arange(Nb).view(-1, 1, 1, 1) / Nb) + 0.3, dt)   # statistics differ per image
    w = q(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5, dt)
    b = torch.randn(Cout, generator=g)
    gamma, beta = 1 + 0.2 * torch.randn(Cin, generator=g), 0.2 * torch.randn(Cin, generator=g)
    hn = F.silu(F.group_norm(x.double(), groups, gamma.double(), beta.double(), eps=1e-6))
    hn = q(hn.float(), dt).double()
    ref = F.conv2d(hn, w.double(), b.double(), padding=1)
    r = q(torch.randn(Nb, Cout, H, H, generator=g), dt) if res else None
    if res:
        ref = ref + r.double()
    X = x.permute(0, 2, 3, 1).contiguous().to(DEV, tdt(dt))
    Wp = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(DEV, tdt(dt))
    bd, gd, btd = b.to(DEV), gamma.to(DEV), beta.to(DEV)
    ws = torch.empty(Nb * (((H * H + 1023) // 1024) * groups * 16 + Cin * 8) + 256, dtype=torch.uint8, device=DEV)
    outs = []
    for wide in ("2", "0"):
        switches(IVG_CONV_WIDE=wide)
        Y = torch.full((Nb, H, H, Cout), float("nan"), device=DEV, dtype=tdt(dt))
        if res:
            Y.copy_(r.permute(0, 2, 3, 1))
        a = _args(L, X, Wp, Y, Y if res else None, bd, Nb, H, Cin, Cout, 0, 1 | (4 if res else 0))
        n0 = _wide_launches(l)
        assert l.ivg_op_gn_conv(C.byref(a), code(dt), groups, P(gd), P(btd), 1e-6, P(ws), stream()) == 0
        torch.cuda.synchronize()
        assert _wide_launches(l) == n0 + (1 if wide == "2" else 0)
        outs.append(Y)
    assert torch.isfinite(outs[0].float()).all()
    assert rel_err(outs[0].float().permute(0, 3, 1, 2), ref) < TOL[dt]
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("H,Cin,Cout,ups,Nb,grid,res", [(64, 128, 128, 0, 3, 8, 1), (16, 128, 256, 0, 19, 16, 0), (16, 64, 128, 1, 6, 8, 0),
                                                        (16, 128, 768, 0, 16, 48, 1)])
def test_wide_conv3x3_epilogue_groupnorm_statistics(H, Cin, Cout, ups, Nb, grid, res, switches):
    """GroupNorm statistics of the output from the persistent kernel's epilogue: same chunk layout ([image][spatial tile x N tile]
    [group]) and the same partial sums as the 256-pixel kernel writes, incl. groups that straddle two N tiles (768 channels)."""
    L, l = lib()
    dt, groups = "bf16", 32
    g = torch.Generator().manual_seed(H + Cin + Cout + 1)
    x = q(torch.randn(Nb, Cin, H, H, generator=g), dt)
    w = q(torch.randn(Cout, Cin, 3, 3, generator=g) / (Cin * 9) ** 0.5, dt)
    b = torch.randn(Cout, generator=g)
    gamma, beta = 1 + 0.2 * torch.randn(Cout, generator=g), 0.2 * torch.randn(Cout, generator=g)
    Ho = 2 * H if ups else H
    r = q(torch.randn(Nb, Cout, Ho, Ho, generator=g), dt) if res else None
    X = x.permute(0, 2, 3, 1).contiguous().to(DEV, tdt(dt))
    Wp = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(DEV, tdt(dt))
    bd, gd, btd = b.to(DEV), gamma.to(DEV), beta.to(DEV)
    bound = ((Ho * Ho + 255) // 256) * ((Cout + 63) // 64)
    got = []
    for wide in ("2", "0"):
        switches(IVG_CONV_WIDE=wide, IVG_CONV_WIDE_GRID=str(grid))
        Y = torch.full((Nb, Ho, Ho, Cout), float("nan"), device=DEV, dtype=tdt(dt))
        if res:
            Y.copy_(r.permute(0, 2, 3, 1))
        a = _args(L, X, Wp, Y, Y if res else None, bd, Nb, H, Cin, Cout, ups, 1 | (4 if res else 0))
        part = torch.full((Nb * bound * groups * 2,), float("nan"), dtype=torch.float64, device=DEV)
        out = torch.full((Nb, Ho, Ho, Cout), float("nan"), device=DEV, dtype=tdt(dt))
        n0 = _wide_launches(l)
        chunks = l.ivg_op_conv_gn(C.byref(a), code(dt), P(part), groups, P(gd), P(btd), P(out), 1e-6, 1, stream())
        torch.cuda.synchronize()
        assert _wide_launches(l) == n0 + (1 if wide == "2" else 0)
        assert 0 < chunks <= bound, chunks
        got.append((Y, out, part[:Nb * chunks * groups * 2].clone()))
    (Y, out, part), (Y0, out0, part0) = got
    stored = Y.float().permute(0, 3, 1, 2).cpu().double()
    ref = F.silu(F.group_norm(stored, groups, gamma.double(), beta.double(), eps=1e-6))
    assert torch.isfinite(out.float()).all()
    assert rel_err(out.float().permute(0, 3, 1, 2), ref) < TOL[dt]
    assert torch.equal(Y, Y0) and torch.equal(part, part0) and torch.equal(out, out0)


# the decoder's shapes at config 2 (64 x 64, 114 M tokenizer): (H, Cin, Cout, ups, fused input norm), 128 frames = the context decoder's batch
FULL = [(64, 128, 128, 0, 1), (64, 256, 128, 0, 1), (32, 256, 256, 0, 1), (32, 512, 256, 0, 1), (16, 512, 512, 0, 1), (32, 256, 256, 1, 0),
        (16, 512, 512, 1, 0), (16, 512, 512, 0, 0)]


@pytest.mark.parametrize("H,Cin,Cout,ups,gn", FULL)
def test_wide_conv3x3_equals_the_256_pixel_kernel_at_model_shapes(H, Cin, Cout, ups, gn, switches):
    """default grid (one workgroup per CU), 128 frames: output, residual and statistics partials bit for bit"""
    L, l = lib()
    dt, groups, Nb = "bf16", 32, 128
    g = torch.Generator(device=DEV).manual_seed(H + Cin + Cout + ups)
    Ho = 2 * H if ups else H
    X = torch.randn(Nb, H, H, Cin, device=DEV, generator=g).to(tdt(dt))
    Wp = (torch.randn(Cout, 9 * Cin, device=DEV, generator=g) / (Cin * 9) ** 0.5).to(tdt(dt))
    bd = torch.randn(Cout, device=DEV, generator=g)
    R0 = torch.randn(Nb, Ho, Ho, Cout, device=DEV, generator=g).to(tdt(dt))
    gd, btd = 1 + 0.2 * torch.randn(Cin, device=DEV, generator=g), 0.2 * torch.randn(Cin, device=DEV, generator=g)
    ws = torch.empty(Nb * (((H * H + 1023) // 1024) * groups * 16 + Cin * 8) + 256, dtype=torch.uint8, device=DEV)
    bound = ((Ho * Ho + 255) // 256) * ((Cout + 63) // 64)
    got = []
    for wide in ("2", "0"):
        switches(IVG_CONV_WIDE=wide, IVG_CONV_WIDE_GRID=None)
        Y = R0.clone()
        a = _args(L, X, Wp, Y, Y, bd, Nb, H, Cin, Cout, ups, 1 | 4)
        n0 = _wide_launches(l)
        if gn:
            assert l.ivg_op_gn_conv(C.byref(a), code(dt), groups, P(gd), P(btd), 1e-6, P(ws), stream()) == 0
            part = None
        else:
            part = torch.full((Nb * bound * groups * 2,), float("nan"), dtype=torch.float64, device=DEV)
            out = torch.empty_like(Y)
            go, bo = torch.ones(Cout, device=DEV), torch.zeros(Cout, device=DEV)
            assert l.ivg_op_conv_gn(C.byref(a), code(dt), P(part), groups, P(go), P(bo), P(out), 1e-6, 0, stream()) > 0
        torch.cuda.synchronize()
        assert _wide_launches(l) == n0 + (1 if wide == "2" else 0)
        got.append((Y, part))
    assert torch.isfinite(got[0][0].float()).all()
    assert torch.equal(got[0][0], got[1][0])
    if not gn:
        n = Nb * (Ho * Ho // 256) * (Cout // 128) * groups * 2
        assert torch.equal(got[0][1][:n], got[1][1][:n])
