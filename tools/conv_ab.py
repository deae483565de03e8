"""A/B of the 3x3 convolution kernels at the decoder's shapes through the C ABI (development aid): for every (H, Cin, Cout, ups, gn)
the 256-pixel kernel of conv3x3.hip (IVG_CONV_WIDE=0) beside the persistent two-tile kernel of conv3x3w.hip (IVG_CONV_WIDE=1),
or over a preset of development variants.   python tools/conv_ab.py [N=896] [res=64|256] [preset=wide|pf|probe]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ivideogpt_amd import _lib, switches  # noqa: E402

# (H, Cin, Cout, ups, gn, launches per decoder pass): 64 x 64 tokenizer (vae.py:250-284 with configs/ctx_vae64)
SHAPES64 = [(64, 128, 128, 0, 1, 5), (64, 256, 128, 0, 1, 1), (32, 256, 256, 0, 1, 5), (32, 512, 256, 0, 1, 1), (16, 512, 512, 0, 1, 10),
            (32, 256, 256, 1, 0, 1), (16, 512, 512, 1, 0, 1),
            (64, 128, 128, 0, 0, 0), (32, 256, 256, 0, 0, 0), (16, 512, 512, 0, 0, 0)]
SHAPES256 = [(256, 128, 128, 0, 1, 5), (256, 256, 128, 0, 1, 1), (128, 256, 256, 0, 1, 6), (64, 256, 256, 0, 1, 5), (64, 512, 256, 0, 1, 1),
             (32, 512, 512, 0, 1, 5), (32, 768, 512, 0, 1, 1), (16, 768, 768, 0, 1, 10),
             (128, 256, 256, 1, 0, 1), (64, 256, 256, 1, 0, 1), (32, 512, 512, 1, 0, 1), (16, 768, 768, 1, 0, 1)]


def time_conv(lib, H, Cin, Cout, ups, gn, N, iters=6):
    dev = "cuda:0"
    Ho = 2 * H if ups else H
    x = torch.randn(N, H, H, Cin, device=dev).to(torch.bfloat16)
    w = (torch.randn(Cout, 9 * Cin, device=dev) / (9 * Cin) ** 0.5).to(torch.bfloat16)
    y = torch.empty(N, Ho, Ho, Cout, device=dev, dtype=torch.bfloat16)
    bias = torch.randn(Cout, device=dev)
    a = _lib.IvgIgemmArgs()
    a.X, a.W, a.Y, a.bias = x.data_ptr(), w.data_ptr(), y.data_ptr(), bias.data_ptr()
    for k, v in dict(Nimg=N, Hin=H, Win=H, Cin=Cin, ldx=Cin, Hout=Ho, Wout=Ho, KH=3, KW=3, stride=1, pad=1, ups=ups, N=Cout, ldw=9 * Cin,
                     c_img=Ho * Ho * Cout, c_pix=Cout, c_ch=1, c_grp=1, c_grp_stride=0, flags=1, alpha=1.0, nb0=1, nb1=1, nb2=1).items():
        setattr(a, k, v)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if gn:
        groups = 32
        gam, bet = torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev)
        ws = torch.empty(N * (((H * H + 1023) // 1024) * groups * 16 + Cin * 8) + 256, dtype=torch.uint8, device=dev)
        # statistics + coefficients once; the timed launches use the convolution alone with the coefficients in place
        assert lib.ivg_op_gn_conv(C.byref(a), 1, groups, C.c_void_p(gam.data_ptr()), C.c_void_p(bet.data_ptr()), 1e-6, C.c_void_p(ws.data_ptr()), st) == 0

        def call():   # (the op entry re-runs the input's statistics + coefficient kernels: same in both columns; the kernel trace separates them)
            return lib.ivg_op_gn_conv(C.byref(a), 1, groups, C.c_void_p(gam.data_ptr()), C.c_void_p(bet.data_ptr()), 1e-6, C.c_void_p(ws.data_ptr()), st)
    else:
        def call():
            return lib.ivg_op_igemm(C.byref(a), 1, st)
    for _ in range(2):
        assert call() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    return ms, 2.0 * N * Ho * Ho * Cout * 9 * Cin / ms / 1e9


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 896
    res = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    preset = sys.argv[3] if len(sys.argv) > 3 else "wide"
    base = dict(IVG_CONV_WIDE=2, IVG_CONV_WIDE_GRID=None, IVG_CONV_WIDE_PF=None, IVG_CONV_WIDE_PROBE=None, IVG_CONV_WIDE_STAGGER=None)
    variants = {"wide": [("wide", {})],
                "pf": [("pf0", dict(IVG_CONV_WIDE_PF=0)), ("pf1", dict(IVG_CONV_WIDE_PF=1))],
                "stagger": [("st0", dict(IVG_CONV_WIDE_STAGGER=0)), ("st1", dict(IVG_CONV_WIDE_STAGGER=1)), ("st2", dict(IVG_CONV_WIDE_STAGGER=2)),
                            ("st4", dict(IVG_CONV_WIDE_STAGGER=4))],
                "policy": [("default", dict(IVG_CONV_WIDE=1))],
                # WRONG results, timing only: what the epilogue / the in-place input normalisation cost inside the persistent kernel
                "probe": [("full", {}), ("no-epilogue", dict(IVG_CONV_WIDE_PROBE=1)), ("no-norm", dict(IVG_CONV_WIDE_PROBE=2)),
                          ("neither", dict(IVG_CONV_WIDE_PROBE=3))]}[preset]
    lib = _lib.load()
    shapes = SHAPES64 if res == 64 else SHAPES256
    tot = {}
    print(f"# N={N} frames, bf16; ms per launch (TFLOP/s); gn=1 rows include the input's statistics + coefficient kernels in both columns")
    for (H, Cin, Cout, ups, gn, mult) in shapes:
        row = []
        switches.set(**dict(base, IVG_CONV_WIDE=0))
        ms0, tf0 = time_conv(lib, H, Cin, Cout, ups, gn, N)
        row.append(f"narrow {ms0:7.3f} ({tf0:5.0f})")
        tot["narrow"] = tot.get("narrow", 0.0) + ms0 * mult
        for label, env in variants:
            switches.set(**dict(base, **env))
            n0 = lib.ivg_debug_counter(b"conv3x3_wide")
            ms1, tf1 = time_conv(lib, H, Cin, Cout, ups, gn, N)
            ran = lib.ivg_debug_counter(b"conv3x3_wide") > n0
            row.append(f"{label} {ms1:7.3f} ({tf1:5.0f}){'' if ran else ' NOT-RUN'} x{ms0 / ms1:4.2f}")
            tot[label] = tot.get(label, 0.0) + ms1 * mult
        switches.set(**base)
        print(f"H={H:3d} {Cin:3d}->{Cout:3d} ups={ups} gn={gn} x{mult:2d}: " + " | ".join(row), flush=True)
    print("# weighted by launches per decoder pass (ms):", {k: round(v, 2) for k, v in tot.items()})


if __name__ == "__main__":
    main()
