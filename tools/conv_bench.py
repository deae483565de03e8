"""Time one conv shape through the C ABI (development aid).  python tools/conv_bench.py H Cin Cout [ups] [N] [dtype] [gn]
gn = 1: conv3x3(silu(GroupNorm(x))) with the normalisation inside the staging (ivg_op_gn_conv)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ivideogpt_amd import _lib  # noqa: E402


def main():
    H, Cin, Cout = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    ups = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    N = int(sys.argv[5]) if len(sys.argv) > 5 else 896
    dt = sys.argv[6] if len(sys.argv) > 6 else "bf16"
    gn = int(sys.argv[7]) if len(sys.argv) > 7 else 0
    tdt = torch.bfloat16 if dt == "bf16" else torch.float32
    lib = _lib.load()
    dev = "cuda:0"
    Ho = 2 * H if ups else H
    x = torch.randn(N, H, H, Cin, device=dev).to(tdt)
    w = (torch.randn(Cout, 9 * Cin, device=dev) / (9 * Cin) ** 0.5).to(tdt)
    y = torch.empty(N, Ho, Ho, Cout, device=dev, dtype=tdt)
    a = _lib.IvgIgemmArgs()
    a.X, a.W, a.Y = x.data_ptr(), w.data_ptr(), y.data_ptr()
    for k, v in dict(Nimg=N, Hin=H, Win=H, Cin=Cin, ldx=Cin, Hout=Ho, Wout=Ho, KH=3, KW=3, stride=1, pad=1, ups=ups, N=Cout, ldw=9 * Cin,
                     c_img=Ho * Ho * Cout, c_pix=Cout, c_ch=1, c_grp=1, c_grp_stride=0, flags=0, alpha=1.0, nb0=1, nb1=1, nb2=1).items():
        setattr(a, k, v)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    code = 1 if dt == "bf16" else 0
    if gn:
        groups = 32
        gam, bet = torch.ones(Cin, device=dev), torch.zeros(Cin, device=dev)
        ws = torch.empty(N * (((H * H + 1023) // 1024) * groups * 16 + Cin * 8) + 256, dtype=torch.uint8, device=dev)
        ws_p, g_p, b_p = C.c_void_p(ws.data_ptr()), C.c_void_p(gam.data_ptr()), C.c_void_p(bet.data_ptr())
        # the statistics / coefficient kernels run once; the timed loop repeats the convolution with the coefficients in place
        assert lib.ivg_op_gn_conv(C.byref(a), code, groups, g_p, b_p, 1e-6, ws_p, st) == 0
        call = lambda: lib.ivg_op_gn_conv(C.byref(a), code, groups, g_p, b_p, 1e-6, ws_p, st)   # noqa: E731
    else:
        call = lambda: lib.ivg_op_igemm(C.byref(a), code, st)   # noqa: E731
    for _ in range(3):
        assert call() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    iters = 10
    e0.record()
    for _ in range(iters):
        call()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    fl = 2.0 * N * Ho * Ho * Cout * 9 * Cin
    print(f"H={H} Cin={Cin} Cout={Cout} ups={ups} N={N} {dt} gn={gn} c3={os.environ.get('IVG_CONV3X3', '1')}: "
          f"{ms:.3f} ms  {fl / ms / 1e9:.0f} TFLOP/s")


if __name__ == "__main__":
    main()
