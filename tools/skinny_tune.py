"""Sweep the (MF, FN, WAVES) tile shapes of the decode-step GEMM over the four per-layer shapes of a model and print the
kernel time of each (run under rocprofv3 --kernel-trace; this script only launches, tools/skinny_tune_report.py reads the trace).
Weights rotate through 24 copies so that they are not L2-resident (a decode step streams 252 MB of weights between two uses
of the same matrix).  Usage: python tools/skinny_tune.py [hidden] [intermediate] [batch]"""
import ctypes as C
import os
import sys

import torch

from ivideogpt_amd import _lib

H = int(sys.argv[1]) if len(sys.argv) > 1 else 768
I = int(sys.argv[2]) if len(sys.argv) > 2 else 3072
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
l = _lib.load()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
SK_NORM = 64
shapes = [("qkv", 3 * H, H, SK_NORM), ("oproj", H, H, 4), ("gateup", 2 * I, H, 16 | SK_NORM), ("down", H, I, 4)]
combos = [(mf, fn, w) for mf in (1, 2, 4) for fn in (1, 2, 4) for w in (4, 8, 16)]
REP, COPIES = 12, 24
plan = []
for name, N, K, flags in shapes:
    x = torch.randn(B, K, device="cuda").to(torch.bfloat16)
    ws = [(torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16) for _ in range(COPIES)]
    ldy = N // 2 if flags & 16 else N
    y = torch.zeros(B, ldy, device="cuda", dtype=torch.bfloat16)
    for mf, fn, w in [(0, 0, 0)] + combos:
        if (flags & 16) and fn == 1:
            continue
        if mf * fn > 16 or (mf + fn) > 10:
            continue
        if w == 16 and not (mf + fn <= 5 and mf * fn <= 4):
            continue
        if w and K % (w * 32) != 0:
            continue
        if mf or fn:
            os.environ["IVG_SK_FORCE"] = f"{mf},{fn},{w}"
        else:
            os.environ.pop("IVG_SK_FORCE", None)
        ok = True
        for r in range(REP):
            rc = l.ivg_op_skinny(P(x), P(ws[r % COPIES]), P(y), B, N, K, K, K, ldy, 1, flags, 1, st)
            if rc != 0:
                ok = False
                break
        torch.cuda.synchronize()
        if ok:
            plan.append((name, mf, fn, w, REP))
with open(os.environ.get("SK_PLAN", "/tmp/sk_plan.txt"), "w") as f:
    for p in plan:
        f.write(" ".join(map(str, p)) + "\n")
print("launched", len(plan), "configurations")
