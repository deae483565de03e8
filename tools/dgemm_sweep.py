"""Sweep the tile shapes (MF, FN, WAVES, LG) of the decode-step GEMM (csrc/dgemm.hip) over the per-layer shapes of a model.
Each configuration is timed as a hipGraph of REP dependent launches (what a decode step is made of): microseconds per launch
INCLUDING the kernel boundary.  Weights rotate through enough copies that they come from HBM (a decode step streams 252 MB of
weights and ~1.5 GB of KV between two uses of the same matrix).  IVG_DG=0 rows are the first-generation kernel (skinny.hip).
Usage: python tools/dgemm_sweep.py [hidden] [intermediate] [batch] [vocab]"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ivideogpt_amd import _lib  # noqa: E402

H = int(sys.argv[1]) if len(sys.argv) > 1 else 768
I = int(sys.argv[2]) if len(sys.argv) > 2 else 3072
B = int(sys.argv[3]) if len(sys.argv) > 3 else 64
V = int(sys.argv[4]) if len(sys.argv) > 4 else 16386
quick = os.environ.get("SWEEP_QUICK") == "1"
l = _lib.load()
P = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
SK_NORM, RES, GLU, F32 = 64, 4, 16, 32
shapes = [("qkv", 3 * H, H, SK_NORM), ("oproj", H, H, RES), ("gateup", 2 * I, H, GLU | SK_NORM), ("down", H, I, RES),
          ("lm_head", V, H, SK_NORM | F32)]
REP = 48
stream = torch.cuda.Stream()


_bufs = {}


def buffers(name, N, K, flags):
    if name not in _bufs:
        _bufs.clear()
        torch.cuda.empty_cache()
        ldy = N // 2 if flags & GLU else N
        copies = max(8, min(256, int(700e6 // (N * K * 2))))
        x = torch.randn(B, K, device="cuda").to(torch.bfloat16)
        ws = [(torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16) for _ in range(copies)]
        y = torch.zeros(B, ldy, device="cuda", dtype=torch.float32 if flags & F32 else torch.bfloat16)
        _bufs[name] = (x, ws, y, ldy, copies)
    return _bufs[name]


def time_cfg(name, N, K, flags, env):
    for k in ("IVG_DG", "IVG_DG_FORCE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    x, ws, y, ldy, copies = buffers(name, N, K, flags)
    st = C.c_void_p(stream.cuda_stream)
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        rc = l.ivg_op_skinny(P(x), P(ws[0]), P(y), B, N, K, K, K, ldy, 1, flags, 1, st)   # eager: attribute set-up
        if rc != 0:
            return None
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for r in range(REP):
                rc |= l.ivg_op_skinny(P(x), P(ws[r % copies]), P(y), B, N, K, K, K, ldy, 1, flags, 1, st)
        if rc != 0:
            return None
        g.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            g.replay()
            b.record(stream)
            torch.cuda.synchronize()
            ts.append(a.elapsed_time(b) * 1e3 / REP)
    return sorted(ts)[len(ts) // 2]


best = {}
for name, N, K, flags in shapes:
    chunks = K * 2 // 16
    base = time_cfg(name, N, K, flags, {"IVG_DG": "0"})
    dflt = time_cfg(name, N, K, flags, {})
    print(f"{name:8s} gen1            {base:7.2f} us/launch", flush=True)
    print(f"{name:8s} gen2 default    {dflt if dflt is not None else float('nan'):7.2f} us/launch", flush=True)
    best[name] = (dflt or 1e9, "default")
    splits = [(w, lg) for lg in (3, 2, 1) for w in (1, 2, 4, 6, 8, 12, 16) if chunks % (w * 8 * lg) == 0 and chunks // (w * 8 * lg) <= 8]
    if quick:
        splits = splits[:3]
    for mf in (1, 2, 4):
        for fn in (1, 2, 4):
            if (flags & GLU) and fn < 2:
                continue
            for w, lg in splits:
                if w * lg * mf * 2048 > 160 * 1024:
                    continue
                t = time_cfg(name, N, K, flags, {"IVG_DG_FORCE": f"{mf},{fn},{w},{lg}"})
                if t is None:
                    continue
                print(f"{name:8s} {mf},{fn},{w:2d},{lg}        {t:7.2f} us/launch", flush=True)
                if t < best[name][0]:
                    best[name] = (t, f"{mf},{fn},{w},{lg}")
print("best:", best)
