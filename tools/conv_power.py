"""Socket power and shader clock WHILE a 3x3 convolution kernel runs back to back (development aid): is a launch bound by what the
socket may draw?  For each (shape, kernel) arm the convolution is launched in a loop for ~2.5 s while `rocm-smi --showpower
--showclocks` is sampled every 0.25 s.  python tools/conv_power.py [seconds per arm]
Beside tools/ubench/mfma_power.hip: pure MFMA streams on zero / random operands with and without LDS fragment traffic."""
import ctypes as C
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ivideogpt_amd import _lib, switches  # noqa: E402


class Sampler(threading.Thread):
    def __init__(self):
        super().__init__(daemon=True)
        self.rows, self.stop = [], False

    def run(self):
        while not self.stop:
            try:
                out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
            except Exception:   # noqa: BLE001
                out = ""
            p = re.search(r"Power \(W\): ([\d.]+)", out)
            c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
            if p:
                self.rows.append((time.time(), float(p.group(1)), int(c.group(1)) if c else -1))
            time.sleep(0.25)


def arm(lib, name, H, Cin, Cout, ups, gn, N, secs, env, data="randn"):
    switches.set(**env)
    dev = "cuda:0"
    Ho = 2 * H if ups else H
    if data == "zeros":
        x = torch.zeros(N, H, H, Cin, device=dev, dtype=torch.bfloat16)
        w = torch.zeros(Cout, 9 * Cin, device=dev, dtype=torch.bfloat16)
    else:
        x = torch.randn(N, H, H, Cin, device=dev).to(torch.bfloat16)
        w = (torch.randn(Cout, 9 * Cin, device=dev) / (9 * Cin) ** 0.5).to(torch.bfloat16)
    y = torch.empty(N, Ho, Ho, Cout, device=dev, dtype=torch.bfloat16)
    a = _lib.IvgIgemmArgs()
    a.X, a.W, a.Y = x.data_ptr(), w.data_ptr(), y.data_ptr()
    for k, v in dict(Nimg=N, Hin=H, Win=H, Cin=Cin, ldx=Cin, Hout=Ho, Wout=Ho, KH=3, KW=3, stride=1, pad=1, ups=ups, N=Cout, ldw=9 * Cin,
                     c_img=Ho * Ho * Cout, c_pix=Cout, c_ch=1, c_grp=1, c_grp_stride=0, flags=0, alpha=1.0, nb0=1, nb1=1, nb2=1).items():
        setattr(a, k, v)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for _ in range(3):
        assert lib.ivg_op_igemm(C.byref(a), 1, st) == 0
    torch.cuda.synchronize()
    smp = Sampler()
    smp.start()
    t0 = time.time()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < secs:
        for _ in range(50):
            lib.ivg_op_igemm(C.byref(a), 1, st)
        n += 50
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    t1 = time.time()
    smp.stop = True
    smp.join()
    ms = e0.elapsed_time(e1) / n
    rows = [(p, c) for t, p, c in smp.rows if t0 + 0.6 < t < t1]
    pw = [p for p, _ in rows] or [float("nan")]
    ck = [c for _, c in rows] or [-1]
    print(f"{name:58s} {ms:7.3f} ms {2.0 * N * Ho * Ho * Cout * 9 * Cin / ms / 1e9:6.0f} TFLOP/s | power W mean {sum(pw) / len(pw):6.0f} max {max(pw):6.0f} | "
          f"sclk MHz mean {sum(ck) / len(ck):5.0f} min {min(ck)} ({len(rows)} samples)", flush=True)


def main():
    secs = float(sys.argv[1]) if len(sys.argv) > 1 else 2.5
    lib = _lib.load()
    N = 896
    narrow = {}
    print(f"# {N} frames, bf16, plain 3x3 convolution (no bias), ~{secs} s per arm; rocm-smi sampled every 0.25 s")
    print("# (the persistent two-tile arms of round 5 -- profiles/r05_conv_power.txt -- ran against tools/ubench/conv3x3w.hip, no longer in libivg)")
    arm(lib, "16x16 512->512  256-pixel kernel, random data", 16, 512, 512, 0, 0, N, secs, narrow)
    arm(lib, "16x16 512->512  256-pixel kernel, ZERO data", 16, 512, 512, 0, 0, N, secs, narrow, data="zeros")
    arm(lib, "64x64 128->128  256-pixel kernel, random data", 64, 128, 128, 0, 0, N, secs, narrow)
    arm(lib, "32->64 256->256 upsampling, random data", 32, 256, 256, 1, 0, N, secs, narrow)


if __name__ == "__main__":
    main()
