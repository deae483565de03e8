"""Launch the rollout's sampler kernel a few times on bench-shaped logits (run under rocprofv3 --kernel-trace to time it)."""
import ctypes as C
import sys

import torch

from ivideogpt_amd import _lib

V = int(sys.argv[1]) if len(sys.argv) > 1 else 16386
l = _lib.load()
g = torch.Generator().manual_seed(0)
lg = (torch.randn(64, V, generator=g) * 3).cuda()
u = torch.rand(64, generator=g).cuda()
out = torch.zeros(64, dtype=torch.int64, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for k in (100, 100, 100, 100, 1000, 1000):
    assert l.ivg_op_sample(C.c_void_p(lg.data_ptr()), 64, V, k, 1.0, C.c_void_p(u.data_ptr()), C.c_void_p(out.data_ptr()), st) == 0
for _ in range(3):
    assert l.ivg_op_sample(C.c_void_p(lg.data_ptr()), 64, V, 100, 1.0, None, C.c_void_p(out.data_ptr()), st) == 0
torch.cuda.synchronize()
print("ok", out[:4].tolist())
