#!/bin/bash
# round 5, session 20: the model-level parity tests with the persistent two-tile convolution forced wherever it covers the shape
# (IVG_CONV_WIDE=2): the opt-in path inside the real decoders (residuals, output statistics, fused norms, 64 / 256 resolution)
set -u
R=$(pwd); O=$R/gpurun_out/r05_s20; mkdir -p $O
IVG_CONV_WIDE=2 timeout 1500 python -m pytest tests/test_gpu_models.py tests/test_gpu_fullsize.py tests/test_gpu_edges.py tests/test_gpu_bf16_deviation.py -q -p no:cacheprovider --tb=short > $O/pytest_wide_models.txt 2>&1
tail -6 $O/pytest_wide_models.txt
IVG_CONV_WIDE=2 python - <<'PY'
import ctypes as C, torch
from ivideogpt_amd import _lib, CompressiveVQModel, weights as W
l = _lib.load()
tcfg = W.tokenizer_config(**W.CTX_VAE64)
tok = CompressiveVQModel(tcfg, W.random_tokenizer_state_dict(tcfg, 0, 0.4), encode_dtype="fp32", decode_dtype="bf16").to("cuda:0")
n0 = l.ivg_debug_counter(b"conv3x3_wide")
ids = torch.randint(0, 8192, (64, 257 * 2 - 1 + 17 * 14), device="cuda:0")
ids[:, 256] = 16384; ids[:, 513::17] = 16385
fr = tok.detokenize(ids, 2)
torch.cuda.synchronize()
print("persistent-kernel launches in one config-2 detokenize:", l.ivg_debug_counter(b"conv3x3_wide") - n0, "finite", bool(torch.isfinite(fr).all()))
PY
echo done > $O/done.txt
