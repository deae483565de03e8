#!/bin/bash
# round-3 GPU session 5: whole-line gemm256 (tests + A/B), split-operand estimate for the fp32 convolutions
set -u
O=gpurun_out/r03_s5; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py -q -m gpu --tb=short -p no:cacheprovider -k "gemm256 or glu or igemm" > $O/pytest_ops.txt 2>&1
tail -4 $O/pytest_ops.txt
timeout 900 python -m pytest tests/test_gpu_models.py -q -m gpu --tb=short -p no:cacheprovider -k "llama or flash or bf16" > $O/pytest_models.txt 2>&1
tail -4 $O/pytest_models.txt
for e in "IVG_G256_LINE=0" "IVG_G256_LINE=1"; do
  echo "== $e" >> $O/bench.txt; env $e timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-fp32-mode >> $O/bench.txt 2>&1
done
python - <<'PY'
import json
for l in open("gpurun_out/r03_s5/bench.txt"):
    if l.startswith("=="): print(l.strip())
    if l.startswith("{"):
        d=json.loads(l); print(d["value"], d["ms_per_step"], d["stage_ms"], [ (r["kernel"][:22], round(r["kernel_ms_per_step"],1), round(r["frac"],3)) for r in [d["roofline"]]+d["roofline_other"]])
PY
( echo "# fp32 3x3 convolutions of the encoder (128 context frames of config 2) on the f32-input MFMA path vs the same layer as a bf16 convolution over"
  echo "# 6x (bf16x6 split operands: hi*hi + hi*mid + mid*hi + hi*lo + lo*hi + mid*mid) and 3x the input channels -- the MFMA-side cost of a"
  echo "# split-operand scheme, WITHOUT the pass that would split the activations and without the 3x larger activation tensor in HBM."
  for cfg in "64 128 128" "32 256 256" "16 512 512"; do
    set -- $cfg
    python tools/conv_bench.py $1 $2 $3 0 128 fp32
    python tools/conv_bench.py $1 $(( $2 * 6 )) $3 0 128 bf16
    python tools/conv_bench.py $1 $(( $2 * 3 )) $3 0 128 bf16
  done ) > $O/split_operand_estimate.txt 2>&1
cat $O/split_operand_estimate.txt
echo done > $O/done.txt
