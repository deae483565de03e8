#!/bin/bash
# round 6, session 3: sub-pixel form of the upsampling convolutions -- op test vs fp64, the model-level decode parity tests, then the
# decode stage with IVG_SUBPIXEL on / off at 64x64 (B = 64) and 256x256 (B = 16), bf16 and x3
set -u
R=$(pwd); O=$R/gpurun_out/r06_s3; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -p no:cacheprovider --tb=short -k "subpixel or conv3x3 or conv_modes" > $O/pytest_ops.txt 2>&1
tail -6 $O/pytest_ops.txt
timeout 1500 python -m pytest tests/test_gpu_models.py tests/test_gpu_x3.py tests/test_gpu_edges.py tests/test_gpu_fullsize.py tests/test_gpu_bf16_deviation.py -q -x -p no:cacheprovider --tb=short -k "not llama and not decode_path and not rollout" > $O/pytest_models.txt 2>&1
tail -6 $O/pytest_models.txt
for SUB in 1 0 1 0; do
  echo "64x64 bf16 IVG_SUBPIXEL=$SUB: $(IVG_DEV=1 IVG_SUBPIXEL=$SUB timeout 300 python tools/quick_bench.py --decode-only --iters 6 2>&1 | tail -1)"
done
for SUB in 1 0; do
  echo "256x256 bf16 IVG_SUBPIXEL=$SUB: $(IVG_DEV=1 IVG_SUBPIXEL=$SUB timeout 300 python tools/quick_bench.py --decode-only --iters 4 --res 256 --batch 16 2>&1 | tail -1)"
  echo "64x64 x3 IVG_SUBPIXEL=$SUB: $(IVG_DEV=1 IVG_SUBPIXEL=$SUB timeout 300 python tools/quick_bench.py --decode-only --iters 4 --dec x3 2>&1 | tail -1)"
done
echo done > $O/done.txt
