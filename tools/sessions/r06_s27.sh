#!/bin/bash
# round 6, session 27: GroupNorm + SiLU inside the consuming convolution's staging -- per layer width?
set -u
R=$(pwd); O=$R/gpurun_out/r06_s27; mkdir -p $O; export TMPDIR=/tmp
for arm in 1 2 3 0 1 2 3 0; do
echo "64x64 decode IVG_GN_APPLY_FUSE=$arm: $(IVG_DEV=1 IVG_GN_APPLY_FUSE=$arm timeout 300 python tools/quick_bench.py --decode-only --iters 6 2>&1 | tail -1 | cut -c1-100)"
done
for arm in 1 2 3 0; do
echo "256x256 decode IVG_GN_APPLY_FUSE=$arm: $(IVG_DEV=1 IVG_GN_APPLY_FUSE=$arm timeout 300 python tools/quick_bench.py --decode-only --iters 4 --res 256 --batch 16 2>&1 | tail -1 | cut -c1-100)"
done
echo done > $O/done.txt
