#!/bin/bash
# round 5, session 16: decode attention with fewer workgroups per CU (LDS request padded) while four batches are in flight.
# NOTE: the switch it drove, IVG_ATTN_LDS_PAD_KB (launch_decode_attn: smem = max(smem, pad)), existed only for this session and was
# removed afterwards (measured, not kept: NOTES_r05.md section 5b); the script is kept as the record of what was run.
set -u
R=$(pwd); O=$R/gpurun_out/r05_s16; mkdir -p $O
run() {
  tag=$1; shift
  env "$@" timeout 400 python bench.py --steps 12 --warmup 2 --no-cpu-baseline --no-fp32-mode --no-other-configs --only-lanes > $O/bench_$tag.json 2> $O/bench_$tag.err
}
run base X=1
run pad54 IVG_ATTN_LDS_PAD_KB=54
run pad41 IVG_ATTN_LDS_PAD_KB=41
run base2 X=1
run pad54b IVG_ATTN_LDS_PAD_KB=54
echo done > $O/done.txt
