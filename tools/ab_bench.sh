#!/bin/bash
# same-box A/B of the default bench under environment switches: bash tools/ab_bench.sh "APE_USE_PSP_FOLD=0" "APE_USE_PSP_FOLD=1 GEMM_DBG=64" ...
# (each arm twice, interleaved; prints frames/s and ms/step; EXTRA = more bench.py flags)
for rep in 1 2; do
  for arm in "$@"; do
    out=$(env $arm python tools/bench_dbg.py --no-cpu-baseline --no-modes --no-staged --no-sweep --no-latency --no-pose-leg --no-label-leg --steps 30 --warmup 5 $EXTRA 2>/dev/null | tail -1)
    echo "$arm  $(python -c "import json,sys; d=json.loads(sys.argv[1]); print(d['value'], d['ms_per_step'])" "$out")"
  done
done
