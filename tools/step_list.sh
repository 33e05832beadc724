#!/bin/bash
# one step's launch list of the default bench (single stream) under rocprofv3's kernel trace -> gpurun_out/<name>_step_list.txt
set -o pipefail
N=${1:-steplist}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$N
mkdir -p $O
timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-overlap --no-modes --no-staged --no-sweep --no-latency --no-pose-leg --no-label-leg --no-step-check > $O/prof.log 2>&1 || exit 1
python tools/prof_step_list.py $O/prof/*/*_kernel_trace.csv > gpurun_out/${N}_step_list.txt || exit 1
rm -rf $O/prof
tail -1 gpurun_out/${N}_step_list.txt
