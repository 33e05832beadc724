#!/bin/bash
# Regenerates every profiles/rNN_* artefact from the tree as it is (run on the MI355X box through gpurun, from the repo root):
#     gpurun --timeout 1200 -- 'bash tools/make_profiles.sh r04'
# Output goes to gpurun_out/<round>/ (merged back by gpurun); copy it into profiles/ and commit.  The counters run in their own passes
# (--pmc never together with --stats-less tracing domains), every pass with the program itself behind `--`.
set -o pipefail
R=${1:-r05}
PSTEPS=3        # steps of the profiled runs (stored in rNN_meta.json: tests/test_profiles_fresh.py divides the launch counts by it)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$R
mkdir -p $O
T="timeout -k 10 280"
# 1. single stream under the kernel trace: per-kernel stats, timed region, one step's launch list
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_no -- python3 bench.py --steps $PSTEPS --warmup 1 --no-cpu-baseline --no-overlap --no-modes --no-staged --no-sweep --no-latency --no-pose-leg --no-label-leg --no-step-check > $O/prof_no.log 2>&1 || exit 1
cp $O/prof_no/*/*_kernel_stats.csv $O/${R}_bf16x3_kernel_stats.csv
python tools/prof_summary.py $O/prof_no/*/*_kernel_trace.csv --steps $PSTEPS --warmup 2 > $O/${R}_bf16x3_timed_region.json || exit 1
python tools/prof_step_list.py $O/prof_no/*/*_kernel_trace.csv > $O/${R}_step_launch_list.txt || exit 1
# 2. the DEFAULT command (software-pipelined loop) under the kernel trace
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_def -- python3 bench.py --steps $PSTEPS --warmup 1 --no-cpu-baseline --no-modes --no-staged --no-sweep --no-latency --no-pose-leg --no-label-leg --no-step-check > $O/prof_def.log 2>&1 || exit 1
cp $O/prof_def/*/*_kernel_stats.csv $O/${R}_default_overlap_kernel_stats.csv
python tools/prof_summary.py $O/prof_def/*/*_kernel_trace.csv --steps $PSTEPS --warmup 2 --tail-steps 3 > $O/${R}_default_overlap_timed_region.json || exit 1
# 3. HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes, split per layer shape
$T rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-overlap --no-modes --no-staged --no-sweep --no-latency --no-pose-leg --no-label-leg --no-step-check --dump-launches $O/launches.json > $O/pmc_fetch.log 2>&1 || exit 1
$T rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-overlap --no-modes --no-staged --no-sweep --no-latency --no-pose-leg --no-label-leg --no-step-check > $O/pmc_write.log 2>&1 || exit 1
python tools/pmc_summary.py $O/pmc_fetch/*/*_counter_collection.csv $O/pmc_write/*/*_counter_collection.csv --shapes $O/launches.json > $O/${R}_pmc_traffic.json || exit 1
cp $O/${R}_pmc_traffic.json profiles/      # (this box's copy of the tree: step 6's bench line reads its `traffic` fields from the newest profiles/*_pmc_traffic.json)
# 4. matrix-pipe / vector / LDS counters of the largest kernels
$T rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_util -- python3 tools/pmc_big3.py > $O/pmc_util.log 2>&1 || exit 1
python tools/pmc_big3.py --reduce $O/pmc_util/*/*_counter_collection.csv > $O/${R}_pmc_utilisation.txt || exit 1
# 5. the label workload
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_label -- python3 bench.py --workload label --steps 3 --warmup 1 --no-cpu-baseline > $O/prof_label.log 2>&1 || exit 1
cp $O/prof_label/*/*_kernel_stats.csv $O/${R}_label_kernel_stats.csv
# 5b. the pose workload (BASELINE configs[1]: 32 crops -> PoseNet + 2 refiner passes + ADD-S, and the k-NN kernel at the training loss's size)
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_pose -- python3 bench.py --workload pose --steps 20 --warmup 3 --no-cpu-baseline > $O/prof_pose.log 2>&1 || exit 1
cp $O/prof_pose/*/*_kernel_stats.csv $O/${R}_pose_kernel_stats.csv
# 6. the un-profiled default line of the same tree (with the CPU baseline and the parity block) and the freshness stamp
$T python3 bench.py --steps 20 --warmup 5 > $O/bench_default.log 2>&1 || exit 1
python - "$O" "$R" "$PSTEPS" <<'PY'
import json, sys, os
sys.path.insert(0, "tools")
from profile_stamp import source_hash
o, r, psteps = sys.argv[1], sys.argv[2], int(sys.argv[3])
line = [l for l in open(os.path.join(o, "bench_default.log")) if l.startswith("{")][-1]
json.dump({"source_hash": source_hash(), "round": int(r[1:]), "profile_steps": psteps, "bench_line": json.loads(line),
           "made_by": "tools/make_profiles.sh %s (rocprofv3 passes of bench.py on one MI355X, all from one tree)" % r}, open(os.path.join(o, r + "_meta.json"), "w"), indent=1)
PY
rm -rf $O/prof_no $O/prof_def $O/pmc_fetch $O/pmc_write $O/pmc_util $O/prof_label $O/prof_pose
ls -la $O
