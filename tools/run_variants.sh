for v in "$@"; do
  echo "== $v"
  APE_HIP_LIB=autoposeestimation_amd/libape_hip_$v.so python tools/stress_upfuse.py --reps 20 --shapes 2x240x320,16x240x320 2>&1 | grep -v amdgpu.ids
done
