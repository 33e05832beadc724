"""bench.py with the kernels' debug bits set from the environment (GEMM_DBG, HALO_DBG): same-box A/B of kernel variants in the whole loop"""
import os, sys, runpy
# the switches exist in the ablation build only (make -C autoposeestimation_amd/csrc ablations); plain A/B runs keep the product library
# (HALO_DBG bit 4096 = the one-barrier-per-tap halo_s32 of rounds 2-5, selected on the host side: exists in the product library too)
if int(os.environ.get("GEMM_DBG", "0")) or (int(os.environ.get("HALO_DBG", "0")) & ~4096):
    os.environ.setdefault("APE_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "autoposeestimation_amd", "libape_hip_abl.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoposeestimation_amd import _lib
_lib.lib().ape_conv_gemm_s32_debug(int(os.environ.get("GEMM_DBG", "0")))
_lib.lib().ape_conv3x3_halo_s32_debug(int(os.environ.get("HALO_DBG", "0")))
sys.argv = ["bench.py"] + sys.argv[1:]
runpy.run_path(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"), run_name="__main__")
