"""`install_dropin(reference_root=...)` followed by `import main` -- the reference's TUI module, unmodified --
must import and bind the hot-path entry points to this package while everything else stays the reference's.  Build container only
(skipped where /root/reference is absent, i.e. on the GPU box); third-party modules the image lacks (open3d, cv2, smp, ...) are empty
stand-ins for import purposes only.  Runs in a child interpreter so that the aliases do not leak into the test process."""
import os
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE = "/root/reference"

SCRIPT = r'''
import sys, types, warnings
warnings.filterwarnings("ignore")
sys.path.insert(0, %(repo)r)
sys.path.insert(0, %(repo)r + "/tools")
import numpy as np
if not hasattr(np, "float"):
    np.float = float
from ref_shim import _Anything, _STUBBED
for name in list(_STUBBED) + ["matplotlib.patches", "mpl_toolkits", "mpl_toolkits.mplot3d", "mpl_toolkits.mplot3d.proj3d", "sklearn",
                              "sklearn.model_selection", "skimage", "skimage.measure", "albumentations", "tqdm"]:
    if name not in sys.modules:
        try:
            __import__(name)
        except Exception:
            sys.modules[name] = _Anything(name)
if isinstance(sys.modules.get("matplotlib.patches"), _Anything):
    sys.modules["matplotlib.patches"].FancyArrowPatch = type("FancyArrowPatch", (), {})
import autoposeestimation_amd as A
A.install_dropin(reference_root=%(ref)r)
import main                                                   # the reference TUI, unchanged
import autoposeestimation_amd.pipeline.utils as ours
assert main.full_prediction is ours.full_prediction and main.get_prediction_models is ours.get_prediction_models
assert main.get_selection.__module__.startswith("_ape_reference."), main.get_selection.__module__      # pipeline/utils.py:24, via `import *`
assert callable(main.get_True_or_False) and callable(main.run_live_prediction) and callable(main.create_pose_data)
assert main.label_gen.create_pose_data.__module__ == "autoposeestimation_amd.label_generator.create_labels"
assert main.segmentation_utils.get_model.__module__ == "autoposeestimation_amd.segmentation.utils"
assert hasattr(main.segmentation_utils, "jaccard_loss") and hasattr(main.segmentation_utils, "IoU")      # the reference's training helpers
assert main.get_mask_prediction.__module__ == "autoposeestimation_amd.background_subtraction.utils"
assert main.grasp_utils.__file__.startswith(%(ref)r) and main.data_gen.__file__.startswith(%(ref)r)       # un-mirrored modules: the reference's
assert main.pose_estimation.train_step.__module__ == "autoposeestimation_amd.DenseFusion.tools.train"
from DenseFusion.lib.network import PoseNet
assert PoseNet.__module__ == "autoposeestimation_amd.DenseFusion.lib.network"
print("DROPIN-OK")
'''


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference checkout exists in the build container only")
def test_reference_main_imports_after_install_dropin():
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"repo": REPO, "ref": REFERENCE}], capture_output=True, text=True, timeout=600,
                       cwd=REFERENCE)
    assert r.returncode == 0 and "DROPIN-OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
