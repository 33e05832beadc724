"""MI355X-native hot path of KochPJ/AutoPoseEstimation (seg -> DenseFusion -> ICP) behind the reference's Python call signatures.

`install_dropin()` aliases the mirrored modules under the reference's own import names, so an unmodified `main.py`
("Run Live Prediction", "Create Pose labels") resolves `DenseFusion.lib.network`, `segmentation.utils`, `pipeline.utils`, ...
to this package."""
import importlib
import sys

DROPIN_MODULES = (
    "DenseFusion", "DenseFusion.lib", "DenseFusion.lib.network", "DenseFusion.lib.knn", "DenseFusion.lib.loss",
    "DenseFusion.lib.loss_refiner", "DenseFusion.lib.transformations", "DenseFusion.tools", "DenseFusion.tools.utils", "DenseFusion.tools.train",
    "DenseFusion.datasets", "DenseFusion.datasets.myDatasetAugmented", "DenseFusion.datasets.myDatasetAugmented.dataset",
    "segmentation", "segmentation.utils", "pipeline", "pipeline.utils", "label_generator", "label_generator.create_labels",
    "pc_reconstruction", "pc_reconstruction.open3d_utils", "pc_reconstruction.create_pointcloud", "experiments", "experiments.eval",
    "background_subtraction", "background_subtraction.utils",
)


def install_dropin(force=False):
    """Register the mirrored modules in sys.modules under the reference's names; returns the list of names installed.
    Existing entries (the real reference already imported) are left alone unless `force`."""
    done = []
    for name in DROPIN_MODULES:
        if force or name not in sys.modules:
            sys.modules[name] = importlib.import_module("autoposeestimation_amd." + name)
            done.append(name)
    return done
