"""MI355X-native hot path of KochPJ/AutoPoseEstimation (seg -> DenseFusion -> ICP) behind the reference's Python call signatures.

`install_dropin()` aliases the mirrored LEAF modules under the reference's own import names, so the reference's import lines
(`from DenseFusion.lib.network import PoseNet`, `import pc_reconstruction.open3d_utils as pc_utils`, ...) resolve to this package.

`install_dropin(reference_root=<checkout of the reference>)` additionally keeps the REST of the reference importable, which is what an
unmodified `main.py` needs (main.py:1-18): every aliased package gets the reference's directory appended to its `__path__` (so
`pipeline.grasping_utils`, `label_generator.make_train_and_test_dataset`, `data_generation.getData`, `segmentation.dataset` ... load from
the reference tree), `segmentation`'s package body (the reference's training code, `segmentation/__init__.py`) is executed in the alias
package, and every public name a mirrored module does not define (`get_selection`, `get_True_or_False`, the visualisers of
`pipeline/utils.py:24-380`; `jaccard_loss`, `IoU`, `animate` of `segmentation/utils.py:71-296`; ...) is copied in from the reference's
module of the same name, so `from pipeline.utils import *` sees both.  The mirrored entry points keep priority.  INTEGRATION.md section 1."""
import importlib
import importlib.util
import os
import sys
import warnings

DROPIN_MODULES = (
    "DenseFusion", "DenseFusion.lib", "DenseFusion.lib.network", "DenseFusion.lib.knn", "DenseFusion.lib.loss",
    "DenseFusion.lib.loss_refiner", "DenseFusion.lib.transformations", "DenseFusion.tools", "DenseFusion.tools.utils", "DenseFusion.tools.train",
    "DenseFusion.datasets", "DenseFusion.datasets.myDatasetAugmented", "DenseFusion.datasets.myDatasetAugmented.dataset",
    "segmentation", "segmentation.utils", "pipeline", "pipeline.utils", "label_generator", "label_generator.create_labels",
    "pc_reconstruction", "pc_reconstruction.open3d_utils", "pc_reconstruction.create_pointcloud", "experiments", "experiments.eval",
    "background_subtraction", "background_subtraction.utils", "data_generation",
)


def _is_package(mod):
    return hasattr(mod, "__path__")


def _load_reference_module(name, path):
    """Execute the reference's own source file `path` as a private module (its imports resolve through sys.modules, i.e. to the mirrored
    modules where those exist).  Returns None (with a warning) when it cannot be imported, e.g. a third-party dependency is missing."""
    spec = importlib.util.spec_from_file_location("_ape_reference." + name, path)
    mod = importlib.util.module_from_spec(spec)
    try:
        spec.loader.exec_module(mod)
    except Exception as e:      # noqa: BLE001 -- whatever the reference's import needs and the host lacks
        warnings.warn("install_dropin: reference module %s not importable (%s: %s); its extra names are not forwarded" % (name, type(e).__name__, e))
        return None
    return mod


def install_dropin(force=False, reference_root=None):
    """Register the mirrored modules in sys.modules under the reference's names; returns the list of names installed.
    Existing entries (the real reference already imported) are left alone unless `force`.  `reference_root`: see the module docstring."""
    done = []
    for name in DROPIN_MODULES:
        if force or name not in sys.modules:
            sys.modules[name] = importlib.import_module("autoposeestimation_amd." + name)
            done.append(name)
    if reference_root is None:
        return done
    reference_root = os.path.abspath(reference_root)
    if reference_root not in sys.path:
        sys.path.append(reference_root)            # the reference's un-mirrored top-level packages (robot_controller, depth_camera, ...)
    # 1. packages: the reference's directory behind ours, so un-mirrored submodules still import
    for name in DROPIN_MODULES:
        mod = sys.modules[name]
        ref_dir = os.path.join(reference_root, *name.split("."))
        if _is_package(mod) and os.path.isdir(ref_dir) and ref_dir not in list(mod.__path__):
            mod.__path__.append(ref_dir)
    # 2. leaves: names the mirror lacks come from the reference's module of the same name (mirrored names keep priority)
    for name in DROPIN_MODULES:
        mod = sys.modules[name]
        if _is_package(mod):
            continue
        ref_file = os.path.join(reference_root, *name.split(".")) + ".py"
        if not os.path.isfile(ref_file):
            continue
        ref = _load_reference_module(name, ref_file)
        if ref is None:
            continue
        for attr, val in vars(ref).items():
            if not attr.startswith("_") and not hasattr(mod, attr):
                setattr(mod, attr, val)
    # 3. `import segmentation` (main.py:9) runs the reference's training code in segmentation/__init__.py: give the alias package
    #    that body too (it needs segmentation_models_pytorch; skipped with a warning where that is absent)
    seg_init = os.path.join(reference_root, "segmentation", "__init__.py")
    if os.path.isfile(seg_init):
        ref = _load_reference_module("segmentation.__init__", seg_init)
        if ref is not None:
            seg = sys.modules["segmentation"]
            for attr, val in vars(ref).items():
                if not attr.startswith("_") and not hasattr(seg, attr):
                    setattr(seg, attr, val)
    return done
