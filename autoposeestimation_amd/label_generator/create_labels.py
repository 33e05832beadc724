"""Drop-in for `get_default_model` of label_generator/create_labels.py (reference :20-37).

The reference hard-codes smp's Unet-resnet34; that third-party model is unavailable (see segmentation/utils.py), so the
default here is the in-repo PSPNet ('PsPNet', resnet34 encoder) with the same config keys and checkpoint convention:
`<root>/segmentation/trained_models/<ds_name>/<name>_<encoder>.ckpt` holding {'state_dict': ...}."""
import os

import torch

from autoposeestimation_amd.segmentation.utils import get_model


def get_default_model(root, ds_name, n_classes, name="PsPNet", encoder_name="resnet34", load=True):
    segmentation_config = {"encoder_name": encoder_name,
                           "encoder_weights": None,
                           "activation": "softmax",
                           "in_channels": 3,
                           "classes": n_classes}
    model = get_model(name, segmentation_config)
    if load:
        cp = torch.load(os.path.join(root, "segmentation", "trained_models", ds_name,
                                     "{}_{}.ckpt".format(name, segmentation_config["encoder_name"])),
                        map_location=torch.device("cpu"))
        model.load_state_dict(cp["state_dict"])
    return model
