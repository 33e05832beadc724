"""On-disk sample format of the reference, host side:

    data_generation/data/<obj>/<dir>/<id>.color.png   RGB8            (data_generation/getData.py:177-221)
    data_generation/data/<obj>/<dir>/<id>.depth.png   uint16 sensor units
    data_generation/data/<obj>/<dir>/<id>.meta.json   joints, pose, object_pose[16], robot2endEff_tf[16], intr{...},
                                                      depth_scale, symmetric, hand_eye_calibration[16], view_point_id
    label_generator/data/<obj>/<dir>/<id>.{gen,pred,new_pred}.label.png   uint8 {0,255}
    label_generator/data/<obj>/<dir>/<id>.meta.json   pose label (label_generator/create_labels.py:422-429)

PNG decode/encode goes through Pillow on the host (the decoded arrays are uploaded once and everything after stays on
the GPU); ids are the 6-digit stems the reference uses (`'{:06d}'.format(idx)`)."""
import json
import os

import numpy as np
from PIL import Image


def list_samples(directory):
    """sorted ids of a sample directory (reference: `sorted([d[:-10] for d in samples if '.color.png' in d])`)"""
    return sorted(f[:-10] for f in os.listdir(directory) if f.endswith(".color.png"))


def read_meta(directory, sample_id):
    with open(os.path.join(directory, "{}.meta.json".format(sample_id))) as f:
        return json.load(f)


def read_color(directory, sample_id):
    with open(os.path.join(directory, "{}.color.png".format(sample_id)), "rb") as f:
        return np.array(Image.open(f).convert("RGB"), dtype=np.uint8)


def read_depth(directory, sample_id):
    with open(os.path.join(directory, "{}.depth.png".format(sample_id)), "rb") as f:
        d = np.array(Image.open(f))
    if d.dtype != np.uint16:
        d = d.astype(np.uint16)
    return d


def read_label(label_dir, sample_id, mode):
    with open(os.path.join(label_dir, "{}.{}.label.png".format(sample_id, mode)), "rb") as f:
        return np.array(Image.open(f), dtype=np.uint8)


def write_label(label_dir, sample_id, mode, label):
    os.makedirs(label_dir, exist_ok=True)
    Image.fromarray(np.asarray(label, dtype=np.uint8)).save(os.path.join(label_dir, "{}.{}.label.png".format(sample_id, mode)))


def write_sample(directory, sample_id, rgb, depth, meta):
    os.makedirs(directory, exist_ok=True)
    Image.fromarray(np.asarray(rgb, dtype=np.uint8)).save(os.path.join(directory, "{}.color.png".format(sample_id)))
    Image.fromarray(np.asarray(depth, dtype=np.uint16)).save(os.path.join(directory, "{}.depth.png".format(sample_id)))
    with open(os.path.join(directory, "{}.meta.json".format(sample_id)), "w") as f:
        json.dump(meta, f)


def robot2cam(meta):
    """robot2endEff_tf . hand_eye_calibration (create_labels.py:104-106, create_pointcloud.py:246-249)"""
    return np.dot(np.array(meta.get("robot2endEff_tf"), dtype=np.float64).reshape(4, 4),
                  np.array(meta.get("hand_eye_calibration"), dtype=np.float64).reshape(4, 4))
