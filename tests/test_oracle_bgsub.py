"""oracle/bgsub_oracle.py against the goldens produced by the reference's own get_mask_prediction / do_cca
(tools/gen_golden_bgsub.py -> tests/golden/bgsub.npz).  CPU only."""
import os

import numpy as np

from oracle import bgsub_oracle as O

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "bgsub.npz"))


def test_pil_hsv_matches_pillow_sample():
    assert np.array_equal(O.pil_rgb_to_hsv(G["hsv_rgb"]), G["hsv_ref"])


def test_pil_hsv_matches_installed_pillow():
    from PIL import Image
    rng = np.random.default_rng(3)
    rgb = rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)
    rgb[0, :8] = [[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [1, 0, 0], [254, 255, 255], [7, 7, 7]]
    assert np.array_equal(O.pil_rgb_to_hsv(rgb), np.array(Image.fromarray(rgb, "RGB").convert("HSV")))


def _gate(tag):
    if tag == "nogate":
        return O.depth_gate(None)
    meta = {"robot2endEff_tf": G["robot2endEff_tf"].tolist(), "hand_eye_calibration": G["hand_eye_calibration"].tolist()}
    return O.depth_gate(O.measure_distance(meta, G["reference_point"]))


def test_features_match_reference_bitwise():
    for tag in ("gate", "nogate"):
        gate = _gate(tag)
        for i in range(G["f_rgb"].shape[0]):
            x_u8, x = O.subtraction_features(G["f_rgb"][i], G["b_rgb"][i], G["f_depth"][i], G["b_depth"][i], gate)
            assert np.array_equal(x, G["x_" + tag][i]), (tag, i)
            assert x_u8.shape == (48, 64, 7)
    assert not np.array_equal(G["x_gate"], G["x_nogate"])       # the gate really changes the depth channel


def test_depth_channel_wraps_like_numpy_cast():
    f = np.zeros((2, 2), np.uint16)
    b = np.zeros((2, 2), np.uint16)
    f[0, 0], b[0, 0] = 1400, 100           # |diff| = 1300 -> 1300 & 255 = 20
    f[0, 1], b[0, 1] = 1600, 100           # f beyond the 1500 gate -> both zero
    f[1, 0], b[1, 0] = 700, 0              # b invalid -> both zero
    rgb = np.zeros((2, 2, 3), np.uint8)
    x_u8, _ = O.subtraction_features(rgb, rgb, f, b, O.depth_gate(None))
    assert x_u8[..., 6].tolist() == [[20, 0], [0, 0]]


def test_do_cca_matches_reference():
    for tag in ("gate", "nogate"):
        out = O.do_cca(G["preds"])
        assert np.array_equal((out != 0).astype(np.uint8) * 255, G["label_" + tag])
    assert np.array_equal(O.do_cca(G["cca_in"]), G["cca_out"])
    assert G["cca_out"][0, 4:10, 4:10].all() and not G["cca_out"][0, 30:36, 40:46].any()    # first blob wins the tie
    assert not G["cca_out"][1].any()                                                         # no foreground -> empty label
