"""Point-cloud files in the layouts open3d 0.9 writes by default (what an existing reference dataset holds,
pc_reconstruction/create_pointcloud.py:324-344): binary_little_endian PLY with double xyz (+ normals, colours), `DATA binary` PCD with
float32 fields, their ASCII forms, and the round trip of this package's own writer.  The byte layouts are spelled here with struct,
independently of the reader."""
import os
import struct

import numpy as np
import pytest

from autoposeestimation_amd.pc_reconstruction import pointcloud as PC


def _pts():
    return np.random.default_rng(4).standard_normal((11, 3)) * 120.0


def test_read_open3d_binary_ply_with_extra_properties(tmp_path):
    pts = _pts()
    raw = (b"ply\nformat binary_little_endian 1.0\ncomment Created by Open3D\nelement vertex 11\nproperty double x\nproperty double y\n"
           b"property double z\nproperty double nx\nproperty double ny\nproperty double nz\nproperty uchar red\nproperty uchar green\n"
           b"property uchar blue\nend_header\n")
    raw += b"".join(struct.pack("<6d3B", *p, 0.0, 0.0, 1.0, 10, 20, 30) for p in pts)
    assert np.array_equal(PC._read_ply(raw), pts)
    big = raw.replace(b"binary_little_endian", b"binary_big_endian").split(b"end_header\n")[0] + b"end_header\n" + \
        b"".join(struct.pack(">6d3B", *p, 0.0, 0.0, 1.0, 10, 20, 30) for p in pts)
    assert np.array_equal(PC._read_ply(big), pts)
    f32 = (b"ply\nformat binary_little_endian 1.0\nelement vertex 11\nproperty float x\nproperty float y\nproperty float z\n"
           b"element face 0\nproperty list uchar int vertex_indices\nend_header\n") + b"".join(struct.pack("<3f", *p) for p in pts)
    assert np.array_equal(PC._read_ply(f32), pts.astype(np.float32).astype(np.float64))


def test_read_binary_and_ascii_pcd(tmp_path):
    pts = _pts()
    head = (b"# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z rgb\nSIZE 4 4 4 4\nTYPE F F F F\nCOUNT 1 1 1 1\n"
            b"WIDTH 11\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\nPOINTS 11\nDATA binary\n")
    raw = head + b"".join(struct.pack("<4f", *p, 0.25) for p in pts)
    assert np.array_equal(PC._read_pcd(raw), pts.astype(np.float32).astype(np.float64))
    asc = head.replace(b"DATA binary", b"DATA ascii") + "".join("%r %r %r 0.25\n" % tuple(float(v) for v in p) for p in pts).encode()
    assert np.array_equal(PC._read_pcd(asc), pts)
    with pytest.raises(NotImplementedError):
        PC._read_pcd(head.replace(b"DATA binary", b"DATA binary_compressed") + b"\0" * 16)


@pytest.mark.parametrize("ext", [".ply", ".pcd"])
@pytest.mark.parametrize("ascii_", [False, True])
def test_writer_layout_and_round_trip(tmp_path, ext, ascii_):
    pts = _pts()

    class Cloud:
        points = pts

    path = os.path.join(tmp_path, "c" + ext)
    PC.write_point_cloud(path, Cloud(), write_ascii=ascii_)
    raw = open(path, "rb").read()
    if ext == ".ply":
        assert (b"format ascii 1.0" if ascii_ else b"format binary_little_endian 1.0") in raw and b"property double x" in raw
        assert np.array_equal(PC._read_ply(raw), pts)
        if not ascii_:      # open3d's default: the doubles follow the header directly
            assert raw.endswith(pts.astype("<f8").tobytes())
    else:
        assert (b"DATA ascii" if ascii_ else b"DATA binary") in raw and b"SIZE 8 8 8" in raw
        assert np.array_equal(PC._read_pcd(raw), pts)
