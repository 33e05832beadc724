"""profiles/ must describe the tree it is committed with (the r03 profiles predated a kernel-routing commit, so a
per-kernel PMC average was divided by another population of launches).  The newest round's `rNN_meta.json` (tools/make_profiles.sh) holds
the hash of the kernel / routing sources it was generated from and the un-profiled bench line of the same tree:
  * the hash must equal the tree's (tools/profile_stamp.py): a kernel or routing change without regenerated profiles fails here;
  * every kernel of the line's roofline block must show the same launches per step in the rocprofv3 timed-region summary;
  * every per-shape HBM-traffic row the line cites must exist in the PMC summary."""
import glob
import json
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))


def _newest():
    import re
    metas = [(int(m.group(1)), p) for p in glob.glob(os.path.join(REPO, "profiles", "r*_meta.json"))
             for m in [re.match(r"r(\d+)_meta\.json$", os.path.basename(p))] if m]
    if not metas:
        pytest.skip("no rNN_meta.json yet (profiles of this round not generated)")
    _, path = max(metas)                              # numerically newest round (r100 sorts after r99)
    meta = json.load(open(path))
    return os.path.basename(path)[:-len("_meta.json")], meta


def test_profiles_were_generated_from_this_tree():
    from profile_stamp import source_hash
    rnd, meta = _newest()
    assert meta["source_hash"] == source_hash(), ("profiles/%s_* predate a change of the kernels / routing / bench accounting: regenerate them "
                                                  "(gpurun -- 'bash tools/make_profiles.sh %s') and commit" % (rnd, rnd))


def test_launches_per_step_agree_between_the_bench_line_and_rocprof():
    rnd, meta = _newest()
    line = meta["bench_line"]
    timed = json.load(open(os.path.join(REPO, "profiles", "%s_default_overlap_timed_region.json" % rnd)))
    by_name = {k.replace(" ", ""): v for k, v in timed.items()}
    steps_prof = meta.get("profile_steps", 3)        # tools/make_profiles.sh stores the step count of its profiled runs (r04 and older: 3)
    assert line["roofline"]["kernels"], "the line carries no roofline kernels"
    for k in line["roofline"]["kernels"]:
        name = k["kernel"].replace(" ", "")
        assert name in by_name, "%s of the bench line is not in the rocprofv3 summary" % name
        per_step_line = k["launches"] / line["steps"]
        per_step_prof = by_name[name]["timed_region"]["launches"] / steps_prof
        assert per_step_line == per_step_prof, (name, per_step_line, per_step_prof)
        # and the live HIP-event duration agrees with rocprofv3's within 15 % (another box, the profiler's own overhead)
        assert abs(k["isolated_avg_launch_us"] - by_name[name]["isolated_pass"]["avg_us"]) <= 0.15 * k["isolated_avg_launch_us"], name


def test_per_shape_traffic_rows_exist_for_the_roofline_kernels():
    rnd, meta = _newest()
    pmc = json.load(open(os.path.join(REPO, "profiles", "%s_pmc_traffic.json" % rnd)))["kernels"]
    for k in meta["bench_line"]["roofline"]["kernels"]:
        name = k["kernel"].replace(" ", "")
        assert name in pmc and "shapes" in pmc[name], name
        for s in k["shapes"]:
            assert s["shape"] in pmc[name]["shapes"], (name, s["shape"])
            assert s.get("traffic") == pmc[name]["shapes"][s["shape"]]["hbm_bytes_per_launch"]
