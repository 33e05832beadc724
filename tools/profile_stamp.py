"""Hash of the sources that decide WHICH kernels a bench step launches and how fast they are (the HIP kernels, the host routing, bench.py's
own launch accounting).  tools/make_profiles.sh stores it beside the profiles it generates; tests/test_profiles_fresh.py recomputes it, so
profiles/ that predate a kernel or routing change fail the CPU suite instead of being cited stale.

    python tools/profile_stamp.py            -> prints the hash"""
import glob
import hashlib
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = (["bench.py", "autoposeestimation_amd/engine.py", "autoposeestimation_amd/DenseFusion/lib/network.py", "autoposeestimation_amd/pipeline/utils.py",
          "autoposeestimation_amd/segmentation/utils.py", "autoposeestimation_amd/csrc/Makefile"]
         + sorted(os.path.relpath(p, REPO) for p in glob.glob(os.path.join(REPO, "autoposeestimation_amd", "csrc", "*.hip")))
         + sorted(os.path.relpath(p, REPO) for p in glob.glob(os.path.join(REPO, "autoposeestimation_amd", "csrc", "*.h"))))


def source_hash():
    h = hashlib.sha256()
    for rel in FILES:
        h.update(rel.encode())
        with open(os.path.join(REPO, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_hash())
