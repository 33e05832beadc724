"""Summarise a rocprofv3 --kernel-trace CSV of `bench.py --steps K --warmup W`: per-kernel totals over the whole process
and over the TIMED REGION only (the last K/(K+W) of the pipeline's launches -- set-up launches, e.g. the segmentor
last-layer fit, come first and are dropped), so the average launch duration can be compared with bench.py's live figure.

    python tools/prof_summary.py gpurun_out/prof_bench/*/*_kernel_trace.csv --steps 3 --warmup 1 > profiles/rNN_....json
"""
import argparse
import collections
import csv
import json

ap = argparse.ArgumentParser()
ap.add_argument("trace")
ap.add_argument("--steps", type=int, required=True)
ap.add_argument("--warmup", type=int, required=True)
ap.add_argument("--tail-steps", type=int, default=0,
                help="steps bench.py runs AFTER the timed region (its 3-step single-stream pass for the isolated kernel timings when the "
                     "loop is software-pipelined); they are summarised separately as `isolated_pass`")
a = ap.parse_args()
assert a.warmup >= 1, "needs at least one warm-up step to locate the start of the timed region"
rows = list(csv.DictReader(open(a.trace)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# Step structure: every step labels the connected components of its segmentation exactly once (ccl_init_kernel, right after
# the segmentation CNN and its head) and ends with pose_compose_kernel launches.  The timed region starts after the last
# pose_compose of warm-up step W.
argmax = [i for i, r in enumerate(rows) if "ccl_init_kernel" in r["Kernel_Name"]]
assert len(argmax) == a.steps + a.warmup + a.tail_steps, (len(argmax), a.steps, a.warmup, a.tail_steps)
last_compose = max(i for i, r in enumerate(rows[:argmax[a.warmup]]) if "pose_compose_kernel" in r["Kernel_Name"])
t0 = int(rows[last_compose]["End_Timestamp"])
t1 = 1 << 62
if a.tail_steps:       # the tail starts after a fence: nothing of it begins before the timed region's last pose_compose has ended
    end_compose = max(i for i, r in enumerate(rows[:argmax[a.warmup + a.steps]]) if "pose_compose_kernel" in r["Kernel_Name"])
    t1 = int(rows[end_compose]["End_Timestamp"])
by = collections.defaultdict(list)
timed = collections.defaultdict(list)
tail = collections.defaultdict(list)
for r in rows:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    by[r["Kernel_Name"]].append(d)
    if t0 <= int(r["Start_Timestamp"]) < t1:
        timed[r["Kernel_Name"]].append(d)
    elif int(r["Start_Timestamp"]) >= t1:
        tail[r["Kernel_Name"]].append(d)
out = {}
tot = sum(sum(v) for v in by.values())
ttot = sum(sum(v) for v in timed.values())
for name, d in sorted(by.items(), key=lambda kv: -sum(timed.get(kv[0], [0]))):
    t = timed.get(name, [])
    short = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")[:80]
    out[short] = {"process": {"launches": len(d), "total_ms": round(sum(d) / 1e3, 3), "avg_us": round(sum(d) / len(d), 2),
                              "share": round(sum(d) / tot, 4)},
                  "timed_region": {"launches": len(t), "total_ms": round(sum(t) / 1e3, 3),
                                   "avg_us": round(sum(t) / len(t), 2) if t else None,
                                   "share": round(sum(t) / ttot, 4) if t else 0.0}}
    if a.tail_steps and tail.get(name):
        out[short]["isolated_pass"] = {"launches": len(tail[name]), "avg_us": round(sum(tail[name]) / len(tail[name]), 2)}
print(json.dumps(out, indent=1))
