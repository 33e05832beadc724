"""Ordered launch list of the LAST step of a rocprofv3 --kernel-trace CSV of bench.py (development aid): one line per
launch with its grid, duration and the gap to the previous launch, so every layer of the step can be priced.

    python tools/prof_step_list.py gpurun_out/prof/*/*_kernel_trace.csv > gpurun_out/step_list.txt
"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ccl = [i for i, r in enumerate(rows) if "ccl_init_kernel" in r["Kernel_Name"]]
# a step ends with its last pose_compose launch; the last step starts right after the previous step's last pose_compose
prev_end = max(i for i, r in enumerate(rows[:ccl[-1]]) if "pose_compose_kernel" in r["Kernel_Name"])
step = rows[prev_end + 1:]
last = max(i for i, r in enumerate(step) if "pose_compose_kernel" in r["Kernel_Name"])
step = step[:last + 1]
t_prev = int(rows[prev_end]["End_Timestamp"])
t0 = int(step[0]["Start_Timestamp"])
tot = 0.0
for r in step:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    name = name.split("(")[0][:70]
    grid = r.get("Grid_Size_X", r.get("Grid_Size", "?"))
    wg = r.get("Workgroup_Size_X", r.get("Workgroup_Size", "?"))
    d = (e - s) / 1e3
    tot += d
    print("%9.1f us  +%7.1f gap  t=%8.2f ms  grid %9s wg %4s  %s" % (d, (s - t_prev) / 1e3, (s - t0) / 1e6, grid, wg, name))
    t_prev = e
print("launches %d  kernel time %.2f ms  wall %.2f ms" % (len(step), tot / 1e3, (t_prev - t0) / 1e6))
