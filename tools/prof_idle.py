"""How long is the GPU idle inside the software-pipelined loop?  From a rocprofv3 --kernel-trace CSV of the default bench command: the window between
the first and the last `ccl_init_kernel` launch of the last N steps (a fixed point of every step), the union of all kernels' busy intervals in it, and
the idle gaps by size.
    python tools/prof_idle.py <kernel_trace.csv> [steps]"""
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ccl = [int(r["Start_Timestamp"]) for r in rows if "ccl_init_kernel" in r["Kernel_Name"]]
t0, t1 = ccl[-1 - n], ccl[-1]
iv = sorted((max(int(r["Start_Timestamp"]), t0), min(int(r["End_Timestamp"]), t1)) for r in rows if int(r["End_Timestamp"]) > t0 and int(r["Start_Timestamp"]) < t1)
busy, cur_s, cur_e, gaps = 0, None, None, []
for s, e in iv:
    if cur_e is None or s > cur_e:
        if cur_e is not None:
            busy += cur_e - cur_s
            gaps.append((s - cur_e, cur_e))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
wall = t1 - t0
print("steps %d  wall %.3f ms/step  busy %.3f ms/step  idle %.3f ms/step (%.2f %%)" % (n, wall / n / 1e6, busy / n / 1e6, (wall - busy) / n / 1e6, 100.0 * (wall - busy) / wall))
gaps.sort(reverse=True)
print("largest idle gaps (us):", " ".join("%.1f" % (g / 1e3) for g, _ in gaps[:12]))
print("gaps > 2 us: %d per step, summing %.3f ms/step" % (sum(1 for g, _ in gaps if g > 2000) / n, sum(g for g, _ in gaps if g > 2000) / n / 1e6))
# what runs ALONE: time during which exactly one kernel is resident, by kernel
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if e > t0 and s < t1:
        ev.append((max(s, t0), 1, r["Kernel_Name"])); ev.append((min(e, t1), -1, r["Kernel_Name"]))
ev.sort(key=lambda x: (x[0], x[1]))
active, last, alone, both = {}, t0, {}, 0
for t, d, name in ev:
    k = sum(active.values())
    if k == 1:
        nm = [a for a, c in active.items() if c][0]
        alone[nm] = alone.get(nm, 0) + t - last
    elif k >= 2:
        both += t - last
    active[name] = active.get(name, 0) + d
    last = t
print("two or more kernels resident: %.3f ms/step" % (both / n / 1e6))
for nm, v in sorted(alone.items(), key=lambda kv: -kv[1])[:12]:
    print("  alone %.3f ms/step  %s" % (v / n / 1e6, nm.replace("(anonymous namespace)::", "").replace("void ", "")[:90]))
