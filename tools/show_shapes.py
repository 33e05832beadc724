import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(sys.argv[1], d["value"], d["ms_per_step"])
for k in d["roofline"]["kernels"]:
    for s in k["shapes"]:
        print("  %-28s %-44s n=%3d  %8.1f us  iso %s" % (k["kernel"][:28], s["shape"], s["launches"], s["avg_launch_us"], s.get("isolated_avg_launch_us")))
