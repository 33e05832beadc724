"""Print the per-kernel mean of every counter in a rocprofv3 --pmc counter_collection CSV (development aid)."""
import collections
import csv
import sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-60:]
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for name, cs in acc.items():
    if len(sys.argv) > 2 and sys.argv[2] not in name:
        continue
    print(name)
    for c, v in sorted(cs.items()):
        print("   %-34s n=%3d mean=%.4g" % (c, len(v), sum(v) / len(v)))
