"""Per-kernel HBM traffic from two rocprofv3 PMC passes of bench.py (FETCH_SIZE and WRITE_SIZE need separate passes: TCC has 4
slots, FETCH_SIZE takes 3, WRITE_SIZE 2 -- MI355X_MICROARCH.md "rocprofv3 PMC slots"):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    python tools/pmc_summary.py gpurun_out/pmc_fetch/*/*_counter_collection.csv gpurun_out/pmc_write/*/*_counter_collection.csv > profiles/rNN_pmc_traffic.json

Units and gfx950 correction (MI355X_MICROARCH.md "HBM"): both counters are in KiB; FETCH_SIZE reports exactly HALF the bytes of
wide coalesced streaming reads (every kernel here loads 16 B per lane) -> doubled; WRITE_SIZE is exact for 16-B stores.
Cross-check on a known byte count: seg_head_kernel reads 64 x 480 x 640 x 64 fp32 = 5.03 GB per launch and FETCH_SIZE x 1024 x 2
gives 5.04 GB."""
import collections
import csv
import json
import sys


def load(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].replace(" ", "")
        agg[name][0] += 1
        agg[name][1] += float(r["Counter_Value"])
    return agg


fetch, write = load(sys.argv[1]), load(sys.argv[2])
out = {}
for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, [0, 0])[1] * 2 + write.get(k, [0, 0])[1])):
    n = max(fetch.get(k, [0, 0])[0], write.get(k, [0, 0])[0])
    fb = fetch.get(k, [0, 0.0])[1] * 1024 * 2
    wb = write.get(k, [0, 0.0])[1] * 1024
    out[k] = {"launches": n, "fetch_bytes_per_launch": round(fb / n), "write_bytes_per_launch": round(wb / n),
              "hbm_bytes_per_launch": round((fb + wb) / n)}
print(json.dumps({"_note": "whole process (set-up + 1 warm-up + 2 timed steps); FETCH_SIZE doubled per the gfx950 correction", "kernels": out},
                 indent=1))
