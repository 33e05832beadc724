"""Per-kernel HBM traffic from two rocprofv3 PMC passes of bench.py (FETCH_SIZE and WRITE_SIZE need separate passes: TCC has 4
slots, FETCH_SIZE takes 3, WRITE_SIZE 2 -- MI355X_MICROARCH.md "rocprofv3 PMC slots"):

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline
    python tools/pmc_summary.py gpurun_out/pmc_fetch/*/*_counter_collection.csv gpurun_out/pmc_write/*/*_counter_collection.csv [--shapes launches.json] > profiles/rNN_pmc_traffic.json

--shapes: `bench.py --no-overlap --dump-launches launches.json` writes the (kernel, layer shape) list of one step in launch order; the
LAST step of each pass is then split per shape (the i-th dispatch of a kernel in a step is the i-th entry of that kernel in the list), so
that bench.py compares a kernel's HBM bytes with its algorithmic bytes shape by shape (a kernel such as gemm_s32_kernel<256> serves
1.3 GB segmentation layers and 50 MB pose layers: one average over all of them says nothing).

Units and gfx950 correction (MI355X_MICROARCH.md "HBM"): both counters are in KiB; FETCH_SIZE reports exactly HALF the bytes of
wide coalesced streaming reads (every kernel here loads 16 B per lane) -> doubled; WRITE_SIZE is exact for 16-B stores.
Cross-check on a known byte count: seg_head_kernel reads 64 x 480 x 640 x 64 fp32 = 5.03 GB per launch and FETCH_SIZE x 1024 x 2
gives 5.04 GB."""
import collections
import csv
import json
import sys


def load(path):
    """-> {kernel: [dispatch count, counter sum, [per-dispatch values in dispatch order]]}"""
    agg = collections.defaultdict(lambda: [0, 0.0, []])
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Dispatch_Id"]))
    for r in rows:
        name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0].replace(" ", "")
        agg[name][0] += 1
        agg[name][1] += float(r["Counter_Value"])
        agg[name][2].append(float(r["Counter_Value"]))
    return agg


args = [a for a in sys.argv[1:] if a != "--shapes"]
shape_list = None
if "--shapes" in sys.argv:
    shape_list = json.load(open(sys.argv[sys.argv.index("--shapes") + 1]))
    args.remove(sys.argv[sys.argv.index("--shapes") + 1])
fetch, write = load(args[0]), load(args[1])
per_kernel_shapes = collections.defaultdict(list)          # kernel -> shapes of one step, in launch order
for name, shape in shape_list or []:
    per_kernel_shapes[name.replace(" ", "")].append(shape)
out = {}
for k in sorted(set(fetch) | set(write), key=lambda k: -(fetch.get(k, [0, 0])[1] * 2 + write.get(k, [0, 0])[1])):
    n = max(fetch.get(k, [0, 0])[0], write.get(k, [0, 0])[0])
    fb = fetch.get(k, [0, 0.0])[1] * 1024 * 2
    wb = write.get(k, [0, 0.0])[1] * 1024
    out[k] = {"launches": n, "fetch_bytes_per_launch": round(fb / n), "write_bytes_per_launch": round(wb / n),
              "hbm_bytes_per_launch": round((fb + wb) / n)}
    step = per_kernel_shapes.get(k)
    if step and k in fetch and k in write and len(fetch[k][2]) >= len(step) and len(write[k][2]) >= len(step):
        fl, wl = fetch[k][2][-len(step):], write[k][2][-len(step):]     # the last step of either pass
        rows = collections.defaultdict(lambda: [0, 0.0, 0.0])
        for shape, f, w in zip(step, fl, wl):
            rows[shape][0] += 1
            rows[shape][1] += f * 1024 * 2
            rows[shape][2] += w * 1024
        out[k]["shapes"] = {sh: {"launches": c, "fetch_bytes_per_launch": round(f / c), "write_bytes_per_launch": round(w / c),
                                 "hbm_bytes_per_launch": round((f + w) / c)} for sh, (c, f, w) in rows.items()}
print(json.dumps({"_note": "whole process (set-up + 1 warm-up + 2 timed steps); FETCH_SIZE doubled per the gfx950 correction; `shapes` = the last step's "
                           "dispatches of the kernel split by layer shape (bench.py --dump-launches)", "kernels": out},
                 indent=1))
