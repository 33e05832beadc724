"""The clock the chip holds under halo_s32's two schedules, on one box in one process (DVFS give-back, MI355X_MICROARCH.md): 512 -> 512 d 4 at
bench size, N launches of the ping-pong kernel and N of the lockstep kernel (debug bit 4096), interleaved in blocks of 5 behind 10 warm-up launches.

    rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d <dir> -- python3 tools/pmc_halo_clock.py
    python tools/pmc_halo_clock.py --reduce <dir>/*/*_counter_collection.csv <dir>/*/*_kernel_trace.csv
-> per kernel: mean duration, cycles per XCD (GRBM_GUI_ACTIVE / 8), clock = cycles / duration, matrix pipe busy."""
import collections, csv, os, sys

if len(sys.argv) > 3 and sys.argv[1] == "--reduce":
    ctr = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(sys.argv[2])):
        if "halo_s32_kernel" in r["Kernel_Name"]:
            ctr[r["Dispatch_Id"]][r["Counter_Name"]] += float(r["Counter_Value"])
            ctr[r["Dispatch_Id"]]["name"] = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    dur = {}
    for r in csv.DictReader(open(sys.argv[3])):
        if "halo_s32_kernel" in r["Kernel_Name"]:
            dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    by = collections.defaultdict(list)
    ids = sorted(ctr, key=int)[10:]                 # (drop the warm-up launches)
    for d in ids:
        if d in dur:
            by[ctr[d]["name"]].append((dur[d], ctr[d]["GRBM_GUI_ACTIVE"] / 8, ctr[d].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / 1024))
    fetch = [c.get("FETCH_SIZE") for c in (ctr[d] for d in ids) if "FETCH_SIZE" in c]
    if fetch:
        print("FETCH_SIZE: %.3f GB fetched beyond L2 per launch (%d launches; KiB x 2, the gfx950 correction of tools/pmc_summary.py)" % (sum(fetch) / len(fetch) * 1024 * 2 / 1e9, len(fetch)))
    for name, v in by.items():
        n = len(v)
        us, cyc, mf = sum(a for a, _, _ in v) / n, sum(b for _, b, _ in v) / n, sum(c for _, _, c in v) / n
        if cyc:
            print("%-32s launches %2d  %8.1f us  %.3f M cycles per XCD  ->  %.2f GHz held;  matrix pipe busy %.3f (%.3f M cycles per SIMD)"
                  % (name, n, us, cyc / 1e6, cyc / us / 1e3, mf / cyc, mf / 1e6))
    sys.exit(0)

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from autoposeestimation_amd import _lib, engine as E  # noqa: E402
torch.manual_seed(0)
x = E.S32.from_f32(torch.relu(torch.randn(64, 60, 80, 512, device="cuda")))
conv = E.Conv(torch.randn(512, 512, 3, 3) / 68, torch.randn(512), 1, 4, 4, E.ACT_RELU, device="cuda", precision="bf16x3")
out = torch.empty(64, 60, 80, 512, device="cuda")
for _ in range(10):
    conv(x, out=out, out_fmt=E.FMT_S32)
# ARM_BITS=<bits>: ONE arm (20 launches with these debug bits, e.g. 8 = one channel tile per XCD) -- for counters whose rows carry no arm
arms = (int(os.environ["ARM_BITS"]),) if "ARM_BITS" in os.environ else (0, 4096)
for rnd in range(4):
    for bits in arms:
        _lib.lib().ape_conv3x3_halo_s32_debug(bits)
        for _ in range(5):
            conv(x, out=out, out_fmt=E.FMT_S32)
_lib.lib().ape_conv3x3_halo_s32_debug(0)
torch.cuda.synchronize()
