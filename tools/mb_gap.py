"""Where does the ~11 us between two back-to-back halo_s32 launches in a kernel trace come from?  (ablation build; run under
rocprofv3 --kernel-trace, then `python tools/mb_gap.py --reduce <kernel_trace.csv>`).  Groups of 8 back-to-back launches of the
512 -> 512 d 4 layer under debug bits, a torch fill between the groups as the marker."""
import csv, ctypes, os, sys
if len(sys.argv) > 2 and sys.argv[1] == "--reduce":
    rows = sorted(csv.DictReader(open(sys.argv[2])), key=lambda r: int(r["Start_Timestamp"]))
    groups, cur = [], []
    for r in rows:
        if "halo_s32" in r["Kernel_Name"] or "touch" in r["Kernel_Name"]:
            cur.append(r)
        elif cur:
            groups.append(cur); cur = []
    if cur: groups.append(cur)
    for g in groups:
        gaps = [(int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e3 for a, b in zip(g, g[1:])]
        durs = [(int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e3 for a in g]
        print("%2d launches  grid %8s  avg %8.1f us  gaps %s" % (len(g), g[0].get("Grid_Size_X", g[0].get("Grid_Size")), sum(durs) / len(durs), " ".join("%.1f" % x for x in gaps)))
    sys.exit(0)
os.environ.setdefault("APE_HIP_LIB", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "autoposeestimation_amd", "libape_hip_abl.so"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from autoposeestimation_amd import _lib, engine as E
b, h, w, cin, cout, dil = 64, 60, 80, 512, 512, 4
conv = E.Conv(torch.randn(cout, cin, 3, 3) / (3 * cin ** 0.5), torch.randn(cout), pad=dil, dil=dil, act=E.ACT_RELU, device="cuda", precision="bf16x3")
xs = E.S32.from_f32(torch.relu(torch.randn(b, h, w, cin, device="cuda")))
out = E.S32(torch.empty(b, h, w, cout, device="cuda"))
p = E.ConvParams(B=b, H=h, W=w, Cin=cin, ldx=cin, xoff=0, Ho=h, Wo=w, Cout=cout, ldy=cout, yoff=0, KH=3, KW=3, stride=1, pad=dil, dil=dil, act=E.ACT_RELU,
                 alpha=0.0, bias_bstride=0, ldr=0, roff=0, ups=0)
def run():
    _lib.check(_lib.lib().ape_conv3x3_halo_s32(_lib.dptr(xs.t, torch.float32), _lib.dptr(conv.s32k()), _lib.dptr(conv.bias), None, 0, _lib.dptr(out.t, torch.float32),
                                               E.FMT_S32, ctypes.byref(p), None), "s32")
mark = torch.zeros(1024, device="cuda")
for bits in (0, 4, 2, 512, 1024 | 2048 | 512, 1024 | 2048 | 512 | 4, 4096, 0):
    _lib.lib().ape_conv3x3_halo_s32_debug(bits)
    print("bits", bits, flush=True)
    for _ in range(8): run()
    mark.fill_(1.0)
    torch.cuda.synchronize()
_lib.lib().ape_conv3x3_halo_s32_debug(0)
