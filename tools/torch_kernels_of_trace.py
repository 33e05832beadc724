"""names of the kernels of a rocprofv3 kernel trace that the library does not own (at::native, rocclr blits, RCCL): one per line, for
tools/scan_pk_src1.py --names-file.   python tools/torch_kernels_of_trace.py <kernel_trace.csv> [> names.txt]"""
import csv, sys
names = {}
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        n = row.get("Kernel_Name") or row.get("kernel_name") or ""
        if n.startswith(("void at::", "at::", "__amd_rocclr", "void c10", "ncclDevKernel", "void rccl")) or "at::native" in n:
            names[n] = names.get(n, 0) + 1
for n, c in sorted(names.items(), key=lambda kv: -kv[1]):
    print(n)
