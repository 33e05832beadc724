"""Static audit of hipcc's vmcnt waits in one kernel of a -save-temps .s file: walks the instruction stream in text order, keeps the list of
outstanding vector-memory loads (they return in order) with their destination VGPRs, applies every `s_waitcnt vmcnt(N)`, and reports any
instruction that reads or overwrites a VGPR an outstanding load has yet to write.  Branches are ignored (text order), so findings next to a
label need a look at the control flow; none means every use in straight-line code is covered by a sufficient wait.
    python tools/probes/vmcnt_check.py file.s <mangled-kernel-name-substring>"""
import re
import sys

src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and key in l)
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])


def regs(tok):
    """v12 -> {12}; v[4:7] -> {4,5,6,7}"""
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


pending = []      # [(line number, dest regs)]
issues = 0
for ln in range(start + 1, end):
    t = lines[ln].split(";")[0].strip()
    if not t or t.endswith(":") or t.startswith("."):
        continue
    op, _, rest = t.partition(" ")
    toks = [x.strip() for x in re.split(r",\s*(?![^\[]*\])", rest)] if rest else []
    if op == "s_waitcnt":
        m = re.search(r"vmcnt\((\d+)\)", t)
        if m:
            n = int(m.group(1))
            pending = pending[len(pending) - n:] if n < len(pending) else pending
            if n == 0:
                pending = []
        continue
    used = set()
    for x in toks:
        used |= regs(x.split(" ")[0])
    busy = set().union(*[d for _, d in pending]) if pending else set()
    hit = used & busy
    if hit and not op.startswith(("global_load", "buffer_load", "scratch_load", "flat_load")):
        issues += 1
        who = [(l, sorted(d)) for l, d in pending if d & hit]
        print("line %d: `%s` touches v%s while load(s) %s are outstanding (%d pending)" % (ln + 1, t[:90], sorted(hit), who[:2], len(pending)))
    if op.startswith(("global_load", "buffer_load", "scratch_load", "flat_load")) and "lds" not in t:
        pending.append((ln + 1, regs(toks[0]) if toks else set()))
    elif op.startswith(("global_store", "buffer_store", "scratch_store", "flat_store", "global_atomic", "buffer_atomic")):
        pending.append((ln + 1, set()))      # stores count in vmcnt on gfx9-family parts
print("kernel lines %d..%d: %d finding(s)" % (start + 1, end + 1, issues))
