"""Static audit of the wait counters in a gfx950 kernel's ISA (`hipcc -save-temps` .s): the standing check behind the kernels whose hazards the
compiler cannot see (inline-asm `ds_read_b128` whose results hipcc believes are ready at once, LDS-DMA `buffer_load ... lds` kept in flight
across barriers behind hand-counted `s_waitcnt vmcnt(N)`).  Runs on the CPU (hipcc cross-compiles), used by tests/test_isa_waits.py.  Beside the wait counters: wait states software owes around matrix
instructions when one side of the pair is inside an asm statement (mfma_asm_hazards), VALU-written SGPRs in front of asm vector-memory
instructions (sgpr_vmem_hazards) and the packed-f32 operand form that proved faulty on gfx950 (pk_src1_swizzles).

Counter model (MI355X_MICROARCH.md "s_waitcnt"): vector-memory operations -- loads, stores, atomics, LDS-DMA -- retire in issue order on
vmcnt; LDS operations retire in order on lgkmcnt; scalar-memory loads share lgkmcnt and may return out of order (with one pending only
lgkmcnt(0) proves anything).  `s_waitcnt vmcnt(N)` / `lgkmcnt(N)` returns when at most N operations of that kind are outstanding.
What is checked, and how: see audit().  Spills come from the kernel's metadata (`.vgpr_spill_count`, `.private_segment_fixed_size`,
`.sgpr_spill_count`).  The kernels are compiled with -DAPE_NO_ABLATIONS: their timing-only debug switches ("no barrier", "no MFMAs" ...)
are wrong-result paths that the product never takes.
"""
import os
import re
import subprocess
import sys

HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "autoposeestimation_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize",          # = csrc/Makefile ...
         "-DAPE_NO_ABLATIONS"]     # ... without the kernels' timing-only ablation switches (wrong-result debug paths, e.g. "no barrier")


def compile_to_asm(hip_file, out_dir, defs=()):
    """-> path of the gfx950 .s of `hip_file` (cached under out_dir while it is newer than the source and its headers); `defs`: extra -D
    switches (give such a build its own out_dir)"""
    os.makedirs(out_dir, exist_ok=True)
    stem = os.path.splitext(os.path.basename(hip_file))[0]
    asm = os.path.join(out_dir, stem + "-hip-amdgcn-amd-amdhsa-gfx950.s")
    deps = [hip_file] + [os.path.join(CSRC, h) for h in os.listdir(CSRC) if h.endswith(".h")] + [os.path.join(REPO, "include", "ape_hip.h")]
    if os.path.exists(asm) and all(os.path.getmtime(asm) >= os.path.getmtime(d) for d in deps):
        return asm
    subprocess.check_call([HIPCC] + FLAGS + list(defs) + ["-save-temps", "-c", hip_file, "-I" + CSRC, "-o", os.path.join(out_dir, stem + ".o")], cwd=out_dir,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return asm


def kernel_symbols(asm_path):
    """{demangled-ish key: mangled symbol} of the kernels defined in the file (from their `.amdhsa_kernel` directives)"""
    syms = re.findall(r"^\s*\.amdhsa_kernel\s+(\S+)", open(asm_path).read(), re.M)
    return syms


def kernel_metadata(asm_path, symbol):
    text = open(asm_path).read()
    text = text[text.rfind("amdhsa.kernels:"):]
    blocks = re.split(r"\n  - (?=\.)", text)            # one YAML list item per kernel
    blk = next((b for b in blocks if re.search(r"\.name:\s+" + re.escape(symbol) + r"\s", b)), "")
    out = {}
    for key in (".vgpr_count", ".vgpr_spill_count", ".sgpr_spill_count", ".private_segment_fixed_size", ".sgpr_count", ".group_segment_fixed_size"):
        mm = re.search(re.escape(key) + r":\s+(\d+)", blk)
        out[key[1:]] = int(mm.group(1)) if mm else None
    return out


_VREG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def _vregs(text):
    regs = set()
    for m in _VREG.finditer(text):
        if m.group(1) is not None:
            regs.add(int(m.group(1)))
        else:
            regs.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return regs


class Insn:
    __slots__ = ("line", "op", "text", "regs", "dest", "kind", "target", "vm_wait", "lgkm_wait")

    def __init__(self, line, text):
        self.line, self.text = line, text
        self.op, _, rest = text.partition(" ")
        self.regs = _vregs(rest)
        self.dest, self.kind, self.target, self.vm_wait, self.lgkm_wait = set(), None, None, None, None
        op = self.op
        first = rest.split(",")[0].strip() if rest else ""
        if op.startswith(("global_load", "buffer_load", "scratch_load", "flat_load")):
            if re.search(r"\blds\b", rest):
                self.kind = "dma"                      # LDS-DMA: no VGPR destination, counts on vmcnt
            else:
                self.kind, self.dest = "vload", _vregs(first)
        elif op.startswith(("global_store", "buffer_store", "scratch_store", "flat_store", "global_atomic", "buffer_atomic", "flat_atomic")):
            self.kind = "vstore"
        elif op.startswith("ds_"):
            if op.startswith(("ds_read", "ds_bpermute", "ds_permute", "ds_swizzle", "ds_consume", "ds_append")) or "rtn" in op:
                self.kind, self.dest = "dsread", _vregs(first)
            else:
                self.kind = "dswrite"
        elif op.startswith(("s_load", "s_buffer_load", "s_memtime", "s_memrealtime", "s_dcache")):
            self.kind = "smem"
        elif op == "s_waitcnt":
            m = re.search(r"vmcnt\((\d+)\)", rest)
            self.vm_wait = int(m.group(1)) if m else None
            m = re.search(r"lgkmcnt\((\d+)\)", rest)
            self.lgkm_wait = int(m.group(1)) if m else None
            if m is None and self.vm_wait is None and re.fullmatch(r"\s*(0x[0-9a-fA-F]+|\d+)\s*", rest or ""):
                imm = int(rest.strip(), 0)             # raw immediate (gfx9 layout): vmcnt = [3:0] | [15:14] << 4, lgkmcnt = [11:8]
                self.vm_wait = (imm & 15) | (((imm >> 14) & 3) << 4)
                self.lgkm_wait = (imm >> 8) & 15
                self.vm_wait = None if self.vm_wait == 63 else self.vm_wait
                self.lgkm_wait = None if self.lgkm_wait == 15 else self.lgkm_wait
            self.kind = "wait"
        elif op == "s_barrier":
            self.kind = "barrier"
        elif op in ("s_branch",) or op.startswith("s_cbranch"):
            self.kind = "branch"
            self.target = rest.strip().split()[-1] if rest else None
        elif op in ("s_endpgm", "s_setpc_b64", "s_trap"):
            self.kind = "end"


def parse_kernel(asm_path, symbol):
    lines = open(asm_path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(symbol + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end") or lines[i].strip().startswith(".section"))
    blocks, labels, cur = [], {}, []

    def close():
        nonlocal cur
        if cur:
            blocks.append(cur)
        cur = []

    for ln in range(start + 1, end):
        t = lines[ln].split(";")[0].strip()
        if not t or t.startswith("."):
            m = re.match(r"^(\.L\w+):", t)
            if m:
                close()
                labels[m.group(1)] = len(blocks)
            continue
        m = re.match(r"^(\.?\w+):$", t)
        if m:
            close()
            labels[m.group(1)] = len(blocks)
            continue
        ins = Insn(ln + 1, t)
        cur.append(ins)
        if ins.kind in ("branch", "end"):
            close()
    close()
    succ = []
    for bi, blk in enumerate(blocks):
        last = blk[-1]
        s = []
        if last.kind == "branch":
            if last.target in labels and labels[last.target] < len(blocks):
                s.append(labels[last.target])
            if last.op != "s_branch" and bi + 1 < len(blocks):
                s.append(bi + 1)
        elif last.kind != "end" and bi + 1 < len(blocks):
            s.append(bi + 1)
        succ.append(s)
    return blocks, succ


def _fixed_point(blocks, succ, init, transfer, join):
    """forward dataflow to a fixed point: state per block entry; transfer(state, insn) -> state; join(a, b) -> merged (None = unreached)"""
    ins_state = [None] * len(blocks)
    ins_state[0] = init
    work = [0]
    while work:
        bi = work.pop()
        st = ins_state[bi]
        for ins in blocks[bi]:
            st = transfer(st, ins)
        for sj in succ[bi]:
            merged = st if ins_state[sj] is None else join(ins_state[sj], st)
            if merged != ins_state[sj]:
                ins_state[sj] = merged
                work.append(sj)
    return ins_state


def audit(asm_path, symbol, max_findings=20, dma_barrier_slack=0):
    """-> dict(findings=[...], vmcnt_literals=[...], n_insns, n_mfma, n_dsread, n_dma, meta={...})

    Three forward dataflow passes over the kernel's control-flow graph (see the module docstring for the counter model):
      1. LDS side, path-exact: a state is the SET of possible lgkmcnt queues (one per path history).  FINDING "ds_read result used before
         its wait": an instruction touches a VGPR that a still-queued ds_read has yet to write.  This is the pass that covers the
         inline-asm fragment reads, whose completion the compiler does not track.
      2. vector-memory side, register loads: per pending load the MINIMUM number of younger vector-memory operations over all paths
         (`s_waitcnt vmcnt(N)` proves a load back iff at least N operations were issued after it on every path).  FINDING "load result
         used before its wait".
      3. LDS-DMA discipline (structural, text order): between a `buffer_load ... lds` and the next `s_barrier` there is an `s_waitcnt` with a
         vmcnt field (FINDING "barrier behind an un-waited LDS-DMA" otherwise: that barrier could not make the piece visible to the other
         waves); `vmcnt_literals` lists the counts the kernel waits with (halo_s32's closing wait is a switch over vmcnt(0..8) on the
         number of pieces its tap issued: all nine must be there).  Whether a COUNT is the right one depends on run-time trip counts
         (pieces per tap, whether a further k-tile follows) and is what the bit-exact GPU tests cover, not this walk.
    Pass 2 merges paths pessimistically (a load that is followed by fewer operations on ONE path is treated as if on all), so its findings
    are candidates to look at, reported under "candidates", not asserted."""
    blocks, succ = parse_kernel(asm_path, symbol)
    dest_of = {ins.line: frozenset(ins.dest) for blk in blocks for ins in blk}
    findings, candidates, seen = [], [], set()

    def note(kind, ins, detail):
        if (kind, ins.line) not in seen:
            seen.add((kind, ins.line))
            dst = candidates if kind == "load result used before its wait" else findings
            if len(dst) < max_findings:
                dst.append("%s at line %d `%s`: %s" % (kind, ins.line, ins.text[:80], detail))

    # ---- pass 1: lgkmcnt, path-exact sets of queues; queue entry = (line, kind) -------------------------------------------------
    def t1(states, ins):
        out = set()
        for lg in states:
            if ins.kind == "wait":
                if ins.lgkm_wait is not None:
                    if ins.lgkm_wait == 0:
                        lg = ()
                    elif not any(k == "smem" for _, k in lg):
                        lg = lg[len(lg) - ins.lgkm_wait:] if ins.lgkm_wait < len(lg) else lg
                out.add(lg)
                continue
            if ins.regs:
                busy = [l for l, k in lg if k == "dsread" and l != ins.line and dest_of[l] & ins.regs]
                if busy:
                    note("ds_read result used before its wait", ins, "VGPRs of the ds_read(s) at line(s) %s" % busy[:4])
            if ins.kind in ("dsread", "dswrite", "smem"):
                lg = lg + ((ins.line, ins.kind),)
                if len(lg) > 64:
                    lg = lg[-64:]
            out.add(lg)
        if len(out) > 64:       # far beyond anything these kernels produce; keep the longest queues (the most pessimistic)
            out = set(sorted(out, key=len)[-64:])
        return frozenset(out)

    _fixed_point(blocks, succ, frozenset({()}), t1, lambda a, b: a | b)

    # ---- pass 2: vmcnt, pending register loads with the minimum number of younger operations ---------------------------------------
    def t2(st, ins):
        st = dict(st)
        if ins.kind == "wait":
            if ins.vm_wait is not None:
                st = {l: y for l, y in st.items() if y < ins.vm_wait}
            return tuple(sorted(st.items()))
        if ins.regs:
            busy = [l for l in st if l != ins.line and dest_of[l] & ins.regs]
            if busy:
                note("load result used before its wait", ins, "VGPRs of the load(s) at line(s) %s" % busy[:4])
        if ins.kind in ("vload", "vstore", "dma"):
            st = {l: y + 1 for l, y in st.items()}
            if ins.kind == "vload":
                st[ins.line] = 0
        return tuple(sorted(st.items()))

    def j2(a, b):
        da, db = dict(a), dict(b)
        return tuple(sorted({l: min(da.get(l, 1 << 30), db.get(l, 1 << 30)) for l in set(da) | set(db)}.items()))

    _fixed_point(blocks, succ, (), t2, j2)

    # ---- pass 3: LDS-DMA discipline, in TEXT order (the closing wait of a halo_s32 tap is a switch over vmcnt(0..8): the structurizer lowers it to
    # predicated "Flow" blocks through which a path-insensitive walk finds routes that skip every case) -----------------------------------------
    # (blocks in REVERSE POST-ORDER of the control-flow graph, not in text order: hipcc places cold blocks -- the multi-trip loops of a tap's
    # DMA rows -- at the end of the function, where text order would put their pieces in front of whatever barrier happens to follow)
    order, seen_b, stack = [], set(), [(0, iter(succ[0]))] if blocks else []
    seen_b.add(0)
    while stack:
        bi, it = stack[-1]
        nxt = next((sj for sj in it if sj not in seen_b), None)
        if nxt is None:
            order.append(bi)
            stack.pop()
        else:
            seen_b.add(nxt)
            stack.append((nxt, iter(sorted(succ[nxt]))))
    order.reverse()
    flat = [ins for bi in order for ins in blocks[bi]]
    # dma_barrier_slack: how many barriers a piece may cross before its issuing wave's vmcnt wait.  0 for the lockstep kernels (wait, then the
    # barrier that publishes).  1 for halo_s32's ping-pong form: waves 0-3 issue their pieces at the END of an interval, in front of its
    # barrier, and retire them with vmcnt(0) in their next load segment, in front of the NEXT barrier (waves 4-7 issue behind a barrier and
    # wait in front of the next one).
    dma_since_wait = bars_since_dma = 0
    for ins in flat:
        if ins.kind == "wait" and ins.vm_wait is not None:
            dma_since_wait = bars_since_dma = 0
        elif ins.kind == "dma":
            dma_since_wait += 1
        elif ins.kind == "barrier" and dma_since_wait:
            bars_since_dma += 1
            if bars_since_dma > dma_barrier_slack:
                note("barrier behind an un-waited LDS-DMA", ins, "%d LDS-DMA instruction(s) and no s_waitcnt vmcnt between them and this barrier (%d barriers crossed)"
                     % (dma_since_wait, bars_since_dma))
    vm_literals = sorted({ins.vm_wait for ins in flat if ins.kind == "wait" and ins.vm_wait is not None})
    kinds = [ins.kind for blk in blocks for ins in blk]
    return {"findings": findings, "candidates": candidates, "vmcnt_literals": vm_literals, "n_insns": len(kinds), "n_dsread": kinds.count("dsread"),
            "n_dma": kinds.count("dma"), "n_mfma": sum(1 for blk in blocks for ins in blk if ins.op.startswith("v_mfma")),
            "meta": kernel_metadata(asm_path, symbol)}


def sgpr_vmem_hazards(asm_path, symbol, wait_states=5):
    """Inline-asm vector-memory instructions (hipcc's hazard recognizer does not look inside asm statements): an LDS-DMA / buffer load that
    reads an SGPR -- its descriptor quad or its scalar offset -- written by a VALU instruction (`v_readlane` / `v_readfirstlane`: how hipcc
    reloads SPILLED SGPRs, wherever the next use is) fewer than `wait_states` wait states earlier reads a stale value.  Linear walk back
    from every `buffer_load ... offen` INSIDE an asm statement (hipcc pads its own); a label inside the window is reported as "unknown", never
    taken for clean: a predecessor block may end in the reload.  -> [(line, instruction, writer)]"""
    text = open(asm_path).read()
    start = text.index("\n" + symbol + ":")
    body = text[start:text.index("s_endpgm", start)].splitlines()
    found, in_asm = [], False
    for i, line in enumerate(body):
        if ";;#ASMSTART" in line:
            in_asm = True
        elif ";;#ASMEND" in line:
            in_asm = False
        t = line.strip()
        m = re.match(r"buffer_load_dword\w* (?:v\[?[\d:]+\]?, )?v\d+, s\[(\d+):(\d+)\], (s\d+|\d+|0)", t)
        if not m or "offen" not in t or not in_asm:
            continue
        reads = set(range(int(m.group(1)), int(m.group(2)) + 1))
        if m.group(3).startswith("s"):
            reads.add(int(m.group(3)[1:]))
        ws, j = 0, i - 1
        while j >= 0 and ws < wait_states:
            u = body[j].strip()
            j -= 1
            if not u or u.startswith(";") or u.startswith("."):
                if u.startswith(".LBB"):          # a predecessor block may end in the reload: the pair is unknown, not clean
                    found.append((i + 1, t, "label %s inside the %d-state window: unknown" % (u.split(":")[0], wait_states)))
                    break
                continue
            w = re.match(r"v_read(?:first)?lane_b32 s(\d+)", u)
            if w and int(w.group(1)) in reads:
                found.append((i + 1, t, u))
                break
            nop = re.match(r"s_nop (\d+)", u)
            ws += int(nop.group(1)) + 1 if nop else 1
    return found


# ---- wait states that software owes around matrix instructions, where ONE side of the pair sits inside an asm statement ------------------
# hipcc's hazard recognizer pads these pairs for the instructions it emits itself and looks at an INLINEASM statement neither as a vector
# instruction nor as a matrix instruction (cdna_hip_programming.md 5.7 item 2): a pair with its producer or its consumer inside
# `;;#ASMSTART ... ;;#ASMEND` gets nothing.  Wait states are counted like the recognizer counts them: one per instruction, `s_nop N` = N + 1.
_MFMA_PASSES = (("v_mfma_f32_32x32x16", 8), ("v_mfma_f32_16x16x32", 4), ("v_mfma_f32_32x32x2_f32", 16), ("v_mfma_f32_16x16x4_f32", 8),
                ("v_mfma_f32_32x32x8", 8), ("v_mfma_f32_16x16x16", 4), ("v_mfma_f32_4x4", 2), ("v_mfma_scale_f32_16x16x128", 8),
                ("v_mfma_scale_f32_32x32x64", 16), ("v_mfma_f32_16x16x128", 8), ("v_mfma_f32_32x32x64", 16), ("v_mfma_f64", 16))


def _mfma_passes(op):
    for prefix, n in _MFMA_PASSES:
        if op.startswith(prefix):
            return n
    return 16            # an unknown shape: the longest


def _operands(text):
    """-> (op, [operand strings]) of one instruction line"""
    op, _, rest = text.partition(" ")
    return op, [o.strip() for o in rest.split(",")] if rest else []


def _is_valu(op):
    return op.startswith("v_") and not op.startswith(("v_mfma", "v_smfma", "v_nop"))


def _parse_cfg_with_asm(asm_path, symbol):
    """-> (blocks, preds): blocks = lists of (line, text, in_asm); preds[b] = predecessor block indices (fall-through and branch targets)"""
    lines = open(asm_path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(symbol + ":"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, labels, cur, in_asm = [], {}, [], False
    for ln in range(start + 1, end):
        raw = lines[ln]
        if ";;#ASMSTART" in raw:
            in_asm = True
            continue
        if ";;#ASMEND" in raw:
            in_asm = False
            continue
        t = raw.split(";")[0].strip()
        if not t:
            continue
        m = re.match(r"^(\.?\w+):$", t)
        if m:
            if cur:
                blocks.append(cur)
                cur = []
            labels[m.group(1)] = len(blocks)
            continue
        if t.startswith("."):
            continue
        cur.append((ln + 1, t, in_asm))
        op = t.split(" ")[0]
        if op == "s_branch" or op.startswith("s_cbranch") or op in ("s_endpgm", "s_setpc_b64", "s_trap"):
            blocks.append(cur)
            cur = []
    if cur:
        blocks.append(cur)
    preds = [[] for _ in blocks]
    for bi, blk in enumerate(blocks):
        op, _, rest = blk[-1][1].partition(" ")
        if op == "s_branch" or op.startswith("s_cbranch"):
            tgt = labels.get(rest.strip().split()[-1])
            if tgt is not None and tgt < len(blocks):
                preds[tgt].append(bi)
            if op != "s_branch" and bi + 1 < len(blocks):
                preds[bi + 1].append(bi)
        elif op not in ("s_endpgm", "s_setpc_b64", "s_trap") and bi + 1 < len(blocks):
            preds[bi + 1].append(bi)
    return blocks, preds


def _walk_back(blocks, preds, bi, ii, budget):
    """every instruction that can execute fewer than `budget` wait states before blocks[bi][ii], over all paths of the control-flow graph:
    yields (wait states in between, line, text, in_asm)"""
    stack, best = [(bi, ii - 1, 0)], {}
    while stack:
        b, i, ws = stack.pop()
        while i >= 0 and ws < budget:
            ln, t, ia = blocks[b][i]
            yield ws, ln, t, ia
            nop = re.match(r"s_nop (\d+)", t)
            ws += int(nop.group(1)) + 1 if nop else 1
            i -= 1
        if i < 0 and ws < budget:
            for pb in preds[b]:
                if best.get(pb, 1 << 30) > ws:          # reach a predecessor again only along a path with fewer wait states
                    best[pb] = ws
                    stack.append((pb, len(blocks[pb]) - 1, ws))


def mfma_asm_hazards(asm_path, symbol, slack=4):
    """Walk backwards from every candidate consumer over ALL paths of the control-flow graph (loop back-edges included).
    Pairs looked for, `passes` = the matrix instruction's pass count (4 cycles each):
      RAW/WAW  matrix instruction writes VGPRs -> an instruction INSIDE an asm statement reads or writes them fewer than passes + `slack`
               wait states later (the recognizer's own figures are passes + 2 / + 3, + 1 on gfx950) -- unless that instruction is itself a
               matrix instruction that takes them whole as its C operand and destination (the accumulate chain: 0 states);
      WAR      matrix instruction reads VGPRs as SrcC (other than its destination) -> a vector instruction inside an asm statement writes them
               fewer than passes + `slack` states later;
      VALU->MFMA  a vector instruction inside an asm statement writes VGPRs -> a matrix instruction (anywhere) reads them as A, B or C fewer
               than 2 wait states later.
    -> [(line, text, partner line, partner text, kind, states seen, states owed)]"""
    blocks, preds = _parse_cfg_with_asm(asm_path, symbol)
    found, seen = [], set()

    def note(*rec):
        if (rec[0], rec[2], rec[4]) not in seen:
            seen.add((rec[0], rec[2], rec[4]))
            found.append(rec)

    for bi, blk in enumerate(blocks):
        for ii, (ln, t, ia) in enumerate(blk):
            op, ops = _operands(t)
            is_mfma = op.startswith(("v_mfma", "v_smfma"))
            if ia and (op.startswith("v_") or op.startswith(("ds_", "buffer_", "global_"))) and not op.startswith("v_nop"):
                # a vector instruction touches its destination at once (WAW) and reads its sources; a memory instruction READS its address / data
                # operands now and writes its destination only when the data returns, tens of cycles later (not a wait-state hazard)
                loads = op.startswith(("ds_read", "buffer_load", "global_load")) and not re.search(r"\blds\b", t)
                touched = _vregs(" ".join(ops[1:] if loads else ops))
                writes = _vregs(ops[0]) if (_is_valu(op) and ops) else set()
                # depth-first over the predecessors with the registers still "owned" by an older producer: a compiler instruction in between
                # that redefines a register ends the search for it on that path (that pair -- MFMA write, then the compiler's own write -- is
                # hipcc's to pad, and the asm statement then reads hipcc's value)
                stack, best = [(bi, ii - 1, 0, frozenset(touched))], {}
                while stack:
                    b, i, ws, live = stack.pop()
                    while i >= 0 and ws < 16 + slack and live:
                        pl, pt, pia = blocks[b][i]
                        pop, pops = _operands(pt)
                        if pop.startswith(("v_mfma", "v_smfma")):
                            owed = _mfma_passes(pop) + slack
                            pdst = _vregs(pops[0])
                            psrcc = _vregs(pops[3]) if len(pops) > 3 else set()
                            chain = is_mfma and len(ops) > 3 and _vregs(ops[0]) == pdst and _vregs(ops[3]) == pdst
                            if ws < owed and not chain:
                                if pdst & live:
                                    note(ln, t, pl, pt, "MFMA write -> asm access", ws, owed)
                                elif (psrcc - pdst) & writes & live:
                                    note(ln, t, pl, pt, "MFMA SrcC read -> asm VALU write", ws, owed)
                            live = live - pdst
                        elif not pia and pops and (pop.startswith("v_") or pop.startswith(("ds_read", "buffer_load", "global_load"))) \
                                and not re.search(r"\blds\b", pt):
                            live = live - _vregs(pops[0])
                        nop = re.match(r"s_nop (\d+)", pt)
                        ws += int(nop.group(1)) + 1 if nop else 1
                        i -= 1
                    if i < 0 and ws < 16 + slack and live:
                        for pb in preds[b]:
                            key = (pb, live)
                            if best.get(key, 1 << 30) > ws:
                                best[key] = ws
                                stack.append((pb, len(blocks[pb]) - 1, ws, live))
            if is_mfma:
                reads = _vregs(" ".join(ops[1:]))
                for ws, pl, pt, pia in _walk_back(blocks, preds, bi, ii, 2):
                    pop, pops = _operands(pt)
                    if pia and _is_valu(pop) and pops and _vregs(pops[0]) & reads:
                        note(ln, t, pl, pt, "asm VALU write -> MFMA read", ws, 2)
    return found



def pk_src1_swizzles(asm_path, symbol, asm_only=True):
    """Packed-f32 instructions (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32) whose SECOND or THIRD source is a VGPR pair read through an
    op_sel / op_sel_hi swizzle (anything but the natural "low result from the low dword, high result from the high dword").  On gfx950 that
    form computed the low half of lanes 48-63 from a wrong dword under back-to-back issue (csrc/upconv_fused.hip, note at APE_NO_ASM_MATH;
    reproduced in every launch of tools/stress_upfuse.py's positive control); with the swizzled pair as src0 -- the slot hipcc uses for its own
    broadcasts -- the same instruction is clean.  asm_only: instructions inside asm statements only; asm_only=False also reports hipcc's own
    (its SLP vectoriser forms `v_pk_mul_f32 d, a, b op_sel:[0,1] op_sel_hi:[1,0]` and the like: pose_select_kernel's rotation was built from
    them and gave wrong values in lanes 48-63 beside a busy second stream -- tools/stress_pose_select.py -- which is why the library is built
    with -fno-slp-vectorize and every kernel is scanned).
    -> [(line, text)]"""
    blocks, _ = _parse_cfg_with_asm(asm_path, symbol)
    found = []
    for blk in blocks:
        for ln, t, ia in blk:
            op, ops = _operands(t)
            if not op.startswith(("v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32")) or (asm_only and not ia):
                continue
            nsrc = 3 if op.startswith("v_pk_fma") else 2
            tail = t
            sel = re.search(r"op_sel:\[([01,]+)\]", tail)
            sel_hi = re.search(r"op_sel_hi:\[([01,]+)\]", tail)
            lo = [int(v) for v in sel.group(1).split(",")] if sel else [0] * nsrc
            hi = [int(v) for v in sel_hi.group(1).split(",")] if sel_hi else [1] * nsrc
            srcs = [re.sub(r"\s+op_sel.*$", "", o) for o in ops[1:1 + nsrc]]
            for k in range(1, nsrc):
                if k < len(srcs) and re.fullmatch(r"v\[\d+:\d+\]", srcs[k].strip()) and (lo[k] != 0 or hi[k] != 1):
                    found.append((ln, t))
                    break
    return found


if __name__ == "__main__":
    src = sys.argv[1]
    pats = sys.argv[2:]
    asm = src if src.endswith(".s") else compile_to_asm(os.path.join(CSRC, src) if not os.path.exists(src) else src, "/tmp/ape_isa_audit")
    for sym in kernel_symbols(asm):
        if pats and not any(p in sym for p in pats):
            continue
        r = audit(asm, sym, dma_barrier_slack=1 if (("halo_s32_kernel" in sym or "gemm_s32_" in sym) and "Lb1E" in sym) else 0)
        print(sym)
        print("   %d instructions, %d MFMA, %d ds_read, %d LDS-DMA; vmcnt literals %s; %s" %
              (r["n_insns"], r["n_mfma"], r["n_dsread"], r["n_dma"], r["vmcnt_literals"], r["meta"]))
        for f in r["findings"]:
            print("   FINDING:", f)
        for f in r["candidates"][:5]:
            print("   candidate (pass 2, pessimistic merge):", f)
        for h in sgpr_vmem_hazards(asm, sym):
            print("   SGPR->VMEM:", h)
        for h in mfma_asm_hazards(asm, sym):
            print("   MFMA/asm: line %d `%s` <- line %d `%s`: %s, %d of %d wait states" % h)
        sw = pk_src1_swizzles(asm, sym)
        if sw:
            print("   packed-f32 src1/src2 VGPR swizzle inside asm: %d instruction(s), first at line %d `%s`" % (len(sw), sw[0][0], sw[0][1]))
