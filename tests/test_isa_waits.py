"""Static ISA audit of the kernels whose hazards the compiler cannot see (no GPU needed: hipcc cross-compiles gfx950 with -save-temps,
tools/isa_audit.py walks the instruction stream).  What is asserted for every named kernel:

  * no inline-asm / compiler `ds_read` result is read or overwritten before an `s_waitcnt lgkmcnt` has retired it, on ANY path of the
    control-flow graph (path-exact walk).  This found a real one: where hipcc saw the look-ahead fragment reads of a workgroup's last
    k-tile dead, it gave them all one register quad and re-used the quad for a branch condition while the LDS data was still on its way
    (conv_gemm_s32.hip, odd-k-tile tail) -- the reads' destinations are now kept allocated up to their wait (`keep_regs`);
  * between an LDS-DMA (`buffer_load ... lds`) and the next `s_barrier` there is an `s_waitcnt` with a vmcnt field;
  * no VGPR spill and no scratch (a scratch reload inside a DMA-pipelined loop waits vmcnt(0) and drains the pipeline): found 4..14 spilled
    registers (hoisted loop-invariant epilogue / halo-item addresses) in halo_s32 and in the up_3 kernel, now formed where they are used;
  * the kernel stays inside its register budget (<= 256 VGPRs at two waves per SIMD).
Whether a hand-counted `vmcnt(N)` has the right N depends on run-time trip counts and is what the bit-exact GPU tests cover."""
import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, "tools"))

# file -> [(substring of the mangled kernel name, human name, LDS-DMA kernel?)]
KERNELS = {
    "conv3x3_halo_s32.hip": [("halo_s32_kernelILi%dELb%dE" % (d, pp), "halo_s32_kernel<%d, %s>" % (d, "true" if pp else "false"), True)
                             for d in (1, 2, 4) for pp in (1, 0)],
    "conv3x3_halo_mx.hip": [("halo_mx_kernelILi1E", "halo_mx_kernel<1>", True), ("halo_mx_kernelILi2E", "halo_mx_kernel<2>", True),
                            ("halo_mx_kernelILi4E", "halo_mx_kernel<4>", True)],
    "conv_gemm_s32.hip": [("gemm_s32%s_kernelILi%dELb%dE" % (res, bn, pp), "gemm_s32%s_kernel<%d, %s>" % (res, bn, "true" if pp else "false"), True)
                          for res in ("", "_res") for bn in (128, 192, 256) for pp in (1, 0)],
    "conv3x3_halo.hip": [("conv3x3_halo_kernelILi3ELi1ELi64ELb1ELb1E", "conv3x3_halo_kernel<3,1,64,true,true>", False),
                         ("conv3x3_halo_kernelILi3ELi1ELi64ELb0ELb0E", "conv3x3_halo_kernel<3,1,64,false,false>", False)],
}


@pytest.fixture(scope="module")
def audit_mod():
    import isa_audit
    if not os.path.exists(isa_audit.HIPCC):
        pytest.skip("hipcc not present")
    return isa_audit


@pytest.mark.parametrize("hip_file", sorted(KERNELS))
def test_wait_counters_and_spills(audit_mod, hip_file, tmp_path_factory):
    out_dir = os.path.join(REPO, "autoposeestimation_amd", "csrc", "build", "isa_audit")
    asm = audit_mod.compile_to_asm(os.path.join(audit_mod.CSRC, hip_file), out_dir)
    symbols = audit_mod.kernel_symbols(asm)
    for key, name, has_dma in KERNELS[hip_file]:
        sym = [s for s in symbols if key in s]
        assert len(sym) == 1, (name, sym)
        # (halo_s32's ping-pong form retires a piece in front of the SECOND barrier after its issue: tools/isa_audit.py, dma_barrier_slack)
        r = audit_mod.audit(asm, sym[0], dma_barrier_slack=1 if (hip_file in ("conv3x3_halo_s32.hip", "conv_gemm_s32.hip") and "true" in name) else 0)
        # the walk saw the real kernel (gemm_s32's ping-pong form is ONE rolled k-tile body: 48 MFMAs and 40 reads at BN = 128)
        assert r["n_mfma"] >= 48 and r["n_dsread"] >= 40, (name, r["n_mfma"], r["n_dsread"])
        assert (r["n_dma"] > 0) == has_dma, (name, r["n_dma"])
        assert not r["findings"], "%s:\n  %s" % (name, "\n  ".join(r["findings"]))
        meta = r["meta"]
        assert meta["vgpr_spill_count"] == 0 and meta["private_segment_fixed_size"] == 0, (name, meta)
        assert meta["vgpr_count"] <= 256, (name, meta)


def test_the_audit_catches_a_missing_wait(audit_mod, tmp_path):
    """the checker itself: a hand-made instruction stream with a ds_read result used one instruction early, a wait that is one too weak
    on one of two paths, and a barrier right behind an LDS-DMA"""
    asm = tmp_path / "toy.s"
    asm.write_text("""
toy:
	ds_read_b128 v[4:7], v1
	ds_read_b128 v[8:11], v1 offset:16
	s_waitcnt lgkmcnt(1)
	v_mfma_f32_16x16x32_bf16 v[20:23], v[4:7], v[4:7], v[20:23]
	v_add_u32_e32 v8, 1, v2
	s_waitcnt lgkmcnt(0)
	ds_read_b128 v[12:15], v1
	s_cbranch_scc1 .LBB0_2
	ds_read_b128 v[16:19], v1
.LBB0_2:
	s_waitcnt lgkmcnt(1)
	v_mov_b32_e32 v30, v12
	buffer_load_dwordx4 v3, s[8:11], s0 offen lds
	s_barrier
	s_waitcnt vmcnt(0) lgkmcnt(0)
	s_endpgm
.Lfunc_end0:
""")
    r = audit_mod.audit(str(asm), "toy")
    kinds = sorted(f.split(" at line")[0] for f in r["findings"])
    lines = sorted(int(f.split(" at line ")[1].split(" ")[0]) for f in r["findings"])
    # v8 overwritten while its ds_read is pending (line 7); lgkmcnt(1) retires the read of v12 only on the path that issued a younger read:
    # still pending on the branch-taken path (line 14); barrier behind the DMA (line 16)
    assert kinds == ["barrier behind an un-waited LDS-DMA", "ds_read result used before its wait", "ds_read result used before its wait"], r["findings"]
    assert lines == [7, 14, 16], r["findings"]


def test_inline_asm_vector_memory_reads_no_freshly_reloaded_sgpr(audit_mod, tmp_path_factory):
    """upconv_fused.hip issues its LDS-DMA from an asm statement (hipcc must not know of the transfer).  hipcc reloads spilled SGPRs with
    v_readlane right in front of whatever uses them -- and owes a vector-memory instruction that reads such an SGPR five wait states, which it
    inserts for its own instructions only.  Every shipped variant has such reloads in front of DMA statements; the statement therefore
    starts with `s_nop 4`.  This walks the ISA of every variant and fails if a descriptor / offset SGPR of an asm buffer load was written by
    a VALU instruction fewer than five wait states earlier."""
    out = os.path.join(REPO, "autoposeestimation_amd", "csrc", "build", "isa_audit")
    asm = audit_mod.compile_to_asm(os.path.join(REPO, "autoposeestimation_amd", "csrc", "upconv_fused.hip"), out)
    syms = [s for s in audit_mod.kernel_symbols(asm) if "upconv_fused_kernel" in s]
    assert len(syms) == 4
    for sym in syms:
        assert audit_mod.sgpr_vmem_hazards(asm, sym) == [], sym
    # the walk sees the pattern it is looking for (a reload two instructions in front of the load) and accepts five wait states
    probe = os.path.join(str(tmp_path_factory.mktemp("isa")), "probe.s")
    A, E = "\t;;#ASMSTART\n", "\t;;#ASMEND\n"
    open(probe, "w").write("\nk1:\n\tv_readlane_b32 s37, v9, 3\n" + A + "\ts_mov_b32 m0, s6\n\tbuffer_load_dwordx4 v1, s[36:39], 0 offen lds\n" + E + "\ts_endpgm\n"
                           "\nk2:\n\tv_readlane_b32 s2, v9, 3\n" + A + "\ts_nop 4\n\tbuffer_load_dwordx4 v[4:7], v1, s[36:39], s2 offen offset:64\n" + E + "\ts_endpgm\n"
                           "\nk3:\n\tv_readlane_b32 s2, v9, 3\n" + A + "\tbuffer_load_dwordx4 v[4:7], v1, s[36:39], s2 offen\n" + E + "\ts_endpgm\n"
                           "\nk4:\n\tv_readlane_b32 s2, v9, 3\n\tbuffer_load_dwordx4 v[4:7], v1, s[36:39], s2 offen\n\ts_endpgm\n"       # hipcc's own: its to pad
                           "\nk5:\n\tv_readlane_b32 s2, v9, 3\n.LBB0_3:\n" + A + "\tbuffer_load_dwordx4 v[4:7], v1, s[36:39], s2 offen\n" + E + "\ts_endpgm\n")
    assert len(audit_mod.sgpr_vmem_hazards(probe, "k1")) == 1 and audit_mod.sgpr_vmem_hazards(probe, "k2") == [] and len(audit_mod.sgpr_vmem_hazards(probe, "k3")) == 1
    assert audit_mod.sgpr_vmem_hazards(probe, "k4") == []
    unknown = audit_mod.sgpr_vmem_hazards(probe, "k5")          # a label inside the window: reported, not taken for clean
    assert len(unknown) == 1 and "unknown" in unknown[0][2]



def test_matrix_instruction_wait_states_around_asm_statements(audit_mod, tmp_path):
    """hipcc pads the wait states software owes around matrix instructions for the instructions it emits itself, and treats an asm statement
    neither as a vector nor as a matrix instruction: an MFMA result read (or overwritten) by an instruction inside an asm statement, an MFMA's
    SrcC overwritten from inside one, or an asm vector instruction feeding an MFMA operand gets no pad.  Every kernel that mixes asm statements
    with matrix instructions is walked over ALL control-flow paths (loop back-edges included): none may have such a pair inside the window."""
    out_dir = os.path.join(REPO, "autoposeestimation_amd", "csrc", "build", "isa_audit")
    for hip_file in sorted(KERNELS) + ["upconv_fused.hip"]:
        asm = audit_mod.compile_to_asm(os.path.join(audit_mod.CSRC, hip_file), out_dir)
        syms = audit_mod.kernel_symbols(asm)
        assert syms
        for sym in syms:
            assert audit_mod.mfma_asm_hazards(asm, sym) == [], (hip_file, sym)
    toy = tmp_path / "toy.s"
    toy.write_text("""
toy:
	v_mfma_f32_16x16x32_bf16 v[20:23], v[4:7], v[8:11], v[20:23]
	s_branch .LBB0_1
.LBB0_1:
	s_nop 1
	;;#ASMSTART
	v_pk_fma_f32 v[30:31], v[20:21], v[40:41], 0
	;;#ASMEND
	v_mfma_f32_16x16x32_bf16 v[60:63], v[4:7], v[8:11], v[64:67]
	s_nop 4
	;;#ASMSTART
	v_max_f32 v64, v51, v52
	v_max_f32 v50, v51, v52
	;;#ASMEND
	v_mfma_f32_16x16x4_f32 v[70:73], v50, v74, v[70:73]
	v_mfma_f32_16x16x4_f32 v[70:73], v51, v74, v[70:73]
	s_endpgm
.Lfunc_end0:
""")
    kinds = sorted(h[4] for h in audit_mod.mfma_asm_hazards(str(toy), "toy"))
    # the result read two states later across a block boundary; SrcC v[64:67] overwritten five states later (a 4-pass MFMA is owed eight);
    # an MFMA operand written by the asm one state earlier; the accumulate chain itself is not a finding
    assert kinds == ["MFMA SrcC read -> asm VALU write", "MFMA write -> asm access", "asm VALU write -> MFMA read"], kinds


def test_packed_f32_broadcasts_sit_in_src0(audit_mod):
    """csrc/upconv_fused.hip's interpolations are hand-written v_pk_fma_f32 / v_pk_mul_f32 with an op_sel weight broadcast.  With the weight
    pair (a VGPR pair) as the SECOND source the low half of lanes 48-63 came out wrong under back-to-back issue -- rounds 1-4's "rare whole
    16-pixel groups" (tools/stress_upfuse.py fails in every launch on that build); as src0 the same instruction is clean.  Green: no asm
    statement of the product build swizzles a VGPR pair in src1 / src2.  Red: the audit flags the historic operand order."""
    out = os.path.join(REPO, "autoposeestimation_amd", "csrc", "build", "isa_audit")
    src = os.path.join(REPO, "autoposeestimation_amd", "csrc", "upconv_fused.hip")
    asm = audit_mod.compile_to_asm(src, out)
    syms = [s for s in audit_mod.kernel_symbols(asm) if "upconv_fused_kernel" in s]
    assert len(syms) == 4
    for sym in syms:
        assert audit_mod.pk_src1_swizzles(asm, sym) == [], sym
        n_pk = sum(1 for blk in audit_mod._parse_cfg_with_asm(asm, sym)[0] for _, t, ia in blk if ia and t.startswith("v_pk_"))
        assert n_pk >= 100, (sym, n_pk)             # the walk saw the hand-written interpolations
    old = audit_mod.compile_to_asm(src, out + "_src1", defs=("-DAPE_ASM_SRC1_BCAST=1",))
    for sym in [s for s in audit_mod.kernel_symbols(old) if "upconv_fused_kernel" in s]:
        assert len(audit_mod.pk_src1_swizzles(old, sym)) >= 50, sym


def test_no_kernel_reads_a_swizzled_vgpr_pair_through_src1_of_a_packed_f32_instruction(audit_mod):
    """The operand form behind two wrong-result faults on gfx950 -- v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 with an op_sel / op_sel_hi
    swizzle on a VGPR pair in src1: lanes 48-63 take a wrong dword for the low half under back-to-back issue.  First in upconv_fused's
    hand-written statements (round 4/5), then in COMPILER output: hipcc's SLP vectoriser built pose_select_kernel's rotation from such
    instructions, and the new points of whole 16-lane groups came out wrong whenever a second stream kept the chip busy
    (tools/stress_pose_select.py: 337 of 600 launches beside the crops' CNN; tools/stress_mixed_graphs.py).  The library is therefore built
    with -fno-slp-vectorize (csrc/Makefile) and EVERY kernel's ISA is scanned here, compiler output included; the same scan of pose.hip
    built WITH the pass must find the instructions (the check bites)."""
    out_dir = os.path.join(REPO, "autoposeestimation_amd", "csrc", "build", "isa_audit")
    seen = 0
    for hip_file in sorted(f for f in os.listdir(audit_mod.CSRC) if f.endswith(".hip")):
        asm = audit_mod.compile_to_asm(os.path.join(audit_mod.CSRC, hip_file), out_dir)
        for sym in audit_mod.kernel_symbols(asm):
            seen += 1
            found = audit_mod.pk_src1_swizzles(asm, sym, asm_only=False)
            assert not found, "%s %s:\n  %s" % (hip_file, sym, "\n  ".join(t.strip() for _, t in found[:8]))
    assert seen >= 100                                                       # (every kernel of the library)
    asm = audit_mod.compile_to_asm(os.path.join(audit_mod.CSRC, "pose.hip"), out_dir + "_slp", defs=("-fslp-vectorize",))
    sym = [s for s in audit_mod.kernel_symbols(asm) if "pose_select_kernel" in s]
    assert len(sym) == 1 and len(audit_mod.pk_src1_swizzles(asm, sym[0], asm_only=False)) >= 1
