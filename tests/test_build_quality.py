"""CPU: the product's kernels as hipcc compiles them for gfx950 (the build flags of __graft_entry__): no register spills, no
scratch, the register budget the design counts on.  A regression here costs performance silently -- the kernels still run."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None, reason="needs hipcc")


def _kernel_notes(src, tmp_path):
    out = tmp_path / (src + ".s")
    flags = [f for f in G.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    if src in G.SCHED_OVERRIDE:  # the per-source flags of __graft_entry__._compile_objects
        i = flags.index("-amdgpu-sched-strategy=max-ilp")
        flags[i - 1:i + 1] = ["-mllvm", "-amdgpu-sched-strategy=" + G.SCHED_OVERRIDE[src]] if G.SCHED_OVERRIDE[src] else []
    flags += G.EXTRA_FLAGS.get(src, [])
    subprocess.check_call(["hipcc", *flags, "-S", "--cuda-device-only", "-o", str(out), os.path.join(G.CSRC, src)],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    text = out.read_text()
    notes = {}
    for blk in text.split("  - .agpr_count:")[1:]:  # one metadata block per kernel
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        get = lambda key: int(re.search(rf"\.{key}:\s+(\d+)", blk).group(1))
        notes[name] = dict(agpr=int(blk.split()[0]), vgpr=get("vgpr_count"), vspill=get("vgpr_spill_count"),
                           scratch=get("private_segment_fixed_size"), lds=get("group_segment_fixed_size"))
    return notes


def test_optimise_kernel_keeps_its_register_budget(tmp_path):
    notes = _kernel_notes("dp_w4.hip", tmp_path)
    kernels = {k: v for k, v in notes.items() if "dp_w4_kernel" in k}
    assert len(kernels) == 6, list(notes)  # <4, false>, <4, true> (early stop), <4, true, true> (whole-sequence launches), each also as LONG (n_iter > 256)
    for name, n in kernels.items():
        assert n["lds"] <= 160 * 1024, (name, n)
        if "ILi4ELb1ELb1ELb" in name:  # the whole-sequence instantiations
            # the step loop around the iteration loop keeps more alive: some spills (outside the iteration loop) are accepted there;
            # inside it the hand-padded MFMA groups must not be interleaved with copies of their operands (checked below)
            assert n["vspill"] <= 40 and n["scratch"] <= 160, (name, n)
            continue
        assert n["vspill"] == 0 and n["scratch"] == 0, (name, n)
        assert n["agpr"] == 256, (name, n)          # every accumulator register holds a resident weight


def test_sequence_kernel_moves_no_weight_inside_the_iteration_loop(tmp_path):
    """Found the hard way (round 3): with every accumulator register holding a resident weight, the whole-sequence instantiation made
    the register allocator copy weight groups (v_accvgpr_mov) right in front of the inline-asm MFMA groups that read them -- a
    hazard the hand-padded groups do not cover, and the results were wrong.  It keeps one resident group of bL2 now; this test
    fails if a compiler or source change brings accumulator-register copies or scratch reloads back between the loop's MFMAs."""
    out = tmp_path / "dp_w4.s"
    flags = [f for f in G.HIPCC_FLAGS if f not in ("-shared", "-fPIC")]
    subprocess.check_call(["hipcc", *flags, "-S", "--cuda-device-only", "-o", str(out), os.path.join(G.CSRC, "dp_w4.hip")],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    lines = out.read_text().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_Z12dp_w4_kernelILi4ELb1ELb1ELb0EEv5KArgs:"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    body = lines[start:end]
    mf = [i for i, l in enumerate(body) if "v_mfma" in l]
    loop = body[mf[40]:mf[-1]]  # (the first MFMAs belong to the set-up's transposes)
    bad = [l.strip() for l in loop if "v_accvgpr_mov" in l or "v_accvgpr_write" in l or "scratch_" in l]
    assert not bad, bad[:8]


def test_temporal_kernel_variants_fit_their_occupancy(tmp_path):
    notes = _kernel_notes("dp_temporal.hip", tmp_path)
    kernels = {k: v for k, v in notes.items() if "dp_temporal_kernel" in k}
    # <2, 1> (one workgroup per CU), <2, 1, TEAM> (few sequences: G workgroups per sequence), <4, 1> and <4, 2> (two workgroups per CU; one / two sequences each)
    # ... and <4, 2, PAIR> (round 6: ONE 1024-thread workgroup of two <4, 2> halves per CU, twice the LDS)
    assert len(kernels) == 5, list(notes)
    for name, n in kernels.items():
        assert n["vspill"] == 0 and n["scratch"] == 0, (name, n)
        one_per_cu = "ILi2ELi1ELb1ELb0EE" in name or "ILi4ELi2ELb0ELb1EE" in name  # (a team's workgroups have a CU each: n_seq * G <= CUs / 2; a PAIR workgroup is the CU's only one)
        assert n["lds"] <= (160 if one_per_cu else 80) * 1024, (name, n)  # the others: two workgroups per CU
    assert sorted(n["vgpr"] <= 128 for n in kernels.values()) == [False, False, True, True, True]  # the 2-waves-per-SIMD variants use the full file


@pytest.mark.parametrize("src, one_wave", [("dp_w16.hip", True), ("dp_w16_es.hip", True), ("dp_w16_2w.hip", False), ("dp_w16_2w_es.hip", False)])
def test_w16_instantiations_keep_their_register_budget(tmp_path, src, one_wave):
    """the 16-frames-per-wave kernel, each instantiation with its own flags: one wave per SIMD has the whole register file and must
    not spill; two waves per SIMD run in 256 registers with a bounded spill (latency the partner wave covers); no packed fp32
    instruction anywhere (it would stall for a bf16 MFMA in flight: DESIGN.md 6.4); the 135 KB weight image plus tables in LDS"""
    out = tmp_path / (src + ".s")
    notes = _kernel_notes(src, tmp_path)
    (name, n), = [(k, v) for k, v in notes.items() if "dp_w16_kernel" in k]
    assert 135 * 1024 <= n["lds"] <= 160 * 1024, (name, n)
    if one_wave:
        # no scratch memory; the early-stop unit parks two values in the accumulator half since the round-5 input screening (its loop
        # is the same 2.5 k instructions with 2 more v_accvgpr_write: 0.1 %)
        assert n["scratch"] == 0 and n["vspill"] <= (2 if "_es" in src else 0), (name, n)
    else:
        assert n["vgpr"] <= 256 and n["vspill"] <= 72, (name, n)
    text = out.read_text()
    assert "v_pk_fma_f32" not in text and "v_pk_mul_f32" not in text and "v_pk_add_f32" not in text


def test_hand_written_mfma_groups_keep_their_wait_states():
    """dp_w4.hip's MFMA groups are inline asm: the hazard between an MFMA and a later one that accumulates into its result (two wait
    states for the 2-pass 4x4x1) is the source's responsibility -- and since round 4 one of the two is the `s_nop 0` hipcc itself puts
    between two asm statements (dp_w4.hip: W4_TAIL).  tools/check_mfma_hazards.py walks the generated ISA of all three instantiations."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_mfma_hazards as H

    n_mfma, counts, bad = H.check(H.isa([]))
    # 396 per instantiation + the fixed-count kernel's two extra copies of the bL2 chain with its dead K-groups left out (96 and 72 of 104 MFMAs)
    assert n_mfma == 2 * (3 * 396 + 96 + 72) and counts["C"] > 1000  # (every instantiation twice: the LONG copies for n_iter > 256) and counts["AB"] >= 15 and counts["R"] >= 30 and not bad, (n_mfma, counts, bad[:5])
