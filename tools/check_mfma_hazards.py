#!/usr/bin/env python3
"""Checks, in the ISA hipcc generates for dp_w4.hip, the hazards the hand-written MFMA groups are responsible for themselves (inline
asm is opaque to the compiler's hazard recognizer).  Wait states required on gfx950 for the 2-pass v_mfma_f32_4x4x1_16b_f32 -- the
numbers hipcc itself pads with when the same sequences are written with builtins (tools/ubench/hazard_probe.hip.txt in the header of
this file's history: VALU -> MFMA operand 2; MFMA -> MFMA SrcA/B 4; MFMA -> MFMA SrcC 2; MFMA -> VALU / LDS / VMEM read 4):
  C   an MFMA whose C operand is the D tuple of an earlier MFMA                 >= 2 wait states between the two
  AB  an MFMA whose A or B operand lies in the D tuple of an earlier MFMA        >= 4
  R   any other instruction that reads a register an earlier MFMA wrote           >= 4
  V   an MFMA that reads a register a VALU instruction wrote                      >= 2
A wait state = any instruction issued in between; `s_nop N` counts N + 1.  The scan is linear per kernel (the chains are
straight-line code; a branch target resets nothing, which only makes the check stricter on fall-through paths).
Usage: tools/check_mfma_hazards.py [extra hipcc flags...]   Exit code 1 and a listing if any pair is closer than that."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as G

NEED = {"C": 2, "AB": 4, "R": 4, "V": 2}


def isa(extra, base_flags=None):
    """the ISA of dp_w4.hip as $HIPCC (the compiler build() uses) generates it with the product's flags (or `base_flags`) + `extra`"""
    out = os.path.join(tempfile.mkdtemp(prefix="w4isa_"), "dp_w4.s")
    flags = [f for f in (G.HIPCC_FLAGS if base_flags is None else base_flags) if f not in ("-shared", "-fPIC")] + list(extra)
    subprocess.check_call([os.environ.get("HIPCC", "hipcc"), *flags, "-S", "--cuda-device-only", "-o", out, os.path.join(G.CSRC, "dp_w4.hip")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return open(out).read()


def regs(tok):
    """register set of an operand token: v5, v[2:5], a[0:3] -> {("v", 5)}, ...; anything else -> empty"""
    tok = tok.strip().lstrip("-|").rstrip("|")
    m = re.match(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return {(m.group(1), r) for r in range(int(m.group(2)), int(m.group(3)) + 1)}
    m = re.match(r"([va])(\d+)\b", tok)
    return {(m.group(1), int(m.group(2)))} if m else set()


def check(text):
    counts = {k: 0 for k in NEED}
    bad, n_mfma = [], 0
    parts = re.split(r"^(_Z\w+):[^\n]*$", text, flags=re.M)
    for name, body in zip(parts[1::2], parts[2::2]):
        body = body.split(".Lfunc_end")[0]
        mfma_w = {}  # register -> wait-state index of the MFMA that last wrote it
        valu_w = {}  # register -> index of the VALU instruction that last wrote it
        t = 0
        for line in body.split("\n"):
            l = line.strip()
            l = l.split(";")[0].strip() if not l.startswith(";;") else ""
            if not l or l.startswith(".") or l.endswith(":"):
                continue
            op = l.split()[0]
            if op == "s_nop":
                t += int(l.split()[1]) + 1
                continue
            ops = [o.strip().split()[0] for o in l[len(op):].split(",") if o.strip()]
            if op.startswith("v_mfma"):
                n_mfma += 1
                dst, a, b, c = regs(ops[0]), regs(ops[1]), regs(ops[2]), regs(ops[3])
                for kind, rs in (("C", c), ("AB", a | b)):
                    t0 = max((mfma_w[r] for r in rs if r in mfma_w), default=None)
                    if t0 is not None and t - t0 - 1 < 64:
                        counts[kind] += 1
                        if t - t0 - 1 < NEED[kind]:
                            bad.append((kind, name, l, t - t0 - 1))
                t0 = max((valu_w[r] for r in (a | b | c) if r in valu_w), default=None)
                if t0 is not None and t - t0 - 1 < 64:
                    counts["V"] += 1
                    if t - t0 - 1 < NEED["V"]:
                        bad.append(("V", name, l, t - t0 - 1))
                for r in dst:
                    mfma_w[r] = t
                    valu_w.pop(r, None)
            else:
                rd = set().union(*[regs(o) for o in ops]) if ops else set()
                t0 = max((mfma_w[r] for r in rd if r in mfma_w), default=None)
                if t0 is not None and t - t0 - 1 < 64:
                    counts["R"] += 1
                    if t - t0 - 1 < NEED["R"]:
                        bad.append(("R", name, l, t - t0 - 1))
                if op.startswith("v_") and ops and not op.startswith("v_cmp"):
                    for r in regs(ops[0]):
                        valu_w[r] = t
                        mfma_w.pop(r, None)
                elif op.startswith(("ds_read", "global_load", "buffer_load", "scratch_load", "v_accvgpr")) and ops:
                    for r in regs(ops[0]):  # (a load's result: waited for by s_waitcnt, no wait-state rule; it ends the MFMA's claim)
                        mfma_w.pop(r, None)
                        valu_w.pop(r, None)
            t += 1
    return n_mfma, counts, bad


if __name__ == "__main__":
    argv = sys.argv[1:]
    base = None
    if argv and argv[0] == "--flags":  # tools/build_variant_w4.sh: the variant's complete flag list instead of the product's
        base, argv = argv[1].split(), argv[2:]
    n_mfma, counts, bad = check(isa(argv, base))
    print(f"{n_mfma} MFMAs; dependent pairs checked: {counts}; violations: {len(bad)}")
    for b in bad[:20]:
        print("  ", b)
    sys.exit(1 if bad else 0)
