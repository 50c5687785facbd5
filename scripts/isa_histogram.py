#!/usr/bin/env python3
"""ISA-level account of evalg_kernel<12,1>: where its wavefront-instructions go.

    python scripts/isa_histogram.py > profiles/archive/r05_evalg_isa_histogram.txt

Compiles flac_amd/csrc/flacgpu_evalg.hip for gfx950 to assembly with comment-only markers (`; MARK name`, inserted into a scratch copy
of the source at the boundaries of the kernel's phases), walks the kernel's control-flow graph from its entry carrying "the last
marker passed", and so assigns every instruction to a phase whatever order the compiler laid the blocks out in.  The FIR phases hold
one arm per chain length (the seven-way switch on the folded tap count NPF): arms are told apart by the v_dot2 chains they contain.
Static counts per phase and class are then weighted with the trip counts of the bench's -8 workload (per channel: 1 fixed + 9 LPC
candidates = 5 pairs; 4 pieces of 16 samples per lane; orders from the oracle on the bench signal) into wavefront-instructions per
inter-channel sample, next to the PMC pass's measured totals."""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter, defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "flac_amd", "csrc")
KERNEL = "_ZN7flacgpu12evalg_kernelILi12ELi1EEE"

MARKS = [
    ("\tconst ChanPrep pr = preps[fc];", "setup", "before"),
    ("\twhile(vmask) {", "pair_setup", "inside"),
    ("\t\t\tuint32_t AA[15], BB[14];\n\t\t\tload_piece_first(own, hist, AA, BB);", "first_piece", "before"),
    ("\t\t\tuint32_t AA[15], BB[14];\n\t\t\tload_piece(own + (8 * c - 7) * EG_ROW, AA, BB);", "piece", "before"),
    ("\t\tif(S & 8u) {", "after_pieces", "before"),
    ("\t\t// sums that leave the 32-bit arithmetic of the node passes", "search", "before"),
    ("\teg_decide<MAXORD>(R, P, pr, n, kbest", "decide", "before"),
]


def marked_source():
    src = open(os.path.join(CSRC, "flacgpu_evalg.hip")).read()
    for needle, name, how in MARKS:
        assert src.count(needle) == 1, needle
        m = 'asm volatile("; MARK %s");' % name
        src = src.replace(needle, (m + "\n" + needle) if how == "before" else (needle + "\n" + m))
    return src


def classify(op):
    if op.startswith("v_dot2"):
        return "valu:dot2"
    if op.startswith("v_sad"):
        return "valu:sad"
    if op in ("v_perm_b32", "v_alignbit_b32", "v_alignbyte_b32"):
        return "valu:shifted-word"
    if op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane"):
        return "valu:lane<->sgpr"
    if "_dpp" in op:
        return "valu:dpp"
    if op.startswith("v_"):
        return "valu:other"
    if op == "s_nop":
        return "salu:s_nop"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_cbranch") or op == "s_branch":
        return "salu:branch"
    if op.startswith("s_load") or op.startswith("s_buffer"):
        return "smem"
    if op.startswith("s_"):
        return "salu:other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_"):
        return "vmem"
    return "other"


def parse_kernel(asm):
    lines = asm.split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith(KERNEL) and l.rstrip().split(":")[0].endswith("SE_"))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    blocks, order, cur = {}, [], "entry"
    blocks[cur] = []
    order.append(cur)
    for l in lines[start + 1:end]:
        t = l.strip()
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            order.append(cur)
            continue
        if t.startswith("; MARK "):
            blocks[cur].append(("MARK", t[7:].strip()))
            continue
        if not t or t.startswith(";") or t.startswith("."):
            continue
        parts = t.split(None, 1)
        blocks[cur].append((parts[0], parts[1] if len(parts) > 1 else ""))
    return blocks, order


def walk(blocks, order):
    """region of every instruction: the last marker on a path from the entry (first visit wins; the code is structured)"""
    nxt = {b: order[i + 1] if i + 1 < len(order) else None for i, b in enumerate(order)}
    region_of = {}
    stack = [("entry", "prologue")]
    seen = set()
    while stack:
        b, reg = stack.pop()
        if b is None or b in seen:
            continue
        seen.add(b)
        fall = True
        for i, (op, args) in enumerate(blocks[b]):
            if op == "MARK":
                reg = args
                continue
            region_of[(b, i)] = reg
            if op == "s_branch":
                stack.append((args.split()[0], reg))
                fall = False
                break
            if op.startswith("s_cbranch"):
                stack.append((args.split()[0], reg))
            if op == "s_endpgm":
                fall = False
                break
        if fall:
            stack.append((nxt[b], reg))
    return region_of


NOTES = """
how to read this
 * the walk counts every instruction of a phase as executed once per visit of the phase: blocks behind wave-uniform conditions that
   the bench's workload never takes are in (the integer divisions of the image staging for lane runs that are no power of two: 12
   v_rcp_iflag sequences in `setup`; the e > 0 arms of the search).  The model therefore OVERSTATES set-up and SALU; the PMC pass of
   the same kernel (profiles/pmc_traffic.json, level8) measured 6.90 VALU and 1.99 SALU per sample against the model's totals above.
 * FIR arms = 80 % of the VALU work, and all of it is the formulation's arithmetic: per candidate-sample NPF v_dot2_i32_i16 (the
   folded tap count: 7 at order 12, 6 at 11 and 10, 2 for the fixed order 2), one v_lshrrev, one v_sad_u32.  `valu:other` inside the
   arms is the shift plus the lane-0 selects of the first piece.  Nothing in the arms is addressing, moves or conversions.
 * `salu:s_nop` in the arms: one per v_sad_u32.  The chain (dot2 x NPF + shift) and the sad are separate asm statements, and the
   compiler pads a read of an asm-defined VGPR by the next statement with one wait state (it cannot see into the asm: GCNHazardRecognizer
   treats inline asm as a possible dst-forwarding producer and counts it as zero wait states).  They are issue slots of the wavefront's
   scalar stream, not VALU work; the kernel runs at 4.2 SIMD-cycles per VALU instruction (0.866 ms x 1024 SIMDs x ~2.2 GHz / (65536
   wavefronts x 7066 VALU)), i.e. it is bound by the NUMBER of VALU instructions, and three other wavefronts of the SIMD issue into
   the slot a padded wavefront leaves.  Folding the whole piece into one asm statement needs 33 operands (limit 30).
 * the shifted sample words (v_perm_b32, 13 per pair and piece) are 2 % of the VALU work, the Rice search 11 %, set-up 5-6 %.
what was done with it (round 5), same-box A/B in profiles/archive/r05_b_ab_evalg.txt
 * rice_pass took ilog2 as the exponent of (float)x and the compiler, seeing a 64-bit product behind x, built the float with its
   64-bit sequence (v_lshlrev_b64, v_min, v_or, v_cvt, v_ldexp, v_frexp_exp): 31 - clz(2 x + 1) is three instructions -- the search
   105 -> 90 VALU per pair;  the divisor table of the search (91 integer divisions per channel) now comes from the host (JobTable).
 * together -0.10 VALU per sample (model 6.79 -> 6.69) and -0.3 % of the kernel's time (0.8664 -> 0.8634 ms): the instructions
   that are not the FIR's are too few to matter.  <= 6.3 VALU per sample needs fewer dot2 / shift / sad per candidate-sample, i.e. a
   different formulation of the FIR -- the int8 MFMA split is measured in profiles/archive/r05_mfma_fir_ubench.txt.
"""


def main():
    with tempfile.TemporaryDirectory() as td:
        for f in os.listdir(CSRC):
            if f.endswith(".h"):
                open(os.path.join(td, f), "w").write(open(os.path.join(CSRC, f)).read())
        open(os.path.join(td, "flacgpu_evalg.hip"), "w").write(marked_source())
        out = os.path.join(td, "evalg.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fconstexpr-steps=50000000",
                               "-I" + os.path.join(ROOT, "include"), "-I" + td, "--cuda-device-only", "-S", "-o", out, os.path.join(td, "flacgpu_evalg.hip")],
                              stderr=subprocess.DEVNULL)
        asm = open(out).read()
    blocks, order = parse_kernel(asm)
    region_of = walk(blocks, order)
    # FIR arms: maximal runs of blocks of a FIR region whose dot2 chains all have the same length; count per (region, NPF)
    per_region = defaultdict(Counter)
    arms = defaultdict(lambda: defaultdict(Counter))          # region -> npf -> class counts (both candidates' copies summed)
    for b in order:
        ins = blocks[b]
        # chain length of this block: dot2 instructions between ASMSTART/ASMEND pairs are emitted as separate ops here; use runs
        runs, run = [], 0
        for op, _ in ins:
            if op.startswith("v_dot2"):
                run += 1
            else:
                if run:
                    runs.append(run)
                run = 0
        if run:
            runs.append(run)
        npf = max(set(runs), key=runs.count) if runs else 0
        for i, (op, args) in enumerate(ins):
            if op == "MARK" or (b, i) not in region_of:
                continue
            reg = region_of[(b, i)]
            cls = classify(op)
            if npf and reg in ("first_piece", "piece", "after_pieces"):
                arms[reg][npf][cls] += 1
            else:
                per_region[reg][cls] += 1
    classes = ["valu:dot2", "valu:other", "valu:sad", "valu:shifted-word", "valu:lane<->sgpr", "valu:dpp", "salu:s_nop", "salu:branch", "salu:other", "smem", "lds", "vmem", "wait", "other"]
    print("ISA account of evalg_kernel<12,1> (flac_amd/csrc/flacgpu_evalg.hip, hipcc -O3 gfx950; scripts/isa_histogram.py)")
    print("static instruction counts per phase (control-flow walk from the entry; a phase = the code behind its marker)\n")
    print("%-34s" % "phase" + "".join("%9s" % c.split(":")[-1][:9] for c in classes))
    for reg in ["prologue", "setup", "pair_setup", "first_piece", "piece", "after_pieces", "search", "decide"]:
        c = per_region.get(reg, Counter())
        print("%-34s" % (reg + " (outside the FIR arms)" if reg in arms else reg) + "".join("%9d" % c[k] for k in classes))
        for npf in sorted(arms.get(reg, {})):
            a = arms[reg][npf]
            print("%-34s" % ("  %s: arms with %d-long chains" % (reg, npf)) + "".join("%9d" % a[k] for k in classes))
    # ---- dynamic model: the bench's -8 workload ------------------------------------------------------------------------------
    # per channel (a wavefront): 10 candidates in 5 pairs; a lane's run = 64 samples = the first piece + 3 loop pieces; orders of
    # the 9 LPC candidates from the oracle on the bench signal (scripts' header): 12: 0.589, 11: 0.351, 10: 0.060; fixed order 2
    order_share = {12: 0.589, 11: 0.351, 10: 0.060}
    npf_lpc = {o: (o + 2) // 2 for o in order_share}
    npf_fixed = 2
    pairs, pieces_loop = 5, 3
    def arm_cost(reg, npf):
        a = arms[reg].get(npf)
        if not a:
            return Counter()
        # both candidates' call sites were summed: one executed arm = half
        return Counter({k: v / 2.0 for k, v in a.items()})
    dyn = Counter()
    detail = defaultdict(Counter)
    def add(tag, c, times):
        for k, v in c.items():
            dyn[k] += v * times
            detail[tag][k] += v * times
    add("setup + prologue", per_region["prologue"] + per_region["setup"], 1)
    add("pair set-up (readlanes of the records)", per_region["pair_setup"], pairs)
    add("piece: loads, shifted words, switch", per_region["first_piece"], pairs)
    add("piece: loads, shifted words, switch", per_region["piece"], pairs * pieces_loop)
    add("piece: loads, shifted words, switch", per_region["after_pieces"], pairs)
    for reg, times in (("first_piece", 1), ("piece", pieces_loop)):
        add("FIR arms", arm_cost(reg, npf_fixed), times)                       # the fixed candidate
        for o, share in order_share.items():
            add("FIR arms", arm_cost(reg, npf_lpc[o]), times * 9 * share)
    add("Rice search of a pair (node passes, bookkeeping)", per_region["search"], pairs)
    add("decision", per_region["decide"], 1)
    per_sample = 4.0 / 4096.0          # four channels (wavefronts) per frame of 4096 inter-channel samples
    print("\ndynamic model, wavefront-instructions per inter-channel sample (flac -8 on the bench signal: 4 channels x (1 fixed + 9 LPC) candidates,")
    print("LPC orders 12 / 11 / 10 at 0.589 / 0.351 / 0.060 -- the oracle on 256 frames of the bench signal --, 64-sample lane runs):\n")
    print("%-52s %8s %8s %8s %8s" % ("where", "VALU", "SALU", "LDS", "wait"))
    tv = ts = tl = tw = 0.0
    for tag, c in detail.items():
        v = sum(x for k, x in c.items() if k.startswith("valu")) * per_sample
        s = sum(x for k, x in c.items() if k.startswith("salu")) * per_sample
        l = c["lds"] * per_sample
        w = c["wait"] * per_sample
        tv += v; ts += s; tl += l; tw += w
        print("%-52s %8.3f %8.3f %8.3f %8.3f" % (tag, v, s, l, w))
    print("%-52s %8.3f %8.3f %8.3f %8.3f" % ("model total", tv, ts, tl, tw))
    fir = detail["FIR arms"]
    print("\nof the FIR arms: dot2 %.3f, shift (valu:other) + sad %.3f + %.3f, s_nop %.3f, branch %.3f per sample" % (
        fir["valu:dot2"] * per_sample, fir["valu:other"] * per_sample, fir["valu:sad"] * per_sample, fir["salu:s_nop"] * per_sample, fir["salu:branch"] * per_sample))
    print("VALU by class over the whole kernel: " + ", ".join("%s %.3f" % (k.split(":")[1], dyn[k] * per_sample) for k in classes if k.startswith("valu")))
    print("SALU by class over the whole kernel: " + ", ".join("%s %.3f" % (k.split(":")[1], dyn[k] * per_sample) for k in classes if k.startswith("salu")))
    print(NOTES)


if __name__ == "__main__":
    main()
