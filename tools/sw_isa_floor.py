#!/usr/bin/env python3
"""tools/sw_isa_floor.py [--classes 64,30 32,18 ...] [--mode 0] — VERDICT r05 item 4 (ii): what the packed gapped-DP kernel issues per DP cell, from its ISA.

Compiles unicore_amd/csrc/uc_sw_pk_m<mode>.hip with --save-temps (hipcc, gfx950; seconds), takes `sw_pk_kernel<G, R, MODE, NW>` out of the device assembly,
finds the loop trip (the longest branch-free instruction run of the function: the two unrolled DP steps of the 2-step trip) and lists its instructions by
class, per trip and per DP cell (a trip = 2 steps x R rows x 2 alignments per lane).  The VALU instructions are split by what the source needs them for
(uc_sw_pk_impl.hpp:do_step): the cell recurrence, the end tracking, the per-step overhead that R rows amortise.  Writes a text report to stdout."""
import argparse, collections, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_body(asm, G, R, mode, NW):
    name = "_ZN2uc12sw_pk_kernelILi%dELi%dELi%dELi%dEEEvNS_6SwArgsE" % (G, R, mode, NW)
    out, on = [], False
    for l in asm:
        if l.startswith(name + ":"):
            on = True
            continue
        if on:
            if l.startswith(".Lfunc_end"):
                break
            out.append(l.rstrip("\n"))
    if not out:
        raise SystemExit("kernel %s not found" % name)
    return name, out


def longest_run(body):
    """instructions of the longest run without a label or a branch"""
    best, cur = [], []
    for l in body:
        t = l.strip()
        if not t or t.startswith(";") or t.startswith("."):
            if t.startswith(".LBB"):
                if len(cur) > len(best): best = cur
                cur = []
            continue
        op = t.split()[0]
        if op.startswith("s_cbranch") or op == "s_branch" or op == "s_endpgm":
            if len(cur) > len(best): best = cur
            cur = []
            continue
        cur.append(t)
    if len(cur) > len(best): best = cur
    return best


def classify(ins):
    op = ins.split()[0]
    dpp = "row_shr" in ins or "row_shl" in ins or "wave_shr" in ins or "row_bcast" in ins or "quad_perm" in ins or "row_share" in ins or "_dpp" in op
    if op.startswith("ds_"): return "LDS (profile reads)"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"): return "VMEM"
    if op.startswith("s_load") or op.startswith("s_buffer"): return "SMEM (letter stream through the scalar cache)"
    if op.startswith("s_waitcnt") or op == "s_nop" or op.startswith("s_delay"): return "waits / nops"
    if op.startswith("s_"): return "SALU"
    if dpp: return "VALU: lane shifts (DPP) — H, F, letters from the neighbour lane"
    if op == "v_pk_maximum3_f16": return "VALU: v_pk_maximum3_f16 — H = max3(x, e, f); row / column maxima"
    if op == "v_perm_b32": return "VALU: v_perm_b32 — byte r of the two profile words -> packed u16 score"
    if op in ("v_pk_add_u16", "v_pk_sub_u16"): return "VALU: v_pk_add/sub_u16 clamp — x = max(Hdiag + s, 0); h - open; e - ext; f - ext"
    if op in ("v_pk_max_u16", "v_pk_max_i16", "v_pk_min_u16"): return "VALU: v_pk_max_u16 — E', F', best"
    if op.startswith("v_add") or op.startswith("v_and") or op.startswith("v_lshl") or op.startswith("v_mad") or op.startswith("v_mul") or op.startswith("v_or") or op.startswith("v_lshr") or op.startswith("v_bfe") or op.startswith("v_sub"):
        return "VALU: integer add / and / shift — profile word sums, LDS addresses"
    if op.startswith("v_cmp") or op.startswith("v_cndmask"): return "VALU: compare / select — first column of a new maximum"
    if op.startswith("v_mov") or op.startswith("v_readfirstlane") or op.startswith("v_readlane") or op.startswith("v_writelane") or op.startswith("v_accvgpr"):
        return "VALU: moves"
    if op.startswith("v_"): return "VALU: other (%s)" % op
    return "other (%s)" % op


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--classes", nargs="*", default=["64,30", "32,18", "16,20"])
    ap.add_argument("--mode", type=int, default=0)
    a = ap.parse_args()
    d = tempfile.mkdtemp(prefix="uc_isa_")
    src = os.path.join(ROOT, "unicore_amd", "csrc", "uc_sw_pk_m%d.hip" % a.mode)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-fPIC", "-I" + os.path.join(ROOT, "include"),
                           "-I" + os.path.join(ROOT, "unicore_amd", "csrc"), "--save-temps", "-c", src, "-o", os.path.join(d, "m.o")], cwd=d, stderr=subprocess.DEVNULL)
    sfile = [f for f in os.listdir(d) if f.endswith("gfx950.s")][0]
    asm = open(os.path.join(d, sfile)).read().split("\n")
    print("# ISA of the packed gapped-DP kernel's loop trip (MODE %d), hipcc -O3 --offload-arch=gfx950, %s" % (a.mode, os.path.basename(src)))
    print("# a trip = 2 DP steps; per lane a step updates R rows x 2 alignments (packed u16 halves) = 2R cells\n")
    for c in a.classes:
        G, R = [int(x) for x in c.split(",")]
        NW = 2 if G == 16 else 4 if G == 32 else (2 if R < 14 else 8)
        name, body = kernel_body(asm, G, R, a.mode, NW)
        run = longest_run(body)
        cnt = collections.Counter(classify(i) for i in run)
        # one v_perm_b32 per row and step (MODE 0-6): the run covers that many row-steps of 2 cells each (the known-score modes branch between the two
        # steps of a trip, so their longest run is ONE step)
        nperm = sum(1 for i in run if i.split()[0] == "v_perm_b32")
        steps = max(1, round(nperm / R))
        cells = 2 * R * steps
        valu = sum(v for k, v in cnt.items() if k.startswith("VALU"))
        print("## sw_pk_kernel<%d, %d, %d, %d>: longest branch-free run = %d DP step(s), %d instructions, %d VALU; %d cells per lane -> %.3f VALU instructions per cell" % (G, R, a.mode, NW, steps, len(run), valu, cells, valu / cells))
        for k, v in sorted(cnt.items(), key=lambda kv: (not kv[0].startswith("VALU"), -kv[1])):
            print("  %5d  %6.3f / cell   %s" % (v, v / cells, k))
        rec = (10 if a.mode == 0 else 9.5) * R * steps                      # per step and row: perm + add + sub + max3 + 3 sub + 2 max = 9, + colmax max3 every second row = 0.5, + rowbest max3 every second step = 0.5
        print("  source-level count of the recurrence (uc_sw_pk_impl.hpp:284-300): per row and step 1 perm + 2 (x) + 1 max3 (h) + 3 sat-sub + 2 max = 9, "
              "+ 0.5 column max3 + 0.5 row max3 (MODE 0 only) -> %d per run = %.3f per cell; the rest of the VALU count (%d = %.3f per cell) is per-STEP work "
              "that R rows amortise: profile sums, LDS addressing, DPP shifts, first-column select, best" % (rec, rec / cells, valu - rec, (valu - rec) / cells))
        print()


if __name__ == "__main__":
    main()
