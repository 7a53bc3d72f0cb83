#!/usr/bin/env python3
"""Static VALU instruction mix of the Poseidon leaf kernel, weighted by loop trip counts.

Compiles merlin-zkevm-prover_amd/csrc/poseidon.hip to gfx950 assembly (device only), walks the body of
k_linear_hash_rows_lines<variant>, weights every instruction by the trip counts of the loops around it (full rounds
4, grouped partial rounds 2 x closing loop 11, one pass of the sponge loop = one permutation), leaves out the
rare-fix blocks (the few instructions behind an "s_cbranch_vccz" that the common path jumps over), and sorts the
VALU opcodes into the two issue classes measured on this chip by tools/ubench_int2.hip and tools/ubench_int3.hip
(profiles/r01_ubench_int_issue_rates*.txt, profiles/r02_ubench_int3_issue_rates.txt):

  2-clk class (~930 G wave-instr/s measured, 1229 G at a nominal 2.4 GHz): v_mov_b32, v_add_u32, v_sub_u32, v_and / v_or /
      v_xor / v_not (plain 32-bit VOP1 / VOP2 ALU ops)
  4-clk class (~565 G wave-instr/s measured,  614 G nominal): everything else the kernels use -- v_mad_u64_u32,
      v_lshl_add_u64, carry adds / subs, every compare (32- and 64-bit), v_cndmask (4.4 clk with an SGPR or VCC mask;
      r01's 23-clk figure was the VOP2 form selecting into its own source with a never-written VCC), shifts, v_mov_b64,
      v_add3, v_and_or, v_bfe, v_perm, v_alignbit, DPP moves, multiplies, dot and packed ops

Prints a JSON object: VALU instructions per permutation (static estimate; the PMC count SQ_INSTS_VALU is the
measured one) and the class fractions that bench.py's issue roofline uses.
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "merlin-zkevm-prover_amd", "csrc")
TWO_CLK = ("v_mov_b32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32")


def main():
    variant = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    prefix = "_Z24k_linear_hash_rows_linesILi%dELb0EE" % variant
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "pos.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-gpu-rdc", "-S",
                               "--cuda-device-only", os.path.join(CSRC, "poseidon.hip"), "-o", asm],
                              stderr=subprocess.DEVNULL)
        txt = open(asm).read()
    m = re.search(r"\n(" + re.escape(prefix) + r"\w*):", txt)
    name = m.group(1)
    i = m.start()
    j = txt.index("s_endpgm", i)
    instrs, labels = [], {}
    for l in txt[i:j].split("\n"):
        s = l.strip()
        if not s or s.startswith(";"):
            continue
        if s.startswith(".LBB") and s.split()[0].endswith(":"):
            labels[s.split(":")[0]] = len(instrs)
            continue
        if s.startswith(".") or s.endswith(":"):
            continue
        instrs.append(s)
    weight = [1.0] * len(instrs)
    # rare-fix blocks: a forward s_cbranch_vccz over fewer than 10 instructions
    for k, l in enumerate(instrs):
        m = re.match(r"s_cbranch_vccz\s+(\.LBB\S+)", l)
        if m and m.group(1) in labels and 0 < labels[m.group(1)] - k < 10:
            for t in range(k + 1, labels[m.group(1)]):
                weight[t] = 0.0
    # loops = backward branches; trip counts by what the loop contains
    loops = []
    for k, l in enumerate(instrs):
        m = re.match(r"s_c?branch\w*\s+(\.LBB\S+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] <= k and not l.startswith("s_cbranch_vcc"):
            loops.append((labels[m.group(1)], k))
    loops.sort(key=lambda ab: ab[1] - ab[0])
    trips = []
    for a, b in loops:
        n = b - a
        inner = [(x, y) for (x, y) in loops if x >= a and y <= b and (x, y) != (a, b)]
        if n > 4000:
            trip = 1.0      # the sponge / column-window loops: one pass per permutation
        elif inner:
            # the two groups of 11 partial rounds contain the short closing loop; any other backward branch around a
            # loop is block placement, not iteration
            trip = 2.0 if any(y - x < 400 for (x, y) in inner) else 1.0
        elif n > 1000:
            trip = 4.0      # four full rounds
        else:
            trip = 11.0     # closing dot products of a group
        trips.append(((a, b), trip))
        for t in range(a, b + 1):
            weight[t] *= trip
    # only count what runs once per permutation: restrict to the sponge loop when there is one
    sponge = [ab for ab, tr in trips if tr == 1.0]
    lo, hi = (sponge[-1] if sponge else (0, len(instrs) - 1))
    cls = collections.Counter()
    ops = collections.Counter()
    for k in range(lo, hi + 1):
        op = instrs[k].split()[0]
        if not op.startswith("v_") or weight[k] == 0:
            continue
        base = re.sub(r"_e(32|64)$", "", op)
        ops[base] += weight[k]
        cls["2clk" if (base in TWO_CLK and "dpp" not in instrs[k]) else "4clk"] += weight[k]
    total = cls["2clk"] + cls["4clk"]
    print(json.dumps({
        "kernel": name, "loops": [{"instrs": b - a, "trip": tr} for (a, b), tr in trips],
        "valu_instructions_per_permutation_static": round(total),
        "frac_2clk": round(cls["2clk"] / total, 4), "frac_4clk": round(cls["4clk"] / total, 4),
        "top_opcodes": {k: round(v) for k, v in ops.most_common(10)},
    }, indent=1))


if __name__ == "__main__":
    main()
