#!/usr/bin/env python3
"""Algorithmic VALU floors of the two hot kernels, derived from the ROUND STRUCTURE (not from the compiled code): how many vector
instructions one Poseidon permutation and one NTT element NEED on gfx950 if every instruction does useful arithmetic and nothing is
spent on moving data between register pairs -- the number `bench.py` prints beside the measured counts (`valu.algorithmic_floor_instr`,
`valu.frac_of_floor`, `roofline_lde.valu`), so that "the issue port is full" (valu.frac) and "what fills it is needed" are two figures.

Cost model (one entry per primitive; every VALU instruction of the 4-clock class costs the same issue slot -- DESIGN.md "integer issue
rates" -- so only COUNTS matter):
  pp        1   a 32 x 32 -> 64 partial product with a 64-bit addend: one v_mad_u64_u32 (the only multiplier the ISA has)
  add64     1   a 64-bit add without carry-out (v_lshl_add_u64)
  reduce128 5   128-bit (hi:lo) -> weakly reduced 64-bit: lo + hl * (2^32 - 1) as one multiply-add with carry-out, the carry's wrap
                correction (select + add), the subtraction of hh as a two-instruction borrow chain; the borrow's ~2^-32 correction is
                behind a wave-uniform branch and costs nothing here
  modadd    3   a + b mod p: add, compare, corrective add (weakly reduced operands)
  modsub    3   a - b mod p
  shiftmul  4   x * 2^k mod p (the radix-16 butterflies' twiddles): two shifts, the cross-word subtraction, one correction
A compiled modular multiply is 16 instructions against this model's 4 pp + reduce128 = 9: the difference is the five zero-extending
v_mov_b32 a chained schoolbook product needs on an ISA whose 64-bit operands are aligned register PAIRS, and the two adds that merge the
cross products -- data movement the algorithm does not ask for, but which no instruction sequence found in three rounds avoids
(DESIGN.md section 4 "Poseidon").  The floor is therefore a LOWER BOUND ON COUNT, not a claim that code reaching it exists.
"""
import json
import sys

PP, ADD64, REDUCE128, MODADD, MODSUB, SHIFTMUL = 1, 1, 5, 3, 3, 4
MODMUL = 4 * PP + REDUCE128


def poseidon_floor():
    """t = 12, R_F = 8, R_P = 22, x^7 (poseidon_g_executor.cpp:174-205); MDS entries < 2^6 (hpp:33-50)."""
    full_rounds, partial_rounds, t = 8, 22, 12
    sbox = 4 * MODMUL                                           # x^2, x^3, x^4, x^7
    sboxes = full_rounds * t + partial_rounds                   # 118 S-boxes = 472 modular multiplies
    # full-round MDS on 32-bit halves: 12 terms x 2 halves per output (constants < 2^6: a half's 12-term sum stays below 2^42), the
    # next round's constant rides in an accumulator's initial value; a row closes with acc_hi * 2^32 + acc_lo reduced once
    mds_row = 2 * t * PP + REDUCE128
    mds_full = full_rounds * t * mds_row
    # partial rounds in the grouped sparse form (tools/gen_poseidon_sparse.py section 3; two groups of 11 rounds): per group 11 * 11 terms
    # D.z, 55 terms C.y, 11 * 11 terms W.y, group 0 also 11 * 11 terms PRE.z; a term is a 64-bit constant times a 64-bit value = 4 partial
    # products; 11 + 11 dot products are closed per group (one reduction each: twelve accumulator words gathered -- 6 multiply-adds at the
    # limb offsets -- and folded)
    terms = 2 * (121 + 55 + 121) + 121 + 2 * 11                 # + M00 * y_r, two halves each counted as one term of 2 pp below
    dot_term = 4 * PP
    closings = 2 * (11 + 11)
    dot_close = 6 * PP + 2 * ADD64 + REDUCE128
    partial_linear = (terms - 22) * dot_term + 22 * 2 * PP + closings * dot_close
    adds = (t + t) * MODADD                                     # the first and the 26th round's constants (all others are folded)
    total = sboxes * sbox + mds_full + partial_linear + adds
    return {"modular_multiplies": sboxes * 4, "sbox_instr": sboxes * sbox, "full_round_mds_instr": mds_full, "partial_round_linear_instr": partial_linear,
            "dot_terms": terms, "dot_closings": closings, "constant_adds_instr": adds, "floor_instr_per_permutation": total,
            "model": {"pp": PP, "add64": ADD64, "reduce128": REDUCE128, "modmul": MODMUL, "modadd": MODADD}}


def ntt_floor(log_r=8):
    """One Stockham pass of radix 2^log_r as csrc/ntt.hip runs it: two radix-16 steps (4 layers of radix-2 butterflies each, twiddles inside
    a step are powers of two: 34 non-trivial shift twiddles per 64 butterfly pairs of a 16-point DFT x 2 steps), one general twiddle
    multiply per element between the steps and one between passes.  Per ELEMENT and pass."""
    layers = log_r
    butterflies = layers / 2.0                                  # n/2 butterflies a layer
    shift_tw = 34.0 / 16.0 * (log_r / 4) / 2.0                  # 34 shift twiddles per 16-point DFT (per 16 elements), log_r / 4 DFT steps... per element
    general = 2.0
    per_elem = butterflies * (MODADD + MODSUB) + shift_tw * SHIFTMUL + general * MODMUL
    return {"butterflies_per_element": butterflies, "shift_twiddles_per_element": shift_tw, "general_multiplies_per_element": general,
            "floor_instr_per_element_pass": per_elem, "model": {"butterfly": MODADD + MODSUB, "shiftmul": SHIFTMUL, "modmul": MODMUL}}


def lde_floor(passes_intt=3, passes_ntt=3, blowup=2):
    """LDE N -> blowup * N as 2^23 -> 2^24 runs: INTT passes over N elements, NTT passes over blowup * N (the fused middle pass computes
    both of its halves); + the scale shift^k / N multiply per coefficient.  Per INPUT element."""
    p = ntt_floor()["floor_instr_per_element_pass"]
    return {"floor_instr_per_input_element": passes_intt * p + blowup * passes_ntt * p + MODMUL, "per_element_pass": p}


def main():
    print(json.dumps({"poseidon": poseidon_floor(), "ntt_pass": ntt_floor(), "lde": lde_floor()}, indent=1))


if __name__ == "__main__":
    sys.exit(main())
