"""A complete small STARK, end to end (TEST INFRASTRUCTURE).

The reference's full proof run (BASELINE config 4) needs inputs that are not in the tree, and an earlier benchmark script ran the device phases
on programs that are shaped like the zkEVM's but do not describe a satisfiable system -- its q is not a low-degree polynomial and no
verifier would accept the result.  This module closes that gap at small size: an AIR with a real witness, its two constraint programs
written in the reference's own opcode formats (step42ns, step52ns), a PROVER that strings the product's device entry points together in
`Starks::genProof`'s order (starks.cpp:9-403: commit, constraint polynomial, quotient split, evaluations, FRI polynomial, FRI, queries),
and an independent VERIFIER on the oracle's arithmetic (pil-stark's stark_verify: transcript replay, constraint identity at the
challenge point, Merkle openings, the FRI polynomial recomputed at the query points from the opened rows, fold consistency, degree of
the final polynomial).  A proof made on the GPU must verify; a tampered one must not.

AIR (N rows, columns a, b, c, d, f; constant polynomials L1 = first-row selector, LLAST = last-row selector, T = a lookup table):
    (1 - LLAST) * (a' - b) = 0,   (1 - LLAST) * (b' - a - b) = 0,   L1 * (a - 1) = 0,   L1 * (b - 1) = 0        (x' = value at the next row)
and a second stage like the reference's stage 2: after the first commitment a challenge gamma is drawn, an extension-valued column
    z = (a + gamma) * (b + gamma)
is computed ON THE DEVICE by a base-domain program (the step2prev / step3prev / step3 opcode numbering: results stored into polynomial
memory), extended from device memory, committed, and bound by a fifth constraint  (a + gamma) * (b + gamma) - z = 0;
and a third stage like the reference's stage 3 -- a permutation argument: d is a permutation of c; after the second commitment beta is
drawn, a base-domain program writes c + beta and d + beta into tmpExp_n, the grand product
    p[0] = 1,  p[i+1] = p[i] * (c[i] + beta) / (d[i] + beta)
is computed ON THE DEVICE (mi_calculate_z_dev = Polinomial::calculateZ), extended, committed, and bound by
    p' * (d + beta) - p * (c + beta) = 0   (every row: the wrap-around row holds because the product closes),    L1 * (p - 1) = 0;
and a lookup (plookup, as pil-stark arithmetises it): every f[i] is a row of T.  In stage 2 the sorted columns h1, h2 are computed ON
THE DEVICE (mi_calculate_h1h2_dev = Polinomial::calculateH1H2*) and committed beside z; after that commitment gamma2, beta2 are drawn
and stage 3 carries a second grand product p2 of
    num = (1 + beta2) (gamma2 + f) (gamma2 (1 + beta2) + T + beta2 T'),    den = (gamma2 (1 + beta2) + h1 + beta2 h2) (gamma2 (1 + beta2) + h2 + beta2 h1'),
bound by  p2' * den - p2 * num = 0  and  L1 * (p2 - 1) = 0.  The product closes only if (h1[0], h2[0], h1[1], ...) is f u T sorted
along T -- which is exactly what calculateH1H2 has to deliver, so a verifying proof pins its output convention.
"""
import numpy as np

import glo
import chelpers_programs as cp

P = 0xFFFFFFFF00000001
SHIFT = 49
# Challenge slots as Starks::genProof draws them (starks.cpp:67-68, 149-150, 233, 305, 340-341): [0], [1] after root1, [2], [3] after
# root2, [4] after root3, [7] after root4, [5], [6] after the evaluations.  This AIR: gamma (stage-2 column) and beta (permutation:
# c, d are stage-1 columns) come after root1; the lookup's gamma2, beta2 after root2 (h1, h2 are stage-2 columns); vc, xi, v1, v2.
GAMMA, BETA, GAMMA2, BETA2, VC, V1, V2, XI = 0, 1, 2, 3, 4, 5, 6, 7
N_CHALLENGES = 8
PUBLICS = np.array([1, 1], dtype=np.uint64)              # a[0], b[0]: bound by L1 * (a - publics[0]), L1 * (b - publics[1])


# ------------------------------------------------------------------ extension-field helpers (python ints; x^3 = x + 1)
def e3(v):
    return [int(v[0]) % P, int(v[1]) % P, int(v[2]) % P]


def e3_add(a, b):
    return [(a[i] + b[i]) % P for i in range(3)]


def e3_sub(a, b):
    return [(a[i] - b[i]) % P for i in range(3)]


def e3_mul(a, b):
    return [int(v) for v in glo.e3_mul(np.array(a, dtype=np.uint64), np.array(b, dtype=np.uint64))]


def e3_from_base(v):
    return [int(v) % P, 0, 0]


def e3_pow(a, e):
    r = [1, 0, 0]
    while e:
        if e & 1:
            r = e3_mul(r, a)
        a = e3_mul(a, a)
        e >>= 1
    return r


# ------------------------------------------------------------------ memory map: the reference's (stark_info mapOffsets / mapSectionsN)
class Layout:
    """Sections of the polynomial area in the reference's order: cm1_n | cm2_n | cm3_n | cm4_n | tmpExp_n over n rows,
    then cm1_2ns | cm2_2ns | cm3_2ns | cm4_2ns | q_2ns | f_2ns over n_ext rows; element (row, col) of a section at offset + row * cols
    + col.  Layout(0, 1) is one extended row on its own (the verifier evaluates the FRI-polynomial program over opened rows)."""
    COLS = {"cm1_n": 5, "cm2_n": 5, "cm3_n": 6, "cm4_n": 6, "tmpExp_n": 14, "cm1_2ns": 5, "cm2_2ns": 5, "cm3_2ns": 6, "cm4_2ns": 6, "q_2ns": 3, "f_2ns": 3}
    ORDER = ["cm1_n", "cm2_n", "cm3_n", "cm4_n", "tmpExp_n", "cm1_2ns", "cm2_2ns", "cm3_2ns", "cm4_2ns", "q_2ns", "f_2ns"]

    def __init__(self, n, n_ext):
        self.n, self.ne = n, n_ext
        self.off, o = {}, 0
        for name in self.ORDER:
            self.off[name] = o
            o += self.COLS[name] * (n_ext if name.endswith("2ns") else n)
        self.total = o

    def base_sections(self):
        return [(self.off[k], self.COLS[k], self.n) for k in ("cm1_n", "cm2_n", "cm3_n", "tmpExp_n")]

    def ext_sections(self, with_cm4=False):
        return [(self.off[k], self.COLS[k], self.ne) for k in ("cm1_2ns", "cm2_2ns", "cm3_2ns") + (("cm4_2ns",) if with_cm4 else ())]


# tmpExp_n columns: numerator / denominator of the permutation product (0..2, 3..5), of the lookup product (6..8, 9..11), and the
# lookup's f and t expressions as polynomials (12, 13: what pil-stark's exp2pol names)
TE_PNUM, TE_PDEN, TE_LNUM, TE_LDEN, TE_F, TE_T = 0, 3, 6, 9, 12, 13


# ------------------------------------------------------------------ the AIR's programs in the reference's table formats
def step42ns_program(lay, next_shift=2):
    """q * Z_H = Horner_vc(C1 .. C9) (challenge slots: GAMMA .. VC above; C3, C4 read the public inputs); sections: cm1_2ns (a, b, c, d,
    f), cm2_2ns (z extension-valued, h1, h2), cm3_2ns (p, p2 extension-valued) at the layout's offsets."""
    n_ext = lay.ne
    o1, o2, o3 = lay.off["cm1_2ns"], lay.off["cm2_2ns"], lay.off["cm3_2ns"]
    A_, B_, C_, D_, F_, ST = o1, o1 + 1, o1 + 2, o1 + 3, o1 + 4, 5     # columns of cm1_2ns, its row stride
    H1_, H2_, ST2, ST3 = o2 + 3, o2 + 4, 5, 6
    L1, LLAST, T_ = 0, 1, 2                   # constant polynomials
    ops, args = [], []

    def push(o, ar):
        ops.append(o); args.extend(ar)
    push(35, [0, A_, next_shift, n_ext, ST, B_, ST])      # t0 = a' - b
    push(32, [1, 1, LLAST])                               # t1 = 1 - LLAST
    push(45, [0, 0, 1])                                   # t0 = t0 * t1
    push(59, [0, 0, VC])                                  # acc = t0 * vc
    push(35, [2, B_, next_shift, n_ext, ST, A_, ST])      # t2 = b' - a
    push(22, [2, 2, B_, ST])                              # t2 = t2 - b
    push(45, [2, 2, 1])                                   # t2 = t2 * t1
    push(84, [0, 2, 0, 0, VC, 0])                         # acc = (t2 + acc) * vc
    push(34, [3, A_, ST, 0])                              # t3 = a - publics[0]
    push(49, [3, 3, L1])                                  # t3 = t3 * L1
    push(84, [0, 3, 0, 0, VC, 0])                         # acc = (t3 + acc) * vc
    push(34, [4, B_, ST, 1])                              # t4 = b - publics[1]
    push(49, [4, 4, L1])                                  # t4 = t4 * L1
    push(12, [0, 4, 0])                                   # acc = t4 + acc
    push(16, [1, A_, ST, GAMMA])                          # e1 = a + gamma
    push(16, [2, B_, ST, GAMMA])                          # e2 = b + gamma
    push(71, [1, 1, 2])                                   # e1 = e1 * e2
    push(44, [1, 1, o2, ST2])                             # e1 = e1 - z               (z: cm2_2ns, three columns)
    push(70, [0, VC, 0])                                  # acc = vc * acc
    push(17, [0, 1, 0])                                   # acc = e1 + acc
    push(16, [1, D_, ST, BETA])                           # e1 = d + beta
    push(74, [1, o3, next_shift, n_ext, ST3, 1])          # e1 = p' * e1              (p: cm3_2ns, at the next row)
    push(16, [2, C_, ST, BETA])                           # e2 = c + beta
    push(75, [2, o3, ST3, 2])                             # e2 = p * e2
    push(42, [1, 1, 2])                                   # e1 = e1 - e2
    push(70, [0, VC, 0])
    push(17, [0, 1, 0])                                   # acc = vc * acc + e1
    push(41, [1, o3, ST3, 1])                             # e1 = p - 1
    push(60, [1, L1, 1])                                  # e1 = L1 * e1
    push(70, [0, VC, 0])
    push(17, [0, 1, 0])                                   # acc = vc * acc + e1
    # the lookup's grand product p2 (cm3_2ns columns 3..5): p2' * den - p2 * num
    push(13, [3, 1, BETA2])                               # e3 = 1 + beta2
    push(70, [4, GAMMA2, 3])                              # e4 = gamma2 * e3
    push(82, [5, T_])                                     # t5 = T
    push(83, [6, T_, next_shift, n_ext])                  # t6 = T'
    push(59, [1, 6, BETA2])                               # e1 = t6 * beta2
    push(12, [1, 5, 1])                                   # e1 = t5 + e1
    push(17, [1, 1, 4])                                   # e1 = e1 + e4
    push(16, [2, F_, ST, GAMMA2])                         # e2 = f + gamma2
    push(71, [2, 2, 3])                                   # e2 = e2 * e3
    push(71, [2, 2, 1])                                   # e2 = e2 * e1              = num
    push(75, [2, o3 + 3, ST3, 2])                         # e2 = p2 * e2
    push(62, [1, H2_, ST2, BETA2])                        # e1 = h2 * beta2
    push(15, [1, H1_, ST2, 1])                            # e1 = h1 + e1
    push(17, [1, 1, 4])                                   # e1 = e1 + e4
    push(63, [5, H1_, next_shift, n_ext, ST2, BETA2])     # e5 = h1' * beta2
    push(15, [5, H2_, ST2, 5])                            # e5 = h2 + e5
    push(17, [5, 5, 4])                                   # e5 = e5 + e4
    push(71, [1, 1, 5])                                   # e1 = e1 * e5              = den
    push(74, [1, o3 + 3, next_shift, n_ext, ST3, 1])      # e1 = p2' * e1
    push(42, [1, 1, 2])                                   # e1 = e1 - e2
    push(70, [0, VC, 0])
    push(17, [0, 1, 0])                                   # acc = vc * acc + e1
    push(41, [1, o3 + 3, ST3, 1])                         # e1 = p2 - 1
    push(60, [1, L1, 1])                                  # e1 = L1 * e1
    push(70, [0, VC, 0])
    push(17, [0, 1, 0])                                   # acc = vc * acc + e1
    push(69, [0])                                         # q = zhInv * acc
    return np.array(ops, dtype=np.uint64), np.array(args, dtype=np.uint64)


EV_A, EV_B, EV_AW, EV_BW, EV_L1, EV_LLAST, EV_Q0, EV_Q1, EV_Z, EV_C, EV_D, EV_GP, EV_GPW, EV_F, EV_T, EV_TW, EV_H1, EV_H2, EV_H1W, EV_P2, EV_P2W = range(21)


def step52ns_program(lay):
    """f = ((H c5 + E_xi xDivXSubXi) c5 + E_wxi xDivXSubWXi), H = Horner_c5(every committed column), E_* = Horner_c6(pol - eval);
    sections: cm1_2ns (5 columns), cm2_2ns (z, h1, h2: 5 columns), cm3_2ns (p, p2: 6 columns), cm4_2ns (two extension-valued chunks,
    6 columns) at the layout's offsets; constants L1, LLAST, T."""
    o1, o2, o3, o4 = lay.off["cm1_2ns"], lay.off["cm2_2ns"], lay.off["cm3_2ns"], lay.off["cm4_2ns"]
    ops, args = [], []

    def push(o, ar):
        ops.append(o); args.extend(ar)
    push(0, [o1, 5]); push(16, [o1 + 1, 5]); push(16, [o1 + 2, 5]); push(16, [o1 + 3, 5]); push(16, [o1 + 4, 5])            # H
    push(17, [o2, 5]); push(16, [o2 + 3, 5]); push(16, [o2 + 4, 5]); push(17, [o3, 6]); push(17, [o3 + 3, 6])
    push(17, [o4, 6]); push(17, [o4 + 3, 6])
    push(3, [])                                                                       # tmp1 = H c5
    push(11, [o1, 5, EV_A]); push(4, [])                                              # tmp = (a - a(xi)) c6
    push(18, [o1 + 1, 5, EV_B]); push(18, [o1 + 2, 5, EV_C]); push(18, [o1 + 3, 5, EV_D]); push(18, [o1 + 4, 5, EV_F])
    push(19, [0, EV_L1]); push(19, [1, EV_LLAST]); push(19, [2, EV_T])
    push(20, [o4, 6, EV_Q0]); push(20, [o4 + 3, 6, EV_Q1]); push(20, [o2, 5, EV_Z]); push(18, [o2 + 3, 5, EV_H1]); push(18, [o2 + 4, 5, EV_H2])
    push(20, [o3, 6, EV_GP]); push(20, [o3 + 3, 6, EV_P2])
    push(5, []); push(8, []); push(3, [])                                             # * xDivXSubXi; tmp = tmp1 + tmp; tmp1 = tmp c5
    push(11, [o1, 5, EV_AW]); push(4, []); push(18, [o1 + 1, 5, EV_BW]); push(20, [o3, 6, EV_GPW])
    push(19, [2, EV_TW]); push(18, [o2 + 3, 5, EV_H1W]); push(20, [o3 + 3, 6, EV_P2W])
    push(6, []); push(8, []); push(15, [])                                            # * xDivXSubWXi; tmp = tmp1 + tmp; f = tmp
    return np.array(ops, dtype=np.uint64), np.array(args, dtype=np.uint64)


def stage2_program(lay):
    """step2prev: z = (a + gamma) * (b + gamma) into cm2_n, and -- as pil-stark's step2prev does for every lookup -- the lookup's f and t
    expressions materialised as polynomials in tmpExp_n (f = column 4 of cm1_n, t = constant polynomial 2), where calculateH1H2 finds them
    through exp2pol.  Base-domain steps' numbering."""
    o1, o2, oT, STT = lay.off["cm1_n"], lay.off["cm2_n"], lay.off["tmpExp_n"], lay.COLS["tmpExp_n"]
    ops = [16, 16, 98, 79, 100, 82, 100]
    args = [0, o1, 5, GAMMA,   1, o1 + 1, 5, GAMMA,   o2, 5, 0, 1,   0, o1 + 4, 5,   oT + TE_F, STT, 0,   1, 2,   oT + TE_T, STT, 1]
    return np.array(ops, dtype=np.uint64), np.array(args, dtype=np.uint64)


def stage3_program(lay):
    """step3prev: numerators / denominators of the two grand products into tmpExp_n (TE_PNUM = c + beta, TE_PDEN = d + beta, TE_LNUM /
    TE_LDEN the lookup's); T = constant polynomial 2; h1, h2 = columns 3, 4 of cm2_n."""
    n = lay.n
    o1, o2, oT, STT = lay.off["cm1_n"], lay.off["cm2_n"], lay.off["tmpExp_n"], lay.COLS["tmpExp_n"]
    ops, args = [], []

    def push(o, ar):
        ops.append(o); args.extend(ar)
    push(13, [0, 0, BETA])                                # e0 = 0 + beta
    push(79, [0, o1 + 2, 5]); push(88, [oT + TE_PNUM, STT, 0, 0])        # t0 = c;  tmpExp[0..2] = t0 + e0
    push(79, [1, o1 + 3, 5]); push(88, [oT + TE_PDEN, STT, 1, 0])        # t1 = d;  tmpExp[3..5] = t1 + e0
    push(13, [3, 1, BETA2])                               # e3 = 1 + beta2
    push(70, [4, GAMMA2, 3])                              # e4 = gamma2 * e3
    push(82, [5, 2]); push(83, [6, 2, 1, n])              # t5 = T;  t6 = T'
    push(59, [1, 6, BETA2]); push(12, [1, 5, 1]); push(17, [1, 1, 4])    # e1 = T + beta2 T' + e4
    push(16, [2, o1 + 4, 5, GAMMA2]); push(71, [2, 2, 3])                # e2 = (f + gamma2) * e3
    push(98, [oT + TE_LNUM, STT, 2, 1])                   # tmpExp[6..8] = e2 * e1
    push(62, [1, o2 + 4, 5, BETA2]); push(15, [1, o2 + 3, 5, 1]); push(17, [1, 1, 4])      # e1 = h1 + beta2 h2 + e4
    push(63, [5, o2 + 3, 1, n, 5, BETA2]); push(15, [5, o2 + 4, 5, 5]); push(17, [5, 5, 4])  # e5 = h2 + beta2 h1' + e4
    push(98, [oT + TE_LDEN, STT, 1, 5])                   # tmpExp[9..11] = e1 * e5
    return np.array(ops, dtype=np.uint64), np.array(args, dtype=np.uint64)


def step3_program(lay):
    """step3: nothing left to compute in this AIR once the grand products are there (pil-stark's step3 holds the intermediate
    polynomials of stage 3).  One harmless store so that the step is a program like the others: tmpExp[TE_F] = f again."""
    o1, oT, STT = lay.off["cm1_n"], lay.off["tmpExp_n"], lay.COLS["tmpExp_n"]
    return np.array([79, 100], dtype=np.uint64), np.array([0, o1 + 4, 5,   oT + TE_F, STT, 0], dtype=np.uint64)


# ------------------------------------------------------------------ the STARK as pil-stark would describe it: <name>.starkinfo.json
# polynomial ids (varPolMap): the witness, the stage-2 and stage-3 columns, the expression polynomials, then their extensions
POL_A, POL_B, POL_C, POL_D, POL_F, POL_Z, POL_H1, POL_H2, POL_P, POL_P2, POL_PNUM, POL_PDEN, POL_LNUM, POL_LDEN, POL_FEXP, POL_TEXP = range(16)
EXP_F, EXP_T, EXP_LNUM, EXP_LDEN, EXP_PNUM, EXP_PDEN = 100, 101, 102, 103, 104, 105   # expression ids (keys of exp2pol)


def starkinfo(nbits, n_queries=12):
    """The mini STARK in the format of pil-stark's starkinfo.json -- what Starks::Starks loads (stark_info.cpp:20-447)."""
    nbits_ext = nbits + 1
    lay = Layout(1 << nbits, 1 << nbits_ext)
    base = [("cm1_n", 1, 0), ("cm1_n", 1, 1), ("cm1_n", 1, 2), ("cm1_n", 1, 3), ("cm1_n", 1, 4), ("cm2_n", 3, 0), ("cm2_n", 1, 3), ("cm2_n", 1, 4),
            ("cm3_n", 3, 0), ("cm3_n", 3, 3), ("tmpExp_n", 3, TE_PNUM), ("tmpExp_n", 3, TE_PDEN), ("tmpExp_n", 3, TE_LNUM), ("tmpExp_n", 3, TE_LDEN),
            ("tmpExp_n", 1, TE_F), ("tmpExp_n", 1, TE_T)]
    ext = [(s.replace("_n", "_2ns"), d, p) for (s, d, p) in base[:10]] + [("cm4_2ns", 3, 0), ("cm4_2ns", 3, 3), ("q_2ns", 3, 0), ("f_2ns", 3, 0)]
    vpm = [{"section": s, "dim": d, "sectionPos": p} for (s, d, p) in base + ext]
    E = len(base)                                        # id of the first extended polynomial
    # committed polynomials in genProof's order: stage 1, then the lookups' h1 / h2, then the grand products (lookups, permutations,
    # connections), then whatever else a stage committed (here the stage-2 column z)
    cm_n = [POL_A, POL_B, POL_C, POL_D, POL_F, POL_H1, POL_H2, POL_P2, POL_P, POL_Z]
    cm_2ns = [E + i for i in cm_n]
    CM = {pol: k for k, pol in enumerate(cm_n)}          # polynomial id -> index into cm_n / cm_2ns
    ev = [None] * 21
    def cm(k, pol, prime=False): ev[k] = {"type": "cm", "id": CM[pol], "prime": prime}
    def const(k, i, prime=False): ev[k] = {"type": "const", "id": i, "prime": prime}
    cm(EV_A, POL_A); cm(EV_B, POL_B); cm(EV_AW, POL_A, True); cm(EV_BW, POL_B, True); const(EV_L1, 0); const(EV_LLAST, 1)
    ev[EV_Q0] = {"type": "q", "id": 0, "prime": False}; ev[EV_Q1] = {"type": "q", "id": 1, "prime": False}
    cm(EV_Z, POL_Z); cm(EV_C, POL_C); cm(EV_D, POL_D); cm(EV_GP, POL_P); cm(EV_GPW, POL_P, True); cm(EV_F, POL_F); const(EV_T, 2); const(EV_TW, 2, True)
    cm(EV_H1, POL_H1); cm(EV_H2, POL_H2); cm(EV_H1W, POL_H1, True); cm(EV_P2, POL_P2); cm(EV_P2W, POL_P2, True)
    sec = lambda f: {k: f(k) for k in Layout.ORDER}
    return {
        "starkStruct": {"nBits": nbits, "nBitsExt": nbits_ext, "nQueries": n_queries, "verificationHashType": "GL",
                        "steps": [{"nBits": b} for b in fri_steps(nbits_ext)]},
        "mapTotalN": lay.total, "nConstants": 3, "nPublics": int(PUBLICS.size), "nCm1": 5, "nCm2": 3, "nCm3": 2, "nCm4": 2, "qDeg": 2, "qDim": 3,
        "friExpId": 200, "nExps": 201,
        "mapDeg": sec(lambda k: (1 << nbits_ext) if k.endswith("2ns") else (1 << nbits)),
        "mapOffsets": sec(lambda k: lay.off[k]),
        "mapSections": sec(lambda k: [i for i, v in enumerate(vpm) if v["section"] == k]),
        "mapSectionsN": sec(lambda k: Layout.COLS[k]),
        "mapSectionsN1": sec(lambda k: sum(1 for v in vpm if v["section"] == k and v["dim"] == 1)),
        "mapSectionsN3": sec(lambda k: sum(1 for v in vpm if v["section"] == k and v["dim"] == 3)),
        "varPolMap": vpm, "qs": [E + 10, E + 11], "cm_n": cm_n, "cm_2ns": cm_2ns,
        "peCtx": [{"tExpId": 0, "fExpId": 0, "zId": POL_P, "c1Id": 0, "numId": EXP_PNUM, "denId": EXP_PDEN, "c2Id": 0}],
        "puCtx": [{"tExpId": EXP_T, "fExpId": EXP_F, "h1Id": POL_H1, "h2Id": POL_H2, "zId": POL_P2, "c1Id": 0, "numId": EXP_LNUM, "denId": EXP_LDEN, "c2Id": 0}],
        "ciCtx": [], "evMap": ev,
        "exp2pol": {str(EXP_F): POL_FEXP, str(EXP_T): POL_TEXP, str(EXP_LNUM): POL_LNUM, str(EXP_LDEN): POL_LDEN, str(EXP_PNUM): POL_PNUM,
                    str(EXP_PDEN): POL_PDEN},
    }


def proof_from_zkin(z, nbits):
    """A zkin.json as host/proof2zkinStark.hpp writes it (proof2zkinStark.cpp:8-82) -> the proof dictionary verify() reads."""
    U = lambda x: np.array(x, dtype=object).astype(np.uint64)
    steps = fri_steps(nbits + 1)
    op = lambda t: np.concatenate([U(z["s0_vals" + t]).reshape(len(z["s0_vals" + t]), -1), U(z["s0_siblings" + t]).reshape(len(z["s0_vals" + t]), -1)], axis=1)
    proof = {"nbits": nbits, "publics": U(z["publics"]), "evals": U(z["evals"]).reshape(-1), "final_pol": U(z["finalPol"]).reshape(-1),
             "fri_roots": [U(z["s%d_root" % i]) for i in range(1, len(steps))],
             "s0": {"cm1": op("1"), "cm2": op("2"), "cm3": op("3"), "cm4": op("4"), "const": op("C")}, "fri": {}}
    for k in ("root1", "root2", "root3", "root4"):
        proof[k] = U(z[k])
    for i in range(1, len(steps)):
        v = U(z["s%d_vals" % i]); sb = U(z["s%d_siblings" % i])
        proof["fri"][i] = np.concatenate([v.reshape(v.shape[0], -1), sb.reshape(sb.shape[0], -1)], axis=1)
    return proof


def witness(n):
    out = np.empty((n, 5), dtype=np.uint64)
    a = b = 1
    for i in range(n):
        out[i, 0], out[i, 1] = a, b
        a, b = b, (a + b) % P
    rng = np.random.default_rng(n)
    out[:, 2] = glo.rand_fe(rng, (n,))
    out[:, 3] = out[rng.permutation(n), 2]            # d: a permutation of c
    idx = rng.integers(0, n, size=n)
    idx[: n // 3] = idx[0]                            # (one row of the table takes a third of the lookups)
    out[:, 4] = constants(n)[idx, 2]                  # f: rows of the table T
    return out


def constants(n):
    c = np.zeros((n, 3), dtype=np.uint64)
    c[0, 0] = 1
    c[n - 1, 1] = 1
    i = np.arange(n, dtype=np.uint64)
    c[:, 2] = (i * i) % np.uint64(97) + (i >> np.uint64(3)) * np.uint64(1000)   # the table T: repeated values, adjacent and not
    return c


def fri_steps(nbits_ext):
    steps = [nbits_ext]
    for d in (4, 4):
        if steps[-1] - d >= 3:
            steps.append(steps[-1] - d)
    return steps


class Transcript:
    """transcript.cpp:4-87 -- host state machine, every permutation on the GPU (mi_poseidon_hash_full_result): the mini prover's transcript."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.state = np.zeros(4, dtype=np.uint64)
        self.pending = np.zeros(8, dtype=np.uint64)
        self.out = np.zeros(12, dtype=np.uint64)
        self.pending_cursor = 0
        self.out_cursor = 0

    def _update(self):
        self.out = self.ctx.hash_full_result(np.concatenate([self.pending, self.state]))
        self.out_cursor = 12
        self.pending[:] = 0
        self.pending_cursor = 0
        self.state = self.out[:4].copy()

    def put(self, vals):
        for v in np.asarray(vals, dtype=np.uint64).ravel():
            self.pending[self.pending_cursor] = v
            self.pending_cursor += 1
            self.out_cursor = 0
            if self.pending_cursor == 8:
                self._update()

    def get_fields1(self):
        if self.out_cursor == 0:
            self._update()
        r = int(self.out[(12 - self.out_cursor) % 12])
        self.out_cursor -= 1
        return r

    def get_field(self):
        return np.array([self.get_fields1() for _ in range(3)], dtype=np.uint64)

    def get_permutations(self, n, nbits):
        nfields = (n * nbits - 1) // 63 + 1
        fields = [self.get_fields1() for _ in range(nfields)]
        res, cur_field, cur_bit = [], 0, 0
        for _ in range(n):
            a = 0
            for j in range(nbits):
                if (fields[cur_field] >> cur_bit) & 1:
                    a += 1 << j
                cur_bit += 1
                if cur_bit == 63:
                    cur_bit, cur_field = 0, cur_field + 1
            res.append(a)
        return np.array(res, dtype=np.uint64)


# ------------------------------------------------------------------ prover: the product's device entry points in genProof's order
def prove(ctx, nbits, n_queries=12, native=False, cache_dir=None, tamper=None):
    """Returns the proof (host data only).  native: the constraint programs through the compiled-kernel backend.  The polynomial area is
    ONE device buffer in the reference's memory map (Layout), the programs address it by absolute offsets."""
    import mi_stark
    L = glo.lib()
    nbits_ext = nbits + 1
    N, NE = 1 << nbits, 1 << nbits_ext
    lay = Layout(N, NE)
    steps = fri_steps(nbits_ext)
    # constant polynomials: extended and committed once (the verification key is their root)
    d_const_n = ctx.to_device(constants(N))
    NC = 3
    const_2ns, const_nodes = ctx.empty(NE * NC), ctx.empty((2 * NE - 1) * 4)
    ctx.lde(const_2ns, d_const_n, NE, N, NC)
    ctx.merkle_build(const_nodes, const_2ns, NC, NE)
    mem = ctx.zeros(lay.total)
    S = lambda k: mem[lay.off[k]:lay.off[k] + Layout.COLS[k] * (NE if k.endswith("2ns") else N)]
    cm1, cm2, cm3, cm4 = S("cm1_2ns"), S("cm2_2ns"), S("cm3_2ns"), S("cm4_2ns")
    nodes1, nodes2, nodes3, nodes4 = (ctx.empty((2 * NE - 1) * 4) for _ in range(4))
    w = witness(N)
    if tamper == "perm":                                 # d is no longer a permutation of c: the grand product does not close
        w[N // 2, 3] = (int(w[N // 2, 3]) + 1) % P
    if tamper == "lookup":                               # a value the table does not hold
        w[N // 2, 4] = 5
    S("cm1_n")[:] = ctx.to_device(w).reshape(-1)
    pub = PUBLICS.copy()
    tr = Transcript(ctx)
    tr.put(pub)                                          # starks.cpp:28
    chal = np.zeros(N_CHALLENGES * 3, dtype=np.uint64)
    C = lambda k: slice(3 * k, 3 * k + 3)
    # ---- step 1: commit the witness
    ctx.lde(cm1, S("cm1_n"), NE, N, 5)
    ctx.merkle_build(nodes1, cm1, 5, NE)
    root1 = ctx.to_host(nodes1[-4:])
    tr.put(root1)
    chal[C(GAMMA)] = tr.get_field()                      # challenges [0], [1]
    chal[C(BETA)] = tr.get_field()
    # ---- step 2: the stage-2 column from a base-domain program (compiled kernels), extended from device memory, committed
    def base_prog(gen, step):
        ops, args = gen(lay)
        pr = mi_stark.ChelpersProgram(ctx, ops, args, sections=lay.base_sections(), n_const=NC, nrows_ext=N, step=step)
        pr.build_native(cache_dir=cache_dir)
        return pr
    progb = base_prog(stage2_program, mi_stark.MI_CHELPERS_STEP2PREV)
    x_n = ctx.empty(N)
    ctx.geom_seq(x_n, N, 1, L.glo_w(nbits))
    progb.run_base(mem, d_const_n, NC, chal, pub, x_n, 1, 0, N)
    # the lookup's sorted columns (starks.cpp:92-128): h1, h2 = columns 3, 4 of cm2_n from the f and t expression polynomials in tmpExp_n
    o2, oT, STT = lay.off["cm2_n"], lay.off["tmpExp_n"], Layout.COLS["tmpExp_n"]
    ctx.calculate_h1h2(mem[o2 + 3:], 5, mem[o2 + 4:], 5, mem[oT + TE_F:], STT, mem[oT + TE_T:], STT, 1, N)
    if tamper == "h1h2":                                 # two neighbours of the sorted sequence swapped
        hb = ctx.to_host(S("cm2_n")).reshape(N, 5)
        k = next(i for i in range(N) if hb[i, 3] != hb[i, 4])
        hb[k, 3], hb[k, 4] = hb[k, 4], hb[k, 3]
        S("cm2_n")[:] = ctx.to_device(hb.reshape(-1))
    ctx.lde(cm2, S("cm2_n"), NE, N, 5)
    ctx.merkle_build(nodes2, cm2, 5, NE)
    root2 = ctx.to_host(nodes2[-4:])
    tr.put(root2)
    chal[C(GAMMA2)] = tr.get_field()                     # challenges [2], [3]
    chal[C(BETA2)] = tr.get_field()
    # ---- step 3: numerators / denominators from a base-domain program, the grand products on the device, extended, committed
    progc = base_prog(stage3_program, mi_stark.MI_CHELPERS_STEP3PREV)
    progc.run_base(mem, d_const_n, NC, chal, pub, x_n, 1, 0, N)
    o3 = lay.off["cm3_n"]
    closes2 = ctx.calculate_z(mem[o3 + 3:], 6, mem[oT + TE_LNUM:], STT, mem[oT + TE_LDEN:], STT, N)     # lookups first (starks.cpp:473-536)
    closes = ctx.calculate_z(mem[o3:], 6, mem[oT + TE_PNUM:], STT, mem[oT + TE_PDEN:], STT, N)
    assert closes == (tamper != "perm") and closes2 == (tamper != "h1h2")   # (the reference zkasserts this; a cheating prover goes on)
    progd = base_prog(step3_program, mi_stark.MI_CHELPERS_STEP3)
    progd.run_base(mem, d_const_n, NC, chal, pub, x_n, 1, 0, N)
    ctx.lde(cm3, S("cm3_n"), NE, N, 6)
    ctx.merkle_build(nodes3, cm3, 6, NE)
    root3 = ctx.to_host(nodes3[-4:])
    tr.put(root3)
    chal[C(VC)] = tr.get_field()                         # challenge [4]
    # ---- step 4: constraint polynomial q = C / Z_H on the extended domain, split, committed
    ops42, args42 = step42ns_program(lay, 2)
    prog42 = mi_stark.ChelpersProgram(ctx, ops42, args42, sections=lay.ext_sections(), n_const=NC, nrows_ext=NE)
    ops52, args52 = step52ns_program(lay)
    prog52 = mi_stark.ChelpersProgram(ctx, ops52, args52, sections=lay.ext_sections(with_cm4=True), n_const=NC, nrows_ext=NE, step=52)
    if native:
        prog42.build_native(cache_dir=cache_dir)
        prog52.build_native(cache_dir=cache_dir)
    x_2ns = ctx.empty(NE)
    ctx.geom_seq(x_2ns, NE, SHIFT, L.glo_w(nbits_ext))
    zh = ctx.zhinv(nbits, nbits_ext)
    q_2ns, qq1, qq2 = S("q_2ns"), ctx.empty(NE * 3), ctx.empty(NE * 6)
    prog42.run(mem, const_2ns, NC, chal, pub, x_2ns, 1, zh, q_2ns, 0, NE)
    ctx.ntt(qq1, q_2ns, NE, 3, inverse=True)
    ctx.q_split(qq2, qq1, N, NE, 2)
    ctx.ntt(cm4, qq2, NE, 6)
    ctx.merkle_build(nodes4, cm4, 6, NE)
    root4 = ctx.to_host(nodes4[-4:])
    tr.put(root4)
    xi = tr.get_field()                                  # challenge [7]
    chal[C(XI)] = xi
    # ---- step 5: evaluations at xi and w xi, then the FRI polynomial
    sinv, wN = L.glo_inv(SHIFT), L.glo_w(nbits)
    xis = np.array([L.glo_mul(int(v), sinv) for v in xi], dtype=np.uint64)
    wxi = np.array([L.glo_mul(int(v), wN) for v in xi], dtype=np.uint64)
    wxis = np.array([L.glo_mul(int(v), sinv) for v in wxi], dtype=np.uint64)
    lev, lpev = ctx.empty(N * 3), ctx.empty(N * 3)
    ctx.geom_seq3(lev, N, xis)
    ctx.geom_seq3(lpev, N, wxis)
    ctx.ntt(lev, lev, N, 3, inverse=True)
    ctx.ntt(lpev, lpev, N, 3, inverse=True)
    pols = [(cm1, 0, 1, 5), (cm1, 1, 1, 5), (cm1, 0, 1, 5), (cm1, 1, 1, 5), (const_2ns, 0, 1, NC), (const_2ns, 1, 1, NC), (cm4, 0, 3, 6), (cm4, 3, 3, 6),
            (cm2, 0, 3, 5), (cm1, 2, 1, 5), (cm1, 3, 1, 5), (cm3, 0, 3, 6), (cm3, 0, 3, 6),
            (cm1, 4, 1, 5), (const_2ns, 2, 1, NC), (const_2ns, 2, 1, NC), (cm2, 3, 1, 5), (cm2, 4, 1, 5), (cm2, 3, 1, 5), (cm3, 3, 3, 6), (cm3, 3, 3, 6)]
    prime = [0, 0, 1, 1, 0, 0, 0, 0, 0, 0, 0, 0, 1,   0, 0, 1, 0, 0, 1, 0, 1]
    d_evals = ctx.empty(len(pols) * 3)
    ctx.evmap(d_evals, pols, prime, lev, lpev, N, 1)
    evals = ctx.to_host(d_evals)
    if tamper == "eval":
        evals[3 * EV_B] ^= np.uint64(1)
    tr.put(evals)
    chal[C(V1)] = tr.get_field()                         # challenges [5], [6]
    chal[C(V2)] = tr.get_field()
    xd, xdw, f_2ns = ctx.empty(NE * 3), ctx.empty(NE * 3), S("f_2ns")
    ctx.x_div_x_sub(xd, x_2ns, NE, xi)
    ctx.x_div_x_sub(xdw, x_2ns, NE, wxi)
    prog52.run52(mem, const_2ns, NC, chal, evals, xd, xdw, f_2ns, 0, NE)
    if tamper == "f":                                   # a value that is not on the low-degree polynomial
        h = ctx.to_host(f_2ns)
        h[3 * 5] ^= np.uint64(1)
        f_2ns = ctx.to_device(h)
    # ---- FRI (friProve.cpp:5-190)
    fri_roots, fri_trees, fri_srcs = [], {}, {}
    pol, nxt, pol_bits = f_2ns, ctx.empty(NE * 3), nbits_ext
    for si, cur in enumerate(steps):
        x = tr.get_field()
        ctx.fri_fold(nxt, pol, pol_bits, cur, nbits_ext, x)
        if si < len(steps) - 1:
            nb = steps[si + 1]
            groups, gsz = 1 << nb, (1 << (cur - nb)) * 3
            src = ctx.empty((1 << cur) * 3)
            ctx.fri_transpose(src, nxt, 1 << cur, nb)
            nodes = ctx.empty((2 * groups - 1) * 4)
            ctx.merkle_build(nodes, src, gsz, groups)
            root = ctx.to_host(nodes[-4:])
            tr.put(root)
            fri_roots.append(root)
            fri_trees[si + 1], fri_srcs[si + 1] = nodes, src
        else:
            final_pol = ctx.to_host(nxt[:(1 << cur) * 3])
            if tamper == "final":
                final_pol[0] ^= np.uint64(1)
            tr.put(final_pol)
        pol, nxt = nxt, ctx.empty(NE * 3)
        pol_bits = cur
    ys = tr.get_permutations(n_queries, steps[0])
    # ---- queries (friProve.cpp:219-250)
    def open_tree(nodes, src, height, width, idx):
        buf = ctx.empty(len(idx) * (width + 4 * (height - 1).bit_length()))
        ctx.merkle_group_proofs(buf, nodes, src, height, width, idx)
        return ctx.to_host(buf).reshape(len(idx), -1)
    proof = {"nbits": nbits, "root1": root1, "root2": root2, "root3": root3, "root4": root4, "evals": evals, "fri_roots": fri_roots,
             "final_pol": final_pol, "publics": pub,
             "s0": {"cm1": open_tree(nodes1, cm1, NE, 5, ys), "cm2": open_tree(nodes2, cm2, NE, 5, ys), "cm3": open_tree(nodes3, cm3, NE, 6, ys),
                    "cm4": open_tree(nodes4, cm4, NE, 6, ys), "const": open_tree(const_nodes, const_2ns, NE, NC, ys)},
             "fri": {}, "const_root": ctx.to_host(const_nodes[-4:])}
    y = ys.copy()
    for si in range(1, len(steps)):
        y = y % np.uint64(1 << steps[si])
        gsz = (1 << (steps[si - 1] - steps[si])) * 3
        proof["fri"][si] = open_tree(fri_trees[si], fri_srcs[si], 1 << steps[si], gsz, y)
    if tamper == "opening":
        proof["s0"]["cm1"][0][0] ^= np.uint64(1)
    if tamper == "stage2":
        proof["s0"]["cm2"][1][2] ^= np.uint64(1)
    if tamper == "stage3":
        proof["s0"]["cm3"][2][1] ^= np.uint64(1)
    for p in (progb, progc, progd, prog42, prog52):
        p.close()
    return proof


# ------------------------------------------------------------------ verifier: oracle arithmetic only (pil-stark stark_verify)
def verify(proof, const_root, n_queries=12):
    """Returns (ok, reason)."""
    L = glo.lib()
    nbits = proof["nbits"]
    nbits_ext = nbits + 1
    N, NE = 1 << nbits, 1 << nbits_ext
    steps = fri_steps(nbits_ext)
    if not np.array_equal(proof["const_root"], const_root):
        return False, "constant-polynomial root is not the verification key's"
    # ---- transcript replay
    tr = glo.Transcript()
    pub = [int(v) for v in proof["publics"]]
    tr.put(proof["publics"])
    tr.put(proof["root1"])
    gamma = e3(tr.get_field())
    beta = e3(tr.get_field())
    tr.put(proof["root2"])
    gamma2 = e3(tr.get_field())
    beta2 = e3(tr.get_field())
    tr.put(proof["root3"])
    vc = e3(tr.get_field())
    tr.put(proof["root4"])
    xi = e3(tr.get_field())
    ev = proof["evals"]
    tr.put(ev)
    c5 = tr.get_field()
    c6 = tr.get_field()
    fri_chal = []
    for si in range(len(steps)):
        fri_chal.append(tr.get_field())
        if si < len(steps) - 1:
            tr.put(proof["fri_roots"][si])
        else:
            tr.put(proof["final_pol"])
    ys = tr.get_permutations(n_queries, steps[0])
    E = lambda k: e3(ev[3 * k:3 * k + 3])
    # ---- constraint identity at xi: Horner_vc(C1 .. C9) == Q(xi) * (xi^N - 1)
    one = [1, 0, 0]
    not_last = e3_sub(one, E(EV_LLAST))
    C1 = e3_mul(not_last, e3_sub(E(EV_AW), E(EV_B)))
    C2 = e3_mul(not_last, e3_sub(e3_sub(E(EV_BW), E(EV_A)), E(EV_B)))
    C3 = e3_mul(E(EV_L1), e3_sub(E(EV_A), e3_from_base(pub[0])))
    C4 = e3_mul(E(EV_L1), e3_sub(E(EV_B), e3_from_base(pub[1])))
    C5 = e3_sub(e3_mul(e3_add(E(EV_A), gamma), e3_add(E(EV_B), gamma)), E(EV_Z))
    C6 = e3_sub(e3_mul(E(EV_GPW), e3_add(E(EV_D), beta)), e3_mul(E(EV_GP), e3_add(E(EV_C), beta)))
    C7 = e3_mul(E(EV_L1), e3_sub(E(EV_GP), one))
    C = [0, 0, 0]
    g1 = e3_add(one, beta2)                                                  # the lookup: p2' den - p2 num, L1 (p2 - 1)
    gg = e3_mul(gamma2, g1)
    num = e3_mul(e3_mul(e3_add(E(EV_F), gamma2), g1), e3_add(e3_add(E(EV_T), e3_mul(beta2, E(EV_TW))), gg))
    den = e3_mul(e3_add(e3_add(E(EV_H1), e3_mul(beta2, E(EV_H2))), gg), e3_add(e3_add(E(EV_H2), e3_mul(beta2, E(EV_H1W))), gg))
    C8 = e3_sub(e3_mul(E(EV_P2W), den), e3_mul(E(EV_P2), num))
    C9 = e3_mul(E(EV_L1), e3_sub(E(EV_P2), one))
    for Ck in (C1, C2, C3, C4, C5, C6, C7, C8, C9):
        C = e3_add(e3_mul(C, vc), Ck)
    xiN = e3_pow(xi, N)
    Q = e3_add(E(EV_Q0), e3_mul(xiN, E(EV_Q1)))
    if C != e3_mul(Q, e3_sub(xiN, one)):
        return False, "constraint identity fails at the challenge point"
    # ---- queries
    ops52, args52 = step52ns_program(Layout(0, 1))       # one extended row on its own: cm1 | cm2 | cm3 | cm4 columns
    chal = np.zeros(N_CHALLENGES * 3, dtype=np.uint64)
    chal[3 * V1:3 * V1 + 3], chal[3 * V2:3 * V2 + 3] = c5, c6
    wN = L.glo_w(nbits)
    wxi = [L.glo_mul(v, wN) for v in xi]
    h1, h2, h3, h4, hc = proof["s0"]["cm1"], proof["s0"]["cm2"], proof["s0"]["cm3"], proof["s0"]["cm4"], proof["s0"]["const"]
    y = [int(v) for v in ys]
    for q in range(n_queries):
        idx = y[q]
        for (pr, w, root, name) in ((h1, 5, proof["root1"], "cm1"), (h2, 5, proof["root2"], "cm2"), (h3, 6, proof["root3"], "cm3"),
                                    (h4, 6, proof["root4"], "cm4"), (hc, 3, const_root, "const")):
            if not glo.merkle_verify(root, pr[q][:w], pr[q][w:], idx):
                return False, "Merkle opening of %s fails at query %d" % (name, q)
        # the FRI polynomial at x = shift * w^idx from the opened rows (the same program, over one row)
        x = L.glo_mul(SHIFT, L.glo_pow(L.glo_w(nbits_ext), idx))
        def xdiv(z):
            den = np.array([(x - z[0]) % P, (-z[1]) % P, (-z[2]) % P], dtype=np.uint64)
            return np.array(e3_mul([int(v) for v in glo.e3_inv(den)], [x, 0, 0]), dtype=np.uint64)
        row = np.concatenate([h1[q][:5], h2[q][:5], h3[q][:6], h4[q][:6]]).astype(np.uint64)
        f = np.zeros(3, dtype=np.uint64)
        glo.chelpers_step52ns(ops52, args52, row, np.ascontiguousarray(hc[q][:3]), 3, chal, ev, xdiv(xi), xdiv(wxi), f, 0, 1)
        # level by level: the value must sit in the next group, the group must fold to the value after it
        val, g = f, idx
        for si in range(1, len(steps)):
            prev, cur = steps[si - 1], steps[si]
            gsz = (1 << (prev - cur)) * 3
            pr = proof["fri"][si][q]
            gi = g % (1 << cur)
            if not glo.merkle_verify(proof["fri_roots"][si - 1], pr[:gsz], pr[gsz:], gi):
                return False, "Merkle opening of FRI step %d fails at query %d" % (si, q)
            j = g >> cur
            if not np.array_equal(pr[3 * j:3 * j + 3], val):
                return False, "FRI step %d: the opened group does not contain the previous value (query %d)" % (si, q)
            val = glo.fri_fold_group(pr[:gsz], prev - cur, prev, nbits_ext, gi, fri_chal[si])
            g = gi
        if not np.array_equal(proof["final_pol"][3 * g:3 * g + 3], val):
            return False, "the last fold does not land on the final polynomial (query %d)" % q
    # ---- the final polynomial has degree < 2^(last - (nBitsExt - nBits))
    last = steps[-1]
    coef = glo.ntt(proof["final_pol"].reshape(1 << last, 3), 1 << last, 3, inverse=True).reshape(-1, 3)
    if coef[1 << (last - (nbits_ext - nbits)):].any():
        return False, "final polynomial is not low-degree"
    return True, "ok"
