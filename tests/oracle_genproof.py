"""An oracle-side Starks::genProof (TEST INFRASTRUCTURE: CPU, the oracle's primitives only).

A second reading of the reference's prover, written from /root/reference/src/starkpil/starks.cpp:9-403 (genProof), :405-553 (the
column transposes around calculateH1H2 / calculateZ: which polynomial is f, t, h1, h2, num, den, z), :555-668 (evmap),
fri/friProve.cpp:5-271 (fold, step trees, queries), fri/friProof.hpp + fri/proof2zkinStark.cpp:8-82 (what goes into zkin.json and
in which order), stark_info.cpp:473-482 (getPolinomial) -- NOT from this repo's host/starks.hpp, which it is there to check.  It
proves over one host array laid out as StarkInfo's memory map, exactly as the reference does over pAddress, from the same inputs a
`Starks` is constructed from (starkinfo.json as a dictionary, the constant polynomials, the constant-tree file image, the witness,
the five generated tables), and returns the zkin.json TEXT in the layout of host/standalone/proof2zkinStark.hpp's writer (which is
itself pinned byte for byte against reference-produced files, tests/test_cpp_shims.py).  The device proof must equal it byte for byte
(tests/test_genproof_parity.py).

Deliberate differences from the reference's text, none of which can change a bit:
  * no transposes into pBuffer around calculateH1H2 / calculateZ (they feed sequential loops; the oracle works on strided views);
  * evmap sums in evMap order per evaluation (exact field arithmetic: the reference's per-thread partial sums give the same value);
  * xDivXSubXi = x / (x - xi) through one exact inversion per element where the reference batches inversions.
"""
import ctypes

import numpy as np

import glo

SECTIONS = ["cm1_n", "cm1_2ns", "cm2_n", "cm2_2ns", "cm3_n", "cm3_2ns", "cm4_n", "cm4_2ns", "tmpExp_n", "q_2ns", "f_2ns"]
u64 = ctypes.c_uint64


class OracleStarks:
    """The state a reference `Starks` object holds (starks.hpp:74-183), over host arrays."""

    def __init__(self, si, const_n, const_tree):
        self.si = si
        self.nbits, self.nbits_ext = si["starkStruct"]["nBits"], si["starkStruct"]["nBitsExt"]
        self.N, self.NE = 1 << self.nbits, 1 << self.nbits_ext
        self.n_const = si["nConstants"]
        self.off = {k: int(si["mapOffsets"][k]) for k in SECTIONS}
        self.cols = {k: int(si["mapSectionsN"][k]) for k in SECTIONS}
        self.const_n = glo.A(const_n).reshape(-1)
        self.const_tree = glo.A(const_tree).reshape(-1)
        L = glo.lib()
        # starks.hpp:141-143: the extended constant polynomials are the tree file's, after its two-word header
        self.const_2ns = self.const_tree[2:2 + self.n_const * self.NE]
        self.x_n = glo.geom_seq(self.N, 1, L.glo_w(self.nbits))                   # starks.hpp:149-154
        self.x_2ns = glo.geom_seq(self.NE, L.glo_shift(), L.glo_w(self.nbits_ext))  # :155-160 (and `x`, :176-183: the same sequence)
        self.zhinv = glo.zhinv(self.nbits, self.nbits_ext)                        # zhInv.cpp:7-31

    def pol(self, pid):
        """stark_info.cpp:473-482 -> (offset, stride, dim) in the memory map."""
        v = self.si["varPolMap"][pid]
        return self.off[v["section"]] + int(v["sectionPos"]), self.cols[v["section"]], int(v["dim"])

    def exp_pol(self, exp_id):
        return self.pol(int(self.si["exp2pol"][str(exp_id)]))


_FAST = [False]


def _tree(src, ncols, nrows):
    """fast: the hand-vectorised restatement of oracle/cpu_baseline_avx2.c (6 x the checker's permutation; itself checked bit for bit against
    the checker by tests/test_cpu_baseline.py) -- for the larger shapes, where the scalar tree alone would take a minute"""
    if _FAST[0] and ncols > 4:
        L = glo.lib("baseline")
        nodes = np.zeros((2 * nrows - 1) * 4, dtype=np.uint64)
        L.glb_merkletree(glo.ptr(nodes), glo.ptr(glo.A(src).reshape(-1)), u64(ncols), u64(nrows))
        return nodes
    return glo.merkletree(src, ncols, nrows)


def _extend(src, n_ext, n, ncols):
    if _FAST[0]:
        L = glo.lib("baseline")
        out = np.zeros(n_ext * ncols, dtype=np.uint64)
        L.glb_extend_pol(glo.ptr(out), glo.ptr(glo.A(src).reshape(-1)), u64(n_ext), u64(n), u64(ncols))
        return out
    return glo.extend_pol(src, n_ext, n, ncols).reshape(-1)


def gen_proof(si, progs, const_n, const_tree, witness, publics, fast=False):
    """-> (zkin text, debug dictionary).  progs: {"step2prev" | "step3prev" | "step3" | "step42ns" | "step52ns": (ops, args)}; a missing
    program is a stage without expressions (the step computes nothing).  fast: trees and extensions through the vectorised restatement."""
    _FAST[0] = bool(fast)
    st = OracleStarks(si, const_n, const_tree)
    L = glo.lib()
    N, NE, nbits, nbits_ext = st.N, st.NE, st.nbits, st.nbits_ext
    off, cols = st.off, st.cols
    ext_bits = nbits_ext - nbits
    mem = np.zeros(int(si["mapTotalN"]), dtype=np.uint64)
    sec = lambda k: mem[off[k]:off[k] + cols[k] * (NE if k.endswith("2ns") else N)]
    sec("cm1_n")[:] = glo.A(witness).reshape(-1)                 # the executor's output: cm1_n (prover.cpp:99-120)
    publics = glo.A(publics).reshape(-1)
    n_evals = len(si["evMap"])
    challenges = np.zeros(8 * 3, dtype=np.uint64)               # NUM_CHALLENGES (starks.hpp)
    C = lambda k: slice(3 * k, 3 * k + 3)
    tr = glo.Transcript()
    tr.put(publics[:si["nPublics"]])                             # starks.cpp:28

    def base_step(name):
        if name in progs and len(progs[name][0]):
            ops, args = progs[name]
            glo.chelpers_stepbase(ops, args, mem, st.const_n, st.n_const, challenges, publics, st.x_n, 1, np.arange(N, dtype=np.uint64))

    def commit(src_sec, dst_sec):
        w = cols[src_sec]
        sec(dst_sec)[:] = _extend(sec(src_sec), NE, N, w) if w else 0
        return _tree(sec(dst_sec), w, NE)

    # ---- 1 (starks.cpp:48-61)
    nodes = [None] * 4
    nodes[0] = commit("cm1_n", "cm1_2ns")
    root0 = nodes[0][-4:].copy()
    tr.put(root0)
    # ---- 2 (starks.cpp:66-143)
    challenges[C(0)] = tr.get_field()
    challenges[C(1)] = tr.get_field()
    base_step("step2prev")
    num_commited = int(si["nCm1"])                               # starks.cpp:14
    for i, pu in enumerate(si["puCtx"]):                         # :405-436 which polynomials, :106-124 the call
        f, t = st.exp_pol(pu["fExpId"]), st.exp_pol(pu["tExpId"])
        h1, h2 = st.pol(si["cm_n"][num_commited + 2 * i]), st.pol(si["cm_n"][num_commited + 2 * i + 1])
        assert h1[2] in (1, 3) and f[2] == t[2] == h1[2] == h2[2]
        r = glo.calculate_h1h2(mem, h1[0], h1[1], h2[0], h2[1], f[0], f[1], t[0], t[1], h1[2], N)
        if r:
            raise RuntimeError("calculateH1H2: number not included: w=%d" % (r - 1))
    num_commited += 2 * len(si["puCtx"])                         # :453
    nodes[1] = commit("cm2_n", "cm2_2ns")
    root1 = nodes[1][-4:].copy()
    tr.put(root1)
    # ---- 3 (starks.cpp:148-223)
    challenges[C(2)] = tr.get_field()
    challenges[C(3)] = tr.get_field()
    base_step("step3prev")
    k = 0
    zjobs = []
    for ctx in (si["puCtx"], si["peCtx"], si["ciCtx"]):         # :455-536: lookups, permutations, connections, z = cm_n[numCommited + running index]
        for x in ctx:
            num, den, z = st.exp_pol(x["numId"]), st.exp_pol(x["denId"]), st.pol(si["cm_n"][num_commited + k])
            zjobs.append((z[0], z[1], num[0], num[1], den[0], den[1]))  # (whether the product closes is polinomial.hpp:606's zkassert: a
            k += 1                                                      # release build goes on, and so do the synthetic shapes)
    # the products are independent of one another (each reads its own numerator and denominator and writes its own z column): a few at a
    # time on host threads -- the foreign call releases the interpreter lock -- which changes no value and shortens the large parity runs
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=8) as pool:
        list(pool.map(lambda a: glo.calculate_z(mem, a[0], a[1], a[2], a[3], a[4], a[5], N), zjobs))
    base_step("step3")
    nodes[2] = commit("cm3_n", "cm3_2ns")
    root2 = nodes[2][-4:].copy()
    tr.put(root2)
    # ---- 4 (starks.cpp:228-295)
    challenges[C(4)] = tr.get_field()
    q_dim, q_deg = int(si["qDim"]), int(si["qDeg"])
    assert q_dim == 3
    q_2ns = sec("q_2ns")
    if "step42ns" in progs:
        ops, args = progs["step42ns"]
        glo.chelpers_step42ns(ops, args, mem, st.const_2ns, st.n_const, challenges, publics, st.x_2ns, 1, st.zhinv, q_2ns, 0, NE)
    qq1 = glo.ntt(q_2ns, NE, q_dim, inverse=True).reshape(-1)    # :261
    qq2 = np.zeros(NE * q_deg * q_dim, dtype=np.uint64)          # :232 calloc: rows >= N stay zero
    L.glo_q_split(glo.ptr(qq2), glo.ptr(qq1), u64(N), ctypes.c_uint(q_deg))    # :265-280
    sec("cm4_2ns")[:] = glo.ntt(qq2, NE, q_dim * q_deg).reshape(-1)            # :284
    nodes[3] = _tree(sec("cm4_2ns"), cols["cm4_2ns"], NE)
    root3 = nodes[3][-4:].copy()
    tr.put(root3)
    # ---- 5 (starks.cpp:300-390)
    challenges[C(7)] = tr.get_field()
    xi = challenges[C(7)].copy()
    sinv, w_n = L.glo_inv(L.glo_shift()), L.glo_w(nbits)
    xis = np.array([L.glo_mul(int(v), sinv) for v in xi], dtype=np.uint64)                   # :316
    wxi = np.array([L.glo_mul(int(v), w_n) for v in xi], dtype=np.uint64)                    # :317, :348
    wxis = np.array([L.glo_mul(int(v), sinv) for v in wxi], dtype=np.uint64)                 # :318
    lev = glo.ntt(glo.geom_seq3(N, xis), N, 3, inverse=True).reshape(-1)                     # :320-326
    lpev = glo.ntt(glo.geom_seq3(N, wxis), N, 3, inverse=True).reshape(-1)
    views, prime = [], []
    for ev in si["evMap"]:                                       # :555-590
        if ev["type"] == "const":
            views.append((st.const_2ns, int(ev["id"]), 1, st.n_const))
        elif ev["type"] == "cm":
            o, s, d = st.pol(si["cm_2ns"][ev["id"]])
            views.append((mem, o, d, s))
        elif ev["type"] == "q":
            o, s, d = st.pol(si["qs"][ev["id"]])
            views.append((mem, o, d, s))
        else:
            raise ValueError("Invalid ev type: " + str(ev["type"]))
        prime.append(1 if ev["prime"] else 0)
    evals = _evmap(views, prime, lev, lpev, N, ext_bits) if n_evals else np.zeros(0, dtype=np.uint64)
    for i in range(n_evals):
        tr.put(evals[3 * i:3 * i + 3])                           # :342-345
    challenges[C(5)] = tr.get_field()
    challenges[C(6)] = tr.get_field()
    xd, xdw = _x_div_x_sub(st.x_2ns, xi), _x_div_x_sub(st.x_2ns, wxi)       # :350-365
    f_2ns = sec("f_2ns")
    if "step52ns" in progs:
        ops, args = progs["step52ns"]
        glo.chelpers_step52ns(ops, args, mem, st.const_2ns, st.n_const, challenges, evals if n_evals else np.zeros(3, dtype=np.uint64), xd, xdw, f_2ns, 0, NE)
    # ---- FRI (starks.cpp:391-402, friProve.cpp:5-190)
    steps = [int(s["nBits"]) for s in si["starkStruct"]["steps"]]
    n_queries = int(si["starkStruct"]["nQueries"])
    pol, pol_bits = f_2ns.copy(), nbits_ext
    fri_roots, fri_nodes, fri_srcs = {}, {}, {}
    for s_i, cur in enumerate(steps):
        special_x = tr.get_field()                               # friProve.cpp:30
        pol2 = glo.fri_fold(pol[:3 << pol_bits], pol_bits, cur, nbits_ext, special_x).reshape(-1)      # :44-108
        if s_i < len(steps) - 1:                                 # :110-126
            nxt = steps[s_i + 1]
            n_groups, group = 1 << nxt, (1 << cur) >> nxt
            src = glo.fri_transpose(pol2, 1 << cur, nxt)
            fri_nodes[s_i + 1] = _tree(src, group * 3, n_groups)
            fri_srcs[s_i + 1] = src
            fri_roots[s_i + 1] = fri_nodes[s_i + 1][-4:].copy()
            tr.put(fri_roots[s_i + 1])
        else:
            for i in range(1 << cur):                            # :128-134
                tr.put(pol2[3 * i:3 * i + 3])
        pol, pol_bits = pol2, cur
    final_pol = pol[:3 << steps[-1]]                             # :150 setPol
    ys = [int(v) for v in tr.get_permutations(n_queries, steps[0])]            # :155
    widths = [cols["cm1_n"], cols["cm2_n"], cols["cm3_n"], cols["cm4_2ns"], st.n_const]          # starks.hpp:186-190
    tree_c_nodes = st.const_tree[2 + st.n_const * NE:]          # merkleTreeGL.hpp:24-32
    srcs = [sec("cm1_2ns"), sec("cm2_2ns"), sec("cm3_2ns"), sec("cm4_2ns"), st.const_2ns]
    all_nodes = nodes + [tree_c_nodes]
    queries = {0: []}
    for idx in ys:                                               # :219-236: the five trees at the same index
        queries[0].append([_open(all_nodes[t], srcs[t], NE, widths[t], idx) for t in range(5)])
    y = list(ys)
    for s_i in range(1, len(steps)):
        y = [v % (1 << steps[s_i]) for v in y]                   # :171-177
        g = (1 << steps[s_i - 1]) >> steps[s_i]
        queries[s_i] = [[_open(fri_nodes[s_i], fri_srcs[s_i], 1 << steps[s_i], g * 3, v)] for v in y]
    # ---- zkin.json (friProof.hpp:28-219 + proof2zkinStark.cpp:8-82)
    J = _Json
    o = '{"root1":' + J.arr1(root0) + ',"root2":' + J.arr1(root1) + ',"root3":' + J.arr1(root2) + ',"root4":' + J.arr1(root3)
    o += ',"evals":' + J.arr([J.arr1(evals[3 * i:3 * i + 3]) for i in range(n_evals)])
    for s_i in range(1, len(steps)):
        o += ',"s%d_root":' % s_i + J.arr1(fri_roots[s_i])
        o += ',"s%d_vals":' % s_i + J.arr([J.vals(q[0][0]) for q in queries[s_i]])
        o += ',"s%d_siblings":' % s_i + J.arr([J.sibs(q[0][1]) for q in queries[s_i]])
    names = ["1", "2", "3", "4", "C"]
    present = [t for t in range(5) if not (t in (1, 2) and widths[t] == 0)]
    for t in present:
        o += ',"s0_vals%s":' % names[t] + J.arr([J.vals(q[t][0]) for q in queries[0]])
    for t in present:
        o += ',"s0_siblings%s":' % names[t] + J.arr([J.sibs(q[t][1]) for q in queries[0]])
    o += ',"finalPol":' + J.arr([J.arr1(final_pol[3 * i:3 * i + 3]) for i in range(1 << steps[-1])])
    o += ',"publics":' + J.arr1(publics[:si["nPublics"]]) + "}"
    dbg = {"challenges": challenges, "ys": ys, "mem": mem, "evals": evals, "xDivXSubXi": xd, "xDivXSubWXi": xdw}
    return o, dbg


def _evmap(views, prime, lev, lpev, n, ext_bits):
    return glo.evmap(views, prime, lev, lpev, n, ext_bits).reshape(-1)


def _x_div_x_sub(x, z):
    """starks.cpp:350-365: x_k / (x_k - z) in the cubic extension, x_k in the base field."""
    n = x.size
    den = np.zeros(3 * n, dtype=np.uint64)
    P = np.uint64(glo.P)
    z = [np.uint64(int(v) % glo.P) for v in z]
    xs = x.astype(np.uint64)
    with np.errstate(over="ignore"):
        d0 = np.where(xs >= z[0], xs - z[0], xs + (P - z[0]))
    den[0::3] = d0
    den[1::3] = (P - z[1]) % P
    den[2::3] = (P - z[2]) % P
    inv = glo.batch_inverse3(den)
    out = np.zeros(3 * n, dtype=np.uint64)
    for k in range(3):                                           # ext * base: component-wise
        out[k::3] = _mul_vec(inv[k::3], xs)
    return out


def _mul_vec(a, b):
    """element-wise Goldilocks product of two u64 vectors through Python integers split in 32-bit halves (numpy has no 128-bit type)."""
    a = a.astype(object)
    b = b.astype(object)
    return np.array((a * b) % glo.P, dtype=object).astype(np.uint64)


def _open(nodes, src, height, width, idx):
    """merkleTreeGL.cpp:12-35 getGroupProof -> (the row's values, the siblings level by level)."""
    p = glo.merkle_group_proof(nodes, src, height, width, idx) if width else _open_zero_width(nodes, height, idx)
    return p[:width], p[width:].reshape(-1, 4)


def _open_zero_width(nodes, height, idx):
    nlev = max(height - 1, 0).bit_length()
    out = np.zeros(4 * nlev, dtype=np.uint64)
    o, level = 0, height
    for k in range(nlev):
        out[4 * k:4 * k + 4] = nodes[(o + (idx ^ 1)) * 4:(o + (idx ^ 1)) * 4 + 4]
        idx >>= 1
        o += level
        level >>= 1
    return out


class _Json:
    """friProof.hpp's writers as text: every field element a decimal string."""
    @staticmethod
    def s(v):
        return '"%d"' % int(v)

    @staticmethod
    def arr(items):
        return "[" + ",".join(items) + "]"

    @staticmethod
    def arr1(v):
        return "[" + ",".join('"%d"' % int(x) for x in v) + "]"

    @staticmethod
    def vals(v):                                                 # friProof.hpp:31-48: one element per "linear" -> plain strings
        return _Json.arr1(v)

    @staticmethod
    def sibs(m):                                                 # friProof.hpp:50-60
        return "[" + ",".join(_Json.arr1(r) for r in m) + "]"
