#!/usr/bin/env python3
"""Derive the golden fixtures under tests/golden/ from the reference's own golden STARK proofs.

Inputs (read as DATA at development time; /root/reference does not travel to the GPU box):
  /root/reference/testvectors/aggregatedProof/recursive1.zkin.proof_{0,1,2,3}.json
  /root/reference/testvectors/finalProof/recursive2.zkin.proof_{01,03,23}.json
Key layout of those files: src/starkpil/fri/proof2zkinStark.cpp:8-82.

Output: tests/golden/<name>.npz, one per proof, holding (all uint64):
  root1..root4, s1_root..s4_root, finalPol[64,3], evals, publics,
  q_index[Q]                       leaf index of each kept query in the step-0 trees (recovered below)
  s0_vals{1,3,4,C}[Q,w], s0_siblings{1,3,4,C}[Q,20,4]
  s{1..4}_vals[Q,w], s{1..4}_siblings[Q,levels,4]
  special_x[4,3]                   FRI challenges of steps 1..4, recovered from the fold relations
Two proofs keep all 43 queries, the rest keep the first 6 (fixture size).

What is DERIVED rather than copied (neither is stored in the proof JSON):
  * q_index: brute-forced per query so that climb(linear_hash(vals), siblings, idx) == root, from the
    6-bit s4 tree upward (friProve.cpp:171-177: idx_{s+1} = idx_s mod 2^{bits_{s+1}}).
  * special_x: the fold relation of friProve.cpp:86-104 is linear in (X, X^2, ..., X^{nX-1}); the 43
    queries give an over-determined linear system over F_p^3 whose solution must be a geometric
    sequence -- its first entry is the challenge.  Uses all 43 queries even when only 6 are kept.
The Poseidon/field arithmetic used for the derivation is the repo's CPU oracle (oracle/libgl_oracle.so);
tests/test_oracle_golden.py then re-checks every stored relation.
"""
import ctypes, json, os, sys
import numpy as np

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
L = ctypes.CDLL(os.path.join(ROOT, "oracle", "libgl_oracle.so"))
u64 = ctypes.c_uint64
P = 0xFFFFFFFF00000001
BITS = [20, 16, 12, 9, 6]          # starkStruct.steps of recursive1/recursive2 (path lengths in the files)
NBITS_EXT = 20

def arr(x):
    return np.array(x, dtype=object).astype(np.uint64) if not isinstance(x, np.ndarray) else x

def toint(x):
    if isinstance(x, list):
        return [toint(y) for y in x]
    return int(x)

def cptr(a):
    return a.ctypes.data_as(ctypes.POINTER(u64))

def verify(root, vals, sibs, idx):
    root = np.ascontiguousarray(root, dtype=np.uint64)
    vals = np.ascontiguousarray(vals, dtype=np.uint64)
    sibs = np.ascontiguousarray(sibs, dtype=np.uint64)
    return L.glo_merkle_verify(cptr(root), cptr(vals), u64(len(vals)), cptr(sibs), u64(sibs.shape[0]), u64(idx)) == 1

# ---- tiny F_p^3 arithmetic in python ints (x^3 = x + 1), for the linear solve only
def e_add(a, b): return [(a[i] + b[i]) % P for i in range(3)]
def e_sub(a, b): return [(a[i] - b[i]) % P for i in range(3)]
def e_mul(a, b):
    c = [0] * 5
    for i in range(3):
        for j in range(3):
            c[i + j] += a[i] * b[j]
    # x^3 = x + 1, x^4 = x^2 + x
    return [(c[0] + c[3]) % P, (c[1] + c[3] + c[4]) % P, (c[2] + c[4]) % P]
def e_inv(a):
    out = (u64 * 3)(); inp = (u64 * 3)(*a)
    L.glo3_inv(out, inp)
    r = list(out)
    assert e_mul(r, a) == [1, 0, 0]
    return r
def b_pow(a, e): return pow(a, e, P)

def intt_small(vals, nx):
    """vals: nx ext elements -> coefficients (friProve.cpp:100-102)"""
    nxb = nx.bit_length() - 1
    L.glo_w.restype = u64
    w = L.glo_w(nxb); winv = pow(w, P - 2, P); ninv = pow(nx, P - 2, P)
    out = []
    for k in range(nx):
        acc = [0, 0, 0]
        for i in range(nx):
            t = pow(winv, (i * k) % nx, P)
            acc = [(acc[d] + vals[i][d] * t) % P for d in range(3)]
        out.append([a * ninv % P for a in acc])
    return out

def solve_special_x(coef_rows, targets, nx):
    """rows: per query [c_0..c_{nx-1}] (ext);  sum_k c_k X^k = target.  Unknowns y_k = X^k, k>=1."""
    n_unk = nx - 1
    A = [[r[k] for k in range(1, nx)] + [e_sub(t, r[0])] for r, t in zip(coef_rows, targets)]
    row = 0
    for col in range(n_unk):
        piv = next(i for i in range(row, len(A)) if any(A[i][col]))
        A[row], A[piv] = A[piv], A[row]
        inv = e_inv(A[row][col])
        A[row] = [e_mul(v, inv) for v in A[row]]
        for i in range(len(A)):
            if i != row and any(A[i][col]):
                f = A[i][col]
                A[i] = [e_sub(A[i][j], e_mul(f, A[row][j])) for j in range(n_unk + 1)]
        row += 1
    ys = [A[k][n_unk] for k in range(n_unk)]
    for i in range(n_unk, len(A)):          # over-determined rows must have vanished
        assert all(not any(v) for v in A[i]), "inconsistent fold system"
    x = ys[0]
    cur = x
    for k in range(1, n_unk):
        cur = e_mul(cur, x)
        assert cur == ys[k], "solution is not geometric"
    return x

def process(path, name, keep):
    d = json.load(open(path))
    out = {}
    for k in ["root1", "root2", "root3", "root4", "s1_root", "s2_root", "s3_root", "s4_root"]:
        out[k] = np.array(toint(d[k]), dtype=np.uint64)
    out["finalPol"] = np.array(toint(d["finalPol"]), dtype=np.uint64)
    out["evals"] = np.array(toint(d["evals"]), dtype=np.uint64)
    out["publics"] = np.array(toint(d["publics"]), dtype=np.uint64)
    nq = len(d["s0_vals1"])
    sv = {s: np.array(toint(d[f"s{s}_vals"]), dtype=np.uint64) for s in range(1, 5)}
    ss = {s: np.array(toint(d[f"s{s}_siblings"]), dtype=np.uint64) for s in range(1, 5)}
    s0v = {t: np.array(toint(d[f"s0_vals{t}"]), dtype=np.uint64) for t in "134C"}
    s0s = {t: np.array(toint(d[f"s0_siblings{t}"]), dtype=np.uint64) for t in "134C"}
    # ---- recover indices
    idx0 = []
    for q in range(nq):
        cand = [i for i in range(1 << BITS[4]) if verify(out["s4_root"], sv[4][q], ss[4][q], i)]
        assert len(cand) == 1, (name, q, cand)
        idx = cand[0]
        for s in (3, 2, 1):
            step = 1 << BITS[s + 1]
            cand = [idx + step * t for t in range(1 << (BITS[s] - BITS[s + 1]))
                    if verify(out[f"s{s}_root"], sv[s][q], ss[s][q], idx + step * t)]
            assert len(cand) == 1, (name, q, s, cand)
            idx = cand[0]
        step = 1 << BITS[1]
        found = None
        for t in range(1 << (BITS[0] - BITS[1])):
            i = idx + step * t
            if verify(out["root1"], s0v["1"][q], s0s["1"][q], i):
                assert found is None
                found = i
        assert found is not None
        for tname, rname in (("3", "root3"), ("4", "root4")):
            assert verify(out[rname], s0v[tname][q], s0s[tname][q], found)
        idx0.append(found)
    # ---- recover the FRI challenges from the fold relations (all queries)
    L.glo_w.restype = u64
    xs = []
    for s in range(1, 5):
        prev, cur = BITS[s - 1], BITS[s]
        nx = 1 << (prev - cur)
        sinv0 = pow(pow(49, P - 2, P), 1 << (NBITS_EXT - prev), P)
        wi = pow(L.glo_w(prev), P - 2, P)
        rows, targets = [], []
        for q in range(nq):
            g = idx0[q] % (1 << cur)
            vals = [[int(v) for v in sv[s][q][3 * i:3 * i + 3]] for i in range(nx)]
            c = intt_small(vals, nx)
            sinv = sinv0 * pow(wi, g, P) % P
            rows.append([[cc * pow(sinv, k, P) % P for cc in c[k]] for k in range(nx)])
            if s < 4:
                nxt = BITS[s + 1]
                j = g >> nxt
                targets.append([int(v) for v in sv[s + 1][q][3 * j:3 * j + 3]])
            else:
                targets.append([int(v) for v in out["finalPol"][g]])
        xs.append(solve_special_x(rows, targets, nx))
    out["special_x"] = np.array(xs, dtype=np.uint64)
    # ---- keep a subset of queries
    Q = list(range(nq))[:keep]
    out["q_index"] = np.array([idx0[q] for q in Q], dtype=np.uint64)
    for t in "134C":
        out[f"s0_vals{t}"] = s0v[t][Q]
        out[f"s0_siblings{t}"] = s0s[t][Q]
    for s in range(1, 5):
        out[f"s{s}_vals"] = sv[s][Q]
        out[f"s{s}_siblings"] = ss[s][Q]
    out["steps_bits"] = np.array(BITS, dtype=np.uint64)
    np.savez(os.path.join(HERE, name + ".npz"), **out)
    print(name, "queries kept", len(Q), "idx0[0]", idx0[0], "special_x[3]", xs[3])

if __name__ == "__main__":
    files = [("testvectors/aggregatedProof/recursive1.zkin.proof_%d.json" % i, "recursive1_proof_%d" % i) for i in range(4)]
    files += [("testvectors/finalProof/recursive2.zkin.proof_%s.json" % s, "recursive2_proof_%s" % s) for s in ("01", "03", "23")]
    full = {"recursive1_proof_0", "recursive2_proof_01"}
    for rel, name in files:
        process(os.path.join(REF, rel), name, 43 if name in full else 6)
