"""Algebraic self-checks of the CPU oracle for the parts the golden proofs cannot pin (SURVEY 8c,
"What is not pinned"): large NTT/LDE values, transcript structure, step-4 split, evmap, batch inverse.
BASELINE config 1 (forward NTT over 2^20 x 1 col, CPU, bit-exact) lives here."""
import numpy as np
import pytest
import glo

P = glo.P


def pyntt(col, w):
    """independent O(n log n) recursive Cooley-Tukey on python ints (different algorithm from the oracle's)"""
    n = len(col)
    if n == 1:
        return list(col)
    ev = pyntt(col[0::2], w * w % P)
    od = pyntt(col[1::2], w * w % P)
    out = [0] * n
    t = 1
    for k in range(n // 2):
        x = od[k] * t % P
        out[k] = (ev[k] + x) % P
        out[k + n // 2] = (ev[k] - x) % P
        t = t * w % P
    return out


@pytest.mark.parametrize("n", [1, 2, 4, 64, 256])
def test_ntt_matches_naive_dft(n):
    rng = np.random.default_rng(n)
    x = glo.rand_fe(rng, n)
    assert np.array_equal(glo.ntt(x, n, 1)[:, 0], glo.dft_naive(x))
    assert np.array_equal(glo.ntt(x, n, 1, inverse=True)[:, 0], glo.dft_naive(x, inverse=True))


def test_ntt_multi_column_rowmajor_and_roundtrip():
    rng = np.random.default_rng(1)
    n, c = 512, 7
    x = glo.rand_fe(rng, (n, c))
    X = glo.ntt(x, n, c)
    for j in range(c):
        assert np.array_equal(X[:, j], glo.ntt(np.ascontiguousarray(x[:, j]), n, 1)[:, 0])
    assert np.array_equal(glo.ntt(X, n, c, inverse=True), x)


def test_ntt_identities_and_noncanonical_inputs():
    n = 1024
    imp = np.zeros(n, dtype=np.uint64); imp[0] = 1
    assert np.all(glo.ntt(imp, n, 1) == 1)                       # impulse -> all ones
    w = glo.lib().glo_w(10)
    geo = glo.geom_seq(n, 1, glo.lib().glo_inv(w))               # w^-i -> n * delta_1
    out = glo.ntt(geo, n, 1)[:, 0]
    want = np.zeros(n, dtype=np.uint64); want[1] = n
    assert np.array_equal(out, want)
    x = np.array([P + 5, P, 2**64 - 1, 7] * (n // 4), dtype=np.uint64)   # encodings >= p are accepted
    xc = np.where(x >= np.uint64(P), x - np.uint64(P), x)
    assert np.array_equal(glo.ntt(x, n, 1), glo.ntt(xc, n, 1))
    assert np.all(glo.ntt(x, n, 1) < np.uint64(P))


def test_config1_forward_ntt_2pow20_single_column():
    """BASELINE config 1: forward NTT over 2^20 Goldilocks elements, 1 column -- bit-exact vs an
    independent O(n log n) implementation on python ints (sampled outputs checked by Horner too)."""
    n = 1 << 20
    x = glo.splitmix64(0x5EED0001, n)
    X = glo.ntt(x, n, 1)[:, 0]
    ref = pyntt([int(v) for v in x], glo.lib().glo_w(20))
    assert [int(v) for v in X] == ref
    assert np.array_equal(glo.ntt(X, n, 1, inverse=True)[:, 0], x)


def test_extend_pol_is_coset_evaluation():
    rng = np.random.default_rng(3)
    n, c, ext = 256, 5, 512
    x = glo.rand_fe(rng, (n, c))
    E = glo.extend_pol(x, ext, n, c)
    coef = glo.ntt(x, n, c, inverse=True)
    w_ext = glo.lib().glo_w(9)
    for i in (0, 1, 2, 255, 256, 511):
        pt = 49 * pow(w_ext, i, P) % P                            # shift * w_ext^i (build_const_tree.cpp:160-196)
        for j in range(c):
            acc = 0
            for k in range(n - 1, -1, -1):
                acc = (acc * pt + int(coef[k, j])) % P
            assert acc == int(E[i, j]), (i, j)
    # blow-up 4 also works, and a constant column extends to the same constant
    ones = np.full((n, 1), 9, dtype=np.uint64)
    assert np.all(glo.extend_pol(ones, 4 * n, n, 1) == 9)


def test_ext_field():
    rng = np.random.default_rng(4)
    for _ in range(50):
        a, b = glo.rand_fe(rng, 3), glo.rand_fe(rng, 3)
        # schoolbook with x^3 = x + 1 (polinomial.hpp:195-205 is the Karatsuba form of this)
        c = [0] * 5
        for i in range(3):
            for j in range(3):
                c[i + j] += int(a[i]) * int(b[j])
        want = [(c[0] + c[3]) % P, (c[1] + c[3] + c[4]) % P, (c[2] + c[4]) % P]
        assert [int(v) for v in glo.e3_mul(a, b)] == want
        assert list(glo.e3_mul(a, glo.e3_inv(a))) == [1, 0, 0]
    assert list(glo.e3_inv(np.zeros(3, dtype=np.uint64))) == [0, 0, 0]


def test_merkle_layout_and_group_proof():
    rng = np.random.default_rng(6)
    for h, w in ((1, 5), (2, 3), (8, 9), (64, 18), (32, 4), (16, 1)):
        src = glo.rand_fe(rng, (h, w))
        nodes = glo.merkletree(src, w, h)
        assert nodes.size == (2 * h - 1) * 4
        for r in range(h):
            assert np.array_equal(nodes[4 * r:4 * r + 4], glo.linear_hash(src[r]))
        root = nodes[-4:]
        for idx in {0, h - 1, h // 3}:
            proof = glo.merkle_group_proof(nodes, src, h, w, idx)
            assert np.array_equal(proof[:w], src[idx])
            assert glo.merkle_verify(root, proof[:w], proof[w:], idx)


def test_transcript_structure():
    t = glo.Transcript()
    first = glo.perm(np.zeros(12, dtype=np.uint64))               # empty transcript squeezes perm(0)
    assert [t.get_fields1() for _ in range(12)] == [int(v) for v in first]
    t = glo.Transcript()
    t.put(np.arange(1, 9, dtype=np.uint64))                        # 8 values -> one absorb (transcript.cpp:12-29)
    st = np.zeros(12, dtype=np.uint64); st[:8] = np.arange(1, 9)
    out1 = glo.perm(st)
    assert [int(v) for v in t.get_field()] == [int(v) for v in out1[:3]]
    t.put(np.array([42], dtype=np.uint64))                         # put resets the squeeze cursor
    st2 = np.zeros(12, dtype=np.uint64); st2[0] = 42; st2[8:] = out1[:4]
    assert t.get_fields1() == int(glo.perm(st2)[0])
    # getPermutations: 63 bits per field, LSB first (transcript.cpp:59-87)
    t1, t2 = glo.Transcript(), glo.Transcript()
    t1.put(np.arange(5, dtype=np.uint64)); t2.put(np.arange(5, dtype=np.uint64))
    perms = t1.get_permutations(10, 20)
    nf = (10 * 20 - 1) // 63 + 1
    bits = []
    for _ in range(nf):
        f = t2.get_fields1()
        bits += [(f >> b) & 1 for b in range(63)]
    want = [sum(bits[i * 20 + j] << j for j in range(20)) for i in range(10)]
    assert [int(v) for v in perms] == want


def test_q_split_semantics():
    """starks.cpp:265-280: with q(X) = q0(X) + X^N q1(X) and qq1 = coefficients of q(shift X),
    row k of qq2 holds the k-th coefficients of q0(shift X) and q1(shift X)."""
    rng = np.random.default_rng(8)
    n, qdeg = 64, 2
    qq1 = glo.rand_fe(rng, (2 * n, 3))
    out = glo.q_split(qq1, n, qdeg).reshape(2 * n, qdeg * 3)
    s_in = pow(pow(49, P - 2, P), n, P)
    for k in (0, 1, n - 1):
        for p in range(qdeg):
            f = pow(s_in, p, P)
            assert [int(v) for v in out[k, 3 * p:3 * p + 3]] == [int(v) * f % P for v in qq1[p * n + k]]
    assert not out[n:].any()


def test_batch_inverse_geom_zhinv_evmap():
    rng = np.random.default_rng(9)
    src = glo.rand_fe(rng, (33, 3))
    inv = glo.batch_inverse3(src).reshape(-1, 3)
    for i in range(33):
        assert list(glo.e3_mul(src[i], inv[i])) == [1, 0, 0]
    g = glo.geom_seq(16, 49, glo.lib().glo_w(4))
    assert int(g[0]) == 49 and int(g[5]) == 49 * pow(glo.lib().glo_w(4), 5, P) % P
    z = glo.zhinv(3, 5)                                           # zhInv.cpp:7-31
    sn = pow(49, 8, P)
    for i in range(4):
        assert int(z[i]) * ((sn * pow(glo.lib().glo_w(2), i, P) - 1) % P) % P == 1
    # evmap (starks.cpp:555-668) against a python-int restatement
    n, ext_bits = 32, 1
    cm = glo.rand_fe(rng, (n << ext_bits, 7))                     # 5 base cols + one ext col (3) + pad
    lev, lpev = glo.rand_fe(rng, (n, 3)), glo.rand_fe(rng, (n, 3))
    pols = [(cm, 0, 1, 7), (cm, 3, 1, 7), (cm, 4, 3, 7)]
    prime = [0, 1, 1]
    ev = glo.evmap(pols, prime, lev, lpev, n, ext_bits)
    for i, (arr, off, dim, stride) in enumerate(pols):
        acc = np.zeros(3, dtype=np.uint64)
        L = lpev if prime[i] else lev
        for k in range(n):
            b = arr.ravel()[(k << ext_bits) * stride + off:][:dim]
            t = glo.e3_mul(L[k], np.array([b[0], 0, 0], dtype=np.uint64) if dim == 1 else b)
            acc = np.array([(int(acc[d]) + int(t[d])) % P for d in range(3)], dtype=np.uint64)
        assert np.array_equal(ev[i], acc)
