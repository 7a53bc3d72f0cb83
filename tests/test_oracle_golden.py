"""Pins the CPU oracle against the reference's own golden STARK proofs.

Fixtures: tests/golden/*.npz, derived by tests/golden/make_golden.py from
testvectors/aggregatedProof/recursive1.zkin.proof_{0..3}.json and
testvectors/finalProof/recursive2.zkin.proof_{01,03,23}.json (key layout proof2zkinStark.cpp:8-82).
Checks: Poseidon KATs, linear_hash, Merkle climb for 7 trees per proof (merkleTreeGL.cpp:12-35),
query-index consistency (friProve.cpp:171-177) and the FRI fold relation of every step
(friProve.cpp:86-104)."""
import glob, os
import numpy as np
import pytest
import glo

FILES = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
BITS = [20, 16, 12, 9, 6]


def test_fixture_count():
    assert len(FILES) == 7


def test_poseidon_kats():
    # SURVEY App. C (naive permutation of poseidon_g_executor.cpp:174-205)
    assert [hex(x) for x in glo.perm(np.zeros(12, dtype=np.uint64))[:4]] == \
        ["0x3c18a9786cb0b359", "0xc4055e3364a246c3", "0x7953db0ab48808f4", "0xc71603f33a1144ca"]
    assert [hex(x) for x in glo.perm(np.arange(12, dtype=np.uint64))[:4]] == \
        ["0xd64e1e3efc5b8e9e", "0x53666633020aaa47", "0xd40285597c6a8825", "0x613a4f81e81231d2"]
    assert [hex(x) for x in glo.perm(np.full(12, glo.P - 1, dtype=np.uint64))[:4]] == \
        ["0xbe0085cfc57a8357", "0xd95af71847d05c09", "0xcf55a13d33c1c953", "0x95803a74f4530e82"]
    assert list(glo.linear_hash(np.arange(1, 19, dtype=np.uint64))) == \
        [17347307666344302174, 15108499987079209467, 16054863494716285791, 13973118912684215897]
    assert list(glo.linear_hash(np.array([1, 2, 3], dtype=np.uint64))) == [1, 2, 3, 0]


def test_roots_of_unity():
    L = glo.lib()
    assert [L.glo_w(i) for i in range(1, 9)] == [glo.P - 1, 1 << 48, 1 << 24, 4096, 64, 8, 2198989700608, 4404853092538523347]
    for n in (3, 9, 12, 16, 20, 24, 32):
        w = L.glo_w(n)
        assert L.glo_pow(w, 1 << n) == 1 and L.glo_pow(w, 1 << (n - 1)) == glo.P - 1


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_merkle_paths(path):
    d = np.load(path)
    nq = len(d["q_index"])
    assert nq in (6, 43)
    for q in range(nq):
        idx = int(d["q_index"][q])
        for t, root in (("1", "root1"), ("3", "root3"), ("4", "root4")):
            assert glo.merkle_verify(d[root], d[f"s0_vals{t}"][q], d[f"s0_siblings{t}"][q], idx)
            # a wrong index or a corrupted leaf must fail
            assert not glo.merkle_verify(d[root], d[f"s0_vals{t}"][q], d[f"s0_siblings{t}"][q], idx ^ 1)
        for s in range(1, 5):
            i_s = idx % (1 << BITS[s])
            assert glo.merkle_verify(d[f"s{s}_root"], d[f"s{s}_vals"][q], d[f"s{s}_siblings"][q], i_s)
        bad = d["s0_vals1"][q].copy()
        bad[0] ^= np.uint64(1)
        assert not glo.merkle_verify(d["root1"], bad, d["s0_siblings1"][q], idx)


@pytest.mark.parametrize("path", FILES, ids=[os.path.basename(f)[:-4] for f in FILES])
def test_fri_fold_relation(path):
    d = np.load(path)
    nq = len(d["q_index"])
    for q in range(nq):
        idx = int(d["q_index"][q])
        for s in range(1, 5):
            prev, cur = BITS[s - 1], BITS[s]
            g = idx % (1 << cur)
            got = glo.fri_fold_group(d[f"s{s}_vals"][q], prev - cur, prev, 20, g, d["special_x"][s - 1])
            if s < 4:
                j = g >> BITS[s + 1]
                want = d[f"s{s + 1}_vals"][q][3 * j:3 * j + 3]
            else:
                want = d["finalPol"][g]
            assert list(got) == list(want), (q, s)


def test_fri_group_layout_matches_transpose():
    """The group opened at index g of step s holds pol[i*2^cur + g], i < nX (friProve.cpp:88-92) and the
    step tree's row j holds aux[j*h..] = pol[k*w + j] (friProve.cpp:252-271): the same thing."""
    rng = np.random.default_rng(5)
    prev, cur = 8, 5
    pol = glo.rand_fe(rng, (1 << prev) * 3)
    aux = glo.fri_transpose(pol, 1 << prev, cur).reshape(1 << cur, -1)
    nx = 1 << (prev - cur)
    for g in (0, 7, 31):
        grp = np.concatenate([pol[3 * (i * (1 << cur) + g):3 * (i * (1 << cur) + g) + 3] for i in range(nx)])
        assert np.array_equal(aux[g], grp)
    x = glo.rand_fe(rng, 3)
    folded = glo.fri_fold(pol, prev, cur, 10, x)
    for g in (0, 7, 31):
        assert np.array_equal(folded[g], glo.fri_fold_group(aux[g], prev - cur, prev, 10, g, x))


@pytest.mark.parametrize("path", FILES)
def test_root2_is_the_root_of_a_tree_without_columns(path):
    """The recursive STARKs commit nothing in stage 2 (no s0_vals2 in the files), yet every golden proof carries a root2: the root of
    MerkleTreeGL(2^20, 0 columns).  linear_hash of an empty row is four zeros (no permutation), so root2 is twenty levels of
    hash(node | node | 0000) starting from zeros -- the same value in all seven files, and a reference-held pin of the node hash and
    of the size <= 4 pass-through rule at size 0."""
    g = np.load(path)
    node = np.zeros(4, dtype=np.uint64)
    for _ in range(20):
        node = glo.perm(np.concatenate([node, node, np.zeros(4, dtype=np.uint64)]))[:4]
    assert np.array_equal(node, g["root2"])
    n = 1 << 7                                           # the tree builder takes the same path on a width of 0
    nodes = glo.merkletree(np.zeros((n, 0), dtype=np.uint64), 0, n)
    want = np.zeros(4, dtype=np.uint64)
    for _ in range(7):
        want = glo.perm(np.concatenate([want, want, np.zeros(4, dtype=np.uint64)]))[:4]
    assert np.array_equal(nodes[-4:], want) and not nodes[:4 * n].any()


@pytest.mark.parametrize("path", FILES)
def test_final_polynomial_is_low_degree_under_the_inverse_transform(path):
    """finalPol of a golden proof is the last FRI layer: 64 evaluations over <w_64> (natural order) of a polynomial of degree
    < 2^(6 - 3) = 8 (blow-up 8).  INTT_64 in the oracle's convention -- natural order in and out, w(6) = 8 -- must therefore give exactly
    eight leading coefficients and 56 zeros on the reference's own data: a transform with another root or ordering smears them (checked:
    the bit-reversed reading of the same data has 64 non-zero coefficients)."""
    g = np.load(path)
    fp = g["finalPol"].reshape(64, 3)
    c = glo.ntt(fp, 64, 3, inverse=True).reshape(64, 3)
    assert c[:8].any(axis=1).all() and not c[8:].any()
    br = [int(format(i, "06b")[::-1], 2) for i in range(64)]
    assert glo.ntt(np.ascontiguousarray(fp[br]), 64, 3, inverse=True).reshape(64, 3)[8:].any()
