"""bench.py's CPU-baseline leg (oracle/cpu_baseline_avx2.c: hand-vectorised AVX2 + OpenMP restatement, TEST INFRASTRUCTURE) against the
plain restatement the GPU is checked with (oracle/gl_oracle.c).  A baseline that computed something else would be a meaningless
number; this keeps it the same function of the same inputs, bit for bit."""
import ctypes

import numpy as np
import pytest

import glo

P = glo.P
u64 = ctypes.c_uint64


def _has_avx2():
    try:
        return "avx2" in open("/proc/cpuinfo").read()
    except OSError:
        return False


pytestmark = pytest.mark.skipif(not _has_avx2(), reason="host CPU without AVX2")


def test_vector_permutation_equals_the_oracle_on_1e5_states():
    L = glo.lib("baseline")
    rng = np.random.default_rng(11)
    n = 100_000
    st = glo.rand_fe(rng, (n, 12))
    st[:4] = [np.zeros(12), np.arange(12), np.full(12, P - 1), np.full(12, P - 2)]
    st[4:8] = rng.integers(P, 1 << 64, size=(4, 12), dtype=np.uint64)          # non-canonical encodings in
    got = np.ascontiguousarray(st.copy())
    L.glb_poseidon_perm_batch(glo.ptr(got.reshape(-1)), u64(n))
    for i in range(n):
        assert np.array_equal(got[i], glo.perm(st[i])), i      # (the oracle takes any encoding as well)


@pytest.mark.parametrize("ncols,nrows", [(0, 8), (3, 8), (4, 16), (5, 8), (8, 8), (9, 16), (18, 32), (665, 8), (24, 64), (7, 2), (13, 1)])
def test_vector_merkletree_equals_the_oracle(ncols, nrows):
    L = glo.lib("baseline")
    rng = np.random.default_rng(ncols * 100 + nrows)
    src = glo.rand_fe(rng, (nrows, ncols)) if ncols else np.zeros((nrows, 0), dtype=np.uint64)
    nodes = np.zeros((2 * nrows - 1) * 4, dtype=np.uint64)
    L.glb_merkletree(glo.ptr(nodes), glo.ptr(np.ascontiguousarray(src).reshape(-1)) if ncols else None, u64(ncols), u64(nrows))
    assert np.array_equal(nodes, glo.merkletree(src, ncols, nrows))


@pytest.mark.parametrize("log_n,blow,ncols", [(0, 1, 3), (1, 1, 5), (3, 1, 4), (6, 1, 9), (10, 1, 13), (8, 2, 6), (5, 0, 7), (12, 1, 2)])
def test_vector_lde_equals_the_oracle(log_n, blow, ncols):
    L = glo.lib("baseline")
    n, n_ext = 1 << log_n, 1 << (log_n + blow)
    rng = np.random.default_rng(log_n * 10 + ncols)
    src = glo.rand_fe(rng, (n, ncols))
    src[0, 0] = P - 1
    out = np.zeros(n_ext * ncols, dtype=np.uint64)
    L.glb_extend_pol(glo.ptr(out), glo.ptr(np.ascontiguousarray(src).reshape(-1)), u64(n_ext), u64(n), u64(ncols))
    assert np.array_equal(out.reshape(n_ext, ncols), glo.extend_pol(src, n_ext, n, ncols))
