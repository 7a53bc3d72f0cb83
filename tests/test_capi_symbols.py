"""CPU-side checks of the product library: it loads, exports every symbol include/mi_stark.h declares,
refuses to compute without a GPU, and the inline device arithmetic (run on the host through the
mi_dbg_host_* hooks) agrees bit-for-bit with the oracle."""
import ctypes, os, re
import numpy as np
import pytest
import glo
import mi_stark

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
P = glo.P


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "mi_stark.h")).read()
    declared = set(re.findall(r"\b(mi_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"mi_merkle_num_nodes_elems", "mi_merkle_proof_levels"}   # static inline helpers
    L = mi_stark.lib()
    missing = [s for s in sorted(declared) if not hasattr(L, s)]
    assert not missing, missing
    assert declared == set(mi_stark.EXPORTS), declared ^ set(mi_stark.EXPORTS)
    assert b"gfx950" in L.mi_version()


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = ctypes.c_void_p()
    st = mi_stark.lib().mi_ctx_create(ctypes.byref(h), 0)
    assert st == -1 and not h.value                      # MI_ERR_NO_DEVICE
    assert b"no CPU fallback" in mi_stark.lib().mi_last_error()
    with pytest.raises(mi_stark.MiStarkError):
        mi_stark.Context(0)


@pytest.mark.parametrize("variant", [0, 1, 2])
def test_device_poseidon_math_on_host_matches_oracle(variant):
    rng = np.random.default_rng(variant)
    cases = [np.zeros(12, dtype=np.uint64), np.arange(12, dtype=np.uint64), np.full(12, P - 1, dtype=np.uint64),
             np.full(12, 2**64 - 1, dtype=np.uint64),                 # non-canonical encodings
             np.array([P, P + 1, 2**64 - 1, 0, 1, 2, 3, 2**32 - 1, 2**32, 2**63, P - 2, 5], dtype=np.uint64)]
    cases += [np.array([(1 << ((k + 5 * j) % 64)) for j in range(12)], dtype=np.uint64) for k in range(64)]
    cases += [glo.rand_fe(rng, 12, canonical=False) for _ in range(200)]
    for c in cases:
        assert np.array_equal(mi_stark.dbg_host_permute(c, variant), glo.perm(c))


def test_device_field_math_on_host_matches_oracle():
    rng = np.random.default_rng(7)
    L = glo.lib()
    edge = [0, 1, 2, P - 1, P - 2, 2**32 - 1, 2**32, 2**32 + 1, 2**48, 2**63, 2**64 - 1, P, P + 1]
    for a in edge:
        for b in edge:
            assert mi_stark.dbg_host_mul(a, b) == L.glo_mul(a, b)
    for _ in range(2000):
        a, b = (int(v) for v in rng.integers(0, 1 << 64, size=2, dtype=np.uint64))
        assert mi_stark.dbg_host_mul(a, b) == L.glo_mul(a, b)
    for _ in range(100):
        a, b = glo.rand_fe(rng, 3), glo.rand_fe(rng, 3)
        assert np.array_equal(mi_stark.dbg_host_e3_mul(a, b), glo.e3_mul(a, b))
        assert np.array_equal(mi_stark.dbg_host_e3_inv(a), glo.e3_inv(a))
    assert list(mi_stark.dbg_host_e3_inv(np.zeros(3, dtype=np.uint64))) == [0, 0, 0]


@pytest.mark.parametrize("log_size", [0, 1, 2, 3, 4])
def test_device_register_dft_on_host_matches_oracle(log_size):
    rng = np.random.default_rng(log_size)
    n = 1 << log_size
    for inverse in (False, True):
        x = glo.rand_fe(rng, n)
        got = mi_stark.dbg_host_dft(x, log_size, inverse)
        want = glo.dft_naive(x, inverse=inverse)
        if inverse:                                         # dft_reg is unscaled
            want = np.array([int(v) * n % P for v in want], dtype=np.uint64)
        assert np.array_equal(got, want)


def test_header_is_plain_c():
    """The drop-in boundary is a C ABI: include/mi_stark.h must compile as C99 (what cgo / JNI / ctypes-style bindings consume)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["gcc", "-x", "c", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-fsyntax-only",
                           os.path.join(root, "include", "mi_stark.h")])
