"""End to end: a STARK proof made by the device path verifies; a tampered one does not (tests/ministark.py)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "merlin-zkevm-prover_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import glo
import ministark as ms


def test_the_air_is_satisfied_by_its_witness_and_its_programs_decode():
    """CPU: the witness satisfies the row constraints, d is a permutation of c, every f is in the table, and the programs are
    well-formed tables."""
    import chelpers_programs as cp
    n = 1 << 6
    w, c = ms.witness(n).astype(object), ms.constants(n).astype(object)
    P = ms.P
    for i in range(n):
        a, b, an, bn = w[i][0], w[i][1], w[(i + 1) % n][0], w[(i + 1) % n][1]
        l1, ll = c[i][0], c[i][1]
        assert ((1 - ll) * (an - b)) % P == 0 and ((1 - ll) * (bn - a - b)) % P == 0 and (l1 * (a - 1)) % P == 0 and (l1 * (b - 1)) % P == 0
    assert sorted(w[:, 2]) == sorted(w[:, 3]) and set(w[:, 4]) <= set(c[:, 2]) and len(set(c[:, 2])) < n
    lay = ms.Layout(n, 2 * n)
    ops, args = ms.step42ns_program(lay, 2)
    assert cp.decode(ops, args)[1] == args.size
    ops, args = ms.step52ns_program(lay)
    assert sum(cp.nargs52_of(int(o)) for o in ops) == args.size
    for ops, args in (ms.stage2_program(lay), ms.stage3_program(lay), ms.step3_program(lay)):
        assert cp.decode_base(ops, args)[1] == args.size
    # the starkinfo description agrees with the layout the programs were written against
    si = ms.starkinfo(6)
    assert si["mapTotalN"] == lay.total and si["mapOffsets"]["tmpExp_n"] == lay.off["tmpExp_n"] and len(si["evMap"]) == 21


@pytest.mark.gpu
@pytest.mark.parametrize("native,nbits,lin", [(False, 10, False), (True, 10, False), (True, 10, True), (False, 7, False), (True, 13, True), (True, 19, False)])
def test_a_proof_from_the_device_path_verifies(native, nbits, lin, tmp_path, monkeypatch):
    """lin: the FRI polynomial's polynomial terms through the streaming linear kernel (forced: the AIR has too few of them)."""
    import mi_stark
    if lin:
        monkeypatch.setenv("MI_CHELPERS_LIN_MIN", "1")
    ctx = mi_stark.Context(0)
    proof = ms.prove(ctx, nbits, native=native, cache_dir=str(tmp_path))
    ok, why = ms.verify(proof, proof["const_root"])
    assert ok, why
    # the verification key is part of the statement
    ok, why = ms.verify(proof, np.zeros(4, dtype=np.uint64))
    assert not ok
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("tamper,expect", [("eval", "constraint identity"), ("opening", "Merkle opening of cm1"), ("stage2", "Merkle opening of cm2"), ("stage3", "Merkle opening of cm3"), ("perm", "constraint identity"),
                                           ("h1h2", "constraint identity"),
                                           ("final", "final polynomial"), ("f", "")])
def test_a_tampered_proof_is_rejected(tamper, expect):
    import mi_stark
    ctx = mi_stark.Context(0)
    proof = ms.prove(ctx, 10, n_queries=32, tamper=tamper)
    ok, why = ms.verify(proof, proof["const_root"], n_queries=32)
    assert not ok and expect in why, why
    ctx.close()


@pytest.mark.gpu
def test_a_lookup_of_a_value_outside_the_table_stops_the_prover():
    """Like the reference (polinomial.hpp:321-325, "Number not included"): calculateH1H2 fails, naming the row."""
    import mi_stark
    ctx = mi_stark.Context(0)
    with pytest.raises(mi_stark.MiStarkError, match="number not included: w=512"):
        ms.prove(ctx, 10, tamper="lookup")
    ctx.close()
