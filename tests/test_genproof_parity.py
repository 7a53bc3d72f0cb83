"""BASELINE config 4's substitute as a PARITY test: `Starks::genProof` of the product class (host/starks.hpp, on the GPU) against an
oracle-side genProof (tests/oracle_genproof.py: a second reading of starks.cpp:9-668 + friProve.cpp on the CPU oracle's primitives)
from the same starkinfo.json / constant polynomials / constant-tree file / witness / generated tables -- the two zkin.json texts must
be equal BYTE FOR BYTE: the four roots, every evaluation, every opened row and sibling of the five commitment trees and of the FRI
step trees, the final polynomial.  That pins, against the reference's text rather than against this repo's own verifier: the
transcript order, which cm_n index a lookup's h1 / h2 and a grand product's z land in (starks.cpp:405-553), exp2pol, the evMap
order (starks.cpp:555-668), the quotient split, xDivXSub, the constant polynomials re-extended on the device against the tree file
the reference reads them from, FRI's fold / transposition / query indexing.

CPU side (-m "not gpu"): the oracle prover's proof of the mini STARK is accepted by the independent verifier of tests/ministark.py
and a tampered witness is not -- so the checker is not just self-consistent."""
import json
import os
import time
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import glo
import ministark as ms
import oracle_genproof as og
import starks_runner as sr

# bench_starks.py shapes small enough for the oracle: the zkEVM's SHAPE (two lookups of each dimension, six grand products, 24
# evaluations of committed / constant / quotient polynomials, blow-up 2) and the recursive STARKs' (no stage-2 columns, blow-up 8, seven
# quotient chunks, 13 grand products, per-row Steps).  tools/chelpers_precompile.py puts their kernels into the in-tree cache.
ZKEVM_SMALL = ["--log-n", "12", "--widths", "37", "20", "40", "--tmpexp", "60", "--n-const", "11", "--n-evals", "24", "--n-queries", "16", "--n-lookups", "2", "2",
               "--n-products", "6", "--field-ops", "200", "300", "400", "1500", "700"]
ZKEVM_14 = ["--log-n", "14", "--widths", "96", "40", "71", "--tmpexp", "110", "--n-const", "30", "--n-evals", "200", "--n-queries", "32", "--n-lookups", "3", "3",
            "--n-products", "12", "--field-ops", "300", "700", "900", "3000", "1200"]
RECURSIVE_SMALL = ["--log-n", "10", "--ext-bits", "3", "--qdeg", "7", "--widths", "18", "0", "39", "--tmpexp", "90", "--n-const", "9", "--n-evals", "30",
                   "--n-queries", "8", "--n-lookups", "0", "0", "--n-products", "13", "--fri-steps", "13", "9", "5", "--field-ops", "0", "300", "0", "900", "400"]
RECURSIVE_12 = ["--log-n", "12", "--ext-bits", "3", "--qdeg", "7", "--widths", "18", "0", "39", "--tmpexp", "14", "--n-const", "52", "--n-evals", "118",
                "--n-queries", "43", "--n-lookups", "0", "0", "--n-products", "1", "--fri-steps", "15", "11", "7", "4", "--field-ops", "0", "201", "1761", "3483", "463"]
ZKEVM_FULL_12 = ["--log-n", "12"]        # every count of the zkEVM (665 / 128 / 371 / 265 columns, 218 constants, 1 768 evaluations, 21 lookups, 30 products, 128 queries)
SHAPES = {"zkevm_full_12": ZKEVM_FULL_12, "zkevm_small": ZKEVM_SMALL, "zkevm_14": ZKEVM_14, "recursive_small": RECURSIVE_SMALL, "recursive_12": RECURSIVE_12}


def mini_inputs(nbits, n_queries):
    n = 1 << nbits
    lay = ms.Layout(n, 2 * n)
    si = ms.starkinfo(nbits, n_queries)
    progs = {"step2prev": ms.stage2_program(lay), "step3prev": ms.stage3_program(lay), "step3": ms.step3_program(lay),
             "step42ns": ms.step42ns_program(lay, 2), "step52ns": ms.step52ns_program(lay)}
    const_n = ms.constants(n)
    return si, progs, const_n, sr.const_tree_image(const_n, 3, nbits, nbits + 1), ms.witness(n), ms.PUBLICS.copy()


def shaped_inputs(argv, fast=False):
    import bench_starks as b
    a = b.parse(argv)
    si, progs, secs, off, cols = b.shape(a)
    n = 1 << a.log_n
    witness = glo.splitmix64(0x5EED0104, n * cols["cm1_n"])
    const_n = glo.splitmix64(0x5EED0204, n * a.n_const)
    tree = sr.const_tree_image(const_n, a.n_const, a.log_n, a.log_n + a.ext_bits, fast=fast)
    return si, progs, const_n, tree, witness, np.arange(1, 9, dtype=np.uint64)


_ORACLE = {}


def shaped_case(name):
    """(inputs, the oracle prover's zkin) of a named shape, once per session: three GPU tests prove each shape (one device, per-row steps,
    sharded commits) against the same CPU proof."""
    if name not in _ORACLE:
        inputs = shaped_inputs(SHAPES[name])
        _ORACLE[name] = (inputs, og.gen_proof(*inputs)[0])
    return _ORACLE[name]


def first_difference(a, b):
    """where two zkin texts part, as a key path: for the failure message."""
    ja, jb = json.loads(a), json.loads(b)
    if list(ja.keys()) != list(jb.keys()):
        return "keys %s vs %s" % (list(ja.keys()), list(jb.keys()))
    for k in ja:
        if ja[k] != jb[k]:
            xa, xb = np.array(ja[k], dtype=object), np.array(jb[k], dtype=object)
            if xa.shape != xb.shape:
                return "%s: shape %s vs %s" % (k, xa.shape, xb.shape)
            bad = np.argwhere(xa != xb)
            return "%s: %d of %d entries differ, first at %s: %s vs %s" % (k, len(bad), xa.size, tuple(bad[0]), xa[tuple(bad[0])], xb[tuple(bad[0])])
    return "same JSON values, different text"


# ------------------------------------------------------------------ CPU: the checker itself
@pytest.mark.parametrize("nbits,n_queries", [(7, 8), (10, 12)])
def test_oracle_prover_is_accepted_by_the_independent_verifier(nbits, n_queries):
    si, progs, const_n, tree, witness, publics = mini_inputs(nbits, n_queries)
    zkin, _ = og.gen_proof(si, progs, const_n, tree, witness, publics)
    z = json.loads(zkin)
    assert list(z.keys())[:5] == ["root1", "root2", "root3", "root4", "evals"] and list(z.keys())[-2:] == ["finalPol", "publics"]
    const_root = tree[-4:]
    proof = ms.proof_from_zkin(z, nbits)
    proof["const_root"] = const_root
    ok, why = ms.verify(proof, const_root, n_queries=n_queries)
    assert ok, why
    bad = witness.copy()
    bad[5, 0] = (int(bad[5, 0]) + 1) % glo.P                     # a row that breaks the recurrence: the quotient is no polynomial any more
    z2 = json.loads(og.gen_proof(si, progs, const_n, tree, bad, publics)[0])
    proof2 = ms.proof_from_zkin(z2, nbits)
    proof2["const_root"] = const_root
    ok, why = ms.verify(proof2, const_root, n_queries=n_queries)
    assert not ok


def test_oracle_prover_on_the_shaped_starks_is_internally_consistent():
    """zkEVM-shaped and recursive-shaped synthetic STARKs (no satisfiable system: nothing verifies them) through the oracle prover: openings
    climb to their roots at the replayed transcript's indices and the zkin has the reference's keys (stage 2 absent without columns)."""
    for name in ("zkevm_small", "recursive_small"):
        si, progs, const_n, tree, witness, publics = shaped_inputs(SHAPES[name])
        zkin, dbg = og.gen_proof(si, progs, const_n, tree, witness, publics)
        z = json.loads(zkin)
        assert ("s0_vals2" in z) == (si["mapSectionsN"]["cm2_n"] > 0)
        U = lambda x: np.array(x, dtype=object).astype(np.uint64)
        for q, idx in enumerate(dbg["ys"][:4]):
            for t, root in (("1", "root1"), ("3", "root3"), ("4", "root4")):
                assert glo.merkle_verify(U(z[root]), U(z["s0_vals" + t][q]).reshape(-1), U(z["s0_siblings" + t][q]).reshape(-1), idx)
            assert glo.merkle_verify(tree[-4:], U(z["s0_valsC"][q]).reshape(-1), U(z["s0_siblingsC"][q]).reshape(-1), idx)


# ------------------------------------------------------------------ GPU: the product against the oracle prover, byte for byte
@pytest.mark.gpu
@pytest.mark.parametrize("nbits,n_queries", [(10, 12), (13, 24)])
def test_starks_genproof_equals_the_oracle_prover_mini_stark(nbits, n_queries, tmp_path):
    """The mini STARK (a real AIR: lookup, permutation, stage-2 column): nrowsStepBatch 4 (tables) and 1 (generated per-row code recorded
    and run on the device) against the oracle prover."""
    inputs = mini_inputs(nbits, n_queries)
    want, _ = og.gen_proof(*inputs)
    so = sr.steps_library(inputs[1], str(tmp_path))
    got4, got1 = sr.gen_proof_on_device(*inputs, workdir=str(tmp_path), batches=(4, 1), steps_so=so, env=dict(os.environ, MI_CHELPERS_CACHE=str(tmp_path)))
    assert got4 == want, first_difference(got4, want)
    assert got1 == want, first_difference(got1, want)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["zkevm_small", "zkevm_14", "recursive_small", "recursive_12", "zkevm_full_12"])
def test_starks_genproof_equals_the_oracle_prover_shaped_starks(name, tmp_path):
    """bench_starks.py's synthetic STARKs at sizes the oracle finishes in seconds: lookups of both dimensions, grand products of the three
    kinds, evaluations of committed / constant / quotient polynomials at xi and w xi, blow-up 2 and 8, a stage without columns."""
    inputs, want = shaped_case(name)
    got4, = sr.gen_proof_on_device(*inputs, workdir=str(tmp_path), batches=(4,))
    assert got4 == want, first_difference(got4, want)


@pytest.mark.gpu
@pytest.mark.parametrize("name,layout", [("zkevm_small", {"MI_STARK_TILED_EXT": "0"}), ("recursive_12", {"MI_STARK_TILED_EXT": "0"}),
                                         ("zkevm_14", {"MI_STARK_TILED_EXT": "0", "MI_STARK_TILED_WITNESS": "0", "MI_STARK_TILED_CONSTS": "0"}), ("zkevm_small", {"MI_STARK_TILED_WITNESS": "0"}),
                                         ("recursive_small", {"MI_STARK_TILED_CONSTS": "0"})])
def test_starks_genproof_with_a_row_major_image_equals_the_oracle_prover(name, layout, tmp_path):
    """The image's layout is the device's own business (host/starks.hpp: the witness, the wide extended sections and the constants
    tile-major, the rows of a tile bit-reversed): with any of them kept row-major (the switches a maintainer has; what a multi-device
    proof uses) the proof is the same proof."""
    inputs, want = shaped_case(name)
    got, = sr.gen_proof_on_device(*inputs, workdir=str(tmp_path), batches=(4,), env=dict(os.environ, **layout))
    assert got == want, first_difference(got, want)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["zkevm_small", "recursive_small"])
def test_starks_genproof_with_per_row_steps_equals_the_oracle_prover(name, tmp_path):
    """nrowsStepBatch = 1 as the reference proves c12a / recursive1 / recursive2 (prover.cpp:577,611): the tables written out as generated
    per-row C++, recorded by host/steps_tracer.hpp and run on the device; the same bytes as the oracle prover over the tables."""
    inputs, want = shaped_case(name)
    so = sr.steps_library(inputs[1], str(tmp_path))
    got1, = sr.gen_proof_on_device(*inputs, workdir=str(tmp_path), batches=(1,), steps_so=so)
    assert got1 == want, first_difference(got1, want)


@pytest.mark.gpu
@pytest.mark.parametrize("name,devices", [("zkevm_small", "0,0"), ("zkevm_14", "0,0,0,0"), ("recursive_12", "0,0"), ("zkevm_full_12", "0,0,0,0,0,0,0,0")])
def test_starks_genproof_with_sharded_commits_equals_the_oracle_prover(name, devices, tmp_path):
    """MI_STARK_DEVICES: the stage commits of Starks::genProof sharded over several devices from ONE process (csrc/multi.hip; here logical
    shards on device 0): column-tile LDEs, peer exchange, row-sharded leaf hashing and subtrees, openings whose siblings come from the
    shards -- and still the oracle prover's bytes."""
    inputs, want = shaped_case(name)
    got4, again = sr.gen_proof_on_device(*inputs, workdir=str(tmp_path), batches=(4, 4), env=dict(os.environ, MI_STARK_DEVICES=devices))
    assert got4 == want, first_difference(got4, want)
    assert again == want, first_difference(again, want)                 # (the second proof takes its shard buffers from the pool)
    assert not sr.LAST_CHECK["enabled"] or (sr.LAST_CHECK["checks"] > 100 and sr.LAST_CHECK["violations"] == 0), sr.LAST_CHECK   # logical-shard discipline held


@pytest.mark.gpu
@pytest.mark.parametrize("name,devices,grouped", [("zkevm_small", "0,0", False), ("zkevm_14", "0,0,0,0", False), ("recursive_12", "0,0,0,0", False), ("zkevm_full_12", "0,0,0,0,0,0,0,0", False),
                                                  ("zkevm_14", "0,0,0,0", True), ("recursive_12", "0,0,0,0", True), ("zkevm_full_12", "0,0,0,0,0,0,0,0", True)])
def test_starks_genproof_with_row_sharded_step42ns_equals_the_oracle_prover(name, devices, grouped, tmp_path):
    """MI_STARK_ROW_SHARDED (the default when MI_STARK_DEVICES names distinct devices; forced here on logical shards of device 0): every
    shard but the first evaluates step42ns and step52ns over ITS rows of the extended domain -- from a full-height mirror on its device
    of which the stage commits wrote only those rows and the halo its shifted reads reach (blow-up 2: two rows; recursive_12: eight, and
    the last shard's wrap to row 0), its own extension of the constants, its own x_2ns and x / (x - xi) tables, through its own compiled
    programs -- and sends its q and f rows home; the evaluation map is summed per device over its rows and the shares added.  Still the oracle prover's bytes, twice (the second proof reuses the shards' memory and programs).
    Round 5: every device's image -- this one's too -- is an ADDRESS RANGE with memory under its own rows only (mi_vmm_*), the commits are
    TRANSIENT (rows written once, into the images), and with `grouped` the shards of the one physical device form a device group: one set
    of streams and buffers, the proving key's tables held once (what lets eight shards rehearse the zkEVM's full size on one GPU).
    Ungrouped, MI_MULTI_CHECK holds every buffer, program and event to its logical shard."""
    inputs, want = shaped_case(name)
    log = str(tmp_path / "row_shards.log")
    got4, again = sr.gen_proof_on_device(*inputs, workdir=str(tmp_path), batches=(4, 4),
                                         env=dict(os.environ, MI_STARK_DEVICES=devices, MI_STARK_ROW_SHARDED="1", MI_STARK_ROW_SHARD_LOG=log, MI_MULTI_GROUP_SAME_DEVICE="1" if grouped else "0"))
    assert got4 == want, first_difference(got4, want)
    assert again == want, first_difference(again, want)
    assert not sr.LAST_CHECK["enabled"] or (sr.LAST_CHECK["checks"] > 100 and sr.LAST_CHECK["violations"] == 0), sr.LAST_CHECK
    G = len(devices.split(","))
    n_ext = 1 << inputs[0]["starkStruct"]["nBitsExt"]
    lines = open(log).read().split("\n")[:-1]
    assert len(lines) == 2 * 2 * (G - 1)                                 # both steps, every shard but this device's, in both proofs
    assert sorted(set(lines)) == sorted("step%dns shard %d device 0 rows %d %d" % (st, g, g * n_ext // G, (g + 1) * n_ext // G) for g in range(1, G) for st in (42, 52))


@pytest.mark.gpu
@pytest.mark.parametrize("name,devices", [("zkevm_small", "0,0"), ("recursive_12", "0,0,0,0")])
def test_row_sharded_proof_with_dense_images_equals_the_oracle_prover(name, devices, tmp_path):
    """MI_STARK_SPARSE_IMAGE=0: the image and the row-shard mirrors as plain allocations instead of address ranges (what bench.py's
    single-process leg asks for on real devices at first contact: hipMalloc + hipDeviceEnablePeerAccess); the commits are transient all
    the same and borrow base-domain sections as scratch.  Still the oracle prover's bytes."""
    inputs, want = shaped_case(name)
    got4, again = sr.gen_proof_on_device(*inputs, workdir=str(tmp_path), batches=(4, 4),
                                         env=dict(os.environ, MI_STARK_DEVICES=devices, MI_STARK_ROW_SHARDED="1", MI_STARK_SPARSE_IMAGE="0"))
    assert got4 == want, first_difference(got4, want)
    assert again == want, first_difference(again, want)
    assert not sr.LAST_CHECK["enabled"] or (sr.LAST_CHECK["checks"] > 100 and sr.LAST_CHECK["violations"] == 0), sr.LAST_CHECK


@pytest.mark.gpu
def test_row_sharded_and_per_row_proofs_alternate_on_one_starks(tmp_path):
    """One Starks, several devices configured, proofs of both kinds in turn: the table steps (row shards active: this device's image holds
    only ITS rows of the extension, the others are opened from the devices that hold them) and recorded per-row steps (everything on this
    device: the commits send the whole extension here again).  Each proof must not see what the other kind left in the image."""
    inputs, want = shaped_case("zkevm_small")
    so = sr.steps_library(inputs[1], str(tmp_path))
    got = sr.gen_proof_on_device(*inputs, workdir=str(tmp_path), batches=(4, 1, 4, 1), steps_so=so,
                                 env=dict(os.environ, MI_STARK_DEVICES="0,0,0,0", MI_STARK_ROW_SHARDED="1"))
    for i, g in enumerate(got):
        assert g == want, (i, first_difference(g, want))


def test_fast_oracle_prover_is_the_oracle_prover():
    """The larger parity case below lets the oracle prover build its trees and extensions with the vectorised restatement
    (oracle/cpu_baseline_avx2.c); on a shape both finish quickly the two give the same bytes."""
    inputs = shaped_inputs(SHAPES["zkevm_small"])
    assert og.gen_proof(*inputs)[0] == og.gen_proof(*inputs, fast=True)[0]
    assert np.array_equal(inputs[3], shaped_inputs(SHAPES["zkevm_small"], fast=True)[3])


@pytest.mark.gpu
def test_starks_genproof_equals_the_oracle_prover_at_2p18_rows_full_zkevm_shape(tmp_path):
    """Every count of the zkEVM (665 / 128 / 371 / 265 columns, 218 constants, 1 768 evaluations, 21 lookups, 30 grand products, 128 queries,
    the five programs at their real sizes) at 2^18 rows -> 2^19 (round 5: was 2^16; the oracle prover takes about 105 s of the box's 16 threads), byte for byte.  MI_PARITY_LOG_N
    runs it at another size -- 2^22 once per round, recorded under profiles/ -- and MI_PARITY_DEVICES / MI_PARITY_ROW_SHARDED /
    MI_PARITY_GROUPED prove it with sharded commits, row shards and device groups as well."""
    log_n = int(os.environ.get("MI_PARITY_LOG_N", "18"))          # a one-off at a larger size: profiles/r04_genproof_parity_large.txt
    t0 = time.time()
    inputs = shaped_inputs(["--log-n", str(log_n)], fast=True)
    t1 = time.time()
    # MI_PARITY_ORACLE_FILE: the oracle prover's text is kept there (the inputs are seeded: the same on every run), so that further device
    # configurations at a large size -- sharded commits, row shards -- are compared without spending the oracle's minutes again
    keep = os.environ.get("MI_PARITY_ORACLE_FILE", "")
    if keep and os.path.exists(keep):
        want = open(keep).read()
    else:
        import threading
        stop = threading.Event()

        def heartbeat():                      # a large run is minutes of silent CPU work: a runner that watches the output must see it is alive
            while not stop.wait(60):
                print("... oracle prover at work, %.0f s" % (time.time() - t1), flush=True)
        hb = threading.Thread(target=heartbeat, daemon=True)
        hb.start()
        try:
            want, _ = og.gen_proof(*inputs, fast=True)
        finally:
            stop.set()
        if keep:
            open(keep, "w").write(want)
        if keep and os.environ.get("MI_PARITY_ORACLE_ONLY") == "1":   # a size whose oracle proof and device proof do not fit one time slot
            print("oracle prover at 2^%d rows: inputs %.0f s, proof %.0f s, %d bytes -> %s" % (log_n, t1 - t0, time.time() - t1, len(want), keep), flush=True)
            pytest.skip("the oracle prover's proof is saved; run again without MI_PARITY_ORACLE_ONLY to compare")
    t2 = time.time()
    env = dict(os.environ)
    if os.environ.get("MI_PARITY_DEVICES"):
        env.update(MI_STARK_DEVICES=os.environ["MI_PARITY_DEVICES"], MI_STARK_ROW_SHARDED=os.environ.get("MI_PARITY_ROW_SHARDED", "1"),
                   MI_MULTI_GROUP_SAME_DEVICE=os.environ.get("MI_PARITY_GROUPED", "1"))
    got4, = sr.gen_proof_on_device(*inputs, workdir=str(tmp_path), batches=(4,), env=env)
    print("full zkEVM shape at 2^%d rows: inputs %.0f s, oracle prover %.0f s, Starks::genProof (child process, incl. compiling) %.0f s, zkin.json %d bytes, equal: %s"
          % (log_n, t1 - t0, t2 - t1, time.time() - t2, len(want), got4 == want))
    assert got4 == want, first_difference(got4, want)
