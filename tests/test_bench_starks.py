"""bench_starks.py (one Starks::genProof through libmi_starks.so) at the small size whose kernels tools/chelpers_precompile.py puts
into the in-tree cache: the proof is internally consistent (openings climb to their roots at the replayed transcript's indices, folds
land on the next layer) and sampled rows of q_2ns / f_2ns read back from the device image equal the oracle interpreters' results."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--log-n", "12", "--widths", "37", "20", "40", "--tmpexp", "60", "--n-const", "11", "--n-evals", "24", "--n-queries", "16", "--n-lookups", "2", "2",
         "--n-products", "6", "--field-ops", "200", "300", "400", "1500", "700"]


def test_shape_is_a_consistent_starkinfo():
    sys.path.insert(0, ROOT)
    import bench_starks as b
    for argv in (SMALL, ["--shape", "recursive1"]):
        a = b.parse(argv)
        si, progs, secs, off, cols = b.shape(a)
        n, ne = 1 << a.log_n, 1 << (a.log_n + a.ext_bits)
        assert si["mapTotalN"] == sum(cols[k] * (ne if k.endswith("2ns") else n) for k in b.ORDER)
        for v in si["varPolMap"]:                           # every polynomial lies inside its section
            assert v["sectionPos"] + v["dim"] <= cols[v["section"]], v
        assert len(si["cm_n"]) == len(si["cm_2ns"]) and all(si["varPolMap"][p]["section"].endswith("_n") for p in si["cm_n"])
        for pu in si["puCtx"]:
            assert str(pu["fExpId"]) in si["exp2pol"] and str(pu["numId"]) in si["exp2pol"] and str(pu["denId"]) in si["exp2pol"]
        assert len(si["evMap"]) == a.n_evals and len(si["qs"]) == a.qdeg


@pytest.mark.gpu
@pytest.mark.parametrize("argv", [SMALL, ["--log-n", "10", "--ext-bits", "3", "--qdeg", "7", "--widths", "18", "0", "39", "--tmpexp", "90", "--n-const", "9",
                                          "--n-evals", "30", "--n-queries", "8", "--n-lookups", "0", "0", "--n-products", "13", "--fri-steps", "13", "9", "5",
                                          "--field-ops", "0", "300", "0", "900", "400"]])
def test_starks_flow_proof_is_consistent_and_rows_match_the_oracle(argv, tmp_path):
    env = dict(os.environ, MI_BENCH_TMP=str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench_starks.py")] + argv + ["--check-rows", "8"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    for k, v in out["checks"].items():
        a, b = v.split("/")
        assert a == b and int(b) > 0, (k, v)
    assert out["hbm"]["peak_hbm_gb"] < out["hbm"]["total_gb"] and "STARK_STEP_3_CALCULATE_EXPS_2" in out["phase_ms"]


@pytest.mark.gpu
def test_starks_flow_at_zkevm_size_fits_one_gpu_and_rows_match_the_oracle(tmp_path):
    """The whole flow at the zkEVM's shape (2^23 rows, 665 / 128 / 371 committed columns, blow-up 2, the statistics-matched step42ns
    program of tests/chelpers_programs.ZKEVM_STEP42NS_FIT at 2^24 rows): one genProof, the proof's openings and folds are consistent,
    sampled rows of q_2ns / f_2ns equal the oracle interpreters', and the low-water mark of free HBM stays above zero on ONE GPU."""
    import torch
    if torch.cuda.get_device_properties(0).total_memory < 300e9:
        pytest.skip("needs the MI355X's 288 GiB")
    env = dict(os.environ, MI_BENCH_TMP=str(tmp_path))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench_starks.py"), "--proofs", "1", "--check-rows", "4"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    for k, v in out["checks"].items():
        a, b = v.split("/")
        assert a == b and int(b) > 0, (k, v)
    assert out["hbm"]["fits_one_gpu"] and out["hbm"]["peak_hbm_gb"] < out["hbm"]["total_gb"]
    assert out["config"]["field_ops_per_row"]["step42ns"] == 17986
    for ph in ("STARK_STEP_2_CALCULATE_EXPS", "STARK_STEP_3_CALCULATE_EXPS", "STARK_STEP_3_CALCULATE_EXPS_2", "STARK_STEP_4_CALCULATE_EXPS_2NS"):
        assert out["phase_ms"][ph] > 0


@pytest.mark.gpu
def test_starks_flow_with_generated_per_row_steps_recorded_on_the_device(tmp_path):
    """nrowsStepBatch = 1 as the reference proves the recursive STARKs: bench_starks.py writes the five programs out as generated per-row C++,
    builds a Steps library from it, and Starks::genProof records the functions and runs the recordings on the device; the sampled q_2ns /
    f_2ns rows must equal the oracle interpreters' results on the TABLES the C++ was generated from, and the proof must be consistent."""
    env = dict(os.environ, MI_BENCH_TMP=str(tmp_path))
    env.pop("MI_STEPS_ON_HOST", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench_starks.py")] + SMALL + ["--per-row-steps", "--check-rows", "8"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert "nrowsStepBatch 1" in out["flow"]
    for k, v in out["checks"].items():
        a, b = v.split("/")
        assert a == b and int(b) > 0, (k, v)
