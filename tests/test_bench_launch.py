"""bench.py's launcher contract on a box that cannot satisfy --gpus N: a parent that never touches a GPU, a clear message and a
non-zero exit code (here: no GPU at all; on the 1-GPU box the same path refuses --gpus 2)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_n_without_enough_gpus_exits_non_zero_with_a_message():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64", "--steps", "1"], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 2, (r.returncode, r.stderr[-500:])
    assert "needs 64 GPUs" in r.stderr and r.stdout.strip() == ""


def _bench():
    sys.path.insert(0, ROOT)
    import bench
    return bench


def test_a_hung_child_leg_is_killed_at_its_limit_and_never_waited_for():
    """bench.py's optional legs (the batch proof at N = 1; the single-process commit and the multi-device proof at N > 1) run under
    run_child: a leg that hangs -- here a child that ignores SIGTERM and sleeps, with a grandchild in its process group -- is killed as a
    GROUP at its limit, the call returns within seconds of it, and the verdict says so; a leg whose budget is spent is skipped."""
    import time
    b = _bench()
    code = ("import signal, subprocess, sys, time\n"
            "signal.signal(signal.SIGTERM, signal.SIG_IGN)\n"
            "subprocess.Popen([sys.executable, '-c', 'import time; time.sleep(300)'])\n"
            "print('started', flush=True)\n"
            "time.sleep(300)\n")
    t0 = time.time()
    rc, so, se, note = b.run_child([sys.executable, "-c", code], 6.0)
    dt = time.time() - t0
    assert rc not in (0, None) and "killed at its 6 s limit" in note and "started" in so
    assert dt < 20, dt
    rc, so, se, note = b.run_child([sys.executable, "-c", "print('x')"], 2.0)
    assert rc is None and note.startswith("skipped")
    rc, so, se, note = b.run_child([sys.executable, "-c", "import sys; print('fine'); sys.exit(3)"], 30.0)
    assert rc == 3 and so.strip() == "fine" and note == ""


def test_kernel_families_are_matched_by_exact_name(tmp_path):
    """genproof.kernel_rooflines sums a profile's kernels per family by their EXACT demangled base name: k_ntt_pass_pers (the opt-in
    persistent pass) is not k_ntt_pass, a namespaced or templated name is its base name (lstrip("void ") strips characters)."""
    b = _bench()
    p = tmp_path / "stats.csv"
    p.write_text('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\n'
                 '"void k_ntt_pass<8, false, 5, true>(NttPass)",10,2000000000,1,1,1,1,1\n'
                 '"void k_ntt_pass_pers<false>(NttPass)",10,9000000000,1,1,1,1,1\n'
                 '"void k_lde_mid<7, 1, 5>(LdeMid)",4,1000000000,1,1,1,1,1\n'
                 '"void chp::k_chp_linear<3>(chp::LinSlabD const*)",2,400000000,1,1,1,1,1\n'
                 '"chelpers_chunk",7,600000000,1,1,1,1,1\n'
                 '"void odd_k_evmap_partial(int)",1,5000000000,1,1,1,1,1\n'
                 '"k_evmap_partial(unsigned long long*)",2,60000000,1,1,1,1,1\n')
    zk = {"n": 1 << 23, "n_ext": 1 << 24, "n_evals": 1768, "widths": {"cm1": 665, "cm2": 128, "cm3": 371, "cm4": 6, "tmpexp": 265, "const": 218}}
    out = b.proof_kernel_rooflines(str(p), 2, zk)
    assert abs(out["ntt_passes"]["ms_per_proof"] - 1500.0) < 1e-6 and out["ntt_passes"]["launches_per_proof"] == 7
    assert abs(out["linear_kernel"]["ms_per_proof"] - 200.0) < 1e-6
    assert abs(out["constraint_kernels"]["ms_per_proof"] - 300.0) < 1e-6
    assert abs(out["evmap"]["ms_per_proof"] - 30.0) < 1e-6
    assert b.shape_is_zkevm_default("2^23 rows, sections [665, 128, 371] / tmpExp 265 / 218 constants") and not b.shape_is_zkevm_default("2^20 rows, sections [665, 128, 371]")
