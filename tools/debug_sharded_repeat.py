"""Repeated Starks::genProof with sharded commits against the oracle prover: which proof of a sequence first differs, under a few switches.
usage: python tools/debug_sharded_repeat.py   (GPU box; writes gpurun_out/debug_sharded_repeat.log)"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "merlin-zkevm-prover_amd"))
import test_genproof_parity as tp
import starks_runner as sr

out = open(os.path.join(ROOT, "gpurun_out", "debug_sharded_repeat.log"), "a")
def log(*a):
    print(*a, file=out, flush=True); print(*a, flush=True)

cases = [("zkevm_14", "0,0,0,0", {}), ("zkevm_14", "0,0,0,0", {"MI_MULTI_NO_POOL": "1"}), ("zkevm_14", "0,0", {}), ("zkevm_small", "0,0,0,0", {}),
         ("zkevm_14", "0,0,0,0", {"MI_STARK_ROW_SHARDED": "1"}), ("zkevm_full_12", "0,0,0,0,0,0,0,0", {"MI_TEST_CHILD_TIMEOUT": "150"}), ("zkevm_14", "", {}),
         ("zkevm_full_12", "0,0,0,0,0,0,0,0", {"MI_TEST_CHILD_TIMEOUT": "150", "MI_STARK_ROW_SHARDED": "1"}), ("recursive_12", "0,0,0,0", {"MI_STARK_ROW_SHARDED": "1"})]
if len(sys.argv) > 1:
    cases = [cases[int(i)] for i in sys.argv[1].split(",")]
for name, devs, extra in cases:
    inputs, want = tp.shaped_case(name)
    with tempfile.TemporaryDirectory() as d:
        t0 = time.time()
        try:
            got = sr.gen_proof_on_device(*inputs, workdir=d, batches=(4, 4, 4), env=dict(os.environ, MI_STARK_DEVICES=devs, **extra))
            res = ["same" if g == want else tp.first_difference(g, want)[:90] for g in got]
        except Exception as e:
            res = ["EXC " + str(e)[-5000:]]
        log(name, devs, extra, "%.0fs" % (time.time() - t0), res)
