"""The drop-in claim, tested: reference translation units of the hot path compile UNCHANGED against the Level-0 header
shims (merlin-zkevm-prover_amd/host/) standing in for the absent src/goldilocks submodule.  `g++ -fsyntax-only` on the
files where they lie under /root/reference (nothing is copied); skipped where the reference is not present (GPU box).

What cannot be tried here and why is listed in INTEGRATION.md: everything that includes stark_info.hpp / friProof.hpp needs
nlohmann/json.hpp, which this image does not have."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "merlin-zkevm-prover_amd", "host")
REF = "/root/reference/src"

# (file, why it is on the path)
COMPILES = [
    ("starkpil/transcript/transcript.cpp", "Transcript over PoseidonGoldilocks::hash_full_result (transcript.cpp:23,46)"),
    ("starkpil/merkleTree/merkleTreeGL.cpp", "MerkleTreeGL::merkelize -> PoseidonGoldilocks::merkletree_avx (merkleTreeGL.cpp:37-44)"),
    ("starkpil/zhInv.cpp", "ZhInv tables over Goldilocks::{w, shift, inv}"),
    # not self-contained in the reference either: it uses zkassert, which its includers bring in before it (stark_info.hpp:8-10)
    ("starkpil/polinomial.hpp", "Polinomial view, ext-field element ops, batchInverse*; compiled with -include zkassert.hpp"),
    ("starkpil/transcript/transcript.hpp", ""),
    ("starkpil/merkleTree/merkleTreeGL.hpp", "HASH_SIZE must come through poseidon_goldilocks.hpp (merkleTreeGL.hpp:4-5,63)"),
    ("starkpil/zhInv.hpp", ""),
    ("starkpil/commit_pols_starks.hpp", ""),
    ("starkpil/constant_pols_starks.hpp", ""),
]
# need a third-party header this image lacks: recorded so that the list in INTEGRATION.md stays honest
BLOCKED = [
    ("starkpil/fri/friProve.cpp", "nlohmann/json.hpp"),
    ("starkpil/fri/proof2zkinStark.cpp", "nlohmann/json.hpp"),
]

pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="/root/reference not present")


def _syntax_only(rel):
    inc = [HOST] + [os.path.join(REF, d) for d in ("starkpil", "starkpil/merkleTree", "starkpil/transcript", "starkpil/fri", "utils", "config")]
    pre = ["-include", os.path.join(REF, "utils", "zkassert.hpp")] if rel.endswith("polinomial.hpp") else []
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-fopenmp", "-x", "c++"] + [a for d in inc for a in ("-I", d)] + pre + [os.path.join(REF, rel)]
    return subprocess.run(cmd, capture_output=True, text=True)


@pytest.mark.parametrize("rel,why", COMPILES)
def test_reference_file_compiles_against_the_shims(rel, why):
    r = _syntax_only(rel)
    assert r.returncode == 0, r.stderr[-2000:]


@pytest.mark.parametrize("rel,missing", BLOCKED)
def test_blocked_files_fail_only_for_the_missing_third_party_header(rel, missing):
    r = _syntax_only(rel)
    assert r.returncode != 0
    first = [ln for ln in r.stderr.splitlines() if "error" in ln][0]
    assert "nlohmann" in first, first


def test_reference_bodies_run_beside_the_oracle(tmp_path):
    """The reference's own polinomial.hpp -- calculateH1H2_opt1 / _opt3 / calculateH1H2_ and calculateZ (polinomial.hpp:303-607), header-only
    -- compiled against the Level-0 field classes and RUN beside the oracle's restatements on the same tables (distinct rows, runs of
    equal rows, one heavy value; closing and non-closing products): identical h1 / h2 / z; and ZhInv::ZhInv (zhInv.cpp, the reference's translation unit) beside glo_zhinv; MerkleTreeGL::getGroupProof (merkleTreeGL.cpp) over an oracle-built node array beside glo_merkle_group_proof.  The GPU kernels equal the oracle bit for bit
    (tests/test_lookup.py), so this is reference code deciding their conventions (no longer "parity unpinned" for the
    lookup columns and grand products).  compare_fe.cpp is the reference's too; zklog / exit_process are the stand-ins (the reference's
    drag in utils.hpp and with it gmp / json)."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import glo
    glo.build()
    inc = tmp_path / "ref_inc"
    inc.mkdir()
    os.symlink(os.path.join(REF, "starkpil", "polinomial.hpp"), inc / "polinomial.hpp")
    os.symlink(os.path.join(REF, "utils", "compare_fe.hpp"), inc / "compare_fe.hpp")
    os.symlink(os.path.join(REF, "starkpil", "zhInv.hpp"), inc / "zhInv.hpp")
    os.symlink(os.path.join(REF, "utils", "zkassert.hpp"), inc / "zkassert.hpp")
    os.symlink(os.path.join(REF, "starkpil", "merkleTree", "merkleTreeGL.hpp"), inc / "merkleTreeGL.hpp")
    exe = str(tmp_path / "ref_polinomial")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = ["g++", "-std=c++17", "-O1", "-fopenmp", "-include", os.path.join(REF, "utils", "zkassert.hpp"), "-I", str(inc), "-I", HOST, "-I", os.path.join(HOST, "standalone"),
           os.path.join(root, "tests", "cpp", "test_ref_polinomial.cpp"), os.path.join(REF, "utils", "compare_fe.cpp"), os.path.join(REF, "starkpil", "zhInv.cpp"), os.path.join(REF, "starkpil", "merkleTree", "merkleTreeGL.cpp"), "-o", exe,
           "-L", os.path.join(root, "oracle"), "-lgl_oracle", "-Wl,-rpath," + os.path.join(root, "oracle"),
           "-L", os.path.join(root, "merlin-zkevm-prover_amd"), "-lmi_stark", "-Wl,-rpath," + os.path.join(root, "merlin-zkevm-prover_amd"), "-Wl,-rpath,/opt/rocm/lib", "-L/opt/rocm/lib", "-lamdhip64"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(r.stdout[-3000:], r.stderr[-2000:])
    assert r.returncode == 0 and r.stdout.strip().endswith("OK")
    assert r.stdout.count(": 0 differences") == 24 + 9 + 6 + 5


def test_reference_transcript_state_machine_runs_beside_the_oracle(tmp_path):
    """transcript.{hpp,cpp} of the reference, compiled unchanged over a CPU permutation (the oracle's: tests/cpp/cpu_poseidon), through random
    interleavings of put / getField / getFields1 / getPermutations beside glo_transcript_*: identical values.  The golden proofs cannot
    replay a transcript (the verification key that enters it is absent); this pins the state machine -- buffering of 8, the
    out cursor, 63 bits per field for the query indices -- with the reference's code.  The product's Transcript (hashes on the GPU) is
    compared with the oracle's in the GPU tests."""
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import glo
    glo.build()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "ref_transcript")
    cmd = ["g++", "-std=c++17", "-O1", "-I", os.path.join(root, "tests", "cpp", "cpu_poseidon"), "-I", os.path.join(REF, "starkpil", "transcript"), "-I", HOST,
           os.path.join(root, "tests", "cpp", "test_ref_transcript.cpp"), os.path.join(REF, "starkpil", "transcript", "transcript.cpp"), "-o", exe,
           "-L", os.path.join(root, "oracle"), "-lgl_oracle", "-Wl,-rpath," + os.path.join(root, "oracle")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0 and r.stdout.strip().endswith("OK") and " 0 differences" in r.stdout
