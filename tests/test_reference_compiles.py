"""The drop-in claim, tested: reference translation units of the hot path compile UNCHANGED against the Level-0 header
shims (merlin-zkevm-prover_amd/host/) standing in for the absent src/goldilocks submodule.  `g++ -fsyntax-only` on the
files where they lie under /root/reference (nothing is copied); skipped where the reference is not present (GPU box).

What cannot be tried here and why is listed in INTEGRATION.md: everything that includes stark_info.hpp / friProof.hpp needs
nlohmann/json.hpp, which this image does not have."""
import os
import subprocess

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
