"""The drop-in boundary at the top: class Starks / ZkevmSteps with the reference's signatures (host/starks.hpp, host/chelpers_steps.hpp).

* GPU: tests/cpp/test_starks_genproof.cpp constructs a Starks from files exactly as prover.cpp:128-132 does (starkinfo.json, constant
  polynomials, constant tree, pAddress with the witness), calls genProof(fproof, publics, &zkevmSteps) as prover.cpp:541-544 does, and
  writes zkin.json through proof2zkinStark; the independent verifier of tests/ministark.py must accept it -- with the device steps
  (nrowsStepBatch 4) and with the caller's generated per-row code (nrowsStepBatch 1: recorded and run on the device, or -- on request --
  run on the host), which must give the same proof.
* CPU: the call shapes of prover.cpp compile against host/; the reference's own steps.hpp / zkevmSteps.hpp and its five generated
  tables compile into the translation unit that replaces the *.parser.cpp files (where the reference is present).
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "merlin-zkevm-prover_amd"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "merlin-zkevm-prover_amd", "host")
REF = "/root/reference/src"

import glo
import ministark as ms

LINK = ["-L", os.path.join(ROOT, "merlin-zkevm-prover_amd"), "-lmi_stark", "-L", os.path.join(ROOT, "oracle"), "-lgl_oracle",
        "-Wl,-rpath," + os.path.join(ROOT, "merlin-zkevm-prover_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-Wl,-rpath,/opt/rocm/lib",
        "-L/opt/rocm/lib", "-lamdhip64"]
STANDALONE = ["-I", HOST, "-I", os.path.join(HOST, "standalone")]


def build_exe(tmp_path):
    """-> the test executable, built into the test's own directory (nothing is written into the source tree)."""
    glo.build()
    exe = str(tmp_path / "test_starks_genproof")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-Wall", "-fopenmp"] + STANDALONE + [os.path.join(ROOT, "tests", "cpp", "test_starks_genproof.cpp"), "-o", exe] + LINK)
    return exe


def write_inputs(d, nbits, n_queries):
    """What the reference reads from its config directory: <stark>.starkinfo.json, .const, the committed polynomials, and the generated
    chelpers tables (here as binary files: the C++ side has no generated headers of this AIR)."""
    n = 1 << nbits
    lay = ms.Layout(n, 2 * n)
    si = ms.starkinfo(nbits, n_queries)
    json.dump(si, open(os.path.join(d, "mini.starkinfo.json"), "w"))
    json.dump(si["starkStruct"], open(os.path.join(d, "mini.starkstruct.json"), "w"))
    ms.constants(n).astype(np.uint64).tofile(os.path.join(d, "mini.const"))
    ms.witness(n).astype(np.uint64).tofile(os.path.join(d, "mini.commit"))
    ms.PUBLICS.tofile(os.path.join(d, "mini.publics"))
    for name, (ops, args) in (("step2prev", ms.stage2_program(lay)), ("step3prev", ms.stage3_program(lay)), ("step3", ms.step3_program(lay)),
                              ("step42ns", ms.step42ns_program(lay, 2)), ("step52ns", ms.step52ns_program(lay))):
        ops.tofile(os.path.join(d, name + ".ops"))
        args.tofile(os.path.join(d, name + ".args"))


def test_starks_program_compiles_and_links(tmp_path):
    assert os.path.exists(build_exe(tmp_path))


def test_prover_call_shapes_compile_against_host(tmp_path):
    """prover.cpp:128-132 and :541-552, as text, against host/starks.hpp + zkevmSteps.hpp (syntax only; standalone stand-ins for the
    prover's config / utils headers)."""
    tu = tmp_path / "prover_shapes.cpp"
    tu.write_text('''
#include "starks.hpp"
#include "zkevmSteps.hpp"
#include "proof2zkinStark.hpp"
#define NROWS_STEPS_ 4
Starks *starkZkevm;
void construct(const Config &config, void *pAddress)
{
    starkZkevm = new Starks(config, {config.zkevmConstPols, config.mapConstPolsFile, config.zkevmConstantsTree, config.zkevmStarkInfo}, pAddress);
    starkZkevm->nrowsStepBatch = NROWS_STEPS_;
}
void batchProof(Goldilocks::Element (&publics)[48])
{
    ZkevmSteps zkevmSteps;
    uint64_t polBits = starkZkevm->starkInfo.starkStruct.steps[starkZkevm->starkInfo.starkStruct.steps.size() - 1].nBits;
    FRIProof fproof((1 << polBits), FIELD_EXTENSION, starkZkevm->starkInfo.starkStruct.steps.size(), starkZkevm->starkInfo.evMap.size(), starkZkevm->starkInfo.nPublics);
    starkZkevm->genProof(fproof, &publics[0], &zkevmSteps);
    auto jProof = fproof.proofs.proof2json();
    auto zkin = proof2zkinStark(fproof);
    (void)jProof; (void)zkin;
}
''')
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-fopenmp"] + STANDALONE + [str(tu)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


@pytest.mark.skipif(not os.path.isdir(REF), reason="/root/reference not present")
def test_reference_steps_headers_and_tables_compile_into_the_replacement_unit(tmp_path):
    """The reference's OWN steps.hpp and zkevmSteps.hpp (found before the stand-ins), its five generated tables, and the translation unit
    INTEGRATION.md gives in place of the five *.parser.cpp files: compiled to an object, every batched ZkevmSteps entry point defined."""
    tu = os.path.join(ROOT, "merlin-zkevm-prover_amd", "host", "zkevm_steps_device.cpp")
    obj = str(tmp_path / "zkevm_steps_device.o")
    # the reference's headers of this path, by symbolic link (not the whole src/starkpil: its stark_info.hpp needs nlohmann/json, which this
    # image lacks -- a maintainer's include path has all of src/starkpil and the third-party headers)
    inc_dir = tmp_path / "ref_inc"
    inc_dir.mkdir()
    links = {"steps.hpp": "starkpil/steps.hpp", "zkevmSteps.hpp": "starkpil/zkevm/chelpers/zkevmSteps.hpp", "zhInv.hpp": "starkpil/zhInv.hpp",
             "constant_pols_starks.hpp": "starkpil/constant_pols_starks.hpp", "zkassert.hpp": "utils/zkassert.hpp"}
    for st in ("2prev", "3prev", "3", "42ns", "52ns"):
        links["zkevm.chelpers.step%s.parser.hpp" % st] = "starkpil/zkevm/chelpers/zkevm.chelpers.step%s.parser.hpp" % st
    for name, rel in links.items():
        os.symlink(os.path.join(REF, rel), inc_dir / name)
    inc = ["-I", str(inc_dir), "-I", HOST, "-I", os.path.join(HOST, "standalone")]
    r = subprocess.run(["g++", "-std=c++17", "-O0", "-c", "-fopenmp", "-mavx2"] + inc + [tu, "-o", obj], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    syms = subprocess.run(["nm", "-C", "--defined-only", obj], capture_output=True, text=True).stdout
    for step, flavours in (("step2prev", ["_avx"]), ("step3prev", ["_avx"]), ("step3", ["", "_avx", "_avx_jump"]), ("step42ns", ["", "_avx", "_avx_jump"]),
                           ("step52ns", ["", "_avx"])):
        for fl in flavours:
            assert "ZkevmSteps::%s_parser_first%s(StepsParams&" % (step, fl) in syms, (step, fl)
    for table in ("op2prev", "args3prev", "op3", "args42", "op52"):
        assert any(ln.split()[-1] == table for ln in syms.splitlines() if ln.strip()), table
    # it is the reference's class declarations that were used, not the stand-ins
    pre = subprocess.run(["g++", "-std=c++17", "-E", "-fopenmp", "-mavx2"] + inc + [tu], capture_output=True, text=True).stdout
    assert str(inc_dir / "zkevmSteps.hpp") in pre and str(inc_dir / "steps.hpp") in pre and "standalone/steps.hpp" not in pre


@pytest.mark.skipif(not os.path.isdir(REF), reason="/root/reference not present")
@pytest.mark.parametrize("rel", ["starkRecursive1/chelpers/recursive1.chelpers.step2.cpp", "starkRecursive1/chelpers/recursive1.chelpers.step3prev.cpp",
                                 "starkRecursive1/chelpers/recursive1.chelpers.step3.cpp", "starkRecursive1/chelpers/recursive1.chelpers.step52ns.cpp",
                                 "zkevm/chelpers/zkevm.chelpers.step2.cpp"])
def test_generated_per_row_steps_compile_against_level0(tmp_path, rel):
    """The per-row forms a Steps class consists of (generated C++: Goldilocks::add / Goldilocks3::mul ... over params.pols) are the
    caller's host code in the host-steps mode of Starks::genProof; the reference's own files compile unchanged against host/."""
    inc_dir = tmp_path / "ref_inc"
    inc_dir.mkdir()
    for name, r in {"steps.hpp": "starkpil/steps.hpp", "zkevmSteps.hpp": "starkpil/zkevm/chelpers/zkevmSteps.hpp",
                    "recursive1Steps.hpp": "starkpil/starkRecursive1/chelpers/recursive1Steps.hpp", "zhInv.hpp": "starkpil/zhInv.hpp",
                    "constant_pols_starks.hpp": "starkpil/constant_pols_starks.hpp", "zkassert.hpp": "utils/zkassert.hpp"}.items():
        os.symlink(os.path.join(REF, r), inc_dir / name)
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-fopenmp", "-I", str(inc_dir), "-I", HOST, "-I", os.path.join(HOST, "standalone"),
                        os.path.join(REF, "starkpil", rel)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def build_exe_generated_rows(tmp_path, nbits):
    """The same caller with ZkevmSteps' per-row forms as GENERATED C++ (tests/gen_steps_cpp.py writes the AIR's five programs out in the
    style of the reference's recursive1.chelpers.*.cpp): what Starks::genProof records and runs on the device when nrowsStepBatch is 1."""
    import gen_steps_cpp as gs
    n = 1 << nbits
    lay = ms.Layout(n, 2 * n)
    progs = {"step2prev": ms.stage2_program(lay), "step3prev": ms.stage3_program(lay), "step3": ms.step3_program(lay),
             "step42ns": ms.step42ns_program(lay, 2), "step52ns": ms.step52ns_program(lay)}
    rows = tmp_path / "generated_rows.inc"
    rows.write_text(gs.steps_source("ZkevmSteps", progs))
    exe = str(tmp_path / "test_starks_genproof_rows")
    glo.build()
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-fopenmp", "-DMI_TEST_GENERATED_ROWS=\"%s\"" % rows] + STANDALONE +
                          [os.path.join(ROOT, "tests", "cpp", "test_starks_genproof.cpp"), "-o", exe] + LINK)
    return exe


def check_proofs(d, nbits, n_queries, names):
    const_root = np.array(json.load(open(os.path.join(d, "mini.verkey.json")))["constRoot"], dtype=np.uint64)
    for name in names:
        z = json.load(open(os.path.join(d, name)))
        proof = ms.proof_from_zkin(z, nbits)
        proof["const_root"] = const_root
        ok, why = ms.verify(proof, const_root, n_queries=n_queries)
        assert ok, (name, why)
        # ... and the verifier is not a rubber stamp for this path either
        proof["evals"][3 * ms.EV_B] ^= np.uint64(1)
        ok, why = ms.verify(proof, const_root, n_queries=n_queries)
        assert not ok and "constraint identity" in why


@pytest.mark.gpu
@pytest.mark.parametrize("nbits,n_queries", [(10, 12), (13, 24)])
def test_starks_genproof_is_accepted_by_the_independent_verifier(tmp_path, nbits, n_queries):
    """nrowsStepBatch 4 (the tables on the device), then 1 with generated per-row code: recorded (host/steps_tracer.hpp), compiled and
    run on the device; the two proofs are the same bytes and the independent verifier accepts them."""
    exe = build_exe_generated_rows(tmp_path, nbits)
    d = str(tmp_path)
    write_inputs(d, nbits, n_queries)
    env = dict(os.environ, MI_CHELPERS_CACHE=os.path.join(d, "cache"))
    env.pop("MI_STEPS_ON_HOST", None)
    os.makedirs(env["MI_CHELPERS_CACHE"], exist_ok=True)
    r = subprocess.run([exe, d, "4", "1"], capture_output=True, text=True, timeout=900, env=env)
    print(r.stdout[-3000:], r.stderr[-3000:])
    assert r.returncode == 0 and "ALL OK" in r.stdout
    check_proofs(d, nbits, n_queries, ("zkin.json", "zkin.1.json"))


@pytest.mark.gpu
def test_starks_genproof_with_the_callers_rows_on_the_host(tmp_path):
    """MI_STEPS_ON_HOST=1: per-row code the recorder cannot follow (here the oracle's interpreters: plain C arithmetic) runs as in the
    reference, on the host over pAddress, between copies of the sections it reads and writes; same proof.  Without the variable such a
    Steps class is refused, not silently skipped."""
    nbits, n_queries = 10, 12
    EXE = build_exe(tmp_path)
    d = str(tmp_path)
    write_inputs(d, nbits, n_queries)
    env = dict(os.environ, MI_CHELPERS_CACHE=os.path.join(d, "cache"), MI_STEPS_ON_HOST="1")
    os.makedirs(env["MI_CHELPERS_CACHE"], exist_ok=True)
    r = subprocess.run([EXE, d, "4", "1"], capture_output=True, text=True, timeout=900, env=env)
    print(r.stdout[-3000:], r.stderr[-3000:])
    assert r.returncode == 0 and "ALL OK" in r.stdout
    check_proofs(d, nbits, n_queries, ("zkin.json", "zkin.1.json"))
    env.pop("MI_STEPS_ON_HOST")
    r = subprocess.run([EXE, d, "1"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode != 0 and "recorded nothing" in (r.stdout + r.stderr)


@pytest.mark.gpu
def test_a_starks_whose_hbm_plan_does_not_fit_stops_with_the_reason(tmp_path):
    """One proof's device image lives in ONE arena (host/starks.hpp mi::Arena); a STARK whose plan exceeds the device -- here 2^24 rows x
    3 000 witness columns: 403 GB for cm1_n alone -- must stop in the constructor with the plan's size and the free memory, the
    reference's convention (message + exit), not with a HIP error somewhere inside a proof."""
    import bench_starks as b
    a = b.parse(["--log-n", "10", "--widths", "3000", "20", "40", "--tmpexp", "60", "--n-const", "1", "--n-evals", "8", "--n-queries", "8", "--n-lookups", "1", "1",
                 "--n-products", "4", "--field-ops", "0", "0", "0", "100", "100"])
    si, *_ = b.shape(a)
    # the same map at 2^24 rows: only the sizes matter, nothing is proved
    nbits, n, ne = 24, 1 << 24, 1 << 25
    si["starkStruct"]["nBits"], si["starkStruct"]["nBitsExt"] = nbits, nbits + 1
    si["starkStruct"]["steps"] = [{"nBits": 25}, {"nBits": 20}, {"nBits": 15}, {"nBits": 10}, {"nBits": 6}]
    o = 0
    for k in b.ORDER:
        si["mapOffsets"][k] = o
        si["mapDeg"][k] = ne if k.endswith("2ns") else n
        o += si["mapSectionsN"][k] * si["mapDeg"][k]
    si["mapTotalN"] = o
    json.dump(si, open(tmp_path / "big.starkinfo.json", "w"))
    code = """
import ctypes, numpy as np, sys
L = ctypes.CDLL(%r, mode=ctypes.RTLD_GLOBAL)
L.mis_create.restype = ctypes.c_void_p
const_n = np.zeros(1 << 24, dtype=np.uint64)
tree = np.zeros(16, dtype=np.uint64); tree[0], tree[1] = 1, 1 << 25
pa = np.zeros(16, dtype=np.uint64)
L.mis_create(%r, ctypes.c_void_p(const_n.ctypes.data), ctypes.c_void_p(tree.ctypes.data), ctypes.c_void_p(pa.ctypes.data))
print("constructed")
""" % (os.path.join(ROOT, "merlin-zkevm-prover_amd", "libmi_starks.so"), str(tmp_path / "big.starkinfo.json").encode())
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode != 0 and "constructed" not in r.stdout
    assert "HBM plan needs" in r.stderr and "GB free" in r.stderr, r.stderr[-2000:]
