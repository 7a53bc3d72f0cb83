"""host/steps_tracer.hpp: a Steps class's generated per-row C++ (the form the reference's recursive STARKs ship, starks.cpp:84-88 runs it row
by row) recorded once and turned into a device program.  tests/cpp/test_steps_tracer.cpp links such a class, records its five `_first`
functions, runs the recorded programs through the library's translator + host executors, runs the functions themselves on the same rows,
and compares everything they wrote.  No GPU.

* the mini STARK's five programs (tests/ministark.py) and a synthetic zkEVM-shaped set, written out as per-row C++ by
  tests/gen_steps_cpp.py -- runs everywhere;
* the reference's own recursive1 / recursive2 / c12a generated files, compiled unchanged against host/ -- where /root/reference is
  present.  Their polynomial maps are read off the offsets and strides in the generated code (SURVEY App. A does the same for the zkEVM):
  recursive1 / recursive2: N = 2^17, blow-up 8, cm1 18 | cm2 0 | cm3 39 | cm4 21 | tmpExp 6 columns; c12a: C12A below.
* the GPU leg (the recorded programs through Starks::genProof with nrowsStepBatch 1, proof accepted by the independent verifier and equal
  to the table-driven one) is in tests/test_starks_class.py.
"""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "merlin-zkevm-prover_amd"))
HOST = os.path.join(ROOT, "merlin-zkevm-prover_amd", "host")
REF = "/root/reference/src/starkpil"
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree not present")

import chelpers_programs as cp
import gen_steps_cpp as gs
import ministark as ms

LINK = ["-L", os.path.join(ROOT, "merlin-zkevm-prover_amd"), "-lmi_stark", "-Wl,-rpath," + os.path.join(ROOT, "merlin-zkevm-prover_amd"), "-Wl,-rpath,/opt/rocm/lib",
        "-L/opt/rocm/lib", "-lamdhip64"]
# (tests/cpp/host_zhinv/zhInv.hpp first: the standalone ZhInv asks the device for its table)
INC = ["-I", os.path.join(ROOT, "tests", "cpp", "host_zhinv"), "-I", os.path.join(ROOT, "include"), "-I", HOST, "-I", os.path.join(HOST, "standalone")]
DRIVER = os.path.join(ROOT, "tests", "cpp", "test_steps_tracer.cpp")

GEN_HEADER = gs.GEN_HEADER


def run_driver(tmp_path, sources, header, cls, layout, extra_inc=(), opt="-O1", defines=(), env=None):
    from concurrent.futures import ThreadPoolExecutor
    exe = str(tmp_path / "tracer_test")
    lay = tmp_path / "layout.txt"
    lay.write_text(" ".join(str(int(v)) for v in layout))
    flags = ["-std=c++17", opt, "-fopenmp", "-DSTEPS_HEADER=\"%s\"" % header, "-DSTEPS_CLASS=%s" % cls] + list(defines) + list(extra_inc) + INC

    def compile_one(k_src):
        k, src = k_src
        obj = str(tmp_path / ("obj%d.o" % k))
        r = subprocess.run(["g++"] + flags + ["-c", src, "-o", obj], capture_output=True, text=True)
        assert r.returncode == 0, (src, r.stderr[-4000:])
        return obj

    with ThreadPoolExecutor(max_workers=6) as pool:          # (the generated files are tens of thousands of lines each)
        objs = list(pool.map(compile_one, enumerate([DRIVER] + list(sources))))
    r = subprocess.run(["g++", "-fopenmp"] + objs + ["-o", exe] + LINK, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-4000:]
    r = subprocess.run([exe, str(lay)], capture_output=True, text=True, timeout=900, env=dict(os.environ, **(env or {})))
    print(r.stdout[-3000:], r.stderr[-2000:])
    return r


def test_generated_per_row_code_of_the_mini_stark_is_recorded_and_reproduced(tmp_path):
    nbits = 7
    n = 1 << nbits
    lay = ms.Layout(n, 2 * n)
    progs = {"step2prev": ms.stage2_program(lay), "step3prev": ms.stage3_program(lay), "step3": ms.step3_program(lay),
             "step42ns": ms.step42ns_program(lay, 2), "step52ns": ms.step52ns_program(lay)}
    (tmp_path / "genSteps.hpp").write_text(GEN_HEADER)
    src = tmp_path / "gen_steps.cpp"
    src.write_text(gs.steps_source("GenSteps", progs, header='#include "genSteps.hpp"\n'))
    si = ms.starkinfo(nbits)
    layout = [nbits, nbits + 1, si["nConstants"], si["nPublics"], len(si["evMap"])] + [ms.Layout.COLS[k] for k in ms.Layout.ORDER]
    r = run_driver(tmp_path, [str(src)], "genSteps.hpp", "GenSteps", layout, extra_inc=["-I", str(tmp_path)])
    assert r.returncode == 0 and r.stdout.strip().endswith("OK")
    for step in ("step2prev", "step3prev", "step3", "step42ns", "step52ns"):
        assert re.search(r"%s: \d+ recorded operations .* 0 differ \(translated\) 0 differ \(lowered\)" % step, r.stdout), step


def synthetic_case(tmp_path, seed, nbits=6, ext=2, widths=(9, 7, 12), tmpexp=10, n_const=6, n_pub=3, n_evals=12, sizes=(120, 200, 40), with3prev=False):
    """One random Steps class in generated per-row C++ (every opcode of the three table formats, shifted reads of 2^ext rows in the
    extended domain, stores at shifted rows) through the driver; returns the driver's output.  Also used by tools/steps_tracer_fuzz.py."""
    rng = np.random.default_rng(seed)
    n, ne = 1 << nbits, 1 << (nbits + ext)
    w1, w2, w3 = widths
    order = ["cm1_n", "cm2_n", "cm3_n", "cm4_n", "tmpExp_n", "cm1_2ns", "cm2_2ns", "cm3_2ns", "cm4_2ns", "q_2ns", "f_2ns"]
    allc = dict(cm1_n=w1, cm2_n=w2, cm3_n=w3, cm4_n=6, tmpExp_n=tmpexp, cm1_2ns=w1, cm2_2ns=w2, cm3_2ns=w3, cm4_2ns=6, q_2ns=3, f_2ns=3)
    off, o = {}, 0
    for k in order:
        off[k] = o
        o += allc[k] * (ne if k.endswith("2ns") else n)
    base_secs = [(off[k], allc[k]) for k in ("cm1_n", "cm2_n")]
    progs = {
        "step2prev": cp.synthetic_program_base(rng, n, base_secs, (off["tmpExp_n"], allc["tmpExp_n"]), n_const, 8, n_pub, n_ops=sizes[0]),
        "step3": cp.synthetic_program_base(rng, n, base_secs, (off["cm3_n"], allc["cm3_n"]), n_const, 8, n_pub, n_ops=sizes[1]),
        "step42ns": cp.synthetic_program(rng, ne, [(off[k], allc[k]) for k in ("cm1_2ns", "cm2_2ns", "cm3_2ns")], n_const, 8, n_pub),
        "step52ns": cp.synthetic_program52(rng, [(off[k], allc[k]) for k in ("cm1_2ns", "cm2_2ns", "cm3_2ns", "cm4_2ns")], n_const, n_evals, length=sizes[2]),
    }
    if with3prev:
        progs["step3prev"] = cp.synthetic_program_base(rng, n, base_secs, (off["tmpExp_n"], allc["tmpExp_n"]), n_const, 8, n_pub, n_ops=sizes[0])
        del progs["step2prev"]               # (both write tmpExp_n: one of them per case)
    (tmp_path / "genSteps.hpp").write_text(GEN_HEADER)
    src = tmp_path / "gen_steps.cpp"
    src.write_text(gs.steps_source("GenSteps", progs, header='#include "genSteps.hpp"\n'))
    layout = [nbits, nbits + ext, n_const, n_pub, n_evals] + [allc[k] for k in order]
    return run_driver(tmp_path, [str(src)], "genSteps.hpp", "GenSteps", layout, extra_inc=["-I", str(tmp_path)]), sorted(progs)


def test_generated_per_row_code_of_synthetic_zkevm_shaped_programs(tmp_path):
    """Every opcode of the three table formats, shifted reads in both domains (blow-up 4: shifts of 4 rows), stores at shifted rows."""
    r, steps = synthetic_case(tmp_path, 5)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK")
    assert "step3prev: empty" in r.stdout
    for step in steps:
        assert re.search(r"%s: \d+ recorded operations .* 0 differ \(translated\) 0 differ \(lowered\)" % step, r.stdout), step


def test_code_the_recorder_cannot_follow_is_refused(tmp_path):
    """Arithmetic through operators leaves no addresses: the recorder must say so instead of recording half a program."""
    (tmp_path / "genSteps.hpp").write_text(GEN_HEADER)
    body = """#include "genSteps.hpp"
void GenSteps::step42ns_first(StepsParams &params, uint64_t i) {
     Goldilocks::Element t = params.pols[%d + i*5] * params.pols[%d + i*5];       // operator form
     Goldilocks3::Element u;
     Goldilocks3::mul(u, t, (Goldilocks3::Element &)*params.challenges[0]);
     Goldilocks3::mul((Goldilocks3::Element &)(params.q_2ns[i * 3]), params.zi.zhInv(i), u);
}
void GenSteps::step52ns_first(StepsParams &params, uint64_t i) {
     Goldilocks3::Element u;
     if (i & 1) Goldilocks3::add(u, (Goldilocks3::Element &)*params.challenges[0], (Goldilocks3::Element &)*params.challenges[1]);   // branches on the row
     else Goldilocks3::sub(u, (Goldilocks3::Element &)*params.challenges[0], (Goldilocks3::Element &)*params.challenges[1]);
     Goldilocks3::copy((Goldilocks3::Element &)(params.f_2ns[i * 3]), u);
}
void GenSteps::step3_first(StepsParams &params, uint64_t i) {
     Goldilocks::Element t;
     Goldilocks::add(t, params.pols[0 + i*5], params.pols[1 + i*5]);
     Goldilocks::copy(params.pols[2 + i*5], t);           // writes cm1_n column 2 ...
     Goldilocks::Element v;
     Goldilocks::mul(v, params.pols[2 + ((i + 1)%%%d)*5], t); // ... and reads it at the next row
     Goldilocks::copy(params.pols[3 + i*5], v);
}
void GenSteps::step3prev_first(StepsParams &params, uint64_t i) {
     Goldilocks::Element t;
     Goldilocks::add(t, params.pols[4 + ((i + 1)%%%d)*5], params.pols[1 + i*5]); // reads cm1_n column 4 at the next row ...
     Goldilocks::copy(params.pols[4 + i*5], t);                                  // ... and writes it afterwards
}
#define E(s) void GenSteps::s(StepsParams &, uint64_t) {}
E(step2prev_first) E(step2prev_i) E(step2prev_last) E(step3prev_i) E(step3prev_last) E(step3_i) E(step3_last)
E(step42ns_i) E(step42ns_last) E(step52ns_i) E(step52ns_last)
"""
    nbits = 5
    lay = ms.Layout(1 << nbits, 2 << nbits)
    src = tmp_path / "bad_steps.cpp"
    src.write_text(body % (lay.off["cm1_2ns"], lay.off["cm1_2ns"] + 1, 1 << nbits, 1 << nbits))
    layout = [nbits, nbits + 1, 4, 2, 4] + [ms.Layout.COLS[k] for k in ms.Layout.ORDER]
    r = run_driver(tmp_path, [str(src)], "genSteps.hpp", "GenSteps", layout, extra_inc=["-I", str(tmp_path)])
    assert r.returncode == 1
    assert re.search(r"step42ns: TRACE FAILED: .*operators or value-returning forms", r.stdout)
    assert re.search(r"step52ns: TRACE FAILED: .*does not compute the same program at row 0 and at the last row", r.stdout)
    assert re.search(r"step3: TRACE FAILED: .*a polynomial the step writes is read back at another row", r.stdout)
    assert re.search(r"step3prev: TRACE FAILED: .*a polynomial the step writes is also read from memory", r.stdout)


# the reference's generated per-row files; polynomial maps read off their offsets / strides (see the module docstring)
RECURSIVE = [17, 20, 52, 48, 118, 18, 0, 39, 21, 6, 18, 0, 39, 21, 3, 3]
# c12a: N = 2^20, blow-up 4: cm1 18 | cm2 0 | cm3 78 | cm4 12 | tmpExp 6 columns (c12a.chelpers.*.cpp: pols[18874368 + i*78], pols[113246208 + i*6],
# pols[119537664 + i*18], pols[195035136 + i*78], pols[522190848 + i*12]; ((i + 4)%4194304)); 52 constants, 44 publics, 146 evaluations
C12A = [20, 22, 52, 44, 146, 18, 0, 78, 12, 6, 18, 0, 78, 12, 3, 3]
REF_STARKS = {
    "c12a": ("starkC12a/chelpers", "c12a", "C12aSteps", "c12aSteps.hpp", C12A),
    "recursive1": ("starkRecursive1/chelpers", "recursive1", "Recursive1Steps", "recursive1Steps.hpp", RECURSIVE),
    "recursive2": ("starkRecursive2/chelpers", "recursive2", "Recursive2Steps", "recursive2Steps.hpp", RECURSIVE),
}


@needs_ref
@pytest.mark.parametrize("name", sorted(REF_STARKS))
def test_the_references_generated_steps_are_recorded_and_reproduced(tmp_path, name):
    """The reference's own generated code, compiled as it is: what the recorder makes of recursive1 / recursive2's ~7 000-operation
    step42ns (and the four other steps) computes, on random rows, exactly what that code computes."""
    d, stem, cls, header, layout = REF_STARKS[name]
    inc = tmp_path / "ref_inc"
    inc.mkdir()
    os.symlink(os.path.join(REF, d, header), inc / header)
    srcs = [os.path.join(REF, d, "%s.chelpers.%s.cpp" % (stem, s)) for s in ("step2", "step3prev", "step3", "step42ns", "step52ns")]
    r = run_driver(tmp_path, srcs, header, cls, layout, extra_inc=["-I", str(inc)], opt="-O2")  # (optimised, as a maintainer's build compiles them)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK")
    assert "step2prev: empty" in r.stdout
    for step in ("step3prev", "step3", "step42ns", "step52ns"):
        m = re.search(r"%s: (\d+) recorded operations .* (\d+) words compared, 0 differ \(translated\) 0 differ \(lowered\)" % step, r.stdout)
        assert m and int(m.group(2)) > 0, step
        if step == "step42ns":
            assert int(m.group(1)) > 3000


# the zkEVM's polynomial map (read off the tables' offsets): N = 2^23, blow-up 2, cm1 665 | cm2 128 | cm3 371 | cm4 6 | tmpExp 265;
# constants up to column 217, evaluations up to index 1767 in the per-row files (counted here as 218 and 1768), 48 publics
ZKEVM = [23, 24, 218, 48, 1768, 665, 128, 371, 6, 265, 665, 128, 371, 6, 3, 3]


@needs_ref
def test_the_zkevm_tables_compute_what_the_zkevm_per_row_code_computes(tmp_path):
    """The zkEVM's constraint system exists twice in the reference: as the generated TABLES the batched interpreters run
    (zkevm.chelpers.step*.parser.hpp) and as generated per-row C++ (zkevm.chelpers.step{2,3prev,3,52ns}.cpp; the step42ns file is an absent
    blob).  Both come from one generator, so they compute the same polynomials.  Here the library's TABLE DECODER (its transcription of
    the interpreters' opcodes: mi_chelpers_compile) is run on the reference's tables beside the reference's compiled per-row code, on
    random rows of the 254 GB map (address space only): every word either stores must agree -- 30 / 79 / ... polynomials of tmpExp_n and
    cm3_n, and f_2ns.  That pins the opcode semantics of four of the five numberings with reference code run here; the recorder's programs
    of the same functions are checked alongside."""
    d = "zkevm/chelpers"
    inc = tmp_path / "ref_inc"
    inc.mkdir()
    for name in ["zkevmSteps.hpp"] + ["zkevm.chelpers.%s.parser.hpp" % s for s in ("step2prev", "step3prev", "step3", "step42ns", "step52ns")]:
        os.symlink(os.path.join(REF, d, name), inc / name)
    srcs = [os.path.join(REF, d, "zkevm.chelpers.%s.cpp" % s) for s in ("step2", "step3prev", "step3", "step52ns")]
    srcs.append(os.path.join(ROOT, "tests", "cpp", "test_steps_tables.cpp"))
    r = run_driver(tmp_path, srcs, "zkevmSteps.hpp", "ZkevmSteps", ZKEVM, extra_inc=["-I", str(inc)], opt="-O0", defines=["-DMI_TEST_WITH_TABLES"],
                   env={"MI_TEST_SKIP_STEPS": "3"})
    assert r.returncode == 0 and r.stdout.strip().endswith("OK")
    assert "step42ns: skipped" in r.stdout
    for step in ("step2prev", "step3prev", "step3", "step52ns"):
        m = re.search(r"%s: (\d+) recorded operations .* (\d+) words compared, 0 differ \(translated\) 0 differ \(lowered\)" % step, r.stdout)
        assert m and int(m.group(2)) > 0, step
        m = re.search(r"%s: the reference's TABLE for this step, decoded and run on the same rows: (\d+) words compared, 0 differ" % step, r.stdout)
        assert m and int(m.group(1)) > 0, step
