#!/usr/bin/env python3
"""bench_starks.py -- BASELINE config 4 substitute, through the product's own `class Starks`: ONE Starks::genProof (host/starks.hpp,
behind libmi_starks.so) over a synthetic zkEVM-SHAPED STARK at full size, the way prover.cpp:541-544 calls it.

What is synthetic and why: the zkEVM's starkinfo.json, constant polynomials, witness and generated chelpers tables are not in the
reference tree and its tables may not travel to the GPU box.  This driver generates a STARK of the same SHAPE -- the memory
map of SURVEY App. A (665 / 128 / 371 / 6 / 265 columns over 2^23 rows, 218 constant polynomials, 2^24 extended rows), 21 lookups and
30 grand products described by puCtx / peCtx / ciCtx, 1 768 evaluations, 128 queries, FRI 24/19/14/10/6, and five constraint programs of
the real ones' sizes in the reference's table formats -- writes it as a starkinfo.json + tables, and hands it to Starks exactly like
the mini STARK of the tests.  The programs describe no satisfiable system (the proof is not meant to verify); what is measured is every
phase of genProof at zkEVM size inside the product class, with its HBM plan, and what is CHECKED afterwards is that the proof is
internally consistent: every opening climbs to its root at the index the replayed transcript asks for, every FRI fold lands on the next
layer's opened value, sampled rows of q_2ns / f_2ns equal the oracle interpreters' results over the device image.

Prints one JSON line: total, per-phase milliseconds under the reference's timer names (starks.cpp:12-402), peak HBM (free-memory
low-water mark), the plan, the checks.
"""
import argparse
import ctypes
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "merlin-zkevm-prover_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
ORDER = ["cm1_n", "cm2_n", "cm3_n", "cm4_n", "tmpExp_n", "cm1_2ns", "cm2_2ns", "cm3_2ns", "cm4_2ns", "q_2ns", "f_2ns"]
STEP_ID = {"step2prev": 20, "step3prev": 30, "step3": 31, "step42ns": 42, "step52ns": 52}


def arg_parser():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log-n", type=int, default=23)
    ap.add_argument("--widths", type=int, nargs=3, default=[665, 128, 371], help="columns of cm1, cm2, cm3")
    ap.add_argument("--tmpexp", type=int, default=265, help="columns of tmpExp_n")
    ap.add_argument("--n-const", type=int, default=218, help="constant polynomials (zkEVM: ConstantPols::numPols() = 218, pols_generated/constant_pols.hpp:689)")
    ap.add_argument("--n-evals", type=int, default=1768, help="evaluations (zkEVM: the per-row step52ns reads params.evals[0..1767])")
    ap.add_argument("--n-queries", type=int, default=128)
    ap.add_argument("--n-lookups", type=int, nargs=2, default=[11, 10], help="plookups of dimension 1 and of dimension 3")
    ap.add_argument("--n-products", type=int, default=30, help="grand products (lookups + permutations + connections)")
    ap.add_argument("--field-ops", type=int, nargs=5, default=[1815, 5869, 13158, 17986, 7101],
                    help="field operations per row of the step2prev, step3prev, step3, step42ns, step52ns programs")
    ap.add_argument("--ext-bits", type=int, default=1, help="log2 of the blow-up factor (zkEVM 1; c12a / recursive1 / recursive2: 3)")
    ap.add_argument("--qdeg", type=int, default=2, help="chunks of the quotient polynomial: cm4 has 3 * qdeg columns (recursive STARKs: 7)")
    ap.add_argument("--fri-steps", type=int, nargs="*", default=None, help="FRI layer sizes in bits (default: nBitsExt, then -5, -5, -4, -4)")
    ap.add_argument("--shape", choices=("zkevm", "recursive1", "c12a", "batch"), default="zkevm",
                    help="c12a: bench_starks.C12A; batch: zkevm, c12a and recursive1 one after the other (genBatchProof's three proofs); "
                         "recursive1: the shape the golden proofs testvectors/aggregatedProof/recursive1.zkin.proof_*.json imply (2^17 rows, blow-up 8, "
                         "18 / 0 / 39 / 21 columns, 52 constants, 118 evaluations, 43 queries, FRI 20/16/12/9/6); sets every size argument")
    ap.add_argument("--step42-generic", action="store_true",
                    help="step42ns from the r02 generator defaults instead of the parameters fitted to the real program's statistics "
                         "(tests/chelpers_programs.ZKEVM_STEP42NS_FIT; the fit applies when the three committed sections are all present)")
    ap.add_argument("--per-row-steps", action="store_true",
                    help="nrowsStepBatch = 1, as the reference proves c12a / recursive1 / recursive2 (prover.cpp:577,611): the five programs are written out "
                         "as generated per-row C++ (tests/gen_steps_cpp.py), compiled into a Steps library, and Starks::genProof RECORDS them and runs the "
                         "recordings on the device (host/steps_tracer.hpp).  pAddress is then the whole polynomial map in host memory, as the reference has it")
    ap.add_argument("--proofs", type=int, default=2, help="genProof runs; the last one is reported (the first pays first-touch / code loading)")
    ap.add_argument("--check-rows", type=int, default=6)
    ap.add_argument("--chelpers-batch-rows", type=int, default=0)
    ap.add_argument("--precompile", type=int, nargs=2, default=None, metavar=("SHARD", "NSHARDS"), help="compile the programs' kernels into the cache and exit (no GPU)")
    return ap


# ONE grand product (the connection argument; cm3_n's other 36 columns are the intermediate polynomials step3 writes) and the field
# operations per row are what host/steps_tracer.hpp records from the reference's recursive1.chelpers.*.cpp (tests/test_steps_tracer.py
# prints them: step3prev 201, step3 1761, step42ns 3483, step52ns 463; step2prev empty).  tmpExp_n: 14 columns where the reference has
# 6 (numerator, denominator): the program generator wants spare output columns; the section is never extended or committed.
RECURSIVE1 = ["--log-n", "17", "--ext-bits", "3", "--qdeg", "7", "--widths", "18", "0", "39", "--tmpexp", "14", "--n-const", "52", "--n-evals", "118",
              "--n-queries", "43", "--n-lookups", "0", "0", "--n-products", "1", "--fri-steps", "20", "16", "12", "9", "6",
              "--field-ops", "0", "201", "1761", "3483", "463"]


# c12a (the STARK that wraps the zkEVM proof's verifier circuit, prover.cpp:560-600): the map read off c12a.chelpers.*.cpp as for recursive1
# (2^20 rows, blow-up 4, 18 / 0 / 78 / 12 columns, 52 constants, 146 evaluations; programs of 205 / 1 916 / 3 556 / 603 operations); its
# starkStruct is not in the tree: 64 queries and FRI 22/18/14/10/6 are ASSUMED (blow-up 4 needs about twice recursive1's 43 queries)
C12A = ["--log-n", "20", "--ext-bits", "2", "--qdeg", "4", "--widths", "18", "0", "78", "--tmpexp", "14", "--n-const", "52", "--n-evals", "146",
        "--n-queries", "64", "--n-lookups", "0", "0", "--n-products", "1", "--fri-steps", "22", "18", "14", "10", "6",
        "--field-ops", "0", "205", "1916", "3556", "603"]


def parse(argv=None):
    ap = arg_parser()
    args = ap.parse_args(argv)
    if args.shape == "recursive1":
        args = ap.parse_args(RECURSIVE1 + (list(argv) if argv is not None else sys.argv[1:]))
    if args.shape == "c12a":
        args = ap.parse_args(C12A + (list(argv) if argv is not None else sys.argv[1:]))
    return args


def fri_steps(nbits_ext, explicit=None):
    if explicit:
        assert explicit[0] == nbits_ext
        return list(explicit)
    steps = [nbits_ext]
    for d in (5, 5, 4, 4):
        if steps[-1] - d >= 3:
            steps.append(steps[-1] - d)
    return steps


def shape(args):
    """-> (starkinfo dict, {step: (ops, args)}, sections per step) of the synthetic zkEVM-shaped STARK; deterministic in args."""
    import chelpers_programs as cpg
    nbits, nbits_ext = args.log_n, args.log_n + args.ext_bits
    N, NE = 1 << nbits, 1 << nbits_ext
    w1, w2, w3 = args.widths
    w4 = 3 * args.qdeg
    shift_ext = 1 << args.ext_bits                         # "prime" (next-row) reads in the extended domain
    cols = {"cm1_n": w1, "cm2_n": w2, "cm3_n": w3, "cm4_n": w4, "tmpExp_n": args.tmpexp, "cm1_2ns": w1, "cm2_2ns": w2, "cm3_2ns": w3, "cm4_2ns": w4, "q_2ns": 3, "f_2ns": 3}
    off, o = {}, 0
    for k in ORDER:
        off[k] = o
        o += cols[k] * (NE if k.endswith("2ns") else N)
    total = o
    l1, l3 = args.n_lookups
    n_prod = args.n_products
    assert n_prod >= l1 + l3 and 2 * l1 + 6 * l3 <= w2 and 3 * n_prod <= w3 and l1 + 3 * l3 + 6 * n_prod <= args.tmpexp - 8
    # ---- polynomials (varPolMap): cm1_n columns; cm2_n = h1 / h2 of every lookup (+ spare columns); cm3_n = the grand products z
    # (+ spare columns, which step3 fills); tmpExp_n = the lookups' expression polynomials, then the products' numerators / denominators
    vpm, cm_n = [], []
    def pol(section, dim, pos):
        vpm.append({"section": section, "dim": dim, "sectionPos": pos})
        return len(vpm) - 1
    for c in range(w1):
        cm_n.append(pol("cm1_n", 1, c))
    pu, pos2, posT = [], 0, 0
    exp2pol, next_exp = {}, 1000
    h_ids = []
    for i in range(l1 + l3):
        dim = 1 if i < l1 else 3
        h1, h2 = pol("cm2_n", dim, pos2), pol("cm2_n", dim, pos2 + dim)
        pos2 += 2 * dim
        ft = pol("tmpExp_n", dim, posT)                   # f and t expressions: ONE polynomial serves as both (f is trivially inside t)
        posT += dim
        exp2pol[str(next_exp)] = ft
        pu.append({"tExpId": next_exp, "fExpId": next_exp, "h1Id": h1, "h2Id": h2, "zId": 0, "c1Id": 0, "numId": 0, "denId": 0, "c2Id": 0})
        next_exp += 1
        h_ids += [h1, h2]
    cm_n += h_ids
    spare2 = [pol("cm2_n", 1, c) for c in range(pos2, w2)]
    lookups_T = posT
    z_ids, nd = [], []
    for i in range(n_prod):
        z_ids.append(pol("cm3_n", 3, 3 * i))
        num, den = pol("tmpExp_n", 3, posT), pol("tmpExp_n", 3, posT + 3)
        posT += 6
        exp2pol[str(next_exp)], exp2pol[str(next_exp + 1)] = num, den
        nd.append((next_exp, next_exp + 1))
        next_exp += 2
    for i in range(l1 + l3):                              # the lookups' products come first (starks.cpp:473-536)
        pu[i]["zId"], pu[i]["numId"], pu[i]["denId"] = z_ids[i], nd[i][0], nd[i][1]
    n_pe = (n_prod - l1 - l3) // 2
    pe = [{"tExpId": 0, "fExpId": 0, "zId": z_ids[l1 + l3 + i], "c1Id": 0, "numId": nd[l1 + l3 + i][0], "denId": nd[l1 + l3 + i][1], "c2Id": 0} for i in range(n_pe)]
    ci = [{"zId": z_ids[k], "numId": nd[k][0], "denId": nd[k][1], "c1Id": 0, "c2Id": 0} for k in range(l1 + l3 + n_pe, n_prod)]
    cm_n += z_ids
    spare3 = [pol("cm3_n", 1, c) for c in range(3 * n_prod, w3)]
    cm_n += spare2 + spare3
    n_base = len(vpm)
    ext_of = {}
    for pid in cm_n:                                      # the committed polynomials' extensions, same position in the _2ns section
        v = vpm[pid]
        ext_of[pid] = pol(v["section"].replace("_n", "_2ns"), v["dim"], v["sectionPos"])
    cm_2ns = [ext_of[p] for p in cm_n]
    qs = [pol("cm4_2ns", 3, 3 * i) for i in range(args.qdeg)]
    pol("q_2ns", 3, 0); pol("f_2ns", 3, 0)
    rng = np.random.default_rng(7)
    ev = [{"type": "q", "id": i, "prime": False} for i in range(args.qdeg)]
    while len(ev) < args.n_evals:
        r = rng.random()
        if r < 0.25:
            ev.append({"type": "const", "id": int(rng.integers(0, args.n_const)), "prime": bool(rng.random() < 0.2)})
        else:
            ev.append({"type": "cm", "id": int(rng.integers(0, len(cm_n))), "prime": bool(rng.random() < 0.25)})
    sec = lambda f: {k: f(k) for k in ORDER}
    si = {"starkStruct": {"nBits": nbits, "nBitsExt": nbits_ext, "nQueries": args.n_queries, "verificationHashType": "GL",
                          "steps": [{"nBits": b} for b in fri_steps(nbits_ext, args.fri_steps)]},
          "mapTotalN": total, "nConstants": args.n_const, "nPublics": 8, "nCm1": w1, "nCm2": len(h_ids), "nCm3": n_prod, "nCm4": args.qdeg, "qDeg": args.qdeg, "qDim": 3,
          "friExpId": 1, "nExps": next_exp,
          "mapDeg": sec(lambda k: NE if k.endswith("2ns") else N), "mapOffsets": sec(lambda k: off[k]),
          "mapSections": sec(lambda k: [i for i, v in enumerate(vpm) if v["section"] == k]), "mapSectionsN": sec(lambda k: cols[k]),
          "mapSectionsN1": sec(lambda k: sum(1 for v in vpm if v["section"] == k and v["dim"] == 1)),
          "mapSectionsN3": sec(lambda k: sum(1 for v in vpm if v["section"] == k and v["dim"] == 3)),
          "varPolMap": vpm, "qs": qs, "cm_n": cm_n, "cm_2ns": cm_2ns, "peCtx": pe, "puCtx": pu, "ciCtx": ci, "evMap": ev, "exp2pol": exp2pol}
    # ---- the five programs, in the reference's table formats, over the map's absolute offsets
    f2, f3p, f3, f42, f52 = args.field_ops
    base1 = [(off["cm1_n"], w1)]
    base2 = base1 + ([(off["cm2_n"], w2, pos2)] if pos2 else [])
    progs, secs = {}, {}
    ll = lambda f: min(70, max(4, f // 40))
    if f2 and lookups_T:
        progs["step2prev"] = cpg.synthetic_program_zkevm_shape(np.random.default_rng(20), N, base1, args.n_const, 8, field_ops=f2, next_shift=1, vc=1,
                                                               long_lived=ll(f2), base_out=(off["tmpExp_n"], args.tmpexp, lookups_T + 4))
    if f3p and n_prod:
        progs["step3prev"] = cpg.synthetic_program_zkevm_shape(np.random.default_rng(30), N, base2, args.n_const, 8, field_ops=f3p, next_shift=1,
                                                               vc=3, long_lived=ll(f3p), base_out=(off["tmpExp_n"] + lookups_T, args.tmpexp, 6 * n_prod + 4))
    if f3 and w3 - 3 * n_prod >= 8:
        progs["step3"] = cpg.synthetic_program_zkevm_shape(np.random.default_rng(31), N, base2 + ([(off["cm3_n"], w3, 3 * n_prod)] if n_prod else []), args.n_const, 8,
                                                           field_ops=f3, next_shift=1, vc=3, long_lived=ll(f3), base_out=(off["cm3_n"] + 3 * n_prod, w3, w3 - 3 * n_prod))
    ext3 = [(off[k], cols[k]) for k in ("cm1_2ns", "cm2_2ns", "cm3_2ns") if cols[k]]
    if len(ext3) == 3 and not args.step42_generic:       # the stand-in with the real zkEVM program's statistics, scaled to the requested size
        fit = dict(cpg.ZKEVM_STEP42NS_FIT, field_ops=f42, next_shift=shift_ext)
        fit["burst"] = [fit["burst"][0], max(4, fit["burst"][1] * f42 // 17986)]
        progs["step42ns"] = cpg.synthetic_program_zkevm_shape(np.random.default_rng(42), NE, ext3, args.n_const, 8, **fit)
    else:
        progs["step42ns"] = cpg.synthetic_program_zkevm_shape(np.random.default_rng(42), NE, ext3, args.n_const, 8, field_ops=f42, next_shift=shift_ext, long_lived=ll(f42))
    ext4 = ext3 + [(off["cm4_2ns"], w4)]
    import mi_stark
    probe = mi_stark.ChelpersProgram(None, *cpg.synthetic_program52(np.random.default_rng(52), ext4, args.n_const, args.n_evals, length=200), step=52)
    per_len = probe.stats["field_ops"] / 200.0
    probe.close()
    progs["step52ns"] = cpg.synthetic_program52(np.random.default_rng(52), ext4, args.n_const, args.n_evals, length=max(20, int(round(f52 / per_len))))
    bsec = [(off[k], cols[k], N) for k in ("cm1_n", "cm2_n", "cm3_n", "tmpExp_n") if cols[k]]
    for k in ("step2prev", "step3prev", "step3"):
        secs[k] = bsec
    secs["step42ns"] = [(o_, w_, NE) for (o_, w_) in ext3]
    secs["step52ns"] = [(o_, w_, NE) for (o_, w_) in ext4]
    return si, progs, secs, off, cols


def tiled_witness(si, n, w1):
    """host/starks.hpp's rule for keeping cm1_n tile-major in the image (one device, device steps): no lookup or grand-product operand
    is a witness column read by stride.  The base-domain programs are compiled for that layout (mi_chelpers_set_tiled_section)."""
    if os.environ.get("MI_STARK_TILED_WITNESS", "1")[:1] == "0" or n < 64 or w1 == 0 or len(os.environ.get("MI_STARK_DEVICES", "").split(",")) > 1:
        return False
    ids = [c[k] for c in si["puCtx"] for k in ("fExpId", "tExpId", "numId", "denId")]
    ids += [c[k] for ctx in ("peCtx", "ciCtx") for c in si[ctx] for k in ("numId", "denId")]
    return all(si["varPolMap"][int(si["exp2pol"][str(e)])]["section"] != "cm1_n" for e in ids)


def tiled_ext(ne, w):
    """host/starks.hpp's rule for keeping a wide extended section (cm1_2ns .. cm3_2ns) tile-major in the image: one device, device steps,
    at least one tile of rows, more than 4 columns.  step42ns / step52ns are compiled for that layout."""
    return os.environ.get("MI_STARK_TILED_EXT", "1")[:1] != "0" and ne >= 64 and w > 4 and not os.environ.get("MI_STARK_DEVICES", "")


def tiled_consts(n):
    """host/starks.hpp's rule for keeping the resident constant polynomials tile-major: one device, at least one tile of rows."""
    return os.environ.get("MI_STARK_TILED_CONSTS", "1")[:1] != "0" and n >= 64 and not os.environ.get("MI_STARK_DEVICES", "")


def compiled_programs(args, shard=None):
    """The five programs through mi_chelpers_compile + the native build with the in-tree code-object cache (no GPU needed): what
    Starks does on first use, done ahead so that the GPU box finds every kernel in the cache."""
    import mi_stark
    si, progs, secs, off, cols = shape(args)
    n, ne = 1 << args.log_n, 1 << (args.log_n + args.ext_bits)
    stats = {}
    for name, (ops, ar) in progs.items():
        base = name in ("step2prev", "step3prev", "step3")
        p = mi_stark.ChelpersProgram(None, ops, ar, sections=secs[name], n_const=args.n_const, nrows_ext=n if base else ne, step=STEP_ID[name])
        if base and tiled_witness(si, n, cols["cm1_n"]):
            p.set_tiled_section(off["cm1_n"])
        if base and args.n_const and tiled_consts(n):
            p.set_tiled_consts()
        if not base:
            for k in ("cm1_2ns", "cm2_2ns", "cm3_2ns"):
                if tiled_ext(ne, cols[k]):
                    p.set_tiled_section(off[k])
        if shard is not None:
            p.precompile_shard(*shard)
        else:
            stats[name] = dict(p.build_native(), field_ops=p.stats["field_ops"])
        p.close()
    return stats


def main():
    argv = sys.argv[1:]
    if "--shape" in argv and argv[argv.index("--shape") + 1] == "batch":
        # genBatchProof's three Starks::genProof calls (prover.cpp:541 zkEVM, :577 c12a, :611 recursive1) one after the other in ONE process,
        # as the prover runs them: they share the process's HBM arena like they share pAddress in the reference
        k = argv.index("--shape")
        rest = argv[:k] + argv[k + 2:]
        outs = [run(parse(["--shape", sh] + rest)) for sh in ("zkevm", "c12a", "recursive1")]
        print(json.dumps({"metric": "the three Starks::genProof calls of genBatchProof (zkEVM, c12a, recursive1 shapes), one process, one MI355X", "unit": "ms",
                          "value": sum(o["value"] for o in outs), "higher_is_better": False, "n_gpus": 1,
                          "starks": {sh: {"genproof_ms": o["value"], "workload": o["config"]["workload"], "peak_hbm_gb": o["hbm"]["peak_hbm_gb"], "checks": o["checks"],
                                          "phase_ms": o["phase_ms"], "genproof_wall_ms": o["genproof_wall_ms"], "proof_to_json_ms": o["proof_to_json_ms"],
                                          "hbm": o["hbm"], "setup_s": o["setup_s"], "field_ops_per_row": o["config"]["field_ops_per_row"], "flow": o["flow"]}
                                     for sh, o in zip(("zkevm", "c12a", "recursive1"), outs)}, "dtype": "u64", "data": "synthetic"}))
        return
    args = parse()
    if args.precompile is not None:
        compiled_programs(args, tuple(args.precompile))
        return
    print(json.dumps(run(args)))


def run(args):
    import torch
    import mi_stark
    import glo                     # the oracle: only in the checks after the timed proofs

    si, progs, secs, off, cols = shape(args)
    nbits, nbits_ext = args.log_n, args.log_n + args.ext_bits
    N, NE = 1 << nbits, 1 << nbits_ext
    w1 = cols["cm1_n"]
    ctx = mi_stark.Context(0)
    free0, total_hbm = ctx.mem_info()
    if args.chelpers_batch_rows:
        ctx.set_chelpers_batch_rows(args.chelpers_batch_rows)
    os.environ.setdefault("MI_CHELPERS_CACHE", mi_stark.default_chelpers_cache())
    workdir = os.environ.get("MI_BENCH_TMP", "/tmp")
    si_path = os.path.join(workdir, "bench_starks.starkinfo.json")
    json.dump(si, open(si_path, "w"))
    # ---- host inputs: the witness (cm1_n, the only part of pAddress genProof reads), the constant polynomials, and the image of the
    # constant-tree file -- never written here: zero pages, of which the query phase reads 128 rows and paths
    t0 = time.perf_counter()
    # (per-row steps: the whole map, as the reference allocates it -- the recorder runs each function once at rows 0 and n - 1 over it)
    witness = torch.zeros(si["mapTotalN"], dtype=torch.int64) if args.per_row_steps else torch.empty(N * w1, dtype=torch.int64, pin_memory=False)
    chunk = 1 << 28
    d = ctx.empty(min(chunk, max(N * w1, N * args.n_const)))
    for o_ in range(0, N * w1, chunk):
        k = min(chunk, N * w1 - o_)
        ctx.fill_synthetic(d, k, 0x5EED0104 + o_ // chunk)
        witness[o_:o_ + k].copy_(d[:k])
    const_n = torch.empty(N * args.n_const, dtype=torch.int64)
    for o_ in range(0, N * args.n_const, chunk):
        k = min(chunk, N * args.n_const - o_)
        ctx.fill_synthetic(d, k, 0x5EED0204 + o_ // chunk)
        const_n[o_:o_ + k].copy_(d[:k])
    torch.cuda.synchronize()
    del d
    torch.cuda.empty_cache()
    tree_elems = 2 + args.n_const * NE + (2 * NE - 1) * 4
    const_tree = np.zeros(tree_elems, dtype=np.uint64)      # calloc: untouched pages cost nothing
    const_tree[0], const_tree[1] = args.n_const, NE
    publics = np.arange(1, 9, dtype=np.uint64)
    t_inputs = time.perf_counter() - t0

    L = ctypes.CDLL(os.path.join(ROOT, "merlin-zkevm-prover_amd", "libmi_starks.so"), mode=ctypes.RTLD_GLOBAL)
    L.mis_create.restype = ctypes.c_void_p
    L.mis_hbm_plan_bytes.restype = ctypes.c_uint64
    L.mis_min_free_bytes.restype = ctypes.c_uint64
    L.mis_phase_times.restype = ctypes.c_uint64
    L.mis_zkin.restype = ctypes.c_char_p
    vp = ctypes.c_void_p
    t0 = time.perf_counter()
    h = vp(L.mis_create(si_path.encode(), vp(const_n.data_ptr()), vp(const_tree.ctypes.data), vp(witness.data_ptr())))
    t_create = time.perf_counter() - t0
    for name, (ops, ar) in progs.items():
        ops, ar = np.ascontiguousarray(ops, dtype=np.uint64), np.ascontiguousarray(ar, dtype=np.uint64)
        L.mis_set_tables(h, ctypes.c_int(STEP_ID[name]), vp(ops.ctypes.data), ctypes.c_uint64(ops.size), vp(ar.ctypes.data), ctypes.c_uint64(ar.size))
    t_steps_lib = 0.0
    if args.per_row_steps:
        import subprocess
        import gen_steps_cpp as gs
        t0 = time.perf_counter()
        host = os.path.join(ROOT, "merlin-zkevm-prover_amd", "host")
        src = os.path.join(workdir, "bench_starks_steps.cpp")
        open(os.path.join(workdir, "genSteps.hpp"), "w").write(gs.GEN_HEADER)
        open(src, "w").write(gs.steps_source("GenSteps", progs, header='#include "genSteps.hpp"\n') + gs.GEN_FACTORY)
        so = os.path.join(workdir, "libbench_starks_steps.so")
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-fopenmp", "-fPIC", "-shared", "-I", workdir, "-I", host, "-I", os.path.join(host, "standalone"), src, "-o", so])
        if L.mis_load_steps(h, so.encode()) != 0:
            raise SystemExit("bench_starks: cannot load the generated Steps library")
        t_steps_lib = time.perf_counter() - t0
    batch = 1 if args.per_row_steps else 4
    plan = L.mis_hbm_plan_bytes(h)
    L.mis_phase_timer(1)
    walls, inner, digests = [], [], []
    for it in range(args.proofs):
        L.mis_phase_timer(1)                                 # (resets the low-water mark)
        t0 = time.perf_counter()
        L.mis_gen_proof(h, vp(publics.ctypes.data), ctypes.c_uint64(batch), b"", b"")
        walls.append(1e3 * (time.perf_counter() - t0))
        two = (ctypes.c_double * 2)()
        L.mis_last_wall_ms(h, two)
        inner.append((two[0], two[1]))
        digests.append(hashlib.sha256(L.mis_zkin(h)).hexdigest())  # (after the clock: the proof's zkin text)
        print("genProof %d: %.1f ms" % (it, walls[-1]), file=sys.stderr, flush=True)
    buf = ctypes.create_string_buffer(1 << 16)
    L.mis_phase_times(buf, ctypes.c_uint64(len(buf)))
    phases = {ln.split()[0]: float(ln.split()[1]) for ln in buf.value.decode().splitlines() if ln.strip()}
    min_free = L.mis_min_free_bytes()
    z = json.loads(L.mis_zkin(h).decode())

    # ------------------------------------------------------------------ checks (oracle), after the clock
    checks = {}
    if len(digests) > 1:  # same witness, same publics: every proof of the run is the same text (nothing of a proof leaks into the next one)
        checks["consecutive_proofs_identical"] = "%d/%d" % (sum(d_ == digests[0] for d_ in digests), len(digests))
    U = lambda x: np.array(x, dtype=object).astype(np.uint64)
    steps = fri_steps(nbits_ext, args.fri_steps)
    tr = glo.Transcript()
    tr.put(publics)
    chal = np.zeros(8 * 3, dtype=np.uint64)
    C = lambda k: slice(3 * k, 3 * k + 3)
    tr.put(U(z["root1"])); chal[C(0)] = tr.get_field(); chal[C(1)] = tr.get_field()
    tr.put(U(z["root2"])); chal[C(2)] = tr.get_field(); chal[C(3)] = tr.get_field()
    tr.put(U(z["root3"])); chal[C(4)] = tr.get_field()
    tr.put(U(z["root4"])); chal[C(7)] = tr.get_field()
    evals = U(z["evals"]).reshape(-1)
    tr.put(evals)
    chal[C(5)] = tr.get_field(); chal[C(6)] = tr.get_field()
    fri_chal = []
    for si_ in range(len(steps)):
        fri_chal.append(tr.get_field())
        tr.put(U(z["s%d_root" % (si_ + 1)]) if si_ < len(steps) - 1 else U(z["finalPol"]).reshape(-1))
    ys = [int(v) for v in tr.get_permutations(args.n_queries, steps[0])]
    ok_paths, n_paths = 0, 0
    for q in range(min(args.n_queries, 16)):
        for t_, root in (("1", "root1"), ("2", "root2"), ("3", "root3"), ("4", "root4")):
            if "s0_vals" + t_ not in z:                      # a stage without columns is not opened (proof2zkinStark.cpp:30-77)
                continue
            ok_paths += bool(glo.merkle_verify(U(z[root]), U(z["s0_vals" + t_][q]).reshape(-1), U(z["s0_siblings" + t_][q]).reshape(-1), ys[q]))
            n_paths += 1
    checks["merkle_paths_ok"] = "%d/%d" % (ok_paths, n_paths)
    ok_fold, n_fold = 0, 0
    for q in range(min(args.n_queries, 16)):
        g = ys[q]
        for s_ in range(1, len(steps)):
            prev, cur = steps[s_ - 1], steps[s_]
            vals, sib = U(z["s%d_vals" % s_][q]).reshape(-1), U(z["s%d_siblings" % s_][q]).reshape(-1)
            gi = g % (1 << cur)
            ok_paths_fri = glo.merkle_verify(U(z["s%d_root" % s_]), vals, sib, gi)
            folded = glo.fri_fold_group(vals, prev - cur, prev, nbits_ext, gi, fri_chal[s_])
            if s_ + 1 < len(steps):
                nxt = U(z["s%d_vals" % (s_ + 1)][q]).reshape(-1)
                j = gi >> steps[s_ + 1]
                target = nxt[3 * j:3 * j + 3]
            else:
                target = U(z["finalPol"]).reshape(-1)[3 * gi:3 * gi + 3]
            ok_fold += bool(ok_paths_fri and np.array_equal(folded, target))
            n_fold += 1
            g = gi
    checks["fri_folds_ok"] = "%d/%d" % (ok_fold, n_fold)
    # sampled rows of q_2ns / f_2ns against the oracle interpreters, operands read back from the device image
    def peek(o_, n_):
        out = np.empty(n_, dtype=np.uint64)
        assert L.mis_peek(h, ctypes.c_uint64(o_), ctypes.c_uint64(n_), vp(out.ctypes.data)) == 0
        return out
    rows = ([0, 1, NE - 2, NE - 1] + [int(v) for v in np.random.default_rng(5).integers(0, NE, size=max(0, args.check_rows - 4))])[:args.check_rows]
    zh = ctx.zhinv(nbits, nbits_ext)
    late = np.zeros(3, dtype=np.uint64)
    L.mis_late_offsets(h, vp(late.ctypes.data))              # where the stage-4 re-plan put const_2ns, xDivXSubXi, xDivXSubWXi
    const_off, xd_off, xdw_off = (int(v) for v in late)
    ok_q = ok_f = 0
    for r in rows:
        got = {}
        for rr in sorted({r, (r + (1 << args.ext_bits)) % NE}):
            got[rr] = {k: peek(off[k] + rr * cols[k], cols[k]) if cols[k] else np.zeros(0, dtype=np.uint64) for k in ("cm1_2ns", "cm2_2ns", "cm3_2ns", "cm4_2ns")}
            got[rr]["const"] = peek(const_off + rr * args.n_const, args.n_const)
        ok_q += int(_check_row42(glo, progs["step42ns"], got, r, NE, off, cols, args.n_const, chal, publics, zh, nbits_ext, peek(off["q_2ns"] + 3 * r, 3)))
        ok_f += int(_check_row52(glo, progs["step52ns"], got, r, NE, off, cols, args.n_const, chal, evals, peek(xd_off + 3 * r, 3), peek(xdw_off + 3 * r, 3),
                                 peek(off["f_2ns"] + 3 * r, 3)))
    checks["q_2ns_rows_equal_oracle"] = "%d/%d" % (ok_q, len(rows))
    checks["f_2ns_rows_equal_oracle"] = "%d/%d" % (ok_f, len(rows))

    total = phases.get("STARK_STEP_1", 0) + phases.get("STARK_STEP_2", 0) + phases.get("STARK_STEP_3", 0) + phases.get("STARK_STEP_4", 0) + \
        phases.get("STARK_STEP_5", 0) + phases.get("STARK_STEP_FRI", 0) + phases.get("STARK_INITIALIZATION", 0)
    out = {"metric": "Starks::genProof wall time, synthetic zkEVM-shaped STARK (BASELINE config 4 substitute)", "unit": "ms", "value": inner[-1][0],
           "higher_is_better": False, "n_gpus": 1, "flow": "host/starks.hpp class Starks through libmi_starks.so, " + ("generated per-row Steps code recorded and run on the device (nrowsStepBatch 1)" if args.per_row_steps else "device steps (nrowsStepBatch 4)"),
           "config": {"workload": "2^%d rows, sections %s / tmpExp %d / %d constants, %d + %d lookups, %d grand products, %d evaluations, %d queries, FRI %s"
                      % (nbits, args.widths, args.tmpexp, args.n_const, args.n_lookups[0], args.n_lookups[1], args.n_products, args.n_evals, args.n_queries, steps),
                      "field_ops_per_row": dict(zip(("step2prev", "step3prev", "step3", "step42ns", "step52ns"), args.field_ops))},
           "genproof_wall_ms": [a for a, _ in inner], "proof_to_json_ms": [b for _, b in inner], "call_wall_ms_incl_json": walls, "phase_ms_sum": total, "phase_ms": phases,
           "hbm": {"total_gb": total_hbm / 1e9, "free_before_gb": free0 / 1e9, "plan_gb": plan / 1e9, "peak_hbm_gb": (total_hbm - min_free) / 1e9,
                   "fits_one_gpu": bool(total_hbm - min_free < total_hbm)},
           "setup_s": {"inputs": t_inputs, "starks_ctor_incl_const_upload": t_create, "per_row_steps_library_build": t_steps_lib},
           "checks": checks, "zkin_sha256": digests[0] if digests else None,   # the proof's text: a run on one device and a sharded one must agree on it
           "dtype": "u64", "data": "synthetic"}
    # several devices (MI_STARK_DEVICES): what the driver answered about direct access between them -- a pair without it stages its
    # exchange through the host, and a reader of this line should not have to guess that from the times
    mat = (ctypes.c_int * 256)()
    warn = ctypes.create_string_buffer(2048)
    bad = ctypes.c_int(0)
    G = int(L.mis_peer_access(mat, warn, ctypes.c_uint64(2048), ctypes.byref(bad)))
    L.mis_image_backed_bytes.restype = ctypes.c_uint64
    sp = ctypes.c_int(0)
    out["hbm"]["image_backed_gb"] = int(L.mis_image_backed_bytes(ctypes.byref(sp))) / 1e9
    out["hbm"]["image_is_a_sparse_address_range"] = bool(sp.value)
    chk = (ctypes.c_uint64 * 4)()
    ctypes.CDLL(os.path.join(ROOT, "merlin-zkevm-prover_amd", "libmi_stark.so")).mi_multi_check_stats(chk)
    if chk[0]:
        out["multi_check"] = {"checks": int(chk[1]), "unknown_pointers": int(chk[2]), "violations": int(chk[3])}
    if G:
        out["row_sharded"] = os.environ.get("MI_STARK_ROW_SHARDED")
        out["device_groups"] = os.environ.get("MI_MULTI_GROUP_SAME_DEVICE", "0") == "1"
        out["peer_access"] = {"devices": os.environ.get("MI_STARK_DEVICES"), "matrix": [[int(mat[a * G + b]) for b in range(G)] for a in range(G)],
                              "indirect_pairs": int(bad.value), "warning": warn.value.decode(), "legend": "2 same device, 1 peer access enabled, 0 not possible, -1 enabling failed"}
        out["n_gpus"] = len(set(os.environ.get("MI_STARK_DEVICES", "").split(",")))
    L.mis_destroy(h)
    return out


def _check_row42(glo, prog, got, r, NE, off, cols, n_const, chal, publics, zh, nbits_ext, want):
    ops, ar = prog
    # run the oracle over a sparse copy of the area: only the two rows are populated (the arrays are calloc'ed: untouched pages are free)
    top = off["cm4_2ns"] + NE * cols["cm4_2ns"]
    pols = np.zeros(top, dtype=np.uint64)
    cp = np.zeros(NE * n_const, dtype=np.uint64)
    for rr, g_ in got.items():
        for k in ("cm1_2ns", "cm2_2ns", "cm3_2ns", "cm4_2ns"):
            pols[off[k] + rr * cols[k]:off[k] + (rr + 1) * cols[k]] = g_[k]
        cp[rr * n_const:(rr + 1) * n_const] = g_["const"]
    x = np.zeros(NE, dtype=np.uint64)
    L = glo.lib()
    x[r] = L.glo_mul(49, L.glo_pow(L.glo_w(nbits_ext), r))
    q = np.zeros(NE * 3, dtype=np.uint64)
    glo.chelpers_step42ns(ops, ar, pols, cp, n_const, chal, publics, x, 1, zh, q, r, 1)
    return np.array_equal(q[3 * r:3 * r + 3], want)


def _check_row52(glo, prog, got, r, NE, off, cols, n_const, chal, evals, xd, xdw, want):
    ops, ar = prog
    top = off["cm4_2ns"] + NE * cols["cm4_2ns"]
    pols = np.zeros(top, dtype=np.uint64)
    cp = np.zeros(NE * n_const, dtype=np.uint64)
    g_ = got[r]
    for k in ("cm1_2ns", "cm2_2ns", "cm3_2ns", "cm4_2ns"):
        pols[off[k] + r * cols[k]:off[k] + (r + 1) * cols[k]] = g_[k]
    cp[r * n_const:(r + 1) * n_const] = g_["const"]
    xdv, xdwv, f = np.zeros(NE * 3, dtype=np.uint64), np.zeros(NE * 3, dtype=np.uint64), np.zeros(NE * 3, dtype=np.uint64)
    xdv[3 * r:3 * r + 3], xdwv[3 * r:3 * r + 3] = xd, xdw
    glo.chelpers_step52ns(ops, ar, pols, cp, n_const, chal, evals, xdv, xdwv, f, r, 1)
    return np.array_equal(f[3 * r:3 * r + 3], want)


if __name__ == "__main__":
    main()
