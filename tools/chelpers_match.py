#!/usr/bin/env python3
"""Statistics of a step42ns constraint program as the translator and the native backend see it, and a search for generator parameters
under which the SYNTHETIC zkEVM-shaped program (tests/chelpers_programs.synthetic_program_zkevm_shape -- the reference's generated tables
are reference source and do not travel to the GPU box) has the statistics of the REAL one.

    tools/chelpers_match.py stats            the real program's statistics (needs /root/reference) -> profiles/r03_chelpers_step42ns_target.json
    tools/chelpers_match.py fit [--iters N] [--from-fit]   coordinate search over the generator's parameters against that file
    tools/chelpers_match.py show             the committed fit (tests/chelpers_programs.ZKEVM_STEP42NS_FIT) next to the target

Only NUMBERS about the reference's program are written (counts, fractions): no table text.  No GPU needed."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "merlin-zkevm-prover_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
REF_HPP = "/root/reference/src/starkpil/zkevm/chelpers/zkevm.chelpers.step42ns.parser.hpp"
TARGET = os.path.join(ROOT, "profiles", "r03_chelpers_step42ns_target.json")
N = 1 << 23
NE = 2 * N
SECS = [(1435 * N, 665), (2765 * N, 128), (3021 * N, 371)]
N_CONST = 218                                   # ConstantPols::numPols() (pols_generated/constant_pols.hpp:689)
# the statistics the fit is judged on (5 % each): what decides the generated kernels' cost and memory behaviour
KEYS = ["field_ops", "live_words_rescheduled", "kernels", "estimated_valu_per_row", "operand_loads_per_row", "distinct_operands",
        "spill_words_moved_per_row", "horner_chain_steps", "frac_reads_cm1", "frac_reads_cm2", "frac_reads_cm3", "frac_reads_const", "frac_reads_prime"]


def program_stats(ops, args):
    import chelpers_programs as cp
    import mi_stark
    prog = mi_stark.ChelpersProgram(None, ops, args, sections=[(o, w, NE) for (o, w) in SECS], n_const=N_CONST, nrows_ext=NE)
    st = dict(prog.stats)
    st.update(prog.lower_stats())
    prog.close()
    micro, _ = cp.decode(ops, args)
    cnt = {"cm1": 0, "cm2": 0, "cm3": 0, "const": 0, "prime": 0, "all": 0}
    for (_, _, _, srcs) in micro:
        for k, a in srcs:
            if k in (cp.POL, cp.POLS, cp.POL3, cp.POL3S):
                off = a[0]
                sec = "cm1" if off < SECS[1][0] else "cm2" if off < SECS[2][0] else "cm3"
                cnt[sec] += 1
                cnt["all"] += 1
                cnt["prime"] += k in (cp.POLS, cp.POL3S)
            elif k in (cp.CONST, cp.CONSTS):
                cnt["const"] += 1
                cnt["all"] += 1
                cnt["prime"] += k == cp.CONSTS
    for k in ("cm1", "cm2", "cm3", "const", "prime"):
        st["frac_reads_" + k] = cnt[k] / max(cnt["all"], 1)
    st["field_ops"] = st["after_copy_forwarding"]
    return st


def synthetic(params):
    import chelpers_programs as cp
    return cp.synthetic_program_zkevm_shape(np.random.default_rng(42), NE, SECS, N_CONST, 8, **params)


def distance(st, target):
    return {k: (st[k] - target[k]) / target[k] if target[k] else st[k] for k in KEYS}


def main():
    cmd = sys.argv[1] if len(sys.argv) > 1 else "stats"
    import chelpers_programs as cp
    if cmd == "stats":
        ops, args = cp.parse_reference_tables(open(REF_HPP).read())
        st = program_stats(ops, args)
        out = {"_what": "statistics of the reference's zkEVM step42ns program (zkevm.chelpers.step42ns.parser.hpp: counts and fractions only) as "
                        "tools/chelpers_match.py measures them; the synthetic stand-in that runs on the GPU box is fitted to these",
               "stats": {k: st[k] for k in sorted(st)}}
        json.dump(out, open(TARGET, "w"), indent=1)
        print(json.dumps({k: st[k] for k in KEYS}, indent=1))
        return
    target = json.load(open(TARGET))["stats"]
    if cmd == "show":
        st = program_stats(*synthetic(cp.ZKEVM_STEP42NS_FIT))
        d = distance(st, target)
        for k in KEYS:
            print(f"{k:28s} real {target[k]:12.4f}  synthetic {st[k]:12.4f}  {100 * d[k]:+6.1f} %")
        return
    if cmd == "fit":
        import time
        iters = int(sys.argv[sys.argv.index("--iters") + 1]) if "--iters" in sys.argv else 6
        P = dict(field_ops=17986, long_lived=78, sec_weights=[0.93, 0.012, 0.055], kind_weights={cp.CONST: 0.35, cp.CONSTS: 0.12, cp.POLS: 0.12, cp.POL: 1.0},
                 mean_len=5.4, ext_frac=0.09, run_ops=750, pool_scale=1.3, zipf=0.7, ll_generations=3, ll_use=0.5)
        if "--from-fit" in sys.argv:                       # continue from the committed parameters
            P = json.loads(json.dumps({k: v for k, v in cp.ZKEVM_STEP42NS_FIT.items() if k != "kind_weights"}))
            P["kind_weights"] = dict(cp.ZKEVM_STEP42NS_FIT["kind_weights"])

        def score(par):
            st = program_stats(*synthetic(par))
            d = distance(st, target)
            return sum(min(abs(v), 3.0) ** 2 for v in d.values()), d
        best, bd = score(P)
        print("start %.4f" % best, {k: round(100 * v, 1) for k, v in bd.items()}, flush=True)
        knobs = [("mean_len", [0.85, 1.15]), ("pool_scale", [0.8, 1.25]), ("long_lived", [0.9, 1.1]), ("ll_generations", [-1, +1]), ("ll_use", [0.8, 1.25]),
                 ("ext_frac", [0.8, 1.25]), ("run_ops", [0.75, 1.33]), ("zipf", [0.85, 1.15]), ("sec0", [0.97, 1.03]), ("sec1", [0.7, 1.4]), ("sec2", [0.8, 1.25]),
                 ("kCONST", [0.8, 1.25]), ("kCONSTS", [0.6, 1.6]), ("kPOLS", [0.75, 1.33]), ("kNUM", [0.8, 1.25]), ("kT1", [0.85, 1.18]), ("shared_scale", [0.6, 1.5]),
                 ("burst1", [0.9, 1.1]), ("neighbour", [0.7, 1.4]), ("pol3_frac", [0.7, 1.4]), ("pols_global", [0.85, 1.15])]
        for it in range(iters):
            improved = False
            for name, moves in knobs:
                for mv in moves:
                    Q = json.loads(json.dumps({k: v for k, v in P.items() if k != "kind_weights"}))
                    Q["kind_weights"] = dict(P["kind_weights"])
                    if name.startswith("sec"):
                        i = int(name[3])
                        Q["sec_weights"][i] *= mv
                    elif name.startswith("k"):
                        kk = getattr(cp, name[1:])
                        Q["kind_weights"][kk] = Q["kind_weights"].get(kk, 1.0) * mv
                    elif name == "burst1":
                        Q["burst"] = [Q["burst"][0], max(1, int(round(Q["burst"][1] * mv)))]
                    elif name in ("ll_generations",):
                        Q[name] = max(1, Q[name] + mv)
                    elif name in ("long_lived", "run_ops"):
                        Q[name] = max(1, int(round(Q[name] * mv)))
                    elif name in ("pol3_frac", "pols_global"):
                        Q[name] = min(1.0, max(0.05, Q.get(name, 0.5) * mv))
                    else:
                        Q[name] = Q[name] * mv
                    sc, d = score(Q)
                    if sc < best - 1e-6:
                        best, bd, P, improved = sc, d, Q, True
                        print("it %d %-14s x%-5s -> %.4f  worst %s" % (it, name, mv, best, max(bd.items(), key=lambda kv: abs(kv[1]))), flush=True)
            if not improved:
                break
        print("FIT =", {k: (v if k != "kind_weights" else {int(a): b for a, b in v.items()}) for k, v in P.items()})
        print({k: round(100 * v, 1) for k, v in bd.items()})
        return
    raise SystemExit("unknown command")


if __name__ == "__main__":
    main()
