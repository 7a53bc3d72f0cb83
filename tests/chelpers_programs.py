"""Test-side knowledge of the step42ns program format (TEST INFRASTRUCTURE): what every opcode of the reference interpreter
(zkevm.chelpers.step42ns.parser.cpp, step42ns_parser_first_avx :24-748) takes as arguments, used to
  * generate synthetic programs that exercise every opcode (the reference's own tables cannot travel to the GPU box),
  * enumerate the memory a program touches (to run the reference's real tables on the CPU over a sparse address space),
  * cross-check the argument counts against the reference's source text where it is present.
A third, independent transcription of the opcode table (the product's is csrc/chelpers.hip, the oracle's
oracle/chelpers_oracle.c): a slip in one of the three shows up as a disagreement."""
import re

import numpy as np

T1, T3, POL, POLS, NUM, CONST, CONSTS, CHAL, PUB, POL3, POL3S, X = range(12)
NARGS = {T1: 1, T3: 1, POL: 2, POLS: 4, NUM: 1, CONST: 1, CONSTS: 3, CHAL: 1, PUB: 1, POL3: 2, POL3S: 4, X: 0}
# opcode -> (dst kind, source a, source b) in argument order; argument 0 is the destination temp
OPS = {
    0: (T1, T1, T1), 1: (T1, T1, POL), 2: (T1, T1, NUM), 3: (T1, T1, CONST), 4: (T1, POL, POL), 5: (T1, POLS, POLS), 6: (T1, POL, CONST),
    7: (T1, POL, NUM), 8: (T1, CONST, CONST), 9: (T1, CONSTS, CONSTS), 10: (T1, CONST, NUM), 11: (T1, CONSTS, NUM), 12: (T3, T1, T3),
    13: (T3, NUM, CHAL), 14: (T3, T1, CHAL), 15: (T3, POL, T3), 16: (T3, POL, CHAL), 17: (T3, T3, T3), 18: (T3, T3, CHAL), 19: (T3, POL3, T3),
    20: (T3, POL3, CHAL), 21: (T1, T1, T1), 22: (T1, T1, POL), 23: (T1, T1, POLS), 24: (T1, POL, T1), 25: (T1, POLS, T1), 26: (T1, T1, NUM),
    27: (T1, NUM, T1), 28: (T1, POL, NUM), 29: (T1, POLS, NUM), 30: (T1, NUM, POL), 31: (T1, NUM, POLS), 32: (T1, NUM, CONST),
    33: (T1, NUM, CONSTS), 34: (T1, POL, PUB), 35: (T1, POLS, POL), 36: (T1, POL, POLS), 37: (T1, POL, POL), 38: (T1, POLS, POLS),
    39: (T1, CONST, POL), 40: (T1, T1, CONST), 41: (T3, POL3, NUM), 42: (T3, T3, T3), 43: (T3, T3, CHAL), 44: (T3, T3, POL3), 45: (T1, T1, T1),
    46: (T1, NUM, T1), 47: (T1, POL, T1), 48: (T1, POLS, T1), 49: (T1, T1, CONST), 50: (T1, POL, POL), 51: (T1, POL, POLS), 52: (T1, POLS, POLS),
    53: (T1, NUM, POL), 54: (T1, POL, CONST), 55: (T1, POLS, CONST), 56: (T1, T1, POL), 57: (T1, T1, POLS), 58: (T1, CONST, T1),
    59: (T3, T1, CHAL), 60: (T3, CONST, T3), 61: (T3, T1, T3), 62: (T3, POL, CHAL), 63: (T3, POLS, CHAL), 64: (T3, POL, T3), 65: (T3, POLS, T3),
    66: (T3, NUM, CHAL), 67: (T3, X, CHAL), 68: (T3, X, T3), 69: (None, T3, None), 70: (T3, CHAL, T3), 71: (T3, T3, T3), 72: (T3, POL3, POL3),
    73: (T3, POL3S, CHAL), 74: (T3, POL3S, T3), 75: (T3, POL3, T3), 76: (T3, POL3, CHAL), 77: (T3, POL3S, POL3), 78: (T1, T1, None),
    79: (T1, POL, None), 80: (T1, POLS, None), 81: (T1, NUM, None), 82: (T1, CONST, None), 83: (T1, CONSTS, None),
}
FUSED = {84: [12, 70], 85: [0, 50], 86: [32, 47, 21, 32, 48], 87: [12, 70] * 4, 88: [21, 50, 21, 53, 0, 0, 50, 50, 0, 50, 21, 50]}


def nargs_of(op):
    if op in FUSED:
        return sum(nargs_of(o) for o in FUSED[op])
    d, a, b = OPS[op]
    return (0 if d is None else 1) + NARGS[a] + (NARGS[b] if b is not None else 0)


def decode(ops, args):
    """-> [(opcode, dst kind, dst slot, [(kind, [args])])] micro-operations, and the number of arguments consumed."""
    out, ia = [], 0
    for op in ops:
        for o in FUSED.get(int(op), [int(op)]):
            d, a, b = OPS[o]
            slot = None
            if d is not None:
                slot = int(args[ia]); ia += 1
            srcs = []
            for k in (a, b):
                if k is None:
                    continue
                srcs.append((k, [int(v) for v in args[ia:ia + NARGS[k]]]))
                ia += NARGS[k]
            out.append((o, d, slot, srcs))
    return out, ia


def touched_addresses(micro, rows, numpols):
    """Element indices a program reads in `pols` and in the constant polynomials for the given rows."""
    pols, cpols = set(), set()
    for (_, _, _, srcs) in micro:
        for k, a in srcs:
            for r in rows:
                if k in (POL, POL3):
                    base = a[0] + r * a[1]
                elif k in (POLS, POL3S):
                    base = a[0] + ((r + a[1]) % a[2]) * a[3]
                elif k == CONST:
                    cpols.add(a[0] + r * numpols); continue
                elif k == CONSTS:
                    cpols.add(a[0] + ((r + a[1]) % a[2]) * numpols); continue
                else:
                    continue
                pols.update(range(base, base + (3 if k in (POL3, POL3S) else 1)))
    return pols, cpols


def parse_reference_tables(hpp_text, op_name="op42", args_name="args42"):
    def arr(name):
        m = re.search(name + r"\[[A-Za-z_0-9]*\]\s*=\s*\{([^}]*)\}", hpp_text)
        return np.array([int(x.strip().rstrip("UL")) for x in m.group(1).replace("\n", " ").split(",") if x.strip()], dtype=np.uint64)
    return arr(op_name), arr(args_name)


def synthetic_program(rng, nrows, sections, n_const, n_chal, n_pub, passes=3, nt1=14, nt3=6):
    """A random valid program that uses EVERY opcode 0..88 at least once per pass and whose every result reaches the
    q store (each result is folded into an accumulator at once, so nothing is dead code for the translator).
    sections: [(element offset, row stride)] of `pols`; a section's columns are [0, stride)."""
    ops, args = [], []
    P = 0xFFFFFFFF00000001
    ACC = nt3                                        # accumulator ext temp, beyond the randomly used slots
    def1, def3 = set(), set()

    def gen_src(kind):
        if kind == T1: return [int(rng.choice(sorted(def1)))]
        if kind == T3: return [int(rng.choice(sorted(def3 | {ACC})))]
        if kind == NUM: return [int(rng.integers(0, 1 << 64, dtype=np.uint64)) if rng.random() < 0.5 else int(rng.integers(0, 5))]
        if kind == CONST: return [int(rng.integers(0, n_const))]
        if kind == CONSTS: return [int(rng.integers(0, n_const)), int(rng.integers(1, 4)), nrows]
        if kind == CHAL: return [int(rng.integers(0, n_chal))]
        if kind == PUB: return [int(rng.integers(0, n_pub))]
        off, stride = sections[int(rng.integers(0, len(sections)))]
        col = int(rng.integers(0, stride - (2 if kind in (POL3, POL3S) else 0)))
        if kind in (POL, POL3): return [off + col, stride]
        if kind in (POLS, POL3S): return [off + col, int(rng.integers(1, 4)), nrows, stride]
        return []

    def emit(o):
        d, a, b = OPS[o]
        if (a == T1 or b == T1) and not def1: return False
        if o == 69: return False
        dst = int(rng.integers(0, nt1 if d == T1 else nt3))
        if o == 70:                                  # (dst, challenge, ext temp)
            ar = [dst] + gen_src(CHAL) + gen_src(T3)
        else:
            ar = [dst] + gen_src(a) + (gen_src(b) if b is not None else [])
        return o, ar, d, dst

    def push(o, ar):
        ops.append(o); args.extend(ar)

    # the accumulator starts as a challenge-derived value
    push(13, [ACC, 1, 0])
    order = list(range(84))
    for _ in range(passes):
        rng.shuffle(order)
        pending = list(order)
        stall = 0
        while pending and stall < 400:
            o = pending.pop(0)
            e = emit(o)
            if not e:
                if o != 69: pending.append(o)
                stall += 1
                continue
            o, ar, d, dst = e
            push(o, ar)
            (def1 if d == T1 else def3).add(dst)
            if d == T1: push(12, [ACC, dst, ACC])     # ACC = T1 + ACC
            else: push(17, [ACC, dst, ACC])           # ACC = T3 + ACC
            if rng.random() < 0.3: push(70, [ACC, int(rng.integers(0, n_chal)), ACC])
        # the fused opcodes: their argument lists are the concatenation of their parts'
        for f, parts in FUSED.items():
            ar_all = []
            ok = True
            for part in parts:
                e = emit(part)
                if not e: ok = False; break
                _, ar, d, dst = e
                ar_all += ar
                (def1 if d == T1 else def3).add(dst)
            if ok:
                push(f, ar_all)
                push(12, [ACC, int(rng.choice(sorted(def1))), ACC])
    push(69, [ACC])
    return np.array(ops, dtype=np.uint64), np.array([a % (1 << 64) for a in args], dtype=np.uint64)


def synthetic_program_zkevm_shape(rng, nrows, sections, n_const, n_pub, field_ops=17986, next_shift=2, vc=4, long_lived=70, base_out=None,
                                  sec_weights=None, kind_weights=None, mean_len=4.6, ext_frac=0.09, run_ops=750, pool_scale=1.0, zipf=0.7, ll_generations=1,
                                  ll_use=0.5, burst=None, shared_scale=1.0, partition=False, neighbour=0.0, class_p=(0.46, 0.29, 0.25), pol3_frac=0.0,
                                  pol3_secs=(0.0, 0.23, 0.77), pols_global=0.0):
    """A random valid step42ns program in the SHAPE of the zkEVM one (the real tables cannot travel to the GPU box): ~2 200
    constraint values, each a short base-field expression (on average 5.6 multiplications / additions / subtractions) over
    polynomial elements, shifted ("prime") elements, constants and numbers, every one folded into the running extension
    accumulator as acc = (acc + value) * challenge[vc] (opcode 84, the pattern behind 2 185 of the real program's 2 402 extension
    multiplications); about 9 % of the constraints are extension-valued (value * challenge, an extension product, an extension
    sum or difference, folded with an extension addition).  Operand columns follow a skewed distribution (the real program reads
    1 766 distinct elements 12 525 times).  `long_lived` base values are computed up front and read throughout, so that the
    rescheduled program keeps about as many words live as the real one (86).  Uses challenges 0..vc; ends with the q store.
    sections: [(offset, stride[, readable columns])].  base_out = (offset, stride[, columns]): the same shape in the BASE-DOMAIN steps' numbering (opcodes 0-83 are shared): about every other
    constraint value is stored into a column of that section instead of being accumulated (opcode 100; the zkEVM step3 stores 430 elements) and the accumulator goes to
    three of its columns (opcode 90) instead of q.
    The remaining parameters shape the statistics tools/chelpers_match.py fits to the real program (defaults = the r02 generator, bit for
    bit): sec_weights = probability of each section as the home of a polynomial operand; kind_weights = {operand kind: weight} biasing the
    choice among opcodes of a class by their operand kinds (constants, shifted reads ...); mean_len = mean operations per constraint;
    ext_frac = share of extension-valued constraints; run_ops = field operations per run (a run draws its operands from its own pool);
    pool_scale = size of a run's pool; zipf = skew of the picks inside a pool; ll_generations = the long-lived values are defined in that
    many generations spread over the program instead of all up front (a value then crosses fewer kernel boundaries; 0 = every run
    redefines them from its own operands: values shared inside a state machine's constraints and dead at its end); ll_use = how often a
    two-temporary operation takes a long-lived value as its second operand; burst = (position in [0, 1), values): one run defines that many
    extra shared values and combines them heavily (the real program's peak of live temporaries sits in one state machine's constraints,
    while only a handful of values cross a kernel boundary); partition = the columns of every section (and the constants) are dealt to the
    runs as contiguous slices -- a state machine owns a range of columns -- so that every column is read somewhere and an operand is read
    by one or two kernels, as in the real program (2 167 distinct operands, each loaded by 2.0 kernels on average); neighbour = share of
    a partitioned run's picks that go to the PREVIOUS run's slices instead (constraints that tie two state machines together); class_p =
    shares of multiplications / additions / subtractions among the base-field operations; pol3_frac = share of the extension-valued
    constraints that read EXTENSION-VALUED polynomials (three adjacent columns: the lookups' h1 / h2 of dimension 3 in cm2, the grand products
    z and z' in cm3 -- opcodes 74, 75 / 44 / 41, 72: the real program has 316 such reads of 200 distinct polynomials, 600 staged words),
    pol3_secs = which section they live in."""
    ops, args = [], []
    ACC = 0
    cls_of = lambda o: "add" if o <= 20 else "sub" if o <= 44 else "mul" if o <= 77 else "copy"
    base_ops = [o for o, (d, a, b) in OPS.items() if d == T1 and o < 78 and b is not None]
    first = {c: [o for o in base_ops if cls_of(o) == c and T1 not in OPS[o][1:]] for c in ("add", "sub", "mul")}
    later = {c: [o for o in base_ops if cls_of(o) == c and T1 in OPS[o][1:]] for c in ("add", "sub", "mul")}
    # operand locality like the real program's (2 167 distinct polynomial elements, 12 525 reads, 4 327 loads left after common-
    # subexpression elimination within each of 28 kernels): the constraints come in runs -- one per state machine -- and a run
    # draws its operands from its own pool of elements, plus a few that the whole program shares
    n_runs = max(1, field_ops // run_ops)
    per_run = max(6, int(min(130, field_ops // 30) * pool_scale))
    cur_pool = {"pol": [], "pols": [], "const": []}

    def zipf_pick(pool):
        w = 1.0 / (np.arange(len(pool)) + 4.0) ** zipf
        return pool[int(rng.choice(len(pool), p=w / w.sum()))]

    def rand_col(three=False):
        sec = sections[int(rng.integers(0, len(sections))) if sec_weights is None else int(rng.choice(len(sections), p=np.asarray(sec_weights) / np.sum(sec_weights)))]  # (offset, stride[, readable columns])
        off, stride = sec[0], sec[1]
        return off + int(rng.integers(0, (sec[2] if len(sec) > 2 else stride) - (2 if three else 0))), stride
    n_sh = lambda k: max(0, int(round(k * shared_scale)))    # operands the whole program shares (shared_scale: fit parameter)
    shared = {"pol": [rand_col() for _ in range(n_sh(12))], "pols": [rand_col() for _ in range(n_sh(4))], "const": [int(rng.integers(0, n_const)) for _ in range(n_sh(4))]}

    run_idx = [0]
    shifted_order = {}

    def slice_of(total, r):
        lo, hi = r * total // n_runs, (r + 1) * total // n_runs
        return range(lo, max(hi, lo + 1)) if total else range(0)

    def new_run():
        if partition:
            r = run_idx[0] % n_runs
            run_idx[0] += 1
            per_sec = []
            for sec in sections:
                cols = sec[2] if len(sec) > 2 else sec[1]
                per_sec.append([(sec[0] + c % cols, sec[1]) for c in slice_of(cols, r)])
            cur_pool["sec_prev"] = cur_pool.get("sec", per_sec)
            cur_pool["sec"] = per_sec
            cur_pool["pol"] = [e for lst in per_sec for e in lst] + shared["pol"]
            cur_pool["pols"] = cur_pool["pol"]
            cur_pool["const"] = [c % n_const for c in slice_of(n_const, r)] + shared["const"]
            return
        cur_pool["pol"] = [rand_col() for _ in range(max(3, per_run * 6 // 10))] + shared["pol"]
        cur_pool["pols"] = [rand_col() for _ in range(max(2, per_run * 2 // 10))] + shared["pols"]
        cur_pool["const"] = [int(rng.integers(0, n_const)) for _ in range(max(2, per_run * 2 // 10))] + shared["const"]

    def pick_part():
        """a polynomial operand of the current run: section by sec_weights, column by a skewed pick inside the run's slice of it"""
        w = np.asarray(sec_weights if sec_weights is not None else [1.0] * len(sections), dtype=np.float64)
        w = np.array([wi if cur_pool["sec"][i] else 0.0 for i, wi in enumerate(w)])
        which = "sec_prev" if (neighbour and rng.random() < neighbour) else "sec"
        lst = cur_pool[which][int(rng.choice(len(sections), p=w / w.sum()))]
        return zipf_pick(lst)
    new_run()
    pol_pool, pols_pool, const_pool = cur_pool["pol"], cur_pool["pols"], cur_pool["const"]

    def gen_src(kind, temps):
        if kind == T1: return [temps.pop()]
        if kind == NUM: return [int(rng.integers(0, 1 << 64, dtype=np.uint64)) if rng.random() < 0.3 else int(rng.integers(0, 9))]
        if kind == CONST: return [zipf_pick(cur_pool["const"])]
        if kind == CONSTS: return [(int(rng.integers(0, n_const)) if pol3_frac else zipf_pick(cur_pool["const"])), next_shift, nrows]
        if kind == PUB: return [int(rng.integers(0, n_pub))]
        if kind == POL: return list(pick_part() if partition else zipf_pick(cur_pool["pol"]))
        if kind == POLS:
            if partition and pol3_frac:      # the real program's shifted reads hardly repeat (467 distinct in 540): a uniform pick inside the run's slice
                w = np.asarray(sec_weights, dtype=np.float64)
                w = np.array([wi if cur_pool["sec"][i] else 0.0 for i, wi in enumerate(w)])
                si = int(rng.choice(len(sections), p=w / w.sum()))
                if pols_global and rng.random() < pols_global:       # ... or the section's columns in turn (the real program's 540 shifted reads hit 467 distinct columns:
                    sec = sections[si]                               # each "next row" value is read about once), in an order of its own
                    ncol = sec[2] if len(sec) > 2 else sec[1]
                    if si not in shifted_order:
                        shifted_order[si] = [list(rng.permutation(ncol)), 0]
                    order, cur_i = shifted_order[si]
                    c, st = sec[0] + int(order[cur_i % ncol]), sec[1]
                    shifted_order[si][1] = cur_i + 1
                else:
                    lst = cur_pool["sec"][si]
                    c, st = lst[int(rng.integers(0, len(lst)))]
            else:
                c, st = pick_part() if partition else zipf_pick(cur_pool["pols"])
            return [c, next_shift, nrows, st]
        raise ValueError(kind)

    def push(o, ar):
        ops.append(o); args.extend(ar)

    out_cursor = [0]
    push(13, [ACC, 1, vc])                       # acc = 1 + challenge[vc]
    LL0 = 10
    # long-lived values: generation g (slots LL0 + g * per_gen ...) is (re)defined when the program reaches g / ll_generations of its length
    per_gen = (-(-long_lived // ll_generations) if ll_generations else long_lived) if long_lived else 0
    ll_live = [0]                                # long-lived slots defined so far (the later generations overwrite nothing: own slots)

    def define_generation(g):
        lo, hi = g * per_gen, min(long_lived, (g + 1) * per_gen)
        for k in range(lo, hi):
            push(50, [LL0 + k] + list(zipf_pick(cur_pool["pol"])) + list(zipf_pick(cur_pool["pol"])))
        ll_live[0] = hi
        return hi - lo
    count, slot, eslot = 1 + define_generation(0), 0, 1
    next_gen = 1
    in_burst, BURST0 = [False], LL0 + long_lived + 4
    next_run = count + field_ops // n_runs
    op_w = {}
    if kind_weights is not None:                 # opcode choice biased by the kinds of its operands
        for lst in list(first.values()) + list(later.values()):
            for o in lst:
                op_w[o] = float(np.prod([kind_weights.get(k, 1.0) for k in OPS[o][1:] if k is not None]))

    def pick_op(lst):
        if kind_weights is None:
            return int(rng.choice(lst))
        w = np.array([op_w[o] for o in lst])
        return int(lst[int(rng.choice(len(lst), p=w / w.sum()))])
    while count < field_ops - 1:
        if count >= next_run:
            new_run()
            next_run += field_ops // n_runs
            if ll_generations == 0 and long_lived:      # the shared values belong to the run: redefined (same slots) from its own operands
                per_gen = long_lived
                count += define_generation(0)
            in_burst[0] = burst is not None and count <= burst[0] * field_ops < count + field_ops // n_runs
            if in_burst[0]:
                for k in range(burst[1]):
                    push(50, [BURST0 + k] + list(zipf_pick(cur_pool["pol"])) + list(zipf_pick(cur_pool["pol"])))
                count += burst[1]
        if next_gen < ll_generations and count >= next_gen * field_ops // ll_generations:
            count += define_generation(next_gen)
            next_gen += 1
        g = min(14, 1 + int(rng.geometric(1 / mean_len))) if mean_len <= 8 else min(24, 1 + int(rng.geometric(1 / mean_len)))
        cur = None
        for k in range(g):
            c = ("mul", "add", "sub")[int(rng.choice(3, p=list(class_p)))]
            o = pick_op(first[c] if cur is None else later[c])
            if in_burst[0] and cur is not None and rng.random() < 0.6:
                o = {"mul": 45, "add": 0, "sub": 21}[c]              # temp (op) temp: combines the burst's shared values
            d, a, b = OPS[o]
            dst = slot % 10
            slot += 1
            temps = [cur, cur] if cur is not None else []
            if cur is not None and a == T1 and b == T1:
                if in_burst[0] and rng.random() < 0.9:
                    temps = [cur, BURST0 + int(rng.integers(0, burst[1]))]
                elif long_lived and rng.random() < ll_use:
                    # a long-lived value of the current or the previous generation
                    lo = max(0, ll_live[0] - 2 * per_gen) if ll_generations > 1 else 0
                    temps = [cur, LL0 + int(rng.integers(lo, ll_live[0]))]
                elif k >= 2: temps = [cur, prev] # a product / sum of two values of this constraint
            ar = [dst] + gen_src(a, temps) + gen_src(b, temps)
            push(o, ar)
            prev, cur = (cur if cur is not None else dst), dst
            count += 1
        if base_out is not None and out_cursor[0] + 4 < (base_out[2] if len(base_out) > 2 else base_out[1]) and (count // 8) % 2 == 0:
            # this value is an output (stored, not accumulated: a value that is both keeps every such value alive until the
            # accumulation is scheduled, which the real programs do not do)
            push(100, [base_out[0] + out_cursor[0], base_out[1], cur])
            out_cursor[0] += 1
            count += 1
            continue
        is_ext = rng.random() < ext_frac
        if is_ext and pol3_frac and rng.random() < pol3_frac:   # a constraint over extension-valued polynomials (grand-product shaped)
            e1, e2 = 1 + eslot % 4, 1 + (eslot + 1) % 4
            eslot += 2

            def pol3():
                w3 = np.array([pol3_secs[i] if (len(sections[i]) > 2 and sections[i][2] or sections[i][1]) >= 3 and i < len(pol3_secs) else 0.0 for i in range(len(sections))])
                si = int(rng.choice(len(sections), p=w3 / w3.sum()))
                off, stride = sections[si][0], sections[si][1]
                cols = sections[si][2] if len(sections[si]) > 2 else stride
                c = int(rng.integers(0, cols // 3)) * 3      # (316 reads of 200 distinct polynomials: close to uniform over the section)
                return off + min(c, cols - 3), stride
            c3, st3 = pol3()
            push(59, [e1, cur, int(rng.integers(0, vc))])                   # e1 = value * challenge
            if rng.random() < 0.42: push(74, [e2, c3, next_shift, nrows, st3, e1])   # e2 = z' * e1   (66 of the real program's 316 extension reads are shifted)
            else: push(75, [e2, c3, st3, e1])                               # e2 = z * e1
            c3b, st3b = pol3() if rng.random() < 0.7 else (c3, st3)
            r = rng.random()
            if r < 0.45: push(75, [e1, c3b, st3b, e1])                      # e1 = z * e1
            elif r < 0.8: push(44, [e1, e1, c3b, st3b])                     # e1 = e1 - z
            else: push(41, [e1, c3b, st3b, int(rng.integers(0, 9))])        # e1 = z - number
            push(42, [e1, e2, e1])
            n3 = 4
            if rng.random() < 0.4:
                c3c, st3c = pol3()
                push(72, [e2, c3c, st3c, c3b, st3b])                        # e2 = h1 * z
                push(17, [e1, e1, e2])
                n3 += 2
            push(17, [ACC, e1, ACC])
            push(70, [ACC, vc, ACC])
            count += n3 + 2
        elif is_ext:                             # extension-valued constraint
            e1, e2 = 1 + eslot % 4, 1 + (eslot + 1) % 4
            eslot += 2
            push(59, [e1, cur, int(rng.integers(0, vc))])                   # e1 = value * challenge
            push(62, [e2] + list(zipf_pick(cur_pool["pol"])) + [int(rng.integers(0, vc))])  # e2 = pol * challenge
            push(71, [e1, e1, e2])                                          # e1 = e1 * e2
            push(42 if rng.random() < 0.4 else 17, [e1, e1, e2])            # e1 = e1 -/+ e2
            push(17, [ACC, e1, ACC])
            push(70, [ACC, vc, ACC])
            count += 6
        else:
            if base_out is not None:                                        # (84 is another opcode in the base-domain numbering)
                push(12, [ACC, cur, ACC]); push(70, [ACC, vc, ACC])
            else:
                push(84, [ACC, cur, ACC, ACC, vc, ACC])                     # acc = (value + acc) * challenge[vc]
            count += 2
    if base_out is not None:
        push(90, [base_out[0] + out_cursor[0], base_out[1], ACC, 0])
    else:
        push(69, [ACC])
    return np.array(ops, dtype=np.uint64), np.array([a % (1 << 64) for a in args], dtype=np.uint64)


# Generator parameters under which the synthetic step42ns program has the STATISTICS of the reference's real one (tools/chelpers_match.py
# fit, against profiles/r03_chelpers_step42ns_target.json; checked by tests/test_chelpers.py).  r04 re-fit: the generator
# now also reads extension-valued polynomials (opcodes 74 / 75 / 44 / 41 / 72: the real program has 316 such reads of 200 distinct polynomials)
# and takes its shifted reads from the section's columns in turn (the real program's 540 shifted reads hit 467 distinct columns).  Within
# 5 %: field operations, live words after the reschedule, kernels, estimated VALU instructions per row, OPERAND LOADS PER ROW (4 327 real /
# 4 310: was -13 %), DISTINCT OPERANDS (2 167 / 2 113: was -28 %), words through the kernel-boundary spill, Horner-chain steps, the shares of
# reads in cm1 / cm2 / cm3 / constants; the share of shifted reads -5.7 %.
ZKEVM_STEP42NS_FIT = dict(field_ops=17986, long_lived=4, sec_weights=[1.2442546603165616, 0.00098, 0.0285], kind_weights={CONST: 0.4, CONSTS: 1.0, POLS: 0.21, POL: 2.5, NUM: 0.35},
                          mean_len=5.5, ext_frac=0.065, run_ops=1194, pool_scale=1.0, zipf=0.7110090625, ll_generations=3, ll_use=0.4, burst=[0.5, 86], shared_scale=0.3,
                          partition=True, neighbour=0.049, class_p=[0.56, 0.23, 0.21], pol3_frac=1.0, pols_global=0.9)
ZKEVM_STEP42NS_FIT_TOLERANCE = {"field_ops": 0.05, "live_words_rescheduled": 0.05, "kernels": 0.05, "estimated_valu_per_row": 0.05, "frac_reads_cm1": 0.05,
                                "frac_reads_cm3": 0.05, "frac_reads_const": 0.05, "frac_reads_prime": 0.06, "spill_words_moved_per_row": 0.05,
                                "horner_chain_steps": 0.05, "frac_reads_cm2": 0.05, "operand_loads_per_row": 0.05, "distinct_operands": 0.05}


# ------------------------------------------------------------------ step52ns (zkevm.chelpers.step52ns.parser.cpp): arguments per opcode
NARGS52 = {0: 2, 1: 0, 2: 0, 3: 0, 4: 0, 5: 0, 6: 0, 7: 0, 8: 0, 9: 2, 10: 2, 11: 3, 12: 3, 13: 2, 14: 0, 15: 0}
FUSED52 = {16: [1, 10], 17: [1, 9], 18: [2, 11, 7], 19: [2, 13, 7], 20: [2, 12, 7]}


def nargs52_of(op):
    return sum(NARGS52[o] for o in FUSED52.get(op, [op]))


def max_eval52(ops, args):
    """Highest params.evals index a step52ns program reads (opcodes 11 / 12 carry it third, 13 second, 14 reads evals[0])."""
    top, ia = 0, 0
    for op in ops:
        for o in FUSED52.get(int(op), [int(op)]):
            if o in (11, 12):
                top = max(top, int(args[ia + 2]))
            elif o == 13:
                top = max(top, int(args[ia + 1]))
            ia += NARGS52[o]
    return top


def touched_addresses52(ops, args, rows, numpols):
    pols, cpols, ia = set(), set(), 0
    for op in ops:
        for o in FUSED52.get(int(op), [int(op)]):
            a = [int(v) for v in args[ia:ia + NARGS52[o]]]
            ia += NARGS52[o]
            for r in rows:
                if o in (0, 10, 11):
                    pols.add(a[0] + r * a[1])
                elif o in (9, 12):
                    pols.update(range(a[0] + r * a[1], a[0] + r * a[1] + 3))
                elif o == 13:
                    cpols.add(a[0] + r * numpols)
                elif o == 14:
                    cpols.add(5 + r * numpols)
    return pols, cpols, ia


def synthetic_program52(rng, sections, n_const, n_evals, length=60):
    """A random valid step52ns program in the shape of the generated one (two Horner chains over challenge 5 / challenge 6, the
    first closed by xDivXSubXi, the second by xDivXSubWXi, joined through tmp1), nothing dead, every opcode 0..20 present; about
    `length` chain links in the zkEVM program's proportions (16:17 = 760:136, 18:19:20 = 1227:336:202).
    sections: [(offset, stride)], each with at least 3 columns."""
    ops, args = [], []

    widths = np.array([st for _, st in sections], dtype=np.float64)

    def pol(three):                                # a column of the committed polynomials, uniformly over all of them
        off, stride = sections[int(rng.choice(len(sections), p=widths / widths.sum()))]
        return [off + int(rng.integers(0, stride - (2 if three else 0))), stride]

    def emit(o):
        for part in FUSED52.get(o, [o]):
            if part in (0, 10): args.extend(pol(False))
            elif part == 9: args.extend(pol(True))
            elif part == 11: args.extend(pol(False) + [int(rng.integers(0, n_evals))])
            elif part == 12: args.extend(pol(True) + [int(rng.integers(0, n_evals))])
            elif part == 13: args.extend([int(rng.integers(0, n_const)), int(rng.integers(0, n_evals))])
        ops.append(o)

    def chain5(k):
        for _ in range(k):
            emit(16 if rng.integers(0, 896) < 760 else 17)

    def chain6(k):
        for _ in range(k):
            v = int(rng.integers(0, 1765))
            emit(18 if v < 1227 else 19 if v < 1563 else 20)

    k5 = max(2, length * 9 // 27)
    k6 = max(4, length - k5)
    emit(0); emit(10)            # tmp = pol * c5 + pol
    chain5(k5)
    for o in (1, 9, 1, 10): emit(o)                       # the unfused spellings of 17 and 16
    emit(3)                      # tmp1 = tmp * c5
    emit(14)                     # tmp = const[5] - evals[0]
    chain6(k6 // 2)
    for o in (2, 13, 7, 2, 12, 7, 2, 11, 7): emit(o)      # the unfused spellings of 19, 20, 18
    emit(5)                      # tmp *= xDivXSubXi
    emit(8)                      # tmp = tmp1 + tmp
    emit(3)                      # tmp1 = tmp * c5
    emit(11); emit(4)            # tmp2 = pol - eval; tmp = tmp2 * c6
    chain6(k6 - k6 // 2)
    emit(6)                      # tmp *= xDivXSubWXi
    emit(8)
    emit(15)
    return np.array(ops, dtype=np.uint64), np.array(args, dtype=np.uint64)


# ------------------------------------------------------------------ the base-domain steps (step2prev / step3prev / step3)
# One opcode numbering for the three of them (zkevm.chelpers.step{2prev,3prev,3}.parser.cpp, *_parser_first_avx): cases 0-83 are
# step42ns's with pConstPols / x_n in place of pConstPols2ns / x_2ns; 84-85 two more temp operations; 86-100 results stored into a
# polynomial at the row; 101-114 the same stored at a shifted row; 115 (step3 only) the fusion [0, 50].
DPOL, DPOLS = 20, 21                              # destination kinds: pols[off + i * stride], pols[off + ((i + shift) % n) * stride]
NARGS_DST = {T1: 1, T3: 1, DPOL: 2, DPOLS: 4}
# opcode -> (operation, destination kind, destination dimension, source a, source b); sources in argument order after the destination
OPS_BASE_EXTRA = {
    84: ("add", T1, 1, T1, POLS), 85: ("mul", T1, 1, POLS, NUM),
    86: ("add", DPOL, 1, T1, T1), 87: ("add", DPOL, 1, T1, POL), 88: ("add", DPOL, 3, T1, T3), 89: ("add", DPOL, 3, POL3, T3),
    90: ("add", DPOL, 3, T3, CHAL), 92: ("sub", DPOL, 1, T1, T1), 93: ("sub", DPOL, 1, NUM, T1), 94: ("mul", DPOL, 1, T1, T1),
    95: ("mul", DPOL, 1, POL, T1), 96: ("mul", DPOL, 1, T1, CONST), 98: ("mul", DPOL, 3, T3, T3), 100: ("copy", DPOL, 1, T1, None),
    101: ("add", DPOLS, 1, T1, T1), 102: ("add", DPOLS, 1, T1, POL), 103: ("add", DPOLS, 3, T1, T3), 104: ("add", DPOLS, 3, POL3, T3),
    105: ("add", DPOLS, 3, T3, CHAL), 106: ("sub", DPOLS, 1, T1, T1), 107: ("sub", DPOLS, 1, NUM, T1), 108: ("mul", DPOLS, 1, T1, T1),
    109: ("mul", DPOLS, 1, POL, T1), 110: ("mul", DPOLS, 1, T1, CONST), 111: ("mul", DPOLS, 1, CONSTS, T1), 112: ("mul", DPOLS, 3, T3, T3),
    113: ("copy", DPOLS, 1, T1, None), 114: ("add", DPOLS, 1, T1, POLS),
}
FUSED_BASE = {115: [0, 50]}


def _cls42(o):
    return "add" if o <= 20 else "sub" if o <= 44 else "mul" if o <= 77 and o != 69 else "store" if o == 69 else "copy"


def decode_base(ops, args):
    """-> [(opcode, operation, dst kind, dst dim, dst args, [(kind, [args])])], and the number of arguments consumed."""
    out, ia = [], 0
    for op in ops:
        for o in FUSED_BASE.get(int(op), [int(op)]):
            if o in OPS_BASE_EXTRA:
                c, dk, dd, a, b = OPS_BASE_EXTRA[o]
            else:
                d, a, b = OPS[o]
                c, dk, dd = _cls42(o), d, (3 if d == T3 else 1)
            nd = NARGS_DST[dk]
            dargs = [int(v) for v in args[ia:ia + nd]]
            ia += nd
            srcs = []
            for k in (a, b):
                if k is None:
                    continue
                srcs.append((k, [int(v) for v in args[ia:ia + NARGS[k]]]))
                ia += NARGS[k]
            out.append((o, c, dk, dd, dargs, srcs))
    return out, ia


def touched_addresses_base(dec, rows, numpols):
    """Element indices a base-domain program reads / writes in `pols`, reads in the constant polynomials, for the given rows."""
    reads, writes, cpols = set(), set(), set()
    for (_, _, dk, dd, dargs, srcs) in dec:
        for r in rows:
            if dk == DPOL:
                writes.update(range(dargs[0] + r * dargs[1], dargs[0] + r * dargs[1] + dd))
            elif dk == DPOLS:
                b = dargs[0] + ((r + dargs[1]) % dargs[2]) * dargs[3]
                writes.update(range(b, b + dd))
            for k, a in srcs:
                if k in (POL, POL3):
                    b = a[0] + r * a[1]
                elif k in (POLS, POL3S):
                    b = a[0] + ((r + a[1]) % a[2]) * a[3]
                elif k == CONST:
                    cpols.add(a[0] + r * numpols); continue
                elif k == CONSTS:
                    cpols.add(a[0] + ((r + a[1]) % a[2]) * numpols); continue
                else:
                    continue
                reads.update(range(b, b + (3 if k in (POL3, POL3S) else 1)))
    return reads, writes, cpols


def synthetic_program_base(rng, nrows, sections, out_section, n_const, n_chal, n_pub, n_ops=260):
    """A random valid program in the base-domain steps' numbering: every opcode 0..115 that the three steps can use (not 69, 91, 97,
    99) occurs; every stored result goes to its own column(s) of `out_section` = (offset, stride) -- a column is written once, with
    one shift -- and some later operands read a stored element back (same shift and dimension: the only kind the zkEVM programs have).
    sections: [(offset, stride)] of input polynomials."""
    ops, args = [], []
    def1, def3 = set(), set()
    out_off, out_stride = out_section
    cursor = [0]
    last_dst = [0]
    recent1, recent3 = [], []
    stored = {POL: [], POL3: [], POLS: [], POL3S: []}

    def gen_src(kind):
        if kind == T1: return [int(rng.choice(recent1[-6:]))]      # recently defined: the live ranges stay short, as in the real programs
        if kind == T3: return [int(rng.choice(recent3[-4:]))]
        if kind == NUM: return [int(rng.integers(0, 1 << 64, dtype=np.uint64)) if rng.random() < 0.5 else int(rng.integers(0, 5))]
        if kind == CONST: return [int(rng.integers(0, n_const))]
        if kind == CONSTS: return [int(rng.integers(0, n_const)), int(rng.integers(1, 4)), nrows]
        if kind == CHAL: return [int(rng.integers(0, n_chal))]
        if kind == PUB: return [int(rng.integers(0, n_pub))]
        if kind in stored and stored[kind] and rng.random() < (0.3 if n_ops < 1000 else 0.03):   # read a stored element back (a recent one)
            recent = stored[kind][-8:]
            return list(recent[int(rng.integers(0, len(recent)))])
        off, stride = sections[int(rng.integers(0, len(sections)))]
        col = int(rng.integers(0, stride - (2 if kind in (POL3, POL3S) else 0)))
        if kind in (POL, POL3): return [off + col, stride]
        if kind in (POLS, POL3S): return [off + col, int(rng.integers(1, 4)), nrows, stride]
        return []

    def emit(o):
        if o in OPS_BASE_EXTRA:
            c, dk, dd, a, b = OPS_BASE_EXTRA[o]
        else:
            d, a, b = OPS[o]
            dk, dd = d, (3 if d == T3 else 1)
        for k in (a, b):
            if (k == T1 and not def1) or (k == T3 and not def3):
                return False
        if dk in (DPOL, DPOLS):
            if cursor[0] + dd > out_stride - 16:   # the last columns are kept for the accumulators and the temporaries at the end
                return False
            col = out_off + cursor[0]
            cursor[0] += dd
            if dk == DPOL:
                dargs, new_entry = [col, out_stride], (POL3 if dd == 3 else POL, (col, out_stride))
            else:
                sh = int(rng.integers(1, 4))
                dargs, new_entry = [col, sh, nrows, out_stride], (POL3S if dd == 3 else POLS, (col, sh, nrows, out_stride))
        else:
            slot = int(rng.integers(0, 12 if dk == T1 else 6))
            dargs, new_entry = [slot], None
        if o == 70:
            ar = dargs + gen_src(CHAL) + gen_src(T3)
        else:
            ar = dargs + gen_src(a) + (gen_src(b) if b is not None else [])
        if new_entry is not None:                   # readable only by LATER operations
            stored[new_entry[0]].append(new_entry[1])
        ops.append(o); args.extend(ar)
        last_dst[0] = dargs[0]
        if dk == T1: def1.add(dargs[0]); recent1.append(dargs[0])
        elif dk == T3: def3.add(dargs[0]); recent3.append(dargs[0])
        return True

    for o in (79, 82, 81, 13, 16):          # a few temporaries to start from
        emit(o)
    ACC1, ACC3 = 12, 6                      # accumulators beyond the randomly used slots: every result is folded in, nothing is dead
    ops.append(81); args.extend([ACC1, 1])
    ops.append(13); args.extend([ACC3, 1, 0])
    cand = [o for o in range(0, 86) if o != 69] + [o for o in OPS_BASE_EXTRA if o >= 86]
    pending = list(cand)
    rng.shuffle(pending)
    n = 0
    while n < n_ops:
        o = pending.pop() if pending else int(rng.choice(cand))
        if o in FUSED_BASE:
            continue
        if emit(o):
            n += 1
            dk = OPS_BASE_EXTRA[o][1] if o in OPS_BASE_EXTRA else OPS[o][0]
            if dk == T1:
                ops.append(12); args.extend([ACC3, last_dst[0], ACC3]); n += 1    # one accumulator chain: two would keep each other's operands alive
            elif dk == T3:
                ops.append(17); args.extend([ACC3, ACC3, last_dst[0]]); n += 1
    # the fused opcode, then every live temporary to a column of its own so that nothing is dead
    if def1:
        ar = [int(rng.integers(0, 12))] + gen_src(T1) + gen_src(T1)
        ar += [int(rng.integers(0, 12))] + gen_src(POL) + gen_src(POL)
        ops.append(115); args.extend(ar)
        def1.update([ar[0], ar[3]])
    for t in sorted(def1)[:4]:
        if cursor[0] + 4 <= out_stride:
            ops.append(100); args.extend([out_off + cursor[0], out_stride, t]); cursor[0] += 1
    if cursor[0] + 3 <= out_stride:            # the extension accumulator: acc3 + challenge 1 -> three columns
        ops.append(90); args.extend([out_off + cursor[0], out_stride, ACC3, 1]); cursor[0] += 3
    return np.array(ops, dtype=np.uint64), np.array([a % (1 << 64) for a in args], dtype=np.uint64)
