"""the generated constraint evaluators (chelpers) on the GPU -- the step42ns program interpreter.

PARITY UNPINNED: the reference holds no input / output pair for this step.  What is checked:
  * the product (translator + instruction semantics, csrc/chelpers.hip) against the oracle's opcode-by-opcode restatement
    of the reference interpreter (oracle/chelpers_oracle.c) on synthetic programs that use every opcode -- on the CPU
    through the host debug executor, on the GPU through the C ABI over 2^16 rows;
  * where /root/reference is present: the three transcriptions of the opcode table against the argument bookkeeping of the
    reference's source text, and product vs oracle on the reference's OWN program (11 959 opcodes) over a sparse address
    space the size of the zkEVM memory map."""
import mmap
import tempfile
import os
import re

import numpy as np
import pytest

import glo
import chelpers_programs as cp

P = glo.P
REF_CPP = "/root/reference/src/starkpil/zkevm/chelpers/zkevm.chelpers.step42ns.parser.cpp"
REF_HPP = "/root/reference/src/starkpil/zkevm/chelpers/zkevm.chelpers.step42ns.parser.hpp"
needs_ref = pytest.mark.skipif(not os.path.exists(REF_HPP), reason="/root/reference not present")


def _synthetic_case(seed, nrows, passes=3):
    rng = np.random.default_rng(seed)
    sections = [(0, 40), (nrows * 40, 9), (nrows * 49, 3)]           # three sections of `pols`: 40, 9 and 3 columns
    n_const, n_chal, n_pub = 7, 5, 4
    ops, args = cp.synthetic_program(rng, nrows, sections, n_const, n_chal, n_pub, passes=passes)
    pols = glo.rand_fe(rng, nrows * 52, canonical=False)
    cpols = glo.rand_fe(rng, nrows * n_const)
    chal = glo.rand_fe(rng, n_chal * 3)
    pub = glo.rand_fe(rng, n_pub)
    x = glo.rand_fe(rng, nrows * 2)                                   # x_stride 2: a strided view, as Polinomial allows
    zhinv = glo.rand_fe(rng, 4)
    return ops, args, pols, cpols, n_const, chal, pub, x, 2, zhinv


def _synthetic_sections(nrows):
    return [(0, 40, nrows), (nrows * 40, 9, nrows), (nrows * 49, 3, nrows)]


def test_opcode_table_argument_counts():
    micro_total = sum(len(cp.FUSED.get(o, [o])) for o in range(89))
    assert micro_total == 84 + 2 + 2 + 5 + 8 + 12
    assert cp.nargs_of(5) == 9 and cp.nargs_of(69) == 1 and cp.nargs_of(67) == 2 and cp.nargs_of(86) == 3 + 4 + 3 + 3 + 6


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_translated_program_matches_oracle_on_the_host(seed):
    """Translator (decode, copy forwarding, reschedule, temp re-allocation) + instruction semantics vs the naive oracle."""
    import mi_stark
    nrows = 64
    ops, args, pols, cpols, n_const, chal, pub, x, xs, zhinv = _synthetic_case(seed, nrows)
    assert set(range(89)) - {69} <= set(int(o) for o in ops) | {69}, "every opcode is used"
    want = np.zeros(nrows * 3, dtype=np.uint64)
    glo.chelpers_step42ns(ops, args, pols, cpols, n_const, chal, pub, x, xs, zhinv, want, 0, nrows)
    prog = mi_stark.ChelpersProgram(None, ops, args)
    got = np.zeros(nrows * 3, dtype=np.uint64)
    prog.run_host(pols, cpols, n_const, chal, pub, x, xs, zhinv, got, np.arange(nrows))
    assert np.array_equal(got, want)
    assert want.any() and prog.stats["opcodes"] == ops.size and prog.stats["instructions_per_row"] <= prog.stats["field_ops"]
    prog.close()


def _microops42(ops, args):
    """A step42ns table as the field operations mi_chelpers_compile_micro takes (what host/steps_tracer.hpp records from per-row code)."""
    K = {cp.T1: "T1", cp.T3: "T3", cp.POL: "POL", cp.POLS: "POLS", cp.NUM: "NUM", cp.CONST: "CONST", cp.CONSTS: "CONSTS", cp.CHAL: "CHAL", cp.PUB: "PUB",
         cp.POL3: "POL3", cp.POL3S: "POL3S", cp.X: "X"}
    out = []
    for (o, d, slot, srcs) in cp.decode(ops, args)[0]:
        a = (K[srcs[0][0]], srcs[0][1])
        if o == 69:
            out.append(("STOREQ", "Q", 0, a, ("ZHINV", [])))
            continue
        b = (K[srcs[1][0]], srcs[1][1]) if len(srcs) > 1 else None
        out.append((cp._cls42(o).upper(), K[d], slot, a, b))
    return out


@pytest.mark.parametrize("seed", [1, 4])
def test_a_program_given_as_field_operations_matches_its_table(seed):
    """mi_chelpers_compile_micro: the same program handed over as field operations instead of a table computes the same rows (translated
    and lowered host executors), and malformed operations are refused."""
    import mi_stark
    nrows = 64
    ops, args, pols, cpols, n_const, chal, pub, x, xs, zhinv = _synthetic_case(seed, nrows)
    want = np.zeros(nrows * 3, dtype=np.uint64)
    glo.chelpers_step42ns(ops, args, pols, cpols, n_const, chal, pub, x, xs, zhinv, want, 0, nrows)
    mops = _microops42(ops, args)
    prog = mi_stark.ChelpersProgram.from_microops(None, mops, sections=_synthetic_sections(nrows), n_const=n_const, nrows_ext=nrows)
    got = np.zeros(nrows * 3, dtype=np.uint64)
    prog.run_host(pols, cpols, n_const, chal, pub, x, xs, zhinv, got, np.arange(nrows))
    assert np.array_equal(got, want) and want.any()
    assert prog.stats["field_ops"] == len(mops)
    prog.close()
    bad = list(mops)
    bad[3] = ("MUL", "T1", 5, ("CHAL", [0]), ("T1", [0]))            # an extension operand into a base-field destination
    with pytest.raises(mi_stark.MiStarkError, match="extension operand needs an extension destination"):
        mi_stark.ChelpersProgram.from_microops(None, bad)
    with pytest.raises(mi_stark.MiStarkError, match="operand kind out of range"):
        mi_stark.ChelpersProgram.from_microops(None, [("ADD", "T1", 0, (77, [0]), ("NUM", [1]))])
    with pytest.raises(mi_stark.MiStarkError, match="STOREP"):
        mi_stark.ChelpersProgram.from_microops(None, [("COPY", "T1", 0, ("NUM", [1]), None), ("STOREP", "DPOL", 0, ("T1", [0]), ("DPOL", [0, 5]))])   # step42ns stores q only


def test_compile_rejects_malformed_tables():
    import mi_stark
    ops, args, *_ = _synthetic_case(5, 16, passes=1)
    with pytest.raises(mi_stark.MiStarkError, match="argument"):
        mi_stark.ChelpersProgram(None, ops, args[:-1])                  # one argument short
    with pytest.raises(mi_stark.MiStarkError, match="argument"):
        mi_stark.ChelpersProgram(None, ops, np.append(args, np.uint64(0)))   # one too many
    with pytest.raises(mi_stark.MiStarkError, match="unknown opcode"):
        mi_stark.ChelpersProgram(None, np.append(ops, np.uint64(97)), args)
    with pytest.raises(mi_stark.MiStarkError, match="before it is written"):
        mi_stark.ChelpersProgram(None, np.array([0, 12, 69], dtype=np.uint64), np.array([0, 1, 2, 0, 0, 0, 0], dtype=np.uint64))


@needs_ref
def test_opcode_tables_agree_with_the_reference_source_text():
    """Per `case N:` of step42ns_parser_first_avx: the sum of its `i_args += k` statements is what all three
    transcriptions of the opcode table must consume; the calls it makes name the operation."""
    src = open(REF_CPP).read()
    body = src[src.index("void ZkevmSteps::step42ns_parser_first_avx("):src.index("void ZkevmSteps::step42ns_parser_first(")]
    cases = re.split(r"\n\s*case (\d+):", body)[1:]
    seen = {}
    for num, text in zip(cases[0::2], cases[1::2]):
        text = text.split("default:")[0]
        seen[int(num)] = (sum(int(k) for k in re.findall(r"i_args \+= (\d+);", text)), re.findall(r"Goldilocks3?::(\w+?)_avx", text))
    assert sorted(seen) == list(range(89))
    for op, (n, calls) in seen.items():
        assert cp.nargs_of(op) == n, (op, n, cp.nargs_of(op))
        kinds = [("add" if 0 <= o <= 20 else "sub" if 21 <= o <= 44 else "mul" if 45 <= o <= 77 else "copy") for o in cp.FUSED.get(op, [op])]
        if op != 69:                                   # 69 is a plain Goldilocks3::mul in a loop
            assert [re.match(r"(add|sub|mul|copy)", c).group(1) for c in calls] == kinds, (op, calls)


@needs_ref
def test_reference_program_translates_and_matches_oracle():
    """The reference's own step42ns program: the product consumes exactly NARGS_ arguments, the reschedule brings the live
    temporaries from ~1 400 words down to what fits a lane's LDS share, and product == oracle on sampled rows with random
    memory contents at every address the program touches (zkEVM memory map: a sparse 300 GB mapping)."""
    import mi_stark
    ops, args = cp.parse_reference_tables(open(REF_HPP).read())
    assert ops.size == 11959 and args.size == 68237
    micro, used = cp.decode(ops, args)
    assert used == args.size and len(micro) == 19198
    prog = mi_stark.ChelpersProgram(None, ops, args)
    st = prog.stats
    assert st["field_ops"] == 19198 and st["live_words_as_generated"] > 1000 and st["live_words_rescheduled"] <= 128, st
    assert st["base_temps"] + 3 * st["ext_temps"] <= 160, st
    n_ext, numpols, rows = 1 << 24, 218, [0, 1, 5, (1 << 24) - 3, (1 << 24) - 1, 123456]
    rng = np.random.default_rng(42)
    pa, ca = cp.touched_addresses(micro, rows, numpols)
    size = (max(pa) + 8) * 8
    m = mmap.mmap(-1, size, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS | getattr(mmap, "MAP_NORESERVE", 0x4000))
    pols = np.frombuffer(m, dtype=np.uint64)
    idx = np.fromiter(pa, dtype=np.int64)
    pols[idx] = glo.rand_fe(rng, idx.size)
    mc = mmap.mmap(-1, (max(ca) + 8) * 8, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS | getattr(mmap, "MAP_NORESERVE", 0x4000))
    cpols = np.frombuffer(mc, dtype=np.uint64)
    cidx = np.fromiter(ca, dtype=np.int64)
    cpols[cidx] = glo.rand_fe(rng, cidx.size)
    mx = mmap.mmap(-1, n_ext * 8, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS | getattr(mmap, "MAP_NORESERVE", 0x4000))
    x = np.frombuffer(mx, dtype=np.uint64)
    x[rows] = glo.rand_fe(rng, len(rows))
    chal, pub, zhinv = glo.rand_fe(rng, 8 * 3), glo.rand_fe(rng, 64), glo.rand_fe(rng, 2)
    mq1 = mmap.mmap(-1, n_ext * 24, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS | getattr(mmap, "MAP_NORESERVE", 0x4000))
    mq2 = mmap.mmap(-1, n_ext * 24, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS | getattr(mmap, "MAP_NORESERVE", 0x4000))
    want, got = np.frombuffer(mq1, dtype=np.uint64), np.frombuffer(mq2, dtype=np.uint64)
    for r in rows:
        glo.chelpers_step42ns(ops, args, pols, cpols, numpols, chal, pub, x, 1, zhinv, want, r, 1)
    prog.run_host(pols, cpols, numpols, chal, pub, x, 1, zhinv, got, np.array(rows))
    for r in rows:
        assert np.array_equal(got[3 * r:3 * r + 3], want[3 * r:3 * r + 3]) and want[3 * r:3 * r + 3].any(), r
    prog.close()
    # the same program as the native backend lowers it: the (acc + constraint) * vc accumulation becomes Horner-chain pieces
    N = 1 << 23
    prog = mi_stark.ChelpersProgram(None, ops, args, sections=[(1435 * N, 665, 2 * N), (2765 * N, 128, 2 * N), (3021 * N, 371, 2 * N)], n_const=numpols, nrows_ext=2 * N)
    ls = prog.lower_stats()
    assert ls["horner_chain_steps"] > 4000 and ls["estimated_valu_per_row"] < 350000 and ls["kernels"] >= 10, ls
    for cc in (0, 9000):
        for r in rows:
            got[3 * r:3 * r + 3] = 0
        prog.run_lowered_host(pols, cpols, numpols, chal, pub, x, 1, zhinv, got, np.array(rows), chunk_cost=cc)
        for r in rows:
            assert np.array_equal(got[3 * r:3 * r + 3], want[3 * r:3 * r + 3]), (cc, r)
    # ... and the generated kernels compile (two of the 28 here: a share of a parallel build; the full build takes ~25 s on 8 cores)
    with tempfile.TemporaryDirectory() as td:
        prog.precompile_shard(3, 14, cache_dir=td)
        assert len([f for f in os.listdir(td) if f.endswith(".hsaco")]) == 2
    prog.close()


@pytest.mark.gpu
def test_chelpers_on_gpu_matches_oracle_over_2pow16_rows():
    import mi_stark
    ctx = mi_stark.Context(0)
    nrows = 1 << 16
    for seed in (11, 12):
        ops, args, pols, cpols, n_const, chal, pub, x, xs, zhinv = _synthetic_case(seed, nrows, passes=4)
        want = np.zeros(nrows * 3, dtype=np.uint64)
        glo.chelpers_step42ns(ops, args, pols, cpols, n_const, chal, pub, x, xs, zhinv, want, 0, nrows)
        prog = mi_stark.ChelpersProgram(ctx, ops, args, sections=_synthetic_sections(nrows), n_const=n_const, nrows_ext=nrows)
        q = ctx.to_device(np.full(nrows * 3 + 6, 0xABCD, dtype=np.uint64))
        d_pols, d_c, d_x = ctx.to_device(pols), ctx.to_device(cpols), ctx.to_device(x)
        prog.run(d_pols, d_c, n_const, chal, pub, d_x, xs, zhinv, q, 0, nrows)
        got = ctx.to_host(q)
        assert np.array_equal(got[:nrows * 3], want) and np.all(got[nrows * 3:] == 0xABCD), seed
        # a row range in the middle, not a multiple of the workgroup size: rows outside it stay untouched
        q2 = ctx.to_device(np.zeros(nrows * 3, dtype=np.uint64))
        prog.run(d_pols, d_c, n_const, chal, pub, d_x, xs, zhinv, q2, 1000, 777)
        got2 = ctx.to_host(q2)
        assert np.array_equal(got2[3000:3000 + 3 * 777], want[3000:3000 + 3 * 777]) and not got2[:3000].any() and not got2[3000 + 3 * 777:].any()
        prog.close()
    # operands outside every declared section are refused at compile time
    with pytest.raises(mi_stark.MiStarkError, match="none of the declared sections"):
        mi_stark.ChelpersProgram(ctx, ops, args, sections=_synthetic_sections(nrows)[:2], n_const=n_const, nrows_ext=nrows)
    ctx.close()


# ------------------------------------------------------------------ step52ns (the FRI polynomial)
REF52_CPP = "/root/reference/src/starkpil/zkevm/chelpers/zkevm.chelpers.step52ns.parser.cpp"
REF52_HPP = "/root/reference/src/starkpil/zkevm/chelpers/zkevm.chelpers.step52ns.parser.hpp"


def _case52(seed, nrows):
    rng = np.random.default_rng(seed)
    sections = [(0, 40), (nrows * 40, 9), (nrows * 49, 3)]
    n_const, n_evals = 7, 6
    ops, args = cp.synthetic_program52(rng, sections, n_const, n_evals, length=80)
    pols = glo.rand_fe(rng, nrows * 52, canonical=False)
    cpols = glo.rand_fe(rng, nrows * n_const)
    chal, evals = glo.rand_fe(rng, 7 * 3), glo.rand_fe(rng, n_evals * 3)
    xd, xdw = glo.rand_fe(rng, nrows * 3), glo.rand_fe(rng, nrows * 3)
    return ops, args, pols, cpols, n_const, chal, evals, xd, xdw


@pytest.mark.parametrize("seed", [1, 2])
def test_step52ns_translated_program_matches_oracle_on_the_host(seed):
    import mi_stark
    nrows = 32
    ops, args, pols, cpols, n_const, chal, evals, xd, xdw = _case52(seed, nrows)
    assert set(range(21)) <= set(int(o) for o in ops)
    want, got = np.zeros(nrows * 3, dtype=np.uint64), np.zeros(nrows * 3, dtype=np.uint64)
    glo.chelpers_step52ns(ops, args, pols, cpols, n_const, chal, evals, xd, xdw, want, 0, nrows)
    prog = mi_stark.ChelpersProgram(None, ops, args, step=mi_stark.MI_CHELPERS_STEP52NS)
    prog.run52_host(pols, cpols, n_const, chal, evals, xd, xdw, got, np.arange(nrows))
    assert np.array_equal(got, want) and want.any()
    prog.close()


@needs_ref
def test_step52ns_tables_agree_with_the_reference_and_its_program_matches_the_oracle():
    import mi_stark
    src = open(REF52_CPP).read()
    body = src[src.index("void ZkevmSteps::step52ns_parser_first_avx("):]
    body = body[:body.index("void ZkevmSteps::", 10)]
    cases = re.split(r"\n\s*case (\d+):", body)[1:]
    for num, text in zip(cases[0::2], cases[1::2]):
        assert cp.nargs52_of(int(num)) == sum(int(k) for k in re.findall(r"i_args \+= (\d+);", text.split("default:")[0])), num
    ops, args = cp.parse_reference_tables(open(REF52_HPP).read(), "op52", "args52")
    assert ops.size == 2675 and args.size == 6761
    n_ext, numpols, rows = 1 << 24, 218, [0, 7, (1 << 24) - 1, 999999]
    pa, ca, used = cp.touched_addresses52(ops, args, rows, numpols)
    assert used == args.size
    rng = np.random.default_rng(9)

    def sparse(n_elems):
        return np.frombuffer(mmap.mmap(-1, n_elems * 8, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS | getattr(mmap, "MAP_NORESERVE", 0x4000)), dtype=np.uint64)
    pols, cpols, xd, xdw, want, got = sparse(max(pa) + 8), sparse(max(ca) + 8), sparse(n_ext * 3), sparse(n_ext * 3), sparse(n_ext * 3), sparse(n_ext * 3)
    pols[np.fromiter(pa, dtype=np.int64)] = glo.rand_fe(rng, len(pa))
    cpols[np.fromiter(ca, dtype=np.int64)] = glo.rand_fe(rng, len(ca))
    for r in rows:
        xd[3 * r:3 * r + 3], xdw[3 * r:3 * r + 3] = glo.rand_fe(rng, 3), glo.rand_fe(rng, 3)
    n_evals = cp.max_eval52(ops, args) + 1
    chal, evals = glo.rand_fe(rng, 7 * 3), glo.rand_fe(rng, n_evals * 3)
    N = 1 << 23
    prog = mi_stark.ChelpersProgram(None, ops, args, sections=[(1435 * N, 665, 2 * N), (2765 * N, 128, 2 * N), (3021 * N, 371, 2 * N), (3763 * N, 6, 2 * N)],
                                    n_const=numpols, nrows_ext=2 * N, step=mi_stark.MI_CHELPERS_STEP52NS)
    assert prog.stats["field_ops"] > 6000 and prog.stats["lds_ext_temps"] <= 8
    for r in rows:
        glo.chelpers_step52ns(ops, args, pols, cpols, numpols, chal, evals, xd, xdw, want, r, 1)
    prog.run52_host(pols, cpols, numpols, chal, evals, xd, xdw, got, np.array(rows))
    for r in rows:
        assert np.array_equal(got[3 * r:3 * r + 3], want[3 * r:3 * r + 3]) and want[3 * r:3 * r + 3].any(), r
    # as lowered for the native backend: three Horner chains, the (polynomial - evaluation) leaves folded into per-piece constants
    ls = prog.lower_stats()
    assert ls["horner_chain_steps"] > 5000 and ls["folded_leaves"] > 1700 and ls["estimated_valu_per_row"] < 120000, ls
    for cc in (0, 7000):
        for r in rows:
            got[3 * r:3 * r + 3] = 0
        prog.run52_lowered_host(pols, cpols, numpols, chal, evals, xd, xdw, got, np.array(rows), chunk_cost=cc)
        for r in rows:
            assert np.array_equal(got[3 * r:3 * r + 3], want[3 * r:3 * r + 3]), (cc, r)
    with tempfile.TemporaryDirectory() as td:      # 3 338 linear terms in 3 sums + one generated kernel for the remaining ten operations
        st = prog.build_native(cache_dir=td)
        assert st["kernels"] == 1 and prog.lower_stats()["linear_terms"] > 3000 and prog.lower_stats()["linear_sums"] == 3
    prog.close()


@pytest.mark.gpu
def test_step52ns_on_gpu_matches_oracle():
    import mi_stark
    ctx = mi_stark.Context(0)
    nrows = 1 << 14
    ops, args, pols, cpols, n_const, chal, evals, xd, xdw = _case52(21, nrows)
    want = np.zeros(nrows * 3, dtype=np.uint64)
    glo.chelpers_step52ns(ops, args, pols, cpols, n_const, chal, evals, xd, xdw, want, 0, nrows)
    prog = mi_stark.ChelpersProgram(ctx, ops, args, sections=_synthetic_sections(nrows), n_const=n_const, nrows_ext=nrows, step=mi_stark.MI_CHELPERS_STEP52NS)
    f = ctx.zeros(nrows * 3)
    for ev in (evals, glo.rand_fe(np.random.default_rng(5), evals.size)):   # the evaluations are patched in per run
        glo.chelpers_step52ns(ops, args, pols, cpols, n_const, chal, ev, xd, xdw, want, 0, nrows)
        prog.run52(ctx.to_device(pols), ctx.to_device(cpols), n_const, chal, ev, ctx.to_device(xd), ctx.to_device(xdw), f, 0, nrows)
        assert np.array_equal(ctx.to_host(f), want)
    prog.close()
    ctx.close()


# ------------------------------------------------------------------ native-code backend (chelpers_native.hip)
@pytest.mark.parametrize("chunk_cost,lin_min", [(0, None), (2500, None), (400, None), (0, 1), (300, 1)])
def test_lowered_programs_match_oracle_on_the_host(chunk_cost, lin_min, monkeypatch):
    """Chains, pieces cut at kernel boundaries, folded leaves, coefficient / K tables and spill lists (poisoned between kernels);
    lin_min = 1: the polynomial-leaf chain terms go to the linear kernel's tables whatever their number."""
    import mi_stark
    if lin_min is not None:
        monkeypatch.setenv("MI_CHELPERS_LIN_MIN", str(lin_min))
    nrows = 48
    ops, args, pols, cpols, n_const, chal, pub, x, xs, zhinv = _synthetic_case(5, nrows, passes=3)
    want, got = np.zeros(nrows * 3, dtype=np.uint64), np.zeros(nrows * 3, dtype=np.uint64)
    glo.chelpers_step42ns(ops, args, pols, cpols, n_const, chal, pub, x, xs, zhinv, want, 0, nrows)
    prog = mi_stark.ChelpersProgram(None, ops, args, sections=_synthetic_sections(nrows), n_const=n_const, nrows_ext=nrows)
    prog.run_lowered_host(pols, cpols, n_const, chal, pub, x, xs, zhinv, got, np.arange(nrows), chunk_cost=chunk_cost)
    assert np.array_equal(got, want)
    prog.close()
    ops, args, pols, cpols, n_const, chal, evals, xd, xdw = _case52(6, nrows)
    glo.chelpers_step52ns(ops, args, pols, cpols, n_const, chal, evals, xd, xdw, want, 0, nrows)
    prog = mi_stark.ChelpersProgram(None, ops, args, sections=_synthetic_sections(nrows), n_const=n_const, nrows_ext=nrows, step=mi_stark.MI_CHELPERS_STEP52NS)
    assert prog.lower_stats(chunk_cost)["horner_chain_steps"] > 100
    got[:] = 0
    prog.run52_lowered_host(pols, cpols, n_const, chal, evals, xd, xdw, got, np.arange(nrows), chunk_cost=chunk_cost)
    assert np.array_equal(got, want)
    prog.close()


def test_native_backend_builds_without_a_gpu(tmp_path):
    """The generated kernels compile (hiprtc needs no device); a second build of the same program is served from the cache."""
    import mi_stark
    nrows = 1 << 8
    ops, args, *_ = _synthetic_case(3, nrows, passes=1)
    stats = []
    for _ in range(2):
        prog = mi_stark.ChelpersProgram(None, ops, args, sections=_synthetic_sections(nrows), n_const=7, nrows_ext=nrows)
        stats.append(prog.build_native(cache_dir=str(tmp_path), chunk_cost=6000))
        with pytest.raises(mi_stark.MiStarkError, match="already built"):
            prog.build_native(cache_dir=str(tmp_path))
        prog.close()
    assert stats[0]["kernels"] >= 2 and stats[0]["cache_hits"] == 0 and stats[1]["cache_hits"] == stats[1]["kernels"] == stats[0]["kernels"]
    assert stats[0]["code_bytes"] == stats[1]["code_bytes"] > 0


@pytest.mark.gpu
def test_native_backend_on_a_tiny_domain_with_linear_terms(tmp_path, monkeypatch):
    """128 rows, one tile per batch: a step52ns program whose polynomial terms go through the linear kernel, and a step42ns
    program (x_2ns at a stride of 2, shifted reads wrapping around the end of the domain) through generated kernels only."""
    import mi_stark
    monkeypatch.setenv("MI_CHELPERS_LIN_MIN", "1")
    ctx = mi_stark.Context(0)
    nrows = 128
    ctx.set_chelpers_batch_rows(64)
    ops, args, pols, cpols, n_const, chal, evals, xd, xdw = _case52(78, nrows)
    want = np.zeros(nrows * 3, dtype=np.uint64)
    glo.chelpers_step52ns(ops, args, pols, cpols, n_const, chal, evals, xd, xdw, want, 0, nrows)
    prog = mi_stark.ChelpersProgram(ctx, ops, args, sections=_synthetic_sections(nrows), n_const=n_const, nrows_ext=nrows, step=mi_stark.MI_CHELPERS_STEP52NS)
    ls = prog.lower_stats()
    assert ls["linear_terms"] > 50 and 1 <= ls["linear_sums"] <= 4, ls
    prog.build_native(cache_dir=str(tmp_path))
    f = ctx.to_device(np.full(nrows * 3 + 6, 0xABCD, dtype=np.uint64))
    prog.run52(ctx.to_device(pols), ctx.to_device(cpols), n_const, chal, evals, ctx.to_device(xd), ctx.to_device(xdw), f, 0, nrows)
    got = ctx.to_host(f)
    assert np.array_equal(got[:nrows * 3], want) and np.all(got[nrows * 3:] == 0xABCD)
    prog.close()
    ops, args, pols, cpols, n_const, chal, pub, x, xs, zhinv = _synthetic_case(77, nrows, passes=2)
    glo.chelpers_step42ns(ops, args, pols, cpols, n_const, chal, pub, x, xs, zhinv, want, 0, nrows)
    prog = mi_stark.ChelpersProgram(ctx, ops, args, sections=_synthetic_sections(nrows), n_const=n_const, nrows_ext=nrows)
    prog.build_native(cache_dir=str(tmp_path), chunk_cost=4000)
    q = ctx.to_device(np.full(nrows * 3 + 6, 0xABCD, dtype=np.uint64))
    prog.run(ctx.to_device(pols), ctx.to_device(cpols), n_const, chal, pub, ctx.to_device(x), xs, zhinv, q, 0, nrows)
    got = ctx.to_host(q)
    assert np.array_equal(got[:nrows * 3], want) and np.all(got[nrows * 3:] == 0xABCD)
    prog.close()
    ctx.close()


@pytest.mark.gpu
def test_native_step42ns_matches_oracle_and_interpreter(tmp_path):
    import mi_stark
    ctx = mi_stark.Context(0)
    nrows = 1 << 13
    ops, args, pols, cpols, n_const, chal, pub, x, xs, zhinv = _synthetic_case(31, nrows, passes=3)
    want = np.zeros(nrows * 3, dtype=np.uint64)
    glo.chelpers_step42ns(ops, args, pols, cpols, n_const, chal, pub, x, xs, zhinv, want, 0, nrows)
    prog = mi_stark.ChelpersProgram(ctx, ops, args, sections=_synthetic_sections(nrows), n_const=n_const, nrows_ext=nrows)
    st = prog.build_native(cache_dir=str(tmp_path), chunk_cost=7000)
    assert st["kernels"] >= 3                      # temporaries cross chunk boundaries through the spill
    d_pols, d_c, d_x = ctx.to_device(pols), ctx.to_device(cpols), ctx.to_device(x)
    for batch in (0, 1024, 64):                     # one batch; eight batches; one tile per batch (every shifted read crosses into the halo)
        ctx.set_chelpers_batch_rows(batch)
        q = ctx.to_device(np.full(nrows * 3 + 6, 0xABCD, dtype=np.uint64))
        prog.run(d_pols, d_c, n_const, chal, pub, d_x, xs, zhinv, q, 0, nrows)
        got = ctx.to_host(q)
        assert np.array_equal(got[:nrows * 3], want) and np.all(got[nrows * 3:] == 0xABCD), batch
    # a row range that starts and ends inside tiles and includes the last rows (shifted reads wrap to row 0)
    ctx.set_chelpers_batch_rows(512)
    q2 = ctx.to_device(np.zeros(nrows * 3, dtype=np.uint64))
    r0 = nrows - 1000
    prog.run(d_pols, d_c, n_const, chal, pub, d_x, xs, zhinv, q2, r0, 1000)
    got2 = ctx.to_host(q2)
    assert np.array_equal(got2[3 * r0:], want[3 * r0:]) and not got2[:3 * r0].any()
    prog.close()
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("lin_min", [None, 1])
def test_native_step52ns_matches_oracle(tmp_path, lin_min, monkeypatch):
    """lin_min = 1: the polynomial terms of the chains are summed by the linear kernel (k_chp_linear) instead of generated code."""
    import mi_stark
    if lin_min is not None:
        monkeypatch.setenv("MI_CHELPERS_LIN_MIN", str(lin_min))
    ctx = mi_stark.Context(0)
    nrows = 1 << 13
    ops, args, pols, cpols, n_const, chal, evals, xd, xdw = _case52(41, nrows)
    want = np.zeros(nrows * 3, dtype=np.uint64)
    prog = mi_stark.ChelpersProgram(ctx, ops, args, sections=_synthetic_sections(nrows), n_const=n_const, nrows_ext=nrows, step=mi_stark.MI_CHELPERS_STEP52NS)
    assert prog.build_native(cache_dir=str(tmp_path), chunk_cost=1500)["kernels"] >= (2 if lin_min is None else 1)
    f = ctx.zeros(nrows * 3)
    ctx.set_chelpers_batch_rows(2048)
    for ev in (evals, glo.rand_fe(np.random.default_rng(5), evals.size)):
        glo.chelpers_step52ns(ops, args, pols, cpols, n_const, chal, ev, xd, xdw, want, 0, nrows)
        prog.run52(ctx.to_device(pols), ctx.to_device(cpols), n_const, chal, ev, ctx.to_device(xd), ctx.to_device(xdw), f, 0, nrows)
        assert np.array_equal(ctx.to_host(f), want)
    prog.close()
    ctx.close()


# ------------------------------------------------------------------ the base-domain steps: step2prev / step3prev / step3
REF_BASE = {"step2prev": ("op2prev", "args2prev", 1815, 5828), "step3prev": ("op3prev", "args3prev", 5869, 18927), "step3": ("op3", "args3", 11042, 45647)}


def _base_case(seed, nrows):
    rng = np.random.default_rng(seed)
    secs, out = [(0, 40), (nrows * 40, 9), (nrows * 49, 3)], (nrows * 52, 120)
    ops, args = cp.synthetic_program_base(rng, nrows, secs, out, 7, 5, 4)
    pols = np.zeros(nrows * (52 + 120), dtype=np.uint64)
    pols[:nrows * 52] = glo.rand_fe(rng, nrows * 52, canonical=False)
    return ops, args, pols, glo.rand_fe(rng, nrows * 7), glo.rand_fe(rng, 15), glo.rand_fe(rng, 4), glo.rand_fe(rng, nrows * 2), [(o, w, nrows) for o, w in secs]


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_base_step_programs_match_oracle_on_the_host(seed):
    """Every opcode of the base-domain numbering (results stored into polynomials, read back through the stored temporaries):
    the translated program and the program as the native backend lowers it against the oracle's restatement."""
    import mi_stark
    nrows = 64
    ops, args, pols, cpols, chal, pub, x, sections = _base_case(seed, nrows)
    want, got, got2 = pols.copy(), pols.copy(), pols.copy()
    glo.chelpers_stepbase(ops, args, want, cpols, 7, chal, pub, x, 2, np.arange(nrows))
    prog = mi_stark.ChelpersProgram(None, ops, args, sections=sections, n_const=7, nrows_ext=nrows, step=mi_stark.MI_CHELPERS_STEP3)
    prog.run_base_host(got, cpols, 7, chal, pub, x, 2, np.arange(nrows))
    prog.run_base_host(got2, cpols, 7, chal, pub, x, 2, np.arange(nrows), lowered=True, chunk_cost=1200)
    assert np.array_equal(got, want) and np.array_equal(got2, want) and want[nrows * 52:].any()
    prog.close()


@needs_ref
@pytest.mark.parametrize("step", ["step2prev", "step3prev", "step3"])
def test_base_step_tables_agree_with_the_reference_and_its_programs_match_the_oracle(step):
    """The three base-domain interpreters share one opcode numbering, which the product's / the oracle's / the test generator's tables
    transcribe: argument counts against the reference's source text, the reference's own program consumed exactly, and -- over a
    sparse address space -- every polynomial element the program writes on sampled rows equal between the oracle, the translated
    program and the lowered program; the generated kernels of the program compile."""
    import mi_stark
    opn, argn, nops, nargs = REF_BASE[step]
    src = open("/root/reference/src/starkpil/zkevm/chelpers/zkevm.chelpers.%s.parser.cpp" % step).read()
    body = src[src.index("void ZkevmSteps::%s_parser_first_avx(" % step):]
    body = body[:body.index("void ZkevmSteps::", 10)] if "void ZkevmSteps::" in body[10:] else body
    cases = re.split(r"\n\s*case (\d+):", body)[1:]
    seen = set()
    for num, text in zip(cases[0::2], cases[1::2]):
        o = int(num)
        seen.add(o)
        if o in (69, 91, 97, 99):
            continue
        want = sum(int(k) for k in re.findall(r"i_args \+= (\d+);", text.split("default:")[0]))
        parts = cp.FUSED_BASE.get(o, [o])
        got = 0
        for q in parts:
            if q in cp.OPS_BASE_EXTRA:
                _, dk, _, a, b = cp.OPS_BASE_EXTRA[q]
                got += cp.NARGS_DST[dk] + sum(cp.NARGS[k] for k in (a, b) if k is not None)
            else:
                got += cp.nargs_of(q)
        assert got == want, (step, o, got, want)
    assert set(range(115)) <= seen
    ops, args = cp.parse_reference_tables(open("/root/reference/src/starkpil/zkevm/chelpers/zkevm.chelpers.%s.parser.hpp" % step).read(), opn, argn)
    assert ops.size == nops and args.size == nargs
    dec, used = cp.decode_base(ops, args)
    assert used == args.size
    N, numpols = 1 << 23, 218
    secs = sorted({((a[0] // N) * N, a[1] if k in (cp.POL, cp.POL3) else a[3]) for (_, _, _, _, _, srcs) in dec for k, a in srcs if k in (cp.POL, cp.POL3, cp.POLS, cp.POL3S)})
    rows = [0, 1, N - 2, N - 1, 4242]
    rd, wr, ca = cp.touched_addresses_base(dec, rows, numpols)

    def sparse(n_elems):
        return np.frombuffer(mmap.mmap(-1, n_elems * 8, flags=mmap.MAP_PRIVATE | mmap.MAP_ANONYMOUS | getattr(mmap, "MAP_NORESERVE", 0x4000)), dtype=np.uint64)
    rng = np.random.default_rng(7)
    top = max(max(rd), max(wr)) + 8
    want, got, got2 = sparse(top), sparse(top), sparse(top)
    idx = np.fromiter(rd, dtype=np.int64)
    vals = glo.rand_fe(rng, idx.size)
    for m in (want, got, got2):
        m[idx] = vals
    cpols = sparse(max(ca) + 8)
    cpols[np.fromiter(ca, dtype=np.int64)] = glo.rand_fe(rng, len(ca))
    x = sparse(N)
    x[rows] = glo.rand_fe(rng, len(rows))
    chal, pub = glo.rand_fe(rng, 8 * 3), glo.rand_fe(rng, 64)
    glo.chelpers_stepbase(ops, args, want, cpols, numpols, chal, pub, x, 1, rows)
    sid = {"step2prev": mi_stark.MI_CHELPERS_STEP2PREV, "step3prev": mi_stark.MI_CHELPERS_STEP3PREV, "step3": mi_stark.MI_CHELPERS_STEP3}[step]
    prog = mi_stark.ChelpersProgram(None, ops, args, sections=[(o, s, N) for o, s in secs], n_const=numpols, nrows_ext=N, step=sid)
    prog.run_base_host(got, cpols, numpols, chal, pub, x, 1, rows)
    prog.run_base_host(got2, cpols, numpols, chal, pub, x, 1, rows, lowered=True)
    widx = np.fromiter(wr, dtype=np.int64)
    assert widx.size > 500 and want[widx].all()
    assert np.array_equal(got[widx], want[widx]) and np.array_equal(got2[widx], want[widx])
    with tempfile.TemporaryDirectory() as td:      # one share of a parallel build: the generated kernels compile
        prog.precompile_shard(0, 8, cache_dir=td)
        assert len([f for f in os.listdir(td) if f.endswith(".hsaco")]) >= 1
    prog.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,inplace_max", [(11, None), (12, None), (11, "0"), (12, "64")])
def test_base_step_program_on_gpu_matches_oracle(seed, inplace_max, tmp_path, monkeypatch):
    """The compiled kernels of a base-domain program write the same polynomial elements as the oracle, in one batch and in many;
    the interpreter has no store form and says so."""
    if inplace_max is not None:      # operands through the tile-major copy only / read in place wherever a section allows it
        monkeypatch.setenv("MI_CHELPERS_INPLACE_MAX", inplace_max)
    import mi_stark
    ctx = mi_stark.Context(0)
    nrows = 1 << 12
    ops, args, pols, cpols, chal, pub, x, sections = _base_case(seed, nrows)
    want = pols.copy()
    glo.chelpers_stepbase(ops, args, want, cpols, 7, chal, pub, x, 2, np.arange(nrows))
    prog = mi_stark.ChelpersProgram(ctx, ops, args, sections=sections, n_const=7, nrows_ext=nrows, step=mi_stark.MI_CHELPERS_STEP3PREV)
    d_c, d_x = ctx.to_device(cpols), ctx.to_device(x)
    with pytest.raises(mi_stark.MiStarkError, match="mi_chelpers_build_native first"):
        prog.run_base(ctx.to_device(pols), d_c, 7, chal, pub, d_x, 2, 0, nrows)
    assert prog.build_native(cache_dir=str(tmp_path), chunk_cost=2500)["kernels"] >= 2
    for batch in (0, 512):
        ctx.set_chelpers_batch_rows(batch)
        d_pols = ctx.to_device(pols)
        prog.run_base(d_pols, d_c, 7, chal, pub, d_x, 2, 0, nrows)
        assert np.array_equal(ctx.to_host(d_pols), want), batch
    prog.close()
    ctx.close()


REV6 = np.array([int(format(i, "06b")[::-1], 2) for i in range(64)])


def _tile_major(a, nrows, ncols):
    """[nrows x ncols] row-major -> [nrows / 64][ncols][64], canonical, a row at the bit-reversal of its index inside the tile
    (include/mi_stark.h: mi_chelpers_set_tiled_section)."""
    return (np.asarray(a, dtype=np.uint64).reshape(nrows // 64, 64, ncols) % np.uint64(glo.P)).transpose(0, 2, 1)[:, :, REV6].reshape(-1).copy()


def test_tiled_section_is_declared_before_the_build_and_never_stored_into(tmp_path):
    """mi_chelpers_set_tiled_section: an offset that names no section, a call after the build and a program that stores into the section
    are refused; the kernels of a program with such a section compile; the host executors (row-major) refuse it."""
    import mi_stark
    nrows = 256
    ops, args, pols, cpols, chal, pub, x, sections = _base_case(5, nrows)
    prog = mi_stark.ChelpersProgram(None, ops, args, sections=sections, n_const=7, nrows_ext=nrows, step=mi_stark.MI_CHELPERS_STEP3)
    with pytest.raises(mi_stark.MiStarkError, match="no declared section starts at this offset"):
        prog.set_tiled_section(17)
    prog.set_tiled_section(0)
    prog.set_tiled_section(0)
    with pytest.raises(mi_stark.MiStarkError, match="host executors read row-major"):
        prog.run_base_host(pols.copy(), cpols, 7, chal, pub, x, 2, np.arange(4))
    for shard in (0, 1):
        prog.precompile_shard(shard, 2, cache_dir=str(tmp_path), chunk_cost=2500)
    assert prog.build_native(cache_dir=str(tmp_path), chunk_cost=2500)["cache_hits"] >= 2
    with pytest.raises(mi_stark.MiStarkError, match="comes before mi_chelpers_build_native"):
        prog.set_tiled_section(0)
    prog.close()
    # the output area declared as a read section and tile-major: the program's stores land in it
    out_sec = (nrows * 52, 120, nrows)
    prog = mi_stark.ChelpersProgram(None, ops, args, sections=sections + [out_sec], n_const=7, nrows_ext=nrows, step=mi_stark.MI_CHELPERS_STEP3)
    prog.set_tiled_section(nrows * 52)
    with pytest.raises(mi_stark.MiStarkError, match="stores into the section declared tile-major"):
        prog.build_native(cache_dir=str(tmp_path), chunk_cost=2500)
    prog.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,witness_tiled", [(23, True), (24, False)])
def test_base_step_program_reads_tile_major_constants_in_place(seed, witness_tiled, tmp_path):
    """The resident constant polynomials kept tile-major (host/starks.hpp; mi_chelpers_set_tiled_consts): a base-domain program reads
    them in place -- shifted rows across tile borders and around the end included --, with the witness section tile-major as well or
    row-major, in one batch and in many, from row 0 and from a later multiple of 64."""
    import mi_stark
    ctx = mi_stark.Context(0)
    nrows = 1 << 12
    ops, args, pols, cpols, chal, pub, x, sections = _base_case(seed, nrows)
    want = pols.copy()
    glo.chelpers_stepbase(ops, args, want, cpols, 7, chal, pub, x, 2, np.arange(nrows))
    img = pols.copy()
    if witness_tiled:
        img[:nrows * 40] = _tile_major(pols[:nrows * 40], nrows, 40)
    d_ct = ctx.to_device(_tile_major(cpols, nrows, 7))
    prog = mi_stark.ChelpersProgram(ctx, ops, args, sections=sections, n_const=7, nrows_ext=nrows, step=mi_stark.MI_CHELPERS_STEP3PREV)
    if witness_tiled:
        prog.set_tiled_section(0)
    prog.set_tiled_consts()
    assert prog.build_native(cache_dir=str(tmp_path), chunk_cost=2500)["kernels"] >= 2
    with pytest.raises(mi_stark.MiStarkError, match="comes before mi_chelpers_build_native"):
        prog.set_tiled_consts()
    d_x = ctx.to_device(x)
    for batch in (0, 512):
        ctx.set_chelpers_batch_rows(batch)
        d_pols = ctx.to_device(img)
        prog.run_base(d_pols, d_ct, 7, chal, pub, d_x, 2, 0, nrows)
        assert np.array_equal(ctx.to_host(d_pols)[nrows * 40:], want[nrows * 40:]), batch
    d_pols = ctx.to_device(img)                    # rows [1024, 4096) only
    prog.run_base(d_pols, d_ct, 7, chal, pub, d_x, 2, 1024, nrows - 1024)
    part = pols.copy()
    glo.chelpers_stepbase(ops, args, part, cpols, 7, chal, pub, x, 2, np.arange(1024, nrows))
    assert np.array_equal(ctx.to_host(d_pols)[nrows * 40:], part[nrows * 40:])
    prog.close()
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("seed", [21, 22])
def test_base_step_program_reads_a_tile_major_section_in_place(seed, tmp_path):
    """The witness kept tile-major (host/starks.hpp): mi_tile_major_dev writes [tile][column][64 rows]; a base-domain program compiled
    with that section declared tile-major reads it in place -- shifted rows across tile borders and around the section's end included --
    and writes the oracle's polynomial elements, in one batch and in many, from row 0 and from a later multiple of 64."""
    import mi_stark
    import torch
    ctx = mi_stark.Context(0)
    nrows = 1 << 12
    ops, args, pols, cpols, chal, pub, x, sections = _base_case(seed, nrows)
    want = pols.copy()
    glo.chelpers_stepbase(ops, args, want, cpols, 7, chal, pub, x, 2, np.arange(nrows))
    tiled = pols.copy()
    tiled[:nrows * 40] = _tile_major(pols[:nrows * 40], nrows, 40)
    # the kernel that writes the layout: whole section, and in column chunks at a pitch
    d_rm = ctx.to_device(pols[:nrows * 40])
    d_tm = torch.zeros(nrows * 40, dtype=torch.int64, device="cuda")
    ctx.tile_major(d_tm, 40, 0, d_rm, nrows, 40)
    assert np.array_equal(ctx.to_host(d_tm), tiled[:nrows * 40])
    d_tm2 = torch.zeros(nrows * 40, dtype=torch.int64, device="cuda")
    for c0, w in ((0, 8), (8, 24), (32, 8)):
        chunk = ctx.to_device(np.ascontiguousarray(pols[:nrows * 40].reshape(nrows, 40)[:, c0:c0 + w]))
        ctx.tile_major(d_tm2, 40, c0, chunk, nrows, w)
    assert np.array_equal(ctx.to_host(d_tm2), tiled[:nrows * 40])
    prog = mi_stark.ChelpersProgram(ctx, ops, args, sections=sections, n_const=7, nrows_ext=nrows, step=mi_stark.MI_CHELPERS_STEP3PREV)
    prog.set_tiled_section(0)
    assert prog.build_native(cache_dir=str(tmp_path), chunk_cost=2500)["kernels"] >= 2
    d_c, d_x = ctx.to_device(cpols), ctx.to_device(x)
    for batch in (0, 512):
        ctx.set_chelpers_batch_rows(batch)
        d_pols = ctx.to_device(tiled)
        prog.run_base(d_pols, d_c, 7, chal, pub, d_x, 2, 0, nrows)
        got = ctx.to_host(d_pols)
        assert np.array_equal(got[nrows * 40:], want[nrows * 40:]) and np.array_equal(got[:nrows * 40], tiled[:nrows * 40]), batch
    d_pols = ctx.to_device(tiled)                  # rows [1024, 4096) only
    prog.run_base(d_pols, d_c, 7, chal, pub, d_x, 2, 1024, nrows - 1024)
    part = pols.copy()
    glo.chelpers_stepbase(ops, args, part, cpols, 7, chal, pub, x, 2, np.arange(1024, nrows))
    assert np.array_equal(ctx.to_host(d_pols)[nrows * 40:], part[nrows * 40:])
    with pytest.raises(mi_stark.MiStarkError, match="rows from a multiple of 64"):
        prog.run_base(d_pols, d_c, 7, chal, pub, d_x, 2, 5, 64)
    prog.close()
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("tiled_secs,tiled_consts", [((0,), False), ((0, 1), False), ((0, 1, 2), False), ((0, 1, 2), True), ((), True)])
def test_native_step42ns_reads_tile_major_sections_in_place(tmp_path, tiled_secs, tiled_consts):
    """The extended sections as Starks::genProof keeps them (mi_lde_merkle_dev_tiled): one, two or all of a step42ns program's sections
    tile-major, the others row-major through the per-batch copy, the extended constants tile-major or not; shifted rows across tile
    borders and around the end; batches; a row range."""
    import mi_stark
    ctx = mi_stark.Context(0)
    nrows = 1 << 13
    ops, args, pols, cpols, n_const, chal, pub, x, xs, zhinv = _synthetic_case(33, nrows, passes=3)
    want = np.zeros(nrows * 3, dtype=np.uint64)
    glo.chelpers_step42ns(ops, args, pols, cpols, n_const, chal, pub, x, xs, zhinv, want, 0, nrows)
    secs = _synthetic_sections(nrows)
    img = pols.copy()
    for i in tiled_secs:
        off, w, _ = secs[i]
        img[off:off + nrows * w] = _tile_major(pols[off:off + nrows * w], nrows, w)
    prog = mi_stark.ChelpersProgram(ctx, ops, args, sections=secs, n_const=n_const, nrows_ext=nrows)
    for i in tiled_secs:
        prog.set_tiled_section(secs[i][0])
    if tiled_consts:
        prog.set_tiled_consts()
    assert prog.build_native(cache_dir=str(tmp_path), chunk_cost=7000)["kernels"] >= 3
    d_pols, d_c, d_x = ctx.to_device(img), ctx.to_device(_tile_major(cpols, nrows, n_const) if tiled_consts else cpols), ctx.to_device(x)
    for batch in (0, 1024, 64):
        ctx.set_chelpers_batch_rows(batch)
        q = ctx.to_device(np.full(nrows * 3 + 6, 0xABCD, dtype=np.uint64))
        prog.run(d_pols, d_c, n_const, chal, pub, d_x, xs, zhinv, q, 0, nrows)
        got = ctx.to_host(q)
        assert np.array_equal(got[:nrows * 3], want) and np.all(got[nrows * 3:] == 0xABCD), batch
    ctx.set_chelpers_batch_rows(512)
    q2 = ctx.to_device(np.zeros(nrows * 3, dtype=np.uint64))
    r0 = nrows - 1024
    prog.run(d_pols, d_c, n_const, chal, pub, d_x, xs, zhinv, q2, r0, 1024)
    got2 = ctx.to_host(q2)
    assert np.array_equal(got2[3 * r0:], want[3 * r0:]) and not got2[:3 * r0].any()
    prog.close()
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("lin_min,tiled_secs,tiled_consts", [(None, (0, 1, 2), False), (1, (0, 1, 2), False), (1, (0, 2), False), (1, (1,), False), (1, (0, 1, 2), True), (None, (0,), True)])
def test_native_step52ns_reads_tile_major_sections_in_place(tmp_path, lin_min, tiled_secs, tiled_consts, monkeypatch):
    """... and step52ns, whose polynomial terms the linear kernel sums: out of tile-major sections it takes a lane's own row, out of
    row-major ones it turns a slab through LDS -- both in one program."""
    import mi_stark
    if lin_min is not None:
        monkeypatch.setenv("MI_CHELPERS_LIN_MIN", str(lin_min))
    ctx = mi_stark.Context(0)
    nrows = 1 << 13
    ops, args, pols, cpols, n_const, chal, evals, xd, xdw = _case52(43, nrows)
    want = np.zeros(nrows * 3, dtype=np.uint64)
    glo.chelpers_step52ns(ops, args, pols, cpols, n_const, chal, evals, xd, xdw, want, 0, nrows)
    secs = _synthetic_sections(nrows)
    img = pols.copy()
    for i in tiled_secs:
        off, w, _ = secs[i]
        img[off:off + nrows * w] = _tile_major(pols[off:off + nrows * w], nrows, w)
    prog = mi_stark.ChelpersProgram(ctx, ops, args, sections=secs, n_const=n_const, nrows_ext=nrows, step=mi_stark.MI_CHELPERS_STEP52NS)
    for i in tiled_secs:
        prog.set_tiled_section(secs[i][0])
    if tiled_consts:
        prog.set_tiled_consts()
    prog.build_native(cache_dir=str(tmp_path), chunk_cost=1500)
    f = ctx.zeros(nrows * 3)
    d_c = ctx.to_device(_tile_major(cpols, nrows, n_const) if tiled_consts else cpols)
    for batch in (2048, 0):
        ctx.set_chelpers_batch_rows(batch)
        prog.run52(ctx.to_device(img), d_c, n_const, chal, evals, ctx.to_device(xd), ctx.to_device(xdw), f, 0, nrows)
        assert np.array_equal(ctx.to_host(f), want), batch
    prog.close()
    ctx.close()


def _fit_stats():
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import chelpers_match as cm
    return cm, cm.program_stats(*cm.synthetic(cp.ZKEVM_STEP42NS_FIT))


def test_fitted_synthetic_step42ns_has_the_committed_statistics_of_the_real_program():
    """The synthetic step42ns program that stands in for the reference's on the GPU box (bench_starks.py) against the real program's
    statistics as committed in profiles/r03_chelpers_step42ns_target.json (numbers only): each within the tolerance stated next to the
    fit (5 % for what decides kernel cost and the section mix; loads per row and distinct operands are the stated residual gap).  No GPU,
    no reference needed."""
    import json
    cm, st = _fit_stats()
    target = json.load(open(cm.TARGET))["stats"]
    for k, tol in cp.ZKEVM_STEP42NS_FIT_TOLERANCE.items():
        assert abs(st[k] - target[k]) <= tol * target[k], (k, st[k], target[k])


@needs_ref
def test_committed_target_statistics_are_the_real_programs():
    """... and that file is what tools/chelpers_match.py measures on the reference's own tables today."""
    import json
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import chelpers_match as cm
    real = cm.program_stats(*cp.parse_reference_tables(open(REF_HPP).read()))
    target = json.load(open(cm.TARGET))["stats"]
    for k in cm.KEYS:
        assert abs(real[k] - target[k]) <= 1e-9 * max(1.0, abs(target[k])), (k, real[k], target[k])
