"""TEST INFRASTRUCTURE: constraint programs given as opcode tables (tests/chelpers_programs.py, tests/ministark.py) written out as
generated PER-ROW C++ in the style of the reference's recursive STARKs (recursive1.chelpers.step3.cpp etc.: one straight-line function
per step, a fresh named temporary per operation, operands spelled params.pols[off + i*stride], params.pConstPols->getElement(c,i),
(Goldilocks3::Element &)*params.challenges[k], Goldilocks::fromU64(v), ... and `_first` carrying the ((i + s)%n) wraps).

The reference's own generated files cannot travel to the GPU box; these stand in for them there, so that host/steps_tracer.hpp has a
Steps class of that kind to record.  Nothing here is derived from the reference's text: the operand spellings above are its public
StepsParams interface (steps.hpp:4-18)."""
import chelpers_programs as cp

# step52ns (zkevm.chelpers.step52ns.parser.cpp: an accumulator machine over tmp / tmp1 / tmp2 = extension slots 0 / 1 / 2)
EVAL, XD, XDW = 100, 101, 102
ROLE52 = {
    0: ("mul", 0, (cp.POL, None), (cp.CHAL, 5)), 1: ("mul", 0, (cp.T3, 0), (cp.CHAL, 5)), 2: ("mul", 0, (cp.T3, 0), (cp.CHAL, 6)),
    3: ("mul", 1, (cp.T3, 0), (cp.CHAL, 5)), 4: ("mul", 0, (cp.T3, 2), (cp.CHAL, 6)), 5: ("mul", 0, (cp.T3, 0), (XD, None)),
    6: ("mul", 0, (cp.T3, 0), (XDW, None)), 7: ("add", 0, (cp.T3, 0), (cp.T3, 2)), 8: ("add", 0, (cp.T3, 1), (cp.T3, 0)),
    9: ("add", 0, (cp.T3, 0), (cp.POL3, None)), 10: ("add", 0, (cp.T3, 0), (cp.POL, None)), 11: ("sub", 2, (cp.POL, None), (EVAL, None)),
    12: ("sub", 2, (cp.POL3, None), (EVAL, None)), 13: ("sub", 2, (cp.CONST, None), (EVAL, None)), 14: ("sub", 0, (cp.CONST, 5), (EVAL, 0)),
    15: ("storef", None, (cp.T3, 0), None),
}


def micro42(ops, args):
    out = []
    for (o, d, slot, srcs) in cp.decode(ops, args)[0]:
        if o == 69:
            out.append(("storeq", None, srcs[0], None))
        else:
            out.append((cp._cls42(o), ("t", d, slot), srcs[0], srcs[1] if len(srcs) > 1 else None))
    return out


def micro_base(ops, args):
    out = []
    for (o, c, dk, dd, dargs, srcs) in cp.decode_base(ops, args)[0]:
        dst = ("t", dk, dargs[0]) if dk in (cp.T1, cp.T3) else ("p", dk, dd, dargs)
        out.append((c, dst, srcs[0], srcs[1] if len(srcs) > 1 else None))
    return out


def micro52(ops, args):
    out, ia = [], 0
    for op in ops:
        for o in cp.FUSED52.get(int(op), [int(op)]):
            c, dst, a, b = ROLE52[o]
            srcs = []
            for s in (a, b):
                if s is None:
                    continue
                k, fixed = s
                if fixed is not None:
                    srcs.append((k, [fixed]))
                elif k in (cp.POL, cp.POL3):
                    srcs.append((k, [int(args[ia]), int(args[ia + 1])])); ia += 2
                elif k in (cp.CONST, EVAL):
                    srcs.append((k, [int(args[ia])])); ia += 1
                else:
                    srcs.append((k, []))
            if c == "storef":
                out.append(("storef", None, srcs[0], None))
            else:
                out.append((c, ("t", cp.T3, dst), srcs[0], srcs[1] if len(srcs) > 1 else None))
    assert ia == len(args)
    return out


def _function(name, cls_name, micro, base):
    """One `void Class::name(StepsParams &params, uint64_t i)` body."""
    cpol, x = ("params.pConstPols", "params.x_n") if base else ("params.pConstPols2ns", "params.x_2ns")
    lines, cur, n = [], {}, [0]

    def is3(k):
        return k in (cp.T3, cp.CHAL, cp.POL3, cp.POL3S, EVAL, XD, XDW)

    def pol(a, three):
        e = "params.pols[%d + i*%d]" % (a[0], a[1]) if len(a) == 2 else "params.pols[%d + ((i + %d)%%%d)*%d]" % (a[0], a[1], a[2], a[3])
        return "(Goldilocks3::Element &)(%s)" % e if three else e

    def src(s):
        k, a = s
        if k in (cp.T1, cp.T3):
            return cur[(k, a[0])]
        if k in (cp.POL, cp.POLS):
            return pol(a, False)
        if k in (cp.POL3, cp.POL3S):
            return pol(a, True)
        if k == cp.NUM:
            return "Goldilocks::fromU64(%dULL)" % a[0]
        if k == cp.CONST:
            return "%s->getElement(%d,i)" % (cpol, a[0])
        if k == cp.CONSTS:
            return "%s->getElement(%d,(i + %d)%%%d)" % (cpol, a[0], a[1], a[2])
        if k == cp.CHAL:
            return "(Goldilocks3::Element &)*params.challenges[%d]" % a[0]
        if k == cp.PUB:
            return "params.publicInputs[%d]" % a[0]
        if k == cp.X:
            return "(Goldilocks::Element &)*%s[i]" % x
        if k == EVAL:
            return "(Goldilocks3::Element &)*params.evals[%d]" % a[0]
        if k == XD:
            return "(Goldilocks3::Element &)*params.xDivXSubXi[i]"
        if k == XDW:
            return "(Goldilocks3::Element &)*params.xDivXSubWXi[i]"
        raise ValueError(k)

    for (c, dst, a, b) in micro:
        if c == "storeq":
            lines.append("     Goldilocks3::mul((Goldilocks3::Element &)(params.q_2ns[i * 3]), params.zi.zhInv(i), %s);" % src(a))
            continue
        if c == "storef":
            lines.append("     Goldilocks3::copy((Goldilocks3::Element &)(params.f_2ns[i * 3]), %s);" % src(a))
            continue
        sa, sb = src(a), (src(b) if b is not None else None)
        if dst[0] == "t":
            three = dst[1] == cp.T3
            var = "tmp_%d" % n[0]
            n[0] += 1
            lines.append("     %s %s;" % ("Goldilocks3::Element" if three else "Goldilocks::Element", var))
            d = var
        else:
            three = dst[2] == 3
            d = pol(dst[3], three)
        ns = "Goldilocks3" if three else "Goldilocks"
        lines.append("     %s::%s(%s, %s);" % (ns, c, d, sa if sb is None else sa + ", " + sb))
        if dst[0] == "t":
            cur[(dst[1], dst[2])] = var
    return "void %s::%s(StepsParams &params, uint64_t i) {\n%s\n}\n" % (cls_name, name, "\n".join(lines))


def steps_source(cls_name, programs, header=None):
    """programs: {"step2prev" | "step3prev" | "step3" | "step42ns" | "step52ns": (ops, args)} (a missing step computes nothing).
    -> the C++ source defining cls_name::step*_first (the forms starks.cpp:86,168,206,254,384 call) with empty _i / _last."""
    out = []
    if header:
        out.append(header)
    for step in ("step2prev", "step3prev", "step3", "step42ns", "step52ns"):
        ops, args = programs.get(step, ([], []))
        if step == "step42ns":
            micro = micro42(ops, args)
        elif step == "step52ns":
            micro = micro52(ops, args)
        else:
            micro = micro_base(ops, args)
        out.append(_function(step + "_first", cls_name, micro, base=step in ("step2prev", "step3prev", "step3")))
        out.append("void %s::%s_i(StepsParams &, uint64_t) {}\nvoid %s::%s_last(StepsParams &, uint64_t) {}\n" % (cls_name, step, cls_name, step))
    return "\n".join(out)


GEN_HEADER = """#ifndef GEN_STEPS_HPP
#define GEN_STEPS_HPP
#include "goldilocks_cubic_extension.hpp"
#include "zhInv.hpp"
#include "polinomial.hpp"
#include "constant_pols_starks.hpp"
#include "steps.hpp"
class GenSteps : public Steps
{
public:
#define ROWS(s) void s##_first(StepsParams &params, uint64_t i) override; void s##_i(StepsParams &params, uint64_t i) override; void s##_last(StepsParams &params, uint64_t i) override;
    ROWS(step2prev) ROWS(step3prev) ROWS(step3) ROWS(step42ns) ROWS(step52ns)
#undef ROWS
};
#endif
"""
# for a shared library a host program loads (libmi_starks.so: mis_load_steps)
GEN_FACTORY = 'extern "C" Steps *mi_make_steps() { return new GenSteps(); }\n'
