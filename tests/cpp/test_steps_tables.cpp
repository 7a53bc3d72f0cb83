// test_steps_tables.cpp -- only where the reference tree is present (tests/test_steps_tracer.py): the zkEVM's five generated TABLES, from
// the reference's own *.parser.hpp on the include path (each included once, as in host/zkevm_steps_device.cpp), handed to
// tests/cpp/test_steps_tracer.cpp, which runs the table programs beside the per-row code of the SAME constraint system
// (zkevm.chelpers.step{2,3prev,3,52ns}.cpp): two products of one generator, so what the library's table decoder makes of the opcodes
// must compute what the reference's compiled per-row C++ computes.  Also defines the members of ZkevmSteps whose files are absent or
// excluded (the step42ns blob, the *.parser.cpp interpreters), as empty functions, for the link.
#include <cstdint>
#include "goldilocks_cubic_extension.hpp"
#include "zhInv.hpp"
#include "polinomial.hpp"
#include "constant_pols_starks.hpp"
#include "steps.hpp"
#include "zkevmSteps.hpp"

#include "zkevm.chelpers.step2prev.parser.hpp"
static const uint64_t n_op2prev = NOPS_, n_args2prev = NARGS_;
#undef NOPS_
#undef NARGS_
#undef NTEMP1_
#undef NTEMP3_
#include "zkevm.chelpers.step3prev.parser.hpp"
static const uint64_t n_op3prev = NOPS_, n_args3prev = NARGS_;
#undef NOPS_
#undef NARGS_
#undef NTEMP1_
#undef NTEMP3_
#include "zkevm.chelpers.step3.parser.hpp"
static const uint64_t n_op3 = NOPS_, n_args3 = NARGS_;
#undef NOPS_
#undef NARGS_
#undef NTEMP1_
#undef NTEMP3_
#include "zkevm.chelpers.step42ns.parser.hpp"
static const uint64_t n_op42 = NOPS_, n_args42 = NARGS_;
#undef NOPS_
#undef NARGS_
#undef NTEMP1_
#undef NTEMP3_
#include "zkevm.chelpers.step52ns.parser.hpp"
static const uint64_t n_op52 = NOPS_, n_args52 = NARGS_;

extern "C" int mi_test_tables(int which, const uint64_t **ops, uint64_t *nops, const uint64_t **args, uint64_t *nargs)
{
    switch (which) {
    case 0: *ops = op2prev; *nops = n_op2prev; *args = args2prev; *nargs = n_args2prev; return 0;
    case 1: *ops = op3prev; *nops = n_op3prev; *args = args3prev; *nargs = n_args3prev; return 0;
    case 2: *ops = op3; *nops = n_op3; *args = args3; *nargs = n_args3; return 0;
    case 3: *ops = op42; *nops = n_op42; *args = args42; *nargs = n_args42; return 0;
    case 4: *ops = op52; *nops = n_op52; *args = args52; *nargs = n_args52; return 0;
    }
    return -1;
}

void ZkevmSteps::step42ns_first(StepsParams &, uint64_t) {}
void ZkevmSteps::step42ns_i(StepsParams &, uint64_t) {}
void ZkevmSteps::step42ns_last(StepsParams &, uint64_t) {}
#define P(s, f) void ZkevmSteps::s##_parser_first##f(StepsParams &, uint64_t, uint64_t) {}
P(step2prev, _avx) P(step3prev, _avx) P(step3, ) P(step3, _avx) P(step3, _avx_jump) P(step42ns, ) P(step42ns, _avx) P(step42ns, _avx_jump)
P(step52ns, ) P(step52ns, _avx)
#ifdef __AVX512__
P(step2prev, _avx512) P(step3prev, _avx512) P(step3, _avx512) P(step42ns, _avx512) P(step52ns, _avx512)
#endif
