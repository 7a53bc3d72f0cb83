// zkevm_steps_device.cpp -- the ONE translation unit a maintainer compiles in place of the reference's five
//     src/starkpil/zkevm/chelpers/zkevm.chelpers.step{2prev,3prev,3,42ns,52ns}.parser.cpp
// (the AVX2 / AVX-512 interpreters over the generated tables: 10 001 lines, 295 call shapes of the x86-typed Goldilocks::*_avx helper
// families).  It keeps what those files contribute to the link -- the generated TABLES (each *.parser.hpp is included exactly once,
// as there) and the definitions of ZkevmSteps' batched entry points declared in zkevmSteps.hpp:16-56 -- and hands the tables to
// libmi_stark, which translates them once per proving key, compiles them to gfx950 kernels and runs them over the device image of
// the polynomial area that Starks::genProof (host/starks.hpp) keeps.  No table text lives in this repository: the headers are the
// reference's own, found on the include path.
//
// The per-row forms (step*_first / _i / _last: zkevm.chelpers.step{2,3prev,3,52ns}.cpp and the step42ns blob) are untouched: they are
// scalar C++ over Goldilocks::add / mul ..., compile against the Level-0 headers, and stay in the build.
#include "zkevmSteps.hpp"
#include "chelpers_steps.hpp"

// every *.parser.hpp defines NOPS_ / NARGS_ (and NTEMP1_ / NTEMP3_, the interpreters' temporary counts, unused here) for its own table
#include "zkevm.chelpers.step2prev.parser.hpp"
MI_DEFINE_PARSER_STEP(ZkevmSteps, step2prev, _avx, MI_CHELPERS_STEP2PREV, op2prev, NOPS_, args2prev, NARGS_)
#undef NOPS_
#undef NARGS_
#undef NTEMP1_
#undef NTEMP3_
#include "zkevm.chelpers.step3prev.parser.hpp"
MI_DEFINE_PARSER_STEP(ZkevmSteps, step3prev, _avx, MI_CHELPERS_STEP3PREV, op3prev, NOPS_, args3prev, NARGS_)
#undef NOPS_
#undef NARGS_
#undef NTEMP1_
#undef NTEMP3_
#include "zkevm.chelpers.step3.parser.hpp"
MI_DEFINE_PARSER_STEP(ZkevmSteps, step3, _avx, MI_CHELPERS_STEP3, op3, NOPS_, args3, NARGS_)
#undef NOPS_
#undef NARGS_
#undef NTEMP1_
#undef NTEMP3_
#include "zkevm.chelpers.step42ns.parser.hpp"
MI_DEFINE_PARSER_STEP(ZkevmSteps, step42ns, _avx, MI_CHELPERS_STEP42NS, op42, NOPS_, args42, NARGS_)
#undef NOPS_
#undef NARGS_
#undef NTEMP1_
#undef NTEMP3_
#include "zkevm.chelpers.step52ns.parser.hpp"
MI_DEFINE_PARSER_STEP(ZkevmSteps, step52ns, _avx, MI_CHELPERS_STEP52NS, op52, NOPS_, args52, NARGS_)
#undef NOPS_
#undef NARGS_

// the other flavours ZkevmSteps declares (scalar, "_jump", AVX-512): one program each, whatever the host's vector width
MI_FORWARD_PARSER_STEP(ZkevmSteps, step3, , _avx)
MI_FORWARD_PARSER_STEP(ZkevmSteps, step3, _avx_jump, _avx)
MI_FORWARD_PARSER_STEP(ZkevmSteps, step42ns, , _avx)
MI_FORWARD_PARSER_STEP(ZkevmSteps, step42ns, _avx_jump, _avx)
MI_FORWARD_PARSER_STEP(ZkevmSteps, step52ns, , _avx)
#ifdef __AVX512__
MI_FORWARD_PARSER_STEP(ZkevmSteps, step2prev, _avx512, _avx)
MI_FORWARD_PARSER_STEP(ZkevmSteps, step3prev, _avx512, _avx)
MI_FORWARD_PARSER_STEP(ZkevmSteps, step3, _avx512, _avx)
MI_FORWARD_PARSER_STEP(ZkevmSteps, step42ns, _avx512, _avx)
MI_FORWARD_PARSER_STEP(ZkevmSteps, step52ns, _avx512, _avx)
#endif
