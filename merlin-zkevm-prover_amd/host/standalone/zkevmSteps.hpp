// zkevmSteps.hpp (standalone stand-in) -- class ZkevmSteps as src/starkpil/zkevm/chelpers/zkevmSteps.hpp:10-57 declares it: the
// per-row forms (generated C++ in zkevm.chelpers.step*.cpp) and the batched forms (interpreters over the generated tables in
// zkevm.chelpers.step*.parser.cpp).  host/zkevm_steps_device.hpp DEFINES the batched forms on top of libmi_stark; the per-row
// forms stay whatever the build links (the reference's generated files, or the stubs of MI_ZKEVM_STEPS_ROW_STUBS for a build
// that has none).
#ifndef STARKS_STEPS_HPP
#define STARKS_STEPS_HPP
#include "starks.hpp"

#define MI_STEP_ROW(s) void s##_first(StepsParams &params, uint64_t i); void s##_i(StepsParams &params, uint64_t i); void s##_last(StepsParams &params, uint64_t i);
#define MI_STEP_BATCH(s, flavour) void s##_parser_first##flavour(StepsParams &params, uint64_t nrows, uint64_t nrowsBatch);
class ZkevmSteps : public Steps
{
public:
    MI_STEP_ROW(step2prev) MI_STEP_BATCH(step2prev, _avx)
    MI_STEP_ROW(step3prev) MI_STEP_BATCH(step3prev, _avx)
    MI_STEP_ROW(step3) MI_STEP_BATCH(step3, ) MI_STEP_BATCH(step3, _avx) MI_STEP_BATCH(step3, _avx_jump)
    MI_STEP_ROW(step42ns) MI_STEP_BATCH(step42ns, ) MI_STEP_BATCH(step42ns, _avx) MI_STEP_BATCH(step42ns, _avx_jump)
    MI_STEP_ROW(step52ns) MI_STEP_BATCH(step52ns, ) MI_STEP_BATCH(step52ns, _avx)
#ifdef __AVX512__
    MI_STEP_BATCH(step2prev, _avx512) MI_STEP_BATCH(step3prev, _avx512) MI_STEP_BATCH(step3, _avx512) MI_STEP_BATCH(step42ns, _avx512) MI_STEP_BATCH(step52ns, _avx512)
#endif
};
#undef MI_STEP_ROW
#undef MI_STEP_BATCH
#endif
