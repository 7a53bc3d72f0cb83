// steps.hpp (standalone stand-in) -- the plug-in interface Starks::genProof evaluates its constraint expressions through
// (src/starkpil/steps.hpp:4-59): StepsParams, what a step may read and write, and class Steps, one virtual per step and flavour.
// Field and method names are the reference's (generated chelpers code is written against them); a maintainer's build uses the
// reference's own steps.hpp -- host/starks.hpp needs nothing beyond it.
#ifndef STEPS_HPP
#define STEPS_HPP
#include <cstdint>
#include "goldilocks_base_field.hpp"
#include "polinomial.hpp"
#include "constant_pols_starks.hpp"
#include "zhInv.hpp"

struct StepsParams
{
    Goldilocks::Element *pols;         // the polynomial area (pAddress): every section at its StarkInfo::mapOffsets
    ConstantPolsStarks *pConstPols;    // constant polynomials over the N base-domain rows
    ConstantPolsStarks *pConstPols2ns; // ... over the extended domain
    Polinomial &challenges;
    Polinomial &x_n;
    Polinomial &x_2ns;
    ZhInv &zi;
    Polinomial &evals;
    Polinomial &xDivXSubXi;
    Polinomial &xDivXSubWXi;
    Goldilocks::Element *publicInputs;
    Goldilocks::Element *q_2ns;
    Goldilocks::Element *f_2ns;
};

// per-row forms are pure (a Steps class is its generated code); the batched "parser" forms default to nothing, as in the reference
#define MI_STEP_ROW(s) virtual void s##_first(StepsParams &params, uint64_t i) = 0; \
                       virtual void s##_i(StepsParams &params, uint64_t i) = 0;     \
                       virtual void s##_last(StepsParams &params, uint64_t i) = 0;
#define MI_STEP_BATCH(s, flavour) virtual void s##_parser_first##flavour(StepsParams &params, uint64_t nrows, uint64_t nrowsBatch) {}
class Steps
{
public:
    virtual ~Steps() {}
    MI_STEP_ROW(step2prev) MI_STEP_BATCH(step2prev, _avx) MI_STEP_BATCH(step2prev, _avx512)
    MI_STEP_ROW(step3prev) MI_STEP_BATCH(step3prev, _avx) MI_STEP_BATCH(step3prev, _avx512)
    MI_STEP_ROW(step3) MI_STEP_BATCH(step3, ) MI_STEP_BATCH(step3, _avx) MI_STEP_BATCH(step3, _avx_jump) MI_STEP_BATCH(step3, _avx512)
    MI_STEP_ROW(step42ns) MI_STEP_BATCH(step42ns, ) MI_STEP_BATCH(step42ns, _avx) MI_STEP_BATCH(step42ns, _avx_jump) MI_STEP_BATCH(step42ns, _avx512)
    MI_STEP_ROW(step52ns) MI_STEP_BATCH(step52ns, ) MI_STEP_BATCH(step52ns, _avx) MI_STEP_BATCH(step52ns, _avx512)
};
#undef MI_STEP_ROW
#undef MI_STEP_BATCH
#endif
