// ntt_goldilocks.hpp -- NTT_Goldilocks with the upstream call shapes seen in the reference:
//   NTT_Goldilocks ntt(N); NTT_Goldilocks(nX, 1)                          (starks.hpp:81-82, friProve.cpp:100)
//   ntt.extendPol(out, in, NExtended, N, ncols, buffer)                   (starks.cpp:52,133,214)
//   nttExtended.INTT(dst, src, NExtended, 3, NULL, 2, 1)                  (starks.cpp:261)
//   nttExtended.NTT(dst, src, NExtended, 6)                               (starks.cpp:284)
//   ntt.INTT(LEv, LEv, N, 3)  in place                                    (starks.cpp:325-326)
// Host pointers, natural order in and out.  `buffer`, `nphase`, `nblock` are CPU blocking hints: accepted and
// ignored (the device plans its own passes; twiddle tables are cached per size inside the context).
#ifndef NTT_GOLDILOCKS
#define NTT_GOLDILOCKS
#include "goldilocks_base_field.hpp"
#include "mi_runtime.hpp"

class NTT_Goldilocks
{
    uint64_t maxDomain;

public:
    NTT_Goldilocks(uint64_t maxDomainSize, uint32_t /*nThreads*/ = 0, int /*extension*/ = 1) : maxDomain(maxDomainSize) {}
    void NTT(Goldilocks::Element *dst, Goldilocks::Element *src, uint64_t size, uint64_t ncols = 1, Goldilocks::Element * /*buffer*/ = NULL,
             uint64_t /*nphase*/ = 3, uint64_t /*nblock*/ = 1, bool inverse = false, bool extend = false)
    {
        // extend = true is the private mode extendPol uses between its two transforms upstream; no caller in src/starkpil
        // passes it, and silently ignoring it would return a differently scaled result
        if (extend) { std::fprintf(stderr, "NTT_Goldilocks::NTT: extend=true is not supported, call extendPol\n"); std::exit(-1); }
        mi::check(mi_ntt(mi::ctx(), (uint64_t *)dst, (const uint64_t *)src, size, ncols, inverse ? 1 : 0), "NTT_Goldilocks::NTT");
    }
    void INTT(Goldilocks::Element *dst, Goldilocks::Element *src, uint64_t size, uint64_t ncols = 1, Goldilocks::Element *buffer = NULL,
              uint64_t nphase = 3, uint64_t nblock = 1, bool extend = false)
    {
        NTT(dst, src, size, ncols, buffer, nphase, nblock, true, extend);
    }
    void extendPol(Goldilocks::Element *output, Goldilocks::Element *input, uint64_t N_Extended, uint64_t N, uint64_t ncols,
                   Goldilocks::Element * /*buffer*/ = NULL, uint64_t /*nphase*/ = 3, uint64_t /*nblock*/ = 1)
    {
        mi::check(mi_lde(mi::ctx(), (uint64_t *)output, (const uint64_t *)input, N_Extended, N, ncols), "NTT_Goldilocks::extendPol");
    }
};
#endif
