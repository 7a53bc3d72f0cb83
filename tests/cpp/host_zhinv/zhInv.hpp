// zhInv.hpp for tests that must not touch a GPU (tests/test_steps_tracer.py): the table of host/standalone/zhInv.hpp computed with the
// host field class instead of mi_zhinv.  ZHInv[i] = 1 / (shift^(2^nBits) * w(extendBits)^i - 1)   (zhInv.cpp:7-31)
#ifndef ZHINV
#define ZHINV
#include <vector>
#include "goldilocks_base_field.hpp"
class ZhInv
{
    std::vector<Goldilocks::Element> table_;

public:
    ZhInv() {}
    ZhInv(uint64_t nBits, uint64_t nBitsExt)
    {
        const uint64_t ext = nBitsExt - nBits;
        Goldilocks::Element sn = Goldilocks::shift(), w = Goldilocks::one();
        for (uint64_t i = 0; i < nBits; i++) sn = Goldilocks::square(sn);
        for (uint64_t i = 0; i < (1ULL << ext); i++) {
            table_.push_back(Goldilocks::inv(Goldilocks::sub(Goldilocks::mul(sn, w), Goldilocks::one())));
            w = Goldilocks::mul(w, Goldilocks::w(ext));
        }
    }
    Goldilocks::Element zhInv(int64_t i) { return table_[i % table_.size()]; }
};
#endif
