// zhInv.hpp (standalone stand-in) -- 1 / Z_H over the extended domain: 2^(nBitsExt - nBits) values, periodic (zhInv.hpp:13-25,
// zhInv.cpp:7-31).  The table comes from the library (mi_zhinv); the reference's own zhInv.cpp compiles against Level 0 as well.
#ifndef ZHINV
#define ZHINV
#include <vector>
#include "goldilocks_base_field.hpp"
#include "mi_runtime.hpp"
class ZhInv
{
    std::vector<Goldilocks::Element> table_;

public:
    ZhInv() {}
    ZhInv(uint64_t nBits, uint64_t nBitsExt)
    {
        if (nBits == 0 || nBitsExt == 0) return;
        table_.resize(1ULL << (nBitsExt - nBits));
        mi::check(mi_zhinv(mi::ctx(), (uint64_t *)table_.data(), (unsigned)nBits, (unsigned)nBitsExt), "ZhInv::ZhInv");
    }
    Goldilocks::Element zhInv(int64_t i) { return table_[i % table_.size()]; }
};
#endif
