// constant_pols_starks.hpp (standalone stand-in) -- the view of the constant polynomials the reference hands to the Steps
// (constant_pols_starks.hpp:8-29): numPols columns, element (pol, row) at address[pol + row * numPols].
#ifndef CONSTANT_POLS_STARKS_HPP
#define CONSTANT_POLS_STARKS_HPP
#include <cstdint>
#include "goldilocks_base_field.hpp"
class ConstantPolsStarks
{
    void *_pAddress;
    uint64_t _degree, _numPols;

public:
    ConstantPolsStarks(void *pAddress, uint64_t degree, uint64_t numPols) : _pAddress(pAddress), _degree(degree), _numPols(numPols) {}
    uint64_t numPols(void) { return _numPols; }
    void *address(void) { return _pAddress; }
    uint64_t degree(void) { return _degree; }
    uint64_t size(void) { return _degree * _numPols * sizeof(Goldilocks::Element); }
    Goldilocks::Element &getElement(uint64_t pol, uint64_t evaluation) { return ((Goldilocks::Element *)_pAddress)[pol + evaluation * _numPols]; }
};
#endif
