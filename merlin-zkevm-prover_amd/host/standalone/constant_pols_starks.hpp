// constant_pols_starks.hpp (standalone stand-in) -- the view of the constant polynomials the reference hands to the Steps
// (interface of constant_pols_starks.hpp:8-29): a row-major matrix of `cols` columns, polynomial p at row r is base[r * cols + p].
#ifndef CONSTANT_POLS_STARKS_HPP
#define CONSTANT_POLS_STARKS_HPP
#include <cstdint>
#include "goldilocks_base_field.hpp"
class ConstantPolsStarks
{
public:
    ConstantPolsStarks(void *matrix, uint64_t rows, uint64_t cols) : base((Goldilocks::Element *)matrix), rows_(rows), cols_(cols) {}
    // the reference's accessors, by its names
    Goldilocks::Element &getElement(uint64_t pol, uint64_t evaluation) { return base[evaluation * cols_ + pol]; }
    void *address() { return base; }
    uint64_t numPols() { return cols_; }
    uint64_t degree() { return rows_; }
    uint64_t size() { return rows_ * cols_ * sizeof(Goldilocks::Element); }

private:
    Goldilocks::Element *base;
    uint64_t rows_, cols_;
};
#endif
