// batch_helpers_test_only.hpp -- TEST INFRASTRUCTURE, never on the product's include path.
//
// The `_batch` helper family of the absent src/goldilocks submodule, in exactly the 17 names and the call shapes that the reference's
// SCALAR interpreter of the step42ns table uses (ZkevmSteps::step42ns_parser_first, zkevm.chelpers.step42ns.parser.cpp:762-1441), so that
// that function -- the only reference code that gives every one of the 84 primitive step42ns opcodes a meaning without AVX intrinsics --
// compiles UNCHANGED and runs beside the oracle and the product's table decoder (tests/test_reference_scalar_interpreter.py).
//
// What this pins and what it does not: each helper below is MY restatement of what its name says (a digit is an
// operand's dimension, 1 = base field, 3 = cubic extension; `c` = that operand is one broadcast constant; a batch is AVX_SIZE_ = 4
// consecutive rows; a stride or an offsets[4] array says where lane j of an operand lives), over the Level-0 stand-in arithmetic of
// host/goldilocks_base_field.hpp.  So a green test pins, for all 84 cases, the reference function's OPERAND ADDRESSING (which argument is a
// temporary, a polynomial column, a shifted row, a constant polynomial, a number, a challenge, a public input; strides and wrap-arounds)
// and its CHOICE of helper per case -- not the field arithmetic, which stays pinned by the golden proofs.
//
// Used as:   #define Goldilocks GoldilocksB   /   #define Goldilocks3 Goldilocks3B   around the function's text.
#ifndef BATCH_HELPERS_TEST_ONLY_HPP
#define BATCH_HELPERS_TEST_ONLY_HPP
#include <cstdint>
#include "goldilocks_base_field.hpp"
#include "goldilocks_cubic_extension.hpp"

#define MI_TEST_BATCH 4

namespace batch_test {
typedef Goldilocks::Element E;
// where lane j of an operand is: a stride in elements, or explicit offsets
struct Str { uint64_t s; const E *at(const E *p, int j) const { return p + j * s; } };
struct Off { const uint64_t *o; const E *at(const E *p, int j) const { return p + o[j]; } };
enum Op { ADD, SUB, MUL };
inline E op1(Op o, const E &a, const E &b) { return o == ADD ? Goldilocks::add(a, b) : o == SUB ? Goldilocks::sub(a, b) : Goldilocks::mul(a, b); }
inline void op3(Op o, E *r, const E *a, const E *b) // ext (op) ext
{
    Goldilocks3::Element t, x = {a[0], a[1], a[2]}, y = {b[0], b[1], b[2]};
    if (o == ADD) Goldilocks3::add(t, x, y); else if (o == SUB) Goldilocks3::sub(t, x, y); else Goldilocks3::mul(t, x, y);
    r[0] = t[0]; r[1] = t[1]; r[2] = t[2];
}
inline void op13(Op o, E *r, const E &a, const E *b) // base (op) ext, the base operand standing for (a, 0, 0)
{
    Goldilocks3::Element t, y = {b[0], b[1], b[2]};
    if (o == ADD) Goldilocks3::add(t, a, y); else if (o == SUB) Goldilocks3::sub(t, a, y); else Goldilocks3::mul(t, a, y);
    r[0] = t[0]; r[1] = t[1]; r[2] = t[2];
}
inline void op31(Op o, E *r, const E *a, const E &b) // ext (op) base
{
    Goldilocks3::Element t, x = {a[0], a[1], a[2]};
    if (o == ADD) Goldilocks3::add(t, x, b); else if (o == SUB) Goldilocks3::sub(t, x, b); else Goldilocks3::mul(t, x, b);
    r[0] = t[0]; r[1] = t[1]; r[2] = t[2];
}
template <class LA, class LB> inline void b11(Op o, E *r, const E *a, LA la, const E *b, LB lb)
{
    E t[MI_TEST_BATCH];
    for (int j = 0; j < MI_TEST_BATCH; j++) t[j] = op1(o, *la.at(a, j), *lb.at(b, j));
    for (int j = 0; j < MI_TEST_BATCH; j++) r[j] = t[j];
}
template <class LA> inline void b1c(Op o, E *r, const E *a, LA la, const E &c) // base (op) constant
{
    E t[MI_TEST_BATCH];
    for (int j = 0; j < MI_TEST_BATCH; j++) t[j] = op1(o, *la.at(a, j), c);
    for (int j = 0; j < MI_TEST_BATCH; j++) r[j] = t[j];
}
template <class LB> inline void bc1(Op o, E *r, const E &c, const E *b, LB lb) // constant (op) base
{
    E t[MI_TEST_BATCH];
    for (int j = 0; j < MI_TEST_BATCH; j++) t[j] = op1(o, c, *lb.at(b, j));
    for (int j = 0; j < MI_TEST_BATCH; j++) r[j] = t[j];
}
template <class LA, class LB> inline void b33(Op o, E *r, const E *a, LA la, const E *b, LB lb)
{
    E t[MI_TEST_BATCH][3];
    for (int j = 0; j < MI_TEST_BATCH; j++) op3(o, t[j], la.at(a, j), lb.at(b, j));
    for (int j = 0; j < MI_TEST_BATCH; j++) for (int d = 0; d < 3; d++) r[3 * j + d] = t[j][d];
}
template <class LA, class LB> inline void b13(Op o, E *r, const E *a, LA la, const E *b, LB lb)
{
    E t[MI_TEST_BATCH][3];
    for (int j = 0; j < MI_TEST_BATCH; j++) op13(o, t[j], *la.at(a, j), lb.at(b, j));
    for (int j = 0; j < MI_TEST_BATCH; j++) for (int d = 0; d < 3; d++) r[3 * j + d] = t[j][d];
}
template <class LA> inline void b31c(Op o, E *r, const E *a, LA la, const E &c)
{
    E t[MI_TEST_BATCH][3];
    for (int j = 0; j < MI_TEST_BATCH; j++) op31(o, t[j], la.at(a, j), c);
    for (int j = 0; j < MI_TEST_BATCH; j++) for (int d = 0; d < 3; d++) r[3 * j + d] = t[j][d];
}
} // namespace batch_test

// base field: r, a, b are lanes of MI_TEST_BATCH rows; r is always contiguous
class GoldilocksB : public Goldilocks
{
    typedef batch_test::Str S;
    typedef batch_test::Off O;
#define MI_B1(name, OP)                                                                                                                              \
    static void name(Element *r, const Element *a, const Element *b) { batch_test::b11(OP, r, a, S{1}, b, S{1}); }                                  \
    static void name(Element *r, const Element *a, const Element *b, uint64_t sa, uint64_t sb) { batch_test::b11(OP, r, a, S{sa}, b, S{sb}); }      \
    static void name(Element *r, const Element *a, const Element *b, const uint64_t *oa, const uint64_t *ob) { batch_test::b11(OP, r, a, O{oa}, b, O{ob}); } \
    static void name(Element *r, const Element *a, const Element &c) { batch_test::b1c(OP, r, a, S{1}, c); }                                        \
    static void name(Element *r, const Element *a, const Element &c, uint64_t sa) { batch_test::b1c(OP, r, a, S{sa}, c); }                          \
    static void name(Element *r, const Element *a, const Element &c, const uint64_t *oa) { batch_test::b1c(OP, r, a, O{oa}, c); }                   \
    static void name(Element *r, const Element &c, const Element *b) { batch_test::bc1(OP, r, c, b, S{1}); }                                        \
    static void name(Element *r, const Element &c, const Element *b, uint64_t sb) { batch_test::bc1(OP, r, c, b, S{sb}); }                          \
    static void name(Element *r, const Element &c, const Element *b, const uint64_t *ob) { batch_test::bc1(OP, r, c, b, O{ob}); }
public:
    MI_B1(add_batch, batch_test::ADD)
    MI_B1(sub_batch, batch_test::SUB)
    MI_B1(mul_batch, batch_test::MUL)
#undef MI_B1
    static void copy_batch(Element *r, const Element *a) { for (int j = 0; j < MI_TEST_BATCH; j++) r[j] = a[j]; }
    static void copy_batch(Element *r, const Element *a, uint64_t sa) { for (int j = 0; j < MI_TEST_BATCH; j++) r[j] = a[j * sa]; }
    static void copy_batch(Element *r, const Element *a, const uint64_t *oa) { for (int j = 0; j < MI_TEST_BATCH; j++) r[j] = a[oa[j]]; }
    static void copy_batch(Element *r, const Element &c) { for (int j = 0; j < MI_TEST_BATCH; j++) r[j] = c; }
};

// cubic extension: r is MI_TEST_BATCH consecutive extension elements; a dimension-3 operand at a row is 3 consecutive words; the strides of
// the two-stride forms are in WORDS (the call sites pass FIELD_EXTENSION for a temporary, a section's width for a polynomial, 0 for a challenge)
class Goldilocks3B : public Goldilocks3
{
    typedef Goldilocks::Element E;
    typedef batch_test::Str S;
    typedef batch_test::Off O;
#define MI_B3(name, OP)                                                                                                                              \
    static void name(E *r, const E *a, const E *b) { batch_test::b33(OP, r, a, S{3}, b, S{3}); }                                                    \
    static void name(E *r, const E *a, const E *b, uint64_t sa, uint64_t sb) { batch_test::b33(OP, r, a, S{sa}, b, S{sb}); }                        \
    static void name(E *r, const E *a, const E *b, const uint64_t *oa, const uint64_t *ob) { batch_test::b33(OP, r, a, O{oa}, b, O{ob}); }
public:
    MI_B3(add_batch, batch_test::ADD)
    MI_B3(sub_batch, batch_test::SUB)
    MI_B3(mul_batch, batch_test::MUL)
#undef MI_B3
    // 13: base (lanes) with extension (lanes)
    static void add13_batch(E *r, const E *a, const E *b) { batch_test::b13(batch_test::ADD, r, a, S{1}, b, S{3}); }
    static void add13_batch(E *r, const E *a, const E *b, uint64_t sa, uint64_t sb) { batch_test::b13(batch_test::ADD, r, a, S{sa}, b, S{sb}); }
    static void mul13_batch(E *r, const E *a, const E *b, uint64_t sa, uint64_t sb) { batch_test::b13(batch_test::MUL, r, a, S{sa}, b, S{sb}); }
    static void mul13_batch(E *r, const E *a, const E *b, const uint64_t *oa, const uint64_t *ob) { batch_test::b13(batch_test::MUL, r, a, O{oa}, b, O{ob}); }
    // 13c: base (lanes) with ONE extension constant
    static void add13c_batch(E *r, const E *a, const E *c) { batch_test::b13(batch_test::ADD, r, a, S{1}, c, S{0}); }
    static void add13c_batch(E *r, const E *a, const E *c, uint64_t sa) { batch_test::b13(batch_test::ADD, r, a, S{sa}, c, S{0}); }
    static void mul13c_batch(E *r, const E *a, const E *c) { batch_test::b13(batch_test::MUL, r, a, S{1}, c, S{0}); }
    static void mul13c_batch(E *r, const E *a, const E *c, uint64_t sa) { batch_test::b13(batch_test::MUL, r, a, S{sa}, c, S{0}); }
    static void mul13c_batch(E *r, const E *a, const E *c, const uint64_t *oa) { batch_test::b13(batch_test::MUL, r, a, O{oa}, c, S{0}); }
    // 1c3c: one base constant with one extension constant, broadcast to the lanes
    static void add1c3c_batch(E *r, const E &a, const E *c) { batch_test::b13(batch_test::ADD, r, &a, S{0}, c, S{0}); }
    static void mul1c3c_batch(E *r, const E &a, const E *c) { batch_test::b13(batch_test::MUL, r, &a, S{0}, c, S{0}); }
    // 33c: extension (lanes) with one extension constant
    static void add33c_batch(E *r, const E *a, const E *c) { batch_test::b33(batch_test::ADD, r, a, S{3}, c, S{0}); }
    static void add33c_batch(E *r, const E *a, const E *c, uint64_t sa) { batch_test::b33(batch_test::ADD, r, a, S{sa}, c, S{0}); }
    static void sub33c_batch(E *r, const E *a, const E *c) { batch_test::b33(batch_test::SUB, r, a, S{3}, c, S{0}); }
    static void mul33c_batch(E *r, const E *a, const E *c) { batch_test::b33(batch_test::MUL, r, a, S{3}, c, S{0}); }
    static void mul33c_batch(E *r, const E *a, const E *c, uint64_t sa) { batch_test::b33(batch_test::MUL, r, a, S{sa}, c, S{0}); }
    static void mul33c_batch(E *r, const E *a, const E *c, const uint64_t *oa) { batch_test::b33(batch_test::MUL, r, a, O{oa}, c, S{0}); }
    // 31c: extension (lanes) minus one base constant
    static void sub31c_batch(E *r, const E *a, const E &c, uint64_t sa) { batch_test::b31c(batch_test::SUB, r, a, S{sa}, c); }
};
#endif
