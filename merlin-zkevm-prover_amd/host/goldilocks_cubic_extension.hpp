// goldilocks_cubic_extension.hpp -- F_p[x]/(x^3 - x - 1); multiplication as polinomial.hpp:195-205.
#ifndef GOLDILOCKS_CUBIC_EXTENSION
#define GOLDILOCKS_CUBIC_EXTENSION
#include "goldilocks_base_field.hpp"
#define FIELD_EXTENSION 3

class Goldilocks3
{
public:
    typedef Goldilocks::Element Element[FIELD_EXTENSION];

    static inline void copy(Element &dst, const Element &src) { MI_FIELD_RECORD(3, &dst, 3, &src, 3, nullptr, 0); for (int i = 0; i < 3; i++) dst[i] = src[i]; }
    static inline void copy(Element *dst, const Element *src) { MI_FIELD_RECORD(3, dst, 3, src, 3, nullptr, 0); for (int i = 0; i < 3; i++) (*dst)[i] = (*src)[i]; }
    static inline const Element &one() { static const Element o = {{1}, {0}, {0}}; return o; }
    static inline void one(Element &r) { r[0] = Goldilocks::one(); r[1] = r[2] = Goldilocks::zero(); }
    static inline bool isOne(const Element &a) { return Goldilocks::isOne(a[0]) && Goldilocks::isZero(a[1]) && Goldilocks::isZero(a[2]); }
    static inline void add(Element &r, const Element &a, const Element &b) { MI_FIELD_RECORD(0, &r, 3, &a, 3, &b, 3); for (int i = 0; i < 3; i++) r[i] = a[i] + b[i]; }
    static inline void sub(Element &r, const Element &a, const Element &b) { MI_FIELD_RECORD(1, &r, 3, &a, 3, &b, 3); for (int i = 0; i < 3; i++) r[i] = a[i] - b[i]; }
    static inline void mul(Element &r, const Element &a, const Element &b)
    {
        MI_FIELD_RECORD(2, &r, 3, &a, 3, &b, 3);
        Goldilocks::Element A = (a[0] + a[1]) * (b[0] + b[1]), B = (a[0] + a[2]) * (b[0] + b[2]), C = (a[1] + a[2]) * (b[1] + b[2]);
        Goldilocks::Element D = a[0] * b[0], E = a[1] * b[1], F = a[2] * b[2], G = D - E;
        Goldilocks::Element r0 = (C + G) - F, r1 = ((((A + C) - E) - E) - D), r2 = B - G;
        r[0] = r0; r[1] = r1; r[2] = r2;
    }
    static inline void mul(Element &r, const Element &a, const Goldilocks::Element &b) { MI_FIELD_RECORD(2, &r, 3, &a, 3, &b, 1); for (int i = 0; i < 3; i++) r[i] = a[i] * b; }
    // mixed forms (a base-field operand stands for (a, 0, 0)): what the generated per-row chelpers of the recursive STARKs call, e.g.
    // Goldilocks3::mul(tmp, params.x_n[i], challenge), Goldilocks3::add(tmp, pols[..], tmp3) (recursive1.chelpers.step3prev.cpp:11-19)
    static inline void mul(Element &r, const Goldilocks::Element &a, const Element &b) { MI_FIELD_RECORD(2, &r, 3, &a, 1, &b, 3); for (int i = 0; i < 3; i++) r[i] = a * b[i]; }
    static inline void add(Element &r, const Goldilocks::Element &a, const Element &b) { MI_FIELD_RECORD(0, &r, 3, &a, 1, &b, 3); r[0] = a + b[0]; r[1] = b[1]; r[2] = b[2]; }
    static inline void add(Element &r, const Element &a, const Goldilocks::Element &b) { MI_FIELD_RECORD(0, &r, 3, &a, 3, &b, 1); r[0] = a[0] + b; r[1] = a[1]; r[2] = a[2]; }
    static inline void sub(Element &r, const Goldilocks::Element &a, const Element &b) { MI_FIELD_RECORD(1, &r, 3, &a, 1, &b, 3); r[0] = a - b[0]; r[1] = Goldilocks::neg(b[1]); r[2] = Goldilocks::neg(b[2]); }
    static inline void sub(Element &r, const Element &a, const Goldilocks::Element &b) { MI_FIELD_RECORD(1, &r, 3, &a, 3, &b, 1); r[0] = a[0] - b; r[1] = a[1]; r[2] = a[2]; }
    static inline void copy(Element &dst, const Goldilocks::Element &src) { MI_FIELD_RECORD(3, &dst, 3, &src, 1, nullptr, 0); dst[0] = src; dst[1] = dst[2] = Goldilocks::zero(); }
    static inline void inv(Element *r, const Element *a) { inv(*r, *a); }
    static inline void inv(Element &r, const Element &a)
    {
        // adjugate of the multiplication-by-a matrix (x^3 = x + 1); inv(0) = 0
        Goldilocks::Element a0 = a[0], a1 = a[1], a2 = a[2];
        Goldilocks::Element m11 = a0 + a2, m12 = a1 + a2, m22 = a0 + a2;
        Goldilocks::Element c00 = m11 * m22 - m12 * a1, c01 = a1 * m22 - m12 * a2, c02 = a1 * a1 - m11 * a2;
        Goldilocks::Element det = (a0 * c00 - a2 * c01) + a1 * c02, di = Goldilocks::inv(det);
        r[0] = c00 * di; r[1] = Goldilocks::neg(c01) * di; r[2] = c02 * di;
    }
};
#endif
