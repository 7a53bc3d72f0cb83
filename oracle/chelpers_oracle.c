/*
 * chelpers_oracle.c -- CPU ORACLE (TEST INFRASTRUCTURE, NOT PRODUCT CODE) for the generated constraint evaluators.
 *
 * Restates, opcode by opcode, the reference interpreter of the step42ns program:
 *   cases 0-83  : ZkevmSteps::step42ns_parser_first        (zkevm.chelpers.step42ns.parser.cpp:781-1383, the scalar
 *                 "_batch" form; identical case for case to step42ns_parser_first_avx, :24-660)
 *   cases 84-88 : the fused opcodes of step42ns_parser_first_avx (:661-748), which are what the generated table
 *                 zkevm.chelpers.step42ns.parser.hpp actually uses (the scalar function still numbers its three
 *                 fusions 110-112 and has no 87 / 88).
 * One row at a time (the reference walks rows in batches of nrowsBatch = 4 or 8; the batch index j only ever appears as
 * row i + j).  Temporaries are the reference's tmp1[] / tmp3[] arrays, indexed by the table's own slot numbers -- no
 * rescheduling, no renaming: this is deliberately the naive reading the product's translator has to agree with.
 *
 * PARITY: pinned in part, by reference CODE run in the tests.  The reference tree holds no input / output pair for any chelpers step
 * (zkevm.starkinfo.json, the constant polynomials and every proof-time trace are absent).  But it holds the zkEVM's constraint system
 * a second time, as generated per-row C++ (zkevm.chelpers.step{2,3prev,3,52ns}.cpp), which compiles against the Level-0 field classes:
 * tests/test_steps_tracer.py runs those functions beside the product's table decoder on the reference's tables (identical stores), and
 * tests/test_chelpers.py runs the product's decoder beside these interpreters on the same tables (identical) -- so for step2prev,
 * step3prev, step3 and step52ns this restatement computes what the reference's compiled code computes.  The 19 opcodes only
 * step42ns's table uses (9, 14, 25, 28, 29, 33, 34, 36, 39-42, 55, 60, 69, 72, 74, 75, 77; its per-row file is an absent blob) are
 * decided since round 4 by the reference's SCALAR interpreter itself: tests/test_reference_scalar_interpreter.py compiles
 * step42ns_parser_first unchanged over test-only `_batch` helpers (tests/cpp/batch_helpers_test_only.hpp) and runs the real 11 959-opcode
 * table beside this file and the product's decoder -- identical q_2ns.  The helpers' arithmetic is this repo's Level-0 stand-in, so that
 * pins every case's operand ADDRESSING and helper choice; the field arithmetic itself is pinned by the golden proofs (gl_oracle.h).
 */
#include "gl_oracle.h"
#include <stdlib.h>
#include <string.h>

typedef struct {
    const uint64_t *pols, *cpols, *chal, *pub, *x, *zhinv;
    uint64_t numpols, x_stride, n_zhinv;
    uint64_t *q;
    uint64_t *pols_w; /* the base-domain steps write params.pols */
} chp_env;

/* operand readers at row i, argument position k of the current opcode */
#define A(k) args[ia + (k)]
#define T1(k) tmp1[A(k)]
#define T3(k) (&tmp3[3 * A(k)])
#define NUM(k) glo_canon(A(k))                                               /* Goldilocks::fromU64 */
#define POL(k) glo_canon(e->pols[A(k) + i * A((k) + 1)])                      /* &params.pols[a + i * stride] */
#define POLP(k) (&e->pols[A(k) + i * A((k) + 1)])                             /* ... as the address of 3 elements */
#define POLS(k) glo_canon(e->pols[A(k) + ((i + A((k) + 1)) % A((k) + 2)) * A((k) + 3)]) /* offsets[j] = a + ((i+j+s) % n) * stride */
#define POLSP(k) (&e->pols[A(k) + ((i + A((k) + 1)) % A((k) + 2)) * A((k) + 3)])
#define CST(k) glo_canon(e->cpols[A(k) + i * e->numpols])                     /* pConstPols2ns->getElement(col, i) */
#define CSTS(k) glo_canon(e->cpols[A(k) + ((i + A((k) + 1)) % A((k) + 2)) * e->numpols])
#define CHAL(k) (&e->chal[3 * A(k)])                                         /* params.challenges[k] */
#define PUB(k) glo_canon(e->pub[A(k)])

static void set3(uint64_t *d, uint64_t a, uint64_t b, uint64_t c) { d[0] = a; d[1] = b; d[2] = c; }
static void ld3(uint64_t o[3], const uint64_t *p) { o[0] = glo_canon(p[0]); o[1] = glo_canon(p[1]); o[2] = glo_canon(p[2]); }
/* Goldilocks3::add13 / add1c3c: base + ext touches component 0 only; sub31c: ext - base likewise */
static void add13(uint64_t *d, uint64_t a, const uint64_t *b) { uint64_t t[3]; ld3(t, b); set3(d, glo_add(a, t[0]), t[1], t[2]); }
static void add33(uint64_t *d, const uint64_t *a, const uint64_t *b) { uint64_t s[3], t[3]; ld3(s, a); ld3(t, b); set3(d, glo_add(s[0], t[0]), glo_add(s[1], t[1]), glo_add(s[2], t[2])); }
static void sub33(uint64_t *d, const uint64_t *a, const uint64_t *b) { uint64_t s[3], t[3]; ld3(s, a); ld3(t, b); set3(d, glo_sub(s[0], t[0]), glo_sub(s[1], t[1]), glo_sub(s[2], t[2])); }
static void sub31(uint64_t *d, const uint64_t *a, uint64_t b) { uint64_t s[3]; ld3(s, a); set3(d, glo_sub(s[0], b), s[1], s[2]); }
static void mul13(uint64_t *d, uint64_t a, const uint64_t *b) { uint64_t t[3]; ld3(t, b); set3(d, glo_mul(a, t[0]), glo_mul(a, t[1]), glo_mul(a, t[2])); }
static void mul33(uint64_t *d, const uint64_t *a, const uint64_t *b) { uint64_t s[3], t[3], o[3]; ld3(s, a); ld3(t, b); glo3_mul(o, s, t); set3(d, o[0], o[1], o[2]); }

/* returns the number of arguments consumed, or -1 for an unknown opcode */
static int chp_step42ns_op(uint64_t op, const uint64_t *args, uint64_t ia, uint64_t i, uint64_t *tmp1, uint64_t *tmp3, const chp_env *e)
{
    switch (op) {
    /* ---- Goldilocks::add_batch */
    case 0: T1(0) = glo_add(T1(1), T1(2)); return 3;
    case 1: T1(0) = glo_add(T1(1), POL(2)); return 4;
    case 2: T1(0) = glo_add(T1(1), NUM(2)); return 3;
    case 3: T1(0) = glo_add(T1(1), CST(2)); return 3;
    case 4: T1(0) = glo_add(POL(1), POL(3)); return 5;
    case 5: T1(0) = glo_add(POLS(1), POLS(5)); return 9;
    case 6: T1(0) = glo_add(POL(1), CST(3)); return 4;
    case 7: T1(0) = glo_add(POL(1), NUM(3)); return 4;
    case 8: T1(0) = glo_add(CST(1), CST(2)); return 3;
    case 9: T1(0) = glo_add(CSTS(1), CSTS(4)); return 7;
    case 10: T1(0) = glo_add(CST(1), NUM(2)); return 3;
    case 11: T1(0) = glo_add(CSTS(1), NUM(4)); return 5;
    /* ---- Goldilocks3 additions */
    case 12: add13(T3(0), T1(1), T3(2)); return 3;
    case 13: add13(T3(0), NUM(1), CHAL(2)); return 3;
    case 14: add13(T3(0), T1(1), CHAL(2)); return 3;
    case 15: add13(T3(0), POL(1), T3(3)); return 4;
    case 16: add13(T3(0), POL(1), CHAL(3)); return 4;
    case 17: add33(T3(0), T3(1), T3(2)); return 3;
    case 18: add33(T3(0), T3(1), CHAL(2)); return 3;
    case 19: add33(T3(0), POLP(1), T3(3)); return 4;
    case 20: add33(T3(0), POLP(1), CHAL(3)); return 4;
    /* ---- Goldilocks::sub_batch */
    case 21: T1(0) = glo_sub(T1(1), T1(2)); return 3;
    case 22: T1(0) = glo_sub(T1(1), POL(2)); return 4;
    case 23: T1(0) = glo_sub(T1(1), POLS(2)); return 6;
    case 24: T1(0) = glo_sub(POL(1), T1(3)); return 4;
    case 25: T1(0) = glo_sub(POLS(1), T1(5)); return 6;
    case 26: T1(0) = glo_sub(T1(1), NUM(2)); return 3;
    case 27: T1(0) = glo_sub(NUM(1), T1(2)); return 3;
    case 28: T1(0) = glo_sub(POL(1), NUM(3)); return 4;
    case 29: T1(0) = glo_sub(POLS(1), NUM(5)); return 6;
    case 30: T1(0) = glo_sub(NUM(1), POL(2)); return 4;
    case 31: T1(0) = glo_sub(NUM(1), POLS(2)); return 6;
    case 32: T1(0) = glo_sub(NUM(1), CST(2)); return 3;
    case 33: T1(0) = glo_sub(NUM(1), CSTS(2)); return 5;
    case 34: T1(0) = glo_sub(POL(1), PUB(3)); return 4;
    case 35: T1(0) = glo_sub(POLS(1), POL(5)); return 7;
    case 36: T1(0) = glo_sub(POL(1), POLS(3)); return 7;
    case 37: T1(0) = glo_sub(POL(1), POL(3)); return 5;
    case 38: T1(0) = glo_sub(POLS(1), POLS(5)); return 9;
    case 39: T1(0) = glo_sub(CST(1), POL(2)); return 4;
    case 40: T1(0) = glo_sub(T1(1), CST(2)); return 3;
    /* ---- Goldilocks3 subtractions */
    case 41: sub31(T3(0), POLP(1), NUM(3)); return 4;
    case 42: sub33(T3(0), T3(1), T3(2)); return 3;
    case 43: sub33(T3(0), T3(1), CHAL(2)); return 3;
    case 44: sub33(T3(0), T3(1), POLP(2)); return 4;
    /* ---- Goldilocks::mul_batch */
    case 45: T1(0) = glo_mul(T1(1), T1(2)); return 3;
    case 46: T1(0) = glo_mul(NUM(1), T1(2)); return 3;
    case 47: T1(0) = glo_mul(POL(1), T1(3)); return 4;
    case 48: T1(0) = glo_mul(POLS(1), T1(5)); return 6;
    case 49: T1(0) = glo_mul(T1(1), CST(2)); return 3;
    case 50: T1(0) = glo_mul(POL(1), POL(3)); return 5;
    case 51: T1(0) = glo_mul(POL(1), POLS(3)); return 7;
    case 52: T1(0) = glo_mul(POLS(1), POLS(5)); return 9;
    case 53: T1(0) = glo_mul(NUM(1), POL(2)); return 4;
    case 54: T1(0) = glo_mul(POL(1), CST(3)); return 4;
    case 55: T1(0) = glo_mul(POLS(1), CST(5)); return 6; /* offsets2[j] = a5 + (i + j) * numpols */
    case 56: T1(0) = glo_mul(T1(1), POL(2)); return 4;
    case 57: T1(0) = glo_mul(T1(1), POLS(2)); return 6;
    case 58: T1(0) = glo_mul(CST(1), T1(2)); return 3;
    /* ---- Goldilocks3 multiplications */
    case 59: mul13(T3(0), T1(1), CHAL(2)); return 3;
    case 60: mul13(T3(0), CST(1), T3(2)); return 3;
    case 61: mul13(T3(0), T1(1), T3(2)); return 3;
    case 62: mul13(T3(0), POL(1), CHAL(3)); return 4;
    case 63: mul13(T3(0), POLS(1), CHAL(5)); return 6;
    case 64: mul13(T3(0), POL(1), T3(3)); return 4;
    case 65: mul13(T3(0), POLS(1), T3(5)); return 6;
    case 66: mul13(T3(0), NUM(1), CHAL(2)); return 3;
    case 67: mul13(T3(0), glo_canon(e->x[i * e->x_stride]), CHAL(1)); return 2;
    case 68: mul13(T3(0), glo_canon(e->x[i * e->x_stride]), T3(1)); return 2;
    case 69: mul13(&e->q[3 * i], glo_canon(e->zhinv[i % e->n_zhinv]), T3(0)); return 1; /* q_2ns[i] = zhInv(i) * tmp3 */
    case 70: mul33(T3(0), T3(2), CHAL(1)); return 3; /* mul33c(dst, tmp3[a2], challenges[a1]) */
    case 71: mul33(T3(0), T3(1), T3(2)); return 3;
    case 72: mul33(T3(0), POLP(1), POLP(3)); return 5;
    case 73: mul33(T3(0), POLSP(1), CHAL(5)); return 6;
    case 74: mul33(T3(0), POLSP(1), T3(5)); return 6;
    case 75: mul33(T3(0), POLP(1), T3(3)); return 4;
    case 76: mul33(T3(0), POLP(1), CHAL(3)); return 4;
    case 77: mul33(T3(0), POLSP(1), POLP(5)); return 7;
    /* ---- Goldilocks::copy_batch */
    case 78: T1(0) = T1(1); return 2;
    case 79: T1(0) = POL(1); return 3;
    case 80: T1(0) = POLS(1); return 5;
    case 81: T1(0) = NUM(1); return 2;
    case 82: T1(0) = CST(1); return 2;
    case 83: T1(0) = CSTS(1); return 4;
    }
    return -1;
}

static const int F84[] = {12, 70, -1}, F85[] = {0, 50, -1}, F86[] = {32, 47, 21, 32, 48, -1}, F87[] = {12, 70, 12, 70, 12, 70, 12, 70, -1},
                 F88[] = {21, 50, 21, 53, 0, 0, 50, 50, 0, 50, 21, 50, -1};

/* Runs the program on rows row0 .. row0 + nrows - 1.  Returns 0, -1 (unknown opcode) or -2 (argument count mismatch:
 * the reference asserts i_args == NARGS_ after every row). */
int glo_chelpers_step42ns(const uint64_t *ops, uint64_t nops, const uint64_t *args, uint64_t nargs, const uint64_t *pols,
                          const uint64_t *const_pols, uint64_t numpols, const uint64_t *challenges, const uint64_t *publics,
                          const uint64_t *x, uint64_t x_stride, const uint64_t *zhinv, uint64_t n_zhinv, uint64_t *q,
                          uint64_t row0, uint64_t nrows)
{
    /* temp arrays as large as the highest slot any argument could name (the reference uses NTEMP1_ / NTEMP3_) */
    uint64_t maxarg = 0;
    for (uint64_t k = 0; k < nargs; k++)
        if (args[k] < (1u << 20) && args[k] > maxarg) maxarg = args[k];
    int status = 0;
#pragma omp parallel
    {
        uint64_t *tmp1 = (uint64_t *)calloc(maxarg + 1, sizeof(uint64_t));
        uint64_t *tmp3 = (uint64_t *)calloc(3 * (maxarg + 1), sizeof(uint64_t));
        chp_env e = {pols, const_pols, challenges, publics, x, zhinv, numpols, x_stride, n_zhinv, q, NULL};
#pragma omp for schedule(static)
        for (uint64_t r = 0; r < nrows; r++) {
            const uint64_t i = row0 + r;
            uint64_t ia = 0;
            int bad = 0;
            for (uint64_t kk = 0; kk < nops && !bad; kk++) {
                const int *f = ops[kk] == 84 ? F84 : ops[kk] == 85 ? F85 : ops[kk] == 86 ? F86 : ops[kk] == 87 ? F87 : ops[kk] == 88 ? F88 : NULL;
                if (f) {
                    for (; *f >= 0 && !bad; f++) {
                        const int n = chp_step42ns_op((uint64_t)*f, args, ia, i, tmp1, tmp3, &e);
                        if (n < 0) bad = -1; else ia += (uint64_t)n;
                    }
                } else {
                    const int n = chp_step42ns_op(ops[kk], args, ia, i, tmp1, tmp3, &e);
                    if (n < 0) bad = -1; else ia += (uint64_t)n;
                }
            }
            if (!bad && ia != nargs) bad = -2;
            if (bad) {
#pragma omp critical
                status = bad;
            }
        }
        free(tmp1);
        free(tmp3);
    }
    return status;
}

/* ------------------------------------------------------------------ step52ns: the FRI polynomial f_2ns
 * ZkevmSteps::step52ns_parser_first (zkevm.chelpers.step52ns.parser.cpp:520-690; same 21 cases as the AVX function :10-190):
 * three extension registers tmp / tmp1 / tmp2, challenges 5 and 6, params.evals, xDivXSubXi[i] / xDivXSubWXi[i], output
 * f_2ns[i].  (The reference pre-adds the challenge components for its Karatsuba multiply; any exact extension multiply gives
 * the same element.)  Returns 0, -1 (unknown opcode), -2 (argument count mismatch). */
static int chp_step52ns_op(uint64_t op, const uint64_t *args, uint64_t ia, uint64_t i, uint64_t *tmp, uint64_t *tmp1, uint64_t *tmp2,
                           const uint64_t *pols, const uint64_t *cpols, uint64_t numpols, const uint64_t *chal, const uint64_t *evals,
                           const uint64_t *xd, const uint64_t *xdw, uint64_t *f)
{
#define P52(k) (&pols[A(k) + i * A((k) + 1)])
    switch (op) {
    case 0: mul13(tmp, glo_canon(*P52(0)), &chal[15]); return 2;                    /* tmp = pols * challenges[5] */
    case 1: mul33(tmp, tmp, &chal[15]); return 0;
    case 2: mul33(tmp, tmp, &chal[18]); return 0;
    case 3: mul33(tmp1, tmp, &chal[15]); return 0;
    case 4: mul33(tmp, tmp2, &chal[18]); return 0;
    case 5: mul33(tmp, tmp, &xd[3 * i]); return 0;
    case 6: mul33(tmp, tmp, &xdw[3 * i]); return 0;
    case 7: add33(tmp, tmp, tmp2); return 0;
    case 8: add33(tmp, tmp1, tmp); return 0;
    case 9: add33(tmp, tmp, P52(0)); return 2;
    case 10: add13(tmp, glo_canon(*P52(0)), tmp); return 2;                         /* add31: ext + base */
    case 11: { uint64_t e[3]; ld3(e, &evals[3 * A(2)]); set3(tmp2, glo_sub(glo_canon(*P52(0)), e[0]), glo_sub(0, e[1]), glo_sub(0, e[2])); return 3; } /* sub13c */
    case 12: sub33(tmp2, P52(0), &evals[3 * A(2)]); return 3;
    case 13: { uint64_t e[3]; ld3(e, &evals[3 * A(1)]); set3(tmp2, glo_sub(glo_canon(cpols[A(0) + i * numpols]), e[0]), glo_sub(0, e[1]), glo_sub(0, e[2])); return 2; }
    case 14: { uint64_t e[3]; ld3(e, &evals[0]); set3(tmp, glo_sub(glo_canon(cpols[5 + i * numpols]), e[0]), glo_sub(0, e[1]), glo_sub(0, e[2])); return 0; }
    case 15: { uint64_t t[3]; ld3(t, tmp); set3(&f[3 * i], t[0], t[1], t[2]); return 0; }
    }
#undef P52
    return -1;
}

int glo_chelpers_step52ns(const uint64_t *ops, uint64_t nops, const uint64_t *args, uint64_t nargs, const uint64_t *pols,
                          const uint64_t *const_pols, uint64_t numpols, const uint64_t *challenges, const uint64_t *evals,
                          const uint64_t *xdiv, const uint64_t *xdivw, uint64_t *f, uint64_t row0, uint64_t nrows)
{
    static const int G16[] = {1, 10, -1}, G17[] = {1, 9, -1}, G18[] = {2, 11, 7, -1}, G19[] = {2, 13, 7, -1}, G20[] = {2, 12, 7, -1};
    int status = 0;
#pragma omp parallel for schedule(static)
    for (uint64_t r = 0; r < nrows; r++) {
        const uint64_t i = row0 + r;
        uint64_t tmp[3] = {0, 0, 0}, tmp1[3] = {0, 0, 0}, tmp2[3] = {0, 0, 0}, ia = 0;
        int bad = 0;
        for (uint64_t kk = 0; kk < nops && !bad; kk++) {
            const int *fu = ops[kk] == 16 ? G16 : ops[kk] == 17 ? G17 : ops[kk] == 18 ? G18 : ops[kk] == 19 ? G19 : ops[kk] == 20 ? G20 : NULL;
            if (fu) {
                for (; *fu >= 0 && !bad; fu++) {
                    const int n = chp_step52ns_op((uint64_t)*fu, args, ia, i, tmp, tmp1, tmp2, pols, const_pols, numpols, challenges, evals, xdiv, xdivw, f);
                    if (n < 0) bad = -1; else ia += (uint64_t)n;
                }
            } else {
                const int n = chp_step52ns_op(ops[kk], args, ia, i, tmp, tmp1, tmp2, pols, const_pols, numpols, challenges, evals, xdiv, xdivw, f);
                if (n < 0) bad = -1; else ia += (uint64_t)n;
            }
        }
        if (!bad && ia != nargs) bad = -2;
        if (bad) {
#pragma omp critical
            status = bad;
        }
    }
    return status;
}


/* ------------------------------------------------------------------ the base-domain steps: step2prev / step3prev / step3
 * ZkevmSteps::step{2prev,3prev,3}_parser_first_avx (zkevm.chelpers.step2prev.parser.cpp:9-, step3prev.parser.cpp:9-,
 * step3.parser.cpp:9-): one opcode numbering for the three.  Cases 0-83 are the step42ns cases above, word for word, reading
 * params.pConstPols / params.x_n where step42ns reads pConstPols2ns / x_2ns (the caller passes those); 69 is not used by them.
 * 84-115 below, one line per case of the reference: 86-100 store into &params.pols[a0 + i * a1] (args 0, 1), 101-114 into
 * &params.pols[0] + offsets1, offsets1 = a0 + ((i + a1) % a2) * a3 (args 0..3); 91, 97, 99 are "code not used" asserts there;
 * 115 (step3 only) is the fusion "0, 50".  Rows are walked in order on one thread: a row may store into the next row's cell. */
#define DST (&e->pols_w[A(0) + i * A(1)])
#define DSTS (&e->pols_w[A(0) + ((i + A(1)) % A(2)) * A(3)])
static int chp_stepbase_op(uint64_t op, const uint64_t *args, uint64_t ia, uint64_t i, uint64_t *tmp1, uint64_t *tmp3, const chp_env *e)
{
    if (op <= 83 && op != 69) return chp_step42ns_op(op, args, ia, i, tmp1, tmp3, e);
    switch (op) {
    case 84: T1(0) = glo_add(T1(1), POLS(2)); return 6;
    case 85: T1(0) = glo_mul(POLS(1), NUM(5)); return 6;
    case 86: *DST = glo_add(T1(2), T1(3)); return 4;
    case 87: *DST = glo_add(T1(2), POL(3)); return 5;
    case 88: add13(DST, T1(2), T3(3)); return 4;
    case 89: add33(DST, POLP(2), T3(4)); return 5;
    case 90: add33(DST, T3(2), CHAL(3)); return 4;
    case 92: *DST = glo_sub(T1(2), T1(3)); return 4;
    case 93: *DST = glo_sub(NUM(2), T1(3)); return 4;
    case 94: *DST = glo_mul(T1(2), T1(3)); return 4;
    case 95: *DST = glo_mul(POL(2), T1(4)); return 5;
    case 96: *DST = glo_mul(T1(2), CST(3)); return 4;
    case 98: mul33(DST, T3(2), T3(3)); return 4;
    case 100: *DST = T1(2); return 3;
    case 101: *DSTS = glo_add(T1(4), T1(5)); return 6;
    case 102: *DSTS = glo_add(T1(4), POL(5)); return 7;
    case 103: add13(DSTS, T1(4), T3(5)); return 6;
    case 104: add33(DSTS, POLP(4), T3(6)); return 7;
    case 105: add33(DSTS, T3(4), CHAL(5)); return 6;
    case 106: *DSTS = glo_sub(T1(4), T1(5)); return 6;
    case 107: *DSTS = glo_sub(NUM(4), T1(5)); return 6;
    case 108: *DSTS = glo_mul(T1(4), T1(5)); return 6;
    case 109: *DSTS = glo_mul(POL(4), T1(6)); return 7;
    case 110: *DSTS = glo_mul(T1(4), CST(5)); return 6;
    case 111: *DSTS = glo_mul(CSTS(4), T1(7)); return 8;
    case 112: mul33(DSTS, T3(4), T3(5)); return 6;
    case 113: *DSTS = T1(4); return 5;
    case 114: *DSTS = glo_add(T1(4), POLS(5)); return 9;
    }
    return -1;
}
#undef DST
#undef DSTS

int glo_chelpers_stepbase(const uint64_t *ops, uint64_t nops, const uint64_t *args, uint64_t nargs, uint64_t *pols,
                          const uint64_t *const_pols, uint64_t numpols, const uint64_t *challenges, const uint64_t *publics,
                          const uint64_t *x, uint64_t x_stride, const uint64_t *rows, uint64_t nrows)
{
    static const int G115[] = {0, 50, -1};
    uint64_t maxarg = 0;
    for (uint64_t k = 0; k < nargs; k++)
        if (args[k] < (1u << 20) && args[k] > maxarg) maxarg = args[k];
    chp_env e = {pols, const_pols, challenges, publics, x, NULL, numpols, x_stride, 1, NULL, pols};
    int bad = 0;
    /* Rows are independent but for one thing: a row may store into the NEXT row's cell (cases 101-114), which that row's own evaluation
     * stores too -- with the same value (the generator emits both; the product relies on it as well: DESIGN_HISTORY.md "base-domain steps").
     * Many rows (a whole proof of the oracle prover, tests/oracle_genproof.py) are therefore walked by all threads, each with its own
     * temporaries; a few rows stay on one thread in the given order. */
    const int parallel = nrows >= 4096;
#pragma omp parallel if (parallel)
    {
        uint64_t *tmp1 = (uint64_t *)calloc(maxarg + 1, sizeof(uint64_t));
        uint64_t *tmp3 = (uint64_t *)calloc(3 * (maxarg + 1), sizeof(uint64_t));
        int mybad = 0;
#pragma omp for schedule(static)
        for (uint64_t r = 0; r < nrows; r++) {
            if (mybad) continue;
            const uint64_t i = rows[r];
            uint64_t ia = 0;
            for (uint64_t kk = 0; kk < nops && !mybad; kk++) {
                const int *f = ops[kk] == 115 ? G115 : NULL;
                if (f) {
                    for (; *f >= 0 && !mybad; f++) {
                        const int n = chp_stepbase_op((uint64_t)*f, args, ia, i, tmp1, tmp3, &e);
                        if (n < 0) mybad = -1; else ia += (uint64_t)n;
                    }
                } else {
                    const int n = chp_stepbase_op(ops[kk], args, ia, i, tmp1, tmp3, &e);
                    if (n < 0) mybad = -1; else ia += (uint64_t)n;
                }
            }
            if (!mybad && ia != nargs) mybad = -2;
        }
        if (mybad) {
#pragma omp critical
            bad = mybad;
        }
        free(tmp1);
        free(tmp3);
    }
    return bad;
}
