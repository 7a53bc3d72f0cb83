// goldilocks_base_field.hpp -- same-named stand-in for the upstream header the reference includes
// (polinomial.hpp:4, transcript.hpp:4, merkleTreeGL.hpp:4; API reconstructed in SURVEY.md App. B).
// The standard headers below are the ones reference code relies on getting through this header (polinomial.hpp uses assert,
// std::map, std::vector and the omp_* calls without including them.
// Scalar host arithmetic only (the reference calls these on single elements: shiftIn, x tables, challenges);
// every bulk loop of the hot path goes through libmi_stark.
#ifndef GOLDILOCKS_BASE_FIELD
#define GOLDILOCKS_BASE_FIELD
#include <cassert>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <map>
#include <omp.h>
#include <string>
#include <vector>

#define GOLDILOCKS_PRIME 0xFFFFFFFF00000001ULL

// Recording hook (host/steps_tracer.hpp): while a recorder is installed on the calling thread, every destination-form field operation
// below (and of Goldilocks3) reports its operands' ADDRESSES and dimensions before it computes.  That is how a Steps class's generated
// per-row code (straight-line calls of exactly these functions) is turned into a program for the device by running it once.
// cls: 0 add, 1 sub, 2 mul, 3 copy.  The value-returning forms and operators cannot be followed (their results have no address yet):
// used outside a destination form while a recorder is installed they set `untracked`, and the recorder refuses the function.
// One predictable branch on a plain global per operation when no recorder is installed.
struct MiFieldRecorder
{
    int depth = 0;          // inside a destination form (whose own arithmetic is not the caller's)
    bool untracked = false; // the caller computed with operators / value-returning forms
    virtual void op(int cls, void *r, int rdim, const void *a, int adim, const void *b, int bdim) = 0; // r: written, not read
    virtual ~MiFieldRecorder() {}
};
inline thread_local MiFieldRecorder *mi_field_recorder = nullptr;
// set while ANY thread has a recorder installed and tested first: in -fPIC objects (libmi_starks.so, a generated Steps library) a
// thread_local costs a __tls_get_addr call, which an ordinary run -- MI_STEPS_ON_HOST's row loops, any host code on these headers -- must
// not pay per field operation; a plain global is one load through the GOT.  (Both variables must unify across shared objects: a Steps
// library is loaded RTLD_GLOBAL after the library that records it, INTEGRATION.md.)  A COUNT of installed recorders, not a flag: two
// threads may trace at once (two Starks built in parallel), and the first to finish must not switch the other's recording off -- the
// relaxed load is still one plain load.
inline std::atomic<int> mi_field_recording{0};
#define MI_FIELD_RECORDER() (__builtin_expect(mi_field_recording.load(std::memory_order_relaxed) != 0, 0) ? mi_field_recorder : (MiFieldRecorder *)nullptr)
struct MiFieldScope
{
    MiFieldRecorder *const r;
    MiFieldScope(MiFieldRecorder *rec, int cls, void *d, int ddim, const void *a, int adim, const void *b, int bdim) : r(rec)
    {
        if (__builtin_expect(r != nullptr, 0)) { if (r->depth == 0) r->op(cls, d, ddim, a, adim, b, bdim); r->depth++; }
    }
    ~MiFieldScope() { if (__builtin_expect(r != nullptr, 0)) r->depth--; }
};
#define MI_FIELD_RECORD(cls, r, rdim, a, adim, b, bdim) MiFieldScope mi_field_scope_(MI_FIELD_RECORDER(), cls, r, rdim, a, adim, b, bdim)
#define MI_FIELD_VALUE_FORM() \
    do { if (MiFieldRecorder *mi_r_ = MI_FIELD_RECORDER()) { if (mi_r_->depth == 0) mi_r_->untracked = true; } } while (0)

class Goldilocks
{
public:
    typedef struct { uint64_t fe; } Element;

    static inline const Element &zero() { static const Element z = {0}; return z; }
    static inline const Element &one() { static const Element o = {1}; return o; }
    static inline const Element &shift() { static const Element s = {49}; return s; }
    static inline Element fromU64(uint64_t v) { return {v >= GOLDILOCKS_PRIME ? v - GOLDILOCKS_PRIME : v}; }
    static inline uint64_t toU64(const Element &a) { return a.fe >= GOLDILOCKS_PRIME ? a.fe - GOLDILOCKS_PRIME : a.fe; }
    static inline void toU64(uint64_t &out, const Element &a) { out = toU64(a); }
    static inline std::string toString(const Element &a, int radix = 10)
    {
        uint64_t v = toU64(a);
        if (radix == 16) { char b[32]; std::snprintf(b, sizeof b, "%llx", (unsigned long long)v); return b; }
        return std::to_string(v);
    }
    static inline std::string toString(const Element *a, int radix) { return toString(*a, radix); }

    static inline Element add(const Element &a, const Element &b)
    {
        MI_FIELD_VALUE_FORM();
        uint64_t x = toU64(a), y = toU64(b), s = x + y;
        if (s < x || s >= GOLDILOCKS_PRIME) s -= GOLDILOCKS_PRIME;
        return {s};
    }
    static inline Element sub(const Element &a, const Element &b)
    {
        MI_FIELD_VALUE_FORM();
        uint64_t x = toU64(a), y = toU64(b);
        return {x >= y ? x - y : x + (GOLDILOCKS_PRIME - y)};
    }
    static inline Element mul(const Element &a, const Element &b)
    {
        MI_FIELD_VALUE_FORM();
        unsigned __int128 p = (unsigned __int128)a.fe * b.fe;
        uint64_t lo = (uint64_t)p, hi = (uint64_t)(p >> 64), hh = hi >> 32, hl = hi & 0xFFFFFFFFULL;
        uint64_t t0 = lo - hh;
        if (lo < hh) t0 -= 0xFFFFFFFFULL;
        uint64_t t1 = hl * 0xFFFFFFFFULL, r = t0 + t1;
        if (r < t1) r += 0xFFFFFFFFULL;
        return {r >= GOLDILOCKS_PRIME ? r - GOLDILOCKS_PRIME : r};
    }
    static inline Element square(const Element &a) { return mul(a, a); }
    static inline Element exp(Element base, uint64_t e)
    {
        Element r = one();
        while (e) { if (e & 1) r = mul(r, base); base = mul(base, base); e >>= 1; }
        return r;
    }
    static inline Element inv(const Element &a) { return exp(a, GOLDILOCKS_PRIME - 2); }
    static inline Element neg(const Element &a) { return sub(zero(), a); }
    // out-parameter forms used by the reference (e.g. zhInv.cpp:21-27, starks.hpp:153)
    static inline void add(Element &r, const Element &a, const Element &b) { MI_FIELD_RECORD(0, &r, 1, &a, 1, &b, 1); r = add(a, b); }
    static inline void sub(Element &r, const Element &a, const Element &b) { MI_FIELD_RECORD(1, &r, 1, &a, 1, &b, 1); r = sub(a, b); }
    static inline void mul(Element &r, const Element &a, const Element &b) { MI_FIELD_RECORD(2, &r, 1, &a, 1, &b, 1); r = mul(a, b); }
    static inline void square(Element &r, const Element &a) { MI_FIELD_RECORD(2, &r, 1, &a, 1, &a, 1); r = mul(a, a); }
    static inline void inv(Element &r, const Element &a) { r = inv(a); }
    static inline void copy(Element &r, const Element &a) { MI_FIELD_RECORD(3, &r, 1, &a, 1, nullptr, 0); r = a; }
    static inline bool isZero(const Element &a) { return toU64(a) == 0; }
    static inline bool isOne(const Element &a) { return toU64(a) == 1; }
    static inline bool equal(const Element &a, const Element &b) { return toU64(a) == toU64(b); }
    // primitive 2^nbits-th root of unity: 7277203076849721926^(2^(32-nbits))
    static inline Element w(uint64_t nbits)
    {
        Element r = {7277203076849721926ULL};
        for (uint64_t i = nbits; i < 32; i++) r = mul(r, r);
        return r;
    }
    static inline void parcpy(Element *dst, const Element *src, uint64_t size, int /*num_threads_copy*/ = 64)
    {
        std::memcpy(dst, src, size * sizeof(Element));
    }
};

inline Goldilocks::Element operator+(const Goldilocks::Element &a, const Goldilocks::Element &b) { return Goldilocks::add(a, b); }
inline Goldilocks::Element operator-(const Goldilocks::Element &a, const Goldilocks::Element &b) { return Goldilocks::sub(a, b); }
inline Goldilocks::Element operator*(const Goldilocks::Element &a, const Goldilocks::Element &b) { return Goldilocks::mul(a, b); }
inline Goldilocks::Element operator/(const Goldilocks::Element &a, const Goldilocks::Element &b) { return Goldilocks::mul(a, Goldilocks::inv(b)); }
inline Goldilocks::Element operator-(const Goldilocks::Element &a) { return Goldilocks::neg(a); }
#endif
