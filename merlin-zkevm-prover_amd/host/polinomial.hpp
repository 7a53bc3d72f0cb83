// polinomial.hpp -- the strided view the reference passes around (polinomial.hpp:11-744): element i of a
// polynomial of dimension dim lives at address[i * offset].  Per-element helpers are scalar host code as in the
// reference; the bulk operations on the hot path (batchInverse / batchInverseParallel) run on the GPU.
#ifndef POLINOMIAL
#define POLINOMIAL
#include <cassert>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "goldilocks_base_field.hpp"
#include "goldilocks_cubic_extension.hpp"
#include "mi_runtime.hpp"

class Polinomial
{
    Goldilocks::Element *_pAddress = NULL;
    uint64_t _degree = 0, _dim = 0, _offset = 0;
    bool _allocated = false;
    std::string _name = "";

public:
    Polinomial() {}
    Polinomial(void *pAddress, uint64_t degree, uint64_t dim, uint64_t offset = 0, std::string name = "")
        : _pAddress((Goldilocks::Element *)pAddress), _degree(degree), _dim(dim), _offset(offset), _name(name){};
    Polinomial(uint64_t degree, uint64_t dim, std::string name = "") : _degree(degree), _dim(dim), _name(name)
    {
        if (degree == 0 || dim == 0) return;
        _pAddress = (Goldilocks::Element *)calloc(_degree * _dim, sizeof(Goldilocks::Element));
        if (_pAddress == NULL) { std::fprintf(stderr, "Polinomial::Polinomial() failed allocating polinomial\n"); std::exit(-1); }
        _offset = _dim;
        _allocated = true;
    };
    Polinomial(const Polinomial &) = delete;
    Polinomial &operator=(const Polinomial &) = delete;
    ~Polinomial() { if (_allocated) free(_pAddress); };
    void potConstruct(Goldilocks::Element *pAddress, uint64_t degree, uint64_t dim, uint64_t offset = 0)
    {
        _pAddress = pAddress; _degree = degree; _dim = dim; _offset = offset; _allocated = false;
    }
    inline Goldilocks::Element *address(void) { return _pAddress; }
    inline uint64_t degree(void) { return _degree; }
    inline uint64_t dim(void) { return _dim; }
    inline uint64_t length(void) { return _degree * _dim; }
    inline uint64_t size(void) { return _degree * _dim * sizeof(Goldilocks::Element); }
    inline uint64_t offset(void) { return _offset; }
    Goldilocks::Element *operator[](uint64_t i) { return &_pAddress[i * _offset]; };

    static void copyElement(Polinomial &a, uint64_t idx_a, Polinomial &b, uint64_t idx_b)
    {
        assert(a.dim() == b.dim());
        std::memcpy(a[idx_a], b[idx_b], b.dim() * sizeof(Goldilocks::Element));
    };
    static inline void addElement(Polinomial &out, uint64_t idx_out, Polinomial &in_a, uint64_t idx_a, Polinomial &in_b, uint64_t idx_b)
    {
        for (uint64_t d = 0; d < in_a.dim(); d++) out[idx_out][d] = in_a[idx_a][d] + in_b[idx_b][d];
    }
    static inline void subElement(Polinomial &out, uint64_t idx_out, Polinomial &in_a, uint64_t idx_a, Polinomial &in_b, uint64_t idx_b)
    {
        for (uint64_t d = 0; d < in_a.dim(); d++) out[idx_out][d] = in_a[idx_a][d] - in_b[idx_b][d];
    }
    static inline void mulElement(Polinomial &out, uint64_t idx_out, Polinomial &in_a, uint64_t idx_a, Goldilocks::Element &b)
    {
        Polinomial polB(&b, 1, 1);
        mulElement(out, idx_out, in_a, idx_a, polB, 0);
    }
    static inline void mulElement(Polinomial &out, uint64_t idx_out, Polinomial &in_a, uint64_t idx_a, Polinomial &in_b, uint64_t idx_b)
    {
        if (in_a.dim() == 1) {
            out[idx_out][0] = in_a[idx_a][0] * in_b[idx_b][0];
        } else if (in_a.dim() == 3 && in_b.dim() == 1) {
            Goldilocks::Element b = in_b[idx_b][0];
            for (int d = 0; d < 3; d++) out[idx_out][d] = in_a[idx_a][d] * b;
        } else {
            Goldilocks3::Element r;
            Goldilocks3::mul(r, *(Goldilocks3::Element *)in_a[idx_a], *(Goldilocks3::Element *)in_b[idx_b]);
            for (int d = 0; d < 3; d++) out[idx_out][d] = r[d];
        }
    };
    static inline void divElement(Polinomial &out, uint64_t idx_out, Polinomial &in_a, uint64_t idx_a, Goldilocks::Element &b)
    {
        Goldilocks::Element inv = Goldilocks::inv(b);
        mulElement(out, idx_out, in_a, idx_a, inv);
    }
    static inline void mulAddElement_adim3(Goldilocks::Element *out, Goldilocks::Element *in_a, Polinomial &in_b, uint64_t idx_b)
    {
        if (in_b.dim() == 1) {
            for (int d = 0; d < 3; d++) out[d] = out[d] + in_a[d] * in_b[idx_b][0];
        } else {
            Goldilocks3::Element r;
            Goldilocks3::mul(r, *(Goldilocks3::Element *)in_a, *(Goldilocks3::Element *)in_b[idx_b]);
            for (int d = 0; d < 3; d++) out[d] = out[d] + r[d];
        }
    }
    // polinomial.hpp:612-720 -- element-wise inverse of a dim-3, offset-3 polynomial, on the GPU; res may alias src
    static void batchInverse(Polinomial &res, Polinomial &src)
    {
        if (src.dim() == 1) { // not on the path (the reference's own batchInverse asserts dim 3 through copyElement): host scalars
            for (uint64_t i = 0; i < src.degree(); i++) res[i][0] = Goldilocks::inv(src[i][0]);
            return;
        }
        assert(src.dim() == 3 && src.offset() == 3 && res.offset() == 3);
        mi_ctx *c = mi::ctx();
        const uint64_t bytes = src.degree() * 3 * 8;
        uint64_t *d = (uint64_t *)mi_dev_alloc(c, bytes);
        if (!d) mi::fail("Polinomial::batchInverse (alloc)");
        mi::check(mi_copy_h2d(c, d, src.address(), bytes), "Polinomial::batchInverse (h2d)");
        mi::check(mi_batch_inverse3_dev(c, d, d, src.degree()), "Polinomial::batchInverse");
        mi::check(mi_copy_d2h(c, res.address(), d, bytes), "Polinomial::batchInverse (d2h)");
        mi_dev_free(c, d);
    }
    static void batchInverseParallel(Polinomial &res, Polinomial &src) { batchInverse(res, src); }

    // polinomial.hpp:230-584 -- the plookup columns.  The reference has four bodies (map-based calculateH1H2 / calculateH1H2_, the
    // hash-table _opt1 for dim 1 and _opt3 for dim 3) that produce the same h1 / h2; all of them run on the GPU here.  Host views in
    // and out (packed copies cross PCIe: with the polynomials already in HBM use mi_calculate_h1h2_dev instead, as host/starks.hpp does);
    // buffer / size_keys / size_values are the reference's hash-table scratch and are not used.  A value of f that t does not hold
    // ends the process with "number not included: w=<row>", like the reference.
    static void calculateH1H2(Polinomial &h1, Polinomial &h2, Polinomial &fPol, Polinomial &tPol)
    {
        const uint64_t n = tPol.degree(), dim = tPol.dim();
        assert(fPol.degree() == n && h1.degree() >= n && h2.degree() >= n && fPol.dim() == dim && h1.dim() == dim && h2.dim() == dim);
        if (!n) return;
        mi_ctx *c = mi::ctx();
        std::vector<uint64_t> host(4 * n * dim); // f | t | h1 | h2, packed
        for (uint64_t i = 0; i < n; i++) {
            std::memcpy(&host[i * dim], fPol[i], dim * 8);
            std::memcpy(&host[(n + i) * dim], tPol[i], dim * 8);
        }
        uint64_t *d = (uint64_t *)mi_dev_alloc(c, host.size() * 8);
        if (!d) mi::fail("Polinomial::calculateH1H2 (alloc)");
        mi::check(mi_copy_h2d(c, d, host.data(), 2 * n * dim * 8), "Polinomial::calculateH1H2 (h2d)");
        mi::check(mi_calculate_h1h2_dev(c, d + 2 * n * dim, dim, d + 3 * n * dim, dim, d, dim, d + n * dim, dim, (unsigned)dim, n),
                  "Polinomial::calculateH1H2");
        mi::check(mi_copy_d2h(c, &host[2 * n * dim], d + 2 * n * dim, 2 * n * dim * 8), "Polinomial::calculateH1H2 (d2h)");
        mi_dev_free(c, d);
        for (uint64_t i = 0; i < n; i++) {
            std::memcpy(h1[i], &host[(2 * n + i) * dim], dim * 8);
            std::memcpy(h2[i], &host[(3 * n + i) * dim], dim * 8);
        }
    }
    static void calculateH1H2_(Polinomial &h1, Polinomial &h2, Polinomial &fPol, Polinomial &tPol, uint64_t) { calculateH1H2(h1, h2, fPol, tPol); }
    static void calculateH1H2_opt1(Polinomial &h1, Polinomial &h2, Polinomial &fPol, Polinomial &tPol, uint64_t, uint64_t *, uint64_t, uint64_t)
    {
        calculateH1H2(h1, h2, fPol, tPol);
    }
    static void calculateH1H2_opt3(Polinomial &h1, Polinomial &h2, Polinomial &fPol, Polinomial &tPol, uint64_t, uint64_t *, uint64_t, uint64_t)
    {
        calculateH1H2(h1, h2, fPol, tPol);
    }
    // polinomial.hpp:586-607 -- the grand product z[0] = 1, z[i] = z[i-1] * num[i-1] / den[i-1], on the GPU (host views, packed copies)
    static void calculateZ(Polinomial &z, Polinomial &num, Polinomial &den)
    {
        const uint64_t n = num.degree();
        assert(num.dim() == 3 && den.dim() == 3 && z.dim() == 3 && den.degree() == n && z.degree() >= n);
        if (!n) return;
        mi_ctx *c = mi::ctx();
        std::vector<uint64_t> host(3 * n * 3); // num | den | z
        for (uint64_t i = 0; i < n; i++) {
            std::memcpy(&host[i * 3], num[i], 24);
            std::memcpy(&host[(n + i) * 3], den[i], 24);
        }
        uint64_t *d = (uint64_t *)mi_dev_alloc(c, host.size() * 8);
        if (!d) mi::fail("Polinomial::calculateZ (alloc)");
        mi::check(mi_copy_h2d(c, d, host.data(), 2 * n * 3 * 8), "Polinomial::calculateZ (h2d)");
        int closes = 0;
        mi::check(mi_calculate_z_dev(c, d + 6 * n, 3, d, 3, d + 3 * n, 3, n, &closes), "Polinomial::calculateZ");
        mi::check(mi_copy_d2h(c, &host[6 * n], d + 6 * n, n * 3 * 8), "Polinomial::calculateZ (d2h)");
        mi_dev_free(c, d);
        for (uint64_t i = 0; i < n; i++) std::memcpy(z[i], &host[(6 * n) + i * 3], 24);
        assert(closes); // zkassert(Goldilocks3::isOne(checkVal)) at :606
        (void)closes;
    }
};
#endif
