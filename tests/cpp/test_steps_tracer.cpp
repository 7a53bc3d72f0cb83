// test_steps_tracer.cpp -- host/steps_tracer.hpp against the code it records.
//
// A Steps class with generated per-row code (the reference's recursive1 / recursive2 / c12a chelpers where /root/reference is present,
// tests/gen_steps_cpp.py's output everywhere) is linked in as STEPS_CLASS.  For each of the five steps: the `_first` function is
// recorded (rows 0 and n - 1), the recorded program is translated (mi_chelpers_compile_micro, no GPU) and run by the library's
// host executors over a handful of rows of random polynomial memory; then the function ITSELF is run at those rows; everything either
// wrote -- polynomials the step stores, q_2ns, f_2ns -- must agree word for word.  So the checker here is the recorded code itself.
//
//     usage: test_steps_tracer <layout file>     layout: nBits nBitsExt nConst nPublics nEvals cols[11 sections in eSection order]
#include <cstdio>
#include <sys/mman.h>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <random>
#include <set>
#include <string>
#include <vector>
#include "goldilocks_cubic_extension.hpp"
#include "polinomial.hpp"
#include "zhInv.hpp"
#include "constant_pols_starks.hpp"
#include "steps.hpp"
#include "steps_tracer.hpp"
#include STEPS_HEADER

#ifdef MI_TEST_WITH_TABLES
extern "C" int mi_test_tables(int which, const uint64_t **ops, uint64_t *nops, const uint64_t **args, uint64_t *nargs); // tests/cpp/test_steps_tables.cpp
#endif
static std::mt19937_64 rng(20260);
static uint64_t fe() { return rng() % GOLDILOCKS_PRIME; }

int main(int argc, char **argv)
{
    if (argc < 2) { std::fprintf(stderr, "usage: %s <layout file>\n", argv[0]); return 2; }
    std::ifstream lf(argv[1]);
    uint64_t nBits, nBitsExt, nConst, nPublics, nEvals, cols[11];
    lf >> nBits >> nBitsExt >> nConst >> nPublics >> nEvals;
    for (int i = 0; i < 11; i++) lf >> cols[i];
    if (!lf) { std::fprintf(stderr, "bad layout file\n"); return 2; }
    const uint64_t N = 1ULL << nBits, NExt = 1ULL << nBitsExt, W = 40;
    uint64_t off[12];
    off[0] = 0;
    for (int i = 0; i < 11; i++) off[i + 1] = off[i] + cols[i] * (i < 5 ? N : NExt);
    // the area and the tables: calloc'ed (pages appear when touched), random in three windows of rows, which is where the test looks
    // (address space without commitment: the zkEVM's area is 254 GB, of which a few hundred pages are touched)
    auto reserve = [](uint64_t words) {
        void *p = mmap(nullptr, std::max<uint64_t>(words, 1) * 8, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        return p == MAP_FAILED ? nullptr : (Goldilocks::Element *)p;
    };
    Goldilocks::Element *mem = reserve(off[11]), *cN = reserve(nConst * N), *c2 = reserve(nConst * NExt);
    if (!mem || !cN || !c2) { std::fprintf(stderr, "cannot reserve the address space\n"); return 2; }
    Polinomial challenges(8, 3), evals(std::max<uint64_t>(nEvals, 1), 3), x_n(N, 1), x_2ns(NExt, 1), xd(NExt, 3), xdw(NExt, 3);
    std::vector<Goldilocks::Element> publics(std::max<uint64_t>(nPublics, 1));
    auto windows = [&](uint64_t n, auto f) {
        const uint64_t starts[3] = {0, n / 2, n - std::min(W, n)};
        for (uint64_t s : starts)
            for (uint64_t r = s; r < std::min(n, s + W); r++) f(r);
    };
    for (int s = 0; s < 11; s++)
        windows(s < 5 ? N : NExt, [&](uint64_t r) { for (uint64_t c = 0; c < cols[s]; c++) mem[off[s] + r * cols[s] + c].fe = fe(); });
    windows(N, [&](uint64_t r) { for (uint64_t c = 0; c < nConst; c++) cN[r * nConst + c].fe = fe(); x_n[r][0].fe = fe(); });
    windows(NExt, [&](uint64_t r) {
        for (uint64_t c = 0; c < nConst; c++) c2[r * nConst + c].fe = fe();
        x_2ns[r][0].fe = fe();
        for (int d = 0; d < 3; d++) { xd[r][d].fe = fe(); xdw[r][d].fe = fe(); }
    });
    for (uint64_t k = 0; k < 8; k++) for (int d = 0; d < 3; d++) challenges[k][d].fe = fe();
    for (uint64_t k = 0; k < nEvals; k++) for (int d = 0; d < 3; d++) evals[k][d].fe = fe();
    for (auto &p : publics) p.fe = fe();
    ConstantPolsStarks cpN(cN, N, nConst), cp2(c2, NExt, nConst);
    ZhInv zi(nBits, nBitsExt);
    StepsParams params = {mem, &cpN, &cp2, challenges, x_n, x_2ns, zi, evals, xd, xdw, publics.data(), mem + off[9], mem + off[10]};
    STEPS_CLASS stepsObj;
    Steps *steps = &stepsObj;
    std::vector<uint64_t> zh(NExt / N);
    for (uint64_t k = 0; k < zh.size(); k++) zh[k] = Goldilocks::toU64(zi.zhInv(k));

    static const int ids[5] = {MI_CHELPERS_STEP2PREV, MI_CHELPERS_STEP3PREV, MI_CHELPERS_STEP3, MI_CHELPERS_STEP42NS, MI_CHELPERS_STEP52NS};
    static const char *names[5] = {"step2prev", "step3prev", "step3", "step42ns", "step52ns"};
    int bad = 0;
    const char *skip = std::getenv("MI_TEST_SKIP_STEPS"); // e.g. "3": a step whose per-row file the tree does not have
    for (int which = 0; which < 5; which++) {
        if (skip && std::strchr(skip, '0' + which)) { std::printf("%s: skipped\n", names[which]); continue; }
        const bool base = which <= 2;
        const uint64_t n = base ? N : NExt;
        mi::TraceLayout L;
        L.step = ids[which];
        L.rows = n;
        L.pols = (const uint64_t *)mem;
        for (int s = 0; s < 11; s++) L.secs.push_back({off[s], cols[s], s < 5 ? N : NExt});
        L.qOffset = off[9]; L.fOffset = off[10];
        L.constPols = (const uint64_t *)(base ? cN : c2); L.nConst = nConst;
        L.chal = (const uint64_t *)challenges.address(); L.nChal = 8;
        L.evals = (const uint64_t *)evals.address(); L.nEvals = nEvals;
        L.pub = (const uint64_t *)publics.data(); L.nPub = nPublics;
        L.x = (const uint64_t *)(base ? x_n : x_2ns).address(); L.xStride = 1;
        if (!base) { L.xd = (const uint64_t *)xd.address(); L.xdw = (const uint64_t *)xdw.address(); }
        auto call = [&](uint64_t i) {
            switch (which) {
            case 0: steps->step2prev_first(params, i); break;
            case 1: steps->step3prev_first(params, i); break;
            case 2: steps->step3_first(params, i); break;
            case 3: steps->step42ns_first(params, i); break;
            default: steps->step52ns_first(params, i); break;
            }
        };
        std::vector<mi_chelpers_microop> mops;
        std::string err;
        if (!mi::traceStep(L, call, [&](uint64_t i) { return Goldilocks::toU64(zi.zhInv(i)); }, mops, err)) {
            // a step that computes nothing is fine for the base-domain steps only
            std::printf("%s: TRACE FAILED: %s\n", names[which], err.c_str());
            bad++;
            continue;
        }
        if (mops.empty()) { std::printf("%s: empty\n", names[which]); continue; }
        // the sections the product hands to the translator (chelpers_steps.hpp: stepSections)
        std::vector<mi_chelpers_section> secs;
        const int baseSecs[4] = {0, 1, 2, 4}, extSecs[4] = {5, 6, 7, 8};
        for (int k = 0; k < (base || which == 4 ? 4 : 3); k++) {
            const int s = base ? baseSecs[k] : extSecs[k];
            if (cols[s]) secs.push_back({off[s], cols[s], n});
        }
        mi_chelpers_prog *prog = nullptr;
        if (mi_chelpers_compile_micro(nullptr, &prog, ids[which], mops.data(), mops.size(), secs.data(), secs.size(), nConst, n) != 0) {
            std::printf("%s: TRANSLATE FAILED: %s\n", names[which], mi_last_error());
            bad++;
            continue;
        }
        // where the program writes
        struct Out { uint64_t off, shift, stride; int dim; };
        std::vector<Out> outs;
        for (const mi_chelpers_microop &u : mops) {
            if (u.cls == MI_CHP_STOREP) outs.push_back({u.b.v[0], u.b.kind == MI_CHP_DPOLS ? u.b.v[1] : 0, u.b.kind == MI_CHP_DPOLS ? u.b.v[3] : u.b.v[1], u.a.kind == MI_CHP_T3 ? 3 : 1});
            if (u.cls == MI_CHP_STOREQ) outs.push_back({off[9], 0, 3, 3});
            if (u.cls == MI_CHP_STOREF) outs.push_back({off[10], 0, 3, 3});
        }
        std::vector<uint64_t> rows = {0, 1, 2, 3, n / 2, n / 2 + 1, n - 2, n - 1};
        if (n < 8) rows = {0, n - 1};
        auto snapshot = [&](bool poison) { // read everything first, poison afterwards: a step may store one element from two rows (t at row i + 1 = t' at row i)
            std::vector<uint64_t> v;
            for (int pass = 0; pass < (poison ? 2 : 1); pass++) {
                size_t k = 0;
                for (uint64_t r : rows)
                    for (const Out &o : outs)
                        for (int d = 0; d < o.dim; d++, k++) {
                            Goldilocks::Element &e = mem[o.off + ((r + o.shift) % n) * o.stride + d];
                            if (pass == 0) v.push_back(e.fe);
                            else e.fe = 0xDEAD0000ULL + k;
                        }
            }
            return v;
        };
        mi_chelpers_params hp = {};
        hp.pols = (uint64_t *)mem;
        hp.const_pols = (const uint64_t *)(base ? cN : c2); hp.n_const = nConst;
        hp.challenges = (const uint64_t *)challenges.address(); hp.n_challenges = 8;
        hp.publics = (const uint64_t *)publics.data(); hp.n_publics = nPublics;
        hp.x = (const uint64_t *)(base ? x_n : x_2ns).address(); hp.x_stride = 1;
        hp.zhinv = zh.data(); hp.n_zhinv = zh.size();
        hp.q = (uint64_t *)mem + off[9];
        hp.evals = (const uint64_t *)evals.address(); hp.n_evals = nEvals;
        hp.xdiv = (const uint64_t *)xd.address(); hp.xdivw = (const uint64_t *)xdw.address();
        hp.f = (uint64_t *)mem + off[10];
        snapshot(true);
        int rc = mi_dbg_host_chelpers_run(prog, &hp, rows.data(), rows.size());
        if (rc != 0) { std::printf("%s: HOST EXECUTOR FAILED: %s\n", names[which], mi_last_error()); bad++; mi_chelpers_free(nullptr, prog); continue; }
        const std::vector<uint64_t> a = snapshot(true);
        rc = mi_dbg_host_chelpers_run_lowered(prog, &hp, rows.data(), rows.size(), 0);
        if (rc != 0) { std::printf("%s: LOWERED EXECUTOR FAILED: %s\n", names[which], mi_last_error()); bad++; mi_chelpers_free(nullptr, prog); continue; }
        const std::vector<uint64_t> l = snapshot(true);
        std::vector<uint64_t> t;
#ifdef MI_TEST_WITH_TABLES
        { // the same step as the reference's generated TABLE, through the library's table decoder
            const uint64_t *tops, *targs;
            uint64_t tnops, tnargs;
            mi_test_tables(which, &tops, &tnops, &targs, &tnargs);
            mi_chelpers_prog *tprog = nullptr;
            if (mi_chelpers_compile(nullptr, &tprog, ids[which], tops, tnops, targs, tnargs, secs.data(), secs.size(), nConst, n) != 0 ||
                mi_dbg_host_chelpers_run(tprog, &hp, rows.data(), rows.size()) != 0) {
                std::printf("%s: TABLE PROGRAM FAILED: %s\n", names[which], mi_last_error());
                bad++;
            } else {
                t = snapshot(true);
            }
            if (tprog) mi_chelpers_free(nullptr, tprog);
        }
#endif
        for (uint64_t r : rows) call(r);
        std::vector<uint64_t> b = snapshot(false);
        for (uint64_t &w : b) w = w >= GOLDILOCKS_PRIME ? w - GOLDILOCKS_PRIME : w; // the host classes may leave a non-canonical word; the device writes canonical ones
        size_t diff = 0, diffl = 0;
        for (size_t k = 0; k < a.size(); k++) { diff += a[k] != b[k]; diffl += l[k] != b[k]; }
        uint64_t st[16] = {};
        mi_chelpers_stats(prog, st);
        std::printf("%s: %zu recorded operations (%llu after dead-code removal), %zu stores, %zu words compared, %zu differ (translated) %zu differ (lowered)\n", names[which],
                    mops.size(), (unsigned long long)st[3], outs.size(), a.size(), diff, diffl);
        if (diff || diffl || a.empty()) bad++;
        if (diff) { // say where: (row, offset, shift, stride, dim) of the first few stores that differ
            size_t k = 0, shown = 0;
            for (uint64_t r : rows)
                for (const Out &o : outs)
                    for (int d = 0; d < o.dim; d++, k++)
                        if (a[k] != b[k] && shown < 6) { std::printf("    differs: row %llu store (offset %llu, shift %llu, stride %llu, dim %d) word %d\n", (unsigned long long)r, (unsigned long long)o.off, (unsigned long long)o.shift, (unsigned long long)o.stride, o.dim, d); shown++; }
        }
#ifdef MI_TEST_WITH_TABLES
        size_t difft = t.size() == b.size() ? 0 : b.size();
        for (size_t k = 0; k < t.size() && k < b.size(); k++) difft += t[k] != b[k];
        std::printf("%s: the reference's TABLE for this step, decoded and run on the same rows: %zu words compared, %zu differ\n", names[which], t.size(), difft);
        if (difft || t.empty()) bad++;
#endif
        mi_chelpers_free(nullptr, prog);
    }
    std::printf(bad ? "FAIL\n" : "OK\n");
    return bad ? 1 : 0;
}
