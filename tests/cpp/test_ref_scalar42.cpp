// test_ref_scalar42.cpp -- only where the reference tree is present (tests/test_reference_scalar_interpreter.py builds it).
//
// The reference's SCALAR interpreter of the step42ns table, ZkevmSteps::step42ns_parser_first (zkevm.chelpers.step42ns.parser.cpp:762-1441):
// its text is spliced in UNCHANGED at build time (MI_REF_SCALAR42_INC: the lines of that function, cut out of the reference file into
// the test's temporary directory -- never stored in this repo) and compiled against tests/cpp/batch_helpers_test_only.hpp.  It runs the
// zkEVM's REAL step42ns program over the first rows of a sparse 254 GB map beside
//   (a) the oracle's restatement of the same interpreter (glo_chelpers_step42ns, oracle/chelpers_oracle.c) and
//   (b) the product's table decoder (mi_chelpers_compile -> the translated program on the library's host executor, and the LOWERED
//       program the native backend compiles),
// and all must store the same q_2ns.  The table handed to the reference function has the AVX numbering's fusions rewritten to the
// scalar function's own (84 / 85 / 86 -> 110 / 111 / 112; 87 and 88, which the scalar function lacks, into their primitive opcodes -- the
// Python side does that, tests/chelpers_programs.FUSED, itself checked against the AVX function's text); (a) and (b) get the table as
// the reference ships it.  Which opcodes ran is printed, so that the test can assert that all 84 primitive cases were exercised.
//
//     usage: test_ref_scalar42 <ops.bin> <args.bin> <ops_for_the_scalar_function.bin> <nrows>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <cassert>
#include <random>
#include <set>
#include <string>
#include <vector>
#include <sys/mman.h>
#include "goldilocks_cubic_extension.hpp"
#include "polinomial.hpp"
#include "zhInv.hpp"
#include "constant_pols_starks.hpp"
#include "steps.hpp"
#include "../../include/mi_stark.h"
#include "../../oracle/gl_oracle.h"
#include "batch_helpers_test_only.hpp"

static uint64_t *op42, *args42, g_nops, g_nargs;
struct RefScalar42 { void step42ns_parser_first(StepsParams &params, uint64_t nrows, uint64_t nrowsBatch); };
#define NOPS_ g_nops
#define NARGS_ g_nargs
#define AVX_SIZE_ 4
#define ZkevmSteps RefScalar42
#define Goldilocks GoldilocksB
#define Goldilocks3 Goldilocks3B
using namespace std;
#include MI_REF_SCALAR42_INC
#undef Goldilocks
#undef Goldilocks3
#undef ZkevmSteps

static std::vector<uint64_t> slurp64(const char *path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) { std::fprintf(stderr, "cannot open %s\n", path); std::exit(2); }
    std::vector<uint64_t> v((size_t)f.tellg() / 8);
    f.seekg(0);
    f.read((char *)v.data(), v.size() * 8);
    return v;
}

int main(int argc, char **argv)
{
    if (argc < 5) { std::fprintf(stderr, "usage: %s ops args ops_scalar nrows\n", argv[0]); return 2; }
    std::vector<uint64_t> ops = slurp64(argv[1]), args = slurp64(argv[2]), opsScalar = slurp64(argv[3]);
    const uint64_t nrows = std::strtoull(argv[4], nullptr, 10);
    // the zkEVM's map: N = 2^23, cm1 665 | cm2 128 | cm3 371 | cm4 6 | tmpExp 265 | the same extended | q 3 | f 3; 218 constants
    const uint64_t nBits = 23, nBitsExt = 24, N = 1ULL << nBits, NExt = 1ULL << nBitsExt, nConst = 218, nPublics = 48;
    const uint64_t cols[11] = {665, 128, 371, 6, 265, 665, 128, 371, 6, 3, 3};
    uint64_t off[12];
    off[0] = 0;
    for (int i = 0; i < 11; i++) off[i + 1] = off[i] + cols[i] * (i < 5 ? N : NExt);
    auto reserve = [](uint64_t words) {
        void *p = mmap(nullptr, words * 8, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        return p == MAP_FAILED ? nullptr : (Goldilocks::Element *)p;
    };
    Goldilocks::Element *mem = reserve(off[11]), *c2 = reserve(nConst * NExt), *cN = reserve(16);
    if (!mem || !c2 || !cN) { std::fprintf(stderr, "cannot reserve the address space\n"); return 2; }
    std::mt19937_64 rng(4242);
    auto fe = [&]() { return rng() % GOLDILOCKS_PRIME; };
    // rows the first `nrows` rows can reach: themselves, rows shifted forward, and rows shifted "backward" (a shift of NExt - k wraps to the end)
    const uint64_t W = nrows + 64;
    Polinomial challenges(8, 3), evals(1, 3), x_n(1, 1), x_2ns(NExt, 1), xd(1, 3), xdw(1, 3);
    for (uint64_t pass = 0; pass < 2; pass++)
        for (uint64_t k = 0; k < W; k++) {
            const uint64_t r = pass == 0 ? k : NExt - 1 - k;
            for (int s = 5; s < 9; s++) for (uint64_t c = 0; c < cols[s]; c++) mem[off[s] + r * cols[s] + c].fe = fe();
            for (uint64_t c = 0; c < nConst; c++) c2[r * nConst + c].fe = fe();
            x_2ns[r][0].fe = fe();
        }
    for (uint64_t k = 0; k < 8; k++) for (int d = 0; d < 3; d++) challenges[k][d].fe = fe();
    std::vector<Goldilocks::Element> publics(nPublics);
    for (auto &p : publics) p.fe = fe();
    ConstantPolsStarks cpN(cN, 1, nConst), cp2(c2, NExt, nConst);
    ZhInv zi(nBits, nBitsExt);
    StepsParams params = {mem, &cpN, &cp2, challenges, x_n, x_2ns, zi, evals, xd, xdw, publics.data(), mem + off[9], mem + off[10]};
    std::vector<uint64_t> zh(NExt / N);
    for (uint64_t k = 0; k < zh.size(); k++) zh[k] = Goldilocks::toU64(zi.zhInv(k));
    auto takeQ = [&]() {
        std::vector<uint64_t> q(nrows * 3);
        for (uint64_t k = 0; k < nrows * 3; k++) { q[k] = Goldilocks::toU64(mem[off[9] + k]); mem[off[9] + k].fe = 0xDEAD0000ULL + k; }
        return q;
    };
    takeQ();

    // ---- the reference's scalar interpreter, unchanged
    op42 = opsScalar.data(); g_nops = opsScalar.size(); args42 = args.data(); g_nargs = args.size();
    RefScalar42 ref;
    ref.step42ns_parser_first(params, nrows, AVX_SIZE_);
    const std::vector<uint64_t> qRef = takeQ();
    std::set<uint64_t> ran(opsScalar.begin(), opsScalar.end());
    std::printf("reference scalar interpreter: %zu opcodes over %llu rows; distinct cases run:", opsScalar.size(), (unsigned long long)nrows);
    for (uint64_t o : ran) std::printf(" %llu", (unsigned long long)o);
    std::printf("\n");

    // ---- (a) the oracle's restatement, on the table as shipped
    if (glo_chelpers_step42ns(ops.data(), ops.size(), args.data(), args.size(), (const uint64_t *)mem, (const uint64_t *)c2, nConst, (const uint64_t *)challenges.address(),
                              (const uint64_t *)publics.data(), (const uint64_t *)x_2ns.address(), 1, zh.data(), zh.size(), (uint64_t *)(mem + off[9]), 0, nrows) != 0) {
        std::printf("FAIL: the oracle refuses the table\n");
        return 1;
    }
    const std::vector<uint64_t> qOracle = takeQ();

    // ---- (b) the product's decoder: translated program, then the lowered program, on the library's host executors
    std::vector<mi_chelpers_section> secs;
    for (int s = 5; s < 8; s++) secs.push_back({off[s], cols[s], NExt});
    mi_chelpers_prog *prog = nullptr;
    if (mi_chelpers_compile(nullptr, &prog, MI_CHELPERS_STEP42NS, ops.data(), ops.size(), args.data(), args.size(), secs.data(), secs.size(), nConst, NExt) != 0) {
        std::printf("FAIL: mi_chelpers_compile: %s\n", mi_last_error());
        return 1;
    }
    mi_chelpers_params hp = {};
    hp.pols = (uint64_t *)mem;
    hp.const_pols = (const uint64_t *)c2; hp.n_const = nConst;
    hp.challenges = (const uint64_t *)challenges.address(); hp.n_challenges = 8;
    hp.publics = (const uint64_t *)publics.data(); hp.n_publics = nPublics;
    hp.x = (const uint64_t *)x_2ns.address(); hp.x_stride = 1;
    hp.zhinv = zh.data(); hp.n_zhinv = zh.size();
    hp.q = (uint64_t *)mem + off[9];
    std::vector<uint64_t> rows(nrows);
    for (uint64_t r = 0; r < nrows; r++) rows[r] = r;
    if (mi_dbg_host_chelpers_run(prog, &hp, rows.data(), rows.size()) != 0) { std::printf("FAIL: host executor: %s\n", mi_last_error()); return 1; }
    const std::vector<uint64_t> qProduct = takeQ();
    if (mi_dbg_host_chelpers_run_lowered(prog, &hp, rows.data(), rows.size(), 0) != 0) { std::printf("FAIL: lowered executor: %s\n", mi_last_error()); return 1; }
    const std::vector<uint64_t> qLowered = takeQ();
    mi_chelpers_free(nullptr, prog);

    size_t dOracle = 0, dProduct = 0, dLowered = 0, zeros = 0;
    for (size_t k = 0; k < qRef.size(); k++) {
        dOracle += qRef[k] != qOracle[k]; dProduct += qRef[k] != qProduct[k]; dLowered += qRef[k] != qLowered[k];
        zeros += qRef[k] == 0;
    }
    std::printf("q_2ns, %zu words: %zu differ (oracle) %zu differ (product, translated) %zu differ (product, lowered); %zu zero words\n", qRef.size(), dOracle, dProduct,
                dLowered, zeros);
    const bool ok = !dOracle && !dProduct && !dLowered && zeros < qRef.size() / 8;
    std::printf(ok ? "OK\n" : "FAIL\n");
    return ok ? 0 : 1;
}
