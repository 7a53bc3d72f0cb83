// test_starks_genproof.cpp -- the reference's caller, verbatim in shape (prover.cpp:128-132 and :541-552), on host/starks.hpp:
//
//     starkZkevm = new Starks(config, {constPols, mapConstPolsFile, constantsTree, starkInfo}, pAddress);
//     starkZkevm->nrowsStepBatch = NROWS_STEPS_;
//     ZkevmSteps zkevmSteps;
//     FRIProof fproof((1 << polBits), FIELD_EXTENSION, steps.size(), evMap.size(), nPublics);
//     starkZkevm->genProof(fproof, &publics[0], &zkevmSteps);
//     jProof = fproof.proofs.proof2json();  zkin = proof2zkinStark(jProof);
//
// for the mini STARK of tests/ministark.py: its starkinfo.json, constant polynomials, witness and generated tables are files written
// by tests/test_starks_class.py, which afterwards gives the zkin.json this writes to the independent verifier.  ZkevmSteps' batched
// entry points are DEFINED here over those tables (MI_DEFINE_PARSER_STEP: what replaces the reference's *.parser.cpp); its per-row
// forms -- generated C++ in the reference -- come in two variants (below), so that a second run with nrowsStepBatch = 1 exercises the
// traced-steps path (the per-row code recorded and run on the device) or the host-steps path, and must produce the same proof.
//
//     usage: test_starks_genproof <dir> <nrowsStepBatch> [second nrowsStepBatch]
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <string>
#include <vector>
#include "starks.hpp"
#include "zkevmSteps.hpp"
#include "chelpers_steps.hpp"
#include "build_const_tree.hpp"
#include "proof2zkinStark.hpp"
#include "../../oracle/gl_oracle.h"

static std::vector<uint64_t> slurp64(const std::string &path)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) { std::fprintf(stderr, "cannot open %s\n", path.c_str()); std::exit(2); }
    std::vector<uint64_t> v((size_t)f.tellg() / 8);
    f.seekg(0);
    f.read((char *)v.data(), v.size() * 8);
    return v;
}

// the "generated tables" (in the reference: op42[NOPS_] / args42[NARGS_] of zkevm.chelpers.step42ns.parser.hpp, ...)
struct Tables { std::vector<uint64_t> ops, args; };
static Tables t2prev, t3prev, t3, t42, t52;

MI_DEFINE_PARSER_STEP(ZkevmSteps, step2prev, _avx, MI_CHELPERS_STEP2PREV, t2prev.ops.data(), t2prev.ops.size(), t2prev.args.data(), t2prev.args.size())
MI_DEFINE_PARSER_STEP(ZkevmSteps, step3prev, _avx, MI_CHELPERS_STEP3PREV, t3prev.ops.data(), t3prev.ops.size(), t3prev.args.data(), t3prev.args.size())
MI_DEFINE_PARSER_STEP(ZkevmSteps, step3, _avx, MI_CHELPERS_STEP3, t3.ops.data(), t3.ops.size(), t3.args.data(), t3.args.size())
MI_DEFINE_PARSER_STEP(ZkevmSteps, step42ns, _avx, MI_CHELPERS_STEP42NS, t42.ops.data(), t42.ops.size(), t42.args.data(), t42.args.size())
MI_DEFINE_PARSER_STEP(ZkevmSteps, step52ns, _avx, MI_CHELPERS_STEP52NS, t52.ops.data(), t52.ops.size(), t52.args.data(), t52.args.size())
MI_FORWARD_PARSER_STEP(ZkevmSteps, step3, , _avx)
MI_FORWARD_PARSER_STEP(ZkevmSteps, step3, _avx_jump, _avx)
MI_FORWARD_PARSER_STEP(ZkevmSteps, step42ns, , _avx)
MI_FORWARD_PARSER_STEP(ZkevmSteps, step42ns, _avx_jump, _avx)
MI_FORWARD_PARSER_STEP(ZkevmSteps, step52ns, , _avx)

// per-row forms (in the reference: generated C++, e.g. recursive1.chelpers.step3.cpp).  Two variants:
//   -DMI_TEST_GENERATED_ROWS="file": the AIR's programs written out as generated per-row C++ by tests/gen_steps_cpp.py -- what
//      Starks::genProof RECORDS and runs on the device with nrowsStepBatch = 1 (host/steps_tracer.hpp);
//   otherwise: the CPU oracle's interpreters, one row at a time over params' HOST pointers -- plain C arithmetic the recorder cannot
//      follow, for the host-steps path (MI_STEPS_ON_HOST=1).
static uint64_t g_nConst, g_N, g_NExt;
#ifdef MI_TEST_GENERATED_ROWS
#include MI_TEST_GENERATED_ROWS
#else
static void rowBase(const Tables &t, StepsParams &p, uint64_t i)
{
    if (glo_chelpers_stepbase(t.ops.data(), t.ops.size(), t.args.data(), t.args.size(), (uint64_t *)p.pols, (const uint64_t *)p.pConstPols->address(), g_nConst,
                              (const uint64_t *)p.challenges.address(), (const uint64_t *)p.publicInputs, (const uint64_t *)p.x_n.address(), 1, &i, 1) != 0) std::exit(3);
}
void ZkevmSteps::step2prev_first(StepsParams &p, uint64_t i) { rowBase(t2prev, p, i); }
void ZkevmSteps::step3prev_first(StepsParams &p, uint64_t i) { rowBase(t3prev, p, i); }
void ZkevmSteps::step3_first(StepsParams &p, uint64_t i) { rowBase(t3, p, i); }
void ZkevmSteps::step42ns_first(StepsParams &p, uint64_t i)
{
    std::vector<uint64_t> zh(g_NExt / g_N);
    for (uint64_t k = 0; k < zh.size(); k++) zh[k] = Goldilocks::toU64(p.zi.zhInv(k));
    if (glo_chelpers_step42ns(t42.ops.data(), t42.ops.size(), t42.args.data(), t42.args.size(), (const uint64_t *)p.pols, (const uint64_t *)p.pConstPols2ns->address(),
                              g_nConst, (const uint64_t *)p.challenges.address(), (const uint64_t *)p.publicInputs, (const uint64_t *)p.x_2ns.address(), 1, zh.data(),
                              zh.size(), (uint64_t *)p.q_2ns, i, 1) != 0) std::exit(3);
}
void ZkevmSteps::step52ns_first(StepsParams &p, uint64_t i)
{
    if (glo_chelpers_step52ns(t52.ops.data(), t52.ops.size(), t52.args.data(), t52.args.size(), (const uint64_t *)p.pols, (const uint64_t *)p.pConstPols2ns->address(),
                              g_nConst, (const uint64_t *)p.challenges.address(), (const uint64_t *)p.evals.address(), (const uint64_t *)p.xDivXSubXi.address(),
                              (const uint64_t *)p.xDivXSubWXi.address(), (uint64_t *)p.f_2ns, i, 1) != 0) std::exit(3);
}
#define UNUSED_ROW(s) void ZkevmSteps::s##_i(StepsParams &, uint64_t) {} void ZkevmSteps::s##_last(StepsParams &, uint64_t) {}
UNUSED_ROW(step2prev) UNUSED_ROW(step3prev) UNUSED_ROW(step3) UNUSED_ROW(step42ns) UNUSED_ROW(step52ns)
#endif

int main(int argc, char **argv)
{
    if (argc < 3) { std::fprintf(stderr, "usage: %s <dir> <nrowsStepBatch> [second nrowsStepBatch]\n", argv[0]); return 2; }
    const std::string dir = argv[1];
    auto load = [&](Tables &t, const char *name) { t.ops = slurp64(dir + "/" + name + ".ops"); t.args = slurp64(dir + "/" + name + ".args"); };
    load(t2prev, "step2prev"); load(t3prev, "step3prev"); load(t3, "step3"); load(t42, "step42ns"); load(t52, "step52ns");

    // bctree (tools/starkpil/bctree): constant polynomials -> constant tree file + verification key, on the GPU
    buildConstTree(dir + "/mini.const", dir + "/mini.starkstruct.json", dir + "/mini.consttree", dir + "/mini.verkey.json");

    Config config;
    config.zkevmConstPols = dir + "/mini.const";
    config.zkevmConstantsTree = dir + "/mini.consttree";
    config.zkevmStarkInfo = dir + "/mini.starkinfo.json";
    StarkInfo _starkInfo(config, config.zkevmStarkInfo); // prover.cpp:95-99: the size of the polynomial area
    const uint64_t polsSize = _starkInfo.mapTotalN * sizeof(Goldilocks::Element) +
                              _starkInfo.mapSectionsN.section[eSection::cm3_2ns] * (1ULL << _starkInfo.starkStruct.nBitsExt) * sizeof(Goldilocks::Element);
    void *pAddress = calloc(polsSize, 1);
    const std::vector<uint64_t> witness = slurp64(dir + "/mini.commit"), pub = slurp64(dir + "/mini.publics");
    std::memcpy(pAddress, witness.data(), witness.size() * 8); // the executor's output: cm1_n at offset 0
    g_nConst = _starkInfo.nConstants; g_N = 1ULL << _starkInfo.starkStruct.nBits; g_NExt = 1ULL << _starkInfo.starkStruct.nBitsExt;

    Starks *starkZkevm = new Starks(config, {config.zkevmConstPols, config.mapConstPolsFile, config.zkevmConstantsTree, config.zkevmStarkInfo}, pAddress);
    std::printf("hbm plan: %.3f MB\n", starkZkevm->hbmPlanBytes() / 1e6);
    std::string first;
    for (int run = 2; run < argc; run++) {
        starkZkevm->nrowsStepBatch = std::strtoull(argv[run], nullptr, 10);
        std::vector<Goldilocks::Element> publics(pub.size());
        for (size_t i = 0; i < pub.size(); i++) publics[i] = Goldilocks::fromU64(pub[i]);
        ZkevmSteps zkevmSteps;
        uint64_t polBits = starkZkevm->starkInfo.starkStruct.steps[starkZkevm->starkInfo.starkStruct.steps.size() - 1].nBits;
        FRIProof fproof((1 << polBits), FIELD_EXTENSION, starkZkevm->starkInfo.starkStruct.steps.size(), starkZkevm->starkInfo.evMap.size(), starkZkevm->starkInfo.nPublics);
        starkZkevm->genProof(fproof, &publics[0], &zkevmSteps);
        fproof.publics = publics;
        const std::string zkin = proof2zkinStark(fproof, true), jProof = fproof.proofs.proof2json();
        const std::string tag = run == 2 ? "" : "." + std::string(argv[run]);
        std::ofstream(dir + "/zkin" + tag + ".json") << zkin;
        std::ofstream(dir + "/proof" + tag + ".json") << jProof;
        std::printf("genProof(nrowsStepBatch=%s) done: zkin %zu bytes\n", argv[run], zkin.size());
        if (run == 2) first = zkin;
        else if (zkin != first) { std::printf("FAIL: the proof with nrowsStepBatch=%s differs from the first\n", argv[run]); return 1; }
    }
    // the witness area is the caller's: device steps never write it (host steps do, like the reference)
    delete starkZkevm;
    free(pAddress);
    std::printf("ALL OK\n");
    return 0;
}
