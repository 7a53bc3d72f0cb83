// starks_capi.cpp -> libmi_starks.so: class Starks (host/starks.hpp) behind a C ABI, for callers that are not C++ (bench_starks.py,
// the tests): build a Starks over in-memory or on-disk constants, register the constraint programs' tables, run genProof, read the
// proof as zkin / proof JSON text, the phase times, the free-HBM low-water mark, and -- for checks after the fact -- any range of
// the device image of the polynomial area.  Errors follow the host classes: message on stderr, exit(-1) (the reference's convention).
#include <cstring>
#include <chrono>
#include <dlfcn.h>
#include <fstream>
#include <string>
#include <vector>
#include "starks.hpp"
#include "proof2zkinStark.hpp"

namespace {
// A Steps object over registered tables: the batched forms run them on the device image; the per-row forms are not generated code
// here and do nothing.
class TableSteps : public Steps
{
public:
    struct T { std::vector<uint64_t> ops, args; };
    T t[5]; // step2prev, step3prev, step3, step42ns, step52ns
    static int slot(int step)
    {
        switch (step) {
        case MI_CHELPERS_STEP2PREV: return 0;
        case MI_CHELPERS_STEP3PREV: return 1;
        case MI_CHELPERS_STEP3: return 2;
        case MI_CHELPERS_STEP42NS: return 3;
        case MI_CHELPERS_STEP52NS: return 4;
        }
        return -1;
    }
    void run(int step, StepsParams &params, uint64_t nrows)
    {
        T &x = t[slot(step)];
        if (x.ops.empty()) return; // a STARK without that stage's expressions
        mi::runChelpersStep(step, x.ops.data(), x.ops.size(), x.args.data(), x.args.size(), params, nrows);
    }
#define ROW(s) void s##_first(StepsParams &, uint64_t) override {} void s##_i(StepsParams &, uint64_t) override {} void s##_last(StepsParams &, uint64_t) override {}
    ROW(step2prev) ROW(step3prev) ROW(step3) ROW(step42ns) ROW(step52ns)
#undef ROW
#define BATCH(s, ID) void s##_parser_first_avx(StepsParams &p, uint64_t n, uint64_t) override { run(ID, p, n); } \
                     void s##_parser_first_avx512(StepsParams &p, uint64_t n, uint64_t) override { run(ID, p, n); }
    BATCH(step2prev, MI_CHELPERS_STEP2PREV) BATCH(step3prev, MI_CHELPERS_STEP3PREV) BATCH(step3, MI_CHELPERS_STEP3)
    BATCH(step42ns, MI_CHELPERS_STEP42NS) BATCH(step52ns, MI_CHELPERS_STEP52NS)
#undef BATCH
};
struct Handle {
    Config config;
    Starks *starks = nullptr;
    TableSteps steps;
    std::string zkin, proof;
    double genproofMs = 0, jsonMs = 0;
    Steps *external = nullptr; // a Steps class with per-row code out of a shared library (mis_load_steps): used when nrowsStepBatch is 1
};
} // namespace

extern "C" {
void *mis_create(const char *starkinfo_json, void *const_pols, void *const_tree, void *pAddress)
{
    Handle *h = new Handle();
    StarkInfo info(h->config, starkinfo_json);
    h->starks = new Starks(h->config, info, const_pols, const_tree, pAddress);
    return h;
}
void *mis_create_files(const char *starkinfo_json, const char *const_pols_file, const char *const_tree_file, void *pAddress)
{
    Handle *h = new Handle();
    h->starks = new Starks(h->config, {const_pols_file, false, const_tree_file, starkinfo_json}, pAddress);
    return h;
}
void mis_destroy(void *hv)
{
    Handle *h = (Handle *)hv;
    if (!h) return;
    delete h->starks;
    delete h;
}
int mis_set_tables(void *hv, int step, const uint64_t *ops, uint64_t nops, const uint64_t *args, uint64_t nargs)
{
    Handle *h = (Handle *)hv;
    const int s = TableSteps::slot(step);
    if (!h || s < 0) return -1;
    // the compiled program of this step is cached under the tables' ADDRESS (host/chelpers_steps.hpp), and assign() below keeps the vector's
    // buffer when the new tables fit: forget it, or a later proof would run the old program without any error
    h->starks->forgetProgram(step, h->steps.t[s].ops.data());
    h->steps.t[s].ops.assign(ops, ops + nops);
    h->steps.t[s].args.assign(args, args + nargs);
    return 0;
}
// a Steps class compiled elsewhere (generated per-row C++ against these headers): the library exports `Steps *mi_make_steps()`.  It must see
// THIS library's field-class recorder hook (an inline thread_local of goldilocks_base_field.hpp): load libmi_starks.so with RTLD_GLOBAL.
int mis_load_steps(void *hv, const char *so_path)
{
    Handle *h = (Handle *)hv;
    void *lib = dlopen(so_path, RTLD_NOW | RTLD_GLOBAL);
    if (!lib) { std::fprintf(stderr, "mis_load_steps: %s\n", dlerror()); return -1; }
    typedef Steps *(*Make)();
    Make make = (Make)dlsym(lib, "mi_make_steps");
    if (!make) { std::fprintf(stderr, "mis_load_steps: no mi_make_steps in %s\n", so_path); return -1; }
    h->external = make();
    return h->external ? 0 : -1;
}
uint64_t mis_hbm_plan_bytes(void *hv) { return ((Handle *)hv)->starks->hbmPlanBytes(); }
// genProof as prover.cpp:541-552; the JSON texts are kept in the handle (mis_zkin / mis_proof) and optionally written to files
int mis_gen_proof(void *hv, const uint64_t *publics, uint64_t nrowsStepBatch, const char *zkin_path, const char *proof_path)
{
    Handle *h = (Handle *)hv;
    Starks *st = h->starks;
    st->nrowsStepBatch = nrowsStepBatch;
    std::vector<Goldilocks::Element> pub(st->starkInfo.nPublics);
    for (size_t i = 0; i < pub.size(); i++) pub[i] = Goldilocks::fromU64(publics[i]);
    const uint64_t polBits = st->starkInfo.starkStruct.steps[st->starkInfo.starkStruct.steps.size() - 1].nBits;
    FRIProof fproof((1 << polBits), FIELD_EXTENSION, st->starkInfo.starkStruct.steps.size(), st->starkInfo.evMap.size(), st->starkInfo.nPublics);
    const auto t0 = std::chrono::steady_clock::now();
    st->genProof(fproof, pub.data(), nrowsStepBatch == 1 && h->external ? h->external : &h->steps);
    mi::check(mi_ctx_sync(mi::ctx()), "mis_gen_proof (sync)");
    const auto t1 = std::chrono::steady_clock::now();
    fproof.publics = pub;
    h->zkin = proof2zkinStark(fproof, true);
    h->proof = fproof.proofs.proof2json();
    h->genproofMs = std::chrono::duration<double, std::milli>(t1 - t0).count();                           // Starks::genProof alone (prover.cpp:544)
    h->jsonMs = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t1).count(); // proof2json + proof2zkinStark (prover.cpp:549-552)
    if (zkin_path && zkin_path[0]) std::ofstream(zkin_path) << h->zkin;
    if (proof_path && proof_path[0]) std::ofstream(proof_path) << h->proof;
    return 0;
}
void mis_last_wall_ms(void *hv, double out[2]) { out[0] = ((Handle *)hv)->genproofMs; out[1] = ((Handle *)hv)->jsonMs; }
const char *mis_zkin(void *hv) { return ((Handle *)hv)->zkin.c_str(); }
const char *mis_proof(void *hv) { return ((Handle *)hv)->proof.c_str(); }
// elements [offset, offset + n) of the device image of the polynomial area (StarkInfo::mapOffsets; trees follow at mapTotalN)
int mis_peek(void *hv, uint64_t offset, uint64_t n, uint64_t *out)
{
    Handle *h = (Handle *)hv;
    h->starks->peekImage(offset, n, out);
    return 0;
}
void mis_late_offsets(void *hv, uint64_t out[3])
{
    for (int i = 0; i < 3; i++) out[i] = ((Handle *)hv)->starks->lateOffsets[i];
}
void mis_phase_timer(int enable) { mi::phaseTimer().enabled = enable != 0; mi::phaseTimer().minFree = ~0ULL; }
// "NAME ms\n" per phase of the last genProof, into buf; returns the length needed
uint64_t mis_phase_times(char *buf, uint64_t cap)
{
    std::string s;
    for (auto &p : mi::phaseTimes()) s += p.first + " " + std::to_string(p.second) + "\n";
    if (buf && cap) { std::strncpy(buf, s.c_str(), cap - 1); buf[cap - 1] = 0; }
    return s.size() + 1;
}
uint64_t mis_min_free_bytes(void) { return mi::phaseTimer().minFree; }
// bytes of device memory under the process's image arena (the whole plan when dense; what the proofs so far backed when it is an
// address range: several devices, host/starks.hpp mi::Arena) and whether it is sparse
uint64_t mis_image_backed_bytes(int *sparse)
{
    if (sparse) *sparse = mi::arena().sparse ? 1 : 0;
    return mi::arena().base ? mi::arena().backedBytes() : 0;
}
// MI_STARK_DEVICES: the shards and what the driver said about direct access between their devices (mi_multi_peer_access); returns the
// number of shards, 0 when the proof runs on one device
int mis_peer_access(int *matrix, char *warning, uint64_t warning_cap, int *indirect_pairs)
{
    mi_multi *mm = mi::multi();
    if (!mm) return 0;
    const int bad = mi_multi_peer_access(mm, matrix, warning, warning_cap);
    if (indirect_pairs) *indirect_pairs = bad;
    return mi_multi_shards(mm);
}
}
