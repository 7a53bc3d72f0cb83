// chelpers_steps.hpp -- what stands behind Steps::step*_parser_first_avx when the polynomial area lives in HBM.
//
// In the reference a Steps object evaluates a constraint program over HOST memory: ZkevmSteps::step42ns_parser_first_avx
// (zkevm.chelpers.step42ns.parser.cpp:10-760) interprets the generated tables op42[] / args42[] over params.pols, four rows per
// AVX2 batch.  Here Starks::genProof (host/starks.hpp) keeps a device image of params.pols (a StarkMirror, registered for the
// duration of the proof), and the batched Steps entry points hand their TABLES to the library (mi_chelpers_*): translated once
// per proving key, compiled to gfx950 kernels, run over the image.  MI_DEFINE_PARSER_STEP writes such an entry point.
//
// Reference interfaces: steps.hpp:4-59 (StepsParams, Steps), zkevmSteps.hpp:10-57, call sites starks.cpp:66-90,150-210,237-258,
// 367-388.
#ifndef MI_CHELPERS_STEPS_HPP
#define MI_CHELPERS_STEPS_HPP
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <utility>
#include <vector>
#include "goldilocks_base_field.hpp"
#include "mi_runtime.hpp"
#include "steps.hpp"

namespace mi {
// Device image of one Starks' polynomial area while its genProof runs: section offsets are StarkInfo::mapOffsets, i.e. element e of
// the host area (params.pols[e]) is d_mem[e].
struct StarkMirror
{
    const void *hostPols = nullptr; // the pAddress this image mirrors (StepsParams::pols)
    uint64_t *d_mem = nullptr;
    uint64_t N = 0, NExtended = 0, nBits = 0, nBitsExt = 0, nPublics = 0, nEvals = 0;
    struct Sec { uint64_t offset, cols; };
    Sec cmN[4] = {}, cm2ns[4] = {}; // cm1_n cm2_n cm3_n tmpExp_n / cm1_2ns .. cm4_2ns
    bool tiledWitness = false;      // cm1_n lies tile-major in the image (host/starks.hpp): the base-domain steps read it in place
    bool tiledExt[4] = {};          // cm1_2ns .. cm4_2ns lie tile-major in the image: step42ns / step52ns read them in place
    bool tiledConstN = false;       // d_constN is tile-major (kept that way by the Starks for good): the base-domain steps read it in place
    bool anyTiledExt() const { return tiledExt[0] || tiledExt[1] || tiledExt[2] || tiledExt[3]; }
    uint64_t qOffset = 0, fOffset = 0;
    uint64_t *d_constN = nullptr, *d_const2ns = nullptr, nConst = 0;
    uint64_t *d_xn = nullptr, *d_x2ns = nullptr, *d_xdiv = nullptr, *d_xdivw = nullptr;
    std::vector<uint64_t> zhinv;
    // programs of this proving key, by (step, address of the opcode table): translated and compiled on first use, kept by the Starks
    std::map<std::pair<int, const void *>, mi_chelpers_prog *> *progs = nullptr;
    std::string cacheDir;
    // Row-sharded step42ns / step52ns (several devices, host/starks.hpp): shard g >= 1 evaluates rows [row0, row0 + rows) of the extended
    // domain on ITS device, over a full-height mirror of the extended sections of which the stage commits left its own rows and the halo
    // after them (mi_multi_set_row_images), its own extension of the constant polynomials, its own x_2ns and x / (x - xi) tables; the q /
    // f rows come back into this image.  The rows before rowShards[0].row0 stay with this device.  Empty: everything here.
    struct RowShard
    {
        int shard = 0;                  // index into the mi_multi
        uint64_t *d_mem = nullptr;      // VIRTUAL base: d_mem + offset is that device's copy of element `offset` (valid in the extended sections only)
        uint64_t *d_const2ns = nullptr, *d_x2ns = nullptr, *d_xdiv = nullptr, *d_xdivw = nullptr;
        uint64_t row0 = 0, rows = 0;
        std::map<std::pair<int, const void *>, mi_chelpers_prog *> *progs = nullptr; // that shard's programs (a program is loaded on one device)
    };
    std::vector<RowShard> rowShards;
    mi_multi *multi = nullptr;
    bool syncHomeFirst = false; // a row shard reads THIS device's tables (it is grouped with shard 0): what this context queued comes first
};
// one proof in flight per process (prover.cpp:187-260): the image of the running genProof
inline StarkMirror *&currentMirror()
{
    static StarkMirror *m = nullptr;
    return m;
}

inline bool isBaseStep(int step) { return step == MI_CHELPERS_STEP2PREV || step == MI_CHELPERS_STEP3PREV || step == MI_CHELPERS_STEP3; }
constexpr int MI_STEP_KEY_TRACED = 0x10000, MI_STEP_KEY_TILED = 0x20000, MI_STEP_KEY_TILED_CONST = 0x40000; // flags in the first half of a program-cache key
// a program is compiled for the layout of what it reads: one entry per layout (base-domain steps: the witness, the constants; the others: the extension)
inline int stepLayoutKey(const StarkMirror *m, int step)
{
    return ((isBaseStep(step) ? m->tiledWitness : m->anyTiledExt()) ? MI_STEP_KEY_TILED : 0) | (isBaseStep(step) && m->tiledConstN ? MI_STEP_KEY_TILED_CONST : 0);
}

inline StarkMirror *mirrorOf(StepsParams &params)
{
    StarkMirror *m = currentMirror();
    if (!m || m->hostPols != (const void *)params.pols) {
        // no silent host fallback: the interpreter this replaces does not exist here
        std::fprintf(stderr, "mi_stark: a device step was called on a polynomial area that has no device image "
                             "(these entry points run inside Starks::genProof of host/starks.hpp)\n");
        std::exit(-1);
    }
    return m;
}
// the sections a program of this step may read, as mi_chelpers_compile wants them
inline std::vector<mi_chelpers_section> stepSections(const StarkMirror *m, int step)
{
    const bool base = isBaseStep(step);
    std::vector<mi_chelpers_section> secs;
    const unsigned nsec = base ? 4 : step == MI_CHELPERS_STEP52NS ? 4 : 3;
    for (unsigned s = 0; s < nsec; s++) {
        const StarkMirror::Sec &S = base ? m->cmN[s] : m->cm2ns[s];
        if (S.cols) secs.push_back({S.offset, S.cols, base ? m->N : m->NExtended});
    }
    return secs;
}
inline void buildStepProgram(const StarkMirror *m, int step, mi_chelpers_prog *prog)
{
    if (isBaseStep(step) && m->tiledWitness && m->cmN[0].cols)
        check(mi_chelpers_set_tiled_section(prog, m->cmN[0].offset), "Steps (the witness section is tile-major)");
    if (isBaseStep(step) && m->tiledConstN) check(mi_chelpers_set_tiled_consts(prog), "Steps (the constant polynomials are tile-major)");
    if (!isBaseStep(step))
        for (unsigned s = 0; s < (step == MI_CHELPERS_STEP52NS ? 4u : 3u); s++)
            if (m->tiledExt[s] && m->cm2ns[s].cols) check(mi_chelpers_set_tiled_section(prog, m->cm2ns[s].offset), "Steps (an extended section is tile-major)");
    const char *backend = std::getenv("MI_CHELPERS_BACKEND"); // "interpreter": the extended-domain steps through the SIMT interpreter (A/B)
    if (isBaseStep(step) || !backend || std::string(backend) != "interpreter")
        check(mi_chelpers_build_native(prog, m->cacheDir.empty() ? nullptr : m->cacheDir.c_str(), 0), "Steps (compile the program)");
}
inline bool isRowShardedStep(const StarkMirror *m, int step) { return !m->rowShards.empty() && (step == MI_CHELPERS_STEP42NS || step == MI_CHELPERS_STEP52NS); }
// a translated program over rows [0, nrows) of its domain, on the device image
inline void runStepProgram(StarkMirror *m, int step, const mi_chelpers_prog *prog, StepsParams &params, uint64_t nrows)
{
    const bool base = isBaseStep(step);
    mi_chelpers_params p = {}; // every table the StepsParams of this domain has: a program reads what it reads
    p.pols = m->d_mem;
    p.const_pols = base ? m->d_constN : m->d_const2ns;
    p.n_const = m->nConst;
    p.challenges = (const uint64_t *)params.challenges.address();
    p.n_challenges = params.challenges.degree();
    p.publics = (const uint64_t *)params.publicInputs;
    p.n_publics = m->nPublics;
    p.x = base ? m->d_xn : m->d_x2ns;
    p.x_stride = 1;
    if (step == MI_CHELPERS_STEP42NS) {
        p.zhinv = m->zhinv.data();
        p.n_zhinv = m->zhinv.size();
        p.q = m->d_mem + m->qOffset;
    } else if (step == MI_CHELPERS_STEP52NS) {
        p.evals = (const uint64_t *)params.evals.address();
        p.n_evals = m->nEvals;
        p.xdiv = m->d_xdiv;
        p.xdivw = m->d_xdivw;
        p.f = m->d_mem + m->fOffset;
    }
    const uint64_t mine = (isRowShardedStep(m, step)) ? std::min<uint64_t>(nrows, m->rowShards[0].row0) : nrows;
    check(mi_chelpers_run_dev(ctx(), prog, &p, 0, mine), "Steps (run the program)");
}
// step42ns / step52ns on the other devices: shard S's rows through ITS program over ITS mirror, then its q / f rows into this device's image
inline void runRowShard(StarkMirror *m, int step, const StarkMirror::RowShard &S, const mi_chelpers_prog *prog, StepsParams &params, uint64_t nrows)
{
    if (S.row0 >= nrows) return;
    mi_chelpers_params p = {};
    p.pols = S.d_mem;
    p.const_pols = S.d_const2ns;
    p.n_const = m->nConst;
    p.challenges = (const uint64_t *)params.challenges.address();
    p.n_challenges = params.challenges.degree();
    p.publics = (const uint64_t *)params.publicInputs;
    p.n_publics = m->nPublics;
    p.x = S.d_x2ns;
    p.x_stride = 1;
    uint64_t outOffset = m->qOffset;
    if (step == MI_CHELPERS_STEP42NS) {
        p.zhinv = m->zhinv.data();
        p.n_zhinv = m->zhinv.size();
        p.q = S.d_mem + m->qOffset;
    } else {
        p.evals = (const uint64_t *)params.evals.address();
        p.n_evals = m->nEvals;
        p.xdiv = S.d_xdiv;
        p.xdivw = S.d_xdivw;
        p.f = S.d_mem + m->fOffset;
        outOffset = m->fOffset;
    }
    const uint64_t rows = std::min(S.rows, nrows - S.row0);
    check(mi_multi_set_device(m->multi, S.shard), "Steps (row shard: device)");
    check(mi_chelpers_run_dev(mi_multi_ctx(m->multi, S.shard), prog, &p, S.row0, rows), "Steps (row shard: run the program)");
    check(mi_multi_copy(m->multi, m->d_mem + outOffset + 3 * S.row0, 0, S.d_mem + outOffset + 3 * S.row0, S.shard, rows * 3 * 8), "Steps (row shard: result rows home)");
    check(mi_multi_set_device(m->multi, 0), "Steps (row shard: device)");
    if (const char *log = std::getenv("MI_STARK_ROW_SHARD_LOG")) // tests: which rows left this device
        if (FILE *f = std::fopen(log, "a")) {
            std::fprintf(f, "step%dns shard %d device %d rows %llu %llu\n", step, S.shard, mi_ctx_device(mi_multi_ctx(m->multi, S.shard)), (unsigned long long)S.row0, (unsigned long long)(S.row0 + rows));
            std::fclose(f);
        }
}

// One constraint program, given as the reference's tables, over rows [0, nrows) of its domain, on the device image of params.pols.
inline void runChelpersStep(int step, const uint64_t *ops, uint64_t nops, const uint64_t *args, uint64_t nargs, StepsParams &params, uint64_t nrows)
{
    StarkMirror *m = mirrorOf(params);
    mi_chelpers_prog *&prog = (*m->progs)[{step | stepLayoutKey(m, step), (const void *)ops}];
    if (!prog) {
        const std::vector<mi_chelpers_section> secs = stepSections(m, step);
        const bool base = isBaseStep(step);
        check(mi_chelpers_compile(ctx(), &prog, step, ops, nops, args, nargs, secs.data(), secs.size(), m->nConst, base ? m->N : m->NExtended),
              "Steps::step*_parser_first_avx (translate the program)");
        buildStepProgram(m, step, prog);
    }
    if (isRowShardedStep(m, step)) {
        // the other devices first (their launches return at once), this device's rows beside them, then everybody's q rows are home
        if (m->syncHomeFirst) check(mi_ctx_sync(ctx()), "Steps (row shards: this device's tables are ready)");
        for (const StarkMirror::RowShard &S : m->rowShards) {
            mi_chelpers_prog *&sp = (*S.progs)[{step, (const void *)ops}];
            if (!sp) {
                const std::vector<mi_chelpers_section> secs = stepSections(m, step);
                check(mi_multi_set_device(m->multi, S.shard), "Steps (row shard: device)");
                check(mi_chelpers_compile(mi_multi_ctx(m->multi, S.shard), &sp, step, ops, nops, args, nargs, secs.data(), secs.size(), m->nConst, m->NExtended),
                      "Steps::step*_parser_first_avx (row shard: translate the program)");
                buildStepProgram(m, step, sp);
                check(mi_multi_set_device(m->multi, 0), "Steps (row shard: device)");
            }
            runRowShard(m, step, S, sp, params, nrows);
        }
        runStepProgram(m, step, prog, params, nrows);
        for (const StarkMirror::RowShard &S : m->rowShards) check(mi_multi_sync(m->multi, S.shard), "Steps (row shard: sync)");
        check(mi_multi_set_device(m->multi, 0), "Steps (row shard: device)");
        return;
    }
    runStepProgram(m, step, prog, params, nrows);
}
} // namespace mi

// The batched entry points of a Steps class over its generated tables.  In the translation unit that replaces the class's
// *.parser.cpp files:
//     #include "zkevmSteps.hpp"
//     #include "chelpers_steps.hpp"
//     #include "zkevm.chelpers.step42ns.parser.hpp"                       // op42[NOPS_], args42[NARGS_]
//     MI_DEFINE_PARSER_STEP(ZkevmSteps, step42ns, _avx, MI_CHELPERS_STEP42NS, op42, NOPS_, args42, NARGS_)
//     MI_FORWARD_PARSER_STEP(ZkevmSteps, step42ns, , _avx)                // step42ns_parser_first -> the same program
// nrowsBatch (4 = AVX2 lanes, 8 = AVX-512) has no meaning on the device: a lane owns a row.
#define MI_DEFINE_PARSER_STEP(Class, step, flavour, STEP_ID, ops, nops, args, nargs)                              \
    void Class::step##_parser_first##flavour(StepsParams &params, uint64_t nrows, uint64_t /*nrowsBatch*/)        \
    {                                                                                                              \
        mi::runChelpersStep(STEP_ID, ops, nops, args, nargs, params, nrows);                                       \
    }
#define MI_FORWARD_PARSER_STEP(Class, step, flavour, to)                                                           \
    void Class::step##_parser_first##flavour(StepsParams &params, uint64_t nrows, uint64_t nrowsBatch)             \
    {                                                                                                              \
        step##_parser_first##to(params, nrows, nrowsBatch);                                                        \
    }
#endif
