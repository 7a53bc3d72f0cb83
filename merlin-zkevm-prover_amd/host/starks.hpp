// starks.hpp -- class Starks with the reference's interface (src/starkpil/starks.hpp:19-232, starks.cpp:9-403), on a device image
// of the polynomial area.
//
//     Starks(const Config &, StarkFiles, void *pAddress)                      starks.hpp:74     (prover.cpp:128-132)
//     void genProof(FRIProof &, Goldilocks::Element *publicInputs, Steps *)   starks.hpp:225    (prover.cpp:541-544)
//     public: config, starkInfo, nrowsStepBatch
//
// The caller is unchanged: it fills the witness (cm1_n) at pAddress, constructs the Steps object of its STARK and calls genProof.
// What changes is where the work happens.  The reference keeps every section of StarkInfo's memory map in the host area behind
// pAddress (zkEVM: 254 GB + a 50 GB buffer) and walks it with OpenMP loops; here genProof keeps a DEVICE IMAGE of that area --
// same offsets, so the generated constraint programs address it as they address pAddress -- and strings the library's entry points
// together in the reference's order.  With the batched steps (nrowsStepBatch 4 / 8) pAddress is read (the witness) and never
// written; with per-row steps the caller's functions run on the HOST -- once at row 0 and once at the last row under the recorder,
// or over every row with MI_STEPS_ON_HOST=1 -- and write those rows of pAddress as they do in the reference.  Roots, evaluations
// and the query openings are what returns.
//
// HBM plan (one arena per process, carved per proof; zkEVM sizes in GB, N = 2^23):
//     image of pAddress[0, mapTotalN)     254.1   cm1_n 44.6 | cm2_n 8.6 | cm3_n 24.9 | cm4_n 0.4 | tmpExp_n 17.8 | cm1_2ns 89.3 | cm2_2ns 17.2 |
//                                                 cm3_2ns 49.8 | cm4_2ns 0.8 | q_2ns 0.4 | f_2ns 0.4
//     four Merkle trees                     4.3
//     constant polynomials, base domain    14.6   (218 of them; persistent per Starks: uploaded once, not per proof)
//   and nothing else of size: like the reference (starks.cpp:52 lends p_cm2_2ns, :102-104 reuses cm3_2ns) sections that are not live
//   yet, or no longer, serve as scratch --
//     stage 1 LDE scratch  = the cm2_2ns | cm3_2ns regions         stage 3 LDE scratch = cm1_n | cm2_n (their last reader, step3, has run)
//     stage 2 LDE scratch  = the cm3_2ns region                    stages 4, 5, FRI    = the whole base-domain part [0, cm1_2ns): the extended
//   constant polynomials are RE-EXTENDED there from the resident base-domain ones (an LDE of nConstants columns, 68 ms at zkEVM size,
//   instead of 29 GB over PCIe per proof; the constant TREE is only opened at the query points, from the mapped file), next to
//   x_2ns, the quotient's coefficient buffers, LEv / LpEv, xDivXSubXi / xDivXSubWXi and the FRI polynomials.
//   A STARK whose base-domain part is smaller than that (the recursive ones) gets the difference as extra arena.
//
// Steps: with nrowsStepBatch 4 or 8 (the zkEVM: prover.cpp:129) the batched entry points step*_parser_first_avx[512] run on the image
// (host/chelpers_steps.hpp).  With nrowsStepBatch 1 (c12a, recursive1/2: generated per-row C++, starks.cpp:84-88 ...) the Steps
// callbacks are the CALLER'S compiled host code: each is run once under a recorder (host/steps_tracer.hpp), which writes the row's
// field operations down as a program; that program is compiled for the device like the zkEVM's tables and runs over the image
// ("traced steps"; once per proving key).  A Steps class the recorder cannot follow (it branches on the row, or touches memory the
// recorder does not know) stops the proof with the reason; MI_STEPS_ON_HOST=1 asks for the reference's own loop instead: the sections
// the callbacks read are copied down into pAddress, the loop runs on the host cores, what it wrote is copied back up ("host steps").
#ifndef STARKS_HPP
#define STARKS_HPP
#include <algorithm>
#include <map>
#include <string>
#include <vector>
#include "config.hpp"
#include "utils.hpp"
#include "timer.hpp"
#include "zklog.hpp"
#include "exit_process.hpp"
#include "zkassert.hpp"
#include "constant_pols_starks.hpp"
#include "stark_info.hpp"
#include "friProof.hpp"
#include "friProve.hpp"
#include "transcript.hpp"
#include "zhInv.hpp"
#include "steps.hpp"
#include "merkleTreeGL.hpp"
#include "ntt_goldilocks.hpp"
#include "chelpers_steps.hpp"
#include "steps_tracer.hpp"
#include <set>
#include <typeinfo>

#define STARK_C12_A_NUM_TREES 5
#define NUM_CHALLENGES 8

struct StarkFiles
{
    std::string zkevmConstPols;
    bool mapConstPolsFile;
    std::string zkevmConstantsTree;
    std::string zkevmStarkInfo;
};

namespace mi {
// The process's HBM arena: grown (never shrunk) to the largest plan asked for, so that consecutive proofs -- zkEVM, c12a,
// recursive1, ... share pAddress in the reference and share this here -- allocate nothing.
//
// Two forms.  DENSE (one device): one allocation of the whole plan.  SPARSE (several devices, MI_STARK_DEVICES; MI_STARK_SPARSE_IMAGE=0
// switches it off): an ADDRESS RANGE (mi_vmm_reserve) under which a proof backs what it touches (back(): idempotent, additive).  A
// row-sharded proof keeps only ITS rows of the three wide extended sections on this device -- 20 of their 157 GB at zkEVM size and eight
// devices --, the rest of the range stays addresses; a proof that needs the whole image backs the whole image.
struct Arena
{
    uint64_t *base = nullptr;
    uint64_t elems = 0;
    bool sparse = false;
    uint64_t *reserve(uint64_t want, bool wantSparse = false)
    {
        if (want <= elems) return base;
        mi_ctx *c = ctx();
        if (base) { if (sparse) check(mi_vmm_free(c, base), "Starks (HBM arena: free)"); else mi_dev_free(c, base); }
        base = nullptr;
        sparse = wantSparse;
        if (sparse) {
            // addresses are free: room for any later, larger Starks of this process without moving (1 TiB of addresses: tools/vmm_probe.hip)
            const uint64_t range = std::max<uint64_t>(want * 8, 1ULL << 40);
            void *p = nullptr;
            check(mi_vmm_reserve(c, range, &p), "Starks (HBM arena: address range)");
            base = (uint64_t *)p;
            elems = range / 8;
            return base;
        }
        base = (uint64_t *)mi_dev_alloc(c, want * 8);
        if (!base) {
            uint64_t fr = 0, tot = 0;
            mi_dev_mem_info(c, &fr, &tot);
            std::fprintf(stderr, "mi_stark: the proof's HBM plan needs %.1f GB, the device has %.1f GB free of %.1f\n", want * 8 / 1e9, fr / 1e9, tot / 1e9);
            fail("Starks (HBM arena)");
        }
        elems = want;
        return base;
    }
    // physical memory under elements [off, off + n) (sparse form; the dense form has it all)
    void back(uint64_t off, uint64_t n)
    {
        if (!sparse || !n) return;
        if (mi_vmm_back(ctx(), base, off * 8, n * 8) != MI_OK) {
            std::fprintf(stderr, "mi_stark: the proof's HBM plan does not fit the device: %s\n", mi_last_error());
            fail("Starks (HBM arena: back)");
        }
    }
    uint64_t backedBytes()
    {
        uint64_t b = elems * 8;
        if (sparse) check(mi_vmm_backed_bytes(ctx(), base, &b), "Starks (HBM arena)");
        return b;
    }
};
inline Arena &arena()
{
    static Arena a;
    return a;
}
// The same on the OTHER devices of a row-sharded proof (shard g >= 1): the full-height mirror of the image's extended part -- an address
// range of which the shard's own rows (and the halo after them) are backed.  One per process and device, shared by every Starks of the
// prover (zkEVM, c12a, recursive1, recursive2 live side by side: prover.cpp:128-132) -- a proof fills it anew, one proof is in flight --;
// what belongs to a proving key (its constants and their extension) stays with its Starks.
struct ShardArena
{
    uint64_t *base = nullptr;
    uint64_t elems = 0;
    bool sparse = true;
    uint64_t *reserve(mi_multi *mm, int g, uint64_t want, bool wantSparse = true)
    {
        if (want <= elems) return base;
        check(mi_multi_set_device(mm, g), "Starks (row-shard mirror: device)");
        mi_ctx *c = mi_multi_ctx(mm, g);
        if (base) { if (sparse) check(mi_vmm_free(c, base), "Starks (row-shard mirror: free)"); else mi_dev_free(c, base); }
        sparse = wantSparse;
        if (sparse) {
            const uint64_t range = std::max<uint64_t>(want * 8, 1ULL << 39);
            void *p = nullptr;
            check(mi_vmm_reserve(c, range, &p), "Starks (row-shard mirror: address range)");
            base = (uint64_t *)p;
            elems = range / 8;
        } else { // MI_STARK_SPARSE_IMAGE=0: one allocation of the whole extended part (157 GB at zkEVM size on a device that holds little else)
            base = (uint64_t *)mi_dev_alloc(c, want * 8);
            if (!base) {
                std::fprintf(stderr, "mi_stark: the row-shard mirror on shard %d needs %.1f GB of that device's memory\n", g, want * 8 / 1e9);
                fail("Starks (row-shard mirror)");
            }
            elems = want;
        }
        check(mi_multi_set_device(mm, 0), "Starks (row-shard mirror: device)");
        return base;
    }
    void back(mi_multi *mm, int g, uint64_t off, uint64_t n)
    {
        if (!n || !sparse) return;
        check(mi_multi_set_device(mm, g), "Starks (row-shard mirror: device)");
        if (mi_vmm_back(mi_multi_ctx(mm, g), base, off * 8, n * 8) != MI_OK) {
            std::fprintf(stderr, "mi_stark: the row-shard mirror of shard %d does not fit its device: %s\n", g, mi_last_error());
            fail("Starks (row-shard mirror: back)");
        }
        check(mi_multi_set_device(mm, 0), "Starks (row-shard mirror: device)");
    }
};
inline ShardArena &shardArena(int g)
{
    static ShardArena a[64];
    return a[g];
}
} // namespace mi

class Starks
{
public:
    const Config &config;
    StarkInfo starkInfo;
    uint64_t nrowsStepBatch;

private:
    void *pConstPolsAddress = NULL;
    void *pConstTreeAddress = NULL;
    ConstantPolsStarks *pConstPols = NULL;
    ConstantPolsStarks *pConstPols2ns = NULL; // a view of the constant-tree file's polynomials (the reference copies them out: starks.hpp:141-143)
    StarkFiles starkFiles;
    ZhInv zi;
    uint64_t N, NExtended;
    uint64_t constPolsSize = 0;
    MerkleTreeGL *treesGL[STARK_C12_A_NUM_TREES] = {};
    Goldilocks::Element *mem = NULL;
    void *pAddress;
    // device side
    uint64_t *d_constN = nullptr; // constant polynomials over the base domain: resident for the life of this object
    bool constNTiled = false;     // ... tile-major ([N / 64][nConstants][64]): what the base-domain steps read in place (init())
    uint64_t treeElems = 0, scratchElems = 0;
    std::map<std::pair<int, const void *>, mi_chelpers_prog *> progs;

    uint64_t off(eSection s) const { return starkInfo.mapOffsets.section[s]; }
    uint64_t cols(eSection s) const { return starkInfo.mapSectionsN.section[s]; }
    struct PolRef { uint64_t offset, stride, dim; };
    PolRef polRef(uint64_t idPol) const // stark_info.cpp:473-482 without the host pointer
    {
        const VarPolMap &p = starkInfo.varPolMap[idPol];
        return {starkInfo.mapOffsets.section[p.section] + p.sectionPos, starkInfo.mapSectionsN.section[p.section], p.dim};
    }
    uint64_t exp2pol(uint64_t expId)
    {
        auto it = starkInfo.exp2pol.find(std::to_string(expId));
        if (it == starkInfo.exp2pol.end()) { zklog.error("Starks: expression " + std::to_string(expId) + " has no polynomial (exp2pol)"); exitProcess(); }
        return it->second;
    }
    // stage-4+ buffers: elements beyond what the dead base-domain part of the image offers
    uint64_t lateNeed() const
    {
        const uint64_t qd = starkInfo.qDim, qg = starkInfo.qDeg;
        uint64_t need = starkInfo.nConstants * NExtended + NExtended + 64;                       // const_2ns, x_2ns
        need += NExtended * qd + NExtended * qd * qg + 6 * N + 6 * NExtended;                    // qq1, qq2, LEv | LpEv, xDivXSubXi | xDivXSubWXi
        need += 9 * NExtended + 3 * NExtended + 4 * (2 * NExtended) + starkInfo.evMap.size() * 3 + 4096; // FRI: 3 polynomials, step-tree leaves and nodes, openings
        need += starkInfo.starkStruct.nQueries * (cols(cm1_n) + cols(cm2_n) + cols(cm3_n) + cols(cm4_2ns) + 4 * starkInfo.starkStruct.nBitsExt * 4 + 64);
        need += std::min<uint64_t>(64, std::max<uint64_t>(starkInfo.nConstants, 8)) * 2 * (N + NExtended); // room for an LDE's column chunk
        return need;
    }

public:
    Starks(const Config &config, StarkFiles starkFiles, void *_pAddress)
        : config(config), starkInfo(config, starkFiles.zkevmStarkInfo), starkFiles(starkFiles),
          zi(config.generateProof() ? starkInfo.starkStruct.nBits : 0, config.generateProof() ? starkInfo.starkStruct.nBitsExt : 0),
          N(config.generateProof() ? 1ULL << starkInfo.starkStruct.nBits : 0), NExtended(config.generateProof() ? 1ULL << starkInfo.starkStruct.nBitsExt : 0),
          pAddress(_pAddress)
    {
        nrowsStepBatch = 1;
        if (!config.generateProof()) return; // starks.hpp:90-91
        if (starkFiles.zkevmConstPols.size() == 0) { zklog.error("Starks::Starks() received an empty config.zkevmConstPols"); exitProcess(); }
        if (starkFiles.zkevmConstantsTree.size() == 0) { zklog.error("Starks::Starks() received an empty config.zkevmConstantsTree"); exitProcess(); }
        TimerStart(LOAD_CONST_POLS_TO_MEMORY);
        constPolsSize = starkInfo.nConstants * sizeof(Goldilocks::Element) * N;
        pConstPolsAddress = starkFiles.mapConstPolsFile ? mapFile(starkFiles.zkevmConstPols, constPolsSize, false) : copyFile(starkFiles.zkevmConstPols, constPolsSize);
        TimerStopAndLog(LOAD_CONST_POLS_TO_MEMORY);
        TimerStart(LOAD_CONST_TREE_TO_MEMORY);
        const uint64_t treeBytes = starkInfo.getConstTreeSizeInBytes();
        pConstTreeAddress = config.mapConstantsTreeFile ? mapFile(starkFiles.zkevmConstantsTree, treeBytes, false) : copyFile(starkFiles.zkevmConstantsTree, treeBytes);
        TimerStopAndLog(LOAD_CONST_TREE_TO_MEMORY);
        ownsConstants = true;
        init();
    }
    // Not in the reference: the same object over constants that are already in memory (embedding, benchmarks).  constPols: N x
    // nConstants; constTree: the image of the constant-tree file [nPols, nExt, pols, nodes] -- of which only the rows and paths the
    // queries open are ever read.  Both stay the caller's.
    Starks(const Config &config, const StarkInfo &info, void *constPols, void *constTree, void *_pAddress)
        : config(config), starkInfo(info), zi(info.starkStruct.nBits, info.starkStruct.nBitsExt), N(1ULL << info.starkStruct.nBits),
          NExtended(1ULL << info.starkStruct.nBitsExt), pAddress(_pAddress)
    {
        nrowsStepBatch = 1;
        constPolsSize = starkInfo.nConstants * sizeof(Goldilocks::Element) * N;
        pConstPolsAddress = constPols;
        pConstTreeAddress = constTree;
        init();
    }
    Starks(const Starks &) = delete;
    Starks &operator=(const Starks &) = delete;
    ~Starks()
    {
        if (!pConstPols) return; // nothing was set up (config.generateProof() false)
        if (witnessLocked) (void)mi_host_unregister(mi::ctx(), mem + off(cm1_n));
        if (mi_multi *mm = mi::multi()) {
            for (size_t g = 1; g < rowMem.size(); g++) {
                (void)mi_multi_set_device(mm, (int)g);
                mi_ctx *cg = mi_multi_ctx(mm, (int)g);
                for (auto &p : rowMem[g].progs) if (p.second) mi_chelpers_free(cg, p.second);
                if (rowMem[g].ownsTables)
                    for (uint64_t *q : {rowMem[g].constN, rowMem[g].const2ns, rowMem[g].x2ns, rowMem[g].xdiv, rowMem[g].lev}) if (q) mi_dev_free(cg, q); // (the mirror is the process's)
                if (rowMem[g].evalsPart) mi_dev_free(cg, rowMem[g].evalsPart);
            }
            (void)mi_multi_set_device(mm, 0);
        }
        for (auto &p : progs) if (p.second) mi_chelpers_free(mi::ctx(), p.second);
        if (d_constN) mi_dev_free(mi::ctx(), d_constN);
        delete pConstPols;
        delete pConstPols2ns;
        if (ownsConstants) {
            if (starkFiles.mapConstPolsFile) unmapFile(pConstPolsAddress, constPolsSize); else free(pConstPolsAddress);
            if (config.mapConstantsTreeFile) unmapFile(pConstTreeAddress, starkInfo.getConstTreeSizeInBytes()); else free(pConstTreeAddress);
        }
        for (unsigned i = 0; i < STARK_C12_A_NUM_TREES; i++) delete treesGL[i];
    }
    // bytes of HBM a genProof of this STARK plans for (image + trees + late scratch) plus the resident constant polynomials
    uint64_t hbmPlanBytes() const { return (starkInfo.mapTotalN + 4 * treeElems + scratchElems + starkInfo.nConstants * N) * 8; }

    void genProof(FRIProof &proof, Goldilocks::Element *publicInputs, Steps *steps);

    // Not in the reference: drop the compiled program cached for (step, the address of its opcode table) -- for a caller that REPLACES a
    // step's tables in place (the reference's tables are constants of the binary and never change)
    void forgetProgram(int step, const void *opsTable)
    {
        for (int key : {step, step | mi::MI_STEP_KEY_TILED, step | mi::MI_STEP_KEY_TILED_CONST, step | mi::MI_STEP_KEY_TILED | mi::MI_STEP_KEY_TILED_CONST}) {
            auto it = progs.find({key, opsTable});
            if (it == progs.end()) continue;
            if (it->second) mi_chelpers_free(mi::ctx(), it->second);
            progs.erase(it);
        }
        for (size_t g = 1; g < rowMem.size(); g++) {
            auto it = rowMem[g].progs.find({step, opsTable});
            if (it == rowMem[g].progs.end()) continue;
            (void)mi_multi_set_device(mi::multi(), (int)g);
            if (it->second) mi_chelpers_free(mi_multi_ctx(mi::multi(), (int)g), it->second);
            rowMem[g].progs.erase(it);
            (void)mi_multi_set_device(mi::multi(), 0);
        }
    }

    // device image of the last proof's polynomial area (valid until the next genProof of any Starks): for checks after the fact.
    // lateOffsets: where the stage-4 re-plan put the extended constant polynomials, xDivXSubXi and xDivXSubWXi (elements from the image's start)
    const uint64_t *deviceImage() const { return mi::arena().base; }
    // n elements of the last proof's image from element `offset` (inside one row): from the device that holds the row when the extension is
    // row-sharded (checks after the fact)
    void peekImage(uint64_t offset, uint64_t n, uint64_t *out)
    {
        if (ownRowsOnly)
            for (eSection e : {cm1_2ns, cm2_2ns, cm3_2ns})
                if (cols(e) && offset >= off(e) && offset < off(e) + cols(e) * NExtended) {
                    const uint64_t g = (offset - off(e)) / cols(e) / (NExtended / rowMem.size());
                    if (g == 0) break;
                    mi_multi *mm = mi::multi();
                    mi::check(mi_multi_set_device(mm, (int)g), "Starks::peekImage (device)");
                    mi::check(mi_copy_d2h(mi_multi_ctx(mm, (int)g), out, rowBase((int)g) + offset, n * 8), "Starks::peekImage (row of another device)");
                    mi::check(mi_multi_set_device(mm, 0), "Starks::peekImage (device)");
                    return;
                }
        if (ownRowsOnly && rowMem.size() > 1)
            for (int t = 1; t <= 2; t++) // xDivXSubXi / xDivXSubWXi: every device filled the rows it evaluates
                if (offset >= lateOffsets[t] && offset < lateOffsets[t] + 3 * NExtended) {
                    uint64_t g = (offset - lateOffsets[t]) / 3 / (NExtended / rowMem.size());
                    if (g == 0 || rowMem[g].aliasFrom == 0) break; // (a shard grouped with this device fills this device's table)
                    if (rowMem[g].aliasFrom > 0) g = (uint64_t)rowMem[g].aliasFrom;
                    mi_multi *mm = mi::multi();
                    mi::check(mi_multi_set_device(mm, (int)g), "Starks::peekImage (device)");
                    mi::check(mi_copy_d2h(mi_multi_ctx(mm, (int)g), out, rowMem[g].xdiv + (t - 1) * 3 * NExtended + (offset - lateOffsets[t]), n * 8), "Starks::peekImage (x / (x - xi) on a peer)");
                    mi::check(mi_multi_set_device(mm, 0), "Starks::peekImage (device)");
                    return;
                }
        if (constRowsOnly && offset >= lateOffsets[0] && offset < lateOffsets[0] + starkInfo.nConstants * NExtended &&
            (offset - lateOffsets[0]) / starkInfo.nConstants >= NExtended / rowMem.size()) { // a row of the extended constants this device did not fetch
            mi_multi *mm = mi::multi();
            mi::check(mi_multi_set_device(mm, constRowsFrom), "Starks::peekImage (device)");
            mi::check(mi_copy_d2h(mi_multi_ctx(mm, constRowsFrom), out, rowMem[constRowsFrom].const2ns + (offset - lateOffsets[0]), n * 8), "Starks::peekImage (constants on a peer)");
            mi::check(mi_multi_set_device(mm, 0), "Starks::peekImage (device)");
            return;
        }
        const eSection ext[3] = {cm1_2ns, cm2_2ns, cm3_2ns};
        for (int t = 0; t < 3; t++)
            if (tiledExtLast[t] && offset >= off(ext[t]) && offset < off(ext[t]) + cols(ext[t]) * NExtended) { // a tile-major section: the row's words, gathered
                const uint64_t w = cols(ext[t]), row = (offset - off(ext[t])) / w, col = (offset - off(ext[t])) % w;
                if (col + n > w) mi::fail("Starks::peekImage (a peek into a tile-major section stays inside one row)");
                uint64_t *tmp = mi::devAlloc(n, "Starks::peekImage (alloc)");
                mi::check(mi_untile_dev(mi::ctx(), tmp, n, deviceImage() + off(ext[t]), w, NExtended, col, row, 1, n), "Starks::peekImage (tile-major section)");
                mi::check(mi_copy_d2h(mi::ctx(), out, tmp, n * 8), "Starks::peekImage");
                mi::devFree(tmp);
                return;
            }
        mi::check(mi_copy_d2h(mi::ctx(), out, deviceImage() + offset, n * 8), "Starks::peekImage");
    }
    uint64_t lateOffsets[3] = {0, 0, 0};
    bool tiledExtLast[4] = {false, false, false, false}; // the last proof kept cm1_2ns .. cm3_2ns tile-major in the image (MI_STARK_TILED_EXT)

private:
    bool ownsConstants = false, witnessLocked = false, sparseImage = false;
    // Row-sharded step42ns (several devices): what shard g >= 1 keeps on ITS device from proof to proof -- a full-height mirror of the
    // image's extended part [cm1_2ns, end) of which the stage commits fill its own rows and the halo after them (157 GB of address range
    // at zkEVM size on a device that holds little else; 20 GB of it ever written), the constant polynomials and room for their extension
    // and x_2ns --, and its compiled programs.
    struct RowShardMem
    {
        uint64_t *ext = nullptr, *constN = nullptr, *const2ns = nullptr, *x2ns = nullptr, *xdiv = nullptr, *lev = nullptr; // xdiv: xDivXSubXi | xDivXSubWXi; lev: LEv | LpEv
        uint64_t *evalsPart = nullptr; // this shard's share of the evaluations (3 nEvals words)
        // Tables of the proving key.  A shard that leads its device group (or is alone on its device) OWNS them on its device; a shard
        // grouped with another (mi_multi_create2: logical shards of one physical device) reads its LEADER's -- and a shard grouped with
        // shard 0 reads what THIS device's proof holds (d_constN, the per-proof const_2ns / x_2ns / x-division tables / LEv): aliasFrom
        // is that leader, 0 meaning this device's image.  One physical device then holds one copy, which is what lets eight logical
        // shards rehearse the zkEVM's size on one GPU.
        bool ownsTables = false;
        int aliasFrom = -1;
        std::map<std::pair<int, const void *>, mi_chelpers_prog *> progs;
        bool tablesReady = false; // const2ns and x2ns are computed (once per Starks: they belong to the proving key)
    };
    std::vector<RowShardMem> rowMem; // by shard; [0] unused (shard 0 is this device and its image)
    int constRowsFrom = 1;      // ... the shard whose resident copy a check reads the other rows from
    bool constRowsOnly = false; // (per proof) the extended constants on this device: only its rows (+ halo), fetched from a peer's resident copy
    bool rowSharded = false, ownRowsOnly = false; // ownRowsOnly (per proof): this device's image holds only ITS rows (+ halo) of cm1..3_2ns
    uint64_t *rowBase(int g) const { return rowMem[g].ext - starkInfo.mapOffsets.section[cm1_2ns]; } // virtual: + an extended section's offset = that device's copy
    void init()
    {
        pConstPols = new ConstantPolsStarks(pConstPolsAddress, constPolsSize, starkInfo.nConstants);
        pConstPols2ns = new ConstantPolsStarks((uint8_t *)pConstTreeAddress + MERKLEHASHGOLDILOCKS_HEADER_SIZE * sizeof(Goldilocks::Element), NExtended, starkInfo.nConstants);
        mem = (Goldilocks::Element *)pAddress;
        checkMemoryMap();
        treesGL[4] = new MerkleTreeGL((Goldilocks::Element *)pConstTreeAddress); // opened at the query points from the file (merkleTreeGL.hpp:24-32)
        treeElems = MerklehashGoldilocks::getTreeNumElements(NExtended);
        // everything of stages 4, 5 and FRI fits the dead base-domain part of the image (the zkEVM: 51 of 96 GB), or gets an area of its own
        scratchElems = lateNeed() > off(cm1_2ns) ? lateNeed() : 0;
        if (starkInfo.nConstants) {
            d_constN = (uint64_t *)mi_dev_alloc(mi::ctx(), starkInfo.nConstants * N * 8);
            if (!d_constN) mi::fail("Starks::Starks (constant polynomials)");
            // Kept TILE-MAJOR on one device: every base-domain step reads them (the zkEVM's 218: three tile-major copies of 14.6 GB per
            // proof, 19 ms), nothing writes them, and their one other reader -- the extension of stage 4 -- takes them through an untiling
            // pass per column chunk (6 ms).  Row-major with several devices (the shards extend their own copies from it) and with
            // MI_STARK_TILED_CONSTS=0.
            const char *e = std::getenv("MI_STARK_TILED_CONSTS");
            constNTiled = !mi::multi() && N >= 64 && !(e && e[0] == '0');
            if (constNTiled) {
                uint64_t *tmp = (uint64_t *)mi_dev_alloc(mi::ctx(), starkInfo.nConstants * N * 8); // (before the image is reserved: there is room)
                if (!tmp) mi::fail("Starks::Starks (constant polynomials, staging)");
                mi::check(mi_copy_h2d(mi::ctx(), tmp, pConstPolsAddress, starkInfo.nConstants * N * 8), "Starks::Starks (constant polynomials h2d)");
                mi::check(mi_tile_major_dev(mi::ctx(), d_constN, starkInfo.nConstants, 0, tmp, starkInfo.nConstants, N, starkInfo.nConstants), "Starks::Starks (constant polynomials, tile-major)");
                mi::check(mi_ctx_sync(mi::ctx()), "Starks::Starks (constant polynomials, tile-major)");
                mi_dev_free(mi::ctx(), tmp);
            } else mi::check(mi_copy_h2d(mi::ctx(), d_constN, pConstPolsAddress, starkInfo.nConstants * N * 8), "Starks::Starks (constant polynomials h2d)");
        }
        // the proof's HBM now, not inside the first genProof: like the reference, which allocates pAddress when the prover starts
        // (prover.cpp:99-120).  273 GB of fresh device memory take the driver 5.7 s; a later, larger Starks grows the arena once more.
        {
            const char *sp = std::getenv("MI_STARK_SPARSE_IMAGE");
            sparseImage = mi::multi() != nullptr && !(sp && sp[0] == '0');
        }
        mi::arena().reserve(starkInfo.mapTotalN + 4 * treeElems + scratchElems, sparseImage);
        // Several devices (MI_STARK_DEVICES): the witness section of pAddress is page-locked ONCE, here -- like pAddress itself it lives as
        // long as the prover (prover.cpp:99-120) --, so that every device's DMA engines read their column tiles of stage 1 straight out of
        // it over their own PCIe link (csrc/multi.hip "strided"); pageable, the tiles are packed by host threads first, and eight links
        // wait for one socket's memory bandwidth.  MI_STARK_REGISTER_WITNESS=0 leaves it pageable; a refusal (RLIMIT_MEMLOCK) is not fatal.
        const char *rw = std::getenv("MI_STARK_REGISTER_WITNESS");
        if (mi::multi() && !(rw && rw[0] == '0') && cols(cm1_n) && pAddress) {
            TimerStart(STARK_PAGE_LOCK_WITNESS);
            if (mi_host_register(mi::ctx(), mem + off(cm1_n), N * cols(cm1_n) * 8) == MI_OK) witnessLocked = true;
            else zklog.warning("Starks: the witness section could not be page-locked (" + std::string(mi_last_error()) + "): stage 1 packs its tiles on the host");
            TimerStopAndLog(STARK_PAGE_LOCK_WITNESS);
        }
        // Row-sharded step42ns: on by default when MI_STARK_DEVICES names DISTINCT devices (MI_STARK_ROW_SHARDED=0 / 1 overrides: logical
        // shards on one device rehearse it -- at the zkEVM's full size too, when they form a device group: MI_MULTI_GROUP_SAME_DEVICE=1)
        if (mi_multi *mm = mi::multi()) {
            const int G = mi_multi_shards(mm);
            bool distinct = true;
            for (int a = 0; a < G; a++)
                for (int b = a + 1; b < G; b++) distinct = distinct && mi_ctx_device(mi_multi_ctx(mm, a)) != mi_ctx_device(mi_multi_ctx(mm, b));
            const char *rs = std::getenv("MI_STARK_ROW_SHARDED");
            const uint64_t halo = 1ULL << (starkInfo.starkStruct.nBitsExt - starkInfo.starkStruct.nBits);
            rowSharded = (rs ? rs[0] == '1' : distinct) && NExtended / (uint64_t)G >= 64 && halo <= NExtended / (uint64_t)G;
            if (rowSharded) {
                TimerStart(STARK_ROW_SHARD_SETUP);
                rowMem.resize(G);
                const uint64_t extElems = starkInfo.mapTotalN - off(cm1_2ns);
                for (int g = 1; g < G; g++) {
                    mi::check(mi_multi_set_device(mm, g), "Starks::Starks (row shards: device)");
                    mi_ctx *cg = mi_multi_ctx(mm, g);
                    RowShardMem &R = rowMem[g];
                    R.ext = mi::shardArena(g).reserve(mm, g, extElems, sparseImage); // (re-read at every proof: a later, larger Starks may move it)
                    mi::check(mi_multi_set_device(mm, g), "Starks::Starks (row shards: device)");
                    R.evalsPart = (uint64_t *)mi_dev_alloc(cg, (3 * starkInfo.evMap.size() + 16) * 8);
                    if (!R.evalsPart) mi::fail("Starks::Starks (row shards: device memory)");
                    const int L = mi_multi_lead(mm, g);
                    if (L != g) { R.aliasFrom = L; continue; } // its leader's tables (resolved per proof: the leader may be this device's image)
                    R.ownsTables = true;
                    R.x2ns = (uint64_t *)mi_dev_alloc(cg, NExtended * 8);
                    R.xdiv = (uint64_t *)mi_dev_alloc(cg, 6 * NExtended * 8);
                    R.lev = (uint64_t *)mi_dev_alloc(cg, (6 * N + 16) * 8);
                    if (starkInfo.nConstants) {
                        R.constN = (uint64_t *)mi_dev_alloc(cg, starkInfo.nConstants * N * 8);
                        R.const2ns = (uint64_t *)mi_dev_alloc(cg, starkInfo.nConstants * NExtended * 8);
                    }
                    if (!R.x2ns || !R.xdiv || !R.lev || (starkInfo.nConstants && (!R.constN || !R.const2ns))) mi::fail("Starks::Starks (row shards: device memory)");
                    if (starkInfo.nConstants)
                        mi::check(mi_copy_h2d(cg, R.constN, pConstPolsAddress, starkInfo.nConstants * N * 8), "Starks::Starks (row shards: constant polynomials h2d)");
                }
                mi::check(mi_multi_set_device(mm, 0), "Starks::Starks (row shards: device)");
                TimerStopAndLog(STARK_ROW_SHARD_SETUP);
            }
        }
    }
    // genProof lends sections that are not live as LDE / NTT / FRI scratch and re-plans the base-domain part from stage 4 on (see the
    // header): that is only sound for pil-stark's section order with every section starting where the previous one ends.  The reference
    // takes each offset from starkinfo.json and assumes nothing; a map laid out differently is refused here, with the reason, instead of
    // being proved with scratch that overlaps live polynomials.
    void checkMemoryMap()
    {
        static const eSection order[11] = {cm1_n, cm2_n, cm3_n, cm4_n, tmpExp_n, cm1_2ns, cm2_2ns, cm3_2ns, cm4_2ns, q_2ns, f_2ns};
        static const char *names[11] = {"cm1_n", "cm2_n", "cm3_n", "cm4_n", "tmpExp_n", "cm1_2ns", "cm2_2ns", "cm3_2ns", "cm4_2ns", "q_2ns", "f_2ns"};
        uint64_t next = 0;
        for (int i = 0; i < 11; i++) {
            const eSection e = order[i];
            if (off(e) != next) {
                zklog.error("Starks: starkinfo's memory map is not the contiguous pil-stark layout this prover's HBM plan relies on: section " + std::string(names[i]) +
                            " starts at " + std::to_string(off(e)) + ", expected " + std::to_string(next) + " (the end of the section before it)");
                exitProcess();
            }
            next += cols(e) * (i < 5 ? N : NExtended);
        }
        if (cols(q_2ns) != starkInfo.qDim || cols(f_2ns) != FIELD_EXTENSION || cols(cm1_2ns) != cols(cm1_n) || cols(cm2_2ns) != cols(cm2_n) || cols(cm3_2ns) != cols(cm3_n) ||
            cols(cm4_2ns) != starkInfo.qDim * starkInfo.qDeg) {
            zklog.error("Starks: starkinfo's section widths are inconsistent (cmK_2ns must be as wide as cmK_n, cm4_2ns = qDim * qDeg, q_2ns = qDim, f_2ns = 3)");
            exitProcess();
        }
        if (starkInfo.mapTotalN != next) {
            zklog.error("Starks: starkinfo's mapTotalN is " + std::to_string(starkInfo.mapTotalN) + ", the sections end at " + std::to_string(next));
            exitProcess();
        }
    }
    void hostStep(mi::StarkMirror &m, Steps *steps, StepsParams &params, int which);
    void tracedStep(mi::StarkMirror &m, Steps *steps, StepsParams &params, int which);
    mi_chelpers_prog *tracedProgram(mi::StarkMirror &m, Steps *steps, StepsParams &params, int which);
    std::set<std::pair<int, const void *>> emptySteps; // per-row steps that compute nothing (recursive1's step2prev)
};

// Host steps (nrowsStepBatch == 1): the caller's per-row code over pAddress, as starks.cpp:84-88,166-170,204-208,252-256,382-386 run
// it; sections it may read come down first, sections it may write go back up.  which: 0 step2prev, 1 step3prev, 2 step3, 3 step42ns,
// 4 step52ns.
inline void Starks::hostStep(mi::StarkMirror &m, Steps *steps, StepsParams &params, int which)
{
    mi_ctx *c = mi::ctx();
    auto down = [&](uint64_t o, uint64_t n) { if (n) mi::check(mi_copy_d2h(c, mem + o, m.d_mem + o, n * 8), "Starks::genProof (host steps, d2h)"); };
    auto up = [&](uint64_t o, uint64_t n) { if (n) mi::check(mi_copy_h2d(c, m.d_mem + o, mem + o, n * 8), "Starks::genProof (host steps, h2d)"); };
    const uint64_t baseBegin = off(cm2_n), baseEnd = off(cm1_2ns);
    if (which <= 2) {
        down(baseBegin, baseEnd - baseBegin); // cm2_n .. tmpExp_n as the device left them (cm1_n is the caller's witness: already there)
#pragma omp parallel for
        for (uint64_t i = 0; i < N; i++) {
            if (which == 0) steps->step2prev_first(params, i);
            else if (which == 1) steps->step3prev_first(params, i);
            else steps->step3_first(params, i);
        }
        up(baseBegin, baseEnd - baseBegin);
    } else if (which == 3) {
        down(off(cm1_2ns), off(cm4_2ns) - off(cm1_2ns));
#pragma omp parallel for
        for (uint64_t i = 0; i < NExtended; i++) steps->step42ns_first(params, i);
        up(off(q_2ns), NExtended * starkInfo.qDim);
    } else {
        down(off(cm4_2ns), NExtended * cols(cm4_2ns)); // cm1..3_2ns came down for step42ns and have not changed
        mi::check(mi_copy_d2h(c, params.xDivXSubXi.address(), m.d_xdiv, NExtended * 3 * 8), "Starks::genProof (host steps, xDivXSubXi)");
        mi::check(mi_copy_d2h(c, params.xDivXSubWXi.address(), m.d_xdivw, NExtended * 3 * 8), "Starks::genProof (host steps, xDivXSubWXi)");
#pragma omp parallel for
        for (uint64_t i = 0; i < NExtended; i++) steps->step52ns_first(params, i);
        up(off(f_2ns), NExtended * 3);
    }
}

// Traced steps (nrowsStepBatch == 1): the caller's per-row function recorded once (host/steps_tracer.hpp), compiled, run over the image.
inline void Starks::tracedStep(mi::StarkMirror &m, Steps *steps, StepsParams &params, int which)
{
    static const int ids[5] = {MI_CHELPERS_STEP2PREV, MI_CHELPERS_STEP3PREV, MI_CHELPERS_STEP3, MI_CHELPERS_STEP42NS, MI_CHELPERS_STEP52NS};
    if (mi_chelpers_prog *prog = tracedProgram(m, steps, params, which)) mi::runStepProgram(&m, ids[which], prog, params, which <= 2 ? N : NExtended);
}
// the recorded and compiled program of a per-row step (once per Steps class); null for a step that computes nothing
inline mi_chelpers_prog *Starks::tracedProgram(mi::StarkMirror &m, Steps *steps, StepsParams &params, int which)
{
    static const int ids[5] = {MI_CHELPERS_STEP2PREV, MI_CHELPERS_STEP3PREV, MI_CHELPERS_STEP3, MI_CHELPERS_STEP42NS, MI_CHELPERS_STEP52NS};
    const int step = ids[which];
    const bool base = which <= 2;
    const std::pair<int, const void *> key = {step | mi::MI_STEP_KEY_TRACED | mi::stepLayoutKey(&m, step),
                                              (const void *)&typeid(*steps)}; // per Steps class (one per proving key) and layout of the sections
    if (emptySteps.count(key)) return nullptr;
    mi_chelpers_prog *&prog = progs[key];
    if (!prog) {
        mi::TraceLayout L;
        L.step = step;
        L.rows = base ? N : NExtended;
        L.pols = (const uint64_t *)mem;
        for (int sct = 0; sct < (int)eSectionMax; sct++) {
            const eSection e = (eSection)sct;
            const bool baseDomain = e == cm1_n || e == cm2_n || e == cm3_n || e == cm4_n || e == tmpExp_n; // (the enum interleaves the domains)
            L.secs.push_back({off(e), cols(e), baseDomain ? N : NExtended});
        }
        L.qOffset = off(q_2ns); L.fOffset = off(f_2ns);
        ConstantPolsStarks *cp = base ? params.pConstPols : params.pConstPols2ns;
        L.constPols = cp ? (const uint64_t *)cp->address() : nullptr;
        L.nConst = starkInfo.nConstants;
        L.chal = (const uint64_t *)params.challenges.address(); L.nChal = params.challenges.degree();
        L.evals = (const uint64_t *)params.evals.address(); L.nEvals = params.evals.degree();
        L.pub = (const uint64_t *)params.publicInputs; L.nPub = starkInfo.nPublics;
        Polinomial &x = base ? params.x_n : params.x_2ns;
        L.x = (const uint64_t *)x.address(); L.xStride = x.offset();
        if (!base) { L.xd = (const uint64_t *)params.xDivXSubXi.address(); L.xdw = (const uint64_t *)params.xDivXSubWXi.address(); }
        std::vector<mi_chelpers_microop> mops;
        std::string err;
        auto call = [&](uint64_t i) {
            switch (which) { // the forms the reference's loops call: starks.cpp:86,168,206,254,384
            case 0: steps->step2prev_first(params, i); break;
            case 1: steps->step3prev_first(params, i); break;
            case 2: steps->step3_first(params, i); break;
            case 3: steps->step42ns_first(params, i); break;
            default: steps->step52ns_first(params, i); break;
            }
        };
        if (!mi::traceStep(L, call, [&](uint64_t i) { return Goldilocks::toU64(zi.zhInv(i)); }, mops, err)) mi::fail(err.c_str());
        if (mops.empty()) {
            progs.erase(key);
            emptySteps.insert(key);
            return nullptr;
        }
        const std::vector<mi_chelpers_section> secs = mi::stepSections(&m, step);
        mi::check(mi_chelpers_compile_micro(mi::ctx(), &prog, step, mops.data(), mops.size(), secs.data(), secs.size(), m.nConst, base ? N : NExtended),
                  "Starks::genProof (traced steps: translate the recorded program)");
        mi::buildStepProgram(&m, step, prog);
    }
    return prog;
}

inline void Starks::genProof(FRIProof &proof, Goldilocks::Element *publicInputs, Steps *steps)
{
    TimerStart(STARK_INITIALIZATION);
    mi_ctx *c = mi::ctx();
    const bool parserSteps = nrowsStepBatch == 4 || nrowsStepBatch == 8; // starks.cpp:69-90: the batched ("parser") forms
    const char *onHost = std::getenv("MI_STEPS_ON_HOST");
    const bool hostSteps = !parserSteps && onHost && onHost[0] == '1';
    const bool deviceSteps = !hostSteps; // parser tables or traced per-row code: nothing of the image comes down for a step
    auto perRowStep = [&](mi::StarkMirror &mm, StepsParams &pp, int which) { if (hostSteps) hostStep(mm, steps, pp, which); else tracedStep(mm, steps, pp, which); };
    const uint64_t nBits = starkInfo.starkStruct.nBits, nBitsExt = starkInfo.starkStruct.nBitsExt, extendBits = nBitsExt - nBits;
    const uint64_t nEvals = starkInfo.evMap.size();
    Transcript transcript;
    Polinomial evals(std::max<uint64_t>(nEvals, 1), FIELD_EXTENSION);
    Polinomial challenges(NUM_CHALLENGES, FIELD_EXTENSION);
    // host tables only the host steps read (starks.hpp:149-160; starks.cpp:17-18): empty with device steps
    // (traced steps only need them addressable: the recorder looks at addresses, at rows 0 and n - 1)
    Polinomial x_n(parserSteps ? 0 : N, 1), x_2ns(parserSteps ? 0 : NExtended, 1);
    Polinomial xDivXSubXi(parserSteps ? 0 : NExtended, FIELD_EXTENSION), xDivXSubWXi(parserSteps ? 0 : NExtended, FIELD_EXTENSION);
    Polinomial root0(HASH_SIZE, 1), root1(HASH_SIZE, 1), root2(HASH_SIZE, 1), root3(HASH_SIZE, 1);

    // ---- the HBM plan: image | trees | late scratch
    const uint64_t imageElems = starkInfo.mapTotalN;
    uint64_t *d_mem = mi::arena().reserve(imageElems + 4 * treeElems + scratchElems, sparseImage);
    uint64_t *d_nodes[4];
    for (int t = 0; t < 4; t++) d_nodes[t] = d_mem + imageElems + t * treeElems;
    uint64_t *d_late = d_mem + imageElems + 4 * treeElems;
    auto sec = [&](eSection s) { return d_mem + off(s); };
    mi_multi *mm = mi::multi();
    auto sharded = [&](uint64_t ncols) { return mm && ncols > 4 && NExtended / (uint64_t)mi_multi_shards(mm) >= 64; };
    // Row shards active (several devices, the table steps): NOBODY on this device reads another shard's rows of the extension any more --
    // step42ns, step52ns and evmap take this device's rows, the openings fetch a row's values from the device that holds it -- so the
    // commits no longer send the whole extension here (seven links into one device: 137 GB per zkEVM-size proof, the sharded commit's
    // longest transfer): this device, like the others, receives its own rows and the halo -- and, the image being an address range
    // (mi::Arena, sparse), holds physical memory under those rows only.
    ownRowsOnly = rowSharded && parserSteps;
    {
        const uint64_t G = rowMem.size(), R = G ? NExtended / G : 0, halo = 1ULL << extendBits;
        const eSection wide[3] = {cm1_2ns, cm2_2ns, cm3_2ns};
        mi::Arena &A = mi::arena();
        if (A.sparse) {
            const uint64_t before = A.backedBytes();
            if (ownRowsOnly) {
                A.back(0, off(cm1_2ns));                                                    // the base-domain part (from stage 4: the late plan)
                for (eSection e : wide) A.back(off(e), (sharded(cols(e)) ? R + halo : NExtended) * cols(e)); // this device's rows (+ halo); a narrow section is committed here, whole
                A.back(off(cm4_2ns), imageElems - off(cm4_2ns));                            // the quotient's chunks, q, f: built and folded here
                A.back(imageElems, 4 * treeElems + scratchElems);
            } else A.back(0, imageElems + 4 * treeElems + scratchElems);
            if (A.backedBytes() != before) // the other devices' kernels write this device's rows of a tile straight into the image
                for (int g = 1; g < mi_multi_shards(mm); g++) mi::check(mi_vmm_allow_peer(c, A.base, mi_ctx_device(mi_multi_ctx(mm, g))), "Starks::genProof (image: peer access)");
        }
        // the other shards' mirrors: their rows (+ halo, wrapping) of the extended sections, their rows of q and f
        for (uint64_t g = 1; g < G; g++) {
            mi::ShardArena &S = mi::shardArena((int)g);
            rowMem[g].ext = S.reserve(mm, (int)g, imageElems - off(cm1_2ns), sparseImage);
            if (!S.sparse) continue; // one allocation: everything is there
            mi_ctx *cg = mi_multi_ctx(mm, (int)g);
            uint64_t before = 0, after = 0;
            mi::check(mi_multi_set_device(mm, (int)g), "Starks::genProof (row shards: device)");
            mi::check(mi_vmm_backed_bytes(cg, S.base, &before), "Starks::genProof (row-shard mirror)");
            const uint64_t r0 = g * R, o0 = off(cm1_2ns);
            for (eSection e : {cm1_2ns, cm2_2ns, cm3_2ns, cm4_2ns}) {
                const uint64_t w = cols(e), first = std::min(R + halo, NExtended - r0);
                S.back(mm, (int)g, off(e) - o0 + r0 * w, first * w);
                if (first < R + halo) S.back(mm, (int)g, off(e) - o0, (R + halo - first) * w); // the last shard's halo: the extension's first rows
            }
            for (eSection e : {q_2ns, f_2ns}) S.back(mm, (int)g, off(e) - o0 + r0 * cols(e), R * cols(e));
            mi::check(mi_multi_set_device(mm, (int)g), "Starks::genProof (row shards: device)");
            mi::check(mi_vmm_backed_bytes(cg, S.base, &after), "Starks::genProof (row-shard mirror)");
            if (after != before)
                for (int h = 0; h < mi_multi_shards(mm); h++) mi::check(mi_vmm_allow_peer(cg, S.base, mi_ctx_device(mi_multi_ctx(mm, h))), "Starks::genProof (row-shard mirror: peer access)");
        }
        if (G) mi::check(mi_multi_set_device(mm, 0), "Starks::genProof (row shards: device)");
    }
    auto lend = [&](uint64_t *p, uint64_t elems) { // LDE / NTT scratch out of a region that is not live
        const uint64_t atLeast = 16 * 2 * (N + NExtended);
        mi::check(mi_ctx_lend_workspace(c, elems >= atLeast ? p : nullptr, elems * 8), "Starks::genProof (lend workspace)");
    };
    // The computed base-domain sections start every proof as ZEROS.  Every column of them is extended and committed, written or not
    // (starks.cpp:133,214); a real starkinfo leaves none unwritten, but a map with a column no step stores into (the synthetic shapes
    // have them) must not commit what the PREVIOUS proof's stage-4 plan left there -- the region is scratch between proofs.  Zeros are
    // what the reference's calloc'ed pAddress holds in a column nothing writes (prover.cpp:113).  52 GB at zkEVM size: 17 ms of HBM
    // writes on this stream, beside a stage 1 that is bound by the PCIe link -- queued where nothing waits for them: on one device that
    // is after stage 1 has lent its scratch (lending synchronises the stream: queued here, STARK_INITIALIZATION was 8.8 ms of waiting for
    // this fill), while the host packs the first chunk of the witness; with several devices here, ahead of the shards' buffers.
    bool zeroed = false;
    auto zeroComputed = [&]() {
        if (zeroed) return;
        mi::check(mi_dev_zero(c, sec(cm2_n), (off(cm1_2ns) - off(cm2_n)) * 8), "Starks::genProof (zero the computed base-domain sections)");
        zeroed = true;
    };
    if (mi::multi()) zeroComputed();
    lend(nullptr, 0); // whatever the previous proof lent last (the tail of ITS stage-4 plan: inside what are live sections again now) is not this proof's
    mi::StarkMirror m;
    m.hostPols = mem; m.d_mem = d_mem; m.N = N; m.NExtended = NExtended; m.nBits = nBits; m.nBitsExt = nBitsExt;
    m.nPublics = starkInfo.nPublics; m.nEvals = nEvals;
    const eSection sN[4] = {cm1_n, cm2_n, cm3_n, tmpExp_n}, s2[4] = {cm1_2ns, cm2_2ns, cm3_2ns, cm4_2ns};
    for (int i = 0; i < 4; i++) { m.cmN[i] = {off(sN[i]), cols(sN[i])}; m.cm2ns[i] = {off(s2[i]), cols(s2[i])}; }
    m.qOffset = off(q_2ns); m.fOffset = off(f_2ns);
    m.d_constN = d_constN; m.nConst = starkInfo.nConstants; m.tiledConstN = constNTiled;
    m.zhinv.resize(1ULL << extendBits);
    for (uint64_t i = 0; i < m.zhinv.size(); i++) m.zhinv[i] = Goldilocks::toU64(zi.zhInv(i));
    m.progs = &progs;
    if (const char *e = std::getenv("MI_CHELPERS_CACHE")) m.cacheDir = e;
    // more than one device (MI_STARK_DEVICES, csrc/multi.hip): the commit of a stage is sharded over them -- each device extends column
    // tiles (stage 1: uploaded over ITS OWN PCIe link) and hashes the leaves and the subtree of its rows -- while the row-major extension
    // still lands in this device's image, where the constraint evaluation reads it.  The shard on this device works in the same dead
    // regions the single-device path lends as scratch.
    // The witness TILE-MAJOR in the image ([64 rows][column][row in tile], mi_lde_merkle_host_keep_tiled): the three base-domain steps read
    // nearly every column of cm1_n (the zkEVM's: 497 / 647 / 553 of 665), and from a row-major section each of them first makes a tile-major
    // copy of it (a read and a write of 44.6 GB); kept tile-major -- written that way behind the upload, where the row-major copy was written
    // -- they read it in place.  Only the steps read cm1_n on the device; should a lookup or grand-product operand be a witness column
    // itself (read by stride, polRef), the section stays row-major.  MI_STARK_TILED_WITNESS=0: row-major as before.
    {
        const char *e = std::getenv("MI_STARK_TILED_WITNESS");
        bool ok = deviceSteps && !(e && e[0] == '0') && N >= 64 && cols(cm1_n) > 0 && !sharded(cols(cm1_n));
        auto inWitness = [&](uint64_t idPol) { const PolRef r = polRef(idPol); return r.offset >= off(cm1_n) && r.offset < off(cm1_n) + N * cols(cm1_n); };
        for (auto &x : starkInfo.puCtx) ok = ok && !inWitness(exp2pol(x.fExpId)) && !inWitness(exp2pol(x.tExpId));
        for (auto &x : starkInfo.puCtx) ok = ok && !inWitness(exp2pol(x.numId)) && !inWitness(exp2pol(x.denId));
        for (auto &x : starkInfo.peCtx) ok = ok && !inWitness(exp2pol(x.numId)) && !inWitness(exp2pol(x.denId));
        for (auto &x : starkInfo.ciCtx) ok = ok && !inWitness(exp2pol(x.numId)) && !inWitness(exp2pol(x.denId));
        m.tiledWitness = ok;
    }
    // The wide extended sections TILE-MAJOR as well (mi_lde_merkle_host_tiled / _dev_tiled: the leaf kernel writes them while it absorbs a
    // chunk of the extension, where extendPol's row-major form would have been written): step42ns read nearly every column of them from a
    // tile-major copy it made per batch of rows (87 ms of a 2.15 s zkEVM-size proof, 413 GB moved); now the generated kernels, the linear
    // kernel of step52ns, evmap and the openings read the sections in place.  One device only (a sharded commit delivers row-major rows,
    // csrc/multi.hip); cm4_2ns (the quotient's chunks: qDim * qDeg columns out of an NTT of their own) stays row-major.
    // MI_STARK_TILED_EXT=0: row-major as before.
    {
        const char *e = std::getenv("MI_STARK_TILED_EXT");
        const char *be = std::getenv("MI_CHELPERS_BACKEND"); // (the interpreter, an A/B switch of the extended-domain steps, reads row-major sections)
        const bool on = deviceSteps && !(e && e[0] == '0') && !mm && NExtended >= 64 && !(be && std::string(be) == "interpreter");
        for (int i = 0; i < 3; i++) m.tiledExt[i] = on && cols(s2[i]) > 4;
        for (int i = 0; i < 4; i++) tiledExtLast[i] = m.tiledExt[i];
    }
    mi::currentMirror() = &m;

    transcript.put(&publicInputs[0], starkInfo.nPublics);
    ConstantPolsStarks *cp = pConstPols, *cp2 = pConstPols2ns;
    StepsParams params = {mem, cp, cp2, challenges, x_n, x_2ns, zi, evals, xDivXSubXi, xDivXSubWXi, publicInputs, mem + off(q_2ns), mem + off(f_2ns)};
    // traced steps: record and compile the five per-row functions now (first proof of this Steps class only), the two that every STARK
    // has first -- a Steps class the recorder cannot follow is refused before any work is done
    if (!parserSteps && !hostSteps)
        for (int which : {3, 4, 0, 1, 2}) (void)tracedProgram(m, steps, params, which);
    // x_n (starks.hpp:149-154), N elements read by the base-domain steps only: it borrows the head of the q_2ns section, whose first
    // writer (step42ns) runs after the last of them
    m.d_xn = sec(q_2ns);
    mi::check(mi_geom_seq_dev(c, m.d_xn, N, 1, Goldilocks::toU64(Goldilocks::w(nBits))), "Starks::genProof (x_n)");
    if (!deviceSteps) mi::check(mi_copy_d2h(c, x_n.address(), m.d_xn, N * 8), "Starks::genProof (x_n d2h)");
    TimerStopAndLog(STARK_INITIALIZATION);

    //--------------------------------
    // 1.- Calculate p_cm1_2ns  (starks.cpp:48-61): the witness streams up in column chunks behind the kernels and stays, its
    //     extension and tree are built as the chunks arrive
    //--------------------------------
    TimerStart(STARK_STEP_1);
    TimerStart(STARK_STEP_1_LDE_AND_MERKLETREE);
    mi_multi_tree *mtree[4] = {};
    const int dev0 = mi_ctx_device(c);
    // Scratch of a stage's transforms and of shard 0's commit buffers: a region of the image that is not live.  With the whole image
    // backed, the reference's choices (starks.cpp:52 lends p_cm2_2ns; :102-104 reuses cm3_2ns); with only this device's rows of the
    // extended sections backed (ownRowsOnly), stages 1 and 2 borrow COMPUTED BASE-DOMAIN sections that nothing has written yet --
    // cm2_n .. tmpExp_n before step2prev, cm3_n | cm4_n before step3prev -- and zero them again afterwards (they start every proof as
    // zeros, see above).  Stage 3: cm1_n | cm2_n, whose last reader has run, in both forms.
    struct Scratch { uint64_t *p; uint64_t elems; bool rezero; };
    auto stageScratch = [&](int stage) -> Scratch {
        if (stage == 3) return {sec(cm1_n), off(cm3_n) - off(cm1_n), false};
        if (!ownRowsOnly) return stage == 1 ? Scratch{sec(cm2_2ns), off(cm4_2ns) - off(cm2_2ns), false} : Scratch{sec(cm3_2ns), off(cm4_2ns) - off(cm3_2ns), false};
        return stage == 1 ? Scratch{sec(cm2_n), off(cm1_2ns) - off(cm2_n), true} : Scratch{sec(cm3_n), off(tmpExp_n) - off(cm3_n), true};
    };
    auto doneWith = [&](const Scratch &S) {
        if (!S.rezero || !S.elems) return;
        lend(nullptr, 0);
        mi::check(mi_dev_zero(c, S.p, S.elems * 8), "Starks::genProof (zero the borrowed base-domain sections)");
    };
    // A stage is committed over the shards unless this device could not hold its share: the whole image is planned to the last byte, and
    // shard 0's buffers of an ordinary (windowed) commit must fit the region the stage lends -- at two devices 103 GB of tiles and row
    // windows against the 67 GB of cm2_2ns | cm3_2ns.  Then the stage runs on this device alone, and says so once.  (A
    // row-sharded proof commits transiently: 21.5 GB whatever the number of devices, and its image is sparse.)
    auto shardedStage = [&](int stage, uint64_t ncols) {
        if (!sharded(ncols)) return false;
        if (ownRowsOnly) return true;
        const uint64_t need = mi_multi_windowed_need(N, NExtended, ncols, (uint32_t)mi_multi_shards(mm));
        if (stageScratch(stage).elems >= need) return true;
        uint64_t fr = 0, tot = 0;
        mi::check(mi_dev_mem_info(c, &fr, &tot), "Starks::genProof (device memory)");
        if (fr >= need * 8 + (2ULL << 30)) return true; // (the shard allocates them instead: a small STARK, or a device with room)
        static bool told[4] = {};
        if (!told[stage]) {
            told[stage] = true;
            zklog.warning("Starks: stage " + std::to_string(stage) + " is committed on one device: shard 0 of " + std::to_string(mi_multi_shards(mm)) + " needs " +
                          std::to_string(need * 8 / 1000000000) + " GB of buffers beside the image, the region the stage lends has " +
                          std::to_string(stageScratch(stage).elems * 8 / 1000000000) + " GB (more devices need less; MI_STARK_ROW_SHARDED=1 commits transiently)");
        }
        return false;
    };
    auto commitSharded = [&](int t, const uint64_t *src, int srcDevice, uint64_t ncols, uint64_t *image, uint64_t *base, const Scratch &scr, Goldilocks::Element *root) {
        mi::check(mi_ctx_sync(c), "Starks::genProof (sharded commit: sync)"); // the section and the scratch's last readers ran on this context's stream
        mi::check(mi_multi_lend(mm, 0, scr.p, scr.elems * 8), "Starks::genProof (sharded commit: lend)");
        if (rowSharded) { // every other shard keeps its own rows of this extension (+ the rows its shifted reads reach) on its device, for step42ns
            std::vector<uint64_t *> imgs(rowMem.size(), nullptr);
            for (size_t g = 1; g < rowMem.size(); g++) imgs[g] = rowBase((int)g) + (image - d_mem);
            if (ownRowsOnly) {
                // ... and so does this device: nobody opens values from the tree (the openings read the images), every shard has a row
                // image: a TRANSIENT commit -- a tile's rows are written once, by a kernel of the shard that extended it, straight into
                // their owner's image and absorbed there (csrc/multi.hip)
                imgs[0] = image;
                image = nullptr;
                mi::check(mi_multi_set_transient(mm, 1), "Starks::genProof (sharded commit: transient)");
            }
            mi::check(mi_multi_set_row_images(mm, imgs.data(), ncols, 1ULL << extendBits), "Starks::genProof (sharded commit: row images)");
        }
        mi::check(mi_multi_commit(mm, &mtree[t], src, ncols, srcDevice, N, NExtended, ncols, image, ncols, base, ncols, dev0, (uint64_t *)root),
                  "Starks::genProof (sharded extendPol + merkelize)");
        mi::check(mi_multi_tree_release_rows(mtree[t]), "Starks::genProof (sharded commit: release)"); // the openings read the rows from the image
        if (scr.rezero) doneWith(scr);
        else lend(scr.p, scr.elems); // ... and the region serves this context's transforms until the next stage, as on the one-device path
    };
    // a section whose commit is NOT sharded (at most 4 columns: linear_hash copies such rows) still has to reach the row shards: their rows
    // (+ halo, wrapping) out of this image, contiguous because whole rows of a section are
    auto mirrorRows = [&](eSection e) {
        if (!rowSharded || !cols(e)) return;
        mi::check(mi_ctx_sync(c), "Starks::genProof (row shards: sync)");
        const uint64_t R = NExtended / rowMem.size(), halo = 1ULL << extendBits, w = cols(e);
        for (size_t g = 1; g < rowMem.size(); g++) {
            const uint64_t r0 = g * R, r1 = (r0 + R) % NExtended;
            mi::check(mi_multi_copy(mm, rowBase((int)g) + off(e) + r0 * w, (int)g, sec(e) + r0 * w, 0, R * w * 8), "Starks::genProof (row shards: rows of an unsharded section)");
            mi::check(mi_multi_copy(mm, rowBase((int)g) + off(e) + r1 * w, (int)g, sec(e) + r1 * w, 0, halo * w * 8), "Starks::genProof (row shards: halo of an unsharded section)");
        }
        mi::check(mi_multi_sync(mm, 0), "Starks::genProof (row shards: sync)");
        mi::check(mi_multi_set_device(mm, 0), "Starks::genProof (row shards: device)");
    };
    if (shardedStage(1, cols(cm1_n))) {
        commitSharded(0, (const uint64_t *)(mem + off(cm1_n)), -1, cols(cm1_n), sec(cm1_2ns), sec(cm1_n), stageScratch(1), root0.address());
    } else {
        const Scratch scr = stageScratch(1);
        if (scr.rezero) zeroComputed(); // (the scratch IS part of those sections: zeros first, and again when the stage is done)
        lend(scr.p, scr.elems);
        zeroComputed();
        if (m.tiledExt[0])
            mi::check(mi_lde_merkle_host_tiled(c, d_nodes[0], sec(cm1_2ns), sec(cm1_n), m.tiledWitness ? 0 : cols(cm1_n), (const uint64_t *)(mem + off(cm1_n)), N, NExtended,
                                               cols(cm1_n), 0), "Starks::genProof (stage 1: extendPol + merkelize)");
        else if (m.tiledWitness)
            mi::check(mi_lde_merkle_host_keep_tiled(c, d_nodes[0], sec(cm1_2ns), cols(cm1_n), sec(cm1_n), (const uint64_t *)(mem + off(cm1_n)), N, NExtended,
                                                    cols(cm1_n), 0), "Starks::genProof (stage 1: extendPol + merkelize)");
        else
            mi::check(mi_lde_merkle_host_keep(c, d_nodes[0], sec(cm1_2ns), cols(cm1_n), sec(cm1_n), cols(cm1_n), (const uint64_t *)(mem + off(cm1_n)), N, NExtended,
                                              cols(cm1_n), 0), "Starks::genProof (stage 1: extendPol + merkelize)");
        mi::check(mi_copy_d2h(c, root0.address(), d_nodes[0] + treeElems - HASH_SIZE, HASH_SIZE * 8), "Starks::genProof (root 1)");
        mirrorRows(cm1_2ns);
        doneWith(scr);
    }
    transcript.put(root0.address(), HASH_SIZE);
    TimerStopAndLog(STARK_STEP_1_LDE_AND_MERKLETREE);
    TimerStopAndLog(STARK_STEP_1);

    //--------------------------------
    // 2.- Caluculate plookups h1 and h2  (starks.cpp:66-143)
    //--------------------------------
    TimerStart(STARK_STEP_2);
    transcript.getField(challenges[0]); // u
    transcript.getField(challenges[1]); // defVal
    TimerStart(STARK_STEP_2_CALCULATE_EXPS);
    if (nrowsStepBatch == 4) steps->step2prev_parser_first_avx(params, N, nrowsStepBatch);
    else if (nrowsStepBatch == 8) steps->step2prev_parser_first_avx512(params, N, nrowsStepBatch);
    else perRowStep(m, params, 0);
    TimerStopAndLog(STARK_STEP_2_CALCULATE_EXPS);
    TimerStart(STARK_STEP_2_CALCULATEH1H2);
    uint64_t numCommited = starkInfo.nCm1;
    for (uint64_t i = 0; i < starkInfo.puCtx.size(); i++) { // starks.cpp:106-124 with the transposes of :92,:128 gone: strided views in place
        const PolRef f = polRef(exp2pol(starkInfo.puCtx[i].fExpId)), t = polRef(exp2pol(starkInfo.puCtx[i].tExpId));
        const PolRef h1 = polRef(starkInfo.cm_n[numCommited + i * 2]), h2 = polRef(starkInfo.cm_n[numCommited + i * 2 + 1]);
        if (f.dim != t.dim || h1.dim != t.dim || h2.dim != t.dim) mi::fail("Starks::genProof (plookup polynomials of different dimensions)");
        mi::check(mi_calculate_h1h2_dev(c, d_mem + h1.offset, h1.stride, d_mem + h2.offset, h2.stride, d_mem + f.offset, f.stride, d_mem + t.offset, t.stride,
                                        (unsigned)t.dim, N), "Starks::genProof (calculateH1H2)");
    }
    numCommited += starkInfo.puCtx.size() * 2;
    TimerStopAndLog(STARK_STEP_2_CALCULATEH1H2);
    TimerStart(STARK_STEP_2_LDE_AND_MERKLETREE);
    if (shardedStage(2, cols(cm2_n))) {
        commitSharded(1, sec(cm2_n), dev0, cols(cm2_n), sec(cm2_2ns), nullptr, stageScratch(2), root1.address());
    } else {
        const Scratch scr = stageScratch(2);
        lend(scr.p, scr.elems);
        if (m.tiledExt[1])
            mi::check(mi_lde_merkle_dev_tiled(c, d_nodes[1], sec(cm2_2ns), sec(cm2_n), cols(cm2_n), N, NExtended, cols(cm2_n)), "Starks::genProof (stage 2: extendPol + merkelize)");
        else {
            mi::check(mi_lde_dev(c, sec(cm2_2ns), cols(cm2_n), sec(cm2_n), cols(cm2_n), NExtended, N, cols(cm2_n)), "Starks::genProof (stage 2: extendPol)");
            mi::check(mi_merkle_build_dev(c, d_nodes[1], sec(cm2_2ns), cols(cm2_n), cols(cm2_n), NExtended), "Starks::genProof (stage 2: merkelize)");
        }
        mi::check(mi_copy_d2h(c, root1.address(), d_nodes[1] + treeElems - HASH_SIZE, HASH_SIZE * 8), "Starks::genProof (root 2)");
        mirrorRows(cm2_2ns);
        doneWith(scr);
    }
    transcript.put(root1.address(), HASH_SIZE);
    TimerStopAndLog(STARK_STEP_2_LDE_AND_MERKLETREE);
    TimerStopAndLog(STARK_STEP_2);

    //--------------------------------
    // 3.- Compute Z polynomials  (starks.cpp:148-223)
    //--------------------------------
    TimerStart(STARK_STEP_3);
    transcript.getField(challenges[2]); // gamma
    transcript.getField(challenges[3]); // betta
    TimerStart(STARK_STEP_3_CALCULATE_EXPS);
    if (nrowsStepBatch == 4) steps->step3prev_parser_first_avx(params, N, nrowsStepBatch);
    else if (nrowsStepBatch == 8) steps->step3prev_parser_first_avx512(params, N, nrowsStepBatch);
    else perRowStep(m, params, 1);
    TimerStopAndLog(STARK_STEP_3_CALCULATE_EXPS);
    TimerStart(STARK_STEP_3_CALCULATE_Z);
    {
        struct ND { uint64_t numId, denId; };
        std::vector<ND> gp; // the order of starks.cpp:473-536: lookups, permutations, connections
        for (auto &x : starkInfo.puCtx) gp.push_back({x.numId, x.denId});
        for (auto &x : starkInfo.peCtx) gp.push_back({x.numId, x.denId});
        for (auto &x : starkInfo.ciCtx) gp.push_back({x.numId, x.denId});
        // all of them in one pass over the rows (mi_calculate_z_batch_dev): each reads 2 x 3 words of the same rows of tmpExp_n
        const uint64_t k = gp.size();
        std::vector<uint64_t *> zp(k);
        std::vector<const uint64_t *> np_(k), dp(k);
        std::vector<uint64_t> zs(k), ns(k), ds(k);
        std::vector<int> closes(k, 0);
        for (uint64_t i = 0; i < k; i++) {
            const PolRef num = polRef(exp2pol(gp[i].numId)), den = polRef(exp2pol(gp[i].denId)), z = polRef(starkInfo.cm_n[numCommited + i]);
            zp[i] = d_mem + z.offset; np_[i] = d_mem + num.offset; dp[i] = d_mem + den.offset;
            zs[i] = z.stride; ns[i] = num.stride; ds[i] = den.stride;
        }
        mi::check(mi_calculate_z_batch_dev(c, (unsigned)k, zp.data(), zs.data(), np_.data(), ns.data(), dp.data(), ds.data(), N, closes.data()),
                  "Starks::genProof (calculateZ)");
        for (uint64_t i = 0; i < k; i++) { zkassert(closes[i]); (void)closes[i]; } // polinomial.hpp:606
    }
    TimerStopAndLog(STARK_STEP_3_CALCULATE_Z);
    TimerStart(STARK_STEP_3_CALCULATE_EXPS_2);
    if (nrowsStepBatch == 4) steps->step3_parser_first_avx(params, N, nrowsStepBatch);
    else if (nrowsStepBatch == 8) steps->step3_parser_first_avx512(params, N, nrowsStepBatch);
    else perRowStep(m, params, 2);
    TimerStopAndLog(STARK_STEP_3_CALCULATE_EXPS_2);
    TimerStart(STARK_STEP_3_LDE_AND_MERKLETREE);
    if (shardedStage(3, cols(cm3_n))) { // (scratch: cm1_n | cm2_n, their last reader has run)
        commitSharded(2, sec(cm3_n), dev0, cols(cm3_n), sec(cm3_2ns), nullptr, stageScratch(3), root2.address());
    } else {
        lend(sec(cm1_n), off(cm3_n) - off(cm1_n)); // cm1_n | cm2_n: their last reader has run
        if (m.tiledExt[2])
            mi::check(mi_lde_merkle_dev_tiled(c, d_nodes[2], sec(cm3_2ns), sec(cm3_n), cols(cm3_n), N, NExtended, cols(cm3_n)), "Starks::genProof (stage 3: extendPol + merkelize)");
        else {
            mi::check(mi_lde_dev(c, sec(cm3_2ns), cols(cm3_n), sec(cm3_n), cols(cm3_n), NExtended, N, cols(cm3_n)), "Starks::genProof (stage 3: extendPol)");
            mi::check(mi_merkle_build_dev(c, d_nodes[2], sec(cm3_2ns), cols(cm3_n), cols(cm3_n), NExtended), "Starks::genProof (stage 3: merkelize)");
        }
        mi::check(mi_copy_d2h(c, root2.address(), d_nodes[2] + treeElems - HASH_SIZE, HASH_SIZE * 8), "Starks::genProof (root 3)");
        mirrorRows(cm3_2ns);
    }
    transcript.put(root2.address(), HASH_SIZE);
    TimerStopAndLog(STARK_STEP_3_LDE_AND_MERKLETREE);
    TimerStopAndLog(STARK_STEP_3);

    //--------------------------------
    // 4. Compute C Polynomial  (starks.cpp:228-295).  The base-domain part of the image is dead from here on: it is re-planned as
    //    [const_2ns | x_2ns | qq1 | qq2 | LEv LpEv | xDivXSubXi xDivXSubWXi | FRI scratch ...], continuing in the late scratch
    //--------------------------------
    TimerStart(STARK_STEP_4);
    TimerStart(STARK_STEP_4_INIT);
    const uint64_t poolElems = off(cm1_2ns);
    uint64_t poolUsed = 0, lateUsed = 0;
    auto take = [&](uint64_t elems) -> uint64_t * { // the dead base-domain part first, then the late scratch (128-byte granules)
        const uint64_t e = (elems + 15) & ~15ULL;
        uint64_t *p = nullptr;
        if (!scratchElems && poolUsed + e <= poolElems) { p = d_mem + poolUsed; poolUsed += e; }
        else if (lateUsed + e <= scratchElems) { p = d_late + lateUsed; lateUsed += e; }
        else mi::fail("Starks::genProof (stage-4 scratch exhausted: plan error)");
        return p;
    };
    const uint64_t qDim = starkInfo.qDim, qDeg = starkInfo.qDeg, nConst = starkInfo.nConstants;
    m.d_const2ns = take(nConst * NExtended);
    m.d_x2ns = take(NExtended);
    uint64_t *qq1 = take(NExtended * qDim), *qq2 = take(NExtended * qDim * qDeg);
    uint64_t *lev = take(6 * N), *lpev = lev + 3 * N;
    m.d_xdiv = take(3 * NExtended);
    m.d_xdivw = take(3 * NExtended);
    uint64_t *d_evals = take(nEvals * 3 + 16);
    lateOffsets[0] = m.d_const2ns - d_mem; lateOffsets[1] = m.d_xdiv - d_mem; lateOffsets[2] = m.d_xdivw - d_mem;
    // what is left serves the transforms of stages 4 and 5, then FRI
    uint64_t *rest = scratchElems ? d_late + lateUsed : d_mem + poolUsed;
    const uint64_t restElems = scratchElems ? scratchElems - lateUsed : poolElems - poolUsed;
    lend(rest, restElems);
    // the tables a shard reads: its own, its group leader's, or -- grouped with shard 0 -- this device's (RowShardMem::aliasFrom)
    struct ShardTables { uint64_t *constN, *const2ns, *x2ns, *xdiv, *xdivw, *lev; };
    auto tablesOf = [&](size_t g) -> ShardTables {
        const RowShardMem &R = rowMem[g];
        if (R.aliasFrom < 0) return {R.constN, R.const2ns, R.x2ns, R.xdiv, R.xdiv + 3 * NExtended, R.lev};
        if (R.aliasFrom == 0) return {d_constN, m.d_const2ns, m.d_x2ns, m.d_xdiv, m.d_xdivw, lev};
        const RowShardMem &L = rowMem[R.aliasFrom];
        return {L.constN, L.const2ns, L.x2ns, L.xdiv, L.xdiv + 3 * NExtended, L.lev};
    };
    bool followsHome = false; // a shard reads THIS device's tables: they are computed here in full, and before that shard runs
    std::vector<size_t> tableOwners;
    for (size_t g = 1; g < rowMem.size(); g++) {
        followsHome = followsHome || rowMem[g].aliasFrom == 0;
        if (rowMem[g].ownsTables) tableOwners.push_back(g);
    }
    if (rowSharded && parserSteps) { // the other devices extend the constants and build x_2ns for themselves, beside this one (launches return at once)
        const uint64_t R = NExtended / rowMem.size();
        for (size_t g = 1; g < rowMem.size(); g++) {
            mi::check(mi_multi_set_device(mm, (int)g), "Starks::genProof (row shards: device)");
            mi_ctx *cg = mi_multi_ctx(mm, (int)g);
            if (rowMem[g].ownsTables && !rowMem[g].tablesReady) { // constants of the proving key: extended ONCE and kept (a device that holds 20 GB of a proof has the room)
                if (nConst) mi::check(mi_lde_dev(cg, rowMem[g].const2ns, nConst, rowMem[g].constN, nConst, NExtended, N, nConst), "Starks::genProof (row shards: constant polynomials, extended)");
                mi::check(mi_geom_seq_dev(cg, rowMem[g].x2ns, NExtended, Goldilocks::toU64(Goldilocks::shift()), Goldilocks::toU64(Goldilocks::w(nBitsExt))), "Starks::genProof (row shards: x_2ns)");
                rowMem[g].tablesReady = true;
            }
            const ShardTables T = tablesOf(g);
            mi::StarkMirror::RowShard S;
            S.shard = (int)g; S.d_mem = rowBase((int)g); S.d_const2ns = T.const2ns; S.d_x2ns = T.x2ns; S.d_xdiv = T.xdiv; S.d_xdivw = T.xdivw; S.row0 = g * R; S.rows = R; S.progs = &rowMem[g].progs;
            m.rowShards.push_back(S);
        }
        m.multi = mm;
        m.syncHomeFirst = followsHome;
        mi::check(mi_multi_set_device(mm, 0), "Starks::genProof (row shards: device)");
    }
    constRowsOnly = false;
    if (nConst && ownRowsOnly && !m.rowShards.empty() && !followsHome && !tableOwners.empty()) {
        // this device's image is planned to the last byte and cannot keep the extended constants from proof to proof; the other devices
        // can and do, and of all it would compute here (68 ms at zkEVM size) this device reads only its own rows and the halo: they come
        // across from the others' resident copies, a slice from each (seven links at once: 3.6 GB)
        const uint64_t need = NExtended / rowMem.size() + (1ULL << extendBits), G1 = tableOwners.size(), per = (need + G1 - 1) / G1;
        for (size_t i = 0; i < G1; i++) {
            const size_t g = tableOwners[i];
            const uint64_t r0 = i * per, r1 = std::min(need, r0 + per);
            if (r0 < r1) mi::check(mi_multi_copy(mm, m.d_const2ns + r0 * nConst, 0, rowMem[g].const2ns + r0 * nConst, (int)g, (r1 - r0) * nConst * 8), "Starks::genProof (constant polynomials: this device's rows from a peer)");
        }
        for (size_t g : tableOwners) mi::check(mi_multi_sync(mm, (int)g), "Starks::genProof (constant polynomials: rows arrived)");
        mi::check(mi_multi_set_device(mm, 0), "Starks::genProof (row shards: device)");
        constRowsOnly = true;
        constRowsFrom = (int)tableOwners[0];
    } else if (nConst && constNTiled) {
        // the resident constants are tile-major: a column chunk at a time goes row-major into the head of the remaining scratch and is extended from there
        // (a small STARK's remainder does not hold a chunk beside the transform's scratch -- lend() hands the context nothing then --: an allocation)
        const uint64_t cw0 = std::min<uint64_t>(nConst, 96), stage = (N * cw0 + 15) & ~15ULL;
        const bool fromRest = restElems >= stage + 16 * 2 * (N + NExtended);
        uint64_t *chunk = fromRest ? rest : mi::devAlloc(stage, "Starks::genProof (constant polynomials: a chunk row-major)");
        if (fromRest) lend(rest + stage, restElems - stage);
        for (uint64_t c0 = 0; c0 < nConst; c0 += cw0) {
            const uint64_t cw = std::min(cw0, nConst - c0);
            mi::check(mi_untile_dev(c, chunk, cw, d_constN, nConst, N, c0, 0, N, cw), "Starks::genProof (constant polynomials: a chunk row-major)");
            mi::check(mi_lde_dev(c, m.d_const2ns + c0, nConst, chunk, cw, NExtended, N, cw), "Starks::genProof (constant polynomials, extended)");
        }
        if (fromRest) lend(rest, restElems);
        else {
            mi::check(mi_ctx_sync(c), "Starks::genProof (constant polynomials, extended)");
            mi::devFree(chunk);
        }
    } else if (nConst) mi::check(mi_lde_dev(c, m.d_const2ns, nConst, d_constN, nConst, NExtended, N, nConst), "Starks::genProof (constant polynomials, extended)");
    mi::check(mi_geom_seq_dev(c, m.d_x2ns, NExtended, Goldilocks::toU64(Goldilocks::shift()), Goldilocks::toU64(Goldilocks::w(nBitsExt))),
              "Starks::genProof (x_2ns)"); // starks.hpp:155-160
    transcript.getField(challenges[4]); // gamma
    TimerStopAndLog(STARK_STEP_4_INIT);
    TimerStart(STARK_STEP_4_CALCULATE_EXPS_2NS);
    if (!deviceSteps) mi::check(mi_copy_d2h(c, x_2ns.address(), m.d_x2ns, NExtended * 8), "Starks::genProof (x_2ns d2h)");
    if (nrowsStepBatch == 4) steps->step42ns_parser_first_avx(params, NExtended, nrowsStepBatch);
    else if (nrowsStepBatch == 8) steps->step42ns_parser_first_avx512(params, NExtended, nrowsStepBatch);
    else perRowStep(m, params, 3);
    TimerStopAndLog(STARK_STEP_4_CALCULATE_EXPS_2NS);
    TimerStart(STARK_STEP_4_CALCULATE_EXPS_2NS_INTT);
    mi::check(mi_ntt_dev(c, qq1, qDim, sec(q_2ns), qDim, NExtended, qDim, 1), "Starks::genProof (INTT of q)");
    TimerStopAndLog(STARK_STEP_4_CALCULATE_EXPS_2NS_INTT);
    TimerStart(STARK_STEP_4_CALCULATE_EXPS_2NS_MUL);
    if (qDim != FIELD_EXTENSION) mi::fail("Starks::genProof (qDim != 3)");
    mi::check(mi_q_split_dev(c, qq2, qq1, N, NExtended, (unsigned)qDeg), "Starks::genProof (split of q)"); // starks.cpp:265-280
    TimerStopAndLog(STARK_STEP_4_CALCULATE_EXPS_2NS_MUL);
    TimerStart(STARK_STEP_4_CALCULATE_EXPS_2NS_NTT);
    mi::check(mi_ntt_dev(c, sec(cm4_2ns), qDim * qDeg, qq2, qDim * qDeg, NExtended, qDim * qDeg, 0), "Starks::genProof (NTT of the q chunks)");
    TimerStopAndLog(STARK_STEP_4_CALCULATE_EXPS_2NS_NTT);
    TimerStart(STARK_STEP_4_MERKLETREE);
    if (!m.rowShards.empty()) mirrorRows(cm4_2ns); // step52ns reads the quotient chunks at its rows
    mi::check(mi_merkle_build_dev(c, d_nodes[3], sec(cm4_2ns), cols(cm4_2ns), cols(cm4_2ns), NExtended), "Starks::genProof (stage 4: merkelize)");
    mi::check(mi_copy_d2h(c, root3.address(), d_nodes[3] + treeElems - HASH_SIZE, HASH_SIZE * 8), "Starks::genProof (root 4)");
    transcript.put(root3.address(), HASH_SIZE);
    TimerStopAndLog(STARK_STEP_4_MERKLETREE);
    TimerStopAndLog(STARK_STEP_4);

    //--------------------------------
    // 5. Compute FRI Polynomial  (starks.cpp:300-390)
    //--------------------------------
    TimerStart(STARK_STEP_5);
    TimerStart(STARK_STEP_5_LEv_LpEv);
    transcript.getField(challenges[7]); // xi
    Goldilocks::Element xi[3], wxi[3], xis[3], wxis[3];
    {
        const Goldilocks::Element sinv = Goldilocks::inv(Goldilocks::shift()), wN = Goldilocks::w(nBits);
        for (int d = 0; d < 3; d++) { // starks.cpp:316-318, 347-348
            xi[d] = challenges[7][d];
            wxi[d] = xi[d] * wN;
            xis[d] = xi[d] * sinv;
            wxis[d] = wxi[d] * sinv;
        }
    }
    mi::check(mi_geom_seq3_dev(c, lev, N, (const uint64_t *)xis), "Starks::genProof (LEv)");   // :320-324
    mi::check(mi_geom_seq3_dev(c, lpev, N, (const uint64_t *)wxis), "Starks::genProof (LpEv)");
    mi::check(mi_ntt_dev(c, lev, 3, lev, 3, N, 3, 1), "Starks::genProof (INTT LEv)");            // :325-326
    mi::check(mi_ntt_dev(c, lpev, 3, lpev, 3, N, 3, 1), "Starks::genProof (INTT LpEv)");
    TimerStopAndLog(STARK_STEP_5_LEv_LpEv);
    TimerStart(STARK_STEP_5_EVMAP);
    if (nEvals) { // starks.cpp:555-668
        std::vector<const uint64_t *> ptr(nEvals);
        std::vector<uint32_t> dim(nEvals);
        std::vector<uint64_t> stride(nEvals);
        std::vector<uint8_t> prime(nEvals);
        std::vector<uint64_t> tileCols(nEvals, 0); // a polynomial of a tile-major section: the section's width (its pointer: row 0 = section + 64 * column)
        for (uint64_t i = 0; i < nEvals; i++) {
            const EvMap &ev = starkInfo.evMap[i];
            if (ev.type == EvMap::eType::_const) { ptr[i] = m.d_const2ns + ev.id; stride[i] = nConst; dim[i] = 1; }
            else {
                const PolRef p = polRef(ev.type == EvMap::eType::cm ? starkInfo.cm_2ns[ev.id] : starkInfo.qs[ev.id]);
                ptr[i] = d_mem + p.offset; stride[i] = p.stride; dim[i] = (uint32_t)p.dim;
                for (int t = 0; t < 3; t++)
                    if (m.tiledExt[t] && p.offset >= off(s2[t]) && p.offset < off(s2[t]) + cols(s2[t])) { // (polRef: offset = the element of row 0)
                        ptr[i] = sec(s2[t]) + 64 * (p.offset - off(s2[t]));
                        tileCols[i] = cols(s2[t]);
                    }
            }
            prime[i] = ev.prime ? 1 : 0;
        }
        if (m.rowShards.empty()) {
            if (m.anyTiledExt())
                mi::check(mi_evmap_tiled_dev(c, d_evals, nEvals, N, (unsigned)extendBits, ptr.data(), dim.data(), stride.data(), prime.data(), tileCols.data(), lev, lpev), "Starks::genProof (evmap)");
            else
                mi::check(mi_evmap_dev(c, d_evals, nEvals, N, (unsigned)extendBits, ptr.data(), dim.data(), stride.data(), prime.data(), lev, lpev), "Starks::genProof (evmap)");
            mi::check(mi_copy_d2h(c, evals.address(), d_evals, nEvals * 3 * 8), "Starks::genProof (evals d2h)");
        } else {
            // row shards: a device's rows of the extension hold every 2^extendBits-th row the sums run over, so each device sums ITS base-domain
            // rows [g N/G, (g + 1) N/G) -- from its mirror, its constants and its slice of LEv / LpEv (75 MB at zkEVM size, sent across) --
            // and the G shares of 3 nEvals words are added here, in F_p^3
            const uint64_t G = rowMem.size(), nk = N / G;
            mi::check(mi_ctx_sync(c), "Starks::genProof (evmap shards: LEv / LpEv ready)");
            std::vector<std::vector<Goldilocks::Element>> share(G, std::vector<Goldilocks::Element>(nEvals * 3));
            for (const mi::StarkMirror::RowShard &S : m.rowShards) {
                if (rowMem[S.shard].aliasFrom == 0) continue; // reads this device's LEv / LpEv in place
                uint64_t *lg = tablesOf(S.shard).lev, *lpg = lg + 3 * N;
                const int owner = rowMem[S.shard].aliasFrom > 0 ? rowMem[S.shard].aliasFrom : S.shard; // (its leader's buffer: one per device)
                const uint64_t k0 = S.shard * nk;
                mi::check(mi_multi_copy(mm, lg + 3 * k0, owner, lev + 3 * k0, 0, nk * 3 * 8), "Starks::genProof (evmap shards: LEv slice)");
                mi::check(mi_multi_copy(mm, lpg + 3 * k0, owner, lpev + 3 * k0, 0, nk * 3 * 8), "Starks::genProof (evmap shards: LpEv slice)");
            }
            mi::check(mi_multi_sync(mm, 0), "Starks::genProof (evmap shards: slices sent)");
            for (const mi::StarkMirror::RowShard &S : m.rowShards) {
                std::vector<const uint64_t *> pg(nEvals);
                for (uint64_t i = 0; i < nEvals; i++) {
                    const EvMap &ev = starkInfo.evMap[i];
                    pg[i] = ev.type == EvMap::eType::_const ? S.d_const2ns + ev.id : S.d_mem + (ptr[i] - d_mem);
                }
                uint64_t *lg = tablesOf(S.shard).lev, *lpg = lg + 3 * N, *eg = rowMem[S.shard].evalsPart;
                mi::check(mi_multi_set_device(mm, S.shard), "Starks::genProof (evmap shards: device)");
                mi::check(mi_evmap_range_dev(mi_multi_ctx(mm, S.shard), eg, nEvals, N, (unsigned)extendBits, pg.data(), dim.data(), stride.data(), prime.data(), lg, lpg, S.shard * nk, nk),
                          "Starks::genProof (evmap shards)");
            }
            mi::check(mi_multi_set_device(mm, 0), "Starks::genProof (evmap shards: device)");
            mi::check(mi_evmap_range_dev(c, d_evals, nEvals, N, (unsigned)extendBits, ptr.data(), dim.data(), stride.data(), prime.data(), lev, lpev, 0, nk), "Starks::genProof (evmap, this device's rows)");
            mi::check(mi_copy_d2h(c, share[0].data(), d_evals, nEvals * 3 * 8), "Starks::genProof (evals d2h)");
            for (const mi::StarkMirror::RowShard &S : m.rowShards) {
                mi::check(mi_multi_set_device(mm, S.shard), "Starks::genProof (evmap shards: device)");
                mi::check(mi_copy_d2h(mi_multi_ctx(mm, S.shard), share[S.shard].data(), rowMem[S.shard].evalsPart, nEvals * 3 * 8), "Starks::genProof (evmap shards: share d2h)");
            }
            mi::check(mi_multi_set_device(mm, 0), "Starks::genProof (evmap shards: device)");
            for (uint64_t i = 0; i < nEvals; i++)
                for (int d = 0; d < 3; d++) {
                    Goldilocks::Element a = share[0][3 * i + d];
                    for (uint64_t g = 1; g < G; g++) a = a + share[g][3 * i + d];
                    evals[i][d] = a;
                }
        }
    }
    TimerStopAndLog(STARK_STEP_5_EVMAP);
    TimerStart(STARK_STEP_5_XDIVXSUB);
    // x / (x - xi), x / (x - w xi) (:350-365) depend on xi alone: queued BEFORE the evaluations go into the transcript -- that absorb is 663
    // chained permutations on one wave (10 ms at zkEVM size), on the transcript's own stream (mi_transcript_put), and these kernels run beside it
    for (const mi::StarkMirror::RowShard &S : m.rowShards) { // the other devices build their own tables (full height: addressing as here), beside this one
        mi::check(mi_multi_set_device(mm, S.shard), "Starks::genProof (row shards: device)");
        mi_ctx *cg = mi_multi_ctx(mm, S.shard);
        // (element-wise: a device fills the rows it evaluates, at their place in the full-height tables)
        mi::check(mi_x_div_x_sub_dev(cg, S.d_xdiv + 3 * S.row0, S.d_x2ns + S.row0, S.rows, (const uint64_t *)xi), "Starks::genProof (row shards: xDivXSubXi)");
        mi::check(mi_x_div_x_sub_dev(cg, S.d_xdivw + 3 * S.row0, S.d_x2ns + S.row0, S.rows, (const uint64_t *)wxi), "Starks::genProof (row shards: xDivXSubWXi)");
    }
    if (!m.rowShards.empty()) mi::check(mi_multi_set_device(mm, 0), "Starks::genProof (row shards: device)");
    const uint64_t xdivRows = m.rowShards.empty() ? NExtended : m.rowShards[0].row0; // row shards: this device's rows only
    mi::check(mi_x_div_x_sub_dev(c, m.d_xdiv, m.d_x2ns, xdivRows, (const uint64_t *)xi), "Starks::genProof (xDivXSubXi)");
    mi::check(mi_x_div_x_sub_dev(c, m.d_xdivw, m.d_x2ns, xdivRows, (const uint64_t *)wxi), "Starks::genProof (xDivXSubWXi)");
    // starks.cpp:342-345 puts evals[i] one by one; they are contiguous, so one put absorbs the same elements in the same order
    if (evals.offset() == FIELD_EXTENSION) { if (nEvals) transcript.put(evals[0], nEvals * FIELD_EXTENSION); }
    else for (uint64_t i = 0; i < nEvals; i++) transcript.put(evals[i], 3);
    transcript.getField(challenges[5]); // v1
    transcript.getField(challenges[6]); // v2
    TimerStopAndLog(STARK_STEP_5_XDIVXSUB);
    TimerStart(STARK_STEP_5_CALCULATE_EXPS);
    if (nrowsStepBatch == 4) steps->step52ns_parser_first_avx(params, NExtended, nrowsStepBatch);
    else if (nrowsStepBatch == 8) steps->step52ns_parser_first_avx512(params, NExtended, nrowsStepBatch);
    else perRowStep(m, params, 4);
    TimerStopAndLog(STARK_STEP_5_CALCULATE_EXPS);
    TimerStopAndLog(STARK_STEP_5);

    //--------------------------------
    // FRI over the resident f_2ns (starks.cpp:391-402): the four trees are lent to FRIProve as views, the fifth is the constant tree
    //--------------------------------
    TimerStart(STARK_STEP_FRI);
    {
        MerkleTreeGL views[4];
        MerkleTreeGL *trees[STARK_C12_A_NUM_TREES] = {&views[0], &views[1], &views[2], &views[3], treesGL[4]};
        for (int t = 0; t < 4; t++) {
            views[t].height = NExtended;
            views[t].width = cols(s2[t]);
            if (mtree[t]) {
                views[t].setMultiTree(sec(s2[t]), mtree[t]); // (the view owns the sharded tree and frees it)
                if (ownRowsOnly) {
                    std::vector<const uint64_t *> srcs(rowMem.size(), nullptr);
                    for (size_t g = 1; g < rowMem.size(); g++) srcs[g] = rowBase((int)g) + off(s2[t]);
                    views[t].setShardSources(mm, srcs, NExtended / rowMem.size());
                }
            }
            else views[t].setDeviceTree(sec(s2[t]), d_nodes[t], m.tiledExt[t]);
        }
        // FRI's polynomials, step trees and opening buffers come out of the same remainder (the fold transforms in registers: no
        // NTT scratch is in use any more)
        mi::check(mi_ctx_lend_workspace(c, nullptr, 0), "Starks::genProof (return the workspace)");
        mi::lendScratch(rest, restElems);
        FRIProve::prove(proof, trees, transcript, sec(f_2ns), nBitsExt, starkInfo);
        mi::lendScratch(nullptr, 0);
    }
    proof.proofs.setEvals(evals.address());
    std::memcpy(&proof.proofs.root1[0], root0.address(), HASH_SIZE * sizeof(Goldilocks::Element));
    std::memcpy(&proof.proofs.root2[0], root1.address(), HASH_SIZE * sizeof(Goldilocks::Element));
    std::memcpy(&proof.proofs.root3[0], root2.address(), HASH_SIZE * sizeof(Goldilocks::Element));
    std::memcpy(&proof.proofs.root4[0], root3.address(), HASH_SIZE * sizeof(Goldilocks::Element));
    mi::currentMirror() = nullptr;
    TimerStopAndLog(STARK_STEP_FRI);
}
#endif
