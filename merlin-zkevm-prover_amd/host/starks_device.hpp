// LEGACY (round 2): superseded by host/starks.hpp (class Starks with the reference's signatures).  Kept because tests/cpp/test_starkpil_flow.cpp
// compares each device stage with the oracle through it; no new callers.
// starks_device.hpp -- device-resident driver of the hot-path stages of Starks::genProof.
//
// The Level-0 shims let src/starkpil compile unchanged, but then every extendPol / merkelize moves its operands over PCIe
// (stage 1 alone: 44.6 GB up, 89 GB down, 89 GB up again).  This class is the few lines a maintainer puts in place of
// the stage bodies of starks.cpp so that the extended sections (cm1_2ns .. cm4_2ns, q_2ns), their trees and the FRI
// polynomial never leave HBM:
//
//   starks.cpp:48-59    extendPol(p_cm1_2ns, p_cm1_n ..) + treesGL[0]->merkelize() + getRoot      -> commitStage(0, p_cm1_n, root)
//   starks.cpp:133-140  the same for cm2                                                          -> commitStage(1, ..)
//   starks.cpp:214-221  the same for cm3                                                          -> commitStage(2, ..)
//   starks.cpp:237-248  steps->step42ns_parser_first_avx(params, NExtended, nrowsStepBatch)       -> setStep42nsProgram + step42ns
//   starks.cpp:350-380  xDivXSubXi / xDivXSubWXi, steps->step52ns_parser_first_avx(params, ...)     -> setStep52nsProgram + step52ns
//   starks.cpp:261-292  INTT(qq1, q_2ns) / split / NTT(cm4_2ns, qq2) / treesGL[3]->merkelize()    -> commitQ(root)
//   starks.cpp:300-332  LEv / LpEv, evmap(pAddress, evals, LEv, LpEv)                              -> calculateEvals(evMap, xi, evals)
//   starks.cpp:393-394  FRIProve::prove(proof, treesGL, transcript, friPol, nBitsExt, starkInfo)   -> friProve(proof, transcript, starkInfo, constTree)
//   friProve.cpp:219-250 treesGL[t]->getGroupProof(..)                                            -> getGroupProofs(t, ..)
//
// and, with enableBaseDomain(), the base-domain half of stages 2 and 3 as well (cm1_n .. cm3_n, tmpExp_n resident: nothing but
// challenges and roots crosses PCIe between the upload of the witness and the openings):
//
//   starks.cpp:66-90, 150-170, 190-210  steps->step2prev / step3prev / step3_parser_first_avx      -> setBaseProgram + stepBase
//   starks.cpp:92-128   transposeH1H2Columns / calculateH1H2_opt1,3 / transposeH1H2Rows            -> calculateH1H2
//   starks.cpp:174-187  transposeZColumns / calculateZ / transposeZRows                            -> calculateZ
//   starks.cpp:133-140, 214-221  extendPol(p_cm2_2ns, p_cm2_n ..) + merkelize                       -> commitStageResident(1 / 2, root)
//
// One polynomial area in HBM holds the sections in the reference's order and row-major layout (element (row, col) of
// section s at area[offset_s + row * cols_s + col], stark_info.cpp:473-482), so that the constraint program's offsets --
// which are relative to the start of cm1_2ns here -- address it directly.  The host trace of a stage is streamed up in
// column chunks behind the kernels (mi_lde_merkle_host).  Errors follow the reference: log + exit.
#ifndef STARKS_DEVICE_HPP
#define STARKS_DEVICE_HPP
#include <utility>
#include <vector>
#include "goldilocks_base_field.hpp"
#include "goldilocks_cubic_extension.hpp"
#include "merklehash_goldilocks.hpp"
#include "mi_runtime.hpp"
#include "merkleTreeGL.hpp"
#include "transcript.hpp"
#include "friProve.hpp"

class StarksDevice
{
    uint64_t N, NExtended, nBits, nBitsExt;
    std::vector<uint64_t> cols, offset; // sections 0..2 = cm1..cm3 (_2ns), 3 = cm4_2ns (qDeg * qDim), 4 = q_2ns (qDim)
    uint64_t *d_area = nullptr;
    std::vector<uint64_t *> d_nodes;
    mi_chelpers_prog *prog42 = nullptr, *prog52 = nullptr;
    std::vector<std::pair<int, mi_chelpers_prog *>> progBase; // (step, program): step2prev, step3prev, step3
    std::vector<uint64_t> colsN, offsetN;                     // base-domain sections cm1_n, cm2_n, cm3_n, tmpExp_n
    uint64_t *d_baseArea = nullptr, *d_constN = nullptr, *d_xn = nullptr;
    uint64_t nConstN = 0;
    uint64_t nConst = 0;
    uint64_t *d_const = nullptr, *d_x2ns = nullptr, *d_xdiv = nullptr, *d_xdivw = nullptr, *d_f2ns = nullptr;

    static uint64_t *alloc(uint64_t elems, const char *what)
    {
        uint64_t *p = (uint64_t *)mi_dev_alloc(mi::ctx(), elems * 8);
        if (!p) mi::fail(what);
        return p;
    }
    // per-call scratch of the stages (coefficient buffers of q, LEv / LpEv, evaluations, opening buffers): grown once and kept -- memory
    // handed back to the driver is wiped in the background on the GPU's own bandwidth
    uint64_t *d_scratch = nullptr;
    uint64_t scratchElems = 0;
    uint64_t *scratch(uint64_t elems, const char *what)
    {
        if (elems > scratchElems) {
            if (d_scratch) mi_dev_free(mi::ctx(), d_scratch);
            d_scratch = alloc(elems, what);
            scratchElems = elems;
        }
        return d_scratch;
    }

public:
    // sectionCols: columns of cm1, cm2, cm3; qDeg, qDim as in StarkInfo (zkEVM: 665, 128, 371; 2, 3)
    StarksDevice(uint64_t _nBits, uint64_t _nBitsExt, const std::vector<uint64_t> &sectionCols, uint64_t qDeg = 2, uint64_t qDim = FIELD_EXTENSION)
        : N(1ULL << _nBits), NExtended(1ULL << _nBitsExt), nBits(_nBits), nBitsExt(_nBitsExt)
    {
        cols = sectionCols;
        cols.push_back(qDeg * qDim);
        cols.push_back(qDim);
        uint64_t o = 0;
        for (uint64_t c : cols) { offset.push_back(o); o += NExtended * c; }
        d_area = alloc(o, "StarksDevice (polynomial area)");
        d_nodes.assign(cols.size() - 1, nullptr);
        for (size_t i = 0; i + 1 < cols.size(); i++) d_nodes[i] = alloc(MerklehashGoldilocks::getTreeNumElements(NExtended), "StarksDevice (tree nodes)");
    }
    StarksDevice(const StarksDevice &) = delete;
    StarksDevice &operator=(const StarksDevice &) = delete;
    ~StarksDevice()
    {
        mi_ctx *c = mi::ctx();
        if (prog42) mi_chelpers_free(c, prog42);
        if (prog52) mi_chelpers_free(c, prog52);
        for (auto &pb : progBase) mi_chelpers_free(c, pb.second);
        mi_dev_free(c, d_scratch);
        mi_dev_free(c, d_baseArea); mi_dev_free(c, d_constN); mi_dev_free(c, d_xn);
        mi_dev_free(c, d_xdiv); mi_dev_free(c, d_xdivw); mi_dev_free(c, d_f2ns);
        for (uint64_t *p : d_nodes) mi_dev_free(c, p);
        mi_dev_free(c, d_area); mi_dev_free(c, d_const); mi_dev_free(c, d_x2ns);
    }
    uint64_t *section(unsigned s) { return d_area + offset[s]; } // device pointer: cm_{s+1}_2ns, cm4_2ns, q_2ns
    uint64_t sectionOffset(unsigned s) const { return offset[s]; }
    uint64_t *nodes(unsigned t) { return d_nodes[t]; }

    // stages 1-3: host trace section (N x cols[s], row-major) -> resident extension + tree; 32 bytes come back
    void commitStage(unsigned s, const Goldilocks::Element *p_cm_n, Goldilocks::Element *root)
    {
        mi_ctx *c = mi::ctx();
        mi::check(mi_lde_merkle_host(c, d_nodes[s], section(s), cols[s], (const uint64_t *)p_cm_n, N, NExtended, cols[s], 0), "StarksDevice::commitStage");
        getRoot(s, root);
    }
    void getRoot(unsigned t, Goldilocks::Element *root)
    {
        const uint64_t ne = MerklehashGoldilocks::getTreeNumElements(NExtended);
        mi::check(mi_copy_d2h(mi::ctx(), root, d_nodes[t] + ne - HASH_SIZE, HASH_SIZE * 8), "StarksDevice::getRoot");
    }

    // step 4a: the generated constraint program (tables of zkevm.chelpers.step42ns.parser.hpp) and what it reads besides the
    // committed sections: the extended constant polynomials (host, NExtended x nConst) and x_2ns (built here: shift * w^i)
    void setStep42nsProgram(const uint64_t *ops, uint64_t nops, const uint64_t *args, uint64_t nargs, const Goldilocks::Element *pConstPols2ns,
                            uint64_t _nConst)
    {
        mi_ctx *c = mi::ctx();
        nConst = _nConst;
        std::vector<mi_chelpers_section> secs;
        for (unsigned s = 0; s < 3; s++) secs.push_back({offset[s], cols[s], NExtended});
        mi::check(mi_chelpers_compile(c, &prog42, MI_CHELPERS_STEP42NS, ops, nops, args, nargs, secs.data(), secs.size(), nConst, NExtended),
                  "StarksDevice::setStep42nsProgram");
        if (nConst) {
            d_const = alloc(NExtended * nConst, "StarksDevice (constant polynomials)");
            mi::check(mi_copy_h2d(c, d_const, pConstPols2ns, NExtended * nConst * 8), "StarksDevice (constant polynomials h2d)");
        }
        d_x2ns = alloc(NExtended, "StarksDevice (x_2ns)");
        mi::check(mi_geom_seq_dev(c, d_x2ns, NExtended, Goldilocks::toU64(Goldilocks::shift()), Goldilocks::toU64(Goldilocks::w(nBitsExt))),
                  "StarksDevice (x_2ns)"); // starks.hpp:176-183
    }
    // step 4b: q_2ns = constraint polynomial over the extended domain (starks.cpp:237-248)
    void step42ns(const Goldilocks::Element *challenges, uint64_t nChallenges, const Goldilocks::Element *publicInputs, uint64_t nPublics)
    {
        mi_ctx *c = mi::ctx();
        std::vector<uint64_t> zh(1ULL << (nBitsExt - nBits));
        mi::check(mi_zhinv(c, zh.data(), (unsigned)nBits, (unsigned)nBitsExt), "StarksDevice::step42ns (ZhInv)");
        mi_chelpers_params p = {d_area, d_const, nConst, (const uint64_t *)challenges, nChallenges, (const uint64_t *)publicInputs, nPublics,
                                d_x2ns, 1, zh.data(), zh.size(), section(4)};
        mi::check(mi_chelpers_run_dev(c, prog42, &p, 0, NExtended), "StarksDevice::step42ns");
    }
    // Optional, once per proving key: compile the constraint programs set so far to gfx950 code (hiprtc; cacheDir keeps the code
    // objects between runs).  step42ns / step52ns then launch the compiled kernels instead of the interpreter; same results.
    void buildNative(const char *cacheDir = nullptr)
    {
        if (prog42) mi::check(mi_chelpers_build_native(prog42, cacheDir, 0), "StarksDevice::buildNative (step42ns)");
        if (prog52) mi::check(mi_chelpers_build_native(prog52, cacheDir, 0), "StarksDevice::buildNative (step52ns)");
    }
    // step 5a: the FRI-polynomial program (tables of zkevm.chelpers.step52ns.parser.hpp); call after setStep42nsProgram (it reads
    // the same constant polynomials) and after commitQ has filled cm4_2ns
    void setStep52nsProgram(const uint64_t *ops, uint64_t nops, const uint64_t *args, uint64_t nargs)
    {
        std::vector<mi_chelpers_section> secs;
        for (unsigned s = 0; s < 4; s++) secs.push_back({offset[s], cols[s], NExtended});
        mi::check(mi_chelpers_compile(mi::ctx(), &prog52, MI_CHELPERS_STEP52NS, ops, nops, args, nargs, secs.data(), secs.size(), nConst, NExtended),
                  "StarksDevice::setStep52nsProgram");
    }
    // step 5b (starks.cpp:350-380): xDivXSubXi / xDivXSubWXi over the extended domain, then f_2ns = the program's output, resident
    // (xi: challenges[7], wxi = xi * w(nBits): starks.cpp:308-309,352-353)
    uint64_t *step52ns(const Goldilocks::Element *challenges, uint64_t nChallenges, const Goldilocks::Element *evals, uint64_t nEvals,
                       const Goldilocks::Element *xi, const Goldilocks::Element *wxi)
    {
        mi_ctx *c = mi::ctx();
        if (!d_xdiv) {
            d_xdiv = alloc(NExtended * 3, "StarksDevice (xDivXSubXi)");
            d_xdivw = alloc(NExtended * 3, "StarksDevice (xDivXSubWXi)");
            d_f2ns = alloc(NExtended * 3, "StarksDevice (f_2ns)");
        }
        mi::check(mi_x_div_x_sub_dev(c, d_xdiv, d_x2ns, NExtended, (const uint64_t *)xi), "StarksDevice::step52ns (xDivXSubXi)");
        mi::check(mi_x_div_x_sub_dev(c, d_xdivw, d_x2ns, NExtended, (const uint64_t *)wxi), "StarksDevice::step52ns (xDivXSubWXi)");
        mi_chelpers_params p = {d_area, d_const, nConst, (const uint64_t *)challenges, nChallenges, nullptr, 0, nullptr, 0, nullptr, 0, nullptr,
                                (const uint64_t *)evals, nEvals, d_xdiv, d_xdivw, d_f2ns};
        mi::check(mi_chelpers_run_dev(c, prog52, &p, 0, NExtended), "StarksDevice::step52ns");
        return d_f2ns;
    }
    // step 4c (starks.cpp:261-292): INTT of q, split into qDeg chunks with shift^-N, NTT over qDeg * qDim columns, merkelize
    void commitQ(Goldilocks::Element *root, uint64_t qDeg = 2, uint64_t qDim = FIELD_EXTENSION)
    {
        mi_ctx *c = mi::ctx();
        uint64_t *qq1 = scratch(NExtended * qDim * (1 + qDeg), "StarksDevice::commitQ"), *qq2 = qq1 + NExtended * qDim;
        mi::check(mi_ntt_dev(c, qq1, qDim, section(4), qDim, NExtended, qDim, 1), "StarksDevice::commitQ (INTT)");
        mi::check(mi_q_split_dev(c, qq2, qq1, N, NExtended, (unsigned)qDeg), "StarksDevice::commitQ (split)");
        mi::check(mi_ntt_dev(c, section(3), qDim * qDeg, qq2, qDim * qDeg, NExtended, qDim * qDeg, 0), "StarksDevice::commitQ (NTT)");
        mi::check(mi_merkle_build_dev(c, d_nodes[3], section(3), qDim * qDeg, qDim * qDeg, NExtended), "StarksDevice::commitQ (merkelize)");
        getRoot(3, root);
    }
    // ---- stages 2 and 3 with the base domain in HBM.  Sections 0..2 = cm1_n .. cm3_n (the columns given to the constructor), 3 =
    // tmpExp_n; one area in that order, row-major (stark_info.cpp:473-482), offsets relative to cm1_n -- the offsets the base-domain
    // programs' arguments use.  pConstPolsN: the constant polynomials over the N base-domain rows (host, N x nConstN).
    void enableBaseDomain(uint64_t tmpExpCols, const Goldilocks::Element *pConstPolsN, uint64_t _nConstN)
    {
        mi_ctx *c = mi::ctx();
        colsN = {cols[0], cols[1], cols[2], tmpExpCols};
        uint64_t o = 0;
        offsetN.clear();
        for (uint64_t w : colsN) { offsetN.push_back(o); o += N * w; }
        d_baseArea = alloc(o, "StarksDevice (base-domain polynomial area)");
        nConstN = _nConstN;
        if (nConstN) {
            d_constN = alloc(N * nConstN, "StarksDevice (constant polynomials, base domain)");
            mi::check(mi_copy_h2d(c, d_constN, pConstPolsN, N * nConstN * 8), "StarksDevice (constant polynomials, base domain, h2d)");
        }
        d_xn = alloc(N, "StarksDevice (x_n)");
        mi::check(mi_geom_seq_dev(c, d_xn, N, 1, Goldilocks::toU64(Goldilocks::w(nBits))), "StarksDevice (x_n)"); // starks.hpp:149-160
    }
    uint64_t *baseSection(unsigned s) { return d_baseArea + offsetN[s]; }
    uint64_t baseOffset(unsigned s) const { return offsetN[s]; }
    // the witness (or any base-domain section a host step produced) into HBM
    void loadStage(unsigned s, const Goldilocks::Element *p_cm_n)
    {
        mi::check(mi_copy_h2d(mi::ctx(), baseSection(s), p_cm_n, N * colsN[s] * 8), "StarksDevice::loadStage");
    }
    // starks.cpp:52-59 / 133-140 / 214-221 with the base-domain section already in HBM
    void commitStageResident(unsigned s, Goldilocks::Element *root)
    {
        mi_ctx *c = mi::ctx();
        mi::check(mi_lde_dev(c, section(s), cols[s], baseSection(s), colsN[s], NExtended, N, cols[s]), "StarksDevice::commitStageResident (extendPol)");
        mi::check(mi_merkle_build_dev(c, d_nodes[s], section(s), cols[s], cols[s], NExtended), "StarksDevice::commitStageResident (merkelize)");
        getRoot(s, root);
    }
    // step = MI_CHELPERS_STEP2PREV / STEP3PREV / STEP3 with the tables of zkevm.chelpers.step{2prev,3prev,3}.parser.hpp; these
    // programs store into the polynomial area and exist as compiled kernels only, so the build (hiprtc, cached in cacheDir) is here
    void setBaseProgram(int step, const uint64_t *ops, uint64_t nops, const uint64_t *args, uint64_t nargs, const char *cacheDir = nullptr)
    {
        std::vector<mi_chelpers_section> secs;
        for (unsigned s = 0; s < 4; s++) secs.push_back({offsetN[s], colsN[s], N});
        mi_chelpers_prog *p = nullptr;
        mi::check(mi_chelpers_compile(mi::ctx(), &p, step, ops, nops, args, nargs, secs.data(), secs.size(), nConstN, N), "StarksDevice::setBaseProgram");
        mi::check(mi_chelpers_build_native(p, cacheDir, 0), "StarksDevice::setBaseProgram (build)");
        progBase.push_back({step, p});
    }
    void stepBase(int step, const Goldilocks::Element *challenges, uint64_t nChallenges, const Goldilocks::Element *publicInputs, uint64_t nPublics)
    {
        for (auto &pb : progBase)
            if (pb.first == step) {
                mi_chelpers_params p = {d_baseArea, d_constN, nConstN, (const uint64_t *)challenges, nChallenges, (const uint64_t *)publicInputs,
                                        nPublics, d_xn, 1, nullptr, 0, nullptr};
                mi::check(mi_chelpers_run_dev(mi::ctx(), pb.second, &p, 0, N), "StarksDevice::stepBase");
                return;
            }
        mi::fail("StarksDevice::stepBase: no program set for this step");
    }
    // One plookup (starks.cpp:106-126): offsets of the four polynomials in the base-domain area and their row strides, as
    // starkInfo.getPolinomial gives them (h1, h2 = cm_n[numCommited + 2 i], [.. + 1]; f, t = exp2pol of puCtx[i].fExpId / tExpId).
    // A value of f that t does not hold ends the process like the reference (log + exit).
    void calculateH1H2(uint64_t h1Off, uint64_t h1Stride, uint64_t h2Off, uint64_t h2Stride, uint64_t fOff, uint64_t fStride, uint64_t tOff,
                       uint64_t tStride, uint64_t dim)
    {
        mi::check(mi_calculate_h1h2_dev(mi::ctx(), d_baseArea + h1Off, h1Stride, d_baseArea + h2Off, h2Stride, d_baseArea + fOff, fStride,
                                        d_baseArea + tOff, tStride, (unsigned)dim, N), "StarksDevice::calculateH1H2");
    }
    // One grand product (starks.cpp:179-185); returns whether it closes (the reference zkasserts that)
    bool calculateZ(uint64_t zOff, uint64_t zStride, uint64_t numOff, uint64_t numStride, uint64_t denOff, uint64_t denStride)
    {
        int closes = 0;
        mi::check(mi_calculate_z_dev(mi::ctx(), d_baseArea + zOff, zStride, d_baseArea + numOff, numStride, d_baseArea + denOff, denStride, N,
                                     &closes), "StarksDevice::calculateZ");
        return closes != 0;
    }
    // ---- step 5: evaluations (starks.cpp:300-332).  One entry per starkInfo.evMap element: where the polynomial lives (a section of
    // the extended area, or the extended constant polynomials), its column, its dimension, and whether it is evaluated at w * xi.
    struct EvMapEntry { bool isConst; unsigned section; uint64_t column; uint64_t dim; bool prime; };
    void calculateEvals(const std::vector<EvMapEntry> &evMap, const Goldilocks::Element *xi, Goldilocks::Element *evals /* host, evMap.size() x 3 */)
    {
        mi_ctx *c = mi::ctx();
        Goldilocks::Element sinv = Goldilocks::inv(Goldilocks::shift()), wN = Goldilocks::w(nBits), xis[3], wxis[3];
        for (int d = 0; d < 3; d++) { xis[d] = xi[d] * sinv; wxis[d] = xi[d] * wN * sinv; } // :314-316
        const uint64_t n = evMap.size();
        uint64_t *lev = scratch(2 * N * 3 + n * 3 + 16, "StarksDevice::calculateEvals (LEv)"), *lpev = lev + N * 3;
        mi::check(mi_geom_seq3_dev(c, lev, N, (const uint64_t *)xis), "StarksDevice::calculateEvals (LEv)");     // :318-322
        mi::check(mi_geom_seq3_dev(c, lpev, N, (const uint64_t *)wxis), "StarksDevice::calculateEvals (LpEv)");
        mi::check(mi_ntt_dev(c, lev, 3, lev, 3, N, 3, 1), "StarksDevice::calculateEvals (INTT LEv)");             // :323-324
        mi::check(mi_ntt_dev(c, lpev, 3, lpev, 3, N, 3, 1), "StarksDevice::calculateEvals (INTT LpEv)");
        std::vector<const uint64_t *> ptr(n);
        std::vector<uint32_t> dim(n);
        std::vector<uint64_t> stride(n);
        std::vector<uint8_t> prime(n);
        for (uint64_t i = 0; i < n; i++) {
            const EvMapEntry &e = evMap[i];
            ptr[i] = e.isConst ? d_const + e.column : section(e.section) + e.column;
            stride[i] = e.isConst ? nConst : cols[e.section];
            dim[i] = (uint32_t)e.dim;
            prime[i] = e.prime ? 1 : 0;
        }
        uint64_t *d_evals = lev + 2 * N * 3;
        mi::check(mi_evmap_dev(c, d_evals, n, N, (unsigned)(nBitsExt - nBits), ptr.data(), dim.data(), stride.data(), prime.data(), lev, lpev),
                  "StarksDevice::calculateEvals (evmap)");
        mi::check(mi_copy_d2h(c, evals, d_evals, n * 3 * 8), "StarksDevice::calculateEvals (d2h)");
    }
    // ---- FRI over the resident f_2ns (starks.cpp:393-394): the four resident trees are lent to FRIProve as MerkleTreeGL views, the
    // fifth is the caller's constant tree (as treesGL[4] in the reference); transcript by value like there
    void friProve(FRIProof &proof, Transcript transcript, StarkInfo starkInfo, MerkleTreeGL *constTree)
    {
        if (!d_f2ns) mi::fail("StarksDevice::friProve: step52ns has not run");
        std::vector<MerkleTreeGL *> views;
        MerkleTreeGL *trees[5] = {nullptr, nullptr, nullptr, nullptr, constTree};
        for (unsigned t = 0; t < 4; t++) {
            MerkleTreeGL *v = new MerkleTreeGL();
            v->height = NExtended; v->width = cols[t];
            v->setDeviceTree(section(t), d_nodes[t]);
            trees[t] = v;
            views.push_back(v);
        }
        FRIProve::prove(proof, trees, transcript, d_f2ns, nBitsExt, starkInfo);
        for (MerkleTreeGL *v : views) delete v;
    }
    // friProve.cpp:219-250: proofs[q] = row idx[q] of tree t (cols[t] values) followed by nBitsExt x 4 siblings
    void getGroupProofs(unsigned t, Goldilocks::Element *proofs, const uint64_t *idx, uint64_t nq)
    {
        mi_ctx *c = mi::ctx();
        const uint64_t stride = cols[t] + nBitsExt * HASH_SIZE;
        uint64_t *d = scratch(nq * stride, "StarksDevice::getGroupProofs");
        mi::check(mi_merkle_group_proofs_dev(c, d, d_nodes[t], section(t), cols[t], NExtended, cols[t], idx, nq), "StarksDevice::getGroupProofs");
        mi::check(mi_copy_d2h(c, proofs, d, nq * stride * 8), "StarksDevice::getGroupProofs (d2h)");
    }
};
#endif
