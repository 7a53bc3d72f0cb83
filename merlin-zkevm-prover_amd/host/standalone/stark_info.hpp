// stark_info.hpp (standalone stand-in) -- StarkInfo, the parsed <stark>.starkinfo.json, with the members Starks::genProof,
// FRIProve::prove and the generated Steps read (src/starkpil/stark_info.hpp:21-335, loader stark_info.cpp:20-447): the STARK's
// shape (starkStruct), the memory map of the polynomial area (mapOffsets / mapSectionsN / mapDeg / varPolMap / cm_n / cm_2ns / qs),
// the lookup / permutation / connection contexts that drive calculateH1H2 and calculateZ (puCtx / peCtx / ciCtx, exp2pol), and
// the evaluation map (evMap).  Same class, member and enumerator names, so host/starks.hpp compiles against either this file or
// the reference's own (which needs nlohmann/json and the prover's Config).
// Not loaded: the step2prev .. step52ns operation lists and exps_n / q_2ns / cm4_* / tmpExp_n index vectors -- pil-stark's input
// to its code generator; nothing in the prover reads them (the generated chelpers ARE those lists, compiled).
#ifndef STARK_INFO_HPP
#define STARK_INFO_HPP
#include <cstdint>
#include <map>
#include <string>
#include <vector>
#include "config.hpp"
#include "exit_process.hpp"
#include "goldilocks_base_field.hpp"
#include "merklehash_goldilocks.hpp"
#include "mi_json.hpp"
#include "polinomial.hpp"
#include "zklog.hpp"

class StepStruct
{
public:
    uint64_t nBits;
};

class StarkStruct
{
public:
    uint64_t nBits = 0;
    uint64_t nBitsExt = 0;
    uint64_t nQueries = 0;
    std::string verificationHashType = "GL";
    std::vector<StepStruct> steps;
};

typedef enum { cm1_n = 0, cm1_2ns = 1, cm2_n = 2, cm2_2ns = 3, cm3_n = 4, cm3_2ns = 5, cm4_n = 6, cm4_2ns = 7, tmpExp_n = 8, q_2ns = 9, f_2ns = 10, eSectionMax = 11 } eSection;

inline const char *sectionName(int s)
{
    static const char *names[eSectionMax] = {"cm1_n", "cm1_2ns", "cm2_n", "cm2_2ns", "cm3_n", "cm3_2ns", "cm4_n", "cm4_2ns", "tmpExp_n", "q_2ns", "f_2ns"};
    return names[s];
}
inline eSection string2section(const std::string s)
{
    for (int i = 0; i < eSectionMax; i++)
        if (s == sectionName(i)) return (eSection)i;
    zklog.error("string2section() found invalid string=" + s);
    exitProcess();
    return eSectionMax;
}

class PolsSections { public: uint64_t section[eSectionMax] = {}; };
class PolsSectionsVector { public: std::vector<uint64_t> section[eSectionMax]; };
class VarPolMap { public: eSection section; uint64_t dim; uint64_t sectionPos; };
class PeCtx { public: uint64_t tExpId, fExpId, zId, c1Id, numId, denId, c2Id; };
class PuCtx { public: uint64_t tExpId, fExpId, h1Id, h2Id, zId, c1Id, numId, denId, c2Id; };
class CiCtx { public: uint64_t zId, numId, denId, c1Id, c2Id; };
class EvMap
{
public:
    typedef enum { cm = 0, _const = 1, q = 2 } eType;
    eType type;
    uint64_t id;
    bool prime;
    void setType(std::string s)
    {
        if (s == "cm") type = cm;
        else if (s == "const") type = _const;
        else if (s == "q") type = q;
        else { zklog.error("EvMap::setType() found invalid type: " + s); exitProcess(); }
    }
};

class StarkInfo
{
public:
    StarkStruct starkStruct;
    uint64_t mapTotalN = 0, nConstants = 0, nPublics = 0, nCm1 = 0, nCm2 = 0, nCm3 = 0, nCm4 = 0, qDeg = 0, qDim = 0, friExpId = 0, nExps = 0;
    PolsSections mapDeg, mapOffsets, mapSectionsN, mapSectionsN1, mapSectionsN3;
    PolsSectionsVector mapSections;
    std::vector<VarPolMap> varPolMap;
    std::vector<uint64_t> qs, cm_n, cm_2ns;
    std::vector<PeCtx> peCtx;
    std::vector<PuCtx> puCtx;
    std::vector<CiCtx> ciCtx;
    std::vector<EvMap> evMap;
    std::map<std::string, uint64_t> exp2pol;

    StarkInfo() {}
    StarkInfo(const Config &config, std::string file)
    {
        if (!config.generateProof()) return; // stark_info.cpp:9-11
        try {
            load(mi::Json::parseFile(file));
        } catch (const std::exception &e) {
            zklog.error("StarkInfo::StarkInfo() cannot load " + file + ": " + e.what());
            exitProcess();
        }
    }

    void load(const mi::Json &j)
    {
        const mi::Json &ss = j["starkStruct"];
        starkStruct.nBits = ss["nBits"].u64();
        starkStruct.nBitsExt = ss["nBitsExt"].u64();
        starkStruct.nQueries = ss["nQueries"].u64();
        starkStruct.verificationHashType = ss["verificationHashType"].str();
        for (const mi::Json &s : ss["steps"].items) starkStruct.steps.push_back({s["nBits"].u64()});
        struct { const char *key; uint64_t *dst; } scalars[] = {{"mapTotalN", &mapTotalN}, {"nConstants", &nConstants}, {"nPublics", &nPublics}, {"nCm1", &nCm1},
                                                                 {"nCm2", &nCm2}, {"nCm3", &nCm3}, {"nCm4", &nCm4}, {"friExpId", &friExpId}, {"nExps", &nExps},
                                                                 {"qDim", &qDim}, {"qDeg", &qDeg}};
        for (auto &s : scalars) *s.dst = j[s.key].u64();
        struct { const char *key; PolsSections *dst; } maps[] = {{"mapDeg", &mapDeg}, {"mapOffsets", &mapOffsets}, {"mapSectionsN", &mapSectionsN},
                                                                  {"mapSectionsN1", &mapSectionsN1}, {"mapSectionsN3", &mapSectionsN3}};
        for (auto &m : maps)
            for (int s = 0; s < eSectionMax; s++) m.dst->section[s] = j[m.key][sectionName(s)].u64();
        for (int s = 0; s < eSectionMax; s++)
            for (const mi::Json &v : j["mapSections"][sectionName(s)].items) mapSections.section[s].push_back(v.u64());
        for (const mi::Json &v : j["varPolMap"].items) varPolMap.push_back({string2section(v["section"].str()), v["dim"].u64(), v["sectionPos"].u64()});
        for (const mi::Json &v : j["qs"].items) qs.push_back(v.u64());
        for (const mi::Json &v : j["cm_n"].items) cm_n.push_back(v.u64());
        for (const mi::Json &v : j["cm_2ns"].items) cm_2ns.push_back(v.u64());
        auto id = [](const mi::Json &o, const char *k) { return o[k].u64(); };
        for (const mi::Json &v : j["peCtx"].items) peCtx.push_back({id(v, "tExpId"), id(v, "fExpId"), id(v, "zId"), id(v, "c1Id"), id(v, "numId"), id(v, "denId"), id(v, "c2Id")});
        for (const mi::Json &v : j["puCtx"].items)
            puCtx.push_back({id(v, "tExpId"), id(v, "fExpId"), id(v, "h1Id"), id(v, "h2Id"), id(v, "zId"), id(v, "c1Id"), id(v, "numId"), id(v, "denId"), id(v, "c2Id")});
        for (const mi::Json &v : j["ciCtx"].items) ciCtx.push_back({id(v, "zId"), id(v, "numId"), id(v, "denId"), id(v, "c1Id"), id(v, "c2Id")});
        for (const mi::Json &v : j["evMap"].items) {
            EvMap e;
            e.setType(v["type"].str());
            e.id = v["id"].u64();
            e.prime = v["prime"].boolean();
            evMap.push_back(e);
        }
        if (const mi::Json *e2p = j.find("exp2pol"))
            for (const auto &m : e2p->members) exp2pol[m.first] = m.second.u64();
    }

    // stark_info.cpp:473-482: polynomial idPol as a strided view of the area -- element i at pAddress[offset + sectionPos + i * sectionCols]
    Polinomial getPolinomial(Goldilocks::Element *pAddress, uint64_t idPol)
    {
        const VarPolMap &p = varPolMap[idPol];
        return Polinomial(&pAddress[mapOffsets.section[p.section] + p.sectionPos], mapDeg.section[p.section], p.dim, mapSectionsN.section[p.section],
                          std::to_string(idPol));
    }
    uint64_t getPolSize(uint64_t polId) { return mapDeg.section[varPolMap[polId].section] * varPolMap[polId].dim * sizeof(Goldilocks::Element); }
    // bytes of the constant-tree file: [nPols, nExt, pols, nodes] (stark_info.hpp:328-335, build_const_tree.cpp:366-403)
    uint64_t getConstTreeSizeInBytes(void) const
    {
        const uint64_t NExtended = 1ULL << starkStruct.nBitsExt;
        return (nConstants * NExtended + MerklehashGoldilocks::getTreeNumElements(NExtended) + MERKLEHASHGOLDILOCKS_HEADER_SIZE) * sizeof(Goldilocks::Element);
    }
};
#endif
