// stark_info.hpp -- the three StarkInfo fields the FRI path reads (stark_info.hpp:21-35: starkStruct.{nBits, nBitsExt,
// nQueries, verificationHashType, steps[].nBits}).  Same include guard as the reference's header, so a translation unit
// that already has the real StarkInfo (which needs nlohmann-json and the prover's config) keeps it and this file is inert.
#ifndef STARK_INFO_HPP
#define STARK_INFO_HPP
#include <cstdint>
#include <string>
#include <vector>

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

class StarkInfo
{
public:
    StarkStruct starkStruct;
};
#endif
