// config.hpp (standalone stand-in) -- the three members of the prover's Config that src/starkpil reads (config.hpp:113,119,171).
// Only on the include path where the reference tree is absent (GPU-box tests, standalone tools); a maintainer's build keeps the
// reference's own config.hpp and never sees this file.
#ifndef CONFIG_HPP
#define CONFIG_HPP
#include <string>
class Config
{
public:
    bool runFileGenBatchProof = true;
    std::string zkevmConstPols, zkevmConstantsTree, zkevmStarkInfo; // config.hpp: the three files a Starks is built from (prover.cpp:128)
    bool mapConstPolsFile = false;
    bool mapConstantsTreeFile = false;
    bool generateProof(void) const { return runFileGenBatchProof; }
};
#endif
