// exit_process.hpp (standalone stand-in) -- src/utils/exit_process.cpp:7-22 without the call-stack print and the 5 s grace sleep
#ifndef EXIT_PROCESS_HPP
#define EXIT_PROCESS_HPP
#include <cstdlib>
inline void exitProcess(void) { std::exit(-1); }
#endif
