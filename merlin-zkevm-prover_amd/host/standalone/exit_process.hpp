// exit_process.hpp (standalone stand-in) -- how the reference's call sites give up (src/utils/exit_process.cpp:7-22), without its
// call-stack print and its five-second grace period: the status the reference exits with, at once.  Defines the function (the
// reference's header only declares it, its definition lives in a file that needs gmp and json), so it may follow that header.
#pragma once
#include <cstdlib>
inline void exitProcess(void) { std::exit(-1); }
