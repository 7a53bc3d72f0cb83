// zkassert.hpp (standalone stand-in) -- zkassert as src/utils/zkassert.hpp: checked in debug builds, gone under NDEBUG
#ifndef ZKASSERT_HPP
#define ZKASSERT_HPP
#include <cassert>
#include "exit_process.hpp"
#define zkassert(a) assert(a)
#endif
