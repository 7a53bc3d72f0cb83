// zklog.hpp (standalone stand-in) -- zklog.info / .error as the reference's call sites use them (zklog.hpp), to stderr.
#ifndef ZKLOG_HPP
#define ZKLOG_HPP
#include <cstdio>
#include <cstdlib>
#include <string>
class zkLog
{
public:
    void info(const std::string &m) { if (std::getenv("MI_STARK_VERBOSE")) std::fprintf(stderr, "%s\n", m.c_str()); }
    void warning(const std::string &m) { std::fprintf(stderr, "warning: %s\n", m.c_str()); }
    void error(const std::string &m) { std::fprintf(stderr, "error: %s\n", m.c_str()); }
};
inline zkLog &zklog_instance() { static zkLog z; return z; }
#define zklog (zklog_instance())
#endif
