// utils.hpp (standalone stand-in) -- mapFile / copyFile / unmapFile as src/utils/utils.cpp gives them to Starks::Starks
// (starks.hpp:105-136): a read-only mapping or a heap copy of a whole file of a known size.  Failure = log + exitProcess().
#ifndef UTILS_HPP
#define UTILS_HPP
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fcntl.h>
#include <string>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include "exit_process.hpp"
#include "zklog.hpp"
inline void *mapFile(const std::string &fileName, uint64_t size, bool bOutput)
{
    const int fd = open(fileName.c_str(), bOutput ? (O_RDWR | O_CREAT) : O_RDONLY, 0666);
    if (fd < 0) { zklog.error("mapFile() cannot open " + fileName); exitProcess(); }
    if (bOutput && ftruncate(fd, (off_t)size) != 0) { zklog.error("mapFile() cannot size " + fileName); exitProcess(); }
    struct stat sb;
    if (fstat(fd, &sb) != 0 || (uint64_t)sb.st_size < size) {
        zklog.error("mapFile() found size of file " + fileName + " to be " + std::to_string((long long)sb.st_size) + " < " + std::to_string(size));
        exitProcess();
    }
    void *p = mmap(NULL, size, bOutput ? (PROT_READ | PROT_WRITE) : PROT_READ, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) { zklog.error("mapFile() failed calling mmap() of " + fileName); exitProcess(); }
    return p;
}
inline void unmapFile(void *pAddress, uint64_t size)
{
    if (munmap(pAddress, size) != 0) { zklog.error("unmapFile() failed calling munmap()"); exitProcess(); }
}
inline void *copyFile(const std::string &fileName, uint64_t size)
{
    FILE *f = std::fopen(fileName.c_str(), "rb");
    if (!f) { zklog.error("copyFile() cannot open " + fileName); exitProcess(); }
    void *p = std::malloc(size ? size : 1);
    if (!p || std::fread(p, 1, size, f) != size) { zklog.error("copyFile() cannot read " + std::to_string(size) + " bytes of " + fileName); exitProcess(); }
    std::fclose(f);
    return p;
}
#endif
