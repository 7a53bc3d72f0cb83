// pack_thp_probe.cpp -- are the host threads that pack a 32-column chunk of the witness (csrc/capi.hip upload_packed: 256 bytes out of
// every 5 320-byte row, one row per 4 KiB page) bound by TLB misses?  The same gather over the same matrix with 4 KiB pages and with
// transparent huge pages (madvise(MADV_HUGEPAGE) before the first touch).   g++ -O3 -mavx2 -pthread tools/pack_thp_probe.cpp -o /tmp/pk/p
//   usage: p <log2 rows> <threads> <huge 0|1>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <immintrin.h>
#include <sys/mman.h>
#include <thread>
#include <vector>
int main(int argc, char **argv)
{
    const uint64_t n = 1ull << (argc > 1 ? atoi(argv[1]) : 21), ncols = 665, cw = 32;
    const int T = argc > 2 ? atoi(argv[2]) : 8, huge = argc > 3 ? atoi(argv[3]) : 0;
    const uint64_t bytes = n * ncols * 8;
    uint64_t *m = (uint64_t *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    uint64_t *st = (uint64_t *)mmap(nullptr, n * cw * 8, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (m == MAP_FAILED || st == MAP_FAILED) return 1;
    if (huge) madvise(m, bytes, MADV_HUGEPAGE);
    auto par = [&](auto f) {
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++) th.emplace_back([=] { f(t); });
        for (auto &x : th) x.join();
    };
    const uint64_t per = (n + T - 1) / T;
    par([&](int t) { for (uint64_t r = t * per; r < std::min(n, (t + 1) * per); r++) for (uint64_t c = 0; c < ncols; c += 8) m[r * ncols + c] = r + c; }); // first touch
    memset(st, 1, n * cw * 8);
    double best = 1e9;
    for (int rep = 0; rep < 6; rep++) {
        const uint64_t c0 = 32 * (1 + rep);
        auto t0 = std::chrono::steady_clock::now();
        par([&](int t) {
            const uint64_t r0 = t * per, r1 = std::min(n, (t + 1) * per);
            const uint64_t *src = m + r0 * ncols + c0;
            uint64_t *dst = st + r0 * cw;
            for (uint64_t r = r0; r < r1; r++, src += ncols, dst += cw)
                for (uint64_t j = 0; j < cw; j += 8) {
                    const __m128i v0 = _mm_loadu_si128((const __m128i *)(src + j)), v1 = _mm_loadu_si128((const __m128i *)(src + j + 2));
                    const __m128i v2 = _mm_loadu_si128((const __m128i *)(src + j + 4)), v3 = _mm_loadu_si128((const __m128i *)(src + j + 6));
                    _mm_stream_si128((__m128i *)(dst + j), v0); _mm_stream_si128((__m128i *)(dst + j + 2), v1);
                    _mm_stream_si128((__m128i *)(dst + j + 4), v2); _mm_stream_si128((__m128i *)(dst + j + 6), v3);
                }
            _mm_sfence();
        });
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        best = std::min(best, s);
    }
    FILE *f = fopen("/proc/self/smaps_rollup", "r");
    char line[256];
    long anon_huge = -1;
    while (f && fgets(line, sizeof line, f)) if (!strncmp(line, "AnonHugePages:", 14)) anon_huge = atol(line + 14);
    printf("rows 2^%d, %d threads, huge %d (AnonHugePages %ld MB): one 32-column chunk packed in %.1f ms = %.2f GB/s of chunk bytes (%.2f per thread)\n",
           argc > 1 ? atoi(argv[1]) : 21, T, huge, anon_huge / 1024, best * 1e3, n * cw * 8 / best / 1e9, n * cw * 8 / best / 1e9 / T);
    return 0;
}
