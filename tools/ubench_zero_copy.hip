// ubench_zero_copy.hip -- can a kernel that reads page-locked host memory itself move a strided column chunk of a row-major host
// matrix faster than the DMA engines' 2-D copies (39.5 / 49 / 52.7 GB/s at 32 / 64 / 128 columns, profiles/r02_pcie_chunk_sweep.json)?
// Each lane moves 16 bytes per load; `inflight` independent loads are issued before the first store.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_zero_copy.hip -o tools/ubench_zero_copy && tools/ubench_zero_copy
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

template <int INFLIGHT>
__global__ __launch_bounds__(256) void k_gather(ulonglong2 *__restrict__ dst, const ulonglong2 *__restrict__ src, uint64_t src_pitch2,
                                                uint32_t seg2, uint64_t nrows)
{
    // seg2 = 16-byte pieces per row segment; src_pitch2 = row pitch in 16-byte units; piece p of the chunk = (row p / seg2, p % seg2)
    const uint64_t total = nrows * seg2, stride = (uint64_t)gridDim.x * 256;
    for (uint64_t p0 = (uint64_t)blockIdx.x * 256 + threadIdx.x; p0 < total; p0 += stride * INFLIGHT) {
        ulonglong2 v[INFLIGHT];
#pragma unroll
        for (int j = 0; j < INFLIGHT; j++) {
            const uint64_t p = p0 + j * stride;
            if (p < total) v[j] = src[(p / seg2) * src_pitch2 + (p % seg2)];
        }
#pragma unroll
        for (int j = 0; j < INFLIGHT; j++) {
            const uint64_t p = p0 + j * stride;
            if (p < total) dst[p] = v[j];
        }
    }
}

int main()
{
    const uint64_t rows = 1ull << 21, cols = 666; // even column count: 16-byte alignment of every row (the real trace has 665: see below)
    uint64_t *host = nullptr, *dev = nullptr;
    CK(hipHostMalloc((void **)&host, rows * cols * 8, hipHostMallocDefault));
    for (uint64_t i = 0; i < rows * cols; i += 512) host[i] = i;
    CK(hipMalloc((void **)&dev, rows * 256 * 8));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_it = [&](const char *what, uint64_t bytes, auto fn) {
        fn();
        CK(hipStreamSynchronize(s));
        CK(hipEventRecord(e0, s));
        for (int r = 0; r < 3; r++) fn();
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        std::printf("%-44s %7.2f GB/s\n", what, 3.0 * bytes / ms * 1e-6);
        std::fflush(stdout);
    };
    char name[128];
    for (uint32_t cw : {32u, 64u, 128u}) {
        const uint64_t bytes = rows * cw * 8;
        std::snprintf(name, sizeof name, "hipMemcpy2DAsync, %u columns", cw);
        time_it(name, bytes, [&] { CK(hipMemcpy2DAsync(dev, cw * 8, host + 8, cols * 8, cw * 8, rows, hipMemcpyHostToDevice, s)); });
        for (unsigned wgs : {32u, 128u, 512u}) {
            std::snprintf(name, sizeof name, "kernel gather, %u columns, %u workgroups x4", cw, wgs);
            time_it(name, bytes, [&] { hipLaunchKernelGGL(k_gather<4>, dim3(wgs), dim3(256), 0, s, (ulonglong2 *)dev, (const ulonglong2 *)(host + 8), cols / 2, cw / 2, rows); });
            std::snprintf(name, sizeof name, "kernel gather, %u columns, %u workgroups x8", cw, wgs);
            time_it(name, bytes, [&] { hipLaunchKernelGGL(k_gather<8>, dim3(wgs), dim3(256), 0, s, (ulonglong2 *)dev, (const ulonglong2 *)(host + 8), cols / 2, cw / 2, rows); });
        }
    }
    time_it("hipMemcpyAsync 1-D, 2 GiB", 2ull << 30, [&] { CK(hipMemcpyAsync(dev, host, 2ull << 30, hipMemcpyHostToDevice, s)); });
    return 0;
}
