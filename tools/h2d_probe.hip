// h2d_probe.hip -- which engine moves a large page-locked host buffer to the device?  rocprofv3 of a zkEVM-size proof shows 969
// `__amd_rocclr_copyBuffer` launches of ~46 MB per proof (profiles/r04_starks_genproof_kernel_stats.csv): the stage-1 upload running as
// blit kernels on the CUs.  This probe times the same copy (2 GiB, hipHostMalloc'ed source, non-blocking stream) alone and beside a
// VALU-bound kernel, so that the runtime's switches (see tools/h2d_probe.sh) can be compared under `rocprofv3 --kernel-trace --stats`.
//   hipcc -O3 --offload-arch=gfx950 tools/h2d_probe.hip -o /tmp/h2d_probe && /tmp/h2d_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

// integer multiply-adds only: every CU's issue port busy, no memory traffic (the leaf-hash kernel's profile)
__global__ __launch_bounds__(256) void k_valu(uint64_t *out, uint32_t iters)
{
    uint64_t a = threadIdx.x + 1, b = blockIdx.x + 3, c = 7;
    for (uint32_t i = 0; i < iters; i++) {
        a = a * b + c; b = b * c + a; c = c * a + b;
        a = a * b + c; b = b * c + a; c = c * a + b;
    }
    if (a + b + c == 42) out[0] = a;
}

int main(int argc, char **argv)
{
    const uint64_t bytes = 2ull << 30;
    const int reps = argc > 1 ? std::atoi(argv[1]) : 6;
    char *host = nullptr, *dev = nullptr;
    uint64_t *sink = nullptr;
    CK(hipHostMalloc((void **)&host, bytes, hipHostMallocDefault));
    for (uint64_t i = 0; i < bytes; i += 4096) host[i] = (char)i;
    CK(hipMalloc((void **)&dev, bytes));
    CK(hipMalloc((void **)&sink, 64));
    hipStream_t cs, ks;
    CK(hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&ks, hipStreamNonBlocking));
    hipEvent_t e0, e1, k0, k1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&k0)); CK(hipEventCreate(&k1));
    float ms = 0;
    // 1. the copy alone
    CK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, cs));
    CK(hipStreamSynchronize(cs));
    CK(hipEventRecord(e0, cs));
    for (int r = 0; r < reps; r++) CK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, cs));
    CK(hipEventRecord(e1, cs));
    CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double copy_alone = ms / reps;
    std::printf("copy alone            %8.2f ms per 2 GiB  = %6.2f GB/s\n", copy_alone, bytes / copy_alone * 1e-6);
    // 2. the kernel alone (sized to last about as long as the copies)
    const unsigned blocks = 256 * 8 * 4;
    uint32_t iters = 20000;
    hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), 0, ks, sink, iters);
    CK(hipStreamSynchronize(ks));
    CK(hipEventRecord(k0, ks));
    hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), 0, ks, sink, iters);
    CK(hipEventRecord(k1, ks));
    CK(hipEventSynchronize(k1));
    CK(hipEventElapsedTime(&ms, k0, k1));
    iters = (uint32_t)(iters * (copy_alone * reps) / ms);
    CK(hipEventRecord(k0, ks));
    hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), 0, ks, sink, iters);
    CK(hipEventRecord(k1, ks));
    CK(hipEventSynchronize(k1));
    CK(hipEventElapsedTime(&ms, k0, k1));
    const double kern_alone = ms;
    std::printf("VALU kernel alone     %8.2f ms\n", kern_alone);
    // 3. both at once
    CK(hipEventRecord(k0, ks));
    hipLaunchKernelGGL(k_valu, dim3(blocks), dim3(256), 0, ks, sink, iters);
    CK(hipEventRecord(k1, ks));
    CK(hipEventRecord(e0, cs));
    for (int r = 0; r < reps; r++) CK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, cs));
    CK(hipEventRecord(e1, cs));
    CK(hipEventSynchronize(e1));
    CK(hipEventSynchronize(k1));
    float cms = 0, kms = 0;
    CK(hipEventElapsedTime(&cms, e0, e1));
    CK(hipEventElapsedTime(&kms, k0, k1));
    std::printf("together: copies      %8.2f ms per 2 GiB  = %6.2f GB/s (x%.3f)\n", cms / reps, bytes / (cms / reps) * 1e-6, cms / reps / copy_alone);
    std::printf("together: VALU kernel %8.2f ms (x%.3f)\n", kms, kms / kern_alone);
    return 0;
}
