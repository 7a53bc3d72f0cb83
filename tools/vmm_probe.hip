// vmm_probe.hip -- does this stack's HIP virtual-memory management back a SPARSE full-height mirror?  Reserve a large address range, map
// physical memory only under a few scattered row ranges, touch them from a kernel, leave the rest unmapped.  (What host/starks.hpp's row
// shards would use instead of allocating 157 GB of which 20 GB are written; DESIGN.md section 8 "next".)
//   hipcc --offload-arch=gfx950 -O2 tools/vmm_probe.hip -o /tmp/vmm_probe && /tmp/vmm_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("FAIL %s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_touch(unsigned long long *p, unsigned long long n, unsigned long long tag)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = tag + i;
}
__global__ void k_sum(const unsigned long long *p, unsigned long long n, unsigned long long *out)
{
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(out, p[i]);
}

int main()
{
    CK(hipSetDevice(0));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    printf("granularity %zu bytes\n", gran);
    const size_t range = 1ull << 40; // 1 TiB of address space
    void *base = nullptr;
    CK(hipMemAddressReserve(&base, range, gran, nullptr, 0));
    printf("reserved 1 TiB at %p\n", base);
    const size_t chunk = ((64ull << 20) + gran - 1) / gran * gran; // 64 MiB pieces
    const size_t offs[4] = {0, 100ull << 30, 517ull << 30, range - chunk};
    hipMemGenericAllocationHandle_t h[4];
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    for (int i = 0; i < 4; i++) {
        CK(hipMemCreate(&h[i], chunk, &prop, 0));
        CK(hipMemMap((char *)base + offs[i], chunk, 0, h[i], 0));
        CK(hipMemSetAccess((char *)base + offs[i], chunk, &acc, 1));
    }
    unsigned long long *sum = nullptr;
    CK(hipMalloc((void **)&sum, 8));
    CK(hipMemset(sum, 0, 8));
    const unsigned long long n = chunk / 8;
    unsigned long long want = 0;
    for (int i = 0; i < 4; i++) {
        unsigned long long *p = (unsigned long long *)((char *)base + offs[i]);
        hipLaunchKernelGGL(k_touch, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, p, n, (unsigned long long)(i + 1) << 40);
        hipLaunchKernelGGL(k_sum, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, p, n, sum);
        want += n * ((unsigned long long)(i + 1) << 40) + n * (n - 1) / 2;
    }
    CK(hipDeviceSynchronize());
    unsigned long long got = 0;
    CK(hipMemcpy(&got, sum, 8, hipMemcpyDeviceToHost));
    printf("kernel wrote and read 4 x 64 MiB mapped at offsets 0, 100 GiB, 517 GiB, 1 TiB - 64 MiB: %s\n", got == want ? "ok" : "MISMATCH");
    // a copy engine through the mapping too (hipMemcpy2DAsync is what the commits use)
    unsigned long long hostv[4] = {};
    CK(hipMemcpy(hostv, (char *)base + offs[2] + 8 * 5, 32, hipMemcpyDeviceToHost));
    printf("copy out of the mapping: %s\n", hostv[0] == (3ull << 40) + 5 ? "ok" : "MISMATCH");
    size_t fr = 0, tot = 0;
    CK(hipMemGetInfo(&fr, &tot));
    printf("device memory in use after mapping 256 MiB under 1 TiB of addresses: %.2f GB\n", (tot - fr) / 1e9);
    for (int i = 0; i < 4; i++) { CK(hipMemUnmap((char *)base + offs[i], chunk)); CK(hipMemRelease(h[i])); }
    CK(hipMemAddressFree(base, range));
    printf("vmm probe done\n");
    return 0;
}
