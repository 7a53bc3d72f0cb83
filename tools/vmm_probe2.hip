// vmm_probe2.hip -- which reservations / mappings does the driver take?  (round 5: a 4 MiB piece at offset 400 GiB of a 600 GiB range was
// refused with "invalid argument" while tools/vmm_probe.hip's 64 MiB pieces under 1 TiB were not.)
//   hipcc --offload-arch=gfx950 -O2 tools/vmm_probe2.hip -o /tmp/vmm_probe2 && /tmp/vmm_probe2
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
int main()
{
    hipSetDevice(0);
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gmin = 0, grec = 0;
    hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum);
    hipMemGetAllocationGranularity(&grec, &prop, hipMemAllocationGranularityRecommended);
    printf("granularity: minimum %zu, recommended %zu\n", gmin, grec);
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    const size_t GiB = 1ull << 30, MiB = 1ull << 20;
    for (size_t range : {600 * GiB, 512 * GiB, 1024 * GiB})
        for (size_t align : {(size_t)4096, 2 * MiB})
            for (size_t chunk : {4 * MiB, 64 * MiB, 1024 * MiB}) {
                void *base = nullptr;
                hipError_t e = hipMemAddressReserve(&base, range, align, nullptr, 0);
                if (e != hipSuccess) { printf("reserve %zu GiB align %zu: %s\n", range / GiB, align, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
                for (size_t off : {2 * MiB, range / 3 / (2 * MiB) * (2 * MiB), range / 3 * 2 / (2 * MiB) * (2 * MiB), range - chunk}) {
                    hipMemGenericAllocationHandle_t h;
                    e = hipMemCreate(&h, chunk, &prop, 0);
                    if (e != hipSuccess) { printf("  create %zu MiB: %s\n", chunk / MiB, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
                    e = hipMemMap((char *)base + off, chunk, 0, h, 0);
                    const char *m = hipGetErrorString(e);
                    hipError_t e2 = hipSuccess;
                    if (e == hipSuccess) e2 = hipMemSetAccess((char *)base + off, chunk, &acc, 1);
                    printf("range %4zu GiB (base %p) align %7zu chunk %4zu MiB at %7.1f GiB: map %s, access %s\n", range / GiB, base, align, chunk / MiB, (double)off / GiB, m,
                           e == hipSuccess ? hipGetErrorString(e2) : "-");
                    (void)hipGetLastError();
                    if (e == hipSuccess) hipMemUnmap((char *)base + off, chunk);
                    hipMemRelease(h);
                }
                hipMemAddressFree(base, range);
            }
    return 0;
}
