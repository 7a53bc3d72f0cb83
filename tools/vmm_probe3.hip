// vmm_probe3.hip -- two simultaneous mappings of different sizes under one reservation (what refused mi_vmm_back in round 5?)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
static hipMemAllocationProp prop;
static hipMemAccessDesc acc;
static bool map_at(void *base, size_t off, size_t len, hipMemGenericAllocationHandle_t *h, bool set_access)
{
    hipError_t e = hipMemCreate(h, len, &prop, 0);
    if (e != hipSuccess) { printf("    create %zu: %s\n", len, hipGetErrorString(e)); (void)hipGetLastError(); return false; }
    e = hipMemMap((char *)base + off, len, 0, *h, 0);
    printf("    map %8.2f MiB at %10.2f MiB: %s", len / 1048576.0, off / 1048576.0, hipGetErrorString(e));
    if (e != hipSuccess) { printf("\n"); (void)hipGetLastError(); (void)hipMemRelease(*h); return false; }
    if (set_access) { e = hipMemSetAccess((char *)base + off, len, &acc, 1); printf(", access %s", hipGetErrorString(e)); }
    printf("\n");
    return true;
}
int main()
{
    (void)hipSetDevice(0);
    prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    const size_t GiB = 1ull << 30, MiB = 1ull << 20;
    struct Sc { const char *name; size_t o1, l1, o2, l2; bool acc_first; };
    const Sc scs[] = {
        {"74 MiB at 2 MiB, then 4 MiB at 400 GiB (the failing case)", 2 * MiB, 74 * MiB, 400 * GiB, 4 * MiB, true},
        {"64 MiB at 2 MiB, then 4 MiB at 400 GiB", 2 * MiB, 64 * MiB, 400 * GiB, 4 * MiB, true},
        {"74 MiB at 0, then 4 MiB at 400 GiB", 0, 74 * MiB, 400 * GiB, 4 * MiB, true},
        {"74 MiB at 2 MiB, then 74 MiB at 400 GiB", 2 * MiB, 74 * MiB, 400 * GiB, 74 * MiB, true},
        {"74 MiB at 2 MiB (no access set yet), then 4 MiB at 400 GiB", 2 * MiB, 74 * MiB, 400 * GiB, 4 * MiB, false},
        {"4 MiB at 400 GiB, then 74 MiB at 2 MiB (reverse order)", 400 * GiB, 4 * MiB, 2 * MiB, 74 * MiB, true},
        {"74 MiB at 2 MiB, then 4 MiB at 1 GiB", 2 * MiB, 74 * MiB, 1 * GiB, 4 * MiB, true},
        {"1 GiB at 0, then 1 GiB at 1 GiB, adjacent", 0, 1 * GiB, 1 * GiB, 1 * GiB, true},
    };
    for (const Sc &s : scs) {
        void *base = nullptr;
        if (hipMemAddressReserve(&base, 600 * GiB, 2 * MiB, nullptr, 0) != hipSuccess) { printf("reserve failed\n"); return 1; }
        printf("%s  (base %p)\n", s.name, base);
        hipMemGenericAllocationHandle_t h1, h2;
        const bool a = map_at(base, s.o1, s.l1, &h1, s.acc_first);
        const bool b = map_at(base, s.o2, s.l2, &h2, true);
        if (a) { (void)hipMemUnmap((char *)base + s.o1, s.l1); (void)hipMemRelease(h1); }
        if (b) { (void)hipMemUnmap((char *)base + s.o2, s.l2); (void)hipMemRelease(h2); }
        (void)hipMemAddressFree(base, 600 * GiB);
    }
    return 0;
}
