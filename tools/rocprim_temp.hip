// rocprim::radix_sort_pairs temporary storage as a function of the number of (uint32 key, int32 value) pairs, 21 key bits:
// the bound PVAMD_ORDER_LIBRARY_TEMP_BYTES(P) of include/pvamd.h must hold for every P.
//   hipcc --offload-arch=gfx950 -O2 tools/rocprim_temp.hip -o tools/rocprim_temp.bin && ./tools/rocprim_temp.bin
#include <cstring>
#include <cstdio>
#include <rocprim/device/device_radix_sort.hpp>
#include "../include/pvamd.h"
int main() {
    const long long sizes[] = {1572864, 2097152, 4194304, 16777216, 67108864, 268435456, 1073741824, 2147483647};
    for (long long P : sizes) {
        size_t need = 0;
        unsigned* k = nullptr; int* v = nullptr;
        hipError_t e = rocprim::radix_sort_pairs(nullptr, need, k, k, v, v, (size_t)P, 0u, 21u, (hipStream_t)0);
        printf("P %lld: rocprim needs %zu bytes (%.3f per pair) | header bound %lld | %s (hip %d)\n", P, need, (double)need / P,
               (long long)PVAMD_ORDER_LIBRARY_TEMP_BYTES(P), need <= (size_t)PVAMD_ORDER_LIBRARY_TEMP_BYTES(P) ? "ok" : "TOO SMALL", (int)e);
    }
    return 0;
}
