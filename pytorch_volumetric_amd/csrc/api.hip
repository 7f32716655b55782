// Library identity entry points of the C ABI (include/pvamd.h).
#include "common.h"

extern "C" int pvamd_abi_version(void) { return PVAMD_ABI_VERSION; }

extern "C" const char* pvamd_build_info(void) {
    return "libpvamd gfx950 (MI355X/CDNA4) hipcc " __VERSION__ " built " __DATE__;
}

extern "C" int pvamd_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
