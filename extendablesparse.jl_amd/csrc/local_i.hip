// local_i.hip -- the bucket kernel for short columns with one wave per segment (wavecols.hpp)
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "wavecols.hpp"

namespace esplocal {

bool launch_wave(const Variant &v, unsigned grid, hipStream_t stream, const Args &a) {
    if (!v.fresh || v.pieces || (v.keys != 1 && v.keys != 2)) return false;
    // persistent workgroups: as many as the chip holds at once (each draws tickets until none is left)
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return false;
        ncu = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 1;
    }
    const bool n12 = v.wave_ni <= 12;
    const int W = v.wave_segs;
    // (workgroups per CU: 160 KiB of LDS, 9.4 / 12.4 KiB per wave)
    const int per_cu = std::max(1, (n12 ? 16 : 12) / W);
    const unsigned g = (unsigned)std::min<long long>((long long)grid, (long long)ncu * per_cu);
#define ESP_WAVE_GO(NI, WW) hipLaunchKernelGGL((wave_k<NI, WW>), dim3(g), dim3(WW * ESP_WAVE), 0, stream, a)
    if (W == 4) {
        if (n12) ESP_WAVE_GO(12, 4); else ESP_WAVE_GO(16, 4);
    } else if (W == 8) {
        if (n12) ESP_WAVE_GO(12, 8); else ESP_WAVE_GO(16, 8);
    } else if (W == 16) {
        if (n12) ESP_WAVE_GO(12, 16); else ESP_WAVE_GO(16, 12);
    } else {
        return false;
    }
#undef ESP_WAVE_GO
    return true;
}

}  // namespace esplocal
