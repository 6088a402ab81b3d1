// local_i.hip -- the bucket kernel for short columns with one wave per segment (wavecols.hpp)
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "wavecols.hpp"

namespace esplocal {

bool launch_wave(const Variant &v, unsigned grid, hipStream_t stream, const Args &a) {
    if (!v.fresh || v.pieces || (v.keys != 1 && v.keys != 2)) return false;
    // persistent workgroups: as many as the chip holds at once (each draws tickets until none is left)
    static int per_cu[2] = {0, 0}, ncu = 0;
    if (ncu == 0) {
        int dev = 0;
        hipDeviceProp_t pr;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&pr, dev) != hipSuccess) return false;
        int b12 = 0, b16 = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b12, wave_k<12>, WV_THREADS, 0) != hipSuccess) b12 = 1;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&b16, wave_k<16>, WV_THREADS, 0) != hipSuccess) b16 = 1;
        per_cu[0] = b12 > 0 ? b12 : 1;
        per_cu[1] = b16 > 0 ? b16 : 1;
        ncu = pr.multiProcessorCount > 0 ? pr.multiProcessorCount : 1;
        if (const char *e = getenv("ESP_WAVE_PER_CU")) per_cu[0] = per_cu[1] = std::max(1, atoi(e));  // (experiments)
        if (getenv("ESP_WAVE_DEBUG")) fprintf(stderr, "wave_k: %d CUs, %d / %d workgroups per CU (12 / 16 entries per lane)\n", ncu, per_cu[0], per_cu[1]);
    }
    const bool n12 = v.wave_ni <= 12;
    const unsigned g = (unsigned)std::min<long long>((long long)grid, (long long)ncu * per_cu[n12 ? 0 : 1]);
    if (n12)
        hipLaunchKernelGGL((wave_k<12>), dim3(g), dim3(WV_THREADS), 0, stream, a);
    else
        hipLaunchKernelGGL((wave_k<16>), dim3(g), dim3(WV_THREADS), 0, stream, a);
    return true;
}

}  // namespace esplocal
