// local_h.hip -- the group tier as a kernel of its own with three workgroups per CU (group3.hpp)
#include "group3.hpp"

namespace esplocal {

bool launch_group3(const Variant &v, unsigned grid, hipStream_t stream, const Args &a) {
    if (v.g3hits) {  // (additions over a stored pattern the same mesh built: the sums go to a second value array, all-or-nothing)
        if (v.fresh || v.pieces || (v.keys != 1 && v.keys != 2)) return false;
        if (v.g3wide && v.keys == 1)
            hipLaunchKernelGGL((group3_k<1, ITEMS, true, true>), dim3(grid), dim3(THREADS), 0, stream, a);
        else if (v.g3wide)
            hipLaunchKernelGGL((group3_k<2, ITEMS, true, true>), dim3(grid), dim3(THREADS), 0, stream, a);
        else if (v.keys == 1)
            hipLaunchKernelGGL((group3_k<1, ITEMS, false, true>), dim3(grid), dim3(THREADS), 0, stream, a);
        else
            hipLaunchKernelGGL((group3_k<2, ITEMS, false, true>), dim3(grid), dim3(THREADS), 0, stream, a);
        return true;
    }
    if (!v.fresh || v.pieces) return false;
    if (v.g3wide) {  // (rows of a segment anywhere in the matrix: full rows in LDS, every run sorted twice)
        if (v.keys == 1)
            hipLaunchKernelGGL((group3_k<1, ITEMS, true>), dim3(grid), dim3(THREADS), 0, stream, a);
        else if (v.keys == 2)
            hipLaunchKernelGGL((group3_k<2, ITEMS, true>), dim3(grid), dim3(THREADS), 0, stream, a);
        else
            return false;
        return true;
    }
    if (v.g3k64) {  // (packed keys of one kind: Args::kind32 says which)
        if (a.kind32 == (u32)ESP_UPDATE)
            hipLaunchKernelGGL((group3_k<2, ITEMS, false, false, true>), dim3(grid), dim3(THREADS), 0, stream, a);
        else
            hipLaunchKernelGGL((group3_k<1, ITEMS, false, false, true>), dim3(grid), dim3(THREADS), 0, stream, a);
        return true;
    }
    if (v.keys == 1) {
        hipLaunchKernelGGL((group3_k<1>), dim3(grid), dim3(THREADS), 0, stream, a);
        return true;
    }
    if (v.keys == 2) {
        hipLaunchKernelGGL((group3_k<2>), dim3(grid), dim3(THREADS), 0, stream, a);
        return true;
    }
    return false;
}

}  // namespace esplocal
