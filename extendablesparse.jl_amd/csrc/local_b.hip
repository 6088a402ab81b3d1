// local_b.hip -- bucket kernel instantiations: one buffer per segment, small variant (3 workgroups per CU)
#include "local.hpp"

namespace esplocal {

#define ESP_LOCAL_GO(F, P, B, K, S)                                                                              \
    do {                                                                                                         \
        hipLaunchKernelGGL((local_k<F, P, B, K, S>), dim3(grid), dim3(THREADS), 0, stream, a);                   \
        return true;                                                                                             \
    } while (0)

bool launch_small(const Variant &v, unsigned grid, hipStream_t stream, const Args &a) {
    if (v.keys == 0 && v.fresh == true) ESP_LOCAL_GO(true, false, false, 0, true);
    if (v.keys == 0 && v.fresh == false) ESP_LOCAL_GO(false, false, false, 0, true);
    if (v.keys == 1 && v.fresh == true) ESP_LOCAL_GO(true, false, false, 1, true);
    if (v.keys == 1 && v.fresh == false) ESP_LOCAL_GO(false, false, false, 1, true);
    if (v.keys == 2 && v.fresh == true) ESP_LOCAL_GO(true, false, false, 2, true);
    if (v.keys == 2 && v.fresh == false) ESP_LOCAL_GO(false, false, false, 2, true);
    return false;
}

}  // namespace esplocal
