// local_g.hip -- bucket kernel instantiations: the group-tier kernel for column runs of 17 .. 32 entries (four lanes x 8 keys)
#include "local.hpp"

namespace esplocal {

#define ESP_LOCAL_GO(F, K)                                                                                              \
    do {                                                                                                                \
        hipLaunchKernelGGL((local_k<F, false, false, K, false, true, true>), dim3(grid), dim3(THREADS), 0, stream, a);  \
        return true;                                                                                                    \
    } while (0)

bool launch_group_short(const Variant &v, unsigned grid, hipStream_t stream, const Args &a) {
    if (v.keys == 0 && v.fresh) ESP_LOCAL_GO(true, 0);
    if (v.keys == 0 && !v.fresh) ESP_LOCAL_GO(false, 0);
    if (v.keys == 1 && v.fresh) ESP_LOCAL_GO(true, 1);
    if (v.keys == 1 && !v.fresh) ESP_LOCAL_GO(false, 1);
    if (v.keys == 2 && v.fresh) ESP_LOCAL_GO(true, 2);
    if (v.keys == 2 && !v.fresh) ESP_LOCAL_GO(false, 2);
    return false;
}

}  // namespace esplocal
