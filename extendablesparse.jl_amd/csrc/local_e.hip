// local_e.hip -- bucket kernel instantiations: per-source pieces, small variant
#include "local.hpp"

namespace esplocal {

#define ESP_LOCAL_GO(F, P, B, K, S)                                                                              \
    do {                                                                                                         \
        hipLaunchKernelGGL((local_k<F, P, B, K, S>), dim3(grid), dim3(THREADS), 0, stream, a);                   \
        return true;                                                                                             \
    } while (0)

bool launch_pieces_small(const Variant &v, unsigned grid, hipStream_t stream, const Args &a) {
    if (v.keys == 0 && v.fresh == true) ESP_LOCAL_GO(true, true, false, 0, true);
    if (v.keys == 0 && v.fresh == false) ESP_LOCAL_GO(false, true, false, 0, true);
    if (v.keys == 3 && v.fresh == true) ESP_LOCAL_GO(true, true, false, 3, true);
    if (v.keys == 3 && v.fresh == false) ESP_LOCAL_GO(false, true, false, 3, true);
    if (v.keys == 4 && v.fresh == true) ESP_LOCAL_GO(true, true, false, 4, true);
    if (v.keys == 4 && v.fresh == false) ESP_LOCAL_GO(false, true, false, 4, true);
    if (v.keys == 5 && v.fresh == true) ESP_LOCAL_GO(true, true, false, 5, true);
    if (v.keys == 5 && v.fresh == false) ESP_LOCAL_GO(false, true, false, 5, true);
    if (v.keys == 6 && v.fresh == true) ESP_LOCAL_GO(true, true, false, 6, true);
    if (v.keys == 6 && v.fresh == false) ESP_LOCAL_GO(false, true, false, 6, true);
    if (v.keys == 7 && v.fresh == true) ESP_LOCAL_GO(true, true, false, 7, true);
    if (v.keys == 7 && v.fresh == false) ESP_LOCAL_GO(false, true, false, 7, true);
    return false;
}

}  // namespace esplocal
