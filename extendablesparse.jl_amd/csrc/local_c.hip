// local_c.hip -- bucket kernel instantiations: per-source pieces (column shards, batch + tail), fresh matrix
#include "local.hpp"

namespace esplocal {

#define ESP_LOCAL_GO(F, P, B, K, S)                                                                              \
    do {                                                                                                         \
        hipLaunchKernelGGL((local_k<F, P, B, K, S>), dim3(grid), dim3(THREADS), 0, stream, a);                   \
        return true;                                                                                             \
    } while (0)

bool launch_pieces_fresh(const Variant &v, unsigned grid, hipStream_t stream, const Args &a) {
    if (v.keys == 0 && v.big == true) ESP_LOCAL_GO(true, true, true, 0, false);
    if (v.keys == 0 && v.big == false) ESP_LOCAL_GO(true, true, false, 0, false);
    if (v.keys == 3 && v.big == true) ESP_LOCAL_GO(true, true, true, 3, false);
    if (v.keys == 3 && v.big == false) ESP_LOCAL_GO(true, true, false, 3, false);
    if (v.keys == 4 && v.big == true) ESP_LOCAL_GO(true, true, true, 4, false);
    if (v.keys == 4 && v.big == false) ESP_LOCAL_GO(true, true, false, 4, false);
    if (v.keys == 5 && v.big == true) ESP_LOCAL_GO(true, true, true, 5, false);
    if (v.keys == 5 && v.big == false) ESP_LOCAL_GO(true, true, false, 5, false);
    if (v.keys == 6 && v.big == true) ESP_LOCAL_GO(true, true, true, 6, false);
    if (v.keys == 6 && v.big == false) ESP_LOCAL_GO(true, true, false, 6, false);
    if (v.keys == 7 && v.big == true) ESP_LOCAL_GO(true, true, true, 7, false);
    if (v.keys == 7 && v.big == false) ESP_LOCAL_GO(true, true, false, 7, false);
    return false;
}

}  // namespace esplocal
