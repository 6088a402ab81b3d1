// local_a.hip -- bucket kernel instantiations: one buffer per segment, regular variant (2 workgroups per CU)
#include "local.hpp"

namespace esplocal {

#define ESP_LOCAL_GO(F, P, B, K, S)                                                                              \
    do {                                                                                                         \
        hipLaunchKernelGGL((local_k<F, P, B, K, S>), dim3(grid), dim3(THREADS), 0, stream, a);                   \
        return true;                                                                                             \
    } while (0)

bool launch_regular(const Variant &v, unsigned grid, hipStream_t stream, const Args &a) {
    if (v.keys == 0) {
        if (v.fresh == true && v.big == true) ESP_LOCAL_GO(true, false, true, 0, false);
        if (v.fresh == true && v.big == false) ESP_LOCAL_GO(true, false, false, 0, false);
        if (v.fresh == false && v.big == true) ESP_LOCAL_GO(false, false, true, 0, false);
        if (v.fresh == false && v.big == false) ESP_LOCAL_GO(false, false, false, 0, false);
    }
    if (v.keys == 1) {
        if (v.fresh == true && v.big == true) ESP_LOCAL_GO(true, false, true, 1, false);
        if (v.fresh == true && v.big == false) ESP_LOCAL_GO(true, false, false, 1, false);
        if (v.fresh == false && v.big == true) ESP_LOCAL_GO(false, false, true, 1, false);
        if (v.fresh == false && v.big == false) ESP_LOCAL_GO(false, false, false, 1, false);
    }
    if (v.keys == 2) {
        if (v.fresh == true && v.big == true) ESP_LOCAL_GO(true, false, true, 2, false);
        if (v.fresh == true && v.big == false) ESP_LOCAL_GO(true, false, false, 2, false);
        if (v.fresh == false && v.big == true) ESP_LOCAL_GO(false, false, true, 2, false);
        if (v.fresh == false && v.big == false) ESP_LOCAL_GO(false, false, false, 2, false);
    }
    return false;
}

}  // namespace esplocal
