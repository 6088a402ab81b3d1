// local_j.hip -- group3_k fed with item records: the expansion of an item partition fused into the bucket kernel (group3_items.hpp)
#include "internal.hpp"
#include "group3_items.hpp"

namespace esplocal {

// false: no instantiation for this batch (the caller expands the items and takes the ordinary kernels)
bool launch_group3_items(const esp_handle::LazyItems &lz, unsigned grid, hipStream_t stream, const Args &a, bool hits) {
    ItemArgs ia;
    memset(&ia, 0, sizeof ia);
    ia.src = lz.src;
    if (lz.src == 1) {
        const espitem::Args &it = lz.it;
        if (!it.single) return false;
        ia.sorted = it.sorted_keys;
        ia.low = it.fem.L.rb + ESP_TAG_BITS;
        ia.fem = it.fem;
        if (it.fem.dim == 2 && hits)
            hipLaunchKernelGGL((group3_items_k<1, 3, true, true>), dim3(grid), dim3(THREADS), 0, stream, a, ia);
        else if (it.fem.dim == 2)
            hipLaunchKernelGGL((group3_items_k<1, 3, true>), dim3(grid), dim3(THREADS), 0, stream, a, ia);
        else if (hits)
            hipLaunchKernelGGL((group3_items_k<1, 4, true, true>), dim3(grid), dim3(THREADS), 0, stream, a, ia);
        else
            hipLaunchKernelGGL((group3_items_k<1, 4, true>), dim3(grid), dim3(THREADS), 0, stream, a, ia);
        return true;
    }
    if (lz.src == 2) {
        const espelem::Args &el = lz.el;
        if (!el.cellrec || (el.nloc != 3 && el.nloc != 4)) return false;
        ia.sorted = el.sorted_keys;
        ia.low = el.vrb + ESP_TAG_BITS;
        ia.elmat = el.elmat;
        ia.cellrec = el.cellrec;
        ia.negate = el.negate;
        const bool dg = el.diag != nullptr;
        if (hits) {
            if (el.nloc == 3 && dg)
                hipLaunchKernelGGL((group3_items_k<2, 3, true, true>), dim3(grid), dim3(THREADS), 0, stream, a, ia);
            else if (el.nloc == 3)
                hipLaunchKernelGGL((group3_items_k<2, 3, false, true>), dim3(grid), dim3(THREADS), 0, stream, a, ia);
            else if (dg)
                hipLaunchKernelGGL((group3_items_k<2, 4, true, true>), dim3(grid), dim3(THREADS), 0, stream, a, ia);
            else
                hipLaunchKernelGGL((group3_items_k<2, 4, false, true>), dim3(grid), dim3(THREADS), 0, stream, a, ia);
            return true;
        }
        if (el.nloc == 3 && dg)
            hipLaunchKernelGGL((group3_items_k<2, 3, true>), dim3(grid), dim3(THREADS), 0, stream, a, ia);
        else if (el.nloc == 3)
            hipLaunchKernelGGL((group3_items_k<2, 3, false>), dim3(grid), dim3(THREADS), 0, stream, a, ia);
        else if (dg)
            hipLaunchKernelGGL((group3_items_k<2, 4, true>), dim3(grid), dim3(THREADS), 0, stream, a, ia);
        else
            hipLaunchKernelGGL((group3_items_k<2, 4, false>), dim3(grid), dim3(THREADS), 0, stream, a, ia);
        return true;
    }
    return false;
}

// Base.sum over several buffers: ONE launch over the non-empty (buffer, segment) pairs (group3_items_k<2, ., ., false, MULTI>)
bool launch_group3_items_multi(int nloc, bool diag, const u32 *vlist, const MultiBuf *mbuf, i64 *counts, int S_real, unsigned grid,
                               hipStream_t stream, const Args &a) {
    ItemArgs ia;
    memset(&ia, 0, sizeof ia);
    ia.src = 2;
    ia.vlist = vlist;
    ia.mbuf = mbuf;
    ia.counts = counts;
    ia.S_real = S_real;
    if (nloc == 3 && diag)
        hipLaunchKernelGGL((group3_items_k<2, 3, true, false, true>), dim3(grid), dim3(THREADS), 0, stream, a, ia);
    else if (nloc == 3)
        hipLaunchKernelGGL((group3_items_k<2, 3, false, false, true>), dim3(grid), dim3(THREADS), 0, stream, a, ia);
    else if (nloc == 4 && diag)
        hipLaunchKernelGGL((group3_items_k<2, 4, true, false, true>), dim3(grid), dim3(THREADS), 0, stream, a, ia);
    else if (nloc == 4)
        hipLaunchKernelGGL((group3_items_k<2, 4, false, false, true>), dim3(grid), dim3(THREADS), 0, stream, a, ia);
    else
        return false;
    return true;
}

}  // namespace esplocal
