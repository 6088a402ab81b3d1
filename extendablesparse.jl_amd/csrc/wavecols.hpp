// wavecols.hpp -- the bucket kernel for SHORT columns, one WAVE per segment (round 4).
// local_k (local.hpp) moves a segment of <= 4096 entries over <= 2048 columns through a workgroup of eight waves: a
// counting sort by column over the whole workgroup, then ONE LANE PER COLUMN sorts and folds its run -- with 256 columns
// and 512 lanes half the workgroup idles through the sort and the fold, every phase ends at a workgroup barrier, and the
// kernel is bound by the instructions it issues (1311 per wave for 6 entries per lane), not by its 4.5 GB.
// Here the partition in front is asked for segments of at most 64 whole columns and at most 64 * NI entries (for the
// producers whose append is the partition that costs nothing: the cut is two bits finer), and a segment belongs to one
// wave: 64 lanes = 64 columns, every lane busy in every phase, no workgroup barrier between load and fold -- LDS
// operations of one wave complete in order, a compiler fence is all the phases need between them.  Sort keys are 32 bits
// (row - smallest row of the segment) << 10 | slot: v_min_u32 / v_max_u32 networks of 42 / 63 comparators, the fold's
// records stay in registers under static indices and go to their dense LDS position once the lanes' counts are scanned
// (DPP), the lane writes its own column's colptr.  Four waves = four consecutive segments share a workgroup for ONE
// reason: the look-back chain over the output offsets runs per workgroup (one ticket, one granule, two barriers).
// Serves: a fresh matrix, 4-byte keys of one kind (UPDATE or RAWUPDATE), segments of whole columns (<= 64) that write
// colptr themselves, column runs of at most 16 entries, rows of a segment within 2^22 of each other.  A segment outside that
// reports bit 8 of Args::err (bit 16 beside it: for its rows) and emits nothing; the host then runs the flush again with
// local_k (a fresh-matrix flush has changed nothing) and the handle remembers.
//
// MEASURED (MI355X, 256^3 stencil, same box as local_k's small variant at 1.37 ms -- before local_k's own look-back work,
// which brought THAT kernel to 1.03-1.15 ms; tools/r4_wave.sh, tools/xcc_probe.hip):
//   one workgroup per ticket (no loop)                       1.41 ms   818 VALU per wave (768 entries) against 8 x 722 per 3072
//   persistent, tickets and entries requested ahead          1.50 ms   (1.78 with the ticket counter beside the granules)
//   ... tickets dealt round-robin, no atomic (-DESP_WAVE_STATIC: safe only while every workgroup is resident)   1.14 ms
//   ... and no look-back polls (wrong results)               1.00 ms
// i.e. the kernel is bound by what synchronises it, not by its instructions or bytes: a draw from ONE counter costs 12 ns
// when the whole chip draws (83 M/s: the counter's line travels between the eight L2s; one counter per XCD: 1.9 ns), and draws
// and polls delay each other.  A pool of tickets per XCD (blocks of 16 claimed from the counter, handed out through a word only
// that XCD touches) was built and dropped: whoever refills a pool waits for a cross-XCD round trip while it HOLDS a ticket,
// and every ticket above waits for it in its look-back -- 13 ms.  The finer cut costs the producer 0.17 ms (0.72 against
// 0.55 ms: a dozen runs per tile instead of four).  With 16 segments per ticket (one workgroup of 16 waves per CU) it runs
// 1.19-1.25 ms: slower than local_k is now, behind a costlier producer -- it stays opt-in (ESP_WAVE=1), the record of how
// the bucket kernels' real bound was found and a vehicle for further work on the synchronisation.
#pragma once
#include "local.hpp"

namespace esplocal {

constexpr int WV_CL_BITS = 6;    // at most 64 columns per segment: one lane each
constexpr int WV_IDX_BITS = 10;  // slot index inside the segment (<= 1024 entries)
constexpr int WV_ROW_BITS = 32 - WV_IDX_BITS;
constexpr int WV_MAXRUN = 16;
constexpr int WV_WIN = 32;  // workgroups around the expected ticket whose segment starts are fetched in advance
constexpr u32 WV_PAD = 0xFFFFFC00u;  // sorts behind every real key (rows of a segment span less than 2^22 - 2); slot 0

__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// one column run per lane: sort, fold, count; the records stay in x[] (row relative to the segment's smallest) / xv[]
// under the bits of `emit`
template <int R>
__device__ __forceinline__ void wave_run(const u32 *sk, const double *sv, int rs, int len, bool raw, u32 (&x)[R], double (&xv)[R],
                                         u32 &emit) {
#pragma unroll
    for (int j = 0; j < R; j++) x[j] = sk[rs + j];  // (reads past the run's end stay inside the padded array)
#pragma unroll
    for (int j = 0; j < R; j++) x[j] = j < len ? x[j] : WV_PAD;
#pragma unroll
    for (int q = 0; q < NetOf<R>::net.n; q++) {
        const u32 lo = x[NetOf<R>::net.a[q]], hi = x[NetOf<R>::net.b[q]];
        x[NetOf<R>::net.a[q]] = min(lo, hi);
        x[NetOf<R>::net.b[q]] = max(lo, hi);
    }
#pragma unroll
    for (int j = 0; j < R; j++) xv[j] = sv[x[j] & ((1u << WV_IDX_BITS) - 1u)];
#pragma unroll
    for (int j = 0; j < R; j++) x[j] >>= WV_IDX_BITS;
    // ordered fold (fold.hpp: fold_step_update / RAWUPDATE): a (col,row) starts from +0.0, is present once an update creates
    // it; the record of a group is taken where the group ends
    bool pres = false;
    double acc = 0.0;
    emit = 0;
#pragma unroll
    for (int j = 0; j < R; j++) {
        const bool same = j > 0 && x[j] == x[j - 1];
        acc = (same ? acc : 0.0) + xv[j];
        pres = (same && pres) || raw || xv[j] != 0.0;
        const bool closes = j < len && (j == R - 1 || x[j + 1] != x[j]);  // (a padded key differs from every real one)
        emit |= (closes && pres) ? (1u << j) : 0u;
        xv[j] = acc;
    }
}

template <int R>
__device__ __forceinline__ void wave_put(u32 *sk, double *sv, u32 at, const u32 (&x)[R], const double (&xv)[R], u32 emit) {
#pragma unroll
    for (int j = 0; j < R; j++) {
        if (emit & (1u << j)) {
            sk[at] = x[j];
            sv[at] = xv[j];
            at++;
        }
    }
}

// PERSISTENT: the grid is what the chip holds at once (launch_wave asks the runtime), a workgroup draws one ticket after the
// other -- thread 0 requests the next one while the current one is worked on -- and the keys and values of the NEXT ticket's
// segments are requested (into registers) before the current ticket's records are stored: the ticket, the segment bounds and
// the entries themselves arrive behind work instead of in front of it.  (One workgroup per ticket: 22 us per ticket of which
// the waves issued instructions for 12 % -- a chain of round trips: ticket, bounds, entries, look-back, stores.)
template <int NI>
__device__ __forceinline__ void wave_fetch(const Args &a, i64 beg, int n, int lane, u32 (&k)[NI], double (&v)[NI]) {
    // (unconditional: an empty segment reads entry 0 and ignores it -- a conditional request would keep the previous
    // ticket's registers alive through the whole iteration)
    // (uniform base + 32-bit byte offset per lane: one v_min and two shifts per entry instead of 64-bit address arithmetic)
    // (the segment's start comes out of the loop's carried state, which the register allocator may have moved to vector
    // registers: back to scalar ones here, so that the loads use a scalar base)
    const i64 lbeg = esp_uniform_i64(n > 0 ? beg : 0);
    n = esp_uniform_i32(n);
    const char *kp = reinterpret_cast<const char *>(reinterpret_cast<const u32 *>(a.keys_in) + lbeg);
    const char *vp = reinterpret_cast<const char *>(a.vals_in + lbeg);
    const u32 nlast = n > 0 ? (u32)(n - 1) : 0u;
    u32 at[NI];
#pragma unroll
    for (int i = 0; i < NI; i++) at[i] = min((u32)lane + (u32)(i * ESP_WAVE), nlast);
#pragma unroll
    for (int i = 0; i < NI; i++) k[i] = *reinterpret_cast<const u32 *>(kp + (size_t)(at[i] << 2));
#pragma unroll
    for (int i = 0; i < NI; i++) v[i] = *reinterpret_cast<const double *>(vp + (size_t)(at[i] << 3));
}

// WV_WAVES segments (= waves) per workgroup and ticket: the look-back's granules, polls and ticket draws are per workgroup
template <int NI, int WV_WAVES>
__global__ __launch_bounds__(WV_WAVES * ESP_WAVE, NI <= 12 ? 4 : 3) void wave_k(Args a) {
    static_assert(NI * ESP_WAVE <= (1 << WV_IDX_BITS), "slot index bits");
    constexpr int CAPW = NI * ESP_WAVE;
    __shared__ double sval[WV_WAVES][CAPW];
    __shared__ u32 skey[WV_WAVES][CAPW + WV_MAXRUN];
    __shared__ u32 ccnt[WV_WAVES][ESP_WAVE + 1];
    __shared__ u32 s_tick[2], s_tot[WV_WAVES];
    __shared__ u64 s_dst;

    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int nwg = (a.S + WV_WAVES - 1) / WV_WAVES;
#ifdef ESP_WAVE_STATIC  // (experiment: no ticket atomics -- round-robin over the resident workgroups)
    u32 st_next = blockIdx.x + 2 * gridDim.x;
#define ESP_WAVE_TICKET() (st_next += gridDim.x, st_next - gridDim.x)
    if (t == 0) {
        s_tick[0] = blockIdx.x;
        s_tick[1] = blockIdx.x + gridDim.x;
    }
#else
#define ESP_WAVE_TICKET() atomicAdd(a.ticket, 1u)
    if (t == 0) {
        s_tick[0] = atomicAdd(a.ticket, 1u);
        s_tick[1] = atomicAdd(a.ticket, 1u);
    }
#endif
    u32 *cc = ccnt[w];
    u32 *sk = skey[w];
    double *sv = sval[w];
    __syncthreads();
    // pipeline over the tickets a workgroup draws: iteration i works on ticket(i) -- its entries were requested after the first
    // barrier of iteration i - 1 --, knows ticket(i + 1) and its segment bounds, and thread 0 holds ticket(i + 2)
    int wg = esp_uniform_i32((int)s_tick[0]);
    if (wg >= nwg) return;
    int wgn = esp_uniform_i32((int)s_tick[1]);
    u32 tn = 0;
    if (t == 0) tn = ESP_WAVE_TICKET();
    const bool raw = a.kind32 == (u32)ESP_RAWUPDATE;
    const u32 rowmask32 = (1u << a.rb) - 1u;  // (cl_bits + rb <= 32)
    auto bounds = [&](int g, i64 &b, i64 &e) {
        b = e = 0;
        if (g < nwg && g * WV_WAVES + w < a.S) {
            b = esp_uniform_i64(a.seg_start[g * WV_WAVES + w]);
            e = esp_uniform_i64(a.seg_start[g * WV_WAVES + w + 1]);
        }
    };
    i64 beg, seg_end, nbeg, nend;
    bounds(wg, beg, seg_end);
    u32 k[NI];
    double v[NI];
    wave_fetch<NI>(a, beg, min((int)(seg_end - beg), CAPW), lane, k, v);
    bounds(wgn, nbeg, nend);
    int par = 0;
    for (;;) {
        const int s = wg * WV_WAVES + w;
        if (a.total >= 0 && s == a.S - 1 && seg_end != a.total && lane == 0) atomicOr(a.err, 2u);  // (an entry behind the last column)
        int n = (int)(seg_end - beg);
        if (n > CAPW) {  // (the host checked the longest segment)
            if (lane == 0) atomicOr(a.err, 8u | 32u);
            n = 0;
        }
        u32 rmin = 0, colbase = 0, tot = 0;
        cc[lane] = 0;
        if (lane == 0) cc[ESP_WAVE] = 0;
        if (n > 0) {
#pragma unroll
            for (int i = 0; i < NI; i++) sv[lane + i * ESP_WAVE] = v[i];
            // ---- counting sort by column (a slot past the end counts into the spare counter)
            u32 slot[NI];
            u32 rmax = 0;
            rmin = ~0u;
#pragma unroll
            for (int i = 0; i < NI; i++) {
                const bool valid = lane + i * ESP_WAVE < n;
                const u32 col = valid ? k[i] >> a.rb : (u32)ESP_WAVE;  // (the partition masked the keys: col < 2^cl_bits)
                slot[i] = atomicAdd(&cc[col], 1u);
                const u32 row = k[i] & rowmask32;  // (a slot past the end holds a copy of the last entry: its row changes nothing)
                rmin = min(rmin, row);
                rmax = max(rmax, row);
            }
            wave_lds_sync();
            const u32 cnt = cc[lane];
            const u32 incl = esp_wave_scan_add(cnt);
            const u32 maxrun = esp_wave_max(cnt);
            rmin = ~esp_wave_max(~rmin);
            rmax = esp_wave_max(rmax);
            const int rs = (int)(incl - cnt);
            cc[lane] = (u32)rs;
            if (maxrun > (u32)WV_MAXRUN && lane == 0) atomicMax(a.maxrun_seen, maxrun);
            const bool rows_ok = rmax - rmin < (1u << WV_ROW_BITS) - 2u;
            if (maxrun > (u32)WV_MAXRUN || !rows_ok) {
                if (lane == 0) atomicOr(a.err, rows_ok ? (8u | 32u) : (8u | 16u));
            } else {
                wave_lds_sync();
#pragma unroll
                for (int i = 0; i < NI; i++) {
                    const int p = lane + i * ESP_WAVE;
                    if (p < n) {
                        const u32 col = k[i] >> a.rb;
                        sk[cc[col] + slot[i]] = (((k[i] & rowmask32) - rmin) << WV_IDX_BITS) | (u32)p;
                    }
                }
                wave_lds_sync();
                // sort + fold in the lane; the records go to their dense positions of the wave's own arrays (every key and value
                // of the wave has been read by then)
                if (maxrun <= 12u) {
                    u32 x[12], emit;
                    double xv[12];
                    wave_run<12>(sk, sv, rs, (int)cnt, raw, x, xv, emit);
                    const u32 ec = (u32)__popc(emit);
                    const u32 einc = esp_wave_scan_add(ec);
                    colbase = einc - ec;
                    tot = (u32)__builtin_amdgcn_readlane((int)einc, 63);
                    wave_lds_sync();
                    wave_put<12>(sk, sv, colbase, x, xv, emit);
                } else {
                    u32 x[16], emit;
                    double xv[16];
                    wave_run<16>(sk, sv, rs, (int)cnt, raw, x, xv, emit);
                    const u32 ec = (u32)__popc(emit);
                    const u32 einc = esp_wave_scan_add(ec);
                    colbase = einc - ec;
                    tot = (u32)__builtin_amdgcn_readlane((int)einc, 63);
                    wave_lds_sync();
                    wave_put<16>(sk, sv, colbase, x, xv, emit);
                }
            }
        }
        if (lane == 0) s_tot[w] = tot;
        if (t == 0) s_tick[par] = tn;
        __syncthreads();
        // ---- the next ticket's entries are requested; the workgroup's total is published and the last wave resolves the
        // look-back chain
        wave_fetch<NI>(a, nbeg, min((int)(nend - nbeg), CAPW), lane, k, v);
        // (the waves' totals scanned by every wave for itself: lane i holds wave i's)
        const u32 ti = lane < WV_WAVES ? s_tot[lane] : 0u;
        const u32 tinc = esp_wave_scan_add(ti);
        const u32 wg_total = (u32)__builtin_amdgcn_readlane((int)tinc, 63);
        const u32 before = (u32)__builtin_amdgcn_readlane((int)(tinc - ti), w);
        if (w == WV_WAVES - 1) {
            Args la = a;  // (the look-back runs over the workgroups' granules)
            la.S = nwg;
            LbState lbs;
            lb_publish(la, lbs, wg, wg_total, lane);
#ifdef ESP_WAVE_LB_DELAY
            if (!lbs.finished) __builtin_amdgcn_s_sleep(ESP_WAVE_LB_DELAY);  // (the neighbours in front publish about now: a round of polls that finds nothing costs more than waiting)
#endif
            const u64 excl = lb_complete(la, lbs, wg, wg_total, lane);
            if (lane == 0) s_dst = excl;
        }
        __syncthreads();
        const u64 dst = esp_uniform_u64(s_dst) + (u64)before;
        // ---- the ticket after the next: its bounds are requested, thread 0 draws the one after that; then this ticket's
        // records leave
        const int wgnn = esp_uniform_i32((int)s_tick[par]);
        i64 nnbeg, nnend;
        bounds(wgnn, nnbeg, nnend);
        if (t == 0 && wgnn < nwg) tn = ESP_WAVE_TICKET();
        // coalesced stores; the lane's column starts at dst + colbase
        {
            const u64 dstu = esp_uniform_u64(dst);
            char *orow = reinterpret_cast<char *>(a.out_row + dstu), *oval = reinterpret_cast<char *>(a.out_val + dstu);
            for (u32 p = (u32)lane; p < tot; p += ESP_WAVE) {
                *reinterpret_cast<i64 *>(orow + (size_t)(p << 3)) = (i64)(sk[p] + rmin) + 1;
                *reinterpret_cast<double *>(oval + (size_t)(p << 3)) = sv[p];
            }
        }
        if (s < a.S) {
            const i64 c = (i64)((((u64)s << a.rem_bits) + a.base) >> a.rb) + lane;
            if (lane < (1 << a.cl_bits) && c < a.col_end) a.colptr_out[c] = (i64)(dst + (u64)colbase) + 1;
            if (s == a.S - 1 && lane == 0) a.colptr_out[a.col_end] = (i64)(dst + (u64)tot) + 1;
        }
        if (wgn >= nwg) break;
        wave_lds_sync();  // (the records have been read: the next ticket's values take their place)
        wg = wgn;
        wgn = wgnn;
        beg = nbeg;
        seg_end = nend;
        nbeg = nnbeg;
        nend = nnend;
        par ^= 1;
    }
}

}  // namespace esplocal
