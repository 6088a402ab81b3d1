// stress_handles.hip -- TEST INFRASTRUCTURE: distinct handles of libesparse_hip.so driven side by side.
//
// The reference's multi-threaded assembly gives every task a buffer of its own and lets the tasks run concurrently
// (src/matrix/genericmtextendablesparsematrixcsc.jl:87-99, test/femtools.jl:88-107 `@tasks for part`): the C ABI's
// promise "distinct handles are independent and may be driven from different host threads" is that contract.  This
// program checks it: every handle gets a fixed workload; a SERIAL phase runs each workload alone and records a hash of
// the resulting CSC (three times: it must be stable); the CONCURRENT phase runs the same workloads from T host threads
// (each thread its own handles, fill + flush + read-back per iteration) and every result must hash the same.
//
//   stress_handles [--handles H] [--threads T] [--iters N] [--work mix|elem10|fem4|trip|fd|sum|parts] [--mode threads|lockstep|spawn|interleave|serial]
//                  [--hog none|lds|fill] [--fresh 0|1] [--scale S] [--seed X] [--kind K] [--quiet]
//                  [--writeref FILE] | [--cold 1 --ref FILE]
//   --mode lockstep: every thread fills its handles, a barrier, ALL flush at the same moment, a barrier, read-back
//   --mode spawn: the main thread fills, one NEW thread per handle flushes (what a threaded esp_flush_sum does), main reads
//   --mode interleave: ONE host thread issues the calls of all handles round-robin (appends of all, then flushes of all)
//   --hog lds:  a side thread keeps launching a kernel that scribbles over all of a CU's LDS (what a kernel that reads
//               LDS it did not write would meet);  --hog fill: a side thread keeps the chip busy with a bandwidth kernel
//   --cold 1:   no serial phase: the concurrent phase is the process's first use of every kernel (code objects load lazily on the
//               first launch); references from --ref FILE, written by an earlier run with --writeref FILE
//   --fresh 1:  every iteration destroys the handle and creates a new one (allocation churn beside running pipelines)
// Exit code 0 and "stress_handles: ok" when every result matched; 1 otherwise (the mismatches / errors are listed).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../include/esparse_hip.h"

typedef int64_t i64;
typedef uint64_t u64;

static u64 mix64(u64 z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
struct Rng {
    u64 s;
    explicit Rng(u64 seed) : s(seed) {}
    u64 next() { return mix64(s += 0x9E3779B97F4A7C15ull); }
    i64 below(i64 n) { return (i64)(next() % (u64)n); }
    double unit() { return (double)(next() >> 11) * 0x1.0p-53; }
    double normalish() { return unit() + unit() + unit() - 1.5; }
};

enum WorkKind { W_ELEM10 = 0, W_FEM4 = 1, W_TRIP = 2, W_FD = 3, W_SUM = 4 };

struct Work {
    int kind = 0;
    i64 n = 0;
    // elements
    int nloc = 0;
    i64 ncells = 0;
    std::vector<i64> cn;
    std::vector<double> em, dg;
    // triplets
    std::vector<i64> rows, cols;
    std::vector<double> vals;
    // fdrand
    i64 nx = 0;
    u64 seed = 0;
    // sum: the cells dealt to p buffers
    int p = 0;
    // result of the serial phase
    u64 ref_hash = 0;
    i64 ref_nnz = -1;
    std::vector<i64> ref_colptr, ref_rowval;
    std::vector<double> ref_nzval;
};

static void make_elements(Work &w, Rng &r, int nloc, i64 n, i64 nc, int span) {
    w.nloc = nloc;
    w.n = n;
    w.ncells = nc;
    // a permuted numbering: nothing downstream can lean on grid arithmetic
    std::vector<i64> perm((size_t)n);
    for (i64 i = 0; i < n; i++) perm[(size_t)i] = i + 1;
    for (i64 i = n - 1; i > 0; i--) std::swap(perm[(size_t)i], perm[(size_t)r.below(i + 1)]);
    w.cn.resize((size_t)(nloc * nc));
    w.em.resize((size_t)(nloc * nloc * nc));
    w.dg.resize((size_t)(nloc * nc));
    std::vector<int> loc((size_t)span);
    for (i64 c = 0; c < nc; c++) {
        const i64 start = r.below(n - span + 1);
        for (int i = 0; i < span; i++) loc[(size_t)i] = i;
        for (int i = 0; i < nloc; i++) {  // nloc distinct offsets of the window
            const int j = i + (int)r.below(span - i);
            std::swap(loc[(size_t)i], loc[(size_t)j]);
            w.cn[(size_t)(c * nloc + i)] = perm[(size_t)(start + loc[(size_t)i])];
        }
    }
    for (auto &x : w.em) x = r.normalish();
    for (auto &x : w.dg) x = r.normalish();
}

static int g_kind = ESP_UPDATE;
static void make_work(Work &w, int kind, u64 seed, double scale) {
    Rng r(seed * 7919 + 13);
    w.kind = kind;
    w.seed = seed;
    switch (kind) {
    case W_ELEM10: make_elements(w, r, 10, (i64)(2000000 * scale), (i64)(60000 * scale), 10); break;
    case W_SUM: make_elements(w, r, 10, (i64)(2000000 * scale), (i64)(60000 * scale), 10); w.p = 3; break;
    case W_FEM4: make_elements(w, r, 4, (i64)(1500000 * scale), (i64)(400000 * scale), 8); break;
    case W_TRIP: {
        w.n = (i64)(300000 * scale);
        const i64 E = (i64)(4000000 * scale);
        w.rows.resize((size_t)E), w.cols.resize((size_t)E), w.vals.resize((size_t)E);
        for (i64 e = 0; e < E; e++) {
            const i64 c = r.below(w.n);
            i64 rr = c + r.below(41) - 20;
            rr = rr < 0 ? 0 : (rr >= w.n ? w.n - 1 : rr);
            w.rows[(size_t)e] = rr + 1, w.cols[(size_t)e] = c + 1, w.vals[(size_t)e] = r.normalish();
        }
        break;
    }
    default:
        w.nx = (i64)(40 + 24 * scale);
        w.n = w.nx * w.nx * w.nx;
        break;
    }
}

static u64 hash_bytes(u64 h, const void *p, size_t bytes) {
    const u64 *q = (const u64 *)p;
    for (size_t i = 0; i < bytes / 8; i++) h = mix64(h ^ q[i]) + 0x9E3779B97F4A7C15ull * (i + 1);
    return h;
}

struct Slot {
    int index = 0;
    Work *w = nullptr;
    esp_handle *h = nullptr;
    esp_handle *xs[4] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<i64> colptr, rowval;
    std::vector<double> nzval;
    std::string err;
};

static int g_force = 0, g_force_of[8] = {-1, -1, -1, -1, -1, -1, -1, -1};
static bool slot_open(Slot &s) {
    if (esp_create(s.w->n, s.w->n, 0, 0, &s.h) != ESP_OK) {
        s.err = std::string("esp_create: ") + esp_last_error(nullptr);
        return false;
    }
    if (g_force) (void)esp_debug_force_path(s.h, g_force);
    if (s.index < 8 && g_force_of[s.index] >= 0) (void)esp_debug_force_path(s.h, g_force_of[s.index]);
    for (int k = 0; k < s.w->p; k++)
        if (esp_create(s.w->n, s.w->n, 0, 0, &s.xs[k]) != ESP_OK) {
            s.err = std::string("esp_create: ") + esp_last_error(nullptr);
            return false;
        }
    return true;
}
static void slot_close(Slot &s) {
    for (int k = 0; k < 4; k++) {
        if (s.xs[k]) esp_destroy(s.xs[k]);
        s.xs[k] = nullptr;
    }
    if (s.h) esp_destroy(s.h);
    s.h = nullptr;
}

#define SCK(call)                                                                           \
    do {                                                                                    \
        const int32_t rc_ = (call);                                                         \
        if (rc_ != ESP_OK) {                                                                \
            char b_[700];                                                                   \
            snprintf(b_, sizeof b_, "%s -> %d (%s)", #call, (int)rc_, esp_last_error(s.h)); \
            s.err = b_;                                                                     \
            return false;                                                                   \
        }                                                                                   \
    } while (0)

static bool slot_fill(Slot &s) {
    Work &w = *s.w;
    SCK(esp_reset(s.h));
    switch (w.kind) {
    case W_ELEM10:
    case W_FEM4:
        SCK(esp_append_elements_host(s.h, w.nloc, w.ncells, w.cn.data(), w.em.data(), w.dg.data(), g_kind, ESP_OP_ADD));
        break;
    case W_SUM: {
        const i64 cuts[4] = {0, w.ncells * 17 / 60, w.ncells * 41 / 60, w.ncells};
        for (int k = 0; k < w.p; k++) {
            const i64 a = cuts[k], b = cuts[k + 1];
            const int32_t rc = esp_append_elements_host(s.xs[k], w.nloc, b - a, w.cn.data() + a * w.nloc, w.em.data() + a * w.nloc * w.nloc,
                                                        w.dg.data() + a * w.nloc, g_kind, ESP_OP_ADD);
            if (rc != ESP_OK) {
                s.err = std::string("esp_append_elements_host(buffer): ") + esp_last_error(s.xs[k]);
                return false;
            }
        }
        break;
    }
    case W_TRIP:
        SCK(esp_append_host(s.h, w.rows.data(), w.cols.data(), w.vals.data(), nullptr, ESP_UPDATE, ESP_OP_ADD, (i64)w.rows.size()));
        break;
    default: SCK(esp_generate_fdrand(s.h, w.nx, w.nx, w.nx, w.seed, 1, ESP_UPDATE)); break;
    }
    return true;
}
static bool slot_flush_only(Slot &s) {
    Work &w = *s.w;
    int64_t z = 0;
    int32_t changed = 0;
    if (w.kind == W_SUM)
        SCK(esp_flush_sum(s.h, s.xs, w.p, &z, &changed));
    else
        SCK(esp_flush(s.h, ESP_FLUSH_ROUTED, &z, &changed));
    return true;
}
static bool slot_read(Slot &s, u64 *hash_out, i64 *nnz_out) {
    Work &w = *s.w;
    int64_t z = 0;
    SCK(esp_nnz(s.h, &z));
    s.colptr.resize((size_t)(w.n + 1));
    s.rowval.resize((size_t)z);
    s.nzval.resize((size_t)z);
    SCK(esp_get_csc(s.h, s.colptr.data(), s.rowval.data(), s.nzval.data()));
    u64 hh = 0x1234567ull;
    hh = hash_bytes(hh, s.colptr.data(), sizeof(i64) * s.colptr.size());
    hh = hash_bytes(hh, s.rowval.data(), sizeof(i64) * s.rowval.size());
    hh = hash_bytes(hh, s.nzval.data(), sizeof(double) * s.nzval.size());
    *hash_out = hh;
    *nnz_out = z;
    return true;
}
static bool slot_flush(Slot &s, u64 *hash_out, i64 *nnz_out) { return slot_flush_only(s) && slot_read(s, hash_out, nnz_out); }

// where a result departs from the handle's result alone: columns whose entries differ, the first few in detail
static std::string diff_report(const Slot &s) {
    const Work &w = *s.w;
    if (w.ref_colptr.empty()) return "";
    char b[256];
    std::string out;
    i64 ncols_diff = 0, first = -1, last = -1;
    int shown = 0;
    for (i64 c = 0; c < w.n; c++) {
        const i64 a0 = w.ref_colptr[(size_t)c] - 1, a1 = w.ref_colptr[(size_t)c + 1] - 1;
        const i64 b0 = s.colptr[(size_t)c] - 1, b1 = s.colptr[(size_t)c + 1] - 1;
        bool same = (a1 - a0) == (b1 - b0);
        if (same && b1 <= (i64)s.rowval.size() && b0 >= 0)
            for (i64 k = 0; k < a1 - a0 && same; k++)
                same = w.ref_rowval[(size_t)(a0 + k)] == s.rowval[(size_t)(b0 + k)] &&
                       memcmp(&w.ref_nzval[(size_t)(a0 + k)], &s.nzval[(size_t)(b0 + k)], 8) == 0;
        if (same) continue;
        ncols_diff++;
        if (first < 0) first = c;
        last = c;
        if (shown < 6) {
            shown++;
            snprintf(b, sizeof b, " [col %lld: alone %lld entries at %lld, now %lld at %lld]", (long long)c, (long long)(a1 - a0), (long long)a0, (long long)(b1 - b0), (long long)b0);
            out += b;
            if (b0 >= 0 && b1 <= (i64)s.rowval.size() && b1 - b0 < 400 && a1 - a0 < 400 && shown <= 2) {
                out += " alone rows:";
                for (i64 k = a0; k < a1; k++) out += " " + std::to_string(w.ref_rowval[(size_t)k]);
                out += " now rows:";
                for (i64 k = b0; k < b1; k++) out += " " + std::to_string(s.rowval[(size_t)k]);
            }
        }
    }
    snprintf(b, sizeof b, " => %lld columns differ, first %lld last %lld of %lld", (long long)ncols_diff, (long long)first, (long long)last, (long long)w.n);
    return out + b;
}

struct SpinBarrier {
    std::atomic<int> count{0}, gen{0};
    int n = 1;
    void wait() {
        const int g = gen.load();
        if (count.fetch_add(1) + 1 == n) {
            count.store(0);
            gen.fetch_add(1);
        } else {
            while (gen.load() == g) std::this_thread::yield();
        }
    }
};

// ---- hogs -----------------------------------------------------------------------------------------------------------
__global__ void lds_scribble_k(u64 salt, u64 *sink) {
    extern __shared__ u64 lds[];
    const int words = 65536 / 8;
    for (int i = threadIdx.x; i < words; i += blockDim.x) lds[i] = 0xFFF0DEADBEEF0000ull ^ (salt + (u64)i * 0x9E3779B97F4A7C15ull);
    __syncthreads();
    if (sink && lds[(threadIdx.x * 7) % words] == 1) sink[0] = 1;
}
__global__ void fill_hog_k(u64 *p, size_t words, u64 v) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (size_t)gridDim.x * blockDim.x) p[i] = v + i;
}

int main(int argc, char **argv) {
    int H = 4, T = 4, iters = 50, fresh = 0, quiet = 0;
    double scale = 1.0;
    u64 seed = 503;
    std::string work = "mix", mode = "threads", hog = "none", ref_in, ref_out;
    int cold = 0, force = 0;
    std::vector<int> partmap;
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto val = [&]() -> const char * { return i + 1 < argc ? argv[++i] : ""; };
        if (a == "--handles") H = atoi(val());
        else if (a == "--threads") T = atoi(val());
        else if (a == "--iters") iters = atoi(val());
        else if (a == "--work") work = val();
        else if (a == "--mode") mode = val();
        else if (a == "--hog") hog = val();
        else if (a == "--fresh") fresh = atoi(val());
        else if (a == "--scale") scale = atof(val());
        else if (a == "--seed") seed = strtoull(val(), nullptr, 0);
        else if (a == "--quiet") quiet = 1;
        else if (a == "--kind") g_kind = atoi(val());
        else if (a == "--cold") cold = atoi(val());
        else if (a == "--force") force = atoi(val());
        else if (a.rfind("--force", 0) == 0 && a.size() == 8 && a[7] >= '0' && a[7] <= '7') g_force_of[a[7] - '0'] = atoi(val());
        else if (a == "--parts") {
            for (const char *q = val(); *q; q++)
                if (*q >= '0' && *q <= '9') partmap.push_back(*q - '0');
        }
        else if (a == "--ref") ref_in = val();
        else if (a == "--writeref") ref_out = val();
        else {
            fprintf(stderr, "stress_handles: unknown argument %s\n", a.c_str());
            return 2;
        }
    }
    if (T > H) T = H;
    g_force = force;
    if (hipSetDevice(0) != hipSuccess) {
        fprintf(stderr, "stress_handles: no HIP device\n");
        return 2;
    }
    std::vector<Work> works((size_t)H);
    for (int i = 0; i < H; i++) {
        int kind = W_FD;
        if (work == "mix") kind = i % 5;
        else if (work == "elem10") kind = W_ELEM10;
        else if (work == "fem4") kind = W_FEM4;
        else if (work == "trip") kind = W_TRIP;
        else if (work == "sum") kind = W_SUM;
        if (work == "parts") {
            // the fuzz's failing case: ONE mesh of 60000 ten-node cells, dealt unevenly to the handles (the middle part is small: its
            // flush takes packed keys and the colptr scan, the others 4-byte keys and the direct colptr)
            make_work(works[(size_t)i], W_ELEM10, seed, scale);
            Work &w = works[(size_t)i];
            const i64 cut[4] = {0, w.ncells * 24513 / 60000, w.ncells * 31150 / 60000, w.ncells};
            const int part = partmap.empty() ? i % 3 : partmap[(size_t)i % partmap.size()] % 3;
            const i64 a = cut[part], b = cut[part + 1];
            w.cn = std::vector<i64>(w.cn.begin() + a * w.nloc, w.cn.begin() + b * w.nloc);
            w.em = std::vector<double>(w.em.begin() + a * w.nloc * w.nloc, w.em.begin() + b * w.nloc * w.nloc);
            w.dg = std::vector<double>(w.dg.begin() + a * w.nloc, w.dg.begin() + b * w.nloc);
            w.ncells = b - a;
            continue;
        }
        make_work(works[(size_t)i], kind, seed + (u64)i, scale);
    }
    std::vector<Slot> slots((size_t)H);
    for (int i = 0; i < H; i++) slots[(size_t)i].w = &works[(size_t)i], slots[(size_t)i].index = i;
    int bad = 0;
    // ---- serial phase: the reference results (--cold 1: none -- the concurrent phase is the process's FIRST use of every kernel; the
    // references come from the file an earlier run wrote with --writeref)
    if (cold) {
        FILE *f = fopen(ref_in.c_str(), "r");
        if (!f) {
            fprintf(stderr, "stress_handles: --cold needs --ref FILE (made with --writeref)\n");
            return 2;
        }
        for (int i = 0; i < H; i++) {
            unsigned long long hh = 0;
            long long z = 0;
            if (fscanf(f, "%llx %lld", &hh, &z) != 2) return 2;
            works[(size_t)i].ref_hash = hh, works[(size_t)i].ref_nnz = z;
        }
        fclose(f);
    }
    for (int i = 0; i < H; i++) {
        Slot &s = slots[(size_t)i];
        if (!slot_open(s)) {
            fprintf(stderr, "stress_handles: %s\n", s.err.c_str());
            return 1;
        }
        for (int rep = 0; rep < (cold ? 0 : 3); rep++) {
            u64 hh = 0;
            i64 z = 0;
            if (!slot_fill(s) || !slot_flush(s, &hh, &z)) {
                fprintf(stderr, "stress_handles: serial phase, handle %d: %s\n", i, s.err.c_str());
                return 1;
            }
            if (rep == 0) s.w->ref_hash = hh, s.w->ref_nnz = z, s.w->ref_colptr = s.colptr, s.w->ref_rowval = s.rowval, s.w->ref_nzval = s.nzval;
            else if (hh != s.w->ref_hash) {
                fprintf(stderr, "stress_handles: handle %d (work %d) is not stable ALONE: rep %d hash %016llx, first %016llx\n", i, s.w->kind, rep,
                        (unsigned long long)hh, (unsigned long long)s.w->ref_hash);
                bad++;
            }
        }
        if (!quiet) printf("serial: handle %d work %d n %lld nnz %lld hash %016llx\n", i, s.w->kind, (long long)s.w->n, (long long)s.w->ref_nnz, (unsigned long long)s.w->ref_hash);
    }
    if (!ref_out.empty()) {
        FILE *f = fopen(ref_out.c_str(), "w");
        for (int i = 0; f && i < H; i++) fprintf(f, "%016llx %lld\n", (unsigned long long)works[(size_t)i].ref_hash, (long long)works[(size_t)i].ref_nnz);
        if (f) fclose(f);
    }
    // ---- hog
    std::atomic<bool> stop{false};
    std::thread hog_thread;
    if (hog != "none") {
        hog_thread = std::thread([&] {
            (void)hipSetDevice(0);
            hipStream_t st;
            (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
            u64 *buf = nullptr;
            const size_t words = (size_t)1 << 27;
            if (hog == "fill") (void)hipMalloc((void **)&buf, words * 8);
            (void)hipFuncSetAttribute((const void *)lds_scribble_k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
            u64 salt = 1;
            while (!stop.load()) {
                for (int k = 0; k < 8; k++) {
                    if (hog == "lds") hipLaunchKernelGGL(lds_scribble_k, dim3(1024), dim3(256), 65536, st, salt++, (u64 *)nullptr);
                    else hipLaunchKernelGGL(fill_hog_k, dim3(2048), dim3(256), 0, st, buf, words, salt++);
                }
                (void)hipStreamSynchronize(st);
            }
            if (buf) (void)hipFree(buf);
            (void)hipStreamDestroy(st);
        });
    }
    // ---- concurrent phase
    std::atomic<int> mismatches{0}, errors{0};
    std::vector<std::string> log;
    std::mutex log_m;
    auto note = [&](const std::string &m) {
        std::lock_guard<std::mutex> lk(log_m);
        if (log.size() < 40) log.push_back(m);
    };
    auto step = [&](int i, int it) {
        Slot &s = slots[(size_t)i];
        if (fresh) {
            slot_close(s);
            if (!slot_open(s)) {
                errors++;
                note("iter " + std::to_string(it) + " handle " + std::to_string(i) + ": " + s.err);
                return;
            }
        }
        u64 hh = 0;
        i64 z = 0;
        if (!slot_fill(s) || !slot_flush(s, &hh, &z)) {
            errors++;
            note("iter " + std::to_string(it) + " handle " + std::to_string(i) + " work " + std::to_string(s.w->kind) + ": " + s.err);
            return;
        }
        if (hh != s.w->ref_hash) {
            mismatches++;
            char b[200];
            snprintf(b, sizeof b, "iter %d handle %d work %d: hash %016llx nnz %lld, alone %016llx nnz %lld", it, i, s.w->kind, (unsigned long long)hh,
                     (long long)z, (unsigned long long)s.w->ref_hash, (long long)s.w->ref_nnz);
            note(b);
        }
    };
    if (mode == "threads") {
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t] {
                (void)hipSetDevice(0);
                for (int it = 0; it < iters; it++)
                    for (int i = t; i < H; i += T) step(i, it);
            });
        for (auto &x : th) x.join();
    } else if (mode == "lockstep") {
        // every thread fills its handles, ALL flush at the same moment (a barrier in front), then the results are read
        SpinBarrier bar;
        bar.n = T;
        std::vector<std::thread> th;
        for (int t = 0; t < T; t++)
            th.emplace_back([&, t] {
                (void)hipSetDevice(0);
                for (int it = 0; it < iters; it++) {
                    std::vector<char> okf((size_t)H, 1);
                    for (int i = t; i < H; i += T)
                        if (!slot_fill(slots[(size_t)i])) okf[(size_t)i] = 0, errors++, note("fill " + std::to_string(i) + ": " + slots[(size_t)i].err);
                    bar.wait();
                    for (int i = t; i < H; i += T)
                        if (okf[(size_t)i] && !slot_flush_only(slots[(size_t)i]))
                            okf[(size_t)i] = 0, errors++, note("iter " + std::to_string(it) + " flush " + std::to_string(i) + ": " + slots[(size_t)i].err);
                    bar.wait();
                    for (int i = t; i < H; i += T) {
                        if (!okf[(size_t)i]) continue;
                        Slot &s = slots[(size_t)i];
                        u64 hh = 0;
                        i64 z = 0;
                        if (!slot_read(s, &hh, &z)) errors++, note("read " + std::to_string(i) + ": " + s.err);
                        else if (hh != s.w->ref_hash) {
                            mismatches++;
                            note("iter " + std::to_string(it) + " handle " + std::to_string(i) + " work " + std::to_string(s.w->kind) + ": hash differs, nnz " +
                                 std::to_string(z) + " alone " + std::to_string(s.w->ref_nnz));
                        }
                    }
                }
            });
        for (auto &x : th) x.join();
    } else if (mode == "spawn") {
        // the main thread fills every handle; one NEW host thread per handle runs its flush; the main thread reads the results
        for (int it = 0; it < iters; it++) {
            std::vector<char> okf((size_t)H, 1);
            for (int i = 0; i < H; i++)
                if (!slot_fill(slots[(size_t)i])) okf[(size_t)i] = 0, errors++, note("fill " + std::to_string(i) + ": " + slots[(size_t)i].err);
            std::vector<std::thread> th;
            for (int i = 0; i < H; i++)
                if (okf[(size_t)i])
                    th.emplace_back([&, i, it] {
                        (void)hipSetDevice(0);
                        if (!slot_flush_only(slots[(size_t)i]))
                            okf[(size_t)i] = 0, errors++, note("iter " + std::to_string(it) + " flush " + std::to_string(i) + ": " + slots[(size_t)i].err);
                    });
            for (auto &x : th) x.join();
            for (int i = 0; i < H; i++) {
                if (!okf[(size_t)i]) continue;
                Slot &s = slots[(size_t)i];
                u64 hh = 0;
                i64 z = 0;
                if (!slot_read(s, &hh, &z)) errors++, note("read " + std::to_string(i) + ": " + s.err);
                else if (hh != s.w->ref_hash) {
                    mismatches++;
                    note("iter " + std::to_string(it) + " handle " + std::to_string(i) + " work " + std::to_string(s.w->kind) + ": hash differs, nnz " +
                         std::to_string(z) + " alone " + std::to_string(s.w->ref_nnz) + diff_report(s));
                }
            }
        }
    } else if (mode == "interleave") {
        for (int it = 0; it < iters; it++) {
            std::vector<char> okf((size_t)H, 1);
            for (int i = 0; i < H; i++) {
                Slot &s = slots[(size_t)i];
                if (!slot_fill(s)) {
                    okf[(size_t)i] = 0, errors++;
                    note("iter " + std::to_string(it) + " handle " + std::to_string(i) + ": " + s.err);
                }
            }
            for (int i = 0; i < H; i++) {
                Slot &s = slots[(size_t)i];
                if (!okf[(size_t)i]) continue;
                u64 hh = 0;
                i64 z = 0;
                if (!slot_flush(s, &hh, &z)) {
                    errors++;
                    note("iter " + std::to_string(it) + " handle " + std::to_string(i) + ": " + s.err);
                } else if (hh != s.w->ref_hash) {
                    mismatches++;
                    note("iter " + std::to_string(it) + " handle " + std::to_string(i) + " work " + std::to_string(s.w->kind) + ": hash differs");
                }
            }
        }
    } else {
        for (int it = 0; it < iters; it++)
            for (int i = 0; i < H; i++) step(i, it);
    }
    stop.store(true);
    if (hog_thread.joinable()) hog_thread.join();
    for (auto &s : slots) slot_close(s);
    for (auto &m : log) fprintf(stderr, "stress_handles: %s\n", m.c_str());
    printf("stress_handles: mode %s work %s hog %s fresh %d handles %d threads %d iters %d -> mismatches %d errors %d unstable-alone %d\n", mode.c_str(),
           work.c_str(), hog.c_str(), fresh, H, T, iters, mismatches.load(), errors.load(), bad);
    if (mismatches.load() || errors.load() || bad) return 1;
    printf("stress_handles: ok\n");
    return 0;
}
