// wae_beyn_moments_mgpu -- the Beyn quadrature of ONE host process spread over several GPUs of a node (SURVEY.md 8b:
// `beyn_moments(..., ngpu)`, 8e).  The reference has nothing to port here (serial Julia); the structure follows what the
// hot path offers: quadrature points are independent, so every GPU holds a replica of the family (one handle per device,
// created and set up by the caller), takes a share of the work on its own stream under its own host thread, and the only
// data exchanged over xGMI are (i) the snapshot bases of the projected-guess scheme -- one RCCL all-gather -- and (ii) the
// partial moment tensors -- one RCCL sum-reduce to device 0.
//
// RCCL is bound at first use with dlopen: libwaehip.so carries no link-time dependency on it (a host that already has an
// RCCL in the process -- PyTorch ships its own -- keeps using that one).
#include <dlfcn.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <functional>
#include <future>
#include <mutex>

#include "wae_internal.h"

namespace {

// ---- minimal RCCL binding (types and constants as in <rccl/rccl.h>) ----------------------------------------------------
typedef struct ncclComm *ncclComm_t;
enum { NCCL_SUCCESS = 0 };
enum { NCCL_DOUBLE = 8 };           // ncclFloat64
enum { NCCL_SUM = 0 };
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    int (*CommDestroy)(ncclComm_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
    int (*Reduce)(const void *, void *, size_t, int, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
Rccl &rccl() {
    static Rccl R;
    static std::once_flag once;
    std::call_once(once, []() {
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            R.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (R.lib) break;
        }
        if (!R.lib) return;
        R.CommInitAll = (decltype(R.CommInitAll))dlsym(R.lib, "ncclCommInitAll");
        R.CommDestroy = (decltype(R.CommDestroy))dlsym(R.lib, "ncclCommDestroy");
        R.GroupStart = (decltype(R.GroupStart))dlsym(R.lib, "ncclGroupStart");
        R.GroupEnd = (decltype(R.GroupEnd))dlsym(R.lib, "ncclGroupEnd");
        R.AllGather = (decltype(R.AllGather))dlsym(R.lib, "ncclAllGather");
        R.Reduce = (decltype(R.Reduce))dlsym(R.lib, "ncclReduce");
        R.GetErrorString = (decltype(R.GetErrorString))dlsym(R.lib, "ncclGetErrorString");
    });
    return R;
}
void nccl_check(int rc, const char *what) {
    if (rc != NCCL_SUCCESS) {
        const Rccl &R = rccl();
        throw WaeError(WAE_ERR_HIP, std::string("RCCL ") + what + ": " + (R.GetErrorString ? R.GetErrorString(rc) : "error"));
    }
}

// communicators are kept per device list (creating them costs ~100 ms)
struct CommSet {
    std::vector<int> devs;
    std::vector<ncclComm_t> comms;
};
std::vector<ncclComm_t> comms_for(const std::vector<int> &devs) {       // (by value: the cache may grow under another caller)
    static std::mutex mu;
    static std::vector<CommSet> cache;
    std::lock_guard<std::mutex> lock(mu);
    for (const CommSet &c : cache)
        if (c.devs == devs) return c.comms;
    const Rccl &R = rccl();
    if (!R.lib || !R.CommInitAll || !R.AllGather || !R.Reduce || !R.GroupStart || !R.GroupEnd)
        throw WaeError(WAE_ERR_HIP, "RCCL (librccl.so) could not be loaded: wae_beyn_moments_mgpu needs it for more than one GPU "
                                    "(WAE_MGPU_EXCHANGE=copy exchanges through device-to-device copies instead)");
    CommSet c;
    c.devs = devs;
    c.comms.resize(devs.size());
    nccl_check(R.CommInitAll(c.comms.data(), (int)devs.size(), devs.data()), "ncclCommInitAll");
    cache.push_back(c);
    return c.comms;
}

struct DeviceRestore {               // the entry switches devices: the caller gets its current device back, also on an error
    int dev = 0;
    DeviceRestore() { (void)hipGetDevice(&dev); }
    ~DeviceRestore() { (void)hipSetDevice(dev); }
};

__global__ __launch_bounds__(256) void add_into_kernel(cplx *__restrict__ acc, const cplx *__restrict__ x, size_t n) {
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) { acc[e].x += x[e].x; acc[e].y += x[e].y; }
}

// The two exchange steps of the multi-GPU pass.  Default: RCCL collectives on the devices' streams (xGMI).  WAE_MGPU_EXCHANGE=copy:
// plain device-to-device copies (hipMemcpyPeerAsync) and a sum on device 0 -- no RCCL needed, and the "devices" may then be
// several handles on ONE device ("virtual ranks"): the whole G > 1 logic -- shares, slab merge, basis import, reduction -- runs
// on a one-GPU box (tests/test_gpu_distributed.py).
struct Exchange {
    bool copy = false;
    std::vector<int> devs;
    std::vector<hipStream_t> streams;
    std::vector<ncclComm_t> comms;
    int G() const { return (int)devs.size(); }
    void sync_all() const {
        for (int g = 0; g < G(); ++g) { HIP_CHECK(hipSetDevice(devs[g])); HIP_CHECK(hipStreamSynchronize(streams[g])); }
    }
    // dst[q] (G * count elements, rank-major) <- src[g] (count elements) of every g, on every q
    void all_gather(const std::vector<cplx *> &src, const std::vector<cplx *> &dst, size_t count) const {
        if (!copy) {
            const Rccl &R = rccl();
            struct Group { const Rccl &R; bool open = false; ~Group() { if (open) (void)R.GroupEnd(); } } grp{R};
            nccl_check(R.GroupStart(), "ncclGroupStart");
            grp.open = true;
            for (int g = 0; g < G(); ++g) nccl_check(R.AllGather(src[g], dst[g], count * 2, NCCL_DOUBLE, comms[g], streams[g]), "ncclAllGather");
            grp.open = false;
            nccl_check(R.GroupEnd(), "ncclGroupEnd");
            return;
        }
        sync_all();                                              // every source is complete
        for (int q = 0; q < G(); ++q) {
            HIP_CHECK(hipSetDevice(devs[q]));
            for (int g = 0; g < G(); ++g)
                HIP_CHECK(hipMemcpyPeerAsync(dst[q] + (size_t)g * count, devs[q], src[g], devs[g], count * sizeof(cplx), streams[q]));
        }
        sync_all();
    }
    // bufs[0] <- sum over g of bufs[g]  (fixed order g = 1, 2, ...: deterministic)
    void reduce_to_first(const std::vector<cplx *> &bufs, size_t count) const {
        if (!copy) {
            const Rccl &R = rccl();
            struct Group { const Rccl &R; bool open = false; ~Group() { if (open) (void)R.GroupEnd(); } } grp{R};
            nccl_check(R.GroupStart(), "ncclGroupStart");
            grp.open = true;
            for (int g = 0; g < G(); ++g) nccl_check(R.Reduce(bufs[g], bufs[g], count * 2, NCCL_DOUBLE, NCCL_SUM, 0, comms[g], streams[g]), "ncclReduce");
            grp.open = false;
            nccl_check(R.GroupEnd(), "ncclGroupEnd");
            sync_all();
            return;
        }
        sync_all();
        if (G() == 1) return;
        HIP_CHECK(hipSetDevice(devs[0]));
        DevBuf<cplx> tmp;
        tmp.alloc(count);
        for (int g = 1; g < G(); ++g) {
            HIP_CHECK(hipMemcpyPeerAsync(tmp.p, devs[0], bufs[g], devs[g], count * sizeof(cplx), streams[0]));
            hipLaunchKernelGGL(add_into_kernel, dim3(4096), dim3(256), 0, streams[0], bufs[0], tmp.p, count);
            HIP_CHECK(hipGetLastError());
        }
        HIP_CHECK(hipStreamSynchronize(streams[0]));
    }
};

// slabs[g][s][row][c_local]  ->  store[s][row][g*ls + c_local]   (the probe columns back in order after the all-gather)
__global__ __launch_bounds__(256) void merge_slabs_kernel(const cplx *__restrict__ slabs, cplx *__restrict__ store, int G, int S, int64_t d, int ls) {
    const size_t total = (size_t)S * d * G * ls;
    const int l = G * ls;
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const int c = (int)(e % l);
        const size_t sr = e / l;                             // s * d + row
        const int g = c / ls, cl = c - g * ls;
        store[e] = slabs[((size_t)g * S * d + sr) * ls + cl];
    }
}

// allraw[src][s][row][c] (c < l: every rank's raw snapshots, all columns)  ->  mine[(src * per + s)][row][cl] = column g * ls + cl:
// this rank's column slice of ALL snapshots (hybrid split: points for the solves, columns for the basis)
__global__ __launch_bounds__(256) void slice_cols_kernel(const cplx *__restrict__ allraw, cplx *__restrict__ mine, size_t rows_total, int l, int ls, int g) {
    const size_t total = rows_total * ls;                    // rows_total = G * per * d
    for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
        const size_t r = e / ls;
        const int cl = (int)(e - r * ls);
        mine[e] = allraw[r * l + (size_t)g * ls + cl];
    }
}

// S snapshot points spread evenly through n points, bit-reversal ("every prefix covers the contour") order; the rest
void snapshot_plan(int n, int S, std::vector<int> &snap, std::vector<int> &rest) {
    S = std::min(S, n);
    std::vector<int> idx;
    for (int i = 0; i < S; ++i) {
        const int v = (int)(((double)i + 0.5) * n / std::max(S, 1));
        if (idx.empty() || idx.back() != v) idx.push_back(v);
    }
    std::vector<char> is_snap(n, 0);
    for (int v : idx) is_snap[v] = 1;
    rest.clear();
    for (int i = 0; i < n; ++i)
        if (!is_snap[i]) rest.push_back(i);
    const int m = (int)idx.size();
    if (m < 3) { snap = idx; return; }
    int bits = 1;
    while ((1 << bits) < m) ++bits;
    std::vector<std::pair<int, int>> key(m);
    for (int i = 0; i < m; ++i) {
        int r = 0;
        for (int b = 0; b < bits; ++b)
            if (i >> b & 1) r |= 1 << (bits - 1 - b);
        key[i] = {r, i};
    }
    std::stable_sort(key.begin(), key.end(), [](const std::pair<int, int> &a, const std::pair<int, int> &b) { return a.first < b.first; });
    snap.resize(m);
    for (int i = 0; i < m; ++i) snap[i] = idx[key[i].second];
}

template <class T> std::vector<T> take(const T *src, const std::vector<int> &rows, int width) {
    std::vector<T> out((size_t)rows.size() * width);
    for (size_t i = 0; i < rows.size(); ++i) std::copy(src + (size_t)rows[i] * width, src + (size_t)(rows[i] + 1) * width, out.begin() + i * width);
    return out;
}

}   // namespace

extern "C" int wae_beyn_moments_mgpu(wae_family *const *handles, int32_t ngpu, int32_t npts, const double *z, const double *w, const double *coeff_table,
                                     const double *V, int32_t l, int32_t K, double tol, int32_t maxit, int32_t nsnap, double *A_out,
                                     wae_solve_info *info) {
    try {
        WAE_REQUIRE(handles && ngpu >= 1 && ngpu <= 64 && npts >= 0 && (npts == 0 || (z && w && coeff_table)) && V && l > 0 && K > 0 && A_out,
                    "bad argument");
        DeviceRestore restore_device;
        Exchange ex;
        {
            const char *e = getenv("WAE_MGPU_EXCHANGE");
            ex.copy = e && std::string(e) == "copy";
        }
        int64_t d = 0;
        int32_t T = 0;
        std::vector<int> devs(ngpu);
        std::vector<hipStream_t> streams(ngpu);
        for (int g = 0; g < ngpu; ++g) {
            WAE_REQUIRE(handles[g], "null handle");
            int64_t dg = 0;
            int32_t Tg = 0;
            if (wae_family_info(handles[g], &dg, &Tg, nullptr) != WAE_OK) throw WaeError(WAE_ERR_INVALID, wae_last_error());
            if (g == 0) { d = dg; T = Tg; }
            WAE_REQUIRE(dg == d && Tg == T, "the handles are not replicas of one family");
            devs[g] = wae_internal_device(handles[g]);
            streams[g] = wae_internal_stream(handles[g]);
            for (int q = 0; q < g; ++q) {
                WAE_REQUIRE(handles[q] != handles[g], "one handle passed twice");
                WAE_REQUIRE(ex.copy || devs[q] != devs[g], "two handles on one device (only with WAE_MGPU_EXCHANGE=copy)");
            }
        }
        ex.devs = devs;
        ex.streams = streams;
        if (!ex.copy) ex.comms = comms_for(devs);
        const int npow = 2 * K;
        const size_t acnt = (size_t)d * l * npow;
        wae_solve_info tot;
        memset(&tot, 0, sizeof(tot));
        const auto t_begin = std::chrono::steady_clock::now();
        // per-device moment buffers
        std::vector<DevBuf<cplx>> Ad(ngpu), local(ngpu), slabs(ngpu), store(ngpu);
        for (int g = 0; g < ngpu; ++g) {
            HIP_CHECK(hipSetDevice(devs[g]));
            Ad[g].alloc(acnt);
            HIP_CHECK(hipMemsetAsync(Ad[g].p, 0, acnt * sizeof(cplx), streams[g]));
            HIP_CHECK(hipStreamSynchronize(streams[g]));
        }
        std::vector<int> snap, rest;
        if (nsnap > 0 && npts >= 2 * nsnap) snapshot_plan(npts, nsnap, snap, rest);
        else { rest.resize(npts); for (int i = 0; i < npts; ++i) rest[i] = i; }
        const int S = (int)snap.size();
        // runs f(g) on one host thread per device and folds the solve statistics / the first error
        auto on_all = [&](const std::function<int(int, wae_solve_info *)> &f) {
            std::vector<std::future<std::pair<int, std::string>>> jobs;
            std::vector<wae_solve_info> infos(ngpu);
            for (int g = 0; g < ngpu; ++g)
                jobs.push_back(std::async(std::launch::async, [&, g]() {
                    memset(&infos[g], 0, sizeof(wae_solve_info));
                    const int rc = f(g, &infos[g]);
                    return std::make_pair(rc, rc < 0 ? std::string(wae_last_error()) : std::string());
                }));
            int worst = 0;
            std::string msg;
            for (int g = 0; g < ngpu; ++g) {
                auto r = jobs[g].get();
                if (r.first < 0 && worst >= 0) { worst = r.first; msg = "device " + std::to_string(devs[g]) + ": " + r.second; }
                else if (r.first > worst && worst >= 0) worst = r.first;
                tot.iters_max = std::max(tot.iters_max, infos[g].iters_max);
                tot.iters_total += infos[g].iters_total;
                tot.n_unconverged += infos[g].n_unconverged;
                tot.levels = std::max(tot.levels, infos[g].levels);
                tot.relres_max = std::max(tot.relres_max, infos[g].relres_max);
            }
            if (worst < 0) throw WaeError(worst, msg);
            return worst;
        };
        int code = 0;
        const size_t vecl_all = (size_t)d * l;
        if (S == 0) {
            // no snapshot scheme: every device integrates its share of the points from zero guesses
            code = std::max(code, on_all([&](int g, wae_solve_info *li) {
                std::vector<int> mine;
                for (size_t i = g; i < rest.size(); i += ngpu) mine.push_back(rest[i]);
                const auto zz = take(z, mine, 2), ww = take(w, mine, 2), cc = take(coeff_table, mine, 2 * T);
                return wae_beyn_moments(handles[g], (int32_t)mine.size(), zz.data(), ww.data(), cc.data(), V, l, K, tol, maxit, nullptr,
                                        (uint64_t)(uintptr_t)Ad[g].p, li);
            }));
        } else if (l % ngpu == 0) {
            // The probe columns divide over the devices.  Snapshot phase, WAE_SNAPSHOT_SPLIT:
            //   "columns" (default): device g solves all S points for its columns progressively (mode 0) -- fewer, longer, narrower
            //       recurrences (one rank's share of the 1M-unknown case at 8 GPUs: 0.35 s against 0.16 s ideal);
            //   "hybrid": POINTS for the solves -- device g solves its Sb / ngpu snapshot points for ALL l columns as full-width
            //       batches from zero guesses (mode 3) --, the raw solutions are gathered, and COLUMNS for the basis: device g
            //       orthonormalises and projects all Sb snapshots of ITS l / ngpu columns (mode 4).  Measured (dev/c3_rank_share.py):
            //       0.36 s at 8 GPUs, 0.63 at 4, 1.38 at 2 against 0.35 / 0.54 / 0.90: without the progressive guesses the solves
            //       need half as many iterations again.  An option, not the default.
            // Either way each device then holds a finished basis for its columns: (2) all-gather of the basis vectors, exchange of
            // the small projected terms on the host; (3) the remaining points round-robin, every system from the projection on
            // the full basis.
            const int ls = l / ngpu;
            const char *se = getenv("WAE_SNAPSHOT_SPLIT");
            const bool hybrid = S >= ngpu && se && std::string(se) == "hybrid";
            const int Sb = hybrid ? (S / ngpu) * ngpu : S;      // snapshots in the basis (equal shares); the others join the remaining points
            if (hybrid)
                for (int i = Sb; i < S; ++i) rest.push_back(snap[i]);
            const int S = Sb;                                   // (shadows: from here on the basis size)
            const size_t slab = (size_t)S * d * ls;
            for (int g = 0; g < ngpu; ++g) { HIP_CHECK(hipSetDevice(devs[g])); local[g].alloc(slab); slabs[g].alloc(slab * ngpu); store[g].alloc(slab * ngpu); }
            if (hybrid) {
                const int per = S / ngpu;
                const size_t rawn = (size_t)per * vecl_all;
                std::vector<DevBuf<cplx>> raw(ngpu), allraw(ngpu);
                for (int g = 0; g < ngpu; ++g) { HIP_CHECK(hipSetDevice(devs[g])); raw[g].alloc(rawn); allraw[g].alloc(rawn * ngpu); }
                code = std::max(code, on_all([&](int g, wae_solve_info *li) {
                    std::vector<int> mine;
                    for (int i = g; i < S; i += ngpu) mine.push_back(snap[i]);
                    const auto zz = take(z, mine, 2), ww = take(w, mine, 2), cc = take(coeff_table, mine, 2 * T);
                    return wae_beyn_moments_rb(handles[g], per, zz.data(), ww.data(), cc.data(), V, l, K, tol, maxit, 3, per, 0,
                                               (uint64_t)(uintptr_t)raw[g].p, nullptr, (uint64_t)(uintptr_t)Ad[g].p, 1, 0, 0, li);
                }));
                {
                    std::vector<cplx *> src(ngpu), dst(ngpu);
                    for (int g = 0; g < ngpu; ++g) { src[g] = raw[g].p; dst[g] = allraw[g].p; }
                    ex.all_gather(src, dst, rawn);
                }
                for (int g = 0; g < ngpu; ++g) {
                    HIP_CHECK(hipSetDevice(devs[g]));
                    hipLaunchKernelGGL(slice_cols_kernel, dim3(4096), dim3(256), 0, streams[g], allraw[g].p, local[g].p, (size_t)S * d, l, ls, g);
                    HIP_CHECK(hipGetLastError());
                }
                ex.sync_all();
                std::vector<int> order;                          // the snapshot point behind every slot: [source device][its points]
                for (int g = 0; g < ngpu; ++g)
                    for (int i = g; i < S; i += ngpu) order.push_back(snap[i]);
                const auto zo = take(z, order, 2), wo = take(w, order, 2), co = take(coeff_table, order, 2 * T);
                code = std::max(code, on_all([&](int g, wae_solve_info *li) {
                    return wae_beyn_moments_rb(handles[g], S, zo.data(), wo.data(), co.data(), V + (size_t)2 * d * g * ls, ls, K, tol, maxit, 4, S, S,
                                               (uint64_t)(uintptr_t)local[g].p, nullptr, (uint64_t)(uintptr_t)Ad[g].p, 1, l, g * ls, li);
                }));
                for (int g = 0; g < ngpu; ++g) { HIP_CHECK(hipSetDevice(devs[g])); raw[g].release(); allraw[g].release(); }
            } else {
                const auto zs = take(z, snap, 2), ws = take(w, snap, 2), cs = take(coeff_table, snap, 2 * T);
                code = std::max(code, on_all([&](int g, wae_solve_info *li) {
                    return wae_beyn_moments_rb(handles[g], S, zs.data(), ws.data(), cs.data(), V + (size_t)2 * d * g * ls, ls, K, tol, maxit, 0, S, 0,
                                               (uint64_t)(uintptr_t)local[g].p, nullptr, (uint64_t)(uintptr_t)Ad[g].p, 1, l, g * ls, li);
                }));
            }
            {
                std::vector<cplx *> src(ngpu), dst(ngpu);
                for (int g = 0; g < ngpu; ++g) { src[g] = local[g].p; dst[g] = slabs[g].p; }
                ex.all_gather(src, dst, slab);
            }
            // projected terms and right-hand-side projections of every device's columns
            int32_t Sx = 0, lx = 0, nk = 0;
            if (wae_rb_export(handles[0], &Sx, &lx, &nk, nullptr, nullptr, nullptr) != WAE_OK) throw WaeError(WAE_ERR_INVALID, wae_last_error());
            WAE_REQUIRE(Sx == S && lx == ls, "snapshot basis has an unexpected shape");
            std::vector<int32_t> kact(std::max(nk, 1));
            std::vector<double> Hk_all((size_t)nk * S * S * l * 2), g_all((size_t)S * l * 2), Hk((size_t)nk * S * S * ls * 2), gg((size_t)S * ls * 2);
            for (int g = 0; g < ngpu; ++g) {
                int32_t S2, l2, nk2;
                if (wae_rb_export(handles[g], &S2, &l2, &nk2, kact.data(), Hk.data(), gg.data()) != WAE_OK) throw WaeError(WAE_ERR_INVALID, wae_last_error());
                WAE_REQUIRE(S2 == S && l2 == ls && nk2 == nk, "the devices' snapshot bases differ in shape");
                for (size_t a = 0; a < (size_t)nk * S * S; ++a)           // [ki][s][i][c]: c fastest
                    std::copy(Hk.begin() + a * ls * 2, Hk.begin() + (a + 1) * ls * 2, Hk_all.begin() + (a * l + (size_t)g * ls) * 2);
                for (size_t a = 0; a < (size_t)S; ++a)
                    std::copy(gg.begin() + a * ls * 2, gg.begin() + (a + 1) * ls * 2, g_all.begin() + (a * l + (size_t)g * ls) * 2);
            }
            for (int g = 0; g < ngpu; ++g) {
                HIP_CHECK(hipSetDevice(devs[g]));
                hipLaunchKernelGGL(merge_slabs_kernel, dim3(4096), dim3(256), 0, streams[g], slabs[g].p, store[g].p, ngpu, S, d, ls);
                HIP_CHECK(hipGetLastError());
                HIP_CHECK(hipStreamSynchronize(streams[g]));
                if (wae_rb_import(handles[g], S, l, (uint64_t)(uintptr_t)store[g].p, nk, kact.data(), Hk_all.data(), g_all.data()) != WAE_OK)
                    throw WaeError(WAE_ERR_INVALID, wae_last_error());
            }
            code = std::max(code, on_all([&](int g, wae_solve_info *li) {
                std::vector<int> mine;
                for (size_t i = g; i < rest.size(); i += ngpu) mine.push_back(rest[i]);
                const auto zz = take(z, mine, 2), ww = take(w, mine, 2), cc = take(coeff_table, mine, 2 * T);
                return wae_beyn_moments_rb(handles[g], (int32_t)mine.size(), zz.data(), ww.data(), cc.data(), V, l, K, tol, maxit, 2, S, 0,
                                           (uint64_t)(uintptr_t)store[g].p, nullptr, (uint64_t)(uintptr_t)Ad[g].p, 1, 0, 0, li);
            }));
        } else {
            // l not divisible by the number of devices: the snapshot POINTS are split, the raw snapshots all-gathered, and every
            // device rebuilds the (same) basis from them (mode 1)
            const int per = S / ngpu;
            WAE_REQUIRE(per >= 1, "fewer snapshot points than devices");
            const int Su = per * ngpu;                          // snapshots actually used; the others join the remaining points
            for (int i = Su; i < S; ++i) rest.push_back(snap[i]);
            const size_t slab = (size_t)per * vecl_all;
            for (int g = 0; g < ngpu; ++g) { HIP_CHECK(hipSetDevice(devs[g])); local[g].alloc(slab); store[g].alloc(slab * ngpu); }
            code = std::max(code, on_all([&](int g, wae_solve_info *li) {
                std::vector<int> mine;
                for (int i = g; i < Su; i += ngpu) mine.push_back(snap[i]);
                const auto zz = take(z, mine, 2), ww = take(w, mine, 2), cc = take(coeff_table, mine, 2 * T);
                return wae_beyn_moments_rb(handles[g], per, zz.data(), ww.data(), cc.data(), V, l, K, tol, maxit, 0, per, 0,
                                           (uint64_t)(uintptr_t)local[g].p, nullptr, (uint64_t)(uintptr_t)Ad[g].p, 1, 0, 0, li);
            }));
            {
                std::vector<cplx *> src(ngpu), dst(ngpu);
                for (int g = 0; g < ngpu; ++g) { src[g] = local[g].p; dst[g] = store[g].p; }
                ex.all_gather(src, dst, slab);
            }
            ex.sync_all();
            code = std::max(code, on_all([&](int g, wae_solve_info *li) {
                std::vector<int> mine;
                for (size_t i = g; i < rest.size(); i += ngpu) mine.push_back(rest[i]);
                const auto zz = take(z, mine, 2), ww = take(w, mine, 2), cc = take(coeff_table, mine, 2 * T);
                return wae_beyn_moments_rb(handles[g], (int32_t)mine.size(), zz.data(), ww.data(), cc.data(), V, l, K, tol, maxit, 1, Su, Su,
                                           (uint64_t)(uintptr_t)store[g].p, nullptr, (uint64_t)(uintptr_t)Ad[g].p, 1, 0, 0, li);
            }));
        }
        // sum of the partial moment tensors on device 0 (in place), then to the host
        {
            std::vector<cplx *> bufs(ngpu);
            for (int g = 0; g < ngpu; ++g) bufs[g] = Ad[g].p;
            ex.reduce_to_first(bufs, acnt);
        }
        HIP_CHECK(hipSetDevice(devs[0]));
        HIP_CHECK(hipMemcpy(A_out, Ad[0].p, acnt * sizeof(cplx), hipMemcpyDeviceToHost));
        for (int g = 0; g < ngpu; ++g) {                        // buffers are freed on their own device
            HIP_CHECK(hipSetDevice(devs[g]));
            Ad[g].release(); local[g].release(); slabs[g].release(); store[g].release();
        }
        tot.seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count();
        if (info) *info = tot;
        return tot.n_unconverged > 0 ? std::max(code, (int)WAE_WARN_MAXITER) : WAE_OK;
    } catch (const WaeError &e) {
        wae_set_error(e.what());
        return e.code;
    } catch (const std::exception &e) {
        wae_set_error(e.what());
        return WAE_ERR_INVALID;
    }
}
