// epgx_api.hip -- host side of libepgx.so: the C ABI declared in include/epgx.h.
//
// Every entry point validates its operands against what the kernels and their grids assume
// BEFORE anything is launched (shapes, opcode range, coefficient-table bounds, shift range,
// signal slots), returns a negative status instead of throwing, and records a thread-local
// message for epgx_last_error().  There is no CPU execution path in this library.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types only: librccl.so.1 is loaded on first use (rccl_api)
#include <dlfcn.h>
#include <sched.h>
#include <pthread.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <functional>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <atomic>
#include <string>
#include <thread>
#include <map>
#include <mutex>
#include <unordered_map>
#include <utility>
#include <vector>

#include "epgx_kernels.hip.h"
#include "epgx_small_kernels.hip.h"
#include "epgx_deriv_kernels.hip.h"
#include "epgx_launch.h"
#include "epgx_launch_grow.h"

using namespace epgx;

// ------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(e_ == hipErrorOutOfMemory ? EPGX_ERR_NOMEM : EPGX_ERR_HIP,           \
                        "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,     \
                        __LINE__);                                                           \
    } while (0)

// Measurement knobs of the selection (environment, read ONCE per process): every one defaults to the kernel the library would
// take anyway; tools/ab_kernels.sh flips them for A/B runs.  EPGX_TRACE is read per call (tests switch it on and off).
namespace {
struct Knobs {
    bool rows, rows_deriv, rows_deriv2, drun, runs, grow, contig, split, prefetch;
    int grow_min;
    bool fold;
    double grow_share;
    bool lead_forward;
    int slab_voxels; // EPGX_SLAB_VOXELS (tests): voxels per slab of the two-leg launch at 2048 orders (0: as many as 8 GiB of scratch hold)
    bool split_grow; // EPGX_SPLIT_GROW (default 1): K = 2048 in two legs where it pays (one wavefront per voxel up to 512 populated orders)
    int cgrow;      // EPGX_CGROW: 0 off, 1 (default): growing launches at K = 256 .. 1024, at K = 128 when 60 % of the records run below 64 orders; 2: at K = 128 whenever the other capacities would
};
int env_int(const char *name, int fallback) {
    const char *v = getenv(name);
    return v ? atoi(v) : fallback;
}
const Knobs &knobs() {
    static const Knobs k = {env_int("EPGX_ROWS", 1) != 0,   env_int("EPGX_ROWS_DERIV", 1) != 0, env_int("EPGX_ROWS_DERIV2", 1) != 0,
                            env_int("EPGX_DRUN", 1) != 0,   env_int("EPGX_RUNS", 1) != 0,       env_int("EPGX_GROW", 1) != 0,
                            env_int("EPGX_CONTIG", 1) != 0, env_int("EPGX_SPLIT", 1) != 0,      env_int("EPGX_PREFETCH", 1) != 0,
                            env_int("EPGX_GROW_MIN", 1),    env_int("EPGX_FOLD", 1) != 0,
                            getenv("EPGX_GROW_SHARE") ? atof(getenv("EPGX_GROW_SHARE")) : 0.1, env_int("EPGX_LEAD_FORWARD", 1) != 0,
                            env_int("EPGX_SLAB_VOXELS", 0), env_int("EPGX_SPLIT_GROW", 1) != 0, env_int("EPGX_CGROW", 1)};
    return k;
}
bool tracing() { return getenv("EPGX_TRACE") != nullptr; }

}  // namespace

// ------------------------------------------------------------------------------ objects
struct epgx_ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t own = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipDeviceProp_t prop;
    // caching device allocator: freed blocks are kept and handed out again.  hipMalloc / hipFree
    // cost 1-25 ms each at the sizes of a plan (measured; hipFree also synchronises the device),
    // which dominated a repeated simulate() of the same shape.  Every use of a block is ordered on
    // ctx->stream, so a block can be recycled without waiting for the work that last touched it.
    // (`mem` guards the three containers: ctypes releases the GIL during calls, and one context may be
    // shared by several host threads -- their launches then interleave on the one stream, which is legal)
    std::mutex mem;
    // page-locked host blocks (epgx_host_alloc), recycled the same way; copy stream + events of epgx_run_to_host
    std::vector<std::pair<void *, size_t>> host_cache;
    std::unordered_map<void *, size_t> host_live;
    size_t host_cached_bytes = 0;
    hipStream_t copy_stream = nullptr;
    std::vector<hipEvent_t> slab_events;
    // one slab pipeline at a time per CONTEXT (copy stream, events and staging ring are shared by its host threads);
    // pipelines of different contexts / GPUs run side by side
    std::mutex pipeline;
    // ring of page-locked staging blocks for downloads into ordinary (pageable) host memory (epgx_run_to_host)
    struct Stage { void *host = nullptr; hipEvent_t done = nullptr; };
    std::vector<Stage> stages;
    size_t stage_bytes = 0;
    std::vector<std::pair<void *, size_t>> cache;
    std::unordered_map<void *, size_t> live;
    size_t cached_bytes = 0;
};

static void dev_release_cache_locked(epgx_ctx *ctx) {
    if (ctx->cache.empty()) return;
    (void)hipStreamSynchronize(ctx->stream);
    for (auto &b : ctx->cache) (void)hipFree(b.first);
    ctx->cache.clear();
    ctx->cached_bytes = 0;
}

static void dev_release_cache(epgx_ctx *ctx) {
    std::lock_guard<std::mutex> guard(ctx->mem);
    dev_release_cache_locked(ctx);
}

static hipError_t dev_alloc(epgx_ctx *ctx, void **out, size_t bytes) {
    std::lock_guard<std::mutex> guard(ctx->mem);
    bytes = std::max<size_t>((bytes + 255) & ~(size_t)255, 256);
    int best = -1;
    for (int i = 0; i < (int)ctx->cache.size(); ++i) {
        const size_t n = ctx->cache[i].second;
        if (n >= bytes && n <= bytes + std::max<size_t>(bytes / 4, (size_t)1 << 20) &&
            (best < 0 || n < ctx->cache[best].second))
            best = i;
    }
    if (best >= 0) {
        *out = ctx->cache[best].first;
        ctx->live[*out] = ctx->cache[best].second;
        ctx->cached_bytes -= ctx->cache[best].second;
        ctx->cache.erase(ctx->cache.begin() + best);
        return hipSuccess;
    }
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipErrorOutOfMemory) {   // give the cached blocks back and try once more
        (void)hipGetLastError();
        dev_release_cache_locked(ctx);
        e = hipMalloc(out, bytes);
    }
    if (e == hipSuccess) ctx->live[*out] = bytes;
    return e;
}

static void dev_free(epgx_ctx *ctx, void *p) {
    if (!p) return;
    std::lock_guard<std::mutex> guard(ctx->mem);
    auto it = ctx->live.find(p);
    if (it == ctx->live.end()) {   // not ours (should not happen): plain free
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(p);
        return;
    }
    const size_t n = it->second;
    ctx->live.erase(it);
    ctx->cache.emplace_back(p, n);
    ctx->cached_bytes += n;
    // Keep at most half of the HBM (evict the largest block first) and 96 blocks.  Over the count: the OLDEST blocks below 1 GiB
    // go, in one batch down to 64 blocks behind ONE stream synchronisation -- a small block is cheap to allocate again, the
    // block that was just freed (the likeliest to be asked for next) stays, and the hot path pays one synchronisation per 32
    // frees instead of one per free.  Multi-GB blocks are never evicted on the count rule: giving one back to HIP has a
    // lasting price on this platform -- after one hipFree of 16 GB every later copy into page-locked host memory ran at 28.6
    // instead of 54 GB/s for the rest of the process (tools/release_probe.py; round 3's bench showed it: seven short PGSE calls
    // pushed the count past the cap, the 16 GB signal of the MRF leg went, and the end-to-end leg after it fell from 6.9 to
    // 20.7 ms).
    const size_t limit = (size_t)ctx->prop.totalGlobalMem / 2;
    bool synced = false;
    auto evict = [&](size_t victim) {
        if (!synced) (void)hipStreamSynchronize(ctx->stream);
        synced = true;
        (void)hipFree(ctx->cache[victim].first);
        ctx->cached_bytes -= ctx->cache[victim].second;
        ctx->cache.erase(ctx->cache.begin() + (std::ptrdiff_t)victim);
    };
    while (!ctx->cache.empty() && ctx->cached_bytes > limit) {
        size_t victim = 0;
        for (size_t i = 1; i < ctx->cache.size(); ++i)
            if (ctx->cache[i].second > ctx->cache[victim].second) victim = i;
        evict(victim);
    }
    if (ctx->cache.size() > 96) {
        for (size_t i = 0; i < ctx->cache.size() && ctx->cache.size() > 64;) {      // (the vector is in order of freeing: oldest first)
            if (ctx->cache[i].second < ((size_t)1 << 30)) evict(i);
            else ++i;
        }
    }
}

// one operator range [begin, end) packed into fused records for capacity K, resident on the device
struct PackedRange {
    int begin = 0, end = 0, K = 0;
    Rec *d_recs = nullptr;
    DRec *d_drecs = nullptr;  // derivative plans only
    int n_rec = 0;
    Rec *d_runs = nullptr;    // the same records with runs of identical ones folded (rows_kernel<.., RUNS>), or null
    int n_runs = 0;
    Rec *d_grow = nullptr;    // K = 64: the run-folded records cut where the populated orders outgrow 16 and 32 (rows_grow_kernel), or null
    int n_grow = 0, grow1 = 0, grow2 = 0;   // records [0, grow1) run at 16 orders per voxel, [grow1, grow2) at 32, the rest at 64
    Rec *d_druns = nullptr;   // derivative plans, K = 64: the records with a header in front of every run of same-shape
    DRec *d_ddruns = nullptr; // fused-echo records (drun_kernel), and their DRecs (a header's is all zero); or null
    DRecB *d_bdruns = nullptr; // ... and, when the runs are of records folded at run time (DRUN_FOLD), E_b's logarithmic partials
    int n_druns = 0;
    int drun_code = 0;        // the run shape the headers of d_druns announce (drun_kernel is instantiated per shape)
    // K = 128 .. 1024 from equilibrium (run_contig_grow_kernel): records [0, cgrow[0]) run while at most 64 orders can hold anything,
    // [cgrow[0], cgrow[1]) at most 128, [cgrow[1], cgrow[2]) at most 256, [cgrow[2], cgrow[3]) at most 512; cgrow_share = the share
    // of the records below the capacity
    int cgrow[6] = {0, 0, 0, 0, 0, 0};   // (cgrow[3]: K = 2048, where the second leg starts; cgrow[4], cgrow[5]: at most 1024, 1536 -- where parts 2 and 3 of run_split_kernel join)
    double cgrow_share = 0.0;
    int cgrow_adc3 = 0;       // probe records in front of record cgrow[3] (K = 2048: the first row the second leg writes)
    int dgrow1 = 0, dgrow2 = 0;   // fused echoes from equilibrium: entries [0, dgrow1) of d_druns run with one order per lane, [dgrow1, dgrow2) with two
    int drun_inside = 0, drun_headers = 0, drun_ident = 0;   // records inside runs, runs, runs that repeat one record (EPGX_TRACE)
    bool use_lds = false, has_adc = false, has_pd = false;
    bool has_gs = false;     // some record is a gather shift (three staged arrays per wavefront instead of two)
    bool big_shift = false;  // some record shifts by |n| >= 2 (use_lds is also set by gather shifts)
    bool seq_slots = false;  // the ADC slots of the range are first_slot, first_slot + 1, ...
    int first_slot = 0;
    int pf_count = 0;        // 1 + index of the last record that refers to a per-voxel table for the first time
};

struct epgx_plan {
    epgx_ctx *ctx = nullptr;
    std::vector<epgx_op> ops;  // host copy of the primitive stream (validation, packing)
    std::vector<uint8_t> zero_pattern;  // per op: 1 / 3 = T table with the TX / TY pattern (plan_create), 2 = E table with Im e0 == 0
    std::vector<std::vector<int32_t>> gather_tables;  // per op: host copy of an EPGX_OP_GS table (validation)
    std::vector<epgx_dop> dops;  // first-order partials per op (n_vars > 0)
    std::vector<uint16_t> dpattern; // per op: bits 2v, 2v + 1 = zero pattern of variable v's partial table (1: phi = 0 / real E, 2: real matrix);
                                    // bit 8 + v: the partial is a generated one (14 per entry: with the partial of the constant term)
    int32_t deriv_flags = 0;
    int32_t n_vars = 0;
    std::vector<PackedRange> packed;
    double *d_coef = nullptr;
    int64_t n_coef = 0;
    int64_t n_pool = 0;       // doubles in the device pool: n_coef + the device-generated part; behind it 32 doubles of
                              // padding that start with the identity relaxation {1, 0, 1, 0} (folded records)
    bool fold = true;         // fold precession-free relaxations into neighbouring rotations at run time (pack_records)
    // derivative plans: logarithmic partials (wT, wL per entry, logtab_kernel) of the real relaxation tables that carry a real
    // partial over the same index space -- what drun_kernel's folded records read.  Behind the pool's padding.
    struct LogTab {
        int64_t off = -1;     // doubles from the pool's base
        int space = -1;
        uint32_t any = 0;     // 1: some wT != 0, 2: some wL != 0
    };
    std::vector<LogTab> logtabs;
    std::vector<int32_t> log_of;   // [op * EPGX_MAX_VARS + v] -> index into logtabs, or -1
    // EPGX_OP_T0 operators whose table the host had fused (E_a . T . E_b, epgx_fuse) and whose partial w.r.t. variable v comes
    // from the relaxations alone (epgx_fuse_partial chain without a rotation partial): the log tables of E_a and E_b
    // ([(op * EPGX_MAX_VARS + v) * 2 + {0: a, 1: b}], -1: that side has no partial), or empty.  t0_logd[op * MAX_VARS + v]
    // says whether the variable can take the logarithmic route at all.
    std::vector<int32_t> t0_log;
    std::vector<uint8_t> t0_logd;
    int64_t n_log = 0;             // doubles of log tables behind n_pool + 32
    int32_t ndim = 0, n_spaces = 0, n_adc = 0;
    int64_t shape[EPGX_MAX_DIMS];
    int64_t strides[EPGX_MAX_SPACES][EPGX_MAX_DIMS];
    int64_t nvox_total = 0;
    uint32_t dense_spaces = 0;  // bit s: space s has the grid's own C-order strides
    // `cache_lock` guards `packed` and the vidx cache (two host threads running ranges of one plan)
    std::mutex cache_lock;
    // cached table indices for one voxel range
    int32_t *d_vidx = nullptr;
    int64_t vidx_vox0 = -1, vidx_nvox = 0, vidx_cap = 0;
};

struct epgx_state {
    epgx_ctx *ctx = nullptr;
    int64_t nvox = 0;
    int32_t K = 0;
    d2 *data = nullptr;
    double *dens = nullptr;
};

static int set_device(const epgx_ctx *ctx) {
    HIP_TRY(hipSetDevice(ctx->device));
    return EPGX_OK;
}

static bool supported_K(int K) {
    return K == 64 || K == 128 || K == 256 || K == 512 || K == 1024;
}

// ------------------------------------------------------------------------------ library
extern "C" int epgx_abi_version(void) { return EPGX_ABI_VERSION; }

extern "C" const char *epgx_last_error(void) { return g_err; }

extern "C" int epgx_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return n;
}

extern "C" int epgx_ctx_create(int device, epgx_ctx **out) {
    if (!out) return fail(EPGX_ERR_INVALID, "epgx_ctx_create: out is NULL");
    *out = nullptr;
    int n = epgx_device_count();
    if (n <= 0)
        return fail(EPGX_ERR_NODEVICE,
                    "epgx_ctx_create: no HIP device visible (libepgx has no CPU fallback)");
    if (device < 0 || device >= n)
        return fail(EPGX_ERR_INVALID, "epgx_ctx_create: device %d out of range [0,%d)", device, n);
    epgx_ctx *ctx = new (std::nothrow) epgx_ctx();
    if (!ctx) return fail(EPGX_ERR_NOMEM, "epgx_ctx_create: host allocation failed");
    ctx->device = device;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipGetDeviceProperties(&ctx->prop, device);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->own, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev0);
    if (e == hipSuccess) e = hipEventCreate(&ctx->ev1);
    if (e != hipSuccess) {
        delete ctx;
        return fail(EPGX_ERR_HIP, "epgx_ctx_create: %s", hipGetErrorString(e));
    }
    if (ctx->prop.warpSize != 64) {
        int w = ctx->prop.warpSize;
        epgx_ctx_destroy(ctx);
        return fail(EPGX_ERR_NODEVICE, "epgx_ctx_create: device wavefront size %d, need 64 (CDNA)", w);
    }
    ctx->stream = ctx->own;
    ctx->own_stream = true;
    *out = ctx;
    return EPGX_OK;
}

extern "C" int epgx_ctx_destroy(epgx_ctx *ctx) {
    if (!ctx) return EPGX_OK;
    (void)hipSetDevice(ctx->device);
    if (ctx->own) (void)hipStreamSynchronize(ctx->own);
    if (ctx->copy_stream) {
        (void)hipStreamSynchronize(ctx->copy_stream);
        (void)hipStreamDestroy(ctx->copy_stream);
    }
    for (hipEvent_t ev : ctx->slab_events) (void)hipEventDestroy(ev);
    for (auto &st : ctx->stages) {
        if (st.done) (void)hipEventDestroy(st.done);
        if (st.host) (void)hipHostFree(st.host);
    }
    for (auto &b : ctx->host_cache) (void)hipHostFree(b.first);
    dev_release_cache(ctx);
    if (ctx->ev0) (void)hipEventDestroy(ctx->ev0);
    if (ctx->ev1) (void)hipEventDestroy(ctx->ev1);
    if (ctx->own) (void)hipStreamDestroy(ctx->own);
    delete ctx;
    return EPGX_OK;
}

extern "C" int epgx_ctx_set_stream(epgx_ctx *ctx, void *hip_stream) {
    if (!ctx) return fail(EPGX_ERR_INVALID, "epgx_ctx_set_stream: ctx is NULL");
    if (int rc = set_device(ctx)) return rc;
    // recycled device blocks rely on stream order: drain the old stream before switching
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (hip_stream) {
        ctx->stream = (hipStream_t)hip_stream;
        ctx->own_stream = false;
    } else {
        ctx->stream = ctx->own;
        ctx->own_stream = true;
    }
    return EPGX_OK;
}

extern "C" int epgx_ctx_synchronize(epgx_ctx *ctx) {
    if (!ctx) return fail(EPGX_ERR_INVALID, "epgx_ctx_synchronize: ctx is NULL");
    if (int rc = set_device(ctx)) return rc;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return EPGX_OK;
}

extern "C" int epgx_ctx_info(epgx_ctx *ctx, epgx_device_info *out) {
    if (!ctx || !out) return fail(EPGX_ERR_INVALID, "epgx_ctx_info: NULL argument");
    memset(out, 0, sizeof(*out));
    snprintf(out->name, sizeof(out->name), "%s", ctx->prop.name);
    snprintf(out->arch, sizeof(out->arch), "%s", ctx->prop.gcnArchName);
    out->compute_units = ctx->prop.multiProcessorCount;
    out->wavefront_size = ctx->prop.warpSize;
    out->clock_khz = ctx->prop.clockRate;
    out->hbm_bytes = (int64_t)ctx->prop.totalGlobalMem;
    return EPGX_OK;
}

// ------------------------------------------------------------------------------ memory
extern "C" int epgx_malloc(epgx_ctx *ctx, int64_t bytes, void **dptr) {
    if (!ctx || !dptr || bytes < 0) return fail(EPGX_ERR_INVALID, "epgx_malloc: bad argument");
    *dptr = nullptr;
    if (int rc = set_device(ctx)) return rc;
    HIP_TRY(dev_alloc(ctx, dptr, (size_t)std::max<int64_t>(bytes, 16)));
    return EPGX_OK;
}

extern "C" int epgx_ctx_release_cache(epgx_ctx *ctx) {
    if (!ctx) return fail(EPGX_ERR_INVALID, "epgx_ctx_release_cache: ctx is NULL");
    if (int rc = set_device(ctx)) return rc;
    dev_release_cache(ctx);
    return EPGX_OK;
}

extern "C" int epgx_free(epgx_ctx *ctx, void *dptr) {
    if (!ctx) return fail(EPGX_ERR_INVALID, "epgx_free: ctx is NULL");
    if (!dptr) return EPGX_OK;
    if (int rc = set_device(ctx)) return rc;
    dev_free(ctx, dptr);
    return EPGX_OK;
}

extern "C" int epgx_memset(epgx_ctx *ctx, void *dptr, int value, int64_t bytes) {
    if (!ctx || (!dptr && bytes) || bytes < 0) return fail(EPGX_ERR_INVALID, "epgx_memset: bad argument");
    if (int rc = set_device(ctx)) return rc;
    if (bytes) HIP_TRY(hipMemsetAsync(dptr, value, (size_t)bytes, ctx->stream));
    return EPGX_OK;
}

extern "C" int epgx_memcpy_h2d(epgx_ctx *ctx, void *dptr, const void *host, int64_t bytes) {
    if (!ctx || bytes < 0 || (bytes && (!dptr || !host)))
        return fail(EPGX_ERR_INVALID, "epgx_memcpy_h2d: bad argument");
    if (int rc = set_device(ctx)) return rc;
    if (bytes) {
        HIP_TRY(hipMemcpyAsync(dptr, host, (size_t)bytes, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return EPGX_OK;
}

extern "C" int epgx_memcpy_d2h(epgx_ctx *ctx, void *host, const void *dptr, int64_t bytes) {
    if (!ctx || bytes < 0 || (bytes && (!dptr || !host)))
        return fail(EPGX_ERR_INVALID, "epgx_memcpy_d2h: bad argument");
    if (int rc = set_device(ctx)) return rc;
    if (bytes) {
        HIP_TRY(hipMemcpyAsync(host, dptr, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return EPGX_OK;
}

extern "C" int epgx_memcpy_d2d(epgx_ctx *ctx, void *dst, const void *src, int64_t bytes) {
    if (!ctx || bytes < 0 || (bytes && (!dst || !src)))
        return fail(EPGX_ERR_INVALID, "epgx_memcpy_d2d: bad argument");
    if (int rc = set_device(ctx)) return rc;
    if (bytes) HIP_TRY(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToDevice, ctx->stream));
    return EPGX_OK;
}

// ------------------------------------------------------------------------------ page-locked host memory
extern "C" int epgx_host_alloc(epgx_ctx *ctx, int64_t bytes, int32_t cached_only, void **hptr) {
    if (!ctx || !hptr || bytes < 0) return fail(EPGX_ERR_INVALID, "epgx_host_alloc: bad argument");
    *hptr = nullptr;
    if (int rc = set_device(ctx)) return rc;
    size_t n = std::max<size_t>(((size_t)bytes + 4095) & ~(size_t)4095, 4096);
    {
        std::lock_guard<std::mutex> guard(ctx->mem);
        int best = -1;
        for (int i = 0; i < (int)ctx->host_cache.size(); ++i) {
            const size_t have = ctx->host_cache[i].second;
            if (have >= n && have <= n + n / 4 && (best < 0 || have < ctx->host_cache[best].second)) best = i;
        }
        if (best >= 0) {
            *hptr = ctx->host_cache[best].first;
            ctx->host_live[*hptr] = ctx->host_cache[best].second;
            ctx->host_cached_bytes -= ctx->host_cache[best].second;
            ctx->host_cache.erase(ctx->host_cache.begin() + best);
            return EPGX_OK;
        }
    }
    if (cached_only) return EPGX_OK;   // nothing to recycle: the caller takes its pageable path
    HIP_TRY(hipHostMalloc(hptr, n, hipHostMallocDefault));
    std::lock_guard<std::mutex> guard(ctx->mem);
    ctx->host_live[*hptr] = n;
    return EPGX_OK;
}

extern "C" int epgx_host_free(epgx_ctx *ctx, void *hptr) {
    if (!ctx) return fail(EPGX_ERR_INVALID, "epgx_host_free: ctx is NULL");
    if (!hptr) return EPGX_OK;
    if (int rc = set_device(ctx)) return rc;
    std::vector<void *> victims;
    {
        std::lock_guard<std::mutex> guard(ctx->mem);
        auto it = ctx->host_live.find(hptr);
        if (it == ctx->host_live.end()) return fail(EPGX_ERR_INVALID, "epgx_host_free: not a block of this context");
        ctx->host_cache.emplace_back(hptr, it->second);
        ctx->host_cached_bytes += it->second;
        ctx->host_live.erase(it);
        // keep at most EPGX_PINNED_CACHE_MB (4 GiB) / 16 blocks of pinned memory around (oldest first out)
        static const size_t budget = (size_t)(getenv("EPGX_PINNED_CACHE_MB") ? std::max(0, atoi(getenv("EPGX_PINNED_CACHE_MB"))) : 4096) << 20;
        while (!ctx->host_cache.empty() && (ctx->host_cached_bytes > budget || ctx->host_cache.size() > 16)) {
            victims.push_back(ctx->host_cache.front().first);
            ctx->host_cached_bytes -= ctx->host_cache.front().second;
            ctx->host_cache.erase(ctx->host_cache.begin());
        }
    }
    for (void *v : victims) (void)hipHostFree(v);
    return EPGX_OK;
}

extern "C" int epgx_host_register(epgx_ctx *ctx, void *hptr, int64_t bytes) {
    if (!ctx || !hptr || bytes <= 0) return fail(EPGX_ERR_INVALID, "epgx_host_register: bad argument");
    if (int rc = set_device(ctx)) return rc;
    const hipError_t e = hipHostRegister(hptr, (size_t)bytes, hipHostRegisterDefault);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(e == hipErrorOutOfMemory ? EPGX_ERR_NOMEM : EPGX_ERR_HIP, "epgx_host_register: %s", hipGetErrorString(e));
    }
    return EPGX_OK;
}

extern "C" int epgx_host_unregister(epgx_ctx *ctx, void *hptr) {
    if (!ctx || !hptr) return fail(EPGX_ERR_INVALID, "epgx_host_unregister: NULL argument");
    if (int rc = set_device(ctx)) return rc;
    // (copies into the block may still be in flight on the context's streams)
    if (ctx->copy_stream) (void)hipStreamSynchronize(ctx->copy_stream);
    (void)hipStreamSynchronize(ctx->stream);
    const hipError_t e = hipHostUnregister(hptr);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        return fail(EPGX_ERR_HIP, "epgx_host_unregister: %s", hipGetErrorString(e));
    }
    return EPGX_OK;
}

// ------------------------------------------------------------------------------ timing
extern "C" int epgx_timer_start(epgx_ctx *ctx) {
    if (!ctx) return fail(EPGX_ERR_INVALID, "epgx_timer_start: ctx is NULL");
    if (int rc = set_device(ctx)) return rc;
    HIP_TRY(hipEventRecord(ctx->ev0, ctx->stream));
    return EPGX_OK;
}

extern "C" int epgx_timer_stop(epgx_ctx *ctx, float *elapsed_ms) {
    if (!ctx || !elapsed_ms) return fail(EPGX_ERR_INVALID, "epgx_timer_stop: NULL argument");
    if (int rc = set_device(ctx)) return rc;
    HIP_TRY(hipEventRecord(ctx->ev1, ctx->stream));
    HIP_TRY(hipEventSynchronize(ctx->ev1));
    HIP_TRY(hipEventElapsedTime(elapsed_ms, ctx->ev0, ctx->ev1));
    return EPGX_OK;
}

// ------------------------------------------------------------------------------ plan
static int ncoef_expected(int opcode) {
    switch (opcode) {
    case EPGX_OP_T: return 8;
    case EPGX_OP_MAT: return 10;  // 9 used, padded to 10
    case EPGX_OP_MAT0: return 14;
    case EPGX_OP_T0: return 12;
    case EPGX_OP_E: return 4;
    case EPGX_OP_PD: return 1;
    case EPGX_OP_D: case EPGX_OP_GS: return -1;  // depends on K: checked in epgx_run
    default: return 0;
    }
}

extern "C" int epgx_plan_create(epgx_ctx *ctx, const epgx_plan_desc *d, epgx_plan **out) {
    if (!ctx || !d || !out) return fail(EPGX_ERR_INVALID, "epgx_plan_create: NULL argument");
    *out = nullptr;
    // (the first member: readable whatever the caller's idea of the struct is)
    if (d->struct_size != sizeof(epgx_plan_desc))
        return fail(EPGX_ERR_INVALID, "epgx_plan_create: struct_size = %u, but epgx_plan_desc has %zu bytes in this library (ABI %d): "
                    "set struct_size = sizeof(epgx_plan_desc) and rebuild the caller against include/epgx.h", d->struct_size,
                    sizeof(epgx_plan_desc), EPGX_ABI_VERSION);
    if (d->n_ops <= 0 || !d->ops) return fail(EPGX_ERR_INVALID, "epgx_plan_create: empty operator list");
    if (d->ndim < 1 || d->ndim > EPGX_MAX_DIMS || !d->grid_shape)
        return fail(EPGX_ERR_INVALID, "epgx_plan_create: ndim %d not in [1,%d]", d->ndim, EPGX_MAX_DIMS);
    if (d->n_spaces < 0 || d->n_spaces > EPGX_MAX_SPACES || (d->n_spaces && !d->space_strides))
        return fail(EPGX_ERR_UNSUPPORTED, "epgx_plan_create: %d index spaces, at most %d supported",
                    d->n_spaces, EPGX_MAX_SPACES);
    if (d->n_coef < 0 || (d->n_coef && !d->coef)) return fail(EPGX_ERR_INVALID, "epgx_plan_create: bad coefficient pool");
    if (d->n_coef_generated < 0 || d->n_fuse < 0 || (d->n_fuse && !d->fuse) || d->n_assemble < 0 || (d->n_assemble && !d->assemble))
        return fail(EPGX_ERR_INVALID, "epgx_plan_create: bad generated-table description");
    const int64_t n_pool = d->n_coef + d->n_coef_generated;   // host part + device-generated part
    if (n_pool >= ((int64_t)1 << 29) - 16)
        return fail(EPGX_ERR_UNSUPPORTED, "epgx_plan_create: coefficient pool larger than 4 GiB (split the grid)");
    if (d->n_adc < 0) return fail(EPGX_ERR_INVALID, "epgx_plan_create: n_adc < 0");

    const bool trace = getenv("EPGX_TRACE") != nullptr;
    const auto tic = std::chrono::steady_clock::now();
    auto lap = [&](const char *what) {
        if (trace)
            fprintf(stderr, "[epgx] plan_create %-12s +%.3f ms\n", what,
                    std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tic).count());
    };
    epgx_plan *pl = new (std::nothrow) epgx_plan();
    if (!pl) return fail(EPGX_ERR_NOMEM, "epgx_plan_create: host allocation failed");
    pl->ctx = ctx;
    pl->ndim = d->ndim;
    pl->n_spaces = d->n_spaces;
    pl->n_adc = d->n_adc;
    pl->n_coef = d->n_coef;
    int64_t nvox = 1;
    for (int i = 0; i < EPGX_MAX_DIMS; ++i) pl->shape[i] = 1;
    for (int i = 0; i < d->ndim; ++i) {
        if (d->grid_shape[i] < 1) {
            delete pl;
            return fail(EPGX_ERR_INVALID, "epgx_plan_create: grid_shape[%d] < 1", i);
        }
        pl->shape[i] = d->grid_shape[i];
        nvox *= d->grid_shape[i];
        if (nvox > (int64_t)1 << 40) {
            delete pl;
            return fail(EPGX_ERR_UNSUPPORTED, "epgx_plan_create: grid too large");
        }
    }
    pl->nvox_total = nvox;
    // largest table index each space can produce
    int64_t space_extent[EPGX_MAX_SPACES];
    memset(pl->strides, 0, sizeof(pl->strides));
    for (int s = 0; s < d->n_spaces; ++s) {
        int64_t ext = 0;
        for (int i = 0; i < d->ndim; ++i) {
            int64_t st = d->space_strides[(size_t)s * EPGX_MAX_DIMS + i];
            if (st < 0) {
                delete pl;
                return fail(EPGX_ERR_INVALID, "epgx_plan_create: negative stride in space %d", s);
            }
            pl->strides[s][i] = st;
            ext += (pl->shape[i] - 1) * st;
        }
        if (ext >= ((int64_t)1 << 31)) {
            delete pl;
            return fail(EPGX_ERR_UNSUPPORTED, "epgx_plan_create: table of space %d too large", s);
        }
        space_extent[s] = ext;
        bool dense = true;  // C-order strides of the full grid (axes of extent 1 are irrelevant)
        int64_t acc = 1;
        for (int i = d->ndim - 1; i >= 0; --i) {
            if (pl->shape[i] > 1 && pl->strides[s][i] != acc) dense = false;
            acc *= pl->shape[i];
        }
        if (dense) pl->dense_spaces |= 1u << s;
    }
    // device-assembled tables (epgx_assemble): checked first, operators and fuse recipes may refer to them
    struct Assembled { int32_t ncoef, space; bool im_zero; };   // im_zero: column 1 of a 4-column (relaxation) table is all zero
    std::map<int64_t, Assembled> assembled;
    for (int i = 0; i < d->n_assemble; ++i) {
        const epgx_assemble &as = d->assemble[i];
        const char *why = nullptr;
        if (as.dst_space < -1 || as.dst_space >= d->n_spaces) why = "index space out of range";
        if (!why && (as.ncoef < 1 || as.ncoef > EPGX_MAX_ASM_COLS || as.n_src < 1 || as.n_src > EPGX_MAX_ASM_SRC)) why = "bad column / source count";
        const int64_t entries = why ? 0 : (as.dst_space < 0 ? 0 : space_extent[as.dst_space]) + 1;
        if (!why && (as.dst_off < d->n_coef || as.dst_off + entries * as.ncoef > n_pool)) why = "destination outside the generated part of the pool";
        int64_t src_entries[EPGX_MAX_ASM_SRC] = {0, 0, 0, 0};
        for (int k = 0; k < as.n_src && !why; ++k) {
            const epgx_asm_src &sr = as.src[k];
            int64_t last = 0;
            for (int dd = 0; dd < d->ndim && !why; ++dd) {
                if (sr.strides[dd] < 0) why = "negative source stride";
                const int64_t ds = as.dst_space < 0 ? 0 : pl->strides[as.dst_space][dd];
                if (!why && pl->shape[dd] > 1 && sr.strides[dd] != 0 && ds == 0) why = "a source varies along an axis the destination does not";
                last += (pl->shape[dd] - 1) * sr.strides[dd];
            }
            if (!why && (sr.ncol < 1 || sr.off < 0 || sr.off + (last + 1) * sr.ncol > d->n_coef)) why = "source columns outside the host part of the pool";
            src_entries[k] = last + 1;
        }
        for (int c = 0; c < as.ncoef && !why; ++c)
            if (as.col_src[c] >= as.n_src || as.col_idx[c] >= as.src[as.col_src[c]].ncol) why = "column refers to a missing source column";
        if (why) {
            delete pl;
            return fail(EPGX_ERR_INVALID, "epgx_plan_create: assembled table %d: %s", i, why);
        }
        bool im_zero = false;
        if (as.ncoef == 4) {   // relaxation layout: is Im e0 (column 1) zero in every entry?
            const epgx_asm_src &sr = as.src[as.col_src[1]];
            im_zero = true;
            for (int64_t j = 0; j < src_entries[as.col_src[1]] && im_zero; ++j) im_zero = d->coef[sr.off + j * sr.ncol + as.col_idx[1]] == 0.0;
        }
        assembled[as.dst_off] = {as.ncoef, as.dst_space, im_zero};
    }
    auto is_assembled = [&](int64_t off, int ncoef, int space) {
        const auto hit = assembled.find(off);
        return hit != assembled.end() && hit->second.ncoef == ncoef && hit->second.space == space;
    };
    pl->ops.assign(d->ops, d->ops + d->n_ops);
    for (int i = 0; i < d->n_ops; ++i) {
        const epgx_op &op = pl->ops[i];
        const char *why = nullptr;
        if (op.opcode < 0 || op.opcode >= EPGX_OP__COUNT) why = "unknown opcode";
        int need = why ? 0 : ncoef_expected(op.opcode);
        if (!why && need >= 0 && op.ncoef != need) why = "wrong ncoef for opcode";
        if (!why && need < 0 && op.ncoef <= 0) why = "table-driven operator without a table";
        if (!why && need) {
            if (op.space < -1 || op.space >= d->n_spaces) why = "index space out of range";
            int64_t last = op.space < 0 ? 0 : space_extent[op.space];
            if (!why && (op.coef_off < 0 || op.coef_off + (last + 1) * (int64_t)op.ncoef > n_pool))
                why = "coefficient table exceeds the pool";
        }
        if (!why && need && op.opcode != EPGX_OP_T0 && !is_assembled(op.coef_off, op.ncoef, op.space) &&
            op.coef_off + (op.space < 0 ? 1 : space_extent[op.space] + 1) * (int64_t)op.ncoef > d->n_coef)
            why = "a table in the generated part of the pool needs a recipe (epgx_assemble, or epgx_fuse for EPGX_OP_T0)";
        if (!why && op.opcode == EPGX_OP_S) {
            if (op.ia == 0) why = "shift by 0";
            if (op.ia >= EPGX_MAX_K || op.ia <= -EPGX_MAX_K) why = "shift exceeds EPGX_MAX_K";
            if (op.ib < 0) why = "negative truncation order";
        }
        if (!why && op.opcode == EPGX_OP_ADC) {
            if (op.ia < 0 || op.ia >= d->n_adc) why = "ADC slot out of range";
            if (op.ib != 0 && op.ib != 1) why = "unknown ADC probe";
        }
        if (why) {
            const int opcode = op.opcode;   // `op` lives in the plan that is about to go
            delete pl;
            return fail(EPGX_ERR_INVALID, "epgx_plan_create: operator %d (opcode %d): %s", i, opcode, why);
        }
    }
    if (d->n_vars < 0 || d->n_vars > EPGX_MAX_VARS || (d->n_vars > 0 && !d->dops)) {
        delete pl;
        return fail(EPGX_ERR_INVALID, "epgx_plan_create: n_vars=%d (at most %d) or dops missing", d->n_vars, EPGX_MAX_VARS);
    }
    pl->n_vars = d->n_vars;
    pl->deriv_flags = d->deriv_flags;
    {
        const int env = knobs().fold ? 1 : 0;   // (EPGX_FOLD=0: measurements)
        pl->fold = env != 0 && !(d->deriv_flags & EPGX_PLAN_NO_FOLD) && d->n_vars == 0;
    }
    if (d->n_vars > 0) {
        pl->dops.assign(d->dops, d->dops + d->n_ops);
        for (int i = 0; i < d->n_ops; ++i) {
            const epgx_op &op = pl->ops[i];
            for (int v = 0; v < EPGX_MAX_VARS; ++v) {
                const int64_t off = pl->dops[i].coef_off[v];
                if (off < 0) continue;
                const char *why = nullptr;
                int nc = 0;
                if (v >= d->n_vars) why = "partial for a variable beyond n_vars";
                else if (op.opcode == EPGX_OP_T || op.opcode == EPGX_OP_MAT || op.opcode == EPGX_OP_MAT0 || op.opcode == EPGX_OP_T0) nc = 10;
                else if (op.opcode == EPGX_OP_E) nc = 4;
                else why = "only T / MAT / E operators can carry partial derivatives";
                const int sp = pl->dops[i].space[v];
                if (!why && (sp < -1 || sp >= d->n_spaces)) why = "index space of a partial out of range";
                if (!why) {
                    const int64_t last = sp < 0 ? 0 : space_extent[sp];
                    if (op.opcode == EPGX_OP_T0 && off >= d->n_coef) {   // generated next to the table itself (epgx_fuse_partial)
                        if (off + (last + 1) * 14 > n_pool) why = "generated partial table exceeds the pool";
                    } else if (is_assembled(off, nc, sp)) {              // assembled on the device from per-axis columns (epgx_assemble)
                    } else if (off + (last + 1) * nc > d->n_coef) why = "partial table exceeds the host part of the pool";
                }
                if (why) {
                    delete pl;
                    return fail(EPGX_ERR_INVALID, "epgx_plan_create: operator %d, variable %d: %s", i, v, why);
                }
            }
            if (op.opcode == EPGX_OP_ADC && op.ia + d->n_vars >= d->n_adc) {
                delete pl;
                return fail(EPGX_ERR_INVALID, "epgx_plan_create: operator %d: ADC rows [%d,%d] exceed n_adc=%d", i,
                            op.ia, op.ia + d->n_vars, d->n_adc);
            }
        }
    }
    // exact-zero patterns that let the kernel drop products without changing a single bit:
    // T(alpha, 0): Im m01 = Re m02 = Re m20 = 0;  E with g = 0: Im e0 = 0
    // device-generated tables: check the references, derive their zero pattern from the sources
    std::map<int64_t, uint8_t> generated_pattern;   // dst_off -> 1 if the phi = 0 pattern holds
    // Zero patterns of rotation tables (they select shorter fma chains in the kernels):
    //   1 "TX": Im m01 = Re m02 = Re m20 (= Re o0) = 0 -- rotations about x, phi = 0 or 180 deg
    //   3 "TY": Im m01 = Im m02 = Im m20 (= Im o0) = 0 -- real matrices, phi = +-90 deg
    // The reference computes e^{i phi} with phi in radians, so only phi = 0 gives exact zeros:
    // cos(pi/2) = 6e-17, sin(pi) = 1.2e-16.  A component below 2^-48 (3.6e-15) of its own entry's
    // modulus in EVERY entry of the table counts as the rounding residue it is and is cleared in the
    // device copy of the table (snap list below), so that every kernel sees exact zeros: a deviation
    // of <= 4e-15 relative from the reference's arithmetic, against a parity bar of 1e-6.
    struct Snap { int64_t off; int64_t entries; int nc; uint32_t mask; };
    std::vector<Snap> snaps;
    auto t_pattern = [&](int64_t off, int space, int nc) -> uint8_t {
        const int64_t entries = (space < 0 ? 0 : space_extent[space]) + 1;
        const double *tab = d->coef + off;
        const double tol = 1.0 / 281474976710656.0;   // 2^-48
        bool tx = true, ty = true, exact_x = true, exact_y = true;
        // is `v` (the other component of the same complex entry: `w`) zero / a rounding residue?  (hypot only when needed)
        auto residue = [&](double v, double w, bool &exact) {
            if (v == 0.0) return true;
            exact = false;
            return std::fabs(v) <= tol * std::hypot(v, w);
        };
        for (int64_t j = 0; j < entries && (tx || ty); ++j) {
            const double *c = tab + j * nc;
            bool ep = true;
            const bool im_p = residue(c[2], c[1], ep);
            exact_x = exact_x && ep;
            exact_y = exact_y && ep;
            tx = tx && im_p && residue(c[3], c[4], exact_x) && residue(c[5], c[6], exact_x) && (nc == 8 || residue(c[8], c[9], exact_x));
            ty = ty && im_p && residue(c[4], c[3], exact_y) && residue(c[6], c[5], exact_y) && (nc == 8 || residue(c[9], c[8], exact_y));
        }
        if (tx) {
            if (!exact_x) snaps.push_back({off, entries, nc, (1u << 2) | (1u << 3) | (1u << 5) | (nc == 12 ? 1u << 8 : 0u)});
            return 1;
        }
        if (ty) {
            if (!exact_y) snaps.push_back({off, entries, nc, (1u << 2) | (1u << 4) | (1u << 6) | (nc == 12 ? 1u << 9 : 0u)});
            return 3;
        }
        return 0;
    };
    // relaxation table without a precession term (Im e0 == 0 in every entry)?  Scanned once per table; a table over a
    // 1024 x 1024 grid is 34 MB, so large tables are split over a few threads
    std::map<std::pair<int64_t, int32_t>, bool> e_real_known;
    auto e_is_real = [&](int64_t off, int space) {
        if (off >= d->n_coef) {   // assembled on the device: known from its column
            const auto as = assembled.find(off);
            return as != assembled.end() && as->second.ncoef == 4 && as->second.im_zero;
        }
        const auto key = std::make_pair(off, (int32_t)(space + 1));
        const auto hit = e_real_known.find(key);
        if (hit != e_real_known.end()) return hit->second;
        const int64_t entries = (space < 0 ? 0 : space_extent[space]) + 1;
        const double *tab = d->coef + off;
        std::atomic<bool> real{true};
        auto scan = [&](int64_t j0, int64_t j1) {
            for (int64_t j = j0; j < j1; ++j)
                if (tab[j * 4 + 1] != 0.0) {
                    real.store(false, std::memory_order_relaxed);
                    return;
                }
        };
        const int nthreads = entries >= (1 << 17) ? 4 : 1;
        if (nthreads == 1) scan(0, entries);
        else {
            std::vector<std::thread> pool;
            for (int t = 0; t < nthreads; ++t) pool.emplace_back(scan, entries * t / nthreads, entries * (t + 1) / nthreads);
            for (auto &th : pool) th.join();
        }
        e_real_known[key] = real.load();
        return real.load();
    };
    std::map<std::pair<int64_t, int32_t>, uint8_t> scanned;  // a table referenced by many operators / recipes is scanned once
    auto t_pattern_once = [&](int64_t off, int space, int nc) -> uint8_t {
        const auto key = std::make_pair(off, (int32_t)(nc * 8 + space + 1));
        const auto hit = scanned.find(key);
        if (hit != scanned.end()) return hit->second;
        const uint8_t pat = t_pattern(off, space, nc);
        scanned[key] = pat;
        return pat;
    };
    for (int i = 0; i < d->n_fuse; ++i) {
        const epgx_fuse &fu = d->fuse[i];
        const char *why = nullptr;
        auto space_ok = [&](int sp) { return sp >= -1 && sp < d->n_spaces; };
        auto ext = [&](int sp) { return (sp < 0 ? 0 : space_extent[sp]) + 1; };
        if (!space_ok(fu.dst_space) || !space_ok(fu.src_space) || !space_ok(fu.e_space)) why = "index space out of range";
        if (!why && fu.src_ncoef != 8 && fu.src_ncoef != 12) why = "source must have 8 or 12 coefficients";
        if (!why && (fu.dst_off < d->n_coef || fu.dst_off + ext(fu.dst_space) * 12 > n_pool)) why = "destination outside the generated part of the pool";
        if (!why && !is_assembled(fu.e_off, 4, fu.e_space) && (fu.e_off < 0 || fu.e_off + ext(fu.e_space) * 4 > d->n_coef))
            why = "E source neither in the host part of the pool nor an assembled table";
        if (!why && (fu.src_off < 0 || fu.src_off + ext(fu.src_space) * fu.src_ncoef > n_pool)) why = "rotation source outside the pool";
        if (!why && fu.src_off >= d->n_coef && (fu.src_ncoef != 12 || !generated_pattern.count(fu.src_off)))
            why = "a generated source must be the destination of an earlier entry";
        if (!why)
            for (int dd = 0; dd < d->ndim && !why; ++dd) {
                const int64_t ds = fu.dst_space < 0 ? 0 : pl->strides[fu.dst_space][dd];
                const int64_t ss = fu.src_space < 0 ? 0 : pl->strides[fu.src_space][dd];
                const int64_t es = fu.e_space < 0 ? 0 : pl->strides[fu.e_space][dd];
                if (pl->shape[dd] > 1 && ds == 0 && (ss != 0 || es != 0)) why = "a source varies along an axis the destination does not";
            }
        if (!why && !e_is_real(fu.e_off, fu.e_space)) why = "E source has a precession term (Im e0 != 0)";
        if (why) {
            delete pl;
            return fail(EPGX_ERR_INVALID, "epgx_plan_create: generated table %d: %s", i, why);
        }
        generated_pattern[fu.dst_off] = fu.src_off >= d->n_coef ? generated_pattern[fu.src_off]
                                                                : t_pattern_once(fu.src_off, fu.src_space, fu.src_ncoef);
    }
    // zero pattern of a partial-derivative table in the host part of the pool (scanned once per table):
    //   relaxation (4 per entry):  1 = Im d e0 = 0 in every entry
    //   rotation (10 per entry, ur ui pr pi qr qi tr ti c22): the partial of a rotation about x (phi = 0) or y (phi = +-90) has
    //   the zero pattern of its operator: 1 = Im m00 = Im m01 = Re m02 = Re m20 = 0, 2 = all Im = 0 -- exact, or rounding
    //   residues of e^{i phi} (below 2^-48 of their entry's modulus in every entry), which are then cleared in the device
    //   copy like those of the operator tables (snap_kernel)
    std::map<std::pair<int64_t, int32_t>, uint8_t> dscanned;
    auto d_pattern_once = [&](int64_t off, int sp, bool is_e) -> uint8_t {
        const auto key = std::make_pair(off, (int32_t)((is_e ? 8 : 0) + sp + 1));
        const auto hit = dscanned.find(key);
        if (hit != dscanned.end()) return hit->second;
        if (off >= d->n_coef) {   // assembled on the device: a relaxation partial's pattern is known from its column, a rotation's is general
            const auto as = assembled.find(off);
            const uint8_t pat_as = (is_e && as != assembled.end() && as->second.ncoef == 4 && as->second.im_zero) ? 1 : 0;
            dscanned[key] = pat_as;
            return pat_as;
        }
        const int64_t entries = (sp < 0 ? 0 : space_extent[sp]) + 1;
        const int nc = is_e ? 4 : 10;
        const double *tab = d->coef + off;
        uint8_t pat = 0;
        if (is_e) {
            bool zero = true;
            for (int64_t j = 0; j < entries && zero; ++j) zero = tab[j * nc + 1] == 0.0;
            pat = zero ? 1 : 0;
        } else {
            const double tol = 1.0 / 281474976710656.0;
            bool tx = true, ty = true, exact_x = true, exact_y = true;
            auto residue = [&](double x, double w, bool &exact) {
                if (x == 0.0) return true;
                exact = false;
                return std::fabs(x) <= tol * std::hypot(x, w);
            };
            for (int64_t j = 0; j < entries && (tx || ty); ++j) {
                const double *c = tab + j * nc;
                bool e0 = true;
                const bool diag = residue(c[1], c[0], e0) && residue(c[3], c[2], e0);   // Im m00, Im m01
                exact_x = exact_x && e0;
                exact_y = exact_y && e0;
                tx = tx && diag && residue(c[4], c[5], exact_x) && residue(c[6], c[7], exact_x);
                ty = ty && diag && residue(c[5], c[4], exact_y) && residue(c[7], c[6], exact_y);
            }
            if (tx) {
                pat = 1;
                if (!exact_x) snaps.push_back({off, entries, nc, (1u << 1) | (1u << 3) | (1u << 4) | (1u << 6)});
            } else if (ty) {
                pat = 2;
                if (!exact_y) snaps.push_back({off, entries, nc, (1u << 1) | (1u << 3) | (1u << 5) | (1u << 7)});
            }
        }
        dscanned[key] = pat;
        return pat;
    };
    // partials of generated tables (epgx_fuse_partial): references checked, zero pattern derived from the sources
    std::map<int64_t, uint8_t> generated_dpattern;   // dst_off -> 1 (phi = 0 pattern) / 2 (real matrix) / 0
    if (d->n_fuse_partial < 0 || (d->n_fuse_partial > 0 && (!d->fuse_partial || d->n_vars == 0))) {
        delete pl;
        return fail(EPGX_ERR_INVALID, "epgx_plan_create: n_fuse_partial=%d without a list or without variables", d->n_fuse_partial);
    }
    for (int i = 0; i < d->n_fuse_partial; ++i) {
        const epgx_fuse_partial &fp = d->fuse_partial[i];
        const char *why = nullptr;
        auto space_ok = [&](int sp) { return sp >= -1 && sp < d->n_spaces; };
        auto ext = [&](int sp) { return (sp < 0 ? 0 : space_extent[sp]) + 1; };
        const bool has_dt = fp.dsrc_off >= 0, has_de = fp.de_off >= 0;
        if (!space_ok(fp.dst_space) || !space_ok(fp.src_space) || !space_ok(fp.e_space) || (has_dt && !space_ok(fp.dsrc_space)) ||
            (has_de && !space_ok(fp.de_space)))
            why = "index space out of range";
        if (!why && !has_dt && !has_de) why = "neither the rotation nor the relaxation has a partial";
        if (!why && fp.src_ncoef != 8 && fp.src_ncoef != 12) why = "rotation source must have 8 or 12 coefficients";
        if (!why && has_dt && fp.dsrc_ncoef != 10 && fp.dsrc_ncoef != 14) why = "rotation partial must have 10 or 14 coefficients";
        if (!why && (fp.dst_off < d->n_coef || fp.dst_off + ext(fp.dst_space) * 14 > n_pool)) why = "destination outside the generated part of the pool";
        if (!why && (fp.src_off < 0 || fp.src_off + ext(fp.src_space) * fp.src_ncoef > n_pool)) why = "rotation source outside the pool";
        if (!why && fp.src_off >= d->n_coef && (fp.src_ncoef != 12 || !generated_pattern.count(fp.src_off)))
            why = "a generated rotation source must be the destination of an entry of `fuse`";
        if (!why && !is_assembled(fp.e_off, 4, fp.e_space) && (fp.e_off < 0 || fp.e_off + ext(fp.e_space) * 4 > d->n_coef))
            why = "E source neither in the host part of the pool nor an assembled table";
        if (!why && !e_is_real(fp.e_off, fp.e_space)) why = "E source has a precession term (Im e0 != 0)";
        if (!why && has_dt) {
            if (fp.dsrc_ncoef == 10 ? fp.dsrc_off + ext(fp.dsrc_space) * 10 > d->n_coef
                                    : (fp.dsrc_off < d->n_coef || !generated_dpattern.count(fp.dsrc_off) || fp.dsrc_off + ext(fp.dsrc_space) * 14 > n_pool))
                why = "rotation partial: 10 per entry in the host part of the pool, or 14 per entry written by an earlier entry";
        }
        if (!why && has_de && !is_assembled(fp.de_off, 4, fp.de_space) && fp.de_off + ext(fp.de_space) * 4 > d->n_coef)
            why = "E partial outside the host part of the pool (and not an assembled table)";
        if (!why && has_de && d_pattern_once(fp.de_off, fp.de_space, true) != 1) why = "E partial has a precession term (Im d e0 != 0)";
        if (!why)
            for (int dd = 0; dd < d->ndim && !why; ++dd) {
                auto str = [&](int sp) { return sp < 0 ? (int64_t)0 : pl->strides[sp][dd]; };
                if (pl->shape[dd] > 1 && str(fp.dst_space) == 0 &&
                    (str(fp.src_space) != 0 || str(fp.e_space) != 0 || (has_dt && str(fp.dsrc_space) != 0) || (has_de && str(fp.de_space) != 0)))
                    why = "a source varies along an axis the destination does not";
            }
        if (why) {
            delete pl;
            return fail(EPGX_ERR_INVALID, "epgx_plan_create: generated partial %d: %s", i, why);
        }
        const uint8_t vp = fp.src_off >= d->n_coef ? generated_pattern[fp.src_off] : t_pattern_once(fp.src_off, fp.src_space, fp.src_ncoef);
        const uint8_t dp = !has_dt ? (uint8_t)255 : (fp.dsrc_ncoef == 14 ? generated_dpattern[fp.dsrc_off] : d_pattern_once(fp.dsrc_off, fp.dsrc_space, false));
        generated_dpattern[fp.dst_off] = (vp == 1 && (dp == 1 || dp == 255)) ? 1 : ((vp == 3 && (dp == 2 || dp == 255)) ? 2 : 0);
    }
    lap("validated");
    pl->zero_pattern.assign((size_t)d->n_ops, 0);
    for (int i = 0; i < d->n_ops; ++i) {
        const epgx_op &op = pl->ops[i];
        if (op.opcode != EPGX_OP_T && op.opcode != EPGX_OP_T0 && op.opcode != EPGX_OP_E) continue;
        if (op.opcode == EPGX_OP_E) {
            pl->zero_pattern[i] = e_is_real(op.coef_off, op.space) ? 2 : 0;
            continue;
        }
        if (op.coef_off >= d->n_coef && op.opcode != EPGX_OP_T0 && is_assembled(op.coef_off, op.ncoef, op.space)) {
            pl->zero_pattern[i] = 0;   // an assembled rotation table: general chains (no host copy to scan)
            continue;
        }
        if (op.coef_off >= d->n_coef) {   // generated on the device: pattern known from its sources
            const auto g = generated_pattern.find(op.coef_off);
            if (g == generated_pattern.end() || op.opcode != EPGX_OP_T0) {
                delete pl;
                return fail(EPGX_ERR_INVALID, "epgx_plan_create: operator %d refers to the generated part of the pool but no entry of `fuse` writes there", i);
            }
            pl->zero_pattern[i] = g->second;
            continue;
        }
        pl->zero_pattern[i] = t_pattern_once(op.coef_off, op.space, op.ncoef);
    }
    if (d->n_vars > 0) {
        pl->dpattern.assign((size_t)d->n_ops, 0);
        for (int i = 0; i < d->n_ops; ++i)
            for (int v = 0; v < d->n_vars; ++v) {
                const int64_t off = pl->dops[i].coef_off[v];
                if (off < 0) continue;
                if (off >= d->n_coef && pl->ops[i].opcode == EPGX_OP_T0) {   // a generated partial (checked above): pattern known from its sources
                    const auto g = generated_dpattern.find(off);
                    if (g == generated_dpattern.end()) {
                        delete pl;
                        return fail(EPGX_ERR_INVALID, "epgx_plan_create: operator %d, variable %d: the partial refers to the generated part of the pool but no entry of `fuse_partial` writes there", i, v);
                    }
                    pl->dpattern[i] |= (uint16_t)(((g->second & 3u) << (2 * v)) | (256u << v));
                    continue;
                }
                pl->dpattern[i] |= (uint16_t)((d_pattern_once(off, pl->dops[i].space[v], pl->ops[i].opcode == EPGX_OP_E) & 3u) << (2 * v));
            }
    }
    lap("zero scan");
    pl->gather_tables.resize((size_t)d->n_ops);
    for (int i = 0; i < d->n_ops; ++i) {
        const epgx_op &op = pl->ops[i];
        if (op.opcode != EPGX_OP_GS) continue;
        if (op.space >= 0) {
            delete pl;
            return fail(EPGX_ERR_UNSUPPORTED, "epgx_plan_create: operator %d: gather shifts must be the same for all voxels", i);
        }
        const int32_t *src = (const int32_t *)(d->coef + op.coef_off);
        pl->gather_tables[i].assign(src, src + 2 * (size_t)op.ncoef);
    }
    int rc = set_device(ctx);
    if (rc) { delete pl; return rc; }
    lap("host done");
    hipError_t e = hipSuccess;
    // pool padded so that the fixed-width scalar loads of the last entry stay in bounds
    pl->n_pool = n_pool;
    // derivative plans that may fold (drun_kernel): a table of logarithmic partials per (real relaxation table, real partial
    // table over the same index space)
    struct LogJob { int64_t e_off, de_off, entries; };
    std::vector<LogJob> log_jobs;
    {
        const int env = knobs().fold ? 1 : 0;
        if (d->n_vars > 0 && env != 0 && !(d->deriv_flags & EPGX_PLAN_NO_FOLD)) {
            pl->log_of.assign((size_t)d->n_ops * EPGX_MAX_VARS, -1);
            std::map<std::pair<int64_t, int64_t>, int32_t> seen;
            for (int i = 0; i < d->n_ops; ++i) {
                const epgx_op &op = pl->ops[i];
                if (op.opcode != EPGX_OP_E || op.ncoef != 4 || pl->zero_pattern[i] != 2) continue;
                for (int v = 0; v < d->n_vars; ++v) {
                    const int64_t doff = pl->dops[i].coef_off[v];
                    if (doff < 0 || ((pl->dpattern[i] >> (2 * v)) & 3u) != 1u || pl->dops[i].space[v] != op.space) continue;
                    const auto key = std::make_pair((int64_t)op.coef_off, doff);
                    auto it = seen.find(key);
                    if (it == seen.end()) {
                        epgx_plan::LogTab lt;
                        lt.off = n_pool + 32 + pl->n_log;
                        lt.space = op.space;
                        const int64_t entries = (op.space < 0 ? 0 : space_extent[op.space]) + 1;
                        pl->n_log += 2 * entries;
                        log_jobs.push_back({op.coef_off, doff, entries});
                        it = seen.emplace(key, (int32_t)pl->logtabs.size()).first;
                        pl->logtabs.push_back(lt);
                    }
                    pl->log_of[(size_t)i * EPGX_MAX_VARS + v] = it->second;
                }
            }
            // fused echoes: walk the epgx_fuse_partial chain behind every generated partial of an EPGX_OP_T0 operator
            if (d->n_fuse_partial > 0) {
                std::map<int64_t, int> recipe_of;
                for (int k = 0; k < d->n_fuse_partial; ++k) recipe_of[d->fuse_partial[k].dst_off] = k;
                pl->t0_log.assign((size_t)d->n_ops * EPGX_MAX_VARS * 2, -1);
                pl->t0_logd.assign((size_t)d->n_ops * EPGX_MAX_VARS, 0);
                std::map<std::pair<int64_t, int>, std::pair<int32_t, int32_t>> walked;   // (partial offset) -> (a, b), -2: not logarithmic
                for (int i = 0; i < d->n_ops; ++i) {
                    if (pl->ops[i].opcode != EPGX_OP_T0) continue;
                    for (int v = 0; v < d->n_vars; ++v) {
                        const int64_t doff = pl->dops[i].coef_off[v];
                        if (doff < d->n_coef) continue;   // (no partial, or not a generated one)
                        const auto wkey = std::make_pair(doff, 0);
                        auto w = walked.find(wkey);
                        if (w == walked.end()) {
                            int32_t side[2] = {-1, -1};
                            bool ok = true;
                            int64_t cur = doff;
                            for (int depth = 0; ok && depth < 8; ++depth) {
                                const auto r = recipe_of.find(cur);
                                if (r == recipe_of.end()) { ok = false; break; }
                                const epgx_fuse_partial &fp = d->fuse_partial[r->second];
                                if (fp.de_off >= 0) {
                                    const int which = fp.after ? 0 : 1;
                                    if (side[which] != -1 || fp.de_space != fp.e_space) { ok = false; break; }
                                    const auto key = std::make_pair((int64_t)fp.e_off, (int64_t)fp.de_off);
                                    auto it = seen.find(key);
                                    if (it == seen.end()) {
                                        epgx_plan::LogTab lt;
                                        lt.off = n_pool + 32 + pl->n_log;
                                        lt.space = fp.e_space;
                                        const int64_t entries = (fp.e_space < 0 ? 0 : space_extent[fp.e_space]) + 1;
                                        pl->n_log += 2 * entries;
                                        log_jobs.push_back({fp.e_off, fp.de_off, entries});
                                        it = seen.emplace(key, (int32_t)pl->logtabs.size()).first;
                                        pl->logtabs.push_back(lt);
                                    }
                                    side[which] = it->second;
                                }
                                if (fp.dsrc_off < 0) break;                 // the rotation itself has no partial: relaxations alone
                                if (fp.dsrc_off < d->n_coef) { ok = false; break; }   // a rotation partial of the host: not this route
                                cur = fp.dsrc_off;
                                if (depth == 7) ok = false;
                            }
                            w = walked.emplace(wkey, ok ? std::make_pair(side[0], side[1]) : std::make_pair(-2, -2)).first;
                        }
                        if (w->second.first == -2) continue;
                        pl->t0_log[((size_t)i * EPGX_MAX_VARS + v) * 2] = w->second.first;
                        pl->t0_log[((size_t)i * EPGX_MAX_VARS + v) * 2 + 1] = w->second.second;
                        pl->t0_logd[(size_t)i * EPGX_MAX_VARS + v] = 1;
                    }
                }
            }
        }
    }
    e = dev_alloc(ctx, (void **)&pl->d_coef, sizeof(double) * (size_t)(n_pool + 32 + pl->n_log + (pl->n_log ? 32 : 0)));
    if (e == hipSuccess && pl->n_log)
        e = hipMemsetAsync(pl->d_coef + n_pool + 32 + pl->n_log, 0, sizeof(double) * 32, ctx->stream);
    if (e == hipSuccess)   // (the generated part is written entry by entry: only the padding needs zeros)
        e = hipMemsetAsync(pl->d_coef + n_pool, 0, sizeof(double) * 32, ctx->stream);
    static const double identity_relaxation[4] = {1.0, 0.0, 1.0, 0.0};   // e, Im e, e2, r: what a missing E_a / E_b of a folded record reads
    if (e == hipSuccess)
        e = hipMemcpyAsync(pl->d_coef + n_pool, identity_relaxation, sizeof(identity_relaxation), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess && d->n_coef)
        e = hipMemcpyAsync(pl->d_coef, d->coef, sizeof(double) * (size_t)d->n_coef,
                           hipMemcpyHostToDevice, ctx->stream);
    for (size_t i = 0; i < snaps.size() && e == hipSuccess; ++i) {   // clear the rounding residues (t_pattern)
        const Snap &sn = snaps[i];
        hipLaunchKernelGGL(snap_kernel, dim3((unsigned)((sn.entries + 255) / 256)), dim3(256), 0, ctx->stream,
                           pl->d_coef + sn.off, sn.entries, sn.nc, sn.mask);
        e = hipGetLastError();
    }
    if (d->n_assemble > 0 && e == hipSuccess) {   // tables assembled from per-axis columns: one launch for all recipes
        std::vector<AsmArgs> recipes((size_t)d->n_assemble);
        int64_t most = 1;
        for (int i = 0; i < d->n_assemble; ++i) {
            const epgx_assemble &as = d->assemble[i];
            AsmArgs &aa = recipes[(size_t)i];
            memset(&aa, 0, sizeof(aa));
            aa.dst_off = as.dst_off;
            aa.n_entries = (as.dst_space < 0 ? 0 : space_extent[as.dst_space]) + 1;
            aa.ndim = d->ndim;
            aa.ncoef = as.ncoef;
            aa.n_src = as.n_src;
            for (int dd = 0; dd < d->ndim; ++dd) {
                aa.shape[dd] = pl->shape[dd];
                aa.dst_str[dd] = as.dst_space < 0 ? 0 : pl->strides[as.dst_space][dd];
            }
            for (int k = 0; k < as.n_src; ++k) {
                aa.src_off[k] = as.src[k].off;
                aa.src_ncol[k] = as.src[k].ncol;
                for (int dd = 0; dd < d->ndim; ++dd) aa.src_str[k][dd] = as.src[k].strides[dd];
            }
            memcpy(aa.col_src, as.col_src, sizeof(aa.col_src));
            memcpy(aa.col_idx, as.col_idx, sizeof(aa.col_idx));
            most = std::max(most, aa.n_entries);
        }
        AsmArgs *d_recipes = nullptr;
        e = dev_alloc(ctx, (void **)&d_recipes, sizeof(AsmArgs) * recipes.size());
        if (e == hipSuccess)
            e = hipMemcpyAsync(d_recipes, recipes.data(), sizeof(AsmArgs) * recipes.size(), hipMemcpyHostToDevice, ctx->stream);
        const unsigned bx = (unsigned)std::min<int64_t>((most + 255) / 256, 4096);
        for (int first = 0; first < d->n_assemble && e == hipSuccess; first += 32768) {
            const unsigned by = (unsigned)std::min(d->n_assemble - first, 32768);
            hipLaunchKernelGGL(assemble_kernel, dim3(bx, by), dim3(256), 0, ctx->stream, pl->d_coef, (const AsmArgs *)(d_recipes + first));
            e = hipGetLastError();
        }
        // (`recipes` is a local: the copy must have left the host before it goes; the plan's final synchronise is below,
        // so wait here only for the upload)
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        dev_free(ctx, d_recipes);
    }
    for (int i = 0; i < d->n_fuse && e == hipSuccess; ++i) {
        const epgx_fuse &fu = d->fuse[i];
        FuseArgs fa;
        memset(&fa, 0, sizeof(fa));
        fa.pool = pl->d_coef;
        fa.dst_off = fu.dst_off;
        fa.src_off = fu.src_off;
        fa.e_off = fu.e_off;
        fa.n_entries = (fu.dst_space < 0 ? 0 : space_extent[fu.dst_space]) + 1;
        fa.ndim = d->ndim;
        fa.src_ncoef = fu.src_ncoef;
        fa.after = fu.after;
        for (int dd = 0; dd < d->ndim; ++dd) {
            fa.shape[dd] = pl->shape[dd];
            fa.dst_str[dd] = fu.dst_space < 0 ? 0 : pl->strides[fu.dst_space][dd];
            fa.src_str[dd] = fu.src_space < 0 ? 0 : pl->strides[fu.src_space][dd];
            fa.e_str[dd] = fu.e_space < 0 ? 0 : pl->strides[fu.e_space][dd];
        }
        hipLaunchKernelGGL(fuse_kernel, dim3((unsigned)((fa.n_entries + 255) / 256)), dim3(256), 0, ctx->stream, fa);
        e = hipGetLastError();
    }
    for (int i = 0; i < d->n_fuse_partial && e == hipSuccess; ++i) {
        const epgx_fuse_partial &fp = d->fuse_partial[i];
        FusePartialArgs fa;
        memset(&fa, 0, sizeof(fa));
        fa.pool = pl->d_coef;
        fa.dst_off = fp.dst_off;
        fa.src_off = fp.src_off;
        fa.dsrc_off = fp.dsrc_off < 0 ? -1 : fp.dsrc_off;
        fa.e_off = fp.e_off;
        fa.de_off = fp.de_off < 0 ? -1 : fp.de_off;
        fa.n_entries = (fp.dst_space < 0 ? 0 : space_extent[fp.dst_space]) + 1;
        fa.ndim = d->ndim;
        fa.src_ncoef = fp.src_ncoef;
        fa.dsrc_ncoef = fp.dsrc_ncoef;
        fa.after = fp.after;
        for (int dd = 0; dd < d->ndim; ++dd) {
            auto str = [&](int sp) { return sp < 0 ? (int64_t)0 : pl->strides[sp][dd]; };
            fa.shape[dd] = pl->shape[dd];
            fa.dst_str[dd] = str(fp.dst_space);
            fa.src_str[dd] = str(fp.src_space);
            fa.dsrc_str[dd] = fp.dsrc_off < 0 ? 0 : str(fp.dsrc_space);
            fa.e_str[dd] = str(fp.e_space);
            fa.de_str[dd] = fp.de_off < 0 ? 0 : str(fp.de_space);
        }
        hipLaunchKernelGGL(fuse_partial_kernel, dim3((unsigned)((fa.n_entries + 255) / 256)), dim3(256), 0, ctx->stream, fa);
        e = hipGetLastError();
    }
    uint32_t *d_logflags = nullptr;
    std::vector<uint32_t> logflags(log_jobs.size(), 0u);
    if (e == hipSuccess && !log_jobs.empty()) {
        e = dev_alloc(ctx, (void **)&d_logflags, sizeof(uint32_t) * log_jobs.size());
        if (e == hipSuccess) e = hipMemsetAsync(d_logflags, 0, sizeof(uint32_t) * log_jobs.size(), ctx->stream);
        for (size_t j = 0; j < log_jobs.size() && e == hipSuccess; ++j) {
            LogTabArgs la;
            memset(&la, 0, sizeof(la));
            la.pool = pl->d_coef;
            la.e_off = log_jobs[j].e_off;
            la.de_off = log_jobs[j].de_off;
            la.dst_off = pl->logtabs[j].off;
            la.n_entries = log_jobs[j].entries;
            la.flags = d_logflags;
            la.slot = (int32_t)j;
            hipLaunchKernelGGL(logtab_kernel, dim3((unsigned)((la.n_entries + 255) / 256)), dim3(256), 0, ctx->stream, la);
            e = hipGetLastError();
        }
        if (e == hipSuccess)
            e = hipMemcpyAsync(logflags.data(), d_logflags, sizeof(uint32_t) * log_jobs.size(), hipMemcpyDeviceToHost, ctx->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (d_logflags) dev_free(ctx, d_logflags);
    for (size_t j = 0; j < log_jobs.size(); ++j) {
        pl->logtabs[j].any = logflags[j] & 3u;
        if (logflags[j] & 4u) pl->logtabs[j].off = -1;   // not of the logarithmic form: records that need it do not fold
        if (trace && (j < 8 || (logflags[j] & 4u)))
            fprintf(stderr, "[epgx] plan_create log table %zu: value table at %lld, partial at %lld, %lld entries: wT %s, wL %s%s\n", j,
                    (long long)log_jobs[j].e_off, (long long)log_jobs[j].de_off, (long long)log_jobs[j].entries,
                    (logflags[j] & 1u) ? "nonzero" : "zero", (logflags[j] & 2u) ? "nonzero" : "zero",
                    (logflags[j] & 4u) ? " -- NOT of the logarithmic form" : "");
    }
    if (trace && !pl->t0_logd.empty()) {
        int routed = 0, generated = 0;
        for (int i = 0; i < d->n_ops; ++i)
            for (int v = 0; v < d->n_vars; ++v)
                if (pl->ops[i].opcode == EPGX_OP_T0 && pl->dops[i].coef_off[v] >= 0) (pl->t0_logd[(size_t)i * EPGX_MAX_VARS + v] ? routed : generated)++;
        fprintf(stderr, "[epgx] plan_create fused-echo partials: %d relaxation-only (logarithmic route), %d with a rotation partial\n", routed, generated);
    }
    lap("uploaded");
    if (e != hipSuccess) {
        epgx_plan_destroy(pl);
        return fail(e == hipErrorOutOfMemory ? EPGX_ERR_NOMEM : EPGX_ERR_HIP, "epgx_plan_create: %s",
                    hipGetErrorString(e));
    }
    *out = pl;
    return EPGX_OK;
}

extern "C" int epgx_plan_destroy(epgx_plan *pl) {
    if (!pl) return EPGX_OK;
    (void)hipSetDevice(pl->ctx->device);
    for (auto &pr : pl->packed) {
        dev_free(pl->ctx, pr.d_recs);
        dev_free(pl->ctx, pr.d_drecs);
        dev_free(pl->ctx, pr.d_runs);
        dev_free(pl->ctx, pr.d_grow);
        dev_free(pl->ctx, pr.d_druns);
        dev_free(pl->ctx, pr.d_ddruns);
        dev_free(pl->ctx, pr.d_bdruns);
    }
    dev_free(pl->ctx, pl->d_coef);
    dev_free(pl->ctx, pl->d_vidx);
    delete pl;
    return EPGX_OK;
}

// make sure the plan's cached table-index array covers [vox0, vox0+nvox)
static int ensure_vidx(epgx_plan *pl, int64_t vox0, int64_t nvox) {
    if (pl->n_spaces == 0) return EPGX_OK;
    if (pl->d_vidx && pl->vidx_vox0 == vox0 && pl->vidx_nvox == nvox) return EPGX_OK;
    epgx_ctx *ctx = pl->ctx;
    if (!pl->d_vidx || pl->vidx_cap < nvox) {
        dev_free(ctx, pl->d_vidx);
        pl->d_vidx = nullptr;
        // the 4-space kernel variant reads four rows: always allocate (and zero) that many
        const int rows = pl->n_spaces > 2 ? 4 : pl->n_spaces;
        HIP_TRY(dev_alloc(ctx, (void **)&pl->d_vidx, sizeof(int32_t) * (size_t)nvox * rows));
        HIP_TRY(hipMemsetAsync(pl->d_vidx, 0, sizeof(int32_t) * (size_t)nvox * rows, ctx->stream));
        pl->vidx_cap = nvox;
    }
    IndexArgs ia;
    memset(&ia, 0, sizeof(ia));
    ia.vidx = pl->d_vidx;
    ia.ld = nvox;
    ia.vox0 = vox0;
    ia.nvox = nvox;
    ia.n_spaces = pl->n_spaces;
    ia.ndim = pl->ndim;
    for (int i = 0; i < EPGX_MAX_DIMS; ++i) ia.shape[i] = pl->shape[i];
    memcpy(ia.strides, pl->strides, sizeof(ia.strides));
    const unsigned blocks = (unsigned)((nvox + 255) / 256);
    hipLaunchKernelGGL(index_kernel, dim3(blocks), dim3(256), 0, ctx->stream, ia);
    HIP_TRY(hipGetLastError());
    pl->vidx_vox0 = vox0;
    pl->vidx_nvox = nvox;
    return EPGX_OK;
}

// ------------------------------------------------------------------------------ signal reduction
extern "C" int epgx_signal_reduce(epgx_ctx *ctx, const void *signal, int64_t signal_ld, int32_t row0, int32_t row_step,
                                  int32_t n_rows, int32_t ndim, const int64_t *grid_shape, const uint8_t *reduce_axis,
                                  const void *weights, const int64_t *weight_strides, void *out, int64_t vox0, int64_t nvox_held) {
    if (!ctx || !signal || !grid_shape || !reduce_axis || !out || (weights && !weight_strides))
        return fail(EPGX_ERR_INVALID, "epgx_signal_reduce: NULL argument");
    if (ndim < 1 || ndim > EPGX_MAX_DIMS) return fail(EPGX_ERR_INVALID, "epgx_signal_reduce: ndim %d not in [1,%d]", ndim, EPGX_MAX_DIMS);
    if (n_rows < 0 || row0 < 0 || row_step < 1) return fail(EPGX_ERR_INVALID, "epgx_signal_reduce: bad row range");
    ReduceArgs a;
    memset(&a, 0, sizeof(a));
    int64_t nvox = 1, acc[EPGX_MAX_DIMS];
    for (int d = ndim - 1; d >= 0; --d) {
        if (grid_shape[d] < 1) return fail(EPGX_ERR_INVALID, "epgx_signal_reduce: grid_shape[%d] < 1", d);
        acc[d] = nvox;
        nvox *= grid_shape[d];
    }
    if (vox0 < 0 || nvox_held < 0 || vox0 + nvox_held > nvox)
        return fail(EPGX_ERR_INVALID, "epgx_signal_reduce: slab [%lld,%lld) outside the grid (%lld voxels)", (long long)vox0,
                    (long long)(vox0 + nvox_held), (long long)nvox);
    if (nvox_held > signal_ld) return fail(EPGX_ERR_INVALID, "epgx_signal_reduce: slab of %lld voxels exceeds signal_ld=%lld", (long long)nvox_held, (long long)signal_ld);
    a.n_out = a.n_red_total = 1;
    for (int d = 0; d < ndim; ++d) {
        const int64_t ws = weights ? weight_strides[d] : 0;
        if (ws < 0) return fail(EPGX_ERR_INVALID, "epgx_signal_reduce: negative weight stride");
        if (reduce_axis[d]) {
            a.red_size[a.n_red] = grid_shape[d];
            a.red_stride[a.n_red] = acc[d];
            a.red_wstride[a.n_red++] = ws;
            a.n_red_total *= grid_shape[d];
        } else {
            a.keep_size[a.n_keep] = grid_shape[d];
            a.keep_stride[a.n_keep] = acc[d];
            a.keep_wstride[a.n_keep++] = ws;
            a.n_out *= grid_shape[d];
        }
    }
    if (n_rows == 0) return EPGX_OK;
    if (n_rows > 65535) return fail(EPGX_ERR_UNSUPPORTED, "epgx_signal_reduce: more than 65535 rows per call");
    if (int rc = set_device(ctx)) return rc;
    a.signal = (const d2 *)signal;
    a.ld = signal_ld;
    a.row0 = row0;
    a.row_step = row_step;
    a.n_rows = n_rows;
    a.weights = (const d2 *)weights;
    a.out = (d2 *)out;
    a.vox0 = vox0;
    a.nheld = nvox_held;
    if (reduce_axis[ndim - 1])
        hipLaunchKernelGGL(reduce_wave_kernel, dim3((unsigned)((a.n_out + 3) / 4), (unsigned)n_rows), dim3(256), 0, ctx->stream, a);
    else
        hipLaunchKernelGGL(reduce_thread_kernel, dim3((unsigned)((a.n_out + 255) / 256), (unsigned)n_rows), dim3(256), 0, ctx->stream, a);
    HIP_TRY(hipGetLastError());
    return EPGX_OK;
}

// ------------------------------------------------------------------------------ records complex128 -> complex64
static int launch_narrow(epgx_ctx *ctx, const void *src, int64_t src_ld, void *dst, int64_t dst_ld, int64_t rows, int64_t cols) {
    if (!rows || !cols) return EPGX_OK;
    const unsigned bx = (unsigned)std::min<int64_t>((cols + 1023) / 1024, 4096), by = (unsigned)std::min<int64_t>(rows, 65535);
    hipLaunchKernelGGL(narrow_kernel, dim3(bx, by), dim3(256), 0, ctx->stream, (const d2 *)src, src_ld, (float2 *)dst, dst_ld, rows, cols);
    HIP_TRY(hipGetLastError());
    return EPGX_OK;
}

extern "C" int epgx_signal_narrow(epgx_ctx *ctx, const void *src, int64_t src_ld, void *dst, int64_t dst_ld, int64_t rows, int64_t cols) {
    if (!ctx || rows < 0 || cols < 0 || src_ld < cols || dst_ld < cols || (rows && cols && (!src || !dst)))
        return fail(EPGX_ERR_INVALID, "epgx_signal_narrow: bad argument");
    if (int rc = set_device(ctx)) return rc;
    return launch_narrow(ctx, src, src_ld, dst, dst_ld, rows, cols);
}

// ------------------------------------------------------------------------------ state
extern "C" int epgx_state_create(epgx_ctx *ctx, int64_t nvox, int32_t K, epgx_state **out) {
    if (!ctx || !out) return fail(EPGX_ERR_INVALID, "epgx_state_create: NULL argument");
    *out = nullptr;
    if (nvox < 1) return fail(EPGX_ERR_INVALID, "epgx_state_create: nvox < 1");
    if (!supported_K(K))
        return fail(EPGX_ERR_UNSUPPORTED, "epgx_state_create: K=%d, supported capacities are 64,128,256,512,1024", K);
    if (int rc = set_device(ctx)) return rc;
    epgx_state *st = new (std::nothrow) epgx_state();
    if (!st) return fail(EPGX_ERR_NOMEM, "epgx_state_create: host allocation failed");
    st->ctx = ctx;
    st->nvox = nvox;
    st->K = K;
    hipError_t e = dev_alloc(ctx, (void **)&st->data, sizeof(d2) * (size_t)nvox * 3 * K);
    if (e == hipSuccess) e = dev_alloc(ctx, (void **)&st->dens, sizeof(double) * (size_t)nvox);
    if (e != hipSuccess) {
        epgx_state_destroy(st);
        return fail(e == hipErrorOutOfMemory ? EPGX_ERR_NOMEM : EPGX_ERR_HIP, "epgx_state_create: %s",
                    hipGetErrorString(e));
    }
    const int64_t total = nvox * 3 * K;
    hipLaunchKernelGGL(state_init_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                       st->data, (int)K, st->dens, nvox);
    e = hipGetLastError();
    if (e != hipSuccess) {
        epgx_state_destroy(st);
        return fail(EPGX_ERR_HIP, "epgx_state_create: %s", hipGetErrorString(e));
    }
    *out = st;
    return EPGX_OK;
}

extern "C" int epgx_state_destroy(epgx_state *st) {
    if (!st) return EPGX_OK;
    (void)hipSetDevice(st->ctx->device);
    dev_free(st->ctx, st->data);
    dev_free(st->ctx, st->dens);
    delete st;
    return EPGX_OK;
}

extern "C" int epgx_state_upload(epgx_state *st, const double *half, const double *density) {
    if (!st || !half) return fail(EPGX_ERR_INVALID, "epgx_state_upload: NULL argument");
    if (int rc = epgx_memcpy_h2d(st->ctx, st->data, half, (int64_t)sizeof(d2) * st->nvox * 3 * st->K)) return rc;
    if (density) return epgx_memcpy_h2d(st->ctx, st->dens, density, (int64_t)sizeof(double) * st->nvox);
    return EPGX_OK;
}

extern "C" int epgx_state_download(const epgx_state *st, double *half, double *density) {
    if (!st || !half) return fail(EPGX_ERR_INVALID, "epgx_state_download: NULL argument");
    if (int rc = epgx_memcpy_d2h(st->ctx, half, st->data, (int64_t)sizeof(d2) * st->nvox * 3 * st->K)) return rc;
    if (density) return epgx_memcpy_d2h(st->ctx, density, st->dens, (int64_t)sizeof(double) * st->nvox);
    return EPGX_OK;
}

static int launch_copy(epgx_state *dst, const epgx_state *src, const int32_t *d_map) {
    epgx_ctx *ctx = dst->ctx;
    if (int rc = set_device(ctx)) return rc;
    const int64_t total = dst->nvox * 3 * dst->K;
    hipLaunchKernelGGL(state_copy_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream,
                       dst->data, (int)dst->K, (const d2 *)src->data, (int)src->K, d_map, dst->dens,
                       (const double *)src->dens, dst->nvox);
    HIP_TRY(hipGetLastError());
    return EPGX_OK;
}

extern "C" int epgx_state_copy(epgx_state *dst, const epgx_state *src) {
    if (!dst || !src) return fail(EPGX_ERR_INVALID, "epgx_state_copy: NULL argument");
    if (dst == src) return EPGX_OK;
    if (dst->ctx != src->ctx) return fail(EPGX_ERR_INVALID, "epgx_state_copy: states belong to different contexts");
    if (dst->nvox != src->nvox)
        return fail(EPGX_ERR_INVALID, "epgx_state_copy: nvox mismatch (%lld vs %lld)", (long long)dst->nvox,
                    (long long)src->nvox);
    return launch_copy(dst, src, nullptr);
}

extern "C" int epgx_state_broadcast(epgx_state *dst, const epgx_state *src, const int32_t *src_index) {
    if (!dst || !src || !src_index) return fail(EPGX_ERR_INVALID, "epgx_state_broadcast: NULL argument");
    if (dst == src) return fail(EPGX_ERR_INVALID, "epgx_state_broadcast: dst and src must differ");
    if (dst->ctx != src->ctx) return fail(EPGX_ERR_INVALID, "epgx_state_broadcast: different contexts");
    for (int64_t j = 0; j < dst->nvox; ++j)
        if (src_index[j] < 0 || src_index[j] >= src->nvox)
            return fail(EPGX_ERR_INVALID, "epgx_state_broadcast: src_index[%lld]=%d out of range", (long long)j,
                        src_index[j]);
    epgx_ctx *ctx = dst->ctx;
    if (int rc = set_device(ctx)) return rc;
    int32_t *d_map = nullptr;
    HIP_TRY(dev_alloc(ctx, (void **)&d_map, sizeof(int32_t) * (size_t)dst->nvox));
    hipError_t e = hipMemcpyAsync(d_map, src_index, sizeof(int32_t) * (size_t)dst->nvox, hipMemcpyHostToDevice,
                                  ctx->stream);
    int rc = EPGX_OK;
    if (e != hipSuccess) rc = fail(EPGX_ERR_HIP, "epgx_state_broadcast: %s", hipGetErrorString(e));
    if (!rc) rc = launch_copy(dst, src, d_map);
    (void)hipStreamSynchronize(ctx->stream);   // `src_index` is the caller's
    dev_free(ctx, d_map);
    return rc;
}

extern "C" int epgx_state_axpy(epgx_state *dst, const epgx_state *src, double alpha, int32_t zero_density) {
    if (!dst || !src) return fail(EPGX_ERR_INVALID, "epgx_state_axpy: NULL argument");
    if (dst->ctx != src->ctx) return fail(EPGX_ERR_INVALID, "epgx_state_axpy: states of different contexts");
    if (dst->nvox != src->nvox || dst->K != src->K)
        return fail(EPGX_ERR_INVALID, "epgx_state_axpy: shapes differ (%lld x %d vs %lld x %d)", (long long)dst->nvox, dst->K,
                    (long long)src->nvox, src->K);
    epgx_ctx *ctx = dst->ctx;
    if (int rc = set_device(ctx)) return rc;
    const int64_t n = dst->nvox * 3 * dst->K;
    hipLaunchKernelGGL(state_axpy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, dst->data,
                       (const d2 *)src->data, alpha, n);
    HIP_TRY(hipGetLastError());
    if (zero_density) HIP_TRY(hipMemsetAsync(dst->dens, 0, sizeof(double) * (size_t)dst->nvox, ctx->stream));
    return EPGX_OK;
}

extern "C" int epgx_state_info(const epgx_state *st, int64_t *nvox, int32_t *K, void **data, void **density) {
    if (!st) return fail(EPGX_ERR_INVALID, "epgx_state_info: NULL argument");
    if (nvox) *nvox = st->nvox;
    if (K) *K = st->K;
    if (data) *data = st->data;
    if (density) *density = st->dens;
    return EPGX_OK;
}

// ------------------------------------------------------------------------------ run
// Pack primitives [begin, end) into fused records  [misc] -> [T] -> [E] -> [S] -> [ADC].
// "S E" is rewritten "E S" first: E multiplies every order by the same coefficients and S only
// moves values, so the two commute bit for bit (the wrap value conj(B_1) * e0 equals
// conj(B_1 * conj(e0)) exactly); nothing else is reordered.
// which table of logarithmic partials (epgx_plan::logtabs) the relaxation stage of a record has for every variable
struct ELog {
    int32_t tab[EPGX_MAX_VARS];   // -1: the stage has no partial w.r.t. this variable
    bool blocked;                 // some partial of the stage has no log table: the record cannot fold
    int32_t t_op;                 // primitive index of the record's rotation stage, or -1
};

static void pack_records(const std::vector<epgx_op> &all, const std::vector<uint8_t> &zero_pattern,
                         const std::vector<epgx_dop> &dops, const std::vector<uint16_t> &dpattern, int begin, int end,
                         int K, bool fold, uint32_t identity_off, std::vector<Rec> &out,
                         std::vector<DRec> &dout, bool &use_lds, bool &has_adc, const std::vector<int32_t> *log_of = nullptr,
                         std::vector<ELog> *elog = nullptr) {
    std::vector<epgx_op> ops;
    for (int i = begin; i < end; ++i)
        if (all[i].opcode != EPGX_OP_NOP) {
            ops.push_back(all[i]);
            // travels with the operator through the reordering: zero pattern in the low byte, the
            // primitive's index (for its partial derivatives) above it
            ops.back().reserved = (int32_t)zero_pattern[i] | (i << 8);
        }
    const bool deriv = !dops.empty();
    for (bool swapped = true; swapped;) {
        swapped = false;
        for (size_t i = 0; i + 1 < ops.size(); ++i)
            if (ops[i].opcode == EPGX_OP_S && ops[i + 1].opcode == EPGX_OP_E) {
                std::swap(ops[i], ops[i + 1]);
                swapped = true;
            }
    }
    auto table_ix = [](const epgx_op &op) -> uint32_t {
        if (op.space < 0) return 0u;  // same entry for every voxel
        return (uint32_t)(op.ncoef * 8) | ((uint32_t)op.space << 24);  // entry bytes | index space
    };
    out.clear();
    dout.clear();
    use_lds = has_adc = false;
    Rec cur;
    DRec dcur;
    ELog lcur;
    memset(&cur, 0, sizeof(cur));
    memset(&dcur, 0, sizeof(dcur));
    auto no_logs = [&]() {
        for (int v = 0; v < EPGX_MAX_VARS; ++v) lcur.tab[v] = -1;
        lcur.blocked = false;
        lcur.t_op = -1;
    };
    no_logs();
    if (elog) elog->clear();
    int stage = 0;  // 1 misc, 2 leading S(+1), 3 T/MAT, 4 E, 5 S, 6 ADC
    auto flush = [&]() {   // (the leaf numbers are assigned at the end, after the fold pass)
        if (stage) {
            out.push_back(cur);
            if (deriv) dout.push_back(dcur);
            if (deriv && elog) elog->push_back(lcur);
        }
        memset(&cur, 0, sizeof(cur));
        memset(&dcur, 0, sizeof(dcur));
        no_logs();
        stage = 0;
    };
    auto partials = [&](const epgx_op &op, bool t_stage) {
        if (!deriv) return;
        const epgx_dop &dp = dops[(size_t)(op.reserved >> 8)];
        const uint32_t pattern = dpattern[(size_t)(op.reserved >> 8)];
        for (int v = 0; v < EPGX_MAX_VARS; ++v) {
            if (dp.coef_off[v] < 0) continue;
            const uint32_t pat = (pattern >> (2 * v)) & 3u;
            if (pat == 1) dcur.present |= (t_stage ? 256u : 4096u) << v;
            if (pat == 2 && t_stage) dcur.present |= 65536u << v;
            const bool with_const = t_stage && (pattern & (256u << v));   // generated partial of a T0 table: 14 per entry
            const uint32_t bytes = t_stage ? (with_const ? 112u : 80u) : 32u;
            const uint32_t ix = dp.space[v] < 0 ? 0u : (bytes | ((uint32_t)dp.space[v] << 24));
            if (t_stage) {
                dcur.t_off[v] = (uint32_t)(dp.coef_off[v] * 8);
                dcur.t_ix[v] = ix;
                dcur.present |= 1u << v;
                if (with_const) {   // the partial of the constant term sits where a relaxation partial would: slots 10..12 of the
                    dcur.e_off[v] = dcur.t_off[v] + 80u;   // partial line; such a record has no relaxation stage (below)
                    dcur.e_ix[v] = ix;
                    dcur.present |= 16u << v;
                }
            } else {
                dcur.e_off[v] = (uint32_t)(dp.coef_off[v] * 8);
                dcur.e_ix[v] = ix;
                dcur.present |= 16u << v;
                const int32_t tab = (log_of && !log_of->empty()) ? (*log_of)[(size_t)(op.reserved >> 8) * EPGX_MAX_VARS + v] : -1;
                lcur.tab[v] = tab;
                if (tab < 0) lcur.blocked = true;
            }
        }
    };
    auto is_matrix = [](const epgx_op &op) {
        return op.opcode == EPGX_OP_T || op.opcode == EPGX_OP_T0 || op.opcode == EPGX_OP_MAT || op.opcode == EPGX_OP_MAT0;
    };
    for (size_t oi = 0; oi < ops.size(); ++oi) {
        const epgx_op &op = ops[oi];
        int st;
        switch (op.opcode) {
        case EPGX_OP_T: case EPGX_OP_T0: case EPGX_OP_MAT: case EPGX_OP_MAT0: st = 3; break;
        case EPGX_OP_E: st = 4; break;
        case EPGX_OP_S:
            // "S T ..." : a shift by +1 (no truncation) directly in front of a rotation opens the
            // record of that rotation instead of being a record of its own -- every record costs
            // a dependent scalar fetch that the wave has to sit out
            // (only when the shift could not close the current record anyway, and when the rotation
            // is not followed by an E: those shapes have straight-line bodies)
            // (also behind a lone rotation when the NEXT rotation is followed by a shift of its own -- "T | S T S ..." : the
            // excitation of a train.  The shift saves no record either way, and the first repetition of the train then has the
            // shape of the others, so that the run-length folding takes all of them: EPGX_LEAD_FORWARD=0, measurements)
            st = ((stage == 0 || stage >= 5 ||
                   (stage == 3 && knobs().lead_forward && oi + 2 < ops.size() && ops[oi + 2].opcode == EPGX_OP_S && ops[oi + 2].ia == 1)) &&
                  op.ia == 1 && op.ib >= K - 1 && oi + 1 < ops.size() && is_matrix(ops[oi + 1]) &&
                  !(oi + 2 < ops.size() && ops[oi + 2].opcode == EPGX_OP_E))
                     ? 2
                     : 5;
            break;
        case EPGX_OP_ADC: st = 6; break;
        default: st = 1; break;
        }
        if (st <= stage || st == 1) flush();
        if (deriv && st == 4 && (cur.flags & F_T0) && (dcur.present & 0x70u)) flush();   // (the relaxation-partial slots are taken)
        switch (op.opcode) {
        case EPGX_OP_T: case EPGX_OP_T0: case EPGX_OP_MAT: case EPGX_OP_MAT0:
            cur.flags |= (op.opcode == EPGX_OP_T)    ? F_T
                         : (op.opcode == EPGX_OP_T0) ? (F_T | F_T0)
                         : (op.opcode == EPGX_OP_MAT) ? F_MAT
                                                      : (F_MAT | F_MAT0);
            if ((op.opcode == EPGX_OP_T || op.opcode == EPGX_OP_T0) && (op.reserved & 0xff) == 1) cur.flags |= F_TX;
            if ((op.opcode == EPGX_OP_T || op.opcode == EPGX_OP_T0) && (op.reserved & 0xff) == 3) cur.flags |= F_TY;
            partials(op, true);
            lcur.t_op = op.reserved >> 8;
            cur.t_off = (uint32_t)(op.coef_off * 8);
            cur.t_ix = table_ix(op);
            break;
        case EPGX_OP_E:
            cur.flags |= F_E | ((op.reserved & 0xff) == 2 ? F_ER : 0u);
            partials(op, false);
            cur.e_off = (uint32_t)(op.coef_off * 8);
            cur.e_ix = table_ix(op);
            break;
        case EPGX_OP_S:
            if (st == 2) {
                cur.flags |= F_S0;
                break;
            }
            cur.flags |= F_S;
            cur.shift = op.ia;
            if (op.ib < K - 1) {
                cur.flags |= F_TRUNC;
                cur.kmax = op.ib;
            }
            if (std::abs(op.ia) > 1) use_lds = true;
            break;
        case EPGX_OP_ADC:
            cur.flags |= F_ADC | (op.ib ? F_ADC_Z : 0u);
            cur.slot = op.ia;
            has_adc = true;
            break;
        case EPGX_OP_D: case EPGX_OP_GS:
            cur.flags |= (op.opcode == EPGX_OP_D) ? F_D : F_GS;
            cur.t_off = (uint32_t)(op.coef_off * 8);
            cur.t_ix = table_ix(op);
            if (op.opcode == EPGX_OP_GS) use_lds = true;
            st = 7;  // nothing else may join this record
            break;
        case EPGX_OP_SPOIL: cur.flags |= F_SPOIL; break;
        case EPGX_OP_RESET: cur.flags |= F_RESET; break;
        case EPGX_OP_PD:
            cur.flags |= F_PD | (op.ia ? F_PD_RESET : 0u);
            cur.e_off = (uint32_t)(op.coef_off * 8);
            cur.e_ix = table_ix(op);
            st = 4;  // the E slot of this record is taken
            break;
        default: break;
        }
        stage = st;
    }
    flush();

    // ---- run-time fold (F_FOLD, fold_T in epgx_kernels.hip.h).  A rotation next to precession-free relaxations whose
    // tables do not share its index space -- T over a B1 axis, E over (T1, T2): the product table would be the whole grid
    // PER PULSE, so the host's E.T.E fusion (epgx_fuse) does not apply -- becomes ONE stage  E_a . T . E_b  whose
    // coefficients every wavefront computes for its voxels when it meets the record: 3 instructions per record instead of
    // 6 per order and relaxation.  E_a = the relaxation stage of the record itself; E_b = the relaxation that closes the
    // PREVIOUS record (it commutes with the integer shift and the truncation behind it: E scales every order alike, and
    // the recovery only touches Z_0, which a shift does not move).  An ADC behind E_b pins it; so does a reset or density
    // stage in front of the rotation.  A SPOILER there is folded as well (F_FOLD_SPOIL: zero F columns).  The decisions only look at neighbours inside one ADC-to-ADC
    // span, so the per-timestep launches (ranges cut at the probes) and the state-resident launch of the whole
    // sequence fold alike -- the same chains in the same order in every kernel (same bits; the one exception is the sum /
    // difference form of rotations about x in the 64-order state-resident kernels: last bits, include/epgx.h epgx_run).
    // (The decision is per PLAN, never per launch capacity: the same plan must run the same chains at every K.  A host
    // that runs a plan with 16 orders per voxel sets EPGX_PLAN_NO_FOLD: with one order per lane a relaxation stage is 6
    // instructions per record, less than the fold's extra loads cost -- the 1000-TR MRF train with max_nstate = 10 takes
    // 28.2 ms unfolded and 35.6 ms folded at K = 16; K = 32: 43.8 / 33.2 ms.)
    if (fold && !deriv) {
        const uint32_t misc = F_SPOIL | F_RESET | F_PD | F_PD_RESET;
        for (size_t j = 0; j < out.size(); ++j) {
            Rec &c = out[j];
            if (!(c.flags & F_T) || (c.flags & (F_MAT | F_T0 | F_FOLD | F_D | F_GS | F_PD))) continue;
            if ((c.flags & F_S) && c.shift != 1) continue;             // the shift word is about to carry E_b's table
            if ((c.flags & F_E) && !(c.flags & F_ER)) continue;        // precession behind the rotation: not a real diagonal
            const bool has_a = (c.flags & F_E) != 0;
            // a spoiler right in front of the rotation (and no reset / density stage with it) joins the fold: F <- 0 means
            // that T only sees Z, i.e. the F columns of E_b count as zero; E_b itself commutes with the spoiler
            const bool spoil = (c.flags & F_SPOIL) && !(c.flags & (misc & ~(uint32_t)F_SPOIL));
            Rec *p = j > 0 ? &out[j - 1] : nullptr;
            const bool has_b = p && (p->flags & F_E) && (p->flags & F_ER) && !(c.flags & (misc & ~(uint32_t)F_SPOIL)) &&
                               !(p->flags & (F_ADC | F_ADC_Z | F_PD | F_PD_RESET | F_D | F_GS | F_FOLD));
            if (!has_a && !has_b && !spoil) continue;
            const uint32_t a_off = has_a ? c.e_off : identity_off, a_ix = has_a ? c.e_ix : 0u;
            c.flags = (c.flags & ~(uint32_t)(F_E | F_ER)) | F_FOLD | F_T0;
            if (spoil) c.flags = (c.flags & ~(uint32_t)F_SPOIL) | F_FOLD_SPOIL;
            c.e_off = a_off;
            c.e_ix = a_ix;
            c.shift = (int32_t)(has_b ? p->e_off : identity_off);
            if (has_b) {
                if (p->e_ix & 0xffffffu) c.flags |= F_FOLD_BVOX | (((p->e_ix >> 24) & 3u) << 21);
                p->flags &= ~(uint32_t)(F_E | F_ER);
                p->e_off = p->e_ix = 0;
                // what is left of the previous record: nothing, or a lone S(+1) that can lead this record
                const uint32_t rest = p->flags & 0xffffffu;
                const bool lone_shift = (rest & ~(uint32_t)F_TRUNC) == F_S && p->shift == 1 && !(c.flags & F_S0) &&
                                        (!(rest & F_TRUNC) || !(c.flags & F_S));   // (one kmax per record: the trailing shift's)
                if (lone_shift) {
                    c.flags |= F_S0 | (rest & F_TRUNC);
                    if (rest & F_TRUNC) c.kmax = p->kmax;
                    p->flags = 0;
                }
            }
        }
        out.erase(std::remove_if(out.begin(), out.end(), [](const Rec &r) { return (r.flags & 0xffffffu) == 0; }), out.end());
    }
    for (Rec &r : out)   // K < 64 always runs rows_kernel, whose leaves truncate themselves (see record_leaf)
        r.flags = (r.flags & 0xffffffu) | ((K < 64 ? record_leaf<true>(r.flags, r.shift) : record_leaf<false>(r.flags, r.shift)) << 24);
}

// The run-folded record list of a launch that starts from EQUILIBRIUM (one populated order), cut where the populated orders
// outgrow 16 and 32 (rows_grow_kernel: the reference grows its state matrix the same way, functions.py:135 / shift.py:86).
// `top` = the highest order that can hold anything: every S(+-1) of a record adds one (resets and truncations are ignored:
// `top` only ever over-estimates, which is safe).  Repeat-count records and header runs are cut at the boundaries.
// Returns the share of record executions that run below 64 orders.
static double grow_split(const std::vector<Rec> &runs, std::vector<Rec> &out, int &n1, int &n2) {
    auto shifts_of = [](const Rec &r) { return ((r.flags & F_S0) ? 1 : 0) + ((r.flags & F_S) ? 1 : 0); };
    static const int cap[3] = {15, 31, 1 << 30};
    int top = 0, phase = 0;
    n1 = n2 = -1;
    double work[3] = {0, 0, 0};
    auto next_phase = [&]() {
        if (phase == 0) n1 = (int)out.size();
        else n2 = (int)out.size();
        ++phase;
    };
    for (size_t i = 0; i < runs.size();) {
        const Rec &r = runs[i];
        const uint32_t head = r.flags >> 24;
        const int count = (int)((uint32_t)r.kmax >> 16);
        if (head == LEAF_PAIR || head == LEAF_SINGLE) {
            const int per = head == LEAF_PAIR ? 2 : 1;   // records per repetition
            int d = 0;
            for (int j = 0; j < per; ++j) d += shifts_of(runs[i + 1 + (size_t)j]);
            int done = 0;
            while (done < count) {
                int m = d > 0 ? (cap[phase] - top) / d : count - done;
                m = std::min(m, count - done);
                if (m <= 0) {
                    next_phase();
                    continue;
                }
                Rec h = r;
                h.kmax = m << 16;
                out.push_back(h);
                for (int j = 0; j < per * m; ++j) out.push_back(runs[i + 1 + (size_t)(per * done + j)]);
                work[phase] += (double)per * m;
                top += m * d;
                done += m;
            }
            i += 1 + (size_t)per * (size_t)count;
            continue;
        }
        const int d = shifts_of(r), rep = std::max(count, 1);
        int done = 0;
        while (done < rep) {
            int m = d > 0 ? (cap[phase] - top) / d : rep - done;
            m = std::min(m, rep - done);
            if (m <= 0) {
                next_phase();
                continue;
            }
            Rec c = r;
            c.kmax = (r.kmax & 0xffff) | (m << 16);
            if (r.flags & F_ADC) c.slot = r.slot + done;   // (a repeat count implies consecutive ADC rows)
            out.push_back(c);
            work[phase] += m;
            top += m * d;
            done += m;
        }
        ++i;
    }
    if (n1 < 0) n1 = (int)out.size();
    if (n2 < 0) n2 = (int)out.size();
    n1 = std::min(n1, n2);
    const double all = work[0] + work[1] + work[2];
    return all > 0 ? (work[0] + work[1]) / all : 0.0;
}

static int get_packed(epgx_plan *pl, int begin, int end, int K, const PackedRange **out) {
    for (const auto &pr : pl->packed)
        if (pr.begin == begin && pr.end == end && pr.K == K) {
            *out = &pr;
            return EPGX_OK;
        }
    std::vector<Rec> recs;
    std::vector<DRec> drecs;
    PackedRange pr;
    pr.begin = begin;
    pr.end = end;
    pr.K = K;
    std::vector<ELog> elog;
    pack_records(pl->ops, pl->zero_pattern, pl->dops, pl->dpattern, begin, end, K, pl->fold, (uint32_t)(pl->n_pool * 8), recs, drecs,
                 pr.use_lds, pr.has_adc, &pl->log_of, &elog);
    pr.n_rec = (int)recs.size();
    for (const Rec &r : recs) pr.big_shift = pr.big_shift || ((r.flags & F_S) && !(r.flags & F_FOLD) && std::abs(r.shift) > 1);
    for (const Rec &r : recs) pr.has_gs = pr.has_gs || (r.flags & F_GS);
    pr.seq_slots = true;
    int expect = -1;
    for (const Rec &r : recs) pr.has_pd = pr.has_pd || (r.flags & F_PD);
    for (const Rec &r : recs)
        if (r.flags & F_ADC) {
            if (expect < 0) pr.first_slot = r.slot;
            else if (r.slot != expect) pr.seq_slots = false;
            expect = r.slot + 1;
        }
    {
        std::vector<std::pair<uint32_t, uint32_t>> seen;
        auto fresh = [&](uint32_t off, uint32_t ix) {
            if ((ix & 0xffffffu) == 0) return false;   // same entry for every voxel: hot in the caches
            for (auto &q : seen)
                if (q.first == off && q.second == ix) return false;
            if (seen.size() < 4096) seen.emplace_back(off, ix);
            return true;
        };
        for (int i = 0; i < pr.n_rec; ++i) {
            const Rec &r = recs[(size_t)i];
            bool any = false;
            if (r.flags & (F_T | F_MAT | F_D | F_GS)) any |= fresh(r.t_off, r.t_ix);
            if (r.flags & (F_E | F_PD)) any |= fresh(r.e_off, r.e_ix);
            if (any) pr.pf_count = i + 1;
        }
    }
    // Folded copy of the records for rows_kernel<.., RUNS>:
    //  * a run of identical records (an MSE train: same shape, same table entries, consecutive ADC rows) becomes one
    //    record with a repeat count in the upper half of the kmax word (rows_run);
    //  * a run of >= 4 record PAIRS [T, E, S(+1)?, ADC] [E, S(+1)] of constant shapes but arbitrary tables (the repetitions of an
    //    SSFP / MRF train that cannot be fused) gets a header record in front (leaf byte LEAF_PAIR, shape code, number
    //    of pairs): rows_pair_run.
    // Kept when it saves a quarter of the records or pair runs cover half of them.
    std::vector<Rec> runs;
    if (K <= 64 && drecs.empty() && pr.n_rec) {
        auto leaf_of = [&](const Rec &r) {   // with the truncation handled inside the leaf (K = 64 records carry LEAF_NONE for it)
            const uint32_t l = r.flags >> 24;
            return (l == LEAF_NONE && (r.flags & F_TRUNC)) ? record_leaf<true>(r.flags & 0xffffffu, r.shift) : l;
        };
        auto pair_code = [&](const Rec &a, const Rec &b) -> int {   // -1: not a pair this kernel loops over
            int code = -1;
            for (int c = 0; c < 16 && code < 0; ++c)
                if (leaf_of(a) == leaf_id((c & 1) ? 2 : 1, (c & 2) ? 2 : 1, (c & 8) != 0, true, false) &&
                    leaf_of(b) == leaf_id(0, (c & 4) ? 2 : 1, true, false, false))
                    code = c;
            return code;
        };
        // (same stages AND same table geometry -- entry size and index space: the kernel hoists the per-lane entry offsets)
        auto same_shape = [](const Rec &x, const Rec &y) {
            return x.flags == y.flags && x.shift == y.shift && x.kmax == y.kmax && x.t_ix == y.t_ix && x.e_ix == y.e_ix;
        };
        // run of folded records of one shape (rows_single_run): stages and table geometry equal, table offsets free
        auto single_code = [&](const Rec &a) -> int {
            if (!(a.flags & F_FOLD)) return -1;
            for (int c = 0; c < 16; ++c)
                if (leaf_of(a) == leaf_id((c & 1) ? 4 : 3, 0, (c & 4) != 0, (c & 8) != 0, (c & 2) != 0)) return c;
            return -1;
        };
        auto same_fold_shape = [](const Rec &x, const Rec &y) {
            return x.flags == y.flags && (x.kmax & 0xffff) == (y.kmax & 0xffff) && x.t_ix == y.t_ix && x.e_ix == y.e_ix;
        };
        auto identical = [](const Rec &x, const Rec &y) {
            return x.flags == y.flags && x.shift == y.shift && x.kmax == y.kmax && x.t_off == y.t_off && x.e_off == y.e_off &&
                   x.t_ix == y.t_ix && x.e_ix == y.e_ix;
        };
        size_t in_pairs = 0;
        bool back_is_plain = false;   // runs.back() is an ordinary record (not part of a pair run): a repeat may fold into it
        for (int i = 0; i < pr.n_rec;) {
            const int scode = single_code(recs[(size_t)i]);
            if (scode >= 0 && !(i + 1 < pr.n_rec && identical(recs[(size_t)i], recs[(size_t)i + 1]))) {
                int n = 1;   // (a train of IDENTICAL records is folded into a repeat count instead, below)
                while (i + n < pr.n_rec && n < 0x7fff && same_fold_shape(recs[(size_t)i], recs[(size_t)i + n]) &&
                       !(i + n + 1 < pr.n_rec && identical(recs[(size_t)i + n], recs[(size_t)i + n + 1])))
                    ++n;
                if (n >= 4) {
                    Rec head;
                    memset(&head, 0, sizeof(head));
                    head.flags = (LEAF_SINGLE << 24) | (uint32_t)scode;
                    head.kmax = n << 16;
                    runs.push_back(head);
                    for (int j = 0; j < n; ++j) {
                        runs.push_back(recs[(size_t)i + j]);
                        runs.back().kmax = (runs.back().kmax & 0xffff) | (1 << 16);
                    }
                    in_pairs += (size_t)n;
                    i += n;
                    back_is_plain = false;
                    continue;
                }
            }
            int code = i + 1 < pr.n_rec ? pair_code(recs[(size_t)i], recs[(size_t)i + 1]) : -1;
            int npairs = 0;
            if (code >= 0) {
                npairs = 1;
                while (i + 2 * npairs + 1 < pr.n_rec && npairs < 0x7fff && same_shape(recs[(size_t)i], recs[(size_t)i + 2 * npairs]) &&
                       same_shape(recs[(size_t)i + 1], recs[(size_t)i + 2 * npairs + 1]))
                    ++npairs;
            }
            if (npairs >= 4) {
                Rec head;
                memset(&head, 0, sizeof(head));
                head.flags = (LEAF_PAIR << 24) | (uint32_t)code;
                head.kmax = npairs << 16;
                runs.push_back(head);
                for (int j = 0; j < 2 * npairs; ++j) {
                    runs.push_back(recs[(size_t)i + j]);
                    runs.back().kmax = (runs.back().kmax & 0xffff) | (1 << 16);
                }
                in_pairs += 2 * (size_t)npairs;
                i += 2 * npairs;
                back_is_plain = false;
                continue;
            }
            const Rec &r = recs[(size_t)i];
            bool same = false;
            if (back_is_plain && (r.flags >> 24) != LEAF_NONE) {
                const Rec &q = runs.back();
                const int rep = (int)((uint32_t)q.kmax >> 16);
                same = q.flags == r.flags && q.shift == r.shift && (q.kmax & 0xffff) == r.kmax && q.t_off == r.t_off &&
                       q.e_off == r.e_off && q.t_ix == r.t_ix && q.e_ix == r.e_ix && rep < 0x7fff &&
                       (!(r.flags & F_ADC) || r.slot == q.slot + rep);
            }
            if (same) runs.back().kmax += 1 << 16;
            else {
                runs.push_back(r);
                runs.back().kmax = (r.kmax & 0xffff) | (1 << 16);
                back_is_plain = true;
            }
            ++i;
        }
        if (runs.size() * 4 > (size_t)pr.n_rec * 3 && in_pairs * 2 < (size_t)pr.n_rec) runs.clear();
    }
    // K = 64: the same list cut into phases of 16 / 32 / 64 orders per voxel for launches from equilibrium (rows_grow_kernel); kept
    // when at least a tenth of the record executions run below 64 orders (a 20-echo train: 15 of 20; a 1000-TR train: 30 of 1000)
    std::vector<Rec> grow;
    if (K == 64 && !runs.empty()) {
        const double early = grow_split(runs, grow, pr.grow1, pr.grow2);
        if (early < knobs().grow_share) grow.clear();   // (EPGX_GROW_SHARE, measurements)
        if (knobs().grow_min >= 2) pr.grow1 = 0;   // (EPGX_GROW_MIN=2, measurements: first phase at 2 orders per lane)
        if (tracing())
            for (size_t i = 0; i < grow.size(); ++i)
                fprintf(stderr, "[epgx] grow list %zu: leaf %u flags %06x x %u (orders <= %d)%s\n", i, grow[i].flags >> 24, grow[i].flags & 0xffffffu,
                        (uint32_t)grow[i].kmax >> 16, grow[i].kmax & 0xffff, (int)i == pr.grow1 || (int)i == pr.grow2 ? "   <- next phase" : "");
    }
    // K = 128 .. 2048: where the populated orders of a launch from equilibrium outgrow 64, 128 .. 1536 (run_contig_grow_kernel; the
    // two legs at 2048 orders).  `top` = the highest order that can hold anything, as in grow_split: every shift of a record adds one
    if (K >= 128 && drecs.empty() && pr.n_rec && !pr.use_lds) {
        int top = 0, phase = 0;
        double below = 0;
        static const int cap[6] = {63, 127, 255, 511, 1023, 1535};
        for (int &g : pr.cgrow) g = pr.n_rec;
        for (int i = 0; i < pr.n_rec; ++i) {
            const Rec &r = recs[(size_t)i];
            top += ((r.flags & F_S0) ? 1 : 0) + ((r.flags & F_S) ? 1 : 0);
            while (phase < 6 && top > cap[phase]) pr.cgrow[phase++] = i;
            if (phase < 5 && 64 << phase < K) below += 1;
        }
        for (int i = 0; i < pr.cgrow[3] && i < pr.n_rec; ++i) pr.cgrow_adc3 += (recs[(size_t)i].flags & F_ADC) ? 1 : 0;
        for (int q = 0; q < 6; ++q)
            if (cap[q] + 1 >= K) pr.cgrow[q] = pr.n_rec;     // (no phase at or above the capacity)
        for (int q = 1; q < 6; ++q) pr.cgrow[q] = std::max(pr.cgrow[q], pr.cgrow[q - 1]);
        pr.cgrow_share = below / pr.n_rec;
    }
    // Derivative plans at 64 orders: runs of >= 4 records of one shape get a header (leaf byte LEAF_DRUN, shape code, count) and
    // run on rotating order slots (drun_kernel, epgx_drun_kernels.hip.h); kept when the runs cover at least half of the
    // records.  Two families of shapes:
    //   * fused echoes  [S(+1)?  E.T.E + generated partials  S(+1)?  ADC]  (the host fused the tables: epgx_fuse_partial);
    //   * repetitions FOLDED AT RUN TIME (DRUN_FOLD)  [S(+1)?  E_a . T . E_b  S(+1)?  ADC]  -- a rotation over one index space
    //     between real relaxations over another (MRF / SSFP trains over a (T1, T2, B1) grid).  The fold happens HERE, for this
    //     array only (every other kernel keeps walking the unfolded records of d_recs): E_a = the relaxation stage of the
    //     rotation's own record, E_b = the record in front of it when that is nothing but a real relaxation and a shift by
    //     one.  The rotation's partial is folded like the rotation (a . dT . b per coefficient); a relaxation's partial
    //     enters through its table of logarithmic partials (epgx_plan::logtabs) -- every relaxation partial of both stages
    //     needs one, else the record stays unfolded.  Records the runs leave over are emitted UNFOLDED (their originals).
    std::vector<Rec> druns;
    std::vector<DRec> ddruns;
    std::vector<DRecB> bdruns;
    if ((K == 64 || K == 32 || K == 16) && !drecs.empty() && pr.n_rec) {   // (16 / 32 orders: folded repetitions only, packed_dfold_kernel)
        const int nv = pl->n_vars;
        struct Item { Rec r; DRec d; DRecB b; int lo, hi; bool folded, logd, moved; };
        std::vector<Item> fl;
        fl.reserve((size_t)pr.n_rec);
        const uint32_t identity_off = (uint32_t)(pl->n_pool * 8), zeros_off = (uint32_t)((pl->n_pool + 8) * 8);
        // (plan_create fills log_of for derivative plans that may fold: EPGX_FOLD, EPGX_PLAN_NO_FOLD)
        const bool dfold = !pl->log_of.empty() && elog.size() == recs.size();
        for (int j = 0; j < pr.n_rec; ++j) {
            Item it;
            memset(&it, 0, sizeof(it));
            it.r = recs[(size_t)j];
            it.d = drecs[(size_t)j];
            it.lo = it.hi = j;
            const Rec &c = recs[(size_t)j];
            const uint32_t cf = c.flags & 0xffffffu;
            // (a spoiler in front of the rotation joins the fold at 16 / 32 orders -- where spoiled trains live: F_FOLD_SPOIL, the F
            // columns of E_b count as zero for the STATE; packed_dfold_kernel knows what that means for the derivative states)
            const uint32_t no_spoil = K == 64 ? (uint32_t)F_SPOIL : 0u;
            bool can = dfold && (cf & F_T) && (cf & F_ADC) &&
                       !(cf & (F_MAT | F_MAT0 | F_T0 | F_FOLD | F_D | F_GS | F_PD | F_PD_RESET | no_spoil | F_RESET | F_ADC_Z)) &&
                       !((cf & F_S) && c.shift != 1) && !((cf & F_E) && !(cf & F_ER));
            const bool has_a = (cf & F_E) != 0;
            if (can && has_a && elog[(size_t)j].blocked) can = false;
            // (three derivative states at 64 orders: the kernel carries one partial line of the rotation, epgx_drun_kernels.hip.h)
            if (can && K == 64 && nv == 3 && !EPGX_DF3_SPLIT && __builtin_popcount(drecs[(size_t)j].present & 7u) > 1) can = false;
            bool has_b = false;
            if (can && j > 0 && !fl.empty() && !fl.back().folded && fl.back().lo == j - 1) {
                const Rec &q = recs[(size_t)j - 1];
                const uint32_t rest = q.flags & 0xffffffu;
                has_b = (rest & F_E) && (rest & F_ER) && !(rest & ~(uint32_t)(F_E | F_ER | F_S | F_TRUNC)) &&
                        (!(rest & F_S) || q.shift == 1) && !elog[(size_t)j - 1].blocked &&
                        !((cf & F_S0) && (rest & F_S)) && (!(rest & F_TRUNC) || !(cf & F_S));
            }
            if (!can || (!has_a && !has_b)) {
                // a fused echo (EPGX_OP_T0 from the host's fusion) whose partials w.r.t. some variables come from its relaxations
                // alone: those variables take the logarithmic route (weights of E_a / E_b instead of a generated partial table)
                const int t_op = dfold && K == 64 && (cf & F_T0) ? elog[(size_t)j].t_op : -1;
                if (t_op >= 0 && !pl->t0_logd.empty()) {
                    DRec nd = it.d;
                    DRecB nb;
                    memset(&nb, 0, sizeof(nb));
                    bool any = false, ok = true;
                    for (int v = 0; v < EPGX_MAX_VARS; ++v) nb.off[v] = zeros_off;
                    for (int v = 0; v < nv && ok; ++v) {
                        if (!pl->t0_logd[(size_t)t_op * EPGX_MAX_VARS + v] || !(nd.present & (1u << v))) continue;
                        const int32_t ta = pl->t0_log[((size_t)t_op * EPGX_MAX_VARS + v) * 2], tb = pl->t0_log[((size_t)t_op * EPGX_MAX_VARS + v) * 2 + 1];
                        if ((ta >= 0 && pl->logtabs[(size_t)ta].off < 0) || (tb >= 0 && pl->logtabs[(size_t)tb].off < 0)) continue;   // not of the logarithmic form
                        nd.present &= ~(((1u | 16u | 256u | 65536u) << v));
                        nd.t_off[v] = nd.t_ix[v] = 0;
                        nd.e_off[v] = zeros_off;
                        nd.e_ix[v] = 0;
                        if (ta >= 0) {
                            const auto &lt = pl->logtabs[(size_t)ta];
                            nd.e_off[v] = (uint32_t)(lt.off * 8);
                            nd.e_ix[v] = lt.space < 0 ? 0u : (16u | ((uint32_t)lt.space << 24));
                            nb.logs |= ((lt.any & 1u) ? (1u << v) : 0u) | ((lt.any & 2u) ? (16u << v) : 0u);
                        }
                        if (tb >= 0) {
                            const auto &lt = pl->logtabs[(size_t)tb];
                            nb.off[v] = (uint32_t)(lt.off * 8);
                            nb.ix[v] = lt.space < 0 ? 0u : (16u | ((uint32_t)lt.space << 24));
                            nb.logs |= ((lt.any & 1u) ? (256u << v) : 0u) | ((lt.any & 2u) ? (4096u << v) : 0u);
                        }
                        any = true;
                    }
                    // (three derivative states: one partial line of the rotation at most)
                    if (any && !(nv == 3 && __builtin_popcount(nd.present & 7u) > 1)) {
                        it.d = nd;
                        it.b = nb;
                        it.logd = true;
                    }
                }
                fl.push_back(it);
                continue;
            }
            Rec f = c;
            f.flags = (cf & ~(uint32_t)(F_E | F_ER | F_SPOIL)) | F_FOLD | F_T0 | ((cf & F_SPOIL) ? (uint32_t)F_FOLD_SPOIL : 0u) | (LEAF_NONE << 24);
            f.e_off = has_a ? c.e_off : identity_off;
            f.e_ix = has_a ? c.e_ix : 0u;
            f.shift = (int32_t)identity_off;
            DRec fd;
            memset(&fd, 0, sizeof(fd));
            DRecB fb;
            memset(&fb, 0, sizeof(fb));
            const DRec &dc = drecs[(size_t)j];
            for (int v = 0; v < EPGX_MAX_VARS; ++v) {
                fd.e_off[v] = fb.off[v] = zeros_off;
                if (v < nv && (dc.present & (1u << v))) {   // the rotation's partial: folded like the rotation, constant term included
                    fd.t_off[v] = dc.t_off[v];
                    fd.t_ix[v] = dc.t_ix[v];
                    fd.present |= (dc.present & ((1u << v) | (256u << v) | (65536u << v))) | (16u << v);
                }
                const int32_t ta = has_a ? elog[(size_t)j].tab[v] : -1;
                if (ta >= 0) {
                    const auto &lt = pl->logtabs[(size_t)ta];
                    fd.e_off[v] = (uint32_t)(lt.off * 8);
                    fd.e_ix[v] = lt.space < 0 ? 0u : (16u | ((uint32_t)lt.space << 24));
                    fb.logs |= ((lt.any & 1u) ? (1u << v) : 0u) | ((lt.any & 2u) ? (16u << v) : 0u);
                }
            }
            if (has_b) {
                const Rec &q = recs[(size_t)j - 1];
                const uint32_t rest = q.flags & 0xffffffu;
                f.shift = (int32_t)q.e_off;
                if (q.e_ix & 0xffffffu) f.flags |= F_FOLD_BVOX | (((q.e_ix >> 24) & 3u) << 21);
                if (rest & F_S) {
                    f.flags |= F_S0 | (rest & F_TRUNC);
                    if (rest & F_TRUNC) f.kmax = q.kmax;
                }
                for (int v = 0; v < nv; ++v) {
                    const int32_t tb = elog[(size_t)j - 1].tab[v];
                    if (tb < 0) continue;
                    const auto &lt = pl->logtabs[(size_t)tb];
                    fb.off[v] = (uint32_t)(lt.off * 8);
                    fb.ix[v] = lt.space < 0 ? 0u : (16u | ((uint32_t)lt.space << 24));
                    fb.logs |= ((lt.any & 1u) ? (256u << v) : 0u) | ((lt.any & 2u) ? (4096u << v) : 0u);
                }
                fl.pop_back();
                it.lo = j - 1;
            }
            it.r = f;
            it.d = fd;
            it.b = fb;
            it.folded = true;
            fl.push_back(it);
        }
        auto shape_of = [&](const Item &x) {
            if (x.folded) return dfold_shape(x.r.flags & 0xffffffu, x.d.present, nv, K != 64);
            if (K != 64) return -1;
            const int code = drun_shape(x.r.flags & 0xffffffu, x.r.shift, x.d.present, nv);
            return (code >= 0 && x.logd) ? (code | (int)DRUN_LOGD) : code;
        };
        // A trailing S(+1) that closes the record in front of a train (the excitation pulse: [T S] [T0 S ADC] [S0 T0 S ADC] ...) is
        // the LEADING shift of the train's first record just as well -- same stages in the same order.  Moved, the first echo
        // has the shape of the others and joins their run (20 echoes: 20 records in the run instead of 16 + 4 flag-tested ones).
        if (K == 64)
            for (size_t j = 1; j + 1 < fl.size(); ++j) {
                Item &q = fl[j - 1], &c = fl[j];
                const Item &n = fl[j + 1];
                const uint32_t qf = q.r.flags & 0xffffffu, cf = c.r.flags & 0xffffffu, nf2 = n.r.flags & 0xffffffu;
                if (q.folded || c.folded != n.folded || q.lo != q.hi || c.lo != c.hi) continue;
                if (!(qf & F_S) || q.r.shift != 1 || (qf & (F_TRUNC | F_ADC | F_ADC_Z | F_FOLD))) continue;    // (nothing behind that shift)
                if ((cf & F_S0) || !(nf2 & F_S0) || (cf | F_S0) != nf2 || c.logd != n.logd || shape_of(n) < 0) continue;
                q.r.flags = ((qf & ~(uint32_t)F_S)) | (LEAF_NONE << 24);
                q.r.shift = 0;
                c.r.flags = (cf | F_S0) | (LEAF_NONE << 24);
                q.moved = c.moved = true;             // (emitted from the item itself from now on: see below)
            }
        const int nf = (int)fl.size();
        auto same_shape = [&](int x, int y) {
            const Item &X = fl[(size_t)x], &Y = fl[(size_t)y];
            const Rec &a = X.r, &b = Y.r;
            const DRec &da = X.d, &db = Y.d;
            if (X.folded != Y.folded || X.logd != Y.logd) return false;
            if (((a.flags ^ b.flags) & 0xffffffu) || a.kmax != b.kmax || a.t_ix != b.t_ix || a.e_ix != b.e_ix || da.present != db.present)
                return false;                                       // (the leaf byte of a record inside a run is never read)
            if (!X.folded && a.shift != b.shift) return false;      // (a folded record keeps E_b's table offset there)
            if ((X.folded || X.logd) && X.b.logs != Y.b.logs) return false;
            for (int v = 0; v < nv; ++v) {
                if (da.t_ix[v] != db.t_ix[v] || da.e_ix[v] != db.e_ix[v]) return false;
                if ((X.folded || X.logd) && X.b.ix[v] != Y.b.ix[v]) return false;
            }
            return true;
        };
        auto same_tables = [&](int x, int y) {
            const Item &X = fl[(size_t)x], &Y = fl[(size_t)y];
            const Rec &a = X.r, &b = Y.r;
            const DRec &da = X.d, &db = Y.d;
            if (a.t_off != b.t_off || a.e_off != b.e_off) return false;
            for (int v = 0; v < nv; ++v)
                if (da.t_off[v] != db.t_off[v] || da.e_off[v] != db.e_off[v] || X.b.off[v] != Y.b.off[v]) return false;
            return true;
        };
        // maximal runs of >= 4 same-shape records; the kernel handles ONE shape per launch: the one that covers most records
        struct Found { int first, n, code; };
        std::vector<Found> found;
        std::map<int, size_t> covered;
        for (int i = 0; i < nf;) {
            const int code = shape_of(fl[(size_t)i]);
            int n = 1;
            if (code >= 0)
                while (i + n < nf && n < 0x7fff && same_shape(i, i + n)) ++n;
            if (code >= 0 && n >= 4 && K != 64) {
                // 16 / 32 orders: a train that repeats ONE record (an echo train: same tables in every record) stays with the
                // straight-line leaves of packed_deriv_kernel, which beat the folded loop there (20-echo MSE, 1024 x 1024, three
                // variables: 6.4 / 2.9 ms at 32 / 16 orders against 6.9 / 3.2 folded); new tables per repetition (MRF) fold
                bool ident = true;
                for (int j = 1; j < n && ident; ++j) ident = same_tables(i, i + j);
                if (ident) {
                    i += n;
                    continue;
                }
            }
            if (code >= 0 && n >= 4) {
                // the loops of drun_kernel are unrolled four times (the slot bases come round after four records) and finish a
                // run of any length (up to three more records, then the registers are put back in order); the loop at
                // 16 / 32 orders takes a record at a time anyway
                const int take = n;
                found.push_back({i, take, code});
                covered[code] += (size_t)take * (size_t)(fl[(size_t)i].folded ? 2 : 1);   // (weights: original records covered)
            }
            i += n;
        }
        size_t in_runs = 0;
        for (const auto &c : covered)
            if (c.second > in_runs) {
                in_runs = c.second;
                pr.drun_code = c.first;
            }
        DRec dzero;
        memset(&dzero, 0, sizeof(dzero));
        DRecB bzero;
        memset(&bzero, 0, sizeof(bzero));
        size_t next = 0;
        for (int i = 0; i < nf;) {
            while (next < found.size() && (found[next].first < i || found[next].code != pr.drun_code)) ++next;
            if (next < found.size() && found[next].first == i) {
                const int n = found[next].n;
                bool ident = !fl[(size_t)i].folded;
                for (int j = 1; j < n && ident; ++j) ident = same_tables(i, i + j);
                Rec head;
                memset(&head, 0, sizeof(head));
                head.flags = (LEAF_DRUN << 24) | (uint32_t)pr.drun_code | (ident ? (uint32_t)DRUN_IDENT : 0u);
                head.kmax = n << 16;
                pr.drun_inside += n;
                pr.drun_headers += 1;
                pr.drun_ident += ident ? 1 : 0;
                druns.push_back(head);
                ddruns.push_back(dzero);
                bdruns.push_back(bzero);
                for (int j = 0; j < n; ++j) {
                    druns.push_back(fl[(size_t)i + j].r);
                    ddruns.push_back(fl[(size_t)i + j].d);
                    bdruns.push_back(fl[(size_t)i + j].b);
                }
                i += n;
                continue;
            }
            if (fl[(size_t)i].moved) {           // a record whose shift moved (above): the item's own record, unfolded
                druns.push_back(fl[(size_t)i].r);
                ddruns.push_back(drecs[(size_t)fl[(size_t)i].lo]);
                bdruns.push_back(bzero);
                ++i;
                continue;
            }
            for (int j = fl[(size_t)i].lo; j <= fl[(size_t)i].hi; ++j) {   // outside the runs: the records as they were packed
                druns.push_back(recs[(size_t)j]);
                ddruns.push_back(drecs[(size_t)j]);
                bdruns.push_back(bzero);
            }
            ++i;
        }
        if (in_runs * 2 < (size_t)pr.n_rec) {
            druns.clear();
            ddruns.clear();
            bdruns.clear();
        }
        // Fused echoes with logarithmic partials at 64 orders, from equilibrium: the run list cut where the populated orders
        // outgrow 16 and 32 (cf. grow_split) -- drun_kernel walks the first ranges with one and two orders per lane.  Cutting a
        // run ends its owed E_a update there and starts the next part afresh: the same sums in another association (rounding).
        if (K == 64 && !druns.empty() && (pr.drun_code & (int)DRUN_LOGD) && !(pr.drun_code & (int)DRUN_FOLD)) {
            auto shifts_of = [](const Rec &r) { return ((r.flags & F_S0) ? 1 : 0) + ((r.flags & F_S) ? 1 : 0); };
            static const int cap[3] = {15, 31, 1 << 30};
            std::vector<Rec> r2;
            std::vector<DRec> d2v;
            std::vector<DRecB> b2;
            int top = 0, phase = 0, n1 = -1, n2 = -1;
            double work[3] = {0, 0, 0};
            auto next_phase = [&]() {
                if (phase == 0) n1 = (int)r2.size();
                else n2 = (int)r2.size();
                ++phase;
            };
            for (size_t i = 0; i < druns.size();) {
                const Rec &r = druns[i];
                if ((r.flags >> 24) == LEAF_DRUN) {
                    const int count = (int)((uint32_t)r.kmax >> 16);
                    const int d = shifts_of(druns[i + 1]);
                    int done = 0;
                    while (done < count) {
                        int m = d > 0 ? (cap[phase] - top) / d : count - done;
                        m = std::min(m, count - done);
                        if (m <= 0) {
                            next_phase();
                            continue;
                        }
                        Rec h = r;
                        h.kmax = m << 16;
                        r2.push_back(h);
                        d2v.push_back(ddruns[i]);
                        b2.push_back(bdruns[i]);
                        for (int j = 0; j < m; ++j) {
                            r2.push_back(druns[i + 1 + (size_t)(done + j)]);
                            d2v.push_back(ddruns[i + 1 + (size_t)(done + j)]);
                            b2.push_back(bdruns[i + 1 + (size_t)(done + j)]);
                        }
                        work[phase] += m;
                        top += m * d;
                        done += m;
                    }
                    i += 1 + (size_t)count;
                    continue;
                }
                const int d = shifts_of(r);
                while (top + d > cap[phase]) next_phase();
                r2.push_back(r);
                d2v.push_back(ddruns[i]);
                b2.push_back(bdruns[i]);
                work[phase] += 1;
                top += d;
                ++i;
            }
            if (n1 < 0) n1 = (int)r2.size();
            if (n2 < 0) n2 = (int)r2.size();
            const double all = work[0] + work[1] + work[2];
            if (all > 0 && (work[0] + work[1]) / all >= 0.1) {
                druns.swap(r2);
                ddruns.swap(d2v);
                bdruns.swap(b2);
                pr.dgrow1 = std::min(n1, n2);
                pr.dgrow2 = n2;
                if (knobs().grow_min >= 2) pr.dgrow1 = 0;
            }
        }
    }
    if (pr.n_rec) {
        Rec pad;  // the kernels fetch up to three records past the end (rows_kernel may run the first as a no-op)
        memset(&pad, 0, sizeof(pad));
        recs.push_back(pad);
        recs.push_back(pad);
        recs.push_back(pad);
        epgx_ctx *ctx = pl->ctx;
        hipError_t e = dev_alloc(ctx, (void **)&pr.d_recs, sizeof(Rec) * recs.size());
        if (e == hipSuccess)
            e = hipMemcpyAsync(pr.d_recs, recs.data(), sizeof(Rec) * recs.size(), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess && !drecs.empty()) {
            DRec dpad;   // rows_deriv_kernel fetches up to three entries past the end, like the records
            memset(&dpad, 0, sizeof(dpad));
            drecs.push_back(dpad);
            drecs.push_back(dpad);
            drecs.push_back(dpad);
            e = dev_alloc(ctx, (void **)&pr.d_drecs, sizeof(DRec) * drecs.size());
            if (e == hipSuccess)
                e = hipMemcpyAsync(pr.d_drecs, drecs.data(), sizeof(DRec) * drecs.size(), hipMemcpyHostToDevice,
                                   ctx->stream);
        }
        if (e == hipSuccess && !druns.empty()) {
            pr.n_druns = (int)druns.size();
            DRec dpad;
            memset(&dpad, 0, sizeof(dpad));
            DRecB bpad;
            memset(&bpad, 0, sizeof(bpad));
            for (int k = 0; k < 3; ++k) {
                druns.push_back(pad);
                ddruns.push_back(dpad);
                bdruns.push_back(bpad);
            }
            e = dev_alloc(ctx, (void **)&pr.d_druns, sizeof(Rec) * druns.size());
            if (e == hipSuccess) e = dev_alloc(ctx, (void **)&pr.d_ddruns, sizeof(DRec) * ddruns.size());
            if (e == hipSuccess) e = dev_alloc(ctx, (void **)&pr.d_bdruns, sizeof(DRecB) * bdruns.size());
            if (e == hipSuccess) e = hipMemcpyAsync(pr.d_druns, druns.data(), sizeof(Rec) * druns.size(), hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(pr.d_ddruns, ddruns.data(), sizeof(DRec) * ddruns.size(), hipMemcpyHostToDevice, ctx->stream);
            if (e == hipSuccess) e = hipMemcpyAsync(pr.d_bdruns, bdruns.data(), sizeof(DRecB) * bdruns.size(), hipMemcpyHostToDevice, ctx->stream);
        }
        if (e == hipSuccess && !runs.empty()) {
            pr.n_runs = (int)runs.size();
            runs.push_back(pad);
            runs.push_back(pad);
            e = dev_alloc(ctx, (void **)&pr.d_runs, sizeof(Rec) * runs.size());
            if (e == hipSuccess)
                e = hipMemcpyAsync(pr.d_runs, runs.data(), sizeof(Rec) * runs.size(), hipMemcpyHostToDevice, ctx->stream);
        }
        if (e == hipSuccess && !grow.empty()) {
            pr.n_grow = (int)grow.size();
            grow.push_back(pad);
            grow.push_back(pad);
            e = dev_alloc(ctx, (void **)&pr.d_grow, sizeof(Rec) * grow.size());
            if (e == hipSuccess)
                e = hipMemcpyAsync(pr.d_grow, grow.data(), sizeof(Rec) * grow.size(), hipMemcpyHostToDevice, ctx->stream);
        }
        {   // `recs` / `drecs` / `runs` are locals: nothing may still be reading them when this returns
            const hipError_t es = hipStreamSynchronize(ctx->stream);
            if (e == hipSuccess) e = es;
        }
        if (e != hipSuccess) {
            dev_free(ctx, pr.d_recs);
            dev_free(ctx, pr.d_drecs);
            dev_free(ctx, pr.d_runs);
            dev_free(ctx, pr.d_grow);
            dev_free(ctx, pr.d_druns);
            dev_free(ctx, pr.d_ddruns);
            dev_free(ctx, pr.d_bdruns);
            return fail(EPGX_ERR_HIP, "epgx_run: uploading records failed: %s", hipGetErrorString(e));
        }
    }
    if (pl->packed.size() >= 4096) {  // bound the cache (streams of thousands of distinct ranges)
        for (auto &old : pl->packed) {
            dev_free(pl->ctx, old.d_recs);
            dev_free(pl->ctx, old.d_drecs);
            dev_free(pl->ctx, old.d_runs);
            dev_free(pl->ctx, old.d_grow);
            dev_free(pl->ctx, old.d_druns);
            dev_free(pl->ctx, old.d_ddruns);
            dev_free(pl->ctx, old.d_bdruns);
        }
        pl->packed.clear();
    }
    pl->packed.push_back(pr);
    *out = &pl->packed.back();
    return EPGX_OK;
}

// ------------------------------------------------------------------------------ kernel selection
namespace {
enum Family {
    FAM_RUN,           // run_kernel<M, NSP, HAS_IN>: one wavefront per voxel, K / 64 orders per lane (any operator; state in / out)
    FAM_RUN_CONTIG,    // run_contig_kernel: K = 128 .. 1024 without a state output, K / 64 consecutive orders per lane
    FAM_RUN_CONTIG_GROW, // run_contig_grow_kernel<M, NSP>: the same from equilibrium in phases of 1, 2, 4 .. orders per lane while the state matrix grows
    FAM_RUN_SPLIT,     // run_split_kernel<4, ..>: K = 2048 from equilibrium, four wavefronts per voxel (behind a run_kernel<8, ..> leg where that pays)
    FAM_ROWS,          // rows_kernel<NSP, R, RUNS>: four voxels per wavefront, R = K / 16 orders per lane, state-resident
    FAM_ROWS_GROW,     // rows_grow_kernel<NSP>: the same walked in phases of R = 1, 2, 4 while the state matrix grows (K = 64)
    FAM_DERIV,         // deriv_kernel<M, NSP, V>: one wavefront per voxel, 1 + V states
    FAM_PACKED_DERIV,  // packed_deriv_kernel<NSP, V, KP>: 16 / 32 orders, four / two voxels per wavefront, 1 + V states
    FAM_ROWS_DERIV,    // rows_deriv_kernel<NSP, 4, V>: the rows layout with one or two derivative states
    FAM_DRUN,          // drun_kernel<NSP, V, SHAPE, V0>: rotating order slots, runs of fused / folded records, 1 + V states
    FAM_PACKED_DFOLD   // packed_dfold_kernel: 16 / 32 orders, repetitions folded at run time, 1 + V states
};
struct Choice {
    Family family = FAM_RUN;
    bool runs = false;      // rows kernels: the run-length folded record list
    bool split_grow = false; // K = 2048: two legs -- run_kernel<8, ..> up to 512 populated orders, then run_split_kernel from its state
    bool split3 = false;    // drun_kernel: three derivative states of folded runs in two launches (V0 = 2, then V = 2)
    char name[128] = "";
    const char *why = "";
};
}  // namespace

// THE place where a launch gets its kernel: operators [op_begin, op_end) of a plan at capacity K, with / without a state input
// and output.  Everything the decision depends on is an argument or a field of the plan / its packed range -- no state of the
// context, no launch size -- so epgx_kernel_for can answer without launching (tests pin the kernel of every BASELINE config).
static int choose_kernel(const epgx_plan *pl, const PackedRange *pr, int op_begin, int op_end, int K, bool has_in, bool has_out, Choice *c) {
    const Knobs &kn = knobs();
    const bool packed16 = K == 16 || K == 32, wide = K == 2048 && !has_in && !has_out;
    bool has_general = false, has_nd = false;   // general 3x3 matrices; diffusion / gather shifts
    for (int i = op_begin; i < op_end; ++i) {
        const int oc = pl->ops[i].opcode;
        has_general = has_general || oc == EPGX_OP_MAT || oc == EPGX_OP_MAT0;
        has_nd = has_nd || oc == EPGX_OP_D || oc == EPGX_OP_GS;
    }
    // (the rows kernels and packed_deriv_kernel address the pool through a buffer resource of 2 GiB)
    const bool pool_in_reach = (pl->n_pool + 64 + pl->n_log) * (int64_t)sizeof(double) <= 0x7fffffff;
    if (packed16 && !pool_in_reach)
        return fail(EPGX_ERR_UNSUPPORTED, "epgx_run: K = 16 / 32 need a coefficient pool below 2 GiB (use K = 64)");
    if (wide && (pr->use_lds || pl->n_vars > 0))
        return fail(EPGX_ERR_UNSUPPORTED, "epgx_run: K = 2048 handles rotations, relaxation, shifts by +-1 and probes only (no derivative states)");
    if (packed16 && pr->big_shift) return fail(EPGX_ERR_UNSUPPORTED, "epgx_run: K = 16 / 32 handle shifts by +-1 (and, at K = 16, gather shifts) only");
    if (packed16 && pl->n_vars > 0 && has_in) return fail(EPGX_ERR_UNSUPPORTED, "epgx_run: K = 16 / 32 derivative plans start from equilibrium");
    if (packed16 && pl->n_vars > 0 && pr->use_lds) return fail(EPGX_ERR_UNSUPPORTED, "epgx_run: K = 16 / 32 derivative plans handle shifts by +-1 only");
    const int nsp = pl->n_spaces <= 2 ? pl->n_spaces : 4, V = pl->n_vars;
    const bool plain_ops = !has_general && !has_nd && !pr->use_lds;   // rotations, relaxation, shifts by +-1, probes, SPOILER / RESET / PD
    if (V > 0) {
        if (has_out) return fail(EPGX_ERR_UNSUPPORTED, "epgx_run: derivative plans run state-resident (out = NULL)");
        if (K > 1024) return fail(EPGX_ERR_UNSUPPORTED, "epgx_run: derivative plans support K <= 1024, got %d", K);
        if (K == 1024 && V > 1)
            return fail(EPGX_ERR_UNSUPPORTED, "epgx_run: at K = 1024 a launch carries ONE derivative state (plan has %d variables: one plan per variable)", V);
        const bool resident64 = K == 64 && !has_in && plain_ops && pool_in_reach;
        if (packed16 && kn.drun && pr->d_druns && pr->d_bdruns && !has_in && !pr->use_lds && pool_in_reach) {
            c->family = FAM_PACKED_DFOLD;
            c->why = "16 / 32 orders, mostly runs of repetitions folded at run time";
            snprintf(c->name, sizeof(c->name), "packed_dfold_kernel<%d, %d>", V, K);
        } else if (kn.drun && pr->d_druns && resident64) {
            c->family = FAM_DRUN;
            c->split3 = V == 3 && (pr->drun_code & (int)DRUN_FOLD) && EPGX_DF3_SPLIT;
            c->why = (pr->drun_code & (int)DRUN_FOLD)   ? "64 orders, mostly runs of repetitions folded at run time: rotating order slots"
                     : (pr->drun_code & (int)DRUN_LOGD) ? "64 orders, mostly runs of fused echoes with logarithmic relaxation partials: rotating order slots"
                                                        : "64 orders, mostly runs of fused echoes: rotating order slots";
            // (runs with logarithmic partials exist for four index spaces only; the others for one and four: epgx_launch_drun)
            const int knsp = ((pr->drun_code & 384) || pl->n_spaces > 1) ? 4 : 1;
            if (c->split3) snprintf(c->name, sizeof(c->name), "drun_kernel<%d, 1, %d, 2> + drun_kernel<%d, 2, %d, 0>", knsp, pr->drun_code, knsp, pr->drun_code);
            else snprintf(c->name, sizeof(c->name), "drun_kernel<%d, %d, %d, 0>", knsp, V, pr->drun_code);
        } else if (kn.rows_deriv && (V == 1 || (V == 2 && kn.rows_deriv2)) && resident64) {
            c->family = FAM_ROWS_DERIV;
            c->why = "64 orders from equilibrium, one or two derivative states: four voxels per wavefront";
            snprintf(c->name, sizeof(c->name), "rows_deriv_kernel<%d, 4, %d>", nsp, V);
        } else if (packed16) {
            c->family = FAM_PACKED_DERIV;
            c->why = "16 / 32 orders with derivative states";
            snprintf(c->name, sizeof(c->name), "packed_deriv_kernel<%d, %d, %d>", nsp, V, K);
        } else {
            c->family = FAM_DERIV;
            c->why = "derivative states, one wavefront per voxel";
            const bool contig_orders = kn.contig && K >= 128 && !pr->use_lds && !has_nd && !(V == 3 && (K == 256 || K == 512));
            snprintf(c->name, sizeof(c->name), "deriv_kernel<%d, %d, %d%s>", K / 64, nsp, V, contig_orders ? ", true" : "");
        }
        return EPGX_OK;
    }
    // K = 256 .. 1024 (EPGX_CGROW=2: from 128) from equilibrium with a good share of the records while the state matrix is short: phases of 1, 2, 4 .. orders per lane
    // (at K = 128 the four-voxels-per-wavefront kernel with 8 orders per lane is the alternative: the phases win while the train mostly
    // runs below 64 orders -- 40 echoes 1.08 against 1.17 ms, 63 echoes 1.82 against 1.78; EPGX_CGROW=2: always)
    const bool cgrow = kn.contig && kn.cgrow && K >= 128 && K <= 1024 && !has_in && !has_out && !pr->use_lds && !has_nd &&
                       pr->cgrow_share >= ((K == 128 && kn.cgrow < 2) ? std::max(kn.grow_share, 0.6) : kn.grow_share);
    if (cgrow) {
        c->family = FAM_RUN_CONTIG_GROW;
        c->why = "from equilibrium, a good share of the records while the state matrix is short: K / 64 consecutive orders per lane reached in phases of 1, 2, 4 ..";
        snprintf(c->name, sizeof(c->name), "run_contig_grow_kernel<%d, %d>", K / 64, nsp);
        return EPGX_OK;
    }
    // four voxels per wavefront, K / 16 orders per lane: always at 16 / 32 orders; at 64 / 128 state-resident launches of plain operators
    const bool rows = packed16 || (kn.rows && (K == 64 || K == 128) && !has_in && !has_out && plain_ops && pool_in_reach);
    if (rows) {
        c->runs = kn.runs && pr->d_runs && K <= 64;
        if (c->runs && K == 64 && kn.grow && pr->d_grow) {
            c->family = FAM_ROWS_GROW;
            c->why = "64 orders from equilibrium, a good share of the records while the state matrix is short: phases of 1 / 2 / 4 orders per lane";
            snprintf(c->name, sizeof(c->name), "rows_grow_kernel<%d>", nsp);
        } else {
            c->family = FAM_ROWS;
            c->why = "state-resident, four voxels per wavefront";
            snprintf(c->name, sizeof(c->name), "rows_kernel<%d, %d, %s>", nsp, K / 16, (c->runs && K <= 64) ? "true" : "false");
        }
        return EPGX_OK;
    }
    // one wavefront per voxel (four at K = 2048).  Launches without a state output at
    // K >= 128 are free to choose the order layout: a lane then holds K / 64 consecutive orders and a shift by one costs 8 DPP
    // moves instead of 16 K / 64 moves and selects (epgx_split.hip; the same bits).  Not with shifts by |n| >= 2, gather shifts or diffusion.
    const bool free_layout = !has_out && !pr->use_lds && !has_nd;
    if (K == 2048) {
        c->family = FAM_RUN_SPLIT;
        c->split_grow = kn.split_grow && pr->cgrow_share >= kn.grow_share && pr->cgrow[3] > 0;
        c->why = c->split_grow ? "2048 orders from equilibrium: one wavefront per voxel while at most 512 orders hold anything, then up to four (the state crosses HBM once)"
                               : "2048 orders from equilibrium: four wavefronts per voxel";
        if (c->split_grow) snprintf(c->name, sizeof(c->name), "run_kernel<8, %d, false> + run_split_kernel<4, %d, true>", nsp, nsp);
        else snprintf(c->name, sizeof(c->name), "run_split_kernel<4, %d, false>", nsp);
    } else if (kn.contig && K >= 128 && K <= 1024 && free_layout) {
        c->family = FAM_RUN_CONTIG;
        c->why = "no state output: K / 64 consecutive orders per lane";
        snprintf(c->name, sizeof(c->name), "run_contig_kernel<%d, %d, %s>", K / 64, nsp, has_in ? "true" : "false");
    } else {
        c->family = FAM_RUN;
        c->why = "one wavefront per voxel, state through HBM or operators the other kernels do not take";
        snprintf(c->name, sizeof(c->name), "run_kernel<%d, %d, %s>", K / 64, nsp, has_in ? "true" : "false");
    }
    return EPGX_OK;
}

// The kernel instantiations live in separate translation units (epgx_inst.hip compiled once per
// M, epgx_deriv.hip ...) so that they build in parallel; see epgx_launch.h.
// epgx_run, and (name_out != NULL) epgx_kernel_for: the same checks and the same decision, no launch
static int run_or_name(epgx_ctx *ctx, const epgx_plan *plan_c, int32_t op_begin, int32_t op_end, int64_t vox0, int64_t nvox,
                       const epgx_state *in, epgx_state *out, int32_t K, void *signal, int64_t signal_ld, int64_t signal_col0,
                       char *name_out, int64_t name_bytes) {
    epgx_plan *pl = const_cast<epgx_plan *>(plan_c);
    if (!ctx || !pl) return fail(EPGX_ERR_INVALID, "epgx_run: NULL argument");
    if (pl->ctx != ctx) return fail(EPGX_ERR_INVALID, "epgx_run: plan belongs to another context");
    const int n_ops = (int)pl->ops.size();
    if (op_begin < 0 || op_end > n_ops || op_begin > op_end)
        return fail(EPGX_ERR_INVALID, "epgx_run: operator range [%d,%d) outside [0,%d)", op_begin, op_end, n_ops);
    if (nvox > (int64_t)4 * 0x7fffffff) return fail(EPGX_ERR_UNSUPPORTED, "epgx_run: more than 2^33 voxels in one launch");
    if (nvox < 0 || vox0 < 0 || vox0 + nvox > pl->nvox_total)
        return fail(EPGX_ERR_INVALID, "epgx_run: voxel range [%lld,%lld) outside the grid (%lld voxels)",
                    (long long)vox0, (long long)(vox0 + nvox), (long long)pl->nvox_total);
    if (in && in->ctx != ctx) return fail(EPGX_ERR_INVALID, "epgx_run: `in` belongs to another context");
    if (out && out->ctx != ctx) return fail(EPGX_ERR_INVALID, "epgx_run: `out` belongs to another context");
    if (in) K = in->K;
    if (out) {
        if (in && out->K != in->K)
            return fail(EPGX_ERR_INVALID, "epgx_run: in/out capacities differ (%d vs %d)", in->K, out->K);
        K = out->K;
    }
    // K = 16: four voxels per wavefront (epgx_packed_kernels.hip.h), state-resident launches only
    const bool packed16 = (K == 16 || K == 32);
    if (packed16 && (in || out))
        return fail(EPGX_ERR_UNSUPPORTED, "epgx_run: K = 16 / 32 keep no state in HBM: in and out must be NULL");
    // K = 2048: state-resident from equilibrium only (four wavefronts per voxel; a state matrix of that size has no HBM form)
    const bool wide = K == 2048 && !in && !out;
    if (!packed16 && !wide && !supported_K(K))
        return fail(EPGX_ERR_UNSUPPORTED, "epgx_run: K=%d not one of 16, 32, 2048 (state-resident only), 64, 128, 256, 512, 1024", K);
    if (in && in->nvox != nvox)
        return fail(EPGX_ERR_INVALID, "epgx_run: `in` holds %lld voxels, range has %lld", (long long)in->nvox,
                    (long long)nvox);
    if (out && out->nvox != nvox)
        return fail(EPGX_ERR_INVALID, "epgx_run: `out` holds %lld voxels, range has %lld", (long long)out->nvox,
                    (long long)nvox);
    if (nvox == 0 || op_begin == op_end) {
        if (name_out) snprintf(name_out, (size_t)name_bytes, "none");
        return EPGX_OK;
    }

    for (int i = op_begin; i < op_end; ++i) {
        const epgx_op &op = pl->ops[i];
        if (op.opcode == EPGX_OP_S && std::abs(op.ia) >= K)
            return fail(EPGX_ERR_INVALID, "epgx_run: operator %d shifts by %d, capacity K=%d", i, op.ia, K);
        if (packed16 && (op.opcode == EPGX_OP_MAT || op.opcode == EPGX_OP_MAT0))
            return fail(EPGX_ERR_UNSUPPORTED, "epgx_run: K = 16 / 32 do not handle general matrices (operator %d)", i);
        if (K == 32 && (op.opcode == EPGX_OP_D || op.opcode == EPGX_OP_GS))
            return fail(EPGX_ERR_UNSUPPORTED, "epgx_run: diffusion and gather shifts run with K = 16 or K >= 64 (operator %d)", i);
        if (op.opcode == EPGX_OP_D && op.ncoef != 3 * K)
            return fail(EPGX_ERR_INVALID, "epgx_run: operator %d: D table has %d doubles per entry, need 3*K=%d", i,
                        op.ncoef, 3 * K);
        if (op.opcode == EPGX_OP_GS) {
            if (2 * op.ncoef != 3 * K)
                return fail(EPGX_ERR_INVALID, "epgx_run: operator %d: gather table has %d entries, need 3*K=%d", i,
                            2 * op.ncoef, 3 * K);
            for (int32_t v : pl->gather_tables[i])
                if (v != -1 && ((v & ~(1 << 30)) < 0 || (v & ~(1 << 30)) >= K))
                    return fail(EPGX_ERR_INVALID, "epgx_run: operator %d: gather index %d outside [0,%d)", i, v, K);
        }
    }
    if (int rc = set_device(ctx)) return rc;
    // from here to the launch the plan's caches (packed records, table indices of the voxel range) are read and
    // possibly rebuilt: one host thread at a time per plan (the launch itself is asynchronous)
    std::lock_guard<std::mutex> plan_guard(pl->cache_lock);
    const PackedRange *pr = nullptr;
    if (int rc = get_packed(pl, op_begin, op_end, K, &pr)) return rc;
    if (pr->has_adc && !name_out) {
        if (!signal) return fail(EPGX_ERR_INVALID, "epgx_run: range contains an ADC but signal is NULL");
        if (signal_col0 < 0 || signal_col0 + nvox > signal_ld)
            return fail(EPGX_ERR_INVALID, "epgx_run: signal columns [%lld,%lld) exceed signal_ld=%lld",
                        (long long)signal_col0, (long long)(signal_col0 + nvox), (long long)signal_ld);
    }
    if (pr->n_rec == 0) {  // nothing but NOPs: only a state copy may be needed
        if (name_out) {
            snprintf(name_out, (size_t)name_bytes, "none");
            return EPGX_OK;
        }
        if (out && in && out != in) return epgx_state_copy(out, in);
        return EPGX_OK;
    }
    Choice c;
    if (int rc = choose_kernel(pl, pr, op_begin, op_end, K, in != nullptr, out != nullptr, &c)) return rc;
    if (name_out) {   // epgx_kernel_for: the decision, no launch
        snprintf(name_out, (size_t)name_bytes, "%s", c.name);
        return EPGX_OK;
    }
    if (tracing()) fprintf(stderr, "[epgx] run: %s -- %s\n", c.name, c.why);
    if (int rc = ensure_vidx(pl, vox0, nvox)) return rc;

    hipError_t e = hipSuccess;
    if (pl->n_vars > 0) {
        DerivArgs da;
        memset(&da, 0, sizeof(da));
        da.nvox = nvox;
        da.in = in ? in->data : nullptr;
        da.dens_in = in ? in->dens : nullptr;
        da.recs = pr->d_recs;
        da.drecs = pr->d_drecs;
        da.coef = pl->d_coef;
        da.signal = signal ? (d2 *)signal + signal_col0 : nullptr;
        da.signal_ld = signal_ld;
        da.t.vidx = pl->d_vidx;
        da.t.vidx_ld = pl->vidx_nvox;
        da.t.vox0 = vox0;
        da.t.n_rec = pr->n_rec;
        da.t.dense_spaces = pl->dense_spaces;
        da.t.use_lds = pr->use_lds ? ((pr->has_gs && K < 1024) ? 3 : 2) : 0;      // (gather shifts at 1024 orders stage Z behind F: gather_shift)
        da.through_plain = (pl->deriv_flags & EPGX_DERIV_THROUGH_PLAIN_OPS) ? 1 : 0;
        {   // deriv_kernel at K >= 128: consecutive orders per lane unless the range shifts by |n| >= 2, gathers or diffuses (EPGX_CONTIG=0: never)
            bool nd = false;
            for (int i = op_begin; i < op_end; ++i) nd = nd || pl->ops[i].opcode == EPGX_OP_D || pl->ops[i].opcode == EPGX_OP_GS;
            // (not with three derivative states at 256 / 512 orders: 282 VGPRs / 585 spill instructions there against 249 / 9 lane-strided,
            // measured 111 against 75 ms and 313 against 174 ms)
            da.contig = (knobs().contig && K >= 128 && !pr->use_lds && !nd && !(pl->n_vars == 3 && (K == 256 || K == 512))) ? 1 : 0;
        }
        if (c.family == FAM_DRUN && knobs().grow) {   // (0, 0: four orders per lane throughout; EPGX_GROW=0, measurements)
            da.grow1 = pr->dgrow1;
            da.grow2 = pr->dgrow2;
        }
        if (c.family == FAM_PACKED_DFOLD || c.family == FAM_DRUN) {   // the records with run headers (and E_b's logarithmic partials)
            da.recs = pr->d_druns;
            da.drecs = pr->d_ddruns;
            da.drecs_b = pr->d_bdruns;
            da.t.n_rec = pr->n_druns;
            // (these kernels exist for 1 and 4 index spaces: a space the plan does not have counts as dense -- no index row is read for it)
            da.t.dense_spaces |= 0xfu & ~((1u << pl->n_spaces) - 1u);
            if (tracing())
                fprintf(stderr, "[epgx] run: %d records with headers (%d unfolded); %d runs (%d of one repeated record) hold %d records; [0, %d) "
                                "with one order per lane, [%d, %d) with two\n",
                        pr->n_druns, pr->n_rec, pr->drun_headers, pr->drun_ident, pr->drun_inside, da.grow1, da.grow1, da.grow2);
        }
        switch (c.family) {
        case FAM_PACKED_DFOLD: e = epgx_launch_packed_dfold(ctx->stream, da, K, pl->n_vars); break;
        case FAM_DRUN:
            if (c.split3) {
                // three derivative states of folded runs: the last variable alone (rows shifted by two), then the first two over it
                DerivArgs last = da;
                last.signal = da.signal ? da.signal + 2 * da.signal_ld : nullptr;
                e = epgx_launch_drun(ctx->stream, last, K, pl->n_spaces, 1, pr->drun_code | (int)DRUN_LAST);
                if (e == hipSuccess) e = epgx_launch_drun(ctx->stream, da, K, pl->n_spaces, 2, pr->drun_code);
            } else
                e = epgx_launch_drun(ctx->stream, da, K, pl->n_spaces, pl->n_vars, pr->drun_code);
            break;
        case FAM_ROWS_DERIV:
            if (pl->n_vars == 2) {
                switch (pl->n_spaces) {
                case 0: e = epgx_launch_rows_deriv_v2_nsp0(ctx->stream, da, K); break;
                case 1: e = epgx_launch_rows_deriv_v2_nsp1(ctx->stream, da, K); break;
                case 2: e = epgx_launch_rows_deriv_v2_nsp2(ctx->stream, da, K); break;
                default: e = epgx_launch_rows_deriv_v2_nsp4(ctx->stream, da, K); break;
                }
            } else {
                switch (pl->n_spaces) {
                case 0: e = epgx_launch_rows_deriv_nsp0(ctx->stream, da, K); break;
                case 1: e = epgx_launch_rows_deriv_nsp1(ctx->stream, da, K); break;
                case 2: e = epgx_launch_rows_deriv_nsp2(ctx->stream, da, K); break;
                default: e = epgx_launch_rows_deriv_nsp4(ctx->stream, da, K); break;
                }
            }
            break;
        case FAM_PACKED_DERIV: e = epgx_launch_packed_deriv(ctx->stream, da, K, pl->n_spaces, pl->n_vars); break;
        default: e = epgx_launch_deriv(ctx->stream, da, K, pl->n_spaces, pl->n_vars); break;
        }
        if (e != hipSuccess) return fail(EPGX_ERR_HIP, "epgx_run: launch failed: %s", hipGetErrorString(e));
        return EPGX_OK;
    }
    RunArgs a;
    memset(&a, 0, sizeof(a));
    a.recs = pr->d_recs;
    a.t.n_rec = pr->n_rec;
    a.coef = pl->d_coef;
    a.t.vidx = pl->d_vidx;
    a.t.vidx_ld = pl->vidx_nvox;
    a.nvox = nvox;
    a.in = in ? in->data : nullptr;
    a.out = out ? out->data : nullptr;
    a.dens_in = in ? in->dens : nullptr;
    a.t.dens_out = out ? out->dens : nullptr;
    a.signal = signal ? (d2 *)signal + signal_col0 : nullptr;
    a.signal_ld = signal_ld;
    a.t.use_lds = pr->use_lds ? ((pr->has_gs && K < 1024) ? 3 : 2) : 0;      // (gather shifts at 1024 orders stage Z behind F: gather_shift)
    a.t.seq_slots = pr->seq_slots ? 1 : 0;
    a.t.first_slot = pr->first_slot;
    a.t.vox0 = vox0;
    a.t.dense_spaces = pl->dense_spaces;
    a.t.write_dens = (pr->has_pd || out != in) ? 1 : 0;
    // long record lists over per-voxel tables: prefetch (EPGX_PREFETCH=0 disables, for measurements)
    a.t.prefetch = (knobs().prefetch && !in && pr->n_rec >= 4) ? pr->pf_count : 0;
    // a wave of rows_kernel takes ONE voxel group (rounds 1 and 2 gave it four on big grids; with today's kernels one is faster on
    // every workload measured: MRF C3 44.5 / 45.8 ms, MRF with max_nstate = 10 at 16 orders 27.0 / 28.5 ms, spoiled gradient echo
    // 14.5 / 14.7 ms); a wave of rows_grow_kernel two (C2-L 0.672 -> 0.634 ms per launch; four: the same).  EPGX_GPW=n overrides
    a.groups_per_wave = c.family == FAM_ROWS_GROW ? 2 : 1;
    if (c.runs) {   // run-length folded records (get_packed)
        a.recs = pr->d_runs;
        a.t.n_rec = pr->n_runs;
    }
    switch (c.family) {
    case FAM_ROWS_GROW:
        a.recs = pr->d_grow;
        a.t.n_rec = pr->n_grow;
        if (tracing())
            fprintf(stderr, "[epgx] run: %d records: [0, %d) at 16 orders per voxel, [%d, %d) at 32, the rest at 64\n", pr->n_grow, pr->grow1,
                    pr->grow1, pr->grow2);
        switch (pl->n_spaces) {
        case 0: e = epgx_launch_rows_grow_nsp0(ctx->stream, a, pr->grow1, pr->grow2); break;
        case 1: e = epgx_launch_rows_grow_nsp1(ctx->stream, a, pr->grow1, pr->grow2); break;
        case 2: e = epgx_launch_rows_grow_nsp2(ctx->stream, a, pr->grow1, pr->grow2); break;
        default: e = epgx_launch_rows_grow_nsp4(ctx->stream, a, pr->grow1, pr->grow2); break;
        }
        break;
    case FAM_ROWS:
        switch (K / 16) {
        case 1: e = epgx_launch_rows_r1(ctx->stream, a, pl->n_spaces, c.runs); break;
        case 2: e = epgx_launch_rows_r2(ctx->stream, a, pl->n_spaces, c.runs); break;
        case 4: e = epgx_launch_rows_r4(ctx->stream, a, pl->n_spaces, c.runs); break;
        default: e = epgx_launch_rows_r8(ctx->stream, a, pl->n_spaces, c.runs); break;
        }
        break;
    case FAM_RUN_SPLIT:
        if (!c.split_grow) {
            e = epgx_launch_run_split2048(ctx->stream, a, pl->n_spaces, nullptr, nullptr, 0, 0, 0, 0);
            break;
        }
        {
            // two legs per slab of voxels: records [0, j1) on one wavefront per voxel at 512 orders (run_kernel<8, ..>: the growing kernel
            // cannot hand its state on -- epgx_cgrow.hip), its state [3][512] + density through a scratch buffer (24 KiB per voxel: slabs
            // of at most 8 GiB), then the records [j1, n_rec) on four wavefronts per voxel
            const int j1 = pr->cgrow[3], j2 = pr->cgrow[4], j3 = pr->cgrow[5];
            const int64_t per_voxel = (int64_t)3 * 512 * sizeof(d2) + sizeof(double);
            int64_t slab = std::min<int64_t>((nvox + 3) & ~(int64_t)3, std::max<int64_t>(4, (((int64_t)8 << 30) / per_voxel) & ~(int64_t)3));
            if (knobs().slab_voxels > 0) slab = std::min<int64_t>(slab, (knobs().slab_voxels + 3) & ~3);
            void *scratch = nullptr;
            e = dev_alloc(ctx, &scratch, (size_t)(slab * per_voxel));
            if (e != hipSuccess) break;
            d2 *st = (d2 *)scratch;
            double *dn = (double *)((char *)scratch + slab * 3 * 512 * sizeof(d2));
            if (tracing())
                fprintf(stderr, "[epgx] run: %d records: [0, %d) on one wavefront per voxel, the rest on four (parts 2 and 3 join at %d and %d); slabs of "
                                "%lld voxels\n", pr->n_rec, j1, j2, j3, (long long)slab);
            for (int64_t c0 = 0; c0 < nvox && e == hipSuccess; c0 += slab) {
                RunArgs s = a;
                s.nvox = std::min(slab, nvox - c0);
                s.t.vox0 = vox0 + c0;
                if (s.t.vidx) s.t.vidx += c0;
                if (s.signal) s.signal += c0;
                if (s.dens_in) s.dens_in += c0;
                RunArgs leg = s;      // the first leg: the per-timestep kernel at 512 orders over the records [0, j1), its state to the scratch buffer
                leg.t.n_rec = j1;
                leg.t.prefetch = std::min(leg.t.prefetch, j1);
                leg.out = st;
                leg.t.dens_out = dn;
                leg.t.write_dens = 1;
                e = epgx_launch_run_m8(ctx->stream, leg, pl->n_spaces);
                if (e == hipSuccess && j1 < pr->n_rec)
                    e = epgx_launch_run_split2048(ctx->stream, s, pl->n_spaces, st, dn, j1, j2, j3, pr->first_slot + pr->cgrow_adc3);
            }
            dev_free(ctx, scratch);
        }
        break;
    case FAM_RUN_CONTIG: e = epgx_launch_run_contig(ctx->stream, a, K, pl->n_spaces); break;
    case FAM_RUN_CONTIG_GROW:
        if (tracing())
            fprintf(stderr, "[epgx] run: %d records: [0, %d) at 64 orders per voxel, [%d, %d) at 128, [%d, %d) at 256, [%d, %d) at 512, the rest at %d\n",
                    pr->n_rec, pr->cgrow[0], pr->cgrow[0], pr->cgrow[1], pr->cgrow[1], pr->cgrow[2], pr->cgrow[2], pr->cgrow[3], K);
        {
            const int g4[4] = {pr->cgrow[0], pr->cgrow[1], pr->cgrow[2], pr->cgrow[3]};
            e = epgx_launch_run_contig_grow(ctx->stream, a, K, pl->n_spaces, g4);
        }
        break;
    default:
        switch (K / 64) {
        case 1: e = epgx_launch_run_m1(ctx->stream, a, pl->n_spaces); break;
        case 2: e = epgx_launch_run_m2(ctx->stream, a, pl->n_spaces); break;
        case 4: e = epgx_launch_run_m4(ctx->stream, a, pl->n_spaces); break;
        case 8: e = epgx_launch_run_m8(ctx->stream, a, pl->n_spaces); break;
        default: e = epgx_launch_run_m16(ctx->stream, a, pl->n_spaces); break;
        }
        break;
    }
    if (e != hipSuccess) return fail(EPGX_ERR_HIP, "epgx_run: launch failed: %s", hipGetErrorString(e));
    return EPGX_OK;
}

extern "C" int epgx_run(epgx_ctx *ctx, const epgx_plan *plan, int32_t op_begin, int32_t op_end, int64_t vox0, int64_t nvox,
                        const epgx_state *in, epgx_state *out, int32_t K, void *signal, int64_t signal_ld, int64_t signal_col0) {
    return run_or_name(ctx, plan, op_begin, op_end, vox0, nvox, in, out, K, signal, signal_ld, signal_col0, nullptr, 0);
}

extern "C" int epgx_kernel_for(epgx_ctx *ctx, const epgx_plan *plan, int32_t op_begin, int32_t op_end, int32_t K, const epgx_state *in,
                               epgx_state *out, char *name_out, int64_t name_bytes) {
    if (!name_out || name_bytes < 2) return fail(EPGX_ERR_INVALID, "epgx_kernel_for: no room for the name");
    if (!ctx || !plan) return fail(EPGX_ERR_INVALID, "epgx_kernel_for: NULL argument");
    const int64_t nvox = in ? in->nvox : (out ? out->nvox : plan->nvox_total);
    return run_or_name(ctx, plan, op_begin, op_end, 0, nvox, in, out, K, nullptr, 0, 0, name_out, name_bytes);
}

// ------------------------------------------------------------------------------ RCCL (loaded on first use)
// One gather of signal slabs is the only inter-GPU traffic of the path.  librccl.so.1 is half a
// gigabyte of device code: it is dlopen'ed when the first communicator is asked for, never at library load.
// (When the process has already loaded an RCCL of the same soname -- PyTorch's, say -- dlopen hands back that one.)
namespace {
struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};
}  // namespace

static RcclApi *rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // RCCL must sit on the HIP runtime this library is bound to: a process may hold two ROCm stacks (PyTorch
        // wheels bundle their own libamdhip64 + librccl next to the system's, same sonames), and an RCCL of one on
        // the HIP runtime of the other fails in ncclCommInitRank.  So: the librccl that lives NEXT TO the loaded
        // libamdhip64 first, by full path (a bare soname would match whichever copy happens to be loaded already).
        std::vector<std::string> names;
        if (const char *env = getenv("EPGX_RCCL_LIBRARY")) names.emplace_back(env);
        Dl_info where;
        if (dladdr((const void *)&hipGetDeviceCount, &where) && where.dli_fname) {
            std::string dir(where.dli_fname);
            const size_t slash = dir.rfind('/');
            if (slash != std::string::npos) {
                dir.resize(slash + 1);
                names.push_back(dir + "librccl.so.1");
                names.push_back(dir + "librccl.so");
            }
        }
        names.emplace_back("librccl.so.1");
        for (const std::string &name : names) {
            if (name.empty()) continue;
            api.handle = dlopen(name.c_str(), RTLD_NOW | RTLD_LOCAL);
            if (api.handle) break;
            api.error = dlerror();
        }
        if (!api.handle) return;
        bool ok = true;
        auto sym = [&](const char *name) {
            void *p = dlsym(api.handle, name);
            if (!p) {
                ok = false;
                api.error = std::string("librccl lacks ") + name;
            }
            return p;
        };
        api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
        api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
        api.CommInitAll = (decltype(api.CommInitAll))sym("ncclCommInitAll");
        api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
        api.CommCount = (decltype(api.CommCount))sym("ncclCommCount");
        api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
        api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
        api.Send = (decltype(api.Send))sym("ncclSend");
        api.Recv = (decltype(api.Recv))sym("ncclRecv");
        api.Reduce = (decltype(api.Reduce))sym("ncclReduce");
        api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
        if (!ok) {
            dlclose(api.handle);
            api.handle = nullptr;
        }
    });
    return api.handle ? &api : nullptr;
}

#define RCCL_TRY(api, expr)                                                                                      \
    do {                                                                                                         \
        ncclResult_t r_ = (expr);                                                                                \
        if (r_ != ncclSuccess)                                                                                   \
            return fail(EPGX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, (api)->GetErrorString(r_), __FILE__, __LINE__); \
    } while (0)

struct epgx_comm {
    epgx_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    // every transfer runs on the communicator's own stream: behind `ev_in` (what the context's stream held when the
    // transfer was asked for), in front of `ev_out` (epgx_comm_join: the context's stream waits for it)
    hipStream_t stream = nullptr;
    hipEvent_t ev_in = nullptr, ev_out = nullptr;
};

// every peer -> root, point to point, in ONE group: the transfers run concurrently, each over the peer's own xGMI
// link to the root (a ring or tree collective would be bound by a single link).  Rank r's `nbytes` land at
// gathered + r * stride; the root's own block is a device-to-device copy unless the slab was produced in place.
static int gather_blocks(RcclApi *api, ncclComm_t comm, hipStream_t stream, int rank, int world, const void *send,
                         void *gathered, int64_t nbytes, int64_t stride, int root) {
    const size_t count = (size_t)(nbytes / 8);
    if (rank == root) {
        char *mine = (char *)gathered + (size_t)rank * (size_t)stride;
        if (send != (const void *)mine && nbytes)
            HIP_TRY(hipMemcpyAsync(mine, send, (size_t)nbytes, hipMemcpyDeviceToDevice, stream));
        if (world > 1 && count) {
            RCCL_TRY(api, api->GroupStart());
            for (int peer = 0; peer < world; ++peer)
                if (peer != root)
                    RCCL_TRY(api, api->Recv((char *)gathered + (size_t)peer * (size_t)stride, count, ncclDouble, peer, comm, stream));
            RCCL_TRY(api, api->GroupEnd());
        }
    } else if (count) {
        RCCL_TRY(api, api->Send(send, count, ncclDouble, root, comm, stream));
    }
    return EPGX_OK;
}

extern "C" int epgx_comm_unique_id(void *id_out) {
    if (!id_out) return fail(EPGX_ERR_INVALID, "epgx_comm_unique_id: NULL argument");
    RcclApi *api = rccl_api();
    if (!api) return fail(EPGX_ERR_UNSUPPORTED, "epgx_comm_unique_id: cannot load librccl.so.1 (set EPGX_RCCL_LIBRARY)");
    static_assert(sizeof(ncclUniqueId) == EPGX_COMM_ID_BYTES, "ncclUniqueId size");
    ncclUniqueId id;
    RCCL_TRY(api, api->GetUniqueId(&id));
    memcpy(id_out, &id, sizeof(id));
    return EPGX_OK;
}

extern "C" int epgx_comm_create(epgx_ctx *ctx, const void *id, int32_t rank, int32_t world_size, epgx_comm **out) {
    if (!ctx || !id || !out) return fail(EPGX_ERR_INVALID, "epgx_comm_create: NULL argument");
    *out = nullptr;
    if (world_size < 1 || rank < 0 || rank >= world_size)
        return fail(EPGX_ERR_INVALID, "epgx_comm_create: rank %d of %d", rank, world_size);
    RcclApi *api = rccl_api();
    if (!api) return fail(EPGX_ERR_UNSUPPORTED, "epgx_comm_create: cannot load librccl.so.1 (set EPGX_RCCL_LIBRARY)");
    if (int rc = set_device(ctx)) return rc;
    epgx_comm *cm = new (std::nothrow) epgx_comm();
    if (!cm) return fail(EPGX_ERR_NOMEM, "epgx_comm_create: host allocation failed");
    cm->ctx = ctx;
    cm->rank = rank;
    cm->world = world_size;
    hipError_t e = hipStreamCreateWithFlags(&cm->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&cm->ev_in, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&cm->ev_out, hipEventDisableTiming);
    if (e != hipSuccess) {
        epgx_comm_destroy(cm);
        return fail(EPGX_ERR_HIP, "epgx_comm_create: %s", hipGetErrorString(e));
    }
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    const ncclResult_t r = api->CommInitRank(&cm->comm, world_size, uid, rank);
    if (r != ncclSuccess) {
        epgx_comm_destroy(cm);
        return fail(EPGX_ERR_HIP, "epgx_comm_create: ncclCommInitRank: %s", api->GetErrorString(r));
    }
    *out = cm;
    return EPGX_OK;
}

extern "C" int epgx_comm_destroy(epgx_comm *cm) {
    if (!cm) return EPGX_OK;
    RcclApi *api = rccl_api();
    (void)hipSetDevice(cm->ctx->device);
    if (cm->stream) (void)hipStreamSynchronize(cm->stream);
    (void)hipStreamSynchronize(cm->ctx->stream);
    if (api && cm->comm) (void)api->CommDestroy(cm->comm);
    if (cm->ev_in) (void)hipEventDestroy(cm->ev_in);
    if (cm->ev_out) (void)hipEventDestroy(cm->ev_out);
    if (cm->stream) (void)hipStreamDestroy(cm->stream);
    delete cm;
    return EPGX_OK;
}

extern "C" int epgx_comm_count(const epgx_comm *cm, int32_t *n_ranks) {
    if (!cm || !n_ranks) return fail(EPGX_ERR_INVALID, "epgx_comm_count: NULL argument");
    RcclApi *api = rccl_api();
    if (!api) return fail(EPGX_ERR_UNSUPPORTED, "epgx_comm_count: librccl is not loaded");
    int n = 0;
    RCCL_TRY(api, api->CommCount(cm->comm, &n));
    *n_ranks = n;
    return EPGX_OK;
}

// the communicator's stream takes over from the context's stream: everything enqueued there so far comes first
static int comm_enter(epgx_comm *cm) {
    HIP_TRY(hipEventRecord(cm->ev_in, cm->ctx->stream));
    HIP_TRY(hipStreamWaitEvent(cm->stream, cm->ev_in, 0));
    return EPGX_OK;
}

extern "C" int epgx_comm_join(epgx_comm *cm) {
    if (!cm) return fail(EPGX_ERR_INVALID, "epgx_comm_join: comm is NULL");
    if (int rc = set_device(cm->ctx)) return rc;
    HIP_TRY(hipEventRecord(cm->ev_out, cm->stream));
    HIP_TRY(hipStreamWaitEvent(cm->ctx->stream, cm->ev_out, 0));
    return EPGX_OK;
}

extern "C" int epgx_comm_gather_part(epgx_comm *cm, const void *send, void *gathered, int64_t nbytes, int64_t block_stride,
                                     int32_t root) {
    if (!cm) return fail(EPGX_ERR_INVALID, "epgx_comm_gather_part: comm is NULL");
    if (nbytes < 0 || (nbytes & 7)) return fail(EPGX_ERR_INVALID, "epgx_comm_gather_part: nbytes=%lld must be a non-negative multiple of 8", (long long)nbytes);
    if (block_stride < nbytes) return fail(EPGX_ERR_INVALID, "epgx_comm_gather_part: block_stride=%lld < nbytes=%lld", (long long)block_stride, (long long)nbytes);
    if (root < 0 || root >= cm->world) return fail(EPGX_ERR_INVALID, "epgx_comm_gather_part: root %d of %d ranks", root, cm->world);
    if (nbytes && !send) return fail(EPGX_ERR_INVALID, "epgx_comm_gather_part: send is NULL");
    if (nbytes && cm->rank == root && !gathered) return fail(EPGX_ERR_INVALID, "epgx_comm_gather_part: the root needs a receive buffer");
    RcclApi *api = rccl_api();
    if (!api) return fail(EPGX_ERR_UNSUPPORTED, "epgx_comm_gather_part: librccl is not loaded");
    if (int rc = set_device(cm->ctx)) return rc;
    if (int rc = comm_enter(cm)) return rc;
    return gather_blocks(api, cm->comm, cm->stream, cm->rank, cm->world, send, gathered, nbytes, block_stride, root);
}

extern "C" int epgx_comm_gather(epgx_comm *cm, const void *send, void *gathered, int64_t nbytes, int32_t root) {
    if (int rc = epgx_comm_gather_part(cm, send, gathered, nbytes, nbytes, root)) return rc;
    return epgx_comm_join(cm);
}

extern "C" int epgx_comm_reduce(epgx_comm *cm, const void *send, void *recv, int64_t count, int32_t root) {
    if (!cm) return fail(EPGX_ERR_INVALID, "epgx_comm_reduce: comm is NULL");
    if (count < 0) return fail(EPGX_ERR_INVALID, "epgx_comm_reduce: count < 0");
    if (root < 0 || root >= cm->world) return fail(EPGX_ERR_INVALID, "epgx_comm_reduce: root %d of %d ranks", root, cm->world);
    if (count && !send) return fail(EPGX_ERR_INVALID, "epgx_comm_reduce: send is NULL");
    if (count && cm->rank == root && !recv) return fail(EPGX_ERR_INVALID, "epgx_comm_reduce: the root needs a receive buffer");
    RcclApi *api = rccl_api();
    if (!api) return fail(EPGX_ERR_UNSUPPORTED, "epgx_comm_reduce: librccl is not loaded");
    if (int rc = set_device(cm->ctx)) return rc;
    if (int rc = comm_enter(cm)) return rc;
    if (count)   // (non-root ranks may pass recv = NULL: ncclReduce only writes at the root)
        RCCL_TRY(api, api->Reduce(send, recv, (size_t)count, ncclDouble, ncclSum, root, cm->comm, cm->stream));
    return epgx_comm_join(cm);
}

extern "C" int epgx_memcpy_d2h_2d(epgx_ctx *ctx, void *host, int64_t host_pitch, const void *dptr, int64_t dev_pitch,
                                  int64_t width_bytes, int64_t rows) {
    if (!ctx || width_bytes < 0 || rows < 0 || host_pitch < width_bytes || dev_pitch < width_bytes ||
        (width_bytes && rows && (!host || !dptr)))
        return fail(EPGX_ERR_INVALID, "epgx_memcpy_d2h_2d: bad argument");
    if (int rc = set_device(ctx)) return rc;
    if (width_bytes && rows) {
        HIP_TRY(hipMemcpy2DAsync(host, (size_t)host_pitch, dptr, (size_t)dev_pitch, (size_t)width_bytes, (size_t)rows,
                                 hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    return EPGX_OK;
}

// ------------------------------------------------------------------------------ host copy threads
// A few persistent host threads move staged blocks into pageable destinations (first touch of the pages included): one
// thread copies ~10 GB/s into fresh memory, the PCIe link brings 54.  Process-wide (the pipelines of several contexts
// share them), created on first use, never more than the cores this process may use.
namespace {
int usable_cpus() {
    int n = 0;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof(set), &set) == 0) n = CPU_COUNT(&set);
    if (n <= 0) n = (int)std::thread::hardware_concurrency();
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {   // cgroup v2 CPU quota (a GPU box hands a job a share of its cores)
        char quota[32];
        long period = 0;
        if (fscanf(f, "%31s %ld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0)
            n = std::min<long>(n, std::max<long>(1, (atol(quota) + period - 1) / period));
        fclose(f);
    }
    return std::max(1, n);
}

class CopyPool {
    struct Batch {
        std::mutex m;
        std::condition_variable cv;
        int left = 0;
    };
    struct Task {
        const std::function<void(int)> *fn;
        int index;
        Batch *batch;
    };
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<Task> queue_;
    std::vector<std::thread> workers_;
    bool stop_ = false;

    static void finish(const Task &t) {
        (*t.fn)(t.index);
        std::lock_guard<std::mutex> g(t.batch->m);
        if (--t.batch->left == 0) t.batch->cv.notify_all();
    }
    void loop() {
        for (;;) {
            Task t;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return stop_ || !queue_.empty(); });
                if (queue_.empty()) return;
                t = queue_.front();
                queue_.pop_front();
            }
            finish(t);
        }
    }

    cpu_set_t preferred_;
    bool has_preferred_ = false;

public:
    explicit CopyPool(int n) {
        CPU_ZERO(&preferred_);
        for (int i = 0; i < n; ++i) workers_.emplace_back([this] { loop(); });
    }
    ~CopyPool() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto &w : workers_) w.join();
    }
    int lanes() const { return std::max<int>(1, (int)workers_.size()); }
    // The workers keep to the CPUs of the NUMA node(s) that hold the staging blocks -- the node next to the GPU: a copy
    // thread on the other socket reads every staged byte across the socket link (measured: the same download 9 ms or 16 - 20 ms
    // from one process to the next, depending on where the scheduler had put the threads).  Union over all contexts.
    void prefer(const cpu_set_t &cpus) {
        std::lock_guard<std::mutex> g(m_);
        CPU_OR(&preferred_, &preferred_, &cpus);
        has_preferred_ = true;
        for (auto &w : workers_) (void)pthread_setaffinity_np(w.native_handle(), sizeof(preferred_), &preferred_);
    }
    // fn(0) .. fn(n - 1) on the workers (on the caller when there are none); returns when all are done
    void parallel(int n, const std::function<void(int)> &fn) {
        if (n <= 0) return;
        if (workers_.empty()) {
            for (int i = 0; i < n; ++i) fn(i);
            return;
        }
        Batch batch;
        batch.left = n;
        {
            std::lock_guard<std::mutex> g(m_);
            for (int i = 0; i < n; ++i) queue_.push_back({&fn, i, &batch});
        }
        cv_.notify_all();
        std::unique_lock<std::mutex> g(batch.m);
        batch.cv.wait(g, [&] { return batch.left == 0; });
    }
    static CopyPool &get() {
        static CopyPool *pool = [] {
            const char *env = getenv("EPGX_COPY_THREADS");
            int n = env ? atoi(env) : std::min(12, usable_cpus() - 2);
            return new CopyPool(std::max(0, n));   // (leaked on purpose: no static-destruction order games at exit)
        }();
        return *pool;
    }
};

// NUMA node that holds the (resident) page at `p`, or -1
int node_of_page(void *p) {
    int status = -1;
    void *pages[1] = {p};
    const long rc = syscall(SYS_move_pages, 0, 1UL, pages, nullptr, &status, 0);
    return rc == 0 ? status : -1;
}

// CPUs of a NUMA node (/sys/devices/system/node/nodeN/cpulist: "0-63,128-191"), restricted to those this process may use
bool node_cpus(int node, cpu_set_t *out) {
    char path[96];
    snprintf(path, sizeof(path), "/sys/devices/system/node/node%d/cpulist", node);
    FILE *f = fopen(path, "r");
    if (!f) return false;
    char text[4096];
    const bool got = fgets(text, sizeof(text), f) != nullptr;
    fclose(f);
    if (!got) return false;
    cpu_set_t allowed;
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return false;
    CPU_ZERO(out);
    int n = 0;
    for (char *tok = strtok(text, ",\n"); tok; tok = strtok(nullptr, ",\n")) {
        int a = 0, b = 0;
        const int k = sscanf(tok, "%d-%d", &a, &b);
        if (k < 1) continue;
        if (k == 1) b = a;
        for (int c = a; c <= b && c < CPU_SETSIZE; ++c)
            if (CPU_ISSET(c, &allowed)) {
                CPU_SET(c, out);
                ++n;
            }
    }
    return n > 0;
}

// rows x width bytes, staged contiguously at `src` (pitch = width), into dst (pitch dst_pitch): split by BYTES over the lanes
void scatter_rows(const char *src, char *dst, size_t dst_pitch, size_t width, int64_t rows) {
    CopyPool &pool = CopyPool::get();
    const size_t total = width * (size_t)rows;
    const int lanes = (int)std::min<size_t>((size_t)pool.lanes(), std::max<size_t>(1, total >> 20));   // at least 1 MiB per lane
    pool.parallel(lanes, [&](int t) {
        size_t at = total * (size_t)t / (size_t)lanes, end = total * (size_t)(t + 1) / (size_t)lanes;
        at &= ~(size_t)63;
        if (t + 1 < lanes) end &= ~(size_t)63;
        while (at < end) {
            const size_t r = at / width, c = at - r * width;
            const size_t n = std::min(width - c, end - at);
            memcpy(dst + r * dst_pitch + c, src + at, n);
            at += n;
        }
    });
}

bool host_is_pinned(const void *p, size_t span) {
    auto one = [](const void *q) {
        hipPointerAttribute_t attr;
        memset(&attr, 0, sizeof(attr));
        if (hipPointerGetAttributes(&attr, q) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        return attr.type == hipMemoryTypeHost;
    };
    return one(p) && (span <= 1 || one((const char *)p + span - 1));
}

// a [rows][width] byte region of device memory and where it goes on the host; `after`: event (on the context's
// stream) that marks the kernel whose output this is, or null
struct Region {
    const char *src;
    size_t src_pitch;
    char *dst;
    size_t dst_pitch, width;
    int64_t rows;
    hipEvent_t after;
};
}  // namespace

static int ensure_copy_stream(epgx_ctx *ctx, int n_events) {
    if (!ctx->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    while ((int)ctx->slab_events.size() < n_events) {
        hipEvent_t ev;
        HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        ctx->slab_events.push_back(ev);
    }
    return EPGX_OK;
}

static int ensure_stages(epgx_ctx *ctx) {
    if (!ctx->stages.empty()) return EPGX_OK;
    // 4 x 64 MiB: a block crosses PCIe in 1.2 ms; pinned once per context (~50 ms), EPGX_STAGE_MB / EPGX_STAGES to change
    const char *mb = getenv("EPGX_STAGE_MB"), *nst = getenv("EPGX_STAGES");
    const size_t bytes = (size_t)std::max(1, mb ? atoi(mb) : 64) << 20;
    const int n = std::max(2, nst ? atoi(nst) : 4);
    hipError_t e = hipSuccess;
    for (int i = 0; i < n && e == hipSuccess; ++i) {
        epgx_ctx::Stage st;
        e = hipHostMalloc(&st.host, bytes, hipHostMallocDefault);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&st.done, hipEventDisableTiming);
        if (e == hipSuccess) ctx->stages.push_back(st);
        else if (st.host) (void)hipHostFree(st.host);
    }
    if (e != hipSuccess) {   // all or nothing: a later call tries again
        for (auto &st : ctx->stages) {
            (void)hipEventDestroy(st.done);
            (void)hipHostFree(st.host);
        }
        ctx->stages.clear();
        return fail(e == hipErrorOutOfMemory ? EPGX_ERR_NOMEM : EPGX_ERR_HIP, "staging blocks: %s", hipGetErrorString(e));
    }
    ctx->stage_bytes = bytes;
    {   // the copy threads move next to this memory (EPGX_COPY_AFFINITY=off: wherever the scheduler puts them)
        const char *aff = getenv("EPGX_COPY_AFFINITY");
        cpu_set_t cpus;
        const int node = (aff && strcmp(aff, "off") == 0) ? -1 : node_of_page(ctx->stages[0].host);
        if (node >= 0 && node_cpus(node, &cpus)) CopyPool::get().prefer(cpus);
        if (getenv("EPGX_TRACE")) fprintf(stderr, "[epgx] staging ring: %d x %zu MiB on NUMA node %d\n", n, bytes >> 20, node);
    }
    return EPGX_OK;
}

// The download engine (caller holds ctx->pipeline).  Page-locked destinations receive strided 2-D copies directly;
// pageable ones are filled through the staging ring: tile i + 1 .. i + ring - 1 cross PCIe while the host threads
// scatter tile i into place.  Enqueues on the copy stream; with `pinned` nothing has completed when this returns
// (the caller drains), otherwise everything has.
static int copy_regions(epgx_ctx *ctx, const std::vector<Region> &regions, bool pinned) {
    if (pinned) {
        for (const Region &r : regions) {
            if (!r.rows || !r.width) continue;
            if (r.after) HIP_TRY(hipStreamWaitEvent(ctx->copy_stream, r.after, 0));
            HIP_TRY(hipMemcpy2DAsync(r.dst, r.dst_pitch, r.src, r.src_pitch, r.width, (size_t)r.rows, hipMemcpyDeviceToHost, ctx->copy_stream));
        }
        return EPGX_OK;
    }
    if (int rc = ensure_stages(ctx)) return rc;
    struct Tile { int region; int64_t row0, rows; size_t col0, width; };
    std::vector<Tile> tiles;
    const size_t cap = ctx->stage_bytes;
    for (int ri = 0; ri < (int)regions.size(); ++ri) {
        const Region &r = regions[(size_t)ri];
        if (!r.rows || !r.width) continue;
        if (r.width <= cap) {
            const int64_t per = std::max<int64_t>(1, (int64_t)(cap / r.width));
            for (int64_t r0 = 0; r0 < r.rows; r0 += per) tiles.push_back({ri, r0, std::min(per, r.rows - r0), 0, r.width});
        } else {
            for (int64_t r0 = 0; r0 < r.rows; ++r0)
                for (size_t c0 = 0; c0 < r.width; c0 += cap) tiles.push_back({ri, r0, 1, c0, std::min(cap, r.width - c0)});
        }
    }
    const int ring = (int)ctx->stages.size(), n = (int)tiles.size();
    int waited_region = -1;
    auto issue = [&](int i) -> hipError_t {
        const Tile &t = tiles[(size_t)i];
        const Region &r = regions[(size_t)t.region];
        hipError_t e = hipSuccess;
        if (r.after && t.region != waited_region) e = hipStreamWaitEvent(ctx->copy_stream, r.after, 0);
        waited_region = t.region;
        epgx_ctx::Stage &st = ctx->stages[(size_t)(i % ring)];
        if (e == hipSuccess)
            e = hipMemcpy2DAsync(st.host, t.width, r.src + (size_t)t.row0 * r.src_pitch + t.col0, r.src_pitch, t.width, (size_t)t.rows,
                                 hipMemcpyDeviceToHost, ctx->copy_stream);
        if (e == hipSuccess) e = hipEventRecord(st.done, ctx->copy_stream);
        return e;
    };
    hipError_t e = hipSuccess;
    int issued = 0;
    for (; issued < std::min(ring, n) && e == hipSuccess; ++issued) e = issue(issued);
    for (int i = 0; i < n && e == hipSuccess; ++i) {
        epgx_ctx::Stage &st = ctx->stages[(size_t)(i % ring)];
        e = hipEventSynchronize(st.done);
        if (e != hipSuccess) break;
        const Tile &t = tiles[(size_t)i];
        const Region &r = regions[(size_t)t.region];
        scatter_rows((const char *)st.host, r.dst + (size_t)t.row0 * r.dst_pitch + t.col0, r.dst_pitch, t.width, t.rows);
        if (issued < n) e = issue(issued++);
    }
    if (e != hipSuccess) return fail(EPGX_ERR_HIP, "staged download: %s", hipGetErrorString(e));
    return EPGX_OK;
}

// both streams drain here whatever happened before: the caller may recycle the device scratch and the host block as
// soon as the entry point returns
static int drain_pipeline(epgx_ctx *ctx, int rc, const char *who) {
    const hipError_t e1 = ctx->copy_stream ? hipStreamSynchronize(ctx->copy_stream) : hipSuccess;
    const hipError_t e2 = hipStreamSynchronize(ctx->stream);
    if (!rc && (e1 != hipSuccess || e2 != hipSuccess))
        rc = fail(EPGX_ERR_HIP, "%s: %s", who, hipGetErrorString(e1 != hipSuccess ? e1 : e2));
    return rc;
}

// ------------------------------------------------------------------------------ pipelined run to host memory
extern "C" int epgx_run_to_host(epgx_ctx *ctx, const epgx_plan *plan, int32_t K, int64_t vox0, int64_t nvox, void *signal_dev,
                                int64_t dev_ld, void *signal_host, int64_t host_ld, int64_t host_col0, int64_t slab, int32_t host_dtype) {
    if (!ctx || !plan) return fail(EPGX_ERR_INVALID, "epgx_run_to_host: NULL argument");
    if (host_dtype != EPGX_SIGNAL_C128 && host_dtype != EPGX_SIGNAL_C64)
        return fail(EPGX_ERR_INVALID, "epgx_run_to_host: host_dtype %d is no epgx_signal_dtype", host_dtype);
    const size_t rec = host_dtype == EPGX_SIGNAL_C64 ? sizeof(float2) : sizeof(d2);   // bytes per record on its way to the host
    if (plan->ctx != ctx) return fail(EPGX_ERR_INVALID, "epgx_run_to_host: plan belongs to another context");
    const int n_adc = plan->n_adc;
    if (vox0 < 0 || nvox < 0 || vox0 + nvox > plan->nvox_total)
        return fail(EPGX_ERR_INVALID, "epgx_run_to_host: voxel range [%lld,%lld) outside the grid (%lld voxels)", (long long)vox0,
                    (long long)(vox0 + nvox), (long long)plan->nvox_total);
    if (n_adc > 0 && nvox > 0 && (!signal_dev || !signal_host)) return fail(EPGX_ERR_INVALID, "epgx_run_to_host: signal buffers are NULL");
    if (dev_ld < nvox || host_col0 < 0 || host_ld < host_col0 + nvox)
        return fail(EPGX_ERR_INVALID, "epgx_run_to_host: %lld voxels do not fit rows of %lld (device) / columns [%lld, ..) of %lld (host)",
                    (long long)nvox, (long long)dev_ld, (long long)host_col0, (long long)host_ld);
    if (slab < 0) return fail(EPGX_ERR_INVALID, "epgx_run_to_host: slab < 0");
    if (nvox == 0) return EPGX_OK;
    if (int rc = set_device(ctx)) return rc;
    const auto tic = std::chrono::steady_clock::now();
    if (slab == 0) {   // ~8 slabs, at least 64 Ki voxels each (a launch should fill the chip), whole wavefront groups
        slab = std::max<int64_t>((nvox + 7) / 8, 65536);
        slab = (slab + 63) & ~(int64_t)63;
    }
    slab = std::max(slab, (nvox + 63) / 64);   // (one event per slab, 64 of them)
    const int n_slabs = (int)((nvox + slab - 1) / slab);
    std::lock_guard<std::mutex> one_at_a_time(ctx->pipeline);
    if (int rc = ensure_copy_stream(ctx, n_slabs)) return rc;
    const bool pinned = n_adc > 0 && host_is_pinned((const char *)signal_host + rec * (size_t)host_col0,
                                                    rec * ((size_t)(n_adc - 1) * (size_t)host_ld + (size_t)nvox));
    const int n_ops = (int)plan->ops.size();
    int rc = EPGX_OK;
    // complex64 destination: every slab is narrowed (behind its kernel, on the same stream) into scratch [n_adc][dev_ld] float2
    // of the context's block cache; the copies then move 8 bytes per record
    void *narrow = nullptr;
    if (host_dtype == EPGX_SIGNAL_C64 && n_adc > 0) {
        const hipError_t e = dev_alloc(ctx, &narrow, sizeof(float2) * (size_t)n_adc * (size_t)dev_ld);
        if (e != hipSuccess) return fail(EPGX_ERR_NOMEM, "epgx_run_to_host: scratch for complex64 records: %s", hipGetErrorString(e));
    }
    const char *src_base = narrow ? (const char *)narrow : (const char *)signal_dev;
    std::vector<Region> regions;
    for (int k = 0; k < n_slabs && !rc; ++k) {
        const int64_t j0 = (int64_t)k * slab, nv = std::min(slab, nvox - j0);
        rc = epgx_run(ctx, plan, 0, n_ops, vox0 + j0, nv, nullptr, nullptr, K, signal_dev, dev_ld, j0);
        if (rc || n_adc <= 0) continue;
        if (narrow) rc = launch_narrow(ctx, (const d2 *)signal_dev + j0, dev_ld, (float2 *)narrow + j0, dev_ld, n_adc, nv);
        if (rc) continue;
        hipEvent_t ev = ctx->slab_events[(size_t)k];
        const hipError_t e = hipEventRecord(ev, ctx->stream);
        if (e != hipSuccess) {
            rc = fail(EPGX_ERR_HIP, "epgx_run_to_host: %s", hipGetErrorString(e));
            break;
        }
        Region r = {src_base + rec * (size_t)j0, rec * (size_t)dev_ld,
                    (char *)signal_host + rec * (size_t)(host_col0 + j0), rec * (size_t)host_ld, rec * (size_t)nv,
                    n_adc, ev};
        if (pinned) {   // (straight away: the copy of slab k runs under the kernel of slab k + 1)
            std::vector<Region> single(1, r);
            rc = copy_regions(ctx, single, true);
        } else {
            regions.push_back(r);
        }
    }
    if (!rc && !regions.empty()) rc = copy_regions(ctx, regions, false);
    rc = drain_pipeline(ctx, rc, "epgx_run_to_host");
    if (narrow) dev_free(ctx, narrow);
    if (getenv("EPGX_TRACE"))
        fprintf(stderr, "[epgx] run_to_host %lld voxels x %d rows (%s), %d slabs, %s: %.3f ms\n", (long long)nvox, n_adc,
                narrow ? "complex64" : "complex128", n_slabs, pinned ? "page-locked destination" : "staged into pageable memory",
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tic).count());
    return rc;
}

extern "C" int epgx_download_2d(epgx_ctx *ctx, void *host, int64_t host_pitch, const void *dptr, int64_t dev_pitch,
                                int64_t width_bytes, int64_t rows) {
    if (!ctx || width_bytes < 0 || rows < 0 || host_pitch < width_bytes || dev_pitch < width_bytes ||
        (width_bytes && rows && (!host || !dptr)))
        return fail(EPGX_ERR_INVALID, "epgx_download_2d: bad argument");
    if (!width_bytes || !rows) return EPGX_OK;
    if (int rc = set_device(ctx)) return rc;
    std::lock_guard<std::mutex> one_at_a_time(ctx->pipeline);
    if (int rc = ensure_copy_stream(ctx, 1)) return rc;
    hipEvent_t ev = ctx->slab_events[0];
    HIP_TRY(hipEventRecord(ev, ctx->stream));
    const bool pinned = host_is_pinned(host, (size_t)(rows - 1) * (size_t)host_pitch + (size_t)width_bytes);
    std::vector<Region> regions(1, Region{(const char *)dptr, (size_t)dev_pitch, (char *)host, (size_t)host_pitch, (size_t)width_bytes, rows, ev});
    const int rc = copy_regions(ctx, regions, pinned);
    return drain_pipeline(ctx, rc, "epgx_download_2d");
}

// ------------------------------------------------------------------------------ host-buffer convenience
extern "C" int epgx_simulate_f64(epgx_ctx *ctx, const epgx_plan_desc *desc, int32_t K, const double *init_half,
                                 const double *density, void *signal_out, double *state_out, int32_t signal_dtype) {
    if (!ctx || !desc) return fail(EPGX_ERR_INVALID, "epgx_simulate_f64: NULL argument");
    if (signal_dtype != EPGX_SIGNAL_C128 && signal_dtype != EPGX_SIGNAL_C64)
        return fail(EPGX_ERR_INVALID, "epgx_simulate_f64: signal_dtype %d is no epgx_signal_dtype", signal_dtype);
    if (desc->n_adc > 0 && !signal_out) return fail(EPGX_ERR_INVALID, "epgx_simulate_f64: signal_out is NULL");
    epgx_plan *pl = nullptr;
    int rc = epgx_plan_create(ctx, desc, &pl);
    if (rc) return rc;
    const int64_t nvox = pl->nvox_total;
    epgx_state *st = nullptr;
    void *d_sig = nullptr;
    const bool need_state = init_half || density || state_out;
    if (need_state) {
        rc = epgx_state_create(ctx, nvox, K, &st);
        if (!rc && init_half) rc = epgx_state_upload(st, init_half, density);
        else if (!rc && density) rc = epgx_memcpy_h2d(ctx, st->dens, density, (int64_t)sizeof(double) * nvox);
    }
    const int64_t sig_bytes = (int64_t)sizeof(d2) * desc->n_adc * nvox;
    if (!rc && desc->n_adc) rc = epgx_malloc(ctx, sig_bytes, &d_sig);
    if (!rc) {
        const bool from_eq = !init_half;
        // equilibrium start: `in` = NULL unless a custom density must be honoured
        const epgx_state *in = (from_eq && !density) ? nullptr : st;
        if (from_eq && density) {
            // rebuild the equilibrium state from the uploaded density with a RESET-only plan
            epgx_op r;
            memset(&r, 0, sizeof(r));
            r.opcode = EPGX_OP_RESET;
            r.space = -1;
            epgx_plan_desc d1 = *desc;
            d1.n_ops = 1;
            d1.ops = &r;
            d1.n_adc = 0;
            epgx_plan *p1 = nullptr;
            rc = epgx_plan_create(ctx, &d1, &p1);
            if (!rc) rc = epgx_run(ctx, p1, 0, 1, 0, nvox, st, st, K, nullptr, 0, 0);
            epgx_plan_destroy(p1);
        }
        // nothing but the signal to bring back: voxel slabs, each one's columns on their way to the host while the next computes
        const bool piped = !in && !state_out && desc->n_adc > 0 && sig_bytes >= ((int64_t)32 << 20);
        if (!rc && piped) {
            rc = epgx_run_to_host(ctx, pl, K, 0, nvox, d_sig, nvox, signal_out, nvox, 0, 0, signal_dtype);
            if (!rc) {   // done, signal included
                epgx_free(ctx, d_sig);
                epgx_state_destroy(st);
                epgx_plan_destroy(pl);
                return EPGX_OK;
            }
        }
        if (!rc && !piped) rc = epgx_run(ctx, pl, 0, desc->n_ops, 0, nvox, in, state_out ? st : nullptr, K, d_sig, nvox, 0);
    }
    if (!rc && desc->n_adc && signal_dtype == EPGX_SIGNAL_C64) {   // narrowed on the device, then half the bytes over PCIe
        void *small = nullptr;
        rc = epgx_malloc(ctx, sig_bytes / 2, &small);
        if (!rc) rc = launch_narrow(ctx, d_sig, nvox, small, nvox, desc->n_adc, nvox);
        if (!rc) rc = epgx_download_2d(ctx, signal_out, sig_bytes / 2, small, sig_bytes / 2, sig_bytes / 2, 1);
        if (small) epgx_free(ctx, small);
    } else if (!rc && desc->n_adc) {
        rc = epgx_download_2d(ctx, signal_out, sig_bytes, d_sig, sig_bytes, sig_bytes, 1);
    }
    if (!rc && state_out) rc = epgx_state_download(st, state_out, nullptr);
    if (d_sig) epgx_free(ctx, d_sig);
    epgx_state_destroy(st);
    epgx_plan_destroy(pl);
    return rc;
}

// the contexts of epgx_simulate_sharded_f64: one per device, created on first use and kept for the process (a context carries
// its block cache, its copy stream and -- for results in pageable memory -- a ring of page-locked staging blocks that costs
// ~50 ms to pin: per call that was most of a small simulate)
static int sharded_context(int device, epgx_ctx **out) {
    static std::mutex lock;
    static std::map<int, epgx_ctx *> kept;
    std::lock_guard<std::mutex> guard(lock);
    auto hit = kept.find(device);
    if (hit == kept.end()) {
        epgx_ctx *ctx = nullptr;
        if (int rc = epgx_ctx_create(device, &ctx)) return rc;
        hit = kept.emplace(device, ctx).first;
    }
    *out = hit->second;
    return EPGX_OK;
}

// communicator sets of the single-process gather (EPGX_SHARDED_GATHER=rccl): created once per `ngpu`, kept for the process
static int sharded_comms(RcclApi *api, int ngpu, ncclComm_t **out) {
    static std::mutex lock;
    static std::map<int, std::vector<ncclComm_t>> sets;
    std::lock_guard<std::mutex> guard(lock);
    auto hit = sets.find(ngpu);
    if (hit == sets.end()) {
        std::vector<ncclComm_t> comms((size_t)ngpu, nullptr);
        const ncclResult_t r = api->CommInitAll(comms.data(), ngpu, nullptr);
        if (r != ncclSuccess) return fail(EPGX_ERR_HIP, "epgx_simulate_sharded_f64: ncclCommInitAll: %s", api->GetErrorString(r));
        hit = sets.emplace(ngpu, std::move(comms)).first;
    }
    *out = hit->second.data();
    return EPGX_OK;
}

extern "C" int epgx_simulate_sharded_f64(const epgx_plan_desc *desc, int32_t K, int32_t ngpu,
                                         const double *density, void *signal_out, int32_t signal_dtype) {
    if (!desc || !signal_out) return fail(EPGX_ERR_INVALID, "epgx_simulate_sharded_f64: NULL argument");
    if (signal_dtype != EPGX_SIGNAL_C128 && signal_dtype != EPGX_SIGNAL_C64)
        return fail(EPGX_ERR_INVALID, "epgx_simulate_sharded_f64: signal_dtype %d is no epgx_signal_dtype", signal_dtype);
    const size_t rec = signal_dtype == EPGX_SIGNAL_C64 ? sizeof(float2) : sizeof(d2);
    if (desc->struct_size != sizeof(epgx_plan_desc))
        return fail(EPGX_ERR_INVALID, "epgx_simulate_sharded_f64: struct_size = %u, epgx_plan_desc has %zu bytes (ABI %d)", desc->struct_size,
                    sizeof(epgx_plan_desc), EPGX_ABI_VERSION);
    if (desc->ndim < 1 || desc->ndim > EPGX_MAX_DIMS || !desc->grid_shape)
        return fail(EPGX_ERR_INVALID, "epgx_simulate_sharded_f64: ndim %d not in [1,%d]", desc->ndim, EPGX_MAX_DIMS);
    if (desc->n_adc < 0) return fail(EPGX_ERR_INVALID, "epgx_simulate_sharded_f64: n_adc < 0");
    const int ndev = epgx_device_count();
    if (ngpu < 1 || ngpu > ndev)
        return fail(EPGX_ERR_INVALID, "epgx_simulate_sharded_f64: ngpu=%d, %d device(s) visible", ngpu, ndev);
    if (density) return fail(EPGX_ERR_UNSUPPORTED, "epgx_simulate_sharded_f64: custom density not supported");
    int64_t nvox = 1;
    for (int i = 0; i < desc->ndim; ++i) {
        if (desc->grid_shape[i] < 1) return fail(EPGX_ERR_INVALID, "epgx_simulate_sharded_f64: grid_shape[%d] < 1", i);
        nvox *= desc->grid_shape[i];
    }
    const int64_t slab = (nvox + ngpu - 1) / ngpu;
    const int64_t block = (int64_t)sizeof(d2) * desc->n_adc * slab;   // one rank's padded slab [n_adc][slab]
    // The result is a host array: by default every GPU downloads its own slab (ngpu PCIe links, no collective).
    // EPGX_SHARDED_GATHER=rccl gathers the slabs on GPU 0 first (the device-side route of epgx_comm_*, single-process form).
    const char *how = getenv("EPGX_SHARDED_GATHER");
    const bool use_rccl = how && strcmp(how, "rccl") == 0;
    RcclApi *api = use_rccl ? rccl_api() : nullptr;
    if (use_rccl && !api)
        return fail(EPGX_ERR_UNSUPPORTED, "epgx_simulate_sharded_f64: cannot load librccl.so.1 (set EPGX_RCCL_LIBRARY)");
    std::vector<epgx_ctx *> ctxs(ngpu, nullptr);
    std::vector<epgx_plan *> plans(ngpu, nullptr);
    std::vector<void *> sig(ngpu, nullptr);
    std::vector<int64_t> v0(ngpu), nv(ngpu);
    void *gathered = nullptr;   // on device 0: [ngpu][n_adc][slab]
    int rc = EPGX_OK;
    for (int g = 0; g < ngpu && !rc; ++g) {
        v0[g] = std::min<int64_t>(nvox, g * slab);
        nv[g] = std::min<int64_t>(nvox, (g + 1) * slab) - v0[g];
        rc = sharded_context(g, &ctxs[g]);
        if (!rc) rc = epgx_plan_create(ctxs[g], desc, &plans[g]);
        if (!rc && g == 0 && use_rccl) rc = epgx_malloc(ctxs[0], std::max<int64_t>(block * ngpu, 16), &gathered);
        // (gather route: device 0 writes its slab straight into its block of the gathered buffer)
        if (!rc && !(g == 0 && use_rccl)) rc = epgx_malloc(ctxs[g], std::max<int64_t>(block, 16), &sig[g]);
    }
    if (!rc && !use_rccl) {
        // one host thread per device drives that device's slab pipeline (kernel slabs + their copies into the caller's array)
        std::vector<int> rcs(ngpu, EPGX_OK);
        std::vector<std::string> errs(ngpu);
        auto drive = [&](int g) {
            if (nv[g] <= 0) return;
            rcs[g] = epgx_run_to_host(ctxs[g], plans[g], K, v0[g], nv[g], sig[g], slab, signal_out, nvox, v0[g], 0, signal_dtype);
            if (rcs[g]) errs[g] = g_err;   // (thread-local message: carried back to the caller's thread below)
        };
        std::vector<std::thread> pool;
        for (int g = 1; g < ngpu; ++g) pool.emplace_back(drive, g);
        drive(0);
        for (auto &th : pool) th.join();
        for (int g = 0; g < ngpu && !rc; ++g)
            if (rcs[g]) rc = fail(rcs[g], "epgx_simulate_sharded_f64: device %d: %s", g, errs[g].c_str());
    }
    std::vector<void *> nar(ngpu, nullptr);   // complex64 route: the narrowed slabs (device g; on device 0 the gathered narrowed blocks)
    if (!rc && use_rccl) {
        // enqueue every slab first (async: the devices run concurrently), then ONE gather on the device side -- every GPU
        // sends its slab to GPU 0 over its own xGMI link -- and the download through GPU 0's PCIe link.  complex64 records
        // are narrowed where they were computed: half the bytes cross xGMI as well
        const bool c64 = signal_dtype == EPGX_SIGNAL_C64 && desc->n_adc > 0;
        const int64_t wire = c64 ? block / 2 : block;   // bytes of one rank's block on the wire
        for (int g = 0; g < ngpu && !rc; ++g) {
            void *dst = g == 0 ? gathered : sig[g];
            if (nv[g] < slab && desc->n_adc > 0) rc = epgx_memset(ctxs[g], dst, 0, block);   // (ragged / empty slab: defined padding)
            if (!rc && nv[g] > 0) rc = epgx_run(ctxs[g], plans[g], 0, desc->n_ops, v0[g], nv[g], nullptr, nullptr, K, dst, slab, 0);
            if (!rc && c64) rc = epgx_malloc(ctxs[g], std::max<int64_t>(g == 0 ? wire * ngpu : wire, 16), &nar[g]);
            if (!rc && c64) rc = launch_narrow(ctxs[g], dst, slab, nar[g], slab, desc->n_adc, slab);
        }
        void *root = c64 ? nar[0] : gathered;
        ncclComm_t *comms = nullptr;
        if (!rc && desc->n_adc > 0) rc = sharded_comms(api, ngpu, &comms);
        if (!rc && desc->n_adc > 0) {
            ncclResult_t r = api->GroupStart();
            for (int g = 0; g < ngpu && r == ncclSuccess; ++g) {
                if (hipSetDevice(g) != hipSuccess) { r = ncclUnhandledCudaError; break; }
                if (g == 0) {
                    for (int peer = 1; peer < ngpu && r == ncclSuccess; ++peer)
                        r = api->Recv((char *)root + (size_t)peer * (size_t)wire, (size_t)(wire / 8), ncclDouble, peer,
                                      comms[0], ctxs[0]->stream);
                } else {
                    r = api->Send(c64 ? nar[g] : sig[g], (size_t)(wire / 8), ncclDouble, 0, comms[g], ctxs[g]->stream);
                }
            }
            const ncclResult_t r2 = api->GroupEnd();
            if (r == ncclSuccess) r = r2;
            if (r != ncclSuccess) rc = fail(EPGX_ERR_HIP, "epgx_simulate_sharded_f64: RCCL gather: %s", api->GetErrorString(r));
        }
        // block g = [n_adc][slab] -> columns [v0, v0 + nv) of the caller's [n_adc][nvox] array
        for (int g = 0; g < ngpu && !rc; ++g) {
            if (nv[g] <= 0 || desc->n_adc <= 0) continue;
            if (!rc)   // (ordered behind the receives on GPU 0's stream)
                rc = epgx_download_2d(ctxs[0], (char *)signal_out + rec * v0[g], rec * nvox,
                                      (const char *)root + (size_t)g * (size_t)wire, rec * slab, rec * nv[g], desc->n_adc);
        }
    }
    for (int g = 0; g < ngpu; ++g) {
        if (!ctxs[g]) continue;
        int r2 = epgx_ctx_synchronize(ctxs[g]);
        if (!rc) rc = r2;
    }
    for (int g = 0; g < ngpu; ++g) {
        if (!ctxs[g]) continue;
        if (sig[g]) epgx_free(ctxs[g], sig[g]);
        if (nar[g]) epgx_free(ctxs[g], nar[g]);
        if (g == 0 && gathered) epgx_free(ctxs[0], gathered);
        epgx_plan_destroy(plans[g]);     // (the context stays: sharded_context)
    }
    return rc;
}
