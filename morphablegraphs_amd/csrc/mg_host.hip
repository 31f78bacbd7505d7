// libmg_hip host side: context, primitive constants, time grids, C-ABI entry points.
// gfx950 (MI355X) only.  All model preparation is float64 on the host; kernels live in
// mg_frames*.hip / mg_gmm.hip / mg_score.hip / mg_options.hip.
#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <time.h>

#include "mg_internal.h"
#include <dlfcn.h>

// ---------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";

void mg_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int mg_hip_fail(hipError_t e, const char *what) {
    mg_set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    if (e == hipErrorOutOfMemory) return MG_ERR_OUT_OF_MEMORY;
    if (e == hipErrorNoDevice || e == hipErrorInvalidDevice) return MG_ERR_NO_DEVICE;
    return MG_ERR_HIP;
}

extern "C" const char *mg_version(void) { return "mg_hip 0.1 (gfx950)"; }
extern "C" const char *mg_last_error(void) { return g_err; }
extern "C" const char *mg_status_string(int s) {
    switch (s) {
        case MG_OK: return "ok";
        case MG_ERR_INVALID_ARGUMENT: return "invalid argument";
        case MG_ERR_NO_DEVICE: return "no HIP device";
        case MG_ERR_HIP: return "HIP runtime error";
        case MG_ERR_UNSUPPORTED: return "unsupported shape";
        case MG_ERR_NOT_POSITIVE_DEFINITE: return "covariance not positive definite";
        case MG_ERR_OUT_OF_MEMORY: return "out of device memory";
        default: return "unknown status";
    }
}

#define MG_REQUIRE(cond, ...)            \
    do {                                 \
        if (!(cond)) {                   \
            mg_set_error(__VA_ARGS__);   \
            return MG_ERR_INVALID_ARGUMENT; \
        }                                \
    } while (0)

// ---------------------------------------------------------------------------------------
// RCCL, loaded on first use
// ---------------------------------------------------------------------------------------
struct mg_nccl_id { char internal[MG_DIST_ID_BYTES]; };
struct mg_rccl_api {
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, /* ncclUniqueId by value */ mg_nccl_id, int) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*Broadcast)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
static mg_rccl_api g_rccl;
static int mg_rccl_load() {
    if (g_rccl.lib) return MG_OK;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) { mg_set_error("mg_dist: librccl.so.1 cannot be loaded (%s)", dlerror()); return MG_ERR_UNSUPPORTED; }
    g_rccl.GetUniqueId = (int (*)(void *))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(void **, int, mg_nccl_id, int))dlsym(h, "ncclCommInitRank");
    g_rccl.AllGather = (int (*)(const void *, void *, size_t, int, void *, hipStream_t))dlsym(h, "ncclAllGather");
    g_rccl.Broadcast = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(h, "ncclBroadcast");
    g_rccl.CommDestroy = (int (*)(void *))dlsym(h, "ncclCommDestroy");
    g_rccl.GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.AllGather || !g_rccl.Broadcast || !g_rccl.CommDestroy) {
        mg_set_error("mg_dist: librccl lacks an expected symbol");
        dlclose(h);
        return MG_ERR_UNSUPPORTED;
    }
    g_rccl.lib = h;
    return MG_OK;
}
static int mg_rccl_fail(int rc, const char *what) {
    mg_set_error("RCCL error %d (%s) in %s", rc, g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "?", what);
    return MG_ERR_HIP;
}
// ---------------------------------------------------------------------------------------
// One setting of the HIP runtime, made when the library is loaded (before the runtime initialises, which it does at a process's first HIP call):
// kernel arguments in DEVICE memory.  By default the runtime keeps a dispatch's argument block in host memory, and the first thing every wave of
// the persistent frames kernel does is read it -- across the host link, ~1.5 us before anything else can start (measured on the bench's step:
// 77.9-78.1 -> 76.4 us per step in back-to-back processes on one box, kernel time by its own events likewise).  A value the caller has set stays.
// The library still READS no environment variable.
// ---------------------------------------------------------------------------------------
__attribute__((constructor)) static void mg_runtime_settings() { (void)setenv("HIP_FORCE_DEV_KERNARG", "1", 0); }

// ---------------------------------------------------------------------------------------
// context
// ---------------------------------------------------------------------------------------
extern "C" int mg_context_create(int device, void *stream, mg_context **out) {
    MG_REQUIRE(out != nullptr, "mg_context_create: out is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        mg_set_error("mg_context_create: no HIP device available (%s)", e == hipSuccess ? "count = 0" : hipGetErrorString(e));
        return MG_ERR_NO_DEVICE;
    }
    MG_REQUIRE(device >= 0 && device < n, "mg_context_create: device %d out of range [0,%d)", device, n);
    MG_HIP_CHECK(hipSetDevice(device));
    mg_context *ctx = new (std::nothrow) mg_context();
    if (!ctx) return MG_ERR_OUT_OF_MEMORY;
    ctx->device = device;
    hipDeviceProp_t prop;
    e = hipGetDeviceProperties(&prop, device);
    if (e != hipSuccess) { delete ctx; return mg_hip_fail(e, "hipGetDeviceProperties"); }
    ctx->n_cu = prop.multiProcessorCount;
    ctx->total_mem = (int64_t)prop.totalGlobalMem;
    ctx->max_lds = (int)prop.maxSharedMemoryPerMultiProcessor;
    // (some boxes of the pool report an empty marketing name: the architecture and the CU count always identify the part)
    snprintf(ctx->name, sizeof(ctx->name), "%s (%s, %d CUs)", prop.name[0] ? prop.name : "AMD Instinct", prop.gcnArchName, prop.multiProcessorCount);
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        mg_set_error("mg_context_create: device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
        delete ctx;
        return MG_ERR_UNSUPPORTED;
    }
    if (stream) {
        ctx->stream = (hipStream_t)stream;
        ctx->own_stream = false;
    } else {
        e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { delete ctx; return mg_hip_fail(e, "hipStreamCreate"); }
        ctx->own_stream = true;
    }
    e = hipMalloc(&ctx->argmin_out, 16);
    if (e != hipSuccess) { mg_context_destroy(ctx); return mg_hip_fail(e, "hipMalloc"); }
    int rc = mg_setup_kernel_attributes(ctx);
    if (rc == MG_OK) rc = mg_options_fused_attributes();
    if (rc != MG_OK) { mg_context_destroy(ctx); return rc; }
    *out = ctx;
    return MG_OK;
}

static void mg_vmm_release(mg_context *ctx, mg_context::vmm_alloc &v);
extern "C" void mg_context_destroy(mg_context *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->rccl_comm && g_rccl.CommDestroy) { (void)g_rccl.CommDestroy(ctx->rccl_comm); ctx->rccl_comm = nullptr; }
    for (auto &p : ctx->pending) { (void)hipEventDestroy(p.a); (void)hipEventDestroy(p.b); }
    for (auto &ev : ctx->free_events) (void)hipEventDestroy(ev);
    for (auto &st : ctx->side) if (st) (void)hipStreamDestroy(st);
    for (auto &e : ctx->side_ev) if (e) (void)hipEventDestroy(e);
    if (ctx->scratch) (void)mg_device_free(ctx, ctx->scratch);
    mg_output_release_all(ctx);
    if (ctx->argmin_out) (void)hipFree(ctx->argmin_out);
    for (void *q : {ctx->fused_tab_dev, ctx->fused_counters, ctx->fused_partials, ctx->fused_dyn_dev}) if (q) (void)hipFree(q);
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->flag_block) (void)hipHostFree(ctx->flag_block);
    if (ctx->traj_paths) (void)hipFree(ctx->traj_paths);
    if (ctx->lists_dev) (void)hipFree(ctx->lists_dev);
    for (auto &b : ctx->arena) (void)hipFree(b.base);
    for (auto &v : ctx->vmm) mg_vmm_release(ctx, v);
    (void)hipDeviceSynchronize();   // nothing of this process is in flight when the parked address ranges go back to the runtime
    for (auto &r : ctx->vmm_parked) (void)hipMemAddressFree(r.first, r.second);
    ctx->vmm_parked.clear();
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" int mg_context_set_stream(mg_context *ctx, void *stream) {
    MG_REQUIRE(ctx != nullptr, "mg_context_set_stream: ctx is NULL");
    MG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->own_stream && ctx->stream) (void)hipStreamDestroy(ctx->stream);
    if (stream) {
        ctx->stream = (hipStream_t)stream;
        ctx->own_stream = false;
    } else {
        MG_HIP_CHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->own_stream = true;
    }
    return MG_OK;
}

extern "C" int mg_context_arena_begin(mg_context *ctx, int64_t block_bytes) {
    MG_REQUIRE(ctx != nullptr && block_bytes >= 0, "mg_context_arena_begin: bad arguments");
    ctx->arena_block_bytes = block_bytes > 0 ? (size_t)block_bytes : ((size_t)64 << 20);
    return MG_OK;
}
extern "C" int mg_context_arena_end(mg_context *ctx) {
    MG_REQUIRE(ctx != nullptr, "mg_context_arena_end: ctx is NULL");
    ctx->arena_block_bytes = 0;
    return MG_OK;
}
extern "C" int mg_context_arena_bytes(mg_context *ctx, int64_t *reserved, int64_t *used) {
    MG_REQUIRE(ctx != nullptr, "mg_context_arena_bytes: ctx is NULL");
    int64_t r = 0, u = 0;
    for (auto &b : ctx->arena) { r += (int64_t)b.bytes; u += (int64_t)b.used; }
    if (reserved) *reserved = r;
    if (used) *used = u;
    return MG_OK;
}
static int mg_arena_alloc(mg_context *ctx, size_t bytes, void **out) {
    const size_t need = (bytes + 255) / 256 * 256;   // every array starts on a 256-byte boundary
    if (ctx->arena.empty() || ctx->arena.back().used + need > ctx->arena.back().bytes) {
        mg_context::arena_block b;
        b.bytes = std::max(ctx->arena_block_bytes, need);
        b.used = 0;
        b.live = 0;
        MG_HIP_CHECK(hipMalloc((void **)&b.base, b.bytes));
        ctx->arena.push_back(b);
    }
    *out = ctx->arena.back().base + ctx->arena.back().used;
    ctx->arena.back().used += need;
    ctx->arena.back().live++;
    return MG_OK;
}
void mg_dev_free(mg_context *ctx, void *p) {
    if (!p) return;
    for (size_t i = 0; i < ctx->arena.size(); i++) {
        mg_context::arena_block &b = ctx->arena[i];
        if ((char *)p >= b.base && (char *)p < b.base + b.bytes) {
            // the last array of a block that is no longer being filled takes the block with it
            if (--b.live == 0 && !(ctx->arena_block_bytes > 0 && i + 1 == ctx->arena.size())) {
                (void)hipFree(b.base);
                ctx->arena.erase(ctx->arena.begin() + (long)i);
            }
            return;
        }
    }
    (void)hipFree(p);
}

extern "C" int mg_dist_unique_id(void *id_out) {
    MG_REQUIRE(id_out != nullptr, "mg_dist_unique_id: id_out is NULL");
    int rc = mg_rccl_load();
    if (rc != MG_OK) return rc;
    int nrc = g_rccl.GetUniqueId(id_out);
    return nrc == 0 ? MG_OK : mg_rccl_fail(nrc, "ncclGetUniqueId");
}
// What can fail on ONE rank before the collective set-up: librccl and its entry points, the context's device.  A rank tells the
// others how this went BEFORE any of them enters ncclCommInitRank, where a rank left alone would wait for good.
extern "C" int mg_dist_preflight(mg_context *ctx) {
    MG_REQUIRE(ctx != nullptr, "mg_dist_preflight: ctx is NULL");
    int rc = mg_rccl_load();
    if (rc != MG_OK) return rc;
    MG_HIP_CHECK(hipSetDevice(ctx->device));
    MG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MG_OK;
}
// The communicator as RCCL itself sees it: out[0] = this rank (ncclCommUserRank), out[1] = ranks (ncclCommCount), out[2] = the
// device RCCL bound it to (ncclCommCuDevice); -1 where the installed librccl lacks the query.  {-1, 0, -1} without a communicator.
extern "C" int mg_dist_info(mg_context *ctx, int32_t *out3) {
    MG_REQUIRE(ctx && out3, "mg_dist_info: NULL argument");
    out3[0] = -1; out3[1] = 0; out3[2] = -1;
    if (!ctx->rccl_comm || !g_rccl.lib) return MG_OK;
    typedef int (*q_fn)(void *, int *);
    const char *names[3] = {"ncclCommUserRank", "ncclCommCount", "ncclCommCuDevice"};
    for (int i = 0; i < 3; i++) {
        q_fn f = (q_fn)dlsym(g_rccl.lib, names[i]);
        int v = -1;
        if (f && f(ctx->rccl_comm, &v) == 0) out3[i] = v; else out3[i] = -1;
    }
    return MG_OK;
}
extern "C" int mg_dist_init(mg_context *ctx, int32_t rank, int32_t n_ranks, const void *id) {
    MG_REQUIRE(ctx && id && n_ranks >= 1 && rank >= 0 && rank < n_ranks, "mg_dist_init: bad arguments");
    MG_REQUIRE(ctx->rccl_comm == nullptr, "mg_dist_init: the context already has a communicator");
    int rc = mg_rccl_load();
    if (rc != MG_OK) return rc;
    MG_HIP_CHECK(hipSetDevice(ctx->device));
    mg_nccl_id uid;
    memcpy(uid.internal, id, MG_DIST_ID_BYTES);
    int nrc = g_rccl.CommInitRank(&ctx->rccl_comm, n_ranks, uid, rank);
    if (nrc != 0) { ctx->rccl_comm = nullptr; return mg_rccl_fail(nrc, "ncclCommInitRank"); }
    ctx->dist_rank = rank;
    ctx->dist_ranks = n_ranks;
    return MG_OK;
}
extern "C" int mg_dist_all_gather(mg_context *ctx, const void *local_dev, void *gathered_dev, int64_t count, int dtype) {
    MG_REQUIRE(ctx && count >= 0 && (count == 0 || (local_dev && gathered_dev)), "mg_dist_all_gather: bad arguments");
    MG_REQUIRE(dtype == MG_F32 || dtype == MG_F64, "mg_dist_all_gather: bad dtype %d", dtype);
    MG_REQUIRE(ctx->rccl_comm != nullptr, "mg_dist_all_gather: mg_dist_init has not been called on this context");
    if (count == 0) return MG_OK;
    MG_HIP_CHECK(hipSetDevice(ctx->device));
    int nrc = g_rccl.AllGather(local_dev, gathered_dev, (size_t)count, dtype == MG_F64 ? 8 /* ncclFloat64 */ : 7 /* ncclFloat32 */,
                               ctx->rccl_comm, ctx->stream);
    return nrc == 0 ? MG_OK : mg_rccl_fail(nrc, "ncclAllGather");
}
extern "C" int mg_dist_broadcast(mg_context *ctx, void *buf_dev, int64_t bytes, int32_t root) {
    MG_REQUIRE(ctx && ctx->rccl_comm, "mg_dist_broadcast: mg_dist_init has not been called on this context");
    MG_REQUIRE(bytes >= 0 && (bytes == 0 || buf_dev) && root >= 0 && root < ctx->dist_ranks, "mg_dist_broadcast: bad arguments");
    if (bytes == 0) return MG_OK;
    MG_HIP_CHECK(hipSetDevice(ctx->device));
    const int rcb = g_rccl.Broadcast(buf_dev, buf_dev, (size_t)bytes, 0 /* ncclInt8 */, root, ctx->rccl_comm, ctx->stream);
    return rcb == 0 ? MG_OK : mg_rccl_fail(rcb, "ncclBroadcast");
}
extern "C" int mg_dist_finalize(mg_context *ctx) {
    MG_REQUIRE(ctx != nullptr, "mg_dist_finalize: ctx is NULL");
    if (!ctx->rccl_comm) return MG_OK;
    MG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    int nrc = g_rccl.CommDestroy(ctx->rccl_comm);
    ctx->rccl_comm = nullptr;
    ctx->dist_rank = 0; ctx->dist_ranks = 1;
    return nrc == 0 ? MG_OK : mg_rccl_fail(nrc, "ncclCommDestroy");
}

extern "C" int mg_context_synchronize(mg_context *ctx) {
    MG_REQUIRE(ctx != nullptr, "mg_context_synchronize: ctx is NULL");
    MG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MG_OK;
}

extern "C" int mg_context_set_reserved_cus(mg_context *ctx, int32_t n) {
    MG_REQUIRE(ctx != nullptr && n >= 0, "mg_context_set_reserved_cus: bad arguments");
    ctx->reserved_cus = std::min<int32_t>(n, ctx->n_cu - 1);
    return MG_OK;
}
extern "C" int mg_context_set_option(mg_context *ctx, int32_t option, int32_t value) {
    MG_REQUIRE(ctx != nullptr, "mg_context_set_option: ctx is NULL");
    MG_REQUIRE(option >= 0 && option < MG_OPT_COUNT, "mg_context_set_option: unknown option %d", option);
    MG_REQUIRE(value >= 0, "mg_context_set_option: value %d < 0", value);
    ctx->opt[option] = value;
    return MG_OK;
}
extern "C" int mg_context_device_info(mg_context *ctx, char *name, int32_t *n_cu, int64_t *total_mem) {
    MG_REQUIRE(ctx != nullptr, "mg_context_device_info: ctx is NULL");
    if (name) { strncpy(name, ctx->name, 255); name[255] = 0; }
    if (n_cu) *n_cu = ctx->n_cu;
    if (total_mem) *total_mem = ctx->total_mem;
    return MG_OK;
}

extern "C" int mg_device_malloc(mg_context *ctx, int64_t bytes, void **out_dev) {
    MG_REQUIRE(ctx && out_dev && bytes >= 0, "mg_device_malloc: bad arguments");
    *out_dev = nullptr;
    MG_HIP_CHECK(hipSetDevice(ctx->device));
    if (bytes == 0) bytes = 16;
    // large buffers are kernel outputs: a piece of a placed region (mg_placement.hip) unless the caller opted out
    if (bytes >= MG_PLACED_MIN_BYTES && !ctx->opt[MG_OPT_PLAIN_MALLOC]) return mg_output_alloc(ctx, bytes, 0, out_dev, nullptr);
    MG_HIP_CHECK(hipMalloc(out_dev, (size_t)bytes));
    return MG_OK;
}
// Unmap and release the physical chunks; the ADDRESS RANGE stays reserved (parked with the context) so that no later
// reservation of this process can land on it while the context lives (see mg_context::vmm_parked).
static void mg_vmm_release(mg_context *ctx, mg_context::vmm_alloc &v) {
    if (v.va) {
        (void)hipMemUnmap(v.va, v.total);
        ctx->vmm_parked.push_back({v.va, v.total});
    }
    for (auto h : v.handles) (void)hipMemRelease(h);
    v.handles.clear();
    v.va = nullptr;
}
extern "C" int mg_device_malloc_chunked(mg_context *ctx, int64_t bytes, int64_t chunk_bytes, void **out_dev) {
    MG_REQUIRE(ctx && out_dev && bytes > 0 && chunk_bytes > 0, "mg_device_malloc_chunked: bad arguments");
    *out_dev = nullptr;
    MG_HIP_CHECK(hipSetDevice(ctx->device));
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = ctx->device;
    size_t gran = 0;
    MG_HIP_CHECK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum));
    if (gran == 0) gran = 4096;
    const size_t chunk = ((size_t)chunk_bytes + gran - 1) / gran * gran;
    const size_t n = ((size_t)bytes + chunk - 1) / chunk;
    mg_context::vmm_alloc v;
    v.total = n * chunk; v.chunk = chunk; v.va = nullptr;
    hipError_t e = hipMemAddressReserve(&v.va, v.total, 0, nullptr, 0);
    if (e != hipSuccess) return mg_hip_fail(e, "hipMemAddressReserve");
    size_t mapped = 0;
    for (size_t i = 0; i < n && e == hipSuccess; i++) {
        hipMemGenericAllocationHandle_t h;
        e = hipMemCreate(&h, chunk, &prop, 0);
        if (e != hipSuccess) break;
        v.handles.push_back(h);
        e = hipMemMap((char *)v.va + i * chunk, chunk, 0, h, 0);
        if (e == hipSuccess) mapped += chunk;
    }
    if (e == hipSuccess) {
        hipMemAccessDesc acc = {};
        acc.location = prop.location;
        acc.flags = hipMemAccessFlagsProtReadWrite;
        e = hipMemSetAccess(v.va, v.total, &acc, 1);
    }
    if (e != hipSuccess) {
        if (mapped) (void)hipMemUnmap(v.va, mapped);
        (void)hipMemAddressFree(v.va, v.total);
        for (auto h : v.handles) (void)hipMemRelease(h);
        return mg_hip_fail(e, "mg_device_malloc_chunked (virtual-memory API)");
    }
    *out_dev = v.va;
    ctx->vmm.push_back(std::move(v));
    return MG_OK;
}
extern "C" int mg_device_free(mg_context *ctx, void *p) {
    MG_REQUIRE(ctx != nullptr, "mg_device_free: ctx is NULL");
    if (!p) return MG_OK;
    MG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (mg_output_free(ctx, p)) return MG_OK;   // a piece of a placed region: back to its free list, the region stays
    return mg_device_free_raw(ctx, p);
}
int mg_device_free_raw(mg_context *ctx, void *p) {
    for (size_t i = 0; i < ctx->vmm.size(); i++)
        if (ctx->vmm[i].va == p) {
            mg_vmm_release(ctx, ctx->vmm[i]);
            ctx->vmm.erase(ctx->vmm.begin() + (long)i);
            return MG_OK;
        }
    MG_HIP_CHECK(hipFree(p));
    return MG_OK;
}
extern "C" int mg_memcpy_h2d(mg_context *ctx, void *dst, const void *src, int64_t bytes) {
    MG_REQUIRE(ctx && (bytes == 0 || (dst && src)) && bytes >= 0, "mg_memcpy_h2d: bad arguments");
    if (bytes == 0) return MG_OK;
    MG_HIP_CHECK(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, ctx->stream));
    MG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MG_OK;
}
extern "C" int mg_memcpy_d2h(mg_context *ctx, void *dst, const void *src, int64_t bytes) {
    MG_REQUIRE(ctx && (bytes == 0 || (dst && src)) && bytes >= 0, "mg_memcpy_d2h: bad arguments");
    if (bytes == 0) return MG_OK;
    MG_HIP_CHECK(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, ctx->stream));
    MG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return MG_OK;
}
extern "C" int mg_memset(mg_context *ctx, void *dst, int value, int64_t bytes) {
    MG_REQUIRE(ctx && (bytes == 0 || dst) && bytes >= 0, "mg_memset: bad arguments");
    if (bytes == 0) return MG_OK;
    MG_HIP_CHECK(hipMemsetAsync(dst, value, (size_t)bytes, ctx->stream));
    return MG_OK;
}

int mg_ctx_scratch(mg_context *ctx, int64_t bytes, void **out) {
    if (bytes > ctx->scratch_bytes) {
        MG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (ctx->scratch) { int rcf = mg_device_free(ctx, ctx->scratch); ctx->scratch = nullptr; ctx->scratch_bytes = 0; if (rcf != MG_OK) return rcf; }
        int64_t want = std::max<int64_t>(bytes, 1 << 20);
        int rcm = mg_device_malloc(ctx, want, &ctx->scratch);   // large scratch blocks hold the *_host entry points' outputs: placed like them
        if (rcm != MG_OK) return rcm;
        ctx->scratch_bytes = want;
    }
    *out = ctx->scratch;
    return MG_OK;
}

// ---------------------------------------------------------------------------------------
// profiling with HIP events on the context's stream
// ---------------------------------------------------------------------------------------
static hipEvent_t mg_get_event(mg_context *ctx) {
    if (!ctx->free_events.empty()) {
        hipEvent_t e = ctx->free_events.back();
        ctx->free_events.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    // timing only: no system-scope fence (cache write-back and invalidation) when the event completes
    (void)hipEventCreateWithFlags(&e, hipEventDisableSystemFence);
    return e;
}
void mg_prof_begin(mg_context *ctx, int slot) {
    if (!ctx->profile) return;
    if (ctx->prof_interval > 1 && (ctx->prof_seen[slot]++ % ctx->prof_interval) != 0) return;   // sampled bracketing
    if (ctx->pending.size() > 4096) (void)mg_prof_resolve(ctx);
    mg_event_pair p;
    p.a = mg_get_event(ctx);
    p.b = mg_get_event(ctx);
    p.slot = slot;
    p.ended = false;
    (void)hipEventRecord(p.a, ctx->stream);
    ctx->pending.push_back(p);
}
bool mg_prof_kernel(mg_context *ctx, int slot, int slot2, hipEvent_t *start, hipEvent_t *stop) {
    if (!ctx->profile) return false;
    if (ctx->prof_interval > 1 && (ctx->prof_seen[slot]++ % ctx->prof_interval) != 0) return false;
    if (ctx->pending.size() > 4096) (void)mg_prof_resolve(ctx);
    mg_event_pair p;
    p.a = mg_get_event(ctx);
    p.b = mg_get_event(ctx);
    p.slot = slot;
    p.slot2 = slot2;
    p.ended = true;
    ctx->pending.push_back(p);
    *start = p.a;
    *stop = p.b;
    return true;
}
void mg_prof_end(mg_context *ctx, int slot) {
    if (!ctx->profile) return;
    for (size_t i = ctx->pending.size(); i-- > 0;) {   // brackets may nest (step > frames)
        mg_event_pair &p = ctx->pending[i];
        if (p.slot == slot && !p.ended) {
            (void)hipEventRecord(p.b, ctx->stream);
            p.ended = true;
            return;
        }
    }
}
int mg_prof_resolve(mg_context *ctx) {
    for (auto &p : ctx->pending) {
        float ms = 0.f;
        if (p.ended && hipEventSynchronize(p.b) == hipSuccess && hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) {
            for (int sl : {p.slot, p.slot2}) {
                if (sl < 0) continue;
                ctx->prof_ms[sl] += (double)ms;
                ctx->prof_n[sl] += 1;
                if (ctx->prof_samples[sl].size() < 65536) ctx->prof_samples[sl].push_back(ms);
            }
        }
        ctx->free_events.push_back(p.a);
        ctx->free_events.push_back(p.b);
    }
    ctx->pending.clear();
    return MG_OK;
}
extern "C" int mg_profile_enable(mg_context *ctx, int enabled) {
    MG_REQUIRE(ctx != nullptr, "mg_profile_enable: ctx is NULL");
    if (!enabled) (void)mg_prof_resolve(ctx);
    ctx->profile = enabled != 0;
    ctx->prof_interval = enabled > 1 ? enabled : 1;
    for (int i = 0; i < MG_PROFILE_SLOTS; i++) ctx->prof_seen[i] = 0;
    return MG_OK;
}
extern "C" int mg_profile_reset(mg_context *ctx) {
    MG_REQUIRE(ctx != nullptr, "mg_profile_reset: ctx is NULL");
    (void)mg_prof_resolve(ctx);
    for (int i = 0; i < MG_PROFILE_SLOTS; i++) { ctx->prof_ms[i] = 0; ctx->prof_n[i] = 0; ctx->prof_samples[i].clear(); }
    return MG_OK;
}
extern "C" int mg_profile_get(mg_context *ctx, int slot, double *total_ms, int64_t *launches) {
    MG_REQUIRE(ctx && slot >= 0 && slot < MG_PROFILE_SLOTS, "mg_profile_get: bad arguments");
    (void)mg_prof_resolve(ctx);
    if (total_ms) *total_ms = ctx->prof_ms[slot];
    if (launches) *launches = ctx->prof_n[slot];
    return MG_OK;
}
extern "C" int mg_profile_get_samples(mg_context *ctx, int slot, float *out_ms, int64_t capacity, int64_t *n) {
    MG_REQUIRE(ctx && slot >= 0 && slot < MG_PROFILE_SLOTS && capacity >= 0 && (capacity == 0 || out_ms) && n,
               "mg_profile_get_samples: bad arguments");
    (void)mg_prof_resolve(ctx);
    const std::vector<float> &v = ctx->prof_samples[slot];
    const int64_t m = std::min<int64_t>(capacity, (int64_t)v.size());
    for (int64_t i = 0; i < m; i++) out_ms[i] = v[(size_t)i];
    *n = m;
    return MG_OK;
}

// ---------------------------------------------------------------------------------------
// float64 host helpers
// ---------------------------------------------------------------------------------------
// FITPACK splev.f interval search + fpbspl.f recurrence (scipy.interpolate.splev, ext=0),
// the arithmetic behind reference motion_spline.py:86,92.
void mg_basis_row(const double *t, int n, double x, int32_t *i0, double *h) {
    const int k = 3;
    int l = k;
    while (!(x < t[l + 1] || l == n - k - 2)) l++;
    double hh[4];
    h[0] = 1.0; h[1] = h[2] = h[3] = 0.0;
    for (int j = 1; j <= k; j++) {
        for (int i = 0; i < j; i++) hh[i] = h[i];
        h[0] = 0.0;
        for (int i = 1; i <= j; i++) {
            int li = l + i, lj = li - j;
            if (t[li] == t[lj]) { h[i] = 0.0; continue; }
            double f = hh[i - 1] / (t[li] - t[lj]);
            h[i - 1] = h[i - 1] + f * (t[li] - x);
            h[i] = f * (x - t[lj]);
        }
    }
    *i0 = l - k;
}

// lower Cholesky factor; returns false if not positive definite
static bool mg_cholesky_lower(const double *S, int L, double *c) {
    std::fill(c, c + (size_t)L * L, 0.0);
    for (int j = 0; j < L; j++) {
        double sum = S[j * L + j];
        for (int p = 0; p < j; p++) sum -= c[j * L + p] * c[j * L + p];
        if (!(sum > 0.0) || !std::isfinite(sum)) return false;
        c[j * L + j] = std::sqrt(sum);
        for (int i = j + 1; i < L; i++) {
            double s2 = S[i * L + j];
            for (int p = 0; p < j; p++) s2 -= c[i * L + p] * c[j * L + p];
            c[i * L + j] = s2 / c[j * L + j];
        }
    }
    return true;
}

template <typename T>
static int mg_upload(mg_context *ctx, const std::vector<T> &h, T **d) {
    *d = nullptr;
    size_t bytes = std::max<size_t>(h.size() * sizeof(T), 16);
    if (ctx->arena_block_bytes > 0) {
        int rc = mg_arena_alloc(ctx, bytes, (void **)d);
        if (rc != MG_OK) return rc;
    } else {
        MG_HIP_CHECK(hipMalloc((void **)d, bytes));
    }
    if (!h.empty()) MG_HIP_CHECK(hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return MG_OK;
}

// ---------------------------------------------------------------------------------------
// time grids
// ---------------------------------------------------------------------------------------
static void mg_free_grid_device(mg_time_grid *g) {
    mg_context *ctx = g->prim->ctx;
    mg_dev_free(ctx, g->d_i0);
    mg_dev_free(ctx, g->d_w);
    mg_dev_free(ctx, g->d_w32);
    mg_dev_free(ctx, g->d_rootm);
    g->d_rootm = nullptr;
    mg_dev_free(ctx, g->d_wtap);
    g->d_wtap = nullptr;
    mg_dev_free(ctx, g->d_chunks);
    g->d_i0 = nullptr; g->d_w = nullptr; g->d_w32 = nullptr; g->d_chunks = nullptr;
}

static int mg_round_stride(int nlocal) {
    // floats per candidate in the LDS coefficient image: == 4 (mod 32) makes the
    // 16-candidate ds_write_b128 of an MFMA accumulator tile conflict-free.
    int s = nlocal;
    while (s % 32 != 4) s++;
    return s;
}

// LDS of the persistent kernel: two buffers (coefficient image + float32 root outputs),
// three per-sample table sets, two float64 root images.
static int mg_lds_bytes(const mg_primitive *p, int stride, int wi, int nbuf = 2, int max_nt = MG_MAX_NT) {
    int buf = (MG_NCAND * stride * 4 + 255) / 256 * 256;
    int rout = MG_RO_BYTES_N(max_nt);
    int tabs = max_nt * 16 + max_nt * 4;
    int root = MG_NCAND * (wi * p->nroot + 1) * 8;
    return nbuf * (buf + rout + tabs) + root + 128;
}

// Split the grid into chunks (runs of consecutive time samples) whose coefficient window
// fits the LDS budget.
static void mg_plan_chunks(mg_primitive *p, mg_time_grid *g) {
    const int Dp = p->Dp;
    g->chunks.clear();
    g->mfma_ok = false;
    if (p->KK == 0 || g->T == 0 || (p->D - p->nroot + 3) / 4 + 1 > 64) return;   // one row group must fit a wave
    // widest window (<= 8 basis functions) whose double-buffered image fits one CU's 160 KiB of LDS
    const int budget1 = 160 * 1024;
    auto fits = [&](int wi, int budget) {
        int nlocal = ((wi * Dp + 15) / 16 + 1) * 16;
        return mg_lds_bytes(p, mg_round_stride(nlocal), wi) <= budget;
    };
    int max_nt = MG_MAX_NT;
    if (const int o = p->ctx->opt[MG_OPT_CHUNK_SAMPLES]) max_nt = std::max(1, std::min(MG_MAX_NT, o));
    auto count_chunks = [&](int w) {   // greedy split under a window of w basis functions
        int n = 0;
        for (int a0 = 0; a0 < g->T; n++) {
            int imin = g->i0[a0], imax = g->i0[a0], b = a0 + 1;
            while (b < g->T && b - a0 < max_nt) {
                int lo = std::min(imin, g->i0[b]), hi = std::max(imax, g->i0[b]);
                if (hi - lo + 4 > w) break;
                imin = lo; imax = hi;
                b++;
            }
            a0 = b;
        }
        return n;
    };
    // Fewer, longer chunks win (each unit has fixed costs -- measured: 6 -> 4 chunks of the 'walk' grid -3 %, while a
    // two-slot ring instead of three costs 0.5 %): the narrowest window that reaches the smallest chunk count among
    // the windows whose two-slot ring fits LDS.
    int W = 0, best_n = 0;
    for (int w = MG_MAX_WI; w >= 4; w--) {
        if (!fits(w, budget1)) continue;
        const int n = count_chunks(w);
        if (W == 0 || n <= best_n) { W = w; best_n = n; }
    }
    if (W == 0) return;  // n_dim too large for the LDS-staged kernel
    if (const int o = p->ctx->opt[MG_OPT_CHUNK_WINDOW]) W = std::max(4, std::min(W, o));
    // the chunk count of the greedy split, evened out
    const int n_greedy = count_chunks(W);
    int a = 0;
    int max_stride = 0, max_wi = 0, longest = 0;
    while (a < g->T) {
        int imin = g->i0[a], imax = g->i0[a];
        int b = a + 1;
        const int left = std::max(1, n_greedy - (int)g->chunks.size());
        const int target = std::min(max_nt, (g->T - a + left - 1) / left);
        while (b < g->T && b - a < target) {
            int lo = std::min(imin, g->i0[b]), hi = std::max(imax, g->i0[b]);
            if (hi - lo + 4 > W) break;
            imin = lo; imax = hi;
            b++;
        }
        mg_chunk c;
        memset(&c, 0, sizeof(c));
        c.t0 = a;
        c.nT = b - a;
        c.imin = imin;
        c.wi = imax - imin + 4;
        c.rt0 = (imin * Dp) / 16;
        int rt1 = ((imin + c.wi) * Dp + 15) / 16;
        c.ntiles = rt1 - c.rt0;
        c.rrt0 = (imin * p->nroot) / 16;
        c.nrt = ((imin + c.wi) * p->nroot + 15) / 16 - c.rrt0;
        g->chunks.push_back(c);
        max_stride = std::max(max_stride, mg_round_stride(c.ntiles * 16));
        max_wi = std::max(max_wi, c.wi);
        longest = std::max(longest, c.nT);
        a = b;
    }
    g->stride = max_stride;
    g->max_wi = max_wi;
    g->max_nt = (longest + 15) / 16 * 16;
    // a third ring slot lets the producers run a full unit ahead of the slowest consumer wave
    g->nbuf = mg_lds_bytes(p, max_stride, max_wi, 3, g->max_nt) <= budget1 ? 3 : 2;
    if (p->ctx->opt[MG_OPT_RING_SLOTS] == 2) g->nbuf = 2;
    g->lds_bytes = mg_lds_bytes(p, max_stride, max_wi, g->nbuf, g->max_nt);
    g->n_chunks = (int32_t)g->chunks.size();
    g->mfma_ok = g->lds_bytes <= budget1;
    // the chunk-stationary kernel: two ring slots, ONE table set, the window's mean' rows; its row producers hold the
    // window's eigenvector fragments in registers
    g->max_tiles = 0;
    for (const mg_chunk &c : g->chunks) g->max_tiles = std::max(g->max_tiles, c.ntiles);
    {
        const int buf = (MG_NCAND * max_stride * 4 + 255) / 256 * 256;
        g->cs_lds_bytes = 2 * (buf + MG_RO_BYTES_N(g->max_nt)) + g->max_nt * 20 + MG_NCAND * (max_wi * p->nroot + 1) * 8 + g->max_tiles * 64 +
                          MG_TAP_FT * MG_TAP_KS * 64 * 8 + 512 + 2 * p->KK * 64 * 4 + 256;   // + tap weights, root means, two latent tiles, 64 counters
        g->cs_ok = g->mfma_ok && g->max_tiles <= mg_cs_max_tiles(p->KK) && g->cs_lds_bytes <= budget1 && (int)g->chunks.size() <= MG_ARG_CHUNKS;
    }
}

static int mg_grid_build(mg_primitive *p, mg_time_grid *g, const double *times, int32_t T,
                         const int32_t *i0_override, const double *w_override) {
    g->prim = p;
    g->T = T;
    g->times.assign(times, times + T);
    g->i0.resize(T);
    g->w.resize((size_t)T * 4);
    for (int f = 0; f < T; f++) {
        if (i0_override) {
            g->i0[f] = i0_override[f];
            for (int j = 0; j < 4; j++) g->w[4 * f + j] = w_override[4 * f + j];
        } else {
            mg_basis_row(p->knots.data(), (int)p->knots.size(), times[f], &g->i0[f], &g->w[4 * f]);
        }
    }
    mg_plan_chunks(p, g);
    std::vector<float> w32(g->w.size());
    for (size_t q = 0; q < g->w.size(); q++) w32[q] = (float)g->w[q];
    MG_HIP_CHECK(hipSetDevice(p->ctx->device));
    int rc;
    if ((rc = mg_upload(p->ctx, g->i0, &g->d_i0)) != MG_OK) return rc;
    if ((rc = mg_upload(p->ctx, g->w, &g->d_w)) != MG_OK) return rc;
    if ((rc = mg_upload(p->ctx, w32, &g->d_w32)) != MG_OK) return rc;
    {   // the root channels' mean part (mean/delta split, mg_primitive_root_mode): M[f][d] = w0 m0, fma(w1, m1, .), fma(w2, m2, .),
        // fma(w3, m3, .) in float64 with m_j = mean'[(i0[f] + j) D + d]; Mhi = (float)M, Mlo = (float)(M - Mhi)
        std::vector<float> rootm((size_t)std::max(T, 1) * 8, 0.0f);
        for (int f = 0; f < T; f++)
            for (int d = 0; d < p->nroot; d++) {
                const double *m = &p->means_[(size_t)g->i0[f] * p->D + d];
                double M = g->w[4 * (size_t)f] * m[0];
                for (int j = 1; j < 4; j++) M = std::fma(g->w[4 * (size_t)f + j], m[(size_t)j * p->D], M);
                const float hi = (float)M;
                rootm[8 * (size_t)f + d] = hi;                              // {Mhi[0..2], Mlo[0], Mlo[1], Mlo[2], 0, 0}: the sweep reads
                rootm[8 * (size_t)f + 3 + d] = (float)(M - (double)hi);    // a float4 and a float2 per sample, every element used
            }
        if ((rc = mg_upload(p->ctx, rootm, &g->d_rootm)) != MG_OK) return rc;
    }
    if ((rc = mg_upload(p->ctx, g->chunks, &g->d_chunks)) != MG_OK) return rc;
    {   // banded tap weights of every chunk as v_mfma_f64_16x16x4_f64 A fragments: lane l supplies
        // W[f = 16 ft + (l & 15)][m = 4 ks + (l >> 4)],  W[f][m] = w[f][m - (i0[f] - imin)] inside the band
        std::vector<double> wtap(std::max<size_t>(g->chunks.size(), 1) * MG_TAP_FT * MG_TAP_KS * 64, 0.0);
        for (size_t c = 0; c < g->chunks.size(); c++) {
            const mg_chunk &ck = g->chunks[c];
            for (int ft = 0; ft < MG_TAP_FT; ft++)
                for (int ks = 0; ks < MG_TAP_KS; ks++)
                    for (int lane = 0; lane < 64; lane++) {
                        int f = ft * 16 + (lane & 15), m = 4 * ks + (lane >> 4);
                        if (f >= ck.nT) continue;
                        int j = m - (g->i0[ck.t0 + f] - ck.imin);
                        if (j >= 0 && j < 4) wtap[((c * MG_TAP_FT + ft) * MG_TAP_KS + ks) * 64 + lane] = g->w[4 * (size_t)(ck.t0 + f) + j];
                    }
        }
        if ((rc = mg_upload(p->ctx, wtap, &g->d_wtap)) != MG_OK) return rc;
    }
    return MG_OK;
}

extern "C" int mg_time_grid_create(mg_primitive *prim, const double *times, int32_t n_times, mg_time_grid **out) {
    MG_REQUIRE(prim && out && n_times >= 0 && (n_times == 0 || times), "mg_time_grid_create: bad arguments");
    *out = nullptr;
    for (int i = 0; i < n_times; i++) MG_REQUIRE(std::isfinite(times[i]), "mg_time_grid_create: times[%d] is not finite", i);
    mg_time_grid *g = new (std::nothrow) mg_time_grid();
    if (!g) return MG_ERR_OUT_OF_MEMORY;
    int rc = mg_grid_build(prim, g, times, n_times, nullptr, nullptr);
    if (rc != MG_OK) { mg_free_grid_device(g); delete g; return rc; }
    *out = g;
    return MG_OK;
}
extern "C" void mg_time_grid_destroy(mg_time_grid *g) {
    if (!g || g->owned_by_prim) return;
    if (g->prim) (void)hipStreamSynchronize(g->prim->ctx->stream);
    mg_free_grid_device(g);
    delete g;
}
extern "C" mg_time_grid *mg_primitive_canonical_grid(mg_primitive *prim) { return prim ? prim->canonical : nullptr; }
extern "C" int mg_time_grid_size(const mg_time_grid *g) { return g ? g->T : 0; }
extern "C" int mg_time_grid_get_tables(const mg_time_grid *g, int32_t *i0, double *weights, double *times) {
    MG_REQUIRE(g != nullptr, "mg_time_grid_get_tables: grid is NULL");
    if (i0) std::copy(g->i0.begin(), g->i0.end(), i0);
    if (weights) std::copy(g->w.begin(), g->w.end(), weights);
    if (times) std::copy(g->times.begin(), g->times.end(), times);
    return MG_OK;
}

// ---------------------------------------------------------------------------------------
// primitive
// ---------------------------------------------------------------------------------------
static void mg_primitive_free(mg_primitive *p) {
    if (!p) return;
    (void)hipSetDevice(p->ctx->device);
    (void)hipStreamSynchronize(p->ctx->stream);
    void *ptrs[] = {p->d_Epack, p->d_Et32, p->d_Et64, p->d_Erpack, p->d_meanroot, p->d_mean32, p->d_mean, p->d_knots,
                    p->d_gP, p->d_gmP, p->d_gconst, p->d_gmean, p->d_gchol, p->d_gPpack, p->d_gmPpad, p->d_gPTpack, p->d_gcholpack, p->d_gmeanpad,
                    p->d_tphi, p->d_tmean};
    for (void *q : ptrs) mg_dev_free(p->ctx, q);
    for (mg_time_grid *g : {p->canonical, p->coeff_grid})
        if (g) { mg_free_grid_device(g); delete g; }
    delete p;
}

extern "C" void mg_primitive_destroy(mg_primitive *p) {
    if (p && p->ctx) p->ctx->fused_next.valid = false;   // (counts drawn ahead are keyed by the primitive's address)
    mg_primitive_free(p);
}

extern "C" int mg_primitive_create(mg_context *ctx, const mg_primitive_desc *d, mg_primitive **out) {
    MG_REQUIRE(ctx && d && out, "mg_primitive_create: NULL argument");
    *out = nullptr;
    MG_REQUIRE(d->n_basis >= 4, "mg_primitive_create: n_basis = %d, a cubic spline needs >= 4", d->n_basis);
    MG_REQUIRE(d->n_dim >= 1 && d->n_components >= 1 && d->n_canonical_frames >= 1,
               "mg_primitive_create: n_dim/n_components/n_canonical_frames must be >= 1");
    MG_REQUIRE(d->n_gmm >= 0, "mg_primitive_create: n_gmm < 0");
    MG_REQUIRE(d->eigen_vectors && d->mean_vector && d->knots, "mg_primitive_create: eigen_vectors/mean_vector/knots are required");
    MG_REQUIRE(d->n_gmm == 0 || (d->gmm_weights && d->gmm_means && d->gmm_covars),
               "mg_primitive_create: gmm arrays are required when n_gmm > 0");
    MG_REQUIRE((int64_t)d->n_basis * d->n_dim < (1 << 24), "mg_primitive_create: n_basis*n_dim too large");
    const int NB = d->n_basis, D = d->n_dim, L = d->n_components, K = d->n_gmm, R = NB * D;
    const int Lg = d->n_gmm_dims > 0 ? d->n_gmm_dims : L;   // the mixture spans the spatial AND the time latents
    MG_REQUIRE(Lg >= L, "mg_primitive_create: n_gmm_dims = %d < n_components = %d", Lg, L);
    const int Lt = d->n_time_components, NBt = d->n_basis_time;
    MG_REQUIRE(Lt >= 0 && (Lt == 0 || (NBt >= 4 && d->eigen_vectors_time && d->mean_time_vector && d->knots_time)),
               "mg_primitive_create: a time model needs n_basis_time >= 4, eigen_vectors_time, mean_time_vector and knots_time");
    for (int i = 0; i + 1 < NB + 4; i++)
        MG_REQUIRE(d->knots[i] <= d->knots[i + 1] && std::isfinite(d->knots[i + 1]), "mg_primitive_create: knots must be finite and non-decreasing");
    MG_REQUIRE(d->knots[3] < d->knots[NB], "mg_primitive_create: knot vector has an empty domain");
    MG_HIP_CHECK(hipSetDevice(ctx->device));

    mg_primitive *p = new (std::nothrow) mg_primitive();
    if (!p) return MG_ERR_OUT_OF_MEMORY;
    p->ctx = ctx;
    p->NB = NB; p->D = D; p->L = L; p->F = d->n_canonical_frames; p->K = K; p->R = R;
    p->nroot = std::min(3, D);
    p->KK = (L <= 4 * MG_MAX_KK) ? (((L + 3) / 4 + 1) / 2) * 2 : 0;
    p->Lg = Lg;
    p->KKg = (Lg <= 4 * MG_MAX_KK) ? (((Lg + 3) / 4 + 1) / 2) * 2 : 0;
    p->Lt = Lt; p->NBt = NBt;
    // padded coefficient rows: column = d + cshift so that the first non-root channel sits on a
    // 16-byte boundary (quads of channels start at d = nroot), pitch a multiple of 4
    p->cshift = (4 - p->nroot) & 3;
    p->Dp = (D + p->cshift + 3) & ~3;
    p->RT = (NB * p->Dp + 15) / 16;
    p->knots.assign(d->knots, d->knots + NB + 4);
    double tm[3] = {1.0, 1.0, 1.0};
    if (d->translation_maxima)
        for (int i = 0; i < 3; i++) tm[i] = d->translation_maxima[i];

    // E' = E * scale_d, mean' = mean * scale_d (float64 products): the reference applies
    // translation_maxima after adding the mean (motion_primitive.py:251-255), which is
    // the same linear map.
    p->Es.resize((size_t)R * L);
    p->means_.resize(R);
    for (int r = 0; r < R; r++) {
        double sc = (r % D < 3) ? tm[r % D] : 1.0;
        for (int k = 0; k < L; k++) {
            double e = d->eigen_is_transposed ? d->eigen_vectors[(size_t)r * L + k] : d->eigen_vectors[(size_t)k * R + r];
            p->Es[(size_t)r * L + k] = e * sc;
        }
        p->means_[r] = d->mean_vector[r] * sc;
    }

    int rc = MG_OK;
    {   // device images of E'
        std::vector<float> et32((size_t)L * R);
        std::vector<double> et64((size_t)L * R);
        for (int r = 0; r < R; r++)
            for (int k = 0; k < L; k++) {
                et64[(size_t)k * R + r] = p->Es[(size_t)r * L + k];
                et32[(size_t)k * R + r] = (float)p->Es[(size_t)r * L + k];
            }
        if (rc == MG_OK) rc = mg_upload(ctx, et32, &p->d_Et32);
        if (rc == MG_OK) rc = mg_upload(ctx, et64, &p->d_Et64);
        if (rc == MG_OK) rc = mg_upload(ctx, p->means_, &p->d_mean);
        if (rc == MG_OK) rc = mg_upload(ctx, p->knots, &p->d_knots);
        {
            std::vector<float> m32((size_t)p->RT * 16, 0.0f);
            for (int i = 0; i < NB; i++)
                for (int dd = p->nroot; dd < D; dd++)   // the root rows' C-in is zero: they hold E'.s alone (the split's delta)
                    m32[(size_t)i * p->Dp + dd + p->cshift] = (float)p->means_[(size_t)i * D + dd];
            if (rc == MG_OK) rc = mg_upload(ctx, m32, &p->d_mean32);
        }
        if (rc == MG_OK && p->KK > 0) {
            // root rows (row = i*nroot + d) as v_mfma_f64_16x16x4_f64 A fragments:
            // lane l supplies A[row = l & 15][k = 4*kk + (l >> 4)].  Image [tile][kk][lane].
            const int KK = p->KK, nr = p->nroot, RR = NB * nr;
            p->RRT = (RR + 15) / 16;
            std::vector<double> rpack((size_t)p->RRT * KK * 64, 0.0), mroot((size_t)p->RRT * 16, 0.0);
            for (int rr = 0; rr < RR; rr++) mroot[rr] = p->means_[(size_t)(rr / nr) * D + rr % nr];
            for (int t = 0; t < p->RRT; t++)
                for (int kk = 0; kk < KK; kk++)
                    for (int lane = 0; lane < 64; lane++) {
                        int rr = t * 16 + (lane & 15), k = 4 * kk + (lane >> 4);
                        if (rr < RR && k < L)
                            rpack[((size_t)t * KK + kk) * 64 + lane] = p->Es[((size_t)(rr / nr) * D + rr % nr) * L + k];
                    }
            rc = mg_upload(ctx, rpack, &p->d_Erpack);
            if (rc == MG_OK) rc = mg_upload(ctx, mroot, &p->d_meanroot);
        }
        if (rc == MG_OK && p->KK > 0) {
            // MFMA A-operand fragments of v_mfma_f32_16x16x4_f32: lane l supplies
            // A[row = l & 15][k = 4*kk + (l >> 4)].  Rows are the padded coefficient rows
            // r' = i*Dp + d + cshift (zero rows elsewhere).  Image [rt][kk/2][lane][2].
            const int KK = p->KK, Dp = p->Dp;
            std::vector<float> pack((size_t)p->RT * KK * 64, 0.0f);
            for (int rt = 0; rt < p->RT; rt++)
                for (int kk = 0; kk < KK; kk++)
                    for (int lane = 0; lane < 64; lane++) {
                        int rp = rt * 16 + (lane & 15), k = 4 * kk + (lane >> 4);
                        int i = rp / Dp, dd = rp - i * Dp - p->cshift;
                        float v = (i < NB && dd >= 0 && dd < D && k < L) ? (float)p->Es[((size_t)i * D + dd) * L + k] : 0.0f;
                        pack[(((size_t)rt * (KK / 2) + kk / 2) * 64 + lane) * 2 + (kk & 1)] = v;
                    }
            {   // ... followed by a second copy for the chunk-stationary kernel's one-time fragment loads: per tile the k-steps in QUADS per lane ([q][64][4],
                // q < KK / 4), then the remaining pair ([64][2]) where KK is no multiple of four -- 16-byte loads, 3 instead of 5 instructions per tile at
                // KK = 10 through a CU's address unit while the whole chip fetches its windows at once (csrc/mg_frames_cs.hip, mg_cs_load_fragments)
                const size_t n1 = pack.size();
                pack.resize(2 * n1, 0.0f);
                for (int rt = 0; rt < p->RT; rt++)
                    for (int kk = 0; kk < KK; kk++)
                        for (int lane = 0; lane < 64; lane++) {
                            const float v = pack[(((size_t)rt * (KK / 2) + kk / 2) * 64 + lane) * 2 + (kk & 1)];
                            const int q = kk / 4, nq = KK / 4;
                            const size_t tile = n1 + (size_t)rt * KK * 64;
                            if (q < nq) pack[tile + ((size_t)q * 64 + lane) * 4 + (kk & 3)] = v;
                            else pack[tile + (size_t)nq * 256 + (size_t)lane * 2 + (kk & 1)] = v;
                        }
            }
            rc = mg_upload(ctx, pack, &p->d_Epack);
        }
    }
    if (rc != MG_OK) { mg_primitive_free(p); return rc; }

    if (K > 0) {
        p->gw.assign(d->gmm_weights, d->gmm_weights + K);
        p->gm.assign(d->gmm_means, d->gmm_means + (size_t)K * Lg);
        p->gc.assign(d->gmm_covars, d->gmm_covars + (size_t)K * Lg * Lg);
        p->gp.assign((size_t)K * Lg * Lg, 0.0);
        std::vector<double> chol((size_t)K * Lg * Lg), inv((size_t)Lg * Lg);
        std::vector<double> gP((size_t)K * Lg * Lg, 0.0), gmP((size_t)K * Lg), gconst(K);
        const double log2pi = std::log(2.0 * M_PI);
        for (int k = 0; k < K; k++) {
            double *c = &chol[(size_t)k * Lg * Lg];
            if (!(p->gw[k] >= 0.0) || !std::isfinite(p->gw[k])) {
                mg_set_error("mg_primitive_create: gmm_weights[%d] is negative or not finite", k);
                mg_primitive_free(p);
                return MG_ERR_INVALID_ARGUMENT;
            }
            if (!mg_cholesky_lower(&p->gc[(size_t)k * Lg * Lg], Lg, c)) {
                mg_set_error("mg_primitive_create: gmm_covars[%d] is not positive definite", k);
                mg_primitive_free(p);
                return MG_ERR_NOT_POSITIVE_DEFINITE;
            }
            // sklearn _compute_precision_cholesky: P = solve_triangular(chol, I, lower=True).T
            std::fill(inv.begin(), inv.end(), 0.0);
            for (int col = 0; col < Lg; col++)
                for (int r = col; r < Lg; r++) {
                    double s = (r == col) ? 1.0 : 0.0;
                    for (int q = col; q < r; q++) s -= c[r * Lg + q] * inv[(size_t)q * Lg + col];
                    inv[(size_t)r * Lg + col] = s / c[r * Lg + r];
                }
            double *P = &p->gp[(size_t)k * Lg * Lg];
            double logdet = 0.0;
            for (int i = 0; i < Lg; i++) {
                for (int j = 0; j < Lg; j++) P[i * Lg + j] = inv[(size_t)j * Lg + i];
                logdet += std::log(P[i * Lg + i]);
            }
            for (int j = 0; j < Lg; j++) {
                double acc = 0.0;
                for (int i = 0; i <= j; i++) {
                    gP[((size_t)k * Lg + j) * Lg + i] = P[i * Lg + j];
                    acc += p->gm[(size_t)k * Lg + i] * P[i * Lg + j];
                }
                gmP[(size_t)k * Lg + j] = acc;
            }
            gconst[k] = std::log(p->gw[k]) + logdet - 0.5 * (double)Lg * log2pi;
        }
        if (p->KKg > 0) {
            // B fragments of v_mfma_f64_16x16x4_f64: lane l supplies B[k = 4*kk + (l >> 4)][col = l & 15]
            const int KK = p->KKg, JT = (Lg + 15) / 16, L = Lg;   // inside this block L is the mixture's dimension
            std::vector<double> ppack((size_t)K * JT * KK * 64, 0.0), mpad((size_t)K * JT * 16, 0.0);
            for (int k = 0; k < K; k++) {
                for (int j = 0; j < L; j++) mpad[(size_t)k * JT * 16 + j] = gmP[(size_t)k * L + j];
                for (int jt = 0; jt < JT; jt++)
                    for (int kk = 0; kk < KK; kk++)
                        for (int lane = 0; lane < 64; lane++) {
                            int i = 4 * kk + (lane >> 4), j = 16 * jt + (lane & 15);
                            if (i < L && j < L && i <= j)
                                ppack[((((size_t)k * JT + jt) * KK) + kk) * 64 + lane] = p->gp[(size_t)k * Lg * Lg + (size_t)i * L + j];
                        }
            }
            // the transposed factor for z = y P_k^T (log_likelihood_jac): lane l supplies B[k = 4*kk + (l >> 4) (= j)][col = i]
            std::vector<double> ptpack((size_t)K * JT * KK * 64, 0.0);
            for (int k = 0; k < K; k++)
                for (int it = 0; it < JT; it++)
                    for (int kk = 0; kk < KK; kk++)
                        for (int lane = 0; lane < 64; lane++) {
                            int j = 4 * kk + (lane >> 4), i = 16 * it + (lane & 15);
                            if (i < L && j < L && i <= j)
                                ptpack[((((size_t)k * JT + it) * KK) + kk) * 64 + lane] = p->gp[(size_t)k * Lg * Lg + (size_t)i * L + j];
                        }
            // sampler: x = mu + z L^T, lane l supplies B[k = 4*kk + (l >> 4) (= j)][col = i] = L_k[i][j] (lower triangular)
            std::vector<double> cpack((size_t)K * JT * KK * 64, 0.0), meanpad((size_t)K * JT * 16, 0.0);
            for (int k = 0; k < K; k++) {
                for (int i = 0; i < L; i++) meanpad[(size_t)k * JT * 16 + i] = p->gm[(size_t)k * L + i];
                for (int it = 0; it < JT; it++)
                    for (int kk = 0; kk < KK; kk++)
                        for (int lane = 0; lane < 64; lane++) {
                            int j = 4 * kk + (lane >> 4), i = 16 * it + (lane & 15);
                            if (i < L && j < L && j <= i)
                                cpack[((((size_t)k * JT + it) * KK) + kk) * 64 + lane] = chol[(size_t)k * Lg * Lg + (size_t)i * L + j];
                        }
            }
            {   // ... followed by the same fragments with the k-steps in pairs per lane ([K][JT][KK/2][64][2]): 16-byte loads in the planner step's sampler
                const size_t n1 = cpack.size();
                cpack.resize(2 * n1, 0.0);
                for (int k = 0; k < K; k++)
                    for (int it = 0; it < JT; it++)
                        for (int q = 0; q < KK / 2; q++)
                            for (int lane = 0; lane < 64; lane++)
                                for (int h = 0; h < 2; h++)
                                    cpack[n1 + ((((size_t)k * JT + it) * (KK / 2) + q) * 64 + lane) * 2 + h] = cpack[((((size_t)k * JT + it) * KK) + 2 * q + h) * 64 + lane];
            }
            if (rc == MG_OK) rc = mg_upload(ctx, cpack, &p->d_gcholpack);
            if (rc == MG_OK) rc = mg_upload(ctx, meanpad, &p->d_gmeanpad);
            {   // ... followed by the same fragments with the k-steps in PAIRS per lane, [K][JT][KK/2][64][2]: one 16-byte load per lane fetches two blocks
                // (the fused step's staged tail and its start-up half: half the load instructions through a CU's address unit, csrc/mg_gmm_device.h)
                const size_t n1 = ppack.size();
                ppack.resize(2 * n1, 0.0);
                for (int k = 0; k < K; k++)
                    for (int jt = 0; jt < JT; jt++)
                        for (int q = 0; q < KK / 2; q++)
                            for (int lane = 0; lane < 64; lane++)
                                for (int h = 0; h < 2; h++)
                                    ppack[n1 + ((((size_t)k * JT + jt) * (KK / 2) + q) * 64 + lane) * 2 + h] = ppack[((((size_t)k * JT + jt) * KK) + 2 * q + h) * 64 + lane];
            }
            if (rc == MG_OK) rc = mg_upload(ctx, ppack, &p->d_gPpack);
            if (rc == MG_OK) rc = mg_upload(ctx, ptpack, &p->d_gPTpack);
            if (rc == MG_OK) rc = mg_upload(ctx, mpad, &p->d_gmPpad);
        }
        if (rc == MG_OK) rc = mg_upload(ctx, gP, &p->d_gP);
        if (rc == MG_OK) rc = mg_upload(ctx, gmP, &p->d_gmP);
        if (rc == MG_OK) rc = mg_upload(ctx, gconst, &p->d_gconst);
        if (rc == MG_OK) rc = mg_upload(ctx, p->gm, &p->d_gmean);
        if (rc == MG_OK) rc = mg_upload(ctx, chol, &p->d_gchol);
        if (rc != MG_OK) { mg_primitive_free(p); return rc; }
    }

    {   // the mean/delta split's accuracy gate (mg_primitive_root_mode)
        std::vector<double> m2(L, 1.0);
        double wsum = 0.0;
        for (int j = 0; j < K; j++) wsum += p->gw[j];
        if (K > 0)
            for (int k = 0; k < L; k++) {
                double acc = 0.0;
                for (int j = 0; j < K; j++) {
                    const double mu = p->gm[(size_t)j * Lg + k];
                    acc += p->gw[j] * (mu * mu + p->gc[(size_t)j * Lg * Lg + (size_t)k * Lg + k]);
                }
                m2[k] = acc / wsum;
            }
        double worst = 0.0;
        for (int i = 0; i < NB; i++)
            for (int dd = 0; dd < p->nroot; dd++) {
                const double *e = &p->Es[((size_t)i * D + dd) * L];
                double ss = 0.0;
                for (int k = 0; k < L; k++) ss += e[k] * e[k] * m2[k];
                worst = std::max(worst, std::sqrt(ss));
            }
        p->root_split_est = (double)(L + 8) * 0x1p-24 * worst;
        p->root_split = std::isfinite(p->root_split_est) && p->root_split_est <= MG_ROOT_SPLIT_MAX_EST;
    }

    if (Lt > 0) {
        // mean time spline and harmonics at the canonical frames 0 .. F-1 (reference motion_primitive.py:258-268,293-296:
        // si.splev(canonical_time_range, (knots_t, coefficients, 3)); column l of eigen_vectors_time = coefficients of harmonic l)
        const int F = p->F;
        for (int i = 0; i + 1 < NBt + 4; i++)
            if (!(d->knots_time[i] <= d->knots_time[i + 1]) || !std::isfinite(d->knots_time[i + 1])) {
                mg_set_error("mg_primitive_create: knots_time must be finite and non-decreasing");
                mg_primitive_free(p);
                return MG_ERR_INVALID_ARGUMENT;
            }
        std::vector<double> tphi((size_t)F * Lt), tmean(F);
        for (int f = 0; f < F; f++) {
            int32_t i0;
            double w[4];
            mg_basis_row(d->knots_time, NBt + 4, (double)f, &i0, w);
            double m = w[0] * d->mean_time_vector[i0];
            for (int j = 1; j < 4; j++) m = std::fma(w[j], d->mean_time_vector[i0 + j], m);
            tmean[f] = m;
            for (int l = 0; l < Lt; l++) {
                double v = w[0] * d->eigen_vectors_time[(size_t)i0 * Lt + l];
                for (int j = 1; j < 4; j++) v = std::fma(w[j], d->eigen_vectors_time[(size_t)(i0 + j) * Lt + l], v);
                tphi[(size_t)f * Lt + l] = v;
            }
        }
        rc = mg_upload(ctx, tphi, &p->d_tphi);
        if (rc == MG_OK) rc = mg_upload(ctx, tmean, &p->d_tmean);
        if (rc != MG_OK) { mg_primitive_free(p); return rc; }
    }
    {   // canonical grid: np.linspace(0, F, int(F * 1.0))  (reference motion_primitive.py:233)
        const int F = p->F;
        std::vector<double> t(F);
        if (F == 1) {
            t[0] = 0.0;
        } else {
            double step = (double)F / (double)(F - 1);
            for (int f = 0; f < F; f++) t[f] = (double)f * step;
            t[F - 1] = (double)F;
        }
        p->canonical = new (std::nothrow) mg_time_grid();
        if (!p->canonical) { mg_primitive_free(p); return MG_ERR_OUT_OF_MEMORY; }
        p->canonical->owned_by_prim = true;
        rc = mg_grid_build(p, p->canonical, t.data(), F, nullptr, nullptr);
        if (rc != MG_OK) { mg_primitive_free(p); return rc; }
    }
    {   // identity grid: row i selects coefficient i, so "frames" are the coefficients
        std::vector<double> t(NB, 0.0), w((size_t)NB * 4, 0.0);
        std::vector<int32_t> i0(NB);
        for (int i = 0; i < NB; i++) {
            int base = std::min(i, NB - 4);
            i0[i] = base;
            w[4 * i + (i - base)] = 1.0;
            t[i] = (double)i;
        }
        p->coeff_grid = new (std::nothrow) mg_time_grid();
        if (!p->coeff_grid) { mg_primitive_free(p); return MG_ERR_OUT_OF_MEMORY; }
        p->coeff_grid->owned_by_prim = true;
        rc = mg_grid_build(p, p->coeff_grid, t.data(), NB, i0.data(), w.data());
        if (rc != MG_OK) { mg_primitive_free(p); return rc; }
    }
    *out = p;
    return MG_OK;
}

extern "C" int mg_primitive_info(const mg_primitive *p, int32_t *o) {
    MG_REQUIRE(p && o, "mg_primitive_info: NULL argument");
    o[0] = p->NB; o[1] = p->D; o[2] = p->L; o[3] = p->F; o[4] = p->K; o[5] = p->KK;
    o[6] = (p->canonical && p->canonical->mfma_ok) ? 1 : 0;
    o[7] = p->canonical ? p->canonical->n_chunks : 0;
    return MG_OK;
}

extern "C" int mg_primitive_root_mode(const mg_primitive *p, int32_t *split, double *estimate) {
    MG_REQUIRE(p != nullptr, "mg_primitive_root_mode: primitive is NULL");
    if (split) *split = mg_frames_root_split(p) ? 1 : 0;
    if (estimate) *estimate = p->root_split_est;
    return MG_OK;
}

extern "C" int mg_primitive_info2(const mg_primitive *p, int32_t *o) {
    MG_REQUIRE(p && o, "mg_primitive_info2: NULL argument");
    o[0] = p->Lg; o[1] = p->Lt; o[2] = p->NBt; o[3] = p->KKg;
    return MG_OK;
}

extern "C" int mg_primitive_get_precisions_cholesky(const mg_primitive *p, double *out) {
    MG_REQUIRE(p && out, "mg_primitive_get_precisions_cholesky: NULL argument");
    MG_REQUIRE(p->K > 0, "mg_primitive_get_precisions_cholesky: primitive has no mixture");
    std::copy(p->gp.begin(), p->gp.end(), out);
    return MG_OK;
}

// ---------------------------------------------------------------------------------------
// hot-path entry points (validation, then launch)
// ---------------------------------------------------------------------------------------
// A process may hold contexts on several devices: kernels and copies of a context must be issued with its device
// current (a no-op compare in the usual one-process-per-GPU deployment).
static int mg_use_device(const mg_context *ctx) {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != ctx->device) MG_HIP_CHECK(hipSetDevice(ctx->device));
    return MG_OK;
}

static int mg_check_latents(const char *fn, const mg_primitive *p, const void *lat, int dt, int64_t B, int64_t ld);
// the mixture's entry points read all n_gmm_dims columns (spatial + time latents)
static int mg_check_mixture_rows(const char *fn, const mg_primitive *p, const void *x, int dt, int64_t B, int64_t ld) {
    int rc = mg_check_latents(fn, p, x, dt, B, ld);
    if (rc != MG_OK) return rc;
    MG_REQUIRE(ld >= p->Lg, "%s: leading dimension %lld < n_gmm_dims %d", fn, (long long)ld, p->Lg);
    return MG_OK;
}
static int mg_check_latents(const char *fn, const mg_primitive *p, const void *lat, int dt, int64_t B, int64_t ld) {
    MG_REQUIRE(p != nullptr, "%s: primitive is NULL", fn);
    { int rc = mg_use_device(p->ctx); if (rc != MG_OK) return rc; }
    MG_REQUIRE(B >= 0, "%s: n_samples = %lld < 0", fn, (long long)B);
    MG_REQUIRE(dt == MG_F32 || dt == MG_F64, "%s: latent dtype %d is neither MG_F32 nor MG_F64", fn, dt);
    MG_REQUIRE(B == 0 || lat != nullptr, "%s: latents pointer is NULL", fn);
    MG_REQUIRE(ld >= p->L, "%s: leading dimension %lld < n_components %d", fn, (long long)ld, p->L);
    return MG_OK;
}

extern "C" int mg_back_project_frames(mg_primitive *p, const mg_time_grid *g, const void *lat, int dt,
                                      int64_t B, int64_t ld, float *out, int path) {
    int rc = mg_check_latents("mg_back_project_frames", p, lat, dt, B, ld);
    if (rc != MG_OK) return rc;
    if (!g) g = p->canonical;
    MG_REQUIRE(g->prim == p, "mg_back_project_frames: grid belongs to another primitive");
    MG_REQUIRE(path == MG_PATH_AUTO || path == MG_PATH_MFMA || path == MG_PATH_DIRECT, "mg_back_project_frames: unknown path %d", path);
    if (B == 0 || g->T == 0) return MG_OK;
    MG_REQUIRE(out != nullptr, "mg_back_project_frames: frames pointer is NULL");
    MG_REQUIRE(B * (int64_t)g->T * p->D < ((int64_t)1 << 40), "mg_back_project_frames: output too large");
    bool use_mfma;
    if (path == MG_PATH_MFMA) {
        if (!g->mfma_ok) {
            mg_set_error("mg_back_project_frames: MFMA path unsupported for this shape (n_components %d > 64 or n_dim %d too large for LDS)", p->L, p->D);
            return MG_ERR_UNSUPPORTED;
        }
        use_mfma = true;
    } else if (path == MG_PATH_DIRECT) {
        use_mfma = false;
    } else {
        use_mfma = g->mfma_ok && B >= 8;
    }
    if (use_mfma) return mg_launch_frames_mfma(p, g, lat, dt, B, ld, out, nullptr, 0);   // timed by events attached to the launch
    mg_prof_begin(p->ctx, 0);
    rc = mg_launch_frames_direct(p, g, lat, dt, B, ld, out, false);
    mg_prof_end(p->ctx, 0);
    return rc;
}

extern "C" int mg_back_project_frames_f64(mg_primitive *p, const mg_time_grid *g, const void *lat, int dt,
                                          int64_t B, int64_t ld, double *out) {
    int rc = mg_check_latents("mg_back_project_frames_f64", p, lat, dt, B, ld);
    if (rc != MG_OK) return rc;
    if (!g) g = p->canonical;
    MG_REQUIRE(g->prim == p, "mg_back_project_frames_f64: grid belongs to another primitive");
    if (B == 0 || g->T == 0) return MG_OK;
    MG_REQUIRE(out != nullptr, "mg_back_project_frames_f64: frames pointer is NULL");
    mg_prof_begin(p->ctx, 0);
    rc = mg_launch_frames_direct(p, g, lat, dt, B, ld, out, true);
    mg_prof_end(p->ctx, 0);
    return rc;
}

extern "C" int mg_back_project_coeffs(mg_primitive *p, const void *lat, int dt, int64_t B, int64_t ld,
                                      void *out, int odt) {
    int rc = mg_check_latents("mg_back_project_coeffs", p, lat, dt, B, ld);
    if (rc != MG_OK) return rc;
    MG_REQUIRE(odt == MG_F32 || odt == MG_F64, "mg_back_project_coeffs: bad output dtype %d", odt);
    if (B == 0) return MG_OK;
    MG_REQUIRE(out != nullptr, "mg_back_project_coeffs: coeffs pointer is NULL");
    const mg_time_grid *g = p->coeff_grid;
    if (odt == MG_F64) return mg_launch_frames_direct(p, g, lat, dt, B, ld, out, true);
    if (g->mfma_ok && B >= 8) return mg_launch_frames_mfma(p, g, lat, dt, B, ld, (float *)out, nullptr);
    return mg_launch_frames_direct(p, g, lat, dt, B, ld, out, false);
}

extern "C" int mg_spline_evaluate(mg_primitive *p, const mg_time_grid *g, const double *coeffs, int64_t n, double *out) {
    MG_REQUIRE(p != nullptr && n >= 0, "mg_spline_evaluate: bad arguments");
    if (!g) g = p->canonical;
    MG_REQUIRE(g->prim == p, "mg_spline_evaluate: grid belongs to another primitive");
    if (n == 0 || g->T == 0) return MG_OK;
    MG_REQUIRE(coeffs && out, "mg_spline_evaluate: NULL pointer");
    { int rc = mg_use_device(p->ctx); if (rc != MG_OK) return rc; }
    mg_prof_begin(p->ctx, 5);
    int rc = mg_launch_spline_eval(p, g, coeffs, n, out);
    mg_prof_end(p->ctx, 5);
    return rc;
}

extern "C" int mg_gmm_log_prob(mg_primitive *p, const void *x, int xdt, int64_t B, int64_t ld, void *out, int odt) {
    int rc = mg_check_mixture_rows("mg_gmm_log_prob", p, x, xdt, B, ld);
    if (rc != MG_OK) return rc;
    MG_REQUIRE(p->K > 0, "mg_gmm_log_prob: primitive has no mixture");
    MG_REQUIRE(odt == MG_F32 || odt == MG_F64, "mg_gmm_log_prob: bad output dtype %d", odt);
    if (B == 0) return MG_OK;
    MG_REQUIRE(out != nullptr, "mg_gmm_log_prob: output pointer is NULL");
    mg_prof_begin(p->ctx, 1);
    rc = mg_launch_gmm_logp(p, x, xdt, B, ld, out, odt);
    mg_prof_end(p->ctx, 1);
    return rc;
}

extern "C" int mg_time_function_canonical(mg_primitive *p, const void *gamma, int gdt, int64_t B, int64_t ld, double *out) {
    MG_REQUIRE(p != nullptr, "mg_time_function_canonical: primitive is NULL");
    MG_REQUIRE(p->Lt > 0, "mg_time_function_canonical: the primitive has no time model");
    MG_REQUIRE(B >= 0 && (gdt == MG_F32 || gdt == MG_F64) && ld >= p->Lt, "mg_time_function_canonical: bad arguments (ld %lld, n_time_components %d)",
               (long long)ld, p->Lt);
    if (B == 0) return MG_OK;
    MG_REQUIRE(gamma && out, "mg_time_function_canonical: NULL pointer");
    { int rc0 = mg_use_device(p->ctx); if (rc0 != MG_OK) return rc0; }
    return mg_launch_time_function(p, gamma, gdt, B, ld, out);
}

// back_project_time_function for a batch (reference motion_primitive.py:268-319): the spline's time function t'(t) of every
// candidate -- 0, the inverse of its canonical time function at linspace(1, t(F-2), num), F - 1 -- padded to t_cap per row.
extern "C" int mg_time_function_sample(mg_primitive *p, const void *gamma, int gdt, int64_t B, int64_t ld, double speed, double *times, int32_t *lengths,
                                       int32_t t_cap, double *canonical_out) {
    MG_REQUIRE(p != nullptr, "mg_time_function_sample: primitive is NULL");
    MG_REQUIRE(p->Lt > 0, "mg_time_function_sample: the primitive has no time model");
    MG_REQUIRE(B >= 0 && B < ((int64_t)1 << 31) && (gdt == MG_F32 || gdt == MG_F64) && ld >= p->Lt, "mg_time_function_sample: bad arguments (ld %lld, n_time_components %d)",
               (long long)ld, p->Lt);
    MG_REQUIRE(speed > 0.0 && std::isfinite(speed) && t_cap >= 2, "mg_time_function_sample: speed must be positive and finite, t_cap >= 2");
    if (B == 0) return MG_OK;
    MG_REQUIRE(gamma && times && lengths, "mg_time_function_sample: NULL pointer");
    { int rc0 = mg_use_device(p->ctx); if (rc0 != MG_OK) return rc0; }
    return mg_launch_timewarp(p, gamma, gdt, B, ld, speed, times, lengths, t_cap, canonical_out);
}

// MotionSpline.get_motion_vector of every candidate at ITS OWN time function (reference motion_spline.py:71-86 behind
// back_project(s, True), motion_primitive.py:206-234): times (B, t_cap) float64, lengths (B) (NULL: every row has t_cap samples;
// rows with length <= 0 are skipped), out (B, t_cap, D) float64 or float32 (the float64 value rounded once); samples beyond a
// row's length are left as they were.
extern "C" int mg_back_project_frames_at(mg_primitive *p, const void *lat, int dt, int64_t B, int64_t ld, const double *times, const int32_t *lengths,
                                         int32_t t_cap, void *out, int odt) {
    int rc = mg_check_latents("mg_back_project_frames_at", p, lat, dt, B, ld);
    if (rc != MG_OK) return rc;
    MG_REQUIRE(odt == MG_F32 || odt == MG_F64, "mg_back_project_frames_at: bad output dtype %d", odt);
    MG_REQUIRE(t_cap >= 1 && B < ((int64_t)1 << 31), "mg_back_project_frames_at: t_cap must be >= 1");
    if (B == 0) return MG_OK;
    MG_REQUIRE(times && out, "mg_back_project_frames_at: NULL pointer");
    return mg_launch_frames_at(p, lat, dt, B, ld, times, lengths, t_cap, out, odt);
}

// rows [row_begin, row_begin + row_count) of the draw of n rows (mg_gmm_sample: all of them)
extern "C" int mg_gmm_sample_rows(mg_primitive *p, int64_t n, const int64_t *counts, uint64_t seed, int64_t row_begin, int64_t row_count,
                                  void *x, int xdt, int64_t ld, int32_t *comp) {
    MG_REQUIRE(p != nullptr && n >= 0, "mg_gmm_sample: bad arguments");
    MG_REQUIRE(p->K > 0, "mg_gmm_sample: primitive has no mixture");
    MG_REQUIRE(xdt == MG_F32 || xdt == MG_F64, "mg_gmm_sample: bad dtype %d", xdt);
    MG_REQUIRE(counts != nullptr, "mg_gmm_sample: counts is NULL");
    MG_REQUIRE(ld >= p->Lg, "mg_gmm_sample: leading dimension %lld < n_gmm_dims %d", (long long)ld, p->Lg);
    MG_REQUIRE(row_begin >= 0 && row_count >= 0 && row_begin + row_count <= n, "mg_gmm_sample_rows: rows [%lld, %lld) outside the draw of %lld",
               (long long)row_begin, (long long)(row_begin + row_count), (long long)n);
    { int rc0 = mg_use_device(p->ctx); if (rc0 != MG_OK) return rc0; }
    // [0 .. K]: row prefix sums; [K+1 .. 2K+1]: prefix sums of 16-row tiles (a tile never straddles two components)
    std::vector<int64_t> cum(2 * (size_t)p->K + 2, 0);
    for (int k = 0; k < p->K; k++) {
        MG_REQUIRE(counts[k] >= 0, "mg_gmm_sample: counts[%d] < 0", k);
        cum[k + 1] = cum[k] + counts[k];
        cum[p->K + 1 + k + 1] = cum[p->K + 1 + k] + (counts[k] + 15) / 16;
    }
    MG_REQUIRE(cum[p->K] == n, "mg_gmm_sample: counts sum to %lld, expected %lld", (long long)cum[p->K], (long long)n);
    if (row_count == 0) return MG_OK;
    MG_REQUIRE(x != nullptr, "mg_gmm_sample: output pointer is NULL");
    // the tiles that hold the first and the last wanted row
    auto tile_of = [&](int64_t row) {
        int c = 0;
        while (c + 1 < p->K && row >= cum[c + 1]) c++;
        return cum[p->K + 1 + c] + (row - cum[c]) / 16;
    };
    const int64_t tile0 = tile_of(row_begin), tile_end = tile_of(row_begin + row_count - 1) + 1;
    int rc;
    if (mg_gmm_sample_takes_host_prefix(p)) {   // prefix sums travel as a kernel argument: nothing to upload or wait for
        mg_prof_begin(p->ctx, 4);
        rc = mg_launch_gmm_sample(p, n, nullptr, cum.data(), tile_end, seed, x, xdt, ld, comp, tile0, row_begin, row_begin + row_count);
        mg_prof_end(p->ctx, 4);
        return rc;
    }
    void *scr = nullptr;
    rc = mg_ctx_scratch(p->ctx, (int64_t)cum.size() * 8, &scr);
    if (rc != MG_OK) return rc;
    MG_HIP_CHECK(hipMemcpyAsync(scr, cum.data(), cum.size() * 8, hipMemcpyHostToDevice, p->ctx->stream));
    MG_HIP_CHECK(hipStreamSynchronize(p->ctx->stream));  // cum is a stack-lifetime host buffer
    mg_prof_begin(p->ctx, 4);
    rc = mg_launch_gmm_sample(p, n, (const int64_t *)scr, nullptr, tile_end, seed, x, xdt, ld, comp, tile0, row_begin, row_begin + row_count);
    mg_prof_end(p->ctx, 4);
    return rc;
}
extern "C" int mg_gmm_sample(mg_primitive *p, int64_t n, const int64_t *counts, uint64_t seed, void *x, int xdt,
                             int64_t ld, int32_t *comp) {
    return mg_gmm_sample_rows(p, n, counts, seed, 0, n < 0 ? 0 : n, x, xdt, ld, comp);
}

// ---- constraint sets ---------------------------------------------------------------------
// A constraint's row of the scorers' parameter table: {type, weight, target[3], ref_dir[3]}.  For the direction constraint the row
// carries what its residual derives from the target alone -- the unit target (tx, tz) and sqrt(tx^2 + tz^2) -- computed here once per
// set instead of by every lane for every candidate: the residual's own statements (mg_score_device.h), correctly rounded IEEE
// square roots and divisions on both sides, no contraction: the same bits.
static void mg_constraint_par_row(const mg_keyframe_constraint &c, double *q) {
    q[0] = (double)c.type; q[1] = c.weight_factor;
    for (int i = 0; i < 3; i++) { q[2 + i] = c.target[i]; q[5 + i] = c.ref_dir[i]; }
    if (c.type == MG_CONSTRAINT_DIRECTION_2D) {
        const double tn = std::sqrt(c.target[0] * c.target[0] + c.target[1] * c.target[1]);
        const double tx = c.target[0] / tn, tz = c.target[1] / tn;
        q[2] = tx; q[3] = tz; q[4] = std::sqrt(tx * tx + tz * tz);
    }
}

extern "C" int mg_constraint_set_create_full(mg_primitive *p, const mg_skeleton_desc *sk, const mg_keyframe_constraint *cons,
                                             int32_t n, const mg_pose_constraint *poses, int32_t n_poses,
                                             const mg_alignment_desc *al, mg_constraint_set **out) {
    MG_REQUIRE(p && out && n >= 0 && (n == 0 || cons) && n_poses >= 0 && (n_poses == 0 || poses), "mg_constraint_set_create: bad arguments");
    *out = nullptr;
    // animated joints of the skeleton in skeleton order: a pose block is root xyz + one quaternion per slot
    std::vector<int> slot;
    int n_slots = 0;
    if (sk) {
        slot.assign((size_t)std::max(sk->n_joints, 0), -1);
        for (int j = 0; j < sk->n_joints; j++)
            if (sk->quat_channel && sk->quat_channel[j] >= 0) slot[(size_t)j] = n_slots++;
    }
    std::vector<double> pose_tab;
    std::vector<size_t> pose_off((size_t)std::max(n, 1), 0);
    const int nch = std::min(7, p->D), L = p->L, D = p->D;
    if (sk) {
        MG_REQUIRE(sk->n_joints > 0 && sk->parents && sk->offsets && sk->quat_channel, "mg_constraint_set_create_fk: incomplete skeleton");
        MG_REQUIRE(sk->parents[0] < 0, "mg_constraint_set_create_fk: joint 0 must be the root");
        for (int j = 0; j < sk->n_joints; j++) {
            MG_REQUIRE(sk->parents[j] < j, "mg_constraint_set_create_fk: joint %d: parents must precede their children", j);
            MG_REQUIRE(sk->quat_channel[j] < 0 || sk->quat_channel[j] + 4 <= D,
                       "mg_constraint_set_create_fk: joint %d: quaternion channel %d outside n_dim = %d", j, sk->quat_channel[j], D);
        }
    }
    // chains (root first) and row counts
    std::vector<std::vector<int>> chains(n), chains2(n);   // chains2: second joint of a midpoint constraint
    std::vector<int32_t> woff(n + 2, 0), chain_len(n, 0);
    std::vector<int> al_chain;   // root .. aligning node, every one of their quaternions turns the heading
    if (al) {
        MG_REQUIRE(D >= 7, "mg_constraint_set_create_aligned: alignment needs the root quaternion, n_dim = %d", D);
        MG_REQUIRE(al->joint == MG_ALIGN_START_POSE || al->joint == 0 || (sk && al->joint > 0 && al->joint < sk->n_joints),
                   "mg_constraint_set_create_aligned: aligning joint %d needs a skeleton that has it", al->joint);
        const double hn = std::sqrt(al->heading[0] * al->heading[0] + al->heading[1] * al->heading[1]);
        MG_REQUIRE(std::isfinite(hn) && hn > 0.0 && std::isfinite(al->position[0]) && std::isfinite(al->position[2]),
                   "mg_constraint_set_create_aligned: previous heading / position not finite or zero");
        if (al->joint == MG_ALIGN_START_POSE) { /* no chain: the rotation is given, not derived from a heading */ }
        else if (al->joint == 0) al_chain.push_back(0);
        else for (int j = al->joint; j >= 0; j = sk->parents[j]) al_chain.insert(al_chain.begin(), j);
        MG_REQUIRE((int)al_chain.size() <= MG_MAX_CHAIN, "mg_constraint_set_create_aligned: chain of %d joints exceeds %d", (int)al_chain.size(), MG_MAX_CHAIN);
    }
    for (int c = 0; c < n; c++) {
        const int type = cons[c].type;
        MG_REQUIRE(type >= MG_CONSTRAINT_POSITION && type <= MG_CONSTRAINT_VALUE_HEADING,
                   "mg_constraint_set_create: constraint %d has unknown type %d", c, type);
        MG_REQUIRE(std::isfinite(cons[c].canonical_keyframe), "mg_constraint_set_create: constraint %d keyframe not finite", c);
        MG_REQUIRE((type == MG_CONSTRAINT_VALUE_POSITION || type == MG_CONSTRAINT_VALUE_HEADING) ? (cons[c].target[0] == 0.0 || cons[c].target[0] == 2.0 ||
                   (type == MG_CONSTRAINT_VALUE_POSITION && cons[c].target[0] == 1.0)) : true, "mg_constraint_set_create: constraint %d: value constraints select a component with target[0] = 0, (1,) 2", c);
        MG_REQUIRE(type == MG_CONSTRAINT_POSITION ? D >= 3 : D >= 7,
                   "mg_constraint_set_create: constraint %d needs more pose channels than n_dim = %d", c, D);
        int rows = nch;
        if (type == MG_CONSTRAINT_JOINT_POSITION) {
            MG_REQUIRE(sk != nullptr, "mg_constraint_set_create: constraint %d needs a skeleton (mg_constraint_set_create_fk)", c);
            MG_REQUIRE(cons[c].joint >= 0 && cons[c].joint < sk->n_joints, "mg_constraint_set_create_fk: constraint %d: joint %d out of range", c, cons[c].joint);
            for (int j = cons[c].joint; j >= 0; j = sk->parents[j]) chains[c].insert(chains[c].begin(), j);
            int m = (int)chains[c].size() - 1;   // joints below the root on the chain
            // a point given in the joint's own frame hangs below it like one more joint: the joint's own quaternion turns it
            if (cons[c].ref_dir[0] != 0.0 || cons[c].ref_dir[1] != 0.0 || cons[c].ref_dir[2] != 0.0) m++;
            MG_REQUIRE(m <= MG_MAX_CHAIN, "mg_constraint_set_create_fk: constraint %d: chain of %d joints exceeds %d", c, m, MG_MAX_CHAIN);
            chain_len[c] = m;
            rows = 3 + 4 * std::max(m, 1);
        } else if (type == MG_CONSTRAINT_POSE) {
            MG_REQUIRE(sk != nullptr, "mg_constraint_set_create: constraint %d (pose) needs a skeleton", c);
            MG_REQUIRE(cons[c].joint >= 0 && cons[c].joint < n_poses, "mg_constraint_set_create: constraint %d refers to pose %d of %d", c, cons[c].joint, n_poses);
            const mg_pose_constraint &pc = poses[cons[c].joint];
            MG_REQUIRE(pc.n_points > 0 && pc.n_points <= MG_MAX_POSE_POINTS && pc.joints && pc.points && pc.weights,
                       "mg_constraint_set_create: pose %d: 1..%d points with joints, points and weights", cons[c].joint, MG_MAX_POSE_POINTS);
            double sw = 0.0;
            for (int i = 0; i < pc.n_points; i++) {
                MG_REQUIRE(pc.joints[i] >= 0 && pc.joints[i] < sk->n_joints, "mg_constraint_set_create: pose %d: joint %d out of range", cons[c].joint, pc.joints[i]);
                MG_REQUIRE(std::isfinite(pc.weights[i]) && std::isfinite(pc.points[3 * i]) && std::isfinite(pc.points[3 * i + 1]) && std::isfinite(pc.points[3 * i + 2]),
                           "mg_constraint_set_create: pose %d: point %d not finite", cons[c].joint, i);
                sw += pc.weights[i];
            }
            MG_REQUIRE(sw != 0.0, "mg_constraint_set_create: pose %d: weights sum to zero", cons[c].joint);
            const int block = 3 + 4 * n_slots;
            pose_off[(size_t)c] = pose_tab.size();
            pose_tab.resize(pose_tab.size() + MG_POSE_HDR + (size_t)pc.n_points * MG_POSE_REC, 0.0);
            double *tb = &pose_tab[pose_off[(size_t)c]];
            tb[0] = pc.n_points; tb[1] = pc.has_velocity ? 1.0 : 0.0; tb[2] = pc.velocity[0]; tb[3] = pc.velocity[1]; tb[4] = pc.velocity[2];
            tb[5] = block;
            for (int i = 0; i < pc.n_points; i++) {
                double *rec = tb + MG_POSE_HDR + (size_t)i * MG_POSE_REC;
                rec[0] = pc.points[3 * i]; rec[1] = pc.points[3 * i + 1]; rec[2] = pc.points[3 * i + 2]; rec[3] = pc.weights[i];
                std::vector<int> ch;
                for (int j = pc.joints[i]; j >= 0; j = sk->parents[j]) ch.insert(ch.begin(), j);
                const int m = (int)ch.size() - 1;
                MG_REQUIRE(m <= MG_MAX_CHAIN, "mg_constraint_set_create: pose %d: chain of %d joints exceeds %d", cons[c].joint, m, MG_MAX_CHAIN);
                rec[4] = m;
                for (int k = 0; k < m; k++) {   // link k: rotation of chain joint k, offset of chain joint k + 1
                    rec[5 + 4 * k] = slot[(size_t)ch[(size_t)k]] >= 0 ? 3 + 4 * slot[(size_t)ch[(size_t)k]] : -1.0;
                    for (int e = 0; e < 3; e++) rec[6 + 4 * k + e] = sk->offsets[(size_t)ch[(size_t)k + 1] * 3 + e];
                }
            }
            rows = block * (pc.has_velocity ? 2 : 1);
        } else if (type == MG_CONSTRAINT_LOOK_AT) {
            MG_REQUIRE(cons[c].joint == 0 || (sk && cons[c].joint > 0 && cons[c].joint < sk->n_joints),
                       "mg_constraint_set_create_fk: constraint %d: joint %d needs a skeleton that has it", c, cons[c].joint);
            if (cons[c].joint == 0) chains[c].push_back(0);
            else for (int j = cons[c].joint; j >= 0; j = sk->parents[j]) chains[c].insert(chains[c].begin(), j);
            const int m = (int)chains[c].size();       // quaternions root .. joint; the first m - 1 place the joint
            MG_REQUIRE(m <= MG_MAX_CHAIN, "mg_constraint_set_create_fk: constraint %d: chain of %d joints exceeds %d", c, m, MG_MAX_CHAIN);
            const double rn = std::sqrt(cons[c].ref_dir[0] * cons[c].ref_dir[0] + cons[c].ref_dir[1] * cons[c].ref_dir[1] + cons[c].ref_dir[2] * cons[c].ref_dir[2]);
            MG_REQUIRE(std::isfinite(rn) && rn > 0.0, "mg_constraint_set_create: constraint %d: zero reference vector", c);
            MG_REQUIRE(std::isfinite(cons[c].target[0]) && std::isfinite(cons[c].target[1]) && std::isfinite(cons[c].target[2]),
                       "mg_constraint_set_create: constraint %d: look-at target not finite", c);
            chain_len[c] = m;
            rows = 3 + 4 * m;
        } else if (type == MG_CONSTRAINT_JOINT_MIDPOINT) {
            MG_REQUIRE(sk != nullptr, "mg_constraint_set_create: constraint %d needs a skeleton (mg_constraint_set_create_fk)", c);
            MG_REQUIRE(cons[c].joint >= 0 && cons[c].joint < sk->n_joints && cons[c].joint2 >= 0 && cons[c].joint2 < sk->n_joints,
                       "mg_constraint_set_create_fk: constraint %d: joints %d, %d out of range", c, cons[c].joint, cons[c].joint2);
            for (int j = cons[c].joint; j >= 0; j = sk->parents[j]) chains[c].insert(chains[c].begin(), j);
            for (int j = cons[c].joint2; j >= 0; j = sk->parents[j]) chains2[c].insert(chains2[c].begin(), j);
            const int m = (int)chains[c].size() - 1, m2 = (int)chains2[c].size() - 1;
            MG_REQUIRE(m <= MG_MAX_CHAIN && m2 <= MG_MAX_CHAIN, "mg_constraint_set_create_fk: constraint %d: chain exceeds %d joints", c, MG_MAX_CHAIN);
            chain_len[c] = m | (m2 << 16);
            rows = 3 + 4 * std::max(m, 1) + 4 * std::max(m2, 1);
        } else if (type == MG_CONSTRAINT_VALUE_HEADING) {
            MG_REQUIRE(cons[c].joint == 0 || (sk && cons[c].joint > 0 && cons[c].joint < sk->n_joints),
                       "mg_constraint_set_create_fk: constraint %d: joint %d needs a skeleton that has it", c, cons[c].joint);
            if (cons[c].joint == 0) chains[c].push_back(0);
            else for (int j = cons[c].joint; j >= 0; j = sk->parents[j]) chains[c].insert(chains[c].begin(), j);
            const int m = (int)chains[c].size();
            MG_REQUIRE(m <= MG_MAX_CHAIN, "mg_constraint_set_create_fk: constraint %d: chain of %d joints exceeds %d", c, m, MG_MAX_CHAIN);
            const double rn = std::sqrt(cons[c].ref_dir[0] * cons[c].ref_dir[0] + cons[c].ref_dir[1] * cons[c].ref_dir[1] + cons[c].ref_dir[2] * cons[c].ref_dir[2]);
            MG_REQUIRE(std::isfinite(rn) && rn > 0.0, "mg_constraint_set_create: constraint %d: zero reference vector", c);
            chain_len[c] = m;
            rows = 4 * m;
        } else if (type == MG_CONSTRAINT_JOINT_ORIENTATION) {
            MG_REQUIRE(cons[c].joint == 0 || (sk && cons[c].joint > 0 && cons[c].joint < sk->n_joints),
                       "mg_constraint_set_create_fk: constraint %d: joint %d needs a skeleton that has it", c, cons[c].joint);
            if (cons[c].joint == 0) chains[c].push_back(0);
            else for (int j = cons[c].joint; j >= 0; j = sk->parents[j]) chains[c].insert(chains[c].begin(), j);
            const int m = (int)chains[c].size();       // every quaternion root .. joint turns the vector
            MG_REQUIRE(m <= MG_MAX_CHAIN, "mg_constraint_set_create_fk: constraint %d: chain of %d joints exceeds %d", c, m, MG_MAX_CHAIN);
            const double tn = std::sqrt(cons[c].target[0] * cons[c].target[0] + cons[c].target[1] * cons[c].target[1] + cons[c].target[2] * cons[c].target[2]);
            const double rn = std::sqrt(cons[c].ref_dir[0] * cons[c].ref_dir[0] + cons[c].ref_dir[1] * cons[c].ref_dir[1] + cons[c].ref_dir[2] * cons[c].ref_dir[2]);
            MG_REQUIRE(std::isfinite(tn) && tn > 0.0 && std::isfinite(rn) && rn > 0.0, "mg_constraint_set_create: constraint %d: zero target or reference vector", c);
            chain_len[c] = m;
            rows = 4 * m;
        }
        woff[c + 1] = woff[c] + rows;
    }
    woff[n + 1] = woff[n] + (al ? 3 + 4 * (int)al_chain.size() : 0);
    mg_constraint_set *cs = new (std::nothrow) mg_constraint_set();
    if (!cs) return MG_ERR_OUT_OF_MEMORY;
    cs->prim = p; cs->n = n; cs->nch = nch;
    const size_t rows_total = (size_t)woff[n + 1];
    std::vector<double> W(std::max<size_t>(rows_total, 1) * L, 0.0), bias(std::max<size_t>(rows_total, 1), 0.0), par((size_t)std::max(n, 1) * 8, 0.0);
    std::vector<double> choff((size_t)std::max(n, 1) * 2 * MG_MAX_CHAIN * 3, 0.0);
    for (int c = 0; c < n; c++) {
        int32_t i0; double w[4];
        mg_basis_row(p->knots.data(), (int)p->knots.size(), cons[c].canonical_keyframe, &i0, w);
        auto fill_row = [&](size_t row, int d) {   // pose channel d at the keyframe as a (1 x L) matrix + bias
            double b = 0.0;
            for (int j = 0; j < 4; j++) b = b + w[j] * p->means_[(size_t)(i0 + j) * D + d];
            bias[row] = b;
            for (int k = 0; k < L; k++) {
                double acc = 0.0;
                for (int j = 0; j < 4; j++) acc = acc + w[j] * p->Es[((size_t)(i0 + j) * D + d) * L + k];
                W[row * L + k] = acc;
            }
        };
        const size_t r0 = (size_t)woff[c];
        auto fill_quats = [&](size_t row, const std::vector<int> &chain, int count) {   // (w,x,y,z) of chain[0 .. count-1]
            for (int i = 0; i < count; i++) {
                const int ch = sk ? sk->quat_channel[chain[i]] : 3;
                for (int e = 0; e < 4; e++) {
                    if (ch >= 0) fill_row(row + 4 * i + e, ch + e);
                    else bias[row + 4 * i + e] = (e == 0) ? 1.0 : 0.0;   // not animated: identity, zero matrix row
                }
            }
        };
        auto fill_offsets = [&](int which, const std::vector<int> &chain, int m) {
            for (int i = 0; i < m; i++)
                for (int e = 0; e < 3; e++) choff[(((size_t)c * 2 + which) * MG_MAX_CHAIN + i) * 3 + e] = sk->offsets[(size_t)chain[i + 1] * 3 + e];
        };
        if (cons[c].type == MG_CONSTRAINT_POSE) {
            const int block = 3 + 4 * n_slots;
            const mg_pose_constraint &pc = poses[cons[c].joint];
            for (int blk = 0; blk < (pc.has_velocity ? 2 : 1); blk++) {
                if (blk == 1) mg_basis_row(p->knots.data(), (int)p->knots.size(), cons[c].canonical_keyframe + 1.0, &i0, w);   // frame2 = evaluate(t + 1)
                const size_t rb = r0 + (size_t)blk * block;
                for (int d = 0; d < 3; d++) fill_row(rb + d, d);
                for (int j = 0; j < sk->n_joints; j++)
                    if (slot[(size_t)j] >= 0)
                        for (int e = 0; e < 4; e++) fill_row(rb + 3 + 4 * slot[(size_t)j] + e, sk->quat_channel[j] + e);
            }
        } else if (cons[c].type == MG_CONSTRAINT_JOINT_POSITION && chain_len[c] == (int)chains[c].size()) {
            // relative point: every quaternion root .. joint, the skeleton's offsets and the point itself as the last link
            for (int d = 0; d < 3; d++) fill_row(r0 + d, d);
            const int m = chain_len[c];
            fill_quats(r0 + 3, chains[c], m);
            fill_offsets(0, chains[c], m - 1);
            for (int e = 0; e < 3; e++) choff[(((size_t)c * 2) * MG_MAX_CHAIN + (m - 1)) * 3 + e] = cons[c].ref_dir[e];
        } else if (cons[c].type == MG_CONSTRAINT_LOOK_AT) {
            for (int d = 0; d < 3; d++) fill_row(r0 + d, d);
            fill_quats(r0 + 3, chains[c], chain_len[c]);
            fill_offsets(0, chains[c], chain_len[c] - 1);
        } else if (cons[c].type == MG_CONSTRAINT_JOINT_POSITION || cons[c].type == MG_CONSTRAINT_JOINT_MIDPOINT) {
            for (int d = 0; d < 3; d++) fill_row(r0 + d, d);
            const int m = chain_len[c] & 0xffff;
            fill_quats(r0 + 3, chains[c], std::max(m, 1));   // joints 0 .. m-1 (the end joint's own rotation does not move it)
            fill_offsets(0, chains[c], m);
            if (cons[c].type == MG_CONSTRAINT_JOINT_MIDPOINT) {
                const int m2 = chain_len[c] >> 16;
                fill_quats(r0 + 3 + 4 * std::max(m, 1), chains2[c], std::max(m2, 1));
                fill_offsets(1, chains2[c], m2);
            }
        } else if (cons[c].type == MG_CONSTRAINT_JOINT_ORIENTATION || cons[c].type == MG_CONSTRAINT_VALUE_HEADING) {
            fill_quats(r0, chains[c], chain_len[c]);
        } else {
            for (int d = 0; d < nch; d++) fill_row(r0 + d, d);
        }
        double *q = &par[(size_t)c * 8];
        mg_constraint_par_row(cons[c], q);
        if (cons[c].type == MG_CONSTRAINT_POSE) q[2] = (double)pose_off[(size_t)c];
    }
    std::vector<double> align;
    if (al) {
        // the candidate's first control point (new_frames[0] of align_quaternion_frames_automatically): plain rows of E', mean'
        const size_t r0 = (size_t)woff[n];
        auto cp0_row = [&](size_t row, int d) {
            bias[row] = p->means_[d];
            for (int k = 0; k < L; k++) W[row * L + k] = p->Es[(size_t)d * L + k];
        };
        for (int d = 0; d < 3; d++) cp0_row(r0 + d, d);
        for (size_t i = 0; i < al_chain.size(); i++) {
            const int ch = sk ? sk->quat_channel[al_chain[i]] : 3;
            for (int e = 0; e < 4; e++) {
                if (ch >= 0) cp0_row(r0 + 3 + 4 * i + e, ch + e);
                else bias[r0 + 3 + 4 * i + e] = (e == 0) ? 1.0 : 0.0;
            }
        }
        const double hn = std::sqrt(al->heading[0] * al->heading[0] + al->heading[1] * al->heading[1]);
        align = {(double)al_chain.size(), al->heading[0] / hn, al->heading[1] / hn, al->position[0], al->position[2],
                 al->ref_dir[0], al->ref_dir[1], al->ref_dir[2]};
        if (al->joint == MG_ALIGN_START_POSE) { align[5] = al->position[1]; align[6] = align[7] = 0.0; }   // heights += position[1]
    }
    int rc = mg_upload(p->ctx, W, &cs->d_W);
    if (rc == MG_OK && al) rc = mg_upload(p->ctx, align, &cs->d_align);
    if (rc == MG_OK && !pose_tab.empty()) { rc = mg_upload(p->ctx, pose_tab, &cs->d_pose); cs->has_pose = true; }
    if (rc == MG_OK) rc = mg_upload(p->ctx, bias, &cs->d_bias);
    if (rc == MG_OK) rc = mg_upload(p->ctx, par, &cs->d_par);
    if (rc == MG_OK) rc = mg_upload(p->ctx, woff, &cs->d_woff);
    if (rc == MG_OK) { if (chain_len.empty()) chain_len.push_back(0); rc = mg_upload(p->ctx, chain_len, &cs->d_chain); }
    if (rc == MG_OK) rc = mg_upload(p->ctx, choff, &cs->d_choff);
    if (rc == MG_OK && p->KK > 0 && rows_total > 0) {
        // B fragments of v_mfma_f64_16x16x4_f64 for channels = X . W^T: lane l supplies B[k = 4*kk + (l >> 4)][col = l & 15]
        const int KK = p->KK, RT = (int)((rows_total + 15) / 16);
        std::vector<double> wpack((size_t)RT * KK * 64, 0.0), bpad((size_t)RT * 16, 0.0);
        for (size_t r = 0; r < rows_total; r++) bpad[r] = bias[r];
        for (int rt = 0; rt < RT; rt++)
            for (int kk = 0; kk < KK; kk++)
                for (int lane = 0; lane < 64; lane++) {
                    const int k = 4 * kk + (lane >> 4);
                    const size_t row = (size_t)rt * 16 + (lane & 15);
                    if (k < L && row < rows_total) wpack[((size_t)rt * KK + kk) * 64 + lane] = W[row * L + k];
                }
        cs->RT = RT;
        cs->rows = (int32_t)rows_total;
        rc = mg_upload(p->ctx, wpack, &cs->d_Wpack);
        if (rc == MG_OK) rc = mg_upload(p->ctx, bpad, &cs->d_bpad);
    }
    if (rc != MG_OK) { mg_constraint_set_destroy(cs); return rc; }
    cs->structure.assign(cons, cons + n);
    cs->align_joint = al ? al->joint : -1;
    *out = cs;
    return MG_OK;
}

extern "C" int mg_constraint_set_update(mg_constraint_set *cs, const mg_keyframe_constraint *cons, int32_t n, const mg_alignment_desc *al) {
    MG_REQUIRE(cs && cs->prim && n >= 0 && (n == 0 || cons), "mg_constraint_set_update: bad arguments");
    MG_REQUIRE(n == cs->n, "mg_constraint_set_update: %d constraints, the set was built for %d", n, cs->n);
    if (cs->has_pose) { mg_set_error("mg_constraint_set_update: sets with pose constraints are rebuilt, not updated"); return MG_ERR_UNSUPPORTED; }
    MG_REQUIRE((al != nullptr) == (cs->d_align != nullptr) && (!al || al->joint == cs->align_joint),
               "mg_constraint_set_update: the alignment (none / aligning joint) differs from the one the set was built with");
    for (int c = 0; c < n; c++) {
        const mg_keyframe_constraint &o = cs->structure[c], &v = cons[c];
        bool same = o.type == v.type && o.canonical_keyframe == v.canonical_keyframe;
        if (same && v.type >= MG_CONSTRAINT_JOINT_POSITION) same = o.joint == v.joint;
        if (same && v.type == MG_CONSTRAINT_JOINT_MIDPOINT) same = o.joint2 == v.joint2;
        if (same && v.type == MG_CONSTRAINT_JOINT_POSITION)   // a relative point is one more link of the chain, stored with the set
            same = o.ref_dir[0] == v.ref_dir[0] && o.ref_dir[1] == v.ref_dir[1] && o.ref_dir[2] == v.ref_dir[2];
        MG_REQUIRE(same, "mg_constraint_set_update: constraint %d differs in type, joint, keyframe or relative point", c);
        if (v.type == MG_CONSTRAINT_JOINT_ORIENTATION || v.type == MG_CONSTRAINT_LOOK_AT) {
            const double rn = v.ref_dir[0] * v.ref_dir[0] + v.ref_dir[1] * v.ref_dir[1] + v.ref_dir[2] * v.ref_dir[2];
            const double tn = v.target[0] * v.target[0] + v.target[1] * v.target[1] + v.target[2] * v.target[2];
            MG_REQUIRE(std::isfinite(rn) && rn > 0.0 && std::isfinite(tn) && (tn > 0.0 || v.type == MG_CONSTRAINT_LOOK_AT),
                       "mg_constraint_set_update: constraint %d: zero or non-finite target / reference vector", c);
        }
    }
    std::vector<double> values((size_t)n * 8 + 7, 0.0);   // par [n][8], then entries 1..7 of the alignment record
    for (int c = 0; c < n; c++) {
        mg_constraint_par_row(cons[c], &values[(size_t)c * 8]);
    }
    if (al) {
        const double hn = std::sqrt(al->heading[0] * al->heading[0] + al->heading[1] * al->heading[1]);
        MG_REQUIRE(std::isfinite(hn) && hn > 0.0 && std::isfinite(al->position[0]) && std::isfinite(al->position[2]),
                   "mg_constraint_set_update: previous heading / position not finite or zero");
        double *q = &values[(size_t)n * 8];
        q[0] = al->heading[0] / hn; q[1] = al->heading[1] / hn; q[2] = al->position[0]; q[3] = al->position[2];
        q[4] = al->ref_dir[0]; q[5] = al->ref_dir[1]; q[6] = al->ref_dir[2];
        if (al->joint == MG_ALIGN_START_POSE) { q[4] = al->position[1]; q[5] = q[6] = 0.0; }
    }
    { int rc = mg_use_device(cs->prim->ctx); if (rc != MG_OK) return rc; }
    // entry 0 of the alignment record, the chain length, is structure and stays
    int rc = mg_launch_set_params(cs->prim->ctx, values.data(), n * 8, al ? 7 : 0, cs->d_par, al ? cs->d_align + 1 : nullptr);
    if (rc == MG_OK) cs->structure.assign(cons, cons + n);
    return rc;
}
extern "C" int mg_constraint_set_create_aligned(mg_primitive *p, const mg_skeleton_desc *sk, const mg_keyframe_constraint *cons,
                                                int32_t n, const mg_alignment_desc *al, mg_constraint_set **out) {
    return mg_constraint_set_create_full(p, sk, cons, n, nullptr, 0, al, out);
}
extern "C" int mg_constraint_set_create_fk(mg_primitive *p, const mg_skeleton_desc *sk, const mg_keyframe_constraint *cons,
                                           int32_t n, mg_constraint_set **out) {
    return mg_constraint_set_create_aligned(p, sk, cons, n, nullptr, out);
}
extern "C" int mg_constraint_set_create(mg_primitive *p, const mg_keyframe_constraint *cons, int32_t n, mg_constraint_set **out) {
    return mg_constraint_set_create_aligned(p, nullptr, cons, n, nullptr, out);
}
extern "C" void mg_constraint_set_destroy(mg_constraint_set *cs) {
    if (!cs) return;
    if (cs->prim) (void)hipStreamSynchronize(cs->prim->ctx->stream);
    mg_dev_free(cs->prim->ctx, cs->d_W);
    mg_dev_free(cs->prim->ctx, cs->d_bias);
    mg_dev_free(cs->prim->ctx, cs->d_par);
    mg_dev_free(cs->prim->ctx, cs->d_woff);
    mg_dev_free(cs->prim->ctx, cs->d_chain);
    mg_dev_free(cs->prim->ctx, cs->d_choff);
    mg_dev_free(cs->prim->ctx, cs->d_Wpack);
    mg_dev_free(cs->prim->ctx, cs->d_bpad);
    mg_dev_free(cs->prim->ctx, cs->d_align);
    mg_dev_free(cs->prim->ctx, cs->d_pose);
    delete cs;
}

extern "C" int mg_score_constraints(mg_primitive *p, const mg_constraint_set *cs, const void *lat, int dt,
                                    int64_t B, int64_t ld, void *out, int odt) {
    int rc = mg_check_latents("mg_score_constraints", p, lat, dt, B, ld);
    if (rc != MG_OK) return rc;
    MG_REQUIRE(cs && cs->prim == p, "mg_score_constraints: constraint set is NULL or belongs to another primitive");
    MG_REQUIRE(odt == MG_F32 || odt == MG_F64, "mg_score_constraints: bad output dtype %d", odt);
    if (B == 0) return MG_OK;
    MG_REQUIRE(out != nullptr, "mg_score_constraints: output pointer is NULL");
    mg_prof_begin(p->ctx, 2);
    rc = mg_launch_score(p, cs, lat, dt, B, ld, out, odt, nullptr);
    mg_prof_end(p->ctx, 2);
    return rc;
}

// obj_spatial_error_sum_and_naturalness for a batch in ONE launch (reference optimization/objective_functions.py:163-185):
// error_scale * sum of the weighted constraint errors + quality_scale * (-log p(s)).
extern "C" int mg_objective_error_and_naturalness(mg_primitive *p, const mg_constraint_set *cs, const void *lat, int dt, int64_t B, int64_t ld,
                                                  double error_scale, double quality_scale, double *logp_out, double *err_out, double *obj_out) {
    int rc = mg_check_latents("mg_objective_error_and_naturalness", p, lat, dt, B, ld);
    if (rc != MG_OK) return rc;
    MG_REQUIRE(cs && cs->prim == p, "mg_objective_error_and_naturalness: constraint set is NULL or belongs to another primitive");
    MG_REQUIRE(p->K > 0, "mg_objective_error_and_naturalness: the primitive has no mixture");
    MG_REQUIRE(ld >= p->Lg, "mg_objective_error_and_naturalness: ld = %lld < the mixture's %d dimensions", (long long)ld, p->Lg);
    if (B == 0) return MG_OK;
    MG_REQUIRE(logp_out || err_out || obj_out, "mg_objective_error_and_naturalness: no output pointer");
    if (mg_objective_can_fuse(p, cs)) {
        mg_prof_begin(p->ctx, 1);
        rc = mg_launch_objective(p, cs, lat, dt, B, ld, error_scale, quality_scale, logp_out, err_out, obj_out);
        mg_prof_end(p->ctx, 1);
        return rc;
    }
    mg_set_error("mg_objective_error_and_naturalness: this mixture / constraint set does not run on the one-launch kernel (mixture over time "
                 "latents, more than 64 dimensions, tables beyond LDS, or the VALU scorer forced): call mg_gmm_log_prob and mg_score_constraints");
    return MG_ERR_UNSUPPORTED;
}

extern "C" int mg_score_constraint_residuals(mg_primitive *p, const mg_constraint_set *cs, const void *lat, int dt,
                                             int64_t B, int64_t ld, double *res) {
    int rc = mg_check_latents("mg_score_constraint_residuals", p, lat, dt, B, ld);
    if (rc != MG_OK) return rc;
    MG_REQUIRE(cs && cs->prim == p, "mg_score_constraint_residuals: constraint set is NULL or belongs to another primitive");
    if (B == 0 || cs->n == 0) return MG_OK;
    MG_REQUIRE(res != nullptr, "mg_score_constraint_residuals: output pointer is NULL");
    mg_prof_begin(p->ctx, 2);
    rc = mg_launch_score(p, cs, lat, dt, B, ld, nullptr, MG_F64, res);
    mg_prof_end(p->ctx, 2);
    return rc;
}

// The same residual matrix with EVERY candidate aligned to its own previous motion: align_cand (n, 4) = previous heading
// (x, z; unit) and previous root position (x, z) per candidate, in place of the set's one alignment record.
extern "C" int mg_score_constraint_residuals_chained(mg_primitive *p, const mg_constraint_set *cs, const void *lat, int dt, int64_t B, int64_t ld,
                                                     const double *align_cand, double *res) {
    int rc = mg_check_latents("mg_score_constraint_residuals_chained", p, lat, dt, B, ld);
    if (rc != MG_OK) return rc;
    MG_REQUIRE(cs && cs->prim == p, "mg_score_constraint_residuals_chained: constraint set is NULL or belongs to another primitive");
    MG_REQUIRE(cs->d_align != nullptr && cs->align_joint >= 0, "mg_score_constraint_residuals_chained: the set needs a previous-frame alignment (its node and reference vector; the values are per candidate)");
    if (B == 0 || cs->n == 0) return MG_OK;
    MG_REQUIRE(res != nullptr && align_cand != nullptr, "mg_score_constraint_residuals_chained: NULL pointer");
    mg_prof_begin(p->ctx, 2);
    rc = mg_launch_score(p, cs, lat, dt, B, ld, nullptr, MG_F64, res, align_cand);
    mg_prof_end(p->ctx, 2);
    return rc;
}

extern "C" int mg_gmm_log_prob_jac(mg_primitive *p, const void *x, int xdt, int64_t B, int64_t ld, double *jac) {
    int rc = mg_check_mixture_rows("mg_gmm_log_prob_jac", p, x, xdt, B, ld);
    if (rc != MG_OK) return rc;
    MG_REQUIRE(p->K > 0, "mg_gmm_log_prob_jac: primitive has no mixture");
    if (B == 0) return MG_OK;
    MG_REQUIRE(jac != nullptr, "mg_gmm_log_prob_jac: output pointer is NULL");
    mg_prof_begin(p->ctx, 1);
    rc = mg_launch_gmm_jac(p, x, xdt, B, ld, jac);
    mg_prof_end(p->ctx, 1);
    return rc;
}

extern "C" int mg_argmin_first_dev(mg_context *ctx, const void *v, int dt, int64_t n, void *out_dev) {
    MG_REQUIRE(ctx && out_dev && n >= 0 && (n == 0 || v), "mg_argmin_first_dev: bad arguments");
    MG_REQUIRE(dt == MG_F32 || dt == MG_F64, "mg_argmin_first_dev: bad dtype %d", dt);
    { int rc0 = mg_use_device(ctx); if (rc0 != MG_OK) return rc0; }
    mg_prof_begin(ctx, 3);
    int rc = mg_launch_argmin(ctx, v, dt, n, out_dev);
    mg_prof_end(ctx, 3);
    return rc;
}

extern "C" int mg_argmin_first(mg_context *ctx, const void *v, int dt, int64_t n, int64_t *best, double *minv) {
    MG_REQUIRE(ctx != nullptr, "mg_argmin_first: ctx is NULL");
    int rc = mg_argmin_first_dev(ctx, v, dt, n, ctx->argmin_out);
    if (rc != MG_OK) return rc;
    struct { int64_t i; double v; } h;
    MG_HIP_CHECK(hipMemcpyAsync(&h, ctx->argmin_out, 16, hipMemcpyDeviceToHost, ctx->stream));
    MG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (best) *best = h.i;
    if (minv) *minv = h.v;
    return MG_OK;
}

// One hot-path step: frames (float32, MFMA path) and log p(x) (float32) of the same latent batch, back to back
// on the context's stream.  (Running log p(x) on a side stream beside the persistent frames kernel was measured
// slower: 150 vs 142 us per step -- its f64 MFMAs steal VALU issue from the store sweep.)
extern "C" int mg_step_frames_and_logp(mg_primitive *p, const void *lat, int dt, int64_t B, int64_t ld,
                                       float *frames, float *logp) {
    int rc = mg_check_latents("mg_step_frames_and_logp", p, lat, dt, B, ld);
    if (rc != MG_OK) return rc;
    mg_context *ctx = p->ctx;
    const mg_time_grid *g = p->canonical;
    if (B >= 8 && g->T > 0 && frames && logp && mg_frames_can_fuse_gmm(p, g, B)) {
        // one launch: the mixture is scored by the sweep waves while the pipeline of the frames kernel fills
        return mg_launch_frames_mfma(p, g, lat, dt, B, ld, frames, logp, 0, 6);   // one kernel: slot "frames" and slot "step"
    }
    mg_prof_begin(ctx, 6);
    rc = mg_back_project_frames(p, nullptr, lat, dt, B, ld, frames, MG_PATH_AUTO);
    if (rc == MG_OK) rc = mg_gmm_log_prob(p, lat, dt, B, ld, logp, MG_F32);
    mg_prof_end(ctx, 6);
    return rc;
}

static int mg_step_plan_impl(const mg_primitive *p, int64_t B, const void *frames_dev, int32_t *plan);
extern "C" int mg_step_plan(const mg_primitive *p, int64_t B, int32_t *plan) { return mg_step_plan_impl(p, B, nullptr, plan); }
extern "C" int mg_step_plan_for(const mg_primitive *p, int64_t B, const void *frames_dev, int32_t *plan) { return mg_step_plan_impl(p, B, frames_dev, plan); }
static int mg_step_plan_impl(const mg_primitive *p, int64_t B, const void *frames_dev, int32_t *plan) {
    MG_REQUIRE(p && plan && B >= 0, "mg_step_plan: bad arguments");
    const mg_time_grid *g = p->canonical;
    plan[0] = plan[1] = plan[2] = plan[3] = 0;
    if (B == 0 || g->T == 0) return MG_OK;
    const bool fused = B >= 8 && mg_frames_can_fuse_gmm(p, g, B);
    if (!(g->mfma_ok && B >= 8)) {
        plan[2] = (int32_t)std::min<int64_t>((B * (int64_t)g->T * p->D + 255) / 256, (int64_t)p->ctx->n_cu * 32);
        return MG_OK;
    }
    const int which = mg_frames_kernel_choice(p, g, B, fused, frames_dev);
    MG_REQUIRE(which > 0, "mg_step_plan: the chunk-stationary kernel does not cover this shape");
    plan[0] = which;
    plan[1] = fused ? 1 : 0;
    plan[2] = mg_frames_grid(p, g, B, which);
    plan[3] = mg_frames_lds_bytes(p, g, which, fused);
    return MG_OK;
}

// Global positions of a list of joints in every frame of a (N, D) float64 frame block already on the device:
// out (N, n_out, 3).  The data-producing half of the feature maps under the cluster-tree builder (reference
// space_partitioning/features.py:133-153; construction/cluster_tree_builder.py:266-301).
extern "C" int mg_joint_positions(mg_context *ctx, const mg_skeleton_desc *sk, const int32_t *joints, int32_t n_out,
                                  const double *frames_dev, int64_t n_frames, int32_t n_dim, double *out_dev) {
    MG_REQUIRE(ctx && sk && joints && n_out > 0 && n_frames >= 0 && n_dim >= 3, "mg_joint_positions: bad arguments");
    MG_REQUIRE(sk->n_joints > 0 && sk->parents && sk->offsets && sk->quat_channel && sk->parents[0] < 0, "mg_joint_positions: incomplete skeleton");
    std::vector<double> table((size_t)n_out * (1 + 4 * MG_MAX_CHAIN), 0.0);
    for (int o = 0; o < n_out; o++) {
        MG_REQUIRE(joints[o] >= 0 && joints[o] < sk->n_joints, "mg_joint_positions: joint %d out of range", joints[o]);
        std::vector<int> ch;
        for (int j = joints[o]; j >= 0; j = sk->parents[j]) {
            MG_REQUIRE(sk->parents[j] < j, "mg_joint_positions: joint %d: parents must precede their children", j);
            ch.insert(ch.begin(), j);
        }
        const int m = (int)ch.size() - 1;
        MG_REQUIRE(m <= MG_MAX_CHAIN, "mg_joint_positions: chain of %d joints exceeds %d", m, MG_MAX_CHAIN);
        double *rec = &table[(size_t)o * (1 + 4 * MG_MAX_CHAIN)];
        rec[0] = m;
        for (int k = 0; k < m; k++) {   // link k: rotation of chain joint k, offset of chain joint k + 1
            const int qc = sk->quat_channel[ch[(size_t)k]];
            MG_REQUIRE(qc < 0 || qc + 4 <= n_dim, "mg_joint_positions: quaternion channel %d outside n_dim = %d", qc, n_dim);
            rec[1 + 4 * k] = qc;
            for (int e = 0; e < 3; e++) rec[2 + 4 * k + e] = sk->offsets[(size_t)ch[(size_t)k + 1] * 3 + e];
        }
    }
    if (n_frames == 0) return MG_OK;
    MG_REQUIRE(frames_dev && out_dev, "mg_joint_positions: NULL pointer");
    { int rc0 = mg_use_device(ctx); if (rc0 != MG_OK) return rc0; }
    void *scr = nullptr;
    int rc = mg_ctx_scratch(ctx, (int64_t)table.size() * 8, &scr);
    if (rc != MG_OK) return rc;
    MG_HIP_CHECK(hipMemcpyAsync(scr, table.data(), table.size() * 8, hipMemcpyHostToDevice, ctx->stream));
    MG_HIP_CHECK(hipStreamSynchronize(ctx->stream));   // table is a stack-lifetime host buffer
    return mg_launch_joint_positions(ctx, frames_dev, (const double *)scr, n_frames, n_dim, n_out, out_dev);
}

// ---------------------------------------------------------------------------------------
// host-pointer convenience variants
// ---------------------------------------------------------------------------------------
static size_t mg_dt_size(int dt) { return dt == MG_F64 ? 8 : 4; }

struct mg_host_io {
    mg_context *ctx;
    void *d_in = nullptr, *d_out = nullptr;
    int stage(mg_context *c, const void *in, int64_t in_bytes, int64_t out_bytes) {
        ctx = c;
        void *base = nullptr;
        int64_t in_al = (in_bytes + 255) / 256 * 256;
        int rc = mg_ctx_scratch(c, in_al + out_bytes + 256, &base);
        if (rc != MG_OK) return rc;
        d_in = base;
        d_out = (char *)base + in_al;
        if (in_bytes > 0) MG_HIP_CHECK(hipMemcpyAsync(d_in, in, (size_t)in_bytes, hipMemcpyHostToDevice, c->stream));
        return MG_OK;
    }
    int finish(void *out, int64_t out_bytes) {
        if (out_bytes > 0) MG_HIP_CHECK(hipMemcpyAsync(out, d_out, (size_t)out_bytes, hipMemcpyDeviceToHost, ctx->stream));
        MG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        return MG_OK;
    }
};

extern "C" int mg_back_project_frames_host(mg_primitive *p, const mg_time_grid *g, const void *lat, int dt,
                                           int64_t B, int64_t ld, float *frames, int path) {
    int rc = mg_check_latents("mg_back_project_frames_host", p, lat, dt, B, ld);
    if (rc != MG_OK) return rc;
    if (!g) g = p->canonical;
    int64_t ob = B * (int64_t)g->T * p->D * 4;
    mg_host_io io;
    if ((rc = io.stage(p->ctx, lat, B * ld * (int64_t)mg_dt_size(dt), ob)) != MG_OK) return rc;
    if ((rc = mg_back_project_frames(p, g, io.d_in, dt, B, ld, (float *)io.d_out, path)) != MG_OK) return rc;
    return io.finish(frames, ob);
}
extern "C" int mg_back_project_frames_f64_host(mg_primitive *p, const mg_time_grid *g, const void *lat, int dt,
                                               int64_t B, int64_t ld, double *frames) {
    int rc = mg_check_latents("mg_back_project_frames_f64_host", p, lat, dt, B, ld);
    if (rc != MG_OK) return rc;
    if (!g) g = p->canonical;
    int64_t ob = B * (int64_t)g->T * p->D * 8;
    mg_host_io io;
    if ((rc = io.stage(p->ctx, lat, B * ld * (int64_t)mg_dt_size(dt), ob)) != MG_OK) return rc;
    if ((rc = mg_back_project_frames_f64(p, g, io.d_in, dt, B, ld, (double *)io.d_out)) != MG_OK) return rc;
    return io.finish(frames, ob);
}
extern "C" int mg_back_project_coeffs_host(mg_primitive *p, const void *lat, int dt, int64_t B, int64_t ld,
                                           void *coeffs, int odt) {
    int rc = mg_check_latents("mg_back_project_coeffs_host", p, lat, dt, B, ld);
    if (rc != MG_OK) return rc;
    MG_REQUIRE(odt == MG_F32 || odt == MG_F64, "mg_back_project_coeffs_host: bad output dtype %d", odt);
    int64_t ob = B * (int64_t)p->R * (int64_t)mg_dt_size(odt);
    mg_host_io io;
    if ((rc = io.stage(p->ctx, lat, B * ld * (int64_t)mg_dt_size(dt), ob)) != MG_OK) return rc;
    if ((rc = mg_back_project_coeffs(p, io.d_in, dt, B, ld, io.d_out, odt)) != MG_OK) return rc;
    return io.finish(coeffs, ob);
}
extern "C" int mg_spline_evaluate_host(mg_primitive *p, const mg_time_grid *g, const double *coeffs, int64_t n, double *frames) {
    MG_REQUIRE(p != nullptr && n >= 0, "mg_spline_evaluate_host: bad arguments");
    if (!g) g = p->canonical;
    int64_t ib = n * (int64_t)p->R * 8, ob = n * (int64_t)g->T * p->D * 8;
    mg_host_io io;
    int rc;
    if ((rc = io.stage(p->ctx, coeffs, ib, ob)) != MG_OK) return rc;
    if ((rc = mg_spline_evaluate(p, g, (const double *)io.d_in, n, (double *)io.d_out)) != MG_OK) return rc;
    return io.finish(frames, ob);
}
extern "C" int mg_time_function_canonical_host(mg_primitive *p, const void *gamma, int gdt, int64_t B, int64_t ld, double *out) {
    MG_REQUIRE(p != nullptr && B >= 0 && (gdt == MG_F32 || gdt == MG_F64), "mg_time_function_canonical_host: bad arguments");
    int64_t ob = B * (int64_t)p->F * 8;
    mg_host_io io;
    int rc;
    if ((rc = io.stage(p->ctx, gamma, B * ld * (int64_t)mg_dt_size(gdt), ob)) != MG_OK) return rc;
    if ((rc = mg_time_function_canonical(p, io.d_in, gdt, B, ld, (double *)io.d_out)) != MG_OK) return rc;
    return io.finish(out, ob);
}
extern "C" int mg_gmm_log_prob_host(mg_primitive *p, const void *x, int xdt, int64_t B, int64_t ld, void *logp, int odt) {
    int rc = mg_check_mixture_rows("mg_gmm_log_prob_host", p, x, xdt, B, ld);
    if (rc != MG_OK) return rc;
    MG_REQUIRE(odt == MG_F32 || odt == MG_F64, "mg_gmm_log_prob_host: bad output dtype %d", odt);
    int64_t ob = B * (int64_t)mg_dt_size(odt);
    mg_host_io io;
    if ((rc = io.stage(p->ctx, x, B * ld * (int64_t)mg_dt_size(xdt), ob)) != MG_OK) return rc;
    if ((rc = mg_gmm_log_prob(p, io.d_in, xdt, B, ld, io.d_out, odt)) != MG_OK) return rc;
    return io.finish(logp, ob);
}
extern "C" int mg_gmm_sample_host(mg_primitive *p, int64_t n, const int64_t *counts, uint64_t seed, void *x, int xdt,
                                  int64_t ld, int32_t *comp) {
    MG_REQUIRE(p != nullptr && n >= 0 && ld >= (p ? p->Lg : 0), "mg_gmm_sample_host: bad arguments");
    MG_REQUIRE(xdt == MG_F32 || xdt == MG_F64, "mg_gmm_sample_host: bad dtype %d", xdt);
    int64_t xb = n * ld * (int64_t)mg_dt_size(xdt), cb = n * 4;
    void *dx = nullptr, *dc = nullptr;
    int rc = mg_device_malloc(p->ctx, xb, &dx);
    if (rc != MG_OK) return rc;
    rc = mg_device_malloc(p->ctx, cb, &dc);
    if (rc == MG_OK) rc = mg_memset(p->ctx, dx, 0, xb);
    if (rc == MG_OK) rc = mg_gmm_sample(p, n, counts, seed, dx, xdt, ld, (int32_t *)dc);
    if (rc == MG_OK) rc = mg_memcpy_d2h(p->ctx, x, dx, xb);
    if (rc == MG_OK && comp) rc = mg_memcpy_d2h(p->ctx, comp, dc, cb);
    (void)mg_device_free(p->ctx, dx);
    (void)mg_device_free(p->ctx, dc);
    return rc;
}
extern "C" int mg_score_constraint_residuals_host(mg_primitive *p, const mg_constraint_set *cs, const void *lat, int dt,
                                                  int64_t B, int64_t ld, double *res) {
    int rc = mg_check_latents("mg_score_constraint_residuals_host", p, lat, dt, B, ld);
    if (rc != MG_OK) return rc;
    MG_REQUIRE(cs && cs->prim == p, "mg_score_constraint_residuals_host: constraint set is NULL or belongs to another primitive");
    int64_t ob = B * (int64_t)cs->n * 8;
    mg_host_io io;
    if ((rc = io.stage(p->ctx, lat, B * ld * (int64_t)mg_dt_size(dt), ob)) != MG_OK) return rc;
    if ((rc = mg_score_constraint_residuals(p, cs, io.d_in, dt, B, ld, (double *)io.d_out)) != MG_OK) return rc;
    return io.finish(res, ob);
}
extern "C" int mg_gmm_log_prob_jac_host(mg_primitive *p, const void *x, int xdt, int64_t B, int64_t ld, double *jac) {
    int rc = mg_check_mixture_rows("mg_gmm_log_prob_jac_host", p, x, xdt, B, ld);
    if (rc != MG_OK) return rc;
    int64_t ob = B * (int64_t)p->Lg * 8;
    mg_host_io io;
    if ((rc = io.stage(p->ctx, x, B * ld * (int64_t)mg_dt_size(xdt), ob)) != MG_OK) return rc;
    if ((rc = mg_gmm_log_prob_jac(p, io.d_in, xdt, B, ld, (double *)io.d_out)) != MG_OK) return rc;
    return io.finish(jac, ob);
}
extern "C" int mg_score_constraints_host(mg_primitive *p, const mg_constraint_set *cs, const void *lat, int dt,
                                         int64_t B, int64_t ld, void *errors, int odt) {
    int rc = mg_check_latents("mg_score_constraints_host", p, lat, dt, B, ld);
    if (rc != MG_OK) return rc;
    MG_REQUIRE(odt == MG_F32 || odt == MG_F64, "mg_score_constraints_host: bad output dtype %d", odt);
    int64_t ob = B * (int64_t)mg_dt_size(odt);
    mg_host_io io;
    if ((rc = io.stage(p->ctx, lat, B * ld * (int64_t)mg_dt_size(dt), ob)) != MG_OK) return rc;
    if ((rc = mg_score_constraints(p, cs, io.d_in, dt, B, ld, io.d_out, odt)) != MG_OK) return rc;
    return io.finish(errors, ob);
}

// ---------------------------------------------------------------------------------------
// evaluate_samples_using_constraints in one call (reference motion_primitive_generator.py:230-261):
// score every candidate, first-minimum argmin, 16 bytes back -- no allocation, one synchronisation.
// ---------------------------------------------------------------------------------------
static int mg_best_candidate_impl(const char *who, mg_primitive *p, const mg_constraint_set *cs, const void *lat_dev, int dt,
                                  int64_t B, int64_t ld, void *err_dev, int64_t *best, double *minv) {
    MG_REQUIRE(cs && cs->prim == p, "%s: constraint set is NULL or belongs to another primitive", who);
    if (B == 0) {
        if (best) *best = 0;
        if (minv) *minv = INFINITY;
        return MG_OK;
    }
    int rc = mg_score_constraints(p, cs, lat_dev, dt, B, ld, err_dev, MG_F64);
    if (rc != MG_OK) return rc;
    return mg_argmin_first(p->ctx, err_dev, MG_F64, B, best, minv);
}
extern "C" int mg_best_candidate(mg_primitive *p, const mg_constraint_set *cs, const void *lat, int dt, int64_t B,
                                 int64_t ld, int64_t *best, double *minv) {
    int rc = mg_check_latents("mg_best_candidate", p, lat, dt, B, ld);
    if (rc != MG_OK) return rc;
    void *err = nullptr;
    if ((rc = mg_ctx_scratch(p->ctx, std::max<int64_t>(B, 1) * 8, &err)) != MG_OK) return rc;
    return mg_best_candidate_impl("mg_best_candidate", p, cs, lat, dt, B, ld, err, best, minv);
}
extern "C" int mg_best_candidate_host(mg_primitive *p, const mg_constraint_set *cs, const void *lat, int dt, int64_t B,
                                      int64_t ld, int64_t *best, double *minv) {
    int rc = mg_check_latents("mg_best_candidate_host", p, lat, dt, B, ld);
    if (rc != MG_OK) return rc;
    mg_host_io io;
    if ((rc = io.stage(p->ctx, lat, B * ld * (int64_t)mg_dt_size(dt), std::max<int64_t>(B, 1) * 8)) != MG_OK) return rc;
    return mg_best_candidate_impl("mg_best_candidate_host", p, cs, io.d_in, dt, B, ld, io.d_out, best, minv);
}

// One option of a planner step, enqueued without synchronisation (reference graph_walk_planner.py:184-226 evaluates
// the options one after the other): draw n candidates on the device, score them, first-minimum argmin, copy the
// winner next to the result.  result_dev: {int64 index, float64 error, float64 latent[n_components]}.
// rows [row_begin, row_begin + row_count) of the option's draw of n: one rank's share of a sharded step
extern "C" int mg_option_step_rows(mg_primitive *p, const mg_constraint_set *cs, int64_t n, const int64_t *counts, uint64_t seed,
                                   int64_t row_begin, int64_t row_count, void *x_dev, int xdt, int64_t ld, double *errors_dev, void *result_dev) {
    MG_REQUIRE(p && cs && cs->prim == p, "mg_option_step: constraint set is NULL or belongs to another primitive");
    MG_REQUIRE(n > 0 && row_count > 0 && x_dev && errors_dev && result_dev, "mg_option_step: bad arguments");
    int rc = mg_gmm_sample_rows(p, n, counts, seed, row_begin, row_count, x_dev, xdt, ld, nullptr);
    if (rc == MG_OK) rc = mg_score_constraints(p, cs, x_dev, xdt, row_count, ld, errors_dev, MG_F64);
    if (rc == MG_OK) {   // first minimum and the copy of the winner (at its full width) in one launch
        mg_prof_begin(p->ctx, 3);
        rc = mg_launch_argmin_gather(p->ctx, errors_dev, MG_F64, row_count, result_dev, x_dev, xdt, ld, p->Lg, row_begin);
        mg_prof_end(p->ctx, 3);
    }
    return rc;
}
extern "C" int mg_option_step(mg_primitive *p, const mg_constraint_set *cs, int64_t n, const int64_t *counts, uint64_t seed,
                              void *x_dev, int xdt, int64_t ld, double *errors_dev, void *result_dev) {
    return mg_option_step_rows(p, cs, n, counts, seed, 0, n, x_dev, xdt, ld, errors_dev, result_dev);
}

// A small device -> host read-back at the end of a step: through a pinned staging block of the context's (a copy into
// pageable memory goes through the runtime's own staging and costs tens of microseconds more), then one synchronisation.
// the context's pinned (page-locked, device-visible) staging block, at least `bytes` large
static int mg_ctx_pinned(mg_context *ctx, size_t bytes) {
    if (ctx->pinned_bytes < bytes) {
        MG_HIP_CHECK(hipStreamSynchronize(ctx->stream));   // a kernel may still be writing the old block
        if (ctx->pinned) (void)hipHostFree(ctx->pinned);
        ctx->pinned = nullptr; ctx->pinned_bytes = 0;
        const size_t cap = std::max<size_t>(bytes, 64 * 1024);
        MG_HIP_CHECK(hipHostMalloc(&ctx->pinned, cap, hipHostMallocDefault));
        memset(ctx->pinned, 0, cap);   // (completion flags below compare against sequence numbers that start at 1)
        ctx->pinned_bytes = cap;
    }
    return MG_OK;
}
// A planner step's results without a copy and without the runtime's synchronisation: the step's kernel writes the records into
// pinned memory and then, per option, the step's sequence number into a flag word (mg_options.hip); the host spins on the
// flags -- a few hundred nanoseconds after the last workgroup's write instead of the microseconds hipStreamSynchronize takes
// to wake up.  The flags live in a pinned block of their own (nothing else is ever written there) and are zeroed by the host
// before every launch that will be waited for: a flag can only ever read 0 or a sequence number a kernel wrote for THIS slot.
static int mg_ctx_flags(mg_context *ctx, size_t n, unsigned long long **out) {
    if (ctx->flag_cap < n) {
        MG_HIP_CHECK(hipStreamSynchronize(ctx->stream));   // a kernel may still be writing the old block
        if (ctx->flag_block) (void)hipHostFree(ctx->flag_block);
        ctx->flag_block = nullptr; ctx->flag_cap = 0;
        const size_t cap = std::max<size_t>(n, 64);
        MG_HIP_CHECK(hipHostMalloc((void **)&ctx->flag_block, cap * sizeof(unsigned long long), hipHostMallocDefault));
        ctx->flag_cap = cap;
    }
    // (no step is in flight here: every call that hands flags to a kernel waits for them before it returns, or fails the context's stream)
    for (size_t k = 0; k < n; k++) ((volatile unsigned long long *)ctx->flag_block)[k] = 0ull;
    __atomic_thread_fence(__ATOMIC_RELEASE);
    *out = ctx->flag_block;
    return MG_OK;
}
// Bounded by the CLOCK: after 2 ms of polling the stream is synchronised the ordinary way (which also surfaces a failed launch
// -- an iteration count made that wait 0.1-0.5 s).
static int mg_wait_flags(mg_context *ctx, const volatile unsigned long long *flags, int n, unsigned long long seq) {
    struct timespec t0 = {0, 0};
    bool have_t0 = false, synced = false;
    for (int k = 0; k < n; k++) {
        unsigned spins = 0;
        while (flags[k] != seq && !synced) {
            __builtin_ia32_pause();
            if ((++spins & 1023u) != 0) continue;
            struct timespec t1;
            clock_gettime(CLOCK_MONOTONIC, &t1);
            if (!have_t0) { t0 = t1; have_t0 = true; continue; }
            if ((t1.tv_sec - t0.tv_sec) * 1000000000L + (t1.tv_nsec - t0.tv_nsec) > 2000000L) {
                MG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
                synced = true;
            }
        }
        if (flags[k] != seq) { mg_set_error("mg_options_step: the step's kernel finished without writing its result records"); return MG_ERR_HIP; }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    return MG_OK;
}
static int mg_read_back_pinned(mg_context *ctx, void *dst, const void *src_dev, size_t bytes) {
    { int rcp = mg_ctx_pinned(ctx, bytes); if (rcp != MG_OK) return rcp; }
    MG_HIP_CHECK(hipMemcpyAsync(ctx->pinned, src_dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
    MG_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    memcpy(dst, ctx->pinned, bytes);
    return MG_OK;
}

static int mg_ctx_side_streams(mg_context *ctx) {
    if (ctx->side[0]) return MG_OK;
    for (auto &st : ctx->side) MG_HIP_CHECK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    for (auto &e : ctx->side_ev) MG_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return MG_OK;
}

// All outgoing options of a planner step in one call: mg_option_step for each (the primitives share one context, so
// one stream), their result records side by side in results_dev (record k at k * result_stride bytes), then -- if
// results_host is given -- ONE copy back and ONE synchronisation for the whole step.
extern "C" int mg_options_step(int32_t n_options, mg_primitive *const *prims, const mg_constraint_set *const *csets, int64_t n,
                               const int64_t *const *counts, const uint64_t *seeds, void *const *x_dev, int xdt, const int64_t *ld,
                               double *const *errors_dev, void *results_dev, int64_t result_stride, void *results_host) {
    return mg_options_step_rows(n_options, prims, csets, n, counts, seeds, 0, n, x_dev, xdt, ld, errors_dev, results_dev, result_stride, results_host);
}
// The step with the component counts drawn on the device (mg_options.hip: mg_options_counts_kernel): two launches, no host
// work per option.  counts_host (may be NULL): [n_options][16] int64, the counts the device drew.
extern "C" int mg_options_step_device_counts(int32_t n_options, mg_primitive *const *prims, const mg_constraint_set *const *csets, int64_t n,
                                             const uint64_t *seeds, void *const *x_dev, int xdt, const int64_t *ld,
                                             double *const *errors_dev, void *results_dev, int64_t result_stride, void *results_host,
                                             int64_t *counts_host) {
    MG_REQUIRE(n_options > 0 && n_options <= 24 && prims && csets && seeds && x_dev && ld && errors_dev && results_dev && n > 0,
               "mg_options_step_device_counts: bad arguments (at most 24 options)");
    mg_context *ctx = prims[0] ? prims[0]->ctx : nullptr;
    for (int k = 0; k < n_options; k++) {
        MG_REQUIRE(prims[k] && prims[k]->ctx == ctx, "mg_options_step_device_counts: option %d is NULL or lives in another context", k);
        MG_REQUIRE(result_stride >= 16 + 8 * (int64_t)prims[k]->Lg && result_stride % 8 == 0, "mg_options_step_device_counts: result_stride %lld too small for option %d",
                   (long long)result_stride, k);
        MG_REQUIRE(csets[k] && csets[k]->prim == prims[k], "mg_options_step_device_counts: constraint set %d is NULL or belongs to another primitive", k);
        MG_REQUIRE(x_dev[k] && errors_dev[k] && ld[k] >= prims[k]->Lg, "mg_options_step_device_counts: bad arguments for option %d", k);
    }
    MG_REQUIRE(xdt == MG_F32 || xdt == MG_F64, "mg_options_step_device_counts: bad dtype %d", xdt);
    { int rc0 = mg_use_device(ctx); if (rc0 != MG_OK) return rc0; }
    if (!mg_options_can_fuse(n_options, prims, csets, n)) {
        mg_set_error("mg_options_step_device_counts: an option does not run on the one-launch kernel (mixture of more than 16 components or 64 dimensions, "
                     "constraint tables beyond LDS, or a VALU kernel forced): draw the counts on the host and call mg_options_step");
        return MG_ERR_UNSUPPORTED;
    }
    const size_t rec_bytes = ((size_t)n_options * (size_t)result_stride + 63) / 64 * 64,
                 cnt_bytes = (size_t)n_options * MG_SAMPLE_ARG_K * sizeof(int32_t);
    char *rec_host = nullptr;
    int32_t *cnt_host = nullptr;
    unsigned long long *flags = nullptr;
    if (results_host || counts_host) {
        int rcp = mg_ctx_pinned(ctx, rec_bytes + cnt_bytes);
        if (rcp != MG_OK) return rcp;
        rec_host = (char *)ctx->pinned;              // (the records are what the flags wait for: written whenever anything is wanted)
        rcp = mg_ctx_flags(ctx, (size_t)n_options, &flags);
        if (rcp != MG_OK) return rcp;
        if (counts_host) cnt_host = (int32_t *)(rec_host + rec_bytes);
    }
    const unsigned long long seq = ++ctx->fused_seq;
    int rc = mg_launch_options_fused(n_options, prims, csets, n, nullptr, seeds, x_dev, xdt, ld, errors_dev, results_dev, result_stride, 0, n, rec_host, cnt_host,
                                     flags, seq);
    if (rc != MG_OK) return rc;
    if (results_host || counts_host) {
        int rcw = mg_wait_flags(ctx, flags, n_options, seq);   // (the counts were written by an earlier kernel of the same stream)
        if (rcw != MG_OK) return rcw;
        if (results_host) memcpy(results_host, rec_host, (size_t)n_options * (size_t)result_stride);
        if (counts_host)
            for (size_t i = 0; i < (size_t)n_options * MG_SAMPLE_ARG_K; i++) counts_host[i] = cnt_host[i];
    }
    return MG_OK;
}

// One rank's share of a sharded step: global rows [row_begin, row_begin + row_count) of every option's draw of n candidates;
// x_dev[k] (row_count, ld[k]) and errors_dev[k] (row_count) hold the block, the result records carry GLOBAL row indices.
extern "C" int mg_options_step_rows(int32_t n_options, mg_primitive *const *prims, const mg_constraint_set *const *csets, int64_t n,
                                    const int64_t *const *counts, const uint64_t *seeds, int64_t row_begin, int64_t row_count,
                                    void *const *x_dev, int xdt, const int64_t *ld,
                                    double *const *errors_dev, void *results_dev, int64_t result_stride, void *results_host) {
    MG_REQUIRE(n_options > 0 && prims && csets && counts && seeds && x_dev && ld && errors_dev && results_dev, "mg_options_step: bad arguments");
    MG_REQUIRE(row_begin >= 0 && row_count > 0 && row_begin + row_count <= n, "mg_options_step_rows: rows [%lld, %lld) outside the draw of %lld",
               (long long)row_begin, (long long)(row_begin + row_count), (long long)n);
    mg_context *ctx = prims[0] ? prims[0]->ctx : nullptr;
    for (int k = 0; k < n_options; k++) {
        MG_REQUIRE(prims[k] && prims[k]->ctx == ctx, "mg_options_step: option %d is NULL or lives in another context", k);
        MG_REQUIRE(result_stride >= 16 + 8 * (int64_t)prims[k]->Lg && result_stride % 8 == 0, "mg_options_step: result_stride %lld too small for option %d",
                   (long long)result_stride, k);
    }
    { int rc0 = mg_use_device(ctx); if (rc0 != MG_OK) return rc0; }
    for (int k = 0; k < n_options; k++) {
        MG_REQUIRE(csets[k] && csets[k]->prim == prims[k], "mg_options_step: constraint set %d is NULL or belongs to another primitive", k);
        MG_REQUIRE(n > 0 && counts[k] && x_dev[k] && errors_dev[k] && ld[k] >= prims[k]->Lg, "mg_options_step: bad arguments for option %d", k);
    }
    MG_REQUIRE(xdt == MG_F32 || xdt == MG_F64, "mg_options_step: bad dtype %d", xdt);
    // One launch for the whole step where every option runs on the matrix-pipe sampler and scorer (mg_options.hip);
    // more than MG_FUSED_MAX_OPTIONS options go in groups.
    {
        bool fused = true;
        for (int k0 = 0; k0 < n_options && fused; k0 += 24) fused = mg_options_can_fuse(std::min(24, n_options - k0), prims + k0, csets + k0, n);
        if (fused) {
            // results wanted on the host: the kernel leaves a second copy of every record in the context's pinned block -- no copy
            // operation behind the launch (a 4 us blit kernel and its launch gap), only the synchronisation
            char *rec_host = nullptr;
            unsigned long long *flags = nullptr;
            const size_t rec_bytes = ((size_t)n_options * (size_t)result_stride + 63) / 64 * 64;
            if (results_host) {
                int rcp = mg_ctx_pinned(ctx, rec_bytes);
                if (rcp != MG_OK) return rcp;
                rec_host = (char *)ctx->pinned;
                rcp = mg_ctx_flags(ctx, (size_t)n_options, &flags);
                if (rcp != MG_OK) return rcp;
            }
            const unsigned long long seq = ++ctx->fused_seq;
            for (int k0 = 0; k0 < n_options; k0 += 24) {
                int rcf = mg_launch_options_fused(std::min(24, n_options - k0), prims + k0, csets + k0, n, counts + k0, seeds + k0, x_dev + k0, xdt, ld + k0,
                                                  errors_dev + k0, (char *)results_dev + (size_t)k0 * result_stride, result_stride, row_begin, row_count,
                                                  rec_host ? rec_host + (size_t)k0 * result_stride : nullptr, nullptr, flags ? flags + k0 : nullptr, seq);
                if (rcf != MG_OK) return rcf;
            }
            if (results_host) {
                int rcw = mg_wait_flags(ctx, flags, n_options, seq);
                if (rcw != MG_OK) return rcw;
                memcpy(results_host, rec_host, (size_t)n_options * (size_t)result_stride);
            }
            return MG_OK;
        }
    }
    // Otherwise the options are independent chains of three small, launch-latency-bound kernels: four chains run side by side
    // on streams of the context's own (forked from and joined to its stream with events, so the call keeps stream semantics)
    // -- unless an option's sampler stages its prefix sums in the context's ONE scratch buffer (more than MG_SAMPLE_ARG_K
    // components, more than 64 mixture dimensions, or the VALU sampler forced): chains on different streams would overwrite
    // that buffer under each other, so those steps stay on one stream.
    bool shared_scratch = false;
    for (int k = 0; k < n_options; k++) shared_scratch = shared_scratch || !mg_gmm_sample_takes_host_prefix(prims[k]);
    const int S = (n_options >= 4 && !shared_scratch) ? 4 : 1;
    if (S > 1) {
        int rcs = mg_ctx_side_streams(ctx);
        if (rcs != MG_OK) return rcs;
    }
    hipStream_t main_stream = ctx->stream;
    if (S > 1) {
        MG_HIP_CHECK(hipEventRecord(ctx->side_ev[4], main_stream));
        for (int i = 0; i < S; i++) MG_HIP_CHECK(hipStreamWaitEvent(ctx->side[i], ctx->side_ev[4], 0));
    }
    int rc = MG_OK;
    for (int k = 0; k < n_options && rc == MG_OK; k++) {
        if (S > 1) ctx->stream = ctx->side[k % S];
        rc = mg_option_step_rows(prims[k], csets[k], n, counts[k], seeds[k], row_begin, row_count, x_dev[k], xdt, ld[k], errors_dev[k],
                                 (char *)results_dev + k * result_stride);
    }
    ctx->stream = main_stream;
    if (S > 1) {   // join even after an error: nothing may still be running behind the caller's back
        for (int i = 0; i < S; i++) {
            (void)hipEventRecord(ctx->side_ev[i], ctx->side[i]);
            (void)hipStreamWaitEvent(main_stream, ctx->side_ev[i], 0);
        }
    }
    if (rc != MG_OK) return rc;
    if (results_host) return mg_read_back_pinned(ctx, results_host, results_dev, (size_t)(n_options * result_stride));
    return MG_OK;
}

