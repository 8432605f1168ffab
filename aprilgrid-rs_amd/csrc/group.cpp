// group.cpp -- several GPUs of one node driven from ONE process through the C ABI
// (include/aprilgrid_amd.h, "detector groups"): frame sharding and the result gather of
// SURVEY.md 8(e) without torch.  Frames are independent (reference: detect(&self) only reads
// immutable fields, src/detector.rs:17-23,505), so rank r runs the whole saddle chain for its own
// frames on its own device and stream; the one exchange step is the gather of the per-rank result
// slabs (frame table + packed saddle records) to the root device.
//
// Transports of the gather:
//   AGX_GATHER_RCCL   ncclSend / ncclRecv in one ncclGroup over xGMI, enqueued on the detectors'
//                     streams behind their chains.  librccl is opened with dlopen when a group of MORE
//                     THAN ONE rank is created, so a single-GPU host (a group of one, or a process that
//                     never builds a group) does not need it: a group of one moves its slabs with two
//                     device-to-device copies whatever the transport.  AGX_GROUP_RCCL_SELF=1 (environment,
//                     read at agx_group_create; a test switch) makes a group of one bind the library and
//                     send its slabs to itself through it -- what a one-GPU box can exercise of librccl.
//   AGX_GATHER_PEER   hipMemcpyPeerAsync on the producer's stream + an event the root stream waits
//                     on.  Same data path over xGMI, no communicator; also the only transport that
//                     accepts the same device twice (a test configuration for one-GPU boxes).
#include <hip/hip_runtime_api.h>

#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "../../include/aprilgrid_amd.h"
#include "detector_internal.h"

namespace {

// The handful of RCCL entry points the gather needs (rccl.h is not included: the library is
// optional at run time and its handle types are opaque pointers).
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **comms, int ndev, const int *devlist) = nullptr;
    int (*CommDestroy)(void *comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*Send)(const void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t st) = nullptr;
    int (*Recv)(void *buf, size_t count, int dtype, int peer, void *comm, hipStream_t st) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
constexpr int kNcclUint8 = 1;  // ncclUint8 / ncclChar family: 0 = int8, 1 = uint8

// AGX_RCCL_LIBRARY (environment): the library to bind instead of librccl.  The test suite points it at a stand-in
// (tests/stub_rccl) that moves the bytes with peer copies, so that this file's RCCL branch runs on a one-GPU box --
// several ranks on one device, which the real library refuses.
const char *rccl_override()
{
    const char *p = getenv("AGX_RCCL_LIBRARY");
    return (p && *p) ? p : nullptr;
}

bool load_rccl(Rccl &r, std::string &err)
{
    if (r.lib) return true;
    if (const char *over = rccl_override()) {
        r.lib = dlopen(over, RTLD_NOW | RTLD_LOCAL);
    } else {
        // the soname first: a process that already holds an RCCL (e.g. the one bundled with PyTorch)
        // gets that copy back instead of a second runtime
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
    }
    if (!r.lib) {
        const char *why = dlerror();  // (a second call would return NULL: the message is consumed by the first)
        err = std::string("dlopen librccl: ") + (why ? why : "not found");
        return false;
    }
    auto sym = [&](const char *n) { return dlsym(r.lib, n); };
    r.CommInitAll = (decltype(r.CommInitAll))sym("ncclCommInitAll");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.GroupStart = (decltype(r.GroupStart))sym("ncclGroupStart");
    r.GroupEnd = (decltype(r.GroupEnd))sym("ncclGroupEnd");
    r.Send = (decltype(r.Send))sym("ncclSend");
    r.Recv = (decltype(r.Recv))sym("ncclRecv");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
    if (!r.CommInitAll || !r.CommDestroy || !r.GroupStart || !r.GroupEnd || !r.Send || !r.Recv) {
        err = "librccl lacks ncclCommInitAll / ncclSend / ncclRecv";
        dlclose(r.lib);
        r.lib = nullptr;
        return false;
    }
    return true;
}

}  // namespace

struct agx_group {
    int n = 0;
    int transport = AGX_GATHER_PEER;
    std::vector<int> devices;
    std::vector<agx_detector *> dets;
    Rccl rccl;
    std::vector<void *> comms;
    bool rccl_bound = false;  // the library is loaded and a communicator exists (n > 1, or n == 1 with AGX_GROUP_RCCL_SELF=1)
    // per rank, on its own device: result slabs of the last batch
    std::vector<float *> d_saddles;
    std::vector<uint32_t *> d_table;
    std::vector<hipEvent_t> done;  // PEER: rank r's copy to the root has been enqueued up to here
    // on the root device: every rank's slabs, rank-major
    float *d_all_saddles = nullptr;
    uint32_t *d_all_table = nullptr;
    // pinned host mirrors
    float *h_saddles = nullptr;
    uint32_t *h_table = nullptr;
    int frames_per_rank = 0;
    uint32_t slab_records = 0;  // records per rank slab
    size_t cap_frames = 0, cap_records = 0;
    bool enqueued = false;
    std::string last_error;
};

namespace {

thread_local std::string g_group_error;

int gfail(agx_group *g, int status, const std::string &msg)
{
    if (g) g->last_error = msg;
    else g_group_error = msg;
    return status;
}

// nothing unwinds across the C boundary (see agx_guard in detector.cpp)
void set_group_error_noexcept(agx_group *g, const char *msg) noexcept;

template <typename F>
int agx_group_guard(agx_group *g, F &&body) noexcept
{
    try {
        return body();
    } catch (const std::bad_alloc &) {
        set_group_error_noexcept(g, "out of host memory");
        return AGX_ERR_NOMEM;
    } catch (const std::exception &e) {
        set_group_error_noexcept(g, e.what());
        return AGX_ERR_STATE;
    } catch (...) {
        set_group_error_noexcept(g, "unknown exception");
        return AGX_ERR_STATE;
    }
}

#define GHIP(g, expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return gfail((g), AGX_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

void set_group_error_noexcept(agx_group *g, const char *msg) noexcept
{
    try {
        if (g) g->last_error = msg;
        else g_group_error = msg;
    } catch (...) {
    }
}

void free_slabs(agx_group *g)
{
    for (int r = 0; r < g->n; ++r) {
        (void)hipSetDevice(g->devices[r]);
        if (r < (int)g->d_saddles.size() && g->d_saddles[r]) (void)hipFree(g->d_saddles[r]);
        if (r < (int)g->d_table.size() && g->d_table[r]) (void)hipFree(g->d_table[r]);
    }
    g->d_saddles.assign(g->n, nullptr);
    g->d_table.assign(g->n, nullptr);
    if (g->n) (void)hipSetDevice(g->devices[0]);
    if (g->d_all_saddles) (void)hipFree(g->d_all_saddles);
    if (g->d_all_table) (void)hipFree(g->d_all_table);
    if (g->h_saddles) (void)hipHostFree(g->h_saddles);
    if (g->h_table) (void)hipHostFree(g->h_table);
    g->d_all_saddles = nullptr;
    g->d_all_table = nullptr;
    g->h_saddles = nullptr;
    g->h_table = nullptr;
    g->cap_frames = g->cap_records = 0;
}

int ensure_slabs(agx_group *g, int frames_per_rank, uint32_t slab_records)
{
    if ((size_t)frames_per_rank <= g->cap_frames && slab_records <= g->cap_records) return AGX_OK;
    for (agx_detector *d : g->dets) (void)agx_detector_sync(d);
    free_slabs(g);
    const size_t F = (size_t)frames_per_rank, R = slab_records;
    for (int r = 0; r < g->n; ++r) {
        GHIP(g, hipSetDevice(g->devices[r]));
        GHIP(g, hipMalloc((void **)&g->d_saddles[r], std::max<size_t>(R, 1) * 5 * sizeof(float)));
        GHIP(g, hipMalloc((void **)&g->d_table[r], F * 4 * sizeof(uint32_t)));
    }
    GHIP(g, hipSetDevice(g->devices[0]));
    GHIP(g, hipMalloc((void **)&g->d_all_saddles, (size_t)g->n * std::max<size_t>(R, 1) * 5 * sizeof(float)));
    GHIP(g, hipMalloc((void **)&g->d_all_table, (size_t)g->n * F * 4 * sizeof(uint32_t)));
    GHIP(g, hipHostMalloc((void **)&g->h_saddles, (size_t)g->n * std::max<size_t>(R, 1) * 5 * sizeof(float), hipHostMallocDefault));
    GHIP(g, hipHostMalloc((void **)&g->h_table, (size_t)g->n * F * 4 * sizeof(uint32_t), hipHostMallocDefault));
    g->cap_frames = F;
    g->cap_records = R;
    return AGX_OK;
}

}  // namespace

extern "C" {

const char *agx_group_last_error(const agx_group *g) { return g ? g->last_error.c_str() : g_group_error.c_str(); }

int agx_group_create(int family, const agx_params *params, const int *devices, int n_devices, int transport,
                     agx_group **out)
{
    return agx_group_guard(nullptr, [&]() -> int {
    if (!out) return AGX_ERR_ARG;
    *out = nullptr;
    if (n_devices < 1 || n_devices > 64) return gfail(nullptr, AGX_ERR_ARG, "n_devices must be 1..64");
    if (transport != AGX_GATHER_RCCL && transport != AGX_GATHER_PEER) return gfail(nullptr, AGX_ERR_ARG, "unknown gather transport");
    std::unique_ptr<agx_group> g(new agx_group());
    g->n = n_devices;
    g->transport = transport;
    for (int r = 0; r < n_devices; ++r) g->devices.push_back(devices ? devices[r] : r);
    const char *self_env = getenv("AGX_GROUP_RCCL_SELF");
    const bool use_rccl = transport == AGX_GATHER_RCCL && (n_devices > 1 || (self_env && self_env[0] == '1'));
    if (use_rccl) {
        // the library first: only the test suite's stand-in (it exports stub_rccl_stats) may take several ranks on one
        // device -- the real ncclCommInitAll would hang on them, whatever AGX_RCCL_LIBRARY says
        std::string err;
        if (!load_rccl(g->rccl, err)) return gfail(nullptr, AGX_ERR_HIP, err);
        const bool stand_in = dlsym(g->rccl.lib, "stub_rccl_stats") != nullptr;
        if (!stand_in)
            for (int r = 0; r < n_devices; ++r)
                for (int q = 0; q < r; ++q)
                    if (g->devices[q] == g->devices[r])
                        return gfail(nullptr, AGX_ERR_ARG, "the RCCL transport needs distinct devices (use AGX_GATHER_PEER to test on one GPU)");
    }
    g->d_saddles.assign(n_devices, nullptr);
    g->d_table.assign(n_devices, nullptr);
    auto cleanup = [&]() {
        for (agx_detector *d : g->dets) agx_detector_destroy(d);
        for (hipEvent_t e : g->done) (void)hipEventDestroy(e);
    };
    for (int r = 0; r < n_devices; ++r) {
        agx_detector *d = nullptr;
        const int st = agx_detector_create(family, params, g->devices[r], &d);
        if (st != AGX_OK) {
            g_group_error = std::string("rank ") + std::to_string(r) + ": " + agx_last_error(nullptr);
            cleanup();
            return st;
        }
        g->dets.push_back(d);
        hipEvent_t e = nullptr;
        if (hipSetDevice(g->devices[r]) != hipSuccess || hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
            cleanup();
            return gfail(nullptr, AGX_ERR_HIP, "hipEventCreate");
        }
        g->done.push_back(e);
    }
    if (transport == AGX_GATHER_PEER) {
        // direct xGMI access root <- rank (already-enabled is fine)
        for (int r = 1; r < n_devices; ++r)
            if (g->devices[r] != g->devices[0]) {
                (void)hipSetDevice(g->devices[r]);
                (void)hipDeviceEnablePeerAccess(g->devices[0], 0);
                (void)hipSetDevice(g->devices[0]);
                (void)hipDeviceEnablePeerAccess(g->devices[r], 0);
            }
        (void)hipGetLastError();
    } else if (use_rccl) {  // (one rank only with AGX_GROUP_RCCL_SELF=1: its slabs then go through the library as a send to itself)
        g->comms.assign(n_devices, nullptr);
        const int rc = g->rccl.CommInitAll(g->comms.data(), n_devices, g->devices.data());
        if (rc != 0) {
            cleanup();
            return gfail(nullptr, AGX_ERR_HIP, std::string("ncclCommInitAll: ") + (g->rccl.GetErrorString ? g->rccl.GetErrorString(rc) : "error"));
        }
        g->rccl_bound = true;
    }
    *out = g.release();
    return AGX_OK;
    });
}

void agx_group_destroy(agx_group *g)
{
    if (!g) return;
    try {
    for (agx_detector *d : g->dets) (void)agx_detector_sync(d);
    free_slabs(g);
    for (size_t r = 0; r < g->comms.size(); ++r)
        if (g->comms[r]) (void)g->rccl.CommDestroy(g->comms[r]);
    for (hipEvent_t e : g->done) (void)hipEventDestroy(e);
    for (agx_detector *d : g->dets) agx_detector_destroy(d);
    } catch (...) {
    }
    delete g;
}

int agx_group_size(const agx_group *g) { return g ? g->n : 0; }

agx_detector *agx_group_detector(agx_group *g, int rank) { return (g && rank >= 0 && rank < g->n) ? g->dets[rank] : nullptr; }

int agx_group_saddles_enqueue(agx_group *g, const void *const *d_frames, int frames_per_rank, int width, int height,
                              size_t row_stride_bytes, size_t frame_stride_bytes, int format, uint32_t records_per_frame)
{
    return agx_group_guard(g, [&]() -> int {
    if (!g || !d_frames || frames_per_rank <= 0) return gfail(g, AGX_ERR_ARG, "null frames or frames_per_rank <= 0");
    if (!records_per_frame) records_per_frame = 512;
    const unsigned long long slab64 = (unsigned long long)frames_per_rank * records_per_frame;
    if (slab64 > 0x7fffffffull) return gfail(g, AGX_ERR_ARG, "result slab too large");
    const uint32_t slab = (uint32_t)slab64;
    int rc = ensure_slabs(g, frames_per_rank, slab);
    if (rc) return rc;
    g->frames_per_rank = frames_per_rank;
    g->slab_records = slab;
    const size_t sad_bytes = (size_t)slab * 5 * sizeof(float), tab_bytes = (size_t)frames_per_rank * 4 * sizeof(uint32_t);
    // every rank's chain on its own device and stream, results in its own slabs
    for (int r = 0; r < g->n; ++r) {
        if (!d_frames[r]) return gfail(g, AGX_ERR_ARG, "null frame pointer for rank " + std::to_string(r));
        rc = agx_saddles_batch_enqueue_to(g->dets[r], d_frames[r], frames_per_rank, width, height, row_stride_bytes,
                                          frame_stride_bytes, format, g->d_saddles[r], slab, g->d_table[r]);
        if (rc) return gfail(g, rc, std::string("rank ") + std::to_string(r) + ": " + agx_last_error(g->dets[r]));
    }
    // the one exchange step: gather the slabs on the root device, stream-ordered behind the chains
    hipStream_t root = (hipStream_t)agx_internal_stream(g->dets[0]);
    GHIP(g, hipSetDevice(g->devices[0]));
    if (g->n == 1 && g->rccl_bound) {
        // A group of one under AGX_GROUP_RCCL_SELF=1 (test switch): the root's own slabs take the library's path -- a send
        // to itself and the matching receive in one ncclGroup on the root's stream.  (What a one-GPU box can exercise of
        // the real librccl: the binding, the communicator, the datatype constant, the ordering behind the chain on a
        // non-blocking stream.)  Without the switch a group of one never touches the library: two copies, below.
        int e = g->rccl.GroupStart();
        if (e == 0) e = g->rccl.Send(g->d_table[0], tab_bytes, kNcclUint8, 0, g->comms[0], root);
        if (e == 0) e = g->rccl.Recv(g->d_all_table, tab_bytes, kNcclUint8, 0, g->comms[0], root);
        if (e == 0) e = g->rccl.Send(g->d_saddles[0], sad_bytes, kNcclUint8, 0, g->comms[0], root);
        if (e == 0) e = g->rccl.Recv(g->d_all_saddles, sad_bytes, kNcclUint8, 0, g->comms[0], root);
        const int e2 = g->rccl.GroupEnd();
        if (e != 0 || e2 != 0)
            return gfail(g, AGX_ERR_HIP, std::string("RCCL self gather: ") + (g->rccl.GetErrorString ? g->rccl.GetErrorString(e ? e : e2) : "error"));
        g->enqueued = true;
        return AGX_OK;
    }
    GHIP(g, hipMemcpyAsync(g->d_all_table, g->d_table[0], tab_bytes, hipMemcpyDeviceToDevice, root));
    GHIP(g, hipMemcpyAsync(g->d_all_saddles, g->d_saddles[0], sad_bytes, hipMemcpyDeviceToDevice, root));
    if (g->n > 1 && g->rccl_bound) {
        int e = g->rccl.GroupStart();
        for (int r = 1; r < g->n && e == 0; ++r) {
            hipStream_t st = (hipStream_t)agx_internal_stream(g->dets[r]);
            e = g->rccl.Send(g->d_table[r], tab_bytes, kNcclUint8, 0, g->comms[r], st);
            if (e == 0) e = g->rccl.Send(g->d_saddles[r], sad_bytes, kNcclUint8, 0, g->comms[r], st);
            if (e == 0) e = g->rccl.Recv((char *)g->d_all_table + (size_t)r * tab_bytes, tab_bytes, kNcclUint8, r, g->comms[0], root);
            if (e == 0) e = g->rccl.Recv((char *)g->d_all_saddles + (size_t)r * sad_bytes, sad_bytes, kNcclUint8, r, g->comms[0], root);
        }
        const int e2 = g->rccl.GroupEnd();
        if (e != 0 || e2 != 0)
            return gfail(g, AGX_ERR_HIP, std::string("RCCL gather: ") + (g->rccl.GetErrorString ? g->rccl.GetErrorString(e ? e : e2) : "error"));
    } else {
        for (int r = 1; r < g->n; ++r) {
            hipStream_t st = (hipStream_t)agx_internal_stream(g->dets[r]);
            GHIP(g, hipSetDevice(g->devices[r]));
            GHIP(g, hipMemcpyPeerAsync((char *)g->d_all_table + (size_t)r * tab_bytes, g->devices[0], g->d_table[r], g->devices[r], tab_bytes, st));
            GHIP(g, hipMemcpyPeerAsync((char *)g->d_all_saddles + (size_t)r * sad_bytes, g->devices[0], g->d_saddles[r], g->devices[r], sad_bytes, st));
            GHIP(g, hipEventRecord(g->done[r], st));
            GHIP(g, hipSetDevice(g->devices[0]));
            GHIP(g, hipStreamWaitEvent(root, g->done[r], 0));
        }
    }
    g->enqueued = true;
    return AGX_OK;
    });
}

int agx_group_saddles_fetch(agx_group *g, agx_saddle *out, uint32_t cap_per_frame, uint32_t *counts, int *frame_status)
{
    return agx_group_guard(g, [&]() -> int {
    if (!g || !counts || (!out && cap_per_frame)) return gfail(g, AGX_ERR_ARG, "null output");
    if (!g->enqueued) return gfail(g, AGX_ERR_STATE, "no batch enqueued");
    const size_t F = (size_t)g->frames_per_rank;
    const size_t sad_bytes = (size_t)g->slab_records * 5 * sizeof(float), tab_bytes = F * 4 * sizeof(uint32_t);
    hipStream_t root = (hipStream_t)agx_internal_stream(g->dets[0]);
    GHIP(g, hipSetDevice(g->devices[0]));
    GHIP(g, hipMemcpyAsync(g->h_table, g->d_all_table, (size_t)g->n * tab_bytes, hipMemcpyDeviceToHost, root));
    GHIP(g, hipMemcpyAsync(g->h_saddles, g->d_all_saddles, (size_t)g->n * sad_bytes, hipMemcpyDeviceToHost, root));
    GHIP(g, hipStreamSynchronize(root));
    for (int r = 1; r < g->n; ++r) {  // the senders' streams have nothing left either
        const int rc = agx_detector_sync(g->dets[r]);
        if (rc) return gfail(g, rc, agx_last_error(g->dets[r]));
    }
    int first_bad = AGX_OK;
    for (int r = 0; r < g->n; ++r) {
        const uint32_t *tab = g->h_table + (size_t)r * F * 4;
        const float *sad = g->h_saddles + (size_t)r * g->slab_records * 5;
        for (size_t f = 0; f < F; ++f) {
            const size_t gf = (size_t)r * F + f;  // global frame index: rank-major, as the frames were sharded
            const uint32_t cnt = tab[f * 4 + 0], off = tab[f * 4 + 1], status = tab[f * 4 + 2];
            int st = AGX_OK;
            if ((status & (AGX_FRAME_CANDIDATE_OVERFLOW | AGX_FRAME_CLUSTER_OVERFLOW | AGX_FRAME_SADDLE_OVERFLOW)) || cnt > cap_per_frame ||
                (unsigned long long)off + cnt > g->slab_records)
                st = AGX_ERR_CAPACITY;
            // as agx_saddles_batch_fetch: a list that is merely longer than the caller's room reports its
            // length (so the caller can size a retry); a frame whose device-side lists overflowed reports 0
            const bool device_overflow = (status & (AGX_FRAME_CANDIDATE_OVERFLOW | AGX_FRAME_CLUSTER_OVERFLOW | AGX_FRAME_SADDLE_OVERFLOW)) ||
                                         (unsigned long long)off + cnt > g->slab_records;
            counts[gf] = device_overflow ? 0 : cnt;
            if (frame_status) frame_status[gf] = st;
            if (st != AGX_OK) {
                if (first_bad == AGX_OK) {
                    first_bad = st;
                    char buf[160];
                    std::snprintf(buf, sizeof buf, "rank %d frame %zu: capacity exceeded (status 0x%x, %u saddles)", r, f, status, cnt);
                    g->last_error = buf;
                }
                continue;
            }
            std::memcpy(out + gf * (size_t)cap_per_frame, sad + (size_t)off * 5, (size_t)cnt * sizeof(agx_saddle));
        }
    }
    return first_bad;
    });
}

}  // extern "C"
