// stub_rccl.cpp -- TEST-ONLY stand-in for librccl, selected by the environment variable AGX_RCCL_LIBRARY (group.cpp).
// It implements the seven entry points the detector groups bind by name -- with the prototypes of
// /opt/rocm/include/rccl/rccl.h -- on top of peer copies and events, so that group.cpp's RCCL branch (the dlsym'd
// function-pointer signatures, the ncclGroupStart / ncclGroupEnd bracket, one communicator per rank driven from a
// single thread, sends and receives enqueued on the ranks' own streams) runs on a ONE-GPU box with several ranks on
// the same device.  It checks what the real library would check (calls inside a group, known peers, the data type,
// matching byte counts) and counts what it saw; stub_rccl_stats() hands the counts to the test.
//
// Semantics kept from RCCL: a send is ordered behind the work already on the sender's stream, the matching receive
// completes in the receiver's stream order, and nothing moves before ncclGroupEnd.
#include <hip/hip_runtime_api.h>

#include <cstdint>
#include <cstdio>
#include <vector>

namespace {
struct World;
struct Comm {
    World *world;
    int rank, device;
};
struct Op {
    bool send;
    const void *src;
    void *dst;
    size_t bytes;
    int me, peer;
    hipStream_t st;
};
struct World {
    std::vector<Comm *> comms;
    int alive;
};
thread_local int g_depth = 0;
thread_local std::vector<Op> g_ops;
int g_stats[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // groups, sends, recvs, bytes (KiB), errors, comms created, comms destroyed, max ops in a group
enum { ncclSuccess = 0, ncclInvalidArgument = 4, ncclInvalidUsage = 5 };
int fail(int code)
{
    ++g_stats[4];
    return code;
}
}  // namespace

extern "C" {

int ncclCommInitAll(void **comms, int ndev, const int *devlist)
{
    if (!comms || ndev < 1) return fail(ncclInvalidArgument);
    World *w = new World();
    w->alive = ndev;
    for (int r = 0; r < ndev; ++r) {
        Comm *c = new Comm{w, r, devlist ? devlist[r] : r};
        w->comms.push_back(c);
        comms[r] = c;
        ++g_stats[5];
    }
    return ncclSuccess;
}

int ncclCommDestroy(void *comm)
{
    if (!comm) return fail(ncclInvalidArgument);
    Comm *c = static_cast<Comm *>(comm);
    World *w = c->world;
    delete c;
    ++g_stats[6];
    if (--w->alive == 0) delete w;
    return ncclSuccess;
}

int ncclGroupStart()
{
    ++g_depth;
    return ncclSuccess;
}

static int queue(bool send, const void *src, void *dst, size_t count, int dtype, int peer, void *comm, hipStream_t st)
{
    if (g_depth < 1) return fail(ncclInvalidUsage);  // group.cpp issues every send / recv inside one group
    if (!comm || (!src && !dst)) return fail(ncclInvalidArgument);
    if (dtype != 1) return fail(ncclInvalidArgument);  // ncclUint8: bytes
    Comm *c = static_cast<Comm *>(comm);
    if (peer < 0 || peer >= (int)c->world->comms.size() || peer == c->rank) return fail(ncclInvalidArgument);
    g_ops.push_back(Op{send, src, dst, count, c->rank, peer, st});
    ++g_stats[send ? 1 : 2];
    return ncclSuccess;
}
int ncclSend(const void *sendbuff, size_t count, int datatype, int peer, void *comm, hipStream_t stream)
{
    return queue(true, sendbuff, nullptr, count, datatype, peer, comm, stream);
}
int ncclRecv(void *recvbuff, size_t count, int datatype, int peer, void *comm, hipStream_t stream)
{
    return queue(false, nullptr, recvbuff, count, datatype, peer, comm, stream);
}

int ncclGroupEnd()
{
    if (g_depth < 1) return fail(ncclInvalidUsage);
    if (--g_depth) return ncclSuccess;
    ++g_stats[0];
    if ((int)g_ops.size() > g_stats[7]) g_stats[7] = (int)g_ops.size();
    // match the k-th send (me -> peer) with the k-th receive (peer <- me), in issue order, as RCCL does
    std::vector<bool> used(g_ops.size(), false);
    int rc = ncclSuccess;
    for (size_t i = 0; i < g_ops.size() && rc == ncclSuccess; ++i) {
        if (!g_ops[i].send) continue;
        size_t j = 0;
        for (; j < g_ops.size(); ++j)
            if (!used[j] && !g_ops[j].send && g_ops[j].me == g_ops[i].peer && g_ops[j].peer == g_ops[i].me) break;
        if (j == g_ops.size() || g_ops[j].bytes != g_ops[i].bytes) {
            rc = fail(ncclInvalidUsage);
            break;
        }
        used[j] = used[i] = true;
        // the copy rides the sender's stream; the receiver's stream waits for it
        hipEvent_t e = nullptr;
        if (hipMemcpyAsync(g_ops[j].dst, g_ops[i].src, g_ops[i].bytes, hipMemcpyDeviceToDevice, g_ops[i].st) != hipSuccess ||
            hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess || hipEventRecord(e, g_ops[i].st) != hipSuccess ||
            hipStreamWaitEvent(g_ops[j].st, e, 0) != hipSuccess)
            rc = fail(ncclInvalidUsage);
        if (e) (void)hipEventDestroy(e);  // (destruction is deferred until the event has completed)
        g_stats[3] += (int)(g_ops[i].bytes >> 10);
    }
    for (size_t i = 0; i < g_ops.size() && rc == ncclSuccess; ++i)
        if (!used[i]) rc = fail(ncclInvalidUsage);  // a receive without a send
    g_ops.clear();
    return rc;
}

const char *ncclGetErrorString(int code) { return code == ncclSuccess ? "no error" : (code == ncclInvalidArgument ? "invalid argument (stub)" : "invalid usage (stub)"); }

void stub_rccl_stats(int *out8)
{
    for (int i = 0; i < 8; ++i) out8[i] = g_stats[i];
}
}
