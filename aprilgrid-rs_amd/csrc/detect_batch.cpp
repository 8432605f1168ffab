// detect_batch.cpp -- TagDetector::detect (reference src/detector.rs:505-540) over a batch of
// frames: the saddle chain of a chunk of frames runs on the device while a pool of host threads
// runs the uploads and the board search + decode (detect's loop body, :510-539) of the chunks around it.
// The host tail is the reference's exhaustive search (about a millisecond per frame and thread), so the
// end-to-end rate is set by the number of host threads the process is granted; the pool keeps them busy
// and the device work disappears behind them.
#include <hip/hip_runtime_api.h>

#include <sched.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <exception>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/aprilgrid_amd.h"
#include "detector_internal.h"
#include "host_tail.hpp"
#include "tail_kernels.h"

namespace agx {

// A fixed set of worker threads with a task queue; wait() blocks until every submitted task is done.
// submit_front() puts a task ahead of everything queued (agx_detect_batch's uploads: a chunk of frames must not
// wait behind the hundreds of board searches already queued).
class WorkerPool {
public:
    explicit WorkerPool(int n)
    {
        threads_.reserve((size_t)std::max(n, 0));
        try {
            for (int i = 0; i < n; ++i) threads_.emplace_back([this] { run(); });
        } catch (...) {  // no more threads (std::system_error): the ones already running are joined before the error leaves
            stop_and_join();
            throw;
        }
    }
    ~WorkerPool() { stop_and_join(); }
    void stop_and_join() noexcept
    {
        try {
            {
                std::lock_guard<std::mutex> lk(m_);
                stop_ = true;
            }
            cv_.notify_all();
            for (std::thread &t : threads_)
                if (t.joinable()) t.join();
        } catch (...) {
        }
    }
    WorkerPool(const WorkerPool &) = delete;
    WorkerPool &operator=(const WorkerPool &) = delete;
    int size() const { return (int)threads_.size(); }
    void submit(std::function<void()> f) { push(std::move(f), false); }
    void submit_front(std::function<void()> f) { push(std::move(f), true); }
    // Blocks until every submitted task is done; false if one of them ended in an exception since the last wait (a task must
    // not unwind out of its thread: run() catches, the caller of wait() learns of it)
    bool wait()
    {
        std::unique_lock<std::mutex> lk(m_);
        done_.wait(lk, [this] { return pending_ == 0; });
        const bool ok = !failed_;
        failed_ = false;
        return ok;
    }

private:
    void push(std::function<void()> f, bool front)
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            if (front) q_.push_front(std::move(f));
            else q_.push_back(std::move(f));
            ++pending_;
        }
        cv_.notify_one();
    }
    void run()
    {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [this] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;  // stop_ and nothing left
                f = std::move(q_.front());
                q_.pop_front();
            }
            bool threw = false;
            try {
                f();
            } catch (...) {
                threw = true;
            }
            {
                std::lock_guard<std::mutex> lk(m_);
                if (threw) failed_ = true;
                if (--pending_ == 0) done_.notify_all();
            }
        }
    }
    std::vector<std::thread> threads_;
    std::deque<std::function<void()>> q_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    size_t pending_ = 0;
    bool stop_ = false, failed_ = false;
};

void destroy_worker_pool(void *p) { delete static_cast<WorkerPool *>(p); }
void *create_worker_pool(int n_threads) { return new WorkerPool(n_threads); }

// Workers of one frame's board search (option "tail_threads"): n - 1 pool threads plus the caller.
class PoolTailWorkers : public TailWorkers {
public:
    explicit PoolTailWorkers(int n) : n_(n), pool_(n - 1) {}
    int size() const override { return n_; }
    void run(int n, const std::function<void(int)> &f) override
    {
        int submitted = 1;
        std::exception_ptr mine;
        try {
            for (int t = 1; t < n; ++t, ++submitted) pool_.submit([&f, t] { f(t); });
            if (n > 0) f(0);
        } catch (...) {  // (the tasks already queued hold a reference to f: they finish before the error leaves)
            mine = std::current_exception();
        }
        const bool ok = pool_.wait();
        if (mine) std::rethrow_exception(mine);
        if (!ok) throw std::bad_alloc();  // a share of the search ran out of memory on its thread
    }

private:
    int n_;
    WorkerPool pool_;
};
TailWorkers *create_tail_workers(int n_threads) { return n_threads > 1 ? new PoolTailWorkers(std::min(n_threads, 64)) : nullptr; }
void destroy_tail_workers(TailWorkers *w) { delete w; }

}  // namespace agx

using namespace agx;

// CPUs of this process: affinity mask, narrowed by the CPU quota of its cgroup and of the cgroups above it
// (v2: cpu.max "quota period" | "max period"; v1: cpu.cfs_quota_us / cpu.cfs_period_us)
static int cgroup_cpu_quota(const std::string &cg_root = "/sys/fs/cgroup", const char *proc_cgroup = "/proc/self/cgroup")
{
    auto read_pair = [](const std::string &path, long long &a, long long &b) -> bool {
        FILE *f = std::fopen(path.c_str(), "r");
        if (!f) return false;
        char buf[64] = {0};
        const bool got = std::fgets(buf, sizeof buf, f) != nullptr;
        std::fclose(f);
        if (!got) return false;
        if (!std::strncmp(buf, "max", 3)) { a = -1; b = 100000; return true; }
        return std::sscanf(buf, "%lld %lld", &a, &b) >= 1;
    };
    long long best = -1;  // CPUs, rounded up; -1 = no quota found
    auto take = [&](long long quota, long long period) {
        if (quota <= 0 || period <= 0) return;
        const long long cpus = std::max(1ll, (quota + period - 1) / period);
        best = best < 0 ? cpus : std::min(best, cpus);
    };
    // this process's cgroup path (v2: "0::/path"; v1: "N:cpu,cpuacct:/path")
    std::string v2_path, v1_path;
    if (FILE *f = std::fopen(proc_cgroup, "r")) {
        char line[512];
        while (std::fgets(line, sizeof line, f)) {
            std::string l(line);
            while (!l.empty() && (l.back() == '\n' || l.back() == '\r')) l.pop_back();
            const size_t c1 = l.find(':'), c2 = c1 == std::string::npos ? c1 : l.find(':', c1 + 1);
            if (c2 == std::string::npos) continue;
            const std::string ctrl = l.substr(c1 + 1, c2 - c1 - 1), path = l.substr(c2 + 1);
            if (ctrl.empty()) v2_path = path;
            else if (ctrl.find("cpu") != std::string::npos && ctrl.find("cpuset") == std::string::npos) v1_path = path;
        }
        std::fclose(f);
    }
    for (std::string p = v2_path;;) {  // the cgroup and every ancestor (inside a container the namespace root is "/")
        long long q = -1, per = 100000;
        if (read_pair(cg_root + p + (p.empty() || p.back() != '/' ? "/" : "") + "cpu.max", q, per)) take(q, per);
        if (p.empty() || p == "/") break;
        const size_t cut = p.find_last_of('/');
        p = cut == std::string::npos || cut == 0 ? "/" : p.substr(0, cut);
    }
    for (const char *ctrl : {"/cpu", "/cpu,cpuacct"}) {
        for (std::string p = v1_path.empty() ? "/" : v1_path;;) {
            long long q = -1, per = -1, dummy = 0;
            const std::string dir = cg_root + ctrl + p + (p.back() != '/' ? "/" : "");
            if (read_pair(dir + "cpu.cfs_quota_us", q, dummy) && read_pair(dir + "cpu.cfs_period_us", per, dummy)) take(q, per);
            if (p == "/") break;
            const size_t cut = p.find_last_of('/');
            p = cut == std::string::npos || cut == 0 ? "/" : p.substr(0, cut);
        }
    }
    return best < 0 ? 0 : (int)std::min<long long>(best, 1 << 20);
}

extern "C" int agx_debug_cgroup_cpu_quota(const char *cgroup_root, const char *proc_self_cgroup)
{
    try {
        if (!cgroup_root || !proc_self_cgroup) return AGX_ERR_ARG;
        return cgroup_cpu_quota(cgroup_root, proc_self_cgroup);
    } catch (...) {
        return AGX_ERR_NOMEM;
    }
}

extern "C" int agx_host_parallelism(void)
{
    try {
        static const int cached = [] {
            int n = (int)std::max(1u, std::thread::hardware_concurrency());
            cpu_set_t set;
            CPU_ZERO(&set);
            if (sched_getaffinity(0, sizeof set, &set) == 0 && CPU_COUNT(&set) > 0) n = std::min(n, CPU_COUNT(&set));
            const int quota = cgroup_cpu_quota();
            if (quota > 0) n = std::min(n, quota);
            return std::max(n, 1);
        }();
        return cached;
    } catch (...) {
        return 1;
    }
}

static int detect_batch_impl(agx_detector *det, const void *frames, const void *d_frames, int n_frames, int width,
                             int height, size_t row_stride_bytes, size_t frame_stride_bytes, int format, agx_tag *out,
                             uint32_t cap_per_frame, uint32_t *counts, int *frame_status, int n_threads);

extern "C" int agx_detect_batch(agx_detector *det, const void *frames, const void *d_frames, int n_frames, int width,
                                int height, size_t row_stride_bytes, size_t frame_stride_bytes, int format, agx_tag *out,
                                uint32_t cap_per_frame, uint32_t *counts, int *frame_status, int n_threads)
{
    int rc;
    try {
        rc = detect_batch_impl(det, frames, d_frames, n_frames, width, height, row_stride_bytes, frame_stride_bytes, format, out,
                               cap_per_frame, counts, frame_status, n_threads);
    } catch (...) {  // (the worker pool could not be created: nothing has run)
        rc = AGX_ERR_NOMEM;
    }
    // A failure of the call as a whole (not a frame's own capacity status): no frame's result is valid -- say so in every
    // frame's slot, so that a caller who looks at the per-frame arrays only cannot mistake an untouched slot for "no tags"
    if (rc != AGX_OK && rc != AGX_ERR_CAPACITY && counts && n_frames > 0) {
        for (int f = 0; f < n_frames; ++f) {
            counts[f] = 0;
            if (frame_status) frame_status[f] = rc;
        }
    }
    return rc;
}

// agx_detect_batch with option "device_tail": the board search and the decode run on the device behind the chain
// (tail_kernels.hip); per chunk one wait, and a few KB of tags come back instead of the saddle lists.  Frames the kernel
// hands back -- an angle comparison inside its guard band, a list beyond its fixed sizes -- take the host tail on the
// pool, as every frame does without the option: the results are the host tail's either way.
static int detect_batch_device_tail(agx_detector *det, const void *frames, const void *d_frames, int n_frames, int width, int height,
                                    size_t row_stride_bytes, size_t frame_stride_bytes, int format, agx_tag *out,
                                    uint32_t cap_per_frame, uint32_t *counts, int *frame_status, WorkerPool *pool)
{
    const FamilyInfo *fam = static_cast<const FamilyInfo *>(agx_internal_family(det));
    const int max_boards = agx_internal_max_boards(det);
    const int device = agx_internal_device(det);
    constexpr int S = AGX_UPLOAD_STREAMS;
    // A frame's search occupies a workgroup of eight waves for two to five milliseconds (frames whose first seed does not
    // find the whole board cost twice the others), one workgroup per CU: a launch takes what its slowest frame takes, and
    // only launches of several times the CU count average that out.  So: chunks of up to 1024 frames; several per call only
    // so that the next one's upload runs under this one's kernels.
    const int chunk = std::max(1, std::min(n_frames, 1024));
    const int n_chunks = (n_frames + chunk - 1) / chunk;
    const size_t chunk_bytes = (size_t)chunk * frame_stride_bytes;
    uint8_t *d_stage = nullptr;
    hipStream_t up[S] = {nullptr, nullptr, nullptr};
    if (!d_frames) {
        d_stage = static_cast<uint8_t *>(agx_internal_stage(det, (size_t)std::min(S, n_chunks) * chunk_bytes));
        if (!d_stage) return AGX_ERR_HIP;
        if (agx_internal_upload_streams(det, (void **)up) != 0) return AGX_ERR_HIP;
    }
    std::mutex m;
    std::condition_variable cv;
    std::vector<int> uploaded;  // per chunk: 0 pending, 1 on the device, -1 failed   (guarded by m)
    std::atomic<int> first_bad{AGX_OK};
    std::atomic<bool> nomem{false};
    std::deque<std::vector<agx_saddle>> handed_back;  // saddle lists of frames for the host tail (alive until the pool is drained)
    int rc = AGX_OK, n_fallback = 0, n_uncertain = 0;
    bool pending_batch = false;
    // A chunk goes up in P parts on P workers at once (a copy from pageable memory is staged by the calling thread at 10 .. 15 GB/s:
    // it takes four to six of them to fill the link), the chunks one after the other (three large chunks side by side would
    // share the link, and the first -- the one the device waits for -- would arrive with the third).  The parts form ONE
    // in-order list (index ci * P + part); up to P uploader tasks take the lowest part nobody has taken yet, copy it, and take
    // the next, until no released part is left -- then they end.  No task ever waits for another one: a pool of a single
    // thread uploads the parts one by one (a task that waited for "its" turn while lower-numbered parts were still queued
    // behind it deadlocked pools of <= 6 threads on calls of >= 4 chunks).  A chunk's parts are released when its staging
    // slot is free: the first S chunks at once, chunk ci + S when chunk ci's kernels are through.  (Measured on an 8 192-frame
    // stream, profiles/r6_upload_scheduling_ab.txt: 157.5 ms per call like round 5's gated tasks; with the uploaders waiting for
    // their stream instead of their own event 170 .. 194 ms.)
    const int P = std::max(1, std::min(6, pool->size()));
    int next_part = 0, released_parts = 0, live_uploaders = 0;  // (guarded by m)
    bool stop_uploads = false;                                   // an error ended the call: nothing more is taken (guarded by m)
    std::vector<int> parts_left;        // per chunk                                   (guarded by m)
    std::vector<char> chunk_failed;     //                                             (guarded by m)
    auto uploader = [&, device, P] {
        // An uploader waits for ITS copy (an event behind it), not for the stream to run empty: two uploaders share a stream,
        // and the other one -- and whoever takes the parts after it -- keeps enqueueing behind; waiting for the stream starved
        // single parts for tens of milliseconds and let later chunks overtake the one the device was waiting for.
        hipEvent_t mine = nullptr;
        if (hipSetDevice(device) != hipSuccess || hipEventCreateWithFlags(&mine, hipEventDisableTiming) != hipSuccess) mine = nullptr;  // (then: the stream)
        for (;;) {
            int idx = -1;
            {
                std::lock_guard<std::mutex> lk(m);
                if (stop_uploads || next_part >= released_parts) --live_uploaders;  // nothing left to take: this uploader ends
                else idx = next_part++;
            }
            if (idx < 0) break;
            const int ci = idx / P, part = idx % P;
            const int c0 = ci * chunk, nf = std::min(chunk, n_frames - c0), slot = ci % S;
            const int f0 = (int)((long long)nf * part / P), f1 = (int)((long long)nf * (part + 1) / P);
            const bool ok = f1 <= f0 ||
                            (hipSetDevice(device) == hipSuccess &&
                             hipMemcpyAsync(d_stage + (size_t)slot * chunk_bytes + (size_t)f0 * frame_stride_bytes,
                                            (const uint8_t *)frames + (size_t)(c0 + f0) * frame_stride_bytes, (size_t)(f1 - f0) * frame_stride_bytes,
                                            hipMemcpyHostToDevice, up[part % S]) == hipSuccess &&
                             (mine ? hipEventRecord(mine, up[part % S]) == hipSuccess && hipEventSynchronize(mine) == hipSuccess
                                   : hipStreamSynchronize(up[part % S]) == hipSuccess));
            {
                std::lock_guard<std::mutex> lk(m);
                if (!ok) chunk_failed[(size_t)ci] = 1;
                if (--parts_left[(size_t)ci] == 0) uploaded[(size_t)ci] = chunk_failed[(size_t)ci] ? -1 : 1;
            }
            cv.notify_all();
        }
        if (mine) (void)hipEventDestroy(mine);
    };
    // `n_more` chunks' parts may go up: tops the uploaders up to P (ahead of every queued host tail)
    auto release_chunks = [&, P](int n_more) {
        int spawn;
        {
            std::lock_guard<std::mutex> lk(m);
            released_parts += n_more * P;
            spawn = std::max(0, std::min(P - live_uploaders, released_parts - next_part));
            live_uploaders += spawn;
        }
        int started = 0;
        try {
            for (; started < spawn; ++started) pool->submit_front(uploader);
        } catch (...) {  // (a task could not be queued: the count must not include uploaders that will never run)
            std::lock_guard<std::mutex> lk(m);
            live_uploaders -= spawn - started;
            throw;
        }
    };
    try {
    uploaded.assign((size_t)n_chunks, 0);
    parts_left.assign((size_t)n_chunks, P);
    chunk_failed.assign((size_t)n_chunks, 0);
    if (!d_frames) release_chunks(std::min(S, n_chunks));
    const int tail_debug = agx_internal_tail_debug(det);  // AGX_TAIL_DEBUG as the handle read it when it was created
    std::vector<uint32_t> ns, offs;
    std::vector<int> fst;
    for (int ci = 0; ci < n_chunks; ++ci) {
        const int c0 = ci * chunk, nf = std::min(chunk, n_frames - c0);
        const uint8_t *h_chunk = (const uint8_t *)frames + (size_t)c0 * frame_stride_bytes;
        const void *d_chunk;
        if (!d_frames) {
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&] { return uploaded[(size_t)ci] != 0; });
            if (uploaded[(size_t)ci] < 0) { rc = AGX_ERR_HIP; break; }
            d_chunk = d_stage + (size_t)(ci % S) * chunk_bytes;
        } else {
            d_chunk = (const uint8_t *)d_frames + (size_t)c0 * frame_stride_bytes;
        }
        rc = agx_saddles_batch_enqueue(det, d_chunk, nf, width, height, row_stride_bytes, frame_stride_bytes, format);
        if (rc) break;
        pending_batch = true;
        const uint8_t *d_luma = static_cast<const uint8_t *>(d_chunk);  // detector.rs:507: L8 frames are their own u8 luma
        size_t luma_row = row_stride_bytes, luma_frame = frame_stride_bytes;
        if (format != AGX_L8) {
            rc = agx_internal_chunk_luma8(det, d_chunk, nf, width, height, row_stride_bytes, frame_stride_bytes, format, 0, 1, (size_t)chunk,
                                          nullptr, &d_luma);
            if (rc) break;
            luma_row = (size_t)width;
            luma_frame = (size_t)width * (size_t)height;
        }
        rc = agx_internal_enqueue_tail(det, d_luma, luma_row, luma_frame, std::max(cap_per_frame, 1u));
        if (rc) break;
        const agx_tag *tags = nullptr;
        const uint32_t *table = nullptr;
        uint32_t tag_cap = 0;
        rc = agx_internal_fetch_tail(det, &tags, &table, &tag_cap);  // waits for the device
        if (rc) break;
        bool any_back = false;
        if (tail_debug) {
            int h[32] = {0};
            for (int f = 0; f < nf; ++f)
                for (int b = 0; b < 32; ++b) h[b] += (table[4 * f + 1] >> b) & 1u;
            for (int b = 0; b < 32; ++b)
                if (h[b]) std::fprintf(stderr, "tail status bit %d: %d frames\n", b, h[b]);
            std::vector<std::pair<uint32_t, int>> tk;
            double sum = 0;
            for (int f = 0; f < nf; ++f) {
                tk.push_back({table[4 * f + 2], f});
                sum += table[4 * f + 2];
            }
            std::sort(tk.begin(), tk.end());
            std::fprintf(stderr, "tail ticks per frame (100 MHz): mean %.0f median %u p90 %u max %u; slowest:", sum / nf, tk[(size_t)nf / 2].first,
                         tk[(size_t)nf * 9 / 10].first, tk.back().first);
            for (int i = 0; i < 5 && i < nf; ++i) {
                const int f = tk[(size_t)(nf - 1 - i)].second;
                std::fprintf(stderr, " [frame %d: %u ticks, %u saddles, %u seeds, %u tags]", c0 + f, table[4 * f + 2], table[4 * f + 3] & 0xffff, table[4 * f + 3] >> 16, table[4 * f]);
            }
            std::fprintf(stderr, "\n");
        }
        for (int f = 0; f < nf; ++f) {
            const int gf = c0 + f;
            const uint32_t st = table[4 * f + 1], nt = table[4 * f];
            if (st != TAIL_OK) {
                any_back = true;
                continue;
            }
            counts[gf] = nt;
            int stf = AGX_OK;
            if (nt > cap_per_frame) {  // (the kernel was given min(cap_per_frame, 128) as its limit and hands frames beyond it back)
                stf = AGX_ERR_CAPACITY;
                int exp = AGX_OK;
                first_bad.compare_exchange_strong(exp, stf);
            } else if (nt) {
                std::memcpy(out + (size_t)gf * cap_per_frame, tags + (size_t)f * tag_cap, (size_t)nt * sizeof(agx_tag));
            }
            if (frame_status) frame_status[gf] = stf;
        }
        if (any_back) {
            // the saddle lists of the frames handed back (and the status of frames whose chain overflowed)
            ns.resize((size_t)nf);
            offs.resize((size_t)nf);
            fst.resize((size_t)nf);
            const agx_saddle *records = nullptr;
            rc = agx_internal_fetch_compact(det, &records, ns.data(), offs.data(), fst.data());
            pending_batch = false;
            if (rc) break;
            for (int f = 0; f < nf; ++f) {
                if (table[4 * f + 1] == TAIL_OK) continue;
                const int gf = c0 + f;
                if (fst[(size_t)f] != AGX_OK) {  // reported, never truncated
                    counts[gf] = 0;
                    if (frame_status) frame_status[gf] = fst[(size_t)f];
                    int exp = AGX_OK;
                    first_bad.compare_exchange_strong(exp, fst[(size_t)f]);
                    continue;
                }
                ++n_fallback;
                n_uncertain += (table[4 * f + 1] & TAIL_UNCERTAIN) != 0;
                handed_back.emplace_back(records + offs[(size_t)f], records + offs[(size_t)f] + ns[(size_t)f]);
                const std::vector<agx_saddle> *list = &handed_back.back();
                const uint8_t *img = h_chunk + (size_t)f * frame_stride_bytes;
                pool->submit([=, &first_bad, &nomem] {
                    try {
                        std::vector<uint8_t> grey;  // to_luma8 of an L16 / RGB8 frame, on the host this time
                        const uint8_t *g = img;
                        size_t gstride = row_stride_bytes;
                        if (format != AGX_L8) {
                            grey.resize((size_t)width * (size_t)height);
                            (void)luma8(img, width, height, row_stride_bytes, format, grey.data());
                            g = grey.data();
                            gstride = (size_t)width;
                        }
                        const std::vector<agx_tag> &tg = detect_tail_scratch(*fam, max_boards, list->data(), list->size(), g, width, height, gstride);
                        int stf = AGX_OK;
                        counts[gf] = (uint32_t)tg.size();
                        if (tg.size() > cap_per_frame) {
                            stf = AGX_ERR_CAPACITY;
                            int exp = AGX_OK;
                            first_bad.compare_exchange_strong(exp, stf);
                        } else if (!tg.empty()) {
                            std::memcpy(out + (size_t)gf * cap_per_frame, tg.data(), tg.size() * sizeof(agx_tag));
                        }
                        if (frame_status) frame_status[gf] = stf;
                    } catch (...) {
                        counts[gf] = 0;
                        if (frame_status) frame_status[gf] = AGX_ERR_NOMEM;
                        nomem.store(true);
                    }
                });
            }
        } else {
            agx_internal_abandon_batch(det);  // (the stream is idle: the batch is done with)
            pending_batch = false;
        }
        // chain, luma and tail have read the staging slot: the chunk S ahead may go up
        if (!d_frames && ci + S < n_chunks) release_chunks(1);
    }
    } catch (...) {
        rc = AGX_ERR_NOMEM;
    }
    if (rc) {  // the call has failed: the uploaders stop at their next part instead of copying the rest of the caller's frames
        std::lock_guard<std::mutex> lk(m);
        stop_uploads = true;
    }
    (void)pool->wait();
    if (pending_batch) agx_internal_abandon_batch(det);
    agx_internal_tail_stats(det, n_frames, n_fallback, n_uncertain);
    if (rc) return rc;
    if (nomem.load()) return AGX_ERR_NOMEM;
    return first_bad.load();
}

static int detect_batch_impl(agx_detector *det, const void *frames, const void *d_frames, int n_frames, int width,
                             int height, size_t row_stride_bytes, size_t frame_stride_bytes, int format, agx_tag *out,
                             uint32_t cap_per_frame, uint32_t *counts, int *frame_status, int n_threads)
{
    if (!det || !frames || !counts || n_frames <= 0 || (!out && cap_per_frame)) return AGX_ERR_ARG;
    if (format != AGX_L8 && format != AGX_L16 && format != AGX_RGB8) return AGX_ERR_FORMAT;  // the tail derives to_luma8 itself
    if (width < 2 || height < 2) return AGX_ERR_ARG;
    const size_t bpp = format == AGX_L8 ? 1 : (format == AGX_L16 ? 2 : 3);
    if (row_stride_bytes < (size_t)width * bpp || (n_frames > 1 && frame_stride_bytes < row_stride_bytes * (size_t)height)) return AGX_ERR_ARG;
    // one frame: the stride between frames means nothing to the caller (0 is a natural value), but the staging
    // and the upload below are sized by it -- use the frame's own extent
    if (n_frames == 1 && frame_stride_bytes < row_stride_bytes * (size_t)height) frame_stride_bytes = row_stride_bytes * (size_t)height;
    // 0 = every CPU this process may keep busy: its affinity mask or its cgroup quota, whichever is smaller (threads beyond a
    // quota are not merely idle: the quota is spent sooner and EVERY thread of the process, the one driving the device
    // included, is frozen for the rest of the scheduler period -- profiles/r5_host_cpu_quota_and_tail_scaling.txt)
    if (n_threads <= 0) n_threads = agx_host_parallelism();
    WorkerPool *pool = static_cast<WorkerPool *>(agx_internal_pool(det, n_threads));
    if (!pool) return AGX_ERR_ARG;
    const FamilyInfo *fam = static_cast<const FamilyInfo *>(agx_internal_family(det));
    const int max_boards = agx_internal_max_boards(det);
    if (hipSetDevice(agx_internal_device(det)) != hipSuccess) return AGX_ERR_HIP;
    const int device = agx_internal_device(det);
    // The device tail costs a launch whose length is its slowest frame's (1.5 .. 3.5 ms) whatever the batch; the host tail costs
    // ~0.9 ms per frame and thread.  Measured on 16 threads the two meet at ~60 frames (1: 1.0 / 1.7 ms, 16: 2.1 / 3.6,
    // 48: 4.3 / 4.4, 64: 4.6 / 4.7, 96: 6.8 / 5.3, 128: 8.9 / 6.0, 256: 15.7 / 8.5): left to choose, a call of fewer than four
    // frames per host thread keeps the host tail.
    const int tail_mode = agx_internal_device_tail(det);
    bool on_device = tail_mode == 1 || (tail_mode == 2 && n_frames >= 4 * pool->size());
    if (on_device) {  // the one-time set-up (code list, 155 KB of LDS for the kernel): asked for -> its failure is the call's;
        const int prc = agx_internal_tail_prepare(det);  // left to choose -> the host tail, which needs nothing from the device
        if (prc && tail_mode == 1) return prc;
        if (prc) on_device = false;
    }
    if (on_device)
        return detect_batch_device_tail(det, frames, d_frames, n_frames, width, height, row_stride_bytes, frame_stride_bytes, format, out,
                                        cap_per_frame, counts, frame_status, pool);
    agx_internal_tail_stats(det, 0, 0, 0);  // ("last_device_tail_frames" 0: this call's tails run on the host)

    // Chunks of about one frame per worker (8 .. 64): the chain of a chunk takes 0.1 ms on the device, a frame's board
    // search about a millisecond on a host thread, so small chunks cost nothing and the workers start after the first 8 .. 64 frames
    // instead of after a quarter of the batch.  Three kinds of work, none of which waits for another chunk's:
    //   uploads   (host frames only) pool tasks that jump the queue: a copy from pageable memory occupies the calling thread
    //             for its duration (0.3 ms per 16 MB chunk at the PCIe rate: profiles/r5_ubench_h2d_pageable.txt) -- it runs
    //             on the workers, up to AGX_UPLOAD_STREAMS chunks ahead, not on the thread that drives the device;
    //   chain     this thread: enqueue, luma (L16 / RGB8), one wait per chunk, the compact list into the chunk's slot;
    //   tails     pool tasks, one per frame, reading the slot; a slot is refilled when ITS tails are done (a counter per
    //             slot) -- no barrier over the pool between chunks.
    constexpr int S = AGX_UPLOAD_STREAMS, R = 4;
    const int chunk = std::max(1, std::min(n_frames, std::min(std::max(pool->size(), 8), 64)));
    const int n_chunks = (n_frames + chunk - 1) / chunk;
    const size_t chunk_bytes = (size_t)chunk * frame_stride_bytes;
    uint8_t *d_stage = nullptr;
    hipStream_t up[S] = {nullptr, nullptr, nullptr};
    if (!d_frames) {
        d_stage = static_cast<uint8_t *>(agx_internal_stage(det, (size_t)std::min(S, n_chunks) * chunk_bytes));
        if (!d_stage) return AGX_ERR_HIP;
        if (agx_internal_upload_streams(det, (void **)up) != 0) return AGX_ERR_HIP;
    }
    struct Slot {  // one chunk's results while its tails run
        std::vector<agx_saddle> saddles;  // compact
        std::vector<uint32_t> ns, offs;
        std::vector<int> fst;
    };
    std::vector<Slot> slots((size_t)R);
    std::mutex m;
    std::condition_variable cv;
    std::vector<int> uploaded;   // per chunk: 0 pending, 1 on the device, -1 failed        (guarded by m)
    int outstanding[R] = {0, 0, 0, 0};  // tails of the slot's chunk not finished yet        (guarded by m)
    std::atomic<int> first_bad{AGX_OK};
    std::atomic<bool> nomem{false};
    int rc = AGX_OK;
    bool pending_batch = false;  // a chunk is enqueued on the detector and not fetched yet
    auto upload_task = [&, device](int ci) {
        const int c0 = ci * chunk, nf = std::min(chunk, n_frames - c0), slot = ci % S;
        const bool ok = hipSetDevice(device) == hipSuccess &&
                        hipMemcpyAsync(d_stage + (size_t)slot * chunk_bytes, (const uint8_t *)frames + (size_t)c0 * frame_stride_bytes,
                                       (size_t)nf * frame_stride_bytes, hipMemcpyHostToDevice, up[slot]) == hipSuccess &&
                        hipStreamSynchronize(up[slot]) == hipSuccess;
        {
            std::lock_guard<std::mutex> lk(m);
            uploaded[(size_t)ci] = ok ? 1 : -1;
        }
        cv.notify_all();
    };
    // Nothing unwinds through the C boundary: an allocation failure on this thread (slot vectors, a task's
    // std::function) ends the loop like any other error, after the worker tasks -- which hold pointers into
    // the slots and references to the locals above -- have finished.
    try {
    uploaded.assign((size_t)n_chunks, 0);
    if (!d_frames)
        for (int ci = 0; ci < std::min(S, n_chunks); ++ci) pool->submit([&upload_task, ci] { upload_task(ci); });  // (empty queue: in order)
    for (int ci = 0; ci < n_chunks; ++ci) {
        const int c0 = ci * chunk, nf = std::min(chunk, n_frames - c0), r = ci % R;
        const uint8_t *h_chunk = (const uint8_t *)frames + (size_t)c0 * frame_stride_bytes;
        const void *d_chunk;
        {
            std::unique_lock<std::mutex> lk(m);
            cv.wait(lk, [&] { return outstanding[r] == 0 && (d_frames || uploaded[(size_t)ci] != 0); });
            if (!d_frames && uploaded[(size_t)ci] < 0) { rc = AGX_ERR_HIP; break; }
        }
        if (d_frames) d_chunk = (const uint8_t *)d_frames + (size_t)c0 * frame_stride_bytes;
        else d_chunk = d_stage + (size_t)(ci % S) * chunk_bytes;
        rc = agx_saddles_batch_enqueue(det, d_chunk, nf, width, height, row_stride_bytes, frame_stride_bytes, format);
        if (rc) break;
        pending_batch = true;
        // detector.rs:507: u8 luma for the decode.  L8 frames are their own; L16 / RGB8 chunks are converted on
        // the device behind the chain (the frames are there) and come back with the saddles
        const uint8_t *h_luma = nullptr;
        if (format != AGX_L8) {
            rc = agx_internal_chunk_luma8(det, d_chunk, nf, width, height, row_stride_bytes, frame_stride_bytes, format, r, R,
                                          (size_t)chunk, &h_luma, nullptr);
            if (rc) break;
        }
        Slot &sl = slots[(size_t)r];
        sl.ns.resize((size_t)nf);
        sl.offs.resize((size_t)nf);
        sl.fst.resize((size_t)nf);
        const agx_saddle *records = nullptr;
        rc = agx_internal_fetch_compact(det, &records, sl.ns.data(), sl.offs.data(), sl.fst.data());  // waits for the device
        pending_batch = false;
        if (rc) break;
        // chain and luma have read the staging slot: the chunk S ahead may go up (ahead of every queued tail)
        if (!d_frames && ci + S < n_chunks) pool->submit_front([&upload_task, ci] { upload_task(ci + S); });
        size_t total = 0;
        for (int f = 0; f < nf; ++f)
            if (sl.fst[(size_t)f] == AGX_OK) total = std::max(total, (size_t)sl.offs[(size_t)f] + sl.ns[(size_t)f]);
        sl.saddles.assign(records, records + total);  // (the detector's mirror is overwritten by the next chunk)
        int n_tasks = 0;
        for (int f = 0; f < nf; ++f) n_tasks += sl.fst[(size_t)f] == AGX_OK;
        {
            std::lock_guard<std::mutex> lk(m);
            outstanding[r] = n_tasks;
        }
        int submitted = 0;
        try {
        for (int f = 0; f < nf; ++f) {
            const int gf = c0 + f;
            if (sl.fst[(size_t)f] != AGX_OK) {  // reported, never truncated
                counts[gf] = 0;
                if (frame_status) frame_status[gf] = sl.fst[(size_t)f];
                int exp = AGX_OK;
                first_bad.compare_exchange_strong(exp, sl.fst[(size_t)f]);
                continue;
            }
            const agx_saddle *sp = sl.saddles.data() + sl.offs[(size_t)f];
            const uint32_t n_s = sl.ns[(size_t)f];
            const uint8_t *img = h_chunk + (size_t)f * frame_stride_bytes;
            const uint8_t *dev_grey = h_luma ? h_luma + (size_t)f * (size_t)width * (size_t)height : nullptr;
            pool->submit([=, &first_bad, &nomem, &m, &cv, &outstanding] {
              try {  // nothing unwinds out of a worker thread: host memory exhaustion becomes the frame's status
                const uint8_t *g = dev_grey ? dev_grey : img;  // L8: the frame itself, read at its own pitch
                const size_t gstride = dev_grey ? (size_t)width : row_stride_bytes;
                const std::vector<agx_tag> &tags = detect_tail_scratch(*fam, max_boards, sp, n_s, g, width, height, gstride);
                int stf = AGX_OK;
                counts[gf] = (uint32_t)tags.size();
                if (tags.size() > cap_per_frame) {
                    stf = AGX_ERR_CAPACITY;
                    int exp = AGX_OK;
                    first_bad.compare_exchange_strong(exp, stf);
                } else if (!tags.empty()) {
                    std::memcpy(out + (size_t)gf * cap_per_frame, tags.data(), tags.size() * sizeof(agx_tag));
                }
                if (frame_status) frame_status[gf] = stf;
              } catch (...) {  // host memory exhausted inside this frame's search: the call fails as a whole
                counts[gf] = 0;
                if (frame_status) frame_status[gf] = AGX_ERR_NOMEM;
                nomem.store(true);
              }
              bool last;
              {
                  std::lock_guard<std::mutex> lk(m);
                  last = --outstanding[r] == 0;
              }
              if (last) cv.notify_all();
            });
            ++submitted;
        }
        } catch (...) {  // a task could not be queued: the slot's counter must not wait for tails that will never run
            std::lock_guard<std::mutex> lk(m);
            outstanding[r] -= n_tasks - submitted;
            throw;
        }
    }
    } catch (...) {
        rc = AGX_ERR_NOMEM;  // host memory exhausted on this thread
    }
    (void)pool->wait();  // every tail and every upload still queued (an upload may be reading the caller's frames)
    if (pending_batch) agx_internal_abandon_batch(det);  // an error between enqueue and fetch: no stale batch is left to be fetched later
    if (rc) return rc;
    if (nomem.load()) return AGX_ERR_NOMEM;
    return first_bad.load();
}
