// detect_batch.cpp -- TagDetector::detect (reference src/detector.rs:505-540) over a batch of
// frames: the saddle chain of a chunk of frames runs on the device while a pool of host threads
// runs the board search + decode (detect's loop body, :510-539) of the previous chunk.  The host
// tail is the reference's exhaustive search (several milliseconds per frame and thread), so the
// end-to-end rate is set by the number of host threads; the pool keeps them busy and the device
// work disappears behind them.
#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
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

namespace agx {

// A fixed set of worker threads with a task queue; wait() blocks until every submitted task is done.
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
    void submit(std::function<void()> f)
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            q_.push_back(std::move(f));
            ++pending_;
        }
        cv_.notify_one();
    }
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
    if (n_threads <= 0) n_threads = (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 64u);
    WorkerPool *pool = static_cast<WorkerPool *>(agx_internal_pool(det, n_threads));
    if (!pool) return AGX_ERR_ARG;
    const FamilyInfo *fam = static_cast<const FamilyInfo *>(agx_internal_family(det));
    const int max_boards = agx_internal_max_boards(det);
    hipStream_t st = (hipStream_t)agx_internal_stream(det);
    if (hipSetDevice(agx_internal_device(det)) != hipSuccess) return AGX_ERR_HIP;

    // chunks: large enough that the chain runs at batch efficiency, small enough that the pool has
    // work while the next chunk is uploaded and processed
    const int chunk = std::max(1, std::min(n_frames, std::max(2 * pool->size(), 32)));
    const size_t chunk_bytes = (size_t)chunk * frame_stride_bytes;
    uint8_t *d_stage = nullptr;
    hipStream_t up = nullptr;
    hipEvent_t up_done[2] = {nullptr, nullptr}, stage_free[2] = {nullptr, nullptr};
    if (!d_frames) {
        d_stage = static_cast<uint8_t *>(agx_internal_stage(det, 2 * chunk_bytes));  // double-buffered upload
        if (!d_stage) return AGX_ERR_HIP;
        if (agx_internal_upload_stream(det, (void **)&up, (void **)up_done, (void **)stage_free) != 0) return AGX_ERR_HIP;
    }
    // chunk c + 1 goes up on the upload stream while chunk c's chain runs and its results are fetched (the two staging halves
    // alternate: a half is written again when the chain that read it -- two chunks back -- is through)
    auto upload = [&](int c0, int ci) -> bool {
        const int nf = std::min(chunk, n_frames - c0), par = ci & 1;
        if (ci >= 2 && hipStreamWaitEvent(up, stage_free[par], 0) != hipSuccess) return false;
        if (hipMemcpyAsync(d_stage + (size_t)par * chunk_bytes, (const uint8_t *)frames + (size_t)c0 * frame_stride_bytes,
                           (size_t)nf * frame_stride_bytes, hipMemcpyHostToDevice, up) != hipSuccess) return false;
        return hipEventRecord(up_done[par], up) == hipSuccess;
    };
    if (!d_frames && !upload(0, 0)) return AGX_ERR_HIP;
    std::vector<std::vector<agx_saddle>> saddles(2);  // per chunk parity: [chunk frames][cap]
    std::vector<std::vector<uint32_t>> ns(2);
    std::vector<std::vector<int>> fst(2);
    uint32_t cap_s = 16384;  // saddles per frame the staging holds; grown when a chunk has a longer list
    std::atomic<int> first_bad{AGX_OK};
    std::atomic<bool> nomem{false};
    int rc = AGX_OK;
    bool pending_batch = false;  // a chunk is enqueued on the detector and not fetched yet
    // Nothing unwinds through the C boundary: an allocation failure on this thread (staging vectors, a task's
    // std::function) ends the loop like any other error, after the worker tasks -- which hold pointers into
    // saddles[] and a reference to first_bad -- have finished.
    try {
    for (int c0 = 0, ci = 0; c0 < n_frames; c0 += chunk, ++ci) {
        const int nf = std::min(chunk, n_frames - c0);
        const int par = ci & 1;
        const uint8_t *h_chunk = (const uint8_t *)frames + (size_t)c0 * frame_stride_bytes;
        const void *d_chunk;
        if (d_frames) {
            d_chunk = (const uint8_t *)d_frames + (size_t)c0 * frame_stride_bytes;
        } else {
            if (c0 + chunk < n_frames && !upload(c0 + chunk, ci + 1)) { rc = AGX_ERR_HIP; break; }  // the next chunk, under this one's chain
            if (hipStreamWaitEvent(st, up_done[par], 0) != hipSuccess) { rc = AGX_ERR_HIP; break; }
            d_chunk = d_stage + (size_t)par * chunk_bytes;
        }
        rc = agx_saddles_batch_enqueue(det, d_chunk, nf, width, height, row_stride_bytes, frame_stride_bytes, format);
        if (rc) break;
        pending_batch = true;
        // detector.rs:507: u8 luma for the decode.  L8 frames are their own; L16 / RGB8 chunks are converted on
        // the device behind the chain (the frames are there) and come back with the saddles
        const uint8_t *h_luma = nullptr;
        if (format != AGX_L8) {
            if (ci >= 2) (void)pool->wait();  // tails of the chunk two back read this half of the luma staging
            rc = agx_internal_chunk_luma8(det, d_chunk, nf, width, height, row_stride_bytes, frame_stride_bytes, format, par,
                                          (size_t)chunk, &h_luma);
            if (rc) break;
        }
        if (!d_frames && hipEventRecord(stage_free[par], st) != hipSuccess) { rc = AGX_ERR_HIP; break; }  // chain + luma have read this half
        // the tails of the chunk two back read saddles[par]: they must be done before it is refilled
        if (ci >= 2) (void)pool->wait();
        saddles[par].resize((size_t)nf * cap_s);
        ns[par].assign(nf, 0);
        fst[par].assign(nf, 0);
        rc = agx_saddles_batch_fetch(det, saddles[par].data(), cap_s, ns[par].data(), fst[par].data());  // waits for the device
        pending_batch = false;
        if (rc == AGX_ERR_CAPACITY) {
            // a list longer than the staging (pure-noise frames of several megapixels; the reference's
            // Vec has no limit): the batch is still fetchable -- make room and fetch again
            uint32_t longest = 0;
            for (int f = 0; f < nf; ++f) longest = std::max(longest, ns[par][f]);
            if (longest > cap_s) {
                cap_s = longest;
                saddles[par].resize((size_t)nf * cap_s);
                rc = agx_saddles_batch_fetch(det, saddles[par].data(), cap_s, ns[par].data(), fst[par].data());
            }
        }
        if (rc && rc != AGX_ERR_CAPACITY) break;
        rc = AGX_OK;
        for (int f = 0; f < nf; ++f) {
            const int gf = c0 + f;
            if (fst[par][f] != AGX_OK) {  // reported, never truncated
                counts[gf] = 0;
                if (frame_status) frame_status[gf] = fst[par][f];
                int exp = AGX_OK;
                first_bad.compare_exchange_strong(exp, fst[par][f]);
                continue;
            }
            const agx_saddle *sp = saddles[par].data() + (size_t)f * cap_s;
            const uint32_t n_s = ns[par][f];
            const uint8_t *img = h_chunk + (size_t)f * frame_stride_bytes;
            const uint8_t *dev_grey = h_luma ? h_luma + (size_t)f * (size_t)width * (size_t)height : nullptr;
            pool->submit([=, &first_bad, &nomem] {
              try {  // nothing unwinds out of a worker thread: host memory exhaustion becomes the frame's status
                const uint8_t *g = dev_grey ? dev_grey : img;  // L8: the frame itself, read at its own pitch
                const size_t gstride = dev_grey ? (size_t)width : row_stride_bytes;
                std::vector<agx_tag> tags;
                detect_tail(*fam, max_boards, std::vector<agx_saddle>(sp, sp + n_s), g, width, height, gstride, tags);
                int stf = AGX_OK;
                if (tags.size() > cap_per_frame) {
                    stf = AGX_ERR_CAPACITY;
                    int exp = AGX_OK;
                    first_bad.compare_exchange_strong(exp, stf);
                    counts[gf] = (uint32_t)tags.size();
                } else {
                    counts[gf] = (uint32_t)tags.size();
                    if (!tags.empty()) std::memcpy(out + (size_t)gf * cap_per_frame, tags.data(), tags.size() * sizeof(agx_tag));
                }
                if (frame_status) frame_status[gf] = stf;
              } catch (...) {  // host memory exhausted inside this frame's search: the call fails as a whole
                counts[gf] = 0;
                if (frame_status) frame_status[gf] = AGX_ERR_NOMEM;
                nomem.store(true);
              }
            });
        }
    }
    } catch (...) {
        rc = AGX_ERR_NOMEM;  // host memory exhausted on this thread
    }
    (void)pool->wait();
    if (up) (void)hipStreamSynchronize(up);  // (after an error an upload may still be reading the caller's frames)
    if (pending_batch) agx_internal_abandon_batch(det);  // an error between enqueue and fetch: no stale batch is left to be fetched later
    if (rc) return rc;
    if (nomem.load()) return AGX_ERR_NOMEM;
    return first_bad.load();
}
