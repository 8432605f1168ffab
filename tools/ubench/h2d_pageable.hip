// What uploading a chunk of PAGEABLE host memory costs the calling thread (round 5, agx_detect_batch's upload tasks):
//   (a) hipMemcpyAsync from pageable memory + stream sync (the runtime stages through pinned buffers on the calling thread),
//   (b) hipHostRegister + hipMemcpyAsync + sync + hipHostUnregister (the DMA engine reads the caller's pages in place),
//   (c) memcpy into a pinned staging buffer of our own + hipMemcpyAsync + sync.
// Chunks of 16 MB (16 L8 frames) and 48 MB (16 RGB8 frames); wall time and CPU time of the calling thread.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static double cpu() { timespec t; clock_gettime(CLOCK_THREAD_CPUTIME_ID, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
int main()
{
    hipStream_t st; (void)hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (size_t mb : {16, 48}) {
        const size_t bytes = mb << 20;
        const int reps = 20;
        char *src = (char *)malloc(bytes * reps);  // a different chunk per repetition, as a batch's chunks are
        memset(src, 1, bytes * reps);
        void *dst, *pin; (void)hipMalloc(&dst, bytes); (void)hipHostMalloc(&pin, bytes, hipHostMallocDefault);
        for (int mode = 0; mode < 3; ++mode) {
            double w = 0, c = 0;
            for (int pass = 0; pass < 2; ++pass) {  // first pass warms up
                const double w0 = now(), c0 = cpu();
                for (int r = 0; r < reps; ++r) {
                    char *p = src + (size_t)r * bytes;
                    if (mode == 0) { (void)hipMemcpyAsync(dst, p, bytes, hipMemcpyHostToDevice, st); (void)hipStreamSynchronize(st); }
                    else if (mode == 1) {
                        if (hipHostRegister(p, bytes, hipHostRegisterDefault) != hipSuccess) { printf("register failed\n"); return 1; }
                        (void)hipMemcpyAsync(dst, p, bytes, hipMemcpyHostToDevice, st); (void)hipStreamSynchronize(st);
                        (void)hipHostUnregister(p);
                    } else { memcpy(pin, p, bytes); (void)hipMemcpyAsync(dst, pin, bytes, hipMemcpyHostToDevice, st); (void)hipStreamSynchronize(st); }
                }
                w = (now() - w0) / reps; c = (cpu() - c0) / reps;
            }
            const char *names[3] = {"pageable hipMemcpyAsync", "register + copy + unregister", "own pinned staging (memcpy + DMA)"};
            printf("%2zu MB  %-36s wall %.3f ms (%.1f GB/s)   CPU of the calling thread %.3f ms\n", mb, names[mode], w * 1e3, bytes / w * 1e-9, c * 1e3);
        }
        free(src); (void)hipFree(dst); (void)hipHostFree(pin);
    }
    return 0;
}
