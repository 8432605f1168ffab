// Microbenchmark: HBM write bandwidth of K1's store pattern with cached and with non-temporal stores, for strips
// of 216 / 224 / 256 columns and for a strip-planar plane (a wave's rows contiguous), 96-row segments.  Each
// figure is the time per launch of four launches back to back (a single launch of cached stores leaves up to
// 256 MB of its writes in the Infinity Cache).   Build: hipcc --offload-arch=gfx950 -O3 -o write_nt write_nt.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ void store16(float *p, float v)
{
    const f32x4 q = {v, v + 1.0f, v + 2.0f, v + 3.0f};
    if (NT) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(q) : "memory");
    else asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(q) : "memory");
}
// row-major plane 1280 x 800 per frame; wave = (frame, strip of STRIP columns, segment of SEG rows), segment-major
template <int STRIP, int SEG, bool NT, bool PLANAR>
__global__ void __launch_bounds__(256) k_rows(float *dst, int n_frames, float v)
{
    constexpr int NS = (1280 + STRIP - 1) / STRIP, NSEG = (800 + SEG - 1) / SEG;
    const int u = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int per_seg = NS * n_frames;
    const int seg = u / per_seg, r = u - seg * per_seg, frame = r / NS, strip = r - frame * NS;
    if (seg >= NSEG) return;
    const int wcols = min(STRIP, 1280 - strip * STRIP);
    if (4 * lane >= wcols) return;
    const int y0 = seg * SEG, y1 = min(800, y0 + SEG);
    if (PLANAR) {  // the strip's rows are contiguous: pitch = STRIP
        float *p = dst + ((size_t)frame * NS + strip) * (size_t)STRIP * 800 + 4 * lane;
        for (int y = y0; y < y1; ++y) store16<NT>(p + (size_t)y * STRIP, v);
    } else {
        float *p = dst + (size_t)frame * 1280 * 800 + strip * STRIP + 4 * lane;
        for (int y = y0; y < y1; ++y) store16<NT>(p + (size_t)y * 1280, v);
    }
}
template <int STRIP, int SEG, bool NT, bool PLANAR>
void run(const char *name, float *a, double bytes, hipEvent_t e0, hipEvent_t e1)
{
    constexpr int NS = (1280 + STRIP - 1) / STRIP, NSEG = (800 + SEG - 1) / SEG;
    const int grid = (NS * NSEG * 256 + 3) / 4;
    float best = 1e9f;
    for (int i = 0; i < 5; ++i) {
        hipEventRecord(e0);
        for (int k = 0; k < 4; ++k) k_rows<STRIP, SEG, NT, PLANAR><<<grid, 256>>>(a, 256, 3.0f + k);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        ms /= 4;
        if (i && ms < best) best = ms;
    }
    printf("%-44s %.3f ms  %.2f TB/s\n", name, best, bytes / best / 1e9);
}
int main()
{
    const size_t bytes = (size_t)256 * 1280 * 800 * 4;
    float *a; hipMalloc(&a, bytes + (size_t)256 * 6 * 32 * 800 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    run<216, 96, false, false>("216 columns, cached", a, (double)bytes, e0, e1);
    run<216, 96, true, false>("216 columns, nt", a, (double)bytes, e0, e1);
    run<224, 96, false, false>("224 columns, cached", a, (double)bytes, e0, e1);
    run<224, 96, true, false>("224 columns, nt", a, (double)bytes, e0, e1);
    run<256, 96, false, false>("256 columns, cached", a, (double)bytes, e0, e1);
    run<256, 96, true, false>("256 columns, nt", a, (double)bytes, e0, e1);
    run<224, 96, false, true>("224 columns strip-planar, cached", a, (double)bytes, e0, e1);
    run<224, 96, true, true>("224 columns strip-planar, nt", a, (double)bytes, e0, e1);
    run<256, 96, true, true>("256 columns strip-planar, nt", a, (double)bytes, e0, e1);
    run<224, 32, true, false>("224 columns x 32 rows, nt", a, (double)bytes, e0, e1);
    run<224, 800, true, false>("224 columns x 800 rows, nt", a, (double)bytes, e0, e1);
    run<128, 96, true, false>("128 columns, nt", a, (double)bytes, e0, e1);
    run<240, 96, true, false>("240 columns, nt", a, (double)bytes, e0, e1);
    run<240, 96, false, false>("240 columns, cached", a, (double)bytes, e0, e1);
    run<232, 96, true, false>("232 columns, nt", a, (double)bytes, e0, e1);
    run<248, 96, true, false>("248 columns, nt", a, (double)bytes, e0, e1);
    run<224, 96, true, false>("224 columns, nt (again)", a, (double)bytes, e0, e1);
    return 0;
}
