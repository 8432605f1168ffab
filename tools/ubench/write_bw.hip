// Microbenchmark: HBM write bandwidth of plain streaming float4 stores (1 GiB), of K1's store pattern
// (each wave writes 864 contiguous bytes per row, rows 5120 B apart, 128 rows per wave), and of a
// float4 copy -- the ceilings K1's blur-plane stores (82 % of its traffic) should be judged against.
// Build: hipcc --offload-arch=gfx950 -O3 -o write_bw write_bw.hip
#include <hip/hip_runtime.h>
#include <cstdio>

// a 16-byte store the compiler cannot split (its loop passes turn float4 stores into four dword stores)
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void store16(float *p, float v)
{
    const f32x4 q = {v, v + 1.0f, v + 2.0f, v + 3.0f};
    asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(q) : "memory");
}

__global__ void k_fill(float4 *dst, size_t n4, float v)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) dst[i] = make_float4(v, v + 1.0f, v + 2.0f, v + 3.0f);
}
__global__ void k_copy(const float4 *src, float4 *dst, size_t n4)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
// frames of 1280 x 800 f32; wave w owns (frame, strip of 216 columns, segment of 128 rows) like K1
__global__ void __launch_bounds__(256) k_strips(float *dst, int n_frames, float v)
{
    const int u = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int per_seg = 6 * n_frames;
    const int seg = u / per_seg, r = u - seg * per_seg, frame = r / 6, strip = r - frame * 6;
    if (seg >= 7) return;
    const int c0 = strip * 216 - 4 + 4 * lane;
    if (lane < 1 || c0 >= min(1280, strip * 216 + 216)) return;
    float *p = dst + (size_t)frame * 1280 * 800 + c0;
    const int y0 = seg * 128, y1 = min(800, y0 + 128);
    for (int y = y0; y < y1; ++y) store16(p + (size_t)y * 1280, v);
}

// the same waves, but the plane of a frame is laid out strip by strip: a wave's rows are contiguous
// (pitch = the strip's width); PAD: strip pitch rounded up to 256 floats (1 KB: line-aligned rows)
template <int PITCH>
__global__ void __launch_bounds__(256) k_strip_planar(float *dst, int n_frames, float v)
{
    const int u = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int per_seg = 6 * n_frames;
    const int seg = u / per_seg, r = u - seg * per_seg, frame = r / 6, strip = r - frame * 6;
    if (seg >= 7) return;
    const int c0 = -4 + 4 * lane, wcols = min(216, 1280 - strip * 216);
    if (lane < 1 || c0 >= wcols) return;
    float *p = dst + ((size_t)frame * 6 + strip) * (size_t)PITCH * 800 + c0;
    const int y0 = seg * 128, y1 = min(800, y0 + 128);
    for (int y = y0; y < y1; ++y) store16(p + (size_t)y * PITCH, v);
}
// row-major plane, strips of 224 columns (boundaries on 128-byte lines)
__global__ void __launch_bounds__(256) k_strips224(float *dst, int n_frames, float v)
{
    const int u = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int per_seg = 6 * n_frames;
    const int seg = u / per_seg, r = u - seg * per_seg, frame = r / 6, strip = r - frame * 6;
    if (seg >= 7) return;
    const int c0 = strip * 224 - 4 + 4 * lane;
    if (lane < 1 || c0 >= min(1280, strip * 224 + 224)) return;
    float *p = dst + (size_t)frame * 1280 * 800 + c0;
    const int y0 = seg * 128, y1 = min(800, y0 + 128);
    for (int y = y0; y < y1; ++y) store16(p + (size_t)y * 1280, v);
}

// generic row-major variant: STRIP columns per wave (all lanes below STRIP/4 store), SEG rows per wave
template <int STRIP, int SEG>
__global__ void __launch_bounds__(256) k_generic(float *dst, int n_frames, float v)
{
    constexpr int NS = (1280 + STRIP - 1) / STRIP, NSEG = (800 + SEG - 1) / SEG;
    const int u = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int per_seg = NS * n_frames;
    const int seg = u / per_seg, r = u - seg * per_seg, frame = r / NS, strip = r - frame * NS;
    if (seg >= NSEG) return;
    const int c0 = strip * STRIP + 4 * lane;
    if (c0 >= min(1280, strip * STRIP + STRIP)) return;
    float *p = dst + (size_t)frame * 1280 * 800 + c0;
    const int y0 = seg * SEG, y1 = min(800, y0 + SEG);
    for (int y = y0; y < y1; ++y) store16(p + (size_t)y * 1280, v);
}
template <int STRIP, int SEG>
void run_generic(const char *name, float *a, double bytes, hipEvent_t e0, hipEvent_t e1)
{
    constexpr int NS = (1280 + STRIP - 1) / STRIP, NSEG = (800 + SEG - 1) / SEG;
    float best = 1e9f;
    for (int i = 0; i < 6; ++i) {
        hipEventRecord(e0);
        k_generic<STRIP, SEG><<<(NS * NSEG * 256 + 3) / 4, 256>>>(a, 256, 3.0f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (i && ms < best) best = ms;
    }
    printf("%-28s %.3f ms  %.2f TB/s  (%d waves)\n", name, best, bytes / best / 1e9, NS * NSEG * 256);
}

int main()
{
    const size_t bytes = (size_t)256 * 1280 * 800 * 4;  // 1.05 GB = K1's blur planes
    float *a, *b;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time = [&](const char *name, double moved, auto launch) {
        float best = 1e9f;
        for (int i = 0; i < 6; ++i) {
            hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (i && ms < best) best = ms;
        }
        printf("%-28s %.3f ms  %.2f TB/s\n", name, best, moved / best / 1e9);
    };
    time("fill float4 (write only)", (double)bytes, [&] { k_fill<<<256 * 8, 256>>>((float4 *)a, bytes / 16, 1.0f); });
    time("copy float4 (read + write)", 2.0 * bytes, [&] { k_copy<<<256 * 8, 256>>>((const float4 *)a, (float4 *)b, bytes / 16); });
    time("K1 store pattern (write)", (double)bytes, [&] { k_strips<<<(6 * 7 * 256 + 3) / 4, 256>>>(a, 256, 2.0f); });
    time("strips of 224 columns", (double)bytes, [&] { k_strips224<<<(6 * 7 * 256 + 3) / 4, 256>>>(a, 256, 2.0f); });
    float *c; hipMalloc(&c, (size_t)256 * 6 * 256 * 800 * 4);
    time("strip-planar, pitch 216", (double)bytes, [&] { k_strip_planar<216><<<(6 * 7 * 256 + 3) / 4, 256>>>(c, 256, 2.0f); });
    time("strip-planar, pitch 256", (double)bytes, [&] { k_strip_planar<256><<<(6 * 7 * 256 + 3) / 4, 256>>>(c, 256, 2.0f); });
    run_generic<256, 128>("256 cols x 128 rows", a, (double)bytes, e0, e1);
    run_generic<256, 32>("256 cols x 32 rows", a, (double)bytes, e0, e1);
    run_generic<256, 800>("256 cols x 800 rows", a, (double)bytes, e0, e1);
    run_generic<128, 128>("128 cols x 128 rows", a, (double)bytes, e0, e1);
    run_generic<64, 128>("64 cols x 128 rows", a, (double)bytes, e0, e1);
    run_generic<256, 8>("256 cols x 8 rows", a, (double)bytes, e0, e1);
    run_generic<256, 16>("256 cols x 16 rows", a, (double)bytes, e0, e1);
    run_generic<256, 24>("256 cols x 24 rows", a, (double)bytes, e0, e1);
    run_generic<256, 64>("256 cols x 64 rows", a, (double)bytes, e0, e1);
    run_generic<216, 8>("216 cols x 8 rows", a, (double)bytes, e0, e1);
    run_generic<216, 16>("216 cols x 16 rows", a, (double)bytes, e0, e1);
    run_generic<256, 4>("256 cols x 4 rows", a, (double)bytes, e0, e1);
    hipMemset(a, 0, bytes);
    return 0;
}
