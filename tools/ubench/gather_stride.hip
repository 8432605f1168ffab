// Microbenchmark: 9-row window gathers from a 1 GiB plane (what rochade_refine does: 9 rows x 48
// bytes per cluster) as a function of the distance between the rows.  Question: does a layout that
// puts a window's rows closer together (tiles, column blocks, per-strip planes) fetch faster, i.e.
// is the gather bound by DRAM row activations / channel spread, or just by the number of sectors?
// Build: hipcc --offload-arch=gfx950 -O3 -o gather_stride gather_stride.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void __launch_bounds__(64) k(const float *__restrict__ plane, size_t plane_floats, uint32_t stride_floats, float *out, uint32_t seed)
{
    const uint32_t gid = blockIdx.x * 64 + threadIdx.x;
    uint64_t h = (uint64_t)(gid + 1) * 0x9E3779B97F4A7C15ull + seed;
    h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
    const size_t span = (size_t)stride_floats * 9 + 16;
    size_t base = (size_t)(h % (plane_floats - span)) & ~(size_t)3;  // 16-byte aligned window start
    float s = 0.0f;
#pragma unroll
    for (int r = 0; r < 9; ++r) {
        const float4 *p = reinterpret_cast<const float4 *>(plane + base + (size_t)r * stride_floats);
        const float4 a = p[0], b = p[1], c = p[2];
        s += a.x + b.y + c.z;
    }
    out[gid] = s;
}

int main()
{
    const size_t plane_floats = (size_t)256 << 20;  // 1 GiB
    float *plane, *out;
    hipMalloc(&plane, plane_floats * 4);
    hipMemset(plane, 0, plane_floats * 4);
    const int waves = 4096;  // 262144 windows, like 256 frames x 1000 clusters
    hipMalloc(&out, (size_t)waves * 64 * 4);
    float *junk;  // 512 MiB written between runs so that nothing of the plane is cache resident
    hipMalloc(&junk, (size_t)512 << 20);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const uint32_t strides[] = {12, 32, 64, 128, 256, 512, 1280, 3840, 16384};
    for (uint32_t st : strides) {
        float best = 1e9f;
        for (int rep = 0; rep < 4; ++rep) {
            hipMemset(junk, rep, (size_t)512 << 20);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            k<<<waves, 64>>>(plane, plane_floats, st, out, 12345u + rep);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (ms < best) best = ms;
        }
        // the same windows again at once (same seed, nothing written in between): how much of the
        // touched sectors is still in L2 / MALL (Infinity Cache)?
        hipMemset(junk, 7, (size_t)512 << 20);
        k<<<waves, 64>>>(plane, plane_floats, st, out, 777u);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<<<waves, 64>>>(plane, plane_floats, st, out, 777u);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float warm; hipEventElapsedTime(&warm, e0, e1);
        printf("row distance %6u floats (%6u B): %.1f us for %d windows; repeated at once: %.1f us\n", st, st * 4, best * 1e3f, waves * 64, warm * 1e3f);
    }
    return 0;
}
