// Microbenchmark: how many divergent 64-B requests does one CU keep in flight?  Every lane loads one dword from a
// random 64-B segment of a 1 GiB buffer (no reuse), U independent loads per lane per round, W waves per CU.
// rate x latency = requests in flight per CU.   Build: hipcc --offload-arch=gfx950 -O3 gather_mlp.hip -o gather_mlp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

template <int U, int LANES_PER_SEG>
__global__ void gather(const uint32_t *buf, uint32_t seg_mask, int rounds, uint32_t *out)
{
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t acc = 0, h = mix(gid / LANES_PER_SEG + 1u);
    for (int r = 0; r < rounds; ++r) {
        uint32_t v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            h = mix(h + 0x9e3779b9U);
            v[u] = buf[(size_t)(h & seg_mask) * 16 + (threadIdx.x % LANES_PER_SEG)];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    out[gid] = acc;
}

// one lane, dependent chain: latency of one miss
__global__ void chase(const uint32_t *buf, uint32_t seg_mask, int n, uint32_t *out, long long *cycles)
{
    uint32_t h = 12345u;
    const long long t0 = wall_clock64();
    for (int i = 0; i < n; ++i) h = mix(h + buf[(size_t)(h & seg_mask) * 16]);
    const long long t1 = wall_clock64();
    out[0] = h;
    cycles[0] = t1 - t0;
}

template <int U, int LPS>
void run(const uint32_t *buf, uint32_t seg_mask, uint32_t *out, int waves_per_cu)
{
    const int rounds = 2048 / U;
    dim3 block(64), grid(waves_per_cu ? 256 * waves_per_cu : 128);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    gather<U, LPS><<<grid, block>>>(buf, seg_mask, 4, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    gather<U, LPS><<<grid, block>>>(buf, seg_mask, rounds, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double reqs = (double)grid.x * (64 / LPS) * rounds * U;
    printf("lanes/segment %2d  loads in flight/lane %2d  waves/CU %2d: %.3f ms  %.1f G req/s chip  %.3f req/ns/CU  (%.0f GB/s of 64-B segments)\n",
           LPS, U, waves_per_cu, ms, reqs / ms / 1e6, reqs / ms / 1e6 / 256, reqs * 64 / ms / 1e6);
}

int main(int argc, char **)
{
    const size_t bytes = 1ull << 30;
    uint32_t *buf, *out; long long *cyc;
    hipMalloc(&buf, bytes); hipMemset(buf, 1, bytes);
    hipMalloc(&out, 256 * 32 * 64 * 4); hipMalloc(&cyc, 8);
    const uint32_t seg_mask = (uint32_t)(bytes / 64 - 1);
    chase<<<1, 1>>>(buf, seg_mask, 2000, out, cyc);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("dependent misses, idle chip: %.0f ns each\n", c * 10.0 / 2000);
    if (argc > 1) {  // table size sweep: where does the ceiling come from (DRAM / Infinity Cache / L2 / the CU's own path)?
        for (int mb : {2, 16, 64, 128, 192, 512, 1024}) {
            const uint32_t m = (uint32_t)((size_t)mb * (1 << 20) / 64 - 1);
            printf("table %4d MB: ", mb);
            run<4, 1>(buf, m, out, 8);
        }
        printf("half the workgroups (128 waves on the chip):\n");
        run<16, 1>(buf, seg_mask, out, 0);
        return 0;
    }
    for (int w : {1, 2, 4, 8, 16, 32}) { run<1, 1>(buf, seg_mask, out, w); run<4, 1>(buf, seg_mask, out, w); run<16, 1>(buf, seg_mask, out, w); }
    for (int w : {4, 16}) { run<4, 4>(buf, seg_mask, out, w); run<4, 16>(buf, seg_mask, out, w); run<16, 16>(buf, seg_mask, out, w); }
    return 0;
}
