// Microbenchmark: random 64-byte lines fetched 16 bytes per lane -- (A) every lane fetches the four pieces of ITS OWN line in four
// instructions (64 different lines per instruction: what a lane-per-cluster window fetch does), (B) four neighbouring lanes fetch
// the four pieces of ONE line in one instruction (16 lines per instruction).  Same lines, same bytes, same instruction count;
// 1 GiB buffer (no reuse), 16 waves per CU in 1024-thread workgroups, U lines per lane and round in flight.
// Build: hipcc --offload-arch=gfx950 -O3 line_coop.hip -o line_coop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
template <bool COOP, int U>
__global__ void __launch_bounds__(1024) k(const u32x4 *buf, uint32_t line_mask, int rounds, uint32_t *out)
{
    const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x, lane = threadIdx.x & 63u;
    uint32_t acc = 0;
    for (int r = 0; r < rounds; ++r) {
        u32x4 v[4 * U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                // A: lane's own line u, piece p.   B: line (u, p) of the quad's lanes = line index 4*(lane>>2)+p of this round, piece lane&3
                const uint32_t owner = COOP ? (gid & ~3u) + (uint32_t)p : gid;
                const uint32_t line = mix(owner * 0x9e3779b9U + (uint32_t)(r * U + u) * 0x85ebca6bU) & line_mask;
                v[4 * u + p] = buf[(size_t)line * 4 + (COOP ? (lane & 3u) : (uint32_t)p)];
            }
        }
#pragma unroll
        for (int i = 0; i < 4 * U; ++i) acc += v[i].x ^ v[i].w;
    }
    out[gid] = acc;
}
template <bool COOP, int U>
void run(const u32x4 *buf, uint32_t line_mask, uint32_t *out, const char *name)
{
    const int rounds = 256 / U;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<COOP, U><<<256, 1024>>>(buf, line_mask, 2, out);
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int i = 0; i < 3; ++i) {
        hipEventRecord(e0);
        k<COOP, U><<<256, 1024>>>(buf, line_mask, rounds, out);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double lines = 256.0 * 1024 * rounds * U;
    printf("%-34s lines in flight per lane %d: %.3f ms  %.1f G lines/s chip  %.2f lines/ns/CU  %.0f GB/s\n", name, U, best, lines / best / 1e6,
           lines / best / 1e6 / 256, lines * 64 / best / 1e6);
}
int main()
{
    const size_t bytes = (size_t)1 << 30;
    u32x4 *buf; uint32_t *out;
    hipMalloc(&buf, bytes); hipMalloc(&out, 256 * 1024 * 4);
    hipMemset(buf, 1, bytes);
    const uint32_t line_mask = (uint32_t)(bytes / 64 - 1);
    run<false, 2>(buf, line_mask, out, "own line, 4 instructions per line");
    run<true, 2>(buf, line_mask, out, "4 lanes per line, 1 instruction");
    run<false, 4>(buf, line_mask, out, "own line, 4 instructions per line");
    run<true, 4>(buf, line_mask, out, "4 lanes per line, 1 instruction");
    run<false, 7>(buf, line_mask, out, "own line, 4 instructions per line");
    run<true, 7>(buf, line_mask, out, "4 lanes per line, 1 instruction");
    // a 32 MB table (the Infinity Cache holds it): the rate without HBM behind it
    const uint32_t small_mask = (uint32_t)((32u << 20) / 64 - 1);
    run<false, 4>(buf, small_mask, out, "32 MB table: own line");
    run<true, 4>(buf, small_mask, out, "32 MB table: 4 lanes per line");
    // tables that the CU's own L1 holds (16 KB) or the XCD's L2 holds (2 MB): the rate of the L1's look-ups / of L2 hits
    const uint32_t l1_mask = (uint32_t)((16u << 10) / 64 - 1), l2_mask = (uint32_t)((2u << 20) / 64 - 1);
    run<false, 4>(buf, l1_mask, out, "16 KB table: own line");
    run<true, 4>(buf, l1_mask, out, "16 KB table: 4 lanes per line");
    run<false, 4>(buf, l2_mask, out, "2 MB table: own line");
    run<true, 4>(buf, l2_mask, out, "2 MB table: 4 lanes per line");
    return 0;
}
