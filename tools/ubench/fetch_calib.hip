// Calibration of rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths K1 uses:
// reads of a known byte count with 4 B/lane (K1's input pattern) and 16 B/lane, writes with
// 16 B/lane.  Buffers are 1 GiB (past the 256 MiB Infinity Cache).  Run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE   and   --pmc WRITE_SIZE
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void read4(const uint32_t *p, size_t n, uint32_t *out)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void read16(const uint4 *p, size_t n, uint32_t *out)
{
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void write16(uint4 *p, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}
// Random gathers, the sparse kernels' pattern (round 5): every lane touches its own 64-byte line(s) of a 1 GiB buffer in an order
// that is a bijection of the line index (no line is asked for twice by design, nothing is re-used from a cache).
//   gather16   one 16-byte load per lane at the start of a random line: 2^24 lanes -> 2^24 distinct lines = 1 GiB of 64-byte lines
//   gather36   nine consecutive dwords per lane (a row of the 9 x 9 refinement window) starting at dword (i % 16) of a random
//              line: offsets 8 .. 15 run into the next line, so a lane touches 1.5 lines on average -> 2^24 x 1.5 x 64 B = 1.5 GiB
//              of line requests for 2^24 x 36 B = 0.5625 GiB used
__global__ void gather16(const uint4 *p, uint32_t n_lines, uint32_t *out)
{
    uint32_t acc = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_lines; i += gridDim.x * blockDim.x) {
        const uint32_t line = (i * 2654435761u) & (n_lines - 1);  // odd multiplier: a bijection on 2^k lines
        const uint4 v = p[(size_t)line * 4];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}
__global__ void gather36(const uint32_t *p, uint32_t n_lines, uint32_t *out)
{
    uint32_t acc = 0;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_lines; i += gridDim.x * blockDim.x) {
        const uint32_t line = (i * 2654435761u) & (n_lines - 1);
        const size_t o = (size_t)(line == n_lines - 1 ? 0 : line) * 16 + (i & 15);  // (the last line has no successor)
#pragma unroll
        for (int k = 0; k < 9; ++k) acc ^= p[o + k];
    }
    if (acc == 0x12345678u) out[0] = acc;
}
int main()
{
    const size_t bytes = 1ull << 30;
    void *a, *b; uint32_t *o;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&o, 64);
    hipMemset(a, 1, bytes); hipMemset(b, 2, bytes);
    for (int rep = 0; rep < 3; ++rep) {
        read4<<<4096, 256>>>((const uint32_t *)a, bytes / 4, o);
        read16<<<4096, 256>>>((const uint4 *)b, bytes / 16, o);
        write16<<<4096, 256>>>((uint4 *)a, bytes / 16);
        gather16<<<4096, 256>>>((const uint4 *)b, (uint32_t)(bytes / 64), o);
        gather36<<<4096, 256>>>((const uint32_t *)b, (uint32_t)(bytes / 64), o);
    }
    hipDeviceSynchronize();
    printf("read4 / read16 / write16 / gather16 move %zu bytes (gather16: 2^24 distinct 64-byte lines for 16 bytes each); gather36 asks for 1.5 x that in lines\n", bytes);
    return 0;
}
