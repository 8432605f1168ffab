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
    }
    hipDeviceSynchronize();
    printf("each kernel moves %zu bytes\n", bytes);
    return 0;
}
