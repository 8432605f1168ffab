// Checks the raw-buffer range rule K1 relies on (gfx950): with stride 0, an access is out of range
// -- store dropped, load returns 0 -- when voffset >= num_records - soffset, i.e. the SGPR offset
// takes part in the check.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float *buf, int n_bytes, unsigned *ld)
{
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)buf, 0, n_bytes, 0x00020000);
    const u32x4 v = {0x11111111u, 0x22222222u, 0x33333333u, 0x44444444u};
    // in range: row 1 of a 2-row buffer
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, threadIdx.x * 16, n_bytes / 2, 0);
    // soffset == num_records: every lane out of range
    __builtin_amdgcn_raw_buffer_store_b128(v, rs, threadIdx.x * 16, n_bytes, 0);
    ld[threadIdx.x] = __builtin_amdgcn_raw_buffer_load_b32(rs, threadIdx.x * 4, n_bytes, 0);
}
int main()
{
    const int row = 64 * 16, n = 2 * row;  // bytes
    float *d; unsigned *ld;
    if (hipMalloc(&d, 4 * row) != hipSuccess || hipMalloc(&ld, 256) != hipSuccess) return 2;  // 2 rows + 2 guard rows
    (void)hipMemset(d, 0, 4 * row);
    (void)hipMemset(ld, 0xff, 256);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, n, ld);
    static unsigned h[4 * 64 * 4], hl[64];
    if (hipMemcpy(h, d, 4 * row, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    (void)hipMemcpy(hl, ld, 256, hipMemcpyDeviceToHost);
    int in_range = 0, stray = 0, ld_nonzero = 0;
    for (int i = 0; i < 64 * 4; ++i) {
        in_range += h[64 * 4 + i] != 0;
        stray += (h[i] != 0) + (h[2 * 64 * 4 + i] != 0) + (h[3 * 64 * 4 + i] != 0);
    }
    for (int i = 0; i < 64; ++i) ld_nonzero += hl[i] != 0;
    printf("in-range dwords written %d/256, stray dwords %d, out-of-range loads non-zero %d\n", in_range, stray, ld_nonzero);
    return (in_range == 256 && stray == 0 && ld_nonzero == 0) ? 0 : 1;
}
