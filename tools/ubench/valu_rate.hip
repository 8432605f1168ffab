// Microbenchmark: VALU issue rate of scalar f32 mul+add vs packed (v_pk_mul_f32 / v_pk_add_f32)
// vs fma on gfx950, for 1..8 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float *out, int iters, float a, float b)
{
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
    if (MODE == 0) {  // scalar mul + add, 16 independent chains: 32 VALU per iteration
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = x[i] * a + b;
        }
    } else if (MODE == 1) {  // packed mul + add on 8 float2 chains: 16 VALU per iteration, same flops
        v2f p[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) p[i] = v2f{x[2 * i], x[2 * i + 1]};
        const v2f av{a, a}, bv{b, b};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) p[i] = p[i] * av + bv;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) { x[2 * i] = p[i].x; x[2 * i + 1] = p[i].y; }
    } else {  // fma, 16 chains: 16 VALU per iteration
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], a, b);
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, int waves_per_simd, float *d)
{
    const int iters = 4096;
    dim3 block(256), grid(256 * waves_per_simd);  // 256 CUs x (waves_per_simd blocks of 4 waves)
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, block>>>(d, 16, 1.0001f, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<grid, block>>>(d, iters, 1.0001f, 0.5f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flop_ops = (double)grid.x * 256 * iters * 16 * 2;  // mul+add per chain element
    const double insts_per_simd = (double)waves_per_simd * iters * (MODE == 0 ? 32 : 16);
    printf("%-10s waves/SIMD %d: %.3f ms, %.1f Tflop/s, %.2f cycles/VALU-inst/SIMD @2.4GHz\n", name, waves_per_simd, ms,
           flop_ops / ms / 1e9, ms * 1e-3 * 2.4e9 / insts_per_simd);
}

int main()
{
    float *d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float) * 2);
    for (int w : {1, 2, 4, 8}) { run<0>("mul+add", w, d); run<1>("pk mul+add", w, d); run<2>("fma", w, d); }
    return 0;
}
