// checks the v_cmp -> s_and -> v_addc mask-accumulation idiom used by K1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(const float *d, uint32_t *out, uint32_t *ref, int thr_bits_in, uint64_t okm)
{
    const int lane = threadIdx.x;
    uint32_t mw = 0, r = 0;
    int thr_bits;
    asm("v_readfirstlane_b32 %0, %1\n\ts_nop 1" : "=s"(thr_bits) : "v"(__builtin_bit_cast(float, thr_bits_in) * 1.0f));
    const float thr = __builtin_bit_cast(float, thr_bits_in);
    for (int row = 0; row < 32; ++row) {
        const float v = d[row * 64 + lane];
        uint64_t cj;
        asm("v_cmp_lt_f32_e64 %[c0], %[d0], %[thr]\n\t"
            "s_and_b64 %[c0], %[c0], %[k0]\n\t"
            "v_addc_co_u32_e64 %[m0], vcc, %[m0], %[m0], %[c0]"
            : [m0] "+v"(mw), [c0] "=&s"(cj) : [d0] "v"(v), [thr] "s"(thr_bits), [k0] "s"(okm) : "vcc", "scc");
        const bool c = (v < thr) && ((okm >> lane) & 1);
        r |= c ? (1u << row) : 0u;
    }
    out[lane] = __brev(mw);
    ref[lane] = r;
}
int main()
{
    float h[32 * 64];
    for (int i = 0; i < 32 * 64; ++i) h[i] = -((i * 2654435761u) >> 8 & 0xffff) / 65536.0f;
    float *d; uint32_t *o, *r;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, 256); hipMalloc(&r, 256);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    const float thr = -0.5f;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, r, __builtin_bit_cast(int, thr), 0xfffffffffffffff0ull);
    uint32_t ho[64], hr[64];
    hipMemcpy(ho, o, 256, hipMemcpyDeviceToHost); hipMemcpy(hr, r, 256, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 64; ++i) if (ho[i] != hr[i]) { if (bad < 5) printf("lane %d got %08x want %08x\n", i, ho[i], hr[i]); ++bad; }
    printf("mismatching lanes: %d\n", bad);
    return bad != 0;
}
