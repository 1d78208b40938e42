// Raw output of v_cvt_scalef32_2xpk16_fp6_f32 for distinct inputs (development tool): which 6-bit field holds which source.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const float *src, unsigned *dst, float scale)
{
    f32x16 a, b;
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = src[i]; b[i] = src[16 + i]; }
    u32x6 r;
    asm volatile("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(scale));
    if (threadIdx.x == 0)
        for (int i = 0; i < 6; ++i) dst[i] = r[i];
}
static float val(int code) { int e = (code >> 3) & 3, m = code & 7; return e == 0 ? m * 0.125f : (1.0f + m / 8.0f) * std::ldexp(1.0f, e - 1); }
int main()
{
    float h[32], *d; unsigned *o, r[6];
    hipMalloc(&d, 128); hipMalloc(&o, 24);
    for (int t = 0; t < 3; ++t) {
        for (int i = 0; i < 32; ++i) h[i] = t == 0 ? val(i) : t == 1 ? val(31 - i) : (i == 7 ? val(9) : i == 20 ? val(13) : 0.0f);
        hipMemcpy(d, h, 128, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 1.0f);
        hipMemcpy(r, o, 24, hipMemcpyDeviceToHost);
        printf("test %d raw:", t);
        for (int i = 0; i < 6; ++i) printf(" %08x", r[i]);
        printf("\n   fields:");
        for (int j = 0; j < 32; ++j) {
            int bit = 6 * j, w = bit >> 5, off = bit & 31;
            unsigned long long v = r[w]; if (w + 1 < 6) v |= (unsigned long long)r[w + 1] << 32;
            printf(" %d", (int)((v >> off) & 63));
        }
        printf("\n");
    }
    return 0;
}
