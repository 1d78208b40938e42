// Does v_mfma_f32_32x32x16_f16 keep fp16 subnormal inputs, and do v_cvt_pk_f16_f32 / v_fma_mixlo_f16 produce them?
// (the epilogue's second hidden plane is (v - fp16(v)) in fp16: subnormal for |v| < 0.25)   Development tool.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float *out)
{
    const int l = threadIdx.x;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.0f; b[i] = (_Float16)0.0f; }
    // A[row r][k=0] = 1.0 for lane half 0; B[k=0][col c] = subnormal 3 * 2^-24
    if (l < 32) { a[0] = (_Float16)1.0f; b[0] = __builtin_bit_cast(_Float16, (unsigned short)3); }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    // conversions
    const float v = 0.01f + l * 1e-4f;
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    const h2 p = __builtin_convertvector(f2{v, v}, h2);
    float r;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p), "v"(v));
    unsigned lo = 0;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "+v"(lo) : "v"(p), "v"(v));
    if (l == 0) {
        out[0] = c[0];
        out[1] = 3.0f / 16777216.0f;
        out[2] = r;
        out[3] = (float)__builtin_bit_cast(_Float16, (unsigned short)(lo & 0xffff));
        out[4] = (float)(_Float16)1e-6f;
    }
}
int main()
{
    float *d, h[5];
    hipMalloc(&d, 64);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, 20, hipMemcpyDeviceToHost);
    printf("mfma(1.0 x subnormal 3*2^-24) = %g (exact %g)\nresidual of 0.01 in f32 %g, via v_fma_mixlo_f16 %g;  (half)1e-6 = %g\n", h[0], h[1], h[2], h[3], h[4]);
    return 0;
}
