// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 (MX block-scaled fp8) operand / scale layout.
// Development tool for the "fp16 main product + MX-fp8 correction products" idea.
#include <hip/hip_runtime.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// a, b: [64 lanes][8 dwords] raw operand registers; sa, sb: [64] scale dwords; y: [64][16]
extern "C" __global__ void mx_probe(const int* a, const int* b, const int* sa, const int* sb, float* y)
{
    const int l = threadIdx.x;
    i32x8 va, vb;
    for (int i = 0; i < 8; ++i) { va[i] = a[l * 8 + i]; vb[i] = b[l * 8 + i]; }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c, 0, 0, 0, sa[l], 0, sb[l]);
    for (int i = 0; i < 16; ++i) y[l * 16 + i] = c[i];
}

extern "C" int run_probe(const int* a, const int* b, const int* sa, const int* sb, float* y)
{
    hipLaunchKernelGGL(mx_probe, dim3(1), dim3(64), 0, 0, a, b, sa, sb, y);
    return (int)hipDeviceSynchronize();
}
