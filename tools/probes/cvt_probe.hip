// Probe: semantics of v_cvt_scalef32_pk_fp8_f32 (does the scale multiply or divide?) and of
// v_fma_mix_f32 with an fp16 half as the first factor.  Built by cvt_probe.py.
#include <hip/hip_runtime.h>
typedef short v2i16 __attribute__((ext_vector_type(2)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <bool OVFL>
__global__ void cvt_kernel(const float *x, int n, float scale, int *q_scaled, int *q_plain, float *resid)
{
    // MODE.FP16_OVFL (bit 23): overflowing fp16 / fp8 conversions clamp to the largest finite value
    if (OVFL) __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n) return;
    const float a = x[2 * i], b = x[2 * i + 1];
    v2i16 old = {0, 0};
    const v2i16 q = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, a, b, scale, false);
    q_scaled[i] = (int)(unsigned short)q[0];
    q_plain[i] = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false) & 0xffff;
    const h2 hh = __builtin_convertvector(f2{a, b}, h2);
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(hh), "v"(a));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(hh), "v"(b));
    resid[2 * i] = r0;
    resid[2 * i + 1] = r1;
    if (OVFL) resid[2 * i] = (float)hh[0], resid[2 * i + 1] = (float)hh[1];  // the fp16 conversion itself
}

extern "C" int run_cvt(const float *x, int n, float scale, int *q_scaled, int *q_plain, float *resid, int ovfl)
{
    if (ovfl)
        hipLaunchKernelGGL(cvt_kernel<true>, dim3((n / 2 + 63) / 64), dim3(64), 0, 0, x, n, scale, q_scaled, q_plain, resid);
    else
    hipLaunchKernelGGL(cvt_kernel<false>, dim3((n / 2 + 63) / 64), dim3(64), 0, 0, x, n, scale, q_scaled, q_plain, resid);
    return (int)hipDeviceSynchronize();
}
