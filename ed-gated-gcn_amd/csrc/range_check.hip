// max |x| of a feature matrix: the on-demand range check of the opt-in f16mx8 arithmetic (f16mx8_core.h: fp16 +
// block-scaled fp8 operands need |x|, |w| < 65504, and keep their full accuracy for |x| <= 448).  One pass over
// the matrix, 16-byte loads, one atomic per workgroup.  out[0] = max |x| over the finite entries (as float),
// out[1] = 1.0f if any entry is NaN or infinite.  Not on the forward path: GraphConvolution.validate_range().
#include "common.h"

#include <hip/hip_fp16.h>

namespace ggcn {
namespace {

template <typename ET>
__global__ __launch_bounds__(256) void absmax_kernel(const ET *__restrict__ X, int64_t ld, int64_t M, int K,
                                                     float *__restrict__ out)
{
    __shared__ float red[4], bad[4];
    float m = 0.0f, nf = 0.0f;
    const int64_t total = M * (int64_t)K;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * 256) {
        const int64_t r = idx / K;
        const float v = fabsf((float)X[r * ld + (idx - r * K)]);
        if (v <= 3.4028234664e38f) m = fmaxf(m, v);   // false for NaN and inf
        else nf = 1.0f;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        m = fmaxf(m, __shfl_xor(m, d));
        nf = fmaxf(nf, __shfl_xor(nf, d));
    }
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = m; bad[threadIdx.x >> 6] = nf; }
    __syncthreads();
    if (threadIdx.x == 0) {
        // non-negative floats order like their bit patterns as signed ints
        atomicMax(reinterpret_cast<int *>(out), __float_as_int(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
        atomicMax(reinterpret_cast<int *>(out + 1), __float_as_int(fmaxf(fmaxf(bad[0], bad[1]), fmaxf(bad[2], bad[3]))));
    }
}

// Test hook (ggcn_debug_poison_lds): every CU's whole LDS (160 KiB) is filled with `pattern`, so that the NEXT kernel on
// the stream finds known garbage where the hardware would otherwise leave whatever the previous kernels stored.  A
// workgroup takes all of a CU's LDS (one per CU at a time) and stays for `hold` ticks of the 100 MHz clock, so the first
// 256 workgroups of the grid land on 256 different CUs.  tests/test_gpu_parity.py runs the LDS-heavy kernels behind
// different patterns and demands bit-identical results: a read of LDS the launch itself did not write cannot hide.
constexpr int kPoisonLds = 160 * 1024;
__global__ __launch_bounds__(1024) void lds_poison_kernel(uint32_t pattern, int hold, unsigned int *__restrict__ sink)
{
    extern __shared__ __attribute__((aligned(16))) uint32_t plds[];
    for (int i = threadIdx.x; i < kPoisonLds / 4; i += 1024) plds[i] = pattern;
    __syncthreads();
    const uint64_t t0 = wall_clock64();
    while ((int64_t)(wall_clock64() - t0) < hold) __builtin_amdgcn_s_sleep(16);
    if (plds[(threadIdx.x * 37u + blockIdx.x) % (kPoisonLds / 4)] != pattern) atomicAdd(sink, 1u);   // keeps the stores alive
}
static __device__ unsigned int g_poison_sink;

}  // namespace

int poison_lds(uint32_t pattern, hipStream_t st)
{
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        return fail(GGCN_ELAUNCH, "ggcn_debug_poison_lds: cannot read the device's CU count");
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(lds_poison_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, kPoisonLds) != hipSuccess)
        return fail(GGCN_ELAUNCH, "ggcn_debug_poison_lds: cannot reserve %d bytes of LDS", kPoisonLds);
    unsigned int *sink = nullptr;
    if (hipGetSymbolAddress(reinterpret_cast<void **>(&sink), HIP_SYMBOL(g_poison_sink)) != hipSuccess)
        return fail(GGCN_ELAUNCH, "ggcn_debug_poison_lds: no sink");
    hipLaunchKernelGGL(lds_poison_kernel, dim3((unsigned)(4 * cus)), dim3(1024), kPoisonLds, st, pattern, 500 /* 5 us */, sink);
    return check_launch("ggcn_debug_poison_lds");
}

int absmax(const void *X, int is_half, int64_t ld, int64_t M, int K, float *out, hipStream_t st)
{
    if (!X || !out) return fail(GGCN_EINVAL, "ggcn_absmax: null pointer");
    if (M <= 0 || K <= 0 || ld < K) return fail(GGCN_EINVAL, "ggcn_absmax: M=%lld K=%d ld=%lld", (long long)M, K, (long long)ld);
    hipError_t e = hipMemsetAsync(out, 0, 2 * sizeof(float), st);
    if (e != hipSuccess) return fail(GGCN_ELAUNCH, "ggcn_absmax: %s", hipGetErrorString(e));
    const int64_t total = M * (int64_t)K;
    const unsigned grid = (unsigned)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    if (is_half)
        hipLaunchKernelGGL(absmax_kernel<__half>, dim3(grid), dim3(256), 0, st, static_cast<const __half *>(X), ld, M, K, out);
    else
        hipLaunchKernelGGL(absmax_kernel<float>, dim3(grid), dim3(256), 0, st, static_cast<const float *>(X), ld, M, K, out);
    return check_launch("ggcn_absmax");
}

}  // namespace ggcn
