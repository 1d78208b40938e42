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

}  // namespace

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
