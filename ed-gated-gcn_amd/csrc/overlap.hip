// Gate-diversity regulariser: xy = mean_b sum_f x1[b,f] * y1[b,f]
// (models/bert_amir5.py:638, `(x1 * y1).sum(1).mean()`), deterministic:
// one workgroup per graph reduces its row in a fixed tree order into
// partial[b]; a second single-workgroup launch adds the B partials in a fixed
// order.  2*4*B*F bytes, HBM-bound and tiny next to the layers.
#include "common.h"

namespace ggcn {
namespace {

__device__ __forceinline__ float block_sum(float v, float *lds /*4 floats*/)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) lds[wave] = v;
    __syncthreads();
    const float s = (lds[0] + lds[1]) + (lds[2] + lds[3]);
    __syncthreads();
    return s;
}

__global__ __launch_bounds__(256) void overlap_rows(const float *__restrict__ x1,
                                                    const float *__restrict__ y1, int F,
                                                    float *__restrict__ partial)
{
    __shared__ float lds[4];
    const int64_t base = (int64_t)blockIdx.x * F;
    float s = 0.0f;
    for (int f = threadIdx.x; f < F; f += 256) s = fmaf(x1[base + f], y1[base + f], s);
    s = block_sum(s, lds);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void overlap_finish(const float *__restrict__ partial, int B,
                                                      float *__restrict__ xy)
{
    __shared__ float lds[4];
    float s = 0.0f;
    for (int b = threadIdx.x; b < B; b += 256) s += partial[b];
    s = block_sum(s, lds);
    if (threadIdx.x == 0) *xy = s / (float)B;
}

}  // namespace

size_t overlap_workspace_bytes(int B) { return (size_t)(B > 0 ? B : 1) * sizeof(float); }

int gate_overlap(const float *x1, const float *y1, int B, int F, float *xy, void *workspace,
                 hipStream_t st)
{
    if (!x1 || !y1 || !xy || !workspace) return fail(GGCN_EINVAL, "ggcn_gate_overlap: null pointer");
    if (B <= 0 || F <= 0) return fail(GGCN_EINVAL, "ggcn_gate_overlap: B=%d F=%d must be positive", B, F);
    float *partial = static_cast<float *>(workspace);
    hipLaunchKernelGGL(overlap_rows, dim3((unsigned)B), dim3(256), 0, st, x1, y1, F, partial);
    hipLaunchKernelGGL(overlap_finish, dim3(1), dim3(256), 0, st, partial, B, xy);
    return check_launch("ggcn_gate_overlap");
}

}  // namespace ggcn
