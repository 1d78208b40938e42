// Sub-word -> word pooling: Y[b, r, :] = sum_c A[b, r, c] * X[b, c, :]
// (models/bert_amir5.py:600 `torch.bmm(transform, x)`, the step right before the BiLSTM; the
// transform of data_utils.py:749-766 holds 1/l over the l sub-word positions of word r and zeros
// elsewhere, while x is 12 x 768 = 9216 features wide).  The dense product multiplies ~97 % zeros
// and reads every X row R times; here one workgroup per (b, r, 1024-feature slab) scans the dense
// row of A once, keeps its non-zeros (ascending c: a fixed summation order) in LDS and streams only
// the selected rows of X: HBM-bound, X is read once per word it belongs to.
// A is addressed with element strides, so the reference's non-contiguous slice
// `inputs['transform'][:, :T, :L]` is taken as it is, and the transposed product of the backward
// pass (dX = A^T . dY) is the same launch with the two strides swapped.
#include "common.h"

namespace ggcn {
namespace {

constexpr int kMaxCols = 2048;  // columns of A per row held in LDS as (index, value) pairs
constexpr int kSlab = 1024;     // features per workgroup: 256 threads x 4

template <bool VEC>
__global__ __launch_bounds__(256) void subword_pool_kernel(const float *__restrict__ A, int64_t sa_b, int64_t sa_r,
                                                          int64_t sa_c, const float *__restrict__ X,
                                                          int64_t x_batch, int64_t ldx, float *__restrict__ Y,
                                                          int64_t y_batch, int64_t ldy, int R, int C, int D)
{
    __shared__ int s_idx[kMaxCols];
    __shared__ float s_val[kMaxCols];
    __shared__ int s_wave[4];
    __shared__ int s_total;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x / R, r = blockIdx.x % R;
    const float *arow = A + b * sa_b + r * sa_r;
    if (tid == 0) s_total = 0;
    __syncthreads();
    // ---- compact the non-zeros of row (b, r), ascending column, 256 columns per round ----
    for (int c0 = 0; c0 < C; c0 += 256) {
        const int c = c0 + tid;
        const float v = c < C ? arow[c * sa_c] : 0.0f;
        const bool nz = v != 0.0f;
        const unsigned long long m = __ballot(nz);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) s_wave[wave] = __popcll(m);
        __syncthreads();
        int off = s_total;
        for (int w = 0; w < wave; ++w) off += s_wave[w];
        if (nz) {
            s_idx[off + before] = c;
            s_val[off + before] = v;
        }
        __syncthreads();
        if (tid == 0) s_total += s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
        __syncthreads();
    }
    const int n = s_total;
    const float *xb = X + b * x_batch;
    float *yrow = Y + b * y_batch + (int64_t)r * ldy;
    const int f = blockIdx.y * kSlab + tid * 4;
    if (VEC) {
        if (f >= D) return;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        int e = 0;
        for (; e + 1 < n; e += 2) {  // two rows in flight
            const float4 x0 = *reinterpret_cast<const float4 *>(xb + (int64_t)s_idx[e] * ldx + f);
            const float4 x1 = *reinterpret_cast<const float4 *>(xb + (int64_t)s_idx[e + 1] * ldx + f);
            const float v0 = s_val[e], v1 = s_val[e + 1];
            acc.x = fmaf(v0, x0.x, acc.x); acc.y = fmaf(v0, x0.y, acc.y);
            acc.z = fmaf(v0, x0.z, acc.z); acc.w = fmaf(v0, x0.w, acc.w);
            acc.x = fmaf(v1, x1.x, acc.x); acc.y = fmaf(v1, x1.y, acc.y);
            acc.z = fmaf(v1, x1.z, acc.z); acc.w = fmaf(v1, x1.w, acc.w);
        }
        if (e < n) {
            const float4 x0 = *reinterpret_cast<const float4 *>(xb + (int64_t)s_idx[e] * ldx + f);
            const float v0 = s_val[e];
            acc.x = fmaf(v0, x0.x, acc.x); acc.y = fmaf(v0, x0.y, acc.y);
            acc.z = fmaf(v0, x0.z, acc.z); acc.w = fmaf(v0, x0.w, acc.w);
        }
        *reinterpret_cast<float4 *>(yrow + f) = acc;
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int fq = blockIdx.y * kSlab + q * 256 + tid;  // coalesced scalar form
            if (fq >= D) continue;
            float acc = 0.0f;
            for (int e = 0; e < n; ++e) acc = fmaf(s_val[e], xb[(int64_t)s_idx[e] * ldx + fq], acc);
            yrow[fq] = acc;
        }
    }
}

}  // namespace

int subword_pool(const float *A, int64_t sa_b, int64_t sa_r, int64_t sa_c, const float *X, int64_t x_batch,
                 int64_t ldx, float *Y, int64_t y_batch, int64_t ldy, int B, int R, int C, int D, hipStream_t st)
{
    if (!A || !X || !Y) return fail(GGCN_EINVAL, "ggcn_subword_pool: null pointer");
    if (B <= 0 || R <= 0 || C <= 0 || D <= 0)
        return fail(GGCN_EINVAL, "ggcn_subword_pool: B=%d R=%d C=%d D=%d must be positive", B, R, C, D);
    if (C > kMaxCols) return fail(GGCN_EUNSUPPORTED, "ggcn_subword_pool: C=%d > %d columns", C, kMaxCols);
    if (ldx < D || ldy < D) return fail(GGCN_EINVAL, "ggcn_subword_pool: leading dimension too small");
    if ((int64_t)B * R > (int64_t)INT32_MAX) return fail(GGCN_EUNSUPPORTED, "ggcn_subword_pool: too many rows");
    const bool vec = (D % 4 == 0) && (ldx % 4 == 0) && (ldy % 4 == 0) && (x_batch % 4 == 0) && (y_batch % 4 == 0) &&
                     aligned16(X) && aligned16(Y);
    const dim3 grid((unsigned)(B * R), (unsigned)((D + kSlab - 1) / kSlab));
    if (vec)
        hipLaunchKernelGGL(subword_pool_kernel<true>, grid, dim3(256), 0, st, A, sa_b, sa_r, sa_c, X, x_batch, ldx, Y,
                           y_batch, ldy, R, C, D);
    else
        hipLaunchKernelGGL(subword_pool_kernel<false>, grid, dim3(256), 0, st, A, sa_b, sa_r, sa_c, X, x_batch, ldx, Y,
                           y_batch, ldy, R, C, D);
    return check_launch("ggcn_subword_pool");
}

}  // namespace ggcn
