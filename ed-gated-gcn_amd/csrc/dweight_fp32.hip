// Weight gradient of the layer:  dW[K,F] = X[N,K]^T . dH[N,F]   (backward of models/gcn.py:34
// under train.py:120), exact fp32 on the matrix cores, deterministic.
//
// Both operands have the reduction index n as their ROW index (a "TN" product).  With the
// f32-input MFMA v_mfma_f32_32x32x2_f32 every lane holds ONE element of A[i][k] (i = lane&31,
// k = lane>>5), so a fragment read is 32 consecutive floats of a row of X -- no transpose
// anywhere: rows of X and dH are copied into LDS as they lie and read as fragments directly.
// Split-K over n: grid.z slices of the node rows each write a partial [K,F] slab; a second
// kernel adds the slabs in a fixed order (no float atomics -> bitwise reproducible).
// Tile 128 (k) x 128 (f) x 16 (n) per stage, 4 wavefronts 2 x 2, 64 accumulator registers.
#include "common.h"

namespace ggcn {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
constexpr int TM = 128, TN = 128, TR = 16;   // output rows (k), output cols (f), reduction rows (n)
constexpr int LDS_LD = 128 + 4;

// VEC: rows of X and dH are 16-byte aligned (ld % 4 == 0, aligned base): 16-byte loads; otherwise element loads
template <bool VEC>
__global__ __launch_bounds__(256) void dweight_partial_kernel(
    const float *__restrict__ X, int64_t ldx, const float *__restrict__ G, int64_t ldg,
    int64_t N, int K, int F, int64_t rows_per_slice, float *__restrict__ slabs)
{
    __shared__ __attribute__((aligned(16))) float As[TR][LDS_LD];
    __shared__ __attribute__((aligned(16))) float Bs[TR][LDS_LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int k0 = blockIdx.y * TM, f0 = blockIdx.x * TN;
    const int64_t n_begin = (int64_t)blockIdx.z * rows_per_slice;
    const int64_t n_end = (n_begin + rows_per_slice < N) ? n_begin + rows_per_slice : N;

    // staging: 16 rows x 128 floats per operand = 512 float4; thread -> (row = tid/32 (+8), col4 = (tid%32)*4)
    const int s_r = tid >> 5, s_c = (tid & 31) * 4;
    float4 ra[2], rb[2];
    auto load = [&](int64_t n0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t n = n0 + s_r + h * 8;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
            if (n < n_end) {
                const float *px = X + n * ldx + k0 + s_c;
                const float *pg = G + n * ldg + f0 + s_c;
                if (VEC && k0 + s_c + 3 < K) a = *reinterpret_cast<const float4 *>(px);
                else { if (k0 + s_c < K) a.x = px[0]; if (k0 + s_c + 1 < K) a.y = px[1]; if (k0 + s_c + 2 < K) a.z = px[2];
                       if (!VEC && k0 + s_c + 3 < K) a.w = px[3]; }
                if (VEC && f0 + s_c + 3 < F) b = *reinterpret_cast<const float4 *>(pg);
                else { if (f0 + s_c < F) b.x = pg[0]; if (f0 + s_c + 1 < F) b.y = pg[1]; if (f0 + s_c + 2 < F) b.z = pg[2];
                       if (!VEC && f0 + s_c + 3 < F) b.w = pg[3]; }
            }
            ra[h] = a; rb[h] = b;
        }
    };
    auto store = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            *reinterpret_cast<float4 *>(&As[s_r + h * 8][s_c]) = ra[h];
            *reinterpret_cast<float4 *>(&Bs[s_r + h * 8][s_c]) = rb[h];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int fk = lane >> 5, fi = lane & 31;
    load(n_begin);
    for (int64_t n0 = n_begin; n0 < n_end; n0 += TR) {
        store();
        __syncthreads();
        if (n0 + TR < n_end) load(n0 + TR);
#pragma unroll
        for (int kk = 0; kk < TR; kk += 2) {
            const float a0 = As[kk + fk][wm * 64 + fi], a1 = As[kk + fk][wm * 64 + 32 + fi];
            const float b0 = Bs[kk + fk][wn * 64 + fi], b1 = Bs[kk + fk][wn * 64 + 32 + fi];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
    float *slab = slabs + (int64_t)blockIdx.z * K * F;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gf = f0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gk = k0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (gk < K && gf < F) slab[(int64_t)gk * F + gf] = acc[i][j][r];
            }
        }
}

__global__ __launch_bounds__(256) void dweight_reduce_kernel(const float *__restrict__ slabs, int n_slices,
                                                             int64_t kf, int K, int F, float *__restrict__ dW,
                                                             int64_t lddw)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= kf) return;
    float s = 0.0f;
    for (int z = 0; z < n_slices; ++z) s += slabs[(int64_t)z * kf + i];  // fixed order
    dW[(i / F) * lddw + (i % F)] = s;
}

int n_slices_for(int64_t N, int K, int F)
{
    const int64_t tiles = (int64_t)((K + TM - 1) / TM) * ((F + TN - 1) / TN);
    int64_t s = (2048 + tiles - 1) / tiles;          // ~8 workgroups per CU
    const int64_t max_s = (N + 255) / 256;           // at least 256 rows per slice
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    if (s > 65535) s = 65535;
    return (int)s;
}

}  // namespace

size_t dweight_workspace_bytes(int64_t N, int K, int F)
{
    if (N <= 0 || K <= 0 || F <= 0) return 0;
    return (size_t)n_slices_for(N, K, F) * K * F * sizeof(float);
}

int dweight(const float *X, int64_t ldx, const float *G, int64_t ldg, int64_t N, int K, int F, float *dW,
            int64_t lddw, void *workspace, hipStream_t st)
{
    if (!X || !G || !dW || !workspace) return fail(GGCN_EINVAL, "ggcn_dweight: null pointer");
    if (N <= 0 || K <= 0 || F <= 0) return fail(GGCN_EINVAL, "ggcn_dweight: N=%lld K=%d F=%d must be positive", (long long)N, K, F);
    if (ldx < K || ldg < F || lddw < F) return fail(GGCN_EINVAL, "ggcn_dweight: leading dimension too small");
    const bool vec = !((ldx % 4) || (ldg % 4) || !aligned16(X) || !aligned16(G));
    const int S = n_slices_for(N, K, F);
    const int64_t rows = ((N + S - 1) / S + TR - 1) / TR * TR;
    float *slabs = static_cast<float *>(workspace);
    dim3 grid((unsigned)((F + TN - 1) / TN), (unsigned)((K + TM - 1) / TM), (unsigned)S);
    if (vec) hipLaunchKernelGGL(dweight_partial_kernel<true>, grid, dim3(256), 0, st, X, ldx, G, ldg, N, K, F, rows, slabs);
    else hipLaunchKernelGGL(dweight_partial_kernel<false>, grid, dim3(256), 0, st, X, ldx, G, ldg, N, K, F, rows, slabs);
    const int64_t kf = (int64_t)K * F;
    hipLaunchKernelGGL(dweight_reduce_kernel, dim3((unsigned)((kf + 255) / 256)), dim3(256), 0, st, slabs, S, kf, K, F,
                       dW, lddw);
    return check_launch("ggcn_dweight");
}

}  // namespace ggcn
