// Dense linear in EXACT fp32 on the matrix cores:  Y[M,F] = X[M,K] . W[K,F]
// (models/gcn.py:34, `hidden = torch.matmul(text, self.weight)`; W is in x out).
//
// gfx950 has an f32-input MFMA, v_mfma_f32_32x32x2_f32: bit-for-bit a k-ordered
// fp32 fmaf chain (one rounding per product), at the f32 vector rate
// (~155 TFLOP/s chip-wide).  This is the GGCN_PREC_FP32 mode: the tightest
// parity with the reference's fp32 matmul; GGCN_PREC_BF16X3 (linear_split.hip)
// is the fast mode.
//
// Tiling: workgroup 256 threads = 4 wavefronts (2 x 2), tile 128 x 128 x 16;
// each wavefront owns 64 x 64 = 2 x 2 MFMA tiles of 32 x 32 (64 accumulator
// registers).  A is stored k-major in LDS ([k][m], stride 130 floats: the
// transposing store is conflict-free and the fragment read is 32 consecutive
// floats per half-wave); B is stored as it lies ([k][n]).  Operand lane maps
// (cdna guide §3): A: lane l holds A[i = l&31][k = l>>5]; B: B[k = l>>5][j = l&31];
// C/D: col = l&31, row = (r&3) + 8*(r>>2) + 4*(l>>5).
// Next tile's global loads are issued before the current tile's MFMAs.
#include "common.h"

namespace ggcn {
namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int BM = 128, BN = 128, BK = 16;
constexpr int LDA = BM + 2;  // 4*LDA % 32 == 8: the 4 k-quads of a half-wave hit 4 disjoint bank octets
constexpr int LDB = BN + 4;

template <bool AVEC, bool BVEC>
__global__ __launch_bounds__(256) void linear_fp32_kernel(
    const float *__restrict__ X, int64_t ldx, const float *__restrict__ W, int64_t ldw,
    float *__restrict__ Y, int64_t ldy, int64_t M, int K, int F)
{
    __shared__ float As[BK][LDA];
    __shared__ __attribute__((aligned(16))) float Bs[BK][LDB];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int64_t m0 = (int64_t)blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;

    // staging roles
    const int a_row = tid >> 2;          // 0..63 (+64)
    const int a_k = (tid & 3) * 4;       // 0,4,8,12
    const int b_k = tid >> 5;            // 0..7 (+8)
    const int b_n = (tid & 31) * 4;      // 0..124

    float4 ra[2], rb[2];

    auto load_tile = [&](int k0) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int64_t gm = m0 + a_row + h * 64;
            const int gk = k0 + a_k;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gm < M) {
                const float *p = X + gm * ldx + gk;
                if (AVEC) {
                    if (gk < K) v = *reinterpret_cast<const float4 *>(p);
                } else {
                    if (gk + 0 < K) v.x = p[0];
                    if (gk + 1 < K) v.y = p[1];
                    if (gk + 2 < K) v.z = p[2];
                    if (gk + 3 < K) v.w = p[3];
                }
            }
            ra[h] = v;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int gk = k0 + b_k + h * 8;
            const int gn = n0 + b_n;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gk < K) {
                const float *p = W + (int64_t)gk * ldw + gn;
                if (BVEC) {
                    if (gn < F) v = *reinterpret_cast<const float4 *>(p);
                } else {
                    if (gn + 0 < F) v.x = p[0];
                    if (gn + 1 < F) v.y = p[1];
                    if (gn + 2 < F) v.z = p[2];
                    if (gn + 3 < F) v.w = p[3];
                }
            }
            rb[h] = v;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int r = a_row + h * 64;
            As[a_k + 0][r] = ra[h].x;
            As[a_k + 1][r] = ra[h].y;
            As[a_k + 2][r] = ra[h].z;
            As[a_k + 3][r] = ra[h].w;
            *reinterpret_cast<float4 *>(&Bs[b_k + h * 8][b_n]) = rb[h];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int frag_k = lane >> 5;
    const int frag_i = lane & 31;

    load_tile(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        store_tile();
        __syncthreads();
        if (k0 + BK < K) load_tile(k0 + BK);  // in flight under the MFMAs below
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float a0 = As[kk + frag_k][wm * 64 + frag_i];
            const float a1 = As[kk + frag_k][wm * 64 + 32 + frag_i];
            const float b0 = Bs[kk + frag_k][wn * 64 + frag_i];
            const float b1 = Bs[kk + frag_k][wn * 64 + 32 + frag_i];
            acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int gn = n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t gm = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (gm < M && gn < F) Y[gm * ldy + gn] = acc[i][j][r];
            }
        }
}

}  // namespace

int linear_fp32(const float *X, int64_t ldx, const float *W, int64_t ldw, float *Y, int64_t ldy,
                int64_t M, int K, int F, hipStream_t st)
{
    const bool avec = (K % 4 == 0) && (ldx % 4 == 0) && aligned16(X);
    const bool bvec = (F % 4 == 0) && (ldw % 4 == 0) && aligned16(W);
    const int64_t by = (M + BM - 1) / BM;
    if (by > 65535) {
        // grid.y limit: split the rows into chunks of 65535 tiles
        const int64_t rows_per = (int64_t)65535 * BM;
        for (int64_t r = 0; r < M; r += rows_per) {
            const int64_t m = (M - r < rows_per) ? (M - r) : rows_per;
            int rc = linear_fp32(X + r * ldx, ldx, W, ldw, Y + r * ldy, ldy, m, K, F, st);
            if (rc) return rc;
        }
        return GGCN_OK;
    }
    dim3 grid((unsigned)((F + BN - 1) / BN), (unsigned)by);
    if (avec && bvec)
        hipLaunchKernelGGL((linear_fp32_kernel<true, true>), grid, dim3(256), 0, st, X, ldx, W, ldw, Y, ldy, M, K, F);
    else if (avec)
        hipLaunchKernelGGL((linear_fp32_kernel<true, false>), grid, dim3(256), 0, st, X, ldx, W, ldw, Y, ldy, M, K, F);
    else if (bvec)
        hipLaunchKernelGGL((linear_fp32_kernel<false, true>), grid, dim3(256), 0, st, X, ldx, W, ldw, Y, ldy, M, K, F);
    else
        hipLaunchKernelGGL((linear_fp32_kernel<false, false>), grid, dim3(256), 0, st, X, ldx, W, ldw, Y, ldy, M, K, F);
    return check_launch("ggcn_linear(fp32)");
}

}  // namespace ggcn
