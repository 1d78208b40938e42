// Dense linear at ~fp32 accuracy on the bf16 matrix cores (GGCN_PREC_BF16X3):
//     Y[M,F] = X[M,K] . W[K,F]          (models/gcn.py:34)
//
// Why: at hidden=768 the layer is GEMM-bound, not HBM-bound: 154.6
// GFLOP per layer against an f32-MFMA peak of ~155 TFLOP/s is ~1 ms, five times
// the HBM floor.  gfx950 has no xf32/TF32.  So every fp32 operand is split into
// two bf16 terms, x = hi + lo (hi = RNE bf16(x), lo = RNE bf16(x - hi), residual
// <= 2^-16 |x|), and each product is three bf16 MFMAs with fp32 accumulation:
//     x.w ~= hi.hi + lo.hi + hi.lo            (dropped lo.lo <= 2^-16 |x.w|)
// -> |err| ~ 1e-5 * sqrt(K) * rms|x.w| before the mean-aggregation, far inside the
// 1e-4 parity gate, at 1/3 of the bf16 rate (833 TFLOP/s ceiling instead of 155).
//
// Kernel shape, operand lane maps and the measured scheduling notes: bf16x3_core.h (the main
// loop is shared with fused_layer.hip).  This file adds the weight packer and the plain-store
// epilogue.
#include "bf16x3_core.h"

namespace ggcn {
namespace {

using namespace bx3;

// ---- W -> fragment-ordered bf16 hi/lo image --------------------------------------------
// block (n_tile, k_step), 128 threads: thread = (plane, lane)
// TR: pack the TRANSPOSE of the stored matrix (element (k, n) is read from W[n*ldw + k]): the
// backward pass multiplies by W^T (dX = dH . W^T) with the same kernels.
template <bool TR>
__global__ __launch_bounds__(128) void weight_pack_kernel(const float *__restrict__ W, int64_t ldw,
                                                          int K, int F, int k_steps,
                                                          bf16x8 *__restrict__ pack)
{
    const int n_tile = blockIdx.x, k_step = blockIdx.y;
    const int plane = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = n_tile * NT + (lane & 31);
    const int kb = k_step * KSTEP + 8 * (lane >> 5);
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = kb + j;
        const float w = (k < K && n < F) ? (TR ? W[(int64_t)n * ldw + k] : W[(int64_t)k * ldw + n]) : 0.0f;
        const __bf16 hi = (__bf16)w;
        v[j] = plane == 0 ? hi : (__bf16)(w - (float)hi);
    }
    pack[(((int64_t)n_tile * k_steps + k_step) * 2 + plane) * 64 + lane] = v;
}

__device__ __forceinline__ void store_elem(float *p, float v) { *p = v; }
__device__ __forceinline__ void store_elem(__half *p, float v) { *p = __float2half_rn(v); }

// ET: element type of X and Y (float, or __half with fp32 accumulation)
template <typename ET, bool AVEC, bool KFULL>
__global__ __launch_bounds__(kThreads, kWavesPerSimd) void linear_bf16x3_kernel(
    const ET *__restrict__ X, int64_t ldx, const char *__restrict__ wpack,
    ET *__restrict__ Y, int64_t ldy, int64_t M, int K, int F, int m_tiles, int n_wg, int k_steps)
{
    __shared__ __attribute__((aligned(16))) char lds[kLdsBytes];
    int m_tile, n_wgi;
    if (!tile_of_block(blockIdx.x, m_tiles, n_wg, m_tile, n_wgi)) return;  // before any barrier

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int64_t m0 = (int64_t)m_tile * BM;
    const int n_tiles_total = (F + NT - 1) / NT;
    const int nt0 = n_wgi * (BN / NT) + wn * RN;  // this wavefront's first 32-column tile

    // rows past M are clamped to row M-1: a row of A only feeds the same row of Y, never stored
    constexpr int NP = Geom<ET>::NP;
    const ET *arow[NP];
    bool avalid[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        int64_t gm = m0 + stage_row<ET>(i);
        gm = gm < M ? gm : M - 1;
        arow[i] = X + gm * ldx;
        avalid[i] = true;
    }
    f32x16 acc[4][RN];
    mainloop<ET, AVEC, KFULL, false>(arow, avalid, wpack, K, k_steps, wm, nt0, n_tiles_total, lds, acc);

#ifndef GGCN_LAB_NO_STORE
    const bool full_rows = m0 + BM <= M;  // workgroup-uniform: the row guard only exists in the last tile
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        const int gn = (nt0 + j) * NT + (lane & 31);
        if (gn >= F) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t gmb = m0 + wm * 128 + i * 32 + 4 * (lane >> 5);
            ET *yb = Y + gmb * ldy + gn;
            if (full_rows) {
#pragma unroll
                for (int r = 0; r < 16; ++r) store_elem(yb + (int64_t)((r & 3) + 8 * (r >> 2)) * ldy, acc[i][j][r]);
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (gmb + (r & 3) + 8 * (r >> 2) < M)
                        store_elem(yb + (int64_t)((r & 3) + 8 * (r >> 2)) * ldy, acc[i][j][r]);
            }
        }
    }
#else
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j) asm volatile("" ::"v"(acc[i][j]));
    if (M < 0) store_elem(Y, 0.f);
#endif
}

template <typename ET>
int launch_linear(const ET *X, int64_t ldx, const void *wpack, ET *Y, int64_t ldy, int64_t M, int K, int F,
                  hipStream_t st)
{
    if (!wpack) return fail(GGCN_EINVAL, "ggcn_linear(bf16x3): wpack is NULL (call ggcn_weight_pack first)");
    if (!aligned16(wpack)) return fail(GGCN_EINVAL, "ggcn_linear(bf16x3): wpack must be 16-byte aligned");
    constexpr int EPT = Geom<ET>::EPT;
    const bool avec = (K % EPT == 0) && (ldx % EPT == 0) && aligned16(X);
    const bool kfull = (K % BK == 0);
    const int k_steps = round_up(K, BK) / KSTEP;
    const int64_t m_tiles = (M + BM - 1) / BM;
    const int n_wg = (F + BN - 1) / BN;
    const int64_t grid = grid_for(m_tiles, n_wg);
    if (grid > (int64_t)INT32_MAX) return fail(GGCN_EUNSUPPORTED, "ggcn_linear(bf16x3): M too large");
    const char *wp = static_cast<const char *>(wpack);
#define GGCN_LAUNCH(AV, KF)                                                                                     \
    hipLaunchKernelGGL((linear_bf16x3_kernel<ET, AV, KF>), dim3((unsigned)grid), dim3(kThreads), 0, st, X, ldx, \
                       wp, Y, ldy, M, K, F, (int)m_tiles, n_wg, k_steps)
    if (avec && kfull) GGCN_LAUNCH(true, true);
    else if (avec) GGCN_LAUNCH(true, false);
    else GGCN_LAUNCH(false, false);
#undef GGCN_LAUNCH
    return check_launch("ggcn_linear(bf16x3)");
}

}  // namespace

size_t weight_pack_bytes(int K, int F)
{
    if (K <= 0 || F <= 0) return 0;
    const size_t k_steps = (size_t)bx3::round_up(K, bx3::BK) / bx3::KSTEP;  // whole 32-deep stages
    const size_t n_tiles = (size_t)bx3::round_up(F, bx3::NT) / bx3::NT;
    return n_tiles * k_steps * 2 * bx3::FRAG_BYTES;
}

int weight_pack(const float *W, int64_t ldw, int K, int F, void *wpack, bool transposed, hipStream_t st)
{
    if (!W || !wpack) return fail(GGCN_EINVAL, "ggcn_weight_pack: null pointer");
    if (K <= 0 || F <= 0 || ldw < (transposed ? K : F))
        return fail(GGCN_EINVAL, "ggcn_weight_pack: bad shape K=%d F=%d ldw=%lld", K, F, (long long)ldw);
    if (!aligned16(wpack)) return fail(GGCN_EINVAL, "ggcn_weight_pack: wpack must be 16-byte aligned");
    const int k_steps = round_up(K, BK) / KSTEP;
    const int n_tiles = round_up(F, NT) / NT;
    if (k_steps > 65535) return fail(GGCN_EUNSUPPORTED, "ggcn_weight_pack: K too large");
    if (transposed)
        hipLaunchKernelGGL(weight_pack_kernel<true>, dim3((unsigned)n_tiles, (unsigned)k_steps), dim3(128), 0, st, W,
                           ldw, K, F, k_steps, static_cast<bf16x8 *>(wpack));
    else
        hipLaunchKernelGGL(weight_pack_kernel<false>, dim3((unsigned)n_tiles, (unsigned)k_steps), dim3(128), 0, st, W,
                           ldw, K, F, k_steps, static_cast<bf16x8 *>(wpack));
    return check_launch("ggcn_weight_pack");
}

int linear_bf16x3(const float *X, int64_t ldx, const void *wpack, float *Y, int64_t ldy, int64_t M,
                  int K, int F, hipStream_t st)
{
    return launch_linear<float>(X, ldx, wpack, Y, ldy, M, K, F, st);
}

int linear_bf16x3_h(const void *X, int64_t ldx, const void *wpack, void *Y, int64_t ldy, int64_t M,
                    int K, int F, hipStream_t st)
{
    return launch_linear<__half>(static_cast<const __half *>(X), ldx, wpack, static_cast<__half *>(Y), ldy, M, K, F, st);
}

}  // namespace ggcn
