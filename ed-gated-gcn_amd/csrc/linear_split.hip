// Dense linear at ~fp32 accuracy on the low-precision matrix cores (GGCN_PREC_BF16X3 and
// GGCN_PREC_F16MX8; the second scheme is described in f16mx8_core.h):
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
// loop is shared with fused_layer.hip).  This file adds the weight packers of both schemes and the
// store epilogue (16-byte row stores through LDS for aligned fp32 output).
#include "f16_core.h"
#include "f16mx6_core.h"

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

// the same image for k-steps [ks0, ks0 + gridDim.y) of a longer k axis
__global__ __launch_bounds__(128) void weight_pack_slice_kernel(const float *__restrict__ W, int64_t ldw, int K, int F,
                                                                int k_steps, int ks0, bf16x8 *__restrict__ pack)
{
    const int n_tile = blockIdx.x, k_step = ks0 + blockIdx.y;
    const int plane = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int n = n_tile * NT + (lane & 31);
    const int kb = k_step * KSTEP + 8 * (lane >> 5);
    bf16x8 v;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = kb + j;
        const float w = (k < K && n < F) ? W[(int64_t)k * ldw + n] : 0.0f;
        const __bf16 hi = (__bf16)w;
        v[j] = plane == 0 ? hi : (__bf16)(w - (float)hi);
    }
    pack[(((int64_t)n_tile * k_steps + k_step) * 2 + plane) * 64 + lane] = v;
}

// ---- W -> f16mx8 image: per (32-column tile, 32-deep stage) [f16 frag k-step 0][k-step 1]
// [MX operand 64 x 32 B][scales 64 x 4 B]; see f16mx8_core.h.  block = 64 threads = one wavefront.
template <bool TR>
__global__ __launch_bounds__(64) void weight_pack_mx8_kernel(const float *__restrict__ W, int64_t ldw, int K, int F,
                                                            int stages, char *__restrict__ pack)
{
    const int n_tile = blockIdx.x, st = blockIdx.y, lane = threadIdx.x;
    const int c = lane & 31, h = lane >> 5;
    const int n = n_tile * NT + c;
    char *base = pack + ((int64_t)n_tile * stages + st) * mx8::STAGE_PACK_BYTES;
    auto wat = [&](int k) -> float {
        return (k < K && n < F) ? (TR ? W[(int64_t)n * ldw + k] : W[(int64_t)k * ldw + n]) : 0.0f;
    };
    // fp16 fragments: k-step s, lane (c, h): k = 32 st + 16 s + 8 h + j
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        mx8::f16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (_Float16)wat(32 * st + 16 * s2 + 8 * h + j);
        *reinterpret_cast<mx8::f16x8 *>(base + s2 * 1024 + lane * 16) = v;
    }
    // fp8 operand of the correction: block 0 = fp8(wh * 2^s0) is made in the main loop from the fp16
    // fragments; stored here is block 1 = fp8(wl * 2^s1), lane (c, h): bytes 0-7 <-> k = 32 st + 8h + j,
    // bytes 8-15 <-> k = 32 st + 16 + 8h + j (the k of the lane's two fp16 fragments); one power-of-two
    // scale per (column, block) over the 32 k of the stage
    float wh[16], wl[16], m0 = 0.0f, m1 = 0.0f;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        const float w = wat(32 * st + 16 * (jj >> 3) + 8 * h + (jj & 7));
        wh[jj] = (float)(_Float16)w;
        wl[jj] = w - wh[jj];
        m0 = fmaxf(m0, fabsf(wh[jj]));
        m1 = fmaxf(m1, fabsf(wl[jj]));
        if (fabsf(w) >= 65504.0f) atomicOr(&mx8::g_range_flag, 1u);   // a weight beyond fp16: the sticky range flag (f16mx8_core.h)
    }
    m0 = fmaxf(m0, __shfl_xor(m0, 32));
    m1 = fmaxf(m1, __shfl_xor(m1, 32));
    auto shift_for = [](float m) -> int {  // largest s with m * 2^s < 2^8 (e4m3 max is 448)
        if (!(m > 0.0f)) return 0;
        int e;
        (void)frexpf(m, &e);               // m = f * 2^e, f in [0.5, 1)
        int s = 8 - e;
        return s > 100 ? 100 : (s < -100 ? -100 : s);
    };
    const int s0 = shift_for(m0), s1 = shift_for(m1);
    int q[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        int b8 = 0;
        b8 = __builtin_amdgcn_cvt_pk_fp8_f32(ldexpf(wl[4 * t], s1), ldexpf(wl[4 * t + 1], s1), b8, false);
        b8 = __builtin_amdgcn_cvt_pk_fp8_f32(ldexpf(wl[4 * t + 2], s1), ldexpf(wl[4 * t + 3], s1), b8, true);
        q[t] = b8;
    }
    int *mxp = reinterpret_cast<int *>(base + mx8::kOffWl8 + lane * 16);
#pragma unroll
    for (int t = 0; t < 4; ++t) mxp[t] = q[t];
    if constexpr (GGCN_WH8_STORED) {   // block 0 = fp8(wh * 2^s0), same byte <-> k map (what the main loop used to convert per stage)
        int *whp = reinterpret_cast<int *>(base + mx8::kOffWh8 + lane * 16);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            int b8 = 0;
            b8 = __builtin_amdgcn_cvt_pk_fp8_f32(ldexpf(wh[4 * t], s0), ldexpf(wh[4 * t + 1], s0), b8, false);
            b8 = __builtin_amdgcn_cvt_pk_fp8_f32(ldexpf(wh[4 * t + 2], s0), ldexpf(wh[4 * t + 3], s0), b8, true);
            whp[t] = b8;
        }
    }
    // E8M0 (value = 2^(byte-127)): byte 0 = the scale of this lane's block for the MFMA (lane c: block 0,
    // lane c + 32: block 1), byte 1 = the scale of block 0 for the in-loop fp16 -> fp8 conversion
    *reinterpret_cast<int *>(base + mx8::kOffScales + lane * 4) = ((127 - s0) << 8) | (127 - (h ? s1 : s0));
}

// ---- W -> f16mx6 image (f16mx6_core.h): per (32-column tile, 32-deep stage) [f16 frag k-step 0][k-step 1]
// [64 lanes x {fp6 block 24 B, E8M0 scale dword, pad}]: lane (c, 0) holds fp6(wh / th) of column c's 32 k, lane (c, 1)
// fp6(wl / tl); field f <-> k = f (the order the A side's converts produce); th, tl: the power of two that
// puts the block's largest magnitude in (3.75, 7.5].  The hardware's own converter rounds (RNE, saturating).
template <bool TR>
__global__ __launch_bounds__(64) void weight_pack_mx6_kernel(const float *__restrict__ W, int64_t ldw, int K, int F,
                                                            int stages, char *__restrict__ pack)
{
    const int n_tile = blockIdx.x, st = blockIdx.y, lane = threadIdx.x;
    const int c = lane & 31, h = lane >> 5;
    const int n = n_tile * NT + c;
    char *base = pack + ((int64_t)n_tile * stages + st) * mx6::STAGE_PACK_BYTES;
    auto wat = [&](int k) -> float {
        return (k < K && n < F) ? (TR ? W[(int64_t)n * ldw + k] : W[(int64_t)k * ldw + n]) : 0.0f;
    };
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {   // fp16 fragments: k-step s, lane (c, h): k = 32 st + 16 s + 8 h + j
        mx8::f16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (_Float16)wat(32 * st + 16 * s2 + 8 * h + j);
        *reinterpret_cast<mx8::f16x8 *>(base + s2 * 1024 + lane * 16) = v;
    }
    float v[32], m = 0.0f;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        const float w = wat(32 * st + k);
        const float wh = (float)(_Float16)w;
        v[k] = h ? w - wh : wh;
        m = fmaxf(m, fabsf(v[k]));
    }
    int te = 0;
    if (m > 0.0f) {
        int e;
        const float f = frexpf(m, &e);          // m = f * 2^e, f in [0.5, 1)
        te = f * 8.0f <= 7.5f ? e - 3 : e - 2;   // m / 2^te in (3.75, 7.5]
        te = te < -126 ? -126 : (te > 127 ? 127 : te);
    }
    // v_cvt_scalef32_2xpk16_fp6_f32: field 2i <- a[i], field 2i + 1 <- b[i] (tools/probes/fp6_cvt_probe.hip)
    typedef float f32x16v __attribute__((ext_vector_type(16)));
    f32x16v a, b;
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = ldexpf(v[2 * i], -te); b[i] = ldexpf(v[2 * i + 1], -te); }
    mx6::u32x6 q;
    const float one = 1.0f;
    asm volatile("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=v"(q) : "v"(a), "v"(b), "v"(one));
    uint32_t *dst = reinterpret_cast<uint32_t *>(base + 2048 + lane * 32);
#pragma unroll
    for (int i = 0; i < 6; ++i) dst[i] = q[i];
    dst[6] = (uint32_t)(127 + te);
    dst[7] = 0u;
}

__device__ __forceinline__ void store_elem(float *p, float v) { *p = v; }
__device__ __forceinline__ void store_elem(__half *p, float v) { *p = __float2half_rn(v); }

// ET: element type of X and Y (float, or __half with fp32 accumulation); SCH: 0 = bf16x3, 1 = f16mx8
// VST (fp32 output, F and ldy multiples of 4, Y 16-byte aligned): 16-byte row stores through LDS
// XS (ggcn_linear_scaled, f16mx8 fast shapes only): *amax (device memory) = the largest |x| of the whole matrix, left there by the
// kernel that produced X; every workgroup derives the same power of two s from it (|x| * s < 256: inside fp16's range and the fp8
// correction's window whatever the magnitudes were -- gradients live far below fp16's normal range), multiplies x by s before
// the split and the accumulators by 1 / s at the store.  Powers of two: both multiplications are exact.
__device__ __forceinline__ void scale_from_amax(const float *amax, float &s, float &inv_s)
{
    const uint32_t bits = __builtin_amdgcn_readfirstlane(__float_as_uint(*amax));
    const int e = (int)((bits >> 23) & 0xFFu);          // biased exponent of amax (amax >= 0)
    int se = 261 - e;                                    // s = 2^(7 - (e - 127)): amax * s in [128, 256)
    se = se > 230 ? 230 : se;                            // (amax below 2^-96, zero included: a large finite scale)
    const bool ok = e != 255;                            // inf / NaN: no scaling (the result is inf / NaN either way)
    s = ok ? __uint_as_float((uint32_t)se << 23) : 1.0f;
    inv_s = ok ? __uint_as_float((uint32_t)(254 - se) << 23) : 1.0f;
}
template <int SCH, typename ET, bool AVEC, bool KFULL, bool VST, bool XS = false>
__global__ __launch_bounds__(kThreads, kWavesPerSimd) void linear_split_kernel(
    const ET *__restrict__ X, int64_t ldx, const char *__restrict__ wpack,
    ET *__restrict__ Y, int64_t ldy, int64_t M, int K, int F, int m_tiles, int n_wg, int k_steps, const float *__restrict__ amax_in = nullptr)
{
    static_assert(!XS || (SCH == 1 && AVEC && KFULL && VST && sizeof(ET) == 4), "the scaled form exists for the fast fp32 shapes of f16mx8");
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
    if constexpr (SCH == 0)
        bx3::mainloop<ET, AVEC, KFULL, false>(arow, avalid, wpack, K, k_steps, wm, nt0, n_tiles_total, lds, acc);
    else {
        constexpr bool BUF = AVEC && KFULL && sizeof(ET) == 4;   // buffer loads (f16mx8_core.h): offsets from the tile's first row
        mx8::BufX<ET> bx;
        if constexpr (BUF) {
            int rel[NP];
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int64_t gm = m0 + stage_row<ET>(i);
                rel[i] = (int)((gm < M ? gm : M - 1) - m0);
            }
            bx = mx8::make_bufx<ET>(X, ldx, m0, M, rel, tid);
        }
        if constexpr (XS) {
            float xs, inv_xs;
            scale_from_amax(amax_in, xs, inv_xs);
            mx8::mainloop<ET, AVEC, KFULL, false, false, BUF, 4, true>(arow, avalid, wpack, K, k_steps / 2, wm, nt0, n_tiles_total, lds, acc, 0, 4,
                                                                     nullptr, &bx, xs);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < RN; ++j)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] *= inv_xs;
        } else
        mx8::mainloop<ET, AVEC, KFULL, false, false, BUF>(arow, avalid, wpack, K, k_steps / 2, wm, nt0, n_tiles_total, lds, acc, 0, 4, nullptr, &bx);
    }

    const bool full_rows = m0 + BM <= M;  // workgroup-uniform: the row guard only exists in the last tile
    if constexpr (VST) {
        static_assert(!VST || RN == 2, "the staged epilogue is written for 64 columns per wavefront");
        // fp32 output as 16-byte row stores: each 32-row block of this wavefront's 64 columns is staged in
        // its 8 KiB of the (idle) A buffers and leaves as 4 rows x 256 contiguous bytes per instruction
        // (same scheme as the fused layer's epilogue: rows with bit 2 set swap their 32-column halves so
        // that the two lane halves hit different banks)
        float *stage_lds = reinterpret_cast<float *>(lds) + wave * (32 * 64);
        const int c = lane & 31, h = lane >> 5;
        const int colq = (lane & 15) * 4;
        const int gcol = nt0 * NT + colq;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < RN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row0 = (r & 3) + 8 * (r >> 2);
                    stage_lds[(row0 + 4 * h) * 64 + ((32 * j + c) ^ (32 * h))] = acc[i][j][r];
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int64_t gm0 = m0 + wm * 128 + i * 32;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = 4 * it + (lane >> 4);
                const float4 v4 = *reinterpret_cast<const float4 *>(&stage_lds[row * 64 + (colq ^ (32 * ((row >> 2) & 1)))]);
                if ((full_rows || gm0 + row < M) && gcol < F)
                    *reinterpret_cast<float4 *>(reinterpret_cast<float *>(Y) + (gm0 + row) * ldy + gcol) = v4;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        return;
    }
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
}

// fp16 features x fp16 image of W (f16_core.h): Y[M,F] (fp16) = X[M,K] (fp16) . W, fp32 accumulation
template <bool AVEC, bool KFULL>
__global__ __launch_bounds__(kThreads, kWavesPerSimd) void linear_f16_kernel(
    const __half *__restrict__ X, int64_t ldx, const char *__restrict__ wpack, __half *__restrict__ Y, int64_t ldy,
    int64_t M, int K, int F, int m_tiles, int n_wg, int records)
{
    __shared__ __attribute__((aligned(16))) char lds[kLdsBytes];
    int m_tile, n_wgi;
    if (!tile_of_block(blockIdx.x, m_tiles, n_wg, m_tile, n_wgi)) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave % WN;
    const int64_t m0 = (int64_t)m_tile * BM;
    const int n_tiles_total = (F + NT - 1) / NT;
    const int nt0 = n_wgi * (BN / NT) + wn * RN;
    const __half *arow[f16::NP16];
    bool avalid[f16::NP16];
#pragma unroll
    for (int i = 0; i < f16::NP16; ++i) {
        int64_t gm = m0 + 32 * i + (tid >> 3);
        gm = gm < M ? gm : M - 1;     // a row of A only feeds the same row of Y, never stored
        arow[i] = X + gm * ldx;
        avalid[i] = true;
    }
    f32x16 acc[4][RN];
    f16::mainloop<AVEC, KFULL, false>(arow, avalid, wpack, K, records, 0, nt0, n_tiles_total, lds, acc);
    // fp16 output: two rows x 32 columns of a tile per store instruction (64-byte row segments)
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        const int gn = (nt0 + j) * NT + (lane & 31);
        if (gn >= F) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t gmb = m0 + i * 32 + 4 * (lane >> 5);
            __half *yb = Y + gmb * ldy + gn;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (gmb + (r & 3) + 8 * (r >> 2) < M)
                    yb[(int64_t)((r & 3) + 8 * (r >> 2)) * ldy] = __float2half_rn(acc[i][j][r]);
        }
    }
}

int launch_linear_f16(const __half *X, int64_t ldx, const void *wpack, __half *Y, int64_t ldy, int64_t M, int K, int F,
                      hipStream_t st)
{
    if (!wpack) return fail(GGCN_EINVAL, "ggcn_linear_h(f16): wpack is NULL (ggcn_weight_pack with GGCN_PREC_F16MX8)");
    if (!aligned16(wpack)) return fail(GGCN_EINVAL, "ggcn_linear_h(f16): wpack must be 16-byte aligned");
    const bool avec = (K % 8 == 0) && (ldx % 8 == 0) && aligned16(X);
    const bool kfull = (K % f16::BK2 == 0);
    const int records = round_up(K, BK) / BK;
    const int64_t m_tiles = (M + BM - 1) / BM;
    const int n_wg = (F + BN - 1) / BN;
    const int64_t grid = grid_for(m_tiles, n_wg);
    if (grid > (int64_t)INT32_MAX) return fail(GGCN_EUNSUPPORTED, "ggcn_linear_h(f16): M too large");
    const char *wp = static_cast<const char *>(wpack);
#define GGCN_LAUNCH(AV, KF)                                                                                           \
    hipLaunchKernelGGL((linear_f16_kernel<AV, KF>), dim3((unsigned)grid), dim3(kThreads), 0, st, X, ldx, wp, Y, ldy, M, K, \
                       F, (int)m_tiles, n_wg, records)
    if (avec && kfull) GGCN_LAUNCH(true, true);
    else if (avec) GGCN_LAUNCH(true, false);
    else GGCN_LAUNCH(false, false);
#undef GGCN_LAUNCH
    return check_launch("ggcn_linear_h(f16)");
}

template <int SCH, typename ET>
int launch_linear(const ET *X, int64_t ldx, const void *wpack, ET *Y, int64_t ldy, int64_t M, int K, int F,
                  hipStream_t st)
{
    if (!wpack) return fail(GGCN_EINVAL, "ggcn_linear(bf16x3): wpack is NULL (call ggcn_weight_pack first)");
    if (!aligned16(wpack)) return fail(GGCN_EINVAL, "ggcn_linear(bf16x3): wpack must be 16-byte aligned");
    constexpr int EPT = Geom<ET>::EPT;
    // (the fast shapes address a tile's 128 rows with 32-bit byte offsets from its first row: buffer loads)
    const bool avec = (K % EPT == 0) && (ldx % EPT == 0) && aligned16(X) && (int64_t)ldx * (int64_t)sizeof(ET) * 257 < ((int64_t)1 << 31);
    const bool kfull = (K % BK == 0);
    const int k_steps = round_up(K, BK) / KSTEP;
    const int64_t m_tiles = (M + BM - 1) / BM;
    const int n_wg = (F + BN - 1) / BN;
    const int64_t grid = grid_for(m_tiles, n_wg);
    if (grid > (int64_t)INT32_MAX) return fail(GGCN_EUNSUPPORTED, "ggcn_linear(bf16x3): M too large");
    const char *wp = static_cast<const char *>(wpack);
    const bool vst = std::is_same<ET, float>::value && (F % 4 == 0) && (ldy % 4 == 0) && aligned16(Y);
#define GGCN_LAUNCH(AV, KF, VS)                                                                                     \
    hipLaunchKernelGGL((linear_split_kernel<SCH, ET, AV, KF, VS && std::is_same<ET, float>::value>), dim3((unsigned)grid), \
                       dim3(kThreads), 0, st, X, ldx, wp, Y, ldy, M, K, F, (int)m_tiles, n_wg, k_steps)
    if (avec && kfull && vst) GGCN_LAUNCH(true, true, true);
    else if (avec && kfull) GGCN_LAUNCH(true, true, false);
    else if (avec) GGCN_LAUNCH(true, false, false);
    else GGCN_LAUNCH(false, false, false);
#undef GGCN_LAUNCH
    return check_launch("ggcn_linear(bf16x3)");
}

// trailer of the f16mx8 / f16 / f16mx6 images: max over the columns of sum_k |w[k,f]|.  One workgroup per 64 columns, four
// k-groups of 64 threads each (coalesced over the columns of a row-major W, 8 loads in flight per thread), the groups'
// partial sums meet in LDS; non-negative floats order like their bit patterns.  (One thread per column over all k took
// 178 us for a 768 x 768 weight: a weight is packed on every training step.)
template <bool TR>
__global__ __launch_bounds__(256) void weight_colabs_kernel(const float *__restrict__ W, int64_t ldw, int K, int F,
                                                            unsigned int *__restrict__ trailer)
{
    __shared__ float part[4][64];
    const int c = threadIdx.x & 63, kg = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    float s = 0.0f;
    if (n < F) {
#pragma unroll 8
        for (int k = kg; k < K; k += 4) s += fabsf(TR ? W[(int64_t)n * ldw + k] : W[(int64_t)k * ldw + n]);
    }
    part[kg][c] = s;
    __syncthreads();
    if (kg == 0) {
        s = (part[0][c] + part[1][c]) + (part[2][c] + part[3][c]);
        if (!(s <= 3.0e38f)) s = __builtin_inff();   // NaN / overflow: no bound
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) s = fmaxf(s, __shfl_xor(s, d));
        if (c == 0) atomicMax(trailer, __float_as_uint(s));
    }
}

int pack_trailer(const float *W, int64_t ldw, int K, int F, bool transposed, char *trailer, hipStream_t st)
{
    if (hipMemsetAsync(trailer, 0, mx8::PACK_TRAILER_BYTES, st) != hipSuccess) return fail(GGCN_ELAUNCH, "ggcn_weight_pack: memset of the trailer failed");
    const dim3 grid((unsigned)((F + 63) / 64));
    if (transposed) hipLaunchKernelGGL(weight_colabs_kernel<true>, grid, dim3(256), 0, st, W, ldw, K, F, reinterpret_cast<unsigned int *>(trailer));
    else hipLaunchKernelGGL(weight_colabs_kernel<false>, grid, dim3(256), 0, st, W, ldw, K, F, reinterpret_cast<unsigned int *>(trailer));
    return check_launch("ggcn_weight_pack(trailer)");
}

}  // namespace

size_t weight_pack_bytes(int K, int F, int precision)
{
    if (K <= 0 || F <= 0) return 0;
    const size_t stages = (size_t)bx3::round_up(K, bx3::BK) / bx3::BK;  // whole 32-deep stages
    const size_t n_tiles = (size_t)bx3::round_up(F, bx3::NT) / bx3::NT;
    // (+ the trailer: max_f sum_k |w[k,f]|, the factor of the hidden-value bound of the one-launch layers, f16mx8_core.h)
    if (precision == GGCN_PREC_F16MX8 || precision == GGCN_PREC_F16) return n_tiles * stages * mx8::STAGE_PACK_BYTES + mx8::PACK_TRAILER_BYTES;
    if (precision == GGCN_PREC_F16MX6) return n_tiles * stages * mx6::STAGE_PACK_BYTES + mx8::PACK_TRAILER_BYTES;
    return n_tiles * stages * 2 * 2 * bx3::FRAG_BYTES;
}

int weight_pack(const float *W, int64_t ldw, int K, int F, int precision, bool transposed, void *wpack, hipStream_t st)
{
    if (!W || !wpack) return fail(GGCN_EINVAL, "ggcn_weight_pack: null pointer");
    if (K <= 0 || F <= 0 || ldw < (transposed ? K : F))
        return fail(GGCN_EINVAL, "ggcn_weight_pack: bad shape K=%d F=%d ldw=%lld", K, F, (long long)ldw);
    if (!aligned16(wpack)) return fail(GGCN_EINVAL, "ggcn_weight_pack: wpack must be 16-byte aligned");
    const int k_steps = round_up(K, BK) / KSTEP;
    const int n_tiles = round_up(F, NT) / NT;
    if (k_steps > 65535) return fail(GGCN_EUNSUPPORTED, "ggcn_weight_pack: K too large");
    if (precision == GGCN_PREC_F16MX8 || precision == GGCN_PREC_F16) {   // GGCN_PREC_F16 reads the fp16 part of the same image
        const dim3 grid((unsigned)n_tiles, (unsigned)(k_steps / 2));
        if (transposed)
            hipLaunchKernelGGL(weight_pack_mx8_kernel<true>, grid, dim3(64), 0, st, W, ldw, K, F, k_steps / 2,
                               static_cast<char *>(wpack));
        else
            hipLaunchKernelGGL(weight_pack_mx8_kernel<false>, grid, dim3(64), 0, st, W, ldw, K, F, k_steps / 2,
                               static_cast<char *>(wpack));
        const int rc = check_launch("ggcn_weight_pack(f16mx8)");
        return rc ? rc : pack_trailer(W, ldw, K, F, transposed, static_cast<char *>(wpack) + (size_t)n_tiles * (k_steps / 2) * mx8::STAGE_PACK_BYTES, st);
    }
    if (precision == GGCN_PREC_F16MX6) {
        const dim3 grid((unsigned)n_tiles, (unsigned)(k_steps / 2));
        if (transposed)
            hipLaunchKernelGGL(weight_pack_mx6_kernel<true>, grid, dim3(64), 0, st, W, ldw, K, F, k_steps / 2,
                               static_cast<char *>(wpack));
        else
            hipLaunchKernelGGL(weight_pack_mx6_kernel<false>, grid, dim3(64), 0, st, W, ldw, K, F, k_steps / 2,
                               static_cast<char *>(wpack));
        const int rc = check_launch("ggcn_weight_pack(f16mx6)");
        return rc ? rc : pack_trailer(W, ldw, K, F, transposed, static_cast<char *>(wpack) + (size_t)n_tiles * (k_steps / 2) * mx6::STAGE_PACK_BYTES, st);
    }
    if (precision != GGCN_PREC_BF16X3) return fail(GGCN_EINVAL, "ggcn_weight_pack: precision %d has no packed image", precision);
    if (transposed)
        hipLaunchKernelGGL(weight_pack_kernel<true>, dim3((unsigned)n_tiles, (unsigned)k_steps), dim3(128), 0, st, W,
                           ldw, K, F, k_steps, static_cast<bf16x8 *>(wpack));
    else
        hipLaunchKernelGGL(weight_pack_kernel<false>, dim3((unsigned)n_tiles, (unsigned)k_steps), dim3(128), 0, st, W,
                           ldw, K, F, k_steps, static_cast<bf16x8 *>(wpack));
    return check_launch("ggcn_weight_pack");
}

// bf16 hi/lo image of a [K_valid x F] matrix laid out for k_steps_total k-steps (rows past K_valid are zeros):
// the split-K weight gradient packs dH over a node axis padded to its chunk grid (dweight_bx3.hip)
int weight_pack_rows(const float *W, int64_t ldw, int64_t K_valid, int F, int k_steps_total, void *pack, hipStream_t st)
{
    const int n_tiles = round_up(F, NT) / NT;
    if (K_valid > (int64_t)INT32_MAX) return fail(GGCN_EUNSUPPORTED, "weight_pack_rows: too many rows");
    // grid.y is limited to 65535: walk the k-steps in slices
    for (int ks0 = 0; ks0 < k_steps_total; ks0 += 65535) {
        const int n = k_steps_total - ks0 < 65535 ? k_steps_total - ks0 : 65535;
        hipLaunchKernelGGL(weight_pack_slice_kernel, dim3((unsigned)n_tiles, (unsigned)n), dim3(128), 0, st, W, ldw,
                           (int)K_valid, F, k_steps_total, ks0, static_cast<bf16x8 *>(pack));
    }
    return check_launch("weight_pack_rows");
}

int linear_packed(const float *X, int64_t ldx, const void *wpack, float *Y, int64_t ldy, int64_t M,
                  int K, int F, int precision, hipStream_t st)
{
    if (precision == GGCN_PREC_F16MX8) return launch_linear<1, float>(X, ldx, wpack, Y, ldy, M, K, F, st);
    return launch_linear<0, float>(X, ldx, wpack, Y, ldy, M, K, F, st);
}

// dX of the backward (train.py:120 through gcn.py:34) on the two-unit f16mx8 product: rows scaled into range by a power of two
// derived on the device from *amax.  Fast shapes only (K % 32 == 0, F % 4 == 0, 16-byte aligned rows); the caller keeps bf16x3 otherwise.
bool linear_scaled_takes(const float *X, int64_t ldx, const float *Y, int64_t ldy, int K, int F)
{
    return (K % BK == 0) && (ldx % 4 == 0) && aligned16(X) && (int64_t)ldx * 4 * 257 < ((int64_t)1 << 31) && (F % 4 == 0) && (ldy % 4 == 0) && aligned16(Y);
}
int linear_scaled(const float *X, int64_t ldx, const void *wpack, float *Y, int64_t ldy, int64_t M, int K, int F, const float *amax,
                  hipStream_t st)
{
    if (!X || !wpack || !Y || !amax) return fail(GGCN_EINVAL, "ggcn_linear_scaled: null pointer");
    if (M <= 0 || K <= 0 || F <= 0 || ldx < K || ldy < F) return fail(GGCN_EINVAL, "ggcn_linear_scaled: M=%lld K=%d F=%d", (long long)M, K, F);
    if (!aligned16(wpack)) return fail(GGCN_EINVAL, "ggcn_linear_scaled: wpack must be 16-byte aligned");
    if (!linear_scaled_takes(X, ldx, Y, ldy, K, F))
        return fail(GGCN_EUNSUPPORTED, "ggcn_linear_scaled: needs K %% 32 == 0, F %% 4 == 0 and 16-byte aligned rows (use ggcn_linear with GGCN_PREC_BF16X3)");
    const int k_steps = round_up(K, BK) / KSTEP;
    const int64_t m_tiles = (M + BM - 1) / BM;
    const int n_wg = (F + BN - 1) / BN;
    const int64_t grid = grid_for(m_tiles, n_wg);
    if (grid > (int64_t)INT32_MAX) return fail(GGCN_EUNSUPPORTED, "ggcn_linear_scaled: M too large");
    hipLaunchKernelGGL((linear_split_kernel<1, float, true, true, true, true>), dim3((unsigned)grid), dim3(kThreads), 0, st, X, ldx,
                       static_cast<const char *>(wpack), Y, ldy, M, K, F, (int)m_tiles, n_wg, k_steps, amax);
    return check_launch("ggcn_linear_scaled");
}

int linear_packed_h(const void *X, int64_t ldx, const void *wpack, void *Y, int64_t ldy, int64_t M,
                    int K, int F, int precision, hipStream_t st)
{
    const __half *x = static_cast<const __half *>(X);
    __half *y = static_cast<__half *>(Y);
    if (precision == GGCN_PREC_F16) return launch_linear_f16(x, ldx, wpack, y, ldy, M, K, F, st);
    if (precision == GGCN_PREC_F16MX8) return launch_linear<1, __half>(x, ldx, wpack, y, ldy, M, K, F, st);
    return launch_linear<0, __half>(x, ldx, wpack, y, ldy, M, K, F, st);
}

GGCN_RANGE_FLAG_TU(range_flag_linear)

}  // namespace ggcn
