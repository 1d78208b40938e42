// Weight gradient on the bf16 matrix cores:  dW[K,F] = X[N,K]^T . dH[N,F]  (backward of
// models/gcn.py:34 under train.py:120), the bf16x3 counterpart of dweight_fp32.hip.
//
// The reduction runs over the N node rows, which is the ROW index of both operands.  Instead of
// a new "TN" main loop, the product is brought into the shape the forward linear already solves:
//     dW = Xt . dH,   Xt = X^T  [K x N]  (one tiled transpose pass, fp32),
//                     dH as the packed "weight" image of ggcn_weight_pack (hi/lo bf16 fragments)
// and the shared bf16x3 main loop (bf16x3_core.h) runs on row tiles of Xt with the node axis
// split into S chunks (split-K): every (chunk, row tile, column tile) workgroup writes a partial
// [128 x 256] block of a [S][K][F] slab array and a last kernel adds the slabs in a fixed order
// (no float atomics: bitwise reproducible).  All workgroups of one chunk get block ids of one
// residue mod 8, i.e. one XCD, so a chunk's rows of Xt and its fragments of dH are fetched from
// HBM once and shared through that XCD's L2.
// Measured on config 2 (N = 131 072, K = F = 768): see DESIGN.md 4.6.
#include "bf16x3_core.h"
#include "common.h"

#include <cstdlib>

namespace ggcn {
namespace {

using namespace bx3;

// Xt, chunk by chunk: out[s][c][r - s*chunk] = in[r][c] for the rows r of chunk s (zeros for R <= r < Rpad),
// i.e. every chunk is its own compact [C x chunk] matrix (a chunk's rows stay within ~20 KB of each
// other instead of 4*Rpad bytes apart).  64 x 64 tiles through LDS; chunk is a multiple of 32.
__global__ __launch_bounds__(256) void transpose_pad_kernel(const float *__restrict__ in, int64_t ldi, int64_t R, int C,
                                                           float *__restrict__ out, int64_t chunk, int64_t Rpad)
{
    __shared__ float tile[64][65];
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    const int c0 = blockIdx.y * 64;
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;  // 64 x 4
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int64_t r = r0 + ty + 4 * i;
        const int c = c0 + tx;
        tile[ty + 4 * i][tx] = (r < R && c < C) ? in[r * ldi + c] : 0.0f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int c = c0 + ty + 4 * i;
        const int64_t r = r0 + tx;
        if (c < C && r < Rpad) {
            const int64_t sp = r / chunk;
            out[(sp * C + c) * chunk + (r - sp * chunk)] = tile[tx][ty + 4 * i];
        }
    }
}

__global__ __launch_bounds__(kThreads, kWavesPerSimd) void dweight_bx3_kernel(
    const float *__restrict__ Xt, int64_t ldxt, const char *__restrict__ gpack, float *__restrict__ slabs, int Kf, int F,
    int ksteps_total, int chunk_ksteps, int n_splits, int m_tiles, int n_wg)
{
    __shared__ __attribute__((aligned(16))) char lds[kLdsBytes];
    const int xcd = blockIdx.x & 7, q = blockIdx.x >> 3;
    const int tiles = m_tiles * n_wg;
    const int split = xcd + 8 * (q / tiles), tile = q % tiles;
    if (split >= n_splits) return;  // before any barrier
    const int m_tile = tile / n_wg, n_wgi = tile % n_wg;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int m0 = m_tile * BM;
    const int n_tiles_total = (F + NT - 1) / NT;
    const int nt0 = n_wgi * (BN / NT) + wn * RN;
    const int ks0 = split * chunk_ksteps;

    constexpr int NP = Geom<float>::NP;
    const float *arow[NP];
    bool avalid[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        int gm = m0 + stage_row<float>(i);
        gm = gm < Kf ? gm : Kf - 1;  // clamped: a row of Xt only feeds the same row of dW, never stored
        arow[i] = Xt + ((int64_t)split * Kf + gm) * ldxt;  // chunk `split` is a compact [Kf x ldxt] matrix
        avalid[i] = true;
    }
    f32x16 acc[4][RN];
    mainloop<float, true, true, false>(arow, avalid, gpack + (int64_t)ks0 * 2 * FRAG_BYTES, chunk_ksteps * KSTEP,
                                       ksteps_total, wm, nt0, n_tiles_total, lds, acc);

    float *slab = slabs + (int64_t)split * Kf * F;
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        const int gn = (nt0 + j) * NT + (lane & 31);
        if (gn >= F) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int gmb = m0 + wm * 128 + i * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int gm = gmb + (r & 3) + 8 * (r >> 2);
                if (gm < Kf) slab[(int64_t)gm * F + gn] = acc[i][j][r];
            }
        }
    }
}

__global__ __launch_bounds__(256) void slab_sum_kernel(const float *__restrict__ slabs, int n_slabs, int64_t kf, int F,
                                                      float *__restrict__ dW, int64_t lddw)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= kf) return;
    float s = 0.0f;
    for (int z = 0; z < n_slabs; ++z) s += slabs[(int64_t)z * kf + i];  // fixed order
    dW[(i / F) * lddw + (i % F)] = s;
}

struct Plan {
    int m_tiles, n_wg, n_splits, chunk_ksteps, ksteps_total;
    int64_t n_pad;                    // columns of Xt = node rows covered by the chunks
    size_t xt_bytes, pack_bytes, slab_bytes;
};

Plan plan_for(int64_t N, int K, int F)
{
    Plan p;
    p.m_tiles = (K + BM - 1) / BM;
    p.n_wg = (F + BN - 1) / BN;
    const int tiles = p.m_tiles * p.n_wg;
    int per_xcd = 64 / tiles;          // chunks that fit the 64 resident workgroups of an XCD at once
    if (per_xcd < 1) per_xcd = 1;
    int64_t s = 8 * (int64_t)per_xcd;
    const int64_t max_s = (N + 511) / 512;  // at least 512 node rows per chunk
    if (s > max_s) s = max_s;
    if (s < 1) s = 1;
    p.n_splits = (int)s;
    const int64_t rows = (N + s - 1) / s;
    p.chunk_ksteps = (int)((rows + BK - 1) / BK) * (BK / KSTEP);
    p.ksteps_total = p.chunk_ksteps * p.n_splits;
    p.n_pad = (int64_t)p.ksteps_total * KSTEP;
    p.xt_bytes = (size_t)K * p.n_pad * sizeof(float);
    const size_t n_tiles = (size_t)(F + NT - 1) / NT;
    p.pack_bytes = n_tiles * p.ksteps_total * 2 * FRAG_BYTES + 4096;  // + the one-step-ahead read past the last chunk
    p.slab_bytes = (size_t)p.n_splits * K * F * sizeof(float);
    return p;
}

inline size_t up256(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace

size_t dweight_bx3_workspace_bytes(int64_t N, int K, int F)
{
    if (N <= 0 || K <= 0 || F <= 0) return 0;
    const Plan p = plan_for(N, K, F);
    const size_t a = up256(p.xt_bytes) + up256(p.pack_bytes) + up256(p.slab_bytes), b = dweight_tn_workspace_bytes(N, K, F);
    return a > b ? a : b;   // (the caller does not know yet which form its pointers will take)
}

int dweight_bx3(const float *X, int64_t ldx, const float *G, int64_t ldg, int64_t N, int K, int F, float *dW,
                int64_t lddw, void *workspace, hipStream_t st)
{
    if (!X || !G || !dW || !workspace) return fail(GGCN_EINVAL, "ggcn_dweight: null pointer");
    if (N <= 0 || K <= 0 || F <= 0) return fail(GGCN_EINVAL, "ggcn_dweight: N=%lld K=%d F=%d must be positive", (long long)N, K, F);
    if (ldx < K || ldg < F || lddw < F) return fail(GGCN_EINVAL, "ggcn_dweight: leading dimension too small");
    if (!aligned16(workspace)) return fail(GGCN_EINVAL, "ggcn_dweight: workspace must be 16-byte aligned");
    // 16-byte aligned rows: the native TN form (dweight_tn.hip: both operands read as they lie, no transpose / pack pass)
    if (dweight_tn_takes(X, ldx, G, ldg, N, K, F) && !getenv("GGCN_DWEIGHT_TRANSPOSE")) return dweight_tn(X, ldx, G, ldg, N, K, F, dW, lddw, workspace, st);
    const Plan p = plan_for(N, K, F);
    if (p.ksteps_total > 65535 * 16 || p.n_pad > (int64_t)INT32_MAX)
        return fail(GGCN_EUNSUPPORTED, "ggcn_dweight: too many node rows for the bf16x3 form");
    char *ws = static_cast<char *>(workspace);
    float *xt = reinterpret_cast<float *>(ws);
    char *gpack = ws + up256(p.xt_bytes);
    float *slabs = reinterpret_cast<float *>(ws + up256(p.xt_bytes) + up256(p.pack_bytes));

    // 1. Xt = X^T, zero-padded to the chunk grid
    const dim3 tgrid((unsigned)((p.n_pad + 63) / 64), (unsigned)((K + 63) / 64));
    const int64_t chunk = (int64_t)p.chunk_ksteps * KSTEP;
    hipLaunchKernelGGL(transpose_pad_kernel, tgrid, dim3(256), 0, st, X, ldx, N, K, xt, chunk, p.n_pad);
    // 2. dH -> fragment-ordered hi/lo image over the padded node axis (rows past N read as zeros)
    int rc = weight_pack_rows(G, ldg, N, F, p.ksteps_total, gpack, st);
    if (rc) return rc;
    // 3. split-K bf16x3 GEMM: ids of one residue mod 8 (one XCD) share a chunk
    const int tiles = p.m_tiles * p.n_wg;
    const int64_t grid = (int64_t)8 * tiles * ((p.n_splits + 7) / 8);
    hipLaunchKernelGGL(dweight_bx3_kernel, dim3((unsigned)grid), dim3(kThreads), 0, st, xt, chunk, gpack, slabs, K, F,
                       p.ksteps_total, p.chunk_ksteps, p.n_splits, p.m_tiles, p.n_wg);
    // 4. fixed-order sum of the slabs
    const int64_t kf = (int64_t)K * F;
    hipLaunchKernelGGL(slab_sum_kernel, dim3((unsigned)((kf + 255) / 256)), dim3(256), 0, st, slabs, p.n_splits, kf, F, dW,
                       lddw);
    return check_launch("ggcn_dweight(bf16x3)");
}

}  // namespace ggcn
