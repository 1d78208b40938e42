// Backward of the gate / max-pool epilogue AND the transposed aggregation in one launch, on the matrix cores (graphs of <= 32
// nodes, 0/1 adjacency, no gate dropout):
//
//   dY[t] = d_out[t]*sg + [t = argmax_a] d_pa*ga + [t = argmax_b] d_pb*gb         models/bert_amir5.py:627-640 backwards
//   dH    = A^T . D . dY,   D = diag(1 / (rowsum(A) + 1))                           models/gcn.py:35,41 backwards (train.py:120)
//
// gate_pool_backward_agg_kernel (gate_pool_backward.hip) keeps 32 x 4 sums per thread and scatters every dY row into the rows
// its mask names: 1024 scalar bit tests per thread, 431 us per layer at config 2 -- the largest kernel of the training step.
// Here dH_g [32 x F] = (A_g^T) [32 x 32] . (D.dY_g) [32 x F] is what it looks like: one MFMA chain per (graph, 32 columns), as
// in the forward's epilogue (fused_common.h):
//   * ONE workgroup per graph -- its rows are one contiguous block of `out`, `d_out` and dH, streamed once: a first form with a
//     workgroup per (graph, 256 columns) spread a graph's 3 KiB rows over three XCDs and ran 612 us where this one runs 292 --
//     whose four wavefronts walk the 64-column groups w, w + 4, ...; a group = two 32-column tiles; lane (c, h) holds, of column c, the 16 rows
//     (r & 3) + 8 (r >> 2) + 4 h, r = 0..15 -- the accumulator's register -> row map, which is also the k order the aggregation
//     MFMAs of this library use (element e of k-step s = register 8 s + e), so what a lane loads is what it multiplies;
//   * both arg-maxima of a column (the pools' winners, ties to the smaller row like the sequential kernel) meet across the two
//     lane halves; dY x 1/(deg + 1) is split into THREE bf16 planes (residual 2^-25: gradients have no range contract, and
//     the parity test wants dH to a few fp32 ulps of its scale) and multiplied by A^T, an exact 0/1 operand: 6 MFMAs per tile;
//   * A^T's fragments come from ggcn_graph_operands blocks of the TRANSPOSED row masks (ggcn_rowmask_transpose, once per
//     adjacency tensor), 1/(deg + 1) in register order from the graph's own blocks -- nothing about a graph is computed here;
//   * dH leaves through LDS as 16-byte row stores; max |dH| (ggcn_linear_scaled's scale) falls out of the accumulators.
// Traffic: out + d_out read once (the scalar kernel reads d_out twice), dH written once: 1.2 GB in 292 us at config 2 (4.1 TB/s;
// the scalar form on the same box: 395-400 us).  150 VGPRs, three workgroups per CU.
#include "bf16x3_core.h"
#include "common.h"

namespace ggcn {
namespace {

using namespace bx3;

constexpr int kOpsBytesB = GGCN_GRAPH_OPS_BYTES;   // [0,1024) A fragments k-step 0, [1024,2048) k-step 1, [2048,2176) 1/(deg+1) float[h][16]

__device__ __forceinline__ int row_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// v -> three bf16 planes (p0 + p1 + p2 = v to 2^-25) as B-operand fragments of the two k-steps
__device__ __forceinline__ void split3(const float (&v)[16], bf16x8 (&f)[3][2])
{
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = v[8 * s + e];
            const __bf16 p0 = (__bf16)x;
            const float r1 = x - (float)p0;
            const __bf16 p1 = (__bf16)r1;
            const float r2 = r1 - (float)p1;
            f[0][s][e] = p0;
            f[1][s][e] = p1;
            f[2][s][e] = (__bf16)r2;
        }
}

__global__ __launch_bounds__(256, 3) void gate_pool_backward_mma_kernel(
    const float *__restrict__ out, int64_t ldo, const float *__restrict__ store_gate, const float *__restrict__ gate_a,
    const float *__restrict__ gate_b, const float *__restrict__ d_out, int64_t ldd, const float *__restrict__ d_pa,
    const float *__restrict__ d_pb, const char *__restrict__ ops, const char *__restrict__ ops_t, int T, int F, int n_slabs,
    float *__restrict__ dH, int64_t ldh, float *__restrict__ d_sg, float *__restrict__ d_ga, float *__restrict__ d_gb,
    float *__restrict__ d_bsum, unsigned int *__restrict__ dh_amax)
{
    __shared__ __attribute__((aligned(16))) float stage_all[4][32 * 64];   // per wavefront: 32 rows x 64 columns on their way to 16-byte stores
    const int b = blockIdx.x;          // one workgroup per graph: its rows are ONE contiguous block of `out` / `d_out` / dH
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 31, h = lane >> 5;
    float *stage = stage_all[wave];

    // the graph's operands: A^T as the aggregation MFMA's A fragments (0xFFFF elements -> bf16 1.0), 1/(deg + 1) in register order
    bf16x8 aft[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const uint4 raw = *reinterpret_cast<const uint4 *>(ops_t + (int64_t)b * kOpsBytesB + s * 1024 + lane * 16);
        union { bf16x8 v; uint32_t w[4]; } u;
        u.w[0] = raw.x & 0x3F803F80u; u.w[1] = raw.y & 0x3F803F80u; u.w[2] = raw.z & 0x3F803F80u; u.w[3] = raw.w & 0x3F803F80u;
        aft[s] = u.v;
    }
    float rinv[16];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float4 t = *reinterpret_cast<const float4 *>(ops + (int64_t)b * kOpsBytesB + 2048 + h * 64 + q * 16);
        rinv[4 * q] = t.x; rinv[4 * q + 1] = t.y; rinv[4 * q + 2] = t.z; rinv[4 * q + 3] = t.w;
    }

    float amax = 0.0f;
#pragma unroll 1
    for (int col0 = wave * 64; col0 < F; col0 += 256) {   // wavefront-uniform walk over this wavefront's 64-column groups
#pragma unroll 1   // (one tile at a time: ~100 registers instead of 175 -- four wavefronts per SIMD keep more loads in flight than two)
    for (int j = 0; j < 2; ++j) {
        if (col0 + 32 * j >= F) break;   // wavefront-uniform: the group's last tile
        const int col = col0 + 32 * j + c;
        const bool cok = col < F;
        const int colc = cok ? col : 0;      // (a lane past F works on column 0's data and stores nothing)
        const int64_t gf = (int64_t)b * F + colc;
        const float sg = store_gate ? store_gate[gf] : 1.0f;
        const float inv_sg = store_gate ? (sg != 0.0f ? 1.0f / sg : 0.0f) : 1.0f;
        const float ga = gate_a ? gate_a[gf] : 1.0f, gb = gate_b ? gate_b[gf] : 1.0f;
        const float dpa = d_pa ? d_pa[gf] : 0.0f, dpb = d_pb ? d_pb[gf] : 0.0f;
        // the graph's rows of `out` and `d_out` behind buffer resources that END with its last row: a lane's 16 rows are one lane
        // offset (its column, its half's 4-row shift) + 16 SCALAR offsets, and rows past T read zeros by the hardware's range check
        // -- no 64-bit lane arithmetic (32 address pairs cost 64 registers), no select per value
        const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(out + (int64_t)b * T * ldo), 0,
                                                                              (int)((((int64_t)T - 1) * ldo + F) * 4), 0x00020000);
        const __amdgpu_buffer_rsrc_t drsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>((d_out ? d_out : out) + (int64_t)b * T * (d_out ? ldd : ldo)), 0,
                                                                              d_out ? (int)((((int64_t)T - 1) * ldd + F) * 4) : 0, 0x00020000);
        const int ovoff = (int)((4 * h * ldo + colc) * 4), dvoff = (int)((4 * h * ldd + colc) * 4);
        float ov[16], dv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int rs = (r & 3) + 8 * (r >> 2);   // this lane's row is rs + 4 h
            ov[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(orsrc, ovoff, (int)(rs * ldo * 4), 0));
            dv[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(drsrc, dvoff, (int)(rs * ldd * 4), 0));   // (no d_out: an empty resource, zeros)
        }
        // forward values and the pools' winners (bert_amir5.py:627-640): first maximum in ascending row order
        float best_a = -INFINITY, best_b = -INFINITY, ya = 0.0f, yb = 0.0f, acc_sg = 0.0f;
        int ia = 0, ib = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row_of(r, h);
            const bool valid = row < T;
            const float y = ov[r] * inv_sg;          // (rows past T: 0)
            const float va = y * ga, vb = y * gb;
            if (valid && va > best_a) { best_a = va; ia = row; ya = y; }
            if (valid && vb > best_b) { best_b = vb; ib = row; yb = y; }
            acc_sg = fmaf(dv[r], y, acc_sg);
        }
        {   // the two lane halves hold different rows of the same column: the smaller row wins a tie
            const float oa = __shfl_xor(best_a, 32), oya = __shfl_xor(ya, 32);
            const int oia = __shfl_xor(ia, 32);
            const bool ta = oa > best_a || (oa == best_a && oia < ia);
            best_a = ta ? oa : best_a; ia = ta ? oia : ia; ya = ta ? oya : ya;
            const float ob = __shfl_xor(best_b, 32), oyb = __shfl_xor(yb, 32);
            const int oib = __shfl_xor(ib, 32);
            const bool tb = ob > best_b || (ob == best_b && oib < ib);
            best_b = tb ? ob : best_b; ib = tb ? oib : ib; yb = tb ? oyb : yb;
        }
        // dY and D.dY in register order
        float gw[16], bsum = 0.0f;
        const float pa_g = dpa * ga, pb_g = dpb * gb;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row_of(r, h);
            float g = dv[r] * sg;
            if (d_pa && row == ia && row < T) g += pa_g;
            if (d_pb && row == ib && row < T) g += pb_g;
            bsum += g;
            gw[r] = g * rinv[r];                                   // gcn.py:35: the row's own 1 / (deg + 1)
        }
        bsum += __shfl_xor(bsum, 32);
        acc_sg += __shfl_xor(acc_sg, 32);
        // dH tile = A^T . (D.dY): three bf16 planes, smallest first
        bf16x8 pl[3][2];
        split3(gw, pl);
        f32x16 y;
#pragma unroll
        for (int r = 0; r < 16; ++r) y[r] = 0.0f;
#pragma unroll
        for (int p = 2; p >= 0; --p)
#pragma unroll
            for (int s = 0; s < 2; ++s) y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(aft[s], pl[p][s], y, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            amax = fmaxf(amax, fabsf(y[r]));   // (plain C: an inline-asm reader of a fresh MFMA result is a hazard the compiler does not pad)
            // staged for the 16-byte row stores below (the forward's scheme: rows with bit 2 set swap their 32-column halves)
            stage[row_of(r, h) * 64 + ((32 * j + c) ^ (32 * h))] = y[r];
        }
        if (h == 0 && cok) {
            if (d_bsum) d_bsum[gf] = bsum;
            if (d_sg) d_sg[gf] = acc_sg;
            if (d_ga) d_ga[gf] = dpa * ya;
            if (d_gb) d_gb[gf] = dpb * yb;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
        const int colq = (lane & 15) * 4;
        float *gbase = dH + (int64_t)b * T * ldh + col0 + colq;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = 4 * it + (lane >> 4);
            const float4 v4 = *reinterpret_cast<const float4 *>(&stage[row * 64 + (colq ^ (32 * ((row >> 2) & 1)))]);
            if (row < T && col0 + colq < F) *reinterpret_cast<float4 *>(gbase + (int64_t)row * ldh) = v4;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   // (the next group's tiles overwrite the staging area)
    __builtin_amdgcn_wave_barrier();
    }   // column groups
    if (dh_amax) {
        if (!(amax <= 3.0e38f)) amax = __builtin_inff();   // NaN / inf in the gradients: say so
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) amax = fmaxf(amax, __shfl_xor(amax, d));
        if (lane == 0) atomicMax(dh_amax, __float_as_uint(amax));
    }
}

// row masks of the transposed adjacency, graphs of <= 32 nodes: one wavefront per graph, lane t holds row t's word; bit t of
// the word of row s = bit s of the word of row t (32 ballots)
__global__ __launch_bounds__(256) void rowmask_transpose_kernel(const uint32_t *__restrict__ rowmask, int B, int T,
                                                                uint32_t *__restrict__ rowmask_t)
{
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= B) return;   // wavefront-uniform
    const uint32_t m = lane < T ? rowmask[(int64_t)g * T + lane] : 0u;
    uint32_t mt = 0u;
#pragma unroll
    for (int s = 0; s < 32; ++s) {
        const unsigned long long bal = __ballot((m >> s) & 1u);
        if (lane == s) mt = (uint32_t)bal;
    }
    if (lane < T) rowmask_t[(int64_t)g * T + lane] = mt;
}

}  // namespace

int rowmask_transpose(const uint32_t *rowmask, int B, int T, uint32_t *rowmask_t, hipStream_t st)
{
    if (!rowmask || !rowmask_t) return fail(GGCN_EINVAL, "ggcn_rowmask_transpose: null pointer");
    if (B <= 0 || T <= 0) return fail(GGCN_EINVAL, "ggcn_rowmask_transpose: B=%d T=%d must be positive", B, T);
    if (T > 32) return fail(GGCN_EUNSUPPORTED, "ggcn_rowmask_transpose: T=%d > 32 (one word per node)", T);
    hipLaunchKernelGGL(rowmask_transpose_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, rowmask, B, T, rowmask_t);
    return check_launch("ggcn_rowmask_transpose");
}

int gate_pool_backward_mma(const float *out, int64_t ldo, const float *store_gate, const float *gate_a, const float *gate_b,
                           const float *d_out, int64_t ldd, const float *d_pa, const float *d_pb, const void *graph_ops,
                           const void *graph_ops_t, int B, int T, int F, float *dH, int64_t ldh, float *d_sg, float *d_ga,
                           float *d_gb, float *d_bsum, float *dh_amax, hipStream_t st)
{
    const char *who = "ggcn_gate_pool_backward_mma";
    if (!out || !dH || !graph_ops || !graph_ops_t) return fail(GGCN_EINVAL, "%s: null pointer", who);
    if (B <= 0 || T <= 0 || F <= 0) return fail(GGCN_EINVAL, "%s: B=%d T=%d F=%d must be positive", who, B, T, F);
    if (T > 32) return fail(GGCN_EUNSUPPORTED, "%s: T=%d > 32; use ggcn_gate_pool_backward + ggcn_aggregate_t", who, T);
    if (ldo < F || ldh < F || (d_out && ldd < F)) return fail(GGCN_EINVAL, "%s: leading dimension smaller than F=%d", who, F);
    if ((int64_t)T * (ldo > ldd ? ldo : ldd) * 4 >= ((int64_t)1 << 31)) return fail(GGCN_EUNSUPPORTED, "%s: a graph's rows exceed 2 GiB", who);
    if (F % 4 != 0 || ldh % 4 != 0 || !aligned16(dH) || !aligned16(graph_ops) || !aligned16(graph_ops_t))
        return fail(GGCN_EUNSUPPORTED, "%s: needs F %% 4 == 0, ldh %% 4 == 0 and 16-byte aligned dH / operand blocks; use "
                                       "ggcn_gate_pool_backward_agg or the two calls", who);
    const int n_slabs = 1;   // (one workgroup per graph; its wavefronts walk the column groups)
    const int64_t blocks = (int64_t)B;
    if (blocks > (int64_t)INT32_MAX) return fail(GGCN_EUNSUPPORTED, "%s: grid too large", who);
    hipLaunchKernelGGL(gate_pool_backward_mma_kernel, dim3((unsigned)blocks), dim3(256), 0, st, out, ldo, store_gate, gate_a, gate_b,
                       d_out, ldd, d_pa, d_pb, static_cast<const char *>(graph_ops), static_cast<const char *>(graph_ops_t), T, F,
                       n_slabs, dH, ldh, d_sg, d_ga, d_gb, d_bsum, reinterpret_cast<unsigned int *>(dh_amax));
    return check_launch(who);
}

}  // namespace ggcn
