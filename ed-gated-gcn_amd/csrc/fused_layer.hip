// One gated-GCN layer in ONE launch, for graphs of at most 32 nodes with a binary adjacency
// (the reference's case: ACE sentences, ORI_ML = 31, 0/1 dependency matrices, graph.py:66-74):
//
//   hidden = X.W                                   models/gcn.py:34        f16mx8 / bf16x3 main loop
//   agg    = adj.hidden                            models/gcn.py:41        MFMA on the accumulator
//   y      = agg / (rowsum(adj)+1) + bias          models/gcn.py:35,41,43  registers
//   out    = y * store_gate ; pools = max_t(y*g)   models/bert_amir5.py:627-640
//
// hidden never goes to HBM: a 32x32 fp32 accumulator tile of v_mfma_f32_32x32x16_bf16 IS
// "graph g's 32 nodes x 32 features", with the node index in the registers and the feature on
// the lane -- exactly the B-operand layout of a following MFMA that sums over nodes (cdna
// guide §3 "An accumulator tile as the next MFMA's operand").  So the neighbour sum is
//     agg[32 x 32] = ADJ_g[32 x 32] . hidden_g[32 x 32]
// with ADJ_g the graph's 0/1 matrix as an exact bf16 A operand, expanded in registers from a
// 32-bit row mask (BatchedCSR.rowmask: bit j of word i = edge i<-j; 4 B per node), and
// hidden split into two bf16 planes (residual 2^-17 |hidden|, ~4e-6 after the mean).
// Cost: 4 MFMAs per tile on top of the main loop's 72 (f16mx8) or 144 (bf16x3), no extra pass; the
// [N,F] output is staged through the idle A buffers in LDS so that it leaves as 16-byte row stores.
// k order inside a step: element j of lane half h is node 16s + 8(j>>2) + 4h + (j&3) for both
// operands (the register->row map of the accumulator), so no data moves between lanes.
//
// HBM traffic per layer = X in + out + 4 B/node masks + gates + W: the algorithmic bytes.
// Graphs with T < 32 occupy a 32-row slot (rows >= T read as zeros, are never stored and never
// pooled); T > 32 or weighted adjacency -> the unfused path (linear + aggregate.hip).
#include "f16mx8_core.h"

namespace ggcn {
namespace {

using namespace bx3;

// acc -> two bf16 planes (hi + lo, residual <= 2^-17 |v|) as B-operand fragments of the two k-steps
__device__ __forceinline__ void split2(const f32x16 &acc, bf16x8 (&frag)[2][2])
{
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = acc[8 * s + j];
            const __bf16 p0 = (__bf16)v;
            frag[0][s][j] = p0;
            frag[1][s][j] = (__bf16)(v - (float)p0);
        }
}

// FULLT: T == 32 and B % 4 == 0 (every row of every tile is a real node): drops every guard.
#ifdef GGCN_LAB_TRACE  // timeline probe: per workgroup {block, HW_ID, XCC_ID, t0, t1, t2, t3} in 10 ns ticks
__device__ unsigned long long ggcn_trace_buf[8192 * 8];
#define GGCN_TRACE(slot)                                                              \
    do {                                                                              \
        if (threadIdx.x == 0 && blockIdx.x < 8192) ggcn_trace_buf[blockIdx.x * 8 + (slot)] = wall_clock64(); \
    } while (0)
#else
#define GGCN_TRACE(slot) do { } while (0)
#endif

// SCH: 0 = bf16x3 main loop, 1 = f16mx8 (f16mx8_core.h)
// VST: the [N,F] output leaves through LDS as 16-byte row stores (needs F, ldo multiples of 4 and a 16-byte aligned out)
template <int SCH, bool AVEC, bool KFULL, bool FULLT, bool VST>
__global__ __launch_bounds__(kThreads, kWavesPerSimd) void layer_fused_kernel(
    const float *__restrict__ X, int64_t ldx, const char *__restrict__ wpack,
    const uint32_t *__restrict__ rowmask, const float *__restrict__ bias, int B, int T, int K, int F,
    const float *__restrict__ store_gate, const float *__restrict__ pool_gate_a,
    const float *__restrict__ pool_gate_b, float *__restrict__ out, int ldo,
    float *__restrict__ pool_a, float *__restrict__ pool_b, float *__restrict__ ov_partial,
    const float *__restrict__ ov_in, float *__restrict__ ov_out, int g_tiles, int n_wg, int k_steps)
{
#if defined(GGCN_LAB_LDS_PAD)
    __shared__ __attribute__((aligned(16))) char lds[kLdsBytes + GGCN_LAB_LDS_PAD];  // occupancy experiment
#else
    __shared__ __attribute__((aligned(16))) char lds[kLdsBytes];
#endif
    // bert_amir5.py:638 for the launch BEFORE this one on the stream: block 0 adds the per-(graph,
    // 64-column group) partial dot products that launch left in ov_in, in a fixed order
    if (ov_in && blockIdx.x == 0) {
        float *red = reinterpret_cast<float *>(lds);
        const int n_part = B * ((F + 63) / 64);
        float sdot = 0.0f;
        for (int idx = threadIdx.x; idx < n_part; idx += kThreads) sdot += ov_in[idx];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) sdot += __shfl_xor(sdot, d);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sdot;
        __syncthreads();
        if (threadIdx.x == 0) *ov_out = ((red[0] + red[1]) + (red[2] + red[3])) / (float)B;
        __syncthreads();
    }
    const int vid = blockIdx.x;
    int g_tile, n_wgi;
    if (!tile_of_block(vid, g_tiles, n_wg, g_tile, n_wgi)) return;
#ifdef GGCN_LAB_ONLY_N0  // probe: only the first column tile of every row block runs (how much of X is fetched once?)
    if (n_wgi != 0) return;
#endif
#ifdef GGCN_LAB_TRACE
    if (threadIdx.x == 0 && blockIdx.x < 8192) {
        ggcn_trace_buf[blockIdx.x * 8 + 0] = blockIdx.x;
        ggcn_trace_buf[blockIdx.x * 8 + 1] = __builtin_amdgcn_s_getreg(4 | (31 << 11));    // HW_REG_HW_ID
        ggcn_trace_buf[blockIdx.x * 8 + 2] = __builtin_amdgcn_s_getreg(20 | (31 << 11));   // HW_REG_XCC_ID
    }
    GGCN_TRACE(3);
#endif

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int gt0 = g_tile * (4 * WM);  // graph slots of 32 rows in this workgroup's tile
    const int g0 = gt0 + wm * 4;        // this wavefront's 4 graphs
    const int n_tiles_total = (F + NT - 1) / NT;
    const int nt0 = n_wgi * (BN / NT) + wn * RN;

    // tile row 32*slot + r  <->  node r of graph g0+slot
    constexpr int NP = Geom<float>::NP;
    const float *arow[NP];
    bool avalid[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int row = stage_row<float>(i);
        const int g = gt0 + (row >> 5), r = row & 31;
        avalid[i] = (g < B) && (FULLT || r < T);
        const int64_t node = avalid[i] ? (int64_t)g * T + r : 0;  // clamped, zeroed by the select
        arow[i] = X + node * ldx;
    }
    // this lane's adjacency row (node lane&31) of each of the 4 graphs: in flight under the main loop
    uint32_t mask[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int g = g0 + i;
        const bool ok = (g < B) && (FULLT || (lane & 31) < T);
        const uint32_t m = rowmask[ok ? (int64_t)g * T + (lane & 31) : 0];
        mask[i] = ok ? m : 0u;
    }

    f32x16 acc[4][RN];
    GGCN_TRACE(4);
#if defined(GGCN_LAB_PHASE)  // timing probe (wrong results): half the K loop for the odd row blocks of the first
                             // (PHASE=1) or the last (PHASE=2, control) round: does a half-period phase shift pay?
    const bool lab_short = (g_tile & 1) && (GGCN_LAB_PHASE == 1 ? blockIdx.x < 512 : blockIdx.x + 512 >= gridDim.x);
    const int K_loop = lab_short ? K / 2 : K;
#else
    const int K_loop = K;
#endif
    if constexpr (SCH == 0)
        bx3::mainloop<float, AVEC, KFULL, !FULLT>(arow, avalid, wpack, K_loop, k_steps, wm, nt0, n_tiles_total, lds, acc);
    else
#ifdef GGCN_MX_LAB_ROT
        mx8::mainloop<float, AVEC, KFULL, !FULLT>(arow, avalid, wpack, K_loop, k_steps / 2, wm, nt0, n_tiles_total, lds, acc,
                                                  (n_wgi * GGCN_MX_LAB_ROT) % ((K_loop + BK - 1) / BK));
#else
        mx8::mainloop<float, AVEC, KFULL, !FULLT>(arow, avalid, wpack, K_loop, k_steps / 2, wm, nt0, n_tiles_total, lds, acc);
#endif

    GGCN_TRACE(5);
    const int c = lane & 31, h = lane >> 5;

    // bias and the gates of the 4 graphs x 2 column tiles: all loads issued together, one latency
    float vb[RN], vsg[4][RN], vga[4][RN], vgb[4][RN];
    bool col_ok[RN];
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        const int gn = (nt0 + j) * NT + c;
        col_ok[j] = gn < F;
        const int gnc = col_ok[j] ? gn : 0;
        vb[j] = bias ? bias[gnc] : 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t at = (int64_t)(g0 + i < B ? g0 + i : 0) * F + gnc;
            vsg[i][j] = store_gate ? store_gate[at] : 1.0f;
            vga[i][j] = pool_gate_a ? pool_gate_a[at] : 1.0f;
            vgb[i][j] = pool_gate_b ? pool_gate_b[at] : 1.0f;
        }
    }
    const int lane_off = 4 * h * ldo + c;  // this lane's element inside a (graph, column tile) block
    // the A buffers are free after the main loop's last barrier: 8 KiB per wavefront = 32 rows x 64 columns
    float *stage_lds = reinterpret_cast<float *>(lds) + wave * (32 * 64);
    const int perm_base = 16 * h;          // ds_bpermute byte address of lane 4h (+ 4*row0 per register)

#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int g = g0 + i;
        if (g >= B) break;  // workgroup-uniform
        // ---- adjacency fragments of graph g from this lane's row mask: element j of k-step s is
        // node 16s + 8(j>>2) + 4h + (j&3); two neighbouring elements = two neighbouring mask bits ----
        const uint32_t mh = mask[i] >> (4 * h);
        union { bf16x8 v; uint32_t w[4]; } af[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int q = 0; q < 4; ++q) {  // element pair (2q, 2q+1): bits b, b+1 of mh
                const int b = 16 * s + 8 * (q >> 1) + 2 * (q & 1);
                const uint32_t two = (mh >> b) & 3u;
                af[s].w[q] = (two & 1u) * 0x3F80u + (two >> 1) * 0x3F800000u;  // bf16 1.0 = 0x3F80
            }
        // 1 / (rowsum(adj) + 1) of node lane&31 (gcn.py:35): one IEEE division per node, then the
        // value of row row0 + 4h is fetched per accumulator register through the LDS crossbar
        const float inv = 1.0f / (float)(__popc(mask[i]) + 1);
        float rinv[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row0 = (r & 3) + 8 * (r >> 2);
            rinv[r] = __int_as_float(__builtin_amdgcn_ds_bpermute(perm_base + 4 * row0, __float_as_int(inv)));
        }

        float dot = 0.0f;  // this wavefront's share of sum_f x1[g,f] * y1[g,f] (bert_amir5.py:638)
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            if (nt0 + j >= n_tiles_total) break;  // wavefront-uniform: column tile past F
            const int gn = (nt0 + j) * NT + c;

            bf16x8 hfrag[2][2];
            split2(acc[i][j], hfrag);
            f32x16 y;
#pragma unroll
            for (int r = 0; r < 16; ++r) y[r] = 0.0f;
#pragma unroll
            for (int p = 1; p >= 0; --p)  // small plane first
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s].v, hfrag[p][s], y, 0, 0, 0);

            // a gate is constant over the rows of a graph and rounding is monotonic, so
            // max_t fl(y_t * g) == fl(g * max_t y_t) for g >= 0 (and g * min_t y_t for g < 0):
            // track max and min of y once, apply both pool gates at the end (bert_amir5.py:635-640)
            float vmax = -INFINITY, vmin = INFINITY;
            float *tile = out ? out + ((int64_t)g * T) * ldo + (nt0 + j) * NT : nullptr;  // wave-uniform
            const float sg = vsg[i][j];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row0 = (r & 3) + 8 * (r >> 2);  // this lane's row is row0 + 4h
                const float v = y[r] * rinv[r] + vb[j];   // gcn.py:41,43
                if (VST) {
                    // staged for the 16-byte row stores below; columns of the rows with bit 2 set are
                    // swapped between the two 32-column halves so that h = 0 / 1 hit different banks
                    stage_lds[(row0 + 4 * h) * 64 + ((32 * j + c) ^ (32 * h))] = v * sg;
                }
                if (FULLT || row0 + 4 * h < T) {
                    if (!VST && tile && col_ok[j]) tile[lane_off + row0 * ldo] = v * sg;  // bert_amir5.py:626 / :639
                    vmax = fmaxf(vmax, v);
                    vmin = fminf(vmin, v);
                }
            }
            vmax = fmaxf(vmax, __shfl_xor(vmax, 32));
            vmin = fminf(vmin, __shfl_xor(vmin, 32));
            if (h == 0 && col_ok[j]) {
                const float ga = vga[i][j], gb = vgb[i][j];
                const float pa = ga * (ga >= 0.0f ? vmax : vmin), pb = gb * (gb >= 0.0f ? vmax : vmin);
                if (pool_a) pool_a[(int64_t)g * F + gn] = pa;
                if (pool_b) pool_b[(int64_t)g * F + gn] = pb;
                dot = fmaf(pa, pb, dot);
            }
        }
        if (ov_partial && nt0 < n_tiles_total) {  // fixed butterfly order; lanes with h = 1 hold 0
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) dot += __shfl_xor(dot, d);
            if (lane == 0) ov_partial[(int64_t)g * ((F + 63) / 64) + (nt0 >> 1)] = dot;
        }
        if (VST) {
            // rows of 64 columns (both column tiles of this wavefront) leave as 16 B per lane: one
            // instruction stores 4 rows x 256 contiguous bytes instead of 2 rows x 128 B
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int colq = (lane & 15) * 4;
            const int gcol = nt0 * NT + colq;
            float *gbase = out + ((int64_t)g * T) * ldo + gcol;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = 4 * it + (lane >> 4);
                const float4 v4 = *reinterpret_cast<const float4 *>(&stage_lds[row * 64 + (colq ^ (32 * ((row >> 2) & 1)))]);
                if ((FULLT || row < T) && gcol < F) *reinterpret_cast<float4 *>(gbase + row * ldo) = v4;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    GGCN_TRACE(6);
}

// rowmask from a batched CSR (T <= 32): one thread per node
__global__ __launch_bounds__(256) void rowmask_kernel(const int32_t *__restrict__ rowptr,
                                                      const int32_t *__restrict__ colidx, int64_t n, int T,
                                                      uint32_t *__restrict__ rowmask)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t base = i / T * T;
    uint32_t m = 0;
    for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) m |= 1u << (uint32_t)(colidx[e] - base);
    rowmask[i] = m;
}

}  // namespace

#ifdef GGCN_LAB_TRACE
extern "C" int ggcn_lab_trace_read(void *dst, size_t bytes)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(ggcn_trace_buf), bytes, 0, hipMemcpyDeviceToHost);
}
#endif

int csr_rowmask(const int32_t *rowptr, const int32_t *colidx, int B, int T, uint32_t *rowmask, hipStream_t st)
{
    if (!rowptr || !colidx || !rowmask) return fail(GGCN_EINVAL, "ggcn_csr_rowmask: null pointer");
    if (B <= 0 || T <= 0) return fail(GGCN_EINVAL, "ggcn_csr_rowmask: B=%d T=%d must be positive", B, T);
    if (T > 32) return fail(GGCN_EUNSUPPORTED, "ggcn_csr_rowmask: T=%d > 32 (row masks are 32-bit)", T);
    const int64_t n = (int64_t)B * T;
    hipLaunchKernelGGL(rowmask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, rowptr, colidx, n, T,
                       rowmask);
    return check_launch("ggcn_csr_rowmask");
}

int layer_fused(const float *X, int64_t ldx, const void *wpack, const uint32_t *rowmask, const float *bias,
                int B, int T, int K, int F, const float *store_gate, const float *pool_gate_a,
                const float *pool_gate_b, float *out, int64_t ldo, float *pool_a, float *pool_b,
                float *overlap_partial, const float *overlap_in, float *overlap_out, int precision, hipStream_t st)
{
    if ((overlap_in == nullptr) != (overlap_out == nullptr))
        return fail(GGCN_EINVAL, "ggcn_layer_fused: overlap_in and overlap_out go together");
    if (precision != GGCN_PREC_BF16X3 && precision != GGCN_PREC_F16MX8)
        return fail(GGCN_EUNSUPPORTED, "ggcn_layer_fused: precision %d (use bf16x3 or f16mx8)", precision);
    if (!X || !wpack || !rowmask) return fail(GGCN_EINVAL, "ggcn_layer_fused: null input pointer");
    if (B <= 0 || T <= 0 || K <= 0 || F <= 0)
        return fail(GGCN_EINVAL, "ggcn_layer_fused: B=%d T=%d K=%d F=%d must be positive", B, T, K, F);
    if (T > 32) return fail(GGCN_EUNSUPPORTED, "ggcn_layer_fused: T=%d > 32; use ggcn_linear + ggcn_aggregate", T);
    if (!out && !pool_a && !pool_b) return fail(GGCN_EINVAL, "ggcn_layer_fused: no output requested");
    if (ldx < K || (out && ldo < F)) return fail(GGCN_EINVAL, "ggcn_layer_fused: leading dimension too small");
    if (out && (int64_t)T * ldo >= (int64_t)INT32_MAX)
        return fail(GGCN_EUNSUPPORTED, "ggcn_layer_fused: T*ldo does not fit 32-bit offsets");
    if (!aligned16(wpack)) return fail(GGCN_EINVAL, "ggcn_layer_fused: wpack must be 16-byte aligned");
    const bool avec = (K % 4 == 0) && (ldx % 4 == 0) && aligned16(X);
    const bool kfull = (K % BK == 0);
    const int k_steps = round_up(K, BK) / KSTEP;
    const int64_t g_tiles = ((int64_t)B + 4 * WM - 1) / (4 * WM);
    const int n_wg = (F + BN - 1) / BN;
    const int64_t grid = grid_for(g_tiles, n_wg);
    if (grid > (int64_t)INT32_MAX) return fail(GGCN_EUNSUPPORTED, "ggcn_layer_fused: batch too large");
    const char *wp = static_cast<const char *>(wpack);
    const bool fullt = (T == 32) && (B % (4 * WM) == 0);
    const bool vst = out && (F % 4 == 0) && (ldo % 4 == 0) && aligned16(out);
#define GGCN_LAUNCH(SC, AV, KF, FT, VS)                                                                                  \
    hipLaunchKernelGGL((layer_fused_kernel<SC, AV, KF, FT, VS>), dim3((unsigned)grid), dim3(kThreads), 0, st, X, ldx, wp, \
                       rowmask, bias, B, T, K, F, store_gate, pool_gate_a, pool_gate_b, out, (int)ldo, pool_a,            \
                       pool_b, overlap_partial, overlap_in, overlap_out, (int)g_tiles, n_wg, k_steps)
#define GGCN_PICK(SC)                                                           \
    do {                                                                        \
        if (avec && kfull && fullt && vst) GGCN_LAUNCH(SC, true, true, true, true);   \
        else if (avec && kfull && fullt) GGCN_LAUNCH(SC, true, true, true, false);    \
        else if (avec && kfull && vst) GGCN_LAUNCH(SC, true, true, false, true);      \
        else if (avec && kfull) GGCN_LAUNCH(SC, true, true, false, false);            \
        else if (avec) GGCN_LAUNCH(SC, true, false, false, false);                    \
        else GGCN_LAUNCH(SC, false, false, false, false);                             \
    } while (0)
    if (precision == GGCN_PREC_F16MX8) GGCN_PICK(1);
    else GGCN_PICK(0);
#undef GGCN_PICK
#undef GGCN_LAUNCH
    return check_launch("ggcn_layer_fused");
}

}  // namespace ggcn
