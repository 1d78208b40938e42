// One gated-GCN layer -- or the reference's whole two-layer gated block -- in ONE launch, for graphs of
// at most 32 nodes with a binary adjacency (the reference's case: ACE sentences, ORI_ML = 31, 0/1
// dependency matrices, graph.py:66-74):
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
// THE BLOCK IN ONE LAUNCH (ggcn_block_fused).  bert_amir5.py:626-640 feeds gc2 with the UNGATED gcn1 and
// applies no non-linearity in between (gcn.py:19 declares a Tanh and never uses it), so with
// D = diag(1/(rowsum(A)+1)):
//     gcn1 = D.A.X.W1 + 1.b1^T
//     gcn2 = D.A.gcn1.W2 + 1.b2^T = D.A.( D.A.(X.W12) + 1.c^T ) + 1.b2^T,   W12 = W1.W2,  c = W2^T.b1
// Both layers are then products of the SAME input X: one grid covers the column tiles of [W1 | W12]; the
// W1 tiles end in the layer-1 epilogue (pools x1, y1 and the regulariser's partial sums; gcn1 itself is
// written only if the caller asks for it), the W12 tiles apply the adjacency twice ("mid" bias c in
// between) and end in the layer-2 epilogue (x = gate2*gcn2 stored, pool out).  gcn1 never makes its
// 4.N.F-byte round trip through HBM and X is read once for both layers; flops are unchanged (W12 and c
// are made once per weight update by the exact-fp32 linear).
//
// HBM traffic per layer = X in + out + 4 B/node masks + gates + W: the algorithmic bytes.
// Graphs with T < 32 occupy a 32-row slot (rows >= T read as zeros, are never stored and never
// pooled); T > 32 or weighted adjacency -> the unfused path (linear + aggregate.hip).
#include "fused_common.h"

#include <cstdlib>

namespace ggcn {
namespace {

__global__ __launch_bounds__(256) void graph_operands_kernel(const uint32_t *__restrict__ rowmask, int B, int T,
                                                             char *__restrict__ ops)
{
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= B) return;   // wavefront-uniform
    const int r = lane & 31, h = lane >> 5;
    const uint32_t m = r < T ? rowmask[(int64_t)g * T + r] : 0u;   // T <= 32: one word per node
    char *blk = ops + (int64_t)g * kOpsBytes;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        uint32_t w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int b = 16 * s + 8 * (q >> 1) + 2 * (q & 1) + 4 * h;
            const uint32_t two = (m >> b) & 3u;
            w[q] = (two & 1u) * 0xFFFFu + (two >> 1) * 0xFFFF0000u;
        }
        *reinterpret_cast<uint4 *>(blk + s * 1024 + lane * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    }
    // lane idx (0..31) writes entry [h' = idx >> 4][r' = idx & 15]
    const int rr = lane & 15, hh = (lane >> 4) & 1;
    const int row = (rr & 3) + 8 * (rr >> 2) + 4 * hh;
    const uint32_t mrow = __shfl(m, row);
    if (lane < 32) reinterpret_cast<float *>(blk + 2048)[lane] = 1.0f / (float)(__popc(mrow) + 1);
}

// ggcn_graph_operands2: M2 = (D.A)^2 per graph, scaled by 2^10, as hi / lo A-operand fragments of the plane type (layout:
// fused_common.h), and rowsum(D.A).  One wavefront per graph; lane (r, h) owns row r and the 16 columns its fragments hold:
//   M2[r][c] = 1/(deg_r + 1) * sum_{j in N(r)} A[j][c] / (deg_j + 1)          (fp32, j ascending)
template <int PLANE>   // 0: bf16 pairs (GGCN_PREC_BF16X3), 1: fp16 pairs (GGCN_PREC_F16MX8)
__global__ __launch_bounds__(256) void graph_operands2_kernel(const uint32_t *__restrict__ rowmask, int B, int T,
                                                              char *__restrict__ ops2)
{
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= B) return;   // wavefront-uniform
    const int r = lane & 31, h = lane >> 5;
    const uint32_t m = r < T ? rowmask[(int64_t)g * T + r] : 0u;   // T <= 32: one word per node; lanes r and r + 32 hold row r
    const float inv_r = 1.0f / (float)(__popc(m) + 1);             // gcn.py:35
    float a2[2][8];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) a2[s][e] = 0.0f;
    for (int j = 0; j < T; ++j) {   // wavefront-uniform walk over the possible neighbours
        const uint32_t mj = (uint32_t)__builtin_amdgcn_readlane((int)m, j);   // row j's mask
        const float invj = 1.0f / (float)(__popc(mj) + 1);
        if ((m >> j) & 1u) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int c = 16 * s + 8 * (e >> 2) + 4 * h + (e & 3);
                    a2[s][e] += ((mj >> c) & 1u) ? invj : 0.0f;
                }
        }
    }
    char *blk = ops2 + (int64_t)g * kOps2Bytes;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        union { uint4 q; unsigned short u[8]; } hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = a2[s][e] * inv_r * kM2Scale;
            if constexpr (PLANE == 1) {
                const _Float16 vh = (_Float16)v, vl = (_Float16)(v - (float)vh);
                hi.u[e] = __builtin_bit_cast(unsigned short, vh);
                lo.u[e] = __builtin_bit_cast(unsigned short, vl);
            } else {
                const __bf16 vh = (__bf16)v, vl = (__bf16)(v - (float)vh);
                hi.u[e] = __builtin_bit_cast(unsigned short, vh);
                lo.u[e] = __builtin_bit_cast(unsigned short, vl);
            }
        }
        *reinterpret_cast<uint4 *>(blk + s * 1024 + lane * 16) = hi.q;
        *reinterpret_cast<uint4 *>(blk + 2048 + s * 1024 + lane * 16) = lo.q;
    }
    // rowsum(D.A) = deg / (deg + 1) in accumulator order: lane idx (0..31) writes entry [h' = idx >> 4][r' = idx & 15]
    const int rr = lane & 15, hh = (lane >> 4) & 1;
    const int row = (rr & 3) + 8 * (rr >> 2) + 4 * hh;
    const uint32_t mrow = __shfl(m, row);
    if (lane < 32) reinterpret_cast<float *>(blk + 4096)[lane] = (float)__popc(mrow) / (float)(__popc(mrow) + 1);
}

// ggcn_graph_operands_weighted: a REAL-valued adjacency (gcn.py:33 takes any `adj`; the reference's own graphs are 0/1) in the
// format of the (D.A)^2 blocks above, so that the one-launch layer's MID epilogue applies it: M = D.A_w with
// D = diag(1 / (rowsum(A_w) + 1)) (gcn.py:35), scaled by 2^10, as hi / lo fragments of the plane type; the rowsum field is 0 (no
// `mid` bias rides along).  One wavefront per graph, lane (r, h) walks row r of the CSR (edges in CSR order, as ggcn_aggregate
// sums them) and keeps the 16 columns its fragments hold.  flag (optional): bit 0 is set when an entry does not fit the plane
// type (|M| * 2^10 >= 60000 in fp16 planes, or not finite) -- the caller then keeps linear + aggregate.
template <int PLANE>
__global__ __launch_bounds__(256) void graph_operands_w_kernel(const int32_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
                                                               const float *__restrict__ vals, int B, int T, char *__restrict__ ops,
                                                               int *__restrict__ flag)
{
    const int lane = threadIdx.x & 63;
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (g >= B) return;   // wavefront-uniform
    const int r = lane & 31, h = lane >> 5;
    const int64_t node0 = (int64_t)g * T;
    float a[16], wsum = 0.0f;
#pragma unroll
    for (int e = 0; e < 16; ++e) a[e] = 0.0f;
    if (r < T) {
        const int e1 = rowptr[node0 + r + 1];
        for (int e = rowptr[node0 + r]; e < e1; ++e) {
            const int c = colidx[e] - (int)node0;
            const float w = vals ? vals[e] : 1.0f;
            wsum += w;
            // column c sits in k-step c >> 4 as element 4 ((c >> 3) & 1) + (c & 3) of the lane half (c >> 2) & 1
            const int idx = ((unsigned)c < 32u && ((c >> 2) & 1) == h) ? (c >> 4) * 8 + ((c >> 3) & 1) * 4 + (c & 3) : -1;
#pragma unroll
            for (int q = 0; q < 16; ++q) a[q] += idx == q ? w : 0.0f;
        }
    }
    const float inv = 1.0f / (wsum + 1.0f);   // gcn.py:35
    char *blk = ops + (int64_t)g * kOps2Bytes;
    bool bad = false;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        union { uint4 q; unsigned short u[8]; } hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = a[8 * s + e] * inv * kM2Scale;
            bad = bad || !(fabsf(v) < (PLANE == 1 ? 60000.0f : 3.0e38f));
            if constexpr (PLANE == 1) {
                const _Float16 vh = (_Float16)v, vl = (_Float16)(v - (float)vh);
                hi.u[e] = __builtin_bit_cast(unsigned short, vh);
                lo.u[e] = __builtin_bit_cast(unsigned short, vl);
            } else {
                const __bf16 vh = (__bf16)v, vl = (__bf16)(v - (float)vh);
                hi.u[e] = __builtin_bit_cast(unsigned short, vh);
                lo.u[e] = __builtin_bit_cast(unsigned short, vl);
            }
        }
        *reinterpret_cast<uint4 *>(blk + s * 1024 + lane * 16) = hi.q;
        *reinterpret_cast<uint4 *>(blk + 2048 + s * 1024 + lane * 16) = lo.q;
    }
    if (lane < 32) reinterpret_cast<float *>(blk + 4096)[lane] = 0.0f;
    if (bad && flag) atomicOr(flag, 1);
}

// SCH: 0 = bf16x3 main loop, 1 = f16mx8 (f16mx8_core.h)
// FULLT: T == 32 and B % 4 == 0 (every row of every tile is a real node): drops every guard.
// VST: the [N,F] output leaves through LDS as 16-byte row stores (needs F, ldo multiples of 4 and a 16-byte aligned out)
// STAMP (diagnostic instantiation only, ggcn_debug_block_fused_stamped): thread 0 of every workgroup stamps s_memtime (shader
// cycles) and s_memrealtime (100 MHz) around the main loop into a.stamps, a buffer nothing else reads -- the clock the chip
// holds under THIS kernel's load (MI355X_MICROARCH.md "DVFS give-back" (6)).  No product launch executes a stamp.
template <int SCH, bool AVEC, bool KFULL, bool FULLT, bool VST, bool STAMP = false>
__global__ __launch_bounds__(kThreads, kWavesPerSimd) void layer_fused_kernel(const FusedArgs a)
{
    __shared__ __attribute__((aligned(16))) char lds[kLdsBytes + kEpiLdsBytes + GGCN_LAB_LDS_PAD];
    const int B = a.B, T = a.T, K = a.K, F = a.F;
    // bert_amir5.py:638 for the launch BEFORE this one on the stream: block 0 adds the per-(graph,
    // 64-column group) partial dot products that launch left in ov_in, in a fixed order
    if (a.ov_in && blockIdx.x == 0) reduce_partials(a.ov_in, B * ((F + 63) / 64), B, a.ov_out, reinterpret_cast<float *>(lds));
    int g_tile, n_wgi;
    bool second = false;
    if (a.n_parts == 1) {
        if (!tile_of_block(blockIdx.x, a.g_tiles, a.n_wg, g_tile, n_wgi)) return;
    } else {
        // Two parts: the block ids that share an XCD (id & 7; observed dispatch, speed only) all work on the
        // SAME part -- four XCDs take the W1 tiles, four the W12 tiles -- so that each XCD's 4 MiB L2 holds one
        // weight image for the whole launch.  (Both images side by side, 3.7-5 MB, do not fit: with the column
        // tiles of both parts mixed on every XCD the main loop ran 17 % slower, W streaming from beyond L2.)
        // (Handing the W1 group a few of the W12 row blocks to even out the ~4 % costlier W12 tiles was
        // measured: no gain at 4096 graphs, +12 % time at 512 where it adds a round.)
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        second = xcd >= 4;
        g_tile = (slot / a.n_wg) * 4 + (xcd & 3);
        n_wgi = slot % a.n_wg;
        if (g_tile >= a.g_tiles) return;
    }
    GGCN_TRACE_IDS();
    GGCN_TRACE(3);
    // the part this workgroup's column tiles belong to (workgroup-uniform: scalar selects)
    const LayerPart &lp = a.part[second ? 1 : 0];
    const char *__restrict__ wpack = lp.wpack;

    const int tid = threadIdx.x;
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
        arow[i] = a.X + node * a.ldx;
    }

#if !GGCN_LAB_NO_DMA_STAGE
    // Round 5: whole tiles with aligned rows bring the epilogue's operands to LDS by LDS-DMA (stage_epilogue_operands_dma,
    // fused_common.h: no registers, no wait in front of the main loop) -- block 618.3 -> 608.6 us, a 512-graph shard 83.0 -> 82.1 us
    // in the same process, bit-identical (found on the eight-wavefront experiment, fused_block8.hip, where the register form of the
    // staging stands alone on its CU: 4.7 -> 1.5 us per workgroup)
    const bool dma_stage = SCH == 1 && (a.n_wg * BN == F) && (gt0 + 4 <= B) && a.drop.thr == 0 && (F % 4 == 0) &&
                           ((reinterpret_cast<uintptr_t>(lp.store_gate) | reinterpret_cast<uintptr_t>(lp.pool_gate_a) | reinterpret_cast<uintptr_t>(lp.pool_gate_b) |
                             reinterpret_cast<uintptr_t>(lp.bias) | reinterpret_cast<uintptr_t>(lp.mid)) & 15u) == 0;
    if (dma_stage) stage_epilogue_operands_dma<kLdsBytes>(a, lp, g0, n_wgi, lds, tid);
    else
#else
    constexpr bool dma_stage = false;
#endif
    stage_epilogue_operands<kLdsBytes>(a, lp, g0, n_wgi, lds, tid);   // g0 = gt0: one wavefront row
    f32x16 acc[4][RN];
    GGCN_TRACE(4);
    unsigned long long stamp_c = 0, stamp_w = 0;
    if constexpr (STAMP) {
        stamp_c = __builtin_readcyclecounter();
        stamp_w = __builtin_amdgcn_s_memrealtime();
    }
    if constexpr (SCH == 0)
        bx3::mainloop<float, AVEC, KFULL, !FULLT>(arow, avalid, wpack, K, a.k_steps, wm, nt0, n_tiles_total, lds, acc);
    else {
        float amax;
        // fast shapes (16-byte rows, K % 32 == 0): buffer loads -- the tile's X rows behind one resource (the launcher
        // checked that 128 rows of ldx floats stay below 2 GiB), lane offsets fixed before the loop
        constexpr bool BUF = AVEC && KFULL;
        mx8::BufX<float> bx;
        if constexpr (BUF) {
            int rel[NP];
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int row = stage_row<float>(i);
                rel[i] = avalid[i] ? (row >> 5) * T + (row & 31) : -1;   // padding rows: outside the descriptor, zeros
            }
            bx = mx8::make_bufx<float>(a.X, a.ldx, (int64_t)gt0 * T, (int64_t)B * T, rel, tid);   // the tile's first node is always a real one
        }
        mx8::mainloop<float, AVEC, KFULL, !FULLT, false, BUF>(arow, avalid, wpack, K, a.k_steps / 2, wm, nt0, n_tiles_total, lds, acc, 0, 4,
                                                             &amax, &bx);
        if (dma_stage) dma_range_verdict<kLdsBytes>(amax, wpack, (int64_t)n_tiles_total * (a.k_steps / 2) * mx8::STAGE_PACK_BYTES, lds, tid & 63);
        else fused_range_verdict<kLdsBytes>(amax, wpack, (int64_t)n_tiles_total * (a.k_steps / 2) * mx8::STAGE_PACK_BYTES, lds, true);
    }
    GGCN_TRACE(5);
    if constexpr (STAMP) {
        asm volatile("" :: "v"(acc[3][RN - 1][15]));   // the loop's last MFMA has retired
        const unsigned long long c1 = __builtin_readcyclecounter(), w1 = __builtin_amdgcn_s_memrealtime();
        if (tid == 0 && a.stamps) {
            a.stamps[2 * (size_t)blockIdx.x] = c1 - stamp_c;
            a.stamps[2 * (size_t)blockIdx.x + 1] = w1 - stamp_w;
        }
    }
    if constexpr (((GGCN_LAB_OFF) & 128) != 0) {   // ladder: no epilogue (the accumulators only have to stay live)
        float s = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < RN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += acc[i][j][r];
        if (s == 123.456f && lp.pool_a) lp.pool_a[tid] = s;
        return;
    }
    // `mid` and `out` are workgroup-uniform run-time facts (the W1 / W12 tiles of the block): four straight-line
    // epilogues instead of scalar branches inside one -- a branch per tile ends the basic block, and nothing (the
    // other column tile's split, the next graph's loads) can then be scheduled into the shadow of a tile's MFMA chain
    if (a.drop.thr != 0) {   // training with dropout on the gates: one layer per launch (no mid bias)
        if (lp.out) epilogue<SCH, FULLT, VST, false, true, kLdsBytes, true>(a, lp, acc, g0, nt0, n_tiles_total, lds, tid);
        else epilogue<SCH, FULLT, VST, false, false, kLdsBytes, true>(a, lp, acc, g0, nt0, n_tiles_total, lds, tid);
    } else if (lp.mid) {
        if (lp.out) epilogue<SCH, FULLT, VST, true, true>(a, lp, acc, g0, nt0, n_tiles_total, lds, tid);
        else epilogue<SCH, FULLT, VST, true, false>(a, lp, acc, g0, nt0, n_tiles_total, lds, tid);
    } else {
        if (lp.out) epilogue<SCH, FULLT, VST, false, true>(a, lp, acc, g0, nt0, n_tiles_total, lds, tid);
        else epilogue<SCH, FULLT, VST, false, false>(a, lp, acc, g0, nt0, n_tiles_total, lds, tid);
    }
    GGCN_TRACE(6);
}


// bert_amir5.py:638 after ggcn_block_fused: the per-(graph, 64-column group) partials -> one scalar.
// One workgroup of 1024 threads, 16-byte loads with four independent sums per thread (config 2: 196 KB in
// ~3 us; the 256-thread scalar loop took 67 us), fixed summation order: deterministic.
constexpr int kRedThreads = 1024;
__global__ __launch_bounds__(kRedThreads) void overlap_reduce_kernel(const float *__restrict__ part, int n_part, int B,
                                                                     float *__restrict__ dst)
{
    __shared__ float red[kRedThreads / 64];
    const int tid = threadIdx.x;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    const int n4 = (reinterpret_cast<uintptr_t>(part) & 15u) == 0 ? n_part / 4 : 0;   // float4 pieces
    const float4 *p4 = reinterpret_cast<const float4 *>(part);
    int i = tid;
    for (; i + 3 * kRedThreads < n4; i += 4 * kRedThreads) {
        const float4 a = p4[i], b = p4[i + kRedThreads], c = p4[i + 2 * kRedThreads], d = p4[i + 3 * kRedThreads];
        s0 += (a.x + a.y) + (a.z + a.w);
        s1 += (b.x + b.y) + (b.z + b.w);
        s2 += (c.x + c.y) + (c.z + c.w);
        s3 += (d.x + d.y) + (d.z + d.w);
    }
    for (; i < n4; i += kRedThreads) {
        const float4 a = p4[i];
        s0 += (a.x + a.y) + (a.z + a.w);
    }
    for (int j = 4 * n4 + tid; j < n_part; j += kRedThreads) s1 += part[j];
    float sdot = (s0 + s1) + (s2 + s3);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sdot += __shfl_xor(sdot, d);
    if ((tid & 63) == 0) red[tid >> 6] = sdot;
    __syncthreads();
    if (tid == 0) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < kRedThreads / 64; ++w) t += red[w];
        *dst = t / (float)B;
    }
}

// rowmask from a batched CSR (T <= GGCN_MASK_MAX_T): one thread per node, ceil(T/32) words each
__global__ __launch_bounds__(256) void rowmask_kernel(const int32_t *__restrict__ rowptr,
                                                      const int32_t *__restrict__ colidx, int64_t n, int T,
                                                      uint32_t *__restrict__ rowmask)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t base = i / T * T;
    const int W = (T + 31) >> 5;
    uint32_t m[GGCN_MASK_MAX_T / 32] = {};
    for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) {
        const uint32_t j = (uint32_t)(colidx[e] - base);
#pragma unroll
        for (int w = 0; w < GGCN_MASK_MAX_T / 32; ++w)   // static indices: m stays in registers
            if ((j >> 5) == (uint32_t)w) m[w] |= 1u << (j & 31u);
    }
#pragma unroll
    for (int w = 0; w < GGCN_MASK_MAX_T / 32; ++w)
        if (w < W) rowmask[i * W + w] = m[w];
}

// shared argument checks + launch of layer_fused_kernel for 1 or 2 parts
int launch_fused(const char *who, FusedArgs &a, int precision, hipStream_t st)
{
    if (precision != GGCN_PREC_BF16X3 && precision != GGCN_PREC_F16MX8 && precision != GGCN_PREC_F16MX6)
        return fail(GGCN_EUNSUPPORTED, "%s: precision %d (use bf16x3 or f16mx8)", who, precision);
#ifndef GGCN_WITH_F16MX6
    if (precision == GGCN_PREC_F16MX6)
        return fail(GGCN_EUNSUPPORTED, "%s: f16mx6 is an experiment this library was built without (make F16MX6=1); use f16mx8", who);
#endif
    if (!a.X) return fail(GGCN_EINVAL, "%s: null input pointer", who);
    if (a.T > 32 ? !a.rowmask : !a.graph_ops)
        return fail(GGCN_EINVAL, "%s: graphs of %d nodes need %s", who, a.T,
                    a.T > 32 ? "the row masks" : "the per-graph operand blocks of ggcn_graph_operands");
    if (a.T <= 32 && !aligned16(a.graph_ops)) return fail(GGCN_EINVAL, "%s: graph_ops must be 16-byte aligned", who);
    if (a.T > 128 && a.graph_ops && !aligned16(a.graph_ops)) return fail(GGCN_EINVAL, "%s: the edge-list blocks must be 16-byte aligned", who);
    if (a.B <= 0 || a.T <= 0 || a.K <= 0 || a.F <= 0)
        return fail(GGCN_EINVAL, "%s: B=%d T=%d K=%d F=%d must be positive", who, a.B, a.T, a.K, a.F);
    if (a.T > GGCN_MASK_MAX_T)
        return fail(GGCN_EUNSUPPORTED, "%s: T=%d > %d; use ggcn_linear + ggcn_aggregate", who, a.T, GGCN_MASK_MAX_T);
    if (a.T > 32 && a.n_parts != 1)
        return fail(GGCN_EUNSUPPORTED, "%s: the two-layer form takes graphs of <= 32 nodes (T=%d): one ggcn_layer_fused per layer", who, a.T);
    if (a.ldx < a.K) return fail(GGCN_EINVAL, "%s: ldx < K", who);
    bool vst = true, any_out = false;
    for (int p = 0; p < a.n_parts; ++p) {
        const LayerPart &lp = a.part[p];
        if (!lp.wpack) return fail(GGCN_EINVAL, "%s: null weight image", who);
        if (!aligned16(lp.wpack)) return fail(GGCN_EINVAL, "%s: wpack must be 16-byte aligned", who);
        if (!lp.out && !lp.pool_a && !lp.pool_b) return fail(GGCN_EINVAL, "%s: no output requested", who);
        if (lp.out) {
            if (lp.ldo < a.F) return fail(GGCN_EINVAL, "%s: leading dimension of the output too small", who);
            if ((int64_t)a.T * lp.ldo >= (int64_t)INT32_MAX)
                return fail(GGCN_EUNSUPPORTED, "%s: T*ldo does not fit 32-bit offsets", who);
            any_out = true;
            // (the 16-byte row stores address a graph's rows with 32-bit byte offsets from its first row)
            vst = vst && (a.F % 4 == 0) && (lp.ldo % 4 == 0) && aligned16(lp.out) && (int64_t)a.T * lp.ldo * 4 < ((int64_t)1 << 31);
        }
    }
    vst = vst && any_out;
    // (the fast shapes address a tile's 128 rows with 32-bit byte offsets from its first row: buffer loads)
    const bool avec = (a.K % 4 == 0) && (a.ldx % 4 == 0) && aligned16(a.X) && (int64_t)a.ldx * 4 * 257 < ((int64_t)1 << 31);
    const bool kfull = (a.K % BK == 0);
    a.k_steps = round_up(a.K, BK) / KSTEP;
    a.n_wg = (a.F + BN - 1) / BN;
    if (a.drop.thr != 0 && (a.n_parts != 1 || precision == GGCN_PREC_F16MX6))
        return fail(GGCN_EUNSUPPORTED, "%s: gate dropout is built into the one-launch LAYER (bf16x3 / f16mx8), not the two-layer block", who);
    if ((int64_t)a.B * a.T * a.F >= ((int64_t)1 << 32) && a.drop.thr != 0)
        return fail(GGCN_EUNSUPPORTED, "%s: gate dropout indexes elements with 32 bits (B*T*F = %lld)", who, (long long)a.B * a.T * a.F);
    if (a.T > 32 && precision == GGCN_PREC_F16MX6)
        return fail(GGCN_EUNSUPPORTED, "%s: f16mx6 takes graphs of <= 32 nodes (T=%d); use f16mx8", who, a.T);
    if (a.T > 32) {   // 64-, 128- or 256-row graph slots: layer_fused_wide_kernel
        const int sb = a.T <= 64 ? 2 : a.T <= 128 ? 4 : 8;
        const int gpt = sb == 2 ? 2 : 1;   // graphs per workgroup
        const int64_t gt = ((int64_t)a.B + gpt - 1) / gpt;
        const int64_t gridw = grid_for(gt, a.n_wg);
        if (gridw > (int64_t)INT32_MAX) return fail(GGCN_EUNSUPPORTED, "%s: batch too large", who);
        a.g_tiles = (int)gt;
        const bool fast = avec && kfull;
        if (sb == 8 && !GGCN_LAB_WIDE_SB8) return launch_fused_wide8(who, a, precision, fast, vst, gridw, st);   // 129..256 nodes: eight wavefronts per graph
        return launch_fused_wide(who, a, precision, sb, fast, vst, gridw, st);
    }
    const int64_t g_tiles = ((int64_t)a.B + 4 * WM - 1) / (4 * WM);
    const int64_t grid = a.n_parts == 1 ? grid_for(g_tiles, a.n_wg) : (g_tiles + 3) / 4 * a.n_wg * 8;
    if (grid > (int64_t)INT32_MAX) return fail(GGCN_EUNSUPPORTED, "%s: batch too large", who);
    a.g_tiles = (int)g_tiles;
    const bool fullt = (a.T == 32) && (a.B % (4 * WM) == 0);
    if (precision == GGCN_PREC_F16MX6) {
        if (!(avec && kfull) || (int64_t)128 * a.ldx * 4 >= ((int64_t)1 << 31))
            return fail(GGCN_EUNSUPPORTED, "%s: f16mx6 needs K %% 32 == 0 and 16-byte aligned rows of X (K=%d ldx=%lld); use f16mx8",
                        who, a.K, (long long)a.ldx);
        return launch_fused6(who, a, fullt, vst, grid, st);
    }
#define GGCN_LAUNCH(SC, AV, KF, FT, VS) \
    hipLaunchKernelGGL((layer_fused_kernel<SC, AV, KF, FT, VS>), dim3((unsigned)grid), dim3(kThreads), 0, st, a)
#define GGCN_PICK(SC)                                                                 \
    do {                                                                              \
        if (avec && kfull && fullt && vst) GGCN_LAUNCH(SC, true, true, true, true);   \
        else if (avec && kfull && fullt) GGCN_LAUNCH(SC, true, true, true, false);    \
        else if (avec && kfull && vst) GGCN_LAUNCH(SC, true, true, false, true);      \
        else if (avec && kfull) GGCN_LAUNCH(SC, true, true, false, false);            \
        else if (avec) GGCN_LAUNCH(SC, true, false, false, false);                    \
        else GGCN_LAUNCH(SC, false, false, false, false);                             \
    } while (0)
    if (a.stamps) {   // the diagnostic instantiation exists for the benchmark's shape class only
        if (!(precision == GGCN_PREC_F16MX8 && avec && kfull && fullt && vst))
            return fail(GGCN_EUNSUPPORTED, "%s: the stamped form takes f16mx8, T = 32, B %% 4 == 0, K %% 32 == 0, 16-byte rows", who);
        hipLaunchKernelGGL((layer_fused_kernel<1, true, true, true, true, true>), dim3((unsigned)grid), dim3(kThreads), 0, st, a);
    } else if (precision == GGCN_PREC_F16MX8) GGCN_PICK(1);
    else GGCN_PICK(0);
#undef GGCN_PICK
#undef GGCN_LAUNCH
    return check_launch(who);
}

}  // namespace

GGCN_TRACE_READER

int csr_rowmask(const int32_t *rowptr, const int32_t *colidx, int B, int T, uint32_t *rowmask, hipStream_t st)
{
    if (!rowptr || !colidx || !rowmask) return fail(GGCN_EINVAL, "ggcn_csr_rowmask: null pointer");
    if (B <= 0 || T <= 0) return fail(GGCN_EINVAL, "ggcn_csr_rowmask: B=%d T=%d must be positive", B, T);
    if (T > GGCN_MASK_MAX_T)
        return fail(GGCN_EUNSUPPORTED, "ggcn_csr_rowmask: T=%d > %d (row masks cover graphs of at most %d nodes)", T, GGCN_MASK_MAX_T, GGCN_MASK_MAX_T);
    const int64_t n = (int64_t)B * T;
    hipLaunchKernelGGL(rowmask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, rowptr, colidx, n, T,
                       rowmask);
    return check_launch("ggcn_csr_rowmask");
}

__global__ __launch_bounds__(256) void dropout_mask_kernel(int64_t n, int F, DropSpec d, int sel, float *__restrict__ out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = drop_keep(drop_hash((uint32_t)i, d.seed_lo, d.seed_hi), sel, d.thr, d.scale);
}

int dropout_mask(int64_t rows, int F, float p, uint64_t seed, int sel, float *out, hipStream_t st)
{
    if (!out) return fail(GGCN_EINVAL, "ggcn_dropout_mask: null pointer");
    if (rows <= 0 || F <= 0 || sel < 0 || sel > 2 || !(p >= 0.0f && p < 1.0f))
        return fail(GGCN_EINVAL, "ggcn_dropout_mask: rows=%lld F=%d sel=%d p=%g", (long long)rows, F, sel, (double)p);
    const int64_t n = rows * F;
    if (n >= ((int64_t)1 << 32)) return fail(GGCN_EUNSUPPORTED, "ggcn_dropout_mask: rows*F must fit 32 bits");
    const DropSpec d = make_drop_spec(p, seed, 0, 0, 0);
    hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, F, d, sel, out);
    return check_launch("ggcn_dropout_mask");
}

int graph_operands(const uint32_t *rowmask, int B, int T, void *ops, hipStream_t st)
{
    if (!rowmask || !ops) return fail(GGCN_EINVAL, "ggcn_graph_operands: null pointer");
    if (B <= 0 || T <= 0) return fail(GGCN_EINVAL, "ggcn_graph_operands: B=%d T=%d must be positive", B, T);
    if (T > 32) return fail(GGCN_EUNSUPPORTED, "ggcn_graph_operands: T=%d > 32 (larger graphs are applied from their row masks)", T);
    if (!aligned16(ops)) return fail(GGCN_EINVAL, "ggcn_graph_operands: ops must be 16-byte aligned");
    hipLaunchKernelGGL(graph_operands_kernel, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, st, rowmask, B, T,
                       static_cast<char *>(ops));
    return check_launch("ggcn_graph_operands");
}

int graph_operands2(const uint32_t *rowmask, int B, int T, int plane, void *ops2, hipStream_t st)
{
    if (!rowmask || !ops2) return fail(GGCN_EINVAL, "ggcn_graph_operands2: null pointer");
    if (B <= 0 || T <= 0) return fail(GGCN_EINVAL, "ggcn_graph_operands2: B=%d T=%d must be positive", B, T);
    if (T > 32) return fail(GGCN_EUNSUPPORTED, "ggcn_graph_operands2: T=%d > 32 (the two-layer block takes graphs of <= 32 nodes)", T);
    if (plane != 0 && plane != 1) return fail(GGCN_EINVAL, "ggcn_graph_operands2: plane %d (0 = bf16 pairs, 1 = fp16 pairs)", plane);
    if (!aligned16(ops2)) return fail(GGCN_EINVAL, "ggcn_graph_operands2: ops2 must be 16-byte aligned");
    const dim3 grid((unsigned)((B + 3) / 4));
    if (plane == 1) hipLaunchKernelGGL(graph_operands2_kernel<1>, grid, dim3(256), 0, st, rowmask, B, T, static_cast<char *>(ops2));
    else hipLaunchKernelGGL(graph_operands2_kernel<0>, grid, dim3(256), 0, st, rowmask, B, T, static_cast<char *>(ops2));
    return check_launch("ggcn_graph_operands2");
}

int graph_operands_weighted(const int32_t *rowptr, const int32_t *colidx, const float *vals, int B, int T, int plane, void *ops,
                            int *flag, hipStream_t st)
{
    const char *who = "ggcn_graph_operands_weighted";
    if (!rowptr || !colidx || !ops) return fail(GGCN_EINVAL, "%s: null pointer", who);
    if (B <= 0 || T <= 0) return fail(GGCN_EINVAL, "%s: B=%d T=%d must be positive", who, B, T);
    if (T > 32) return fail(GGCN_EUNSUPPORTED, "%s: T=%d > 32 (weighted graphs of more nodes: ggcn_linear + ggcn_aggregate)", who, T);
    if (plane != 0 && plane != 1) return fail(GGCN_EINVAL, "%s: plane %d (0 = bf16 pairs, 1 = fp16 pairs)", who, plane);
    if (!aligned16(ops)) return fail(GGCN_EINVAL, "%s: the blocks must be 16-byte aligned", who);
    if ((int64_t)B * T >= (int64_t)INT32_MAX) return fail(GGCN_EUNSUPPORTED, "%s: B*T does not fit int32 node ids", who);
    const dim3 grid((unsigned)((B + 3) / 4));
    if (plane == 1) hipLaunchKernelGGL(graph_operands_w_kernel<1>, grid, dim3(256), 0, st, rowptr, colidx, vals, B, T, static_cast<char *>(ops), flag);
    else hipLaunchKernelGGL(graph_operands_w_kernel<0>, grid, dim3(256), 0, st, rowptr, colidx, vals, B, T, static_cast<char *>(ops), flag);
    return check_launch(who);
}

// gcn.py:30-45 with a real-valued adjacency for graphs of <= 32 nodes in ONE launch: the W12 column tiles' form of the block
// (MID epilogue: one split of `hidden`, 6 MFMAs with the hi / lo operand) on ggcn_graph_operands_weighted blocks; zero_mid is the
// all-zero `mid` row that form reads ([F] floats)
int layer_fused_weighted(const float *X, int64_t ldx, const void *wpack, const void *graph_opsw, const float *bias, const float *zero_mid,
                         int B, int T, int K, int F, const float *store_gate, const float *pool_gate_a, const float *pool_gate_b,
                         float *out, int64_t ldo, float *pool_a, float *pool_b, float *overlap_partial, const float *overlap_in,
                         float *overlap_out, int precision, hipStream_t st)
{
    const char *who = "ggcn_layer_fused_weighted";
    if ((overlap_in == nullptr) != (overlap_out == nullptr)) return fail(GGCN_EINVAL, "%s: overlap_in and overlap_out go together", who);
    if (!graph_opsw || !zero_mid) return fail(GGCN_EINVAL, "%s: the weighted operand blocks and the zero row are required", who);
    if (T > 32) return fail(GGCN_EUNSUPPORTED, "%s: T=%d > 32 (use ggcn_linear + ggcn_aggregate)", who, T);
    if (precision == GGCN_PREC_F16MX6) return fail(GGCN_EUNSUPPORTED, "%s: bf16x3 or f16mx8", who);
    if (out && ldo > (int64_t)INT32_MAX) return fail(GGCN_EUNSUPPORTED, "%s: ldo too large", who);
    FusedArgs a = {};
    a.X = X; a.ldx = ldx; a.graph_ops = static_cast<const char *>(graph_opsw); a.graph_ops2 = a.graph_ops;
    a.ov_in = overlap_in; a.ov_out = overlap_out;
    a.B = B; a.T = T; a.K = K; a.F = F; a.n_parts = 1;
    a.part[0] = LayerPart{static_cast<const char *>(wpack), bias, zero_mid, nullptr, store_gate, pool_gate_a, pool_gate_b,
                          out, pool_a, pool_b, overlap_partial, (int)ldo};
    return launch_fused(who, a, precision, st);
}

int layer_fused(const float *X, int64_t ldx, const void *wpack, const uint32_t *rowmask, const void *graph_ops,
                const float *bias, int B, int T, int K, int F, const float *store_gate, const float *pool_gate_a,
                const float *pool_gate_b, float *out, int64_t ldo, float *pool_a, float *pool_b,
                float *overlap_partial, const float *overlap_in, float *overlap_out, int precision, hipStream_t st,
                const DropSpec *drop, const float *bias_pre)
{
    if ((overlap_in == nullptr) != (overlap_out == nullptr))
        return fail(GGCN_EINVAL, "ggcn_layer_fused: overlap_in and overlap_out go together");
    if (out && ldo > (int64_t)INT32_MAX) return fail(GGCN_EUNSUPPORTED, "ggcn_layer_fused: ldo too large");
    FusedArgs a = {};
    a.X = X; a.ldx = ldx; a.rowmask = rowmask; a.graph_ops = static_cast<const char *>(graph_ops);
    a.ov_in = overlap_in; a.ov_out = overlap_out;
    a.B = B; a.T = T; a.K = K; a.F = F; a.n_parts = 1;
    if (drop) a.drop = *drop;
    a.part[0] = LayerPart{static_cast<const char *>(wpack), bias, nullptr, bias_pre, store_gate, pool_gate_a, pool_gate_b,
                          out, pool_a, pool_b, overlap_partial, (int)ldo};
    if (bias_pre && T <= 32)
        return fail(GGCN_EUNSUPPORTED, "ggcn_layer_fused_prebias: graphs of <= 32 nodes fold both layers in ggcn_block_fused (its eval form); bias_pre is for 33..256 nodes");
    if (bias_pre && drop)
        return fail(GGCN_EUNSUPPORTED, "ggcn_layer_fused_prebias: an inference form (no gate dropout)");
    return launch_fused("ggcn_layer_fused", a, precision, st);
}

int block_fused(const float *X, int64_t ldx, const void *wpack1, const void *wpack12, const void *graph_ops,
                const void *graph_ops2, const float *bias1, const float *bias_mid, const float *bias2, int B, int T, int K, int F,
                const float *gate1, const float *gate2, float *gcn1, int64_t ld1, float *x_out, int64_t ld2,
                float *x1, float *y1, float *pool_out, float *overlap_partial, int precision, hipStream_t st,
                unsigned long long *stamps)
{
    if (!gate2) return fail(GGCN_EINVAL, "ggcn_block_fused: gate2 is required");
    // train.py:227 keeps the logits only, and those need `out` alone (bert_amir5.py:640,643): with x1 = y1 = gcn1 =
    // overlap_partial = NULL the W1 column tiles are not launched at all (half of the matrix work) -- the EVAL form
    const bool layer1 = x1 || y1 || gcn1 || overlap_partial;
    if (layer1 && (!x1 || !y1 || !gate1))
        return fail(GGCN_EINVAL, "ggcn_block_fused: layer 1's outputs go together (gate1, x1 and y1; all NULL with gcn1 and overlap_partial = the eval form)");
    if (!x_out && !pool_out) return fail(GGCN_EINVAL, "ggcn_block_fused: neither x nor its pool requested");
    if ((gcn1 && ld1 > (int64_t)INT32_MAX) || (x_out && ld2 > (int64_t)INT32_MAX))
        return fail(GGCN_EUNSUPPORTED, "ggcn_block_fused: leading dimension too large");
    FusedArgs a = {};
    if (!graph_ops2 || !aligned16(graph_ops2))
        return fail(GGCN_EINVAL, "ggcn_block_fused: graph_ops2 (ggcn_graph_operands2 blocks, 16-byte aligned) is required");
    // large batches, every output, f16mx8: the eight-wavefront workgroup that stages a row block's X planes once for a W1 and a W12
    // column slice (fused_block8.hip: bit-identical results, 2 % less time in steady state from 2048 graphs up).
    // GGCN_BLOCK_FORM=4 (read per call) keeps the four-wavefront kernel: A/B timing, tests.
    if (layer1 && !gcn1 && !stamps && precision == GGCN_PREC_F16MX8 && X && wpack1 && wpack12 && bias_mid &&
        block8_takes(X, ldx, B, T, K, F, gate1, gate2, bias1, bias_mid, bias2, graph_ops, graph_ops2, x_out, ld2)) {
        const char *form = getenv("GGCN_BLOCK_FORM");
        if (!(form && form[0] == '4'))
            return lab_block_fused8(X, ldx, wpack1, wpack12, graph_ops, graph_ops2, bias1, bias_mid, bias2, B, T, K, F, gate1, gate2, x_out,
                                    ld2, x1, y1, pool_out, overlap_partial, st, nullptr, kBlock8RowMajor | kBlock8Product);
    }
    a.X = X; a.ldx = ldx; a.graph_ops = static_cast<const char *>(graph_ops); a.graph_ops2 = static_cast<const char *>(graph_ops2);
    a.B = B; a.T = T; a.K = K; a.F = F; a.n_parts = layer1 ? 2 : 1;
    a.stamps = stamps;
    // bert_amir5.py:639-640: x = gate2 * gc2(gcn1), out = max_t x.  A NULL mid bias (gc1 without bias) still
    // needs the second aggregation: a vector of zeros cannot be conjured here, so the caller passes one.
    if (!bias_mid) return fail(GGCN_EINVAL, "ggcn_block_fused: bias_mid (W2^T.b1, zeros when gc1 has no bias) is required");
    const LayerPart second = LayerPart{static_cast<const char *>(wpack12), bias2, bias_mid, nullptr, gate2, gate2, nullptr,
                                       x_out, pool_out, nullptr, nullptr, (int)ld2};
    if (layer1) {
        // bert_amir5.py:626-636: gcn1 (ungated; optional here), x1 = max_t gcn1*gate1, y1 = max_t gcn1*gate2
        a.part[0] = LayerPart{static_cast<const char *>(wpack1), bias1, nullptr, nullptr, nullptr, gate1, gate2,
                              gcn1, x1, y1, overlap_partial, (int)ld1};
        a.part[1] = second;
    } else {
        if (stamps) return fail(GGCN_EUNSUPPORTED, "ggcn_debug_block_fused_stamped: the eval form is not stamped");
        if (!a.graph_ops) a.graph_ops = a.graph_ops2;   // (the W12 tiles read graph_ops2 only; launch_fused insists on a block pointer)
        a.part[0] = second;   // every XCD runs W12 tiles (tile_of_block: the column tiles of a row block share an XCD)
    }
    return launch_fused("ggcn_block_fused", a, precision, st);
}

int overlap_reduce(const float *partials, int B, int F, float *xy, hipStream_t st)
{
    if (!partials || !xy) return fail(GGCN_EINVAL, "ggcn_overlap_reduce: null pointer");
    if (B <= 0 || F <= 0) return fail(GGCN_EINVAL, "ggcn_overlap_reduce: B=%d F=%d must be positive", B, F);
    hipLaunchKernelGGL(overlap_reduce_kernel, dim3(1), dim3(kRedThreads), 0, st, partials, B * ((F + 63) / 64), B, xy);
    return check_launch("ggcn_overlap_reduce");
}

GGCN_RANGE_FLAG_TU(range_flag_fused)

}  // namespace ggcn

