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
#include "f16mx8_core.h"
#include "f16mx6_core.h"
#include "dropout_hash.h"
#include "lab_hooks.h"

namespace ggcn {
namespace {

using namespace bx3;

// what one group of column tiles ("part") computes: part 0 = the layer itself (or layer 1 of the block),
// part 1 = layer 2 of the block through W12
struct LayerPart {
    const char *wpack;          // ggcn_weight_pack image of this part's [K, F] matrix
    const float *bias;          // added after the (last) normalised aggregation, or NULL
    const float *mid;           // NULL: one aggregation.  Else: y = D.A.(D.A.h + mid) + bias
    const float *store_gate;    // [B,F] or NULL (ones)
    const float *pool_gate_a;   // [B,F] or NULL (ones)
    const float *pool_gate_b;
    float *out;                 // [N, ldo] or NULL
    float *pool_a, *pool_b;     // [B,F] or NULL
    float *ov_partial;          // [B, ceil(F/64)] or NULL: sum_f pool_a*pool_b per graph and 64 columns
    int ldo;
};

struct FusedArgs {
    const float *X;
    int64_t ldx;
    const uint32_t *rowmask;    // graphs of 33..256 nodes (layer_fused_wide_kernel)
    const char *graph_ops;      // graphs of <= 32 nodes: ggcn_graph_operands blocks (layer_fused_kernel)
    const float *ov_in;         // partials an EARLIER launch wrote: block 0 reduces them to *ov_out first
    float *ov_out;
    int B, T, K, F;
    int g_tiles, n_wg, n_parts, k_steps;
    LayerPart part[2];
    DropSpec drop;              // training-mode keep masks of the gates (thr = 0: none); one part only
};

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

// the [N,F] output leaves in 16-byte pieces; GGCN_LAB_NT_STORE (lab_hooks.h) makes them non-temporal
__device__ __forceinline__ void store_out4(float *p, const float4 &v)
{
#if GGCN_LAB_NT_STORE
    typedef float f4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(f4{v.x, v.y, v.z, v.w}, reinterpret_cast<f4 *>(p));
#else
    *reinterpret_cast<float4 *>(p) = v;
#endif
}

// lanes 0-31 receive the value of lane + 32 (lanes 32-63: unspecified, their own lower-half partner's value)
__device__ __forceinline__ float upper_half_to_lower(float v)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[1]);   // r[1] = the "src" operand after the swap: its lanes 0-31 hold v of lanes 32-63
}

// 0/1 adjacency block as the MFMA A operand: bit b of `m` (already shifted by 4h) -> element pairs of the two
// k-steps; element j of k-step s is node 16s + 8(j>>2) + 4h + (j&3).  Per dword (two elements = two neighbouring bits):
// both halves of a register hold the 16 mask bits of the k-step, a packed shift brings bit b + 1 / bit b to the top of
// the high / low half, a packed arithmetic shift spreads them (0 or 0xFFFF) and one AND leaves bf16 1.0 = 0x3F80 --
// three VALU per dword (+ one per k-step) where the scalar bit-field form took four.
__device__ __forceinline__ void expand_mask(uint32_t mh, bf16x8 (&af)[2])
{
    typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
    typedef short i16x2 __attribute__((ext_vector_type(2)));
    union { bf16x8 v; uint32_t w[4]; } u[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const uint32_t m16 = s == 0 ? (mh & 0xFFFFu) : (mh >> 16);
        const u16x2 both = __builtin_bit_cast(u16x2, m16 | (m16 << 16));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int b = 8 * (q >> 1) + 2 * (q & 1);
            const u16x2 top = both << u16x2{(unsigned short)(15 - b), (unsigned short)(14 - b)};   // low half: bit b, high half: bit b + 1
            const i16x2 spread = __builtin_bit_cast(i16x2, top) >> i16x2{15, 15};
            u[s].w[q] = __builtin_bit_cast(uint32_t, spread) & 0x3F803F80u;
        }
    }
    af[0] = u[0].v;
    af[1] = u[1].v;
}

// ADJ_g . t for one 32x32 tile: 4 MFMAs (2 planes x 2 k-steps), small plane first
__device__ __forceinline__ f32x16 adj_times(const bf16x8 (&af)[2], const f32x16 &t)
{
    bf16x8 hfrag[2][2];
    split2(t, hfrag);
    f32x16 y;
#pragma unroll
    for (int r = 0; r < 16; ++r) y[r] = 0.0f;
#pragma unroll
    for (int p = 1; p >= 0; --p)
#pragma unroll
        for (int s = 0; s < 2; ++s) y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], hfrag[p][s], y, 0, 0, 0);
    return y;
}


// ---- the epilogue's operands per graph (ggcn_graph_operands; GGCN_GRAPH_OPS_BYTES each) ------------------------------
//   [0, 1024)    adjacency as the A operand of the aggregation MFMA, k-step 0: lane l (row = l & 31, h = l >> 5)
//                -> 16 B at 16 l; 16-bit element j = 0xFFFF where adj[row][node 16s + 8(j>>2) + 4h + (j&3)] != 0
//   [1024, 2048) the same for k-step 1
//   [2048, 2176) 1 / (rowsum(adj) + 1) (gcn.py:35) in accumulator order: float [h][16], entry r = the value of row
//                (r & 3) + 8 (r >> 2) + 4 h
// One AND with the plane type's 1.0 pattern turns the 0xFFFF elements into an exact MFMA operand; nothing about a graph
// is computed per column tile any more (the expansion of the row masks, the IEEE division and the 16 ds_bpermute per
// graph were ~45 VALU + 16 LDS operations per graph and wavefront: a tenth of the epilogue's instructions).
constexpr int kOpsBytes = GGCN_GRAPH_OPS_BYTES;
static_assert(kOpsBytes == 2048 + 128, "layout above");

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

// acc -> two fp16 planes (hi = RNE fp16(v), lo = fp16(v - hi): residual <= 2^-22 |v| + 2^-25, fp16 subnormals are kept
// by the conversions and by the MFMA -- tools/probes/denorm_probe.hip) as B-operand fragments of the two k-steps:
// 1.5 VALU instructions per value (v_cvt_pk_f16_f32 per pair, v_fma_mixlo/hi_f16 per value) against 3 for the bf16 pair
typedef _Float16 f16x8e __attribute__((ext_vector_type(8)));
__device__ __forceinline__ void split2h(const f32x16 &acc, f16x8e (&frag)[2][2])
{
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    union { f16x8e v; uint32_t w[4]; } hi[2], lo[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float v0 = acc[8 * s + 2 * q], v1 = acc[8 * s + 2 * q + 1];
            const h2 p = __builtin_convertvector(f2{v0, v1}, h2);
            uint32_t l;
            asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(p), "v"(v0));
            asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(p), "v"(v1));
            hi[s].w[q] = __builtin_bit_cast(uint32_t, p);
            lo[s].w[q] = l;
        }
    frag[0][0] = hi[0].v; frag[0][1] = hi[1].v;
    frag[1][0] = lo[0].v; frag[1][1] = lo[1].v;
}

// plane type of the aggregation MFMAs: bf16 pairs for bf16x3 (full fp32 range), fp16 pairs for f16mx8 (whose inputs
// are fp16-ranged anyway; a hidden value beyond 65504 becomes inf - inf = NaN in the output, never a silent clamp)
template <int SCH> struct AggPlane;
template <> struct AggPlane<0> {
    typedef bf16x8 frag;
    static constexpr uint32_t kOne = 0x3F803F80u;
    static __device__ __forceinline__ void split(const f32x16 &t, frag (&f)[2][2]) { split2(t, f); }
    static __device__ __forceinline__ f32x16 mma(const frag &a, const frag &b, const f32x16 &c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
};
template <> struct AggPlane<1> {
    typedef f16x8e frag;
    static constexpr uint32_t kOne = 0x3C003C00u;
    static __device__ __forceinline__ void split(const f32x16 &t, frag (&f)[2][2]) { split2h(t, f); }
    static __device__ __forceinline__ f32x16 mma(const frag &a, const frag &b, const f32x16 &c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
};

template <> struct AggPlane<2> : AggPlane<1> {};   // f16mx6: fp16 planes as well

// mean_b sum_f of the per-(graph, 64-column group) partials, in a fixed order (deterministic); one workgroup
__device__ __forceinline__ void reduce_partials(const float *__restrict__ part, int n_part, int B, float *__restrict__ dst,
                                                float *red)
{
    float sdot = 0.0f;
    for (int idx = threadIdx.x; idx < n_part; idx += kThreads) sdot += part[idx];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) sdot += __shfl_xor(sdot, d);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sdot;
    __syncthreads();
    if (threadIdx.x == 0) *dst = ((red[0] + red[1]) + (red[2] + red[3])) / (float)B;
    __syncthreads();
}


// ---- LDS behind the main loop's buffers: what the epilogue reads, fetched BEFORE the main loop ----------------------
// A global load issued inside the epilogue waits 1-2 us on a chip whose memory queues are full (measured: operand
// blocks requested one graph ahead made the layer 10 % slower than expanding the masks in registers), so everything the
// epilogue needs is copied to LDS at kernel start -- the 4 graphs of a workgroup are the same for its 4 wavefronts:
//   [kEpiOps  ]  4 x GGCN_GRAPH_OPS_BYTES   operand blocks of graphs g0 .. g0+3
//   [kEpiGate ]  3 x [4 graphs][256 columns] floats: store gate, pool gate a, pool gate b (1.0 where the gate is NULL)
//   [kEpiBias ]  2 x [256 columns] floats: bias, mid bias (0 where NULL)
template <int BASE>
struct EpiLds {   // byte offsets of the staged operands; the store staging of the epilogue is always [0, 32 KiB)
    static constexpr int kOps = BASE;
    static constexpr int kGate = kOps + 4 * kOpsBytes;
    static constexpr int kBias = kGate + 3 * 4 * BN * 4;
    static constexpr int kEnd = kBias + 2 * BN * 4;
};
constexpr int kEpiLdsBytes = EpiLds<0>::kEnd;   // 8704 + 12288 + 2048 = 23040
static_assert(WM == 1, "one wavefront row: the workgroup's 4 graphs are every wavefront's 4 graphs");
static_assert(mx6::kRaw >= 32768 && EpiLds<mx6::kRaw>::kEnd <= mx6::kLdsBytes6, "f16mx6: the operands go where the RAW stages were");

template <int BASE>
__device__ __forceinline__ void stage_epilogue_operands(const FusedArgs &a, const LayerPart &lp, int g0, int n_wgi, char *lds, int tid)
{
    constexpr int kEpiOps = EpiLds<BASE>::kOps, kEpiGate = EpiLds<BASE>::kGate, kEpiBias = EpiLds<BASE>::kBias;
    const int B = a.B, F = a.F;
    // operand blocks: 544 pieces of 16 B
    uint4 piece[3];
    const int n_pieces = 4 * kOpsBytes / 16;
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const int idx = tid + it * kThreads;
        const int idc = idx < n_pieces ? idx : 0;
        const int gi = (idc * 16) / kOpsBytes;
        // a graph past the batch reads graph g0's bytes instead (never used)
        const int64_t off = (int64_t)g0 * kOpsBytes + (g0 + gi < B ? idc * 16 : idc * 16 - gi * kOpsBytes);
        piece[it] = *reinterpret_cast<const uint4 *>(a.graph_ops + off);
    }
    // gates and biases of this workgroup's 256 columns
    const int col = n_wgi * BN + tid;
    const bool cok = col < F;
    const float *dummy = a.X;
    const float *gp[3] = {lp.store_gate, lp.pool_gate_a, lp.pool_gate_b};
    float gv[3][4];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = gp[k] && cok && g0 + i < B;
            const float v = (gp[k] ? gp[k] : dummy)[ok ? (int64_t)(g0 + i) * F + col : 0];
            gv[k][i] = ok ? v : 1.0f;
        }
    const float vbias = (lp.bias ? lp.bias : dummy)[lp.bias && cok ? col : 0];
    const float vmidb = (lp.mid ? lp.mid : dummy)[lp.mid && cok ? col : 0];
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        const int idx = tid + it * kThreads;
        if (idx < n_pieces) *reinterpret_cast<uint4 *>(lds + kEpiOps + idx * 16) = piece[it];
    }
    float *gl = reinterpret_cast<float *>(lds + kEpiGate);
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int i = 0; i < 4; ++i) gl[(k * 4 + i) * BN + tid] = gv[k][i];
    float *bl = reinterpret_cast<float *>(lds + kEpiBias);
    bl[tid] = lp.bias && cok ? vbias : 0.0f;
    bl[BN + tid] = lp.mid && cok ? vmidb : 0.0f;
    // visible to every wavefront after the main loop's first barrier
}

// ---- the epilogue of one wavefront: its 4 graphs x RN column tiles ------------------------------------------------
// MID: the block's second layer through W12 (two aggregations with the `mid` bias in between); OUT: the [N,F] output is
// stored.  Per graph: both column tiles are split, multiplied by the adjacency and finished side by side, so that one
// tile's element-wise work issues under the other's MFMA chain.
template <int SCH, bool FULLT, bool VST, bool MID, bool OUT, int BASE = kLdsBytes, bool DROP = false>
__device__ __forceinline__ void epilogue(const FusedArgs &a, const LayerPart &lp, f32x16 (&acc)[4][RN], int g0, int nt0,
                                         int n_tiles_total, char *lds, int tid)
{
    constexpr int kEpiOps = EpiLds<BASE>::kOps, kEpiGate = EpiLds<BASE>::kGate, kEpiBias = EpiLds<BASE>::kBias;
    using P = AggPlane<SCH>;
    typedef typename P::frag frag;
    const int B = a.B, T = a.T, F = a.F;
    float *__restrict__ out = lp.out, *__restrict__ pool_a = lp.pool_a, *__restrict__ pool_b = lp.pool_b;
    float *__restrict__ ov_partial = lp.ov_partial;
    const int ldo = lp.ldo;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;

    // operands staged in LDS before the main loop (stage_epilogue_operands): adjacency fragments, reciprocal denominators,
    // gates, biases
    const char *ops_lds = lds + kEpiOps;
    const float *gate_lds = reinterpret_cast<const float *>(lds + kEpiGate);
    const float *bias_lds = reinterpret_cast<const float *>(lds + kEpiBias);
    const int wn = wave % WN;
    float vb[RN], vmid[RN];
    bool col_ok[RN];
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        col_ok[j] = (nt0 + j) * NT + c < F;
        vb[j] = bias_lds[wn * (RN * NT) + j * NT + c];
        vmid[j] = bias_lds[BN + wn * (RN * NT) + j * NT + c];
    }
    const int lane_off = 4 * h * ldo + c;  // this lane's element inside a (graph, column tile) block
    // the A buffers are free after the main loop's last barrier: 8 KiB per wavefront = 32 rows x 64 columns
    float *stage_lds = reinterpret_cast<float *>(lds) + wave * (32 * 64);
    constexpr bool vst = VST && OUT;
    constexpr bool direct_store = !VST && OUT;

#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int g = g0 + i;
        if (!FULLT && g >= B) break;  // workgroup-uniform
        // adjacency fragments of graph g: the stored 0xFFFF elements become the plane type's 1.0
        frag afv[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const uint4 raw = *reinterpret_cast<const uint4 *>(ops_lds + i * kOpsBytes + s * 1024 + lane * 16);
            union { frag v; uint32_t w[4]; } u;
            u.w[0] = raw.x & P::kOne; u.w[1] = raw.y & P::kOne;
            u.w[2] = raw.z & P::kOne; u.w[3] = raw.w & P::kOne;
            afv[s] = u.v;
        }
        float4 rv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) rv[q] = *reinterpret_cast<const float4 *>(ops_lds + i * kOpsBytes + 2048 + h * 64 + q * 16);
        float vsg[RN], vga[RN], vgb[RN];
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            const int at = i * BN + wn * (RN * NT) + j * NT + c;
            vsg[j] = gate_lds[at];
            vga[j] = gate_lds[4 * BN + at];
            vgb[j] = gate_lds[8 * BN + at];
        }
        bool tile_ok[RN];
#pragma unroll
        for (int j = 0; j < RN; ++j) tile_ok[j] = nt0 + j < n_tiles_total;   // wavefront-uniform: column tile past F

        // gcn.py:41 (layer 1 / the layer): agg = ADJ_g . hidden_g, small plane first; the column tiles' chains are
        // issued one behind the other, so that a tile's split and element-wise work sit under the other's MFMAs
        f32x16 y[RN];
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            frag hf[2][2];
            P::split(acc[i][j], hf);
#pragma unroll
            for (int r = 0; r < 16; ++r) y[j][r] = 0.0f;
#pragma unroll
            for (int p = 1; p >= 0; --p)
#pragma unroll
                for (int s = 0; s < 2; ++s) { if constexpr (!((GGCN_LAB_EPI) & 2)) y[j] = P::mma(afv[s], hf[p][s], y[j]); else y[j][s] += (float)hf[p][s][0]; }
        }
        float rinv[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            rinv[4 * q] = rv[q].x; rinv[4 * q + 1] = rv[q].y; rinv[4 * q + 2] = rv[q].z; rinv[4 * q + 3] = rv[q].w;
        }
        if constexpr (MID) {   // the block's second layer through W12 = W1.W2 (header): D.A.(X.W12) + c, then gcn.py:41 again
#pragma unroll
            for (int j = 0; j < RN; ++j) {
                f32x16 u;
#pragma unroll
                for (int r = 0; r < 16; ++r) u[r] = y[j][r] * rinv[r] + vmid[j];
                frag hf[2][2];
                P::split(u, hf);
#pragma unroll
                for (int r = 0; r < 16; ++r) y[j][r] = 0.0f;
#pragma unroll
                for (int p = 1; p >= 0; --p)
#pragma unroll
                    for (int s = 0; s < 2; ++s) y[j] = P::mma(afv[s], hf[p][s], y[j]);
            }
        }
        float dot = 0.0f;  // this wavefront's share of sum_f x1[g,f] * y1[g,f] (bert_amir5.py:638)
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            if (!tile_ok[j]) break;
            const int gn = (nt0 + j) * NT + c;
            // a gate is constant over the rows of a graph and rounding is monotonic, so
            // max_t fl(y_t * g) == fl(g * max_t y_t) for g >= 0 (and g * min_t y_t for g < 0):
            // track max and min of y once, apply both pool gates at the end (bert_amir5.py:635-640).
            // DROP (training, bert_amir5.py:621-625): every (token, feature) has its own keep factor per gate stream, so
            // the gated values themselves are maximised.
            float vmax = -INFINITY, vmin = INFINITY, pmax_a = -INFINITY, pmax_b = -INFINITY;
            float *tile = OUT ? out + ((int64_t)g * T) * ldo + (nt0 + j) * NT : nullptr;  // wave-uniform
            const float sg = vsg[j];
            const float bj = vb[j];
            const uint32_t didx0 = DROP ? (uint32_t)(((int64_t)g * T + 4 * h) * F + gn) : 0u;   // element of row 4h; rows add row0 * F
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row0 = (r & 3) + 8 * (r >> 2);  // this lane's row is row0 + 4h
                const float v = y[j][r] * rinv[r] + bj;   // gcn.py:41,43
                float vs = v * sg;
                if constexpr (DROP) {
                    const uint32_t hh = drop_hash(didx0 + (uint32_t)(row0 * F), a.drop.seed_lo, a.drop.seed_hi);
                    vs *= drop_keep(hh, a.drop.sel[0], a.drop.thr, a.drop.scale);
                    if (FULLT || row0 + 4 * h < T) {
                        pmax_a = fmaxf(pmax_a, v * vga[j] * drop_keep(hh, a.drop.sel[1], a.drop.thr, a.drop.scale));
                        pmax_b = fmaxf(pmax_b, v * vgb[j] * drop_keep(hh, a.drop.sel[2], a.drop.thr, a.drop.scale));
                    }
                }
                if constexpr (vst) {
                    // staged for the 16-byte row stores below; columns of the rows with bit 2 set are
                    // swapped between the two 32-column halves so that h = 0 / 1 hit different banks
                    stage_lds[(row0 + 4 * h) * 64 + ((32 * j + c) ^ (32 * h))] = vs;
                }
                if (FULLT || row0 + 4 * h < T) {
                    if (direct_store && col_ok[j]) tile[lane_off + row0 * ldo] = vs;  // bert_amir5.py:626 / :639
                    if constexpr (!DROP) {
                        vmax = fmaxf(vmax, v);
                        vmin = fminf(vmin, v);
                    }
                }
            }
            // the other lane half's value: v_permlane32_swap (one VALU instruction; __shfl_xor(.., 32) is a
            // ds_bpermute, an LDS round trip in front of the pooled stores).  Only lanes 0-31 use the result.
            float pa, pb;
            if constexpr (DROP) {
                pa = fmaxf(pmax_a, upper_half_to_lower(pmax_a));
                pb = fmaxf(pmax_b, upper_half_to_lower(pmax_b));
            } else {
                vmax = fmaxf(vmax, upper_half_to_lower(vmax));
                vmin = fminf(vmin, upper_half_to_lower(vmin));
                const float ga = vga[j], gb = vgb[j];
                pa = ga * (ga >= 0.0f ? vmax : vmin);
                pb = gb * (gb >= 0.0f ? vmax : vmin);
            }
            if (h == 0 && col_ok[j]) {
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
        if constexpr (vst) {
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
                if ((FULLT || row < T) && gcol < F && !((GGCN_LAB_EPI) & 1)) store_out4(gbase + row * ldo, v4);   // (GGCN_LAB_EPI 1: timing build without the stores)
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (i == 0) GGCN_TRACE(7);
    }
}

// SCH: 0 = bf16x3 main loop, 1 = f16mx8 (f16mx8_core.h)
// FULLT: T == 32 and B % 4 == 0 (every row of every tile is a real node): drops every guard.
// VST: the [N,F] output leaves through LDS as 16-byte row stores (needs F, ldo multiples of 4 and a 16-byte aligned out)
template <int SCH, bool AVEC, bool KFULL, bool FULLT, bool VST>
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

    stage_epilogue_operands<kLdsBytes>(a, lp, g0, n_wgi, lds, tid);   // g0 = gt0: one wavefront row
    f32x16 acc[4][RN];
    GGCN_TRACE(4);
    if constexpr (SCH == 0)
        bx3::mainloop<float, AVEC, KFULL, !FULLT>(arow, avalid, wpack, K, a.k_steps, wm, nt0, n_tiles_total, lds, acc);
    else
        mx8::mainloop<float, AVEC, KFULL, !FULLT>(arow, avalid, wpack, K, a.k_steps / 2, wm, nt0, n_tiles_total, lds, acc);
    GGCN_TRACE(5);
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


// ---- the same layer / block with the f16mx6 main loop (f16mx6_core.h): 96 instead of 128 matrix-pipe cycles per 32^3 block ----
// Fast path only: K % 32 == 0, 16-byte aligned rows of X (LDS-DMA moves 16 bytes per lane).  LDS: PLANE + RAW =
// 64 KiB, two workgroups per CU; the epilogue's operands are fetched into the RAW stages once the main loop has released them.
template <bool FULLT, bool VST>
__global__ __launch_bounds__(kThreads, kWavesPerSimd) void layer_fused6_kernel(const FusedArgs a)
{
    __shared__ __attribute__((aligned(16))) char lds[mx6::kLdsBytes6];
    const int B = a.B, T = a.T, K = a.K, F = a.F;
    if (a.ov_in && blockIdx.x == 0) reduce_partials(a.ov_in, B * ((F + 63) / 64), B, a.ov_out, reinterpret_cast<float *>(lds));
    int g_tile, n_wgi;
    bool second = false;
    if (a.n_parts == 1) {
        if (!tile_of_block(blockIdx.x, a.g_tiles, a.n_wg, g_tile, n_wgi)) return;
    } else {   // four XCDs take the W1 tiles, four the W12 tiles (layer_fused_kernel)
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        second = xcd >= 4;
        g_tile = (slot / a.n_wg) * 4 + (xcd & 3);
        n_wgi = slot % a.n_wg;
        if (g_tile >= a.g_tiles) return;
    }
    const LayerPart &lp = a.part[second ? 1 : 0];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g0 = g_tile * 4;
    const int n_tiles_total = (F + NT - 1) / NT;
    const int nt0 = n_wgi * (BN / NT) + wave * RN;

    // DMA sources: piece j of this wavefront = tile rows 32 wave + 8 j + (lane >> 3), chunk (lane & 7) ^ swizzle(row)
    const float *xtile = a.X + (int64_t)g0 * T * a.ldx;   // workgroup-uniform; 128 rows x ldx floats stay below 4 GiB (launcher)
    uint32_t aoff[2];   // pieces 2, 3 = pieces 0, 1 sixteen rows further (same swizzle): a uniform stride in the FULLT build
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int row = 32 * wave + 8 * j + (lane >> 3);
        const int g = g0 + (row >> 5), r = row & 31;
        const bool ok = (g < B) && (FULLT || r < T);
        const int node = ok ? (row >> 5) * T + r : 0;       // inside the tile; clamped: the split zeroes such a block
        aoff[j] = (uint32_t)(((int64_t)node * a.ldx + 4 * ((lane & 7) ^ ((row >> 1) & 7))) * 4);
    }
    const int urow = 32 * wave + (lane >> 1);   // the row whose half block this lane splits
    const bool uvalid = (g0 + (urow >> 5) < B) && (FULLT || (urow & 31) < T);

    f32x16 acc[4][RN];
    const int64_t rows_left = (int64_t)B * T - (int64_t)g0 * T;
    const uint32_t xtile_bytes = (uint32_t)(((rows_left < 128 ? rows_left : 128) - 1) * a.ldx * 4 + (int64_t)K * 4);
    mx6::mainloop<!FULLT>(xtile, xtile_bytes, aoff, (uint32_t)(16 * a.ldx * 4), uvalid, lp.wpack, K, a.k_steps / 2, nt0, n_tiles_total, lds, acc);
    if constexpr (((GGCN_LAB_OFF) & 128) != 0) {   // ladder: no epilogue
        float sacc = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < RN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += acc[i][j][r];
        if (sacc == 123.456f && lp.pool_a) lp.pool_a[tid] = sacc;
        return;
    }
    // everything the epilogue derives from the thread id is derived AFTER the loop (the asm makes the id opaque): hipcc
    // otherwise computes those per-lane offsets and pointers up front and keeps ~25 registers alive across a loop that
    // has none to spare
    int tid_e = threadIdx.x;
    asm volatile("" : "+v"(tid_e));
    stage_epilogue_operands<mx6::kRaw>(a, lp, g0, n_wgi, lds, tid_e);   // the RAW stages are free: the loop ended on a barrier
    __syncthreads();
    if (lp.mid) {
        if (lp.out) epilogue<2, FULLT, VST, true, true, mx6::kRaw>(a, lp, acc, g0, nt0, n_tiles_total, lds, tid_e);
        else epilogue<2, FULLT, VST, true, false, mx6::kRaw>(a, lp, acc, g0, nt0, n_tiles_total, lds, tid_e);
    } else {
        if (lp.out) epilogue<2, FULLT, VST, false, true, mx6::kRaw>(a, lp, acc, g0, nt0, n_tiles_total, lds, tid_e);
        else epilogue<2, FULLT, VST, false, false, mx6::kRaw>(a, lp, acc, g0, nt0, n_tiles_total, lds, tid_e);
    }
}

// ---- graphs of 33..256 nodes (LitBank: ORI_ML = 100, constant.py:227; ACE cased: ORI_ML = 231, constant.py:267)
// in the same one launch per layer ----
// A graph occupies SB = 2 or 4 consecutive 32-row blocks of a wavefront's 128-row tile (64- or 128-row slot;
// T in 65..96 takes the 128-row slot), its adjacency is SB x SB blocks of 32 x 32 bits (row masks of
// ceil(T/32) words), and the neighbour sum of output block io is
//     agg[io] = sum_ii ADJ[io][ii] . hidden[ii]          (SB x 4 MFMAs per 32 x 32 output tile)
// with every hidden[ii] taken from the accumulator tiles as in the 32-node kernel.  All accumulators are first
// split into their two bf16 planes IN PLACE (same register count), then each output block is produced,
// normalised, gated, pooled and stored.  One part only (the two-layer block form stays with T <= 32).
// SB = 8 (T in 129..256), the FIRST form of the 256-row slot, kept behind GGCN_LAB_WIDE_SB8 for comparison: the workgroup
// runs the main loop TWICE (rows 0-127, then 128-255 of its graph) and keeps the first half's planes in registers
// meanwhile -- 256 registers of planes in the epilogue, so it is built for one wavefront per SIMD and the main loop runs
// at its lone-wavefront speed: measured 4-18 % SLOWER than linear + aggregate at T = 231 (tools/wide_timing.py).  The
// launcher takes layer_fused_wide8_kernel (below: eight wavefronts, both halves in flight) for these graphs.
template <int SCH, bool AVEC, bool KFULL, bool VST, int SB>
__global__ __launch_bounds__(kThreads, SB == 8 ? 1 : kWavesPerSimd) void layer_fused_wide_kernel(const FusedArgs a)
{
    static_assert(SB == 2 || SB == 4 || SB == 8, "a graph slot is 64, 128 or 256 rows");
    __shared__ __attribute__((aligned(16))) char lds[kLdsBytes];
    const int B = a.B, T = a.T, K = a.K, F = a.F;
    if (a.ov_in && blockIdx.x == 0) reduce_partials(a.ov_in, B * ((F + 63) / 64), B, a.ov_out, reinterpret_cast<float *>(lds));
    int g_tile, n_wgi;
    if (!tile_of_block(blockIdx.x, a.g_tiles, a.n_wg, g_tile, n_wgi)) return;
    const LayerPart &lp = a.part[0];
    const float *__restrict__ bias = lp.bias, *__restrict__ store_gate = lp.store_gate;
    const float *__restrict__ pool_gate_a = lp.pool_gate_a, *__restrict__ pool_gate_b = lp.pool_gate_b;
    float *__restrict__ out = lp.out, *__restrict__ pool_a = lp.pool_a, *__restrict__ pool_b = lp.pool_b;
    float *__restrict__ ov_partial = lp.ov_partial;
    const int ldo = lp.ldo;
    constexpr int S = 32 * SB;                     // rows per graph slot
    constexpr int GPT = SB >= 4 ? 1 : 4 / SB;      // graphs per workgroup
    constexpr int HALVES = SB == 8 ? 2 : 1;        // 128-row passes through the main loop

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    static_assert(WM == 1, "the wide-graph kernel is written for one wavefront row per workgroup");
    const int g0 = g_tile * GPT;
    const int n_tiles_total = (F + NT - 1) / NT;
    const int nt0 = n_wgi * (BN / NT) + wn * RN;
    const int W = (T + 31) >> 5;

    constexpr int NP = Geom<float>::NP;
    // every accumulator tile -> its two bf16 planes (B-operand fragments of the aggregation MFMAs), in place
    bf16x8 hf[4 * HALVES][RN][2][2];
#pragma unroll
    for (int hh = 0; hh < HALVES; ++hh) {
        const float *arow[NP];
        bool avalid[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int row = stage_row<float>(i) + 128 * hh;
            const int g = g0 + row / S, r = row % S;
            avalid[i] = (g < B) && (r < T);
            const int64_t node = avalid[i] ? (int64_t)g * T + r : 0;
            arow[i] = a.X + node * a.ldx;
        }
        f32x16 acc[4][RN];
        if constexpr (SCH == 0)
            bx3::mainloop<float, AVEC, KFULL, true>(arow, avalid, lp.wpack, K, a.k_steps, wm, nt0, n_tiles_total, lds, acc);
        else
            mx8::mainloop<float, AVEC, KFULL, true>(arow, avalid, lp.wpack, K, a.k_steps / 2, wm, nt0, n_tiles_total, lds, acc);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < RN; ++j) split2(acc[i][j], hf[4 * hh + i][j]);
    }

    const int c = lane & 31, h = lane >> 5;
    float vb[RN], vsg[GPT][RN], vga[GPT][RN], vgb[GPT][RN];
    bool col_ok[RN];
    {
        const float *dummy = a.X;
        const float *pb = bias ? bias : dummy, *psg = store_gate ? store_gate : dummy;
        const float *pga = pool_gate_a ? pool_gate_a : dummy, *pgb = pool_gate_b ? pool_gate_b : dummy;
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            const int gn = (nt0 + j) * NT + c;
            col_ok[j] = gn < F;
            const int gnc = col_ok[j] ? gn : 0;
            vb[j] = pb[bias ? gnc : 0];
#pragma unroll
            for (int s = 0; s < GPT; ++s) {
                const int64_t at = (int64_t)(g0 + s < B ? g0 + s : 0) * F + gnc;
                vsg[s][j] = psg[store_gate ? at : 0];
                vga[s][j] = pga[pool_gate_a ? at : 0];
                vgb[s][j] = pgb[pool_gate_b ? at : 0];
            }
        }
    }
    float *stage_lds = reinterpret_cast<float *>(lds) + wave * (32 * 64);
    const int perm_base = 16 * h;
    const int lane_off = 4 * h * ldo + c;
    auto graphs = [&](auto has_out) {
        constexpr bool vst = VST && decltype(has_out)::value;
        constexpr bool direct_store = !VST && decltype(has_out)::value;
#pragma unroll
        for (int s = 0; s < GPT; ++s) {
            const int g = g0 + s;
            if (g >= B) break;  // workgroup-uniform
            // this lane's adjacency rows: node 32*io + (lane & 31), word ii; SB <= 4: for the whole graph (one
            // latency), SB = 8: one output block ahead (64 words would not stay in registers)
            constexpr int MB = SB == 8 ? 2 : SB;
            uint32_t mw[MB][SB];
            auto load_masks = [&](int io, uint32_t (&m)[SB]) {
                const int node = 32 * io + c;
                const bool ok = node < T;
#pragma unroll
                for (int ii = 0; ii < SB; ++ii) {
                    const bool okw = ok && ii < W;
                    const uint32_t v = a.rowmask[okw ? ((int64_t)g * T + node) * W + ii : 0];
                    m[ii] = okw ? v : 0u;
                }
            };
            if constexpr (SB == 8) {
                load_masks(0, mw[0]);
            } else {
#pragma unroll
                for (int io = 0; io < SB; ++io) load_masks(io, mw[io]);
            }
            float vmax[RN], vmin[RN];
#pragma unroll
            for (int j = 0; j < RN; ++j) { vmax[j] = -INFINITY; vmin[j] = INFINITY; }
#pragma unroll
            for (int io = 0; io < SB; ++io) {
                const int node0 = 32 * io;
                if (node0 >= T) break;  // workgroup-uniform: block of padding rows
                const int mi = SB == 8 ? (io & 1) : io;
                if constexpr (SB == 8)
                    if (io + 1 < SB) load_masks(io + 1, mw[(io + 1) & 1]);   // rows past T read as zeros
                int deg = 0;
                bf16x8 af[SB][2];
#pragma unroll
                for (int ii = 0; ii < SB; ++ii) {
                    deg += __popc(mw[mi][ii]);
                    expand_mask(mw[mi][ii] >> (4 * h), af[ii]);
                }
                const float inv = 1.0f / (float)(deg + 1);                  // gcn.py:35
                float rinv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row0 = (r & 3) + 8 * (r >> 2);
                    rinv[r] = __int_as_float(__builtin_amdgcn_ds_bpermute(perm_base + 4 * row0, __float_as_int(inv)));
                }
#pragma unroll
                for (int j = 0; j < RN; ++j) {
                    if (nt0 + j >= n_tiles_total) break;  // wavefront-uniform: column tile past F
                    f32x16 y;
#pragma unroll
                    for (int r = 0; r < 16; ++r) y[r] = 0.0f;
#pragma unroll
                    for (int p = 1; p >= 0; --p)  // small plane first
#pragma unroll
                        for (int ii = 0; ii < SB; ++ii)
#pragma unroll
                            for (int ks = 0; ks < 2; ++ks)
                                y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ii][ks], hf[s * SB + ii][j][p][ks], y, 0, 0, 0);   // gcn.py:41
                    float *tile = decltype(has_out)::value ? out + ((int64_t)g * T + node0) * ldo + (nt0 + j) * NT : nullptr;
                    const float sg = store_gate ? vsg[s][j] : 1.0f;
                    const float bj = bias ? vb[j] : 0.0f;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row0 = (r & 3) + 8 * (r >> 2);  // this lane's row is row0 + 4h
                        const float v = y[r] * rinv[r] + bj;      // gcn.py:41,43
                        if (vst) stage_lds[(row0 + 4 * h) * 64 + ((32 * j + c) ^ (32 * h))] = v * sg;
                        if (node0 + row0 + 4 * h < T) {
                            if (direct_store && col_ok[j]) tile[lane_off + row0 * ldo] = v * sg;
                            vmax[j] = fmaxf(vmax[j], v);
                            vmin[j] = fminf(vmin[j], v);
                        }
                    }
                }
                if (vst) {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const int colq = (lane & 15) * 4;
                    const int gcol = nt0 * NT + colq;
                    float *gbase = out + ((int64_t)g * T + node0) * ldo + gcol;
#pragma unroll
                    for (int it = 0; it < 8; ++it) {
                        const int row = 4 * it + (lane >> 4);
                        const float4 v4 = *reinterpret_cast<const float4 *>(&stage_lds[row * 64 + (colq ^ (32 * ((row >> 2) & 1)))]);
                        if (node0 + row < T && gcol < F) store_out4(gbase + row * ldo, v4);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
            // pools of graph g: max over ALL its rows (bert_amir5.py:635-640), both gates from max and min of y
            float dot = 0.0f;
#pragma unroll
            for (int j = 0; j < RN; ++j) {
                if (nt0 + j >= n_tiles_total) break;
                const float mx = fmaxf(vmax[j], upper_half_to_lower(vmax[j]));
                const float mn = fminf(vmin[j], upper_half_to_lower(vmin[j]));
                if (h == 0 && col_ok[j]) {
                    const int gn = (nt0 + j) * NT + c;
                    const float ga = pool_gate_a ? vga[s][j] : 1.0f, gb = pool_gate_b ? vgb[s][j] : 1.0f;
                    const float pa = ga * (ga >= 0.0f ? mx : mn), pb = gb * (gb >= 0.0f ? mx : mn);
                    if (pool_a) pool_a[(int64_t)g * F + gn] = pa;
                    if (pool_b) pool_b[(int64_t)g * F + gn] = pb;
                    dot = fmaf(pa, pb, dot);
                }
            }
            if (ov_partial && nt0 < n_tiles_total) {
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) dot += __shfl_xor(dot, d);
                if (lane == 0) ov_partial[(int64_t)g * ((F + 63) / 64) + (nt0 >> 1)] = dot;
            }
        }
    };
    if (out) graphs(std::true_type{});
    else graphs(std::false_type{});
}

// plain v_max / v_min (fmaxf first quiets a possible signalling NaN of its operands: an extra instruction per value)
__device__ __forceinline__ float vmaxf_raw(float x, float y) { float d; asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y)); return d; }
__device__ __forceinline__ float vminf_raw(float x, float y) { float d; asm("v_min_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y)); return d; }

// ---- graphs of 129..256 nodes, two wavefronts per SIMD: EIGHT wavefronts per (graph, 256 columns) -----------------------
// Wavefront (rg, cg) = rows 128 rg .. + 127 of the graph's 256-row slot x columns 64 cg .. + 63: the two row groups run
// the SAME main loop side by side on their own stage buffers (thread ids taken mod 256 inside the loop), so the loop keeps
// the two-wavefronts-per-SIMD speed of the 32-node kernel instead of the lone wavefront of the SB = 8 form above, and no
// half of the graph waits in registers.  A neighbour sum needs hidden rows of BOTH row groups, so the accumulators go to
// LDS per 32-column tile -- as an fp32 tile [256 rows][32 columns] per column group, 128 KiB over the dead stage buffers --
// and the sums run over per-row EDGE LISTS (made once per workgroup from the row masks) with 8 lanes x 16 B per row, 8
// source rows in flight: exact fp32 sums, ~5 LDS reads + 20 adds per row of a parse.  160 KiB of LDS, one workgroup per CU.
// Measured (tools/wide_timing.py, f16mx8, 512 x 231 x 768): 399 us against 498 us for linear + aggregate (507 us for the
// lone-wavefront form); 228 us of it is the main loop (timing build without the epilogue), the rest runs under nothing:
// with one workgroup per CU the phases are serial.  The FIRST form of this epilogue (GGCN_LAB_WIDE8_DENSE: bf16 plane
// fragments exchanged through LDS, dense 32 x 32 adjacency blocks on the MFMAs, mask words expanded into operands, empty
// blocks skipped) took 456 us: 8 x 4 MFMAs and ~100 VALU of expansion per block against a handful of edges per row.
constexpr int kW8Threads = 512;
constexpr int kW8Ex = 8 * 16 * 1024;          // per wavefront: 4 row blocks x (2 planes x 2 k-steps) x 1 KiB
constexpr int kW8Cap = 16;                   // source ids per row kept in LDS (rows with more neighbours walk their mask words)
constexpr int kW8Stage = 4096;                // per wavefront: 32 rows x 32 columns of output on their way to 16-byte stores
constexpr int kW8Lds = kW8Ex + 8 * kW8Stage;  // 160 KiB
static_assert(2 * kLdsBytes <= kW8Ex && kW8Lds <= 160 * 1024, "the stage buffers of both row groups lie under the exchange area");

template <int SCH, bool AVEC, bool KFULL, bool VST>
__global__ __launch_bounds__(kW8Threads, 2) void layer_fused_wide8_kernel(const FusedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char lds8[];
    const int B = a.B, T = a.T, K = a.K, F = a.F;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (a.ov_in && blockIdx.x == 0) {   // reduce_partials with the first four wavefronts summing (same order, same result)
        float *red = reinterpret_cast<float *>(lds8);
        const int n_part = B * ((F + 63) / 64);
        float sdot = 0.0f;
        if (tid < kThreads)
            for (int idx = tid; idx < n_part; idx += kThreads) sdot += a.ov_in[idx];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) sdot += __shfl_xor(sdot, d);
        if (lane == 0 && wave < 4) red[wave] = sdot;
        __syncthreads();
        if (tid == 0) *a.ov_out = ((red[0] + red[1]) + (red[2] + red[3])) / (float)B;
        __syncthreads();
    }
    int g, n_wgi;
    if (!tile_of_block(blockIdx.x, a.g_tiles, a.n_wg, g, n_wgi)) return;   // one graph per workgroup: g_tiles = B
    const LayerPart &lp = a.part[0];
    const float *__restrict__ bias = lp.bias, *__restrict__ store_gate = lp.store_gate;
    const float *__restrict__ pool_gate_a = lp.pool_gate_a, *__restrict__ pool_gate_b = lp.pool_gate_b;
    float *__restrict__ out = lp.out, *__restrict__ pool_a = lp.pool_a, *__restrict__ pool_b = lp.pool_b;
    const int ldo = lp.ldo;
    const int rg = wave >> 2, cg = wave & 3;
    const int n_tiles_total = (F + NT - 1) / NT;
    const int nt0 = n_wgi * (BN / NT) + cg * RN;
    const int W = (T + 31) >> 5;
    static_assert(WM == 1 && RN == 2, "written for 128 x 64 wavefront tiles");

    // ---- hidden = X . W for both row groups at once ----
    constexpr int NP = Geom<float>::NP;
    f32x16 acc[4][RN];
    {
        const float *arow[NP];
        bool avalid[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int r = stage_row<float>(i) + 128 * rg;
            avalid[i] = r < T;
            arow[i] = a.X + ((int64_t)g * T + (avalid[i] ? r : 0)) * a.ldx;
        }
        char *stage = lds8 + rg * kLdsBytes;
        // 32-row blocks of this row group that hold nodes: T = 160 leaves the second group one block of four -- its other MFMAs
        // are skipped, and the SIMD it shares with a first-group wavefront gets through a stage that much sooner
        const int rows_here = T - 128 * rg;
        const int nblk = rows_here >= 128 ? 4 : rows_here <= 0 ? 0 : (rows_here + 31) >> 5;
        if constexpr (SCH == 0)
            bx3::mainloop<float, AVEC, KFULL, true, true>(arow, avalid, lp.wpack, K, a.k_steps, 0, nt0, n_tiles_total, stage, acc, nblk);
        else
            mx8::mainloop<float, AVEC, KFULL, true, true>(arow, avalid, lp.wpack, K, a.k_steps / 2, 0, nt0, n_tiles_total, stage, acc, 0, nblk);
    }
    if constexpr (GGCN_LAB_WIDE8_DENSE) {
    // every accumulator tile -> its two bf16 planes (B-operand fragments of the aggregation MFMAs), in place
    bf16x8 hf[4][RN][2][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j) split2(acc[i][j], hf[i][j]);

    const int c = lane & 31, h = lane >> 5;
    float vb[RN], vsg[RN], vga[RN], vgb[RN];
    bool col_ok[RN];
    {
        const float *dummy = a.X;
        const float *pb = bias ? bias : dummy, *psg = store_gate ? store_gate : dummy;
        const float *pga = pool_gate_a ? pool_gate_a : dummy, *pgb = pool_gate_b ? pool_gate_b : dummy;
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            const int gn = (nt0 + j) * NT + c;
            col_ok[j] = gn < F;
            const int gnc = col_ok[j] ? gn : 0;
            const int64_t at = (int64_t)g * F + gnc;
            vb[j] = bias ? pb[gnc] : 0.0f;
            vsg[j] = store_gate ? psg[at] : 1.0f;
            vga[j] = pool_gate_a ? pga[at] : 1.0f;
            vgb[j] = pool_gate_b ? pgb[at] : 1.0f;
        }
    }
    // this lane's adjacency rows of output block io (node 32 io + c), one block ahead of their use; words past the graph
    // and rows past T read as zeros
    auto load_masks = [&](int io, uint32_t (&m)[8]) {
        const int node = 32 * io + c;
        const bool ok = node < T;
#pragma unroll
        for (int ii = 0; ii < 8; ++ii) {
            const bool okw = ok && ii < W;
            const uint32_t v = a.rowmask[okw ? ((int64_t)g * T + node) * W + ii : 0];
            m[ii] = okw ? v : 0u;
        }
    };
    uint32_t mw[2][8];
    load_masks(4 * rg, mw[0]);
    float *stage_lds = reinterpret_cast<float *>(lds8 + kW8Ex + wave * kW8Stage);
    const int perm_base = 16 * h;
    const int lane_off = 4 * h * ldo + c;
    float vmax[RN], vmin[RN];
#pragma unroll
    for (int j = 0; j < RN; ++j) { vmax[j] = -INFINITY; vmin[j] = INFINITY; }

    auto tiles = [&](auto has_out) {
        constexpr bool vst = VST && decltype(has_out)::value;
        constexpr bool direct_store = !VST && decltype(has_out)::value;
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            __syncthreads();   // the fragments of column tile j - 1 (j = 0: the last stage's operand planes) have been read
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
                        *reinterpret_cast<bf16x8 *>(lds8 + (((wave * 4 + i) * 2 + p) * 2 + ks) * 1024 + lane * 16) = hf[i][j][p][ks];
            __syncthreads();
            const bool tile_ok = nt0 + j < n_tiles_total;   // wavefront-uniform: column tile past F
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = 4 * j + i, io = 4 * rg + i, node0 = 32 * io;
                // the next block's masks (wraps to this wavefront's first block for the second column tile)
                if (t + 1 < 4 * RN) load_masks(4 * rg + ((i + 1) & 3), mw[(t + 1) & 1]);
                if (node0 >= T || !tile_ok) continue;   // wavefront-uniform: a block of padding rows (no barrier below)
                const uint32_t (&m)[8] = mw[t & 1];
                int deg = 0;
#pragma unroll
                for (int ii = 0; ii < 8; ++ii) deg += __popc(m[ii]);
                const float inv = 1.0f / (float)(deg + 1);                  // gcn.py:35
                f32x16 y;
#pragma unroll
                for (int r = 0; r < 16; ++r) y[r] = 0.0f;
                // The four operand fragments of block ii + 1 are asked for before block ii's MFMAs (an empty block's are read in
                // vain): read just in front of their use, every pair of MFMAs waited out an LDS round trip.
                auto frag_src = [&](int ii) { return lds8 + ((((ii >> 2) * 4 + cg) * 4 + (ii & 3)) * 4) * 1024 + lane * 16; };
                bf16x8 fr[2][4];
#pragma unroll
                for (int q = 0; q < 4; ++q) fr[0][q] = *reinterpret_cast<const bf16x8 *>(frag_src(0) + q * 1024);
#pragma unroll
                for (int ii = 0; ii < 8; ++ii) {
                    if (32 * ii >= T) break;   // workgroup-uniform: source blocks of padding rows
                    if (ii + 1 < 8) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) fr[(ii + 1) & 1][q] = *reinterpret_cast<const bf16x8 *>(frag_src(ii + 1) + q * 1024);
                    }
                    // a block without an edge adds nothing (dependency arcs are mostly short: away from the diagonal most
                    // blocks of a parse are empty); wavefront-uniform
                    if (__builtin_amdgcn_ballot_w64(m[ii] != 0u) == 0) continue;
                    bf16x8 af[2];
                    if constexpr (((GGCN_LAB_OFF) & 32) != 0) {   // (timing build: no expansion)
                        union { bf16x8 v; uint32_t w[4]; } u;
                        u.w[0] = u.w[1] = u.w[2] = u.w[3] = m[ii] & 0x3F803F80u;
                        af[0] = af[1] = u.v;
                    } else
                    expand_mask(m[ii] >> (4 * h), af);   // once per block: both planes use it (small plane first)
#pragma unroll
                    for (int p = 1; p >= 0; --p)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks)
                            if constexpr (!((GGCN_LAB_OFF) & 64))
                                y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], fr[ii & 1][2 * p + ks], y, 0, 0, 0);   // gcn.py:41
                }
                float rinv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row0 = (r & 3) + 8 * (r >> 2);
                    rinv[r] = __int_as_float(__builtin_amdgcn_ds_bpermute(perm_base + 4 * row0, __float_as_int(inv)));
                }
                float *tile = decltype(has_out)::value ? out + ((int64_t)g * T + node0) * ldo + (nt0 + j) * NT : nullptr;
                auto finish = [&](auto whole_c) {   // whole: all 32 rows of the block are nodes (wavefront-uniform) -- no row test
                    constexpr bool WHOLE = decltype(whole_c)::value;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row0 = (r & 3) + 8 * (r >> 2);  // this lane's row is row0 + 4h
                        const float v = y[r] * rinv[r] + vb[j];   // gcn.py:41,43
                        if (vst) stage_lds[(row0 + 4 * h) * 32 + c] = v * vsg[j];
                        if (WHOLE || node0 + row0 + 4 * h < T) {
                            if (direct_store && col_ok[j]) tile[lane_off + row0 * ldo] = v * vsg[j];
                            vmax[j] = vmaxf_raw(vmax[j], v);
                            vmin[j] = vminf_raw(vmin[j], v);
                        }
                    }
                };
                if constexpr (((GGCN_LAB_OFF) & 16) != 0) {   // (timing build: nothing behind the neighbour sums)
                    asm volatile("" :: "v"(y[0]), "v"(y[5]), "v"(y[10]), "v"(y[15]), "v"(rinv[3]));
                    continue;
                }
                if (node0 + 32 <= T) finish(std::true_type{});
                else finish(std::false_type{});
                if (vst) {   // 32 rows x 128 B leave as 16 B per lane: 8 rows per instruction
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const int colq = (lane & 7) * 4;
                    const int gcol = (nt0 + j) * NT + colq;
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int row = 8 * it + (lane >> 3);
                        const float4 v4 = *reinterpret_cast<const float4 *>(&stage_lds[row * 32 + colq]);
                        if (node0 + row < T && gcol < F) store_out4(tile + row * ldo + colq, v4);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
    };
    if constexpr (!((GGCN_LAB_OFF) & 128)) {
        if (out) tiles(std::true_type{});
        else tiles(std::false_type{});
    } else {
        asm volatile("" :: "v"(hf[0][0][0][0]), "v"(hf[3][1][1][1]), "v"(hf[1][0][1][0]), "v"(hf[2][1][0][1]));
    }

    // pools of the graph: max over ALL its rows (bert_amir5.py:635-640) -- the two row groups meet in LDS
    if (pool_a || pool_b || lp.ov_partial) {
        float *pl = reinterpret_cast<float *>(lds8 + kW8Ex);   // [wavefront][max / min][column tile][32] over the store staging
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            const float mx = fmaxf(vmax[j], upper_half_to_lower(vmax[j]));
            const float mn = fminf(vmin[j], upper_half_to_lower(vmin[j]));
            if (h == 0) {
                pl[((wave * 2 + 0) * RN + j) * 32 + c] = mx;
                pl[((wave * 2 + 1) * RN + j) * 32 + c] = mn;
            }
        }
        __syncthreads();
        if (rg == 0) {
            float dot = 0.0f;
#pragma unroll
            for (int j = 0; j < RN; ++j) {
                if (nt0 + j >= n_tiles_total) break;
                const float mx = fmaxf(pl[((wave * 2 + 0) * RN + j) * 32 + c], pl[(((wave + 4) * 2 + 0) * RN + j) * 32 + c]);
                const float mn = fminf(pl[((wave * 2 + 1) * RN + j) * 32 + c], pl[(((wave + 4) * 2 + 1) * RN + j) * 32 + c]);
                if (h == 0 && col_ok[j]) {
                    const int gn = (nt0 + j) * NT + c;
                    const float pa = vga[j] * (vga[j] >= 0.0f ? mx : mn), pb = vgb[j] * (vgb[j] >= 0.0f ? mx : mn);
                    if (pool_a) pool_a[(int64_t)g * F + gn] = pa;
                    if (pool_b) pool_b[(int64_t)g * F + gn] = pb;
                    dot = fmaf(pa, pb, dot);
                }
            }
            if (lp.ov_partial && nt0 < n_tiles_total) {
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) dot += __shfl_xor(dot, d);
                if (lane == 0) lp.ov_partial[(int64_t)g * ((F + 63) / 64) + (nt0 >> 1)] = dot;
            }
        }
    }

    } else {
    // ---- neighbour sums over the EDGES, out of an fp32 tile in LDS (gcn.py:41) ----------------------------------------
    // Per 32-column tile j every wavefront writes its accumulators, as they are, into its column group's tile
    // [256 rows][32 columns] fp32 (4 x 32 KiB over the dead stage buffers).  Then 8 lanes x 16 B cover a row: a wavefront
    // sums 8 destination rows at once, each 8-lane group reading its row's edge list (up to kW8Cap source ids, made once
    // per workgroup from the row masks: ids and degrees in LDS) and then the source rows' 16-byte pieces, 8 in flight --
    // two dependent LDS round trips per 8 edges, ~5 reads and 20 adds per row of a parse, instead of 8 blocks x 4 MFMAs and
    // the expansion of 8 mask words into MFMA operands.  The sums are exact fp32; rows leave straight from registers as
    // 16-byte stores.  (A first form that walked the mask bits one LDS read at a time took 630 us where the MFMA form
    // took 470: every edge waited out its own round trip.)
    const int q8 = lane >> 3, cl = lane & 7;
    const int c = lane & 31, h = lane >> 5;
    char *tile_cg = lds8 + cg * (256 * 128);
    unsigned short *s_ids = reinterpret_cast<unsigned short *>(lds8 + kW8Ex + 4096);   // [256 rows][kW8Cap]
    int *s_deg = reinterpret_cast<int *>(lds8 + kW8Ex + 4096 + 256 * kW8Cap * 2);      // [256]
    float *s_inv = reinterpret_cast<float *>(lds8 + kW8Ex + 4096 + 256 * kW8Cap * 2 + 1024);   // [256] 1 / (deg + 1)
    const int zero_off = kW8Ex + 4096 + 256 * kW8Cap * 2 + 2048;                         // 128 B of zeros
    // 16-byte chunk `chunk` of row `row`: the 64-byte half is flipped on rows 2, 3 (mod 4), so that the four 8-lane groups a
    // ds_read_b128 serves together (two read chunks 0-3, two chunks 4-7 of their rows) collide on one row pair in four
    auto tile_off = [](int row, int chunk) { return row * 128 + ((chunk ^ (((row >> 1) & 1) << 2)) << 4); };
    // edge lists of the graph's rows, once per workgroup (thread t < 256: row t)
    if (tid < 256) {
        const int row = tid;
        uint32_t mwd[8];
#pragma unroll
        for (int wi = 0; wi < 8; ++wi) {
            const bool ok = row < T && wi < W && !((GGCN_LAB_OFF) & 32);   // (timing build: no masks, empty lists)
            const uint32_t v = a.rowmask[ok ? ((int64_t)g * T + row) * W + wi : 0];
            mwd[wi] = ok ? v : 0u;
        }
        int deg = 0, e = 0;
#pragma unroll
        for (int wi = 0; wi < 8; ++wi) {
            uint32_t w = mwd[wi];
            deg += __popc(w);
            while (w && e < kW8Cap) {   // stored: the source row's byte offset in a tile (chunk 0; a lane XORs its 16 cl in)
                s_ids[row * kW8Cap + e++] = (unsigned short)tile_off(32 * wi + __builtin_ctz(w), 0);
                w &= w - 1;
            }
        }
        // the rest of the list points at row 255: a padding row for T < 256, all zeros in every tile (its X row was staged as
        // zeros), so the sums need no test per slot; T = 256 has no such row and masks the slots instead
        for (; e < kW8Cap; ++e) s_ids[row * kW8Cap + e] = (unsigned short)tile_off(255, 0);
        s_deg[row] = deg;
        s_inv[row] = 1.0f / (float)(deg + 1);                               // gcn.py:35
        if (tid < 32) reinterpret_cast<float *>(lds8 + zero_off)[tid] = 0.0f;
    }
    float vmax[RN][4], vmin[RN][4];
#pragma unroll
    for (int j = 0; j < RN; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) { vmax[j][k] = -INFINITY; vmin[j][k] = INFINITY; }
    // bias and store gate of this lane's columns for both column tiles: asked for here, used behind two barriers
    const float *dummy = a.X;
    float b4[RN][4], sg4[RN][4];
    bool cok[RN][4];
#pragma unroll
    for (int j = 0; j < RN; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int col = (nt0 + j) * NT + 4 * cl + k;
            cok[j][k] = col < F;
            const int cc = cok[j][k] ? col : 0;
            b4[j][k] = bias ? (bias ? bias : dummy)[cc] : 0.0f;
            sg4[j][k] = store_gate ? (store_gate ? store_gate : dummy)[(int64_t)g * F + cc] : 1.0f;
        }
    const int tile_lane = cg * (256 * 128) + 16 * cl, zero_lane = zero_off + 16 * cl;
    const int n_steps = ((GGCN_LAB_OFF) & 16) ? 1 : 16;   // (timing build: one row step only)
    const bool full_slot = T == 256;
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        __syncthreads();   // the tile of column tile j - 1 (j = 0: the last stage's operand planes) has been read
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 128 * rg + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                *reinterpret_cast<float *>(tile_cg + tile_off(row, c >> 2) + (c & 3) * 4) = acc[i][j][r];
            }
        __syncthreads();   // (also: the edge lists are complete)
        if (nt0 + j >= n_tiles_total) continue;   // wavefront-uniform: column tile past F (the barriers above are met)
        const int col0 = (nt0 + j) * NT + 4 * cl;   // this lane's four columns
        // degree, reciprocal and the first 8 list entries of a row step are read one step ahead: a step is then ONE
        // dependent LDS round trip (the source rows) instead of two
        int deg_n = s_deg[128 * rg + q8];
        float inv_n = s_inv[128 * rg + q8];
        uint4 idq_n = *reinterpret_cast<const uint4 *>(s_ids + (128 * rg + q8) * kW8Cap);
        for (int it = 0; it < n_steps; ++it) {
            if (128 * rg + 8 * it >= T) break;   // wavefront-uniform: only padding rows from here on
            const int row = 128 * rg + 8 * it + q8;
            const int deg = deg_n;
            const float inv = inv_n;
            const uint4 idq0 = idq_n;
            if (it + 1 < 16) {   // (rows 248..255 of the second row group exist in LDS: the lists cover all 256 slots)
                deg_n = s_deg[row + 8];
                inv_n = s_inv[row + 8];
                idq_n = *reinterpret_cast<const uint4 *>(s_ids + (row + 8) * kW8Cap);
            }
            float s4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            auto pass = [&](const uint4 &idq, int e0) {
                const uint32_t idw[4] = {idq.x, idq.y, idq.z, idq.w};
                float4 v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int base = (int)((idw[e >> 1] >> (16 * (e & 1))) & 0xFFFFu);
                    int off = tile_lane ^ base;   // (base has no bits below 64; tile_lane = cg base + 16 cl)
                    if (full_slot) off = e0 + e < deg ? off : zero_lane;   // workgroup-uniform: T = 256
                    v[e] = *reinterpret_cast<const float4 *>(lds8 + off);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) { s4[0] += v[e].x; s4[1] += v[e].y; s4[2] += v[e].z; s4[3] += v[e].w; }
            };
            if (deg <= kW8Cap) {
                pass(idq0, 0);   // (a row without a neighbour adds eight zeros)
                if (deg > 8) pass(*reinterpret_cast<const uint4 *>(s_ids + row * kW8Cap + 8), 8);   // (divergent per 8-lane group)
            } else {   // more neighbours than a list holds: walk the mask words themselves (rare, slow, same sums in another order)
                for (int wi = 0; wi < W; ++wi) {
                    uint32_t w = a.rowmask[((int64_t)g * T + row) * W + wi];
                    while (w) {
                        const int src = 32 * wi + __builtin_ctz(w);
                        w &= w - 1;
                        const float4 v = *reinterpret_cast<const float4 *>(tile_cg + tile_off(src, cl));
                        s4[0] += v.x; s4[1] += v.y; s4[2] += v.z; s4[3] += v.w;
                    }
                }
            }
            if (row < T) {
                float o4[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float v = s4[k] * inv + b4[j][k];                 // gcn.py:41,43
                    o4[k] = v * sg4[j][k];
                    vmax[j][k] = vmaxf_raw(vmax[j][k], v);
                    vmin[j][k] = vminf_raw(vmin[j][k], v);
                }
                if (out && !((GGCN_LAB_OFF) & 8)) {   // (timing build: no stores)
                    float *dst = out + ((int64_t)g * T + row) * ldo + col0;
                    if constexpr (VST) {
                        if (cok[j][0]) store_out4(dst, make_float4(o4[0], o4[1], o4[2], o4[3]));
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (cok[j][k]) dst[k] = o4[k];
                    }
                }
            }
        }
    }
    // pools of the graph: max over ALL its rows (bert_amir5.py:635-640): across the 8 row classes of the wavefront (lanes 8
    // apart), then the two row groups meet in LDS
    if ((pool_a || pool_b || lp.ov_partial) && !((GGCN_LAB_OFF) & 64)) {   // (timing build: no pools)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int d = 8; d <= 32; d <<= 1) {
                    vmax[j][k] = vmaxf_raw(vmax[j][k], __shfl_xor(vmax[j][k], d));
                    vmin[j][k] = vminf_raw(vmin[j][k], __shfl_xor(vmin[j][k], d));
                }
        float *pl = reinterpret_cast<float *>(lds8 + kW8Ex);   // [wavefront][max / min][column tile][32]
        __syncthreads();
        if (q8 == 0) {
#pragma unroll
            for (int j = 0; j < RN; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    pl[((wave * 2 + 0) * RN + j) * 32 + 4 * cl + k] = vmax[j][k];
                    pl[((wave * 2 + 1) * RN + j) * 32 + 4 * cl + k] = vmin[j][k];
                }
        }
        __syncthreads();
        if (rg == 0) {
            float dot = 0.0f;
            if (lane < 32) {
#pragma unroll
                for (int j = 0; j < RN; ++j) {
                    const int gn = (nt0 + j) * NT + lane;
                    if (nt0 + j < n_tiles_total && gn < F) {
                        const float mx = fmaxf(pl[((wave * 2 + 0) * RN + j) * 32 + lane], pl[(((wave + 4) * 2 + 0) * RN + j) * 32 + lane]);
                        const float mn = fminf(pl[((wave * 2 + 1) * RN + j) * 32 + lane], pl[(((wave + 4) * 2 + 1) * RN + j) * 32 + lane]);
                        const float ga = pool_gate_a ? pool_gate_a[(int64_t)g * F + gn] : 1.0f;
                        const float gb = pool_gate_b ? pool_gate_b[(int64_t)g * F + gn] : 1.0f;
                        const float pa = ga * (ga >= 0.0f ? mx : mn), pb = gb * (gb >= 0.0f ? mx : mn);
                        if (pool_a) pool_a[(int64_t)g * F + gn] = pa;
                        if (pool_b) pool_b[(int64_t)g * F + gn] = pb;
                        dot = fmaf(pa, pb, dot);
                    }
                }
            }
            if (lp.ov_partial && nt0 < n_tiles_total) {   // fixed butterfly order; lanes 32-63 hold 0
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) dot += __shfl_xor(dot, d);
                if (lane == 0) lp.ov_partial[(int64_t)g * ((F + 63) / 64) + (nt0 >> 1)] = dot;
            }
        }
    }
    }
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
        return fail(GGCN_EUNSUPPORTED, "%s: precision %d (use bf16x3, f16mx8 or f16mx6)", who, precision);
    if (!a.X) return fail(GGCN_EINVAL, "%s: null input pointer", who);
    if (a.T > 32 ? !a.rowmask : !a.graph_ops)
        return fail(GGCN_EINVAL, "%s: graphs of %d nodes need %s", who, a.T,
                    a.T > 32 ? "the row masks" : "the per-graph operand blocks of ggcn_graph_operands");
    if (a.T <= 32 && !aligned16(a.graph_ops)) return fail(GGCN_EINVAL, "%s: graph_ops must be 16-byte aligned", who);
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
            vst = vst && (a.F % 4 == 0) && (lp.ldo % 4 == 0) && aligned16(lp.out);
        }
    }
    vst = vst && any_out;
    const bool avec = (a.K % 4 == 0) && (a.ldx % 4 == 0) && aligned16(a.X);
    const bool kfull = (a.K % BK == 0);
    a.k_steps = round_up(a.K, BK) / KSTEP;
    a.n_wg = (a.F + BN - 1) / BN;
    if (a.drop.thr != 0 && (a.T > 32 || a.n_parts != 1 || precision == GGCN_PREC_F16MX6))
        return fail(GGCN_EUNSUPPORTED, "%s: gate dropout is built into the one-launch layer of graphs of <= 32 nodes (bf16x3 / f16mx8)", who);
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
#define GGCN_LAUNCHW(SC, AV, KF, VS, SBV) \
    hipLaunchKernelGGL((layer_fused_wide_kernel<SC, AV, KF, VS, SBV>), dim3((unsigned)gridw), dim3(kThreads), 0, st, a)
#define GGCN_PICKW(SC, SBV)                                              \
    do {                                                                 \
        if (fast && vst) GGCN_LAUNCHW(SC, true, true, true, SBV);        \
        else if (fast) GGCN_LAUNCHW(SC, true, true, false, SBV);         \
        else GGCN_LAUNCHW(SC, false, false, false, SBV);                 \
    } while (0)
        if (sb == 8 && !GGCN_LAB_WIDE_SB8) {   // 129..256 nodes: eight wavefronts per graph (layer_fused_wide8_kernel)
#define GGCN_LAUNCH8(SC, AV, KF, VS)                                                                                      \
    do {                                                                                                                  \
        auto kern = layer_fused_wide8_kernel<SC, AV, KF, VS>;                                                             \
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,         \
                                kW8Lds) != hipSuccess)                                                                    \
            return fail(GGCN_ELAUNCH, "%s: cannot reserve %d bytes of LDS", who, kW8Lds);                                 \
        hipLaunchKernelGGL(kern, dim3((unsigned)gridw), dim3(kW8Threads), kW8Lds, st, a);                                 \
    } while (0)
#define GGCN_PICK8(SC)                                      \
    do {                                                    \
        if (fast && vst) GGCN_LAUNCH8(SC, true, true, true);        \
        else if (fast) GGCN_LAUNCH8(SC, true, true, false);         \
        else GGCN_LAUNCH8(SC, false, false, false);                 \
    } while (0)
            if (precision == GGCN_PREC_F16MX8) GGCN_PICK8(1);
            else GGCN_PICK8(0);
#undef GGCN_PICK8
#undef GGCN_LAUNCH8
            return check_launch(who);
        }
        if (precision == GGCN_PREC_F16MX8) { if (sb == 2) GGCN_PICKW(1, 2); else if (sb == 4) GGCN_PICKW(1, 4); else GGCN_PICKW(1, 8); }
        else { if (sb == 2) GGCN_PICKW(0, 2); else if (sb == 4) GGCN_PICKW(0, 4); else GGCN_PICKW(0, 8); }
#undef GGCN_PICKW
#undef GGCN_LAUNCHW
        return check_launch(who);
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
#define GGCN_LAUNCH6(FT, VS) hipLaunchKernelGGL((layer_fused6_kernel<FT, VS>), dim3((unsigned)grid), dim3(kThreads), 0, st, a)
        if (fullt && vst) GGCN_LAUNCH6(true, true);
        else if (fullt) GGCN_LAUNCH6(true, false);
        else if (vst) GGCN_LAUNCH6(false, true);
        else GGCN_LAUNCH6(false, false);
#undef GGCN_LAUNCH6
        return check_launch(who);
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
    if (precision == GGCN_PREC_F16MX8) GGCN_PICK(1);
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

int layer_fused(const float *X, int64_t ldx, const void *wpack, const uint32_t *rowmask, const void *graph_ops,
                const float *bias, int B, int T, int K, int F, const float *store_gate, const float *pool_gate_a,
                const float *pool_gate_b, float *out, int64_t ldo, float *pool_a, float *pool_b,
                float *overlap_partial, const float *overlap_in, float *overlap_out, int precision, hipStream_t st,
                const DropSpec *drop)
{
    if ((overlap_in == nullptr) != (overlap_out == nullptr))
        return fail(GGCN_EINVAL, "ggcn_layer_fused: overlap_in and overlap_out go together");
    if (out && ldo > (int64_t)INT32_MAX) return fail(GGCN_EUNSUPPORTED, "ggcn_layer_fused: ldo too large");
    FusedArgs a = {};
    a.X = X; a.ldx = ldx; a.rowmask = rowmask; a.graph_ops = static_cast<const char *>(graph_ops);
    a.ov_in = overlap_in; a.ov_out = overlap_out;
    a.B = B; a.T = T; a.K = K; a.F = F; a.n_parts = 1;
    if (drop) a.drop = *drop;
    a.part[0] = LayerPart{static_cast<const char *>(wpack), bias, nullptr, store_gate, pool_gate_a, pool_gate_b,
                          out, pool_a, pool_b, overlap_partial, (int)ldo};
    return launch_fused("ggcn_layer_fused", a, precision, st);
}

int block_fused(const float *X, int64_t ldx, const void *wpack1, const void *wpack12, const void *graph_ops,
                const float *bias1, const float *bias_mid, const float *bias2, int B, int T, int K, int F,
                const float *gate1, const float *gate2, float *gcn1, int64_t ld1, float *x_out, int64_t ld2,
                float *x1, float *y1, float *pool_out, float *overlap_partial, int precision, hipStream_t st)
{
    if (!gate1 || !gate2) return fail(GGCN_EINVAL, "ggcn_block_fused: gate1 and gate2 are required");
    if (!x1 || !y1) return fail(GGCN_EINVAL, "ggcn_block_fused: x1 and y1 are required");
    if (!x_out && !pool_out) return fail(GGCN_EINVAL, "ggcn_block_fused: neither x nor its pool requested");
    if ((gcn1 && ld1 > (int64_t)INT32_MAX) || (x_out && ld2 > (int64_t)INT32_MAX))
        return fail(GGCN_EUNSUPPORTED, "ggcn_block_fused: leading dimension too large");
    FusedArgs a = {};
    a.X = X; a.ldx = ldx; a.graph_ops = static_cast<const char *>(graph_ops);
    a.B = B; a.T = T; a.K = K; a.F = F; a.n_parts = 2;
    // bert_amir5.py:626-636: gcn1 (ungated; optional here), x1 = max_t gcn1*gate1, y1 = max_t gcn1*gate2
    a.part[0] = LayerPart{static_cast<const char *>(wpack1), bias1, nullptr, nullptr, gate1, gate2,
                          gcn1, x1, y1, overlap_partial, (int)ld1};
    // bert_amir5.py:639-640: x = gate2 * gc2(gcn1), out = max_t x.  A NULL mid bias (gc1 without bias) still
    // needs the second aggregation: a vector of zeros cannot be conjured here, so the caller passes one.
    if (!bias_mid) return fail(GGCN_EINVAL, "ggcn_block_fused: bias_mid (W2^T.b1, zeros when gc1 has no bias) is required");
    a.part[1] = LayerPart{static_cast<const char *>(wpack12), bias2, bias_mid, gate2, gate2, nullptr,
                          x_out, pool_out, nullptr, nullptr, (int)ld2};
    return launch_fused("ggcn_block_fused", a, precision, st);
}

int overlap_reduce(const float *partials, int B, int F, float *xy, hipStream_t st)
{
    if (!partials || !xy) return fail(GGCN_EINVAL, "ggcn_overlap_reduce: null pointer");
    if (B <= 0 || F <= 0) return fail(GGCN_EINVAL, "ggcn_overlap_reduce: B=%d F=%d must be positive", B, F);
    hipLaunchKernelGGL(overlap_reduce_kernel, dim3(1), dim3(kRedThreads), 0, st, partials, B * ((F + 63) / 64), B, xy);
    return check_launch("ggcn_overlap_reduce");
}

// this translation unit's copy of the sticky f16mx8 range flag (f16mx8_core.h): OR it into *dst (device memory), clear on request
__global__ void range_flag_fused_kernel(unsigned int *dst, int clear)
{
    const unsigned int v = mx8::g_range_flag;
    if (v) atomicOr(dst, v);
    if (clear) mx8::g_range_flag = 0u;
}
int range_flag_fused(unsigned int *dst, int clear, hipStream_t st)
{
    hipLaunchKernelGGL(range_flag_fused_kernel, dim3(1), dim3(1), 0, st, dst, clear);
    return check_launch("ggcn_range_flag");
}

}  // namespace ggcn
