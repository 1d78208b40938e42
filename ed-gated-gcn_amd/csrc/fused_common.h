// Shared by the one-launch layer kernels (fused_layer.hip: graphs of <= 32 nodes and the two-layer block; fused_wide.hip:
// 33..128 nodes; fused_wide8.hip: 129..256 nodes; fused6.hip: the optional f16mx6 form): argument blocks, the split of an
// accumulator tile into MFMA operand planes, the operands staged in LDS for the 32-node epilogue, and that epilogue.
#pragma once
#include "f16mx8_core.h"
#include "dropout_hash.h"
#include "lab_hooks.h"

namespace ggcn {

// what one group of column tiles ("part") computes: part 0 = the layer itself (or layer 1 of the block),
// part 1 = layer 2 of the block through W12
struct LayerPart {
    const char *wpack;          // ggcn_weight_pack image of this part's [K, F] matrix
    const float *bias;          // added after the (last) normalised aggregation, or NULL
    const float *mid;           // NULL: one aggregation.  Else: y = D.A.(D.A.h + mid) + bias
    const float *pre;           // NULL, or [F] added to `hidden` BEFORE the aggregation (graphs of 33..256 nodes): y = D.A.(h + 1.pre^T) + bias
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
    const char *graph_ops2;     // the block's W12 tiles: ggcn_graph_operands2 blocks ((D.A)^2 in the launch's plane type)
    const float *ov_in;         // partials an EARLIER launch wrote: block 0 reduces them to *ov_out first
    float *ov_out;
    int B, T, K, F;
    int g_tiles, n_wg, n_parts, k_steps;
    LayerPart part[2];
    DropSpec drop;              // training-mode keep masks of the gates (thr = 0: none); one part only
    unsigned long long *stamps; // diagnostics only (ggcn_debug_block_fused_stamped): [workgroup][2] = d(s_memtime), d(s_memrealtime) around the main loop
};


// per-file launchers behind launch_fused (fused_layer.hip); shapes and flags are validated there
int launch_fused_wide(const char *who, const FusedArgs &a, int precision, int sb, bool fast, bool vst, int64_t grid, hipStream_t st);
int launch_fused_wide8(const char *who, const FusedArgs &a, int precision, bool fast, bool vst, int64_t grid, hipStream_t st);
int launch_fused6(const char *who, const FusedArgs &a, bool fullt, bool vst, int64_t grid, hipStream_t st);   // GGCN_WITH_F16MX6 builds

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

// the same with the non-temporal hint always: the eight-wavefront kernel (129..256-node graphs) re-reads a graph's X rows
// from L2 once per column tile, and output lines allocated in L2 evict them (512 graphs: 129 nodes 284 -> 264 us, 160: 304 -> 284,
// 192: 342 -> 331, 231: 412 -> 406; the 32-node and 128-row kernels do not move and keep plain stores)
__device__ __forceinline__ void store_out4_stream(float *p, const float4 &v)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store(f4{v.x, v.y, v.z, v.w}, reinterpret_cast<f4 *>(p));
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
// ggcn_graph_operands2 (the block's second layer, MID): M2 = (D.A)^2 * 2^10 as A-operand fragments in the plane type
//   [0, 1024) hi part, k-step 0;  [1024, 2048) hi, k-step 1;  [2048, 3072) lo, k-step 0;  [3072, 4096) lo, k-step 1
//   (lane l at 16 l; element j <-> node 16s + 8(j>>2) + 4h + (j&3) as above)
//   [4096, 4224) rowsum(D.A) = deg / (deg + 1) in accumulator order: float [h][16]
constexpr int kOps2Bytes = GGCN_GRAPH_OPS2_BYTES;
static_assert(kOps2Bytes == 4096 + 128, "layout above");
constexpr float kM2Scale = 1024.0f, kM2InvScale = 1.0f / 1024.0f;


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
//   [kEpiBias ]  2 x [256 columns] floats: bias, mid bias (0 where NULL); [kMidMax] 4 floats: max |mid bias| per wavefront
template <int BASE>
struct EpiLds {   // byte offsets of the staged operands; the store staging of the epilogue is always [0, 32 KiB)
    static constexpr int kOps = BASE;
    static constexpr int kGate = kOps + 4 * kOps2Bytes;   // (the W12 tiles of the block stage the larger (D.A)^2 blocks)
    static constexpr int kBias = kGate + 3 * 4 * BN * 4;
    static constexpr int kMidMax = kBias + 2 * BN * 4;   // 4 floats: max |mid bias| per wavefront (the hidden-value bound of the range flag)
    static constexpr int kEnd = kMidMax + 16;
};
constexpr int kEpiLdsBytes = EpiLds<0>::kEnd;   // 16896 + 12288 + 2048 + 16 = 31248
static_assert(WM == 1, "one wavefront row: the workgroup's 4 graphs are every wavefront's 4 graphs");

template <int BASE>
__device__ __forceinline__ void stage_epilogue_operands(const FusedArgs &a, const LayerPart &lp, int g0, int n_wgi, char *lds, int tid)
{
    constexpr int kEpiOps = EpiLds<BASE>::kOps, kEpiGate = EpiLds<BASE>::kGate, kEpiBias = EpiLds<BASE>::kBias;
    const int B = a.B, F = a.F;
    // operand blocks: 544 pieces of 16 B (a layer, layer 1 of the block) or 1056 (the block's W12 tiles: (D.A)^2)
    constexpr int kPieceIts = (4 * kOps2Bytes / 16 + kThreads - 1) / kThreads;   // 5
    uint4 piece[kPieceIts];
    const bool second = lp.mid != nullptr;   // workgroup-uniform
    const int blk_bytes = second ? kOps2Bytes : kOpsBytes;
    const char *ops_src = second ? a.graph_ops2 : a.graph_ops;
    const int n_pieces = 4 * blk_bytes / 16;
#pragma unroll
    for (int it = 0; it < kPieceIts; ++it) {
        const int idx = tid + it * kThreads;
        const int idc = idx < n_pieces ? idx : 0;
        const int gi = (idc * 16) / blk_bytes;
        // a graph past the batch reads graph g0's bytes instead (never used)
        const int64_t off = (int64_t)g0 * blk_bytes + (g0 + gi < B ? idc * 16 : idc * 16 - gi * blk_bytes);
        piece[it] = *reinterpret_cast<const uint4 *>(ops_src + off);
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
    for (int it = 0; it < kPieceIts; ++it) {
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
    float mm = lp.mid && cok ? fabsf(vmidb) : 0.0f;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mm = fmaxf(mm, __shfl_xor(mm, d));
    if ((tid & 63) == 0) reinterpret_cast<float *>(lds + EpiLds<BASE>::kMidMax)[tid >> 6] = mm;
    // visible to every wavefront after the main loop's first barrier
}

// The sticky range flag of the fp16-plane epilogue (f16mx8_core.h): |hidden[r,f]| <= max_k |x[r,k]| * sum_k |w[k,f]| is all the
// epilogue rounds to fp16 (the block's W12 tiles included: one application of (D.A)^2 to X.W12; max |mid| stays in the
// bound as slack); every element of the tile is split by exactly one lane, so the lane that owns a row's largest element
// speaks for the row.  pack_bytes: size of the image without its trailer.
template <int BASE>
__device__ __forceinline__ void fused_range_verdict(float amax, const char *wpack, int64_t pack_bytes, const char *lds, bool window)
{
    const float bw = *reinterpret_cast<const float *>(wpack + pack_bytes);
    const float4 mm = *reinterpret_cast<const float4 *>(lds + EpiLds<BASE>::kMidMax);
    mx8::range_verdict(amax, bw, fmaxf(fmaxf(mm.x, mm.y), fmaxf(mm.z, mm.w)), window);
}

// The epilogue's operands (EpiLds<BASE>: the 4 graphs' operand blocks, 3 x 4 gate rows, bias and mid rows) brought to LDS by LDS-DMA:
// no registers, no wait -- stage_epilogue_operands (fused_common.h) loads them into registers and stores them, which costs a
// workgroup alone on its CU 2.6-4.6 us in front of its main loop (tools/block8_timing.py stamps).  The pieces land under the main
// loop's first stages (its vmcnt waits and barriers cover them long before the epilogue).  Whole tiles only (4 real graphs, 256
// real columns, 16-byte aligned gate rows): the caller falls back otherwise.  tid: 0..255 inside the group.
template <int BASE>
__device__ __forceinline__ void stage_epilogue_operands_dma(const FusedArgs &a, const LayerPart &lp, int g0, int n_wgi, char *lds, int tid)
{
    constexpr int kEpiOps = EpiLds<BASE>::kOps, kEpiGate = EpiLds<BASE>::kGate, kEpiBias = EpiLds<BASE>::kBias;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int F = a.F;
    const bool second = lp.mid != nullptr;
    const int total = 4 * (second ? kOps2Bytes : kOpsBytes);
    const char *src = (second ? a.graph_ops2 : a.graph_ops) + (int64_t)g0 * (second ? kOps2Bytes : kOpsBytes);
    for (int piece = wave; piece * 1024 < total; piece += 4)
        if (piece * 1024 + lane * 16 < total)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + piece * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void *)(lds + kEpiOps + piece * 1024), 16, 0, 0);
    const float *gp[3] = {lp.store_gate, lp.pool_gate_a, lp.pool_gate_b};
    float *gl = reinterpret_cast<float *>(lds + kEpiGate);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (gp[k]) {   // (workgroup-uniform) wavefront w brings graph w's row of this gate: 1 KiB = 256 columns
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gp[k] + (int64_t)(g0 + wave) * F + n_wgi * BN + lane * 4),
                                             (__attribute__((address_space(3))) void *)(lds + kEpiGate + ((k * 4 + wave) * BN) * 4), 16, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) gl[(k * 4 + i) * BN + tid] = 1.0f;
        }
    }
    float *bl = reinterpret_cast<float *>(lds + kEpiBias);
    if (lp.bias) {
        if (wave == 0) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(lp.bias + n_wgi * BN + lane * 4),
                                                        (__attribute__((address_space(3))) void *)(lds + kEpiBias), 16, 0, 0);
    } else bl[tid] = 0.0f;
    if (lp.mid) {
        if (wave == 1) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(lp.mid + n_wgi * BN + lane * 4),
                                                        (__attribute__((address_space(3))) void *)(lds + kEpiBias + BN * 4), 16, 0, 0);
    } else bl[BN + tid] = 0.0f;
}
// the range verdict with max |mid| taken from the staged row by every wavefront itself (no cross-wavefront hand-over)
template <int BASE>
__device__ __forceinline__ void dma_range_verdict(float amax, const char *wpack, int64_t pack_bytes, const char *lds, int lane)
{
    const float4 m4 = *reinterpret_cast<const float4 *>(lds + EpiLds<BASE>::kBias + BN * 4 + lane * 16);
    float mm = fmaxf(fmaxf(fabsf(m4.x), fabsf(m4.y)), fmaxf(fabsf(m4.z), fabsf(m4.w)));
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) mm = fmaxf(mm, __shfl_xor(mm, d));
    const float bw = *reinterpret_cast<const float *>(wpack + pack_bytes);
    mx8::range_verdict(amax, bw, mm, true);
}

// plain v_max / v_min (fmaxf first quiets a possible signalling NaN of an operand the compiler cannot prove canonical -- after a
// lane exchange, say: one extra instruction per operand)
__device__ __forceinline__ float vmax_raw(float x, float y) { float d; asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y)); return d; }
__device__ __forceinline__ float vmin_raw(float x, float y) { float d; asm("v_min_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y)); return d; }

// The two lane halves hold different rows of the same column.  max over both halves AND min over both halves, both
// delivered to lanes 0-31, in 2 swaps + 2 VALU and no copies: v_permlane32_swap exchanges the upper half of its first
// operand with the lower half of its second, so swap(vmax, vmin) leaves {vmax.lo | vmin.lo}, {vmax.hi | vmin.hi}: their
// max is the full maximum in lanes 0-31, their min the full minimum in lanes 32-63; a second swap brings that one down.
// (The earlier form copied each value, swapped it with itself and canonicalised both sides: ~14 instructions.)
__device__ __forceinline__ void meet_halves(float &vmax, float &vmin)
{
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(vmax), __float_as_uint(vmin), false, false);
    const float a = __uint_as_float(r[0]), b = __uint_as_float(r[1]);
    const float m = vmax_raw(a, b), n = vmin_raw(a, b);
    const auto q = __builtin_amdgcn_permlane32_swap(__float_as_uint(n), __float_as_uint(a), false, false);
    vmax = m;                            // lanes 0-31
    vmin = __uint_as_float(q[1]);        // lanes 0-31: n of lanes 32-63
}

// sum over the 64 lanes, delivered to lane 0, without an LDS round trip per step (a ds_bpermute butterfly is six dependent
// LDS operations, each waited for): four DPP adds leave every 16-lane row's sum in all of its lanes, lane 0 adds the other
// three rows' from scalar registers.  Fixed order: deterministic.
__device__ __forceinline__ float wave_sum_to_lane0(float v)
{
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __uint_as_float(__builtin_amdgcn_update_dpp(0, __float_as_uint(v), 0x140, 0xF, 0xF, true));   // row_mirror
    const float r1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 16));
    const float r2 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 32));
    const float r3 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), 48));
    return (v + r1) + (r2 + r3);
}

// a global array behind a buffer resource: stores take a 32-bit lane offset and a scalar offset instead of a 64-bit address
// per lane (v_ashrrev + v_lshl_add_u64 per store, a sixth of the epilogue's VALU instructions)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t out_rsrc(void *p)
{
    return __builtin_amdgcn_make_buffer_rsrc(p, 0, 0x7fffffff, 0x00020000);
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
        // the graph's operands out of LDS.  A layer / layer 1 of the block: the adjacency as 0xFFFF elements (one AND with the
        // plane type's 1.0 makes the exact A operand) + 1 / (rowsum + 1).  MID (layer 2 of the block through W12 = W1.W2):
        // M2 = (D.A)^2 * 2^10 as hi and lo fragments + rowsum(D.A) -- D.A.(D.A.H + 1.mid^T) + 1.b^T = M2.H + rowsum(D.A).mid^T
        // + 1.b^T: ONE split of H and 6 MFMAs (M2hi.Hhi, M2hi.Hlo, M2lo.Hhi; the lo.lo term is 2^-22 of the result) where the
        // two applications took two splits, an intermediate normalise pass and 8 MFMAs.
        constexpr int kBlk = MID ? kOps2Bytes : kOpsBytes;
        constexpr int kRowOff = MID ? 4096 : 2048;
        frag afv[2], afl[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const uint4 raw = *reinterpret_cast<const uint4 *>(ops_lds + i * kBlk + s * 1024 + lane * 16);
            union { frag v; uint32_t w[4]; } u;
            if constexpr (MID) {
                u.w[0] = raw.x; u.w[1] = raw.y; u.w[2] = raw.z; u.w[3] = raw.w;
                afv[s] = u.v;
                const uint4 rl = *reinterpret_cast<const uint4 *>(ops_lds + i * kBlk + 2048 + s * 1024 + lane * 16);
                u.w[0] = rl.x; u.w[1] = rl.y; u.w[2] = rl.z; u.w[3] = rl.w;
                afl[s] = u.v;
            } else {
                u.w[0] = raw.x & P::kOne; u.w[1] = raw.y & P::kOne;
                u.w[2] = raw.z & P::kOne; u.w[3] = raw.w & P::kOne;
                afv[s] = u.v;
            }
        }
        float4 rv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) rv[q] = *reinterpret_cast<const float4 *>(ops_lds + i * kBlk + kRowOff + h * 64 + q * 16);
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

        // gcn.py:41: agg = ADJ_g . hidden_g (MID: M2_g . hidden_g), small terms first; the column tiles' chains are
        // issued one behind the other, so that a tile's split and element-wise work sit under the other's MFMAs
        f32x16 y[RN];
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            frag hf[2][2];
            P::split(acc[i][j], hf);
#pragma unroll
            for (int r = 0; r < 16; ++r) y[j][r] = 0.0f;
            if constexpr (MID) {
#pragma unroll
                for (int s = 0; s < 2; ++s) y[j] = P::mma(afl[s], hf[0][s], y[j]);
#pragma unroll
                for (int s = 0; s < 2; ++s) y[j] = P::mma(afv[s], hf[1][s], y[j]);
#pragma unroll
                for (int s = 0; s < 2; ++s) y[j] = P::mma(afv[s], hf[0][s], y[j]);
            } else {
#pragma unroll
                for (int p = 1; p >= 0; --p)
#pragma unroll
                    for (int s = 0; s < 2; ++s) { if constexpr (!((GGCN_LAB_EPI) & 2)) y[j] = P::mma(afv[s], hf[p][s], y[j]); else y[j][s] += (float)hf[p][s][0]; }
            }
        }
        // per row: the factor of the aggregate and what is added to it.  A layer: 1 / (rowsum + 1) and the bias (gcn.py:41,43);
        // MID: 2^-10 (M2's scale) and rowsum(D.A) * mid + bias
        float rinv[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            rinv[4 * q] = rv[q].x; rinv[4 * q + 1] = rv[q].y; rinv[4 * q + 2] = rv[q].z; rinv[4 * q + 3] = rv[q].w;
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
                const float v = MID ? fmaf(y[j][r], kM2InvScale, fmaf(rinv[r], vmid[j], bj)) : y[j][r] * rinv[r] + bj;   // gcn.py:41,43
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
                meet_halves(vmax, vmin);
                const float ga = vga[j], gb = vgb[j];
                pa = ga * (ga >= 0.0f ? vmax : vmin);
                pb = gb * (gb >= 0.0f ? vmax : vmin);
            }
            if (h == 0 && col_ok[j]) {
                // pooled rows through buffer resources: lane offset = column, scalar offset = graph row (the graph's F floats
                // from the array's base: rebuilt per graph on the scalar unit, so B * F may exceed 32 bits)
                if (pool_a) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(pa), out_rsrc(pool_a + (int64_t)g * F), 4 * gn, 0, 0);
                if (pool_b) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(pb), out_rsrc(pool_b + (int64_t)g * F), 4 * gn, 0, 0);
                dot = fmaf(pa, pb, dot);
            }
        }
        if (ov_partial && nt0 < n_tiles_total) {  // fixed order; lanes with h = 1 hold 0
            const float tot = wave_sum_to_lane0(dot);
            if (lane == 0) ov_partial[(int64_t)g * ((F + 63) / 64) + (nt0 >> 1)] = tot;
        }
        if constexpr (vst) {
            // rows of 64 columns (both column tiles of this wavefront) leave as 16 B per lane: one
            // instruction stores 4 rows x 256 contiguous bytes instead of 2 rows x 128 B
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            const int colq = (lane & 15) * 4;
            const int gcol = nt0 * NT + colq;
            // the graph's rows behind a buffer resource (base rebuilt per graph on the scalar unit): lane offset = this
            // lane's row of the first group + its columns, scalar offset = 4 rows per step -- no 64-bit lane arithmetic
            const __amdgpu_buffer_rsrc_t orsrc = out_rsrc(out + ((int64_t)g * T) * ldo);
            const int voff = ((lane >> 4) * ldo + gcol) * 4;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = 4 * it + (lane >> 4);
                const float4 v4 = *reinterpret_cast<const float4 *>(&stage_lds[row * 64 + (colq ^ (32 * ((row >> 2) & 1)))]);
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                if ((FULLT || row < T) && gcol < F && !((GGCN_LAB_EPI) & 1))   // (GGCN_LAB_EPI 1: timing build without the stores)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v4), orsrc, voff, 4 * it * ldo * 4, (GGCN_LAB_NT_STORE) ? 2 : 0);
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (i == 0) GGCN_TRACE(7);
    }
}


}  // namespace
}  // namespace ggcn
