// Gated CSR aggregation -- the hand-written SpMM of the hot path.
//
// Replaces, in ONE pass over the linear's output Hd [N,F]:
//   models/gcn.py:35        denom = sum(adj, dim=2) + 1
//   models/gcn.py:41        matmul(adj, hidden) / denom       (87 % zeros when dense)
//   models/gcn.py:43        + bias
//   models/bert_amir5.py:627,631,639   * gate  (a [B,F] gate broadcast over the T tokens;
//                                      the reference materialises it as [B,T,F], :621-622)
//   models/bert_amir5.py:635,636,640   max over the T tokens of a graph
//
// Mapping (MI355X, wave64):
//   * workgroup = 4 wavefronts = (graph b, 256-column slab); wavefront w owns the
//     destination rows b*T + w, w+4, ...  -> one wavefront per destination node;
//   * a lane owns 4 consecutive columns (one 16-byte load), so a wavefront reads a
//     source row segment as ONE coalesced 1 KiB global_load_dwordx4 and the
//     neighbour sum never crosses lanes;
//   * rowptr/colidx/vals are wave-uniform -> scalar loads (s_load), no VGPR traffic;
//   * neighbours are fetched four at a time (4 x 1 KiB in flight per wavefront);
//   * bias and the three gates of the graph sit in registers (each lane only needs
//     its own 4 columns);
//   * the pooled max is lane-local over a wavefront's rows, then crosses the 4
//     wavefronts once through 4 KiB of LDS.
// A graph's rows are contiguous (96 KiB at T=32,F=768), so the ~deg re-reads of a
// source row are served by the XCD's L2; HBM sees each byte of Hd once.
//
// Algorithmic bytes per layer: 2*4*N*F (Hd in, out) + 4*(N+1) + 4*nnz
// + 4*B*F per gate + 4*F.
#include "common.h"

#include <hip/hip_fp16.h>

#include <cfloat>

namespace ggcn {
namespace {

constexpr int kWaves = 4;

// VEC consecutive feature elements of one row as one global access: fp32 x4 / fp16 x8 = 16 B
// (one coalesced 1 KiB read per wavefront), or a single element for unaligned shapes.
template <typename E, int VEC>
struct Seg;
template <>
struct Seg<float, 4> {
    static __device__ __forceinline__ void load(const float *p, float (&v)[4])
    {
        const float4 t = *reinterpret_cast<const float4 *>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void store(float *p, const float (&v)[4])
    {
        *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
};
template <>
struct Seg<float, 1> {
    static __device__ __forceinline__ void load(const float *p, float (&v)[1]) { v[0] = *p; }
    static __device__ __forceinline__ void store(float *p, const float (&v)[1]) { *p = v[0]; }
};
template <>
struct Seg<__half, 8> {
    static __device__ __forceinline__ void load(const __half *p, float (&v)[8])
    {
        const uint4 t = *reinterpret_cast<const uint4 *>(p);
        const __half2 *h = reinterpret_cast<const __half2 *>(&t);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float2 f = __half22float2(h[i]);
            v[2 * i] = f.x;
            v[2 * i + 1] = f.y;
        }
    }
    static __device__ __forceinline__ void store(__half *p, const float (&v)[8])
    {
        uint4 t;
        __half2 *h = reinterpret_cast<__half2 *>(&t);
#pragma unroll
        for (int i = 0; i < 4; ++i) h[i] = __floats2half2_rn(v[2 * i], v[2 * i + 1]);
        *reinterpret_cast<uint4 *>(p) = t;
    }
};
template <>
struct Seg<__half, 4> {  // 8-B accesses: half the registers of the x8 form -> twice the wavefronts in flight
    static __device__ __forceinline__ void load(const __half *p, float (&v)[4])
    {
        const uint2 t = *reinterpret_cast<const uint2 *>(p);
        const __half2 *h = reinterpret_cast<const __half2 *>(&t);
        const float2 a = __half22float2(h[0]), b = __half22float2(h[1]);
        v[0] = a.x; v[1] = a.y; v[2] = b.x; v[3] = b.y;
    }
    static __device__ __forceinline__ void store(__half *p, const float (&v)[4])
    {
        uint2 t;
        __half2 *h = reinterpret_cast<__half2 *>(&t);
        h[0] = __floats2half2_rn(v[0], v[1]);
        h[1] = __floats2half2_rn(v[2], v[3]);
        *reinterpret_cast<uint2 *>(p) = t;
    }
};
template <>
struct Seg<__half, 1> {
    static __device__ __forceinline__ void load(const __half *p, float (&v)[1]) { v[0] = __half2float(*p); }
    static __device__ __forceinline__ void store(__half *p, const float (&v)[1]) { *p = __float2half_rn(v[0]); }
};

// max for floats through integer atomics (order-independent, hence deterministic): values with a
// clear sign bit order like signed ints, values with the sign bit set like unsigned ints reversed.
// -inf is the identity.  The branch is on the SIGN BIT, not on v >= 0: -0.0 (what y * gate gives for
// a gate that dropout zeroed, bert_amir5.py:623-625) must take the unsigned-min side, where
// 0x80000000 beats the -inf preset 0xFF800000; as a signed max it is INT_MIN and never wins.
__device__ __forceinline__ void atomic_max_float(float *p, float v)
{
    if (__float_as_int(v) >= 0) atomicMax(reinterpret_cast<int *>(p), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned int *>(p), __float_as_uint(v));
}

template <int VEC>
__device__ __forceinline__ void load_f32(const float *p, float (&v)[VEC])
{
    if constexpr (VEC % 4 == 0) {
#pragma unroll
        for (int q = 0; q < VEC / 4; ++q) {
            const float4 t = reinterpret_cast<const float4 *>(p)[q];
            v[4 * q] = t.x; v[4 * q + 1] = t.y; v[4 * q + 2] = t.z; v[4 * q + 3] = t.w;
        }
    } else {
#pragma unroll
        for (int k = 0; k < VEC; ++k) v[k] = p[k];
    }
}

// grid.x = B * n_slabs ; block = 256.  slab = 64*VEC columns.  E = feature element type
// (float, or __half with fp32 accumulation: BASELINE configs[3]); bias, gates, pools are fp32.
// NORM = true: the forward (divide by rowsum+1, bias, gates, pools).  NORM = false: the plain
// weighted sum out[i] = sum_e vals[e] * src_scale[colidx[e]] * Hd[colidx[e]] used by the backward
// pass (the transposed adjacency applied to D.dY: src_scale = 1/(rowsum+1) of the SOURCE row).
// TILED: the graph's T x (64*VEC) slab of Hd is first copied into LDS (every source row read
// from HBM/L2 exactly once, T <= kTiledMaxT) and the neighbour sum reads LDS.  PMC on the direct
// form at config 2: 910 MB fetched for 403 MB of Hd -- the ~deg re-reads of a row miss the XCD's
// 4 MiB L2 half of the time because 256 resident workgroups x 32 KiB of slab exceed it.
constexpr int kChunkRows = 16;   // destination rows per workgroup when a graph is cut into chunks
constexpr int kTiledMaxT = 48;  // static 8-16 KiB + 48 KiB of tile stays within the default 64 KiB LDS limit

template <typename E, int VEC, bool HAS_VALS, bool NORM = true, bool TILED = false>
__global__ __launch_bounds__(256) void aggregate_rows(
    const E *__restrict__ Hd, int64_t ldh, const int32_t *__restrict__ rowptr,
    const int32_t *__restrict__ colidx, const float *__restrict__ vals,
    const float *__restrict__ src_scale,
    const float *__restrict__ bias, int n_graphs, int T, int F, int n_slabs, int n_chunks, int chunk_rows,
    const float *__restrict__ store_gate, const float *__restrict__ pool_gate_a,
    const float *__restrict__ pool_gate_b, E *__restrict__ out, int64_t ldo,
    float *__restrict__ pool_a, float *__restrict__ pool_b)
{
    constexpr int kSlab = kWave * VEC;
    __shared__ float red[2][kWaves][kSlab];
    extern __shared__ __attribute__((aligned(16))) char tile_raw[];  // TILED: [T][kSlab] of E
    E *tile = reinterpret_cast<E *>(tile_raw);

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // block -> (graph b, row chunk, column slab).  Long graphs (T > kTiledMaxT) are cut into
    // chunks of chunk_rows destination rows so that 256 x 512-token graphs still fill the chip
    // (BASELINE configs[3]); the pooled max of a chunked graph is combined with atomics.
    const int per_graph = n_slabs * n_chunks;
    int b, rem;
    if (n_chunks == 1) {
        b = blockIdx.x / per_graph;
        rem = blockIdx.x - b * per_graph;
    } else {
        // XCD-affine order (ids congruent mod 8 share an XCD, observed dispatch; speed only): an
        // XCD walks through ITS graphs one after the other, all pieces of a graph back to back, so
        // only a few graphs' feature blocks (T x F, 1-2 MiB at config 4) compete for its 4 MiB L2.
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        b = (slot / per_graph) * 8 + xcd;
        rem = slot % per_graph;
        if (b >= n_graphs) return;  // whole workgroup, before any barrier
    }
    const int chunk = rem % n_chunks;       // chunk fastest: the chunks of one column slab share rows
    const int slab = rem / n_chunks;
    const int t_begin = chunk * chunk_rows;
    const int t_end = (t_begin + chunk_rows < T) ? t_begin + chunk_rows : T;
    const int col = slab * kSlab + lane * VEC;
    const bool live = col < F;  // F % VEC == 0 is checked on the host

    float vb[VEC], vsg[VEC], vga[VEC], vgb[VEC], pa[VEC], pb[VEC];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
        vb[k] = 0.0f; vsg[k] = 1.0f; vga[k] = 1.0f; vgb[k] = 1.0f;
        pa[k] = -INFINITY; pb[k] = -INFINITY;
    }
    if (live) {  // fp32 side inputs of this lane's columns: 16-B loads when VEC allows
        const int64_t g = (int64_t)b * F + col;
        if (bias) load_f32<VEC>(bias + col, vb);
        if (store_gate) load_f32<VEC>(store_gate + g, vsg);
        if (pool_gate_a) load_f32<VEC>(pool_gate_a + g, vga);
        if (pool_gate_b) load_f32<VEC>(pool_gate_b + g, vgb);
    }

    const E *hcol = Hd + col;
    if constexpr (TILED) {
        // copy phase: wavefront w brings rows w, w+4, ... (one coalesced 1 KiB load each, all in
        // flight together), then one barrier; raw element type, converted when read back
        constexpr int kMaxRows = kTiledMaxT / kWaves;
        float stage[kMaxRows][VEC];
#pragma unroll
        for (int q = 0; q < kMaxRows; ++q) {
            const int t = wave + q * kWaves;
#pragma unroll
            for (int k = 0; k < VEC; ++k) stage[q][k] = 0.0f;
            if (t < T && live) Seg<E, VEC>::load(hcol + ((int64_t)b * T + t) * ldh, stage[q]);
        }
#pragma unroll
        for (int q = 0; q < kMaxRows; ++q) {
            const int t = wave + q * kWaves;
            if (t < T) Seg<E, VEC>::store(tile + (size_t)t * kSlab + lane * VEC, stage[q]);
        }
        __syncthreads();
    }
    // Two destination rows per wavefront at a time (t and t+4): their index loads and their 2 x 4
    // source-row reads are issued together, which halves the dependent-latency chains per row
    // (rowptr -> colidx -> features) that bound this kernel (measured 217 -> see DESIGN.md).
    for (int t = t_begin + wave; t < t_end; t += 2 * kWaves) {
        const bool two = t + kWaves < t_end;  // wave-uniform
        int64_t row[2];
        int e[2], end[2], cnt[2];
        float acc[2][VEC], wsum[2];
        row[0] = (int64_t)b * T + t;
        row[1] = two ? row[0] + kWaves : row[0];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            e[r] = rowptr[row[r]];
            end[r] = rowptr[row[r] + 1];
            cnt[r] = end[r] - e[r];
            wsum[r] = 0.0f;
#pragma unroll
            for (int k = 0; k < VEC; ++k) acc[r][k] = 0.0f;
        }
        if (!two) end[1] = e[1];  // no second row: zero trips
        while (e[0] < end[0] || e[1] < end[1]) {
            int c[2][4];
            bool v[2][4];
            float w[2][4];
            float h[2][4][VEC];
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) {  // wave-uniform neighbour ids / weights: scalar loads
                    v[r][q] = e[r] + q < end[r];
                    c[r][q] = colidx[v[r][q] ? e[r] + q : 0];
                    w[r][q] = 1.0f;
                    if constexpr (HAS_VALS) w[r][q] = vals[v[r][q] ? e[r] + q : 0];
                    if constexpr (!NORM) w[r][q] *= src_scale[c[r][q]];
                }
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int q = 0; q < 4; ++q) {  // 8 independent row-segment reads in flight per wavefront
#pragma unroll
                    for (int k = 0; k < VEC; ++k) h[r][q][k] = 0.0f;
                    if constexpr (TILED) {
                        if (v[r][q]) Seg<E, VEC>::load(tile + (size_t)(c[r][q] - b * T) * kSlab + lane * VEC, h[r][q]);
                    } else {
                        if (live && v[r][q]) Seg<E, VEC>::load(hcol + (int64_t)c[r][q] * ldh, h[r][q]);
                    }
                }
#pragma unroll
            for (int r = 0; r < 2; ++r) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (v[r][q]) {  // wave-uniform; a masked-out neighbour must not contribute 0*inf
                        if constexpr (HAS_VALS || !NORM) {
#pragma unroll
                            for (int k = 0; k < VEC; ++k) acc[r][k] = fmaf(w[r][q], h[r][q][k], acc[r][k]);
                            wsum[r] += w[r][q];
                        } else {
#pragma unroll
                            for (int k = 0; k < VEC; ++k) acc[r][k] += h[r][q][k];
                        }
                    }
                }
                e[r] += 4;
            }
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (r == 1 && !two) break;
            // gcn.py:35,41: divide by rowsum+1 -- one IEEE reciprocal per row (the denominator is
            // wave-uniform), then a multiply per element: <= 1 ulp from the reference's division
            const float inv = NORM ? 1.0f / ((HAS_VALS ? wsum[r] : (float)cnt[r]) + 1.0f) : 1.0f;
            if (live) {
                float y[VEC], o[VEC];
#pragma unroll
                for (int k = 0; k < VEC; ++k) {
                    y[k] = NORM ? acc[r][k] * inv + vb[k] : acc[r][k];  // gcn.py:41,43
                    o[k] = y[k] * vsg[k];
                    pa[k] = fmaxf(pa[k], y[k] * vga[k]);
                    pb[k] = fmaxf(pb[k], y[k] * vgb[k]);
                }
                if (out) Seg<E, VEC>::store(out + row[r] * ldo + col, o);
            }
        }
    }

    if (pool_a || pool_b) {
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            red[0][wave][lane * VEC + k] = pa[k];
            red[1][wave][lane * VEC + k] = pb[k];
        }
        __syncthreads();
        for (int tl = threadIdx.x; tl < kSlab; tl += 256) {
            const int tcol = slab * kSlab + tl;
            if (tcol >= F) break;
            float ma = red[0][0][tl], mb = red[1][0][tl];
#pragma unroll
            for (int w = 1; w < kWaves; ++w) {
                ma = fmaxf(ma, red[0][w][tl]);
                mb = fmaxf(mb, red[1][w][tl]);
            }
            if (n_chunks == 1) {
                if (pool_a) pool_a[(int64_t)b * F + tcol] = ma;
                if (pool_b) pool_b[(int64_t)b * F + tcol] = mb;
            } else {  // pools were preset to -inf by the launcher
                if (pool_a) atomic_max_float(pool_a + (int64_t)b * F + tcol, ma);
                if (pool_b) atomic_max_float(pool_b + (int64_t)b * F + tcol, mb);
            }
        }
    }
}

bool side_aligned(const float *bias, const float *sg, const float *ga, const float *gb)
{
    return (!bias || aligned16(bias)) && (!sg || aligned16(sg)) && (!ga || aligned16(ga)) && (!gb || aligned16(gb));
}

// ---- long graphs (kTiledMaxT < T <= kNarrowMaxT): the graph in LDS, 128 bytes of columns at a time ----
// The direct form re-reads every source row ~deg times from L2 (config 4, PMC: HBM traffic is the
// compulsory 0.54 GB but 2.2 GB cross L2 -> CU and the wavefronts wait 2/3 of their cycles).  Here
// a workgroup owns (graph, 128-byte column slab): it copies the graph's T x 128 B slab into LDS once
// (<= 96 KiB, two workgroups per CU at T = 512) and every neighbour sum reads LDS.  8 lanes of 16 B
// cover a row of the slab, so a wavefront works on 8 destination rows at once (32 per workgroup
// step) with per-row edge lists; pooled maxima stay in registers and meet in LDS at the end: one
// workgroup sees all T rows, so no atomics and no pool preset.  The graph's CSR (row pointers and
// 16-bit local column ids, <= kIdxCap edges) is staged in LDS as well, so the dependent chain
// rowptr -> colidx -> features of every row runs on LDS latency, not on L2 latency; a graph with
// more edges reads its indices from global memory (same code path, workgroup-uniform switch).
// Used for fp16 features (config 4: 350 -> 250 us); with fp32 a 128-byte slab is only 32 columns and
// the chunked direct form above stays faster (317 vs 344 us), so fp32 keeps it.
constexpr int kNarrowMaxT = 768;  // 96 KiB of slab
constexpr int kIdxCap = 4096;     // edges per graph whose column ids are staged (8 KiB as uint16)
// acc[0..7] += w * (the 8 halves of v): v_fma_mix_f32 reads the fp16 operand straight from its half of the dword
// (fp32 accumulate, no separate conversion: 8 VALU per 16 bytes instead of 16)
__device__ __forceinline__ void fma_half8(const uint4 &v, float w, float (&acc)[8])
{
    const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc[2 * q]) : "v"(d[q]), "v"(w));
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc[2 * q + 1]) : "v"(d[q]), "v"(w));
    }
}

// NW: wavefronts per workgroup (4, or 8 for T >= 256: the slab's LDS is the same, so twice the wavefronts share a CU's
// two resident workgroups -- the row loop is a chain of dependent LDS reads and lives on wavefront count)
template <typename E, bool HAS_VALS, bool NORM, int NW, int MAXQ>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 4 : 2) void aggregate_narrow(   // two workgroups per CU either way
    const E *__restrict__ Hd, int64_t ldh, const int32_t *__restrict__ rowptr,
    const int32_t *__restrict__ colidx, const float *__restrict__ vals, const float *__restrict__ src_scale,
    const float *__restrict__ bias, int n_graphs, int T, int F, int n_slabs,
    const float *__restrict__ store_gate, const float *__restrict__ pool_gate_a,
    const float *__restrict__ pool_gate_b, E *__restrict__ out, int64_t ldo,
    float *__restrict__ pool_a, float *__restrict__ pool_b)
{
    constexpr int EPL = 16 / (int)sizeof(E);  // elements per lane: 4 fp32 / 8 fp16
    constexpr int kCols = 8 * EPL;            // columns per slab (128 bytes)
    __shared__ float red[2][NW][kCols];
    __shared__ int s_rp[kNarrowMaxT + 1];        // row pointers relative to the graph's first edge
    __shared__ unsigned short s_col[kIdxCap];    // column ids relative to the graph's first node
    extern __shared__ __attribute__((aligned(16))) char tile_raw[];  // [T][128 B]
    E *tile = reinterpret_cast<E *>(tile_raw);

    // XCD-affine order as in the chunked form: an XCD walks through its own graphs one after the other
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int b = (slot / n_slabs) * 8 + xcd;
    const int slab = slot % n_slabs;
    if (b >= n_graphs) return;  // whole workgroup, before any barrier
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int grp = lane >> 3, piece = lane & 7;
    const int col = slab * kCols + piece * EPL;
    const bool live = col < F;  // F % EPL == 0 is checked on the host
    const int64_t node0 = (int64_t)b * T;

    float vb[EPL], vsg[EPL], vga[EPL], vgb[EPL], pa[EPL], pb[EPL];
#pragma unroll
    for (int k = 0; k < EPL; ++k) {
        vb[k] = 0.0f; vsg[k] = 1.0f; vga[k] = 1.0f; vgb[k] = 1.0f;
        pa[k] = -INFINITY; pb[k] = -INFINITY;
    }
    if (live) {
        const int64_t g = (int64_t)b * F + col;
        if (bias) load_f32<EPL>(bias + col, vb);
        if (store_gate) load_f32<EPL>(store_gate + g, vsg);
        if (pool_gate_a) load_f32<EPL>(pool_gate_a + g, vga);
        if (pool_gate_b) load_f32<EPL>(pool_gate_b + g, vgb);
    }

    // ---- the slab's rows leave for LDS first (up to 24 x 16 B per thread in flight), the graph's CSR
    // is fetched and staged while they travel ----
    const int r_first = wave * 8 + grp;
    constexpr int kRowsPerStep = 8 * NW;
    constexpr int kMaxSteps = MAXQ;   // ceil(T / kRowsPerStep) <= MAXQ (the launcher picks): 16-byte loads in flight per thread
    uint4 st4[kMaxSteps];
#pragma unroll
    for (int q = 0; q < kMaxSteps; ++q) {
        const int r = r_first + kRowsPerStep * q;
        st4[q] = make_uint4(0u, 0u, 0u, 0u);
        if (r < T && live) st4[q] = *reinterpret_cast<const uint4 *>(Hd + (node0 + r) * ldh + col);
    }
    const int e_base = rowptr[node0];
    const int nnz_g = rowptr[node0 + T] - e_base;
    const bool staged = nnz_g <= kIdxCap;  // workgroup-uniform
    for (int i = threadIdx.x; i <= T; i += 64 * NW) s_rp[i] = rowptr[node0 + i] - e_base;
    if (staged)
        for (int j = threadIdx.x; j < nnz_g; j += 64 * NW) s_col[j] = (unsigned short)(colidx[e_base + j] - (int)node0);
#pragma unroll
    for (int q = 0; q < kMaxSteps; ++q) {
        const int r = r_first + kRowsPerStep * q;
        if (r < T) *reinterpret_cast<uint4 *>(tile_raw + (size_t)r * 128 + piece * 16) = st4[q];
    }
    __syncthreads();

    // ---- neighbour sums out of LDS: this lane group's rows r_first, r_first + 32, ... ----
    const int32_t *cg = colidx + e_base;
    const float *vg = HAS_VALS ? vals + e_base : nullptr;
    auto col_of = [&](int e) -> int { return staged ? (int)s_col[e] : cg[e] - (int)node0; };  // local source row
    for (int r = r_first; r < T; r += kRowsPerStep) {   // rows differ per lane group: plain divergent control flow
        const int64_t node = node0 + r;
        int e = s_rp[r];
        const int end = s_rp[r + 1];
        const int cnt = end - e;
        float acc[EPL], wsum = 0.0f;
#pragma unroll
        for (int k = 0; k < EPL; ++k) acc[k] = 0.0f;
        for (; e + 1 < end; e += 2) {  // two source rows in flight
            const int c0 = col_of(e), c1 = col_of(e + 1);
            float w0 = 1.0f, w1 = 1.0f;
            if constexpr (HAS_VALS) { w0 = vg[e]; w1 = vg[e + 1]; }
            if constexpr (!NORM) { w0 *= src_scale[node0 + c0]; w1 *= src_scale[node0 + c1]; }
            if constexpr (std::is_same<E, __half>::value) {   // fp16 slab: fused convert-and-add
                const uint4 r0 = *reinterpret_cast<const uint4 *>(tile + (size_t)c0 * kCols + piece * EPL);
                const uint4 r1 = *reinterpret_cast<const uint4 *>(tile + (size_t)c1 * kCols + piece * EPL);
                fma_half8(r0, w0, acc);
                fma_half8(r1, w1, acc);
                if constexpr (HAS_VALS || !NORM) { wsum += w0; wsum += w1; }
                continue;
            }
            float h0[EPL], h1[EPL];
            Seg<E, EPL>::load(tile + (size_t)c0 * kCols + piece * EPL, h0);
            Seg<E, EPL>::load(tile + (size_t)c1 * kCols + piece * EPL, h1);
            if constexpr (HAS_VALS || !NORM) {
#pragma unroll
                for (int k = 0; k < EPL; ++k) acc[k] = fmaf(w1, h1[k], fmaf(w0, h0[k], acc[k]));
                wsum += w0;
                wsum += w1;
            } else {
#pragma unroll
                for (int k = 0; k < EPL; ++k) acc[k] = (acc[k] + h0[k]) + h1[k];
            }
        }
        if (e < end) {
            const int c0 = col_of(e);
            float w0 = 1.0f;
            if constexpr (HAS_VALS) w0 = vg[e];
            if constexpr (!NORM) w0 *= src_scale[node0 + c0];
            if constexpr (std::is_same<E, __half>::value) {
                const uint4 r0 = *reinterpret_cast<const uint4 *>(tile + (size_t)c0 * kCols + piece * EPL);
                fma_half8(r0, w0, acc);
                if constexpr (HAS_VALS || !NORM) wsum += w0;
            } else {
                float h0[EPL];
                Seg<E, EPL>::load(tile + (size_t)c0 * kCols + piece * EPL, h0);
                if constexpr (HAS_VALS || !NORM) {
#pragma unroll
                    for (int k = 0; k < EPL; ++k) acc[k] = fmaf(w0, h0[k], acc[k]);
                    wsum += w0;
                } else {
#pragma unroll
                    for (int k = 0; k < EPL; ++k) acc[k] += h0[k];
                }
            }
        }
        const float inv = NORM ? 1.0f / ((HAS_VALS ? wsum : (float)cnt) + 1.0f) : 1.0f;  // gcn.py:35
        if (live) {
            float o[EPL];
#pragma unroll
            for (int k = 0; k < EPL; ++k) {
                const float y = NORM ? acc[k] * inv + vb[k] : acc[k];  // gcn.py:41,43
                o[k] = y * vsg[k];
                pa[k] = fmaxf(pa[k], y * vga[k]);
                pb[k] = fmaxf(pb[k], y * vgb[k]);
            }
            if (out) Seg<E, EPL>::store(out + node * ldo + col, o);
        }
    }

    if (pool_a || pool_b) {
        // the 8 lane groups of a wavefront hold the same columns: butterfly over the group index
#pragma unroll
        for (int k = 0; k < EPL; ++k) {
#pragma unroll
            for (int d = 8; d <= 32; d <<= 1) {
                pa[k] = fmaxf(pa[k], __shfl_xor(pa[k], d));
                pb[k] = fmaxf(pb[k], __shfl_xor(pb[k], d));
            }
        }
        if (grp == 0) {
#pragma unroll
            for (int k = 0; k < EPL; ++k) {
                red[0][wave][piece * EPL + k] = pa[k];
                red[1][wave][piece * EPL + k] = pb[k];
            }
        }
        __syncthreads();
        if (threadIdx.x < kCols) {
            const int tcol = slab * kCols + threadIdx.x;
            if (tcol < F) {
                float ma = red[0][0][threadIdx.x], mb = red[1][0][threadIdx.x];
#pragma unroll
                for (int w = 1; w < NW; ++w) {
                    ma = fmaxf(ma, red[0][w][threadIdx.x]);
                    mb = fmaxf(mb, red[1][w][threadIdx.x]);
                }
                if (pool_a) pool_a[(int64_t)b * F + tcol] = ma;
                if (pool_b) pool_b[(int64_t)b * F + tcol] = mb;
            }
        }
    }
}

// true (and launched) when the narrow-tile form applies: 16-byte pieces, kTiledMaxT < T <= kNarrowMaxT
template <typename E, bool NORM>
bool launch_narrow(const E *Hd, int64_t ldh, const int32_t *rowptr, const int32_t *colidx, const float *vals,
                   const float *src_scale, const float *bias, int B, int T, int F, const float *sg, const float *ga,
                   const float *gb, E *out, int64_t ldo, float *pa, float *pb, hipStream_t st, int &rc)
{
    constexpr int EPL = 16 / (int)sizeof(E);
    if (T <= kTiledMaxT || T > kNarrowMaxT) return false;
    if ((F % EPL) || (ldh % EPL) || !aligned16(Hd) || (out && ((ldo % EPL) || !aligned16(out)))) return false;
    if (!side_aligned(bias, sg, ga, gb)) return false;
    const int n_slabs = (F + 8 * EPL - 1) / (8 * EPL);
    const int64_t blocks = ((int64_t)B + 7) / 8 * 8 * n_slabs;
    if (blocks > (int64_t)INT32_MAX) return false;
    const size_t lds = (size_t)T * 128;
    const int nw = T >= 256 ? 8 : 4;
    auto go = [&](auto kern) {
        // dynamic LDS above the default 64 KiB limit needs the attribute (cheap; set on every launch: one static flag
        // per lambda would be shared by the instantiations)
        {
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      kNarrowMaxT * 128);
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * nw), lds, st, Hd, ldh, rowptr, colidx, vals, src_scale, bias,
                           B, T, F, n_slabs, sg, ga, gb, out, ldo, pa, pb);
    };
    if (nw == 8 && T > 512) {          // 513..768 rows: 12 steps of 64 rows
        if (vals) go(aggregate_narrow<E, true, NORM, 8, 12>);
        else go(aggregate_narrow<E, false, NORM, 8, 12>);
    } else if (nw == 8) {              // 256..512 rows: 8 steps of 64 rows
        if (vals) go(aggregate_narrow<E, true, NORM, 8, 8>);
        else go(aggregate_narrow<E, false, NORM, 8, 8>);
    } else {                           // < 256 rows: 8 steps of 32 rows
        if (vals) go(aggregate_narrow<E, true, NORM, 4, 8>);
        else go(aggregate_narrow<E, false, NORM, 4, 8>);
    }
    rc = check_launch(NORM ? "ggcn_aggregate" : "ggcn_aggregate_t");
    return true;
}

template <typename E, int VEC>
int launch(const E *Hd, int64_t ldh, const int32_t *rowptr, const int32_t *colidx,
           const float *vals, const float *bias, int B, int T, int F, const float *sg,
           const float *ga, const float *gb, E *out, int64_t ldo, float *pa, float *pb,
           hipStream_t st)
{
    const int slab = kWave * VEC;
    const int n_slabs = (F + slab - 1) / slab;
    const bool tiled = (VEC > 1) && (T <= kTiledMaxT);
    const int chunk_rows = tiled ? T : kChunkRows;
    const int n_chunks = tiled ? 1 : (T + kChunkRows - 1) / kChunkRows;
    const int64_t blocks = (n_chunks == 1 ? (int64_t)B : ((int64_t)B + 7) / 8 * 8) * n_slabs * n_chunks;
    if (blocks > (int64_t)INT32_MAX) return fail(GGCN_EUNSUPPORTED, "ggcn_aggregate: grid too large");
    if (n_chunks > 1) {  // -inf = 0xFF800000: identity of the atomic max (a memset node, graph-capturable)
        if (pa && hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(pa), (int)0xFF800000, (size_t)B * F, st) != hipSuccess)
            return fail(GGCN_ELAUNCH, "ggcn_aggregate: pool preset failed");
        if (pb && hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(pb), (int)0xFF800000, (size_t)B * F, st) != hipSuccess)
            return fail(GGCN_ELAUNCH, "ggcn_aggregate: pool preset failed");
    }
    const size_t lds = tiled ? (size_t)T * slab * sizeof(E) : 0;
#define GGCN_AGG(HV, TL)                                                                                      \
    hipLaunchKernelGGL((aggregate_rows<E, VEC, HV, true, TL>), dim3((unsigned)blocks), dim3(256), lds, st, Hd, \
                       ldh, rowptr, colidx, vals, nullptr, bias, B, T, F, n_slabs, n_chunks, chunk_rows, sg, ga, \
                       gb, out, ldo, pa, pb)
    if (vals && tiled) GGCN_AGG(true, true);
    else if (vals) GGCN_AGG(true, false);
    else if (tiled) GGCN_AGG(false, true);
    else GGCN_AGG(false, false);
#undef GGCN_AGG
    return check_launch("ggcn_aggregate");
}

template <int VEC>
int launch_t(const float *G, int64_t ldg, const int32_t *rowptr, const int32_t *colidx, const float *vals,
             const float *src_scale, int B, int T, int F, float *out, int64_t ldo, hipStream_t st)
{
    const int slab = kWave * VEC;
    const int n_slabs = (F + slab - 1) / slab;
    const bool tiled = (VEC > 1) && (T <= kTiledMaxT);
    const int chunk_rows = tiled ? T : kChunkRows;
    const int n_chunks = tiled ? 1 : (T + kChunkRows - 1) / kChunkRows;
    const int64_t blocks = (n_chunks == 1 ? (int64_t)B : ((int64_t)B + 7) / 8 * 8) * n_slabs * n_chunks;
    if (blocks > (int64_t)INT32_MAX) return fail(GGCN_EUNSUPPORTED, "ggcn_aggregate_t: grid too large");
    const size_t lds = tiled ? (size_t)T * slab * sizeof(float) : 0;
#define GGCN_AGGT(HV, TL)                                                                                        \
    hipLaunchKernelGGL((aggregate_rows<float, VEC, HV, false, TL>), dim3((unsigned)blocks), dim3(256), lds, st, G, \
                       ldg, rowptr, colidx, vals, src_scale, nullptr, B, T, F, n_slabs, n_chunks, chunk_rows, nullptr, \
                       nullptr, nullptr, out, ldo, nullptr, nullptr)
    if (vals && tiled) GGCN_AGGT(true, true);
    else if (vals) GGCN_AGGT(true, false);
    else if (tiled) GGCN_AGGT(false, true);
    else GGCN_AGGT(false, false);
#undef GGCN_AGGT
    return check_launch("ggcn_aggregate_t");
}

// inv[i] = 1 / (sum of row i's values + 1)   (gcn.py:35), one thread per node
__global__ __launch_bounds__(256) void inv_denominator_kernel(const int32_t *__restrict__ rowptr,
                                                              const float *__restrict__ vals, int64_t n,
                                                              float *__restrict__ inv)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 1.0f;
    if (vals) {
        for (int e = rowptr[i]; e < rowptr[i + 1]; ++e) s += vals[e];
    } else {
        s += (float)(rowptr[i + 1] - rowptr[i]);
    }
    inv[i] = 1.0f / s;
}

int check_args(const void *Hd, int64_t ldh, const int32_t *rowptr, const int32_t *colidx, int B, int T, int F,
               const void *out, int64_t ldo, const float *pool_a, const float *pool_b)
{
    if (!Hd || !rowptr || !colidx) return fail(GGCN_EINVAL, "ggcn_aggregate: null input pointer");
    if (B <= 0 || T <= 0 || F <= 0)
        return fail(GGCN_EINVAL, "ggcn_aggregate: B=%d T=%d F=%d must be positive", B, T, F);
    if (!out && !pool_a && !pool_b) return fail(GGCN_EINVAL, "ggcn_aggregate: no output requested");
    if (ldh < F || (out && ldo < F))
        return fail(GGCN_EINVAL, "ggcn_aggregate: leading dimension smaller than F=%d", F);
    if ((int64_t)B * T >= (int64_t)INT32_MAX)
        return fail(GGCN_EUNSUPPORTED, "ggcn_aggregate: B*T does not fit int32 node ids");
    return GGCN_OK;
}

}  // namespace

int aggregate(const float *Hd, int64_t ldh, const int32_t *rowptr, const int32_t *colidx,
              const float *vals, const float *bias, int B, int T, int F, const float *store_gate,
              const float *pool_gate_a, const float *pool_gate_b, float *out, int64_t ldo,
              float *pool_a, float *pool_b, hipStream_t st)
{
    if (int rc = check_args(Hd, ldh, rowptr, colidx, B, T, F, out, ldo, pool_a, pool_b)) return rc;
    const bool vec = (F % 4 == 0) && (ldh % 4 == 0) && aligned16(Hd) && (!out || ((ldo % 4 == 0) && aligned16(out))) &&
                     side_aligned(bias, store_gate, pool_gate_a, pool_gate_b);
    if (vec)
        return launch<float, 4>(Hd, ldh, rowptr, colidx, vals, bias, B, T, F, store_gate, pool_gate_a,
                                pool_gate_b, out, ldo, pool_a, pool_b, st);
    return launch<float, 1>(Hd, ldh, rowptr, colidx, vals, bias, B, T, F, store_gate, pool_gate_a,
                            pool_gate_b, out, ldo, pool_a, pool_b, st);
}

int aggregate_t(const float *G, int64_t ldg, const int32_t *rowptr_t, const int32_t *colidx_t,
                const float *vals_t, const float *src_scale, int B, int T, int F, float *out, int64_t ldo,
                hipStream_t st)
{
    if (int rc = check_args(G, ldg, rowptr_t, colidx_t, B, T, F, out, ldo, nullptr, nullptr)) return rc;
    if (!src_scale || !out) return fail(GGCN_EINVAL, "ggcn_aggregate_t: null pointer");
    const bool vec = (F % 4 == 0) && (ldg % 4 == 0) && aligned16(G) && (ldo % 4 == 0) && aligned16(out);
    return vec ? launch_t<4>(G, ldg, rowptr_t, colidx_t, vals_t, src_scale, B, T, F, out, ldo, st)
               : launch_t<1>(G, ldg, rowptr_t, colidx_t, vals_t, src_scale, B, T, F, out, ldo, st);
}

int inv_denominators(const int32_t *rowptr, const float *vals, int64_t n, float *inv, hipStream_t st)
{
    if (!rowptr || !inv || n <= 0) return fail(GGCN_EINVAL, "ggcn_inv_denominators: bad argument");
    hipLaunchKernelGGL(inv_denominator_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, rowptr, vals, n, inv);
    return check_launch("ggcn_inv_denominators");
}

int aggregate_h(const void *Hd, int64_t ldh, const int32_t *rowptr, const int32_t *colidx,
                const float *vals, const float *bias, int B, int T, int F, const float *store_gate,
                const float *pool_gate_a, const float *pool_gate_b, void *out, int64_t ldo,
                float *pool_a, float *pool_b, hipStream_t st)
{
    if (int rc = check_args(Hd, ldh, rowptr, colidx, B, T, F, out, ldo, pool_a, pool_b)) return rc;
    const __half *h = static_cast<const __half *>(Hd);
    __half *o = static_cast<__half *>(out);
    {
        int rc = GGCN_OK;
        if (launch_narrow<__half, true>(h, ldh, rowptr, colidx, vals, nullptr, bias, B, T, F, store_gate, pool_gate_a,
                                        pool_gate_b, o, ldo, pool_a, pool_b, st, rc))
            return rc;
    }
    const bool vec = (F % 8 == 0) && (ldh % 8 == 0) && aligned16(Hd) && (!out || ((ldo % 8 == 0) && aligned16(out))) &&
                     side_aligned(bias, store_gate, pool_gate_a, pool_gate_b);
#if defined(GGCN_HALF_VEC8)
    if (vec)
        return launch<__half, 8>(h, ldh, rowptr, colidx, vals, bias, B, T, F, store_gate, pool_gate_a,
                                 pool_gate_b, o, ldo, pool_a, pool_b, st);
#else
    // 8-B accesses (4 halves per lane) measured faster than 16-B ones here: the x8 form needs
    // 140 VGPRs (3 wavefronts per SIMD) and this kernel lives on latency hiding
    const bool vec4 = (F % 4 == 0) && (ldh % 4 == 0) && ((reinterpret_cast<uintptr_t>(Hd) & 7u) == 0) &&
                      (!out || ((ldo % 4 == 0) && ((reinterpret_cast<uintptr_t>(out) & 7u) == 0))) &&
                      side_aligned(bias, store_gate, pool_gate_a, pool_gate_b);
    (void)vec;
    if (vec4)
        return launch<__half, 4>(h, ldh, rowptr, colidx, vals, bias, B, T, F, store_gate, pool_gate_a,
                                 pool_gate_b, o, ldo, pool_a, pool_b, st);
#endif
    return launch<__half, 1>(h, ldh, rowptr, colidx, vals, bias, B, T, F, store_gate, pool_gate_a,
                             pool_gate_b, o, ldo, pool_a, pool_b, st);
}

}  // namespace ggcn
