// Backward of the gate / max-pool epilogue of one layer (models/bert_amir5.py:627-640 under
// train.py:120 `loss.backward()`), in ONE pass over the stored layer output.
//
// Forward (per graph g, feature f; y = the ungated layer output of gcn.py:45):
//     out[t]  = y[t] * sg          pa = max_t y[t] * ga          pb = max_t y[t] * gb
// Given d_out [N,F] (optional), d_pa, d_pb [B,F] (optional):
//     dY[t]   = d_out[t]*sg + [t = argmax_a] d_pa*ga + [t = argmax_b] d_pb*gb
//     d_sg    = sum_t d_out[t]*y[t]      d_ga = d_pa * y[argmax_a]      d_gb = d_pb * y[argmax_b]
//     d_bsum  = sum_t dY[t]              (per graph; ggcn_colsum over the graphs then gives db = sum_rows dY)
// y is recovered as out/sg when a store gate was applied (sigmoid gates are > 0; sg == 0 -> y := 0),
// argmax = the FIRST row attaining the maximum of y*g (torch.max's choice, bert_amir5.py:635).
// Mapping: workgroup = (graph, 256-column slab), thread = one column, loop over the T rows twice
// (find the maxima, then write dY): coalesced 1 KiB row segments, HBM-bound:
// 4*N*F (out) [+ 4*N*F d_out] read, 4*N*F (dY) written.
// DROP (training, bert_amir5.py:621-625): every gate carries a per-(token, feature) keep factor k[t] in {0, 1/(1-p)}
// (dropout_hash.h; the forward's epilogue draws the same ones): sg, ga, gb above become sg*ks[t], ga*ka[t], gb*kb[t].  A
// token whose STORE factor is 0 has out[t] = 0 and its y is not recoverable: it is taken as 0, which is exact whenever the
// pool gates in use share the store gate's stream or there is no store gate -- the block's two cases (:627-640).
#include "common.h"
#include "dropout_hash.h"

namespace ggcn {
namespace {

template <bool DROP>
__global__ __launch_bounds__(256) void gate_pool_backward_kernel(
    const float *__restrict__ out, int64_t ldo, const float *__restrict__ store_gate,
    const float *__restrict__ gate_a, const float *__restrict__ gate_b,
    const float *__restrict__ d_out, int64_t ldd, const float *__restrict__ d_pa,
    const float *__restrict__ d_pb, int T, int F, int n_slabs, float *__restrict__ dY, int64_t ldy,
    float *__restrict__ d_sg, float *__restrict__ d_ga, float *__restrict__ d_gb, float *__restrict__ d_bsum, DropSpec drop)
{
    const int b = blockIdx.x / n_slabs;
    const int f = (blockIdx.x - b * n_slabs) * 256 + threadIdx.x;
    if (f >= F) return;
    const int64_t gf = (int64_t)b * F + f;
    const float sg = store_gate ? store_gate[gf] : 1.0f;
    const float inv_sg = store_gate ? (sg != 0.0f ? 1.0f / sg : 0.0f) : 1.0f;
    const float ga = gate_a ? gate_a[gf] : 1.0f;
    const float gb = gate_b ? gate_b[gf] : 1.0f;
    const float dpa = d_pa ? d_pa[gf] : 0.0f;
    const float dpb = d_pb ? d_pb[gf] : 0.0f;
    const float *o = out + (int64_t)b * T * ldo + f;

    float best_a = -INFINITY, best_b = -INFINITY, ya = 0.0f, yb = 0.0f, acc_sg = 0.0f;
    int ia = 0, ib = 0;
    const uint32_t e0 = DROP ? (uint32_t)((int64_t)b * T * F + f) : 0u;   // element (node b*T + t, feature f) = e0 + t*F
    for (int t = 0; t < T; ++t) {
        float ks = 1.0f, ka = 1.0f, kb = 1.0f;
        if constexpr (DROP) {
            const uint32_t hh = drop_hash(e0 + (uint32_t)t * (uint32_t)F, drop.seed_lo, drop.seed_hi);
            ks = drop_keep(hh, drop.sel[0], drop.thr, drop.scale);
            ka = drop_keep(hh, drop.sel[1], drop.thr, drop.scale);
            kb = drop_keep(hh, drop.sel[2], drop.thr, drop.scale);
        }
        const float o_t = o[(int64_t)t * ldo];
        const float y = DROP ? (ks != 0.0f ? o_t * inv_sg / ks : 0.0f) : o_t * inv_sg;
        const float va = y * ga * ka, vb = y * gb * kb;
        if (va > best_a) { best_a = va; ia = t; ya = y * ka; }
        if (vb > best_b) { best_b = vb; ib = t; yb = y * kb; }
        if (d_out) acc_sg = fmaf(d_out[((int64_t)b * T + t) * ldd + f], y * ks, acc_sg);
    }
    float bsum = 0.0f;
    for (int t = 0; t < T; ++t) {
        float ks = 1.0f, ka = 1.0f, kb = 1.0f;
        if constexpr (DROP) {
            const uint32_t hh = drop_hash(e0 + (uint32_t)t * (uint32_t)F, drop.seed_lo, drop.seed_hi);
            ks = drop_keep(hh, drop.sel[0], drop.thr, drop.scale);
            ka = drop_keep(hh, drop.sel[1], drop.thr, drop.scale);
            kb = drop_keep(hh, drop.sel[2], drop.thr, drop.scale);
        }
        float g = d_out ? d_out[((int64_t)b * T + t) * ldd + f] * sg * ks : 0.0f;
        if (d_pa && t == ia) g = fmaf(dpa, ga * ka, g);
        if (d_pb && t == ib) g = fmaf(dpb, gb * kb, g);
        dY[((int64_t)b * T + t) * ldy + f] = g;
        bsum += g;
    }
    if (d_bsum) d_bsum[gf] = bsum;
    if (d_sg) d_sg[gf] = acc_sg;
    if (d_ga) d_ga[gf] = dpa * ya;
    if (d_gb) d_gb[gf] = dpb * yb;
}

// The same, four columns per thread (16-byte loads and stores: a quarter of the memory instructions; needs F % 4 == 0,
// leading dimensions multiples of 4 and 16-byte aligned pointers).  Component by component the arithmetic of the kernel
// above, in the same order: identical results.
template <bool DROP>
__global__ __launch_bounds__(256) void gate_pool_backward_kernel4(
    const float *__restrict__ out, int64_t ldo, const float *__restrict__ store_gate,
    const float *__restrict__ gate_a, const float *__restrict__ gate_b,
    const float *__restrict__ d_out, int64_t ldd, const float *__restrict__ d_pa,
    const float *__restrict__ d_pb, int T, int F, int n_slabs, float *__restrict__ dY, int64_t ldy,
    float *__restrict__ d_sg, float *__restrict__ d_ga, float *__restrict__ d_gb, float *__restrict__ d_bsum, DropSpec drop)
{
    const int b = blockIdx.x / n_slabs;
    const int f = (blockIdx.x - b * n_slabs) * 1024 + 4 * threadIdx.x;
    if (f >= F) return;
    const int64_t gf = (int64_t)b * F + f;
    auto ld4 = [](const float *p, float (&v)[4]) {
        const float4 t = *reinterpret_cast<const float4 *>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    };
    float sg[4] = {1.0f, 1.0f, 1.0f, 1.0f}, inv_sg[4] = {1.0f, 1.0f, 1.0f, 1.0f}, ga[4] = {1.0f, 1.0f, 1.0f, 1.0f},
          gb[4] = {1.0f, 1.0f, 1.0f, 1.0f}, dpa[4] = {0.0f, 0.0f, 0.0f, 0.0f}, dpb[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (store_gate) {
        ld4(store_gate + gf, sg);
#pragma unroll
        for (int c = 0; c < 4; ++c) inv_sg[c] = sg[c] != 0.0f ? 1.0f / sg[c] : 0.0f;
    }
    if (gate_a) ld4(gate_a + gf, ga);
    if (gate_b) ld4(gate_b + gf, gb);
    if (d_pa) ld4(d_pa + gf, dpa);
    if (d_pb) ld4(d_pb + gf, dpb);
    const float *o = out + (int64_t)b * T * ldo + f;
    const float *dd = d_out ? d_out + (int64_t)b * T * ldd + f : nullptr;

    float best_a[4], best_b[4], ya[4], yb[4], acc_sg[4];
    int ia[4], ib[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { best_a[c] = -INFINITY; best_b[c] = -INFINITY; ya[c] = 0.0f; yb[c] = 0.0f; acc_sg[c] = 0.0f; ia[c] = 0; ib[c] = 0; }
    const uint32_t e0 = DROP ? (uint32_t)((int64_t)b * T * F + f) : 0u;   // element (node b*T + t, feature f + c) = e0 + t*F + c
#pragma unroll 4
    for (int t = 0; t < T; ++t) {
        float o_t[4], d_t[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        ld4(o + (int64_t)t * ldo, o_t);
        if (dd) ld4(dd + (int64_t)t * ldd, d_t);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float ks = 1.0f, ka = 1.0f, kb = 1.0f;
            if constexpr (DROP) {
                const uint32_t hh = drop_hash(e0 + (uint32_t)t * (uint32_t)F + (uint32_t)c, drop.seed_lo, drop.seed_hi);
                ks = drop_keep(hh, drop.sel[0], drop.thr, drop.scale);
                ka = drop_keep(hh, drop.sel[1], drop.thr, drop.scale);
                kb = drop_keep(hh, drop.sel[2], drop.thr, drop.scale);
            }
            const float y = DROP ? (ks != 0.0f ? o_t[c] * inv_sg[c] / ks : 0.0f) : o_t[c] * inv_sg[c];
            const float va = y * ga[c] * ka, vb = y * gb[c] * kb;
            if (va > best_a[c]) { best_a[c] = va; ia[c] = t; ya[c] = y * ka; }
            if (vb > best_b[c]) { best_b[c] = vb; ib[c] = t; yb[c] = y * kb; }
            if (dd) acc_sg[c] = fmaf(d_t[c], y * ks, acc_sg[c]);
        }
    }
    float bsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 4
    for (int t = 0; t < T; ++t) {
        float d_t[4] = {0.0f, 0.0f, 0.0f, 0.0f}, g[4];
        if (dd) ld4(dd + (int64_t)t * ldd, d_t);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float ks = 1.0f, ka = 1.0f, kb = 1.0f;
            if constexpr (DROP) {
                const uint32_t hh = drop_hash(e0 + (uint32_t)t * (uint32_t)F + (uint32_t)c, drop.seed_lo, drop.seed_hi);
                ks = drop_keep(hh, drop.sel[0], drop.thr, drop.scale);
                ka = drop_keep(hh, drop.sel[1], drop.thr, drop.scale);
                kb = drop_keep(hh, drop.sel[2], drop.thr, drop.scale);
            }
            g[c] = dd ? d_t[c] * sg[c] * ks : 0.0f;
            if (d_pa && t == ia[c]) g[c] = fmaf(dpa[c], ga[c] * ka, g[c]);
            if (d_pb && t == ib[c]) g[c] = fmaf(dpb[c], gb[c] * kb, g[c]);
            bsum[c] += g[c];
        }
        *reinterpret_cast<float4 *>(dY + ((int64_t)b * T + t) * ldy + f) = make_float4(g[0], g[1], g[2], g[3]);   // (non-temporal: this kernel -2 %, the step +1 % -- the next kernel reads dY)
    }
    auto st4 = [](float *p, const float (&v)[4]) { *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]); };
    if (d_bsum) st4(d_bsum + gf, bsum);
    if (d_sg) st4(d_sg + gf, acc_sg);
    if (d_ga) { const float v[4] = {dpa[0] * ya[0], dpa[1] * ya[1], dpa[2] * ya[2], dpa[3] * ya[3]}; st4(d_ga + gf, v); }
    if (d_gb) { const float v[4] = {dpb[0] * yb[0], dpb[1] * yb[1], dpb[2] * yb[2], dpb[3] * yb[3]}; st4(d_gb + gf, v); }
}

// The same pass FOLLOWED by the transposed aggregation dH = A^T . D . dY (gcn.py:41 under train.py:120; what
// ggcn_aggregate_t does in a launch of its own) for graphs of up to 32 nodes with a 0/1 adjacency given as row masks: dY is
// consumed nowhere else, so it never exists in memory -- a thread (four columns) keeps the 32 x 4 sums of its graph's dH rows in
// registers and SCATTERS row t of dY, scaled by 1 / (deg_t + 1), into the rows s with A[t][s] = 1.  The row mask is
// workgroup-uniform: the test of bit s is a scalar compare and branch around four adds, so only real edges cost vector
// instructions (~5 of 32 per row), and the register index s is a compile-time constant.  Additions per dH row run over t
// ascending -- ggcn_aggregate_t's order.  HBM: out + d_out read, dH written: the 2 x 4 N F bytes of dY are gone.
template <bool DROP>
__global__ __launch_bounds__(256) void gate_pool_backward_agg_kernel(
    const float *__restrict__ out, int64_t ldo, const float *__restrict__ store_gate,
    const float *__restrict__ gate_a, const float *__restrict__ gate_b,
    const float *__restrict__ d_out, int64_t ldd, const float *__restrict__ d_pa,
    const float *__restrict__ d_pb, const uint32_t *__restrict__ rowmask, int T, int F, int n_slabs,
    float *__restrict__ dH, int64_t ldh, float *__restrict__ d_sg, float *__restrict__ d_ga, float *__restrict__ d_gb,
    float *__restrict__ d_bsum, DropSpec drop, unsigned int *__restrict__ dh_amax)
{
    const int b = blockIdx.x / n_slabs;
    const int f = (blockIdx.x - b * n_slabs) * 1024 + 4 * threadIdx.x;
    if (f >= F) return;
    const int64_t gf = (int64_t)b * F + f;
    auto ld4 = [](const float *p, float (&v)[4]) {
        const float4 t = *reinterpret_cast<const float4 *>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    };
    float sg[4] = {1.0f, 1.0f, 1.0f, 1.0f}, inv_sg[4] = {1.0f, 1.0f, 1.0f, 1.0f}, ga[4] = {1.0f, 1.0f, 1.0f, 1.0f},
          gb[4] = {1.0f, 1.0f, 1.0f, 1.0f}, dpa[4] = {0.0f, 0.0f, 0.0f, 0.0f}, dpb[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (store_gate) {
        ld4(store_gate + gf, sg);
#pragma unroll
        for (int c = 0; c < 4; ++c) inv_sg[c] = sg[c] != 0.0f ? 1.0f / sg[c] : 0.0f;
    }
    if (gate_a) ld4(gate_a + gf, ga);
    if (gate_b) ld4(gate_b + gf, gb);
    if (d_pa) ld4(d_pa + gf, dpa);
    if (d_pb) ld4(d_pb + gf, dpb);
    const float *o = out + (int64_t)b * T * ldo + f;
    const float *dd = d_out ? d_out + (int64_t)b * T * ldd + f : nullptr;

    float best_a[4], best_b[4], ya[4], yb[4], acc_sg[4];
    int ia[4], ib[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) { best_a[c] = -INFINITY; best_b[c] = -INFINITY; ya[c] = 0.0f; yb[c] = 0.0f; acc_sg[c] = 0.0f; ia[c] = 0; ib[c] = 0; }
    const uint32_t e0 = DROP ? (uint32_t)((int64_t)b * T * F + f) : 0u;
#pragma unroll 4
    for (int t = 0; t < T; ++t) {
        float o_t[4], d_t[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        ld4(o + (int64_t)t * ldo, o_t);
        if (dd) ld4(dd + (int64_t)t * ldd, d_t);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float ks = 1.0f, ka = 1.0f, kb = 1.0f;
            if constexpr (DROP) {
                const uint32_t hh = drop_hash(e0 + (uint32_t)t * (uint32_t)F + (uint32_t)c, drop.seed_lo, drop.seed_hi);
                ks = drop_keep(hh, drop.sel[0], drop.thr, drop.scale);
                ka = drop_keep(hh, drop.sel[1], drop.thr, drop.scale);
                kb = drop_keep(hh, drop.sel[2], drop.thr, drop.scale);
            }
            const float y = DROP ? (ks != 0.0f ? o_t[c] * inv_sg[c] / ks : 0.0f) : o_t[c] * inv_sg[c];
            const float va = y * ga[c] * ka, vb = y * gb[c] * kb;
            if (va > best_a[c]) { best_a[c] = va; ia[c] = t; ya[c] = y * ka; }
            if (vb > best_b[c]) { best_b[c] = vb; ib[c] = t; yb[c] = y * kb; }
            if (dd) acc_sg[c] = fmaf(d_t[c], y * ks, acc_sg[c]);
        }
    }
    float bsum[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    float acc[32][4];
#pragma unroll
    for (int s = 0; s < 32; ++s)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[s][c] = 0.0f;
    const uint32_t *mrow = rowmask + (int64_t)b * T;   // one word per node (T <= 32)
    for (int t = 0; t < T; ++t) {
        float d_t[4] = {0.0f, 0.0f, 0.0f, 0.0f}, gw[4];
        if (dd) ld4(dd + (int64_t)t * ldd, d_t);
        const uint32_t m = __builtin_amdgcn_readfirstlane(mrow[t]);                     // (workgroup-uniform)
        const float w = 1.0f / (float)(__popc(m) + 1);                                    // gcn.py:35
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            float ks = 1.0f, ka = 1.0f, kb = 1.0f;
            if constexpr (DROP) {
                const uint32_t hh = drop_hash(e0 + (uint32_t)t * (uint32_t)F + (uint32_t)c, drop.seed_lo, drop.seed_hi);
                ks = drop_keep(hh, drop.sel[0], drop.thr, drop.scale);
                ka = drop_keep(hh, drop.sel[1], drop.thr, drop.scale);
                kb = drop_keep(hh, drop.sel[2], drop.thr, drop.scale);
            }
            float g = dd ? d_t[c] * sg[c] * ks : 0.0f;
            if (d_pa && t == ia[c]) g = fmaf(dpa[c], ga[c] * ka, g);
            if (d_pb && t == ib[c]) g = fmaf(dpb[c], gb[c] * kb, g);
            bsum[c] += g;
            gw[c] = g * w;
        }
#pragma unroll
        for (int s8 = 0; s8 < 32; s8 += 8) {
            if ((m >> s8) & 0xFFu) {                                                      // (most bytes of a parse's row mask are empty)
#pragma unroll
                for (int s = s8; s < s8 + 8; ++s)
                    if (m & (1u << s)) {                                                  // scalar test: an edge t -> s
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[s][c] += gw[c];
                    }
            }
        }
    }
#pragma unroll
    for (int s = 0; s < 32; ++s)
        if (s < T) *reinterpret_cast<float4 *>(dH + ((int64_t)b * T + s) * ldh + f) = make_float4(acc[s][0], acc[s][1], acc[s][2], acc[s][3]);
    if (dh_amax) {   // the largest |dH| of the launch (ggcn_linear_scaled derives its power-of-two scale from it): one atomic per wavefront
        // (the launcher passes dh_amax only when F % 256 == 0: a wavefront is then wholly inside or wholly past F, and the lane
        // exchange below reads live lanes only)
        float m = 0.0f;
#pragma unroll
        for (int s = 0; s < 32; ++s) {   // v_max3_f32 with |.| modifiers: two values per instruction
            asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m) : "v"(acc[s][0]), "v"(acc[s][1]));
            asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m) : "v"(acc[s][2]), "v"(acc[s][3]));
        }
        if (!(m <= 3.0e38f)) m = __builtin_inff();      // NaN / inf in the gradients: say so (the scaled linear then leaves the data alone)
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
        if ((threadIdx.x & 63) == 0) atomicMax(dh_amax, __float_as_uint(m));   // non-negative floats order like their bit patterns
    }
    auto st4 = [](float *p, const float (&v)[4]) { *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]); };
    if (d_bsum) st4(d_bsum + gf, bsum);
    if (d_sg) st4(d_sg + gf, acc_sg);
    if (d_ga) { const float v[4] = {dpa[0] * ya[0], dpa[1] * ya[1], dpa[2] * ya[2], dpa[3] * ya[3]}; st4(d_ga + gf, v); }
    if (d_gb) { const float v[4] = {dpb[0] * yb[0], dpb[1] * yb[1], dpb[2] * yb[2], dpb[3] * yb[3]}; st4(d_gb + gf, v); }
}

// out[f] = sum_r X[r, f], deterministic: S row slabs each leave a partial row (fixed order inside a slab: the 4
// wavefronts take rows r = w, w+4, ... and are added (w0+w1)+(w2+w3)), then the slabs are added in order.
constexpr int kColsumSlabs = 64;
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float *__restrict__ X, int64_t ld, int64_t M, int F,
                                                             int64_t rows_per_slab, float *__restrict__ part)
{
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int f = blockIdx.x * 64 + lane;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_slab;
    const int64_t r1 = r0 + rows_per_slab < M ? r0 + rows_per_slab : M;
    float s = 0.0f;
    if (f < F)
        for (int64_t r = r0 + w; r < r1; r += 4) s += X[r * ld + f];
    red[w][lane] = s;
    __syncthreads();
    if (w == 0 && f < F) part[(int64_t)blockIdx.y * F + f] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}
__global__ __launch_bounds__(256) void colsum_final_kernel(const float *__restrict__ part, int S, int F, float *__restrict__ out)
{
    const int f = blockIdx.x * 256 + threadIdx.x;
    if (f >= F) return;
    float s = 0.0f;
#pragma unroll 16   // the loads of a group are independent: one latency per 16 slabs, the adds stay in slab order
    for (int z = 0; z < S; ++z) s += part[(int64_t)z * F + f];
    out[f] = s;
}

}  // namespace

size_t colsum_workspace_bytes(int F) { return F > 0 ? (size_t)kColsumSlabs * F * sizeof(float) : 0; }

int colsum(const float *X, int64_t ld, int64_t M, int F, float *out, void *workspace, hipStream_t st)
{
    if (!X || !out || !workspace) return fail(GGCN_EINVAL, "ggcn_colsum: null pointer");
    if (M <= 0 || F <= 0 || ld < F) return fail(GGCN_EINVAL, "ggcn_colsum: M=%lld F=%d ld=%lld", (long long)M, F, (long long)ld);
    const int S = (int)(M < kColsumSlabs ? M : kColsumSlabs);
    const int64_t rows = (M + S - 1) / S;
    float *part = static_cast<float *>(workspace);
    hipLaunchKernelGGL(colsum_partial_kernel, dim3((unsigned)((F + 63) / 64), (unsigned)S), dim3(256), 0, st, X, ld, M, F, rows, part);
    hipLaunchKernelGGL(colsum_final_kernel, dim3((unsigned)((F + 255) / 256)), dim3(256), 0, st, part, S, F, out);
    return check_launch("ggcn_colsum");
}

int gate_pool_backward(const float *out, int64_t ldo, const float *store_gate, const float *gate_a,
                       const float *gate_b, const float *d_out, int64_t ldd, const float *d_pa,
                       const float *d_pb, int B, int T, int F, float *dY, int64_t ldy, float *d_sg,
                       float *d_ga, float *d_gb, float *d_bsum, hipStream_t st, const DropSpec *drop)
{
    if (!out || !dY) return fail(GGCN_EINVAL, "ggcn_gate_pool_backward: null pointer");
    if (B <= 0 || T <= 0 || F <= 0)
        return fail(GGCN_EINVAL, "ggcn_gate_pool_backward: B=%d T=%d F=%d must be positive", B, T, F);
    if (ldo < F || ldy < F || (d_out && ldd < F))
        return fail(GGCN_EINVAL, "ggcn_gate_pool_backward: leading dimension smaller than F=%d", F);
    auto al16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool vec4 = F % 4 == 0 && ldo % 4 == 0 && ldy % 4 == 0 && (!d_out || ldd % 4 == 0) && al16(out) && al16(dY) && al16(d_out) &&
                      al16(store_gate) && al16(gate_a) && al16(gate_b) && al16(d_pa) && al16(d_pb) && al16(d_sg) && al16(d_ga) &&
                      al16(d_gb) && al16(d_bsum);
    const bool dropping = drop && drop->thr != 0;
    if (dropping && (int64_t)B * T * F >= ((int64_t)1 << 32))
        return fail(GGCN_EUNSUPPORTED, "ggcn_gate_pool_backward: gate dropout indexes elements with 32 bits");
    if (vec4) {     // four columns per thread: 16-byte loads and stores
        const int n4 = (F + 1023) / 1024;
        const int64_t blocks4 = (int64_t)B * n4;
        if (blocks4 > (int64_t)INT32_MAX) return fail(GGCN_EUNSUPPORTED, "ggcn_gate_pool_backward: grid too large");
        if (dropping)
            hipLaunchKernelGGL(gate_pool_backward_kernel4<true>, dim3((unsigned)blocks4), dim3(256), 0, st, out, ldo, store_gate,
                               gate_a, gate_b, d_out, ldd, d_pa, d_pb, T, F, n4, dY, ldy, d_sg, d_ga, d_gb, d_bsum, *drop);
        else
            hipLaunchKernelGGL(gate_pool_backward_kernel4<false>, dim3((unsigned)blocks4), dim3(256), 0, st, out, ldo, store_gate,
                               gate_a, gate_b, d_out, ldd, d_pa, d_pb, T, F, n4, dY, ldy, d_sg, d_ga, d_gb, d_bsum, DropSpec{});
        return check_launch("ggcn_gate_pool_backward");
    }
    const int n_slabs = (F + 255) / 256;
    const int64_t blocks = (int64_t)B * n_slabs;
    if (blocks > (int64_t)INT32_MAX) return fail(GGCN_EUNSUPPORTED, "ggcn_gate_pool_backward: grid too large");
    if (dropping) {
        hipLaunchKernelGGL(gate_pool_backward_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, out, ldo, store_gate,
                           gate_a, gate_b, d_out, ldd, d_pa, d_pb, T, F, n_slabs, dY, ldy, d_sg, d_ga, d_gb, d_bsum, *drop);
    } else {
        hipLaunchKernelGGL(gate_pool_backward_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, out, ldo, store_gate,
                           gate_a, gate_b, d_out, ldd, d_pa, d_pb, T, F, n_slabs, dY, ldy, d_sg, d_ga, d_gb, d_bsum, DropSpec{});
    }
    return check_launch("ggcn_gate_pool_backward");
}

int gate_pool_backward_agg(const float *out, int64_t ldo, const float *store_gate, const float *gate_a,
                           const float *gate_b, const float *d_out, int64_t ldd, const float *d_pa,
                           const float *d_pb, const uint32_t *rowmask, int B, int T, int F, float *dH, int64_t ldh,
                           float *d_sg, float *d_ga, float *d_gb, float *d_bsum, hipStream_t st, const DropSpec *drop, float *dh_amax)
{
    const char *who = "ggcn_gate_pool_backward_agg";
    if (!out || !dH || !rowmask) return fail(GGCN_EINVAL, "%s: null pointer", who);
    if (B <= 0 || T <= 0 || F <= 0) return fail(GGCN_EINVAL, "%s: B=%d T=%d F=%d must be positive", who, B, T, F);
    if (T > 32) return fail(GGCN_EUNSUPPORTED, "%s: T=%d > 32; use ggcn_gate_pool_backward + ggcn_aggregate_t", who, T);
    if (ldo < F || ldh < F || (d_out && ldd < F)) return fail(GGCN_EINVAL, "%s: leading dimension smaller than F=%d", who, F);
    auto al16 = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (F % 4 != 0 || ldo % 4 != 0 || ldh % 4 != 0 || (d_out && ldd % 4 != 0) || !al16(out) || !al16(dH) || !al16(d_out) ||
        !al16(store_gate) || !al16(gate_a) || !al16(gate_b) || !al16(d_pa) || !al16(d_pb) || !al16(d_sg) || !al16(d_ga) ||
        !al16(d_gb) || !al16(d_bsum))
        return fail(GGCN_EUNSUPPORTED, "%s: needs F %% 4 == 0, leading dimensions %% 4 == 0 and 16-byte aligned pointers; use "
                                       "ggcn_gate_pool_backward + ggcn_aggregate_t", who);
    const bool dropping = drop && drop->thr != 0;
    if (dropping && (int64_t)B * T * F >= ((int64_t)1 << 32))
        return fail(GGCN_EUNSUPPORTED, "%s: gate dropout indexes elements with 32 bits", who);
    if (dh_amax && F % 256 != 0)
        return fail(GGCN_EUNSUPPORTED, "%s: dh_amax needs F %% 256 == 0 (F=%d): whole wavefronts of columns", who, F);
    const int n4 = (F + 1023) / 1024;
    const int64_t blocks = (int64_t)B * n4;
    if (blocks > (int64_t)INT32_MAX) return fail(GGCN_EUNSUPPORTED, "%s: grid too large", who);
    if (dropping)
        hipLaunchKernelGGL(gate_pool_backward_agg_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, out, ldo, store_gate, gate_a,
                           gate_b, d_out, ldd, d_pa, d_pb, rowmask, T, F, n4, dH, ldh, d_sg, d_ga, d_gb, d_bsum, *drop, reinterpret_cast<unsigned int *>(dh_amax));
    else
        hipLaunchKernelGGL(gate_pool_backward_agg_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, out, ldo, store_gate, gate_a,
                           gate_b, d_out, ldd, d_pa, d_pb, rowmask, T, F, n4, dH, ldh, d_sg, d_ga, d_gb, d_bsum, DropSpec{}, reinterpret_cast<unsigned int *>(dh_amax));
    return check_launch(who);
}

}  // namespace ggcn
