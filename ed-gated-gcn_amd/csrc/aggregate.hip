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

#include <cfloat>

namespace ggcn {
namespace {

template <int VEC>
struct Vec;
template <>
struct Vec<4> {
    using type = float4;
};
template <>
struct Vec<1> {
    using type = float;
};

__device__ __forceinline__ float4 ld(const float4 *p) { return *p; }
__device__ __forceinline__ float ld(const float *p) { return *p; }
__device__ __forceinline__ float4 splat4(float v) { return make_float4(v, v, v, v); }

__device__ __forceinline__ void fma_sel(float4 &acc, bool on, float w, const float4 &h)
{
    if (on) {
        acc.x = fmaf(w, h.x, acc.x);
        acc.y = fmaf(w, h.y, acc.y);
        acc.z = fmaf(w, h.z, acc.z);
        acc.w = fmaf(w, h.w, acc.w);
    }
}
__device__ __forceinline__ void fma_sel(float &acc, bool on, float w, const float &h)
{
    if (on) acc = fmaf(w, h, acc);
}
__device__ __forceinline__ void add_sel(float4 &acc, bool on, const float4 &h)
{
    if (on) {
        acc.x += h.x;
        acc.y += h.y;
        acc.z += h.z;
        acc.w += h.w;
    }
}
__device__ __forceinline__ void add_sel(float &acc, bool on, const float &h)
{
    if (on) acc += h;
}

// y = acc / denom + bias ; IEEE division like torch's `/` (gcn.py:41)
__device__ __forceinline__ float4 finish(const float4 &a, float denom, const float4 &b)
{
    return make_float4(a.x / denom + b.x, a.y / denom + b.y, a.z / denom + b.z, a.w / denom + b.w);
}
__device__ __forceinline__ float finish(const float &a, float denom, const float &b) { return a / denom + b; }

__device__ __forceinline__ float4 mul(const float4 &a, const float4 &b)
{
    return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w);
}
__device__ __forceinline__ float mul(const float &a, const float &b) { return a * b; }
__device__ __forceinline__ float4 vmax(const float4 &a, const float4 &b)
{
    return make_float4(fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w));
}
__device__ __forceinline__ float vmax(const float &a, const float &b) { return fmaxf(a, b); }

__device__ __forceinline__ void st_lds(float *p, const float4 &v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ void st_lds(float *p, const float &v) { *p = v; }

constexpr int kWaves = 4;

// grid.x = B * n_slabs ; block = 256.  slab = 64*VEC columns.
template <int VEC, bool HAS_VALS>
__global__ __launch_bounds__(256) void aggregate_rows(
    const float *__restrict__ Hd, int64_t ldh, const int32_t *__restrict__ rowptr,
    const int32_t *__restrict__ colidx, const float *__restrict__ vals,
    const float *__restrict__ bias, int T, int F, int n_slabs,
    const float *__restrict__ store_gate, const float *__restrict__ pool_gate_a,
    const float *__restrict__ pool_gate_b, float *__restrict__ out, int64_t ldo,
    float *__restrict__ pool_a, float *__restrict__ pool_b)
{
    using V = typename Vec<VEC>::type;
    constexpr int kSlab = kWave * VEC;
    __shared__ float red[2][kWaves][kSlab];

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x / n_slabs;
    const int slab = blockIdx.x - b * n_slabs;
    const int col = slab * kSlab + lane * VEC;
    const bool live = col < F;  // F % VEC == 0 is checked on the host

    const V one = [] { if constexpr (VEC == 4) return splat4(1.0f); else return 1.0f; }();
    const V zero = [] { if constexpr (VEC == 4) return splat4(0.0f); else return 0.0f; }();
    const V ninf = [] { if constexpr (VEC == 4) return splat4(-INFINITY); else return -INFINITY; }();

    V vb = zero, vsg = one, vga = one, vgb = one;
    if (live) {
        const int64_t g = (int64_t)b * F + col;
        if (bias) vb = ld(reinterpret_cast<const V *>(bias + col));
        if (store_gate) vsg = ld(reinterpret_cast<const V *>(store_gate + g));
        if (pool_gate_a) vga = ld(reinterpret_cast<const V *>(pool_gate_a + g));
        if (pool_gate_b) vgb = ld(reinterpret_cast<const V *>(pool_gate_b + g));
    }
    V pa = ninf, pb = ninf;

    const float *hcol = Hd + col;
    for (int t = wave; t < T; t += kWaves) {
        const int64_t row = (int64_t)b * T + t;
        const int beg = rowptr[row];
        const int end = rowptr[row + 1];
        V acc = zero;
        float wsum = 0.0f;
        for (int e = beg; e < end; e += 4) {
            // wave-uniform neighbour ids / weights: scalar loads
            const bool v1 = e + 1 < end, v2 = e + 2 < end, v3 = e + 3 < end;
            const int c0 = colidx[e];
            const int c1 = v1 ? colidx[e + 1] : c0;
            const int c2 = v2 ? colidx[e + 2] : c0;
            const int c3 = v3 ? colidx[e + 3] : c0;
            V h0 = zero, h1 = zero, h2 = zero, h3 = zero;
            if (live) {  // four independent 1 KiB row-segment reads in flight
                h0 = ld(reinterpret_cast<const V *>(hcol + (int64_t)c0 * ldh));
                h1 = ld(reinterpret_cast<const V *>(hcol + (int64_t)c1 * ldh));
                h2 = ld(reinterpret_cast<const V *>(hcol + (int64_t)c2 * ldh));
                h3 = ld(reinterpret_cast<const V *>(hcol + (int64_t)c3 * ldh));
            }
            if constexpr (HAS_VALS) {
                const float w0 = vals[e];
                const float w1 = v1 ? vals[e + 1] : 0.0f;
                const float w2 = v2 ? vals[e + 2] : 0.0f;
                const float w3 = v3 ? vals[e + 3] : 0.0f;
                fma_sel(acc, true, w0, h0);
                fma_sel(acc, v1, w1, h1);
                fma_sel(acc, v2, w2, h2);
                fma_sel(acc, v3, w3, h3);
                wsum += w0;
                wsum += w1;
                wsum += w2;
                wsum += w3;
            } else {
                add_sel(acc, true, h0);
                add_sel(acc, v1, h1);
                add_sel(acc, v2, h2);
                add_sel(acc, v3, h3);
            }
        }
        const float denom = (HAS_VALS ? wsum : (float)(end - beg)) + 1.0f;  // gcn.py:35
        const V y = finish(acc, denom, vb);                                  // gcn.py:41,43
        if (live) {
            if (out) *reinterpret_cast<V *>(out + row * ldo + col) = mul(y, vsg);
            pa = vmax(pa, mul(y, vga));
            pb = vmax(pb, mul(y, vgb));
        }
    }

    if (pool_a || pool_b) {
        st_lds(&red[0][wave][lane * VEC], pa);
        st_lds(&red[1][wave][lane * VEC], pb);
        __syncthreads();
        const int tcol = slab * kSlab + threadIdx.x;
        if ((int)threadIdx.x < kSlab && tcol < F) {
            float ma = red[0][0][threadIdx.x], mb = red[1][0][threadIdx.x];
#pragma unroll
            for (int w = 1; w < kWaves; ++w) {
                ma = fmaxf(ma, red[0][w][threadIdx.x]);
                mb = fmaxf(mb, red[1][w][threadIdx.x]);
            }
            if (pool_a) pool_a[(int64_t)b * F + tcol] = ma;
            if (pool_b) pool_b[(int64_t)b * F + tcol] = mb;
        }
    }
}

template <int VEC>
int launch(const float *Hd, int64_t ldh, const int32_t *rowptr, const int32_t *colidx,
           const float *vals, const float *bias, int B, int T, int F, const float *sg,
           const float *ga, const float *gb, float *out, int64_t ldo, float *pa, float *pb,
           hipStream_t st)
{
    const int slab = kWave * VEC;
    const int n_slabs = (F + slab - 1) / slab;
    const int64_t blocks = (int64_t)B * n_slabs;
    if (blocks > (int64_t)INT32_MAX) return fail(GGCN_EUNSUPPORTED, "ggcn_aggregate: grid too large");
    if (vals)
        hipLaunchKernelGGL((aggregate_rows<VEC, true>), dim3((unsigned)blocks), dim3(256), 0, st, Hd, ldh,
                           rowptr, colidx, vals, bias, T, F, n_slabs, sg, ga, gb, out, ldo, pa, pb);
    else
        hipLaunchKernelGGL((aggregate_rows<VEC, false>), dim3((unsigned)blocks), dim3(256), 0, st, Hd, ldh,
                           rowptr, colidx, vals, bias, T, F, n_slabs, sg, ga, gb, out, ldo, pa, pb);
    return check_launch("ggcn_aggregate");
}

}  // namespace

int aggregate(const float *Hd, int64_t ldh, const int32_t *rowptr, const int32_t *colidx,
              const float *vals, const float *bias, int B, int T, int F, const float *store_gate,
              const float *pool_gate_a, const float *pool_gate_b, float *out, int64_t ldo,
              float *pool_a, float *pool_b, hipStream_t st)
{
    if (!Hd || !rowptr || !colidx) return fail(GGCN_EINVAL, "ggcn_aggregate: null input pointer");
    if (B <= 0 || T <= 0 || F <= 0)
        return fail(GGCN_EINVAL, "ggcn_aggregate: B=%d T=%d F=%d must be positive", B, T, F);
    if (!out && !pool_a && !pool_b) return fail(GGCN_EINVAL, "ggcn_aggregate: no output requested");
    if (ldh < F || (out && ldo < F))
        return fail(GGCN_EINVAL, "ggcn_aggregate: leading dimension smaller than F=%d", F);
    if ((int64_t)B * T >= (int64_t)INT32_MAX)
        return fail(GGCN_EUNSUPPORTED, "ggcn_aggregate: B*T does not fit int32 node ids");
    const bool vec = (F % 4 == 0) && (ldh % 4 == 0) && aligned16(Hd) &&
                     (!out || ((ldo % 4 == 0) && aligned16(out))) && (!bias || aligned16(bias)) &&
                     (!store_gate || aligned16(store_gate)) && (!pool_gate_a || aligned16(pool_gate_a)) &&
                     (!pool_gate_b || aligned16(pool_gate_b));
    if (vec)
        return launch<4>(Hd, ldh, rowptr, colidx, vals, bias, B, T, F, store_gate, pool_gate_a,
                         pool_gate_b, out, ldo, pool_a, pool_b, st);
    return launch<1>(Hd, ldh, rowptr, colidx, vals, bias, B, T, F, store_gate, pool_gate_a,
                     pool_gate_b, out, ldo, pool_a, pool_b, st);
}

}  // namespace ggcn
