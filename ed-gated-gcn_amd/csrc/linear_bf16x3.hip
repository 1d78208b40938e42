// Dense linear at ~fp32 accuracy on the bf16 matrix cores (GGCN_PREC_BF16X3):
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
// Kernel shape (one workgroup per CU, 512 threads = 8 wavefronts = 2 per SIMD):
//   * tile 256 (M) x 256 (F), K advanced 32 at a time; wavefronts 2 (M) x 4 (F),
//     each owning 128 x 64 = 4 x 2 MFMA tiles of 32x32 (128 accumulator VGPRs);
//   * X is read as fp32 (16 B per lane, one 128-B line per 8 lanes), split in
//     registers (v_cvt_pk_bf16_f32) and written as two bf16 planes to LDS,
//     double-buffered: ONE barrier per 32-deep step.  Rows are 64 B; the 16-B chunk
//     index is XORed with (row>>2)&3 so each ds_read_b128 lane group touches 16
//     distinct 16-B slots (bank-conflict free; cdna guide §5.5 T2 for 64-B rows);
//   * W never touches LDS: ggcn_weight_pack stores it ONCE in MFMA B-fragment
//     order -- [n_tile][k_step][hi|lo][lane][8 x bf16] -- so a wavefront fetches a
//     fragment with a single coalesced 1 KiB global_load_dwordx4 from L2 (the
//     2.4 MB image is L2-resident), one k-step ahead of its use;
//   * the workgroup id is remapped so the F-tiles of one row block run on the
//     same XCD back to back and share that row block through the XCD's L2.
//
// Operand lane maps of v_mfma_f32_32x32x16_bf16 (cdna guide §3): lane l, r = l&31,
// h = l>>5: A[row r][k = 8h+j], B[k = 8h+j][col r], j = 0..7;
// C/D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5).
#include "common.h"

namespace ggcn {
namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 256, BN = 256, BK = 32;
constexpr int KSTEP = 16;                 // K per MFMA
constexpr int NT = 32;                    // columns per MFMA tile
constexpr int FRAG_BYTES = 64 * 16;       // one B fragment: 64 lanes x 8 bf16
constexpr int kThreads = 512;

__host__ __device__ inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// ---- W -> fragment-ordered bf16 hi/lo image --------------------------------------------
// block (n_tile, k_step), 128 threads: thread = (plane, lane)
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
        const float w = (k < K && n < F) ? W[(int64_t)k * ldw + n] : 0.0f;
        const __bf16 hi = (__bf16)w;
        v[j] = plane == 0 ? hi : (__bf16)(w - (float)hi);
    }
    pack[(((int64_t)n_tile * k_steps + k_step) * 2 + plane) * 64 + lane] = v;
}

// LDS image of one A plane: [256 rows][4 chunks of 16 B], chunk XOR-swizzled by (row>>2)&3
__device__ __forceinline__ int a_lds_off(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

template <bool AVEC>
__global__ __launch_bounds__(kThreads, 2) void linear_bf16x3_kernel(
    const float *__restrict__ X, int64_t ldx, const char *__restrict__ wpack,
    float *__restrict__ Y, int64_t ldy, int64_t M, int K, int F, int m_tiles, int n_wg, int k_steps)
{
    // [buffer][plane][256 x 64 B]
    __shared__ __attribute__((aligned(16))) char lds[2][2][BM * 64];

    // XCD-aware remap: ids congruent mod 8 share an XCD (observed round-robin dispatch);
    // inside one XCD's sequence consecutive ids walk the F-tiles of the same row block.
    const int id = blockIdx.x;
    const int xcd = id & 7, slot = id >> 3;
    const int m_tile = (slot / n_wg) * 8 + xcd;
    const int n_wgi = slot % n_wg;
    if (m_tile >= m_tiles) return;  // whole workgroup leaves before any barrier

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int64_t m0 = (int64_t)m_tile * BM;
    const int n_tiles_total = (F + NT - 1) / NT;
    const int nt0 = n_wgi * (BN / NT) + wn * 2;  // this wavefront's first 32-column tile

    // ---- A staging roles: 4 x (row = i*64 + tid/8, k = (tid%8)*4 .. +3) ----
    const int s_row = tid >> 3;
    const int s_k4 = (tid & 7) * 4;
    float4 ra[4];
    auto load_a = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t gm = m0 + i * 64 + s_row;
            const int gk = k0 + s_k4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (gm < M) {
                const float *p = X + gm * ldx + gk;
                if (AVEC) {
                    if (gk < K) v = *reinterpret_cast<const float4 *>(p);
                } else {
                    if (gk + 0 < K) v.x = p[0];
                    if (gk + 1 < K) v.y = p[1];
                    if (gk + 2 < K) v.z = p[2];
                    if (gk + 3 < K) v.w = p[3];
                }
            }
            ra[i] = v;
        }
    };
    auto store_a = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = i * 64 + s_row;
            const int off = a_lds_off(row, s_k4 >> 3) + (s_k4 & 4) * 2;
            bf16x4 hi, lo;
            hi[0] = (__bf16)ra[i].x; hi[1] = (__bf16)ra[i].y; hi[2] = (__bf16)ra[i].z; hi[3] = (__bf16)ra[i].w;
            lo[0] = (__bf16)(ra[i].x - (float)hi[0]);
            lo[1] = (__bf16)(ra[i].y - (float)hi[1]);
            lo[2] = (__bf16)(ra[i].z - (float)hi[2]);
            lo[3] = (__bf16)(ra[i].w - (float)hi[3]);
            *reinterpret_cast<bf16x4 *>(&lds[buf][0][off]) = hi;
            *reinterpret_cast<bf16x4 *>(&lds[buf][1][off]) = lo;
        }
    };

    // ---- B fragments straight from the packed image (L2) ----
    const bool nt_live0 = nt0 < n_tiles_total, nt_live1 = nt0 + 1 < n_tiles_total;
    const char *bbase0 = wpack + ((int64_t)nt0 * k_steps) * 2 * FRAG_BYTES + lane * 16;
    const char *bbase1 = bbase0 + (int64_t)k_steps * 2 * FRAG_BYTES;
    auto load_b = [&](int ks, bf16x8 (&b)[2][2]) {  // [col tile][plane]
        const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        const int64_t o = (int64_t)ks * 2 * FRAG_BYTES;
        const bool k_live = ks < k_steps;
        b[0][0] = (nt_live0 && k_live) ? *reinterpret_cast<const bf16x8 *>(bbase0 + o) : z;
        b[0][1] = (nt_live0 && k_live) ? *reinterpret_cast<const bf16x8 *>(bbase0 + o + FRAG_BYTES) : z;
        b[1][0] = (nt_live1 && k_live) ? *reinterpret_cast<const bf16x8 *>(bbase1 + o) : z;
        b[1][1] = (nt_live1 && k_live) ? *reinterpret_cast<const bf16x8 *>(bbase1 + o + FRAG_BYTES) : z;
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int f_row = wm * 128 + (lane & 31);
    const int f_half = lane >> 5;

    auto mma_step = [&](int buf, int s, const bf16x8 (&b)[2][2]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int off = a_lds_off(f_row + i * 32, s * 2 + f_half);
            const bf16x8 a_hi = *reinterpret_cast<const bf16x8 *>(&lds[buf][0][off]);
            const bf16x8 a_lo = *reinterpret_cast<const bf16x8 *>(&lds[buf][1][off]);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, b[j][0], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b[j][1], acc[i][j], 0, 0, 0);
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b[j][0], acc[i][j], 0, 0, 0);
            }
        }
    };

    bf16x8 b0[2][2], b1[2][2];
    const int stages = (K + BK - 1) / BK;
    load_a(0);
    load_b(0, b0);
    for (int st = 0; st < stages; ++st) {
        const int buf = st & 1;
        store_a(buf);
        __syncthreads();  // the only barrier of the stage (double-buffered LDS)
        if (st + 1 < stages) load_a((st + 1) * BK);
        load_b(st * 2 + 1, b1);
        mma_step(buf, 0, b0);
        load_b(st * 2 + 2, b0);
        mma_step(buf, 1, b1);
    }

    // ---- epilogue: plain store (fused variants live in fused_layer.hip) ----
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int gn = (nt0 + j) * NT + (lane & 31);
        if (gn >= F) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t gm = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                if (gm < M) Y[gm * ldy + gn] = acc[i][j][r];
            }
    }
}

}  // namespace

size_t weight_pack_bytes(int K, int F)
{
    if (K <= 0 || F <= 0) return 0;
    const size_t k_steps = (size_t)round_up(K, BK) / KSTEP;  // padded to whole 32-deep stages
    const size_t n_tiles = (size_t)round_up(F, NT) / NT;
    return n_tiles * k_steps * 2 * FRAG_BYTES;
}

int weight_pack(const float *W, int64_t ldw, int K, int F, void *wpack, hipStream_t st)
{
    if (!W || !wpack) return fail(GGCN_EINVAL, "ggcn_weight_pack: null pointer");
    if (K <= 0 || F <= 0 || ldw < F) return fail(GGCN_EINVAL, "ggcn_weight_pack: bad shape K=%d F=%d ldw=%lld", K, F, (long long)ldw);
    if (!aligned16(wpack)) return fail(GGCN_EINVAL, "ggcn_weight_pack: wpack must be 16-byte aligned");
    const int k_steps = round_up(K, BK) / KSTEP;
    const int n_tiles = round_up(F, NT) / NT;
    if (k_steps > 65535) return fail(GGCN_EUNSUPPORTED, "ggcn_weight_pack: K too large");
    hipLaunchKernelGGL(weight_pack_kernel, dim3((unsigned)n_tiles, (unsigned)k_steps), dim3(128), 0, st, W,
                       ldw, K, F, k_steps, static_cast<bf16x8 *>(wpack));
    return check_launch("ggcn_weight_pack");
}

int linear_bf16x3(const float *X, int64_t ldx, const void *wpack, float *Y, int64_t ldy, int64_t M,
                  int K, int F, hipStream_t st)
{
    if (!wpack) return fail(GGCN_EINVAL, "ggcn_linear(bf16x3): wpack is NULL (call ggcn_weight_pack first)");
    if (!aligned16(wpack)) return fail(GGCN_EINVAL, "ggcn_linear(bf16x3): wpack must be 16-byte aligned");
    const bool avec = (K % 4 == 0) && (ldx % 4 == 0) && aligned16(X);
    const int k_steps = round_up(K, BK) / KSTEP;
    const int64_t m_tiles = (M + BM - 1) / BM;
    const int n_wg = (F + BN - 1) / BN;
    const int64_t grid = (m_tiles + 7) / 8 * 8 * n_wg;
    if (grid > (int64_t)INT32_MAX || m_tiles > (int64_t)INT32_MAX)
        return fail(GGCN_EUNSUPPORTED, "ggcn_linear(bf16x3): M too large");
    const char *wp = static_cast<const char *>(wpack);
    if (avec)
        hipLaunchKernelGGL((linear_bf16x3_kernel<true>), dim3((unsigned)grid), dim3(kThreads), 0, st, X, ldx, wp,
                           Y, ldy, M, K, F, (int)m_tiles, n_wg, k_steps);
    else
        hipLaunchKernelGGL((linear_bf16x3_kernel<false>), dim3((unsigned)grid), dim3(kThreads), 0, st, X, ldx, wp,
                           Y, ldy, M, K, F, (int)m_tiles, n_wg, k_steps);
    return check_launch("ggcn_linear(bf16x3)");
}

}  // namespace ggcn
