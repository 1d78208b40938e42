// One launch per gated layer for LONG graphs with fp16 features (BASELINE configs[3]: 512-token documents,
// hidden 1024): the reference's two steps (gcn.py:34 hidden = text.W, gcn.py:41 adj.hidden / denom + bias) in one
// kernel, `hidden` never leaving the CU.
//
// A workgroup of 8 wavefronts owns (graph, 128 columns): all 512 row slots of the graph, so every source row a
// neighbour sum can ask for is produced by the SAME workgroup.
//   1. main loop (plain fp16 MFMA as f16_core.h, fp32 accumulation): wavefront (rg, cg) = 128 rows x 64 columns,
//      4 row groups x 2 column groups; the 512 x 64-k stage of X is staged by all 512 threads (8 passes of 16 B,
//      register staging one stage ahead) into a double buffer of 2 x 64 KiB; W fragments come straight from L2
//      (the fp16 part of the f16mx8 image), the four row groups asking for the same fragments back to back.
//   2. the accumulators are rounded to fp16 -- the same rounding `hidden` gets on the two-launch path -- and
//      written over the dead stage buffers: the graph's hidden tile [512][128] fp16 = 128 KiB of LDS.
//   3. neighbour sums out of LDS as in aggregate_narrow (aggregate.hip): the graph's CSR (row pointers, 16-bit
//      local column ids) was staged at kernel start; 8 lanes x 32 B cover a row of the tile, a wavefront works on
//      8 destination rows at once with 8 source rows each in flight; normalise, bias, gate, fp16 row stores, running
//      max / min in registers meeting in LDS at the end (one workgroup sees all rows of its graph: no atomics).
// HBM traffic = X in + out + CSR + W: the algorithmic bytes of SURVEY 8(d) (the 2 x 268 MB round trip of
// `hidden` at config 4 is gone).  LDS: 128 KiB tile + 10 KiB CSR + 16 KiB pool scratch: one workgroup per CU,
// two wavefronts per SIMD -- the occupancy of the two-launch linear.
#include "f16mx8_core.h"
#include "lab_hooks.h"

namespace ggcn {
namespace {

using namespace bx3;
using mx8::f16x8;

constexpr int LR = 512;                     // row slots per workgroup (= kLongMaxT)
constexpr int LC = 128;                     // columns per workgroup
constexpr int LTHR = 512;                   // threads: 8 wavefronts
constexpr int LBK = 64;                     // k per stage
constexpr int LNP = 8;                      // staging passes: 512 rows x 128 B = 4096 pieces of 16 B / 512 threads
constexpr int kPlane = LR * ROWB;           // one k-half of a stage: [512 rows][64 B] = 32 KiB
constexpr int kStage = 2 * kPlane;          // 64 KiB
constexpr int kTile = 2 * kStage;           // double buffer = the hidden tile afterwards: 128 KiB
constexpr int kIdxCapL = 4096;              // staged column ids per graph (more: read from global memory)
constexpr int kOffRp = kTile + LC * 2;                          // (row LR of the tile: all zeros) int[LR + 1]
constexpr int kOffCol = kOffRp + ((LR + 1) * 4 + 15) / 16 * 16; // unsigned short[kIdxCapL]
constexpr int kOffRed = kOffCol + (kIdxCapL + 8) * 2;           // (8 spare ids: the 8-wide reads overrun a row) float[2][16][LC]
constexpr int kLongLds = kOffRed + 2 * 16 * LC * 4;

#define GGCN_SB() __builtin_amdgcn_sched_barrier(0)


// hidden tile: row r at r * 256 B, plain.  A ds_read_b128 is served in four groups of 16 lanes made of {4, 4, 8}
// lanes of two neighbouring 16-lane rows (MI355X_MICROARCH.md, LDS): with one tile row per 16 lanes every group covers
// the 16 chunk positions once, whatever the two rows are -- conflict-free without a swizzle (an XOR keyed on the row
// would break exactly that).
__device__ __forceinline__ int tile_off(int row, int chunk) { return row * (LC * 2) + (chunk << 4); }

// acc[0..7] += w * (the 8 halves of v)   (v_fma_mix_f32: the fp16 operand is read from its half of the dword)
__device__ __forceinline__ void fma_half8l(const uint4 &v, float w, float (&acc)[8])
{
    const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel_hi:[1,0,0]" : "+v"(acc[2 * q]) : "v"(d[q]), "v"(w));
        asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc[2 * q + 1]) : "v"(d[q]), "v"(w));
    }
}

__device__ __forceinline__ void ld8(const float *p, float (&v)[8])
{
    const float4 x = *reinterpret_cast<const float4 *>(p), y = *reinterpret_cast<const float4 *>(p + 4);
    v[0] = x.x; v[1] = x.y; v[2] = x.z; v[3] = x.w; v[4] = y.x; v[5] = y.y; v[6] = y.z; v[7] = y.w;
}

// plain v_max / v_min (fmaxf would first quiet a possible signalling NaN of an inline-asm result: 3 instructions)
__device__ __forceinline__ float vmaxf(float x, float y) { float d; asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y)); return d; }
__device__ __forceinline__ float vminf(float x, float y) { float d; asm("v_min_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y)); return d; }

struct LongArgs {
    const __half *X; int64_t ldx;
    const char *wpack;
    const int32_t *rowptr, *colidx;
    const float *vals;
    const float *bias, *store_gate, *pool_gate_a, *pool_gate_b;
    __half *out; int64_t ldo;
    float *pool_a, *pool_b;
    int B, T, K, F, n_ct, records;
};

template <bool HAS_VALS, bool FULLT>
__global__ __launch_bounds__(LTHR, 2) void layer_fused_long_kernel(const LongArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    int *s_rp = reinterpret_cast<int *>(lds + kOffRp);
    unsigned short *s_col = reinterpret_cast<unsigned short *>(lds + kOffCol);
    float *s_red = reinterpret_cast<float *>(lds + kOffRed);

    // XCD-affine order: the column tiles of a graph run back to back on ONE XCD, whose L2 keeps the graph's X rows
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int b = (slot / a.n_ct) * 8 + xcd;
    const int ct = slot % a.n_ct;
    if (b >= a.B) return;   // whole workgroup, before any barrier
    const int T = a.T, K = a.K, F = a.F;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rg = wave >> 1, cg = wave & 1;
    const int64_t node0 = (int64_t)b * T;
    const int n_tiles_total = (F + NT - 1) / NT;
    const int nt0 = ct * (LC / NT) + cg * RN;
    const int stages = K / LBK;             // the launcher guarantees K % 64 == 0

    GGCN_LT(0);
    // ---- the graph's CSR leaves for LDS (read again only after the main loop's barriers) ----
    const int e_base = a.rowptr[node0];
    const int nnz_g = a.rowptr[node0 + T] - e_base;
    const bool staged = nnz_g <= kIdxCapL;  // workgroup-uniform
    for (int i = tid; i <= T; i += LTHR) s_rp[i] = a.rowptr[node0 + i] - e_base;
    if (staged)
        for (int j = tid; j < nnz_g; j += LTHR) s_col[j] = (unsigned short)(a.colidx[e_base + j] - (int)node0);

    // ---- 1. hidden = X . W ----
    const int piece = tid & 7;              // 16-byte piece of a row's 128 B: k = 8 piece .. 8 piece + 7
    const int s_row = tid >> 3;             // + 64 per pass
    const __half *arow[LNP];
    bool avalid[LNP];
#pragma unroll
    for (int p = 0; p < LNP; ++p) {
        const int r = 64 * p + s_row;
        avalid[p] = FULLT || r < T;
        arow[p] = a.X + (node0 + (avalid[p] ? r : 0)) * a.ldx + 8 * piece;
    }
    uint4 ra[LNP];
    auto load_a_pass = [&](int p, int st) {
        st = st < stages ? st : stages - 1;
        ra[p] = *reinterpret_cast<const uint4 *>(arow[p] + st * LBK);
    };
    auto write_pass = [&](int buf, int p) {
        uint4 v = ra[p];
        if constexpr (!FULLT)
            if (!avalid[p]) v = make_uint4(0u, 0u, 0u, 0u);
        char *plane = lds + buf * kStage + (piece >> 2) * kPlane;
        *reinterpret_cast<uint4 *>(plane + a_lds_off(64 * p + s_row, piece & 3)) = v;
    };
    const char *bbase[RN];
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        const int ntc = nt0 + j < n_tiles_total ? nt0 + j : n_tiles_total - 1;
        bbase[j] = a.wpack + (int64_t)ntc * a.records * mx8::STAGE_PACK_BYTES + lane * 16;
    }
    auto load_b = [&](int r, f16x8 (&bf)[RN][2]) {
        r = r < a.records ? r : a.records - 1;
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            const char *p = bbase[j] + (int64_t)r * mx8::STAGE_PACK_BYTES;
            bf[j][0] = *reinterpret_cast<const f16x8 *>(p);
            bf[j][1] = *reinterpret_cast<const f16x8 *>(p + 1024);
        }
    };
    const int f_row = 128 * rg + (lane & 31), f_half = lane >> 5;
    auto read_a = [&](int buf, int hh, int i, f16x8 (&af)[2]) {
        const char *plane = lds + buf * kStage + hh * kPlane;
        af[0] = *reinterpret_cast<const f16x8 *>(plane + a_lds_off(f_row + 32 * i, f_half));
        af[1] = *reinterpret_cast<const f16x8 *>(plane + a_lds_off(f_row + 32 * i, 2 + f_half));
    };

    f32x16 acc[4][RN];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    f16x8 b0[RN][2], b1[RN][2];
#pragma unroll
    for (int p = 0; p < LNP; ++p) load_a_pass(p, 0);
    load_b(0, b0);
#pragma unroll
    for (int p = 0; p < LNP; ++p) write_pass(0, p);
#pragma unroll
    for (int p = 0; p < LNP; ++p) load_a_pass(p, 1);
    __syncthreads();

    // One stage = two 32-k halves of 4 row blocks x 4 MFMAs; behind the MFMAs of every (half, row block) go the LDS
    // store of one staging pass of the next stage and its reload one stage further on.
    auto stage = [&](int st, auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        f16x8 af[2][2];
        read_a(buf, 0, 0, af[0]);
        load_b(2 * st + 1, b1);
        GGCN_SB();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < 3) read_a(buf, 0, i + 1, af[(i + 1) & 1]);
            else read_a(buf, 1, 0, af[0]);
            GGCN_SB();
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i & 1][0], b0[0][0], acc[i][0], 0, 0, 0);
            GGCN_SB();
            write_pass(buf ^ 1, i);
            GGCN_SB();
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i & 1][0], b0[1][0], acc[i][1], 0, 0, 0);
            GGCN_SB();
            load_a_pass(i, st + 2);
            GGCN_SB();
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i & 1][1], b0[0][1], acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i & 1][1], b0[1][1], acc[i][1], 0, 0, 0);
            GGCN_SB();
        }
        load_b(2 * st + 2, b0);   // b0 is dead: next stage's first record
        GGCN_SB();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i < 3) read_a(buf, 1, i + 1, af[(i + 1) & 1]);
            GGCN_SB();
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i & 1][0], b1[0][0], acc[i][0], 0, 0, 0);
            GGCN_SB();
            write_pass(buf ^ 1, 4 + i);
            GGCN_SB();
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i & 1][0], b1[1][0], acc[i][1], 0, 0, 0);
            GGCN_SB();
            load_a_pass(4 + i, st + 2);
            GGCN_SB();
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i & 1][1], b1[0][1], acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i & 1][1], b1[1][1], acc[i][1], 0, 0, 0);
            GGCN_SB();
        }
        __syncthreads();
    };
    int st = 0;
    for (; st + 1 < stages; st += 2) {
        stage(st, std::integral_constant<int, 0>{});
        stage(st + 1, std::integral_constant<int, 1>{});
    }
    if (st < stages) stage(st, std::integral_constant<int, 0>{});
    // (the last stage's stores went to the buffer nobody reads, and landed before its closing barrier)

    GGCN_LT(1);
    // ---- 2. hidden -> fp16 -> the LDS tile [512][128] over the stage buffers ----
    // Lane pairs (l, l ^ 1) hold neighbouring columns of the same rows: they swap one register of every pair (r, r + 1)
    // so that the even lane stores columns (c, c + 1) of row(r) and the odd lane those of row(r + 1) as ONE dword each:
    // half the LDS store instructions of 2-byte stores (the phase is bound by the store path).
    {
        const int c = lane & 31, h = lane >> 5;
        const bool odd = lane & 1;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < RN; ++j) {
                char *base = lds + (128 * rg + 32 * i + 4 * h + (odd ? 1 : 0)) * (LC * 2) + 2 * (64 * cg + 32 * j + (c & ~1));
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    const float mine0 = acc[i][j][r], mine1 = acc[i][j][r + 1];
                    const float send = odd ? mine0 : mine1;
                    const float recv = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(send), 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
                    const float lo = odd ? recv : mine0, hi = odd ? mine1 : recv;
                    const __half2 v = __floats2half2_rn(lo, hi);
                    *reinterpret_cast<__half2 *>(base + (8 * (r >> 2) + (r & 3)) * (LC * 2)) = v;
                }
            }
    }
    if (tid < 16) *reinterpret_cast<uint4 *>(lds + kTile + 16 * tid) = make_uint4(0u, 0u, 0u, 0u);   // row LR
    __syncthreads();

    GGCN_LT(2);
    // ---- 3. neighbour sums out of the tile, normalise, bias, gates, stores, pools ----
    // 8 lanes x 2 chunks of 16 B cover a row of the tile: a wavefront works on 8 destination rows at once and every
    // per-edge instruction (id, mask, address) serves 16 columns.  Lane (q8, cl) owns chunks cl + 8 par and
    // cl + 8 (1 - par), par = q8 & 1, and reads them in THAT order: the four 16-lane groups of a ds_read_b128 ({0-3,
    // 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS) then cover the 16 chunk positions once each -- conflict-free for any
    // rows (both reads from chunk cl first would be 2-way).
    const int q8 = lane >> 3, cl = lane & 7, par = q8 & 1;
    const int ch[2] = {cl + 8 * par, cl + 8 * (par ^ 1)};
    float vb[2][8], vsg[2][8], vmax[2][8], vmin[2][8];
    bool live[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int col0 = ct * LC + 8 * ch[h];
        live[h] = col0 < F;                 // F % 8 == 0 (launcher)
#pragma unroll
        for (int k = 0; k < 8; ++k) { vb[h][k] = 0.0f; vsg[h][k] = 1.0f; vmax[h][k] = -INFINITY; vmin[h][k] = INFINITY; }
        if (live[h]) {
            if (a.bias) ld8(a.bias + col0, vb[h]);
            if (a.store_gate) ld8(a.store_gate + (int64_t)b * F + col0, vsg[h]);
        }
    }
    const int32_t *cgp = a.colidx + e_base;
    const float *vg = HAS_VALS ? a.vals + e_base : nullptr;
    const int o0 = 16 * ch[0], o1 = 16 * ch[1];
    // Rows differ per 8-lane group: plain divergent control flow.  Up to 8 source rows of a destination row are in
    // flight at once (their 8 column ids are read first, then the 16 tile chunks): two dependent LDS round trips per
    // 8 edges; slots past the row's last edge read the all-zero row LR.  The next row's pointers are fetched a step ahead.
    // (two instantiations of the row loop: ids staged in LDS / ids from global memory -- workgroup-uniform choice)
    auto rows = [&](auto staged_c) {
        constexpr bool STAGED = decltype(staged_c)::value;
        int r = 8 * wave + q8;
        int e_nx = 0, end_nx = 0;
        if (r < T) { e_nx = s_rp[r]; end_nx = s_rp[r + 1]; }
        for (; r < T; r += 64) {
            int e = e_nx;
            const int end = end_nx;
            if (r + 64 < T) { e_nx = s_rp[r + 64]; end_nx = s_rp[r + 65]; }
            float acc[2][8], wsum = 0.0f;
#pragma unroll
            for (int k = 0; k < 8; ++k) { acc[0][k] = 0.0f; acc[1][k] = 0.0f; }
            const int cnt = end - e;
            if constexpr (STAGED) {
                for (; e < end; e += 8) {
                    const int rem = end - e;
                    const unsigned short *ce = s_col + e;
                    int c[8];
                    float w[8];
#pragma unroll
                    for (int j = 0; j < 8; ++j) c[j] = ce[j];      // may overrun the row (and the graph, by < 8 ids): masked below
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        w[j] = 1.0f;
                        if constexpr (HAS_VALS) { w[j] = rem > j ? vg[e + j] : 0.0f; wsum += w[j]; }
                    }
                    uint4 t[8][2];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int base = rem > j ? (c[j] << 8) : (LR << 8);
                        t[j][0] = *reinterpret_cast<const uint4 *>(lds + base + o0);
                        t[j][1] = *reinterpret_cast<const uint4 *>(lds + base + o1);
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        fma_half8l(t[j][0], w[j], acc[0]);
                        fma_half8l(t[j][1], w[j], acc[1]);
                    }
                }
            } else {                                               // > kIdxCapL edges: ids from global memory, one at a time
                for (; e < end; ++e) {
                    const int c0 = cgp[e] - (int)node0;
                    float w0 = 1.0f;
                    if constexpr (HAS_VALS) { w0 = vg[e]; wsum += w0; }
                    fma_half8l(*reinterpret_cast<const uint4 *>(lds + (c0 << 8) + o0), w0, acc[0]);
                    fma_half8l(*reinterpret_cast<const uint4 *>(lds + (c0 << 8) + o1), w0, acc[1]);
                }
            }
            const float inv = 1.0f / ((HAS_VALS ? wsum : (float)cnt) + 1.0f);   // gcn.py:35
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    v[k] = fmaf(acc[h][k], inv, vb[h][k]);                      // gcn.py:41,43
                    vmax[h][k] = vmaxf(vmax[h][k], v[k]);
                    vmin[h][k] = vminf(vmin[h][k], v[k]);
                }
                if (a.out && live[h]) {
                    union { uint4 u; __half2 hh[4]; } o;
#pragma unroll
                    for (int k = 0; k < 4; ++k) o.hh[k] = __floats2half2_rn(v[2 * k] * vsg[h][2 * k], v[2 * k + 1] * vsg[h][2 * k + 1]);
                    *reinterpret_cast<uint4 *>(a.out + (node0 + r) * a.ldo + ct * LC + 8 * ch[h]) = o.u;
                }
            }
        }
    };
    if (staged) rows(std::true_type{});
    else rows(std::false_type{});
    GGCN_LT(3);
    GGCN_LT_WAVE7(7);
    // pools (bert_amir5.py:635-640): max_t (v_t * gate) = gate * (gate >= 0 ? max_t v_t : min_t v_t), exactly
    if (a.pool_a || a.pool_b) {
        // max / min over the 8 row groups of the wavefront (lanes l, l ^ 8, l ^ 16, l ^ 32 hold the same chunk pair only
        // when their parity agrees: xor 16 and 32 keep it; xor 8 flips it and swaps the two chunks)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                vmax[h][k] = fmaxf(vmax[h][k], __shfl_xor(vmax[h][k], 16));
                vmax[h][k] = fmaxf(vmax[h][k], __shfl_xor(vmax[h][k], 32));
                vmin[h][k] = fminf(vmin[h][k], __shfl_xor(vmin[h][k], 16));
                vmin[h][k] = fminf(vmin[h][k], __shfl_xor(vmin[h][k], 32));
            }
        // s_red[0 = max / 1 = min][slot = 2 wave + par][column]: 16 partial rows per kind
        if (q8 < 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    s_red[(0 * 16 + 2 * wave + par) * LC + 8 * ch[h] + k] = vmax[h][k];
                    s_red[(1 * 16 + 2 * wave + par) * LC + 8 * ch[h] + k] = vmin[h][k];
                }
        }
        GGCN_LT(5);
        __syncthreads();
        GGCN_LT(6);
        if (tid < LC && ct * LC + tid < F) {
            float mx = s_red[tid], mn = s_red[16 * LC + tid];
#pragma unroll
            for (int w = 1; w < 16; ++w) {
                mx = fmaxf(mx, s_red[w * LC + tid]);
                mn = fminf(mn, s_red[(16 + w) * LC + tid]);
            }
            const int64_t g = (int64_t)b * F + ct * LC + tid;
            if (a.pool_a) { const float ga = a.pool_gate_a ? a.pool_gate_a[g] : 1.0f; a.pool_a[g] = ga * (ga >= 0.0f ? mx : mn); }
            if (a.pool_b) { const float gb = a.pool_gate_b ? a.pool_gate_b[g] : 1.0f; a.pool_b[g] = gb * (gb >= 0.0f ? mx : mn); }
        }
    }
    GGCN_LT(4);
}
#undef GGCN_SB

}  // namespace

GGCN_LT_READER

int layer_fused_h(const void *X, int64_t ldx, const void *wpack, const int32_t *rowptr, const int32_t *colidx,
                  const float *vals, const float *bias, int B, int T, int K, int F, const float *store_gate,
                  const float *pool_gate_a, const float *pool_gate_b, void *out, int64_t ldo, float *pool_a,
                  float *pool_b, hipStream_t st)
{
    const char *who = "ggcn_layer_fused_h";
    if (!X || !wpack || !rowptr || !colidx) return fail(GGCN_EINVAL, "%s: null input pointer", who);
    if (B <= 0 || T <= 0 || K <= 0 || F <= 0) return fail(GGCN_EINVAL, "%s: B=%d T=%d K=%d F=%d must be positive", who, B, T, K, F);
    if (!out && !pool_a && !pool_b) return fail(GGCN_EINVAL, "%s: no output requested", who);
    if (T > LR) return fail(GGCN_EUNSUPPORTED, "%s: T=%d > %d; use ggcn_linear_h + ggcn_aggregate_h", who, T, LR);
    if (K % LBK != 0 || F % 8 != 0)
        return fail(GGCN_EUNSUPPORTED, "%s: needs K %% 64 == 0 and F %% 8 == 0 (K=%d F=%d); use ggcn_linear_h + ggcn_aggregate_h", who, K, F);
    if (ldx < K || ldx % 8 != 0 || !aligned16(X)) return fail(GGCN_EUNSUPPORTED, "%s: X must be 16-byte aligned with ldx %% 8 == 0, ldx >= K", who);
    if (out && (ldo < F || ldo % 8 != 0 || !aligned16(out)))
        return fail(GGCN_EUNSUPPORTED, "%s: out must be 16-byte aligned with ldo %% 8 == 0, ldo >= F", who);
    if (!aligned16(wpack)) return fail(GGCN_EINVAL, "%s: wpack must be 16-byte aligned", who);
    if ((bias && !aligned16(bias)) || (store_gate && !aligned16(store_gate)) || (pool_gate_a && !aligned16(pool_gate_a)) ||
        (pool_gate_b && !aligned16(pool_gate_b)))
        return fail(GGCN_EUNSUPPORTED, "%s: bias and gates must be 16-byte aligned", who);
    if ((int64_t)B * T >= (int64_t)INT32_MAX) return fail(GGCN_EUNSUPPORTED, "%s: B*T does not fit int32 node ids", who);
    LongArgs a;
    a.X = static_cast<const __half *>(X); a.ldx = ldx; a.wpack = static_cast<const char *>(wpack);
    a.rowptr = rowptr; a.colidx = colidx; a.vals = vals;
    a.bias = bias; a.store_gate = store_gate; a.pool_gate_a = pool_gate_a; a.pool_gate_b = pool_gate_b;
    a.out = static_cast<__half *>(out); a.ldo = ldo; a.pool_a = pool_a; a.pool_b = pool_b;
    a.B = B; a.T = T; a.K = K; a.F = F;
    a.n_ct = (F + LC - 1) / LC;
    a.records = K / BK;
    const int64_t grid = ((int64_t)B + 7) / 8 * 8 * a.n_ct;
    if (grid > (int64_t)INT32_MAX) return fail(GGCN_EUNSUPPORTED, "%s: batch too large", who);
    const bool fullt = T == LR;
#define GGCN_GO(HV, FT)                                                                                                  \
    do {                                                                                                                 \
        auto kern = layer_fused_long_kernel<HV, FT>;                                                                     \
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,        \
                                kLongLds) != hipSuccess)                                                                 \
            return fail(GGCN_ELAUNCH, "%s: cannot reserve %d bytes of LDS", who, kLongLds);                              \
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(LTHR), kLongLds, st, a);                                     \
    } while (0)
    if (vals && fullt) GGCN_GO(true, true);
    else if (vals) GGCN_GO(true, false);
    else if (fullt) GGCN_GO(false, true);
    else GGCN_GO(false, false);
#undef GGCN_GO
    return check_launch(who);
}

}  // namespace ggcn
