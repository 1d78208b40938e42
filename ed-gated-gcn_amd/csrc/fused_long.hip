// One launch per gated layer for LONG graphs with fp16 features (BASELINE configs[3]: 512-token documents,
// hidden 1024): the reference's two steps (gcn.py:34 hidden = text.W, gcn.py:41 adj.hidden / denom + bias) in one
// kernel, `hidden` never leaving the CU.
//
// A workgroup of 8 wavefronts owns (graph, 128 columns): all 512 row slots of the graph, so every source row a
// neighbour sum can ask for is produced by the SAME workgroup.
//   1. main loop (plain fp16 MFMA, the arithmetic of f16_core.h, fp32 accumulation): wavefront (rg, cg) = 128 rows x
//      64 columns, 4 row groups x 2 column groups.  Both operands reach LDS by LDS-DMA (global_load_lds: no staging
//      registers, no ds_write pass): the X stage is [512 rows][128 B] per buffer with the 16-byte chunk XORed by
//      (row >> 1) & 7 -- applied to the SOURCE address, a DMA piece being 8 rows x 128 B = 1 KiB lane-linear in LDS --
//      and the W fragments of the stage (16 x 1 KiB, already in lane order in the packed image: the fp16 part of the
//      f16mx8 records) go through LDS as well, fetched once for the four row groups.  Two buffers of 64 + 16 KiB = all
//      160 KiB; a stage's ten pieces per wavefront leave one stage ahead, behind the first MFMAs; the stage ends in
//      vmcnt(0) + barrier.
//   2. the accumulators are rounded to fp16 -- the same rounding `hidden` gets on the two-launch path -- and
//      written over the dead X buffers: the graph's hidden tile [512][128] fp16 = 128 KiB of LDS.
//   3. neighbour sums out of LDS as in aggregate_narrow (aggregate.hip): the graph's CSR (row pointers, 16-bit
//      local column ids; held in registers during the main loop, stored where the W buffers were); 8 lanes x 32 B cover
//      a row of the tile, a wavefront works on 8 destination rows at once with 8 source rows each in flight;
//      normalise, bias, gate, fp16 row stores, running max / min in registers meeting in LDS at the end (one
//      workgroup sees all rows of its graph: no atomics).
// HBM traffic = X in + out + CSR + W: the algorithmic bytes of SURVEY 8(d) (the 2 x 268 MB round trip of
// `hidden` at config 4 is gone).  One workgroup per CU, two wavefronts per SIMD -- the occupancy of the two-launch
// linear.
#include "f16mx8_core.h"
#include "lab_hooks.h"

namespace ggcn {
namespace {

using namespace bx3;
using mx8::f16x8;

constexpr int LR = 512;                     // row slots per workgroup (GGCN_LONG_MAX_T)
constexpr int LC = 128;                     // columns per workgroup
constexpr int LTHR = 512;                   // threads: 8 wavefronts
constexpr int LBK = 64;                     // k per stage
constexpr int kDmaA = LR * 2 * LBK;         // X stage buffer: [512 rows][128 B] = 64 KiB
constexpr int kDmaB = (LC / NT) * (LBK / KSTEP) * FRAG_BYTES;   // W stage buffer: 4 column tiles x 4 k-steps x 1 KiB = 16 KiB
constexpr int kTile = 2 * kDmaA;            // the two X buffers = the hidden tile afterwards: 128 KiB
constexpr int kDmaLds = kTile + 2 * kDmaB;  // 160 KiB: everything a CU has
constexpr int kIdxCapL = 4096;              // staged column ids per graph (more: read from global memory)
// after the main loop, where the W buffers were:
constexpr int kOffRp = kTile + LC * 2;                          // (row LR of the tile: all zeros) int[LR + 1]
constexpr int kColSpare = 48;                                   // codes behind the last staged one: reads overrun a row (8) / gather one k-step past a row block (32)
constexpr int kOffCol = kOffRp + ((LR + 1) * 4 + 15) / 16 * 16; // unsigned short[kIdxCapL + kColSpare]: ROW CODES (below)
constexpr int kOffRed = kOffCol + (kIdxCapL + kColSpare) * 2;   // float[2][16][LC]
constexpr int kOffInv = kOffRed + 2 * 16 * LC * 4;              // float[LR]: 1 / (degree + 1) per row (the MFMA form of the neighbour sums)
constexpr int kOffLut = (kOffInv + LR * 4 + 127) / 128 * 128;   // uint2[16]: 4 mask bits -> 4 fp16 (1.0 / 0.0), the selection fragments of the MFMA sums (128-aligned: the entry offset is ORed in)
static_assert(kOffLut + 16 * 8 <= kDmaLds, "the epilogue's LDS map must fit where the W buffers were");
// A staged column id is kept as the ROW CODE (id << 2) | class: code << 6 is the byte address of the row's 64-byte unit
// `class` in the hidden tile.  class = id & 3 when the tile is stored unit-swizzled (MFMA neighbour sums: unit u of row r lies
// at unit u ^ (r & 3)), 0 when it is stored plain (lane sums).  Slots past the graph's last edge hold the all-zero row LR.
constexpr int kCodeZero = LR << 2;

#define GGCN_SB() __builtin_amdgcn_sched_barrier(0)
// timing-only switches of the MFMA neighbour sums (lab builds; wrong results): 1 no MFMAs, 2 no transposed reads, 4 no row
// stores, 8 no selection fragments, 16 no normalise / pools / stores at all
#ifndef GGCN_LAB_LONG
#define GGCN_LAB_LONG 0
#endif
// timing-only switches of the main loop's fill (lab builds; wrong results): 1 all ten pieces of the next stage at the top of the
// stage, 2 no W pieces, 4 no X pieces, 8 only the first four X pieces, 16 no MFMAs / fragment reads, 32 X in half-line pieces
#ifndef GGCN_LAB_LONG_MAIN
#define GGCN_LAB_LONG_MAIN 0
#endif
#ifndef GGCN_LAB_LONG_AUX
#define GGCN_LAB_LONG_AUX 2     // cache policy of the row stores (buffer intrinsic aux: 1 sc0, 2 nt, 16 sc1)
#endif
#ifndef GGCN_LAB_LONG_XAUX
#define GGCN_LAB_LONG_XAUX 0    // ... of the X pieces' LDS-DMA loads
#endif
#ifndef GGCN_LAB_LONG_WAUX
#define GGCN_LAB_LONG_WAUX 0    // ... of the W pieces' LDS-DMA loads
#endif

// hidden tile: row r at r * 256 B (LC * 2).  Lane sums (weighted edges, > kIdxCapL edges): plain -- their reads are
// conflict-free by the ORDER in which a lane takes its two chunks (see there); an XOR keyed on the row would break exactly
// that.  MFMA sums: the 64-byte unit (32 columns = one MFMA column block) XORed with row & 3 -- a transposed read takes the
// same unit of FOUR source rows per 32-lane half, and plain rows would all sit on the same 16 banks (4-way).

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
    int lane_sums;   // 1: neighbour sums by lanes for every graph (GGCN_LONG_LANE_SUMS=1: the round-3 form, for A/B timing)
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
    const int stages = K / LBK;             // the launcher guarantees K % 64 == 0

    GGCN_LT(0);
    // ---- the graph's CSR ----
    const int e_base = a.rowptr[node0];
    const int nnz_g = a.rowptr[node0 + T] - e_base;
    const bool staged = nnz_g <= kIdxCapL;  // workgroup-uniform
    // the W buffers occupy the CSR's place during the main loop: the CSR waits in registers (unconditional loads from
    // clamped indices: a branch per load would serialise eight memory latencies)
    int rp_reg = a.rowptr[node0 + (tid < T ? tid : 0)] - e_base;
    const int rp_next = a.rowptr[node0 + (tid < T ? tid + 1 : 1)] - e_base;
    // neighbour sums on the MFMAs (unweighted edges, ids staged in LDS) or by lanes: workgroup-uniform
    const bool mma = !HAS_VALS && staged && !a.lane_sums;
    unsigned short col_reg[kIdxCapL / LTHR];
    {
        const int32_t *cbase = nnz_g > 0 ? a.colidx + e_base : a.rowptr + node0;
        const int last = nnz_g > 0 ? nnz_g - 1 : 0;
#pragma unroll
        for (int j = 0; j < kIdxCapL / LTHR; ++j) {
            const int idx = tid + LTHR * j;
            col_reg[j] = (unsigned short)(cbase[idx < last ? idx : last] - (int)node0);
        }
    }

    // ---- 1. hidden = X . W ----
    f32x16 acc[4][RN];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    {
        const int khalf = lane >> 5;
        // sources of this wavefront's 8 X pieces (rows 64 p + 8 wave .. + 7) and 2 W pieces per stage
        const __half *asrc[8];
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const int row = 64 * p + 8 * wave + (lane >> 3);
            const int lc = (lane & 7) ^ ((row >> 1) & 7);
            const int rowc = FULLT ? row : (row < T ? row : T - 1);   // rows past T: any valid memory (never gathered, never stored)
            asrc[p] = a.X + (node0 + rowc) * a.ldx + 8 * lc;
        }
        const int ntw = ct * (LC / NT) + (wave >> 1);
        const char *bsrc = a.wpack + (int64_t)(ntw < n_tiles_total ? ntw : n_tiles_total - 1) * a.records * mx8::STAGE_PACK_BYTES + lane * 16;
        const int bdst = 2 * kDmaA + ((wave >> 1) * 4 + 2 * (wave & 1)) * 1024;
        auto issue_piece = [&](int t, int st, int buf) {   // t = 0..7: X piece t; 8, 9: the two k-steps of this wavefront's W record
            st = st < stages ? st : stages - 1;
            if (((GGCN_LAB_LONG_MAIN) & 2) && t >= 8) return;
            if (((GGCN_LAB_LONG_MAIN) & 4) && t < 8) return;
            if (((GGCN_LAB_LONG_MAIN) & 8) && t >= 4 && t < 8) return;
            if (((GGCN_LAB_LONG_MAIN) & 32) && t < 8) {   // half-line pieces: 16 rows x 64 B, the k-low halves first (t < 4), the k-high halves later
                const int row = 128 * (t & 3) + 16 * wave + (lane >> 2);
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(a.X + (node0 + row) * a.ldx + 8 * ((lane & 3) + 4 * (t >> 2)) + st * LBK),
                                                 (__attribute__((address_space(3))) void *)(lds + buf * kDmaA + (64 * t + 8 * wave) * 128), 16, 0, GGCN_LAB_LONG_XAUX);
            } else if (t < 8) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(asrc[t] + st * LBK),
                                                 (__attribute__((address_space(3))) void *)(lds + buf * kDmaA + (64 * t + 8 * wave) * 128), 16, 0, GGCN_LAB_LONG_XAUX);
            } else {
                int rec = 2 * st + (wave & 1);
                rec = rec < a.records ? rec : a.records - 1;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(bsrc + (int64_t)rec * mx8::STAGE_PACK_BYTES + (t - 8) * 1024),
                                                 (__attribute__((address_space(3))) void *)(lds + bdst + buf * kDmaB + (t - 8) * 1024), 16, 0, GGCN_LAB_LONG_WAUX);
            }
        };
        const int sw = ((lane & 31) >> 1) & 7;
        const int aoff = (128 * rg + (lane & 31)) * 128;
        auto read_a = [&](int buf, int s, int i, f16x8 &af) {
            if ((GGCN_LAB_LONG_MAIN) & 128) { asm volatile("" : "+v"(af)); return; }
            af = *reinterpret_cast<const f16x8 *>(lds + buf * kDmaA + aoff + i * 4096 + (((2 * s + khalf) ^ sw) << 4));
        };
        auto read_b = [&](int buf, int s, f16x8 (&bf)[RN]) {
            if ((GGCN_LAB_LONG_MAIN) & 256) { asm volatile("" : "+v"(bf[0]), "+v"(bf[1])); return; }
#pragma unroll
            for (int j = 0; j < RN; ++j)
                bf[j] = *reinterpret_cast<const f16x8 *>(lds + 2 * kDmaA + buf * kDmaB + ((2 * cg + j) * 4 + s) * 1024 + 16 * lane);
        };
#pragma unroll
        for (int t = 0; t < 10; ++t) issue_piece(t, 0, 0);
        __syncthreads();   // (vmcnt(0) + barrier: the pieces of every wavefront have landed)
        // One stage = 4 k-steps x 4 row blocks x 2 MFMAs; the ten pieces of the NEXT stage leave one at a time behind the
        // first MFMAs (their buffer was last read a stage ago; its last pieces -- clamped repeats -- land before the closing
        // barrier of the last stage, i.e. before the tile and the CSR overwrite the buffers).
        auto stage = [&](int st, auto bufc) {
            constexpr int buf = decltype(bufc)::value;
            constexpr int AH = 2;     // X fragment reads run two steps (of 2 MFMAs) ahead of their use, W fragments of a k-step three
            f16x8 af[AH + 1], bf[2][RN];
            if ((GGCN_LAB_LONG_MAIN) & 1) {
#pragma unroll
                for (int t = 0; t < 10; ++t) issue_piece(t, st + 1, buf ^ 1);
            }
            if ((GGCN_LAB_LONG_MAIN) & 16) {
#pragma unroll
                for (int t = 0; t < 10; ++t) issue_piece(t, st + 1, buf ^ 1);
                __syncthreads();
                return;
            }
            read_b(buf, 0, bf[0]);
#pragma unroll
            for (int t = 0; t < AH; ++t) read_a(buf, t >> 2, t & 3, af[t]);
            GGCN_SB();
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int t = 4 * s4 + i;
                    if (t + AH < 16) read_a(buf, (t + AH) >> 2, (t + AH) & 3, af[(t + AH) % (AH + 1)]);
                    if (i == 1 && s4 < 3) read_b(buf, s4 + 1, bf[(s4 + 1) & 1]);
                    GGCN_SB();
                    acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t % (AH + 1)], bf[s4 & 1][0], acc[i][0], 0, 0, 0);
                    GGCN_SB();
                    if (t < 10 && !((GGCN_LAB_LONG_MAIN) & 1)) issue_piece(t, st + 1, buf ^ 1);
                    GGCN_SB();
                    acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[t % (AH + 1)], bf[s4 & 1][1], acc[i][1], 0, 0, 0);
                    GGCN_SB();
                }
            if ((GGCN_LAB_LONG_MAIN) & 64) asm volatile("s_waitcnt vmcnt(10)\n\ts_barrier" ::: "memory");   // timing of a ring one stage deeper: waits for the pieces of the stage BEFORE
            else __syncthreads();
        };
        int st = 0;
        for (; st + 1 < stages; st += 2) {
            stage(st, std::integral_constant<int, 0>{});
            stage(st + 1, std::integral_constant<int, 1>{});
        }
        if (st < stages) stage(st, std::integral_constant<int, 0>{});
        // the CSR out of its registers into the W buffers' place (every DMA has landed: the stage's closing barrier)
        if (tid < T) s_rp[tid] = rp_reg;
        if (tid == 0) s_rp[T] = nnz_g;
        reinterpret_cast<float *>(lds + kOffInv)[tid] = tid < T ? 1.0f / ((float)(rp_next - rp_reg) + 1.0f) : 1.0f;   // gcn.py:35
        if (staged) {
            const int cls = mma ? 3 : 0;
#pragma unroll
            for (int j = 0; j < kIdxCapL / LTHR; ++j) {
                const int idx = tid + LTHR * j, id = col_reg[j];
                s_col[idx] = (unsigned short)(idx < nnz_g ? (id << 2) | (id & cls) : kCodeZero);
            }
            if (tid < kColSpare) s_col[kIdxCapL + tid] = (unsigned short)kCodeZero;
            if (tid < 16)
                *reinterpret_cast<uint2 *>(lds + kOffLut + 8 * tid) =
                    make_uint2(((tid & 1) ? 0x3C00u : 0u) | ((tid & 2) ? 0x3C000000u : 0u), ((tid & 4) ? 0x3C00u : 0u) | ((tid & 8) ? 0x3C000000u : 0u));
        }
    }
    GGCN_LT(1);
    // ---- 2. hidden -> fp16 -> the LDS tile [512][128] over the stage buffers ----
    // Lane pairs (l, l ^ 1) hold neighbouring columns of the same rows: they swap one register of every pair (r, r + 1)
    // so that the even lane stores columns (c, c + 1) of row(r) and the odd lane those of row(r + 1) as ONE dword each:
    // half the LDS store instructions of 2-byte stores (the phase is bound by the store path).
    {
        const int c = lane & 31, h = lane >> 5;
        const bool odd = lane & 1;
        // unit swizzle (MFMA sums): the rows of this lane have row & 3 = (odd + (r & 3)) & 3, r even
        const int x0 = mma ? (odd ? 1 : 0) << 6 : 0, x2 = mma ? (odd ? 3 : 2) << 6 : 0;
        const unsigned selp2 = odd ? 0x03020706u : 0x05040100u;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < RN; ++j) {
                const int boff = (128 * rg + 32 * i + 4 * h + (odd ? 1 : 0)) * (LC * 2) + 2 * (64 * cg + 32 * j + (c & ~1));
                char *base0 = lds + (boff ^ x0), *base2 = lds + (boff ^ x2);
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    // round first, (row r, row r + 1) of the own column in one register; then one DPP move and one byte
                    // permute with a per-lane selector: even lanes (mine.lo, partner.lo), odd lanes (partner.hi, mine.hi)
                    const unsigned mu = __builtin_bit_cast(unsigned, __floats2half2_rn(acc[i][j][r], acc[i][j][r + 1]));
                    const unsigned pu = (unsigned)__builtin_amdgcn_mov_dpp((int)mu, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
                    *reinterpret_cast<unsigned *>(((r & 3) ? base2 : base0) + (8 * (r >> 2) + (r & 3)) * (LC * 2)) =
                        __builtin_amdgcn_perm(pu, mu, selp2);
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
        if (live[h] && !mma) {
            if (a.bias) ld8(a.bias + col0, vb[h]);
            if (a.store_gate) ld8(a.store_gate + (int64_t)b * F + col0, vsg[h]);
        }
    }
    // the pools' gates of this thread's column (used after the last barrier: asked for now, so that their latency is not the tail)
    float pga = 1.0f, pgb = 1.0f;
    if (tid < LC && ct * LC + tid < F) {
        const int64_t g = (int64_t)b * F + ct * LC + tid;
        if (a.pool_a && a.pool_gate_a) pga = a.pool_gate_a[g];
        if (a.pool_b && a.pool_gate_b) pgb = a.pool_gate_b[g];
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
                        const int base = (rem > j ? c[j] : kCodeZero) << 6;    // row codes of class 0 here (plain tile)
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
                    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                    __builtin_nontemporal_store(__builtin_bit_cast(u32x4, o.u), reinterpret_cast<u32x4 *>(a.out + (node0 + r) * a.ldo + ct * LC + 8 * ch[h]));
                }
            }
        }
    };
    // ---- 3'. the same sums on the MFMAs (unweighted edges, ids staged): out[32 rows][128 columns] = S . H, where the k
    // axis runs over the EDGE SLOTS of the row block (CSR order: the slots of row m are [rp[m], rp[m + 1])), S[m][k] = 1 if
    // slot k belongs to row m -- built in registers from the two row pointers of the lane, no memory -- and H[k][.] is the
    // source row of slot k, gathered AND transposed by ds_read_b64_tr_b16: lane 4 q + p of a 16-lane group supplies the
    // address of source row q (its own row code), every lane receives 4 k-values of its own column.  16 slots per MFMA
    // step, 4 column blocks per step; slots past the block's last edge belong to later rows (S = 0, finite H) or hold the
    // zero row.  fp32 accumulation of exact products (1.0 x fp16): the sum differs from the lane form only in the order
    // of its fp32 additions.
    auto rows_mma = [&]() {
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        const int m = lane & 31, kg = lane >> 5, qs = (lane & 15) >> 2;
        const int lanepart = ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
        const bool odd = lane & 1;
        const float *s_inv = reinterpret_cast<const float *>(lds + kOffInv);
        const int nrb = (T + 31) >> 5;
        float bj[4], gj[4], pmx[4], pmn[4];
        const int64_t ldo = a.ldo;
        const int cols_here = F - ct * LC < LC ? F - ct * LC : LC;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = ct * LC + 32 * j + m;
            const bool live_c = col < F;
            bj[j] = (a.bias && live_c) ? a.bias[col] : 0.0f;
            gj[j] = (a.store_gate && live_c) ? a.store_gate[(int64_t)b * F + col] : 1.0f;
            pmx[j] = -INFINITY; pmn[j] = INFINITY;
        }
        // row stores: even lanes (row, columns c, c + 1), odd lanes (row + 1, columns c - 1, c) of column block 2 jp + (lane >> 5)
        // (after the blocks' exchange of halves, see finish); dead columns: past the buffer
        int lane_off2[2];
#pragma unroll
        for (int jp = 0; jp < 2; ++jp) {
            const int cb = 32 * (2 * jp + kg) + (m & ~1);
            lane_off2[jp] = ct * LC + cb < F ? (int)(((odd ? 1 : 0) * ldo + cb) * 2) : 0x40000000;
        }
        __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc(
            a.out ? (void *)(a.out + node0 * ldo + ct * LC) : (void *)a.rowptr, 0,
            a.out ? (int)((((int64_t)T - 1) * ldo + cols_here) * 2) : 0, 0x00020000);   // rows >= T and an absent `out`: out of range, dropped
        const unsigned selp = odd ? 0x03020706u : 0x05040100u;   // v_perm_b32(partner, mine): even (mine.lo, partner.lo), odd (partner.hi, mine.hi)
        const unsigned short *codes = s_col + 8 * kg + qs;
        unsigned lutbase = kOffLut;
        asm volatile("" : "+v"(lutbase));    // (kept in a register: an address too large for the ds offset field, ORed into the entry offset)
        for (int rb = wave; rb < nrb; rb += 8) {
            const int R0 = 32 * rb;
            const int r_lo = R0 + m < T ? R0 + m : T, r_hi = R0 + m + 1 < T ? R0 + m + 1 : T;
            const int lo = s_rp[r_lo], hi = s_rp[r_hi];
            const int sb = __builtin_amdgcn_readlane(lo, 0), se = __builtin_amdgcn_readlane(hi, 31);
            const int nks = (se - sb + 15) >> 4;
            f32x16 sum[4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum[j][r] = 0.0f;
            // slots of this lane in k-step ks: sb + 16 ks + 8 kg + 0..7
            const unsigned short *cp = codes + sb;
            typedef __attribute__((address_space(3))) s16x4 *lds_s16x4;
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            typedef __attribute__((address_space(3))) const u32x2 *lds_u2;
            // S[m][8 kg + i], i = 0..7: 1.0 where lo <= slot < hi -- the 8 mask bits from the lane's two row pointers, the
            // fp16 fragments of their two nibbles from a 16-entry table in LDS (128 B = 32 banks: no conflicts)
            int rl = lo - sb - 8 * kg, rh = hi - sb - 8 * kg;     // the row's slot range relative to this lane's first slot of step 0
            unsigned ga[2][4];
            auto addresses = [&](int c0, int c1) {
                ga[0][0] = (unsigned)c0 << 6 | lanepart; ga[1][0] = (unsigned)c1 << 6 | lanepart;   // (the tile starts at LDS byte 0)
#pragma unroll
                for (int j = 1; j < 4; ++j) { ga[0][j] = ga[0][0] ^ (j << 6); ga[1][j] = ga[1][0] ^ (j << 6); }
            };
            auto gather = [&](f16x8 (&bf)[4]) {
                if ((GGCN_LAB_LONG) & 2) return;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    union { f16x8 v; s16x4 h[2]; } u;
                    u.h[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(size_t)ga[0][j]);
                    u.h[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4)(size_t)ga[1][j]);
                    bf[j] = u.v;
                }
            };
            auto select = [&](f16x8 &sel) {
                if ((GGCN_LAB_LONG) & 8) return;
                const int clo = min(max(rl, 0), 8), chi = min(max(rh, clo), 8);
                unsigned mask;
                asm("v_bfm_b32 %0, %1, %2" : "=v"(mask) : "v"(chi - clo), "v"(clo));     // ((1 << width) - 1) << offset
                rl -= 16; rh -= 16;
                union { f16x8 v; u32x2 d[2]; } u;
                u.d[0] = *(lds_u2)(size_t)(((mask << 3) & 0x78u) | lutbase);
                u.d[1] = *(lds_u2)(size_t)(((mask >> 1) & 0x78u) | lutbase);
                sel = u.v;
            };
            // One step (source order = issue order): between the 4 MFMAs of step k leave, unconditionally, the 8 transposed
            // reads of step k + 1 (addresses ready since step k - 1), its selection fragment, the addresses of step k + 2
            // and the codes of step k + 3 -- every LDS result has at least two MFMAs and their companions between issue
            // and use.  (A branch around them would make the compiler wait for them before this step's MFMAs; past the
            // block they touch rows of later slots or the zero row -- finite, unused.)
            f16x8 bf[2][4], sel[2];
            addresses(cp[0], cp[4]);
            gather(bf[0]);
            select(sel[0]);
            addresses(cp[16], cp[20]);
            int n0 = cp[32], n1 = cp[36];
            auto step = [&](int k, auto cur_c) {
                constexpr int cur = decltype(cur_c)::value;
                GGCN_SB();
                if (!((GGCN_LAB_LONG) & 1)) sum[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sel[cur], bf[cur][0], sum[0], 0, 0, 0);
                GGCN_SB();
                gather(bf[cur ^ 1]);
                GGCN_SB();
                if (!((GGCN_LAB_LONG) & 1)) sum[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sel[cur], bf[cur][1], sum[1], 0, 0, 0);
                GGCN_SB();
                select(sel[cur ^ 1]);
                GGCN_SB();
                if (!((GGCN_LAB_LONG) & 1)) sum[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sel[cur], bf[cur][2], sum[2], 0, 0, 0);
                GGCN_SB();
                addresses(n0, n1);
                n0 = cp[16 * (k + 3)]; n1 = cp[16 * (k + 3) + 4];
                GGCN_SB();
                if (!((GGCN_LAB_LONG) & 1)) sum[3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(sel[cur], bf[cur][3], sum[3], 0, 0, 0);
                GGCN_SB();
            };
            int ks = 0;
            for (; ks + 1 < nks; ks += 2) {
                step(ks, std::integral_constant<int, 0>{});
                step(ks + 1, std::integral_constant<int, 1>{});
            }
            if (ks < nks) step(ks, std::integral_constant<int, 0>{});
            // normalise, bias, pools, gate, fp16 rows: lane (m, kg) holds column m of rows R0 + 4 kg + 8 (r >> 2) + (r & 3)
            float inv[16];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float4 q4 = *reinterpret_cast<const float4 *>(s_inv + R0 + 4 * kg + 8 * t);
                inv[4 * t] = q4.x; inv[4 * t + 1] = q4.y; inv[4 * t + 2] = q4.z; inv[4 * t + 3] = q4.w;
            }
            const int s_row = (int)(R0 * ldo * 2);
            // (straight-line instantiations: whole row block or one that straddles T, with or without the row stores)
            auto finish = [&](auto whole_c, auto out_c) {
                constexpr bool WHOLE = decltype(whole_c)::value, OUT = decltype(out_c)::value;
#pragma unroll
                for (int jp = 0; jp < 2; ++jp)
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        unsigned o[2];
#pragma unroll
                        for (int jj = 0; jj < 2; ++jj) {
                            const int j = 2 * jp + jj;
                            const float v0 = fmaf(sum[j][r], inv[r], bj[j]), v1 = fmaf(sum[j][r + 1], inv[r + 1], bj[j]);   // gcn.py:41,43
                            if constexpr (WHOLE) {
                                asm("v_max3_f32 %0, %0, %1, %2" : "+v"(pmx[j]) : "v"(v0), "v"(v1));
                                asm("v_min3_f32 %0, %0, %1, %2" : "+v"(pmn[j]) : "v"(v0), "v"(v1));
                            } else {
                                const int row = R0 + 4 * kg + 8 * (r >> 2) + (r & 3);
                                const float a0 = row < T ? v0 : -INFINITY, a1 = row + 1 < T ? v1 : -INFINITY;
                                const float i0 = row < T ? v0 : INFINITY, i1 = row + 1 < T ? v1 : INFINITY;
                                asm("v_max3_f32 %0, %0, %1, %2" : "+v"(pmx[j]) : "v"(a0), "v"(a1));
                                asm("v_min3_f32 %0, %0, %1, %2" : "+v"(pmn[j]) : "v"(i0), "v"(i1));
                            }
                            if constexpr (OUT && !((GGCN_LAB_LONG) & 4)) {
                                const __half2 mine = __floats2half2_rn(v0 * gj[j], v1 * gj[j]);     // (row r, row r + 1) of column m
                                const unsigned mu = __builtin_bit_cast(unsigned, mine);
                                const unsigned pu = (unsigned)__builtin_amdgcn_mov_dpp((int)mu, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
                                o[jj] = __builtin_amdgcn_perm(pu, mu, selp);
                            }
                        }
                        if constexpr (OUT && !((GGCN_LAB_LONG) & 4)) {
                            // the two column blocks trade halves: afterwards one register holds rows (r, r + 1) of the lower lanes'
                            // row set for BOTH blocks -- 128 contiguous bytes per row and store instruction, whole lines -- and
                            // the other the upper lanes' row set (4 rows further)
                            const auto sw = __builtin_amdgcn_permlane32_swap(o[0], o[1], false, false);
                            const int so = s_row + (int)((8 * (r >> 2) + (r & 3)) * ldo * 2);
                            __builtin_amdgcn_raw_buffer_store_b32(sw[0], orsrc, lane_off2[jp], so, GGCN_LAB_LONG_AUX);
                            __builtin_amdgcn_raw_buffer_store_b32(sw[1], orsrc, lane_off2[jp], so + (int)(4 * ldo * 2), GGCN_LAB_LONG_AUX);
                        }
                    }
            };
            const bool whole = R0 + 32 <= T;        // wavefront-uniform
            if ((GGCN_LAB_LONG) & 16) { pmx[0] = fmaxf(pmx[0], sum[0][0] + sum[1][0] + sum[2][0] + sum[3][0]); continue; }
            if (whole) { if (a.out) finish(std::true_type{}, std::true_type{}); else finish(std::true_type{}, std::false_type{}); }
            else { if (a.out) finish(std::false_type{}, std::true_type{}); else finish(std::false_type{}, std::false_type{}); }
        }
        // the two lane halves hold different rows of the same columns
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            pmx[j] = fmaxf(pmx[j], __shfl_xor(pmx[j], 32));
            pmn[j] = fminf(pmn[j], __shfl_xor(pmn[j], 32));
        }
        if ((a.pool_a || a.pool_b) && lane < 32) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
            {                                       // (slots 0..7: the reduction below reads 8 in this form)
                s_red[(0 * 16 + wave) * LC + 32 * j + m] = pmx[j];
                s_red[(1 * 16 + wave) * LC + 32 * j + m] = pmn[j];
            }
        }
    };
    if (mma) rows_mma();
    else if (staged) rows(std::true_type{});
    else rows(std::false_type{});
    GGCN_LT(3);
    GGCN_LT_WAVE7(7);
    // pools (bert_amir5.py:635-640): max_t (v_t * gate) = gate * (gate >= 0 ? max_t v_t : min_t v_t), exactly
    if (a.pool_a || a.pool_b) {
        // max / min over the 8 row groups of the wavefront (lanes l, l ^ 8, l ^ 16, l ^ 32 hold the same chunk pair only
        // when their parity agrees: xor 16 and 32 keep it; xor 8 flips it and swaps the two chunks)
        if (!mma) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                vmax[h][k] = fmaxf(vmax[h][k], __shfl_xor(vmax[h][k], 16));
                vmax[h][k] = fmaxf(vmax[h][k], __shfl_xor(vmax[h][k], 32));
                vmin[h][k] = fminf(vmin[h][k], __shfl_xor(vmin[h][k], 16));
                vmin[h][k] = fminf(vmin[h][k], __shfl_xor(vmin[h][k], 32));
            }
        }
        // s_red[0 = max / 1 = min][slot = 2 wave + par][column]: 16 partial rows per kind
        if (!mma && q8 < 2) {
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
            for (int w = 1; w < 8; ++w) {
                mx = fmaxf(mx, s_red[w * LC + tid]);
                mn = fminf(mn, s_red[(16 + w) * LC + tid]);
            }
            if (!mma) {
#pragma unroll
                for (int w = 8; w < 16; ++w) {
                    mx = fmaxf(mx, s_red[w * LC + tid]);
                    mn = fminf(mn, s_red[(16 + w) * LC + tid]);
                }
            }
            const int64_t g = (int64_t)b * F + ct * LC + tid;
            if (a.pool_a) a.pool_a[g] = pga * (pga >= 0.0f ? mx : mn);
            if (a.pool_b) a.pool_b[g] = pgb * (pgb >= 0.0f ? mx : mn);
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
    const char *env = getenv("GGCN_LONG_LANE_SUMS");      // read per call: tests compare the two forms in one process
    a.lane_sums = (env && env[0] == '1') || (int64_t)T * ldo * 2 >= (int64_t)1 << 30;   // (the MFMA form addresses `out` with 32-bit offsets per graph)
    const int64_t grid = ((int64_t)B + 7) / 8 * 8 * a.n_ct;
    if (grid > (int64_t)INT32_MAX) return fail(GGCN_EUNSUPPORTED, "%s: batch too large", who);
    const bool fullt = T == LR;
    constexpr int kLds = kDmaLds;
#define GGCN_GO(HV, FT)                                                                                                  \
    do {                                                                                                                 \
        auto kern = layer_fused_long_kernel<HV, FT>;                                                                     \
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,        \
                                kLds) != hipSuccess)                                                                 \
            return fail(GGCN_ELAUNCH, "%s: cannot reserve %d bytes of LDS", who, kLds);                              \
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(LTHR), kLds, st, a);                                     \
    } while (0)
    if (vals && fullt) GGCN_GO(true, true);
    else if (vals) GGCN_GO(true, false);
    else if (fullt) GGCN_GO(false, true);
    else GGCN_GO(false, false);
#undef GGCN_GO
    return check_launch(who);
}

}  // namespace ggcn
