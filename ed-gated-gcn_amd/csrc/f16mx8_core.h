// Main loop of the "f16mx8" linear: fp32-accurate X.W with TWO MFMA units per product instead
// of the three of bf16x3 (bf16x3_core.h), same tiling, same LDS footprint.
//
//   x = xh + xl,  xh = fp16(x) (11 bits, RNE),  xl = x - xh  (exact in fp32, |xl| <= 2^-11 |x|)
//   w = wh + wl   likewise (made once by ggcn_weight_pack)
//   x.w = xh.wh + (xl.wh + xh.wl) + xl.wl
//         `--- v_mfma_f32_32x32x16_f16, exact products, fp32 accumulate
//                  `--- ONE block-scaled fp8 MFMA (v_mfma_scale_f32_32x32x64_f8f6f4, e4m3):
//                       its 64-deep K is [block 0 | block 1] = [xl.wh | xh.wl] over the SAME 32 k,
//                       each block with its own power-of-two scale (MX); it runs at twice the bf16
//                       rate, so it costs what ONE bf16 MFMA pair costs
//                               `--- dropped: <= 2^-22 |x.w|
// The correction is ~2^-12 of the product, so fp8's 2^-4 relative precision leaves ~2^-16 -- the
// same order as bf16x3 -- for 2/3 of the matrix-pipe time (128 instead of 192 cycles per 32 k of a
// 32x32 tile).
//
// MX operand facts measured with tools/probes/mx_fp8_probe.py (exact-integer data, gfx950):
//   * A (and B) operand: lane l = (r = l&31, h = l>>5) holds 32 bytes; byte j is the hardware's
//     k = 32*(j>>4) + 16*h + (j&15): bytes 0-15 belong to scale block 0, bytes 16-31 to block 1
//     (only the block matters here: inside a block both operands use the order described below);
//   * the E8M0 scale of (row r, block b) is byte 0 (opsel 0) of the scale VGPR of lane r + 32*b;
//     value = stored * 2^(scale - 127).
// A-side scales are FIXED: xl is stored as xl*2^11 (scale byte 116), xh as it is (127).  fp8 e4m3
// then covers |x| in [2^-9, 448]; below, the (already ~2^-12-small) correction flushes gradually;
// above, it saturates and the result degrades gracefully to the fp16 product's 2^-12.  fp16 itself
// needs |x| < 65504: this mode is for activations of ordinary magnitude (LSTM / GCN outputs);
// bf16x3 keeps the full fp32 range.
//
// LDS per stage and buffer: plane 0 = xh as fp16 [128 rows][64 B], plane 1 = per row [xl8 of lane half 0 |
// xl8 of lane half 1 | xh8 of half 0 | xh8 of half 1], 16 B each, in the k order of the fp16 fragments --
// 128 B per row like bf16x3, same XOR chunk swizzle, same conflict-free reads; the MX operand is two plain
// 16-byte reads.
// Packed weight per (32-column tile, 32-deep stage): [f16 frag k-step 0: 1 KiB][k-step 1: 1 KiB]
// [fp8(wl * 2^s1): 64 lanes x 16 B][scales: 64 lanes x 4 B] = 3328 B.  The other half of the MX
// operand, fp8(wh * 2^s0), is NOT stored: every lane converts the 16 fp16 values of its two fragments
// (8 x v_cvt_scalef32_pk_fp8_f16 per column tile and stage).  That fixes the k order inside a scale
// block -- lane half h covers k = 8h..8h+7 and 16+8h..16+8h+7, the k of its fp16 fragments -- and the
// A side reads its xl8/xh8 groups in the same order (any order works as long as both operands agree).
// 23 % fewer W bytes through L2 for 16 VALU per stage: fused layer -2 %, plain linear -3 %.
#pragma once
#include "bf16x3_core.h"
#include "lab_hooks.h"

namespace ggcn {
namespace mx8 {

using namespace bx3;  // geometry, LDS addressing, tile mapping are shared

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

// GGCN_WH8_STORED (default 0): 1 = the image also holds fp8(wh * 2^s0), the other half of the MX operand (1 KiB more per
// record: W traffic through L2 +31 %) instead of every lane converting it from its fp16 fragments in the loop (16
// v_cvt_scalef32_pk_fp8_f16 + 4 VALU per stage).  Bit-identical results; measured twice: round 2 had the in-loop form 2 %
// faster, round 4 (the loop's other VALU work cut by a quarter, the ladder pricing the converts at 29 us) block 624 vs
// 622 us, one layer 355 vs 344, plain linear 335 vs 324 -- W's path from L2 is the scarcer resource.
#ifndef GGCN_WH8_STORED
#define GGCN_WH8_STORED 0
#endif
constexpr int kWh8Bytes = GGCN_WH8_STORED ? 64 * 16 : 0;
constexpr int kOffWl8 = 2 * 1024, kOffWh8 = kOffWl8 + 64 * 16, kOffScales = kOffWh8 + kWh8Bytes;
constexpr int STAGE_PACK_BYTES = kOffScales + 64 * 4;  // 3328 (4352 with the stored wh8)
constexpr int XL_SHIFT = 11;                                    // xl is stored as xl * 2^11
constexpr int SCALE_XL = 127 - XL_SHIFT, SCALE_XH = 127;

// ---- the split of 4 consecutive fp32 values: 10 VALU instructions -----------------------------
//   2 x v_cvt_pk_f16_f32 (RNE)            xh as two fp16 pairs
//   4 x v_fma_mix_f32  x - float(xh)      the residual, exact (fp16 factor read straight from the pair)
//   2 x v_cvt_scalef32_pk_fp8_f32         fp8(xl / 2^-11): the scale operand DIVIDES (cvt_probe.py)
//   2 x v_cvt_pk_fp8_f32 of x itself      fp8(x) stands in for fp8(xh): it only feeds the correction
// Sticky range flag: f16mx8 needs |x| < 65504 on its fp32 inputs (larger values saturate in the fp16 plane: MODE.FP16_OVFL).
// Every main loop keeps the running maximum of |x| it splits (one v_max3 per two values) and ORs 1 into this word when it
// reaches fp16's largest finite value -- one atomic, only when hit.  One copy per translation unit that includes this file
// (no relocatable device code); ggcn_range_flag (capi.hip) reads / clears all of them.
// Bits of the flag (ggcn.h GGCN_RANGE_*): 1 = a value reached 65504 (fp16 saturates: results wrong); 2 = a value left the
// window |x| <= 448 in which the fp8 correction is exact to its format (beyond it the correction saturates and the product
// falls to plain fp16 accuracy, 2^-12 relative: outside the 1e-4 parity gate); 4 = the one-launch layer of graphs of
// <= 32 nodes could not rule out an fp16 overflow of its `hidden` planes (max|x| * max_f sum_k |w[k,f]| (+ max|mid|) >= 65504).
static __device__ unsigned int g_range_flag;
constexpr float kWindow = 448.0f;       // largest e4m3 value: fp8(x) and fp8(xl * 2^11) are unsaturated up to here
constexpr float kHalfMax = 65504.0f;
// verdict of one lane: amax = the largest |x| it split; hb_scale / hb_add: bound of the hidden values (0: not applicable)
__device__ __forceinline__ void range_verdict(float amax, float hb_scale = 0.0f, float hb_add = 0.0f, bool window = true)
{
    unsigned bits = 0u;
    if (window && amax > kWindow) bits |= 2u;   // (f16mx6's true block scales have no such window)
    if (amax >= kHalfMax) bits |= 1u;   // (inf included; a NaN input shows in the output instead)
    if (amax * hb_scale + hb_add >= kHalfMax) bits |= 4u;
    if (bits) atomicOr(&g_range_flag, bits);
}
// this translation unit's copy of the flag: OR it into *dst (device memory), clear on request
#define GGCN_RANGE_FLAG_TU(fn)                                                                          \
    namespace {                                                                                         \
    __global__ void fn##_kernel(unsigned int *dst, int clear)                                           \
    {                                                                                                   \
        const unsigned int v = mx8::g_range_flag;                                                       \
        if (v) atomicOr(dst, v);                                                                        \
        if (clear) mx8::g_range_flag = 0u;                                                              \
    }                                                                                                   \
    }                                                                                                   \
    int fn(unsigned int *dst, int clear, hipStream_t st)                                                \
    {                                                                                                   \
        hipLaunchKernelGGL(fn##_kernel, dim3(1), dim3(1), 0, st, dst, clear);                           \
        return check_launch("ggcn_range_flag");                                                         \
    }
// The packed weight images end in a 16-byte trailer: float[0] = max_f sum_k |w[k,f]| (ggcn_weight_pack), the factor of the
// hidden-value bound above.
constexpr int PACK_TRAILER_BYTES = 16;
__device__ __forceinline__ float amax3(float x0, float x1, float m)
{
    float d;
    asm("v_max3_f32 %0, |%1|, |%2|, %3" : "=v"(d) : "v"(x0), "v"(x1), "v"(m));
    return d;
}

struct Split4 {
    uint32_t h01, h23;  // fp16 pairs
    int l8, h8;         // 4 x fp8 of xl * 2^11, 4 x fp8 of x
};
// A register with unspecified contents, for free: the packed fp8 converts write HALF of their destination and keep the
// other half (a tied "old" operand), and both halves get written here, so what the destination starts with is irrelevant --
// but a seed that is still live elsewhere costs a v_mov, and so does a zero.  An empty asm "defines" a fresh register
// without an instruction; the register allocator then places it wherever the result has to live (e.g. inside the
// 8-register MX operand).  amax: the running maximum of |x| (range flag).
__device__ __forceinline__ int undef_vgpr() { int r; asm volatile("" : "=v"(r)); return r; }
__device__ __forceinline__ Split4 split4(float x0, float x1, float x2, float x3, float &amax)
{
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef short s2 __attribute__((ext_vector_type(2)));
    const h2 p0 = __builtin_convertvector(f2{x0, x1}, h2);
    const h2 p1 = __builtin_convertvector(f2{x2, x3}, h2);
    float r0, r1, r2, r3;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(p0), "v"(x0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(p0), "v"(x1));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r2) : "v"(p1), "v"(x2));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r3) : "v"(p1), "v"(x3));
    constexpr float inv = 1.0f / (float)(1 << XL_SHIFT);
    s2 q = __builtin_bit_cast(s2, undef_vgpr());
    q = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(q, r0, r1, inv, false);
    q = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(q, r2, r3, inv, true);
    amax = amax3(x2, x3, amax3(x0, x1, amax));
    int h8 = __builtin_amdgcn_cvt_pk_fp8_f32(x0, x1, undef_vgpr(), false);
    h8 = __builtin_amdgcn_cvt_pk_fp8_f32(x2, x3, h8, true);
    Split4 o;
    o.h01 = __builtin_bit_cast(uint32_t, p0);
    o.h23 = __builtin_bit_cast(uint32_t, p1);
    o.l8 = __builtin_bit_cast(int, q);
    o.h8 = h8;
    return o;
}

// MODE.FP16_OVFL (hwreg MODE bit 23): overflowing fp16 and fp8 conversions clamp to the largest finite
// value instead of producing inf / NaN (checked by tools/probes/cvt_probe.py), so |x| > 448 only
// costs the correction its accuracy and |x| > 65504 saturates like any fp16 pipeline.
__device__ __forceinline__ void set_cvt_saturate(bool on) { __builtin_amdgcn_s_setreg(1 | (23 << 6), on ? 1 : 0); }

// Chunk address in the fp8 plane: the A planes' swizzle (a_lds_off) XOR 2 on rows 2, 3 (mod 4).  The plane is written
// one dword per lane (ds_write_b32: 32 lanes = 4 rows x 8 dwords); with the plain swizzle rows r and r + 2 put their 8
// dwords on the same 32 banks (2-way: SQ_LDS_BANK_CONFLICT 12 % of the LDS cycles); with the extra bit rows r, r + 1 take
// the two xl chunks' banks and rows r + 2, r + 3 the other two.  Rows with equal row & 3 get the same extra bit, so the
// 16-byte fragment reads stay conflict-free.
__device__ __forceinline__ int q_lds_off(int row, int chunk)
{
    return row * ROWB + ((chunk ^ ((row >> 2) & 3) ^ (((row >> 1) & 1) << 1)) << 4);
}

#define GGCN_SB() __builtin_amdgcn_sched_barrier(0)
// timing-only elimination ladder (lab_hooks.h: GGCN_LAB_OFF, 0 in the product build)
#define GGCN_ON(bit) constexpr (!((GGCN_LAB_OFF) & (bit)))

// RBLK: only the first `nblk` of this wavefront's four 32-row blocks hold nodes (wavefront-uniform run-time count: the second
// row group of a 256-row graph slot, fused_layer.hip wide8) -- the MFMAs of the others are skipped, their rows stay zero.
// BUF (needs AVEC and KFULL): X and W are read through buffer resources -- a base in SGPRs, a loop-invariant 32-bit lane
// offset and a SCALAR offset per stage -- instead of 64-bit pointers in vector registers: the per-stage address arithmetic
// (10 of ~90 VALU per stage: v_mad_i64_i32 / v_lshl_add_u64 per load) moves to the scalar unit.  bufx: the tile's X rows.
template <typename AT>
struct BufX {
    const AT *base;                       // first row of the workgroup's tile (workgroup-uniform)
    uint32_t bytes;                       // bytes of X from there to the end of the batch (reads beyond return zeros)
    uint32_t off[Geom<AT>::NP];           // this lane's piece of pass i at stage 0, in bytes from base (padding rows: 2^31, beyond any descriptor)
};
// X [total_rows, ldx]; the tile starts at base_row (a real row); rel[i] = this lane's row of pass i relative to it, -1 for a
// padding row (no select per value in the loop: the hardware's range check zeroes it).  The launchers admit the buffer path only when 257 rows of ldx elements stay below 2 GiB.
template <typename AT>
__device__ __forceinline__ BufX<AT> make_bufx(const AT *X, int64_t ldx, int64_t base_row, int64_t total_rows,
                                              const int (&rel)[Geom<AT>::NP], int tid)
{
    BufX<AT> b;
    const int64_t left = (total_rows - base_row) * ldx * (int64_t)sizeof(AT);
    b.base = X + base_row * ldx;
    b.bytes = (uint32_t)(left < 0x7fffffff ? left : 0x7fffffff);
#pragma unroll
    for (int i = 0; i < Geom<AT>::NP; ++i)   // rel < 0: a padding row -- an offset no descriptor reaches (bytes <= 2^31 - 1): its loads return zeros
        b.off[i] = rel[i] < 0 ? 0x80000000u
                              : (uint32_t)(((int64_t)rel[i] * ldx + (tid % Geom<AT>::TPR) * Geom<AT>::EPT) * (int64_t)sizeof(AT));
    return b;
}
// NB (fp32 rows: one staging pass = one 32-row block): only the first NB of this wavefront's four blocks exist at all -- the
// staging passes, fragment reads and MFMAs of the others are not compiled in (the second row group of a 129..224-node graph
// in a 256-row slot, fused_wide8.hip): their LDS rows are never written or read, their accumulators stay zero.
// W2 (GGCN_LAB_W2, lab): a second register set for W's fp16 fragments -- the next stage's are asked for at the TOP of a
// stage (a whole stage of lead) instead of behind its first MX MFMA (7 MFMAs of lead).
#ifndef GGCN_LAB_W2
#define GGCN_LAB_W2 0
#endif
// XS (ggcn_linear_scaled: the backward's dX): every value is multiplied by the power of two `xscale` before it is split -- rows
// whose magnitudes lie anywhere in the fp32 range (gradients) are brought to |x| < 256 first; the caller undoes it at the store.
// XP (fused_block8.hip, an experiment): this thread STAGES only XP of the tile's passes -- pass0, pass0 + 1, ... in the slots 0 ..
// XP - 1 of arow / avalid / bufx -- while its wavefront still multiplies all NB row blocks: two four-wavefront groups of one
// 512-thread workgroup share one set of X planes (each stages half of it) and meet at the same barriers.  SLOT0: the stage's slot
// (= 32-row block of MFMAs) behind which this thread's first staging pass sits -- the second group takes slots 2, 3, so that the two
// wavefronts of a SIMD (one of each group) do their split / store / load work behind DIFFERENT MFMAs.
template <typename AT, bool AVEC, bool KFULL, bool ZROWS, bool RBLK = false, bool BUF = false, int NB = 4, bool XS = false, int XP = 4, int SLOT0 = 0>
__device__ __forceinline__ void mainloop(const AT *const (&arow)[Geom<AT>::NP], const bool (&avalid)[Geom<AT>::NP],
                                         const char *__restrict__ wpack, int K, int stages_packed, int wm,
                                         int nt0, int n_tiles_total, char *lds, f32x16 (&acc)[4][RN], int rot = 0, int nblk = 4,
                                         float *amax_out = nullptr, const BufX<AT> *bufx = nullptr, float xscale = 1.0f, int pass0 = 0)
{
    static_assert(!BUF || (AVEC && KFULL && sizeof(AT) == 4), "buffer loads: whole 16-byte pieces of fp32 rows");
    static_assert(NB >= 1 && NB <= 4 && (NB == 4 || (sizeof(AT) == 4 && !RBLK)), "NB < 4: fp32 rows (pass i = block i), no run-time count");
    constexpr int NPL0 = Geom<AT>::NP < NB ? Geom<AT>::NP : NB;   // live staging passes
    constexpr int NPL = NPL0 < XP ? NPL0 : XP;                     // ... of which this thread performs XP (slot i = pass pass0 + i)
    static_assert(XP == 4 || (sizeof(AT) == 4 && NB == 4 && !RBLK), "XP < 4: fp32 rows, whole tiles");
    static_assert(SLOT0 >= 0 && SLOT0 + XP <= 4, "the staging passes sit behind the stage's four row-block slots");
    // rot: the K loop starts at stage `rot` and wraps around (same sum, another order).  The column
    // tiles of one row block run side by side on one XCD and read the same rows of X: started one
    // stage apart they find each other's lines in L2 instead of missing on them at the same moment.
    using G = Geom<AT>;
    constexpr int EPT = G::EPT, NP = G::NP, NQ = EPT / 4;
    static_assert(BK == 32 && NP <= 4 && RN == 2, "the slot schedule below is written for BK = 32, <= 4 passes, RN = 2");
    const int tid = threadIdx.x & (kThreads - 1);
    const int lane = tid & 63;
    const int s_k = (tid % G::TPR) * EPT;

    set_cvt_saturate(true);
    float amax = 0.0f;   // running max |x| of what this lane splits (range flag below)

    constexpr int kRsrcFlags = 0x00020000;   // raw buffer, 32-bit elements (gfx9 family)
    __amdgpu_buffer_rsrc_t xr, wr;
    if constexpr (BUF) {
        xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<AT *>(bufx->base), 0, (int)bufx->bytes, kRsrcFlags);
        wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(wpack), 0, 0x7fffffff, kRsrcFlags);
    }
    float ra[NP][EPT];
    auto load_a_pass = [&](int i, int k0) {
        const int gk = k0 + s_k;
        if constexpr (BUF) {
            const float4 t = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(xr, bufx->off[i], k0 * (int)sizeof(AT), 0));
            ra[i][0] = t.x; ra[i][1] = t.y; ra[i][2] = t.z; ra[i][3] = t.w;
        } else if constexpr (AVEC) {
            load16<AT>(arow[i] + ((KFULL || gk < K) ? gk : 0), ra[i]);
        } else {
#pragma unroll
            for (int c = 0; c < EPT; ++c) ra[i][c] = elem_to_float(arow[i][(gk + c < K) ? gk + c : 0]);
        }
    };
    // pass i of the stage that starts at k0: registers -> split -> LDS buffer `buf`
    // (plane 0: fp16 xh, 2 B per k; plane 1: per 4 k one {xl8 x4, xh8 x4} group of 8 B, so both planes
    //  take the same 8-byte write per lane and the same swizzled 16-byte fragment reads)
    Split4 sp[NQ];
    auto split_pass = [&](int i, int k0) {
        const int gk = k0 + s_k;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            float x[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                x[c] = ra[i][4 * q + c];
                if constexpr (!KFULL || ZROWS) {
                    bool in = true;
                    if constexpr (!KFULL) in = gk + 4 * q + c < K;
                    if constexpr (ZROWS && !BUF) in = in && avalid[i];   // (BUF: padding rows lie outside the descriptor and arrive as zeros)
                    x[c] = in ? x[c] : 0.0f;
                }
                if constexpr (XS) x[c] *= xscale;
            }
            sp[q] = split4(x[0], x[1], x[2], x[3], amax);
        }
    };
    auto write_pass = [&](int buf, int i) {
        char *h_plane = lds + buf * (2 * BM * ROWB);
        char *q_plane = h_plane + BM * ROWB;
        const int row = stage_row<AT>(XP == 4 ? i : i + pass0);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int kq = s_k + 4 * q;  // k offset inside the 32-deep stage
            *reinterpret_cast<uint2 *>(h_plane + a_lds_off(row, kq >> 3) + (kq & 4) * 2) = make_uint2(sp[q].h01, sp[q].h23);
            // q plane: 16-byte chunk hh = the four xl8 dwords lane half hh reads (k = 8hh..8hh+7, 16+8hh..16+8hh+7),
            // chunk 2+hh = its four xh8 dwords -- the MX operand's two scale blocks are two plain 16-byte reads
            const int g = kq >> 2, hh = (g >> 1) & 1, pos = (g & 1) + 2 * (g >> 2);
            *reinterpret_cast<uint32_t *>(q_plane + q_lds_off(row, hh) + 4 * pos) = (uint32_t)sp[q].l8;
            *reinterpret_cast<uint32_t *>(q_plane + q_lds_off(row, 2 + hh) + 4 * pos) = (uint32_t)sp[q].h8;
        }
    };

    // packed B: per stage [f16 k-step 0][f16 k-step 1][mx][scales]
    const char *bbase[RN];
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        const int ntc = nt0 + j < n_tiles_total ? nt0 + j : n_tiles_total - 1;
        bbase[j] = wpack + (int64_t)ntc * stages_packed * STAGE_PACK_BYTES + lane * 16;
    }
    uint32_t wtile[RN];   // BUF: uniform byte offsets of this wavefront's two column tiles inside the image
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        const int ntc = nt0 + j < n_tiles_total ? nt0 + j : n_tiles_total - 1;
        wtile[j] = (uint32_t)ntc * (uint32_t)stages_packed * STAGE_PACK_BYTES;
    }
    const uint32_t lane16 = lane * 16, lane4 = lane * 4;
    auto load_bf = [&](int st, f16x8 (&b0)[RN], f16x8 (&b1)[RN]) {  // fp16 fragments of both k-steps of stage st
        st = st < stages_packed ? st : stages_packed - 1;
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            if constexpr (BUF) {
                const uint32_t so = wtile[j] + (uint32_t)st * STAGE_PACK_BYTES;
                b0[j] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(wr, lane16, so, 0));
                b1[j] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(wr, lane16, so + 1024, 0));
            } else {
                const char *p = bbase[j] + (int64_t)st * STAGE_PACK_BYTES;
                b0[j] = *reinterpret_cast<const f16x8 *>(p);
                b1[j] = *reinterpret_cast<const f16x8 *>(p + 1024);
            }
        }
    };
    auto load_bq = [&](int st, i32x4 (&b)[RN], int (&sc)[RN]) {  // fp8 residual operand + scales of stage st
        st = st < stages_packed ? st : stages_packed - 1;
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            if constexpr (BUF) {
                const uint32_t so = wtile[j] + (uint32_t)st * STAGE_PACK_BYTES;
                b[j] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, lane16, so + kOffWl8, 0));
                sc[j] = __builtin_amdgcn_raw_buffer_load_b32(wr, lane4, so + kOffScales, 0);
            } else {
                const char *p = bbase[j] + (int64_t)st * STAGE_PACK_BYTES;  // bbase already holds lane * 16
                b[j] = *reinterpret_cast<const i32x4 *>(p + kOffWl8);
                sc[j] = *reinterpret_cast<const int *>(p + kOffScales - lane * 12);  // + lane * 4
            }
        }
    };
    auto load_wh8 = [&](int st, i32x4 (&b)[RN]) {   // GGCN_WH8_STORED: fp8(wh * 2^s0) of stage st as packed
        st = st < stages_packed ? st : stages_packed - 1;
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            if constexpr (BUF) b[j] = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, lane16, wtile[j] + (uint32_t)st * STAGE_PACK_BYTES + kOffWh8, 0));
            else b[j] = *reinterpret_cast<const i32x4 *>(bbase[j] + (int64_t)st * STAGE_PACK_BYTES + kOffWh8);
        }
    };
    // fp8(wh * 2^s0) of this lane's 16 fp16 values (its two fragments): byte 1 of the scale dword is the
    // E8M0 scale of block 0 (byte 0 is the scale the MFMA reads for this lane's own block)
    auto wh8_of = [&](const f16x8 &f0, const f16x8 &f1, int sc) -> i32x4 {
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        typedef short s2 __attribute__((ext_vector_type(2)));
        const float inv = __builtin_bit_cast(float, (sc & 0xff00) << 15);  // 2^(byte - 127): the converts DIVIDE by it
        i32x4 o;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const f16x8 &f = d < 2 ? f0 : f1;
            const int e = (d & 1) * 4;
            s2 q = __builtin_bit_cast(s2, undef_vgpr());   // (seeding with the fragment registers cost 3 v_mov per tile: the operand is a tuple)
            q = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(q, h2{f[e], f[e + 1]}, inv, false);
            q = __builtin_amdgcn_cvt_scalef32_pk_fp8_f16(q, h2{f[e + 2], f[e + 3]}, inv, true);
            o[d] = __builtin_bit_cast(int, q);
        }
        return o;
    };

    const int f_row = wm * 128 + (lane & 31);
    const int f_half = lane >> 5;
    const int scale_a = f_half ? SCALE_XH : SCALE_XL;  // lane r+32b carries the scale of block b
    auto read_h = [&](int buf, int i, f16x8 (&a)[2]) {
        const char *h_plane = lds + buf * (2 * BM * ROWB);
        const int row = f_row + i * 32;
        a[0] = *reinterpret_cast<const f16x8 *>(h_plane + a_lds_off(row, f_half));
        a[1] = *reinterpret_cast<const f16x8 *>(h_plane + a_lds_off(row, 2 + f_half));
    };
    auto read_q = [&](int buf, int i, i32x8 &a) {
        const char *q_plane = lds + buf * (2 * BM * ROWB) + BM * ROWB;
        const int row = f_row + i * 32;
        // chunk h: xl8 of k = 8h..8h+7, 16+8h..16+8h+7 (scale block 0); chunk 2+h: xh8 of the same k (block 1).
        // The dwords arrive in operand order: no register shuffle sits between the reads and the MFMA, so the
        // reads' latency hides behind the MFMAs issued before their first use.  (The earlier {xl8,xh8} pair
        // layout needed 8 moves per operand right behind the reads: an exposed LDS round trip four times a stage.)
        const i32x4 lo = *reinterpret_cast<const i32x4 *>(q_plane + q_lds_off(row, f_half));
        const i32x4 hi = *reinterpret_cast<const i32x4 *>(q_plane + q_lds_off(row, 2 + f_half));
        a = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
    };

#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    constexpr bool W2 = (GGCN_LAB_W2) != 0;
    f16x8 bset0[W2 ? 2 : 1][RN], bset1[W2 ? 2 : 1][RN];   // fp16 fragments of the two k-steps: [register set][column tile]
    f16x8 (&b0)[RN] = bset0[0], (&b1)[RN] = bset1[0];
    i32x4 bq[RN];  // fp8(wl) of the stage
    i32x4 bw[RN];  // fp8(wh) of the stage (GGCN_WH8_STORED)
    int sq[RN];
    const int stages = (K + BK - 1) / BK;

    // n-th stage of this workgroup's loop -> stage of the K axis (n past the end repeats the last one)
    auto kstage = [&](int n) {
        n = n < stages ? n : stages - 1;
        const int s = n + rot;
        return s >= stages ? s - stages : s;
    };
#pragma unroll
    for (int p = 0; p < NPL; ++p) load_a_pass(p, kstage(0) * BK);
    load_bf(kstage(0), b0, b1);
#pragma unroll
    for (int p = 0; p < NPL; ++p) {
        split_pass(p, kstage(0) * BK);
        write_pass(0, p);
    }
#pragma unroll
    for (int p = 0; p < NPL; ++p) load_a_pass(p, kstage(1) * BK);
    __syncthreads();

    // One stage = 8 slots of 128 matrix-pipe cycles each; the source order below IS the issue order
    // (a sched_barrier after every group), so every MFMA has a few independent VALU / LDS / memory
    // instructions behind it to issue in its shadow:
    //   slots 0-3 (block i): 4 fp16 MFMAs  | split + LDS write of pass i of the NEXT stage, the
    //                                        global load of pass i two stages ahead, the LDS reads
    //                                        of the next block's fragments (one slot ahead of use)
    //   slots 4-7 (block i): 2 MX  MFMAs   | the fp16 B fragments of the next stage, LDS reads
    // The MX operand of B is loaded at the top of its stage (used from slot 4 on).
    // ladder: operands the switched-off steps would have produced (dead code when GGCN_LAB_OFF == 0)
    f16x8 lab_h[2];
    i32x8 lab_q;
    i32x8 lab_bm[RN];
    if constexpr ((GGCN_LAB_OFF) != 0) {
        read_h(0, 0, lab_h);
        read_q(0, 0, lab_q);
        load_bq(kstage(0), bq, sq);
        if constexpr (GGCN_WH8_STORED) load_wh8(kstage(0), bw);
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            const i32x4 w = GGCN_WH8_STORED ? bw[j] : wh8_of(b0[j], b1[j], sq[j]);
            lab_bm[j] = i32x8{w[0], w[1], w[2], w[3], bq[j][0], bq[j][1], bq[j][2], bq[j][3]};
        }
    }
    auto stage = [&](int st, auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        const int k_next1 = kstage(st + 1) * BK;
        const int ka = kstage(st + 2) * BK;
        f16x8 ah[2][2];
        i32x8 aq[2];
        if GGCN_ON(8) read_h(buf, 0, ah[0]); else { ah[0][0] = lab_h[0]; ah[0][1] = lab_h[1]; }
        if GGCN_ON(16) { load_bq(kstage(st), bq, sq); if constexpr (GGCN_WH8_STORED) load_wh8(kstage(st), bw); }
        // W2: this stage's fragments live in set `buf` (stage parity = buffer parity); the next stage's are asked for now, BEHIND
        // this stage's bq (vmcnt retires in order: the MX phase's wait for bq must not wait for them)
        f16x8 (&b0)[RN] = bset0[W2 ? buf : 0], (&b1)[RN] = bset1[W2 ? buf : 0];
        if constexpr (W2) { if GGCN_ON(16) load_bf(kstage(st + 1), bset0[buf ^ 1], bset1[buf ^ 1]); }
        GGCN_SB();
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if GGCN_ON(8) {
                if (i < NB - 1) read_h(buf, i + 1, ah[(i + 1) & 1]);
                else read_q(buf, 0, aq[0]);
            } else {
                if (i < NB - 1) { ah[(i + 1) & 1][0] = lab_h[0]; ah[(i + 1) & 1][1] = lab_h[1]; }
                else aq[0] = lab_q;
            }
            GGCN_SB();
            if (!RBLK || i < nblk) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i & 1][0], b0[0], acc[i][0], 0, 0, 0);
            GGCN_SB();
            if GGCN_ON(2) { if (i - SLOT0 >= 0 && i - SLOT0 < NP && i - SLOT0 < GGCN_LAB_XPASSES && i - SLOT0 < XP) split_pass(i - SLOT0, k_next1); }
            GGCN_SB();
            if (!RBLK || i < nblk) acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i & 1][0], b0[1], acc[i][1], 0, 0, 0);
            GGCN_SB();
            if GGCN_ON(4) { if (i - SLOT0 >= 0 && i - SLOT0 < NP && i - SLOT0 < GGCN_LAB_XPASSES && i - SLOT0 < XP) write_pass(buf ^ 1, i - SLOT0); }
            else { _Pragma("unroll") for (int q = 0; q < NQ; ++q) asm volatile("" :: "v"(sp[q].h01), "v"(sp[q].h23), "v"(sp[q].l8), "v"(sp[q].h8)); }
            GGCN_SB();
            if (!RBLK || i < nblk) acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i & 1][1], b1[0], acc[i][0], 0, 0, 0);
            GGCN_SB();
            if GGCN_ON(1) { if (i - SLOT0 >= 0 && i - SLOT0 < NP && i - SLOT0 < GGCN_LAB_XPASSES && i - SLOT0 < XP) load_a_pass(i - SLOT0, ka); }
            else { if (i < NP) { _Pragma("unroll") for (int c = 0; c < EPT; ++c) asm volatile("" : "+v"(ra[i][c])); } }
            GGCN_SB();
            if (!RBLK || i < nblk) acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i & 1][1], b1[1], acc[i][1], 0, 0, 0);
            GGCN_SB();
        }
        // the MX operand of W = {fp8(wh) made here from the fp16 fragments, fp8(wl) as loaded}; b0/b1 are
        // dead afterwards and take the next stage's fragments.  The second column tile's converts sit behind the first
        // tile's first MX MFMA (GGCN_LAB_WH8 = 0: both tiles' converts in front of the MX phase, the round-2 order)
        i32x8 bm[RN];
        auto make_bm = [&](int j) {
            if GGCN_ON(32) {
                i32x4 w;
                if constexpr (GGCN_WH8_STORED) w = bw[j];
                else w = wh8_of(b0[j], b1[j], sq[j]);
                bm[j] = i32x8{w[0], w[1], w[2], w[3], bq[j][0], bq[j][1], bq[j][2], bq[j][3]};
            } else {
                bm[j] = lab_bm[j];
            }
        };
        make_bm(0);
        if constexpr (!(GGCN_LAB_WH8)) make_bm(1);
        GGCN_SB();
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            if GGCN_ON(8) { if (i < NB - 1) read_q(buf, i + 1, aq[(i + 1) & 1]); }
            else { if (i < NB - 1) aq[(i + 1) & 1] = lab_q; }
            GGCN_SB();
            if (!RBLK || i < nblk) acc[i][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(aq[i & 1], bm[0], acc[i][0], GGCN_LAB_MXFMT, GGCN_LAB_MXFMT, 0, scale_a, 0, sq[0]);
            GGCN_SB();
            if constexpr (GGCN_LAB_WH8) { if (i == 0) make_bm(1); }
            if constexpr (!W2) { if GGCN_ON(16) { if (i == 0) load_bf(kstage(st + 1), b0, b1); } }  // b0/b1 are dead: the fp16 MFMAs of this stage are all issued
            GGCN_SB();
            if (!RBLK || i < nblk) acc[i][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(aq[i & 1], bm[1], acc[i][1], GGCN_LAB_MXFMT, GGCN_LAB_MXFMT, 0, scale_a, 0, sq[1]);
            GGCN_SB();
        }
        if GGCN_ON(64) __syncthreads();
    };
    int st = 0;
    for (; st + 1 < stages; st += 2) {
        stage(st, std::integral_constant<int, 0>{});
        stage(st + 1, std::integral_constant<int, 1>{});
    }
    if (st < stages) stage(st, std::integral_constant<int, 0>{});
    set_cvt_saturate(false);
    if (amax_out) *amax_out = amax;   // the caller adds its own bound to the verdict (fused_layer.hip)
    else range_verdict(amax);
}
#undef GGCN_SB
#undef GGCN_ON

}  // namespace mx8
}  // namespace ggcn
