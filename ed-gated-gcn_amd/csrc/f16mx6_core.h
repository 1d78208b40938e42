// Main loop of the "f16mx6" linear: the f16mx8 scheme (f16mx8_core.h) with the correction products in fp6 (e2m3)
// instead of fp8 (e4m3).  gfx950 runs v_mfma_scale_f32_32x32x64_f8f6f4 in 32 cycles when BOTH operands are fp6/fp4
// and in 64 when either is fp8 (tools/probes/rate_probe.hip), so a 32x32x32 block costs 64 (fp16) + 32 = 96 matrix-pipe
// cycles instead of 128:
//
//   x = xh + xl,  xh = fp16(x),  xl = x - xh;   w = wh + wl likewise (ggcn_weight_pack, GGCN_PREC_F16MX6)
//   x.w = xh.wh                                   2 x v_mfma_f32_32x32x16_f16 per 32 k
//       + [ fp6(xl / sl) . fp6(wh / th) | fp6(xh / sx) . fp6(wl / tl) ]   ONE fp6 MX MFMA over the same 32 k
//
// e2m3 has the mantissa of e4m3 (4 significant bits) but only 2 exponent bits (values 0.125 .. 7.5): the fixed A-side
// scales of f16mx8 would leave the correction at fp16-product accuracy for small activations, so the A side gets TRUE
// block scales: per (row, 32 k) sx = 2^(E-2) with E the exponent of the block's largest |x| (the maximum lands in
// [4, 8) and saturates at 7.5), sl = sx * 2^-12 (|xl| <= 2^(E-11): [0, 8]).  numpy model on config-2 statistics
// (tools/mx_error_model.py): linear error max 5.6e-5 / rms 1.06e-5 against fp8's 4.9e-5 / 0.96e-5.
//
// Operand facts measured on gfx950 (tools/probes/fp6_scale_probe.hip, fp6_cvt_probe.hip), different from fp8's:
//   * with fp6 operands lane l = (r = l & 31, h = l >> 5) holds ALL 32 values of scale block h of row r (24 bytes), and
//     the scale byte of lane l covers exactly that lane's values: lane r carries block 0 = fp6(xl), lane r + 32 block 1 =
//     fp6(xh); the k order inside a block is free as long as A and B agree (here: field f <-> k = f);
//   * v_cvt_scalef32_pk32_fp6_f16 d, a[32 x f16], s: field j = fp6(a[j] / s), RNE, saturating at +-7.5; 64 cycles.
//
// What the per-block scales cost, and how the loop pays for it.  A convert instruction has ONE scale operand per lane, so
// the values it converts must belong to one (row, 32 k) block -- but a lane that loads many contiguous bytes of one row
// drags the vector memory pipe down 2.4x (tools/probes/ta_probe.hip: 64 lines per wave-instruction).  So X goes global ->
// LDS by LDS-DMA in the coalesced shape (8 lanes per 128-byte line: no registers, no address arithmetic), and the split
// reads it back from LDS with ONE HALF BLOCK (row, 16 k) per lane:
//   RAW[2]    fp32 stage as loaded: [128 rows][128 B], 16-byte chunk c of row r at chunk c ^ ((r >> 1) & 7)
//   PLANE[2]  H: fp16(x) [128 rows][64 B] (fragment order of f16mx8);  Q: [128 rows][fp6(xl) 24 B, E8M0 of sl, - | fp6(xh) 24 B,
//             E8M0 of sx, -]: a lane's operand and its scale are two 16-byte reads
// Per lane and stage: 4 x ds_read_b128, the maximum of its 16 values and one DPP exchange with the lane that holds the
// other half (the block's scale), 8 v_cvt_pk_f16_f32 + 16 v_fma_mixlo/hi_f16 (the residual as fp16: 11 of its <= 13 bits),
// 8 v_pk_mul_f16 (xl * 2^12, so that ONE convert with ONE scale turns [16 x xh | 16 x xl * 2^12] into both blocks' fields),
// a 3-dword DPP exchange with the neighbouring lane and 4 LDS stores of 16 bytes.  128 rows x 2 halves = 256 lanes: every wavefront splits in every stage, nobody waits at the barrier for a
// wavefront with more to do (a first version gave two of the four wavefronts a whole block per lane every other stage:
// 24 % slower than f16mx8, the other two idling), and the converts' operands are ordinary short-lived values -- no tuple
// is updated in place (that version needed 50 fixed registers and hand-written asm to keep hipcc from spilling 500).
#pragma once
#include "f16mx8_core.h"

namespace ggcn {
namespace mx6 {

using namespace bx3;
using mx8::f16x8;
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

constexpr int STAGE_PACK_BYTES = 2 * 1024 + 64 * 32;   // 4096: [f16 frag k-step 0][k-step 1][64 lanes x {fp6 block 24 B, E8M0 dword, pad}]
constexpr int kPlane = 0;                               // PLANE[buf] at buf * 16384: H at +0, Q at +8192
constexpr int kRaw = 32768;                             // RAW[buf] at kRaw + buf * 16384
constexpr int kLdsBytes6 = 65536;                       // two workgroups per CU

#define GGCN_SB6() __builtin_amdgcn_sched_barrier(0)

// 16-byte chunk c of row r inside an H or Q plane ([128 rows][64 B]): the A planes' swizzle (a_lds_off) with bit 0 of the chunk
// flipped on rows 2, 3 (mod 4).  A ds_write_b128 is served 8 lanes at a time over 32 banks (128 B): the 8 lanes of 4 split rows
// x 2 halves put rows r and r + 2 on the same banks with the plain swizzle (2-way: SQ_LDS_BANK_CONFLICT 20 % of the LDS
// cycles); with the extra bit they take the two chunk pairs of the 128-byte window.  The fragment reads (16 lanes over 64 banks)
// stay conflict-free: rows with equal row & 3 inside a lane group still differ in the chunk.
__device__ __forceinline__ int p6_off(int row, int chunk) { return row * ROWB + ((chunk ^ ((row >> 2) & 3) ^ ((row >> 1) & 1)) << 4); }

// 16-byte chunk c of row r inside a RAW stage
__device__ __forceinline__ int raw_off(int row, int c) { return row * 128 + ((c ^ ((row >> 1) & 7)) << 4); }

// xtile (workgroup-uniform, xtile_bytes valid bytes behind it) + aoff[j] (bytes, 32-bit): per DMA piece (rows 32 wave + 8 j ..
// + 7) this lane's source for stage 0 (its row, its swizzled chunk; rows past the batch are clamped to valid memory by the
// caller); uvalid: this lane's split row (32 wave + lane / 2) is a real node (false: the planes get zeros).  Every global
// access goes through a buffer resource (SGPRs) + a 32-bit lane offset + a scalar offset: no 64-bit address lives in a
// vector register.
template <bool ZROWS>
__device__ __forceinline__ void mainloop(const float *__restrict__ xtile, uint32_t xtile_bytes, const uint32_t (&aoff)[2],
                                         uint32_t piece_stride, bool uvalid, const char *__restrict__ wpack, int K,
                                         int stages_packed, int nt0, int n_tiles_total, char *lds, f32x16 (&acc)[4][RN],
                                         float *amax_out = nullptr)
{
    static_assert(RN == 2 && BK == 32, "written for 128 x 64 wavefront tiles and 32-deep stages");
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int stages = K / BK;
    constexpr int kRsrcFlags = 0x00020000;   // raw buffer, 32-bit elements (gfx9 family)
    // xtile_bytes: what is left of X from the tile's first row, at most 128 rows: a piece that reaches past the batch
    // (pieces 2, 3 of a graph slot with no graph) reads zeros instead of foreign memory
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(xtile), 0, (int)xtile_bytes, kRsrcFlags);

    // ---- X: LDS-DMA, 4 pieces of 1 KiB (8 rows x 128 B) per wavefront and stage.  Pieces j and j + 2 (16 rows apart) have
    // the same swizzle: two lane offsets + a uniform stride serve the four ----
    auto issue_piece = [&](int j, int st, int buf) {
        st = st < stages ? st : stages - 1;   // past the end: a harmless repeat (never read)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (__attribute__((address_space(3))) void *)(lds + kRaw + buf * 16384 + (wave * 4 + j) * 1024),
                                                 16, aoff[j & 1], st * (BK * 4) + (j >> 1) * piece_stride, 0, 0);
    };
    auto issue_dma = [&](int st, int buf) {
#pragma unroll
        for (int j = 0; j < 4; ++j) issue_piece(j, st, buf);
    };

    // ---- the split: lane = (row 32 wave + lane / 2, half hb = lane & 1) ----
    unsigned amax_bits = 0u;
    const int urow = 32 * wave + (lane >> 1), hb = lane & 1;
    const int s_raw = raw_off(urow, 4 * hb);              // chunk 4 hb + c = this address ^ (c << 4)
    const int s_h = p6_off(urow, 2 * hb);               // H plane chunks 2 hb, 2 hb + 1 (^ 16)
    const int s_qa = p6_off(urow, 2 * hb);              // Q plane: the even lane stores chunks 0, 1 (xl block), the odd lane 2, 3
    // The split in pieces, so that a stage can place them between its MFMA groups: the state lives in a SplitState.
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    typedef float f2 __attribute__((ext_vector_type(2)));
    struct SplitState { float x[16]; uint32_t ph[8], pl[8]; int eb; float sx; };
    auto split_read = [&](SplitState &t, int buf) {          // 4 x ds_read_b128 of RAW[buf]
        const char *raw = lds + kRaw + buf * 16384;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float4 v = *reinterpret_cast<const float4 *>(raw + (s_raw ^ (c << 4)));
            t.x[4 * c] = v.x; t.x[4 * c + 1] = v.y; t.x[4 * c + 2] = v.z; t.x[4 * c + 3] = v.w;
        }
    };
    // (hipcc moves side-effect-free work -- the maximum, the converts -- up to where its inputs arrive, in front of the MFMAs
    //  it was meant to follow; an empty volatile asm on the inputs pins a piece behind the scheduling barrier before it)
#define GGCN_PIN4(a, b, c, d) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d))
    auto split_scale = [&](SplitState &t) {                  // block maximum (16 values, then the neighbouring lane's) -> scales
#pragma unroll
        for (int i = 0; i < 16; i += 4) GGCN_PIN4(t.x[i], t.x[i + 1], t.x[i + 2], t.x[i + 3]);
        if constexpr (ZROWS) {
            if (!uvalid) {
#pragma unroll
                for (int i = 0; i < 16; ++i) t.x[i] = 0.0f;
            }
        }
        float m;
        asm("v_max3_f32 %0, |%1|, |%2|, |%3|" : "=v"(m) : "v"(t.x[0]), "v"(t.x[1]), "v"(t.x[2]));
#pragma unroll
        for (int i = 3; i < 15; i += 2) asm("v_max3_f32 %0, %1, |%2|, |%3|" : "=v"(m) : "v"(m), "v"(t.x[i]), "v"(t.x[i + 1]));
        asm("v_max_f32 %0, %1, |%2|" : "=v"(m) : "v"(m), "v"(t.x[15]));
        const float mo = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(m), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
        const unsigned mb = __float_as_uint(m) > __float_as_uint(mo) ? __float_as_uint(m) : __float_as_uint(mo);   // both >= 0: integer order
        amax_bits = amax_bits > mb ? amax_bits : mb;   // the sticky range flag of f16mx8_core.h: largest |x| this lane has split
        int eb = (int)(mb >> 23);
        t.eb = eb < 15 ? 15 : (eb > 254 ? 254 : eb);   // kept >= 15: eb - 14 is an E8M0 byte; inf / NaN: the fp16 product decides
        t.sx = __uint_as_float((unsigned)(t.eb - 2) << 23);   // 2^(E-2): the convert DIVIDES by it
    };
    auto split_pairs = [&](SplitState &t, int t0, int n) {   // fp16 pairs t0 .. t0 + n - 1, residuals as fp16 x 2^12
#pragma unroll
        for (int k = 0; k < n; k += 2) GGCN_PIN4(t.x[2 * (t0 + k)], t.x[2 * (t0 + k) + 1], t.x[2 * (t0 + k) + 2], t.x[2 * (t0 + k) + 3]);
#pragma unroll
        for (int k = 0; k < n; ++k) {
            const int p = t0 + k;
            const h2 pa = __builtin_convertvector(f2{t.x[2 * p], t.x[2 * p + 1]}, h2);
            // the residual, rounded to fp16 and scaled: v_fma_mixlo / mixhi_f16 + v_pk_mul_f16
            const h2 r = {(_Float16)(t.x[2 * p] - (float)pa[0]), (_Float16)(t.x[2 * p + 1] - (float)pa[1])};
            const h2 scaled = r * h2{(_Float16)4096.0f, (_Float16)4096.0f};
            t.ph[p] = __builtin_bit_cast(uint32_t, pa);
            t.pl[p] = __builtin_bit_cast(uint32_t, scaled);
        }
    };
    auto split_store_h = [&](SplitState &t, int buf, int half) {   // 4 pairs = 8 k -> one 16-byte chunk of the H plane
        char *hp = lds + kPlane + buf * 16384;
        *reinterpret_cast<uint4 *>(hp + (s_h ^ (half << 4))) = make_uint4(t.ph[4 * half], t.ph[4 * half + 1], t.ph[4 * half + 2], t.ph[4 * half + 3]);
    };
    auto split_convert = [&](SplitState &t, int buf) {       // [16 x xh | 16 x xl 2^12] -> both blocks' fields, Q plane
        char *qp = lds + kPlane + buf * 16384 + 8192;
        GGCN_PIN4(t.ph[0], t.ph[1], t.pl[0], t.pl[1]);
        const i32x16 src = {(int)t.ph[0], (int)t.ph[1], (int)t.ph[2], (int)t.ph[3], (int)t.ph[4], (int)t.ph[5], (int)t.ph[6], (int)t.ph[7],
                            (int)t.pl[0], (int)t.pl[1], (int)t.pl[2], (int)t.pl[3], (int)t.pl[4], (int)t.pl[5], (int)t.pl[6], (int)t.pl[7]};
        const u32x6 q = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(f16x32, src), t.sx);
        // q[0..2]: this half's 16 fields of fp6(xh / sx) (block 1 of the row), q[3..5]: of fp6(xl / sl) (block 0).  A block's 24
        // bytes come from two lanes; scattered 4- and 8-byte stores of both (10 store instructions under two exec masks,
        // 2- to 4-way bank conflicts) cost 67 us of a 700 us launch, so the lanes trade halves through DPP -- the even lane
        // takes the whole xl block, the odd lane the whole xh block -- and each stores its block and its scale as two
        // conflict-free 16-byte writes.
        if constexpr (((GGCN_LAB_OFF) & 4) != 0) { asm volatile("" :: "v"(q[0]), "v"(q[1]), "v"(q[2]), "v"(q[3]), "v"(q[4]), "v"(q[5])); return; }
        uint32_t recv[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const uint32_t send = hb ? q[3 + i] : q[i];      // what the OTHER lane's block lacks
            recv[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)send, 0xB1, 0xF, 0xF, true);   // quad_perm [1,0,3,2]
        }
        // even lane (hb 0): [own xl half | partner's xl half], odd lane: [partner's xh half | own xh half]
        const uint32_t d0 = hb ? recv[0] : q[3], d1 = hb ? recv[1] : q[4], d2 = hb ? recv[2] : q[5];
        const uint32_t d3 = hb ? q[0] : recv[0], d4 = hb ? q[1] : recv[1], d5 = hb ? q[2] : recv[2];
        const uint32_t e8 = (uint32_t)(t.eb - (hb ? 2 : 14));   // E8M0 of sx (xh block) / of sl = sx * 2^-12 (xl block)
        *reinterpret_cast<uint4 *>(qp + s_qa) = make_uint4(d0, d1, d2, d3);
        *reinterpret_cast<uint4 *>(qp + (s_qa ^ 16)) = make_uint4(d4, d5, e8, 0u);
    };
    auto split_half = [&](int buf) {   // the whole split in one go (prologue): RAW[buf] -> PLANE[buf]
        SplitState t;
        split_read(t, buf);
        split_scale(t);
        split_pairs(t, 0, 8);
        split_store_h(t, buf, 0);
        split_store_h(t, buf, 1);
        split_convert(t, buf);
    };

    // ---- packed B: per stage [f16 k-step 0][f16 k-step 1][fp6 block + scale, 32 B per lane] ----
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(wpack), 0, 0x7fffffff, kRsrcFlags);
    uint32_t wtile[RN];   // uniform byte offsets of this wavefront's two column tiles
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        const int ntc = nt0 + j < n_tiles_total ? nt0 + j : n_tiles_total - 1;
        wtile[j] = (uint32_t)ntc * (uint32_t)stages_packed * STAGE_PACK_BYTES;
    }
    const uint32_t lane16 = lane * 16, lane32 = lane * 32;
    auto load_bf0 = [&](int st, f16x8 (&b0)[RN]) {
        st = st < stages_packed ? st : stages_packed - 1;
#pragma unroll
        for (int j = 0; j < RN; ++j)
            b0[j] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(wr, lane16, wtile[j] + (uint32_t)st * STAGE_PACK_BYTES, 0));
    };
    auto load_bf1 = [&](int st, f16x8 (&b1)[RN]) {
        st = st < stages_packed ? st : stages_packed - 1;
#pragma unroll
        for (int j = 0; j < RN; ++j)
            b1[j] = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(wr, lane16, wtile[j] + (uint32_t)st * STAGE_PACK_BYTES + 1024, 0));
    };
    auto load_bq = [&](int st, i32x8 (&bm)[RN]) {   // dwords 0-5: the fp6 block, dword 6: its E8M0 scale
        st = st < stages_packed ? st : stages_packed - 1;
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            const uint32_t so = wtile[j] + (uint32_t)st * STAGE_PACK_BYTES + 2048;
            const i32x4 lo = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, lane32, so, 0));
            const i32x4 hi = __builtin_bit_cast(i32x4, __builtin_amdgcn_raw_buffer_load_b128(wr, lane32, so + 16, 0));
            bm[j] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, -1);
        }
    };

    const int f_row = lane & 31, f_half = lane >> 5;
    auto read_h1 = [&](int buf, int i, int s, f16x8 &a) {   // k-step s of row block i
        a = *reinterpret_cast<const f16x8 *>(lds + kPlane + buf * 16384 + p6_off(f_row + i * 32, 2 * s + f_half));
    };
    auto read_q = [&](int buf, int i, i32x8 &a, int &sc) {
        const char *qp = lds + kPlane + buf * 16384 + 8192;
        const int row = f_row + i * 32;
        const i32x4 lo = *reinterpret_cast<const i32x4 *>(qp + p6_off(row, 2 * f_half));
        const i32x4 hi = *reinterpret_cast<const i32x4 *>(qp + p6_off(row, 2 * f_half + 1));   // {dwords 4, 5, scale, -}
        a = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        sc = hi[2];
    };

#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    f16x8 b0[RN], b1[RN];
    i32x8 bm[RN];

    // ---- prologue: stages 0 and 1 on their way, stage 0 split ----
    issue_dma(0, 0);
    issue_dma(1, 1);
    load_bf0(0, b0);
    load_bf1(0, b1);
    load_bq(0, bm);
    __syncthreads();                      // vmcnt(0) + barrier: every piece has landed
    split_half(0);
    __syncthreads();

    // One stage: the split of stage st + 1 (RAW[buf ^ 1] -> PLANE[buf ^ 1]); 8 fp16 MFMAs of k-step 0 (b0), 8 of k-step 1
    // (b1), 8 fp6 MFMAs (bm).  Each W register set is re-requested for the next stage as soon as its last MFMA has issued --
    // b0 after the first eight, b1 after the second eight, bm at the end -- so every load has more than half a stage to land
    // (with 32-cycle correction MFMAs the whole MX phase is 256 cycles: fragments requested at its start, as in f16mx8,
    // arrived late and cost 70 us).
    auto stage = [&](int st, auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        f16x8 ah[3];
        i32x8 aq[2];
        int sa[2];
        constexpr bool SPLIT = !((GGCN_LAB_OFF) & 2);
        SplitState sp;
        if constexpr (SPLIT) split_read(sp, buf ^ 1);   // (the last stage splits a repeat of itself: nobody reads it)
        read_h1(buf, 0, 0, ah[0]);
        read_h1(buf, 1, 0, ah[1]);
        GGCN_SB6();
        // ---- k-step 0 ----
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i + 2 < 4) read_h1(buf, i + 2, 0, ah[(i + 2) % 3]);
            else read_h1(buf, i - 2, 1, ah[(i + 2) % 3]);        // i = 2, 3: the first fragments of k-step 1
            GGCN_SB6();
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i % 3], b0[0], acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i % 3], b0[1], acc[i][1], 0, 0, 0);
            GGCN_SB6();
            if constexpr (!((GGCN_LAB_OFF) & 1)) issue_piece(i, st + 2, buf);  // one DMA piece per row block (RAW[buf] was split a stage ago)
            if constexpr (SPLIT) {   // the split of stage st + 1, a piece behind every MFMA pair
                if (i == 0) split_scale(sp);
                if (i == 1) split_pairs(sp, 0, 2);
                if (i == 2) { split_pairs(sp, 2, 2); split_store_h(sp, buf ^ 1, 0); }
                if (i == 3) split_pairs(sp, 4, 2);
            }
            GGCN_SB6();
        }
        if constexpr (!((GGCN_LAB_OFF) & 16)) load_bf0(st + 1, b0);
        GGCN_SB6();
        // ---- k-step 1 (fragment t = 4 + i sits in ah[(4 + i) % 3]) ----
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i + 2 < 4) read_h1(buf, i + 2, 1, ah[(4 + i + 2) % 3]);
            else read_q(buf, i - 2, aq[i - 2], sa[i - 2]);    // i = 2, 3: the first two fp6 operands
            GGCN_SB6();
            acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[(4 + i) % 3], b1[0], acc[i][0], 0, 0, 0);
            acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[(4 + i) % 3], b1[1], acc[i][1], 0, 0, 0);
            GGCN_SB6();
            if constexpr (SPLIT) {
                if (i == 0) { split_pairs(sp, 6, 2); split_store_h(sp, buf ^ 1, 1); }
                if (i == 1) split_convert(sp, buf ^ 1);
                GGCN_SB6();
            }
        }
        if constexpr (!((GGCN_LAB_OFF) & 16)) load_bf1(st + 1, b1);
        GGCN_SB6();
        // ---- fp6 corrections ----
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            acc[i][0] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(aq[i & 1], bm[0], acc[i][0], 2, 2, 0, sa[i & 1], 0, bm[0][6]);
            acc[i][1] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(aq[i & 1], bm[1], acc[i][1], 2, 2, 0, sa[i & 1], 0, bm[1][6]);
            GGCN_SB6();
            if (i + 2 < 4) read_q(buf, i + 2, aq[i & 1], sa[i & 1]);
            GGCN_SB6();
        }
        if constexpr (!((GGCN_LAB_OFF) & 16)) load_bq(st + 1, bm);
        __syncthreads();
    };
    int st = 0;
    for (; st + 1 < stages; st += 2) {
        stage(st, std::integral_constant<int, 0>{});
        stage(st + 1, std::integral_constant<int, 1>{});
    }
    if (st < stages) stage(st, std::integral_constant<int, 0>{});
    // the sticky range flag (f16mx8_core.h): the block scales have no accuracy window, the fp16 main product has fp16's range
    const float amax = amax_bits >= 0x7F800000u ? __builtin_inff() : __uint_as_float(amax_bits);
    if (amax_out) *amax_out = amax;
    else mx8::range_verdict(amax, 0.0f, 0.0f, false);
}
#undef GGCN_SB6
#undef GGCN_PIN4

}  // namespace mx6
}  // namespace ggcn
