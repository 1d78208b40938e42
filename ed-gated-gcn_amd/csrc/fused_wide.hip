// One launch per gated layer for graphs of 33..128 nodes (LitBank: ORI_ML = 100, constant.py:227): see fused_layer.hip for
// the scheme (models/gcn.py:34-45 + bert_amir5.py:627-640 in one kernel, `hidden` never leaving the CU).
#include "fused_common.h"

namespace ggcn {
namespace {

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
// DROP (training, bert_amir5.py:621-625): the three gates are dropped per (token, feature) -- keep factors from the
// counter-based hash of dropout_hash.h, the same ones ggcn_gate_pool_backward_drop and ggcn_dropout_mask draw -- and the
// pools maximise the gated, kept values themselves (every element has its own factor: no max / min shortcut).
template <int SCH, bool AVEC, bool KFULL, bool VST, int SB, bool DROP = false>
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
        else {
            constexpr bool BUF = AVEC && KFULL;   // buffer loads (f16mx8_core.h): offsets from the workgroup's first graph
            mx8::BufX<float> bx;
            if constexpr (BUF) {
                int rel[NP];
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    const int row = stage_row<float>(i) + 128 * hh;
                    rel[i] = avalid[i] ? (row / S) * T + row % S : -1;
                }
                bx = mx8::make_bufx<float>(a.X, a.ldx, (int64_t)g0 * T, (int64_t)B * T, rel, tid);
            }
            mx8::mainloop<float, AVEC, KFULL, true, false, BUF>(arow, avalid, lp.wpack, K, a.k_steps / 2, wm, nt0, n_tiles_total, lds, acc, 0, 4,
                                                                 nullptr, &bx);
        }
        // `pre` (ggcn_layer_fused_prebias): y = D.A.(hidden + 1.pre^T) + bias -- the folded second layer of the eval form, whose
        // input rows are already aggregated once (gated_block.py).  Workgroup-uniform; padding rows are nobody's neighbours.
        if (lp.pre) {
#pragma unroll
            for (int j = 0; j < RN; ++j) {
                const int gn = (nt0 + j) * NT + (lane & 31);
                const float pv = lp.pre[gn < F ? gn : 0];
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[i][j][r] += pv;
            }
        }
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
            float vmax[RN], vmin[RN], pmax_a[RN], pmax_b[RN];
#pragma unroll
            for (int j = 0; j < RN; ++j) { vmax[j] = -INFINITY; vmin[j] = INFINITY; pmax_a[j] = -INFINITY; pmax_b[j] = -INFINITY; }
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
                    const float gaj = pool_gate_a ? vga[s][j] : 1.0f, gbj = pool_gate_b ? vgb[s][j] : 1.0f;
                    const uint32_t didx0 = DROP ? (uint32_t)(((int64_t)g * T + node0 + 4 * h) * F + (nt0 + j) * NT + c) : 0u;   // element of row 4h of the block
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row0 = (r & 3) + 8 * (r >> 2);  // this lane's row is row0 + 4h
                        const float v = y[r] * rinv[r] + bj;      // gcn.py:41,43
                        float vs = v * sg;
                        if constexpr (DROP) {
                            const uint32_t hh = drop_hash(didx0 + (uint32_t)(row0 * F), a.drop.seed_lo, a.drop.seed_hi);
                            vs *= drop_keep(hh, a.drop.sel[0], a.drop.thr, a.drop.scale);
                            if (node0 + row0 + 4 * h < T) {
                                pmax_a[j] = fmaxf(pmax_a[j], v * gaj * drop_keep(hh, a.drop.sel[1], a.drop.thr, a.drop.scale));
                                pmax_b[j] = fmaxf(pmax_b[j], v * gbj * drop_keep(hh, a.drop.sel[2], a.drop.thr, a.drop.scale));
                            }
                        }
                        if (vst) stage_lds[(row0 + 4 * h) * 64 + ((32 * j + c) ^ (32 * h))] = vs;
                        if (node0 + row0 + 4 * h < T) {
                            if (direct_store && col_ok[j]) tile[lane_off + row0 * ldo] = vs;
                            if constexpr (!DROP) {
                                vmax[j] = fmaxf(vmax[j], v);
                                vmin[j] = fminf(vmin[j], v);
                            }
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
                float pa, pb;
                if constexpr (DROP) {
                    pa = fmaxf(pmax_a[j], upper_half_to_lower(pmax_a[j]));
                    pb = fmaxf(pmax_b[j], upper_half_to_lower(pmax_b[j]));
                } else {
                    const float mx = fmaxf(vmax[j], upper_half_to_lower(vmax[j]));
                    const float mn = fminf(vmin[j], upper_half_to_lower(vmin[j]));
                    const float ga = pool_gate_a ? vga[s][j] : 1.0f, gb = pool_gate_b ? vgb[s][j] : 1.0f;
                    pa = ga * (ga >= 0.0f ? mx : mn);
                    pb = gb * (gb >= 0.0f ? mx : mn);
                }
                if (h == 0 && col_ok[j]) {
                    const int gn = (nt0 + j) * NT + c;
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


}  // namespace

int launch_fused_wide(const char *who, const FusedArgs &a, int precision, int sb, bool fast, bool vst, int64_t gridw, hipStream_t st)
{
#define GGCN_LAUNCHW(SC, AV, KF, VS, SBV, DR) \
    hipLaunchKernelGGL((layer_fused_wide_kernel<SC, AV, KF, VS, SBV, DR>), dim3((unsigned)gridw), dim3(kThreads), 0, st, a)
#define GGCN_PICKWD(SC, SBV, DR)                                             \
    do {                                                                     \
        if (fast && vst) GGCN_LAUNCHW(SC, true, true, true, SBV, DR);        \
        else if (fast) GGCN_LAUNCHW(SC, true, true, false, SBV, DR);         \
        else GGCN_LAUNCHW(SC, false, false, false, SBV, DR);                 \
    } while (0)
#define GGCN_PICKW(SC, SBV)                                                  \
    do {                                                                     \
        if (a.drop.thr != 0) GGCN_PICKWD(SC, SBV, true);                     \
        else GGCN_PICKWD(SC, SBV, false);                                    \
    } while (0)
#if GGCN_LAB_WIDE_SB8
    if (sb == 8) { if (precision == GGCN_PREC_F16MX8) GGCN_PICKW(1, 8); else GGCN_PICKW(0, 8); return check_launch(who); }
#endif
    if (sb != 2 && sb != 4) return fail(GGCN_EUNSUPPORTED, "%s: no %d-row graph slot in this build", who, 32 * sb);
    if (precision == GGCN_PREC_F16MX8) { if (sb == 2) GGCN_PICKW(1, 2); else GGCN_PICKW(1, 4); }
    else { if (sb == 2) GGCN_PICKW(0, 2); else GGCN_PICKW(0, 4); }
#undef GGCN_PICKW
#undef GGCN_PICKWD
#undef GGCN_LAUNCHW
    return check_launch(who);
}

GGCN_RANGE_FLAG_TU(range_flag_wide)

}  // namespace ggcn
