// EXPERIMENT (`ggcn_lab_block_fused8`, nothing in the product calls it): the two-layer block with ONE
// workgroup of EIGHT wavefronts per (row block, 256-column slice) that shares a row block's X planes between its W1 and its W12
// column tiles.  Wavefronts 0-3 (group 0) own the slice's W1 tiles, wavefronts 4-7 (group 1) its W12 tiles; every thread stages 2
// of the stage's 4 passes (mx8::mainloop<..., XP = 2>), so X is loaded, split and written to LDS once for both layers -- half the
// X-side work per output (48 + 24 us of the elimination ladder) and one fetch of X per slice instead of one per part.
// What it gives up is what DESIGN.md 5b says it gives up: 8 x 235 registers fill the CU, so ONE workgroup is resident and both
// groups' epilogues run under nothing.  Block ids that share an XCD take a contiguous run of (slice-major) work items, so an XCD
// holds at most two column slices of both weight images (2.5 MB of its 4 MiB L2) -- the price: a row block's three slices sit on
// three XCDs.  Same tiles, same arithmetic, same order as ggcn_block_fused: results are bit-identical (test).  f16mx8, T <= 32,
// fast shapes (K % 32 == 0, 16-byte rows), no gcn1 output.
#include "fused_common.h"

#include <cstdlib>

namespace ggcn {
namespace {

constexpr int kB8Threads = 512;
constexpr int kB8Lds = kLdsBytes + 2 * kEpiLdsBytes;   // one set of stage buffers, two sets of epilogue operands: 94.5 KiB

__device__ __forceinline__ bool getenv_slot_same(const FusedArgs &a) { return (a.k_steps & 0x20000000) != 0; }   // (lab: both groups stage behind slots 0, 1)

template <bool FULLT, bool VST>
__global__ __launch_bounds__(kB8Threads, 1) void block_fused8_kernel(const FusedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int B = a.B, T = a.T, K = a.K, F = a.F;
    // work item w = n_wgi * g_tiles + g_tile (slice-major); XCD x (= id & 7: observed dispatch, speed only) takes a contiguous run
    int n_wgi, g_tile;
    if (a.n_parts == 2) {
        const int xcd = blockIdx.x & 7, qb = blockIdx.x >> 3;
        const int total = a.g_tiles * a.n_wg, per_xcd = (total + 7) >> 3;
        const int w = xcd * per_xcd + qb;
        if (qb >= per_xcd || w >= total) return;   // whole workgroup, before any barrier
        n_wgi = w / a.g_tiles;
        g_tile = w - n_wgi * a.g_tiles;
    } else {   // (GGCN_LAB_BLOCK8_ROWMAJOR=1: a row block's slices on ONE XCD -- X fetched once, both weight images in every L2)
        if (!tile_of_block(blockIdx.x, a.g_tiles, a.n_wg, g_tile, n_wgi)) return;
    }

    const int tid8 = threadIdx.x;
    const unsigned long long t_start = a.stamps ? __builtin_amdgcn_s_memrealtime() : 0;   // (lab: phase stamps in 10 ns ticks, per group)
    const int group = __builtin_amdgcn_readfirstlane(tid8 >> 8);   // 0: the W1 tiles, 1: the W12 tiles (wavefront-uniform)
    const int tid = tid8 & 255;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const LayerPart &lp = a.part[group];
    const int gt0 = g_tile * 4, g0 = gt0;
    const int n_tiles_total = (F + NT - 1) / NT;
    const int nt0 = n_wgi * (BN / NT) + wave * RN;

    // this thread's two staging passes: passes 2 group, 2 group + 1 of the tile (pass p = rows 32 p .. 32 p + 31 = graph gt0 + p)
    constexpr int NP = Geom<float>::NP;
    const float *arow[NP];
    bool avalid[NP];
    int rel[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
        const int row = stage_row<float>(i < 2 ? i + 2 * group : 0);
        const int g = gt0 + (row >> 5), r = row & 31;
        avalid[i] = i < 2 && (g < B) && (FULLT || r < T);
        arow[i] = a.X;   // (buffer path only)
        rel[i] = avalid[i] ? (row >> 5) * T + (row & 31) : -1;
    }
    // whole tiles take the LDS-DMA staging (workgroup-uniform; a.k_steps & 1 carries the launcher's alignment verdict)
    const bool dma = (a.n_wg * BN == F) && (gt0 + 4 <= B) && a.ov_in == nullptr && a.ov_out == nullptr && (a.k_steps & 0x40000000) != 0;
    const int k_steps = a.k_steps & 0x1fffffff;
    if (dma) {
        if (group == 0) stage_epilogue_operands_dma<kLdsBytes>(a, lp, g0, n_wgi, lds, tid);
        else stage_epilogue_operands_dma<kLdsBytes + kEpiLdsBytes>(a, lp, g0, n_wgi, lds, tid);
    } else {
        if (group == 0) stage_epilogue_operands<kLdsBytes>(a, lp, g0, n_wgi, lds, tid);
        else stage_epilogue_operands<kLdsBytes + kEpiLdsBytes>(a, lp, g0, n_wgi, lds, tid);
    }

    f32x16 acc[4][RN];
    float amax;
    const unsigned long long t_loop0 = a.stamps ? __builtin_amdgcn_s_memrealtime() : 0;
    mx8::BufX<float> bx = mx8::make_bufx<float>(a.X, a.ldx, (int64_t)gt0 * T, (int64_t)B * T, rel, tid);
    // (two instantiations that meet at the same barriers: group 1's staging passes sit behind the stage's LAST two row blocks)
    if (group == 0 || getenv_slot_same(a))
        mx8::mainloop<float, true, true, !FULLT, false, true, 4, false, 2, 0>(arow, avalid, lp.wpack, K, k_steps / 2, 0, nt0, n_tiles_total, lds, acc, 0, 4,
                                                                               &amax, &bx, 1.0f, 2 * group);
    else
        mx8::mainloop<float, true, true, !FULLT, false, true, 4, false, 2, 2>(arow, avalid, lp.wpack, K, k_steps / 2, 0, nt0, n_tiles_total, lds, acc, 0, 4,
                                                                               &amax, &bx, 1.0f, 2 * group);
    const int64_t pack_bytes = (int64_t)n_tiles_total * (k_steps / 2) * mx8::STAGE_PACK_BYTES;
    unsigned long long t_loop1 = 0;
    if (a.stamps) { asm volatile("" :: "v"(acc[3][RN - 1][15])); t_loop1 = __builtin_amdgcn_s_memrealtime(); }
    if (group == 0) {
        if (dma) dma_range_verdict<kLdsBytes>(amax, lp.wpack, pack_bytes, lds, tid & 63);
        else fused_range_verdict<kLdsBytes>(amax, lp.wpack, pack_bytes, lds, true);
        epilogue<1, FULLT, VST, false, false, kLdsBytes>(a, lp, acc, g0, nt0, n_tiles_total, lds, tid);
    } else {
        if (dma) dma_range_verdict<kLdsBytes + kEpiLdsBytes>(amax, lp.wpack, pack_bytes, lds, tid & 63);
        else fused_range_verdict<kLdsBytes + kEpiLdsBytes>(amax, lp.wpack, pack_bytes, lds, true);
        if (lp.out) epilogue<1, FULLT, VST, true, true, kLdsBytes + kEpiLdsBytes>(a, lp, acc, g0, nt0, n_tiles_total, lds, tid);
        else epilogue<1, FULLT, VST, true, false, kLdsBytes + kEpiLdsBytes>(a, lp, acc, g0, nt0, n_tiles_total, lds, tid);
    }
    if (a.stamps && tid == 0) {
        unsigned long long *o = a.stamps + ((size_t)blockIdx.x * 2 + group) * 4;
        o[0] = t_start; o[1] = t_loop0; o[2] = t_loop1; o[3] = __builtin_amdgcn_s_memrealtime();
    }
}

}  // namespace

// What ggcn_block_fused hands to this kernel (fused_layer.hip): the whole block with all of its outputs, f16mx8, on batches that
// make 6 and more rounds of one eight-wavefront workgroup per CU (or 3 and more whole rounds).  There, with the board at its power
// cap, the X planes staged once for a W1 and a W12 column slice are worth 2 % of the launch IN STEADY STATE (each form alone for
// 1.5 s: 603-604 vs 615-618 us at 4096 graphs on two boxes, 639 vs 653 on a slower one, 304-309 vs 313 at 2048; 512 graphs: 85 vs
// 80 us, so shards keep the four-wavefront kernel) -- while five-launch interleaved timings had shown it 3-6 % SLOWER:
// the power controller averages over milliseconds, and the benchmark's timed region is a steady state.
// One eight-wavefront workgroup per CU: a partial last round costs a whole one (1536 graphs = 4.5 rounds: 243 vs 237 us), so the
// kernel is taken from 6 rounds up (2048 graphs x 768 columns) or for 3 and more WHOLE rounds (1024 graphs: 149.5 vs 152-153 us).
int device_cu_count()
{
    static int cus = 0;   // (one device kind per process: MI355X, 256)
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cus = n;
    }
    return cus;
}
// the shape's share of the rule (ggcn_block_fused_form: what bench.py labels its roofline line with, what the tests ask)
bool block8_shape(int B, int T, int K, int F)
{
    if (B <= 0 || T > 32 || T <= 0 || F <= 0 || K <= 0 || F % BN != 0 || K % BK != 0) return false;
    const int64_t total = (int64_t)((B + 3) / 4) * (F / BN), cus = device_cu_count();
    return total >= 6 * cus || (total >= 3 * cus && total % cus == 0);
}
bool block8_takes(const float *X, int64_t ldx, int B, int T, int K, int F, const float *gate1, const float *gate2, const float *bias1,
                  const float *bias_mid, const float *bias2, const void *graph_ops, const void *graph_ops2, const float *x_out, int64_t ld2)
{
    if (!block8_shape(B, T, K, F)) return false;
    if (!((ldx % 4 == 0) && aligned16(X) && (int64_t)ldx * 4 * 257 < ((int64_t)1 << 31))) return false;
    if (!graph_ops || !graph_ops2 || !gate1 || !gate2 || !bias_mid) return false;
    if (!(aligned16(gate1) && aligned16(gate2) && aligned16(bias1) && aligned16(bias2) && aligned16(bias_mid) && aligned16(graph_ops) &&
          aligned16(graph_ops2)))
        return false;   // (the LDS-DMA staging of the epilogue's operands: the form that was measured)
    if (x_out && !((ld2 % 4 == 0) && aligned16(x_out) && (int64_t)T * ld2 * 4 < ((int64_t)1 << 31))) return false;
    return true;
}

int lab_block_fused8(const float *X, int64_t ldx, const void *wpack1, const void *wpack12, const void *graph_ops, const void *graph_ops2,
                     const float *bias1, const float *bias_mid, const float *bias2, int B, int T, int K, int F, const float *gate1,
                     const float *gate2, float *x_out, int64_t ld2, float *x1, float *y1, float *pool_out, float *overlap_partial,
                     hipStream_t st, unsigned long long *stamps, int flags)
{
    // flags (common.h kBlock8*): ROWMAJOR = a row block's three column slices on one XCD (the product's map: ggcn_block_fused takes
    // this kernel for large batches), else column slices pinned to XCDs; NODMA / SAMESLOTS: lab switches of ggcn_lab_block_fused8
    const char *who = (flags & kBlock8Product) ? "ggcn_block_fused" : "ggcn_lab_block_fused8";
    if (!X || !wpack1 || !wpack12 || !graph_ops || !graph_ops2 || !gate1 || !gate2 || !x1 || !y1 || !bias_mid)
        return fail(GGCN_EINVAL, "%s: null pointer", who);
    if (B <= 0 || T <= 0 || T > 32 || K <= 0 || F <= 0) return fail(GGCN_EUNSUPPORTED, "%s: B=%d T=%d K=%d F=%d (graphs of <= 32 nodes)", who, B, T, K, F);
    const bool avec = (K % 4 == 0) && (ldx % 4 == 0) && aligned16(X) && (int64_t)ldx * 4 * 257 < ((int64_t)1 << 31);
    if (!avec || K % BK != 0) return fail(GGCN_EUNSUPPORTED, "%s: fast shapes only (K %% 32 == 0, 16-byte aligned rows)", who);
    if (!x_out && !pool_out) return fail(GGCN_EINVAL, "%s: neither x nor its pool requested", who);
    FusedArgs a = {};
    a.X = X; a.ldx = ldx; a.graph_ops = static_cast<const char *>(graph_ops); a.graph_ops2 = static_cast<const char *>(graph_ops2);
    const bool rowmajor = (flags & kBlock8RowMajor) != 0;
    a.B = B; a.T = T; a.K = K; a.F = F; a.n_parts = rowmajor ? 3 : 2;
    a.part[0] = LayerPart{static_cast<const char *>(wpack1), bias1, nullptr, nullptr, nullptr, gate1, gate2, nullptr, x1, y1, overlap_partial, 0};
    a.part[1] = LayerPart{static_cast<const char *>(wpack12), bias2, bias_mid, nullptr, gate2, gate2, nullptr, x_out, pool_out, nullptr, nullptr, (int)ld2};
    a.stamps = stamps;
    a.k_steps = round_up(K, BK) / KSTEP;
    // (bit 30 of k_steps: every gate / bias row this launch reads starts 16-byte aligned -- the LDS-DMA staging may be used)
    const bool al = (F % 4 == 0) && aligned16(gate1) && aligned16(gate2) && aligned16(bias1) && aligned16(bias2) && aligned16(bias_mid) &&
                    aligned16(graph_ops) && aligned16(graph_ops2) && !(flags & kBlock8NoDma);
    if (al) a.k_steps |= 0x40000000;
    if (flags & kBlock8SameSlots) a.k_steps |= 0x20000000;
    a.n_wg = (F + BN - 1) / BN;
    a.g_tiles = (B + 3) / 4;
    const bool vst = x_out && (F % 4 == 0) && (ld2 % 4 == 0) && aligned16(x_out) && (int64_t)T * ld2 * 4 < ((int64_t)1 << 31);
    if (x_out && !vst) return fail(GGCN_EUNSUPPORTED, "%s: x needs F %% 4 == 0 and 16-byte aligned rows", who);
    const bool fullt = (T == 32) && (B % 4 == 0);
    const int64_t total = (int64_t)a.g_tiles * a.n_wg;
    const int64_t grid = rowmajor ? grid_for(a.g_tiles, a.n_wg) : 8 * ((total + 7) / 8);
    if (grid > (int64_t)INT32_MAX) return fail(GGCN_EUNSUPPORTED, "%s: batch too large", who);
#define GGCN_L8(FT, VS)                                                                                                           \
    do {                                                                                                                          \
        auto kern = block_fused8_kernel<FT, VS>;                                                                                  \
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, kB8Lds) != hipSuccess) \
            return fail(GGCN_ELAUNCH, "%s: cannot reserve %d bytes of LDS", who, kB8Lds);                                         \
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kB8Threads), kB8Lds, st, a);                                          \
    } while (0)
    if (fullt && vst) GGCN_L8(true, true);
    else if (fullt) GGCN_L8(true, false);
    else if (vst) GGCN_L8(false, true);
    else GGCN_L8(false, false);
#undef GGCN_L8
    return check_launch(who);
}

GGCN_RANGE_FLAG_TU(range_flag_block8)

}  // namespace ggcn
