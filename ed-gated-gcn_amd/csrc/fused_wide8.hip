// One launch per gated layer for graphs of 129..256 nodes (ACE cased: ORI_ML = 231, constant.py:267): eight wavefronts per
// (graph, 256 columns); see fused_layer.hip for the scheme (models/gcn.py:34-45 + bert_amir5.py:627-640 in one kernel).
#include "fused_common.h"

namespace ggcn {
namespace {

// plain v_max / v_min (fmaxf first quiets a possible signalling NaN of its operands: an extra instruction per value)
__device__ __forceinline__ float vmaxf_raw(float x, float y) { float d; asm("v_max_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y)); return d; }
__device__ __forceinline__ float vminf_raw(float x, float y) { float d; asm("v_min_f32 %0, %1, %2" : "=v"(d) : "v"(x), "v"(y)); return d; }

// ---- graphs of 129..256 nodes, two wavefronts per SIMD: EIGHT wavefronts per (graph, 256 columns) -----------------------
// Wavefront (rg, cg) = rows 128 rg .. + 127 of the graph's 256-row slot x columns 64 cg .. + 63: the two row groups run
// the SAME main loop side by side on their own stage buffers (thread ids taken mod 256 inside the loop), so the loop keeps
// the two-wavefronts-per-SIMD speed of the 32-node kernel instead of the lone wavefront of the SB = 8 form above, and no
// half of the graph waits in registers.  A neighbour sum needs hidden rows of BOTH row groups, so the accumulators go to
// LDS per 32-column tile -- as an fp32 tile [256 rows][32 columns] per column group, 128 KiB over the dead stage buffers --
// and the sums run over per-row EDGE LISTS (made once per workgroup from the row masks) with 8 lanes x 16 B per row, 8
// source rows in flight: exact fp32 sums, ~5 LDS reads + 20 adds per row of a parse.  160 KiB of LDS, one workgroup per CU.
// Measured (tools/wide_timing.py, f16mx8, 512 x 231 x 768): 399 us against 498 us for linear + aggregate (507 us for the
// lone-wavefront form); 228 us of it is the main loop (timing build without the epilogue), the rest runs under nothing:
// with one workgroup per CU the phases are serial.  The FIRST form of this epilogue (GGCN_LAB_WIDE8_DENSE: bf16 plane
// fragments exchanged through LDS, dense 32 x 32 adjacency blocks on the MFMAs, mask words expanded into operands, empty
// blocks skipped) took 456 us: 8 x 4 MFMAs and ~100 VALU of expansion per block against a handful of edges per row.
constexpr int kW8Threads = 512;
constexpr int kW8Ex = 8 * 16 * 1024;          // per wavefront: 4 row blocks x (2 planes x 2 k-steps) x 1 KiB
constexpr int kW8Cap = 16;                   // source ids per row kept in LDS (rows with more neighbours walk their mask words)
constexpr int kW8Stage = 4096;                // per wavefront: 32 rows x 32 columns of output on their way to 16-byte stores
constexpr int kW8Lds = kW8Ex + 8 * kW8Stage;  // 160 KiB
constexpr int kW8ListBytes = GGCN_EDGE_LISTS_BYTES;   // ids 8192 + degrees 1024 + reciprocals 1024 + zero row 128, padded to 11 KiB
static_assert(4096 + kW8ListBytes <= 8 * kW8Stage, "the lists' LDS image lies behind both row groups' stage buffers");
static_assert(2 * kLdsBytes <= kW8Ex && kW8Lds <= 160 * 1024, "the stage buffers of both row groups lie under the exchange area");

// DROP (training, bert_amir5.py:621-625): per-(token, feature) keep factors of the three gates (dropout_hash.h), the pools
// maximise the gated, kept values (edge-list epilogue only).
template <int SCH, bool AVEC, bool KFULL, bool VST, bool DROP = false>
__global__ __launch_bounds__(kW8Threads, 2) void layer_fused_wide8_kernel(const FusedArgs a)
{
    static_assert(!(DROP && GGCN_LAB_WIDE8_DENSE), "gate dropout lives in the edge-list epilogue");
    extern __shared__ __attribute__((aligned(16))) char lds8[];
    const int B = a.B, T = a.T, K = a.K, F = a.F;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (a.ov_in && blockIdx.x == 0) {   // reduce_partials with the first four wavefronts summing (same order, same result)
        float *red = reinterpret_cast<float *>(lds8);
        const int n_part = B * ((F + 63) / 64);
        float sdot = 0.0f;
        if (tid < kThreads)
            for (int idx = tid; idx < n_part; idx += kThreads) sdot += a.ov_in[idx];
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) sdot += __shfl_xor(sdot, d);
        if (lane == 0 && wave < 4) red[wave] = sdot;
        __syncthreads();
        if (tid == 0) *a.ov_out = ((red[0] + red[1]) + (red[2] + red[3])) / (float)B;
        __syncthreads();
    }
    int g, n_wgi;
    if (!tile_of_block(blockIdx.x, a.g_tiles, a.n_wg, g, n_wgi)) return;   // one graph per workgroup: g_tiles = B
    const LayerPart &lp = a.part[0];
    const float *__restrict__ bias = lp.bias, *__restrict__ store_gate = lp.store_gate;
    const float *__restrict__ pool_gate_a = lp.pool_gate_a, *__restrict__ pool_gate_b = lp.pool_gate_b;
    float *__restrict__ out = lp.out, *__restrict__ pool_a = lp.pool_a, *__restrict__ pool_b = lp.pool_b;
    const int ldo = lp.ldo;
    const int rg = wave >> 2, cg = wave & 3;
    const int n_tiles_total = (F + NT - 1) / NT;
    const int nt0 = n_wgi * (BN / NT) + cg * RN;
    const int W = (T + 31) >> 5;
    static_assert(WM == 1 && RN == 2, "written for 128 x 64 wavefront tiles");

    // ---- the edge lists of the graph's rows (what the epilogue's neighbour sums walk), made BEFORE the main loop: their LDS
    // lies behind both row groups' stage buffers, and the row masks' trip from global memory hides under the loop's first
    // stages instead of standing between the loop and the epilogue ----
    if constexpr (!GGCN_LAB_WIDE8_DENSE) {
      if (a.graph_ops) {
          // ggcn_graph_edge_lists made the lists once per adjacency tensor: the block IS the LDS image (ids, degrees, reciprocals, the
          // zero row), 11 pieces of 1 KiB brought in by LDS-DMA -- no registers, no instructions beyond the 1-2 issues per wavefront;
          // they land under the main loop's first stages (its per-stage vmcnt(0) + barrier cover them long before the epilogue)
          const char *blk = a.graph_ops + (int64_t)g * kW8ListBytes;
          for (int piece = wave; piece < kW8ListBytes / 1024; piece += 8)
              __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(blk + piece * 1024 + lane * 16),
                                               (__attribute__((address_space(3))) void *)(lds8 + kW8Ex + 4096 + piece * 1024), 16, 0, 0);
      } else {
        unsigned short *s_ids = reinterpret_cast<unsigned short *>(lds8 + kW8Ex + 4096);   // [256 rows][kW8Cap]
        int *s_deg = reinterpret_cast<int *>(lds8 + kW8Ex + 4096 + 256 * kW8Cap * 2);      // [256]
        float *s_inv = reinterpret_cast<float *>(lds8 + kW8Ex + 4096 + 256 * kW8Cap * 2 + 1024);   // [256] 1 / (deg + 1)
        const int zero_off = kW8Ex + 4096 + 256 * kW8Cap * 2 + 2048;                         // 128 B of zeros
        auto tile_off = [](int row, int chunk) { return row * 128 + ((chunk ^ (((row >> 1) & 1) << 2)) << 4); };
        // thread t < 256: row t
        if (tid < 256) {
            const int row = tid;
            uint32_t mwd[8];
#pragma unroll
            for (int wi = 0; wi < 8; ++wi) {
                const bool ok = row < T && wi < W && !((GGCN_LAB_OFF) & 32);   // (timing build: no masks, empty lists)
                const uint32_t v = a.rowmask[ok ? ((int64_t)g * T + row) * W + wi : 0];
                mwd[wi] = ok ? v : 0u;
            }
            int deg = 0, e = 0;
#pragma unroll
            for (int wi = 0; wi < 8; ++wi) {
                uint32_t w = mwd[wi];
                deg += __popc(w);
                while (w && e < kW8Cap) {   // stored: the source row's byte offset in a tile (chunk 0; a lane XORs its 16 cl in)
                    s_ids[row * kW8Cap + e++] = (unsigned short)tile_off(32 * wi + __builtin_ctz(w), 0);
                    w &= w - 1;
                }
            }
            // the rest of the list points at row 255: a padding row for T < 256, all zeros in every tile (its X row was staged as
            // zeros), so the sums need no test per slot; T = 256 has no such row and masks the slots instead
            for (; e < kW8Cap; ++e) s_ids[row * kW8Cap + e] = (unsigned short)tile_off(255, 0);
            s_deg[row] = deg;
            s_inv[row] = 1.0f / (float)(deg + 1);                               // gcn.py:35
            if (tid < 32) reinterpret_cast<float *>(lds8 + zero_off)[tid] = 0.0f;
        }
      }
    }

    // ---- hidden = X . W for both row groups at once ----
    constexpr int NP = Geom<float>::NP;
    f32x16 acc[4][RN];
    {
        const float *arow[NP];
        bool avalid[NP];
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int r = stage_row<float>(i) + 128 * rg;
            avalid[i] = r < T;
            arow[i] = a.X + ((int64_t)g * T + (avalid[i] ? r : 0)) * a.ldx;
        }
        char *stage = lds8 + rg * kLdsBytes;
        // 32-row blocks of this row group that hold nodes: T = 160 leaves the second group one block of four -- its other MFMAs
        // are skipped, and the SIMD it shares with a first-group wavefront gets through a stage that much sooner
        const int rows_here = T - 128 * rg;
        const int nblk = rows_here >= 128 ? 4 : rows_here <= 0 ? 0 : (rows_here + 31) >> 5;
        if constexpr (SCH == 0)
            bx3::mainloop<float, AVEC, KFULL, true, true>(arow, avalid, lp.wpack, K, a.k_steps, 0, nt0, n_tiles_total, stage, acc, nblk);
        else {
            constexpr bool BUF = AVEC && KFULL;   // buffer loads (f16mx8_core.h): offsets from the graph's first node
            mx8::BufX<float> bx;
            if constexpr (BUF) {
                int rel[NP];
#pragma unroll
                for (int i = 0; i < NP; ++i) {
                    const int r = stage_row<float>(i) + 128 * rg;
                    rel[i] = avalid[i] ? r : -1;
                }
                bx = mx8::make_bufx<float>(a.X, a.ldx, (int64_t)g * T, (int64_t)B * T, rel, tid & (kThreads - 1));
            }
            // The second row group of a graph of 129..224 nodes holds 1-3 blocks of real rows: its own instantiation of the
            // loop without the others' staging passes, fragment reads and MFMAs (the two groups meet at the same barriers
            // from different code; a run-time count could only skip the MFMAs -- guarding the staging passes broke the
            // issue order the loop lives on).  The SIMD such a wavefront shares with a first-group wavefront is mostly the
            // latter's: 512 x 129 x 768 ... (tools/wide_timing.py)
            if (rg == 1 && nblk == 1)
                mx8::mainloop<float, AVEC, KFULL, true, false, BUF, 1>(arow, avalid, lp.wpack, K, a.k_steps / 2, 0, nt0, n_tiles_total, stage, acc, 0, 4, nullptr, &bx);
            else if (rg == 1 && nblk == 2)
                mx8::mainloop<float, AVEC, KFULL, true, false, BUF, 2>(arow, avalid, lp.wpack, K, a.k_steps / 2, 0, nt0, n_tiles_total, stage, acc, 0, 4, nullptr, &bx);
            else if (rg == 1 && nblk == 3)
                mx8::mainloop<float, AVEC, KFULL, true, false, BUF, 3>(arow, avalid, lp.wpack, K, a.k_steps / 2, 0, nt0, n_tiles_total, stage, acc, 0, 4, nullptr, &bx);
            else
                mx8::mainloop<float, AVEC, KFULL, true, false, BUF, 4>(arow, avalid, lp.wpack, K, a.k_steps / 2, 0, nt0, n_tiles_total, stage, acc, 0, 4, nullptr, &bx);
        }
    }
    if constexpr (GGCN_LAB_WIDE8_DENSE) {
    // every accumulator tile -> its two bf16 planes (B-operand fragments of the aggregation MFMAs), in place
    bf16x8 hf[4][RN][2][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j) split2(acc[i][j], hf[i][j]);

    const int c = lane & 31, h = lane >> 5;
    float vb[RN], vsg[RN], vga[RN], vgb[RN];
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
            const int64_t at = (int64_t)g * F + gnc;
            vb[j] = bias ? pb[gnc] : 0.0f;
            vsg[j] = store_gate ? psg[at] : 1.0f;
            vga[j] = pool_gate_a ? pga[at] : 1.0f;
            vgb[j] = pool_gate_b ? pgb[at] : 1.0f;
        }
    }
    // this lane's adjacency rows of output block io (node 32 io + c), one block ahead of their use; words past the graph
    // and rows past T read as zeros
    auto load_masks = [&](int io, uint32_t (&m)[8]) {
        const int node = 32 * io + c;
        const bool ok = node < T;
#pragma unroll
        for (int ii = 0; ii < 8; ++ii) {
            const bool okw = ok && ii < W;
            const uint32_t v = a.rowmask[okw ? ((int64_t)g * T + node) * W + ii : 0];
            m[ii] = okw ? v : 0u;
        }
    };
    uint32_t mw[2][8];
    load_masks(4 * rg, mw[0]);
    float *stage_lds = reinterpret_cast<float *>(lds8 + kW8Ex + wave * kW8Stage);
    const int perm_base = 16 * h;
    const int lane_off = 4 * h * ldo + c;
    float vmax[RN], vmin[RN];
#pragma unroll
    for (int j = 0; j < RN; ++j) { vmax[j] = -INFINITY; vmin[j] = INFINITY; }

    auto tiles = [&](auto has_out) {
        constexpr bool vst = VST && decltype(has_out)::value;
        constexpr bool direct_store = !VST && decltype(has_out)::value;
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            __syncthreads();   // the fragments of column tile j - 1 (j = 0: the last stage's operand planes) have been read
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks)
                        *reinterpret_cast<bf16x8 *>(lds8 + (((wave * 4 + i) * 2 + p) * 2 + ks) * 1024 + lane * 16) = hf[i][j][p][ks];
            __syncthreads();
            const bool tile_ok = nt0 + j < n_tiles_total;   // wavefront-uniform: column tile past F
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int t = 4 * j + i, io = 4 * rg + i, node0 = 32 * io;
                // the next block's masks (wraps to this wavefront's first block for the second column tile)
                if (t + 1 < 4 * RN) load_masks(4 * rg + ((i + 1) & 3), mw[(t + 1) & 1]);
                if (node0 >= T || !tile_ok) continue;   // wavefront-uniform: a block of padding rows (no barrier below)
                const uint32_t (&m)[8] = mw[t & 1];
                int deg = 0;
#pragma unroll
                for (int ii = 0; ii < 8; ++ii) deg += __popc(m[ii]);
                const float inv = 1.0f / (float)(deg + 1);                  // gcn.py:35
                f32x16 y;
#pragma unroll
                for (int r = 0; r < 16; ++r) y[r] = 0.0f;
                // The four operand fragments of block ii + 1 are asked for before block ii's MFMAs (an empty block's are read in
                // vain): read just in front of their use, every pair of MFMAs waited out an LDS round trip.
                auto frag_src = [&](int ii) { return lds8 + ((((ii >> 2) * 4 + cg) * 4 + (ii & 3)) * 4) * 1024 + lane * 16; };
                bf16x8 fr[2][4];
#pragma unroll
                for (int q = 0; q < 4; ++q) fr[0][q] = *reinterpret_cast<const bf16x8 *>(frag_src(0) + q * 1024);
#pragma unroll
                for (int ii = 0; ii < 8; ++ii) {
                    if (32 * ii >= T) break;   // workgroup-uniform: source blocks of padding rows
                    if (ii + 1 < 8) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) fr[(ii + 1) & 1][q] = *reinterpret_cast<const bf16x8 *>(frag_src(ii + 1) + q * 1024);
                    }
                    // a block without an edge adds nothing (dependency arcs are mostly short: away from the diagonal most
                    // blocks of a parse are empty); wavefront-uniform
                    if (__builtin_amdgcn_ballot_w64(m[ii] != 0u) == 0) continue;
                    bf16x8 af[2];
                    if constexpr (((GGCN_LAB_OFF) & 32) != 0) {   // (timing build: no expansion)
                        union { bf16x8 v; uint32_t w[4]; } u;
                        u.w[0] = u.w[1] = u.w[2] = u.w[3] = m[ii] & 0x3F803F80u;
                        af[0] = af[1] = u.v;
                    } else
                    expand_mask(m[ii] >> (4 * h), af);   // once per block: both planes use it (small plane first)
#pragma unroll
                    for (int p = 1; p >= 0; --p)
#pragma unroll
                        for (int ks = 0; ks < 2; ++ks)
                            if constexpr (!((GGCN_LAB_OFF) & 64))
                                y = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks], fr[ii & 1][2 * p + ks], y, 0, 0, 0);   // gcn.py:41
                }
                float rinv[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row0 = (r & 3) + 8 * (r >> 2);
                    rinv[r] = __int_as_float(__builtin_amdgcn_ds_bpermute(perm_base + 4 * row0, __float_as_int(inv)));
                }
                float *tile = decltype(has_out)::value ? out + ((int64_t)g * T + node0) * ldo + (nt0 + j) * NT : nullptr;
                auto finish = [&](auto whole_c) {   // whole: all 32 rows of the block are nodes (wavefront-uniform) -- no row test
                    constexpr bool WHOLE = decltype(whole_c)::value;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int row0 = (r & 3) + 8 * (r >> 2);  // this lane's row is row0 + 4h
                        const float v = y[r] * rinv[r] + vb[j];   // gcn.py:41,43
                        if (vst) stage_lds[(row0 + 4 * h) * 32 + c] = v * vsg[j];
                        if (WHOLE || node0 + row0 + 4 * h < T) {
                            if (direct_store && col_ok[j]) tile[lane_off + row0 * ldo] = v * vsg[j];
                            vmax[j] = vmaxf_raw(vmax[j], v);
                            vmin[j] = vminf_raw(vmin[j], v);
                        }
                    }
                };
                if constexpr (((GGCN_LAB_OFF) & 16) != 0) {   // (timing build: nothing behind the neighbour sums)
                    asm volatile("" :: "v"(y[0]), "v"(y[5]), "v"(y[10]), "v"(y[15]), "v"(rinv[3]));
                    continue;
                }
                if (node0 + 32 <= T) finish(std::true_type{});
                else finish(std::false_type{});
                if (vst) {   // 32 rows x 128 B leave as 16 B per lane: 8 rows per instruction
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    const int colq = (lane & 7) * 4;
                    const int gcol = (nt0 + j) * NT + colq;
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int row = 8 * it + (lane >> 3);
                        const float4 v4 = *reinterpret_cast<const float4 *>(&stage_lds[row * 32 + colq]);
                        if (node0 + row < T && gcol < F) store_out4_stream(tile + row * ldo + colq, v4);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
    };
    if constexpr (!((GGCN_LAB_OFF) & 128)) {
        if (out) tiles(std::true_type{});
        else tiles(std::false_type{});
    } else {
        asm volatile("" :: "v"(hf[0][0][0][0]), "v"(hf[3][1][1][1]), "v"(hf[1][0][1][0]), "v"(hf[2][1][0][1]));
    }

    // pools of the graph: max over ALL its rows (bert_amir5.py:635-640) -- the two row groups meet in LDS
    if (pool_a || pool_b || lp.ov_partial) {
        float *pl = reinterpret_cast<float *>(lds8 + kW8Ex);   // [wavefront][max / min][column tile][32] over the store staging
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RN; ++j) {
            const float mx = fmaxf(vmax[j], upper_half_to_lower(vmax[j]));
            const float mn = fminf(vmin[j], upper_half_to_lower(vmin[j]));
            if (h == 0) {
                pl[((wave * 2 + 0) * RN + j) * 32 + c] = mx;
                pl[((wave * 2 + 1) * RN + j) * 32 + c] = mn;
            }
        }
        __syncthreads();
        if (rg == 0) {
            float dot = 0.0f;
#pragma unroll
            for (int j = 0; j < RN; ++j) {
                if (nt0 + j >= n_tiles_total) break;
                const float mx = fmaxf(pl[((wave * 2 + 0) * RN + j) * 32 + c], pl[(((wave + 4) * 2 + 0) * RN + j) * 32 + c]);
                const float mn = fminf(pl[((wave * 2 + 1) * RN + j) * 32 + c], pl[(((wave + 4) * 2 + 1) * RN + j) * 32 + c]);
                if (h == 0 && col_ok[j]) {
                    const int gn = (nt0 + j) * NT + c;
                    const float pa = vga[j] * (vga[j] >= 0.0f ? mx : mn), pb = vgb[j] * (vgb[j] >= 0.0f ? mx : mn);
                    if (pool_a) pool_a[(int64_t)g * F + gn] = pa;
                    if (pool_b) pool_b[(int64_t)g * F + gn] = pb;
                    dot = fmaf(pa, pb, dot);
                }
            }
            if (lp.ov_partial && nt0 < n_tiles_total) {
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) dot += __shfl_xor(dot, d);
                if (lane == 0) lp.ov_partial[(int64_t)g * ((F + 63) / 64) + (nt0 >> 1)] = dot;
            }
        }
    }

    } else {
    // ---- neighbour sums over the EDGES, out of an fp32 tile in LDS (gcn.py:41) ----------------------------------------
    // Per 32-column tile j every wavefront writes its accumulators, as they are, into its column group's tile
    // [256 rows][32 columns] fp32 (4 x 32 KiB over the dead stage buffers).  Then 8 lanes x 16 B cover a row: a wavefront
    // sums 8 destination rows at once, each 8-lane group reading its row's edge list (up to kW8Cap source ids, made once
    // per workgroup from the row masks: ids and degrees in LDS) and then the source rows' 16-byte pieces, 8 in flight --
    // two dependent LDS round trips per 8 edges, ~5 reads and 20 adds per row of a parse, instead of 8 blocks x 4 MFMAs and
    // the expansion of 8 mask words into MFMA operands.  The sums are exact fp32; rows leave straight from registers as
    // 16-byte stores.  (A first form that walked the mask bits one LDS read at a time took 630 us where the MFMA form
    // took 470: every edge waited out its own round trip.)
    const int q8 = lane >> 3, cl = lane & 7;
    const int c = lane & 31, h = lane >> 5;
    char *tile_cg = lds8 + cg * (256 * 128);
    unsigned short *s_ids = reinterpret_cast<unsigned short *>(lds8 + kW8Ex + 4096);   // [256 rows][kW8Cap]
    int *s_deg = reinterpret_cast<int *>(lds8 + kW8Ex + 4096 + 256 * kW8Cap * 2);      // [256]
    float *s_inv = reinterpret_cast<float *>(lds8 + kW8Ex + 4096 + 256 * kW8Cap * 2 + 1024);   // [256] 1 / (deg + 1)
    const int zero_off = kW8Ex + 4096 + 256 * kW8Cap * 2 + 2048;                         // 128 B of zeros
    // 16-byte chunk `chunk` of row `row`: the 64-byte half is flipped on rows 2, 3 (mod 4), so that the four 8-lane groups a
    // ds_read_b128 serves together (two read chunks 0-3, two chunks 4-7 of their rows) collide on one row pair in four
    auto tile_off = [](int row, int chunk) { return row * 128 + ((chunk ^ (((row >> 1) & 1) << 2)) << 4); };
    // (the edge lists of the graph's rows were made before the main loop: see there)
    float vmax[RN][4], vmin[RN][4];
#pragma unroll
    for (int j = 0; j < RN; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) { vmax[j][k] = -INFINITY; vmin[j][k] = DROP ? -INFINITY : INFINITY; }
    // bias and store gate of this lane's columns for both column tiles: asked for here, used behind two barriers
    const float *dummy = a.X;
    float b4[RN][4], sg4[RN][4], ga4[RN][4], gb4[RN][4];   // (the pool gates per lane column: DROP only)
    bool cok[RN][4];
#pragma unroll
    for (int j = 0; j < RN; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int col = (nt0 + j) * NT + 4 * cl + k;
            cok[j][k] = col < F;
            const int cc = cok[j][k] ? col : 0;
            b4[j][k] = bias ? (bias ? bias : dummy)[cc] : 0.0f;
            sg4[j][k] = store_gate ? (store_gate ? store_gate : dummy)[(int64_t)g * F + cc] : 1.0f;
            if constexpr (DROP) {
                ga4[j][k] = pool_gate_a ? (pool_gate_a ? pool_gate_a : dummy)[(int64_t)g * F + cc] : 1.0f;
                gb4[j][k] = pool_gate_b ? (pool_gate_b ? pool_gate_b : dummy)[(int64_t)g * F + cc] : 1.0f;
            }
        }
    // `pre` (ggcn_layer_fused_prebias): hidden + 1.pre^T goes to the tile -- for the graph's real rows only: the padding rows
    // (row 255 is the edge lists' filler) stay zero
    float pre_c[RN];
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        const int gn = (nt0 + j) * NT + c;
        pre_c[j] = lp.pre ? lp.pre[gn < F ? gn : 0] : 0.0f;
    }
    const int tile_lane = cg * (256 * 128) + 16 * cl, zero_lane = zero_off + 16 * cl;
    const int n_steps = ((GGCN_LAB_OFF) & 16) ? 1 : 16;   // (timing build: one row step only)
    const bool full_slot = T == 256;
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        __syncthreads();   // the tile of column tile j - 1 (j = 0: the last stage's operand planes) has been read
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 128 * rg + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * h;
                *reinterpret_cast<float *>(tile_cg + tile_off(row, c >> 2) + (c & 3) * 4) = acc[i][j][r] + (row < T ? pre_c[j] : 0.0f);
            }
        __syncthreads();   // (also: the edge lists are complete)
        if (nt0 + j >= n_tiles_total) continue;   // wavefront-uniform: column tile past F (the barriers above are met)
        const int col0 = (nt0 + j) * NT + 4 * cl;   // this lane's four columns
        // degree, reciprocal and the first 8 list entries of a row step are read one step ahead: a step is then ONE
        // dependent LDS round trip (the source rows) instead of two
        // The two wavefronts of a column group take the row steps (8 rows each) ALTERNATELY -- step 2 it + rg -- whatever row
        // group their accumulators came from: a 129-node graph is 17 steps, 9 and 8 of them instead of 16 and 1 (the tile in
        // LDS holds every row; the pools meet in LDS anyway).
        // (two instantiations of the row steps: a full 256-node slot has no padding row and tests every list slot against
        // the degree; decided per workgroup -- as a run-time flag it cost a compare, a scalar OR and a select per edge)
        auto row_steps = [&](auto full_c) {
        constexpr bool FULL_SLOT = decltype(full_c)::value;
        typedef float f32x4v __attribute__((ext_vector_type(4)));
        typedef __attribute__((address_space(3))) const f32x4v *lds_f4;
        int deg_n = s_deg[8 * rg + q8];
        float inv_n = s_inv[8 * rg + q8];
        uint4 idq_n = *reinterpret_cast<const uint4 *>(s_ids + (8 * rg + q8) * kW8Cap);
        for (int it = 0; it < n_steps; ++it) {
            const int step = 2 * it + rg;
            if (8 * step >= T) break;   // wavefront-uniform: only padding rows from here on
            const int row = 8 * step + q8;
            const int deg = deg_n;
            const float inv = inv_n;
            const uint4 idq0 = idq_n;
            if (it + 1 < 16) {   // (the lists cover all 256 row slots)
                deg_n = s_deg[row + 16];
                inv_n = s_inv[row + 16];
                idq_n = *reinterpret_cast<const uint4 *>(s_ids + (row + 16) * kW8Cap);
            }
            float s4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
            auto pass = [&](const uint4 &idq, int e0) {
                const uint32_t idw[4] = {idq.x, idq.y, idq.z, idq.w};
                float4 v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int base = (int)((idw[e >> 1] >> (16 * (e & 1))) & 0xFFFFu);
                    int off = tile_lane ^ base;   // (base has no bits below 64; tile_lane = cg base + 16 cl)
                    if constexpr (FULL_SLOT) off = e0 + e < deg ? off : zero_lane;
                    const f32x4v t4 = *(lds_f4)(size_t)off;    // (the dynamic LDS block starts at byte 0)
                    v[e] = make_float4(t4[0], t4[1], t4[2], t4[3]);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) { s4[0] += v[e].x; s4[1] += v[e].y; s4[2] += v[e].z; s4[3] += v[e].w; }
            };
            if (deg <= kW8Cap) {
                pass(idq0, 0);   // (a row without a neighbour adds eight zeros)
                if (deg > 8) pass(*reinterpret_cast<const uint4 *>(s_ids + row * kW8Cap + 8), 8);   // (divergent per 8-lane group)
            } else {   // more neighbours than a list holds: walk the mask words themselves (rare, slow, same sums in another order)
                for (int wi = 0; wi < W; ++wi) {
                    uint32_t w = a.rowmask[((int64_t)g * T + row) * W + wi];
                    while (w) {
                        const int src = 32 * wi + __builtin_ctz(w);
                        w &= w - 1;
                        const float4 v = *reinterpret_cast<const float4 *>(tile_cg + tile_off(src, cl));
                        s4[0] += v.x; s4[1] += v.y; s4[2] += v.z; s4[3] += v.w;
                    }
                }
            }
            if (row < T) {
                float o4[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float v = s4[k] * inv + b4[j][k];                 // gcn.py:41,43
                    o4[k] = v * sg4[j][k];
                    if constexpr (DROP) {   // vmax / vmin carry the two pools' running maxima of the gated, kept values
                        const uint32_t hh = drop_hash((uint32_t)(((int64_t)g * T + row) * F + col0 + k), a.drop.seed_lo, a.drop.seed_hi);
                        o4[k] *= drop_keep(hh, a.drop.sel[0], a.drop.thr, a.drop.scale);
                        vmax[j][k] = vmaxf_raw(vmax[j][k], v * ga4[j][k] * drop_keep(hh, a.drop.sel[1], a.drop.thr, a.drop.scale));
                        vmin[j][k] = vmaxf_raw(vmin[j][k], v * gb4[j][k] * drop_keep(hh, a.drop.sel[2], a.drop.thr, a.drop.scale));
                    } else {
                        vmax[j][k] = vmaxf_raw(vmax[j][k], v);
                        vmin[j][k] = vminf_raw(vmin[j][k], v);
                    }
                }
                if (out && !((GGCN_LAB_OFF) & 8)) {   // (timing build: no stores)
                    float *dst = out + ((int64_t)g * T + row) * ldo + col0;
                    if constexpr (VST) {
                        if (cok[j][0]) store_out4_stream(dst, make_float4(o4[0], o4[1], o4[2], o4[3]));
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (cok[j][k]) dst[k] = o4[k];
                    }
                }
            }
        }
        };   // row_steps
        if (full_slot) row_steps(std::true_type{});
        else row_steps(std::false_type{});
    }
    // pools of the graph: max over ALL its rows (bert_amir5.py:635-640): across the 8 row classes of the wavefront (lanes 8
    // apart), then the two row groups meet in LDS
    if ((pool_a || pool_b || lp.ov_partial) && !((GGCN_LAB_OFF) & 64)) {   // (timing build: no pools)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int d = 8; d <= 32; d <<= 1) {
                    vmax[j][k] = vmaxf_raw(vmax[j][k], __shfl_xor(vmax[j][k], d));
                    vmin[j][k] = DROP ? vmaxf_raw(vmin[j][k], __shfl_xor(vmin[j][k], d)) : vminf_raw(vmin[j][k], __shfl_xor(vmin[j][k], d));
                }
        float *pl = reinterpret_cast<float *>(lds8 + kW8Ex);   // [wavefront][max / min][column tile][32]
        __syncthreads();
        if (q8 == 0) {
#pragma unroll
            for (int j = 0; j < RN; ++j)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    pl[((wave * 2 + 0) * RN + j) * 32 + 4 * cl + k] = vmax[j][k];
                    pl[((wave * 2 + 1) * RN + j) * 32 + 4 * cl + k] = vmin[j][k];
                }
        }
        __syncthreads();
        if (rg == 0) {
            float dot = 0.0f;
            if (lane < 32) {
#pragma unroll
                for (int j = 0; j < RN; ++j) {
                    const int gn = (nt0 + j) * NT + lane;
                    if (nt0 + j < n_tiles_total && gn < F) {
                        const float mx = fmaxf(pl[((wave * 2 + 0) * RN + j) * 32 + lane], pl[(((wave + 4) * 2 + 0) * RN + j) * 32 + lane]);
                        const float m1 = pl[((wave * 2 + 1) * RN + j) * 32 + lane], m2 = pl[(((wave + 4) * 2 + 1) * RN + j) * 32 + lane];
                        const float mn = DROP ? fmaxf(m1, m2) : fminf(m1, m2);
                        const float ga = pool_gate_a ? pool_gate_a[(int64_t)g * F + gn] : 1.0f;
                        const float gb = pool_gate_b ? pool_gate_b[(int64_t)g * F + gn] : 1.0f;
                        const float pa = DROP ? mx : ga * (ga >= 0.0f ? mx : mn), pb = DROP ? mn : gb * (gb >= 0.0f ? mx : mn);
                        if (pool_a) pool_a[(int64_t)g * F + gn] = pa;
                        if (pool_b) pool_b[(int64_t)g * F + gn] = pb;
                        dot = fmaf(pa, pb, dot);
                    }
                }
            }
            if (lp.ov_partial && nt0 < n_tiles_total) {   // fixed butterfly order; lanes 32-63 hold 0
#pragma unroll
                for (int d = 32; d >= 1; d >>= 1) dot += __shfl_xor(dot, d);
                if (lane == 0) lp.ov_partial[(int64_t)g * ((F + 63) / 64) + (nt0 >> 1)] = dot;
            }
        }
    }
    }
}


// ggcn_graph_edge_lists: per graph of 129..256 nodes the LDS image the eight-wavefront kernel's neighbour sums read -- per row up to
// kW8Cap source rows as byte offsets into an fp32 tile [256][32] (its chunk swizzle included), the degree and 1 / (deg + 1)
// (gcn.py:35), a zero row -- made ONCE per adjacency tensor instead of by every (graph, 256 columns) workgroup of every launch.
__global__ __launch_bounds__(256) void graph_edge_lists_kernel(const uint32_t *__restrict__ rowmask, int T, char *__restrict__ lists)
{
    const int g = blockIdx.x, row = threadIdx.x;
    const int W = (T + 31) >> 5;
    char *blk = lists + (int64_t)g * kW8ListBytes;
    unsigned short *ids = reinterpret_cast<unsigned short *>(blk);
    int *deg_o = reinterpret_cast<int *>(blk + 256 * kW8Cap * 2);
    float *inv_o = reinterpret_cast<float *>(blk + 256 * kW8Cap * 2 + 1024);
    auto tile_off = [](int r, int chunk) { return r * 128 + ((chunk ^ (((r >> 1) & 1) << 2)) << 4); };
    int deg = 0, e = 0;
    for (int wi = 0; wi < 8; ++wi) {
        uint32_t w = (row < T && wi < W) ? rowmask[((int64_t)g * T + row) * W + wi] : 0u;
        deg += __popc(w);
        while (w && e < kW8Cap) {
            ids[row * kW8Cap + e++] = (unsigned short)tile_off(32 * wi + __builtin_ctz(w), 0);
            w &= w - 1;
        }
    }
    for (; e < kW8Cap; ++e) ids[row * kW8Cap + e] = (unsigned short)tile_off(255, 0);
    deg_o[row] = deg;
    inv_o[row] = 1.0f / (float)(deg + 1);
    // the zero row and the padding behind it
    for (int i = row; i < (kW8ListBytes - (256 * kW8Cap * 2 + 2048)) / 4; i += 256) reinterpret_cast<float *>(blk + 256 * kW8Cap * 2 + 2048)[i] = 0.0f;
}

}  // namespace

int graph_edge_lists(const uint32_t *rowmask, int B, int T, void *lists, hipStream_t st)
{
    if (!rowmask || !lists) return fail(GGCN_EINVAL, "ggcn_graph_edge_lists: null pointer");
    if (B <= 0 || T <= 128 || T > 256) return fail(GGCN_EUNSUPPORTED, "ggcn_graph_edge_lists: B=%d T=%d (graphs of 129..256 nodes: the eight-wavefront layer)", B, T);
    if (!aligned16(lists)) return fail(GGCN_EINVAL, "ggcn_graph_edge_lists: lists must be 16-byte aligned");
    hipLaunchKernelGGL(graph_edge_lists_kernel, dim3((unsigned)B), dim3(256), 0, st, rowmask, T, static_cast<char *>(lists));
    return check_launch("ggcn_graph_edge_lists");
}

namespace {
}  // namespace

int launch_fused_wide8(const char *who, const FusedArgs &a, int precision, bool fast, bool vst, int64_t gridw, hipStream_t st)
{
#define GGCN_LAUNCH8D(SC, AV, KF, VS, DR)                                                                                 \
    do {                                                                                                                  \
        auto kern = layer_fused_wide8_kernel<SC, AV, KF, VS, DR>;                                                           \
        if (hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,         \
                                kW8Lds) != hipSuccess)                                                                    \
            return fail(GGCN_ELAUNCH, "%s: cannot reserve %d bytes of LDS", who, kW8Lds);                                 \
        hipLaunchKernelGGL(kern, dim3((unsigned)gridw), dim3(kW8Threads), kW8Lds, st, a);                                 \
    } while (0)
#define GGCN_LAUNCH8(SC, AV, KF, VS)                                            \
    do {                                                                        \
        if (a.drop.thr != 0 && !GGCN_LAB_WIDE8_DENSE) GGCN_LAUNCH8D(SC, AV, KF, VS, !GGCN_LAB_WIDE8_DENSE); \
        else GGCN_LAUNCH8D(SC, AV, KF, VS, false);                              \
    } while (0)
#define GGCN_PICK8(SC)                                      \
    do {                                                    \
        if (fast && vst) GGCN_LAUNCH8(SC, true, true, true);        \
        else if (fast) GGCN_LAUNCH8(SC, true, true, false);         \
        else GGCN_LAUNCH8(SC, false, false, false);                 \
    } while (0)
    if (precision == GGCN_PREC_F16MX8) GGCN_PICK8(1);
    else GGCN_PICK8(0);
#undef GGCN_PICK8
#undef GGCN_LAUNCH8
#undef GGCN_LAUNCH8D
    return check_launch(who);
}

GGCN_RANGE_FLAG_TU(range_flag_wide8)

}  // namespace ggcn
