// Per-element keep masks of the training-mode gates (models/bert_amir5.py:621-625: F.dropout on the gates REPEATED to
// [B,T,H], one Bernoulli draw per token and feature) without a [B,T,H] tensor: a counter-based hash of (seed, element)
// that the forward epilogue, the backward pass and ggcn_dropout_mask all evaluate the same way.
//   h    = mix32(node * F + f + seed_lo) ^ seed_hi-derived constant   (lowbias32 finaliser: two multiplies, three xor-shifts)
//   gate stream 1 keeps the element when (h & 0xFFFF) >= thr, stream 2 when (h >> 16) >= thr,  thr = round(p * 65536)
// The block's two gates use the two streams; gate2's stream serves layer 1's second pool AND layer 2's store gate / pool,
// as the reference's ONE dropped copy of gate2 does (:623,:631,:639).
#pragma once
#include <cstdint>

namespace ggcn {

struct DropSpec {
    uint32_t seed_lo, seed_hi;
    uint32_t thr;      // drop when the stream's 16 bits are below thr (0: nothing is dropped)
    float scale;       // 1 / (1 - p)
    int sel[3];        // stream (0 none, 1, 2) of the store gate, pool gate a, pool gate b
};

__host__ __device__ inline uint32_t drop_hash(uint32_t idx, uint32_t seed_lo, uint32_t seed_hi)
{
    uint32_t x = idx + seed_lo;
    x ^= x >> 16;
    x *= 0x7feb352du;
    x ^= x >> 15;
    x ^= seed_hi;
    x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
// keep factor (0 or scale) of stream `sel` for hash h
__host__ __device__ inline float drop_keep(uint32_t h, int sel, uint32_t thr, float scale)
{
    if (sel == 0) return 1.0f;
    const uint32_t bits = sel == 1 ? (h & 0xFFFFu) : (h >> 16);
    return bits >= thr ? scale : 0.0f;
}
inline DropSpec make_drop_spec(float p, uint64_t seed, int sel_store, int sel_a, int sel_b)
{
    DropSpec d;
    d.seed_lo = (uint32_t)seed;
    d.seed_hi = (uint32_t)(seed >> 32) * 0x9E3779B9u + 0x85EBCA6Bu;
    const double t = (double)p * 65536.0 + 0.5;
    d.thr = p <= 0.0f ? 0u : (t >= 65535.0 ? 65535u : (uint32_t)t);
    d.scale = p <= 0.0f ? 1.0f : 1.0f / (1.0f - p);
    d.sel[0] = sel_store; d.sel[1] = sel_a; d.sel[2] = sel_b;
    return d;
}

}  // namespace ggcn
