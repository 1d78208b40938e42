// Which values does the scale of lane l cover in v_mfma_scale_f32_32x32x64_f8f6f4 with fp6 operands, and does
// v_cvt_scalef32_2xpk16_fp6_f32 saturate above 7.5?  Development tool.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(const unsigned *a, const unsigned *b, const int *sa, const int *sb, float *y)
{
    const int l = threadIdx.x;
    i32x8 va = {}, vb = {};
    for (int i = 0; i < 6; ++i) { va[i] = a[l * 6 + i]; vb[i] = b[l * 6 + i]; }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c, 2, 2, 0, sa[l], 0, sb[l]);
    for (int i = 0; i < 16; ++i) y[l * 16 + i] = c[i];
}
__global__ void cvt(const float *src, unsigned *dst, float scale)
{
    f32x16 a, b;
#pragma unroll
    for (int i = 0; i < 16; ++i) { a[i] = src[i]; b[i] = src[16 + i]; }
    u32x6 r;
    asm volatile("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(scale));
    if (threadIdx.x == 0) for (int i = 0; i < 6; ++i) dst[i] = r[i];
}
static void set_field(unsigned *w, int j, int code)
{
    const int bit = 6 * j, word = bit >> 5, off = bit & 31;
    unsigned long long v = (unsigned long long)code << off;
    w[word] |= (unsigned)v;
    if (word + 1 < 6) w[word + 1] |= (unsigned)(v >> 32);
}
static float val(int code) { int s = code >> 5, e = (code >> 3) & 3, m = code & 7; float v = e == 0 ? m * 0.125f : (1.0f + m / 8.0f) * std::ldexp(1.0f, e - 1); return s ? -v : v; }
int main()
{
    unsigned ha[64 * 6] = {}, hb[64 * 6] = {}, *da, *db; int hs[64], hsb[64], *dsa, *dsb; float *dy, hy[64 * 16];
    hipMalloc(&da, sizeof ha); hipMalloc(&db, sizeof hb); hipMalloc(&dsa, 256); hipMalloc(&dsb, 256); hipMalloc(&dy, sizeof hy);
    // A: every field 1.0 (code 8) except lane 37 (row 5, h = 1) fields 0..15 = 0;  B: all ones
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 32; ++j) { if (!(l == 37 && j < 16)) set_field(&ha[l * 6], j, 8); set_field(&hb[l * 6], j, 8); }
    for (int t = 0; t < 3; ++t) {
        for (int l = 0; l < 64; ++l) { hs[l] = 127; hsb[l] = 127; }
        if (t == 1) hs[5] = 128;
        if (t == 2) hs[37] = 128;
        hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice); hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
        hipMemcpy(dsa, hs, 256, hipMemcpyHostToDevice); hipMemcpy(dsb, hsb, 256, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dy);
        hipMemcpy(hy, dy, sizeof hy, hipMemcpyDeviceToHost);
        // D[5][0]: row 5 = reg 1 (row0 = 1) + 4h with h = 1 -> lane 32, reg 1
        printf("case %d: D[5][0] = %g   D[6][0] = %g\n", t, hy[32 * 16 + 1], hy[32 * 16 + 2]);
    }
    printf("expected: base 48; scale of lane 5 doubled: 64 if it covers fields 0-15 of lanes 5 and 37 (block 0 of row 5), 80 if lane 5's own 32 values;\n"
           "          scale of lane 37 doubled: 80 if block 1 of row 5 (fields 16-31 of both lanes), 64 if lane 37's own values\n");
    float hsrc[32], *dsrc; unsigned *dd, r[6];
    hipMalloc(&dsrc, 128); hipMalloc(&dd, 24);
    const float probe[16] = {7.4f, 7.6f, 7.74f, 7.76f, 8.0f, 100.0f, 1e30f, -9.0f, INFINITY, NAN, 0.0624f, 0.0626f, 0.19f, 1.0625f, 1.1875f, -0.05f};
    for (int i = 0; i < 32; ++i) hsrc[i] = i < 16 ? probe[i] : 0.0f;
    hipMemcpy(dsrc, hsrc, 128, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(cvt, dim3(1), dim3(64), 0, 0, dsrc, dd, 1.0f);
    hipMemcpy(r, dd, 24, hipMemcpyDeviceToHost);
    printf("convert:");
    for (int i = 0; i < 16; ++i) {
        int bit = 6 * (2 * i), w = bit >> 5, off = bit & 31;
        unsigned long long v = r[w]; if (w + 1 < 6) v |= (unsigned long long)r[w + 1] << 32;
        printf("  %g->%g", probe[i], val((int)((v >> off) & 63)));
    }
    printf("\n");
    return 0;
}
