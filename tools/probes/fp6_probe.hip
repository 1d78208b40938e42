// fp6 (e2m3) operand probe (development tool, standalone): before the f16mx6 main loop is written,
//  1. where v_cvt_scalef32_2xpk16_fp6_f32 / v_cvt_scalef32_pk32_fp6_f16 put their 32 results and what the scale does;
//  2. which k a lane's 32 fp6 values of v_mfma_scale_f32_32x32x64_f8f6f4 (cbsz = blgp = 2) stand for, and which
//     lane's scale byte belongs to which (row, 32-k block).
//   hipcc --offload-arch=gfx950 -O3 -o fp6_probe fp6_probe.hip && ./fp6_probe
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));

__global__ void cvt_f32(const float *src, const float *scale, unsigned *dst, int ovfl)
{
    if (ovfl) __builtin_amdgcn_s_setreg(1 | (23 << 6), 1);
    f32x16 a, b;
    for (int i = 0; i < 16; ++i) { a[i] = src[threadIdx.x * 32 + i]; b[i] = src[threadIdx.x * 32 + 16 + i]; }
    const u32x6 r = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(a, b, scale[threadIdx.x]);
    for (int i = 0; i < 6; ++i) dst[threadIdx.x * 6 + i] = r[i];
}
__global__ void cvt_f16(const float *src, const float *scale, unsigned *dst)
{
    f16x32 a;
    for (int i = 0; i < 32; ++i) a[i] = (_Float16)src[threadIdx.x * 32 + i];
    const u32x6 r = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(a, scale[threadIdx.x]);
    for (int i = 0; i < 6; ++i) dst[threadIdx.x * 6 + i] = r[i];
}
__global__ void mfma6(const unsigned *a, const unsigned *b, const int *sa, const int *sb, float *y)
{
    const int l = threadIdx.x;
    i32x8 va = {}, vb = {};
    for (int i = 0; i < 6; ++i) { va[i] = a[l * 6 + i]; vb[i] = b[l * 6 + i]; }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(va, vb, c, 2, 2, 0, sa[l], 0, sb[l]);
    for (int i = 0; i < 16; ++i) y[l * 16 + i] = c[i];
}

static float e2m3_value(int code)   // 6 bits: s e e m m m
{
    const int s = code >> 5, e = (code >> 3) & 3, m = code & 7;
    const float v = e == 0 ? m * 0.125f : (1.0f + m / 8.0f) * std::ldexp(1.0f, e - 1);
    return s ? -v : v;
}
static int field(const unsigned *w, int j) // j-th 6-bit field of a 192-bit little-endian string
{
    const int bit = 6 * j, word = bit >> 5, off = bit & 31;
    unsigned long long v = w[word];
    if (word + 1 < 6) v |= (unsigned long long)w[word + 1] << 32;
    return (int)((v >> off) & 63);
}
static void set_field(unsigned *w, int j, int code)
{
    const int bit = 6 * j, word = bit >> 5, off = bit & 31;
    unsigned long long v = (unsigned long long)code << off;
    w[word] |= (unsigned)v;
    if (word + 1 < 6) w[word + 1] |= (unsigned)(v >> 32);
}

int main()
{
    float *d_src, *d_scale, *d_y;
    unsigned *d_dst, *d_a, *d_b;
    int *d_sa, *d_sb;
    hipMalloc(&d_src, 64 * 32 * 4);
    hipMalloc(&d_scale, 64 * 4);
    hipMalloc(&d_dst, 64 * 6 * 4);
    hipMalloc(&d_a, 64 * 6 * 4);
    hipMalloc(&d_b, 64 * 6 * 4);
    hipMalloc(&d_sa, 64 * 4);
    hipMalloc(&d_sb, 64 * 4);
    hipMalloc(&d_y, 64 * 16 * 4);
    std::vector<float> src(64 * 32), scale(64, 1.0f);
    std::vector<unsigned> dst(64 * 6);

    // ---- 1. conversion: source i holds the value of code i (codes 0..31 = the non-negative e2m3 values) ----
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 32; ++i) src[l * 32 + i] = e2m3_value(i);
    for (int pass = 0; pass < 3; ++pass) {
        std::vector<float> s2 = src;
        float sc = 1.0f;
        if (pass == 1) { sc = 4.0f; for (auto &v : s2) v *= 4.0f; }       // scale divides?
        if (pass == 2) { sc = 0.25f; for (auto &v : s2) v *= 4.0f; }      // or multiplies?
        for (auto &v : scale) v = sc;
        hipMemcpy(d_src, s2.data(), s2.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(d_scale, scale.data(), 64 * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(cvt_f32, dim3(1), dim3(64), 0, 0, d_src, d_scale, d_dst, 0);
        hipMemcpy(dst.data(), d_dst, dst.size() * 4, hipMemcpyDeviceToHost);
        printf("2xpk16_fp6_f32, inputs x%g, scale %g: field j -> code:", pass ? 4.0 : 1.0, sc);
        for (int j = 0; j < 32; ++j) printf(" %d", field(&dst[5 * 6], j));
        printf("\n");
    }
    {
        for (auto &v : scale) v = 1.0f;
        hipMemcpy(d_src, src.data(), src.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(d_scale, scale.data(), 64 * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(cvt_f16, dim3(1), dim3(64), 0, 0, d_src, d_scale, d_dst);
        hipMemcpy(dst.data(), d_dst, dst.size() * 4, hipMemcpyDeviceToHost);
        printf("pk32_fp6_f16, scale 1:             field j -> code:");
        for (int j = 0; j < 32; ++j) printf(" %d", field(&dst[5 * 6], j));
        printf("\n");
    }
    // rounding / saturation: a few values, with and without MODE.FP16_OVFL
    for (int ov = 0; ov < 2; ++ov) {
        const float probe[16] = {0.0624f, 0.0626f, 0.1875f, 0.3125f, 1.0625f, 1.1875f, 7.4f, 7.6f, 7.75f, 8.0f, 100.0f, 1e30f, -9.0f, -0.05f,
                                 INFINITY, NAN};
        std::vector<float> s2(64 * 32, 0.0f);
        for (int i = 0; i < 16; ++i) s2[i] = probe[i];
        hipMemcpy(d_src, s2.data(), s2.size() * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(cvt_f32, dim3(1), dim3(64), 0, 0, d_src, d_scale, d_dst, ov);
        hipMemcpy(dst.data(), d_dst, dst.size() * 4, hipMemcpyDeviceToHost);
        printf("FP16_OVFL=%d:", ov);
        for (int i = 0; i < 16; ++i) printf("  %g->%g", probe[i], e2m3_value(field(&dst[0], i)));
        printf("\n");
    }

    // ---- 2. MFMA operand layout: exact small integers ----
    const int int_code[8] = {0, 8, 16, 20, 24, 26, 28, 30};   // 0,1,2,3,4,5,6,7
    std::vector<int> A(32 * 64), B(64 * 32);
    srand(3);
    for (auto &v : A) v = rand() % 4;
    for (auto &v : B) v = rand() % 4;
    std::vector<double> ref(32 * 32, 0.0);
    for (int r = 0; r < 32; ++r)
        for (int c = 0; c < 32; ++c)
            for (int k = 0; k < 64; ++k) ref[r * 32 + c] += A[r * 64 + k] * B[k * 32 + c];
    auto run = [&](auto kmap, const std::vector<int> &sa, const std::vector<int> &sb, std::vector<double> &D) {
        std::vector<unsigned> a(64 * 6, 0), b(64 * 6, 0);
        for (int l = 0; l < 64; ++l) {
            const int r = l & 31, h = l >> 5;
            for (int j = 0; j < 32; ++j) {
                set_field(&a[l * 6], j, int_code[A[r * 64 + kmap(h, j)]]);
                set_field(&b[l * 6], j, int_code[B[kmap(h, j) * 32 + r]]);
            }
        }
        hipMemcpy(d_a, a.data(), a.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(d_b, b.data(), b.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(d_sa, sa.data(), 64 * 4, hipMemcpyHostToDevice);
        hipMemcpy(d_sb, sb.data(), 64 * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(mfma6, dim3(1), dim3(64), 0, 0, d_a, d_b, d_sa, d_sb, d_y);
        std::vector<float> y(64 * 16);
        hipMemcpy(y.data(), d_y, y.size() * 4, hipMemcpyDeviceToHost);
        D.assign(32 * 32, 0.0);
        for (int l = 0; l < 64; ++l)
            for (int reg = 0; reg < 16; ++reg) D[((reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = y[l * 16 + reg];
    };
    std::vector<int> ones(64, 127);
    std::vector<double> D;
    struct Hyp { const char *name; int (*f)(int, int); };
    const Hyp hyps[] = {
        {"k = 32*(j>>4) + 16h + (j&15)", [](int h, int j) { return 32 * (j >> 4) + 16 * h + (j & 15); }},
        {"k = 32h + j", [](int h, int j) { return 32 * h + j; }},
        {"k = 16*(j>>3)... 8-groups", [](int h, int j) { return 16 * (j >> 3) + 8 * h + (j & 7); }},
    };
    int best = -1;
    for (int i = 0; i < 3; ++i) {
        run(hyps[i].f, ones, ones, D);
        double e = 0;
        for (int q = 0; q < 1024; ++q) e = std::fmax(e, std::fabs(D[q] - ref[q]));
        printf("mfma fp6 layout hypothesis '%s': max|D - ref| = %g\n", hyps[i].name, e);
        if (e == 0 && best < 0) best = i;
    }
    if (best >= 0) {
        auto f = hyps[best].f;
        for (int lane : {5, 37}) {
            std::vector<int> sa = ones;
            sa[lane] = 128;
            run(f, sa, ones, D);
            printf("scale_a of lane %d doubled -> rows changed:", lane);
            for (int r = 0; r < 32; ++r) {
                bool ch = false;
                for (int c = 0; c < 32; ++c) ch = ch || D[r * 32 + c] != ref[r * 32 + c];
                if (ch) printf(" %d", r);
            }
            // which block? compare with doubling block b of row 5
            for (int blk = 0; blk < 2; ++blk) {
                bool ok = true;
                for (int c = 0; c < 32; ++c) {
                    double extra = 0;
                    for (int k = 32 * blk; k < 32 * blk + 32; ++k) extra += A[5 * 64 + k] * B[k * 32 + c];
                    ok = ok && D[5 * 32 + c] == ref[5 * 32 + c] + extra;
                }
                if (ok) printf("  (= block %d of row 5 doubled)", blk);
            }
            printf("\n");
        }
        for (int lane : {7, 39}) {
            std::vector<int> sb = ones;
            sb[lane] = 128;
            run(f, ones, sb, D);
            printf("scale_b of lane %d doubled -> columns changed:", lane);
            for (int c = 0; c < 32; ++c) {
                bool ch = false;
                for (int r = 0; r < 32; ++r) ch = ch || D[r * 32 + c] != ref[r * 32 + c];
                if (ch) printf(" %d", c);
            }
            printf("\n");
        }
    }
    return 0;
}
