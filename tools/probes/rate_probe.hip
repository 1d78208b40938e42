// Issue-rate probe (development tool): cycles per instruction of the MFMA forms and of the conversion
// instructions the f16mx8 / f16mx6 main loops are built from, one wavefront per SIMD (one 256-thread
// workgroup), s_memtime around a loop of independent instructions.  Standalone: hipcc -> binary.
//   hipcc --offload-arch=gfx950 -O3 -o rate_probe rate_probe.hip && ./rate_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x6 __attribute__((ext_vector_type(6)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));
typedef short s2 __attribute__((ext_vector_type(2)));

constexpr int ITERS = 256;

struct Result { unsigned long long cycles, ticks; };

#define BEGIN()                                                    \
    __syncthreads();                                               \
    const unsigned long long t0 = __builtin_readcyclecounter();    \
    const unsigned long long w0 = wall_clock64();
#define END(slot)                                                                  \
    const unsigned long long t1 = __builtin_readcyclecounter();                    \
    const unsigned long long w1 = wall_clock64();                                  \
    if (threadIdx.x == 0) { res[slot].cycles = t1 - t0; res[slot].ticks = w1 - w0; }

template <int CBSZ, int BLGP>
__global__ __launch_bounds__(256) void mx_rate(const int *src, float *sink, Result *res, int slot)
{
    i32x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = src[threadIdx.x * 8 + i]; b[i] = src[2048 + threadIdx.x * 8 + i]; }
    f32x16 c[4];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) c[j][i] = 0.f;
    BEGIN();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
            c[u & 3] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c[u & 3], CBSZ, BLGP, 0, 127, 0, 127);
    }
    END(slot);
    float s = 0;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) s += c[j][i];
    sink[threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void f16_rate(const int *src, float *sink, Result *res, int slot)
{
    f16x8 a, b;
    i32x4 ta, tb;
    for (int i = 0; i < 4; ++i) { ta[i] = src[threadIdx.x * 4 + i]; tb[i] = src[2048 + threadIdx.x * 4 + i]; }
    a = __builtin_bit_cast(f16x8, ta);
    b = __builtin_bit_cast(f16x8, tb);
    f32x16 c[4];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) c[j][i] = 0.f;
    BEGIN();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) c[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c[u & 3], 0, 0, 0);
    }
    END(slot);
    float s = 0;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) s += c[j][i];
    sink[threadIdx.x] = s;
}

// VALU forms: 16 independent instances per iteration, results folded into the sink after the loop
template <int KIND>
__global__ __launch_bounds__(256) void valu_rate(const int *src, float *sink, Result *res, int slot)
{
    float x[32];
    for (int i = 0; i < 32; ++i) x[i] = __int_as_float(src[threadIdx.x * 32 + i] & 0x3fffffff);
    const float sc = __int_as_float(src[5] | 0x3f800000);
    int r[16] = {};
    double d[16] = {};
    BEGIN();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (KIND == 0) asm volatile("v_add_f32 %0, %1, %2" : "=v"(r[u]) : "v"(x[u]), "v"(x[u + 16]));
            if (KIND == 1) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r[u]) : "v"(x[u]), "v"(x[u + 16]));
            if (KIND == 2) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r[u]) : "v"(x[u]), "v"(x[u + 16]));
            if (KIND == 3) asm volatile("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3" : "+v"(r[u]) : "v"(x[u]), "v"(x[u + 16]), "v"(sc));
            if (KIND == 4) asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2" : "+v"(r[u]) : "v"(x[u]), "v"(x[u + 16]));
            if (KIND == 5) asm volatile("v_cvt_scalef32_pk_fp8_f16 %0, %1, %2" : "+v"(r[u]) : "v"(x[u]), "v"(sc));
            if (KIND == 6) asm volatile("v_perm_b32 %0, %1, %2, %3" : "=v"(r[u]) : "v"(x[u]), "v"(x[u + 16]), "v"(sc));
            if (KIND == 7) asm volatile("v_and_or_b32 %0, %1, %2, %3" : "=v"(r[u]) : "v"(x[u]), "v"(x[u + 16]), "v"(sc));
            if (KIND == 8) asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(r[u]) : "v"(x[u]), "v"(x[u + 16]), "v"(sc));
            if (KIND == 9) asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(d[u]) : "v"(*(double *)&x[2 * (u & 7)]), "v"(*(double *)&x[16 + 2 * (u & 7)]));
            if (KIND == 10) asm volatile("v_cvt_scalef32_pk_fp4_f32 %0, %1, %2, %3" : "+v"(r[u]) : "v"(x[u]), "v"(x[u + 16]), "v"(sc));
            if (KIND == 11) asm volatile("v_rcp_f32 %0, %1" : "=v"(r[u]) : "v"(x[u]));
        }
    }
    END(slot);
    float s = 0;
    for (int i = 0; i < 16; ++i) s += r[i] + (float)d[i];
    sink[threadIdx.x] = s + sc;
}

// the 32-value fp6 converts: 4 independent instances per unrolled group of 16 -> count = 4 per "u & 3 == 0"
__global__ __launch_bounds__(256) void v_2xpk16_fp6(const int *src, float *sink, Result *res, int slot)
{
    f32x16 x0, x1;
    for (int i = 0; i < 16; ++i) {
        x0[i] = __int_as_float(src[threadIdx.x * 32 + i] & 0x3fffffff);
        x1[i] = __int_as_float(src[threadIdx.x * 32 + 16 + i] & 0x3fffffff);
    }
    const float sc = __int_as_float(src[5] | 0x3f800000);
    u32x6 r[4] = {};
    BEGIN();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
            asm volatile("v_cvt_scalef32_2xpk16_fp6_f32 %0, %1, %2, %3" : "=v"(r[u & 3]) : "v"(x0), "v"(x1), "v"(sc));
    }
    END(slot);
    unsigned s = 0;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 6; ++i) s += r[j][i];
    sink[threadIdx.x] = (float)s;
}
__global__ __launch_bounds__(256) void v_pk32_fp6_f16(const int *src, float *sink, Result *res, int slot)
{
    i32x8 q0, q1;
    for (int i = 0; i < 8; ++i) { q0[i] = src[threadIdx.x * 16 + i] & 0x3bff3bff; q1[i] = src[threadIdx.x * 16 + 8 + i] & 0x3bff3bff; }
    typedef int i32x16 __attribute__((ext_vector_type(16)));
    i32x16 t;
    for (int i = 0; i < 8; ++i) { t[i] = q0[i]; t[8 + i] = q1[i]; }
    const f16x32 x = __builtin_bit_cast(f16x32, t);
    const float sc = __int_as_float(src[5] | 0x3f800000);
    u32x6 r[4] = {};
    BEGIN();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
            asm volatile("v_cvt_scalef32_pk32_fp6_f16 %0, %1, %2" : "=v"(r[u & 3]) : "v"(x), "v"(sc));
    }
    END(slot);
    unsigned s = 0;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 6; ++i) s += r[j][i];
    sink[threadIdx.x] = (float)s;
}

// one f16 MFMA per gap + NF fillers of a kind: what hides in a 32-cycle gap
template <int KIND, int NF>
__global__ __launch_bounds__(256) void gap_fill(const int *src, float *sink, Result *res, int slot)
{
    f16x8 a, b;
    i32x4 ta, tb;
    for (int i = 0; i < 4; ++i) { ta[i] = src[threadIdx.x * 4 + i]; tb[i] = src[2048 + threadIdx.x * 4 + i]; }
    a = __builtin_bit_cast(f16x8, ta);
    b = __builtin_bit_cast(f16x8, tb);
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = __int_as_float(src[threadIdx.x * 32 + i] & 0x3fffffff);
    const float sc = __int_as_float(src[5] | 0x3f800000);
    int r[8] = {};
    f32x16 c[4];
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) c[j][i] = 0.f;
    BEGIN();
    for (int it = 0; it < ITERS; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(c[u & 3]) : "v"(a), "v"(b));
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                if (KIND == 0) asm volatile("v_add_f32 %0, %1, %2" : "=v"(r[f]) : "v"(x[f]), "v"(x[f + 8]));
                if (KIND == 1) asm volatile("v_cvt_scalef32_pk_fp8_f32 %0, %1, %2, %3" : "+v"(r[f]) : "v"(x[f]), "v"(x[f + 8]), "v"(sc));
                if (KIND == 2) asm volatile("v_cvt_pk_fp8_f32 %0, %1, %2" : "+v"(r[f]) : "v"(x[f]), "v"(x[f + 8]));
                if (KIND == 3) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(r[f]) : "v"(x[f]), "v"(x[f + 8]));
                if (KIND == 4) asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r[f]) : "v"(x[f]), "v"(x[f + 8]));
                if (KIND == 5) asm volatile("v_cvt_scalef32_pk_fp8_f16 %0, %1, %2" : "+v"(r[f]) : "v"(x[f]), "v"(sc));
            }
        }
    }
    END(slot);
    float s = 0;
    for (int j = 0; j < 4; ++j)
        for (int i = 0; i < 16; ++i) s += c[j][i];
    for (int f = 0; f < 8; ++f) s += r[f];
    sink[threadIdx.x] = s;
}

int main()
{
    int *src;
    float *sink;
    Result *res;
    hipMalloc(&src, 1 << 20);
    hipMalloc(&sink, 4096);
    hipMalloc(&res, 64 * sizeof(Result));
    std::vector<int> h(1 << 18);
    srand(1);
    for (auto &v : h) v = rand() ^ (rand() << 16);
    hipMemcpy(src, h.data(), 1 << 20, hipMemcpyHostToDevice);
    hipMemset(res, 0, 64 * sizeof(Result));
    int slot = 0;
    std::vector<const char *> names;
    std::vector<int> per_iter;
#define RUN(K, NAME, N)                                                      \
    do {                                                                     \
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(K, dim3(1), dim3(256), 0, 0, src, sink, res, slot); \
        names.push_back(NAME);                                               \
        per_iter.push_back(N);                                               \
        ++slot;                                                              \
    } while (0)
    RUN(f16_rate, "mfma_f32_32x32x16_f16", 16);
    RUN((mx_rate<0, 0>), "mfma_scale 32x32x64 fp8 x fp8", 16);
    RUN((mx_rate<2, 2>), "mfma_scale 32x32x64 fp6(e2m3) x fp6(e2m3)", 16);
    RUN((mx_rate<3, 3>), "mfma_scale 32x32x64 bf6(e3m2) x bf6(e3m2)", 16);
    RUN((mx_rate<4, 4>), "mfma_scale 32x32x64 fp4 x fp4", 16);
    RUN((mx_rate<0, 2>), "mfma_scale 32x32x64 fp8 x fp6", 16);
    RUN((mx_rate<2, 0>), "mfma_scale 32x32x64 fp6 x fp8", 16);
    RUN((mx_rate<2, 4>), "mfma_scale 32x32x64 fp6 x fp4", 16);
    RUN((valu_rate<0>), "v_add_f32", 16);
    RUN((valu_rate<1>), "v_cvt_pk_f16_f32", 16);
    RUN((valu_rate<2>), "v_fma_mix_f32", 16);
    RUN((valu_rate<3>), "v_cvt_scalef32_pk_fp8_f32", 16);
    RUN((valu_rate<4>), "v_cvt_pk_fp8_f32", 16);
    RUN((valu_rate<5>), "v_cvt_scalef32_pk_fp8_f16", 16);
    RUN((valu_rate<6>), "v_perm_b32", 16);
    RUN((valu_rate<7>), "v_and_or_b32", 16);
    RUN((valu_rate<8>), "v_max3_f32", 16);
    RUN((valu_rate<9>), "v_pk_mul_f32", 16);
    RUN((valu_rate<10>), "v_cvt_scalef32_pk_fp4_f32", 16);
    RUN((valu_rate<11>), "v_rcp_f32", 16);
    RUN(v_2xpk16_fp6, "v_cvt_scalef32_2xpk16_fp6_f32 (32 values)", 16);
    RUN(v_pk32_fp6_f16, "v_cvt_scalef32_pk32_fp6_f16 (32 values)", 16);
    RUN((gap_fill<0, 0>), "f16 mfma + 0 fillers", 16);
    RUN((gap_fill<0, 4>), "f16 mfma + 4 v_add", 16);
    RUN((gap_fill<0, 6>), "f16 mfma + 6 v_add", 16);
    RUN((gap_fill<1, 2>), "f16 mfma + 2 cvt_scalef32_pk_fp8_f32", 16);
    RUN((gap_fill<1, 4>), "f16 mfma + 4 cvt_scalef32_pk_fp8_f32", 16);
    RUN((gap_fill<2, 4>), "f16 mfma + 4 cvt_pk_fp8_f32", 16);
    RUN((gap_fill<3, 4>), "f16 mfma + 4 cvt_pk_f16_f32", 16);
    RUN((gap_fill<4, 4>), "f16 mfma + 4 fma_mix", 16);
    RUN((gap_fill<5, 4>), "f16 mfma + 4 cvt_scalef32_pk_fp8_f16", 16);
    hipDeviceSynchronize();
    std::vector<Result> r(64);
    hipMemcpy(r.data(), res, 64 * sizeof(Result), hipMemcpyDeviceToHost);
    for (int i = 0; i < slot; ++i) {
        const double cyc = (double)r[i].cycles / (ITERS * per_iter[i]);
        const double ghz = r[i].ticks ? (double)r[i].cycles / (r[i].ticks * 10.0) : 0.0;   // wall_clock64: 100 MHz
        printf("%-48s %8.2f cycles per instruction (or per gap)   clock %.2f GHz\n", names[i], cyc, ghz);
    }
    return 0;
}
