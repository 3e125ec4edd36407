// Practical fp32 MFMA ceiling on this box: chains of v_mfma_f32_32x32x2_f32 only (no loads in the loop), optionally
// interleaved with V packed FMAs per MFMA.  hipcc --offload-arch=gfx950 -O3 mfma_peak.hip -o mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int CHAINS, int VALU>
__global__ __launch_bounds__(256) void k(int iters, float *out, float a0, float b0) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    f32x2 v[4] = {{a, b}, {b, a}, {a, a}, {b, b}};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < VALU; ++u) v[u & 3] = __builtin_elementwise_fma(v[u & 3], v[(u + 1) & 3], v[(u + 2) & 3]);
        }
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c)
        for (int i = 0; i < 16; ++i) s += acc[c][i];
    for (int u = 0; u < 4; ++u) s += v[u].x + v[u].y;
    if (s == 12345.f) out[0] = s;
}

template <int CHAINS, int VALU>
void run(int blocks, int iters) {
    float *out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k<CHAINS, VALU><<<blocks, 256>>>(iters, out, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<CHAINS, VALU><<<blocks, 256>>>(iters, out, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = 4096.0 * CHAINS * iters * (double)blocks * 4;
    double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * CHAINS * blocks * 4 / 1024.0);
    printf("chains %d valu/mfma %d blocks %d: %.3f ms  %.1f TFLOP/s  (%.1f cycles@2.4GHz per MFMA slot per SIMD)\n", CHAINS, VALU, blocks, ms,
           flops / ms / 1e9, cyc);
}


// The forward kernel's loop shape without any memory traffic: two 9-deep MFMA chains, then relu (32 v_max) and the policy
// head's 48 packed FMAs on the results.
template <int NMAX, int NFMA>
__global__ __launch_bounds__(256) void k_shape(int iters, float *out, float a0, float b0) {
    float a = a0 + threadIdx.x, b = b0;
    f32x2 acc[6] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}, {0, 0}};
    f32x2 w[6] = {{a, b}, {b, a}, {a, a}, {b, b}, {a, 1.f}, {b, 2.f}};
    for (int it = 0; it < iters; ++it) {
        f32x16 c0, c1;
        for (int i = 0; i < 16; ++i) { c0[i] = a; c1[i] = b; }
#pragma unroll
        for (int ks = 0; ks < 9; ++ks) {
            c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, c1, 0, 0, 0);
        }
        a += 1.f;
        f32x2 h[16];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            h[i] = NMAX ? f32x2{fmaxf(c0[2 * i], 0.f), fmaxf(c0[2 * i + 1], 0.f)} : f32x2{c0[2 * i], c0[2 * i + 1]};
            h[8 + i] = NMAX ? f32x2{fmaxf(c1[2 * i], 0.f), fmaxf(c1[2 * i + 1], 0.f)} : f32x2{c1[2 * i], c1[2 * i + 1]};
        }
        if (NFMA == 0) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i % 6] += h[i];
        }
#pragma unroll
        for (int i = 0; i < NFMA; ++i) acc[i % 6] = __builtin_elementwise_fma(w[i % 6], h[i % 16], acc[i % 6]);
    }
    float s = 0;
    for (int u = 0; u < 6; ++u) s += acc[u].x + acc[u].y;
    if (s == 12345.f) out[0] = s;
}

template <int NMAX, int NFMA>
void run_shape(int blocks, int iters) {
    float *out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k_shape<NMAX, NFMA><<<blocks, 256>>>(iters, out, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_shape<NMAX, NFMA><<<blocks, 256>>>(iters, out, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double flops = 4096.0 * 18 * iters * (double)blocks * 4;
    double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * blocks * 4 / 1024.0);
    printf("shape relu %d fma %d blocks %d: %.3f ms  %.1f TFLOP/s (MFMA)  %.0f cycles@2.4GHz per iteration per SIMD (MFMA alone: 1152)\n", NMAX, NFMA,
           blocks, ms, flops / ms / 1e9, cyc);
}

// Same as k<> but with SCALAR fp32 FMAs (2 per packed one) / v_max beside the MFMAs.
template <int CHAINS, int VALU, int KIND>
__global__ __launch_bounds__(256) void k_scalar(int iters, float *out, float a0, float b0) {
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c)
        for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    float v[8] = {a, b, b, a, a, a, b, b};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int c = 0; c < CHAINS; ++c) {
            acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < VALU; ++u) {
                float &d = v[u & 7];
                const float p = v[(u + 1) & 7], q = v[(u + 2) & 7];
                if (KIND == 0) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(d) : "v"(p), "v"(q));
                else if (KIND == 1) asm volatile("v_max_f32 %0, %1, %0" : "+v"(d) : "v"(p));
                else if (KIND == 2) asm volatile("v_add_f32 %0, %1, |%1|" : "+v"(d) : "v"(p));
                else if (KIND == 3) asm volatile("v_max_i32 %0, %1, %0" : "+v"(d) : "v"(p));
                else if (KIND == 4) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(d) : "v"(p));
                else if (KIND == 5) asm volatile("v_fma_f32 %0, %1, |%2|, %0" : "+v"(d) : "v"(p), "v"(q));
                else if (KIND == 6) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(d) : "v"(p), "v"(q));
                else if (KIND == 7) asm volatile("v_max_f32 %0, 0, %1" : "=v"(d) : "v"(p));
                else if (KIND == 8) asm volatile("v_mov_b32 %0, %1" : "=v"(d) : "v"(p));
            }
        }
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c)
        for (int i = 0; i < 16; ++i) s += acc[c][i];
    for (int u = 0; u < 8; ++u) s += v[u];
    if (s == 12345.f) out[0] = s;
}
static const char *kinds[] = {"v_fma_f32", "v_max_f32", "v_add_f32 abs", "v_max_i32", "v_mul_f32", "v_fma_f32 abs", "v_fmac_f32", "v_max_f32 0,x (indep)", "v_mov_b32"};
template <int CHAINS, int VALU, int KIND>
void run_scalar(int blocks, int iters) {
    float *out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    k_scalar<CHAINS, VALU, KIND><<<blocks, 256>>>(iters, out, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_scalar<CHAINS, VALU, KIND><<<blocks, 256>>>(iters, out, 1.f, 2.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * CHAINS * blocks * 4 / 1024.0);
    printf("scalar %s: chains %d valu/mfma %d blocks %d: %.3f ms  %.1f cycles@2.4GHz per MFMA slot per SIMD\n", kinds[KIND], CHAINS, VALU, blocks, ms, cyc);
}

int main() {
    run_scalar<2, 16, 0>(1024, 10000);
    run_scalar<2, 16, 1>(1024, 10000);
    run_scalar<2, 16, 2>(1024, 10000);
    run_scalar<2, 16, 3>(1024, 10000);
    run_scalar<2, 16, 4>(1024, 10000);
    run_scalar<2, 16, 5>(1024, 10000);
    run_scalar<2, 16, 6>(1024, 10000);
    run_scalar<2, 16, 7>(1024, 10000);
    run_scalar<2, 16, 8>(1024, 10000);
    return 0;
    run_shape<0, 0>(1024, 2000);
    run_shape<1, 0>(1024, 2000);
    run_shape<1, 16>(1024, 2000);
    run_shape<1, 48>(1024, 2000);
    run_shape<0, 48>(1024, 2000);
    run_shape<1, 48>(512, 2000);
    run_shape<1, 48>(256, 2000);
    run_shape<1, 48>(2048, 2000);

    run<1, 0>(1024, 20000);
    run<2, 0>(1024, 10000);
    run<4, 0>(1024, 5000);
    run<4, 0>(2048, 5000);
    run<4, 0>(256, 5000);
    run<2, 1>(1024, 10000);
    run<2, 2>(1024, 10000);
    run<2, 4>(1024, 10000);
    run<2, 8>(1024, 10000);
    run<2, 4>(2048, 10000);
    run<2, 4>(3072, 10000);
    return 0;
}
