// Issue cost of VALU / LDS instruction classes on gfx950, in SIMD cycles per wave64 instruction: the per-class constants of the
// "issue" roof in bench.py (tools/isa_hist.py gives a kernel's static mix, SQ_INSTS_VALU its dynamic count).
//
//   hipcc --offload-arch=gfx950 -O3 tools/micro/valu_issue.hip -o /tmp/valu_issue && /tmp/valu_issue [--json out.json]
//
// Method: every wave runs ITERS x 32 copies of ONE instruction on 8 independent register sets (no dependent chain shorter than 8
// instructions).  ONE workgroup of 256 W threads per CU -- 96 KB of dynamic LDS per workgroup make a second one impossible, so every
// SIMD holds exactly W waves (a 256-thread grid is placed unevenly: some CUs get 1.5x the blocks and the figure scales with it) -- 256
// workgroups.  cycles = s_memtime delta of a wave / (32 ITERS W): with W waves sharing a SIMD's issue port that is the port time of one
// instruction.  s_memtime counts shader clocks; the clock it implies against s_memrealtime (100 MHz) is printed, and the hipEvent
// figure (kernel time x that clock / instructions per SIMD) beside it.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define REP32(X) REP8(X) REP8(X) REP8(X) REP8(X)

struct Result {
    const char *name;
    const char *cls;
    double cyc_memtime, cyc_wall;
};

struct Ticks {
    uint64_t core, real;
};
#define TICK_BEGIN() \
    const uint64_t t0 = __builtin_amdgcn_s_memtime(), q0 = __builtin_amdgcn_s_memrealtime()
#define TICK_END()                                                                                         \
    const uint64_t t1 = __builtin_amdgcn_s_memtime(), q1 = __builtin_amdgcn_s_memrealtime();               \
    if ((threadIdx.x & 63) == 0) ticks[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = Ticks{t1 - t0, q1 - q0}

#define KERNEL32(NAME, ASM, CONSTRAINT_T, INIT)                                                        \
    __global__ __launch_bounds__(1024) void k_##NAME(int iters, Ticks *ticks, uint32_t *sink) {        \
        CONSTRAINT_T r[8];                                                                             \
        for (int i = 0; i < 8; ++i) r[i] = (CONSTRAINT_T)(INIT + i + threadIdx.x);                     \
        CONSTRAINT_T s = (CONSTRAINT_T)(INIT + 3);                                                     \
        asm volatile("s_mov_b64 vcc, 0x55555555\n s_mov_b64 s[20:21], 0x33333333" ::: "vcc", "s20", "s21"); \
        TICK_BEGIN();                                                                                  \
        for (int it = 0; it < iters; ++it) {                                                           \
            _Pragma("unroll") for (int u = 0; u < 4; ++u) {                                            \
                _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(r[i]) : "v"(s)); \
            }                                                                                          \
        }                                                                                              \
        TICK_END();                                                                                    \
        CONSTRAINT_T acc = r[0];                                                                       \
        for (int i = 1; i < 8; ++i) acc += r[i];                                                       \
        if (acc == (CONSTRAINT_T)12345) sink[0] = 1;                                                   \
    }

// 32-bit classes ("%0" is read and written, "%1" a second VGPR operand)
KERNEL32(v_fma_f32, "v_fma_f32 %0, %0, %1, %0", float, 1.0f)
KERNEL32(v_add_f32, "v_add_f32 %0, %0, %1", float, 1.0f)
KERNEL32(v_mul_f32, "v_mul_f32 %0, %0, %1", float, 1.0f)
KERNEL32(v_min_f32, "v_min_f32 %0, %0, %1", float, 1.0f)
KERNEL32(v_med3_f32, "v_med3_f32 %0, %0, %1, 1.0", float, 1.0f)
KERNEL32(v_mov_b32, "v_mov_b32 %0, %1", float, 1.0f)
KERNEL32(v_max_f32, "v_max_f32 %0, %0, %1", float, 1.0f)
KERNEL32(v_sub_f32, "v_sub_f32 %0, %0, %1", float, 1.0f)
KERNEL32(v_fmac_f32, "v_fmac_f32 %0, %1, %1", float, 1.0f)
KERNEL32(v_cndmask_b32, "v_cndmask_b32 %0, %0, %1, vcc", float, 1.0f)
KERNEL32(v_cndmask_e64, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]", float, 1.0f)
KERNEL32(v_cndmask_0_e64, "v_cndmask_b32_e64 %0, 0, %1, s[20:21]", float, 1.0f)
KERNEL32(cmp_then_cndmask, "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc", float, 1.0f)
KERNEL32(v_bfe_i32, "v_bfe_i32 %0, %0, 3, 1", uint32_t, 77u)
KERNEL32(v_bfe_u32, "v_bfe_u32 %0, %0, 3, 5", uint32_t, 77u)
KERNEL32(v_and_or_b32, "v_and_or_b32 %0, %0, %1, %0", uint32_t, 77u)
KERNEL32(v_bfi_b32, "v_bfi_b32 %0, %1, %0, %0", uint32_t, 77u)
KERNEL32(v_lshlrev_b32, "v_lshlrev_b32 %0, 3, %0", uint32_t, 77u)
KERNEL32(v_lshrrev_b32, "v_lshrrev_b32 %0, 3, %0", uint32_t, 77u)
KERNEL32(v_xor_b32, "v_xor_b32 %0, %0, %1", uint32_t, 77u)
KERNEL32(v_add3_u32, "v_add3_u32 %0, %0, %1, %1", uint32_t, 77u)
KERNEL32(v_sub_co_u32, "v_sub_co_u32 %0, vcc, %0, %1", uint32_t, 77u)
KERNEL32(v_cvt_f32_i32, "v_cvt_f32_i32 %0, %0", uint32_t, 77u)
KERNEL32(v_ldexp_f32, "v_ldexp_f32 %0, %0, 1", float, 1.0f)
KERNEL32(v_or_b32, "v_or_b32 %0, %0, %1", uint32_t, 77u)
KERNEL32(v_sub_u32, "v_sub_u32 %0, %0, %1", uint32_t, 77u)
KERNEL32(v_subrev_f32, "v_subrev_f32 %0, %0, %1", float, 1.0f)
KERNEL32(v_ashrrev_i32, "v_ashrrev_i32 %0, 3, %0", uint32_t, 77u)
KERNEL32(v_not_b32, "v_not_b32 %0, %0", uint32_t, 77u)
KERNEL32(v_mul_u32_u24, "v_mul_u32_u24 %0, %0, %1", uint32_t, 77u)
KERNEL32(v_lshl_or_b32, "v_lshl_or_b32 %0, %0, 3, %1", uint32_t, 77u)
KERNEL32(v_cmp_eq_u32, "v_cmp_eq_u32 vcc, %0, %1", uint32_t, 77u)
KERNEL32(v_cmp_ne_u32_sgpr, "v_cmp_ne_u32 s[20:21], %0, %1", uint32_t, 77u)
KERNEL32(v_cvt_u32_f32, "v_cvt_u32_f32 %0, %0", float, 1.0f)
KERNEL32(v_trunc_f32, "v_trunc_f32 %0, %0", float, 1.0f)
KERNEL32(v_alignbit_b32, "v_alignbit_b32 %0, %0, %1, 7", uint32_t, 77u)
KERNEL32(v_fma_f32_sgpr, "v_fma_f32 %0, %0, s20, %0", float, 1.0f)
KERNEL32(v_add_f32_sgpr, "v_add_f32 %0, s20, %0", float, 1.0f)
KERNEL32(v_cmp_lt_f32, "v_cmp_lt_f32 vcc, %0, %1", float, 1.0f)
KERNEL32(v_cmp_sgpr, "v_cmp_lt_f32 s[20:21], %0, %1", float, 1.0f)
KERNEL32(v_and_b32, "v_and_b32 %0, %0, %1", uint32_t, 77u)
KERNEL32(v_add_u32, "v_add_u32 %0, %0, %1", uint32_t, 77u)
KERNEL32(v_lshl_add_u32, "v_lshl_add_u32 %0, %0, 2, %1", uint32_t, 77u)
KERNEL32(v_add_co_u32, "v_add_co_u32 %0, vcc, %0, %1", uint32_t, 77u)
KERNEL32(v_addc_co_u32, "v_addc_co_u32 %0, vcc, %0, %1, vcc", uint32_t, 77u)
KERNEL32(v_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1", uint32_t, 77u)
KERNEL32(v_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1", uint32_t, 77u)
KERNEL32(v_mad_u32_u24, "v_mad_u32_u24 %0, %0, %1, %0", uint32_t, 77u)
KERNEL32(v_cvt_f32_u32, "v_cvt_f32_u32 %0, %0", uint32_t, 77u)
KERNEL32(v_cvt_i32_f32, "v_cvt_i32_f32 %0, %0", float, 1.0f)
KERNEL32(v_rndne_f32, "v_rndne_f32 %0, %0", float, 1.0f)
KERNEL32(v_rcp_f32, "v_rcp_f32 %0, %0", float, 1.0f)
KERNEL32(v_exp_f32, "v_exp_f32 %0, %0", float, 1.0f)
KERNEL32(v_log_f32, "v_log_f32 %0, %0", float, 1.0f)
KERNEL32(v_readfirstlane, "v_readfirstlane_b32 s20, %0", float, 1.0f)
KERNEL32(v_mov_dpp, "v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf", float, 1.0f)
// 64-bit classes (register pairs)
KERNEL32(v_mov_b64, "v_mov_b64 %0, %1", double, 1.0)
KERNEL32(v_lshl_add_u64, "v_lshl_add_u64 %0, %0, 2, %1", uint64_t, 77ull)
KERNEL32(v_lshlrev_b64, "v_lshlrev_b64 %0, 3, %0", uint64_t, 77ull)
KERNEL32(v_add_f64, "v_add_f64 %0, %0, %1", double, 1.0)
KERNEL32(v_fma_f64, "v_fma_f64 %0, %0, %1, %0", double, 1.0)
KERNEL32(v_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %0", double, 1.0)
KERNEL32(v_pk_mul_f32, "v_pk_mul_f32 %0, %0, %1", double, 1.0)
KERNEL32(v_pk_add_f32, "v_pk_add_f32 %0, %0, %1", double, 1.0)

// mixed widths: written by hand
__global__ __launch_bounds__(1024) void k_v_cvt_f64_f32(int iters, Ticks *ticks, uint32_t *sink) {
    float r[8];
    double d[8];
    for (int i = 0; i < 8; ++i) r[i] = 1.0f + i + threadIdx.x;
    TICK_BEGIN();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(r[i]));
        }
    }
    TICK_END();
    double acc = 0;
    for (int i = 0; i < 8; ++i) acc += d[i];
    if (acc == 12345.0) sink[0] = 1;
}

__global__ __launch_bounds__(1024) void k_v_mad_u64_u32(int iters, Ticks *ticks, uint32_t *sink) {
    uint64_t r[8];
    uint32_t a = 77u + threadIdx.x, b = 5u;
    for (int i = 0; i < 8; ++i) r[i] = 77ull + i + threadIdx.x;
    TICK_BEGIN();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b) : "vcc");
        }
    }
    TICK_END();
    uint64_t acc = 0;
    for (int i = 0; i < 8; ++i) acc += r[i];
    if (acc == 12345ull) sink[0] = 1;
}

// ds_add_u64 (no return), WAYS lanes of a wave on the same address: lane l of wave w adds into word ((w & 3) * 64 + l / WAYS) * STRIDE.
// (dynamic LDS: the 96 KB that keep a second workgroup off the CU; the table is its first 16 KB)
template <int WAYS, int STRIDE>
__global__ __launch_bounds__(1024) void k_ds_add_u64(int iters, Ticks *ticks, uint32_t *sink) {
    extern __shared__ unsigned long long tab[];
    for (int i = threadIdx.x; i < 256 * 8; i += blockDim.x) tab[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3;
    unsigned long long *p = tab + ((wave * 64 + lane / WAYS) * STRIDE) % (256 * 8);
    unsigned long long v = threadIdx.x + 1;
    TICK_BEGIN();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) asm volatile("ds_add_u64 %0, %1" ::"v"((uint32_t)(uintptr_t)p), "v"(v) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    TICK_END();
    __syncthreads();
    if (tab[threadIdx.x] == 12345ull) sink[0] = 1;
}

// ds_read_b128 gathers: WAYS lanes per 16-byte address, consecutive addresses STRIDE x 16 bytes apart
template <int WAYS, int STRIDE>
__global__ __launch_bounds__(1024) void k_ds_read_b128(int iters, Ticks *ticks, uint32_t *sink) {
    extern __shared__ unsigned long long tab[];
    float4 *tab4 = reinterpret_cast<float4 *>(tab);
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) tab4[i] = float4{1, 2, 3, 4};
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = (threadIdx.x >> 6) & 3;
    const float4 *p = tab4 + ((wave * 64 + lane / WAYS) * STRIDE) % 1024;
    float4 acc = {0, 0, 0, 0};
    TICK_BEGIN();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            float4 x;
            asm volatile("ds_read_b128 %0, %1" : "=v"(x) : "v"((uint32_t)(uintptr_t)p) : "memory");
            asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
            acc.x += x.x;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    TICK_END();
    if (acc.x == 12345.0f) sink[0] = 1;
}

typedef void (*kern_t)(int, Ticks *, uint32_t *);
#define HIP_OK(x)                                                                        \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                      \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

static const int kLdsBytes = 96 * 1024;  // per workgroup: more than half of a CU's 160 KB

struct Measured {
    double cyc_core, cyc_wall, mhz;
};

// one workgroup of 256 * waves_per_simd threads on each of the 256 CUs
static Measured measure(kern_t k, int waves_per_simd, int iters, int per_iter = 32) {
    const int blocks = 256, threads = 256 * waves_per_simd, waves = blocks * threads / 64;
    Ticks *ticks;
    uint32_t *sink;
    HIP_OK(hipMalloc(&ticks, sizeof(Ticks) * waves));
    HIP_OK(hipMalloc(&sink, 4));
    HIP_OK(hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes));
    hipEvent_t e0, e1;
    HIP_OK(hipEventCreate(&e0));
    HIP_OK(hipEventCreate(&e1));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), kLdsBytes, 0, iters / 8 + 1, ticks, sink);
    HIP_OK(hipDeviceSynchronize());
    HIP_OK(hipEventRecord(e0));
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), kLdsBytes, 0, iters, ticks, sink);
    HIP_OK(hipEventRecord(e1));
    HIP_OK(hipEventSynchronize(e1));
    float ms;
    HIP_OK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<Ticks> h((size_t)waves);
    HIP_OK(hipMemcpy(h.data(), ticks, h.size() * sizeof(Ticks), hipMemcpyDeviceToHost));
    double core = 0, real = 0;
    for (const Ticks &x : h) {
        core += (double)x.core;
        real += (double)x.real;
    }
    const double n = (double)per_iter * iters * waves_per_simd;  // instructions one SIMD issued while a wave ran
    Measured m;
    m.mhz = core / real * 100.0;  // s_memrealtime: 100 MHz
    m.cyc_core = core / (double)waves / n;
    m.cyc_wall = ms * 1e-3 * m.mhz * 1e6 / n;
    HIP_OK(hipFree(ticks));
    HIP_OK(hipFree(sink));
    return m;
}

int main(int argc, char **argv) {
    const char *json = nullptr;
    int waves = 4, iters = 4000;
    for (int i = 1; i < argc; ++i) {
        if (!strcmp(argv[i], "--json") && i + 1 < argc) json = argv[++i];
        else if (!strcmp(argv[i], "--waves") && i + 1 < argc) waves = atoi(argv[++i]);
        else if (!strcmp(argv[i], "--iters") && i + 1 < argc) iters = atoi(argv[++i]);
    }
    hipDeviceProp_t prop;
    HIP_OK(hipGetDeviceProperties(&prop, 0));
    printf("device %s, %d CUs, clockRate %.0f MHz, %d waves per SIMD\n", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1e3, waves);
    struct Entry {
        const char *name, *cls;
        kern_t k;
        int per_iter;
    };
#define E(NAME, CLS) {#NAME, CLS, k_##NAME, 32}
    // class labels: tools/isa_hist.py sorts a kernel's instructions into the same classes (fast32 = the opcodes that measure ~1.8 cycles;
    // every other 32-bit VALU opcode is priced as slow32)
    const Entry entries[] = {
        E(v_fma_f32, "fast32"), E(v_add_f32, "fast32"), E(v_sub_f32, "fast32"), E(v_subrev_f32, "fast32"), E(v_mul_f32, "fast32"),
        E(v_fma_f32_sgpr, "fast32"), E(v_add_f32_sgpr, "fast32"), E(v_and_b32, "fast32"), E(v_or_b32, "fast32"), E(v_xor_b32, "fast32"),
        E(v_not_b32, "fast32"), E(v_lshrrev_b32, "fast32"), E(v_add_u32, "fast32"), E(v_sub_u32, "fast32"), E(v_mov_b32, "mov32"),
        E(v_fmac_f32, "fmac"), E(v_min_f32, "slow32"), E(v_max_f32, "slow32"), E(v_med3_f32, "slow32"), E(v_cndmask_e64, "cndmask"),
        E(v_cndmask_0_e64, "cndmask"), {"v_cmp + v_cndmask (pair, vcc)", "pair", k_cmp_then_cndmask, 32},
        {"v_cndmask_b32 vcc, back to back", "other", k_v_cndmask_b32, 32}, E(v_cmp_lt_f32, "slow32"), E(v_cmp_sgpr, "slow32"),
        E(v_cmp_eq_u32, "slow32"), E(v_cmp_ne_u32_sgpr, "slow32"), E(v_bfe_i32, "slow32"), E(v_bfe_u32, "slow32"), E(v_and_or_b32, "slow32"),
        E(v_bfi_b32, "slow32"), E(v_lshlrev_b32, "slow32"), E(v_ashrrev_i32, "slow32"), E(v_lshl_or_b32, "slow32"), E(v_alignbit_b32, "slow32"),
        E(v_add3_u32, "slow32"), E(v_lshl_add_u32, "slow32"), E(v_add_co_u32, "slow32"), E(v_sub_co_u32, "slow32"), E(v_addc_co_u32, "slow32"),
        E(v_mul_u32_u24, "slow32"), E(v_mad_u32_u24, "slow32"), E(v_cvt_f32_u32, "slow32"), E(v_cvt_f32_i32, "slow32"), E(v_cvt_i32_f32, "slow32"),
        E(v_cvt_u32_f32, "slow32"), E(v_rndne_f32, "slow32"), E(v_trunc_f32, "slow32"), E(v_ldexp_f32, "slow32"), E(v_mov_dpp, "slow32"),
        E(v_mul_lo_u32, "mul32"), E(v_mul_hi_u32, "mul32"), E(v_rcp_f32, "trans32"), E(v_exp_f32, "trans32"), E(v_log_f32, "trans32"),
        E(v_readfirstlane, "dpp_move"), E(v_mov_b64, "mov64"), E(v_lshl_add_u64, "int64"), E(v_lshlrev_b64, "int64"), E(v_add_f64, "f64"),
        E(v_fma_f64, "f64"), E(v_cvt_f64_f32, "cvt64"), E(v_mad_u64_u32, "mad_u64"), E(v_pk_fma_f32, "pk32"), E(v_pk_mul_f32, "pk32"),
        E(v_pk_add_f32, "pk32"),
        {"ds_add_u64 distinct, stride 1", "lds", k_ds_add_u64<1, 1>, 32}, {"ds_add_u64 distinct, stride 4", "lds", k_ds_add_u64<1, 4>, 32},
        {"ds_add_u64 distinct, stride 5", "lds", k_ds_add_u64<1, 5>, 32}, {"ds_add_u64 4 lanes/address", "lds", k_ds_add_u64<4, 5>, 32},
        {"ds_add_u64 8 lanes/address", "lds", k_ds_add_u64<8, 5>, 32}, {"ds_add_u64 64 lanes/address", "lds", k_ds_add_u64<64, 5>, 32},
        {"ds_read_b128 distinct, stride 1", "lds", k_ds_read_b128<1, 1>, 32}, {"ds_read_b128 distinct, stride 4", "lds", k_ds_read_b128<1, 4>, 32},
        {"ds_read_b128 8 lanes/address", "lds", k_ds_read_b128<8, 4>, 32},
    };
    std::string out;
    double mhz_sum = 0;
    int count = 0;
    for (const Entry &e : entries) {
        Measured m = measure(e.k, waves, iters, e.per_iter);
        const int insts = strstr(e.name, "pair") ? 2 : 1;
        printf("%-34s %-8s %6.2f cycles per wave-instruction%s (s_memtime)   %6.2f (hipEvents)   clock %.0f MHz\n", e.name, e.cls, m.cyc_core,
               insts == 2 ? " PAIR" : "", m.cyc_wall, m.mhz);
        char buf[256];
        snprintf(buf, sizeof buf, "%s  \"%s\": {\"class\": \"%s\", \"cycles\": %.3f, \"cycles_hipevents\": %.3f, \"clock_mhz\": %.0f}", count ? ",\n" : "",
                 e.name, e.cls, m.cyc_core, m.cyc_wall, m.mhz);
        out += buf;
        mhz_sum += m.mhz;
        ++count;
    }
    if (json) {
        FILE *f = fopen(json, "w");
        if (f) {
            fprintf(f, "{\n \"device\": \"%s\", \"waves_per_simd\": %d, \"mean_clock_mhz\": %.0f,\n \"how\": \"tools/micro/valu_issue.hip: one workgroup per CU, "
                       "s_memtime delta of a wave / instructions its SIMD issued\",\n \"cycles\": {\n%s\n }\n}\n",
                    prop.gcnArchName, waves, mhz_sum / count, out.c_str());
            fclose(f);
        }
    }
    return 0;
}
