// What does a grid-wide barrier cost inside a persistent kernel on this part?  (r05: would the step's tail -- finish, backward, reduction,
// optimiser, the next table launch: five launches of a few microseconds of work each -- be cheaper as phases of one launch?)
//   hipcc --offload-arch=gfx950 -O3 tools/micro/grid_barrier.hip -o /tmp/grid_barrier && /tmp/grid_barrier
// Every workgroup writes `bytes` of its own slice, barrier, reads the slice of workgroup (b + 97) % n written before the barrier and checks
// it.  Reported: microseconds per (write + barrier + read) round against the same work as separate launches, and against a graph of them.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define OK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                   \
            std::exit(1);                                                                  \
        }                                                                                  \
    } while (0)

struct Barrier {
    unsigned int count, gen;
};

__device__ __forceinline__ void grid_barrier(Barrier *b, unsigned int n_blocks) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned int gen = __hip_atomic_load(&b->gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();  // release: this workgroup's writes (and its XCD's dirty lines) reach memory
        if (__hip_atomic_fetch_add(&b->count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == n_blocks - 1) {
            __hip_atomic_store(&b->count, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&b->gen, gen + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (__hip_atomic_load(&b->gen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == gen) __builtin_amdgcn_s_sleep(1);
        }
        __threadfence();  // acquire
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_persistent(float *buf, int words, int rounds, Barrier *bar, int *bad) {
    const int n = gridDim.x;
    for (int r = 0; r < rounds; ++r) {
        float *mine = buf + (size_t)blockIdx.x * words;
        for (int i = threadIdx.x; i < words; i += 256) mine[i] = (float)(r * 1000 + blockIdx.x);
        grid_barrier(bar, n);
        const int other = (blockIdx.x + 97) % n;
        const float *theirs = buf + (size_t)other * words;
        float s = 0.f;
        for (int i = threadIdx.x; i < words; i += 256) s += __builtin_nontemporal_load(theirs + i) - (float)(r * 1000 + other);
        if (s != 0.f) atomicAdd(bad, 1);
        grid_barrier(bar, n);  // (nobody overwrites a slice that is still being read)
    }
}

__global__ __launch_bounds__(256) void k_write(float *buf, int words, int r) {
    float *mine = buf + (size_t)blockIdx.x * words;
    for (int i = threadIdx.x; i < words; i += 256) mine[i] = (float)(r * 1000 + blockIdx.x);
}
__global__ __launch_bounds__(256) void k_read(const float *buf, int words, int r, int *bad) {
    const int n = gridDim.x, other = (blockIdx.x + 97) % n;
    const float *theirs = buf + (size_t)other * words;
    float s = 0.f;
    for (int i = threadIdx.x; i < words; i += 256) s += theirs[i] - (float)(r * 1000 + other);
    if (s != 0.f) atomicAdd(bad, 1);
}

int main() {
    const int blocks = 256, rounds = 200;
    for (int kb : {1, 4, 43, 256}) {
        const int words = kb * 256;
        float *buf;
        Barrier *bar;
        int *bad;
        OK(hipMalloc(&buf, (size_t)blocks * words * 4));
        OK(hipMalloc(&bar, sizeof(Barrier)));
        OK(hipMalloc(&bad, 4));
        OK(hipMemset(bar, 0, sizeof(Barrier)));
        OK(hipMemset(bad, 0, 4));
        hipEvent_t e0, e1;
        OK(hipEventCreate(&e0));
        OK(hipEventCreate(&e1));
        hipStream_t st;
        OK(hipStreamCreate(&st));
        k_persistent<<<blocks, 256, 0, st>>>(buf, words, 5, bar, bad);
        OK(hipStreamSynchronize(st));
        OK(hipEventRecord(e0, st));
        k_persistent<<<blocks, 256, 0, st>>>(buf, words, rounds, bar, bad);
        OK(hipEventRecord(e1, st));
        OK(hipStreamSynchronize(st));
        float ms_p;
        OK(hipEventElapsedTime(&ms_p, e0, e1));
        // the same as 2 launches per round, replayed from a graph
        hipGraph_t g;
        hipGraphExec_t ge;
        OK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int r = 0; r < rounds; ++r) {
            k_write<<<blocks, 256, 0, st>>>(buf, words, r);
            k_read<<<blocks, 256, 0, st>>>(buf, words, r, bad);
        }
        OK(hipStreamEndCapture(st, &g));
        OK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        OK(hipGraphLaunch(ge, st));
        OK(hipStreamSynchronize(st));
        OK(hipEventRecord(e0, st));
        OK(hipGraphLaunch(ge, st));
        OK(hipEventRecord(e1, st));
        OK(hipStreamSynchronize(st));
        float ms_g;
        OK(hipEventElapsedTime(&ms_g, e0, e1));
        int h_bad = 0;
        OK(hipMemcpy(&h_bad, bad, 4, hipMemcpyDeviceToHost));
        std::printf("%4d KB per workgroup (%6.1f MB per round): persistent %.2f us per round (2 barriers) = %.2f us per barrier+phase; graph of 2 launches %.2f us per round "
                    "= %.2f us per launch; mismatches %d\n",
                    kb, blocks * words * 4 / 1e6, ms_p * 1e3 / rounds, ms_p * 1e3 / rounds / 2, ms_g * 1e3 / rounds, ms_g * 1e3 / rounds / 2, h_bad);
        OK(hipFree(buf));
        OK(hipFree(bar));
        OK(hipFree(bad));
    }
    return 0;
}
