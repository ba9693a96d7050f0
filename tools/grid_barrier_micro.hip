// Developer measurement ("if a grid barrier costs more than it saves, commit the micro-benchmark that shows it"):
// what one dependent step of a batch-1 stage one costs (a) as its own kernel launch and (b) as a phase of ONE persistent kernel behind a
// grid-wide barrier. The step is the shape of the text encoder's small ops at 128 tokens: every block reads a 96 KB activation
// ([192][128] fp32) that ALL blocks of the previous step wrote, does a few hundred cycles of work, and writes its slice of the next one.
//   launches : N back-to-back launches of the step kernel on one stream (what the engine does today)
//   counter  : persistent kernel, one monotonic device-scope counter (release fence, arrive, relaxed sc1 poll, acquire fence)
//   xcd      : persistent kernel, XCD-hierarchical barrier (per-XCC counter -> leader -> top counter -> per-XCC generation)
// Build: hipcc --offload-arch=gfx950 -O3 tools/grid_barrier_micro.hip -o tools/bin/grid_barrier_micro ; run: tools/bin/grid_barrier_micro [blocks] [threads]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                                   \
    do {                                                                                        \
        hipError_t e_ = (x);                                                                    \
        if (e_ != hipSuccess) {                                                                 \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_));   \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

constexpr int ROWS = 192, COLS = 128, ELEMS = ROWS * COLS;

// one step: out[i] = in[(i * 97 + 13) % ELEMS] + 1 for the block's slice (every block reads lines written by many other blocks)
__device__ __forceinline__ void step_body(const float* in, float* out, int vb, int nb, int tid, int nt) {
    const int per = (ELEMS + nb - 1) / nb;
    const int beg = vb * per, end = min(beg + per, ELEMS);
    for (int i = beg + tid; i < end; i += nt) out[i] = in[(int)(((long long)i * 97 + 13) % ELEMS)] + 1.0f;
}

__global__ void step_kernel(const float* in, float* out) { step_body(in, out, blockIdx.x, gridDim.x, threadIdx.x, blockDim.x); }

struct Bar {
    unsigned* top;        // one counter
    unsigned* xcc_count;  // [8] arrivals per XCC (monotonic)
    unsigned* xcc_gen;    // [8] generation per XCC
    unsigned* xcc_size;   // [8] blocks per XCC (census)
    unsigned* nxcc;       // XCCs that hold at least one block
};

__device__ __forceinline__ unsigned ld_relaxed(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// single monotonic counter
__device__ __forceinline__ void barrier_counter(unsigned* ctr, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (ld_relaxed(ctr) < target) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

// XCD-hierarchical: arrivals on the block's XCC counter; the last arriver of an XCC (its leader for this generation) releases, arrives at the
// top counter, waits for every XCC, acquires and bumps the XCC's generation; everybody else polls the generation of its own XCC
__device__ __forceinline__ void barrier_xcd(const Bar& b, unsigned xcc, unsigned gen /* 1, 2, ... */) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned size = b.xcc_size[xcc];
        const unsigned prev = __hip_atomic_fetch_add(b.xcc_count + xcc * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (prev + 1 == size * gen) {
            __hip_atomic_fetch_add(b.top, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned want = *b.nxcc * gen;
            while (ld_relaxed(b.top) < want) __builtin_amdgcn_s_sleep(1);
            __hip_atomic_store(b.xcc_gen + xcc * 32, gen, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            while (ld_relaxed(b.xcc_gen + xcc * 32) < gen) __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
}

template <int KIND>  // 0 counter, 1 xcd
__global__ void persistent_kernel(float* a, float* b, int steps, Bar bar, unsigned base_gen) {
    unsigned xcc = 0;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7;
    const float* in = a;
    float* out = b;
    for (int s = 0; s < steps; ++s) {
        step_body(in, out, blockIdx.x, gridDim.x, threadIdx.x, blockDim.x);
        if (KIND == 0) barrier_counter(bar.top, (base_gen + s + 1) * gridDim.x);
        else barrier_xcd(bar, xcc, base_gen + s + 1);
        float* t = const_cast<float*>(in);
        in = out;
        out = t;
    }
}

__global__ void census_kernel(Bar bar) {
    if (threadIdx.x == 0) {
        unsigned xcc = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        atomicAdd(bar.xcc_size + (xcc & 7), 1u);
    }
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 256, threads = argc > 2 ? atoi(argv[2]) : 256;
    const int steps = 200, reps = 20;
    float *a, *b;
    CK(hipMalloc(&a, ELEMS * 4));
    CK(hipMalloc(&b, ELEMS * 4));
    CK(hipMemset(a, 0, ELEMS * 4));
    unsigned* mem;
    CK(hipMalloc(&mem, 4096 * 4));
    CK(hipMemset(mem, 0, 4096 * 4));
    Bar bar{mem, mem + 64, mem + 64 + 256, mem + 1024, mem + 1040};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto check = [&](const char* what, int total_steps) {
        std::vector<float> h(ELEMS);
        CK(hipMemcpy(h.data(), (total_steps & 1) ? b : a, ELEMS * 4, hipMemcpyDeviceToHost));
        long bad = 0;
        for (int i = 0; i < ELEMS; ++i) bad += h[i] != (float)total_steps;
        printf("  %-10s result check: %ld of %d elements wrong after %d steps\n", what, bad, ELEMS, total_steps);
    };
    // (a) launches
    {
        float best = 1e9f;
        for (int r = 0; r < reps; ++r) {
            CK(hipMemset(a, 0, ELEMS * 4));
            CK(hipEventRecord(e0));
            for (int s = 0; s < steps; ++s) hipLaunchKernelGGL(step_kernel, dim3(blocks), dim3(threads), 0, 0, (s & 1) ? b : a, (s & 1) ? a : b);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        check("launches", steps);
        printf("launches : %7.2f us per dependent step (%d blocks x %d threads, best of %d x %d steps)\n", 1e3 * best / steps, blocks, threads, reps, steps);
    }
    // census: blocks per XCC for this grid (the same grid shape is dispatched the same way: block b -> XCC b % 8, observed, not promised)
    hipLaunchKernelGGL(census_kernel, dim3(blocks), dim3(threads), 0, 0, bar);
    CK(hipDeviceSynchronize());
    unsigned hs[8], n = 0;
    CK(hipMemcpy(hs, bar.xcc_size, 32, hipMemcpyDeviceToHost));
    for (int i = 0; i < 8; ++i) n += hs[i] != 0;
    CK(hipMemcpy(bar.nxcc, &n, 4, hipMemcpyHostToDevice));
    printf("census: blocks per XCC %u %u %u %u %u %u %u %u\n", hs[0], hs[1], hs[2], hs[3], hs[4], hs[5], hs[6], hs[7]);
    for (int kind = 0; kind < 2; ++kind) {
        float best = 1e9f;
        unsigned gen = 0;
        CK(hipMemset(mem, 0, 1024 * 4));
        for (int r = 0; r < reps; ++r) {
            CK(hipMemset(a, 0, ELEMS * 4));
            CK(hipEventRecord(e0));
            if (kind == 0) hipLaunchKernelGGL(persistent_kernel<0>, dim3(blocks), dim3(threads), 0, 0, a, b, steps, bar, gen);
            else hipLaunchKernelGGL(persistent_kernel<1>, dim3(blocks), dim3(threads), 0, 0, a, b, steps, bar, gen);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            gen += steps;
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        check(kind == 0 ? "counter" : "xcd", steps);
        printf("%-9s: %7.2f us per dependent step inside ONE persistent launch\n", kind == 0 ? "counter" : "xcd", 1e3 * best / steps);
    }
    return 0;
}
