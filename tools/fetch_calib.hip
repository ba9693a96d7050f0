// fetch_calib.hip — calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md §HBM: "calibrate on a known
// byte count in your own access pattern"). Pure streaming kernels over a 1 GiB buffer (4x the 256 MiB Infinity Cache), one
// launch each, in the three load forms the conv kernels use:
//   calib_read_dword    : global_load_dword, 64 lanes = 256 B coalesced   (the register-staged kernels)
//   calib_read_dwordx4  : global_load_dwordx4, 16 B per lane
//   calib_read_lds_dma  : buffer_load_dwordx4 ... lds (LDS-DMA, 1 KiB per wave instruction: the producer wave)
//   calib_copy_dwordx4  : dwordx4 read + dwordx4 write (WRITE_SIZE)
//   calib_write_dword   : dword write only
// Usage: rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out -- tools/bin/fetch_calib   (and a second pass with WRITE_SIZE);
// tools/fetch_calib_reduce.py turns the two CSVs into factors = known bytes / reported bytes.
#include <hip/hip_runtime.h>

#include <cstdio>

constexpr size_t N = (size_t)1 << 28;  // floats = 1 GiB

__global__ void calib_read_dword(const float* __restrict__ x, float* sink, size_t n) {
    float s = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += x[i];
    if (s == 12345.678f) sink[threadIdx.x] = s;
}
__global__ void calib_read_dwordx4(const float4* __restrict__ x, float* sink, size_t n4) {
    float s = 0.f;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = x[i];
        s += v.x + v.y + v.z + v.w;
    }
    if (s == 12345.678f) sink[threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void calib_read_lds_dma(const float* x, float* sink, size_t n) {
    __shared__ __attribute__((aligned(16))) float tile[4 * 256 * 4];  // 4 waves x 1 KiB x 4 pieces in flight
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, 0x7fffffff, 0x00020000);
    float s = 0.f;
    // every wave streams its own 4 KiB slices: piece p of slice k covers bytes [k*4096 + p*1024, +1024)
    const size_t slices = n * 4 / 4096;
    for (size_t k = (size_t)blockIdx.x * 4 + wid; k < slices; k += (size_t)gridDim.x * 4) {
        // (offsets are 32-bit in the buffer form: split base into the descriptor-free scalar offset)
        const unsigned long long base = k * 4096ull;
        const __amdgpu_buffer_rsrc_t r2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(x)) + base, 0, 4096, 0x00020000);
#pragma unroll
        for (int p = 0; p < 4; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r2, (__attribute__((address_space(3))) void*)(tile + (wid * 4 + p) * 256), 16, lane * 16, p * 1024, 0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s += tile[(wid * 4) * 256 + lane];
    }
    (void)rs;
    if (s == 12345.678f) sink[threadIdx.x] = s;
}
__global__ void calib_copy_dwordx4(const float4* __restrict__ x, float4* __restrict__ y, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) y[i] = x[i];
}
__global__ void calib_write_dword(float* __restrict__ y, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = 1.0f;
}

#define CK(e)                                                                 \
    do {                                                                      \
        hipError_t e_ = (e);                                                  \
        if (e_ != hipSuccess) {                                               \
            std::printf("%s: %s\n", #e, hipGetErrorString(e_));               \
            return 1;                                                         \
        }                                                                     \
    } while (0)

int main() {
    float *x, *y, *sink;
    CK(hipMalloc(&x, N * 4));
    CK(hipMalloc(&y, N * 4));
    CK(hipMalloc(&sink, 4096));
    CK(hipMemset(x, 0, N * 4));
    CK(hipMemset(y, 0, N * 4));
    CK(hipDeviceSynchronize());
    const int grid = 256 * 16;
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    auto report = [&](const char* name, double bytes) {
        float ms = 0;
        hipEventSynchronize(b);
        hipEventElapsedTime(&ms, a, b);
        std::printf("%-22s %8.3f ms  %7.1f GB/s  (%.0f bytes known)\n", name, ms, bytes / ms / 1e6, bytes);
    };
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(calib_read_dword, dim3(grid), dim3(256), 0, 0, x, sink, N);
        hipEventRecord(b);
        report("calib_read_dword", N * 4.0);
        hipEventRecord(a);
        hipLaunchKernelGGL(calib_read_dwordx4, dim3(grid), dim3(256), 0, 0, (const float4*)x, sink, N / 4);
        hipEventRecord(b);
        report("calib_read_dwordx4", N * 4.0);
        hipEventRecord(a);
        hipLaunchKernelGGL(calib_read_lds_dma, dim3(grid), dim3(256), 0, 0, x, sink, N);
        hipEventRecord(b);
        report("calib_read_lds_dma", N * 4.0);
        hipEventRecord(a);
        hipLaunchKernelGGL(calib_copy_dwordx4, dim3(grid), dim3(256), 0, 0, (const float4*)x, (float4*)y, N / 4);
        hipEventRecord(b);
        report("calib_copy_dwordx4", N * 8.0);
        hipEventRecord(a);
        hipLaunchKernelGGL(calib_write_dword, dim3(grid), dim3(256), 0, 0, y, N);
        hipEventRecord(b);
        report("calib_write_dword", N * 4.0);
    }
    CK(hipDeviceSynchronize());
    return 0;
}
