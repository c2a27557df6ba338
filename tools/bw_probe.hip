// Streaming-bandwidth probe for gfx950: what do the access structures used by the VAE kernels reach?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void copy_stride(const f32x4* __restrict__ in, f32x4* __restrict__ out, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = in[i];
}
// persistent workgroups, 16 KiB tiles, next tile prefetched into registers, staged through LDS with 2 barriers
template <int NPRE>
__global__ __launch_bounds__(256) void copy_tiles(const f32x4* __restrict__ in, f32x4* __restrict__ out, int n_tiles) {
    __shared__ f32x4 lds[256 * NPRE];
    f32x4 pre[NPRE];
    int tile = blockIdx.x;
    if (tile < n_tiles) for (int u = 0; u < NPRE; ++u) pre[u] = in[(size_t)tile * 256 * NPRE + u * 256 + threadIdx.x];
    for (; tile < n_tiles; tile += gridDim.x) {
        __syncthreads();
        for (int u = 0; u < NPRE; ++u) lds[u * 256 + threadIdx.x] = pre[u];
        __syncthreads();
        const int nt = tile + gridDim.x;
        if (nt < n_tiles) for (int u = 0; u < NPRE; ++u) pre[u] = in[(size_t)nt * 256 * NPRE + u * 256 + threadIdx.x];
        for (int u = 0; u < NPRE; ++u) out[(size_t)tile * 256 * NPRE + u * 256 + threadIdx.x] = lds[u * 256 + (threadIdx.x ^ 1)];
    }
}
__global__ void read_only(const f32x4* __restrict__ in, float* out, long n) {
    f32x4 acc = {0, 0, 0, 0};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) acc += in[i];
    if (acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) out[0] = 1;
}
__global__ void write_only(f32x4* __restrict__ out, long n) {
    const f32x4 v = {1.f, 2.f, 3.f, 4.f};
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = v;
}
// 1 read : 4 writes (the shape of the 32-channel transposed-conv forward: 67 MB in, 268 MB out)
__global__ void read1_write4(const f32x4* __restrict__ in, f32x4* __restrict__ out, long n_in) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n_in; i += (long)gridDim.x * blockDim.x) {
        const f32x4 v = in[i];
        const long blk = i >> 6, l = i & 63;
#pragma unroll
        for (int k = 0; k < 4; ++k) out[(blk * 4 + k) * 64 + l] = v;
    }
}
int main() {
    const size_t bytes = 1ull << 30; const long n = bytes / 16;
    f32x4 *a, *b; hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char* name, auto launch, double moved) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); for (int i = 0; i < 10; ++i) launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %8.1f GB/s\n", name, moved * 10 / ms / 1e6);
    };
    for (int g : {1024, 2048, 4096, 16384})
        timeit(("copy grid-stride grid=" + std::to_string(g)).c_str(), [&] { hipLaunchKernelGGL(copy_stride, dim3(g), dim3(256), 0, 0, a, b, n); }, 2.0 * bytes);
    for (int g : {512, 1024, 2048})
        timeit(("read-only grid=" + std::to_string(g)).c_str(), [&] { hipLaunchKernelGGL(read_only, dim3(g), dim3(256), 0, 0, a, (float*)b, n); }, 1.0 * bytes);
    for (int g : {512, 1024, 2048, 8192})
        timeit(("write-only grid=" + std::to_string(g)).c_str(), [&] { hipLaunchKernelGGL(write_only, dim3(g), dim3(256), 0, 0, b, n); }, 1.0 * bytes);
    for (int g : {512, 1024, 2048, 8192})
        timeit(("read1:write4 grid=" + std::to_string(g)).c_str(), [&] { hipLaunchKernelGGL(read1_write4, dim3(g), dim3(256), 0, 0, a, b, n / 4); }, 1.25 * bytes);
    for (int g : {256, 512, 1024, 2048}) {
        timeit(("tile copy 16KiB prefetch1 grid=" + std::to_string(g)).c_str(), [&] { hipLaunchKernelGGL(copy_tiles<4>, dim3(g), dim3(256), 0, 0, a, b, (int)(n / 1024)); }, 2.0 * bytes);
        timeit(("tile copy 32KiB prefetch1 grid=" + std::to_string(g)).c_str(), [&] { hipLaunchKernelGGL(copy_tiles<8>, dim3(g), dim3(256), 0, 0, a, b, (int)(n / 2048)); }, 2.0 * bytes);
    }
    return 0;
}
