// Diagnostic probe (GPU box): cycles per k-step of the deep-layer consumer loop shape - 4 x ds_read_b128 + 4 x MFMA 32x32x16 bf16
// per wave, 4 or 8 waves per workgroup, one workgroup per CU - with the reads on / off and padded / unpadded rows.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/mfma_lds_probe tools/diag/mfma_lds_probe.hip && gpurun_out/mfma_lds_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int MODE>   // 0: MFMA only; 1: reads (pitch 80) + MFMA, 1-deep prefetch; 2: reads only; 3: like 1 but pitch 64 (conflicts)
__global__ __launch_bounds__(512) void probe(float* out, long long* cyc, int iters, int nwaves) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    for (int i = tid; i < 140000 / 4; i += blockDim.x) reinterpret_cast<float*>(smem)[i] = 0.001f * (i & 255);
    __syncthreads();
    if (wave >= nwaves) return;
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int i = 0; i < 16; ++i) acc[a][b][i] = 0.f;
    constexpr int PITCH = MODE == 3 ? 64 : 80;
    const int wm = wave & 1, wn = (wave >> 1) & 1;
    // A: pixel = wm*64 + mt*32 + r as 8x16 tile, stride-2 rows in a 40-pixel-per-row patch; B: [k/8][128 n][16 B]
    const char* pa = smem + ((wm * 4 * 2 * 40 + (r >> 4) * 2 * 40 + (r & 15)) * PITCH) + h * 16;
    const char* pb = smem + 60000 + (h * 128 + wn * 64 + r) * 16;
    bf16x8 af[2][2], bf[2][2];
    auto load = [&](int st, int buf) __attribute__((always_inline)) {
        const int kx = st >> 1, ks = st & 1;
        for (int mt = 0; mt < 2; ++mt) af[buf][mt] = *reinterpret_cast<const bf16x8*>(pa + mt * 2 * 2 * 40 * PITCH + kx * PITCH + ks * 32);
        for (int nt = 0; nt < 2; ++nt) bf[buf][nt] = *reinterpret_cast<const bf16x8*>(pb + nt * 512 + (kx * 4 + ks * 2) * 2048);
    };
    load(0, 0); load(1, 1);
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1 || MODE == 3) load(0, 0);
#pragma unroll
        for (int st = 0; st < 6; ++st) {
            if ((MODE == 1 || MODE == 3) && st + 1 < 6) load(st + 1, (st + 1) & 1);
            if (MODE == 2) { load(st, st & 1); asm volatile("" :: "v"(af[st & 1][0]), "v"(af[st & 1][1]), "v"(bf[st & 1][0]), "v"(bf[st & 1][1])); }
            __builtin_amdgcn_sched_barrier(0);
            if (MODE != 2)
                for (int nt = 0; nt < 2; ++nt) for (int mt = 0; mt < 2; ++mt)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[st & 1][mt], bf[st & 1][nt], acc[mt][nt], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = clock64();
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
    for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int i = 0; i < 16; ++i) out[(((size_t)blockIdx.x * 8 + wave) * 64 + (a * 2 + b) * 16 + i) * 64 + lane] = acc[a][b][i];
}

template <int MODE> void run(const char* name, int nwaves, int grid) {
    float* out; long long* cyc;
    hipMalloc(&out, (size_t)grid * 8 * 64 * 64 * 4); hipMalloc(&cyc, grid * 8 * 8); hipMemset(cyc, 0, grid * 8 * 8);
    const int iters = 200;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 150000);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(512), 150000, 0, out, cyc, iters, nwaves);
    hipDeviceSynchronize();
    std::vector<long long> h(grid * 8); hipMemcpy(h.data(), cyc, grid * 8 * 8, hipMemcpyDeviceToHost);
    double s = 0; int n = 0; for (int b = 0; b < grid; ++b) for (int w = 0; w < nwaves; ++w) { s += h[b * 8 + w]; ++n; }
    printf("%-44s waves %d grid %3d: %.1f cycles per k-step (4 reads + 4 MFMA = 128 MFMA cycles)\n", name, nwaves, grid, s / n / iters / 6);
    hipFree(out); hipFree(cyc);
}
int main() {
    for (int grid : {1, 256}) {
        run<0>("MFMA only", 4, grid); run<1>("reads pitch 80 + MFMA", 4, grid); run<3>("reads pitch 64 + MFMA", 4, grid); run<2>("reads only pitch 80", 4, grid);
        run<0>("MFMA only", 8, grid); run<1>("reads pitch 80 + MFMA", 8, grid);
    }
    return 0;
}
