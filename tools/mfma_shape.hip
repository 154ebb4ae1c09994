// Power-limited MFMA rate by instruction shape with operands that CHANGE from one MFMA to the next (eight random
// register pairs in rotation), as they do in a convolution kernel: v_mfma_f32_32x32x16_bf16 over 4 accumulators of 16
// registers against v_mfma_f32_16x16x32_bf16 over 16 accumulators of 4 (the same output tile and FLOPs per round).
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_shape.hip -o build_exp/mfma_shape && build_exp/mfma_shape
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;

__device__ __forceinline__ void fill(bf16x8 (&a)[8], bf16x8 (&b)[8]) {
    unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    for (int q = 0; q < 8; ++q)
        for (int j = 0; j < 8; ++j) {
            h = h * 1664525u + 1013904223u; a[q][j] = (__bf16)(((int)(h >> 8) % 2001 - 1000) * 1e-3f);
            h = h * 1664525u + 1013904223u; b[q][j] = (__bf16)(((int)(h >> 8) % 2001 - 1000) * 1e-3f);
        }
}

// wave tile 32 x 128: one weight fragment a[q] against four pixel fragments b[..] per k16 step
__global__ void __launch_bounds__(256) loop32(float* out, int iters) {
    bf16x8 a[8], b[8];
    fill(a, b);
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[q], b[(q + i) & 7], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
// the same tile as 2 x 8 blocks of 16 x 16: two weight fragments against eight pixel fragments per k32 step
__global__ void __launch_bounds__(256) loop16(float* out, int iters) {
    bf16x8 a[8], b[8];
    fill(a, b);
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 4; ++q)          // four k32 steps = eight k16 steps of loop32
#pragma unroll
            for (int i = 0; i < 16; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(2 * q + (i >> 3)) & 7], b[(q + i) & 7], acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
void run(const char* name, K kernel, int blocks_per_cu, int iters, double flops_per_iter_per_wave) {
    const int blocks = 256 * blocks_per_cu;
    float* out;
    (void)hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    kernel<<<blocks, 256>>>(out, 64);
    (void)hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        kernel<<<blocks, 256>>>(out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double tf = (double)blocks * 4 * iters * flops_per_iter_per_wave / (ms * 1e-3) / 1e12;
        printf("%s waves/SIMD=%d: %8.3f ms  %7.1f TFLOP/s\n", name, blocks_per_cu, ms, tf);
    }
    (void)hipFree(out);
}

int main() {
    const double f32 = 32.0 * 2 * 32 * 32 * 16, f16 = 64.0 * 2 * 16 * 16 * 32;     // per iteration and wave: equal
    for (int w = 1; w <= 2; ++w) {
        run("32x32x16 rotating operands", loop32, w, 20000, f32);
        run("16x16x32 rotating operands", loop16, w, 20000, f16);
    }
    return 0;
}
