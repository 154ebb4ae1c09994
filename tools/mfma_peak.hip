// Practical MFMA rate of the GPU this runs on: every wave issues independent
// v_mfma_f32_32x32x16_bf16 (or, SHAPE16, v_mfma_f32_16x16x32_bf16 over the same number of accumulator
// registers) back to back from registers, no memory traffic at all.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o build_exp/mfma_peak && build_exp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(4 * sizeof(float)))) float f32x4;

// the same FLOPs per iteration as mfma_loop<NACC>: 4 * NACC 16x16x32 blocks per 32x32x16 block... a 16x16x32
// MFMA is half the FLOPs of a 32x32x16 one, so two of them per 32x32x16
template <int NACC, bool RANDOM>
__global__ void __launch_bounds__(256) mfma_loop16(float* out, int iters) {
    f32x4 acc[2 * NACC];
    bf16x8 a, b;
    unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    for (int j = 0; j < 8; ++j) {
        if (RANDOM) {
            h = h * 1664525u + 1013904223u; a[j] = (__bf16)(((int)(h >> 8) % 2001 - 1000) * 1e-3f);
            h = h * 1664525u + 1013904223u; b[j] = (__bf16)(((int)(h >> 8) % 2001 - 1000) * 1e-3f);
        } else {
            a[j] = (__bf16)(threadIdx.x * 0.001f + j); b[j] = (__bf16)(1.0f - j * 0.01f);
        }
    }
    for (int i = 0; i < 2 * NACC; ++i)
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 2 * NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < 2 * NACC; ++i)
        for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, bool RANDOM>
__global__ void __launch_bounds__(256) mfma_loop(float* out, int iters) {
    f32x16 acc[NACC];
    bf16x8 a, b;
    unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
    for (int j = 0; j < 8; ++j) {
        if (RANDOM) {      // operands with random mantissas and signs: realistic switching activity
            h = h * 1664525u + 1013904223u; a[j] = (__bf16)(((int)(h >> 8) % 2001 - 1000) * 1e-3f);
            h = h * 1664525u + 1013904223u; b[j] = (__bf16)(((int)(h >> 8) % 2001 - 1000) * 1e-3f);
        } else {
            a[j] = (__bf16)(threadIdx.x * 0.001f + j); b[j] = (__bf16)(1.0f - j * 0.01f);
        }
    }
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, bool RANDOM, bool SHAPE16 = false>
void run(int blocks_per_cu, int iters) {
    const int blocks = 256 * blocks_per_cu;
    float* out;
    (void)hipMalloc(&out, sizeof(float) * blocks * 256);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    if (SHAPE16) mfma_loop16<NACC, RANDOM><<<blocks, 256>>>(out, 64);
    else mfma_loop<NACC, RANDOM><<<blocks, 256>>>(out, 64);
    (void)hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        if (SHAPE16) mfma_loop16<NACC, RANDOM><<<blocks, 256>>>(out, iters);
        else mfma_loop<NACC, RANDOM><<<blocks, 256>>>(out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double n_mfma = (double)blocks * 4 * iters * 4 * NACC;
        const double tf = n_mfma * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
        // cycles per MFMA per SIMD at a nominal 2.4 GHz
        const double cyc = ms * 1e-3 * 2.4e9 / (n_mfma / 1024.0);
        printf("%s %s acc=%d waves/SIMD=%d iters=%d: %8.3f ms  %7.1f TFLOP/s  (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", SHAPE16 ? "16x16x32" : "32x32x16", RANDOM ? "random  " : "constant", NACC, blocks_per_cu,
               iters, ms, tf, cyc);
    }
    (void)hipFree(out);
}

int main() {
    run<4, false>(2, 40000);
    run<4, true>(2, 40000);
    run<8, true>(1, 20000);
    run<4, true, true>(2, 40000);
    run<8, true, true>(1, 20000);
    run<4, false, true>(2, 40000);
    return 0;
}
