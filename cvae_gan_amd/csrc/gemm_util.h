// Helpers shared by the gather-based and the patch-staged GEMM kernels.
#pragma once
#include "common.h"

// LDS tiles keep the natural [pixel][channel] order; the MFMA operands need
// [channel][8 consecutive pixels], i.e. a transpose.  bf16 uses the hardware
// transposing read ds_read_b64_tr_b16: per 16-lane group it fetches a 4-pixel x
// 16-channel block (lane 4q+p supplies the address of pixel q, channels 4p..4p+3)
// and hands lane i channel i of the four pixels.  f32 reads single dwords.
template <typename T> struct TrLoad;
template <> struct TrLoad<bf16> {
    __device__ static __forceinline__ bf16x8 ld(const char* base, int pitch, int m0, int c0, int lane) {
        const int i = lane & 15, q = i >> 2, p = i & 3, blk = (lane >> 4) & 1, h = lane >> 5;
        const char* ptr = base + (m0 + 8 * h + q) * pitch + (c0 + 16 * blk + 4 * p) * 2;
        typedef __attribute__((address_space(3))) bf16x4* lds_ptr;
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)ptr);
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(ptr + 4 * pitch));
        bf16x8 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) { r[j] = lo[j]; r[4 + j] = hi[j]; }
        return r;
    }
};
template <> struct TrLoad<float> {
    __device__ static __forceinline__ f32x8 ld(const char* base, int pitch, int m0, int c0, int lane) {
        const int l31 = lane & 31, h = lane >> 5;
        f32x8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = *(const float*)(base + (m0 + 2 * j + h) * pitch + (c0 + l31) * 4);
        return r;
    }
};
// For T = float the slot (h, j) holds pixel 2j+h, for bf16 pixel 8h+j; both operands
// of one MFMA use the same loader, so the products pair up correctly.

// ---- canonical <-> GEMM layout map ---------------------------------------------
__device__ __forceinline__ int wmap_index(const cvg_wmap& m, int n, int k) {
    if (n >= m.n_valid) return -1;
    const int j = k / m.kd, c = k - j * m.kd;
    if (c >= m.kc_valid || j >= m.ntaps) return -1;
    int t = j;
    if (m.tap_mode == 1) t = m.ntaps - 1 - j;
    else if (m.tap_mode == 2) t = m.tap_table[j];
    if (t < 0) return -1;
    const int nq = n / m.nd, nr = n - nq * m.nd;
    if (m.nr_valid > 0 && nr >= m.nr_valid) return -1;
    return nq * m.na + nr * m.nb + t * m.ka + c * m.kb;
}


// wmap_index for R consecutive k of one row: the two integer divisions and the tap lookup are done
// once when the run stays inside one tap (it almost always does: runs start at multiples of R and
// the channel block kd is a multiple of 4 or 8), instead of once per element.
template <int R>
__device__ __forceinline__ void wmap_index_run(const cvg_wmap& m, int n, int k, int (&ci)[R]) {
    const int j = k / m.kd, c = k - j * m.kd;
    if (c + R > m.kd) {
#pragma unroll
        for (int e = 0; e < R; ++e) ci[e] = wmap_index(m, n, k + e);
        return;
    }
    int t = j;
    bool ok = n < m.n_valid && j < m.ntaps;
    if (ok) {
        if (m.tap_mode == 1) t = m.ntaps - 1 - j;
        else if (m.tap_mode == 2) t = m.tap_table[j];
        ok = t >= 0;
    }
    const int nq = n / m.nd, nr = n - nq * m.nd;
    if (m.nr_valid > 0 && nr >= m.nr_valid) ok = false;
    const int base = nq * m.na + nr * m.nb + t * m.ka;
#pragma unroll
    for (int e = 0; e < R; ++e) ci[e] = (ok && c + e < m.kc_valid) ? base + (c + e) * m.kb : -1;
}

// partial slabs [nsplit][n_rows][k_ld] -> canonical gradient (fixed summation order).  One thread
// per 4 consecutive k (k_ld is a multiple of 4): 16-byte slab reads, four slabs in flight.
static __global__ void __launch_bounds__(256) wgrad_reduce(const float* __restrict__ ws, float* __restrict__ grad,
                                                           const cvg_wmap map, int n_rows, int k_ld, int nsplit,
                                                           int accumulate) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int k4 = k_ld >> 2;
    if (idx >= (int64_t)map.n_valid * k4) return;
    const int n = (int)(idx / k4), k = (int)(idx % k4) * 4;
    int ci[4];
    bool any = false;
    wmap_index_run<4>(map, n, k, ci);
#pragma unroll
    for (int e = 0; e < 4; ++e) any |= ci[e] >= 0;
    if (!any) return;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    const size_t stride = (size_t)n_rows * k_ld;
    const float* p = ws + (size_t)n * k_ld + k;
    int z = 0;
    for (; z + 4 <= nsplit; z += 4) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *(const f32x4*)(p + (size_t)(z + u) * stride);
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] += v[u][e];
    }
    for (; z < nsplit; ++z) {
        const f32x4 v = *(const f32x4*)(p + (size_t)z * stride);
#pragma unroll
        for (int e = 0; e < 4; ++e) s[e] += v[e];
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (ci[e] >= 0) grad[ci[e]] = accumulate ? grad[ci[e]] + s[e] : s[e];
}
