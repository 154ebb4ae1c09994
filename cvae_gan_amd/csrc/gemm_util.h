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

// One staged 16-byte piece times LeakyReLU'(g) per element (g > 0 ? 1 : slope), rounded once more to T.
template <typename T> __device__ __forceinline__ u32x4 mul_gate(u32x4 v, u32x4 g);
template <> __device__ __forceinline__ u32x4 mul_gate<float>(u32x4 v, u32x4 g) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (!(__uint_as_float(g[j]) > 0.f)) v[j] = __float_as_uint(__uint_as_float(v[j]) * LRELU_SLOPE);
    return v;
}
template <> __device__ __forceinline__ u32x4 mul_gate<bf16>(u32x4 v, u32x4 g) {
    bf16x8 vb = __builtin_bit_cast(bf16x8, v);
    const bf16x8 gb = __builtin_bit_cast(bf16x8, g);
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if (!((float)gb[j] > 0.f)) vb[j] = (__bf16)((float)vb[j] * LRELU_SLOPE);
    return __builtin_bit_cast(u32x4, vb);
}

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
// per 4 consecutive k (k_ld is a multiple of 4) and slab range: 16-byte slab reads, four slabs in
// flight.  zg (1, 2, 4 or 8, chosen by the launcher) slab ranges per output share a workgroup and
// meet in LDS in range order: small layers (a 64 -> 64 conv has 9216 such outputs) otherwise leave
// most CUs idle while a few dozen workgroups walk hundreds of slabs each.
static __global__ void __launch_bounds__(256) wgrad_reduce(const float* __restrict__ ws, float* __restrict__ grad,
                                                           const cvg_wmap map, int n_rows, int k_ld, int nsplit,
                                                           int accumulate, int zg) {
    __shared__ f32x4 part[256];
    const int lo_n = 256 / zg, lo = threadIdx.x % lo_n, zi = threadIdx.x / lo_n;
    const int64_t idx = (int64_t)blockIdx.x * lo_n + lo;
    const int k4 = k_ld >> 2;
    const bool live = idx < (int64_t)map.n_valid * k4;
    const int n = live ? (int)(idx / k4) : 0, k = live ? (int)(idx % k4) * 4 : 0;
    int ci[4] = {-1, -1, -1, -1};
    bool any = false;
    if (live) {
        wmap_index_run<4>(map, n, k, ci);
#pragma unroll
        for (int e = 0; e < 4; ++e) any |= ci[e] >= 0;
    }
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    if (any) {
        const size_t stride = (size_t)n_rows * k_ld;
        const float* p = ws + (size_t)n * k_ld + k;
        const int per = (nsplit + zg - 1) / zg;
        int z = zi * per;
        const int z_end = min(nsplit, z + per);
        for (; z + 4 <= z_end; z += 4) {
            f32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *(const f32x4*)(p + (size_t)(z + u) * stride);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) s[e] += v[u][e];
        }
        for (; z < z_end; ++z) {
            const f32x4 v = *(const f32x4*)(p + (size_t)z * stride);
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] += v[e];
        }
    }
    if (zg > 1) {
        part[threadIdx.x] = f32x4{s[0], s[1], s[2], s[3]};
        __syncthreads();
        if (zi != 0) return;
        for (int q = 1; q < zg; ++q) {
            const f32x4 v = part[q * lo_n + lo];
#pragma unroll
            for (int e = 0; e < 4; ++e) s[e] += v[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
        if (ci[e] >= 0) grad[ci[e]] = accumulate ? grad[ci[e]] + s[e] : s[e];
}
// slab ranges per output for wgrad_reduce: enough threads for the whole GPU, at least 4 slabs per range
static inline int wgrad_reduce_zg(int64_t outputs4, int nsplit) {
    int zg = 1;
    while (zg < 8 && outputs4 * zg < 131072 && nsplit >= 8 * zg) zg *= 2;
    return zg;
}

// The same for the layouts whose canonical tensor is contiguous along the NINE taps of one (n, c) pair: convolution
// weights W[n][c][3][3] reduced from slabs [n][tap * kd + c] (pattern A) or from slabs [tap * nd + n][c] (pattern B,
// the one-tap GEMMs of the low-resolution up-convolution gradients).  wgrad_reduce gives every thread four
// consecutive k, i.e. four c, and stores them 36 bytes apart: one 4-byte store per 64-byte segment, which for the
// 2.4 M-element layers costs more than reading the slabs.  Here a workgroup takes 64 consecutive c of a few n: each
// tap's slab values are read coalesced along c, the sums meet in LDS, and the 64 * 9 canonical floats of an n leave
// as one contiguous run.  zg (1, 2 or 4) slab ranges per pair share the workgroup and are added in range order.
template <int T>
static __global__ void __launch_bounds__(256) wgrad_reduce_taps(const float* __restrict__ ws, float* __restrict__ grad,
                                                                int n_valid, int c_valid, int RN, size_t RT, size_t zstride,
                                                                int SN, int nsplit, int accumulate, int zg) {
    __shared__ float sm[4][64 * T];
    const int cl = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const int nb = 4 / zg, nl = grp / zg, zi = grp - nl * zg;
    const int cblocks = (c_valid + 63) >> 6;
    const int nblk = blockIdx.x / cblocks, cb = blockIdx.x - nblk * cblocks;
    const int n = nblk * nb + nl, c = cb * 64 + cl;
    float s[T];
#pragma unroll
    for (int t = 0; t < T; ++t) s[t] = 0.f;
    if (n < n_valid && c < c_valid) {
        const float* p = ws + (size_t)n * RN + c;
        const int per = (nsplit + zg - 1) / zg;
        int z = zi * per;
        const int z_end = min(nsplit, z + per);
        for (; z + 2 <= z_end; z += 2) {           // two slabs (18 loads) in flight, added in slab order
            float v[2][T];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int t = 0; t < T; ++t) v[u][t] = p[(size_t)(z + u) * zstride + t * RT];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int t = 0; t < T; ++t) s[t] += v[u][t];
        }
        if (z < z_end) {
#pragma unroll
            for (int t = 0; t < T; ++t) s[t] += p[(size_t)z * zstride + t * RT];
        }
    }
#pragma unroll
    for (int t = 0; t < T; ++t) sm[grp][cl * T + t] = s[t];
    __syncthreads();
    for (int i = threadIdx.x; i < nb * 64 * T; i += 256) {
        const int nl_ = i / (64 * T), j = i - nl_ * (64 * T);
        const int n_ = nblk * nb + nl_, c_ = cb * 64 + j / T;
        if (n_ >= n_valid || c_ >= c_valid) continue;
        float v = 0.f;
        for (int q = 0; q < zg; ++q) v += sm[nl_ * zg + q][j];
        const size_t idx = (size_t)n_ * SN + (size_t)cb * 64 * T + j;
        grad[idx] = accumulate ? grad[idx] + v : v;
    }
}

// slabs -> canonical gradient with whichever of the two kernels fits the layout
static inline void launch_wgrad_reduce(const float* ws, float* grad, const cvg_wmap* map, int n_rows, int k_ld, int nsplit,
                                       int accumulate, hipStream_t s) {
    const bool pat_a = map->nd == 1 && map->tap_mode == 0 && map->ka == 1 && map->kb == 9 && map->ntaps == 9 &&
                       map->nr_valid == 0 && map->nb == 0;
    const bool pat_b = map->ntaps == 1 && map->na == 1 && map->ka == 0 && map->kb == 9 && map->nr_valid == 0 &&
                       map->tap_mode == 0 && map->n_valid == 9 * map->nd;
    if (pat_a || pat_b) {
        const int n_valid = pat_a ? map->n_valid : map->nd, c_valid = map->kc_valid;
        const size_t RT = pat_a ? (size_t)map->kd : (size_t)map->nd * k_ld;
        const int SN = pat_a ? map->na : map->nb;
        const int64_t pairs = (int64_t)n_valid * c_valid;
        int zg = 1;
        while (zg < 4 && pairs * zg < 131072 && nsplit >= 8 * zg) zg *= 2;
        const int nb = 4 / zg;
        const int64_t blocks = (int64_t)ceil_div(n_valid, nb) * ceil_div(c_valid, 64);
        wgrad_reduce_taps<9><<<(int)blocks, 256, 0, s>>>(ws, grad, n_valid, c_valid, k_ld, RT, (size_t)n_rows * k_ld, SN, nsplit,
                                                         accumulate, zg);
        return;
    }
    const int64_t n = (int64_t)map->n_valid * (k_ld / 4);
    const int zg = wgrad_reduce_zg(n, nsplit);
    wgrad_reduce<<<ceil_div(n, 256 / zg), 256, 0, s>>>(ws, grad, *map, n_rows, k_ld, nsplit, accumulate, zg);
}
