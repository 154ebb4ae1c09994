// Shared device/host helpers for the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/cvaegan.h"

typedef __bf16 bf16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) float f32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define LRELU_SLOPE 0.2f
#define BN_EPS 1e-5f
#define BN_MOMENTUM 0.1f

// ---- error reporting -----------------------------------------------------------
void cvg_set_error(const char* fmt, ...);
#define CVG_FAIL(...) do { cvg_set_error(__VA_ARGS__); return -1; } while (0)
#define CVG_CHECK(cond, ...) do { if (!(cond)) CVG_FAIL(__VA_ARGS__); } while (0)
#define CVG_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); \
    if (e_ != hipSuccess) CVG_FAIL("%s:%d launch failed: %s", __FILE__, __LINE__, hipGetErrorString(e_)); } while (0)

// ---- benchmarking switches (environment variables, read once per process; none changes results) ----
struct CvgSwitches {
    bool no_patch;            // CVG_NO_PATCH: every convolution on the gather kernels
    bool no_xcd_remap;        // CVG_NO_XCD_REMAP: weight-gradient workgroups in launch order
    bool patch_2w;            // CVG_PATCH_2W: two-wave workgroups for 64-channel patch_fwd layers
    bool wgrad_s2_gather;     // CVG_WGRAD_S2_GATHER: stride-2 weight gradients on the gather kernel
    bool wgrad_v1;            // CVG_WGRAD_V1: tap-split weight-gradient kernel instead of the nine-tap one
    bool wgrad_4w;            // CVG_WGRAD_4W: four-wave nine-tap weight-gradient workgroups
    bool no_res_rows;         // CVG_NO_RES_ROWS: the tail's forward pass on the plain four-wave kernel (weights through the ring, one tile per workgroup)
    bool no_walk64;           // CVG_NO_WALK64: 64-channel 3x3 layers never walk (one tile per workgroup)
    bool no_lowk;             // CVG_NO_LOWK: the stem and the tail's input gradient on patch_fwd instead of the resident seven-row-tap kernel
    bool wgrad_wide;          // CVG_WGRAD_WIDE: 128 x 256 output tiles in the gather weight-gradient kernel
    bool bn_raw_sums;         // CVG_BN_RAW_SUMS: BatchNorm sums about 0 instead of about row 0
    bool split_finalize;      // CVG_SPLIT_FINALIZE: column reductions finish in a second kernel instead of in the same launch
    char patch_wide;          // CVG_PATCH_WIDE: '0' no eight-wave tiles, '1' 64-channel wave tiles
    char patch_persist;       // CVG_PATCH_PERSIST: '0' no walking workgroups, '1' on every eligible layer, '2' the
                              // walking kernel with one image group per workgroup, '3' the 64-channel layers walk
                              // whatever the descriptor says (measured beside weight gradients: 6.21 against 6.19 ms
                              // per c3 step); unset: the descriptor's flag
    char patch_perm;          // CVG_PATCH_PERM: '0' natural lane order of the LDS reads, '1' no paired rows in 8 x 8 tiles
    int lds_extra;            // CVG_PATCH_LDS_EXTRA: bytes of LDS added per workgroup (occupancy experiments)
    int wgrad_wgs;            // CVG_WGRAD_WGS: workgroups a weight-gradient launch aims for
    bool tile_set; int tile[3];   // CVG_PATCH_TILE=th,tw,tb
};
const CvgSwitches& cvg_switches();

// ---- exact division by a runtime constant ---------------------------------------
// q = floor(m / d) for 0 <= m < 2^31 with one v_mul_hi_u32 and a shift.
// mul = ceil(2^(31+l) / d), l = ceil(log2 d); d == 1 is encoded as mul == 0.
struct FastDiv {
    uint32_t mul, sh, d, pad;
};
static inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d; f.pad = 0;
    if (d <= 1) { f.mul = 0; f.sh = 0; return f; }
    uint32_t l = 0;
    while ((1u << l) < d) ++l;
    uint64_t num = 1ull << (31 + l);
    f.mul = (uint32_t)((num + d - 1) / d);
    f.sh = l - 1;
    return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t m, const FastDiv& f) {
    return f.mul ? (__umulhi(m, f.mul) >> f.sh) : m;
}

// ---- element traits --------------------------------------------------------------
template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int DT = CVG_F32;
    static constexpr int VEC = 4;                 // elements per 16-byte chunk
    typedef f32x8 frag;                           // 8 k-values of one GEMM row
};
template <> struct Elem<bf16> {
    static constexpr int DT = CVG_BF16;
    static constexpr int VEC = 8;
    typedef bf16x8 frag;
};

__device__ __forceinline__ float to_f32(float v) { return v; }
__device__ __forceinline__ float to_f32(bf16 v) { return (float)v; }
template <typename T> __device__ __forceinline__ T from_f32(float v);
template <> __device__ __forceinline__ float from_f32<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float v) { return (bf16)v; }

// 8 consecutive elements <-> 8 floats (16 B of bf16, 32 B of f32).
template <typename T> __device__ __forceinline__ void load8(const T* p, float (&v)[8]);
template <> __device__ __forceinline__ void load8<float>(const float* p, float (&v)[8]) {
    f32x4 a = *(const f32x4*)p, b = *(const f32x4*)(p + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = a[i]; v[4 + i] = b[i]; }
}
template <> __device__ __forceinline__ void load8<bf16>(const bf16* p, float (&v)[8]) {
    bf16x8 a = *(const bf16x8*)p;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = (float)a[i];
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float (&v)[8]);
template <> __device__ __forceinline__ void store8<float>(float* p, const float (&v)[8]) {
    f32x4 a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = v[i]; b[i] = v[4 + i]; }
    *(f32x4*)p = a; *(f32x4*)(p + 4) = b;
}
template <> __device__ __forceinline__ void store8<bf16>(bf16* p, const float (&v)[8]) {
    bf16x8 a;
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = (bf16)v[i];
    *(bf16x8*)p = a;
}
// 4 consecutive elements (8 B of bf16, 16 B of f32).
template <typename T> __device__ __forceinline__ void load4(const T* p, float (&v)[4]);
template <> __device__ __forceinline__ void load4<float>(const float* p, float (&v)[4]) {
    f32x4 a = *(const f32x4*)p;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = a[i];
}
template <> __device__ __forceinline__ void load4<bf16>(const bf16* p, float (&v)[4]) {
    bf16x4 a = *(const bf16x4*)p;
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = (float)a[i];
}
template <typename T> __device__ __forceinline__ void store4(T* p, const float (&v)[4]);
template <> __device__ __forceinline__ void store4<float>(float* p, const float (&v)[4]) {
    f32x4 a;
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = v[i];
    *(f32x4*)p = a;
}
template <> __device__ __forceinline__ void store4<bf16>(bf16* p, const float (&v)[4]) {
    bf16x4 a;
#pragma unroll
    for (int i = 0; i < 4; ++i) a[i] = (bf16)v[i];
    *(bf16x4*)p = a;
}

// ---- MFMA wrappers: acc(32x32) += A(32 x 16) * B(16 x 32) -------------------------
// Fragment element j of lane l belongs to k-slot (l >> 5, j); both operands use the
// same slot numbering, so any consistent assignment of k values to slots is valid.
__device__ __forceinline__ void mma32(f32x16& acc, const bf16x8& a, const bf16x8& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}
__device__ __forceinline__ void mma32(f32x16& acc, const f32x8& a, const f32x8& b) {
#pragma unroll
    for (int j = 0; j < 8; ++j)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
}

// ---- wave / block reductions -------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }
static inline int ilog2_exact(int v) {          // -1 if v is not a power of two
    if (v <= 0 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}
