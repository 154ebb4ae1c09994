// Implicit-GEMM convolution kernels for gfx950 (MI355X).
//
// One gather geometry (cvg_conv_desc) serves Conv2d forward, input gradient, Linear
// layers and weight gradients.  Activations are NHWC, so the 8 consecutive k values
// of an MFMA fragment are 8 consecutive channels of one tap: a single 16-byte read.
//
//   igemm_fwd    Y[m][n] = sum_k gather(X)[m][k] * W[n][k]         (reduce over k)
//   igemm_wgrad  P[n][k] = sum_m dY[m][n] * gather(X)[m][k]         (reduce over m)
//
// Both are LDS-staged, register-prefetched two-stage pipelines with 4 waves per
// workgroup and 64x64 (or 64x32 / 32x32) per-wave tiles of 32x32 MFMA blocks.
// T = bf16 uses v_mfma_f32_32x32x16_bf16; T = float uses v_mfma_f32_32x32x2_f32,
// which is an exact f32 fma chain and serves as the tight-tolerance parity mode.
#include <stdlib.h>
#include "common.h"
#include "gemm_util.h"
#include "conv_patch.h"

namespace {

struct Geom {
    int H, W, Cin, cin_log2;
    int x_img_pitch, x_row_pitch, x_pix_pitch;
    int KW, kw_inv, ntaps, stride, tstep_y, tstep_x, off_y, off_x, pad_mode;
    int Kvalid, M, OW, OHW;
    FastDiv div_ohw, div_ow;
};

struct FwdArgs {
    Geom g;
    const void* x; const void* w; void* y;
    const float* bias; const float* cbias; const int* labels;
    const void* res; const void* gate; float* ws;
    int N, w_ld;
    int y_img_pitch, y_row_pitch, y_pix_pitch, oy_mul, oy_off, ox_mul, ox_off;
    int nk_total, nk_per_split, ksplit;
};

struct WgradArgs {
    Geom g;
    const void* x; const void* dy; float* ws;
    int dy_ld, n_rows, k_ld;          // partial layout [split][n_rows][k_ld]
    int nsteps_total, steps_per_split;
};

// tap index -> gathered-coordinate offset
__device__ __forceinline__ void tap_offset(const Geom& g, int tap, int& dy, int& dx) {
    int ty = (tap * g.kw_inv) >> 16;
    int tx = tap - ty * g.KW;
    dy = ty * g.tstep_y + g.off_y;
    dx = tx * g.tstep_x + g.off_x;
}

// Coordinates after padding; returns false when the element is a structural zero.
__device__ __forceinline__ bool pad_coord(const Geom& g, int& iy, int& ix) {
    if (g.pad_mode == CVG_PAD_REFLECT) {
        iy = iy < 0 ? -iy : iy;
        iy = iy >= g.H ? 2 * g.H - 2 - iy : iy;
        ix = ix < 0 ? -ix : ix;
        ix = ix >= g.W ? 2 * g.W - 2 - ix : ix;
        return true;
    }
    return (unsigned)iy < (unsigned)g.H && (unsigned)ix < (unsigned)g.W;
}

// ================================================================================
// forward / input-gradient kernel
// ================================================================================
template <typename T, int BM, int BN, int WM, int WN>
__device__ __forceinline__ void igemm_fwd_tile(const FwdArgs& a, const int zsplit) {
    constexpr int VEC = Elem<T>::VEC;
    constexpr int BK = 8 * VEC;                 // 128 bytes of k per row and step
    constexpr int PITCH = 144;                  // LDS row pitch: 128 B + 16 B pad (conflict-free b128 reads)
    constexpr int AR = BM / 32, BR = BN / 32;   // rows per thread per tile
    constexpr int WTM = BM / WM, WTN = BN / WN, MI = WTM / 32, NI = WTN / 32;
    constexpr int STAGE = (BM + BN) * PITCH;
    typedef typename Elem<T>::frag frag;
    static_assert(WM * WN == 4, "4 waves");
    // one LDS stage (the next k-step waits in registers): three or four workgroups share a CU and
    // cover each other's barriers
    constexpr int OUT_BYTES = BM * (BN * (int)sizeof(T) + 16);   // the staged output tile reuses the buffer
    __shared__ __attribute__((aligned(16))) char smem[STAGE > OUT_BYTES ? STAGE : OUT_BYTES];

    const Geom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
    const T* __restrict__ xp = (const T*)a.x;
    const T* __restrict__ wp = (const T*)a.w;

    // ---- loader bookkeeping: this thread fetches 16-byte chunk `cc` of rows r0+32i
    const int cc = tid & 7, r0 = tid >> 3;
    int riy[AR], rix[AR];
    size_t rbase[AR];                            // element offset of the row's image: may exceed 32 bits
    unsigned rmask = 0;
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        int m = m0 + r0 + 32 * i;
        riy[i] = rix[i] = 0;
        rbase[i] = 0;
        if (m < g.M) {
            uint32_t b = fdiv(m, g.div_ohw);
            uint32_t rem = m - b * g.OHW;
            uint32_t oy = fdiv(rem, g.div_ow);
            uint32_t ox = rem - oy * g.OW;
            riy[i] = oy * g.stride;
            rix[i] = ox * g.stride;
            rbase[i] = (size_t)b * g.x_img_pitch;
            rmask |= 1u << i;
        }
    }
    const size_t wrow = (size_t)(n0 + r0) * a.w_ld;

    u32x4 ra[AR], rb[BR];
    auto load_tile = [&](int ks) {
        const int k0 = ks * BK + cc * VEC;
        const bool kv = k0 < g.Kvalid;
        int tap = 0, c = k0;
        if (g.cin_log2 >= 0) { tap = k0 >> g.cin_log2; c = k0 & (g.Cin - 1); }
        int dy, dx;
        tap_offset(g, tap, dy, dx);
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            int iy = riy[i] + dy, ix = rix[i] + dx;
            bool ok = kv && ((rmask >> i) & 1u);
            ok = pad_coord(g, iy, ix) && ok;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (ok) v = *(const u32x4*)(xp + (rbase[i] + (size_t)(iy * g.x_row_pitch + ix * g.x_pix_pitch + c)));
            ra[i] = v;
        }
#pragma unroll
        for (int i = 0; i < BR; ++i)
            rb[i] = *(const u32x4*)(wp + wrow + (size_t)(32 * i) * a.w_ld + (ks * BK + cc * VEC));
    };
    auto store_tile = [&]() {
        char* As = smem;
        char* Ws = As + BM * PITCH;
#pragma unroll
        for (int i = 0; i < AR; ++i) *(u32x4*)(As + (r0 + 32 * i) * PITCH + cc * 16) = ra[i];
#pragma unroll
        for (int i = 0; i < BR; ++i) *(u32x4*)(Ws + (r0 + 32 * i) * PITCH + cc * 16) = rb[i];
    };

    f32x16 acc[NI][MI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;

    const int ks_beg = zsplit * a.nk_per_split;
    const int ks_end = min(a.nk_total, ks_beg + a.nk_per_split);

    if (ks_beg < ks_end) load_tile(ks_beg);
    for (int ks = ks_beg; ks < ks_end; ++ks) {
        __syncthreads();                         // everyone is done reading the previous k-step
        store_tile();
        __syncthreads();
        if (ks + 1 < ks_end) load_tile(ks + 1);
        const char* As = smem;
        const char* Ws = As + BM * PITCH;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            const int koff = (kk * 16 + 8 * h) * (int)sizeof(T);
            frag wf[NI], xf[MI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
                wf[ni] = *(const frag*)(Ws + (wn * WTN + ni * 32 + l31) * PITCH + koff);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
                xf[mi] = *(const frag*)(As + (wm * WTM + mi * 32 + l31) * PITCH + koff);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) mma32(acc[ni][mi], wf[ni], xf[mi]);
        }
    }
    __syncthreads();

    // ---- epilogue: D[row = n][col = m]; lane owns pixel column l31, 4 consecutive n per register group.
    // Split-K partials go straight to the workspace; final values are staged in LDS (free after the
    // loop's last barrier) and leave as whole rows, 16 bytes per lane.
    T* __restrict__ yp = (T*)a.y;
    const T* __restrict__ resp = (const T*)a.res;
    const T* __restrict__ gatep = (const T*)a.gate;
    constexpr int ES = (int)sizeof(T);
    constexpr int OP = BN * ES + 16;             // staged row pitch
    static_assert(BM * OP == OUT_BYTES, "staged tile size");
    constexpr int PPR = BN * ES / 16;            // 16-byte pieces per staged row
    constexpr int EPP = 16 / ES;
    constexpr int NPC = BM * PPR / 256;          // pieces per thread in the second half
    // output gating (y = conv * LeakyReLU'(gate)): the gate pieces this thread will need in the second
    // half are requested now, so that they arrive while the accumulators go through LDS
    const bool gate_only = gatep && !resp;
    u32x4 gq[NPC];
    if (gate_only) {
#pragma unroll
        for (int i = 0; i < NPC; ++i) {
            const int idx = tid + 256 * i;
            const int ml = idx / PPR, pc = idx % PPR;
            const int m = m0 + ml, n = n0 + pc * EPP;
            gq[i] = u32x4{0u, 0u, 0u, 0u};
            if (m >= g.M || n >= a.N) continue;
            const uint32_t b = fdiv(m, g.div_ohw);
            const uint32_t rem = m - b * g.OHW;
            const uint32_t oy = fdiv(rem, g.div_ow);
            const uint32_t ox = rem - oy * g.OW;
            const size_t yoff = (size_t)b * a.y_img_pitch + (size_t)(oy * a.oy_mul + a.oy_off) * a.y_row_pitch +
                                (size_t)(ox * a.ox_mul + a.ox_off) * a.y_pix_pitch;
            if (n + EPP <= a.N) gq[i] = *(const u32x4*)(gatep + yoff + n);
            else { const unsigned long long g8 = *(const unsigned long long*)(gatep + yoff + n); gq[i][0] = (unsigned)g8; gq[i][1] = (unsigned)(g8 >> 32); }
        }
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int ml = wm * WTM + mi * 32 + l31, m = m0 + ml;
        const bool live = m < g.M;
        const uint32_t b = fdiv(live ? m : 0, g.div_ohw);
        const uint32_t rem = (live ? m : 0) - b * g.OHW;
        const uint32_t oy = fdiv(rem, g.div_ow);
        const uint32_t ox = rem - oy * g.OW;
        const size_t yoff = (size_t)b * a.y_img_pitch + (size_t)(oy * a.oy_mul + a.oy_off) * a.y_row_pitch +
                            (size_t)(ox * a.ox_mul + a.ox_off) * a.y_pix_pitch;
        const float* cb = (a.cbias && live) ? a.cbias + (size_t)a.labels[b] * a.N : nullptr;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int nl = wn * WTN + ni * 32 + 8 * q + 4 * h, n = n0 + nl;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[ni][mi][4 * q + j];
                if (a.ksplit > 1) {
                    if (live && n < a.N) {
                        f32x4 o = {v[0], v[1], v[2], v[3]};
                        *(f32x4*)(a.ws + ((size_t)zsplit * g.M + m) * a.N + n) = o;
                    }
                    continue;
                }
                if (live && n < a.N) {
                    if (a.bias) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] += a.bias[n + j];
                    }
                    if (cb) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] += cb[n + j];
                    }
                    if (resp) {
                        float r[4];
                        load4<T>(resp + yoff + n, r);
                        if (gatep) {
                            float gt[4];
                            load4<T>(gatep + yoff + n, gt);
#pragma unroll
                            for (int j = 0; j < 4; ++j) r[j] *= gt[j] > 0.f ? 1.f : LRELU_SLOPE;
                        }
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] += r[j];
                    }
                }
                store4<T>((T*)(smem + ml * OP) + nl, v);
            }
        }
    }
    if (a.ksplit > 1) return;                    // uniform over the grid
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NPC; ++i) {
        const int idx = tid + 256 * i;
        const int ml = idx / PPR, pc = idx % PPR;
        const int m = m0 + ml, n = n0 + pc * EPP;
        if (m >= g.M || n >= a.N) continue;
        const uint32_t b = fdiv(m, g.div_ohw);
        const uint32_t rem = m - b * g.OHW;
        const uint32_t oy = fdiv(rem, g.div_ow);
        const uint32_t ox = rem - oy * g.OW;
        const size_t yoff = (size_t)b * a.y_img_pitch + (size_t)(oy * a.oy_mul + a.oy_off) * a.y_row_pitch +
                            (size_t)(ox * a.ox_mul + a.ox_off) * a.y_pix_pitch;
        u32x4 v = *(const u32x4*)(smem + ml * OP + pc * 16);
        if (gate_only) v = mul_gate<T>(v, gq[i]);
        if (n + EPP <= a.N) {
            *(u32x4*)(yp + yoff + n) = v;
        } else {                                 // N is a multiple of 4: half a 16-bit piece
            *(unsigned long long*)(yp + yoff + n) = ((unsigned long long)v[1] << 32) | v[0];
        }
    }
}

template <typename T, int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(256) igemm_fwd(const FwdArgs a) {
    igemm_fwd_tile<T, BM, BN, WM, WN>(a, blockIdx.z);
}

// Up to four convolutions over the same output grid in one launch (blockIdx.z picks one): the
// output-parity classes of a stride-2 convolution's input gradient.
struct FwdArgsGroup { FwdArgs a[4]; };
template <typename T, int BM, int BN, int WM, int WN>
__global__ void __launch_bounds__(256) igemm_fwd_group(const FwdArgsGroup m) {
    igemm_fwd_tile<T, BM, BN, WM, WN>(m.a[blockIdx.z], 0);
}

// Measurement aid: what the matrix pipes of this GPU deliver right now with nothing else in the way.  Every wave issues
// independent v_mfma_f32_32x32x16_bf16 back to back from registers (8 accumulators, operands with random mantissas and
// signs for realistic switching activity); 32 x 16 x 32768 FLOPs per wave and iteration.  The rate it reaches (clock
// under matrix load included) is the practical ceiling the convolution kernels are compared with beside the data-sheet peak.
__global__ void __launch_bounds__(256) mfma_rate_probe(float* __restrict__ out, int iters) {
    f32x16 acc[8];
    bf16x8 a, b;
    unsigned h = (blockIdx.x * 256u + threadIdx.x) * 2654435761u + 12345u;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        h = h * 1664525u + 1013904223u; a[j] = (bf16)(((int)(h >> 8) % 2001 - 1000) * 1e-3f);
        h = h * 1664525u + 1013904223u; b[j] = (bf16)(((int)(h >> 8) % 2001 - 1000) * 1e-3f);
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) mma32(acc[i], a, b);
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[i][r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = sum;
}

// split-K finish: y[m][n] = sum_z ws[z][m][n] + bias[n]   (dense rows of y_ld)
template <typename T>
__global__ void splitk_finish(const float* __restrict__ ws, T* __restrict__ y, const float* __restrict__ bias,
                              int M, int N, int y_ld, int ksplit) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // over M * N/4
    const int n4 = N / 4;
    if (idx >= (int64_t)M * n4) return;
    const int m = (int)(idx / n4), n = (int)(idx % n4) * 4;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int z = 0; z < ksplit; ++z) s += *(const f32x4*)(ws + ((size_t)z * M + m) * N + n);
    float v[4] = {s[0], s[1], s[2], s[3]};
    if (bias) {
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += bias[n + j];
    }
    store4<T>(y + (size_t)m * y_ld + n, v);
}

// ================================================================================
// weight-gradient kernel
// ================================================================================
template <typename T, int BNW, int BKW, int WN, int WK>
__global__ void __launch_bounds__(256, 2) igemm_wgrad(const WgradArgs a) {
    constexpr int VEC = Elem<T>::VEC;
    constexpr int BMS = 128 / (int)sizeof(T);                 // pixels per step: 64 bf16 / 32 f32
    constexpr int PG = BNW * (int)sizeof(T) + 64, PA = BKW * (int)sizeof(T) + 64;
    constexpr int CG = BNW / VEC, CA = BKW / VEC;             // 16-byte chunks per row
    constexpr int RG = 256 / CG, RA = 256 / CA;               // rows covered per pass
    constexpr int PGN = BMS / RG, PAN = BMS / RA;             // passes
    constexpr int WTN = BNW / WN, WTK = BKW / WK, NI = WTN / 32, KI = WTK / 32;
    constexpr int STAGE = BMS * (PG + PA);
    typedef typename Elem<T>::frag frag;
    static_assert(WN * WK == 4 && PGN >= 1 && PAN >= 1, "tile");
    // one LDS stage (the next step waits in registers): 41 KB, so three workgroups share a CU and
    // cover each other's barriers
    __shared__ __attribute__((aligned(16))) char smem[STAGE];

    const Geom& g = a.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave / WK, wk = wave % WK;
    const int l31 = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.y * BNW, kb0 = blockIdx.x * BKW;
    const T* __restrict__ xp = (const T*)a.x;
    const T* __restrict__ gp = (const T*)a.dy;

    // dY chunk of this thread
    const int ccg = tid % CG, rg0 = tid / CG;
    const bool gcol_ok = n0 + ccg * VEC + VEC <= a.dy_ld;
    // gathered-x chunk of this thread: k is fixed for the whole m loop
    const int cca = tid % CA, ra0 = tid / CA;
    const int k0 = kb0 + cca * VEC;
    const bool kv = k0 < g.Kvalid;
    int tap = 0, c = k0;
    if (g.cin_log2 >= 0) { tap = k0 >> g.cin_log2; c = k0 & (g.Cin - 1); }
    int dy, dx;
    tap_offset(g, tap, dy, dx);

    const int st_beg = blockIdx.z * a.steps_per_split;
    const int st_end = min(a.nsteps_total, st_beg + a.steps_per_split);

    u32x4 rgv[PGN], rav[PAN];
    auto load_tile = [&](int st) {
        const int mb = st * BMS;
#pragma unroll
        for (int i = 0; i < PGN; ++i) {
            const int m = mb + rg0 + RG * i;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (m < g.M && gcol_ok) v = *(const u32x4*)(gp + (size_t)m * a.dy_ld + n0 + ccg * VEC);
            rgv[i] = v;
        }
#pragma unroll
        for (int i = 0; i < PAN; ++i) {
            const int m = mb + ra0 + RA * i;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (m < g.M && kv) {
                const uint32_t b = fdiv(m, g.div_ohw);
                const uint32_t rem = m - b * g.OHW;
                const uint32_t oy = fdiv(rem, g.div_ow);
                const uint32_t ox = rem - oy * g.OW;
                int iy = oy * g.stride + dy, ix = ox * g.stride + dx;
                if (pad_coord(g, iy, ix))
                    v = *(const u32x4*)(xp + ((size_t)b * g.x_img_pitch + (size_t)(iy * g.x_row_pitch + ix * g.x_pix_pitch + c)));
            }
            rav[i] = v;
        }
    };
    auto store_tile = [&]() {
        char* Gs = smem;
        char* As = Gs + BMS * PG;
#pragma unroll
        for (int i = 0; i < PGN; ++i) *(u32x4*)(Gs + (rg0 + RG * i) * PG + ccg * 16) = rgv[i];
#pragma unroll
        for (int i = 0; i < PAN; ++i) *(u32x4*)(As + (ra0 + RA * i) * PA + cca * 16) = rav[i];
    };

    f32x16 acc[NI][KI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int ki = 0; ki < KI; ++ki)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ni][ki][r] = 0.f;

    if (st_beg < st_end) load_tile(st_beg);
    for (int st = st_beg; st < st_end; ++st) {
        __syncthreads();                     // everyone is done reading the previous step
        store_tile();
        __syncthreads();
        if (st + 1 < st_end) load_tile(st + 1);
        const char* Gs = smem;
        const char* As = Gs + BMS * PG;
#pragma unroll
        for (int sub = 0; sub < BMS / 16; ++sub) {
            frag gf[NI], af[KI];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) gf[ni] = TrLoad<T>::ld(Gs, PG, sub * 16, wn * WTN + ni * 32, lane);
#pragma unroll
            for (int ki = 0; ki < KI; ++ki) af[ki] = TrLoad<T>::ld(As, PA, sub * 16, wk * WTK + ki * 32, lane);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int ki = 0; ki < KI; ++ki) mma32(acc[ni][ki], gf[ni], af[ki]);
        }
    }

    // partials: ws[z][n][k], D[row = n][col = k]
    float* out = a.ws + (size_t)blockIdx.z * a.n_rows * a.k_ld;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int ki = 0; ki < KI; ++ki) {
            const int k = kb0 + wk * WTK + ki * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + wn * WTN + ni * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                out[(size_t)n * a.k_ld + k] = acc[ni][ki][r];
            }
        }
}

template <typename T>
__global__ void weight_pack(const float* __restrict__ canonical, T* __restrict__ packed, const cvg_wmap map,
                            int rows, int ld) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)rows * ld) return;
    const int n = (int)(idx / ld), k = (int)(idx % ld);
    const int ci = wmap_index(map, n, k);
    packed[idx] = from_f32<T>(ci >= 0 ? canonical[ci] : 0.f);
}

// every layout of a network in one launch: block -> job by bisection over block0, 8 elements per thread
template <typename T>
__device__ __forceinline__ void pack_group(const cvg_pack_job& j, int64_t g) {
    int n, k;
    if (j.mode == 1) {                       // fragment order, see patch_pack
        // Thread -> fragment piece with the tap varying fastest, then the row: canonical weights are
        // [..][..][tap], so neighbouring threads read neighbouring floats (whole cache lines instead
        // of one float per 32-byte sector); the 16-byte pieces they write interleave in L2.
        const int T_ = j.ntaps;
        int64_t u = g;
        const int tap = (int)(u % T_); u /= T_;
        const int nl = (int)(u & 31); u >>= 5;
        const int hh = (int)(u & 1); u >>= 1;
        const int sub = (int)(u & 1); u >>= 1;
        const int chunk = (int)(u % j.nchunks);
        const int nblk = (int)(u / j.nchunks);
        const int kstep = (chunk * T_ + tap) * 2 + sub;
        g = ((int64_t)nblk * (j.nchunks * T_ * 2) + kstep) * 64 + hh * 32 + nl;
        n = nblk * 32 + nl;
        k = tap * j.Cin + chunk * 32 + sub * 16 + 8 * hh;
    } else if (j.mode == 2) {
        // row-major, canonical storage contiguous along the row index nq = n / nd (a transposed Linear weight:
        // W[k][n] read as rows n): 64 consecutive threads take 16 consecutive nq x 4 consecutive k groups, so
        // every load covers 64 contiguous bytes of 16 lanes and every row receives 64 contiguous bytes,
        // instead of one 4-byte sector per lane (8x the traffic; 27 ms for a 524288 x 5024 weight)
        const int G4 = j.ld >> 5, nd = j.map.nd;
        const int l = (int)(g & 63);
        int64_t tile = g >> 6;
        const int kq4 = (int)(tile % G4); tile /= G4;
        const int nr = (int)(tile % nd);
        const int nq = (int)(tile / nd) * 16 + (l >> 2);
        n = nq * nd + nr;
        k = (kq4 * 4 + (l & 3)) * 8;
        g = ((int64_t)n * j.ld + k) >> 3;
    } else {
        // row-major [n][tap * kd + c]: inside a row the threads walk the taps fastest for the same
        // reason; the padding groups behind the last tap keep their places
        const int G = j.ld >> 3, kd = j.map.kd, T_ = j.map.ntaps;
        n = (int)(g / G);
        const int u = (int)(g - (int64_t)n * G);
        k = u * 8;
        if ((kd & 7) == 0 && u < T_ * (kd >> 3)) {
            const int cg = u / T_, tap = u - cg * T_;
            k = tap * kd + cg * 8;
            g = ((int64_t)n * j.ld + k) >> 3;
        }
    }
    float v[8];
    int ci[8];
    wmap_index_run<8>(j.map, n, k, ci);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = ci[e] >= 0 ? j.canonical[ci[e]] : 0.f;
    store8<T>((T*)j.packed + g * 8, v);
}
__global__ void __launch_bounds__(256) weight_pack_batch(const cvg_pack_job* __restrict__ jobs, int n_jobs) {
    int lo = 0, hi = n_jobs - 1;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (jobs[mid].block0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    const cvg_pack_job& j = jobs[lo];
    const int64_t g = (int64_t)(blockIdx.x - j.block0) * 256 + threadIdx.x;
    if (g >= j.n_groups) return;
    if (j.dtype == CVG_BF16) pack_group<bf16>(j, g);
    else pack_group<float>(j, g);
}

// ---- host side -------------------------------------------------------------------
int elem_size(int dtype) { return dtype == CVG_BF16 ? 2 : 4; }

bool patch_enabled() {
    return !cvg_switches().no_patch;
}

int make_geom(const cvg_conv_desc* d, Geom& g) {
    CVG_CHECK(d->dtype == CVG_F32 || d->dtype == CVG_BF16, "conv: bad dtype %d", d->dtype);
    CVG_CHECK(d->B > 0 && d->H > 0 && d->W > 0 && d->OH > 0 && d->OW > 0, "conv: bad dims");
    CVG_CHECK(d->Cin > 0 && d->Cin % 8 == 0, "conv: Cin %d must be a multiple of 8", d->Cin);
    CVG_CHECK(d->ntaps >= 1 && d->ntaps <= 64 && d->KW >= 1 && d->ntaps % d->KW == 0, "conv: bad taps");
    g.cin_log2 = -1;
    if (d->ntaps > 1) {
        g.cin_log2 = ilog2_exact(d->Cin);
        CVG_CHECK(g.cin_log2 >= 0, "conv: Cin %d must be a power of two when ntaps > 1", d->Cin);
    }
    const int64_t M = (int64_t)d->B * d->OH * d->OW;
    CVG_CHECK(M < (1ll << 31), "conv: too many rows");
    // (image bases are 64-bit in every kernel; offsets inside one image are 32-bit, which the descriptor's
    // int32 pitches guarantee)
    CVG_CHECK((d->x_pix_pitch * elem_size(d->dtype)) % 8 == 0 && d->x_row_pitch % 4 == 0 && d->x_img_pitch % 4 == 0,
              "conv: pitches must keep 8-byte alignment");
    if (d->pad_mode == CVG_PAD_REFLECT) {
        // a mirrored coordinate must land inside the tensor after ONE reflection
        const int kh = d->ntaps / d->KW;
        int lo_y = d->off_y < (kh - 1) * d->tstep_y + d->off_y ? d->off_y : (kh - 1) * d->tstep_y + d->off_y;
        int hi_y = (d->OH - 1) * d->stride + (d->off_y > (kh - 1) * d->tstep_y + d->off_y ? d->off_y : (kh - 1) * d->tstep_y + d->off_y);
        int lo_x = d->off_x < (d->KW - 1) * d->tstep_x + d->off_x ? d->off_x : (d->KW - 1) * d->tstep_x + d->off_x;
        int hi_x = (d->OW - 1) * d->stride + (d->off_x > (d->KW - 1) * d->tstep_x + d->off_x ? d->off_x : (d->KW - 1) * d->tstep_x + d->off_x);
        CVG_CHECK(-lo_y < d->H && hi_y - (d->H - 1) < d->H && -lo_x < d->W && hi_x - (d->W - 1) < d->W,
                  "conv: reflection padding wider than the tensor");
    }
    g.H = d->H; g.W = d->W; g.Cin = d->Cin;
    g.x_img_pitch = d->x_img_pitch; g.x_row_pitch = d->x_row_pitch; g.x_pix_pitch = d->x_pix_pitch;
    g.KW = d->KW; g.kw_inv = (65536 + d->KW - 1) / d->KW; g.ntaps = d->ntaps;
    g.stride = d->stride; g.tstep_y = d->tstep_y; g.tstep_x = d->tstep_x;
    g.off_y = d->off_y; g.off_x = d->off_x; g.pad_mode = d->pad_mode;
    g.Kvalid = d->ntaps * d->Cin; g.M = (int)M; g.OW = d->OW; g.OHW = d->OH * d->OW;
    g.div_ohw = make_fastdiv(g.OHW); g.div_ow = make_fastdiv(g.OW);
    return 0;
}

int bk_of(int dtype) { return dtype == CVG_BF16 ? 64 : 32; }

// split-K decision of the forward kernel (Linear layers with few output tiles)
int fwd_ksplit(const cvg_conv_desc* d, int M, bool dense) {
    const int BK = bk_of(d->dtype);
    const int nk = ceil_div((int64_t)d->ntaps * d->Cin, BK);
    const int tiles = ceil_div(M, 128) * ceil_div(d->N, 128);
    if (!dense || tiles >= 192 || nk < 16) return 1;
    int ks = 512 / tiles;                        // two workgroups fit a CU
    if (ks > nk / 4) ks = nk / 4;
    if (ks > 64) ks = 64;
    return ks < 1 ? 1 : ks;
}
bool fwd_dense_out(const cvg_conv_desc* d) {
    return d->oy_mul == 1 && d->ox_mul == 1 && d->oy_off == 0 && d->ox_off == 0 &&
           d->y_row_pitch == d->OW * d->y_pix_pitch && d->y_img_pitch == d->OH * d->y_row_pitch;
}

template <typename T>
int launch_fwd(const cvg_conv_desc* d, FwdArgs& a, hipStream_t s) {
    const int M = a.g.M, N = d->N;
    dim3 blk(256);
    if (N > 64) {
        dim3 grid(ceil_div(M, 128), ceil_div(N, 128), a.ksplit);
        igemm_fwd<T, 128, 128, 2, 2><<<grid, blk, 0, s>>>(a);
    } else if (N > 32) {
        dim3 grid(ceil_div(M, 256), 1, a.ksplit);
        igemm_fwd<T, 256, 64, 4, 1><<<grid, blk, 0, s>>>(a);
    } else {
        dim3 grid(ceil_div(M, 256), 1, a.ksplit);
        igemm_fwd<T, 256, 32, 4, 1><<<grid, blk, 0, s>>>(a);
    }
    CVG_LAUNCH_CHECK();
    if (a.ksplit > 1) {
        const int64_t n = (int64_t)M * (N / 4);
        splitk_finish<T><<<ceil_div(n, 256), 256, 0, s>>>(a.ws, (T*)a.y, a.bias, M, N, d->y_pix_pitch, a.ksplit);
        CVG_LAUNCH_CHECK();
    }
    return 0;
}

template <typename T>
int launch_fwd_group(int n, int M, int N, const FwdArgsGroup& m, hipStream_t s) {
    dim3 blk(256);
    if (N > 64) {
        dim3 grid(ceil_div(M, 128), ceil_div(N, 128), n);
        igemm_fwd_group<T, 128, 128, 2, 2><<<grid, blk, 0, s>>>(m);
    } else if (N > 32) {
        dim3 grid(ceil_div(M, 256), 1, n);
        igemm_fwd_group<T, 256, 64, 4, 1><<<grid, blk, 0, s>>>(m);
    } else {
        dim3 grid(ceil_div(M, 256), 1, n);
        igemm_fwd_group<T, 256, 32, 4, 1><<<grid, blk, 0, s>>>(m);
    }
    CVG_LAUNCH_CHECK();
    return 0;
}

// weight-gradient tiling: output tile (BNW x BKW), pixels split over gridDim.z
struct WgradPlan { int bnw, bkw, n_rows, k_ld, nsteps, nsplit, steps_per_split; };
WgradPlan plan_wgrad(const cvg_conv_desc* d, int M) {
    WgradPlan p;
    p.bnw = d->N > 64 ? 128 : (d->N > 32 ? 64 : 32);
    // 128 x 256 output tiles are opt-in: measured 0.05 ms per c3 step slower than 128 x 128 (two waves per SIMD
    // instead of four hide less of the gather latency)
    p.bkw = (d->dtype == CVG_BF16 && p.bnw == 128 && (int64_t)d->ntaps * d->Cin >= 256 && cvg_switches().wgrad_wide) ? 256 : 128;
    p.n_rows = ceil_div(d->N, p.bnw) * p.bnw;
    p.k_ld = ceil_div((int64_t)d->ntaps * d->Cin, p.bkw) * p.bkw;
    const int bms = d->dtype == CVG_BF16 ? 64 : 32;
    p.nsteps = ceil_div(M, bms);
    const int tiles = (p.n_rows / p.bnw) * (p.k_ld / p.bkw);
    int want = ceil_div(1024, tiles);
    int maxs = p.nsteps / 4 > 0 ? p.nsteps / 4 : 1;
    p.nsplit = want < maxs ? want : maxs;
    if (p.nsplit > 512) p.nsplit = 512;
    p.steps_per_split = ceil_div(p.nsteps, p.nsplit);
    p.nsplit = ceil_div(p.nsteps, p.steps_per_split);
    return p;
}

template <typename T>
int launch_wgrad(const WgradPlan& p, WgradArgs& a, hipStream_t s) {
    dim3 grid(p.k_ld / p.bkw, p.n_rows / p.bnw, p.nsplit), blk(256);
    if (p.bnw == 128 && p.bkw == 256) igemm_wgrad<T, 128, 256, 2, 2><<<grid, blk, 0, s>>>(a);
    else if (p.bnw == 128) igemm_wgrad<T, 128, 128, 2, 2><<<grid, blk, 0, s>>>(a);
    else if (p.bnw == 64) igemm_wgrad<T, 64, 128, 2, 2><<<grid, blk, 0, s>>>(a);
    else igemm_wgrad<T, 32, 128, 1, 4><<<grid, blk, 0, s>>>(a);
    CVG_LAUNCH_CHECK();
    return 0;
}

}  // namespace

extern "C" {

size_t cvg_conv_fwd_workspace(const cvg_conv_desc* d) {
    const int64_t M = (int64_t)d->B * d->OH * d->OW;
    const int ks = fwd_ksplit(d, (int)M, fwd_dense_out(d));
    return ks > 1 ? (size_t)ks * M * d->N * sizeof(float) : 0;
}

int cvg_conv_fwd(const cvg_conv_desc* d, const void* x, const void* w, void* y, const float* bias,
                 const float* cbias, const int32_t* labels, const void* res, const void* gate, void* ws,
                 size_t ws_bytes, cvg_stream stream) {
    if (d && d->form == CVG_FORM_S2_TRANSPOSED) {
        S2tPlan sp;
        patch_dgrad_s2_plan(d, sp);
        CVG_CHECK(sp.ok, "conv_fwd: this transposed stride-2 geometry is not supported (dY channels must be a multiple of "
                  "32, dX channels of 8); use four cvg_conv_fwd_group classes");
        CVG_CHECK(x && w && y, "conv_fwd: null tensor");
        CVG_CHECK(!bias && !cbias && !res, "conv_fwd: the transposed stride-2 form takes no bias / res");
        return patch_dgrad_s2_launch(d, sp, x, w, y, gate, (hipStream_t)stream);
    }
    FwdArgs a;
    if (make_geom(d, a.g)) return -1;
    const int BK = bk_of(d->dtype);
    CVG_CHECK(d->N > 0 && d->N % 4 == 0, "conv_fwd: N %d must be a multiple of 4", d->N);
    CVG_CHECK(d->w_ld % BK == 0 && d->w_ld >= a.g.Kvalid, "conv_fwd: w_ld %d does not cover K %d", d->w_ld, a.g.Kvalid);
    CVG_CHECK(d->w_rows % 128 == 0 && d->w_rows >= d->N, "conv_fwd: w_rows %d must be N padded to 128", d->w_rows);
    CVG_CHECK(x && w && y, "conv_fwd: null tensor");
    CVG_CHECK(!cbias || labels, "conv_fwd: cbias needs labels");
    CVG_CHECK(d->y_pix_pitch % 4 == 0 && d->y_row_pitch % 4 == 0 && d->y_img_pitch % 4 == 0, "conv_fwd: y pitches");
    const bool dense = fwd_dense_out(d) && !res && !cbias;
    a.x = x; a.w = w; a.y = y; a.bias = bias; a.cbias = cbias; a.labels = labels; a.res = res; a.gate = gate;
    a.ws = (float*)ws; a.N = d->N; a.w_ld = d->w_ld;
    a.y_img_pitch = d->y_img_pitch; a.y_row_pitch = d->y_row_pitch; a.y_pix_pitch = d->y_pix_pitch;
    a.oy_mul = d->oy_mul; a.oy_off = d->oy_off; a.ox_mul = d->ox_mul; a.ox_off = d->ox_off;
    hipStream_t s = (hipStream_t)stream;
    const bool gate_only = gate && !res;         // y = conv * LeakyReLU'(gate): gather kernel, no split-K
    if (patch_enabled()) {
        PatchPlan pp;
        patch_fwd_plan(d, pp);
        CVG_CHECK(!(pp.ok && gate_only), "conv_fwd: gate without res is implemented by the gather kernel only; this layer's weights are packed for the patch kernel");
        if (pp.ok) return patch_fwd_launch(d, pp, x, w, y, bias, cbias, labels, res, gate, s);
    }
    a.nk_total = ceil_div(a.g.Kvalid, BK);
    a.ksplit = gate_only ? 1 : fwd_ksplit(d, a.g.M, dense);
    a.nk_per_split = ceil_div(a.nk_total, a.ksplit);
    a.ksplit = ceil_div(a.nk_total, a.nk_per_split);
    if (a.ksplit > 1)
        CVG_CHECK(ws && ws_bytes >= (size_t)a.ksplit * a.g.M * d->N * sizeof(float), "conv_fwd: workspace too small");
    return d->dtype == CVG_BF16 ? launch_fwd<bf16>(d, a, s) : launch_fwd<float>(d, a, s);
}

int32_t cvg_conv_stats_partials(const cvg_conv_desc* d) {
    PatchPlan pp;
    pp.ok = 0;
    if (patch_enabled()) patch_fwd_plan(d, pp);
    return pp.ok ? patch_stat_tiles(pp) : 0;
}

int cvg_conv_fwd_stats(const cvg_conv_desc* d, const void* x, const void* w, void* y, const float* bias,
                       const float* cbias, const int32_t* labels, float* partials, cvg_stream stream) {
    CVG_CHECK(d && x && w && y && partials, "conv_fwd_stats: null argument");
    CVG_CHECK(!cbias || labels, "conv_fwd_stats: cbias needs labels");
    CVG_CHECK(d->N > 0 && d->N % 4 == 0, "conv_fwd_stats: N %d must be a multiple of 4", d->N);
    PatchPlan pp;
    pp.ok = 0;
    if (patch_enabled()) patch_fwd_plan(d, pp);
    CVG_CHECK(pp.ok, "conv_fwd_stats: this layer runs on the gather kernel; use cvg_conv_fwd + cvg_bn_stats");
    return patch_fwd_launch(d, pp, x, w, y, bias, cbias, labels, nullptr, nullptr, (hipStream_t)stream, nullptr, nullptr, 0,
                            partials);
}

int cvg_conv_bn_act_fwd(const cvg_conv_desc* d, const void* x, const void* w, void* y, const float* bias,
                        const float* cbias, const int32_t* labels, const float* scale, const float* shift,
                        const void* res, cvg_stream stream) {
    CVG_CHECK(d && x && w && y && scale && shift, "conv_bn_act_fwd: null argument");
    CVG_CHECK(!cbias || labels, "conv_bn_act_fwd: cbias needs labels");
    CVG_CHECK(d->N > 0 && d->N % 4 == 0, "conv_bn_act_fwd: N %d must be a multiple of 4", d->N);
    PatchPlan pp;
    pp.ok = 0;
    if (patch_enabled()) patch_fwd_plan(d, pp);
    CVG_CHECK(pp.ok, "conv_bn_act_fwd: this layer runs on the gather kernel; use cvg_conv_fwd + cvg_bn_act_fwd");
    return patch_fwd_launch(d, pp, x, w, y, bias, cbias, labels, res, nullptr, (hipStream_t)stream, scale, shift, 1);
}

int cvg_conv_fwd_group(int32_t n, const cvg_conv_desc* descs, const void* x, const void* const* w, void* y,
                       const void* gate, cvg_stream stream) {
    CVG_CHECK(n >= 1 && n <= 4 && descs && x && w && y, "conv_fwd_group: bad args");
    FwdArgsGroup m;
    const cvg_conv_desc* d0 = &descs[0];
    for (int i = 0; i < n; ++i) {
        const cvg_conv_desc* d = &descs[i];
        FwdArgs& a = m.a[i];
        if (make_geom(d, a.g)) return -1;
        const int BK = bk_of(d->dtype);
        CVG_CHECK(d->dtype == d0->dtype && d->N == d0->N && a.g.M == m.a[0].g.M,
                  "conv_fwd_group: members must share dtype, N and the output grid");
        CVG_CHECK(d->N > 0 && d->N % 4 == 0, "conv_fwd_group: N %d must be a multiple of 4", d->N);
        CVG_CHECK(d->w_ld % BK == 0 && d->w_ld >= a.g.Kvalid, "conv_fwd_group: w_ld %d does not cover K %d", d->w_ld, a.g.Kvalid);
        CVG_CHECK(d->w_rows % 128 == 0 && d->w_rows >= d->N, "conv_fwd_group: w_rows %d must be N padded to 128", d->w_rows);
        CVG_CHECK(d->y_pix_pitch % 4 == 0 && d->y_row_pitch % 4 == 0 && d->y_img_pitch % 4 == 0, "conv_fwd_group: y pitches");
        CVG_CHECK(w[i], "conv_fwd_group: null weights");
        if (patch_enabled()) {                   // members are packed for the gather kernel
            PatchPlan pp;
            patch_fwd_plan(d, pp);
            CVG_CHECK(!pp.ok, "conv_fwd_group: member %d belongs to the patch kernel", i);
        }
        a.x = x; a.w = w[i]; a.y = y; a.bias = nullptr; a.cbias = nullptr; a.labels = nullptr; a.res = nullptr;
        a.gate = gate; a.ws = nullptr; a.N = d->N; a.w_ld = d->w_ld;
        a.y_img_pitch = d->y_img_pitch; a.y_row_pitch = d->y_row_pitch; a.y_pix_pitch = d->y_pix_pitch;
        a.oy_mul = d->oy_mul; a.oy_off = d->oy_off; a.ox_mul = d->ox_mul; a.ox_off = d->ox_off;
        a.nk_total = ceil_div(a.g.Kvalid, BK);
        a.ksplit = 1; a.nk_per_split = a.nk_total;
    }
    for (int i = n; i < 4; ++i) m.a[i] = m.a[0];
    hipStream_t s = (hipStream_t)stream;
    return d0->dtype == CVG_BF16 ? launch_fwd_group<bf16>(n, m.a[0].g.M, d0->N, m, s)
                                 : launch_fwd_group<float>(n, m.a[0].g.M, d0->N, m, s);
}

int cvg_conv_wgrad_strided(const cvg_conv_desc* d, const void* x, const void* dy, const cvg_wmap* map, float* grad,
                           int32_t accumulate, void* ws, size_t ws_bytes, cvg_stream stream) {
    CVG_CHECK(d && x && dy && map && grad && ws, "conv_wgrad_strided: null tensor");
    CVG_CHECK(map->kd > 0 && map->nd > 0, "conv_wgrad_strided: bad map");
    CVG_CHECK(d->y_pix_pitch % 4 == 0 && d->y_row_pitch % 4 == 0 && d->y_img_pitch % 4 == 0 && d->N % 8 == 0,
              "conv_wgrad_strided: dY pitches must keep 8-byte alignment, N a multiple of 8");
    PatchWgradPlan wp;
    patch_wgrad_plan(d, wp);
    CVG_CHECK(patch_enabled() && wp.ok, "conv_wgrad_strided: only the patch-staged kernels read strided dY");
    CVG_CHECK(ws_bytes >= wp.ws_bytes, "conv_wgrad_strided: workspace too small");
    return patch_wgrad_launch(d, wp, x, dy, d->y_pix_pitch, d->y_img_pitch, d->y_row_pitch, d->N, map, grad, accumulate,
                              (float*)ws, (hipStream_t)stream);
}

const char* cvg_conv_fwd_kernel(const cvg_conv_desc* d) {
    if (d->form == CVG_FORM_S2_TRANSPOSED) {
        S2tPlan sp;
        patch_dgrad_s2_plan(d, sp);
        return sp.ok ? "patch_dgrad_s2" : "unsupported";
    }
    if (patch_enabled()) {
        PatchPlan pp;
        patch_fwd_plan(d, pp);
        if (pp.ok) {
            // (the kernel a plain launch runs on; one with a residual or statistics epilogue stays on patch_fwd proper)
            if (lowk_fwd_takes(d, pp)) return "patch_fwd -> lowk_fwd<256px x 64ch, raw-pixel staging>";
            if (pp.BN == 128) return pp.BM == 256 ? "patch_fwd<256px x 128ch>" : "patch_fwd<128px x 128ch>";
            if (pp.walk && pp.KW == 1) return "patch_fwd<256px x 32ch, resident weights>";
            return pp.BN == 64 ? "patch_fwd<256px x 64ch>" : "patch_fwd<256px x 32ch>";
        }
    }
    return d->N > 64 ? "igemm_fwd<128x128>" : (d->N > 32 ? "igemm_fwd<256x64>" : "igemm_fwd<256x32>");
}

const char* cvg_conv_wgrad_kernel(const cvg_conv_desc* d) {
    if (patch_enabled()) {
        PatchWgradPlan wp;
        patch_wgrad_plan(d, wp);
        if (wp.ok) return wp.BNW == 128 ? "patch_wgrad<128ch>" : (wp.BNW == 64 ? "patch_wgrad<64ch>" : "patch_wgrad<32ch>");
    }
    return d->N > 64 ? "igemm_wgrad<128x128>" : (d->N > 32 ? "igemm_wgrad<64x128>" : "igemm_wgrad<32x128>");
}

size_t cvg_conv_wgrad_workspace(const cvg_conv_desc* d) {
    const int64_t M = (int64_t)d->B * d->OH * d->OW;
    WgradPlan p = plan_wgrad(d, (int)M);
    size_t bytes = (size_t)p.nsplit * p.n_rows * p.k_ld * sizeof(float);
    PatchWgradPlan wp;
    patch_wgrad_plan(d, wp);
    if (wp.ok && wp.ws_bytes > bytes) bytes = wp.ws_bytes;
    return bytes;
}

int cvg_conv_wgrad(const cvg_conv_desc* d, const void* x, const void* dy, int32_t dy_ld, const cvg_wmap* map,
                   float* grad, int32_t accumulate, void* ws, size_t ws_bytes, cvg_stream stream) {
    WgradArgs a;
    if (make_geom(d, a.g)) return -1;
    CVG_CHECK(x && dy && map && grad && ws, "conv_wgrad: null tensor");
    CVG_CHECK(dy_ld % 8 == 0 && dy_ld >= d->N, "conv_wgrad: dy_ld %d", dy_ld);
    CVG_CHECK(map->kd > 0 && map->nd > 0, "conv_wgrad: bad map");
    if (patch_enabled()) {
        PatchWgradPlan wp;
        patch_wgrad_plan(d, wp);
        if (wp.ok) {
            CVG_CHECK(ws_bytes >= wp.ws_bytes, "conv_wgrad: workspace too small");
            return patch_wgrad_launch(d, wp, x, dy, dy_ld, d->OH * d->OW * dy_ld, d->OW * dy_ld, dy_ld, map, grad,
                                      accumulate, (float*)ws, (hipStream_t)stream);
        }
    }
    WgradPlan p = plan_wgrad(d, a.g.M);
    CVG_CHECK(ws_bytes >= (size_t)p.nsplit * p.n_rows * p.k_ld * sizeof(float), "conv_wgrad: workspace too small");
    CVG_CHECK(map->n_valid <= p.n_rows && map->kd > 0 && map->nd > 0, "conv_wgrad: bad map");
    a.x = x; a.dy = dy; a.ws = (float*)ws; a.dy_ld = dy_ld; a.n_rows = p.n_rows; a.k_ld = p.k_ld;
    a.nsteps_total = p.nsteps; a.steps_per_split = p.steps_per_split;
    hipStream_t s = (hipStream_t)stream;
    int rc = d->dtype == CVG_BF16 ? launch_wgrad<bf16>(p, a, s) : launch_wgrad<float>(p, a, s);
    if (rc) return rc;
    launch_wgrad_reduce(a.ws, grad, map, p.n_rows, p.k_ld, p.nsplit, accumulate, s);
    CVG_LAUNCH_CHECK();
    return 0;
}

int cvg_weight_pack(const cvg_conv_desc* d, const cvg_wmap* map, const float* canonical, void* packed,
                    cvg_stream stream) {
    CVG_CHECK(d && map && canonical && packed && d->w_rows > 0 && d->w_ld > 0, "weight_pack: bad args");
    CVG_CHECK(map->kd > 0 && map->nd > 0, "weight_pack: bad map");
    hipStream_t s = (hipStream_t)stream;
    if (d->form == CVG_FORM_S2_TRANSPOSED) {
        S2tPlan sp;
        patch_dgrad_s2_plan(d, sp);
        CVG_CHECK(sp.ok, "weight_pack: unsupported transposed stride-2 geometry");
        PatchPlan pp;
        pp.nchunks = sp.nchunks; pp.T = 9;
        return patch_pack_launch(d, pp, map, canonical, packed, s);
    }
    if (patch_enabled()) {
        PatchPlan pp;
        patch_fwd_plan(d, pp);
        if (pp.ok) return patch_pack_launch(d, pp, map, canonical, packed, s);
    }
    const int rows = d->w_rows, ld = d->w_ld;
    const int64_t n = (int64_t)rows * ld;
    if (d->dtype == CVG_BF16) weight_pack<bf16><<<ceil_div(n, 256), 256, 0, s>>>(canonical, (bf16*)packed, *map, rows, ld);
    else weight_pack<float><<<ceil_div(n, 256), 256, 0, s>>>(canonical, (float*)packed, *map, rows, ld);
    CVG_LAUNCH_CHECK();
    return 0;
}

int cvg_weight_pack_job(const cvg_conv_desc* d, const cvg_wmap* map, const float* canonical, void* packed,
                        int32_t block0, cvg_pack_job* job) {
    CVG_CHECK(d && map && canonical && packed && job && d->w_rows > 0 && d->w_ld > 0, "weight_pack_job: bad args");
    CVG_CHECK(map->kd > 0 && map->nd > 0, "weight_pack_job: bad map");
    job->map = *map; job->canonical = canonical; job->packed = packed;
    job->block0 = block0; job->dtype = d->dtype; job->Cin = d->Cin;
    job->rows = d->w_rows; job->ld = d->w_ld; job->mode = 0; job->nchunks = 0; job->ntaps = d->ntaps;
    int64_t n = (int64_t)d->w_rows * d->w_ld;
    if (d->form == CVG_FORM_S2_TRANSPOSED) {
        S2tPlan sp;
        patch_dgrad_s2_plan(d, sp);
        CVG_CHECK(sp.ok, "weight_pack_job: unsupported transposed stride-2 geometry");
        CVG_CHECK((int64_t)sp.nchunks * 9 * 32 <= d->w_ld, "weight_pack_job: packed buffer too small");
        job->mode = 1; job->nchunks = sp.nchunks; job->ntaps = 9;
        n = (int64_t)d->w_rows * sp.nchunks * 9 * 32;
    } else if (patch_enabled()) {
        PatchPlan pp;
        patch_fwd_plan(d, pp);
        if (pp.ok) {
            CVG_CHECK((int64_t)pp.nchunks * pp.T * 32 <= d->w_ld, "weight_pack_job: packed buffer too small");
            job->mode = 1; job->nchunks = pp.nchunks; job->ntaps = pp.T;
            n = (int64_t)d->w_rows * pp.nchunks * pp.T * 32;
        }
    }
    // transposed layouts (canonical contiguous along the row index): tiled thread order, see pack_group
    if (job->mode == 0 && map->na == 1 && map->nb != 1 && map->kb != 1 && map->ka != 1 && d->w_ld % 32 == 0 &&
        d->w_rows % (16 * map->nd) == 0)
        job->mode = 2;
    CVG_CHECK(n % 8 == 0 && d->w_ld % 8 == 0, "weight_pack_job: packed size must be a multiple of 8");
    job->n_groups = n / 8;
    const int64_t blocks = (job->n_groups + 255) / 256;
    CVG_CHECK(blocks + block0 < (1ll << 31), "weight_pack_job: too many blocks");
    return (int)blocks;
}

int cvg_mfma_rate_probe(float* out, int32_t workgroups, int32_t iters, cvg_stream stream) {
    CVG_CHECK(out && workgroups > 0 && iters > 0, "mfma_rate_probe: bad args");
    mfma_rate_probe<<<workgroups, 256, 0, (hipStream_t)stream>>>(out, iters);
    CVG_LAUNCH_CHECK();
    return 0;
}

int cvg_weight_pack_batch(const cvg_pack_job* jobs_dev, int32_t n_jobs, int32_t total_blocks, cvg_stream stream) {
    CVG_CHECK(jobs_dev && n_jobs > 0 && total_blocks > 0, "weight_pack_batch: bad args");
    weight_pack_batch<<<total_blocks, 256, 0, (hipStream_t)stream>>>(jobs_dev, n_jobs);
    CVG_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
