// Patch-staged convolution kernels for gfx950 (stride-1 convolutions with >= 32 channels).
//
// A workgroup owns TB x TH x TW output pixels.  Per 32-channel chunk it stages the
// (TH+KH-1) x (TW+KW-1) input patch of each of its images in LDS once; every tap then
// reads its MFMA operand from the patch at a shifted pixel, so each input element
// comes from L2 about 1.3 times instead of KH*KW times.
//
//   patch_fwd    Y[pix][n]  = sum_{tap,c} X[pix+tap][c] * W[n][tap][c]
//                weights arrive pre-packed in MFMA-fragment order and stream straight
//                from L2 into registers (1 KiB coalesced per wave-instruction, no LDS)
//   patch_wgrad  P[n][tap][c] = sum_pix dY[pix][n] * X[pix+tap][c]
//                both operands are read from LDS with transposing reads; the dY
//                fragment is reused across all taps
#include <stdlib.h>
#include <stdio.h>
#include <mutex>
#include "conv_patch.h"
#include "gemm_util.h"

namespace {

constexpr int CK = 32;            // channels per staged chunk

struct TileGeom {
    int TB, TH, TW, PH, PW, npx;           // npx = TB*TH*TW valid tile pixels
    int tiles_x;
    int pair4;                             // tile pixels are numbered row r, row r + 4, row r + 1, ... (see tile_px)
    FastDiv div_tw, div_thw, div_pw, div_phw, div_tx;
};

struct PatchFwdArgs {
    const void* x; const void* wf; void* y;
    const float* bias; const float* cbias; const int* labels; const void* res; const void* gate;
    const float* scale; const float* shift; int act;     // eval-mode BatchNorm + LeakyReLU in the epilogue
    float* stat_ws;                                       // per-tile column sums of the stored output, or null
    int B, H, W, Cin, x_img_pitch, x_row_pitch, x_pix_pitch;
    int off_y, off_x, pad_mode, KW, T;
    int OH, OW, N;
    int y_img_pitch, y_row_pitch, y_pix_pitch, oy_mul, oy_off, ox_mul, ox_off;
    int nchunks, patch_px, patch_bytes;
    int lane_perm;                          // lane groups of an LDS read take consecutive pixels: lane_pixel()
    int row_fast;                           // tile rows hold a multiple of the epilogue's pixels per thread
    // A workgroup keeps its tile position and output-channel tile and walks over image groups
    // blockIdx.z, blockIdx.z + gstride, ... < groups (gstride = gridDim.z).  Every per-thread offset is
    // the same for each of them, so the next group's first patches and the weight ring are requested
    // while the current group is still being multiplied.  gstride == groups: one group per workgroup.
    int groups, gstride;
    int stage_off;                          // LDS offset of the epilogue's staging area (0: over the patch buffers)
    int stage_shared;                       // walking form: the staging area lies over the SECOND patch buffer (see patch_fwd_plan)
#ifdef EXP_TIMING
    long long* dbg;
#endif
    TileGeom t;
};

__device__ __forceinline__ bool pad_xy(int pad_mode, int H, int W, int& iy, int& ix) {
    if (pad_mode == CVG_PAD_REFLECT) {
        iy = iy < 0 ? -iy : iy;
        iy = iy >= H ? 2 * H - 2 - iy : iy;
        ix = ix < 0 ? -ix : ix;
        ix = ix >= W ? 2 * W - 2 - ix : ix;
        return true;
    }
    return (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
}

// tile pixel q -> (tb, ty, tx).  pair4 (8 x 8 tiles over a 10-pixel-wide patch): 16 consecutive q are row g and row
// g + 4 of an image.  A ds_read_b128 lane group takes 16 consecutive q; its 16 patch pixels hit distinct banks when
// their patch indices are distinct mod 16 (80-byte pitch), and rows g, g + 4 start 40 = 8 mod 16 pixels apart, so
// the two half rows cover all 16 residues.  In natural order (rows g, g + 1: 10 apart) six of the sixteen collide.
// (Only patch_fwd numbers its pixels this way; the other kernels use tile_px, which has no such case: one more
// live value in the nine-tap weight-gradient kernel, which sits at 256 registers, would spill.)
__device__ __forceinline__ void tile_px_fwd(const TileGeom& t, int q, int& tb, int& ty, int& tx) {
    tb = fdiv(q, t.div_thw);
    const int r = q - tb * (t.TH * t.TW);
    if (t.pair4) {
        ty = (r >> 4) + 4 * ((r >> 3) & 1);
        tx = r & 7;
        return;
    }
    ty = fdiv(r, t.div_tw);
    tx = r - ty * t.TW;
}
__device__ __forceinline__ void tile_px(const TileGeom& t, int q, int& tb, int& ty, int& tx) {
    tb = fdiv(q, t.div_thw);
    const int r = q - tb * (t.TH * t.TW);
    ty = fdiv(r, t.div_tw);
    tx = r - ty * t.TW;
}

// ds_read_b128 is served in 16-lane groups {0-3,12-15,20-27} and {4-11,16-19,28-31} (likewise for
// the upper half-wave).  With 16-pixel tile rows the natural lane -> pixel order puts pixels of
// two different tile rows in one group and they collide on LDS banks (the rows are 18 or 22 patch
// pixels apart).  This order gives each group 16 pixels of ONE tile row: conflict-free.
__device__ __forceinline__ int lane_pixel(int l31, int permute) {
    if (!permute) return l31;
    if (l31 < 4) return l31;
    if (l31 < 12) return l31 + 12;          // 4..11  -> 16..23
    if (l31 < 16) return l31 - 8;           // 12..15 -> 4..7
    if (l31 < 20) return l31 + 8;           // 16..19 -> 24..27
    if (l31 < 28) return l31 - 12;          // 20..27 -> 8..15
    return l31;                             // 28..31 -> 28..31
}

// One staged 16-byte piece plus its residual term: v + r * LeakyReLU'(g) per element (g > 0 ? 1 : slope;
// no gate: v + r), then LeakyReLU if act; in f32, rounded once more to T.
template <typename T> __device__ __forceinline__ u32x4 add_gated(u32x4 v, u32x4 r, u32x4 g, bool gated, bool act);
template <> __device__ __forceinline__ u32x4 add_gated<float>(u32x4 v, u32x4 r, u32x4 g, bool gated, bool act) {
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float rv = __uint_as_float(r[j]);
        const float k = (!gated || __uint_as_float(g[j]) > 0.f) ? 1.f : LRELU_SLOPE;
        float f = __uint_as_float(v[j]) + rv * k;
        if (act) f = f > 0.f ? f : LRELU_SLOPE * f;
        o[j] = __float_as_uint(f);
    }
    return o;
}
template <> __device__ __forceinline__ u32x4 add_gated<bf16>(u32x4 v, u32x4 r, u32x4 g, bool gated, bool act) {
    const bf16x8 vb = __builtin_bit_cast(bf16x8, v), rb = __builtin_bit_cast(bf16x8, r), gb = __builtin_bit_cast(bf16x8, g);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float k = (!gated || (float)gb[j] > 0.f) ? 1.f : LRELU_SLOPE;
        float f = (float)vb[j] + (float)rb[j] * k;
        if (act) f = f > 0.f ? f : LRELU_SLOPE * f;
        o[j] = (__bf16)f;
    }
    return __builtin_bit_cast(u32x4, o);
}

// values of one staged 16-byte piece as floats (8 for bf16, 4 for f32)
template <typename T> __device__ __forceinline__ void piece_values(u32x4 v, float (&f)[16 / sizeof(T)]);
template <> __device__ __forceinline__ void piece_values<float>(u32x4 v, float (&f)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) f[j] = __uint_as_float(v[j]);
}
template <> __device__ __forceinline__ void piece_values<bf16>(u32x4 v, float (&f)[8]) {
    const bf16x8 b = __builtin_bit_cast(bf16x8, v);
#pragma unroll
    for (int j = 0; j < 8; ++j) f[j] = (float)b[j];
}

// ======================================================================================
// forward
// ======================================================================================
// WALK: the workgroup goes on to further image groups (see PatchFwdArgs::gstride); the weight ring and
// the next patches then stay live across the epilogue, which is arranged to need fewer registers.
// RES > 0: the layer has RES chunks and the wave keeps ALL of its weight fragments (RES * TU steps) in registers for
// the whole launch instead of streaming them through the ring: for short-K layers whose workgroups walk over many
// tiles (the tail's forward pass: 7 row taps x 64 channels = 14 steps, 112 registers).
template <typename T, int WM, int WN, int MI, int NI, int KWT, bool WALK = false, int RES = 0>
__global__ void __launch_bounds__(64 * WM * WN, (sizeof(T) == 2 && NI == 1) ? 2 : 1)     // waves per SIMD
patch_fwd(const PatchFwdArgs a) {
    constexpr int ES = (int)sizeof(T);
    constexpr int CPP = CK * ES / 16;            // 16-byte pieces per patch pixel
    constexpr int PITCH = CK * ES + 16;          // conflict-free ds_read_b128 across pixels
    constexpr int NT = 64 * WM * WN;             // 4 waves, or 2 (then four workgroups share a CU)
    // patch pixels a workgroup can stage (the walking four-wave form has no registers for more than six pieces a thread)
    constexpr int MAXPATCH = NT == 128 ? 192 : ((WALK && NT == 256 && !RES) ? 384 : 512);
    constexpr int NIT = (MAXPATCH * CPP + NT - 1) / NT;                           // patch pieces per thread
    constexpr int WTM = MI * 32, WTN = NI * 32, BN = WN * WTN;
    constexpr int FRAG = 64 * 8;                 // elements per packed fragment (one wave-load)
    // Taps unrolled per loop body = weight steps in flight per wave: all 9 taps of a 3x3 kernel,
    // one tap row of a 7x7 kernel.  The body is branch-free and every load in it is
    // unconditional, so the compiler can count outstanding loads exactly and a step only waits
    // for its own fragments (issued TU steps, i.e. > 2000 cycles of MFMA work, earlier).
    // KWT: 3 = 3x3, 7 = 7x7, 1 = 7x1 (seven vertical taps; the stem and the tail's input gradient,
    // whose 8-pixel x 4-channel windows play the part of 32 "channels")
    constexpr int TU = KWT == 1 ? 7 : ((KWT == 3 && sizeof(T) == 2) ? 9 : KWT);   // f32 fragments are twice as wide: one tap row
    typedef typename Elem<T>::frag frag;
    static_assert(NT == 512 || NT == 256 || NT == 128, "8, 4 or 2 waves");
    static_assert(NIT * NT >= MAXPATCH * CPP, "patch pieces");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const TileGeom& t = a.t;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, h = lane >> 5;
    const int tile_y = fdiv(blockIdx.x, t.div_tx), tile_x = blockIdx.x - tile_y * t.tiles_x;
    const int y0 = tile_y * t.TH, x0 = tile_x * t.TW, b0 = blockIdx.z * t.TB, n0 = blockIdx.y * BN;
    // 64-bit base of this workgroup's first image; every offset below is relative to it and fits 32 bits
    // (at most TB images), so a batch may hold more than 2^31 elements
    const T* __restrict__ xp = (const T*)a.x + (size_t)b0 * a.x_img_pitch;
    const int my_groups = WALK ? (a.groups - (int)blockIdx.z + a.gstride - 1) / a.gstride : 1;     // >= 1
    const size_t xstep = (size_t)a.gstride * t.TB * a.x_img_pitch;
#ifdef EXP_TIMING
    long long tt[5], tsum[4] = {0, 0, 0, 0};     // phase stamps of the current image group; sums over the groups
    const long long wall0 = wall_clock64();
    const long long tstart = __builtin_readcyclecounter();
    tt[0] = tstart;
#endif

    // ---- weight fragment streams (one per 32-row block of this wave).  The ring of the first TU
    // steps is requested before anything else: its 2*TU*NI KiB per wave take the texture path a few
    // thousand cycles, which the address arithmetic below fills.
    const int nsteps = a.nchunks * a.T;          // (chunk, tap) steps; two k16 fragments each
    const T* wptr[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
        wptr[ni] = (const T*)a.wf + ((size_t)((n0 + wn * WTN) / 32 + ni) * nsteps * 2) * FRAG + lane * 8;
    constexpr int NWR = RES ? RES * TU : TU;
    frag wr[NWR][NI][2];
    auto load_w = [&](int step, frag (&w)[NI][2]) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
            w[ni][0] = *(const frag*)(wptr[ni] + (size_t)(step * 2) * FRAG);
            w[ni][1] = *(const frag*)(wptr[ni] + (size_t)(step * 2 + 1) * FRAG);
        }
    };
#pragma unroll
    for (int j = 0; j < NWR; ++j) load_w(j, wr[j]);         // nsteps >= TU for every instantiated tap count (== NWR if RES)
    int wnext = TU < nsteps ? TU : 0;            // next step to request; wraps, so the ring runs on into the next image group

    // ---- patch pieces of this thread: global element offset at chunk 0; padding pieces read
    // element 0 and are zeroed by `pmask` afterwards (no branch around the loads)
    int gofs[NIT];
    unsigned pmask = 0;
    const int nitems = a.patch_px * CPP;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int idx = tid + NT * i;
        gofs[i] = 0;
        if (idx < nitems) {
            const int p = idx / CPP, ci = idx % CPP;
            const int tb = fdiv(p, t.div_phw);
            const int r = p - tb * (t.PH * t.PW);
            const int py = fdiv(r, t.div_pw), px = r - py * t.PW;
            int iy = y0 + py + a.off_y, ix = x0 + px + a.off_x;
            const int b = b0 + tb;
            if (b < a.B && pad_xy(a.pad_mode, a.H, a.W, iy, ix)) {
                gofs[i] = tb * a.x_img_pitch + iy * a.x_row_pitch + ix * a.x_pix_pitch + ci * (16 / ES);
                pmask |= 1u << i;
            }
        }
    }
    u32x4 pv[NIT];
    // Patch requests run two chunks ahead of the multiplications along the sequence (group 0: chunks
    // 0..nchunks-1, group 1: ...); past the end they repeat the last chunk instead of branching.
    const T* __restrict__ pf_ptr = xp;
    int pf_chunk = 0, pf_groups_left = my_groups - 1;
    auto load_patch = [&]() {
#pragma unroll
        for (int i = 0; i < NIT; ++i) pv[i] = *(const u32x4*)(pf_ptr + (size_t)gofs[i] + pf_chunk * CK);
        if (pf_chunk + 1 < a.nchunks) {
            ++pf_chunk;
        } else if (pf_groups_left > 0) {
            pf_chunk = 0; pf_ptr += xstep; --pf_groups_left;
        }
    };
    load_patch();                                // in flight during the rest of the set-up
    // Branch-free on purpose: a store inside a conditional block puts the wait for its load inside
    // that block too, and the loop then has to drain every outstanding load (the whole weight ring)
    // before the next patch load may reuse the registers.  Pieces beyond the patch go to the 16
    // padding bytes of patch pixel 0, which nothing reads.
    int sofs[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int idx = tid + NT * i;
        sofs[i] = idx < nitems ? (idx / CPP) * PITCH + (idx % CPP) * 16 : CK * ES;
    }
    auto store_patch = [&](char* buf) {
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            u32x4 v = pv[i];
            if (!((pmask >> i) & 1u)) v = u32x4{0u, 0u, 0u, 0u};
            *(u32x4*)(buf + sofs[i]) = v;
        }
    };

    // ---- this lane's pixel columns
    int abase[MI];
    const int lpx = lane_pixel(l31, a.lane_perm);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        int q = wm * WTM + mi * 32 + lpx;
        if (q >= t.npx) q = 0;
        int tb, ty, tx;
        tile_px_fwd(t, q, tb, ty, tx);
        abase[mi] = ((tb * t.PH + ty) * t.PW + tx) * PITCH + 8 * h * ES;
    }

    f32x16 acc[NI][MI];
#ifdef EXP_TIMING
    long long tp[3];
    tp[0] = __builtin_readcyclecounter();
    tp[1] = __builtin_readcyclecounter();
#endif
    store_patch(smem);
#ifdef EXP_TIMING
    tp[2] = __builtin_readcyclecounter();
#endif
    __syncthreads();
#ifdef EXP_TIMING
    tt[1] = __builtin_readcyclecounter();
#endif
    load_patch();

    // groups of TU taps per chunk: a compile-time count, so the compiler knows how many weight loads
    // are younger than the patch loads when store_patch waits for them (with a run-time count it has
    // to assume none and drains the whole ring at every chunk boundary)
    constexpr int NGRP = (KWT == 1 ? 7 : KWT * KWT) / TU;
    int bsel = 0;                                // patch buffer of the chunk being multiplied
    // A walking workgroup keeps its output channels: their bias values are fetched once, here, and the
    // epilogue has no load of its own in front of the staging (a load there is the youngest one in flight,
    // so waiting for it would also wait for the patches and weights requested for the next image group).
    // (the resident-weights form has no registers for them, and its layers carry no bias)
    constexpr bool BPRE = WALK && !RES;
    float bpre[NI][4][4];
    if (BPRE) {
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int n = n0 + wn * WTN + ni * 32 + 8 * g4 + 4 * h;
#pragma unroll
                for (int j = 0; j < 4; ++j) bpre[ni][g4][j] = (a.bias && n < a.N) ? a.bias[n + j] : 0.f;
            }
    }
    for (int gi = 0; gi < my_groups; ++gi) {
    const int bcur = b0 + gi * a.gstride * t.TB;     // first image of this group
#ifdef EXP_TIMING
    if (gi > 0) tt[0] = tt[1] = __builtin_readcyclecounter();
#endif
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ni][mi][r] = 0.f;
    auto chunk_body = [&](const int wbase) __attribute__((always_inline)) {
        const char* buf = smem + bsel * a.patch_bytes;
#pragma unroll 1
        for (int grp = 0; grp < NGRP; ++grp) {
            // PIPE (one wave per SIMD, registers to spare): the LDS reads of step j+1 are issued
            // before the MFMAs of step j, so the matrix pipe never waits for LDS inside a group.
            constexpr bool PIPE = NI >= 2;
            frag xf[2][2][MI];
            auto read_x = [&](int j, frag (&x)[2][MI]) {
                const int kh = KWT == 1 ? j : (TU == 9 ? j / 3 : grp), kw = KWT == 1 ? 0 : (TU == 9 ? j % 3 : j);
                const int tapoff = (kh * t.PW + kw) * PITCH;
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        x[sub][mi] = *(const frag*)(buf + abase[mi] + tapoff + sub * 16 * ES);
            };
            if (PIPE) read_x(0, xf[0]);
#pragma unroll
            for (int j = 0; j < TU; ++j) {
                if (PIPE) {
                    if (j + 1 < TU) read_x(j + 1, xf[(j + 1) & 1]);
                } else {
                    // all LDS reads of the step go out back to back into their own registers, then
                    // the MFMAs run: one LDS round trip per step, hidden by the SIMD's other wave
                    // (EXP_NOX, EXP_NOW, EXP_NOP, EXP_NOBAR: timing-only ablation builds, tools/ablate_patch_fwd.sh)
#ifdef EXP_NOX
                    if (j == 0 && grp == 0) { read_x(0, xf[0]); read_x(1, xf[1]); }
#else
                    read_x(j, xf[j & 1]);
#endif
                    __builtin_amdgcn_sched_barrier(0);
                }
                // the wave that has its fragments runs its MFMAs ahead of the other wave's address arithmetic and
                // loads (measured: 27.8 -> 27.6 ms on the 128 x 128 inference pass, nothing on the 64 x 64 train step;
                // priority 3 the same, raising it around the LDS reads instead nothing)
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                        for (int mi = 0; mi < MI; ++mi) mma32(acc[ni][mi], wr[wbase + j][ni][sub], xf[j & 1][sub][mi]);
                __builtin_amdgcn_s_setprio(0);
#ifdef EXP_NOW
                if (false) {
#else
                if (!RES) {
#endif
                    load_w(wnext, wr[j]);
                    wnext = wnext + 1 < nsteps ? wnext + 1 : 0;
                }
                // keep each step's LDS reads and MFMAs together: without this the scheduler hoists
                // the reads of all TU steps to the top of the body and runs out of registers
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // chunk boundary: publish the next patch (the next group's first one after the last chunk; a
        // repeat of the last one at the very end), fetch the one after it
        bsel ^= 1;
        store_patch(smem + bsel * a.patch_bytes);
#ifndef EXP_NOBAR
        __syncthreads();
#endif
#ifndef EXP_NOP
        load_patch();
#endif
    };
    if (RES) {
#pragma unroll
        for (int chunk = 0; chunk < (RES ? RES : 1); ++chunk) chunk_body(chunk * TU);
    } else {
        for (int chunk = 0; chunk < a.nchunks; ++chunk) chunk_body(0);
    }

    // ---- epilogue.  D[row = n][col = pixel]: a lane holds 4 consecutive channels of one pixel, so
    // direct stores would scatter 8-byte pieces over 32 cache lines per instruction.  The tile goes
    // through LDS instead and leaves as whole pixel rows, 16 bytes per lane, consecutive lanes on
    // consecutive addresses.  With one image group per workgroup the staging area lies over the patch
    // buffers (free: every wave passed the barrier that follows its last read); a workgroup that goes on
    // to another group has the next patch in one of them already and stages behind them (stage_off).
#ifdef EXP_TIMING
    tt[2] = __builtin_readcyclecounter();
#endif
    constexpr int OP = BN * ES + 16;             // staged row pitch
    // Mapping of the second half (below): thread -> (16-byte piece pc of the staged row, PT consecutive
    // tile pixels).  It is set up here, before the staging, so that a residual term (res *
    // LeakyReLU'(gate): the skip branch of a res block's input gradient) can be requested now, as whole
    // 16-byte pieces on consecutive lanes, and arrive while the accumulators go through LDS.
    // The walking form recomputes the epilogue's thread mapping from an opaque copy of the thread index in
    // every round: hoisted out of the loop over image groups, its values would occupy registers during the
    // multiplications, where none are free.
    // (the lane index comes back from the execution mask and the wave index is wave-uniform, so not even the
    // thread index has to survive the multiplications in a vector register)
    int etid = tid;
    if (WALK) {
        int zero;
        asm volatile("s_mov_b32 %0, 0" : "=s"(zero));
        etid = __builtin_amdgcn_readfirstlane(tid >> 6) * 64 +
               (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, (unsigned)zero));
    }
    const int elane = etid & 63, ewave = etid >> 6, ewm = ewave / WN, ewn = ewave % WN, eh = elane >> 5;
    const int elpx = lane_pixel(elane & 31, a.lane_perm);
    char* const stg = smem + (WALK ? a.stage_off : 0);
    const size_t ybase = (size_t)bcur * a.y_img_pitch;    // res and gate are laid out as y
    T* __restrict__ yp = (T*)a.y + ybase;
    constexpr int PPR = BN * ES / 16;            // 16-byte pieces per staged row
    constexpr int BMT = WM * WTM;
    constexpr int EPP = 16 / ES;                 // elements per piece
    constexpr int PT = BMT * PPR / NT;           // pixels per thread
    const int pc = etid % PPR, q0 = (etid / PPR) * PT;
    const int n_out = n0 + pc * EPP;
    const bool whole = n_out + EPP <= a.N;       // else N is a multiple of 4: half a 16-bit piece
    int yo[PT];                                  // element offset of pixel q0 + i in y, -1: not stored
    if (a.row_fast) {                            // the PT pixels share one output row
        int tb, ty, tx;
        tile_px_fwd(t, q0 < t.npx ? q0 : 0, tb, ty, tx);
        const int b = bcur + tb, oy = y0 + ty, ox = x0 + tx;
        const bool ok = n_out < a.N && q0 < t.npx && b < a.B && oy < a.OH;
        const int step = a.ox_mul * a.y_pix_pitch;
        int yoff = tb * a.y_img_pitch + (oy * a.oy_mul + a.oy_off) * a.y_row_pitch + (ox * a.ox_mul + a.ox_off) * a.y_pix_pitch;
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            yo[i] = (ok && ox + i < a.OW) ? yoff + n_out : -1;
            yoff += step;
        }
    } else {
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int q = q0 + i;
            int tb, ty, tx;
            tile_px_fwd(t, q < t.npx ? q : 0, tb, ty, tx);
            const int b = bcur + tb, oy = y0 + ty, ox = x0 + tx;
            const bool ok = n_out < a.N && q < t.npx && b < a.B && oy < a.OH && ox < a.OW;
            yo[i] = ok ? tb * a.y_img_pitch + (oy * a.oy_mul + a.oy_off) * a.y_row_pitch + (ox * a.ox_mul + a.ox_off) * a.y_pix_pitch + n_out : -1;
        }
    }
    const bool res_mode = !RES && a.res != nullptr;      // with a.scale: y = act(scale * conv + shift + res)
    u32x4 rr[PT], gg[PT];
    auto fetch_res = [&]() __attribute__((always_inline)) {
        const T* __restrict__ resp = (const T*)a.res + ybase;
        auto fetch = [&](const T* __restrict__ src, u32x4 (&dst)[PT]) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < PT; ++i) {       // branch-free: pixels that are not stored read element 0
                const int o = yo[i] >= 0 ? yo[i] : 0;
                if (whole || yo[i] < 0) {
                    dst[i] = *(const u32x4*)(src + o);
                } else {
                    const unsigned long long r8 = *(const unsigned long long*)(src + o);
                    dst[i] = u32x4{(unsigned)r8, (unsigned)(r8 >> 32), 0u, 0u};
                }
            }
        };
        fetch(resp, rr);
        if (a.gate) fetch((const T*)a.gate + ybase, gg);
    };
    if (res_mode && !WALK) fetch_res();          // WALK: after the staging, when the accumulators are dead
    const bool per_pixel = a.cbias != nullptr;   // the class bias needs the pixel's image
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        const int ql = ewm * WTM + mi * 32 + elpx;
        const float* cb = nullptr;
        if (per_pixel) {
            int tb, ty, tx;
            tile_px_fwd(t, ql < t.npx ? ql : 0, tb, ty, tx);
            const int b = bcur + tb;
            if (ql < t.npx && b < a.B) cb = a.cbias + (size_t)a.labels[b] * a.N;
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int nl = ewn * WTN + ni * 32 + 8 * g4 + 4 * eh, n = n0 + nl;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[ni][mi][4 * g4 + j];
                if (n < a.N) {
                    if (BPRE) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] += bpre[ni][g4][j];
                    } else if (a.bias) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] += a.bias[n + j];
                    }
                    if (cb) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] += cb[n + j];
                    }
                    if (a.scale) {               // y = act(scale * conv + shift [+ res: second half])
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = v[j] * a.scale[n + j] + a.shift[n + j];
                        if (a.act && !res_mode) {
#pragma unroll
                            for (int j = 0; j < 4; ++j) v[j] = v[j] > 0.f ? v[j] : LRELU_SLOPE * v[j];
                        }
                    }
                }
                store4<T>((T*)(stg + ql * OP) + nl, v);
            }
        }
    }
    __syncthreads();
    if (WALK) {
        // defined on every path: a value set under a condition only would count as carried from one image
        // group to the next and hold its registers through the multiplications
#pragma unroll
        for (int i = 0; i < PT; ++i) rr[i] = gg[i] = u32x4{0u, 0u, 0u, 0u};
        if (res_mode) fetch_res();
    }
#ifdef EXP_TIMING
    tt[3] = __builtin_readcyclecounter();
    auto finish = [&]() {
        if (!WALK || gi == my_groups - 1) __builtin_amdgcn_s_waitcnt(0);     // (a walking workgroup keeps its requests in flight)
        tt[4] = __builtin_readcyclecounter();
        for (int k = 0; k < 4; ++k) tsum[k] += tt[k + 1] - tt[k];
        if (etid == 0 && gi == my_groups - 1) {
            const size_t lin = blockIdx.x + (size_t)gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z);
            if (lin < 65536) {
                long long c = tstart;
                a.dbg[lin * 5] = c;
                for (int k = 0; k < 4; ++k) { c += tsum[k]; a.dbg[lin * 5 + k + 1] = c; }
                for (int k = 0; k < 2; ++k) a.dbg[65536 * 5 + lin * 3 + k] = tp[k] - tstart;
                a.dbg[65536 * 5 + lin * 3 + 2] = wall_clock64() - wall0;
            }
        }
    };
#endif
    // Second half: whole pixel rows leave LDS, 16 bytes per lane, consecutive lanes on consecutive
    // addresses; the residual term joins here.
    const bool gated = a.gate != nullptr;
    const bool want_stats = a.stat_ws != nullptr;     // BatchNorm batch statistics of this output: sum, sum of squares
    float st_s[EPP], st_q[EPP];
#pragma unroll
    for (int j = 0; j < EPP; ++j) st_s[j] = st_q[j] = 0.f;
#pragma unroll
    for (int i = 0; i < PT; ++i) {
        if (yo[i] < 0) continue;
        u32x4 v = *(const u32x4*)(stg + (q0 + i) * OP + pc * 16);
        if (res_mode) v = add_gated<T>(v, rr[i], gg[i], gated, a.scale && a.act);
        if (want_stats) {                             // of the values as stored (rounded to T), in pixel order
            float f[EPP];
            piece_values<T>(v, f);
#pragma unroll
            for (int j = 0; j < EPP; ++j) { st_s[j] += f[j]; st_q[j] += f[j] * f[j]; }
        }
        if (whole) *(u32x4*)(yp + yo[i]) = v;
        else *(unsigned long long*)(yp + yo[i]) = ((unsigned long long)v[1] << 32) | v[0];
    }
    if (want_stats) {
        // Threads with equal pc hold sums of the same EPP channels: inside a wave they sit PPR lanes apart
        // (butterfly, both partners compute the same sum), the waves meet in LDS and are added in wave
        // order, so a tile's partial does not depend on scheduling.  One partial row per pixel tile:
        // ws[which][tile][channel], tile = blockIdx.x + gridDim.x * image group.
        __shared__ float stat_sm[2][NT / 64][BN];
#pragma unroll
        for (int o = PPR; o < 64; o <<= 1)
#pragma unroll
            for (int j = 0; j < EPP; ++j) {
                st_s[j] += __shfl_xor(st_s[j], o, 64);
                st_q[j] += __shfl_xor(st_q[j], o, 64);
            }
        if (elane < PPR) {
#pragma unroll
            for (int j = 0; j < EPP; ++j) {
                stat_sm[0][ewave][elane * EPP + j] = st_s[j];
                stat_sm[1][ewave][elane * EPP + j] = st_q[j];
            }
        }
        __syncthreads();
        if (etid < BN && n0 + etid < a.N) {
            float s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int w = 0; w < NT / 64; ++w) { s0 += stat_sm[0][w][etid]; s1 += stat_sm[1][w][etid]; }
            const size_t nblk = (size_t)gridDim.x * a.groups, tile = blockIdx.x + (size_t)gridDim.x * (blockIdx.z + gi * a.gstride);
            a.stat_ws[tile * a.N + n0 + etid] = s0;
            a.stat_ws[(nblk + tile) * a.N + n0 + etid] = s1;
        }
    }
#ifdef EXP_TIMING
    finish();
#endif
    // staging area over the second patch buffer: the next group's second chunk is stored there after its first
    // chunk's multiplications, with no barrier in between, so every wave has to be done reading the staged tile
    if (WALK && a.stage_shared) __syncthreads();
    }   // image groups
}

// weights -> fragment order: element ((nblk * nk2*2 + kstep) * 64 + lane) * 8 + j holds
// W[n = nblk*32 + (lane & 31)][tap][c = chunk*32 + (kstep & 1)*16 + 8*(lane >> 5) + j]
template <typename T>
__global__ void patch_pack(const float* __restrict__ canonical, T* __restrict__ packed, const cvg_wmap map, int rows,
                           int nchunks, int ntaps, int Cin) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int nk = nchunks * ntaps * 2;
    if (idx >= (int64_t)rows * nk * 16) return;      // rows/32 * nk * 64 * 8 elements
    const int j = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
    const int64_t f = idx >> 9;
    const int kstep = (int)(f % nk), nblk = (int)(f / nk);
    const int n = nblk * 32 + (lane & 31);
    const int chunk = kstep / (2 * ntaps), tap = (kstep >> 1) % ntaps, sub = kstep & 1;
    const int c = chunk * CK + sub * 16 + 8 * (lane >> 5) + j;
    const int ci = wmap_index(map, n, tap * Cin + c);
    packed[idx] = from_f32<T>(ci >= 0 ? canonical[ci] : 0.f);
}

// ======================================================================================
// forward, seven row taps over 8-pixel x 4-channel windows (K = 224): the stem and the tail's input gradient
// ======================================================================================
// These layers have so little to multiply per output tile (56 MFMAs per wave) that patch_fwd spends two thirds of a
// tile's time outside its main loop (address arithmetic, first patch, weight ring, then the epilogue) and stages every
// 64-byte window separately although neighbouring windows share 56 of their bytes.  Here a workgroup stays resident
// and walks over tiles: its weights (14 fragments per wave) live in registers for the whole launch, the tile's RAW
// pixels ((TH + 6) x (TW + 7) pixels of 8 bytes, a seventh of the window bytes) are staged once, double-buffered,
// with the next tile's pixels requested before the current tile is multiplied, and a window's fragment is two
// 8-byte LDS reads at 8-byte alignment.  Epilogue as patch_fwd's (bias, class bias, LDS-staged row stores); the sums
// are taken in the same order, so the results are bit-identical to patch_fwd's.
struct LowKArgs {
    const void* x; const void* wf; void* y;
    const float* bias; const float* cbias; const int* labels;
    const float* scale; const float* shift; int act;      // eval-mode BatchNorm + LeakyReLU in the epilogue (or null)
    int B, H, W, x_img_pitch, x_row_pitch;     // W: input pixels (of 4 channels) per row
    int off_y, off_x;
    int OH, OW, N;
    int y_img_pitch, y_row_pitch, y_pix_pitch;
    int tiles_y, ntiles;                        // tiles per image group; tiles in all
    int RP, raw_px, raw_bytes;                  // raw row pitch in pixels; staged pixels per tile; bytes of one raw buffer
    TileGeom t;                                 // PH = TH + 6, PW = TW + 7 (raw pixels)
    FastDiv div_tiles;                          // by tiles per image group
#ifdef EXP_TIMING
    long long* dbg;
#endif
};

__global__ void __launch_bounds__(256, 2) lowk_fwd(const LowKArgs a) {
    constexpr int NIT = 4;                      // raw 8-byte pixels per thread (<= 1024 per tile)
    constexpr int OP = 64 * 2 + 16;             // staged output row pitch: 64 channels of bf16 + padding
    constexpr int FRAG = 64 * 8;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const TileGeom& t = a.t;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.y * 64;

    // ---- this wave's weights: 32 rows x 224, resident
    bf16x8 wr[7][2];
    {
        const bf16* wp = (const bf16*)a.wf + ((size_t)(n0 / 32 + wn) * 14) * FRAG + lane * 8;
#pragma unroll
        for (int j = 0; j < 7; ++j)
#pragma unroll
            for (int sub = 0; sub < 2; ++sub) wr[j][sub] = *(const bf16x8*)(wp + (size_t)(j * 2 + sub) * FRAG);
    }
    const bool affine = a.scale != nullptr;      // y = act(scale * (conv + bias terms) + shift), as patch_fwd's epilogue

    // ---- raw pixels of this thread: position inside the patch (the same for every tile)
    // (image, row and column packed into one register each, 10 bits apiece: raw_px <= 1024; 0x3ff images = absent)
    int rpos[NIT], rso[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int idx = tid + 256 * i;
        rpos[i] = 0x3ff << 20; rso[i] = a.raw_bytes - 8;                  // absent: a slot nothing reads
        if (idx < a.raw_px) {
            const int tb = fdiv(idx, t.div_phw);
            const int r = idx - tb * (t.PH * t.PW);
            const int py = fdiv(r, t.div_pw), px = r - py * t.PW;
            rpos[i] = (tb << 20) | (py << 10) | px;
            rso[i] = ((tb * t.PH + py) * a.RP + px) * 8;
        }
    }
    unsigned long long rv[NIT];
    auto tile_origin = [&](int tile, int& b0, int& y0, int& x0) {
        const int g = fdiv(tile, a.div_tiles);
        const int r = tile - g * (a.tiles_y * t.tiles_x);
        const int tyi = fdiv(r, t.div_tx);
        b0 = g * t.TB; y0 = tyi * t.TH; x0 = (r - tyi * t.tiles_x) * t.TW;
    };
    auto load_raw = [&](int tile) {
        int b0, y0, x0;
        tile_origin(tile, b0, y0, x0);
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int rtb = rpos[i] >> 20, b = b0 + rtb, iy = y0 + ((rpos[i] >> 10) & 0x3ff) + a.off_y, ix = x0 + (rpos[i] & 0x3ff) + a.off_x;
            const bool ok = rtb != 0x3ff && b < a.B && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
            const size_t off = ok ? (size_t)b * a.x_img_pitch + (size_t)iy * a.x_row_pitch + (size_t)ix * 4 : 0;
            const unsigned long long v = *(const unsigned long long*)((const bf16*)a.x + off);
            rv[i] = ok ? v : 0ull;
        }
    };

    // ---- this lane's pixel columns inside the raw patch, and the second half's thread mapping (fixed per launch)
    int abase[4];
#pragma unroll
    for (int mi = 0; mi < 4; ++mi) {
        int q = wm * 128 + mi * 32 + l31;
        if (q >= t.npx) q = 0;
        int tb, ty, tx;
        tile_px(t, q, tb, ty, tx);
        abase[mi] = ((tb * t.PH + ty) * a.RP + tx + 2 * h) * 8;
    }
    constexpr int PPR = 8, PT = 8;               // 16-byte pieces per staged row; pixels per thread in the store phase
    const int pc = tid % PPR, q0 = (tid / PPR) * PT;
    const int n_out = n0 + pc * 8;
    const bool whole = n_out + 8 <= a.N;
    char* const stg = smem + 2 * a.raw_bytes;

    int tile = blockIdx.x;
#ifdef EXP_TIMING
    const long long tstart = __builtin_readcyclecounter(), wall0 = wall_clock64();
    long long tsum[4] = {0, 0, 0, 0}, tt[5];
#endif
    if (tile < a.ntiles) load_raw(tile);
    int bsel = 0;
    for (; tile < a.ntiles; tile += gridDim.x) {
        char* buf = smem + bsel * a.raw_bytes;
#ifdef EXP_TIMING
        tt[0] = __builtin_readcyclecounter();
#endif
#pragma unroll
        for (int i = 0; i < NIT; ++i) *(unsigned long long*)(buf + rso[i]) = rv[i];
        __syncthreads();
#ifdef EXP_TIMING
        tt[1] = __builtin_readcyclecounter();
#endif
        int b0, y0, x0;
        tile_origin(tile, b0, y0, x0);
        const int next = tile + gridDim.x;
        load_raw(next < a.ntiles ? next : tile);             // (past the end: the same tile again, unused)
        // class bias rows of this lane's pixels: the label now, the row in the epilogue
        // (one image per tile: one label, fetched now; several images: looked up per pixel in the epilogue)
        const int lab1 = (a.cbias && t.TB == 1) ? a.labels[b0] : -1;
        f32x16 acc[4];
#pragma unroll
        for (int mi = 0; mi < 4; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][r] = 0.f;
        // the window fragments of step s + 1 are requested before the MFMAs of step s: one wave hides its own LDS
        // round trips (left to the compiler, every step waited for its reads: 5.6k cycles per tile for 1.8k of MFMAs)
        bf16x8 xf[2][4];
        auto read_x = [&](int s_, bf16x8 (&x)[4]) __attribute__((always_inline)) {
            const int j = s_ >> 1, sub = s_ & 1;
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) {
                const char* src = buf + abase[mi] + (j * a.RP + sub * 4) * 8;
                const unsigned long long lo = *(const unsigned long long*)src, hi = *(const unsigned long long*)(src + 8);
                const u32x4 v = {(unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32)};
                x[mi] = __builtin_bit_cast(bf16x8, v);
            }
        };
        read_x(0, xf[0]);
#pragma unroll
        for (int s_ = 0; s_ < 14; ++s_) {
            if (s_ + 1 < 14) read_x(s_ + 1, xf[(s_ + 1) & 1]);
#pragma unroll
            for (int mi = 0; mi < 4; ++mi) mma32(acc[mi], wr[s_ >> 1][s_ & 1], xf[s_ & 1][mi]);
            __builtin_amdgcn_sched_barrier(0);
        }
#ifdef EXP_TIMING
        tt[2] = __builtin_readcyclecounter();
#endif
        // ---- epilogue: bias, class bias, through LDS, whole pixel rows out
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) {
            const int ql = wm * 128 + mi * 32 + l31;
            int lab = lab1;
            if (a.cbias && t.TB != 1) {
                int tb, ty, tx;
                tile_px(t, ql < t.npx ? ql : 0, tb, ty, tx);
                lab = (ql < t.npx && b0 + tb < a.B) ? a.labels[b0 + tb] : -1;
            }
            const float* cb = (a.cbias && lab >= 0) ? a.cbias + (size_t)lab * a.N : nullptr;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int nl = wn * 32 + 8 * g4 + 4 * h, n = n0 + nl;
                float v[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) v[jj] = acc[mi][4 * g4 + jj];
                if (n < a.N) {
                    if (a.bias) {                // (fetched here, beside the class bias: no registers held through the MFMAs)
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) v[jj] += a.bias[n + jj];
                    }
                    if (cb) {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) v[jj] += cb[n + jj];
                    }
                    if (affine) {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) v[jj] = v[jj] * a.scale[n + jj] + a.shift[n + jj];
                        if (a.act) {
#pragma unroll
                            for (int jj = 0; jj < 4; ++jj) v[jj] = v[jj] > 0.f ? v[jj] : LRELU_SLOPE * v[jj];
                        }
                    }
                }
                store4<bf16>((bf16*)(stg + ql * OP) + nl, v);
            }
        }
        __syncthreads();
#ifdef EXP_TIMING
        tt[3] = __builtin_readcyclecounter();
#endif
        bf16* __restrict__ yp = (bf16*)a.y + (size_t)b0 * a.y_img_pitch;
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            const int q = q0 + i;
            int tb, ty, tx;
            tile_px(t, q < t.npx ? q : 0, tb, ty, tx);
            const int b = b0 + tb, oy = y0 + ty, ox = x0 + tx;
            if (!(n_out < a.N && q < t.npx && b < a.B && oy < a.OH && ox < a.OW)) continue;
            const u32x4 v = *(const u32x4*)(stg + q * OP + pc * 16);
            bf16* dst = yp + (size_t)tb * a.y_img_pitch + (size_t)oy * a.y_row_pitch + (size_t)ox * a.y_pix_pitch + n_out;
            if (whole) *(u32x4*)dst = v;
            else *(unsigned long long*)dst = ((unsigned long long)v[1] << 32) | v[0];
        }
        bsel ^= 1;
#ifdef EXP_TIMING
        tt[4] = __builtin_readcyclecounter();
        for (int k = 0; k < 4; ++k) tsum[k] += tt[k + 1] - tt[k];
#endif
    }
#ifdef EXP_TIMING
    if (tid == 0) {          // phases: raw store + barrier, multiplications (with the next tile's requests), staging, stores
        const size_t lin = blockIdx.x + (size_t)gridDim.x * blockIdx.y;
        if (lin < 65536) {
            long long c = tstart;
            a.dbg[lin * 5] = c;
            for (int k = 0; k < 4; ++k) { c += tsum[k]; a.dbg[lin * 5 + k + 1] = c; }
            a.dbg[65536 * 5 + lin * 3] = 0; a.dbg[65536 * 5 + lin * 3 + 1] = 0;
            a.dbg[65536 * 5 + lin * 3 + 2] = wall_clock64() - wall0;
        }
    }
#endif
}

// ======================================================================================
// weight gradient
// ======================================================================================
struct PatchWgradArgs {
    const void* x; const void* dy; float* ws;
    int B, H, W, Cin, x_img_pitch, x_row_pitch, x_pix_pitch;
    int off_y, off_x, pad_mode, KW, T;
    int OH, OW, dy_cols;                       // dY[b][oy][ox][n]: pitches below, n < dy_cols readable
    int dy_img_pitch, dy_row_pitch, dy_pix_pitch;
    int patch_px, ntiles, tiles_per_split, tiles_per_group;
    int n_rows, k_ld;
    int nchunks, nchunks_ntiles, xcd_remap;
    TileGeom t;
    FastDiv div_tpg, div_chunks, div_cn;
};

// Transposed operand fetch with per-pixel addresses: slot (h, j) of the fragment is tile
// pixel q0 + 8h + j (bf16) or q0 + 2j + h (f32), whose LDS row address comes from `addr`.
template <typename T> struct TrGather;
template <> struct TrGather<bf16> {
    static constexpr int NADDR = 2;
    typedef __attribute__((address_space(3))) bf16x4* lds_ptr;
    template <typename F>
    __device__ static __forceinline__ bf16x8 ld(F addr, int q0, int c0, int lane) {
        const int i = lane & 15, q = i >> 2, p = i & 3, blk = (lane >> 4) & 1, h = lane >> 5;
        const int cofs = (c0 + 16 * blk + 4 * p) * 2;
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(addr(q0 + 8 * h + q) + cofs));
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(addr(q0 + 8 * h + q + 4) + cofs));
        bf16x8 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) { r[j] = lo[j]; r[4 + j] = hi[j]; }
        return r;
    }
    // channel block 0..31: row addresses (with this lane's channel offset folded in)
    template <typename F>
    __device__ static __forceinline__ void rows(F addr, int q0, int lane, const char* (&out)[2]) {
        const int i = lane & 15, q = i >> 2, p = i & 3, blk = (lane >> 4) & 1, h = lane >> 5;
        const int cofs = (16 * blk + 4 * p) * 2;
        out[0] = addr(q0 + 8 * h + q) + cofs;
        out[1] = addr(q0 + 8 * h + q + 4) + cofs;
    }
    __device__ static __forceinline__ bf16x8 ld_rows(const char* const (&rows_)[2], int off, int lane) {
        bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(rows_[0] + off));
        bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(rows_[1] + off));
        bf16x8 r;
#pragma unroll
        for (int j = 0; j < 4; ++j) { r[j] = lo[j]; r[4 + j] = hi[j]; }
        return r;
    }
};
template <> struct TrGather<float> {
    static constexpr int NADDR = 8;
    template <typename F>
    __device__ static __forceinline__ f32x8 ld(F addr, int q0, int c0, int lane) {
        const int l31 = lane & 31, h = lane >> 5;
        f32x8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = *(const float*)(addr(q0 + 2 * j + h) + (c0 + l31) * 4);
        return r;
    }
    template <typename F>
    __device__ static __forceinline__ void rows(F addr, int q0, int lane, const char* (&out)[8]) {
        const int l31 = lane & 31, h = lane >> 5;
#pragma unroll
        for (int j = 0; j < 8; ++j) out[j] = addr(q0 + 2 * j + h) + l31 * 4;
    }
    __device__ static __forceinline__ f32x8 ld_rows(const char* const (&rows_)[8], int off, int lane) {
        f32x8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = *(const float*)(rows_[j] + off);
        return r;
    }
};

template <typename T, int WN, int WT, int TPW, int NI>
__global__ void __launch_bounds__(256, 1) patch_wgrad(const PatchWgradArgs a) {
    constexpr int ES = (int)sizeof(T);
    constexpr int BNW = 32 * WN * NI;
    constexpr int BMP = ES == 2 ? 256 : 128;                  // tile pixels
    constexpr int NSUB = BMP / 16;                             // 16-pixel MFMA sub-steps per tile
    constexpr int CPP = CK * ES / 16;
    constexpr int PP = CK * ES;                                // patch pixel pitch: 64 B rows are conflict-free for tr reads
    constexpr int GW = BNW * ES;                               // dY row bytes
    constexpr int PG = (GW == 64 || ES == 4) ? GW : GW + 64;
    constexpr int NIT = 8;                                     // patch pieces per thread (512 / 256 px)
    constexpr int GPR = GW / 16;                               // dY pieces per row
    constexpr int GIT = BMP * GPR / 256;                       // dY pieces per thread
    constexpr int NA = TrGather<T>::NADDR;
    typedef typename Elem<T>::frag frag;
    static_assert(WN * WT == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const TileGeom& t = a.t;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave / WT, wt = wave % WT;
    const int l31 = lane & 31, h = lane >> 5;
    // Workgroups are dealt to the 8 XCDs round-robin by launch index.  Renumber them so that each
    // XCD owns one contiguous range of (split, n tile, chunk) with the chunk fastest: the
    // workgroups that read the same dY tile (all chunks of one n tile and split) then run at the
    // same time behind the same L2 instead of fetching it once per XCD.
    int wg = blockIdx.x;
    if (a.xcd_remap) {
        const int nwg = gridDim.x, per = nwg >> 3, rem = nwg & 7, xcd = wg & 7;
        wg = xcd * per + min(xcd, rem) + (wg >> 3);
    }
    const int zq = fdiv(wg, a.div_cn);                        // split
    const int cn = wg - zq * a.nchunks_ntiles;
    const int nt = fdiv(cn, a.div_chunks);
    const int chunk = cn - nt * a.nchunks, n0 = nt * BNW;
    const T* __restrict__ xp = (const T*)a.x;
    const T* __restrict__ gp = (const T*)a.dy;
    char* Ps = smem;
    char* Gs = smem + ((a.patch_px * PP + 255) & ~255);

    const int tl_beg = zq * a.tiles_per_split;
    const int tl_end = min(a.ntiles, tl_beg + a.tiles_per_split);
    const int nitems = a.patch_px * CPP;

    // Per piece, fixed for the whole kernel: element offset relative to the tile origin and the
    // packed (image, row, column) within the patch / tile for the bounds test.  Zero padding tests
    // the coordinates; reflection recomputes the address (slow path, the 7x7 tail only).
    constexpr unsigned NONE = 0xFFFFFFFFu;
    int prel[NIT], grel[GIT];
    unsigned pgeo[NIT], ggeo[GIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int idx = tid + 256 * i;
        prel[i] = 0; pgeo[i] = NONE;
        if (idx < nitems) {
            const int p = idx / CPP, ci = idx % CPP;
            const int tb = fdiv(p, t.div_phw);
            const int rr = p - tb * (t.PH * t.PW);
            const int py = fdiv(rr, t.div_pw), px = rr - py * t.PW;
            prel[i] = tb * a.x_img_pitch + py * a.x_row_pitch + px * a.x_pix_pitch + chunk * CK + ci * (16 / ES);
            pgeo[i] = (unsigned)tb << 16 | (unsigned)py << 8 | (unsigned)px;
        }
    }
#pragma unroll
    for (int i = 0; i < GIT; ++i) {
        const int idx = tid + 256 * i;
        const int q = idx / GPR, ci = idx % GPR;
        grel[i] = 0; ggeo[i] = NONE;
        const int col = n0 + ci * (16 / ES);
        if (q < t.npx && col + (16 / ES) <= a.dy_cols) {
            int tb, ty, tx;
            tile_px(t, q, tb, ty, tx);
            grel[i] = tb * a.dy_img_pitch + ty * a.dy_row_pitch + tx * a.dy_pix_pitch + col;
            ggeo[i] = (unsigned)tb << 16 | (unsigned)ty << 8 | (unsigned)tx;
        }
    }
    const bool reflect = a.pad_mode == CVG_PAD_REFLECT;
    u32x4 pv[NIT], gv[GIT];
    auto load_tile = [&](int tl) {
        const int grp = fdiv(tl, a.div_tpg);
        const int r = tl - grp * a.tiles_per_group;
        const int tile_y = fdiv(r, t.div_tx), tile_x = r - tile_y * t.tiles_x;
        const int y0 = tile_y * t.TH, x0 = tile_x * t.TW, b0 = grp * t.TB;
        const int iy0 = y0 + a.off_y, ix0 = x0 + a.off_x;
        const T* xt = xp + ((ptrdiff_t)b0 * a.x_img_pitch + (ptrdiff_t)iy0 * a.x_row_pitch + (ptrdiff_t)ix0 * a.x_pix_pitch);
        const T* gt = gp + ((ptrdiff_t)b0 * a.dy_img_pitch + (ptrdiff_t)y0 * a.dy_row_pitch + (ptrdiff_t)x0 * a.dy_pix_pitch);
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const unsigned g = pgeo[i];
            const int b = b0 + (int)(g >> 16);
            int iy = iy0 + (int)((g >> 8) & 255u), ix = ix0 + (int)(g & 255u);
            u32x4 v = {0u, 0u, 0u, 0u};
            if (g != NONE && b < a.B) {
                if (reflect) {
                    const int ry = iy, rx = ix;
                    pad_xy(CVG_PAD_REFLECT, a.H, a.W, iy, ix);
                    v = *(const u32x4*)(xt + (prel[i] + (iy - ry) * a.x_row_pitch + (ix - rx) * a.x_pix_pitch));
                } else if ((unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W) {
                    v = *(const u32x4*)(xt + prel[i]);
                }
            }
            pv[i] = v;
        }
#pragma unroll
        for (int i = 0; i < GIT; ++i) {
            const unsigned g = ggeo[i];
            const int b = b0 + (int)(g >> 16), oy = y0 + (int)((g >> 8) & 255u), ox = x0 + (int)(g & 255u);
            u32x4 v = {0u, 0u, 0u, 0u};
            if (g != NONE && b < a.B && oy < a.OH && ox < a.OW) v = *(const u32x4*)(gt + grel[i]);
            gv[i] = v;
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int idx = tid + 256 * i;
            if (idx < nitems) *(u32x4*)(Ps + (idx / CPP) * PP + (idx % CPP) * 16) = pv[i];
        }
#pragma unroll
        for (int i = 0; i < GIT; ++i) {
            const int idx = tid + 256 * i;
            *(u32x4*)(Gs + (idx / GPR) * PG + (idx % GPR) * 16) = gv[i];
        }
    };

    f32x16 acc[NI][TPW];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int tt = 0; tt < TPW; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ni][tt][r] = 0.f;

    // tap range of this wave; taps past the end are clamped to the last one (their products are
    // computed and dropped) so that the inner loop stays branch-free and the LDS reads of all taps
    // of a sub-step are in flight together
    const int tap0 = wt * TPW;
    int toff[TPW];
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
        const int tap = min(tap0 + tt, a.T - 1);
        const int kh = tap / a.KW, kw = tap - kh * a.KW;
        toff[tt] = (kh * t.PW + kw) * PP;
    }
    auto g_addr = [&](int q) -> const char* { return Gs + q * PG; };
    auto p_addr = [&](int q) -> const char* {
        if (q >= t.npx) q = 0;
        int tb, ty, tx;
        tile_px(t, q, tb, ty, tx);
        return Ps + ((tb * t.PH + ty) * t.PW + tx) * PP;
    };
    // LDS row addresses of this lane's pixels at tap (0, 0): the same for every tile, so the tile
    // loop carries no address arithmetic besides one add per tap
    const char* prow[NSUB][NA];
#pragma unroll
    for (int s = 0; s < NSUB; ++s) TrGather<T>::rows(p_addr, s * 16, lane, prow[s]);

    if (tl_beg < tl_end) load_tile(tl_beg);
    for (int tl = tl_beg; tl < tl_end; ++tl) {
        __syncthreads();                     // everyone is done reading the previous tile
        store_tile();
        __syncthreads();
        if (tl + 1 < tl_end) load_tile(tl + 1);
        // all NSUB sub-steps run even for a short tile: its missing pixels have zero dY rows
#pragma unroll
        for (int s = 0; s < NSUB; ++s) {
            frag gf[NI], xf[TPW];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) gf[ni] = TrGather<T>::ld(g_addr, s * 16, (wn * NI + ni) * 32, lane);
#pragma unroll
            for (int tt = 0; tt < TPW; ++tt) xf[tt] = TrGather<T>::ld_rows(prow[s], toff[tt], lane);
#pragma unroll
            for (int tt = 0; tt < TPW; ++tt)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) mma32(acc[ni][tt], gf[ni], xf[tt]);
        }
    }

    // partials [split][n][tap * Cin + chunk*32 + c], D[row = n][col = c]
    float* out = a.ws + (size_t)zq * a.n_rows * a.k_ld;
#pragma unroll
    for (int tt = 0; tt < TPW; ++tt) {
        const int tap = tap0 + tt;
        if (tap >= a.T) continue;
        const int k = tap * a.Cin + chunk * CK + l31;
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + (wn * NI + ni) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                out[(size_t)n * a.k_ld + k] = acc[ni][tt][r];
            }
    }
}


// --------------------------------------------------------------------------------------
// 3x3 kernels in bf16 with >= 64 output channels.  A wave owns 32 output channels x all nine
// taps x one 32-channel chunk (nine accumulators, no idle tap slots); WN waves split the
// output channels, WC waves take neighbouring chunks (WC = 2 for 64-channel layers, so a dY
// tile still feeds four waves).  Tiles are 128 pixels and double-buffered in LDS: the next
// tile is fetched to registers during the MFMAs, stored into the other buffer after them, and
// one barrier per tile separates the generations.
// --------------------------------------------------------------------------------------
__device__ __forceinline__ bf16x8 tr_read2(const char* base, int o0, int o1) {
    typedef __attribute__((address_space(3))) bf16x4* lds_ptr;
    bf16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(base + o0));
    bf16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_ptr)(base + o1));
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 4; ++j) { r[j] = lo[j]; r[4 + j] = hi[j]; }
    return r;
}

// S: stride of the convolution (1 or 2).  With S = 2 the patch of a tile is (2 TH + 1) x (2 TW + 1)
// pixels per image, 4.5 x the tile instead of 1.4 x, but still one staging for all nine taps
// (the gather kernel fetches every tap's operand separately).
template <int WN, int WC, int S = 1>
__global__ void __launch_bounds__(64 * WN * WC, 1) patch_wgrad9(const PatchWgradArgs a) {
    typedef bf16 T;
    constexpr int ES = 2, TAPS = 9;
    constexpr int BNW = 32 * WN;                               // output channels per workgroup
    constexpr int TPX = 128, NSUB = TPX / 16;                  // tile pixels, 16-pixel MFMA sub-steps
    constexpr int CPP = CK * ES / 16;                          // 16-byte pieces per patch pixel and chunk
    constexpr int PP = CK * ES;                                // patch pixel pitch (one chunk)
    constexpr int GW = BNW * ES, PG = GW == 64 ? GW : GW + 64; // dY row bytes / pitch
    constexpr int NT = 64 * WN * WC;                           // threads: 4 waves, or 8 (two per SIMD)
    constexpr bool LEAN = NT == 512;                           // 256 registers per wave: no second fragment set
    constexpr int GPR = GW / 16, GIT = TPX * GPR / NT;         // dY pieces per row / per thread
    constexpr int MAXP = S == 1 ? 256 : 592;                   // patch pixels per chunk the staging registers cover
    constexpr int NIT = (MAXP * CPP * WC + NT - 1) / NT;       // patch pieces per thread
    static_assert(NT == 256 || NT == 512, "4 or 8 waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const TileGeom& t = a.t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave / WC, wc = wave % WC;
    const int l31 = lane & 31, h = lane >> 5;
    int wg = blockIdx.x;
    if (a.xcd_remap) {                                         // see patch_wgrad
        const int nwg = gridDim.x, per = nwg >> 3, rem = nwg & 7, xcd = wg & 7;
        wg = xcd * per + min(xcd, rem) + (wg >> 3);
    }
    const int zq = fdiv(wg, a.div_cn);
    const int cn = wg - zq * a.nchunks_ntiles;
    const int nt = fdiv(cn, a.div_chunks);
    const int cg = cn - nt * a.nchunks, n0 = nt * BNW;         // cg: group of WC chunks
    const T* __restrict__ xp = (const T*)a.x;
    const T* __restrict__ gp = (const T*)a.dy;
    const int chunk_stride = a.patch_px * PP;
    const int patch_bytes = (WC * chunk_stride + 255) & ~255;
    const int buf_bytes = patch_bytes + TPX * PG;

    const int tl_beg = zq * a.tiles_per_split;
    const int tl_end = min(a.ntiles, tl_beg + a.tiles_per_split);
    const int nitems = a.patch_px * CPP * WC;

    // Per piece, fixed for the whole kernel: element offset relative to the tile origin and the
    // packed (image, row, column) within the patch / tile for the bounds test (zero padding only).
    // A tile then costs one scalar origin and a few adds and compares per piece.
    constexpr unsigned NONE = 0xFFFFFFFFu;
    int prel[NIT], grel[GIT];
    unsigned pgeo[NIT], ggeo[GIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int idx = tid + NT * i;
        prel[i] = 0; pgeo[i] = NONE;
        if (idx < nitems) {
            const int p = idx / (CPP * WC), cc = idx % (CPP * WC);           // cc: piece within WC*64 bytes
            const int tb = fdiv(p, t.div_phw);
            const int rr = p - tb * (t.PH * t.PW);
            const int py = fdiv(rr, t.div_pw), px = rr - py * t.PW;
            prel[i] = tb * a.x_img_pitch + py * a.x_row_pitch + px * a.x_pix_pitch + cg * (CK * WC) + cc * (16 / ES);
            pgeo[i] = (unsigned)tb << 16 | (unsigned)py << 8 | (unsigned)px;
        }
    }
#pragma unroll
    for (int i = 0; i < GIT; ++i) {
        const int idx = tid + NT * i;
        const int q = idx / GPR, ci = idx % GPR;
        grel[i] = 0; ggeo[i] = NONE;
        const int col = n0 + ci * (16 / ES);
        if (q < t.npx && col + (16 / ES) <= a.dy_cols) {
            int tb, ty, tx;
            tile_px(t, q, tb, ty, tx);
            grel[i] = tb * a.dy_img_pitch + ty * a.dy_row_pitch + tx * a.dy_pix_pitch + col;
            ggeo[i] = (unsigned)tb << 16 | (unsigned)ty << 8 | (unsigned)tx;
        }
    }
    u32x4 pv[NIT], gv[GIT];
    auto load_tile = [&](int tl) __attribute__((always_inline)) {
        const int grp = fdiv(tl, a.div_tpg);
        const int r = tl - grp * a.tiles_per_group;
        const int tile_y = fdiv(r, t.div_tx), tile_x = r - tile_y * t.tiles_x;
        const int y0 = tile_y * t.TH, x0 = tile_x * t.TW, b0 = grp * t.TB;
        const int iy0 = y0 * S + a.off_y, ix0 = x0 * S + a.off_x;
        const T* xt = xp + ((ptrdiff_t)b0 * a.x_img_pitch + (ptrdiff_t)iy0 * a.x_row_pitch + (ptrdiff_t)ix0 * a.x_pix_pitch);
        const T* gt = gp + ((ptrdiff_t)b0 * a.dy_img_pitch + (ptrdiff_t)y0 * a.dy_row_pitch + (ptrdiff_t)x0 * a.dy_pix_pitch);
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const unsigned g = pgeo[i];
            const int b = b0 + (int)(g >> 16), iy = iy0 + (int)((g >> 8) & 255u), ix = ix0 + (int)(g & 255u);
            u32x4 v = {0u, 0u, 0u, 0u};
            if (g != NONE && b < a.B && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W)
                v = *(const u32x4*)(xt + prel[i]);
            pv[i] = v;
        }
#pragma unroll
        for (int i = 0; i < GIT; ++i) {
            const unsigned g = ggeo[i];
            const int b = b0 + (int)(g >> 16), oy = y0 + (int)((g >> 8) & 255u), ox = x0 + (int)(g & 255u);
            u32x4 v = {0u, 0u, 0u, 0u};
            if (g != NONE && b < a.B && oy < a.OH && ox < a.OW) v = *(const u32x4*)(gt + grel[i]);
            gv[i] = v;
        }
    };
    auto store_tile = [&](char* buf) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int idx = tid + NT * i;
            if (idx < nitems) {
                const int p = idx / (CPP * WC), cc = idx % (CPP * WC);
                *(u32x4*)(buf + (cc / CPP) * chunk_stride + p * PP + (cc % CPP) * 16) = pv[i];
            }
        }
#pragma unroll
        for (int i = 0; i < GIT; ++i) {
            const int idx = tid + NT * i;
            *(u32x4*)(buf + patch_bytes + (idx / GPR) * PG + (idx % GPR) * 16) = gv[i];
        }
    };

    f32x16 acc[TAPS];
#pragma unroll
    for (int tt = 0; tt < TAPS; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tt][r] = 0.f;

    int toff[TAPS];
#pragma unroll
    for (int tt = 0; tt < TAPS; ++tt) toff[tt] = ((tt / 3) * t.PW + (tt % 3)) * PP;

    // LDS offsets (within a buffer) of this lane's rows at tap (0, 0) / of its dY rows, per sub-step:
    // fragment slot (h, j) is tile pixel 16 s + 8 h + j
    // (the eight-wave form recomputes them per sub-step instead of holding 32 registers)
    constexpr int NROW = LEAN ? 1 : NSUB;
    int prow[NROW][2], grow[NROW][2];
    const int lq = (lane & 15) >> 2, cofs = (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;
    auto row_offsets = [&](int s_, int (&pr)[2], int (&gr)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int qq = s_ * 16 + 8 * h + lq + 4 * e;
            int tb, ty, tx;
            tile_px(t, qq < t.npx ? qq : 0, tb, ty, tx);
            pr[e] = wc * chunk_stride + ((tb * t.PH + ty * S) * t.PW + tx * S) * PP + cofs;
            gr[e] = patch_bytes + qq * PG + wn * 64 + cofs;
        }
    };
    if (!LEAN) {
#pragma unroll
        for (int s_ = 0; s_ < NSUB; ++s_) row_offsets(s_, prow[s_ % NROW], grow[s_ % NROW]);
    }

    if (tl_beg < tl_end) {
        load_tile(tl_beg);
        store_tile(smem);
    }
    __syncthreads();
    int cur = 0;
    for (int tl = tl_beg; tl < tl_end; ++tl) {
        const bool more = tl + 1 < tl_end;
        if (more) load_tile(tl + 1);
        const char* buf = smem + cur * buf_bytes;
        constexpr int NSET = LEAN ? 1 : 2;
        bf16x8 gf[NSET], xf[NSET][TAPS];
        auto read_sub = [&](int s_, bf16x8& g, bf16x8 (&x)[TAPS]) __attribute__((always_inline)) {
            if (LEAN) row_offsets(s_, prow[0], grow[0]);
            g = tr_read2(buf, grow[s_ % NROW][0], grow[s_ % NROW][1]);
#pragma unroll
            for (int tt = 0; tt < TAPS; ++tt) x[tt] = tr_read2(buf + toff[tt], prow[s_ % NROW][0], prow[s_ % NROW][1]);
        };
        if (!LEAN) read_sub(0, gf[0], xf[0]);
        // all NSUB sub-steps run even for a short tile: its missing pixels have zero dY rows
#pragma unroll
        for (int s_ = 0; s_ < NSUB; ++s_) {
            if (LEAN) {
                // the SIMD's other wave covers the LDS round trip
                read_sub(s_, gf[0], xf[0]);
            } else if (s_ + 1 < NSUB) {
                read_sub(s_ + 1, gf[(s_ + 1) % NSET], xf[(s_ + 1) % NSET]);
            }
#pragma unroll
            for (int tt = 0; tt < TAPS; ++tt) mma32(acc[tt], gf[s_ % NSET], xf[s_ % NSET][tt]);
            __builtin_amdgcn_sched_barrier(0);
        }
        cur ^= 1;
        if (more) store_tile(smem + cur * buf_bytes);
        __syncthreads();
    }

    // partials [split][n][tap * Cin + chunk*32 + c], D[row = n][col = c]
    float* out = a.ws + (size_t)zq * a.n_rows * a.k_ld;
#pragma unroll
    for (int tt = 0; tt < TAPS; ++tt) {
        const int k = tt * a.Cin + (cg * WC + wc) * CK + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            out[(size_t)n * a.k_ld + k] = acc[tt][r];
        }
    }
}


// ======================================================================================
// input gradient of a 3x3, stride-2, zero-padding-1 convolution: all four output-parity classes in one pass
// ======================================================================================
// dX[2a + py][2b + px] = sum over the taps (kh, kw) with (kh != 1, kw != 1) == (py, px) of
// dY[a + (kh == 0)][b + (kw == 0)] . W[.][.][kh][kw]: class (0, 0) has one tap, (0, 1) and (1, 0) two, (1, 1) four.
// A workgroup owns TB x TH x TW positions (a, b) of the dY grid and stages their (TH + 1) x (TW + 1) patch once per
// 32-channel chunk; the nine taps then read shifted fragments from it, each into the accumulators of its own
// class, so every staged byte and every weight fragment is used exactly once and no MFMA multiplies a structural
// zero (run as four separate convolutions, dY leaves L2 once per class and the one-tap class is a 1 x 1
// convolution that re-stages its input for every 32 x 32 block of work).  Wave tile: 32 channels x 64 positions
// x 4 classes (eight accumulator blocks); the four classes of a position are the 2 x 2 output pixels it owns.
struct S2tArgs {
    const void* x; const void* wf; void* y; const void* gate;
    int B, H, W;                                   // dY grid
    int x_img_pitch, x_row_pitch, x_pix_pitch;
    int N;                                         // channels of the result
    int y_img_pitch, y_row_pitch, y_pix_pitch;     // result [B][2H][2W][N]; gate is laid out the same
    int nchunks, patch_px, patch_bytes, lane_perm, row_fast;
    TileGeom t;
};

template <typename T, int WM, int WN>
__global__ void __launch_bounds__(64 * WM * WN, sizeof(T) == 2 ? 2 : 1)
patch_dgrad_s2(const S2tArgs a) {
    constexpr int ES = (int)sizeof(T);
    constexpr int CPP = CK * ES / 16, PITCH = CK * ES + 16;
    constexpr int NT = 64 * WM * WN;
    constexpr int MI = 2, WTM = MI * 32, BMQ = WM * WTM, BN = WN * 32;
    constexpr int MAXPATCH = 256;                                    // patch pixels a workgroup can stage
    constexpr int NIT = (MAXPATCH * CPP + NT - 1) / NT;
    constexpr int FRAG = 64 * 8;
    constexpr int TU = ES == 2 ? 9 : 3, NGRP = 9 / TU;               // weight steps in flight (see patch_fwd)
    typedef typename Elem<T>::frag frag;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const TileGeom& t = a.t;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, h = lane >> 5;
    const int tile_y = fdiv(blockIdx.x, t.div_tx), tile_x = blockIdx.x - tile_y * t.tiles_x;
    const int y0 = tile_y * t.TH, x0 = tile_x * t.TW, b0 = blockIdx.z * t.TB, n0 = blockIdx.y * BN;
    const T* __restrict__ xp = (const T*)a.x + (size_t)b0 * a.x_img_pitch;

    const int nsteps = a.nchunks * 9;
    const T* wptr = (const T*)a.wf + ((size_t)((n0 + wn * 32) / 32) * nsteps * 2) * FRAG + lane * 8;
    frag wr[TU][2];
    auto load_w = [&](int step, frag (&w)[2]) {
        w[0] = *(const frag*)(wptr + (size_t)(step * 2) * FRAG);
        w[1] = *(const frag*)(wptr + (size_t)(step * 2 + 1) * FRAG);
    };
#pragma unroll
    for (int j = 0; j < TU; ++j) load_w(j, wr[j]);
    int wnext = TU < nsteps ? TU : 0;

    int gofs[NIT], sofs[NIT];
    unsigned pmask = 0;
    const int nitems = a.patch_px * CPP;
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
        const int idx = tid + NT * i;
        gofs[i] = 0;
        sofs[i] = CK * ES;                       // pieces beyond the patch: the padding bytes of patch pixel 0
        if (idx < nitems) {
            const int p = idx / CPP, ci = idx % CPP;
            sofs[i] = p * PITCH + ci * 16;
            const int tb = fdiv(p, t.div_phw);
            const int r = p - tb * (t.PH * t.PW);
            const int py = fdiv(r, t.div_pw), px = r - py * t.PW;
            const int iy = y0 + py, ix = x0 + px, b = b0 + tb;
            if (b < a.B && iy < a.H && ix < a.W) {
                gofs[i] = tb * a.x_img_pitch + iy * a.x_row_pitch + ix * a.x_pix_pitch + ci * (16 / ES);
                pmask |= 1u << i;
            }
        }
    }
    u32x4 pv[NIT];
    int pf_chunk = 0;
    auto load_patch = [&]() {
#pragma unroll
        for (int i = 0; i < NIT; ++i) pv[i] = *(const u32x4*)(xp + (size_t)gofs[i] + pf_chunk * CK);
        if (pf_chunk + 1 < a.nchunks) ++pf_chunk;                    // past the end: the last chunk again
    };
    auto store_patch = [&](char* buf) {
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            u32x4 v = pv[i];
            if (!((pmask >> i) & 1u)) v = u32x4{0u, 0u, 0u, 0u};
            *(u32x4*)(buf + sofs[i]) = v;
        }
    };
    load_patch();

    int abase[MI];
    const int lpx = lane_pixel(l31, a.lane_perm);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        int q = wm * WTM + mi * 32 + lpx;
        if (q >= t.npx) q = 0;
        int tb, ty, tx;
        tile_px(t, q, tb, ty, tx);
        abase[mi] = ((tb * t.PH + ty) * t.PW + tx) * PITCH + 8 * h * ES;
    }

    f32x16 acc[4][MI];
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][mi][r] = 0.f;

    store_patch(smem);
    __syncthreads();
    load_patch();
    int bsel = 0;
    for (int chunk = 0; chunk < a.nchunks; ++chunk) {
        const char* buf = smem + bsel * a.patch_bytes;
#pragma unroll
        for (int grp = 0; grp < NGRP; ++grp) {
#pragma unroll
            for (int j = 0; j < TU; ++j) {
                const int tap = grp * TU + j, kh = tap / 3, kw = tap % 3;          // constants after unrolling
                const int cls = (kh != 1 ? 2 : 0) + (kw != 1 ? 1 : 0);
                const int tapoff = ((kh == 0 ? t.PW : 0) + (kw == 0 ? 1 : 0)) * PITCH;
                frag xf[2][MI];
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
                        xf[sub][mi] = *(const frag*)(buf + abase[mi] + tapoff + sub * 16 * ES);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int sub = 0; sub < 2; ++sub)
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) mma32(acc[cls][mi], wr[j][sub], xf[sub][mi]);
                load_w(wnext, wr[j]);
                wnext = wnext + 1 < nsteps ? wnext + 1 : 0;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        bsel ^= 1;
        store_patch(smem + bsel * a.patch_bytes);
        __syncthreads();
        load_patch();
    }

    // ---- epilogue: one output row parity at a time (two classes = the even and odd columns of those rows)
    // through LDS, staged row 2 q + px for position q, so that consecutive staged rows are consecutive output
    // pixels; whole pixel rows then leave 16 bytes per lane.  The gate pieces (LeakyReLU'(r) of the activation
    // in front of the convolution) are requested before the staging and multiplied on the way out.
    constexpr int OP = BN * ES + 16, PPR = BN * ES / 16, EPP = 16 / ES;
    constexpr int BMT = 2 * BMQ, PT = BMT * PPR / NT;
    static_assert(PT % 2 == 0, "a thread's staged rows start at an even row");
    const int pc = tid % PPR, r0 = (tid / PPR) * PT;
    const int n_out = n0 + pc * EPP;
    const bool col_ok = n_out < a.N;                 // N is a multiple of EPP: pieces are whole
    T* __restrict__ yp = (T*)a.y + (size_t)b0 * a.y_img_pitch;
    const T* __restrict__ gp = a.gate ? (const T*)a.gate + (size_t)b0 * a.y_img_pitch : nullptr;
#pragma unroll
    for (int py = 0; py < 2; ++py) {
        int yo[PT];
        if (a.row_fast) {                            // the PT staged rows of a thread lie in one output row
            const int ql = r0 >> 1;
            int tb, ty, tx;
            tile_px(t, ql < t.npx ? ql : 0, tb, ty, tx);
            const bool ok = col_ok && ql < t.npx && b0 + tb < a.B && y0 + ty < a.H;
            const int yoff = tb * a.y_img_pitch + (2 * (y0 + ty) + py) * a.y_row_pitch + 2 * (x0 + tx) * a.y_pix_pitch + n_out;
#pragma unroll
            for (int i = 0; i < PT; ++i) yo[i] = (ok && x0 + tx + (i >> 1) < a.W) ? yoff + i * a.y_pix_pitch : -1;
        } else {
#pragma unroll
            for (int i = 0; i < PT; ++i) {
                const int r = r0 + i, ql = r >> 1, px = r & 1;
                int tb, ty, tx;
                tile_px(t, ql < t.npx ? ql : 0, tb, ty, tx);
                const bool ok = col_ok && ql < t.npx && b0 + tb < a.B && y0 + ty < a.H && x0 + tx < a.W;
                yo[i] = ok ? tb * a.y_img_pitch + (2 * (y0 + ty) + py) * a.y_row_pitch + (2 * (x0 + tx) + px) * a.y_pix_pitch + n_out : -1;
            }
        }
        u32x4 gg[PT];
        if (gp) {
#pragma unroll
            for (int i = 0; i < PT; ++i) gg[i] = *(const u32x4*)(gp + (yo[i] >= 0 ? yo[i] : 0));
        }
        if (py) __syncthreads();                     // the first half has left the staging area
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
            const int ql = wm * WTM + mi * 32 + lpx;
#pragma unroll
            for (int px = 0; px < 2; ++px)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const int nl = wn * 32 + 8 * g4 + 4 * h;
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = acc[py * 2 + px][mi][4 * g4 + j];
                    store4<T>((T*)(smem + (2 * ql + px) * OP) + nl, v);
                }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PT; ++i) {
            if (yo[i] < 0) continue;
            u32x4 v = *(const u32x4*)(smem + (r0 + i) * OP + pc * 16);
            if (gp) v = mul_gate<T>(v, gg[i]);
            *(u32x4*)(yp + yo[i]) = v;
        }
    }
}

// ---- host side -----------------------------------------------------------------------------
void fill_tile(TileGeom& t, int TB, int TH, int TW, int KH, int KW, int tiles_x, int S = 1) {
    t.TB = TB; t.TH = TH; t.TW = TW; t.PH = (TH - 1) * S + KH; t.PW = (TW - 1) * S + KW;
    t.npx = TB * TH * TW; t.tiles_x = tiles_x; t.pair4 = 0;
    t.div_tw = make_fastdiv(TW); t.div_thw = make_fastdiv(TH * TW);
    t.div_pw = make_fastdiv(t.PW); t.div_phw = make_fastdiv(t.PH * t.PW);
    t.div_tx = make_fastdiv(tiles_x);
}

bool patch_eligible(const cvg_conv_desc* d) {
    if (d->stride != 1 || d->tstep_y != 1 || d->tstep_x != 1) return false;
    if (d->Cin % CK != 0) return false;
    if (d->x_pix_pitch != d->Cin && !(d->Cin == CK && d->pad_mode == CVG_PAD_ZERO)) return false;   // overlapping windows: one chunk
    if (!((d->KW == 3 && d->ntaps == 9) || (d->KW == 7 && d->ntaps == 49) || (d->KW == 1 && d->ntaps == 7)))
        return false;                                                                        // instantiated kernels
    return true;
}

// Offsets are 32-bit relative to a workgroup's (or tile's) first image: the TB images of one tile must fit.
// Tiles of images larger than the tile hold one image (TB = 1), so 256 x 256 and 512 x 512 inputs stay on
// the patch kernels; only a tile spanning many small images could come near the limit.
static bool offsets_fit(const cvg_conv_desc* d, int TB) {
    return (long long)TB * d->x_img_pitch < (1ll << 31) && (long long)TB * d->y_img_pitch < (1ll << 31);
}

// Smallest number of tiles, then the smallest patch, over all TH x TW <= BM.
// whole_groups: a tile whose image count does not divide the batch counts as 4 % more tiles (only whole
// groups let a workgroup walk over several of them).
bool choose_tile(int B, int OH, int OW, int KH, int KW, int BM, int max_patch_px, int& TB, int& TH, int& TW, int S = 1,
                 int whole_groups = 0) {      // > 0: a tiling that leaves a partial last image group counts 1/whole_groups more tiles
    long best_tiles = -1, best_patch = 0;
    for (int th = 1; th <= OH && th <= BM; ++th)
        for (int tw = 1; tw <= OW && th * tw <= BM; ++tw) {
            const int pp = ((th - 1) * S + KH) * ((tw - 1) * S + KW);
            int tb = BM / (th * tw);
            if (tb > B) tb = B;
            if (tb > max_patch_px / pp) tb = max_patch_px / pp;
            if (tb < 1) continue;
            long tiles = (long)ceil_div(OH, th) * ceil_div(OW, tw) * ceil_div(B, tb);
            const long patch = tiles * tb * pp;
            if (whole_groups && B % tb != 0) tiles += tiles / whole_groups;
            if (best_tiles < 0 || tiles < best_tiles || (tiles == best_tiles && patch < best_patch)) {
                best_tiles = tiles; best_patch = patch; TB = tb; TH = th; TW = tw;
            }
        }
    return best_tiles > 0;
}

// Tiling for workgroups that walk over the image groups at a fixed tile position, `slots` of them resident at once
// (two per CU, so the caller's max_patch_px must let two workgroups' buffers fit the LDS): whole groups only.  A
// round takes about the same time whatever the tile's shape (measured on the tail's forward pass: 5.6..6.5 us for
// every tiling tried, at 10 to 16 rounds), so the cost is the number of rounds a workgroup makes, times 1.25 when two
// workgroups share a CU (each then runs slower, but the pair faster than one after the other).  Ties go to fewer
// tiles in all, then to the smaller patch.  False if no tiling walks at least twice with tiles that
// are at least 60 % full.
bool choose_tile_walking(int B, int OH, int OW, int KH, int KW, int BM, int max_patch_px, int slots, int& TB, int& TH, int& TW) {
    long best_cost = -1, best_tiles = 0, best_patch = 0;
    for (int th = 1; th <= OH && th <= BM; ++th)
        for (int tw = 1; tw <= OW && th * tw <= BM; ++tw) {
            const int pp = (th - 1 + KH) * (tw - 1 + KW);
            int tb = BM / (th * tw);
            if (tb > B) tb = B;
            if (tb > max_patch_px / pp) tb = max_patch_px / pp;
            while (tb > 1 && B % tb != 0) --tb;
            if (tb < 1) continue;
            const int fixed = ceil_div(OH, th) * ceil_div(OW, tw), groups = B / tb;
            if (fixed > slots || groups < 2) continue;
            const int stride = slots / fixed < groups ? slots / fixed : groups, rounds = ceil_div(groups, stride);
            if (rounds < 2) continue;
            const long wgs = (long)fixed * ceil_div(groups, rounds);
            const long tiles = (long)fixed * groups, patch = (long)tb * pp;
            const long cost = (long)rounds * (wgs > slots / 2 ? 5 : 4);
            if (10L * B * OH * OW < 6L * tiles * BM) continue;
            if (best_cost < 0 || cost < best_cost || (cost == best_cost && (tiles < best_tiles || (tiles == best_tiles && patch < best_patch)))) {
                best_cost = cost; best_tiles = tiles; best_patch = patch; TB = tb; TH = th; TW = tw;
            }
        }
    return best_cost > 0;
}

// Dynamic LDS beyond 64 KB has to be granted per kernel and per device.  The grants are remembered in a
// small table keyed by (kernel address, device): instantiations of one template share a function-pointer
// TYPE, so the key must be the address.
int set_lds_impl(const void* kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return 0;
    struct Grant { const void* fn; int dev; size_t bytes; };
    constexpr int MAX_GRANTS = 512;
    static Grant table[MAX_GRANTS];
    static int n_grants = 0;
    static std::mutex mu;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) CVG_FAIL("set_lds: no current device");
    std::lock_guard<std::mutex> lock(mu);
    Grant* g = nullptr;
    for (int i = 0; i < n_grants; ++i)
        if (table[i].fn == kernel && table[i].dev == dev) { g = &table[i]; break; }
    if (g && g->bytes >= bytes) return 0;
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) CVG_FAIL("hipFuncSetAttribute(%zu B LDS): %s", bytes, hipGetErrorString(e));
    if (!g && n_grants < MAX_GRANTS) { g = &table[n_grants++]; g->fn = kernel; g->dev = dev; g->bytes = 0; }
    if (g) g->bytes = bytes;             // a full table only costs the repeated attribute call
    return 0;
}
template <typename K>
int set_lds(K kernel, size_t bytes) { return set_lds_impl((const void*)kernel, bytes); }

template <typename T, int WM, int WN, int MI, int NI, int KWT>
int launch_patch_fwd2(const PatchPlan& p, const PatchFwdArgs& a, hipStream_t s) {
    dim3 grid(p.tiles_y * p.tiles_x, ceil_div(a.N, WN * NI * 32), p.gstride);
    if constexpr (sizeof(T) == 2 && WM == 2 && (WN == 4 || WN == 2) && NI == 1 && KWT == 3) {
        if (p.walk) {                            // a workgroup walks over several image groups
            if (set_lds(patch_fwd<T, WM, WN, MI, NI, KWT, true>, p.lds_bytes)) return -1;
            patch_fwd<T, WM, WN, MI, NI, KWT, true><<<grid, 64 * WM * WN, p.lds_bytes, s>>>(a);
            CVG_LAUNCH_CHECK();
            return 0;
        }
    }
    if (p.walk) CVG_FAIL("patch_fwd: no walking form of this kernel");
    if (set_lds(patch_fwd<T, WM, WN, MI, NI, KWT>, p.lds_bytes)) return -1;
    patch_fwd<T, WM, WN, MI, NI, KWT><<<grid, 64 * WM * WN, p.lds_bytes, s>>>(a);
    CVG_LAUNCH_CHECK();
    return 0;
}
template <typename T, int WM, int WN, int MI, int NI>
int launch_patch_fwd(const PatchPlan& p, const PatchFwdArgs& a, hipStream_t s) {
    if (p.KW == 3) return launch_patch_fwd2<T, WM, WN, MI, NI, 3>(p, a, s);
    if (p.KW == 1) return launch_patch_fwd2<T, WM, WN, MI, NI, 1>(p, a, s);
    return launch_patch_fwd2<T, WM, WN, MI, NI, 7>(p, a, s);
}

template <typename T, int WN, int WT, int TPW, int NI>
int launch_patch_wgrad(const PatchWgradPlan& p, const PatchWgradArgs& a, hipStream_t s) {
    if (set_lds(patch_wgrad<T, WN, WT, TPW, NI>, p.lds_bytes)) return -1;
    PatchWgradArgs b = a;
    b.nchunks = p.nchunks; b.nchunks_ntiles = p.nchunks * p.n_tiles;
    b.xcd_remap = cvg_switches().no_xcd_remap ? 0 : 1;
    b.div_chunks = make_fastdiv(p.nchunks); b.div_cn = make_fastdiv(b.nchunks_ntiles);
    const int64_t nwg = (int64_t)b.nchunks_ntiles * p.nsplit;
    if (nwg >= (1ll << 31)) CVG_FAIL("patch_wgrad: too many workgroups");
    patch_wgrad<T, WN, WT, TPW, NI><<<(int)nwg, 256, p.lds_bytes, s>>>(b);
    CVG_LAUNCH_CHECK();
    return 0;
}

}  // namespace

#ifdef EXP_TIMING
#include <vector>
#include <algorithm>
static long long* g_dbg = nullptr;
static long long g_dbg_blocks = 0;
extern "C" void cvg_debug_timing_dump() {
    if (!g_dbg) return;
    (void)hipDeviceSynchronize();
    const long long n = std::min<long long>(g_dbg_blocks, 65536);
    std::vector<long long> h(n * 5);
    (void)hipMemcpy(h.data(), g_dbg, n * 5 * 8, hipMemcpyDeviceToHost);
    long long tmin = h[0], tmax = h[4];
    double ph[4] = {0, 0, 0, 0};
    for (long long i = 0; i < n; ++i) {
        tmin = std::min(tmin, h[i * 5]); tmax = std::max(tmax, h[i * 5 + 4]);
        for (int k = 0; k < 4; ++k) ph[k] += (double)(h[i * 5 + k + 1] - h[i * 5 + k]);
    }
    printf("timing over %lld workgroups: span %lld ticks; mean prologue %.0f  main loop %.0f  stage %.0f  store %.0f  (ticks)\n",
           n, tmax - tmin, ph[0] / n, ph[1] / n, ph[2] / n, ph[3] / n);
    {
        std::vector<long long> q(n * 3);
        (void)hipMemcpy(q.data(), g_dbg + 65536 * 5, n * 3 * 8, hipMemcpyDeviceToHost);
        double m[3] = {0, 0, 0};
        for (long long i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) m[k] += (double)q[i * 3 + k];
        printf("  geometry done at %.0f ticks; workgroup lifetime %.0f wall ticks (100 MHz) = %.2f us -> shader clock %.0f MHz\n", m[0] / n,
               m[2] / n, m[2] / n / 100.0, (ph[0] + ph[1] + ph[2] + ph[3]) / m[2] * 100.0);
    }
    // start-time histogram in 10 bins of the span
    int hist[10] = {0};
    for (long long i = 0; i < n; ++i) hist[std::min<long long>(9, (h[i * 5] - tmin) * 10 / std::max<long long>(1, tmax - tmin))]++;
    printf("start-time histogram:"); for (int k = 0; k < 10; ++k) printf(" %d", hist[k]); printf("\n");
}
#endif

void patch_fwd_plan(const cvg_conv_desc* d, PatchPlan& p) {
    p.ok = 0;
    if (!patch_eligible(d)) return;
    const int es = d->dtype == CVG_BF16 ? 2 : 4;
    p.KW = d->KW; p.KH = d->ntaps / d->KW; p.T = d->ntaps;
    p.BN = d->N > 64 ? 128 : (d->N > 32 ? 64 : 32);
    p.BM = d->N > 64 ? 128 : 256;
    // 64-channel layers move as many bytes as they multiply: four independent two-wave workgroups per
    // CU (128 px x 64 ch each) overlap one tile's loads and stores with another's MFMAs better than two
    // four-wave ones
    const CvgSwitches& sw = cvg_switches();
    int max_patch = 512;
    if (sw.patch_2w && d->dtype == CVG_BF16 && d->N > 32 && d->N <= 64) { p.BM = 128; max_patch = 192; }
    // Layers with two or more 128-channel output tiles run as ONE eight-wave workgroup per CU over 256
    // pixels (wave tile unchanged: 32 ch x 128 px): each weight fragment is fetched from L2 once per
    // 256 pixels and its second reader hits the L1, and each input patch is re-read by half as many
    // workgroups.  Layer by layer -2..-25 % for N >= 256 and +5 % for N = 128 in isolation, but inside
    // the step (weight-gradient stream running beside it) using it for every N > 64 is fastest.
    // CVG_PATCH_WIDE: '0' never, '1' the four-wave form with 64 ch x 128 px wave tiles.
    const char mode = sw.patch_wide ? sw.patch_wide : 'a';
    if (d->dtype == CVG_BF16 && d->N > 64 && mode != '0') p.BM = 256;
    // Walking workgroups (below) where the caller says the launch has the GPU to itself (CVG_CONV_ALONE): layer
    // by layer they are 4..8 % faster where a workgroup has several short tiles (K <= 1152); beside
    // weight-gradient launches on other streams they measured 0.3..0.8 % slower (256 resident workgroups holding
    // 122 KB of LDS each leave those launches no CU to share).
    const bool want_walk = sw.patch_persist ? (sw.patch_persist == '1' || sw.patch_persist == '2') : (d->flags & CVG_CONV_ALONE) != 0;
    // the tail's forward pass (7 row taps x 64 channels, <= 32 output columns) walks too, with its 14 steps of
    // weights resident in registers (template RES); four-wave workgroups, two per CU
    const bool res_rows = d->dtype == CVG_BF16 && p.BN == 32 && p.BM == 256 && p.KW == 1 && p.T == 7 && d->Cin == 2 * CK &&
                          d->x_pix_pitch == d->Cin && !sw.no_res_rows;
    // 64-channel 3x3 layers walk as four-wave workgroups, two per CU: their tiles are short (K = 576: 10.6k cycles of
    // multiplications against 6.2k of set-up, most of it waiting for the first patch) and the next group's patch is
    // already on its way while the current one is multiplied.  Two workgroups fit a CU because the staging area lies
    // over the second patch buffer, which is free by then when the number of chunks is even (`stage_shared`).
    const bool walk64 = (want_walk || sw.patch_persist == '3') && d->dtype == CVG_BF16 && p.BN == 64 && p.BM == 256 && p.KW == 3 && (d->Cin / CK) % 2 == 0 &&
                        !sw.no_walk64;
    const bool may_walk = (want_walk && d->dtype == CVG_BF16 && p.BN == 128 && p.BM == 256 && mode != '1' && p.KW == 3) || res_rows || walk64;
    // (the resident-weights form gains a third, so its tiling is chosen for walking)
    if (!(res_rows && choose_tile_walking(d->B, d->OH, d->OW, p.KH, p.KW, p.BM, 368, 512, p.TB, p.TH, p.TW)) &&
        !choose_tile(d->B, d->OH, d->OW, p.KH, p.KW, p.BM, max_patch, p.TB, p.TH, p.TW, 1, may_walk ? 25 : 0)) return;
    if (sw.tile_set) {   // CVG_PATCH_TILE=th,tw,tb: tile override (benchmarking)
        const int th = sw.tile[0], tw = sw.tile[1], tb = sw.tile[2];
        if (th * tw * tb <= p.BM && tb * (th + p.KH - 1) * (tw + p.KW - 1) <= max_patch) { p.TH = th; p.TW = tw; p.TB = tb; }
    }
    if (!offsets_fit(d, p.TB)) return;
    p.PH = p.TH + p.KH - 1; p.PW = p.TW + p.KW - 1;
    p.tiles_y = ceil_div(d->OH, p.TH); p.tiles_x = ceil_div(d->OW, p.TW); p.groups = ceil_div(d->B, p.TB);
    p.patch_px = p.TB * p.PH * p.PW;
    p.nchunks = d->Cin / CK;
    // use it only when the tiling wastes little of the MFMA work
    const double eff = (double)d->B * d->OH * d->OW / ((double)p.tiles_y * p.tiles_x * p.groups * p.BM);
    if (eff < 0.5 || p.groups > 65535) return;
    const int patch_bytes = (p.patch_px * (CK * es + 16) + 255) & ~255;
    p.lds_bytes = 2 * (size_t)patch_bytes;
    const size_t stage_bytes = (size_t)p.BM * (p.BN * es + 16);     // the output tile is staged over the patch buffers
    if (p.lds_bytes < stage_bytes) p.lds_bytes = stage_bytes;
    p.lds_bytes += (size_t)sw.lds_extra;                           // occupancy experiments
    // One image group per workgroup unless the layer qualifies for the walking form: eight-wave
    // workgroups (one per CU), whole groups only, at most one workgroup per CU in the grid and at least
    // two groups for each.  The workgroup then pays its set-up (address arithmetic, first patch, weight
    // ring: about 5k cycles against 28k of multiplications per 256 x 128 tile at K = 1152) once instead
    // of once per tile.
    p.gstride = p.groups; p.stage_off = 0; p.walk = 0; p.stage_shared = 0;
    const int slots = (res_rows || walk64) ? 512 : 256;                           // workgroups resident at once: one per CU (two of the four-wave form)
    const int fixed = p.tiles_y * p.tiles_x * ceil_div(d->N, p.BN);   // workgroups per image group
    if (may_walk && d->B % p.TB == 0 && fixed <= slots && p.groups >= 2 && (!walk64 || p.patch_px <= 384)) {
        int stride = slots / fixed;
        if (stride > p.groups) stride = p.groups;
        const int walks = ceil_div(p.groups, stride);
        stride = ceil_div(p.groups, walks);                           // same number of rounds, evenly filled
        size_t need = 2 * (size_t)patch_bytes + stage_bytes;
        int stage_off = 2 * patch_bytes;
        if (walk64) {                                                 // staged over the second patch buffer and on behind it
            stage_off = patch_bytes;
            need = (size_t)patch_bytes + (stage_bytes > (size_t)patch_bytes ? stage_bytes : (size_t)patch_bytes);
        }
        if (walks >= 2 && need + 8 * 1024 <= (walk64 ? 80 : 160) * 1024) {      // + the statistics scratch of the kernel
            p.gstride = stride; p.stage_off = stage_off; p.lds_bytes = need; p.walk = 1; p.stage_shared = walk64 ? 1 : 0;
            if (sw.patch_persist == '2') p.gstride = p.groups;        // the walking kernel, one group each (benchmarking)
        }
    }
    p.ok = 1;
}

// The resident seven-row-tap kernel takes the launch when the descriptor is one of its layers (bf16, 8-pixel x
// 4-channel windows, plain output, nothing but bias terms or the folded BatchNorm in the epilogue).  lowk_plan:
// eligibility, tiling and geometry (everything but the pointers); false if the layer stays on patch_fwd.
static bool lowk_plan(const cvg_conv_desc* d, const PatchPlan& p, LowKArgs& a, size_t& lds) {
    if (cvg_switches().no_lowk) return false;
    if (d->dtype != CVG_BF16 || p.KW != 1 || p.T != 7 || d->Cin != 32 || d->x_pix_pitch != 4 || p.nchunks != 1) return false;
    if (d->pad_mode != CVG_PAD_ZERO || d->oy_mul != 1 || d->ox_mul != 1 || d->oy_off != 0 || d->ox_off != 0) return false;
    if (d->N > 65535 * 64 || d->N % 4 != 0 || d->y_pix_pitch % 8 != 0 || d->y_row_pitch % 8 != 0 || d->y_img_pitch % 8 != 0) return false;
    if (d->x_row_pitch % 4 != 0 || d->x_img_pitch % 4 != 0) return false;
    // a tiling of its own: 256 output pixels per tile, fewest tiles, then fewest raw pixels ((TH + 6) x (TW + 7) per image)
    int TB, TH, TW;
    // (one image per tile where an image has 256 pixels or more: tiles of several small pieces of several images
    // save a few tiles and cost more raw pixels, LDS and address arithmetic)
    const int b_tile = (int64_t)d->OH * d->OW >= 256 ? 1 : d->B;
    if (!choose_tile(b_tile, d->OH, d->OW, 7, 8, 256, 1024, TB, TH, TW) || !offsets_fit(d, TB)) return false;
    const int tiles_y = ceil_div(d->OH, TH), tiles_x = ceil_div(d->OW, TW), groups = ceil_div(d->B, TB);
    a.B = d->B; a.H = d->H; a.W = d->W; a.x_img_pitch = d->x_img_pitch; a.x_row_pitch = d->x_row_pitch;
    a.off_y = d->off_y; a.off_x = d->off_x; a.OH = d->OH; a.OW = d->OW; a.N = d->N;
    a.y_img_pitch = d->y_img_pitch; a.y_row_pitch = d->y_row_pitch; a.y_pix_pitch = d->y_pix_pitch;
    fill_tile(a.t, TB, TH, TW, 7, 8, tiles_x);                // PH = TH + 6, PW = TW + 7 raw pixels
    a.tiles_y = tiles_y;
    const int64_t ntiles = (int64_t)tiles_y * tiles_x * groups;
    if (ntiles >= (1ll << 30)) return false;
    a.ntiles = (int)ntiles;
    a.div_tiles = make_fastdiv(tiles_y * tiles_x);
    // raw rows TW + 32 pixels apart: the tile rows that one 32-lane read group covers follow each other in bank
    // space (16-pixel rows 384 bytes apart: two of them fill the 64 banks once)
    a.RP = TW + 32;
    a.raw_px = TB * a.t.PH * a.t.PW;
    if (a.raw_px > 1024) return false;
    a.raw_bytes = (TB * a.t.PH * a.RP * 8 + 8 + 255) & ~255;
    lds = 2 * (size_t)a.raw_bytes + 256 * (64 * 2 + 16);
    return lds <= 72 * 1024;                                   // two workgroups per CU
}

bool lowk_fwd_takes(const cvg_conv_desc* d, const PatchPlan& p) {
    LowKArgs a;
    size_t lds;
    return lowk_plan(d, p, a, lds);
}

// Returns 1 if the resident kernel took the launch, 0 if not, -1 on error.
static int lowk_fwd_try(const cvg_conv_desc* d, const PatchPlan& p, const void* x, const void* w, void* y,
                        const float* bias, const float* cbias, const int32_t* labels, const float* scale,
                        const float* shift, int act, hipStream_t s) {
    LowKArgs a;
    size_t lds;
    if (!lowk_plan(d, p, a, lds)) return 0;
    a.x = x; a.wf = w; a.y = y; a.bias = bias; a.cbias = cbias; a.labels = labels;
    a.scale = scale; a.shift = shift; a.act = act;
    if (set_lds(lowk_fwd, lds)) return -1;
    const int wgs = a.ntiles < 512 ? a.ntiles : 512;
#ifdef EXP_TIMING
    {
        static long long* dbg = nullptr;
        if (!dbg) { if (hipMalloc(&dbg, 65536 * 8 * 8) != hipSuccess) return -1; }
        a.dbg = dbg; g_dbg = dbg; g_dbg_blocks = (long long)wgs * ceil_div(d->N, 64);
    }
#endif           // two resident workgroups per CU walk over the tiles
    lowk_fwd<<<dim3(wgs, ceil_div(d->N, 64)), 256, lds, s>>>(a);
    if (hipGetLastError() != hipSuccess) CVG_FAIL("lowk_fwd launch failed");
    return 1;
}

int patch_fwd_launch(const cvg_conv_desc* d, const PatchPlan& p, const void* x, const void* w, void* y,
                     const float* bias, const float* cbias, const int32_t* labels, const void* res,
                     const void* gate, hipStream_t s, const float* scale, const float* shift, int act, float* stat_ws) {
    if (!res && !gate && !stat_ws) {
        const int got = lowk_fwd_try(d, p, x, w, y, bias, cbias, labels, scale, shift, act, s);
        if (got) return got < 0 ? -1 : 0;
    }
    PatchFwdArgs a;
    a.scale = scale; a.shift = shift; a.act = act; a.stat_ws = stat_ws;
    const int es = d->dtype == CVG_BF16 ? 2 : 4;
    a.x = x; a.wf = w; a.y = y; a.bias = bias; a.cbias = cbias; a.labels = labels; a.res = res; a.gate = gate;
    a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.x_img_pitch = d->x_img_pitch; a.x_row_pitch = d->x_row_pitch; a.x_pix_pitch = d->x_pix_pitch;
    // overlapping windows (pixel pitch < Cin): the last whole window starts Cin/pitch - 1 pixels before the row end
    if (d->x_pix_pitch != d->Cin) a.W = d->W - d->Cin / d->x_pix_pitch + 1;
    a.off_y = d->off_y; a.off_x = d->off_x; a.pad_mode = d->pad_mode; a.KW = p.KW; a.T = p.T;
    a.OH = d->OH; a.OW = d->OW; a.N = d->N;
    a.y_img_pitch = d->y_img_pitch; a.y_row_pitch = d->y_row_pitch; a.y_pix_pitch = d->y_pix_pitch;
    a.oy_mul = d->oy_mul; a.oy_off = d->oy_off; a.ox_mul = d->ox_mul; a.ox_off = d->ox_off;
    a.nchunks = p.nchunks; a.patch_px = p.patch_px;
    a.patch_bytes = (p.patch_px * (CK * es + 16) + 255) & ~255;
    a.groups = p.groups; a.gstride = p.gstride; a.stage_off = p.stage_off; a.stage_shared = p.stage_shared;
    // Each ds_read_b128 lane group then reads 16 CONSECUTIVE tile pixels.  With 16-pixel tile rows that is
    // one row (conflict-free); with other widths the 16 pixels cross a row boundary and two of them collide
    // (2 LDS cycles instead of 1), where the natural lane order spreads a group over 28 consecutive pixels
    // and collides up to three ways: measured -5..-7 % on the 8 x 8 layers and on the decoder's
    // input-gradient layers (10-, 18- and 66-pixel-wide grids), none slower.  CVG_PATCH_PERM=0: as before.
    a.lane_perm = cvg_switches().patch_perm == '0' ? ((p.TW % 16 == 0) ? 1 : 0) : 1;
    {   // pixels per thread in the store phase: BM * (BN * es / 16) / threads
        const bool four = cvg_switches().patch_wide == '1';
        const int threads = (p.BN == 64 && p.BM == 128) ? 128 : ((p.BN == 128 && p.BM == 256 && !four) ? 512 : 256);
        const int pt = p.BM * (p.BN * es / 16) / threads;
        a.row_fast = (p.TW % pt == 0) ? 1 : 0;
    }
#ifdef EXP_TIMING
    {
        static long long* dbg = nullptr;
        if (!dbg) { if (hipMalloc(&dbg, 65536 * 8 * 8) != hipSuccess) return -1; }
        a.dbg = dbg;
        g_dbg = dbg; g_dbg_blocks = (long long)p.tiles_y * p.tiles_x * ceil_div(d->N, p.BN) * p.gstride;
    }
#endif
    fill_tile(a.t, p.TB, p.TH, p.TW, p.KH, p.KW, p.tiles_x);
    a.t.pair4 = (p.TH == 8 && p.TW == 8 && p.PW == 10 && a.lane_perm && cvg_switches().patch_perm != '1') ? 1 : 0;
    // wave tile = 32 output channels x 128 pixels (64 for the narrowest layers): every wave streams
    // its own weight rows, so no weight byte crosses the L1 twice inside a workgroup.
    if (d->dtype == CVG_BF16) {
        if (p.BN == 128 && p.BM == 256) {
            if (cvg_switches().patch_wide == '1') return launch_patch_fwd<bf16, 2, 2, 4, 2>(p, a, s);
            return launch_patch_fwd<bf16, 2, 4, 4, 1>(p, a, s);       // eight waves, 32 ch x 128 px each
        }
        if (p.BN == 128) return launch_patch_fwd<bf16, 1, 4, 4, 1>(p, a, s);
        if (p.BN == 64 && p.BM == 128) return launch_patch_fwd<bf16, 1, 2, 4, 1>(p, a, s);
        if (p.BN == 64) return launch_patch_fwd<bf16, 2, 2, 4, 1>(p, a, s);
        if (p.walk && p.KW == 1) {                   // the tail's forward pass: resident weights, walking (see patch_fwd_plan)
            dim3 grid(p.tiles_y * p.tiles_x, 1, p.gstride);
            if (set_lds(patch_fwd<bf16, 4, 1, 2, 1, 1, true, 2>, p.lds_bytes)) return -1;
            patch_fwd<bf16, 4, 1, 2, 1, 1, true, 2><<<grid, 256, p.lds_bytes, s>>>(a);
            CVG_LAUNCH_CHECK();
            return 0;
        }
        return launch_patch_fwd<bf16, 4, 1, 2, 1>(p, a, s);
    }
    if (p.BN == 128) return launch_patch_fwd<float, 1, 4, 4, 1>(p, a, s);
    if (p.BN == 64) return launch_patch_fwd<float, 2, 2, 4, 1>(p, a, s);
    return launch_patch_fwd<float, 4, 1, 2, 1>(p, a, s);
}

int patch_pack_launch(const cvg_conv_desc* d, const PatchPlan& p, const cvg_wmap* map, const float* canonical,
                      void* packed, hipStream_t s) {
    const int64_t n = (int64_t)d->w_rows * p.nchunks * p.T * 2 * 16;
    CVG_CHECK((int64_t)p.nchunks * p.T * CK <= d->w_ld, "patch_pack: packed buffer too small");
    if (d->dtype == CVG_BF16)
        patch_pack<bf16><<<ceil_div(n, 256), 256, 0, s>>>(canonical, (bf16*)packed, *map, d->w_rows, p.nchunks, p.T, d->Cin);
    else
        patch_pack<float><<<ceil_div(n, 256), 256, 0, s>>>(canonical, (float*)packed, *map, d->w_rows, p.nchunks, p.T, d->Cin);
    CVG_LAUNCH_CHECK();
    return 0;
}

// workgroups one weight-gradient launch aims for (the split count follows from it): every CU, or its share
// of them when the caller runs co_launches such launches side by side.  Two launches over 128 CUs each
// write and re-read half the partial slabs of two launches over 256 CUs one after the other.
// CVG_WGRAD_WGS overrides (benchmarking).
static int wgrad_target_wgs(const cvg_conv_desc* d) {
    const int forced = cvg_switches().wgrad_wgs;
    if (forced >= 16) return forced;
    const int share = d->co_launches > 1 ? d->co_launches : 1;
    return 256 / (share > 8 ? 8 : share);
}

void patch_wgrad_plan(const cvg_conv_desc* d, PatchWgradPlan& p) {
    p.ok = 0;
    p.stride = 1;
    const bool no_s2 = cvg_switches().wgrad_s2_gather;
    if (!no_s2 && d->stride == 2 && d->tstep_y == 1 && d->tstep_x == 1 && d->dtype == CVG_BF16 && d->KW == 3 &&
        d->ntaps == 9 && d->N > 64 && d->Cin % CK == 0 && d->x_pix_pitch == d->Cin && d->pad_mode == CVG_PAD_ZERO &&
        choose_tile(d->B, d->OH, d->OW, 3, 3, 128, 592, p.TB, p.TH, p.TW, 2) && offsets_fit(d, p.TB)) {
        // the nine-tap kernel, four waves, one 32-channel chunk per workgroup
        p.KW = 3; p.KH = 3; p.T = 9; p.WN = 4; p.WT = 1; p.BNW = 128;
        p.PH = 2 * p.TH + 1; p.PW = 2 * p.TW + 1;
        p.tiles_y = ceil_div(d->OH, p.TH); p.tiles_x = ceil_div(d->OW, p.TW); p.groups = ceil_div(d->B, p.TB);
        p.ntiles = p.tiles_y * p.tiles_x * p.groups;
        p.patch_px = p.TB * p.PH * p.PW;
        p.nchunks = d->Cin / CK;
        p.n_tiles = ceil_div(d->N, p.BNW);
        const double eff = (double)d->B * d->OH * d->OW / ((double)p.ntiles * 128);
        int want = ceil_div(wgrad_target_wgs(d), p.nchunks * p.n_tiles);
        if (want > p.ntiles) want = p.ntiles;
        if (want < 1) want = 1;
        p.tiles_per_split = ceil_div(p.ntiles, want);
        p.nsplit = ceil_div(p.ntiles, p.tiles_per_split);
        p.lds_bytes = 2 * ((size_t)((p.patch_px * CK * 2 + 255) & ~255) + (size_t)128 * (128 * 2 + 64));
        p.ws_bytes = (size_t)p.nsplit * (p.n_tiles * p.BNW) * ((size_t)p.T * d->Cin) * sizeof(float);
        if (eff >= 0.5 && p.lds_bytes <= 160 * 1024) { p.nine = 1; p.stride = 2; p.ok = 1; return; }
    }
    if (!patch_eligible(d)) return;
    const int es = d->dtype == CVG_BF16 ? 2 : 4;
    p.KW = d->KW; p.KH = d->ntaps / d->KW; p.T = d->ntaps;
    p.nine = 0;
    const bool no_nine = cvg_switches().wgrad_v1;
    if (!no_nine && d->dtype == CVG_BF16 && d->KW == 3 && d->ntaps == 9 && d->N > 32 && d->x_pix_pitch == d->Cin &&
        d->pad_mode == CVG_PAD_ZERO) {
        // chunks per workgroup: two, with eight waves for wide layers (a dY tile then feeds twice the
        // MFMA work per byte fetched) and four waves for 64-channel ones
        const bool no8 = cvg_switches().wgrad_4w;
        const int wc = (d->N > 64 && (no8 || d->Cin % (2 * CK) != 0)) ? 1 : 2;
        const int wn9 = d->N > 64 ? 4 : 2;
        if (d->Cin % (CK * wc) == 0 &&
            choose_tile(d->B, d->OH, d->OW, 3, 3, 128, 256, p.TB, p.TH, p.TW) && offsets_fit(d, p.TB)) {
            p.WN = wn9; p.WT = wc; p.BNW = 32 * p.WN;
            p.PH = p.TH + 2; p.PW = p.TW + 2;
            p.tiles_y = ceil_div(d->OH, p.TH); p.tiles_x = ceil_div(d->OW, p.TW); p.groups = ceil_div(d->B, p.TB);
            p.ntiles = p.tiles_y * p.tiles_x * p.groups;
            p.patch_px = p.TB * p.PH * p.PW;
            p.nchunks = d->Cin / (CK * wc);                                   // chunk groups
            p.n_tiles = ceil_div(d->N, p.BNW);
            const double eff9 = (double)d->B * d->OH * d->OW / ((double)p.ntiles * 128);
            int want9 = ceil_div(wgrad_target_wgs(d), p.nchunks * p.n_tiles);
            if (want9 > p.ntiles) want9 = p.ntiles;
            if (want9 < 1) want9 = 1;
            p.tiles_per_split = ceil_div(p.ntiles, want9);
            p.nsplit = ceil_div(p.ntiles, p.tiles_per_split);
            const int gw = p.BNW * 2, pg = gw == 64 ? gw : gw + 64;
            p.lds_bytes = 2 * ((size_t)((wc * p.patch_px * CK * 2 + 255) & ~255) + (size_t)128 * pg);
            p.ws_bytes = (size_t)p.nsplit * (p.n_tiles * p.BNW) * ((size_t)p.T * d->Cin) * sizeof(float);
            if (eff9 >= 0.5 && p.lds_bytes <= 160 * 1024) { p.nine = 1; p.ok = 1; return; }
        }
    }
    // waves: WN over 32*NI-channel blocks x WT over tap groups.  Wide layers give every wave two
    // channel blocks, so each patch fragment read from LDS feeds two MFMAs.
    int ni = 1;
    if (d->N > 64) { p.WN = 2; p.WT = 2; ni = 2; }
    else if (d->N > 32) { p.WN = 2; p.WT = 2; }
    else { p.WN = 1; p.WT = 4; }
    p.BNW = 32 * p.WN * ni;
    const int tpw = p.T == 7 ? (p.WN == 1 ? 3 : 5) : ceil_div(p.T, p.WT);   // 7 taps ride the 3- / 5-tap kernels
    if (!(tpw == 5 || tpw == 3 || tpw == 13)) return;                        // instantiated tap counts
    const int bmp = es == 2 ? 256 : 128;
    if (!choose_tile(d->B, d->OH, d->OW, p.KH, p.KW, bmp, 2 * bmp, p.TB, p.TH, p.TW)) return;
    if (!offsets_fit(d, p.TB)) return;
    p.PH = p.TH + p.KH - 1; p.PW = p.TW + p.KW - 1;
    p.tiles_y = ceil_div(d->OH, p.TH); p.tiles_x = ceil_div(d->OW, p.TW); p.groups = ceil_div(d->B, p.TB);
    p.ntiles = p.tiles_y * p.tiles_x * p.groups;
    p.patch_px = p.TB * p.PH * p.PW;
    p.nchunks = d->Cin / CK;
    p.n_tiles = ceil_div(d->N, p.BNW);
    const double eff = (double)d->B * d->OH * d->OW / ((double)p.ntiles * bmp);
    if (eff < 0.5) return;
    // one workgroup per CU fits (accumulators take the whole register file): aim for ~256 of them,
    // which also halves the partial-slab traffic of the reduce pass compared with two rounds
    int want = ceil_div(wgrad_target_wgs(d), p.nchunks * p.n_tiles);
    if (want > p.ntiles) want = p.ntiles;
    if (want < 1) want = 1;
    p.tiles_per_split = ceil_div(p.ntiles, want);
    p.nsplit = ceil_div(p.ntiles, p.tiles_per_split);
    if (p.nsplit > 65535) return;
    const int gw = p.BNW * es;
    const int pg = (gw == 64 || es == 4) ? gw : gw + 64;
    p.lds_bytes = (size_t)((p.patch_px * CK * es + 255) & ~255) + (size_t)bmp * pg;
    p.ws_bytes = (size_t)p.nsplit * (p.n_tiles * p.BNW) * ((size_t)p.T * d->Cin) * sizeof(float);
    p.ok = 1;
}

int patch_wgrad_launch(const cvg_conv_desc* d, const PatchWgradPlan& p, const void* x, const void* dy, int dy_ld,
                       int dy_img_pitch, int dy_row_pitch, int dy_cols,
                       const cvg_wmap* map, float* grad, int accumulate, float* ws, hipStream_t s) {
    PatchWgradArgs a;
    a.x = x; a.dy = dy; a.ws = ws;
    a.B = d->B; a.H = d->H; a.W = d->W; a.Cin = d->Cin; a.x_img_pitch = d->x_img_pitch; a.x_row_pitch = d->x_row_pitch; a.x_pix_pitch = d->x_pix_pitch;
    // overlapping windows (pixel pitch < Cin): the last whole window starts Cin/pitch - 1 pixels before the row end
    if (d->x_pix_pitch != d->Cin) a.W = d->W - d->Cin / d->x_pix_pitch + 1;
    a.off_y = d->off_y; a.off_x = d->off_x; a.pad_mode = d->pad_mode; a.KW = p.KW; a.T = p.T;
    a.OH = d->OH; a.OW = d->OW; a.dy_cols = dy_cols;
    a.dy_pix_pitch = dy_ld; a.dy_row_pitch = dy_row_pitch; a.dy_img_pitch = dy_img_pitch;
    a.patch_px = p.patch_px; a.ntiles = p.ntiles; a.tiles_per_split = p.tiles_per_split;
    a.tiles_per_group = p.tiles_y * p.tiles_x;
    a.n_rows = p.n_tiles * p.BNW; a.k_ld = p.T * d->Cin;
    fill_tile(a.t, p.TB, p.TH, p.TW, p.KH, p.KW, p.tiles_x, p.stride);
    a.div_tpg = make_fastdiv(a.tiles_per_group);
    const int tpw = p.T == 7 ? (p.WN == 1 ? 3 : 5) : ceil_div(p.T, p.WT);
    int rc = -1;
    if (p.nine) {
        PatchWgradArgs b = a;
        b.nchunks = p.nchunks; b.nchunks_ntiles = p.nchunks * p.n_tiles;
        b.div_chunks = make_fastdiv(p.nchunks); b.div_cn = make_fastdiv(b.nchunks_ntiles);
        b.xcd_remap = cvg_switches().no_xcd_remap ? 0 : 1;
        const int64_t nwg = (int64_t)b.nchunks_ntiles * p.nsplit;
        if (nwg >= (1ll << 31)) CVG_FAIL("patch_wgrad: too many workgroups");
        if (p.stride == 2) {
            if (set_lds(patch_wgrad9<4, 1, 2>, p.lds_bytes)) return -1;
            patch_wgrad9<4, 1, 2><<<(int)nwg, 256, p.lds_bytes, s>>>(b);
        } else if (p.WT == 1) {
            if (set_lds(patch_wgrad9<4, 1>, p.lds_bytes)) return -1;
            patch_wgrad9<4, 1><<<(int)nwg, 256, p.lds_bytes, s>>>(b);
        } else if (p.WN == 4) {
            if (set_lds(patch_wgrad9<4, 2>, p.lds_bytes)) return -1;
            patch_wgrad9<4, 2><<<(int)nwg, 512, p.lds_bytes, s>>>(b);
        } else {
            if (set_lds(patch_wgrad9<2, 2>, p.lds_bytes)) return -1;
            patch_wgrad9<2, 2><<<(int)nwg, 256, p.lds_bytes, s>>>(b);
        }
        CVG_LAUNCH_CHECK();
        rc = 0;
    } else if (d->dtype == CVG_BF16) {
        if (p.BNW == 128 && tpw == 5) rc = launch_patch_wgrad<bf16, 2, 2, 5, 2>(p, a, s);
        else if (p.WN == 2 && tpw == 5) rc = launch_patch_wgrad<bf16, 2, 2, 5, 1>(p, a, s);
        else if (p.WN == 1 && tpw == 3) rc = launch_patch_wgrad<bf16, 1, 4, 3, 1>(p, a, s);
        else if (p.WN == 1 && tpw == 13) rc = launch_patch_wgrad<bf16, 1, 4, 13, 1>(p, a, s);
        else CVG_FAIL("patch_wgrad: no kernel for WN %d taps/wave %d", p.WN, tpw);
    } else {
        if (p.BNW == 128 && tpw == 5) rc = launch_patch_wgrad<float, 2, 2, 5, 2>(p, a, s);
        else if (p.WN == 2 && tpw == 5) rc = launch_patch_wgrad<float, 2, 2, 5, 1>(p, a, s);
        else if (p.WN == 1 && tpw == 3) rc = launch_patch_wgrad<float, 1, 4, 3, 1>(p, a, s);
        else if (p.WN == 1 && tpw == 13) rc = launch_patch_wgrad<float, 1, 4, 13, 1>(p, a, s);
        else CVG_FAIL("patch_wgrad: no kernel for WN %d taps/wave %d", p.WN, tpw);
    }
    if (rc) return rc;
    launch_wgrad_reduce(ws, grad, map, a.n_rows, a.k_ld, p.nsplit, accumulate, s);
    CVG_LAUNCH_CHECK();
    return 0;
}

// ---- input gradient of the stride-2 convolutions (CVG_FORM_S2_TRANSPOSED) -------------------------------
void patch_dgrad_s2_plan(const cvg_conv_desc* d, S2tPlan& p) {
    p.ok = 0;
    if (d->form != CVG_FORM_S2_TRANSPOSED || d->KW != 3 || d->ntaps != 9) return;
    if (d->dtype != CVG_BF16 && d->dtype != CVG_F32) return;
    const int es = d->dtype == CVG_BF16 ? 2 : 4;
    if (d->Cin % CK != 0 || d->x_pix_pitch != d->Cin || d->N % (16 / es) != 0 || d->y_pix_pitch % (16 / es) != 0 ||
        d->y_row_pitch % (16 / es) != 0 || d->y_img_pitch % (16 / es) != 0) return;
    p.BN = (d->dtype == CVG_BF16 && d->N > 64) ? 128 : 64;
    if (!choose_tile(d->B, d->H, d->W, 2, 2, 128, 256, p.TB, p.TH, p.TW)) return;
    if (!offsets_fit(d, p.TB)) return;
    p.tiles_y = ceil_div(d->H, p.TH); p.tiles_x = ceil_div(d->W, p.TW); p.groups = ceil_div(d->B, p.TB);
    // a grid that leaves CUs without a 128-channel workgroup: 64-channel workgroups (two fit a CU) instead
    if (p.BN == 128 && (int64_t)p.tiles_y * p.tiles_x * p.groups * ceil_div(d->N, 128) < 200) p.BN = 64;
    p.patch_px = p.TB * (p.TH + 1) * (p.TW + 1);
    p.nchunks = d->Cin / CK;
    const double eff = (double)d->B * d->H * d->W / ((double)p.tiles_y * p.tiles_x * p.groups * 128);
    if (eff < 0.25 || p.groups > 65535) return;
    const size_t patch_bytes = ((size_t)p.patch_px * (CK * es + 16) + 255) & ~(size_t)255;
    const size_t stage_bytes = (size_t)256 * (p.BN * es + 16);
    p.lds_bytes = 2 * patch_bytes > stage_bytes ? 2 * patch_bytes : stage_bytes;
    p.ok = 1;
}

int patch_dgrad_s2_launch(const cvg_conv_desc* d, const S2tPlan& p, const void* dy, const void* w, void* dx,
                          const void* gate, hipStream_t s) {
    S2tArgs a;
    const int es = d->dtype == CVG_BF16 ? 2 : 4;
    a.x = dy; a.wf = w; a.y = dx; a.gate = gate;
    a.B = d->B; a.H = d->H; a.W = d->W;
    a.x_img_pitch = d->x_img_pitch; a.x_row_pitch = d->x_row_pitch; a.x_pix_pitch = d->x_pix_pitch;
    a.N = d->N; a.y_img_pitch = d->y_img_pitch; a.y_row_pitch = d->y_row_pitch; a.y_pix_pitch = d->y_pix_pitch;
    a.nchunks = p.nchunks; a.patch_px = p.patch_px;
    a.patch_bytes = (p.patch_px * (CK * es + 16) + 255) & ~255;
    a.lane_perm = 1;
    fill_tile(a.t, p.TB, p.TH, p.TW, 2, 2, p.tiles_x);
    const int threads = p.BN == 128 ? 512 : 256;
    const int pt = 256 * (p.BN * es / 16) / threads;                 // staged rows per thread in the store phase
    a.row_fast = ((2 * p.TW) % pt == 0) ? 1 : 0;
    const dim3 grid(p.tiles_y * p.tiles_x, ceil_div(d->N, p.BN), p.groups);
    if (d->dtype == CVG_BF16 && p.BN == 128) {
        if (set_lds(patch_dgrad_s2<bf16, 2, 4>, p.lds_bytes)) return -1;
        patch_dgrad_s2<bf16, 2, 4><<<grid, 512, p.lds_bytes, s>>>(a);
    } else if (d->dtype == CVG_BF16) {
        if (set_lds(patch_dgrad_s2<bf16, 2, 2>, p.lds_bytes)) return -1;
        patch_dgrad_s2<bf16, 2, 2><<<grid, 256, p.lds_bytes, s>>>(a);
    } else {
        if (set_lds(patch_dgrad_s2<float, 2, 2>, p.lds_bytes)) return -1;
        patch_dgrad_s2<float, 2, 2><<<grid, 256, p.lds_bytes, s>>>(a);
    }
    CVG_LAUNCH_CHECK();
    return 0;
}
