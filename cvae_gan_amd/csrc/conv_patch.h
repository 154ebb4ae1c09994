// Patch-staged convolution kernels: the input tile plus its halo is staged ONCE per
// 32-channel chunk in LDS and every tap reads its MFMA operand from the shifted patch,
// instead of re-gathering the im2col rows from L2 for every tap.
#pragma once
#include "common.h"

struct PatchPlan {
    int ok;
    int BM, BN;                  // pixels x output channels per workgroup
    int TB, TH, TW;              // tile: images x rows x cols (TB*TH*TW <= BM)
    int PH, PW;                  // patch rows/cols per image = TH+KH-1, TW+KW-1
    int KH, KW, T;
    int tiles_y, tiles_x, groups;
    int patch_px, nchunks;
    int gstride;                 // grid depth: a workgroup takes image groups z, z + gstride, ... (== groups: one each)
    int stage_off;               // LDS offset of the output staging area (0: over the patch buffers)
    int stage_shared;            // the staging area lies over the second patch buffer (walking 64-channel form)
    int walk;                    // 1: the kernel form whose workgroups go on to further image groups
    size_t lds_bytes;
};

// Forward / input-gradient.  plan.ok == 0: not eligible (the gather kernel handles it).
void patch_fwd_plan(const cvg_conv_desc* d, PatchPlan& p);
// true: a launch without residual, gate or statistics epilogue runs on the resident seven-row-tap kernel (lowk_fwd)
bool lowk_fwd_takes(const cvg_conv_desc* d, const PatchPlan& p);
int patch_fwd_launch(const cvg_conv_desc* d, const PatchPlan& p, const void* x, const void* w, void* y,
                     const float* bias, const float* cbias, const int32_t* labels, const void* res,
                     const void* gate, hipStream_t s, const float* scale = nullptr, const float* shift = nullptr,
                     int act = 0, float* stat_ws = nullptr);
// stat_ws: f32 [2][tiles][N] with tiles = patch_stat_tiles(plan): per pixel tile, the column sums and sums of
// squares of the stored output (BatchNorm batch statistics without a pass over the output).
inline int patch_stat_tiles(const PatchPlan& p) { return p.tiles_y * p.tiles_x * p.groups; }
// Weights in MFMA-fragment order: [n/32][chunk][tap][k16 half][lane][8].
int patch_pack_launch(const cvg_conv_desc* d, const PatchPlan& p, const cvg_wmap* map, const float* canonical,
                      void* packed, hipStream_t s);

// Weight gradient.
struct PatchWgradPlan {
    int ok;
    int BNW, WN, WT;             // output channels per workgroup, waves over n, waves over taps
    int nine;                    // 1: the nine-tap double-buffered kernel (WT then counts chunks per workgroup)
    int stride;                  // 1, or 2 (nine-tap kernel only)
    int TB, TH, TW, PH, PW, KH, KW, T;
    int tiles_y, tiles_x, groups, ntiles, nsplit, tiles_per_split;
    int patch_px, nchunks, n_tiles;
    size_t lds_bytes, ws_bytes;
};
void patch_wgrad_plan(const cvg_conv_desc* d, PatchWgradPlan& p);
// dY[b][oy][ox][n] at dy + b*dy_img_pitch + oy*dy_row_pitch + ox*dy_ld + n, n < dy_cols readable.
int patch_wgrad_launch(const cvg_conv_desc* d, const PatchWgradPlan& p, const void* x, const void* dy, int dy_ld,
                       int dy_img_pitch, int dy_row_pitch, int dy_cols,
                       const cvg_wmap* map, float* grad, int accumulate, float* ws, hipStream_t s);

// Input gradient of a 3x3 stride-2 zero-padded convolution, all four output-parity classes in one launch
// (descriptors with form == CVG_FORM_S2_TRANSPOSED).  Weights in fragment order [n/32][chunk][tap 0..8][k16 half]
// [lane][8] with n the result's channel and k the dY channel.
struct S2tPlan {
    int ok;
    int BN;                      // result channels per workgroup (128 positions of the dY grid each)
    int TB, TH, TW, tiles_y, tiles_x, groups, patch_px, nchunks;
    size_t lds_bytes;
};
void patch_dgrad_s2_plan(const cvg_conv_desc* d, S2tPlan& p);
int patch_dgrad_s2_launch(const cvg_conv_desc* d, const S2tPlan& p, const void* dy, const void* w, void* dx,
                          const void* gate, hipStream_t s);
