// Host-side exercise of libcvaegan_hip's validation and planning layer, built with AddressSanitizer and
// UndefinedBehaviorSanitizer on the HOST code (`make asan`).  No GPU is needed: every call either is a pure host
// query (kernel names, workspace sizes, pack jobs) or must be rejected by validation before any launch.
#include <stdio.h>
#include <string.h>
#include <vector>
#include "cvaegan.h"

static int failures = 0;
#define EXPECT(cond, ...) do { if (!(cond)) { ++failures; fprintf(stderr, "FAIL %s:%d: ", __FILE__, __LINE__); \
    fprintf(stderr, __VA_ARGS__); fprintf(stderr, "\n"); } } while (0)

static cvg_conv_desc conv3x3(int dtype, int B, int H, int cin, int cout, int stride) {
    cvg_conv_desc d;
    memset(&d, 0, sizeof(d));
    d.dtype = dtype; d.B = B; d.H = H; d.W = H; d.Cin = cin;
    d.x_pix_pitch = cin; d.x_row_pitch = H * cin; d.x_img_pitch = H * H * cin;
    d.OH = H / stride; d.OW = H / stride; d.N = cout; d.KW = 3; d.ntaps = 9;
    d.stride = stride; d.tstep_y = 1; d.tstep_x = 1; d.off_y = -1; d.off_x = -1; d.pad_mode = CVG_PAD_ZERO;
    d.y_pix_pitch = cout; d.y_row_pitch = d.OW * cout; d.y_img_pitch = d.OH * d.OW * cout;
    d.oy_mul = 1; d.ox_mul = 1;
    const int bk = dtype == CVG_BF16 ? 64 : 32;
    d.w_ld = (9 * cin + bk - 1) / bk * bk;
    d.w_rows = (cout + 127) / 128 * 128;
    return d;
}

int main() {
    EXPECT(cvg_version() >= 1, "version");
    // plan queries over the shapes the networks use (and some odd ones): names must be non-empty, sizes finite
    const int dtypes[2] = {CVG_F32, CVG_BF16};
    const int chans[6] = {8, 16, 32, 64, 128, 512};
    const int sizes[5] = {1, 2, 8, 20, 64};
    for (int dt : dtypes)
        for (int cin : chans)
            for (int cout : chans)
                for (int H : sizes)
                    for (int stride = 1; stride <= 2; ++stride) {
                        if (stride == 2 && H < 2) continue;
                        cvg_conv_desc d = conv3x3(dt, 3, H, cin, cout, stride);
                        const char* k = cvg_conv_fwd_kernel(&d);
                        EXPECT(k && k[0], "fwd kernel name");
                        k = cvg_conv_wgrad_kernel(&d);
                        EXPECT(k && k[0], "wgrad kernel name");
                        (void)cvg_conv_fwd_workspace(&d);
                        EXPECT(cvg_conv_wgrad_workspace(&d) < ((size_t)1 << 40), "wgrad workspace");
                        for (int co = 1; co <= 4; co *= 2) { d.co_launches = co; (void)cvg_conv_wgrad_workspace(&d); }
                        d.flags = CVG_CONV_ALONE;
                        (void)cvg_conv_fwd_kernel(&d);
                        // the transposed stride-2 form on the same sizes
                        if (stride == 2) {
                            cvg_conv_desc t = conv3x3(dt, 3, H / 2, cout, cin, 1);
                            t.form = CVG_FORM_S2_TRANSPOSED;
                            t.y_pix_pitch = cin; t.y_row_pitch = H * cin; t.y_img_pitch = H * H * cin;
                            k = cvg_conv_fwd_kernel(&t);
                            EXPECT(k && k[0], "transposed kernel name");
                        }
                        // pack jobs are filled on the host
                        cvg_wmap m;
                        memset(&m, 0, sizeof(m));
                        m.nd = 1; m.na = cin * 9; m.kd = cin; m.ka = 1; m.kb = 9; m.n_valid = cout; m.kc_valid = cin; m.ntaps = 9;
                        cvg_pack_job job;
                        float dummy_c = 0.f; char dummy_p = 0;
                        const int nb = cvg_weight_pack_job(&d, &m, &dummy_c, &dummy_p, 0, &job);
                        EXPECT(nb > 0, "pack job blocks %d (%s)", nb, cvg_last_error());
                    }
    // malformed descriptors must be rejected before any launch, with a message
    {
        cvg_conv_desc d = conv3x3(CVG_BF16, 1, 4, 12, 8, 1);            // Cin not a multiple of 8 / not a power of two
        EXPECT(cvg_conv_fwd(&d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr) != 0, "bad Cin accepted");
        EXPECT(strstr(cvg_last_error(), "Cin") != nullptr, "message: %s", cvg_last_error());
        d = conv3x3(CVG_BF16, 1, 8, 16, 16, 1);                         // valid geometry, null tensors
        EXPECT(cvg_conv_fwd(&d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr) != 0, "null tensors accepted");
        d.dtype = 7;
        EXPECT(cvg_conv_fwd(&d, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr) != 0, "bad dtype accepted");
        cvg_conv_desc t = conv3x3(CVG_BF16, 2, 4, 16, 8, 1);            // 16 dY channels: the transposed form needs 32
        t.form = CVG_FORM_S2_TRANSPOSED;
        EXPECT(strcmp(cvg_conv_fwd_kernel(&t), "unsupported") == 0, "transposed form with 16 channels");
        EXPECT(cvg_conv_fwd(&t, &t, &t, &t, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, nullptr) != 0, "unsupported transposed form accepted");
    }
    // row kernels: argument validation
    EXPECT(cvg_bn_stats(nullptr, CVG_BF16, 16, 12, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr) != 0, "bn_stats C = 12 accepted");
    EXPECT(cvg_bn_stats_blocks(1 << 20, 64) > 0 && cvg_bn_stats_blocks(7, 8) > 0, "bn_stats_blocks");
    EXPECT(cvg_colsum(nullptr, CVG_F32, 4, 8, 8, nullptr, 1, 1, 0, nullptr, nullptr) != 0, "colsum null accepted");
    EXPECT(cvg_upsample_taps_bwd(nullptr, nullptr, CVG_BF16, 1, 1, 1, 8, nullptr) != 0, "upsample_taps_bwd null accepted");
    EXPECT(cvg_adamw(nullptr, nullptr, nullptr, nullptr, 4, 1e-3, .5, .999, 1e-8, .01, 1, nullptr) != 0, "adamw null accepted");
    if (failures) { fprintf(stderr, "%d failure(s)\n", failures); return 1; }
    printf("abi_check: ok\n");
    return 0;
}
