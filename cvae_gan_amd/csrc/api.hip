// Library-level entry points: version and last-error text.
#include "common.h"
#include <string.h>
#include <stdlib.h>

static thread_local char g_err[512] = "";

void cvg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// Benchmarking (A/B) switches: the environment is read once, here, when the first kernel is planned.
const CvgSwitches& cvg_switches() {
    static const CvgSwitches sw = [] {
        CvgSwitches w;
        memset(&w, 0, sizeof(w));
        auto on = [](const char* name) { return getenv(name) != nullptr; };
        auto ch = [](const char* name) { const char* e = getenv(name); return e ? e[0] : '\0'; };
        auto num = [](const char* name) { const char* e = getenv(name); return e ? atoi(e) : 0; };
        w.no_patch = on("CVG_NO_PATCH");
        w.no_xcd_remap = on("CVG_NO_XCD_REMAP");
        w.patch_2w = on("CVG_PATCH_2W");
        w.wgrad_s2_gather = on("CVG_WGRAD_S2_GATHER");
        w.wgrad_v1 = on("CVG_WGRAD_V1");
        w.wgrad_4w = on("CVG_WGRAD_4W");
        w.wgrad_wide = on("CVG_WGRAD_WIDE");
        w.no_lowk = on("CVG_NO_LOWK");
        w.no_walk64 = on("CVG_NO_WALK64");
        w.no_res_rows = on("CVG_NO_RES_ROWS");
        w.bn_raw_sums = on("CVG_BN_RAW_SUMS");
        w.split_finalize = on("CVG_SPLIT_FINALIZE");
        w.patch_wide = ch("CVG_PATCH_WIDE");
        w.patch_persist = ch("CVG_PATCH_PERSIST");
        w.patch_perm = ch("CVG_PATCH_PERM");
        w.lds_extra = num("CVG_PATCH_LDS_EXTRA");
        w.wgrad_wgs = num("CVG_WGRAD_WGS");
        const char* tile = getenv("CVG_PATCH_TILE");
        w.tile_set = tile && sscanf(tile, "%d,%d,%d", &w.tile[0], &w.tile[1], &w.tile[2]) == 3;
        return w;
    }();
    return sw;
}

extern "C" {
int cvg_version(void) { return 2; }
const char* cvg_last_error(void) { return g_err; }
}
