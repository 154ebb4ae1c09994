// Library-level entry points: version and last-error text.
#include "common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void cvg_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {
int cvg_version(void) { return 1; }
const char* cvg_last_error(void) { return g_err; }
}
