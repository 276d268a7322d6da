// Error reporting + version for the C ABI.
#include "common.h"
#include <stdarg.h>

thread_local char g_tcvom_err[512] = "";

int tcvom_fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_tcvom_err, sizeof(g_tcvom_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char* tcvom_last_error(void) { return g_tcvom_err; }
extern "C" int tcvom_abi_version(void) { return 1; }
