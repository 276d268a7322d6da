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
// the 16-bit storage type this build computes in: 0 = bf16 (libtcvom_hip.so), 1 = IEEE fp16 (libtcvom_hip_f16.so)
extern "C" int tcvom_act_dtype(void) {
#ifdef TCVOM_F16
    return 1;
#else
    return 0;
#endif
}
