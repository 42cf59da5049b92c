// Library-level entry points of libmofo_hip.so: version and the thread-local error string.
#include <stdarg.h>
#include <stdio.h>

#include "../../include/mofo_hip.h"

static thread_local char g_err[512] = "";

void mofo_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int mofo_version(void) { return MOFO_ABI_VERSION; }
extern "C" const char* mofo_last_error(void) { return g_err; }
