// Error reporting for the C ABI: thread-local last-error string.
#include "bpb_common.h"

static thread_local char g_bpb_err[512] = "";

extern "C" const char* bpb_last_error(void) { return g_bpb_err; }

int bpb_set_error(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_bpb_err, sizeof(g_bpb_err), fmt, ap);
    va_end(ap);
    return code;
}
