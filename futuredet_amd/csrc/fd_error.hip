// Error reporting for the C ABI: thread-local last-error text, never exit().
#include <stdarg.h>

#include "fd_common.h"

namespace fd {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace fd

extern "C" const char *fd_last_error(void) { return fd::g_err; }
extern "C" int fd_abi_version(void) { return 1; }
