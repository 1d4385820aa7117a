// Library-level entry points: ABI version, thread-local error text, device count.
#include "common.h"
#include "../../include/mi355_retrieval.h"

#include <stdarg.h>

namespace mi355 {
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
}  // namespace mi355

extern "C" {
int mi355_abi_version(void) { return MI355_ABI_VERSION; }
const char* mi355_last_error(void) { return mi355::g_err; }
int mi355_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}
}
