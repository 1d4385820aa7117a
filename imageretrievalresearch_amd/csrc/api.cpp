// Library-level entry points: ABI version, thread-local error text, device count.
#include "common.h"
#include "../../include/mi355_retrieval.h"

#include <dlfcn.h>
#include <stdarg.h>
#include <stdlib.h>

namespace mi355 {
static thread_local char g_err[1024] = "";
void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- roctx ranges (tracing).  The marker library is looked up at run time - nothing links against it,
// and without it (or with ranges off) the calls are no-ops.  On: MI355_ROCTX=1 in the environment, or
// mi355_model_set_option(m, "roctx", 1).  `rocprofv3 --marker-trace` then shows one range per executor op
// ("embed/pw 96->576 @14x14", ...) and per rank phase.
static int (*g_roctx_push)(const char*) = nullptr;
static int (*g_roctx_pop)() = nullptr;
static int g_roctx_state = -1;     // -1 not looked up, 0 unavailable, 1 ready
static bool g_roctx_on = false;
static void roctx_lookup() {
    if (g_roctx_state >= 0) return;
    g_roctx_state = 0;
    for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
        void* h = dlopen(name, RTLD_LAZY | RTLD_LOCAL);
        if (!h) continue;
        g_roctx_push = (int (*)(const char*))dlsym(h, "roctxRangePushA");
        g_roctx_pop = (int (*)())dlsym(h, "roctxRangePop");
        if (g_roctx_push && g_roctx_pop) { g_roctx_state = 1; break; }
    }
    const char* e = getenv("MI355_ROCTX");
    if (e && e[0] && e[0] != '0') g_roctx_on = true;
}
void roctx_enable(bool on) { roctx_lookup(); g_roctx_on = on; }
bool roctx_active() { roctx_lookup(); return g_roctx_on && g_roctx_state == 1; }
void roctx_push(const char* label) { if (roctx_active()) g_roctx_push(label); }
void roctx_pop() { if (roctx_active()) g_roctx_pop(); }
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
