// Error state and version of the C-ABI (include/gcnpt.h).  The entry points themselves live next to
// their kernels in tree_kernels.hip and layer_kernels.hip.
#include <stdarg.h>

#include "gcnpt_common.h"

namespace gcnpt {

char* err_buf() {
    static thread_local char buf[512] = {0};
    return buf;
}

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(err_buf(), 512, fmt, ap);
    va_end(ap);
    return code;
}

#ifdef GCNPT_STAMPS
void* g_debug_stamps = nullptr;
int g_debug_knob = 0;
#endif

}  // namespace gcnpt

#ifdef GCNPT_STAMPS
// libgcnpt_stamps.so only (make stamps; never shipped as libgcnpt.so): where the in-kernel s_memtime stamps go; not part of the ABI
extern "C" void gcnpt_debug_set_stamps(void* dev_ptr) { gcnpt::g_debug_stamps = dev_ptr; }
extern "C" void gcnpt_debug_set_knob(int k) { gcnpt::g_debug_knob = k; }   // timing experiments of the stamps build (wrong results)
#endif

extern "C" int gcnpt_abi_version(void) { return GCNPT_ABI_VERSION; }
extern "C" const char* gcnpt_last_error(void) { return gcnpt::err_buf(); }
