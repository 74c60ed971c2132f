// Error state and version of the C-ABI (include/gcnpt.h).  The entry points themselves live next to
// their kernels in tree_kernels.hip and layer_kernels.hip.
#include <stdarg.h>
#include <stdlib.h>

#include <atomic>

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

// The library's ONE piece of process state: the option table of include/gcnpt.h (gcnpt_set_option).  Relaxed atomics, read by value
// at the top of a call; the environment is consulted exactly once, when the library is loaded, for the defaults.
static std::atomic<int> g_options[GCNPT_OPT_COUNT];
static const bool g_options_init = [] {
    const int defaults[GCNPT_OPT_COUNT] = {0, -1, 192, -1};
    const char* names[GCNPT_OPT_COUNT] = {"GCNPT_DETERMINISTIC", "GCNPT_WAVES4", "GCNPT_SIDE_TILES", "GCNPT_COL_SPLIT"};
    for (int i = 0; i < GCNPT_OPT_COUNT; ++i) {
        const char* e = getenv(names[i]);
        g_options[i].store((e && e[0]) ? atoi(e) : defaults[i], std::memory_order_relaxed);
    }
    return true;
}();
int option(int key) { return g_options[key].load(std::memory_order_relaxed); }

// shape of the calling thread's most recent layer-path launch (gcnpt_last_launch: what bench.py's launch-floor leg mirrors)
static thread_local int t_last_launch[4] = {0, 0, 0, 0};
void note_launch(int grid, int block, size_t lds, size_t kernarg) {
    t_last_launch[0] = grid; t_last_launch[1] = block; t_last_launch[2] = (int)lds; t_last_launch[3] = (int)kernarg;
}

// launch-floor probe: a kernel with a given grid / workgroup size / LDS / kernel-argument size whose body returns at entry
template <int BYTES> struct KernargPad { unsigned char b[BYTES]; };
template <int BYTES> __global__ void empty_kernel(const KernargPad<BYTES>) {}
template <int BYTES> static int launch_empty(hipStream_t s, int grid, int block, int lds) {
    GCNPT_LDS_ATTR_ONCE(empty_kernel<BYTES>, 160 * 1024);
    KernargPad<BYTES> pad{};
    hipLaunchKernelGGL(empty_kernel<BYTES>, dim3(grid), dim3(block), (size_t)lds, s, pad);
    GCNPT_HIP_CHECK(hipGetLastError());
    return GCNPT_OK;
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

extern "C" int gcnpt_set_option(int option, int value) {
    if (option < 0 || option >= GCNPT_OPT_COUNT) return gcnpt::fail(GCNPT_E_INVALID, "set_option: unknown option %d", option);
    if (option == GCNPT_OPT_DETERMINISTIC && value != 0 && value != 1) return gcnpt::fail(GCNPT_E_INVALID, "set_option: deterministic is 0 or 1");
    if (option == GCNPT_OPT_FOUR_WAVES && (value < -1 || value > 1)) return gcnpt::fail(GCNPT_E_INVALID, "set_option: four_waves is -1 (auto), 0 or 1");
    if (option == GCNPT_OPT_SIDE_TILES && value < 0) return gcnpt::fail(GCNPT_E_INVALID, "set_option: side_tiles must be >= 0");
    if (option == GCNPT_OPT_COL_SPLIT && (value < -1 || value > 8)) return gcnpt::fail(GCNPT_E_INVALID, "set_option: col_split is -1 (auto), 0 or 1..8 workgroups per row tile");
    gcnpt::g_options[option].store(value, std::memory_order_relaxed);
    return GCNPT_OK;
}
extern "C" int gcnpt_get_option(int option) {
    if (option < 0 || option >= GCNPT_OPT_COUNT) return gcnpt::fail(GCNPT_E_INVALID, "get_option: unknown option %d", option);
    return gcnpt::option(option);
}

extern "C" int gcnpt_last_launch(int* grid, int* block, int* lds_bytes, int* kernarg_bytes) {
    if (!grid || !block || !lds_bytes || !kernarg_bytes) return gcnpt::fail(GCNPT_E_INVALID, "last_launch: null pointer");
    *grid = gcnpt::t_last_launch[0]; *block = gcnpt::t_last_launch[1]; *lds_bytes = gcnpt::t_last_launch[2]; *kernarg_bytes = gcnpt::t_last_launch[3];
    return GCNPT_OK;
}

extern "C" int gcnpt_launch_empty(void* stream, int grid, int block, int lds_bytes, int kernarg_bytes);
extern "C" int gcnpt_launch_empty_seq(void* stream, int n, const int* grid, const int* block, const int* lds_bytes, const int* kernarg_bytes) {
    if (n < 0 || (n > 0 && (!grid || !block || !lds_bytes || !kernarg_bytes))) return gcnpt::fail(GCNPT_E_INVALID, "launch_empty_seq: bad argument");
    for (int i = 0; i < n; ++i) {
        const int rc = gcnpt_launch_empty(stream, grid[i], block[i], lds_bytes[i], kernarg_bytes[i]);
        if (rc != GCNPT_OK) return rc;
    }
    return GCNPT_OK;
}

extern "C" int gcnpt_launch_empty(void* stream, int grid, int block, int lds_bytes, int kernarg_bytes) {
    if (grid <= 0 || block <= 0 || block > 1024 || lds_bytes < 0 || lds_bytes > 160 * 1024 || kernarg_bytes < 0)
        return gcnpt::fail(GCNPT_E_INVALID, "launch_empty: grid %d block %d lds %d kernarg %d", grid, block, lds_bytes, kernarg_bytes);
    hipStream_t s = (hipStream_t)stream;
    if (kernarg_bytes <= 64) return gcnpt::launch_empty<64>(s, grid, block, lds_bytes);
    if (kernarg_bytes <= 128) return gcnpt::launch_empty<128>(s, grid, block, lds_bytes);
    if (kernarg_bytes <= 256) return gcnpt::launch_empty<256>(s, grid, block, lds_bytes);
    if (kernarg_bytes <= 512) return gcnpt::launch_empty<512>(s, grid, block, lds_bytes);
    if (kernarg_bytes <= 1024) return gcnpt::launch_empty<1024>(s, grid, block, lds_bytes);
    return gcnpt::launch_empty<2048>(s, grid, block, lds_bytes);
}
