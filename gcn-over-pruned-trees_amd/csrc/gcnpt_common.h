// Shared host/device helpers for libgcnpt.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <mutex>
#include <stdint.h>
#include <stdio.h>

#include "../../include/gcnpt.h"

namespace gcnpt {

// ---- error reporting across the C boundary (no exceptions leave the library) -------------------
char* err_buf();   // thread-local, defined in capi.hip
#ifdef GCNPT_STAMPS
extern int g_debug_knob;      // diagnostic builds: timing-experiment switches (results become wrong)
extern void* g_debug_stamps;   // diagnostic builds (-DGCNPT_STAMPS): device buffer for in-kernel time stamps
#else
// the shipped library keeps no mutable global state but the option table of gcnpt_set_option: the diagnostic hooks are compile-time constants here
constexpr int g_debug_knob = 0;
constexpr void* g_debug_stamps = nullptr;
#endif

#ifdef GCNPT_STAMPS
// one stamp = shader-clock counter of wave 0 / lane 0 of the workgroup; slot 15 = 100 MHz real time at entry
#define GCNPT_STAMP(buf, slot)                                                                           \
    do {                                                                                                 \
        if ((buf) && threadIdx.x == 0) {                                                                 \
            unsigned long long _t;                                                                       \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");                  \
            (buf)[(size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 16 + (slot)] = _t; \
        }                                                                                                \
    } while (0)
#define GCNPT_STAMP_REAL(buf)                                                                            \
    do {                                                                                                 \
        if ((buf) && threadIdx.x == 0) {                                                                 \
            unsigned long long _t;                                                                       \
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(_t)::"memory");              \
            (buf)[(size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 16 + 15] = _t; \
        }                                                                                                \
    } while (0)
#else
#define GCNPT_STAMP(buf, slot) do {} while (0)
#define GCNPT_STAMP_REAL(buf) do {} while (0)
#endif
int fail(int code, const char* fmt, ...);
void note_launch(int grid, int block, size_t lds, size_t kernarg);      // thread-local record for gcnpt_last_launch (capi.hip)
int option(int key);      // GCNPT_OPT_* of include/gcnpt.h (capi.hip): one relaxed atomic load, no environment access

#define GCNPT_HIP_CHECK(expr)                                                                  \
    do {                                                                                       \
        hipError_t _e = (expr);                                                                \
        if (_e != hipSuccess)                                                                  \
            return gcnpt::fail(GCNPT_E_HIP, "%s failed: %s", #expr, hipGetErrorString(_e));    \
    } while (0)

// hipFuncSetAttribute(max dynamic LDS) exactly once per kernel instantiation, whichever thread comes first (std::call_once);
// it is not a stream operation, so it must not recur inside a graph capture either (the first launch is an eager warm-up)
#define GCNPT_LDS_ATTR_ONCE(kern, bytes)                                                                              \
    do {                                                                                                              \
        static std::once_flag _once;                                                                                  \
        static hipError_t _rc = hipSuccess;                                                                           \
        std::call_once(_once, [&] { _rc = hipFuncSetAttribute((const void*)(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)(bytes)); }); \
        GCNPT_HIP_CHECK(_rc);                                                                                         \
    } while (0)

#define GCNPT_REQUIRE(cond, ...)                                      \
    do {                                                              \
        if (!(cond)) return gcnpt::fail(GCNPT_E_INVALID, __VA_ARGS__); \
    } while (0)

constexpr int WAVE = 64;
constexpr int FWD_BOUND = 42;      // utils/constant.py:14  DEPREL_FORWARD_BOUND
constexpr int SELF_LOOP_ID = 84;   // utils/constant.py:12,29  DEPREL_TO_ID['self_loop']

// LDS operations of ONE wave complete in issue order, so data handed from lane to lane of the same wave through LDS only
// needs the compiler kept from moving LDS accesses across the hand-over and the wave's own outstanding LDS operations
// waited for: no workgroup barrier
__device__ __forceinline__ void wave_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__host__ __device__ inline int round_up(int x, int m) { return (x + m - 1) / m * m; }
__host__ __device__ inline int ceil_div(int x, int m) { return (x + m - 1) / m; }

// ---- bf16 <-> f32 (storage type is a raw 16-bit pattern) -------------------------------------------
typedef unsigned short bf16_t;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((unsigned)v) << 16); }

// round-to-nearest-even; a plain cast lowers to v_cvt_pk_bf16_f32 on gfx950 and keeps NaNs NaN
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

}  // namespace gcnpt
