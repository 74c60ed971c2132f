// Developer probe (not part of the product): what does a device-wide barrier between two phases of ONE kernel cost on
// MI355X, compared with a kernel boundary?  157 workgroups x 512 threads (the row-tile grid at B=50, T=100); every round each
// workgroup writes a 12.8 KB slab (a tile's output rows), makes it visible device-wide, meets the others at an atomic counter
// (BOUNDED spin: the probe cannot hang), then reads the slab of a workgroup that ran on another XCD and checks it.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/grid_barrier_probe tools/grid_barrier_probe.hip && /tmp/grid_barrier_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int WGS = 157, THREADS = 512, SLAB = 3200;      // 3200 floats = 12.8 KB per workgroup and round

__global__ __launch_bounds__(THREADS) void rounds_kernel(float* buf, unsigned* counter, int rounds, unsigned* bad, unsigned* timeouts) {
    const int wg = blockIdx.x, tid = threadIdx.x;
    for (int r = 0; r < rounds; ++r) {
        float* mine = buf + ((size_t)(r & 1) * WGS + wg) * SLAB;
        for (int i = tid; i < SLAB; i += THREADS) mine[i] = (float)(r * 1000 + wg);
        __threadfence();                                   // release: the slab leaves this XCD's L2
        __syncthreads();
        if (tid == 0) {
            atomicAdd(counter, 1u);
            const unsigned target = (unsigned)(r + 1) * WGS;
            int it = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && it < (1 << 22)) { __builtin_amdgcn_s_sleep(1); ++it; }
            if (it >= (1 << 22)) atomicAdd(timeouts, 1u);
        }
        __syncthreads();
        __threadfence();                                   // acquire side
        const int other = (wg + 3) % WGS;                  // blockIdx % 8 differs: another XCD
        const float* theirs = buf + ((size_t)(r & 1) * WGS + other) * SLAB;
        unsigned wrong = 0;
        for (int i = tid; i < SLAB; i += THREADS) wrong += __builtin_nontemporal_load(theirs + i) != (float)(r * 1000 + other);
        if (wrong) atomicAdd(bad, wrong);
    }
}

// variant: two-level barrier (one counter per blockIdx % 8 group = XCD, the last arriver of a group bumps the top counter),
// optionally without the fences (what the counters alone cost)
__global__ __launch_bounds__(THREADS) void rounds2_kernel(float* buf, unsigned* counters /*[9*32]*/, int rounds, unsigned* bad,
                                                          unsigned* timeouts, int fences) {
    const int wg = blockIdx.x, tid = threadIdx.x, grp = wg & 7;
    const unsigned members = (WGS - grp + 7) / 8;
    for (int r = 0; r < rounds; ++r) {
        float* mine = buf + ((size_t)(r & 1) * WGS + wg) * SLAB;
        for (int i = tid; i < SLAB; i += THREADS) mine[i] = (float)(r * 1000 + wg);
        if (fences) __threadfence();
        __syncthreads();
        if (tid == 0) {
            const unsigned before = atomicAdd(counters + grp * 32, 1u);
            if (before + 1 == (unsigned)(r + 1) * members) atomicAdd(counters + 8 * 32, 1u);
            const unsigned target = (unsigned)(r + 1) * 8;
            int it = 0;
            while (__hip_atomic_load(counters + 8 * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && it < (1 << 22)) { __builtin_amdgcn_s_sleep(1); ++it; }
            if (it >= (1 << 22)) atomicAdd(timeouts, 1u);
        }
        __syncthreads();
        if (fences) __threadfence();
        const int other = (wg + 3) % WGS;
        const float* theirs = buf + ((size_t)(r & 1) * WGS + other) * SLAB;
        unsigned wrong = 0;
        for (int i = tid; i < SLAB; i += THREADS) wrong += __builtin_nontemporal_load(theirs + i) != (float)(r * 1000 + other);
        if (wrong) atomicAdd(bad, wrong);
    }
}

// variant: the publish / consume recipe of cdna_hip_programming.md (ONE agent-scope release by lane 0 after every wave has
// drained its stores, a relaxed ticket, relaxed polling, ONE agent-scope acquire by lane 0) on the two-level counters
__global__ __launch_bounds__(THREADS) void rounds3_kernel(float* buf, unsigned* counters, int rounds, unsigned* bad, unsigned* timeouts) {
    const int wg = blockIdx.x, tid = threadIdx.x, grp = wg & 7;
    const unsigned members = (WGS - grp + 7) / 8;
    for (int r = 0; r < rounds; ++r) {
        float* mine = buf + ((size_t)(r & 1) * WGS + wg) * SLAB;
        for (int i = tid; i < SLAB; i += THREADS) mine[i] = (float)(r * 1000 + wg);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned before = __hip_atomic_fetch_add(counters + grp * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (before + 1 == (unsigned)(r + 1) * members) __hip_atomic_fetch_add(counters + 8 * 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const unsigned target = (unsigned)(r + 1) * 8;
            int it = 0;
            while (__hip_atomic_load(counters + 8 * 32, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && it < (1 << 22)) { __builtin_amdgcn_s_sleep(1); ++it; }
            if (it >= (1 << 22)) atomicAdd(timeouts, 1u);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        const int other = (wg + 3) % WGS;
        const float* theirs = buf + ((size_t)(r & 1) * WGS + other) * SLAB;
        unsigned wrong = 0;
        for (int i = tid; i < SLAB; i += THREADS) wrong += theirs[i] != (float)(r * 1000 + other);
        if (wrong) atomicAdd(bad, wrong);
    }
}

__global__ __launch_bounds__(THREADS) void one_phase_kernel(float* buf, int r, unsigned* bad) {      // the same work, one round per LAUNCH
    const int wg = blockIdx.x, tid = threadIdx.x;
    float* mine = buf + ((size_t)(r & 1) * WGS + wg) * SLAB;
    const int other = (wg + 3) % WGS;
    const float* theirs = buf + ((size_t)((r + 1) & 1) * WGS + other) * SLAB;                        // what the previous launch wrote
    unsigned wrong = 0;
    if (r > 0) for (int i = tid; i < SLAB; i += THREADS) wrong += theirs[i] != (float)((r - 1) * 1000 + other);
    for (int i = tid; i < SLAB; i += THREADS) mine[i] = (float)(r * 1000 + wg);
    if (wrong) atomicAdd(bad, wrong);
}

int main() {
    float* buf; unsigned *counter, *bad, *timeouts;
    CHECK(hipMalloc(&buf, sizeof(float) * 2 * WGS * SLAB));
    CHECK(hipMalloc(&counter, 4)); CHECK(hipMalloc(&bad, 4)); CHECK(hipMalloc(&timeouts, 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int rounds : {1, 2, 4, 8, 16, 64}) {
        float best = 1e9f; unsigned hb = 0, ht = 0;
        for (int rep = 0; rep < 20; ++rep) {
            CHECK(hipMemset(counter, 0, 4)); CHECK(hipMemset(bad, 0, 4)); CHECK(hipMemset(timeouts, 0, 4));
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(rounds_kernel, dim3(WGS), dim3(THREADS), 0, 0, buf, counter, rounds, bad, timeouts);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
            unsigned b, t; CHECK(hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&t, timeouts, 4, hipMemcpyDeviceToHost));
            hb += b; ht += t;
        }
        printf("one launch, %2d rounds with a device-wide barrier each: %7.2f us (%.2f us per round), wrong values %u, timeouts %u\n",
               rounds, best * 1e3, best * 1e3 / rounds, hb, ht);
    }
    unsigned* counters; CHECK(hipMalloc(&counters, 4 * 9 * 32));
    for (int fences : {1, 0})
        for (int rounds : {16, 64}) {
            float best = 1e9f; unsigned hb = 0, ht = 0;
            for (int rep = 0; rep < 20; ++rep) {
                CHECK(hipMemset(counters, 0, 4 * 9 * 32)); CHECK(hipMemset(bad, 0, 4)); CHECK(hipMemset(timeouts, 0, 4));
                CHECK(hipDeviceSynchronize());
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(rounds2_kernel, dim3(WGS), dim3(THREADS), 0, 0, buf, counters, rounds, bad, timeouts, fences);
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
                unsigned b, t; CHECK(hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&t, timeouts, 4, hipMemcpyDeviceToHost));
                hb += b; ht += t;
            }
            printf("one launch, %2d rounds, two-level barrier, fences %d:       %7.2f us (%.2f us per round), wrong values %u, timeouts %u\n",
                   rounds, fences, best * 1e3, best * 1e3 / rounds, hb, ht);
        }
    for (int rounds : {1, 2, 16, 64}) {
        float best = 1e9f; unsigned hb = 0, ht = 0;
        for (int rep = 0; rep < 20; ++rep) {
            CHECK(hipMemset(counters, 0, 4 * 9 * 32)); CHECK(hipMemset(bad, 0, 4)); CHECK(hipMemset(timeouts, 0, 4));
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(rounds3_kernel, dim3(WGS), dim3(THREADS), 0, 0, buf, counters, rounds, bad, timeouts);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
            unsigned b, t; CHECK(hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&t, timeouts, 4, hipMemcpyDeviceToHost));
            hb += b; ht += t;
        }
        printf("one launch, %2d rounds, lane-0 release/acquire recipe:       %7.2f us (%.2f us per round), wrong values %u, timeouts %u\n",
               rounds, best * 1e3, best * 1e3 / rounds, hb, ht);
    }
    for (int rounds : {16, 64}) {
        float best = 1e9f; unsigned hb = 0;
        for (int rep = 0; rep < 20; ++rep) {
            CHECK(hipMemset(bad, 0, 4)); CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < rounds; ++r) hipLaunchKernelGGL(one_phase_kernel, dim3(WGS), dim3(THREADS), 0, 0, buf, r, bad);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
            unsigned b; CHECK(hipMemcpy(&b, bad, 4, hipMemcpyDeviceToHost)); hb += b;
        }
        printf("%2d launches of the same phase back to back:            %7.2f us (%.2f us per launch), wrong values %u\n", rounds, best * 1e3, best * 1e3 / rounds, hb);
    }
    return 0;
}
