// Grid-wide barrier for persistent kernels whose workgroups are all co-resident (one per CU).
//
// XCD-hierarchical: a workgroup arrives on the counter of its XCD (`HW_REG_XCC_ID`), the last
// arriver of an XCD arrives on the global counter, the last XCD releases eight per-XCD flags; every
// workgroup spins on the flag of its own XCD.  All words are agent-scope relaxed atomics (sc1), the
// hand-off recipe of the CDNA4 guide (Guideline 16, R1): the data handed over must itself have been
// written with write-through stores, and every storing wave drains before the workgroup arrives.
// Spins are bounded by the wall clock: after 1 s an error word is raised and every workgroup leaves
// (a workgroup that is not resident can then never hang the device).
//
// Memory: GL_BAR_WORDS unsigned words, zeroed by the host before every launch.
//   [0] arrivals at the start-up rendezvous   [32] error word   [64 + 32 x] workgroups on XCD x
//   [320 + 32 x] arrivals of XCD x            [576] XCDs arrived  [608 + 32 x] release flag of XCD x
#pragma once

#include <hip/hip_runtime.h>

constexpr int GL_BAR_WORDS = 1024;

struct GlGridBar {
    unsigned *bar;    // GL_BAR_WORDS words
    int rendezvous;   // workgroups expected at the start-up rendezvous (= grid size)
};

struct GlBarState {
    unsigned xcc;    // my XCD (group label only: any value is correct, placement changes speed only)
    unsigned n_my;   // workgroups in my group
    unsigned nx;     // groups in use
    unsigned epoch;  // barriers passed
};

__device__ __forceinline__ unsigned gl_ld_u32(const unsigned *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// spin until *p >= target; false on error word / timeout
__device__ inline bool gl_spin_until(const GlGridBar &G, const unsigned *p, unsigned target) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
    while (gl_ld_u32(p) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (gl_ld_u32(G.bar + 32) != 0u) return false;
        if (__builtin_amdgcn_s_memrealtime() - t0 > 100000000ull) {  // 1 s: a workgroup is not resident
            __hip_atomic_store(G.bar + 32, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
    }
    return true;
}

// called by thread 0 only, once at kernel start, BEFORE anything is modified: rendezvous of all
// workgroups (false: they are not co-resident) and the sizes of the XCD groups
__device__ inline bool gl_barrier_setup(const GlGridBar &G, GlBarState &B) {
    B.xcc = __builtin_amdgcn_s_getreg(63508) & 7u;  // hwreg(HW_REG_XCC_ID, 0, 32)
    B.epoch = 0;
    __hip_atomic_fetch_add(G.bar + 64 + 32 * B.xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_fetch_add(G.bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!gl_spin_until(G, G.bar, (unsigned)G.rendezvous)) return false;
    B.n_my = gl_ld_u32(G.bar + 64 + 32 * B.xcc);
    B.nx = 0;
    for (int x = 0; x < 8; ++x) B.nx += gl_ld_u32(G.bar + 64 + 32 * x) != 0u;
    return true;
}

// all threads of the workgroup call it; s_ok is a __shared__ int of the caller
__device__ inline bool gl_grid_barrier(const GlGridBar &G, GlBarState &B, int *s_ok) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave: its write-through stores have left
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned e = ++B.epoch;
        // level 1: the counter of my XCD.  Agent scope, as the HSA memory model requires between workgroups
        // (a workgroup-scope add happens to work -- every participant shares the XCD's L2, which executes it --
        // and measured 0.3 us faster, but nothing guarantees it across compiler versions)
        const unsigned a = __hip_atomic_fetch_add(G.bar + 320 + 32 * B.xcc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if (a == B.n_my * e) {  // last of my XCD
            const unsigned t = __hip_atomic_fetch_add(G.bar + 576, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
            if (t == B.nx * e)  // last XCD: release everybody
                for (int x = 0; x < 8; ++x)
                    __hip_atomic_store(G.bar + 608 + 32 * x, e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        *s_ok = gl_spin_until(G, G.bar + 608 + 32 * B.xcc, e) ? 1 : 0;
    }
    __syncthreads();
    return *s_ok != 0;
}
