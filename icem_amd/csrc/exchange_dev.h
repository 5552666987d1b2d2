// exchange_dev.h -- consumer side of the in-library elite exchange (exchange.hip): wait until every rank's push of
// the running exchange has landed in THIS rank's block, then make the records visible to the calling workgroup.
#pragma once
#include "icem_fused.h"

namespace icem {

// One wavefront calls this (all 64 lanes); the caller follows it with a workgroup barrier before other waves read the
// records.  Lane r < world polls flag r of the local block (relaxed, system scope: peers write it over xGMI) with
// s_sleep between polls; when all have REACHED `seq` (sequence numbers only grow: a rank that gave up on a timeout
// meets its peers again at the next exchange instead of missing them for ever), ONE system-scope acquire.  Bounded:
// after w.max_polls polls (default XCHG_MAX_POLLS, a few seconds) the status word -- host-visible memory, read by
// icem_plan_step_sharded before the next MPC step -- is set and the wait gives up: a lost peer must not hang the GPU.
constexpr unsigned XCHG_MAX_POLLS = 6u << 20;

__device__ __forceinline__ void xchg_wait(const XchgWait& w, int lane) {
    if (w.flags == nullptr) return;
    const unsigned* f = w.flags + (lane < w.world ? lane : 0);
    unsigned polls = 0;
    bool ok = false;
    while (true) {
        const unsigned v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        ok = __all(lane >= w.world || (int)(v - w.seq) >= 0);
        if (ok || ++polls > w.max_polls) break;
        __builtin_amdgcn_s_sleep(8);
    }
    if (!ok && lane == 0) __hip_atomic_store(w.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

// the same wait for monotonically growing flags (>= seq): the probe's rounds may overtake each other by one
__device__ __forceinline__ void xchg_wait_at_least(const XchgWait& w, int lane) {
    const unsigned* f = w.flags + (lane < w.world ? lane : 0);
    unsigned polls = 0;
    bool ok = false;
    while (true) {
        const unsigned v = __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        ok = __all(lane >= w.world || (int)(v - w.seq) >= 0);
        if (ok || ++polls > w.max_polls) break;
        __builtin_amdgcn_s_sleep(2);
    }
    if (!ok && lane == 0) __hip_atomic_store(w.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}

}  // namespace icem
