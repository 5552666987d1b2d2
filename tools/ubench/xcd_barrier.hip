// What would an iteration boundary cost INSIDE one XCD?  ("the five iterations inside ONE XCD" -- 256
// tiles on the 32 CUs of one XCD, the iteration boundary a 32-arrival barrier in the XCD's own L2 instead of a kernel
// boundary.)  1024 workgroups are launched; each reads HW_REG_XCC_ID, the first 32 that find themselves on XCD 0 stay (a
// claim counter), the rest exit.  The 32 then run R rounds of
//   publish : every member stores an 80-byte "candidate list" (10 packed keys) with PLAIN stores (L1 is write-through: the
//             line sits in the XCD's L2), drains them (s_waitcnt vmcnt(0)),
//   arrive  : one lane adds to a counter -- (a) WORKGROUP-scope RMW (no sc bits: performed in the XCD's own L2), or
//             (b) AGENT-scope RMW (sc1: performed memory-side, what members on different XCDs would need),
//   wait    : one lane polls the counter with sc1 loads (bypass L1, served by L2) + s_sleep, then __syncthreads,
//   read    : every member reads all 32 lists (2560 B) with sc1 loads and folds them (the next prologue's selection input).
// Reported: µs per round for (a) and (b), with and without the read, and how many of the 32 members really sat on XCD 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int MEMBERS = 32, KEYS = 10;

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xF;
}
__device__ __forceinline__ unsigned long long load_sc1_u64(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // global_load_dwordx2 ... sc1
}

template <bool AGENT, bool READ>
__global__ __launch_bounds__(256) void xcd_rounds(unsigned* claim, unsigned* counter, unsigned long long* lists, int rounds,
                                                  long long* out, unsigned* where, unsigned long long* sink) {
    __shared__ int member_s;
    if (threadIdx.x == 0) {
        int me = -1;
        if (xcc_id() == 0) {
            const unsigned t = atomicAdd(claim, 1u);
            if (t < MEMBERS) me = (int)t;
        }
        member_s = me;
    }
    __syncthreads();
    const int me = member_s;
    if (me < 0) return;
    if (threadIdx.x == 0) where[me] = xcc_id();
    unsigned long long acc = 0;
    long long t0 = 0;
    for (int r = 0; r <= rounds; ++r) {
        if (r == 1 && threadIdx.x == 0) t0 = wall_clock64();   // (round 0: everybody has arrived once -- the members start together)
        // publish
        if (threadIdx.x < KEYS) lists[(size_t)(r & 1) * MEMBERS * KEYS + me * KEYS + threadIdx.x] = ((unsigned long long)r << 32) | (unsigned)(me * 16 + threadIdx.x);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            if (AGENT) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const unsigned target = (unsigned)(r + 1) * MEMBERS;
            unsigned polls = 0;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target && ++polls < (1u << 22)) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        if (READ) {
            // all 32 lists of this round: 320 keys over 256 threads (sc1: L1 bypassed, the XCD's L2 serves them)
            for (int e = threadIdx.x; e < MEMBERS * KEYS; e += 256) {
                const unsigned long long k = load_sc1_u64(lists + (size_t)(r & 1) * MEMBERS * KEYS + e);
                acc += k;
                if ((k >> 32) != (unsigned long long)r) acc |= 1ull << 63;   // a stale key: flagged
            }
        }
    }
    if (threadIdx.x == 0 && me == 0) out[0] = wall_clock64() - t0;
    if (READ) sink[me * 256 + threadIdx.x] = acc;
}

template <bool AGENT, bool READ>
static void run(const char* what, unsigned* claim, unsigned* counter, unsigned long long* lists, long long* out, unsigned* where,
                unsigned long long* sink) {
    const int rounds = 400;
    double best = 1e9;
    int on0 = 0, stale = 0;
    for (int rep = 0; rep < 5; ++rep) {
        (void)hipMemset(claim, 0, 4);
        (void)hipMemset(counter, 0, 4);
        (void)hipMemset(where, 0xFF, MEMBERS * 4);
        (void)hipMemset(sink, 0, MEMBERS * 256 * 8);
        hipLaunchKernelGGL((xcd_rounds<AGENT, READ>), dim3(1024), dim3(256), 0, 0, claim, counter, lists, rounds, out, where, sink);
        (void)hipDeviceSynchronize();
        long long t;
        (void)hipMemcpy(&t, out, 8, hipMemcpyDeviceToHost);
        std::vector<unsigned> w(MEMBERS);
        (void)hipMemcpy(w.data(), where, MEMBERS * 4, hipMemcpyDeviceToHost);
        on0 = 0;
        for (unsigned x : w) on0 += x == 0 ? 1 : 0;
        std::vector<unsigned long long> s(MEMBERS * 256);
        (void)hipMemcpy(s.data(), sink, s.size() * 8, hipMemcpyDeviceToHost);
        for (auto v : s) stale += (v >> 63) ? 1 : 0;
        best = std::min(best, t / 100.0 / rounds);
    }
    printf("%-58s %.2f us per round (best of 5; %d of %d members on XCD 0; %d threads saw a stale key)\n", what, best, on0, MEMBERS, stale);
}

int main() {
    unsigned *claim, *counter, *where;
    unsigned long long *lists, *sink;
    long long* out;
    (void)hipMalloc(&claim, 4); (void)hipMalloc(&counter, 4); (void)hipMalloc(&where, MEMBERS * 4);
    (void)hipMalloc(&lists, 2 * MEMBERS * KEYS * 8); (void)hipMalloc(&sink, MEMBERS * 256 * 8); (void)hipMalloc(&out, 8);
    run<false, false>("32-arrival barrier, counter in the XCD's L2 (workgroup-scope RMW):", claim, counter, lists, out, where, sink);
    run<true, false>("32-arrival barrier, counter memory-side (agent-scope RMW):", claim, counter, lists, out, where, sink);
    run<false, true>("... + every member reads the 32 lists (sc1 loads), L2 counter:", claim, counter, lists, out, where, sink);
    run<true, true>("... + every member reads the 32 lists (sc1 loads), agent counter:", claim, counter, lists, out, where, sink);
    return 0;
}
