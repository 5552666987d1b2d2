// exchange.hip -- the in-library elite exchange between the GPUs of one node (world > 1): the all-gather of every
// rank's K sorted candidate records before the replicated refit, without a host call between an iteration's local
// launch and its merge.  Reference analogue: the pipe gather of icem/models/gt_par_model.py:77-94.
//
// Every rank owns ONE exchange block in its own HBM:
//     flags   [2][XCHG_MAX_WORLD] u32   sequence number of the last push of rank r, per parity
//     (status u32 -- superseded: the status word now lives in pinned, device-mapped HOST memory so that the host
//      sees a timed-out wait without a copy or a synchronisation; bit 0: a wait timed out, bit 1: the probe's payload
//      check failed)
//     records [2][world][K][2 + h*d]     the gathered records, per parity -- the layout icem_plan_iter_merge reads
// and maps the blocks of all other ranks (HIP IPC between processes, plain pointers inside one process).  A push is
// ONE launch on the pushing rank's stream (exchange_push_kernel, `world` workgroups): workgroup r copies the rank's K
// records into slot [parity][rank] of rank r's block with peer-to-peer stores (over xGMI between GPUs; 7.3 KB per
// peer at d = 6), fences at system scope and then stores the sequence number into that block's flag.  The consumer is
// the merge itself: the selection wavefront of the next launch's merge prologue (or merge_single / merge_refit) polls
// the `world` flags of its OWN block -- local memory -- and takes a system-scope acquire once they all carry the
// sequence number, then reads the records from the local block.  Consecutive exchanges alternate the parity: a rank
// can only be one exchange ahead of the slowest peer (its next push needs that peer's previous one), so two slots
// suffice.  No rank ever waits on the host, and the xGMI traffic is one small posted write burst per peer.
//
// The blocks are fine-grained device memory where the runtime grants it (remote stores must not be shadowed by the
// owner's L2); the flag polls and the acquire are system scope either way.  Every wait is bounded: on a timeout the
// block's status word is set and the kernel goes on (garbage in, but no hung GPU); icem_exchange_status reports it.
#include "host_common.h"
#include "exchange_dev.h"

namespace icem {

// flag words of a block: [0, 32) exchange flags (2 parities x 16 ranks), 32 status, [48, 64) probe flags
constexpr int XCHG_PROBE_FLAG0 = 48;

struct Exchange {
    unsigned char* block = nullptr;              // this rank's block (device)
    size_t bytes = 0;
    size_t rec_off = 0, rec_slot = 0;            // byte offset of records[0], bytes of one parity slot
    bool finegrained = false;
    std::vector<unsigned char*> peers;           // [world] device pointers of all blocks (own included)
    std::vector<bool> opened;                    // mapped through hipIpcOpenMemHandle
    unsigned char** peers_dev = nullptr;         // the same, on the device
    unsigned* status_host = nullptr;             // pinned + mapped: the kernels' timeout / payload-check reports
    unsigned* status_dev = nullptr;              // ... its device address
    unsigned seq = 0;                            // pushes so far (= sequence number of the last one)
    unsigned probe_seq = 0;
    long long* ticks_dev = nullptr;
    bool loopback = false;                       // measurement: every "peer" is this rank's own block, waits look at rank 0 only
    unsigned max_polls = XCHG_MAX_POLLS;         // ICEM_XCHG_MAX_POLLS overrides (tests of the timeout path)
    bool connected = false;
    // a peer connected by pointer lives in this process, possibly behind the same stream: its launches are ordered with
    // ours, so nothing of ours may wait for something it has not launched yet (no riding pack)
    bool local_peers = false;
    // fault injection (ICEM_XCHG_FAIL=connect|selftest|timeout[:rank[:pushes]], for first-contact drills on one GPU: bench.py
    // must end every such run on a fallback path and say which -- tests/test_gpu_exchange_faults.py): 1 = this rank's
    // icem_exchange_connect fails, 2 = its self-test reports a corrupted payload, 3 = after `fail_after` pushes it stops
    // publishing its records: every rank's bounded wait for them runs out and the next MPC step reports it
    int fail_mode = 0;
    unsigned fail_after = 60;
};

static void parse_fault(Exchange* x, int rank, int world) {
    x->fail_mode = 0;
#ifndef ICEM_FAULT_INJECTION
    // the product library carries no fault injection: libicem_hip_faults.so (icem_amd/build.py: this unit compiled with
    // -DICEM_FAULT_INJECTION, everything else the same objects) is what tests/test_gpu_exchange_faults.py loads
    (void)rank;
    (void)world;
    return;
#else
    const char* e = getenv("ICEM_XCHG_FAIL");
    if (!e || !*e) return;
    std::string s(e);
    std::string mode = s.substr(0, s.find(':'));
    int who = world - 1;
    unsigned after = 60;
    const size_t c1 = s.find(':');
    if (c1 != std::string::npos) {
        who = atoi(s.c_str() + c1 + 1);
        const size_t c2 = s.find(':', c1 + 1);
        if (c2 != std::string::npos) after = (unsigned)std::max(0, atoi(s.c_str() + c2 + 1));
    }
    if (who != rank) return;
    x->fail_mode = mode == "connect" ? 1 : mode == "selftest" ? 2 : mode == "timeout" ? 3 : 0;
    x->fail_after = after;
#endif
}

namespace {

// copy `words` 32-bit words of this rank's records into peer block `blk` (slot at rec_byte_off), then publish `seq` in
// the flag word `flag_idx` of that block: payload stores -> system-scope fence -> drained -> release store of the flag
__device__ __forceinline__ void push_to_block(const uint32_t* __restrict__ mine, int words, unsigned char* blk, size_t rec_byte_off,
                                              int flag_idx, unsigned seq) {
    uint32_t* dst = reinterpret_cast<uint32_t*>(blk + rec_byte_off);
    if ((words & 3) == 0) {
        const uint4* s4 = reinterpret_cast<const uint4*>(mine);
        uint4* d4 = reinterpret_cast<uint4*>(dst);
        for (int e = threadIdx.x; e < words / 4; e += blockDim.x) d4[e] = s4[e];
    } else {
        for (int e = threadIdx.x; e < words; e += blockDim.x) dst[e] = mine[e];
    }
    // (every thread waits for its stores, the workgroup meets, ONE thread releases at system scope: pack_records_body)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
        __hip_atomic_store(reinterpret_cast<unsigned*>(blk) + flag_idx, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256) void exchange_push_kernel(const uint32_t* __restrict__ mine, int words, unsigned char* const* peers,
                                                            size_t rec_byte_off, int rank, int parity, unsigned seq) {
    push_to_block(mine, words, peers[blockIdx.x], rec_byte_off, parity * XCHG_MAX_WORLD + rank, seq);
}

// the word the probe of rank `rank` writes at position e of its slot in round `seq`
__device__ __forceinline__ uint32_t probe_word(unsigned seq, int rank, int e) {
    return (seq * 0x9E3779B9u) ^ ((uint32_t)rank << 24) ^ (uint32_t)e;
}

// Self-test + measurement: `rounds` back-to-back exchanges (every rank pushes K records' worth of a round-stamped
// pattern to every block, waits for all ranks' flags, then CHECKS every rank's payload in its own block) inside ONE
// launch per rank, timed with the 100 MHz wall clock: the latency of one exchange as the planning loop sees it,
// without kernel launches around it.  The payload check is what catches a block whose peer-written records are
// shadowed by the owner's L2 (coarse-grained memory): the slots are rewritten every round with a different pattern,
// so a stale line shows.  Uses its own flag words (probe area); the parity slots are free between MPC steps.
__global__ __launch_bounds__(256) void exchange_probe_kernel(int words, unsigned char* const* peers, size_t rec_off, size_t rec_slot,
                                                             size_t rec_bytes, int rank, int world, int n_wait, unsigned base,
                                                             int rounds, unsigned max_polls, unsigned* status, long long* ticks_out) {
    unsigned char* own = peers[rank];
    __shared__ unsigned bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    const long long t0 = wall_clock64();
    for (int r = 1; r <= rounds; ++r) {
        const unsigned seq = base + (unsigned)r;
        const int parity = (int)(seq & 1u);
        for (int p = 0; p < world; ++p) {
            uint32_t* dst = reinterpret_cast<uint32_t*>(peers[p] + rec_off + (size_t)parity * rec_slot + (size_t)rank * rec_bytes);
            if ((words & 3) == 0) {
                for (int e = threadIdx.x; e < words / 4; e += blockDim.x)
                    reinterpret_cast<uint4*>(dst)[e] = uint4{probe_word(seq, rank, 4 * e), probe_word(seq, rank, 4 * e + 1),
                                                             probe_word(seq, rank, 4 * e + 2), probe_word(seq, rank, 4 * e + 3)};
            } else {
                for (int e = threadIdx.x; e < words; e += blockDim.x) dst[e] = probe_word(seq, rank, e);
            }
            __threadfence_system();
            __syncthreads();
            if (threadIdx.x == 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __hip_atomic_store(reinterpret_cast<unsigned*>(peers[p]) + XCHG_PROBE_FLAG0 + rank, seq, __ATOMIC_RELEASE,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
            }
            __syncthreads();
        }
        if (threadIdx.x < 64) {
            XchgWait w;
            w.flags = reinterpret_cast<const unsigned*>(own) + XCHG_PROBE_FLAG0;
            w.status = status;
            w.seq = seq;
            w.world = n_wait;
            w.max_polls = max_polls;
            xchg_wait_at_least(w, threadIdx.x);
        }
        __syncthreads();
        // the payload every waited-for rank pushed into THIS rank's block in this round (plain loads, as the merges read it)
        for (int q = 0; q < n_wait; ++q) {
            const int src_rank = n_wait == world ? q : rank;   // loopback measurement: only this rank's own slot is live
            const uint32_t* got = reinterpret_cast<const uint32_t*>(own + rec_off + (size_t)parity * rec_slot + (size_t)src_rank * rec_bytes);
            for (int e = threadIdx.x; e < words; e += blockDim.x)
                if (got[e] != probe_word(seq, src_rank, e)) bad = 1;
        }
        __syncthreads();   // (also: nobody overwrites a slot before every thread has checked it ... on THIS rank; a peer
                           //  two rounds ahead cannot exist: its next push to this parity needs this rank's flag of round seq+1)
    }
    if (threadIdx.x == 0) {
        ticks_out[0] = wall_clock64() - t0;
        if (bad) __hip_atomic_fetch_or(status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace


// this rank's records of the running iteration -> every rank's block; returns the arguments the merge waits with
int xchg_push(icem_handle* h, const void* my_records, hipStream_t st, XchgWait* wait_out) {
    Exchange* x = h->xchg;
    if (!x || !x->connected) return fail(ICEM_E_STATE, "icem_exchange_connect has not been called");
    const int world = h->cfg.world, K = h->cfg.num_elites;
    const size_t rec_bytes = (size_t)K * (h->hd + 2) * h->tsize;
    const unsigned seq = ++x->seq;
    const int parity = (int)(seq & 1u);
    const size_t off = x->rec_off + (size_t)parity * x->rec_slot + (size_t)h->cfg.rank * rec_bytes;
    if (!(x->fail_mode == 3 && seq > x->fail_after))   // (injected: a rank that stops publishing)
        hipLaunchKernelGGL(exchange_push_kernel, dim3(world), dim3(256), 0, st, (const uint32_t*)my_records, (int)(rec_bytes / 4),
                           x->peers_dev, off, h->cfg.rank, parity, seq);
    ICEM_HIP_TRY(hipGetLastError());
    wait_out->flags = reinterpret_cast<const unsigned*>(x->block) + parity * XCHG_MAX_WORLD;
    wait_out->status = x->status_dev;
    wait_out->seq = seq;
    wait_out->world = x->loopback ? 1 : world;
    wait_out->max_polls = x->max_polls;
    wait_out->records = x->block + x->rec_off + (size_t)parity * x->rec_slot;
    return ICEM_OK;
}

// the same exchange with the push folded into the caller's own kernel (pack_records_kernel): arguments for it
int xchg_begin(icem_handle* h, XchgPush* push_out, XchgWait* wait_out) {
    Exchange* x = h->xchg;
    if (!x || !x->connected) return fail(ICEM_E_STATE, "icem_exchange_connect has not been called");
    const int world = h->cfg.world, K = h->cfg.num_elites;
    const size_t rec_bytes = (size_t)K * (h->hd + 2) * h->tsize;
    const unsigned seq = ++x->seq;
    const int parity = (int)(seq & 1u);
    push_out->peers = x->peers_dev;
    push_out->rec_byte_off = x->rec_off + (size_t)parity * x->rec_slot + (size_t)h->cfg.rank * rec_bytes;
    push_out->world = (x->fail_mode == 3 && seq > x->fail_after) ? 0 : world;   // (injected: a rank that stops publishing)
    push_out->flag_idx = parity * XCHG_MAX_WORLD + h->cfg.rank;
    push_out->seq = seq;
    wait_out->flags = reinterpret_cast<const unsigned*>(x->block) + parity * XCHG_MAX_WORLD;
    wait_out->status = x->status_dev;
    wait_out->seq = seq;
    wait_out->world = x->loopback ? 1 : world;
    wait_out->max_polls = x->max_polls;
    wait_out->records = x->block + x->rec_off + (size_t)parity * x->rec_slot;
    return ICEM_OK;
}

bool xchg_connected(const icem_handle* h) { return h->xchg && h->xchg->connected; }
// what the kernels have reported so far (bit 0: a wait for a peer timed out); a plain read of host memory, not cleared
unsigned xchg_status_peek(const icem_handle* h) {
    return (h->xchg && h->xchg->status_host) ? __atomic_load_n(h->xchg->status_host, __ATOMIC_ACQUIRE) : 0u;
}
bool xchg_concurrent_peers(const icem_handle* h) { return xchg_connected(h) && !h->xchg->local_peers; }

void xchg_destroy(icem_handle* h) {
    Exchange* x = h->xchg;
    if (!x) return;
    for (size_t r = 0; r < x->peers.size(); ++r)
        if (x->opened[r] && x->peers[r]) (void)hipIpcCloseMemHandle(x->peers[r]);
    if (x->peers_dev) (void)hipFree(x->peers_dev);
    if (x->ticks_dev) (void)hipFree(x->ticks_dev);
    if (x->status_host) (void)hipHostFree(x->status_host);
    if (x->block) (void)hipFree(x->block);
    delete x;
    h->xchg = nullptr;
}

}  // namespace icem

using namespace icem;

extern "C" {

int icem_exchange_create(icem_handle* h, void* ipc_out_host) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (!ipc_out_host) return fail(ICEM_E_INVALID, "null output");
    if (h->cfg.world > XCHG_MAX_WORLD) return fail(ICEM_E_UNSUPPORTED, "world > 16");
    static_assert(sizeof(hipIpcMemHandle_t) <= ICEM_IPC_HANDLE_BYTES, "IPC handle size");
    xchg_destroy(h);
    Exchange* x = new Exchange();
    const size_t rec_bytes = (size_t)h->cfg.num_elites * (h->hd + 2) * h->tsize;
    x->rec_off = 256;  // flags (128 B) + status, padded
    x->rec_slot = (size_t)h->cfg.world * rec_bytes;
    x->bytes = x->rec_off + 2 * x->rec_slot;
    void* p = nullptr;
    hipIpcMemHandle_t ipc;
    std::memset(&ipc, 0, sizeof(ipc));
    // fine-grained device memory first: peers' stores must not be shadowed by this GPU's L2
    if (hipExtMallocWithFlags(&p, x->bytes, hipDeviceMallocFinegrained) == hipSuccess && p) {
        if (hipIpcGetMemHandle(&ipc, p) == hipSuccess) {
            x->finegrained = true;
        } else {
            (void)hipGetLastError();
            (void)hipFree(p);
            p = nullptr;
        }
    } else {
        (void)hipGetLastError();
        p = nullptr;
    }
    if (!p) {
        hipError_t e = hipMalloc(&p, x->bytes);
        if (e != hipSuccess) {
            delete x;
            return fail(ICEM_E_HIP, std::string("hipMalloc(exchange block): ") + hipGetErrorString(e));
        }
        e = hipIpcGetMemHandle(&ipc, p);
        if (e != hipSuccess) {
            (void)hipGetLastError();
            std::memset(&ipc, 0, sizeof(ipc));  // same-process peers can still connect by pointer
        }
    }
    hipError_t e = hipMemset(p, 0, x->bytes);
    if (e != hipSuccess) {
        (void)hipFree(p);
        delete x;
        return fail(ICEM_E_HIP, std::string("hipMemset(exchange block): ") + hipGetErrorString(e));
    }
    x->block = (unsigned char*)p;
    parse_fault(x, h->cfg.rank, h->cfg.world);
    // the status word: pinned host memory the kernels write on a timed-out wait (rare) and the host reads for free
    e = hipHostMalloc((void**)&x->status_host, 64, hipHostMallocMapped | hipHostMallocCoherent);
    if (e == hipSuccess) e = hipHostGetDevicePointer((void**)&x->status_dev, x->status_host, 0);
    if (e != hipSuccess) {
        if (x->status_host) (void)hipHostFree(x->status_host);
        (void)hipFree(p);
        delete x;
        return fail(ICEM_E_HIP, std::string("hipHostMalloc(exchange status): ") + hipGetErrorString(e));
    }
    std::memset(x->status_host, 0, 64);
    std::memset(ipc_out_host, 0, ICEM_IPC_HANDLE_BYTES);
    std::memcpy(ipc_out_host, &ipc, sizeof(ipc));
    h->xchg = x;
    return ICEM_OK;
}

int icem_exchange_disable(icem_handle* h) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (h->pm_pending) return fail(ICEM_E_STATE, "a deferred merge is pending: finish the MPC step first");
    xchg_destroy(h);
    return ICEM_OK;
}

void* icem_exchange_block(icem_handle* h) { return (h && h->xchg) ? h->xchg->block : nullptr; }

int icem_exchange_connect(icem_handle* h, const void* handles_host, void* const* local_blocks) {
    if (check_handle(h)) return ICEM_E_INVALID;
    Exchange* x = h->xchg;
    if (!x) return fail(ICEM_E_STATE, "icem_exchange_create must be called first");
    // option xchg_loopback = 1 (tools/sharded_rank_bench.py): time ONE rank of a sharded run without its peers -- every
    // push lands in this rank's own block and the merges wait for this rank's flag only
    x->loopback = opt_i(OPT_XCHG_LOOPBACK) != 0 && h->cfg.rank == 0;
    if (!handles_host && !local_blocks && !x->loopback) return fail(ICEM_E_INVALID, "neither IPC handles nor local block pointers");
    if (x->fail_mode == 1) return fail(ICEM_E_HIP, "injected fault (ICEM_XCHG_FAIL=connect): mapping the peers' exchange blocks failed on this rank");
    const int world = h->cfg.world, rank = h->cfg.rank;
    x->peers.assign(world, nullptr);
    x->opened.assign(world, false);
    x->local_peers = false;
    for (int r = 0; r < world; ++r) {
        if (r == rank || x->loopback) {
            x->peers[r] = x->block;
        } else if (local_blocks && local_blocks[r]) {
            x->peers[r] = (unsigned char*)local_blocks[r];
            x->local_peers = true;
        } else {
            if (!handles_host) return fail(ICEM_E_INVALID, "no IPC handle for a rank outside this process");
            hipIpcMemHandle_t ipc;
            std::memcpy(&ipc, (const unsigned char*)handles_host + (size_t)r * ICEM_IPC_HANDLE_BYTES, sizeof(ipc));
            void* p = nullptr;
            ICEM_HIP_TRY(hipIpcOpenMemHandle(&p, ipc, hipIpcMemLazyEnablePeerAccess));
            x->peers[r] = (unsigned char*)p;
            x->opened[r] = true;
        }
    }
    if (!x->peers_dev) ICEM_HIP_TRY(hipMalloc((void**)&x->peers_dev, (size_t)XCHG_MAX_WORLD * sizeof(unsigned char*)));
    ICEM_HIP_TRY(hipMemcpy(x->peers_dev, x->peers.data(), (size_t)world * sizeof(unsigned char*), hipMemcpyHostToDevice));
    if (opt(OPT_XCHG_MAX_POLLS) >= 1.0) x->max_polls = (unsigned)opt(OPT_XCHG_MAX_POLLS);
    x->connected = true;
    return ICEM_OK;
}

int icem_exchange_status(icem_handle* h, int32_t* status_host, int32_t* finegrained_host) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (!h->xchg || !status_host) return fail(ICEM_E_STATE, "no exchange block / null output");
    // (read and clear: host-mapped memory, no copy command and no synchronisation -- kernels still in flight report later)
    const unsigned s = __atomic_exchange_n(h->xchg->status_host, 0u, __ATOMIC_ACQ_REL);
    *status_host = (int32_t)s;
    if (finegrained_host) *finegrained_host = h->xchg->finegrained ? 1 : 0;
    return ICEM_OK;
}

// Measurement: average latency [us] of one in-library exchange (push of K records to every rank + wait for all ranks'),
// over `rounds` back-to-back exchanges inside one launch.  Collective: every rank calls it at the same time; synchronises
// the stream.
int icem_exchange_probe(icem_handle* h, int32_t rounds, void* stream, double* us_out) {
    if (check_handle(h)) return ICEM_E_INVALID;
    Exchange* x = h->xchg;
    if (!x || !x->connected || !us_out || rounds < 1) return fail(ICEM_E_STATE, "exchange not connected / bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const size_t rec_bytes = (size_t)h->cfg.num_elites * (h->hd + 2) * h->tsize;
    if (!x->ticks_dev) ICEM_HIP_TRY(hipMalloc((void**)&x->ticks_dev, sizeof(long long)));
    hipLaunchKernelGGL(exchange_probe_kernel, dim3(1), dim3(256), 0, st, (int)(rec_bytes / 4), x->peers_dev, x->rec_off, x->rec_slot,
                       rec_bytes, h->cfg.rank, h->cfg.world, x->loopback ? 1 : h->cfg.world, x->probe_seq, rounds, x->max_polls,
                       x->status_dev, x->ticks_dev);
    x->probe_seq += (unsigned)rounds;
    ICEM_HIP_TRY(hipGetLastError());
    ICEM_HIP_TRY(hipStreamSynchronize(st));
    if (x->fail_mode == 2) __atomic_fetch_or(x->status_host, 2u, __ATOMIC_ACQ_REL);   // injected: "the payload check failed"
    long long ticks = 0;
    ICEM_HIP_TRY(hipMemcpy(&ticks, x->ticks_dev, sizeof(ticks), hipMemcpyDeviceToHost));
    *us_out = (double)ticks / 100.0 / (double)rounds;  // wall_clock64: 100 MHz
    return ICEM_OK;
}

}  // extern "C"
