// exchange.hip -- the in-library elite exchange between the GPUs of one node (world > 1): the all-gather of every
// rank's K sorted candidate records before the replicated refit, without a host call between an iteration's local
// launch and its merge.  Reference analogue: the pipe gather of icem/models/gt_par_model.py:77-94.
//
// Every rank owns ONE exchange block in its own HBM:
//     flags   [2][XCHG_MAX_WORLD] u32   sequence number of the last push of rank r, per parity
//     status  u32                        != 0: a wait for a peer timed out
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
    unsigned seq = 0;                            // pushes so far (= sequence number of the last one)
    unsigned probe_seq = 0;
    long long* ticks_dev = nullptr;
    bool loopback = false;                       // measurement: every "peer" is this rank's own block, waits look at rank 0 only
    unsigned max_polls = XCHG_MAX_POLLS;         // ICEM_XCHG_MAX_POLLS overrides (tests of the timeout path)
    bool connected = false;
    // a peer connected by pointer lives in this process, possibly behind the same stream: its launches are ordered with
    // ours, so nothing of ours may wait for something it has not launched yet (no riding pack)
    bool local_peers = false;
};

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
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(reinterpret_cast<unsigned*>(blk) + flag_idx, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

__global__ __launch_bounds__(256) void exchange_push_kernel(const uint32_t* __restrict__ mine, int words, unsigned char* const* peers,
                                                            size_t rec_byte_off, int rank, int parity, unsigned seq) {
    push_to_block(mine, words, peers[blockIdx.x], rec_byte_off, parity * XCHG_MAX_WORLD + rank, seq);
}

// Measurement only: `rounds` back-to-back exchanges (every rank pushes its K records to every block, then waits for
// all of them) inside ONE launch per rank, timed with the 100 MHz wall clock: the latency of one exchange as the
// planning loop sees it, without kernel launches around it.  Uses its own flag words (probe area) and parity slots.
__global__ __launch_bounds__(256) void exchange_probe_kernel(const uint32_t* __restrict__ mine, int words, unsigned char* const* peers,
                                                             size_t rec_off, size_t rec_slot, size_t rec_bytes, int rank, int world,
                                                             unsigned base, int rounds, long long* ticks_out) {
    unsigned char* own = peers[rank];
    const long long t0 = wall_clock64();
    for (int r = 1; r <= rounds; ++r) {
        const unsigned seq = base + (unsigned)r;
        const int parity = (int)(seq & 1u);
        for (int p = 0; p < world; ++p) {
            push_to_block(mine, words, peers[p], rec_off + (size_t)parity * rec_slot + (size_t)rank * rec_bytes,
                          XCHG_PROBE_FLAG0 + rank, seq);
            __syncthreads();
        }
        if (threadIdx.x < 64) {
            XchgWait w;
            w.flags = reinterpret_cast<const unsigned*>(own) + XCHG_PROBE_FLAG0;
            w.status = reinterpret_cast<unsigned*>(own) + 2 * XCHG_MAX_WORLD;
            w.seq = seq;
            w.world = world;
            w.max_polls = XCHG_MAX_POLLS;
            xchg_wait_at_least(w, threadIdx.x);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) ticks_out[0] = wall_clock64() - t0;
}

}  // namespace

size_t xchg_status_off() { return (size_t)2 * XCHG_MAX_WORLD * sizeof(unsigned); }

// this rank's records of the running iteration -> every rank's block; returns the arguments the merge waits with
int xchg_push(icem_handle* h, const void* my_records, hipStream_t st, XchgWait* wait_out) {
    Exchange* x = h->xchg;
    if (!x || !x->connected) return fail(ICEM_E_STATE, "icem_exchange_connect has not been called");
    const int world = h->cfg.world, K = h->cfg.num_elites;
    const size_t rec_bytes = (size_t)K * (h->hd + 2) * h->tsize;
    const unsigned seq = ++x->seq;
    const int parity = (int)(seq & 1u);
    const size_t off = x->rec_off + (size_t)parity * x->rec_slot + (size_t)h->cfg.rank * rec_bytes;
    hipLaunchKernelGGL(exchange_push_kernel, dim3(world), dim3(256), 0, st, (const uint32_t*)my_records, (int)(rec_bytes / 4),
                       x->peers_dev, off, h->cfg.rank, parity, seq);
    ICEM_HIP_TRY(hipGetLastError());
    wait_out->flags = reinterpret_cast<const unsigned*>(x->block) + parity * XCHG_MAX_WORLD;
    wait_out->status = reinterpret_cast<unsigned*>(x->block + xchg_status_off());
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
    push_out->world = world;
    push_out->flag_idx = parity * XCHG_MAX_WORLD + h->cfg.rank;
    push_out->seq = seq;
    wait_out->flags = reinterpret_cast<const unsigned*>(x->block) + parity * XCHG_MAX_WORLD;
    wait_out->status = reinterpret_cast<unsigned*>(x->block + xchg_status_off());
    wait_out->seq = seq;
    wait_out->world = x->loopback ? 1 : world;
    wait_out->max_polls = x->max_polls;
    wait_out->records = x->block + x->rec_off + (size_t)parity * x->rec_slot;
    return ICEM_OK;
}

bool xchg_connected(const icem_handle* h) { return h->xchg && h->xchg->connected; }
bool xchg_concurrent_peers(const icem_handle* h) { return xchg_connected(h) && !h->xchg->local_peers; }

void xchg_destroy(icem_handle* h) {
    Exchange* x = h->xchg;
    if (!x) return;
    for (size_t r = 0; r < x->peers.size(); ++r)
        if (x->opened[r] && x->peers[r]) (void)hipIpcCloseMemHandle(x->peers[r]);
    if (x->peers_dev) (void)hipFree(x->peers_dev);
    if (x->ticks_dev) (void)hipFree(x->ticks_dev);
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
    // ICEM_XCHG_LOOPBACK=1 (tools/sharded_rank_bench.py): time ONE rank of a sharded run without its peers -- every
    // push lands in this rank's own block and the merges wait for this rank's flag only
    const char* lb = getenv("ICEM_XCHG_LOOPBACK");
    x->loopback = lb && atoi(lb) != 0 && h->cfg.rank == 0;
    if (!handles_host && !local_blocks && !x->loopback) return fail(ICEM_E_INVALID, "neither IPC handles nor local block pointers");
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
    if (const char* e = getenv("ICEM_XCHG_MAX_POLLS")) x->max_polls = (unsigned)std::max(1, atoi(e));
    x->connected = true;
    return ICEM_OK;
}

int icem_exchange_status(icem_handle* h, int32_t* status_host, int32_t* finegrained_host) {
    if (check_handle(h)) return ICEM_E_INVALID;
    if (!h->xchg || !status_host) return fail(ICEM_E_STATE, "no exchange block / null output");
    unsigned s = 0;
    ICEM_HIP_TRY(hipMemcpy(&s, h->xchg->block + xchg_status_off(), sizeof(s), hipMemcpyDeviceToHost));
    *status_host = (int32_t)s;
    if (finegrained_host) *finegrained_host = h->xchg->finegrained ? 1 : 0;
    if (s) {
        const unsigned zero = 0;
        ICEM_HIP_TRY(hipMemcpy(h->xchg->block + xchg_status_off(), &zero, sizeof(zero), hipMemcpyHostToDevice));
    }
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
    // the payload: whatever sits in this rank's slot of its own block
    const unsigned char* mine = x->block + x->rec_off + (size_t)h->cfg.rank * rec_bytes;
    hipLaunchKernelGGL(exchange_probe_kernel, dim3(1), dim3(256), 0, st, (const uint32_t*)mine, (int)(rec_bytes / 4), x->peers_dev,
                       x->rec_off, x->rec_slot, rec_bytes, h->cfg.rank, h->cfg.world, x->probe_seq, rounds, x->ticks_dev);
    x->probe_seq += (unsigned)rounds;
    ICEM_HIP_TRY(hipGetLastError());
    ICEM_HIP_TRY(hipStreamSynchronize(st));
    long long ticks = 0;
    ICEM_HIP_TRY(hipMemcpy(&ticks, x->ticks_dev, sizeof(ticks), hipMemcpyDeviceToHost));
    *us_out = (double)ticks / 100.0 / (double)rounds;  // wall_clock64: 100 MHz
    return ICEM_OK;
}

}  // extern "C"
