// Populations of up to 65 536 trajectories (4 096 tiles of 16) of the recurrent state-space model rollout (icem_rssm.h).  The
// fused kernel of icem_rssm.hip puts one workgroup per tile on a CU and streams ALL weights (0.9 MB per model step)
// through that CU's L1, reward head included, on the recurrence's own critical path.  Here ONE launch holds two kinds
// of workgroups:
//   * blocks [0, tiles): the RECURRENCE of one tile -- x, GRU, p, z' per model step, four barriers, nothing else; the
//     last of the `horizon` transitions is never run (a step's cost is the reward of the state it STARTS from);
//   * blocks [tiles, 2 tiles): the REWARD HEAD of one tile, its weights loaded ONCE into registers (148 per lane),
//     consuming the states (h_t, z_t) as the recurrence publishes them.  (Above 256 tiles: two tiles per recurrence
//     workgroup, which share every weight chunk, and 512 reward workgroups that walk the tiles.)
// State t of a tile travels through an 8 KB item in global memory: wave 7 of the recurrence workgroup -- which owns one
// GRU output block where waves 0..4 own two -- copies [h_t | z_t] out of LDS during phase 2 of step t with written-through
// stores, waits for them itself and raises the tile's flag to t + 1; no other wave ever waits for a store.  The reward
// workgroup polls the flag (one wave), reads the item with agent-scope loads (once, into LDS), and resets the flag to 0
// behind the last item (so a captured launch can be replayed).  Producers have the lower block indices: they are all
// dispatched before the first consumer and they wait for nobody -- up to 128 tiles both kinds are resident together
// (a reward workgroup runs a few microseconds behind its recurrence); beyond that the reward workgroups take the CUs the
// recurrence workgroups leave and find their states waiting.
// The arithmetic, its order and its rounding points are those of rssm_rollout_kernel<1>: costs are bit-identical.
#include "rssm_dev.h"

#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>

namespace icem {
namespace {
using namespace rssm_dev;

constexpr int SR = 256;              // bf16 row of an item: [h: 7 k-blocks (200 + zero padding) | z: 1 k-block]
constexpr int ITEM = 16 * SR;        // elements
constexpr int XS = SR + 8;           // its row stride in the reward head's LDS (conflict-free operand reads)
// two tiles per recurrence workgroup: resident chunks / ring slots of the two wave classes (accumulators take twice the room)
constexpr int RES_A2 = 2, RING_A2 = 3, RES_B2 = 3, RING_B2 = 2;
constexpr int LDS_CHUNKS = 10;       // chunks parked in LDS (one tile per workgroup): two per class-A wave
// Where each of a wave's chunks lives, in the order the wave consumes them every step (recurrence_steps):
//   REG  in registers for the whole rollout;  LDS  in ProducerLds::wl for the whole rollout;
//   STR  streamed: the streamed chunks rotate through a ring of RING register slots, the i-th one's slot refilled with
//        the (i + RING)-th right behind its MFMAs.
// Every ring slot is refilled (streamed chunks / RING) times per step and a refill is one loaded L2 round trip (1.2 us with
// 70 KB in flight per CU, 1.7 with 105), so a step cannot be shorter than that many round trips: WHERE the streamed
// chunks sit in the order matters as much as how many they are.  With all of them in phase 2 that phase was two exposed
// round trips long (3.7 us of a 5.3 us step); spread two by two between the resident ones, over all four phases, a
// request is a quarter of a step ahead of its use (EXPERIMENTS R4.11: six placements measured).
enum { REG = 0, STR = 1, LDS = 2 };
#ifndef ICEM_RSSM_PLAN
#define ICEM_RSSM_PLAN 0   // (development: other placements, measured in EXPERIMENTS R4.11)
#endif
// element offset of chunk c of the class-A wave that owns GRU / W4 blocks ob0 and ob1 (the order the wave consumes them in)
__host__ __device__ constexpr size_t class_a_chunk(int c, int ob0, int ob1) {
    return c < 3 ? WGH + (size_t)(c * DETB + ob0) * DETK * BLK
         : c < 6 ? WGH + (size_t)((c - 3) * DETB + ob1) * DETK * BLK
         : c < 9 ? WGI + (size_t)((c - 6) * DETB + ob0) * HIDK * BLK
         : c < 12 ? WGI + (size_t)((c - 9) * DETB + ob1) * HIDK * BLK
         : W4 + (size_t)(c == 12 ? ob0 : ob1) * DETK * BLK;
}
template <bool CLASS_A, int TT>
struct ChunkPlan {
    static constexpr int NCH = CLASS_A ? 14 : 7;
    static constexpr int RING = TT == 1 ? (CLASS_A ? (ICEM_RSSM_PLAN >= 1 && ICEM_RSSM_PLAN <= 3 ? 3 : 2) : 0) : (CLASS_A ? RING_A2 : RING_B2);
    static constexpr int kind(int c) {
        if (TT == 1) {
            if (!CLASS_A) return REG;
            // H(w) r u n | H(w+8) r u n | I(w) r u n | I(w+8) r u n | W4(w) W4(w+8)   (class_a_chunk)
#if ICEM_RSSM_PLAN == 1     // three ring slots, the two LDS chunks where the first form stalled
            return c == 4 || c == 7 ? LDS : c == 2 || c == 5 || c == 8 ? REG : STR;
#elif ICEM_RSSM_PLAN == 2   // three ring slots, every gate's n chunk resident
            return c == 11 || c == 13 ? LDS : c == 2 || c == 5 || c == 8 ? REG : STR;
#elif ICEM_RSSM_PLAN == 3
            return c == 4 || c == 10 ? LDS : c == 2 || c == 7 || c == 12 ? REG : STR;
#elif ICEM_RSSM_PLAN == 4   // two ring slots, the streamed pairs a quarter of a step apart
            return c == 8 || c == 11 ? LDS : c == 2 || c == 3 || c == 6 || c == 7 ? REG : STR;
#elif ICEM_RSSM_PLAN == 5
            return c == 10 || c == 11 ? LDS : c == 2 || c == 3 || c == 6 || c == 7 ? REG : STR;
#else                       // two ring slots: four chunks in registers, eight streamed
            return c == 11 || c == 13 ? LDS : c == 2 || c == 5 || c == 8 || c == 12 ? REG : STR;
#endif
        }
        return c < (CLASS_A ? RES_A2 : RES_B2) ? REG : STR;
    }
    static constexpr int count(int k, int upto) {
        int n = 0;
        for (int c = 0; c < upto; ++c) n += kind(c) == k;
        return n;
    }
    static constexpr int RES = count(REG, NCH), NSTR = count(STR, NCH);
    static_assert(count(LDS, NCH) == (TT == 1 && CLASS_A ? 2 : 0), "two LDS chunks per class-A wave (ProducerLds::wl)");
    static_assert(RING == 0 ? NSTR == 0 : NSTR % RING == 0, "a chunk's slot must not depend on the step");
    static constexpr int nth_streamed(int i) {
        for (int c = 0, n = 0; c < NCH; ++c)
            if (kind(c) == STR && n++ == i) return c;
        return -1;
    }
    struct Tab { int kind[NCH], slot[NCH], next[NCH], first[RING ? RING : 1], lds[2]; };
    static constexpr Tab make() {
        Tab t{};
        for (int c = 0; c < NCH; ++c) {
            t.kind[c] = kind(c);
            t.slot[c] = kind(c) == REG ? count(REG, c) : kind(c) == STR ? RES + count(STR, c) % (RING ? RING : 1) : count(LDS, c);
            if (kind(c) == LDS) t.lds[count(LDS, c)] = c;
            t.next[c] = kind(c) == STR ? nth_streamed((count(STR, c) + RING) % (NSTR ? NSTR : 1)) : -1;
        }
        for (int i = 0; i < RING; ++i) t.first[i] = nth_streamed(i);
        return t;
    }
    static constexpr Tab tab = make();
};
constexpr unsigned MAX_POLLS = 1u << 22;   // x s_sleep(2): a few hundred ms, then the reward workgroup gives up (costs = NaN)

template <int TT>
struct ProducerLds {                       // TT tiles of 16 trajectories: rows tt * 16 + j
    unsigned short zA[TT * 16 * ZS];
    unsigned short hb[2][TT * 16 * RS];
    unsigned short xb[TT * 16 * RS];
    float h32[TT * 16 * HS];
    float bs[NBIAS];
    unsigned short w1[HIDB * K1K * BLK];   // A-operand blocks that stay here: W1 and W5
    unsigned short w5[STB * HIDK * BLK];
    float ob[232];                         // obs0, parked once
    // One tile per workgroup leaves 73 KB of the CU's LDS free: two chunks of every class-A wave (ChunkPlan: the n gate's
    // input side and W4, both of block w + 8) stay here for the whole rollout -- 70 KB less to stream every model step.
    unsigned short wl[TT == 1 ? LDS_CHUNKS * DETK * BLK : 8];
};
struct ConsumerLds {
    unsigned short x[2][16 * XS];          // the state at hand (ping-pong)
    unsigned short r1[2][16 * RS];
    unsigned short r2[2][16 * RS];
    int gave_up, avail[2];
};
template <int TT>
constexpr size_t lds_bytes() { return sizeof(ProducerLds<TT>) > sizeof(ConsumerLds) ? sizeof(ProducerLds<TT>) : sizeof(ConsumerLds); }

__device__ __forceinline__ void store_wt(unsigned short* p, unsigned long long v) {   // written through, 8 bytes
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long load_ag(const unsigned short* p) {     // past the non-coherent caches, 8 bytes
    return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// wave-wide copy of state t of the workgroup's tiles out of LDS into their items, then the flags (called by ONE wave);
// items / flags: those of the workgroup's first tile; ntl: how many of its TT tiles exist
template <int TT>
__device__ __forceinline__ void publish(const ProducerLds<TT>& s, int cur, unsigned short* items, size_t tile_stride, int t,
                                        unsigned* flag, int ntl, int lane) {
    // h: 16 rows x 208 bf16 = 52 8-byte words per row; z: 16 rows x 32 bf16 = 8 words per row
    asm volatile("" : "+v"(lane));   // (the 15 address pairs are step-invariant: hoisted out of the step loop they are spilled)
#pragma unroll
    for (int tt = 0; tt < TT; ++tt) {
        if (tt < ntl) {
            unsigned short* item = items + tt * tile_stride + (size_t)t * ITEM;
#pragma unroll
            for (int i = 0; i < 13; ++i) {
                const int e = lane + 64 * i, row = e / 52, c = e % 52;
                store_wt(item + row * SR + 4 * c, *reinterpret_cast<const unsigned long long*>(s.hb[cur] + (tt * 16 + row) * RS + 4 * c));
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int e = lane + 64 * i, row = e / 8, c = e % 8;
                store_wt(item + row * SR + 224 + 4 * c, *reinterpret_cast<const unsigned long long*>(s.zA + (tt * 16 + row) * ZS + 4 * c));
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane < ntl) __hip_atomic_store(flag + lane, (unsigned)t + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// The model steps of the recurrence for one CLASS of waves.  The 0.65 MB of weights a step reads are cut into CHUNKS
// (one output block of one matrix: 7 A-operand blocks = 28 registers per lane, 7 KB per wave) and every chunk has an owner:
//   class A, waves 0..4:  the GRU's output blocks w and w + 8, W4's blocks w and w + 8                  14 chunks
//   class B, waves 5..7:  the GRU's block w, W4's block w          7 chunks: H(w) r u n, I(w) r u n, W4(w)
// (H = hidden side, needs h_t only; I = input side, needs x: phase 2; W4: phase 3).  W1 and W5 live in LDS.
// A wave consumes its chunks in a fixed order every step; ChunkPlan says where each one lives: in registers for the
// whole rollout, in LDS for the whole rollout (one tile per workgroup: 70 KB of the CU's LDS are free), or streamed
// through a ring of RING register slots, a slot's next chunk requested right behind the MFMAs of the one it held.
// The two classes are two instances of this function behind one wave-uniform branch: inside each the code is
// straight-line, so the compiler's in-order vmcnt bookkeeping is exact (a wave only ever waits for the chunk it is
// about to use), and nobody issues a load it does not need (a "dummy" broadcast load costs the L1 more than a real
// one: measured).  TT tiles per workgroup share every chunk (one request, TT accumulation chains).
template <bool CLASS_A, int TT>
__device__ __forceinline__ void recurrence_steps(ProducerLds<TT>& s, int w, int lane, int base, int n, int horizon,
                                                 const unsigned short* __restrict__ Pg, const float* __restrict__ actions,
                                                 unsigned short* items, size_t tile_stride, unsigned* flag, int ntl,
                                                 long long* stamps, bool stamp) {
    using CP = ChunkPlan<CLASS_A, TT>;
    constexpr int NCH = CP::NCH, RES = CP::RES, RING = CP::RING;
    // one tile per workgroup: the hidden side runs ahead of the step (see hidden0 below); two tiles: as written in the
    // header comment (their accumulators take twice the room, and what is bound there is the matrix pipe)
    constexpr bool EARLY = TT == 1;
    const int ob0 = w, ob1 = w + WAVES;
    v4i slot[RES + RING][DETK];
    // (c is a constant after unrolling)   class A: H(w) r u n | H(w+8) r u n | I(w) r u n | I(w+8) r u n | W4(w) W4(w+8)
    //                                      class B: H(w) r u n | I(w) r u n | W4(w)
    auto chunk = [&](gptr P, int cpos, int l8) -> gptr {
        const int c = EARLY || !CLASS_A ? cpos : cpos < 3 ? cpos : cpos < 6 ? cpos + 3 : cpos < 9 ? cpos - 3 : cpos;   // (not EARLY: H I H I W4 W4)
        if (c < 3) return P + WGH + (size_t)(c * DETB + ob0) * DETK * BLK + l8;
        if (!CLASS_A) {
            if (c < 6) return P + WGI + (size_t)((c - 3) * DETB + ob0) * HIDK * BLK + l8;
            return P + W4 + (size_t)ob0 * DETK * BLK + l8;
        }
        if (c < 6) return P + WGH + (size_t)((c - 3) * DETB + ob1) * DETK * BLK + l8;
        if (c < 9) return P + WGI + (size_t)((c - 6) * DETB + ob0) * HIDK * BLK + l8;
        if (c < 12) return P + WGI + (size_t)((c - 9) * DETB + ob1) * HIDK * BLK + l8;
        return P + W4 + (size_t)(c == 12 ? ob0 : ob1) * DETK * BLK + l8;   // (class_a_chunk, spelled out: through the function one register more is live, and spilled)
    };
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < NCH; ++c)
        if (CP::tab.kind[c] == REG) request<DETK>(chunk((gptr)Pg, c, lane * 8), slot[CP::tab.slot[c]]);
#pragma unroll
    for (int i = 0; i < RING; ++i) request<DETK>(chunk((gptr)Pg, CP::tab.first[i], lane * 8), slot[RES + i]);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    if (stamp) stamps[1] = wall_clock64();
    int cur = 0;
    // (the parameters do not depend on t: re-derive the pointer behind an opaque barrier every step, or the optimizer
    // keeps what fits of them in registers across steps and spills -- see icem_rssm.hip; and the lane's offsets are
    // recomputed every step: carried across the loop they are spilled, and a scratch reload is a vmcnt(0) wait in the
    // middle of the weight requests)
    gptr P;
    int j, g, l8, xr, zr, xo, zo, ho;
    auto derive = [&]() {
        P = (gptr)Pg;
        asm volatile("" : "+s"(P));
        int ln = lane;
        asm volatile("" : "+v"(ln));
        j = ln & 15; g = ln >> 4; l8 = ln * 8;
        xr = j * RS + 8 * g; zr = j * ZS + 8 * g;
        xo = j * RS + 4 * g; zo = j * ZS + 4 * g; ho = j * HS + 4 * g;
    };
    // chunk c through the matrix pipe for every tile (B: the lane's operand row in tile 0, bts: its stride between
    // tiles); if it is a streamed chunk, the request that refills its slot
    auto use = [&](int c, const unsigned short* B, int bts, v4f (&acc)[TT], v4f bias) {
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) acc[tt] = bias;
        if (CP::tab.kind[c] == LDS) {   // (one tile per workgroup) both operands from LDS; the same chain in the same order
            const unsigned short* A = s.wl + (size_t)(2 * w + CP::tab.slot[c]) * DETK * BLK + l8;
#pragma unroll
            for (int kb = 0; kb < DETK; ++kb)
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(v8bf, *reinterpret_cast<const v4i*>(A + (size_t)kb * BLK)),
                    __builtin_bit_cast(v8bf, *reinterpret_cast<const v4i*>(B + kb * 32)), acc[0], 0, 0, 0);
            return;
        }
#pragma unroll
        for (int kb = 0; kb < DETK; ++kb)
#pragma unroll
            for (int tt = 0; tt < TT; ++tt)
                acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                    __builtin_bit_cast(v8bf, slot[CP::tab.slot[c]][kb]),
                    __builtin_bit_cast(v8bf, *reinterpret_cast<const v4i*>(B + tt * bts + kb * 32)), acc[tt], 0, 0, 0);
        if (CP::tab.kind[c] == STR) {
            __builtin_amdgcn_sched_barrier(0);
            request<DETK>(chunk(P, CP::tab.next[c], l8), slot[CP::tab.slot[c]]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto gates = [&](int ob, const v4f (&ir)[TT], const v4f (&iu)[TT], const v4f (&in)[TT], const v4f (&hr)[TT],
                     const v4f (&hu)[TT], const v4f (&hn)[TT]) {
#pragma unroll
        for (int tt = 0; tt < TT; ++tt) {
            float* hp = s.h32 + tt * 16 * HS + ho + ob * 16;
            float nh[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                nh[r] = gru_out(ir[tt][r], iu[tt][r], in[tt][r], hr[tt][r], hu[tt][r], hn[tt][r], hp[r]);
                hp[r] = nh[r];
            }
            *reinterpret_cast<v4s*>(s.hb[cur ^ 1] + tt * 16 * RS + xo + ob * 16) = pack4(nh[0], nh[1], nh[2], nh[3]);
        }
    };
    // The hidden side of the GRU needs h_t only and is off the step's dependency chain (z -> x -> input side -> gates -> h'
    // -> p -> z'): block w's runs beside z' at the END of the step before (phase 4), block w + 8's beside x (phase 1) --
    // the two phases that are otherwise a barrier's latency long, so that phase 2 keeps the input side and the gates only
    // and the streamed chunks' uses are spread over the whole step.
    v4f hr[TT], hu[TT], hn[TT];
    auto hidden0 = [&]() {
        const unsigned short* H = s.hb[cur] + xr;
        const int bi = ob0 * 16 + 4 * g;
        use(0, H, 16 * RS, hr, bias4(s.bs, BGH, bi));
        use(1, H, 16 * RS, hu, bias4(s.bs, BGH, 16 * DETB + bi));
        use(2, H, 16 * RS, hn, bias4(s.bs, BGH, 32 * DETB + bi));
    };
    derive();
    if (EARLY) hidden0();
    for (int t = 0; t + 1 < horizon; ++t) {
        const bool st = stamp && t == 5;
        if (st) stamps[2] = wall_clock64();
        derive();
        const unsigned short* X = s.xb + xr;
        const unsigned short* H = s.hb[cur] + xr;
        // the next action of the trajectories (every wave asks, wave 2 stores: a load inside a branch would cost the
        // waves behind the branch their exact vmcnt).  One tile: asked for in phase 3, a phase ahead of its use (four
        // registers less across phase 2, where the wave's pressure peaks -- and a spilled register is a scratch reload,
        // a vmcnt(0) wait among the weight requests); two tiles: at the step's top (measured: 3 % faster there)
        float an[TT][4];
        auto ask_actions = [&]() {
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) {
                const int rr = base + tt * 16 + j;
                const float* ap = actions + ((size_t)(rr < n ? rr : n - 1) * horizon + (t + 1)) * ACT;
                const int o = g & 1 ? 4 : 0;
                an[tt][0] = ap[o]; an[tt][1] = ap[o + 1]; an[tt][2] = ap[g & 1 ? 5 : 2]; an[tt][3] = ap[g & 1 ? 5 : 3];
            }
        };
        if (!EARLY) ask_actions();
        // ---- phase 1 (reads z_t, a_t, h_t): x = relu(W1 [z | a] + b1); class A: the hidden side of its second GRU block ----
#pragma unroll
        for (int i = 0; i < (CLASS_A ? 2 : 1); ++i) {
            const int ob = w + WAVES * i;
            v4i A1[K1K];
            request<K1K>(s.w1 + (size_t)ob * K1K * BLK + l8, A1);
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) {
                const v4f a = mma<K1K>(A1, s.zA + tt * 16 * ZS + zr, bias4(s.bs, B1, ob * 16 + 4 * g));
                *reinterpret_cast<v4s*>(s.xb + tt * 16 * RS + xo + ob * 16) = relu_pack(a);
            }
        }
        if (!EARLY) hidden0();
        v4f hr1[TT], hu1[TT], hn1[TT];   // (class A)
        auto hidden1 = [&](int c0) {
            const int bi = ob1 * 16 + 4 * g;
            use(c0, H, 16 * RS, hr1, bias4(s.bs, BGH, bi));
            use(c0 + 1, H, 16 * RS, hu1, bias4(s.bs, BGH, 16 * DETB + bi));
            use(c0 + 2, H, 16 * RS, hn1, bias4(s.bs, BGH, 32 * DETB + bi));
        };
        if constexpr (CLASS_A && EARLY) hidden1(3);
        __syncthreads();
        if (st) stamps[3] = wall_clock64();
        // ---- phase 2: the input side, GRU h' = (1 - u) n + u h; wave 7 publishes state t ----
        {
            v4f ir[TT], iu[TT], in[TT];
            int bi = ob0 * 16 + 4 * g;
            constexpr int i0 = CLASS_A && EARLY ? 6 : 3;
            use(i0, X, 16 * RS, ir, bias4(s.bs, BGI, bi));
            use(i0 + 1, X, 16 * RS, iu, bias4(s.bs, BGI, 16 * DETB + bi));
            use(i0 + 2, X, 16 * RS, in, bias4(s.bs, BGI, 32 * DETB + bi));
            gates(ob0, ir, iu, in, hr, hu, hn);
            if constexpr (CLASS_A) {
                if constexpr (!EARLY) hidden1(6);
                bi = ob1 * 16 + 4 * g;
                use(9, X, 16 * RS, ir, bias4(s.bs, BGI, bi));
                use(10, X, 16 * RS, iu, bias4(s.bs, BGI, 16 * DETB + bi));
                use(11, X, 16 * RS, in, bias4(s.bs, BGI, 32 * DETB + bi));
                gates(ob1, ir, iu, in, hr1, hu1, hn1);
            } else if (w == WAVES - 1) {
                int ln = lane;
                asm volatile("" : "+v"(ln));
                publish<TT>(s, cur, items, tile_stride, t, flag, ntl, ln);
            }
        }
        __syncthreads();
        if (st) stamps[4] = wall_clock64();
        // ---- phase 3: p = relu(W4 h' + b4) ----
        if (EARLY) ask_actions();
        {
            const unsigned short* Hn = s.hb[cur ^ 1] + xr;
            v4f p0[TT];
            use(CLASS_A ? 12 : 6, Hn, 16 * RS, p0, bias4(s.bs, B4, ob0 * 16 + 4 * g));
#pragma unroll
            for (int tt = 0; tt < TT; ++tt) *reinterpret_cast<v4s*>(s.xb + tt * 16 * RS + xo + ob0 * 16) = relu_pack(p0[tt]);
            if (CLASS_A) {
                v4f p1[TT];
                use(13, Hn, 16 * RS, p1, bias4(s.bs, B4, ob1 * 16 + 4 * g));
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) *reinterpret_cast<v4s*>(s.xb + tt * 16 * RS + xo + ob1 * 16) = relu_pack(p1[tt]);
            }
        }
        __syncthreads();
        if (st) stamps[5] = wall_clock64();
        // ---- phase 4: z' = W5 p + b5 (waves 0, 1, from LDS) and the next action (wave 2) -> [z | a]; every wave: the
        //      hidden side of its first GRU block for step t + 1 (behind the last step: computed and dropped, the ring
        //      of streamed chunks keeps its order) ----
        if (CLASS_A) {
            if (w < STB) {
                v4i A5[HIDK];
                request<HIDK>(s.w5 + (size_t)w * HIDK * BLK + l8, A5);
#pragma unroll
                for (int tt = 0; tt < TT; ++tt) {
                    const v4f a = mma<HIDK>(A5, s.xb + tt * 16 * RS + xr, bias4(s.bs, B5, w * 16 + 4 * g));
                    *reinterpret_cast<v4s*>(s.zA + tt * 16 * ZS + zo + w * 16) = pack4(a[0], a[1], a[2], a[3]);
                }
            } else if (w == STB && g < 2) {   // lanes (j, 0): a[0..3]; lanes (j, 1): a[4], a[5], 0, 0
#pragma unroll
                for (int tt = 0; tt < TT; ++tt)
                    *reinterpret_cast<v4s*>(s.zA + (tt * 16 + j) * ZS + 32 + 4 * g) =
                        pack4(an[tt][0], an[tt][1], g == 0 ? an[tt][2] : 0.f, g == 0 ? an[tt][3] : 0.f);
            }
        }
        cur ^= 1;
        if (EARLY) hidden0();
        __syncthreads();
        if (st) stamps[6] = wall_clock64();
    }
    if (stamp) stamps[7] = wall_clock64();
    // the state the last step starts from
    if (w == WAVES - 1) publish<TT>(s, cur, items, tile_stride, horizon - 1, flag, ntl, lane);
}

// wg: the workgroup's index among the recurrence workgroups (tiles wg * TT ..)
template <int TT>
__device__ __forceinline__ void recurrence(ProducerLds<TT>& s, int wg, int tiles, int n, int horizon, const unsigned short* __restrict__ Pg,
                                           const float* __restrict__ obs0, const float* __restrict__ actions,
                                           unsigned short* stage, unsigned* flags, long long* stamps) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tile0 = wg * TT, base = tile0 * 16;
    const int ntl = tiles - tile0 < TT ? tiles - tile0 : TT;
    const bool stamp = stamps && wg == 0 && tid == 0;   // development aid (icem_debug_stamps(NULL, buffer))
    if (stamp) stamps[0] = wall_clock64();
    const size_t tile_stride = (size_t)horizon * ITEM;
    unsigned short* items = stage + (size_t)tile0 * tile_stride;
    unsigned* flag = flags + tile0;
    // Initial state: everything this workgroup reads from global memory before its first model step is requested at
    // once (one round trip, not one per array), parked in LDS, and the activation rows are built from there.
    {
        const float ob = tid < DET + STOCH ? obs0[tid] : 0.f;
        constexpr size_t boffs[5] = {B1, BGI, BGH, B4, B5};
        float bv[5][2];
#pragma unroll
        for (int l = 0; l < 5; ++l)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int e = tid + NTHR * q;
                bv[l][q] = e < bias_len(boffs[l]) ? reinterpret_cast<const float*>(Pg + boffs[l])[e] : 0.f;
            }
        constexpr int N1 = HIDB * K1K * BLK / 8, N5 = STB * HIDK * BLK / 8;
        v4i c1[(N1 + NTHR - 1) / NTHR], c5[(N5 + NTHR - 1) / NTHR];
#pragma unroll
        for (int q = 0; q < (N1 + NTHR - 1) / NTHR; ++q)
            if (tid + NTHR * q < N1) c1[q] = reinterpret_cast<const v4i*>(Pg + W1)[tid + NTHR * q];
#pragma unroll
        for (int q = 0; q < (N5 + NTHR - 1) / NTHR; ++q)
            if (tid + NTHR * q < N5) c5[q] = reinterpret_cast<const v4i*>(Pg + W5)[tid + NTHR * q];
        float av = 0.f;
        if (tid < TT * 16 * ACT) {
            const int jj = tid / ACT;
            av = actions[(size_t)(base + jj < n ? base + jj : n - 1) * horizon * ACT + tid % ACT];
        }
        s.ob[tid < 232 ? tid : 231] = tid < 232 ? ob : 0.f;
#pragma unroll
        for (int l = 0; l < 5; ++l)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                if (tid + NTHR * q < bias_len(boffs[l])) s.bs[bias_slot(boffs[l]) + tid + NTHR * q] = bv[l][q];
#pragma unroll
        for (int q = 0; q < (N1 + NTHR - 1) / NTHR; ++q)
            if (tid + NTHR * q < N1) reinterpret_cast<v4i*>(s.w1)[tid + NTHR * q] = c1[q];
#pragma unroll
        for (int q = 0; q < (N5 + NTHR - 1) / NTHR; ++q)
            if (tid + NTHR * q < N5) reinterpret_cast<v4i*>(s.w5)[tid + NTHR * q] = c5[q];
        if (tid < TT * 16 * ACT) s.zA[(tid / ACT) * ZS + 32 + tid % ACT] = to_bf16(av);
        if constexpr (TT == 1) {   // the class-A waves' two LDS chunks (ChunkPlan)
            constexpr int PC = DETK * BLK / 8, NP = LDS_CHUNKS * PC;   // 16-byte pieces per chunk, in all
            v4i cl[(NP + NTHR - 1) / NTHR];
#pragma unroll
            for (int q = 0; q < (NP + NTHR - 1) / NTHR; ++q) {
                const int e = tid + NTHR * q;
                if (e < NP) {
                    const int ci = e / PC, wv = ci >> 1;
                    const size_t src = class_a_chunk(ChunkPlan<true, 1>::tab.lds[ci & 1], wv, wv + WAVES);
                    cl[q] = reinterpret_cast<const v4i*>(Pg + src)[e % PC];
                }
            }
#pragma unroll
            for (int q = 0; q < (NP + NTHR - 1) / NTHR; ++q)
                if (tid + NTHR * q < NP) reinterpret_cast<v4i*>(s.wl)[tid + NTHR * q] = cl[q];
        }
    }
    __syncthreads();
    for (int e = tid; e < TT * 16 * RS; e += NTHR) {
        const int k = e % RS;
        s.hb[0][e] = to_bf16(k < DET ? s.ob[k] : 0.f);
        s.hb[1][e] = 0;
        s.xb[e] = 0;
    }
    for (int e = tid; e < TT * 16 * HS; e += NTHR) s.h32[e] = (e % HS) < DET ? s.ob[e % HS] : 0.f;
    for (int e = tid; e < TT * 16 * ZS; e += NTHR) {
        const int k = e % ZS;
        if (k < 32) s.zA[e] = to_bf16(k < STOCH ? s.ob[DET + k] : 0.f);
        else if (k >= 32 + ACT) s.zA[e] = 0;
    }
#ifdef ICEM_RSSM_ROT   // (development: the class-A waves' blocks rotated per workgroup -- do CUs in lockstep on the same lines cost?)
    if (w < 5) recurrence_steps<true, TT>(s, __builtin_amdgcn_readfirstlane((w + wg) % 5), lane, base, n, horizon, Pg, actions, items, tile_stride, flag, ntl, stamps, stamp);
    else
#endif
    if (w < 5) recurrence_steps<true, TT>(s, w, lane, base, n, horizon, Pg, actions, items, tile_stride, flag, ntl, stamps, stamp);
    else recurrence_steps<false, TT>(s, w, lane, base, n, horizon, Pg, actions, items, tile_stride, flag, ntl, stamps, stamp);
}

// tiles first, first + stride, ...: one tile per workgroup while both workgroup kinds fit the chip together, several for
// the large populations whose reward workgroups run behind the recurrence workgroups (the head's weights are loaded once)
__device__ __forceinline__ void reward_head(ConsumerLds& s, int first, int stride, int tiles, int n, int horizon, int cost_mode,
                                            const unsigned short* __restrict__ Pg, float* __restrict__ costs,
                                            const unsigned short* stage, unsigned* flags, unsigned* status, long long* stamps) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 15, g = lane >> 4;
    const bool stamp = stamps && first == 0 && tid == 0;
    if (stamp) stamps[8] = wall_clock64();
    gptr Plane = (gptr)Pg + lane * 8;
    // the head's weights and biases, once
    v4i A6[NOB][K6K], A7[NOB][HIDK], A8[HIDK];
    v4f b6[NOB], b7[NOB];
#pragma unroll
    for (int i = 0; i < NOB; ++i) {
        const int ob = own_block(w, i);
        request<K6K>(Plane + W6 + (size_t)ob * K6K * BLK, A6[i]);
        request<HIDK>(Plane + W7 + (size_t)ob * HIDK * BLK, A7[i]);
        b6[i] = *reinterpret_cast<const v4f*>(reinterpret_cast<const float*>(Pg + B6) + ob * 16 + 4 * g);
        b7[i] = *reinterpret_cast<const v4f*>(reinterpret_cast<const float*>(Pg + B7) + ob * 16 + 4 * g);
    }
    request<HIDK>(Plane + W8, A8);
    const v4f b8 = *reinterpret_cast<const v4f*>(reinterpret_cast<const float*>(Pg + B8) + 4 * g);
    for (int e = tid; e < 2 * 16 * RS; e += NTHR) { (&s.r1[0][0])[e] = 0; (&s.r2[0][0])[e] = 0; }   // the K padding of the rows
    // A reward workgroup that gave up in an EARLIER launch left this staging area's flags in an unknown state (its stalled
    // producer kept storing behind the reset): the status word stays raised until the host has zeroed the flags
    // (launch_rssm_split), and until then every launch reports NaN instead of scoring stale states.
    if (tid == 0) s.gave_up = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u;
    __syncthreads();
    const int xr = j * RS + 8 * g, xo = j * RS + 4 * g;
    if (stamp) stamps[9] = wall_clock64();
    for (int tile = first; tile < tiles; tile += stride) {
    const unsigned short* items = stage + (size_t)tile * horizon * ITEM;
    unsigned* flag = flags + tile;
    float acc_cost = 0.f;
    // how many of the tile's states are published: read once (behind the recurrence workgroups' launch-wide lead --
    // large populations -- it already says `horizon`, and no state of the tile costs a poll of its own any more)
    if (tid == 0) s.avail[1] = (int)__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    int avail = s.avail[1];   // (slot 1: the polls of states 0, 1, .. use slots 0, 1, ..: a slot is rewritten two barriers after its last read)
    unsigned long long nlo = 0, nhi = 0;   // the next state, requested while this one is scored
    bool have_next = false;
    for (int t = 0; t < horizon; ++t) {
        const int par = t & 1;
        if (avail < t + 1) {   // (uniform) one wave polls (the flag is read past L2: eight waves of 64 workgroups polling
            // the same 256 bytes of memory are traffic the recurrence workgroups' own requests queue behind)
            if (w == 0) {
                unsigned polls = 0;
                unsigned v;
                while ((int)((v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) - (unsigned)(t + 1)) < 0) {
                    if (++polls > MAX_POLLS) {
                        if (lane == 0) {
                            s.gave_up = 1;
                            __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // host-visible
                        }
                        v = (unsigned)horizon;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(4);
                }
                if (lane == 0) s.avail[par] = (int)v;
            }
            __syncthreads();
            avail = s.avail[par];
        }
        asm volatile("" ::: "memory");
        if (stamp && tile == 0 && t == 5) stamps[10] = wall_clock64();
        if (stamp && tile == 0 && t == horizon - 1) stamps[12] = wall_clock64();
        // the item once per workgroup (the loads go past L2: eight waves fetching the same 8 KB each cost more than a
        // trip through LDS): thread e fetches 16 bytes
        {
            unsigned long long lo, hi;
            if (have_next) {
                lo = nlo; hi = nhi;
            } else {
                const unsigned short* it = items + (size_t)t * ITEM + tid * 8;
                lo = load_ag(it); hi = load_ag(it + 4);
            }
            unsigned long long* dst = reinterpret_cast<unsigned long long*>(s.x[par] + (tid >> 5) * XS + (tid & 31) * 8);
            dst[0] = lo; dst[1] = hi;
            have_next = t + 1 < horizon && avail >= t + 2;
            if (have_next) {
                const unsigned short* it = items + (size_t)(t + 1) * ITEM + tid * 8;
                nlo = load_ag(it); nhi = load_ag(it + 4);
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NOB; ++i) {
            v4f a = b6[i];
#pragma unroll
            for (int kb = 0; kb < K6K; ++kb)
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, A6[i][kb]),
                                                            __builtin_bit_cast(v8bf, *reinterpret_cast<const v4i*>(s.x[par] + j * XS + 8 * g + kb * 32)), a, 0, 0, 0);
            if (w + WAVES * i < 13) *reinterpret_cast<v4s*>(s.r1[par] + xo + own_block(w, i) * 16) = relu_pack(a);
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NOB; ++i) {
            const v4f a = mma<HIDK>(A7[i], s.r1[par] + xr, b7[i]);
            if (w + WAVES * i < 13) *reinterpret_cast<v4s*>(s.r2[par] + xo + own_block(w, i) * 16) = relu_pack(a);
        }
        __syncthreads();
        if (w == WAVES - 1) {
            const v4f a = mma<HIDK>(A8, s.r2[par] + xr, b8);
            const float c = -a[0];   // output 0 of trajectory j lives in lane (j, g = 0), register 0
            if (t == 0 || cost_mode == 2) acc_cost = c;
            else if (cost_mode == 0) acc_cost += c;
            else acc_cost = (c < acc_cost || c != c) ? c : acc_cost;
        }
        if (stamp && tile == 0 && t == 5) stamps[11] = wall_clock64();
    }
    if (stamp && tile == 0) stamps[13] = wall_clock64();
    if (tid == 0) __hip_atomic_store(flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    if (w == WAVES - 1 && g == 0 && tile * 16 + j < n)
        costs[tile * 16 + j] = s.gave_up ? __builtin_nanf("") : acc_cost;   // (a workgroup that gave up once reports NaN from there on)
    }
}

// blocks [0, prods): recurrence of TT tiles each; blocks [prods, prods + tiles): the reward head of one tile each
template <int TT>
__global__ __launch_bounds__(NTHR) void rssm_split_kernel(int n, int horizon, int cost_mode, const unsigned short* __restrict__ Pg,
                                                         const float* __restrict__ obs0, const float* __restrict__ actions,
                                                         float* __restrict__ costs, unsigned short* stage, unsigned* flags,
                                                         unsigned* status, int tiles, int prods, long long* stamps) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[lds_bytes<TT>()];
    const int b = blockIdx.x;
    if (b < prods) recurrence<TT>(*reinterpret_cast<ProducerLds<TT>*>(smem), b, tiles, n, horizon, Pg, obs0, actions, stage, flags, stamps);
    else reward_head(*reinterpret_cast<ConsumerLds*>(smem), b - prods, (int)gridDim.x - prods, tiles, n, horizon, cost_mode, Pg, costs, stage,
                     flags, status, stamps);
}

// One staging area per (device, stream): launches on a stream are ordered, so they may share it.  (Areas live until
// rssm_split_trim(): 25 MB for populations up to 4096, up to 403 MB at 65 536 rows, per stream ever used.)
struct Staging {
    std::mutex mu;                   // this area's own lock: a wedged stream holds up its own launches, nobody else's
    unsigned short* stage = nullptr;
    unsigned* flags = nullptr;
    unsigned* status = nullptr;      // pinned, device-mapped word: raised by a reward workgroup whose wait timed out
    unsigned* status_dev = nullptr;  // (its device address)
    size_t items = 0;
};
std::mutex g_mu;
long long* g_stamps = nullptr;
std::map<std::pair<int, hipStream_t>, Staging> g_staging;
}  // namespace

void rssm_set_stamps(long long* dev_ptr) { g_stamps = dev_ptr; }

// free every staging area (the caller has no launch of this path in flight)
void rssm_split_trim() {
    std::lock_guard<std::mutex> lk(g_mu);
    for (auto& kv : g_staging) {
        std::lock_guard<std::mutex> la(kv.second.mu);
        if (kv.second.stage) (void)hipFree(kv.second.stage);
        if (kv.second.flags) (void)hipFree(kv.second.flags);
        if (kv.second.status) (void)hipHostFree(kv.second.status);
    }
    g_staging.clear();
}

bool rssm_split_ok(int n, int horizon) {
    const bool on = icem::opt_i(icem::OPT_RSSM_SPLIT) != 0;
    const int max_tiles = [] {   // (development: option rssm_split_max_n moves the population limit)
        const int v = icem::opt_i(icem::OPT_RSSM_SPLIT_MAX_N) / 16;
        return v < 1 ? 1 : v > rssm::SPLIT_TILE_LIMIT ? rssm::SPLIT_TILE_LIMIT : v;
    }();
    return on && n > 0 && horizon >= 1 && (n + 15) / 16 <= max_tiles;
}

hipError_t launch_rssm_split(int n, int horizon, int cost_mode, const unsigned short* params, const float* obs0,
                             const float* actions, float* costs, hipStream_t st) {
    const int tiles = (n + 15) / 16;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    // The map's lock covers the look-up alone (nodes of a std::map do not move); the launch stays inside the AREA's lock:
    // another host thread that grows this stream's staging synchronises the stream and frees the old area, which must not
    // happen between reading the pointers and enqueuing the kernel that uses them.
    Staging* sp;
    {
        std::lock_guard<std::mutex> lk(g_mu);
        sp = &g_staging[{dev, st}];
    }
    Staging& s = *sp;
    std::lock_guard<std::mutex> lk(s.mu);
    // a capturing stream must not be synchronised (it would invalidate the capture): the two cases below that need to --
    // recovery from a timed-out wait, growing the area -- report instead
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    const bool capturing = hipStreamIsCapturing(st, &cap) == hipSuccess && cap != hipStreamCaptureStatusNone;
    (void)hipGetLastError();
    if (!s.status) {
        if ((e = hipHostMalloc((void**)&s.status, sizeof(unsigned), hipHostMallocMapped)) != hipSuccess) return e;
        *s.status = 0u;
        if ((e = hipHostGetDevicePointer((void**)&s.status_dev, s.status, 0)) != hipSuccess) return e;
    }
    if (__atomic_load_n(s.status, __ATOMIC_ACQUIRE) != 0u) {
        // a reward workgroup of an earlier launch on this stream gave up waiting: that launch returned NaN costs and its
        // flags are in an unknown state.  Quiesce, reset, and tell the caller (once) instead of launching on top of it.
        if (capturing) return hipErrorStreamCaptureUnsupported;
        (void)hipStreamSynchronize(st);
        if (s.flags) (void)hipMemsetAsync(s.flags, 0, rssm::SPLIT_TILE_LIMIT * sizeof(unsigned), st);
        (void)hipStreamSynchronize(st);
        __atomic_store_n(s.status, 0u, __ATOMIC_RELEASE);   // (behind the synchronise: no kernel is storing to it any more)
        return hipErrorLaunchTimeOut;
    }
    // (8 KB per tile and step: 25 MB cover the populations up to 4096 at h = 12; larger ones grow it, to 403 MB at 65 536)
    const size_t want = (size_t)(tiles > rssm::SPLIT_TT1_TILES ? tiles : rssm::SPLIT_TT1_TILES) * horizon;
    if (s.items < want) {   // (first call on this stream, or a longer horizon: not inside a capture)
        if (s.stage) {
            if (capturing) return hipErrorStreamCaptureUnsupported;   // (growing means synchronising: size the area with one launch outside the capture)
            if ((e = hipStreamSynchronize(st)) != hipSuccess) return e;
            (void)hipFree(s.stage);
            s.stage = nullptr; s.items = 0;
        }
        if ((e = hipMalloc(&s.stage, want * ITEM * sizeof(unsigned short))) != hipSuccess) return e;
        if ((e = hipMemsetAsync(s.stage, 0, want * ITEM * sizeof(unsigned short), st)) != hipSuccess) return e;   // K padding stays zero
        s.items = want;
    }
    if (!s.flags) {
        if ((e = hipMalloc(&s.flags, rssm::SPLIT_TILE_LIMIT * sizeof(unsigned))) != hipSuccess) return e;
        if ((e = hipMemsetAsync(s.flags, 0, rssm::SPLIT_TILE_LIMIT * sizeof(unsigned), st)) != hipSuccess) return e;
    }
    const Staging& sg = s;
    // Up to 256 tiles: one tile per recurrence workgroup and one reward workgroup per tile (up to 128 tiles both kinds are
    // resident together).  Beyond: two tiles per recurrence workgroup (they share every weight chunk) and 512 reward
    // workgroups that walk the tiles behind them.  (option rssm_split_tt = 1 | 2 overrides the tiles per workgroup.)
    const int tt_opt = icem::opt_i(icem::OPT_RSSM_SPLIT_TT);
    const int tt_env = (tt_opt == 1 || tt_opt == 2) ? tt_opt : 0;
    const int tt = tt_env ? tt_env : (tiles > rssm::SPLIT_TT1_TILES ? 2 : 1);
    const int heads = tiles > rssm::SPLIT_TT1_TILES ? std::min(tiles, 512) : tiles;
    if (tt == 2) {
        const int prods = (tiles + 1) / 2;
        hipLaunchKernelGGL(rssm_split_kernel<2>, dim3(prods + heads), dim3(NTHR), 0, st, n, horizon, cost_mode, params, obs0, actions,
                           costs, sg.stage, sg.flags, sg.status_dev, tiles, prods, g_stamps);
    } else {
        hipLaunchKernelGGL(rssm_split_kernel<1>, dim3(tiles + heads), dim3(NTHR), 0, st, n, horizon, cost_mode, params, obs0, actions, costs,
                           sg.stage, sg.flags, sg.status_dev, tiles, tiles, g_stamps);
    }
    return hipGetLastError();
}
}  // namespace icem
