// dev_transcript.cuh -- the Fiat-Shamir transcript kept ON THE DEVICE between sumcheck rounds.
//
// A sumcheck round is "reduce the partials -> round message -> absorb -> challenge -> fold by the challenge".
// With the sponge on the host every round costs a device->host->device round trip (~100 us on MI355X, more than
// the kernels of all but the first few rounds).  Here the one-workgroup finish kernel of a round also runs the
// transcript step, writes the challenge to device memory, and the next round's fold kernel reads it from there:
// the host enqueues all rounds back to back and synchronises once per sumcheck.  The host sponge
// (transcript.h) exports / imports its 25-lane state + fill, so large absorbs (tables) stay on the CPU.
//
// Semantics restated: transcripts/src/fiat_shamir/fiat_shamir_transcript.rs:22-43
//   append -> update ; sample -> finalize a CLONE, absorb the 32-byte digest ; challenge = from_le_bytes_mod_order.
// Round messages: basic sumcheck prover.rs:50-58 (two sums, big-endian), GKR sumcheck
// sumcheck_gkr_protocol.rs:41-55 (Lagrange coefficients over 0..d, little-endian).
#pragma once
#include "sumcheck_kernels.cuh"

#ifndef TS          // timing hooks, defined by tools/microbench_finish.hip only
#define TS(k)
#endif

namespace zk {

struct DevSponge {          // mirrors Keccak256's private state (transcript.h)
    uint64_t a[25];
    uint32_t fill;          // bytes absorbed into the current block, < 136
    uint32_t pad_;
};

constexpr int kFinishBlock = 1024;   // 16 waves: <= 4 partials per lane at the largest reduction grid

__device__ __constant__ const uint64_t kKeccakRC[24] = {
    0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull,
    0x000000000000808bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,
    0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,
    0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull,
    0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
    0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};

__device__ __constant__ const uint8_t kKeccakRho[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};

__device__ __forceinline__ uint64_t rotl64(uint64_t x, unsigned s) { return s ? (x << s) | (x >> (64 - s)) : x; }

// LDS exchange inside ONE wave: the LDS pipeline serves a wave's requests in order, so lanes see each other's earlier
// writes; the fences only stop the compiler from moving LDS accesses across the exchange point.
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Keccak-f[1600] by one wave, uniform control flow: lane l owns state lane (x, y) = (l % 5, l / 5) of l % 25 (lanes
// 25..63 mirror lanes 0..24 and store the same values).  Per round two LDS exchanges: the column parities of theta
// (10 reads of A) and the rho-pi scatter + chi gather (1 write, 3 reads of B).  A single lane needs ~5000 dependent
// VALU instructions per permutation (~10 us); this form ~24 x (13 LDS reads + ~40 VALU), measured in DESIGN.md.
__device__ __forceinline__ void keccak_f1600_wave(uint64_t *A, uint64_t *B, unsigned lane) {
    const unsigned l = lane % 25u, x = l % 5u, y = l / 5u;
    const unsigned xm = (x + 4u) % 5u, xp = (x + 1u) % 5u;
    const unsigned rot = kKeccakRho[l];
    const unsigned dst = y + 5u * ((2u * x + 3u * y) % 5u);
    const unsigned i1 = (x + 1u) % 5u + 5u * y, i2 = (x + 2u) % 5u + 5u * y;
    const uint64_t rc_lane = kKeccakRC[lane < 24u ? lane : 0u];   // round constant r lives in lane r (no memory access per round)
    uint64_t a = A[l];
#pragma unroll 1
    for (int round = 0; round < 24; round++) {
        if (round) {
            A[l] = a;
            wave_lds_sync();
        }
        uint64_t cm = A[xm] ^ A[xm + 5] ^ A[xm + 10] ^ A[xm + 15] ^ A[xm + 20];
        uint64_t cp = A[xp] ^ A[xp + 5] ^ A[xp + 10] ^ A[xp + 15] ^ A[xp + 20];
        a ^= cm ^ rotl64(cp, 1);                             // theta
        B[dst] = rotl64(a, rot);                             // rho, pi
        wave_lds_sync();
        uint64_t b0 = B[l], b1 = B[i1], b2 = B[i2];
        a = b0 ^ (~b1 & b2);                                 // chi
        const uint64_t rc = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)rc_lane, round) |
                            ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(rc_lane >> 32), round) << 32);
        a ^= l == 0 ? rc : 0ull;                             // iota
        wave_lds_sync();                                     // B is rewritten next round: keep the reads above it
    }
    A[l] = a;
    wave_lds_sync();
}

// One wave, uniform control flow: absorb msg[0 .. nbytes), sample (finalize a clone, append the digest to msg and absorb
// it back).  The digest words end up in msg[nbytes/4 .. nbytes/4 + 8).  `st` / `cl` / `tmp` are 25-lane LDS states,
// `fill` the running block fill (uniform).  nbytes is a multiple of 4; `fill` may be anything (the host can have absorbed
// arbitrary byte strings before): 136 and all message lengths are multiples of 4, so `fill` keeps its alignment
// for the whole call -- whole words when it is word-aligned, single bytes otherwise.
__device__ __forceinline__ void sponge_absorb_sample_wave(uint64_t *st, uint64_t *cl, uint64_t *tmp, uint32_t &fill, uint32_t *msg,
                                                          unsigned nbytes, unsigned lane) {
    uint8_t *sb = reinterpret_cast<uint8_t *>(st);
    uint32_t *sw = reinterpret_cast<uint32_t *>(st);
    const uint8_t *mb = reinterpret_cast<const uint8_t *>(msg);
    const unsigned total = nbytes + 32;
    const bool words = (fill & 3u) == 0;
    unsigned i = 0;
    bool sampled = false;
    while (true) {
        uint64_t *tgt = st;
        bool perm = false;
        if (i == nbytes && !sampled) {                       // finalize a clone: pad 0x01 .. 0x80 (Keccak, not SHA-3)
            if (lane < 25) cl[lane] = st[lane];
            wave_lds_sync();
            if (lane == 0) {
                uint8_t *cb = reinterpret_cast<uint8_t *>(cl);
                cb[fill] ^= 0x01;
                cb[135] ^= 0x80;
            }
            wave_lds_sync();
            tgt = cl;
            perm = true;
        } else if (i == total) {
            break;
        } else {                                             // as much as fits before the block boundary / end of this phase
            const unsigned end = i < nbytes ? nbytes : total;
            unsigned n = end - i;
            if (n > 136u - fill) n = 136u - fill;
            if (words) {
                if (lane < (n >> 2)) sw[(fill >> 2) + lane] ^= msg[(i >> 2) + lane];
            } else {
                for (unsigned k = lane; k < n; k += 64) sb[fill + k] ^= mb[i + k];
            }
            i += n;
            fill += n;
            if (fill == 136) { fill = 0; perm = true; }
            wave_lds_sync();
        }
        if (perm) keccak_f1600_wave(tgt, tmp, lane);
        if (tgt == cl) {                                     // digest = first 32 bytes of the squeezed clone
            const uint32_t *cw = reinterpret_cast<const uint32_t *>(cl);
            if (lane < 8) msg[(nbytes >> 2) + lane] = cw[lane];
            wave_lds_sync();
            sampled = true;
        }
    }
}

// F::from_le_bytes_mod_order(digest) in Montgomery form: the 256-bit little-endian value V times R^2 through one
// Montgomery product.  V < R = 2^(32 N) is all the product needs of its left operand (V R^2 / R + p < 2 p before the
// conditional subtraction), so the result is the canonical V R mod p without reducing V first.
template <class F> __device__ __forceinline__ Fe<F> challenge_from_digest(const uint32_t *digest_words) {
    Fe<F> v;
#pragma unroll
    for (int i = 0; i < F::N; i++) v.l[i] = i < 8 ? digest_words[i] : 0u;
    return fe_from_canonical<F>(v);
}

// ---- host-assisted transcript step: a mailbox in pinned, coherent host memory -------------------------------------------------------
// Keccak-f[1600] costs one CPU core ~0.2 us and one GPU wave ~4.7 us (24 dependent rounds of two LDS exchanges each), and a round
// of a sumcheck needs two of them plus a handful of dependent field products: ~12 us of the ~19 us a round's transcript step takes on the
// device.  A GPU wave <-> host thread round trip through fine-grained host memory measures 2.9 us (tools/microbench_mailbox.hip).  So the
// kernel that has reduced a round's evaluations writes them to the mailbox, bumps `gpu_seq` and spins (bounded) until the host thread that
// is driving the proof has run the transcript step on its sponge (zkmle_sumcheck.hip HostRounds::service) and answered with the challenge.
// The proving call polls instead of sleeping in hipStreamSynchronize; nothing else changes: same messages, same bytes absorbed.
constexpr size_t kHostMailboxBytes = 16384;              // zkmle_core.hip host_mailbox allocates this much
struct HostMailbox {
    uint64_t gpu_seq;                   // last request the GPU posted
    uint64_t pad0[15];
    uint32_t ev[(kMaxFactors + 1) * 12];                // the round's evaluations, stored (Montgomery) form
    uint32_t fin[kMaxProducts * kMaxFactors * 12];      // a tail's last post: the fully folded tables
    uint64_t pad1[8];
    uint64_t cpu_seq;                   // last request the host answered
    uint64_t pad2[15];
    uint32_t chal[12];                  // the answer: the round's challenge
    uint32_t aux[3][12];                // a layer link's answer: alpha, beta; a multi-round exchange's challenges 1..3 (basic_multi.cuh)
    uint64_t aborted;                   // set by a kernel whose spin budget ran out (the host died or stalled > ~2 s): the kernel still ends
    uint64_t pad3[7];
    // A round's answer in ONE 64-byte line, so that the poll that sees it has the challenge already (a second read over PCIe costs ~1 us
    // per round): dword 0 and dword 15 = the request number (low 32 bits), dwords 1 .. 12 = the challenge's limbs.  The host writes the
    // limbs first and the two tags last; a tag in each 32-byte half keeps the test sound even if the line were fetched as two halves.
    uint32_t ans[16];
    uint32_t ans8[8][16];               // the same for an exchange that answers with up to eight challenges (basic_multi.cuh): one line each
    // an exchange of the basic sumcheck posts up to 2^8 segment sums (basic_multi.cuh): not part of what a proof's start resets
    uint32_t big[256 * 12];
};
static_assert(offsetof(HostMailbox, ans) % 64 == 0, "the answer line must not straddle two lines");
static_assert(offsetof(HostMailbox, ans8) % 64 == 0, "the answer lines must not straddle lines");
static_assert(sizeof(HostMailbox) <= kHostMailboxBytes, "the mailbox is one pinned block (zkmle_core.hip host_mailbox)");
constexpr long long kMailboxSpinBudget = 2000000;       // polls of ~0.7-1.5 us each: 1.5-3 s

// wave 0, uniform: post `nel` elements from `src` (LDS) as request `seq`
template <class F> __device__ __forceinline__ void mailbox_post(HostMailbox *mb, uint32_t *dst, const Fe<F> *src, int nel, uint64_t seq, unsigned lane) {
    for (int k = (int)lane; k < nel * F::N; k += 64) dst[(k / F::N) * 12 + k % F::N] = src[k / F::N].l[k % F::N];
    __threadfence_system();
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) __atomic_store_n(&mb->gpu_seq, seq, __ATOMIC_RELEASE);
}
// wave 0, uniform: wait (bounded) until the host has answered request `seq`
__device__ __forceinline__ void mailbox_wait(HostMailbox *mb, uint64_t seq, unsigned lane) {
    if (lane == 0) {
        long long spins = 0;
        while (__atomic_load_n(&mb->cpu_seq, __ATOMIC_ACQUIRE) < seq) {
            // give up after the budget, and at once if an earlier wait of this proof already did: a proof whose host side is gone must
            // drain in seconds, not in (rounds x budget)
            if (++spins > kMailboxSpinBudget || __atomic_load_n(&mb->aborted, __ATOMIC_RELAXED) != 0) { mb->aborted = seq; break; }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    __builtin_amdgcn_wave_barrier();
    __threadfence_system();
}
// wave 0, uniform: wait (bounded) for the answer line of request `seq` and return the challenge it carries (mailbox_wait + one read, fused)
template <class F> __device__ __forceinline__ Fe<F> mailbox_wait_challenge(HostMailbox *mb, uint64_t seq, unsigned lane) {
    const uint32_t tag = (uint32_t)seq;
    uint32_t v = 0;
    long long spins = 0;
    for (;;) {
        if (lane < 16) v = __hip_atomic_load(&mb->ans[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const uint32_t ta = __builtin_amdgcn_readlane(v, 0), tb = __builtin_amdgcn_readlane(v, 15);
        if (ta == tag && tb == tag) break;
        int stop = 0;
        if (lane == 0 && ((++spins & 63) == 0)) {            // now and then: has the host given up on this proof, or is the budget spent
            if (spins > kMailboxSpinBudget || __atomic_load_n(&mb->aborted, __ATOMIC_RELAXED) != 0 ||
                __atomic_load_n(&mb->cpu_seq, __ATOMIC_ACQUIRE) > seq + ((uint64_t)1 << 40)) { mb->aborted = seq; stop = 1; }
        }
        if (__builtin_amdgcn_readfirstlane(stop)) break;
        __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_wave_barrier();
    Fe<F> e;
#pragma unroll
    for (int i = 0; i < F::N; i++) e.l[i] = __builtin_amdgcn_readlane(v, 1 + i);
    return e;
}
// wave 0, uniform: the same for an exchange of m <= 8 challenges, one answer line each (four lines per load of the wave): lane i < m
// returns challenge i
template <class F> __device__ __forceinline__ Fe<F> mailbox_wait_challenges(HostMailbox *mb, uint64_t seq, unsigned lane, unsigned m) {
    const uint32_t tag = (uint32_t)seq;
    uint32_t v0 = 0, v1 = 0;
    long long spins = 0;
    for (;;) {
        v0 = __hip_atomic_load(&mb->ans8[0][0] + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (m > 4) v1 = __hip_atomic_load(&mb->ans8[4][0] + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        bool ok = true;
        for (unsigned i = 0; i < m; i++) {
            const uint32_t v = i < 4 ? v0 : v1;
            ok = ok && __builtin_amdgcn_readlane(v, 16 * (i & 3u)) == tag && __builtin_amdgcn_readlane(v, 16 * (i & 3u) + 15) == tag;
        }
        if (ok) break;
        int stop = 0;
        if (lane == 0 && ((++spins & 63) == 0)) {
            if (spins > kMailboxSpinBudget || __atomic_load_n(&mb->aborted, __ATOMIC_RELAXED) != 0 ||
                __atomic_load_n(&mb->cpu_seq, __ATOMIC_ACQUIRE) > seq + ((uint64_t)1 << 40)) { mb->aborted = seq; stop = 1; }
        }
        if (__builtin_amdgcn_readfirstlane(stop)) break;
        __builtin_amdgcn_s_sleep(1);
    }
    __builtin_amdgcn_wave_barrier();
    Fe<F> e;
    const unsigned line = lane < m ? lane : 0u;
#pragma unroll
    for (int i = 0; i < F::N; i++) {
        const uint32_t a = (uint32_t)__shfl((int)v0, (int)(16 * (line & 3u) + 1 + i)), b = (uint32_t)__shfl((int)v1, (int)(16 * (line & 3u) + 1 + i));
        e.l[i] = line < 4 ? a : b;
    }
    return e;
}
template <class F> __device__ __forceinline__ Fe<F> mailbox_element(const uint32_t *src) {
    Fe<F> e;
#pragma unroll
    for (int i = 0; i < F::N; i++) e.l[i] = __builtin_nontemporal_load(src + i);
    return e;
}

constexpr int kMaxPts = kMaxFactors + 1;

// One wave, uniform: the challenge every lane holds, as the uniform multiplier of the fused round that folds by it (ufield.cuh UniMul: row i = r 2^(29 (i + 2))
// mod p, one product, taken by lane i).  Costs the wave one product's latency once per round; every wave of the next launch then reads 81 words with
// scalar loads instead of working the rows out itself (sumcheck_kernels.cuh unimul_load).
template <class F> __device__ __forceinline__ void challenge_expand(uint32_t *exp_out, const Fe<F> &r, unsigned lane) {
    if constexpr (LazyProducts<F>::value) {
        if (!exp_out) return;
        constexpr int L = UParams<F>::L;
        const Ufe<F> row = unimul_row<F>(r, lane < (unsigned)L ? (int)lane : 0);
        if (lane < (unsigned)L) {
#pragma unroll
            for (int j = 0; j < L; j++) exp_out[lane * L + j] = row.l[j];
        }
    }
}
// The round's exchange in the producer's last workgroup (sumcheck_kernels.cuh RoundFin): wave 0 of every workgroup, lane t < npts holding
// the workgroup's sum of evaluation t.
template <class F> __device__ __forceinline__ void round_finish_in_producer(const RoundFin &f, const Fe<F> &tot) {
    const unsigned lane = threadIdx.x & 63u;
    const bool mine = (int)lane < f.npts && !(f.skip1 && lane == 1u);
    unsigned long long *acc = reinterpret_cast<unsigned long long *>(f.acc) + (size_t)lane * F::N * kMultiAccStride;
    if (mine) {
#pragma unroll
        for (int k = 0; k < F::N; k++)
            __hip_atomic_fetch_add(acc + (size_t)k * kMultiAccStride, (unsigned long long)tot.l[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    unsigned last = 0;
    if (lane == 0) {
        const unsigned g = blockIdx.x / f.group, ngroups = (gridDim.x + f.group - 1) / f.group;
        const unsigned members = (g + 1) * f.group <= gridDim.x ? f.group : gridDim.x - g * f.group;
        unsigned *gc = f.counter + 16u * (1u + g);
        if (__hip_atomic_fetch_add(gc, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == members - 1u) {
            __hip_atomic_store(gc, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (__hip_atomic_fetch_add(f.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == ngroups - 1u) {
                __hip_atomic_store(f.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                last = 1u;
            }
        }
    }
    if (!__builtin_amdgcn_readfirstlane(last)) return;
    Fe<F> e = fe_zero<F>();
    unsigned long long v[F::N + 1];
#pragma unroll
    for (int k = 0; k <= F::N; k++) v[k] = 0;
    if (mine) {
#pragma unroll
        for (int k = 0; k < F::N; k++) v[k] = __hip_atomic_load(acc + (size_t)k * kMultiAccStride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int k = 0; k < F::N; k++) __hip_atomic_store(acc + (size_t)k * kMultiAccStride, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (f.limbs_out) {                                       // sums of 32-bit limbs, carries unpropagated: what the all-reduce adds up
        if ((int)lane < f.npts) {
#pragma unroll
            for (int k = 0; k <= F::N; k++) f.limbs_out[(size_t)lane * (F::N + 1) + k] = v[k];
        }
        return;
    }
    if (mine) {
        Wide<F> w;
        unsigned long long c = 0;
#pragma unroll
        for (int k = 0; k <= F::N; k++) {
            const unsigned long long x = v[k] + c;
            w.l[k] = (uint32_t)x;
            c = x >> 32;
        }
        e = wide_reduce<F>(w);
    }
    if (f.per2) {                                            // two rounds: nine sums out, two challenges back (zkmle_sumcheck.hip serve_round2)
        if ((int)lane < f.npts) {
#pragma unroll
            for (int k = 0; k < F::N; k++) f.mb->big[lane * 12 + k] = e.l[k];
        }
        __threadfence_system();
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) __atomic_store_n(&f.mb->gpu_seq, f.seq, __ATOMIC_RELEASE);
        const Fe<F> rr = mailbox_wait_challenges<F>(f.mb, f.seq, lane, 2u);
        if (lane < 2) fe_store<F>(f.proof, f.chal_slot + f.per2 * lane, rr);
        return;
    }
    if ((int)lane < f.npts) {                                // the evaluations, as mailbox_post lays them out
#pragma unroll
        for (int k = 0; k < F::N; k++) f.mb->ev[lane * 12 + k] = e.l[k];
    }
    __threadfence_system();
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) __atomic_store_n(&f.mb->gpu_seq, f.seq, __ATOMIC_RELEASE);
    const Fe<F> r = mailbox_wait_challenge<F>(f.mb, f.seq, lane);
    if (lane == 0) fe_store<F>(f.proof, f.chal_slot, r);
    challenge_expand<F>(f.exp_out, r, lane);
}

// what a round's transcript step needs besides the evaluations
struct RoundCtx {
    int npts;                // evaluations per round (2 = basic sumcheck halves, d + 1 for the GKR sumcheck)
    int mode;                // 0: message = the evaluations, big-endian (prover.rs:50-55)
                             // 1: message = Lagrange coefficients over 0..d, little-endian (sumcheck_gkr_protocol.rs:46-52)
    DevSponge *sponge;
    const void *basis;       // mode 1: basis[i * npts + d] = coefficient d of l_i (Montgomery form), followed by the same
                             // npts^2 coefficients as canonical integers (ev * canonical = canonical product)
    void *proof;             // Fe slots
    HostMailbox *mb;         // non-null: the transcript step runs on the host (mailbox above); sponge and basis are then unused
};
// The evaluation at 1 comes for free: the round polynomial the prover sent last satisfies p_k(r_k) = e_{k+1}(0) + e_{k+1}(1)
// identically (both sides are the sum over the folded table), whatever sum was claimed from outside.  A producer that skips
// the products of the point 1 (fold_round_evals_kernel, `skip1`) leaves ev[1] to be derived as p_k(r_k) - ev[0]; p_k(r_k) is
// re-evaluated from the previous round's proof slots by a spare wave while the other waves reduce the partials, so it costs
// no time (d products), against one product per pair index saved in the producer.  Same field elements as evaluating the
// point 1 directly.
enum { kDerive1 = 1 };
// previous round's message evaluated at its challenge (one lane)
template <class F> __device__ __forceinline__ Fe<F> previous_claim(const RoundCtx &c, size_t prev_msg_slot, size_t prev_chal_slot) {
    const Fe<F> r = fe_load<F>(c.proof, prev_chal_slot);
    if (c.mode == 0) {
        Fe<F> e0 = fe_load<F>(c.proof, prev_msg_slot), e1 = fe_load<F>(c.proof, prev_msg_slot + 1);
        return fe_add<F>(e0, fe_mul<F>(r, fe_sub<F>(e1, e0)));
    }
    Fe<F> nc = fe_load<F>(c.proof, prev_msg_slot + c.npts - 1);
#pragma unroll 1
    for (int d = c.npts - 2; d >= 0; d--) nc = fe_add<F>(fe_mul<F>(nc, r), fe_load<F>(c.proof, prev_msg_slot + d));
    return nc;
}

template <class F> struct RoundShared {                      // LDS of the transcript step
    Fe<F> ev[kMaxPts];
    Fe<F> pr[2 * kMaxPts * kMaxPts];
    Fe<F> claim;
    Fe<F> chal;
    uint32_t msg[(kMaxPts + 1) * F::N + 8];
    uint64_t st[25], cl[25], tmp[25];
};

// Wave 0, uniform control flow.  In: S.ev[0 .. npts) and the sponge in S.st / fill.  Forms the round message, stores it in
// the proof slots, absorbs it, samples; the challenge goes to proof[chal_slot] and to S.chal.
template <class F>
__device__ __forceinline__ void round_message_and_challenge(RoundShared<F> &S, const RoundCtx &c, int with_claim, size_t claim_slot,
                                                            size_t msg_slot, size_t chal_slot, uint32_t &fill, unsigned lane, int flags = 0) {
    const int npts = c.npts;
    const int nmsg = npts + (with_claim ? 1 : 0);
    if (flags & kDerive1) {                                  // S.claim was left by the helper wave / lane (see kDerive1)
        if (lane == 0) S.ev[1] = fe_sub<F>(S.claim, S.ev[0]);
        wave_lds_sync();
    }
    if (c.mode == 0) {
        if ((int)lane < nmsg) {
            const int t = (int)lane;
            Fe<F> m;
            size_t slot;
            if (with_claim && t == 0) { m = fe_add<F>(S.ev[0], S.ev[1]); slot = claim_slot; }      // prover.rs:28,40-41
            else { int k = t - (with_claim ? 1 : 0); m = S.ev[k]; slot = msg_slot + k; }
            fe_store<F>(c.proof, slot, m);
            Fe<F> cn = fe_to_canonical<F>(m);
#pragma unroll
            for (int k = 0; k < F::N; k++) S.msg[t * F::N + k] = __builtin_bswap32(cn.l[F::N - 1 - k]);
        }
    } else {
        // coefficient d = sum_i ev[i] * basis[i][d]: the npts^2 products in the stored form and the npts^2 canonical ones
        // (for the bytes) are taken by 2 npts^2 lanes at once, then npts + npts lanes add them up
        // with_claim (mode 1): the sumcheck's claimed sum, already in proof[claim_slot], is absorbed first, big-endian
        // (sumcheck_gkr_protocol.rs:35) -- one message = the two appends; its canonical form rides along as one more product
        const int n2 = npts * npts;
        const int off = with_claim ? F::N : 0;
        if ((int)lane < 2 * n2) S.pr[lane] = fe_mul<F>(S.ev[((int)lane % n2) / npts], fe_load<F>(c.basis, lane));
        else if (with_claim && (int)lane == 2 * n2) {
            Fe<F> one = fe_zero<F>();
            one.l[0] = 1;
            const Fe<F> cn = fe_mul<F>(fe_load<F>(c.proof, claim_slot), one);            // into_bigint()
#pragma unroll
            for (int k = 0; k < F::N; k++) S.msg[k] = __builtin_bswap32(cn.l[F::N - 1 - k]);
        }
        wave_lds_sync();
        if ((int)lane < 2 * npts) {
            const int which = (int)lane / npts, d = (int)lane % npts;
            Fe<F> m = S.pr[which * n2 + d];
#pragma unroll 1
            for (int i = 1; i < npts; i++) m = fe_add<F>(m, S.pr[which * n2 + i * npts + d]);
            if (which == 0) fe_store<F>(c.proof, msg_slot + d, m);
            else {
#pragma unroll
                for (int k = 0; k < F::N; k++) S.msg[off + d * F::N + k] = m.l[k];
            }
        }
    }
    wave_lds_sync();
    TS(2);
    const unsigned nbytes = (unsigned)nmsg * 4u * F::N;
    sponge_absorb_sample_wave(S.st, S.cl, S.tmp, fill, S.msg, nbytes, lane);
    TS(3);
    if (lane == 0) {
        Fe<F> ch = challenge_from_digest<F>(S.msg + (nbytes >> 2));
        fe_store<F>(c.proof, chal_slot, ch);
        S.chal = ch;
    }
    wave_lds_sync();
}

struct FinishArgs {
    const void *partials;    // [t * count + block], t < npts
    size_t count;
    RoundCtx ctx;
    int with_claim;          // mode 0, round 0: absorb evals[0] + evals[1] first (prover.rs:28,40-41)
    int flags;               // kDerive1: the block carries one extra wave that evaluates the previous round's message
    size_t claim_slot, msg_slot, chal_slot;
    size_t prev_msg_slot, prev_chal_slot;   // kDerive1
    uint64_t seq;            // host-assisted step: the request number of this round
};

// One workgroup (64..1024 lanes, a multiple of 64).  Stage 1: every wave reduces its share of the partials, all npts
// sums interleaved.  Everything after the single __syncthreads runs in wave 0: the cross-wave sums (16-lane
// groups), the round message, the transcript step and the challenge.
template <class F>
__global__ void __launch_bounds__(kFinishBlock) sumcheck_finish_kernel(FinishArgs a) {
    __shared__ Wide<F> sh[kMaxPts * 16];                     // [t * 16 + wave]
    __shared__ RoundShared<F> S;
    const unsigned tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    // with kDerive1 the last wave is the helper: it evaluates the previous round's message at its challenge meanwhile
    const unsigned nwaves = (blockDim.x >> 6) - ((a.flags & kDerive1) ? 1u : 0u), nred = nwaves << 6;
    const int npts = a.ctx.npts;
    TS(0);
    if (wave == nwaves) {                                    // helper wave (only present with kDerive1)
        if (lane == 0) S.claim = previous_claim<F>(a.ctx, a.prev_msg_slot, a.prev_chal_slot);
    } else {
        Wide<F> acc[kMaxPts];                                // lazy sums (mle_kernels.cuh); unused ones stay zero
#pragma unroll
        for (int t = 0; t < kMaxPts; t++) {
            acc[t] = wide_zero<F>();
            if (t < npts)
                for (size_t i = tid; i < a.count; i += nred) wide_add_fe<F>(acc[t], fe_load<F>(a.partials, (size_t)t * a.count + i));
        }
        wave_reduce_wide<F, kMaxPts>(acc, npts);
        if (lane == 63) {
#pragma unroll
            for (int t = 0; t < kMaxPts; t++) sh[t * 16 + wave] = acc[t];
        }
    }
    if (tid < 25 && !a.ctx.mb) S.st[tid] = a.ctx.sponge->a[tid];
    __syncthreads();
    if (wave != 0) return;
    {   // cross-wave sums: lane t * 16 + w holds wave w's sum of evaluation t
        const unsigned t = lane >> 4, w = lane & 15u;
        Wide<F> v = w < nwaves ? sh[lane] : wide_zero<F>();
        row_reduce_wide<F>(v);                               // a 16-lane group is one DPP row: sum in its lane 15
        if (w == 15 && (int)t < npts) S.ev[t] = wide_reduce<F>(v);
        wave_lds_sync();
    }
    TS(1);
    if (a.ctx.mb) {                                          // the host runs the transcript step; the challenge comes back through the mailbox
        mailbox_post<F>(a.ctx.mb, a.ctx.mb->ev, S.ev, npts, a.seq, lane);
        const Fe<F> r = mailbox_wait_challenge<F>(a.ctx.mb, a.seq, lane);
        if (lane == 0) fe_store<F>(a.ctx.proof, a.chal_slot, r);
        return;
    }
    uint32_t fill = a.ctx.sponge->fill;
    round_message_and_challenge<F>(S, a.ctx, a.with_claim, a.claim_slot, a.msg_slot, a.chal_slot, fill, lane, a.flags);
    if (lane < 25) a.ctx.sponge->a[lane] = S.st[lane];
    if (lane == 0) a.ctx.sponge->fill = fill;
    TS(4);
}

// ---- sharded provers (one process per GPU): the transcript step split around the exchange --------------------------------
// A rank reduces its partials to npts lazy sums and publishes them as (N + 1) 32-bit limbs spread over 64-bit words
// (`limbs[t * (N + 1) + k]`): the sum over <= 2^32 ranks of such vectors is a plain element-wise integer sum, i.e. ONE
// all-reduce(SUM, int64) over RCCL, with no carries to move between words.  Every rank then runs the same transcript step on
// the summed limbs (carry, reduce mod p, message, absorb, challenge): replicated sponge, no broadcast, no host round trip.
// (r3: the words are written by the round kernel's last workgroup, round_finish_in_producer with `limbs_out`; r1-r2 ran a one-workgroup
// partials_to_limbs_kernel behind every round kernel for this.)

struct LimbsFinishArgs {
    const uint64_t *limbs;   // element-wise sums over the ranks of the round kernels' limbs_out
    RoundCtx ctx;
    int with_claim;
    int flags;               // kDerive1: launched with 128 lanes, the second wave evaluates the previous round's message
    size_t claim_slot, msg_slot, chal_slot;
    size_t prev_msg_slot, prev_chal_slot;
    uint64_t seq;            // host-assisted step: the request number of this round
};
template <class F> __global__ void __launch_bounds__(128) limbs_finish_kernel(LimbsFinishArgs a) {
    __shared__ RoundShared<F> S;
    const unsigned lane = threadIdx.x & 63u;
    if (threadIdx.x >= 64) {
        if (lane == 0 && !a.ctx.mb) S.claim = previous_claim<F>(a.ctx, a.prev_msg_slot, a.prev_chal_slot);
    } else if (lane < 25 && !a.ctx.mb) S.st[lane] = a.ctx.sponge->a[lane];
    if (threadIdx.x < 64 && (int)lane < a.ctx.npts) {
        Wide<F> w;
        uint64_t c = 0;
#pragma unroll
        for (int k = 0; k <= F::N; k++) {                    // words hold sums of 32-bit limbs: propagate the carries
            uint64_t v = a.limbs[lane * (F::N + 1) + k] + c;
            w.l[k] = (uint32_t)v;
            c = v >> 32;
        }
        S.ev[lane] = wide_reduce<F>(w);
    }
    __syncthreads();
    if (threadIdx.x >= 64) return;
    if (a.ctx.mb) {                                          // every rank's host runs the same transcript step on the summed evaluations
        mailbox_post<F>(a.ctx.mb, a.ctx.mb->ev, S.ev, a.ctx.npts, a.seq, lane);
        const Fe<F> r = mailbox_wait_challenge<F>(a.ctx.mb, a.seq, lane);
        if (lane == 0) fe_store<F>(a.ctx.proof, a.chal_slot, r);
        return;
    }
    uint32_t fill = a.ctx.sponge->fill;
    round_message_and_challenge<F>(S, a.ctx, a.with_claim, a.claim_slot, a.msg_slot, a.chal_slot, fill, lane, a.flags);
    if (lane < 25) a.ctx.sponge->a[lane] = S.st[lane];
    if (lane == 0) a.ctx.sponge->fill = fill;
}

// ---- between two layers of a GKR proof (gkr_protocol.rs:109-133), on the device ---------------------------------------------------
// wb = W(rb*) and wc = W(rc*) are final values of the layer's sumcheck (proof slots); append wb, sample alpha, append wc, sample beta,
// next claim = alpha wb + beta wc.  One wave; the sponge stays on the device, so a whole GKR proof is one stream of kernels.
struct LinkArgs {
    DevSponge *sponge;
    void *proof;
    size_t wb_src, wc_src;                   // slots holding W(rb*), W(rc*)
    size_t wb_slot, wc_slot, alpha_slot, beta_slot, claim_slot;
};
template <class F> __global__ void __launch_bounds__(64) gkr_link_kernel(LinkArgs a) {
    __shared__ RoundShared<F> S;
    const unsigned lane = threadIdx.x;
    if (lane < 25) S.st[lane] = a.sponge->a[lane];
    uint32_t fill = a.sponge->fill;
    wave_lds_sync();
    const Fe<F> wb = fe_load<F>(a.proof, a.wb_src), wc = fe_load<F>(a.proof, a.wc_src);
    Fe<F> ch[2];
#pragma unroll 1
    for (int k = 0; k < 2; k++) {
        if (lane == 0) {
            const Fe<F> cn = fe_to_canonical<F>(k == 0 ? wb : wc);                       // field_element_to_bytes: big-endian (:125,:128)
#pragma unroll
            for (int i = 0; i < F::N; i++) S.msg[i] = __builtin_bswap32(cn.l[F::N - 1 - i]);
        }
        wave_lds_sync();
        sponge_absorb_sample_wave(S.st, S.cl, S.tmp, fill, S.msg, 4u * F::N, lane);
        ch[k] = challenge_from_digest<F>(S.msg + F::N);
        wave_lds_sync();
    }
    if (lane == 0) {
        fe_store<F>(a.proof, a.wb_slot, wb);
        fe_store<F>(a.proof, a.wc_slot, wc);
        fe_store<F>(a.proof, a.alpha_slot, ch[0]);
        fe_store<F>(a.proof, a.beta_slot, ch[1]);
        fe_store<F>(a.proof, a.claim_slot, fe_add<F>(fe_mul<F>(ch[0], wb), fe_mul<F>(ch[1], wc)));   // :132
    }
    if (lane < 25) a.sponge->a[lane] = S.st[lane];
    if (lane == 0) a.sponge->fill = fill;
}

// host-assisted layer link: the host already has wb and wc (the tails' final values) and answers with alpha and beta
struct LinkWaitArgs {
    HostMailbox *mb;
    void *proof;
    uint64_t seq;
    size_t alpha_slot, beta_slot;
};
template <class F> __global__ void __launch_bounds__(64) gkr_link_wait_kernel(LinkWaitArgs a) {
    const unsigned lane = threadIdx.x;
    mailbox_wait(a.mb, a.seq, lane);
    if (lane == 0) {
        fe_store<F>(a.proof, a.alpha_slot, mailbox_element<F>(a.mb->aux[0]));
        fe_store<F>(a.proof, a.beta_slot, mailbox_element<F>(a.mb->aux[1]));
    }
}

// ---- tail of a sumcheck: every round from a table of <= 4 kTailBlock entries down to one entry in ONE launch ----------
// Below ~2^11 entries a round is pure latency (one lane's chain of ~16 dependent products + the transcript step); as two
// launches per round it costs ~36 us, most of it launch, partial-sum round trip and a second reduction.  One workgroup keeps
// the sponge in LDS, folds + evaluates (lane i owns pair index i), reduces straight to the evaluations, runs the
// transcript step in wave 0 and broadcasts the challenge through LDS.  Same arithmetic, same bytes absorbed.
constexpr int kTailBlock = 512;
constexpr size_t kTailLen = 4 * (size_t)kTailBlock;
constexpr int kTailSplit = 512;      // (pair index, table) lanes of a split round: 2 x 32 B each in LDS

struct TailArgs {
    SumPolyTables tabs;      // in[k]: current tables of `len` entries
    void *buf[2];            // ping-pong storage: round j writes table k at buf[j & 1] + k * (out length) elements
    int nprod, ntab;
    size_t len;              // 2 .. 4 * kTailBlock (kTailLen), a power of two
    RoundCtx ctx;
    size_t round;            // index of the round whose challenge folds `in` (already in proof[chal_base + per * round])
    size_t msg_base, chal_base, per;   // slots of round k: messages at msg_base + per k, challenge at chal_base + per k
    size_t fin_slot;         // ntab final values (only written when fin_slot != ~0)
    uint64_t seq0;           // host-assisted step: request number of the tail's first round; the final values go out as one more request
    uint64_t *trace;         // ZK_TAIL_TRACE=1: 6 wall_clock64() stamps per round (measurement only), else nullptr
    // first_evals: `in` are the tables of round `round` BEFORE any exchange -- the launch starts with that round's evaluations (what round_evals_kernel
    // + its exchange do for a long table: 15-17 us as a launch of its own, 16 of them in a depth-8 GKR proof) and goes on as usual.  with_claim /
    // claim_slot: proof[claim_slot] is absorbed in front of that first message (sumcheck_gkr_protocol.rs:35).
    int first_evals, with_claim;
    size_t claim_slot;
    // pending2: TWO challenges are pending at entry (rounds `round` and `round + 1`, after a two-round exchange of split2_round_kernel): the tables are
    // folded by the first one before anything else
    int pending2;
    // two_rounds (host-assisted step, two-factor products; 0 = never): while the tables have at most this many (product, quad) pairs, TWO rounds per
    // exchange (see the kernel)
    int two_rounds;
};
#define ZK_TAIL_STAMP(k) do { if (a.trace && tid == 0) a.trace[6 * j + (k)] = wall_clock64(); } while (0)

template <class F, int NFAC>
__global__ void __launch_bounds__(kTailBlock) sumcheck_tail_kernel(TailArgs a) {
    __shared__ Wide<F> sh[(NFAC + 1) * kTailBlock / 64];
    __shared__ RoundShared<F> S;
    const unsigned tid = threadIdx.x, lane = tid & 63u;
    const size_t esz = 4 * F::N;
    HostMailbox *const mb = a.ctx.mb;
    if (tid < 25 && !mb) S.st[tid] = a.ctx.sponge->a[tid];
    uint32_t fill = mb ? 0u : a.ctx.sponge->fill;
    uint64_t seq = a.seq0;
    Fe<F> r = fe_zero<F>();
    // table k of the current round: the caller's tables first, then slice k of the previous round's output buffer
    const char *prev = nullptr;
    size_t cl = a.len, round = a.round;
    int j = 0;
    __syncthreads();
    if (a.first_evals) {                                     // round `round` itself: evaluations of the tables as they are, exchange, challenge
        const size_t half = cl / 2;
        Wide<F> acc[NFAC + 1];
#pragma unroll
        for (int t = 0; t <= NFAC; t++) acc[t] = wide_zero<F>();
        for (size_t i = tid; i < half; i += kTailBlock) {
            for (int p = 0; p < a.nprod; p++) {
                Fe<F> lo[NFAC], hi[NFAC];
#pragma unroll
                for (int f = 0; f < NFAC; f++) {
                    if (NFAC == 2 && f == 1 && a.tabs.in[p * NFAC + 1] == nullptr) { lo[1] = hi[1] = const_factor<F>(a.tabs, p); continue; }
                    lo[f] = fe_load<F>(a.tabs.in[p * NFAC + f], i);
                    hi[f] = fe_load<F>(a.tabs.in[p * NFAC + f], i + half);
                }
                accumulate_terms<F, NFAC>(lo, hi, acc);
            }
        }
        Fe<F> tot;
        const size_t lanes0 = half < (size_t)kTailBlock ? half : (size_t)kTailBlock;
        if (block_reduce_wide<F, NFAC + 1>(acc, sh, tot, (int)((lanes0 + 63) / 64))) S.ev[tid] = tot;
        __syncthreads();
        if (tid < 64) {
            if (mb) {
                mailbox_post<F>(mb, mb->ev, S.ev, NFAC + 1, seq, lane);
                const Fe<F> rn = mailbox_wait_challenge<F>(mb, seq, lane);
                if (lane == 0) {
                    S.chal = rn;
                    fe_store<F>(a.ctx.proof, a.chal_base + a.per * round, S.chal);
                }
                seq++;
            } else {
                round_message_and_challenge<F>(S, a.ctx, a.with_claim, a.claim_slot, a.msg_base + a.per * round, a.chal_base + a.per * round, fill, lane);
            }
        }
        __syncthreads();
        r = S.chal;
    } else {
        r = fe_load<F>(a.ctx.proof, a.chal_base + a.per * a.round);
    }
    if (a.pending2) {                                        // fold by the first pending challenge; the second one is the `r` of what follows
        const size_t ol = cl / 2;
        char *d1 = (char *)a.buf[j & 1];
        const Multiplier<F> mr(r);
        for (size_t t = tid; t < (size_t)a.ntab * ol; t += kTailBlock) {
            const size_t k = t / ol, idx = t - k * ol;
            if (a.tabs.in[k] == nullptr) continue;
            const Fe<F> x = fe_load<F>(a.tabs.in[k], idx), y = fe_load<F>(a.tabs.in[k], idx + ol);
            fe_store<F>(d1 + k * ol * esz, idx, fe_add<F>(x, mr.times(fe_sub<F>(y, x))));
        }
        __syncthreads();
        prev = d1;
        cl = ol;
        round++;
        j++;
        r = fe_load<F>(a.ctx.proof, a.chal_base + a.per * round);
    }
    // Two rounds per exchange.  A round on a short table is latency: ~3 us of arithmetic, then the workgroup's reduction (2 us) and the exchange with the
    // host (3.4 us) -- and the round AFTER it is a polynomial in this round's challenge whose coefficients are known before the challenge is.  With the
    // folded tables T' (4 q' entries each) cut into quads a = T'[i], b = T'[i + q'], c = T'[i + 2 q'], d = T'[i + 3 q'] (first variable: a,b | c,d;
    // second: a | b), per product of two factors:
    //   round A (first variable):  e(0) = P0 + P1, e(1) = Q0 + Q1, e(inf) = D0 + D1,   P = a a', b b';  Q = c c', d d';  D = (c - a)(c' - a'), (d - b)(d' - b')
    //   round B after folding by rA: lo = a + rA (c - a), hi = b + rA (d - b):
    //     e(0) = lo lo' = P0 + rA (Q0 - P0 - D0) + rA^2 D0,   e(1) = P1 + rA (Q1 - P1 - D1) + rA^2 D1,
    //     e(inf) = (hi - lo)(hi' - lo') = EE + rA (FF - EE - GG) + rA^2 GG,   E = b - a, F = d - c, G = F - E.
    // Nine sums -- as many products as the two rounds take one after the other -- leave in ONE post; the host runs both transcript steps
    // (zkmle_sumcheck.hip serve_round2: the same messages, the same bytes absorbed) and answers with both challenges.  One WAVE per sum: lane m of
    // wave s takes (product, quad) m of sum s -- one product per lane, one DPP reduction per wave; wave 0 takes the ninth sum as well.
    // (A first form with one lane per (product, quad) doing all nine products lost to the single rounds: 256 VGPRs, spills, a nine-sum reduction.)
    __shared__ Wide<F> w9[9];                                // two rounds per exchange (below)
    __shared__ Fe<F> ch2[2];
    const unsigned wv = tid >> 6;
    for (;;) {
        if constexpr (NFAC == 2) {
            while (mb && a.two_rounds && cl >= 8 && (size_t)a.nprod * (cl / 8) <= (size_t)a.two_rounds) {
                const size_t ol = cl / 2, qq = ol / 4, o2 = ol / 2;
                char *d1 = (char *)a.buf[j & 1], *d2 = (char *)a.buf[(j + 1) & 1];
                ZK_TAIL_STAMP(0);
                {   // fold by r: one lane task per (table, output index)
                    const Multiplier<F> mr(r);
                    for (size_t t = tid; t < (size_t)a.ntab * ol; t += kTailBlock) {
                        const size_t k = t / ol, idx = t - k * ol;
                        if (a.tabs.in[k] == nullptr) continue;
                        const void *src = prev ? (const void *)(prev + k * cl * esz) : a.tabs.in[k];
                        const Fe<F> x = fe_load<F>(src, idx), y = fe_load<F>(src, idx + ol);
                        fe_store<F>(d1 + k * ol * esz, idx, fe_add<F>(x, mr.times(fe_sub<F>(y, x))));
                    }
                }
                __syncthreads();
                ZK_TAIL_STAMP(1);
                {
                    const size_t M = (size_t)a.nprod * qq;                 // (product, quad) pairs: <= kTwoRoundPairs
                    // the operand pair (first factor, second factor) of sum `kind` at (product, quad) m
                    // kinds: 0 P0 = a a', 1 P1 = b b', 2 Q0 = c c', 3 Q1 = d d', 4 D0 = (c-a)(..), 5 D1 = (d-b)(..), 6 EE = (b-a)(..), 7 FF = (d-c)(..), 8 GG = ((d-c)-(b-a))(..)
                    auto term = [&](unsigned kind, size_t m) -> Fe<F> {
                        const size_t p = m / qq, i = m - p * qq;
                        Fe<F> o[2];
#pragma unroll
                        for (int f = 0; f < 2; f++) {
                            if (a.tabs.in[p * 2 + f] == nullptr) {                         // a constant factor: a = b = c = d
                                o[f] = kind < 4 ? const_factor<F>(a.tabs, (int)p) : fe_zero<F>();
                                continue;
                            }
                            const void *tb = d1 + (p * 2 + f) * ol * esz;
                            if (kind < 4) {
                                o[f] = fe_load<F>(tb, i + kind * qq);
                            } else if (kind < 8) {
                                const size_t hi = kind == 4 ? 2 : kind == 5 ? 3 : kind == 6 ? 1 : 3, lo = kind == 4 ? 0 : kind == 5 ? 1 : kind == 6 ? 0 : 2;
                                o[f] = fe_sub<F>(fe_load<F>(tb, i + hi * qq), fe_load<F>(tb, i + lo * qq));
                            } else {
                                o[f] = fe_sub<F>(fe_sub<F>(fe_load<F>(tb, i + 3 * qq), fe_load<F>(tb, i + 2 * qq)), fe_sub<F>(fe_load<F>(tb, i + qq), fe_load<F>(tb, i)));
                            }
                        }
                        return fe_mul<F>(o[0], o[1]);
                    };
                    if (M <= 32) {
                        // short tables: TWO sums per wave, one in each half (lanes 0..31 / 32..63), nine sums in five waves and ONE pass; the half-wave sums are
                        // what the row reduction and the first cross-row step leave in lanes 31 and 63
                        const unsigned kind = 2 * wv + (lane >> 5), m = lane & 31u;
                        if (wv < 5) {
                            Wide<F> acc = wide_zero<F>();
                            if (kind < 9 && m < M) wide_add_fe<F>(acc, term(kind, m));
                            row_reduce_wide<F>(acc);
                            wide_dpp_step<F, 0x142, 0xa>(acc);
                            if ((lane & 31u) == 31u && kind < 9) w9[kind] = acc;
                        }
                    } else {
#pragma unroll 1
                        for (unsigned pass = 0; pass < 2; pass++) {        // every wave its own sum; wave 0 the ninth one after that
                            if (pass == 1 && wv != 0) break;
                            const unsigned kind = pass == 0 ? wv : 8u;
                            Wide<F> acc[1] = {wide_zero<F>()};
#pragma unroll 1
                            for (size_t m = lane; m < M; m += 64) wide_add_fe<F>(acc[0], term(kind, m));
                            wave_reduce_wide<F, 1>(acc);
                            if (lane == 63) w9[kind] = acc[0];             // the nine lazy sums are reduced together by the posting lanes
                        }
                    }
                }
                __syncthreads();
                ZK_TAIL_STAMP(2);
                if (tid < 64) {
                    if (lane < 9) {
                        const Fe<F> e9 = wide_reduce<F>(w9[lane]);
    #pragma unroll
                        for (int w = 0; w < F::N; w++) mb->big[lane * 12 + w] = e9.l[w];
                    }
                    __threadfence_system();
                    __builtin_amdgcn_wave_barrier();
                    if (lane == 0) __atomic_store_n(&mb->gpu_seq, seq, __ATOMIC_RELEASE);
                    ZK_TAIL_STAMP(3);
                    const Fe<F> rr = mailbox_wait_challenges<F>(mb, seq, lane, 2u);
                    if (lane < 2) {
                        ch2[lane] = rr;
                        fe_store<F>(a.ctx.proof, a.chal_base + a.per * (round + 1 + lane), rr);
                    }
                    seq++;
                }
                __syncthreads();
                ZK_TAIL_STAMP(4);
                {   // fold T' by the first of the two challenges; the second one folds the result in the next iteration
                    const Multiplier<F> mr(ch2[0]);
                    for (size_t t = tid; t < (size_t)a.ntab * o2; t += kTailBlock) {
                        const size_t k = t / o2, idx = t - k * o2;
                        if (a.tabs.in[k] == nullptr) continue;
                        const void *src = d1 + k * ol * esz;
                        const Fe<F> x = fe_load<F>(src, idx), y = fe_load<F>(src, idx + o2);
                        fe_store<F>(d2 + k * o2 * esz, idx, fe_add<F>(x, mr.times(fe_sub<F>(y, x))));
                    }
                }
                __syncthreads();
                ZK_TAIL_STAMP(5);
                r = ch2[1];
                prev = d2;
                cl = o2;
                round += 2;
                j += 2;
            }
        }
        if (cl < 4) break;
        // fold by r AND evaluate the next round (sumcheck_kernels.cuh)
        const size_t q = cl / 4, ol = cl / 2;
        char *dst = (char *)a.buf[j & 1];
        ZK_TAIL_STAMP(0);
        Wide<F> acc[NFAC + 1];
#pragma unroll
        for (int t = 0; t <= NFAC; t++) acc[t] = wide_zero<F>();
        // A lane's share of a round is a chain of products (14 for two products of two factors) and the round lasts as long as
        // that chain, so idle lanes take over parts of it once the table is short: one lane per (pair index, product) while
        // q * nprod lanes exist, one lane per (pair index, table) -- the factors then meet through LDS -- while q * ntab <= kTailSplit.
        // Shortest tables: one lane per (pair index, table, lo | hi) for the folds, then one per (pair index, product, point) for the terms:
        // ONE product per lane and phase.  A round this short is a lone wave's instruction stream (~8 cycles per instruction, ~250 per
        // product), and the per-table split makes a wave issue 2 products for the folds plus 2 + 1 (divergent) for the terms.
        int split = 0;
        if constexpr (NFAC == 2) {
            if (2 * q * (size_t)a.ntab <= (size_t)kTailBlock) split = 3;
            else if (q * (size_t)a.ntab <= (size_t)kTailSplit) split = 2;
            else if (q * (size_t)a.nprod <= (size_t)kTailBlock) split = 1;
        }
        if (split == 0) {
            if (tid < q) {
                const Multiplier<F> mr(r);
                for (int p = 0; p < a.nprod; p++) {
                    Fe<F> lo[NFAC], hi[NFAC];
#pragma unroll
                    for (int f = 0; f < NFAC; f++) {
                        if (NFAC == 2 && f == 1 && a.tabs.in[p * NFAC + 1] == nullptr) { lo[1] = hi[1] = const_factor<F>(a.tabs, p); continue; }
                        const void *src = prev ? (const void *)(prev + (size_t)(p * NFAC + f) * cl * esz) : a.tabs.in[p * NFAC + f];
                        void *out = dst + (size_t)(p * NFAC + f) * ol * esz;
                        Fe<F> a0 = fe_load<F>(src, tid), a1 = fe_load<F>(src, tid + q);
                        Fe<F> b0 = fe_load<F>(src, tid + 2 * q), b1 = fe_load<F>(src, tid + 3 * q);
                        lo[f] = fe_add<F>(a0, mr.times(fe_sub<F>(b0, a0)));
                        hi[f] = fe_add<F>(a1, mr.times(fe_sub<F>(b1, a1)));
                        fe_store<F>(out, tid, lo[f]);
                        fe_store<F>(out, tid + q, hi[f]);
                    }
                    accumulate_terms<F, NFAC>(lo, hi, acc);
                }
            }
        } else if constexpr (NFAC == 2) {
            __shared__ Fe<F> exch[2 * kTailSplit];               // split 2: (lo, hi) of table k at pair index i: exch[2 (k q + i) ..]; split 3: exch[(2 k + h) q + i]
            const unsigned qlog = 31u - (unsigned)__builtin_clz((unsigned)q);
            const unsigned i = tid & ((unsigned)q - 1u), grp = tid >> qlog;      // pair index, product (split 1) or table (split 2)
            const Multiplier<F> mr(r);
            if (split == 1) {
                if (grp < (unsigned)a.nprod) {
                    Fe<F> lo[NFAC], hi[NFAC];
#pragma unroll
                    for (int f = 0; f < NFAC; f++) {
                        const int k = (int)grp * NFAC + f;
                        if (f == 1 && a.tabs.in[k] == nullptr) { lo[1] = hi[1] = const_factor<F>(a.tabs, (int)grp); continue; }
                        const void *src = prev ? (const void *)(prev + (size_t)k * cl * esz) : a.tabs.in[k];
                        void *out = dst + (size_t)k * ol * esz;
                        Fe<F> a0 = fe_load<F>(src, i), a1 = fe_load<F>(src, i + q);
                        Fe<F> b0 = fe_load<F>(src, i + 2 * q), b1 = fe_load<F>(src, i + 3 * q);
                        lo[f] = fe_add<F>(a0, mr.times(fe_sub<F>(b0, a0)));
                        hi[f] = fe_add<F>(a1, mr.times(fe_sub<F>(b1, a1)));
                        fe_store<F>(out, i, lo[f]);
                        fe_store<F>(out, i + q, hi[f]);
                    }
                    accumulate_terms<F, NFAC>(lo, hi, acc);
                }
            } else if (split == 3) {
                if (grp < 2u * (unsigned)a.ntab) {                   // role = (table, half): entry i + h q of the folded table
                    const unsigned k = grp >> 1, h = grp & 1u;
                    Fe<F> v;
                    if (a.tabs.in[k] == nullptr) {
                        v = const_factor<F>(a.tabs, (int)(k >> 1));
                    } else {
                        const void *src = prev ? (const void *)(prev + (size_t)k * cl * esz) : a.tabs.in[k];
                        const Fe<F> x = fe_load<F>(src, i + h * q), y = fe_load<F>(src, i + h * q + 2 * q);
                        v = fe_add<F>(x, mr.times(fe_sub<F>(y, x)));
                        fe_store<F>(dst + (size_t)k * ol * esz, i + h * q, v);
                    }
                    exch[grp * (unsigned)q + i] = v;
                }
                __syncthreads();
                if (grp < 3u * (unsigned)a.nprod) {                  // role = (product, point)
                    const unsigned p = grp / 3u, t = grp - 3u * p, b0 = 4u * p * (unsigned)q + i;
                    const Fe<F> lo = exch[b0], hi = exch[b0 + (unsigned)q], lo2 = exch[b0 + 2u * (unsigned)q], hi2 = exch[b0 + 3u * (unsigned)q];
                    const Fe<F> x = t == 0 ? lo : t == 1 ? hi : fe_sub<F>(hi, lo);            // t = 2: the node infinity (sumcheck_kernels.cuh header)
                    const Fe<F> y = t == 0 ? lo2 : t == 1 ? hi2 : fe_sub<F>(hi2, lo2);
                    const Fe<F> pr = fe_mul<F>(x, y);
                    if (t == 0) wide_add_fe<F>(acc[0], pr);
                    else if (t == 1) wide_add_fe<F>(acc[1], pr);
                    else wide_add_fe<F>(acc[2], pr);
                }
            } else {
                const bool act = grp < (unsigned)a.ntab;
                Fe<F> lo = fe_zero<F>(), hi = fe_zero<F>();
                if (act) {
                    if (a.tabs.in[grp] == nullptr) {
                        lo = hi = const_factor<F>(a.tabs, (int)(grp >> 1));
                    } else {
                        const void *src = prev ? (const void *)(prev + (size_t)grp * cl * esz) : a.tabs.in[grp];
                        void *out = dst + (size_t)grp * ol * esz;
                        Fe<F> a0 = fe_load<F>(src, i), a1 = fe_load<F>(src, i + q);
                        Fe<F> b0 = fe_load<F>(src, i + 2 * q), b1 = fe_load<F>(src, i + 3 * q);
                        lo = fe_add<F>(a0, mr.times(fe_sub<F>(b0, a0)));
                        hi = fe_add<F>(a1, mr.times(fe_sub<F>(b1, a1)));
                        fe_store<F>(out, i, lo);
                        fe_store<F>(out, i + q, hi);
                    }
                    exch[2 * (grp * (unsigned)q + i)] = lo;
                    exch[2 * (grp * (unsigned)q + i) + 1] = hi;
                }
                __syncthreads();
                if (act) {
                    const unsigned other = grp ^ 1u;             // the other factor of the same product
                    const Fe<F> lo2 = exch[2 * (other * (unsigned)q + i)], hi2 = exch[2 * (other * (unsigned)q + i) + 1];
                    if ((grp & 1u) == 0) {                       // nodes 0 and infinity: lo lo', (hi - lo)(hi' - lo')
                        wide_add_fe<F>(acc[0], fe_mul<F>(lo, lo2));
                        wide_add_fe<F>(acc[2], fe_mul<F>(fe_sub<F>(hi, lo), fe_sub<F>(hi2, lo2)));
                    } else {                                     // point 1: hi hi'
                        wide_add_fe<F>(acc[1], fe_mul<F>(hi, hi2));
                    }
                }
            }
        }
        Fe<F> tot;
        ZK_TAIL_STAMP(1);
        // waves past the last lane with a term hold zeros
        const size_t term_lanes = split == 3 ? 3 * q * (size_t)a.nprod : split == 2 ? q * (size_t)a.ntab : split == 1 ? q * (size_t)a.nprod : q;
        if (block_reduce_wide<F, NFAC + 1>(acc, sh, tot, (int)((term_lanes + 63) / 64))) S.ev[tid] = tot;
        __syncthreads();
        ZK_TAIL_STAMP(2);
        round++;
        if (tid < 64) {
            if (mb) {                                        // transcript step on the host (HostMailbox)
                mailbox_post<F>(mb, mb->ev, S.ev, NFAC + 1, seq, lane);
                ZK_TAIL_STAMP(3);
                const Fe<F> rn = mailbox_wait_challenge<F>(mb, seq, lane);
                ZK_TAIL_STAMP(4);
                if (lane == 0) {
                    S.chal = rn;
                    fe_store<F>(a.ctx.proof, a.chal_base + a.per * round, S.chal);
                }
                seq++;
            } else {
                round_message_and_challenge<F>(S, a.ctx, 0, 0, a.msg_base + a.per * round, a.chal_base + a.per * round, fill, lane);
            }
        }
        __syncthreads();                                     // also orders this round's global stores before the next round's loads
        ZK_TAIL_STAMP(5);
        r = S.chal;
        prev = dst;
        cl = ol;
        j++;
    }
    if (cl == 2 && (int)tid < a.ntab) {                      // last round: 2 entries -> 1 (nothing left to sum)
        const Multiplier<F> mr(r);
        Fe<F> v;
        if (a.tabs.in[tid] == nullptr) {
            v = const_factor<F>(a.tabs, (int)(tid / NFAC));
        } else {
            const void *src = prev ? (const void *)(prev + (size_t)tid * cl * esz) : a.tabs.in[tid];
            Fe<F> y1 = fe_load<F>(src, 0), y2 = fe_load<F>(src, 1);
            v = fe_add<F>(y1, mr.times(fe_sub<F>(y2, y1)));
        }
        fe_store<F>((char *)a.buf[j & 1] + (size_t)tid * esz, 0, v);
        if (a.fin_slot != ~(size_t)0) fe_store<F>(a.ctx.proof, a.fin_slot + tid, v);
        if (mb) {
#pragma unroll
            for (int i = 0; i < F::N; i++) mb->fin[tid * 12 + i] = v.l[i];
        }
    }
    if (mb) {                                                // the final values: one more post, no answer awaited
        __threadfence_system();
        __syncthreads();
        if (tid == 0) __atomic_store_n(&mb->gpu_seq, seq, __ATOMIC_RELEASE);
        return;
    }
    if (tid < 25) a.ctx.sponge->a[tid] = S.st[tid];
    if (tid == 0) a.ctx.sponge->fill = fill;
}

}  // namespace zk
