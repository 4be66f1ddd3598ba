// context.h -- library-internal: status helpers, HIP error capture, per-device scratch, tables.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>

#include "../../include/zkmle.h"

namespace zk {

void set_last_error(const std::string &s);

#define ZK_HIP(call)                                                                          \
    do {                                                                                      \
        hipError_t e__ = (call);                                                              \
        if (e__ != hipSuccess) {                                                              \
            ::zk::set_last_error(std::string(#call) + ": " + hipGetErrorString(e__));         \
            return (e__ == hipErrorNoDevice || e__ == hipErrorInvalidDevice ||                \
                    e__ == hipErrorInsufficientDriver) ? ZK_E_NO_DEVICE : ZK_E_HIP;           \
        }                                                                                     \
    } while (0)

#define ZK_TRY(expr)                         \
    do {                                     \
        int rc__ = (expr);                   \
        if (rc__ != ZK_OK) return rc__;      \
    } while (0)

// Fails loudly (ZK_E_NO_DEVICE) when there is no GPU: there is no CPU fallback in this library.
int require_device();
// The calling thread's current stream (zk_set_stream; the null stream by default).  Every kernel launch, copy and
// synchronisation of the library goes through it, so threads that set different streams run their calls concurrently on
// the device.  Scratch buffers, the caching pool and the statistics are per thread as well.
hipStream_t cur_stream();
hipError_t memcpy_on_stream(void *dst, const void *src, size_t bytes, hipMemcpyKind kind);   // async copy + stream sync
hipError_t memset_on_stream(void *dst, int value, size_t bytes);
// wait until the current stream has drained, polling hipStreamQuery for the first ~0.3 ms (a blocking hipStreamSynchronize wakes up
// 20-30 us after the last kernel has ended: a quarter of a small proof) before falling back to it
hipError_t stream_wait_idle();
// one block of words per thread and device, zero when handed out: the arrival counter (first 64 bytes) and the segment accumulators of
// the passes that run their exchange themselves (basic_multi.cuh MultiFin).  Whoever uses it leaves it zero; sync_words_reset() after a
// proof that may not have.
constexpr size_t kSyncCounterBytes = 64 * (1 + 256);        // the pass's counter and one per segment, 64 bytes apart
constexpr size_t kSyncWordsBytes = kSyncCounterBytes + (size_t)256 * 12 * 128;   // + one 128-byte line per (segment, limb) accumulator word
int sync_words(void **out);
int sync_words_reset();
// device scratch for reduction partials: at least `bytes` bytes, owned per device, reused
int scratch(size_t bytes, void **out);
// pinned host staging for small results (a few field elements)
int host_staging(size_t bytes, void **out);
// one block (16 KiB) of pinned, COHERENT (fine-grained) host memory per thread and device, mapped into the device: the mailbox of the
// host-assisted transcript step (dev_transcript.cuh HostMailbox).  *host and *dev address the same memory.
int host_mailbox(void **host, void **dev);

// Caching device allocator for per-call scratch (the MSM allocates several GB per call; hipMalloc / hipFree of
// such blocks costs milliseconds).  Freed blocks are kept per THREAD and device and reused by that thread's later calls of
// similar size (stream-ordered: a thread's work runs on its current stream, and changing the stream synchronises the
// old one first); zk_release_cached_memory() returns the calling thread's blocks to the driver.
int pool_alloc(size_t bytes, void **out);
void pool_free(void *p);

inline bool is_pow2(size_t x) { return x && !(x & (x - 1)); }
inline unsigned ilog2(size_t x) { unsigned k = 0; while (x >>= 1) k++; return k; }
inline int field_limbs64(int field) { return field == ZK_FQ381 ? 6 : (field >= 0 && field <= 3 ? 4 : -1); }

}  // namespace zk

struct zk_table {
    int field;
    size_t len;
    void *dptr;
    int owned;      // 0 = view of caller memory (zk_table_wrap), 1 = hipMalloc, 2 = block of the scratch pool
};

namespace zk {
class Transcript;
// transcript.append(convert_to_bytes(table)) (evaluation_form.rs:35-43, prover.rs:38-39): the GPU converts Montgomery ->
// canonical big-endian chunk by chunk into pinned host buffers while the host hashes the previous chunk (zkmle_sumcheck.hip)
int transcript_absorb_table(Transcript &t, const zk_table *table);
// Proof slots shared by a proof made of several sumchecks (zkmle_sumcheck.hip): the sponge, the interpolation basis and every slot
// (coefficients, challenges, final values, layer links) live in ONE device block; rounds() and link() only enqueue kernels on the
// current stream, collect() is the single download.  Slot layout of rounds(): round k's nfac + 1 coefficients at s0 + per k, its
// challenge at s0 + per k + nfac + 1, then the nprod * nfac final values (per = nfac + 2).
struct ProofSlotsBase {
    virtual ~ProofSlotsBase() {}
    virtual void *slot_ptr(size_t s) const = 0;                    // device address of slot s
    virtual int upload_slot(size_t s, const uint64_t *el) = 0;
    // enqueue the rounds of a sum-of-products sumcheck (constant second factors: tables[p * 2 + 1] == null, value from const_host
    // (nprod elements) or const_dev[p] (device memory)); with_claim: proof[claim_slot] is absorbed in front of round 0's message
    virtual int rounds(size_t s0, const zk_table *const *tables, size_t nprod, size_t nfac, const uint64_t *const_host, const void *const *const_dev,
                       int with_claim, size_t claim_slot) = 0;
    // gkr_protocol.rs:109-133 on the device: append wb, alpha, append wc, beta, claim = alpha wb + beta wc
    // every sumcheck enqueued through rounds() proves a sum the prover computed itself; the first one's is told here (the later ones' are the
    // running claim of the previous phase or link), so that round 0 may derive e(1) = claim - e(0)
    virtual void set_claim(const uint64_t *el) = 0;
    virtual int link(size_t wb_src, size_t wc_src, size_t wb_slot, size_t wc_slot, size_t alpha_slot, size_t beta_slot, size_t claim_slot) = 0;
    virtual int collect(Transcript &tr, uint64_t *host_slots) = 0;   // every slot (field elements, u64 limbs) + the sponge back into tr
};
int proof_slots_new(int field, Transcript &tr, size_t npts, size_t nslots, ProofSlotsBase **out);
// open_and_prove's body with the subtracted value and the leftover entry exposed (zkmle_kzg.hip; the sharded opening builds on it)
int kzg_open_core(const zk_table *poly, const zk_g1_bases *g1_powers, const zk_kzg_opening_key *key, const uint64_t *opening, size_t nopen,
                  const uint64_t *v_given, uint64_t *evaluation, uint64_t *proofs, uint64_t *last);
int kzg_key_total(const zk_kzg_opening_key *k, const zk_g1_bases *g1, uint64_t *out12);
// two pinned host staging buffers of at least `bytes` each, owned per device
int pinned_pair(size_t bytes, void *out[2]);
// a temporary table backed by the caching pool (internal provers: dozens of same-sized temporaries per proof)
int table_alloc_pooled(int field, size_t len, zk_table **out);
// alpha fold(in, rb) + beta fold(in, rc) over the k <= 8 top variables in one pass (gkr/src/utils.rs:23-68; zkmle_core.hip)
int mle_fold_alpha_beta(const zk_table *in, size_t k, const uint64_t *alpha, const uint64_t *beta, const uint64_t *rb, const uint64_t *rc, zk_table *out);
}
