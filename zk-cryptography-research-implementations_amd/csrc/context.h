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
// device scratch for reduction partials: at least `bytes` bytes, owned per device, reused
int scratch(size_t bytes, void **out);
// pinned host staging for small results (a few field elements)
int host_staging(size_t bytes, void **out);

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
// two pinned host staging buffers of at least `bytes` each, owned per device
int pinned_pair(size_t bytes, void *out[2]);
// a temporary table backed by the caching pool (internal provers: dozens of same-sized temporaries per proof)
int table_alloc_pooled(int field, size_t len, zk_table **out);
}
