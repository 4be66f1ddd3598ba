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
// device scratch for reduction partials: at least `bytes` bytes, owned per device, reused
int scratch(size_t bytes, void **out);
// pinned host staging for small results (a few field elements)
int host_staging(size_t bytes, void **out);

// Caching device allocator for per-call scratch (the MSM allocates several GB per call; hipMalloc / hipFree of
// such blocks costs milliseconds).  Freed blocks are kept per device and reused by later calls of similar size;
// zk_release_cached_memory() returns them to the driver.
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
