// zkmle_host.hip -- C ABI: tiny host-side field helpers (control path: transcripts, fixtures).
// These run on the CPU by design (a few elements per call); table-sized work never comes here.
#include <string.h>

#include "context.h"
#include "ufield.cuh"
#include "host_field.h"

using namespace zk;

extern "C" {

int zk_fe_from_u64(int field, uint64_t v, uint64_t *out) {
    if (!out) return ZK_E_ARG;
    ZK_DISPATCH_FIELD(field, { Fe<F> e = fe_from_u64<F>(v); memcpy(out, e.l, 4 * F::N); });
    return ZK_OK;
}
int zk_fe_to_bytes_be(int field, const uint64_t *a, uint8_t *out) {
    if (!a || !out) return ZK_E_ARG;
    ZK_DISPATCH_FIELD(field, { Fe<F> e; memcpy(e.l, a, 4 * F::N); host_to_bytes_be<F>(e, out); });
    return ZK_OK;
}
int zk_fe_from_le_bytes_mod_order(int field, const uint8_t *bytes, size_t n, uint64_t *out) {
    if ((!bytes && n) || !out) return ZK_E_ARG;
    ZK_DISPATCH_FIELD(field, { Fe<F> e = host_from_le_bytes_mod_order<F>(bytes, n); memcpy(out, e.l, 4 * F::N); });
    return ZK_OK;
}
#define ZK_HOST_BINOP(name, expr)                                                          \
    int name(int field, const uint64_t *a, const uint64_t *b, uint64_t *out) {             \
        if (!a || !b || !out) return ZK_E_ARG;                                             \
        ZK_DISPATCH_FIELD(field, {                                                         \
            Fe<F> x, y;                                                                    \
            memcpy(x.l, a, 4 * F::N);                                                      \
            memcpy(y.l, b, 4 * F::N);                                                      \
            Fe<F> o = expr;                                                                \
            memcpy(out, o.l, 4 * F::N);                                                    \
        });                                                                                \
        return ZK_OK;                                                                      \
    }
ZK_HOST_BINOP(zk_fe_add, fe_add<F>(x, y))
ZK_HOST_BINOP(zk_fe_sub, fe_sub<F>(x, y))
ZK_HOST_BINOP(zk_fe_mul, fe_mul<F>(x, y))
int zk_fe_inv(int field, const uint64_t *a, uint64_t *out) {
    if (!a || !out) return ZK_E_ARG;
    ZK_DISPATCH_FIELD(field, { Fe<F> x; memcpy(x.l, a, 4 * F::N); Fe<F> o = fe_inv<F>(x); memcpy(out, o.l, 4 * F::N); });
    return ZK_OK;
}
int zk_vec_from_canonical(int field, const uint64_t *canon, size_t n, uint64_t *mont) {
    if (!canon || !mont) return ZK_E_ARG;
    ZK_DISPATCH_FIELD(field, {
        for (size_t i = 0; i < n; i++) {
            Fe<F> c;
            memcpy(c.l, canon + i * (F::N / 2), 4 * F::N);
            Fe<F> m = fe_from_canonical<F>(c);
            memcpy(mont + i * (F::N / 2), m.l, 4 * F::N);
        }
    });
    return ZK_OK;
}
int zk_vec_to_canonical(int field, const uint64_t *mont, size_t n, uint64_t *canon) {
    if (!canon || !mont) return ZK_E_ARG;
    ZK_DISPATCH_FIELD(field, {
        for (size_t i = 0; i < n; i++) {
            Fe<F> m;
            memcpy(m.l, mont + i * (F::N / 2), 4 * F::N);
            Fe<F> c = fe_to_canonical<F>(m);
            memcpy(canon + i * (F::N / 2), c.l, 4 * F::N);
        }
    });
    return ZK_OK;
}

}  // extern "C"
