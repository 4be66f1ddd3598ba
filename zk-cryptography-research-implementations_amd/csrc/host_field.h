// host_field.h -- host-side byte conversions of field elements (transcript encodings).
//   to_bytes_be : into_bigint().to_bytes_be()  evaluation_form.rs:39, prover.rs:92, sumcheck_gkr_protocol.rs:153
//   to_bytes_le : into_bigint().to_bytes_le()  sumcheck_gkr_protocol.rs:148
//   from_le_bytes_mod_order : fiat_shamir_transcript.rs:42 [ark-ff]
#pragma once
#include "ufield.cuh"

namespace zk {

template <class F> inline void host_to_bytes_le(const Fe<F> &a, uint8_t *out) {
    Fe<F> c = fe_to_canonical<F>(a);
    for (int i = 0; i < F::N; i++)
        for (int k = 0; k < 4; k++) out[4 * i + k] = (uint8_t)(c.l[i] >> (8 * k));
}
template <class F> inline void host_to_bytes_be(const Fe<F> &a, uint8_t *out) {
    uint8_t le[4 * F::N];
    host_to_bytes_le<F>(a, le);
    for (int i = 0; i < 4 * F::N; i++) out[i] = le[4 * F::N - 1 - i];
}
// little-endian integer of any length reduced mod p.  Up to 4 N bytes (every Fiat-Shamir challenge: a 32-byte digest) the integer fits
// the limbs: a few subtractions of p, then one product by R^2 -- the step sits on the latency path of every sumcheck round.  Longer
// inputs: Horner over bytes from the top, acc = acc * 256 + byte, all in Montgomery form.
template <class F> inline Fe<F> host_from_le_bytes_mod_order(const uint8_t *b, size_t n) {
    if (n <= (size_t)(4 * F::N)) {
        Fe<F> c = fe_zero<F>();
        for (size_t i = 0; i < n; i++) c.l[i / 4] |= (uint32_t)b[i] << (8 * (i % 4));
        for (;;) {                                           // c < 2^(32 N) < 16 p for the four moduli
            Fe<F> d;
            uint64_t borrow = 0;
            for (int i = 0; i < F::N; i++) {
                const uint64_t v = (uint64_t)c.l[i] - F::p(i) - borrow;
                d.l[i] = (uint32_t)v;
                borrow = (v >> 32) & 1u;
            }
            if (borrow) break;                               // c < p
            c = d;
        }
        return fe_from_canonical<F>(c);
    }
    Fe<F> acc = fe_zero<F>();
    const Fe<F> c256 = fe_from_u64<F>(256);
    for (size_t i = n; i-- > 0;) {
        acc = fe_mul<F>(acc, c256);
        acc = fe_add<F>(acc, fe_from_u64<F>(b[i]));
    }
    return acc;
}

}  // namespace zk
