// transcript.h -- host-side Fiat-Shamir transcript (control path; strictly sequential, tiny).
// Mirrors transcripts/src/fiat_shamir/fiat_shamir_transcript.rs:5-43:
//   append                              -> hasher.update                      (:22-24)
//   sample_random_challenge             -> finalize a CLONE of the running sponge, then absorb the
//                                          32-byte digest back; the state is NOT reset (:29-36)
//   random_challenge_as_field_element   -> F::from_le_bytes_mod_order(digest) (:38-43)
// Keccak-256 = sha3 0.10.8 `Keccak256`: rate 136 bytes, original pad 0x01 .. 0x80 (not SHA3-256).
#pragma once
#include <stddef.h>
#include <stdint.h>
#include <string.h>

#include "host_field.h"

namespace zk {

// Keccak-f[1600] on 25 lanes.  The same source is compiled twice: for the baseline x86-64 ISA and with BMI / BMI2 (andn for
// chi, rorx for the rotations: +14 % on one core of the GPU box's host, 0.62 -> 0.71 GB/s); chosen once at run time.  The
// whole-table absorb of a prover is bound by this loop.
static inline uint64_t keccak_rotl(uint64_t x, unsigned s) { return s ? (x << s) | (x >> (64 - s)) : x; }
#define ZK_KECCAK_F1600_BODY                                                                                                       \
    static const uint64_t rc[24] = {                                                                                               \
        0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull,                                \
        0x000000000000808bull, 0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull,                                \
        0x000000000000008aull, 0x0000000000000088ull, 0x0000000080008009ull, 0x000000008000000aull,                                \
        0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull, 0x8000000000008003ull,                                \
        0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,                                \
        0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};                               \
    /* rho offsets indexed [x + 5 y] */                                                                                            \
    static const unsigned rho[25] = {0, 1, 62, 28, 27, 36, 44, 6, 55, 20, 3, 10, 43, 25, 39, 41, 45, 15, 21, 8, 18, 2, 61, 56, 14};  \
    for (int round = 0; round < 24; round++) {                                                                                     \
        uint64_t c[5], b[25];                                                                                                      \
        for (int x = 0; x < 5; x++) c[x] = a[x] ^ a[x + 5] ^ a[x + 10] ^ a[x + 15] ^ a[x + 20];                                    \
        for (int x = 0; x < 5; x++) {                                                                                              \
            uint64_t d = c[(x + 4) % 5] ^ keccak_rotl(c[(x + 1) % 5], 1);                                                          \
            for (int y = 0; y < 5; y++) a[x + 5 * y] ^= d;                                                                         \
        }                                                                                                                          \
        for (int x = 0; x < 5; x++)                                                                                                \
            for (int y = 0; y < 5; y++) b[y + 5 * ((2 * x + 3 * y) % 5)] = keccak_rotl(a[x + 5 * y], rho[x + 5 * y]);              \
        for (int y = 0; y < 5; y++)                                                                                                \
            for (int x = 0; x < 5; x++) a[x + 5 * y] = b[x + 5 * y] ^ (~b[(x + 1) % 5 + 5 * y] & b[(x + 2) % 5 + 5 * y]);          \
        a[0] ^= rc[round];                                                                                                         \
    }
static inline void keccak_f1600_generic(uint64_t *a) { ZK_KECCAK_F1600_BODY }
#if defined(__x86_64__)
__attribute__((target("bmi,bmi2"))) static inline void keccak_f1600_bmi2(uint64_t *a) { ZK_KECCAK_F1600_BODY }
static inline void keccak_f1600(uint64_t *a) {
    static const bool fast = __builtin_cpu_supports("bmi2") && __builtin_cpu_supports("bmi");
    if (fast) keccak_f1600_bmi2(a); else keccak_f1600_generic(a);
}
#else
static inline void keccak_f1600(uint64_t *a) { keccak_f1600_generic(a); }
#endif
#undef ZK_KECCAK_F1600_BODY

class Keccak256 {
  public:
    Keccak256() { memset(a_, 0, sizeof a_); fill_ = 0; }
    void update(const uint8_t *data, size_t n) {
        // (1) finish an open block byte by byte, (2) whole blocks straight from the caller's buffer as 17 little-endian lanes
        // (the table absorb of Prover::prove, prover.rs:38-39, is 32 * 2^n bytes of sequential sponge input and arrives in
        // chunks that are not multiples of the rate), (3) keep the tail
        if (fill_ != 0) {
            size_t room = kRate - fill_, take = n < room ? n : room;
            xor_in(data, take);
            data += take;
            n -= take;
            if (fill_ == kRate) { permute(); fill_ = 0; }
        }
        while (n >= kRate) {                                 // here fill_ == 0
            for (int i = 0; i < 17; i++) {
                uint64_t w;
                memcpy(&w, data + 8 * i, 8);
                a_[i] ^= w;            // x86-64 / little-endian hosts only (the library targets ROCm hosts)
            }
            permute();
            data += kRate;
            n -= kRate;
        }
        if (n) xor_in(data, n);                              // < rate bytes into an empty block
    }
    // digest of everything absorbed so far; *this is left untouched
    void finalize_copy(uint8_t out[32]) const {
        Keccak256 c = *this;
        uint8_t pad = 0x01;
        c.xor_in(&pad, 1);
        c.lane_xor(kRate - 1, 0x80);
        c.permute();
        for (int i = 0; i < 32; i++) out[i] = (uint8_t)(c.a_[i / 8] >> (8 * (i % 8)));
    }

    // the sponge moves between host and device (dev_transcript.cuh): 25 lanes + bytes in the open block
    void export_state(uint64_t a[25], uint32_t *fill) const { memcpy(a, a_, sizeof a_); *fill = (uint32_t)fill_; }
    void import_state(const uint64_t a[25], uint32_t fill) { memcpy(a_, a, sizeof a_); fill_ = fill; }

  private:
    static constexpr size_t kRate = 136;
    uint64_t a_[25];
    size_t fill_;

    void lane_xor(size_t byte_pos, uint8_t v) { a_[byte_pos / 8] ^= (uint64_t)v << (8 * (byte_pos % 8)); }
    void xor_in(const uint8_t *d, size_t n) {
        size_t i = 0;
        if (fill_ % 8 == 0)                                  // whole lanes at once (round messages are 32-byte elements on a 32-byte grid)
            for (; i + 8 <= n; i += 8) {
                uint64_t w;
                memcpy(&w, d + i, 8);
                a_[(fill_ + i) / 8] ^= w;                    // little-endian hosts (see update)
            }
        for (; i < n; i++) lane_xor(fill_ + i, d[i]);
        fill_ += n;
    }
    void permute() { keccak_f1600(a_); }
};

class Transcript {
  public:
    void append(const uint8_t *data, size_t n) { h_.update(data, n); }
    void sample_random_challenge(uint8_t out[32]) {
        h_.finalize_copy(out);
        h_.update(out, 32);
    }
    template <class F> Fe<F> random_challenge_as_field_element() {
        uint8_t d[32];
        sample_random_challenge(d);
        return host_from_le_bytes_mod_order<F>(d, 32);
    }
    template <class F> void append_be(const Fe<F> &x) {   // field_element_to_bytes prover.rs:91-93
        uint8_t b[4 * F::N];
        host_to_bytes_be<F>(x, b);
        append(b, sizeof b);
    }
    template <class F> void append_le(const Fe<F> &x) {   // univariate_to_bytes sumcheck_gkr_protocol.rs:145-150
        uint8_t b[4 * F::N];
        host_to_bytes_le<F>(x, b);
        append(b, sizeof b);
    }

    Keccak256 &sponge() { return h_; }

  private:
    Keccak256 h_;
};

}  // namespace zk

struct zk_transcript {
    zk::Transcript t;
};
