/*
 * field.c -- oracle: prime fields (Montgomery, R = 2^(64*limbs)), Keccak-256, transcript.
 * TEST INFRASTRUCTURE ONLY (see zkoracle.h).
 *
 * The reference never touches limbs; it calls ark-ff 0.5.0 `Fp` ops
 * (evaluation_form.rs:53,88-89,118,138,159; prover.rs:28,82-83).  ark-ff keeps elements
 * fully reduced in Montgomery form with R = 2^(64*N); this file restates the published
 * CIOS algorithm so that in-memory limbs equal arkworks' [ext].
 * Constants R, R^2 and -p^-1 are DERIVED from the modulus at first use and pinned
 * against SURVEY.md Appendix A in tests/test_oracle_fields.py.
 */
#include "zk_internal.h"

static field_t g_fields[ORC_NFIELDS] = {
    /* BLS12-381 Fr */
    {4, {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL}, {0}, {0}, 0, 0},
    /* BLS12-381 Fq */
    {6, {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL, 0x64774b84f38512bfULL,
         0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL}, {0}, {0}, 0, 0},
    /* BN254 Fq */
    {4, {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}, {0}, {0}, 0, 0},
    /* BN254 Fr */
    {4, {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL}, {0}, {0}, 0, 0},
};

static int geq_p(const field_t *F, const uint64_t *a) {
    for (int i = F->n - 1; i >= 0; i--) {
        if (a[i] > F->p[i]) return 1;
        if (a[i] < F->p[i]) return 0;
    }
    return 1;
}
static void sub_p(const field_t *F, uint64_t *a) {
    uint64_t borrow = 0;
    for (int i = 0; i < F->n; i++) {
        u128 d = (u128)a[i] - F->p[i] - borrow;
        a[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
}
/* a = 2a mod p, canonical arithmetic (used only to derive constants) */
static void dbl_mod(const field_t *F, uint64_t *a) {
    uint64_t carry = 0;
    for (int i = 0; i < F->n; i++) {
        uint64_t nc = a[i] >> 63;
        a[i] = (a[i] << 1) | carry;
        carry = nc;
    }
    if (carry || geq_p(F, a)) sub_p(F, a);
}

static void field_init(field_t *F) {
    /* inv = -p^-1 mod 2^64 by Newton iteration */
    uint64_t x = 1;
    for (int i = 0; i < 6; i++) x *= 2 - F->p[0] * x;
    F->inv = (uint64_t)0 - x;
    uint64_t a[MAXL] = {1, 0, 0, 0, 0, 0};
    for (int i = 0; i < 64 * F->n; i++) dbl_mod(F, a);
    memcpy(F->r, a, sizeof a);
    for (int i = 0; i < 64 * F->n; i++) dbl_mod(F, a);
    memcpy(F->r2, a, sizeof a);
    F->ready = 1;
}

const field_t *orc_fld(int id) {
    if (id < 0 || id >= ORC_NFIELDS) return NULL;
    field_t *F = &g_fields[id];
    if (!F->ready) field_init(F);
    return F;
}

void fe_zero(fe *o) { memset(o, 0, sizeof *o); }
void fe_one(const field_t *F, fe *o) { fe_zero(o); memcpy(o->l, F->r, 8 * (size_t)F->n); }

void fe_add(const field_t *F, fe *o, const fe *a, const fe *b) {
    uint64_t t[MAXL] = {0};
    uint64_t carry = 0;
    for (int i = 0; i < F->n; i++) {
        u128 s = (u128)a->l[i] + b->l[i] + carry;
        t[i] = (uint64_t)s;
        carry = (uint64_t)(s >> 64);
    }
    if (carry || geq_p(F, t)) sub_p(F, t);
    fe_zero(o);
    memcpy(o->l, t, 8 * (size_t)F->n);
}

void fe_sub(const field_t *F, fe *o, const fe *a, const fe *b) {
    uint64_t t[MAXL] = {0};
    uint64_t borrow = 0;
    for (int i = 0; i < F->n; i++) {
        u128 d = (u128)a->l[i] - b->l[i] - borrow;
        t[i] = (uint64_t)d;
        borrow = (uint64_t)(d >> 64) & 1;
    }
    if (borrow) {
        uint64_t carry = 0;
        for (int i = 0; i < F->n; i++) {
            u128 s = (u128)t[i] + F->p[i] + carry;
            t[i] = (uint64_t)s;
            carry = (uint64_t)(s >> 64);
        }
    }
    fe_zero(o);
    memcpy(o->l, t, 8 * (size_t)F->n);
}

void fe_neg(const field_t *F, fe *o, const fe *a) {
    fe z;
    fe_zero(&z);
    fe_sub(F, o, &z, a);
}

/* CIOS Montgomery product: o = a*b*R^-1 mod p, fully reduced */
void fe_mul(const field_t *F, fe *o, const fe *a, const fe *b) {
    const int n = F->n;
    uint64_t t[MAXL + 2] = {0};
    for (int i = 0; i < n; i++) {
        uint64_t carry = 0;
        for (int j = 0; j < n; j++) {
            u128 x = (u128)a->l[j] * b->l[i] + t[j] + carry;
            t[j] = (uint64_t)x;
            carry = (uint64_t)(x >> 64);
        }
        u128 x = (u128)t[n] + carry;
        t[n] = (uint64_t)x;
        t[n + 1] = (uint64_t)(x >> 64);
        uint64_t m = t[0] * F->inv;
        x = (u128)m * F->p[0] + t[0];
        carry = (uint64_t)(x >> 64);
        for (int j = 1; j < n; j++) {
            x = (u128)m * F->p[j] + t[j] + carry;
            t[j - 1] = (uint64_t)x;
            carry = (uint64_t)(x >> 64);
        }
        x = (u128)t[n] + carry;
        t[n - 1] = (uint64_t)x;
        t[n] = t[n + 1] + (uint64_t)(x >> 64);
    }
    if (t[n] || geq_p(F, t)) sub_p(F, t);
    fe_zero(o);
    memcpy(o->l, t, 8 * (size_t)n);
}

void fe_from_canonical(const field_t *F, fe *o, const uint64_t *canon) {
    fe c, r2;
    fe_zero(&c); fe_zero(&r2);
    memcpy(c.l, canon, 8 * (size_t)F->n);
    memcpy(r2.l, F->r2, 8 * (size_t)F->n);
    fe_mul(F, o, &c, &r2);
}

void fe_from_u64(const field_t *F, fe *o, uint64_t v) {
    uint64_t c[MAXL] = {v, 0, 0, 0, 0, 0};
    fe_from_canonical(F, o, c);
}

void fe_to_canonical(const field_t *F, uint64_t *out, const fe *a) {
    fe one_raw, t;
    fe_zero(&one_raw);
    one_raw.l[0] = 1;
    fe_mul(F, &t, a, &one_raw);
    memcpy(out, t.l, 8 * (size_t)F->n);
}

/* a^(p-2) */
void fe_inv(const field_t *F, fe *o, const fe *a) {
    uint64_t e[MAXL];
    memcpy(e, F->p, sizeof e);
    /* p - 2: p is odd and p[0] >= 3 for every field here */
    e[0] -= 2;
    fe acc, base = *a;
    fe_one(F, &acc);
    for (int i = 0; i < 64 * F->n; i++) {
        if ((e[i / 64] >> (i % 64)) & 1) fe_mul(F, &acc, &acc, &base);
        fe_mul(F, &base, &base, &base);
    }
    *o = acc;
}

/* F::from_le_bytes_mod_order [ark-ff]: the little-endian integer reduced mod p.
 * Horner from the most significant byte: acc = acc*256 + byte. */
void fe_from_le_bytes(const field_t *F, fe *o, const uint8_t *b, size_t n) {
    fe acc, c256, d;
    fe_zero(&acc);
    fe_from_u64(F, &c256, 256);
    for (size_t i = n; i-- > 0;) {
        fe_mul(F, &acc, &acc, &c256);
        fe_from_u64(F, &d, b[i]);
        fe_add(F, &acc, &acc, &d);
    }
    *o = acc;
}

void fe_to_le_bytes(const field_t *F, uint8_t *out, const fe *a) {
    uint64_t c[MAXL];
    fe_to_canonical(F, c, a);
    for (int i = 0; i < F->n; i++)
        for (int k = 0; k < 8; k++) out[8 * i + k] = (uint8_t)(c[i] >> (8 * k));
}
void fe_to_be_bytes(const field_t *F, uint8_t *out, const fe *a) {
    uint8_t le[8 * MAXL];
    fe_to_le_bytes(F, le, a);
    int nb = 8 * F->n;
    for (int i = 0; i < nb; i++) out[i] = le[nb - 1 - i];
}

/* ---- public wrappers ---------------------------------------------------- */
int orc_field_limbs(int field) {
    const field_t *F = orc_fld(field);
    return F ? F->n : ORC_E_ARG;
}
int orc_field_constants(int field, uint64_t *modulus, uint64_t *r, uint64_t *r2, uint64_t *inv) {
    const field_t *F = orc_fld(field);
    if (!F) return ORC_E_ARG;
    memcpy(modulus, F->p, 8 * (size_t)F->n);
    memcpy(r, F->r, 8 * (size_t)F->n);
    memcpy(r2, F->r2, 8 * (size_t)F->n);
    *inv = F->inv;
    return ORC_OK;
}
#define GETF const field_t *F = orc_fld(field); if (!F) return ORC_E_ARG
int orc_fe_from_u64(int field, uint64_t v, uint64_t *out) {
    GETF; fe o; fe_from_u64(F, &o, v); fe_store(F, out, &o); return ORC_OK;
}
int orc_fe_from_le_bytes_mod_order(int field, const uint8_t *bytes, size_t n, uint64_t *out) {
    GETF; fe o; fe_from_le_bytes(F, &o, bytes, n); fe_store(F, out, &o); return ORC_OK;
}
int orc_fe_to_bytes_be(int field, const uint64_t *a, uint8_t *out) {
    GETF; fe x; fe_load(F, &x, a); fe_to_be_bytes(F, out, &x); return ORC_OK;
}
int orc_fe_to_bytes_le(int field, const uint64_t *a, uint8_t *out) {
    GETF; fe x; fe_load(F, &x, a); fe_to_le_bytes(F, out, &x); return ORC_OK;
}
#define BINOP(name, fn) \
    int name(int field, const uint64_t *a, const uint64_t *b, uint64_t *out) { \
        GETF; fe x, y, o; fe_load(F, &x, a); fe_load(F, &y, b); fn(F, &o, &x, &y); \
        fe_store(F, out, &o); return ORC_OK; }
BINOP(orc_fe_add, fe_add)
BINOP(orc_fe_sub, fe_sub)
BINOP(orc_fe_mul, fe_mul)
int orc_fe_neg(int field, const uint64_t *a, uint64_t *out) {
    GETF; fe x, o; fe_load(F, &x, a); fe_neg(F, &o, &x); fe_store(F, out, &o); return ORC_OK;
}
int orc_fe_inv(int field, const uint64_t *a, uint64_t *out) {
    GETF; fe x, o; fe_load(F, &x, a); fe_inv(F, &o, &x); fe_store(F, out, &o); return ORC_OK;
}
int orc_vec_from_canonical(int field, const uint64_t *canon, size_t n, uint64_t *mont) {
    GETF;
    for (size_t i = 0; i < n; i++) {
        fe o;
        fe_from_canonical(F, &o, canon + i * F->n);
        fe_store(F, mont + i * F->n, &o);
    }
    return ORC_OK;
}
int orc_vec_to_canonical(int field, const uint64_t *mont, size_t n, uint64_t *canon) {
    GETF;
    for (size_t i = 0; i < n; i++) {
        fe x;
        fe_load(F, &x, mont + i * F->n);
        fe_to_canonical(F, canon + i * F->n, &x);
    }
    return ORC_OK;
}

/* ---- Keccak-256 (original Keccak padding 0x01..0x80, rate 136; sha3 0.10.8 Keccak256) --- */
static const uint64_t RC[24] = {
    0x0000000000000001ULL, 0x0000000000008082ULL, 0x800000000000808aULL, 0x8000000080008000ULL,
    0x000000000000808bULL, 0x0000000080000001ULL, 0x8000000080008081ULL, 0x8000000000008009ULL,
    0x000000000000008aULL, 0x0000000000000088ULL, 0x0000000080008009ULL, 0x000000008000000aULL,
    0x000000008000808bULL, 0x800000000000008bULL, 0x8000000000008089ULL, 0x8000000000008003ULL,
    0x8000000000008002ULL, 0x8000000000000080ULL, 0x000000000000800aULL, 0x800000008000000aULL,
    0x8000000080008081ULL, 0x8000000000008080ULL, 0x0000000080000001ULL, 0x8000000080008008ULL};
static const int ROTC[24] = {1, 3, 6, 10, 15, 21, 28, 36, 45, 55, 2, 14, 27, 41, 56, 8, 25, 43, 62, 18, 39, 61, 20, 44};
static const int PILN[24] = {10, 7, 11, 17, 18, 3, 5, 16, 8, 21, 24, 4, 15, 23, 19, 13, 12, 2, 20, 14, 22, 9, 6, 1};
#define ROL(x, s) (((x) << (s)) | ((x) >> (64 - (s))))
static void keccakf(uint64_t st[25]) {
    for (int round = 0; round < 24; round++) {
        uint64_t bc[5];
        for (int i = 0; i < 5; i++) bc[i] = st[i] ^ st[i + 5] ^ st[i + 10] ^ st[i + 15] ^ st[i + 20];
        for (int i = 0; i < 5; i++) {
            uint64_t t = bc[(i + 4) % 5] ^ ROL(bc[(i + 1) % 5], 1);
            for (int j = 0; j < 25; j += 5) st[j + i] ^= t;
        }
        uint64_t t = st[1];
        for (int i = 0; i < 24; i++) {
            int j = PILN[i];
            uint64_t b = st[j];
            st[j] = ROL(t, ROTC[i]);
            t = b;
        }
        for (int j = 0; j < 25; j += 5) {
            for (int i = 0; i < 5; i++) bc[i] = st[j + i];
            for (int i = 0; i < 5; i++) st[j + i] ^= (~bc[(i + 1) % 5]) & bc[(i + 2) % 5];
        }
        st[0] ^= RC[round];
    }
}
static void absorb_block(uint64_t st[25], const uint8_t *blk) {
    for (int i = 0; i < 17; i++) {
        uint64_t w = 0;
        for (int k = 0; k < 8; k++) w |= (uint64_t)blk[8 * i + k] << (8 * k);
        st[i] ^= w;
    }
    keccakf(st);
}

orc_transcript *orc_transcript_new(void) { return (orc_transcript *)calloc(1, sizeof(orc_transcript)); }
void orc_transcript_free(orc_transcript *t) { free(t); }
void orc_transcript_append(orc_transcript *t, const uint8_t *data, size_t n) {
    while (n) {
        size_t take = 136 - t->pos;
        if (take > n) take = n;
        memcpy(t->buf + t->pos, data, take);
        t->pos += take; data += take; n -= take;
        if (t->pos == 136) { absorb_block(t->st, t->buf); t->pos = 0; }
    }
}
static void finalize_clone(const orc_transcript *t, uint8_t out[32]) {
    orc_transcript c = *t;
    memset(c.buf + c.pos, 0, 136 - c.pos);
    c.buf[c.pos] ^= 0x01;
    c.buf[135] ^= 0x80;
    absorb_block(c.st, c.buf);
    for (int i = 0; i < 4; i++)
        for (int k = 0; k < 8; k++) out[8 * i + k] = (uint8_t)(c.st[i] >> (8 * k));
}
/* fiat_shamir_transcript.rs:29-36: finalize a CLONE, absorb the digest back, running state kept */
void orc_transcript_sample(orc_transcript *t, uint8_t out[32]) {
    finalize_clone(t, out);
    orc_transcript_append(t, out, 32);
}
int orc_transcript_challenge(orc_transcript *t, int field, uint64_t *out) {
    GETF;
    uint8_t d[32];
    orc_transcript_sample(t, d);
    fe o;
    fe_from_le_bytes(F, &o, d, 32);
    fe_store(F, out, &o);
    return ORC_OK;
}
void orc_keccak256(const uint8_t *data, size_t n, uint8_t out[32]) {
    orc_transcript t;
    memset(&t, 0, sizeof t);
    orc_transcript_append(&t, data, n);
    finalize_clone(&t, out);
}
