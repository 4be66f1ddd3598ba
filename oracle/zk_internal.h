/* zk_internal.h -- internals shared by the oracle's C files (test infrastructure only). */
#ifndef ZK_INTERNAL_H
#define ZK_INTERNAL_H
#include "zkoracle.h"
#include <stdlib.h>
#include <string.h>

#define MAXL 6
typedef unsigned __int128 u128;
typedef struct { uint64_t l[MAXL]; } fe;

typedef struct {
    int n;              /* limbs */
    uint64_t p[MAXL];   /* modulus */
    uint64_t r[MAXL];   /* R mod p  (Montgomery one) */
    uint64_t r2[MAXL];  /* R^2 mod p */
    uint64_t inv;       /* -p^-1 mod 2^64 */
    int ready;
} field_t;

const field_t *orc_fld(int id);

/* element load/store from packed arrays (n limbs per element) */
static inline void fe_load(const field_t *F, fe *o, const uint64_t *src) {
    memset(o, 0, sizeof *o);
    memcpy(o->l, src, 8 * (size_t)F->n);
}
static inline void fe_store(const field_t *F, uint64_t *dst, const fe *a) {
    memcpy(dst, a->l, 8 * (size_t)F->n);
}
static inline int fe_eq(const field_t *F, const fe *a, const fe *b) {
    return memcmp(a->l, b->l, 8 * (size_t)F->n) == 0;
}
static inline int fe_is_zero(const field_t *F, const fe *a) {
    uint64_t x = 0;
    for (int i = 0; i < F->n; i++) x |= a->l[i];
    return x == 0;
}
void fe_zero(fe *o);
void fe_one(const field_t *F, fe *o);
void fe_add(const field_t *F, fe *o, const fe *a, const fe *b);
void fe_sub(const field_t *F, fe *o, const fe *a, const fe *b);
void fe_neg(const field_t *F, fe *o, const fe *a);
void fe_mul(const field_t *F, fe *o, const fe *a, const fe *b);
void fe_inv(const field_t *F, fe *o, const fe *a);
void fe_from_u64(const field_t *F, fe *o, uint64_t v);
void fe_to_canonical(const field_t *F, uint64_t *out, const fe *a);    /* into_bigint() */
void fe_from_canonical(const field_t *F, fe *o, const uint64_t *canon); /* canon < p */
void fe_from_le_bytes(const field_t *F, fe *o, const uint8_t *b, size_t n);
void fe_to_be_bytes(const field_t *F, uint8_t *out, const fe *a);
void fe_to_le_bytes(const field_t *F, uint8_t *out, const fe *a);

struct orc_transcript {
    uint64_t st[25];
    uint8_t buf[136];
    size_t pos;
};

/* MLE on fe-arrays (mle.c) */
int mle_partial_evaluate(const field_t *F, const uint64_t *poly, size_t len, size_t var,
                         const fe *value, uint64_t *out);

#endif
