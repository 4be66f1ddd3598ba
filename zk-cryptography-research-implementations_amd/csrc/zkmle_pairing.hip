// zkmle_pairing.hip -- C ABI of the verifier-side pairing code (host only; csrc/pairing.h): G2 helpers, pairings, the G2 half
// of the trusted setup (trusted_setup.rs:62-72) and MultilinearKZG::verify (multilinear_kzg.rs:131-158).
#include <string.h>

#include <vector>

#include "context.h"
#include "pairing.h"

using namespace zk;
using namespace zk::pairing;

namespace {

FqE fq_load(const uint64_t *src) { FqE e; memcpy(e.l, src, 48); return e; }
void fq_store(uint64_t *dst, const FqE &e) { memcpy(dst, e.l, 48); }
G1Affine g1_load(const uint64_t *p12) { return G1Affine{fq_load(p12), fq_load(p12 + 6)}; }
G2Affine g2_load(const uint64_t *p24) {
    return G2Affine{Fq2{fq_load(p24), fq_load(p24 + 6)}, Fq2{fq_load(p24 + 12), fq_load(p24 + 18)}};
}
void g2_store(uint64_t *out24, const G2Affine &p) {
    fq_store(out24, p.x.c0); fq_store(out24 + 6, p.x.c1); fq_store(out24 + 12, p.y.c0); fq_store(out24 + 18, p.y.c1);
}
Fe<Fr381> fr_canonical(const uint64_t *s) {
    Fe<Fr381> e;
    memcpy(e.l, s, 32);
    return fe_to_canonical<Fr381>(e);                       // into_bigint()
}

}  // namespace

extern "C" {

int zk_g2_generator(uint64_t *out24) {
    if (!out24) return ZK_E_ARG;
    g2_store(out24, g2_generator());
    return ZK_OK;
}
int zk_g2_is_on_curve(const uint64_t *p24) {
    if (!p24) return ZK_E_ARG;
    return g2_on_curve(g2_load(p24)) ? 1 : 0;
}
int zk_g2_add(const uint64_t *p24, const uint64_t *q24, uint64_t *out24) {
    if (!p24 || !q24 || !out24) return ZK_E_ARG;
    g2_store(out24, g2j_to_affine(g2j_add(g2j_from_affine(g2_load(p24)), g2j_from_affine(g2_load(q24)))));
    return ZK_OK;
}
int zk_g2_mul_fr(const uint64_t *p24, const uint64_t *scalar_fr, uint64_t *out24) {
    if (!p24 || !scalar_fr || !out24) return ZK_E_ARG;
    Fe<Fr381> k = fr_canonical(scalar_fr);
    g2_store(out24, g2j_to_affine(g2_mul_canonical(g2_load(p24), k.l, 8)));
    return ZK_OK;
}

int zk_pairing(const uint64_t *g1_12, const uint64_t *g2_24, uint64_t *gt72) {
    if (!g1_12 || !g2_24 || !gt72) return ZK_E_ARG;
    Fq12 e = pairing_product({PairIn{g1_load(g1_12), g2_load(g2_24)}});
    for (int k = 0; k < 6; k++) {                           // coefficients of w^0 .. w^5, (c0, c1) each, Montgomery limbs
        const Fq2 &c = f12_coeff(e, k);
        fq_store(gt72 + 12 * k, c.c0);
        fq_store(gt72 + 12 * k + 6, c.c1);
    }
    return ZK_OK;
}
int zk_pairing_product_is_one(const uint64_t *g1s, const uint64_t *g2s, size_t n, int *ok) {
    if ((!g1s || !g2s) && n) return ZK_E_ARG;
    if (!ok) return ZK_E_ARG;
    std::vector<PairIn> in(n);
    for (size_t i = 0; i < n; i++) in[i] = PairIn{g1_load(g1s + 12 * i), g2_load(g2s + 24 * i)};
    *ok = f12_eq(pairing_product(in), f12_one()) ? 1 : 0;
    return ZK_OK;
}

// compute_g2_powers_of_tau  trusted_setup.rs:62-72 : out[i] = [tau_i] G2
int zk_kzg_setup_g2(const uint64_t *taus, size_t ntaus, uint64_t *out) {
    if (!taus || !out || ntaus == 0) return ZK_E_ARG;       // "requires at least one variable" :64
    const G2Affine g2 = g2_generator();
    for (size_t i = 0; i < ntaus; i++) {
        Fe<Fr381> k = fr_canonical(taus + 4 * i);
        g2_store(out + 24 * i, g2j_to_affine(g2_mul_canonical(g2, k.l, 8)));
    }
    return ZK_OK;
}

// MultilinearKZG::verify  multilinear_kzg.rs:131-158
//   e(C - [v] G1, G2) == prod_i e(pi_i, [tau_i] G2 - [x_i] G2)   <=>   e(C - [v] G1, G2) * prod_i e(-pi_i, ...) == 1
int zk_kzg_verify(const uint64_t *commitment12, const uint64_t *opening_values, size_t nopen, const uint64_t *evaluation,
                  const uint64_t *proofs, size_t nproofs, const uint64_t *g2_powers, size_t ng2, int *ok) {
    if (!commitment12 || !evaluation || !ok || (!opening_values && nopen) || (!proofs && nproofs) || (!g2_powers && ng2)) return ZK_E_ARG;
    if (nopen != nproofs) return ZK_E_KZG_LEN;              // :137-141
    if (ng2 > nproofs) return ZK_E_KZG_LEN;                 // proofs[i] / opening_values[i] out of range in the loop :149-154
    const G1Affine g1 = g1_generator();
    const G2Affine g2 = g2_generator();
    std::vector<PairIn> in;
    {   // :144-146
        Fe<Fr381> v = fr_canonical(evaluation);
        G1Affine vg = g1_to_affine(g1_mul_canonical(g1, v.l, 8));
        G1Affine lhs = g1_to_affine(g1_madd(g1_from_affine(g1_load(commitment12)), g1_neg(vg)));
        in.push_back(PairIn{lhs, g2});
    }
    for (size_t i = 0; i < ng2; i++) {                      // :149-154
        Fe<Fr381> x = fr_canonical(opening_values + 4 * i);
        G2Jac q = g2j_add(g2j_from_affine(g2_load(g2_powers + 24 * i)), g2j_from_affine(g2_neg(g2j_to_affine(g2_mul_canonical(g2, x.l, 8)))));
        in.push_back(PairIn{g1_neg(g1_load(proofs + 12 * i)), g2j_to_affine(q)});
    }
    *ok = f12_eq(pairing_product(in), f12_one()) ? 1 : 0;   // :156
    return ZK_OK;
}

}  // extern "C"
