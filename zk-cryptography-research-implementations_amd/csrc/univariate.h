// univariate.h -- host-side dense univariate helpers for the sumcheck control path.
// Mirrors polynomials/src/univariate/dense_univariate.rs: evaluate :57-68, lagrange_interpolate :74-98
// (lagrange_basis :101-126).  O(d^2) on d+1 <= 4 points per round: negligible, host only.
#pragma once
#include <vector>

#include "ufield.cuh"

namespace zk {

template <class F> inline Fe<F> uni_evaluate(const std::vector<Fe<F>> &c, const Fe<F> &x) {
    Fe<F> result = fe_zero<F>(), power = fe_one<F>();
    for (const Fe<F> &coef : c) {
        result = fe_add<F>(result, fe_mul<F>(coef, power));
        power = fe_mul<F>(power, x);
    }
    return result;
}

// n points -> n coefficients (leading zeros kept: the numerator always has n terms)
template <class F> inline std::vector<Fe<F>> lagrange_interpolate(const std::vector<Fe<F>> &xs, const std::vector<Fe<F>> &ys) {
    size_t n = xs.size();
    std::vector<Fe<F>> out(n, fe_zero<F>());
    for (size_t i = 0; i < n; i++) {
        std::vector<Fe<F>> num{fe_one<F>()};
        for (size_t k = 0; k < n; k++) {
            if (fe_eq<F>(xs[k], xs[i])) continue;              // compares values, as the reference does (:112)
            std::vector<Fe<F>> nxt(num.size() + 1, fe_zero<F>());
            Fe<F> nx = fe_neg<F>(xs[k]);
            for (size_t d = 0; d < num.size(); d++) {          // times (x - x_k)
                nxt[d] = fe_add<F>(nxt[d], fe_mul<F>(num[d], nx));
                nxt[d + 1] = fe_add<F>(nxt[d + 1], num[d]);
            }
            num.swap(nxt);
        }
        Fe<F> scale = fe_mul<F>(ys[i], fe_inv<F>(uni_evaluate<F>(num, xs[i])));
        for (size_t d = 0; d < num.size(); d++) out[d] = fe_add<F>(out[d], fe_mul<F>(scale, num[d]));
    }
    return out;
}

// The interpolation nodes of the sumcheck are always 0..d, so the d+1 basis polynomials l_i(x) are the same in
// every round: build them once per prover call (d+1 inversions) and turn each round's evaluations into
// coefficients with (d+1)^2 multiplications.  Same field elements as lagrange_interpolate (exact arithmetic).
template <class F> inline std::vector<std::vector<Fe<F>>> lagrange_basis_matrix(const std::vector<Fe<F>> &xs) {
    size_t n = xs.size();
    std::vector<std::vector<Fe<F>>> basis(n);
    for (size_t i = 0; i < n; i++) {
        std::vector<Fe<F>> ys(n, fe_zero<F>());
        ys[i] = fe_one<F>();
        basis[i] = lagrange_interpolate<F>(xs, ys);
    }
    return basis;
}
template <class F> inline std::vector<Fe<F>> interpolate_with_basis(const std::vector<std::vector<Fe<F>>> &basis, const std::vector<Fe<F>> &ys) {
    size_t n = ys.size();
    std::vector<Fe<F>> out(n, fe_zero<F>());
    for (size_t i = 0; i < n; i++)
        for (size_t d = 0; d < n; d++) out[d] = fe_add<F>(out[d], fe_mul<F>(ys[i], basis[i][d]));
    return out;
}

}  // namespace zk
