// sumcheck_kernels.cuh -- HIP kernels for the degree-d sumcheck over a SumPolynomial
// (sum of NPROD products of NFAC multilinear tables), the GKR round polynomial.
//
// Reference (per round): generate_round_univariate sumcheck_gkr_protocol.rs:113-143 makes
// (NFAC+1) x NPROD x NFAC separate fold passes + element-wise product / sum passes, then
// SumPolynomial::partial_evaluate (sum_polynomial.rs:40-53) folds every table again.
// Here one kernel per round streams every table once:
//   round_evals_kernel        evals e(t) = sum_i sum_p prod_f X_t[p][f][i],  X_t = lo + t (hi - lo),
//                             t = 0..NFAC  (t-steps are additions only)
//   fold_round_evals_kernel   folds every table by the previous challenge AND produces the next
//                             round's e(t) from the folded values in the same pass.
// Algorithmic traffic of the fused kernel on NT = NPROD*NFAC tables of 2^m entries:
//   NT * 32 B * 2^m read + NT * 32 B * 2^(m-1) written; NT*2^(m-1) fold multiplications +
//   (NFAC+1)(NFAC-1) NPROD 2^(m-2) evaluation multiplications.
#pragma once
#include "mle_kernels.cuh"

namespace zk {

constexpr int kMaxProducts = 8;
constexpr int kMaxFactors = 3;

struct SumPolyTables {
    const void *in[kMaxProducts * kMaxFactors];   // [p * nfac + f]
    void *out[kMaxProducts * kMaxFactors];
};

// accumulate the NFAC+1 evaluation terms of one product at one pair index
template <class F, int NFAC>
__device__ __forceinline__ void accumulate_terms(const Fe<F> (&lo)[NFAC], const Fe<F> (&hi)[NFAC], Wide<F> (&acc)[NFAC + 1], int skip1 = 0) {
    Fe<F> v[NFAC], d[NFAC];
#pragma unroll
    for (int f = 0; f < NFAC; f++) {
        v[f] = lo[f];
        d[f] = fe_sub<F>(hi[f], lo[f]);
    }
#pragma unroll
    for (int t = 0; t <= NFAC; t++) {
        if (!(skip1 && t == 1)) {                            // the point 1 is derived from the running claim (dev_transcript.cuh)
            Fe<F> term = v[0];
#pragma unroll
            for (int f = 1; f < NFAC; f++) term = fe_mul<F>(term, v[f]);
            wide_add_fe<F>(acc[t], term);
        }
        if (t < NFAC) {
#pragma unroll
            for (int f = 0; f < NFAC; f++) v[f] = fe_add<F>(v[f], d[f]);   // X_{t+1} = X_t + (hi - lo)
        }
    }
}

template <class F, int NFAC>
__device__ __forceinline__ void write_partials(Wide<F> (&acc)[NFAC + 1], Wide<F> *sh, void *partials) {
    Fe<F> tot;
    if (block_reduce_wide<F, NFAC + 1>(acc, sh, tot)) fe_store<F>(partials, (size_t)threadIdx.x * gridDim.x + blockIdx.x, tot);
}

// tables of `2 * half` entries; partials[t * gridDim.x + block]
template <class F, int NFAC>
__global__ void __launch_bounds__(kBlock) round_evals_kernel(SumPolyTables tabs, int nprod, size_t half, void *__restrict__ partials) {
    __shared__ Wide<F> sh[(NFAC + 1) * kBlock / 64];
    Wide<F> acc[NFAC + 1];
#pragma unroll
    for (int t = 0; t <= NFAC; t++) acc[t] = wide_zero<F>();
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += stride) {
        for (int p = 0; p < nprod; p++) {
            Fe<F> lo[NFAC], hi[NFAC];
#pragma unroll
            for (int f = 0; f < NFAC; f++) {
                lo[f] = fe_load<F>(tabs.in[p * NFAC + f], i);
                hi[f] = fe_load<F>(tabs.in[p * NFAC + f], i + half);
            }
            accumulate_terms<F, NFAC>(lo, hi, acc);
        }
    }
    write_partials<F, NFAC>(acc, sh, partials);
}

// tables of 4q entries in, 2q out; lane i folds outputs i and i+q of every table, then uses them
// as the (lo, hi) pair of the NEXT round.
template <class F, int NFAC>
__global__ void __launch_bounds__(kBlock) fold_round_evals_kernel(SumPolyTables tabs, int nprod, size_t q, Fe<F> r, void *__restrict__ partials,
                                                                  const void *__restrict__ rp = nullptr, int skip1 = 0) {
    __shared__ Wide<F> sh[(NFAC + 1) * kBlock / 64];
    Wide<F> acc[NFAC + 1];
#pragma unroll
    for (int t = 0; t <= NFAC; t++) acc[t] = wide_zero<F>();
    size_t stride = (size_t)gridDim.x * blockDim.x;
    const Multiplier<F> mr(challenge_arg<F>(r, rp));
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < q; i += stride) {
        for (int p = 0; p < nprod; p++) {
            Fe<F> lo[NFAC], hi[NFAC];
#pragma unroll
            for (int f = 0; f < NFAC; f++) {
                const void *src = tabs.in[p * NFAC + f];
                void *dst = tabs.out[p * NFAC + f];
                Fe<F> a0 = fe_load<F>(src, i), a1 = fe_load<F>(src, i + q);
                Fe<F> b0 = fe_load<F>(src, i + 2 * q), b1 = fe_load<F>(src, i + 3 * q);
                lo[f] = fe_add<F>(a0, mr.times(fe_sub<F>(b0, a0)));
                hi[f] = fe_add<F>(a1, mr.times(fe_sub<F>(b1, a1)));
                fe_store<F>(dst, i, lo[f]);
                fe_store<F>(dst, i + q, hi[f]);
            }
            accumulate_terms<F, NFAC>(lo, hi, acc, skip1);
        }
    }
    write_partials<F, NFAC>(acc, sh, partials);
}

// element-wise reduce of a SumPolynomial to one table: out[i] = sum_p prod_f X[p][f][i]
// (add_polynomials_element_wise sum_polynomial.rs:57-76 over multiply_polynomials_element_wise
//  product_polynomial.rs:58-73)
template <class F>
__global__ void sumpoly_reduce_kernel(SumPolyTables tabs, int nprod, int nfac, size_t len, void *__restrict__ out) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < len; i += stride) {
        Fe<F> acc = fe_zero<F>();
        for (int p = 0; p < nprod; p++) {
            Fe<F> term = fe_load<F>(tabs.in[p * nfac], i);
            for (int f = 1; f < nfac; f++) term = fe_mul<F>(term, fe_load<F>(tabs.in[p * nfac + f], i));
            acc = fe_add<F>(acc, term);
        }
        fe_store<F>(out, i, acc);
    }
}

// dst[pos[k]] = one  (wiring predicates add_i / mul_i, arithmetic_circuit.rs:136-156)
template <class F>
__global__ void scatter_one_kernel(void *__restrict__ dst, const uint64_t *__restrict__ pos, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) fe_store<F>(dst, pos[i], fe_one<F>());
}

}  // namespace zk
