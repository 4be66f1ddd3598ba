// zkmle_gkr.hip -- C ABI: layered circuit, dense wiring predicates and the GKR prover / verifier.
// Host mirror of circuit/src/arithmetic_circuit.rs, gkr/src/utils.rs and gkr/src/gkr_protocol.rs;
// every table-sized step (wiring-predicate folds, alpha/beta combination, outer sum / product of W,
// the sumcheck itself, W evaluations) runs on the GPU through the table API.
#include <string.h>

#include <stdlib.h>

#include <algorithm>
#include <array>
#include <memory>
#include <vector>

#include "context.h"
#include "sumcheck_kernels.cuh"
#include "transcript.h"

using namespace zk;

namespace zk {
int circuit_evaluate_device(int field, const zk_gate *gates, const size_t *gate_counts, size_t nlayers, const size_t *widths,
                            const uint64_t *inputs, uint64_t *evals);
}

namespace {

struct TableDeleter { void operator()(zk_table *t) const { zk_table_free(t); } };
using TablePtr = std::unique_ptr<zk_table, TableDeleter>;

template <class F> Fe<F> load_el(const uint64_t *src) { Fe<F> e; memcpy(e.l, src, 4 * F::N); return e; }
template <class F> void store_el(uint64_t *dst, const Fe<F> &e) { memcpy(dst, e.l, 4 * F::N); }

int alloc_table(int field, size_t len, TablePtr &out) {
    zk_table *t = nullptr;
    ZK_TRY(table_alloc_pooled(field, len, &t));
    out.reset(t);
    return ZK_OK;
}
int upload_table(int field, const uint64_t *host, size_t len, TablePtr &out) {
    zk_table *t = nullptr;
    ZK_TRY(zk_table_upload(field, host, len, &t));
    out.reset(t);
    return ZK_OK;
}

// format!("{:0>width$b}") arithmetic_circuit.rs:198-200 : at least `width` digits, never truncated
size_t padded_bits(size_t v, size_t width) {
    size_t nb = 1;
    while ((v >> nb) != 0) nb++;
    return nb > width ? nb : width;
}

size_t layer_len(const zk_gate *g, size_t n) {   // :73-80 max output index + 1 (0 gates -> 1)
    size_t mx = 0;
    for (size_t i = 0; i < n; i++) if (g[i].out > mx) mx = g[i].out;
    return mx + 1;
}

// fold `src` by variable 0 with vals[0..k) in turn (utils.rs:38-56 chains); result in `out`
int fold_chain(const zk_table *src, const uint64_t *vals, size_t k, int limbs, TablePtr &out) {
    const zk_table *cur = src;
    TablePtr a, b;
    for (size_t i = 0; i < k; i++) {
        if (cur->len < 2) return ZK_E_NOT_POW2;
        TablePtr nx;
        ZK_TRY(alloc_table(src->field, cur->len / 2, nx));
        ZK_TRY(zk_mle_fold(cur, 0, vals + i * limbs, nx.get(), nullptr));
        a = std::move(b);
        b = std::move(nx);
        cur = b.get();
    }
    if (k == 0) {
        zk_table *c = nullptr;
        ZK_TRY(zk_table_clone(src, &c));
        out.reset(c);
    } else {
        out = std::move(b);
    }
    return ZK_OK;
}

// compute_new_add_i_mul_i utils.rs:23-68 for one predicate: alpha * fold(x, rb) + beta * fold(x, rc)
int alpha_beta_fold(const zk_table *abc, const uint64_t *alpha, const uint64_t *beta, const uint64_t *rb, const uint64_t *rc,
                    size_t k, int limbs, TablePtr &out) {
    if (k == 0) return ZK_E_RANGE;                       // rb_values[0] :38
    if (k <= 8 && (abc->len >> k) != 0) {                // both chains and the combination as one weighted fold (mle_kernels.cuh fold_alpha_beta_kernel)
        ZK_TRY(alloc_table(abc->field, abc->len >> k, out));
        return mle_fold_alpha_beta(abc, k, alpha, beta, rb, rc, out.get());
    }
    TablePtr frb, frc;
    ZK_TRY(fold_chain(abc, rb, k, limbs, frb));
    ZK_TRY(fold_chain(abc, rc, k, limbs, frc));
    ZK_TRY(zk_mle_scalar_mul(frb.get(), alpha, frb.get(), nullptr));      // :58-59 (in place: element-wise)
    ZK_TRY(zk_mle_scalar_mul(frc.get(), beta, frc.get(), nullptr));
    ZK_TRY(alloc_table(abc->field, frb->len, out));
    ZK_TRY(zk_mle_add(frb.get(), frc.get(), out.get(), nullptr));
    return ZK_OK;
}

struct CircuitEval {
    std::vector<size_t> goff, eoff, lsz;
    std::vector<uint64_t> evals;
};

template <class F> int circuit_evaluate(const zk_gate *gates, const size_t *gate_counts, size_t nlayers, const uint64_t *inputs,
                                        size_t ninputs, CircuitEval &ce) {
    const size_t L64 = F::N / 2;
    ce.goff.assign(nlayers + 1, 0);
    ce.eoff.assign(nlayers + 2, 0);
    ce.lsz.assign(nlayers + 1, 0);
    for (size_t l = 0; l < nlayers; l++) ce.goff[l + 1] = ce.goff[l] + gate_counts[l];
    for (size_t l = 0; l < nlayers; l++) ce.lsz[l] = layer_len(gates + ce.goff[l], gate_counts[l]);   // :73-80
    ce.lsz[nlayers] = ninputs;
    for (size_t l = 0; l <= nlayers; l++) ce.eoff[l + 1] = ce.eoff[l] + ce.lsz[l];
    ce.evals.assign(ce.eoff[nlayers + 1] * L64, 0);
    // one kernel launch per layer, lane per output wire (gates grouped by output: the += of :96)
    return circuit_evaluate_device(F::ID, gates, gate_counts, nlayers, ce.lsz.data(), inputs, ce.evals.data());
}

// positions of a layer's gates in its dense predicates (:139-155), appended to `pos`: the add gates' first, then the mul gates'
int layer_positions(const zk_gate *g, size_t ngates, size_t layer_index, std::vector<uint64_t> &pos, size_t *na, size_t *nm) {
    const size_t n = (size_t)1 << zk_num_of_layer_variables(layer_index);
    std::vector<uint64_t> pm;
    const size_t before = pos.size();
    for (size_t k = 0; k < ngates; k++) {
        size_t at = zk_wiring_index(layer_index, g[k].out, g[k].left, g[k].right);
        if (at >= n) return ZK_E_RANGE;
        (g[k].op == 0 ? pos : pm).push_back(at);
    }
    *na = pos.size() - before;
    *nm = pm.size();
    pos.insert(pos.end(), pm.begin(), pm.end());
    return ZK_OK;
}
// add_i / mul_i from positions that are already on the device (na add positions, then nm mul positions): two fills and two scatters, nothing to wait for
template <class F> int add_mul_mle_dev(const uint64_t *dpos, size_t na, size_t nm, size_t layer_index, TablePtr &add_i, TablePtr &mul_i) {
    size_t n = (size_t)1 << zk_num_of_layer_variables(layer_index);
    ZK_TRY(alloc_table(F::ID, n, add_i));
    ZK_TRY(alloc_table(F::ID, n, mul_i));
    ZK_HIP(hipMemsetAsync(add_i->dptr, 0, n * 4 * F::N, cur_stream()));                     // vec![F::zero(); 2^vars] :133-134
    ZK_HIP(hipMemsetAsync(mul_i->dptr, 0, n * 4 * F::N, cur_stream()));
    if (na) scatter_one_kernel<F><<<grid_for(na), kBlock, 0, cur_stream()>>>(add_i->dptr, dpos, na);
    if (nm) scatter_one_kernel<F><<<grid_for(nm), kBlock, 0, cur_stream()>>>(mul_i->dptr, dpos + na, nm);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}
template <class F> int add_mul_mle(const zk_gate *g, size_t ngates, size_t layer_index, TablePtr &add_i, TablePtr &mul_i) {
    std::vector<uint64_t> pos;
    size_t na = 0, nm = 0;
    ZK_TRY(layer_positions(g, ngates, layer_index, pos, &na, &nm));
    void *dpos = nullptr;
    if (!pos.empty()) {
        ZK_TRY(scratch(pos.size() * 8, &dpos));
        ZK_HIP(zk::memcpy_on_stream(dpos, pos.data(), pos.size() * 8, hipMemcpyHostToDevice));   // returns when the copy has been made
    }
    return add_mul_mle_dev<F>((const uint64_t *)dpos, na, nm, layer_index, add_i, mul_i);
}

// what a proof uploads ONCE instead of once per layer: every layer's wiring positions and every layer's evaluations
struct PooledBlock {
    void *p = nullptr;
    ~PooledBlock() { pool_free(p); }
    int alloc(size_t bytes) { return pool_alloc(bytes ? bytes : 8, &p); }
};
struct LayerPositions {
    PooledBlock dev;
    std::vector<size_t> off, na, nm;
    const uint64_t *at(size_t L) const { return (const uint64_t *)dev.p + off[L]; }
};
int upload_positions(const zk_gate *gates, const size_t *gate_counts, size_t nlayers, LayerPositions &lp) {
    std::vector<uint64_t> pos;
    lp.off.assign(nlayers, 0); lp.na.assign(nlayers, 0); lp.nm.assign(nlayers, 0);
    size_t goff = 0;
    for (size_t L = 0; L < nlayers; L++) {
        lp.off[L] = pos.size();
        ZK_TRY(layer_positions(gates + goff, gate_counts[L], L, pos, &lp.na[L], &lp.nm[L]));
        goff += gate_counts[L];
    }
    ZK_TRY(lp.dev.alloc(pos.size() * 8));
    if (!pos.empty()) ZK_HIP(zk::memcpy_on_stream(lp.dev.p, pos.data(), pos.size() * 8, hipMemcpyHostToDevice));
    return ZK_OK;
}

// compute_fbc_polynomial utils.rs:8-21 : [add_i_bc, W(b)+W(c), mul_i_bc, W(b)*W(c)]
int build_fbc(TablePtr &add_bc, TablePtr &mul_bc, const zk_table *w, TablePtr &add_w, TablePtr &mul_w) {
    if (w->len * w->len != add_bc->len) return ZK_E_NVARS;                            // ProductPolynomial::new product_polynomial.rs:16-21
    ZK_TRY(alloc_table(w->field, w->len * w->len, add_w));
    ZK_TRY(alloc_table(w->field, w->len * w->len, mul_w));
    ZK_TRY(zk_mle_tensor_add(w, w, add_w.get(), nullptr));
    ZK_TRY(zk_mle_tensor_mul(w, w, mul_w.get(), nullptr));
    return ZK_OK;
}

template <class F> int derive_bc(const zk_gate *layer_gates, size_t ngates, size_t L, const uint64_t *ra, const uint64_t *alpha,
                                 const uint64_t *beta, const uint64_t *rb, const uint64_t *rc, size_t nr, TablePtr &add_bc,
                                 TablePtr &mul_bc, const LayerPositions *lp = nullptr) {
    const int limbs = F::N / 2;
    TablePtr add_abc, mul_abc;
    if (lp) ZK_TRY((add_mul_mle_dev<F>(lp->at(L), lp->na[L], lp->nm[L], L, add_abc, mul_abc)));   // gkr_protocol.rs:58, positions uploaded with the other layers'
    else ZK_TRY((add_mul_mle<F>(layer_gates, ngates, L, add_abc, mul_abc)));
    if (L == 0) {                                                                     // :60-72
        ZK_TRY(alloc_table(F::ID, add_abc->len / 2, add_bc));
        ZK_TRY(alloc_table(F::ID, mul_abc->len / 2, mul_bc));
        ZK_TRY(zk_mle_fold(add_abc.get(), 0, ra, add_bc.get(), nullptr));
        ZK_TRY(zk_mle_fold(mul_abc.get(), 0, ra, mul_bc.get(), nullptr));
    } else {                                                                          // :73-82
        ZK_TRY(alpha_beta_fold(add_abc.get(), alpha, beta, rb, rc, nr, limbs, add_bc));
        ZK_TRY(alpha_beta_fold(mul_abc.get(), alpha, beta, rb, rc, nr, limbs, mul_bc));
    }
    return ZK_OK;
}

template <class F> int gkr_prove(const zk_gate *gates, const size_t *gate_counts, size_t nlayers, const uint64_t *inputs, size_t ninputs,
                                 uint64_t *circuit_output, size_t *output_len, uint64_t *claimed_sum, uint64_t *layer_claims,
                                 uint64_t *coeffs, uint64_t *challenges, uint64_t *wb_evals, uint64_t *wc_evals) {
    const size_t L64 = F::N / 2;
    CircuitEval ce;
    ZK_TRY((circuit_evaluate<F>(gates, gate_counts, nlayers, inputs, ninputs, ce)));  // :27
    *output_len = ce.lsz[0];
    memcpy(circuit_output, ce.evals.data(), ce.lsz[0] * L64 * 8);
    zk_transcript tr;
    // w0, padded [x] -> [x, 0]  :39-47
    std::vector<uint64_t> w0(ce.evals.begin(), ce.evals.begin() + ce.lsz[0] * L64);
    if (ce.lsz[0] == 1) w0.resize(2 * L64, 0);
    size_t w0len = w0.size() / L64;
    TablePtr w0t;
    ZK_TRY(upload_table(F::ID, w0.data(), w0len, w0t));                               // new(): pow2 assert
    ZK_TRY(transcript_absorb_table(tr.t, w0t.get()));                                // :49
    uint64_t ra[6], claim[6], alpha[6] = {0}, beta[6] = {0};
    store_el<F>(ra, tr.t.random_challenge_as_field_element<F>());                    // :50
    ZK_TRY(zk_mle_evaluate(w0t.get(), ra, 1, claim));                                // :51
    std::vector<uint64_t> rb, rc;
    size_t nr = 0, coff = 0, choff = 0;
    // one upload each for what every layer needs from the host: the wiring positions and the layer evaluations (w_i are views into the block)
    LayerPositions lp;
    ZK_TRY(upload_positions(gates, gate_counts, nlayers, lp));
    PooledBlock wdev;
    ZK_TRY(wdev.alloc(ce.evals.size() * 8));
    ZK_HIP(zk::memcpy_on_stream(wdev.p, ce.evals.data(), ce.evals.size() * 8, hipMemcpyHostToDevice));
    for (size_t L = 0; L < nlayers; L++) {                                           // :57
        TablePtr add_bc, mul_bc, add_w, mul_w;
        ZK_TRY((derive_bc<F>(gates + ce.goff[L], gate_counts[L], L, ra, alpha, beta, rb.data(), rc.data(), nr, add_bc, mul_bc, &lp)));
        if (!is_pow2(ce.lsz[L + 1])) return ZK_E_NOT_POW2;                           // :88-89 w_i_polynomial: new() asserts a power of two
        zk_table wview{F::ID, ce.lsz[L + 1], (char *)wdev.p + ce.eoff[L + 1] * L64 * 8, 0};
        zk_table *const w_ptr = &wview;
        ZK_TRY(build_fbc(add_bc, mul_bc, w_ptr, add_w, mul_w));                      // :95
        size_t bclen = add_bc->len, rounds = ilog2(bclen);
        memcpy(layer_claims + L * L64, claim, L64 * 8);
        const zk_table *tabs[4] = {add_bc.get(), add_w.get(), mul_bc.get(), mul_w.get()};
        uint64_t *lco = coeffs + coff * L64, *lch = challenges + choff * L64;
        ZK_TRY(zk_sumcheck_gkr_prove(tabs, 2, 2, claim, &tr, lco, lch));             // :99
        if (L + 1 < nlayers) {                                                       // :109
            size_t mid = rounds / 2;                                                 // :120 / utils.rs:75
            uint64_t wbe[6], wce[6];
            ZK_TRY(zk_mle_evaluate(w_ptr, lch, mid, wbe));                           // utils.rs:78
            ZK_TRY(zk_mle_evaluate(w_ptr, lch + mid * L64, rounds - mid, wce));      // :79
            memcpy(wb_evals + L * L64, wbe, L64 * 8);                                // :116-117
            memcpy(wc_evals + L * L64, wce, L64 * 8);
            rb.assign(lch, lch + mid * L64);                                         // :121-123
            rc.assign(lch + mid * L64, lch + rounds * L64);
            nr = mid;
            tr.t.append_be<F>(load_el<F>(wbe));                                      // :125
            Fe<F> a = tr.t.random_challenge_as_field_element<F>();
            tr.t.append_be<F>(load_el<F>(wce));                                      // :128
            Fe<F> b = tr.t.random_challenge_as_field_element<F>();
            store_el<F>(alpha, a);
            store_el<F>(beta, b);
            store_el<F>(claim, fe_add<F>(fe_mul<F>(a, load_el<F>(wbe)), fe_mul<F>(b, load_el<F>(wce))));   // :132
        }
        coff += rounds * 3;
        choff += rounds;
    }
    memcpy(claimed_sum, claim, L64 * 8);
    return ZK_OK;
}

template <class F> int gkr_verify(const zk_gate *gates, const size_t *gate_counts, size_t nlayers, const uint64_t *inputs, size_t ninputs,
                                  const uint64_t *circuit_output, size_t output_len, const uint64_t *layer_claims, const uint64_t *coeffs,
                                  const uint64_t *wb_evals, const uint64_t *wc_evals, int *ok, uint64_t *last_challenges = nullptr) {
    // inputs == nullptr: verify_succinct (succinct_gkr_protocol.rs:172-285) -- the input layer is not evaluated here; its two
    // openings are checked by the caller against the commitment, at the last layer's challenges returned in last_challenges
    const size_t L64 = F::N / 2;
    const bool succinct = inputs == nullptr;
    *ok = 0;
    zk_transcript tr;
    std::vector<uint64_t> w0(circuit_output, circuit_output + output_len * L64);      // :153-159
    if (output_len == 1) w0.resize(2 * L64, 0);
    TablePtr w0t;
    ZK_TRY(upload_table(F::ID, w0.data(), w0.size() / L64, w0t));
    ZK_TRY(transcript_absorb_table(tr.t, w0t.get()));                                // :161
    uint64_t ra[6], claim[6], alpha[6] = {0}, beta[6] = {0};
    store_el<F>(ra, tr.t.random_challenge_as_field_element<F>());                    // :162
    ZK_TRY(zk_mle_evaluate(w0t.get(), ra, 1, claim));                                // :164
    std::vector<uint64_t> prev;
    size_t goff = 0, coff = 0;
    for (size_t L = 0; L < nlayers; L++) {                                           // :166
        if (memcmp(claim, layer_claims + L * L64, L64 * 8) != 0) return ZK_OK;       // :167-169
        size_t rounds = zk_gkr_rounds(L), mid = rounds / 2;
        std::vector<uint64_t> ch(rounds * L64);
        uint64_t last[6];
        int sok = 0;
        ZK_TRY(zk_sumcheck_gkr_verify(F::ID, layer_claims + L * L64, coeffs + coff * L64, rounds, 3, &tr, ch.data(), last, &sok));   // :172
        if (!sok) return ZK_OK;                                                      // :174-176
        uint64_t wbe[6], wce[6];
        if (L + 1 < nlayers) {                                                       // :183-187
            memcpy(wbe, wb_evals + L * L64, L64 * 8);
            memcpy(wce, wc_evals + L * L64, L64 * 8);
        } else if (succinct) {                                                       // succinct_gkr_protocol.rs:214-215: zero
            memset(wbe, 0, sizeof wbe);
            memset(wce, 0, sizeof wce);
            if (last_challenges) memcpy(last_challenges, ch.data(), rounds * L64 * 8);
        } else {                                                                     // :188-194 the verifier's own inputs
            TablePtr in;
            ZK_TRY(upload_table(F::ID, inputs, ninputs, in));
            ZK_TRY(zk_mle_evaluate(in.get(), ch.data(), mid, wbe));
            ZK_TRY(zk_mle_evaluate(in.get(), ch.data() + mid * L64, rounds - mid, wce));
        }
        Fe<F> wb = load_el<F>(wbe), wc = load_el<F>(wce);
        if (!(succinct && L + 1 == nlayers)) {                                       // the succinct verifier skips this for the input layer (:217)
            TablePtr add_bc, mul_bc;
            size_t k = prev.size() / L64 / 2;
            ZK_TRY((derive_bc<F>(gates + goff, gate_counts[L], L, ra, alpha, beta, prev.data(), prev.data() + k * L64, k, add_bc, mul_bc)));   // utils.rs:84-135
            uint64_t ar[6], mr[6];
            ZK_TRY(zk_mle_evaluate(add_bc.get(), ch.data(), rounds, ar));
            ZK_TRY(zk_mle_evaluate(mul_bc.get(), ch.data(), rounds, mr));
            Fe<F> expect = fe_add<F>(fe_mul<F>(load_el<F>(ar), fe_add<F>(wb, wc)), fe_mul<F>(load_el<F>(mr), fe_mul<F>(wb, wc)));
            if (!fe_eq<F>(expect, load_el<F>(last))) return ZK_OK;                   // :221-223
        }
        prev = ch;                                                                   // :225
        tr.t.append_be<F>(wb);                                                       // :227
        Fe<F> a = tr.t.random_challenge_as_field_element<F>();
        tr.t.append_be<F>(wc);                                                       // :230
        Fe<F> b = tr.t.random_challenge_as_field_element<F>();
        store_el<F>(alpha, a);
        store_el<F>(beta, b);
        store_el<F>(claim, fe_add<F>(fe_mul<F>(a, wb), fe_mul<F>(b, wc)));           // :233
        goff += gate_counts[L];
        coff += rounds * 3;
    }
    *ok = 1;
    return ZK_OK;
}

bool dense_tables_forced() {
    const char *e = getenv("ZK_GKR_DENSE_TABLES");                   // read per call: tests switch it
    return e && e[0] == '1';
}
// Layer i (i >= 1) writes exactly the wires 0 .. 2^i - 1 and reads wires below 2^(i+1); layer 0 writes wire 0 (and maybe 1); 2^nlayers inputs; ops 0 / 1;
// no gate twice in a layer (the dense predicates are indicators -- a repeated gate is ONE entry there but two terms of a gate list).  Everything else
// (the shapes the reference panics on, repeated gates) stays with the dense path, which reproduces those outcomes.
bool reference_shaped(const zk_gate *gates, const size_t *gate_counts, size_t nlayers, size_t ninputs, size_t *lsz0) {
    if (nlayers == 0 || nlayers > 24 || ninputs != ((size_t)1 << nlayers)) return false;
    size_t off = 0;
    std::vector<std::array<uint64_t, 4>> seen;
    for (size_t L = 0; L < nlayers; L++) {
        const size_t n = gate_counts[L], wout = L == 0 ? 2 : (size_t)1 << L, win = (size_t)1 << (L + 1);
        if (n == 0) return false;
        size_t mx = 0;
        seen.clear();
        for (size_t k = 0; k < n; k++) {
            const zk_gate &g = gates[off + k];
            if (g.out >= wout || g.left >= win || g.right >= win || g.op > 1) return false;
            if (g.out > mx) mx = g.out;
            seen.push_back({g.op, g.out, g.left, g.right});
        }
        if (L >= 1 && mx + 1 != wout) return false;
        if (L == 0) *lsz0 = mx + 1;
        std::sort(seen.begin(), seen.end());
        if (std::adjacent_find(seen.begin(), seen.end()) != seen.end()) return false;
        off += n;
    }
    return true;
}
// the gate lists of the last circuit this thread proved, compiled (grouped by wire on the device): a prover calls prove() on one circuit many times
struct CompiledCache {
    std::vector<uint8_t> key;
    zk_sparse_circuit *c = nullptr;
    size_t lsz0 = 0;
    ~CompiledCache() { if (c) zk_sparse_circuit_free(c); }
};
// *out = the compiled gate lists of this circuit if it is well-formed (null otherwise: the caller takes the dense tables).  The shape test (a sort
// of each layer's gates) runs once per circuit: a call with the gates of the thread's last circuit is one memcmp.
int compiled_for(const zk_gate *gates, const size_t *gate_counts, size_t nlayers, size_t ninputs, const zk_sparse_circuit **out, size_t *lsz0) {
    static thread_local CompiledCache cache;
    *out = nullptr;
    if (nlayers == 0 || nlayers > 24 || ninputs != ((size_t)1 << nlayers)) return ZK_OK;
    int dev = 0;
    ZK_HIP(hipGetDevice(&dev));
    size_t total = 0;
    for (size_t L = 0; L < nlayers; L++) total += gate_counts[L];
    std::vector<uint8_t> key(sizeof(int) + (2 + nlayers) * sizeof(size_t) + total * sizeof(zk_gate));
    uint8_t *w = key.data();
    memcpy(w, &dev, sizeof dev); w += sizeof dev;
    memcpy(w, &nlayers, sizeof nlayers); w += sizeof nlayers;
    memcpy(w, &ninputs, sizeof ninputs); w += sizeof ninputs;
    memcpy(w, gate_counts, nlayers * sizeof(size_t)); w += nlayers * sizeof(size_t);
    memcpy(w, gates, total * sizeof(zk_gate));
    if (!cache.c || cache.key != key) {
        size_t l0 = 0;
        if (!reference_shaped(gates, gate_counts, nlayers, ninputs, &l0)) return ZK_OK;
        if (cache.c) { zk_sparse_circuit_free(cache.c); cache.c = nullptr; cache.key.clear(); }
        std::vector<uint32_t> out_bits(nlayers);
        for (size_t L = 0; L < nlayers; L++) out_bits[L] = L == 0 ? 1u : (uint32_t)L;      // arithmetic_circuit.rs:166-178
        ZK_TRY(zk_sparse_circuit_new(gates, gate_counts, nlayers, out_bits.data(), ninputs, &cache.c));
        cache.key = std::move(key);
        cache.lsz0 = l0;
    }
    *out = cache.c;
    *lsz0 = cache.lsz0;
    return ZK_OK;
}

}  // namespace

extern "C" {

size_t zk_num_of_layer_variables(size_t layer_index) {
    if (layer_index == 0) return 3;                       // arithmetic_circuit.rs:167-169
    return layer_index + 2 * (layer_index + 1);           // :171-177
}
size_t zk_wiring_index(size_t layer_index, size_t a, size_t b, size_t c) {
    size_t wb = padded_bits(b, layer_index + 1), wc = padded_bits(c, layer_index + 1);
    return (((a << wb) | b) << wc) | c;                   // a-digits ++ b-digits ++ c-digits, :186-195
}
size_t zk_gkr_rounds(size_t layer_index) { return 2 * (layer_index + 1); }
size_t zk_circuit_eval_size(const zk_gate *gates, const size_t *gate_counts, size_t nlayers, size_t ninputs) {
    size_t tot = ninputs, off = 0;
    for (size_t l = 0; l < nlayers; l++) { tot += layer_len(gates + off, gate_counts[l]); off += gate_counts[l]; }
    return tot;
}
int zk_circuit_evaluate(int field, const zk_gate *gates, const size_t *gate_counts, size_t nlayers, const uint64_t *inputs,
                        size_t ninputs, size_t *layer_sizes, uint64_t *evals) {
    if (!gates || !gate_counts || !inputs || !layer_sizes || !evals) return ZK_E_ARG;
    ZK_TRY(require_device());
    ZK_DISPATCH_FIELD(field, {
        CircuitEval ce;
        ZK_TRY((circuit_evaluate<F>(gates, gate_counts, nlayers, inputs, ninputs, ce)));
        for (size_t l = 0; l <= nlayers; l++) layer_sizes[l] = ce.lsz[l];
        memcpy(evals, ce.evals.data(), ce.evals.size() * 8);
    });
    return ZK_OK;
}
int zk_circuit_add_mul_mle(int field, const zk_gate *layer_gates, size_t ngates, size_t layer_index, zk_table **add_i, zk_table **mul_i) {
    if ((!layer_gates && ngates) || !add_i || !mul_i) return ZK_E_ARG;
    ZK_TRY(require_device());
    TablePtr a, m;
    ZK_DISPATCH_FIELD(field, ZK_TRY((add_mul_mle<F>(layer_gates, ngates, layer_index, a, m))));
    *add_i = a.release();
    *mul_i = m.release();
    return ZK_OK;
}
int zk_gkr_prove(int field, const zk_gate *gates, const size_t *gate_counts, size_t nlayers, const uint64_t *inputs, size_t ninputs,
                 uint64_t *circuit_output, size_t *output_len, uint64_t *claimed_sum, uint64_t *layer_claims, uint64_t *coeffs,
                 uint64_t *challenges, uint64_t *wb_evals, uint64_t *wc_evals) {
    if (!gates || !gate_counts || !inputs || !circuit_output || !output_len || !claimed_sum || !layer_claims || !coeffs || !challenges)
        return ZK_E_ARG;
    if (nlayers > 1 && (!wb_evals || !wc_evals)) return ZK_E_ARG;
    ZK_TRY(require_device());
    // A well-formed circuit of the reference's shape is proved from its gate lists (zkmle_gkr_sparse.hip): the same transcript, hence the same proof,
    // without the dense add_i / mul_i tables (2^(3 i + 2) entries for layer i) -- ZK_GKR_DENSE_TABLES=1 keeps the reference's representation.
    if (field_limbs64(field) > 0 && !dense_tables_forced()) {
        size_t lsz0 = 0;
        const zk_sparse_circuit *c = nullptr;
        ZK_TRY(compiled_for(gates, gate_counts, nlayers, ninputs, &c, &lsz0));
        if (c) {
            const size_t L64 = (size_t)field_limbs64(field);
            uint64_t out2[2 * 6], outch[6];
            ZK_TRY(zk_gkr_sparse_prove_compiled(field, c, inputs, ninputs, out2, claimed_sum, layer_claims, coeffs, challenges, wb_evals, wc_evals, outch, nullptr));
            memcpy(circuit_output, out2, lsz0 * L64 * 8);          // a one-wire output layer was padded to two (gkr_protocol.rs:43-47)
            *output_len = lsz0;
            return ZK_OK;
        }
    }
    ZK_DISPATCH_FIELD(field, return gkr_prove<F>(gates, gate_counts, nlayers, inputs, ninputs, circuit_output, output_len, claimed_sum,
                                                 layer_claims, coeffs, challenges, wb_evals, wc_evals));
    return ZK_OK;
}
int zk_gkr_verify(int field, const zk_gate *gates, const size_t *gate_counts, size_t nlayers, const uint64_t *inputs, size_t ninputs,
                  const uint64_t *circuit_output, size_t output_len, const uint64_t *layer_claims, const uint64_t *coeffs,
                  const uint64_t *wb_evals, const uint64_t *wc_evals, int *ok) {
    if (!gates || !gate_counts || !inputs || !circuit_output || !layer_claims || !coeffs || !ok) return ZK_E_ARG;
    ZK_TRY(require_device());
    ZK_DISPATCH_FIELD(field, return gkr_verify<F>(gates, gate_counts, nlayers, inputs, ninputs, circuit_output, output_len, layer_claims,
                                                  coeffs, wb_evals, wc_evals, ok));
    return ZK_OK;
}

// verify_succinct  gkr/src/succinct_gkr_protocol.rs:172-285 (BLS12-381 Fr): the GKR rounds as above without the input
// layer, then MultilinearKZG::verify of the two input openings at the last layer's challenges (:262-283)
int zk_gkr_verify_succinct(const zk_gate *gates, const size_t *gate_counts, size_t nlayers, const uint64_t *circuit_output, size_t output_len,
                           const uint64_t *layer_claims, const uint64_t *coeffs, const uint64_t *wb_evals, const uint64_t *wc_evals,
                           const uint64_t *commitment12, const uint64_t *rb_evaluation, const uint64_t *rb_proofs, size_t n_rb_proofs,
                           const uint64_t *rc_evaluation, const uint64_t *rc_proofs, size_t n_rc_proofs, const uint64_t *g2_powers, size_t ng2,
                           int *ok) {
    if (!gates || !gate_counts || !circuit_output || !layer_claims || !coeffs || !commitment12 || !rb_evaluation || !rc_evaluation || !ok ||
        nlayers == 0)
        return ZK_E_ARG;
    ZK_TRY(require_device());
    size_t rounds = zk_gkr_rounds(nlayers - 1), mid = rounds / 2;
    std::vector<uint64_t> ch(rounds * 4);
    ZK_TRY((gkr_verify<Fr381>(gates, gate_counts, nlayers, nullptr, 0, circuit_output, output_len, layer_claims, coeffs, wb_evals, wc_evals, ok,
                              ch.data())));
    if (!*ok) return ZK_OK;
    int okb = 0, okc = 0;
    ZK_TRY(zk_kzg_verify(commitment12, ch.data(), mid, rb_evaluation, rb_proofs, n_rb_proofs, g2_powers, ng2, &okb));                    // :266-271
    ZK_TRY(zk_kzg_verify(commitment12, ch.data() + mid * 4, rounds - mid, rc_evaluation, rc_proofs, n_rc_proofs, g2_powers, ng2, &okc));   // :272-277
    *ok = (okb && okc) ? 1 : 0;
    return ZK_OK;
}

}  // extern "C"
