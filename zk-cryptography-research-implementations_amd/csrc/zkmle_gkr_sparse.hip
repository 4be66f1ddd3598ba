// zkmle_gkr_sparse.hip -- C ABI: linear-time (sparse-wiring) GKR prover, the generalisation that BASELINE
// config 4 (depth-3 circuit, 2^22 gates per layer) needs.  See include/zkmle.h for the contract.
//
// Per layer the reference (gkr/src/gkr_protocol.rs:57-133) builds dense add_i / mul_i tables of 2^(3i+2)
// entries, folds them by ra (or alpha/beta-combines two fold chains, utils.rs:23-68), forms the dense
// f(b,c) = add(b,c)(W(b)+W(c)) + mul(b,c) W(b) W(c) over 2^(2i+2) entries and runs a 2(i+1)-round sumcheck.
// With w_g = eq(ra, out_g) (or alpha eq(rb', out_g) + beta eq(rc', out_g)) the same round polynomials are:
//   rounds over b:  sum_c f(b,c) = W(b) H1(b) + H0(b),
//                   H1(b) = sum_{g: left_g = b} w_g ([add] + [mul] W(right_g)),   H0(b) = sum_{g add: left_g = b} w_g W(right_g)
//   rounds over c (b fixed to rb*, u = W(rb*)):
//                   f(rb*, c) = A(c)(u + W(c)) + M(c) u W(c),   A / M(c) = sum_{g add / mul: right_g = c} w_g eq(rb*, left_g)
// Both phases are the product's generic sum-of-products sumcheck on 4 tables of 2^k entries
// ([W,H1],[H0,1] then [A,u+W],[M,uW]), so all table work reuses the fused HIP round kernels.
#include <string.h>

#include <memory>
#include <vector>

#include "context.h"
#include "eq_table.cuh"
#include "sumcheck_kernels.cuh"
#include "transcript.h"

using namespace zk;

namespace {

struct TableDeleter { void operator()(zk_table *t) const { zk_table_free(t); } };
using TablePtr = std::unique_ptr<zk_table, TableDeleter>;
struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { pool_free(p); }                     // per-call scratch from the caching pool (context.h)
    int alloc(size_t bytes) { return pool_alloc(bytes, &p); }
    int upload(const void *src, size_t bytes) { ZK_TRY(alloc(bytes)); if (bytes) ZK_HIP(zk::memcpy_on_stream(p, src, bytes, hipMemcpyHostToDevice)); return ZK_OK; }
};
template <class F> Fe<F> load_el(const uint64_t *src) { Fe<F> e; memcpy(e.l, src, 4 * F::N); return e; }
template <class F> void store_el(uint64_t *dst, const Fe<F> &e) { memcpy(dst, e.l, 4 * F::N); }

int alloc_table(int field, size_t len, TablePtr &out) {
    zk_table *t = nullptr;
    ZK_TRY(table_alloc_pooled(field, len, &t));
    out.reset(t);
    return ZK_OK;
}

// ---- kernels ------------------------------------------------------------------------------------------------
struct GateArrays {
    const uint32_t *out, *left, *right, *op;      // SoA on the device
};
// w[g] = eqA[out_g] + eqB[out_g] where the caller's tables already carry the weights: eqA = alpha eq(rb, .), eqB = beta eq(rc, .)
// (eq_table.cuh: the constant is folded into a 64-entry half table), eqB null on layer 0 (w[g] = eq(ra, out_g)).  No products here:
// one per gate costs ~40 us of whole-GPU VALU time at 2^22 gates (profiles/r2/gkr_round_kernel_variants.md).
// (`out` is the output index of gate i in whatever order the caller wants the weights: gate order, or one of the grouped orders)
template <class F> __global__ void gate_weights_kernel(const uint32_t *__restrict__ out, size_t n, const void *eqA, const void *eqB, void *__restrict__ w) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fe<F> v = fe_load<F>(eqA, out[i]);
    if (eqB) v = fe_add<F>(v, fe_load<F>(eqB, out[i]));
    fe_store<F>(w, i, v);
}
// H1 / H0 (phase 1) and C / A (phase 2): for every index b of the grouping variable, two sums over the gates grouped under b (start[b] ..
// start[b + 1], their other index and operation stored once more in grouped order and read sequentially; the weight (gate order) and
// the table entry at the other index are the two gathers per gate).
// Round 1 ran one lane per b over its own gates: a lane's gates are a serial chain of dependent gathers (~4 us each) and a wave lasts as long
// as its longest lane -- 5-6 gates with random wiring, where the mean is 1: 300 us for 2^22 gates, latency x divergence, nowhere near the
// ~150 us its traffic costs.  Now a workgroup owns a range of consecutive b, i.e. ONE contiguous range of the grouped gate list: one lane per GATE
// does the gathers and the product (every lane exactly one chain link, all in flight together) and leaves the gate's two contributions in
// LDS; then one lane per b adds up its run from LDS.  A workgroup of 256 lanes owns kPhaseGroups = 224 consecutive b: with one gate per b
// on average (random wiring: 224 +- 15 gates) their gates fit ONE pass of 256 gate lanes 98 % of the time; longer ranges (skewed circuits)
// take more passes of 256.  Same sums, other order.
constexpr int kPhaseGroups = kBlock - 32;
// Where a gate's weight w_g = alpha eq(rb, out_g) + beta eq(rc, out_g) comes from.  Small layers: the table w (gate order), indexed through the
// group's order list.  Layers of > kEqSmallBits output bits: straight from the HALF tables of the two eq tables (entry o = hi[o >> lbits] *
// lo[o & mask], eq_table.cuh; the constants ride on the high halves), indexed by the gate's output index stored in grouped order -- four
// 64 KB tables that live in L2 and two products per gate, instead of building two 2^out_bits tables, adding them up gate by gate into w
// (three passes over 128 MB at 2^22) and gathering w from HBM.
struct GateWeights {
    const void *w;                       // non-null: the table of weights in gate order
    const void *ah, *al, *bh, *bl;       // else: half tables of alpha eq(rb, .) and (bh non-null) beta eq(rc, .)
    unsigned lbits;
    const void *uah, *ual, *ubh, *ubl;   // the same half tables in the products' internal form (UHalves below); uah non-null: the kernels use these
};
// the table gathered at a gate's other index: an array (W in phase 1), or the half tables of an eq table (eq(rb*, .) in phase 2)
struct OtherTable {
    const void *tab;                     // non-null: the table itself
    const void *hi, *lo;                 // else its half tables
    unsigned lbits;
    const void *uhi, *ulo;               // the half tables in the internal form (with GateWeights::uah)
};
// ---- half tables in the products' internal form (r3) ------------------------------------------------------------------------------
// The table kernels are VALU-bound, and more than half of their VALU work was not multiply-adds but what surrounds a product on stored
// operands: 32-bit -> 29-bit limbs of one operand, digit extraction of the other, conditional subtraction and 29-bit -> 32-bit limbs of
// the result -- ~190 instructions around 162 multiply-adds, four products per gate.  The half tables are 2^11 entries each and are
// built once per layer, so they are converted once: L limbs of 29 bits per entry (ufield.cuh), and the Montgomery factors are chosen so that
// a gate's chain of products needs no conversion until its result is stored.  With R = 2^(32 N) (the stored form) and U = 2^(29 L):
//   weights:  ah' = limbs29(ah R), al' = al U  ->  umul2(ah', al', bh', bl') = (ah al + bh bl) R = the weight in the stored form, below 1.06 p;
//   other:    hi' = hi U, lo' = lo U            ->  umul(hi', lo') = hi lo U;   umul(w R, h U) = w h R, the stored form of the product;
//   or a stored table entry x R                 ->  umul_std(w R, x R) = w x R  (digits of x: the one conversion left).
// U / p = 2^6.1, so every one of these is below 1.5 p and one conditional subtraction gives the canonical limbs the old chain gave.
template <class F> constexpr int u_stride_words() { return (UParams<F>::L + 3) / 4 * 4; }          // 16-byte aligned entries
template <class F> __device__ __forceinline__ Ufe<F> ufe_load(const void *tab, uint32_t idx) {
    const uint32_t *p = reinterpret_cast<const uint32_t *>(tab) + (size_t)idx * u_stride_words<F>();
    Ufe<F> r;
#pragma unroll
    for (int j = 0; j < UParams<F>::L; j++) r.l[j] = p[j];
    return r;
}
// out[i] = in[i] as 29-bit limbs: the stored integer itself (mont = 0) or times U / R (mont = 1), fully reduced; two tables per launch
template <class F> __global__ void halves_to_u_kernel(const void *__restrict__ in0, uint32_t n0, int mont0, void *__restrict__ out0,
                                                      const void *__restrict__ in1, uint32_t n1, int mont1, void *__restrict__ out1) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool second = i >= n0;
    if (second) i -= n0;
    if (second && i >= n1) return;
    const Fe<F> x = fe_load<F>(second ? in1 : in0, i);
    const Ufe<F> u = (second ? mont1 : mont0) ? u_reduce_once<F>(u_from_std<F>(x)) : u_from_limbs32<F>(x);
    uint32_t *o = reinterpret_cast<uint32_t *>(second ? out1 : out0) + (size_t)i * u_stride_words<F>();
#pragma unroll
    for (int j = 0; j < u_stride_words<F>(); j++) o[j] = j < UParams<F>::L ? u.l[j] : 0u;
}
template <class F> int halves_to_u(const void *hi, int mont_hi, const void *lo, unsigned nbits, unsigned lbits, DevBuf &uhi, DevBuf &ulo) {
    const uint32_t nhi = 1u << (nbits - lbits), nlo = 1u << lbits;
    ZK_TRY(uhi.alloc((size_t)nhi * u_stride_words<F>() * 4));
    ZK_TRY(ulo.alloc((size_t)nlo * u_stride_words<F>() * 4));
    halves_to_u_kernel<F><<<(nhi + nlo + kBlock - 1) / kBlock, kBlock, 0, cur_stream()>>>(hi, nhi, mont_hi, uhi.p, lo, nlo, 1, ulo.p);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}
template <class F> __device__ __forceinline__ Ufe<F> gate_weight_u(const GateWeights &g, uint32_t idx) {        // the weight, stored form, as 29-bit limbs below 1.06 p
    const uint32_t h = idx >> g.lbits, l = idx & ((1u << g.lbits) - 1u);
    if (g.ubh) return umul2<F>(ufe_load<F>(g.uah, h), ufe_load<F>(g.ual, l), ufe_load<F>(g.ubh, h), ufe_load<F>(g.ubl, l));
    return umul<F>(ufe_load<F>(g.uah, h), ufe_load<F>(g.ual, l));
}
template <class F> __device__ __forceinline__ Ufe<F> other_times_u(const OtherTable &t, uint32_t idx, const Ufe<F> &wu) {   // w x entry, stored form, below 1.5 p
    if (t.tab) return umul_std<F>(wu, fe_load<F>(t.tab, idx));
    return umul<F>(wu, umul<F>(ufe_load<F>(t.uhi, idx >> t.lbits), ufe_load<F>(t.ulo, idx & ((1u << t.lbits) - 1u))));
}
template <class F> __device__ __forceinline__ Fe<F> other_entry(const OtherTable &t, uint32_t idx) {
    if (t.tab) return fe_load<F>(t.tab, idx);
    return fe_mul<F>(fe_load<F>(t.hi, idx >> t.lbits), fe_load<F>(t.lo, idx & ((1u << t.lbits) - 1u)));
}
template <class F> __device__ __forceinline__ Fe<F> gate_weight(const GateWeights &g, uint32_t idx) {
    if (g.w) return fe_load<F>(g.w, idx);
    const uint32_t h = idx >> g.lbits, l = idx & ((1u << g.lbits) - 1u);
    if (g.bh) return fe_mul2_u<F>(fe_load<F>(g.ah, h), fe_load<F>(g.al, l), fe_load<F>(g.bh, h), fe_load<F>(g.bl, l));   // one reduction for both products
    return fe_mul<F>(fe_load<F>(g.ah, h), fe_load<F>(g.al, l));
}
struct Phase1Op {            // per gate: w and t = w W[right]; add gate: H1 += w, H0 += t; mul gate: H1 += t
    static constexpr bool kNeedsWeight = true;
    template <class F> static __device__ __forceinline__ void terms(const Fe<F> &wg, const Fe<F> &t, uint32_t op, Fe<F> &x, Fe<F> &y) {
        if (op == 0) { x = wg; y = t; } else { x = t; y = fe_zero<F>(); }
    }
};
struct Phase2Op {            // per gate: t = w eqL[left]; add gate: A += t; mul gate: M += t
    static constexpr bool kNeedsWeight = false;
    template <class F> static __device__ __forceinline__ void terms(const Fe<F> &, const Fe<F> &t, uint32_t op, Fe<F> &x, Fe<F> &y) {
        if (op == 0) { x = t; y = fe_zero<F>(); } else { x = fe_zero<F>(); y = t; }
    }
};
template <class F, class Op, bool UH>
__device__ __forceinline__ void grouped_pair_sums(const uint32_t *__restrict__ start, size_t nb, const GateWeights &gw, const uint32_t *__restrict__ order,
                                                  const uint32_t *__restrict__ other, const uint32_t *__restrict__ op, const OtherTable &tab,
                                                  Fe<F> &sx, Fe<F> &sy) {
    __shared__ Fe<F> cx[kBlock], cy[kBlock];
    const unsigned tid = threadIdx.x;
    const size_t b0 = (size_t)blockIdx.x * kPhaseGroups, b = b0 + tid;
    const size_t nbk = nb - b0 < (size_t)kPhaseGroups ? nb - b0 : (size_t)kPhaseGroups;
    const bool mine = tid < nbk;                                                        // this lane owns index b
    const uint32_t e0 = start[b0], e1 = start[b0 + nbk];
    const uint32_t rs = mine ? start[b] : e1, re = mine ? start[b + 1] : e1;             // its run of the grouped list
    sx = fe_zero<F>();
    sy = fe_zero<F>();
    for (uint32_t base = e0; base < e1; base += kBlock) {
        const uint32_t e = base + tid;                                                  // one lane per gate
        if (e < e1) {
            const uint32_t oi = order[e], ti = other[e], pi = op[e];
            if constexpr (UH) {                                                         // half tables in the internal form: no conversions inside the chain
                const Ufe<F> wu = gate_weight_u<F>(gw, oi);
                const Fe<F> t = u_to_limbs32<F>(u_reduce_once<F>(other_times_u<F>(tab, ti, wu)));
                const Fe<F> wg = Op::kNeedsWeight ? u_to_limbs32<F>(u_reduce_once<F>(wu)) : fe_zero<F>();
                Op::template terms<F>(wg, t, pi, cx[tid], cy[tid]);
            } else {
                const Fe<F> x = other_entry<F>(tab, ti);
                const Fe<F> wg = gate_weight<F>(gw, oi);
                Op::template terms<F>(wg, fe_mul<F>(wg, x), pi, cx[tid], cy[tid]);
            }
        }
        __syncthreads();
        const uint32_t lo = rs > base ? rs : base, hi = re < base + kBlock ? re : base + (uint32_t)kBlock;
        for (uint32_t k = lo; k < hi; k++) {
            sx = fe_add<F>(sx, cx[k - base]);
            sy = fe_add<F>(sy, cy[k - base]);
        }
        __syncthreads();
    }
}
// `widx`: the group's order list (gw.w) or the gates' output indices in grouped order (half tables)
template <class F, bool UH> __global__ void __launch_bounds__(kBlock) phase1_tables_kernel(const uint32_t *__restrict__ start, size_t nb, GateWeights gw,
                                                        const uint32_t *__restrict__ widx, const uint32_t *__restrict__ right_l,
                                                        const uint32_t *__restrict__ op_l,
                                                        const void *__restrict__ W, void *__restrict__ H1, void *__restrict__ H0) {
    Fe<F> h1, h0;
    grouped_pair_sums<F, Phase1Op, UH>(start, nb, gw, widx, right_l, op_l, OtherTable{W, nullptr, nullptr, 0, nullptr, nullptr}, h1, h0);
    const size_t b = (size_t)blockIdx.x * kPhaseGroups + threadIdx.x;
    if (threadIdx.x >= (unsigned)kPhaseGroups || b >= nb) return;
    fe_store<F>(H1, b, h1);
    fe_store<F>(H0, b, h0);
}
// gates grouped by right index c: A(c) = sum of the add gates' w eqL[left], M(c) the same over the mul gates;
// stored: C = A + u M and A, so that phase 2 is the sumcheck of C(c) W(c) + u A(c)  (= A (u + W) + M u W)
template <class F, bool UH> __global__ void __launch_bounds__(kBlock) phase2_tables_kernel(const uint32_t *__restrict__ start, size_t nc, GateWeights gw,
                                                        const uint32_t *__restrict__ widx, const uint32_t *__restrict__ left_r,
                                                        const uint32_t *__restrict__ op_r,
                                                        OtherTable eqL, const void *__restrict__ u_dev, void *__restrict__ Cc, void *__restrict__ A) {
    Fe<F> a, m;
    grouped_pair_sums<F, Phase2Op, UH>(start, nc, gw, widx, left_r, op_r, eqL, a, m);
    const size_t c = (size_t)blockIdx.x * kPhaseGroups + threadIdx.x;
    if (threadIdx.x >= (unsigned)kPhaseGroups || c >= nc) return;
    const Fe<F> u = fe_load<F>(u_dev, 0);                      // W(rb*): a final value of phase 1, still on the device
    fe_store<F>(Cc, c, fe_add<F>(a, fe_mul<F>(u, m)));
    fe_store<F>(A, c, a);
}
// out[o] = sum over the gates with output o of op(in[left], in[right])   (arithmetic_circuit.rs:86-97, += semantics)
template <class F> __global__ void circuit_layer_kernel(GateArrays g, const uint32_t *__restrict__ order, const uint32_t *__restrict__ start,
                                                        size_t nout, const void *__restrict__ in, void *__restrict__ out) {
    size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= nout) return;
    Fe<F> acc = fe_zero<F>();
    for (uint32_t e = start[o]; e < start[o + 1]; e++) {
        uint32_t i = order[e];
        Fe<F> a = fe_load<F>(in, g.left[i]), b = fe_load<F>(in, g.right[i]);
        acc = fe_add<F>(acc, g.op[i] == 0 ? fe_add<F>(a, b) : fe_mul<F>(a, b));
    }
    fe_store<F>(out, o, acc);
}
// partial sums over gates of w_g eqL[left] eqR[right] split by gate type (verifier-side check)
template <class F> __global__ void wiring_eval_kernel(GateArrays g, size_t n, const void *w, const void *eqL, const void *eqR, void *partials) {
    __shared__ Wide<F> sh[2 * kBlock / 64];
    Wide<F> am[2] = {wide_zero<F>(), wide_zero<F>()};            // [0] add gates, [1] mul gates
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        Fe<F> t = fe_mul<F>(fe_mul<F>(fe_load<F>(w, i), fe_load<F>(eqL, g.left[i])), fe_load<F>(eqR, g.right[i]));
        if (g.op[i] == 0) wide_add_fe<F>(am[0], t); else wide_add_fe<F>(am[1], t);
    }
    Fe<F> tot;
    if (block_reduce_wide<F, 2>(am, sh, tot)) fe_store<F>(partials, (size_t)threadIdx.x * gridDim.x + blockIdx.x, tot);
}

// ---- host-side circuit structure ------------------------------------------------------------------------------
struct LayerDev {
    size_t ngates = 0;
    uint32_t out_bits = 0, in_bits = 0;
    DevBuf out, left, right, op;                       // u32[ngates]
    DevBuf ord_left, st_left, ord_right, st_right, ord_out, st_out;
    DevBuf l_right, l_op, r_left, r_op;                // right / op grouped by left index, left / op grouped by right index
    DevBuf l_out, r_out;                               // output index in the two grouped orders (gate weights from half tables)
    GateArrays arrays() const { return GateArrays{(const uint32_t *)out.p, (const uint32_t *)left.p, (const uint32_t *)right.p, (const uint32_t *)op.p}; }
};
// counting sort of gate ids by key: order[], start[nbins + 1]
void group_by(const std::vector<uint32_t> &key, size_t nbins, std::vector<uint32_t> &order, std::vector<uint32_t> &start) {
    start.assign(nbins + 1, 0);
    for (uint32_t k : key) start[k + 1]++;
    for (size_t b = 0; b < nbins; b++) start[b + 1] += start[b];
    order.resize(key.size());
    std::vector<uint32_t> cur(start.begin(), start.end() - 1);
    for (size_t i = 0; i < key.size(); i++) order[cur[key[i]]++] = (uint32_t)i;
}
// ---- circuit compile on the device: split the gate records, then three counting sorts (by left / right / output index) -----
// The order of the gates inside a group does not matter: the tables built from it are sums in the field, exact and commutative.
__global__ void split_gates_kernel(const zk_gate *__restrict__ g, size_t n, uint32_t out_bits, uint32_t in_bits, uint32_t *__restrict__ out,
                                   uint32_t *__restrict__ left, uint32_t *__restrict__ right, uint32_t *__restrict__ op, int *__restrict__ bad) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    zk_gate x = g[i];
    if ((x.out >> out_bits) || (x.left >> in_bits) || (x.right >> in_bits) || x.op > 1) { *bad = 1; x.out = x.left = x.right = x.op = 0; }
    out[i] = (uint32_t)x.out; left[i] = (uint32_t)x.left; right[i] = (uint32_t)x.right; op[i] = (uint32_t)x.op;
}
__global__ void key_hist_kernel(const uint32_t *__restrict__ key, size_t n, uint32_t *__restrict__ counts) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&counts[key[i]], 1u);
}
// exclusive scan of counts[0 .. nbins) in place, counts[nbins] = total.  Three launches over tiles of 4096 bins: tile totals,
// a one-workgroup scan of the totals, then the scan inside every tile.  (r1: the first version scanned everything in ONE
// workgroup, each lane walking its own contiguous run -- uncoalesced and serial: 6.4 ms for the 2^23 bins of a 2^22-gate
// layer, 83 % of the GPU time of a circuit compile.)
constexpr unsigned kScanThreads = 1024, kScanPer = 4, kScanTileBins = kScanThreads * kScanPer;
__device__ __forceinline__ uint32_t block_exclusive_scan_1024(uint32_t v, uint32_t *sh, uint32_t *total) {
    // sh: 17 words.  wave-level inclusive scan by shuffles, then the 16 wave totals by the first wave
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t o = __shfl_up(inc, off, 64);
        if (lane >= (unsigned)off) inc += o;
    }
    if (lane == 63) sh[wave] = inc;
    __syncthreads();
    if (threadIdx.x < 64) {
        uint32_t w = threadIdx.x < 16 ? sh[threadIdx.x] : 0u, winc = w;
#pragma unroll
        for (int off = 1; off < 16; off <<= 1) {
            uint32_t o = __shfl_up(winc, off, 64);
            if (threadIdx.x >= (unsigned)off) winc += o;
        }
        if (threadIdx.x < 16) sh[threadIdx.x] = winc - w;          // exclusive wave offsets
        if (threadIdx.x == 15) sh[16] = winc;
    }
    __syncthreads();
    uint32_t r = sh[wave] + inc - v;
    if (total) *total = sh[16];
    __syncthreads();
    return r;
}
__global__ void __launch_bounds__(1024) scan_tile_totals_kernel(const uint32_t *__restrict__ counts, size_t nbins, uint32_t *__restrict__ tile_sums) {
    __shared__ uint32_t sh[17];
    const size_t base = (size_t)blockIdx.x * kScanTileBins + (size_t)threadIdx.x * kScanPer;
    uint32_t sum = 0;
#pragma unroll
    for (unsigned k = 0; k < kScanPer; k++)
        if (base + k < nbins) sum += counts[base + k];
    uint32_t total;
    block_exclusive_scan_1024(sum, sh, &total);
    if (threadIdx.x == 0) tile_sums[blockIdx.x] = total;
}
__global__ void __launch_bounds__(1024) scan_tile_offsets_kernel(uint32_t *__restrict__ tile_sums, size_t ntiles, uint32_t *__restrict__ grand_total) {
    __shared__ uint32_t sh[17];
    uint32_t carry = 0;
    for (size_t t0 = 0; t0 < ntiles; t0 += kScanThreads) {         // one workgroup; ntiles is nbins / 4096
        const size_t i = t0 + threadIdx.x;
        const uint32_t v = i < ntiles ? tile_sums[i] : 0u;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan_1024(v, sh, &total);
        if (i < ntiles) tile_sums[i] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) *grand_total = carry;
}
__global__ void __launch_bounds__(1024) scan_tiles_kernel(uint32_t *__restrict__ counts, size_t nbins, const uint32_t *__restrict__ tile_off) {
    __shared__ uint32_t sh[17];
    const size_t base = (size_t)blockIdx.x * kScanTileBins + (size_t)threadIdx.x * kScanPer;
    uint32_t c[kScanPer], sum = 0;
#pragma unroll
    for (unsigned k = 0; k < kScanPer; k++) {
        c[k] = base + k < nbins ? counts[base + k] : 0u;
        sum += c[k];
    }
    uint32_t run = tile_off[blockIdx.x] + block_exclusive_scan_1024(sum, sh, nullptr);
#pragma unroll
    for (unsigned k = 0; k < kScanPer; k++) {
        if (base + k < nbins) counts[base + k] = run;
        run += c[k];
    }
}
int exclusive_scan_device(uint32_t *counts, size_t nbins) {
    const size_t ntiles = (nbins + kScanTileBins - 1) / kScanTileBins;
    DevBuf tiles;
    ZK_TRY(tiles.alloc(ntiles * 4));
    scan_tile_totals_kernel<<<(unsigned)ntiles, kScanThreads, 0, cur_stream()>>>(counts, nbins, (uint32_t *)tiles.p);
    scan_tile_offsets_kernel<<<1, kScanThreads, 0, cur_stream()>>>((uint32_t *)tiles.p, ntiles, counts + nbins);
    scan_tiles_kernel<<<(unsigned)ntiles, kScanThreads, 0, cur_stream()>>>(counts, nbins, (const uint32_t *)tiles.p);
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipStreamSynchronize(cur_stream()));           // `tiles` goes back to the pool
    return ZK_OK;
}
__global__ void key_scatter_kernel(const uint32_t *__restrict__ key, size_t n, const uint32_t *__restrict__ start, uint32_t *__restrict__ cursor,
                                   uint32_t *__restrict__ order) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t k = key[i];
    order[start[k] + atomicAdd(&cursor[k], 1u)] = (uint32_t)i;
}
inline unsigned blocks(size_t n);
__global__ void permute2_kernel(const uint32_t *__restrict__ order, size_t n, const uint32_t *__restrict__ a, const uint32_t *__restrict__ b,
                                uint32_t *__restrict__ oa, uint32_t *__restrict__ ob) {
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    uint32_t i = order[e];
    oa[e] = a[i]; ob[e] = b[i];
}
__global__ void permute1_kernel(const uint32_t *__restrict__ order, size_t n, const uint32_t *__restrict__ a, uint32_t *__restrict__ oa) {
    size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n) oa[e] = a[order[e]];
}
int group_by_device(const uint32_t *d_key, size_t n, size_t nbins, DevBuf &order, DevBuf &start) {
    DevBuf cursor;
    ZK_TRY(order.alloc((n ? n : 1) * 4));
    ZK_TRY(start.alloc((nbins + 1) * 4));
    ZK_TRY(cursor.alloc(nbins * 4));
    ZK_HIP(hipMemsetAsync(start.p, 0, (nbins + 1) * 4, cur_stream()));
    ZK_HIP(hipMemsetAsync(cursor.p, 0, nbins * 4, cur_stream()));
    if (n) key_hist_kernel<<<blocks(n), kBlock, 0, cur_stream()>>>(d_key, n, (uint32_t *)start.p);
    ZK_TRY(exclusive_scan_device((uint32_t *)start.p, nbins));
    if (n) key_scatter_kernel<<<blocks(n), kBlock, 0, cur_stream()>>>(d_key, n, (const uint32_t *)start.p, (uint32_t *)cursor.p, (uint32_t *)order.p);
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipStreamSynchronize(cur_stream()));           // `cursor` goes back to the pool
    return ZK_OK;
}
int upload_layer(const zk_gate *g, size_t n, uint32_t out_bits, uint32_t in_bits, LayerDev &L) {
    L.ngates = n; L.out_bits = out_bits; L.in_bits = in_bits;
    DevBuf raw, bad;
    ZK_TRY(raw.upload(g, (n ? n : 1) * sizeof(zk_gate)));
    ZK_TRY(bad.alloc(4));
    ZK_HIP(hipMemsetAsync(bad.p, 0, 4, cur_stream()));
    ZK_TRY(L.out.alloc((n ? n : 1) * 4));
    ZK_TRY(L.left.alloc((n ? n : 1) * 4));
    ZK_TRY(L.right.alloc((n ? n : 1) * 4));
    ZK_TRY(L.op.alloc((n ? n : 1) * 4));
    if (n) split_gates_kernel<<<blocks(n), kBlock, 0, cur_stream()>>>((const zk_gate *)raw.p, n, out_bits, in_bits, (uint32_t *)L.out.p, (uint32_t *)L.left.p,
                                                                      (uint32_t *)L.right.p, (uint32_t *)L.op.p, (int *)bad.p);
    ZK_HIP(hipGetLastError());
    int flag = 0;
    ZK_HIP(zk::memcpy_on_stream(&flag, bad.p, 4, hipMemcpyDeviceToHost));
    if (flag) return ZK_E_RANGE;                             // index panic in the reference
    ZK_TRY(group_by_device((const uint32_t *)L.left.p, n, (size_t)1 << in_bits, L.ord_left, L.st_left));
    ZK_TRY(group_by_device((const uint32_t *)L.right.p, n, (size_t)1 << in_bits, L.ord_right, L.st_right));
    ZK_TRY(group_by_device((const uint32_t *)L.out.p, n, (size_t)1 << out_bits, L.ord_out, L.st_out));
    for (DevBuf *d : {&L.l_right, &L.l_op, &L.r_left, &L.r_op, &L.l_out, &L.r_out}) ZK_TRY(d->alloc((n ? n : 1) * 4));
    if (n) {
        permute2_kernel<<<blocks(n), kBlock, 0, cur_stream()>>>((const uint32_t *)L.ord_left.p, n, (const uint32_t *)L.right.p, (const uint32_t *)L.op.p,
                                                               (uint32_t *)L.l_right.p, (uint32_t *)L.l_op.p);
        permute2_kernel<<<blocks(n), kBlock, 0, cur_stream()>>>((const uint32_t *)L.ord_right.p, n, (const uint32_t *)L.left.p, (const uint32_t *)L.op.p,
                                                               (uint32_t *)L.r_left.p, (uint32_t *)L.r_op.p);
        permute1_kernel<<<blocks(n), kBlock, 0, cur_stream()>>>((const uint32_t *)L.ord_left.p, n, (const uint32_t *)L.out.p, (uint32_t *)L.l_out.p);
        permute1_kernel<<<blocks(n), kBlock, 0, cur_stream()>>>((const uint32_t *)L.ord_right.p, n, (const uint32_t *)L.out.p, (uint32_t *)L.r_out.p);
        ZK_HIP(hipGetLastError());
        ZK_HIP(hipStreamSynchronize(cur_stream()));
    }
    return ZK_OK;
}

inline unsigned blocks(size_t n) { return (unsigned)((n + kBlock - 1) / kBlock); }

// eq(point, .) over 2^nbits entries: outer products of the half tables (eq_table.cuh), a handful of launches
// the same from challenges that live in the proof slots on the device (element i at dev + i * stride elements)
template <class F> int eq_table_dev(const void *dev, size_t stride, uint32_t nbits, TablePtr &out, const void *scale_dev = nullptr) {
    ZK_TRY(alloc_table(F::ID, (size_t)1 << nbits, out));
    EqBuilder<F> eb;
    return eb.build_dev(dev, stride, nbits, out->dptr, scale_dev);
}
template <class F> int eq_table(const uint64_t *point, uint32_t nbits, TablePtr &out, const Fe<F> *scale = nullptr) {
    ZK_TRY(alloc_table(F::ID, (size_t)1 << nbits, out));
    EqBuilder<F> eb;
    return eb.build(point, nbits, out->dptr, scale);
}

template <class F> int evaluate_layers(std::vector<LayerDev> &layers, const uint64_t *inputs, size_t ninputs, std::vector<TablePtr> &W) {
    size_t nl = layers.size();
    W.resize(nl + 1);
    zk_table *t = nullptr;
    ZK_TRY(zk_table_upload(F::ID, inputs, ninputs, &t));
    W[nl].reset(t);
    for (size_t l = nl; l-- > 0;) {
        size_t nout = (size_t)1 << layers[l].out_bits;
        ZK_TRY(alloc_table(F::ID, nout, W[l]));
        circuit_layer_kernel<F><<<blocks(nout), kBlock, 0, cur_stream()>>>(layers[l].arrays(), (const uint32_t *)layers[l].ord_out.p, (const uint32_t *)layers[l].st_out.p,
                                                           nout, W[l + 1]->dptr, W[l]->dptr);
        ZK_HIP(hipGetLastError());
    }
    ZK_HIP(hipStreamSynchronize(cur_stream()));
    return ZK_OK;
}

int build_layers(const zk_gate *gates, const size_t *gate_counts, size_t nlayers, const uint32_t *out_bits, size_t ninputs,
                 std::vector<LayerDev> &layers) {
    if (!is_pow2(ninputs)) return ZK_E_NOT_POW2;
    layers.resize(nlayers);
    size_t off = 0;
    for (size_t l = 0; l < nlayers; l++) {
        uint32_t in_bits = l + 1 < nlayers ? out_bits[l + 1] : ilog2(ninputs);
        if (out_bits[l] > 30 || in_bits > 30 || in_bits == 0) return ZK_E_ARG;
        ZK_TRY(upload_layer(gates + off, gate_counts[l], out_bits[l], in_bits, layers[l]));
        off += gate_counts[l];
    }
    return ZK_OK;
}

template <class F> int sparse_prove(std::vector<LayerDev> &layers, const uint64_t *inputs, size_t ninputs, uint64_t *circuit_output,
                                    uint64_t *claimed_sum, uint64_t *layer_claims, uint64_t *coeffs, uint64_t *challenges,
                                    uint64_t *wb_evals, uint64_t *wc_evals, uint64_t *output_challenges, float *ms_layers) {
    const size_t L64 = F::N / 2, esz = 4 * F::N;
    const size_t nlayers = layers.size();
    std::vector<uint32_t> out_bits_v(nlayers);
    for (size_t l = 0; l < nlayers; l++) out_bits_v[l] = layers[l].out_bits;
    const uint32_t *out_bits = out_bits_v.data();
    if (ninputs != ((size_t)1 << layers[nlayers - 1].in_bits)) return ZK_E_LEN_MISMATCH;
    std::vector<TablePtr> W;
    ZK_TRY((evaluate_layers<F>(layers, inputs, ninputs, W)));                        // Circuit::evaluate (arithmetic_circuit.rs:65-109)
    ZK_TRY(zk_table_download(W[0].get(), circuit_output));
    zk_transcript tr;
    // transcript.append(w0 bytes) gkr_protocol.rs:49 (an output layer of one wire is one gate plus a zero pad, :43-47)
    ZK_TRY(transcript_absorb_table(tr.t, W[0].get()));
    std::vector<uint64_t> ra((size_t)out_bits[0] * L64);
    for (uint32_t i = 0; i < out_bits[0]; i++) store_el<F>(ra.data() + i * L64, tr.t.random_challenge_as_field_element<F>());   // :50
    memcpy(output_challenges, ra.data(), ra.size() * 8);
    uint64_t claim0[6];
    ZK_TRY(zk_mle_evaluate(W[0].get(), ra.data(), out_bits[0], claim0));             // :51
    tr.t.append_be<F>(load_el<F>(claim0));                                           // layer 0's sumcheck claim (sumcheck_gkr_protocol.rs:35)
    // From here to the end the proof is ONE stream of kernels: the sponge, every round's coefficients and challenge, the final values of
    // each phase (u = W(rb*), W(rc*)) and the alpha / beta / claim links between layers live in device proof slots; the table-building
    // kernels read challenges and u from there.  One download at the end (round 1 synchronised 4 times per layer).
    // Slots of layer l (k = in_bits): phase 1 at base: 4 k round slots + 4 final values; phase 2 the same; then wb, wc, alpha, beta, claim.
    const size_t per = 4;
    std::vector<size_t> base(nlayers + 1, 0);
    for (size_t l = 0; l < nlayers; l++) base[l + 1] = base[l] + 2 * (per * layers[l].in_bits + 4) + 5;
    ProofSlotsBase *psp = nullptr;
    ZK_TRY(proof_slots_new(F::ID, tr.t, 3, base[nlayers], &psp));
    std::unique_ptr<ProofSlotsBase> ps(psp);
    ps->set_claim(claim0);                                                           // layer 0's claimed sum, computed above
    std::vector<hipEvent_t> ev(ms_layers ? 2 * nlayers : 0);
    for (hipEvent_t &e : ev) ZK_HIP(hipEventCreate(&e));
    struct EvGuard { std::vector<hipEvent_t> &v; ~EvGuard() { for (hipEvent_t e : v) (void)hipEventDestroy(e); } } ev_guard{ev};
    uint64_t one[2 * 6];
    store_el<F>(one + L64, fe_one<F>());
    for (size_t l = 0; l < nlayers; l++) {                                           // :57
        if (ms_layers) ZK_HIP(hipEventRecord(ev[2 * l], cur_stream()));
        LayerDev &Ly = layers[l];
        const uint32_t k = Ly.in_bits;
        const size_t nk = (size_t)1 << k, ng = Ly.ngates;
        const zk_table *Wn = W[l + 1].get();
        const size_t s1 = base[l], s2 = s1 + per * k + 4, lk = s2 + per * k + 4;      // phase 1, phase 2, link slots
        // gate weights w_g = eq(ra, out_g) (layer 0) or alpha eq(rb, out_g) + beta eq(rc, out_g): the previous layer's challenges and
        // alpha / beta are read from its slots, the constants folded into the eq half tables (eq_table.cuh)
        TablePtr eqA, eqB;
        DevBuf w, uah, ual, ubh, ubl, uhi, ulo;
        EqBuilder<F> ebA, ebB;                                 // the half tables live until the layer's kernels are enqueued (stream-ordered pool)
        GateWeights gw{};
        static const bool want_table = [] { const char *e = getenv("ZK_GKR_WEIGHT_TABLE"); return e && e[0] == '1'; }();   // measurements / tests
        const bool halves = Ly.out_bits > (uint32_t)kEqSmallBits && ng > 0 && !want_table;
        if (halves) {
            if (l == 0) {
                ZK_TRY(ebA.halves(ra.data(), Ly.out_bits, &gw.ah, &gw.al, &gw.lbits));
            } else {
                const size_t p1 = base[l - 1], p2 = p1 + per * layers[l - 1].in_bits + 4, plk = p2 + per * layers[l - 1].in_bits + 4;
                unsigned lb2 = 0;
                ZK_TRY(ebA.halves_dev(ps->slot_ptr(p1 + 3), per, Ly.out_bits, &gw.ah, &gw.al, &gw.lbits, ps->slot_ptr(plk + 2)));   // alpha eq(rb, .)
                ZK_TRY(ebB.halves_dev(ps->slot_ptr(p2 + 3), per, Ly.out_bits, &gw.bh, &gw.bl, &lb2, ps->slot_ptr(plk + 3)));        // beta eq(rc, .)
            }
            // the kernels read them in the products' internal form (UHalves above): high halves as they are stored, low halves times U / R
            ZK_TRY((halves_to_u<F>(gw.ah, 0, gw.al, Ly.out_bits, gw.lbits, uah, ual)));
            gw.uah = uah.p; gw.ual = ual.p;
            if (gw.bh) {
                ZK_TRY((halves_to_u<F>(gw.bh, 0, gw.bl, Ly.out_bits, gw.lbits, ubh, ubl)));
                gw.ubh = ubh.p; gw.ubl = ubl.p;
            }
        } else {
            ZK_TRY(w.alloc((ng ? ng : 1) * esz));
            if (l == 0) {
                ZK_TRY((eq_table<F>(ra.data(), Ly.out_bits, eqA)));
                if (ng) gate_weights_kernel<F><<<blocks(ng), kBlock, 0, cur_stream()>>>((const uint32_t *)Ly.out.p, ng, eqA->dptr, nullptr, w.p);
            } else {
                const size_t p1 = base[l - 1], p2 = p1 + per * layers[l - 1].in_bits + 4, plk = p2 + per * layers[l - 1].in_bits + 4;
                ZK_TRY((eq_table_dev<F>(ps->slot_ptr(p1 + 3), per, Ly.out_bits, eqA, ps->slot_ptr(plk + 2))));     // alpha eq(rb, .)
                ZK_TRY((eq_table_dev<F>(ps->slot_ptr(p2 + 3), per, Ly.out_bits, eqB, ps->slot_ptr(plk + 3))));     // beta eq(rc, .)
                if (ng) gate_weights_kernel<F><<<blocks(ng), kBlock, 0, cur_stream()>>>((const uint32_t *)Ly.out.p, ng, eqA->dptr, eqB->dptr, w.p);
            }
            gw.w = w.p;
        }
        ZK_HIP(hipGetLastError());
        // phase 1 (rounds over b): f = W(b) H1(b) + H0(b) * 1 -- the second product's factor is the constant one, never a table
        TablePtr H1, H0;
        ZK_TRY(alloc_table(F::ID, nk, H1));
        ZK_TRY(alloc_table(F::ID, nk, H0));
        {
            const unsigned grid = (unsigned)((nk + kPhaseGroups - 1) / kPhaseGroups);
            const uint32_t *st = (const uint32_t *)Ly.st_left.p, *wi = (const uint32_t *)(halves ? Ly.l_out.p : Ly.ord_left.p);
            const uint32_t *oth = (const uint32_t *)Ly.l_right.p, *ops = (const uint32_t *)Ly.l_op.p;
            if (gw.uah) phase1_tables_kernel<F, true><<<grid, kBlock, 0, cur_stream()>>>(st, nk, gw, wi, oth, ops, Wn->dptr, H1->dptr, H0->dptr);
            else phase1_tables_kernel<F, false><<<grid, kBlock, 0, cur_stream()>>>(st, nk, gw, wi, oth, ops, Wn->dptr, H1->dptr, H0->dptr);
        }
        ZK_HIP(hipGetLastError());
        const zk_table *t1[4] = {Wn, H1.get(), H0.get(), nullptr};
        // the layer's claim (alpha wb + beta wc of the previous link, already in its slot) is absorbed in front of round 0
        ZK_TRY(ps->rounds(s1, t1, 2, 2, one, nullptr, l > 0 ? 1 : 0, l > 0 ? base[l - 1] + 2 * (per * layers[l - 1].in_bits + 4) + 4 : 0));
        // phase 2 (rounds over c, b fixed to rb*, u = W(rb*) = phase 1's first final value):
        // A(c) (u + W(c)) + M(c) u W(c) = C(c) W(c) + A(c) * u with C = A + u M
        const void *u_dev = ps->slot_ptr(s1 + per * k);
        TablePtr eqL, Cc, A;
        EqBuilder<F> ebL;
        OtherTable tl{};
        if (k > (uint32_t)kEqSmallBits && !want_table) {           // eq(rb*, .) is only gathered once per gate: half tables do (as for the weights)
            ZK_TRY(ebL.halves_dev(ps->slot_ptr(s1 + 3), per, k, &tl.hi, &tl.lo, &tl.lbits));
            if (gw.uah) {                                           // both halves times U / R: their product is eq(rb*, .) U (UHalves above)
                ZK_TRY((halves_to_u<F>(tl.hi, 1, tl.lo, k, tl.lbits, uhi, ulo)));
                tl.uhi = uhi.p; tl.ulo = ulo.p;
            }
        } else {
            ZK_TRY((eq_table_dev<F>(ps->slot_ptr(s1 + 3), per, k, eqL)));
            tl.tab = eqL->dptr;
        }
        ZK_TRY(alloc_table(F::ID, nk, Cc));
        ZK_TRY(alloc_table(F::ID, nk, A));
        {
            const unsigned grid = (unsigned)((nk + kPhaseGroups - 1) / kPhaseGroups);
            const uint32_t *st = (const uint32_t *)Ly.st_right.p, *wi = (const uint32_t *)(halves ? Ly.r_out.p : Ly.ord_right.p);
            const uint32_t *oth = (const uint32_t *)Ly.r_left.p, *ops = (const uint32_t *)Ly.r_op.p;
            if (gw.uah) phase2_tables_kernel<F, true><<<grid, kBlock, 0, cur_stream()>>>(st, nk, gw, wi, oth, ops, tl, u_dev, Cc->dptr, A->dptr);
            else phase2_tables_kernel<F, false><<<grid, kBlock, 0, cur_stream()>>>(st, nk, gw, wi, oth, ops, tl, u_dev, Cc->dptr, A->dptr);
        }
        ZK_HIP(hipGetLastError());
        const zk_table *t2[4] = {Cc.get(), Wn, A.get(), nullptr};
        const void *cdev[2] = {nullptr, u_dev};
        ZK_TRY(ps->rounds(s2, t2, 2, 2, nullptr, cdev, 0, 0));
        if (l + 1 < nlayers)                                                         // gkr_protocol.rs:109-133, on the device
            ZK_TRY(ps->link(s1 + per * k, s2 + per * k + 1, lk, lk + 1, lk + 2, lk + 3, lk + 4));
        if (ms_layers) ZK_HIP(hipEventRecord(ev[2 * l + 1], cur_stream()));
    }
    std::vector<uint64_t> hs(base[nlayers] * L64);
    ZK_TRY(ps->collect(tr.t, hs.data()));                                            // the proof's only download
    // (a clean proof's collect does not wait for the stream to drain -- the host already holds every byte: the layer timings do)
    if (ms_layers && nlayers) ZK_HIP(hipEventSynchronize(ev[2 * nlayers - 1]));
    auto slot = [&](size_t s) { return hs.data() + s * L64; };
    size_t coff = 0, choff = 0;
    for (size_t l = 0; l < nlayers; l++) {
        const uint32_t k = layers[l].in_bits;
        const size_t s1 = base[l], s2 = s1 + per * k + 4, lk = s2 + per * k + 4;
        memcpy(layer_claims + l * L64, l == 0 ? claim0 : slot(base[l - 1] + 2 * (per * layers[l - 1].in_bits + 4) + 4), L64 * 8);
        for (int ph = 0; ph < 2; ph++)
            for (uint32_t j = 0; j < k; j++) {
                const size_t s = (ph ? s2 : s1) + per * j;
                memcpy(coeffs + (coff + ((size_t)ph * k + j) * 3) * L64, slot(s), 3 * L64 * 8);
                memcpy(challenges + (choff + (size_t)ph * k + j) * L64, slot(s + 3), L64 * 8);
            }
        if (l + 1 < nlayers) {
            memcpy(wb_evals + l * L64, slot(lk), L64 * 8);
            memcpy(wc_evals + l * L64, slot(lk + 1), L64 * 8);
        }
        coff += (size_t)2 * k * 3;
        choff += (size_t)2 * k;
        if (ms_layers) (void)hipEventElapsedTime(&ms_layers[l], ev[2 * l], ev[2 * l + 1]);
    }
    memcpy(claimed_sum, layer_claims + (nlayers - 1) * L64, L64 * 8);
    return ZK_OK;
}

template <class F> int wiring_eval(const zk_gate *g, size_t ngates, uint32_t out_bits, uint32_t in_bits, const uint64_t *alpha, const uint64_t *pa,
                                   const uint64_t *beta, const uint64_t *pb, const uint64_t *rb, const uint64_t *rc, uint64_t *add_r, uint64_t *mul_r) {
    const size_t esz = 4 * F::N;
    LayerDev L;
    ZK_TRY(upload_layer(g, ngates, out_bits, in_bits, L));
    TablePtr eqA, eqB, eqL, eqR;
    const Fe<F> al = pb ? load_el<F>(alpha) : fe_zero<F>(), be = pb ? load_el<F>(beta) : fe_zero<F>();
    ZK_TRY((eq_table<F>(pa, out_bits, eqA, pb ? &al : nullptr)));
    if (pb) ZK_TRY((eq_table<F>(pb, out_bits, eqB, &be)));
    ZK_TRY((eq_table<F>(rb, in_bits, eqL)));
    ZK_TRY((eq_table<F>(rc, in_bits, eqR)));
    DevBuf w;
    ZK_TRY(w.alloc(ngates * esz));
    gate_weights_kernel<F><<<blocks(ngates), kBlock, 0, cur_stream()>>>((const uint32_t *)L.out.p, ngates, eqA->dptr, pb ? eqB->dptr : nullptr, w.p);
    int grid = reduce_grid_for(ngates);
    void *part;
    ZK_TRY(scratch(esz * ((size_t)grid * 2 + 2), &part));
    void *res = (char *)part + esz * (size_t)grid * 2;
    wiring_eval_kernel<F><<<grid, kBlock, 0, cur_stream()>>>(L.arrays(), ngates, w.p, eqL->dptr, eqR->dptr, part);
    finish_sums_kernel<F><<<1, kBlock, 0, cur_stream()>>>(part, (size_t)grid, 2, res);
    ZK_HIP(hipGetLastError());
    uint64_t both[12];
    ZK_HIP(zk::memcpy_on_stream(both, res, esz * 2, hipMemcpyDeviceToHost));
    memcpy(add_r, both, esz);
    memcpy(mul_r, both + F::N / 2, esz);
    return ZK_OK;
}

}  // namespace

struct zk_sparse_circuit {
    std::vector<LayerDev> layers;
};

namespace zk {
// Circuit::evaluate (arithmetic_circuit.rs:65-109) on the GPU for arbitrary layer widths: widths[l] outputs for
// layer l (max output index + 1, :73-80), widths[nlayers] = ninputs.  evals = layer 0 .. inputs concatenated.
int circuit_evaluate_device(int field, const zk_gate *gates, const size_t *gate_counts, size_t nlayers, const size_t *widths,
                            const uint64_t *inputs, uint64_t *evals) {
    ZK_DISPATCH_FIELD(field, {
        const size_t L64 = F::N / 2, esz = 4 * F::N;
        std::vector<size_t> eoff(nlayers + 2, 0), goff(nlayers + 1, 0);
        for (size_t l = 0; l <= nlayers; l++) eoff[l + 1] = eoff[l] + widths[l];
        for (size_t l = 0; l < nlayers; l++) goff[l + 1] = goff[l] + gate_counts[l];
        DevBuf dev;
        ZK_TRY(dev.alloc(eoff[nlayers + 1] * esz));
        ZK_HIP(zk::memcpy_on_stream((char *)dev.p + eoff[nlayers] * esz, inputs, widths[nlayers] * esz, hipMemcpyHostToDevice));
        for (size_t l = nlayers; l-- > 0;) {
            size_t n = gate_counts[l], nout = widths[l], nin = widths[l + 1];
            std::vector<uint32_t> out(n), left(n), right(n), op(n), order, start;
            for (size_t i = 0; i < n; i++) {
                const zk_gate &g = gates[goff[l] + i];
                if (g.left >= nin || g.right >= nin || g.out >= nout) return ZK_E_RANGE;     // index panic in the reference
                out[i] = (uint32_t)g.out; left[i] = (uint32_t)g.left; right[i] = (uint32_t)g.right; op[i] = g.op ? 1u : 0u;
            }
            group_by(out, nout, order, start);
            DevBuf d_out, d_left, d_right, d_op, d_ord, d_st;
            ZK_TRY(d_out.upload(out.data(), n * 4));
            ZK_TRY(d_left.upload(left.data(), n * 4));
            ZK_TRY(d_right.upload(right.data(), n * 4));
            ZK_TRY(d_op.upload(op.data(), n * 4));
            ZK_TRY(d_ord.upload(order.data(), n * 4));
            ZK_TRY(d_st.upload(start.data(), start.size() * 4));
            GateArrays ga{(const uint32_t *)d_out.p, (const uint32_t *)d_left.p, (const uint32_t *)d_right.p, (const uint32_t *)d_op.p};
            circuit_layer_kernel<F><<<blocks(nout), kBlock, 0, cur_stream()>>>(ga, (const uint32_t *)d_ord.p, (const uint32_t *)d_st.p, nout,
                                                               (const char *)dev.p + eoff[l + 1] * esz, (char *)dev.p + eoff[l] * esz);
            ZK_HIP(hipGetLastError());
            ZK_HIP(hipStreamSynchronize(cur_stream()));
        }
        ZK_HIP(zk::memcpy_on_stream(evals, dev.p, eoff[nlayers + 1] * esz, hipMemcpyDeviceToHost));
        (void)L64;
    });
    return ZK_OK;
}
}  // namespace zk

extern "C" {

int zk_gkr_sparse_prove(int field, const zk_gate *gates, const size_t *gate_counts, size_t nlayers, const uint32_t *out_bits,
                        const uint64_t *inputs, size_t ninputs, uint64_t *circuit_output, uint64_t *claimed_sum, uint64_t *layer_claims,
                        uint64_t *coeffs, uint64_t *challenges, uint64_t *wb_evals, uint64_t *wc_evals, uint64_t *output_challenges,
                        float *ms_layers) {
    if (!gates || !gate_counts || !out_bits || !inputs || !circuit_output || !claimed_sum || !layer_claims || !coeffs || !challenges ||
        !output_challenges || nlayers == 0)
        return ZK_E_ARG;
    if (nlayers > 1 && (!wb_evals || !wc_evals)) return ZK_E_ARG;
    zk_sparse_circuit *c = nullptr;
    ZK_TRY(zk_sparse_circuit_new(gates, gate_counts, nlayers, out_bits, ninputs, &c));
    int rc = zk_gkr_sparse_prove_compiled(field, c, inputs, ninputs, circuit_output, claimed_sum, layer_claims, coeffs, challenges, wb_evals,
                                          wc_evals, output_challenges, ms_layers);
    zk_sparse_circuit_free(c);
    return rc;
}
int zk_sparse_circuit_new(const zk_gate *gates, const size_t *gate_counts, size_t nlayers, const uint32_t *out_bits, size_t ninputs,
                          zk_sparse_circuit **out) {
    if (!gates || !gate_counts || !out_bits || !out || nlayers == 0) return ZK_E_ARG;
    if (out_bits[0] == 0) return ZK_E_ARG;      // a single output wire is padded to two by the caller (gkr_protocol.rs:43-47)
    ZK_TRY(require_device());
    std::unique_ptr<zk_sparse_circuit> c(new zk_sparse_circuit());
    ZK_TRY(build_layers(gates, gate_counts, nlayers, out_bits, ninputs, c->layers));
    *out = c.release();
    return ZK_OK;
}
int zk_sparse_circuit_free(zk_sparse_circuit *c) {
    delete c;
    return ZK_OK;
}
int zk_gkr_sparse_prove_compiled(int field, const zk_sparse_circuit *c, const uint64_t *inputs, size_t ninputs, uint64_t *circuit_output,
                                 uint64_t *claimed_sum, uint64_t *layer_claims, uint64_t *coeffs, uint64_t *challenges, uint64_t *wb_evals,
                                 uint64_t *wc_evals, uint64_t *output_challenges, float *ms_layers) {
    if (!c || !inputs || !circuit_output || !claimed_sum || !layer_claims || !coeffs || !challenges || !output_challenges) return ZK_E_ARG;
    if (c->layers.size() > 1 && (!wb_evals || !wc_evals)) return ZK_E_ARG;
    ZK_TRY(require_device());
    std::vector<LayerDev> &layers = const_cast<zk_sparse_circuit *>(c)->layers;   // read-only use of device buffers
    ZK_DISPATCH_FIELD(field, return sparse_prove<F>(layers, inputs, ninputs, circuit_output, claimed_sum, layer_claims, coeffs, challenges,
                                                    wb_evals, wc_evals, output_challenges, ms_layers));
    return ZK_OK;
}
int zk_gkr_sparse_wiring_eval(int field, const zk_gate *layer_gates, size_t ngates, uint32_t out_bits, uint32_t in_bits, const uint64_t *alpha,
                              const uint64_t *pa, const uint64_t *beta, const uint64_t *pb, const uint64_t *rb, const uint64_t *rc,
                              uint64_t *add_r, uint64_t *mul_r) {
    if (!layer_gates || !pa || !rb || !rc || !add_r || !mul_r || (pb && (!alpha || !beta))) return ZK_E_ARG;
    ZK_TRY(require_device());
    ZK_DISPATCH_FIELD(field, return wiring_eval<F>(layer_gates, ngates, out_bits, in_bits, alpha, pa, beta, pb, rb, rc, add_r, mul_r));
    return ZK_OK;
}
int zk_sparse_circuit_evaluate(int field, const zk_gate *gates, const size_t *gate_counts, size_t nlayers, const uint32_t *out_bits,
                               const uint64_t *inputs, size_t ninputs, uint64_t *evals) {
    if (!gates || !gate_counts || !out_bits || !inputs || !evals) return ZK_E_ARG;
    ZK_TRY(require_device());
    std::vector<LayerDev> layers;
    ZK_TRY(build_layers(gates, gate_counts, nlayers, out_bits, ninputs, layers));
    ZK_DISPATCH_FIELD(field, {
        std::vector<TablePtr> W;
        ZK_TRY((evaluate_layers<F>(layers, inputs, ninputs, W)));
        size_t off = 0;
        for (size_t l = 0; l <= nlayers; l++) {
            ZK_TRY(zk_table_download(W[l].get(), evals + off * (F::N / 2)));
            off += W[l]->len;
        }
    });
    return ZK_OK;
}

}  // extern "C"
