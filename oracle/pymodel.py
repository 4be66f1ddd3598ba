"""Second, independent restatement of the hot path with Python big integers.

TEST INFRASTRUCTURE ONLY (same rule as zkoracle.h).  Written separately from the C oracle --
plain canonical integers mod p, no Montgomery form, no limbs -- so that the two restatements
check each other on random inputs (tests/test_oracle_crosscheck.py).  Small inputs only.
Citations are to files under the reference repository.
"""

P = {
    "bls12_381_fr": 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001,
    "bls12_381_fq": 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab,
    "bn254_fq": 0x30644e72e131a029b85045b68181585d97816a916871ca8d3c208c16d87cfd47,
    "bn254_fr": 0x30644e72e131a029b85045b68181585d2833e84879b9709143e1f593f0000001,
}

# ---- Keccak-256 (FIPS-202 permutation, original 0x01 padding) ------------------------------
_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B,
       0x0000000080000001, 0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088,
       0x0000000080008009, 0x000000008000000A, 0x000000008000808B, 0x800000000000008B, 0x8000000000008089,
       0x8000000000008003, 0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
       0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_M = (1 << 64) - 1


def _rol(x, n):
    n %= 64
    return ((x << n) | (x >> (64 - n))) & _M if n else x


def _f1600(a):
    # a[x][y] lanes; textbook theta/rho/pi/chi/iota
    for rnd in range(24):
        c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
        d = [c[(x - 1) % 5] ^ _rol(c[(x + 1) % 5], 1) for x in range(5)]
        a = [[a[x][y] ^ d[x] for y in range(5)] for x in range(5)]
        b = [[0] * 5 for _ in range(5)]
        x, y = 1, 0
        b[0][0] = a[0][0]
        for t in range(24):
            b[y][(2 * x + 3 * y) % 5] = _rol(a[x][y], (t + 1) * (t + 2) // 2)
            x, y = y, (2 * x + 3 * y) % 5
        a = [[b[x][y] ^ ((~b[(x + 1) % 5][y]) & b[(x + 2) % 5][y] & _M) for y in range(5)] for x in range(5)]
        a[0][0] ^= _RC[rnd]
    return a


def keccak256(data):
    rate = 136
    msg = bytearray(data)
    msg.append(0x01)
    while len(msg) % rate:
        msg.append(0)
    msg[-1] |= 0x80
    a = [[0] * 5 for _ in range(5)]
    for off in range(0, len(msg), rate):
        for i in range(rate // 8):
            a[i % 5][i // 5] ^= int.from_bytes(msg[off + 8 * i: off + 8 * i + 8], "little")
        a = _f1600(a)
    return b"".join(a[i % 5][i // 5].to_bytes(8, "little") for i in range(4))


class Transcript:
    """fiat_shamir_transcript.rs:5-43 modelled as `everything absorbed so far`."""

    def __init__(self):
        self.buf = bytearray()

    def append(self, data):                               # :22
        self.buf += bytes(data)

    def sample(self):                                     # :29-36 finalize a clone, absorb the digest
        d = keccak256(self.buf)
        self.buf += d
        return d

    def challenge(self, p):                               # :38-43 from_le_bytes_mod_order
        return int.from_bytes(self.sample(), "little") % p


def be32(v, nbytes=32):
    return int(v).to_bytes(nbytes, "big")


def le32(v, nbytes=32):
    return int(v).to_bytes(nbytes, "little")


# ---- MLE (evaluation_form.rs) ----------------------------------------------------------------
def partial_evaluate(poly, var, r, p):
    n = len(poly).bit_length() - 1
    power = n - 1 - var
    out = []
    for j in range(len(poly)):
        if (j >> power) & 1 == 0:                         # y1 indices, in increasing order (:98-102)
            y1, y2 = poly[j], poly[j | (1 << power)]
            out.append((y1 + r * (y2 - y1)) % p)          # :88
    return out


def evaluate(poly, values, p):
    cur = list(poly)
    for v in values:                                      # :27-30
        if len(cur) < 2:
            raise AssertionError("Evaluated values must be a power of 2")
        cur = partial_evaluate(cur, 0, v, p)
    return cur[0]


def tensor(wb, wc, op, p):
    return [op(b, c) % p for b in wb for c in wc]         # :116-120 b-major


# ---- univariate (dense_univariate.rs) ----------------------------------------------------------
def lagrange_interpolate(xs, ys, p):
    n = len(xs)
    out = [0] * n
    for i in range(n):
        num = [1]
        for k in range(n):
            if xs[k] != xs[i]:
                nxt = [0] * (len(num) + 1)
                for d, c in enumerate(num):
                    nxt[d] = (nxt[d] - c * xs[k]) % p
                    nxt[d + 1] = (nxt[d + 1] + c) % p
                num = nxt
        den = sum(c * pow(xs[i], d, p) for d, c in enumerate(num)) % p
        s = ys[i] * pow(den, -1, p) % p
        for d, c in enumerate(num):
            out[d] = (out[d] + s * c) % p
    return out


def uni_eval(coeffs, x, p):
    return sum(c * pow(x, d, p) for d, c in enumerate(coeffs)) % p


# ---- basic sumcheck (prover.rs / verifier.rs) ---------------------------------------------------
def sumcheck_basic_prove(table, p):
    t = Transcript()
    claimed = sum(table) % p                              # prover.rs:28
    t.append(b"".join(be32(v) for v in table))            # :38-39
    t.append(be32(claimed))                               # :40-41
    cur, rounds, chal = list(table), [], []
    while len(cur) > 1:                                   # :46
        h = len(cur) // 2
        uni = [sum(cur[:h]) % p, sum(cur[h:]) % p]        # :74-89
        rounds.append(uni)
        t.append(be32(uni[0]) + be32(uni[1]))             # :52-55
        r = t.challenge(p)                                # :58
        chal.append(r)
        cur = partial_evaluate(cur, 0, r, p)              # :61
    return claimed, rounds, chal


def sumcheck_basic_verify(table, claimed, rounds, p):
    n = len(table).bit_length() - 1
    if len(rounds) != n:
        return False
    t = Transcript()
    t.append(b"".join(be32(v) for v in table))
    t.append(be32(claimed))
    cur, chal = claimed, []
    for uni in rounds:
        if (uni[0] + uni[1]) % p != cur:                  # verifier.rs:51-56
            return False
        t.append(be32(uni[0]) + be32(uni[1]))
        r = t.challenge(p)
        chal.append(r)
        cur = (uni[0] + r * (uni[1] - uni[0])) % p        # :64
    return evaluate(table, chal, p) == cur                # :67-70


# ---- GKR sumcheck (sumcheck_gkr_protocol.rs) -----------------------------------------------------
def round_evals(tables, p):
    """tables[prod][fac] -> evaluations at t = 0..deg of sum_i sum_prod prod_fac X_t[i]   (:113-143)"""
    deg = len(tables[0])
    h = len(tables[0][0]) // 2
    out = []
    for tpt in range(deg + 1):
        acc = 0
        for i in range(h):
            for prod in tables:
                term = 1
                for x in prod:
                    term = term * (x[i] + tpt * (x[i + h] - x[i])) % p
                acc += term
        out.append(acc % p)
    return out


def sumcheck_gkr_prove(tables, claimed, t, p):
    deg = len(tables[0])
    t.append(be32(claimed))                               # :35
    cur = [[list(x) for x in prod] for prod in tables]
    polys, chal = [], []
    while len(cur[0][0]) > 1:
        ev = round_evals(cur, p)
        co = lagrange_interpolate(list(range(deg + 1)), ev, p)     # :46-50
        t.append(b"".join(le32(c) for c in co))           # :52 little-endian coefficients
        polys.append(co)
        r = t.challenge(p)                                # :55
        cur = [[partial_evaluate(x, 0, r, p) for x in prod] for prod in cur]   # :57
        chal.append(r)
    return polys, chal


def sumcheck_gkr_verify(claimed, polys, t, p):
    t.append(be32(claimed))
    cur, chal = claimed, []
    for co in polys:
        if (uni_eval(co, 0, p) + uni_eval(co, 1, p)) % p != cur:
            return False, [], cur
        t.append(b"".join(le32(c) for c in co))
        r = t.challenge(p)
        cur = uni_eval(co, r, p)
        chal.append(r)
    return True, chal, cur


# ---- circuit + GKR (arithmetic_circuit.rs, gkr_protocol.rs, utils.rs) ----------------------------
ADD, MUL = 0, 1


def circuit_evaluate(layers, inputs, p):
    cur, evs = list(inputs), [list(inputs)]
    for layer in reversed(layers):                        # :72
        res = [0] * (max([g[2] for g in layer] + [0]) + 1)
        for (l, r, o, op) in layer:
            res[o] = (res[o] + (cur[l] + cur[r] if op == ADD else cur[l] * cur[r])) % p   # :90-96
        cur = res
        evs.append(list(cur))
    evs.reverse()
    return evs


def num_vars(layer_index):
    return 3 if layer_index == 0 else 3 * layer_index + 2


def wiring_index(layer_index, a, b, c):
    s = format(a, "0>%db" % layer_index) + format(b, "0>%db" % (layer_index + 1)) + format(c, "0>%db" % (layer_index + 1))
    return int(s, 2)                                      # :180-196, via the same digit strings


def add_mul_mle(layer, layer_index):
    n = 1 << num_vars(layer_index)
    add, mul = [0] * n, [0] * n
    for (l, r, o, op) in layer:
        (add if op == ADD else mul)[wiring_index(layer_index, o, l, r)] = 1
    return add, mul


def _fold_all(tab, vals, p):
    for v in vals:
        tab = partial_evaluate(tab, 0, v, p)
    return tab


def gkr_prove(layers, inputs, p):
    evs = circuit_evaluate(layers, inputs, p)
    t = Transcript()
    w0 = list(evs[0])
    if len(w0) == 1:
        w0.append(0)                                      # gkr_protocol.rs:43-47
    t.append(b"".join(be32(v) for v in w0))
    ra = t.challenge(p)
    claim = evaluate(w0, [ra], p)
    alpha = beta = 0
    rb, rc = [], []
    proofs, wbs, wcs = [], [], []
    for L, layer in enumerate(layers):
        add, mul = add_mul_mle(layer, L)
        if L == 0:
            add_bc, mul_bc = partial_evaluate(add, 0, ra, p), partial_evaluate(mul, 0, ra, p)
        else:                                             # utils.rs:23-68
            add_bc = [(alpha * x + beta * y) % p for x, y in zip(_fold_all(add, rb, p), _fold_all(add, rc, p))]
            mul_bc = [(alpha * x + beta * y) % p for x, y in zip(_fold_all(mul, rb, p), _fold_all(mul, rc, p))]
        w = evs[L + 1]
        fbc = [[add_bc, tensor(w, w, lambda a, b: a + b, p)], [mul_bc, tensor(w, w, lambda a, b: a * b, p)]]
        polys, chal = sumcheck_gkr_prove(fbc, claim, t, p)
        proofs.append(dict(claimed_sum=claim, polys=polys, challenges=chal))
        if L < len(layers) - 1:
            mid = len(chal) // 2
            rb, rc = chal[:mid], chal[mid:]
            wb, wc = evaluate(w, rb, p), evaluate(w, rc, p)
            wbs.append(wb)
            wcs.append(wc)
            t.append(be32(wb))
            alpha = t.challenge(p)
            t.append(be32(wc))
            beta = t.challenge(p)
            claim = (alpha * wb + beta * wc) % p
    return dict(circuit_output=evs[0], claimed_sum=claim, sumcheck_proofs=proofs, wb=wbs, wc=wcs)


def gkr_verify(layers, proof, inputs, p):
    t = Transcript()
    w0 = list(proof["circuit_output"])
    if len(w0) == 1:
        w0.append(0)
    t.append(b"".join(be32(v) for v in w0))
    ra = t.challenge(p)
    claim = evaluate(w0, [ra], p)
    alpha = beta = 0
    prev = []
    for L, layer in enumerate(layers):
        sp = proof["sumcheck_proofs"][L]
        if claim != sp["claimed_sum"]:
            return False
        ok, chal, last = sumcheck_gkr_verify(sp["claimed_sum"], sp["polys"], t, p)
        if not ok:
            return False
        mid = len(chal) // 2
        if L < len(layers) - 1:
            wb, wc = proof["wb"][L], proof["wc"][L]
        else:
            wb, wc = evaluate(inputs, chal[:mid], p), evaluate(inputs, chal[mid:], p)
        add, mul = add_mul_mle(layer, L)
        if L == 0:
            add_bc, mul_bc = partial_evaluate(add, 0, ra, p), partial_evaluate(mul, 0, ra, p)
        else:
            k = len(prev) // 2
            add_bc = [(alpha * x + beta * y) % p for x, y in zip(_fold_all(add, prev[:k], p), _fold_all(add, prev[k:], p))]
            mul_bc = [(alpha * x + beta * y) % p for x, y in zip(_fold_all(mul, prev[:k], p), _fold_all(mul, prev[k:], p))]
        expect = (evaluate(add_bc, chal, p) * (wb + wc) + evaluate(mul_bc, chal, p) * (wb * wc)) % p
        if expect != last:
            return False
        prev = chal
        t.append(be32(wb))
        alpha = t.challenge(p)
        t.append(be32(wc))
        beta = t.challenge(p)
        claim = (alpha * wb + beta * wc) % p
    return True


# ---- BLS12-381 G1, affine big-int arithmetic ------------------------------------------------------
# ---- the dense definition generalised to per-layer widths (independent oracle of the sparse GKR prover) --------------------
# The reference ties a layer's width to its index (arithmetic_circuit.rs:166-178: layer i has i bits of `a` and i + 1 bits each
# of `b`, `c`).  Here layer l has out_bits[l] bits of `a` and in_bits[l] = out_bits[l + 1] (log2 #inputs for the last layer) bits
# each of `b` and `c`; everything else is the reference's definition, term for term: the wiring predicates are DENSE 0/1 tables
# indexed a || b || c MSB first (:126-163, :180-196), folded by the output challenges / alpha-beta-combined over rb, rc
# (gkr_protocol.rs:60-82, utils.rs:23-68), f(b, c) = add(b,c) (W(b) + W(c)) + mul(b,c) (W(b) W(c)) from DENSE outer sums /
# products (utils.rs:8-21), and the degree-2 sumcheck of sumcheck_gkr_protocol.rs:24-67 over its 2 in_bits variables.  With
# out_bits[0] = k0 the output claim takes k0 successive challenges (the reference has k0 = 1).  Nothing here knows about gate
# lists, eq tables or the two-phase split of the product's linear-time prover.
def circuit_evaluate_wide(layers, out_bits, inputs, p):
    cur, evs = list(inputs), [list(inputs)]
    for l in range(len(layers) - 1, -1, -1):              # arithmetic_circuit.rs:72
        res = [0] * (1 << out_bits[l])
        for (lft, rgt, o, op) in layers[l]:
            res[o] = (res[o] + (cur[lft] + cur[rgt] if op == ADD else cur[lft] * cur[rgt])) % p   # :90-96
        cur = res
        evs.append(list(cur))
    evs.reverse()
    return evs


def add_mul_mle_wide(layer, ka, kb):
    n = 1 << (ka + 2 * kb)
    add, mul = [0] * n, [0] * n
    for (lft, rgt, o, op) in layer:
        (add if op == ADD else mul)[(o << (2 * kb)) | (lft << kb) | rgt] = 1      # a || b || c, MSB first (:180-196)
    return add, mul


def gkr_prove_wide(layers, out_bits, inputs, p):
    nl = len(layers)
    in_bits = [out_bits[l + 1] if l + 1 < nl else len(inputs).bit_length() - 1 for l in range(nl)]
    evs = circuit_evaluate_wide(layers, out_bits, inputs, p)
    t = Transcript()
    w0 = list(evs[0])
    t.append(b"".join(be32(v) for v in w0))               # gkr_protocol.rs:49
    ra = [t.challenge(p) for _ in range(out_bits[0])]     # :50, one challenge per output variable
    claim = evaluate(w0, ra, p)                           # :51
    alpha = beta = 0
    rb, rc = [], []
    out = dict(circuit_output=evs[0], output_challenges=ra, layer_claims=[], coeffs=[], challenges=[], wb=[], wc=[])
    for l, layer in enumerate(layers):
        add, mul = add_mul_mle_wide(layer, out_bits[l], in_bits[l])
        if l == 0:
            add_bc, mul_bc = _fold_all(add, ra, p), _fold_all(mul, ra, p)                   # :60-72
        else:                                             # utils.rs:23-68
            add_bc = [(alpha * x + beta * y) % p for x, y in zip(_fold_all(add, rb, p), _fold_all(add, rc, p))]
            mul_bc = [(alpha * x + beta * y) % p for x, y in zip(_fold_all(mul, rb, p), _fold_all(mul, rc, p))]
        w = evs[l + 1]
        fbc = [[add_bc, tensor(w, w, lambda a, b: a + b, p)], [mul_bc, tensor(w, w, lambda a, b: a * b, p)]]   # utils.rs:8-21
        out["layer_claims"].append(claim)
        polys, chal = sumcheck_gkr_prove(fbc, claim, t, p)                                  # :99
        out["coeffs"] += polys
        out["challenges"] += chal
        if l < nl - 1:                                    # :109-133
            mid = len(chal) // 2
            rb, rc = chal[:mid], chal[mid:]
            wb, wc = evaluate(w, rb, p), evaluate(w, rc, p)
            out["wb"].append(wb)
            out["wc"].append(wc)
            t.append(be32(wb))
            alpha = t.challenge(p)
            t.append(be32(wc))
            beta = t.challenge(p)
            claim = (alpha * wb + beta * wc) % p
    out["claimed_sum"] = claim
    return out


Q = P["bls12_381_fq"]
R = P["bls12_381_fr"]
G1 = (0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
      0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1)


def g1_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    if a[0] == b[0]:
        if (a[1] + b[1]) % Q == 0:
            return None
        lam = 3 * a[0] * a[0] * pow(2 * a[1], -1, Q) % Q
    else:
        lam = (b[1] - a[1]) * pow(b[0] - a[0], -1, Q) % Q
    x = (lam * lam - a[0] - b[0]) % Q
    return x, (lam * (a[0] - x) - a[1]) % Q


def g1_mul(a, k):
    acc = None
    while k:
        if k & 1:
            acc = g1_add(acc, a)
        a = g1_add(a, a)
        k >>= 1
    return acc


def lagrange_basis(taus):                                 # trusted_setup.rs:24-49
    n = len(taus)
    out = []
    for idx in range(1 << n):
        e = 1
        for i in range(n):
            e = e * (taus[i] if (idx >> (n - 1 - i)) & 1 else 1 - taus[i]) % R
        out.append(e)
    return out


def kzg_setup_g1(taus):
    return [g1_mul(G1, e) for e in lagrange_basis(taus)]


def kzg_commit(values, points):
    acc = None
    for v, b in zip(values, points):
        acc = g1_add(acc, g1_mul(b, v % R))
    return acc


def kzg_open(values, points, opening):
    v = evaluate(values, opening, R)
    sub = [(x - v) % R for x in values]
    proofs = []
    for i, x in enumerate(opening):
        h = len(sub) // 2
        q = [(sub[h + k] - sub[k]) % R for k in range(h)]
        blown = q * (1 << (i + 1))                        # blow_up :181-209
        proofs.append(kzg_commit(blown, points))
        sub = partial_evaluate(sub, 0, x, R)
    return v, proofs
