// zkmle_sumcheck.hip -- C ABI: transcript, basic sumcheck prover/verifier, SumPolynomial kernels, the GKR sumcheck
// prover/verifier and the device-resident rounds handle (zk_rounds_*) of the sharded provers.  Tables stay in HBM.  The host
// absorbs the table-sized transcript input (pipelined with the GPU's byte conversion) and hands the sponge to the device;
// every round -- reduction, round message, Keccak absorb + sample, challenge, fold -- then runs there (dev_transcript.cuh),
// with one synchronisation per sumcheck.
#include <string.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <map>
#include <mutex>
#include <thread>
#include <vector>

#include "context.h"
#include "dev_transcript.cuh"
#include "basic_multi.cuh"
#include "fold_multi.h"
#include "sumcheck_kernels.cuh"
#include "transcript.h"
#include "univariate.h"

using namespace zk;

namespace {

template <class F> Fe<F> load_el(const uint64_t *src) {
    Fe<F> e;
    memcpy(e.l, src, 4 * F::N);
    return e;
}
template <class F> void store_el(uint64_t *dst, const Fe<F> &e) { memcpy(dst, e.l, 4 * F::N); }

// host-clock split of the last prover call on this thread (zk_sumcheck_last_stats)
thread_local zk_sumcheck_stats g_stats{};
inline double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct DevBuf {   // RAII device allocation
    void *p = nullptr;
    ~DevBuf() { pool_free(p); }                     // per-call scratch from the caching pool (context.h)
    int alloc(size_t bytes) { return pool_alloc(bytes, &p); }
};

// absorb convert_to_bytes(table) (evaluation_form.rs:35-43) chunk by chunk: the GPU converts Montgomery -> canonical
// big-endian and copies chunk k + 1 into a pinned buffer while the host sponge absorbs chunk k.  The sponge is
// sequential (~0.64 GB/s on one core of the GPU box's host); the copies (0.2 s of 1.06 s at 2^24 when they were
// synchronous pageable copies) now hide behind it.
template <class F> int absorb_table(Transcript &t, const void *dptr, size_t len) {
    const size_t esz = 4 * F::N;
    const size_t chunk = (size_t)1 << 18;   // elements per chunk (8 MiB for 32-byte elements)
    const size_t cl = len < chunk ? len : chunk;
    DevBuf tmp[2];
    void *host[2];
    ZK_TRY(tmp[0].alloc(cl * esz));
    ZK_TRY(tmp[1].alloc(cl * esz));
    ZK_TRY(pinned_pair(chunk * 4 * Fq381::N, host));
    hipEvent_t ev[2];
    ZK_HIP(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
    ZK_HIP(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
    int rc = ZK_OK;
    size_t pending_n = 0;
    int pending = -1;
    for (size_t off = 0, k = 0; off < len && rc == ZK_OK; off += chunk, k++) {
        const int b = (int)(k & 1);
        const size_t n = len - off < chunk ? len - off : chunk;
        elementwise_kernel<F, OP_TO_CANONICAL_BE><<<grid_for(n), kBlock, 0, cur_stream()>>>((const char *)dptr + off * esz, nullptr, tmp[b].p, n, fe_zero<F>());
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(host[b], tmp[b].p, n * esz, hipMemcpyDeviceToHost, cur_stream());
        if (e == hipSuccess) e = hipEventRecord(ev[b], cur_stream());
        if (e == hipSuccess && pending >= 0) {                        // hash the previous chunk while this one is in flight
            e = hipEventSynchronize(ev[pending]);
            if (e == hipSuccess) t.append((const uint8_t *)host[pending], pending_n * esz);
        }
        if (e != hipSuccess) { set_last_error(hipGetErrorString(e)); rc = ZK_E_HIP; break; }
        pending = b;
        pending_n = n;
    }
    if (rc == ZK_OK && pending >= 0) {
        hipError_t e = hipEventSynchronize(ev[pending]);
        if (e == hipSuccess) t.append((const uint8_t *)host[pending], pending_n * esz);
        else { set_last_error(hipGetErrorString(e)); rc = ZK_E_HIP; }
    }
    (void)hipStreamSynchronize(cur_stream());
    (void)hipEventDestroy(ev[0]);
    (void)hipEventDestroy(ev[1]);
    return rc;
}

template <class F> int download_elems(const void *d, size_t n, Fe<F> *out) {
    ZK_HIP(zk::memcpy_on_stream(out, d, n * 4 * F::N, hipMemcpyDeviceToHost));
    return ZK_OK;
}

// The host side of the mailbox is served by a helper thread that makes NO HIP call: one per proving thread, created at its first
// proof, spinning only while a proof is in flight.  The proving thread itself keeps enqueueing kernels and may block inside the
// HIP runtime (an allocation, another thread's hipFree waiting for the device to drain ...): a kernel waiting for its challenge
// must never depend on a thread that can be stuck behind that very kernel.
class ServiceWorker {
  public:
    ServiceWorker() : th_([this] { run(); }) {}
    ~ServiceWorker() {
        { std::lock_guard<std::mutex> lk(mu_); quit_ = true; }
        cv_.notify_one();
        th_.join();
    }
    void start(std::function<void()> job) {                    // the previous job has finished (callers wait for it)
        { std::lock_guard<std::mutex> lk(mu_); job_ = std::move(job); pending_.store(true, std::memory_order_release); }
        cv_.notify_one();
    }
  private:
    void run() {
        for (;;) {
            std::function<void()> job;
            {   // proofs come in bursts: keep polling for ~0.3 ms after a job before sleeping (a wake-up costs tens of microseconds,
                // a small proof lasts 200)
                const double t0 = now_ms();
                while (!pending_.load(std::memory_order_acquire) && now_ms() - t0 < 0.3) {}
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [this] { return pending_.load(std::memory_order_acquire) || quit_; });
                if (quit_ && !pending_.load()) return;
                job = std::move(job_);
                pending_.store(false, std::memory_order_release);
            }
            job();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_;
    std::function<void()> job_;
    std::atomic<bool> pending_{false};
    bool quit_ = false;
    std::thread th_;                                           // last member: started after the others exist
};
static ServiceWorker &service_worker() {
    static thread_local ServiceWorker w;
    return w;
}
static std::atomic<int> g_stall_service_ms{0};               // zk_debug_stall_service_once
// The last proof of this thread saw every kernel's last post: nothing on ITS stream looks at ITS mailbox any more.  The mailbox is per (thread,
// device) and the kernels that may still poll it sit on the stream that proof ran on, so the mark names both: a proof on another device or
// another stream of this thread does not make a mailbox it never touched safe to reset.
struct CleanProofMark { const void *mbox = nullptr; hipStream_t stream = nullptr; bool clean = false; };
static thread_local CleanProofMark g_last_proof;
static thread_local int g_host_rounds_active = 0;           // one proof at a time per proving thread owns the mailbox and the worker

// Whether the transcript step of a round runs on the host thread that drives the proof (dev_transcript.cuh HostMailbox: default) or on
// the device (ZK_HOST_TRANSCRIPT=0; always for the zk_rounds_* handle of the multi-GPU provers, whose step sits behind an all-reduce).
static bool host_transcript_default() {
    static const bool v = [] { const char *e = getenv("ZK_HOST_TRANSCRIPT"); return !(e && e[0] == '0'); }();
    return v;
}

// Rounds of one proof (dev_transcript.cuh): one pooled block holds the sponge, the interpolation basis and the proof slots.
// Device mode: uploaded once before the first round and downloaded once after the last, every transcript step on the device.
// Host mode: the kernels post each round's evaluations to the mailbox and wait for the challenge; this object runs the transcript
// step (service()) on the caller's host sponge, in request order, and keeps the proof slots on the host; the device slots keep
// what kernels read (challenges, final values, link values).
template <class F> struct DeviceRounds {
    static constexpr size_t kHead = 256;               // DevSponge, padded
    DevBuf buf;
    void *syncw = nullptr;                             // the arrival counter of the passes that run their exchange themselves (basic_multi.cuh MultiFin)
    size_t nbasis = 0, nslots = 0;
    std::vector<uint8_t> host;
    // host mode
    bool host_mode = false, owns_service = false;
    Transcript *htr = nullptr;
    HostMailbox *mb = nullptr, *mb_dev = nullptr;
    std::vector<Fe<F>> hs;                             // the proof slots, host copy
    std::vector<Fe<F>> hbasis;                         // basis[i * npts + d], stored form
    struct Req { int kind, mode, npts, with_claim, derive1, ntab; size_t claim_slot, msg_slot, chal_slot, fin_slot, s[7]; };
    enum { kRound = 0, kFinal = 1, kLink = 2, kMulti = 3, kRound2 = 4 };
    std::vector<Req> reqs;                             // appended by the proving thread, consumed in order by the service thread
    std::mutex req_mu;
    size_t served = 0;                                 // service thread only
    std::atomic<bool> closing{false}, finished{true};
    std::atomic<int> service_rc{ZK_OK};
    Fe<F> running_claim = fe_zero<F>();                // p_k(r_k) of the last served round = e_{k+1}(0) + e_{k+1}(1)

    char *base() const { return (char *)buf.p; }
    DevSponge *sponge() const { return (DevSponge *)buf.p; }
    void *basis() const { return base() + kHead; }
    void *proof() const { return base() + kHead + nbasis * 4 * F::N; }
    void *slot_ptr(size_t s) const { return (char *)proof() + s * 4 * F::N; }
    size_t bytes() const { return kHead + (nbasis + nslots) * 4 * F::N; }
    // behind the slots: the current challenge as a uniform multiplier (ufield.cuh UniMul), written by the exchange that receives it and read by
    // the fused round that folds by it.  One area per proof: a fused round has read it before its last workgroup writes the next one.
    static constexpr size_t kExpBytes = 512;
    uint32_t *uexp() const { return (uint32_t *)(base() + ((bytes() + 15) & ~(size_t)15)); }
    int init(Transcript &tr, const std::vector<Fe<F>> &basis_flat, size_t slots, bool allow_host = true) {
        static_assert(sizeof(DevSponge) <= kHead, "sponge header");
        nbasis = basis_flat.size();
        nslots = slots;
        ZK_TRY(buf.alloc(((bytes() + 15) & ~(size_t)15) + kExpBytes));
        ZK_TRY(sync_words(&syncw));
        host_mode = allow_host && host_transcript_default() && g_host_rounds_active == 0;
        void *h = nullptr, *d = nullptr;
        if (host_mode && host_mailbox(&h, &d) != ZK_OK) host_mode = false;   // no coherent pinned page on this system: the device runs the step
        if (host_mode) {
            g_host_rounds_active++;
            owns_service = true;
            mb = (HostMailbox *)h;
            mb_dev = (HostMailbox *)d;
            // nothing of an earlier proof may still be looking at the mailbox: a proof that ended cleanly has seen every kernel's last post
            // (collect); anything else on this thread's stream -- a proof that failed, other work -- is waited for
            if (!(g_last_proof.clean && g_last_proof.mbox == h && g_last_proof.stream == cur_stream())) ZK_HIP(stream_wait_idle());
            g_last_proof.clean = false;
            memset((void *)mb, 0, offsetof(HostMailbox, big));
            htr = &tr;
            hs.assign(nslots, fe_zero<F>());
            hbasis = basis_flat;
            reqs.clear();
            served = 0;
            closing = false;
            finished = false;
            service_rc = ZK_OK;
            service_worker().start([this] { service_loop(); });
            return ZK_OK;
        }
        host.assign(kHead + nbasis * 4 * F::N, 0);
        DevSponge sp{};
        tr.sponge().export_state(sp.a, &sp.fill);
        memcpy(host.data(), &sp, sizeof sp);
        if (nbasis) memcpy(host.data() + kHead, basis_flat.data(), nbasis * 4 * F::N);
        ZK_HIP(zk::memcpy_on_stream(buf.p, host.data(), host.size(), hipMemcpyHostToDevice));
        return ZK_OK;
    }
    RoundCtx ctx(int npts, int mode) const { return RoundCtx{npts, mode, sponge(), basis(), proof(), host_mode ? mb_dev : nullptr}; }

    // ---- host mode: the transcript steps, in request order ----
    static Fe<F> mb_get(const uint32_t *src) {
        Fe<F> e;
        for (int i = 0; i < F::N; i++) e.l[i] = __atomic_load_n(src + i, __ATOMIC_RELAXED);
        return e;
    }
    static void mb_put(uint32_t *dst, const Fe<F> &e) {
        for (int i = 0; i < F::N; i++) __atomic_store_n(dst + i, e.l[i], __ATOMIC_RELAXED);
    }
    // the answer of a round: the challenge, in the polled line (limbs first, the two tags last: dev_transcript.cuh) and in `chal`
    void publish_challenge(uint64_t seq, const Fe<F> &r) {
        mb_put(mb->chal, r);
        for (int i = 0; i < F::N; i++) __atomic_store_n(&mb->ans[1 + i], r.l[i], __ATOMIC_RELAXED);
        __atomic_store_n(&mb->ans[0], (uint32_t)seq, __ATOMIC_RELEASE);
        __atomic_store_n(&mb->ans[15], (uint32_t)seq, __ATOMIC_RELEASE);
        __atomic_store_n(&mb->cpu_seq, seq, __ATOMIC_RELEASE);
    }
    // The kernel that posted the evaluations is spinning until the challenge is back, so only what the challenge depends on runs before
    // the answer goes out; the proof's copy of the message and the running claim are worked out while the GPU is already in the next round.
    void serve_round(const Req &q, uint64_t seq) {
        Fe<F> ev[kMaxPts];
        for (int t = 0; t < q.npts; t++) ev[t] = mb_get(mb->ev + 12 * t);
        if (q.derive1) ev[1] = fe_sub<F>(running_claim, ev[0]);                // the producer skipped the point 1 (dev_transcript.cuh kDerive1)
        Fe<F> r;
        if (q.mode == 0) {                                                     // prover.rs:50-58: the two half sums, big-endian
            if (q.with_claim) {
                hs[q.claim_slot] = fe_add<F>(ev[0], ev[1]);                    // :28
                htr->template append_be<F>(hs[q.claim_slot]);                  // :40-41
            }
            for (int t = 0; t < q.npts; t++) htr->template append_be<F>(ev[t]);
            r = htr->template random_challenge_as_field_element<F>();
            publish_challenge(seq, r);
            for (int t = 0; t < q.npts; t++) hs[q.msg_slot + t] = ev[t];
            running_claim = fe_add<F>(ev[0], fe_mul<F>(r, fe_sub<F>(ev[1], ev[0])));
        } else {                                                               // sumcheck_gkr_protocol.rs:46-55: Lagrange coefficients, little-endian
            if (q.with_claim) htr->template append_be<F>(hs[q.claim_slot]);    // :35
            // coefficient d as the canonical integer straight away: evaluations (stored form) x the basis as canonical integers
            const size_t n2 = (size_t)q.npts * q.npts;
            uint8_t bytes[kMaxPts * 4 * F::N];
            for (int d = 0; d < q.npts; d++) {
                Fe<F> cc = fe_mul<F>(ev[0], hbasis[n2 + d]);
                for (int i = 1; i < q.npts; i++) cc = fe_add<F>(cc, fe_mul<F>(ev[i], hbasis[n2 + (size_t)i * q.npts + d]));
                memcpy(bytes + (size_t)d * 4 * F::N, cc.l, 4 * F::N);          // univariate_to_bytes :145-150: little-endian limbs as they lie (LE host)
            }
            htr->append(bytes, (size_t)q.npts * 4 * F::N);
            r = htr->template random_challenge_as_field_element<F>();
            publish_challenge(seq, r);
            Fe<F> c[kMaxPts];
            for (int d = 0; d < q.npts; d++) {
                c[d] = fe_mul<F>(ev[0], hbasis[d]);
                for (int i = 1; i < q.npts; i++) c[d] = fe_add<F>(c[d], fe_mul<F>(ev[i], hbasis[(size_t)i * q.npts + d]));
                hs[q.msg_slot + d] = c[d];
            }
            Fe<F> acc = c[q.npts - 1];
            for (int d = q.npts - 2; d >= 0; d--) acc = fe_add<F>(fe_mul<F>(acc, r), c[d]);
            running_claim = acc;
        }
        hs[q.chal_slot] = r;
    }
    // one transcript step of a two-factor GKR round from its evaluations at 0, 1, infinity (sumcheck_gkr_protocol.rs:46-55): absorbs the coefficients
    // (canonical, little-endian) and samples the challenge -- only what the challenge depends on; keep_gkr3 does the bookkeeping afterwards
    Fe<F> step_gkr3(const Fe<F> (&ev)[3]) {
        const size_t n2 = 9;
        uint8_t bytes[3 * 4 * F::N];
        for (int d = 0; d < 3; d++) {
            Fe<F> cc = fe_mul<F>(ev[0], hbasis[n2 + d]);
            for (int i = 1; i < 3; i++) cc = fe_add<F>(cc, fe_mul<F>(ev[i], hbasis[n2 + (size_t)i * 3 + d]));
            memcpy(bytes + (size_t)d * 4 * F::N, cc.l, 4 * F::N);
        }
        htr->append(bytes, sizeof bytes);
        return htr->template random_challenge_as_field_element<F>();
    }
    // the proof's copy of that round (coefficients in the stored form, the challenge) and the running claim
    void keep_gkr3(const Fe<F> (&ev)[3], const Fe<F> &r, size_t msg_slot, size_t chal_slot) {
        Fe<F> c[3];
        for (int d = 0; d < 3; d++) {
            c[d] = fe_mul<F>(ev[0], hbasis[d]);
            for (int i = 1; i < 3; i++) c[d] = fe_add<F>(c[d], fe_mul<F>(ev[i], hbasis[(size_t)i * 3 + d]));
            hs[msg_slot + d] = c[d];
        }
        hs[chal_slot] = r;
        running_claim = fe_add<F>(fe_mul<F>(fe_add<F>(fe_mul<F>(c[2], r), c[1]), r), c[0]);
    }
    // TWO rounds of a two-factor GKR sumcheck from the nine sums the tail posts (dev_transcript.cuh, sumcheck_tail_kernel): round A's evaluations are sums of
    // them; round B's are polynomials in round A's challenge with those sums as coefficients.  q.s[0] = slots per round.  The kernel is spinning: both
    // challenges go out before any bookkeeping.
    void serve_round2(const Req &q, uint64_t seq) {
        Fe<F> S[9];
        for (int t = 0; t < 9; t++) S[t] = mb_get(mb->big + 12 * t);
        const Fe<F> &P0 = S[0], &P1 = S[1], &Q0 = S[2], &Q1 = S[3], &D0 = S[4], &D1 = S[5], &EE = S[6], &FF = S[7], &GG = S[8];
        const size_t per = q.s[0];
        const Fe<F> evA[3] = {fe_add<F>(P0, P1), fe_add<F>(Q0, Q1), fe_add<F>(D0, D1)};
        const Fe<F> rA = step_gkr3(evA);
        auto quad = [&](const Fe<F> &k0, const Fe<F> &k1, const Fe<F> &k2) {          // k0 + rA (k1 - k0 - k2) + rA^2 k2
            const Fe<F> mid = fe_sub<F>(fe_sub<F>(k1, k0), k2);
            return fe_add<F>(fe_mul<F>(fe_add<F>(fe_mul<F>(k2, rA), mid), rA), k0);
        };
        const Fe<F> evB[3] = {quad(P0, Q0, D0), quad(P1, Q1, D1), quad(EE, FF, GG)};
        const Fe<F> rB = step_gkr3(evB);
        const Fe<F> both[2] = {rA, rB};
        for (int i = 0; i < 2; i++)
            for (int k = 0; k < F::N; k++) __atomic_store_n(&mb->ans8[i][1 + k], both[i].l[k], __ATOMIC_RELAXED);
        for (int i = 0; i < 2; i++) {                                          // the answer lines' tags last (dev_transcript.cuh)
            __atomic_store_n(&mb->ans8[i][0], (uint32_t)seq, __ATOMIC_RELEASE);
            __atomic_store_n(&mb->ans8[i][15], (uint32_t)seq, __ATOMIC_RELEASE);
        }
        keep_gkr3(evA, rA, q.msg_slot, q.chal_slot);
        keep_gkr3(evB, rB, q.msg_slot + per, q.chal_slot + per);
    }
    // basic sumcheck, q.npts rounds from the 2^npts segment sums of the current table (basic_multi.cuh): the basic sumcheck on the
    // table of the sums, S -- round i sends its two half sums and folds its top variable by the challenge
    void serve_multi(const Req &q, uint64_t seq) {
        const int m = q.npts;
        const size_t per = q.s[0];
        size_t n = (size_t)1 << m;
        Fe<F> S[1 << kMultiMax];
        for (size_t t = 0; t < n; t++) S[t] = mb_get(mb->big + 12 * t);
        for (int i = 0; i < m; i++) {
            const size_t half = n / 2;
            Fe<F> a0 = S[0], a1 = S[half];                                     // prover.rs:50 (split_polynomial_and_sum_each :74-89)
            for (size_t j = 1; j < half; j++) { a0 = fe_add<F>(a0, S[j]); a1 = fe_add<F>(a1, S[half + j]); }
            if (i == 0 && q.with_claim) {
                hs[q.claim_slot] = fe_add<F>(a0, a1);                          // :28
                htr->template append_be<F>(hs[q.claim_slot]);                  // :40-41
            }
            hs[q.msg_slot + per * i] = a0; hs[q.msg_slot + per * i + 1] = a1;
            htr->template append_be<F>(a0);                                    // :52-55
            htr->template append_be<F>(a1);
            const Fe<F> r = htr->template random_challenge_as_field_element<F>();   // :58
            hs[q.chal_slot + per * i] = r;
            running_claim = fe_add<F>(a0, fe_mul<F>(r, fe_sub<F>(a1, a0)));
            for (int k = 0; k < F::N; k++) __atomic_store_n(&mb->ans8[i][1 + k], r.l[k], __ATOMIC_RELAXED);
            for (size_t j = 0; j < half; j++) S[j] = fe_add<F>(S[j], fe_mul<F>(r, fe_sub<F>(S[half + j], S[j])));   // :61-63
            n = half;
        }
        for (int i = 0; i < m; i++) {                                          // the answer lines' tags last (dev_transcript.cuh)
            __atomic_store_n(&mb->ans8[i][0], (uint32_t)seq, __ATOMIC_RELEASE);
            __atomic_store_n(&mb->ans8[i][15], (uint32_t)seq, __ATOMIC_RELEASE);
        }
    }
    void serve_link(const Req &q) {                                            // gkr_protocol.rs:125-132
        const Fe<F> wb = hs[q.s[0]], wc = hs[q.s[1]];
        htr->template append_be<F>(wb);
        const Fe<F> al = htr->template random_challenge_as_field_element<F>();
        htr->template append_be<F>(wc);
        const Fe<F> be = htr->template random_challenge_as_field_element<F>();
        hs[q.s[2]] = wb; hs[q.s[3]] = wc; hs[q.s[4]] = al; hs[q.s[5]] = be;
        hs[q.s[6]] = fe_add<F>(fe_mul<F>(al, wb), fe_mul<F>(be, wc));
        running_claim = hs[q.s[6]];
        mb_put(mb->aux[0], al);
        mb_put(mb->aux[1], be);
    }
    // the service thread's job for this proof: answer the requests in order as the kernels post them, until the proving thread has
    // closed the list and everything is answered.  Bounded: a kernel that gave up sets `aborted`; 8 s without a post ends the job.
    void service_loop() {
        // fault injection (zk_debug_stall_service_once): the host side of ONE proof goes deaf for a while, as a descheduled or dying
        // process would -- the kernels' spin budget must run out, every kernel must still end, the call must fail and say why
        const int stall_ms = g_stall_service_ms.exchange(0);
        if (stall_ms > 0) std::this_thread::sleep_for(std::chrono::milliseconds(stall_ms));
        for (;;) {
            Req q;
            bool have;
            // `closing` is read BEFORE the list: every request is pushed before the list is closed, so a list seen empty after
            // `closing` was seen set really is finished (the other order can miss the last requests if this thread is descheduled
            // between the two reads)
            const bool was_closing = closing.load(std::memory_order_acquire);
            {
                std::lock_guard<std::mutex> lk(req_mu);
                have = served < reqs.size();
                if (have) q = reqs[served];
            }
            if (!have) {
                if (was_closing) break;
                continue;
            }
            const uint64_t seq = served + 1;
            if (q.kind != kLink) {
                const double t0 = now_ms();
                unsigned long polls = 0;
                bool ok = true;
                while (__atomic_load_n(&mb->gpu_seq, __ATOMIC_ACQUIRE) < seq) {
                    if ((++polls & 0xffff) == 0 && (__atomic_load_n(&mb->aborted, __ATOMIC_RELAXED) || now_ms() - t0 > 8000.0)) { ok = false; break; }
                }
                if (!ok) { service_rc = ZK_E_HIP; break; }
            }
            if (q.kind == kRound) serve_round(q, seq);           // publishes its answer itself, as early as it can
            else {
                if (q.kind == kMulti) serve_multi(q, seq);
                else if (q.kind == kRound2) serve_round2(q, seq);
                else if (q.kind == kLink) serve_link(q);
                else for (int k = 0; k < q.ntab; k++) hs[q.fin_slot + k] = mb_get(mb->fin + 12 * k);
                __atomic_store_n(&mb->cpu_seq, seq, __ATOMIC_RELEASE);
            }
            served++;
        }
        // on failure, release every kernel that may still be waiting (they get a stale challenge; the call reports the error)
        if (service_rc.load() != ZK_OK) __atomic_store_n(&mb->cpu_seq, ~(uint64_t)0 >> 1, __ATOMIC_RELEASE);
        finished.store(true, std::memory_order_release);
    }
    void push_req(const Req &q) {
        std::lock_guard<std::mutex> lk(req_mu);
        reqs.push_back(q);
    }
    size_t nreq() {
        std::lock_guard<std::mutex> lk(req_mu);
        return reqs.size();
    }
    int close_service() {                                   // no further requests: wait for the service thread to answer the rest
        if (host_mode && !finished.load(std::memory_order_acquire)) {
            closing.store(true, std::memory_order_release);
            const double t0 = now_ms();
            while (!finished.load(std::memory_order_acquire)) {
                std::this_thread::yield();                   // the service thread may need this core
                if (now_ms() - t0 > 60000.0) {               // its own limits are far below this: only a lost service thread (fork) gets here
                    set_last_error("host-assisted transcript step: the service thread did not finish");
                    service_rc = ZK_E_HIP;
                    break;
                }
            }
        }
        if (owns_service) { owns_service = false; g_host_rounds_active--; }
        return service_rc.load();
    }

    // derive_prev = 1: the producer skipped the point 1; it is derived from the previous round's message, whose slots precede
    // this round's by `per` (dev_transcript.cuh kDerive1)
    int launch_finish(const void *partials, size_t count, int npts, int mode, int with_claim, size_t claim_slot, size_t msg_slot,
                      size_t chal_slot, int derive_prev = 0, size_t per = 0) {
        FinishArgs a{};
        a.partials = partials; a.count = count; a.ctx = ctx(npts, mode); a.with_claim = with_claim; a.flags = derive_prev ? kDerive1 : 0;
        a.claim_slot = claim_slot; a.msg_slot = msg_slot; a.chal_slot = chal_slot;
        a.prev_msg_slot = msg_slot - per; a.prev_chal_slot = chal_slot - per;
        if (host_mode) {                                    // the derivation moves to the host with the rest of the step
            push_req(Req{kRound, mode, npts, with_claim, derive_prev, 0, claim_slot, msg_slot, chal_slot, 0, {0, 0, 0, 0, 0, 0, 0}});
            a.seq = nreq();
            a.flags = 0;
            derive_prev = 0;
        }
        size_t threads = (count + 63) / 64 * 64;          // one partial per lane up to 1024 lanes
        const size_t cap = (size_t)kFinishBlock - (derive_prev ? 64 : 0);
        if (threads > cap) threads = cap;
        if (derive_prev) threads += 64;                   // the helper wave
        sumcheck_finish_kernel<F><<<1, (int)threads, 0, cur_stream()>>>(a);
        ZK_HIP(hipGetLastError());
        return ZK_OK;
    }
    // host mode: the round's exchange runs in the last workgroup of the kernel that produces its evaluations (sumcheck_kernels.cuh RoundFin,
    // dev_transcript.cuh round_finish_in_producer) instead of in a finish kernel: registers the request, hands back what that launch needs.
    // `grid` = the producer's workgroups.
    int round_fin(int grid, int npts, int mode, int with_claim, size_t claim_slot, size_t msg_slot, size_t chal_slot, int derive_prev, RoundFin *out,
                  bool expand = false) {
        if (!host_mode) return ZK_E_ARG;
        push_req(Req{kRound, mode, npts, with_claim, derive_prev, 0, claim_slot, msg_slot, chal_slot, 0, {0, 0, 0, 0, 0, 0, 0}});
        unsigned group = 32;
        while ((unsigned)(grid + group - 1) / group > 200u) group *= 2;   // one 64-byte counter slot per group (context.h kSyncCounterBytes)
        *out = RoundFin{(unsigned *)syncw, (uint64_t *)((char *)syncw + kSyncCounterBytes), npts, derive_prev, group, nullptr, mb_dev, (uint64_t)nreq(), proof(),
                        chal_slot, expand ? uexp() : nullptr};
        return ZK_OK;
    }
    // a TWO-round exchange in the last workgroup of split2_round_kernel: nine sums out, the challenges of rounds `first` and `first + 1` back
    // (messages at msg_slot, msg_slot + per; challenges at chal_slot, chal_slot + per)
    int round2_fin(int grid, size_t msg_slot, size_t chal_slot, size_t per, RoundFin *out) {
        if (!host_mode) return ZK_E_ARG;
        push_req(Req{kRound2, 1, 3, 0, 0, 0, 0, msg_slot, chal_slot, 0, {per, 0, 0, 0, 0, 0, 0}});
        unsigned group = 32;
        while ((unsigned)(grid + group - 1) / group > 200u) group *= 2;
        *out = RoundFin{(unsigned *)syncw, (uint64_t *)((char *)syncw + kSyncCounterBytes), 9, 0, group, nullptr, mb_dev, (uint64_t)nreq(), proof(), chal_slot, nullptr, per};
        return ZK_OK;
    }
    // the same for a sharded table: the kernel's last workgroup leaves the npts sums as limb words for the all-reduce, nothing is posted
    RoundFin round_fin_limbs(int grid, int npts, int skip1, uint64_t *limbs_out) const {
        unsigned group = 32;
        while ((unsigned)(grid + group - 1) / group > 200u) group *= 2;
        return RoundFin{(unsigned *)syncw, (uint64_t *)((char *)syncw + kSyncCounterBytes), npts, skip1, group, limbs_out, nullptr, 0, nullptr, 0, nullptr};
    }
    // host mode, basic sumcheck (basic_multi.cuh): rounds round .. round + m - 1 from the 2^m segment sums an all-reduce has left as limb
    // words (sharded table); slots of round k: 1 + 3 k, 2 + 3 k (sums), 3 + 3 k (challenge), claim 0
    int launch_multi(const uint64_t *limbs_in, int m, size_t round) {
        if (!host_mode || !limbs_in || m < 1 || m > kMultiMax) return ZK_E_ARG;
        push_req(Req{kMulti, 0, m, round == 0 ? 1 : 0, 0, 0, 0, 1 + 3 * round, 3 + 3 * round, 0, {3, 0, 0, 0, 0, 0, 0}});
        MultiArgs a{limbs_in, m, mb_dev, (uint64_t)nreq(), proof(), 3 + 3 * round, 3};
        multi_finish_kernel<F><<<1, 64, 0, cur_stream()>>>(a);
        ZK_HIP(hipGetLastError());
        return ZK_OK;
    }
    // the same exchange run by the LAST workgroup of the pass that produces the segment sums (basic_multi.cuh multi_finish_in_producer):
    // registers the request and hands back what that launch needs.  `limbs_out` instead: a sharded table's pass leaves the sums as limbs
    // for the all-reduce and posts nothing (launch_multi with limbs_in follows the collective).
    int multi_fin(int m, size_t round, MultiFin *out) {
        if (!host_mode || m < 1 || m > kMultiMax) return ZK_E_ARG;
        push_req(Req{kMulti, 0, m, round == 0 ? 1 : 0, 0, 0, 0, 1 + 3 * round, 3 + 3 * round, 0, {3, 0, 0, 0, 0, 0, 0}});
        *out = MultiFin{(unsigned *)syncw, (uint64_t *)((char *)syncw + kSyncCounterBytes), m, nullptr, mb_dev, (uint64_t)nreq(), proof(), 3 + 3 * round, 3, nullptr};
        static const bool trace = [] { const char *e = getenv("ZK_PROOF_TRACE"); return e && e[0] == '1'; }();
        if (trace && nfin_traced < 8) {
            if (!fin_trace.p) { ZK_TRY(fin_trace.alloc(8 * 4 * 8)); ZK_HIP(hipMemsetAsync(fin_trace.p, 0, 8 * 4 * 8, cur_stream())); }
            out->trace = (uint64_t *)fin_trace.p + 4 * nfin_traced++;
        }
        return ZK_OK;
    }
    MultiFin multi_fin_limbs(int m, uint64_t *limbs_out) const { return MultiFin{(unsigned *)syncw, (uint64_t *)((char *)syncw + kSyncCounterBytes), m, limbs_out, nullptr, 0, nullptr, 0, 0, nullptr}; }
    // host mode, basic sumcheck: every round of a table of <= kTailLen entries (rounds round ..), one launch (basic_multi.cuh)
    int launch_basic_tail(const void *in, void *buf, size_t len, size_t round) {
        if (!host_mode || len < 2 || len > kTailLen) return ZK_E_ARG;
        BasicTailArgs a{in, buf, len, mb_dev, (uint64_t)nreq() + 1, proof(), 3 + 3 * round, 3};
        size_t rd = round;
        for (size_t cl = len; cl >= 2;) {
            const int lg = (int)ilog2(cl), m = lg < kTailMultiMax ? lg : kTailMultiMax;
            push_req(Req{kMulti, 0, m, rd == 0 ? 1 : 0, 0, 0, 0, 1 + 3 * rd, 3 + 3 * rd, 0, {3, 0, 0, 0, 0, 0, 0}});
            rd += (size_t)m;
            cl >>= m;
        }
        basic_tail_kernel<F><<<1, kTailBlock, 0, cur_stream()>>>(a);
        ZK_HIP(hipGetLastError());
        return ZK_OK;
    }
    // every remaining round of a sumcheck whose tables have <= kTailLen entries, one launch (dev_transcript.cuh)
    // first_evals: the tail starts with round `round`'s own evaluations and exchange (no round_evals launch in front of it); with_claim /
    // claim_slot as for that round
    int launch_tail(const SumPolyTables &tabs, void *buf0, void *buf1, int nprod, int nfac, size_t len, int mode, size_t round,
                    size_t msg_base, size_t chal_base, size_t per, size_t fin_slot, int first_evals = 0, int with_claim = 0, size_t claim_slot = 0,
                    int pending2 = 0) {
        TailArgs a{};
        a.tabs = tabs; a.buf[0] = buf0; a.buf[1] = buf1; a.nprod = nprod; a.ntab = nprod * nfac; a.len = len;
        a.ctx = ctx(nfac + 1, mode); a.round = round; a.msg_base = msg_base; a.chal_base = chal_base; a.per = per; a.fin_slot = fin_slot;
        a.first_evals = first_evals; a.with_claim = with_claim; a.claim_slot = claim_slot; a.pending2 = pending2;
        // ZK_TAIL_TWO_ROUNDS = the most (product, quad) pairs a two-round exchange takes (0: never; measurement / fallback switch)
        static const int two = [] { const char *e = getenv("ZK_TAIL_TWO_ROUNDS"); int v = e ? atoi(e) : 128; return v < 0 ? 0 : (v > 4096 ? 4096 : v); }();   // r4 sweep (0 / 64 / 128 / 256 / 512): 4 x 2^12 0.155 / 0.145 / 0.134 / 0.139 / 0.139 ms, depth-8 GKR 1.26 / 1.20 / 1.18 / 1.17 / 1.18
        a.two_rounds = (host_mode && nfac == 2 && mode == 1) ? two : 0;
        static const bool want_trace = [] { const char *e = getenv("ZK_TAIL_TRACE"); return e && e[0] == '1'; }();
        if (want_trace) {                                   // measurement only: per-phase stamps of this tail, printed after the launch
            ZK_TRY(tail_trace.alloc(6 * 16 * sizeof(uint64_t)));
            ZK_HIP(hipMemsetAsync(tail_trace.p, 0, 6 * 16 * sizeof(uint64_t), cur_stream()));
            a.trace = (uint64_t *)tail_trace.p;
        }
        if (host_mode) {
            a.seq0 = nreq() + 1;
            size_t rd = round;
            if (first_evals)                                 // round `round` itself: the request round_fin() would have registered for a launch of its own
                push_req(Req{kRound, mode, nfac + 1, with_claim, 0, 0, claim_slot, msg_base + per * rd, chal_base + per * rd, 0, {0, 0, 0, 0, 0, 0, 0}});
            // the kernel's own schedule (dev_transcript.cuh): single rounds while the tables are long, two rounds per exchange once (product, quad) pairs fit a wave
            size_t cl0 = len;
            if (pending2) { cl0 = len / 2; rd++; }           // the kernel folds by the first pending challenge before its first exchange
            for (size_t cl = cl0; cl >= 4;) {
                if (a.two_rounds && cl >= 8 && (size_t)nprod * (cl / 8) <= (size_t)a.two_rounds) {
                    push_req(Req{kRound2, mode, nfac + 1, 0, 0, 0, 0, msg_base + per * (rd + 1), chal_base + per * (rd + 1), 0, {per, 0, 0, 0, 0, 0, 0}});
                    rd += 2;
                    cl /= 4;
                } else {
                    rd++;
                    push_req(Req{kRound, mode, nfac + 1, 0, 0, 0, 0, msg_base + per * rd, chal_base + per * rd, 0, {0, 0, 0, 0, 0, 0, 0}});
                    cl /= 2;
                }
            }
            const bool want_fin = fin_slot != ~(size_t)0;
            push_req(Req{kFinal, mode, nfac + 1, 0, 0, want_fin ? nprod * nfac : 0, 0, 0, 0, want_fin ? fin_slot : 0, {0, 0, 0, 0, 0, 0, 0}});
        }
        if (nfac == 1) sumcheck_tail_kernel<F, 1><<<1, kTailBlock, 0, cur_stream()>>>(a);
        else if (nfac == 2) sumcheck_tail_kernel<F, 2><<<1, kTailBlock, 0, cur_stream()>>>(a);
        else sumcheck_tail_kernel<F, 3><<<1, kTailBlock, 0, cur_stream()>>>(a);
        ZK_HIP(hipGetLastError());
        if (a.trace && !host_mode) {                        // (host mode: printed by collect, after the service has answered)
            ZK_HIP(hipStreamSynchronize(cur_stream()));
            print_tail_trace(len);
        }
        traced_len = a.trace ? len : 0;
        return ZK_OK;
    }
    DevBuf tail_trace, fin_trace;
    size_t traced_len = 0;
    int nfin_traced = 0;
    void print_fin_trace() {
        uint64_t st[8 * 4];
        if (!fin_trace.p || zk::memcpy_on_stream(st, fin_trace.p, sizeof st, hipMemcpyDeviceToHost) != hipSuccess) return;
        for (int j = 0; j < nfin_traced; j++)
            fprintf(stderr, "[proof trace]   exchange %d: pass %.1f us until the last workgroup arrived, sums + reduce %.1f us, post -> answer %.1f us\n", j,
                    (double)(st[4 * j + 1] - st[4 * j]) * 0.01, (double)(st[4 * j + 2] - st[4 * j + 1]) * 0.01, (double)(st[4 * j + 3] - st[4 * j + 2]) * 0.01);
    }
    void print_tail_trace(size_t len) {
        uint64_t st[6 * 16];
        if (zk::memcpy_on_stream(st, tail_trace.p, sizeof st, hipMemcpyDeviceToHost) != hipSuccess) return;
        static const char *names[5] = {"fold+terms", "reduce", "post", "wait", "sync"};      // (a two-round iteration: fold, nine sums, post, wait, second fold)
        size_t cl = len;
        for (int j = 0; j < 16; j++) {
            if (st[6 * j] == 0 && st[6 * j + 5] == 0) continue;
            fprintf(stderr, "tail round %2d (len %4zu):", j, cl);
            for (int k = 0; k < 5; k++) fprintf(stderr, " %s %.2f us", names[k], (double)(st[6 * j + k + 1] - st[6 * j + k]) * 0.01);
            fprintf(stderr, " | total %.2f us\n", (double)(st[6 * j + 5] - st[6 * j]) * 0.01);
        }
    }
    // the layer link of a GKR proof in host mode: the host answers with alpha and beta once the two tails' final values are in
    int launch_link_host(size_t wb_src, size_t wc_src, size_t wb_slot, size_t wc_slot, size_t alpha_slot, size_t beta_slot, size_t claim_slot) {
        push_req(Req{kLink, 1, 0, 0, 0, 0, 0, 0, 0, 0, {wb_src, wc_src, wb_slot, wc_slot, alpha_slot, beta_slot, claim_slot}});
        LinkWaitArgs a{mb_dev, proof(), (uint64_t)nreq(), alpha_slot, beta_slot};
        gkr_link_wait_kernel<F><<<1, 64, 0, cur_stream()>>>(a);
        ZK_HIP(hipGetLastError());
        return ZK_OK;
    }
    // the single synchronisation of the sumcheck: proof slots + sponge back to the host
    int collect(Transcript &tr) {
        if (host_mode) {
            static const bool trace = [] { const char *e = getenv("ZK_PROOF_TRACE"); return e && e[0] == '1'; }();
            const double tc0 = now_ms();
            const int rc = close_service();
            const double tc1 = now_ms();
            // Every byte of the proof is on the host once the service thread has answered the last request, and no kernel touches the
            // mailbox after its last post (final posts are not acknowledged): the last kernel is retiring, its completion signal takes
            // another ~16 us to arrive (r3 host trace).  A clean proof does not wait for it -- the next call's work is ordered behind it on
            // the stream, and the next proof's mailbox reset races with nothing.
            if (rc != ZK_OK || mb->aborted || trace) ZK_HIP(stream_wait_idle());
            if (trace) fprintf(stderr, "[proof trace]   collect: service done +%.1f us, stream idle +%.1f us\n", (tc1 - tc0) * 1e3, (now_ms() - tc1) * 1e3);
            if (rc != ZK_OK || mb->aborted) (void)sync_words_reset();     // a pass that gave up may have left its arrival counter half-way
            if (rc != ZK_OK) { set_last_error("host-assisted transcript step: the device did not post a round (aborted or stalled)"); return rc; }
            if (mb->aborted) { set_last_error("host-assisted transcript step: a kernel gave up waiting for the host"); return ZK_E_HIP; }
            g_last_proof = CleanProofMark{(const void *)mb, cur_stream(), !trace};
            if (traced_len) print_tail_trace(traced_len);
            if (nfin_traced) print_fin_trace();
            return ZK_OK;                                   // `tr` is the sponge the steps ran on
        }
        host.resize(bytes());
        ZK_HIP(zk::memcpy_on_stream(host.data(), buf.p, bytes(), hipMemcpyDeviceToHost));
        DevSponge sp;
        memcpy(&sp, host.data(), sizeof sp);
        tr.sponge().import_state(sp.a, sp.fill);
        return ZK_OK;
    }
    Fe<F> slot(size_t s) const {
        if (host_mode) return hs[s];
        Fe<F> e;
        memcpy(e.l, host.data() + kHead + (nbasis + s) * 4 * F::N, 4 * F::N);
        return e;
    }
    ~DeviceRounds() {                                       // never leave a kernel waiting on a mailbox nobody serves, nor the service
        if (host_mode && !finished.load()) {                // thread inside an object that is going away
            (void)close_service();
            (void)hipStreamSynchronize(cur_stream());
        }
        if (owns_service) { owns_service = false; g_host_rounds_active--; }
    }
};

// ---- several rounds per pass (basic_multi.cuh) -------------------------------------------------------
static int multi_kmax() {                                  // ZK_BASIC_ROUNDS_PER_PASS = 1..8: for measurements (default 7)
    static const int v = [] {
        const char *e = getenv("ZK_BASIC_ROUNDS_PER_PASS");
        int k = e ? atoi(e) : 7;
        return k < 1 ? 1 : k > kMultiMax ? kMultiMax : k;
    }();
    return v;
}
// rounds of the next pass over a table of `global_len` entries (> kTailLen): never past the length the tail takes over at, and the rounds
// left are spread evenly over the passes they need (13 rounds = 7 + 6 rather than 8 + 5: the host's share of an exchange grows with 2^m,
// and the fold of 8 variables reads 256 streams per lane -- r3 sweep at 2^24 / 2^20, ms per proof: m <= 4 0.338 / 0.123, 5 0.346 / 0.120,
// 6 0.330 / 0.121, 7 0.332 / 0.137 (7 + 2), 8 0.348 / 0.191)
static int multi_pass_rounds(size_t global_len) {
    const int left = (int)ilog2(global_len) - (int)ilog2(kTailLen), kmax = multi_kmax();
    const int passes = (left + kmax - 1) / kmax;
    return (left + passes - 1) / passes;
}
template <class F> int launch_seg_sums(const void *in, size_t len, int m, void *part, unsigned *bps_out, const MultiFin *fin = nullptr) {
    const size_t seglen = len >> m;
    const unsigned bps = multi_bps(seglen, m);
    seg_sums_kernel<F><<<bps << m, kBlock, 0, cur_stream()>>>(in, seglen, bps, part, fin ? *fin : MultiFin{});
    ZK_HIP(hipGetLastError());
    *bps_out = bps;
    return ZK_OK;
}
// One pass that folds k variables (challenges at rp[0 .. k)) of `in` (n << k entries) into `out` (n entries).  A pass that leaves no
// segment sums and a SHORT output (the last one before the tail: 2^11 entries from 2^17) takes the split form, eight lanes per output
// (mle_kernels.cuh foldk_split_kernel: one lane per output was 34 us for 4 MB, two passes of k / 2 variables 19 us).
template <class F>
int fold_pass(const void *in, void *out, size_t n, int k, const void *const *rp, int m_next, void *part, unsigned *bps, const MultiFin *fin) {
    if (m_next == 0 && k >= 4 && n <= ((size_t)1 << 13)) return launch_foldk_split<F>(in, out, n, k, rp);
    return launch_foldk<F, true>(in, out, n, k, rp, m_next, part, bps, fin);
}

// ---- basic sumcheck prover: prover.rs:22-71 ----------------------------------------------------------
template <class F> int basic_prove(const zk_table *table, Transcript &tr, uint64_t *claimed_sum, uint64_t *round_polys, uint64_t *challenges) {
    const size_t esz = 4 * F::N, L64 = F::N / 2;
    size_t len = table->len;
    unsigned nvars = ilog2(len);
    double t0 = now_ms();
    ZK_TRY(absorb_table<F>(tr, table->dptr, len));                     // :38-39
    double t1 = now_ms();
    g_stats = zk_sumcheck_stats{nvars, (float)(t1 - t0), 0.f};
    // working buffers: len/2 and len/4 elements, plus reduction partials
    DevBuf bufA, bufB;
    ZK_TRY(bufA.alloc((len / 2) * esz));
    ZK_TRY(bufB.alloc((len / 4) * esz));
    void *part;
    ZK_TRY(scratch(esz * ((size_t)kMaxReduceBlocks * 2 + 2), &part));
    Fe<F> sums[2];
    if (len == 1) {                                                    // zero variables: sum = the entry, no rounds
        ZK_TRY(download_elems<F>(table->dptr, 1, sums));
        store_el<F>(claimed_sum, sums[0]);
        tr.append_be<F>(sums[0]);
        return ZK_OK;
    }
    // Proof slots: 0 = claimed sum; round k: 1+3k, 2+3k = the two half sums, 3+3k = the challenge.
    // Every transcript step from here on runs on the device (dev_transcript.cuh): no host round trip per round.
    DeviceRounds<F> dr;
    ZK_TRY(dr.init(tr, std::vector<Fe<F>>(), 1 + 3 * (size_t)nvars));
    if (dr.host_mode) {                                                // several rounds per pass over the table (basic_multi.cuh)
        const void *cur = table->dptr;
        void *dst = bufA.p, *other = bufB.p;
        size_t cl = len, round = 0;
        if (cl > kTailLen) {
            int m = multi_pass_rounds(cl);
            unsigned bps;
            MultiFin fin;
            // every pass's last workgroup runs the exchange on the segment sums the pass leaves: rounds round .. round + m - 1 (:50-58)
            ZK_TRY(dr.multi_fin(m, round, &fin));
            ZK_TRY((launch_seg_sums<F>(cur, cl, m, part, &bps, &fin)));    // :74-89, by 2^m segments
            for (;;) {
                const size_t n = cl >> m;
                const int mn = n > kTailLen ? multi_pass_rounds(n) : 0;
                const void *rp[kMultiMax];
                for (int i = 0; i < m; i++) rp[i] = dr.slot_ptr(3 + 3 * (round + (size_t)i));
                if (mn) ZK_TRY(dr.multi_fin(mn, round + (size_t)m, &fin));
                // :61-63 m times, fused with :74-89 and the exchange of the next rounds
                ZK_TRY((fold_pass<F>(cur, dst, n, m, rp, mn, part, &bps, mn ? &fin : nullptr)));
                cur = dst;
                void *nx = other;
                other = dst;
                dst = nx;
                cl = n;
                round += (size_t)m;
                if (!mn) break;
                m = mn;
            }
        }
        ZK_TRY(dr.launch_basic_tail(cur, dst, cl, round));
        ZK_TRY(dr.collect(tr));
        g_stats.ms_rounds = (float)(now_ms() - t1);
        store_el<F>(claimed_sum, dr.slot(0));
        for (unsigned rd = 0; rd < nvars; rd++) {
            store_el<F>(round_polys + (size_t)(2 * rd) * L64, dr.slot(1 + 3 * (size_t)rd));
            store_el<F>(round_polys + (size_t)(2 * rd + 1) * L64, dr.slot(2 + 3 * (size_t)rd));
            if (challenges) store_el<F>(challenges + (size_t)rd * L64, dr.slot(3 + 3 * (size_t)rd));
        }
        return ZK_OK;
    }
    {   // round-0 half sums (split_polynomial_and_sum_each :74-89); claimed sum = their sum (:28), absorbed first (:40-41)
        size_t seg = len / 2;
        int grid = reduce_grid_for(seg);
        segment_sums_kernel<F><<<grid, kBlock, 0, cur_stream()>>>(table->dptr, seg, 2, part);
        ZK_HIP(hipGetLastError());
        ZK_TRY(dr.launch_finish(part, (size_t)grid, 2, 0, 1, 0, 1, 3));               // :50-58 of round 0
    }
    const void *cur = table->dptr;
    void *dst = bufA.p, *other = bufB.p;
    size_t cl = len;
    unsigned round = 0;
    for (; cl > kTailLen; round++) {                                   // :46
        const void *rp = dr.slot_ptr(3 + 3 * (size_t)round);           // this round's challenge (:58), on the device
        size_t q = cl / 4;                                             // :61-63 fused with the next round's :50
        int grid = reduce_grid_for(q);
        fold_half_sums_kernel<F><<<grid, kBlock, 0, cur_stream()>>>(cur, dst, q, fe_zero<F>(), part, rp);
        ZK_HIP(hipGetLastError());
        ZK_TRY(dr.launch_finish(part, (size_t)grid, 2, 0, 0, 0, 1 + 3 * (size_t)(round + 1), 3 + 3 * (size_t)(round + 1)));
        cur = dst;
        void *nx = other;
        other = dst;
        dst = nx;
        cl /= 2;
    }
    {   // rounds on <= kTailLen entries: one launch; the half sums are the evaluations at 0 and 1 of a 1-factor product
        SumPolyTables tabs{};
        tabs.in[0] = cur;
        ZK_TRY(dr.launch_tail(tabs, dst, other, 1, 1, cl, 0, round, 1, 3, 3, ~(size_t)0));
    }
    ZK_TRY(dr.collect(tr));
    g_stats.ms_rounds = (float)(now_ms() - t1);
    store_el<F>(claimed_sum, dr.slot(0));
    for (unsigned round = 0; round < nvars; round++) {
        store_el<F>(round_polys + (size_t)(2 * round) * L64, dr.slot(1 + 3 * (size_t)round));
        store_el<F>(round_polys + (size_t)(2 * round + 1) * L64, dr.slot(2 + 3 * (size_t)round));
        if (challenges) store_el<F>(challenges + (size_t)round * L64, dr.slot(3 + 3 * (size_t)round));
    }
    return ZK_OK;
}

// ---- basic sumcheck verifier: verifier.rs:23-71 -------------------------------------------------------
template <class F> int basic_verify(const zk_table *table, const uint64_t *claimed_sum, const uint64_t *round_polys,
                                    size_t nrounds, int *ok) {
    const size_t L64 = F::N / 2;
    *ok = 0;
    if (nrounds != ilog2(table->len)) return ZK_OK;                    // :26-30
    Transcript tr;
    ZK_TRY(absorb_table<F>(tr, table->dptr, table->len));              // :34-35
    Fe<F> cur = load_el<F>(claimed_sum);
    tr.append_be<F>(cur);                                              // :36-37
    std::vector<uint64_t> chal(nrounds * L64 + L64);
    for (size_t i = 0; i < nrounds; i++) {                             // :47
        Fe<F> e0 = load_el<F>(round_polys + (2 * i) * L64), e1 = load_el<F>(round_polys + (2 * i + 1) * L64);
        // evaluating a 2-entry table at 0 / 1 returns its entries (:51-52)
        if (!fe_eq<F>(fe_add<F>(e0, e1), cur)) return ZK_OK;           // :53-56
        tr.append_be<F>(e0);                                           // :58-59
        tr.append_be<F>(e1);
        Fe<F> c = tr.random_challenge_as_field_element<F>();           // :61
        store_el<F>(chal.data() + i * L64, c);
        cur = fe_add<F>(e0, fe_mul<F>(c, fe_sub<F>(e1, e0)));          // :64
    }
    uint64_t fin[6];
    ZK_TRY(zk_mle_evaluate(table, chal.data(), nrounds, fin));         // :67 -- the GPU fold chain
    *ok = fe_eq<F>(load_el<F>(fin), cur) ? 1 : 0;                      // :70
    return ZK_OK;
}

// ---- SumPolynomial helpers ------------------------------------------------------------------------------
int check_sumpoly(const zk_table *const *tables, size_t nprod, size_t nfac) {
    if (!tables || nprod == 0 || nfac == 0) return ZK_E_ARG;
    if (nprod > (size_t)kMaxProducts || nfac > (size_t)kMaxFactors) return ZK_E_ARG;
    for (size_t k = 0; k < nprod * nfac; k++) {
        if (!tables[k]) return ZK_E_ARG;
        if (tables[k]->field != tables[0]->field) return ZK_E_ARG;
        if (!is_pow2(tables[k]->len)) return ZK_E_NOT_POW2;
        if (tables[k]->len != tables[0]->len) return ZK_E_NVARS;      // product_polynomial.rs:16-21, sum_polynomial.rs:17-23
    }
    return ZK_OK;
}

// the same checks with constant second factors allowed (null entries at odd positions, two-factor products)
int check_sumpoly_cf(const zk_table *const *tables, size_t nprod, size_t nfac, const uint64_t *const_factors) {
    if (!tables || nprod == 0 || nfac != 2 || nprod > (size_t)kMaxProducts || !tables[0]) return ZK_E_ARG;
    for (size_t k = 0; k < nprod * nfac; k++) {
        if (!tables[k]) {
            if ((k & 1) == 0 || !const_factors) return ZK_E_ARG;
            continue;
        }
        if (tables[k]->field != tables[0]->field) return ZK_E_ARG;
        if (!is_pow2(tables[k]->len)) return ZK_E_NOT_POW2;
        if (tables[k]->len != tables[0]->len) return ZK_E_NVARS;
    }
    return ZK_OK;
}

// skip1: the two-factor lazy kernel leaves out the products of the point 1 (the caller derives e(1) = claim - e(0)); others ignore it.
// fin (host-assisted transcript step): the launch's last workgroup runs the round's exchange, nothing is left in `part`.
template <class F> int launch_round_evals(const SumPolyTables &tabs, int nprod, int nfac, size_t half, void *part, int grid, int skip1 = 0,
                                          const RoundFin *fin = nullptr) {
    const RoundFin f = fin ? *fin : RoundFin{};
    if (nfac == 1) round_evals_kernel<F, 1><<<grid, kBlock, 0, cur_stream()>>>(tabs, nprod, half, part, f);
    else if (nfac == 2 && skip1 && LazyProducts<F>::value) round_evals_kernel<F, 2, true><<<grid, kBlock, 0, cur_stream()>>>(tabs, nprod, half, part, f);
    else if (nfac == 2) round_evals_kernel<F, 2><<<grid, kBlock, 0, cur_stream()>>>(tabs, nprod, half, part, f);
    else round_evals_kernel<F, 3><<<grid, kBlock, 0, cur_stream()>>>(tabs, nprod, half, part, f);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}
// the grid the fused round will be launched with (`grid` = reduce_grid_for(q) on entry), before the launch: the split kernel has its own
inline bool fold_round_takes_split(int nprod, int nfac, size_t q, bool may_split) {
    static const size_t split_max_q = [] { const char *e = getenv("ZK_SPLIT_ROUND_BITS"); int b = e ? atoi(e) : 0; return b >= 6 && b <= 24 ? (size_t)1 << b : kSplitRoundMaxQ; }();
    return may_split && nfac == 2 && nprod >= 2 && nprod <= 4 && q >= 64 && q <= split_max_q;
}
// `grid` = the number of partials written per evaluation point.  With `may_split` the caller lets short rounds of two-factor
// products take the split kernel, which changes `grid` (sumcheck_kernels.cuh).
template <class F> int launch_fold_round_evals(const SumPolyTables &tabs, int nprod, int nfac, size_t q, const Fe<F> &r, void *part, int &grid,
                                               const void *rp = nullptr, int skip1 = 0, bool may_split = false, const RoundFin *fin = nullptr,
                                               const uint32_t *rexp = nullptr) {
    const RoundFin f = fin ? *fin : RoundFin{};
    if (fold_round_takes_split(nprod, nfac, q, may_split)) {
        grid = (int)(q / 64);
        fold_round_evals_split_kernel<F><<<grid, 64 * 2 * nprod, 0, cur_stream()>>>(tabs, q, r, part, rp, skip1, f);
        ZK_HIP(hipGetLastError());
        return ZK_OK;
    }
    UniArg ua{};                                                      // a challenge the host knows travels as the uniform multiplier's rows (ufield.cuh UniMul)
    if constexpr (LazyProducts<F>::value) {
        if (nfac == 2 && !rp && !rexp) {
            UniMul<F> um;
            unimul_from<F>(um, r);
            static_assert(sizeof um.t == sizeof ua.t, "nine rows of nine limbs");
            memcpy(ua.t, um.t, sizeof ua.t);
            ua.valid = 1;
        }
    }
    if (skip1) {
        if (nfac == 1) fold_round_evals_kernel<F, 1, true><<<grid, kBlock, 0, cur_stream()>>>(tabs, nprod, q, r, part, rp, f, rexp, ua);
        else if (nfac == 2) fold_round_evals_kernel<F, 2, true><<<grid, kBlock, 0, cur_stream()>>>(tabs, nprod, q, r, part, rp, f, rexp, ua);
        else fold_round_evals_kernel<F, 3, true><<<grid, kBlock, 0, cur_stream()>>>(tabs, nprod, q, r, part, rp, f, rexp, ua);
    } else if (nfac == 1) fold_round_evals_kernel<F, 1><<<grid, kBlock, 0, cur_stream()>>>(tabs, nprod, q, r, part, rp, f, rexp, ua);
    else if (nfac == 2) fold_round_evals_kernel<F, 2><<<grid, kBlock, 0, cur_stream()>>>(tabs, nprod, q, r, part, rp, f, rexp, ua);
    else fold_round_evals_kernel<F, 3><<<grid, kBlock, 0, cur_stream()>>>(tabs, nprod, q, r, part, rp, f, rexp, ua);
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}

// The public evaluation entry points return e(0), e(1), e(2) (include/zkmle.h); the two-factor kernels produce e(0), e(1), e(inf) = c2:
// e(2) = c0 + 2 c1 + 4 c2 = 2 (e(1) + e(inf)) - e(0).
template <class F> void evals_node2_from_infinity(uint64_t *evals) {
    const size_t L64 = F::N / 2;
    const Fe<F> e0 = load_el<F>(evals), e1 = load_el<F>(evals + L64), ei = load_el<F>(evals + 2 * L64);
    const Fe<F> s = fe_add<F>(e1, ei);
    store_el<F>(evals + 2 * L64, fe_sub<F>(fe_add<F>(s, s), e0));
}
template <class F> int round_evals(const zk_table *const *tables, size_t nprod, size_t nfac, uint64_t *out) {
    const size_t esz = 4 * F::N;
    size_t len = tables[0]->len;
    if (len < 2) return ZK_E_NOT_POW2;
    SumPolyTables tabs{};
    for (size_t k = 0; k < nprod * nfac; k++) tabs.in[k] = tables[k]->dptr;
    size_t half = len / 2;
    int grid = reduce_grid_for(half);
    size_t npts = nfac + 1;
    void *part;
    ZK_TRY(scratch(esz * ((size_t)kMaxReduceBlocks * (kMaxFactors + 1) + kMaxFactors + 1), &part));
    void *res = (char *)part + esz * (size_t)grid * npts;
    ZK_TRY((launch_round_evals<F>(tabs, (int)nprod, (int)nfac, half, part, grid)));
    if (!out) return ZK_OK;                                            // enqueue only (measurements: the producer kernel alone)
    finish_sums_kernel<F><<<1, kBlock, 0, cur_stream()>>>(part, (size_t)grid, (int)npts, res);
    ZK_HIP(hipGetLastError());
    ZK_HIP(zk::memcpy_on_stream(out, res, esz * npts, hipMemcpyDeviceToHost));
    if (nfac == 2) evals_node2_from_infinity<F>(out);
    return ZK_OK;
}

// The interpolation nodes of the GKR sumcheck are always 0..d (sumcheck_gkr_protocol.rs:46-48), so the Lagrange basis
// (d + 1 host inversions, ~0.15 ms) is built once per (field, d) and kept: Montgomery form, then canonical integers
// (FinishArgs::basis).
// d = 2 (every GKR round: products of two factors): the kernels evaluate at 0, 1 and INFINITY -- e(inf) = the X^2 coefficient
// (sumcheck_kernels.cuh header) -- and the matrix below turns those into the coefficients: c0 = e0, c1 = e1 - e0 - e(inf), c2 = e(inf).
// The round message is the coefficient vector of the same polynomial the reference interpolates from e(0), e(1), e(2).
template <class F> const std::vector<Fe<F>> &sumcheck_basis(size_t npts) {
    static std::mutex mu;
    static std::map<size_t, std::vector<Fe<F>>> cache;
    std::lock_guard<std::mutex> lock(mu);
    auto it = cache.find(npts);
    if (it != cache.end()) return it->second;
    std::vector<Fe<F>> xs(npts);
    for (size_t i = 0; i < npts; i++) xs[i] = fe_from_u64<F>(i);
    std::vector<std::vector<Fe<F>>> basis = lagrange_basis_matrix<F>(xs);
    if (npts == 3) {                                           // basis[i][d] = the weight of evaluation i in coefficient d
        const Fe<F> one = fe_one<F>(), zero = fe_zero<F>(), minus = fe_sub<F>(zero, one);
        basis = {{one, minus, zero}, {zero, one, zero}, {zero, minus, one}};
    }
    std::vector<Fe<F>> flat(2 * npts * npts);
    for (size_t i = 0; i < npts; i++)
        for (size_t d = 0; d < npts; d++) {
            flat[i * npts + d] = basis[i][d];
            flat[npts * npts + i * npts + d] = fe_to_canonical<F>(basis[i][d]);
        }
    return cache.emplace(npts, std::move(flat)).first->second;
}

// ---- GKR sumcheck prover: sumcheck_gkr_protocol.rs:24-67 --------------------------------------------------
// Enqueue every round of a sum-of-products sumcheck on proof slots that already exist (no upload, no synchronisation): round k's
// coefficients at s0 + per k, its challenge at s0 + per k + npts, the ntab final values at s0 + per nvars (per = npts + 1).
// Constant second factors: tables[p * 2 + 1] == null, value from const_host (nprod elements) or const_dev[p] (device memory).
// with_claim: proof[claim_slot] (already on the device) is absorbed big-endian in front of round 0's message (:35).
// own_claim: the claimed sum of this sumcheck was computed by the prover itself (a layer of a GKR proof: the previous phase's / link's
// running claim), so round 0 may take e(1) = claim - e(0) like every later round does; a claim that comes from outside (zk_sumcheck_gkr_prove)
// may be wrong, and the reference then sends the true e(1).
template <class F> int gkr_rounds_enqueue(DeviceRounds<F> &dr, size_t s0, const zk_table *const *tables, size_t nprod, size_t nfac,
                                          const uint64_t *const_host, const void *const *const_dev, int with_claim, size_t claim_slot,
                                          bool own_claim = false) {
    const size_t esz = 4 * F::N, L64 = F::N / 2;
    const size_t ntab = nprod * nfac, npts = nfac + 1;                 // degree() = polynomials.len() (:114, sum_polynomial.rs:88)
    size_t len = tables[0]->len;
    unsigned nvars = ilog2(len);                                       // :29
    DevBuf bufA, bufB;                                                 // pooled: stream-ordered, safe to release once enqueued
    ZK_TRY(bufA.alloc(ntab * (len / 2) * esz));
    ZK_TRY(bufB.alloc(ntab * (len / 4 ? len / 4 : 1) * esz));
    void *part;
    ZK_TRY(scratch(esz * ((size_t)kMaxReduceBlocks * (kMaxFactors + 1) + kMaxFactors + 1), &part));
    // Interpolation, absorb and challenge of every round run in the finish kernel (dev_transcript.cuh).
    const size_t per = npts + 1, fin_slot = s0 + per * nvars;
    SumPolyTables tabs{};
    for (size_t k = 0; k < ntab; k++) {
        tabs.in[k] = tables[k] ? tables[k]->dptr : nullptr;
        if (!tables[k]) {
            if (const_dev && const_dev[k / nfac]) tabs.cptr[k / nfac] = const_dev[k / nfac];
            else memcpy(tabs.cval[k / nfac], const_host + (k / nfac) * L64, esz);
        }
    }
    // The fused round of the two-factor lazy kernel folds by the challenge as a UNIFORM multiplier (ufield.cuh UniMul); the exchange that receives
    // the challenge of a round whose fold is such a launch leaves it in that form (host-assisted step; otherwise the kernel's waves work it out).
    const size_t tail_from0 = (nfac == 2 && nprod >= 2) ? kTailLen / 2 : kTailLen;
    // ZK_GRID_TWO_ROUNDS=0: the short grid-wide rounds stay one round per launch (measurement / fallback switch)
    static const bool grid_two = [] { const char *e = getenv("ZK_GRID_TWO_ROUNDS"); return !(e && e[0] == '0'); }();
    static const size_t grid_two_max_q = [] { const char *e = getenv("ZK_GRID_TWO_BITS"); int b = e ? atoi(e) : 17; return (size_t)1 << (b < 6 ? 6 : (b > 24 ? 24 : b)); }();   // r4 sweep, 4 x 2^22 (15 / 16 / 17 / 18 / 19): 0.710 / 0.724 / 0.692 / 0.731 / 0.741 ms
    const bool two_regime_ok = grid_two && dr.host_mode && nfac == 2 && nprod == 2;
    auto takes_uniform = [&](size_t cl_folded) {                     // cl_folded: the length of the tables that challenge folds
        return dr.host_mode && nfac == 2 && LazyProducts<F>::value && cl_folded > tail_from0 && !fold_round_takes_split((int)nprod, (int)nfac, cl_folded / 4, true) &&
               !(two_regime_ok && cl_folded >= 512 && cl_folded / 4 <= grid_two_max_q && (ilog2(cl_folded) & 1u));
    };
    const bool tail_takes_all = len <= tail_from0;                   // every round, the first one's evaluations included, in the one-workgroup tail
    if (!tail_takes_all) {   // round 0 evaluations
        size_t half = len / 2;
        int grid = reduce_grid_for(half);
        // host-assisted step only: the host keeps the running claim (the device variant reads the previous round's slots, which round 0 has not)
        const int skip0 = (own_claim && dr.host_mode && nfac == 2 && LazyProducts<F>::value && half >= ((size_t)1 << 14)) ? 1 : 0;
        if (dr.host_mode) {                                            // the exchange runs in the kernel's last workgroup
            RoundFin fin;
            ZK_TRY(dr.round_fin(grid, (int)npts, 1, with_claim, claim_slot, s0, s0 + npts, skip0, &fin, takes_uniform(len)));
            ZK_TRY((launch_round_evals<F>(tabs, (int)nprod, (int)nfac, half, part, grid, skip0, &fin)));
        } else {
            ZK_TRY((launch_round_evals<F>(tabs, (int)nprod, (int)nfac, half, part, grid, skip0)));
            ZK_TRY(dr.launch_finish(part, (size_t)grid, (int)npts, 1, with_claim, claim_slot, s0, s0 + npts, skip0, per));
        }
    }
    char *dst = (char *)bufA.p, *other = (char *)bufB.p;
    size_t cl = len;
    unsigned round = 0;
    // The one-workgroup tail takes over at 2048 entries, at 1024 for two or more two-factor products: its round on 2048 entries is a
    // chain of 14 products per lane (28 us measured, tools ZK_TAIL_TRACE), the grid-wide round + finish launch pair takes 19.
    const size_t tail_from = (nfac == 2 && nprod >= 2) ? kTailLen / 2 : kTailLen;
    for (; cl > tail_from; round++) {                                  // :37
        const void *rp = dr.slot_ptr(s0 + per * round + npts);         // :55, on the device
        size_t ol = cl / 2, q = cl / 4;
        // (entered at an ODD log2 of the table length: the launches then end on 2^10 entries, which is where the one-workgroup tail is cheapest to enter)
        if (two_regime_ok && cl >= 512 && q <= grid_two_max_q && (ilog2(cl) & 1u)) {
            // From here down to the tail: TWO rounds per launch and exchange (sumcheck_kernels.cuh split2_round_kernel).  The first launch folds by the one
            // pending challenge; every later one by the two its predecessor's exchange brought; the tail starts by folding with the first of its two.
            auto launch2 = [&](size_t in_len, size_t first_new_round, const void *rp0, const void *rp1) -> int {
                const size_t out_len = rp1 ? in_len / 4 : in_len / 2, qq = out_len / 4;
                for (size_t k = 0; k < ntab; k++) tabs.out[k] = tables[k] ? dst + k * out_len * esz : nullptr;
                RoundFin fin;
                ZK_TRY(dr.round2_fin((int)(qq / 64), s0 + per * first_new_round, s0 + per * first_new_round + npts, per, &fin));
                split2_round_kernel<F><<<(unsigned)(qq / 64), 1024, 0, cur_stream()>>>(tabs, qq, rp0, rp1, fin);
                ZK_HIP(hipGetLastError());
                for (size_t k = 0; k < ntab; k++) tabs.in[k] = tabs.out[k];
                char *nx = other;
                other = dst;
                dst = nx;
                return ZK_OK;
            };
            ZK_TRY(launch2(cl, round + 1, rp, nullptr));               // challenges of rounds round + 1, round + 2 come back
            cl = ol;
            size_t R = round + 1;                                      // the first of the two pending challenges
            while (cl > kTailLen) {                                    // 16 qq entries in, qq >= 128
                ZK_TRY(launch2(cl, R + 2, dr.slot_ptr(s0 + per * R + npts), dr.slot_ptr(s0 + per * (R + 1) + npts)));
                cl /= 4;
                R += 2;
            }
            return dr.launch_tail(tabs, dst, other, (int)nprod, (int)nfac, cl, 1, R, s0, s0 + npts, per, fin_slot, 0, 0, 0, 1);
        }
        for (size_t k = 0; k < ntab; k++) tabs.out[k] = tables[k] ? dst + k * ol * esz : nullptr;
        int grid = reduce_grid_for(q);                                 // :57 fused with next round's :41
        // large rounds skip the products of the point 1: e(1) = p_round(r_round) - e(0), derived in the finish step
        // (dev_transcript.cuh kDerive1).  Below ~2^14 pair indices the helper wave's two products take longer than the
        // reduction they hide behind, so small rounds evaluate the point 1 directly (measured r1: 4 x 2^22 1.28 -> 1.23 ms).
        const int skip1 = q >= ((size_t)1 << 14) ? 1 : 0;
        if (dr.host_mode) {
            RoundFin fin;
            const int g = fold_round_takes_split((int)nprod, (int)nfac, q, true) ? (int)(q / 64) : grid;
            ZK_TRY(dr.round_fin(g, (int)npts, 1, 0, 0, s0 + per * (round + 1), s0 + per * (round + 1) + npts, skip1, &fin, takes_uniform(ol)));
            ZK_TRY((launch_fold_round_evals<F>(tabs, (int)nprod, (int)nfac, q, fe_zero<F>(), part, grid, rp, skip1, true, &fin,
                                               takes_uniform(cl) ? dr.uexp() : nullptr)));
        } else {
            ZK_TRY((launch_fold_round_evals<F>(tabs, (int)nprod, (int)nfac, q, fe_zero<F>(), part, grid, rp, skip1, true)));
            ZK_TRY(dr.launch_finish(part, (size_t)grid, (int)npts, 1, 0, 0, s0 + per * (round + 1), s0 + per * (round + 1) + npts, skip1, per));
        }
        for (size_t k = 0; k < ntab; k++) tabs.in[k] = tabs.out[k];
        char *nx = other;
        other = dst;
        dst = nx;
        cl = ol;
    }
    // rounds on <= kTailLen entries, the last fold and the final values: one launch
    return dr.launch_tail(tabs, dst, other, (int)nprod, (int)nfac, cl, 1, round, s0, s0 + npts, per, fin_slot, tail_takes_all ? 1 : 0, with_claim, claim_slot);
}

// `const_factors` (nprod elements, may be null): where tables[p * 2 + 1] is null the second factor of product p is that constant
// (sumcheck_kernels.cuh const_factor; two-factor products only)
template <class F> int gkr_sumcheck_rounds(const zk_table *const *tables, size_t nprod, size_t nfac, Transcript &tr,
                                           uint64_t *round_coeffs, uint64_t *challenges, uint64_t *final_values,
                                           const uint64_t *const_factors = nullptr) {
    const size_t esz = 4 * F::N, L64 = F::N / 2;
    const size_t ntab = nprod * nfac, npts = nfac + 1;
    size_t len = tables[0]->len;
    unsigned nvars = ilog2(len);
    if (nvars == 0) {
        if (final_values)
            for (size_t k = 0; k < ntab; k++) {
                if (tables[k]) ZK_HIP(zk::memcpy_on_stream(final_values + k * L64, tables[k]->dptr, esz, hipMemcpyDeviceToHost));
                else memcpy(final_values + k * L64, const_factors + (k / nfac) * L64, esz);
            }
        return ZK_OK;
    }
    const std::vector<Fe<F>> &basis_flat = sumcheck_basis<F>(npts);    // :46-50, nodes 0..d never change
    // Proof slots: round k: (npts+1)k .. +npts-1 = coefficients (:49-52), +npts = challenge (:55); then ntab final values.
    const size_t per = npts + 1, fin_slot = per * nvars;
    const double t1 = now_ms();
    DeviceRounds<F> dr;
    ZK_TRY(dr.init(tr, basis_flat, fin_slot + ntab));
    ZK_TRY((gkr_rounds_enqueue<F>(dr, 0, tables, nprod, nfac, const_factors, nullptr, 0, 0)));
    ZK_TRY(dr.collect(tr));
    g_stats = zk_sumcheck_stats{nvars, 0.f, (float)(now_ms() - t1)};
    for (unsigned round = 0; round < nvars; round++) {
        for (size_t i = 0; i < npts; i++) store_el<F>(round_coeffs + ((size_t)round * npts + i) * L64, dr.slot(per * round + i));
        store_el<F>(challenges + (size_t)round * L64, dr.slot(per * round + npts));   // :59
    }
    if (final_values)
        for (size_t k = 0; k < ntab; k++) store_el<F>(final_values + k * L64, dr.slot(fin_slot + k));
    return ZK_OK;
}

template <class F> int gkr_sumcheck_prove(const zk_table *const *tables, size_t nprod, size_t nfac, const uint64_t *claimed_sum,
                                          Transcript &tr, uint64_t *round_coeffs, uint64_t *challenges) {
    tr.append_be<F>(load_el<F>(claimed_sum));                          // :35
    return gkr_sumcheck_rounds<F>(tables, nprod, nfac, tr, round_coeffs, challenges, nullptr);
}

template <class F> int gkr_sumcheck_verify(const uint64_t *claimed_sum, const uint64_t *round_coeffs, size_t nrounds, size_t ncoef,
                                           Transcript &tr, uint64_t *challenges, uint64_t *last, int *ok) {
    const size_t L64 = F::N / 2;
    Fe<F> cur = load_el<F>(claimed_sum);
    tr.append_be<F>(cur);                                              // :73
    *ok = 1;
    for (size_t r = 0; r < nrounds; r++) {                             // :78
        std::vector<Fe<F>> c(ncoef);
        for (size_t i = 0; i < ncoef; i++) c[i] = load_el<F>(round_coeffs + (r * ncoef + i) * L64);
        Fe<F> e0 = uni_evaluate<F>(c, fe_zero<F>()), e1 = uni_evaluate<F>(c, fe_one<F>());   // :81-82
        if (!fe_eq<F>(fe_add<F>(e0, e1), cur)) { *ok = 0; break; }     // :84-90
        for (size_t i = 0; i < ncoef; i++) tr.append_le<F>(c[i]);      // :92
        Fe<F> ch = tr.random_challenge_as_field_element<F>();          // :94
        cur = uni_evaluate<F>(c, ch);                                  // :96
        store_el<F>(challenges + r * L64, ch);
    }
    store_el<F>(last, cur);
    return ZK_OK;
}

// ---- device-resident rounds as a handle: sharded (one process per GPU) sumcheck provers -------------------------------------
// The caller (zkmle_amd/sharded.py, or a Rust shim) owns the collective; this object owns the sponge, the proof slots and
// the round counter.  Sequence per round: evals / fold_evals (local tables -> limb sums in device memory) ->
// all-reduce(SUM, int64) over RCCL by the caller -> absorb (transcript step on the summed limbs, challenge stays on the device).
struct RoundsBase {
    virtual ~RoundsBase() {}
    virtual size_t limbs_len() const = 0;
    virtual int evals(const zk_table *const *tables, uint64_t *limbs) = 0;
    virtual int fold_evals(const zk_table *const *in, zk_table *const *out, uint64_t *limbs) = 0;
    virtual int absorb(const uint64_t *limbs) = 0;
    virtual int tail(const zk_table *const *tables) = 0;
    virtual unsigned multi_max() const = 0;
    virtual int multi_evals(const zk_table *table, unsigned m, uint64_t *limbs) = 0;
    virtual int multi_absorb(const uint64_t *limbs, unsigned m) = 0;
    virtual int multi_fold_evals(const zk_table *in, zk_table *out, unsigned k, unsigned m_next, uint64_t *limbs) = 0;
    virtual int multi_tail(const zk_table *table) = 0;
    virtual int collect(zk_transcript *t, uint64_t *claimed_sum, uint64_t *messages, uint64_t *challenges, uint64_t *final_values) = 0;
};

template <class F> struct RoundsImpl : RoundsBase {
    int mode;                                            // 0 basic (1 table, half sums, big-endian), 1 GKR sumcheck
    size_t nprod, nfac, ntab, npts, nrounds;
    size_t msg_base, chal_base, per, fin_slot;
    size_t round = 0;                                    // rounds absorbed so far
    bool tail_done = false;
    bool skipped1 = false;                               // the limbs on their way lack the point 1 (derived in absorb)
    DeviceRounds<F> dr;
    DevBuf tailbuf;

    int init(int mode_, size_t nprod_, size_t nfac_, size_t nrounds_, Transcript &tr) {
        mode = mode_; nprod = nprod_; nfac = nfac_; ntab = nprod * nfac; npts = nfac + 1; nrounds = nrounds_;
        if (mode == 0) { msg_base = 1; chal_base = 3; per = 3; fin_slot = 1 + 3 * nrounds; }
        else { msg_base = 0; chal_base = npts; per = npts + 1; fin_slot = per * nrounds; }
        static const std::vector<Fe<F>> none;
        return dr.init(tr, mode == 1 ? sumcheck_basis<F>(npts) : none, fin_slot + ntab);
    }
    size_t limbs_len() const override { return (mode == 0 ? (size_t)1 << kMultiMax : npts) * (F::N + 1); }
    int check_tables(const zk_table *const *t, size_t minlen) const {
        ZK_TRY(check_sumpoly(t, nprod, nfac));
        if (t[0]->field != F::ID || t[0]->len < minlen) return ZK_E_ARG;
        return ZK_OK;
    }
    int evals(const zk_table *const *tables, uint64_t *limbs) override {
        ZK_TRY(check_tables(tables, 2));
        if (!limbs || round >= nrounds) return ZK_E_ARG;
        SumPolyTables tabs{};
        for (size_t k = 0; k < ntab; k++) tabs.in[k] = tables[k]->dptr;
        size_t half = tables[0]->len / 2;
        int grid = reduce_grid_for(half);
        void *part;
        ZK_TRY(scratch(4 * F::N * ((size_t)kMaxReduceBlocks * (kMaxFactors + 1) + kMaxFactors + 1), &part));
        const RoundFin fin = dr.round_fin_limbs(grid, (int)npts, 0, limbs);          // the kernel's last workgroup writes the limbs
        ZK_TRY((launch_round_evals<F>(tabs, (int)nprod, (int)nfac, half, part, grid, 0, &fin)));
        skipped1 = false;
        return ZK_OK;
    }
    int fold_evals(const zk_table *const *in, zk_table *const *out, uint64_t *limbs) override {
        ZK_TRY(check_tables(in, 2));
        if (!out || round == 0 || round > nrounds) return ZK_E_ARG;      // needs the challenge of an absorbed round
        size_t len = in[0]->len;
        for (size_t k = 0; k < ntab; k++)
            if (!out[k] || out[k]->field != F::ID || out[k]->len < len / 2 || out[k]->dptr == in[k]->dptr) return ZK_E_ARG;
        SumPolyTables tabs{};
        for (size_t k = 0; k < ntab; k++) { tabs.in[k] = in[k]->dptr; tabs.out[k] = out[k]->dptr; }
        const void *rp = dr.slot_ptr(chal_base + per * (round - 1));
        if (len >= 4) {
            if (!limbs) return ZK_E_ARG;
            size_t q = len / 4;
            int grid = reduce_grid_for(q);
            void *part;
            ZK_TRY(scratch(4 * F::N * ((size_t)kMaxReduceBlocks * (kMaxFactors + 1) + kMaxFactors + 1), &part));
            const int skip1 = q >= ((size_t)1 << 14) ? 1 : 0;
            const int g = fold_round_takes_split((int)nprod, (int)nfac, q, true) ? (int)(q / 64) : grid;
            const RoundFin fin = dr.round_fin_limbs(g, (int)npts, skip1, limbs);
            ZK_TRY((launch_fold_round_evals<F>(tabs, (int)nprod, (int)nfac, q, fe_zero<F>(), part, grid, rp, skip1, true, &fin)));
            skipped1 = skip1 != 0;                                       // e(1) = claim - e(0), after the all-reduce
        } else {                                                         // 2 entries -> 1: nothing left to evaluate
            for (size_t k = 0; k < ntab; k++) {
                fold_kernel<F><<<1, 64, 0, cur_stream()>>>(tabs.in[k], tabs.out[k], 1, 0, fe_zero<F>(), rp);
                ZK_HIP(hipGetLastError());
            }
        }
        for (size_t k = 0; k < ntab; k++) out[k]->len = len / 2;
        return ZK_OK;
    }
    int absorb(const uint64_t *limbs) override {
        if (!limbs || round >= nrounds) return ZK_E_ARG;
        LimbsFinishArgs a{};
        a.limbs = limbs; a.ctx = dr.ctx((int)npts, mode); a.with_claim = (mode == 0 && round == 0) ? 1 : 0;
        a.flags = skipped1 ? kDerive1 : 0;
        a.claim_slot = 0; a.msg_slot = msg_base + per * round; a.chal_slot = chal_base + per * round;
        a.prev_msg_slot = a.msg_slot - per; a.prev_chal_slot = a.chal_slot - per;      // used with kDerive1 only (round >= 1)
        if (dr.host_mode) {                                  // the step runs on this rank's host (every rank's host does the same)
            dr.push_req(typename DeviceRounds<F>::Req{DeviceRounds<F>::kRound, mode, (int)npts, a.with_claim, skipped1 ? 1 : 0, 0, a.claim_slot,
                                                       a.msg_slot, a.chal_slot, 0, {0, 0, 0, 0, 0, 0, 0}});
            a.seq = dr.nreq();
            a.flags = 0;
        }
        limbs_finish_kernel<F><<<1, (skipped1 && !dr.host_mode) ? 128 : 64, 0, cur_stream()>>>(a);
        ZK_HIP(hipGetLastError());
        round++;
        return ZK_OK;
    }
    // every remaining round on tables every rank holds in full (<= kTailLen entries); round `round - 1` was absorbed
    int tail(const zk_table *const *tables) override {
        ZK_TRY(check_tables(tables, 2));
        size_t len = tables[0]->len;
        if (len > kTailLen || round == 0 || round - 1 + ilog2(len) != nrounds) return ZK_E_ARG;
        ZK_TRY(tailbuf.alloc(ntab * (len / 2 + len / 4 + 1) * 4 * F::N));
        SumPolyTables tabs{};
        for (size_t k = 0; k < ntab; k++) tabs.in[k] = tables[k]->dptr;
        char *b0 = (char *)tailbuf.p, *b1 = b0 + ntab * (len / 2) * 4 * F::N;
        ZK_TRY(dr.launch_tail(tabs, b0, b1, (int)nprod, (int)nfac, len, mode, round - 1, msg_base, chal_base, per, fin_slot));
        round = nrounds;
        tail_done = true;
        return ZK_OK;
    }
    // ---- basic sumcheck, several rounds per pass (basic_multi.cuh): host-assisted transcript step only ----
    unsigned multi_max() const override { return (mode == 0 && dr.host_mode) ? (unsigned)multi_kmax() : 0u; }
    int multi_table(const zk_table *t, size_t minlen) const {
        if (multi_max() == 0 || !t || t->field != F::ID || !is_pow2(t->len) || t->len < minlen) return ZK_E_ARG;
        return ZK_OK;
    }
    // limbs == nullptr (here and in multi_fold_evals): ONE rank -- there is nothing to all-reduce, the pass's last workgroup runs the
    // exchange itself and the rounds count as absorbed (no multi_absorb)
    int multi_evals(const zk_table *table, unsigned m, uint64_t *limbs) override {
        ZK_TRY(multi_table(table, 2));
        if (m < 1 || m > multi_max() || ((size_t)1 << m) > table->len || round + m > nrounds) return ZK_E_ARG;
        void *part;
        ZK_TRY(scratch(4 * F::N * ((size_t)kMaxReduceBlocks * (kMaxFactors + 1) + kMaxFactors + 1), &part));
        unsigned bps;
        MultiFin fin = dr.multi_fin_limbs((int)m, limbs);
        if (!limbs) ZK_TRY(dr.multi_fin((int)m, round, &fin));
        ZK_TRY((launch_seg_sums<F>(table->dptr, table->len, (int)m, part, &bps, &fin)));
        if (!limbs) round += m;
        return ZK_OK;
    }
    int multi_absorb(const uint64_t *limbs, unsigned m) override {
        if (!limbs || m < 1 || m > multi_max() || round + m > nrounds) return ZK_E_ARG;
        ZK_TRY(dr.launch_multi(limbs, (int)m, round));
        round += m;
        return ZK_OK;
    }
    int multi_fold_evals(const zk_table *in, zk_table *out, unsigned k, unsigned m_next, uint64_t *limbs) override {
        ZK_TRY(multi_table(in, 2));
        if (k < 1 || k > multi_max() || k > round || ((size_t)1 << k) > in->len || m_next > multi_max()) return ZK_E_ARG;
        const size_t n = in->len >> k;
        if (!out || out->field != F::ID || out->len < n || out->dptr == in->dptr) return ZK_E_ARG;
        if (m_next && (((size_t)1 << m_next) > n || round + m_next > nrounds)) return ZK_E_ARG;
        void *part;
        ZK_TRY(scratch(4 * F::N * ((size_t)kMaxReduceBlocks * (kMaxFactors + 1) + kMaxFactors + 1), &part));
        const void *rp[kMultiMax];
        for (unsigned i = 0; i < k; i++) rp[i] = dr.slot_ptr(chal_base + per * (round - k + i));
        unsigned bps;
        MultiFin fin = dr.multi_fin_limbs((int)m_next, limbs);
        if (m_next && !limbs) ZK_TRY(dr.multi_fin((int)m_next, round, &fin));
        ZK_TRY((fold_pass<F>(in->dptr, out->dptr, n, (int)k, rp, (int)m_next, part, &bps, m_next ? &fin : nullptr)));
        out->len = n;
        if (m_next && !limbs) round += m_next;
        return ZK_OK;
    }
    // every remaining round on a table every rank holds in full (<= kTailLen entries); none of its rounds has started
    int multi_tail(const zk_table *table) override {
        ZK_TRY(multi_table(table, 2));
        const size_t len = table->len;
        if (len > kTailLen || round + ilog2(len) != nrounds) return ZK_E_ARG;
        ZK_TRY(tailbuf.alloc((len / 2) * 4 * F::N));
        ZK_TRY(dr.launch_basic_tail(table->dptr, tailbuf.p, len, round));
        round = nrounds;
        return ZK_OK;
    }
    int collect(zk_transcript *t, uint64_t *claimed_sum, uint64_t *messages, uint64_t *challenges, uint64_t *final_values) override {
        const size_t L64 = F::N / 2;
        if (!t || round != nrounds) return ZK_E_ARG;
        ZK_TRY(dr.collect(t->t));
        if (dr.host_mode && dr.htr != &t->t) t->t = *dr.htr;                 // the steps ran on the transcript given at creation
        if (claimed_sum && mode == 0) store_el<F>(claimed_sum, dr.slot(0));
        for (size_t r = 0; r < nrounds; r++) {
            if (messages)
                for (size_t i = 0; i < npts; i++) store_el<F>(messages + (r * npts + i) * L64, dr.slot(msg_base + per * r + i));
            if (challenges) store_el<F>(challenges + r * L64, dr.slot(chal_base + per * r));
        }
        if (final_values) {
            if (!tail_done) return ZK_E_ARG;
            for (size_t k = 0; k < ntab; k++) store_el<F>(final_values + k * L64, dr.slot(fin_slot + k));
        }
        return ZK_OK;
    }
};

}  // namespace

// ---- proof slots shared by a whole multi-sumcheck proof (the sparse GKR prover): one upload, one stream of kernels, one download -------
namespace zk {
template <class F> struct ProofSlotsImpl : ProofSlotsBase {
    DeviceRounds<F> dr;
    void *slot_ptr(size_t s) const override { return dr.slot_ptr(s); }
    int upload_slot(size_t s, const uint64_t *el) override {
        ZK_HIP(hipMemcpyAsync(dr.slot_ptr(s), el, 4 * F::N, hipMemcpyHostToDevice, cur_stream()));
        ZK_HIP(hipStreamSynchronize(cur_stream()));                    // `el` is the caller's stack
        return ZK_OK;
    }
    int rounds(size_t s0, const zk_table *const *tables, size_t nprod, size_t nfac, const uint64_t *const_host, const void *const *const_dev,
               int with_claim, size_t claim_slot) override {
        return gkr_rounds_enqueue<F>(dr, s0, tables, nprod, nfac, const_host, const_dev, with_claim, claim_slot, true);
    }
    void set_claim(const uint64_t *el) override {                      // the claimed sum of the first sumcheck (later ones: running claim / link)
        Fe<F> e;
        memcpy(e.l, el, 4 * F::N);
        dr.running_claim = e;
    }
    int link(size_t wb_src, size_t wc_src, size_t wb_slot, size_t wc_slot, size_t alpha_slot, size_t beta_slot, size_t claim_slot) override {
        if (dr.host_mode) return dr.launch_link_host(wb_src, wc_src, wb_slot, wc_slot, alpha_slot, beta_slot, claim_slot);
        LinkArgs a{dr.sponge(), dr.proof(), wb_src, wc_src, wb_slot, wc_slot, alpha_slot, beta_slot, claim_slot};
        gkr_link_kernel<F><<<1, 64, 0, cur_stream()>>>(a);
        ZK_HIP(hipGetLastError());
        return ZK_OK;
    }
    int collect(Transcript &tr, uint64_t *host_slots) override {
        ZK_TRY(dr.collect(tr));
        for (size_t k = 0; k < dr.nslots; k++) store_el<F>(host_slots + k * (F::N / 2), dr.slot(k));
        return ZK_OK;
    }
};
int proof_slots_new(int field, Transcript &tr, size_t npts, size_t nslots, ProofSlotsBase **out) {
    ZK_DISPATCH_FIELD(field, {
        auto *p = new ProofSlotsImpl<F>();
        int rc = p->dr.init(tr, sumcheck_basis<F>(npts), nslots);
        if (rc != ZK_OK) { delete p; return rc; }
        *out = p;
    });
    return ZK_OK;
}
}  // namespace zk

namespace zk {
int transcript_absorb_table(Transcript &t, const zk_table *table) {
    if (!table) return ZK_E_ARG;
    ZK_TRY(require_device());
    ZK_DISPATCH_FIELD(table->field, return absorb_table<F>(t, table->dptr, table->len));
    return ZK_OK;
}
}  // namespace zk

extern "C" {

int zk_transcript_new(zk_transcript **out) {
    if (!out) return ZK_E_ARG;
    *out = new zk_transcript();
    return ZK_OK;
}
int zk_transcript_free(zk_transcript *t) { delete t; return ZK_OK; }
int zk_transcript_append(zk_transcript *t, const uint8_t *data, size_t n) {
    if (!t || (!data && n)) return ZK_E_ARG;
    t->t.append(data, n);
    return ZK_OK;
}
int zk_transcript_sample(zk_transcript *t, uint8_t out32[32]) {
    if (!t || !out32) return ZK_E_ARG;
    t->t.sample_random_challenge(out32);
    return ZK_OK;
}
int zk_transcript_challenge(zk_transcript *t, int field, uint64_t *out) {
    if (!t || !out) return ZK_E_ARG;
    ZK_DISPATCH_FIELD(field, store_el<F>(out, t->t.random_challenge_as_field_element<F>()));
    return ZK_OK;
}
int zk_transcript_export_state(const zk_transcript *t, uint64_t lanes25[25], uint32_t *fill) {
    if (!t || !lanes25 || !fill) return ZK_E_ARG;
    const_cast<zk_transcript *>(t)->t.sponge().export_state(lanes25, fill);
    return ZK_OK;
}
int zk_transcript_import_state(zk_transcript *t, const uint64_t lanes25[25], uint32_t fill) {
    if (!t || !lanes25 || fill >= 136) return ZK_E_ARG;
    t->t.sponge().import_state(lanes25, fill);
    return ZK_OK;
}
int zk_keccak256(const uint8_t *data, size_t n, uint8_t out32[32]) {
    if ((!data && n) || !out32) return ZK_E_ARG;
    Keccak256 h;
    h.update(data, n);
    h.finalize_copy(out32);
    return ZK_OK;
}

int zk_uni_evaluate(int field, const uint64_t *coeffs, size_t n, const uint64_t *x, uint64_t *out) {
    if ((!coeffs && n) || !x || !out) return ZK_E_ARG;
    ZK_DISPATCH_FIELD(field, {
        std::vector<Fe<F>> c(n);
        for (size_t i = 0; i < n; i++) c[i] = load_el<F>(coeffs + i * (F::N / 2));
        store_el<F>(out, uni_evaluate<F>(c, load_el<F>(x)));
    });
    return ZK_OK;
}
int zk_uni_lagrange_interpolate(int field, const uint64_t *xs, const uint64_t *ys, size_t n, uint64_t *out) {
    if (!xs || !ys || !out || n == 0) return ZK_E_ARG;
    ZK_DISPATCH_FIELD(field, {
        std::vector<Fe<F>> x(n), y(n);
        for (size_t i = 0; i < n; i++) { x[i] = load_el<F>(xs + i * (F::N / 2)); y[i] = load_el<F>(ys + i * (F::N / 2)); }
        std::vector<Fe<F>> c = lagrange_interpolate<F>(x, y);
        for (size_t i = 0; i < n; i++) store_el<F>(out + i * (F::N / 2), c[i]);
    });
    return ZK_OK;
}

int zk_debug_stall_service_once(int milliseconds) {
    // fault injection: process-global, so it only arms in a process that asked for it
    { const char *e = getenv("ZK_ENABLE_FAULT_INJECTION"); if (!(e && e[0] == '1')) { set_last_error("zk_debug_stall_service_once: set ZK_ENABLE_FAULT_INJECTION=1 in the environment to use the test hook"); return ZK_E_ARG; } }
    if (milliseconds < 0) return ZK_E_ARG;
    g_stall_service_ms.store(milliseconds);
    return ZK_OK;
}

int zk_sumcheck_last_stats(zk_sumcheck_stats *out) {
    if (!out) return ZK_E_ARG;
    *out = g_stats;
    return ZK_OK;
}

int zk_sumcheck_basic_prove(const zk_table *table, uint64_t *claimed_sum, uint64_t *round_polys, uint64_t *challenges) {
    if (!table || !claimed_sum || !round_polys) return ZK_E_ARG;
    if (!is_pow2(table->len)) return ZK_E_NOT_POW2;      // Prover::init -> MultilinearPolynomial::new (prover.rs:23)
    ZK_TRY(require_device());
    Transcript tr;                                         // Prover::init: Transcript::new() (prover.rs:24)
    ZK_DISPATCH_FIELD(table->field, return basic_prove<F>(table, tr, claimed_sum, round_polys, challenges));
    return ZK_OK;
}
int zk_sumcheck_basic_prove_on(const zk_table *table, zk_transcript *transcript, uint64_t *claimed_sum, uint64_t *round_polys, uint64_t *challenges) {
    if (!table || !transcript || !claimed_sum || !round_polys) return ZK_E_ARG;
    if (!is_pow2(table->len)) return ZK_E_NOT_POW2;
    ZK_TRY(require_device());
    ZK_DISPATCH_FIELD(table->field, return basic_prove<F>(table, transcript->t, claimed_sum, round_polys, challenges));
    return ZK_OK;
}
int zk_sumcheck_basic_verify(const zk_table *table, const uint64_t *claimed_sum, const uint64_t *round_polys, size_t nrounds, int *ok) {
    if (!table || !claimed_sum || (!round_polys && nrounds) || !ok) return ZK_E_ARG;
    if (!is_pow2(table->len)) return ZK_E_NOT_POW2;
    ZK_TRY(require_device());
    ZK_DISPATCH_FIELD(table->field, return basic_verify<F>(table, claimed_sum, round_polys, nrounds, ok));
    return ZK_OK;
}

int zk_sumpoly_round_evals(const zk_table *const *tables, size_t nprod, size_t nfac, uint64_t *out) {
    ZK_TRY(check_sumpoly(tables, nprod, nfac));
    // the reference reduces with add_polynomials_element_wise (:134), which asserts > 1 (sum_polynomial.rs:58-61)
    if (nprod < 2 || nfac < 2) return ZK_E_NEED_TWO;
    ZK_TRY(require_device());
    ZK_DISPATCH_FIELD(tables[0]->field, return round_evals<F>(tables, nprod, nfac, out));
    return ZK_OK;
}
int zk_sumpoly_fold_round_evals(const zk_table *const *in, zk_table *const *out, size_t nprod, size_t nfac, const uint64_t *value,
                                uint64_t *out_evals) {
    ZK_TRY(check_sumpoly(in, nprod, nfac));
    if (!out || !value) return ZK_E_ARG;
    if (nprod < 2 || nfac < 2) return ZK_E_NEED_TWO;
    size_t len = in[0]->len;
    if (len < 4) return ZK_E_ARG;
    for (size_t k = 0; k < nprod * nfac; k++)
        if (!out[k] || out[k]->field != in[0]->field || out[k]->len < len / 2 || out[k]->dptr == in[k]->dptr) return ZK_E_ARG;
    ZK_TRY(require_device());
    ZK_DISPATCH_FIELD(in[0]->field, {
        const size_t esz = 4 * F::N, npts = nfac + 1;
        SumPolyTables tabs{};
        for (size_t k = 0; k < nprod * nfac; k++) { tabs.in[k] = in[k]->dptr; tabs.out[k] = out[k]->dptr; }
        size_t q = len / 4;
        int grid = reduce_grid_for(q);
        void *part;
        ZK_TRY(scratch(esz * ((size_t)kMaxReduceBlocks * (kMaxFactors + 1) + kMaxFactors + 1), &part));
        void *res = (char *)part + esz * (size_t)grid * npts;
        // out_evals == NULL: the tables are folded and the kernel is only enqueued (no reduction, no read-back); skip_point1 as in the provers' large rounds
        ZK_TRY((launch_fold_round_evals<F>(tabs, (int)nprod, (int)nfac, q, load_el<F>(value), part, grid, nullptr, out_evals ? 0 : 1)));
        if (out_evals) {
            finish_sums_kernel<F><<<1, kBlock, 0, cur_stream()>>>(part, (size_t)grid, (int)npts, res);
            ZK_HIP(hipGetLastError());
            ZK_HIP(zk::memcpy_on_stream(out_evals, res, esz * npts, hipMemcpyDeviceToHost));
            if (nfac == 2) evals_node2_from_infinity<F>(out_evals);
        }
    });
    for (size_t k = 0; k < nprod * nfac; k++) out[k]->len = len / 2;
    return ZK_OK;
}
int zk_sumpoly_reduce(const zk_table *const *tables, size_t nprod, size_t nfac, zk_table *out) {
    ZK_TRY(check_sumpoly(tables, nprod, nfac));
    if (!out || out->field != tables[0]->field || out->len < tables[0]->len) return ZK_E_ARG;
    if (nprod < 2 || nfac < 2) return ZK_E_NEED_TWO;     // sum_polynomial.rs:58-61 / product_polynomial.rs:59-62
    ZK_TRY(require_device());
    SumPolyTables tabs{};
    for (size_t k = 0; k < nprod * nfac; k++) tabs.in[k] = tables[k]->dptr;
    size_t len = tables[0]->len;
    ZK_DISPATCH_FIELD(tables[0]->field, (sumpoly_reduce_kernel<F><<<grid_for(len), kBlock, 0, cur_stream()>>>(tabs, (int)nprod, (int)nfac, len, out->dptr)));
    ZK_HIP(hipGetLastError());
    out->len = len;
    return ZK_OK;
}
int zk_prodpoly_reduce(const zk_table *const *tables, size_t nfac, zk_table *out) {
    ZK_TRY(check_sumpoly(tables, 1, nfac));
    if (!out || out->field != tables[0]->field || out->len < tables[0]->len) return ZK_E_ARG;
    if (nfac < 2) return ZK_E_NEED_TWO;                  // product_polynomial.rs:59-62
    ZK_TRY(require_device());
    SumPolyTables tabs{};
    for (size_t k = 0; k < nfac; k++) tabs.in[k] = tables[k]->dptr;
    size_t len = tables[0]->len;
    ZK_DISPATCH_FIELD(tables[0]->field, (sumpoly_reduce_kernel<F><<<grid_for(len), kBlock, 0, cur_stream()>>>(tabs, 1, (int)nfac, len, out->dptr)));
    ZK_HIP(hipGetLastError());
    out->len = len;
    return ZK_OK;
}
int zk_sumpoly_evaluate(const zk_table *const *tables, size_t nprod, size_t nfac, const uint64_t *values, size_t nvalues, uint64_t *out) {
    ZK_TRY(check_sumpoly(tables, nprod, nfac));
    if (!out) return ZK_E_ARG;
    ZK_DISPATCH_FIELD(tables[0]->field, {
        Fe<F> result = fe_zero<F>();                      // sum_polynomial.rs:31
        for (size_t p = 0; p < nprod; p++) {
            Fe<F> prod = fe_one<F>();                     // product_polynomial.rs:27
            for (size_t f = 0; f < nfac; f++) {
                uint64_t e[6];
                ZK_TRY(zk_mle_evaluate(tables[p * nfac + f], values, nvalues, e));
                prod = fe_mul<F>(prod, load_el<F>(e));
            }
            result = fe_add<F>(result, prod);
        }
        store_el<F>(out, result);
    });
    return ZK_OK;
}
int zk_sumcheck_gkr_prove(const zk_table *const *tables, size_t nprod, size_t nfac, const uint64_t *claimed_sum,
                          zk_transcript *t, uint64_t *round_coeffs, uint64_t *challenges) {
    ZK_TRY(check_sumpoly(tables, nprod, nfac));
    if (!claimed_sum || !t || !round_coeffs || !challenges) return ZK_E_ARG;
    if ((nprod < 2 || nfac < 2) && tables[0]->len > 1) return ZK_E_NEED_TWO;   // first round's generate_round_univariate panics
    ZK_TRY(require_device());
    ZK_DISPATCH_FIELD(tables[0]->field, return gkr_sumcheck_prove<F>(tables, nprod, nfac, claimed_sum, t->t, round_coeffs, challenges));
    return ZK_OK;
}
int zk_sumcheck_gkr_rounds(const zk_table *const *tables, size_t nprod, size_t nfac, zk_transcript *t, uint64_t *round_coeffs,
                           uint64_t *challenges, uint64_t *final_values) {
    ZK_TRY(check_sumpoly(tables, nprod, nfac));
    if (!t || !round_coeffs || !challenges) return ZK_E_ARG;
    if ((nprod < 2 || nfac < 2) && tables[0]->len > 1) return ZK_E_NEED_TWO;
    ZK_TRY(require_device());
    ZK_DISPATCH_FIELD(tables[0]->field, return gkr_sumcheck_rounds<F>(tables, nprod, nfac, t->t, round_coeffs, challenges, final_values));
    return ZK_OK;
}
int zk_sumcheck_gkr_rounds_cf(const zk_table *const *tables, size_t nprod, size_t nfac, const uint64_t *const_factors, zk_transcript *t,
                              uint64_t *round_coeffs, uint64_t *challenges, uint64_t *final_values) {
    ZK_TRY(check_sumpoly_cf(tables, nprod, nfac, const_factors));
    if (!t || !round_coeffs || !challenges) return ZK_E_ARG;
    if (nprod < 2 && tables[0]->len > 1) return ZK_E_NEED_TWO;
    ZK_TRY(require_device());
    ZK_DISPATCH_FIELD(tables[0]->field, return gkr_sumcheck_rounds<F>(tables, nprod, nfac, t->t, round_coeffs, challenges, final_values, const_factors));
    return ZK_OK;
}
int zk_sumcheck_gkr_verify(int field, const uint64_t *claimed_sum, const uint64_t *round_coeffs, size_t nrounds, size_t ncoef,
                           zk_transcript *t, uint64_t *challenges, uint64_t *last_claimed_sum, int *ok) {
    if (!claimed_sum || (!round_coeffs && nrounds) || !t || !challenges || !last_claimed_sum || !ok) return ZK_E_ARG;
    ZK_DISPATCH_FIELD(field, return gkr_sumcheck_verify<F>(claimed_sum, round_coeffs, nrounds, ncoef, t->t, challenges, last_claimed_sum, ok));
    return ZK_OK;
}

struct zk_rounds {
    RoundsBase *impl;
};

int zk_rounds_new(int field, int mode, size_t nprod, size_t nfac, size_t nrounds, zk_transcript *t, zk_rounds **out) {
    if (!t || !out || nrounds == 0 || (mode != 0 && mode != 1)) return ZK_E_ARG;
    if (nprod == 0 || nfac == 0 || nprod > (size_t)kMaxProducts || nfac > (size_t)kMaxFactors) return ZK_E_ARG;
    if (mode == 0 && (nprod != 1 || nfac != 1)) return ZK_E_ARG;
    if (mode == 1 && (nprod < 2 || nfac < 2)) return ZK_E_NEED_TWO;
    ZK_TRY(require_device());
    ZK_DISPATCH_FIELD(field, {
        auto *r = new RoundsImpl<F>();
        int rc = r->init(mode, nprod, nfac, nrounds, t->t);
        if (rc != ZK_OK) { delete r; return rc; }
        *out = new zk_rounds{r};
    });
    return ZK_OK;
}
int zk_rounds_free(zk_rounds *r) {
    if (r) { delete r->impl; delete r; }
    return ZK_OK;
}
size_t zk_rounds_limbs_len(const zk_rounds *r) { return r ? r->impl->limbs_len() : 0; }
int zk_rounds_evals(zk_rounds *r, const zk_table *const *tables, uint64_t *limbs_dev) {
    if (!r || !tables) return ZK_E_ARG;
    return r->impl->evals(tables, limbs_dev);
}
int zk_rounds_fold_evals(zk_rounds *r, const zk_table *const *in, zk_table *const *out, uint64_t *limbs_dev) {
    if (!r || !in) return ZK_E_ARG;
    return r->impl->fold_evals(in, out, limbs_dev);
}
int zk_rounds_absorb(zk_rounds *r, const uint64_t *limbs_dev) {
    if (!r) return ZK_E_ARG;
    return r->impl->absorb(limbs_dev);
}
int zk_rounds_tail(zk_rounds *r, const zk_table *const *tables) {
    if (!r || !tables) return ZK_E_ARG;
    return r->impl->tail(tables);
}
unsigned zk_rounds_multi_max(const zk_rounds *r) { return r ? r->impl->multi_max() : 0u; }
int zk_rounds_multi_evals(zk_rounds *r, const zk_table *table, unsigned m, uint64_t *limbs_dev) {
    if (!r) return ZK_E_ARG;
    return r->impl->multi_evals(table, m, limbs_dev);
}
int zk_rounds_multi_absorb(zk_rounds *r, const uint64_t *limbs_dev, unsigned m) {
    if (!r) return ZK_E_ARG;
    return r->impl->multi_absorb(limbs_dev, m);
}
int zk_rounds_multi_fold_evals(zk_rounds *r, const zk_table *in, zk_table *out, unsigned k, unsigned m_next, uint64_t *limbs_dev) {
    if (!r) return ZK_E_ARG;
    return r->impl->multi_fold_evals(in, out, k, m_next, limbs_dev);
}
int zk_rounds_multi_tail(zk_rounds *r, const zk_table *table) {
    if (!r) return ZK_E_ARG;
    return r->impl->multi_tail(table);
}
int zk_rounds_collect(zk_rounds *r, zk_transcript *t, uint64_t *claimed_sum, uint64_t *messages, uint64_t *challenges, uint64_t *final_values) {
    if (!r) return ZK_E_ARG;
    return r->impl->collect(t, claimed_sum, messages, challenges, final_values);
}


}  // extern "C"
