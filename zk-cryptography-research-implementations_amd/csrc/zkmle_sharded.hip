// zkmle_sharded.hip -- C ABI: the multi-GPU provers (one process per GPU) and their communicator.
//
// the tables shard by the LOW index bits (rank g of G holds the entries i == g mod G as a contiguous local table),
// so every round that folds variable 0 (prover.rs:62, sumcheck_gkr_protocol.rs:57) is local and the only per-round exchange is
// the sum over the ranks of 2 (basic) or d + 1 (GKR) evaluations.  Here that sum is ONE ncclAllReduce(ncclInt64, ncclSum) of
// (d + 1) x 9 words in device memory, enqueued on the prover's stream between the producer kernel and the transcript kernel
// (dev_transcript.cuh limbs_finish_kernel): RCCL over xGMI, no host round trip inside a sumcheck.  Once the GLOBAL table has
// <= kTailLen entries the ranks all-gather what is left and every rank finishes the proof replicated in one launch
// (sumcheck_tail_kernel), so the latency-bound small rounds cost no collective at all.
// RCCL is opened with dlopen at first use (librccl.so.1): single-GPU users never load it.
#include <dlfcn.h>
#include <string.h>
#include <time.h>

#include <condition_variable>
#include <atomic>
#include <mutex>
#include <string>
#include <vector>

#include <rccl/rccl.h>

#include "context.h"
#include "mle_kernels.cuh"
#include "transcript.h"

using namespace zk;

namespace {

// ---- RCCL, bound at run time ------------------------------------------------------------------------------
struct RcclApi {
    void *lib = nullptr;
    std::string err;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
};

RcclApi &rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        // a copy the process already holds (e.g. the one PyTorch ships) wins: one RCCL per process
        const char *names[] = {getenv("ZK_RCCL_LIB"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for (const char *n : names) {
            if (!n || !*n) continue;
            api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
            if (api.lib) break;
        }
        for (const char *n : names) {
            if (api.lib) break;
            if (!n || !*n) continue;
            api.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        }
        if (!api.lib) { api.err = std::string("librccl.so.1 not found: ") + (dlerror() ? dlerror() : ""); return; }
#define ZK_SYM(field, name)                                                              \
    api.field = (decltype(api.field))dlsym(api.lib, name);                               \
    if (!api.field && api.err.empty()) api.err = std::string("RCCL symbol missing: ") + name;
        ZK_SYM(GetUniqueId, "ncclGetUniqueId")
        ZK_SYM(CommInitRank, "ncclCommInitRank")
        ZK_SYM(CommDestroy, "ncclCommDestroy")
        ZK_SYM(GetErrorString, "ncclGetErrorString")
        ZK_SYM(AllReduce, "ncclAllReduce")
        ZK_SYM(AllGather, "ncclAllGather")
        ZK_SYM(Broadcast, "ncclBroadcast")
        ZK_SYM(Send, "ncclSend")
        ZK_SYM(Recv, "ncclRecv")
        ZK_SYM(GroupStart, "ncclGroupStart")
        ZK_SYM(GroupEnd, "ncclGroupEnd")
#undef ZK_SYM
    });
    return api;
}

int rccl_ready() {
    RcclApi &a = rccl();
    if (!a.lib || !a.err.empty()) { set_last_error(a.err); return ZK_E_COMM; }
    return ZK_OK;
}

#define ZK_NCCL(call)                                                                                        \
    do {                                                                                                     \
        ncclResult_t r__ = (call);                                                                           \
        if (r__ != ncclSuccess) {                                                                            \
            set_last_error(std::string(#call) + ": " + rccl().GetErrorString(r__));                          \
            return ZK_E_COMM;                                                                                \
        }                                                                                                    \
    } while (0)

#define ZK_CB(call, what)                                                                                    \
    do {                                                                                                     \
        if ((call) != 0) { set_last_error(std::string("exchange callback failed: ") + what); return ZK_E_COMM; } \
    } while (0)

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { pool_free(p); }
    int alloc(size_t bytes) { return pool_alloc(bytes, &p); }
};

struct TableSet {     // pooled temporaries, freed with the set
    std::vector<zk_table *> t;
    ~TableSet() { for (zk_table *x : t) zk_table_free(x); }
    int alloc(int field, size_t len, size_t count) {
        for (size_t k = 0; k < count; k++) {
            zk_table *x = nullptr;
            ZK_TRY(table_alloc_pooled(field, len, &x));
            t.push_back(x);
        }
        return ZK_OK;
    }
};

}  // namespace

// The ranks as threads of one process: a generation barrier and one published pointer per rank.  Every exchange is
// publish -> barrier -> read the other ranks' host buffers -> barrier (nobody reuses a buffer another rank still reads).
struct zk_comm_local_group {
    int nranks;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    bool aborted = false;
    std::atomic<int> ends{0};                              // communicators made from this group and not yet freed
    std::vector<const void *> ptr;
    explicit zk_comm_local_group(int n) : nranks(n), ptr((size_t)n, nullptr) {}
    int barrier() {
        std::unique_lock<std::mutex> lk(mu);
        if (aborted) return 1;
        const uint64_t gen = generation;
        if (++arrived == nranks) {
            arrived = 0;
            generation++;
            cv.notify_all();
            return 0;
        }
        cv.wait(lk, [&] { return generation != gen || aborted; });
        return generation != gen ? 0 : 1;
    }
    void abort() {
        std::lock_guard<std::mutex> lk(mu);
        aborted = true;
        cv.notify_all();
    }
};

namespace {
struct LocalEnd { zk_comm_local_group *g; int rank; };
int local_all_reduce(void *ctx, int64_t *buf, size_t count) {
    LocalEnd *e = (LocalEnd *)ctx;
    e->g->ptr[(size_t)e->rank] = buf;
    if (e->g->barrier()) return 1;
    std::vector<int64_t> acc(count, 0);
    for (int r = 0; r < e->g->nranks; r++) {
        const int64_t *src = (const int64_t *)e->g->ptr[(size_t)r];
        for (size_t i = 0; i < count; i++) acc[i] += src[i];
    }
    if (e->g->barrier()) return 1;
    memcpy(buf, acc.data(), count * 8);
    return 0;
}
int local_all_gather(void *ctx, const void *send, void *recv, size_t bytes) {
    LocalEnd *e = (LocalEnd *)ctx;
    e->g->ptr[(size_t)e->rank] = send;
    if (e->g->barrier()) return 1;
    for (int r = 0; r < e->g->nranks; r++) memcpy((char *)recv + (size_t)r * bytes, e->g->ptr[(size_t)r], bytes);
    return e->g->barrier();
}
int local_gather(void *ctx, const void *send, void *recv, size_t bytes, int root) {
    LocalEnd *e = (LocalEnd *)ctx;
    e->g->ptr[(size_t)e->rank] = send;
    if (e->g->barrier()) return 1;
    if (e->rank == root)
        for (int r = 0; r < e->g->nranks; r++) memcpy((char *)recv + (size_t)r * bytes, e->g->ptr[(size_t)r], bytes);
    return e->g->barrier();
}
int local_broadcast(void *ctx, void *buf, size_t bytes, int root) {
    LocalEnd *e = (LocalEnd *)ctx;
    if (e->rank == root) e->g->ptr[(size_t)root] = buf;
    if (e->g->barrier()) return 1;
    if (e->rank != root) memcpy(buf, e->g->ptr[(size_t)root], bytes);
    return e->g->barrier();
}
}  // namespace

struct zk_comm {
    int kind;                 // 0 = RCCL, 1 = host callbacks (kind 1 with `local` set: the ranks are threads of this process)
    int nranks, rank;
    ncclComm_t nccl;
    zk_comm_host_ops ops;
    uint64_t bytes_rx, ncoll;
    std::vector<uint8_t> hs, hr;     // host staging of the callback kind
    LocalEnd *local = nullptr;
    bool in_step = false;            // set once a sharded proof has been through its last exchange (sharded_rounds), cleared by the entry points

    int all_reduce_i64(void *dev, size_t count) {
        ncoll++;
        bytes_rx += nranks > 1 ? count * 8 : 0;
        if (kind == 0) {
            ZK_NCCL(rccl().AllReduce(dev, dev, count, ncclInt64, ncclSum, nccl, cur_stream()));
            return ZK_OK;
        }
        if (nranks == 1) return ZK_OK;
        hs.resize(count * 8);
        ZK_HIP(memcpy_on_stream(hs.data(), dev, count * 8, hipMemcpyDeviceToHost));
        ZK_CB(ops.all_reduce_sum_i64(ops.ctx, (int64_t *)hs.data(), count), "all_reduce_sum_i64");
        ZK_HIP(memcpy_on_stream(dev, hs.data(), count * 8, hipMemcpyHostToDevice));
        return ZK_OK;
    }
    int all_gather(const void *dev_send, void *dev_recv, size_t bytes) {
        ncoll++;
        bytes_rx += (uint64_t)(nranks - 1) * bytes;
        if (kind == 0) {
            ZK_NCCL(rccl().AllGather(dev_send, dev_recv, bytes, ncclUint8, nccl, cur_stream()));
            return ZK_OK;
        }
        if (nranks == 1) {
            ZK_HIP(hipMemcpyAsync(dev_recv, dev_send, bytes, hipMemcpyDeviceToDevice, cur_stream()));
            return ZK_OK;
        }
        hs.resize(bytes);
        hr.resize(bytes * (size_t)nranks);
        ZK_HIP(memcpy_on_stream(hs.data(), dev_send, bytes, hipMemcpyDeviceToHost));
        ZK_CB(ops.all_gather(ops.ctx, hs.data(), hr.data(), bytes), "all_gather");
        ZK_HIP(memcpy_on_stream(dev_recv, hr.data(), hr.size(), hipMemcpyHostToDevice));
        return ZK_OK;
    }
    // dev_recv (root only): nranks x bytes in rank order
    int gather(const void *dev_send, void *dev_recv, size_t bytes, int root) {
        ncoll++;
        if (rank == root) bytes_rx += (uint64_t)(nranks - 1) * bytes;
        if (kind == 0) {
            if (rank == root) {
                ZK_HIP(hipMemcpyAsync((char *)dev_recv + (size_t)root * bytes, dev_send, bytes, hipMemcpyDeviceToDevice, cur_stream()));
                if (nranks == 1) return ZK_OK;
                ZK_NCCL(rccl().GroupStart());
                for (int p = 0; p < nranks; p++)
                    if (p != root) ZK_NCCL(rccl().Recv((char *)dev_recv + (size_t)p * bytes, bytes, ncclUint8, p, nccl, cur_stream()));
                ZK_NCCL(rccl().GroupEnd());
            } else {
                ZK_NCCL(rccl().Send(dev_send, bytes, ncclUint8, root, nccl, cur_stream()));
            }
            return ZK_OK;
        }
        if (nranks == 1) {
            ZK_HIP(hipMemcpyAsync(dev_recv, dev_send, bytes, hipMemcpyDeviceToDevice, cur_stream()));
            return ZK_OK;
        }
        hs.resize(bytes);
        ZK_HIP(memcpy_on_stream(hs.data(), dev_send, bytes, hipMemcpyDeviceToHost));
        if (rank == root) hr.resize(bytes * (size_t)nranks);
        ZK_CB(ops.gather(ops.ctx, hs.data(), rank == root ? hr.data() : nullptr, bytes, root), "gather");
        if (rank == root) ZK_HIP(memcpy_on_stream(dev_recv, hr.data(), bytes * (size_t)nranks, hipMemcpyHostToDevice));
        return ZK_OK;
    }
    int broadcast(void *dev, size_t bytes, int root) {
        ncoll++;
        if (rank != root) bytes_rx += bytes;
        if (kind == 0) {
            ZK_NCCL(rccl().Broadcast(dev, dev, bytes, ncclUint8, root, nccl, cur_stream()));
            return ZK_OK;
        }
        if (nranks == 1) return ZK_OK;
        hs.resize(bytes);
        if (rank == root) ZK_HIP(memcpy_on_stream(hs.data(), dev, bytes, hipMemcpyDeviceToHost));
        ZK_CB(ops.broadcast(ops.ctx, hs.data(), bytes, root), "broadcast");
        if (rank != root) ZK_HIP(memcpy_on_stream(dev, hs.data(), bytes, hipMemcpyHostToDevice));
        return ZK_OK;
    }
};

namespace {

// rep[k][j * G + r] = recv[r][k][j]: the gathered local tables (send layout: ntab tables of L entries, one after the other)
// interleaved back into global index order (global i = j G + r).  One lane per 16-byte word.
__global__ void interleave_kernel(const uint4 *__restrict__ recv, uint4 *__restrict__ rep, size_t ntab, size_t L, size_t G, int words) {
    size_t total = ntab * L * G * (size_t)words, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        size_t w = i % words, e = i / words;           // e = output element index: (k, j, r)
        size_t r = e % G, j = (e / G) % L, k = e / (G * L);
        rep[i] = recv[((r * ntab + k) * L + j) * words + w];
    }
}

struct Sponge208 {            // what rank 0 broadcasts after the table absorb: 25 lanes + block fill
    uint64_t a[25];
    uint64_t fill;
};

// transcript.append(convert_to_bytes(table)) (prover.rs:38-39) for a table sharded over the ranks: every rank converts its
// entries to canonical big-endian bytes on its GPU and streams them to rank 0 chunk by chunk; rank 0 interleaves a chunk back
// into global order on its GPU, copies it to a pinned buffer and absorbs it while the next chunk is in flight; at the end the
// sponge state (208 bytes) is broadcast.  The sponge is sequential, so one rank has to see every byte; nobody else does.
template <class F> int sharded_absorb(zk_comm *c, Transcript &tr, const zk_table *shard) {
    const size_t esz = 4 * F::N, L = shard->len, G = (size_t)c->nranks;
    if (G == 1) return transcript_absorb_table(tr, shard);
    const bool root = c->rank == 0;
    size_t cl = ((size_t)1 << 18) / G;                    // local elements per chunk: 8 MiB of global bytes per chunk
    if (cl < 1) cl = 1;
    if (cl > L) cl = L;
    DevBuf snd[2], rcv[2], il[2];
    void *host[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
    for (int b = 0; b < 2; b++) {
        ZK_TRY(snd[b].alloc(cl * esz));
        if (root) {
            ZK_TRY(rcv[b].alloc(G * cl * esz));
            ZK_TRY(il[b].alloc(G * cl * esz));
        }
    }
    if (root) {
        ZK_TRY(pinned_pair(((size_t)1 << 18) * 4 * Fq381::N, host));
        ZK_HIP(hipEventCreateWithFlags(&ev[0], hipEventDisableTiming));
        ZK_HIP(hipEventCreateWithFlags(&ev[1], hipEventDisableTiming));
    }
    int rc = ZK_OK, pending = -1;
    size_t pending_bytes = 0;
    for (size_t off = 0, k = 0; off < L && rc == ZK_OK; off += cl, k++) {
        const int b = (int)(k & 1);
        const size_t n = L - off < cl ? L - off : cl;
        elementwise_kernel<F, OP_TO_CANONICAL_BE><<<grid_for(n), kBlock, 0, cur_stream()>>>((const char *)shard->dptr + off * esz, nullptr, snd[b].p, n, fe_zero<F>());
        if (hipGetLastError() != hipSuccess) { rc = ZK_E_HIP; break; }
        rc = c->gather(snd[b].p, root ? rcv[b].p : nullptr, n * esz, 0);
        if (rc != ZK_OK || !root) continue;
        interleave_kernel<<<grid_for(n * G * (esz / 16)), kBlock, 0, cur_stream()>>>((const uint4 *)rcv[b].p, (uint4 *)il[b].p, 1, n, G, (int)(esz / 16));
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipMemcpyAsync(host[b], il[b].p, n * G * esz, hipMemcpyDeviceToHost, cur_stream());
        if (e == hipSuccess) e = hipEventRecord(ev[b], cur_stream());
        if (e == hipSuccess && pending >= 0) {
            e = hipEventSynchronize(ev[pending]);
            if (e == hipSuccess) tr.append((const uint8_t *)host[pending], pending_bytes);
        }
        if (e != hipSuccess) { set_last_error(hipGetErrorString(e)); rc = ZK_E_HIP; break; }
        pending = b;
        pending_bytes = n * G * esz;
    }
    if (root && rc == ZK_OK && pending >= 0) {
        hipError_t e = hipEventSynchronize(ev[pending]);
        if (e == hipSuccess) tr.append((const uint8_t *)host[pending], pending_bytes);
        else { set_last_error(hipGetErrorString(e)); rc = ZK_E_HIP; }
    }
    if (root) {
        (void)hipEventDestroy(ev[0]);
        (void)hipEventDestroy(ev[1]);
    }
    ZK_TRY(rc);
    // the sponge after the absorb: rank 0 -> everyone
    DevBuf st;
    ZK_TRY(st.alloc(sizeof(Sponge208)));
    Sponge208 sp{};
    if (root) {
        uint32_t fill = 0;
        tr.sponge().export_state(sp.a, &fill);
        sp.fill = fill;
        ZK_HIP(memcpy_on_stream(st.p, &sp, sizeof sp, hipMemcpyHostToDevice));
    }
    ZK_TRY(c->broadcast(st.p, sizeof sp, 0));
    if (!root) {
        ZK_HIP(memcpy_on_stream(&sp, st.p, sizeof sp, hipMemcpyDeviceToHost));
        tr.sponge().import_state(sp.a, (uint32_t)sp.fill);
    } else {
        ZK_HIP(hipStreamSynchronize(cur_stream()));
    }
    return ZK_OK;
}

// The rounds of a sharded sumcheck (mode 0: basic, one table; mode 1: GKR sumcheck on nprod x nfac tables), driven through the
// zk_rounds handle (zkmle_sumcheck.hip).  `t` holds everything absorbed before the rounds and gets the sponge back.
int sharded_rounds(zk_comm *c, int field, int mode, const zk_table *const *tabs, size_t nprod, size_t nfac, zk_transcript *t,
                   uint64_t *claimed, uint64_t *messages, uint64_t *challenges, uint64_t *final_values) {
    const size_t ntab = nprod * nfac, G = (size_t)c->nranks, esz = (size_t)field_limbs64(field) * 8;
    size_t L = tabs[0]->len;
    const size_t nrounds = ilog2(L) + ilog2(G);
    if (G > 1024 || !is_pow2(G)) return ZK_E_ARG;
    // ZK_PROOF_TRACE=1 (measurement): host clock of this call's phases on stderr
    static const bool trace = [] { const char *e = getenv("ZK_PROOF_TRACE"); return e && e[0] == '1'; }();
    struct Clock {
        bool on; double t0, last; static double now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }
        void mark(const char *what) { if (!on) return; const double t = now(); fprintf(stderr, "[proof trace] %-28s +%7.1f us (%8.1f)\n", what, t - last, t - t0); last = t; }
    } clk{trace, Clock::now(), 0};
    clk.last = clk.t0;
    zk_rounds *r = nullptr;
    ZK_TRY(zk_rounds_new(field, mode, nprod, nfac, nrounds, t, &r));
    clk.mark("rounds handle + init");
    // The ranks agree on the outcome: a rank whose host side stalled (its kernels gave up and ended; it kept its place in every
    // collective, so nobody hangs) knows that its proof failed -- the others hold sums it contributed garbage to and cannot tell.  One
    // more word is summed over the ranks at the very end; any failure fails the proof everywhere.
    auto collect_agreed = [&]() -> int {
        const int rc = zk_rounds_collect(r, t, claimed, messages, challenges, final_values);
        if (G == 1) return rc;
        const std::string mine = rc != ZK_OK ? std::string(zk_last_error()) : std::string();
        DevBuf flag;
        ZK_TRY(flag.alloc(8));
        uint64_t f = rc != ZK_OK ? 1 : 0;
        ZK_HIP(memcpy_on_stream(flag.p, &f, 8, hipMemcpyHostToDevice));
        ZK_TRY(c->all_reduce_i64(flag.p, 1));
        c->in_step = true;                                 // every collective of the proof ran on this rank: an error from here on is an AGREED one
        ZK_HIP(memcpy_on_stream(&f, flag.p, 8, hipMemcpyDeviceToHost));
        if (rc != ZK_OK) { set_last_error(mine); return rc; }
        if (f != 0) { set_last_error("sharded proof: another rank's proof failed (its host-assisted transcript step stalled or a kernel gave up): this rank's proof is not valid"); return ZK_E_COMM; }
        return ZK_OK;
    };
    struct Guard { zk_rounds *r; ~Guard() { zk_rounds_free(r); } } guard{r};
    DevBuf limbs;
    ZK_TRY(limbs.alloc(zk_rounds_limbs_len(r) * 8));
    uint64_t *lp = (uint64_t *)limbs.p;
    const size_t kTail = 2048;                                           // kTailLen (dev_transcript.cuh): one-launch replicated tail
    TableSet ping, pong;
    std::vector<const zk_table *> cur(tabs, tabs + ntab);
    bool absorbed = false;                                               // the evaluations of `cur` are in the transcript
    const size_t round_words = (mode == 0 ? 2 : nfac + 1) * ((size_t)field_limbs64(field) * 2 + 1);   // one round's evaluations as limb words
    const unsigned kmax = mode == 0 ? zk_rounds_multi_max(r) : 0u;
    if (kmax > 0) {
        // basic sumcheck, several rounds per pass and per all-reduce (basic_multi.cuh): the top-bit segments of the global table are
        // the top-bit segments of every rank's shard, so the 2^m segment sums add up over the ranks like the two half sums do
        const size_t W = (size_t)field_limbs64(field) * 2 + 1;
        auto pass = [&](size_t global_len) {                               // the rounds left, spread evenly over the passes they need
            const unsigned left = (unsigned)(ilog2(global_len) - ilog2(kTail)), passes = (left + kmax - 1) / kmax;
            return (left + passes - 1) / passes;
        };
        const zk_table *one = tabs[0];
        if (L * G > kTail) {
            ZK_TRY(ping.alloc(field, L / 2, 1));
            ZK_TRY(pong.alloc(field, L / 4 ? L / 4 : 1, 1));
            TableSet *dst = &ping, *other = &pong;
            unsigned m = pass(L * G);
            // one rank: nothing to all-reduce -- every pass's last workgroup runs the exchange itself (limbs = null)
            uint64_t *xl = G > 1 ? lp : nullptr;
            ZK_TRY(zk_rounds_multi_evals(r, one, m, xl));
            for (;;) {
                if (xl) {
                    ZK_TRY(c->all_reduce_i64(lp, ((size_t)1 << m) * W));   // the only exchange of these m rounds, on the stream
                    ZK_TRY(zk_rounds_multi_absorb(r, lp, m));
                }                                                          // (one rank: no exchange ran, none is counted -- zk_comm_stats)
                const size_t n = L >> m;
                const unsigned mn = n * G > kTail ? pass(n * G) : 0u;
                ZK_TRY(zk_rounds_multi_fold_evals(r, one, dst->t[0], m, mn, xl));
                one = dst->t[0];
                TableSet *x = dst; dst = other; other = x;
                L = n;
                if (!mn) break;
                m = mn;
            }
        }
        if (G == 1) {
            ZK_TRY(zk_rounds_multi_tail(r, one));
            clk.mark("every round enqueued");
            const int rc = zk_rounds_collect(r, t, claimed, messages, challenges, final_values);
            clk.mark("collect");
            return rc;
        }
        DevBuf rcv, rep;                                                   // global index = j G + rank
        ZK_TRY(rcv.alloc(G * L * esz));
        ZK_TRY(rep.alloc(G * L * esz));
        ZK_TRY(c->all_gather(one->dptr, rcv.p, L * esz));
        interleave_kernel<<<grid_for(L * G * (esz / 16)), kBlock, 0, cur_stream()>>>((const uint4 *)rcv.p, (uint4 *)rep.p, 1, L, G, (int)(esz / 16));
        ZK_HIP(hipGetLastError());
        const zk_table view{field, L * G, rep.p, 0};
        ZK_TRY(zk_rounds_multi_tail(r, &view));
        return collect_agreed();
    }
    if (L * G > kTail) {
        ZK_TRY(ping.alloc(field, L / 2, ntab));
        ZK_TRY(pong.alloc(field, L / 4 ? L / 4 : 1, ntab));
        ZK_TRY(zk_rounds_evals(r, cur.data(), lp));
        ZK_TRY(c->all_reduce_i64(lp, round_words));           // the round's only exchange, on the stream
        ZK_TRY(zk_rounds_absorb(r, lp));
        absorbed = true;
        TableSet *dst = &ping, *other = &pong;
        while (L * G > kTail) {                                          // local rounds: fold + next evaluations, all-reduce, transcript
            ZK_TRY(zk_rounds_fold_evals(r, cur.data(), dst->t.data(), lp));
            ZK_TRY(c->all_reduce_i64(lp, round_words));
            ZK_TRY(zk_rounds_absorb(r, lp));
            for (size_t k = 0; k < ntab; k++) cur[k] = dst->t[k];
            TableSet *x = dst; dst = other; other = x;
            L /= 2;
        }
    }
    if (G == 1) {                                                        // the local table IS the global table
        if (!absorbed) {
            ZK_TRY(zk_rounds_evals(r, cur.data(), lp));
            ZK_TRY(c->all_reduce_i64(lp, round_words));       // a one-rank RCCL communicator still runs its collective
            ZK_TRY(zk_rounds_absorb(r, lp));
        }
        ZK_TRY(zk_rounds_tail(r, cur.data()));
        return collect_agreed();
    }
    // <= kTailLen entries left in the global table: gather them on every rank (global index = j G + rank) and finish replicated
    DevBuf snd, rcv;
    ZK_TRY(snd.alloc(ntab * L * esz));
    ZK_TRY(rcv.alloc(G * ntab * L * esz));
    for (size_t k = 0; k < ntab; k++)
        ZK_HIP(hipMemcpyAsync((char *)snd.p + k * L * esz, cur[k]->dptr, L * esz, hipMemcpyDeviceToDevice, cur_stream()));
    ZK_TRY(c->all_gather(snd.p, rcv.p, ntab * L * esz));
    DevBuf repbuf;                                                       // one block, ntab replicated tables inside
    ZK_TRY(repbuf.alloc(ntab * L * G * esz));
    interleave_kernel<<<grid_for(ntab * L * G * (esz / 16)), kBlock, 0, cur_stream()>>>((const uint4 *)rcv.p, (uint4 *)repbuf.p, ntab, L, G, (int)(esz / 16));
    ZK_HIP(hipGetLastError());
    std::vector<zk_table> views(ntab);
    std::vector<const zk_table *> vp(ntab);
    for (size_t k = 0; k < ntab; k++) {
        views[k] = zk_table{field, L * G, (char *)repbuf.p + k * L * G * esz, 0};
        vp[k] = &views[k];
    }
    if (!absorbed) {                                                     // small from the start: first evaluations, already global
        ZK_TRY(zk_rounds_evals(r, vp.data(), lp));
        ZK_TRY(zk_rounds_absorb(r, lp));
    }
    ZK_TRY(zk_rounds_tail(r, vp.data()));
    return collect_agreed();
}

}  // namespace

extern "C" {

int zk_comm_unique_id(uint8_t out128[128]) {
    if (!out128) return ZK_E_ARG;
    ZK_TRY(rccl_ready());
    ncclUniqueId id;
    static_assert(sizeof(id) == 128, "ncclUniqueId");
    ZK_NCCL(rccl().GetUniqueId(&id));
    memcpy(out128, &id, 128);
    return ZK_OK;
}
int zk_comm_init_rccl(const uint8_t id128[128], int nranks, int rank, zk_comm **out) {
    if (!id128 || !out || nranks < 1 || rank < 0 || rank >= nranks) return ZK_E_ARG;
    ZK_TRY(require_device());
    ZK_TRY(rccl_ready());
    ncclUniqueId id;
    memcpy(&id, id128, 128);
    ncclComm_t comm = nullptr;
    ZK_NCCL(rccl().CommInitRank(&comm, nranks, id, rank));
    *out = new zk_comm{0, nranks, rank, comm, zk_comm_host_ops{}, 0, 0, {}, {}};
    return ZK_OK;
}
int zk_comm_from_host_ops(const zk_comm_host_ops *ops, int nranks, int rank, zk_comm **out) {
    if (!ops || !out || nranks < 1 || rank < 0 || rank >= nranks) return ZK_E_ARG;
    if (nranks > 1 && (!ops->all_reduce_sum_i64 || !ops->all_gather || !ops->gather || !ops->broadcast)) return ZK_E_ARG;
    *out = new zk_comm{1, nranks, rank, nullptr, *ops, 0, 0, {}, {}};
    return ZK_OK;
}
// The ranks-as-threads barrier has no timeout: a rank that returns early from a sharded call (an allocation that failed, a bad argument)
// would leave its peers waiting in the next exchange for ever.  Every sharded entry point therefore aborts the group when it returns an
// error BEFORE its last exchange; the peers come back with ZK_E_COMM.  (RCCL and caller-supplied callbacks have their own failure paths.)
// (A failure the ranks AGREED on -- the status word summed at the end of a sharded proof -- left every rank in step: the group stays usable.)
static int local_group_fail(zk_comm *c, int rc) {
    if (rc != ZK_OK && c && c->local && !c->in_step) c->local->g->abort();
    if (c) c->in_step = false;
    return rc;
}
int zk_comm_local_group_new(int nranks, zk_comm_local_group **out) {
    if (!out || nranks < 1 || nranks > 1024) return ZK_E_ARG;
    *out = new zk_comm_local_group(nranks);
    return ZK_OK;
}
int zk_comm_local_group_free(zk_comm_local_group *g) {
    if (!g) return ZK_OK;
    if (g->ends.load() != 0) {                              // an end still points into the group: freeing it now would be a use after free
        set_last_error("zk_comm_local_group_free: free every communicator made from the group first (zk_comm_free)");
        return ZK_E_ARG;
    }
    delete g;
    return ZK_OK;
}
int zk_comm_local_group_abort(zk_comm_local_group *g) {
    if (!g) return ZK_E_ARG;
    g->abort();
    return ZK_OK;
}
int zk_comm_from_local_group(zk_comm_local_group *g, int rank, zk_comm **out) {
    if (!g || !out || rank < 0 || rank >= g->nranks) return ZK_E_ARG;
    LocalEnd *e = new LocalEnd{g, rank};
    zk_comm_host_ops ops{e, local_all_reduce, local_all_gather, local_gather, local_broadcast};
    zk_comm *c = new zk_comm{1, g->nranks, rank, nullptr, ops, 0, 0, {}, {}};
    c->local = e;
    g->ends++;
    *out = c;
    return ZK_OK;
}
int zk_comm_free(zk_comm *c) {
    if (!c) return ZK_OK;
    if (c->kind == 0 && c->nccl) (void)rccl().CommDestroy(c->nccl);
    if (c->local) c->local->g->ends--;
    delete c->local;
    delete c;
    return ZK_OK;
}
int zk_comm_rank(const zk_comm *c) { return c ? c->rank : -1; }
int zk_comm_size(const zk_comm *c) { return c ? c->nranks : -1; }
const char *zk_comm_backend(const zk_comm *c) { return !c ? "" : c->kind == 0 ? "rccl" : c->local ? "local-threads" : "host-ops"; }
int zk_comm_stats(const zk_comm *c, uint64_t *bytes_received, uint64_t *collectives) {
    if (!c) return ZK_E_ARG;
    if (bytes_received) *bytes_received = c->bytes_rx;
    if (collectives) *collectives = c->ncoll;
    return ZK_OK;
}
int zk_comm_all_reduce_sum_i64(zk_comm *c, void *dev_buf, size_t count) {
    if (!c || !dev_buf) return ZK_E_ARG;
    return c->all_reduce_i64(dev_buf, count);
}
int zk_comm_all_gather(zk_comm *c, const void *dev_send, void *dev_recv, size_t bytes) {
    if (!c || !dev_send || !dev_recv) return ZK_E_ARG;
    return c->all_gather(dev_send, dev_recv, bytes);
}
int zk_comm_host_exchange(zk_comm *c, int op, void *host_buf, void *host_recv, size_t n, int root) {
    if (!c || !host_buf || c->kind != 1 || root < 0 || root >= c->nranks) return ZK_E_ARG;
    if ((op == 1 || (op == 2 && c->rank == root)) && !host_recv) return ZK_E_ARG;
    if (c->nranks == 1) {
        if (op == 1 || op == 2) memcpy(host_recv, host_buf, n);
        return op >= 0 && op <= 3 ? ZK_OK : ZK_E_ARG;
    }
    switch (op) {
    case 0: ZK_CB(c->ops.all_reduce_sum_i64(c->ops.ctx, (int64_t *)host_buf, n), "all_reduce_sum_i64"); return ZK_OK;
    case 1: ZK_CB(c->ops.all_gather(c->ops.ctx, host_buf, host_recv, n), "all_gather"); return ZK_OK;
    case 2: ZK_CB(c->ops.gather(c->ops.ctx, host_buf, c->rank == root ? host_recv : nullptr, n, root), "gather"); return ZK_OK;
    case 3: ZK_CB(c->ops.broadcast(c->ops.ctx, host_buf, n, root), "broadcast"); return ZK_OK;
    }
    return ZK_E_ARG;
}
int zk_comm_broadcast(zk_comm *c, void *dev_buf, size_t bytes, int root) {
    if (!c || !dev_buf || root < 0 || root >= c->nranks) return ZK_E_ARG;
    return c->broadcast(dev_buf, bytes, root);
}

static int sharded_sumcheck_basic_prove_impl(zk_comm *c, const zk_table *shard, int absorb_table, uint64_t *claimed_sum, uint64_t *round_polys,
                                    uint64_t *challenges) {
    if (!c || !shard || !claimed_sum || !round_polys) return ZK_E_ARG;
    if (!is_pow2(shard->len)) return ZK_E_NOT_POW2;                       // Prover::init -> MultilinearPolynomial::new (prover.rs:23)
    if (!is_pow2((size_t)c->nranks)) return ZK_E_ARG;
    ZK_TRY(require_device());
    zk_transcript t;
    if (absorb_table) ZK_DISPATCH_FIELD(shard->field, ZK_TRY(sharded_absorb<F>(c, t.t, shard)));   // prover.rs:38-39
    if (shard->len * (size_t)c->nranks == 1) {                            // zero variables: the sum is the entry, no rounds
        ZK_HIP(memcpy_on_stream(claimed_sum, shard->dptr, (size_t)field_limbs64(shard->field) * 8, hipMemcpyDeviceToHost));
        return ZK_OK;
    }
    const zk_table *tabs[1] = {shard};
    return sharded_rounds(c, shard->field, 0, tabs, 1, 1, &t, claimed_sum, round_polys, challenges, nullptr);
}
int zk_sharded_sumcheck_basic_prove(zk_comm *c, const zk_table *shard, int absorb_table, uint64_t *claimed_sum, uint64_t *round_polys,
                                    uint64_t *challenges) {
    const int rc = sharded_sumcheck_basic_prove_impl(c, shard, absorb_table, claimed_sum, round_polys, challenges);
    return local_group_fail(c, rc);
}


static int sharded_sumcheck_gkr_prove_impl(zk_comm *c, const zk_table *const *shards, size_t nprod, size_t nfac, const uint64_t *claimed_sum,
                                  zk_transcript *t, uint64_t *round_coeffs, uint64_t *challenges, uint64_t *final_values) {
    if (!c || !shards || !claimed_sum || !t || !round_coeffs || !challenges) return ZK_E_ARG;
    if (nprod == 0 || nfac == 0 || nprod * nfac > 64) return ZK_E_ARG;
    for (size_t k = 0; k < nprod * nfac; k++) {
        if (!shards[k] || shards[k]->field != shards[0]->field) return ZK_E_ARG;
        if (!is_pow2(shards[k]->len)) return ZK_E_NOT_POW2;
        if (shards[k]->len != shards[0]->len) return ZK_E_NVARS;          // sum_polynomial.rs:17-23
    }
    if (!is_pow2((size_t)c->nranks)) return ZK_E_ARG;
    ZK_TRY(require_device());
    const int field = shards[0]->field;
    const size_t L64 = (size_t)field_limbs64(field);
    uint8_t be[48];
    ZK_TRY(zk_fe_to_bytes_be(field, claimed_sum, be));
    t->t.append(be, L64 * 8);                                             // sumcheck_gkr_protocol.rs:35
    if (shards[0]->len * (size_t)c->nranks == 1) {
        if (final_values)
            for (size_t k = 0; k < nprod * nfac; k++)
                ZK_HIP(memcpy_on_stream(final_values + k * L64, shards[k]->dptr, L64 * 8, hipMemcpyDeviceToHost));
        return ZK_OK;
    }
    if (nprod < 2 || nfac < 2) return ZK_E_NEED_TWO;                      // generate_round_univariate panics (sum_polynomial.rs:58-61)
    return sharded_rounds(c, field, 1, shards, nprod, nfac, t, nullptr, round_coeffs, challenges, final_values);
}
int zk_sharded_sumcheck_gkr_prove(zk_comm *c, const zk_table *const *shards, size_t nprod, size_t nfac, const uint64_t *claimed_sum,
                                  zk_transcript *t, uint64_t *round_coeffs, uint64_t *challenges, uint64_t *final_values) {
    const int rc = sharded_sumcheck_gkr_prove_impl(c, shards, nprod, nfac, claimed_sum, t, round_coeffs, challenges, final_values);
    return local_group_fail(c, rc);
}


static int sharded_mle_evaluate_impl(zk_comm *c, const zk_table *shard, const uint64_t *values, size_t nvalues, uint64_t *out) {
    if (!c || !shard || (!values && nvalues) || !out) return ZK_E_ARG;
    if (!is_pow2(shard->len) || !is_pow2((size_t)c->nranks)) return ZK_E_NOT_POW2;
    ZK_TRY(require_device());
    const size_t G = (size_t)c->nranks, m = ilog2(shard->len), k = ilog2(G);
    if (nvalues != m + k) return ZK_E_RANGE;
    const int field = shard->field;
    const size_t L64 = (size_t)field_limbs64(field), esz = L64 * 8;
    // evaluation_form.rs:27-29, local: variable 0 pairs equal low bits, so the first m values fold the shard down to this rank's entry
    // of the G-entry table of what is left (zk_mle_evaluate: up to four variables per pass)
    uint64_t mine[6];
    ZK_TRY(zk_mle_evaluate(shard, values, m, mine));
    DevBuf snd, all;
    ZK_TRY(snd.alloc(esz));
    ZK_TRY(all.alloc(G * esz));
    ZK_HIP(memcpy_on_stream(snd.p, mine, esz, hipMemcpyHostToDevice));
    ZK_TRY(c->all_gather(snd.p, all.p, esz));                             // entry g = rank g's value = global index g
    zk_table rep{field, G, all.p, 0};
    return zk_mle_evaluate(&rep, values + m * L64, k, out);
}
int zk_sharded_mle_evaluate(zk_comm *c, const zk_table *shard, const uint64_t *values, size_t nvalues, uint64_t *out) {
    const int rc = sharded_mle_evaluate_impl(c, shard, values, nvalues, out);
    return local_group_fail(c, rc);
}


static int sharded_msm_g1_impl(zk_comm *c, const zk_table *scalars_slice, const zk_g1_bases *bases_slice, int window_bits, uint64_t *out12,
                      zk_msm_stats *stats) {
    if (!c || !scalars_slice || !bases_slice || !out12) return ZK_E_ARG;
    uint64_t mine[12];
    ZK_TRY(zk_msm_g1(scalars_slice, bases_slice, window_bits, mine, stats));
    const size_t G = (size_t)c->nranks;
    DevBuf snd, rcv;
    ZK_TRY(snd.alloc(96));
    ZK_TRY(rcv.alloc(96 * G));
    ZK_HIP(memcpy_on_stream(snd.p, mine, 96, hipMemcpyHostToDevice));
    ZK_TRY(c->all_gather(snd.p, rcv.p, 96));                              // G affine points, 96 B each
    std::vector<uint64_t> all(12 * G);
    ZK_HIP(memcpy_on_stream(all.data(), rcv.p, 96 * G, hipMemcpyDeviceToHost));
    uint64_t acc[12];
    memcpy(acc, all.data(), 96);
    for (size_t g = 1; g < G; g++) {                                      // G - 1 additions, same order on every rank
        uint64_t nx[12];
        ZK_TRY(zk_g1_add(acc, all.data() + 12 * g, nx));
        memcpy(acc, nx, 96);
    }
    memcpy(out12, acc, 96);
    return ZK_OK;
}
int zk_sharded_msm_g1(zk_comm *c, const zk_table *scalars_slice, const zk_g1_bases *bases_slice, int window_bits, uint64_t *out12,
                      zk_msm_stats *stats) {
    const int rc = sharded_msm_g1_impl(c, scalars_slice, bases_slice, window_bits, out12, stats);
    return local_group_fail(c, rc);
}


// open_and_prove (multilinear_kzg.rs:50-126) of a low-bit-sharded table.  Proof i < m (the local variables): quotient and pre-summed
// bases of round i pair global indices that share their low bits, so pi_i = sum over the ranks of a LOCAL MSM (local quotient x the
// opening key of the rank's own bases P_{jG+g}) -- one all-gather of m points per rank and G - 1 additions per proof.  The last k
// proofs come from the G-entry table of leftovers (one per rank) against the per-rank totals of the bases, both all-gathered, and
// are computed replicated.
static int sharded_kzg_open_impl(zk_comm *c, const zk_table *shard, const zk_g1_bases *bases_local, const zk_kzg_opening_key *key_local,
                        const uint64_t *opening, size_t nopen, uint64_t *evaluation, uint64_t *proofs) {
    if (!c || !shard || !bases_local || !opening || !evaluation || !proofs) return ZK_E_ARG;
    if (shard->field != ZK_FR381) return ZK_E_ARG;
    if (!is_pow2(shard->len) || !is_pow2((size_t)c->nranks)) return ZK_E_NOT_POW2;
    const size_t G = (size_t)c->nranks, m = ilog2(shard->len), k = ilog2(G);
    if (nopen != m + k) return ZK_E_KZG_LEN;                                       // :55-59 on the global table
    if (zk_g1_bases_len(bases_local) != shard->len) return ZK_E_KZG_LEN;
    ZK_TRY(require_device());
    zk_kzg_opening_key *own = nullptr;
    if (!key_local && m >= 1) {
        ZK_TRY(zk_kzg_opening_key_new(bases_local, &own));
        key_local = own;
    }
    struct KeyGuard { zk_kzg_opening_key *k; ~KeyGuard() { zk_kzg_opening_key_free(k); } } key_guard{own};
    uint64_t v[4], left[4], dummy[4];
    ZK_TRY(zk_sharded_mle_evaluate(c, shard, opening, nopen, v));                  // :70, the global evaluation
    memcpy(evaluation, v, 32);
    // local rounds: this rank's share of the first m proofs, and its entry of (f - v) folded over the local variables
    std::vector<uint64_t> mine(12 * (m ? m : 1), 0);
    ZK_TRY(kzg_open_core(shard, bases_local, key_local, opening, m, v, dummy, mine.data(), left));
    if (G == 1) {
        memcpy(proofs, mine.data(), 96 * m);
        return ZK_OK;
    }
    // exchange: [m partial proofs | total of the local bases | leftover entry] per rank, one all-gather
    uint64_t total[12];
    ZK_TRY(kzg_key_total(key_local, bases_local, total));
    const size_t words = 12 * m + 12 + 4;
    std::vector<uint64_t> sendh(words), allh(words * G);
    memcpy(sendh.data(), mine.data(), 96 * m);
    memcpy(sendh.data() + 12 * m, total, 96);
    memcpy(sendh.data() + 12 * m + 12, left, 32);
    DevBuf snd, rcv;
    ZK_TRY(snd.alloc(words * 8));
    ZK_TRY(rcv.alloc(words * 8 * G));
    ZK_HIP(memcpy_on_stream(snd.p, sendh.data(), words * 8, hipMemcpyHostToDevice));
    ZK_TRY(c->all_gather(snd.p, rcv.p, words * 8));
    ZK_HIP(memcpy_on_stream(allh.data(), rcv.p, words * 8 * G, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < m; i++) {                                               // pi_i = sum_g pi_i^(g), same order on every rank
        uint64_t acc[12], nx[12];
        memcpy(acc, allh.data() + 12 * i, 96);
        for (size_t g = 1; g < G; g++) {
            ZK_TRY(zk_g1_add(acc, allh.data() + g * words + 12 * i, nx));
            memcpy(acc, nx, 96);
        }
        memcpy(proofs + 12 * i, acc, 96);
    }
    // the last k rounds on the G leftovers (entry g = rank g: the low index bits are the last variables) against the G totals
    std::vector<uint64_t> tab(4 * G), pts(12 * G);
    for (size_t g = 0; g < G; g++) {
        memcpy(pts.data() + 12 * g, allh.data() + g * words + 12 * m, 96);
        memcpy(tab.data() + 4 * g, allh.data() + g * words + 12 * m + 12, 32);
    }
    zk_table *t = nullptr;
    zk_g1_bases *b = nullptr;
    ZK_TRY(zk_table_upload(ZK_FR381, tab.data(), G, &t));
    int rc = zk_g1_bases_upload(pts.data(), G, &b);
    const uint64_t zero[4] = {0, 0, 0, 0};
    if (rc == ZK_OK) rc = kzg_open_core(t, b, nullptr, opening + 4 * m, k, zero, dummy, proofs + 12 * m, nullptr);
    zk_table_free(t);
    zk_g1_bases_free(b);
    return rc;
}
int zk_sharded_kzg_open(zk_comm *c, const zk_table *shard, const zk_g1_bases *bases_local, const zk_kzg_opening_key *key_local,
                        const uint64_t *opening, size_t nopen, uint64_t *evaluation, uint64_t *proofs) {
    const int rc = sharded_kzg_open_impl(c, shard, bases_local, key_local, opening, nopen, evaluation, proofs);
    return local_group_fail(c, rc);
}


}  // extern "C"
