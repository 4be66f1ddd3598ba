// zkmle_core.hip -- C ABI: device context, HBM-resident tables and the MLE operations.
// Product path only: every compute entry point runs HIP kernels; there is no CPU fallback.
#include <string.h>

#include <chrono>
#include <mutex>
#include <vector>

#include "context.h"
#include "mle_kernels.cuh"
#include "fold_multi.h"

namespace zk {

static thread_local std::string g_last_error;
void set_last_error(const std::string &s) { g_last_error = s; }

static thread_local hipStream_t g_stream = nullptr;
hipStream_t cur_stream() { return g_stream; }
hipError_t stream_wait_idle() {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t e = hipStreamQuery(g_stream);
        if (e != hipErrorNotReady) return e;
        if (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() > 300.0) return hipStreamSynchronize(g_stream);
    }
}
// Small copies (a few field elements down, a list of positions up) wait by polling: a blocking hipStreamSynchronize wakes up 20-30 us after the copy has
// landed, and a proof made of many small calls (the dense GKR prover: ~30 such copies) spent a third of its time there.  Large ones block at once.
hipError_t memcpy_on_stream(void *dst, const void *src, size_t bytes, hipMemcpyKind kind) {
    hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, g_stream);
    if (e != hipSuccess) return e;
    return bytes <= ((size_t)1 << 20) ? stream_wait_idle() : hipStreamSynchronize(g_stream);
}
hipError_t memset_on_stream(void *dst, int value, size_t bytes) { return hipMemsetAsync(dst, value, bytes, g_stream); }

struct DeviceScratch {
    void *dev = nullptr;
    size_t dev_bytes = 0;
    void *host = nullptr;
    size_t host_bytes = 0;
    void *syncw = nullptr;
    void *pair[2] = {nullptr, nullptr};      // pinned double buffer of the transcript absorb
    size_t pair_bytes = 0;
    void *mbox = nullptr, *mbox_dev = nullptr;   // coherent mailbox page (host-assisted transcript step)
};
static std::mutex g_mu;                                  // guards the caching pool
// Reduction partials and staging buffers are per THREAD and per device: two threads driving distinct handles never share
// them (a handle itself is used by one thread at a time, include/zkmle.h).  Freed when the thread ends.
struct ScratchSet {
    std::vector<DeviceScratch> v;
    ~ScratchSet() {
        for (DeviceScratch &s : v) {
            if (s.dev) (void)hipFree(s.dev);
            if (s.syncw) (void)hipFree(s.syncw);
            if (s.host) (void)hipHostFree(s.host);
            for (void *p : s.pair)
                if (p) (void)hipHostFree(p);
            if (s.mbox) (void)hipHostFree(s.mbox);
        }
    }
};
static thread_local ScratchSet g_scratch;

int require_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_last_error(std::string("no usable HIP device: ") + (e != hipSuccess ? hipGetErrorString(e) : "device count 0"));
        return ZK_E_NO_DEVICE;
    }
    return ZK_OK;
}

static int current_scratch(DeviceScratch **out) {
    int dev = 0;
    ZK_HIP(hipGetDevice(&dev));
    if ((int)g_scratch.v.size() <= dev) g_scratch.v.resize(dev + 1);
    *out = &g_scratch.v[dev];
    return ZK_OK;
}

int scratch(size_t bytes, void **out) {
    DeviceScratch *s;
    ZK_TRY(current_scratch(&s));
    if (s->dev_bytes < bytes) {
        if (s->dev) ZK_HIP(hipFree(s->dev));
        s->dev = nullptr;
        s->dev_bytes = 0;
        size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
        ZK_HIP(hipMalloc(&s->dev, want));
        s->dev_bytes = want;
    }
    *out = s->dev;
    return ZK_OK;
}

int sync_words(void **out) {
    DeviceScratch *s;
    ZK_TRY(current_scratch(&s));
    if (!s->syncw) {
        ZK_HIP(hipMalloc(&s->syncw, kSyncWordsBytes));
        ZK_HIP(hipMemsetAsync(s->syncw, 0, kSyncWordsBytes, g_stream));
        ZK_HIP(hipStreamSynchronize(g_stream));
    }
    *out = s->syncw;
    return ZK_OK;
}
int sync_words_reset() {
    DeviceScratch *s;
    ZK_TRY(current_scratch(&s));
    if (s->syncw) {
        ZK_HIP(hipMemsetAsync(s->syncw, 0, kSyncWordsBytes, g_stream));
        ZK_HIP(hipStreamSynchronize(g_stream));
    }
    return ZK_OK;
}

int host_staging(size_t bytes, void **out) {
    DeviceScratch *s;
    ZK_TRY(current_scratch(&s));
    if (s->host_bytes < bytes) {
        if (s->host) ZK_HIP(hipHostFree(s->host));
        s->host = nullptr;
        s->host_bytes = 0;
        size_t want = bytes < 4096 ? 4096 : bytes;
        ZK_HIP(hipHostMalloc(&s->host, want, hipHostMallocDefault));
        s->host_bytes = want;
    }
    *out = s->host;
    return ZK_OK;
}

int host_mailbox(void **host, void **dev) {
    DeviceScratch *s;
    ZK_TRY(current_scratch(&s));
    if (!s->mbox) {
        ZK_HIP(hipHostMalloc(&s->mbox, 16384, hipHostMallocCoherent | hipHostMallocMapped));   // dev_transcript.cuh kHostMailboxBytes
        memset(s->mbox, 0, 16384);
        ZK_HIP(hipHostGetDevicePointer(&s->mbox_dev, s->mbox, 0));
    }
    *host = s->mbox;
    *dev = s->mbox_dev;
    return ZK_OK;
}

int pinned_pair(size_t bytes, void *out[2]) {
    DeviceScratch *s;
    ZK_TRY(current_scratch(&s));
    if (s->pair_bytes < bytes) {
        for (int k = 0; k < 2; k++) {
            if (s->pair[k]) ZK_HIP(hipHostFree(s->pair[k]));
            s->pair[k] = nullptr;
        }
        s->pair_bytes = 0;
        for (int k = 0; k < 2; k++) ZK_HIP(hipHostMalloc(&s->pair[k], bytes, hipHostMallocDefault));
        s->pair_bytes = bytes;
    }
    out[0] = s->pair[0];
    out[1] = s->pair[1];
    return ZK_OK;
}

struct PoolBlock { void *p; size_t cap; int dev; bool busy; hipStream_t stream; };   // stream = the one its last user ran on
static std::vector<PoolBlock> g_pool;
static size_t g_pool_bytes = 0;
// cached + in-use scratch per process: a third of the current device's memory (96 GB of an MI355X's 288), asked of the device the first time it matters
static size_t pool_limit() {
    static const size_t v = [] {
        size_t free_b = 0, total_b = 0;
        return hipMemGetInfo(&free_b, &total_b) == hipSuccess && total_b ? total_b / 3 : (size_t)96 << 30;
    }();
    return v;
}

int pool_alloc(size_t bytes, void **out) {
    if (bytes == 0) bytes = 16;
    int dev = 0;
    ZK_HIP(hipGetDevice(&dev));
    {
        std::lock_guard<std::mutex> lk(g_mu);
        PoolBlock *best = nullptr;
        for (PoolBlock &b : g_pool)
            if (!b.busy && b.dev == dev && b.stream == cur_stream() && b.cap >= bytes && b.cap <= 2 * bytes + (1u << 20) && (!best || b.cap < best->cap)) best = &b;
        if (best) { best->busy = true; *out = best->p; return ZK_OK; }
    }
    if (g_pool_bytes + bytes > pool_limit()) ZK_TRY(zk_release_cached_memory());
    void *p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e == hipErrorOutOfMemory) {                         // blocks cached for other streams / sizes may be what is in the way: give them back and try once more
        (void)hipGetLastError();
        ZK_TRY(zk_release_cached_memory());
        e = hipMalloc(&p, bytes);
    }
    ZK_HIP(e);
    std::lock_guard<std::mutex> lk(g_mu);
    g_pool.push_back(PoolBlock{p, bytes, dev, true, cur_stream()});
    g_pool_bytes += bytes;
    *out = p;
    return ZK_OK;
}
void pool_free(void *p) {
    if (!p) return;
    std::lock_guard<std::mutex> lk(g_mu);
    for (PoolBlock &b : g_pool)
        if (b.p == p) { b.busy = false; return; }
}

int table_alloc_pooled(int field, size_t len, zk_table **out) {
    int limbs = field_limbs64(field);
    if (limbs < 0 || !out || len == 0) return ZK_E_ARG;
    ZK_TRY(require_device());
    void *d = nullptr;
    ZK_TRY(pool_alloc(len * (size_t)limbs * 8, &d));
    *out = new zk_table{field, len, d, 2};
    return ZK_OK;
}

template <class F> static Fe<F> load_host(const uint64_t *src) {
    Fe<F> e;
    memcpy(e.l, src, sizeof(uint32_t) * F::N);
    return e;
}

// alpha fold(in, rb) + beta fold(in, rc), both folding the k top variables (gkr/src/utils.rs:23-68), in one pass (mle_kernels.cuh
// fold_alpha_beta_kernel).  out: a table of in->len >> k entries.  Host values in, nothing uploaded: they travel as kernel arguments.
int mle_fold_alpha_beta(const zk_table *in, size_t k, const uint64_t *alpha, const uint64_t *beta, const uint64_t *rb, const uint64_t *rc, zk_table *out) {
    if (!in || !out || !alpha || !beta || !rb || !rc) return ZK_E_ARG;
    if (k < 1 || k > (size_t)kFoldABMax) return ZK_E_RANGE;
    if (!is_pow2(in->len) || (in->len >> k) == 0) return ZK_E_NOT_POW2;
    const size_t n = in->len >> k;
    if (out->field != in->field || out->len < n || out->dptr == in->dptr) return ZK_E_ARG;
    ZK_TRY(require_device());
    const int limbs = field_limbs64(in->field);
    ZK_DISPATCH_FIELD(in->field, {
        FoldABArgs<F> a{};
        for (size_t l = 0; l < k; l++) { a.rb[l] = load_host<F>(rb + l * limbs); a.rc[l] = load_host<F>(rc + l * limbs); }
        a.alpha = load_host<F>(alpha);
        a.beta = load_host<F>(beta);
        a.k = (int)k;
        // lanes per output: enough lanes to fill the chip (2^20), at least two inputs per lane, at least 16 neighbouring outputs per part (512 contiguous bytes)
        unsigned split = 1;
        while ((n * split) < ((size_t)1 << 20) && split * 2 <= (1u << k) / 2 && split * 2 <= (unsigned)kBlock / 16) split *= 2;
        const size_t nblk = (n + kBlock / split - 1) / (kBlock / split);
        void *w = nullptr;
        ZK_TRY(pool_alloc(sizeof(Ufe<F>) << k, &w));
        fold_alpha_beta_weights_kernel<F><<<1, 1 << kFoldABMax, 0, cur_stream()>>>(a, (Ufe<F> *)w);
        fold_alpha_beta_kernel<F><<<(unsigned)(nblk < (size_t)kMaxBlocks ? nblk : (size_t)kMaxBlocks), kBlock, 0, cur_stream()>>>(in->dptr, out->dptr, n, (int)k,
                                                                                                                                (const Ufe<F> *)w, split);
        pool_free(w);                                                // stream-ordered: this thread's next user of the block runs behind the kernel
    });
    ZK_HIP(hipGetLastError());
    out->len = n;
    return ZK_OK;
}


}  // namespace zk

using namespace zk;

template <int OP> static int elementwise(const zk_table *a, const zk_table *b, const uint64_t *scalar, zk_table *out,
                                         size_t outlen, void *stream) {
    ZK_TRY(require_device());
    hipStream_t s = stream ? (hipStream_t)stream : cur_stream();
    uint64_t zero[6] = {0, 0, 0, 0, 0, 0};
    const uint64_t *sc = scalar ? scalar : zero;
    ZK_DISPATCH_FIELD(a->field, (elementwise_kernel<F, OP><<<grid_for(outlen), kBlock, 0, s>>>(
                                    a->dptr, b ? b->dptr : nullptr, out->dptr, outlen, load_host<F>(sc))));
    ZK_HIP(hipGetLastError());
    out->len = outlen;
    return ZK_OK;
}
extern "C" {

const char *zk_status_message(int status) {
    switch (status) {
        case ZK_OK: return "ok";
        case ZK_E_NOT_POW2: return "Evaluated values must be a power of 2";
        case ZK_E_LEN_MISMATCH: return "Different polynomial length";
        case ZK_E_NVARS: return "different number of variables";
        case ZK_E_NEED_TWO: return "more than one polynomial required for mul operation";
        case ZK_E_KZG_LEN: return "Polynomial evaluation must match g1 length";
        case ZK_E_RANGE: return "index out of range";
        case ZK_E_ARG: return "bad argument";
        case ZK_E_NOMEM: return "out of memory";
        case ZK_E_NO_DEVICE: return "no usable HIP device (libzkmle_amd has no CPU fallback)";
        case ZK_E_HIP: return "HIP runtime error";
        case ZK_E_NOT_INIT: return "Can't prove without init";
        case ZK_E_COMM: return "communicator error (RCCL or exchange callback)";
        default: return "unknown status";
    }
}
const char *zk_last_error(void) { return g_last_error.c_str(); }
const char *zk_version(void) { return "zkmle_amd 0.1 (gfx950)"; }

int zk_device_count(int *count) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    *count = (e == hipSuccess) ? n : 0;
    return ZK_OK;
}
int zk_init(int device) {
    ZK_TRY(require_device());
    ZK_HIP(hipSetDevice(device));
    return ZK_OK;
}
int zk_field_limbs(int field) {
    int n = field_limbs64(field);
    return n < 0 ? ZK_E_ARG : n;
}
int zk_release_cached_memory(void) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return ZK_OK;
    std::lock_guard<std::mutex> lk(g_mu);
    for (size_t i = 0; i < g_pool.size();) {
        if (!g_pool[i].busy && g_pool[i].dev == dev) {
            (void)hipFree(g_pool[i].p);
            g_pool_bytes -= g_pool[i].cap;
            g_pool[i] = g_pool.back();
            g_pool.pop_back();
        } else {
            i++;
        }
    }
    return ZK_OK;
}
int zk_device_synchronize(void) {
    ZK_TRY(require_device());
    ZK_HIP(hipDeviceSynchronize());
    return ZK_OK;
}
// The calling thread's stream for everything the library launches from now on (NULL = the default stream).  Cached scratch
// blocks are reused only on the stream they were last used on, so no synchronisation is needed here.
int zk_set_stream(void *stream) {
    g_stream = (hipStream_t)stream;
    return ZK_OK;
}
void *zk_get_stream(void) { return (void *)g_stream; }

// ---- tables ---------------------------------------------------------------------------------------
int zk_table_alloc(int field, size_t len, zk_table **out) {
    int limbs = field_limbs64(field);
    if (limbs < 0 || !out || len == 0) return ZK_E_ARG;
    ZK_TRY(require_device());
    void *d = nullptr;
    ZK_HIP(hipMalloc(&d, len * (size_t)limbs * 8));
    *out = new zk_table{field, len, d, 1};
    return ZK_OK;
}
int zk_table_upload(int field, const uint64_t *host, size_t len, zk_table **out) {
    if (!host || !out) return ZK_E_ARG;
    if (!is_pow2(len)) return ZK_E_NOT_POW2;   // MultilinearPolynomial::new, evaluation_form.rs:13
    return zk_table_upload_raw(field, host, len, out);
}
int zk_table_upload_raw(int field, const uint64_t *host, size_t len, zk_table **out) {
    if (!host || !out) return ZK_E_ARG;
    ZK_TRY(zk_table_alloc(field, len, out));
    hipError_t e = zk::memcpy_on_stream((*out)->dptr, host, len * (size_t)field_limbs64(field) * 8, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        zk_table_free(*out);
        *out = nullptr;
        ZK_HIP(e);
    }
    return ZK_OK;
}
int zk_table_download(const zk_table *t, uint64_t *host) {
    if (!t || !host) return ZK_E_ARG;
    ZK_HIP(zk::memcpy_on_stream(host, t->dptr, t->len * (size_t)field_limbs64(t->field) * 8, hipMemcpyDeviceToHost));
    return ZK_OK;
}
int zk_table_free(zk_table *t) {
    if (!t) return ZK_OK;
    if (t->owned == 1 && t->dptr) ZK_HIP(hipFree(t->dptr));
    if (t->owned == 2) pool_free(t->dptr);
    delete t;
    return ZK_OK;
}
size_t zk_table_len(const zk_table *t) { return t ? t->len : 0; }
int zk_table_field(const zk_table *t) { return t ? t->field : ZK_E_ARG; }
void *zk_table_device_ptr(zk_table *t) { return t ? t->dptr : nullptr; }
int zk_table_wrap(int field, void *device_ptr, size_t len, zk_table **out) {
    if (field_limbs64(field) < 0 || !device_ptr || !out || len == 0) return ZK_E_ARG;
    *out = new zk_table{field, len, device_ptr, 0};
    return ZK_OK;
}
int zk_table_clone(const zk_table *t, zk_table **out) {
    if (!t || !out) return ZK_E_ARG;
    ZK_TRY(zk_table_alloc(t->field, t->len, out));
    ZK_HIP(zk::memcpy_on_stream((*out)->dptr, t->dptr, t->len * (size_t)field_limbs64(t->field) * 8, hipMemcpyDeviceToDevice));
    return ZK_OK;
}
int zk_table_fill_random(zk_table *t, uint64_t seed) {
    if (!t) return ZK_E_ARG;
    ZK_TRY(require_device());
    ZK_DISPATCH_FIELD(t->field, (fill_random_kernel<F><<<grid_for(t->len), kBlock, 0, cur_stream()>>>(t->dptr, t->len, seed, 0)));
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}
int zk_table_fill_random_strided(zk_table *t, uint64_t seed, size_t first, size_t stride) {
    if (!t || stride == 0) return ZK_E_ARG;
    ZK_TRY(require_device());
    ZK_DISPATCH_FIELD(t->field, (fill_random_kernel<F><<<grid_for(t->len), kBlock, 0, cur_stream()>>>(t->dptr, t->len, seed, first, stride)));
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}
int zk_host_fill_random(int field, uint64_t seed, size_t first, size_t count, uint64_t *out) {
    if (!out) return ZK_E_ARG;
    ZK_DISPATCH_FIELD(field, {
        for (size_t i = 0; i < count; i++) {
            Fe<F> e = random_element<F>(seed, first + i);
            memcpy(out + i * (F::N / 2), e.l, sizeof(uint32_t) * F::N);
        }
    });
    return ZK_OK;
}

// ---- fold -----------------------------------------------------------------------------------------
int zk_mle_fold_ptr(int field, const void *d_in, size_t len, size_t var, const uint64_t *value, void *d_out,
                    void *stream) {
    if (!d_in || !d_out || !value || field_limbs64(field) < 0) return ZK_E_ARG;
    if (!is_pow2(len)) return ZK_E_NOT_POW2;
    unsigned n = ilog2(len);
    if (len < 2) return ZK_E_NOT_POW2;      // result would be empty: new() asserts (evaluation_form.rs:105 -> :13)
    if (var + 1 > n) return ZK_E_RANGE;     // power = n - 1 - var underflows (:80)
    ZK_TRY(require_device());
    size_t half = len / 2;
    unsigned power = n - 1 - (unsigned)var;
    hipStream_t s = stream ? (hipStream_t)stream : cur_stream();
    if (var == 0 && (half + kBlock - 1) / kBlock <= (size_t)0x7fffffff) {
        unsigned grid = (unsigned)((half + kBlock - 1) / kBlock);
        ZK_DISPATCH_FIELD(field, (fold0_kernel<F><<<grid, kBlock, 0, s>>>(d_in, d_out, half, load_host<F>(value))));
    } else {
        ZK_DISPATCH_FIELD(field, (fold_kernel<F><<<grid_for(half), kBlock, 0, s>>>(d_in, d_out, half, power, load_host<F>(value))));
    }
    ZK_HIP(hipGetLastError());
    return ZK_OK;
}
int zk_mle_fold(const zk_table *in, size_t var, const uint64_t *value, zk_table *out, void *stream) {
    if (!in || !out) return ZK_E_ARG;
    if (out->field != in->field || out->len < in->len / 2 || out->dptr == in->dptr) return ZK_E_ARG;
    ZK_TRY(zk_mle_fold_ptr(in->field, in->dptr, in->len, var, value, out->dptr, stream));
    out->len = in->len / 2;
    return ZK_OK;
}

// ---- sums -----------------------------------------------------------------------------------------
static int sums_impl(const zk_table *t, int nseg, uint64_t *out) {
    if (!t || !out) return ZK_E_ARG;
    ZK_TRY(require_device());
    size_t esz = (size_t)field_limbs64(t->field) * 8;
    size_t seglen = t->len / nseg;
    int grid = reduce_grid_for(seglen);
    void *part, *host;
    ZK_TRY(scratch(esz * ((size_t)grid * nseg + nseg), &part));
    ZK_TRY(host_staging(esz * nseg, &host));
    void *res = (char *)part + esz * (size_t)grid * nseg;
    ZK_DISPATCH_FIELD(t->field, {
        segment_sums_kernel<F><<<grid, kBlock, 0, cur_stream()>>>(t->dptr, seglen, nseg, part);
        finish_sums_kernel<F><<<1, kBlock, 0, cur_stream()>>>(part, (size_t)grid, nseg, res);
    });
    ZK_HIP(hipGetLastError());
    ZK_HIP(zk::memcpy_on_stream(host, res, esz * nseg, hipMemcpyDeviceToHost));
    memcpy(out, host, esz * nseg);
    return ZK_OK;
}
int zk_mle_sum(const zk_table *t, uint64_t *out) { return sums_impl(t, 1, out); }
int zk_mle_half_sums(const zk_table *t, uint64_t *out2) {
    if (!t) return ZK_E_ARG;
    if (t->len < 2) return ZK_E_ARG;
    return sums_impl(t, 2, out2);
}

int zk_mle_fold_half_sums(const zk_table *in, const uint64_t *value, zk_table *out, uint64_t *out2, void *stream) {
    if (!in || !out || !value || !out2) return ZK_E_ARG;
    if (out->field != in->field || out->len < in->len / 2 || out->dptr == in->dptr) return ZK_E_ARG;
    if (!is_pow2(in->len)) return ZK_E_NOT_POW2;
    if (in->len < 4) return ZK_E_ARG;       // callers finish tables of < 4 entries on the host
    ZK_TRY(require_device());
    size_t esz = (size_t)field_limbs64(in->field) * 8;
    size_t q = in->len / 4;
    int grid = reduce_grid_for(q);
    void *part, *host;
    ZK_TRY(scratch(esz * ((size_t)grid * 2 + 2), &part));
    ZK_TRY(host_staging(esz * 2, &host));
    void *res = (char *)part + esz * (size_t)grid * 2;
    hipStream_t s = stream ? (hipStream_t)stream : cur_stream();
    ZK_DISPATCH_FIELD(in->field, {
        fold_half_sums_kernel<F><<<grid, kBlock, 0, s>>>(in->dptr, out->dptr, q, load_host<F>(value), part);
        finish_sums_kernel<F><<<1, kBlock, 0, s>>>(part, (size_t)grid, 2, res);
    });
    ZK_HIP(hipGetLastError());
    ZK_HIP(hipMemcpyAsync(host, res, esz * 2, hipMemcpyDeviceToHost, s));
    ZK_HIP(hipStreamSynchronize(s));
    memcpy(out2, host, esz * 2);
    out->len = in->len / 2;
    return ZK_OK;
}

// ---- element-wise / tensor ---------------------------------------------------------------------------
int zk_mle_scalar_mul(const zk_table *a, const uint64_t *scalar, zk_table *out, void *stream) {
    if (!a || !scalar || !out || out->field != a->field || out->len < a->len) return ZK_E_ARG;
    return elementwise<OP_SCALAR_MUL>(a, nullptr, scalar, out, a->len, stream);
}
int zk_mle_sub_scalar(const zk_table *a, const uint64_t *scalar, zk_table *out, void *stream) {
    if (!a || !scalar || !out || out->field != a->field || out->len < a->len) return ZK_E_ARG;
    return elementwise<OP_SUB_SCALAR>(a, nullptr, scalar, out, a->len, stream);
}
int zk_mle_add(const zk_table *a, const zk_table *b, zk_table *out, void *stream) {
    if (!a || !b || !out || a->field != b->field || out->field != a->field) return ZK_E_ARG;
    if (a->len != b->len) return ZK_E_LEN_MISMATCH;      // evaluation_form.rs:149-153
    if (out->len < a->len) return ZK_E_ARG;
    return elementwise<OP_ADD>(a, b, nullptr, out, a->len, stream);
}
static int tensor_impl(const zk_table *wb, const zk_table *wc, zk_table *out, void *stream, bool mul) {
    if (!wb || !wc || !out || wb->field != wc->field || out->field != wb->field) return ZK_E_ARG;
    if (wb->len != wc->len) return ZK_E_LEN_MISMATCH;    // :112 / :129-132
    size_t total = wb->len * wc->len;
    if (out->len < total) return ZK_E_ARG;
    ZK_TRY(require_device());
    hipStream_t s = stream ? (hipStream_t)stream : cur_stream();
    if (mul) {
        ZK_DISPATCH_FIELD(wb->field, (tensor_kernel<F, true><<<grid_for(total), kBlock, 0, s>>>(wb->dptr, wc->dptr, out->dptr, wb->len)));
    } else {
        ZK_DISPATCH_FIELD(wb->field, (tensor_kernel<F, false><<<grid_for(total), kBlock, 0, s>>>(wb->dptr, wc->dptr, out->dptr, wb->len)));
    }
    ZK_HIP(hipGetLastError());
    out->len = total;
    return ZK_OK;
}
int zk_mle_tensor_add(const zk_table *wb, const zk_table *wc, zk_table *out, void *stream) {
    return tensor_impl(wb, wc, out, stream, false);
}
int zk_mle_tensor_mul(const zk_table *wb, const zk_table *wc, zk_table *out, void *stream) {
    return tensor_impl(wb, wc, out, stream, true);
}

int zk_mle_to_bytes(const zk_table *t, uint8_t *host_out) {
    if (!t || !host_out) return ZK_E_ARG;
    zk_table *tmp = nullptr;
    ZK_TRY(zk_table_alloc(t->field, t->len, &tmp));
    int rc = elementwise<OP_TO_CANONICAL_BE>(t, nullptr, nullptr, tmp, t->len, nullptr);
    if (rc == ZK_OK) rc = zk_table_download(tmp, (uint64_t *)host_out);
    zk_table_free(tmp);
    return rc;
}

// ---- evaluate: clone + nvalues folds of variable 0, element 0 (evaluation_form.rs:21-33) ----------------
int zk_mle_evaluate(const zk_table *t, const uint64_t *values, size_t nvalues, uint64_t *out) {
    if (!t || !out || (nvalues && !values)) return ZK_E_ARG;
    if (!is_pow2(t->len)) return ZK_E_NOT_POW2;
    if (nvalues > ilog2(t->len)) return ZK_E_NOT_POW2;   // folding a 1-entry table: empty Vec -> assert :13
    ZK_TRY(require_device());
    int limbs = field_limbs64(t->field);
    size_t esz = (size_t)limbs * 8;
    if (nvalues == 0) {
        ZK_HIP(zk::memcpy_on_stream(out, t->dptr, esz, hipMemcpyDeviceToHost));
        return ZK_OK;
    }
    // ping-pong between two halves of one scratch allocation (len/2 + len/4 elements)
    zk_table *a = nullptr, *b = nullptr;
    ZK_TRY(table_alloc_pooled(t->field, t->len / 2, &a));
    int rc = ZK_OK;
    if (t->len >= 4) rc = table_alloc_pooled(t->field, t->len / 4, &b);
    const zk_table *cur = t;
    zk_table *dst = a, *other = b;
    // Large tables: up to kMultiMax variables per pass (foldk_seg_sums_kernel: the K folds as one lazily reduced weighted sum -- the
    // same field element as K successive folds, 1 + 2^-K table lengths of traffic per K variables instead of 3 (1 - 2^-K)).  The values
    // are read from device memory by that kernel.
    struct PoolBuf { void *p = nullptr; ~PoolBuf() { pool_free(p); } int alloc(size_t bytes) { return pool_alloc(bytes, &p); } } vdev;
    const size_t kMultiFrom = (size_t)1 << 16;
    if (rc == ZK_OK && nvalues >= 2 && t->len >= kMultiFrom) {
        rc = vdev.alloc(nvalues * esz);
        if (rc == ZK_OK && zk::memcpy_on_stream(vdev.p, values, nvalues * esz, hipMemcpyHostToDevice) != hipSuccess) rc = ZK_E_HIP;
    }
    size_t i = 0;
    while (rc == ZK_OK && i < nvalues) {
        if (cur->len <= 2 * (size_t)kEvalTailBlock && nvalues - i <= (size_t)kEvalTailVars) {   // the rest in one launch
            const int nv = (int)(nvalues - i);
            ZK_DISPATCH_FIELD(t->field, {
                EvalTailValues<F> vals;
                for (int k = 0; k < kEvalTailVars; k++) vals.r[k] = k < nv ? load_host<F>(values + (i + k) * limbs) : fe_zero<F>();
                evaluate_tail_kernel<F><<<1, kEvalTailBlock, 0, cur_stream()>>>(cur->dptr, dst->dptr, cur->len, vals, nv);
            });
            hipError_t e = hipGetLastError();
            if (e != hipSuccess) { set_last_error(hipGetErrorString(e)); rc = ZK_E_HIP; }
            cur = dst;
            break;
        }
        size_t k = nvalues - i < (size_t)kEvalMultiMax ? nvalues - i : (size_t)kEvalMultiMax;
        while (k > 1 && (cur->len >> k) < 2 * (size_t)kEvalTailBlock) k--;                       // leave the tail its share
        if (vdev.p && k >= 2 && cur->len >= kMultiFrom) {
            const void *rp[kMultiMax];
            for (size_t j = 0; j < k; j++) rp[j] = (const char *)vdev.p + (i + j) * esz;
            dst->len = cur->len >> k;
            ZK_DISPATCH_FIELD(t->field, { rc = (launch_foldk<F>(cur->dptr, dst->dptr, dst->len, (int)k, rp, 0, nullptr, nullptr)); });
            i += k;
        } else {
            dst->len = cur->len / 2;
            rc = zk_mle_fold_ptr(t->field, cur->dptr, cur->len, 0, values + i * limbs, dst->dptr, nullptr);
            i += 1;
        }
        cur = dst;
        zk_table *nx = other;
        other = dst;
        dst = nx;
    }
    if (rc == ZK_OK) {
        hipError_t e = zk::memcpy_on_stream(out, cur->dptr, esz, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { set_last_error(hipGetErrorString(e)); rc = ZK_E_HIP; }
    }
    zk_table_free(a);
    zk_table_free(b);
    return rc;
}

// ---- stateless host-buffer conveniences ------------------------------------------------------------------
int zk_host_partial_evaluate(int field, const uint64_t *poly, size_t len, size_t var, const uint64_t *value,
                             uint64_t *out) {
    if (!poly || !value || !out) return ZK_E_ARG;
    zk_table *in = nullptr, *o = nullptr;
    ZK_TRY(zk_table_upload(field, poly, len, &in));
    int rc = len >= 2 ? zk_table_alloc(field, len / 2, &o) : ZK_E_NOT_POW2;
    if (rc == ZK_OK) rc = zk_mle_fold(in, var, value, o, nullptr);
    if (rc == ZK_OK) rc = zk_table_download(o, out);
    zk_table_free(in);
    zk_table_free(o);
    return rc;
}
int zk_host_evaluate(int field, const uint64_t *poly, size_t len, const uint64_t *values, size_t nvalues,
                     uint64_t *out) {
    if (!poly || !out) return ZK_E_ARG;
    zk_table *in = nullptr;
    ZK_TRY(zk_table_upload(field, poly, len, &in));
    int rc = zk_mle_evaluate(in, values, nvalues, out);
    zk_table_free(in);
    return rc;
}

}  // extern "C"
