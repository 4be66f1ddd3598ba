// microbench_shapes.hip -- what HBM rate the access SHAPE of the fused GKR round reaches with no arithmetic behind it, against the shapes next to it
// (r4: the fused round's loads and stores alone reach 5.3-5.4 TB/s where the plain fold's reach 5.9; which property of the shape costs the 10 %?).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench_shapes.hip -o tools/microbench_shapes.bin ; tools/microbench_shapes.bin [log_n = 22] [reps = 100]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct Tabs { const uint4 *in[4]; uint4 *out[4]; };
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
template <int NT> __device__ __forceinline__ uint4 ld(const uint4 *p) {
    if (!NT) return *p;
    const v4u v = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
template <int NT> __device__ __forceinline__ void st(uint4 *p, uint4 v) {
    if (NT) { v4u w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, reinterpret_cast<v4u *>(p)); } else *p = v;
}
__device__ __forceinline__ uint4 x4(uint4 a, uint4 b) { return make_uint4(a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w); }
// one 32-byte element = two uint4
// the fused round: per pair index i, per table: entries i, i + q, i + 2 q, i + 3 q in; i, i + q out
template <int NT, int STORE> __global__ void __launch_bounds__(256) fused_shape(Tabs t, int ntab, size_t q) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < q; i += (size_t)gridDim.x * blockDim.x)
        for (int k = 0; k < ntab; k++) {
            const uint4 *s = t.in[k];
            uint4 a0l = ld<NT>(s + 2 * i), a0h = ld<NT>(s + 2 * i + 1), a1l = ld<NT>(s + 2 * (i + q)), a1h = ld<NT>(s + 2 * (i + q) + 1);
            uint4 b0l = ld<NT>(s + 2 * (i + 2 * q)), b0h = ld<NT>(s + 2 * (i + 2 * q) + 1), b1l = ld<NT>(s + 2 * (i + 3 * q)), b1h = ld<NT>(s + 2 * (i + 3 * q) + 1);
            uint4 *o = t.out[k];
            if (STORE) { st<NT>(o + 2 * i, x4(a0l, b0l)); st<NT>(o + 2 * i + 1, x4(a0h, b0h)); st<NT>(o + 2 * (i + q), x4(a1l, b1l)); st<NT>(o + 2 * (i + q) + 1, x4(a1h, b1h)); }
            else { uint4 v = x4(x4(a0l, b0l), x4(a1h, b1h)); v = x4(v, x4(x4(a0h, b0h), x4(a1l, b1l))); if (v.x == 0x12345678u && v.y == 0x9abcdef0u) o[0] = v; }
        }
}
// the same bytes with every store instruction writing 1 KiB contiguous (whole 128-byte lines) instead of the first or second half of 64 elements
template <int NT, int CLOAD> __global__ void __launch_bounds__(256) fused_shape_linestores(Tabs t, int ntab, size_t q) {
    const unsigned lane = threadIdx.x & 63u;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < q; i += (size_t)gridDim.x * blockDim.x) {
        const size_t w0 = i - lane;                              // the wave's first pair index
        for (int k = 0; k < ntab; k++) {
            const uint4 *s = t.in[k];
            uint4 a0l, a0h, a1l, a1h, b0l, b0h, b1l, b1h;
            if (CLOAD) {
                a0l = ld<NT>(s + 2 * w0 + lane); a0h = ld<NT>(s + 2 * w0 + 64 + lane); a1l = ld<NT>(s + 2 * (w0 + q) + lane); a1h = ld<NT>(s + 2 * (w0 + q) + 64 + lane);
                b0l = ld<NT>(s + 2 * (w0 + 2 * q) + lane); b0h = ld<NT>(s + 2 * (w0 + 2 * q) + 64 + lane); b1l = ld<NT>(s + 2 * (w0 + 3 * q) + lane); b1h = ld<NT>(s + 2 * (w0 + 3 * q) + 64 + lane);
            } else {
                a0l = ld<NT>(s + 2 * i); a0h = ld<NT>(s + 2 * i + 1); a1l = ld<NT>(s + 2 * (i + q)); a1h = ld<NT>(s + 2 * (i + q) + 1);
                b0l = ld<NT>(s + 2 * (i + 2 * q)); b0h = ld<NT>(s + 2 * (i + 2 * q) + 1); b1l = ld<NT>(s + 2 * (i + 3 * q)); b1h = ld<NT>(s + 2 * (i + 3 * q) + 1);
            }
            uint4 *o = t.out[k];
            st<NT>(o + 2 * w0 + lane, x4(a0l, b0l)); st<NT>(o + 2 * w0 + 64 + lane, x4(a0h, b0h));
            st<NT>(o + 2 * (w0 + q) + lane, x4(a1l, b1l)); st<NT>(o + 2 * (w0 + q) + 64 + lane, x4(a1h, b1h));
        }
    }
}
// the same round on tables of 36-byte elements (nine 29-bit limbs in 32-bit words: the products' internal form): what its
// loads and stores alone cost.  AoS: three loads per element (16 + 16 + 4 bytes at a 36-byte stride); SoA9: nine word planes, one coalesced dword each.
struct E36 { uint32_t w[9]; };
__device__ __forceinline__ E36 ld36(const E36 *p) { return *p; }
__device__ __forceinline__ E36 x36(const E36 &a, const E36 &b) { E36 r; for (int k = 0; k < 9; k++) r.w[k] = a.w[k] ^ b.w[k]; return r; }
__global__ void __launch_bounds__(256) fused_shape_36(Tabs t, int ntab, size_t q) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < q; i += (size_t)gridDim.x * blockDim.x)
        for (int k = 0; k < ntab; k++) {
            const E36 *s = reinterpret_cast<const E36 *>(t.in[k]);
            const E36 a0 = ld36(s + i), a1 = ld36(s + i + q), b0 = ld36(s + i + 2 * q), b1 = ld36(s + i + 3 * q);
            E36 *o = reinterpret_cast<E36 *>(t.out[k]);
            o[i] = x36(a0, b0);
            o[i + q] = x36(a1, b1);
        }
}
__global__ void __launch_bounds__(256) fused_shape_soa9(Tabs t, int ntab, size_t q) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < q; i += (size_t)gridDim.x * blockDim.x)
        for (int k = 0; k < ntab; k++) {
            const uint32_t *s = reinterpret_cast<const uint32_t *>(t.in[k]);
            uint32_t *o = reinterpret_cast<uint32_t *>(t.out[k]);
            uint32_t a0[9], a1[9], b0[9], b1[9];
            for (int w = 0; w < 9; w++) { a0[w] = s[(size_t)w * 4 * q + i]; a1[w] = s[(size_t)w * 4 * q + i + q]; b0[w] = s[(size_t)w * 4 * q + i + 2 * q]; b1[w] = s[(size_t)w * 4 * q + i + 3 * q]; }
            for (int w = 0; w < 9; w++) { o[(size_t)w * 2 * q + i] = a0[w] ^ b0[w]; o[(size_t)w * 2 * q + i + q] = a1[w] ^ b1[w]; }
        }
}
// wave w of a workgroup takes table w: 64 consecutive pair indices per workgroup
template <int NT> __global__ void __launch_bounds__(256) wave_per_table_shape(Tabs t, size_t q) {
    const int k = threadIdx.x >> 6;
    for (size_t i = (size_t)blockIdx.x * 64 + (threadIdx.x & 63); i < q; i += (size_t)gridDim.x * 64) {
        const uint4 *s = t.in[k];
        uint4 a0l = ld<NT>(s + 2 * i), a0h = ld<NT>(s + 2 * i + 1), a1l = ld<NT>(s + 2 * (i + q)), a1h = ld<NT>(s + 2 * (i + q) + 1);
        uint4 b0l = ld<NT>(s + 2 * (i + 2 * q)), b0h = ld<NT>(s + 2 * (i + 2 * q) + 1), b1l = ld<NT>(s + 2 * (i + 3 * q)), b1h = ld<NT>(s + 2 * (i + 3 * q) + 1);
        uint4 *o = t.out[k];
        st<NT>(o + 2 * i, x4(a0l, b0l)); st<NT>(o + 2 * i + 1, x4(a0h, b0h)); st<NT>(o + 2 * (i + q), x4(a1l, b1l)); st<NT>(o + 2 * (i + q) + 1, x4(a1h, b1h));
    }
}
// the plain fold: i, i + half in; i out; one output per lane
template <int NT> __global__ void __launch_bounds__(256) fold_shape(const uint4 *in, uint4 *out, size_t half) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    st<NT>(out + 2 * i, x4(ld<NT>(in + 2 * i), ld<NT>(in + 2 * (i + half))));
    st<NT>(out + 2 * i + 1, x4(ld<NT>(in + 2 * i + 1), ld<NT>(in + 2 * (i + half) + 1)));
}
int main(int argc, char **argv) {
    const int lg = argc > 1 ? atoi(argv[1]) : 22, reps = argc > 2 ? atoi(argv[2]) : 100;
    const size_t n = (size_t)1 << lg, q = n / 4, half = n / 2;
    // One slab; table k of the inputs at k * (n * 36 + skew), of the outputs behind them at k * (half * 36 + skew): `skew` bytes (argv[3], default
    // 0) on top of a spacing that is a power of two only when the elements are 32 bytes AND skew = 0 -- set ZK_SHAPES_POW2=1 to space the tables
    // n * 32 bytes apart exactly as separate power-of-two allocations lie.
    const size_t skew = argc > 3 ? (size_t)atol(argv[3]) : 0;
    const bool pow2 = getenv("ZK_SHAPES_POW2") != nullptr;
    const size_t in_sp = n * (pow2 ? 32 : 36) + skew, out_sp = half * (pow2 ? 32 : 36) + skew;
    char *slab;
    CK(hipMalloc((void **)&slab, 4 * in_sp + 4 * out_sp + (1 << 20)));
    CK(hipMemset(slab, 3, 4 * in_sp + 4 * out_sp));
    Tabs t{};
    for (int k = 0; k < 4; k++) { t.in[k] = (const uint4 *)(slab + k * in_sp); t.out[k] = (uint4 *)(slab + 4 * in_sp + k * out_sp); }
    printf("{\"tables\": \"inputs %zu bytes apart, outputs %zu\"}\n", in_sp, out_sp);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto time_it = [&](auto &&fn) -> float {
        for (int i = 0; i < 10; i++) fn();
        (void)hipEventRecord(e0);
        for (int i = 0; i < reps; i++) fn();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        return ms * 1000 / reps;
    };
    const double full = 4.0 * (n + half) * 32, ro = 4.0 * n * 32;
    auto rep = [&](const char *name, float us, double bytes) { printf("{\"shape\": \"%s\", \"us\": %.2f, \"GBps\": %.1f}\n", name, us, bytes / us / 1e3); };
    for (int grid : {768, 1536, 4096}) {
        char nm[128];
        snprintf(nm, sizeof nm, "fused round, 4 tables, grid %d", grid); rep(nm, time_it([&] { fused_shape<0, 1><<<grid, 256>>>(t, 4, q); }), full);
        snprintf(nm, sizeof nm, "fused round, 4 tables, grid %d, nontemporal", grid); rep(nm, time_it([&] { fused_shape<1, 1><<<grid, 256>>>(t, 4, q); }), full);
        snprintf(nm, sizeof nm, "fused round, 4 tables, grid %d, loads only", grid); rep(nm, time_it([&] { fused_shape<0, 0><<<grid, 256>>>(t, 4, q); }), ro);
    }
    rep("fused round, 4 tables, grid 1536, whole-line stores", time_it([&] { fused_shape_linestores<0, 0><<<1536, 256>>>(t, 4, q); }), full);
    rep("fused round, 4 tables, grid 1536, whole-line stores and loads", time_it([&] { fused_shape_linestores<0, 1><<<1536, 256>>>(t, 4, q); }), full);
    rep("fused round, 4 tables, grid 1536, whole-line stores, nontemporal", time_it([&] { fused_shape_linestores<1, 0><<<1536, 256>>>(t, 4, q); }), full);
    if (in_sp >= n * 36 && out_sp >= half * 36) {              // the 36-byte kernels touch n * 36 bytes per table: never on tables spaced n * 32 apart
        rep("fused round, 4 tables of 36-byte elements (AoS), grid 1536", time_it([&] { fused_shape_36<<<1536, 256>>>(t, 4, q); }), full * 36 / 32);
        rep("fused round, 4 tables of 36-byte elements (nine word planes), grid 1536", time_it([&] { fused_shape_soa9<<<1536, 256>>>(t, 4, q); }), full * 36 / 32);
    }
    rep("fused round, 1 table at a time (4 launches), grid 4096", time_it([&] { for (int k = 0; k < 4; k++) { Tabs u = t; u.in[0] = t.in[k]; u.out[0] = t.out[k]; fused_shape<0, 1><<<4096, 256>>>(u, 1, q); } }), full);
    rep("wave per table, grid q / 64", time_it([&] { wave_per_table_shape<0><<<(int)(q / 64), 256>>>(t, q); }), full);
    rep("wave per table, grid 4096", time_it([&] { wave_per_table_shape<0><<<4096, 256>>>(t, q); }), full);
    rep("wave per table, grid 4096, nontemporal", time_it([&] { wave_per_table_shape<1><<<4096, 256>>>(t, q); }), full);
    rep("plain fold of each table (4 launches)", time_it([&] { for (int k = 0; k < 4; k++) fold_shape<0><<<(int)(half / 256), 256>>>(t.in[k], t.out[k], half); }), full);
    rep("plain fold of each table (4 launches), nontemporal", time_it([&] { for (int k = 0; k < 4; k++) fold_shape<1><<<(int)(half / 256), 256>>>(t.in[k], t.out[k], half); }), full);
    return 0;
}
