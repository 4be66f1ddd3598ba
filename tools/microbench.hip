// microbench.hip -- measures the gfx950 instruction rates that bound the field kernels
// (SURVEY.md 7 "Integer throughput vs HBM": must be micro-benchmarked first), the
// register-resident Montgomery-multiply ceiling, and the AoS streaming bandwidth of the fold's
// access pattern.  Prints one JSON object per line.  Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <vector>

#include "../zk-cryptography-research-implementations_amd/csrc/mle_kernels.cuh"
#include "../zk-cryptography-research-implementations_amd/csrc/ufield.cuh"

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITERS = 4096;
constexpr int CHAINS = 8;

// one kernel per instruction: 8 independent dependency chains per lane, ITERS x 8 instructions
#define RATE_KERNEL(name, T, AT, INITACC, INITA, INITB, ASM, CLOB)                     \
    __global__ void name(uint32_t *out, uint32_t seed) {                               \
        T acc[CHAINS];                                                                 \
        AT a = (AT)(INITA), b = (AT)(INITB);                                           \
        for (int k = 0; k < CHAINS; k++) acc[k] = (T)(INITACC);                        \
        for (int it = 0; it < ITERS; it++) {                                           \
            _Pragma("unroll") for (int k = 0; k < CHAINS; k++)                         \
                asm volatile(ASM : "+v"(acc[k]) : "v"(a), "v"(b) : CLOB);              \
        }                                                                              \
        T r = acc[0];                                                                  \
        for (int k = 1; k < CHAINS; k++) r += acc[k];                                  \
        if (r == (T)123456789) out[threadIdx.x] = 1;                                   \
    }

RATE_KERNEL(k_add_u32, uint32_t, uint32_t, k + seed + threadIdx.x, seed | 1u, 3u, "v_add_u32 %0, %0, %1", "memory")
RATE_KERNEL(k_addc_u32, uint32_t, uint32_t, k + seed + threadIdx.x, seed | 1u, 3u, "v_addc_co_u32 %0, vcc, %0, %1, vcc", "vcc")
RATE_KERNEL(k_lshl_add_u64, uint64_t, uint64_t, k + seed + threadIdx.x, seed | 1u, 3u, "v_lshl_add_u64 %0, %0, 0, %1", "memory")
RATE_KERNEL(k_mad_u64_u32, uint64_t, uint32_t, k + seed, seed | 1u, threadIdx.x * 2654435761u + 7u, "v_mad_u64_u32 %0, vcc, %1, %2, %0", "vcc")
RATE_KERNEL(k_mul_lo_u32, uint32_t, uint32_t, k + seed + threadIdx.x, seed | 1u, 3u, "v_mul_lo_u32 %0, %0, %1", "memory")
RATE_KERNEL(k_mul_hi_u32, uint32_t, uint32_t, ~(k + seed + threadIdx.x), seed | 0x80000001u, 3u, "v_mul_hi_u32 %0, %0, %1", "memory")
RATE_KERNEL(k_mad_u32_u24, uint32_t, uint32_t, k + seed, seed | 1u, threadIdx.x + 3u, "v_mad_u32_u24 %0, %1, %2, %0", "memory")
RATE_KERNEL(k_mul_hi_u32_u24, uint32_t, uint32_t, k + seed + threadIdx.x, seed | 1u, 3u, "v_mul_hi_u32_u24 %0, %0, %1", "memory")
RATE_KERNEL(k_fma_f32, float, float, k + 0.5f, 1.0000001f, 1e-9f * threadIdx.x, "v_fma_f32 %0, %0, %1, %2", "memory")
RATE_KERNEL(k_fma_f64, double, double, k + 0.5, 1.0000001, 1e-9 * threadIdx.x, "v_fma_f64 %0, %0, %1, %2", "memory")
RATE_KERNEL(k_mul_f64, double, double, k + 0.5, 1.0000001, 1.0, "v_mul_f64 %0, %0, %1", "memory")
RATE_KERNEL(k_add_f64, double, double, k + 0.5, 1.0000001, 1.0, "v_add_f64 %0, %0, %1", "memory")

// register-resident Montgomery chain: x <- x*y, MULS multiplications per thread
template <class F, int MULS> __global__ void k_mont_chain(void *out, uint64_t seed) {
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    zk::Fe<F> x = zk::random_element<F>(seed, gid), y = zk::random_element<F>(seed + 1, gid);
    for (int i = 0; i < MULS; i++) x = zk::fe_mul<F>(x, y);
    zk::fe_store<F>(out, gid, x);
}
// unsaturated 29-bit-limb chain: x <- x*y in the internal form
template <class F, int MULS> __global__ void k_umul_chain(void *out, uint64_t seed) {
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    zk::Ufe<F> x = zk::u_from_limbs32<F>(zk::random_element<F>(seed, gid)), y = zk::u_from_limbs32<F>(zk::random_element<F>(seed + 1, gid));
    for (int i = 0; i < MULS; i++) x = zk::umul<F>(x, y);
    zk::fe_store<F>(out, gid, zk::u_to_limbs32<F>(x));
}
// drop-in product through the unsaturated scan (conversions included)
template <class F, int MULS> __global__ void k_mulu_chain(void *out, uint64_t seed) {
    size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    zk::Fe<F> x = zk::random_element<F>(seed, gid), y = zk::random_element<F>(seed + 1, gid);
    for (int i = 0; i < MULS; i++) x = zk::fe_mul_u<F>(x, y);
    zk::fe_store<F>(out, gid, x);
}
template <class F> __global__ void k_fold_u(const void *__restrict__ in, void *__restrict__ out, size_t half, zk::Ufe<F> ru) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    zk::Fe<F> lo = zk::fe_load<F>(in, i), hi = zk::fe_load<F>(in, i + half);
    zk::fe_store<F>(out, i, zk::fe_add<F>(lo, zk::fe_mul_u_pre<F>(ru, zk::fe_sub<F>(hi, lo))));
}
template <class F> __device__ __forceinline__ zk::Fe<F> nt_load(const void *base, size_t idx);
template <class F> __device__ __forceinline__ void nt_store(void *base, size_t idx, const zk::Fe<F> &a) {
    uint32_t *p = reinterpret_cast<uint32_t *>(base) + idx * F::N;
#pragma unroll
    for (int k = 0; k < F::N; k++) __builtin_nontemporal_store(a.l[k], p + k);
}
template <class F, int EPT, bool NTS, bool NTL> __global__ void k_fold_u2(const void *__restrict__ in, void *__restrict__ out, size_t half, zk::Ufe<F> ru) {
    size_t chunk = half / EPT;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= chunk) return;
    zk::Fe<F> lo[EPT], hi[EPT];
#pragma unroll
    for (int e = 0; e < EPT; e++) {
        lo[e] = NTL ? nt_load<F>(in, i + e * chunk) : zk::fe_load<F>(in, i + e * chunk);
        hi[e] = NTL ? nt_load<F>(in, i + e * chunk + half) : zk::fe_load<F>(in, i + e * chunk + half);
    }
#pragma unroll
    for (int e = 0; e < EPT; e++) {
        zk::Fe<F> o = zk::fe_add<F>(lo[e], zk::fe_mul_u_pre<F>(ru, zk::fe_sub<F>(hi[e], lo[e])));
        if (NTS) nt_store<F>(out, i + e * chunk, o); else zk::fe_store<F>(out, i + e * chunk, o);
    }
}
// the fold's memory pattern with trivial arithmetic (2 streams in, 1 out, 32 B per lane)
template <class F> __global__ void k_stream3(const void *in, void *out, size_t half) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < half; i += stride) {
        zk::Fe<F> a = zk::fe_load<F>(in, i), b = zk::fe_load<F>(in, i + half);
#pragma unroll
        for (int k = 0; k < F::N; k++) a.l[k] ^= b.l[k];
        zk::fe_store<F>(out, i, a);
    }
}
// the same three streams with 16 B per lane, fully coalesced (n16 = output length in uint4): is the 32-byte-per-lane pattern what the fold pays for?
__global__ void k_stream3_16(const uint4 *in, uint4 *out, size_t n16) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        uint4 a = in[i], b = in[i + n16];
        a.x ^= b.x; a.y ^= b.y; a.z ^= b.z; a.w ^= b.w;
        out[i] = a;
    }
}
__global__ void k_copy16(const uint4 *in, uint4 *out, size_t n) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = in[i];
}


// ---- fold variants (same arithmetic, different launch / memory shapes) ---------------------------
template <class F> __device__ __forceinline__ zk::Fe<F> nt_load(const void *base, size_t idx) {
    const uint4 *p = reinterpret_cast<const uint4 *>(base) + idx * (F::N / 4);
    zk::Fe<F> r;
#pragma unroll
    for (int k = 0; k < F::N / 4; k++) {
        uint4 v;
        v.x = __builtin_nontemporal_load(&p[k].x); v.y = __builtin_nontemporal_load(&p[k].y);
        v.z = __builtin_nontemporal_load(&p[k].z); v.w = __builtin_nontemporal_load(&p[k].w);
        r.l[4 * k + 0] = v.x; r.l[4 * k + 1] = v.y; r.l[4 * k + 2] = v.z; r.l[4 * k + 3] = v.w;
    }
    return r;
}
template <class F, int EPT, bool NT> __global__ void k_fold_v(const void *__restrict__ in, void *__restrict__ out, size_t half, zk::Fe<F> r) {
    // EPT elements per thread, strided by half/EPT so every access stays coalesced
    size_t chunk = half / EPT;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= chunk) return;
    zk::Fe<F> lo[EPT], hi[EPT];
#pragma unroll
    for (int e = 0; e < EPT; e++) {
        lo[e] = NT ? nt_load<F>(in, i + e * chunk) : zk::fe_load<F>(in, i + e * chunk);
        hi[e] = NT ? nt_load<F>(in, i + e * chunk + half) : zk::fe_load<F>(in, i + e * chunk + half);
    }
#pragma unroll
    for (int e = 0; e < EPT; e++)
        zk::fe_store<F>(out, i + e * chunk, zk::fe_add<F>(lo[e], zk::fe_mul<F>(r, zk::fe_sub<F>(hi[e], lo[e]))));
}

template <class Fn> static float time_ms(Fn fn, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    fn();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; i++) fn();
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char **argv) {
    int dev = 0;
    CK(hipSetDevice(dev));
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, dev));
    int cus = prop.multiProcessorCount;
    double clk = prop.clockRate * 1e3;   // Hz
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %.0f}\n", prop.name, cus, clk / 1e6);
    uint32_t *dout;
    CK(hipMalloc(&dout, 4096));
    const int blocks = cus * 8, threads = 256;
    double lanes = (double)blocks * threads;
#define RUN_RATE(name)                                                                         \
    {                                                                                          \
        float ms = time_ms([&] { name<<<blocks, threads>>>(dout, 12345u); }, 5);               \
        double ops = lanes * ITERS * CHAINS;                                                   \
        double rate = ops / (ms * 1e-3);                                                       \
        /* cycles per wave64 instruction per SIMD at the nominal clock */                      \
        double cyc = (clk * cus * 4) / (rate / 64.0);                                          \
        printf("{\"instr\": \"%s\", \"ms\": %.4f, \"lane_ops_per_s\": %.4e, \"cycles_per_wave_instr_per_simd\": %.2f}\n", #name, ms, rate, cyc); \
        fflush(stdout);                                                                        \
    }
    if (argc > 1 && !strcmp(argv[1], "--only-mad")) {        // bench.py: the MSM roofline's peak, re-measured in the run
        for (int warm = 0; warm < 3; warm++) RUN_RATE(k_mad_u64_u32)
        return 0;
    }
    RUN_RATE(k_add_u32)
    RUN_RATE(k_addc_u32)
    RUN_RATE(k_lshl_add_u64)
    RUN_RATE(k_mad_u64_u32)
    RUN_RATE(k_mul_lo_u32)
    RUN_RATE(k_mul_hi_u32)
    RUN_RATE(k_mad_u32_u24)
    RUN_RATE(k_mul_hi_u32_u24)
    RUN_RATE(k_fma_f32)
    RUN_RATE(k_fma_f64)
    RUN_RATE(k_mul_f64)
    RUN_RATE(k_add_f64)

    // Montgomery chain
    {
        void *buf;
        CK(hipMalloc(&buf, (size_t)blocks * threads * 48));
        constexpr int MULS = 512;
        float ms = time_ms([&] { k_mont_chain<zk::Fr381, MULS><<<blocks, threads>>>(buf, 7); }, 5);
        printf("{\"kernel\": \"mont_chain_fr381\", \"ms\": %.4f, \"field_mul_per_s\": %.4e}\n", ms, lanes * MULS / (ms * 1e-3));
        ms = time_ms([&] { k_mont_chain<zk::Fq381, MULS><<<blocks, threads>>>(buf, 7); }, 5);
        printf("{\"kernel\": \"mont_chain_fq381\", \"ms\": %.4f, \"field_mul_per_s\": %.4e}\n", ms, lanes * MULS / (ms * 1e-3));
        ms = time_ms([&] { k_umul_chain<zk::Fr381, MULS><<<blocks, threads>>>(buf, 7); }, 5);
        printf("{\"kernel\": \"umul_chain_fr381\", \"ms\": %.4f, \"field_mul_per_s\": %.4e}\n", ms, lanes * MULS / (ms * 1e-3));
        ms = time_ms([&] { k_umul_chain<zk::Fq381, MULS><<<blocks, threads>>>(buf, 7); }, 5);
        printf("{\"kernel\": \"umul_chain_fq381\", \"ms\": %.4f, \"field_mul_per_s\": %.4e}\n", ms, lanes * MULS / (ms * 1e-3));
        ms = time_ms([&] { k_mulu_chain<zk::Fr381, MULS><<<blocks, threads>>>(buf, 7); }, 5);
        printf("{\"kernel\": \"fe_mul_u_chain_fr381\", \"ms\": %.4f, \"field_mul_per_s\": %.4e}\n", ms, lanes * MULS / (ms * 1e-3));
        CK(hipFree(buf));
    }
    // streaming: 2^24-element Fr table (512 MiB) -> 2^23
    {
        size_t n = (size_t)1 << 24, half = n / 2;
        void *in, *out;
        CK(hipMalloc(&in, n * 32));
        CK(hipMalloc(&out, half * 32));
        zk::fill_random_kernel<zk::Fr381><<<2048, 256>>>(in, n, 99, 0);
        CK(hipDeviceSynchronize());
        for (int grid : {1024, 2048, 4096, 8192, 32768}) {
            float ms = time_ms([&] { k_stream3<zk::Fr381><<<grid, 256>>>(in, out, half); }, 10);
            printf("{\"kernel\": \"stream3_aos32\", \"grid\": %d, \"ms\": %.4f, \"GBps\": %.1f}\n", grid, ms, 96.0 * half / (ms * 1e-3) / 1e9);
            zk::Fe<zk::Fr381> r = zk::random_element<zk::Fr381>(5, 5);
            ms = time_ms([&] { zk::fold_kernel<zk::Fr381><<<grid, 256>>>(in, out, half, 23, r); }, 10);
            printf("{\"kernel\": \"fold_fr381_2p24\", \"grid\": %d, \"ms\": %.4f, \"GBps\": %.1f, \"field_mul_per_s\": %.4e}\n", grid, ms,
                   96.0 * half / (ms * 1e-3) / 1e9, half / (ms * 1e-3));
            fflush(stdout);
        }

        {
            zk::Fe<zk::Fr381> r = zk::random_element<zk::Fr381>(5, 5);
#define RUN_V(EPT, NT, BS)                                                                                  \
            {                                                                                               \
                int grid = (int)((half / EPT + BS - 1) / BS);                                               \
                float ms = time_ms([&] { k_fold_v<zk::Fr381, EPT, NT><<<grid, BS>>>(in, out, half, r); }, 20); \
                printf("{\"kernel\": \"fold_v\", \"ept\": %d, \"nt\": %d, \"block\": %d, \"grid\": %d, \"ms\": %.4f, \"GBps\": %.1f}\n", EPT, (int)NT, BS, grid, ms, 96.0 * half / (ms * 1e-3) / 1e9); \
                fflush(stdout);                                                                             \
            }
            {
                zk::Ufe<zk::Fr381> ru = zk::u_from_limbs32<zk::Fr381>(r);
                int grid = (int)((half + 255) / 256);
                float ms = time_ms([&] { k_fold_u<zk::Fr381><<<grid, 256>>>(in, out, half, ru); }, 20);
                printf("{\"kernel\": \"fold_u29\", \"grid\": %d, \"ms\": %.4f, \"GBps\": %.1f}\n", grid, ms, 96.0 * half / (ms * 1e-3) / 1e9);
            }
            {
                zk::Ufe<zk::Fr381> ru = zk::u_from_limbs32<zk::Fr381>(r);
#define RUN_U2(EPT, NTS, NTL, BS)                                                                                 \
                {                                                                                                 \
                    int grid = (int)((half / EPT + BS - 1) / BS);                                                 \
                    float ms = time_ms([&] { k_fold_u2<zk::Fr381, EPT, NTS, NTL><<<grid, BS>>>(in, out, half, ru); }, 30); \
                    printf("{\"kernel\": \"fold_u2\", \"ept\": %d, \"nt_store\": %d, \"nt_load\": %d, \"block\": %d, \"ms\": %.4f, \"GBps\": %.1f}\n", EPT, (int)NTS, (int)NTL, BS, ms, 96.0 * half / (ms * 1e-3) / 1e9); \
                    fflush(stdout);                                                                               \
                }
                RUN_U2(1, false, false, 256) RUN_U2(1, true, false, 256) RUN_U2(1, true, true, 256) RUN_U2(1, false, true, 256)
                RUN_U2(2, false, false, 256) RUN_U2(2, true, false, 256) RUN_U2(1, false, false, 512) RUN_U2(1, false, false, 1024)
                RUN_U2(1, false, false, 128) RUN_U2(4, false, false, 256) RUN_U2(2, true, true, 512)
            }
            RUN_V(1, false, 256) RUN_V(1, false, 512) RUN_V(1, false, 1024) RUN_V(1, false, 128) RUN_V(1, false, 64)
            RUN_V(2, false, 256) RUN_V(4, false, 256) RUN_V(2, false, 128) RUN_V(2, false, 512)
            RUN_V(1, true, 256) RUN_V(2, true, 256)
        }
        for (int grid : {2048, 8192, 65536}) {
            float ms3 = time_ms([&] { k_stream3_16<<<grid, 256>>>((const uint4 *)in, (uint4 *)out, half * 2); }, 10);
            printf("{\"kernel\": \"stream3_coalesced16\", \"grid\": %d, \"ms\": %.4f, \"GBps\": %.1f}\n", grid, ms3, 96.0 * half / (ms3 * 1e-3) / 1e9);
            float ms3a = time_ms([&] { k_stream3<zk::Fr381><<<grid, 256>>>(in, out, half); }, 10);
            printf("{\"kernel\": \"stream3_aos32\", \"grid\": %d, \"ms\": %.4f, \"GBps\": %.1f}\n", grid, ms3a, 96.0 * half / (ms3a * 1e-3) / 1e9);
        }
        float ms = time_ms([&] { k_copy16<<<2048, 256>>>((const uint4 *)in, (uint4 *)out, half * 2); }, 10);
        printf("{\"kernel\": \"copy16\", \"ms\": %.4f, \"GBps\": %.1f}\n", ms, 2.0 * 32 * half / (ms * 1e-3) / 1e9);
        CK(hipFree(in));
        CK(hipFree(out));
    }
    return 0;
}
