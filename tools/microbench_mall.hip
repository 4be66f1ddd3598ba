// microbench_mall.hip -- does the ORDER in which a launch walks a table matter to the launch behind it?  The 256 MiB Infinity Cache keeps
// about the last 256 MiB a launch touched; a 2^24-entry table is 512 MiB.  A launch that walks the table in the same direction as the one
// before it finds none of it; one that walks it backwards starts in what the other left behind.  Shapes: the fold (element i and i + half
// in, element i out), read-only (the first round's sums), both with no arithmetic but an XOR per word.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench_mall.hip -o tools/microbench_mall.bin ; tools/microbench_mall.bin [log_n = 24] [reps = 40]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__device__ __forceinline__ uint4 x4(uint4 a, uint4 b) { return make_uint4(a.x ^ b.x, a.y ^ b.y, a.z ^ b.z, a.w ^ b.w); }
// one 32-byte element = two uint4; one element pair per lane, one launch-wide pass
template <int STORE> __global__ void __launch_bounds__(256) fold_shape(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t half, int rev) {
    const size_t b = rev ? (size_t)gridDim.x - 1 - blockIdx.x : blockIdx.x;
    const size_t i = b * blockDim.x + threadIdx.x;
    if (i >= half) return;
    const uint4 al = in[2 * i], ah = in[2 * i + 1], bl = in[2 * (i + half)], bh = in[2 * (i + half) + 1];
    if (STORE) { out[2 * i] = x4(al, bl); out[2 * i + 1] = x4(ah, bh); }
    else { const uint4 v = x4(x4(al, bl), x4(ah, bh)); if (v.x == 0x12345678u && v.y == 0x9abcdef0u) out[0] = v; }
}
// the fold's shape with non-temporal stores (NT & 1) and loads (NT & 2): every byte is touched once per launch
typedef uint32_t v4u __attribute__((ext_vector_type(4)));
template <int NT> __device__ __forceinline__ uint4 ldn(const uint4 *p) {
    if (!(NT & 2)) return *p;
    const v4u v = __builtin_nontemporal_load(reinterpret_cast<const v4u *>(p));
    return make_uint4(v.x, v.y, v.z, v.w);
}
template <int NT> __device__ __forceinline__ void stn(uint4 *p, uint4 v) {
    if (NT & 1) { v4u w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, reinterpret_cast<v4u *>(p)); } else *p = v;
}
template <int NT> __global__ void __launch_bounds__(256) fold_shape_nt(const uint4 *__restrict__ in, uint4 *__restrict__ out, size_t half) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= half) return;
    const uint4 al = ldn<NT>(in + 2 * i), ah = ldn<NT>(in + 2 * i + 1), bl = ldn<NT>(in + 2 * (i + half)), bh = ldn<NT>(in + 2 * (i + half) + 1);
    stn<NT>(out + 2 * i, x4(al, bl));
    stn<NT>(out + 2 * i + 1, x4(ah, bh));
}
template <int NT> static float time_nt(const uint4 *in, uint4 *out, size_t half, unsigned grid, int reps, hipStream_t s, hipEvent_t e0, hipEvent_t e1) {
    for (int w = 0; w < 20; w++) hipLaunchKernelGGL(fold_shape_nt<NT>, dim3(grid), dim3(256), 0, s, in, out, half);
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int w = 0; w < reps; w++) hipLaunchKernelGGL(fold_shape_nt<NT>, dim3(grid), dim3(256), 0, s, in, out, half);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}
int main(int argc, char **argv) {
    const int log_n = argc > 1 ? atoi(argv[1]) : 24, reps = argc > 2 ? atoi(argv[2]) : 40;
    if (log_n < 12 || log_n > 26) { fprintf(stderr, "log_n in 12 .. 26\n"); return 2; }
    const size_t n = (size_t)1 << log_n, half = n / 2;
    uint4 *in = nullptr, *out = nullptr;
    CK(hipMalloc(&in, n * 32));
    CK(hipMalloc(&out, half * 32));
    CK(hipMemset(in, 0x5a, n * 32));
    CK(hipMemset(out, 0, half * 32));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)((half + 255) / 256);
    for (int store = 0; store < 2; store++)
        for (int mode = 0; mode < 2; mode++) {                      // 0: every launch forwards; 1: directions alternate
            for (int w = 0; w < 20; w++) {
                const int rev = mode ? (w & 1) : 0;
                if (store) hipLaunchKernelGGL(fold_shape<1>, dim3(grid), dim3(256), 0, s, in, out, half, rev);
                else hipLaunchKernelGGL(fold_shape<0>, dim3(grid), dim3(256), 0, s, in, out, half, rev);
            }
            CK(hipStreamSynchronize(s));
            CK(hipEventRecord(e0, s));
            for (int w = 0; w < reps; w++) {
                const int rev = mode ? (w & 1) : 0;
                if (store) hipLaunchKernelGGL(fold_shape<1>, dim3(grid), dim3(256), 0, s, in, out, half, rev);
                else hipLaunchKernelGGL(fold_shape<0>, dim3(grid), dim3(256), 0, s, in, out, half, rev);
            }
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const double us = ms * 1e3 / reps, bytes = (double)n * 32 + (store ? (double)half * 32 : 0);
            printf("{\"log_n\": %d, \"shape\": \"%s\", \"order\": \"%s\", \"us\": %.2f, \"TBps\": %.3f}\n", log_n, store ? "fold (2 reads + 1 write)" : "read only",
                   mode ? "alternating" : "forwards", us, bytes / us * 1e-6);
        }
    for (int rep = 0; rep < 2; rep++) {
        const float t[4] = {time_nt<0>(in, out, half, grid, reps, s, e0, e1), time_nt<1>(in, out, half, grid, reps, s, e0, e1),
                            time_nt<2>(in, out, half, grid, reps, s, e0, e1), time_nt<3>(in, out, half, grid, reps, s, e0, e1)};
        const char *nm[4] = {"plain", "nt stores", "nt loads", "nt loads + stores"};
        for (int k = 0; k < 4; k++)
            printf("{\"log_n\": %d, \"shape\": \"fold (2 reads + 1 write), forwards\", \"policy\": \"%s\", \"us\": %.2f, \"TBps\": %.3f}\n", log_n, nm[k], t[k],
                   ((double)n * 32 + (double)half * 32) / t[k] * 1e-6);
    }
    CK(hipFree(in));
    CK(hipFree(out));
    return 0;
}
