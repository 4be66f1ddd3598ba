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
    CK(hipFree(in));
    CK(hipFree(out));
    return 0;
}
