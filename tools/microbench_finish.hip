// microbench_finish.hip -- where the time of one sumcheck_finish_kernel launch goes (csrc/dev_transcript.cuh): the kernel is
// compiled with its TS() hooks writing the 100 MHz wall clock at the stage boundaries: reduce | message | transcript | challenge.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench_finish.hip -o tools/microbench_finish.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
#include <vector>
#define ZK_FINISH_TIMING 1
__device__ unsigned long long g_ts[16];
#define TS(k) do { if (threadIdx.x == 0) g_ts[k] = wall_clock64(); } while (0)
#include "../zk-cryptography-research-implementations_amd/csrc/dev_transcript.cuh"
using namespace zk;
int main(int argc, char **argv) {
    using F = Fr381;
    size_t count = argc > 1 ? atol(argv[1]) : 256; int npts = 3;
    void *part, *buf;
    CK(hipMalloc(&part, 32 * count * 4)); CK(hipMemset(part, 1, 32 * count * 4));
    int threads = (int)((count + 63) / 64 * 64); if (threads > kFinishBlock - 64) threads = kFinishBlock - 64; threads += 64;   // + the helper wave
    CK(hipMalloc(&buf, 4096)); CK(hipMemset(buf, 0, 4096));
    FinishArgs a{};
    a.partials = part; a.count = count; a.ctx.npts = npts; a.ctx.mode = 1; a.with_claim = 0;
    a.ctx.sponge = (DevSponge *)buf; a.ctx.basis = (char *)buf + 256; a.ctx.proof = (char *)buf + 1024; a.flags = kDerive1; a.prev_msg_slot = 8; a.prev_chal_slot = 11; a.msg_slot = 0; a.chal_slot = 3;
    for (int it = 0; it < 5; it++) {
        sumcheck_finish_kernel<F><<<1, threads>>>(a);
        CK(hipDeviceSynchronize());
        unsigned long long ts[16];
        CK(hipMemcpyFromSymbol(ts, HIP_SYMBOL(g_ts), sizeof ts));
        printf("it %d:", it);
        for (int k = 1; k < 5; k++) printf(" %.2f", (double)(ts[k] - ts[k - 1]) / 100.0);
        printf(" us (100 MHz clock)\n");
    }
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0));
    for (int i = 0; i < 200; i++) sumcheck_finish_kernel<F><<<1, threads>>>(a);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("back-to-back: %.2f us per launch\n", ms * 1000 / 200);
    return 0;
}
