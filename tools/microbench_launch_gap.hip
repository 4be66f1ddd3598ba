// microbench_launch_gap.hip -- how long the GPU sits between two dependent kernels of one stream: every kernel stamps the 100 MHz wall clock when its
// first workgroup starts and when its last workgroup ends; gap = start[k + 1] - end[k].  (rocprofv3's kernel trace shows back-to-back kernels as
// contiguous: the boundary is inside the durations it reports.)  Grid sizes as the provers launch: 1536 x 256 (a fused round) and 64 x 256 (a short round).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_launch_gap.hip -o tools/microbench_launch_gap.bin
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void stamp_kernel(unsigned long long *ts, unsigned *counter, int k, int spin) {
    if (blockIdx.x == 0 && threadIdx.x == 0) ts[2 * k] = wall_clock64();
    unsigned long long t0 = wall_clock64();
    while ((long long)(wall_clock64() - t0) < spin) { }                      // `spin` ticks of 10 ns of "work" per workgroup
    __syncthreads();
    if (threadIdx.x == 0) {
        if (atomicAdd(counter + k, 1u) == gridDim.x - 1) ts[2 * k + 1] = wall_clock64();
    }
}
int main() {
    const int N = 64;
    unsigned long long *ts; unsigned *cnt;
    CK(hipMalloc(&ts, 2 * N * 8)); CK(hipMalloc(&cnt, N * 4));
    for (int grid : {1536, 64, 1}) for (int spin : {100, 1000}) {
        CK(hipMemset(cnt, 0, N * 4)); CK(hipMemset(ts, 0, 2 * N * 8));
        for (int k = 0; k < N; k++) stamp_kernel<<<grid, 256>>>(ts, cnt, k, spin);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(2 * N);
        CK(hipMemcpy(h.data(), ts, 2 * N * 8, hipMemcpyDeviceToHost));
        std::vector<double> gap, dur;
        for (int k = 8; k + 1 < N; k++) { gap.push_back((double)(h[2 * (k + 1)] - h[2 * k + 1]) * 0.01); dur.push_back((double)(h[2 * k + 1] - h[2 * k]) * 0.01); }
        std::sort(gap.begin(), gap.end()); std::sort(dur.begin(), dur.end());
        printf("{\"grid\": %d, \"work_us_per_workgroup\": %.0f, \"kernel_us_median\": %.2f, \"gap_us_median\": %.2f, \"gap_us_min\": %.2f, \"gap_us_p90\": %.2f}\n", grid, spin * 0.01,
               dur[dur.size() / 2], gap[gap.size() / 2], gap[0], gap[gap.size() * 9 / 10]);
    }
    return 0;
}
