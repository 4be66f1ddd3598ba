// microbench_host_step.hip -- what the HOST's share of an exchange costs (no GPU needed): the basic sumcheck's m rounds on 2^m segment sums
// (zkmle_sumcheck.hip serve_multi) and one GKR round's transcript step (serve_round), timed in a loop on this machine's CPU.
//   hipcc -O3 -std=c++17 tools/microbench_host_step.hip -o tools/microbench_host_step.bin ; tools/microbench_host_step.bin
#include <chrono>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "../zk-cryptography-research-implementations_amd/csrc/ufield.cuh"
#include "../zk-cryptography-research-implementations_amd/csrc/transcript.h"
using namespace zk;
using F = Fr381;
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    Transcript tr;
    std::vector<Fe<F>> S0(256);
    for (int i = 0; i < 256; i++) { uint8_t b[32]; memset(b, i + 1, 32); tr.append(b, 32); S0[i] = tr.random_challenge_as_field_element<F>(); }
    for (int m : {6, 7, 8}) {
        const int reps = 2000;
        double t_sum = 0, t_hash = 0, t_fold = 0;
        Fe<F> sink = fe_zero<F>();
        const double t0 = now_us();
        for (int rep = 0; rep < reps; rep++) {
            Fe<F> S[256];
            size_t n = (size_t)1 << m;
            for (size_t t = 0; t < n; t++) S[t] = S0[t];
            for (int i = 0; i < m; i++) {
                const size_t half = n / 2;
                double a = now_us();
                Fe<F> a0 = S[0], a1 = S[half];
                for (size_t j = 1; j < half; j++) { a0 = fe_add<F>(a0, S[j]); a1 = fe_add<F>(a1, S[half + j]); }
                double b = now_us();
                tr.append_be<F>(a0);
                tr.append_be<F>(a1);
                const Fe<F> r = tr.random_challenge_as_field_element<F>();
                double c = now_us();
                for (size_t j = 0; j < half; j++) S[j] = fe_add<F>(S[j], fe_mul<F>(r, fe_sub<F>(S[half + j], S[j])));
                double d = now_us();
                t_sum += b - a; t_hash += c - b; t_fold += d - c;
                n = half;
            }
            sink = fe_add<F>(sink, S[0]);
        }
        const double tot = (now_us() - t0) / reps;
        printf("{\"what\": \"serve_multi host work\", \"m\": %d, \"us_per_exchange\": %.2f, \"us_sums\": %.2f, \"us_hash\": %.2f, \"us_fold\": %.2f, \"sink\": %u}\n", m, tot, t_sum / reps,
               t_hash / reps, t_fold / reps, sink.l[0]);
    }
    {   // one fe_mul, one Keccak step (append 64 B + challenge)
        const int reps = 200000;
        Fe<F> x = S0[1], y = S0[2];
        double t0 = now_us();
        for (int i = 0; i < reps; i++) x = fe_mul<F>(x, y);
        const double mul = (now_us() - t0) / reps;
        t0 = now_us();
        for (int i = 0; i < reps; i++) { tr.append_be<F>(x); tr.append_be<F>(y); x = tr.random_challenge_as_field_element<F>(); }
        const double step = (now_us() - t0) / reps;
        printf("{\"what\": \"host primitives\", \"us_fe_mul\": %.4f, \"us_two_appends_and_challenge\": %.3f, \"sink\": %u}\n", mul, step, x.l[0]);
    }
    return 0;
}
