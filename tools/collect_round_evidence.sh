set -o pipefail
mkdir -p gpurun_out/r4e
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/p_fold -- python3 $R/bench.py --no-msm --no-config5 --no-configs --no-cpu-baseline > $R/gpurun_out/r4e/bench_fold_profiled.json 2> $R/gpurun_out/r4e/bench_fold_profiled.err
python3 $R/tools/rocprof_summary.py /tmp/p_fold fold0_kernel > $R/gpurun_out/r4e/fold_2p24_kernel_summary.txt 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/p_gkr -- python3 $R/tools/profile_gkr_round.py 22 20 > $R/gpurun_out/r4e/gkr_run.log 2>&1
python3 $R/tools/rocprof_summary.py /tmp/p_gkr fold_round_evals_kernel round_evals_kernel > $R/gpurun_out/r4e/gkr_sumcheck_2p22_kernel_summary.txt 2>&1
python3 $R/tools/rocprof_timeline.py /tmp/p_gkr 11 "round_evals_kernel<zk::Fr381, 2, false>" > $R/gpurun_out/r4e/gkr_sumcheck_2p22_timeline.txt 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d /tmp/p_sp -- python3 $R/tools/bench_gkr_sparse.py 22 3 random > $R/gpurun_out/r4e/sparse_run.log 2>&1
python3 $R/tools/rocprof_summary.py /tmp/p_sp fold_round_evals_kernel > $R/gpurun_out/r4e/gkr_sparse_config4_kernel_summary.txt 2>&1
python3 $R/tools/rocprof_timeline.py /tmp/p_sp 24 "phase1_tables_kernel<zk::Fr381, true>" > $R/gpurun_out/r4e/gkr_sparse_config4_layer_timeline.txt 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/p_b24 -- python3 $R/tools/profile_config5_sumcheck.py 24 > $R/gpurun_out/r4e/basic_run.log 2>&1
python3 $R/tools/rocprof_timeline.py /tmp/p_b24 4 "seg_sums_kernel" > $R/gpurun_out/r4e/basic_sumcheck_2p24_trace.txt 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/p_gd -- python3 $R/tools/profile_gkr_dense.py 8 5 > $R/gpurun_out/r4e/gkr_dense_run.log 2>&1
python3 $R/tools/rocprof_summary.py /tmp/p_gd > $R/gpurun_out/r4e/gkr_reference_shape_depth8_kernel_summary.txt 2>&1
grep -E "depth|verify" $R/gpurun_out/r4e/gkr_dense_run.log >> $R/gpurun_out/r4e/gkr_reference_shape_depth8_kernel_summary.txt
cd $R
timeout -k 10 120 tools/microbench_round.bin 22 100 1536 > gpurun_out/r4e/microbench_round_2p22.jsonl 2>&1
timeout -k 10 120 tools/microbench_round.bin 22 100 1536 3 >> gpurun_out/r4e/microbench_round_2p22.jsonl 2>&1
echo profiles done
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4e/pytest_gpu_full.log 2>&1
tail -n 2 gpurun_out/r4e/pytest_gpu_full.log
timeout -k 10 300 python bench.py > gpurun_out/r4e/bench_default.json 2> gpurun_out/r4e/bench_default.err
tail -c 300 gpurun_out/r4e/bench_default.json
