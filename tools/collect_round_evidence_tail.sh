set -o pipefail
mkdir -p gpurun_out/r4e
R=$PWD
cd /tmp && export TMPDIR=/tmp
ZK_HOST_TRANSCRIPT=0 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d /tmp/p_gkr_dev -- python3 $R/tools/profile_gkr_round.py 22 10 > $R/gpurun_out/r4e/gkr_dev_run.log 2>&1
python3 $R/tools/rocprof_timeline.py /tmp/p_gkr_dev 30 "round_evals_kernel<zk::Fr381, 2, false>" > $R/gpurun_out/r4e/gkr_sumcheck_2p22_timeline_device_transcript.txt 2>&1
cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r4e/pytest_gpu_full.log 2>&1
tail -n 2 gpurun_out/r4e/pytest_gpu_full.log
timeout -k 10 300 python bench.py > gpurun_out/r4e/bench_default.json 2> gpurun_out/r4e/bench_default.err
tail -c 300 gpurun_out/r4e/bench_default.json
