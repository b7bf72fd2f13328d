# HBM-side traffic of bench.py's kernels: two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), each its own run
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc_bench
rm -rf $O; mkdir -p $O
timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o p -- python3 $R/bench.py --steps 1 --warmup 1 --batch 2 --lanes 1 --no-cpu-baseline > $O/fetch.log 2>&1
timeout -s KILL 170 rocprofv3 --pmc WRITE_SIZE -d $O/write -o p -- python3 $R/bench.py --steps 1 --warmup 1 --batch 2 --lanes 1 --no-cpu-baseline > $O/write.log 2>&1
python3 $R/tools/pmc_summary.py $O/fetch srmi
python3 $R/tools/pmc_summary.py $O/write srmi
