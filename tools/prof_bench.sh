# rocprofv3 kernel trace of bench.py (args: output tag, extra bench flags...) -> gpurun_out/prof_<tag>/
set -e
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > $R/gpurun_out/prof_$tag.log 2>&1
find $R/gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1 | xargs cat | cut -c1-160
tail -1 $R/gpurun_out/prof_$tag.log | cut -c1-200
