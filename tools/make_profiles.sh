# Produces the artefacts kept under profiles/: the bench line, rocprofv3 kernel-trace summaries of the same command
# (default lanes and --lanes 1) and the two PMC passes for the L2<->fabric traffic.  Usage (on the GPU box):
#   bash tools/make_profiles.sh <tag>      -> gpurun_out/<tag>/
set -e
tag=${1:-r01}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 -u $R/bench.py > $O/bench.json 2> $O/bench.err
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_default -o p -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-side-configs > $O/trace_default.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_lanes1 -o p -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-side-configs --lanes 1 > $O/trace_lanes1.log 2>&1
timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch -o p -- python3 $R/bench.py --steps 1 --warmup 1 --batch 2 --lanes 1 --no-cpu-baseline --no-side-configs > $O/pmc_fetch.log 2>&1
timeout -s KILL 170 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write -o p -- python3 $R/bench.py --steps 1 --warmup 1 --batch 2 --lanes 1 --no-cpu-baseline --no-side-configs > $O/pmc_write.log 2>&1
python3 $R/tools/pmc_summary.py $O/pmc_fetch srmi > $O/pmc_summary.txt
python3 $R/tools/pmc_summary.py $O/pmc_write srmi >> $O/pmc_summary.txt
# the same two passes for the RCAN and EDSR side configurations (one bench step each, one lane)
for net in rcan edsr; do
  timeout -s KILL 170 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_fetch_$net -o p -- python3 $R/bench.py --model $net --steps 1 --warmup 1 --batch 2 --lanes 1 --no-cpu-baseline --no-side-configs > $O/pmc_fetch_$net.log 2>&1
  timeout -s KILL 170 rocprofv3 --pmc WRITE_SIZE -d $O/pmc_write_$net -o p -- python3 $R/bench.py --model $net --steps 1 --warmup 1 --batch 2 --lanes 1 --no-cpu-baseline --no-side-configs > $O/pmc_write_$net.log 2>&1
  python3 $R/tools/pmc_summary.py $O/pmc_fetch_$net srmi > $O/pmc_summary_$net.txt
  python3 $R/tools/pmc_summary.py $O/pmc_write_$net srmi >> $O/pmc_summary_$net.txt
  rm -rf $O/pmc_fetch_$net $O/pmc_write_$net
done
cp $O/trace_default/*kernel_stats.csv $O/kernel_stats_default.csv
cp $O/trace_lanes1/*kernel_stats.csv $O/kernel_stats_lanes1.csv
grep -h '^{' $O/trace_default.log > $O/bench_under_rocprof_default.json || true
grep -h '^{' $O/trace_lanes1.log > $O/bench_under_rocprof_lanes1.json || true
rm -rf $O/trace_default $O/trace_lanes1 $O/pmc_fetch $O/pmc_write
cat $O/bench.json; cat $O/pmc_summary.txt; cut -c1-150 $O/kernel_stats_default.csv | head -4; cut -c1-150 $O/kernel_stats_lanes1.csv | head -4
