# rocprofv3 kernel-trace summary of the HBM-bound kernels at two frame sizes -> gpurun_out/<tag>/aux_report.md
set -e
tag=${1:-r03g}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
: > $O/aux_report.md
for size in 510x339 1920x1080; do
  export SRMI_AUX_SIZE=$size
  rm -rf $O/prof_aux_$size
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_aux_$size -o p -- python3 $R/tools/bench_aux.py > $O/prof_aux_$size.log 2>&1
  echo "## frame $size, 64 channels, f16 (rocprofv3 --kernel-trace --stats -- python3 tools/bench_aux.py)" >> $O/aux_report.md
  python3 $R/tools/bench_aux.py --report $(ls $O/prof_aux_$size/*kernel_stats.csv | head -1) >> $O/aux_report.md
  echo >> $O/aux_report.md
  rm -rf $O/prof_aux_$size
done
cat $O/aux_report.md
