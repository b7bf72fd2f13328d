# MFMA-pipe utilisation of the conv kernels over one bench step of each network on the current library (one PMC
# pass per network; the program itself directly behind `--`).  Usage on the GPU box: bash tools/pmc_mfma_all.sh <tag>
#   -> gpurun_out/<tag>/mfma_busy_{rrdbnet,edsr,rcan}.txt
set -e
tag=${1:-r03}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for net in rrdbnet edsr rcan; do
  rm -rf $O/pmc_mfma_$net
  timeout -s KILL 170 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/pmc_mfma_$net -o p -- python3 $R/bench.py --model $net --steps 1 --warmup 1 --batch 2 --lanes 1 --no-cpu-baseline --no-side-configs > $O/pmc_mfma_$net.log 2>&1
  python3 $R/tools/pmc_mfma_report.py $O/pmc_mfma_$net $net > $O/mfma_busy_$net.txt
  cat $O/mfma_busy_$net.txt
  rm -rf $O/pmc_mfma_$net
done
