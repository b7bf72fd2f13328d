# MFMA busy fraction of the conv kernel over bench.py --lanes 1 (one PMC pass)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc_mfma
rm -rf $O; mkdir -p $O
timeout -s KILL 170 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d $O/sq -o p -- python3 $R/bench.py --steps 1 --warmup 1 --batch 2 --lanes 1 --no-cpu-baseline > $O/sq.log 2>&1
python3 $R/tools/pmc_summary.py $O/sq conv3x3
