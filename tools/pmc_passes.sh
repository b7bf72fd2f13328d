# PMC passes (one rocprofv3 run each) for single conv configurations: bash tools/pmc_passes.sh "cin cout rows" ...
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc
rm -rf $O; mkdir -p $O
for cfg in "$@"; do
  tag=$(echo $cfg | tr ' ' '_')
  timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/sq_$tag -o p -- python3 $R/tools/one_conv.py $cfg > $O/sq_$tag.log 2>&1
  timeout -s KILL 90 rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM -d $O/sq2_$tag -o p -- python3 $R/tools/one_conv.py $cfg > $O/sq2_$tag.log 2>&1
  for p in sq sq2; do echo "== $p $cfg"; python3 $R/tools/pmc_summary.py $O/${p}_$tag; done
done
