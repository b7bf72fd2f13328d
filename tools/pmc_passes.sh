set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc
rm -rf $O; mkdir -p $O
for cfg in "192 32 3" "192 64 3"; do
  tag=$(echo $cfg | tr ' ' '_')
  timeout -s KILL 90 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/sq_$tag -o p -- python3 $R/tools/one_conv.py $cfg > $O/sq_$tag.log 2>&1
  timeout -s KILL 90 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum -d $O/tcc_$tag -o p -- python3 $R/tools/one_conv.py $cfg > $O/tcc_$tag.log 2>&1
  timeout -s KILL 90 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum -d $O/tcp_$tag -o p -- python3 $R/tools/one_conv.py $cfg > $O/tcp_$tag.log 2>&1
  for p in sq tcc tcp; do echo "== $p $cfg"; python3 $R/tools/pmc_summary.py $O/${p}_$tag; done
done
