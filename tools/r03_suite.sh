# One GPU-box session: the GPU suite, the bench line, the MFMA-busy PMC passes and the 8-rank gloo rehearsals.
# Usage: bash tools/r03_suite.sh <tag>   (outputs under gpurun_out/<tag>/)
set -e
tag=${1:-r03a}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$tag
mkdir -p $O
cd $R
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q -rx --timeout 300 --timeout-method thread > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -15 $O/tests.log
timeout -k 10 500 python3 -u bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cat $O/bench.json
bash tools/pmc_mfma_all.sh $tag
cd $R
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 8 --backend gloo --steps 3 --warmup 1 --no-side-configs --no-cpu-baseline > $O/gloo_world8.json 2> $O/gloo_world8.err || { tail -20 $O/gloo_world8.err; exit 1; }
cat $O/gloo_world8.json
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 8 --backend gloo --tiles 32 --steps 3 --warmup 1 --no-side-configs --no-cpu-baseline > $O/gloo_world8_tiles.json 2> $O/gloo_world8_tiles.err || { tail -20 $O/gloo_world8_tiles.err; exit 1; }
cat $O/gloo_world8_tiles.json
