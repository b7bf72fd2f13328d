"""Tile mode across ranks on the GPU: processes launched by torch.distributed.run upscale the halo-overlapped tiles
of a frame and gather the u8 pieces; the stitched result must equal the single-process result bit for bit."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("ranks,backend", [(2, "gloo"), (1, "nccl")], ids=["two_ranks_gloo", "one_rank_rccl"])
def test_tiled_sharded(ranks, backend):
    """gloo: two ranks share the GPU, pieces staged through the host.  nccl: one rank, so that the RCCL gather
    of device buffers (the branch a multi-GPU node takes) runs on the one-GPU test box."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(ranks), "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "run_tiled_sharded.py"), backend]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0
    assert "whole frame: True" in r.stdout
