"""Tile mode across ranks on the GPU: two processes (sharing the one GPU of the test box, gloo for the final gather)
upscale the halo-overlapped tiles of a frame; the stitched result must equal the single-process result bit for bit."""
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


def test_tiled_sharded_two_ranks():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "run_tiled_sharded.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=240)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0
    assert "whole frame: True" in r.stdout
