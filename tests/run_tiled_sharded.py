"""Worker of tests/test_sharding_gpu.py: every rank upscales its share of the halo-overlapped tiles of one frame on
the GPU, rank 0 gathers and compares with the single-process tiled result (launched by torch.distributed.run)."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from superresolution_amd import RRDBNet, sharding, synth  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(int(os.environ["LOCAL_RANK"]) % torch.cuda.device_count())
    # "nccl" (= RCCL) with one rank runs the gather of device buffers that a multi-GPU node would run
    backend = sys.argv[1] if len(sys.argv) > 1 else "gloo"
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", torch.cuda.current_device()))
    else:
        dist.init_process_group("gloo")
    model = RRDBNet(num_block=1, compute_dtype="f16")
    model.load_state_dict({k: torch.from_numpy(v) for k, v in synth.rrdbnet_state_dict(num_block=1, seed=3).items()},
                          strict=True)
    eng = model.engine()
    img = torch.from_numpy(synth.smooth_image(150, 200, 9)).cuda()
    fwd = lambda t: eng.forward_u8(t)
    got = sharding.upscale_tiled_sharded(fwd, img, 4, 64, 96, 24, rank, world, always_gather=True)
    ok = True
    if rank == 0:
        want = sharding.upscale_tiled(fwd, img, 4, 64, 96, 24)
        whole = eng.forward_u8(img)
        ok = got is not None and torch.equal(got, want) and torch.equal(got, whole)
        print("tiled-sharded result equals the single-process tiled result and the whole frame:", ok)
    else:
        ok = got is None
    flag = torch.tensor([0 if ok else 1], device="cuda" if backend == "nccl" else "cpu")
    dist.all_reduce(flag)
    dist.destroy_process_group()
    sys.exit(int(flag.item() != 0))


if __name__ == "__main__":
    main()
