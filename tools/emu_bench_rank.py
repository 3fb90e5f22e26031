"""One rank of bench.py's REAL main() (not --dry-run) on the whole-library emulator build: the encoder underneath is
compress_amd/libkcgpu_emu.so, "device" tensors are CPU tensors (tools/emu_torch_shim.py) and the process group is gloo instead of
RCCL — so the N > 1 control flow of the timed loop (contiguous shards, two contexts + three destination buffers, FrameGather of real
frames overlapped with the next step, barrier + MAX timing, dictionary broadcast on C5) executes end to end without a GPU.
TEST INFRASTRUCTURE; launched by tools/emu_bench_flow.sh-style commands:

    KC_LIB_TAG=emu PYTHONPATH=tools python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/emu_bench_rank.py \
        --gpus 2 --gib 0.001953125 --steps 3 --warmup 1 --no-also --no-cpu-baseline
"""
import os
import runpy
import sys

import emu_torch_shim  # noqa: F401
import torch.distributed as dist

_init = dist.init_process_group


def _gloo(backend=None, *a, **k):
    k.pop("device_id", None)
    return _init("gloo", *a, **k)


dist.init_process_group = _gloo
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[1:]
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
