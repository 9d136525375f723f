"""RCCL bring-up on the one GPU of the test box: a one-rank "nccl" process group carries the trainer's collectives (bucketed
all-reduce(AVG) launched from the gradient hooks, parameter / optimiser-state broadcasts, bench.py's MAX / all_gather) on the
device.  N > 1 arithmetic is covered by the world-2 gloo tests (tests/test_distributed_gloo.py); this covers what they cannot:
that the calls this package makes are ones RCCL accepts on an MI355X (AVG, async handles on arena slices, device_id binding)."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["DIQT_ROOT"])
import torch
import torch.distributed as dist
from diffusioniqt_amd import distributed as D
from diffusioniqt_amd.imagen_pytorch3D import SRUnet256

world, rank, device = D.init_from_env()
assert (world, rank) == (1, 0) and device.type == "cuda"
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=device)      # what init_from_env does for WORLD_SIZE > 1
assert dist.get_backend() == "nccl"

torch.manual_seed(0)
unet = SRUnet256(img_size=8, dim=16, init_dim=16, dim_mults=(1, 2), channels=1, num_resnet_blocks=(1, 1), init_conv_kernel_size=3,
                 lowres_cond=True, init_cross_embed=False, att_type='linear', attn_dim_head=16, attend_at_middle=False,
                 attend_at_enc=[False] * 2, attend_at_enc_depth=[1] * 2, attend_at_enc_heads=[2] * 2, memory_efficient=False,
                 use_se_attn='True,', pixel_shuffle_upsample=True, boundary=False, batch_sample=False, batch_sample_factor=3,
                 deep_feature=False).to(device)
params = [p for p in unet.parameters()]
arena = D.FlatArena(params)
D.broadcast_arena(arena)                                   # world_size() == 1: a no-op by contract ...
dist.broadcast(arena.flat, src=0)                          # ... so drive the RCCL broadcast of the flat buffer directly
red = D.BucketedGradReducer(arena, bucket_cap_mb=0.05, first_bucket_mb=0.01, force=True)
assert red.active and len(red.buckets) >= 3, len(red.buckets)

g = torch.Generator().manual_seed(5)
x = torch.randn(2, 1, 8, 8, 8, generator=g).to(device)
lr = torch.randn(2, 1, 8, 8, 8, generator=g).to(device)
t = torch.rand(2, generator=g).to(device)


def loss():
    return unet(x, t, t * 0 + 0.3, lowres_cond_img=lr).float().pow(2).mean()     # (x, time, log-SNR cond of the low-res image)


grads = []
for it in range(3):
    arena.begin_backward()
    red.prepare_backward(sync=True)
    loss().backward()
    red.finalize_backward()
    torch.cuda.synchronize()
    grads.append(arena.grad.clone())
    arena.grad.zero_()
# reference: the same backward with no collective at all
red.active = False
arena.begin_backward()
loss().backward()
arena.collect()
torch.cuda.synchronize()
ref = arena.grad.clone()
assert float(ref.abs().max()) > 0
for it, gq in enumerate(grads):
    assert torch.equal(gq, ref), (it, float((gq - ref).abs().max()))
assert red.used is not None and len(red.used) > 0 and red.stragglers_seen == 0

# bench.py's collectives: MAX over ranks of the timed interval, all_gather of the world size, barrier
tm = torch.tensor([1.25], device=device, dtype=torch.float64)
dist.all_reduce(tm, op=dist.ReduceOp.MAX)
got = [torch.zeros(1, device=device, dtype=torch.int64)]
dist.all_gather(got, torch.tensor([dist.get_world_size()], device=device, dtype=torch.int64))
dist.barrier()
assert float(tm.item()) == 1.25 and int(got[0].item()) == 1
assert D.broadcast_ints([3, 4], device) == [3, 4]
dist.destroy_process_group()
print("RCCL_OK buckets=%d used=%d" % (len(red.buckets), len(red.used)))
'''


@pytest.mark.gpu
def test_single_rank_rccl_group_carries_the_trainer_collectives(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "rccl_worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, DIQT_ROOT=ROOT, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
