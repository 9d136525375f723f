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


WORKER_GRAPH = r'''
import json, os, sys
sys.path.insert(0, os.environ["DIQT_ROOT"])
import numpy as np
import torch
import torch.distributed as dist
from diffusioniqt_amd import distributed as D, graphs
from diffusioniqt_amd.imagen_pytorch3D import SRUnet256, Imagen, NullUnet
from diffusioniqt_amd.trainer import ImagenTrainer
from oracle import iqt_oracle as O
from tests.conftest import load_golden

world, rank, device = D.init_from_env()
dist.init_process_group(backend="nccl", rank=0, world_size=1, device_id=device)
warm = torch.ones(4, device=device)
dist.all_reduce(warm)                                      # communicator and its watchdog thread are up before any capture
torch.cuda.synchronize()
gu = load_golden('unetA_tiny')
configs = {'Data': {'norm': 'z-score'}, 'Train': {'batch_sample': False, 'patch_size_sub': 8, 'pred_obj': 'x_start'}, 'Eval': {'repeat': 1}}


def run(graph_mode):
    unet = SRUnet256(**json.loads(str(gu['kwargs'])))
    unet.load_state_dict(O.hash_fill_state_dict(unet.state_dict(), 0))
    imagen = Imagen(unets=(NullUnet(), unet), configs=configs, min_bound=float(gu['min_bound']), image_sizes=(8, 8), channels=1,
                    pred_objectives='x_start', timesteps=4, dynamic_thresholding=False, p2_loss_weight_gamma=0.0, cond_drop_prob=0.0).to(device)
    ImagenTrainer.locked = False
    tr = ImagenTrainer(configs=configs, imagen=imagen, verbose=False, gradient_accumulation_steps=2, precision='bf16')
    tr.validate_and_set_unet_being_trained(2)
    # what a multi-rank run has: a live bucketed reducer over the arena (one-rank RCCL mean = identity)
    tr.unet_being_trained.reducer = D.BucketedGradReducer(tr._arena, bucket_cap_mb=0.05, first_bucket_mb=0.01, force=True)
    graphs.TRAIN_ENABLED, graphs.TRAIN_FORCE = graph_mode != 0, graph_mode == 2
    torch.manual_seed(3)
    g = torch.Generator().manual_seed(9)
    losses = []
    for i in range(12):
        hr, lr = torch.randn(2, 1, 8, 8, 8, generator=g), torch.randn(2, 1, 8, 8, 8, generator=g)
        losses.append(tr.forward(hr, lowres_img=lr, unet_number=2, max_batch_size=2)[0])
    torch.cuda.synchronize()
    out = (losses, [p.detach().clone() for p in imagen.unets[1].parameters()], tr._train_graphs.replays, tr._train_graphs.summary())
    tr._train_graphs.clear()
    return out


la, wa, ra, sa = run(2)
lb, wb, rb, sb = run(0)
assert not any("error" in e for e in sa), sa
# accumulation steps 2: every second micro-step is synchronised (eager, with the RCCL all-reduces); the others go through the graph
assert ra >= 2 and rb == 0, (ra, sa)
assert la == lb, (la, lb)
assert all(torch.equal(a, b) for a, b in zip(wa, wb))
dist.destroy_process_group()
print("RCCL_GRAPH_OK replays=%d" % ra)
'''


@pytest.mark.gpu
def test_captured_micro_steps_between_rccl_synchronised_ones(tmp_path):
    """The accumulation micro-steps of a data-parallel run replay as a captured hipGraph while an RCCL communicator (and its watchdog
    thread) is alive and the synchronised micro-steps in between run eagerly with the bucketed all-reduces: same losses and weights as
    the all-eager run."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    script = tmp_path / "rccl_graph_worker.py"
    script.write_text(WORKER_GRAPH)
    env = dict(os.environ, DIQT_ROOT=ROOT, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""), MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(port), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-X", "faulthandler", str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RCCL_GRAPH_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
