"""N>1 path on CPU: world_size-2 gloo runs of the flat-arena bucketed reducer and of ImagenTrainer's
split-batch data parallelism (device ops replaced by tests/cpu_doubles.py, U-Net by the CPU oracle)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _setup(rank, world, port):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(2)


def _reducer_worker(rank, world, port, out):
    _setup(rank, world, port)
    from diffusioniqt_amd import distributed as D
    D.init_from_env(device_type="cpu")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(40, 300), torch.nn.Tanh(), torch.nn.Linear(300, 300), torch.nn.Tanh(),
                              torch.nn.Linear(300, 5))
    unused = torch.nn.Linear(7, 7)                       # never takes part in forward (like mid_block)
    params = list(net.parameters()) + list(unused.parameters())
    arena = D.FlatArena(params)
    if rank == 1:
        with torch.no_grad():
            arena.flat.add_(1.0)                         # replicas start different ...
    D.broadcast_arena(arena)                             # ... and are synchronised from rank 0
    red = D.BucketedGradReducer(arena, bucket_cap_mb=0.02, first_bucket_mb=0.005)
    assert len(red.buckets) >= 3
    g = torch.Generator().manual_seed(100)
    X = torch.randn(3, 8, 40, generator=g)               # 3 iterations of a global batch of 8
    per = 8 // world
    res = {}
    for it in range(3):
        sync = it != 1                                   # iteration 1 is a no_sync accumulation micro-step
        red.prepare_backward(sync=sync)
        xs = X[it, rank * per:(rank + 1) * per]
        net(xs).pow(2).mean().backward()
        red.finalize_backward()
        res[it] = arena.grad.clone()
        if sync:
            arena.grad.zero_()
    # single-process reference on the full batch
    ref = {}
    full = [p.detach().clone().requires_grad_() for p in params]

    def fwd(x):
        h = torch.tanh(torch.nn.functional.linear(x, full[0], full[1]))
        h = torch.tanh(torch.nn.functional.linear(h, full[2], full[3]))
        return torch.nn.functional.linear(h, full[4], full[5])
    acc = None
    for it in range(3):
        gs = torch.autograd.grad(fwd(X[it]).pow(2).mean(), full[:6])
        flat = torch.zeros_like(arena.grad)
        for gq, o, p in zip(gs, arena.offsets, params):
            flat[o:o + p.numel()] = gq.reshape(-1)
        if it == 1:
            acc = None       # local-only on each rank: checked separately
        ref[it] = flat
    ok0 = torch.allclose(res[0], ref[0], atol=1e-6)
    # iteration 1 (no_sync): gradient is the LOCAL shard's; iteration 2 adds the synced gradient of batch 2 on top
    loc = torch.autograd.grad(fwd(X[1, rank * per:(rank + 1) * per]).pow(2).mean(), full[:6])
    flat1 = torch.zeros_like(arena.grad)
    for gq, o, p in zip(loc, arena.offsets, params):
        flat1[o:o + p.numel()] = gq.reshape(-1)
    ok1 = torch.allclose(res[1], flat1, atol=1e-6)
    # after sync at it=2 every rank holds mean over ranks of (local grad it1 + local grad it2) = full-batch grads summed
    ok2 = torch.allclose(res[2], ref[1] + ref[2], atol=1e-6)
    unused_zero = float(arena.grad[arena.offsets[6]:].abs().max()) == 0.0
    same_w = [torch.zeros_like(arena.flat) for _ in range(world)]
    dist.all_gather(same_w, arena.flat)
    out.put((rank, ok0, ok1, ok2, unused_zero, torch.equal(same_w[0], same_w[1]), sorted(red.used) == list(range(6))))
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_bucketed_reducer_world2_gloo(world):
    """Bucketed all-reduce from the gradient hooks, a no_sync micro-step in between, unused parameters, replicas synchronised from
    rank 0 -- at 2 and at 4 ranks (power-of-two worlds take gloo's other reduction schedule)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_reducer_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    results = [q.get(timeout=120) for _ in range(world)]
    [p.join(60) for p in procs]
    for r in results:
        assert all(r[1:]), r


def _trainer_worker(rank, world, port, out):
    _setup(rank, world, port)
    from tests.cpu_doubles import cpu_op_doubles
    from tests.test_host_trainer import make_trainer, T
    from tests.conftest import load_golden
    g = load_golden('trainerA_trace')
    with cpu_op_doubles():
        trainer, unet = make_trainer()
        assert trainer.world_size == world and trainer.is_distributed and (trainer.use_ema == (rank == 0))
        trainer.training = True
        idx = unet.names.index('final_conv.weight')
        for i in range(4):
            hr = T(g['hr'][i])[rank:rank + 1]            # split_batches: each rank takes its 1/2 of the batch of 2
            lr = T(g['lowres'][i])[rank:rank + 1]
            times = T(g['times'][i])[rank:rank + 1]
            trainer.imagen.noise_schedulers[1].sample_random_times = lambda b, device, t=times: t.clone()
            trainer.forward(hr, lowres_img=lr, unet_number=2, max_batch_size=1, noise=T(g['noise'][i])[rank:rank + 1])
        w = unet.plist[idx].detach().flatten()
        ws = [torch.zeros_like(w) for _ in range(world)]
        dist.all_gather(ws, w)
        ref = T(g['final_conv_w'][3])
        out.put((rank, torch.equal(ws[0], ws[1]), bool(torch.allclose(w, ref, atol=5e-6, rtol=1e-3)),
                 float((w - ref).abs().max())))
    dist.destroy_process_group()


def test_trainer_data_parallel_world2_matches_single_process_reference_trace():
    """2 ranks x 1 patch with no_sync on micro-steps 1-3 and an all-reduce on the 4th reproduce the weights the
    single-process reference reached with batches of 2 (mean over ranks of per-rank means == full-batch mean)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_trainer_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    results = [q.get(timeout=300) for _ in range(2)]
    [p.join(60) for p in procs]
    for r in results:
        assert r[1] and r[2], r


def test_shard_batch_completes_a_short_last_batch_like_accelerate_even_batches():
    """drop_last=False loaders (train.py:56/67) end an epoch on a short batch: it is completed to the full batch size from the
    epoch's first batch (accelerate BatchSamplerShard, split_batches + even_batches), never asserted on."""
    from diffusioniqt_amd.distributed import shard_batch
    first = torch.arange(8.).view(8, 1)
    last = torch.arange(100., 103.).view(3, 1)                    # 3 of 8 samples left
    got = [shard_batch(last, 4, r, full=8, initial=first).flatten().tolist() for r in range(4)]
    assert got == [[100., 101.], [102., 0.], [1., 2.], [3., 4.]]
    assert [shard_batch(first, 4, r, full=8, initial=first).flatten().tolist() for r in range(4)] == \
        [[0., 1.], [2., 3.], [4., 5.], [6., 7.]]
    # degenerate: the first batch is short too -> cycled
    tiny = torch.tensor([[7.], [8.]])
    got = torch.cat([shard_batch(tiny[:1], 4, r, full=4, initial=tiny) for r in range(4)]).flatten().tolist()
    assert got == [7., 7., 8., 7.]
    # no loader batch size known: next multiple of world, from the batch's own head
    assert shard_batch(last, 2, 1).flatten().tolist() == [102., 100.]
    assert shard_batch(last, 1, 0) is last


def _straggler_worker(rank, world, port, out):
    _setup(rank, world, port)
    from diffusioniqt_amd import distributed as D
    D.init_from_env(device_type="cpu")
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(40, 300), torch.nn.Tanh(), torch.nn.Linear(300, 5))
    late = torch.nn.Linear(5, 5)                         # joins the graph only from iteration 1 on
    params = list(net.parameters()) + list(late.parameters())
    arena = D.FlatArena(params)
    D.broadcast_arena(arena)
    red = D.BucketedGradReducer(arena, bucket_cap_mb=0.02, first_bucket_mb=0.0001)
    g = torch.Generator().manual_seed(7)
    X = torch.randn(3, 8, 40, generator=g)
    per = 8 // world
    full = [p.detach().clone().requires_grad_() for p in params]
    ok = []
    for it in range(3):
        red.prepare_backward(sync=True)
        arena.begin_backward()
        xs = X[it, rank * per:(rank + 1) * per]
        y = net(xs)
        if it >= 1:
            y = late(y)
        y.pow(2).mean().backward()
        red.finalize_backward()
        arena.collect()
        h = torch.nn.functional.linear(torch.tanh(torch.nn.functional.linear(X[it], full[0], full[1])), full[2], full[3])
        if it >= 1:
            h = torch.nn.functional.linear(h, full[4], full[5])
        gs = torch.autograd.grad(h.pow(2).mean(), full if it >= 1 else full[:4])
        ref = torch.zeros_like(arena.grad)
        for gq, o, p in zip(gs, arena.offsets, params):
            ref[o:o + p.numel()] = gq.reshape(-1)
        ok.append(bool(torch.allclose(arena.grad, ref, atol=1e-6)))
        arena.grad.zero_()
    out.put((rank, ok, red.stragglers_seen, sorted(red.used)))
    dist.destroy_process_group()


def test_reducer_reduces_late_parameters_instead_of_racing_world2_gloo():
    """A parameter outside the learned ``used`` set that receives a gradient later is reduced in a second pass of its bucket
    (and joins the set) — no in-place add into a range whose collective is in flight, no silent replica divergence."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_straggler_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    results = [q.get(timeout=120) for _ in range(2)]
    [p.join(60) for p in procs]
    for r in results:
        assert r[1] == [True, True, True], r
        assert r[2] == 2 and r[3] == list(range(6)), r          # weight + bias of `late` seen once as stragglers, then used


def _resume_worker(rank, world, port, ckdir, out):
    _setup(rank, world, port)
    from tests.cpu_doubles import cpu_op_doubles
    from tests.test_host_trainer import make_trainer, T
    from tests.conftest import load_golden
    g = load_golden('trainerA_trace')
    with cpu_op_doubles():
        trainer, unet = make_trainer(checkpoint_path=ckdir, checkpoint_every=1000)      # only rank 0 reads the folder
        trainer.training = True
        for i in range(4, 6):                                                            # micro-steps 5 and 6 of the trace
            sl = slice(rank, rank + 1)
            times = T(g['times'][i])[sl]
            trainer.imagen.noise_schedulers[1].sample_random_times = lambda b, device, t=times: t.clone()
            trainer.forward(T(g['hr'][i])[sl], lowres_img=T(g['lowres'][i])[sl], unet_number=2, max_batch_size=1,
                            noise=T(g['noise'][i])[sl])
        opt = trainer.optim1
        idx = unet.names.index('final_conv.weight')
        w = unet.plist[idx].detach().flatten()
        ws = [torch.zeros_like(w) for _ in range(world)]
        dist.all_gather(ws, w)
        m = [torch.zeros_like(opt.exp_avg) for _ in range(world)]
        dist.all_gather(m, opt.exp_avg)
        out.put((rank, torch.equal(ws[0], ws[1]), torch.equal(m[0], m[1]) and float(m[0].abs().max()) > 0, opt.step_count,
                 int(trainer.steps[1]), trainer._micro_step))
    dist.destroy_process_group()


def test_resume_under_world2_hands_optimizer_state_to_every_replica(tmp_path):
    """Rank 0 alone loads the checkpoint folder (trainer.py:368-370); wrap_unet must broadcast Adam's moments / step, ``steps``
    and the accumulation phase with the weights, or the replicas diverge at the first optimiser step after a resume."""
    from tests.cpu_doubles import cpu_op_doubles
    from tests.test_host_trainer import make_trainer, T
    from tests.conftest import load_golden
    g = load_golden('trainerA_trace')
    ckdir = str(tmp_path / 'ck')
    with cpu_op_doubles():
        trainer, unet = make_trainer()
        trainer.training = True
        for i in range(4):                               # one optimiser step
            times = T(g['times'][i])
            trainer.imagen.noise_schedulers[1].sample_random_times = lambda b, device, t=times: t.clone()
            trainer.forward(T(g['hr'][i]), lowres_img=T(g['lowres'][i]), unet_number=2, max_batch_size=2, noise=T(g['noise'][i]))
        trainer.save(os.path.join(ckdir, 'checkpoint.4.pt'))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_resume_worker, args=(r, 2, port, ckdir, q)) for r in range(2)]
    [p.start() for p in procs]
    results = [q.get(timeout=300) for _ in range(2)]
    [p.join(60) for p in procs]
    for r in results:
        assert r[1] and r[2], r
        assert r[3:] == (1, 6, 2), r                      # Adam step count, `steps`, micro-step phase agree on both ranks
