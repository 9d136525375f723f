"""One rank of tests/test_gpu_ddp_trace.py: the product's ImagenTrainer + HIP SRUnet256 replaying its share of the REAL reference's trainer
trace (tests/golden/trainerA_trace.npz) as rank RANK of WORLD_SIZE on cuda:0, collectives over gloo on device tensors
(DIQT_DIST_BACKEND=gloo, DIQT_SHARE_DEVICE=1: RCCL refuses two ranks on one card -- the reducer code, bucket order, hooks, no_sync
cadence and the kernels are the production ones).  Prints one line ``DDP_TRACE rank ... OK`` or raises."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np   # noqa: E402
import torch         # noqa: E402
import torch.distributed as dist   # noqa: E402

from tests.conftest import load_golden   # noqa: E402
from tests.test_gpu_trainer_trace import make_gpu_trainer   # noqa: E402

T = lambda a: torch.from_numpy(np.asarray(a))


def main():
    g = load_golden('trainerA_trace')
    trainer, unet = make_gpu_trainer()
    world, rank = trainer.world_size, trainer.rank if hasattr(trainer, 'rank') else int(os.environ["RANK"])
    assert world == int(os.environ["WORLD_SIZE"]) == 2 and trainer.is_distributed, (world, trainer.is_distributed)
    assert trainer.use_ema == (rank == 0)                                  # EMA + checkpoints on rank 0 only (trainer.py:319, 407)
    assert next(unet.parameters()).is_cuda
    trainer.training = True
    unet.train()
    w = unet.final_conv.weight
    losses = []
    for i in range(4):                                                     # one optimiser step: 3 no_sync micro-steps + 1 synchronised
        sl = slice(rank, rank + 1)                                         # split_batches: each rank takes its half of the batch of 2
        times = T(g['times'][i])[sl]
        trainer.imagen.noise_schedulers[1].sample_random_times = lambda b, device, t=times: t.clone().to(device)
        before = w.detach().clone()
        loss, *_ = trainer.forward(T(g['hr'][i])[sl], lowres_img=T(g['lowres'][i])[sl], unet_number=2, max_batch_size=1,
                                   noise=T(g['noise'][i])[sl])
        losses.append(float(loss))
        changed = not torch.equal(before, w.detach())
        assert changed == bool(g['trace'][i][1]), f"micro-step {i}: Adam cadence (changed={changed})"
        assert int(trainer.steps[1].item()) == int(g['trace'][i][0])
    # the mean over ranks of the per-rank (1-sample) losses is the reference's batch-of-2 loss
    lt = torch.tensor(losses, dtype=torch.float64)
    both = [torch.zeros_like(lt) for _ in range(world)]
    dist.all_gather(both, lt)
    mean = (both[0] + both[1]) / 2
    ref_l = torch.tensor([float(v) for v in g['losses'][:4]], dtype=torch.float64)
    assert torch.allclose(mean, ref_l, rtol=2e-5, atol=0), (mean, ref_l)
    wv = w.detach().flatten().cpu()
    ws = [torch.zeros_like(wv) for _ in range(world)]
    dist.all_gather(ws, wv)
    assert torch.equal(ws[0], ws[1]), "replicas diverged"
    ref = T(g['final_conv_w'][3])
    err = float((wv - ref).abs().max())
    assert torch.allclose(wv, ref, atol=2e-6, rtol=1e-4), f"weights after the synchronised Adam step: max diff {err:.3e}"
    from diffusioniqt_amd import _lib
    assert "libdiqt_hip.so" in _lib.load()._name
    print(f"DDP_TRACE rank {rank} OK loss_mean {mean.tolist()} w_err {err:.3e}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
