"""N processes sharing GPU 0, gloo all_reduce (async) on CUDA tensors -- is a hang of `bench.py --gpus 4 --rehearse` the reducer's or
the (artificial) gloo-on-one-shared-GPU set-up's?   python tools/gloo_cuda_probe.py N [MB per tensor]"""
import os, sys, time, socket
import torch, torch.distributed as dist
import torch.multiprocessing as mp


def worker(rank, world, port, mb):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import faulthandler
    faulthandler.dump_traceback_later(60, exit=True)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ts = [torch.full((int(m * (1 << 18)),), float(rank + 1), device="cuda") for m in (1, mb, mb)]
    for it in range(4):
        t0 = time.time()
        hs = [dist.all_reduce(t, async_op=True) for t in ts]
        for h in hs:
            h.wait()
        torch.cuda.synchronize()
        if rank == 0:
            print(f"iter {it}: {time.time() - t0:.3f} s, value {ts[1][0].item()}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    world = int(sys.argv[1])
    mb = float(sys.argv[2]) if len(sys.argv) > 2 else 25.0
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    mp.start_processes(worker, args=(world, port, mb), nprocs=world, start_method="spawn")
    print("PROBE_DONE")
