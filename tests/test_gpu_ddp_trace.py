"""T5 on the device with N = 2: two fresh processes share the one MI355X of the test box, collectives over
gloo on DEVICE tensors (RCCL refuses two ranks on one card), and replay the real reference's trainer trace with the batch of 2 split
across them -- bucketed all-reduce from the gradient hooks on the 4th micro-step, ``no_sync`` on the first three, fused Adam, the HIP
U-Net kernels.  Asserted in each rank (tests/ddp_trace_worker.py): Adam cadence and ``steps`` per micro-step, mean-over-ranks loss =
the reference's batch loss, identical replicas, ``final_conv.weight`` after the synchronised step = the reference's.
Reference: /root/reference/trainer.py:296-301, 487, 1099-1128."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_two_ranks_on_one_card_reproduce_the_reference_trainer_trace():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   DIQT_DIST_BACKEND="gloo", DIQT_SHARE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "ddp_trace_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=600))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, (p, (so, se)) in enumerate(zip(procs, outs)):
        assert p.returncode == 0 and f"DDP_TRACE rank {r} OK" in so, f"rank {r} rc {p.returncode}\n{so[-2000:]}\n{se[-3000:]}"
