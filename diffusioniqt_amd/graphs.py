"""hipGraph capture + replay of a no-grad U-Net evaluation (sampling loops).

The small stages of a cascade are launch-bound: one `Unet3D` eval of the C5 cascade's first stage (32 frames x 32 x 32, batch 8, autocast)
is ~560 kernel launches and 8.8 ms of GPU work but 15 ms of wall time when every launch is issued from Python.  A sampler calls the
same U-Net with the same shapes 2 N - 1 times per stage, so after two eager calls the evaluation is captured once into a hipGraph
(`torch.cuda.CUDAGraph`: the ctypes launches of libdiqt_hip.so go to torch's current stream, which is the capture stream) and replayed:
inputs are copied into the captured buffers, the output is copied out.  Nothing about the computation changes -- the same kernels with the
same arguments -- so results are bit-identical to the eager calls.  Only evaluations that ARE launch-bound are captured: the last eager call
is timed on the host and on the GPU, and a GPU-bound evaluation (the 64^3 stage: 55 ms of kernels behind 6 ms of launching) stays eager.

A captured graph also froze what the eager calls cached on the host side (packed weight copies, position-bias tables): the key holds the
parameters' version counters, so an optimizer step or a `load_state_dict` retires the graph.  `DIQT_GRAPHS=0` switches the cache off, `=2` captures regardless of the timing (tests).
"""
import os
import time
import weakref

import torch

from . import _lib, ops

ENABLED = os.environ.get("DIQT_GRAPHS", "1") != "0"
FORCE = os.environ.get("DIQT_GRAPHS") == "2"       # capture whether or not the evaluation is launch-bound (tests)


class _Uncapturable(Exception):
    """An argument the key cannot describe by value (an arbitrary object): such a call stays eager -- ``id()`` can be reused after GC."""


def _sig(v):
    if torch.is_tensor(v):
        return ("T", tuple(v.shape), v.dtype, v.device.index)
    if isinstance(v, (list, tuple)):
        return tuple(_sig(x) for x in v)
    if isinstance(v, (int, float, bool, str, type(None))):
        return v
    raise _Uncapturable(type(v).__name__)


_CACHES = weakref.WeakSet()


def drop_all():
    """A parameter update made every captured graph stale (the keys hold ``ops._WEIGHT_EPOCH``): drop them NOW, not when their module is next
    sampled from -- while a graph lives, ops.retire() parks every packed-weight copy the training steps replace (one set per optimiser
    step: the bf16 training micro-step of bench.py went from 17.4 to 20-21 ms behind a live sampling graph, and the pins grew without bound)."""
    for c in list(_CACHES):
        c.clear()


class GraphCache:
    def __init__(self, warm=2, max_entries=4):
        _CACHES.add(self)
        self.entries = {}
        self.warm, self.max_entries = warm, max_entries
        self.replays = 0                                 # diagnostics / tests

    def _drop(self, keys):
        n = sum(1 for k in keys if self.entries.pop(k)["graph"] is not None)
        if n:
            ops.graphs_alive(-n)                         # the last graph gone: ops releases what the host caches retired meanwhile

    def clear(self):
        self._drop(list(self.entries))

    def __del__(self):
        try:
            self.clear()
        except Exception:                                # noqa: BLE001 -- interpreter shutdown
            pass

    def run(self, owner, fn, args, kwargs):
        """``fn(*args, **kwargs)`` -- eagerly the first ``warm`` times a (module, shapes, precision, parameter versions) combination is seen,
        through a captured graph afterwards.  ``owner``: the nn.Module whose parameters ``fn`` reads."""
        if not ENABLED or torch.is_grad_enabled() or not args[0].is_cuda or ops.TIMER.enabled:   # timer on: its events must not be captured
            return fn(*args, **kwargs)
        params = list(owner.parameters())
        bufs = list(owner.buffers())
        # what the captured launches depend on besides the inputs: the parameters (version counters; the fused optimiser and any raw-pointer
        # write bump ops._WEIGHT_EPOCH), train / eval mode, buffers, the compute type and the library's run-time switches
        state = (ops.lp_mode(), ops._WEIGHT_EPOCH, _lib.SWITCH_EPOCH, bool(owner.training), sum(p._version for p in params),
                 sum(b._version for b in bufs), params[0].data_ptr() if params else 0)
        try:
            key = (id(owner), state, _sig(args), tuple(sorted((k, _sig(v)) for k, v in kwargs.items())))
        except _Uncapturable:
            return fn(*args, **kwargs)
        ent = self.entries.get(key)
        if ent is None:
            # entries of this module captured under another state (older weights, other switches) can never be hit again: drop them now,
            # so their pools and whatever ops.retire() parked for them are released
            self._drop([k for k in self.entries if k[0] == key[0] and k[1] != state])
            if len(self.entries) >= self.max_entries:
                self.clear()
            ent = self.entries[key] = dict(calls=0, graph=None, failed=False)
        if ent["failed"]:
            return fn(*args, **kwargs)
        if ent["graph"] is None:
            ent["calls"] += 1
            if ent["calls"] < self.warm:
                return fn(*args, **kwargs)
            if ent["calls"] == self.warm:
                # the last eager call is timed on both sides: issuing its launches from Python (host) against its span on the GPU
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                t0 = time.perf_counter()
                out = fn(*args, **kwargs)
                ent["host_ms"] = (time.perf_counter() - t0) * 1e3
                e1.record()
                ent["events"] = (e0, e1)
                return out
            if "launch_bound" not in ent:
                e0, e1 = ent.pop("events")
                e1.synchronize()
                ent["gpu_ms"] = e0.elapsed_time(e1)
                # GPU-bound: the host runs ahead and spends a fraction of the GPU's time issuing; launch-bound: the two are the same
                ent["launch_bound"] = FORCE or ent["host_ms"] > 0.7 * ent["gpu_ms"]
                if not ent["launch_bound"]:
                    ent["failed"] = True                 # stays eager: a replay would only add the input / output copies
                    return fn(*args, **kwargs)
            self._capture(ent, fn, args, kwargs)
            if ent["failed"]:
                return fn(*args, **kwargs)
        for dst, src in zip(ent["args"], args):
            if torch.is_tensor(dst):
                dst.copy_(src)
        for k, dst in ent["kwargs"].items():
            if torch.is_tensor(dst):
                dst.copy_(kwargs[k])
        ent["graph"].replay()
        self.replays += 1
        return ent["out"].clone()

    @staticmethod
    def _capture(ent, fn, args, kwargs):
        clone = lambda v: v.clone() if torch.is_tensor(v) else v
        s_args, s_kwargs = [clone(a) for a in args], {k: clone(v) for k, v in kwargs.items()}
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                # one more eager run on a side stream: every lazily built cache is warm
                fn(*s_args, **s_kwargs)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            ops.capture_begins()
            with torch.cuda.graph(g):
                out = fn(*s_args, **s_kwargs)
            ent.update(graph=g, args=s_args, kwargs=s_kwargs, out=out)
            ops.graphs_alive(+1)
        except Exception as e:                           # noqa: BLE001 -- capture refused (an op that synchronises): stay eager for this key
            torch.cuda.synchronize()
            ent["failed"] = True
            ent["error"] = repr(e)
            if FORCE:                                    # strict mode for tests
                raise


# ------------------------------------------------------------------------------------------------------------------------------
# training micro-steps
# ------------------------------------------------------------------------------------------------------------------------------
TRAIN_ENABLED = os.environ.get("DIQT_TRAIN_GRAPH", "1") != "0"
TRAIN_FORCE = os.environ.get("DIQT_TRAIN_GRAPH") == "2"


class TrainStepGraphs:
    """hipGraph capture + replay of ONE training micro-step: forward, loss, backward and the hand-over of the parameter gradients to
    the flat arena (trainer.ImagenTrainer.forward; trainer.py:1099-1128 of the reference is the loop it sits in).

    The bf16 micro-step of the C2 U-Net is ~675 launches and 12.6 ms of kernel time, but 14.8 ms of wall time when Python (the forward)
    and the autograd engine (the backward) issue them one by one.  The step is the same launch sequence every time -- same shapes, same
    kernels, activations and gradients at the same addresses of the graph's private pool, parameters and the gradient arena at fixed
    addresses -- so after ``warm`` eager micro-steps of a key it is captured once and replayed: the per-step inputs (images, noise,
    noise-schedule coefficients) are copied into the captured buffers; loss / prediction are read from them.  Same kernels, same
    arguments: bit-identical to the eager step.

    What a replay must not depend on is anything the host cached from the WEIGHTS (16-bit packed copies, merged projections): those are
    re-derived inside the graph -- the capture runs with ``ops._WEIGHT_EPOCH`` bumped, so every such cache misses and its pack launch is
    part of the graph -- and a replay after an optimiser step therefore reads the new weights.  Only launch-bound steps are captured
    (host time of the last eager step > 0.7 x its GPU span): the fp32 step (40 ms of kernels behind ~14 ms of issuing) stays eager.
    ``DIQT_TRAIN_GRAPH=0`` switches this off, ``=2`` captures regardless of the timing (tests)."""

    def __init__(self, warm=3, max_entries=2):
        self.entries = {}
        self.warm, self.max_entries = warm, max_entries
        self.replays = 0

    def clear(self):
        n = sum(1 for e in self.entries.values() if e["graph"] is not None)
        self.entries = {}
        if n:
            ops.graphs_alive(-n)

    def __del__(self):
        try:
            self.clear()
        except Exception:                                # noqa: BLE001 -- interpreter shutdown
            pass

    def summary(self):
        """[{captured, host_ms, gpu_ms, error}] per key seen (diagnostics: bench.py prints it beside the training lines)."""
        return [dict(captured=e["graph"] is not None, calls=e["calls"], host_ms=round(e.get("host_ms", 0.0), 3), gpu_ms=round(e.get("gpu_ms", 0.0), 3),
                     **({"error": e["error"]} if e.get("error") else {})) for e in self.entries.values()]

    def run(self, key, fn, tensors, recover, stream=None, prepare=None):
        """``fn(*tensors)`` (tensors: device tensors or None) -> tuple of tensors / None.  ``recover()`` restores the host-side state a
        failed capture left half-changed (nothing has run on the GPU then).  ``stream``: the (non-default) stream the steps run on -- the
        capture happens on it as well, so autograd nodes bound to it stay inside the capture.  ``prepare()``: launches to capture in front of
        the step (the trainer re-derives all packed weights there with one multi-weight launch)."""
        if not TRAIN_ENABLED or ops.TIMER.enabled:
            return fn(*tensors)
        key = key + (_sig(tensors), ops.lp_mode(), _lib.SWITCH_EPOCH)
        ent = self.entries.get(key)
        if ent is None:
            if len(self.entries) >= self.max_entries:
                self.clear()
            ent = self.entries[key] = dict(calls=0, graph=None, failed=False)
        if ent["failed"]:
            return fn(*tensors)
        if ent["graph"] is None:
            ent["calls"] += 1
            if ent["calls"] < self.warm:
                return fn(*tensors)
            if ent["calls"] == self.warm:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                t0 = time.perf_counter()
                out = fn(*tensors)
                ent["host_ms"] = (time.perf_counter() - t0) * 1e3
                e1.record()
                ent["events"] = (e0, e1)
                return out
            e0, e1 = ent.pop("events")
            e1.synchronize()
            ent["gpu_ms"] = e0.elapsed_time(e1)
            if not (TRAIN_FORCE or ent["host_ms"] > 0.7 * ent["gpu_ms"]):
                ent["failed"] = True                     # GPU-bound: the host already runs ahead of the kernels
                return fn(*tensors)
            static = [t.clone() if torch.is_tensor(t) else t for t in tensors]
            try:
                torch.cuda.synchronize()
                ops.bump_weight_epoch()                  # every weight-derived host cache misses: its launch becomes part of the graph
                g = torch.cuda.CUDAGraph()
                ops.capture_begins()
                # thread_local: a capture lasts tens of milliseconds in the middle of a training run -- other threads of the process (a
                # DataLoader's pin-memory thread allocating host memory, RCCL's watchdog) must not be refused their runtime calls meanwhile
                with torch.cuda.graph(g, stream=stream, capture_error_mode="thread_local"):
                    if prepare is not None:
                        prepare()
                    out = fn(*static)
                ent.update(graph=g, static=static, out=out)
                ops.graphs_alive(+1)
            except Exception as e:                       # noqa: BLE001 -- capture refused: this key stays eager
                torch.cuda.synchronize()
                ent["failed"] = True
                ent["error"] = repr(e)
                recover()
                if TRAIN_FORCE:
                    raise
                return fn(*tensors)
        else:
            for dst, src in zip(ent["static"], tensors):
                if torch.is_tensor(dst):
                    dst.copy_(src)
        ent["graph"].replay()
        self.replays += 1
        return tuple(o.clone() if torch.is_tensor(o) else o for o in ent["out"])
