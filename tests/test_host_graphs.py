"""Host logic of the hipGraph cache (diffusioniqt_amd/graphs.py) that needs no GPU: keys, the eager gates, entry bookkeeping."""
import torch

from diffusioniqt_amd import graphs


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.ones(3))
        self.calls = 0

    def forward(self, x, t, flag=None):
        self.calls += 1
        return x * self.w.sum() + t


def test_cpu_tensors_and_grad_mode_stay_eager():
    net, cache = _Net(), graphs.GraphCache()
    x, t = torch.ones(2, 3), torch.zeros(2, 1)
    with torch.no_grad():
        for _ in range(5):
            y = cache.run(net, net.forward, (x, t), dict(flag=None))
    assert net.calls == 5 and cache.replays == 0 and not cache.entries          # CPU tensors: never keyed, never captured
    assert torch.equal(y, x * 3)
    y = cache.run(net, net.forward, (x, t), {})                                  # autograd on: eager too
    assert y.requires_grad and net.calls == 6


def test_signature_distinguishes_shapes_dtypes_and_plain_values():
    a, b = torch.zeros(2, 3), torch.zeros(2, 4)
    assert graphs._sig(a) != graphs._sig(b) and graphs._sig(a) == graphs._sig(torch.ones(2, 3))
    assert graphs._sig(a) != graphs._sig(a.double())
    assert graphs._sig((a, 1.5, None)) == graphs._sig((torch.ones(2, 3), 1.5, None))
    assert graphs._sig(0.5) == 0.5 and graphs._sig("x") == "x"
    import pytest
    with pytest.raises(graphs._Uncapturable):                                    # id() can be reused after GC: such calls stay eager
        graphs._sig(object())


def test_retired_cache_tensors_are_pinned_while_a_graph_is_alive():
    """A captured graph holds raw addresses of host-cached tensors (scratch arena, packed weights, position-bias tables): a cache that
    replaces one hands the old tensor to ops.retire(), which keeps it until the last graph is dropped (two use-after-frees)."""
    from diffusioniqt_amd import ops
    ops.graphs_alive(-10 ** 6)
    a, b = torch.zeros(4), torch.zeros(5)
    ops.retire(a)
    assert not ops._GRAPH_PINS                                                   # no graph alive: nothing to pin
    ops.graphs_alive(+2)
    ops.retire(a, None, b)
    assert [t.numel() for t in ops._GRAPH_PINS] == [4, 5]
    ops.graphs_alive(-1)
    assert len(ops._GRAPH_PINS) == 2                                             # one graph left
    cache = graphs.GraphCache()
    cache.entries[(1, 'old')] = dict(graph=object(), failed=False, calls=9)
    cache._drop([(1, 'old')])
    assert not ops._GRAPH_PINS and ops._LIVE_GRAPHS == 0


def test_a_weight_update_drops_every_captured_graph():
    """The fused optimiser bumps ops._WEIGHT_EPOCH: every captured graph is stale from then on (its key holds the epoch) and is dropped at
    once -- a live graph makes ops.retire() park every packed-weight copy training replaces, one set per optimiser step."""
    from diffusioniqt_amd import ops
    ops.graphs_alive(-10 ** 6)
    c1, c2 = graphs.GraphCache(), graphs.GraphCache()
    c1.entries[(1, 'a')] = dict(graph=object(), failed=False, calls=9)
    c2.entries[(2, 'b')] = dict(graph=object(), failed=False, calls=9)
    c2.entries[(3, 'c')] = dict(graph=None, failed=False, calls=1)               # still in its eager warm-up
    ops.graphs_alive(+2)
    ops.retire(torch.zeros(3))
    assert len(ops._GRAPH_PINS) == 1
    ops.bump_weight_epoch()
    assert not c1.entries and not c2.entries and ops._LIVE_GRAPHS == 0 and not ops._GRAPH_PINS
    e = ops._WEIGHT_EPOCH
    ops.bump_weight_epoch()                                                      # nothing alive: just the counter
    assert ops._WEIGHT_EPOCH == e + 1


def test_tensors_born_after_the_newest_capture_are_not_pinned():
    """With a captured TRAINING step alive for a whole run, every validation / sampling pass in between re-packs the weights; those copies
    were created after the newest capture began -- no graph can address them (or they live in that graph's own pool) -- and must not pile
    up in ops._GRAPH_PINS; a copy made BEFORE a capture may be addressed by it and stays pinned."""
    from diffusioniqt_amd import ops
    ops.graphs_alive(-10 ** 6)
    old = ops.born(torch.zeros(3))                   # cached before the capture: the graph may hold its address
    ops.capture_begins()
    ops.graphs_alive(+1)
    new = ops.born(torch.zeros(4))                   # cached after it
    ops.retire(new)
    assert not ops._GRAPH_PINS
    ops.retire(old)
    assert [t.numel() for t in ops._GRAPH_PINS] == [3]
    ops.capture_begins()                             # a later capture (a sampling graph) may address `new` after all
    ops.retire(new)
    assert [t.numel() for t in ops._GRAPH_PINS] == [3, 4]
    ops.graphs_alive(-1)
    assert not ops._GRAPH_PINS


def test_train_step_graphs_stay_eager_when_switched_off_or_timed():
    """graphs.TrainStepGraphs: with the switch off (or the kernel timer on) the micro-step function is simply called."""
    from diffusioniqt_amd import ops
    g = graphs.TrainStepGraphs()
    calls = []
    fn = lambda a, b: (calls.append(1), (a + 1, None))[1]
    old = graphs.TRAIN_ENABLED
    graphs.TRAIN_ENABLED = False
    try:
        for _ in range(6):
            out = g.run(("k",), fn, (torch.zeros(2), None), lambda: None)
    finally:
        graphs.TRAIN_ENABLED = old
    assert len(calls) == 6 and g.replays == 0 and not g.entries and torch.equal(out[0], torch.ones(2))
    ops.TIMER.enabled = True
    try:
        g.run(("k",), fn, (torch.zeros(2), None), lambda: None)
    finally:
        ops.TIMER.enabled = False
    assert len(calls) == 7 and not g.entries and g.summary() == []
