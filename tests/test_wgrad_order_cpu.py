"""Host logic of functions.WgradOrder (which order of weight gradient and input-gradient GEMM a workload's backward plan uses): the
two orders are alternated for the first calls of a workload, finished event pairs are harvested without synchronising, "early" is
kept only if its median beats "late"'s by the margin; "late" / "early" modes never measure.  (No GPU: events are faked.)"""
import pytest

from reactranker_amd import functions as Fn


class FakeEvent:
    clock = 0.0
    cost = {False: 5.0, True: 5.0}
    current = False

    def __init__(self, enable_timing=True):
        self.t = None

    def record(self, stream=None):
        self.t = FakeEvent.clock

    def query(self):
        return True

    def elapsed_time(self, other):
        return other.t - self.t


@pytest.fixture
def tuner(monkeypatch):
    monkeypatch.setattr(Fn.torch.cuda, "Event", FakeEvent)
    monkeypatch.setattr(Fn.WgradOrder, "mode", "auto")
    Fn.WgradOrder.reset()
    yield Fn.WgradOrder
    Fn.WgradOrder.reset()


def _run_calls(W, sig, n, cost):
    orders = []
    for _ in range(n):
        early, ev = W.begin(sig)
        orders.append(early)
        if ev is not None:
            ev[1].record()
            FakeEvent.clock += cost[early]
            ev[2].record()
            W.end(sig, ev)
        else:
            FakeEvent.clock += cost[early]
    return orders


@pytest.mark.parametrize("cost,want", [({False: 5.00, True: 4.90}, True),      # early clearly faster: kept
                                       ({False: 5.00, True: 4.99}, False),     # inside the margin: the order in use stays
                                       ({False: 5.00, True: 5.10}, False)])
def test_the_order_is_measured_and_then_frozen(tuner, cost, want):
    sig = (300, 3, 3, 2, 8)
    orders = _run_calls(tuner, sig, 40, cost)
    assert orders[:tuner.warm] == [False] * tuner.warm                         # warm-up calls run the order in use
    measured = orders[tuner.warm:tuner.warm + 2 * tuner.samples]
    assert measured.count(True) == tuner.samples and measured.count(False) == tuner.samples
    assert tuner.settled()
    assert all(o == want for o in orders[tuner.warm + 2 * tuner.samples + 1:])
    assert tuner.choices()[str(sig)] is want


def test_workloads_are_tuned_separately_and_fixed_modes_never_measure(tuner, monkeypatch):
    a, b = (300, 3, 3, 2, 8), (600, 6, 6, 2, 8)
    _run_calls(tuner, a, 20, {False: 5.0, True: 4.0})
    assert tuner.settled()
    early, ev = tuner.begin(b)                                                  # a new workload starts over
    assert early is False and ev is None and not tuner.settled()
    _run_calls(tuner, b, 20, {False: 5.0, True: 6.0})
    assert tuner.choices() == {str(a): True, str(b): False}
    monkeypatch.setattr(tuner, "mode", "early")
    assert tuner.begin((1, 1, 1, 0, 0)) == (True, None) and tuner.settled()
    monkeypatch.setattr(tuner, "mode", "late")
    assert tuner.begin((1, 1, 1, 0, 0)) == (False, None)
