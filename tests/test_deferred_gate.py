"""CPU: the protocol of functional.defer_gate / defer_trunk on plain torch modules (no kernels but the package's axpby, which
the gate uses to add a second pass's gradients: kernel interpreter).  The reference's step back-propagates every loss with a
``backward()`` call of its own (train.py:486 ``retain_graph=True``, 499, 510); behind a gate each region of the graph is walked
ONCE per forward -- also with a second gate further up the model (the PAD decoder's, where its two halves exchange attention maps)
-- and the gradients equal those of the call-per-loss walk.  The full-model versions run on the GPU (tests/test_models_gpu.py:
test_deferred_trunk_backward, test_train_step_replay_vs_reference_caller) and, opt-in, under the interpreter
(tests/test_models_emu.py with SEGSDE_SLOW_TESTS=1)."""
import pytest
import torch

import emu


pytestmark = pytest.mark.filterwarnings("ignore:Full backward hook is firing")


@pytest.fixture(scope="module", autouse=True)
def _emu():
    emu.install()


class _Net(torch.nn.Module):
    """trunk -> (two "first halves" that both read the trunk) -> cross-over -> two heads, like PAD on an encoder"""

    def __init__(self):
        super().__init__()
        lin = torch.nn.Linear
        self.t1, self.t2 = lin(6, 24), lin(24, 24)                # trunk: two features
        self.a1, self.b1 = lin(24, 16), lin(24, 16)               # first halves
        self.xa, self.xb = lin(16, 16), lin(16, 16)               # the cross-over ("attention" of the other branch)
        self.a2, self.b2 = lin(16 + 24, 1), lin(16 + 24, 1)       # second halves: also read a trunk feature (skip connection)
        self.side = lin(16, 1)                                    # a low-resolution output of branch a's first half
        self.defer = self.gate2 = False
        self.calls = {}
        for name in ("t1", "t2", "a1", "b1", "a2", "b2"):
            getattr(self, name).register_full_backward_hook(lambda m, gi, go, name=name: self.calls.__setitem__(name, self.calls.get(name, 0) + 1))

    def forward(self, x):
        from improving_segmentation_with_selfsupervised_depth_amd import functional as Fn
        f1 = torch.relu(self.t1(x))
        f2 = torch.relu(self.t2(f1))
        feats = [f1, f2]
        if self.defer:
            feats = Fn.defer_trunk(feats, id(self))
        ha, hb = torch.relu(self.a1(feats[1])), torch.relu(self.b1(feats[1]))
        side = self.side(ha)
        if self.defer and self.gate2:
            ha, side, hb = Fn.defer_gate([ha, side, hb], id(self) + 1)
        za, zb = ha + self.xb(hb), hb + self.xa(ha)
        ya = self.a2(torch.cat([za, feats[0]], 1))
        yb = self.b2(torch.cat([zb, feats[0]], 1))
        return ya, yb, side


def _losses(net, x, t):
    ya, yb, side = net(x)
    return ((ya[:, 0] - t[:, 0]) ** 2).mean() + (side[:, 0] - t[:, 0]).abs().mean(), ((yb[:, 0] - t[:, 1]) ** 2).mean()


def _grads(net):
    return {k: p.grad.clone() for k, p in net.named_parameters()}


@pytest.mark.parametrize("gate2", [False, True], ids=["trunk_gate", "trunk_and_crossover_gates"])
def test_every_region_is_walked_once_and_gradients_match(gate2):
    from improving_segmentation_with_selfsupervised_depth_amd import functional as Fn
    torch.manual_seed(3)
    net = _Net()
    x, t = torch.randn(8, 6), torch.randn(8, 2)
    # the reference's walk: one backward() per loss, everything shared is walked twice
    la, lb = _losses(net, x, t)
    la.backward(retain_graph=True)
    lb.backward()
    want, calls_ref = _grads(net), dict(net.calls)
    assert calls_ref["t1"] == 2 and calls_ref["a1"] == 2 and calls_ref["b1"] == 2
    net.zero_grad(set_to_none=True)
    net.calls.clear()
    net.defer, net.gate2 = True, gate2
    t0 = (Fn.TrunkGateFn.trunk_backwards, Fn.TrunkGateFn.parked_passes)
    la, lb = _losses(net, x, t)
    la.backward(retain_graph=True)
    assert net.t1.weight.grad is None                       # the keeping pass parks at the gate: the trunk has not been walked
    assert Fn.pending_deferred_trunks() == (2 if gate2 else 1)
    if gate2:
        assert net.a1.weight.grad is None and net.b1.weight.grad is None
    lb.backward()
    assert Fn.pending_deferred_trunks() == 0
    got = _grads(net)
    for k in want:
        torch.testing.assert_close(got[k], want[k], rtol=1e-5, atol=1e-6, msg=k)
    assert net.calls["t1"] == 1 and net.calls["t2"] == 1    # one trunk backward for the two calls
    assert (net.calls["a1"], net.calls["b1"]) == ((1, 1) if gate2 else (2, 2))
    n = 2 if gate2 else 1
    assert (Fn.TrunkGateFn.trunk_backwards - t0[0], Fn.TrunkGateFn.parked_passes - t0[1]) == (n, n)
    # one call on the sum: nothing is parked, the result is the same
    net.zero_grad(set_to_none=True)
    la, lb = _losses(net, x, t)
    (la + lb).backward()
    assert Fn.pending_deferred_trunks() == 0
    got = _grads(net)
    for k in want:
        torch.testing.assert_close(got[k], want[k], rtol=1e-5, atol=1e-6, msg=k)


def test_a_parked_gradient_never_goes_unnoticed():
    from improving_segmentation_with_selfsupervised_depth_amd import functional as Fn
    torch.manual_seed(4)
    net = _Net()
    net.defer = net.gate2 = True
    x, t = torch.randn(4, 6), torch.randn(4, 2)
    la, lb = _losses(net, x, t)
    la.backward(retain_graph=True)                          # a monodepth-only step: its only backward() keeps the graph (train.py:486)
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    with pytest.raises(RuntimeError, match="deferred trunk backward"):
        opt.step()                                          # any optimizer step refuses to run on half a gradient ...
    Fn.flush_deferred_trunks()                              # ... until it is flushed by hand
    assert Fn.pending_deferred_trunks() == 0 and net.t1.weight.grad is not None and net.a1.weight.grad is not None
    opt.step()
    la, lb = _losses(net, x, t)
    la.backward(retain_graph=True)
    with pytest.raises(RuntimeError, match="deferred trunk backward"):
        net(x)                                              # the next forward says so once and drops the lost gradients ...
    assert Fn.pending_deferred_trunks() == 0
    la, lb = _losses(net, x, t)                             # ... a loop that catches the error can go on
    (la + lb).backward()
    opt.step()
    with torch.no_grad():                                   # without a gradient (validation) no gate is built
        ya, _, _ = net(x)
    assert not ya.requires_grad and Fn.pending_deferred_trunks() == 0
    # a pass that comes after the releasing one finds the graph behind the gate consumed, like any graph walked twice
    la, lb = _losses(net, x, t)
    lb.backward()
    with pytest.raises(RuntimeError):
        la.backward()
    Fn.flush_deferred_trunks()
