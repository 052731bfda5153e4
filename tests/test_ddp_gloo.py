"""CPU, world_size 2 over gloo: the bucketed gradient all-reducer gives every rank the average of the per-rank
gradients (= the gradient of the mean loss over the global batch), skips parameters that never get a gradient, and
keeps working across steps with overlap hooks -- also when a step calls backward() several times like the reference's
train step (train.py:486-510, 698, 724), with and without no_sync(), and when a parameter gets its first gradient on a
later step."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(6, 50)
        self.b = torch.nn.Linear(50, 50)
        self.dead = torch.nn.Linear(3, 3)     # never used -> no gradient (like the seg decoder's dispconvs)
        self.frozen = torch.nn.Linear(50, 4)
        for p in self.frozen.parameters():
            p.requires_grad = False
        self.c = torch.nn.Linear(50, 2)
        self.late = torch.nn.Linear(50, 2)    # only used from step 2 on (a loss branch that is switched on later)

    def forward(self, x, use_late=False):
        h = torch.relu(self.b(torch.relu(self.a(x))))
        y = self.c(h + 0 * self.frozen.weight.sum())
        if use_late:
            y = y + self.late(h)
        return y


def _worker_multi(rank, world, port, q, mode):
    """two backward() calls per step (loss split in two terms, second graph shares the first layers); mode:
    'nosync' = first backward under no_sync(), 'plain' = both in sync mode (dirty-bucket path), 'late' = single backward
    with a parameter that joins at step 2"""
    import warnings
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from improving_segmentation_with_selfsupervised_depth_amd.ddp import GradAllReducer
    torch.manual_seed(7)
    net = Net()
    red = GradAllReducer(net, bucket_mb=0.005)
    ref = Net()
    ref.load_state_dict(net.state_dict())
    torch.manual_seed(0)
    data, tgt = torch.randn(4, 8, 6), torch.randn(4, 8, 2)
    worst = 0.0
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for step in range(4):
            xs, ts = data[step].chunk(world)[rank], tgt[step].chunk(world)[rank]
            late = mode == "late" and step >= 2
            net.zero_grad(set_to_none=(step % 2 == 0))      # both zero_grad flavours
            ref.zero_grad(set_to_none=True)
            if mode == "late":
                ((net(xs, late) - ts) ** 2).mean().backward()
                ((ref(data[step], late) - tgt[step]) ** 2).mean().backward()
            else:
                out = net(xs)
                l1, l2 = ((out[:, 0] - ts[:, 0]) ** 2).mean(), (out[:, 1] - ts[:, 1]).abs().mean()
                if mode == "nosync":
                    with red.no_sync():
                        l1.backward(retain_graph=True)
                else:
                    l1.backward(retain_graph=True)
                l2.backward()
                o = ref(data[step])
                (((o[:, 0] - tgt[step][:, 0]) ** 2).mean() + (o[:, 1] - tgt[step][:, 1]).abs().mean()).backward()
            red.finish()
            for (k, p), (_, r) in zip(net.named_parameters(), ref.named_parameters()):
                if r.grad is None or (float(r.grad.abs().max()) == 0 and p.grad is None):
                    continue
                if p.grad is None:
                    worst = 1e9
                    continue
                worst = max(worst, float((p.grad - r.grad).abs().max() / (r.grad.abs().max() + 1e-12)))
            with torch.no_grad():
                for p, r in zip(net.parameters(), ref.parameters()):
                    if p.grad is not None and r.grad is not None:
                        p -= 0.1 * p.grad
                        r -= 0.1 * r.grad
    q.put((rank, worst, red.rebuilds))
    dist.destroy_process_group()


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from improving_segmentation_with_selfsupervised_depth_amd.ddp import GradAllReducer
    torch.manual_seed(100 + rank)           # different initial weights per rank: broadcast must fix that
    net = Net()
    red = GradAllReducer(net, bucket_mb=0.005)
    torch.manual_seed(0)
    data = torch.randn(3, 8, 6)             # 3 steps, global batch 8
    tgt = torch.randn(3, 8, 2)
    ref = Net()
    ref.load_state_dict(net.state_dict())
    ok = True
    for step in range(3):
        xs, ts = data[step].chunk(world)[rank], tgt[step].chunk(world)[rank]
        net.zero_grad(set_to_none=True)
        ((net(xs) - ts) ** 2).mean().backward()
        red.finish()
        ref.zero_grad(set_to_none=True)
        ((ref(data[step]) - tgt[step]) ** 2).mean().backward()
        for (k, p), (_, r) in zip(net.named_parameters(), ref.named_parameters()):
            if r.grad is None:
                ok = ok and (p.grad is None)
            else:
                ok = ok and p.grad is not None and torch.allclose(p.grad, r.grad, rtol=1e-5, atol=1e-6)
        with torch.no_grad():
            for p, r in zip(net.parameters(), ref.parameters()):
                if p.grad is not None:
                    p -= 0.1 * p.grad
                    r -= 0.1 * r.grad
    nb = len(red.buckets)
    q.put((rank, bool(ok), nb))
    dist.destroy_process_group()


def test_grad_allreducer_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(r[1] for r in res), res
    assert res[0][2] > 1, "expected several buckets"


@pytest.mark.parametrize("mode", ["nosync", "plain", "late"])
def test_grad_allreducer_multi_backward_and_late_params(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + {"nosync": 0, "plain": 1, "late": 2}[mode]
    procs = [ctx.Process(target=_worker_multi, args=(r, 2, port, q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert all(r[1] < 1e-5 for r in res), res
    if mode == "late":
        assert all(r[2] == 2 for r in res), res     # built on step 0, rebuilt once when `late` got its first gradient
