"""planerecnet_amd.optim.FusedAdam (one launch over all parameter tensors) against torch.optim.Adam -- the optimizer of the
reference's train.py:251-256 -- over several steps: per-group learning rates, a learning-rate change (warm-up, train.py:320-323), a
skipped update (`found_inf`, train.py:353), a parameter without a gradient, odd tensor sizes."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    shapes = [(64, 3, 7, 7), (27,), (256, 64, 1, 1), (5,), (1,), (128, 128, 3, 3), (4099,), (2, 3, 5, 7)]
    return [torch.randn(*s, generator=g).cuda().requires_grad_(True) for s in shapes]


def test_fused_adam_matches_torch_adam():
    from planerecnet_amd.optim import FusedAdam
    pa, pb = _params(0), _params(0)
    groups = lambda ps: [{"params": ps[:3], "lr": 5e-3}, {"params": ps[3:6], "lr": 1e-3}, {"params": ps[6:], "lr": 2e-3}]      # noqa: E731
    ref = torch.optim.Adam(groups(pa), lr=1e-3, fused=True)
    mine = FusedAdam(groups(pb), lr=1e-3)
    assert getattr(mine, "_step_supports_amp_scaling", False)
    g = torch.Generator().manual_seed(1)
    for it in range(7):
        for a, b in zip(pa, pb):
            gr = torch.randn(a.shape, generator=g).cuda() * (10.0 ** (it % 3 - 1))
            a.grad, b.grad = gr.clone(), gr.clone()
        pa[4].grad = pb[4].grad = None                              # a parameter that never receives a gradient is left alone
        if it == 4:                                                 # learning-rate change between steps (warm-up / decay)
            for o in (ref, mine):
                for grp in o.param_groups:
                    grp["lr"] *= 0.5
        skip = torch.tensor(1.0 if it == 2 else 0.0, device="cuda")
        ref.found_inf = mine.found_inf = skip                       # the device-side "do not update" flag of a GradScaler
        ref.grad_scale = mine.grad_scale = None
        before = [b.detach().clone() for b in pb]
        versions = [b._version for b in pb]
        ref.step()
        mine.step()
        # parameters that were stepped look modified to torch (weight caches keyed on _version), the one without a gradient does not
        assert all((b._version > v) == (i != 4) for i, (b, v) in enumerate(zip(pb, versions)))
        if it == 2:
            assert all(torch.equal(x, y.detach()) for x, y in zip(before, pb))
        for i, (a, b) in enumerate(zip(pa, pb)):
            assert torch.allclose(a, b, rtol=2e-6, atol=1e-7), (it, i, float((a - b).abs().max()))
    for i, (a, b) in enumerate(zip(pa, pb)):
        if i == 4:
            assert "exp_avg" not in mine.state.get(b, {})
            continue
        sa, sb = ref.state[a], mine.state[b]
        # (moments of O(1)..O(100) gradients; exp_avg has cancelling terms, hence the absolute part)
        assert torch.allclose(sa["exp_avg"], sb["exp_avg"], rtol=2e-6, atol=2e-6) and torch.allclose(sa["exp_avg_sq"], sb["exp_avg_sq"], rtol=2e-6, atol=1e-9)
        assert float(sa["step"]) == float(sb["step"]) == 6.0        # seven calls, one skipped
    sd = mine.state_dict()
    assert len(sd["param_groups"]) == 3 and len(sd["state"]) == 7


def test_fused_adam_runs_ahead_of_the_device():
    """Many steps enqueued behind a parked stream: the page-locked staging buffers of the gradient-pointer table must not be
    rewritten before their copies have executed (each step's gradients live at different addresses)."""
    from planerecnet_amd.optim import FusedAdam
    pa, pb = _params(2), _params(2)
    ref, mine = torch.optim.Adam(pa, lr=1e-2, fused=True), FusedAdam(pb, lr=1e-2)
    g = torch.Generator().manual_seed(3)
    grads = [[torch.randn(a.shape, generator=g).cuda() for a in pa] for _ in range(10)]
    torch.cuda.synchronize()
    torch.cuda._sleep(int(0.05 * 2.4e9))                            # ~50 ms: the host enqueues all ten steps meanwhile
    for it in range(10):
        for a, b, gr in zip(pa, pb, grads[it]):
            a.grad, b.grad = gr, gr.clone()                         # fresh allocation per step for the device under test
        ref.step()
        mine.step()
    torch.cuda.synchronize()
    for a, b in zip(pa, pb):
        assert torch.allclose(a, b, rtol=5e-6, atol=1e-7)


def test_fused_adam_survives_state_reload_and_storage_moves():
    """The launch tables hold raw pointers: after load_state_dict (new moment tensors, `step` possibly a host tensor) and after a
    parameter's storage was replaced (p.data = ...) the next step must work on the live tensors, like torch.optim.Adam does."""
    from planerecnet_amd.optim import FusedAdam
    pa, pb = _params(5), _params(5)
    ref, mine = torch.optim.Adam(pa, lr=1e-2), FusedAdam(pb, lr=1e-2)
    g = torch.Generator().manual_seed(6)

    def both_step():
        for a, b in zip(pa, pb):
            gr = torch.randn(a.shape, generator=g).cuda()
            a.grad, b.grad = gr.clone(), gr.clone()
        ref.step()
        mine.step()

    both_step(); both_step()
    sd_ref, sd_mine = ref.state_dict(), mine.state_dict()
    for st in sd_mine["state"].values():                            # a checkpoint that kept the counter on the host
        st["step"] = st["step"].cpu()
    ref.load_state_dict(sd_ref)
    mine.load_state_dict(sd_mine)
    both_step()
    for a, b in zip(pa, pb):                                        # storage replaced behind the optimizer's back
        a.data = a.data.clone()
        b.data = b.data.clone()
    both_step()
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(pa, pb)):
        assert torch.allclose(a, b, rtol=5e-6, atol=1e-7), (i, float((a - b).abs().max()))
        assert float(mine.state[b]["step"]) == 4.0 and mine.state[b]["step"].is_cuda


def test_fused_adam_skips_tensors_no_rank_had_a_gradient_for():
    """Data-parallel runs: the exchange zero-fills a gradient this rank did not produce; when NO rank produced one (all-reduced
    presence count 0) the tensor and its moments must stay untouched, like optim.Adam with .grad = None (reference train.py:362)."""
    from planerecnet_amd.optim import FusedAdam
    pa, pb = _params(7), _params(7)
    ref, mine = torch.optim.Adam(pa, lr=1e-2), FusedAdam(pb, lr=1e-2)

    class Exchange:                                               # the two attributes FusedAdam reads from parallel.GradAllReduce
        active = True
        index = {p: len(pb) - 1 - i for i, p in enumerate(pb)}    # (a different order than the optimizer's)
        presence = torch.full((len(pb),), 2.0, device="cuda")
    mine.exchange = Exchange
    g = torch.Generator().manual_seed(8)
    for it in range(3):
        absent = {2, 5} if it == 1 else set()
        Exchange.presence.fill_(2.0)
        for i, (a, b) in enumerate(zip(pa, pb)):
            gr = torch.randn(a.shape, generator=g).cuda()
            if i in absent:
                a.grad, b.grad = None, torch.zeros_like(b)         # what the exchange leaves behind
                Exchange.presence[Exchange.index[b]] = 0.0
            else:
                a.grad, b.grad = gr.clone(), gr.clone()
        ref.step()
        mine.step()
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(zip(pa, pb)):
        assert torch.allclose(a, b, rtol=5e-6, atol=1e-7), i
        assert torch.allclose(ref.state[a]["exp_avg_sq"], mine.state[b]["exp_avg_sq"], rtol=5e-6, atol=1e-9), i


def test_fused_adam_joins_the_deferred_weight_gradients_itself():
    """With ops.set_wgrad_async(True) the convolution weight gradients are launched on the side stream and written to `.grad` there; the contract is
    ops.wgrad_join() after backward().  A loop that forgets it must still step with every gradient: FusedAdam.step joins (idempotent).  Two steps of a small
    conv stack with and without the explicit join end in bit-identical parameters."""
    from planerecnet_amd import ops
    from planerecnet_amd.optim import FusedAdam
    d = torch.device("cuda:0")

    def run(explicit_join):
        torch.manual_seed(3)
        ws = [torch.nn.Parameter(torch.randn(32, 32, 3, 3, device=d) * 0.05) for _ in range(3)]
        opt = FusedAdam(ws, lr=1e-2)
        x = torch.randn(2, 32, 20, 24, generator=torch.Generator().manual_seed(4)).to(d)
        ops.set_wgrad_async(True)
        try:
            for _ in range(2):
                opt.zero_grad()
                h = x
                for w in ws:
                    h = ops.conv2d(h, w, None, 1, 1)
                h.square().mean().backward()
                if explicit_join:
                    ops.wgrad_join()
                opt.step()
        finally:
            ops.wgrad_join()
            ops.set_wgrad_async(False)
        torch.cuda.synchronize()
        return [w.detach().clone() for w in ws]

    a, b = run(True), run(False)
    assert all(torch.equal(u, v) for u, v in zip(a, b))
    assert all(torch.isfinite(u).all() for u in a)
