"""The optimiser loop around the path (SURVEY 8.8 "next" row 3): `multires_opt` and its fused per-iteration tail.
The reference's loop (core/optimizer.py:44-84) is torch statements -- `n.grad[mask] = 0`, torch.optim.Adam.step(),
`n.clamp_(min=1)` -- so the pin for the fused HIP pass is those very statements run by torch on the same device."""
import os

import numpy as np
import pytest
import torch


def test_masked_adam_has_no_cpu_path_and_validates():
    from adjointnonlinearraytracing_amd import optimizer
    n = torch.zeros(4, 4, 4, requires_grad=True)
    o = optimizer.MaskedAdam([n])
    n.grad = torch.ones_like(n)
    with pytest.raises(RuntimeError, match="cuda"):
        o.step()
    for bad in (dict(lr=-1.0), dict(eps=-1e-8), dict(betas=(1.0, 0.999)), dict(betas=(0.9, -0.1)), dict(weight_decay=-1.0)):
        with pytest.raises(ValueError):
            optimizer.MaskedAdam([n], **bad)
    # the state layout is torch.optim.Adam's
    assert set(optimizer.MaskedAdam([n]).param_groups[0]) >= {"lr", "betas", "eps", "weight_decay"}


@pytest.mark.gpu
@pytest.mark.parametrize("shape,wd", [((20, 18, 16), 0.0), ((7, 9, 33), 0.01), ((2, 2, 2), 0.0)])
def test_masked_adam_equals_mask_adam_clamp_in_torch(gpu, shape, wd):
    """Six steps with fresh random gradients: parameter, both moments, the step counter and the masked gradient equal
    the reference's three statements run by torch (fp32 rounding only: 2e-6 relative)."""
    from adjointnonlinearraytracing_amd import optimizer
    g = torch.Generator().manual_seed(5)
    n0 = (1.0 + 0.4 * torch.rand(shape, generator=g)).to(gpu)
    a = n0.clone().requires_grad_(True)
    b = n0.clone().requires_grad_(True)
    oa = torch.optim.Adam([a], lr=3e-2, weight_decay=wd)
    ob = optimizer.MaskedAdam([b], lr=3e-2, weight_decay=wd)
    mask = torch.ones_like(a, dtype=torch.bool)
    mask[1:-1, 1:-1, 1:-1] = 0                                         # core/optimizer.py:54-55
    for k in range(6):
        grad = (torch.randn(shape, generator=g) * (10.0 ** (k - 3))).to(gpu)
        a.grad = grad.clone(); b.grad = grad.clone()
        with torch.no_grad():
            a.grad[mask] = 0                                           # :61
        oa.step()                                                      # :63
        with torch.no_grad():
            a.clamp_(min=1)                                            # :66
        ob.step()
        assert torch.equal(b.grad, a.grad)                             # masked in place, like the reference
        sa, sb = oa.state[a], ob.state[b]
        assert float(sa["step"]) == float(sb["step"]) == k + 1
        for x, y in ((a.detach(), b.detach()), (sa["exp_avg"], sb["exp_avg"]), (sa["exp_avg_sq"], sb["exp_avg_sq"])):
            # fp32 rounding of a different (fused-multiply-add) operation order: relative to the array's scale
            assert float((x - y).abs().max()) <= 3e-6 * float(x.abs().max()), (k, (x - y).abs().max(), x.abs().max())
        assert float(b.detach().min()) >= 1.0
    # options off: plain Adam
    c = n0.clone().requires_grad_(True); d = n0.clone().requires_grad_(True)
    oc = torch.optim.Adam([c], lr=1e-2); od = optimizer.MaskedAdam([d], lr=1e-2, mask_boundary=False, clamp_min=None)
    grad = torch.randn(shape, generator=g).to(gpu)
    c.grad = grad.clone(); d.grad = grad.clone()
    oc.step(); od.step()
    assert float((c - d).detach().abs().max()) <= 3e-6 * float(c.detach().abs().max())
    assert torch.equal(d.grad, grad)


@pytest.mark.gpu
def test_multires_opt_fused_equals_literal(gpu, tmp_path):
    """Two resolution levels of core/optimizer.py:44-84 on a cheap differentiable loss: the fused loop and the loop run
    with the reference's literal statements end at the same volume, the same loss history and the same checkpoint
    layout (the Adam moments travel through upres_scene / reload_opto on both paths)."""
    from adjointnonlinearraytracing_amd import optimizer
    g = torch.Generator().manual_seed(9)
    eta = torch.ones(8, 8, 8, device=gpu)
    targets = {r: (1.0 + 0.3 * torch.rand((r, r, r), generator=g)).to(gpu) for r in (8, 16)}
    seen = []

    def func(n):
        t = targets[n.shape[0]]
        return ((n - t) ** 2).sum() + 0.1 * (n[1:] - n[:-1]).abs().sum()

    out = {}
    for fused in (True, False):
        path = str(tmp_path / f"state_{int(fused)}.pt")
        n, hist = optimizer.multires_opt(func, eta, 3, [8, 16], log_func=lambda i, n_: seen.append(i), lr=2e-2,
                                         statename=path, fused=fused)
        ck = torch.load(path, weights_only=True)
        assert set(ck) == {"rif", "opto_state_dict", "loss_hist"}
        st = ck["opto_state_dict"]["state"][0]
        assert set(st) >= {"step", "exp_avg", "exp_avg_sq"} and float(st["step"]) == 3 + 6      # carried across levels
        assert st["exp_avg"].shape == (16, 16, 16)
        out[fused] = (n.detach().cpu(), hist)
    assert seen == list(range(9)) * 2
    assert out[True][0].shape == (16, 16, 16) and len(out[True][1]) == 9
    assert torch.allclose(out[True][0], out[False][0], rtol=1e-5, atol=1e-6)
    assert np.allclose(out[True][1], out[False][1], rtol=1e-5)
    assert float(out[True][0].min()) >= 1.0


@pytest.mark.gpu
def test_masked_adam_step_is_visible_to_autograd(gpu):
    """MaskedAdam writes the parameter through a raw pointer; it must bump the tensor's version counter like any in-place
    torch op, because the tracer's save_for_backward check and the pair-copy reuse token key on it: a retained-graph
    backward AFTER step() has to raise (as it does with torch.optim.Adam) instead of pairing the new grid with the exit
    rays -- and the pair copy -- of the old one."""
    from adjointnonlinearraytracing_amd import optimizer, tracer
    R = 17
    h = 1.0 / (R - 1); ds = h / 2
    n = (1.0 + 0.1 * torch.rand(R, R, R, device=gpu)).requires_grad_(True)
    x = torch.rand(256, 3, device=gpu) * 0.8 + 0.1
    x[:, 1] = -0.3 * ds
    v = torch.zeros(256, 3, device=gpu); v[:, 1] = 1.0
    opto = optimizer.MaskedAdam([n], lr=1e-2)
    xt, vt = tracer.BackTracerC.apply(n, x, v, h, ds)
    loss = (xt ** 2).sum()
    ver = n._version
    loss.backward(retain_graph=True)
    opto.step()
    assert n._version > ver
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        loss.backward()
