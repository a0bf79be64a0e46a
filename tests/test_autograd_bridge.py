"""The nn.Parameter / torch.autograd surface over the explicit engines (occm_amd/autograd_bridge.py), driven on the CPU by a toy
engine with the engines' interface (flat P / G, ref_views, zero_grad, explicit backward that ACCUMULATES into G): parameter aliasing,
the three gradient hand-over modes (fresh after zero_grad(set_to_none), accumulate, mixed) and the one-tape rule, all against plain
torch autograd of the same function.  The reference loop it serves: oc_training.py:320-328, 363-385."""
import pytest
import torch

from occm_amd.autograd_bridge import AliasGuard, attach_parameters, run_engine


class ToyEngine:
    """y = x @ W^T + b with W kept [in-major] inside (so the reference-shaped view is a transpose) -- explicit forward / backward."""

    def __init__(self, i=5, o=3):
        self.i, self.o = i, o
        self.P, self.G = torch.zeros(i * o + 4), torch.zeros(i * o + 4)
        g = torch.Generator().manual_seed(0)
        self.P[: i * o + o] = torch.randn(i * o + o, generator=g)
        self.ctx, self.grads_cleared = None, True

    def ref_views(self, flat):
        i, o = self.i, self.o
        return {"lin.weight": flat[: i * o].view(i, o).t(), "lin.bias": flat[i * o: i * o + o]}

    def zero_grad(self):
        self.G.zero_(); self.grads_cleared = True

    def forward(self, x):
        v = self.ref_views(self.P)
        self.ctx = x
        return x @ v["lin.weight"].t() + v["lin.bias"], (x * x).sum(1)

    def backward(self, dy, want_dx):
        x, g = self.ctx, self.ref_views(self.G)
        g["lin.weight"].add_(dy.t() @ x); g["lin.bias"].add_(dy.sum(0))
        self.ctx, self.grads_cleared = None, False
        return dy @ self.ref_views(self.P)["lin.weight"] if want_dx else None


class Toy(AliasGuard, torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.engine = ToyEngine()
        self.param_set = attach_parameters(self, self.engine)

    def forward(self, x):
        eng = self.engine

        def bwd(grads, needs):
            dy = grads[0] if grads[0] is not None else torch.zeros(x.shape[0], eng.o)
            return (eng.backward(dy, needs[0]),)

        return run_engine(self.param_set, eng.forward, bwd, x)


def _twin(m):
    lin = torch.nn.Linear(5, 3)
    with torch.no_grad():
        lin.weight.copy_(m.lin.weight); lin.bias.copy_(m.lin.bias)
    return lin


def test_parameters_carry_reference_names_shapes_and_alias_the_flat_buffer():
    m = Toy()
    names = dict(m.named_parameters())
    assert set(names) == {"lin.weight", "lin.bias"} and names["lin.weight"].shape == (3, 5)
    assert names["lin.weight"].data_ptr() == m.engine.P.data_ptr() and not names["lin.weight"].is_contiguous()
    with torch.no_grad():
        names["lin.bias"].add_(1.0)
    assert float(m.engine.P[15]) != 0 and m.engine.P._version > 0          # the version counter sync_operands() relies on
    m.to("cpu"); m.float()
    with pytest.raises(RuntimeError):
        m.double()


def test_reference_loop_equals_plain_autograd_over_three_adam_steps():
    m = Toy()
    ref = _twin(m)
    opt, ropt = torch.optim.Adam(m.parameters(), lr=1e-2), torch.optim.Adam(ref.parameters(), lr=1e-2)
    for step in range(3):
        x = torch.randn(7, 5, generator=torch.Generator().manual_seed(step))
        opt.zero_grad(); ropt.zero_grad()                                     # (set_to_none: every grad is None again)
        y, _ = m(x)
        loss = (y ** 2).mean(); loss.backward(); opt.step()
        rl = (ref(x) ** 2).mean(); rl.backward(); ropt.step()
        assert m.lin.weight.grad.data_ptr() == m.engine.G.data_ptr()          # .grad is the engine's buffer, not a copy
        torch.testing.assert_close(loss, rl)
    torch.testing.assert_close(m.lin.weight.detach(), ref.weight.detach()); torch.testing.assert_close(m.lin.bias.detach(), ref.bias.detach())


def test_gradients_accumulate_merge_and_reach_the_input():
    m = Toy()
    ref = _twin(m)
    x = torch.randn(4, 5, generator=torch.Generator().manual_seed(9), requires_grad=True)
    xr = x.detach().clone().requires_grad_(True)
    for _ in range(2):                                                        # second backward without zero_grad: accumulates
        m(x)[0].sum().backward(); ref(xr).sum().backward()
    torch.testing.assert_close(m.lin.weight.grad, ref.weight.grad); torch.testing.assert_close(x.grad, xr.grad)
    m.lin.bias.grad = None                                                    # mixed: one grad None, one aliasing, ...
    ref.bias.grad = None
    m(x)[0].sum().backward(); ref(xr).sum().backward()
    torch.testing.assert_close(m.lin.weight.grad, ref.weight.grad); torch.testing.assert_close(m.lin.bias.grad, ref.bias.grad)
    m.lin.weight.grad = m.lin.weight.grad.clone()                             # ... or a tensor of the caller's
    m(x)[0].sum().backward(); ref(xr).sum().backward()
    torch.testing.assert_close(m.lin.weight.grad, ref.weight.grad); torch.testing.assert_close(m.lin.bias.grad, ref.bias.grad)
    for p in m.parameters():                                                  # zero_grad(set_to_none=False) keeps the aliasing tensors
        p.grad = None
    m(x)[0].sum().backward()
    torch.optim.SGD(m.parameters(), lr=0.1).zero_grad(set_to_none=False)
    assert float(m.engine.G.abs().max()) == 0.0
    m(x)[0].sum().backward()
    torch.testing.assert_close(m.lin.bias.grad, torch.full((3,), 4.0))


def test_no_grad_eval_and_the_one_tape_rule():
    m = Toy()
    x = torch.randn(2, 5)
    with torch.no_grad():
        y, _ = m(x)
    assert not y.requires_grad
    y1, _ = m(x)
    y2, _ = m(x)                                                              # overwrites the engine's tape
    with pytest.raises(RuntimeError, match="tape"):
        y1.sum().backward()
    y2.sum().backward()
    m.lin.bias.requires_grad_(False); m.lin.bias.grad = None; m.lin.weight.grad = None
    m(x)[0].sum().backward()
    assert m.lin.bias.grad is None and m.lin.weight.grad is not None          # a frozen parameter gets no .grad, as in torch
    for p in m.parameters():
        p.requires_grad_(False)
    assert not m(x)[0].requires_grad                                          # nothing to differentiate: plain call
