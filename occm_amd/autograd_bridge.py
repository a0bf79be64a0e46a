"""torch.autograd / nn.Parameter surface over the explicit forward/backward engines.

The reference trains with plain PyTorch idiom (oc_training.py:320-328, 363-385; test_dataloader_v2.py:107-130)::

    optimizer = optim.Adam(aasist.parameters(), lr=0.00001)
    ...
    optimizer.zero_grad(); com, des = aasist(inputs); loss = ...; loss.backward(); optimizer.step()

The engines here (AasistBackend, SeResNet34Backend, LcnnBackend, XlsrFineTuner) keep every parameter in ONE flat f32 buffer ``P`` with
a flat gradient buffer ``G`` and run hand-written backward passes.  This module makes that drivable by the loop above:

* ``attach_parameters`` registers one ``nn.Parameter`` per reference tensor, under the reference's dotted name and with the reference's
  shape, as a VIEW of ``P`` (permuted / sliced where the kernels' layout differs: channels-last conv weights, packed attention weights,
  fused q|k|v) -- ``optim.Adam(model.parameters())`` therefore updates the very memory the kernels read;
* ``EngineFunction`` is the ``torch.autograd.Function`` whose ``backward`` calls the engine's explicit backward.  Parameter gradients are
  not returned through autograd: the engine accumulates into ``G`` and every ``param.grad`` is (re-)pointed at its view of ``G``
  (``optimizer.zero_grad()`` sets grads to None by default; a None grad means "start from zero", an aliasing grad means "accumulate",
  anything else is added to).

``occm_amd.trainer.OcTrainer`` stays the fused fast path (one Adam launch over the flat buffers, HIP-graph replay, overlapped all-reduce);
both drive the same kernels through the same C ABI.
"""
import torch


class ParamHolder(torch.nn.Module):
    """Empty container: exists so that ``named_parameters()`` yields the reference's dotted names (``encoder.0.0.conv1.weight``)."""


def _holder_for(root, dotted):
    """Walk / create the chain of holder sub-modules for ``a.b.c.weight`` under ``root``; returns (module, leaf name)."""
    parts = dotted.split(".")
    mod = root
    for part in parts[:-1]:
        nxt = mod._modules.get(part)
        if nxt is None:
            nxt = ParamHolder()
            mod.add_module(part, nxt)
        mod = nxt
    return mod, parts[-1]


class ParamSet:
    """The (parameter, gradient view) pairs of one engine + the bookkeeping ``EngineFunction`` needs."""

    def __init__(self, engine):
        self.engine = engine
        self.params, self.gviews, self.names = [], [], []
        self.tape_id = 0                      # incremented per taped forward: a backward of an overwritten tape is refused

    def add(self, name, param, gview):
        self.names.append(name); self.params.append(param); self.gviews.append(gview)

    # ------------------------------------------------------------------------------------------------ gradients --
    def _aliases(self, p, g):
        gr = p.grad
        return gr is not None and gr.data_ptr() == g.data_ptr() and gr.shape == g.shape and gr.stride() == g.stride() and gr.dtype == g.dtype

    def before_backward(self):
        """Decide how this backward's gradients meet what ``param.grad`` holds.  Returns a token for ``after_backward``."""
        eng = self.engine
        live = [(p, g) for p, g in zip(self.params, self.gviews) if p.requires_grad]      # frozen parameters never receive a .grad (as in torch)
        if all(p.grad is None for p, _ in live):          # after optimizer.zero_grad(): start from zero
            eng.zero_grad()
            return ("fresh", None)
        if all(self._aliases(p, g) for p, g in live):
            if hasattr(eng, "grads_cleared"):
                eng.grads_cleared = False     # the buffer holds gradients the engine did not see being cleared: accumulate, never store
            return ("accumulate", None)
        saved = eng.G.clone()                 # some grads are None / foreign tensors: compute into a zeroed buffer, merge afterwards
        eng.zero_grad()
        return ("mixed", saved)

    def after_backward(self, token):
        mode, saved = token
        if mode == "accumulate":
            return
        if mode == "fresh":
            for p, g in zip(self.params, self.gviews):
                if p.requires_grad:
                    p.grad = g
            return
        sv = self.engine.ref_views(saved)
        for name, p, g in zip(self.names, self.params, self.gviews):
            if not p.requires_grad:
                continue
            if p.grad is None:
                p.grad = g
            elif self._aliases(p, g):
                g.add_(sv[name])              # G was cleared for this pass: bring the earlier gradient back
            else:
                p.grad.add_(g)


def attach_parameters(root, engine, prefix=""):
    """Register the engine's tensors on ``root`` as nn.Parameters named like the reference's (``engine.ref_views`` gives
    {reference name: view in the reference shape} of a flat buffer).  Returns the ParamSet."""
    ps = ParamSet(engine)
    pv, gv = engine.ref_views(engine.P), engine.ref_views(engine.G)
    for name, view in pv.items():
        mod, leaf = _holder_for(root, prefix + name)
        param = torch.nn.Parameter(view, requires_grad=True)
        if param.data_ptr() != view.data_ptr():
            raise RuntimeError("nn.Parameter copied %s instead of aliasing the engine buffer" % name)
        mod.register_parameter(leaf, param)
        ps.add(name, param, gv[name])
    engine.param_set = ps
    return ps


class EngineFunction(torch.autograd.Function):
    """forward(run_fwd, run_bwd, pset, n_in, *inputs, *params): ``run_fwd(*inputs) -> tuple of tensors`` runs the engine's taped
    forward; ``run_bwd(grad_outputs, needs_input_grad) -> tuple of input gradients (or None)`` runs its explicit backward, which
    accumulates parameter gradients into the engine's flat G.  The parameters are inputs only so that autograd knows the outputs
    depend on them; their gradients travel through ``ParamSet`` (see the module docstring)."""

    @staticmethod
    def forward(ctx, run_fwd, run_bwd, pset, n_in, *args):
        outs = run_fwd(*args[:n_in])
        pset.tape_id += 1
        ctx.run_bwd, ctx.pset, ctx.n_in, ctx.n_args, ctx.tape = run_bwd, pset, n_in, len(args), pset.tape_id
        return tuple(outs) if isinstance(outs, (tuple, list)) else outs

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, *grads):
        ps = ctx.pset
        if ctx.tape != ps.tape_id:
            raise RuntimeError("occm_amd: this forward's tape was overwritten by a later forward of the same model (the engines keep "
                               "one tape: call backward() before the next training-mode forward)")
        token = ps.before_backward()
        din = ctx.run_bwd(grads, ctx.needs_input_grad[4:4 + ctx.n_in])
        ps.after_backward(token)
        ps.tape_id += 1                       # the tape is consumed
        din = tuple(din) if din is not None else (None,) * ctx.n_in
        return (None, None, None, None) + din + (None,) * (ctx.n_args - ctx.n_in)


def run_engine(pset, run_fwd, run_bwd, *inputs):
    """Taped forward through autograd when gradients are being recorded and anything can receive one; plain call otherwise."""
    live = [p for p in pset.params if p.requires_grad]
    if torch.is_grad_enabled() and (live or any(torch.is_tensor(x) and x.requires_grad for x in inputs)):
        return EngineFunction.apply(run_fwd, run_bwd, pset, len(inputs), *inputs, *pset.params)
    with torch.no_grad():
        return run_fwd(*inputs)


class AliasGuard:
    """Mixin for the facade modules: ``.to()`` / ``.cuda()`` / ``.float()`` must not move the Parameters off the engine's buffers."""

    def _apply(self, fn, recurse=True):
        before = [(p, p.data_ptr()) for p in self.parameters()]
        out = super()._apply(fn, recurse)
        for p, ptr in before:
            if p.data_ptr() != ptr:
                raise RuntimeError("occm_amd modules live on the GPU buffers their HIP engines read (f32, cuda): "
                                   ".to()/.half()/.cpu() that would move or cast the parameters is not supported")
        return out
