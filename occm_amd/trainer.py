"""One training step of the hot path: (RawBoost ->) XLS-R front-end -> AASIST or SE-ResNet34 back-end -> losses ->
backward -> gradient all-reduce -> Adam.  Mirrors the loop bodies of oc_training.py:363-385 and
test_dataloader_v2.py:107-130."""
import torch

from . import ops
from .parallel import FlatGradAllReducer


class OcTrainer:
    """model: occm_amd.models.sslassist.AModel or occm_amd.models.senet.ssl_resnet34 (anything with ``.ssl_model.model`` and a
    ``.backend`` engine offering forward/backward/zero_grad and flat P/G buffers).  Loss weights default to the committed
    0.0 / 1.0 (oc_training.py:380-381); the SE-ResNet script uses 0.1 / 0.9 (test_dataloader_v2.py:127)."""

    def __init__(self, model, lr=1e-5, w_compact=0.0, w_descr=1.0, train_frontend=False, group_size=None, dropout_masks=None, rawboost_algo=0,
                 rawboost_args=None, seed=0, rank=0, graph_backend=True, grad_wire_dtype=None, graph_frontend=None):
        """graph_backend: replay the back-end section of a step (zero_grad, forward, losses, backward: ~560 small launches for AASIST) from a
        HIP graph once a batch shape has come twice in a row (fixed-length training); until then, and for every other shape, steps run eagerly.  Needs device-drawn dropout masks
        (``dropout_masks=None``): injected masks run eagerly."""
        self.model = model
        self.graph_backend = graph_backend
        import os as _os
        # off unless asked for (graph_frontend=True / OCC_FE_GRAPH=1): measured on one MI355X the replayed front-end is no faster than eager
        # launches (configs[2] 50.62 vs 50.38 ms per step, configs[4]'s shard 86.66 vs 86.39): the step is GPU-bound, see _step_finetune
        self.graph_frontend = (_os.environ.get("OCC_FE_GRAPH", "0") == "1") if graph_frontend is None else bool(graph_frontend)
        self._fe_graphs, self._fe_last_key = {}, None
        self._graphs, self._last_key = {}, None
        # every data-parallel rank draws its own augmentation parameters: the rank is part of the RawBoost seed
        self.rawboost_algo, self.rawboost_args, self.seed, self.nstep = rawboost_algo, rawboost_args, seed * 4096 + rank, 0
        self.dropout_masks = dropout_masks      # None: draw masks on the device (normal training); {}: no dropout; dict: injected keep-masks
        self.be = model.backend
        self.fe = model.ssl_model.model
        # ... and its own dropout / layerdrop streams: DataParallel replicas draw independent masks (each replica runs its own forward),
        # so two ranks must not share keep-masks.  Rank 0 with seed 0 keeps the streams of a single-GPU run.
        if hasattr(self.be, "rng_seed"):
            self.be.rng_seed = int(self.be.rng_seed) + self.seed
        if hasattr(self.fe, "drop_seed"):
            self.fe.drop_seed = int(self.fe.drop_seed) + self.seed
            self.fe.seed_layerdrop(self.seed)
        self.train_frontend = train_frontend
        if train_frontend and not hasattr(self.fe, "forward_train"):
            raise ValueError("train_frontend=True needs a model built with finetune_ssl=True")
        self.w_c, self.w_d = w_compact, w_descr
        self.group_size = group_size
        params, self._grads, mirrors = [self.be.P], [self.be.G], [None]
        flat = [self.be.G]
        self._chunk_of = {}
        if train_frontend:                     # XLS-R is trained: its bf16 GEMM operands are written by the optimizer kernel itself
            fe = self.fe
            flat.append(fe.G)
            Wb = getattr(fe, "Wb", None)
            if hasattr(fe, "layer_grad_range") and Wb is not None:
                # one optimizer "tensor" per transformer layer (its twelve tensors are one contiguous slice of the flat buffers) + the rest:
                # a layer is updated on a side stream as soon as its gradients are final, under the remaining backward pass
                cuts = [fe.layer_grad_range(i) for i in range(fe.cfg.layers)]
                assert cuts[0][0] == 0 and all(cuts[i][1] == cuts[i + 1][0] for i in range(len(cuts) - 1))
                cuts.append((cuts[-1][1], fe.P.numel()))
                for lo, hi in cuts:
                    self._chunk_of[(lo, hi)] = len(params)
                    params.append(fe.P[lo:hi]); self._grads.append(fe.G[lo:hi]); mirrors.append(Wb[lo:hi])
            else:
                params.append(fe.P); self._grads.append(fe.G); mirrors.append(Wb)
        self.opt = ops.AdamMulti(params, lr=lr, bf16_copies=mirrors)
        import os
        # Off by default: measured on one MI355X (bench.py, configs[2]) the per-layer Adam launches on a second stream slowed the GEMMs they
        # ran beside by as much as they hid (51.70 vs 51.68 / 52.38 ms per step, GEMM family 32.0 -> 33.4 ms); OCC_OPT_OVERLAP=1 enables it.
        self.overlap_optimizer = os.environ.get("OCC_OPT_OVERLAP", "0") == "1"
        self._opt_stream = None
        # grad_wire_dtype=torch.bfloat16 (or OCC_GRAD_WIRE=bf16): the XLS-R gradients (1.26 GB f32) cross xGMI as bf16; the small back-end
        # buffer stays f32
        if grad_wire_dtype is None and os.environ.get("OCC_GRAD_WIRE", "").lower() in ("bf16", "bfloat16"):
            grad_wire_dtype = torch.bfloat16
        self.reducers = [FlatGradAllReducer(g, wire_dtype=grad_wire_dtype if i == 1 else None) for i, g in enumerate(flat)]
        self.reducer = self.reducers[0]
        self.last = None
        # frozen front-end: features of the NEXT batch can be computed on a side stream while the back-end trains on this one
        self._side = None
        self._pref = None              # (wav tensor, features, ready event) of the prefetched batch
        self._pref_aug = None          # fine-tuning: (wav tensor, augmented wav, ready event) of the next batch
        self._fbuf = [None, None]

    def _augment(self, wav, step_idx):
        if not self.rawboost_algo:
            return wav
        # on-GPU RawBoost (data_utils_SSL.py:111-173; the call the reference leaves commented at oc_training.py:221)
        from .RawBoost import rawboost_batch_device
        from .oc_training import rawboost_args
        return rawboost_batch_device(wav, self.rawboost_args or rawboost_args(), self.rawboost_algo, seed=self.seed, step=step_idx)

    def _prefetch(self, next_wav, step_idx):
        """Data stage of the next step (RawBoost + frozen front-end) on a side stream, concurrent with this step's back-end work.
        The frozen front-end does not depend on the optimizer update, so results are identical to the sequential order."""
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream(device=next_wav.device)
        side = self._side
        side.wait_stream(main)                      # inputs produced on the main stream; the feature buffer's previous reader has been enqueued
        par = step_idx & 1
        with torch.cuda.stream(side):
            f = self.model.ssl_model.model.forward(self._augment(next_wav, step_idx), out_dtype=torch.float32)
            if self._fbuf[par] is None or self._fbuf[par].shape != f.shape:
                self._fbuf[par] = torch.empty_like(f)
            self._fbuf[par].copy_(f)
            ready = torch.cuda.Event()
            ready.record(side)
        self._pref = (next_wav, self._fbuf[par], ready)

    def step(self, wav, labels, next_wav=None):
        """wav f32 [B,L] cuda, labels i64 [B] cuda.  Returns device tensors (loss_c, loss_d); no host sync.
        next_wav: the batch of the following step() call (the same tensor object must then be passed as ``wav``); with a frozen
        front-end its RawBoost + XLS-R features are computed concurrently with this step's back-end forward/backward/Adam."""
        be = self.be
        raw_wav = wav
        if self.train_frontend:
            # fine-tuning: only the augmentation of the next batch can run ahead (its features depend on this step's update)
            if self._pref_aug is not None and self._pref_aug[0] is raw_wav:
                _, wav, ready = self._pref_aug
                torch.cuda.current_stream().wait_event(ready)
            else:
                wav = self._augment(wav, self.nstep)
            self._pref_aug = None
            step_idx = self.nstep
            self.nstep += 1
            return self._step_finetune(wav, labels, next_wav if self.rawboost_algo else None, step_idx + 1)
        if not (self._pref is not None and self._pref[0] is raw_wav):
            wav = self._augment(wav, self.nstep)
        step_idx = self.nstep
        self.nstep += 1
        if self._pref is not None and self._pref[0] is raw_wav:
            _, feats, ready = self._pref
            torch.cuda.current_stream().wait_event(ready)
        else:
            feats = self.model.ssl_model.model.forward(wav, out_dtype=torch.float32)
            if next_wav is not None:                # the front-end may return a buffer it reuses on every call: keep this step's copy
                par = step_idx & 1
                if self._fbuf[par] is None or self._fbuf[par].shape != feats.shape:
                    self._fbuf[par] = torch.empty_like(feats)
                self._fbuf[par].copy_(feats)
                feats = self._fbuf[par]
        self._pref = None
        if next_wav is not None:
            self._prefetch(next_wav, step_idx + 1)
        lc, ld, _ = self._backend_section(feats, labels, False)
        self.reducer.all_reduce()
        self.opt.step(self._grads, grad_scale=self.reducer.grad_scale)
        self.last = (lc, ld)
        return lc, ld

    def _backend_eager(self, feats, labels, want_dfeats):
        be = self.be
        be.zero_grad()
        emb, logits = be.forward(feats, train=True, masks=self.dropout_masks)
        B = emb.shape[0]
        ng = 1 if not self.group_size else B // self.group_size
        lc, demb = ops.compactness_loss(emb, n_groups=ng, group=self.group_size or B, scale=self.w_c, want_grad=True)
        ld, dlog = ops.ce_loss(logits, labels, scale=self.w_d, want_grad=True)
        dfeats = be.backward(demb, dlog, want_dfeats=True) if want_dfeats else be.backward(demb, dlog)
        return lc, ld, dfeats

    def _backend_section(self, feats, labels, want_dfeats):
        """zero_grad -> back-end forward -> losses -> back-end backward, eagerly or as one HIP-graph replay (same kernels, same order,
        same arithmetic; the dropout masks come from a device-side step counter that the graph itself advances)."""
        if not (self.graph_backend and self.dropout_masks is None and getattr(self.be, "graph_safe", False)):
            return self._backend_eager(feats, labels, want_dfeats)
        key = (tuple(feats.shape), str(feats.dtype), bool(want_dfeats))
        rec = self._graphs.get(key)
        if rec is not None and rec["graph"] is not None:
            rec["feats"].copy_(feats); rec["labels"].copy_(labels)
            rec["graph"].replay()
            self._last_key = key
            lc, ld, dfe = rec["out"]
            return lc.clone(), ld.clone(), dfe               # (the graph rewrites its own loss scalars on the next replay: hand out copies)
        out = self._backend_eager(feats, labels, want_dfeats)
        repeat = self._last_key == key
        self._last_key = key
        if rec is None or not repeat:
            # capture only when a shape comes twice in a row (fixed-length training): variable-length groups would pay a capture
            # per step for graphs that are evicted before they are replayed
            self._graphs[key] = {"graph": None}
            while len(self._graphs) > 4:
                self._graphs.pop(next(iter(self._graphs)))
            return out
        # this shape again, back to back: capture the section for the following steps (capturing enqueues nothing)
        rec["feats"], rec["labels"] = feats.clone(), labels.clone()
        rec["keep"] = dict(getattr(self.be, "_ws", {}))     # the persistent (bordered) buffers the graph writes to must outlive the engine's cache
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        step_before = getattr(self.be, "rng_step", 0)
        with torch.cuda.graph(g, capture_error_mode="thread_local"):
            rec["out"] = self._backend_eager(rec["feats"], rec["labels"], want_dfeats)
        if hasattr(self.be, "rng_step"):
            self.be.rng_step = step_before               # the host mirror counts executed steps, not captured ones
        rec["graph"] = g
        return out

    def _prefetch_augment(self, next_wav, step_idx):
        """RawBoost of the next batch (f64 FIR bank + noise: ~0.9 ms of VALU work at bs 64) on a side stream, started once the front-end's
        forward has been enqueued so that it runs under the back-end section, whose small kernels leave most of the chip idle.  Same
        (seed, step) as the sequential order: identical samples."""
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream(device=next_wav.device)
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            aug = self._augment(next_wav, step_idx)
            ready = torch.cuda.Event()
            ready.record(self._side)
        aug.record_stream(main)                            # allocated on the side stream, consumed on the main one
        self._pref_aug = (next_wav, aug, ready)

    def _frontend_graphable(self):
        """The fine-tuning front-end can be replayed from HIP graphs when nothing in it depends on per-step host decisions: one rank (the
        per-layer all-reduce hooks are host callbacks), no active fairseq dropout / layerdrop (their Philox stream ids and keep decisions
        are host values), fp8 scales past their first (measuring) step, no per-launch profiling."""
        fe = self.fe
        if not self.graph_frontend or ops.PROFILE is not None or self.overlap_optimizer or any(r.world > 1 for r in self.reducers):
            return False
        tc = getattr(fe, "train_cfg", None)
        if tc is not None and getattr(fe, "dropout_active", True) and tc.any_dropout():
            return False
        if getattr(fe, "fp8", False) and fe.f8["warm"]:
            return False
        return getattr(fe, "inject_masks", None) is None and getattr(fe, "inject_keep", None) is None

    def _step_finetune(self, wav, labels, next_wav=None, next_step=0):
        """Fine-tuning step.  A step enqueues ~1300 launches (2500 for XLS-R-1B) through ctypes: measured 46.6 ms of host time per step
        against 50.6 ms of wall time at configs[2], 79.9 against 87.1 ms on configs[4]'s shard -- close enough to host-bound to check.  With
        graph_frontend the front-end forward and {front-end backward, Adam, operand refresh} are each replayed from a HIP graph once a batch
        shape has come twice in a row (as the back-end section is); the glue between them (the gradient clear on the side stream, RawBoost
        of the next batch, the back-end section's own graph) stays eager.  Same kernels, same order, same arithmetic -- and, measured, the
        same step time: the host runs ahead of the GPU either way, so this stays an option (it frees the host for the data loader)."""
        fe = self.fe
        key = (tuple(wav.shape), bool(getattr(fe, "fp8", False)))
        ok = self._frontend_graphable()
        rec = self._fe_graphs.get(key) if ok else None
        if rec is not None and rec.get("fwd") is not None:
            main = torch.cuda.current_stream()
            if self._side is None:
                self._side = torch.cuda.Stream(device=wav.device)
            rec["wav"].copy_(wav)
            self._side.wait_stream(main)
            with torch.cuda.stream(self._side):
                fe.zero_grad()
                cleared = torch.cuda.Event()
                cleared.record(self._side)
            rec["fwd"].replay()
            if next_wav is not None:
                self._prefetch_augment(next_wav, next_step)
            main.wait_event(cleared)
            lc, ld, dfeats = self._backend_section(rec["feats"], labels, True)
            rec["dfeats"].copy_(dfeats.view_as(rec["dfeats"]))
            rec["bwd"].replay()
            self._fe_last_key = key
            self.last = (lc, ld)
            return lc, ld
        out = self._step_finetune_eager(wav, labels, next_wav, next_step)
        repeat = ok and self._fe_last_key == key
        self._fe_last_key = key if ok else None
        if repeat and self._frontend_graphable():               # (fp8: the step just taken may have been the measuring one)
            try:
                self._capture_frontend(key, wav)
            except Exception as e:                              # capture is an optimisation: never let it break training
                self._fe_graphs.pop(key, None)
                self.graph_frontend = False
                import warnings
                warnings.warn("occm_amd: front-end HIP-graph capture failed (%s); continuing with eager launches" % (e,))
        return out

    def _capture_frontend(self, key, wav):
        fe = self.fe
        rec = {"wav": wav.clone()}
        torch.cuda.synchronize()
        g1 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1, capture_error_mode="thread_local"):
            rec["feats"] = fe.forward_train(rec["wav"])
        rec["dfeats"] = torch.zeros_like(rec["feats"])
        g2 = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g2, capture_error_mode="thread_local"):
            fe.backward(rec["dfeats"])
            self.opt.step(self._grads, grad_scale=self.reducer.grad_scale)
            fe.refresh_operands(cast=not hasattr(fe, "Wb"))
        self.opt.step_count -= 1                                 # capturing enqueued nothing: the host-side step counter counts executed steps
        rec["fwd"], rec["bwd"] = g1, g2
        self._fe_graphs[key] = rec
        while len(self._fe_graphs) > 2:
            self._fe_graphs.pop(next(iter(self._fe_graphs)))

    def _step_finetune_eager(self, wav, labels, next_wav=None, next_step=0):
        be, fe = self.be, self.fe
        # the front-end's gradient buffer (1.26 GB) is cleared on the side stream while the forward pass runs: nothing reads it before
        # the backward pass, and the previous step's optimizer (its last reader) is already enqueued on the main stream
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream(device=wav.device)
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side):
            fe.zero_grad()
            cleared = torch.cuda.Event()
            cleared.record(self._side)
        feats = fe.forward_train(wav)
        if next_wav is not None:
            self._prefetch_augment(next_wav, next_step)
        main.wait_event(cleared)
        lc, ld, dfeats = self._backend_section(feats, labels, True)
        # each transformer layer's 50 MB of gradients go to RCCL as soon as that layer's backward is enqueued (last layer first);
        # the conv stack, the back-end and anything left over follow at the end
        # ... and the optimizer follows layer by layer: as soon as a layer's gradients are final (and, with several ranks, summed) its Adam
        # update -- HBM-bound, 30 B per parameter -- runs on its own stream beside the MFMA-bound backward of the layers below.  A layer's
        # f32 masters / bf16 forward operands are not read again in this step (backward multiplies by the transposed copies, which are
        # refreshed after the step), and the gradient slices are disjoint.
        red = self.reducers[1]
        scale = self.reducer.grad_scale
        overlap = self.overlap_optimizer and bool(self._chunk_of) and red.wire is None
        self.opt.set_grads(self._grads)
        stepped = set()
        if overlap and self._opt_stream is None:
            self._opt_stream = torch.cuda.Stream(device=wav.device)

        def ready(lo, hi):
            works = red.reduce_range(lo, hi)
            idx = self._chunk_of.get((lo, hi)) if overlap else None
            if idx is None or idx in stepped:
                return
            ev = torch.cuda.Event()
            ev.record(main)
            with torch.cuda.stream(self._opt_stream):
                self._opt_stream.wait_event(ev)
                for w in works:
                    w.wait()                                   # (stream-ordered for RCCL: the optimizer stream waits, not the host)
                self.opt.step_range(idx, 1, scale)
            stepped.add(idx)

        fe.backward(dfeats, grad_ready=ready if (overlap or red.world > 1) else None)
        for r in self.reducers:
            r.all_reduce()
        n, i = len(self.opt.params), 0
        while i < n:                                             # everything not updated yet, in runs of consecutive tensors
            if i in stepped:
                i += 1
                continue
            j = i
            while j < n and j not in stepped:
                j += 1
            self.opt.step_range(i, j - i, scale)
            i = j
        if stepped:
            main.wait_stream(self._opt_stream)
        fe.refresh_operands(cast=not hasattr(fe, "Wb"))
        self.last = (lc, ld)
        return lc, ld
