"""Minimal optimisation step of the path and the synthetic waveform batches it is timed on.

The reference trains through skorch (utils/train.py:260-263; ``get_loss`` :342-349): per
batch ``module.train(); out = module(**batch); loss = criterion(out, Y_trgt);
loss.backward(); Adam(lr=1e-3).step()``.  ``Trainer.step`` is that sequence with the
data-parallel gradient all-reduce in front of the optimizer.
Host code (Adam, all-reduce, control flow) is PyTorch, as the north star prescribes.
"""
from __future__ import annotations

import json
import os
from typing import Optional

import torch
import torch.distributed as dist

from .parallel import BucketedGradReducer, FlatParameters


def synthetic_waveform_batch(B: int, C: int, T: int, seed: int, device, dx: int = 1, dy: int = 2, n_z_eps: int = 0,
                             z_dim: int = 0):
    """Random-source-parameter, random-frequency-grid batch: per task a smooth
    inspiral-like amplitude/phase pair on ``C + T`` sorted random frequencies in [-1, 1];
    a random subset of C points is the context, the other T are the targets.  Generated on
    the device; throughput does not depend on the values."""
    assert dx == 1 and dy == 2
    g = torch.Generator(device=device).manual_seed(seed)
    N = C + T
    theta = torch.rand(B, 4, generator=g, device=device)
    f, _ = torch.sort(torch.rand(B, N, generator=g, device=device) * 2 - 1, dim=1)
    ft = 1.5 + f / 2  # in [1, 2]
    amp = ft.pow(-7.0 / 6.0) * (1 + 0.5 * theta[:, 0:1])
    phase = theta[:, 3:4] + theta[:, 0:1] * ft.pow(-5.0 / 3.0) + theta[:, 1:2] / ft + theta[:, 2:3] * ft
    phase = (phase - phase.mean()) / phase.std()
    amp = (amp - amp.mean()) / amp.std()
    Y = torch.stack([amp, phase], dim=-1)
    X = f.unsqueeze(-1)
    perm = torch.argsort(torch.rand(B, N, generator=g, device=device), dim=1)
    ci, ti = perm[:, :C], perm[:, C:]
    take = lambda t, idx: torch.gather(t, 1, idx.unsqueeze(-1).expand(-1, -1, t.shape[-1])).contiguous()  # noqa: E731
    return dict(X_cntxt=take(X, ci), Y_cntxt=take(Y, ci), X_trgt=take(X, ti), Y_trgt=take(Y, ti))


def get_exponential_decay_gamma(scheduling_factor, max_epochs):
    """Per-epoch factor that reduces the learning rate by ``scheduling_factor`` over ``max_epochs``
    (utils/helpers.py:35-46, used with ExponentialLR in utils/train.py:233-237)."""
    return (1 / scheduling_factor) ** (1 / max_epochs)


class Trainer:
    """forward -> loss -> backward -> (bucketed all-reduce) -> Adam, on flat parameter /
    gradient buffers."""

    def __init__(self, model: torch.nn.Module, criterion: torch.nn.Module, lr: float = 1e-3, world: Optional[int] = None,
                 bucket_bytes: int = 2 << 20, use_graph: bool = False, defer_input_check: bool = False):
        """``use_graph``: capture the step in a HIP graph after a few eager steps and replay it -- a c2-sized step is
        launched eagerly in 7.4 ms of host time against 6.7 ms of device time.  One rank: the whole step (forward, loss,
        backward, Adam) is one graph.  Several ranks: forward + loss + backward + the gather into the flat gradient are one
        graph, Adam a second one, and between the two replays ONE all-reduce of the flat gradient is issued eagerly (no
        collective inside a captured region; the exchange is latency-bound -- 3-5 MB -- and is then not hidden behind the
        backward pass, which costs less than the host time of an eager step at these sizes; ``use_graph=False`` keeps the
        bucketed all-reduce overlapped with the backward pass).  Needs fixed batch shapes (a new shape or learning rate
        re-captures).  The range check of the inputs (base.py:241-247) stays in the step as a device reduction; its verdict
        is read at the next sync point: call :meth:`check_inputs` (e.g. once per epoch) to get the reference's ValueError."""
        self.model, self.criterion = model, criterion
        # ``defer_input_check``: eagerly launched steps, too, keep the range check of base.py:241-247 on the device and leave its
        # verdict to :meth:`check_inputs` -- the reference's synchronous check is a host sync per step (measured on a config-2
        # step: 4 ms of the host's 7 waiting in ``.item()``, and the forward pass then starts on an idle GPU: 6.9 against 6.3 ms)
        if defer_input_check and getattr(model, "validate_inputs", False) is True:
            model.validate_inputs = "deferred"
        self.flat = FlatParameters(model.parameters())
        self.reducer = BucketedGradReducer(self.flat, world=world, bucket_bytes=bucket_bytes)
        self.use_graph = bool(use_graph) and self.flat.flat.is_cuda
        if self.use_graph and hasattr(getattr(model, "n_z_samples_train", None), "rvs"):
            raise ValueError("use_graph=True freezes the step at capture time, but this model draws a random number of "
                             "latent samples per forward (n_z_samples_train is a random variable)")
        # one fused kernel per step on the flat buffer (the default implementation is ~8 launches)
        self.opt = torch.optim.Adam([self.flat.flat], lr=lr, fused=self.flat.flat.is_cuda, capturable=self.use_graph)
        self.model.train()
        self.criterion.train()
        self._graph = None
        self._graph_opt = None
        self._eager_steps = 0
        # set to a list to time the phases of every eager step with HIP events on the launch stream (bench.py): entries are
        # (start of backward, end of backward, gradient exchange finished); ``phase_times()`` reads them after a sync
        self.phase_events = None
        self.sync_replicas()

    def sync_replicas(self) -> None:
        """Data-parallel replicas must start from the same weights and optimizer state: rank 0's flat
        parameter buffer (and Adam moments / step count, once they exist) are broadcast to every rank.
        Called at construction and after a checkpoint load; a no-op on a single rank."""
        if self.reducer.world <= 1:
            return
        if not dist.is_initialized():
            raise RuntimeError(f"Trainer(world={self.reducer.world}) needs an initialised torch.distributed process group")
        group = self.reducer.group
        dist.broadcast(self.flat.flat.data, src=0, group=group)
        have = torch.tensor([1.0 if self.opt.state.get(self.flat.flat) else 0.0], device=self.flat.flat.device)
        dist.broadcast(have, src=0, group=group)
        if have.item() > 0:
            st = self.opt.state[self.flat.flat]
            if not st:  # rank 0 resumed from a checkpoint, this rank did not: allocate the same state
                st["step"] = torch.zeros((), dtype=torch.float32, device=self.flat.flat.device)
                st["exp_avg"] = torch.zeros_like(self.flat.flat.data)
                st["exp_avg_sq"] = torch.zeros_like(self.flat.flat.data)
            for key in ("exp_avg", "exp_avg_sq"):
                dist.broadcast(st[key], src=0, group=group)
            step = st["step"] if torch.is_tensor(st["step"]) else torch.tensor(float(st["step"]))
            step_dev = step.to(self.flat.flat.device, torch.float32).reshape(1).clone()
            dist.broadcast(step_dev, src=0, group=group)
            if torch.is_tensor(st["step"]):
                st["step"].copy_(step_dev.reshape(()).to(st["step"].device))
            else:
                st["step"] = float(step_dev.item())

    def step(self, batch: dict, eager: bool = False) -> torch.Tensor:
        """One optimisation step; ``eager``: bypass the captured graph for this step (instrumented runs)."""
        if self.use_graph and not eager:
            return self._graph_step(batch)
        return self._eager_step(batch)

    def check_inputs(self) -> None:
        """Graph mode: raise if a batch since the last call had features outside [-1, 1] (one host sync)."""
        if hasattr(self.model, "check_deferred_inputs"):
            self.model.check_deferred_inputs()

    def _graph_step(self, batch: dict) -> torch.Tensor:
        sig = (tuple((k, tuple(v.shape)) for k, v in sorted(batch.items())), float(self.opt.param_groups[0]["lr"]))
        if self._graph is not None and sig != self._graph_sig:
            self._graph = None  # new shapes / learning rate: capture again
        if self._graph is None:
            if self._eager_steps < 3:  # allocator warm-up and lazy initialisations happen eagerly
                self._eager_steps += 1
                return self._eager_step(batch)
            was = self.model.validate_inputs
            self.model.validate_inputs = "deferred" if was else False
            # (the running maximum lives outside the graph's memory pool: a tensor created during capture would be
            # re-initialised by every replay)
            seen = getattr(self.model, "_range_seen", None)
            if seen is None or seen.device != self.flat.flat.device:  # (kept across re-captures: a verdict not yet read survives)
                self.model._range_seen = torch.zeros((), device=self.flat.flat.device)
            try:
                self._static = {k: v.clone() for k, v in batch.items()}
                torch.cuda.synchronize()
                self._graph = torch.cuda.CUDAGraph()
                if self.reducer.world == 1:
                    with torch.cuda.graph(self._graph):
                        self._static_loss = self._eager_step(self._static)
                    self._graph_opt = None
                else:
                    # several ranks: the collective library keeps a watchdog thread that queries events; "thread_local" keeps
                    # its calls from invalidating this thread's capture.  Should the capture fail all the same, the ranks go
                    # on launching their steps eagerly (every rank takes the same decision: the captured work is the same).
                    self.reducer.deferred = True
                    try:
                        with torch.cuda.graph(self._graph, capture_error_mode="thread_local"):
                            self._static_loss = self._forward_backward(self._static)
                        self._graph_opt = torch.cuda.CUDAGraph()
                        with torch.cuda.graph(self._graph_opt, pool=self._graph.pool(), capture_error_mode="thread_local"):
                            self.opt.step()
                    except RuntimeError as e:
                        import warnings

                        warnings.warn(f"HIP-graph capture of the data-parallel step failed ({e}); launching the steps eagerly")
                        self.use_graph, self._graph, self._graph_opt = False, None, None
                        torch.cuda.synchronize()
                    finally:
                        self.reducer.deferred = False
            finally:
                self.model.validate_inputs = was
            if self._graph is None:
                return self._eager_step(batch)
            self._graph_sig = sig  # (a capture records, it does not execute: this batch runs in the replay below)
        for k, v in batch.items():
            self._static[k].copy_(v)
        timed = self.phase_events is not None
        if timed:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            ev[0].record()
        self._graph.replay()
        if self._graph_opt is not None:
            if timed:
                ev[1].record()
            self.reducer.exchange()
            if timed:
                ev[2].record()
                self.phase_events.append(ev)
            self._graph_opt.replay()
        return self._static_loss.clone()

    def _forward_backward(self, batch: dict) -> torch.Tensor:
        """Forward, loss, backward and the gather of the LOCAL gradients into the flat buffer (``reducer.deferred``)."""
        for p in self.flat.params:
            p.grad = None
        self.reducer.reset()
        out = self.model(batch["X_cntxt"], batch["Y_cntxt"], batch["X_trgt"], batch["Y_trgt"])
        loss = self.criterion(out, batch["Y_trgt"])
        loss.backward()
        self.flat.flat.grad = self.reducer.finish()
        return loss.detach()

    def _eager_step(self, batch: dict) -> torch.Tensor:
        for p in self.flat.params:
            p.grad = None
        self.reducer.reset()
        out = self.model(batch["X_cntxt"], batch["Y_cntxt"], batch["X_trgt"], batch["Y_trgt"])
        loss = self.criterion(out, batch["Y_trgt"])
        timed = self.phase_events is not None and loss.is_cuda
        if timed:
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            ev[0].record()
        loss.backward()
        if timed:
            ev[1].record()
        self.flat.flat.grad = self.reducer.finish()
        if timed:
            ev[2].record()
            self.phase_events.append(ev)
        self.opt.step()
        return loss.detach()

    def phase_times(self) -> dict:
        """Means over the steps recorded in ``phase_events`` (call after a device sync).  Eager steps: ``backward_ms`` = the
        backward pass on the launch stream (its bucketed all-reduces start inside it, on the collective's own stream);
        graph steps of several ranks: ``backward_ms`` = the replay of forward + backward, and the exchange is the one
        all-reduce between the two graphs;
        ``allreduce_ms_exposed`` = what the launch stream spends behind the backward pass until the averaged gradient is
        ready (the wait for the collectives still in flight, the gather of late buckets, the 1/world scaling) -- the part of
        the exchange the backward pass did not hide."""
        evs = self.phase_events or []
        if not evs:
            return {}
        n = len(evs)
        return {"backward_ms": sum(e[0].elapsed_time(e[1]) for e in evs) / n,
                "allreduce_ms_exposed": sum(e[1].elapsed_time(e[2]) for e in evs) / n, "steps_timed": n}

    # ---- what the reference gets from skorch callbacks (utils/train.py:203-241) ------------------
    def set_lr_decay(self, decay_lr: float, max_epochs: int) -> None:
        """Exponential decay by ``decay_lr`` in total over ``max_epochs`` (ExponentialLR per epoch)."""
        self._sched = torch.optim.lr_scheduler.ExponentialLR(self.opt, gamma=get_exponential_decay_gamma(decay_lr, max_epochs))

    def end_epoch(self) -> float:
        """Advance the learning-rate schedule (if any); returns the new learning rate."""
        if getattr(self, "_sched", None) is not None:
            self._sched.step()
        return self.opt.param_groups[0]["lr"]

    # The reference checkpoints through skorch's ``Checkpoint`` callback (utils/train.py:203-221): ``params.pt`` = the
    # module's state_dict, ``optimizer.pt`` = ``torch.optim.Adam(module.parameters()).state_dict()`` -- one
    # ``exp_avg`` / ``exp_avg_sq`` / ``step`` entry PER PARAMETER, in ``module.parameters()`` order -- and
    # ``history.json``.  This Trainer runs Adam on ONE flat buffer; the two functions below translate.
    def optimizer_state_dict(self) -> dict:
        """Adam's state in the per-parameter layout of ``torch.optim.Adam(model.parameters())`` (what skorch writes to
        ``optimizer.pt``): every parameter's moments are the matching slice of the flat moments."""
        flat_sd = self.opt.state_dict()
        group = {k: v for k, v in flat_sd["param_groups"][0].items() if k != "params"}
        group["fused"] = None  # (how THIS process steps is not part of the optimisation state)
        group["capturable"] = False
        group["params"] = list(range(len(self.flat.params)))
        state = {}
        st = self.opt.state.get(self.flat.flat)
        if st:
            step = st["step"].detach().float().cpu().reshape(()) if torch.is_tensor(st["step"]) else torch.tensor(float(st["step"]))
            for i, (p, o, n) in enumerate(zip(self.flat.params, self.flat.offsets, self.flat.sizes)):
                state[i] = {"step": step.clone(),
                            "exp_avg": st["exp_avg"][o:o + n].detach().view_as(p).cpu().clone(),
                            "exp_avg_sq": st["exp_avg_sq"][o:o + n].detach().view_as(p).cpu().clone()}
        return {"state": state, "param_groups": [group]}

    def load_optimizer_state_dict(self, sd: dict) -> None:
        """Inverse of :meth:`optimizer_state_dict`; also takes an ``optimizer.pt`` written by the reference's skorch
        harness for the same model (same parameters in the same order), or this Trainer's older flat layout."""
        groups, state = sd["param_groups"], sd["state"]
        if len(groups) != 1:
            raise ValueError("optimizer state with several parameter groups")
        ids = list(groups[0]["params"])
        dev = self.flat.flat.device
        # a captured step holds the ADDRESSES of the moment / step tensors: whatever is loaded is copied into the existing
        # tensors, and the graph is dropped anyway (the next step re-captures after its eager warm-up)
        self._graph, self._eager_steps = None, 0
        if len(ids) == 1 and len(self.flat.params) != 1:  # the flat layout (one parameter = the whole buffer)
            old = dict(self.opt.state.get(self.flat.flat) or {})
            self.opt.load_state_dict(sd)
            st = self.opt.state.get(self.flat.flat)
            if st and old:
                for key in ("exp_avg", "exp_avg_sq", "step"):
                    if torch.is_tensor(old.get(key)) and torch.is_tensor(st.get(key)) and old[key].shape == st[key].shape:
                        old[key].copy_(st[key].to(old[key].device, old[key].dtype))
                        st[key] = old[key]
        else:
            if len(ids) != len(self.flat.params):
                raise ValueError(f"optimizer state for {len(ids)} parameters, the model has {len(self.flat.params)}")
            for k in ("lr", "betas", "eps", "weight_decay", "amsgrad"):
                if k in groups[0]:
                    self.opt.param_groups[0][k] = groups[0][k]
            if state:
                exp_avg, exp_avg_sq = torch.zeros_like(self.flat.flat.data), torch.zeros_like(self.flat.flat.data)
                steps = set()
                for pid, p, o, n in zip(ids, self.flat.params, self.flat.offsets, self.flat.sizes):
                    e = state[pid]
                    if tuple(e["exp_avg"].shape) != tuple(p.shape):
                        raise ValueError(f"optimizer state of parameter {pid} has shape {tuple(e['exp_avg'].shape)}, "
                                         f"the model's is {tuple(p.shape)}")
                    exp_avg[o:o + n].copy_(e["exp_avg"].reshape(-1))
                    exp_avg_sq[o:o + n].copy_(e["exp_avg_sq"].reshape(-1))
                    steps.add(float(e["step"]))
                if len(steps) != 1:
                    raise ValueError("per-parameter step counts differ: not one Adam run over all parameters")
                st = self.opt.state[self.flat.flat]
                like = st.get("step")
                step = torch.tensor(steps.pop(), dtype=torch.float32)
                if torch.is_tensor(like):
                    like.copy_(step.to(like.device, like.dtype))
                else:
                    st["step"] = step.to(dev) if self.flat.flat.is_cuda else step
                for key, new in (("exp_avg", exp_avg), ("exp_avg_sq", exp_avg_sq)):
                    if torch.is_tensor(st.get(key)) and st[key].shape == new.shape:
                        st[key].copy_(new)
                    else:
                        st[key] = new

    def save_checkpoint(self, dirname: str, history: Optional[list] = None) -> None:
        """skorch ``Checkpoint`` layout, interchangeable with the reference's: ``params.pt`` = the module's state_dict
        (the reference's key names), ``optimizer.pt`` = the per-parameter Adam state, ``history.json``."""
        os.makedirs(dirname, exist_ok=True)
        torch.save({k: v.detach().cpu() for k, v in self.model.state_dict().items()}, os.path.join(dirname, "params.pt"))
        torch.save(self.optimizer_state_dict(), os.path.join(dirname, "optimizer.pt"))
        with open(os.path.join(dirname, "history.json"), "w") as f:
            json.dump(history or [], f)

    def load_checkpoint(self, dirname: str, load_optimizer: bool = True) -> list:
        """Inverse of :meth:`save_checkpoint`; ``params.pt`` / ``optimizer.pt`` may also be ones written by the reference
        (e.g. ``results/pretrained/*/run_0/``).  Loaded with ``weights_only=True``."""
        sd = torch.load(os.path.join(dirname, "params.pt"), map_location="cpu", weights_only=True)
        self.model.load_state_dict(sd, strict=True)  # parameters are views of the flat buffer: copied in place
        self._graph, self._eager_steps = None, 0     # (a captured step is re-captured after the load)
        opt_path = os.path.join(dirname, "optimizer.pt")
        if load_optimizer and os.path.exists(opt_path):
            try:
                self.load_optimizer_state_dict(torch.load(opt_path, map_location="cpu", weights_only=True))
            except (KeyError, RuntimeError) as e:
                raise ValueError(f"{opt_path} does not hold an Adam state for this model: {e}") from e
        self.sync_replicas()  # (ranks that loaded nothing, or something else, follow rank 0)
        hist = os.path.join(dirname, "history.json")
        return json.load(open(hist)) if os.path.exists(hist) else []
