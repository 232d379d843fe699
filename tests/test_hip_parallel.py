"""The data-parallel train step on real kernels (BASELINE config 4's code path): two ranks of
the HIP ``Trainer`` on device 0 exchanging gradients over gloo must reproduce the single-process step on
the global batch -- averaged flat gradient and post-Adam weights -- although every rank but 0 was built
from different initial weights (``Trainer`` broadcasts rank 0's).  The reference has no distributed code
(utils/train.py:163-164 trains on one device); the contract is the global-batch step."""
import os
import socket
import subprocess
import sys

import pytest
import torch

from dp_cases import CASES, build, global_batch
from helpers import EpsIndependent

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _single_process(case):
    from npf_gwwaveform_amd.train import Trainer

    model, crit = build(case, seed=0)
    trainer = Trainer(model, crit, lr=1e-3, world=1)
    batch = global_batch(case)
    eps = batch.pop("eps", None)
    losses, grads = [], []
    for _ in range(case.get("steps", 2)):
        if eps is not None:
            EpsIndependent.eps = eps
        losses.append(float(trainer.step(batch).item()))
        grads.append(trainer.flat.flat_grad.detach().cpu().clone())
    st = trainer.opt.state[trainer.flat.flat]
    adam = dict(exp_avg=st["exp_avg"].detach().cpu().clone(), exp_avg_sq=st["exp_avg_sq"].detach().cpu().clone(),
                step=float(st["step"]))
    return losses, grads, trainer.flat.flat.detach().cpu().clone(), adam


@pytest.mark.timeout(600)
@pytest.mark.parametrize("name", list(CASES))
def test_two_rank_trainer_equals_global_batch_step(name, tmp_path):
    case, world, port = CASES[name], 2, _free_port()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    outs = [str(tmp_path / f"rank{r}.pt") for r in range(world)]
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "dp_worker.py"), str(r), str(world), str(port), outs[r], name],
                              env=env) for r in range(world)]
    try:
        for p in procs:
            assert p.wait(timeout=500) == 0, "a data-parallel rank failed"
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    ref_losses, ref_grads, ref_w, ref_adam = _single_process(case)
    res = [torch.load(o, weights_only=True) for o in outs]
    # replicas agree bit for bit with each other (same averaged gradient, same Adam state)
    assert torch.equal(res[0]["weights"], res[1]["weights"])
    for s in range(len(ref_grads)):
        assert torch.equal(res[0]["grads"][s], res[1]["grads"][s])
        # the local losses are means over the local tasks: their mean is the global loss
        mean_loss = sum(r["losses"][s] for r in res) / world
        assert abs(mean_loss - ref_losses[s]) <= 2e-6 * abs(ref_losses[s]), (s, mean_loss, ref_losses[s])
        g, want = res[0]["grads"][s].double(), ref_grads[s].double()
        assert (g - want).abs().max() <= 1e-6 * want.abs().max(), (s, float((g - want).abs().max()), float(want.abs().max()))
    # the optimizer path: Adam's moments are linear (exp_avg) and quadratic (exp_avg_sq) in the averaged gradients, so the
    # 1e-6-of-max agreement of the gradients carries over to them at 1e-6 / 2e-6 of their largest entry, and the step
    # counts are equal -- this pins the all-reduce -> Adam hand-over without Adam's sign-like division in the way
    assert res[0]["step"] == res[1]["step"] == ref_adam["step"] == float(len(ref_grads))
    for key, tol in (("exp_avg", 1e-6), ("exp_avg_sq", 2e-6)):
        assert torch.equal(res[0][key], res[1][key]), key
        got, want = res[0][key].double(), ref_adam[key].double()
        assert (got - want).abs().max() <= tol * want.abs().max(), (key, float((got - want).abs().max()), float(want.abs().max()))
    # post-Adam weights after the steps: a sanity bound only (the moments above are the sharp check).  Adam divides by
    # sqrt(v): its step is lr * sign-like in the gradient, so an entry whose gradient is at the summation-order noise can move
    # by up to a step per step; entries with a gradient >= 1e-3 of the largest must agree to 1 % of lr
    lr, steps = 1e-3, len(ref_grads)
    d = (res[0]["weights"].double() - ref_w.double()).abs()
    gmin = torch.stack([g.double().abs() for g in ref_grads]).min(0).values
    gmax = max(float(g.abs().max()) for g in ref_grads)
    big = gmin >= 1e-3 * gmax
    assert d[big].max() <= 1e-5, float(d[big].max())
    assert d.max() <= 2.5 * lr * steps, float(d.max())


@pytest.mark.timeout(900)
@pytest.mark.parametrize("dp_graph", [False, True], ids=["eager_overlapped", "two_graphs"])
def test_bench_two_rank_line_carries_the_exchange_figures(dp_graph):
    """``bench.py --gpus 2 --config c4`` end to end on one GPU (``NPF_BENCH_REHEARSAL=1``: both ranks on device 0, gloo): the
    N > 1 control flow of BASELINE config 4 -- bench.py spawns its own ranks; the steps are launched eagerly with the bucketed,
    overlapped all-reduce (the default) or replayed from two HIP graphs around one all-reduce (``--dp-graph``) -- and the line
    carries what makes "overlap" a number: the exposed all-reduce time and the backward time."""
    import json

    env = dict(os.environ, NPF_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    cmd = [sys.executable, os.path.join(os.path.dirname(HERE), "bench.py"), "--gpus", "2", "--config", "c4", "--batch", "16",
           "--steps", "3", "--warmup", "1", "--no-roofline"] + (["--dp-graph"] if dp_graph else [])
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=800)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["hip_graph"] is dp_graph and line["config"]["global_tasks"] == 32
    assert "REHEARSAL" in line["data"]
    assert line["allreduce_ms_exposed"] >= 0 and line["backward_ms"] > 0
    if dp_graph:
        assert line["allreduce_ms_exposed"] > 0  # (the one all-reduce between the two replays is not hidden)
    eager = line["eager"]
    assert eager["ms_per_step"] > 0 and eager["backward_ms"] > 0 and eager["allreduce_ms_exposed"] >= 0
    assert line["value"] > 0 and abs(line["value"] - 2 * 16 * 1024 / (line["ms_per_step"] * 1e-3)) <= 1e-6 * line["value"]
