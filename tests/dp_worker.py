"""One data-parallel rank of tests/test_hip_parallel.py (started as a child process, never imported by
pytest): the real HIP ``Trainer`` on device 0, gradients exchanged over gloo -- the one-GPU rehearsal of
the N > 1 path.

    python tests/dp_worker.py <rank> <world> <port> <out.pt> <case>
"""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def make_case(name: str):
    from dp_cases import CASES

    return CASES[name]


def main():
    rank, world, port, out_path, case_name = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dp_cases import build, global_batch
    from npf_gwwaveform_amd.parallel import shard_range
    from npf_gwwaveform_amd.train import Trainer

    case = make_case(case_name)
    # every rank but 0 starts from DIFFERENT weights: only the broadcast in Trainer.__init__ makes the replicas agree
    model, crit = build(case, seed=0 if rank == 0 else 100 + rank)
    trainer = Trainer(model, crit, lr=1e-3, world=world, use_graph=bool(case.get("use_graph")))
    batch = global_batch(case)
    a, b = shard_range(case["B"], rank, world)
    local = {k: v[:, a:b].contiguous() if k == "eps" else v[a:b].contiguous() for k, v in batch.items()}
    eps = local.pop("eps", None)
    losses, grads = [], []
    for _ in range(case.get("steps", 2)):
        if eps is not None:
            from helpers import EpsIndependent

            EpsIndependent.eps = eps
        losses.append(float(trainer.step(local).item()))
        grads.append(trainer.flat.flat_grad.detach().cpu().clone())
    torch.cuda.synchronize()
    if case.get("use_graph"):
        assert trainer._graph is not None and trainer._graph_opt is not None, "the step was not replayed from the two graphs"
    st = trainer.opt.state[trainer.flat.flat]
    torch.save({"rank": rank, "losses": losses, "grads": grads, "weights": trainer.flat.flat.detach().cpu().clone(),
                "exp_avg": st["exp_avg"].detach().cpu().clone(), "exp_avg_sq": st["exp_avg_sq"].detach().cpu().clone(),
                "step": float(st["step"])}, out_path)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
