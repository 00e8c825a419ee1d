"""Two REAL processes on ONE GPU (gloo moves the CUDA tensors): the expert-sharded step end to end - ExpertParallel on real engines, world size 2 -
against the single-engine step.  Run by tests/test_gpu_ep.py::test_two_processes_one_gpu (RCCL refuses two ranks on one device, gloo does not)."""
import os
import sys

import numpy as np


def worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from opentf_amd import libntf
    from opentf_amd.ep import ExpertParallel, expert_shards
    from opentf_amd.synth import make_dataset, init_params
    torch.cuda.set_device(0)
    ds = make_dataset("dblp", d=128, seed=3, n_rows=1500, n_experts=3000)
    dims = [128, 128, ds["M"]]
    order = np.random.default_rng(4).permutation(ds["N"])[:577].astype(np.int64)
    stream = torch.cuda.Stream()
    with torch.cuda.stream(stream):
        e = libntf.Engine(dims, bayesian=True, input_mode=libntf.INPUT_MEANPOOL, max_batch=256, ns=5, nsd="uniform", seed=5, fuse_adam=1,
                          stream=stream.cuda_stream, expert_shard=expert_shards(ds["M"], world)[rank], ep_world=world)
        e.set_skill_table(ds["table"]); e.set_skill_csr(ds["skill"]); e.set_member(ds["member"]); e.load_state_dict(init_params(dims, True, 0))
        ep = ExpertParallel(e)
        t_loss = ep.train_epoch(order, 256)
        v_loss = ep.eval_epoch(order[:300], 256)
        sd = ep.state_dict()
    if rank == 0:
        np.savez(os.path.join(out_dir, "ep2.npz"), t_loss=t_loss, v_loss=v_loss, **sd)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    worker(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4])
