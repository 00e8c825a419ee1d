"""Two REAL processes on ONE GPU (gloo moves the CUDA tensors; both ranks drive cuda:0): the Bnn plugin's learn() / test() under torch.distributed at world size 2,
in the expert-sharded and in the data-parallel form.  Run by tests/test_gpu_plugin.py::test_plugin_two_processes_one_gpu."""
import os
import sys

import numpy as np
import scipy.sparse


class Cfg(dict):
    def __getattr__(self, k):
        if k.startswith("__"): raise AttributeError(k)
        return self.get(k)


def dataset():
    """a small synthetic teamsvecs with enough experts to shard over two ranks (3 tiles of 256) and a mean-pool table (deterministic kernels end to end)"""
    rng = np.random.default_rng(11)
    N, S, M, d = 400, 60, 700, 128
    skill = scipy.sparse.random(N, S, density=0.08, random_state=1, format="csr", dtype=np.float32)
    skill.data[:] = 1; skill = skill.astype(np.uint8).tolil()
    for i in range(N):
        if not skill.rows[i]: skill[i, int(rng.integers(S))] = 1
    member = scipy.sparse.random(N, M, density=0.006, random_state=2, format="csr", dtype=np.float32)
    member.data[:] = 1; member = member.astype(np.uint8).tolil()
    for i in range(N):
        if not member.rows[i]: member[i, int(rng.integers(M))] = 1
    table = rng.standard_normal((S, d)).astype(np.float32)
    sk = scipy.sparse.csr_matrix(skill, dtype=np.float32)
    dense = np.asarray((sk @ table) / sk.sum(axis=1), dtype=np.float32)
    tv = {"skill": dense, "original_skill": skill, "member": member, "skill_table": table, "loc": None}
    idx = rng.permutation(N)
    splits = {"test": idx[:60], "folds": {0: {"train": idx[60:340], "valid": idx[340:]}}}
    return tv, splits


CFG = dict(b=64, e=2, ns=3, lr=0.01, es=5, h=[128], spe=0, l="bce", tpw=10, tnw=1, nsd="uniform", nmc=2)


def predictions(m, tv, splits, writer):
    """test() twice: dense predictions (kept as f0.test.dense.pred) and the top-5 sparse form (f0.test.pred)"""
    import shutil
    m.test(tv, splits, Cfg(per_epoch=False, on_train=False, topK=None))
    if writer: shutil.copy(f"{m.output}/f0.test.pred", f"{m.output}/f0.test.dense.pred")
    m.test(tv, splits, Cfg(per_epoch=False, on_train=False, topK=5))


def worker(rank, world, port, out_dir, mode):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_RANK="0", NTF_PARALLEL=mode)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from opentf_amd.mdl.bnn import Bnn
    tv, splits = dataset()
    m = Bnn(os.path.join(out_dir, mode), "cuda:0", 0, Cfg(CFG))
    m.learn(tv, splits, None)
    assert type(m._runner).__name__ == {"ep": "ExpertParallel", "dp": "DataParallel"}[mode]
    predictions(m, tv, splits, rank == 0)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    worker(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5])
