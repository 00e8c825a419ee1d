"""`tests/test_gpu_ep.py`'s band for the parameters after a short epoch on expert shards against the single engine was widened in round 4
(atol 2e-5 -> a handful of outliers allowed, none beyond 3e-4) in the same diff that moved `adam_step` onto `v_sqrt_f32` / `v_rcp_f32`.  Which of the two is it?

This prints, for every case of that test, how the gathered shards differ from the single engine after the first step and after the short epoch: elements outside
(rtol 1e-4, atol 2e-5), the largest absolute deviation, and |gradient scale| where it happens.  Run it once with the shipped library and once with
`NTF_LIB_PATH=<a -DNTF_ADAM_IEEE build>` (profiles/mk_variants.sh ieee: correctly rounded square root and division, round 3's `adam_step`): both sides of the comparison run the SAME
`adam_step` either way, so if the outliers are there with both builds they are not the hardware reciprocal's.

    python profiles/ep_tolerance.py > profiles/r5_ep_tolerance_hw.txt
    NTF_LIB_PATH=$PWD/scratch/var/adam_ieee.so python profiles/ep_tolerance.py > profiles/r5_ep_tolerance_ieee.txt
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))

from opentf_amd.ep import expert_shards            # noqa: E402
from opentf_amd.synth import make_dataset          # noqa: E402
import test_gpu_ep as T                            # noqa: E402


def main():
    print(f"library: {os.environ.get('NTF_LIB_PATH', 'opentf_amd/libopentf_amd.so (shipped)')}")
    print("case | after step 1: hidden-layer elements outside (1e-5, 2e-6), max |dev| | after the epoch: elements outside (1e-4, 2e-5) / all, max |dev|, worst tensor")
    for case in sorted(T.CASES):
        bayesian, mkdims, nsd, G, B, multihot = T.CASES[case][:6]
        fuse_adam = T.CASES[case][6] if len(T.CASES[case]) > 6 else 1
        ds = make_dataset("dblp", d=128, seed=3, n_rows=1500, n_experts=3000)
        dims = mkdims(ds)
        order = np.random.default_rng(4).permutation(ds["N"])[: 2 * B + 77].astype(np.int64)
        shards = expert_shards(ds["M"], G)
        full = T._mk(ds, dims, bayesian, B, nsd, multihot=multihot, fuse_adam=fuse_adam)
        eng = [T._mk(ds, dims, bayesian, B, nsd, shard=s, world=G, multihot=multihot, fuse_adam=fuse_adam) for s in shards]
        T._full_epoch(full, order[:B], B); T._ep_epoch(eng, order[:B], B)
        a, b = T._gathered(eng, exact_replicas=False), full.state_dict()
        n1 = sum(int((~np.isclose(a[k], b[k], rtol=1e-5, atol=2e-6)).sum()) for k in b)
        d1 = max(float(np.abs(a[k] - b[k]).max()) for k in b)
        T._full_epoch(full, order, B); T._ep_epoch(eng, order, B)
        a, b = T._gathered(eng, exact_replicas=False), full.state_dict()
        n2 = {k: int((~np.isclose(a[k], b[k], rtol=1e-4, atol=2e-5)).sum()) for k in b}
        d2 = {k: float(np.abs(a[k] - b[k]).max()) for k in b}
        worst = max(d2, key=d2.get)
        print(f"{case} | {n1}, {d1:.3e} | {sum(n2.values())} / {sum(v.size for v in b.values())}, {d2[worst]:.3e}, {worst}")
        for e in eng + [full]: e.close()


if __name__ == "__main__":
    main()
