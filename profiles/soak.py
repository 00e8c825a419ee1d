"""Soak run of the default train step at BASELINE config 2 (synthetic dblp shapes): many steps back to back; the loss must stay finite and fall, no step may
leave the fp16 window (range fallback), every step after the first must start on operands the previous step's dW epilogue wrote."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from opentf_amd import libntf                                  # noqa: E402
from opentf_amd.synth import make_dataset, init_params         # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
B = 1000
ds = make_dataset("dblp", d=128, seed=0)
dims = [128, 128, ds["M"]]
e = libntf.Engine(dims, bayesian=True, input_mode=libntf.INPUT_MEANPOOL, max_batch=B, ns=5, nsd="uniform", tpw=10.0, tnw=1.0, lr=1e-3, seed=0, fuse_adam=1)   # as the plugin creates it (opentf_amd/mdl/fnn.py)
e.set_skill_table(ds["table"]); e.set_skill_csr(ds["skill"]); e.set_member(ds["member"]); e.load_state_dict(init_params(dims, True, 0))
rng = np.random.default_rng(0)
chunk = 2000
t0 = time.perf_counter()
done = 0
while done < steps:
    n = min(chunk, steps - done)
    order = rng.integers(0, ds["N"], n * B).astype(np.int64)
    loss = e.train_epoch(order, B)
    done += n
    if done % 10_000 == 0 or done == steps:
        print(f"steps {done:7d}  mean loss of the last {n} steps {loss:12.4f}  range fallbacks {e.range_fallbacks()}  prefetched {e.prefetched_steps()}  {(time.perf_counter() - t0):7.1f} s", flush=True)
    assert np.isfinite(loss)
sd = e.state_dict()
print("parameters finite:", all(np.isfinite(v).all() for v in sd.values()), " max |mu| output layer", float(np.abs(sd["layers.1.mu_weight"]).max()), " max rho", float(sd["layers.1.rho_weight"].max()))
e.close()
