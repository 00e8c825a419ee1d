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
what = sys.argv[2] if len(sys.argv) > 2 else "bnn"      # round 6: "fnn" (the non-Bayesian pipeline) and "config3" (multi-hot input, unigram sampler: the one-pass first layer); an
B = 1000                                                # evaluation epoch (k_out_fwd_h3e / the lean producer) runs between the train chunks in every mode
ds = make_dataset("dblp", d=128, seed=0)
bayes, multihot = what != "fnn", what == "config3"
dims = [ds["S"] if multihot else 128, 128, ds["M"]]
e = libntf.Engine(dims, bayesian=bayes, input_mode=libntf.INPUT_MULTIHOT if multihot else libntf.INPUT_MEANPOOL, max_batch=B, ns=5, nsd="unigram" if multihot else "uniform", tpw=10.0, tnw=1.0,
                  lr=1e-3, seed=0, fuse_adam=1)   # as the plugin creates it (opentf_amd/mdl/fnn.py)
if not multihot: e.set_skill_table(ds["table"])
e.set_skill_csr(ds["skill"]); e.set_member(ds["member"]); e.load_state_dict(init_params(dims, bayes, 0))
if multihot: e.set_unigram(np.bincount(ds["member"][1], minlength=ds["M"]) / ds["N"])
rng = np.random.default_rng(0)
chunk = 2000
t0 = time.perf_counter()
done = 0
while done < steps:
    n = min(chunk, steps - done)
    order = rng.integers(0, ds["N"], n * B).astype(np.int64)
    loss = e.train_epoch(order, B)
    v_loss = e.eval_epoch(order[: 20 * B], B)
    done += n
    if done % 10_000 == 0 or done == steps:
        print(f"{what}: steps {done:7d}  mean loss of the last {n} steps {loss:12.4f}  validation {v_loss:12.4f}  range fallbacks {e.range_fallbacks()}  prefetched {e.prefetched_steps()}  "
              f"head hits {e.head_prefetch_hits()}  first-layer sweeps {e.first_layer_sweeps()}  {(time.perf_counter() - t0):7.1f} s", flush=True)
    assert np.isfinite(loss) and np.isfinite(v_loss)
sd = e.state_dict()
print("parameters finite:", all(np.isfinite(v).all() for v in sd.values()), " max |w| output layer", float(np.abs(sd["layers.1.mu_weight" if bayes else "layers.1.weight"]).max()),
      (" max rho %g" % float(sd["layers.1.rho_weight"].max())) if bayes else "")
e.close()
