#!/usr/bin/env python3
"""Prove the mount INTEGRATION.md section 1 prescribes: `make_fnn / make_bnn / make_tntf` applied to THE REFERENCE'S OWN `mdl.ntf.Ntf`
(src/mdl/ntf.py:5-31), constructed exactly as src/main.py:168-172 constructs models.

Build container only (imports /root/reference with the six shims of SURVEY.md 8c; never shipped: listed in .gpurunignore).  No GPU here, so
`learn()` must get as far as creating the engine and fail THERE - everything in front of it (the reference base class's constructor, naming,
output directory, seeding, parameter init, fold loop entry) runs for real.

    python tests/golden/check_reference_mount.py          # prints one line per check, exits non-zero on the first failure
"""
import os
import pickle
import sys
import tempfile

import numpy as np
import torch
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
from make_golden import install_shims, Cfg, REF  # noqa: E402


def ok(what, cond, detail=""):
    print(("ok   " if cond else "FAIL ") + what + (f"  [{detail}]" if detail else ""))
    if not cond:
        sys.exit(1)


def main():
    SummaryWriter, _ = install_shims()            # chdir to /root/reference/src, stubs omegaconf / tensorboardX, makes pkgmgr.install_import harmless
    from mdl.ntf import Ntf                        # the reference's base class
    from mdl.fnn import Fnn as RefFnn              # ... and its own Fnn, to compare init draws and names with
    import pkgmgr
    from opentf_amd.mdl.fnn import make_fnn
    from opentf_amd.mdl.bnn import make_bnn
    from opentf_amd.mdl.tntf import make_tntf

    # the three shim files of INTEGRATION.md section 1
    Fnn = make_fnn(Ntf)
    Bnn = make_bnn(Fnn)
    tNtf = make_tntf(Ntf)
    ok("class names select the reference's config sections (main.py:172)", (Fnn.__name__, Bnn.__name__, tNtf.__name__) == ("Fnn", "Bnn", "tNtf"))
    ok("the factory classes derive from the reference's Ntf", issubclass(Fnn, Ntf) and issubclass(Bnn, Fnn) and issubclass(tNtf, Ntf))

    # cfg.models.config[cls.__name__.lower()] of src/mdl/__config__.yaml, interpolations resolved by hand (hydra / omegaconf are absent)
    raw = yaml.safe_load(open(f"{REF}/src/mdl/__config__.yaml"))
    top = {k: v for k, v in raw.items() if not isinstance(v, dict)}

    def section(name):
        return Cfg({k: (top[v[2:-1]] if isinstance(v, str) and v.startswith("${") else v) for k, v in raw[name].items()})
    cfg_fnn, cfg_bnn = section("fnn"), section("bnn")
    cfg_fnn["spe"] = cfg_bnn["spe"] = 10            # main.py overrides it from the root config

    out = tempfile.mkdtemp(prefix="mount_")
    seed = 0
    ref = RefFnn(out + "/ref", "cpu", seed, cfg_fnn)
    ref_model = ref.init(10, 13)
    ref_sd = {k: v.detach().clone() for k, v in ref_model.state_dict().items()}

    m = Fnn(out + "/ours", "cuda:0", seed, cfg_fnn)            # main.py:172: cls(output_, cfg.acceleration, cfg.seed, cfg.models.config['fnn'])
    ok("Fnn.name() is the reference's directory name", m.name() == ref.name(), m.name())
    ok("Fnn.output = output + name(), created by the reference's constructor", m.output == out + "/ours" + ref.name() and os.path.isdir(m.output))
    ok("Fnn.writer is what the reference's constructor set", m.writer is SummaryWriter)
    ok("Ntf.torch / Ntf.dataset were set by the reference's constructor", Ntf.torch is torch and Ntf.dataset is not None)
    ok("Fnn.is_bayesian is False, cfg / seed / device kept", m.is_bayesian is False and m.cfg is cfg_fnn and m.seed == seed and m.device == "cuda:0")
    ok("evaluate / adila / name are the reference's own methods", "evaluate" not in Fnn.__dict__ and "adila" not in Fnn.__dict__ and Fnn.evaluate is Ntf.evaluate and Fnn.name is Ntf.name)
    sd = m.init(10, 13)
    same = all(torch.equal(sd[k], ref_sd[k]) for k in ref_sd) and list(sd) == list(ref_sd)
    ok("init() draws the reference's initial weights from the seed (keys, order, values)", same, ", ".join(sd))

    b = Bnn(out + "/ours", "cuda:0", seed, cfg_bnn)
    ok("Bnn.is_bayesian, directory prefix 'bnn.'", b.is_bayesian is True and os.path.basename(b.output).startswith("bnn.") and b.name() == f"/bnn.{pkgmgr.cfg2str(cfg_bnn)}")
    bsd = b.init(10, 13)
    ok("Bnn.init(): the committed checkpoints' key layout", list(bsd) == [f"layers.{i}.{n}" for i in range(2) for n in ("mu_weight", "rho_weight", "mu_bias", "rho_bias")])

    inner = Fnn(out + "/t", "cuda:0", seed, cfg_fnn)
    i2y = [(0, 1990), (12, 1995), (24, 2000)]
    t = tNtf(out + "/t", "cuda:0", seed, Cfg(step_ahead=1, tfolds=3), inner, i2y)      # main.py:168-170
    ok("tNtf wraps the inner model and adopts its output directory", t.model is inner and t.output == inner.output and t.name() == "")

    # learn(): everything up to the engine runs; without a HIP device the engine's creation raises (there is no CPU fallback)
    with open(f"{REF}/output/dblp/toy.dblp.v12.json/teamsvecs.pkl", "rb") as f: tv = pickle.load(f)
    with open(f"{REF}/output/dblp/toy.dblp.v12.json/splits.f3.r0.85.pkl", "rb") as f: sp = pickle.load(f)
    try:
        m.learn(tv, sp, None)
        ok("learn() without a GPU raises at engine creation", False, "it returned")
    except Exception as e:   # noqa: BLE001
        msg = f"{type(e).__name__}: {e}"
        ok("learn() without a GPU raises at engine creation", any(s in msg.lower() for s in ("hip", "device", "gpu", "engine", "libopentf")), msg[:160])
    print("mount check passed: the factory classes survive the reference's Ntf.__init__ and are called as src/main.py calls them")


if __name__ == "__main__":
    main()
