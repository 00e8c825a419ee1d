"""bench.py's own arithmetic (no GPU): the whole-step roofline's FLOP / byte counts are SURVEY 8d's, and `roofline.traffic` is refused when the committed PMC file is about
another kernel than the one the run launches."""
import os
import sys
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _args(**k):
    d = dict(dataset="dblp", model="bnn", batch=1000, d=128, hidden=128, input="meanpool", rows=0, experts=0, mfma="default", no_fused=False)
    d.update(k)
    return types.SimpleNamespace(**d)


def test_step_roofline_counts_are_the_surveys():
    a = _args()
    head = {"eB": 1000, "Mloc": 233_629}
    ds = {"skill": (np.array([0, 8]), None), "N": 1}
    r = bench.step_roofline(a, True, head, ds, False, 1.35e-3)
    assert r["flop_per_step"] == 1000 * (12.0 * 128 * 233_629 + 8.0 * 128 * 128)            # Bnn 12 H M + 8 D H per team
    assert r["bytes_per_step"] == (8.0 * 128 * 233_629 + 4.0 * 1024 * 233_728) + (4.0 * 1024 * 233_728 + 56 * 128 * 233_629)
    assert r["mfma_peak"] == pytest.approx(2516.6 / 3) and r["hbm_peak"] == 8000.0
    assert r["mfma_frac"] == pytest.approx(r["flop_per_step"] / 1.35e-3 / 1e12 / (2516.6 / 3))
    f = bench.step_roofline(_args(model="fnn"), False, head, ds, False, 0.88e-3)
    assert f["flop_per_step"] == 1000 * 6.0 * (128 * 128 + 128 * 233_629)                  # Fnn 6 (D H + H M)
    v = bench.validation_step({"ms_per_step": 0.5, "steps": 40, "mean_loss": 1.0}, a, True, 1000, 128, 233_629)
    assert v["flop_per_step"] == 2 * 2.0 * 1000 * 128 * 233_629 and 0.2 < v["mfma_frac"] < 0.4


def test_pmc_traffic_is_refused_for_another_kernel(monkeypatch):
    a = _args()
    monkeypatch.delenv("NTF_FWD_KERNEL", raising=False); monkeypatch.delenv("NTF_DW_KERNEL", raising=False)
    t, src = bench.pmc_traffic("out_fused_fwd_loss_dh", a, None)
    assert src and src.endswith("_pmc_traffic_and_sq.json") and 1.1e9 < t < 1.4e9          # the forward kernel's 1.22 GB (planes read once, packed dz written once)
    t2, _ = bench.pmc_traffic("out_fused_dw_adam", a, None)
    assert 2.5e9 < t2 < 2.9e9
    monkeypatch.setenv("NTF_FWD_KERNEL", "0")                                              # the run launches k_out_fwd_b6: the file (k_out_fwd_h3p) says nothing about it
    assert bench.pmc_traffic("out_fused_fwd_loss_dh", a, None) == (None, None)
    monkeypatch.delenv("NTF_FWD_KERNEL")
    assert bench.pmc_traffic("out_fused_fwd_loss_dh", _args(hidden=64), None) == (None, None)      # another shape than the profiled one
    assert bench.pmc_traffic("out_fused_fwd_loss_dh", _args(input="multihot"), None) == (None, None)
