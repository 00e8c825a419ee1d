"""The metric oracle against the reference's committed per-instance eval results (pytrec_eval output the authors committed)."""
import numpy as np

from conftest import golden
from oracle import metric_oracle as MO


def test_metric_oracle_matches_committed_reference_results():
    g = golden("g10_metrics")
    names = [str(n) for n in g["names"]]
    assert len(names) >= 10
    for name in names:
        cols, table = MO.instance_table(g[f"{name}.y_pred"], g[f"{name}.truth_indptr"], g[f"{name}.truth_indices"], g[f"{name}.skill_indptr"],
                                        g[f"{name}.skill_indices"], g[f"{name}.cov_indptr"], g[f"{name}.cov_indices"], topK=1000)
        assert cols == [str(c) for c in g[f"{name}.columns"]], name
        np.testing.assert_allclose(table, g[f"{name}.expected"], atol=6e-6, err_msg=name)  # the csv keeps 5 decimals
