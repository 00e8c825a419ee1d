#!/usr/bin/env python3
"""Fixture g14: the node2vec tables the reference's authors COMMITTED for toy dblp (output/dblp/toy.dblp.v12.json/splits.f3.r0.85/n2v.*/f{k}.pt):
`embedding.weight` [54, 128] (10 skill + 13 member + 31 team nodes, block order as PyG's HeteroData happened to hold them), with the epoch /
loss fields.  Data only; build container only.      python tests/golden/make_golden_n2v.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden_bnn import REF, tload  # noqa: E402  (stub unpickler for the omegaconf cfg inside the checkpoints)


def main():
    d = f"{REF}/dblp/toy.dblp.v12.json/splits.f3.r0.85/n2v.b1000.e100.ns5.lr0.001.es5.spe10.d128.add.stm.w5.wl5.wn10"
    arrs = {"dirname": os.path.basename(d)}
    for k in range(3):
        ck = tload(f"{d}/f{k}.pt")
        assert list(ck["model_state_dict"].keys()) == ["embedding.weight"]
        arrs[f"f{k}.embedding.weight"] = ck["model_state_dict"]["embedding.weight"].numpy()
        arrs[f"f{k}.e"] = ck["e"]; arrs[f"f{k}.t_loss"] = ck["t_loss"]; arrs[f"f{k}.v_loss"] = ck["v_loss"]
        arrs[f"f{k}.keys"] = np.array(list(ck.keys()))
    np.savez_compressed(f"{HERE}/g14_n2v_dblp.npz", **arrs)
    print({k: (v.shape if hasattr(v, "shape") else v) for k, v in arrs.items()})


if __name__ == "__main__":
    main()
