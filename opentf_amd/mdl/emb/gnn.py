"""`Gnn` with the node2vec method (`data.embedding.class_method=mdl.amd.emb.gnn.Gnn_n2v`): the reference's team2vec plugin
(src/mdl/emb/gnn.py) for its random-walk branch, trained on the MI355X, and the bridge that puts the CSR gather of
`get_dense_vecs` on the hot path of the UNMODIFIED caller.

What the caller (src/main.py:100-153) does and what it gets here:
  * `t2v = cls(output, acceleration, seed, cfg, method)`; `t2v.learn(teamsvecs, splits)`                       (main.py:112,122)
      per fold: the 'stm' graph (skill - team - member) without the member-team edges of the test and of that fold's validation teams
      (gnn.py:84-116), `torch_geometric.nn.Node2Vec` (p = q = 1) trained by `_train_rw` (gnn.py:401-453: shuffled node batches, Adam,
      ReduceLROnPlateau on v_loss, EarlyStopping, `f{k}.pt` / `f{k}.e{e}.pt` checkpoints holding `embedding.weight`).  Here the walks,
      the skip-gram loss, its gradient and the dense Adam run in `opentf_amd/csrc/ntf_n2v.hip`; control flow, file names and
      checkpoint keys are the reference's.  An existing `f{k}.pt` is loaded instead, as gnn.py:402-405 does: one written by THIS plugin carries the node order its rows are
      sliced by; one the reference trained follows the node-store order of ITS pickled graph (`{structure}.{dup_edge}.graph.pkl` beside the splits directory,
      gnn.py:21-23) - that file is read (opentf_amd/mdl/emb/pyg_reader.py: a restricted unpickler, no torch_geometric) and the rows are re-stacked; without the graph file
      such a table is refused rather than sliced wrong.
  * `skill_vecs = t2v.get_dense_vecs(teamsvecs, vectype='skill')`                                                  (main.py:148)
      the mean-pool `(skill @ E_skill) / skill.sum(1)` (gnn.py:484-486) by the gather kernel, returned as the dense [N, d] matrix the
      caller stores in `teamsvecs['skill']` — AND `teamsvecs['skill_table'] = E_skill` is registered, so that the Fnn / Bnn plugin of this
      package (opentf_amd/mdl/fnn.py) finds the table and runs the gather INSIDE every training step from `original_skill`
      (which main.py:152 stores itself) instead of reading pre-pooled rows.

Node ids of the homogeneous graph: [skills | members | teams].  (PyG's HeteroData orders the three blocks by Python set iteration, i.e.
arbitrarily per process; consumers select by type, gnn.py:505-509, so the order is not part of the contract.)
"""
from __future__ import annotations

import logging
import os
import re
import time

import numpy as np
import scipy.sparse

from ..earlystopping import EarlyStopping, PlateauLR
from ..fnn import index_order, parse_devices
from ..ntf import cfg_get, dist_rank, summary_writer
from .t2v import T2v, gather_meanpool

log = logging.getLogger(__name__)


def _csr(mat):
    m = scipy.sparse.csr_matrix(mat)
    m.sort_indices()
    return m.indptr.astype(np.int64), m.indices.astype(np.int64)


def stm_graph(skill, member, drop_teams=()):
    """(rowptr int64, col int32, offsets, num_nodes) of the undirected skill-team-member graph; member-team edges of `drop_teams` removed"""
    s_ip, s_ix = _csr(skill)
    m_ip, m_ix = _csr(member)
    n_team, S, M = skill.shape[0], skill.shape[1], member.shape[1]
    off = {"skill": 0, "member": S, "team": S + M}
    n = S + M + n_team
    team_of_s = np.repeat(np.arange(n_team, dtype=np.int64), np.diff(s_ip))
    team_of_m = np.repeat(np.arange(n_team, dtype=np.int64), np.diff(m_ip))
    keep = ~np.isin(team_of_m, np.asarray(list(drop_teams), dtype=np.int64))
    src = np.concatenate([s_ix + off["skill"], m_ix[keep] + off["member"]])
    dst = np.concatenate([team_of_s + off["team"], team_of_m[keep] + off["team"]])
    a, b = np.concatenate([src, dst]), np.concatenate([dst, src])
    order = np.lexsort((b, a))
    a, b = a[order], b[order]
    rowptr = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rowptr, a + 1, 1)
    return np.cumsum(rowptr), b.astype(np.int32), off, n


def member_team_edges(member, teams, off):
    """homogeneous (member node, team node) pairs of the given teams, both directions (gnn.py:100-104,118-124)"""
    sub = scipy.sparse.csr_matrix(member)[np.asarray(teams, dtype=np.int64)]
    t = np.repeat(np.asarray(teams, dtype=np.int64), np.diff(sub.indptr)) + off["team"]
    m = sub.indices.astype(np.int64) + off["member"]
    return np.concatenate([m, t]), np.concatenate([t, m])


NODE_ORDER = "skill|member|team"     # row blocks of the table this plugin trains (stm_graph)


def _barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


_WAIT_GROUP = None


def _wait_group():
    """a gloo group of its own with a week's timeout: the ranks other than 0 sit in a collective on it for as long as rank 0 trains a fold (up to cfg.e epochs) -
    on the default group that wait is cut by the process group's watchdog after 10 (nccl) / 30 (gloo) minutes and the job is torn down"""
    global _WAIT_GROUP
    import datetime
    import torch.distributed as dist
    if _WAIT_GROUP is None:
        _WAIT_GROUP = dist.new_group(backend="gloo", timeout=datetime.timedelta(days=7))       # a collective itself: learn() creates it up front, where the ranks are together
    return _WAIT_GROUP


def _sync_torch_rng():
    """rank 0's torch CPU generator state on every rank (a collective: also orders the ranks behind rank 0's file writes); on the long-timeout group, see _wait_group"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return
    box = [torch.get_rng_state() if dist.get_rank() == 0 else None]
    dist.broadcast_object_list(box, src=0, group=_wait_group(), device=torch.device("cpu"))
    if dist.get_rank() != 0: torch.set_rng_state(box[0])


def _rank0_says(flag):
    """rank 0's value of a bool on every rank (file-system checks that must come out the same job-wide)"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return bool(flag)
    box = [bool(flag)]
    dist.broadcast_object_list(box, src=0)
    return box[0]


class Gnn(T2v):
    def __init__(self, output, device, seed, cfg, model):
        super().__init__(output, device, seed, cfg, model)
        if self.name != "n2v":
            raise NotImplementedError(f"opentf_amd trains the node2vec table on the device; '{self.name}' stays the reference's mdl.emb.gnn.Gnn")
        self.writer = summary_writer()
        self.offsets = None

    def _mcfg(self):
        return cfg_get(self.cfg, self.name)

    def _dirname(self):
        m, g = self._mcfg(), cfg_get(self.cfg, "graph")
        structure = cfg_get(g, "structure")
        name = (f"/{self.name}.b{cfg_get(m, 'b')}.e{cfg_get(m, 'e')}.ns{cfg_get(m, 'ns')}.lr{cfg_get(m, 'lr')}.es{cfg_get(m, 'es')}.spe{cfg_get(m, 'spe')}"
                f".d{cfg_get(m, 'd')}.{cfg_get(g, 'dup_edge')}.{structure[1]}")
        name += ".pre" if cfg_get(g, "pre") else ""
        return name + f".w{cfg_get(m, 'w')}.wl{cfg_get(m, 'wl')}.wn{cfg_get(m, 'wn')}"

    def learn(self, teamsvecs, splits=None, time_indexes=None):
        import torch
        from ... import libntf
        m = self._mcfg()
        structure = cfg_get(cfg_get(self.cfg, "graph"), "structure")
        if structure[1] != "stm":
            raise NotImplementedError("only the skill-team-member ('stm') graph structure is built here")
        self.graph_file = self.output + f"/../{structure[1]}.{cfg_get(cfg_get(self.cfg, 'graph'), 'dup_edge') or 'dup'}.graph.pkl"      # gnn.py:21
        self.output += self._dirname()
        os.makedirs(self.output, exist_ok=True)
        skill = teamsvecs.get("original_skill", teamsvecs["skill"]) if hasattr(teamsvecs, "get") else teamsvecs["skill"]
        member = teamsvecs["member"]
        d = int(cfg_get(m, "d"))
        w = None
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            _barrier(); _wait_group()      # the long-timeout group is made HERE, all ranks arriving together: made lazily inside _sync_torch_rng, rank 0 would reach new_group only
                                           # after training the whole first fold, and the others' wait inside gloo's connect would be cut by the store's own (short) timeout
        for foldidx in splits["folds"].keys():
            drop = np.concatenate([np.asarray(splits["test"]), np.asarray(splits["folds"][foldidx]["valid"])])
            rowptr, col, off, n = stm_graph(skill, member, drop)
            self.offsets = off
            init = torch.nn.Embedding(n, d).weight.detach().numpy()       # the draw Node2Vec's constructor makes (gnn.py:153-160)
            path = f"{self.output}/f{foldidx}.pt"
            _barrier()                                                     # torchrun: nobody looks for the file while rank 0 may still be writing the previous one
            if not _rank0_says(os.path.exists(path)):                      # gnn.py:402-405: a trained table is loaded, not retrained (rank 0 decides for the job)
                # ONE trainer (as d2v.py): the table's gradient sums are f32 atomics - run per rank, losses, early-stop epochs and tables would differ between
                # the ranks - and no collective sits inside a loop whose trip count a rank decides for itself; the other ranks wait below and read rank 0's file
                if dist_rank() == 0:
                    if w is None: w = self.writer(log_dir=f"{self.output}/logs4tboard/run_{int(time.time())}")
                    self._train_fold(libntf, rowptr, col, init, n, member, splits, foldidx, off, m, w, path)
                _sync_torch_rng()                                          # (also the barrier) the trainer's loader shuffles consumed torch's CPU generator: the next fold's initial draw stays job-wide
            self.model = self._load_fold(path, off, n)
        if w is not None: w.close()

    def _train_fold(self, libntf, rowptr, col, init, n, member, splits, foldidx, off, m, w, path):
        """`_train_rw` (gnn.py:401-453) for one fold on this rank's device; writes f{k}.e{e}.pt / f{k}.pt"""
        b, lr = int(cfg_get(m, "b")), float(cfg_get(m, "lr"))
        wl, ctx, wn, ns = int(cfg_get(m, "wl")), int(cfg_get(m, "w")), int(cfg_get(m, "wn")), int(cfg_get(m, "ns"))
        spe = cfg_get(m, "spe")
        val_src, val_dst = member_team_edges(member, splits["folds"][foldidx]["valid"], off)
        assert len(val_src), "Empty valid member-team edge set!"
        net = libntf.Node2Vec(rowptr, col, init, seed=int(self.seed or 0) + 7919 * int(foldidx), device=parse_devices(self.device)[0])
        scheduler = PlateauLR(lr, factor=0.1, patience=2)
        earlystopping = EarlyStopping(patience=int(cfg_get(m, "es")), verbose=True, delta=0, trace_func=log.info)
        cur_lr, e, t_loss, v_loss = lr, -1, 0.0, 0.0
        for e in range(int(cfg_get(m, "e"))):
            order = index_order(n, b, True)                            # DataLoader(range(num_nodes), batch_size=b, shuffle=True)
            nb, t_loss = 0, 0.0
            for o in range(0, n, b):
                t_loss += net.train_batch(order[o:o + b], wl, ctx, wn, ns, cur_lr); nb += 1
            t_loss /= nb
            v_loss = net.edge_bce(val_src, val_dst) / len(val_src)     # the reference divides the mean once more (gnn.py:433)
            w.add_scalar(tag=f"{foldidx}_t_loss", scalar_value=t_loss, global_step=e)
            w.add_scalar(tag=f"{foldidx}_v_loss", scalar_value=v_loss, global_step=e)
            log.info(f"Fold {foldidx}/{len(splits['folds']) - 1}, Epoch {e}, Train Loss: {t_loss:.4f}, Valid Loss: {v_loss:.4f}")
            if spe and (e == 0 or ((e + 1) % spe) == 0):
                self._save(net.weight(), foldidx, e, t_loss, v_loss, f"{self.output}/f{foldidx}.e{e}.pt")
            cur_lr = scheduler.step(v_loss)
            if earlystopping(v_loss, None).early_stop:
                log.info(f"Early stopping triggered at epoch: {e}")
                break
        weight = net.weight()
        net.close()
        self._save(weight, foldidx, e, t_loss, v_loss, path)

    def _load_fold(self, path, off, n):
        """gnn.py:402-405.  A table THIS plugin saved carries its node order ([skills | members | teams], stm_graph).  A table the reference trained follows the order in
        which its pickled HeteroData holds the node stores (src/mdl/emb/gnn.py:29-47: the iteration order of a Python set, recorded only in `*.graph.pkl`): that file is
        read and the row blocks are re-stacked; a foreign table without its graph file is refused - sliced by the wrong order it would train silently on the wrong rows."""
        from . import pyg_reader
        ref = pyg_reader.reference_table(path)      # restricted unpickler, the ONLY load of the file: the pickled cfg (omegaconf or not) resolves to inert holders, this plugin's
                                                    # node_order / node_offsets markers are plain builtins and come through it - plugin or reference is decided from those alone
        if ref["node_order"] is not None and ref["node_order"] != NODE_ORDER:
            raise RuntimeError(f"{path}: node order {ref['node_order']!r}, this build writes and reads {NODE_ORDER!r}")
        if ref["node_order"] == NODE_ORDER:
            if ref["node_offsets"] != (off["skill"], off["member"], off["team"], n):
                raise RuntimeError(f"{path}: written for node offsets {ref['node_offsets']}, teamsvecs gives {(off['skill'], off['member'], off['team'], n)}")
            log.info(f"Loading the model {path} ...")
            return ref["weight"]
        graph = getattr(self, "graph_file", None)
        if not graph or not os.path.exists(graph):
            raise RuntimeError(f"{path} was not written by opentf_amd.mdl.emb.gnn and the graph file that holds its node order ({graph}) is missing: "
                               "remove the table to retrain it on the device, or put the reference's *.graph.pkl back")
        blocks = pyg_reader.node_blocks(graph)
        weight, counts = pyg_reader.blocks_to_order(ref["weight"], blocks, NODE_ORDER.split("|"))
        want = {"skill": off["member"] - off["skill"], "member": off["team"] - off["member"], "team": n - off["team"]}
        if counts != want:
            raise RuntimeError(f"{path}: the graph {graph} has {counts} nodes, teamsvecs has {want}")
        log.info(f"Loading the model {path} (trained by the reference: rows re-stacked from {[t for t, _ in blocks]} of {os.path.basename(graph)}) ...")
        return weight

    def _save(self, weight, foldidx, e, t_loss, v_loss, path):
        """keys and order of gnn.py:445,453, + the node order the rows are sliced by; written whole, then moved into place.  Called by the training rank only: no collective"""
        import torch
        off = self.offsets
        tmp = f"{path}.tmp.{os.getpid()}"
        torch.save({"model_state_dict": {"embedding.weight": torch.from_numpy(np.ascontiguousarray(weight))}, "cfg": self.cfg, "f": foldidx, "e": e,
                    "t_loss": t_loss, "v_loss": v_loss, "node_order": NODE_ORDER,
                    "node_offsets": (off["skill"], off["member"], off["team"], int(weight.shape[0]))}, tmp)
        os.replace(tmp, path)

    def _node_emb(self, teamsvecs, node_type):
        if self.model is None:
            raise RuntimeError("no trained table: call learn() first")
        skill = teamsvecs.get("original_skill", teamsvecs["skill"]) if hasattr(teamsvecs, "get") else teamsvecs["skill"]
        S, M = skill.shape[1], teamsvecs["member"].shape[1]
        off = {"skill": (0, S), "member": (S, S + M), "team": (S + M, self.model.shape[0])}[node_type]
        return self.model[off[0]:off[1]]

    def get_dense_vecs(self, teamsvecs, vectype="skill"):
        """gnn.py:484-486.  For 'skill' the table is also registered as teamsvecs['skill_table'] (see the module docstring)."""
        if vectype not in ("skill", "member") or vectype not in teamsvecs:
            return self._node_emb(teamsvecs, vectype)
        table = np.ascontiguousarray(self._node_emb(teamsvecs, vectype), dtype=np.float32)
        rows = teamsvecs.get("original_skill", teamsvecs[vectype]) if (vectype == "skill" and hasattr(teamsvecs, "get")) else teamsvecs[vectype]
        dense = gather_meanpool(rows, table, self.device)
        if vectype == "skill":
            teamsvecs["skill_table"] = table
        return dense
