"""Independent per-subject chains sharded over the GPUs of one node.

The reference's "distributed" variants use MPI only to map a rank to a subject / data file and never communicate
(``Nonseparable_model_mpisim.py:41-43,305-306``, ``Nonseparable_model_distributed.py:239-264``); results are combined
afterwards by an offline script that reads per-subject pickles (``Post_Process/past/reduce_distributed_results.py:53-74``).
Here: one process per GPU (``torch.distributed``; backend ``nccl`` = RCCL over xGMI on the GPU box, ``gloo`` on CPU for
tests), a static subject -> rank partition with NO collective on the evaluation path, per-subject failure isolation
(the reference wraps ``train()`` in try/except -> NegLog = inf, ``Nonseparable_model_mpisim.py:330-334``), and ONE
reduction at the end: an all-reduce(sum) of a small statistics vector plus an all-gather of the per-subject rows.
"""
from __future__ import annotations

import math

import numpy as np

ROW = 8   # per-subject row: [subject_id, ok, evals, NegLog, loglik, lp_l, lp_uL, lp_s2]


def partition(num_subjects, world_size, rank):
    """Subjects owned by `rank`: s mod world_size == rank (round-robin keeps the load even when subjects differ in N)."""
    if not (0 <= rank < world_size):
        raise ValueError("rank %d outside world of size %d" % (rank, world_size))
    return [s for s in range(num_subjects) if s % world_size == rank]


def run_local(subjects, evaluate, evals_per_subject=1):
    """Run `evaluate(subject_id) -> 5-vector (NegLog, loglik, lp_l, lp_uL, lp_s2)` for every local subject.
    A failing subject is recorded (ok = 0, NegLog = inf) and does not stop the others."""
    rows = np.zeros((len(subjects), ROW))
    for k, s in enumerate(subjects):
        rows[k, 0] = s
        try:
            out = None
            for _ in range(evals_per_subject):
                out = np.asarray(evaluate(s), dtype=np.float64)
            rows[k, 1] = 1.0
            rows[k, 2] = evals_per_subject
            rows[k, 3:8] = out[:5]
        except Exception:     # noqa: BLE001 -- isolate the subject, keep the job alive
            rows[k, 1] = 0.0
            rows[k, 3] = math.inf
    return rows


def reduce_rows(rows, num_subjects, world_size, device=None):
    """The only collective step.  Returns (stats, table) on every rank:
    stats = [subjects_ok, subjects_failed, evals_done, sum NegLog over ok subjects] (all-reduce sum),
    table = [num_subjects, ROW] ordered by subject id (all-gather of fixed-size, padded blocks)."""
    import torch
    import torch.distributed as dist
    ok = rows[:, 1] > 0
    stats = np.array([ok.sum(), (~ok).sum(), rows[:, 2].sum(), rows[ok, 3].sum()], dtype=np.float64)
    per_rank = (num_subjects + world_size - 1) // world_size
    block = np.full((per_rank, ROW), -1.0)
    block[:rows.shape[0]] = rows
    # (a single rank with an initialised process group -- bench.py --dist-selftest -- goes through the collectives too)
    if not (dist.is_available() and dist.is_initialized()):
        table = block
        gathered = [block]
    else:
        dev = device if device is not None else "cpu"
        t_stats = torch.from_numpy(stats).to(dev)
        dist.all_reduce(t_stats, op=dist.ReduceOp.SUM)
        stats = t_stats.cpu().numpy()
        t_block = torch.from_numpy(block).to(dev)
        outs = [torch.empty_like(t_block) for _ in range(world_size)]
        dist.all_gather(outs, t_block)
        gathered = [o.cpu().numpy() for o in outs]
    table = np.concatenate(gathered, 0)
    table = table[table[:, 0] >= 0]
    table = table[np.argsort(table[:, 0], kind="stable")]
    return stats, table


def map_subjects(subjects, xs, Ys, init_pars, hyper_pars, world_size, rank, N_opt=2000, lr=2e-1, device=None, make_map=None):
    """Config-4 style job: the MAP loop for every subject, sharded over the ranks.  `subjects` = the global subject ids whose
    data are given (xs [S, N], Ys [S, N, M], init_pars [S, P], in that order); this rank advances its share
    (``partition``) in lock-step as ONE multi-subject batch (``drivers.BatchedMAP``; `make_map(xs, Ys, hyper, pars)` lets a
    test substitute a CPU evaluator) and the per-subject results meet in the single reduction (``reduce_rows``).
    Returns (pars_local [S_local, P], rows_local, stats, table)."""
    mine = [k for k, s_id in enumerate(subjects) if s_id % world_size == rank]
    rows = np.zeros((len(mine), ROW))
    pars_local = np.zeros((len(mine), np.asarray(init_pars).shape[1]))
    if mine:
        if make_map is None:
            from .drivers import BatchedMAP
            make_map = lambda a, b, h, p: BatchedMAP(a, b, h, p, lr=lr)      # noqa: E731
        bm = make_map(np.asarray(xs)[mine], np.asarray(Ys)[mine], hyper_pars, np.asarray(init_pars)[mine])
        last = {}
        pars_local, hist, alive = bm.run(N_opt, callback=lambda i, h, out: last.update(out=out))
        for k, idx in enumerate(mine):
            rows[k, 0] = subjects[idx]
            rows[k, 1] = 1.0 if alive[k] else 0.0
            rows[k, 2] = N_opt
            # (N_opt = 0: no iteration ran, there is no objective value to report -- inf, like a subject that failed)
            rows[k, 3:8] = last["out"][k] if (alive[k] and "out" in last) else [math.inf, 0.0, 0.0, 0.0, 0.0]
    stats, table = reduce_rows(rows, len(subjects), world_size, device=device)
    return pars_local, rows, stats, table
