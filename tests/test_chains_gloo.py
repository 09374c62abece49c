"""The multi-GPU path (N>1) on CPU: world_size-2 gloo processes shard independent subjects with no data-path
collective and reduce once at the end.  The evaluator here is the CPU oracle (tests may use it); on the GPU box the
same harness drives libnmgp_hip.so (bench.py --workload subjects)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

NUM_SUBJECTS, N, M = 7, 12, 2


def _evaluate_factory():
    from nonstationary_multivariate_gaussian_process_amd import sim
    from oracle import nmgp_oracle as O

    def evaluate(s):
        if s == 5:
            raise RuntimeError("subject 5 is broken on purpose")
        d = sim.simulate_nonseparable(N, M, seed=s)
        return O.nlogpos_obj_SVC(d["pars_true"], d["Y"], d["x"], **sim.HYPER_SVC, verbose=True)
    return evaluate


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from nonstationary_multivariate_gaussian_process_amd import chains
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = chains.partition(NUM_SUBJECTS, world, rank)
    rows = chains.run_local(mine, _evaluate_factory(), evals_per_subject=2)
    stats, table = chains.reduce_rows(rows, NUM_SUBJECTS, world)
    q.put((rank, mine, stats, table))
    dist.barrier()
    dist.destroy_process_group()


def test_partition_is_a_disjoint_cover():
    from nonstationary_multivariate_gaussian_process_amd import chains
    for world in (1, 2, 3, 8):
        parts = [chains.partition(64, world, r) for r in range(world)]
        assert sorted(sum(parts, [])) == list(range(64))
        assert max(map(len, parts)) - min(map(len, parts)) <= 1
    assert chains.partition(64, 8, 3) == list(range(3, 64, 8))     # 8 subjects per GPU, config 4
    with pytest.raises(ValueError):
        chains.partition(4, 2, 2)


def test_two_rank_gloo_sharding_and_reduction():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    got.sort(key=lambda t: t[0])
    assert got[0][1] == [0, 2, 4, 6] and got[1][1] == [1, 3, 5]
    # every rank holds the same reduced result
    assert np.array_equal(got[0][2], got[1][2]) and np.array_equal(got[0][3], got[1][3])
    stats, table = got[0][2], got[0][3]
    ev = _evaluate_factory()
    serial = {s: np.array(ev(s)) for s in range(NUM_SUBJECTS) if s != 5}
    assert stats[0] == NUM_SUBJECTS - 1 and stats[1] == 1 and stats[2] == 2 * (NUM_SUBJECTS - 1)
    assert abs(stats[3] - sum(v[0] for v in serial.values())) < 1e-9 * abs(stats[3])
    assert table.shape == (NUM_SUBJECTS, 8) and list(table[:, 0]) == list(range(NUM_SUBJECTS))
    for s_id, v in serial.items():
        assert table[s_id, 1] == 1 and np.array_equal(table[s_id, 3:8], v)
    assert table[5, 1] == 0 and np.isinf(table[5, 3])               # failed subject is excluded, not fatal


# ---- the MAP job of config 4 sharded over two ranks (CPU evaluator: the oracle's value + gradient) -----------------------
MAP_SUBJECTS, MAP_N, MAP_M, MAP_STEPS = 5, 10, 2, 6


def _map_inputs():
    from nonstationary_multivariate_gaussian_process_amd import sim
    subs = [sim.simulate_nonseparable(MAP_N, MAP_M, seed=40 + s) for s in range(MAP_SUBJECTS)]
    xs = np.stack([d["x"] for d in subs])
    Ys = np.stack([d["Y"] for d in subs])
    p0 = np.stack([sim.perturb(d["pars_true"], 0.05, 0.2 * s) for s, d in enumerate(subs)])
    return xs, Ys, p0


def _oracle_map_factory():
    from nonstationary_multivariate_gaussian_process_amd.drivers import LockStepMAP
    from oracle import nmgp_oracle as O

    class OracleMAP(LockStepMAP):
        def __init__(self, xs, Ys, hyper, pars):
            super().__init__(pars, lr=0.05)
            self.xs, self.Ys, self.hyper = xs, Ys, hyper

        def value_and_grad(self, P):
            outs, grads = [], []
            for b in range(P.shape[0]):
                r, g = O.nlogpos_obj_SVC(P[b], self.Ys[b], self.xs[b], **self.hyper, verbose=True, grad=True)
                outs.append(r)
                grads.append(g)
            return np.array(outs), np.array(grads), np.zeros(P.shape[0], dtype=np.int32)
    return OracleMAP


def _map_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from nonstationary_multivariate_gaussian_process_amd import chains, sim
    dist.init_process_group("gloo", rank=rank, world_size=world)
    xs, Ys, p0 = _map_inputs()
    OracleMAP = _oracle_map_factory()
    pars, rows, stats, table = chains.map_subjects(list(range(MAP_SUBJECTS)), xs, Ys, p0, sim.HYPER_SVC, world, rank,
                                                   N_opt=MAP_STEPS, make_map=lambda a, b, h, p: OracleMAP(a, b, h, p))
    q.put((rank, pars, stats, table))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_map_job_equals_the_serial_run():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_map_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert np.array_equal(got[0][3], got[1][3]) and np.array_equal(got[0][2], got[1][2])
    stats, table = got[0][2], got[0][3]
    assert stats[0] == MAP_SUBJECTS and stats[1] == 0 and table.shape == (MAP_SUBJECTS, 8)
    # serial run of all subjects in one lock-step batch: rows are independent, so the sharded job must reproduce it exactly
    from nonstationary_multivariate_gaussian_process_amd import sim
    xs, Ys, p0 = _map_inputs()
    OracleMAP = _oracle_map_factory()
    last = {}
    serial = OracleMAP(xs, Ys, sim.HYPER_SVC, p0)
    pars, hist, alive = serial.run(MAP_STEPS, callback=lambda i, h, out: last.update(out=out))
    assert np.all(alive) and np.array_equal(table[:, 3:8], last["out"])
    assert np.array_equal(got[0][1], pars[[0, 2, 4]]) and np.array_equal(got[1][1], pars[[1, 3]])
    assert np.all(np.isfinite(hist))
