"""CPU, world_size 2, gloo: the data-parallel protocol of the PPO update (tests/dp_protocol.py — a host statement of what dril_api.hip does in C++; the LIBRARY's own world_size > 1 code is executed by tests/test_gpu_dataparallel.py; mirrored in C++ by
dril_api.hip: ppo_step) reproduces the single-process result.  The compute backend here is the CPU oracle (the checker);
the all-reduce is torch.distributed over gloo on 127.0.0.1."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), ORACLE_THREADS="2")
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
    import __graft_entry__ as g
    pkg = g.load_package()
    import oracle_lib
    import dp_protocol as D
    dist.init_process_group("gloo", rank=rank, world_size=world)
    capi = pkg._capi

    def allreduce(x):
        t = torch.from_numpy(np.ascontiguousarray(x, np.float64).copy()); dist.all_reduce(t); return t.numpy()

    E, T, B, epochs = 6, 10, 40, 2                       # global: 12 envs, N = 120, 3 optimiser steps per epoch
    cfg = capi.default_config(capi.ENV_CARTPOLE)
    cfg.n_envs, cfg.n_steps, cfg.batch_size, cfg.epochs, cfg.episode_len = E, T, B, epochs, 7
    cfg.rank, cfg.world_size, cfg.seed = rank, world, 11
    o = oracle_lib.Oracle(cfg)
    flat = (np.random.default_rng(5).standard_normal(o.P) * 0.3).astype(np.float32)
    o.set_params(flat); o.env_reset(cfg.seed); o.collect_rollout()        # rollout + GAE: no communication
    st0, _ = o.env_get_state()
    cfg_n = capi.default_config(capi.ENV_CARTPOLE)                         # local-gradient backend: advantages pre-normalised
    cfg_n.n_envs, cfg_n.n_steps, cfg_n.normalize_advantage = 2, 2, 0
    og = oracle_lib.Oracle(cfg_n)
    bufs = {k: o.buffer(getattr(capi, "BUF_" + k)) for k in ("OBSERVATIONS", "ACTIONS", "ADVANTAGES", "RETURNS", "LOGPROBS", "VALUES")}
    Bl = D.local_batch_size(B, world); Nl = E * T
    perms = [np.random.default_rng(100 * rank + e).permutation(Nl) for e in range(epochs)]
    used = []
    for e in range(epochs):
        for k in range(-(-Nl // Bl)):
            pos0, cnt = D.minibatch_bounds(Nl, Bl, k)
            idx = perms[e][pos0:pos0 + cnt]
            used.append(idx)
            mean, den, n = D.global_moments(bufs["ADVANTAGES"][idx], allreduce)
            advn = ((bufs["ADVANTAGES"][idx] - mean) / den).astype(np.float32)
            batch = (bufs["OBSERVATIONS"][idx], bufs["ACTIONS"][idx], advn, bufs["RETURNS"][idx], bufs["LOGPROBS"][idx], bufs["VALUES"][idx])
            og.set_params(o.get_params())
            grads, stats = D.data_parallel_gradient(lambda b: og.ppo_loss_grad(*b), batch, n, allreduce)
            o.apply_gradients(grads)                                        # identical on every rank
    # NormalizeWrapperEnv statistics over all ranks' envs: three env steps of (E, 3) observations
    rms = (np.zeros(3, np.float32), np.ones(3, np.float32), 0); obs_steps = []
    for step in range(3):
        x = np.random.default_rng(1000 * rank + step).normal(step, 1 + rank, (E, 3)).astype(np.float32)
        obs_steps.append(x)
        bm, bv, n = D.global_batch_moments(x, allreduce)
        rms = D.rms_merge(*rms, bm, bv, n)
    q.put((rank, o.get_params(), {k: v for k, v in bufs.items()}, [u.tolist() for u in used], st0, rms, obs_steps))
    dist.barrier(); dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_data_parallel_update_equals_single_process(pkg, oracle_mod):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    [p.start() for p in procs]
    res = sorted([q.get(timeout=240) for _ in range(2)], key=lambda r: r[0])
    [p.join(60) for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    (r0, p0, b0, u0, s0, rms0, x0), (r1, p1, b1, u1, s1, rms1, x1) = res
    assert np.array_equal(p0, p1)                                           # replicas stay bit-identical
    # NormalizeWrapperEnv: both ranks hold the same RunningMeanStd, and it is the oracle's update! over the union of their envs per step
    assert all(np.array_equal(a, b) for a, b in zip(rms0[:2], rms1[:2])) and rms0[2] == rms1[2] == 3 * 12
    import ctypes as C
    L = oracle_mod.lib(); pp = lambda a: a.ctypes.data_as(C.c_void_p)
    mean, var, cnt = np.zeros(3, np.float32), np.ones(3, np.float32), C.c_int64(0)
    for a, b in zip(x0, x1):
        u = np.ascontiguousarray(np.concatenate([a, b]), np.float32)
        L.orc_rms_update(pp(mean), pp(var), C.byref(cnt), 3, pp(u), u.shape[0])
    np.testing.assert_allclose(rms0[0], mean, rtol=1e-5, atol=1e-6); np.testing.assert_allclose(rms0[1], var, rtol=1e-4, atol=1e-6)
    # env sharding: rank 1's env e is global env 6 + e => same reset state as a 12-env single process (seed + i)
    capi = pkg._capi
    cfg = capi.default_config(capi.ENV_CARTPOLE); cfg.n_envs, cfg.n_steps, cfg.seed = 12, 2, 11
    o = oracle_mod.Oracle(cfg)
    flat = (np.random.default_rng(5).standard_normal(o.P) * 0.3).astype(np.float32)
    # single process over the union of the shards: the global minibatch of step k = rank0's slice ++ rank1's slice
    cfgu = capi.default_config(capi.ENV_CARTPOLE); cfgu.n_envs, cfgu.n_steps = 2, 2
    ou = oracle_mod.Oracle(cfgu); ou.set_params(flat)
    for i0, i1 in zip(u0, u1):
        i0, i1 = np.asarray(i0), np.asarray(i1)
        cat = lambda k: np.concatenate([b0[k][i0], b1[k][i1]])
        loss, stats, grads = ou.ppo_loss_grad(cat("OBSERVATIONS"), cat("ACTIONS"), cat("ADVANTAGES"), cat("RETURNS"), cat("LOGPROBS"), cat("VALUES"))
        ou.apply_gradients(grads)
    np.testing.assert_allclose(p0, ou.get_params(), rtol=2e-5, atol=2e-7)
    assert not np.allclose(p0, flat)


def test_env_shards_are_seeded_by_global_index(pkg, oracle_mod):
    """rank r's local env e == global env r*E + e of a single-process run (wrapper_utils.jl:39-44)"""
    capi = pkg._capi
    cfg = capi.default_config(capi.ENV_CARTPOLE); cfg.n_envs, cfg.n_steps = 8, 2
    o = oracle_mod.Oracle(cfg); o.env_reset(77)
    full, _ = o.env_get_state()
    for rank in range(2):
        c = capi.default_config(capi.ENV_CARTPOLE); c.n_envs, c.n_steps, c.rank, c.world_size, c.batch_size = 4, 2, rank, 2, 2
        s = oracle_mod.Oracle(c); s.env_reset(77)
        assert np.array_equal(s.env_get_state()[0], full[4 * rank:4 * rank + 4])
    import dp_protocol as D
    assert D.env_seed(77, 1, 4, 2) == 77 + 6 and D.local_batch_size(64, 8) == 8
    with pytest.raises(ValueError):
        D.local_batch_size(10, 4)
