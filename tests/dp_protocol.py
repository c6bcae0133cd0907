"""TEST INFRASTRUCTURE (moved out of the product package in round 2: nothing in the product imports it).

Data-parallel protocol of the PPO update, stated on the host.

libdril_hip.so runs this protocol in C++ on its own stream (dril_api.hip: ppo_step) with RCCL all-reduces; this
module states the same protocol over a pluggable `allreduce(np.ndarray) -> np.ndarray` (sum over ranks) so that the
sharding maths can be exercised with torch.distributed/gloo on CPU (tests/test_distributed_gloo.py) and so that hosts
(bench.py, the Julia shim) share one definition of who owns what.

    envs        rank r owns global env indices [r*E, (r+1)*E); env i is seeded seed + i (wrapper_utils.jl:39-44)
    rollout     no communication (envs and GAE are independent per env)
    minibatch   shard-local permutation: every rank draws B/world positions of ITS buffer shard per optimiser step
                (equivalent in distribution to the reference's global shuffle, ppo.jl:188-195; deviation documented in DESIGN.md)
    moments     (sum a, sum a^2, n) of the global minibatch: one 3-double all-reduce, then mean / corrected std (ppo.jl:350-356)
    gradient    each rank accumulates SUMS with the global 1/B folded in; ONE all-reduce of [grads | loss sums] (P + 8 floats)
    apply       identical norm / clip / KL check / Adam on every rank => replicas stay bit-identical
    normalise   NormalizeWrapperEnv's batch moments cover every env of the job: per env step each rank folds its (sum x, sum x^2) per
                statistic into one 16-double row, ONE all-reduce sums the rows, every rank merges mean / var(corrected = false) with
                n = world * E into its RunningMeanStd (normalizeWrapperEnv.jl:21-50) => the statistics stay identical on all ranks
"""
from __future__ import annotations

import numpy as np


def local_batch_size(batch_size: int, world_size: int) -> int:
    if batch_size % world_size:
        raise ValueError("batch_size must be divisible by world_size")
    return batch_size // world_size


def global_env_index(rank: int, n_envs_local: int, local_env: int) -> int:
    return rank * n_envs_local + local_env


def env_seed(seed: int, rank: int, n_envs_local: int, local_env: int) -> int:
    """seed of a rank's local env: Random.seed!(penv, seed) seeds sub-env i with seed + i - 1 (wrapper_utils.jl:39-44)"""
    return seed + global_env_index(rank, n_envs_local, local_env)


def minibatch_bounds(n_local: int, batch_local: int, k: int):
    """[pos0, pos0 + count) of optimiser step k inside a rank's epoch order; the partial last batch is kept"""
    pos0 = k * batch_local
    return pos0, min(batch_local, n_local - pos0)


def global_moments(adv_local: np.ndarray, allreduce):
    """mean and (std + 1e-8) of the GLOBAL minibatch from local (sum, sumsq, n) — normalize!, ppo.jl:350-356"""
    a = adv_local.astype(np.float64)
    s, q, n = allreduce(np.array([a.sum(), (a * a).sum(), float(a.size)], np.float64))
    mean = s / n
    var = max((q - s * mean) / (n - 1.0), 0.0)
    return np.float32(mean), np.float32(np.float32(np.sqrt(var)) + np.float32(1e-8)), int(n)


def global_batch_moments(x_local: np.ndarray, allreduce):
    """batch mean / var(corrected=false) / count over ALL ranks' rows of x (n_local, dims) from local sums — what the device path feeds
    update_from_moments! (normalizeWrapperEnv.jl:21-50) in data-parallel runs (dril_api.hip: global_partials)"""
    x = np.asarray(x_local, np.float32).reshape(len(x_local), -1).astype(np.float64)
    row = allreduce(np.concatenate([x.sum(0), (x * x).sum(0), [float(x.shape[0])]]))
    d = x.shape[1]; n = row[-1]
    mean = row[:d] / n
    var = np.maximum(row[d:2 * d] / n - mean * mean, 0.0)
    return mean.astype(np.float32), var.astype(np.float32), int(n)


def rms_merge(mean, var, count, bmean, bvar, bcount):
    """update_from_moments! (normalizeWrapperEnv.jl:28-50) in f32"""
    mean, var, bmean, bvar = (np.asarray(a, np.float32) for a in (mean, var, bmean, bvar))
    if count == 0:
        return bmean.copy(), bvar.copy(), bcount
    tot = count + bcount
    delta = bmean - mean
    new_mean = mean + delta * np.float32(bcount) / np.float32(tot)
    m2 = var * np.float32(count) + bvar * np.float32(bcount) + delta * delta * np.float32(count) * np.float32(bcount) / np.float32(tot)
    return new_mean.astype(np.float32), (m2 / np.float32(tot)).astype(np.float32), tot


def data_parallel_gradient(local_loss_grad, batch_local, n_global: int, allreduce):
    """One optimiser step's gradient.  `local_loss_grad(batch) -> (loss, stats7, grads)` must return MEANS over the local
    batch with advantages already normalised (it is the single-rank entry point dril_ppo_loss_grad / the oracle's twin);
    means are turned back into sums, all-reduced once as [grads | sums | n], and divided by the global count."""
    loss, stats, grads = local_loss_grad(batch_local)
    n_local = float(len(batch_local[2]))
    flat = np.concatenate([grads.astype(np.float64) * n_local, stats.astype(np.float64) * n_local, [n_local]])
    flat = allreduce(flat)
    n = flat[-1]
    assert int(n) == n_global
    return (flat[:grads.size] / n).astype(np.float32), (flat[grads.size:-1] / n).astype(np.float32)
