#!/usr/bin/env python3
"""PPO on YOUR OWN envs: the envs step on the host (any object with the reference's AbstractEnv verbs, interfaces/environments.jl:21-39),
the agent lives on the device (DRIL_ENV_EXTERNAL: any observation / action / hidden width).

    env = BroadcastedParallelEnv([MyEnv() for _ in 1:32])   ->  HostParallelEnv([MyEnv() for _ in range(32)])
    train!(agent, env, alg, max_steps)

usage: python examples/ppo_host_envs.py [n_envs=32] [iterations=40]"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import __graft_entry__ as g

pkg = g.load_package()


class ReacherEnv:
    """two-link arm in the plane: 8-dim observation (cos/sin of both joints, joint velocities, target), 2 torques in Box(-1, 1),
    reward = -distance(fingertip, target) - 0.01 |a|^2, 50-step episodes"""

    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.reset_()

    def observation_space(self):
        return pkg.Box(low=(-1.0,) * 4 + (-10.0,) * 2 + (-2.0,) * 2, high=(1.0,) * 4 + (10.0,) * 2 + (2.0,) * 2)

    def action_space(self):
        return pkg.Box(low=(-1.0, -1.0), high=(1.0, 1.0))

    def reset_(self):
        self.q = self.rng.uniform(-np.pi, np.pi, 2); self.dq = np.zeros(2); self.t = 0
        self.target = self.rng.uniform(-1.4, 1.4, 2)

    def _tip(self):
        return np.array([np.cos(self.q[0]) + np.cos(self.q.sum()), np.sin(self.q[0]) + np.sin(self.q.sum())])

    def observe(self):
        return np.concatenate([np.cos(self.q), np.sin(self.q), self.dq, self.target]).astype(np.float32)

    def act_(self, a):
        a = np.asarray(a, np.float64)
        self.dq = np.clip(0.9 * self.dq + 0.5 * a, -10, 10); self.q = self.q + 0.1 * self.dq; self.t += 1
        return float(-np.linalg.norm(self._tip() - self.target) - 0.01 * (a ** 2).sum())

    def terminated(self):
        return False

    def truncated(self):
        return self.t >= 50


n_envs = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
env = pkg.HostParallelEnv([ReacherEnv(s) for s in range(n_envs)])
alg = pkg.PPO(n_steps=100, batch_size=n_envs * 100 // 4, epochs=8, learning_rate=1e-3)
agent = pkg.Agent(pkg.ActorCriticLayer(env.observation_space(), env.action_space(), hidden_dims=(96, 64), log_std_init=-0.5), alg, seed=0)
buf = pkg.RolloutBuffer(alg.n_steps, n_envs, alg.gae_lambda, alg.gamma)
pkg.collect_rollout_(buf, agent, alg, env)
print(f"before: mean reward per step {buf.rewards.mean():.3f}")
stats, timer = pkg.train_(agent, env, alg, iters * alg.n_steps * n_envs)
pkg.collect_rollout_(buf, agent, alg, env)
print(f"after {iters} iterations ({timer['training_loop']:.1f} s: rollouts on the host {timer['collect_rollout']:.1f} s, updates on the device {timer['epoch loop']:.2f} s): "
      f"mean reward per step {buf.rewards.mean():.3f}, explained variance {stats['explained_variances'][-1]:.2f}")
