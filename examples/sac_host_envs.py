#!/usr/bin/env python3
"""SAC on YOUR OWN continuous-control envs: the envs step on the host, the actor, both critics, the targets, the replay ring and every update! live
on the device (DRIL_ENV_EXTERNAL for SAC: dril_sac_predict_actions + dril_sac_ext_push + dril_sac_update).

    env = BroadcastedParallelEnv([MyEnv() for _ in 1:8])   ->  HostParallelEnv([MyEnv() for _ in range(8)])
    agent, replay_buffer, training_stats, to = train!(agent, replay_buffer, env, alg, max_steps)

usage: python examples/sac_host_envs.py [n_envs=8] [env_steps=12000]"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import __graft_entry__ as g

pkg = g.load_package()


class ReacherEnv:
    """two-link arm in the plane: 8-dim observation, 2 torques in Box(-1, 1), reward = -distance(fingertip, target) - 0.01 |a|^2, 50-step episodes"""

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

    def observe(self):
        return np.concatenate([np.cos(self.q), np.sin(self.q), self.dq, self.target]).astype(np.float32)

    def act_(self, a):
        a = np.asarray(a, np.float64)
        self.dq = np.clip(0.9 * self.dq + 0.5 * a, -10, 10); self.q = self.q + 0.1 * self.dq; self.t += 1
        tip = np.array([np.cos(self.q[0]) + np.cos(self.q.sum()), np.sin(self.q[0]) + np.sin(self.q.sum())])
        return float(-np.linalg.norm(tip - self.target) - 0.01 * (a ** 2).sum())

    def terminated(self):
        return False

    def truncated(self):
        return self.t >= 50


n_envs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
max_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12000
env = pkg.HostParallelEnv([ReacherEnv(s) for s in range(n_envs)])
alg = pkg.SAC(start_steps=1000, buffer_capacity=100_000, batch_size=256, gradient_steps=n_envs, learning_rate=1e-3)
agent = pkg.SACAgent(pkg.SACLayer(env.observation_space(), env.action_space(), hidden_dims=(256, 256)), alg, seed=0)
agent, rb, stats, timer = pkg.sac_train_(agent, env, alg, max_steps)
r = rb.rewards
print(f"{agent.steps_taken} env steps on the host (collection loop {timer['collect_rollout']:.1f} s), {agent.gradient_updates} gradient steps on the device "
      f"(all device calls, predict / push / update: {timer['device']:.1f} s)")
print(f"mean reward per step: random start phase {r[:1000].mean():.3f} -> last 2000 transitions {r[-2000:].mean():.3f}; entropy coefficient {stats['entropy_coefficients'][-1]:.3f}")
