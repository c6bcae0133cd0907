#!/usr/bin/env python3
"""SAC on Pendulum-v1 on the device path (reference: src/algorithms/sac.jl; BASELINE.json configs[4] uses n_envs = 4096):

    env = MultiThreadedParallelEnv([PendulumEnv() for _ in 1:n]) -> DeviceParallelEnv(PendulumEnv(), n)
    alg = SAC();  layer = SACLayer(observation_space(env), action_space(env));  agent = Agent(layer, alg)
    agent, replay_buffer, training_stats, to = train!(agent, env, alg, max_steps)

usage: python examples/sac_pendulum.py [n_envs=64] [env_steps=200000]"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import __graft_entry__ as g

pkg = g.load_package()
n_envs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
max_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200_000
env = pkg.DeviceParallelEnv(pkg.PendulumEnv(max_steps=200), n_envs, seed=0)
alg = pkg.SAC(start_steps=5000, buffer_capacity=200_000, gradient_steps=max(1, n_envs // 8))
agent = pkg.SACAgent(pkg.SACLayer(env.observation_space(), env.action_space(), hidden_dims=(256, 256)), alg, seed=0)
agent, rb, stats, timer = pkg.sac_train_(agent, env, alg, max_steps)
r = rb.rewards
k = n_envs * 200
print(f"{agent.steps_taken} env steps, {agent.gradient_updates} gradient steps in {timer['training_loop']:.1f} s")
print(f"mean reward per step: first {k} transitions {r[:k].mean():.3f} -> last {k} transitions {r[-k:].mean():.3f}")
print(f"critic loss {np.mean(stats['critic_losses'][:50]):.3f} -> {np.mean(stats['critic_losses'][-50:]):.3f}; entropy coefficient {stats['entropy_coefficients'][-1]:.3f}")
