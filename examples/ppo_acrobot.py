#!/usr/bin/env python3
"""PPO on Acrobot-v1 as a DEVICE env (six observation dims: fused kernels at hidden [64,64], [128,128], [256,256] since round 3 — four first-layer k-steps, ppo_grad_pair_kernel from 65 536 samples per minibatch —, the generic kernels for any other hidden_dims; DRIL_FORCE_GENERIC=1 runs this example on them).

usage: python examples/ppo_acrobot.py [n_envs=1024] [iterations=40]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import __graft_entry__ as g

pkg = g.load_package()
n_envs = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 40
env = pkg.MonitorWrapperEnv(pkg.DeviceParallelEnv(pkg.AcrobotEnv(max_steps=500), n_envs, seed=0), stats_window=200)
alg = pkg.PPO(n_steps=128, batch_size=n_envs * 128 // 8, epochs=4, ent_coef=0.0, learning_rate=1e-3, gae_lambda=0.94, gamma=0.99)
agent = pkg.Agent(pkg.ActorCriticLayer(env.observation_space(), env.action_space(), hidden_dims=(64, 64)), alg, seed=0)
print("before:", pkg.evaluate_agent(agent, env, n_eval_episodes=20))
stats, timer = pkg.train_(agent, env, alg, iters * alg.n_steps * n_envs)
print(f"trained {iters} iterations ({iters * alg.n_steps * n_envs:,} env steps) in {timer['training_loop']:.2f} s; rollout fps {sum(stats['fps']) / len(stats['fps']):.3g}")
print("after: ", pkg.evaluate_agent(agent, env, n_eval_episodes=20))
