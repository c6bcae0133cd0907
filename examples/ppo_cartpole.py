#!/usr/bin/env python3
"""The reference README quick-start (README.md:50-73 of DRiL.jl) on the device path:

    env   = MultiThreadedParallelEnv([CartPoleEnv() for _ in 1:4])      ->  DeviceParallelEnv(CartPoleEnv(), n_envs)
    layer = ActorCriticLayer(observation_space(env), action_space(env))
    alg   = PPO();  agent = Agent(layer, alg)
    train!(agent, env, alg, max_steps)
    evaluate_agent(agent, env; n_eval_episodes = 10)

usage: python examples/ppo_cartpole.py [n_envs=256] [iterations=30]"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import __graft_entry__ as g

pkg = g.load_package()
n_envs = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
env = pkg.MonitorWrapperEnv(pkg.DeviceParallelEnv(pkg.CartPoleEnv(max_steps=500), n_envs, seed=0), stats_window=100)
alg = pkg.PPO(n_steps=128, batch_size=n_envs * 128 // 4, epochs=4, ent_coef=0.01, learning_rate=1e-3)
agent = pkg.Agent(pkg.ActorCriticLayer(env.observation_space(), env.action_space()), alg, seed=0)
print("before:", pkg.evaluate_agent(agent, env, n_eval_episodes=20))
stats, timer = pkg.train_(agent, env, alg, iters * alg.n_steps * n_envs)
print(f"trained {iters} iterations in {timer['training_loop']:.2f} s; last loss {stats['losses'][-1]:.4f}, mean rollout fps {sum(stats['fps']) / len(stats['fps']):.3g}")
print("after: ", pkg.evaluate_agent(agent, env, n_eval_episodes=20))
