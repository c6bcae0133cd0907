"""diagnostic (GPU): the README quick-start example over several seeds — final evaluation reward after 30 iterations (is a change of arithmetic visible in learning?)"""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
import __graft_entry__ as g
pkg = g.load_package()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n_envs = int(sys.argv[2]) if len(sys.argv) > 2 else 256      # 4096: minibatches of 131 072 samples = ppo_grad_pair_kernel (DRIL_GRAD_VARIANT=0: the exact-f32 kernel)
out = []
for seed in range(10):
    env = pkg.MonitorWrapperEnv(pkg.DeviceParallelEnv(pkg.CartPoleEnv(max_steps=500), n_envs, seed=seed), stats_window=100)
    alg = pkg.PPO(n_steps=128, batch_size=n_envs * 128 // 4, epochs=4, ent_coef=0.01, learning_rate=1e-3)
    agent = pkg.Agent(pkg.ActorCriticLayer(env.observation_space(), env.action_space()), alg, seed=seed)
    pkg.train_(agent, env, alg, iters * alg.n_steps * n_envs)
    out.append(round(pkg.evaluate_agent(agent, env, n_eval_episodes=20)["mean_reward"], 1))
print("n_envs", n_envs, "final rewards over seeds 0..9:", out, "mean %.1f" % (sum(out) / len(out)))
