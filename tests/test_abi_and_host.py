"""CPU: the C-ABI library loads and exports every declared symbol (no compute without a GPU); host-side logic."""
import ctypes as C
import re
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parents[1]


def test_library_exports_every_declared_symbol(pkg):
    so = ROOT / "dril.jl_amd" / "csrc" / "libdril_hip.so"
    assert so.exists(), "build with __graft_entry__.build()"
    header = "".join(p.read_text() for p in sorted((ROOT / "include").glob("*.h")))      # dril_hip.h + dril_sac.h
    declared = set(re.findall(r"\b(dril_[a-z0-9_]+)\s*\(", header))
    assert declared == set(pkg._capi.EXPORTED_SYMBOLS)          # the ctypes table covers the whole header
    lib = pkg._capi.load_library()                               # types every entry point; AttributeError if one is missing
    out = subprocess.run(["nm", "-D", "--defined-only", str(so)], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (dril_[a-z0-9_]+)", out))
    assert declared <= exported
    assert lib.dril_version().startswith(b"dril_hip")
    assert lib.dril_kernel_name(pkg._capi.K_PPO_GRAD) == b"ppo_grad_kernel"


def test_config_struct_layout_matches_c(pkg, tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "dril_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu\\n", sizeof(dril_config),'
                   ' offsetof(dril_config, batch_size), offsetof(dril_config, seed), offsetof(dril_config, profile_events), sizeof(dril_ppo_stats));return 0;}')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", str(ROOT / "include"), str(src), "-o", str(exe)], check=True)
    sz, off_b, off_s, off_p, sz_st = map(int, subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split())
    K = pkg._capi.DrilConfig
    assert (C.sizeof(K), K.batch_size.offset, K.seed.offset, K.profile_events.offset) == (sz, off_b, off_s, off_p)
    assert C.sizeof(pkg._capi.DrilPPOStats) == sz_st


def test_config_default_matches_reference_defaults(pkg):
    """dril_config_default (host code, no GPU) == PPO() defaults, src/algorithms/ppo.jl:26-39"""
    lib = pkg._capi.load_library()
    c = pkg._capi.DrilConfig()
    assert lib.dril_config_default(C.byref(c), pkg._capi.ENV_CARTPOLE) == 0
    py = pkg._capi.default_config(pkg._capi.ENV_CARTPOLE)
    for name, _ in pkg._capi.DrilConfig._fields_:
        if name == "reserved":
            continue
        assert getattr(c, name) == pytest.approx(getattr(py, name)), name
    assert (c.gamma, c.gae_lambda, c.clip_range, c.vf_coef, c.max_grad_norm) == pytest.approx((0.99, 0.95, 0.2, 0.5, 0.5))
    assert (c.n_steps, c.batch_size, c.epochs, c.adam_eps) == (2048, 64, 10, pytest.approx(1e-5))
    assert lib.dril_config_default(C.byref(c), 99) == pkg._capi.ERR_INVALID_ARG


def test_null_and_bad_arguments_fail_loudly(pkg):
    """error convention: status codes + dril_last_error, never a crash, never a silent fallback"""
    lib = pkg._capi.load_library()
    h = C.c_void_p()
    assert lib.dril_create(None, C.byref(h)) == pkg._capi.ERR_INVALID_ARG
    cfg = pkg._capi.default_config(0); cfg.abi_version = 77
    assert lib.dril_create(C.byref(cfg), C.byref(h)) == pkg._capi.ERR_INVALID_ARG
    assert b"abi_version" in lib.dril_last_error(None)
    cfg = pkg._capi.default_config(0); cfg.hidden1 = cfg.hidden2 = 2000           # any width up to 1024 is served (fused or generic kernels)
    assert lib.dril_create(C.byref(cfg), C.byref(h)) == pkg._capi.ERR_INVALID_ARG and b"1..1024" in lib.dril_last_error(None)
    assert lib.dril_synchronize(None) == pkg._capi.ERR_NOT_INITIALISED
    assert lib.dril_param_count(None) == -1
    assert lib.dril_gae(0, 4, 0.9, 0.9, None, None, None, None, None, None, None) == pkg._capi.ERR_INVALID_ARG


def test_missing_library_raises(pkg, tmp_path):
    with pytest.raises(pkg._capi.DrilLibraryMissing):
        pkg._capi.load_library(tmp_path / "libdril_hip.so")


def test_flatten_roundtrip_and_layout(pkg):
    env = pkg.PendulumEnv()
    layer = pkg.ActorCriticLayer(env.observation_space(), env.action_space(), log_std_init=-0.5)
    ps = layer.initialparameters(np.random.default_rng(4))
    flat = pkg.flatten_params(ps)
    back = pkg.unflatten_params(flat, ps)
    for head in ("actor_head", "critic_head"):
        for l in ("layer_1", "layer_2", "layer_3"):
            assert np.array_equal(back[head][l]["weight"], ps[head][l]["weight"])
            assert np.array_equal(back[head][l]["bias"], ps[head][l]["bias"])
    assert np.array_equal(back["log_std"], ps["log_std"]) and flat[-1] == np.float32(-0.5)
    W1 = ps["actor_head"]["layer_1"]["weight"]                    # (64 x 3), column-major in the flat vector
    assert flat[1] == W1[1, 0] and flat[64] == W1[0, 1]
    # orthogonal init gains (layer_constructors.jl:16-20): hidden sqrt(2), actor 0.01, value 1.0; zero bias
    W2 = ps["actor_head"]["layer_2"]["weight"].astype(np.float64)
    np.testing.assert_allclose(W2 @ W2.T, 2.0 * np.eye(64), atol=1e-5)
    W3a = ps["actor_head"]["layer_3"]["weight"].astype(np.float64); W3c = ps["critic_head"]["layer_3"]["weight"].astype(np.float64)
    np.testing.assert_allclose(W3a @ W3a.T, 1e-4 * np.eye(1), atol=1e-9)
    np.testing.assert_allclose(W3c @ W3c.T, np.eye(1), atol=1e-5)
    assert not ps["critic_head"]["layer_2"]["bias"].any()
    assert "log_std" not in pkg.ActorCriticLayer(pkg.CartPoleEnv().observation_space(), pkg.CartPoleEnv().action_space()).initialparameters(np.random.default_rng(0))


def test_reference_order_map(pkg, oracle_mod):
    """RolloutBuffer.to_reference_order reproduces the completion order the oracle records while collecting
    trajectories the reference's way (rollout_buffer.jl:70-80, trajectory.jl:46-75)."""
    cfg = pkg._capi.default_config(0); cfg.n_envs, cfg.n_steps, cfg.episode_len, cfg.batch_size, cfg.epochs = 6, 120, 25, 10, 1
    o = oracle_mod.Oracle(cfg)
    flat = (np.random.default_rng(2).standard_normal(o.P) * 0.5).astype(np.float32)
    flat[4608:4610] = (2.0, -2.0)                                 # actor output bias: push mostly one way so poles fall
    o.set_params(flat)
    o.env_reset(3); o.collect_rollout()
    buf = pkg.RolloutBuffer(cfg.n_steps, cfg.n_envs, cfg.gae_lambda, cfg.gamma, flags=o.buffer(pkg._capi.BUF_FLAGS))
    assert np.array_equal(buf.to_reference_order(), o.ref_order())
    assert (o.buffer(pkg._capi.BUF_FLAGS) & 1).any()              # real CartPole terminations occurred in this rollout


def test_perm_bijection(oracle_mod):
    L = oracle_mod.lib()
    for n in (1, 2, 31, 64, 1000, 4097):
        key = L.orc_perm_key(42, 3, 1)
        idx = np.array([L.orc_perm_index(p, n, key) for p in range(n)])
        assert np.array_equal(np.sort(idx), np.arange(n))
    a = np.array([L.orc_perm_index(p, 4097, L.orc_perm_key(42, 0, 0)) for p in range(4097)])
    b = np.array([L.orc_perm_index(p, 4097, L.orc_perm_key(42, 0, 1)) for p in range(4097)])
    assert (a != b).mean() > 0.99 and abs(np.corrcoef(a, np.arange(4097))[0, 1]) < 0.1


def test_checkpoint_roundtrip_schema(pkg, tmp_path):
    """test/test_ppo_integration.jl:42-83 (parameters identical after save -> load into a freshly initialised agent) + the SAC aux"""
    env = pkg.CartPoleEnv()
    alg = pkg.PPO(n_steps=16, batch_size=16, epochs=2)
    layer = pkg.ActorCriticLayer(env.observation_space(), env.action_space(), hidden_dims=(32, 32))
    a, b = pkg.Agent(layer, alg, seed=77), pkg.Agent(layer, alg, seed=88)
    fa, fb = pkg.flatten_params(a.train_state.parameters), pkg.flatten_params(b.train_state.parameters)
    assert not np.array_equal(fa, fb)
    path = pkg.save_policy_params_and_state(a, tmp_path / "ppo_agent")
    keys = set(np.load(path).files)
    assert {"layer", "states", "parameters/actor_head/layer_1/weight", "parameters/critic_head/layer_3/bias"} <= keys       # agent_methods.jl:129-136
    pkg.load_policy_params_and_state_(b, alg, path)
    np.testing.assert_array_equal(pkg.flatten_params(b.train_state.parameters), fa)
    penv = pkg.PendulumEnv()
    sl = pkg.SACLayer(penv.observation_space(), penv.action_space(), hidden_dims=(32, 32))
    s1, s2 = pkg.SACAgent(sl, pkg.SAC(), seed=1), pkg.SACAgent(sl, pkg.SAC(), seed=2)
    s1.log_ent_coef = -0.25
    pkg.load_policy_params_and_state_(s2, s1.alg, pkg.save_policy_params_and_state(s1, tmp_path / "sac_agent"))
    np.testing.assert_array_equal(pkg.sac_flatten_params(s2.parameters), pkg.sac_flatten_params(s1.parameters))
    np.testing.assert_array_equal(s2.q_target_parameters, s1.q_target_parameters)
    assert s2.log_ent_coef == pytest.approx(-0.25)


def test_sac_struct_layout_and_defaults(pkg, tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "dril_sac.h"\nint main(){printf("%zu %zu %zu %zu %zu\\n", sizeof(dril_sac_config),'
                   ' offsetof(dril_sac_config, buffer_capacity), offsetof(dril_sac_config, seed), offsetof(dril_sac_config, profile_events), sizeof(dril_sac_stats));return 0;}')
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", str(ROOT / "include"), str(src), "-o", str(exe)], check=True)
    sz, off_b, off_s, off_p, sz_st = map(int, subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split())
    K = pkg._capi.DrilSacConfig
    assert (C.sizeof(K), K.buffer_capacity.offset, K.seed.offset, K.profile_events.offset) == (sz, off_b, off_s, off_p)
    assert C.sizeof(pkg._capi.DrilSacStats) == sz_st
    lib = pkg._capi.load_library()
    c = K()
    assert lib.dril_sac_config_default(C.byref(c), pkg._capi.ENV_PENDULUM) == 0                 # SAC() sac.jl:25-36, SACLayer :72-85
    assert (c.buffer_capacity, c.start_steps, c.batch_size, c.train_freq, c.gradient_steps, c.target_update_interval) == (1_000_000, 100, 256, 1, 1, 1)
    assert (c.hidden1, c.hidden2, c.activation, c.auto_ent_coef, c.auto_target_entropy) == (512, 512, 1, 1, 1)
    assert (c.tau, c.gamma, c.learning_rate, c.adam_eps, c.ent_coef_init) == pytest.approx((0.005, 0.99, 3e-4, 1e-8, 1.0))
    py = pkg.make_sac_config(pkg.PendulumEnv(), 1, pkg.SAC(), pkg.SACLayer(pkg.PendulumEnv().observation_space(), pkg.PendulumEnv().action_space()))
    for name, _ in K._fields_:
        if name not in ("reserved",):
            assert getattr(c, name) == pytest.approx(getattr(py, name)), name
    assert lib.dril_sac_config_default(C.byref(c), pkg._capi.ENV_CARTPOLE) == pkg._capi.ERR_INVALID_ARG   # Box action space required (sac.jl:74)
    h = C.c_void_p()
    assert lib.dril_sac_create(None, C.byref(h)) == pkg._capi.ERR_INVALID_ARG
    assert b"null" in lib.dril_sac_last_error(None)


def test_keyed_bijection_32bit_form(tmp_path):
    """the DataLoader order of the device (perm_index / perm_position, dril_device.h) has a 32-bit fast path for buffers of up to 2^32 samples: it must be the
    SAME permutation as the 64-bit form (and its inverse the same inverse), for every width and for ragged sizes with cycle walking"""
    import subprocess
    root = Path(__file__).resolve().parents[1]
    src = (root / "dril.jl_amd" / "csrc" / "dril_device.h").read_text()
    a = src.index("// keyed bijection on [0, n)"); b = src.index("// ---------------------------------------------------------------------------------------------\n// math")
    (tmp_path / "perm32_extract.inc").write_text(src[a:b])
    exe = tmp_path / "perm32_check"
    subprocess.run(["g++", "-O2", "-I", str(tmp_path), "-o", str(exe), str(root / "tests" / "perm32_check.cpp")], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0 and "mismatches 0" in r.stdout, r.stdout + r.stderr
