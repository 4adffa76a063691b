"""Batched device env: see envbatch_cases.py. CPU: the kernel sources on the SIMT emulator; GPU: 65536 envs per game."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))

import envbatch_cases as ec  # noqa: E402
from pokerrl_amd import _native  # noqa: E402
from test_host_golden import ENV_FUZZ  # noqa: E402


@pytest.fixture(scope="module")
def EMU():
    import build_emu
    return _native.bind(build_emu.build())


@pytest.mark.parametrize("name", ["StandardLeduc", "DiscretizedNLLeduc_B5_short", "LimitHoldem_short", "NoLimitHoldem_short", "Flop5Holdem_short"])
def test_emu_envbatch_replays_reference_episodes(EMU, name):
    cls, stack, bets = ENV_FUZZ[name]
    ec.check_envbatch_vs_reference(EMU, name, cls, stack, bets, n_envs=300)  # 150 episodes, each twice; 5 waves, the last one ragged


def test_emu_envbatch_rollout_equals_host_rollout(EMU):
    cls, stack, bets = ENV_FUZZ["DiscretizedNLHoldem_B5"]
    ec.check_rollout_matches_host(EMU, cls, stack, bets, n_envs=130, n_steps=40)


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(ENV_FUZZ))
def test_gpu_envbatch_65536_envs_replay_reference_episodes(name):
    _native.require_device()
    cls, stack, bets = ENV_FUZZ[name]
    ec.check_envbatch_vs_reference(_native.lib(), name, cls, stack, bets, n_envs=65536 + 37)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["StandardLeduc", "DiscretizedNLHoldem_B5", "NoLimitHoldem_short", "LimitHoldem"])
def test_gpu_envbatch_rollout_equals_host_rollout(name):
    _native.require_device()
    cls, stack, bets = ENV_FUZZ[name]
    ec.check_rollout_matches_host(_native.lib(), cls, stack, bets, n_envs=20000, n_steps=64)


# ---- the whole PokerEnv.step: cards, payouts, rewards, observation vectors (prl_envbatch_create_with_cards) -------------------------------
FULL_GAMES = ["StandardLeduc", "BigLeduc_short", "DiscretizedNLLeduc_B5_short", "LimitHoldem", "DiscretizedNLHoldem_B5", "DiscretizedNLHoldem_OT11_short",
              "Flop5Holdem"]


@pytest.mark.parametrize("name", FULL_GAMES)
def test_emu_full_env_replays_reference_episodes_with_cards(EMU, name):
    """(obs, reward, done) of n envs from one call = the reference's PokerEnv.step on its own recorded episodes and cards (env_obs.npz)"""
    ec.check_full_env_vs_reference(EMU, name, n_envs=100)


def test_emu_full_env_rollout_equals_host_rollout(EMU):
    from pokerrl_amd.game import games as G
    ec.check_full_rollout_matches_host(EMU, G.LimitHoldem, 48, [0.0], n_envs=130, n_steps=40)
    ec.check_full_rollout_matches_host(EMU, G.StandardLeduc, 13, [0.0], n_envs=70, n_steps=40)


@pytest.mark.gpu
@pytest.mark.parametrize("name", FULL_GAMES)
def test_gpu_full_env_65536_envs_replay_reference_episodes_with_cards(name):
    _native.require_device()
    ec.check_full_env_vs_reference(_native.lib(), name, n_envs=65536 + 37)


@pytest.mark.gpu
def test_gpu_full_env_rollout_equals_host_rollout_and_steps_in_hbm():
    """whole hands in registers = the host engine (steps, hands, showdowns, payout checksum); the same play one whole step per launch (state,
    observations, rewards in HBM) finishes the same hands with the same pots as the betting-only rollout"""
    from pokerrl_amd.game import bet_sets
    from pokerrl_amd.game import games as G
    _native.require_device()
    L = _native.lib()
    ec.check_full_rollout_matches_host(L, G.LimitHoldem, 48, [0.0], n_envs=20000, n_steps=64)
    ec.check_full_rollout_matches_host(L, G.DiscretizedNLHoldem, 20000, bet_sets.B_5, n_envs=20000, n_steps=64)
    from helpers import env_args
    game, rules = G.DiscretizedNLHoldem.native_game(env_args(G.DiscretizedNLHoldem, 20000, bet_sets.B_5)), G.DiscretizedNLHoldem.native_rules()
    b = _native.NativeEnvBatch.with_cards(game, rules, 30000, deck_seed=3)
    s_full = b.random_steps_full(48, 9)[:3]
    plain = _native.NativeEnvBatch(game, 30000)
    assert s_full == plain.random_steps(48, 9)[:3]


def test_emu_random_steps_whole_chunk_kernel_equals_the_general_one(EMU):
    cls, stack, bets = ENV_FUZZ["DiscretizedNLHoldem_B5"]
    ec.check_random_steps_outputs(EMU, cls, stack, bets, n_envs=1536, n_launches=9, grid_cap=2)  # two workgroups, three chunks each


@pytest.mark.gpu
@pytest.mark.parametrize("n_envs", [1 << 18, 65536 + 256, 70000])
def test_gpu_random_steps_whole_chunk_kernel_equals_the_general_one(n_envs):
    _native.require_device()
    cls, stack, bets = ENV_FUZZ["DiscretizedNLHoldem_B5"]
    ec.check_random_steps_outputs(_native.lib(), cls, stack, bets, n_envs=n_envs, n_launches=12)
