"""Batched env (include/pokerrl_hip.h section 4b) against the reference's recorded episodes (tests/golden/env_fuzz.npz, captured
from PokerRL's PokerEnv by tests/golden/make_golden.py): env i of the batch replays episode i % n_episodes of the fixture --
legal and ILLEGAL actions alike -- and after every batch step all envs are compared with the recorded public state, legal
actions, terminal / chance flags and pots. Shared by the GPU suite and the emulator suite."""
import numpy as np

from helpers import env_args, golden
from pokerrl_amd import _native
from pokerrl_amd.game import games as G

MAXL = 16
STATE_KEYS = ("round", "main_pot", "bet0", "bet1", "stack0", "stack1", "allin0", "allin1", "folded0", "folded1", "acted0", "acted1", "cur",
              "last_raiser", "capped_happened", "capped_raiser", "capped_cant_reopen", "n_actions_ep", "n_raises_round", "last_action_type",
              "last_action_amount", "last_action_seat")


def episodes_of(rows):
    """fixture rows -> list of episodes, each a list of (action, amount, n_legal, legal[:16], terminal, chance, pot, state22) per event
    (event 0 = the reset)"""
    eps = {}
    for r in rows:
        eps.setdefault(int(r[0]), []).append(r)
    return [eps[k] for k in sorted(eps)]


def check_envbatch_vs_reference(L, name, cls, stack, bets, n_envs):
    rows = golden("env_fuzz.npz")[name]
    eps = episodes_of(rows)
    game = cls.native_game(env_args(cls, stack, bets))
    is_limit, is_nl = cls.IS_FIXED_LIMIT_GAME, cls._GAME_TYPE == G.GAME_NOLIMIT
    b = _native.NativeEnvBatch(game, n_envs, _lib=L)
    which = np.arange(n_envs) % len(eps)
    max_len = max(len(e) for e in eps)
    n_checked = 0
    for k in range(max_len):
        # what the fixture says about event k of every episode (-1 rows = the episode is over)
        ev = np.full((len(eps), rows.shape[1]), -1, np.int64)
        for e, ep in enumerate(eps):
            if k < len(ep):
                ev[e] = ep[k]
        if k > 0:
            act = ev[which, 2].astype(np.int32)
            act[ev[which, 1] < 0] = -1  # finished episodes: skip
            info = b.step(act, ev[which, 3].astype(np.int32) if is_nl else None)
            live = ev[which, 1] >= 0
            assert np.array_equal(info[0][live], ev[which, 5 + MAXL][live].astype(np.int32)), (name, k, "is_terminal")
            assert np.all(info[0][~live] == -1)
            assert np.array_equal(info[1][live], ev[which, 6 + MAXL][live].astype(np.int32)), (name, k, "chance_acts")
            term = live & (ev[which, 5 + MAXL] == 1)
            assert np.array_equal(info[2][term], ev[which, 7 + MAXL][term].astype(np.int32)), (name, k, "pot before payout")
        st = b.state()
        masks, counts = b.legal_masks()
        running = (ev[which, 1] >= 0) & (ev[which, 5 + MAXL] != 1)  # envs whose fixture row carries a live state
        assert np.array_equal(st["done"][ev[which, 1] >= 0], (ev[which, 5 + MAXL] == 1)[ev[which, 1] >= 0].astype(np.int32))
        ref_state = ev[which, 8 + MAXL:]
        for j, key in enumerate(STATE_KEYS):
            if key == "n_raises_round" and not is_limit:
                continue  # exists only in fixed-limit games (PokerEnv.py:1196-1197)
            assert np.array_equal(st[key][running].astype(np.int64), ref_state[running, j]), (name, k, key)
        assert np.array_equal(counts[running].astype(np.int64), ev[which, 4][running]), (name, k, "n_legal")
        for e in range(len(eps)):  # legal action sets, once per episode (all replicas were just shown to hold the same state)
            if k < len(eps[e]) and ev[e, 5 + MAXL] != 1:
                i = int(np.argmax(which == e))
                mine = [a for a in range(128) if (int(masks[a >> 5, i]) >> (a & 31)) & 1]
                nl = int(ev[e, 4])
                assert mine[:MAXL] == [int(x) for x in ev[e, 5:5 + min(nl, MAXL)]], (name, k, e)
                n_checked += 1
        act_ids = b.active()
        assert np.array_equal(np.sort(act_ids), np.nonzero(st["done"] == 0)[0])
    assert n_checked > 200
    # masked reset: the finished envs restart, the others keep their state
    b.reset(np.ones(n_envs, np.uint8))
    assert len(b.active()) == n_envs
    return n_checked


def check_rollout_matches_host(L, cls, stack, bets, n_envs, n_steps, seed=7):
    game = cls.native_game(env_args(cls, stack, bets))
    b = _native.NativeEnvBatch(game, n_envs, _lib=L)
    steps, hands, pots, _ms = b.random_rollout(n_steps, seed)
    assert (steps, hands, pots) == _native.env_random_rollout_host(game, n_envs, n_steps, seed, _lib=L)
    assert steps == n_envs * n_steps and hands > 0
    b2 = _native.NativeEnvBatch(game, n_envs, _lib=L)  # one step per launch, state through HBM: the same hands
    assert b2.random_steps(n_steps, seed)[:3] == (steps, hands, pots)
    return steps, hands
