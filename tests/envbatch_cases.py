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


# ---- the whole PokerEnv.step (cards, payouts, rewards, observations): tests/golden/env_obs.npz ------------------------------------------
def check_full_env_vs_reference(L, name, n_envs):
    """env i replays episode i % 40 of the reference's recorded full episodes with the reference's own cards: observation vectors after the
    reset and after every step, the final rewards and the done flags -- bit for bit"""
    g = golden("env_obs.npz")
    obs_ref, meta = g[name + "_obs"], g[name + "_meta"]
    cls, stack, bset = _env_obs_cfg()[name]
    args = env_args(cls, stack, bset)
    game, rules = cls.native_game(args), cls.native_rules()
    nh, nb = rules.n_hole_cards, rules.n_board_cards
    eps, cur = [], None
    for row, o in zip(meta, obs_ref):  # meta rows: ep, action (-1 = reset), done, r0, r1 [, cards at the last row of the episode]
        if int(row[1]) == -1:
            cur = dict(rows=[], obs=[])
            eps.append(cur)
        cur["rows"].append(row)
        cur["obs"].append(o)
    reward_scalar = (float(stack + stack) / 2.0 / 5.0) if getattr(args, "scale_rewards", False) else 1.0
    b = _native.NativeEnvBatch.with_cards(game, rules, n_envs, deck_seed=1, reward_scalar=reward_scalar, _lib=L)
    assert b.obs_dim == obs_ref.shape[1]
    which = np.arange(n_envs) % len(eps)
    cards = np.zeros((len(eps), 2 * nh + nb), np.int8)
    for e, ep in enumerate(eps):
        last = ep["rows"][-1]
        board = [int(c) for c in last[5:5 + nb]]
        hands = [int(c) for c in last[5 + nb:5 + nb + 2 * nh]]
        used = set(c for c in board + hands if c >= 0)
        spare = [c for c in range(rules.n_cards) if c not in used]
        board = [c if c >= 0 else spare.pop() for c in board]  # streets the episode never reached: any unused card (never looked at)
        cards[e] = hands + board
    b.reset_full()
    b.set_cards(cards[which])
    o = b.observe()
    first = np.stack([ep["obs"][0] for ep in eps])
    assert np.array_equal(o, first[which]), (name, "observation after the reset")
    max_len = max(len(ep["rows"]) for ep in eps)
    for k in range(1, max_len):
        act = np.array([int(ep["rows"][k][1]) if k < len(ep["rows"]) else -1 for ep in eps], np.int32)
        live = (act >= 0)[which]
        obs, rew, done, info = b.step_full(act[which])
        want_obs = np.stack([ep["obs"][k] if k < len(ep["rows"]) else np.zeros(b.obs_dim, np.float32) for ep in eps])
        assert np.array_equal(obs[live], want_obs[which][live]), (name, k, "observation")
        want_done = np.array([int(ep["rows"][k][2]) if k < len(ep["rows"]) else 1 for ep in eps])
        assert np.array_equal(done[live].astype(int), want_done[which][live]), (name, k, "done")
        want_rew = np.array([[ep["rows"][k][3], ep["rows"][k][4]] if k < len(ep["rows"]) else [0.0, 0.0] for ep in eps], np.float64)
        assert np.array_equal(rew[live], want_rew[which][live]), (name, k, "rewards")
        assert np.all(done[~live] == 1) and np.all(obs[~live] == 0)
    return len(eps), max_len


def _env_obs_cfg():
    """fixture name -> (game class, stack, bet set): ENV_FUZZ of tests/golden/make_golden.py"""
    from pokerrl_amd.game import bet_sets
    return {"StandardLeduc": (G.StandardLeduc, 13, [0.0]), "BigLeduc_short": (G.BigLeduc, 9, [0.0]),
            "DiscretizedNLLeduc_B5_short": (G.DiscretizedNLLeduc, 900, bet_sets.B_5), "LimitHoldem": (G.LimitHoldem, 48, [0.0]),
            "DiscretizedNLHoldem_B5": (G.DiscretizedNLHoldem, 20000, bet_sets.B_5),
            "DiscretizedNLHoldem_OT11_short": (G.DiscretizedNLHoldem, 2300, bet_sets.OFF_TREE_11), "Flop5Holdem": (G.Flop5Holdem, 20000, [0.0])}


def check_full_rollout_matches_host(L, cls, stack, bets, n_envs, n_steps, seed=5):
    """whole hands in registers on the device = the same hands on the host: steps, hands, showdowns and the payout checksum"""
    args = env_args(cls, stack, bets)
    game, rules = cls.native_game(args), cls.native_rules()
    b = _native.NativeEnvBatch.with_cards(game, rules, n_envs, deck_seed=77, reward_scalar=1.0, _lib=L)
    dev = b.random_rollout_full(n_steps, seed)[:4]
    host = _native.env_random_rollout_full_host(game, rules, n_envs, n_steps, seed, deck_seed=77, reward_scalar=1.0, _lib=L)
    assert dev == host and dev[0] == n_envs * n_steps and dev[1] > 0 and dev[2] > 0, (dev, host)
    return dev


def check_random_steps_outputs(L, cls, stack, bets, n_envs, n_launches, seed=3, grid_cap=None):
    """prl_envbatch_random_steps_full: (a) its specialised kernel for whole chunks of 256 envs (straight-line 16-byte observation stores, next
    chunk's loads ahead of them) = the general kernel, bit for bit -- observations, rewards, done flags, state; (b) the observation vectors it
    leaves = prl_envbatch_observe of the state it leaves (zeros where the hand just ended)."""
    import os
    args = env_args(cls, stack, bets)
    game, rules = cls.native_game(args), cls.native_rules()
    outs = []
    for general in (True, False):
        if general:
            os.environ["PRL_EB_NO_WHOLE"] = "1"
        else:
            os.environ.pop("PRL_EB_NO_WHOLE", None)
        if grid_cap:
            os.environ["PRL_EB_GRID_CAP"] = str(grid_cap)  # fewer workgroups than chunks: every workgroup walks several
        try:
            b = _native.NativeEnvBatch.with_cards(game, rules, n_envs, deck_seed=21, _lib=L)
            stats = b.random_steps_full(n_launches, seed)[:3]
            obs, rew, done = b.last_outputs()
            cols = {k: v.copy() for k, v in b.state().items()}
            now = b.observe()
        finally:
            os.environ.pop("PRL_EB_NO_WHOLE", None)
            os.environ.pop("PRL_EB_GRID_CAP", None)
        live = done == 0
        assert live.any() and (~live).any()
        assert np.array_equal(obs[live], now[live]) and not obs[~live].any() and not rew[live].any()
        outs.append((stats, obs, rew, done, cols))
    a, b2 = outs
    assert a[0] == b2[0] and np.array_equal(a[1], b2[1]) and np.array_equal(a[2], b2[2]) and np.array_equal(a[3], b2[3])
    for k in a[4]:
        assert np.array_equal(a[4][k], b2[4][k]), k
