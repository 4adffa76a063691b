"""
Tabular agents for the evaluators: the policy a CFR solver left in its PublicTree, served (a) to the host evaluators as an EvalAgent and (b) to the
batched GPU evaluators (BatchedLBR / BatchedHead2Head, agent kind "table") as a table in HBM -- the same float32 numbers on both sides, so the batched
engines stay bit-identical to LocalLBRWorker / LocalHead2HeadMaster playing the host agent. This is what the reference's evaluators get from
EvalAgentBase.get_a_probs_for_each_hand / get_action (PokerRL/rl/base_cls/EvalAgentBase.py:35-62, used by eval/lbr/LocalLBRWorker.py:120-160,241-281 and
eval/head_to_head/LocalHead2HeadMaster.py:100-118) when the agent is tabular.

    cfr = CFRPlus(name, chief, game_cls=StandardLeduc, agent_bet_set=None); cfr.reset(); cfr.iterations(500)
    table = PolicyTable.from_cfr(cfr)                                    # average strategy, float32 [row][action][hand]
    BatchedLBR(t_prof, agent_kind="table", table=table).run(agent_seat_id=0, n_hands=1 << 20)
    LocalLBRWorker(t_prof, chief_handle=None, eval_agent_cls=make_table_agent_cls(EvalAgentBase, table))   # the same hands on the host

A row belongs to a NODE of the agent's public tree, not to a public state (two betting histories can meet in one state), so rows are addressed by a
history key: key(root) = state_key(root state, seed), key(next) = state_key(next state, seed = key(now)) after every env step, with the new board
cards on the table when the step ends a round (csrc/prl_lbr_batch.hip: lbrb_hist_step; two 32-bit chains make the 64-bit key). The chain runs over
STATES, not action ids: LBR raising by a pot fraction of its own bet set reaches the agent's node whenever the agent's tree has a raise to the same
amount. A history the table does not hold (LBR left the tree) plays uniformly over the legal actions, on both sides.

Hold'em-sized tables never pass through the host: PolicyTable.from_solver(cfr or tree) builds the table on the device from the solver's own columns
(prl_policy_table_from_solver: the fused engine's sorted board columns expanded to hand order a chunk of boards at a time). The table of a WHOLE-GAME
solve (suit classes, prl_solver_create_weighted) is SUIT-CANONICAL: it holds the class representatives' rows only, and both the kernels and the host
twin below relabel the dealt board to its representative (the lexicographically smallest relabelling, first permutation that attains it) before they
hash it, and read the row at the hand relabelled by the same permutation:

    cfr = CFRPlus(name, chief, game_cls=Flop5Holdem, agent_bet_set=None); cfr.reset(); cfr.iterations(100)   # all 2 598 960 boards as 134 459 classes
    table = PolicyTable.from_solver(cfr)                                 # 806 758 rows, 12.8 GB in HBM, no host copy
    BatchedLBR(t_prof, agent_kind="table", table=table).run(agent_seat_id=0, n_hands=1 << 20)
"""
import ctypes
import weakref
from itertools import permutations

import numpy as np

from pokerrl_amd import _native
from pokerrl_amd.game.PublicTree import KIND_CHANCE, KIND_DECISION
from pokerrl_amd.rl.hash_agent import make_agent_cls, state_key

KEY_SEED = 0x7AB1E5
KEY_SEED_HI_XOR = 0x5BD1E995  # LBRB_KEY_SEED_HI


def hist_root(key_seed=KEY_SEED):
    return int(key_seed) & 0xFFFFFFFF, (int(key_seed) ^ KEY_SEED_HI_XOR) & 0xFFFFFFFF


def hist_step(hk, env, canon=False):
    """the history key after the env reached its present state (lbrb_hist_step); canon: with the suit-canonical form of the board on the table"""
    if canon:
        env = _CanonBoardView(env)
    return int(state_key(env, hk[0])), int(state_key(env, hk[1]))


def suit_canon(board_1d, n_suits=4):
    """(canonical board: cards ascending, the permutation number) -- the lexicographically smallest of the board's relabellings card -> rank * n_suits +
    p[suit], the FIRST permutation p in lexicographic order that attains it (csrc/prl_policy.h: prl_suit_canon; written independently of it: the tests
    compare the two over every board)"""
    b = np.asarray(board_1d, np.int64)
    rank, suit = b // n_suits, b % n_suits
    best, best_k = None, 0
    for k, p in enumerate(permutations(range(n_suits))):
        m = tuple(sorted(int(x) for x in rank * n_suits + np.asarray(p, np.int64)[suit]))
        if best is None or m < best:
            best, best_k = m, k
    return np.asarray(best, np.int8), best_k


_HAND_PERMS = {}


def hand_perm(k, n_cards=52, n_suits=4):
    """int32 [R]: the range index of every 2-card hand with both cards relabelled by suit permutation number k"""
    key = (k, n_cards, n_suits)
    if key not in _HAND_PERMS:
        p = np.asarray(list(permutations(range(n_suits)))[k], np.int64)
        c = np.arange(n_cards)
        new = (c // n_suits) * n_suits + p[c % n_suits]
        idx = np.full((n_cards, n_cards), -1, np.int64)
        i = 0
        for a in range(n_cards):
            for b in range(a + 1, n_cards):
                idx[a, b] = idx[b, a] = i
                i += 1
        a, b = np.triu_indices(n_cards, 1)  # hands in range-index order
        _HAND_PERMS[key] = idx[new[a], new[b]].astype(np.int32)
    return _HAND_PERMS[key]


class _CanonBoardView:
    """an env seen with the suit-canonical form of its board (what state_key hashes for a suit-canonical table)"""

    def __init__(self, env):
        self._env = env
        b2 = np.asarray(env.board)
        dealt = b2[:, 0] >= 0
        out = np.array(b2, copy=True)
        if dealt.any():
            ns = int(getattr(env, "N_SUITS", 4))
            b1 = b2[dealt, 0].astype(np.int64) * ns + b2[dealt, 1].astype(np.int64)
            cb, self.k = suit_canon(b1, ns)
            out[:len(cb), 0], out[:len(cb), 1] = cb // ns, cb % ns
        else:
            self.k = 0
        self.board = out

    def __getattr__(self, name):
        return getattr(self._env, name)


def _key64(hk):
    return ((hk[1] << 32) | hk[0]) or 1


def _first_slot(hk, mask):
    return (hk[0] ^ ((hk[1] * 0x9E3779B1) & 0xFFFFFFFF)) & mask


class PolicyTable:
    """probs: float32 [n_rows][n_actions][range_size]; hist_keys: the (lo, hi) history key of every row. A table built by from_solver lives on the
    device only (`probs` is None; row_probs fetches single rows for the host twin)."""

    def __init__(self, hist_keys, probs, key_seed=KEY_SEED, node_keys=None):
        self.probs = np.ascontiguousarray(probs, dtype=np.float32)
        self.n_rows, self.n_actions, self.range_size = self.probs.shape
        assert len(hist_keys) == self.n_rows
        self.key_seed = int(key_seed)
        self.suit_canon = False
        self.node_keys = node_keys or {}  # tree node index -> history key of the tree the table was built from (diagnostics; look-ups go by path: key_of_node)
        cap = 2
        while cap < 2 * self.n_rows:
            cap *= 2
        self.capacity = cap
        self.keys = np.zeros(cap, np.uint64)
        self.rows = np.full(cap, -1, np.int32)
        self._row_of = {}
        for r, hk in enumerate(hist_keys):
            k = _key64(hk)
            assert k not in self._row_of, "two tree nodes share a 64-bit history key: build the table under another key_seed"
            self._row_of[k] = r
            i = _first_slot(hk, cap - 1)
            while self.keys[i] != 0:
                i = (i + 1) & (cap - 1)
            self.keys[i], self.rows[i] = k, r
        self._dev = None
        self._node_key_cache = weakref.WeakKeyDictionary()
        self._row_cache = {}

    # ---- building it from a public tree ----------------------------------------------------------------------------------------------------------
    @classmethod
    def from_tree(cls, tree, columns=None, key_seed=KEY_SEED):
        """one row per expanded decision node of `tree` (pokerrl_amd.game.PublicTree); columns: [n_cols, R] strategy columns in the tree's order
        (column first_col[node] + j = P(j-th allowed action | hand)), default: the strategy the tree holds now. Host path: the columns pass through
        NumPy (Leduc-sized trees); from_solver is the device path."""
        cols = np.asarray(tree.solver.get("strategy") if columns is None else columns)
        kind, parent, first_col, n_children = tree._kind, tree._parent, tree._first_col, tree._n_children
        n_actions = int(tree._env_bldr.N_ACTIONS)
        env = tree._get_replay_env()
        key_of, keys, rows, holder, twin_of = {}, [], [], {}, {}

        def dec_parent(i):
            p = int(parent[i])
            while p >= 0 and kind[p] != KIND_DECISION:
                p = int(parent[p])
            return p

        for i in range(len(kind)):  # parents come before their children in the tree's node order
            p = int(parent[i])
            if kind[i] == KIND_CHANCE:
                key_of[i] = key_of[p]  # the deal is part of the step that ended the round: the outcome's state carries the new board
                continue
            if kind[i] != KIND_DECISION:
                continue
            env.load_state_dict(tree.node(i).env_state, blank_private_info=True)
            key_of[i] = hist_step(hist_root(key_seed) if p < 0 else key_of[p], env)
            if n_children[i] == 0:
                continue  # unexpanded node of a partial tree
            k64 = _key64(key_of[i])
            if k64 in holder:
                # two children of one node that the env turns into ONE public state (a bet size below the minimum raise is raised to it, the next size
                # is the minimum raise itself): they repeat each other node for node and no look-up could tell them apart -- the first keeps the rows
                a, pa, pb = holder[k64], dec_parent(holder[k64]), dec_parent(i)
                if pa >= 0 and pb >= 0 and twin_of.get(pa, pa) == twin_of.get(pb, pb) and tree._board_id[a] == tree._board_id[i]:
                    twin_of[i] = twin_of.get(a, a)
                    continue
                raise ValueError("nodes %d and %d (actions %s / %s) share a 64-bit history key without being such twins: build the table under another "
                                 "key_seed" % (a, i, tree.node(a).action, tree.node(i).action))
            holder[k64] = i
            pr = np.zeros((n_actions, cols.shape[1]), np.float32)
            for j, a in enumerate(tree.node(i).allowed_actions):
                pr[a] = cols[first_col[i] + j].astype(np.float32)
            keys.append(key_of[i])
            rows.append(pr)
        return cls(keys, np.stack(rows), key_seed=key_seed, node_keys=key_of)

    @classmethod
    def from_cfr(cls, cfr, t_idx=0, key_seed=KEY_SEED):
        """the AVERAGE strategy of a CFR instance's tree (what the reference evaluates, _CFRBase.py:218-262)"""
        return cls.from_tree(cfr._trees[t_idx], columns=cfr.average_strategy(t_idx), key_seed=key_seed)

    @classmethod
    def from_solver(cls, obj, t_idx=0, key_seed=KEY_SEED):
        """The average strategy of a solver as a table built ON THE DEVICE (prl_policy_table_from_solver): no host copy of the columns, any size the GPU
        holds -- the whole-game Flop5Holdem solve is 806 758 rows, 12.8 GB. obj: a CFR instance (its tree t_idx), a PublicTree, or a NativeSolver.
        A solve over suit classes gives a suit-canonical table (module docstring)."""
        solver = obj
        if hasattr(obj, "_trees"):
            solver = obj._trees[t_idx].solver
        elif hasattr(obj, "native_tree"):
            solver = obj.solver
        L = solver._L
        L.prl_policy_table_from_solver.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32, ctypes.POINTER(ctypes.c_void_p)]
        L.prl_policy_table_from_solver.restype = ctypes.c_int32
        h = ctypes.c_void_p()
        _native.check(L.prl_policy_table_from_solver(solver._h, solver.tree.handle, int(key_seed) & 0xFFFFFFFF, ctypes.byref(h)), L)
        return cls._wrap_device(L, h)

    @classmethod
    def _wrap_device(cls, L, h):
        self = cls.__new__(cls)
        L.prl_policy_table_info.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.prl_policy_table_info.restype = ctypes.c_int32
        L.prl_policy_table_export_keys.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.prl_policy_table_export_keys.restype = ctypes.c_int32
        L.prl_policy_table_get_rows.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p]
        L.prl_policy_table_get_rows.restype = ctypes.c_int32
        info = np.zeros(6, np.int64)
        _native.check(L.prl_policy_table_info(h, info.ctypes.data_as(ctypes.c_void_p)), L)
        self.n_rows, self.n_actions, self.range_size, self.capacity = (int(x) for x in info[:4])
        self.suit_canon, self.key_seed = bool(info[4]), int(info[5])
        self.probs, self.node_keys = None, {}
        self.keys, self.rows = np.zeros(self.capacity, np.uint64), np.zeros(self.capacity, np.int32)
        _native.check(L.prl_policy_table_export_keys(h, self.keys.ctypes.data_as(ctypes.c_void_p), self.rows.ctypes.data_as(ctypes.c_void_p)), L)
        self._row_of = None  # looked up in the open-addressed arrays (row_of): a dict of 800 k keys is not needed
        self._dev = (L, h)
        self._node_key_cache = weakref.WeakKeyDictionary()
        self._row_cache = {}
        return self

    # ---- host lookups --------------------------------------------------------------------------------------------------------------------------------
    def row_of(self, hk):
        if self._row_of is not None:
            return self._row_of.get(_key64(hk), -1)
        k, mask = _key64(hk), self.capacity - 1
        i = _first_slot(hk, mask)
        while True:  # the same probing as the device (lbrb_table_row)
            ki = int(self.keys[i])
            if ki == k:
                return int(self.rows[i])
            if ki == 0:
                return -1
            i = (i + 1) & mask

    def row_probs(self, r):
        """float32 [n_actions, R] of row r, in the table's own hand labelling (fetched from the device for device-built tables)"""
        if self.probs is not None:
            return self.probs[r]
        if r not in self._row_cache:
            if len(self._row_cache) > 256:
                self._row_cache.clear()
            out = np.zeros((self.n_actions, self.range_size), np.float32)
            L, h = self._dev
            _native.check(L.prl_policy_table_get_rows(h, int(r), 1, out.ctypes.data_as(ctypes.c_void_p)), L)
            self._row_cache[r] = out
        return self._row_cache[r]

    def policy(self, hk, legal, canon_k=0):
        """float32 [R, N_ACTIONS] at history `hk`; uniform over `legal` when the table does not hold it. canon_k: the suit permutation number that took
        the board to its canonical form (suit-canonical tables: hand h's probabilities are the row's at hand_perm(canon_k)[h])"""
        r = self.row_of(hk)
        if r >= 0:
            pr = self.row_probs(r)
            if self.suit_canon and canon_k:
                pr = pr[:, hand_perm(canon_k)]
            return np.ascontiguousarray(pr.T)
        p = np.zeros((self.range_size, self.n_actions), np.float32)
        p[:, list(legal)] = np.float32(1.0 / len(legal))
        return p

    def key_of_node(self, node):
        """the history key of a PublicTree node, from the node's PATH (its ancestors' env states) -- any tree of the game will do, whatever its node
        numbering (an evaluator's own tree, put_out_new_round_after_limit, another board list)"""
        tree = node.tree
        cache = self._node_key_cache.setdefault(tree, {})

        def rec(i):
            if i in cache:
                return cache[i]
            p = int(tree._parent[i])
            pk = hist_root(self.key_seed) if p < 0 else rec(p)
            if tree._kind[i] == KIND_CHANCE:
                cache[i] = pk  # the deal belongs to the step that ended the round
            else:
                env = tree._get_replay_env()
                env.load_state_dict(tree.node(i).env_state, blank_private_info=True)
                cache[i] = hist_step(pk, env, canon=self.suit_canon)
            return cache[i]

        return rec(int(node._i))

    # ---- the device copy -------------------------------------------------------------------------------------------------------------------------
    def device(self):
        if self._dev is None:
            L = _native.lib()
            _native.require_device()
            h = L.prl_policy_table_create(self.keys.ctypes.data_as(ctypes.c_void_p), self.rows.ctypes.data_as(ctypes.c_void_p), self.capacity,
                                          self.probs.ctypes.data_as(ctypes.c_void_p), self.n_rows, self.n_actions, self.range_size, self.key_seed & 0xFFFFFFFF)
            if not h:
                raise _native.NativeError("prl_policy_table_create: " + L.prl_last_error().decode("utf-8", "replace"))
            self._dev = (L, h)
        return self._dev[1]

    def probe(self, hist_keys, actions, hands):
        """(rows, probs) of n look-ups done ON THE DEVICE: what the batched engines read for history key i, action i, hand i"""
        n = len(hist_keys)
        lo = np.ascontiguousarray([k[0] for k in hist_keys], np.uint32)
        hi = np.ascontiguousarray([k[1] for k in hist_keys], np.uint32)
        a, h = np.ascontiguousarray(actions, np.int32), np.ascontiguousarray(hands, np.int32)
        rows, probs = np.zeros(n, np.int32), np.zeros(n, np.float32)
        dev = self.device()
        L = self._dev[0]
        _native.check(L.prl_policy_table_probe(dev, n, *(x.ctypes.data_as(ctypes.c_void_p) for x in (lo, hi, a, h, rows, probs))), L)
        return rows, probs

    def close(self):
        if self._dev is not None:
            self._dev[0].prl_policy_table_destroy(self._dev[1])
            self._dev = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def make_table_agent_cls(EvalAgentBase, table, seed=7, record=None):
    """An EvalAgent that plays `table`: the host twin of agent kind "table" (action draws: the counter-based stream of the hash agent, rl/hash_agent.py)."""
    Base = make_agent_cls(EvalAgentBase, seed=seed, record=record)

    class TableAgent(Base):
        ALL_MODES = ["TABLE", "HASH2"]  # head-to-head tests pit the table against the hash agent (seed + 1): one class, two modes
        TABLE = table

        def __init__(self, t_prof, mode=None, device=None):
            super().__init__(t_prof=t_prof, mode=mode, device=device)
            self._hk = hist_root(self.TABLE.key_seed)

        def _advance(self):
            self._hk = hist_step(self._hk, self._internal_env_wrapper.env, canon=self.TABLE.suit_canon)

        def reset(self, deck_state_dict=None):
            super().reset(deck_state_dict=deck_state_dict)
            self._hk = hist_root(self.TABLE.key_seed)
            self._advance()

        def notify_of_reset(self):
            super().notify_of_reset()
            self._hk = hist_root(self.TABLE.key_seed)
            self._advance()

        def notify_of_action(self, p_id_acted, action_he_did):
            super().notify_of_action(p_id_acted=p_id_acted, action_he_did=action_he_did)
            self._advance()

        def notify_of_processed_tuple_action(self, p_id_acted, action_he_did):
            super().notify_of_processed_tuple_action(p_id_acted=p_id_acted, action_he_did=action_he_did)
            self._advance()

        def notify_of_raise_frac_action(self, p_id_acted, frac):
            super().notify_of_raise_frac_action(p_id_acted=p_id_acted, frac=frac)
            self._advance()

        def set_to_public_tree_node_state(self, node):
            super().set_to_public_tree_node_state(node=node)
            self._hk = self.TABLE.key_of_node(node)  # by the node's path, not its index: the evaluator's tree need not be the table's

        def env_state_dict(self):
            return {"env": super().env_state_dict(), "hk": self._hk}

        def load_env_state_dict(self, state_dict):
            super().load_env_state_dict(state_dict["env"])
            self._hk = state_dict["hk"]

        def get_a_probs_for_each_hand(self):
            if self._mode == "HASH2":
                return super().get_a_probs_for_each_hand()
            env = self._internal_env_wrapper.env
            k = _CanonBoardView(env).k if self.TABLE.suit_canon else 0
            return self.TABLE.policy(self._hk, env.get_legal_actions(), canon_k=k)

        def get_action(self, step_env=True, need_probs=False):
            out = super().get_action(step_env=step_env, need_probs=need_probs)
            if step_env:
                self._advance()
            return out

    return TableAgent
