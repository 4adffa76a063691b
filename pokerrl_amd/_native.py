"""
ctypes binding of libpokerrl_hip.so (C ABI declared in include/pokerrl_hip.h).

This is the ONLY way the package reaches the hot path. There is no Python / NumPy / CPU fallback: if the shared library
is missing the import raises, and every device entry point raises `NativeError` when no HIP device is usable. (Same role
as the reference's PokerRL/_/CppWrapper.py:10-27, but with flat pointers + explicit sizes and int32 status codes.)
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# prl_exchange_fn (include/pokerrl_hip.h): int32 (*)(void* user, const void* local_dev, void* gathered_dev, uint64 bytes_per_rank)
EXCHANGE_FN = ctypes.CFUNCTYPE(ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint64)
LIB_PATH = os.environ.get("POKERRL_AMD_LIB", os.path.join(_HERE, "lib", "libpokerrl_hip.so"))

PRL_MAX_BET_SIZES = 96


class NativeError(RuntimeError):
    """A negative status from the library; `.status` is the PRL_ERR_* code (include/pokerrl_hip.h)."""
    status = None


ERR_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_UNSUPPORTED, ERR_OOM, ERR_STATE = -1, -2, -3, -4, -5, -6


class PrlRules(ctypes.Structure):
    _fields_ = [
        ("n_hole_cards", ctypes.c_int32), ("n_ranks", ctypes.c_int32), ("n_suits", ctypes.c_int32),
        ("n_cards", ctypes.c_int32), ("range_size", ctypes.c_int32), ("n_rounds", ctypes.c_int32),
        ("board_cards_in_round", ctypes.c_int32 * 4), ("n_board_cards", ctypes.c_int32),
        ("btn_first_postflop", ctypes.c_int32), ("rank_rule", ctypes.c_int32),
    ]


class PrlGame(ctypes.Structure):
    _fields_ = [
        ("game_type", ctypes.c_int32), ("n_rounds", ctypes.c_int32),
        ("small_blind", ctypes.c_int32), ("big_blind", ctypes.c_int32), ("ante", ctypes.c_int32),
        ("small_bet", ctypes.c_int32), ("big_bet", ctypes.c_int32), ("round_big_bet_starts", ctypes.c_int32),
        ("max_raises", ctypes.c_int32 * 4), ("first_action_no_call", ctypes.c_int32),
        ("btn_first_postflop", ctypes.c_int32), ("pot_size_raise", ctypes.c_int32), ("n_bet_sizes", ctypes.c_int32),
        ("start_stack", ctypes.c_int32 * 2), ("bet_fracs", ctypes.c_double * PRL_MAX_BET_SIZES),
    ]


class PrlEnvState(ctypes.Structure):
    _fields_ = [
        ("round", ctypes.c_int32), ("main_pot", ctypes.c_int32), ("bet", ctypes.c_int32 * 2),
        ("stack", ctypes.c_int32 * 2), ("allin", ctypes.c_int8 * 2), ("folded", ctypes.c_int8 * 2),
        ("acted", ctypes.c_int8 * 2), ("cur", ctypes.c_int8), ("last_raiser", ctypes.c_int8),
        ("capped_happened", ctypes.c_int8), ("capped_raiser", ctypes.c_int8), ("capped_cant_reopen", ctypes.c_int8),
        ("pad0", ctypes.c_int8), ("n_actions_ep", ctypes.c_int32), ("n_raises_round", ctypes.c_int32),
        ("last_action", ctypes.c_int32 * 3),
    ]


class PrlStepInfo(ctypes.Structure):
    _fields_ = [
        ("is_terminal", ctypes.c_int32), ("chance_acts", ctypes.c_int32), ("terminal_is_fold", ctypes.c_int32),
        ("rundown", ctypes.c_int32), ("pot_before_payout", ctypes.c_int32), ("fixed_type", ctypes.c_int32),
        ("fixed_amount", ctypes.c_int32),
    ]


# tree info / field ids (include/pokerrl_hip.h)
TI_N_NODES, TI_N_COLS, TI_N_BOARDS, TI_BOARD_LEN, TI_N_LEVELS, TI_RANGE_SIZE, TI_N_DECISION, TI_N_TERMINAL, TI_COUNT = range(9)
TREE_FIELDS = dict(kind=0, actor=1, parent=2, child_idx=3, action=4, acted_last=5, round=6, board_id=7, main_pot=8,
                   depth=9, n_children=10, first_col=11, subtree_size=12, child_start=13, child_list=14, col_action=15,
                   col_node=16, level_start=17, level_nodes=18)

_lib = None


def _ptr(a, ctype=None):
    return a.ctypes.data_as(ctypes.c_void_p)


def lib():
    """Loads the shared library once. Raises NativeError (never falls back) if it is absent."""
    global _lib
    if _lib is None:
        # PyTorch-ROCm ships its own HIP runtime. Whichever HIP runtime a process initialises first owns the GPU for that process:
        # with this library loaded first, a later `import torch` finds "No HIP GPUs" (seen on the MI355X box). Neural agents and
        # torch.distributed (RCCL) live in the same process as the solver, so torch goes first whenever it is installed.
        # (POKERRL_AMD_NO_TORCH_PRELOAD=1 skips this for purely tabular use; a torch that fails to import must not take this package down.)
        if os.environ.get("POKERRL_AMD_NO_TORCH_PRELOAD", "0") in ("", "0"):
            try:
                import torch  # noqa: F401
            except Exception as e:  # ImportError, or OSError / RuntimeError of a broken install
                if not isinstance(e, ImportError):
                    import warnings
                    warnings.warn("pokerrl_amd: `import torch` failed (%s: %s); continuing without it" % (type(e).__name__, e))
        _lib = bind(LIB_PATH)
    return _lib


def bind(path):
    """dlopen + prototype declaration. The package itself only ever binds LIB_PATH (the hipcc build)."""
    if not os.path.isfile(path):
        raise NativeError(
            "pokerrl_amd: native library %s not found. Build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback." % path)
    L = ctypes.CDLL(path)
    L.prl_last_error.restype = ctypes.c_char_p
    L.prl_build_flavor.restype = ctypes.c_char_p
    L.prl_device_available.restype = ctypes.c_int32
    L.prl_set_device.argtypes = [ctypes.c_int32]
    L.prl_set_device.restype = ctypes.c_int32
    for name in ("prl_lut_idx_2_hole_cards", "prl_lut_hole_cards_2_idx", "prl_lut_card_in_what_range_idxs"):
        getattr(L, name).argtypes = [ctypes.POINTER(PrlRules), ctypes.c_void_p]
        getattr(L, name).restype = ctypes.c_int32
    L.prl_hand_rank_7.argtypes = [ctypes.c_void_p, ctypes.c_int8, ctypes.c_int8]
    L.prl_hand_rank_7.restype = ctypes.c_int32
    L.prl_hand_rank_boards.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
    L.prl_hand_rank_boards.restype = ctypes.c_int32
    L.prl_hand_rank_boards_device.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]
    L.prl_hand_rank_boards_device.restype = ctypes.c_int32
    L.prl_tree_build.argtypes = [ctypes.POINTER(PrlGame), ctypes.POINTER(PrlRules), ctypes.c_void_p, ctypes.c_int32,
                                 ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p)]
    L.prl_tree_build.restype = ctypes.c_int32
    L.prl_tree_destroy.argtypes = [ctypes.c_void_p]
    L.prl_tree_destroy.restype = None
    L.prl_tree_info.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
    L.prl_tree_info.restype = ctypes.c_int32
    L.prl_tree_get.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
    L.prl_tree_get.restype = ctypes.c_int32
    L.prl_env_reset_host.argtypes = [ctypes.POINTER(PrlGame), ctypes.POINTER(PrlEnvState)]
    L.prl_env_reset_host.restype = ctypes.c_int32
    L.prl_env_step_host.argtypes = [ctypes.POINTER(PrlGame), ctypes.POINTER(PrlEnvState), ctypes.c_int32,
                                    ctypes.POINTER(PrlStepInfo)]
    L.prl_env_step_host.restype = ctypes.c_int32
    L.prl_env_step_processed_host.argtypes = [ctypes.POINTER(PrlGame), ctypes.POINTER(PrlEnvState), ctypes.c_int32,
                                              ctypes.c_int32, ctypes.POINTER(PrlStepInfo)]
    L.prl_env_step_processed_host.restype = ctypes.c_int32
    L.prl_env_apply_action_host.argtypes = [ctypes.POINTER(PrlGame), ctypes.POINTER(PrlEnvState), ctypes.c_int32, ctypes.c_int32,
                                            ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(PrlStepInfo)]
    L.prl_env_apply_action_host.restype = ctypes.c_int32
    L.prl_env_legal_actions_host.argtypes = [ctypes.POINTER(PrlGame), ctypes.POINTER(PrlEnvState), ctypes.c_void_p,
                                             ctypes.POINTER(ctypes.c_int32)]
    L.prl_env_legal_actions_host.restype = ctypes.c_int32
    L.prl_env_fraction_of_pot_raise_host.argtypes = [ctypes.POINTER(PrlEnvState), ctypes.c_double, ctypes.c_int32,
                                                     ctypes.POINTER(ctypes.c_int32)]
    L.prl_env_fraction_of_pot_raise_host.restype = ctypes.c_int32
    _bind_solver(L)
    return L


class _Lenient:
    """prototype declarations on a library that may lack newer symbols (same-box A/B runs bind older builds through POKERRL_AMD_LIB): a missing symbol's
    declaration goes to a throw-away object; CALLING it still fails loudly on the CDLL itself"""

    class _Missing:
        argtypes = restype = None

    def __init__(self, lib):
        object.__setattr__(self, "_lib", lib)

    def __getattr__(self, name):
        lib = object.__getattribute__(self, "_lib")
        return getattr(lib, name) if hasattr(lib, name) else _Lenient._Missing()


def _bind_solver(L):
    """Device-side solver entry points (declared in include/pokerrl_hip.h section 5)."""
    if not hasattr(L, "prl_solver_create"):
        return
    L = _Lenient(L)
    vp, i32, f64 = ctypes.c_void_p, ctypes.c_int32, ctypes.c_double
    L.prl_solver_create.argtypes = [vp, i32, i32, ctypes.POINTER(vp)]
    L.prl_solver_create_ex.argtypes = [vp, i32, i32, i32, ctypes.POINTER(vp)]
    L.prl_solver_create_ex.restype = i32
    if hasattr(L, "prl_solver_create_weighted"):  # (A/B runs bind older builds of the library)
        L.prl_solver_create_weighted.argtypes = [vp, i32, i32, i32, vp, i32, ctypes.POINTER(vp)]
        L.prl_solver_create_weighted.restype = i32
    L.prl_solver_create_sharded.argtypes = [vp, i32, i32, i32, i32, EXCHANGE_FN, vp, ctypes.POINTER(vp)]
    L.prl_solver_create_sharded.restype = i32
    L.prl_solver_create_sharded_ragged.argtypes = [vp, i32, i32, i32, i32, ctypes.c_int64, ctypes.c_int64, EXCHANGE_FN, vp, ctypes.POINTER(vp)]
    L.prl_solver_create_sharded_ragged.restype = i32
    L.prl_solver_time_iterations_ex.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(i32)]
    L.prl_solver_time_iterations_ex.restype = i32
    L.prl_solver_time_evaluations.argtypes = [vp, i32, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(i32)]
    L.prl_solver_time_evaluations.restype = i32
    L.prl_lbr_checkdown_equity.argtypes = [ctypes.POINTER(PrlRules), vp, i32, vp, vp, i32, vp]
    L.prl_lbr_checkdown_equity.restype = i32
    L.prl_lbr_batch_run.argtypes = [ctypes.POINTER(PrlGame), ctypes.POINTER(PrlGame), ctypes.POINTER(PrlRules), i32, i32, i32, i32,
                                    ctypes.c_uint32, ctypes.c_uint32, ctypes.c_double, ctypes.c_double, vp, vp, vp, ctypes.POINTER(ctypes.c_float)]
    L.prl_lbr_batch_run.restype = i32
    L.prl_h2h_batch_run.argtypes = [ctypes.POINTER(PrlGame), ctypes.POINTER(PrlRules), i32, i32, i32, ctypes.c_uint32, i32, ctypes.c_uint32,
                                    ctypes.c_uint32, ctypes.c_double, ctypes.c_double, vp, vp, vp, ctypes.POINTER(ctypes.c_float)]
    L.prl_h2h_batch_run.restype = i32
    L.prl_policy_table_create.argtypes = [vp, vp, ctypes.c_uint32, vp, i32, i32, i32, ctypes.c_uint32]
    L.prl_policy_table_create.restype = vp
    L.prl_policy_table_destroy.argtypes = [vp]
    L.prl_policy_table_destroy.restype = None
    L.prl_policy_table_probe.argtypes = [vp, i32, vp, vp, vp, vp, vp, vp]
    L.prl_policy_table_probe.restype = i32
    L.prl_lbr_batch_run_table.argtypes = [ctypes.POINTER(PrlGame), ctypes.POINTER(PrlGame), ctypes.POINTER(PrlRules), i32, i32, i32, vp,
                                          ctypes.c_uint32, ctypes.c_uint32, ctypes.c_double, ctypes.c_double, vp, vp, vp, ctypes.POINTER(ctypes.c_float)]
    L.prl_lbr_batch_run_table.restype = i32
    L.prl_h2h_batch_run_tables.argtypes = [ctypes.POINTER(PrlGame), ctypes.POINTER(PrlRules), i32, i32, i32, ctypes.c_uint32, vp, i32, ctypes.c_uint32, vp,
                                           ctypes.c_uint32, ctypes.c_double, ctypes.c_double, vp, vp, vp, ctypes.POINTER(ctypes.c_float)]
    L.prl_h2h_batch_run_tables.restype = i32
    L.prl_solver_iterations_many.argtypes = [ctypes.POINTER(vp), i32, i32]
    L.prl_solver_iterations_many.restype = i32
    L.prl_deal_decks.argtypes = [i32, i32, i32, ctypes.c_uint64, ctypes.c_uint64, vp]
    L.prl_deal_decks.restype = i32
    L.prl_envbatch_create.argtypes = [ctypes.POINTER(PrlGame), i32, ctypes.POINTER(vp)]
    L.prl_envbatch_create.restype = i32
    L.prl_envbatch_destroy.argtypes = [vp]
    L.prl_envbatch_destroy.restype = None
    for name, args in (("prl_envbatch_reset", [vp, vp]), ("prl_envbatch_step", [vp, vp, vp, vp]), ("prl_envbatch_step_device", [vp, vp, vp, vp]),
                       ("prl_envbatch_legal_masks", [vp, vp, vp]), ("prl_envbatch_active", [vp, vp, vp]), ("prl_envbatch_get_state", [vp, vp]),
                       ("prl_envbatch_state_device", [vp, ctypes.POINTER(vp), ctypes.POINTER(vp)]),
                       ("prl_envbatch_random_rollout", [vp, i32, ctypes.c_uint32, vp, ctypes.POINTER(ctypes.c_float)]),
                       ("prl_envbatch_random_steps", [vp, i32, ctypes.c_uint32, vp, ctypes.POINTER(ctypes.c_float)]),
                       ("prl_env_random_rollout_host", [ctypes.POINTER(PrlGame), i32, i32, ctypes.c_uint32, vp])):
        getattr(L, name).argtypes = args
        getattr(L, name).restype = i32
    L.prl_solver_state_size.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    L.prl_solver_state_size.restype = i32
    L.prl_solver_save_state.argtypes = [vp, vp, ctypes.c_uint64]
    L.prl_solver_save_state.restype = i32
    L.prl_solver_load_state.argtypes = [vp, vp, ctypes.c_uint64]
    L.prl_solver_load_state.restype = i32
    L.prl_solver_get_stream.argtypes = [vp, ctypes.POINTER(vp)]
    L.prl_solver_get_stream.restype = i32
    L.prl_solver_set_exchange_async.argtypes = [vp, i32]
    L.prl_solver_set_exchange_async.restype = i32
    L.prl_chance_sum_host.argtypes = [vp, i32, i32, i32, vp]
    L.prl_chance_sum_host.restype = i32
    L.prl_chance_sum_host_ragged.argtypes = [vp, i32, i32, i32, i32, vp]
    L.prl_chance_sum_host_ragged.restype = i32
    L.prl_solver_get.argtypes = [vp, i32, vp]
    L.prl_solver_get.restype = i32
    L.prl_solver_create.restype = i32
    L.prl_solver_destroy.argtypes = [vp]
    L.prl_solver_destroy.restype = None
    for name, args in {
        "prl_solver_reset": [vp],
        "prl_solver_iteration": [vp],
        "prl_solver_iterations": [vp, i32],
        "prl_solver_fill_uniform": [vp],
        "prl_solver_set_strategy": [vp, vp, i32],
        "prl_solver_update_reach": [vp],
        "prl_solver_compute_ev": [vp],
        "prl_solver_get": [vp, i32, vp],
        "prl_solver_exploitability": [vp, vp],
        "prl_solver_eval_avg": [vp, vp],
        "prl_solver_sync": [vp],
        "prl_solver_set_option": [vp, i32, i32],
        "prl_solver_time_iterations": [vp, i32, vp],
    }.items():
        if hasattr(L, name):
            getattr(L, name).argtypes = args
            getattr(L, name).restype = i32


def check(status, L=None):
    if status != 0:
        L = L or lib()
        e = NativeError("libpokerrl_hip error %d: %s" % (status, L.prl_last_error().decode("utf-8", "replace")))
        e.status = int(status)
        raise e


def device_available():
    return bool(lib().prl_device_available())


def set_device(ordinal):
    """one process per GPU: bind this process's handles to HIP device `ordinal` (LOCAL_RANK)"""
    check(lib().prl_set_device(int(ordinal)))


def require_device():
    if not device_available():
        raise NativeError("pokerrl_amd: no usable HIP device (MI355X / gfx950 required). There is no CPU fallback for "
                          "the hot path; tests that need the GPU are marked @pytest.mark.gpu.")


def build_flavor():
    return lib().prl_build_flavor().decode()


# ---------------------------------------------------------------------------------------------------------------------
# thin wrappers
# ---------------------------------------------------------------------------------------------------------------------
def lut_idx_2_hole_cards(rules):
    out = np.empty((rules.range_size, rules.n_hole_cards), dtype=np.int8)
    check(lib().prl_lut_idx_2_hole_cards(ctypes.byref(rules), _ptr(out)))
    return out


def lut_hole_cards_2_idx(rules):
    shape = (rules.n_cards, 1) if rules.n_hole_cards == 1 else (rules.n_cards, rules.n_cards)
    out = np.empty(shape, dtype=np.int16)
    check(lib().prl_lut_hole_cards_2_idx(ctypes.byref(rules), _ptr(out)))
    return out


def lut_card_in_what_range_idxs(rules):
    shape = (rules.n_cards, 1) if rules.n_hole_cards == 1 else (rules.n_cards, rules.n_cards - 1)
    out = np.empty(shape, dtype=np.int32)
    check(lib().prl_lut_card_in_what_range_idxs(ctypes.byref(rules), _ptr(out)))
    return out


def hand_rank_7(board_1d, c1, c2):
    b = np.ascontiguousarray(board_1d, dtype=np.int8)
    assert b.shape == (5,)
    return int(lib().prl_hand_rank_7(_ptr(b), int(c1), int(c2)))


def hand_rank_boards(boards_1d):
    """[n, 5] int8 boards -> [n, 1326] int32 ranks (-1 = blocked) on the GPU."""
    require_device()
    b = np.ascontiguousarray(boards_1d, dtype=np.int8)
    assert b.ndim == 2 and b.shape[1] == 5
    out = np.empty((b.shape[0], 1326), dtype=np.int32)
    check(lib().prl_hand_rank_boards(_ptr(b), b.shape[0], _ptr(out)))
    return out


EB_COLS = ("round", "main_pot", "bet0", "bet1", "stack0", "stack1", "flags", "seats", "n_actions_ep", "n_raises_round",
           "last_action_type", "last_action_amount", "last_action_seat")


class NativeEnvBatch:
    """Owns a prl_envbatch_t*: n_envs heads-up envs stepped together on the GPU, public state as struct-of-arrays in HBM
    (include/pokerrl_hip.h section 4b). Cards are the caller's, as with the single-env engine."""

    def __init__(self, game, n_envs, _lib=None):
        self._L = _lib or lib()
        self._h = ctypes.c_void_p()
        self._game = game
        self.n_envs = int(n_envs)
        check(self._L.prl_envbatch_create(ctypes.byref(game), self.n_envs, ctypes.byref(self._h)), self._L)

    def reset(self, mask=None):
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        assert m is None or m.shape == (self.n_envs,)
        check(self._L.prl_envbatch_reset(self._h, None if m is None else _ptr(m)), self._L)

    def step(self, actions, amounts=None):
        """-> info int32 [4, n_envs]: is_terminal (-1 = env skipped), chance_acts, pot before the payout, terminal kind"""
        a = np.ascontiguousarray(actions, dtype=np.int32)
        b = None if amounts is None else np.ascontiguousarray(amounts, dtype=np.int32)
        assert a.shape == (self.n_envs,) and (b is None or b.shape == a.shape)
        info = np.empty((4, self.n_envs), np.int32)
        check(self._L.prl_envbatch_step(self._h, _ptr(a), None if b is None else _ptr(b), _ptr(info)), self._L)
        return info

    def legal_masks(self):
        """-> (uint32 [4, n_envs] bit sets of legal action ints, int32 [n_envs] counts)"""
        m, c = np.empty((4, self.n_envs), np.uint32), np.empty(self.n_envs, np.int32)
        check(self._L.prl_envbatch_legal_masks(self._h, _ptr(m), _ptr(c)), self._L)
        return m, c

    def legal_actions(self, i):
        m, _c = self.legal_masks()
        return [a for a in range(128) if (int(m[a >> 5, i]) >> (a & 31)) & 1]

    def active(self):
        idx, n = np.empty(self.n_envs, np.int32), np.zeros(1, np.int32)
        check(self._L.prl_envbatch_active(self._h, _ptr(idx), _ptr(n)), self._L)
        return idx[:int(n[0])]

    def state(self):
        """-> dict of int32 [n_envs] columns (EB_COLS) + the unpacked flag / seat fields under the PrlEnvState names"""
        cols = np.empty((len(EB_COLS), self.n_envs), np.int32)
        check(self._L.prl_envbatch_get_state(self._h, _ptr(cols)), self._L)
        d = {k: cols[i] for i, k in enumerate(EB_COLS)}
        f, w = d["flags"], d["seats"]
        for k, bit in (("allin0", 0), ("allin1", 1), ("folded0", 2), ("folded1", 3), ("acted0", 4), ("acted1", 5), ("cur", 6),
                       ("capped_happened", 7), ("done", 8)):
            d[k] = (f >> bit) & 1
        d["last_raiser"], d["capped_raiser"], d["capped_cant_reopen"] = (w & 0xFF) - 1, ((w >> 8) & 0xFF) - 1, ((w >> 16) & 0xFF) - 1
        return d

    def random_rollout(self, n_steps, seed):
        """n_steps uniform-random legal steps per env on the device -> (steps, finished hands, sum of their pots, kernel ms)"""
        st, ms = np.zeros(3, np.uint64), ctypes.c_float()
        check(self._L.prl_envbatch_random_rollout(self._h, int(n_steps), int(seed) & 0xFFFFFFFF, _ptr(st), ctypes.byref(ms)), self._L)
        return int(st[0]), int(st[1]), int(st[2]), float(ms.value)

    def random_steps(self, n_launches, seed):
        """the same play, ONE step per env and kernel launch (state in HBM between the steps) -> (steps, hands, pots, device ms)"""
        st, ms = np.zeros(3, np.uint64), ctypes.c_float()
        check(self._L.prl_envbatch_random_steps(self._h, int(n_launches), int(seed) & 0xFFFFFFFF, _ptr(st), ctypes.byref(ms)), self._L)
        return int(st[0]), int(st[1]), int(st[2]), float(ms.value)

    # ---- the whole PokerEnv.step: cards, payouts, rewards, observations (prl_envbatch_create_with_cards) --------------------------
    @classmethod
    def with_cards(cls, game, rules, n_envs, deck_seed=0, reward_scalar=1.0, _lib=None):
        """n envs that also hold their hole cards and board (dealt from a counter-based deck at every reset, or set by the caller):
        step_full() returns (obs, reward, done, info) like n calls of PokerEnv.step (PokerEnv.py:737-789)."""
        self = cls.__new__(cls)
        self._L = _lib or lib()
        self._h = ctypes.c_void_p()
        self._game, self._rules = game, rules
        self.n_envs = int(n_envs)
        L = self._L
        L.prl_envbatch_create_with_cards.argtypes = [ctypes.POINTER(PrlGame), ctypes.POINTER(PrlRules), ctypes.c_int32, ctypes.c_uint64, ctypes.c_double,
                                                     ctypes.POINTER(ctypes.c_void_p)]
        L.prl_envbatch_create_with_cards.restype = ctypes.c_int32
        for name, args in (("prl_envbatch_obs_dim", [ctypes.c_void_p, ctypes.c_void_p]), ("prl_envbatch_reset_full", [ctypes.c_void_p] * 3),
                           ("prl_envbatch_set_cards", [ctypes.c_void_p] * 2), ("prl_envbatch_get_cards", [ctypes.c_void_p] * 2),
                           ("prl_envbatch_observe", [ctypes.c_void_p] * 2), ("prl_envbatch_step_full", [ctypes.c_void_p] * 7),
                           ("prl_envbatch_random_rollout_full", [ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p])):
            getattr(L, name).argtypes = args
            getattr(L, name).restype = ctypes.c_int32
        check(L.prl_envbatch_create_with_cards(ctypes.byref(game), ctypes.byref(rules), self.n_envs, int(deck_seed), float(reward_scalar), ctypes.byref(self._h)), L)
        d = ctypes.c_int32()
        check(L.prl_envbatch_obs_dim(self._h, ctypes.byref(d)), L)
        self.obs_dim = int(d.value)
        self.n_deal = 2 * rules.n_hole_cards + rules.n_board_cards
        return self

    def reset_full(self, mask=None):
        """reset the masked envs (all if None): public state, a fresh hand from the deck -> observations float32 [n_envs, obs_dim] of ALL envs' states"""
        m = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
        check(self._L.prl_envbatch_reset_full(self._h, None if m is None else _ptr(m), None), self._L)
        return self.observe()

    def set_cards(self, cards):
        c = np.ascontiguousarray(cards, dtype=np.int8)
        assert c.shape == (self.n_envs, self.n_deal)
        check(self._L.prl_envbatch_set_cards(self._h, _ptr(c)), self._L)

    def get_cards(self):
        c = np.empty((self.n_envs, self.n_deal), np.int8)
        check(self._L.prl_envbatch_get_cards(self._h, _ptr(c)), self._L)
        return c

    def observe(self):
        o = np.empty((self.n_envs, self.obs_dim), np.float32)
        check(self._L.prl_envbatch_observe(self._h, _ptr(o)), self._L)
        return o

    def step_full(self, actions, amounts=None):
        """-> (obs float32 [n, obs_dim], reward float64 [n, 2], done uint8 [n], info int32 [4, n])"""
        a = np.ascontiguousarray(actions, dtype=np.int32)
        b = None if amounts is None else np.ascontiguousarray(amounts, dtype=np.int32)
        obs, rew = np.empty((self.n_envs, self.obs_dim), np.float32), np.empty((self.n_envs, 2), np.float64)
        done, info = np.empty(self.n_envs, np.uint8), np.empty((4, self.n_envs), np.int32)
        check(self._L.prl_envbatch_step_full(self._h, _ptr(a), None if b is None else _ptr(b), _ptr(obs), _ptr(rew), _ptr(done), _ptr(info)), self._L)
        return obs, rew, done, info

    def last_outputs(self):
        """-> (obs, reward, done) of the last random_steps_full launch (the batch's own output buffers)"""
        obs, rew, done = np.empty((self.n_envs, self.obs_dim), np.float32), np.empty((self.n_envs, 2), np.float64), np.empty(self.n_envs, np.uint8)
        self._L.prl_envbatch_last_outputs.argtypes = [ctypes.c_void_p] * 4
        self._L.prl_envbatch_last_outputs.restype = ctypes.c_int32
        check(self._L.prl_envbatch_last_outputs(self._h, _ptr(obs), _ptr(rew), _ptr(done)), self._L)
        return obs, rew, done

    def random_steps_full(self, n_launches, seed):
        """one WHOLE step (obs, rewards, done out) per env and launch, state in HBM between the launches -> (steps, hands, pots, device ms)"""
        st, ms = np.zeros(3, np.uint64), ctypes.c_float()
        self._L.prl_envbatch_random_steps_full.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
        self._L.prl_envbatch_random_steps_full.restype = ctypes.c_int32
        check(self._L.prl_envbatch_random_steps_full(self._h, int(n_launches), int(seed) & 0xFFFFFFFF, _ptr(st), ctypes.byref(ms)), self._L)
        return int(st[0]), int(st[1]), int(st[2]), float(ms.value)

    def random_rollout_full(self, n_steps, seed):
        """whole hands in registers -> (steps, finished hands, showdowns, payout checksum, kernel ms)"""
        st, ms = np.zeros(4, np.uint64), ctypes.c_float()
        check(self._L.prl_envbatch_random_rollout_full(self._h, int(n_steps), int(seed) & 0xFFFFFFFF, _ptr(st), ctypes.byref(ms)), self._L)
        return int(st[0]), int(st[1]), int(st[2]), int(st[3]), float(ms.value)

    def __del__(self):
        try:
            if self._h:
                self._L.prl_envbatch_destroy(self._h)
                self._h = None
        except Exception:
            pass


def env_random_rollout_full_host(game, rules, n_envs, n_steps, seed, deck_seed=0, reward_scalar=1.0, _lib=None):
    L = _lib or lib()
    st = np.zeros(4, np.uint64)
    L.prl_env_random_rollout_full_host.argtypes = [ctypes.POINTER(PrlGame), ctypes.POINTER(PrlRules), ctypes.c_int32, ctypes.c_int32, ctypes.c_uint32,
                                                   ctypes.c_uint64, ctypes.c_double, ctypes.c_void_p]
    L.prl_env_random_rollout_full_host.restype = ctypes.c_int32
    check(L.prl_env_random_rollout_full_host(ctypes.byref(game), ctypes.byref(rules), int(n_envs), int(n_steps), int(seed) & 0xFFFFFFFF, int(deck_seed),
                                             float(reward_scalar), _ptr(st)), L)
    return tuple(int(x) for x in st)


def env_random_rollout_host(game, n_envs, n_steps, seed, _lib=None):
    L = _lib or lib()
    st = np.zeros(3, np.uint64)
    check(L.prl_env_random_rollout_host(ctypes.byref(game), int(n_envs), int(n_steps), int(seed) & 0xFFFFFFFF, _ptr(st)), L)
    return int(st[0]), int(st[1]), int(st[2])


class NativeTree:
    """Owns a prl_tree_t* (host-side flat public tree)."""

    def __init__(self, game, rules, boards_1d, _lib=None, stop_at_round=None):
        boards = np.ascontiguousarray(boards_1d, dtype=np.int8)
        assert boards.ndim == 2
        self._L = _lib or lib()
        self._h = ctypes.c_void_p()
        self._game, self._rules = game, rules
        self.is_partial = stop_at_round is not None
        self._L.prl_tree_build_partial.argtypes = [ctypes.POINTER(PrlGame), ctypes.POINTER(PrlRules), ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32,
                                                   ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p)]
        self._L.prl_tree_build_partial.restype = ctypes.c_int32
        check(self._L.prl_tree_build_partial(ctypes.byref(game), ctypes.byref(rules), _ptr(boards), boards.shape[0], boards.shape[1],
                                             -1 if stop_at_round is None else int(stop_at_round), ctypes.byref(self._h)), self._L)
        info = np.zeros(TI_COUNT, dtype=np.int32)
        check(self._L.prl_tree_info(self._h, _ptr(info)), self._L)
        self.info = info
        self.n_nodes, self.n_cols = int(info[TI_N_NODES]), int(info[TI_N_COLS])
        self.n_boards, self.board_len = int(info[TI_N_BOARDS]), int(info[TI_BOARD_LEN])
        self.n_levels, self.range_size = int(info[TI_N_LEVELS]), int(info[TI_RANGE_SIZE])
        self.n_decision, self.n_terminal = int(info[TI_N_DECISION]), int(info[TI_N_TERMINAL])
        self.boards = boards  # the caller's run-outs
        rows = np.empty((self.n_boards, self.board_len), np.int8)  # the board table node.board_id indexes (prefix rows, -1 = not dealt)
        self._L.prl_tree_get_boards.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self._L.prl_tree_get_boards.restype = ctypes.c_int32
        check(self._L.prl_tree_get_boards(self._h, _ptr(rows)), self._L)
        self.board_rows = rows
        self._cache = {}

    @classmethod
    def for_game(cls, game_cls, stack, bet_sizes, boards_1d, _lib=None):
        """The public tree of `game_cls` (pokerrl_amd.game.games) with equal starting stacks and the given agent bet set over
        `boards_1d` -- PublicTree(env_bldr, stack_size).build_tree() of the reference (PublicTree.py:30-126) in one call."""
        args = game_cls.ARGS_CLS(n_seats=2, starting_stack_sizes_list=[stack, stack], bet_sizes_list_as_frac_of_pot=bet_sizes)
        return cls(game_cls.native_game(args), game_cls.native_rules(), boards_1d, _lib=_lib)

    @property
    def handle(self):
        return self._h

    def field(self, name):
        if name not in self._cache:
            n = {"child_start": self.n_nodes + 1, "child_list": max(self.n_nodes - 1, 0), "col_action": self.n_cols,
                 "col_node": self.n_cols, "level_start": self.n_levels + 1}.get(name, self.n_nodes)
            out = np.empty(n, dtype=np.int32)
            check(self._L.prl_tree_get(self._h, TREE_FIELDS[name], _ptr(out)), self._L)
            self._cache[name] = out
        return self._cache[name]

    def __del__(self):
        try:
            if self._h:
                self._L.prl_tree_destroy(self._h)
                self._h = None
        except Exception:
            pass


# solver field ids (include/pokerrl_hip.h)
SF = dict(reach=0, ev=1, ev_br=2, strategy=3, strat_f64=4, regret=5, avg=6, avg_f64=7, avg_sum=8, br_idx=9,
          expl_history=10, iter=11, constants=12, bytes_allocated=13, engine=14, graph_replay=15, explicit_strategy=16, exchanges=17, vmm_ranges=18)
VARIANTS = {"vanilla": 0, "plus": 1, "linear": 2}
ENGINES = {"auto": 0, "levels": 1, "fused": 2}


class NativeSolver:
    """Owns a prl_solver_t* : the device-resident CFR / best-response solver of one public tree.

    shard=(world_size, rank, exchange): sharded solve, `tree` holds this rank's contiguous block of the global board list
    and `exchange(local_ptr, gathered_ptr, bytes_per_rank)` all-gathers device buffers (pokerrl_amd.dist.TorchExchange);
    every call that evaluates the tree is then collective. Results are bit-identical to the unsharded solve.
    shard=(world_size, rank, exchange, shard_boards, total_boards): ragged shards -- every rank before the last holds shard_boards
    boards, the last one the rest (prl_solver_create_sharded_ragged)."""

    def __init__(self, tree, variant, delay=0, engine="auto", _lib=None, shard=None, avg_dtype="f64", place=None, probe_iters=4, board_mult=None,
                 symmetrize=False):
        """place=k (unsharded solves): placement selection -- the library builds up to k solvers side by side, times probe_iters steady-state
        iterations of each and keeps the fastest (prl_solver_create_placed; .placement_ms / .placement_chosen say what it saw).
        board_mult=int array [n_boards] (+ symmetrize=True): weighted boards / suit isomorphism (prl_solver_create_weighted) -- the listed boards
        stand for board_mult[i] boards each; with symmetrize they are suit-class representatives (pokerrl_amd.game.board_enum.
        single_deal_board_classes) and the chance node's values are averaged over every hand's suit orbit: the WHOLE game from its classes.
        The library checks that claim (representatives, orbit sizes, the whole game covered); symmetrize="subset" admits a subset of the classes."""
        self._L = _lib or tree._L
        self.placement_ms, self.placement_chosen = None, None
        if _lib is None and self._L is lib():
            require_device()
        self.tree = tree
        self._h = ctypes.c_void_p()
        v = VARIANTS[variant] if isinstance(variant, str) else int(variant)
        e = ENGINES[engine] if isinstance(engine, str) else int(engine)
        self._exchange_cb = None
        if board_mult is not None and shard is not None:
            raise ValueError("weighted boards (board_mult=): one GPU -- not with shard=")
        if symmetrize and board_mult is None:
            raise ValueError("symmetrize= goes with board_mult= (suit-class representatives and their orbit sizes)")
        if shard is not None and isinstance(shard[0], str):
            # ("rccl", world, rank, unique_id bytes[, shard_boards, total_boards]): the exchange lives in the library (ncclAllGather on
            # the solver's stream, prl_solver_create_sharded_rccl) -- no Python in the iteration loop. pokerrl_amd.dist.rccl_shard builds it.
            assert shard[0] == "rccl"
            world, rank, uid = shard[1:4]
            uid = (ctypes.c_char * 128).from_buffer_copy(bytes(uid))
            sb, tb = (int(shard[4]), int(shard[5])) if len(shard) == 6 else (0, 0)
            self._L.prl_solver_create_sharded_rccl.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p,
                                                               ctypes.c_int64, ctypes.c_int64, ctypes.POINTER(ctypes.c_void_p)]
            self._L.prl_solver_create_sharded_rccl.restype = ctypes.c_int32
            check(self._L.prl_solver_create_sharded_rccl(tree.handle, v, int(delay), int(world), int(rank), ctypes.cast(uid, ctypes.c_void_p), sb, tb,
                                                         ctypes.byref(self._h)), self._L)
            shard = None
        elif shard is not None:
            world, rank, exchange = shard[:3]

            def _cb(_user, local_ptr, gathered_ptr, nbytes):
                try:
                    exchange(int(local_ptr or 0), int(gathered_ptr or 0), int(nbytes))
                    return 0
                except Exception:  # an exception must not unwind through the C frame
                    import traceback
                    traceback.print_exc()
                    return 1

            self._exchange_cb = EXCHANGE_FN(_cb)  # kept alive with the solver
            if len(shard) == 5:
                check(self._L.prl_solver_create_sharded_ragged(tree.handle, v, int(delay), int(world), int(rank), int(shard[3]), int(shard[4]),
                                                               self._exchange_cb, None, ctypes.byref(self._h)), self._L)
            else:
                check(self._L.prl_solver_create_sharded(tree.handle, v, int(delay), int(world), int(rank), self._exchange_cb, None,
                                                        ctypes.byref(self._h)), self._L)
        elif board_mult is not None:
            m = np.ascontiguousarray(board_mult, np.int32)
            assert m.shape == (tree.n_boards,), "one multiplicity per listed board"
            sym = 2 if symmetrize == "subset" else (1 if symmetrize else 0)  # PRL_SYMMETRIZE_SUBSET: class representatives that do not cover the game
            if place is not None:  # placement selection among `place` candidates built side by side (prl_solver_create_weighted_placed)
                n = int(place)
                ms, chosen = (ctypes.c_float * n)(), ctypes.c_int32(0)
                self._L.prl_solver_create_weighted_placed.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_void_p, ctypes.c_int32,
                                                                      ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32),
                                                                      ctypes.POINTER(ctypes.c_void_p)]
                self._L.prl_solver_create_weighted_placed.restype = ctypes.c_int32
                check(self._L.prl_solver_create_weighted_placed(tree.handle, v, int(delay), 1 if avg_dtype == "f32" else 0, _ptr(m), sym, n, int(probe_iters), ms,
                                                                ctypes.byref(chosen), ctypes.byref(self._h)), self._L)
                self.placement_ms, self.placement_chosen = [float(x) for x in ms], int(chosen.value)
            else:
                check(self._L.prl_solver_create_weighted(tree.handle, v, int(delay), 1 if avg_dtype == "f32" else 0, _ptr(m), sym, ctypes.byref(self._h)), self._L)
        elif not self._h and place is not None:
            # placement selection inside the library (prl_solver_create_placed): `place` candidates built side by side, the fastest kept
            n = int(place)
            ms = (ctypes.c_float * n)()
            chosen = ctypes.c_int32(0)
            self._L.prl_solver_create_placed.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32,
                                                         ctypes.c_int32, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int32),
                                                         ctypes.POINTER(ctypes.c_void_p)]
            self._L.prl_solver_create_placed.restype = ctypes.c_int32
            check(self._L.prl_solver_create_placed(tree.handle, v, int(delay), e, 1 if avg_dtype == "f32" else 0, n, int(probe_iters), ms,
                                                   ctypes.byref(chosen), ctypes.byref(self._h)), self._L)
            self.placement_ms = [float(x) for x in ms]
            self.placement_chosen = int(chosen.value)
        elif not self._h:
            if avg_dtype == "f32":  # opt-in: the running average stored as float32 (prl_solver_create_opts: PRL_SOLVER_AVG_F32); not the reference's numerics
                self._L.prl_solver_create_opts.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, ctypes.POINTER(ctypes.c_void_p)]
                self._L.prl_solver_create_opts.restype = ctypes.c_int32
                check(self._L.prl_solver_create_opts(tree.handle, v, int(delay), e, 1, ctypes.byref(self._h)), self._L)
            else:
                assert avg_dtype == "f64"
                check(self._L.prl_solver_create_ex(tree.handle, v, int(delay), e, ctypes.byref(self._h)), self._L)
        if shard is not None and hasattr(shard[2], "bind_stream"):
            # a stream-ordered exchange (RCCL under the solver's own stream): no host synchronisation per pass
            sp = ctypes.c_void_p()
            check(self._L.prl_solver_get_stream(self._h, ctypes.byref(sp)), self._L)
            if shard[2].bind_stream(sp.value or 0):
                check(self._L.prl_solver_set_exchange_async(self._h, 1), self._L)
        self.n_nodes, self.n_cols, self.R = tree.n_nodes, tree.n_cols, tree.range_size
        eng = np.zeros(1, np.int32)
        self._call("prl_solver_get", SF["engine"], _ptr(eng))
        self.engine = {1: "levels", 2: "fused"}[int(eng[0])]

    def __del__(self):
        try:
            if self._h:
                self._L.prl_solver_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def _call(self, name, *args):
        check(getattr(self._L, name)(self._h, *args), self._L)

    def reset(self):
        self._call("prl_solver_reset")

    def iteration(self):
        self._call("prl_solver_iteration")

    def iterations(self, n):
        self._call("prl_solver_iterations", int(n))

    def fill_uniform(self):
        self._call("prl_solver_fill_uniform")

    def set_strategy(self, strategy_cols):
        a = np.ascontiguousarray(strategy_cols)
        if a.dtype not in (np.float32, np.float64):
            a = a.astype(np.float32)
        assert a.shape == (self.n_cols, self.R), (a.shape, (self.n_cols, self.R))
        self._call("prl_solver_set_strategy", _ptr(a), int(a.dtype == np.float64))

    def set_strategy_device(self, device_ptr, n_actions):
        """float32 [n_decision_nodes, R, n_actions] ALREADY IN HBM (e.g. a torch tensor's data_ptr(): a batched network forward's output; decision nodes
        in node order) -> the solver's strategy, scattered on the GPU (prl_solver_set_strategy_device): no host copy. The caller synchronises the
        producer first (torch.cuda.synchronize())."""
        self._L.prl_solver_set_strategy_device.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]
        self._L.prl_solver_set_strategy_device.restype = ctypes.c_int32
        self._call("prl_solver_set_strategy_device", self.tree.handle, ctypes.c_void_p(int(device_ptr)), int(n_actions))

    def set_strategy_mixed(self, strategy_cols_f64, node_is_f64):
        """float64 [n_cols, R] + uint8 [n_nodes]: 1 where the node's strategy is float64 in the reference's sense (else the stored
        values are float32-representable and the node's arithmetic is float32)"""
        a = np.ascontiguousarray(strategy_cols_f64, dtype=np.float64)
        f = np.ascontiguousarray(node_is_f64, dtype=np.uint8)
        assert a.shape == (self.n_cols, self.R)
        self._L.prl_solver_set_strategy_mixed.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p]
        self._L.prl_solver_set_strategy_mixed.restype = ctypes.c_int32
        self._call("prl_solver_set_strategy_mixed", _ptr(a), 1, _ptr(f))

    def update_reach(self):
        self._call("prl_solver_update_reach")

    def compute_ev(self):
        self._call("prl_solver_compute_ev")

    def sync(self):
        self._call("prl_solver_sync")

    def time_iterations(self, n):
        """n iterations bracketed by HIP events on the solver's stream -> elapsed device milliseconds."""
        ms = ctypes.c_float()
        self._call("prl_solver_time_iterations", int(n), ctypes.byref(ms))
        return float(ms.value)

    def save_state(self):
        """The solver's persistent state as one uint8 array (checkpoint): np.save it, load_state() it into a solver built on
        the same tree / variant / delay / engine and the run continues bit-identically."""
        n = ctypes.c_uint64()
        self._call("prl_solver_state_size", ctypes.byref(n))
        blob = np.zeros(int(n.value), np.uint8)
        self._call("prl_solver_save_state", _ptr(blob), int(n.value))
        return blob

    def load_state(self, blob):
        blob = np.ascontiguousarray(blob, dtype=np.uint8)
        self._call("prl_solver_load_state", _ptr(blob), int(blob.shape[0]))

    @property
    def graph_replay(self):
        """True once the LEVELS engine replays a captured hipGraph per iteration (launch-bound small trees)."""
        v = np.zeros(1, np.int32)
        self._call("prl_solver_get", SF["graph_replay"], _ptr(v))
        return bool(v[0])

    def time_iterations_ex(self, n):
        """-> (total device ms, summed ms of the board-pass kernel launches, number of those launches)."""
        ms, pms, cnt = ctypes.c_float(), ctypes.c_float(), ctypes.c_int32()
        self._call("prl_solver_time_iterations_ex", int(n), ctypes.byref(ms), ctypes.byref(pms), ctypes.byref(cnt))
        return float(ms.value), float(pms.value), int(cnt.value)

    def time_evaluations(self, n):
        """n x (update_reach + compute_ev) of the loaded strategy -> (total device ms, summed board-pass kernel ms, launches)."""
        ms, pms, cnt = ctypes.c_float(), ctypes.c_float(), ctypes.c_int32()
        self._call("prl_solver_time_evaluations", int(n), ctypes.byref(ms), ctypes.byref(pms), ctypes.byref(cnt))
        return float(ms.value), float(pms.value), int(cnt.value)

    @staticmethod
    def iterations_many(solvers, n):
        """n iterations of many independent small-tree solvers in ONE launch, one workgroup per solver (prl_solver_iterations_many)"""
        if not solvers:
            return
        L = solvers[0]._L
        arr = (ctypes.c_void_p * len(solvers))(*[s._h for s in solvers])
        check(L.prl_solver_iterations_many(arr, len(solvers), int(n)), L)

    def exploitability(self):
        out = np.zeros(2, np.float32)
        self._call("prl_solver_exploitability", _ptr(out))
        return out

    def eval_avg(self):
        out = np.zeros(2, np.float32)
        self._call("prl_solver_eval_avg", _ptr(out))
        return out

    @property
    def iter(self):
        out = np.zeros(1, np.int32)
        self._call("prl_solver_get", SF["iter"], _ptr(out))
        return int(out[0])

    def get_cols(self, name, col_begin, n_cols):
        """columns [col_begin, col_begin + n_cols) of "regret" / "avg" / "avg_sum": streams a big array through a small host buffer"""
        out = np.zeros((int(n_cols), self.R), np.float64 if name == "avg" else np.float32)
        self._L.prl_solver_get_cols.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p]
        self._L.prl_solver_get_cols.restype = ctypes.c_int32
        self._call("prl_solver_get_cols", SF[name], int(col_begin), int(n_cols), _ptr(out))
        return out

    def sha256_of(self, name, cols_per_piece=65536):
        """SHA-256 of a per-action-column array in the flat tree's column order (-0.0 folded into +0.0 like tests/helpers.h32), streamed"""
        import hashlib
        h = hashlib.sha256()
        for c0 in range(0, self.n_cols, cols_per_piece):
            a = self.get_cols(name, c0, min(cols_per_piece, self.n_cols - c0))
            h.update(np.ascontiguousarray(a + a.dtype.type(0)).tobytes())
        return h.hexdigest()

    def sha256_of_many(self, names, cols_per_piece=32768):
        """{name: SHA-256} of several per-action-column arrays at once: the pieces are fetched in turn (one staging buffer, one stream) while one hasher
        thread per array digests them -- a 60-80 GB state is hashed in the time of its biggest array instead of the sum (hashlib releases the GIL)"""
        import hashlib
        import queue
        import threading
        hs = {n: hashlib.sha256() for n in names}
        qs = {n: queue.Queue(maxsize=3) for n in names}

        def work(n):
            while True:
                a = qs[n].get()
                if a is None:
                    return
                hs[n].update(np.ascontiguousarray(a + a.dtype.type(0)).tobytes())

        th = [threading.Thread(target=work, args=(n,), daemon=True) for n in names]
        for t in th:
            t.start()
        try:
            for c0 in range(0, self.n_cols, cols_per_piece):
                for n in names:
                    qs[n].put(self.get_cols(n, c0, min(cols_per_piece, self.n_cols - c0)))
        finally:
            for n in names:
                qs[n].put(None)
            for t in th:
                t.join()
        return {n: h.hexdigest() for n, h in hs.items()}

    def get(self, name):
        n, c, R = self.n_nodes, self.n_cols, self.R
        shape, dtype = {
            "reach": ((n, 2, R), np.float32), "ev": ((n, 2, R), np.float32), "ev_br": ((n, 2, R), np.float32),
            "strategy": ((c, R), np.float64), "strat_f64": ((n,), np.uint8), "regret": ((c, R), np.float32),
            "avg": ((c, R), np.float64), "avg_f64": ((n,), np.uint8), "avg_sum": ((c, R), np.float32),
            "br_idx": ((n, R), np.int32), "constants": ((2,), np.float32), "bytes_allocated": ((1,), np.int64),
            "explicit_strategy": ((1,), np.int32), "exchanges": ((1,), np.int64), "vmm_ranges": ((2,), np.int64),
        }.get(name, (None, None))
        if name == "expl_history":
            shape, dtype = (self.iter + 1, 2), np.float32
        out = np.zeros(shape, dtype)
        self._call("prl_solver_get", SF[name], _ptr(out))
        return out
