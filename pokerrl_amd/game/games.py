"""
The poker games of the reference (PokerRL/game/games.py:18-269), as configuration objects for the native engine.

Each class keeps the reference's class attributes (blinds, bet sizes, raise caps, EV_NORMALIZER, WIN_METRIC, RULES,
ARGS_CLS, DEFAULT_STACK_SIZE ...) because CFR / BR / user scripts read them (`game_cls.DEFAULT_STACK_SIZE`,
`env_cls.EV_NORMALIZER`, `_CFRBase.py:45-56,201-203`). The betting logic itself lives in csrc/prl_env.h; a game class
only knows how to describe itself to it (`native_game`).
"""
from pokerrl_amd import _native
from pokerrl_amd.game.Poker import Poker
from pokerrl_amd.game.game_rules import BigLeducRules, FlopHoldemRules, HoldemRules, LeducRules
from pokerrl_amd.game.poker_env_args import DiscretizedPokerEnvArgs, LimitPokerEnvArgs, NoLimitPokerEnvArgs

GAME_LIMIT, GAME_DISCRETIZED, GAME_NOLIMIT = 0, 1, 2


class _GameBase:
    RULES = None
    ARGS_CLS = None
    _GAME_TYPE = None
    IS_FIXED_LIMIT_GAME = False
    IS_POT_LIMIT_GAME = False
    SMALL_BLIND = BIG_BLIND = ANTE = 0
    SMALL_BET = BIG_BET = 0
    MAX_N_RAISES_PER_ROUND = {}
    ROUND_WHERE_BIG_BET_STARTS = 0
    FIRST_ACTION_NO_CALL = False
    _POT_SIZE_RAISE = False
    DEFAULT_STACK_SIZE = None
    EV_NORMALIZER = None
    WIN_METRIC = None

    def __new__(cls, env_args, lut_holder=None, is_evaluating=True):
        """`game_cls(env_args=..., lut_holder=..., is_evaluating=...)` is how the reference builds an env (a game class IS a PokerEnv
        subclass there, PokerEnv.py:53): here it returns the native-backed PokerEnv facade of that game."""
        from pokerrl_amd.game.poker_env import PokerEnv
        return PokerEnv(env_cls=cls, env_args=env_args, lut_holder=lut_holder if lut_holder is not None else cls.get_lut_holder(),
                        is_evaluating=is_evaluating)

    @classmethod
    def get_lut_holder(cls):
        return cls.RULES.get_lut_holder()

    @classmethod
    def native_rules(cls):
        return cls.RULES.to_native()

    @classmethod
    def native_game(cls, env_args):
        """env args (+ class constants) -> PrlGame POD of the C ABI."""
        if env_args.n_seats != 2:
            raise NotImplementedError("the MI355X hot path is heads-up only, like the reference's CFR / BR / LBR "
                                      "(PokerRL/cfr/_CFRBase.py:40, eval/br/LocalBRMaster.py:23)")
        g = _native.PrlGame()
        g.game_type = cls._GAME_TYPE
        g.n_rounds = len(cls.RULES.ALL_ROUNDS_LIST)
        g.small_blind, g.big_blind, g.ante = cls.SMALL_BLIND, cls.BIG_BLIND, cls.ANTE
        g.small_bet, g.big_bet = cls.SMALL_BET, cls.BIG_BET
        g.round_big_bet_starts = cls.ROUND_WHERE_BIG_BET_STARTS
        for r in range(4):
            g.max_raises[r] = cls.MAX_N_RAISES_PER_ROUND.get(r, 0)
        g.first_action_no_call = int(cls.FIRST_ACTION_NO_CALL)
        g.btn_first_postflop = int(cls.RULES.BTN_IS_FIRST_POSTFLOP)
        g.pot_size_raise = int(cls._POT_SIZE_RAISE)
        stacks = [cls.DEFAULT_STACK_SIZE if s is None else int(s) for s in env_args.starting_stack_sizes_list]
        g.start_stack[0], g.start_stack[1] = stacks
        if cls._GAME_TYPE == GAME_DISCRETIZED:
            fracs = sorted(env_args.bet_sizes_list_as_frac_of_pot)
            if len(fracs) > _native.PRL_MAX_BET_SIZES:
                raise ValueError("at most %d bet sizes" % _native.PRL_MAX_BET_SIZES)
            g.n_bet_sizes = len(fracs)
            for i, f in enumerate(fracs):
                g.bet_fracs[i] = float(f)
        return g


# ---- Leduc family -----------------------------------------------------------------------------------------------------
class _LimitLeduc(_GameBase):
    ARGS_CLS = LimitPokerEnvArgs
    _GAME_TYPE = GAME_LIMIT
    IS_FIXED_LIMIT_GAME = True
    ANTE = 1
    SMALL_BET, BIG_BET = 2, 4
    ROUND_WHERE_BIG_BET_STARTS = Poker.FLOP
    EV_NORMALIZER = 1000.0 / ANTE  # milli antes
    WIN_METRIC = Poker.MeasureAnte


class StandardLeduc(LeducRules, _LimitLeduc):
    RULES = LeducRules
    MAX_N_RAISES_PER_ROUND = {Poker.PREFLOP: 2, Poker.FLOP: 2}
    DEFAULT_STACK_SIZE = 13


class BigLeduc(BigLeducRules, _LimitLeduc):
    RULES = BigLeducRules
    MAX_N_RAISES_PER_ROUND = {Poker.PREFLOP: 6, Poker.FLOP: 6}
    DEFAULT_STACK_SIZE = 100


class _BlindsGame(_GameBase):
    SMALL_BLIND, BIG_BLIND, ANTE = 50, 100, 0
    DEFAULT_STACK_SIZE = 20000
    EV_NORMALIZER = 1000.0 / BIG_BLIND  # milli big blinds
    WIN_METRIC = Poker.MeasureBB


class NoLimitLeduc(LeducRules, _BlindsGame):
    RULES = LeducRules
    ARGS_CLS = NoLimitPokerEnvArgs
    _GAME_TYPE = GAME_NOLIMIT


class DiscretizedNLLeduc(LeducRules, _BlindsGame):
    RULES = LeducRules
    ARGS_CLS = DiscretizedPokerEnvArgs
    _GAME_TYPE = GAME_DISCRETIZED


# ---- Hold'em family ---------------------------------------------------------------------------------------------------
class LimitHoldem(HoldemRules, _GameBase):
    RULES = HoldemRules
    ARGS_CLS = LimitPokerEnvArgs
    _GAME_TYPE = GAME_LIMIT
    IS_FIXED_LIMIT_GAME = True
    MAX_N_RAISES_PER_ROUND = {Poker.PREFLOP: 4, Poker.FLOP: 4, Poker.TURN: 4, Poker.RIVER: 4}
    ROUND_WHERE_BIG_BET_STARTS = Poker.TURN
    SMALL_BLIND, BIG_BLIND, ANTE = 1, 2, 0
    SMALL_BET, BIG_BET = 2, 4
    DEFAULT_STACK_SIZE = 48
    EV_NORMALIZER = 1000.0 / BIG_BLIND
    WIN_METRIC = Poker.MeasureBB


class NoLimitHoldem(HoldemRules, _BlindsGame):
    RULES = HoldemRules
    ARGS_CLS = NoLimitPokerEnvArgs
    _GAME_TYPE = GAME_NOLIMIT


class DiscretizedNLHoldem(HoldemRules, _BlindsGame):
    RULES = HoldemRules
    ARGS_CLS = DiscretizedPokerEnvArgs
    _GAME_TYPE = GAME_DISCRETIZED


class Flop5Holdem(FlopHoldemRules, _BlindsGame):
    """2 rounds, 5 board cards at once, pot-size raises through the fixed-limit env (games.py:222-254)."""
    RULES = FlopHoldemRules
    ARGS_CLS = LimitPokerEnvArgs
    _GAME_TYPE = GAME_LIMIT
    IS_FIXED_LIMIT_GAME = True
    MAX_N_RAISES_PER_ROUND = {Poker.PREFLOP: 2, Poker.FLOP: 2}  # the big blind counts as a raise pre-flop
    ROUND_WHERE_BIG_BET_STARTS = Poker.TURN
    UNITS_SMALL_BET = None
    UNITS_BIG_BET = None
    FIRST_ACTION_NO_CALL = True
    _POT_SIZE_RAISE = True


ALL_ENVS = [StandardLeduc, BigLeduc, NoLimitLeduc, DiscretizedNLLeduc, LimitHoldem, NoLimitHoldem, DiscretizedNLHoldem,
            Flop5Holdem]
