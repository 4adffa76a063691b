"""Global poker constants; same names and values as the reference (PokerRL/game/Poker.py:7-46)."""
import numpy as np


class Poker:
    PREFLOP, FLOP, TURN, RIVER = 0, 1, 2, 3
    INT2STRING_ROUND = {0: "preflop", 1: "flop", 2: "turn", 3: "river"}
    STRING2INT_ROUND = {v: k for k, v in INT2STRING_ROUND.items()}

    FOLD, CHECK_CALL, BET_RAISE = 0, 1, 2

    CARD_NOT_DEALT_TOKEN_1D = -127
    CARD_NOT_DEALT_TOKEN_2D = np.array([-127, -127])

    MeasureAnte = "MA_per_G"   # milli antes per game
    MeasureBB = "MBB_per_G"    # milli big blinds per game
