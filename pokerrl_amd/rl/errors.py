"""Exception names downstream EvalAgent implementations import from PokerRL.rl.errors."""


class UnknownModeError(ValueError):
    """raised by an agent asked for a mode it does not have (PokerRL/rl/errors.py:4-7: prints the offending mode)"""

    def __init__(self, var):
        super().__init__("Mode %r is unknown" % (var,))
        print("Mode", var, "is unknown")
