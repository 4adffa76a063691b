"""PokerRL/eval/head_to_head/H2HArgs.py: number of hands per seat assignment."""


class H2HArgs:
    def __init__(self, n_hands):
        self.n_hands = n_hands
