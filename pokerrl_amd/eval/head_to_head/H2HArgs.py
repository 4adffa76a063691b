"""Arguments of the head-to-head evaluator, read from ``t_prof.module_args["h2h"]`` (the reference's H2HArgs has this one field)."""


class H2HArgs:
    __slots__ = ("n_hands",)

    def __init__(self, n_hands):
        n = int(n_hands)
        if n <= 0:
            raise ValueError("n_hands: hands per seat assignment, > 0")
        self.n_hands = n
