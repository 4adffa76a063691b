"""
pokerrl_amd -- MI355X-native (gfx950, hand-written HIP behind a C ABI) implementation of PokerRL's tabular hot path:
public-tree CFR / CFR+ / Linear CFR, exact best response, range-vs-range terminal equity, the 7-card hand evaluator and
the card / hand index LUTs, behind PokerRL's own plugin surface (same class and method names as PokerRL.cfr,
PokerRL.eval.br, PokerRL.game). See DESIGN.md and INTEGRATION.md.
"""
__version__ = "0.1.0"
