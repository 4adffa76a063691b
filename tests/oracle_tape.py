"""Oracle tapes (TEST INFRASTRUCTURE): what the CPU oracle says in a parity check, recorded once and replayed on the GPU box.

The multi-street parity checks of the GPU suite ran the oracle LIVE on the GPU box's host cores -- half a minute to a minute each, a quarter of the
driver's time limit between them (round 5's verdict, "weak" 9). The oracle's side of such a check does not depend on the GPU: it is a fixed sequence of
values (exploitabilities, array digests) for a fixed problem. A check asks for them through `tape.take(tag, fn)`; with a tape file present (tests/golden/
oracle_tapes/, written by tests/golden/make_oracle_tapes.py = the same check functions run in RECORD mode, oracle only, no GPU) the values come from the
file, in order, each under its tag; without one the oracle runs live exactly as before. Big arrays travel as SHA-256 digests (-0.0 folded into +0.0 like
helpers.h32), small ones verbatim. PRL_ORACLE_TAPE=live ignores the files (full arrays are then compared entry by entry: the diagnostic mode),
PRL_ORACLE_TAPE=record writes them.
"""
import hashlib
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TAPE_DIR = os.path.join(HERE, "golden", "oracle_tapes")


def digest(a):
    a = np.ascontiguousarray(a)
    return hashlib.sha256((a + a.dtype.type(0)).tobytes()).hexdigest()


def _key(parts):
    h = hashlib.sha1()
    for p in parts:
        if isinstance(p, np.ndarray):
            h.update(("nd:%s:%s:" % (p.dtype, p.shape)).encode())
            h.update(np.ascontiguousarray(p).tobytes())
        else:
            h.update(repr(p).encode())
        h.update(b"|")
    return h.hexdigest()[:16]


class Tape:
    def __init__(self, kind, *parts):
        self.mode = os.environ.get("PRL_ORACLE_TAPE", "auto")
        self.name = "%s_%s.npz" % (kind, _key(parts))
        self.path = os.path.join(TAPE_DIR, self.name)
        self._items, self._pos, self._rec = None, 0, None
        if self.mode == "record":
            self._rec = []
        elif self.mode != "live" and os.path.isfile(self.path):
            with np.load(self.path, allow_pickle=False) as z:
                tags = [str(t) for t in z["tags"]]
                self._items = [(t, z["v%d" % i]) for i, t in enumerate(tags)]

    @property
    def live(self):
        """the oracle has to run (no tape, or recording one)"""
        return self._items is None

    @property
    def recording(self):
        return self._rec is not None

    def step(self, fn):
        if self.live:
            fn()

    def take(self, tag, fn):
        if self.live:
            v = fn()
            if self._rec is not None:
                self._rec.append((tag, np.asarray(v)))
            return v
        t, v = self._items[self._pos]
        assert t == tag, "oracle tape %s: expected %r at position %d, found %r -- the check changed: regenerate (tests/golden/make_oracle_tapes.py)" % (self.name, tag, self._pos, t)
        self._pos += 1
        return v if v.dtype.kind not in "US" else str(v)

    def close(self):
        if self._rec is not None:
            os.makedirs(TAPE_DIR, exist_ok=True)
            np.savez_compressed(self.path, tags=np.array([t for t, _ in self._rec]), **{"v%d" % i: v for i, (_, v) in enumerate(self._rec)})
        elif not self.live:
            assert self._pos == len(self._items), "oracle tape %s: %d of %d entries used -- the check changed: regenerate" % (self.name, self._pos, len(self._items))


def same(got, want, what):
    """got: an array of the device solver; want: the oracle's array (live) or its digest (tape)"""
    if isinstance(want, str):
        assert digest(got) == want, "%s differs from the oracle's tape (PRL_ORACLE_TAPE=live compares entry by entry)" % what
    else:
        want = np.asarray(want)
        assert np.array_equal(got, want), "%s differs in %d entries, first %s" % (what, int(np.sum(got != want)), np.argwhere(got != want)[:3].tolist())
