"""Kept so that `import synth` in the tests / tools keeps working: the helpers live in benchdata/synth.py."""
from benchdata.synth import *  # noqa: F401,F403
from benchdata.synth import _rng  # noqa: F401
