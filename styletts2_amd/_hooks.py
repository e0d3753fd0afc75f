"""Test-only selectors.  The product has ONE implementation of every stage -- the C++ launch plans of csrc/st2_engine.hip
over the HIP kernels -- and reads no environment variable to pick another.  What the parity net needs on top of that is
reachable from tests/ and tools/ only, through `override(...)`:

  plan            "engine" (always, outside tests): a module forward is ONE C-ABI call into its C++ launch plan.
                  "python": the per-kernel Python plans (decoder.py / diffusion.py / text.py / style.py) -- the same kernels
                  with the same arguments issued from Python.  They exist because tap points and the CPU plan tests (which
                  substitute per-kernel CPU contracts for the HIP wrappers) need a plan that can be stepped through; the
                  C++ plans are held bitwise / to 1e-5 against them and both are held to the oracle.
  conv_path       "xs" (st2_act_split + st2_conv1d_xs, the library's routing) or "fused" (st2_conv1d_f16s everywhere):
                  lets the contract tests drive each conv kernel on every shape.
  conv_precision  "f16s" (split-f16 MFMA) or "f32" (the exact-fp32 MFMA build of the same contract: the tests' reference
                  kernel, and what the mel front-end's DFT uses explicitly).
  lstm            "coop" (cooperative BiLSTM, with the library's own refusal -> single-CU path) or "single".
  lstm_recover    True (always, outside tests): cooperative launches carry their in-stream safety net
                  (st2_lstm_bidir_coop_recovering); False = the bare st2_lstm_bidir_coop, whose time-out is only reported.
"""
import contextlib

plan = "engine"
conv_path = "xs"
conv_precision = "f16s"
lstm = "coop"
lstm_recover = True

_CHOICES = {"plan": ("engine", "python"), "conv_path": ("xs", "fused"), "conv_precision": ("f16s", "f32"),
            "lstm": ("coop", "single"), "lstm_recover": (True, False)}


@contextlib.contextmanager
def override(**kw):
    """with _hooks.override(plan="python"): ...   -- tests / tools only."""
    g = globals()
    old = {}
    for k, v in kw.items():
        if k not in _CHOICES or v not in _CHOICES[k]:
            raise ValueError("unknown selector %s=%r" % (k, v))
        old[k] = g[k]
        g[k] = v
    try:
        yield
    finally:
        g.update(old)
