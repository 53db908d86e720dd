"""Import the UNMODIFIED reference (/root/reference) in this container.

Test infrastructure only: used by tests/golden/gen_golden.py to produce the
committed golden fixtures.  /root/reference does not exist on the GPU box, so
nothing under `-m gpu`, smoke() or bench.py may import this module.

Stubs needed:
  * `tensorflow`  - main.py:8 imports it at module scope; only net ctors touch it.
  * `uvloop`      - main.py:5-6 installs its event-loop policy.
  * asyncio.Semaphore - `with await self.sem` (main.py:342) is a TypeError on
    Python >= 3.9 (quirk Q9); replaced by an awaitable subclass, the reference
    source itself is untouched.
"""
import asyncio
import contextlib
import io
import os
import sys
import types

REF = os.environ.get("CCHESS_REFERENCE", "/root/reference")


def available():
    return os.path.isfile(os.path.join(REF, "main.py"))


class _AwaitableSemaphore(asyncio.Semaphore):
    """`with await sem:` support (pre-3.9 behaviour the reference relies on)."""

    def __await__(self):
        yield from self.acquire().__await__()
        return _Releaser(self)


class _Releaser:
    def __init__(self, sem):
        self._sem = sem

    def __enter__(self):
        return None

    def __exit__(self, *exc):
        self._sem.release()
        return False


_main = None


def load_main():
    """Returns the reference `main` module (cached)."""
    global _main
    if _main is not None:
        return _main
    if not available():
        raise RuntimeError("reference not present at %s" % REF)
    if "tensorflow" not in sys.modules:
        sys.modules["tensorflow"] = types.ModuleType("tensorflow")
    if "uvloop" not in sys.modules:
        uv = types.ModuleType("uvloop")
        uv.EventLoopPolicy = asyncio.DefaultEventLoopPolicy
        sys.modules["uvloop"] = uv
    if REF not in sys.path:
        sys.path.insert(0, REF)
    try:
        asyncio.get_event_loop()
    except RuntimeError:
        asyncio.set_event_loop(asyncio.new_event_loop())
    # our repo also has a main.py; make sure the reference one wins here.
    saved = sys.modules.pop("main", None)
    import importlib.util
    spec = importlib.util.spec_from_file_location("cchess_ref_main", os.path.join(REF, "main.py"))
    mod = importlib.util.module_from_spec(spec)
    # the reference does `from policy_value_network import *` -> resolve to the
    # reference's own files, not this repo's drop-in modules.
    saved_pv = {k: sys.modules.pop(k, None) for k in ("policy_value_network", "policy_value_network_gpus")}
    try:
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved_pv.items():
            sys.modules.pop(k, None)
            if v is not None:
                sys.modules[k] = v
        if saved is not None:
            sys.modules["main"] = saved
    mod.asyncio.Semaphore = _AwaitableSemaphore
    _main = mod
    return mod


def new_mcts(state, forward, search_threads=1):
    m = load_main()
    asyncio.set_event_loop(asyncio.new_event_loop())
    return m.MCTS_tree(state, forward, search_threads)


@contextlib.contextmanager
def quiet():
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        yield buf


def load_chessboard():
    """Second oracle: ChessBoard.py + chessman/* (GUI rules). Returns module."""
    if REF not in sys.path:
        sys.path.insert(0, REF)
    import importlib
    cb = importlib.import_module("ChessBoard")
    return cb
