"""Import harness for the upstream Python reference (build container only).

The reference (`/root/reference`, mp3stego-lib 1.1.8) needs `numba` and
`bitarray`, neither of which is installed here and neither of which can be
installed (no network).  Both are used trivially:

* `numba.njit` is only a decorator (`@njit`, `@njit(fastmath=True)`), so an
  identity decorator executes the very same Python source under CPython;
* `bitarray.bitarray().frombytes()` + iteration is used once
  (reference `mp3stego/steganography.py:20-22`).

This module writes those two shims into a temp dir, puts them and the
reference on `sys.path`, and returns the imported `mp3stego` package.  It is
tooling for `gen_golden.py` only: nothing here ships to the GPU box, the
`-m gpu` tests, `bench.py` or `smoke()` never import it.
"""
import os
import sys
import tempfile

REFERENCE_ROOT = os.environ.get("MP3S_REFERENCE_ROOT", "/root/reference")

_NUMBA_SHIM = '''
def njit(*args, **kwargs):
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    def deco(f):
        return f
    return deco
'''

_BITARRAY_SHIM = '''
class bitarray(list):
    def frombytes(self, b):
        for x in b:
            for n in range(7, -1, -1):
                self.append((x >> n) & 1)
'''

_loaded = None


def load_reference():
    """Return the reference's `mp3stego` package (imported under the shims)."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not os.path.isdir(REFERENCE_ROOT):
        raise RuntimeError(f"reference not present at {REFERENCE_ROOT}")
    d = tempfile.mkdtemp(prefix="mp3s_refshim_")
    os.makedirs(os.path.join(d, "numba"))
    with open(os.path.join(d, "numba", "__init__.py"), "w") as f:
        f.write(_NUMBA_SHIM)
    with open(os.path.join(d, "bitarray.py"), "w") as f:
        f.write(_BITARRAY_SHIM)
    # the reference must win over any same-named package (our drop-in is also
    # called `mp3stego`), so it goes to the very front
    sys.path[:0] = [d, REFERENCE_ROOT]
    for k in [k for k in sys.modules if k == "mp3stego" or k.startswith("mp3stego.")]:
        del sys.modules[k]
    import warnings
    warnings.filterwarnings("ignore")
    import mp3stego  # noqa
    _loaded = mp3stego
    return mp3stego
