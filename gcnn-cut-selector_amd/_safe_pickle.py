"""Restricted unpickling for the two pickle formats of the reference that this package reads from disk:

  * checkpoints: 62 consecutive `pickle.dump(np.ndarray)` records (/root/reference/model.py:53-56, 64-67);
  * samples: gzip(pickle({'data': [state, improvements]})) with `state` a tuple of dicts of str lists and ndarrays
    (/root/reference/data_collector.py:135-140).

Both only ever hold ndarrays inside builtin containers, so `find_class` admits exactly the globals NumPy's array / dtype /
scalar reductions name and nothing else: a crafted file cannot import or call anything.  Builtin containers, str, int and
float are pickle opcodes and never reach `find_class`."""

from __future__ import annotations

import pickle

_ALLOWED = {
    ("numpy", "ndarray"), ("numpy", "dtype"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
    ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer"),
    ("_codecs", "encode"),   # protocol <= 2 spells a bytes payload as _codecs.encode(str, "latin1"): a pure str -> bytes map
}


class _NumpyOnlyUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _ALLOWED:
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f"refusing to unpickle global {module}.{name}: only NumPy arrays in builtin "
                                     "containers are allowed in checkpoint / sample files")


def load(file):
    """`pickle.load(file)` restricted to NumPy arrays, dtypes and scalars inside builtin containers."""
    return _NumpyOnlyUnpickler(file).load()
