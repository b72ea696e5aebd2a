"""Sample files and batch collation: the counterpart of the reference's `utils.load_batch` (/root/reference/utils.py:339-426)
and of the sample format `data_collector.py` writes (data_collector.py:135-140):

    gzip(pickle({'data': [(cons, cons_edge, var, cut, cut_edge), improvements]}))

where cons/var/cut are {'features': [...], 'values': ndarray} and the edge dicts add 'indices' ([2,E], row 0 = left id).
`load_batch` returns the same 11-tuple as the reference (NumPy arrays instead of tf.Tensors): stacked features, edge
indices shifted per sample (disjoint-union batching), per-sample count vectors and the stacked improvements."""

from __future__ import annotations

import gzip
import pickle

import numpy as np

from . import _safe_pickle


def save_sample(path: str, state, improvements):
    """Write one (state, improvements) pair in the reference's on-disk format (data_collector.py:135-140)."""
    with gzip.open(path, "wb") as file:
        pickle.dump({"data": [state, np.asarray(improvements)]}, file)


def load_sample(path: str):
    """One sample file -> [state, improvements]; unpickled with NumPy arrays / builtin containers admitted only."""
    with gzip.open(path, "rb") as file:
        sample = _safe_pickle.load(file)
    return sample["data"]


def decode_files(files, workers=8, prefetch=4):
    """Yield `load_sample(f)` for every file, in order, with the gunzip work (~90 % of a load) spread over `workers` child
    processes.  The children are fresh interpreters running `_decode_worker.py` by path -- standard library only: no torch, no
    re-import of the caller's `__main__`, no `if __name__ == "__main__"` guard needed in the calling script -- fed file names over a
    pipe and answering with the decompressed bytes, which are unpickled HERE with the NumPy-only unpickler.  They end on end of
    input with exit code 0 (no signal is ever sent to a child that is still running normally).  `workers=0`: decode in-process."""
    files = [str(f) for f in files]
    if not workers or len(files) < 2:
        for f in files:
            yield load_sample(f)
        return
    import io
    import os
    import struct
    import subprocess
    import sys
    n = min(int(workers), len(files))
    script = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_decode_worker.py")
    procs = [subprocess.Popen([sys.executable, "-S", "-E", script], stdin=subprocess.PIPE, stdout=subprocess.PIPE) for _ in range(n)]
    try:   # a megabyte of pipe per child (Linux; the default is 64 KiB): a child can finish a sample while the parent reads another's
        import fcntl
        for p in procs:
            fcntl.fcntl(p.stdout.fileno(), getattr(fcntl, "F_SETPIPE_SZ", 1031), 1 << 20)
    except (ImportError, OSError):
        pass
    sent = [0] * n    # files handed to each child so far (child w serves files w, w + n, w + 2n, ...)

    def feed(w):       # a few names ahead of what has been read back: the children never starve, the name pipe never fills
        i = w + sent[w] * n
        if i < len(files):
            procs[w].stdin.write(files[i].encode("utf-8", "surrogateescape") + b"\n")
            procs[w].stdin.flush()
            sent[w] += 1
        elif procs[w].stdin and not procs[w].stdin.closed:
            procs[w].stdin.close()     # end of input: the child exits by itself

    def read_exact(pipe, k):
        buf = pipe.read(k)
        if len(buf) != k:
            raise RuntimeError("sample decoder child ended early")
        return buf

    ok = False
    try:
        for w in range(n):
            for _ in range(prefetch):
                feed(w)
        for i, f in enumerate(files):
            w = i % n
            out = procs[w].stdout
            (size,) = struct.unpack("<Q", read_exact(out, 8))
            if size == 2 ** 64 - 1:
                (k,) = struct.unpack("<Q", read_exact(out, 8))
                raise OSError(f"{f}: {read_exact(out, k).decode('utf-8', 'replace')}")
            sample = _safe_pickle.load(io.BytesIO(read_exact(out, size)))
            feed(w)
            yield sample["data"]
        ok = True
    finally:
        for p in procs:                # normal end: stdin is closed already and the child is on its way out
            if p.stdin and not p.stdin.closed:
                p.stdin.close()
            p.stdout.close()           # abnormal end (an exception above, an abandoned generator): the child sees a closed pipe
        for p in procs:
            try:
                p.wait(timeout=30 if ok else 5)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()


def collate(samples):
    """The array half of `utils.load_batch` (utils.py:389-426) on already loaded (state, improvements) pairs."""
    cons = [s[0][0]["values"] for s in samples]
    var = [s[0][2]["values"] for s in samples]
    cut = [s[0][3]["values"] for s in samples]
    n_cons = [c.shape[0] for c in cons]
    n_vars = [v.shape[0] for v in var]
    n_cuts = [k.shape[0] for k in cut]
    # index offset of every sample inside the disjoint union: exclusive prefix sums of the row / column counts, one
    # column per sample (the formula of utils.py:401-407 is the batching contract)
    cons_shift = np.cumsum([[0] + n_cons[:-1], [0] + n_vars[:-1]], axis=1)
    cut_shift = np.cumsum([[0] + n_cuts[:-1], [0] + n_vars[:-1]], axis=1)
    cei = np.concatenate([s[0][1]["indices"] + cons_shift[:, j:j + 1] for j, s in enumerate(samples)], axis=1)
    kei = np.concatenate([s[0][4]["indices"] + cut_shift[:, j:j + 1] for j, s in enumerate(samples)], axis=1)
    f32, i32 = np.float32, np.int32
    return (np.concatenate(cons, 0).astype(f32), cei.astype(i32),
            np.concatenate([s[0][1]["values"] for s in samples], 0).astype(f32),
            np.concatenate(var, 0).astype(f32), np.concatenate(cut, 0).astype(f32), kei.astype(i32),
            np.concatenate([s[0][4]["values"] for s in samples], 0).astype(f32),
            np.asarray(n_cons, i32), np.asarray(n_vars, i32), np.asarray(n_cuts, i32),
            np.concatenate([np.asarray(s[1]) for s in samples]).astype(f32))


def load_batch(sample_files):
    """`utils.load_batch` (utils.py:339-426): gunzip + unpickle each file, then collate."""
    return collate([load_sample(f) for f in sample_files])


def state_to_inputs(state):
    """A single `get_state` 5-tuple (utils.py:35-238) -> the model's 10-tuple, as the SCIP plugins build it
    (model_evaluator.py:84-101)."""
    cons, cons_edge, var, cut, cut_edge = state
    f32, i32 = np.float32, np.int32
    return (np.asarray(cons["values"], f32), np.asarray(cons_edge["indices"], i32), np.asarray(cons_edge["values"], f32),
            np.asarray(var["values"], f32), np.asarray(cut["values"], f32), np.asarray(cut_edge["indices"], i32),
            np.asarray(cut_edge["values"], f32), cons["values"].shape[0], var["values"].shape[0], cut["values"].shape[0])
