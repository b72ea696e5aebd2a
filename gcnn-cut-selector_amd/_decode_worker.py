"""Decoder child of `utils.decode_files`: gunzips sample files for the parent, which then unpickles them restrictively.

Run BY PATH as a fresh interpreter (`python -S -E .../_decode_worker.py`), never imported through the package and never a
multiprocessing worker: it must not pull in torch or the caller's `__main__` (a spawn pool re-imports both, and such a child is
slow to start and -- torn down by Pool.terminate()'s SIGTERM while it unwinds torch -- can hang the parent).  Standard library only.

Protocol (binary, stdin -> stdout): the parent writes one path per line; for each path the child answers with an 8-byte
little-endian length followed by the gunzipped bytes, or with the length 2**64-1 followed by an 8-byte length and a UTF-8 error
message.  End of input ends the child with exit code 0."""
import os
import struct
import sys
import zlib


def main():
    inp, out = sys.stdin.buffer, sys.stdout.buffer
    for line in inp:
        path = line.rstrip(b"\n").decode("utf-8", "surrogateescape")
        try:
            with open(path, "rb") as f:
                raw = f.read()
            data = zlib.decompress(raw, wbits=47)   # gzip or zlib header, like gzip.open for a single-member file
            out.write(struct.pack("<Q", len(data)))
            out.write(data)
        except Exception as exc:  # noqa: BLE001 -- reported to the parent, which raises
            msg = f"{type(exc).__name__}: {exc}".encode("utf-8", "replace")
            out.write(struct.pack("<QQ", 2 ** 64 - 1, len(msg)))
            out.write(msg)
        out.flush()


if __name__ == "__main__":
    try:
        main()
    except BrokenPipeError:   # the parent stopped reading (it raised, or dropped the generator): nothing to clean up, leave quietly
        os._exit(0)
