"""Timestamp filter for progress-bar output: splits stdin on \\r / \\n and prefixes each piece with seconds since start.
`python -u script | python tools/stamp.py > log` -- used by tools/script_head_to_head.sh to time phases of an UNMODIFIED script."""
import os
import sys
import time

t0 = time.time()
buf = b""
out = sys.stdout
while True:
    chunk = os.read(0, 65536)
    if not chunk:
        break
    buf += chunk
    while True:
        cut = min((i for i in (buf.find(b"\r"), buf.find(b"\n")) if i >= 0), default=-1)
        if cut < 0:
            break
        piece, buf = buf[:cut], buf[cut + 1:]
        if piece.strip():
            out.write(f"{time.time() - t0:9.3f} {piece.decode(errors='replace')}\n")
            out.flush()
if buf.strip():
    out.write(f"{time.time() - t0:9.3f} {buf.decode(errors='replace')}\n")
