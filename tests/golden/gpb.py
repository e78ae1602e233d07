"""GPB1 — the little-endian container the Julia golden generator and the Python tests exchange (tests/golden/make_golden.jl).

  magic "GPB1" | uint32 count | count × { uint32 len | utf-8 name | uint32 ndim | int64 dims[ndim] | float64 data, COLUMN-major }

Column-major because that is Julia's native order: `read!(io, Array{Float64}(undef, dims...))` on that side, `order="F"` here.
Every value is float64 (integers such as the kernel kind are stored as float64 scalars), so both languages read identical bits."""
import struct
from pathlib import Path

import numpy as np

MAGIC = b"GPB1"


def write(path, arrays: dict) -> None:
    with open(path, "wb") as f:
        f.write(MAGIC)
        f.write(struct.pack("<I", len(arrays)))
        for name, a in arrays.items():
            a = np.asarray(a, dtype=np.float64)
            nb = name.encode("utf-8")
            f.write(struct.pack("<I", len(nb)))
            f.write(nb)
            f.write(struct.pack("<I", a.ndim))
            for d in a.shape:
                f.write(struct.pack("<q", d))
            f.write(np.asfortranarray(a).ravel(order="F").astype("<f8").tobytes())


def read(path) -> dict:
    buf = Path(path).read_bytes()
    if buf[:4] != MAGIC:
        raise ValueError(f"{path}: not a GPB1 file")
    (count,) = struct.unpack_from("<I", buf, 4)
    off, out = 8, {}
    for _ in range(count):
        (ln,) = struct.unpack_from("<I", buf, off)
        off += 4
        name = buf[off:off + ln].decode("utf-8")
        off += ln
        (nd,) = struct.unpack_from("<I", buf, off)
        off += 4
        dims = struct.unpack_from(f"<{nd}q", buf, off) if nd else ()
        off += 8 * nd
        cnt = int(np.prod(dims)) if nd else 1
        data = np.frombuffer(buf, dtype="<f8", count=cnt, offset=off).astype(np.float64)
        off += 8 * cnt
        out[name] = data.reshape(dims, order="F") if nd else data[0]
    return out
