"""Python restatement of raxmlng_reader (ipk/src/ar.cpp:144-270) -- TEST INFRASTRUCTURE ONLY.

Text -> float follows fast-cpp-csv-parser's parse_float<float> (the un-vendored "strasser" submodule):
digit accumulation in float32, fraction by pos /= 10; x += d * pos, exponent by repeated multiplication.
That algorithm is restated from the library's public source, not from /root/reference: PARITY UNPINNED.
"""
import numpy as np

from oracle import ipk_oracle

f32 = np.float32
AA_IPK_FROM_RAXML = [1, 8, 11, 3, 6, 15, 16, 2, 5, 4, 7, 14, 0, 9, 10, 12, 13, 17, 18, 19]   # ar.cpp:227-234


def parse_float(text):
    s = text.strip(" ")
    i, neg = 0, False
    if i < len(s) and s[i] in "+-":
        neg = s[i] == "-"; i += 1
    x = f32(0)
    while i < len(s) and s[i].isdigit():
        x = f32(f32(x * f32(10)) + f32(int(s[i]))); i += 1
    if i < len(s) and s[i] in ".,":
        i += 1
        pos = f32(1)
        while i < len(s) and s[i].isdigit():
            pos = f32(pos / f32(10))
            x = f32(x + f32(f32(int(s[i])) * pos)); i += 1
    if i < len(s) and s[i] in "eE":
        e = int(s[i + 1:])
        if e != 0:
            base = f32(0.1) if e < 0 else f32(10)
            e = abs(e)
            while e != 1:
                if e % 2 == 0:
                    base = f32(base * base); e //= 2
                else:
                    x = f32(x * base); e -= 1
            x = f32(x * base)
    elif i != len(s):
        raise ValueError("no digit")
    return f32(-x) if neg else x


def read_file(path, sigma):
    """{label: float32 [sites, sigma] log10 matrix}, labels in order of first appearance."""
    out, order = {}, []
    with open(path) as fh:
        fh.readline()                                        # header, ar.cpp:159
        for line in fh:
            line = line.rstrip("\n").rstrip("\r")
            if not line or line.startswith("."):
                continue
            f = line.split("\t")
            label = f[0].strip(" ")
            vals = [parse_float(v) for v in f[3:3 + sigma]]
            if sigma == 20:
                vals = [vals[j] for j in AA_IPK_FROM_RAXML]
            row = ipk_oracle.log10f(np.array(vals, dtype=f32))
            if label not in out:
                out[label] = []; order.append(label)
            out[label].append(row)
    return {k: np.stack(out[k]) for k in order}, order
