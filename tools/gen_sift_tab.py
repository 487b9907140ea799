#!/usr/bin/env python3
"""Writes the 64-entry table of cv::hal::exp32f (core/mathfuncs_core: expTab[i] = 2^(i/64) * EXPPOLY_32F_A0, converted
to float as the vector body of exp32f uses it) as C hex-float literals into oracle/sift_exptab.inc and
evenvizion_amd/csrc/sift_exptab.inc (two copies: the product never includes anything under oracle/)."""
import os
import struct

A0 = .9670371139572337719125840413672004409288e-2
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
vals = []
for i in range(64):
    d = (2.0 ** (i / 64.0)) * A0
    f = struct.unpack("f", struct.pack("f", d))[0]
    vals.append(float.hex(f) + "f")
text = ",\n".join(", ".join(vals[r:r + 4]) for r in range(0, 64, 4)) + "\n"
for rel in ("oracle/sift_exptab.inc", "evenvizion_amd/csrc/sift_exptab.inc"):
    path = os.path.join(root, rel)
    if not os.path.exists(path) or open(path).read() != text:
        open(path, "w").write(text)
