#!/usr/bin/env python3
"""The 256-entry p(0) -> (mps, state) table the reference's callers can ever ask for, derived INDEPENDENTLY of this
repository's C code: numpy, double precision, straight from the formula in the reference's text.

    ref: CABAC/CABAC_ContextModelsInit.cpp:124-148 (xMapProbabilityToState): zero is the MPS iff p0 >= 0.5;
         pLPS = min(p0, 1 - p0) floored at 0.01875; state = clip(round(62 * log10(2 pLPS) / log10(0.0375)), 0, 62)
    callers always pass k / 255 (ref: CABAC/cabacDemo.m:77-78, ISS/+coder/cabacEncode.m:30-31, cabacDecode.m:13)

CABAC_ContextModelsInit.cpp includes mex.h and cannot be compiled in this image, so this table -- not a run of the
reference -- is what pins row a9 of SURVEY 8(a); tests/test_oracle_golden.py additionally holds the 21 anchor values the
survey's probe of the real code printed.  Writes tests/golden/p0_state_lut.json: packed[k] = (state << 1) | mps.
"""
import json
import os

import numpy as np

k = np.arange(256, dtype=np.float64)
p0 = k / 255.0
zero_is_mps = p0 >= 0.5
p_lps = np.where(zero_is_mps, 1.0 - p0, p0)
p_lps = np.maximum(p_lps, 0.01875)
x = 62.0 * np.log10(2.0 * p_lps) / np.log10(2.0 * 0.01875)
state = np.clip(np.floor(x + 0.5), 0, 62).astype(np.int64)  # C round(): halves away from zero (x >= 0 here)
mps = np.where(zero_is_mps, 0, 1)
packed = (state << 1) | mps
# no value sits within 1e-9 of a rounding boundary, so libm / numpy differences in log10 cannot change an entry
assert np.abs((x + 0.5) - np.round(x + 0.5)).min() > 1e-9
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "p0_state_lut.json")
json.dump(dict(note="packed[k] = (state << 1) | mps for p0 = k / 255; derived with numpy from the formula in "
                    "CABAC/CABAC_ContextModelsInit.cpp:124-148 (tests/golden/make_p0_fixture.py)",
               packed=packed.tolist()), open(out, "w"))
print("wrote", out)
