"""Extracts the reference's own dump of its instantiated equations (src/old test.txt, a UTF-16 capture of
the commented-out `writeTerms` calls at src/Main.hs:357-390) into tests/golden/term_pins.json.

The dump prints every term list REVERSED (writeTermsOrig folds over `reverse terms`, src/Main.hs:324); the
lists are stored here in list order.  A right-hand-side term prints as
    approxDeriv ( <value> <show denom> <show side> )
where denom is the direction of the inner derivative and side the face of the outer one
(integSurface_adj, src/CalculusUtils.hs:80); a left-hand-side term as `<v>X` (Unknown) or `<v>` (Constant).
The values belong to an older configuration (16 x 16 grid, unit fields) and are kept only for their
signs; the structure -- order, directions, faces, signs -- is what the current code can be held to.

    python tests/golden/make_term_pins.py [/root/reference]
"""
import json
import os
import re
import sys

ref = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
with open(os.path.join(ref, "src", "old test.txt"), "rb") as f:
    lines = f.read().decode("utf-16").splitlines()

HEADS = {"continuity": "Density", "U Momentum": "U", "V Momentum": "V", "W Momentum": "W", "ENERGY": "Temperature"}
out = {"source": "src/old test.txt", "equations": {}}
i = 0
while i < len(lines):
    m = re.match(r"\s*([A-Za-z ]+?)\s*-{6,}\s*$", lines[i])
    if m and m.group(1).strip() in HEADS:
        name = HEADS[m.group(1).strip()]
        rhs_line, lhs_line = lines[i + 1], lines[i + 3]
        assert lines[i + 2].strip() == "=", lines[i + 2]
        rhs = []
        for v, d, s in re.findall(r"approxDeriv \(\s*(\S+)\s+([XYZ])\s+(\w+)\s*\)", rhs_line):
            rhs.append([d, s, float(v)])
        if not rhs:                                     # continuity: the single constant 0
            rhs = [["C", float(t)] for t in rhs_line.split("+")]
        lhs = []
        for t in lhs_line.split("+"):
            t = t.strip()
            lhs.append(["X", float(t[:-1])] if t.endswith("X") else ["C", float(t)])
        out["equations"][name] = {"line_rhs": i + 2, "line_lhs": i + 4,
                                  "rhs": rhs[::-1], "lhs": lhs[::-1]}
        i += 4
    i += 1
dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "term_pins.json")
with open(dst, "w") as f:
    json.dump(out, f, indent=1)
print({k: (len(v["lhs"]), len(v["rhs"])) for k, v in out["equations"].items()})
