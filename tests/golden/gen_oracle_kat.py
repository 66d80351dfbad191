"""Generates tests/golden/oracle_kat.json: float64 known answers for the reference's own
native known-answer inputs.  Independent of oracle/ (pure numpy, float64 accumulation).

Inputs restate make_vec(n, seed) from
  /root/reference/jvector-native/src/main/native/tests/test_helpers.cpp:78-87
(float32 arithmetic, element by element) at the 19 canonical lengths (:49-76), with the seeds the
reference tests use (0.7 / 1.3: test_similarity.cpp:94-95; 0.9: :150; 1.0: :184).
Run:  python tests/golden/gen_oracle_kat.py
"""
import json
import os

import numpy as np

LENGTHS = [1, 3, 4, 5, 7, 8, 9, 15, 16, 17, 19, 32, 33, 37, 64, 71, 100, 128, 255]


def make_vec(n, seed):
    f = np.float32
    v = np.empty(n, f)
    for i in range(n):
        x = f(seed) * (f(1.0) + f(i % 7) * f(0.13))
        if i % 3 == 0:
            x = -x
        v[i] = f(x + f(0.5))
    return v


def main():
    out = {"lengths": LENGTHS, "cases": []}
    for n in LENGTHS:
        a, b = make_vec(n, 0.7), make_vec(n, 1.3)
        a64, b64 = a.astype(np.float64), b.astype(np.float64)
        out["cases"].append({
            "n": n,
            "a_first": [float(x) for x in a[:4]],
            "dot": float(a64 @ b64),
            "l2": float(((a64 - b64) ** 2).sum()),
            "cosine": float((a64 @ b64) / np.sqrt((a64 @ a64) * (b64 @ b64))),
        })
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle_kat.json")
    with open(path, "w") as fh:
        json.dump(out, fh, indent=1)
    print("wrote", path)


if __name__ == "__main__":
    main()
