"""One-off validation aid (CPU): the traversal kernel's compacted pair form (gs_body.h "PAIRC": rows of 33 ... 64 neighbours, codes by
ordinal, two lanes per fresh neighbour up to M = 96, four above) and the pair kernels' four-lane path (gs_quad) on the lane emulator
against the oracle's sequential GraphSearcher, random shapes.   usage: python scripts/fuzz_pairc_emulated.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gsearch_emulated as TG  # noqa: E402
from oracle import oracle as O  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
emu = TG.emu.__wrapped__()
t_end = time.time() + budget
cases = searches = 0
while time.time() < t_end:
    M = int(rng.choice([16, 32, 48, 64, 96, 128, 192]))
    compact = bool(rng.random() < 0.6)
    deg = int(rng.integers(33, 65)) if compact else int(rng.choice([8, 16, 24, 32]))
    levels = int(rng.integers(1, 4))
    N = int(rng.integers(600, 2500))
    lv, entry, entry_level, opq, codes, q = TG.problem(int(rng.integers(1, 1 << 30)), N, 8 * M, M, levels, deg=deg, nq=int(rng.integers(2, 7)))
    og = O.OracleGraph(codes.shape[0], lv, entry, entry_level)
    for _ in range(2):
        vsf = int(rng.integers(0, 3))
        rk = int(rng.choice([1, 10, 60, 150]))
        fused = (not compact) and bool(rng.random() < 0.6)
        os.environ["GS_EMU_QUAD"] = str(int(rng.integers(0, 2)))
        os.environ["EMU_LANE_ORDER"] = str(rng.choice(["", "reverse", "random:%d" % int(rng.integers(1, 99))]))
        wi, ws, wst = og.search(opq, codes, None, q, vsf, rk, rk, fused=fused)
        ids, sc, st, status, _ = TG.run_emu(emu, lv, entry, entry_level, opq, codes, q, vsf, rk, fused, pair=2 if compact else 1,
                                            cand_cap=int(rng.choice([128, 256])), v1_log2=int(rng.choice([0, 8, 10])))
        TG.check(ids, sc, st, status, wi, ws, wst)
        searches += 1
    cases += 1
print(f"fuzz_pairc_emulated: {cases} random graphs, {searches} searches, all bit-identical to the oracle (seed {seed}, {budget:.0f} s)")
