"""Fits ops._TILE_MODEL (microseconds per K-step with <= 128 / > 128 workgroups on the chip, fixed microseconds per workgroup round) to the
rows tools/gemm_bench.py printed:  t = rounds x (u x K-steps + f).  Least squares per tile over every (shape, M) row that timed it.
usage: python tools/fit_tile_model.py profiles/r04_gemm_bench.jsonl [more.jsonl ...]"""
import json
import sys

import numpy as np

TILES = {0: (128, 128, 2), 4: (256, 256, 1), 7: (160, 256, 1), 8: (128, 128, 1)}


def main():
    rows = []
    for path in sys.argv[1:]:
        for line in open(path):
            try:
                rows.append(json.loads(line))
            except ValueError:
                pass
    for tile, (bm, bn, per_cu) in TILES.items():
        a, y, used = [], [], []
        for r in rows:
            key = f"native_t{tile}_us"
            if key not in r:
                continue
            m, n, k = r["M"], r["N"], r["K"]
            tiles = -(-m // bm) * -(-n // bn)
            rounds = -(-tiles // (256 * per_cu))
            full = tiles > (256 if per_cu > 1 else 128)
            ks = -(-k // 64)
            a.append([0.0 if full else rounds * ks, rounds * ks if full else 0.0, rounds])
            y.append(r[key])
            used.append((r["shape"], m, tiles, rounds, "full" if full else "light", r[key]))
        if len(a) < 3:
            print(f"tile {tile}: {len(a)} rows — not enough")
            continue
        a, y = np.array(a), np.array(y)
        cols = [i for i in range(3) if np.any(a[:, i] != 0)]
        sol, *_ = np.linalg.lstsq(a[:, cols], y, rcond=None)
        full_sol = [None, None, None]
        for c, v in zip(cols, sol):
            full_sol[c] = float(v)
        pred = a[:, cols] @ sol
        err = np.abs(pred - y) / y
        print(f"tile {tile}: ({bm}, {bn}, {per_cu}, u_light {full_sol[0]}, u_full {full_sol[1]}, fixed {full_sol[2]})   rows {len(y)}  "
              f"median |err| {np.median(err):.1%}  worst {err.max():.1%}")
        for u, p_ in zip(used, pred):
            print(f"    {u[0]:4s} M {u[1]:6d} tiles {u[2]:5d} rounds {u[3]:2d} {u[4]:5s} measured {u[5]:7.1f} us  model {p_:7.1f}")


if __name__ == "__main__":
    main()
