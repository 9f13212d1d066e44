"""Dump / compare tsde_brownian_query outputs over a fixed set of queries (refactoring aid: the kernel's results must
not change by a single bit).  python tools/query_regress.py dump|check <file>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import torchsde_amd  # noqa: E402


def run():
    out = {}
    rng = np.random.default_rng(0)
    for dtype in (torch.float32, torch.float64):
        for levy in ("none", "space-time"):
            for shape in ((64, 8), (33, 5)):
                for grid in ("single", "dt", "tol"):
                    kw = dict(t0=0.0, t1=1.0, size=shape, dtype=dtype, device="cuda", entropy=1234,
                              levy_area_approximation=levy)
                    if grid == "dt":
                        kw["dt"] = 0.013
                    if grid == "tol":
                        kw.update(tol=1e-3, halfway_tree=True)
                    bm = torchsde_amd.BrownianInterval(**kw)
                    pts = sorted(rng.uniform(0, 1, size=6).tolist()) + [0.0, 1.0, 0.5, 0.25, 0.013 * 7, 0.013 * 8]
                    qs = [(0.0, 1.0), (0.0, 0.5), (0.5, 1.0), (0.013 * 7, 0.013 * 8), (0.013 * 7, 0.013 * 19),
                          (0.013 * 7 + 1e-6, 0.013 * 8 - 2e-6), (0.1, 0.1000001)]
                    for i in range(0, 6, 2):
                        qs.append((pts[i], pts[i + 1]))
                    qs.append((pts[0], pts[5]))
                    for (a, b) in qs:
                        key = f"{dtype}-{levy}-{shape}-{grid}-{a!r}-{b!r}"
                        if levy == "none":
                            out[key] = bm(a, b).cpu()
                        else:
                            W, U = bm(a, b, return_U=True)
                            out[key] = torch.stack([W, U]).cpu()
    return out


if __name__ == "__main__":
    mode, path = sys.argv[1], sys.argv[2]
    res = run()
    if mode == "dump":
        torch.save(res, path)
        print("dumped", len(res), "queries")
    else:
        ref = torch.load(path)
        bad = [k for k in ref if not torch.equal(ref[k], res[k])]
        print("checked", len(ref), "queries;", len(bad), "differ")
        for k in bad[:10]:
            print("  DIFF", k, (ref[k] - res[k]).abs().max().item())
        sys.exit(1 if bad else 0)
