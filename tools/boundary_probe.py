#!/usr/bin/env python3
"""Per-iteration wall times of the boundary calls (ctypes twin of the LM loop) at a preset size, with and without
registered caller arrays: python tools/boundary_probe.py --preset final13682 --iterations 3"""
import argparse
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_cx():
    pkg = os.path.join(ROOT, "ceres-solver-ceres-solver_amd")
    spec = importlib.util.spec_from_file_location("cxschur", os.path.join(pkg, "__init__.py"), submodule_search_locations=[pkg])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["cxschur"] = mod
    spec.loader.exec_module(mod)
    return mod


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="final13682")
    ap.add_argument("--iterations", type=int, default=3)
    ap.add_argument("--shards", type=int, default=1)
    ap.add_argument("--register", type=int, nargs="+", default=[0, 1])
    ap.add_argument("--torch", action="store_true", help="import torch first (as bench.py does)")
    args = ap.parse_args()
    if args.torch:
        import torch  # noqa: F401
    cx = load_cx()
    prob = cx.bal.make_preset(args.preset)
    kw = dict(type=cx.ITERATIVE_SCHUR, preconditioner_type=cx.JACOBI, num_eliminate_blocks=prob.num_points,
              max_num_iterations=500, min_num_iterations=0, residual_reset_period=10)
    ctx = cx.Context(0) if args.shards == 1 else cx.Context(devices=[0] * args.shards)
    for reg in args.register:
        loop = cx.boundary.BoundaryLoop(ctx, prob, kw, eta=0.1, register_arrays=reg)
        rep = loop.run(args.iterations)
        loop.close()
        for r in rep["per_iteration"]:
            print(json.dumps({"register": reg, "shards": args.shards, "through_interfaces_ms": round(r["through_interfaces_ms"], 2),
                              "calls_ms": {k: round(v, 2) for k, v in r["calls_ms"].items()}, "caller_ms": round(r["caller_ms"], 1),
                              "h2d_ms": round(r["h2d_ms"], 2), "d2h_ms": round(r["d2h_ms"], 2), "cg": r["cg_iterations"]}), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
