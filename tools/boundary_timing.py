#!/usr/bin/env python3
"""What one LM iteration costs THROUGH the Evaluator / SparseMatrix / LinearSolver adapters with host vectors, at a
BASELINE size: writes a bal.py preset as the binary problem file host/test_host_adapter --time reads, then runs that
program (the real adapter classes, the unmodified LM call sequence) for one shard and for logical shards on device 0,
with the adapters' three opt-ins and without.  One JSON line per run, appended to the file named by --out.

  python tools/boundary_timing.py --preset final13682 --shards 1 4 --out gpurun_out/boundary.jsonl
"""
import argparse
import importlib.util
import json
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_cx():
    pkg = os.path.join(ROOT, "ceres-solver-ceres-solver_amd")
    spec = importlib.util.spec_from_file_location("cxschur", os.path.join(pkg, "__init__.py"), submodule_search_locations=[pkg])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["cxschur"] = mod
    spec.loader.exec_module(mod)
    return mod


def write_problem(prob, path):
    """The program order of a Schur-type solve: residual blocks grouped by point (stable)."""
    order = np.argsort(prob.point_index, kind="stable")
    with open(path, "wb") as f:
        np.array([prob.num_cameras, prob.num_points, prob.num_observations], dtype=np.int64).tofile(f)
        np.ascontiguousarray(prob.camera_index[order], dtype=np.int32).tofile(f)
        np.ascontiguousarray(prob.point_index[order], dtype=np.int32).tofile(f)
        np.ascontiguousarray(prob.observations[order], dtype=np.float64).tofile(f)
        np.ascontiguousarray(prob.state(), dtype=np.float64).tofile(f)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--preset", default="final13682")
    ap.add_argument("--shards", type=int, nargs="+", default=[1, 4])
    ap.add_argument("--iterations", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--eta", type=float, default=0.1)
    ap.add_argument("--plain-too", action="store_true", help="also run without the three opt-ins")
    ap.add_argument("--pin", default=None, help="CX_PIN for the child (0: no registration)")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    cx = load_cx()
    prob = cx.bal.make_preset(args.preset)
    path = os.path.join(os.environ.get("TMPDIR", "/tmp"), "cx_boundary_%s.bin" % args.preset)
    write_problem(prob, path)
    exe = os.path.join(ROOT, "ceres-solver-ceres-solver_amd", "host", "test_host_adapter")
    env = dict(os.environ)
    if args.pin is not None:
        env["CX_PIN"] = args.pin
    lines = []
    for shards in args.shards:
        for plain in ([False, True] if args.plain_too else [False]):
            cmd = [exe, "--time", "--problem", path, "--iterations", str(args.iterations), "--warmup", str(args.warmup),
                   "--eta", str(args.eta), "--shards", str(shards)] + (["--plain"] if plain else [])
            out = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True, check=True).stdout.strip().splitlines()[-1]
            rec = json.loads(out)
            rec["preset"] = args.preset
            rec["CX_PIN"] = args.pin
            print(json.dumps(rec), flush=True)
            lines.append(rec)
    if args.out:
        with open(args.out, "a") as f:
            for rec in lines:
                f.write(json.dumps(rec) + "\n")
    os.remove(path)


if __name__ == "__main__":
    main()
