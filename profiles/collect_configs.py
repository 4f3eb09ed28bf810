"""Runs bench.py on every BASELINE.json configuration (and the solver variants DESIGN.md §7 tabulates) and writes one
JSON line per run to profiles/r<round>_config_runs.jsonl (round = CX_ROUND, default 02): ms per LinearSolver::Solve at eta = 0.1 and at eta = 1e-2,
phases, per-kernel averages.  Usage (GPU box, repo root): python profiles/collect_configs.py"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RUNS = [
    "--workload ladybug49 --solver iterative_schur --steps 20 --warmup 3",
    "--workload ladybug49 --solver cgnr --steps 20 --warmup 3",
    "--workload dubrovnik356 --solver dense_schur --steps 10 --warmup 3",
    "--workload dubrovnik356 --solver iterative_schur --steps 10 --warmup 3",
    "--workload dubrovnik356 --solver iterative_schur --preconditioner cluster_jacobi --steps 10 --warmup 3",
    "--workload final13682 --solver iterative_schur --steps 5 --warmup 2",
    "--workload final13682 --solver iterative_schur --mixed --steps 5 --warmup 2",
    "--workload final13682 --solver iterative_schur --preconditioner schur_jacobi --explicit-schur --steps 3 --warmup 1",
    "--workload final13682 --solver iterative_schur --preconditioner cluster_jacobi --steps 3 --warmup 1",
    "--workload final13682 --solver iterative_schur --preconditioner cluster_tridiagonal --steps 3 --warmup 1",
    "--workload final13682 --solver sparse_schur --steps 3 --warmup 1",
    "--workload final13682 --solver sparse_schur --mixed --steps 3 --warmup 1",
    "--workload final13682 --solver sparse_schur --mixed --refinements 1 --steps 3 --warmup 1",
    "--workload dubrovnik356 --solver sparse_schur --steps 10 --warmup 3 --force-tile-sparse",
    "--workload synthetic10M --solver cgnr --steps 10 --warmup 3",
    "--workload synthetic10M --solver cgnr --mixed --steps 10 --warmup 3",
    # the second scene (round 3): the Final sizes with loop closures (5 % of the points also seen half a ring away)
    "--workload final13682_revisit --solver iterative_schur --steps 5 --warmup 2",
    "--workload final13682_revisit --solver sparse_schur --steps 3 --warmup 1",
    "--workload final13682_revisit --solver sparse_schur --mixed --steps 3 --warmup 1",
    "--workload final13682_revisit --solver iterative_schur --preconditioner cluster_tridiagonal --steps 3 --warmup 1",
]


def bench(args, eta):
    env = dict(os.environ)
    if "--force-tile-sparse" in args:      # SPARSE_SCHUR stays dense below 512 cameras unless told otherwise
        env["CX_SPARSE_CHOLESKY"] = "1"
        args = args.replace(" --force-tile-sparse", "")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-sparse-schur", "--no-boundary", "--eta", str(eta)] + args.split()
    out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True, check=True, env=env).stdout
    return json.loads([l for l in out.splitlines() if l.startswith("{")][-1])


def main():
    # (through gpurun only gpurun_out/ travels back from the GPU box: copy the file into profiles/ afterwards)
    out_dir = os.path.join(ROOT, "gpurun_out") if os.environ.get("GRAFT_REPO_ROOT") else os.path.join(ROOT, "profiles")
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, "r%s_config_runs.jsonl" % os.environ.get("CX_ROUND", "02"))
    with open(path, "w") as f:
        for args in RUNS:
            a = bench(args, 0.1)
            rec = {"args": args, "ms": round(a["value"], 3), "cg_iterations": a["config"].get("cg_iterations"),
                   "phases_ms": {k: round(v, 3) for k, v in a["phases_ms_per_solve"].items() if k != "setup_ms" and not k.startswith("allreduce")},
                   "values_update_ms": {k: round(v, 3) for k, v in (a.get("values_update_per_lm_iteration") or {}).items() if k.endswith("_ms")},
                   "kernels_avg_ms": {k: round(v["avg_ms"], 4) for k, v in (a.get("kernels") or {}).items()},
                   "lm_iteration_ms": round(a["lm_iteration_ms"], 3) if a.get("lm_iteration_ms") else None,
                   "spmv_frac_of_8TBps": [round(a["spmv"]["right_frac_of_8TBps"], 3), round(a["spmv"]["left_frac_of_8TBps"], 3)]}
            if "sparse_schur" not in args and "dense_schur" not in args:
                b = bench(args, 0.01)
                it = b["config"].get("cg_iterations") or 0
                rec["eta_0.01"] = {"ms": round(b["value"], 3), "cg_iterations": it,
                                   "cg_ms_per_iteration": round(b["phases_ms_per_solve"]["reduced_solve_ms"] / it, 4) if it else None}
            f.write(json.dumps(rec) + "\n")
            f.flush()
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
