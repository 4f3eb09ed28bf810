#!/usr/bin/env python3
"""Bundle adjustment of a BAL problem on one MI355X through libcxschur -- the flags, flow and
progress table of the reference's examples/bundle_adjuster.cc (flags :76-150, SetLinearSolver
:158-190, SetMinimizerOptions :271-297, SolveProblem :363-395) for the solvers this library
implements.  The whole LM loop runs device-resident (cx_minimize).

  python bundle_adjuster.py --input problem-49-7776-pre.txt --linear_solver iterative_schur \\
         --preconditioner schur_jacobi --num_iterations 10
  python bundle_adjuster.py --preset ladybug49 --linear_solver dense_schur --robustify

There is no CPU path: without a gfx950 device and the built library this exits with an error.
"""
import argparse
import importlib.util
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)


def load_package():
    spec = importlib.util.spec_from_file_location("cxschur", os.path.join(PKG, "__init__.py"),
                                                  submodule_search_locations=[PKG])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["cxschur"] = mod
    spec.loader.exec_module(mod)
    return mod


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--input", help="BAL text file (examples/bal_problem.cc format)")
    ap.add_argument("--preset", help="synthetic BAL-shaped problem instead of a file: " + ", ".join(
        ["ladybug16", "ladybug49", "dubrovnik356", "final13682", "synthetic10M"]))
    ap.add_argument("--linear_solver", default="iterative_schur",
                    choices=["dense_schur", "sparse_schur", "iterative_schur", "cgnr"])
    ap.add_argument("--preconditioner", default="jacobi",
                    choices=["identity", "jacobi", "schur_jacobi", "schur_power_series_expansion", "cluster_jacobi",
                             "cluster_tridiagonal"])
    ap.add_argument("--visibility_clustering", default="canonical_views", choices=["canonical_views", "single_linkage"])
    ap.add_argument("--explicit_schur_complement", action="store_true",
                    help="ITERATIVE_SCHUR on the explicitly computed block-sparse S (needs schur_jacobi)")
    ap.add_argument("--num_iterations", type=int, default=5)
    ap.add_argument("--max_linear_solver_iterations", type=int, default=500)
    ap.add_argument("--eta", type=float, default=1e-2)
    ap.add_argument("--robustify", action="store_true", help="HuberLoss(1.0) on every residual block")
    ap.add_argument("--use_quaternions", action="store_true",
                    help="quaternion cameras on ProductManifold<QuaternionManifold, EuclideanManifold<6>> "
                         "(the reference's --use_quaternions --use_manifolds)")
    ap.add_argument("--mixed_precision_solves", action="store_true", help="bundle_adjuster.cc:141, 170-171")
    ap.add_argument("--max_num_refinement_iterations", type=int, default=0, help="bundle_adjuster.cc:142, 172-173")
    ap.add_argument("--nonmonotonic_steps", action="store_true")
    ap.add_argument("--rotation_sigma", type=float, default=0.0)
    ap.add_argument("--translation_sigma", type=float, default=0.0)
    ap.add_argument("--point_sigma", type=float, default=0.0)
    ap.add_argument("--random_seed", type=int, default=38401)
    ap.add_argument("--no_normalize", action="store_true", help="skip BALProblem::Normalize")
    ap.add_argument("--final_bal", help="write the optimised problem to this BAL file")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--devices", help="comma separated device list: several shards behind one set of handles in this process "
                                      "(cx_context_create_multi) -- distinct devices exchange through RCCL, a repeated id means "
                                      "logical shards on that GPU, e.g. 0,0,0,0")
    args = ap.parse_args(argv)
    if bool(args.input) == bool(args.preset):
        ap.error("exactly one of --input / --preset")

    cx = load_package()
    bal = cx.bal
    t0 = time.time()
    if args.input:
        prob = bal.read_bal(args.input)
        if not args.no_normalize:
            prob = bal.normalize(prob)  # bundle_adjuster.cc:371
    else:
        prob = bal.make_preset(args.preset)
    if args.rotation_sigma or args.translation_sigma or args.point_sigma:
        prob = bal.perturb(prob, args.rotation_sigma, args.translation_sigma, args.point_sigma, args.random_seed)
    print("problem: %d cameras, %d points, %d observations (loaded in %.2f s)" %
          (prob.num_cameras, prob.num_points, prob.num_observations, time.time() - t0))

    ctx = cx.Context(devices=[int(d) for d in args.devices.split(",")]) if args.devices else cx.Context(args.device)
    print("device:", ctx.name, "(%d shards)" % ctx.num_shards if args.devices else "")
    t0 = time.time()
    ev = cx.Evaluator(ctx, prob)
    if args.robustify:
        ev.set_loss(cx.binding.LOSS_HUBER, 1.0)  # bundle_adjuster.cc:327-328
    if args.use_quaternions:
        ev.set_camera_model(cx.binding.CAMERA_QUATERNION_MANIFOLD)  # bundle_adjuster.cc:316-346
    stype = getattr(cx.binding, args.linear_solver.upper())
    solver = cx.Solver(ctx, type=stype, preconditioner_type=getattr(cx.binding, args.preconditioner.upper()),
                       num_eliminate_blocks=0 if stype == cx.binding.CGNR else prob.num_points,
                       max_num_iterations=args.max_linear_solver_iterations,
                       use_explicit_schur_complement=int(args.explicit_schur_complement),
                       use_mixed_precision_solves=int(args.mixed_precision_solves),
                       max_num_refinement_iterations=args.max_num_refinement_iterations,
                       visibility_clustering_type=getattr(cx.binding, args.visibility_clustering.upper()))
    preprocess_s = time.time() - t0
    opts = cx.binding.minimizer_options(max_num_iterations=args.num_iterations, eta=args.eta,
                                        use_nonmonotonic_steps=int(args.nonmonotonic_steps))
    start = bal.state_quaternion(prob) if args.use_quaternions else prob.state()
    state, summary, iterations = cx.binding.minimize(ev, solver, start, opts)

    # LoggingCallback's table (callbacks.cc:97-118)
    print("iter      cost      cost_change  |gradient|   |step|    tr_ratio  tr_radius  ls_iter  iter_time  total_time")
    total = 0.0
    for it in iterations:
        total += it["iteration_ms"] * 1e-3
        print("% 4d % 8e   % 3.2e   % 3.2e  % 3.2e  % 3.2e % 3.2e     % 4d   % 3.2e   % 3.2e" %
              (it["iteration"], it["cost"], it["cost_change"], it["gradient_max_norm"], it["step_norm"],
               it["relative_decrease"], it["trust_region_radius"], it["linear_solver_iterations"],
               it["iteration_ms"] * 1e-3, total))
    term = {0: "CONVERGENCE", 1: "NO_CONVERGENCE", 2: "FAILURE"}[summary["termination_type"]]
    lin_ms = sum(it["linear_solver_ms"] for it in iterations)
    jac_ms = sum(it["jacobian_ms"] for it in iterations)
    res_ms = sum(it["residual_ms"] for it in iterations)
    print("\nCost:\nInitial   % e\nFinal     % e\nChange    % e" %
          (summary["initial_cost"], summary["final_cost"], summary["initial_cost"] - summary["final_cost"]))
    print("\nMinimizer iterations %d  (successful %d, unsuccessful %d)" %
          (len(iterations), summary["num_successful_steps"], summary["num_unsuccessful_steps"]))
    print("Time (s): preprocessor %.4f  linear solver %.4f  jacobian evaluation %.4f  residual evaluation %.4f  "
          "minimizer %.4f" % (preprocess_s, lin_ms * 1e-3, jac_ms * 1e-3, res_ms * 1e-3, summary["total_ms"] * 1e-3))
    print("Termination: %s (%s)" % (term, summary["message"]))
    if args.final_bal:
        P = prob.num_points
        if args.use_quaternions:
            out = bal.cameras_from_quaternion_state(prob, state)
        else:
            out = bal.dataclasses.replace(prob, points=state[:3 * P].reshape(P, 3).copy(),
                                          cameras=state[3 * P:].reshape(prob.num_cameras, 9).copy())
        bal.write_bal(args.final_bal, out)
    solver.close()
    ev.close()
    ctx.close()
    return 0 if summary["termination_type"] != 2 else 1


if __name__ == "__main__":
    sys.exit(main())
