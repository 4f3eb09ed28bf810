function lsqp = load_trust_region_problem()
lsqp.num_rows = 340;
lsqp.num_cols = 174;
tmp = load('ceres_solver_iteration_000_A.txt', '-ascii');
lsqp.A = sparse(tmp(:, 1) + 1, tmp(:, 2) + 1, tmp(:, 3), 340, 174);
lsqp.D = load('ceres_solver_iteration_000_D.txt', '-ascii');
lsqp.b = load('ceres_solver_iteration_000_b.txt', '-ascii');
lsqp.x = load('ceres_solver_iteration_000_x.txt', '-ascii');
