// CxBalEvaluator: libcxschur's bundle-adjustment evaluator behind Ceres' Evaluator interface (evaluator.h:60-167).
//
// It stands where Evaluator::Create (evaluator.cc:53-99) would return
// ProgramEvaluator<BlockEvaluatePreparer, BlockJacobianWriter> for DENSE_SCHUR / SPARSE_SCHUR / ITERATIVE_SCHUR, for
// programs that are bundle adjustment with SnavelyReprojectionError (examples/snavely_reprojection_error.h:53-104):
// residuals, cost, gradient and the block-sparse Jacobian are computed by k_bal_evaluate on the device, and the
// Jacobian it hands out (CreateJacobian) is a CxDeviceJacobian whose values stay in HBM -- the 5.6 GB of J per LM
// iteration on Final-13682 are never copied (the reference's own CUDA CGNR path pays that copy every iteration,
// cgnr_solver.cc:343-348).
//
// TryCreate is the line the factory patch calls: it returns nullptr -- and the factory falls through
// to the reference's ProgramEvaluator -- unless the program is exactly what the device kernels implement.
#ifndef CX_BAL_EVALUATOR_H_
#define CX_BAL_EVALUATOR_H_

#include <memory>
#include <string>
#include <vector>

#include "cx_device_jacobian.h"
#include "cx_loss_probe.h"
#ifdef CX_USE_CERES_HEADERS
#include "ceres/manifold.h"
#include "ceres/product_manifold.h"
#include "ceres/autodiff_cost_function.h"
#include "ceres/evaluator.h"
#include "ceres/execution_summary.h"
#include "ceres/parameter_block.h"
#include "ceres/program.h"
#include "ceres/residual_block.h"
#include "snavely_reprojection_error.h"
#endif

namespace ceres::internal {

cx_context* CxSharedContext();  // cx_linear_solver.h: the context of the configured device list

// A bundle-adjustment program in the order the Schur preprocessing leaves it (reorder_program.cc:446-540):
// parameter blocks = points 0..P-1 then cameras 0..C-1, residual blocks grouped by point in ascending point order.
struct CxBalProblemView {
  int32_t num_cameras = 0;
  int32_t num_points = 0;
  int64_t num_observations = 0;
  const int32_t* camera_index = nullptr;    // [num_observations], residual-block order
  const int32_t* point_index = nullptr;     // [num_observations], non-decreasing
  const double* observations_xy = nullptr;  // [2 * num_observations]
  // cx_camera_model: CX_CAMERA_ANGLE_AXIS (9 parameters) or CX_CAMERA_QUATERNION_MANIFOLD (10 ambient, 9 tangent)
  int32_t camera_model = CX_CAMERA_ANGLE_AXIS;
  // the robust loss every residual block carries (cx_loss_type and its constructor arguments)
  int32_t loss_type = CX_LOSS_NONE;
  double loss_a = 0.0, loss_b = 0.0;
};

class CxBalEvaluator final : public Evaluator {
 public:
  // nullptr + *error when the view is not in Schur order or the device refuses it.
  static std::unique_ptr<CxBalEvaluator> Create(const Evaluator::Options& options, const CxBalProblemView& view,
                                                std::string* error) {
    for (int64_t i = 1; i < view.num_observations; ++i) {
      if (view.point_index[i] < view.point_index[i - 1]) {
        *error = "residual blocks are not grouped by point (LexicographicallyOrderResidualBlocks has not run)";
        return nullptr;
      }
    }
    cx_context* ctx = CxSharedContext();
    if (ctx == nullptr) {
      *error = std::string("cxschur: ") + cx_last_error();
      return nullptr;
    }
    // cx_evaluator_create_bal takes observations in INPUT order and orders them as
    // LexicographicallyOrderResidualBlocks does (buckets by point, each filled from its back,
    // reorder_program.cc:296-314).  The program here is already in that order; feeding it back to front makes the
    // device's rows come out in exactly the program's residual-block order (checked below), so `residuals`, `b` of
    // the linear solver and the rows of J all share one numbering and nothing is permuted at run time.
    const int64_t O = view.num_observations;
    std::vector<int32_t> cam(static_cast<size_t>(O)), pt(static_cast<size_t>(O));
    std::vector<double> obs(static_cast<size_t>(2 * O));
    for (int64_t i = 0; i < O; ++i) {
      const int64_t k = O - 1 - i;
      cam[size_t(k)] = view.camera_index[i];
      pt[size_t(k)] = view.point_index[i];
      obs[size_t(2 * k)] = view.observations_xy[2 * i];
      obs[size_t(2 * k + 1)] = view.observations_xy[2 * i + 1];
    }
    auto handle = std::make_shared<CxEvaluatorHandle>();
    handle->ctx = ctx;
    if (cx_evaluator_create_bal(ctx, view.num_cameras, view.num_points, O, cam.data(), pt.data(), obs.data(),
                                &handle->evaluator) != CX_OK) {
      *error = std::string("cxschur: ") + cx_last_error();
      return nullptr;
    }
    std::vector<int64_t> row(static_cast<size_t>(O));
    if (cx_evaluator_row_of_observation(handle->evaluator, row.data()) != CX_OK) {
      *error = std::string("cxschur: ") + cx_last_error();
      return nullptr;
    }
    for (int64_t i = 0; i < O; ++i) {
      if (row[size_t(O - 1 - i)] != i) {
        *error = "device row order differs from the program's residual-block order";
        return nullptr;
      }
    }
    if (cx_evaluator_set_camera_model(handle->evaluator, view.camera_model) != CX_OK) {
      *error = std::string("cxschur: ") + cx_last_error();
      return nullptr;
    }
    std::unique_ptr<CxBalEvaluator> e(new CxBalEvaluator(options, std::move(handle), view));
    if (view.loss_type != CX_LOSS_NONE && !e->SetLoss(view.loss_type, view.loss_a, view.loss_b)) {
      *error = std::string("cxschur: ") + cx_last_error();
      return nullptr;
    }
    return e;
  }

  // The factory hook: a CxBalEvaluator when `program` is a bundle-adjustment program the device kernels cover,
  // nullptr (with the reason in *why_not) otherwise.  Covered -- the programs examples/bundle_adjuster.cc:306-346
  // builds: a Schur-type linear solver with num_eliminate_blocks = number of points > 0; all point blocks first (size 3,
  // Euclidean), then all camera blocks, which are either
  //   * size 9, Euclidean, with AutoDiffCostFunction<SnavelyReprojectionError, 2, 9, 3> residual blocks, or
  //   * size 10 on ProductManifold<QuaternionManifold, EuclideanManifold<6>> (--use_quaternions --use_manifolds) with
  //     AutoDiffCostFunction<SnavelyReprojectionErrorWithQuaternions, 2, 10, 3> residual blocks;
  // every residual block on (camera, point) with no loss function or with ONE built-in loss (the same type and
  // arguments on all blocks; --robustify: HuberLoss(1.0)), recovered through LossFunction::Evaluate (cx_loss_probe.h);
  // no constant blocks (the preprocessor has removed them, trust_region_preprocessor.cc:95-120), no bounds, no
  // evaluation callback.
  static std::unique_ptr<Evaluator> TryCreate(const Evaluator::Options& options, Program* program, std::string* why_not) {
    using Snavely = AutoDiffCostFunction<examples::SnavelyReprojectionError, 2, 9, 3>;
    using SnavelyQuaternion = AutoDiffCostFunction<examples::SnavelyReprojectionErrorWithQuaternions, 2, 10, 3>;
    using CameraManifold = ProductManifold<QuaternionManifold, EuclideanManifold<6>>;
    auto no = [&](const char* why) { *why_not = why; return std::unique_ptr<Evaluator>(); };
    if (options.linear_solver_type != DENSE_SCHUR && options.linear_solver_type != SPARSE_SCHUR &&
        options.linear_solver_type != ITERATIVE_SCHUR)
      return no("linear solver is not of Schur type");
    if (options.evaluation_callback != nullptr) return no("evaluation callback present");
    if (options.dynamic_sparsity) return no("dynamic sparsity");
    const int P = options.num_eliminate_blocks;
    const std::vector<ParameterBlock*>& blocks = program->parameter_blocks();
    const int C = int(blocks.size()) - P;
    if (P <= 0 || C <= 0) return no("no e-blocks or no f-blocks");
    const bool quaternion = blocks[size_t(P)]->Size() == 10;
    const int camera_size = quaternion ? 10 : 9;
    for (int j = 0; j < P + C; ++j) {
      const ParameterBlock* b = blocks[size_t(j)];
      if (j < P || !quaternion) {
        const int want = j < P ? 3 : 9;
        if (b->Size() != want || b->TangentSize() != want || b->manifold() != nullptr) return no("parameter block is not a Euclidean 3-point / 9-camera");
      } else if (b->Size() != 10 || b->TangentSize() != 9 || dynamic_cast<const CameraManifold*>(b->manifold()) == nullptr) {
        return no("10-parameter camera is not on ProductManifold<QuaternionManifold, EuclideanManifold<6>>");
      }
      if (b->IsConstant()) return no("constant parameter block");
      if (b->lower_bounds() != nullptr || b->upper_bounds() != nullptr) return no("bounds");
      if (b->index() != j || b->state_offset() != (j < P ? 3 * j : 3 * P + camera_size * (j - P)) ||
          b->delta_offset() != (j < P ? 3 * j : 3 * P + 9 * (j - P)))
        return no("parameter offsets are not those of [points | cameras]");
    }
    const std::vector<ResidualBlock*>& residual_blocks = program->residual_blocks();
    const int64_t O = int64_t(residual_blocks.size());
    std::vector<int32_t> cam(static_cast<size_t>(O)), pt(static_cast<size_t>(O));
    std::vector<double> obs(static_cast<size_t>(2 * O));
    // the loss: identified once, then every other block's object must behave like the first (bundle_adjuster news a
    // HuberLoss per residual block) -- compared on three abscissae around the loss's own scale
    const LossFunction* first_loss = O > 0 ? residual_blocks[0]->loss_function() : nullptr;
    int32_t loss_type = CX_LOSS_NONE;
    double loss_a = 0.0, loss_b = 0.0;
    if (!CxIdentifyLoss(first_loss, &loss_type, &loss_a, &loss_b)) return no("loss function is not one of the built-in losses the device evaluates");
    const double probe_scale = loss_type == CX_LOSS_NONE ? 1.0 : std::max(loss_a * loss_a, loss_b);
    double first_rho[3][3];
    if (first_loss != nullptr)
      for (int k = 0; k < 3; ++k) first_loss->Evaluate((k == 0 ? 0.3 : (k == 1 ? 1.7 : 40.0)) * probe_scale, first_rho[k]);
    for (int64_t i = 0; i < O; ++i) {
      const ResidualBlock* rb = residual_blocks[size_t(i)];
      if (quaternion) {
        const auto* cost = dynamic_cast<const SnavelyQuaternion*>(rb->cost_function());
        if (cost == nullptr) return no("cost function is not AutoDiffCostFunction<SnavelyReprojectionErrorWithQuaternions, 2, 10, 3>");
        obs[size_t(2 * i)] = cost->functor().observed_x;
        obs[size_t(2 * i + 1)] = cost->functor().observed_y;
      } else {
        const auto* cost = dynamic_cast<const Snavely*>(rb->cost_function());
        if (cost == nullptr) return no("cost function is not AutoDiffCostFunction<SnavelyReprojectionError, 2, 9, 3>");
        obs[size_t(2 * i)] = cost->functor().observed_x;
        obs[size_t(2 * i + 1)] = cost->functor().observed_y;
      }
      const LossFunction* loss = rb->loss_function();
      if (loss != first_loss) {
        if (loss == nullptr || first_loss == nullptr) return no("some residual blocks have a loss function and some have none");
        for (int k = 0; k < 3; ++k) {
          double rho[3];
          loss->Evaluate((k == 0 ? 0.3 : (k == 1 ? 1.7 : 40.0)) * probe_scale, rho);
          if (rho[0] != first_rho[k][0] || rho[1] != first_rho[k][1] || rho[2] != first_rho[k][2]) return no("residual blocks carry different loss functions");
        }
      }
      const int ci = rb->parameter_blocks()[0]->index(), pi = rb->parameter_blocks()[1]->index();
      if (ci < P || pi >= P) return no("residual block is not (camera, point)");
      cam[size_t(i)] = ci - P;
      pt[size_t(i)] = pi;
    }
    CxBalProblemView view;
    view.camera_model = quaternion ? CX_CAMERA_QUATERNION_MANIFOLD : CX_CAMERA_ANGLE_AXIS;
    view.loss_type = loss_type;
    view.loss_a = loss_a;
    view.loss_b = loss_b;
    view.num_cameras = C;
    view.num_points = P;
    view.num_observations = O;
    view.camera_index = cam.data();
    view.point_index = pt.data();
    view.observations_xy = obs.data();
    std::unique_ptr<CxBalEvaluator> e = Create(options, view, why_not);
    return std::unique_ptr<Evaluator>(e.release());
  }

  // Jacobi scaling inside the evaluation.  TrustRegionMinimizer computes jacobian_scaling_ once and calls
  // jacobian->ScaleColumns(jacobian_scaling_.data()) after EVERY Evaluate (trust_region_minimizer.cc:263-279): a second
  // pass over all of J per iteration.  With this switch on, the vector of the first ScaleColumns on this evaluator's
  // Jacobian is registered with the device evaluator (cx_evaluator_set_column_scale), later evaluations write
  // J diag(scale) themselves -- bit for bit what Evaluate + ScaleColumns leaves -- and the ScaleColumns that follows is a
  // no-op.  Opt-in, like residual aliasing: SparseMatrix::ScaleColumns promises nothing about its argument staying the
  // same array with the same contents, so the caller vouches that it does (TrustRegionMinimizer: it does) and that
  // exactly one ScaleColumns follows every Jacobian evaluation.  The gradient Evaluate returns stays unscaled.
  void set_fuse_jacobi_scaling(bool on) {
    handle_->fuse_scaling = on;
    if (!on) {
      cx_evaluator_set_column_scale(handle_->evaluator, nullptr, CX_HOST);
      handle_->registered_scale_host = nullptr;
      handle_->values_carry_registered_scale = false;
    }
    // a ScaleColumns (iteration 0) or the evaluator's own gather pass produces the camera-major copy: the evaluation
    // kernel need not scatter it
    cx_evaluator_set_emit_camera_major(handle_->evaluator, on ? 0 : 1);
  }

  // Robust loss by (type, arguments): what TryCreate recovers from the program's LossFunction objects, or what a caller
  // of Create states directly.
  bool SetLoss(int32_t loss_type, double a, double b) {
    if (cx_evaluator_set_loss(handle_->evaluator, loss_type, a, b) != CX_OK) return false;
    loss_type_ = loss_type;
    loss_a_ = a;
    loss_b_ = b;
    return true;
  }

  // ---- Evaluator
  // The layout of BlockJacobianWriter::CreateJacobian (block_jacobian_writer.cc:198-263) for this program: E cells
  // first, then F cells, row blocks in residual-block order.
  std::unique_ptr<SparseMatrix> CreateJacobian() const final {
    auto bs = std::make_unique<CompressedRowBlockStructure>();
    const int P = num_points_, C = num_cameras_;
    const int64_t O = int64_t(camera_index_.size());
    bs->cols.resize(size_t(P) + size_t(C));
    for (int j = 0; j < P; ++j) bs->cols[size_t(j)] = Block(3, 3 * j);
    for (int i = 0; i < C; ++i) bs->cols[size_t(P + i)] = Block(9, 3 * P + 9 * i);
    bs->rows.resize(size_t(O));
    for (int64_t r = 0; r < O; ++r) {
      CompressedRow& row = bs->rows[size_t(r)];
      row.block = Block(2, int(2 * r));
      row.cells.resize(2);
      row.cells[0] = Cell(point_index_[size_t(r)], int(6 * r));
      row.cells[1] = Cell(P + camera_index_[size_t(r)], int(6 * O + 18 * r));
      row.nnz = 24;
      row.cumulative_nnz = int(24 * (r + 1));
    }
    return std::make_unique<CxDeviceJacobian>(handle_, std::move(bs));
  }

  bool Evaluate(const Evaluator::EvaluateOptions& evaluate_options, const double* state, double* cost, double* residuals,
                double* gradient, SparseMatrix* jacobian) final {
    ScopedExecutionTimer total_timer("Evaluator::Total", &execution_summary_);  // program_evaluator.h:143-147
    ScopedExecutionTimer call_type_timer(gradient == nullptr && jacobian == nullptr ? "Evaluator::Residual" : "Evaluator::Jacobian",
                                         &execution_summary_);
    CxDeviceJacobian* device_jacobian = nullptr;
    if (jacobian != nullptr) {
      device_jacobian = dynamic_cast<CxDeviceJacobian*>(jacobian);
      if (device_jacobian == nullptr || device_jacobian->handle().get() != handle_.get()) return false;  // not from CreateJacobian()
    }
    (void)evaluate_options.new_evaluation_point;  // no per-point cache on the device side
    cx_evaluator* e = handle_->evaluator;
    // apply_loss_function = false: "evaluate the cost without the loss function" (evaluator.h:100-104)
    const bool suspend_loss = !evaluate_options.apply_loss_function && loss_type_ != CX_LOSS_NONE;
    if (suspend_loss && cx_evaluator_set_loss(e, CX_LOSS_NONE, 0.0, 0.0) != CX_OK) return false;
    const int rc = cx_evaluator_evaluate(e, state, cost, residuals, gradient, jacobian != nullptr ? 1 : 0, CX_HOST);
    if (suspend_loss) cx_evaluator_set_loss(e, loss_type_, loss_a_, loss_b_);
    if (rc != CX_OK) return false;
    if (device_jacobian != nullptr) {
      device_jacobian->DeviceValuesChanged();
      handle_->values_carry_registered_scale = handle_->fuse_scaling && handle_->registered_scale_host != nullptr;
    }
    if (residuals != nullptr) handle_->last_residuals_host = residuals;
    return true;
  }

  bool Plus(const double* state, const double* delta, double* state_plus_delta) const final {
    // the loop of program_evaluator.h:306-320.  All blocks Euclidean: x + delta on the host (three vectors of num_cols
    // doubles are cheaper to add here than to send across PCIe and back).  Quaternion cameras: the manifold's Plus
    // (manifold.cc:27-59) is device code, cx_evaluator_plus.
    if (camera_model_ == CX_CAMERA_QUATERNION_MANIFOLD)
      return cx_evaluator_plus(handle_->evaluator, state, delta, state_plus_delta, CX_HOST) == CX_OK;
    const int64_t n = NumParameters();
    for (int64_t i = 0; i < n; ++i) state_plus_delta[i] = state[i] + delta[i];
    return true;
  }
  int NumParameters() const final { return int(cx_evaluator_num_parameters(handle_->evaluator)); }
  int NumEffectiveParameters() const final { return int(cx_evaluator_num_effective_parameters(handle_->evaluator)); }
  int NumResiduals() const final { return int(2 * camera_index_.size()); }
  std::map<std::string, CallStatistics> Statistics() const final { return execution_summary_.statistics(); }

  int32_t loss_type() const { return loss_type_; }
  double loss_a() const { return loss_a_; }
  double loss_b() const { return loss_b_; }
  int32_t camera_model() const { return camera_model_; }
  // device time of the last k_bal_evaluate launch
  double last_kernel_ms() const { return cx_evaluator_last_kernel_ms(handle_->evaluator); }
  const std::shared_ptr<CxEvaluatorHandle>& handle() const { return handle_; }

 private:
  CxBalEvaluator(const Evaluator::Options& options, std::shared_ptr<CxEvaluatorHandle> handle, const CxBalProblemView& view)
      : options_(options), handle_(std::move(handle)), num_cameras_(view.num_cameras), num_points_(view.num_points),
        camera_index_(view.camera_index, view.camera_index + view.num_observations),
        point_index_(view.point_index, view.point_index + view.num_observations), camera_model_(view.camera_model) {}

  Evaluator::Options options_;
  std::shared_ptr<CxEvaluatorHandle> handle_;
  int32_t num_cameras_, num_points_;
  std::vector<int32_t> camera_index_, point_index_;
  int32_t camera_model_ = CX_CAMERA_ANGLE_AXIS;
  int32_t loss_type_ = CX_LOSS_NONE;
  double loss_a_ = 0.0, loss_b_ = 0.0;
  ExecutionSummary execution_summary_;
};

}  // namespace ceres::internal
#endif
