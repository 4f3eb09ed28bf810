// CxDeviceJacobian: the SparseMatrix that CxBalEvaluator::CreateJacobian returns -- the block-sparse Jacobian of a
// bundle-adjustment program whose values live in HBM and never cross PCIe during a minimisation.
//
// BlockSparseMatrix is `final` in the reference (block_sparse_matrix.h:60), so this is a sibling implementation of the
// abstract SparseMatrix (sparse_matrix.h:66-113), the type TrustRegionMinimizer and LevenbergMarquardtStrategy hold
// (trust_region_minimizer.h:151, levenberg_marquardt_strategy.cc:69-73).  Everything those two callers do with a
// Jacobian goes through SparseMatrix virtuals and runs on the device here:
//   SquaredColumnNorm   trust_region_minimizer.cc:270, levenberg_marquardt_strategy.cc:85   -> k_cam_sqnorm / k_sqnorm_e_239
//   ScaleColumns        trust_region_minimizer.cc:277-278                                    -> k_scale_239
//   RightMultiplyAndAccumulate (model cost change) trust_region_minimizer.cc:430-433         -> k_right_239
//   LinearSolver::Solve(jacobian, ...)  levenberg_marquardt_strategy.cc:113                  -> CxLinearSolver, no upload
// Only vectors (num_cols / num_rows doubles) travel.  values() / mutable_values() exist for the components that want
// host data (ToTextFile dumps, gradient checking): the host copy is materialised on first request and, after
// mutable_values(), written back before the next device use.
#ifndef CX_DEVICE_JACOBIAN_H_
#define CX_DEVICE_JACOBIAN_H_

#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "../../include/cxschur.h"
#ifdef CX_USE_CERES_HEADERS
#include "ceres/block_structure.h"
#include "ceres/sparse_matrix.h"
#else
#include "ceres_mirror.h"
#endif

namespace ceres::internal {

// The reference aborts on programming errors (CHECK); device / library failures inside a void SparseMatrix virtual
// have no other channel either.
#define CX_ADAPTER_CHECK(call)                                                              \
  do {                                                                                      \
    if ((call) != CX_OK) {                                                                  \
      std::fprintf(stderr, "cxschur: %s failed: %s\n", #call, cx_last_error());             \
      std::abort();                                                                         \
    }                                                                                       \
  } while (0)

// Shared ownership of the device evaluator (and through it of the device matrix): the Jacobian handed to the
// minimizer (Minimizer::Options::jacobian, a shared_ptr, trust_region_preprocessor.cc:348) may outlive the evaluator.
struct CxEvaluatorHandle {
  cx_context* ctx = nullptr;
  cx_evaluator* evaluator = nullptr;
  // the host array the evaluator last wrote residuals to; their device copy is cx_evaluator_device_residuals()
  const double* last_residuals_host = nullptr;
  // Jacobi scaling folded into the evaluation (CxBalEvaluator::set_fuse_jacobi_scaling): the vector of the first
  // ScaleColumns is registered with the device evaluator, which from then on writes J diag(scale) itself; the
  // ScaleColumns call that follows such an evaluation finds its work done
  bool fuse_scaling = false;
  const double* registered_scale_host = nullptr;
  bool values_carry_registered_scale = false;
  // three entries of the registered vector (first, middle, last) as they were when it was registered: the cheap check
  // that the array the caller passes again still holds what the evaluator applies (pointer identity alone would miss a
  // caller that refills the same array)
  double registered_scale_probe[3] = {0.0, 0.0, 0.0};
  int64_t registered_scale_size = 0;
  void RememberScale(const double* scale, int64_t n) {
    registered_scale_host = scale;
    registered_scale_size = n;
    if (n > 0) {
      registered_scale_probe[0] = scale[0];
      registered_scale_probe[1] = scale[n / 2];
      registered_scale_probe[2] = scale[n - 1];
    }
  }
  bool IsRegisteredScale(const double* scale) const {
    const int64_t n = registered_scale_size;
    return scale == registered_scale_host && scale != nullptr &&
           (n == 0 || (scale[0] == registered_scale_probe[0] && scale[n / 2] == registered_scale_probe[1] && scale[n - 1] == registered_scale_probe[2]));
  }
  ~CxEvaluatorHandle() {
    if (evaluator) cx_evaluator_destroy(evaluator);
    // the minimizer's vectors this evaluator's calls were handed (registered with the HIP runtime on first sight,
    // cx_host_registration_policy) are about to be freed by their owner
    cx_host_registrations_release();
  }
};

class CxDeviceJacobian final : public SparseMatrix {
 public:
  // block_structure: the layout BlockJacobianWriter::CreateJacobian would have produced (block_jacobian_writer.cc:
  // 198-263); kept for callers that inspect it and for the host-side dumps.
  CxDeviceJacobian(std::shared_ptr<CxEvaluatorHandle> handle, std::unique_ptr<CompressedRowBlockStructure> block_structure)
      : handle_(std::move(handle)), matrix_(cx_evaluator_jacobian(handle_->evaluator)), block_structure_(std::move(block_structure)) {}

  cx_matrix* device_matrix() {
    FlushHostValues();
    return matrix_;
  }
  const std::shared_ptr<CxEvaluatorHandle>& handle() const { return handle_; }
  const CompressedRowBlockStructure* block_structure() const { return block_structure_.get(); }
  // the evaluator has just rewritten the device values: a host copy made earlier is stale
  void DeviceValuesChanged() { host_valid_ = host_dirty_ = false; }
  // how often the values were copied to / from host memory (0 over a whole minimisation is the point of this class)
  int num_downloads() const { return num_downloads_; }
  int num_uploads() const { return num_uploads_; }

  // ---- SparseMatrix / LinearOperator (sparse_matrix.h:66-113)
  void SetZero() final {
    host_valid_ = host_dirty_ = false;
    CX_ADAPTER_CHECK(cx_matrix_set_zero(matrix_));
  }
  // When the caller vouches that y is zero whenever it calls RightMultiplyAndAccumulate -- TrustRegionMinimizer does:
  // model_residuals_.setZero() immediately before the only product it takes (trust_region_minimizer.cc:430-433) -- the
  // num_rows zeros are not uploaded (464 MB on Final-13682): y = J x is computed into a cleared device vector and
  // copied back.  Opt-in like the residual aliasing; a few entries of y are sampled anyway, and a non-zero one sends the
  // call down the accumulating path.
  void set_assume_zeroed_product_target(bool on) { assume_zeroed_product_target_ = on; }
  void RightMultiplyAndAccumulate(const double* x, double* y) const final {
    Flush();
    if (assume_zeroed_product_target_ && LooksZero(y, num_rows())) {
      CX_ADAPTER_CHECK(cx_matrix_right_multiply_overwrite(matrix_, x, y, CX_HOST));
      return;
    }
    CX_ADAPTER_CHECK(cx_matrix_right_multiply(matrix_, x, y, CX_HOST));
  }
  void LeftMultiplyAndAccumulate(const double* x, double* y) const final {
    Flush();
    CX_ADAPTER_CHECK(cx_matrix_left_multiply(matrix_, x, y, CX_HOST));
  }
  void SquaredColumnNorm(double* x) const final {
    Flush();
    CX_ADAPTER_CHECK(cx_matrix_squared_column_norm(matrix_, x, CX_HOST));
  }
  void ScaleColumns(const double* scale) final {
    Flush();
    if (handle_->fuse_scaling && handle_->evaluator != nullptr) {
      if (handle_->values_carry_registered_scale && handle_->IsRegisteredScale(scale)) {
        handle_->values_carry_registered_scale = false;  // the evaluation already wrote J diag(scale): this call's work is done
        return;
      }
      if (handle_->registered_scale_host == nullptr) {  // the first call (iteration 0): scale now, and tell the evaluator
        host_valid_ = false;
        CX_ADAPTER_CHECK(cx_matrix_scale_columns(matrix_, scale, CX_HOST));
        CX_ADAPTER_CHECK(cx_evaluator_set_column_scale(handle_->evaluator, scale, CX_HOST));
        handle_->RememberScale(scale, num_cols());
        return;
      }
      if (handle_->values_carry_registered_scale) {
        // the values already carry the registered scale and the caller now brings ANOTHER vector (or new contents):
        // the promise behind set_fuse_jacobi_scaling is broken and the product of both scalings is not what was asked for
        std::fprintf(stderr, "cxschur: ScaleColumns with a vector other than the one registered by set_fuse_jacobi_scaling\n");
        std::abort();
      }
    }
    host_valid_ = false;
    CX_ADAPTER_CHECK(cx_matrix_scale_columns(matrix_, scale, CX_HOST));
  }
  void ToDenseMatrix(Matrix* dense_matrix) const final {
    dense_matrix->resize(num_rows(), num_cols());
    dense_matrix->setZero();
    ForEachEntry([&](int r, int c, double v) { (*dense_matrix)(r, c) = v; });
  }
  void ToTextFile(FILE* file) const final {  // block_sparse_matrix.cc:625-650
    ForEachEntry([&](int r, int c, double v) { std::fprintf(file, "% 10d % 10d %17f\n", r, c, v); });
  }
  double* mutable_values() final {
    Materialise();
    host_dirty_ = true;
    return host_values_.data();
  }
  const double* values() const final {
    Materialise();
    return host_values_.data();
  }
  int num_rows() const final { return int(cx_matrix_num_rows(matrix_)); }
  int num_cols() const final { return int(cx_matrix_num_cols(matrix_)); }
  int num_nonzeros() const final { return int(cx_matrix_num_nonzeros(matrix_)); }

 private:
  void Materialise() const {
    if (host_valid_) return;
    host_values_.resize(size_t(cx_matrix_num_nonzeros(matrix_)));
    CX_ADAPTER_CHECK(cx_matrix_get_values(matrix_, host_values_.data()));
    host_valid_ = true;
    ++num_downloads_;
  }
  void FlushHostValues() {
    if (!host_dirty_) return;
    CX_ADAPTER_CHECK(cx_matrix_set_values(matrix_, host_values_.data(), CX_HOST));
    host_dirty_ = false;
    ++num_uploads_;
  }
  void Flush() const { const_cast<CxDeviceJacobian*>(this)->FlushHostValues(); }
  template <typename Fn>
  void ForEachEntry(Fn fn) const {
    Flush();
    Materialise();
    const CompressedRowBlockStructure* bs = block_structure_.get();
    for (const CompressedRow& row : bs->rows)
      for (const Cell& cell : row.cells) {
        const Block& col = bs->cols[cell.block_id];
        for (int i = 0; i < row.block.size; ++i)
          for (int j = 0; j < col.size; ++j)
            fn(row.block.position + i, col.position + j, host_values_[size_t(cell.position + i * col.size + j)]);
      }
  }

  static bool LooksZero(const double* y, int64_t n) {
    const int64_t stride = n > 64 ? n / 64 : 1;
    for (int64_t i = 0; i < n; i += stride)
      if (y[i] != 0.0) return false;
    return n == 0 || y[n - 1] == 0.0;
  }

  std::shared_ptr<CxEvaluatorHandle> handle_;
  bool assume_zeroed_product_target_ = false;
  cx_matrix* matrix_;  // owned by the evaluator behind handle_
  std::unique_ptr<CompressedRowBlockStructure> block_structure_;
  mutable std::vector<double> host_values_;
  mutable bool host_valid_ = false;
  bool host_dirty_ = false;
  mutable int num_downloads_ = 0;
  int num_uploads_ = 0;
};

}  // namespace ceres::internal
#endif
