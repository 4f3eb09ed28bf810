// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_common.h).
// Visibility clustering and the structure of the CLUSTER_JACOBI / CLUSTER_TRIDIAGONAL preconditioners
// (orc_visibility.cpp).
#ifndef ORC_VISIBILITY_H_
#define ORC_VISIBILITY_H_

#include <functional>
#include <map>
#include <utility>
#include <vector>

#include "orc_api.h"
#include "orc_common.h"

namespace orc {

// graph.h:104-206 (WeightedGraph<int>) over vertices 0..n-1; neighbours in ascending id
struct WGraph {
  int n;
  std::vector<double> vertex_weight;
  std::vector<std::map<int, double>> adj;
  explicit WGraph(int n_) : n(n_), vertex_weight(size_t(n_), 1.0), adj(size_t(n_)) {}
  void AddEdge(int u, int v, double w);
  double EdgeWeight(int u, int v) const;
};

struct CanonicalViewsOptions {  // canonical_views_clustering.h:104-121
  int min_views = 3;
  double size_penalty_weight = 5.75;
  double similarity_penalty_weight = 100;
  double view_score_weight = 0.0;
};

std::vector<std::vector<int>> ComputeVisibility(const BS& bs, int num_eliminate_blocks);
using SumFn = std::function<void(double*, int64_t)>;  // in-place sum over the shards of a sharded solve
WGraph SchurComplementGraph(const std::vector<std::vector<int>>& visibility, const SumFn* sum);
void CanonicalViews(const CanonicalViewsOptions& options, const WGraph& graph, std::vector<int>* centers,
                    std::vector<int>* membership);
int SingleLinkage(double min_similarity, const WGraph& graph, std::vector<int>* membership);
std::vector<std::pair<int, int>> Degree2MaximumSpanningForest(const WGraph& graph);

struct VisibilityStructure {
  int num_clusters = 0;
  std::vector<int> membership;                      // camera -> cluster
  std::vector<std::pair<int, int>> cluster_pairs;   // (c1 <= c2), lexicographic
  std::vector<std::pair<int, int>> block_pairs;     // f-block pairs (b1 <= b2) of the preconditioner, lexicographic
};
VisibilityStructure ComputeVisibilityStructure(const cx_block_structure* s, int num_eliminate_blocks, int preconditioner_type,
                                               int clustering_type, const SumFn* sum);

}  // namespace orc
#endif
