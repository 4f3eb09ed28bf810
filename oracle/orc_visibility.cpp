// ORACLE -- TEST INFRASTRUCTURE ONLY (see orc_common.h).
// Visibility based preconditioning (CLUSTER_JACOBI / CLUSTER_TRIDIAGONAL), restating
//   visibility.cc:50-146                       ComputeVisibility, CreateSchurComplementGraph
//   canonical_views_clustering.cc:94-222       greedy canonical views
//   single_linkage_clustering.cc:42-92         single linkage
//   graph_algorithms.h:259-339                 Degree2MaximumSpanningForest
//   visibility_based_preconditioner.cc:121-575 cluster pairs, block pairs, membership flattening
//
// Where the reference's outcome depends on the iteration order of std::unordered_set / unordered_map
// (which view wins an exact tie of the quality difference, the order neighbours are summed in, the
// contiguous cluster numbers handed out by FlattenMembershipMap and, through them, ties between equal
// integer weights in the spanning forest) the reference does not define a result -- it changes with the
// standard library.  This restatement fixes those choices: candidates and neighbours in ascending id,
// clusters numbered by first appearance in ascending camera id.  Everything else is the reference's
// arithmetic; pinned by the known answers of canonical_views_clustering_test.cc, single_linkage_clustering_
// test.cc, graph_algorithms_test.cc and visibility_test.cc (tests/test_oracle_visibility.py).
#include "orc_visibility.h"

#include <algorithm>
#include <cmath>
#include <limits>
#include <map>
#include <set>

namespace orc {

// visibility.cc:50-75
std::vector<std::vector<int>> ComputeVisibility(const BS& bs, int num_eliminate_blocks) {
  std::vector<std::set<int>> sets(size_t(bs.C - num_eliminate_blocks));
  for (int r = 0; r < bs.R; ++r) {
    const int first = bs.rcb[r];
    if (first == bs.rcb[r + 1]) continue;
    const int block_id = bs.cells[first].block_id;
    if (block_id >= num_eliminate_blocks) continue;
    for (int j = first + 1; j < bs.rcb[r + 1]; ++j) sets[size_t(bs.cells[j].block_id - num_eliminate_blocks)].insert(block_id);
  }
  std::vector<std::vector<int>> visibility;
  for (const auto& s : sets) visibility.emplace_back(s.begin(), s.end());
  return visibility;
}

void WGraph::AddEdge(int u, int v, double w) {
  adj[size_t(u)][v] = w;
  adj[size_t(v)][u] = w;
}
double WGraph::EdgeWeight(int u, int v) const {
  auto it = adj[size_t(u)].find(v);
  return it == adj[size_t(u)].end() ? 0.0 : it->second;
}

// visibility.cc:77-146.  sum (optional): the rows of the structure are one shard of the points; the number of
// points a camera sees and the number two cameras share are then sums over the shards (every point lives in
// exactly one), taken on dense arrays before the weights are formed.
WGraph SchurComplementGraph(const std::vector<std::vector<int>>& visibility, const SumFn* sum) {
  const int n = int(visibility.size());
  int num_points = 0;
  for (const auto& v : visibility)
    if (!v.empty()) num_points = std::max(num_points, v.back() + 1);
  std::vector<std::vector<int>> inverse(static_cast<size_t>(num_points));
  for (size_t i = 0; i < visibility.size(); ++i)
    for (int p : visibility[i]) inverse[size_t(p)].push_back(int(i));
  std::map<std::pair<int, int>, int> camera_pairs;
  for (const auto& cams : inverse)
    for (size_t a = 0; a < cams.size(); ++a)
      for (size_t b = a + 1; b < cams.size(); ++b) ++camera_pairs[{cams[a], cams[b]}];
  std::vector<size_t> size(static_cast<size_t>(n));
  for (int i = 0; i < n; ++i) size[size_t(i)] = visibility[size_t(i)].size();
  if (sum) {
    std::vector<double> dense(size_t(n) * n + size_t(n), 0.0);
    for (const auto& pc : camera_pairs) dense[size_t(pc.first.first) * n + pc.first.second] = double(pc.second);
    for (int i = 0; i < n; ++i) dense[size_t(n) * n + i] = double(size[size_t(i)]);
    (*sum)(dense.data(), int64_t(dense.size()));
    camera_pairs.clear();
    for (int i = 0; i < n; ++i) {
      size[size_t(i)] = size_t(dense[size_t(n) * n + i]);
      for (int j = i + 1; j < n; ++j)
        if (dense[size_t(i) * n + j] > 0.0) camera_pairs[{i, j}] = int(dense[size_t(i) * n + j]);
    }
  }
  WGraph g(n);
  for (int i = 0; i < g.n; ++i) g.AddEdge(i, i, 1.0);  // kSelfEdgeWeight
  for (const auto& pc : camera_pairs) {
    const int c1 = pc.first.first, c2 = pc.first.second;
    const double weight = static_cast<double>(pc.second) / (std::sqrt(static_cast<double>(size[size_t(c1)] * size[size_t(c2)])));
    g.AddEdge(c1, c2, weight);
  }
  return g;
}

// canonical_views_clustering.cc:94-222; membership[v] = index of v's centre in `centers`, or -1
void CanonicalViews(const CanonicalViewsOptions& options, const WGraph& graph, std::vector<int>* centers,
                    std::vector<int>* membership) {
  const int n = graph.n;
  centers->clear();
  std::vector<char> valid(size_t(n), 1);  // FindValidViews: a weight compares unequal to NaN, always
  int num_valid = n;
  std::vector<int> view_to_canonical(size_t(n), -1);
  std::vector<double> similarity(size_t(n), 0.0);  // FindWithDefault(..., 0.0)
  while (num_valid > 0) {
    double best_difference = -std::numeric_limits<double>::max();
    int best_view = 0;
    for (int view = 0; view < n; ++view) {
      if (!valid[size_t(view)]) continue;
      // ComputeClusteringQualityDifference :152-181
      double difference = options.view_score_weight * graph.vertex_weight[size_t(view)];
      for (const auto& nb : graph.adj[size_t(view)]) {
        const double old_similarity = similarity[size_t(nb.first)];
        const double new_similarity = nb.second;
        if (new_similarity > old_similarity) difference += new_similarity - old_similarity;
      }
      difference -= options.size_penalty_weight;
      for (int center : *centers) difference -= options.similarity_penalty_weight * graph.EdgeWeight(center, view);
      if (difference > best_difference) {
        best_difference = difference;
        best_view = view;
      }
    }
    if (best_difference <= 0 && int(centers->size()) >= options.min_views) break;
    centers->push_back(best_view);
    valid[size_t(best_view)] = 0;
    --num_valid;
    // UpdateCanonicalViewAssignments :184-196
    for (const auto& nb : graph.adj[size_t(best_view)]) {
      if (nb.second > similarity[size_t(nb.first)]) {
        view_to_canonical[size_t(nb.first)] = best_view;
        similarity[size_t(nb.first)] = nb.second;
      }
    }
  }
  // ComputeClusterMembership :199-222
  std::map<int, int> center_to_cluster_id;
  for (size_t i = 0; i < centers->size(); ++i) center_to_cluster_id[(*centers)[i]] = int(i);
  membership->assign(size_t(n), -1);
  for (int v = 0; v < n; ++v)
    if (view_to_canonical[size_t(v)] >= 0) (*membership)[size_t(v)] = center_to_cluster_id.at(view_to_canonical[size_t(v)]);
}

static int Find(std::vector<int>& parent, int v) {  // graph_algorithms.h:229-239
  if (parent[size_t(v)] != v) parent[size_t(v)] = Find(parent, parent[size_t(v)]);
  return parent[size_t(v)];
}

// single_linkage_clustering.cc:42-92; membership[v] = smallest vertex of v's cluster
int SingleLinkage(double min_similarity, const WGraph& graph, std::vector<int>* membership) {
  const int n = graph.n;
  membership->resize(size_t(n));
  for (int v = 0; v < n; ++v) (*membership)[size_t(v)] = v;
  for (int v1 = 0; v1 < n; ++v1)
    for (const auto& nb : graph.adj[size_t(v1)]) {
      const int v2 = nb.first;
      if (v1 > v2 || nb.second < min_similarity) continue;
      const int c1 = Find(*membership, v1), c2 = Find(*membership, v2);
      if (c1 == c2) continue;
      if (c1 < c2) (*membership)[size_t(c2)] = c1; else (*membership)[size_t(c1)] = c2;
    }
  int num_clusters = 0;
  for (int v = 0; v < n; ++v) {
    (*membership)[size_t(v)] = Find(*membership, v);
    if ((*membership)[size_t(v)] == v) ++num_clusters;
  }
  return num_clusters;
}

// graph_algorithms.h:259-339; returns the forest's edges (u < v) in the order they were accepted
std::vector<std::pair<int, int>> Degree2MaximumSpanningForest(const WGraph& graph) {
  std::vector<std::pair<double, std::pair<int, int>>> weighted_edges;
  for (int v1 = 0; v1 < graph.n; ++v1)
    for (const auto& nb : graph.adj[size_t(v1)])
      if (v1 < nb.first) weighted_edges.push_back({nb.second, {v1, nb.first}});
  std::sort(weighted_edges.rbegin(), weighted_edges.rend());
  std::vector<int> disjoint(static_cast<size_t>(graph.n)), degree(size_t(graph.n), 0);
  for (int v = 0; v < graph.n; ++v) disjoint[size_t(v)] = v;
  std::vector<std::pair<int, int>> forest;
  for (const auto& e : weighted_edges) {
    const int v1 = e.second.first, v2 = e.second.second;
    if (degree[size_t(v1)] == 2 || degree[size_t(v2)] == 2) continue;
    int root1 = Find(disjoint, v1), root2 = Find(disjoint, v2);
    if (root1 == root2) continue;
    forest.push_back({v1, v2});
    ++degree[size_t(v1)];
    ++degree[size_t(v2)];
    if (root2 < root1) std::swap(root1, root2);
    disjoint[size_t(root2)] = root1;
  }
  return forest;
}

// visibility_based_preconditioner.cc:121-201, 482-575
VisibilityStructure ComputeVisibilityStructure(const cx_block_structure* s, int num_eliminate_blocks, int preconditioner_type,
                                               int clustering_type, const SumFn* sum) {
  BS bs(s);
  VisibilityStructure out;
  const auto visibility = ComputeVisibility(bs, num_eliminate_blocks);
  const int num_blocks = int(visibility.size());
  const WGraph graph = SchurComplementGraph(visibility, sum);
  // ClusterCameras :173-201
  std::vector<int> raw;
  int num_clusters = 0;
  if (clustering_type == CX_CANONICAL_VIEWS) {
    CanonicalViewsOptions o;
    o.size_penalty_weight = 3.0;        // kCanonicalViewsSizePenaltyWeight
    o.similarity_penalty_weight = 0.0;  // kCanonicalViewsSimilarityPenaltyWeight
    std::vector<int> centers;
    CanonicalViews(o, graph, &centers, &raw);
    num_clusters = int(centers.size());
  } else {
    num_clusters = SingleLinkage(0.9 /* kSingleLinkageMinSimilarity */, graph, &raw);
  }
  // FlattenMembershipMap :539-575 (cluster numbers by first appearance in ascending camera id, see the header)
  out.num_clusters = num_clusters;
  out.membership.assign(size_t(num_blocks), -1);
  std::map<int, int> cluster_id_to_index;
  for (int camera = 0; camera < num_blocks; ++camera) {
    int cluster_id = raw[size_t(camera)];
    if (cluster_id == -1) cluster_id = camera % num_clusters;
    auto it = cluster_id_to_index.find(cluster_id);
    if (it == cluster_id_to_index.end()) it = cluster_id_to_index.emplace(cluster_id, int(cluster_id_to_index.size())).first;
    out.membership[size_t(camera)] = it->second;
  }
  std::set<std::pair<int, int>> cluster_pairs;
  for (int i = 0; i < num_clusters; ++i) cluster_pairs.emplace(i, i);
  if (preconditioner_type == CX_CLUSTER_TRIDIAGONAL) {
    // ComputeClusterVisibility + CreateClusterGraph :482-529
    std::vector<std::set<int>> cluster_visibility(static_cast<size_t>(num_clusters));
    for (int i = 0; i < num_blocks; ++i)
      cluster_visibility[size_t(out.membership[size_t(i)])].insert(visibility[size_t(i)].begin(), visibility[size_t(i)].end());
    WGraph cluster_graph(num_clusters);
    std::vector<double> shared(size_t(num_clusters) * num_clusters, 0.0);
    for (int i = 0; i < num_clusters; ++i)
      for (int j = i + 1; j < num_clusters; ++j) {
        std::vector<int> intersection;
        std::set_intersection(cluster_visibility[size_t(i)].begin(), cluster_visibility[size_t(i)].end(),
                              cluster_visibility[size_t(j)].begin(), cluster_visibility[size_t(j)].end(),
                              std::back_inserter(intersection));
        shared[size_t(i) * num_clusters + j] = double(intersection.size());
      }
    if (sum) (*sum)(shared.data(), int64_t(shared.size()));  // points of the other shards
    for (int i = 0; i < num_clusters; ++i)
      for (int j = i + 1; j < num_clusters; ++j)
        if (shared[size_t(i) * num_clusters + j] > 0.0) cluster_graph.AddEdge(i, j, shared[size_t(i) * num_clusters + j]);
    // ForestToClusterPairs :458-477
    for (const auto& e : Degree2MaximumSpanningForest(cluster_graph)) cluster_pairs.emplace(e.first, e.second);
  }
  out.cluster_pairs.assign(cluster_pairs.begin(), cluster_pairs.end());
  // ComputeBlockPairsInPreconditioner :223-302
  auto in_preconditioner = [&](int b1, int b2) {
    int c1 = out.membership[size_t(b1)], c2 = out.membership[size_t(b2)];
    if (c1 > c2) std::swap(c1, c2);
    return cluster_pairs.count({c1, c2}) > 0;
  };
  std::set<std::pair<int, int>> block_pairs;
  for (int i = 0; i < num_blocks; ++i) block_pairs.emplace(i, i);
  int r = 0;
  while (r < bs.R) {
    const int e_block_id = bs.cells[bs.rcb[r]].block_id;
    if (e_block_id >= num_eliminate_blocks) break;
    std::set<int> f_blocks;
    for (; r < bs.R; ++r) {
      if (bs.cells[bs.rcb[r]].block_id != e_block_id) break;
      for (int c = bs.rcb[r] + 1; c < bs.rcb[r + 1]; ++c) f_blocks.insert(bs.cells[c].block_id - num_eliminate_blocks);
    }
    for (auto b1 = f_blocks.begin(); b1 != f_blocks.end(); ++b1) {
      auto b2 = b1;
      for (++b2; b2 != f_blocks.end(); ++b2)
        if (in_preconditioner(*b1, *b2)) block_pairs.emplace(*b1, *b2);
    }
  }
  for (; r < bs.R; ++r)
    for (int i = bs.rcb[r]; i < bs.rcb[r + 1]; ++i) {
      const int b1 = bs.cells[i].block_id - num_eliminate_blocks;
      for (int j = bs.rcb[r]; j < bs.rcb[r + 1]; ++j) {
        const int b2 = bs.cells[j].block_id - num_eliminate_blocks;
        if (b1 <= b2 && in_preconditioner(b1, b2)) block_pairs.emplace(b1, b2);
      }
    }
  out.block_pairs.assign(block_pairs.begin(), block_pairs.end());
  return out;
}

}  // namespace orc

using namespace orc;

static WGraph GraphFromEdges(int n, const double* vertex_weights, int num_edges, const int32_t* u, const int32_t* v, const double* w) {
  WGraph g(n);
  if (vertex_weights) g.vertex_weight.assign(vertex_weights, vertex_weights + n);
  for (int k = 0; k < num_edges; ++k) g.AddEdge(u[k], v[k], w[k]);
  return g;
}

extern "C" {

int orc_schur_complement_graph(const cx_block_structure* bs, int num_eliminate_blocks, int32_t* u, int32_t* v, double* w,
                               int capacity) {
  BS b(bs);
  const WGraph g = SchurComplementGraph(ComputeVisibility(b, num_eliminate_blocks), nullptr);
  int count = 0;
  for (int i = 0; i < g.n; ++i)
    for (const auto& nb : g.adj[size_t(i)]) {
      if (nb.first < i) continue;  // self edges included, as in the reference's graph
      if (count < capacity) { u[count] = i; v[count] = nb.first; w[count] = nb.second; }
      ++count;
    }
  return count;
}

int orc_canonical_views(int n, const double* vertex_weights, int num_edges, const int32_t* u, const int32_t* v,
                        const double* w, int min_views, double size_penalty_weight, double similarity_penalty_weight,
                        double view_score_weight, int32_t* centers, int32_t* membership) {
  CanonicalViewsOptions o;
  o.min_views = min_views;
  o.size_penalty_weight = size_penalty_weight;
  o.similarity_penalty_weight = similarity_penalty_weight;
  o.view_score_weight = view_score_weight;
  std::vector<int> c, m;
  CanonicalViews(o, GraphFromEdges(n, vertex_weights, num_edges, u, v, w), &c, &m);
  std::copy(c.begin(), c.end(), centers);
  std::copy(m.begin(), m.end(), membership);
  return int(c.size());
}

int orc_single_linkage(int n, int num_edges, const int32_t* u, const int32_t* v, const double* w, double min_similarity,
                       int32_t* membership) {
  std::vector<int> m;
  const int k = SingleLinkage(min_similarity, GraphFromEdges(n, nullptr, num_edges, u, v, w), &m);
  std::copy(m.begin(), m.end(), membership);
  return k;
}

int orc_degree2_forest(int n, int num_edges, const int32_t* u, const int32_t* v, const double* w, int32_t* fu, int32_t* fv) {
  const auto forest = Degree2MaximumSpanningForest(GraphFromEdges(n, nullptr, num_edges, u, v, w));
  for (size_t k = 0; k < forest.size(); ++k) { fu[k] = forest[k].first; fv[k] = forest[k].second; }
  return int(forest.size());
}

int64_t orc_visibility_structure(const cx_block_structure* bs, int num_eliminate_blocks, int preconditioner_type,
                                 int clustering_type, int32_t* membership, int32_t* num_clusters, int32_t* num_cluster_pairs,
                                 int32_t* cluster_pair_1, int32_t* cluster_pair_2, int32_t cluster_pair_capacity,
                                 int32_t* block_pair_1, int32_t* block_pair_2, int64_t block_pair_capacity) {
  const VisibilityStructure vs = ComputeVisibilityStructure(bs, num_eliminate_blocks, preconditioner_type, clustering_type, nullptr);
  if (membership) std::copy(vs.membership.begin(), vs.membership.end(), membership);
  if (num_clusters) *num_clusters = vs.num_clusters;
  if (num_cluster_pairs) *num_cluster_pairs = int32_t(vs.cluster_pairs.size());
  for (size_t k = 0; k < vs.cluster_pairs.size() && int32_t(k) < cluster_pair_capacity; ++k) {
    cluster_pair_1[k] = vs.cluster_pairs[k].first;
    cluster_pair_2[k] = vs.cluster_pairs[k].second;
  }
  for (size_t k = 0; k < vs.block_pairs.size() && int64_t(k) < block_pair_capacity; ++k) {
    block_pair_1[k] = vs.block_pairs[k].first;
    block_pair_2[k] = vs.block_pairs[k].second;
  }
  return int64_t(vs.block_pairs.size());
}

}  // extern "C"
