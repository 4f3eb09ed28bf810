// Structure of the visibility based preconditioners, host side, once per matrix structure:
//   visibility + Schur complement graph   visibility.cc:50-146
//   clustering                            canonical_views_clustering.cc:94-222 (size penalty 3, similarity penalty 0,
//                                         min_views 3: visibility_based_preconditioner.cc:64-66, 180-188) or
//                                         single_linkage_clustering.cc:42-92 (min similarity 0.9)
//   cluster pairs                         CLUSTER_JACOBI: the diagonal; CLUSTER_TRIDIAGONAL: edges of the degree-2
//                                         maximum spanning forest of the cluster graph (graph_algorithms.h:259-339,
//                                         visibility_based_preconditioner.cc:139-157, 458-529)
//   block pairs                           the cells of S whose cluster pair is in that set (:223-302)
// and then this library's own part: the camera order and band layout that make the factorisation of the
// preconditioner a banded Cholesky (cx_visibility.h).
//
// Flat arrays instead of the reference's hash containers.  Where the reference leaves a choice to the iteration
// order of std::unordered_set / unordered_map (exact ties between candidate views, summation order over
// neighbours, the contiguous cluster numbers of FlattenMembershipMap and with them ties between equal integer
// weights in the forest) this code takes candidates and neighbours in ascending id and numbers clusters by first
// appearance in ascending camera id -- the same rule as the oracle, stated in DESIGN.md.
#include "cx_visibility.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <limits>
#include <numeric>
#include <thread>
#include <unordered_map>
#include <array>
#include <cstdio>
#include <cstdlib>
#include <memory>

#include "cx_schur.h"

namespace {

struct Csr {
  std::vector<int64_t> start;
  std::vector<int32_t> idx;
};

// WeightedGraph<int> over cameras: symmetric, neighbours ascending, self edge (weight 1) included
struct Graph {
  int n = 0;
  std::vector<int64_t> start;
  std::vector<int32_t> nb;
  std::vector<double> w;
};

// ComputeVisibility: points of every camera and cameras of every point, both ascending and distinct
// (cells: two per row block, point cell then camera cell, rows sorted by point -- the static <2,3,9> layout)
void Visibility(const cx_cell* cells, int C, int P, int64_t O, Csr* cam_pts, Csr* pt_cams, int first_camera_block = -1) {
  // (first_camera_block: column block of camera 0 when it is not P -- an embedding's dummy points sit between)
  if (first_camera_block < 0) first_camera_block = P;
  pt_cams->start.assign(size_t(P) + 1, 0);
  pt_cams->idx.clear();
  pt_cams->idx.reserve(size_t(O));
  std::vector<int32_t> tmp;
  int64_t r = 0;
  for (int p = 0; p < P; ++p) {
    tmp.clear();
    while (r < O && cells[size_t(2 * r)].block_id == p) {
      tmp.push_back(cells[size_t(2 * r + 1)].block_id - first_camera_block);
      ++r;
    }
    std::sort(tmp.begin(), tmp.end());
    tmp.erase(std::unique(tmp.begin(), tmp.end()), tmp.end());
    pt_cams->idx.insert(pt_cams->idx.end(), tmp.begin(), tmp.end());
    pt_cams->start[size_t(p) + 1] = int64_t(pt_cams->idx.size());
  }
  cam_pts->start.assign(size_t(C) + 1, 0);
  for (int32_t c : pt_cams->idx) cam_pts->start[size_t(c) + 1]++;
  for (int c = 0; c < C; ++c) cam_pts->start[size_t(c) + 1] += cam_pts->start[size_t(c)];
  cam_pts->idx.resize(pt_cams->idx.size());
  std::vector<int64_t> fill(cam_pts->start.begin(), cam_pts->start.end() - 1);
  for (int p = 0; p < P; ++p)
    for (int64_t k = pt_cams->start[size_t(p)]; k < pt_cams->start[size_t(p) + 1]; ++k)
      cam_pts->idx[size_t(fill[size_t(pt_cams->idx[size_t(k)])]++)] = p;
}

using PairCounts = std::vector<std::vector<std::pair<int32_t, int32_t>>>;  // per camera c1: (c2 > c1, shared points), ascending c2

// points shared by every pair of cameras
PairCounts SharedPointCounts(int C, const Csr& cam_pts, const Csr& pt_cams) {
  PairCounts upper(static_cast<size_t>(C));
  const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  std::vector<std::thread> workers;
  for (unsigned t = 0; t < hw; ++t)
    workers.emplace_back([&, t]() {
      std::vector<int32_t> count(size_t(C), 0), touched;
      for (int c1 = int(t); c1 < C; c1 += int(hw)) {
        touched.clear();
        for (int64_t k = cam_pts.start[size_t(c1)]; k < cam_pts.start[size_t(c1) + 1]; ++k) {
          const int32_t p = cam_pts.idx[size_t(k)];
          for (int64_t q = pt_cams.start[size_t(p)]; q < pt_cams.start[size_t(p) + 1]; ++q) {
            const int32_t c2 = pt_cams.idx[size_t(q)];
            if (c2 > c1 && count[size_t(c2)]++ == 0) touched.push_back(c2);
          }
        }
        std::sort(touched.begin(), touched.end());
        for (int32_t c2 : touched) {
          upper[size_t(c1)].push_back({c2, count[size_t(c2)]});
          count[size_t(c2)] = 0;
        }
      }
    });
  for (auto& w : workers) w.join();
  return upper;
}

// in-place sum of a host array over the ranks of a sharded matrix (through the context's transport)
int SumOverRanks(cx_context* ctx, std::vector<double>* host) {
  DevBuf<double> tmp;
  CX_TRY(tmp.upload(*host, ctx->stream));
  CX_TRY(cx_allreduce_device(ctx, tmp.p, int64_t(host->size())));
  CX_TRY(cx_read_back(ctx, host->data(), tmp.p, host->size() * sizeof(double), ctx->stream));
  CX_TRY(cx_stream_sync(ctx, ctx->stream));
  return CX_OK;
}

// Sharded matrix: a rank sees the points of its shard only; every point lives in exactly one shard, so the number
// of points a camera sees and the number two cameras share are sums over the ranks (dense, once per structure).
int SumCountsOverRanks(cx_context* ctx, int C, PairCounts* upper, std::vector<size_t>* sizes) {
  if (int64_t(C) > 16384) {
    cx_set_error("visibility based preconditioners on a sharded matrix are limited to 16384 cameras (dense count exchange)");
    return CX_ERR_UNSUPPORTED;
  }
  std::vector<double> dense(size_t(C) * C + size_t(C), 0.0);
  for (int c1 = 0; c1 < C; ++c1)
    for (const auto& e : (*upper)[size_t(c1)]) dense[size_t(c1) * C + e.first] = double(e.second);
  for (int c = 0; c < C; ++c) dense[size_t(C) * C + c] = double((*sizes)[size_t(c)]);
  CX_TRY(SumOverRanks(ctx, &dense));
  for (int c1 = 0; c1 < C; ++c1) {
    (*upper)[size_t(c1)].clear();
    (*sizes)[size_t(c1)] = size_t(dense[size_t(C) * C + c1]);
    for (int c2 = c1 + 1; c2 < C; ++c2)
      if (dense[size_t(c1) * C + c2] > 0.0) (*upper)[size_t(c1)].push_back({c2, int32_t(dense[size_t(c1) * C + c2])});
  }
  return CX_OK;
}

// CreateSchurComplementGraph: edge (c1, c2) weighs |V1 n V2| / sqrt(|V1| |V2|)
Graph SchurComplementGraph(int C, const PairCounts& upper, const std::vector<size_t>& sizes) {
  Graph g;
  g.n = C;
  g.start.assign(size_t(C) + 1, 0);
  for (int c1 = 0; c1 < C; ++c1) {
    g.start[size_t(c1) + 1] += 1 + int64_t(upper[size_t(c1)].size());
    for (const auto& e : upper[size_t(c1)]) g.start[size_t(e.first) + 1]++;
  }
  for (int c = 0; c < C; ++c) g.start[size_t(c) + 1] += g.start[size_t(c)];
  g.nb.resize(size_t(g.start[size_t(C)]));
  g.w.resize(size_t(g.start[size_t(C)]));
  std::vector<int64_t> fill(g.start.begin(), g.start.end() - 1);
  // lower neighbours arrive in ascending c1, then the self edge, then the upper neighbours in ascending c2
  for (int c1 = 0; c1 < C; ++c1) {
    {
      const int64_t slot = fill[size_t(c1)]++;
      g.nb[size_t(slot)] = c1;
      g.w[size_t(slot)] = 1.0;  // kSelfEdgeWeight
    }
    for (const auto& e : upper[size_t(c1)]) {
      const double weight = static_cast<double>(e.second) / (std::sqrt(static_cast<double>(sizes[size_t(c1)] * sizes[size_t(e.first)])));
      int64_t slot = fill[size_t(c1)]++;
      g.nb[size_t(slot)] = e.first;
      g.w[size_t(slot)] = weight;
      slot = fill[size_t(e.first)]++;
      g.nb[size_t(slot)] = c1;
      g.w[size_t(slot)] = weight;
    }
  }
  return g;
}

// Greedy canonical views with the preconditioner's fixed options.  difference(v) only changes when the
// similarity of one of v's neighbours changes, so it is cached and recomputed (in the same summation order, hence
// to the same value) for those views only.
void CanonicalViews(const Graph& g, std::vector<int32_t>* centers, std::vector<int32_t>* membership) {
  constexpr double kSizePenaltyWeight = 3.0;  // similarity penalty 0, view score weight 0: both terms vanish
  constexpr size_t kMinViews = 3;
  const int n = g.n;
  centers->clear();
  std::vector<char> valid(size_t(n), 1), dirty(size_t(n), 1);
  std::vector<double> similarity(size_t(n), 0.0), difference(size_t(n), 0.0);
  std::vector<int32_t> to_canonical(size_t(n), -1);
  int num_valid = n;
  while (num_valid > 0) {
    double best_difference = -std::numeric_limits<double>::max();
    int best_view = 0;
    for (int v = 0; v < n; ++v) {
      if (!valid[size_t(v)]) continue;
      if (dirty[size_t(v)]) {
        double d = 0.0;
        for (int64_t k = g.start[size_t(v)]; k < g.start[size_t(v) + 1]; ++k) {
          const double old_similarity = similarity[size_t(g.nb[size_t(k)])];
          if (g.w[size_t(k)] > old_similarity) d += g.w[size_t(k)] - old_similarity;
        }
        d -= kSizePenaltyWeight;
        difference[size_t(v)] = d;
        dirty[size_t(v)] = 0;
      }
      if (difference[size_t(v)] > best_difference) {
        best_difference = difference[size_t(v)];
        best_view = v;
      }
    }
    if (best_difference <= 0 && centers->size() >= kMinViews) break;
    centers->push_back(best_view);
    valid[size_t(best_view)] = 0;
    --num_valid;
    for (int64_t k = g.start[size_t(best_view)]; k < g.start[size_t(best_view) + 1]; ++k) {
      const int32_t u = g.nb[size_t(k)];
      if (g.w[size_t(k)] > similarity[size_t(u)]) {
        to_canonical[size_t(u)] = best_view;
        similarity[size_t(u)] = g.w[size_t(k)];
        for (int64_t q = g.start[size_t(u)]; q < g.start[size_t(u) + 1]; ++q) dirty[size_t(g.nb[size_t(q)])] = 1;
      }
    }
  }
  std::vector<int32_t> cluster_of_center(size_t(n), -1);
  for (size_t i = 0; i < centers->size(); ++i) cluster_of_center[size_t((*centers)[i])] = int32_t(i);
  membership->assign(size_t(n), -1);
  for (int v = 0; v < n; ++v)
    if (to_canonical[size_t(v)] >= 0) (*membership)[size_t(v)] = cluster_of_center[size_t(to_canonical[size_t(v)])];
}

int32_t Find(std::vector<int32_t>& parent, int32_t v) {
  int32_t root = v;
  while (parent[size_t(root)] != root) root = parent[size_t(root)];
  while (parent[size_t(v)] != root) {
    const int32_t next = parent[size_t(v)];
    parent[size_t(v)] = root;
    v = next;
  }
  return root;
}

// ComputeSingleLinkageClustering: components of the edges with weight >= min_similarity; a cluster is named
// by its smallest vertex
int SingleLinkage(const Graph& g, double min_similarity, std::vector<int32_t>* membership) {
  const int n = g.n;
  membership->resize(size_t(n));
  std::iota(membership->begin(), membership->end(), 0);
  for (int v1 = 0; v1 < n; ++v1)
    for (int64_t k = g.start[size_t(v1)]; k < g.start[size_t(v1) + 1]; ++k) {
      const int32_t v2 = g.nb[size_t(k)];
      if (v1 > v2 || g.w[size_t(k)] < min_similarity) continue;
      const int32_t c1 = Find(*membership, v1), c2 = Find(*membership, v2);
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

struct WeightedEdge {
  double w;
  int32_t a, b;
};

// CreateClusterGraph + Degree2MaximumSpanningForest: edge weight = points seen from both clusters
int ClusterForest(cx_context* ctx, int K, const std::vector<int32_t>& membership, const Csr& pt_cams,
                  std::vector<std::pair<int32_t, int32_t>>* forest_out) {
  const int P = int(pt_cams.start.size()) - 1;
  std::vector<WeightedEdge> edges;
  std::vector<int32_t> ks;
  const bool dense = int64_t(K) * K <= (int64_t(1) << 26);
  if (!dense && ctx != nullptr) {
    cx_set_error("CLUSTER_TRIDIAGONAL on a sharded matrix is limited to 8192 clusters (dense count exchange)");
    return CX_ERR_UNSUPPORTED;
  }
  std::vector<int32_t> counts;
  std::unordered_map<uint64_t, int32_t> sparse_counts;
  if (dense) counts.assign(size_t(K) * K, 0);
  for (int p = 0; p < P; ++p) {
    ks.clear();
    for (int64_t q = pt_cams.start[size_t(p)]; q < pt_cams.start[size_t(p) + 1]; ++q) ks.push_back(membership[size_t(pt_cams.idx[size_t(q)])]);
    std::sort(ks.begin(), ks.end());
    ks.erase(std::unique(ks.begin(), ks.end()), ks.end());
    for (size_t i = 0; i < ks.size(); ++i)
      for (size_t j = i + 1; j < ks.size(); ++j) {
        if (dense) counts[size_t(ks[i]) * K + ks[j]]++;
        else sparse_counts[(uint64_t(uint32_t(ks[i])) << 32) | uint32_t(ks[j])]++;
      }
  }
  if (dense && ctx != nullptr) {  // the other shards' points
    std::vector<double> sum(counts.begin(), counts.end());
    CX_TRY(SumOverRanks(ctx, &sum));
    for (size_t i = 0; i < sum.size(); ++i) counts[i] = int32_t(sum[i]);
  }
  if (dense) {
    for (int a = 0; a < K; ++a)
      for (int b = a + 1; b < K; ++b)
        if (counts[size_t(a) * K + b] > 0) edges.push_back({double(counts[size_t(a) * K + b]), a, b});
  } else {
    for (const auto& e : sparse_counts) edges.push_back({double(e.second), int32_t(e.first >> 32), int32_t(e.first & 0xffffffffu)});
  }
  // decreasing (weight, (a, b)): what std::sort over reverse iterators of pair<double, pair<int, int>> gives
  std::sort(edges.begin(), edges.end(), [](const WeightedEdge& x, const WeightedEdge& y) {
    if (x.w != y.w) return x.w > y.w;
    if (x.a != y.a) return x.a > y.a;
    return x.b > y.b;
  });
  std::vector<int32_t> component(static_cast<size_t>(K)), degree(size_t(K), 0);
  std::iota(component.begin(), component.end(), 0);
  std::vector<std::pair<int32_t, int32_t>>& forest = *forest_out;
  forest.clear();
  for (const WeightedEdge& e : edges) {
    if (degree[size_t(e.a)] == 2 || degree[size_t(e.b)] == 2) continue;
    int32_t r1 = Find(component, e.a), r2 = Find(component, e.b);
    if (r1 == r2) continue;
    forest.push_back({e.a, e.b});
    degree[size_t(e.a)]++;
    degree[size_t(e.b)]++;
    if (r2 < r1) std::swap(r1, r2);
    component[size_t(r2)] = r1;
  }
  return CX_OK;
}

// Clusters and cluster pairs (everything VisibilityBasedPreconditioner derives from the structure alone).  ctx is the
// context of a sharded matrix (the counts are then summed over its ranks) or null: no device is touched without it.
int ComputeClusters(const cx_cell* cells, int C, int P, int64_t O, cx_context* ctx, int preconditioner_type, int clustering_type,
                    cx_vis_plan* plan, std::vector<std::array<int32_t, 2>>* partner_out, int first_camera_block = -1) {
  plan->preconditioner_type = preconditioner_type;
  plan->clustering_type = clustering_type;
  const bool sharded = ctx != nullptr && ctx->nranks > 1;
  Csr cam_pts, pt_cams;
  Visibility(cells, C, P, O, &cam_pts, &pt_cams, first_camera_block);
  PairCounts upper = SharedPointCounts(C, cam_pts, pt_cams);
  std::vector<size_t> sizes(static_cast<size_t>(C));
  for (int c = 0; c < C; ++c) sizes[size_t(c)] = size_t(cam_pts.start[size_t(c) + 1] - cam_pts.start[size_t(c)]);
  if (sharded) CX_TRY(SumCountsOverRanks(ctx, C, &upper, &sizes));
  const Graph graph = SchurComplementGraph(C, upper, sizes);
  plan->global_pair_c1.clear();
  plan->global_pair_c2.clear();
  if (sharded) {  // the S cells of the whole matrix: every diagonal cell and every co-visible pair (filtered in BuildPlan)
    for (int c1 = 0; c1 < C; ++c1) {
      plan->global_pair_c1.push_back(c1);
      plan->global_pair_c2.push_back(c1);
      for (const auto& e : upper[size_t(c1)]) {
        plan->global_pair_c1.push_back(c1);
        plan->global_pair_c2.push_back(e.first);
      }
    }
  }
  PairCounts().swap(upper);
  // ClusterCameras
  std::vector<int32_t> raw;
  int K = 0;
  if (clustering_type == CX_CANONICAL_VIEWS) {
    std::vector<int32_t> centers;
    CanonicalViews(graph, &centers, &raw);
    K = int(centers.size());
  } else {
    K = SingleLinkage(graph, 0.9 /* kSingleLinkageMinSimilarity */, &raw);
  }
  // FlattenMembershipMap: a view without a centre joins cluster (camera id mod number of clusters); contiguous
  // numbers in order of first appearance
  plan->num_clusters = K;
  plan->membership.assign(size_t(C), -1);
  {
    std::unordered_map<int32_t, int32_t> index_of;
    for (int c = 0; c < C; ++c) {
      int32_t id = raw[size_t(c)];
      if (id == -1) id = c % K;
      auto it = index_of.find(id);
      if (it == index_of.end()) it = index_of.emplace(id, int32_t(index_of.size())).first;
      plan->membership[size_t(c)] = it->second;
    }
  }
  // cluster pairs; partner[k] = the (at most two) forest neighbours of cluster k
  std::vector<std::pair<int32_t, int32_t>> forest;
  if (preconditioner_type == CX_CLUSTER_TRIDIAGONAL) CX_TRY(ClusterForest(sharded ? ctx : nullptr, K, plan->membership, pt_cams, &forest));
  std::vector<std::array<int32_t, 2>>& partner = *partner_out;
  partner.assign(size_t(K), std::array<int32_t, 2>{-1, -1});
  for (const auto& e : forest) {
    for (int side = 0; side < 2; ++side) {
      const int32_t k = side ? e.second : e.first, other = side ? e.first : e.second;
      partner[size_t(k)][partner[size_t(k)][0] < 0 ? 0 : 1] = other;
    }
  }
  plan->cluster_pairs.clear();
  for (int k = 0; k < K; ++k) plan->cluster_pairs.push_back({k, k});
  for (const auto& e : forest) plan->cluster_pairs.push_back({std::min(e.first, e.second), std::max(e.first, e.second)});
  std::sort(plan->cluster_pairs.begin(), plan->cluster_pairs.end());
  return CX_OK;
}

int BuildPlan(cx_matrix* A, int preconditioner_type, int clustering_type, cx_vis_plan* plan) {
  const int C = A->C;
  std::vector<std::array<int32_t, 2>> partner;
  // (an embedding's inner matrix: real points and e-rows only, cx_matrix::P_vis)
  CX_TRY(ComputeClusters(A->cells.data(), C, A->P_vis >= 0 ? A->P_vis : A->P, A->O_vis >= 0 ? A->O_vis : A->O, A->ctx, preconditioner_type,
                         clustering_type, plan, &partner, A->P));
  const int K = plan->num_clusters;
  // block pairs: the S cells (all diagonal cells + co-visible pairs, lexicographic: cxs_build_pair_lists) whose
  // cluster pair is in the preconditioner
  CX_TRY(cxs_build_pair_lists(A));
  if (A->pairs_state != 1) {
    cx_set_error("the visibility based preconditioners need the row pair lists of S, and this structure has more than 2^28 pairs");
    return CX_ERR_UNSUPPORTED;
  }
  plan->sel_cells.clear();
  std::vector<int32_t> sel_items, sel_offdiag;
  for (int64_t cell = 0; cell < A->num_cells; ++cell) {
    const int32_t k1 = plan->membership[size_t(A->h_cell_c1[size_t(cell)])], k2 = plan->membership[size_t(A->h_cell_c2[size_t(cell)])];
    if (k1 != k2 && partner[size_t(k1)][0] != k2 && partner[size_t(k1)][1] != k2) continue;
    plan->sel_cells.push_back(int32_t(cell));
    sel_offdiag.push_back(k1 != k2 ? 1 : 0);
    for (int32_t it = A->h_cell_item_start[size_t(cell)]; it < A->h_cell_item_start[size_t(cell) + 1]; ++it) sel_items.push_back(it);
  }
  plan->num_sel_items = int64_t(sel_items.size());
  if (!plan->global_pair_c1.empty()) {  // sharded: the same filter on the cells of all ranks
    size_t kept = 0;
    for (size_t k = 0; k < plan->global_pair_c1.size(); ++k) {
      const int32_t k1 = plan->membership[size_t(plan->global_pair_c1[k])], k2 = plan->membership[size_t(plan->global_pair_c2[k])];
      if (k1 != k2 && partner[size_t(k1)][0] != k2 && partner[size_t(k1)][1] != k2) continue;
      plan->global_pair_c1[kept] = plan->global_pair_c1[k];
      plan->global_pair_c2[kept] = plan->global_pair_c2[k];
      ++kept;
    }
    plan->global_pair_c1.resize(kept);
    plan->global_pair_c2.resize(kept);
  }

  // ---- paths of the forest, longest first
  struct Path { std::vector<int32_t> clusters; int64_t rows = 0; };
  std::vector<Path> paths;
  {
    std::vector<int32_t> cluster_size(size_t(K), 0);
    for (int c = 0; c < C; ++c) cluster_size[size_t(plan->membership[size_t(c)])]++;
    std::vector<char> seen(size_t(K), 0);
    for (int k = 0; k < K; ++k) {
      if (seen[size_t(k)] || partner[size_t(k)][1] >= 0) continue;  // start at an end (at most one neighbour)
      Path path;
      int32_t prev = -1, cur = k;
      while (cur >= 0) {
        seen[size_t(cur)] = 1;
        path.clusters.push_back(cur);
        path.rows += 9 * int64_t(cluster_size[size_t(cur)]);
        int32_t next = -1;
        for (int32_t cand : partner[size_t(cur)])
          if (cand >= 0 && cand != prev) next = cand;
        prev = cur;
        cur = next;
      }
      paths.push_back(std::move(path));
    }
    std::stable_sort(paths.begin(), paths.end(), [](const Path& x, const Path& y) { return (x.rows + 31) / 32 > (y.rows + 31) / 32; });
  }
  // ---- rows: path by path (each starting at a multiple of 32), cluster by cluster, camera by camera
  std::vector<std::vector<int32_t>> cameras_of(static_cast<size_t>(K));
  for (int c = 0; c < C; ++c) cameras_of[size_t(plan->membership[size_t(c)])].push_back(c);
  plan->cam_row.assign(size_t(C), 0);
  plan->path_first_blk.clear();
  plan->path_num_blk.clear();
  std::vector<int32_t> col_end;  // per row
  int64_t row = 0;
  for (const Path& path : paths) {
    const int64_t first = row;
    std::vector<int64_t> cluster_end;
    for (int32_t k : path.clusters) {
      for (int32_t c : cameras_of[size_t(k)]) {
        plan->cam_row[size_t(c)] = int32_t(row);
        row += 9;
      }
      cluster_end.push_back(row);
    }
    const int64_t padded = (row + 31) / 32 * 32;
    if (padded > std::numeric_limits<int32_t>::max() / 2) {
      cx_set_error("too many cameras for the banded preconditioner");
      return CX_ERR_UNSUPPORTED;
    }
    col_end.resize(size_t(padded));
    int64_t r = first;
    for (size_t i = 0; i < path.clusters.size(); ++i) {
      const int64_t reach = cluster_end[std::min(i + 1, path.clusters.size() - 1)];
      for (; r < cluster_end[i]; ++r) col_end[size_t(r)] = int32_t(reach);
    }
    for (; r < padded; ++r) col_end[size_t(r)] = int32_t(r + 1);
    plan->path_first_blk.push_back(int32_t(first / 32));
    plan->path_num_blk.push_back(int32_t((padded - first) / 32));
    row = padded;
  }
  plan->N = int32_t(row);
  plan->num_paths = int32_t(paths.size());
  plan->row_src.assign(size_t(plan->N), -1);
  for (int c = 0; c < C; ++c)
    for (int a = 0; a < 9; ++a) plan->row_src[size_t(plan->cam_row[size_t(c)] + a)] = 9 * c + a;
  const int num_blocks = plan->N / 32;
  plan->blk_cend.assign(size_t(num_blocks), 0);
  int32_t band = 32;
  for (int b = 0; b < num_blocks; ++b) {
    int32_t e = 0;
    for (int i = 0; i < 32; ++i) e = std::max(e, col_end[size_t(32 * b + i)]);
    plan->blk_cend[size_t(b)] = e;
    band = std::max(band, e - 32 * b);
  }
  plan->ld = (band + 15) / 16 * 16;
  const int max_blocks = plan->num_paths ? plan->path_num_blk[0] : 0;
  plan->step_paths.assign(size_t(max_blocks), 0);
  plan->step_tiles.assign(size_t(max_blocks), 0);
  for (int p = 0; p < plan->num_paths; ++p)
    for (int s = 0; s < plan->path_num_blk[size_t(p)]; ++s) {
      const int b = plan->path_first_blk[size_t(p)] + s;
      const int rem = plan->blk_cend[size_t(b)] - 32 * (b + 1);
      const int T = (rem + 63) / 64;
      plan->step_paths[size_t(s)] = p + 1;
      plan->step_tiles[size_t(s)] = std::max(plan->step_tiles[size_t(s)], T * (T + 1) / 2);
    }
  // two band matrices (working copy and factor)
  const double bytes = 2.0 * double(plan->N) * double(plan->ld + 1) * 8.0;
  if (bytes > 120e9) {
    cx_set_error("the visibility clusters of this problem are too large for the banded preconditioner (%.1f GB)", bytes * 1e-9);
    return CX_ERR_UNSUPPORTED;
  }
  hipStream_t st = A->ctx->stream;
  CX_TRY(plan->d_sel_cells.upload(plan->sel_cells, st));
  CX_TRY(plan->d_sel_items.upload(sel_items, st));
  CX_TRY(plan->d_sel_offdiag.upload(sel_offdiag, st));
  CX_TRY(plan->d_cam_row.upload(plan->cam_row, st));
  CX_TRY(plan->d_row_src.upload(plan->row_src, st));
  CX_TRY(plan->d_path_first_blk.upload(plan->path_first_blk, st));
  CX_TRY(plan->d_path_num_blk.upload(plan->path_num_blk, st));
  CX_TRY(plan->d_blk_cend.upload(plan->blk_cend, st));
  if (std::getenv("CX_VISIBILITY_VERBOSE"))
    std::fprintf(stderr, "[cxschur] visibility preconditioner %d: %d cameras, %d clusters, %d cluster pairs, %lld of %lld S cells, %d paths (longest %d blocks), band %d, %.2f GB per band matrix\n",
                 preconditioner_type, C, K, int(plan->cluster_pairs.size()) - K, (long long)plan->sel_cells.size(), (long long)A->num_cells,
                 plan->num_paths, max_blocks, plan->ld, double(plan->N) * (plan->ld + 1) * 8e-9);
  return CX_OK;
}

}  // namespace

int cxv_get_plan(cx_matrix* A, int preconditioner_type, int clustering_type, cx_vis_plan** out) {
  if (A->embed) A = A->embed->inner;
  if (!A->is239) {
    cx_set_error("CLUSTER_JACOBI / CLUSTER_TRIDIAGONAL are available for the static <2,3,9> layout only");
    return CX_ERR_UNSUPPORTED;
  }
  if (preconditioner_type != CX_CLUSTER_JACOBI && preconditioner_type != CX_CLUSTER_TRIDIAGONAL) return CX_ERR_INVALID_ARGUMENT;
  if (clustering_type != CX_CANONICAL_VIEWS && clustering_type != CX_SINGLE_LINKAGE) {
    cx_set_error("Unknown visibility clustering algorithm.");
    return CX_ERR_INVALID_ARGUMENT;
  }
  if (A->C == 0) {
    cx_set_error("Jacobian should have at least 1 f_block for visibility based preconditioning.");
    return CX_ERR_INVALID_ARGUMENT;
  }
  for (const auto& plan : A->vis_plans)
    if (plan->preconditioner_type == preconditioner_type && plan->clustering_type == clustering_type) {
      *out = plan.get();
      return CX_OK;
    }
  auto plan = std::make_shared<cx_vis_plan>();
  const auto t0 = std::chrono::steady_clock::now();
  CX_TRY(BuildPlan(A, preconditioner_type, clustering_type, plan.get()));
  if (std::getenv("CX_VISIBILITY_VERBOSE"))
    std::fprintf(stderr, "[cxschur] visibility plan built in %.2f s\n",
                 std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  A->vis_plans.push_back(plan);
  *out = plan.get();
  return CX_OK;
}

extern "C" int cx_visibility_structure(cx_matrix* A, int32_t preconditioner_type, int32_t clustering_type, int32_t* membership,
                                       int32_t* num_clusters, int32_t* num_cluster_pairs, int32_t* cluster_pair_1,
                                       int32_t* cluster_pair_2, int32_t cluster_pair_capacity, int64_t* num_block_pairs,
                                       int32_t* block_pair_1, int32_t* block_pair_2, int64_t block_pair_capacity) {
  CX_CHECK_ARG(A != nullptr);
  CX_HIP(hipSetDevice(A->ctx->device));
  cx_vis_plan* plan = nullptr;
  CX_TRY(cxv_get_plan(A, preconditioner_type, clustering_type, &plan));
  if (membership) std::copy(plan->membership.begin(), plan->membership.end(), membership);
  if (num_clusters) *num_clusters = plan->num_clusters;
  if (num_cluster_pairs) *num_cluster_pairs = int32_t(plan->cluster_pairs.size());
  for (size_t k = 0; k < plan->cluster_pairs.size() && int64_t(k) < cluster_pair_capacity; ++k) {
    cluster_pair_1[k] = plan->cluster_pairs[k].first;
    cluster_pair_2[k] = plan->cluster_pairs[k].second;
  }
  if (num_block_pairs) *num_block_pairs = int64_t(plan->sel_cells.size());
  for (size_t k = 0; k < plan->sel_cells.size() && int64_t(k) < block_pair_capacity; ++k) {
    block_pair_1[k] = A->h_cell_c1[size_t(plan->sel_cells[k])];
    block_pair_2[k] = A->h_cell_c2[size_t(plan->sel_cells[k])];
  }
  return CX_OK;
}

// The same clustering without a device or a cx_matrix: straight from the flat block structure (static <2,3,9> layout:
// two cells per row block, e-block first, rows sorted by e-block).  Used by the CPU tests of the host logic.
extern "C" int cx_visibility_clusters_host(const cx_block_structure* bs, int32_t num_eliminate_blocks, int32_t preconditioner_type,
                                           int32_t clustering_type, int32_t* membership, int32_t* num_clusters,
                                           int32_t* num_cluster_pairs, int32_t* cluster_pair_1, int32_t* cluster_pair_2,
                                           int32_t cluster_pair_capacity) {
  CX_CHECK_ARG(bs != nullptr && num_eliminate_blocks > 0 && num_eliminate_blocks < bs->num_col_blocks);
  CX_CHECK_ARG(preconditioner_type == CX_CLUSTER_JACOBI || preconditioner_type == CX_CLUSTER_TRIDIAGONAL);
  CX_CHECK_ARG(clustering_type == CX_CANONICAL_VIEWS || clustering_type == CX_SINGLE_LINKAGE);
  const int P = num_eliminate_blocks, C = bs->num_col_blocks - num_eliminate_blocks;
  const int64_t O = bs->num_row_blocks;
  int32_t last = -1;
  for (int64_t r = 0; r < O; ++r) {
    const int32_t first = bs->row_cell_begin[r];
    const bool ok = bs->row_cell_begin[r + 1] - first == 2 && bs->cells[first].block_id < P && bs->cells[first].block_id >= last &&
                    bs->cells[first + 1].block_id >= P && first == 2 * r;
    if (!ok) {
      cx_set_error("cx_visibility_clusters_host: static <2,3,9> layout expected (row block %lld)", (long long)r);
      return CX_ERR_UNSUPPORTED;
    }
    last = bs->cells[first].block_id;
  }
  cx_vis_plan plan;
  std::vector<std::array<int32_t, 2>> partner;
  CX_TRY(ComputeClusters(bs->cells, C, P, O, nullptr, preconditioner_type, clustering_type, &plan, &partner));
  if (membership) std::copy(plan.membership.begin(), plan.membership.end(), membership);
  if (num_clusters) *num_clusters = plan.num_clusters;
  if (num_cluster_pairs) *num_cluster_pairs = int32_t(plan.cluster_pairs.size());
  for (size_t k = 0; k < plan.cluster_pairs.size() && int64_t(k) < cluster_pair_capacity; ++k) {
    cluster_pair_1[k] = plan.cluster_pairs[k].first;
    cluster_pair_2[k] = plan.cluster_pairs[k].second;
  }
  return CX_OK;
}
