// Host-side integer work of the path: the automatic Schur ordering of a bundle-adjustment
// program (ComputeStableSchurOrdering, parameter_block_ordering.cc:50-83 with
// StableIndependentSetOrdering, graph_algorithms.h:165-227).  Outputs must match the reference
// exactly; tests compare with the oracle's graph-based restatement.
#include <algorithm>
#include <numeric>
#include <vector>

#include "cx_internal.h"

extern "C" int cx_stable_schur_ordering(int32_t C, int32_t P, int64_t O, const int32_t* cam, const int32_t* pt,
                                        int32_t* ordering, int32_t* independent_set_size) {
  CX_CHECK_ARG(C >= 0 && P >= 0 && O >= 0 && (O == 0 || (cam && pt)) && ordering && independent_set_size);
  for (int64_t i = 0; i < O; ++i) CX_CHECK_ARG(cam[i] >= 0 && cam[i] < C && pt[i] >= 0 && pt[i] < P);
  // The Hessian graph of a BAL program is bipartite: camera i (vertex i) -- point j (vertex C + j)
  // for every residual block (parameter_block_ordering.cc:126-160).  Degrees count DISTINCT neighbours.
  std::vector<int64_t> key(O);
  for (int64_t i = 0; i < O; ++i) key[i] = int64_t(cam[i]) * P + pt[i];
  std::sort(key.begin(), key.end());
  key.erase(std::unique(key.begin(), key.end()), key.end());
  const int64_t E = int64_t(key.size());
  const int n = C + P;
  std::vector<int32_t> degree(n, 0);
  std::vector<int64_t> cam_start(size_t(C) + 1, 0), pt_start(size_t(P) + 1, 0);
  for (int64_t e = 0; e < E; ++e) {
    const int32_t c = int32_t(key[e] / P), p = int32_t(key[e] % P);
    degree[c]++;
    degree[C + p]++;
    cam_start[c + 1]++;
    pt_start[p + 1]++;
  }
  std::partial_sum(cam_start.begin(), cam_start.end(), cam_start.begin());
  std::partial_sum(pt_start.begin(), pt_start.end(), pt_start.begin());
  std::vector<int32_t> cam_nbr(E), pt_nbr(E);
  {
    std::vector<int64_t> cc(cam_start.begin(), cam_start.end() - 1), pc(pt_start.begin(), pt_start.end() - 1);
    for (int64_t e = 0; e < E; ++e) {
      const int32_t c = int32_t(key[e] / P), p = int32_t(key[e] % P);
      cam_nbr[cc[c]++] = p;
      pt_nbr[pc[p]++] = c;
    }
  }
  // vertex queue in program order (cameras, then points: bundle_adjuster.cc:253-267), stable sort by degree
  std::vector<int32_t> queue(n);
  std::iota(queue.begin(), queue.end(), 0);
  std::stable_sort(queue.begin(), queue.end(), [&](int32_t a, int32_t b) { return degree[a] < degree[b]; });
  std::vector<char> color(n, 0);  // 0 white, 1 grey, 2 black
  int k = 0;
  for (int32_t v : queue) {
    if (color[v] != 0) continue;
    ordering[k++] = v;
    color[v] = 2;
    if (v < C) {
      for (int64_t e = cam_start[v]; e < cam_start[v + 1]; ++e) color[C + cam_nbr[e]] = 1;
    } else {
      for (int64_t e = pt_start[v - C]; e < pt_start[v - C + 1]; ++e) color[pt_nbr[e]] = 1;
    }
  }
  *independent_set_size = k;
  for (int32_t v : queue)
    if (color[v] != 2) ordering[k++] = v;
  return CX_OK;
}
