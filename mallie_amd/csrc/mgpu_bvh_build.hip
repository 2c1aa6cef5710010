// mgpu_bvh_build.hip -- BVHAccel::Build on the device, producing the reference's tree BYTE FOR BYTE.
//
// The reference builder (bvh_accel.cc:321-443) is a depth-first recursion whose result depends on more than the split
// planes: the ORDER std::partition leaves the triangle indices in decides the order triangles are tested inside a leaf
// (exact-t ties) and how later splits see their ranges.  So this is not "a" GPU BVH builder but a parallel evaluation
// of that very algorithm:
//
//   * the tree is built breadth-first, one kernel launch per level, one 1024-thread workgroup per node of the level;
//   * per node: box = union of the (pre-computed) triangle boxes -/+ 1024*eps  -- identical to the reference's
//     per-vertex min/max because x -> fl(x -/+ pad) is monotone (bvh_accel.cc:285-315);
//   * 2 x 3 x 64 histogram of the triangle boxes' min / max cells in LDS (bvh_accel.cc:82-142), cut selection by three
//     lanes walking the 63 planes with the reference's running counts and cost expression (bvh_accel.cc:156-255);
//   * the partition is libstdc++'s bidirectional two-pointer scheme (the one bvh_accel.cc:402 instantiates) in closed
//     form: with m = number of "left" items, the k-th misplaced item of [l, l+m) counted from the left swaps with the
//     k-th misplaced item of [l+m, r) counted from the RIGHT; both position lists come from workgroup prefix scans;
//   * a degenerate partition keeps the order and splits at the median (bvh_accel.cc:405-409);
//   * finally sub-tree sizes (bottom-up) give every node its depth-first pre-order number, i.e. the reference's index.
//
// All fp64 expressions are the host builder's (mallie_amd/csrc/host/bvh_build.cc), compiled with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include <cfloat>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/mgpu.h"

namespace {

constexpr int kBBMax = 1024;           // threads per workgroup at the top of the tree (one workgroup per node; 256 / 64 further down)
constexpr double kPad = 2.220446049250313e-16 * 1024;
constexpr int kMaxBins = 256;          // LDS histogram capacity per (min|max, axis)

struct TriRec {   // 72 bytes
  double lo[3], hi[3], csum[3];
};

struct BNode {    // breadth-first work / result record
  double bmin[3], bmax[3];
  uint32_t l, r;
  uint32_t child[2];  // BFS ids
  uint32_t parent;
  uint32_t subtree;   // nodes in the sub-tree (filled bottom-up)
  uint32_t pre;       // depth-first pre-order number (filled top-down)
  int16_t depth;
  int8_t leaf, axis;
};

struct BuildOpts {
  double costTaabb;
  int minLeaf, maxDepth, binSize;
};

__global__ void k_tri_records(const double *__restrict__ verts, const uint32_t *__restrict__ faces, size_t nf,
                              TriRec *__restrict__ rec, uint32_t *__restrict__ idx) {
  const size_t f = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (f >= nf) return;
  const double *p0 = verts + 3 * (size_t)faces[3 * f + 0];
  const double *p1 = verts + 3 * (size_t)faces[3 * f + 1];
  const double *p2 = verts + 3 * (size_t)faces[3 * f + 2];
  TriRec t;
  for (int k = 0; k < 3; ++k) {
    const double a = p0[k], b = p1[k], c = p2[k];
    double lo = (b < a) ? b : a;   // std::min(a, b)
    lo = (c < lo) ? c : lo;
    double hi = (a < b) ? b : a;   // std::max(a, b)
    hi = (hi < c) ? c : hi;
    t.lo[k] = lo;
    t.hi[k] = hi;
    t.csum[k] = a + b + c;         // SAHPred's centre sum, (p0 + p1) + p2 (bvh_accel.cc:274)
  }
  rec[f] = t;
  idx[f] = (uint32_t)f;
}

__device__ __forceinline__ double box_area(const double lo[3], const double hi[3]) {
  const double a = hi[0] - lo[0], b = hi[1] - lo[1], c = hi[2] - lo[2];
  return 2.0 * (a * b + b * c + c * a);
}

// workgroup-wide exclusive scan of one flag per thread; returns this thread's rank and the total
template <int kBB>
__device__ __forceinline__ uint32_t block_scan(uint32_t flag, uint32_t *s_wave, uint32_t &total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long m = __ballot(flag != 0);
  const uint32_t in_wave = __popcll(m & ((1ull << lane) - 1ull));
  __syncthreads();
  if (lane == 0) s_wave[wave] = (uint32_t)__popcll(m);
  __syncthreads();
  uint32_t base = 0, sum = 0;
  for (int w = 0; w < kBB / 64; ++w) {
    const uint32_t c = s_wave[w];
    if (w < wave) base += c;
    sum += c;
  }
  total = sum;
  return base + in_wave;
}

template <int kBB>
__global__ __launch_bounds__(kBB) void k_build_level(BNode *__restrict__ nodes, uint32_t level_begin, uint32_t level_count,
                                                     const TriRec *__restrict__ rec, uint32_t *__restrict__ idx,
                                                     uint32_t *__restrict__ lpos, uint32_t *__restrict__ rpos,
                                                     uint32_t *__restrict__ next_count, uint32_t next_begin,
                                                     uint32_t max_nodes, BuildOpts opt) {
  __shared__ double s_red[kBB / 64][6];
  __shared__ double s_box[6];
  __shared__ uint32_t s_bins[2 * 3 * kMaxBins];
  __shared__ uint32_t s_wave[kBB / 64];
  __shared__ double s_cost[3], s_cut[3];
  __shared__ uint32_t s_u[4];
  const uint32_t nid = level_begin + blockIdx.x;
  if (blockIdx.x >= level_count) return;
  BNode *nd = nodes + nid;
  const uint32_t l = nd->l, r = nd->r, n = r - l;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // ---- 1. bounds ------------------------------------------------------------------------------------------------
  double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
  for (uint32_t i = l + tid; i < r; i += kBB) {
    const TriRec &t = rec[idx[i]];
    for (int k = 0; k < 3; ++k) {
      lo[k] = (t.lo[k] < lo[k]) ? t.lo[k] : lo[k];
      hi[k] = (hi[k] < t.hi[k]) ? t.hi[k] : hi[k];
    }
  }
  for (int off = 32; off; off >>= 1)
    for (int k = 0; k < 3; ++k) {
      const double a = __shfl_down(lo[k], off), b = __shfl_down(hi[k], off);
      lo[k] = (a < lo[k]) ? a : lo[k];
      hi[k] = (hi[k] < b) ? b : hi[k];
    }
  if (lane == 0)
    for (int k = 0; k < 3; ++k) { s_red[wave][k] = lo[k]; s_red[wave][3 + k] = hi[k]; }
  __syncthreads();
  if (tid < 6) {
    double v = s_red[0][tid];
    for (int w = 1; w < kBB / 64; ++w) {
      const double x = s_red[w][tid];
      v = (tid < 3) ? ((x < v) ? x : v) : ((v < x) ? x : v);
    }
    v = (tid < 3) ? v - kPad : v + kPad;
    s_box[tid] = v;
    if (tid < 3) nd->bmin[tid] = v; else nd->bmax[tid - 3] = v;
  }
  __syncthreads();
  const double bmin[3] = {s_box[0], s_box[1], s_box[2]}, bmax[3] = {s_box[3], s_box[4], s_box[5]};

  // ---- 2. leaf? ---------------------------------------------------------------------------------------------------
  if (n < (uint32_t)opt.minLeaf || nd->depth >= opt.maxDepth) {
    if (tid == 0) {
      nd->leaf = 1;
      nd->axis = 0;
      nd->child[0] = nd->child[1] = 0xFFFFFFFFu;
    }
    return;
  }

  // ---- 3. histogram (bvh_accel.cc:82-142) ---------------------------------------------------------------------------
  const int nb = opt.binSize;
  const double fb = (double)nb;
  for (int i = tid; i < 6 * nb; i += kBB) s_bins[i] = 0;
  __syncthreads();
  double scale[3];
  for (int k = 0; k < 3; ++k) {
    const double ext = bmax[k] - bmin[k];
    scale[k] = (ext > kPad) ? fb / ext : 0.0;
  }
  for (uint32_t i = l + tid; i < r; i += kBB) {
    const TriRec &t = rec[idx[i]];
    for (int k = 0; k < 3; ++k) {
      unsigned long long ilo = (unsigned int)floor((t.lo[k] - bmin[k]) * scale[k]);
      unsigned long long ihi = (unsigned int)floor((t.hi[k] - bmin[k]) * scale[k]);
      if ((double)ilo >= fb) ilo = (unsigned long long)(fb - 1);
      if ((double)ihi >= fb) ihi = (unsigned long long)(fb - 1);
      atomicAdd(&s_bins[k * nb + (int)ilo], 1u);
      atomicAdd(&s_bins[(3 + k) * nb + (int)ihi], 1u);
    }
  }
  __syncthreads();

  // ---- 4. cut (bvh_accel.cc:156-255): three lanes, one per axis, walk the nb-1 interior planes ---------------------
  if (tid < 3) {
    const int j = tid;
    const double Ta = opt.costTaabb, Tt = 1.0 - opt.costTaabb;
    const double total = box_area(bmin, bmax);
    const double inv_total = (total > kPad) ? 1.0 / total : 0.0;
    const double step = (bmax[j] - bmin[j]) * (1.0 / nb);
    double best_pos = bmin[j] + 0.5 * step, best_cost = DBL_MAX;
    double loL[3] = {bmin[0], bmin[1], bmin[2]}, hiL[3] = {bmax[0], bmax[1], bmax[2]};
    double loR[3] = {bmin[0], bmin[1], bmin[2]}, hiR[3] = {bmax[0], bmax[1], bmax[2]};
    unsigned long long nl = 0, nr = n;
    for (int i = 0; i < nb - 1; ++i) {
      nl += s_bins[j * nb + i];
      nr -= s_bins[(3 + j) * nb + i];
      const double pos = bmin[j] + (i + 0.5) * step;
      hiL[j] = pos;
      loR[j] = pos;
      const double cost = 2.0 * Ta + (box_area(loL, hiL) * inv_total) * (double)nl * Tt +
                          (box_area(loR, hiR) * inv_total) * (double)nr * Tt;
      if (cost < best_cost) { best_cost = cost; best_pos = pos; }
    }
    s_cost[j] = best_cost;
    s_cut[j] = best_pos;
  }
  __syncthreads();
  int axis = 0;
  {
    double c = s_cost[0];
    if (c > s_cost[1]) { axis = 1; c = s_cost[1]; }
    if (c > s_cost[2]) { axis = 2; }
  }
  const double pos3 = s_cut[axis] * 3.0; // SAHPred: centre sum < pos * 3.0

  // ---- 5. partition, libstdc++ element order in closed form ----------------------------------------------------------
  // 5a. m = number of items that go left
  uint32_t cnt = 0;
  for (uint32_t i = l + tid; i < r; i += kBB) cnt += (rec[idx[i]].csum[axis] < pos3) ? 1u : 0u;
  for (int off = 32; off; off >>= 1) cnt += __shfl_down(cnt, off);
  __syncthreads();
  if (lane == 0) s_wave[wave] = cnt;
  __syncthreads();
  if (tid == 0) {
    uint32_t m = 0;
    for (int w = 0; w < kBB / 64; ++w) m += s_wave[w];
    s_u[0] = m;
  }
  __syncthreads();
  const uint32_t m = s_u[0];
  uint32_t mid = l + m;
  if (m == 0 || m == n) {
    mid = l + (n >> 1); // degenerate: order untouched, object median (bvh_accel.cc:405-409)
  } else {
    // 5b. misplaced items of the left part, in left-to-right order
    uint32_t carry = 0;
    for (uint32_t base = l; base < mid; base += kBB) {
      const uint32_t i = base + tid;
      const uint32_t bad = (i < mid && !(rec[idx[i]].csum[axis] < pos3)) ? 1u : 0u;
      uint32_t tot;
      const uint32_t k = block_scan<kBB>(bad, s_wave, tot);
      if (bad) lpos[l + carry + k] = i;
      carry += tot;
    }
    const uint32_t nswap = carry;
    // 5c. misplaced items of the right part, in right-to-left order
    carry = 0;
    for (uint32_t off = 0; off < r - mid; off += kBB) {
      const uint32_t o = off + tid;
      const bool in = o < r - mid;
      const uint32_t i = in ? (r - 1 - o) : 0;
      const uint32_t good = (in && (rec[idx[i]].csum[axis] < pos3)) ? 1u : 0u;
      uint32_t tot;
      const uint32_t k = block_scan<kBB>(good, s_wave, tot);
      if (good) rpos[l + carry + k] = i;
      carry += tot;
    }
    __syncthreads();
    // 5d. swap the pairs
    for (uint32_t k = tid; k < nswap; k += kBB) {
      const uint32_t a = lpos[l + k], b = rpos[l + k];
      const uint32_t va = idx[a], vb = idx[b];
      idx[a] = vb;
      idx[b] = va;
    }
  }

  // ---- 6. children ------------------------------------------------------------------------------------------------------
  if (tid == 0) {
    const uint32_t slot = atomicAdd(next_count, 2u);
    const uint32_t c0 = next_begin + slot, c1 = c0 + 1;
    if (c1 >= max_nodes) { // cannot happen for a well-formed build (see the host-side bound); never write out of range
      nd->leaf = 1;
      nd->child[0] = nd->child[1] = 0xFFFFFFFFu;
      return;
    }
    nd->leaf = 0;
    nd->axis = (int8_t)axis;
    nd->child[0] = c0;
    nd->child[1] = c1;
    BNode a;
    memset(&a, 0, sizeof(a));
    a.parent = nid;
    a.depth = (int16_t)(nd->depth + 1);
    BNode b = a;
    a.l = l; a.r = mid;
    b.l = mid; b.r = r;
    nodes[c0] = a;
    nodes[c1] = b;
  }
}

// ---- the top of the tree: several workgroups per node -----------------------------------------------------------------
// One workgroup streaming through a 10 M-triangle node five times takes 87 ms (root), 43 ms (level 1), ...: the first
// eight levels were 172 of 283 ms.  Here every node of a level gets G workgroups, each owning one contiguous chunk of the
// node's index range; the per-node quantities (box, histogram, number of left items, the two lists of misplaced items)
// are assembled from per-chunk pieces across grid-wide barriers.  Every workgroup of a node evaluates the node-level
// decisions (box, leaf test, cut) itself from the same combined data, so they agree bit for bit; box min / max and the
// integer histogram do not depend on the order of combination, and the misplaced-item lists get their libstdc++ order
// from per-chunk counts: chunk c's "bad" left items start after those of chunks < c, its "good" right items (listed right
// to left) after those of chunks > c.  Result: the same idx permutation and the same nodes as k_build_level.
struct BigScratch {
  double *part_box;    // [blocks][6]
  uint32_t *hist;      // [nodes of the level][6 * kMaxBins], zeroed before the launch
  uint32_t *part_cnt;  // [blocks][3]: left items, bad items in the left part, good items in the right part
  uint32_t *bar;       // grid barrier counter, zeroed before the launch
};

__device__ __forceinline__ void grid_barrier(uint32_t *bar, uint32_t nblocks, uint32_t &gen) {
  __threadfence(); // release this workgroup's writes to the device
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t target = (++gen) * nblocks;
    atomicAdd(bar, 1u);
    while (atomicAdd(bar, 0u) < target) __builtin_amdgcn_s_sleep(2);
  }
  __syncthreads();
  __threadfence(); // acquire: the other workgroups may sit on another XCD (another L2)
}

__global__ __launch_bounds__(kBBMax) void k_build_level_big(BNode *__restrict__ nodes, uint32_t level_begin, uint32_t level_count,
                                                            uint32_t G, const TriRec *__restrict__ rec,
                                                            uint32_t *__restrict__ idx, uint32_t *__restrict__ lpos,
                                                            uint32_t *__restrict__ rpos, uint32_t *__restrict__ next_count,
                                                            uint32_t next_begin, uint32_t max_nodes, BuildOpts opt,
                                                            BigScratch S) {
  constexpr int kBB = kBBMax;
  __shared__ double s_red[kBB / 64][6];
  __shared__ double s_box[6];
  __shared__ uint32_t s_bins[2 * 3 * kMaxBins];
  __shared__ uint32_t s_wave[kBB / 64];
  __shared__ double s_cost[3], s_cut[3];
  __shared__ uint32_t s_u[4];
  const uint32_t nblocks = level_count * G;
  const uint32_t node_k = blockIdx.x / G, part = blockIdx.x % G;
  const uint32_t nid = level_begin + node_k;
  BNode *nd = nodes + nid;
  const uint32_t l = nd->l, r = nd->r, n = r - l;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  uint32_t gen = 0;
  // this workgroup's chunk [a, b) of [l, r): equal chunks, a multiple of the workgroup size long
  const uint32_t ch = (((n + G - 1) / G + kBB - 1) / kBB) * kBB;
  const uint32_t a = (l + (unsigned long long)part * ch < r) ? l + part * ch : r;
  const uint32_t b = (a + (unsigned long long)ch < r) ? a + ch : r;

  // ---- 1. bounds: chunk partial, then every workgroup of the node combines the G partials -----------------------------
  double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
  for (uint32_t i = a + tid; i < b; i += kBB) {
    const TriRec &t = rec[idx[i]];
    for (int k = 0; k < 3; ++k) {
      lo[k] = (t.lo[k] < lo[k]) ? t.lo[k] : lo[k];
      hi[k] = (hi[k] < t.hi[k]) ? t.hi[k] : hi[k];
    }
  }
  for (int off = 32; off; off >>= 1)
    for (int k = 0; k < 3; ++k) {
      const double x = __shfl_down(lo[k], off), y = __shfl_down(hi[k], off);
      lo[k] = (x < lo[k]) ? x : lo[k];
      hi[k] = (hi[k] < y) ? y : hi[k];
    }
  if (lane == 0)
    for (int k = 0; k < 3; ++k) { s_red[wave][k] = lo[k]; s_red[wave][3 + k] = hi[k]; }
  __syncthreads();
  if (tid < 6) {
    double v = s_red[0][tid];
    for (int w = 1; w < kBB / 64; ++w) {
      const double x = s_red[w][tid];
      v = (tid < 3) ? ((x < v) ? x : v) : ((v < x) ? x : v);
    }
    S.part_box[(size_t)blockIdx.x * 6 + tid] = v;
  }
  grid_barrier(S.bar, nblocks, gen);
  if (tid < 6) {
    const double *pb = S.part_box + (size_t)node_k * G * 6;
    double v = __builtin_nontemporal_load(&pb[tid]);
    for (uint32_t g = 1; g < G; ++g) {
      const double x = __builtin_nontemporal_load(&pb[(size_t)g * 6 + tid]);
      v = (tid < 3) ? ((x < v) ? x : v) : ((v < x) ? x : v);
    }
    v = (tid < 3) ? v - kPad : v + kPad;
    s_box[tid] = v;
    if (part == 0) { if (tid < 3) nd->bmin[tid] = v; else nd->bmax[tid - 3] = v; }
  }
  __syncthreads();
  const double bmin[3] = {s_box[0], s_box[1], s_box[2]}, bmax[3] = {s_box[3], s_box[4], s_box[5]};

  // ---- 2. leaf?  (uniform over the node's workgroups; a leaf's workgroups keep attending the barriers) ----------------
  const bool is_leaf = (n < (uint32_t)opt.minLeaf || nd->depth >= opt.maxDepth);
  if (is_leaf && part == 0 && tid == 0) {
    nd->leaf = 1;
    nd->axis = 0;
    nd->child[0] = nd->child[1] = 0xFFFFFFFFu;
  }

  // ---- 3. histogram: chunk histogram in LDS, added into the node's -----------------------------------------------------
  const int nb = opt.binSize;
  const double fb = (double)nb;
  for (int i = tid; i < 6 * nb; i += kBB) s_bins[i] = 0;
  __syncthreads();
  double scale[3];
  for (int k = 0; k < 3; ++k) {
    const double ext = bmax[k] - bmin[k];
    scale[k] = (ext > kPad) ? fb / ext : 0.0;
  }
  if (!is_leaf)
    for (uint32_t i = a + tid; i < b; i += kBB) {
      const TriRec &t = rec[idx[i]];
      for (int k = 0; k < 3; ++k) {
        unsigned long long ilo = (unsigned int)floor((t.lo[k] - bmin[k]) * scale[k]);
        unsigned long long ihi = (unsigned int)floor((t.hi[k] - bmin[k]) * scale[k]);
        if ((double)ilo >= fb) ilo = (unsigned long long)(fb - 1);
        if ((double)ihi >= fb) ihi = (unsigned long long)(fb - 1);
        atomicAdd(&s_bins[k * nb + (int)ilo], 1u);
        atomicAdd(&s_bins[(3 + k) * nb + (int)ihi], 1u);
      }
    }
  __syncthreads();
  uint32_t *gh = S.hist + (size_t)node_k * 6 * kMaxBins;
  if (!is_leaf)
    for (int i = tid; i < 6 * nb; i += kBB)
      if (s_bins[i]) atomicAdd(&gh[i], s_bins[i]);
  grid_barrier(S.bar, nblocks, gen);
  for (int i = tid; i < 6 * nb; i += kBB) s_bins[i] = __builtin_nontemporal_load(&gh[i]);
  __syncthreads();

  // ---- 4. cut (bvh_accel.cc:156-255), evaluated by every workgroup of the node from the same histogram ----------------
  if (tid < 3) {
    const int j = tid;
    const double Ta = opt.costTaabb, Tt = 1.0 - opt.costTaabb;
    const double total = box_area(bmin, bmax);
    const double inv_total = (total > kPad) ? 1.0 / total : 0.0;
    const double step = (bmax[j] - bmin[j]) * (1.0 / nb);
    double best_pos = bmin[j] + 0.5 * step, best_cost = DBL_MAX;
    double loL[3] = {bmin[0], bmin[1], bmin[2]}, hiL[3] = {bmax[0], bmax[1], bmax[2]};
    double loR[3] = {bmin[0], bmin[1], bmin[2]}, hiR[3] = {bmax[0], bmax[1], bmax[2]};
    unsigned long long nl = 0, nr = n;
    for (int i = 0; i < nb - 1; ++i) {
      nl += s_bins[j * nb + i];
      nr -= s_bins[(3 + j) * nb + i];
      const double pos = bmin[j] + (i + 0.5) * step;
      hiL[j] = pos;
      loR[j] = pos;
      const double cost = 2.0 * Ta + (box_area(loL, hiL) * inv_total) * (double)nl * Tt +
                          (box_area(loR, hiR) * inv_total) * (double)nr * Tt;
      if (cost < best_cost) { best_cost = cost; best_pos = pos; }
    }
    s_cost[j] = best_cost;
    s_cut[j] = best_pos;
  }
  __syncthreads();
  int axis = 0;
  {
    double c = s_cost[0];
    if (c > s_cost[1]) { axis = 1; c = s_cost[1]; }
    if (c > s_cost[2]) { axis = 2; }
  }
  const double pos3 = s_cut[axis] * 3.0; // SAHPred: centre sum < pos * 3.0

  // ---- 5a. m = number of items that go left: chunk counts, summed by everybody ------------------------------------------
  uint32_t cnt = 0;
  if (!is_leaf)
    for (uint32_t i = a + tid; i < b; i += kBB) cnt += (rec[idx[i]].csum[axis] < pos3) ? 1u : 0u;
  for (int off = 32; off; off >>= 1) cnt += __shfl_down(cnt, off);
  __syncthreads();
  if (lane == 0) s_wave[wave] = cnt;
  __syncthreads();
  if (tid == 0) {
    uint32_t c = 0;
    for (int w = 0; w < kBB / 64; ++w) c += s_wave[w];
    S.part_cnt[(size_t)blockIdx.x * 3 + 0] = c;
  }
  grid_barrier(S.bar, nblocks, gen);
  const uint32_t *pc = S.part_cnt + (size_t)node_k * G * 3;
  if (tid == 0) {
    uint32_t m = 0;
    for (uint32_t g = 0; g < G; ++g) m += __builtin_nontemporal_load(&pc[(size_t)g * 3 + 0]);
    s_u[0] = m;
  }
  __syncthreads();
  const uint32_t m = s_u[0];
  const bool degenerate = (m == 0 || m == n);
  const uint32_t mid = degenerate ? l + (n >> 1) : l + m; // degenerate: order untouched, object median (bvh_accel.cc:405-409)

  // ---- 5b/5c. this chunk's misplaced items: counts first, then (after everybody's counts are known) their list slots ----
  const uint32_t la = a < mid ? a : mid, lb = b < mid ? b : mid; // chunk ∩ [l, mid)
  const uint32_t ra = a > mid ? a : mid, rb = b > mid ? b : mid; // chunk ∩ [mid, r)
  uint32_t nbad = 0, ngood = 0;
  if (!is_leaf && !degenerate) {
    for (uint32_t i = la + tid; i < lb; i += kBB) nbad += !(rec[idx[i]].csum[axis] < pos3) ? 1u : 0u;
    for (uint32_t i = ra + tid; i < rb; i += kBB) ngood += (rec[idx[i]].csum[axis] < pos3) ? 1u : 0u;
  }
  for (int off = 32; off; off >>= 1) { nbad += __shfl_down(nbad, off); ngood += __shfl_down(ngood, off); }
  __syncthreads();
  if (lane == 0) { s_wave[wave] = nbad; s_red[wave][0] = (double)ngood; }
  __syncthreads();
  if (tid == 0) {
    uint32_t x = 0, y = 0;
    for (int w = 0; w < kBB / 64; ++w) { x += s_wave[w]; y += (uint32_t)s_red[w][0]; }
    S.part_cnt[(size_t)blockIdx.x * 3 + 1] = x;
    S.part_cnt[(size_t)blockIdx.x * 3 + 2] = y;
  }
  grid_barrier(S.bar, nblocks, gen);
  if (tid == 0) {
    uint32_t loff = 0, roff = 0, nswap = 0;
    for (uint32_t g = 0; g < G; ++g) {
      const uint32_t x = __builtin_nontemporal_load(&pc[(size_t)g * 3 + 1]);
      const uint32_t y = __builtin_nontemporal_load(&pc[(size_t)g * 3 + 2]);
      if (g < part) loff += x;
      if (g > part) roff += y;
      nswap += x;
    }
    s_u[1] = loff; s_u[2] = roff; s_u[3] = nswap;
  }
  __syncthreads();
  const uint32_t nswap = s_u[3];
  if (!is_leaf && !degenerate) {
    uint32_t carry = s_u[1];
    for (uint32_t base = la; base < lb; base += kBB) { // left to right
      const uint32_t i = base + tid;
      const uint32_t bad = (i < lb && !(rec[idx[i]].csum[axis] < pos3)) ? 1u : 0u;
      uint32_t tot;
      const uint32_t k = block_scan<kBB>(bad, s_wave, tot);
      if (bad) lpos[l + carry + k] = i;
      carry += tot;
    }
    carry = s_u[2];
    for (uint32_t off = 0; off < rb - ra; off += kBB) { // right to left
      const uint32_t o = off + tid;
      const bool in = o < rb - ra;
      const uint32_t i = in ? (rb - 1 - o) : 0;
      const uint32_t good = (in && (rec[idx[i]].csum[axis] < pos3)) ? 1u : 0u;
      uint32_t tot;
      const uint32_t k = block_scan<kBB>(good, s_wave, tot);
      if (good) rpos[l + carry + k] = i;
      carry += tot;
    }
  }
  grid_barrier(S.bar, nblocks, gen);
  // ---- 5d. swap the pairs, shared out over the node's workgroups ----------------------------------------------------------
  if (!is_leaf && !degenerate)
    for (uint32_t k = part * kBB + tid; k < nswap; k += G * kBB) {
      const uint32_t pa = __builtin_nontemporal_load(&lpos[l + k]), pb = __builtin_nontemporal_load(&rpos[l + k]);
      const uint32_t va = idx[pa], vb = idx[pb];
      idx[pa] = vb;
      idx[pb] = va;
    }

  // ---- 6. children ------------------------------------------------------------------------------------------------------
  if (!is_leaf && part == 0 && tid == 0) {
    const uint32_t slot = atomicAdd(next_count, 2u);
    const uint32_t c0 = next_begin + slot, c1 = c0 + 1;
    if (c1 >= max_nodes) {
      nd->leaf = 1;
      nd->child[0] = nd->child[1] = 0xFFFFFFFFu;
      return;
    }
    nd->leaf = 0;
    nd->axis = (int8_t)axis;
    nd->child[0] = c0;
    nd->child[1] = c1;
    BNode x;
    memset(&x, 0, sizeof(x));
    x.parent = nid;
    x.depth = (int16_t)(nd->depth + 1);
    BNode y = x;
    x.l = l; x.r = mid;
    y.l = mid; y.r = r;
    nodes[c0] = x;
    nodes[c1] = y;
  }
}

__global__ void k_subtree(BNode *nodes, uint32_t begin, uint32_t count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  BNode &n = nodes[begin + i];
  n.subtree = n.leaf ? 1u : 1u + nodes[n.child[0]].subtree + nodes[n.child[1]].subtree;
}

__global__ void k_preorder(BNode *nodes, uint32_t begin, uint32_t count) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const BNode &n = nodes[begin + i];
  if (!n.leaf) {
    nodes[n.child[0]].pre = n.pre + 1;
    nodes[n.child[1]].pre = n.pre + 1 + nodes[n.child[0]].subtree;
  }
}

__global__ void k_emit(const BNode *__restrict__ nodes, uint32_t count, MgpuNode *__restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const BNode &n = nodes[i];
  MgpuNode o;
  for (int k = 0; k < 3; ++k) { o.bmin[k] = n.bmin[k]; o.bmax[k] = n.bmax[k]; }
  o.flag = n.leaf ? 1 : 0;
  o.axis = n.leaf ? 0 : n.axis;
  if (n.leaf) {
    o.data[0] = n.r - n.l;
    o.data[1] = n.l;
  } else {
    o.data[0] = nodes[n.child[0]].pre;
    o.data[1] = nodes[n.child[1]].pre;
  }
  out[n.pre] = o;
}

struct DevBuf {
  void *p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
  template <typename T> T *as() { return (T *)p; }
};

} // namespace

#define B_TRY(expr)                                                                                        \
  do {                                                                                                     \
    hipError_t e_ = (expr);                                                                                \
    if (e_ != hipSuccess) {                                                                                \
      fprintf(stderr, "mgpu_bvh_build_device: %s failed: %s\n", #expr, hipGetErrorString(e_));            \
      return MGPU_ERR_HIP;                                                                                 \
    }                                                                                                      \
  } while (0)

extern "C" int mgpu_bvh_build_device(const double *verts, size_t nv, const uint32_t *faces, size_t nf, double costTaabb,
                                     int minLeafPrimitives, int maxTreeDepth, int binSize, int device,
                                     MgpuNode **nodes_out, size_t *nn_out, uint32_t **indices_out, int stats[3],
                                     double *device_ms) {
  if (!verts || !faces || !nodes_out || !nn_out || !indices_out || nf == 0 || nv == 0) return MGPU_ERR_INVALID;
  if (binSize < 2 || binSize > kMaxBins || nf > 0x7FFFFFF0ull || maxTreeDepth > 32000) return MGPU_ERR_INVALID;
  if (device < 0 || device >= mgpu_device_count()) return MGPU_ERR_NO_DEVICE;
  B_TRY(hipSetDevice(device));
  const auto t0 = std::chrono::steady_clock::now();
  const size_t max_nodes = 2 * nf + 2; // a binary tree over nf non-empty leaves has at most 2*nf - 1 nodes
  DevBuf d_verts, d_faces, d_rec, d_idx, d_lpos, d_rpos, d_nodes, d_cnt, d_out;
  B_TRY(d_verts.alloc(sizeof(double) * 3 * nv));
  B_TRY(d_faces.alloc(sizeof(uint32_t) * 3 * nf));
  B_TRY(d_rec.alloc(sizeof(TriRec) * nf));
  B_TRY(d_idx.alloc(sizeof(uint32_t) * nf));
  B_TRY(d_lpos.alloc(sizeof(uint32_t) * nf));
  B_TRY(d_rpos.alloc(sizeof(uint32_t) * nf));
  B_TRY(d_nodes.alloc(sizeof(BNode) * max_nodes));
  B_TRY(d_cnt.alloc(sizeof(uint32_t)));
  // scratch of the multi-workgroup levels (k_build_level_big): at most kBigBlocks workgroups per launch
  constexpr uint32_t kBigBlocks = 256;
  DevBuf d_pbox, d_hist, d_pcnt, d_bar;
  B_TRY(d_pbox.alloc(sizeof(double) * 6 * kBigBlocks));
  B_TRY(d_hist.alloc(sizeof(uint32_t) * 6 * kMaxBins * kBigBlocks));
  B_TRY(d_pcnt.alloc(sizeof(uint32_t) * 3 * kBigBlocks));
  B_TRY(d_bar.alloc(sizeof(uint32_t)));
  int num_cu = 0;
  {
    hipDeviceProp_t prop;
    B_TRY(hipGetDeviceProperties(&prop, device));
    num_cu = prop.multiProcessorCount;
  }
  B_TRY(hipMemcpy(d_verts.p, verts, sizeof(double) * 3 * nv, hipMemcpyHostToDevice));
  B_TRY(hipMemcpy(d_faces.p, faces, sizeof(uint32_t) * 3 * nf, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  B_TRY(hipEventCreate(&e0));
  B_TRY(hipEventCreate(&e1));
  B_TRY(hipEventRecord(e0, 0));
  hipLaunchKernelGGL(k_tri_records, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, 0, d_verts.as<double>(),
                     d_faces.as<uint32_t>(), nf, d_rec.as<TriRec>(), d_idx.as<uint32_t>());
  BNode root;
  memset(&root, 0, sizeof(root));
  root.l = 0;
  root.r = (uint32_t)nf;
  root.parent = 0xFFFFFFFFu;
  B_TRY(hipMemcpy(d_nodes.p, &root, sizeof(root), hipMemcpyHostToDevice));
  BuildOpts opt{costTaabb, minLeafPrimitives, maxTreeDepth, binSize};
  std::vector<uint32_t> level_begin, level_count;
  uint32_t begin = 0, count = 1;
  while (count) {
    level_begin.push_back(begin);
    level_count.push_back(count);
    if ((size_t)begin + count > max_nodes) {
      fprintf(stderr, "mgpu_bvh_build_device: node budget exceeded\n");
      return MGPU_ERR_INVALID;
    }
    B_TRY(hipMemsetAsync(d_cnt.p, 0, sizeof(uint32_t), 0));
    // workgroup size by the level's average node size: one 1024-thread group per node wastes 1000 threads on the
    // hundreds of thousands of 16..100-triangle nodes near the leaves (10 M triangles: levels 14-24 took 106 of 283 ms)
    const size_t avg = nf / count;
#define LAUNCH_LEVEL(BB)                                                                                              \
  hipLaunchKernelGGL(k_build_level<BB>, dim3(count), dim3(BB), 0, 0, d_nodes.as<BNode>(), begin, count,                  \
                     d_rec.as<TriRec>(), d_idx.as<uint32_t>(), d_lpos.as<uint32_t>(), d_rpos.as<uint32_t>(),             \
                     d_cnt.as<uint32_t>(), begin + count, (uint32_t)max_nodes, opt)
    // all workgroups of a multi-workgroup level must be resident at once (they meet at grid barriers): one per CU at most
    const uint32_t big_cap = (uint32_t)((num_cu < (int)kBigBlocks) ? num_cu : (int)kBigBlocks);
    uint32_t G = 0;
    if (avg >= 65536 && count <= big_cap / 2 && !getenv("MGPU_BVH_NO_BIG")) {
      G = big_cap / count;
      const size_t want = (avg + 16383) / 16384; // no point in chunks below ~16 k items
      if (G > want) G = (uint32_t)want;
    }
    if (G >= 2) {
      B_TRY(hipMemsetAsync(d_hist.p, 0, sizeof(uint32_t) * 6 * kMaxBins * count, 0));
      B_TRY(hipMemsetAsync(d_bar.p, 0, sizeof(uint32_t), 0));
      BigScratch S{d_pbox.as<double>(), d_hist.as<uint32_t>(), d_pcnt.as<uint32_t>(), d_bar.as<uint32_t>()};
      hipLaunchKernelGGL(k_build_level_big, dim3(count * G), dim3(kBBMax), 0, 0, d_nodes.as<BNode>(), begin, count, G,
                         d_rec.as<TriRec>(), d_idx.as<uint32_t>(), d_lpos.as<uint32_t>(), d_rpos.as<uint32_t>(),
                         d_cnt.as<uint32_t>(), begin + count, (uint32_t)max_nodes, opt, S);
    } else if (avg >= 4096) LAUNCH_LEVEL(1024);
    else if (avg >= 256) LAUNCH_LEVEL(256);
    else LAUNCH_LEVEL(64);
#undef LAUNCH_LEVEL
    B_TRY(hipGetLastError());
    uint32_t next = 0;
    B_TRY(hipMemcpy(&next, d_cnt.p, sizeof(uint32_t), hipMemcpyDeviceToHost));
    begin += count;
    count = next;
  }
  const uint32_t nn = begin;
  for (size_t lv = level_begin.size(); lv-- > 0;)
    hipLaunchKernelGGL(k_subtree, dim3((level_count[lv] + 255) / 256), dim3(256), 0, 0, d_nodes.as<BNode>(), level_begin[lv],
                       level_count[lv]);
  for (size_t lv = 0; lv < level_begin.size(); ++lv)
    hipLaunchKernelGGL(k_preorder, dim3((level_count[lv] + 255) / 256), dim3(256), 0, 0, d_nodes.as<BNode>(),
                       level_begin[lv], level_count[lv]);
  B_TRY(d_out.alloc(sizeof(MgpuNode) * nn));
  hipLaunchKernelGGL(k_emit, dim3((nn + 255) / 256), dim3(256), 0, 0, d_nodes.as<BNode>(), nn, d_out.as<MgpuNode>());
  B_TRY(hipGetLastError());
  B_TRY(hipEventRecord(e1, 0));
  B_TRY(hipEventSynchronize(e1));
  float ms = 0.f;
  B_TRY(hipEventElapsedTime(&ms, e0, e1));
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  MgpuNode *hn = (MgpuNode *)malloc(sizeof(MgpuNode) * nn);
  uint32_t *hi = (uint32_t *)malloc(sizeof(uint32_t) * nf);
  if (!hn || !hi) {
    free(hn);
    free(hi);
    return MGPU_ERR_OOM;
  }
  if (hipMemcpy(hn, d_out.p, sizeof(MgpuNode) * nn, hipMemcpyDeviceToHost) != hipSuccess ||
      hipMemcpy(hi, d_idx.p, sizeof(uint32_t) * nf, hipMemcpyDeviceToHost) != hipSuccess) {
    free(hn);
    free(hi);
    return MGPU_ERR_HIP;
  }
  *nodes_out = hn;
  *nn_out = nn;
  *indices_out = hi;
  if (stats) {
    int leaves = 0;
    for (uint32_t i = 0; i < nn; ++i) leaves += hn[i].flag;
    stats[0] = (int)level_begin.size() - 1;
    stats[1] = leaves;
    stats[2] = (int)nn - leaves;
  }
  if (device_ms) *device_ms = ms;
  (void)t0;
  return MGPU_OK;
}
