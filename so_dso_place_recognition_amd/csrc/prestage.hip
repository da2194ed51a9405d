// prestage.hip — GPU pre-stage of both executables (utils/pts_preprocess.h:135-232).
//
// Reference, per pose in file order: append the history points whose id <= pose id to the "nearby" set; transform the set to
// the camera frame; keep the points with |p| < lidarRange (the others leave the set for good); down-sample to the best point
// per cell (voxel grid: smallest camera-y, polar 1 deg x 1 deg: smallest norm, ties: the earlier point); emit the survivors in
// the iteration order of the std::unordered_map<int,...> that was keyed by cell.  A reset pose (|t| < 1) clears the set
// and the next 30 poses emit nothing.
//
// The set evolution is sequential per POSE but independent per POINT, so the GPU formulation is point-major:
//   death      one thread per history point: walk the poses from its birth pose on, first emitting pose where it is out
//              of range = its death (or the next reset).  Alive at pose p  <=>  birth <= p < death.
//   members    per emitting pose, ordered compaction of the alive points (ascending point index = the reference's set
//              order, because the set is append-only and pruning keeps relative order)
//   cells      per (pose, member): camera-frame point, cell id, ordering value; per (pose, cell) atomicMin of the value and
//              of the first member index (= the insertion order of the cell key), then atomicMin of the member index among
//              the members that hold the minimum value (ties -> earlier point, like the strict comparison :78-80 / :117-119)
//   keys       per pose, ordered compaction of the members that are the first of their cell: the key insertion sequence
//   order      one LANE per pose: libstdc++ node-list order of that insertion sequence (hash_order.hpp)
//   gather     per output point: winner of its cell, transformed again, with its intensity
// All arithmetic that decides membership or cells is the reference's expression order in fp64 (compiled with
// -ffp-contract=off); the only declared deviation is device atan2 vs glibc atan2 for a point within an ulp of a polar
// cell edge.  Dense (pose, cell) tables are processed in batches of poses sized to a memory budget.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "frames.hpp"
#include "hash_order.hpp"
#include "kernels.hpp"

namespace pr {
namespace {

__device__ __forceinline__ bool to_camera(const double* __restrict__ w, const double* __restrict__ g, double range, double* l) {
#pragma unroll
  for (int r = 0; r < 3; r++) l[r] = ((w[4 * r] * g[0] + w[4 * r + 1] * g[1]) + w[4 * r + 2] * g[2]) + w[4 * r + 3] * 1.0;   // :141-142
  const double nrm = sqrt((l[0] * l[0] + l[1] * l[1]) + l[2] * l[2]);
  return nrm < range;                                                                                                   // :144
}

__global__ __launch_bounds__(256) void death_kernel(const double* __restrict__ xyz, const int* __restrict__ birth, int64_t T,
                                                    const double* __restrict__ W, const unsigned char* __restrict__ emit,
                                                    const int* __restrict__ next_reset, double range, int* __restrict__ death) {
  const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (j >= T) return;
  const double g[3] = {xyz[3 * j], xyz[3 * j + 1], xyz[3 * j + 2]};
  const int b = birth[j];
  int d = (b >= 0) ? next_reset[b] : -1;     // birth -1: never appended (id larger than every pose id)
  for (int p = b; b >= 0 && p < d; p++) {
    if (!emit[p]) continue;
    double l[3];
    if (!to_camera(W + 12 * (size_t)p, g, range, l)) { d = p; break; }
  }
  death[j] = d;
}

// block-wide exclusive scan of 0/1 flags (256 threads = 4 waves); returns this thread's rank and the block total
__device__ __forceinline__ int block_rank(bool flag, int* total) {
  __shared__ int wsum[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const unsigned long long m = __ballot(flag);
  const int r = __popcll(m & ((1ull << lane) - 1ull));
  if (lane == 0) wsum[w] = __popcll(m);
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int i = 0; i < 4; i++) { if (i < w) base += wsum[i]; tot += wsum[i]; }
  __syncthreads();
  *total = tot;
  return base + r;
}

// members of emitting pose e: alive points of [first_alive[e], cursor[e]) in ascending index.  list == nullptr: count only.
__global__ __launch_bounds__(256) void members_kernel(const int* __restrict__ death, const int* __restrict__ pose_of,
                                                      const int64_t* __restrict__ first_alive, const int64_t* __restrict__ cursor,
                                                      const int64_t* __restrict__ off, int* __restrict__ cnt, int* __restrict__ list) {
  const int e = blockIdx.x, p = pose_of[e];
  const int64_t lo = first_alive[e], hi = cursor[e];
  int64_t base = list ? off[e] : 0;
  int n = 0;
  for (int64_t c = lo; c < hi; c += 256) {
    const int64_t j = c + threadIdx.x;
    const bool alive = j < hi && p < death[j];
    int tot;
    const int r = block_rank(alive, &tot);
    if (list && alive) list[base + r] = (int)j;
    base += tot;
    n += tot;
  }
  if (!list && threadIdx.x == 0) cnt[e] = n;
}

__device__ __forceinline__ unsigned long long orderable(double v) {   // monotone map double -> u64 (-0.0 == +0.0)
  v = v + 0.0;
  const unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

struct Grid { double range, step[3]; int dim[3]; int polar, azi_bins; double inv; };

__device__ __forceinline__ void cell_of(const Grid& g, const double* l, int* cell, unsigned long long* val) {
  if (g.polar) {                                                            // :100-119
    const double xz = sqrt(l[0] * l[0] + l[2] * l[2]);
    const int azi = (int)floor((atan2(l[2], l[0]) + M_PI) * g.inv);
    const int ele = (int)floor((atan2(l[1], xz) + M_PI / 2) * g.inv);
    *cell = azi + ele * g.azi_bins;
    *val = orderable(sqrt((l[0] * l[0] + l[1] * l[1]) + l[2] * l[2]));
  } else {                                                                  // :55-80
    const int xi = (int)floor((l[0] + g.range) * g.step[0]);
    const int yi = (int)floor((l[1] + g.range) * g.step[1]);
    const int zi = (int)floor((l[2] + g.range) * g.step[2]);
    *cell = xi + yi * g.dim[0] + zi * g.dim[0] * g.dim[1];
    *val = orderable(l[1]);
  }
}

__device__ __forceinline__ int pose_ordinal(const int64_t* __restrict__ off, int e0, int e1, int64_t s) {   // off[e] <= s < off[e+1]
  int lo = e0, hi = e1;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (off[mid] <= s) lo = mid; else hi = mid;
  }
  return lo;
}

// phase A over the members [s0, s1) of the poses [e0, e1): cell + value per member, per-cell min value and first member
__global__ __launch_bounds__(256) void cells_kernel(const double* __restrict__ xyz, const int* __restrict__ list,
                                                    const int64_t* __restrict__ off, const int* __restrict__ pose_of, int e0, int e1,
                                                    int64_t s0, int64_t s1, const double* __restrict__ W, Grid g, int64_t C,
                                                    int* __restrict__ cell_out, unsigned long long* __restrict__ val_out,
                                                    unsigned long long* __restrict__ tval, unsigned* __restrict__ tfirst) {
  const int64_t s = s0 + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= s1) return;
  const int e = pose_ordinal(off, e0, e1, s);
  const int j = list[s];
  const double gp[3] = {xyz[3 * (size_t)j], xyz[3 * (size_t)j + 1], xyz[3 * (size_t)j + 2]};
  double l[3];
  (void)to_camera(W + 12 * (size_t)pose_of[e], gp, g.range, l);
  int cell;
  unsigned long long v;
  cell_of(g, l, &cell, &v);
  if (cell < 0 || cell >= C) cell = 0;          // cannot happen for |p| < range; keeps a corrupt input from writing out of bounds
  cell_out[s] = cell;
  val_out[s] = v;
  const size_t t = (size_t)(e - e0) * C + cell;
  atomicMin(&tval[t], v);
  atomicMin(&tfirst[t], (unsigned)(s - off[e]));
}

// phase B: among the members that hold their cell's minimum, the earliest one wins
__global__ __launch_bounds__(256) void winners_kernel(const int64_t* __restrict__ off, int e0, int e1, int64_t s0, int64_t s1, int64_t C,
                                                      const int* __restrict__ cell, const unsigned long long* __restrict__ val,
                                                      const unsigned long long* __restrict__ tval, unsigned* __restrict__ tbest) {
  const int64_t s = s0 + (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (s >= s1) return;
  const int e = pose_ordinal(off, e0, e1, s);
  const size_t t = (size_t)(e - e0) * C + cell[s];
  if (val[s] == tval[t]) atomicMin(&tbest[t], (unsigned)(s - off[e]));
}

// key insertion sequence of pose e: the members that are the first of their cell, in member order; winner of each key
__global__ __launch_bounds__(256) void keys_kernel(const int64_t* __restrict__ off, int e0, int64_t C, const int* __restrict__ cell,
                                                   const int* __restrict__ list, const unsigned* __restrict__ tfirst,
                                                   const unsigned* __restrict__ tbest, int* __restrict__ keys, int* __restrict__ win,
                                                   int* __restrict__ nkeys) {
  const int e = e0 + blockIdx.x;
  const int64_t lo = off[e], hi = off[e + 1];
  int64_t base = lo;
  for (int64_t c = lo; c < hi; c += 256) {
    const int64_t s = c + threadIdx.x;
    bool first = false;
    size_t t = 0;
    if (s < hi) {
      t = (size_t)(e - e0) * C + cell[s];
      first = tfirst[t] == (unsigned)(s - lo);
    }
    int tot;
    const int r = block_rank(first, &tot);
    if (first) {
      keys[base + r] = cell[s];
      win[base + r] = list[lo + tbest[t]];
    }
    base += tot;
  }
  if (threadIdx.x == 0) nkeys[e] = (int)(base - lo);
}

__global__ __launch_bounds__(64) void order_kernel(int E, const int64_t* __restrict__ off, const int* __restrict__ nkeys,
                                                   const int* __restrict__ keys, const int* __restrict__ sched_cnt,
                                                   const int* __restrict__ sched_nb, int nsched, int* __restrict__ next,
                                                   const int64_t* __restrict__ boff, int* __restrict__ bkt, int* __restrict__ order) {
  const int e = blockIdx.x * 64 + threadIdx.x;
  if (e >= E) return;
  hash_order(keys + off[e], nkeys[e], sched_cnt, sched_nb, nsched, next + off[e], bkt + boff[e], order + off[e]);
}

__global__ __launch_bounds__(256) void gather_kernel(int E, const int64_t* __restrict__ off, const int64_t* __restrict__ ooff,
                                                     const int* __restrict__ pose_of, const int* __restrict__ order,
                                                     const int* __restrict__ win, const double* __restrict__ xyz,
                                                     const float* __restrict__ inten, const double* __restrict__ W, double range,
                                                     double* __restrict__ oxyz, float* __restrict__ oint) {
  const int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (o >= ooff[E]) return;
  const int e = pose_ordinal(ooff, 0, E, o);
  const int k = order[off[e] + (o - ooff[e])];
  const int j = win[off[e] + k];
  const double gp[3] = {xyz[3 * (size_t)j], xyz[3 * (size_t)j + 1], xyz[3 * (size_t)j + 2]};
  double l[3];
  (void)to_camera(W + 12 * (size_t)pose_of[e], gp, range, l);
  oxyz[3 * o] = l[0]; oxyz[3 * o + 1] = l[1]; oxyz[3 * o + 2] = l[2];
  oint[o] = inten[j];
}

// The same emission with one workgroup per cloud, which also adds up the cloud's raw moments on the way and leaves its PCA frame
// (pts_align.h:7-46): thread t emits points t, t + 256, ... - the order in which cloud_frames_kernel (sc_gen.hip) reads them - and both
// use reduce_moments_to_frame, so frames[e] has the bits a moments pass over the emitted cloud would produce; the float intensity average
// is added too, and the generators then run their binning pass only (pr_*_generate_frames_dev).
__global__ __launch_bounds__(FRAME_THREADS) void gather_frames_kernel(const int64_t* __restrict__ off, const int64_t* __restrict__ ooff,
                                                                      const int* __restrict__ pose_of, const int* __restrict__ order,
                                                                      const int* __restrict__ win, const double* __restrict__ xyz,
                                                                      const float* __restrict__ inten, const double* __restrict__ W,
                                                                      double range, double* __restrict__ oxyz, float* __restrict__ oint,
                                                                      double* __restrict__ frames) {
  __shared__ double red[FRAME_THREADS / 64][9];
  __shared__ __attribute__((aligned(16))) float stage[2][2048 + 32];
  const int e = blockIdx.x, tid = threadIdx.x;
  const int64_t o0 = ooff[e], P = ooff[e + 1] - o0, m0 = off[e];
  const double* w = W + 12 * (size_t)pose_of[e];
  double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
  for (int64_t i = tid; i < P; i += FRAME_THREADS) {
    const int j = win[m0 + order[m0 + i]];
    const double gp[3] = {xyz[3 * (size_t)j], xyz[3 * (size_t)j + 1], xyz[3 * (size_t)j + 2]};
    double l[3];
    (void)to_camera(w, gp, range, l);
    const int64_t o = o0 + i;
    oxyz[3 * o] = l[0]; oxyz[3 * o + 1] = l[1]; oxyz[3 * o + 2] = l[2];
    oint[o] = inten[j];
    const double x = l[0], y = l[1], z = l[2];
    s[0] += x; s[1] += y; s[2] += z;
    s[3] += x * x; s[4] += x * y; s[5] += x * z; s[6] += y * y; s[7] += y * z; s[8] += z * z;
  }
  reduce_moments_to_frame(s, (double)P, red, frames + (size_t)e * 16);
  // ... and the reference's float average of the emitted intensities (in emission order), so that the generators have nothing left to do
  // but their binning pass: frames[e][14], [15] = 1
  __threadfence_block();
  __syncthreads();                                 // this workgroup's oint stores are visible to it
  const float a = block_sequential_average(oint + o0, P, stage);
  if (tid == 0) { frames[(size_t)e * 16 + 14] = (double)a; frames[(size_t)e * 16 + 15] = 1.0; }
}

Grid make_grid(double range, int polar) {
  Grid g;
  g.range = range;
  g.polar = polar;
  const double res[3] = {30, 60, 30};
  for (int a = 0; a < 3; a++) {                                 // :55-60
    const double r = range / res[a];
    g.step[a] = 1.0 / r;
    g.dim[a] = (int)(floor(2 * range * g.step[a]) + 1);
  }
  const double pres = 1.0 / 180.0 * M_PI;                       // :100-103
  g.inv = 1.0 / pres;
  g.azi_bins = (int)(floor(2 * M_PI * g.inv) + 1);
  return g;
}

}  // namespace

int64_t prestage_cells(double range, int polar) {
  const Grid g = make_grid(range, polar);
  if (polar) return (int64_t)g.azi_bins * ((int64_t)floor(M_PI * g.inv) + 2);
  return (int64_t)g.dim[0] * g.dim[1] * g.dim[2];
}

void launch_death(hipStream_t st, const double* xyz, const int* birth, int64_t T, const double* W, const unsigned char* emit,
                  const int* next_reset, double range, int* death) {
  if (T <= 0) return;
  hipLaunchKernelGGL(death_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, st, xyz, birth, T, W, emit, next_reset, range, death);
}
void launch_members(hipStream_t st, int E, const int* death, const int* pose_of, const int64_t* first_alive, const int64_t* cursor,
                    const int64_t* off, int* cnt, int* list) {
  if (E <= 0) return;
  hipLaunchKernelGGL(members_kernel, dim3(E), dim3(256), 0, st, death, pose_of, first_alive, cursor, off, cnt, list);
}
void launch_cells(hipStream_t st, const double* xyz, const int* list, const int64_t* off, const int* pose_of, int e0, int e1,
                  int64_t s0, int64_t s1, const double* W, double range, int polar, int64_t C, int* cell, unsigned long long* val,
                  unsigned long long* tval, unsigned* tfirst, unsigned* tbest) {
  if (s1 <= s0) return;
  const unsigned blocks = (unsigned)((s1 - s0 + 255) / 256);
  hipLaunchKernelGGL(cells_kernel, dim3(blocks), dim3(256), 0, st, xyz, list, off, pose_of, e0, e1, s0, s1, W, make_grid(range, polar), C,
                     cell, val, tval, tfirst);
  hipLaunchKernelGGL(winners_kernel, dim3(blocks), dim3(256), 0, st, off, e0, e1, s0, s1, C, cell, val, tval, tbest);
}
void launch_keys(hipStream_t st, const int64_t* off, int e0, int e1, int64_t C, const int* cell, const int* list, const unsigned* tfirst,
                 const unsigned* tbest, int* keys, int* win, int* nkeys) {
  if (e1 <= e0) return;
  hipLaunchKernelGGL(keys_kernel, dim3(e1 - e0), dim3(256), 0, st, off, e0, C, cell, list, tfirst, tbest, keys, win, nkeys);
}
void launch_order(hipStream_t st, int E, const int64_t* off, const int* nkeys, const int* keys, const int* sched_cnt, const int* sched_nb,
                  int nsched, int* next, const int64_t* boff, int* bkt, int* order) {
  if (E <= 0) return;
  hipLaunchKernelGGL(order_kernel, dim3((E + 63) / 64), dim3(64), 0, st, E, off, nkeys, keys, sched_cnt, sched_nb, nsched, next, boff, bkt, order);
}
void launch_gather(hipStream_t st, int E, int64_t total, const int64_t* off, const int64_t* ooff, const int* pose_of, const int* order,
                   const int* win, const double* xyz, const float* inten, const double* W, double range, double* oxyz, float* oint,
                   double* frames) {
  if (frames) {                 // one workgroup per cloud: the points and the cloud's PCA frame
    if (E > 0) hipLaunchKernelGGL(gather_frames_kernel, dim3(E), dim3(FRAME_THREADS), 0, st, off, ooff, pose_of, order, win, xyz, inten, W, range,
                                  oxyz, oint, frames);
    return;
  }
  if (total <= 0) return;
  hipLaunchKernelGGL(gather_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, E, off, ooff, pose_of, order, win, xyz, inten,
                     W, range, oxyz, oint);
}

}  // namespace pr
