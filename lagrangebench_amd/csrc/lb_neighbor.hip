// lb_neighbor.hip - per-step neighbor-list construction on gfx950.
//
// Replaces jax_sph.jax_md.partition.neighbor_list (.allocate/.update), which the reference calls
// at lagrangebench/case_setup/case.py:120-130,184-190, and the edge part of
// feature_transform (lagrangebench/case_setup/features.py:110-124).
//
// Design (not a translation of jax-md's dense (N, 3^dim*cap) candidate matrix):
//   1. k_cell_count / two-level scan / k_cell_fill : counting sort of the particles into cells
//      (int atomics); k_cell_fill also writes the cell-sorted fp64 positions so that a cell's
//      particles are one contiguous HBM range.
//   2. stencil search, two kernels with identical results:
//      k_nlw (3^3-cell stencils): one WAVE per receiver walks the particles of its stencil cells -
//        short contiguous runs of the cell-sorted arrays, L2 resident - 64 candidates per sweep
//        straight from global memory; no staging, 1.3 KiB of LDS per wave, full occupancy;
//      k_nl (3^2-cell stencils and the all-pairs case): one workgroup per cell stages the stencil's
//        particles (ids + fp64 positions) in LDS once (in 128-candidate capacity steps: the LDS
//        footprint sets the occupancy) and each wave sweeps the tile for the receivers of the cell.
//      In both the cutoff predicate is evaluated in fp64 exactly as the reference does
//      (metric(pos[sender], pos[receiver]) < r_c^2), reduced with a wavefront ballot + popcount
//      prefix; the row is compacted into LDS and rank-sorted by sender id with lane broadcasts.
//        update path (capacities frozen): ONE sweep writes each receiver's sorted row + its edge
//          features into fixed-stride per-node slots, then a two-level scan of the degrees gives the
//          CSR offsets and k_nl_compact moves the rows into place (pure streaming copy);
//        allocate path (sizes unknown): count sweep -> scan -> host sizes the buffers -> fill sweep.
//   3. k_row_finish : per-trajectory edge counts, did_buffer_overflow flags and the control block,
//      all on the device (no host sync).
// Output: CSR by receiver over the B*N nodes of the batch, senders ascending inside a row, i.e.
// the edge list sorted by (receiver, sender) - deterministic, and directly consumable by the
// atomic-free segmented aggregation.
#include <cstdlib>

#include "lb_features.h"

#define SCAN_THREADS 256
#define SCAN_CHUNK (SCAN_THREADS * 8)
#define NL_SMALL_MAXC 1024

enum { NL_COUNT = 0, NL_FILL = 1, NL_ROWS = 2 };

// ------------------------------------------------------------------------------------ cells
__global__ void k_cell_count(lb_geom g, int64_t BN, const double* __restrict__ win,
                             lb_ctrl* __restrict__ ctrl, int32_t* __restrict__ cell_of,
                             int32_t* __restrict__ cell_count) {
  if (ctrl->overflow_step >= 0) return;
  int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gi == 0) {
    // per-build maxima: only kernels launched AFTER this one (scan, stencil search) write them
    ctrl->max_cell_occ = 0;
    ctrl->max_deg = 0;
    ctrl->row_overflow = 0;
  }
  if (gi >= BN) return;
  const int step = ctrl->step;
  const int b = (int)(gi / g.N);
  int h = 0, mult = 1;
  for (int d = 0; d < g.dim; ++d) {
    double p = lb_pos(win, g, BN, step, g.isl - 1, d, gi);
    int c = __double2int_rz(lb_r(p / g.cell_size[d], g.f32));  // jnp.array(position / cell_size, dtype=i32)
    c = c < 0 ? 0 : (c >= g.ncell[d] ? g.ncell[d] - 1 : c);
    h += c * mult;
    mult *= g.ncell[d];
  }
  const int gc = b * g.ncells + h;
  cell_of[gi] = gc;
  atomicAdd(&cell_count[gc], 1);
}

// Two-level exclusive scan.  Pass 1: one partial sum (and max) per SCAN_CHUNK elements.
__global__ void __launch_bounds__(SCAN_THREADS)
    k_scan_partials(const int32_t* __restrict__ in, int n, int32_t* __restrict__ partial,
                    const lb_ctrl* __restrict__ ctrl, int32_t* __restrict__ max_out) {
  __shared__ int s_sum[SCAN_THREADS], s_mx[SCAN_THREADS];
  if (ctrl->overflow_step >= 0) return;
  const int t = threadIdx.x;
  const int lo = blockIdx.x * SCAN_CHUNK + t * 8;
  int sum = 0, mx = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int v = (lo + i < n) ? in[lo + i] : 0;
    sum += v;
    mx = max(mx, v);
  }
  s_sum[t] = sum;
  s_mx[t] = mx;
  __syncthreads();
  for (int off = SCAN_THREADS / 2; off > 0; off >>= 1) {
    if (t < off) {
      s_sum[t] += s_sum[t + off];
      s_mx[t] = max(s_mx[t], s_mx[t + off]);
    }
    __syncthreads();
  }
  if (t == 0) {
    partial[blockIdx.x] = s_sum[0];
    if (max_out) atomicMax(max_out, s_mx[0]);
  }
}

// Pass 2: every block sums the partials before it (its base) and scans its own chunk.
// out has n+1 entries; out[n] = total.
__global__ void __launch_bounds__(SCAN_THREADS)
    k_scan_apply(const int32_t* __restrict__ in, int32_t* __restrict__ out, int n,
                 const int32_t* __restrict__ partial, const lb_ctrl* __restrict__ ctrl) {
  __shared__ int s_a[SCAN_THREADS];
  if (ctrl->overflow_step >= 0) return;
  const int t = threadIdx.x, b = blockIdx.x;
  int base = 0;
  for (int i = t; i < b; i += SCAN_THREADS) base += partial[i];
  s_a[t] = base;
  __syncthreads();
  for (int off = SCAN_THREADS / 2; off > 0; off >>= 1) {
    if (t < off) s_a[t] += s_a[t + off];
    __syncthreads();
  }
  base = s_a[0];
  __syncthreads();
  const int lo = b * SCAN_CHUNK + t * 8;
  int v[8], sum = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    v[i] = (lo + i < n) ? in[lo + i] : 0;
    sum += v[i];
  }
  s_a[t] = sum;
  __syncthreads();
  for (int off = 1; off < SCAN_THREADS; off <<= 1) {  // Hillis-Steele inclusive
    const int add = (t >= off) ? s_a[t - off] : 0;
    __syncthreads();
    s_a[t] += add;
    __syncthreads();
  }
  int run = base + s_a[t] - sum;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (lo + i < n) out[lo + i] = run;
    run += v[i];
  }
  if (b == gridDim.x - 1 && t == SCAN_THREADS - 1) out[n] = base + s_a[t];
}

__global__ void k_cell_fill(lb_geom g, int64_t BN, const double* __restrict__ win,
                            const lb_ctrl* __restrict__ ctrl, const int32_t* __restrict__ cell_of,
                            const int32_t* __restrict__ cell_start, int32_t* __restrict__ cell_fill,
                            int32_t* __restrict__ cell_part, double* __restrict__ cpos) {
  if (ctrl->overflow_step >= 0) return;
  int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gi >= BN) return;
  const int step = ctrl->step;
  const int gc = cell_of[gi];
  const int slot = cell_start[gc] + atomicAdd(&cell_fill[gc], 1);
  cell_part[slot] = (int32_t)gi;
  for (int d = 0; d < g.dim; ++d) cpos[(int64_t)d * BN + slot] = lb_pos(win, g, BN, step, g.isl - 1, d, gi);
}

// Frozen update path, fixed-stride cells (round 6): cell c owns the slots [c * cap, (c + 1) * cap) of cell_part / cpos - jax-md's
// own layout.  Binning is two launches (zero the counters + the per-build maxima; one returning atomic per particle) instead
// of five (memset, count, two scan passes, fill).  A cell that receives more than `cap` particles drops the surplus and raises
// max_cell_occ above the capacity: did_buffer_overflow, the caller re-allocates (the CSR path keeps them and flags the same).
__global__ void k_cell_zero(lb_ctrl* __restrict__ ctrl, int32_t* __restrict__ cell_count, int n) {
  if (ctrl->overflow_step >= 0) return;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {   // per-build maxima: only kernels launched AFTER this one write them
    ctrl->max_cell_occ = 0;
    ctrl->max_deg = 0;
    ctrl->row_overflow = 0;
  }
  if (i < n) cell_count[i] = 0;
}
__global__ void __launch_bounds__(256) k_cell_bin(lb_geom g, int64_t BN, const double* __restrict__ win,
                                                  lb_ctrl* __restrict__ ctrl, int32_t* __restrict__ cell_of,
                                                  int32_t* __restrict__ cell_count, int32_t* __restrict__ cell_part,
                                                  double* __restrict__ cpos, int32_t cap, int64_t cstride) {
  if (ctrl->overflow_step >= 0) return;
  const int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int step = ctrl->step;
  int occ = 0;
  if (gi < BN) {
    const int b = (int)(gi / g.N);
    int h = 0, mult = 1;
    double p[3] = {0, 0, 0};
    for (int d = 0; d < g.dim; ++d) {
      p[d] = lb_pos(win, g, BN, step, g.isl - 1, d, gi);
      int c = __double2int_rz(lb_r(p[d] / g.cell_size[d], g.f32));  // jnp.array(position / cell_size, dtype=i32)
      c = c < 0 ? 0 : (c >= g.ncell[d] ? g.ncell[d] - 1 : c);
      h += c * mult;
      mult *= g.ncell[d];
    }
    const int gc = b * g.ncells + h;
    cell_of[gi] = gc;
    const int slot = atomicAdd(&cell_count[gc], 1);
    occ = slot + 1;
    if (slot < cap) {
      const int64_t idx = (int64_t)gc * cap + slot;
      cell_part[idx] = (int32_t)gi;
      for (int d = 0; d < g.dim; ++d) cpos[(int64_t)d * cstride + idx] = p[d];
    }
  }
  // largest occupancy: wave maximum, one atomic per wave
  for (int o = 32; o > 0; o >>= 1) occ = max(occ, __shfl_xor(occ, o));
  if ((threadIdx.x & 63) == 0 && occ > 0) atomicMax(&ctrl->max_cell_occ, occ);
}

// Small problems (<= LB_SMALL_N = 4096 particles and cells in the whole batch; measured: TGV2D-2.5k 0.423 -> 0.415,
// RPF2D-3.2k 0.532 -> 0.517 ms per step, but slower than the multi-launch path from ~8 k particles): cell binning in ONE single-workgroup
// launch instead of memset + count + two scan passes + fill (each ~4-5 us of launch floor on a 2.5 k-particle
// trajectory, where the whole step is 0.4 ms).  Same arithmetic and same outputs as the multi-launch path
// (the order of the particles INSIDE a cell is arbitrary in both: the rows are sorted by sender id later).
// cell coordinate of a position, int(position / cell_size) clamped to the grid: a float product locates the quotient and
// the exact fp64 (f32 mode: float-rounded) quotient is evaluated only within 1e-3 of an integer or far outside the grid
// (the comment block in front of k_nl_small has the error budget)
template <bool F32>
__device__ __forceinline__ int lb_cell_coord(double p, float inv_cs32, double cs, int n) {
  const float q = (float)p * inv_cs32;
  const float fl = floorf(q), fr = q - fl;
  int c = (int)fl;
  if (fr < 1e-3f || fr > 0.999f || !(fabsf(q) < 2000.f)) c = __double2int_rz(lb_r(p / cs, F32));
  return c < 0 ? 0 : (c >= n ? n - 1 : c);
}
#define LB_SMALL_N 4096
#define LB_SMALL_T 1024
// Round 4: the same kernel serves one mid-size trajectory that the single-launch builds refuse (DAM2D: 5740 particles,
// 12.6 k cells, 262 mask rows): counts in dynamic LDS (up to LB_CELLS1_NCELL cells), up to LB_CELLS1_PER particles per
// thread - memset + count + two scan passes + fill (25 us of launches on DAM2D) become one ~8 us launch.
#define LB_CELLS1_PER 8
#define LB_CELLS1_N (LB_SMALL_T * LB_CELLS1_PER)
#define LB_CELLS1_NCELL 32768
template <int PER>
__global__ void __launch_bounds__(LB_SMALL_T)
    k_cells_small(lb_geom g, int64_t BN, const double* __restrict__ win, lb_ctrl* __restrict__ ctrl,
                  int32_t* __restrict__ cell_of, int32_t* __restrict__ cell_start, int32_t* __restrict__ cell_part,
                  double* __restrict__ cpos, int ncell_tot) {
  extern __shared__ int s_cnt[];  // [ncell_tot]
  __shared__ int s_scan[LB_SMALL_T];
  __shared__ int s_max;
  // (one workgroup on one CU: the launch is a chain of dependent round trips - the control block is read ONCE, the position
  // loads are in flight while the counts are zeroed)
  const int poisoned = ctrl->overflow_step;
  const int step = ctrl->step;
  if (poisoned >= 0) return;
  const int tid = threadIdx.x;
  int gcs[PER], rk[PER];
  double pos[PER][3];
#pragma unroll
  for (int k = 0; k < PER; ++k) {  // all position loads of a thread in flight together
    const int64_t gi = tid + (int64_t)LB_SMALL_T * k;
    const int64_t gc = gi < BN ? gi : BN - 1;
#pragma unroll
    for (int d = 0; d < 3; ++d) pos[k][d] = d < g.dim ? lb_pos(win, g, BN, step, g.isl - 1, d, gc) : 0.0;
  }
  if (tid == 0) {
    ctrl->max_deg = 0;
    ctrl->row_overflow = 0;
    s_max = 0;
  }
  for (int c = tid; c < ncell_tot; c += LB_SMALL_T) s_cnt[c] = 0;
  __syncthreads();
  float inv_cs[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) inv_cs[d] = d < g.dim ? (float)(1.0 / g.cell_size[d]) : 0.f;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int64_t gi = tid + (int64_t)LB_SMALL_T * k;
    gcs[k] = -1;
    rk[k] = 0;
    if (gi < BN) {
      const int b = (int)(gi / g.N);
      int h = 0, mult = 1;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        if (d < g.dim) {
          // jnp.array(position / cell_size, dtype=i32), clamped (one workgroup does all N: the fp64 division was its time)
          const int c = g.f32 ? lb_cell_coord<true>(pos[k][d], inv_cs[d], g.cell_size[d], g.ncell[d])
                              : lb_cell_coord<false>(pos[k][d], inv_cs[d], g.cell_size[d], g.ncell[d]);
          h += c * mult;
          mult *= g.ncell[d];
        }
      }
      const int gc = b * g.ncells + h;
      cell_of[gi] = gc;
      gcs[k] = gc;
      rk[k] = atomicAdd(&s_cnt[gc], 1);
    }
  }
  __syncthreads();
  // exclusive scan of the counts: thread t owns the cells [t*per, (t+1)*per)
  const int per = (ncell_tot + LB_SMALL_T - 1) / LB_SMALL_T;
  const int c_lo = tid * per;
  int sum = 0, mx = 0;
  for (int j = 0; j < per; ++j) {
    const int c = c_lo + j;
    const int v = c < ncell_tot ? s_cnt[c] : 0;
    sum += v;
    mx = max(mx, v);
  }
  // inclusive scan over the 1024 thread sums: inside the wave with lane shifts, across the 16 waves through LDS
  int incl = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o);
    if ((tid & 63) >= o) incl += v;
  }
  if ((tid & 63) == 63) s_scan[tid >> 6] = incl;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o));  // one LDS atomic per wave (1024 on one address: ~3 us)
  if ((tid & 63) == 0 && mx > 0) atomicMax(&s_max, mx);
  __syncthreads();
  int wbase = 0, total = 0;
#pragma unroll
  for (int w = 0; w < LB_SMALL_T / 64; ++w) {
    const int v = s_scan[w];
    if (w < (tid >> 6)) wbase += v;
    total += v;
  }
  int run = wbase + incl - sum;
  for (int j = 0; j < per; ++j) {
    const int c = c_lo + j;
    if (c < ncell_tot) {
      const int v = s_cnt[c];
      s_cnt[c] = run;  // the counts become the cell starts
      run += v;
    }
  }
  if (tid == LB_SMALL_T - 1) cell_start[ncell_tot] = total;
  if (tid == 0) ctrl->max_cell_occ = s_max;
  __syncthreads();
  for (int c = tid; c < ncell_tot; c += LB_SMALL_T) cell_start[c] = s_cnt[c];  // coalesced (thread t owned t*per .. : stride per)
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int64_t gi = tid + (int64_t)LB_SMALL_T * k;
    if (gi < BN) {
      const int slot = s_cnt[gcs[k]] + rk[k];
      cell_part[slot] = (int32_t)gi;
#pragma unroll
      for (int d = 0; d < 3; ++d)
        if (d < g.dim) cpos[(int64_t)d * BN + slot] = pos[k][d];
    }
  }
}

// Batches (round 4): binning of trajectory b by workgroup b, ONE launch instead of memset + count + two scan passes + fill
// (five launches, ~35 us of kernels + their dependency gaps per step).  Trajectories are independent here: every trajectory
// has exactly N particles, so its cells start at slot b * N of the cell-sorted arrays and no prefix crosses workgroups.
// k_cells_small's arithmetic (same cell coordinates, arbitrary order inside a cell - the rows are sorted by sender id later).
// The per-trajectory maximum occupancy goes to occ[b]; the workgroup that finishes last (ticket counter occ[B], left at 0)
// publishes the maximum to the control block.
template <int PER>
__global__ void __launch_bounds__(LB_SMALL_T)
    k_cells_traj(lb_geom g, int64_t BN, const double* __restrict__ win, lb_ctrl* __restrict__ ctrl,
                 int32_t* __restrict__ cell_of, int32_t* __restrict__ cell_start, int32_t* __restrict__ cell_part,
                 double* __restrict__ cpos, int32_t* __restrict__ occ) {
  extern __shared__ int s_cnt[];  // [g.ncells]
  __shared__ int s_scan[LB_SMALL_T / 64];
  __shared__ int s_max, s_last;
  const int poisoned = ctrl->overflow_step;
  const int step = ctrl->step;
  if (poisoned >= 0) return;
  const int tid = threadIdx.x, b = blockIdx.x, N = g.N, nc = g.ncells;
  const int64_t p0 = (int64_t)b * N;
  int lcs[PER], rk[PER];
  double pos[PER][3];
#pragma unroll
  for (int k = 0; k < PER; ++k) {  // all position loads of a thread in flight together
    const int li = tid + LB_SMALL_T * k;
    const int64_t gi = p0 + (li < N ? li : N - 1);
#pragma unroll
    for (int d = 0; d < 3; ++d) pos[k][d] = d < g.dim ? lb_pos(win, g, BN, step, g.isl - 1, d, gi) : 0.0;
  }
  if (tid == 0) {
    s_max = 0;
    if (b == 0) {
      ctrl->max_deg = 0;
      ctrl->row_overflow = 0;
    }
  }
  for (int c = tid; c < nc; c += LB_SMALL_T) s_cnt[c] = 0;
  __syncthreads();
  float inv_cs[3];
#pragma unroll
  for (int d = 0; d < 3; ++d) inv_cs[d] = d < g.dim ? (float)(1.0 / g.cell_size[d]) : 0.f;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int li = tid + LB_SMALL_T * k;
    lcs[k] = -1;
    rk[k] = 0;
    if (li < N) {
      int h = 0, mult = 1;
#pragma unroll
      for (int d = 0; d < 3; ++d) {
        if (d < g.dim) {
          const int c = g.f32 ? lb_cell_coord<true>(pos[k][d], inv_cs[d], g.cell_size[d], g.ncell[d])
                              : lb_cell_coord<false>(pos[k][d], inv_cs[d], g.cell_size[d], g.ncell[d]);
          h += c * mult;
          mult *= g.ncell[d];
        }
      }
      cell_of[p0 + li] = b * nc + h;
      lcs[k] = h;
      rk[k] = atomicAdd(&s_cnt[h], 1);
    }
  }
  __syncthreads();
  // exclusive scan of the counts: thread t owns the cells [t*per, (t+1)*per)
  const int per = (nc + LB_SMALL_T - 1) / LB_SMALL_T;
  const int c_lo = tid * per;
  int sum = 0, mx = 0;
  for (int j = 0; j < per; ++j) {
    const int c = c_lo + j;
    const int v = c < nc ? s_cnt[c] : 0;
    sum += v;
    mx = max(mx, v);
  }
  int incl = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int v = __shfl_up(incl, o);
    if ((tid & 63) >= o) incl += v;
  }
  if ((tid & 63) == 63) s_scan[tid >> 6] = incl;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o));
  if ((tid & 63) == 0 && mx > 0) atomicMax(&s_max, mx);
  __syncthreads();
  int wbase = 0;
#pragma unroll
  for (int w = 0; w < LB_SMALL_T / 64; ++w)
    if (w < (tid >> 6)) wbase += s_scan[w];
  int run = (int)p0 + wbase + incl - sum;  // global slot of the trajectory's first particle + local start
  for (int j = 0; j < per; ++j) {
    const int c = c_lo + j;
    if (c < nc) {
      const int v = s_cnt[c];
      s_cnt[c] = run;  // the counts become the cell starts
      run += v;
    }
  }
  __syncthreads();
  for (int c = tid; c < nc; c += LB_SMALL_T) cell_start[(int64_t)b * nc + c] = s_cnt[c];
  if (b == (int)gridDim.x - 1 && tid == 0) cell_start[(int64_t)gridDim.x * nc] = (int32_t)BN;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int li = tid + LB_SMALL_T * k;
    if (li < N) {
      const int slot = s_cnt[lcs[k]] + rk[k];
      cell_part[slot] = (int32_t)(p0 + li);
#pragma unroll
      for (int d = 0; d < 3; ++d)
        if (d < g.dim) cpos[(int64_t)d * BN + slot] = pos[k][d];
    }
  }
  // maximum occupancy over the batch: last workgroup to arrive combines (integers: the order does not matter)
  if (tid == 0) {
    __hip_atomic_store(&occ[b], s_max, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    const int ticket = atomicAdd(&occ[gridDim.x], 1);
    s_last = ticket == (int)gridDim.x - 1;
  }
  __syncthreads();
  if (s_last && tid == 0) {
    __threadfence();
    int m = 0;
    for (int i = 0; i < (int)gridDim.x; ++i) m = max(m, __hip_atomic_load(&occ[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    ctrl->max_cell_occ = m;
    __hip_atomic_store(&occ[gridDim.x], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
}

// -------------------------------------------------------------------------- stencil search
struct lb_nl_args {
  const int32_t* cell_of;   // [BN] global cell id of each particle
  const int32_t* cell_start;
  const int32_t* cell_part;
  const double* cpos;       // [dim][cstride] positions in cell-sorted order
  // cell ranges: CSR (cell_cap 0: [cell_start[c], cell_start[c + 1]), cstride = BN) or, on the frozen update path (round 6),
  // FIXED-STRIDE slots (cell c owns slots [c * cell_cap, c * cell_cap + min(cell_cnt[c], cell_cap)), cstride = cells * cap:
  // jax-md's own layout; binning is one atomic per particle, no scan)
  const int32_t* cell_cnt;
  int32_t cell_cap;
  int64_t cstride;
  int32_t* deg;             // [BN]
  const int32_t* row_ptr;   // NL_FILL
  int32_t* senders;         // NL_FILL: CSR arrays; NL_ROWS: per-node slots [BN][maxd]
  int32_t* receivers;       // NL_FILL
  float* efeat;             // NL_FILL: [E][8]; NL_ROWS: [BN][maxd][4]
  double* efeat64;          // optional fp64 copy, same indexing with 4 doubles per edge
  int64_t e_alloc;
  int32_t maxd;
  int32_t row_cap;          // k_nlw: LDS row-buffer entries per wave (>= LB_MAX_ROW)
  lb_feat_job feat;         // NL_ROWS in a rollout step: every search wave also writes the node-feature row of its
  const double* win;        //   receiver (lb_features.h; xnode == null: no job)
  int32_t nb_search;        // k_nl: workgroups of the search proper; the ones behind them write feature rows (0: none)
};
#define NL_FEAT_ROWS 8      // feature rows per wave of a k_nl feature workgroup
__device__ __forceinline__ int lb_cell_begin(const lb_nl_args& a, int gc) { return a.cell_cap ? gc * a.cell_cap : a.cell_start[gc]; }
__device__ __forceinline__ int lb_cell_size(const lb_nl_args& a, int gc, int begin) {
  return a.cell_cap ? min(a.cell_cnt[gc], a.cell_cap) : a.cell_start[gc + 1] - begin;
}

template <int MODE, int NL_THREADS, int MAXC, bool F32 = false>
__global__ void __launch_bounds__(NL_THREADS)
    k_nl(lb_geom g, int64_t BN, lb_ctrl* __restrict__ ctrl, lb_nl_args a) {
  constexpr int NL_WAVES = NL_THREADS / 64;
  __shared__ int s_id[MAXC];
  __shared__ double s_p[3][MAXC];
  __shared__ int s_row[NL_WAVES][LB_MAX_ROW];
  __shared__ int s_cstart[28], s_coff[29];

  if (ctrl->overflow_step >= 0) return;
  if (MODE == NL_ROWS && a.nb_search > 0 && (int)blockIdx.x >= a.nb_search) {
    // rollout step: the workgroups behind the search's write the node-feature rows (NL_FEAT_ROWS consecutive particles
    // per wave, all their loads in flight together; they run beside the per-cell search, which is a latency chain)
    const int64_t first = ((int64_t)(blockIdx.x - a.nb_search) * NL_WAVES + (threadIdx.x >> 6)) * NL_FEAT_ROWS;
    const int cnt = (int)min((int64_t)NL_FEAT_ROWS, BN - first);
    if (cnt > 0)
      lb_node_features_wave_multi(g, BN, a.win, ctrl->step, a.feat, [&](int p) -> int64_t { return first + p; }, cnt);
    return;
  }
  const int tid = threadIdx.x;
  const int gc = blockIdx.x;
  const int b = gc / g.ncells, h = gc % g.ncells;
  const int own_start = lb_cell_begin(a, gc);
  const int own_cnt = lb_cell_size(a, gc, own_start);
  if (own_cnt == 0) return;

  // stencil cells: start + count, exclusive prefix of the counts with a wave scan
  if (tid < 64) {
    int cnt = 0;
    if (tid < g.nstencil) {
      int nh = h;
      if (g.use_cell_list) {
        int c[3] = {h % g.ncell[0], (h / g.ncell[0]) % g.ncell[1], h / (g.ncell[0] * g.ncell[1])};
        int o[3] = {tid % 3 - 1, (tid / 3) % 3 - 1, tid / 9 - 1};
        nh = 0;
        int mult = 1;
        for (int d = 0; d < g.dim; ++d) {
          int cc = c[d] + o[d];  // jax-md rolls the cell buffer: the stencil always wraps
          cc = cc < 0 ? cc + g.ncell[d] : (cc >= g.ncell[d] ? cc - g.ncell[d] : cc);
          nh += cc * mult;
          mult *= g.ncell[d];
        }
      }
      const int ngc = b * g.ncells + nh;
      const int st = lb_cell_begin(a, ngc);
      s_cstart[tid] = st;
      cnt = lb_cell_size(a, ngc, st);
    }
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
      const int v = __shfl_up(incl, off);
      if (tid >= off) incl += v;
    }
    if (tid < g.nstencil) s_coff[tid] = incl - cnt;
    if (tid == g.nstencil - 1) s_coff[g.nstencil] = incl;
  }
  __syncthreads();
  int M = s_coff[g.nstencil];
  if (M > MAXC) {
    // too many stencil candidates for the staged kernel.  Update path: ask for a re-allocation (which detects the
    // density in its counting pass and switches the engine to the dense fall-back); allocation: flag it
    if (tid == 0) {
      if (MODE == NL_ROWS)
        atomicExch(&ctrl->row_overflow, 1);
      else
        atomicExch(&ctrl->density_error, 1);
    }
    M = MAXC;
  }
  // stage the stencil's particles (ids + fp64 positions, contiguous per cell in the sorted arrays)
  for (int j = tid; j < M; j += NL_THREADS) {
    int lo = 0, hi = g.nstencil;  // largest k with s_coff[k] <= j
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (s_coff[mid] <= j) lo = mid; else hi = mid;
    }
    const int src = s_cstart[lo] + (j - s_coff[lo]);
    s_id[j] = a.cell_part[src];
    for (int d = 0; d < g.dim; ++d) s_p[d][j] = a.cpos[(int64_t)d * a.cstride + src];
  }
  __syncthreads();

  const int wave = tid >> 6, lane = tid & 63;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  // the cell's own particles are the centre entry of the staged stencil
  const int centre = g.use_cell_list ? (g.dim == 2 ? 4 : 13) : 0;
  const int own_off = s_coff[centre];
  for (int k = wave; k < own_cnt; k += NL_WAVES) {
    if (own_off + k >= M) break;  // truncated stencil (density error already flagged)
    const int gr = s_id[own_off + k];
    double pr[3] = {0, 0, 0};
    for (int d = 0; d < g.dim; ++d) pr[d] = s_p[d][own_off + k];
    int count = 0;
    for (int c0 = 0; c0 < M; c0 += 64) {
      const int j = c0 + lane;
      bool ok = false;
      if (j < M) {
        // metric_sq(position[sender], position[receiver]): sender = staged candidate,
        // receiver = row owner; sum of squares in x,y,z order, no FMA contraction.
        double dd = lb_disp1(s_p[0][j], pr[0], g.box[0], g.half_box[0], g.periodic, F32);
        double d2 = lb_r(dd * dd, F32);
        for (int d = 1; d < g.dim; ++d) {
          dd = lb_disp1(s_p[d][j], pr[d], g.box[d], g.half_box[d], g.periodic, F32);
          d2 = lb_r(d2 + lb_r(dd * dd, F32), F32);
        }
        ok = d2 < g.rc2;  // strict <
      }
      const unsigned long long mask = __ballot(ok);
      if (MODE != NL_COUNT && ok) {
        const int pos = count + __popcll(mask & lt_mask);
        if (pos < LB_MAX_ROW) s_row[wave][pos] = j;
      }
      count += __popcll(mask);
    }
    if (MODE != NL_FILL && lane == 0) {
      a.deg[gr] = count;
      // (ctrl->max_deg is reduced by the degree scan that follows: a read of that ONE address by every wave
      // serialises at its L2 channel - it was 150 of the 165 us of this kernel on 64 k receivers)
    }
    if (MODE == NL_COUNT) continue;
    if (count > LB_MAX_ROW) {  // (update path: re-allocate - the allocation sizes the dense fall-back's row buffer)
      if (lane == 0) {
        if (MODE == NL_ROWS)
          atomicExch(&ctrl->row_overflow, 1);
        else
          atomicExch(&ctrl->density_error, 2);
      }
      count = LB_MAX_ROW;
    }
    if (MODE == NL_ROWS && count > a.maxd) {
      if (lane == 0) atomicExch(&ctrl->row_overflow, 1);  // per-node slots too small: re-allocate
      count = a.maxd;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    const int64_t base = (MODE == NL_ROWS) ? (int64_t)gr * a.maxd : (int64_t)a.row_ptr[gr];
    for (int t0 = 0; t0 < count; t0 += 64) {
      const int t = t0 + lane;
      const bool act = t < count;
      const int j = act ? s_row[wave][t] : 0;
      const int my = act ? s_id[j] : 0x7fffffff;
      // rank of this sender id inside the row: rows of <= 64 neighbors (the normal case) compare
      // against lane broadcasts (v_readlane, no LDS round trips), longer rows walk the LDS row
      int rank = 0;
      if (count <= 64) {
        for (int u = 0; u < count; ++u) rank += (__builtin_amdgcn_readlane(my, u) < my) ? 1 : 0;
      } else {
        for (int u = 0; u < count; ++u) rank += (s_id[s_row[wave][u]] < my) ? 1 : 0;
      }
      if (!act) continue;
      const int64_t slot = base + rank;
      if (MODE == NL_ROWS || slot < a.e_alloc) {
        a.senders[slot] = my;
        // features.py:115-124: disp(pos[receiver], pos[sender]) / r_c and its norm
        double rd[3] = {0, 0, 0};
        double s2 = 0.0;
        for (int d = 0; d < g.dim; ++d) {
          rd[d] = lb_r(lb_disp1(pr[d], s_p[d][j], g.box[d], g.half_box[d], g.periodic, F32) / g.rc, F32);
          s2 = (d == 0) ? lb_r(rd[d] * rd[d], F32) : lb_r(s2 + lb_r(rd[d] * rd[d], F32), F32);
        }
        const double dist = s2 > 0.0 ? lb_r(sqrt(s2), F32) : 0.0;
        const f32x4 lo = (g.dim == 2) ? f32x4{(float)rd[0], (float)rd[1], (float)dist, 0.f}
                                      : f32x4{(float)rd[0], (float)rd[1], (float)rd[2], (float)dist};
        if (MODE == NL_ROWS) {
          reinterpret_cast<f32x4*>(a.efeat)[slot] = lo;
        } else {
          a.receivers[slot] = gr;
          f32x4* ef = reinterpret_cast<f32x4*>(a.efeat + slot * 8);
          ef[0] = lo;
          ef[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (a.efeat64) {
          double* e64 = a.efeat64 + slot * 4;
          e64[0] = rd[0];
          e64[1] = rd[1];
          e64[2] = rd[2];
          e64[3] = dist;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  }
}

// One WAVE per receiver, nothing staged: the receiver (slot r of the cell-sorted order) walks the
// particles of its 3^dim stencil cells - 3^dim short contiguous runs of the cell-sorted arrays, L2
// resident - 64 candidates per sweep, straight from global memory.  The only LDS is the wave's own
// row buffer and its 3^dim-entry stencil table (1.3 KiB per wave), so a CU keeps its full
// complement of waves in flight; the search is latency bound (three dependent memory round trips
// per receiver), and with one workgroup per CELL the staged stencil (28 B x cell_capacity x 3^dim of
// LDS) capped the occupancy at a handful of waves per CU.  Same predicate, same operand order.
#define NLW_WAVES 4
template <int MODE, bool F32 = false>
__global__ void __launch_bounds__(64 * NLW_WAVES)
    k_nlw(lb_geom g, int64_t BN, lb_ctrl* __restrict__ ctrl, lb_nl_args a) {
  // row buffers in DYNAMIC LDS: per wave row_cap candidate slots + row_cap sender ids (a.row_cap >= LB_MAX_ROW: the
  // dense fall-back sizes it from the largest degree seen - the reference re-allocates for any occupancy)
  extern __shared__ int s_dyn[];
  const int row_cap = a.row_cap;
  __shared__ int s_cstart[NLW_WAVES][28], s_coff[NLW_WAVES][29];
  if (ctrl->overflow_step >= 0) return;
  if (MODE == NL_ROWS && a.nb_search > 0 && (int)blockIdx.x >= a.nb_search) {
    // rollout step: the workgroups behind the search's write the node-feature rows (as in k_nl; inside the receiver's
    // own wave the row cost the batch search +19 us - one more round trip in a chain of five)
    const int64_t first = ((int64_t)(blockIdx.x - a.nb_search) * NLW_WAVES + (threadIdx.x >> 6)) * NL_FEAT_ROWS;
    const int cnt = (int)min((int64_t)NL_FEAT_ROWS, BN - first);
    if (cnt > 0)
      lb_node_features_wave_multi(g, BN, a.win, ctrl->step, a.feat, [&](int p) -> int64_t { return first + p; }, cnt);
    return;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int* const s_row = s_dyn + (size_t)wave * 2 * row_cap;
  int* const s_id = s_row + row_cap;
  const int64_t r = (int64_t)blockIdx.x * NLW_WAVES + wave;  // receiver slot in cell-sorted order
  if (r >= BN) return;
  const int gr = a.cell_part[r];
  const int gc = a.cell_of[gr];
  const int b = gc / g.ncells, h = gc % g.ncells;
  {
    int cnt = 0;
    if (lane < g.nstencil) {
      int nh = h;
      if (g.use_cell_list) {
        int c[3] = {h % g.ncell[0], (h / g.ncell[0]) % g.ncell[1], h / (g.ncell[0] * g.ncell[1])};
        int o[3] = {lane % 3 - 1, (lane / 3) % 3 - 1, lane / 9 - 1};
        nh = 0;
        int mult = 1;
        for (int d = 0; d < g.dim; ++d) {
          int cc = c[d] + o[d];  // jax-md rolls the cell buffer: the stencil always wraps
          cc = cc < 0 ? cc + g.ncell[d] : (cc >= g.ncell[d] ? cc - g.ncell[d] : cc);
          nh += cc * mult;
          mult *= g.ncell[d];
        }
      }
      const int ngc = b * g.ncells + nh;
      const int st = lb_cell_begin(a, ngc);
      s_cstart[wave][lane] = st;
      cnt = lb_cell_size(a, ngc, st);
    }
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
      const int v = __shfl_up(incl, off);
      if (lane >= off) incl += v;
    }
    if (lane < g.nstencil) s_coff[wave][lane] = incl - cnt;
    if (lane == g.nstencil - 1) s_coff[wave][g.nstencil] = incl;
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  const int M = s_coff[wave][g.nstencil];
  double pr[3] = {0, 0, 0};
  for (int d = 0; d < g.dim; ++d) pr[d] = a.cpos[(int64_t)d * a.cstride + r];
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  auto slot_of = [&](int j) -> int {  // candidate j -> slot in the cell-sorted arrays
    int lo = 0, hi = g.nstencil;      // largest k with s_coff[k] <= j
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (s_coff[wave][mid] <= j) lo = mid; else hi = mid;
    }
    return s_cstart[wave][lo] + (j - s_coff[wave][lo]);
  };
  int count = 0;
  for (int c0 = 0; c0 < M; c0 += 64) {
    const int j = c0 + lane;
    bool ok = false;
    int src = 0;
    if (j < M) {
      src = slot_of(j);
      // metric_sq(position[sender], position[receiver]): sum of squares in x,y,z order, no FMA
      double dd = lb_disp1(a.cpos[src], pr[0], g.box[0], g.half_box[0], g.periodic, F32);
      double d2 = lb_r(dd * dd, F32);
      for (int d = 1; d < g.dim; ++d) {
        dd = lb_disp1(a.cpos[(int64_t)d * a.cstride + src], pr[d], g.box[d], g.half_box[d], g.periodic, F32);
        d2 = lb_r(d2 + lb_r(dd * dd, F32), F32);
      }
      ok = d2 < g.rc2;  // strict <
    }
    const unsigned long long mask = __ballot(ok);
    if (MODE != NL_COUNT && ok) {
      const int pos = count + __popcll(mask & lt_mask);
      if (pos < row_cap) s_row[pos] = src;
    }
    count += __popcll(mask);
  }
  if (MODE != NL_FILL && lane == 0) {
    a.deg[gr] = count;
    // (ctrl->max_deg is reduced by the degree scan that follows, see k_nl)
  }
  if (MODE == NL_COUNT) return;
  if (count > row_cap) {
    if (lane == 0) {
      if (MODE == NL_ROWS && row_cap < LB_MAX_ROW_DENSE)
        atomicExch(&ctrl->row_overflow, 1);  // re-allocate with a longer row buffer
      else
        atomicExch(&ctrl->density_error, 2);
    }
    count = row_cap;
  }
  if (MODE == NL_ROWS && count > a.maxd) {
    if (lane == 0) atomicExch(&ctrl->row_overflow, 1);  // per-node slots too small: re-allocate
    count = a.maxd;
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  const int64_t base = (MODE == NL_ROWS) ? (int64_t)gr * a.maxd : (int64_t)a.row_ptr[gr];
  if (count > 64) {  // long rows: the sender ids go to LDS once, the rank loop reads them as broadcasts
    for (int t = lane; t < count; t += 64) s_id[t] = a.cell_part[s_row[t]];
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  }
  for (int t0 = 0; t0 < count; t0 += 64) {
    const int t = t0 + lane;
    const bool act = t < count;
    const int src = act ? s_row[t] : 0;
    const int my = act ? a.cell_part[src] : 0x7fffffff;
    // rank of this sender id inside the row (ids are unique): lane broadcasts for rows <= 64
    int rank = 0;
    if (count <= 64) {
      for (int u = 0; u < count; ++u) rank += (__builtin_amdgcn_readlane(my, u) < my) ? 1 : 0;
    } else {
      for (int u = 0; u < count; ++u) rank += (s_id[u] < my) ? 1 : 0;
    }
    if (!act) continue;
    const int64_t slot = base + rank;
    if (MODE == NL_ROWS || slot < a.e_alloc) {
      a.senders[slot] = my;
      // features.py:115-124: disp(pos[receiver], pos[sender]) / r_c and its norm
      double rd[3] = {0, 0, 0};
      double s2 = 0.0;
      for (int d = 0; d < g.dim; ++d) {
        rd[d] = lb_r(lb_disp1(pr[d], a.cpos[(int64_t)d * a.cstride + src], g.box[d], g.half_box[d], g.periodic, F32) / g.rc, F32);
        s2 = (d == 0) ? lb_r(rd[d] * rd[d], F32) : lb_r(s2 + lb_r(rd[d] * rd[d], F32), F32);
      }
      const double dist = s2 > 0.0 ? lb_r(sqrt(s2), F32) : 0.0;
      const f32x4 lo = (g.dim == 2) ? f32x4{(float)rd[0], (float)rd[1], (float)dist, 0.f}
                                    : f32x4{(float)rd[0], (float)rd[1], (float)rd[2], (float)dist};
      if (MODE == NL_ROWS) {
        reinterpret_cast<f32x4*>(a.efeat)[slot] = lo;
      } else {
        a.receivers[slot] = gr;
        f32x4* ef = reinterpret_cast<f32x4*>(a.efeat + slot * 8);
        ef[0] = lo;
        ef[1] = f32x4{0.f, 0.f, 0.f, 0.f};
      }
      if (a.efeat64) {
        double* e64 = a.efeat64 + slot * 4;
        e64[0] = rd[0];
        e64[1] = rd[1];
        e64[2] = rd[2];
        e64[3] = dist;
      }
    }
  }
}

// One WAVE per CELL (round 6).  k_nlw pays the stencil walk, the candidate addressing, a rank pass
// and one fp64 emission pass (three divisions and a square root, a dozen lanes busy) for EVERY receiver: ~740 VALU + ~390
// SALU instructions per receiver, and its launch keeps the VALU pipes ~95 % busy (profiles/r06_nl_sq.txt).  Here the wave
// of a cell builds the stencil table once, keeps the <= 128 stencil candidates (cell-sorted slot, particle id, fp64
// position: 2 per lane) in registers for all the cell's receivers, appends every receiver's hits to ONE LDS list (ballot +
// popcount prefix as before; the rank of a hit inside its row - ascending particle id - is taken right away with lane
// broadcasts) and emits the list 64 edges at a time, so the fp64 feature arithmetic runs on full waves whatever rows the
// edges belong to.  Predicate, features and operand order are k_nlw's: the edge list is bit-identical.  Stencils with more
// than 128 candidates sweep in 128-candidate chunks per receiver (registers reloaded: dense cells, rare); the list holds
// NLC_LIST hits and is flushed whenever the next row (<= LB_MAX_ROW hits) might not fit.
#define NLC_WAVES 4
#define NLC_LIST 448
#define NLC_BATCH 128
#define NLC_S 4   // candidate slots per lane held in registers (stencils of up to 256 particles in one chunk)
template <int MODE, bool F32, int DIM>
__global__ void __launch_bounds__(64 * NLC_WAVES, 5)
    k_nlc(lb_geom g, int64_t BN, lb_ctrl* __restrict__ ctrl, lb_nl_args a) {
  __shared__ int s_cstart_[NLC_WAVES][28], s_coff_[NLC_WAVES][29];
  __shared__ int s_src_[NLC_WAVES][NLC_LIST], s_id_[NLC_WAVES][NLC_LIST];
  __shared__ unsigned short s_kr_[NLC_WAVES][NLC_LIST];  // row of the batch << 8 | rank inside the row
  if (ctrl->overflow_step >= 0) return;
  if (MODE == NL_ROWS && a.nb_search > 0 && (int)blockIdx.x >= a.nb_search) {
    // rollout step: the workgroups behind the search's write the node-feature rows (as in k_nlw)
    const int64_t first = ((int64_t)(blockIdx.x - a.nb_search) * NLC_WAVES + (threadIdx.x >> 6)) * NL_FEAT_ROWS;
    const int cnt = (int)min((int64_t)NL_FEAT_ROWS, BN - first);
    if (cnt > 0)
      lb_node_features_wave_multi(g, BN, a.win, ctrl->step, a.feat, [&](int p) -> int64_t { return first + p; }, cnt);
    return;
  }
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int gc = __builtin_amdgcn_readfirstlane((int)blockIdx.x * NLC_WAVES + wave);
  if (gc >= g.B * g.ncells) return;
  const int own_start = lb_cell_begin(a, gc);
  const int own_cnt = lb_cell_size(a, gc, own_start);
  if (own_cnt <= 0) return;
  int* const s_cstart = s_cstart_[wave];
  int* const s_coff = s_coff_[wave];
  int* const s_src = s_src_[wave];
  int* const s_id = s_id_[wave];
  unsigned short* const s_kr = s_kr_[wave];
  const int b = gc / g.ncells, h = gc % g.ncells;
  {
    int cnt = 0;
    if (lane < g.nstencil) {
      int nh = h;
      if (g.use_cell_list) {
        int c[3] = {h % g.ncell[0], (h / g.ncell[0]) % g.ncell[1], h / (g.ncell[0] * g.ncell[1])};
        int o[3] = {lane % 3 - 1, (lane / 3) % 3 - 1, lane / 9 - 1};
        nh = 0;
        int mult = 1;
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
          int cc = c[d] + o[d];  // jax-md rolls the cell buffer: the stencil always wraps
          cc = cc < 0 ? cc + g.ncell[d] : (cc >= g.ncell[d] ? cc - g.ncell[d] : cc);
          nh += cc * mult;
          mult *= g.ncell[d];
        }
      }
      const int ngc = b * g.ncells + nh;
      const int st = lb_cell_begin(a, ngc);
      s_cstart[lane] = st;
      cnt = lb_cell_size(a, ngc, st);
    }
    int incl = cnt;
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
      const int v = __shfl_up(incl, off);
      if (lane >= off) incl += v;
    }
    if (lane < g.nstencil) s_coff[lane] = incl - cnt;
    if (lane == g.nstencil - 1) s_coff[g.nstencil] = incl;
  }
  __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  const int M = __builtin_amdgcn_readfirstlane(s_coff[g.nstencil]);
  const bool single = M <= 64 * NLC_S;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  // candidates of one chunk of NLC_S * 64: cell-sorted slot + particle id (loads from clamped addresses, never selects);
  // slots whose 64 candidates lie past M are skipped (wave-uniform)
  int cj[NLC_S], cid[NLC_S];
  auto load_chunk = [&](int c0) {
#pragma unroll
    for (int s = 0; s < NLC_S; ++s) {
      if (c0 + 64 * s >= M) continue;
      const int j = min(c0 + 64 * s + lane, M - 1);
      int lo = 0, hi = g.nstencil;  // largest k with s_coff[k] <= j
      while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (s_coff[mid] <= j) lo = mid; else hi = mid;
      }
      cj[s] = s_cstart[lo] + (j - s_coff[lo]);
      cid[s] = a.cell_part[cj[s]];
    }
  };
  const int cap = (MODE == NL_ROWS) ? min(a.maxd, LB_MAX_ROW) : LB_MAX_ROW;
  int list_len = 0, kfirst = 0;
  // emission of the list: 64 edges per pass, whatever rows they belong to
  auto flush = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    for (int t0 = 0; t0 < list_len; t0 += 64) {
      const int t = t0 + lane;
      const bool act = t < list_len;
      const int tc = act ? t : 0;
      const int src = s_src[tc], my = s_id[tc], kr = s_kr[tc];
      const int rslot = own_start + kfirst + (kr >> 8), rank = kr & 0xff;
      const int gr = a.cell_part[rslot];
      double pr[DIM], sp[DIM];
#pragma unroll
      for (int d = 0; d < DIM; ++d) {
        pr[d] = a.cpos[(int64_t)d * a.cstride + rslot];
        sp[d] = a.cpos[(int64_t)d * a.cstride + src];
      }
      if (!act) continue;
      const int64_t base = (MODE == NL_ROWS) ? (int64_t)gr * a.maxd : (int64_t)a.row_ptr[gr];
      const int64_t slot = base + rank;
      if (MODE == NL_ROWS || slot < a.e_alloc) {
        a.senders[slot] = my;
        // features.py:115-124: disp(pos[receiver], pos[sender]) / r_c and its norm
        double rd[3] = {0, 0, 0};
        double s2 = 0.0;
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
          rd[d] = lb_r(lb_disp1(pr[d], sp[d], g.box[d], g.half_box[d], g.periodic, F32) / g.rc, F32);
          s2 = (d == 0) ? lb_r(rd[d] * rd[d], F32) : lb_r(s2 + lb_r(rd[d] * rd[d], F32), F32);
        }
        const double dist = s2 > 0.0 ? lb_r(sqrt(s2), F32) : 0.0;
        const f32x4 lo4 = (DIM == 2) ? f32x4{(float)rd[0], (float)rd[1], (float)dist, 0.f}
                                     : f32x4{(float)rd[0], (float)rd[1], (float)rd[2], (float)dist};
        if (MODE == NL_ROWS) {
          reinterpret_cast<f32x4*>(a.efeat)[slot] = lo4;
        } else {
          a.receivers[slot] = gr;
          f32x4* ef = reinterpret_cast<f32x4*>(a.efeat + slot * 8);
          ef[0] = lo4;
          ef[1] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        if (a.efeat64) {
          double* e64 = a.efeat64 + slot * 4;
          e64[0] = rd[0];
          e64[1] = rd[1];
          e64[2] = rd[2];
          e64[3] = dist;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
  };
#pragma unroll
  for (int s = 0; s < NLC_S; ++s) cj[s] = cid[s] = 0;
  if (single) load_chunk(0);
  // float pre-filter (single chunk, fp64 geometry).  q = (float) minimal image of (candidate - origin), origin = the cell's
  // first particle, taken in fp64 ONCE per cell; a pair's squared distance in float, |q_s - q_r|^2, differs from the
  // reference's fp64 value by < 4e-6 r_c^2 as long as every |q_d| <= 8 r_c (rounding of q: 2 * 2^-24 * 8 r_c per component,
  // the float arithmetic: a few 2^-24), and it IS the minimal image's as long as 2 max|q_d| <= L_d - 1.001 r_c (then a
  // component beyond L_d / 2 still leaves the image outside the cutoff).  Pairs whose float value is within 1e-4 r_c^2 of
  // the threshold - ~1 cell in 50 has one - send the receiver to the exact fp64 predicate below (positions re-read from
  // L1 / L2), which also serves float32 geometry and chunked stencils; everything else is decided in float: same edge
  // list, bit for bit.  The candidates' fp64 positions never stay in registers.
  float q[NLC_S][DIM];
#pragma unroll
  for (int s = 0; s < NLC_S; ++s)
    _Pragma("unroll") for (int d = 0; d < DIM; ++d) q[s][d] = 0.f;
  bool filt = false;
  const float lo2 = (float)(g.rc2 * (1.0 - 1e-4)), hi2 = (float)(g.rc2 * (1.0 + 1e-4));
  if (!F32 && single) {
    bool bad = false;
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
      const double org = a.cpos[(int64_t)d * a.cstride + own_start];
      float lim = 8.f * (float)g.rc;
      if (g.periodic) lim = fminf(lim, 0.5f * (float)(g.box[d] - 1.001 * g.rc));
      _Pragma("unroll") for (int s = 0; s < NLC_S; ++s) {
        if (64 * s >= M) continue;
        q[s][d] = (float)lb_disp1(a.cpos[(int64_t)d * a.cstride + cj[s]], org, g.box[d], g.half_box[d], g.periodic, 0);
        bad = bad || !(fabsf(q[s][d]) <= lim);
      }
    }
    filt = __ballot(bad) == 0ull;
  }
  const int own_off = __builtin_amdgcn_readfirstlane(s_coff[g.use_cell_list ? (DIM == 2 ? 4 : 13) : 0]);
  for (int k = 0; k <= own_cnt; ++k) {  // (iteration own_cnt only flushes: ONE inlined copy of the emission)
    if (MODE != NL_COUNT &&
        (k == own_cnt ? list_len > 0 : (list_len > NLC_LIST - LB_MAX_ROW || k - kfirst >= NLC_BATCH))) {
      flush();
      list_len = 0;
      kfirst = k;
    }
    if (k == own_cnt) break;
    const int gr = a.cell_part[own_start + k];
    const int kb = k - kfirst;
    int count = 0;
    for (int c0 = 0; c0 < M; c0 += 64 * NLC_S) {
      if (!single) load_chunk(c0);
      bool ok[NLC_S];
#pragma unroll
      for (int s = 0; s < NLC_S; ++s) ok[s] = false;
      bool exact = !filt;
      if (filt) {
        const int idx = own_off + k, sel = idx >> 6, l = idx & 63;  // the receiver is candidate own_off + k
        float pq[DIM];
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
          float v = q[0][d];
          _Pragma("unroll") for (int s = 1; s < NLC_S; ++s) v = sel == s ? q[s][d] : v;
          pq[d] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
        }
        bool unc = false;
#pragma unroll
        for (int s = 0; s < NLC_S; ++s) {
          if (64 * s >= M) continue;
          float dd = q[s][0] - pq[0];
          float d2 = dd * dd;
          _Pragma("unroll") for (int d = 1; d < DIM; ++d) {
            dd = q[s][d] - pq[d];
            d2 = __builtin_fmaf(dd, dd, d2);
          }
          const bool valid = 64 * s + lane < M;
          ok[s] = valid && d2 < lo2;
          unc = unc || (valid && !(d2 < lo2) && !(d2 > hi2));
        }
        exact = __ballot(unc) != 0ull;
      }
      if (exact) {
        double pr[DIM];
#pragma unroll
        for (int d = 0; d < DIM; ++d) pr[d] = a.cpos[(int64_t)d * a.cstride + own_start + k];
#pragma unroll
        for (int s = 0; s < NLC_S; ++s) {
          if (c0 + 64 * s >= M) continue;
          // metric_sq(position[sender], position[receiver]): sum of squares in x,y,z order, no FMA
          double dd = lb_disp1(a.cpos[cj[s]], pr[0], g.box[0], g.half_box[0], g.periodic, F32);
          double d2 = lb_r(dd * dd, F32);
          _Pragma("unroll") for (int d = 1; d < DIM; ++d) {
            dd = lb_disp1(a.cpos[(int64_t)d * a.cstride + cj[s]], pr[d], g.box[d], g.half_box[d], g.periodic, F32);
            d2 = lb_r(d2 + lb_r(dd * dd, F32), F32);
          }
          ok[s] = (c0 + 64 * s + lane < M) && (d2 < g.rc2);  // strict <
        }
      }
#pragma unroll
      for (int s = 0; s < NLC_S; ++s) {
        if (c0 + 64 * s >= M) continue;
        const unsigned long long m = __ballot(ok[s]);
        if (MODE != NL_COUNT) {
          const int p = count + __popcll(m & lt_mask);
          if (ok[s] && p < cap) {
            s_src[list_len + p] = cj[s];
            s_id[list_len + p] = cid[s];
          }
        }
        count += __popcll(m);
      }
    }
    if (MODE != NL_FILL && lane == 0) a.deg[gr] = count;  // (ctrl->max_deg is reduced by the degree scan, see k_nl)
    if (MODE == NL_COUNT) continue;
    if (count > LB_MAX_ROW) {
      if (lane == 0) {
        if (MODE == NL_ROWS)
          atomicExch(&ctrl->row_overflow, 1);  // re-allocate (the allocation switches to the dense fall-back)
        else
          atomicExch(&ctrl->density_error, 2);
      }
      count = LB_MAX_ROW;
    }
    if (MODE == NL_ROWS && count > a.maxd) {
      if (lane == 0) atomicExch(&ctrl->row_overflow, 1);  // per-node slots too small: re-allocate
      count = a.maxd;
    }
    // rank of every hit inside its row (ids are unique): lane broadcasts for rows <= 64, the LDS row otherwise
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    if (count <= 64) {
      const int my = s_id[list_len + min(lane, max(count - 1, 0))];
      int rank = 0;
      for (int u = 0; u < count; ++u) rank += (__builtin_amdgcn_readlane(my, u) < my) ? 1 : 0;
      if (lane < count) s_kr[list_len + lane] = (unsigned short)((kb << 8) | rank);
    } else {
      for (int t = lane; t < count; t += 64) {
        const int my = s_id[list_len + t];
        int rank = 0;
        for (int u = 0; u < count; ++u) rank += (s_id[list_len + u] < my) ? 1 : 0;
        s_kr[list_len + t] = (unsigned short)((kb << 8) | rank);
      }
    }
    list_len += count;
  }
}

// NL_ROWS -> CSR: one 16-lane group per node copies its sorted row into place.
__global__ void __launch_bounds__(256)
    k_nl_compact(int64_t BN, const lb_ctrl* __restrict__ ctrl, const int32_t* __restrict__ deg,
                 const int32_t* __restrict__ row_ptr, int32_t maxd, const int32_t* __restrict__ tsend,
                 const float* __restrict__ tfeat, const double* __restrict__ tfeat64,
                 int32_t* __restrict__ senders, int32_t* __restrict__ receivers,
                 float* __restrict__ efeat, double* __restrict__ efeat64, int64_t e_alloc) {
  if (ctrl->overflow_step >= 0) return;
  const int64_t gnode = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  if (gnode >= BN) return;
  const int d = min(deg[gnode], maxd);
  const int64_t base = row_ptr[gnode];
  for (int k = threadIdx.x & 15; k < d; k += 16) {
    const int64_t src = gnode * maxd + k, dst = base + k;
    if (dst >= e_alloc) break;
    senders[dst] = tsend[src];
    receivers[dst] = (int32_t)gnode;
    f32x4* ef = reinterpret_cast<f32x4*>(efeat + dst * 8);
    ef[0] = reinterpret_cast<const f32x4*>(tfeat)[src];
    ef[1] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (efeat64 && tfeat64)
      for (int c = 0; c < 4; ++c) efeat64[dst * 4 + c] = tfeat64[src * 4 + c];
  }
}

// NL_ROWS -> CSR for batches of at most LB_CSCAN_N nodes: degree scan, k_row_finish and the compaction in ONE launch
// (they were four: two scan passes, finish, compact - ~19 us of a 0.6 ms step on one 8 k-particle trajectory).  Every
// workgroup adds up the degrees of the nodes before its own 16 itself (<= 64 KiB of L2-resident integers, 16-byte
// loads); the last workgroup, which thereby knows every total, does k_row_finish's job.
#define LB_CSCAN_N 16384
__global__ void __launch_bounds__(256)
    k_nl_compact_scan(lb_geom g, int64_t BN, lb_ctrl* __restrict__ ctrl, const int32_t* __restrict__ deg,
                      int32_t* __restrict__ row_ptr, int32_t maxd, const int32_t* __restrict__ tsend,
                      const float* __restrict__ tfeat, const double* __restrict__ tfeat64,
                      int32_t* __restrict__ senders, int32_t* __restrict__ receivers, float* __restrict__ efeat,
                      double* __restrict__ efeat64, int64_t e_alloc, int32_t* __restrict__ overflow,
                      int32_t* __restrict__ nedges_b, int32_t cell_capacity, int32_t e_cap, int32_t* host_flag) {
  __shared__ int s_red[4], s_deg[16], s_b[64];
  if (ctrl->overflow_step >= 0) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int node0 = blockIdx.x * 16;
  const bool last = blockIdx.x == gridDim.x - 1;
  // sum of deg[0 .. node0): node0 is a multiple of 16, so whole int4s
  typedef int i32x4c __attribute__((ext_vector_type(4)));
  const i32x4c* d4 = reinterpret_cast<const i32x4c*>(deg);
  int part = 0;
  for (int i = tid; i < (node0 >> 2); i += 256) {
    const i32x4c v = d4[i];
    part += (v[0] + v[1]) + (v[2] + v[3]);
  }
  for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
  if (lane == 0) s_red[wave] = part;
  if (tid < 16) s_deg[tid] = (node0 + tid < BN) ? deg[node0 + tid] : 0;
  __syncthreads();
  const int base0 = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
  const int q = tid >> 4;  // this 16-lane group's node
  int base = base0;
  for (int j = 0; j < q; ++j) base += s_deg[j];
  const int64_t gnode = (int64_t)node0 + q;
  if (gnode < BN) {
    if ((tid & 15) == 0) row_ptr[gnode] = base;
    const int d = min(s_deg[q], maxd);
    for (int k = tid & 15; k < d; k += 16) {
      const int64_t src = gnode * maxd + k, dst = (int64_t)base + k;
      if (dst >= e_alloc) break;
      senders[dst] = tsend[src];
      receivers[dst] = (int32_t)gnode;
      f32x4* ef = reinterpret_cast<f32x4*>(efeat + dst * 8);
      ef[0] = reinterpret_cast<const f32x4*>(tfeat)[src];
      ef[1] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (efeat64 && tfeat64)
        for (int c = 0; c < 4; ++c) efeat64[dst * 4 + c] = tfeat64[src * 4 + c];
    }
  }
  if (!last) return;
  // ---- k_row_finish's job: per-trajectory edge counts (B <= 64 here), overflow flags, the control block
  int total = base0;
  for (int j = 0; j < 16; ++j) total += s_deg[j];
  for (int b = 0; b < g.B; ++b) {
    int eb = 0;
    for (int i = tid; i < g.N; i += 256) eb += deg[(int64_t)b * g.N + i];
    for (int off = 32; off > 0; off >>= 1) eb += __shfl_xor(eb, off);
    __syncthreads();
    if (lane == 0) s_red[wave] = eb;
    __syncthreads();
    if (tid == 0) s_b[b] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
  }
  __syncthreads();
  if (tid == 0) {
    int any = 0;
    for (int b = 0; b < g.B; ++b) {
      const int eb = s_b[b];
      nedges_b[b] = eb;
      const int ov = (eb > e_cap) || (g.use_cell_list && ctrl->max_cell_occ > cell_capacity) || ctrl->row_overflow;
      overflow[b] = ov;
      any |= ov;
    }
    row_ptr[BN] = total;
    ctrl->n_edges_unclamped = total;
    lb_acct_edges(ctrl, total);
    ctrl->n_edges_total = (int)min((int64_t)total, e_alloc);
    if ((any || (int64_t)total > e_alloc) && ctrl->overflow_step < 0) {
      ctrl->overflow_step = ctrl->step;
      if (host_flag) *host_flag = ctrl->step;
    }
  }
}

// ------------------------------------------------------------------------------- row scan
// After the two-level scan of the degrees: per-trajectory edge counts, did_buffer_overflow flags
// and the control block, all on the device (no host sync).
// (shared by k_row_finish and the single-workgroup k_rows_small; called by every thread of ONE workgroup)
__device__ __forceinline__ void lb_row_finish_body(const lb_geom& g, const int32_t* __restrict__ row_ptr, int n,
                                                   lb_ctrl* __restrict__ ctrl, int32_t* __restrict__ overflow,
                                                   int32_t* __restrict__ nedges_b, int32_t cell_capacity,
                                                   int32_t e_cap, int64_t e_alloc, int frozen, int32_t* host_flag,
                                                   int* s_any) {
  if (threadIdx.x == 0) *s_any = 0;
  __syncthreads();
  for (int b = threadIdx.x; b < g.B; b += blockDim.x) {
    const int eb = row_ptr[(b + 1) * g.N] - row_ptr[b * g.N];
    nedges_b[b] = eb;
    // did_buffer_overflow = cell list overflow | occupancy > max_occupancy (jax-md); a row that
    // outgrew the engine's per-node slots is reported the same way (the driver re-allocates)
    int ov = 0;
    if (frozen)
      ov = (eb > e_cap) || (g.use_cell_list && ctrl->max_cell_occ > cell_capacity) || ctrl->row_overflow;
    overflow[b] = ov;
    if (ov) atomicExch(s_any, 1);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int total = row_ptr[n];
    ctrl->n_edges_unclamped = total;
    lb_acct_edges(ctrl, total);
    ctrl->n_edges_total = (int)min((int64_t)total, e_alloc);
    if (frozen && (*s_any || (int64_t)total > e_alloc) && ctrl->overflow_step < 0) {
      ctrl->overflow_step = ctrl->step;
      if (host_flag) *host_flag = ctrl->step;  // pinned host memory: visible once this kernel retires
    }
  }
}

__global__ void __launch_bounds__(256)
    k_row_finish(lb_geom g, const int32_t* __restrict__ row_ptr, int n, lb_ctrl* __restrict__ ctrl,
                 int32_t* __restrict__ overflow, int32_t* __restrict__ nedges_b,
                 int32_t cell_capacity, int32_t e_cap, int64_t e_alloc, int frozen,
                 int32_t* host_flag) {
  if (ctrl->overflow_step >= 0) return;
  __shared__ int s_any;
  lb_row_finish_body(g, row_ptr, n, ctrl, overflow, nedges_b, cell_capacity, e_cap, e_alloc, frozen, host_flag,
                     &s_any);
}

// Small problems: degree scan (both passes) + k_row_finish in one single-workgroup launch.
__global__ void __launch_bounds__(LB_SMALL_T)
    k_rows_small(lb_geom g, const int32_t* __restrict__ deg, int32_t* __restrict__ row_ptr, int n,
                 lb_ctrl* __restrict__ ctrl, int32_t* __restrict__ overflow, int32_t* __restrict__ nedges_b,
                 int32_t cell_capacity, int32_t e_cap, int64_t e_alloc, int frozen, int32_t* host_flag) {
  __shared__ int s_scan[LB_SMALL_T];
  __shared__ int s_any;
  if (ctrl->overflow_step >= 0) return;
  const int tid = threadIdx.x;
  constexpr int PER = LB_SMALL_N / LB_SMALL_T;
  __shared__ int s_maxdeg;
  if (tid == 0) s_maxdeg = 0;
  __syncthreads();
  int v[PER], sum = 0, mx = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = tid * PER + k;
    v[k] = i < n ? deg[i] : 0;
    sum += v[k];
    mx = max(mx, v[k]);
  }
  s_scan[tid] = sum;
  if (mx > 0) atomicMax(&s_maxdeg, mx);
  __syncthreads();
  if (tid == 0) ctrl->max_deg = s_maxdeg;  // max receiver degree of this build (the search kernels no longer track it)
  for (int off = 1; off < LB_SMALL_T; off <<= 1) {
    const int add = (tid >= off) ? s_scan[tid - off] : 0;
    __syncthreads();
    s_scan[tid] += add;
    __syncthreads();
  }
  int run = s_scan[tid] - sum;
#pragma unroll
  for (int k = 0; k < PER; ++k) {
    const int i = tid * PER + k;
    if (i < n) row_ptr[i] = run;
    run += v[k];
  }
  if (tid == LB_SMALL_T - 1) row_ptr[n] = s_scan[tid];
  __syncthreads();  // row_ptr written by this workgroup is visible to it
  lb_row_finish_body(g, row_ptr, n, ctrl, overflow, nedges_b, cell_capacity, e_cap, e_alloc, frozen, host_flag,
                     &s_any);
}

// ---------------------------------------------------------------- one trajectory, <= 4096 particles: ONE launch
// Round 3.  On a 2.5 k-particle trajectory the update path above is four dependent launches (cells -> search ->
// degree scan -> compaction: 44 us of a 310 us step, every one a chain of 3 - 5 memory round trips).  Here the
// whole build is one launch with one dependency level:
//   * every workgroup stages ALL N newest-frame positions + cell coordinates in LDS (N * (8 dim + 4) bytes);
//   * every workgroup also builds, per dimension, one N-bit mask per cell coordinate ("the particles whose x cell is
//     v") with LDS atomics; a receiver's stencil candidates are then (X[x-1]|X[x]|X[x+1]) & (Y..) & (Z..) - lane w
//     of its wave owns the 64 ids of word w - expanded in ascending id order into the wave's row buffer, and the
//     fp64 predicate (same operand order as k_nl) runs on that short list only.  Same candidate set as the rolled
//     3^dim stencil of k_nl (the masks wrap); the hits come out sorted by sender id, so the rank pass is gone;
//   * rows go straight into the CSR arrays: a workgroup publishes the edge count of its NLS_WAVES receivers in one
//     word (build epoch << 16 | count, agent-scope release store) and adds up the words of the workgroups before it
//     (workgroups are dispatched in index order, so a predecessor is running or done; bounded spin);
//   * the last workgroup also histograms the cells (max_cell_occ for the did_buffer_overflow flag) and does
//     k_row_finish's job.
// cell coordinate of a position: int(position / cell_size) clamped to the grid (k_cells_small's arithmetic).  The fp64
// quotient decides only near an integer: a float product locates the value first (|error| < 4e-4 for |q| < 2000), and
// only inside a 1e-3 band around an integer - or far outside the grid - is the exact (in f32 mode: float-rounded)
// quotient evaluated.  Every workgroup of the single-launch builds recomputes ALL N coordinates, and in fp64 that
// was VALU time (k_nl_mid: 8.5 of 35 us).
#define NLS_WAVES 16  // at most; the launch uses ceil(N / 256) waves so that every CU gets at most one workgroup
#define NLS_THREADS (64 * NLS_WAVES)
#define NLS_CAND 512  // stencil candidates per receiver (row buffer entries; >= LB_MAX_ROW)
struct lb_nls_args {
  const double* win;
  int32_t *senders, *receivers;
  float* efeat;
  double* efeat64;
  int32_t *deg, *row_ptr, *overflow, *nedges_b;
  unsigned long long* wg_sum;  // [ceil(N / NLS_WAVES)]
  int64_t e_alloc;
  int32_t e_cap, cell_capacity, npad;
  int32_t* host_flag;
  lb_feat_job feat;  // node features of this step: every wave writes the row of its receiver (xnode == null: no job)
  long long* dbg;  // -DLB_MS_STAMPS builds: [3 workgroups][8] wall-clock stamps (first, middle, last workgroup)
};
#define NLS_STAMP(k)                                                                                       \
  do {                                                                                                     \
    if (a.dbg && tid == 0) {                                                                               \
      const int wsel = blockIdx.x == 0 ? 0 : ((int)blockIdx.x == (int)gridDim.x / 2 ? 1 : (last ? 2 : -1)); \
      if (wsel >= 0) a.dbg[wsel * 8 + (k)] = wall_clock64();                                               \
    }                                                                                                      \
  } while (0)

// DIM is a template parameter: under a loop bounded by the run-time dimension the per-dimension register arrays are
// indexed by a run-time value and land in scratch (144 B per lane, every access a memory round trip)
template <bool F32, int DIM>
__global__ void __launch_bounds__(NLS_THREADS) k_nl_small(lb_geom g, lb_ctrl* __restrict__ ctrl, lb_nls_args a) {
  extern __shared__ double s_dynd[];
  if (a.dbg && threadIdx.x == 0 && (blockIdx.x == 0 || blockIdx.x == gridDim.x - 1))
    a.dbg[24 + (blockIdx.x == 0 ? 0 : 1)] = wall_clock64();
  if (ctrl->overflow_step >= 0) return;
  const int N = g.N, npad = a.npad;
  const int nwv = blockDim.x >> 6, nwords = npad >> 6;
  double* const s_p = s_dynd;  // [dim][npad]
  unsigned long long* const s_tab = reinterpret_cast<unsigned long long*>(s_p + DIM * npad);  // per dim [ncell[d]][nwords]
  const int tab_off[3] = {0, g.ncell[0] * nwords, (g.ncell[0] + g.ncell[1]) * nwords};
  const int tab_len = (g.ncell[0] + g.ncell[1] + (DIM == 3 ? g.ncell[2] : 0)) * nwords;
  int* const s_cell = reinterpret_cast<int*>(s_tab + (g.use_cell_list ? tab_len : 0));  // [npad] x | y << 11 | z << 22
  int* const s_row = s_cell + npad;                                                      // [waves][NLS_CAND]
  int* const s_cnt = s_row + nwv * NLS_CAND;  // [0..15] row sizes, [16] base, [17] flags, [18] max occ
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int step = ctrl->step;
  const unsigned long long epoch = (unsigned long long)((uint32_t)ctrl->nl_epoch & 0xffffu);
  const bool last = blockIdx.x == gridDim.x - 1;
  NLS_STAMP(0);
  if (tid < 32) s_cnt[tid] = 0;
  if (g.use_cell_list)
    for (int k = tid; k < tab_len; k += blockDim.x) s_tab[k] = 0ull;
  // ---- stage positions + cell coordinates (k_cells_small's arithmetic: int(position / cell_size), clamped);
  // all loads of a thread are issued before the first is used
  {
    constexpr int PER = 4;  // npad <= 256 * waves
    const int slot = (step + g.isl - 1) % g.isl;
    const double* const w0 = a.win + (int64_t)slot * DIM * N;
    double pv[PER][3];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = tid + (int)blockDim.x * k;
      const int ic = i < N ? i : N - 1;
      pv[k][0] = w0[ic];
      pv[k][1] = w0[N + ic];
      pv[k][2] = DIM == 3 ? w0[2 * N + ic] : 0.0;
    }
    float inv_cs[3];
    _Pragma("unroll") for (int d = 0; d < 3; ++d) inv_cs[d] = d < DIM ? (float)(1.0 / g.cell_size[d]) : 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = tid + (int)blockDim.x * k;
      if (i < npad) {
        int packed = 0;
        _Pragma("unroll") for (int d = 0; d < DIM; ++d) {
          const double p = pv[k][d];
          s_p[d * npad + i] = p;
          packed |= lb_cell_coord<F32>(p, inv_cs[d], g.cell_size[d], g.ncell[d]) << (11 * d);
        }
        s_cell[i] = i < N ? packed : -1;
      }
    }
  }
  __syncthreads();  // (positions + cells staged, tables zeroed)
  // the per-dimension masks: lane l owns word l; in step k it takes particle 64 l + ((k + l) & 63) - 64 different LDS
  // banks for the cell reads, 64 different words for the ORs (atomics only against the other waves' steps)
  if (g.use_cell_list && lane < nwords) {
    for (int k = wave; k < 64; k += nwv) {
      const int b = (k + lane) & 63;
      const int pc = s_cell[lane * 64 + b];
      if (pc >= 0)
        _Pragma("unroll") for (int d = 0; d < DIM; ++d)
          atomicOr(&s_tab[tab_off[d] + ((pc >> (11 * d)) & 0x7ff) * nwords + lane], 1ull << b);
    }
  }
  __syncthreads();
  NLS_STAMP(1);
  // ---- pass 1 (bit masks): the candidates of the receiver's 3^dim stencil cells, in id order, into the row buffer
  const int r = blockIdx.x * nwv + wave;
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  int count = 0, same = 0, flags = 0;
  double pr[3] = {0, 0, 0};
  int* const row = s_row + wave * NLS_CAND;
  if (r < N) {
    if (a.feat.xnode) lb_node_features_wave(g, N, a.win, step, a.feat, r);
    _Pragma("unroll") for (int d = 0; d < DIM; ++d) pr[d] = s_p[d * npad + r];
    int ncand = N;  // (no cell list: every particle is a candidate, pass 2 walks the ids)
    if (g.use_cell_list) {
      const int pc = s_cell[r];
      unsigned long long m = ~0ull, ms = ~0ull;
      if (lane < nwords) {
        _Pragma("unroll") for (int d = 0; d < DIM; ++d) {
          const int n = g.ncell[d], c = (pc >> (11 * d)) & 0x7ff;
          const unsigned long long* T = s_tab + tab_off[d] + lane;
          const unsigned long long t0 = T[c * nwords];  // the stencil wraps like jax-md's rolled cell buffer
          m &= t0 | T[(c == 0 ? n - 1 : c - 1) * nwords] | T[(c == n - 1 ? 0 : c + 1) * nwords];
          ms &= t0;
        }
      } else {
        m = 0ull;
        ms = 0ull;
      }
      int mine = __popcll(m), off = mine;
      same = __popcll(ms);
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(off, o);
        if (lane >= o) off += v;
      }
      ncand = __shfl(off, 63);
      off -= mine;
      for (int o = 32; o > 0; o >>= 1) same += __shfl_xor(same, o);
      while (m) {
        const int b = __ffsll((long long)m) - 1;
        if (off < NLS_CAND) row[off] = lane * 64 + b;
        ++off;
        m &= m - 1;
      }
    }
    if (g.use_cell_list && ncand > NLS_CAND) {  // (the host checks cell_capacity * 3^dim <= NLS_CAND; flagged like an over-long row)
      flags = 1;
      ncand = NLS_CAND;
    }
    // ---- pass 2 (fp64): the predicate on the candidates; the hits overwrite the front of the row buffer in order
    for (int c0 = 0; c0 < ncand; c0 += 64) {
      const int t = c0 + lane;
      bool ok = false;
      int j = 0;
      if (t < ncand) {
        j = g.use_cell_list ? row[t] : t;
        // metric_sq(position[sender], position[receiver]) as in k_nl
        double dd = lb_disp1(s_p[j], pr[0], g.box[0], g.half_box[0], g.periodic, F32);
        double d2 = lb_r(dd * dd, F32);
        _Pragma("unroll") for (int d = 1; d < DIM; ++d) {
          dd = lb_disp1(s_p[d * npad + j], pr[d], g.box[d], g.half_box[d], g.periodic, F32);
          d2 = lb_r(d2 + lb_r(dd * dd, F32), F32);
        }
        ok = d2 < g.rc2;
      }
      const unsigned long long mask = __ballot(ok);
      // (in place: hit k of this sweep lands at count + k <= c0 + lane's own index, and every lane has read its j)
      if (ok) {
        const int pos = count + __popcll(mask & lt_mask);
        if (pos < NLS_CAND) row[pos] = j;
      }
      count += __popcll(mask);
    }
    if (count > LB_MAX_ROW) flags = 1;  // re-allocate (-> the dense fall-back), like k_nl
    if (lane == 0) {
      a.deg[r] = count;
      s_cnt[wave] = count;
      if (flags) atomicOr(&s_cnt[17], 1);
      atomicMax(&s_cnt[18], same);
    }
  }
  __syncthreads();
  NLS_STAMP(3);
  // ---- this workgroup's word out (epoch | max occupancy seen | overflow | edge count), its predecessors' in
  if (tid == 0) {
    int tot = 0;
    for (int w = 0; w < nwv; ++w) tot += s_cnt[w];
    const unsigned long long word = (epoch << 48) | ((unsigned long long)(s_cnt[18] & 0xffff) << 32) |
                                    ((unsigned long long)(s_cnt[17] & 1) << 31) | (unsigned long long)(tot & 0x7fffffff);
    __hip_atomic_store(&a.wg_sum[blockIdx.x], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  NLS_STAMP(4);
  {
    int part = 0, occ = 0, ovf = 0;
    bool timed_out = false;
    for (int p = tid; p < (int)blockIdx.x; p += (int)blockDim.x) {
      unsigned long long v = 0;
      int spins = 0;
      do {
        v = __hip_atomic_load(&a.wg_sum[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((v >> 48) == epoch) break;
        __builtin_amdgcn_s_sleep(1);
      } while (++spins < (1 << 22));
      if ((v >> 48) != epoch) timed_out = true;
      part += (int)(v & 0x7fffffffu);
      ovf |= (int)((v >> 31) & 1);
      occ = max(occ, (int)((v >> 32) & 0xffff));
    }
    if (timed_out) {
      atomicExch(&ctrl->persist_error, 2);
      atomicMin(&ctrl->persist_step, step);
    }
    for (int off = 32; off > 0; off >>= 1) {
      part += __shfl_xor(part, off);
      ovf |= __shfl_xor(ovf, off);
      occ = max(occ, __shfl_xor(occ, off));
    }
    if (lane == 0 && (int)blockIdx.x > wave * 64) {
      if (part) atomicAdd(&s_cnt[16], part);
      if (ovf) atomicOr(&s_cnt[17], 1);
      atomicMax(&s_cnt[18], occ);
    }
  }
  __syncthreads();
  NLS_STAMP(5);
  int base = s_cnt[16];
  for (int w = 0; w < wave; ++w) base += s_cnt[w];
  // ---- rows straight into the CSR arrays (ascending sender id by construction)
  if (r < N) {
    if (lane == 0) a.row_ptr[r] = base;
    const int cnt = min(count, LB_MAX_ROW);
    for (int t0 = 0; t0 < cnt; t0 += 64) {
      const int t = t0 + lane;
      if (t >= cnt) break;
      const int j = row[t];
      const int64_t slot = (int64_t)base + t;
      if (slot >= a.e_alloc) break;
      a.senders[slot] = j;
      a.receivers[slot] = r;
      // features.py:115-124: disp(pos[receiver], pos[sender]) / r_c and its norm
      double rd[3] = {0, 0, 0};
      double s2 = 0.0;
      _Pragma("unroll") for (int d = 0; d < DIM; ++d) {
        rd[d] = lb_r(lb_disp1(pr[d], s_p[d * npad + j], g.box[d], g.half_box[d], g.periodic, F32) / g.rc, F32);
        s2 = (d == 0) ? lb_r(rd[d] * rd[d], F32) : lb_r(s2 + lb_r(rd[d] * rd[d], F32), F32);
      }
      const double dist = s2 > 0.0 ? lb_r(sqrt(s2), F32) : 0.0;
      f32x4* ef = reinterpret_cast<f32x4*>(a.efeat + slot * 8);
      ef[0] = (DIM == 2) ? f32x4{(float)rd[0], (float)rd[1], (float)dist, 0.f}
                           : f32x4{(float)rd[0], (float)rd[1], (float)rd[2], (float)dist};
      ef[1] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (a.efeat64) {
        double* e64 = a.efeat64 + slot * 4;
        e64[0] = rd[0];
        e64[1] = rd[1];
        e64[2] = rd[2];
        e64[3] = dist;
      }
    }
  }
  NLS_STAMP(6);
  // ---- k_row_finish's job, by the workgroup that knows the total
  if (last && tid == 0) {
    int total = s_cnt[16];
    for (int w = 0; w < nwv; ++w) total += s_cnt[w];
    a.row_ptr[N] = total;
    a.nedges_b[0] = total;
    const int max_occ = s_cnt[18];
    const int ov = (total > a.e_cap) || (g.use_cell_list && max_occ > a.cell_capacity) || (s_cnt[17] & 1);
    a.overflow[0] = ov;
    ctrl->max_cell_occ = max_occ;
    ctrl->n_edges_unclamped = total;
    lb_acct_edges(ctrl, total);
    ctrl->n_edges_total = (int)min((int64_t)total, a.e_alloc);
    if ((ov || (int64_t)total > a.e_alloc) && ctrl->overflow_step < 0) {
      ctrl->overflow_step = step;
      if (a.host_flag) *a.host_flag = step;
    }
    int ne = ctrl->nl_epoch + 1;
    if ((ne & 0xffff) == 0) ++ne;
    ctrl->nl_epoch = ne;
  }
  NLS_STAMP(7);
}

// ----------------------------------------------- one trajectory, 4096 < N <= 8192 particles (3D: 8 k): k_nl_mid
// k_nl_small's scheme where the positions no longer fit LDS next to the masks: only the per-dimension masks, the cell
// coordinates and the row buffers are staged (TGV3D-8k: 39 + 32 + 32 KiB); candidates' positions come from the window in
// global memory (L2) for the short candidate list only.  A lane owns up to two mask words (ids 64 l .. and 64 (l + 64)
// ..), a wave searches NLM_RPW consecutive receivers one after the other, a workgroup of 16 waves owns 16 NLM_RPW
// consecutive receivers (contiguous, so the CSR offsets work as in k_nl_small).  Replaces, on one 8 k-particle
// trajectory, memset + cell count + 2 scan passes + cell fill + k_nlw + scan / finish / compact: 9 launches.
#define NLM_N 8192
#define NLM_WAVES 16
#define NLM_RPW 2
#define NLM_ROWBUF 1024  // row buffer per wave: the hits of the receivers done (<= 256 each) + 768 candidates
template <bool F32, int DIM>
__global__ void __launch_bounds__(64 * NLM_WAVES) k_nl_mid(lb_geom g, lb_ctrl* __restrict__ ctrl, lb_nls_args a) {
  extern __shared__ double s_dynd[];
  if (ctrl->overflow_step >= 0) return;
  const int N = g.N, npad = a.npad, nwords = npad >> 6;
  unsigned long long* const s_tab = reinterpret_cast<unsigned long long*>(s_dynd);  // per dim [ncell[d]][nwords]
  const int tab_off[3] = {0, g.ncell[0] * nwords, (g.ncell[0] + g.ncell[1]) * nwords};
  const int tab_len = (g.ncell[0] + g.ncell[1] + (DIM == 3 ? g.ncell[2] : 0)) * nwords;
  int* const s_cell = reinterpret_cast<int*>(s_tab + tab_len);  // [npad] x | y << 11 | z << 22 (pad: -1)
  int* const s_row = s_cell + npad;                              // [NLM_WAVES][NLM_ROWBUF]
  int* const s_cnt = s_row + NLM_WAVES * NLM_ROWBUF;               // [0..31] row sizes, [32] base, [33] flags, [34] max occ
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int step = ctrl->step;
  const unsigned long long epoch = (unsigned long long)((uint32_t)ctrl->nl_epoch & 0xffffu);
  const bool last = blockIdx.x == gridDim.x - 1;
  if (tid < 48) s_cnt[tid] = 0;
  for (int k = tid; k < tab_len; k += blockDim.x) s_tab[k] = 0ull;
  const int slot = (step + g.isl - 1) % g.isl;
  const double* const w0 = a.win + (int64_t)slot * DIM * N;
  {
    constexpr int PER = NLM_N / (64 * NLM_WAVES);
    double pv[PER][3];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = tid + 64 * NLM_WAVES * k;
      const int ic = i < N ? i : N - 1;
      pv[k][0] = w0[ic];
      pv[k][1] = w0[N + ic];
      pv[k][2] = DIM == 3 ? w0[2 * N + ic] : 0.0;
    }
    float inv_cs[3];
    _Pragma("unroll") for (int d = 0; d < 3; ++d) inv_cs[d] = d < DIM ? (float)(1.0 / g.cell_size[d]) : 0.f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
      const int i = tid + 64 * NLM_WAVES * k;
      if (i < npad) {
        int packed = 0;
        _Pragma("unroll") for (int d = 0; d < DIM; ++d)
          packed |= lb_cell_coord<F32>(pv[k][d], inv_cs[d], g.cell_size[d], g.ncell[d]) << (11 * d);
        s_cell[i] = i < N ? packed : -1;
      }
    }
  }
  __syncthreads();
  // masks: lane l owns words l and l + 64; in step k it takes particle 64 word + ((k + l) & 63)
  for (int wi = 0; wi < 2; ++wi) {
    const int word = lane + 64 * wi;
    if (word < nwords)
      for (int k = wave; k < 64; k += NLM_WAVES) {
        const int b = (k + lane) & 63;
        const int pc = s_cell[word * 64 + b];
        if (pc >= 0)
          _Pragma("unroll") for (int d = 0; d < DIM; ++d)
            atomicOr(&s_tab[tab_off[d] + ((pc >> (11 * d)) & 0x7ff) * nwords + word], 1ull << b);
      }
  }
  __syncthreads();
  const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
  const int r0 = (blockIdx.x * NLM_WAVES + wave) * NLM_RPW;
  int* const row = s_row + wave * NLM_ROWBUF;
  int cnt_i[NLM_RPW], flags = 0, same_max = 0;
  // pass 1 + 2 per receiver; the hits of receiver i stay in row[hit_base[i] ..)
  int hit_base[NLM_RPW + 1];
  hit_base[0] = 0;
#pragma unroll
  for (int i = 0; i < NLM_RPW; ++i) {
    const int r = r0 + i;
    cnt_i[i] = 0;
    hit_base[i + 1] = hit_base[i];
    if (r >= N) continue;
    if (a.feat.xnode) lb_node_features_wave(g, N, a.win, step, a.feat, r);
    double pr[3] = {0, 0, 0};
    _Pragma("unroll") for (int d = 0; d < DIM; ++d) pr[d] = w0[d * N + r];
    const int pc = s_cell[r];
    unsigned long long m[2], ms[2];
#pragma unroll
    for (int wi = 0; wi < 2; ++wi) {
      const int word = lane + 64 * wi;
      m[wi] = ~0ull;
      ms[wi] = ~0ull;
      if (word < nwords) {
        _Pragma("unroll") for (int d = 0; d < DIM; ++d) {
          const int n = g.ncell[d], c = (pc >> (11 * d)) & 0x7ff;
          const unsigned long long* T = s_tab + tab_off[d] + word;
          const unsigned long long t0 = T[c * nwords];
          m[wi] &= t0 | T[(c == 0 ? n - 1 : c - 1) * nwords] | T[(c == n - 1 ? 0 : c + 1) * nwords];
          ms[wi] &= t0;
        }
      } else {
        m[wi] = 0ull;
        ms[wi] = 0ull;
      }
    }
    // offsets: all ids of the first 64 words come before those of the second 64: two scans in one (16 + 16 bits)
    const int mine = __popcll(m[0]) | (__popcll(m[1]) << 16);
    int sc = mine, same = __popcll(ms[0]) + __popcll(ms[1]);
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(sc, o);
      if (lane >= o) sc += v;
    }
    const int tot = __shfl(sc, 63);
    const int tot0 = tot & 0xffff, ncand_all = tot0 + (tot >> 16);
    for (int o = 32; o > 0; o >>= 1) same += __shfl_xor(same, o);
    same_max = max(same_max, same);
    const int cbase = hit_base[i];  // candidates are expanded behind the previous receiver's hits
    int off0 = cbase + (sc & 0xffff) - (mine & 0xffff), off1 = cbase + tot0 + (sc >> 16) - (mine >> 16);
    const int room = NLM_ROWBUF;
    unsigned long long mm = m[0];
    while (mm) {
      const int b = __ffsll((long long)mm) - 1;
      if (off0 < room) row[off0] = lane * 64 + b;
      ++off0;
      mm &= mm - 1;
    }
    mm = m[1];
    while (mm) {
      const int b = __ffsll((long long)mm) - 1;
      if (off1 < room) row[off1] = (lane + 64) * 64 + b;
      ++off1;
      mm &= mm - 1;
    }
    int ncand = ncand_all;
    if (cbase + ncand > room) {
      flags = 1;
      ncand = max(room - cbase, 0);
    }
    int count = 0;
    // four sweeps of 64 candidates at a time: their position loads (L2) are in flight together - one sweep at a time
    // every sweep was a memory round trip of its own (9 us per receiver)
    for (int c0 = 0; c0 < ncand; c0 += 256) {
      int jj[4];
      double pj[4][DIM];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = c0 + 64 * u + lane;
        jj[u] = t < ncand ? row[cbase + t] : r;
        _Pragma("unroll") for (int d = 0; d < DIM; ++d) pj[u][d] = w0[d * N + jj[u]];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = c0 + 64 * u + lane;
        bool ok = false;
        if (t < ncand) {
          double dd = lb_disp1(pj[u][0], pr[0], g.box[0], g.half_box[0], g.periodic, F32);
          double d2 = lb_r(dd * dd, F32);
          _Pragma("unroll") for (int d = 1; d < DIM; ++d) {
            dd = lb_disp1(pj[u][d], pr[d], g.box[d], g.half_box[d], g.periodic, F32);
            d2 = lb_r(d2 + lb_r(dd * dd, F32), F32);
          }
          ok = d2 < g.rc2;
        }
        const unsigned long long mask = __ballot(ok);
        // in place: a hit lands at or before the slot its candidate was read from, and this batch has been read
        if (ok) row[cbase + count + __popcll(mask & lt_mask)] = jj[u];
        count += __popcll(mask);
      }
    }
    if (count > LB_MAX_ROW) flags = 1;
    cnt_i[i] = count;
    hit_base[i + 1] = cbase + min(count, LB_MAX_ROW);
    if (lane == 0) {
      a.deg[r] = count;
      s_cnt[wave * NLM_RPW + i] = count;
    }
  }
  if (lane == 0) {
    if (flags) atomicOr(&s_cnt[33], 1);
    atomicMax(&s_cnt[34], same_max);
  }
  __syncthreads();
  if (tid == 0) {
    int tot = 0;
    for (int w = 0; w < NLM_WAVES * NLM_RPW; ++w) tot += s_cnt[w];
    const unsigned long long word = (epoch << 48) | ((unsigned long long)(s_cnt[34] & 0xffff) << 32) |
                                    ((unsigned long long)(s_cnt[33] & 1) << 31) | (unsigned long long)(tot & 0x7fffffff);
    __hip_atomic_store(&a.wg_sum[blockIdx.x], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  {
    int part = 0, occ = 0, ovf = 0;
    bool timed_out = false;
    for (int p = tid; p < (int)blockIdx.x; p += (int)blockDim.x) {
      unsigned long long v = 0;
      int spins = 0;
      do {
        v = __hip_atomic_load(&a.wg_sum[p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((v >> 48) == epoch) break;
        __builtin_amdgcn_s_sleep(1);
      } while (++spins < (1 << 22));
      if ((v >> 48) != epoch) timed_out = true;
      part += (int)(v & 0x7fffffffu);
      ovf |= (int)((v >> 31) & 1);
      occ = max(occ, (int)((v >> 32) & 0xffff));
    }
    if (timed_out) {
      atomicExch(&ctrl->persist_error, 2);
      atomicMin(&ctrl->persist_step, step);
    }
    for (int off = 32; off > 0; off >>= 1) {
      part += __shfl_xor(part, off);
      ovf |= __shfl_xor(ovf, off);
      occ = max(occ, __shfl_xor(occ, off));
    }
    if (lane == 0 && (int)blockIdx.x > wave * 64) {
      if (part) atomicAdd(&s_cnt[32], part);
      if (ovf) atomicOr(&s_cnt[33], 1);
      atomicMax(&s_cnt[34], occ);
    }
  }
  __syncthreads();
  int base = s_cnt[32];
  for (int w = 0; w < wave * NLM_RPW; ++w) base += s_cnt[w];
#pragma unroll
  for (int i = 0; i < NLM_RPW; ++i) {
    const int r = r0 + i;
    if (r >= N) continue;
    if (lane == 0) a.row_ptr[r] = base;
    double pr[3] = {0, 0, 0};
    _Pragma("unroll") for (int d = 0; d < DIM; ++d) pr[d] = w0[d * N + r];
    const int cnt = min(cnt_i[i], LB_MAX_ROW);
    for (int t0 = 0; t0 < cnt; t0 += 64) {
      const int t = t0 + lane;
      if (t >= cnt) break;
      const int j = row[hit_base[i] + t];
      const int64_t slot_e = (int64_t)base + t;
      if (slot_e >= a.e_alloc) break;
      a.senders[slot_e] = j;
      a.receivers[slot_e] = r;
      double rd[3] = {0, 0, 0};
      double s2 = 0.0;
      _Pragma("unroll") for (int d = 0; d < DIM; ++d) {
        rd[d] = lb_r(lb_disp1(pr[d], w0[d * N + j], g.box[d], g.half_box[d], g.periodic, F32) / g.rc, F32);
        s2 = (d == 0) ? lb_r(rd[d] * rd[d], F32) : lb_r(s2 + lb_r(rd[d] * rd[d], F32), F32);
      }
      const double dist = s2 > 0.0 ? lb_r(sqrt(s2), F32) : 0.0;
      f32x4* ef = reinterpret_cast<f32x4*>(a.efeat + slot_e * 8);
      ef[0] = (DIM == 2) ? f32x4{(float)rd[0], (float)rd[1], (float)dist, 0.f}
                         : f32x4{(float)rd[0], (float)rd[1], (float)rd[2], (float)dist};
      ef[1] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (a.efeat64) {
        double* e64 = a.efeat64 + slot_e * 4;
        e64[0] = rd[0];
        e64[1] = rd[1];
        e64[2] = rd[2];
        e64[3] = dist;
      }
    }
    base += cnt_i[i];
  }
  if (last && tid == 0) {
    int total = s_cnt[32];
    for (int w = 0; w < NLM_WAVES * NLM_RPW; ++w) total += s_cnt[w];
    a.row_ptr[N] = total;
    a.nedges_b[0] = total;
    const int max_occ = s_cnt[34];
    const int ov = (total > a.e_cap) || (max_occ > a.cell_capacity) || (s_cnt[33] & 1);
    a.overflow[0] = ov;
    ctrl->max_cell_occ = max_occ;
    ctrl->n_edges_unclamped = total;
    lb_acct_edges(ctrl, total);
    ctrl->n_edges_total = (int)min((int64_t)total, a.e_alloc);
    if ((ov || (int64_t)total > a.e_alloc) && ctrl->overflow_step < 0) {
      ctrl->overflow_step = step;
      if (a.host_flag) *a.host_flag = step;
    }
    int ne = ctrl->nl_epoch + 1;
    if ((ne & 0xffff) == 0) ++ne;
    ctrl->nl_epoch = ne;
  }
}

// ----------------------------------------------------------------------------------- host
// `small` = staged-candidate capacity of the one-wave variant to use (a multiple of 128 up to
// NL_SMALL_MAXC), or 0 for the 256-thread / 2048-candidate variant.  The LDS footprint of the
// staged stencil (28 B per candidate) sets how many cells a CU keeps in flight - the search is
// latency bound, so the smallest variant that holds cell_capacity * 3^dim candidates is used.
// which search kernel a build uses: 0 = k_nl (workgroup per cell, staged stencil), 1 = k_nlw (wave per receiver), 2 = k_nlc
// (wave per cell).  Measured (MI355X, B = 8): 3^3-cell stencils favour the per-wave kernels (TGV3D 0.28 -> 0.18 -> 0.12 ms per
// step), 3^2-cell stencils the staged per-cell kernel (DAM2D 0.11 vs 0.15 ms).  LB_NL_KERNEL=cell|wave|nlc overrides.
static int lb_nl_kernel_kind(const lb_engine* e) {
  static const char* force = getenv("LB_NL_KERNEL");
  // dense fall-back (e->nl_dense, sticky): the staged per-cell kernel is bounded by LB_MAX_STENCIL_CAND candidates and
  // LB_MAX_ROW neighbors; beyond that the wave-per-receiver kernel with a row buffer sized from the largest degree runs
  // (the reference re-allocates for ANY occupancy, rollout.py:134-151)
  const bool per_wave = e->nl_dense || e->g.f32 || (force ? force[0] == 'w' : e->g.nstencil == 27);
  const bool cell_wave = e->g.use_cell_list && !e->nl_dense && (e->g.dim == 2 || e->g.dim == 3) && (force ? force[0] == 'n' : per_wave);
  return cell_wave ? 2 : (per_wave ? 1 : 0);
}
template <int MODE>
static void lb_launch_nl(lb_engine* e, int small, const lb_nl_args& a) {
  const int kind = lb_nl_kernel_kind(e);
  const bool per_wave = kind == 1;
  // NL_ROWS in a rollout step: every search wave also writes the node-feature row of its receiver
  const bool ride = MODE == NL_ROWS && e->feat_job.xnode && !a.efeat64;
  if (ride) e->feat_done = true;
  const bool cell_wave = kind == 2;   // round 6: one wave per CELL
  if (cell_wave) {
    const int ncell_all = e->g.B * e->g.ncells;
    const int nb_s = (ncell_all + NLC_WAVES - 1) / NLC_WAVES;
    int nb = nb_s;
    lb_nl_args aw = a;
    if (ride) {
      aw.feat = e->feat_job;
      aw.win = e->win;
      aw.nb_search = nb_s;
      nb = nb_s + (int)((e->BN + NLC_WAVES * NL_FEAT_ROWS - 1) / (NLC_WAVES * NL_FEAT_ROWS));
    }
#define LB_NLC_LAUNCH(F, D) \
  hipLaunchKernelGGL((k_nlc<MODE, F, D>), dim3(nb), dim3(64 * NLC_WAVES), 0, e->stream, e->g, e->BN, e->ctrl, aw)
    if (e->g.f32) {
      if (e->g.dim == 3) LB_NLC_LAUNCH(true, 3); else LB_NLC_LAUNCH(true, 2);
    } else {
      if (e->g.dim == 3) LB_NLC_LAUNCH(false, 3); else LB_NLC_LAUNCH(false, 2);
    }
#undef LB_NLC_LAUNCH
    return;
  }
  if (per_wave) {
    const int nb_s = (int)((e->BN + NLW_WAVES - 1) / NLW_WAVES);
    int nb = nb_s;
    lb_nl_args aw = a;
    if (ride) {
      aw.feat = e->feat_job;
      aw.win = e->win;
      aw.nb_search = nb_s;
      nb = nb_s + (int)((e->BN + NLW_WAVES * NL_FEAT_ROWS - 1) / (NLW_WAVES * NL_FEAT_ROWS));
    }
    aw.row_cap = e->row_cap > LB_MAX_ROW ? e->row_cap : LB_MAX_ROW;
    const size_t lds = sizeof(int) * 2 * NLW_WAVES * (size_t)aw.row_cap;
    if (e->g.f32) {  // dtype=float32 geometry: every result rounded to float
      if (lds > 48 * 1024)
        (void)hipFuncSetAttribute((const void*)k_nlw<MODE, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((k_nlw<MODE, true>), dim3(nb), dim3(64 * NLW_WAVES), lds, e->stream, e->g, e->BN, e->ctrl, aw);
    } else {
      if (lds > 48 * 1024)
        (void)hipFuncSetAttribute((const void*)k_nlw<MODE, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      hipLaunchKernelGGL((k_nlw<MODE>), dim3(nb), dim3(64 * NLW_WAVES), lds, e->stream, e->g, e->BN, e->ctrl, aw);
    }
    return;
  }
  const int ncell_tot = e->g.B * e->g.ncells;
  lb_nl_args ac = a;
  if (ride) {
    ac.feat = e->feat_job;
    ac.win = e->win;
    ac.nb_search = ncell_tot;
  }
  const int ride64 = ride ? (int)((e->BN + NL_FEAT_ROWS - 1) / NL_FEAT_ROWS) : 0;
  const int ride256 = ride ? (int)((e->BN + 4 * NL_FEAT_ROWS - 1) / (4 * NL_FEAT_ROWS)) : 0;
#define LB_NL_CASE(C)                                                                                       \
  case C:                                                                                                   \
    hipLaunchKernelGGL((k_nl<MODE, 64, C>), dim3(ncell_tot + ride64), dim3(64), 0, e->stream, e->g, e->BN,  \
                       e->ctrl, ac);                                                                        \
    break;
  switch (small) {
    LB_NL_CASE(128)
    LB_NL_CASE(256)
    LB_NL_CASE(384)
    LB_NL_CASE(512)
    LB_NL_CASE(640)
    LB_NL_CASE(768)
    LB_NL_CASE(896)
    LB_NL_CASE(1024)
    default:
      hipLaunchKernelGGL((k_nl<MODE, 256, LB_MAX_STENCIL_CAND>), dim3(ncell_tot + ride256), dim3(256), 0,
                         e->stream, e->g, e->BN, e->ctrl, ac);
  }
#undef LB_NL_CASE
}

int lbk_nl_build(lb_engine* e, bool want_efeat64) {
  const lb_geom& g = e->g;
  const int64_t BN = e->BN;
  const int ncell_tot = g.B * g.ncells;
  hipStream_t s = e->stream;
  const int frozen = e->e_cap > 0;

  // LB_SMALL_FUSED=0: always the multi-launch bookkeeping (lb_fused_launches, lb_internal.h)
  const bool small_ok = lb_fused_launches();
  const bool small_rows = small_ok && BN <= LB_SMALL_N;
  const bool small_cells = small_rows && ncell_tot <= LB_SMALL_N;
  // one trajectory of <= 4096 particles on the update path: the whole build is ONE launch (k_nl_small)
  const bool one_ok = small_ok;
  const int nls_npad = (int)((BN + 63) / 64 * 64);
  const int nls_waves = (int)std::min<int64_t>(NLS_WAVES, std::max<int64_t>(1, (BN + 255) / 256));
  const int64_t nls_tab = g.use_cell_list ? (int64_t)(g.ncell[0] + g.ncell[1] + (g.dim == 3 ? g.ncell[2] : 0)) * (nls_npad / 64) : 0;
  const size_t nls_lds = (size_t)nls_npad * (8 * g.dim + 4) + 8 * (size_t)nls_tab + sizeof(int) * ((size_t)nls_waves * NLS_CAND + 32);
  if (one_ok && small_ok && frozen && g.B == 1 && (g.dim == 2 || g.dim == 3) && BN <= LB_SMALL_N && !e->nl_dense && !e->nl_one_off && nls_lds <= 150 * 1024 &&
      (!g.use_cell_list || (int64_t)e->cell_capacity * g.nstencil <= NLS_CAND) &&
      e->nl_wg_sum && g.ncell[0] < 2048 && g.ncell[1] < 2048 && g.ncell[2] < 1024 && NLS_CAND >= LB_MAX_ROW) {
    lb_tic(e, LB_T_NEIGH);
    lb_nls_args a{};
    a.win = e->win;
    a.senders = e->senders;
    a.receivers = e->receivers;
    a.efeat = e->efeat;
    a.efeat64 = want_efeat64 ? e->efeat64 : nullptr;
    a.deg = e->deg;
    a.row_ptr = e->row_ptr;
    a.overflow = e->overflow;
    a.nedges_b = e->nedges_b;
    a.wg_sum = e->nl_wg_sum;
    a.e_alloc = e->e_alloc;
    a.e_cap = e->e_cap;
    a.cell_capacity = e->cell_capacity;
    a.npad = nls_npad;
    a.host_flag = e->host_flag_dev;
#ifdef LB_MS_STAMPS  // debug builds only: wall-clock stamps of three workgroups
    static const bool nls_dbg = true;
#else
    static const bool nls_dbg = false;
#endif
    static long long* dbg_dev = nullptr;
    if (nls_dbg && !dbg_dev) LB_HIP(hipMalloc((void**)&dbg_dev, sizeof(long long) * 32));
    a.dbg = nls_dbg ? dbg_dev : nullptr;
    const int nwv = nls_waves;
    const size_t lds = nls_lds;
    const int nb = (int)((BN + nwv - 1) / nwv);
    if (e->feat_job.xnode && !want_efeat64) {
      a.feat = e->feat_job;
      e->feat_done = true;
    }
  // (the dynamic-LDS limit is raised ONCE per kernel instance - the first launch after an allocation runs outside a
  // hipGraph capture - to the 150 KiB the routing above admits)
#define LB_NLS_LAUNCH(F, D)                                                                                      \
  do {                                                                                                          \
    static bool raised = false;                                                                                 \
    if (!raised) {                                                                                              \
      (void)hipFuncSetAttribute((const void*)k_nl_small<F, D>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); \
      raised = true;                                                                                            \
    }                                                                                                           \
    hipLaunchKernelGGL((k_nl_small<F, D>), dim3(nb), dim3(64 * nwv), lds, s, g, e->ctrl, a);                    \
  } while (0)
    if (g.f32) {
      if (g.dim == 3) LB_NLS_LAUNCH(true, 3); else LB_NLS_LAUNCH(true, 2);
    } else {
      if (g.dim == 3) LB_NLS_LAUNCH(false, 3); else LB_NLS_LAUNCH(false, 2);
    }
#undef LB_NLS_LAUNCH
    if (nls_dbg) {
      long long h[32];
      LB_HIP(hipStreamSynchronize(s));
      LB_HIP(hipMemcpy(h, dbg_dev, sizeof(h), hipMemcpyDeviceToHost));
      static int n_print = 0;
      if (n_print++ % 16 == 8)
        for (int w = 0; w < 3; ++w) {
          fprintf(stderr, "k_nl_small wg%d (10 ns ticks from wg0 start):", w);
          for (int k = 0; k < 8; ++k) fprintf(stderr, " %lld", h[w * 8 + k] - h[0]);
          fprintf(stderr, "  | entry wg0 %lld, last wg %lld\n", h[24] - h[0], h[25] - h[0]);
        }
    }
    lb_toc(e);
    LB_HIP(hipGetLastError());
    return LB_OK;
  }
  // one trajectory of up to 8192 particles with a cell list whose masks fit LDS: k_nl_mid (one launch)
  {
    const int npad_m = (int)((BN + 63) / 64 * 64);
    const int64_t tab_m = (int64_t)(g.ncell[0] + g.ncell[1] + (g.dim == 3 ? g.ncell[2] : 0)) * (npad_m / 64);
    const size_t lds_m = 8 * (size_t)tab_m + sizeof(int) * ((size_t)npad_m + NLM_WAVES * NLM_ROWBUF + 48);
    const bool mid_ok = small_ok;
    if (mid_ok && one_ok && small_ok && frozen && g.B == 1 && BN > LB_SMALL_N && BN <= NLM_N && g.use_cell_list &&
        (g.dim == 2 || g.dim == 3) && !e->nl_dense && !e->nl_one_off && lds_m <= 150 * 1024 &&
        (int64_t)e->cell_capacity * g.nstencil <= NLM_ROWBUF - LB_MAX_ROW * (NLM_RPW - 1) && e->nl_wg_sum && g.ncell[0] < 2048 &&
        g.ncell[1] < 2048 && g.ncell[2] < 1024) {
      lb_tic(e, LB_T_NEIGH);
      lb_nls_args a{};
      a.win = e->win;
      a.senders = e->senders;
      a.receivers = e->receivers;
      a.efeat = e->efeat;
      a.efeat64 = want_efeat64 ? e->efeat64 : nullptr;
      a.deg = e->deg;
      a.row_ptr = e->row_ptr;
      a.overflow = e->overflow;
      a.nedges_b = e->nedges_b;
      a.wg_sum = e->nl_wg_sum;
      a.e_alloc = e->e_alloc;
      a.e_cap = e->e_cap;
      a.cell_capacity = e->cell_capacity;
      a.npad = npad_m;
      a.host_flag = e->host_flag_dev;
      if (e->feat_job.xnode && !want_efeat64) {
        a.feat = e->feat_job;
        e->feat_done = true;
      }
      const int nb = (int)((BN + NLM_WAVES * NLM_RPW - 1) / (NLM_WAVES * NLM_RPW));
#define LB_NLM_LAUNCH(F, D)                                                                                     \
  do {                                                                                                          \
    static bool raised = false;                                                                                 \
    if (!raised) {                                                                                              \
      (void)hipFuncSetAttribute((const void*)k_nl_mid<F, D>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024); \
      raised = true;                                                                                            \
    }                                                                                                           \
    hipLaunchKernelGGL((k_nl_mid<F, D>), dim3(nb), dim3(64 * NLM_WAVES), lds_m, s, g, e->ctrl, a);             \
  } while (0)
      if (g.f32) {
        if (g.dim == 3) LB_NLM_LAUNCH(true, 3); else LB_NLM_LAUNCH(true, 2);
      } else {
        if (g.dim == 3) LB_NLM_LAUNCH(false, 3); else LB_NLM_LAUNCH(false, 2);
      }
#undef LB_NLM_LAUNCH
      lb_toc(e);
      LB_HIP(hipGetLastError());
      return LB_OK;
    }
  }
  lb_tic(e, LB_T_CELLS);
  // (one mid-size trajectory that the single-launch builds above refused, e.g. DAM2D: binning still in one launch)
  const bool cells1_ok = small_ok;
  const bool mid_cells = cells1_ok && small_ok && frozen && g.B == 1 && BN <= LB_CELLS1_N && ncell_tot <= LB_CELLS1_NCELL;
  // batches: one workgroup per trajectory, one launch (LB_SMALL_FUSED=0: the five-launch counting sort)
  const bool cells_traj_ok = small_ok;
  // (measured, profiles/r04_ab_cells_traj.txt: 2.5 k particles x 8 23.1 -> 17.5 us, 5.7 k x 8 27.7 -> 22.7 us, 8 k x 8 30.8 -> 32.5 us -
  // one workgroup per trajectory is bound by its scattered stores from ONE CU; above 6 k particles the five launches stay)
  const bool traj_cells = cells_traj_ok && frozen && !small_cells && g.B > 1 && g.use_cell_list && g.N <= 6144 &&
                          g.ncells <= LB_CELLS1_NCELL;
  // frozen capacities + a search kernel that walks CELLS: fixed-stride slots, two launches (k_cell_zero, k_cell_bin)
  const int64_t strided_slots = (int64_t)ncell_tot * e->cell_capacity;
#ifdef LB_NO_STRIDED   // (A/B builds: tools/build_variant.sh csr -DLB_NO_STRIDED)
  const bool strided = false;
#else
  const bool strided = small_ok && frozen && g.use_cell_list && e->cell_capacity > 0 && lb_nl_kernel_kind(e) != 1 &&
                       strided_slots < ((int64_t)1 << 30);
#endif
  e->cells_strided = strided;
  if (strided) {
    if (strided_slots > e->cell_slots) {   // (first build after an allocation: never inside a graph capture)
      LB_HIP(hipStreamSynchronize(s));
      if (e->cell_part) (void)hipFree(e->cell_part);
      if (e->cpos) (void)hipFree(e->cpos);
      e->cell_part = nullptr;
      e->cpos = nullptr;
      e->cell_slots = strided_slots + strided_slots / 4;
      LB_HIP(hipMalloc((void**)&e->cell_part, sizeof(int32_t) * (size_t)e->cell_slots));
      LB_HIP(hipMalloc((void**)&e->cpos, sizeof(double) * (size_t)g.dim * (size_t)e->cell_slots));
    }
    e->cells_traj_ready = false;   // (cell_count is rewritten: k_cells_traj's occupancy slots must be zeroed again after this)
    hipLaunchKernelGGL(k_cell_zero, dim3((ncell_tot + 255) / 256), dim3(256), 0, s, e->ctrl, e->cell_count, ncell_tot);
    hipLaunchKernelGGL(k_cell_bin, dim3((unsigned)((BN + 255) / 256)), dim3(256), 0, s, g, BN, e->win, e->ctrl, e->cell_of,
                       e->cell_count, e->cell_part, e->cpos, e->cell_capacity, strided_slots);
  } else if (small_cells) {
    hipLaunchKernelGGL((k_cells_small<LB_SMALL_N / LB_SMALL_T>), dim3(1), dim3(LB_SMALL_T), sizeof(int) * (size_t)ncell_tot, s, g,
                       BN, e->win, e->ctrl, e->cell_of, e->cell_start, e->cell_part, e->cpos, ncell_tot);
  } else if (traj_cells) {
    static bool raised_t = false;
    if (!raised_t) {
      (void)hipFuncSetAttribute((const void*)k_cells_traj<LB_CELLS1_PER>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                sizeof(int) * LB_CELLS1_NCELL);
      raised_t = true;
    }
    if (!e->cells_traj_ready) {  // occ[0 .. B] (max occupancy per trajectory, arrival ticket) live in the unused cell_count
      LB_HIP(hipMemsetAsync(e->cell_count, 0, sizeof(int32_t) * (size_t)(g.B + 1), s));
      e->cells_traj_ready = true;
    }
    hipLaunchKernelGGL((k_cells_traj<LB_CELLS1_PER>), dim3(g.B), dim3(LB_SMALL_T), sizeof(int) * (size_t)g.ncells, s, g, BN,
                       e->win, e->ctrl, e->cell_of, e->cell_start, e->cell_part, e->cpos, e->cell_count);
  } else if (mid_cells) {
    static bool raised = false;
    if (!raised) {
      (void)hipFuncSetAttribute((const void*)k_cells_small<LB_CELLS1_PER>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                sizeof(int) * LB_CELLS1_NCELL);
      raised = true;
    }
    hipLaunchKernelGGL((k_cells_small<LB_CELLS1_PER>), dim3(1), dim3(LB_SMALL_T), sizeof(int) * (size_t)ncell_tot, s, g, BN,
                       e->win, e->ctrl, e->cell_of, e->cell_start, e->cell_part, e->cpos, ncell_tot);
  } else {
    e->cells_traj_ready = false;  // (the counting sort uses cell_count: k_cells_traj's occ[] must be zeroed again after it)
    LB_HIP(hipMemsetAsync(e->cell_count, 0, sizeof(int32_t) * 2 * (size_t)ncell_tot, s));
    // (max_cell_occ, max_deg, row_overflow are reset by k_cell_count)
    const int nb = (int)((BN + 255) / 256);
    hipLaunchKernelGGL(k_cell_count, dim3(nb), dim3(256), 0, s, g, BN, e->win, e->ctrl, e->cell_of,
                       e->cell_count);
    const int nsb_c = (ncell_tot + SCAN_CHUNK - 1) / SCAN_CHUNK;
    hipLaunchKernelGGL(k_scan_partials, dim3(nsb_c), dim3(SCAN_THREADS), 0, s, e->cell_count, ncell_tot,
                       e->scan_part, e->ctrl, &e->ctrl->max_cell_occ);
    hipLaunchKernelGGL(k_scan_apply, dim3(nsb_c), dim3(SCAN_THREADS), 0, s, e->cell_count,
                       e->cell_start, ncell_tot, e->scan_part, e->ctrl);
    hipLaunchKernelGGL(k_cell_fill, dim3(nb), dim3(256), 0, s, g, BN, e->win, e->ctrl, e->cell_of,
                       e->cell_start, e->cell_fill, e->cell_part, e->cpos);
  }
  lb_toc(e);

  lb_tic(e, LB_T_NEIGH);
  // one wave per cell when the frozen cell capacity bounds the stencil (a cell holding more than
  // cell_capacity particles flags overflow anyway); the 256-thread / 2048-candidate variant otherwise
  // (the LDS footprint of the staged stencil sets how many cells a CU keeps in flight)
  const int64_t cand_cap = (int64_t)e->cell_capacity * g.nstencil;
  const int small = (frozen && g.use_cell_list && cand_cap <= NL_SMALL_MAXC)
                        ? (int)std::max<int64_t>(128, (cand_cap + 127) / 128 * 128) : 0;
  const bool rows = frozen && e->maxd > 0 && e->tmp_send && (!want_efeat64 || e->tmp_feat64);
  lb_nl_args a{};
  a.cell_of = e->cell_of;
  a.cell_start = e->cell_start;
  a.cell_part = e->cell_part;
  a.cpos = e->cpos;
  a.cell_cnt = e->cell_count;
  a.cell_cap = strided ? e->cell_capacity : 0;
  a.cstride = strided ? strided_slots : BN;
  a.deg = e->deg;
  a.row_ptr = e->row_ptr;
  a.e_alloc = e->e_alloc;
  a.maxd = e->maxd;
  if (rows) {
    a.senders = e->tmp_send;
    a.efeat = e->tmp_feat;
    a.efeat64 = want_efeat64 ? e->tmp_feat64 : nullptr;
    lb_launch_nl<NL_ROWS>(e, small, a);
  } else {
    lb_launch_nl<NL_COUNT>(e, small, a);
  }
  // update path of a small batch: scan + finish + compaction in one launch (LB_SMALL_FUSED=0: the separate launches)
  const bool cscan_ok = small_ok;
  if (rows && cscan_ok && small_ok && BN <= LB_CSCAN_N && g.B <= 64) {
    hipLaunchKernelGGL(k_nl_compact_scan, dim3((int)((BN + 15) / 16)), dim3(256), 0, s, g, BN, e->ctrl, e->deg, e->row_ptr,
                       e->maxd, e->tmp_send, e->tmp_feat, want_efeat64 ? e->tmp_feat64 : (const double*)nullptr,
                       e->senders, e->receivers, e->efeat, want_efeat64 ? e->efeat64 : (double*)nullptr, e->e_alloc,
                       e->overflow, e->nedges_b, e->cell_capacity, e->e_cap, e->host_flag_dev);
    lb_toc(e);
    LB_HIP(hipGetLastError());
    return LB_OK;
  }
  if (small_rows) {
    hipLaunchKernelGGL(k_rows_small, dim3(1), dim3(LB_SMALL_T), 0, s, g, e->deg, e->row_ptr, (int)BN, e->ctrl,
                       e->overflow, e->nedges_b, e->cell_capacity, e->e_cap, e->e_alloc, frozen,
                       e->host_flag_dev);
  } else {
    const int nsb_r = (int)((BN + SCAN_CHUNK - 1) / SCAN_CHUNK);
    hipLaunchKernelGGL(k_scan_partials, dim3(nsb_r), dim3(SCAN_THREADS), 0, s, e->deg, (int)BN,
                       e->scan_part, e->ctrl, &e->ctrl->max_deg);
    hipLaunchKernelGGL(k_scan_apply, dim3(nsb_r), dim3(SCAN_THREADS), 0, s, e->deg, e->row_ptr, (int)BN,
                       e->scan_part, e->ctrl);
    hipLaunchKernelGGL(k_row_finish, dim3(1), dim3(256), 0, s, g, e->row_ptr, (int)BN, e->ctrl,
                       e->overflow, e->nedges_b, e->cell_capacity, e->e_cap, e->e_alloc, frozen,
                       e->host_flag_dev);
  }
  if (rows) {
    hipLaunchKernelGGL(k_nl_compact, dim3((int)((BN + 15) / 16)), dim3(256), 0, s, BN, e->ctrl, e->deg,
                       e->row_ptr, e->maxd, e->tmp_send, e->tmp_feat,
                       want_efeat64 ? e->tmp_feat64 : (const double*)nullptr, e->senders, e->receivers,
                       e->efeat, want_efeat64 ? e->efeat64 : (double*)nullptr, e->e_alloc);
    lb_toc(e);
    LB_HIP(hipGetLastError());
    return LB_OK;
  }
  if (!frozen) {
    // allocate path (host-synchronous by contract): size the edge buffers before the fill pass
    LB_HIP(hipMemcpyAsync(e->ctrl_host, e->ctrl, sizeof(lb_ctrl), hipMemcpyDeviceToHost, s));
    std::vector<int32_t> occ(g.B);
    LB_HIP(hipMemcpyAsync(occ.data(), e->nedges_b, sizeof(int32_t) * g.B, hipMemcpyDeviceToHost, s));
    LB_HIP(hipStreamSynchronize(s));
    // dense fall-back: a stencil with more candidates than the staged kernel holds, or a particle with more
    // neighbors than its row buffer, switches to the wave-per-receiver kernel with a row buffer of the measured size
    if (e->ctrl_host->density_error == 1 && !e->nl_dense) {
      e->nl_dense = true;
      e->ctrl_host->density_error = 0;
      LB_HIP(hipMemcpyAsync(&e->ctrl->density_error, &e->ctrl_host->density_error, sizeof(int32_t), hipMemcpyHostToDevice, s));
      return lbk_nl_build(e, want_efeat64);  // count again with the kernel that has no candidate limit
    }
    if (e->ctrl_host->max_deg > LB_MAX_ROW) e->nl_dense = true;
    if (e->nl_dense) {
      e->row_cap = std::max(LB_MAX_ROW, (e->ctrl_host->max_deg + 63) / 64 * 64);
      if (e->row_cap > LB_MAX_ROW_DENSE) e->row_cap = LB_MAX_ROW_DENSE;  // (> that many neighbors: LB_ERR_DENSITY)
    }
    // room for this list AND for the capacity lb_nl_allocate is about to freeze
    // (B * int(max_b occupancy * multiplier)): growing later would drop the list just built
    int32_t occ_max = 0;
    for (int b = 0; b < g.B; ++b) occ_max = occ[b] > occ_max ? occ[b] : occ_max;
    const double mult = e->desc.capacity_multiplier > 0 ? e->desc.capacity_multiplier : 1.25;
    int64_t need = (int64_t)e->ctrl_host->n_edges_unclamped;
    const int64_t frozen_need = (int64_t)(occ_max * mult) * g.B;
    if (frozen_need > need) need = frozen_need;
    int rc = lb_ensure_edges(e, need);
    if (rc) return rc;
    // the scan clamped n_edges_total against the old allocation: refresh it
    e->ctrl_host->n_edges_total = e->ctrl_host->n_edges_unclamped;
    LB_HIP(hipMemcpyAsync(&e->ctrl->n_edges_total, &e->ctrl_host->n_edges_total, sizeof(int32_t),
                          hipMemcpyHostToDevice, s));
    a.e_alloc = e->e_alloc;
  }
  a.senders = e->senders;
  a.receivers = e->receivers;
  a.efeat = e->efeat;
  a.efeat64 = want_efeat64 ? e->efeat64 : nullptr;
  lb_launch_nl<NL_FILL>(e, small, a);
  lb_toc(e);
  LB_HIP(hipGetLastError());
  return LB_OK;
}

// NeighborList.idx export: (B, 2, E_cap), local ids, pad = N.
__global__ void k_nl_export(lb_geom g, const int32_t* __restrict__ row_ptr,
                            const int32_t* __restrict__ senders,
                            const int32_t* __restrict__ receivers, int32_t e_cap, int64_t e_alloc,
                            int32_t* __restrict__ idx_out, int32_t* __restrict__ n_edges_out) {
  const int b = blockIdx.y;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  const int base = row_ptr[b * g.N];
  const int eb = row_ptr[(b + 1) * g.N] - base;
  if (k == 0 && n_edges_out) n_edges_out[b] = eb;
  if (k >= e_cap) return;
  int r = g.N, s = g.N;
  if (k < eb && (int64_t)base + k < e_alloc) {
    r = receivers[base + k] - b * g.N;
    s = senders[base + k] - b * g.N;
  }
  idx_out[((int64_t)b * 2 + 0) * e_cap + k] = r;
  idx_out[((int64_t)b * 2 + 1) * e_cap + k] = s;
}

int lbk_nl_export(lb_engine* e, int32_t* idx_out, int32_t* n_edges_out) {
  const int ecap = e->e_cap;
  dim3 grid((ecap + 255) / 256 > 0 ? (ecap + 255) / 256 : 1, e->g.B);
  hipLaunchKernelGGL(k_nl_export, grid, dim3(256), 0, e->stream, e->g, e->row_ptr, e->senders,
                     e->receivers, ecap, e->e_alloc, idx_out, n_edges_out);
  LB_HIP(hipGetLastError());
  return LB_OK;
}

__global__ void k_efeat_export(lb_geom g, const int32_t* __restrict__ row_ptr,
                               const double* __restrict__ efeat64, int32_t e_cap, int64_t e_alloc,
                               double* __restrict__ rel_disp, double* __restrict__ rel_dist) {
  const int b = blockIdx.y;
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= e_cap) return;
  const int base = row_ptr[b * g.N];
  const int eb = row_ptr[(b + 1) * g.N] - base;
  double v[4] = {0, 0, 0, 0};
  if (k < eb && (int64_t)base + k < e_alloc)
    for (int d = 0; d < 4; ++d) v[d] = efeat64[((int64_t)base + k) * 4 + d];
  const int64_t o = (int64_t)b * e_cap + k;
  for (int d = 0; d < g.dim; ++d) rel_disp[o * g.dim + d] = v[d];
  rel_dist[o] = v[3];
}

int lbk_edge_features_export(lb_engine* e, double* rel_disp, double* rel_dist) {
  const int ecap = e->e_cap;
  dim3 grid((ecap + 255) / 256 > 0 ? (ecap + 255) / 256 : 1, e->g.B);
  hipLaunchKernelGGL(k_efeat_export, grid, dim3(256), 0, e->stream, e->g, e->row_ptr, e->efeat64,
                     ecap, e->e_alloc, rel_disp, rel_dist);
  LB_HIP(hipGetLastError());
  return LB_OK;
}
