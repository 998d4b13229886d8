// Finalising a two-stage per-channel reduction INSIDE the kernel that produced its first stage.
//
// Every BatchNorm of the training step (src/nn/layer_factory.py:94-106,125-158,161-178,225-265) needs its batch
// statistics before the next kernel can run, and every BatchNorm backward its {sum g, sum g*xhat}.  The producing
// kernels leave one row of partial sums per workgroup; summing the rows used to be a launch of its own - 74 + 62
// launches of 5-8 us per headline step (8 % of it), 250 of the 885 launches of the CVPR 321x321 step.  Here the
// workgroup that finishes LAST does it, at the end of the producing kernel:
//
//   * a workgroup writes its row with agent-scope (sc1, write-through) stores, drains them (s_waitcnt vmcnt(0) in
//     every storing wave, then a workgroup barrier) and takes a ticket from its GROUP's counter (rows are dealt
//     round-robin to at most 32 groups);
//   * the last arriver of a group adds the group's rows in a fixed order in fp64 (agent-scope loads: they were
//     written through, nothing of them sits in this CU's L1 or this XCD's L2), leaves one group row, drains, and
//     takes a ticket from the top counter;
//   * the last of those adds the <= 32 group rows in order and finishes: BatchNorm statistics (mean, invstd,
//     scale, shift, running statistics - arithmetic of reduce.hip:bn_stats_finalize_t) or plain column sums.
//     Both counters are back at zero when the kernel ends (the host zeroes them once, at allocation).
//
// No release / acquire fence: an earlier attempt published the rows with __threadfence() and paid for the L2
// write-back of everything ELSE the kernel had dirtied (16.3 -> 20.0 ms per step).  Write-through rows + drained
// stores + a returning device-scope atomic is the fence-free form MI355X_MICROARCH.md lists as valid ("sc1 payload,
// vmcnt(0), flag"; the flag here is the ticket).  The order of every sum is fixed by row index, not by arrival:
// results are bit-reproducible.
#pragma once
#include "common.h"

#define NASSEG_TAIL_GROUPS 32
#define NASSEG_TAIL_WORDS (NASSEG_TAIL_GROUPS + 1)

struct TailArgs {
  unsigned* tickets;  // [NASSEG_TAIL_WORDS], zero before and after every launch; null: no tail
  float* rows;        // [nent * rpe][cols] first-stage rows (what the kernel writes anyway): rpe rows per workgroup
  float* lvl;         // [NASSEG_TAIL_GROUPS][2][cols] group rows (scratch: the 64 spare rows behind `rows`)
  int nent, rpe, cols, kind;  // kind 1: BatchNorm statistics (cols = 2 C: sum | sum of squares); 2: column sums -> out
  double M;
  float eps, momentum;
  const float* gamma;
  const float* beta;
  float* mean;
  float* invstd;
  float* scale;
  float* shift;
  float* running_mean;
  float* running_var;
  long long* nbt;
  float* out;
};

__device__ __forceinline__ void st_agent(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// two adjacent floats with one agent-scope (sc1) access
__device__ __forceinline__ float2 ld_agent2(const float* p) {
  const unsigned long long u = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED,
                                                 __HIP_MEMORY_SCOPE_AGENT);
  return make_float2(__uint_as_float((unsigned)u), __uint_as_float((unsigned)(u >> 32)));
}
__device__ __forceinline__ void st_agent2(float* p, float a, float b) {
  const unsigned long long u = (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32);
  __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Column sums over nr rows (row r at rowp(r), `cols` floats each, cols even and <= 512) by the 256 threads of a
// workgroup: thread (pair of columns c2, slice sl) adds rows sl, sl + slices, ... in fp64 with up to 16 loads in
// flight, the slices meet in LDS (red: 256 double2) in slice order.  The sums of column pair c2 = tid come back in
// the threads tid < cols / 2.  Order fixed by (r, slices): deterministic.
template <typename RowP>
__device__ __forceinline__ double2 tail_column_sums(RowP rowp, int nr, int cols, double2* red) {
  const int tid = threadIdx.x;
  const int ncol2 = cols >> 1;
  int slices = 256 / ncol2;
  if (slices < 1) slices = 1;
  const int c2 = tid % ncol2, sl = tid / ncol2;
  double2 s = make_double2(0.0, 0.0);
  if (sl < slices) {
    int r = sl;
    for (; r + 15 * slices < nr; r += 16 * slices) {
      float2 v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = ld_agent2(rowp(r + i * slices) + 2 * c2);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        s.x += (double)v[i].x;
        s.y += (double)v[i].y;
      }
    }
    for (; r < nr; r += slices) {
      const float2 v = ld_agent2(rowp(r) + 2 * c2);
      s.x += (double)v.x;
      s.y += (double)v.y;
    }
  }
  red[tid] = s;
  __syncthreads();
  double2 tot = make_double2(0.0, 0.0);
  if (tid < ncol2) {
    for (int q = 0; q < slices; ++q) {
      tot.x += red[q * ncol2 + tid].x;
      tot.y += red[q * ncol2 + tid].y;
    }
  }
  __syncthreads();
  return tot;
}

// Called by ALL threads of a 256-thread workgroup after the rows of its entry `ent` (< t.nent = the grid size;
// rows ent * rpe .. + rpe - 1) were written with st_agent.  lds: 4 KB + 16 B of the caller's shared memory,
// 16-byte aligned, no longer in use.
__device__ __forceinline__ void stats_tail(const TailArgs& t, int ent, void* lds) {
  double2* red = reinterpret_cast<double2*>(lds);
  unsigned* flag = reinterpret_cast<unsigned*>(red + 256);
  const int tid = threadIdx.x;
  const int G = t.nent < NASSEG_TAIL_GROUPS ? t.nent : NASSEG_TAIL_GROUPS;
  const int g = ent % G;
  const int members = (t.nent - g + G - 1) / G;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's row stores have reached memory
  __syncthreads();
  if (tid == 0) {
    const unsigned old = __hip_atomic_fetch_add(t.tickets + 1 + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = old == (unsigned)(members - 1);
  }
  __syncthreads();
  if (!*flag) return;
  // ---- last of group g: lvl[g] = sum of the rows of entries g, g + G, g + 2 G ... in that order ----
  {
    const int rpe = t.rpe, cols = t.cols;
    const float* rows = t.rows;
    const double2 s = tail_column_sums(
        [=](int r) {
          const int e = r / rpe;
          return rows + ((size_t)(g + e * G) * rpe + (r - e * rpe)) * cols;
        },
        members * rpe, cols, red);
    if (tid < (cols >> 1)) {
      // (a double as two floats: the group row costs the sum nothing)
      const float hx = (float)s.x, hy = (float)s.y;
      float* po = t.lvl + (size_t)g * 2 * cols + 2 * tid;
      st_agent2(po, hx, hy);
      st_agent2(po + cols, (float)(s.x - (double)hx), (float)(s.y - (double)hy));
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __hip_atomic_store(t.tickets + 1 + g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned old = __hip_atomic_fetch_add(t.tickets, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = old == (unsigned)(G - 1);
  }
  __syncthreads();
  if (!*flag) return;
  // ---- last of all: columns summed over the group rows (value, residue) in order, then the finish ----
  if (tid == 0) __hip_atomic_store(t.tickets, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  const int cols = t.cols;
  const float* lvl = t.lvl;
  const double2 s = tail_column_sums([=](int r) { return lvl + (size_t)r * cols; }, 2 * G, cols, red);
  // (columns 2 tid, 2 tid + 1 are in thread tid; the BatchNorm finish pairs column c with column C + c: via LDS)
  double* col = reinterpret_cast<double*>(red);
  if (tid < (cols >> 1)) {
    col[2 * tid] = s.x;
    col[2 * tid + 1] = s.y;
  }
  __syncthreads();
  if (t.kind == 2) {
    for (int c = tid; c < cols; c += 256) t.out[c] = (float)col[c];
    return;
  }
  const int C = cols >> 1;
  if (tid == 0 && t.nbt) *t.nbt += 1;
  for (int c = tid; c < C; c += 256) {
    const double s0 = col[c], s1 = col[C + c];
    const double mu = s0 / t.M;
    double var = s1 / t.M - mu * mu;
    if (var < 0.0) var = 0.0;
    const double is = 1.0 / sqrt(var + (double)t.eps);
    t.mean[c] = (float)mu;
    t.invstd[c] = (float)is;
    const double gm = t.gamma ? (double)t.gamma[c] : 1.0;
    const double bt = t.beta ? (double)t.beta[c] : 0.0;
    t.scale[c] = (float)(gm * is);
    t.shift[c] = (float)(bt - mu * gm * is);
    if (t.running_mean) {
      t.running_mean[c] = (float)((1.0 - t.momentum) * (double)t.running_mean[c] + t.momentum * mu);
      const double unb = t.M > 1.0 ? var * t.M / (t.M - 1.0) : var;
      t.running_var[c] = (float)((1.0 - t.momentum) * (double)t.running_var[c] + t.momentum * unb);
    }
  }
}
