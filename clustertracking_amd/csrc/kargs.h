// kargs.h -- what the translation units of the engine share: kernel arguments and the
// functions through which ctrefine.hip (host runtime) gets at the kernels compiled elsewhere.
// The engine is split into several .hip files only so that they compile in parallel.
#ifndef CTREFINE_KARGS_H
#define CTREFINE_KARGS_H

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "ctrefine.h"

struct KArgs {
  ctr_problem prob;
  const void* frames;
  int32_t frame_dtype;
  int32_t n_bin;
  int64_t shape[3];
  int64_t frame_elems;
  const int32_t* frame_index;
  const int32_t* feat_offset;
  const double* params;
  const double* low;
  const double* high;
  double* params_out;
  double* cost;
  int32_t* status;
  int32_t* n_rounds;
  int32_t* n_iter;
  double* params_std;    // [N, n_params] or nullptr (ctr_batch.params_std)
  const double* fmax;
  const int32_t* order;  // cluster ids of this bin
  // small kernel only: part of the bin a launch takes.  split == nullptr: all of it;
  // split_part 1: entries [0, *split) (the likely slow fits, front_load_kernel's count),
  // split_part 2: entries [*split, n_bin)
  const int32_t* split;
  int32_t split_part;
};

// a kernel of another translation unit: host-side handle for hipLaunchKernel /
// hipFuncSetAttribute, dynamic LDS bytes and workgroup size
struct KernelInfo {
  const void* fn;
  size_t smem;
  int threads;
};

// refine_block_kernel<ND, ISO, NT, W>: nt = 1..8; throughput != 0: the fewest wavefronts
KernelInfo ctr_block_kernel_2d(int iso, int nt, int throughput);
KernelInfo ctr_block_kernel_3d(int iso, int nt, int throughput);
// refine_small_kernel<ND, NF, ISO, SG>(KArgs, int* counter); nullptr if not instantiated
const void* ctr_small_kernel(int ndim, int nf, int iso, int sg);

#endif  // CTREFINE_KARGS_H
