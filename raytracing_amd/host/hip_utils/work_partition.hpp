// hip_utils/work_partition.hpp -- rt::make_work_partition
// (reference: include/cuda_utils/work_partition.cuh:5-24), plus the strip partition of a
// frame over GPUs, which has no reference counterpart.
#pragma once
#include <cstddef>

#include "mrt.h"

namespace rt {

struct WorkPartition {
  bool is_uneven_partition;
  std::size_t elements_per_thread;
  std::size_t elements_in_last_thread;
};

inline WorkPartition make_work_partition(std::size_t work_elements, std::size_t num_threads)
{
  WorkPartition part;
  part.elements_in_last_thread = work_elements % num_threads;
  part.is_uneven_partition = part.elements_in_last_thread != 0u;
  part.elements_per_thread = work_elements / num_threads + (part.is_uneven_partition ? 1 : 0);
  return part;
}

// strips of strip_rows rows dealt round-robin: strip s -> rank s % num_ranks
// (rank_span > 1: the strips of that many consecutive ranks in one render -- a weighted multi-GPU frame's rank 0)
inline mrt_partition make_strip_partition(int rank, int num_ranks, int strip_rows = 8, int rank_span = 1)
{
  return mrt_partition{rank, num_ranks, strip_rows, 0, rank_span};
}

} // namespace rt
