// Per-launch log of the library (diagnostic; off unless sdf_launch_log(1) was called): every kernel launch of every entry point goes
// through SDF_LAUNCH, which - while the log is on - brackets the launch with two HIP events on its own stream and notes the kernel,
// its grid (workgroups), block and dynamic LDS.  sdf_launch_log_read() returns the records with their durations: per-kernel time and
// CHIP time (compute units held x duration) of a forward without a profiler attached (bench.py `roofline.by_time`).  Off, the cost is one
// relaxed atomic load per launch.
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>

extern std::atomic<int> g_sdf_launch_log_on;
void sdf_launch_log_begin(const void* fn, dim3 grid, dim3 block, size_t lds, hipStream_t s);
void sdf_launch_log_end(hipStream_t s);

#define SDF_LAUNCH(kernel, grid, block, lds, stream, ...)                                                           \
  do {                                                                                                              \
    const bool sdf_lg__ = g_sdf_launch_log_on.load(std::memory_order_relaxed) != 0;                                 \
    if (sdf_lg__) sdf_launch_log_begin(reinterpret_cast<const void*>(kernel), dim3(grid), dim3(block), (size_t)(lds), (stream)); \
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                              \
    if (sdf_lg__) sdf_launch_log_end(stream);                                                                       \
  } while (0)
