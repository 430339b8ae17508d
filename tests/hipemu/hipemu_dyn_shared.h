// TEST INFRASTRUCTURE ONLY (see hip/hip_runtime.h in this directory): the dynamic LDS of a workgroup.  `extern __shared__ T s_dyn[]` inside a
// kernel (kernels_score.h) becomes `extern thread_local T s_dyn[]` under the emulator's qualifiers; this is its definition -- one per OS thread
// (a workgroup's fibers all run on one), the size of the real machine's LDS -- force-included (-include) into the single translation unit of the
// emulated build, so that the compiler sees a thread-local without a dynamic initialiser.
#pragma once
#include <cstdint>
thread_local __attribute__((aligned(16))) uint8_t s_dyn[160 << 10];
