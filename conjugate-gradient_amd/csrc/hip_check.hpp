// hip_check.hpp -- runtime-error convention of the reference (common.h:7-17): print
// "Error: file:line, code: N, reason: ..." to stderr and exit(1).  There are no error
// codes in the API; logic errors abort through assert as in the reference.
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(call)                                                                      \
  {                                                                                      \
    const hipError_t error_ = (call);                                                    \
    if (error_ != hipSuccess)                                                            \
    {                                                                                    \
      fprintf(stderr, "Error: %s:%d, ", __FILE__, __LINE__);                             \
      fprintf(stderr, "code: %d, reason: %s\n", (int)error_, hipGetErrorString(error_)); \
      exit(1);                                                                           \
    }                                                                                    \
  }

// precondition failures of the C-ABI: same exit path, no HIP error code
#define CG_REQUIRE(cond, msg)                                              \
  {                                                                        \
    if (!(cond))                                                           \
    {                                                                      \
      fprintf(stderr, "Error: %s:%d, reason: %s\n", __FILE__, __LINE__, msg); \
      exit(1);                                                             \
    }                                                                      \
  }
