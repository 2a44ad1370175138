// Dev aid: lets a single kernel file of csrc/ be linked as a standalone debug library (scripts/_dbg/*.so).
#include <hip/hip_runtime.h>
namespace dmp { void set_last_hip_error(hipError_t) {} }
