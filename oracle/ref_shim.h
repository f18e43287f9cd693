/*
 * ref_shim.h -- the launch-index names the reference's kernel body reads.
 *
 * TEST INFRASTRUCTURE.  Used only by `make -C oracle ref`, which compiles the body of
 * GpuIntegrate from where it lies (/root/reference/src/tsdf.cu, the lines from
 * `__global__` to the function's closing brace) with g++ for the host.  That body is
 * scalar C apart from the `__global__` qualifier and the two index variables below;
 * it calls nothing but roundf/fmin from libm.  No header, library or tool of the
 * reference is replaced: the rest of tsdf.cu (the class that needs OpenCV and the CUDA
 * runtime) is not built.
 */
#ifndef ORACLE_REF_SHIM_H
#define ORACLE_REF_SHIM_H
#include <math.h>
struct ref_idx3 { int x, y, z; };
extern thread_local ref_idx3 blockIdx, threadIdx; /* one "thread" at a time per host thread */
#define __global__
#endif
