/*
 * pfv_hip.h -- the whole C ABI of libpfv_hip.so: pfv_hip_core.h (the drop-in boundary: the six plane operators, the sessions,
 * enc::Encoder / dec::Decoder as pfv_encoder / pfv_decoder -- what INTEGRATION.md binds) + pfv_hip_ext.h (throughput forms and
 * diagnostics on top of it).  A binding of the reference includes pfv_hip_core.h alone.
 */
#ifndef PFV_HIP_H
#define PFV_HIP_H
#include "pfv_hip_core.h"
#include "pfv_hip_ext.h"
#endif /* PFV_HIP_H */
