// The FENERF_TAPE_F32_W instantiations of the backward chain kernel (fenerf_siren_bwd16w.hip, TAPE = 2: the fp32 tape, but the second FiLM
// sum is left to the weight-gradient stage) as a translation unit of their own, like fenerf_siren_bwd16w_t16.hip.  Defines
// fenerf::launch_siren_backward16w_w only.
#define FENERF_BW16_T16 2
#include "fenerf_siren_bwd16w.hip"
