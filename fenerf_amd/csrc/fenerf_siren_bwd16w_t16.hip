// The 16-bit-tape instantiations of the backward chain kernel (fenerf_siren_bwd16w.hip, template parameter T16; fenerf_layout.h
// "16-bit tape"), as a translation unit of their own: the chain kernel's 32 instantiations per tape format take over a minute to
// compile, and the two formats compile side by side.  Defines fenerf::launch_siren_backward16w_t16 only.
#define FENERF_BW16_T16 1
#include "fenerf_siren_bwd16w.hip"
