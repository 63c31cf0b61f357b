// sub-pel search kernels for uint16_t planes (see subpel_search.inc)
#define AOMHIP_PIX_T uint16_t
#define AOMHIP_SUBPEL_LAUNCH launch_subpel_u16
#include "subpel_search.inc"
