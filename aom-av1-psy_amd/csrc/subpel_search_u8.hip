// sub-pel search kernels for uint8_t planes (see subpel_search.inc)
#define AOMHIP_PIX_T uint8_t
#define AOMHIP_SUBPEL_LAUNCH launch_subpel_u8
#include "subpel_search.inc"
