// full-pel search kernels for uint16_t planes (see fullpel_search.inc)
#define AOMHIP_PIX_T uint16_t
#define AOMHIP_FPS_LAUNCH launch_fps_u16
#include "fullpel_search.inc"
