// full-pel search kernels for uint8_t planes (see fullpel_search.inc)
#define AOMHIP_PIX_T uint8_t
#define AOMHIP_FPS_LAUNCH launch_fps_u8
#include "fullpel_search.inc"
